# Development helper: the emulated-kernel tests under AddressSanitizer + UBSan (the CPU build is the only one that can have sanitizers on this pool).
#   make -C tests/hipemu -j6 asan && bash scripts/exp/emu_asan.sh [pytest args]
# A kernel that reads or writes past a tensor (the host tensors come from the intercepted allocator: red zones), a shift or a signed overflow UBSan
# objects to, a misaligned 16-byte access -- all fatal here.  Leak detection off (the interpreter); fibres run through ucontext in this build.
R=$PWD
export ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:detect_stack_use_after_return=0:abort_on_error=0:print_summary=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(g++ -print-file-name=libasan.so) $(g++ -print-file-name=libubsan.so)"
export GS_EMU_LIB=$R/tests/hipemu/libgsplat_emu_asan.so OMP_NUM_THREADS=${OMP_NUM_THREADS:-4}
if [ $# -gt 0 ]; then python -m pytest "$@" -x -q -p no:cacheprovider; else python -m pytest tests/test_emulated_kernels.py tests/test_randomized.py -x -q -p no:cacheprovider; fi
