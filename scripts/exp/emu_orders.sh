# Development helper: the emulated-kernel tests with the emulator running a workgroup's READY fibres in REVERSE and in RANDOM order between rendezvous
# (tests/hipemu/hipemu.cpp, HIPEMU_ORDER): a result that depends on that order is a missing barrier.  bash scripts/exp/emu_orders.sh [pytest args]
for o in reverse random:1 random:7; do
  echo "== HIPEMU_ORDER=$o"
  if [ $# -gt 0 ]; then HIPEMU_ORDER=$o python -m pytest "$@" -x -q -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300
  else HIPEMU_ORDER=$o python -m pytest tests/test_emulated_kernels.py tests/test_randomized.py tests/test_golden.py tests/test_mapper.py -m "not gpu" -x -q -p no:cacheprovider 2>&1 | tail -3 | cut -c1-300; fi
done
