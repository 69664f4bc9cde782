# End-of-round evidence run: parity tests, bench line, rocprof kernel summaries, PMC passes.  Run via gpurun from the repo root:
#   gpurun --timeout 2400 -- 'bash scripts/gpu_round_end.sh r06'
# Every step is bounded by its own timeout; outputs land in gpurun_out/final (copy what is to be judged into profiles/).
# The headline workload (bench.py, N = 1) is BASELINE configs[2]'s render: 2 M Gaussians, SH degree 3, 640x480, forward + backward.
R=$PWD; TAG=${1:-r06}
mkdir -p gpurun_out/final
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/final/build_smoke.log 2>&1; echo build+smoke rc=$?
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/final/${TAG}_pytest_gpu.log; tail -4 gpurun_out/final/${TAG}_pytest_gpu.log
cp gpurun_out/final/${TAG}_pytest_gpu.log profiles/ 2>/dev/null      # (bench.py carries the suite's tier tally: parity_vs_oracle.gpu_suite_tally)
timeout 400 python bench.py > gpurun_out/final/${TAG}_bench.json 2> gpurun_out/final/bench.err; echo bench rc=$?; head -c 400 gpurun_out/final/${TAG}_bench.json; echo
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > gpurun_out/final/${TAG}_bench_driver_flags.json 2>/dev/null; head -c 300 gpurun_out/final/${TAG}_bench_driver_flags.json; echo
export TMPDIR=/tmp; cd /tmp
# kernel table of the same command (2 M / SH-3 frames)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o ${TAG}_2m -- python $R/bench.py --steps 50 --warmup 10 --no-extras > $R/gpurun_out/final/${TAG}_2m_prof.log 2>&1; echo prof rc=$?
# PMC passes of the same workload, one counter group per pass, --kernel-trace only (no other trace domain)
pmc() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/final -o ${TAG}_2m_pmc_$name -- python $R/bench.py --steps 6 --warmup 2 --prep-seconds 0 --no-extras > $R/gpurun_out/final/${TAG}_2m_pmc_$name.log 2>&1; echo pmc $name rc=$?; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pmc sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
# the legs that rounds 1-5 supported with hipEvent stage times only: kernel table + counters (FETCH / WRITE in their own passes, one SQ group)
# for configs[1]'s frames, 20 iterations of configs[2]'s loop incl. one densify event, and configs[3]'s step on one GPU incl. the 1-rank RCCL exchange (G = 14 and 59)
leg() { name=$1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o ${TAG}_${name} -- python $R/scripts/prof_legs.py $name > $R/gpurun_out/final/${TAG}_${name}_prof.log 2>&1; echo prof $name rc=$?
  for grp in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
    set -- $grp; g=$1; shift
    STEPS=4 ITERS=12 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/final -o ${TAG}_${name}_pmc_$g -- python $R/scripts/prof_legs.py $name > $R/gpurun_out/final/${TAG}_${name}_pmc_$g.log 2>&1; echo pmc $name $g rc=$?
  done; }
leg c1; leg c2loop; leg c3step
cd $R
python scripts/pmc_summarise.py gpurun_out/final ${TAG}_2m > gpurun_out/final/${TAG}_2m_pmc_summary.txt 2>&1
for name in c1 c2loop c3step; do python scripts/pmc_summarise.py gpurun_out/final ${TAG}_${name} > gpurun_out/final/${TAG}_${name}_pmc_summary.txt 2>&1; done
cp profiles/${TAG}_*_pmc_summary.json gpurun_out/final/ 2>/dev/null
# bench.py's legs read the kernel tables next to the counter summaries (bench.profiled_kernels): both under profiles/ for the second bench run below
cp gpurun_out/final/${TAG}_2m_kernel_stats.csv gpurun_out/final/${TAG}_c1_kernel_stats.csv gpurun_out/final/${TAG}_c2loop_kernel_stats.csv gpurun_out/final/${TAG}_c3step_kernel_stats.csv profiles/ 2>/dev/null
# the bench line once more, now that this round's PMC summary exists (roofline.traffic reads it)
timeout 400 python bench.py > gpurun_out/final/${TAG}_bench.json 2> gpurun_out/final/bench.err; echo bench rc=$?
# BASELINE configs[2]'s loop and the configs[4] substitute (256 x 256 and 512 x 512, PSNR + SSIM, HIP loop vs oracle-backed loop); the planner's top-down camera
timeout 600 python scripts/configs_report.py > gpurun_out/final/${TAG}_configs.json 2> gpurun_out/final/configs.err; echo configs rc=$?
timeout 200 python scripts/topdown_time.py > gpurun_out/final/${TAG}_topdown.json 2> gpurun_out/final/topdown.err; echo topdown rc=$?
# `bench.py --gpus 2` with no launcher environment (both ranks on this one device: gloo): the self-launch path's line
BENCH_SAME_DEVICE=1 timeout 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-extras > gpurun_out/final/${TAG}_bench_gpus2_same_device.json 2> gpurun_out/final/bench2.err; echo bench2 rc=$?
# the same with EIGHT ranks on the one device (the north star's rank count: partition, exchange at row blocks of N / 8, densify cadence, both maps; gloo through host memory)
BENCH_SAME_DEVICE=1 timeout 600 python bench.py --gpus 8 --steps 3 --warmup 1 --no-extras > gpurun_out/final/${TAG}_bench_gpus8_same_device.json 2> gpurun_out/final/bench8.err; echo bench8 rc=$?
# the stand-alone multi-tensor Adam (keyframe batches, event iterations) on configs[2]'s 118 M elements; the mapping iteration at the reference's frames
ELEMS=118000000 timeout 100 python scripts/adam_bench.py > gpurun_out/final/${TAG}_adam.txt 2>&1; tail -1 gpurun_out/final/${TAG}_adam.txt
for mode in "ADAM=0 DIRECT=0" "ADAM=1 DIRECT=0" "ADAM=1 DIRECT=1"; do env $mode PROFILE=0 BATCHES=5 N=200000 W=256 H=256 timeout 120 taskset -c 4 python scripts/exp/map_iter.py 2>&1 | tail -1 | sed "s/^/$mode /"; done > gpurun_out/final/${TAG}_map_iter_256.txt; cat gpurun_out/final/${TAG}_map_iter_256.txt
timeout 200 taskset -c 4 python scripts/exp/harness_time.py 2>&1 | grep "ms total" > gpurun_out/final/${TAG}_harness.txt; cat gpurun_out/final/${TAG}_harness.txt
# where the backward blend's instructions go (control flow replayed from the frame's integer artefacts, priced from the ISA)
(timeout 100 python scripts/exp/bwd_work.py 2>&1 | tail -12; N=500000 timeout 100 python scripts/exp/bwd_work.py 2>&1 | tail -12) > gpurun_out/final/${TAG}_bwd_work.txt
# planner panorama, densify event
timeout 200 python scripts/lookaround_times.py > gpurun_out/final/${TAG}_lookaround.txt 2>&1
N=3000000 timeout 200 python scripts/densify_time.py > gpurun_out/final/${TAG}_densify.txt 2>&1
ls gpurun_out/final | head -80
