mkdir -p gpurun_out/s7
for rep in 1 2; do
for mode in "ADAM=0 DIRECT=0" "ADAM=1 DIRECT=0" "ADAM=1 DIRECT=1"; do
  env $mode PROFILE=0 BATCHES=5 N=200000 W=256 H=256 timeout 120 taskset -c 4 python scripts/exp/map_iter.py 2>&1 | tail -1 | sed "s/^/$mode /"
done; done | tee gpurun_out/s7/map_iter_direct.txt
for mode in "ADAM=1 DIRECT=0" "ADAM=1 DIRECT=1"; do
  env $mode PROFILE=0 BATCHES=4 N=1000000 W=512 H=512 timeout 120 taskset -c 4 python scripts/exp/map_iter.py 2>&1 | tail -1 | sed "s/^/$mode /"
done | tee -a gpurun_out/s7/map_iter_direct.txt
ADAM=1 DIRECT=1 PROFILE=1 BATCHES=2 N=200000 W=256 H=256 timeout 120 taskset -c 4 python scripts/exp/map_iter.py 2>&1 | head -30 > gpurun_out/s7/direct_prof.txt; head -28 gpurun_out/s7/direct_prof.txt
timeout 300 python -m pytest tests -m gpu -q -k "without_autograd or adam_inside or harness" 2>&1 | tail -4
