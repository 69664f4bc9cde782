# session 2 of round 4: the whole -m gpu suite, the top-down camera timing, Adam inside the backward (configs[2]'s loop, 256x256 mapping iteration)
mkdir -p gpurun_out/s2
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/s2/pytest_gpu.log; tail -6 gpurun_out/s2/pytest_gpu.log
timeout 200 python scripts/topdown_time.py > gpurun_out/s2/topdown.json 2> gpurun_out/s2/topdown.err; echo topdown rc=$?; head -c 2500 gpurun_out/s2/topdown.json; echo
timeout 300 python scripts/exp/c2_loop.py > gpurun_out/s2/c2_loop.txt 2>&1; cat gpurun_out/s2/c2_loop.txt | tail -6
for a in 0 1 0 1; do ADAM=$a PROFILE=0 BATCHES=4 N=200000 W=256 H=256 timeout 120 python scripts/exp/map_iter.py 2>&1 | tail -1 | sed "s/^/ADAM=$a /"; done | tee gpurun_out/s2/map_iter.txt
for a in 0 1; do ADAM=$a PROFILE=0 BATCHES=3 K=100 N=500000 W=640 H=480 timeout 120 python scripts/exp/map_iter.py 2>&1 | tail -1 | sed "s/^/ADAM=$a /"; done | tee -a gpurun_out/s2/map_iter.txt
