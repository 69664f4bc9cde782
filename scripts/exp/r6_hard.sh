# round 6: the factorised 2-D covariance / determinant on the device -- GPU suite, the s = 1.2 sweep of profiles/r05_fuzz_hard.txt again, stage times
mkdir -p gpurun_out/r6d
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r6d/pytest_gpu.log; tail -3 gpurun_out/r6d/pytest_gpu.log | cut -c1-400
(HARD=1.2 RGBD=1 SEED0=180000 SEED1=180500 timeout 1500 python scripts/exp/fuzz_gpu.py; HARD=1.2 PLAIN=2 RGBD=1 SEED0=181000 SEED1=181400 timeout 1500 python scripts/exp/fuzz_gpu.py) > gpurun_out/r6d/fuzz_hard.txt 2>&1; tail -12 gpurun_out/r6d/fuzz_hard.txt | cut -c1-600
(RGBD=1 SEED0=20000 SEED1=20400 timeout 900 python scripts/exp/fuzz_gpu.py) > gpurun_out/r6d/fuzz_plain.txt 2>&1; tail -3 gpurun_out/r6d/fuzz_plain.txt | cut -c1-600
N=2000000 SH=3 timeout 200 python scripts/stage_times.py 2>&1 | tail -1 | tee gpurun_out/r6d/stages_2m.txt
N=500000 timeout 200 python scripts/stage_times.py 2>&1 | tail -1 | tee gpurun_out/r6d/stages_c1.txt
