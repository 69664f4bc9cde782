# round 6: a longer soak of the random sweeps on the final arithmetic (fresh seed ranges)
mkdir -p gpurun_out/r6h
(RGBD=1 SEED0=200000 SEED1=202500 timeout 2400 python scripts/exp/fuzz_gpu.py; PLAIN=2 RGBD=1 SEED0=210000 SEED1=210800 timeout 2400 python scripts/exp/fuzz_gpu.py; TWIN=1 RGBD=1 SEED0=220000 SEED1=220400 timeout 1200 python scripts/exp/fuzz_gpu.py) 2>&1 | grep -v "^check_backward\|^decision-matched" > gpurun_out/r6h/soak.txt; grep "^seeds\|^fp32\|FAIL" gpurun_out/r6h/soak.txt | cut -c1-300
SEED0=1500 SEED1=4000 timeout 2400 python scripts/exp/fuzz_raw.py 2>&1 | tail -4 > gpurun_out/r6h/soak_raw.txt; tail -3 gpurun_out/r6h/soak_raw.txt | cut -c1-300
