# round 6, second soak on the final tree: fresh seed ranges for every sweep (the first soak: r6_soak.sh -> profiles/r06_soak.txt)
mkdir -p gpurun_out/r6s
(RGBD=1 SEED0=300000 SEED1=302500 timeout 2400 python scripts/exp/fuzz_gpu.py; PLAIN=2 RGBD=1 SEED0=310000 SEED1=310800 timeout 2400 python scripts/exp/fuzz_gpu.py; TWIN=1 RGBD=1 SEED0=320000 SEED1=320400 timeout 1200 python scripts/exp/fuzz_gpu.py) 2>&1 | grep -v "^check_backward\|^decision-matched" > gpurun_out/r6s/soak2.txt; grep "^seeds\|^fp32\|FAIL" gpurun_out/r6s/soak2.txt | cut -c1-300
SEED0=4000 SEED1=6500 timeout 2400 python scripts/exp/fuzz_raw.py 2>&1 | tail -4 > gpurun_out/r6s/soak2_raw.txt; tail -3 gpurun_out/r6s/soak2_raw.txt | cut -c1-300
