# round 6, third soak: fresh seed ranges for the sweeps r6_soak2.sh did not hold -- Adam inside the backward, the mapper's loss call, densify, the
# strongly anisotropic scenes (HARD = 1.2, incl. the full-size kernels on small images) -- and the raw-parameter entry again under its amended tie verdict
mkdir -p gpurun_out/r6t
SEED0=300 SEED1=1500 timeout 1500 python scripts/exp/fuzz_adam.py 2>&1 | tail -3 > gpurun_out/r6t/fuzz_adam.txt; tail -1 gpurun_out/r6t/fuzz_adam.txt | cut -c1-300
SEED0=600 SEED1=2400 timeout 2000 python scripts/exp/fuzz_get_loss.py 2>&1 | tail -12 > gpurun_out/r6t/fuzz_get_loss.txt; tail -3 gpurun_out/r6t/fuzz_get_loss.txt | cut -c1-400
SEED0=200 SEED1=1000 timeout 1500 python scripts/exp/fuzz_densify.py 2>&1 | tail -3 > gpurun_out/r6t/fuzz_densify.txt; tail -1 gpurun_out/r6t/fuzz_densify.txt | cut -c1-300
(HARD=1.2 RGBD=1 SEED0=340000 SEED1=341200 timeout 1500 python scripts/exp/fuzz_gpu.py; HARD=1.2 PLAIN=2 RGBD=1 SEED0=350000 SEED1=350400 timeout 1500 python scripts/exp/fuzz_gpu.py) 2>&1 | grep -v "^check_backward\|^decision-matched" > gpurun_out/r6t/fuzz_hard.txt; grep "^seeds\|^fp32\|FAIL" gpurun_out/r6t/fuzz_hard.txt | cut -c1-400
SEED0=6500 SEED1=9000 timeout 1500 python scripts/exp/fuzz_raw.py 2>&1 | tail -4 > gpurun_out/r6t/fuzz_raw.txt; tail -3 gpurun_out/r6t/fuzz_raw.txt | cut -c1-300
