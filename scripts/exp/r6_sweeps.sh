# round 6: the round-5 sweeps again on the changed arithmetic (factorised 2-D covariance, non-finite rule): raw-parameter entry, Adam inside the backward, the
# mapper's loss call, densify, and 1 000 ordinary scenes incl. the full-size kernels on small images
mkdir -p gpurun_out/r6e
SEED0=0 SEED1=1500 timeout 1500 python scripts/exp/fuzz_raw.py 2>&1 | tail -4 > gpurun_out/r6e/fuzz_raw.txt; tail -2 gpurun_out/r6e/fuzz_raw.txt | cut -c1-400
SEED0=0 SEED1=300 timeout 900 python scripts/exp/fuzz_adam.py 2>&1 | tail -3 > gpurun_out/r6e/fuzz_adam.txt; tail -1 gpurun_out/r6e/fuzz_adam.txt | cut -c1-300
SEED0=0 SEED1=600 timeout 1500 python scripts/exp/fuzz_get_loss.py 2>&1 | tail -12 > gpurun_out/r6e/fuzz_get_loss.txt; tail -3 gpurun_out/r6e/fuzz_get_loss.txt | cut -c1-400
SEED0=0 SEED1=200 timeout 900 python scripts/exp/fuzz_densify.py 2>&1 | tail -3 > gpurun_out/r6e/fuzz_densify.txt; tail -1 gpurun_out/r6e/fuzz_densify.txt | cut -c1-300
(RGBD=1 SEED0=130000 SEED1=130600 timeout 1500 python scripts/exp/fuzz_gpu.py; PLAIN=2 RGBD=1 SEED0=142000 SEED1=142400 timeout 1500 python scripts/exp/fuzz_gpu.py) 2>&1 | grep -v "^check_backward\|^decision-matched" > gpurun_out/r6e/fuzz.txt; grep "^seeds\|^fp32\|FAIL" gpurun_out/r6e/fuzz.txt | cut -c1-400
