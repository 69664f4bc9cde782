mkdir -p gpurun_out/s8
timeout 300 taskset -c 4 python scripts/exp/harness_time.py > gpurun_out/s8/harness.txt 2>&1; head -60 gpurun_out/s8/harness.txt
