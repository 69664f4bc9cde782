# session 3 of round 4: the failing tests with their full output; A/B of the producer / consumer backward blend
mkdir -p gpurun_out/s3
timeout 900 python -m pytest tests -m gpu -q -k "mapper_harness_gpu or adam_inside or test_bench or producer_consumer or refuses_in_kernel" 2>&1 | tail -150 > gpurun_out/s3/pytest_sel.log; tail -12 gpurun_out/s3/pytest_sel.log
for rep in 1 2; do
  N=2000000 SH=3 STEPS=30 WARMUP=5 timeout 200 python scripts/stage_times.py BWD_PC=0,1 2>&1 | grep -v amdgpu.ids
  N=500000 STEPS=30 WARMUP=5 timeout 200 python scripts/stage_times.py BWD_PC=0,1 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/s3/ab_pc.txt
BWD_PC=1 N=2000000 SH=3 STEPS=30 WARMUP=5 timeout 200 python scripts/stage_times.py CHAIN=1,2,3 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s3/ab_pc.txt
BWD_PC=1 RGBD=1 N=2000000 STEPS=30 WARMUP=5 timeout 200 python scripts/stage_times.py BWD_PC=0,1 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/s3/ab_pc.txt
