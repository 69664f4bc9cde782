# session 4 of round 4: whole -m gpu suite (full log), P/C backward ablations, kernel tables of the mapping iteration with / without Adam in the backward
mkdir -p gpurun_out/s4
timeout 900 python -m pytest tests -m gpu -q 2>&1 > gpurun_out/s4/pytest_gpu_full.log; tail -8 gpurun_out/s4/pytest_gpu_full.log
cp activesplat_amd/libgsplat_hip.so /tmp/new.so
for v in pcabl1 pcabl2; do
  cp activesplat_amd/libgsplat_hip_$v.so activesplat_amd/libgsplat_hip.so
  echo "== $v"; BWD_PC=1 N=2000000 SH=3 STEPS=30 WARMUP=5 timeout 200 python scripts/stage_times.py BWD_PC=1 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/s4/pc_ablations.txt
cp /tmp/new.so activesplat_amd/libgsplat_hip.so
for a in 0 1; do echo "== ADAM=$a 640x480 500k"; ADAM=$a N=500000 W=640 H=480 bash scripts/exp/prof_map_iter.sh 2>&1 | tail -26; rm -rf gpurun_out/prof_mapiter; done > gpurun_out/s4/map_iter_kernels.txt 2>&1
grep -E "==|GPU busy|preprocess_backward|adam|us per mapping" gpurun_out/s4/map_iter_kernels.txt
for a in 0 1; do echo "== ADAM=$a 256x256 200k"; ADAM=$a N=200000 W=256 H=256 bash scripts/exp/prof_map_iter.sh 2>&1 | tail -26; rm -rf gpurun_out/prof_mapiter; done >> gpurun_out/s4/map_iter_kernels.txt 2>&1
grep -E "==|GPU busy|us per mapping" gpurun_out/s4/map_iter_kernels.txt | tail -6
