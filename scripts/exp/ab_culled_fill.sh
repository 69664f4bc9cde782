# RECORD of how profiles/r06_ab_culled_fill.txt was measured: the code it drove (gs_set_culled_fill, the CULLED_FILL knob of scripts/stage_times.py, the lazy / trail /
# lead variants) lost and is in no commit -- the script does not run against this tree.
# round 6: A/B of the culled-row fill (gs_set_culled_fill): the zero rows of unrendered Gaussians stored by filler workgroups inside the backward blend
# (1, the default) against the per-Gaussian backward writing every row itself (0, rounds 1-5); one process per scene, alternating, hipEvent stage times
mkdir -p gpurun_out/abcf
(N=2000000 SH=3 STEPS=40 python scripts/stage_times.py CULLED_FILL=0,1,0,1,0,1
 CULL=0.5 N=2000000 SH=3 STEPS=40 python scripts/stage_times.py CULLED_FILL=0,1,0,1,0,1
 CULL=0.75 N=2000000 SH=3 STEPS=40 python scripts/stage_times.py CULLED_FILL=0,1,0,1
 CULL=0.5 N=2000000 STEPS=40 python scripts/stage_times.py CULLED_FILL=0,1,0,1
 N=500000 STEPS=40 python scripts/stage_times.py CULLED_FILL=0,1,0,1
 N=2000000 STEPS=40 python scripts/stage_times.py CULLED_FILL=0,1,0,1
 N=2000000 RGBD=1 STEPS=40 python scripts/stage_times.py CULLED_FILL=0,1,0,1
 N=200000 W=256 H=256 STEPS=60 python scripts/stage_times.py CULLED_FILL=0,1,0,1) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/abcf/stages.txt | cut -c1-400
for v in 0 1; do CULL=0.5 CULLED_FILL=$v N=2000000 SH=3 bash scripts/exp/prof_kernels.sh cf$v 2>&1 | tail -11; done | tee gpurun_out/abcf/kernels.txt
# the lazy variant (libgsplat_hip_lazy.so = scripts/exp/build_variant.sh lazy -DGS_PBWD_LAZY=1): mode 2 also requests nothing for unrendered rows
cp activesplat_amd/libgsplat_hip.so /tmp/main.so; cp activesplat_amd/libgsplat_hip_lazy.so activesplat_amd/libgsplat_hip.so
(CULL=0.5 N=2000000 SH=3 STEPS=40 python scripts/stage_times.py CULLED_FILL=1,2,1,2
 N=2000000 SH=3 STEPS=40 python scripts/stage_times.py CULLED_FILL=1,2,1,2) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/abcf/stages_lazy.txt | cut -c1-400
cp /tmp/main.so activesplat_amd/libgsplat_hip.so
# where the fillers sit in the index order (build_variant.sh trail -DGS_FILL_PLACEMENT=1 / lead -DGS_FILL_PLACEMENT=2; the shipped library interleaves them)
for v in trail lead; do cp activesplat_amd/libgsplat_hip_$v.so activesplat_amd/libgsplat_hip.so; echo "== $v"
  CULL=0.5 N=2000000 SH=3 STEPS=40 python scripts/stage_times.py CULLED_FILL=1,1 2>&1 | grep -v amdgpu.ids | cut -c1-400; done | tee gpurun_out/abcf/stages_placement.txt
cp /tmp/main.so activesplat_amd/libgsplat_hip.so
