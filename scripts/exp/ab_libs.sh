# Development helper: A/B of two builds of the library in ONE box session (boxes differ by +-3 %): rocprofv3 kernel tables of
# scripts/stage_times.py with activesplat_amd/libgsplat_hip_base.so (built from another revision) and the current library, alternating.
# usage (GPU box, repo root): N=2000000 SH=3 bash scripts/exp/ab_libs.sh
R=$PWD
cp $R/activesplat_amd/libgsplat_hip.so /tmp/new.so; cp $R/activesplat_amd/libgsplat_hip_base.so /tmp/base.so
for v in base new base new; do
  cp /tmp/$v.so $R/activesplat_amd/libgsplat_hip.so
  echo "== $v"; bash $R/scripts/exp/prof_kernels.sh ab_$v 2>&1 | tail -12
done
cp /tmp/new.so $R/activesplat_amd/libgsplat_hip.so
