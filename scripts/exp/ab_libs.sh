# Development helper: A/B of several builds of the library in ONE box session (boxes differ by +-3 %): rocprofv3 kernel tables of
# scripts/stage_times.py with each activesplat_amd/libgsplat_hip_<name>.so in turn, twice.
# usage (GPU box, repo root): N=2000000 SH=3 LIBS="base new" bash scripts/exp/ab_libs.sh     (new = the current library)
R=$PWD
cp $R/activesplat_amd/libgsplat_hip.so /tmp/new.so
for v in ${LIBS:-base new}; do [ $v = new ] || cp $R/activesplat_amd/libgsplat_hip_$v.so /tmp/$v.so; done
for rep in 1 2; do
for v in ${LIBS:-base new}; do
  cp /tmp/$v.so $R/activesplat_amd/libgsplat_hip.so
  echo "== $v"; bash $R/scripts/exp/prof_kernels.sh ab_$v 2>&1 | tail -12
done; done
cp /tmp/new.so $R/activesplat_amd/libgsplat_hip.so
