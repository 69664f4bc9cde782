# Development helper: A/B of activesplat_amd/libgsplat_hip_base.so against the current library in ONE box session, twice: CMD is run with each.
#   CMD='N=2000000 SH=3 python scripts/stage_times.py' bash scripts/exp/ab_generic.sh
R=$PWD
cp $R/activesplat_amd/libgsplat_hip.so /tmp/new.so; cp $R/activesplat_amd/libgsplat_hip_base.so /tmp/base.so
for rep in 1 2; do for v in base new; do
  cp /tmp/$v.so $R/activesplat_amd/libgsplat_hip.so
  echo "== $v: $(bash -c "$CMD" 2>/dev/null | tail -1 | cut -c1-400)"
done; done
cp /tmp/new.so $R/activesplat_amd/libgsplat_hip.so
