R=$PWD
cp $R/activesplat_amd/libgsplat_hip.so /tmp/new.so; cp $R/activesplat_amd/libgsplat_hip_base.so /tmp/base.so
for rep in 1 2; do for v in base new; do
  cp /tmp/$v.so $R/activesplat_amd/libgsplat_hip.so
  echo "== $v: $(timeout 200 python bench.py --workload c4 --no-extras --steps 4 --warmup 1 $C4FLAGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['keyframes_per_s'], d['ms_per_optimiser_step'])")"
done; done
cp /tmp/new.so $R/activesplat_amd/libgsplat_hip.so
