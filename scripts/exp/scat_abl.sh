export TMPDIR=/tmp; R=$PWD; mkdir -p $R/gpurun_out/scat; cd /tmp
for n in 2000000 500000; do
  BWD=0 N=$n STEPS=10 WARMUP=3 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/scat -o s$n -- python $R/scripts/stage_times.py > /dev/null 2>&1
  python - <<PY
import pandas as pd
d=pd.read_csv("$R/gpurun_out/scat/s${n}_kernel_stats.csv")
d=d[d["Name"].str.contains("tile_|preprocess")]
print("N=$n", {n.split("(")[0].replace("void gs::","")[:34]: round(v/1e3,1) for n,v in zip(d["Name"], d["AverageNs"])})
PY
done
