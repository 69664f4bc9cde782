# kernel table of a 256x256 frame over 1M Gaussians (ActiveSplat's mapping resolution: 256 tiles, ~10 k records per tile)
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof256; cd /tmp
W=256 H=256 N=1000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof256 -o p256 -- python $R/scripts/stage_times.py > $R/gpurun_out/prof256/p256.log 2>&1
cd $R; tail -1 gpurun_out/prof256/p256.log | cut -c1-330; python - <<'PY'
import pandas as pd
d=pd.read_csv('gpurun_out/prof256/p256_kernel_stats.csv')
d['Name']=d['Name'].str.slice(0,80)
print(d[['Name','Calls','AverageNs']].head(12).to_string())
PY
