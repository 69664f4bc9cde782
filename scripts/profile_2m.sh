# per-stage (hipEvent) and per-kernel (rocprofv3) times of the north-star configuration: 2M Gaussians, 640x480, SH-0 / SH-3
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof2m
N=2000000 python scripts/stage_times.py 2>&1 | tail -1
SH=3 N=2000000 python scripts/stage_times.py 2>&1 | tail -1
cd /tmp
N=2000000 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2m -o n2m -- python $R/scripts/stage_times.py > $R/gpurun_out/prof2m/n2m.log 2>&1
cd $R; python - <<'PY'
import pandas as pd
d=pd.read_csv('gpurun_out/prof2m/n2m_kernel_stats.csv')
d['Name']=d['Name'].str.slice(0,90)
print(d[['Name','Calls','AverageNs']].head(16).to_string())
PY
