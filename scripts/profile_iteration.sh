R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o iter -- python $R/scripts/iteration_times.py > $R/gpurun_out/prof/iter.log 2>&1
cd $R; python - <<'PY'
import pandas as pd
d=pd.read_csv('gpurun_out/prof/iter_kernel_stats.csv')
d['Name']=d['Name'].str.slice(0,90)
d['per_iter_us']=d['TotalDurationNs']/70/1000      # 2 x (5 warm + 30 timed) iterations
print(d[['Name','Calls','AverageNs','per_iter_us']].head(40).to_string())
print('total per iteration (avg of both modes) us', d['per_iter_us'].sum())
PY
