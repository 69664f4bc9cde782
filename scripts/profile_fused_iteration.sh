# rocprof kernel table of the fully fused mapping iteration (get_loss + backward + Adam), per iteration
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof; cd /tmp
ONLY_FUSED=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o fiter -- python $R/scripts/iteration_times.py > $R/gpurun_out/prof/fiter.log 2>&1
cd $R; tail -1 gpurun_out/prof/fiter.log; python - <<'PY'
import pandas as pd
d=pd.read_csv('gpurun_out/prof/fiter_kernel_stats.csv')
d['Name']=d['Name'].str.slice(0,100)
d['per_iter_us']=d['TotalDurationNs']/35/1000      # 5 warm + 30 timed iterations
d['calls_per_iter']=d['Calls']/35
print(d[['Name','calls_per_iter','AverageNs','per_iter_us']].head(45).to_string())
print('total kernel time per iteration us', d['per_iter_us'].sum(), ' launches per iteration', d['calls_per_iter'].sum())
PY
