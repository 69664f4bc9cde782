# Development helper: GPU busy time per mapping iteration (sum of kernel durations from rocprofv3) next to the wall clock.
# usage (GPU box, repo root): N=200000 W=256 H=256 bash scripts/exp/prof_map_iter.sh
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof_mapiter; cd /tmp
K=100 BATCHES=2 PROFILE=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_mapiter -o mi -- python $R/scripts/exp/map_iter.py 2>&1 | tail -1
cd $R; python - <<PY
import pandas as pd, glob
f = glob.glob('gpurun_out/prof_mapiter/**/*kernel_stats.csv', recursive=True)[0]
d = pd.read_csv(f)
iters = 30 + 2 * 100
d['Name'] = d['Name'].str.replace(r'\(.*', '', regex=True).str.slice(0, 60)
d['us_per_iter'] = d['TotalDurationNs'] / iters / 1e3
print(d[['Name', 'Calls', 'AverageNs', 'us_per_iter']].head(24).to_string())
print('GPU busy per iteration: %.1f us over %d kernels per iteration' % (d['us_per_iter'].sum(), d['Calls'].sum() / iters))
PY
