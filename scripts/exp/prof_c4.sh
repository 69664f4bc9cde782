# Development helper: GPU busy time of configs[3]'s keyframe batch on one GPU (sum of kernel durations from rocprofv3) next to its wall clock.
# usage (GPU box, repo root): bash scripts/exp/prof_c4.sh
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof_c4; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -o c4 -- python $R/bench.py --workload c4 --gpus 1 --steps 3 --warmup 1 --streams ${STREAMS:-1} --c4-one-map 2>/dev/null | tail -1 | cut -c200-330
cd $R; python - <<PY
import pandas as pd, glob
f = glob.glob('gpurun_out/prof_c4/**/*kernel_stats.csv', recursive=True)[0]
d = pd.read_csv(f)
kf = 64 * 4                                   # one untimed step + three timed
d['Name'] = d['Name'].str.replace(r'\(.*', '', regex=True).str.slice(0, 60)
d['us_per_keyframe'] = d['TotalDurationNs'] / kf / 1e3
print(d[['Name', 'Calls', 'AverageNs', 'us_per_keyframe']].head(22).to_string())
print('GPU busy per keyframe: %.1f us' % d['us_per_keyframe'].sum())
PY
