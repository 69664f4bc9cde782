# Development helper: rocprofv3 kernel table of scripts/stage_times.py under the current environment (N, SH, GS_* knobs).
# usage (GPU box, repo root): N=2000000 SH=3 bash scripts/exp/prof_kernels.sh tag
R=$PWD; TAG=${1:-k}; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof_$TAG; cd /tmp
STEPS=${STEPS-12} WARMUP=${WARMUP-3} timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- python $R/scripts/stage_times.py > $R/gpurun_out/prof_$TAG/run.log 2>&1
cd $R; python - <<PY
import pandas as pd, glob
f = glob.glob('gpurun_out/prof_$TAG/**/*kernel_stats.csv', recursive=True)[0]
d = pd.read_csv(f)
d['Name'] = d['Name'].str.replace(r'\(.*', '', regex=True).str.slice(0, 64)
print(d[['Name', 'Calls', 'AverageNs', 'Percentage']].head(14).to_string())
PY
