# Development helper: kernel table (+ optionally SQ counters: PMC=1) of the two loss kernels alone.   bash scripts/exp/loss_prof.sh [tag]
R=$PWD; TAG=${1:-loss}; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/$TAG; cd /tmp
for wh in "640 480" "256 256"; do set -- $wh; W=$1 H=$2 python $R/scripts/exp/loss_only.py 2>&1 | grep "fused mapping"; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG -o k -- python $R/scripts/exp/loss_only.py > /dev/null 2>&1
if [ -n "$PMC" ]; then
CALLS=20 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$TAG -o p1 -- python $R/scripts/exp/loss_only.py > /dev/null 2>&1
CALLS=20 timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/$TAG -o p2 -- python $R/scripts/exp/loss_only.py > /dev/null 2>&1
fi
cd $R; python - <<PY
import pandas as pd, glob
d = pd.read_csv(glob.glob('gpurun_out/$TAG/**/k_kernel_stats.csv', recursive=True)[0])
d['Name'] = d.Name.str.slice(4, 22)
print(d[d.Name.str.contains('loss_')][['Name','Calls','AverageNs','MinNs','MaxNs']].to_string())
for f in sorted(glob.glob('gpurun_out/$TAG/**/p?_counter_collection.csv', recursive=True)):
    c = pd.read_csv(f); c = c[c.Kernel_Name.str.contains('loss_')]; c['k'] = c.Kernel_Name.str.slice(4, 20)
    print(c.groupby(['k','Counter_Name']).Counter_Value.mean().round().to_string())
PY
