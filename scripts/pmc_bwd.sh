# PMC passes over the C2 workload for one backward variant (GS_BWD_VARIANT): counters only + kernel trace
R=$PWD; TAG=${1:-v}; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmcb; cd /tmp
run() { name=$1; shift; STEPS=6 WARMUP=2 timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcb -o ${TAG}_pmc_$name -- python $R/scripts/stage_times.py > $R/gpurun_out/pmcb/${TAG}_$name.log 2>&1; echo $name rc=$?; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $R
python - <<PY
import pandas as pd, glob, re
out={}
for f in sorted(glob.glob("gpurun_out/pmcb/${TAG}_pmc_*_counter_collection.csv")):
    d=pd.read_csv(f); d=d[d["Kernel_Name"].str.contains("blend_backward")]
    for c,g in d.groupby("Counter_Name"):
        v=g.sort_values("Dispatch_Id")["Counter_Value"].tolist(); v=v[len(v)//4:] or v
        out[c]=sum(v)/len(v)
print("${TAG}", {k: round(v) for k,v in sorted(out.items())})
PY
