R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/profla; cd /tmp
N=1000000 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/profla -o la -- python $R/scripts/lookaround_times.py > $R/gpurun_out/profla/la.log 2>&1
cd $R; python - <<'PY'
import pandas as pd
d=pd.read_csv('gpurun_out/profla/la_kernel_trace.csv').sort_values('Start_Timestamp')
b=d[d['Kernel_Name'].str.contains('blend_forward')].copy()
b['start_us']=(b['Start_Timestamp']-b['Start_Timestamp'].iloc[0])/1000; b['dur_us']=(b['End_Timestamp']-b['Start_Timestamp'])/1000
print(b[['start_us','dur_us','Queue_Id']].iloc[60:78].to_string())
PY
