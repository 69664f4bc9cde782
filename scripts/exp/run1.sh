cd /root/repo
N=200000 PROFILE=1 timeout 300 python scripts/exp/iter_host.py 2>&1 | grep -v amdgpu | head -60
N=2000 timeout 300 python scripts/exp/iter_host.py 2>&1 | grep -v amdgpu | head -3
cd /tmp && export TMPDIR=/tmp
N=200000 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/ith -o it -- python /root/repo/scripts/exp/iter_host.py > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/root/repo/gpurun_out/ith/it_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per iteration us', tot/220/1e3, 'launches per iteration', sum(int(r['Calls']) for r in rows)/220)
PY
