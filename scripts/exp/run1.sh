cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 300 python scripts/stage_times.py 2>&1 | tail -1
N=2000000 SH=3 STEPS=20 timeout 600 python scripts/stage_times.py 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
N=2000000 SH=3 STEPS=10 BWD=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/mrg -o m -- python /root/repo/scripts/stage_times.py > /dev/null 2>&1
STEPS=10 BWD=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/mrg -o c2 -- python /root/repo/scripts/stage_times.py > /dev/null 2>&1
grep -h "tile_bin\|tile_scatter\|tile_sort\|tile_merge" /root/repo/gpurun_out/mrg/m_kernel_stats.csv /root/repo/gpurun_out/mrg/c2_kernel_stats.csv | sed 's/(.*)"/"/' | cut -d, -f1,2,4
