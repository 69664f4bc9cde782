# rocprofv3 kernel-trace summary of the bench workload -> gpurun_out/prof (copy the *_kernel_stats.csv into profiles/)
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/prof; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o ${1:-r01} -- python $R/bench.py --steps 50 --warmup 10 --no-extras > $R/gpurun_out/prof/${1:-r01}.log 2>&1
echo prof rc=$?; cd $R; head -30 gpurun_out/prof/${1:-r01}_kernel_stats.csv | cut -c1-200
