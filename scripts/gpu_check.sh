# GPU round script: build, parity tests, bench, rocprof summary (run through gpurun from the repo root)
R=$PWD
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/build.log 2>&1; echo build rc=$?
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.log
[ -n "$EXTRA" ] && bash -c "$EXTRA" 2>&1 | tee gpurun_out/extra.log
timeout 600 python bench.py --steps 50 --warmup 10 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo bench rc=$?; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
if [ -n "$PROF" ]; then
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o r1 -- python $R/bench.py --steps 20 --warmup 5 --no-extras > $R/gpurun_out/prof.log 2>&1; echo prof rc=$?
cd $R; ls -R gpurun_out/prof | head -20
fi
