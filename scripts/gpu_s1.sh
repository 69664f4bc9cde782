# session 1 of round 4: the new tests, the whole bench line, the top-down camera timing
mkdir -p gpurun_out/s1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/s1/pytest_gpu.log; tail -4 gpurun_out/s1/pytest_gpu.log
timeout 200 python scripts/topdown_time.py > gpurun_out/s1/topdown.json 2> gpurun_out/s1/topdown.err; echo topdown rc=$?; head -c 1500 gpurun_out/s1/topdown.json; echo
timeout 500 python bench.py > gpurun_out/s1/bench.json 2> gpurun_out/s1/bench.err; echo bench rc=$?; head -c 600 gpurun_out/s1/bench.json; echo
tail -3 gpurun_out/s1/bench.err
