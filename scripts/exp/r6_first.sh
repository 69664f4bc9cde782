# round 6, first device session: GPU suite, bench line, A/B of the binning chunk and of the pipelined keyframe batch
R=$PWD; mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r6a/pytest_gpu.log; tail -3 gpurun_out/r6a/pytest_gpu.log
timeout 500 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err; echo bench rc=$?; head -c 300 gpurun_out/r6a/bench.json; echo
for rep in 1 2; do for f in "--streams 1" "--streams 2 --c4-pipelined"; do
  echo "== c4 $f: $(timeout 200 python bench.py --workload c4 --no-extras --steps 4 --warmup 1 --c4-one-map $f 2>gpurun_out/r6a/c4.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['keyframes_per_s'], d['ms_per_optimiser_step'])")"
done; done 2>&1 | tee gpurun_out/r6a/ab_pipelined.txt
N=2000000 SH=3 LIBS="base c4096 c8192" bash scripts/exp/ab_libs.sh > gpurun_out/r6a/ab_chunk.txt 2>&1; grep -E "^==|tile_|preprocess_forward" gpurun_out/r6a/ab_chunk.txt | cut -c1-140
