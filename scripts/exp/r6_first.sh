# round 6, first device session: GPU suite, bench line, A/B of the binning chunk and of the pipelined keyframe batch
R=$PWD; mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r6a/pytest_gpu.log; tail -3 gpurun_out/r6a/pytest_gpu.log
timeout 500 python bench.py > gpurun_out/r6a/bench.json 2> gpurun_out/r6a/bench.err; echo bench rc=$?; head -c 300 gpurun_out/r6a/bench.json; echo
# (the A/B of the pipelined keyframe batch ran here and in a second session; the walk was removed after it lost: profiles/r06_ab_pipelined.txt)
# needs libgsplat_hip_{base,c4096,c8192}.so from scripts/exp/build_variant.sh
N=2000000 SH=3 LIBS="base c4096 c8192" bash scripts/exp/ab_libs.sh > gpurun_out/r6a/ab_chunk.txt 2>&1; grep -E "^==|tile_|preprocess_forward" gpurun_out/r6a/ab_chunk.txt | cut -c1-140
