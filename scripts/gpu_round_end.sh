# End-of-round evidence run: parity tests, bench line, rocprof kernel summaries, PMC passes.  Run via gpurun from the repo root:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round_end.sh r02'
# Every step is bounded by its own timeout; outputs land in gpurun_out/final (copy what is to be judged into profiles/).
R=$PWD; TAG=${1:-r02}
mkdir -p gpurun_out/final
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/final/build_smoke.log 2>&1; echo build+smoke rc=$?
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee gpurun_out/final/pytest_gpu.log
timeout 300 python bench.py > gpurun_out/final/${TAG}_bench.json 2> gpurun_out/final/bench.err; echo bench rc=$?; tail -c 400 gpurun_out/final/${TAG}_bench.json
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o ${TAG}_bench -- python $R/bench.py --steps 50 --warmup 10 --no-extras > $R/gpurun_out/final/${TAG}_bench_prof.log 2>&1; echo prof rc=$?
pmc() { name=$1; shift; timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/final -o ${TAG}_pmc_$name -- python $R/bench.py --steps 5 --warmup 2 --no-extras > $R/gpurun_out/final/${TAG}_pmc_$name.log 2>&1; echo pmc $name rc=$?; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pmc sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
cd $R
python scripts/pmc_summarise.py gpurun_out/final ${TAG} > gpurun_out/final/${TAG}_pmc_summary.txt 2>&1
# the north-star configuration (2 M Gaussians, SH-3): kernel table + PMC past the Infinity Cache
bash scripts/pmc_2m.sh ${TAG} > gpurun_out/final/${TAG}_pmc_2m.txt 2>&1
cp gpurun_out/pmc2m/${TAG}_2m_kernel_stats.csv gpurun_out/final/ 2>/dev/null
cp profiles/${TAG}_pmc_summary.json profiles/${TAG}_2m_pmc_summary.json gpurun_out/final/ 2>/dev/null
# configs[2] loop, configs[4] substitute, planner panorama, densify event
timeout 400 python scripts/configs_report.py > gpurun_out/final/${TAG}_configs.json 2> gpurun_out/final/configs.err; echo configs rc=$?
timeout 200 python scripts/lookaround_times.py > gpurun_out/final/${TAG}_lookaround.txt 2>&1
N=3000000 timeout 200 python scripts/densify_time.py > gpurun_out/final/${TAG}_densify.txt 2>&1
ls gpurun_out/final | head -60
