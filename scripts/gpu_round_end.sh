# End-of-round evidence run: parity tests, bench line, rocprof kernel summary, PMC passes.  Run via gpurun from the repo root.
R=$PWD; TAG=${1:-r01}
mkdir -p gpurun_out/final
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/final/build_smoke.log 2>&1; echo build+smoke rc=$?
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -5 | tee gpurun_out/final/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; echo bench rc=$?; cat gpurun_out/final/bench.json
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final -o ${TAG}_bench -- python $R/bench.py --steps 50 --warmup 10 --no-extras > $R/gpurun_out/final/${TAG}_bench.log 2>&1; echo prof rc=$?
pmc() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/final -o ${TAG}_pmc_$name -- python $R/bench.py --steps 5 --warmup 2 --no-extras > $R/gpurun_out/final/${TAG}_pmc_$name.log 2>&1; echo pmc $name rc=$?; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pmc sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
cd $R; ls gpurun_out/final | head -40
# secondary evidence: kernel tables of the fused mapping iteration and of the 2M north-star configuration, configs report
bash scripts/profile_fused_iteration.sh > gpurun_out/final/fused_iteration.txt 2>&1; cp gpurun_out/prof/fiter_kernel_stats.csv gpurun_out/final/${TAG}_fused_iteration_kernel_stats.csv
bash scripts/profile_2m.sh > gpurun_out/final/north_star_2m.txt 2>&1; cp gpurun_out/prof2m/n2m_kernel_stats.csv gpurun_out/final/${TAG}_2m_kernel_stats.csv
python scripts/configs_report.py > gpurun_out/final/${TAG}_configs.json 2> gpurun_out/final/configs.err
tail -3 gpurun_out/final/fused_iteration.txt; head -2 gpurun_out/final/north_star_2m.txt
