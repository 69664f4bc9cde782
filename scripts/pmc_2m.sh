# North-star configuration (2 M Gaussians, 640x480, SH degree 3, forward + backward): rocprofv3 kernel table and PMC passes past
# the 256 MiB Infinity Cache (working set ~1 GB).  One counter group per pass, --kernel-trace only (no other trace domain).
# usage (from the repo root, on the GPU box): bash scripts/pmc_2m.sh r02 ; then python scripts/pmc_summarise.py gpurun_out/pmc2m r02_2m
R=$PWD; TAG=${1:-r02}; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc2m; cd /tmp
export SH=${SH-3} N=${N-2000000} STEPS=${STEPS-8} WARMUP=${WARMUP-3}
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc2m -o ${TAG}_2m -- python $R/scripts/stage_times.py > $R/gpurun_out/pmc2m/${TAG}_2m.log 2>&1; echo stats rc=$?
pmc() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc2m -o ${TAG}_2m_pmc_$name -- python $R/scripts/stage_times.py > $R/gpurun_out/pmc2m/${TAG}_2m_pmc_$name.log 2>&1; echo pmc $name rc=$?; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
pmc sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
cd $R; tail -2 gpurun_out/pmc2m/${TAG}_2m.log
python scripts/pmc_summarise.py gpurun_out/pmc2m ${TAG}_2m
