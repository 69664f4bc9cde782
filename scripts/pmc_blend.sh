# rocprofv3 PMC passes over the C2 workload (counters only; no trace domains besides kernel-trace)
R=$PWD; export TMPDIR=/tmp; mkdir -p $R/gpurun_out/pmc; cd /tmp
rocprofv3 -L > $R/gpurun_out/pmc/counters_list.txt 2>&1
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc -o $name -- python $R/bench.py --steps 5 --warmup 2 --no-extras > $R/gpurun_out/pmc/$name.log 2>&1; echo $name rc=$?; }
run p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run p3 GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_INSTS_FLAT
run p4 FETCH_SIZE
run p5 WRITE_SIZE
cd $R; ls gpurun_out/pmc | head -30
