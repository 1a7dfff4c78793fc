cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc1/a -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --probe-rows 200000000 > $R/gpurun_out/pmc1/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc1/b -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --probe-rows 200000000 > $R/gpurun_out/pmc1/b.log 2>&1
find $R/gpurun_out/pmc1 -name "*.csv" | head; 
