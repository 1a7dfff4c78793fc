# instruction mix of the join kernels (one C3 join) -> gpurun_out/pmc_probe
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_probe
mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $O/a -o j -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/a.log 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/b -o j -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/b.log 2>&1
cd $R
for d in a b; do python tools/pmc_summary.py $(find $O/$d -name "*counter_collection.csv" | head -1) jk_ ; done
find $O -name "*.csv" -size +1M -delete
