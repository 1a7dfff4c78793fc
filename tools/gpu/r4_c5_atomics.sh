# C5 on the XCD-shared layout: how many L2 atomics does the scatter kernel issue, and do they stay in the L2 (TCC_ATOMIC vs TCC_EA0_ATOMIC = sent on to the memory side)?
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4atom
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/a -o c5 -- python $R/tools/bench_c5.py --reps 1 --no-checks > $O/a.log 2>&1
python $R/tools/pmc_summary.py $(find $O/a -name "*counter_collection.csv" | head -1) gbp_scatter > $O/xcd.txt
rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/b -o c5 -- python $R/tools/bench_c5.py --reps 1 --no-checks --force GDF_GBP_NO_XCD > $O/b.log 2>&1
python $R/tools/pmc_summary.py $(find $O/b -name "*counter_collection.csv" | head -1) gbp_scatter > $O/wg.txt
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete
echo XCD-shared; cat $O/xcd.txt; echo per-workgroup; cat $O/wg.txt
