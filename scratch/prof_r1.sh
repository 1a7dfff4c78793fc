set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r1d
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.txt
python bench.py 2>&1 | tail -1 > $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o join -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/pmc_write.log 2>&1
find $O -name "*.csv" | head -20
cat $O/pytest_gpu.txt; cat $O/bench.json
