# C5 after the claims' collection moved behind the regroup: four processes, the group-by GPU tests and the new fused-join test
set -x
TAG=${1:-r4ab5}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for i in 1 2 3 4; do python tools/bench_c5.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd', round(d['ms'],3), d['checks_pass'], d['kernels_ms'])" >> $O/c5.txt; done
python tools/bench_c5.py --null-keys 0.01 2>/dev/null | tail -1 | cut -c1-400 >> $O/c5.txt
cat $O/c5.txt
timeout 1500 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py tests/test_gpu_fused_join.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_subset.txt; cat $O/pytest_subset.txt
