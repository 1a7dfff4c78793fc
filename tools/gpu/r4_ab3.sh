# C5 A/B: plain ranking (decided from the sample) against the leader ballots, alternating processes on one box; then the group-by GPU tests
set -x
TAG=${1:-r4ab3}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for i in 1 2 3; do
  python tools/bench_c5.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain  ', round(d['ms'],3), d['checks_pass'], d['kernels_ms'])" >> $O/c5_rank_ab.txt
  python tools/bench_c5.py --force GDF_GBP_PLAIN_RANK=0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ballots', round(d['ms'],3), d['checks_pass'], d['kernels_ms'])" >> $O/c5_rank_ab.txt
done
python tools/bench_c5.py --null-keys 0.01 2>/dev/null | tail -1 | cut -c1-500 >> $O/c5_rank_ab.txt
cat $O/c5_rank_ab.txt
timeout 1500 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py tests/test_gpu_sort.py -m gpu -q -x 2>&1 | tail -5 > $O/pytest_subset.txt; cat $O/pytest_subset.txt
