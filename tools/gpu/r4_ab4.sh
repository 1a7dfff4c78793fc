# C5 A/B: XCD-shared segments against one segment per workgroup, alternating processes on one box (current build)
set -x
TAG=${1:-r4ab4}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
for i in 1 2 3 4; do
  python tools/bench_c5.py --force GDF_GBP_XCD 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('xcd', round(d['ms'],3), d['checks_pass'], d['kernels_ms'])" >> $O/c5_xcd_ab.txt
  python tools/bench_c5.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wg ', round(d['ms'],3), d['checks_pass'], d['kernels_ms'])" >> $O/c5_xcd_ab.txt
done
python tools/bench_c5.py --force GDF_GBP_XCD --null-keys 0.01 2>/dev/null | tail -1 | cut -c1-500 >> $O/c5_xcd_ab.txt
cat $O/c5_xcd_ab.txt
