# C5 scatter ablations on the LAB build: where does gbp_scatter_hot's time go?  usage: bash tools/gpu/r4_c5_ablate.sh <tag>
TAG=${1:-r4c}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
export LIBGDF_AMD_LAB=1
line() { python tools/bench_c5.py --reps 3 --no-checks 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'ms': round(d['ms'],3), 'k': d['kernels_ms'], 'pass': d['checks_pass']}))"; }
for dbg in ${GDF_ABLATE_LIST:-0 1 2 4 3 6 7 8 16}; do echo "GDF_GBP_HOT_DBG=$dbg $(GDF_GBP_HOT_DBG=$dbg line)" >> $O/ablate.txt; done
echo "GDF_GBP_NO_HOT=1 $(GDF_GBP_NO_HOT=1 line)" >> $O/ablate.txt
cat $O/ablate.txt
