# C5: gbp_scatter_static with the next tile's key words requested before the flush (default) against GDF_GBP_NO_PIPELINE=1; three alternations
# (the kernel's time is bimodal from process to process on one box: 8.0 / 9.5 ms in tools/gpu/r2bf.sh with identical code)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bg; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_groupby.txt
for i in 1 2 3; do
  python tools/bench_c5.py --reps 5 2>/dev/null | tail -1 > $O/c5_pipe_$i.json
  GDF_GBP_NO_PIPELINE=1 python tools/bench_c5.py --reps 5 2>/dev/null | tail -1 > $O/c5_nopipe_$i.json
done
cat $O/pytest_groupby.txt; for f in pipe_1 nopipe_1 pipe_2 nopipe_2 pipe_3 nopipe_3; do echo $f; python -c "import json,sys; d=json.load(open('$O/c5_$f.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; done
