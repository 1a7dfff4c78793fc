# C5: does the number of chunks (= how wide a window of every partition the resident workgroups write into) matter?  + the new signature tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bc; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py -m gpu -q -x -k "static_signatures" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_sig.txt
for ch in 1024 4096 16384 512; do
  GDF_GBP_CHUNKS=$ch python tools/bench_c5.py --reps 3 2>/dev/null | tail -1 > $O/c5_chunks_$ch.json
done
cat $O/pytest_sig.txt; for ch in 1024 4096 16384 512; do echo $ch; python -c "import json,sys; d=json.load(open('$O/c5_chunks_$ch.json')); print(round(d['ms'],2), d['kernels_ms'], d['checks_pass'])"; done
