set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_f
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q --durations=40 > $O/pytest_gpu.txt 2>&1; tail -60 $O/pytest_gpu.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; cut -c1-1200 $O/bench.json
