# re-entry check of the restored tree: GPU suite, headline line, a longer randomised join stress with fresh seeds
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2ba; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > $O/pytest_gpu.txt
python bench.py --cpu-sample 0 --pandas-sample 0 2>$O/bench.err | grep '^{' | tail -1 > $O/bench.json
timeout 200 python tools/stress_join.py --seconds 150 --seed 77 > $O/stress77.txt 2>&1
timeout 200 python tools/stress_join.py --seconds 120 --seed 78 --max-build 30000000 --max-probe 200000000 > $O/stress78.txt 2>&1
cat $O/pytest_gpu.txt; cut -c1-300 $O/bench.json; tail -3 $O/stress77.txt $O/stress78.txt
