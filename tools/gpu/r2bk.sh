# insurance: longer randomised stress with fresh seeds on the build the round ends on
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bk; mkdir -p $O; cd $R
timeout 400 python tools/stress_join.py --seconds 240 --seed 101 > $O/stress101.txt 2>&1
timeout 400 python tools/stress_join.py --seconds 200 --seed 102 --max-build 40000000 --max-probe 300000000 > $O/stress102.txt 2>&1
GDF_STRESS_SECONDS=120 timeout 1500 python -m pytest tests/test_gpu_stress.py -m gpu -q -x -k "not test_join_properties" 2>&1 | grep -E "passed|failed|error|Error|assert" | head -10 > $O/pytest_stress.txt
tail -n 2 $O/stress101.txt; tail -n 2 $O/stress102.txt; cat $O/pytest_stress.txt
