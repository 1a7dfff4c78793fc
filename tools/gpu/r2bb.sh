# C5: static-signature gbp_count / gbp_scatter (all of a tile's requests issued together, four barriers per tile) against the old kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2bb; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_groupby.py tests/test_gpu_c5.py "tests/test_gpu_stress.py::test_group_by_random_shapes_against_the_oracle" -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 > $O/pytest_groupby.txt
python tools/bench_c5.py 2>/dev/null | tail -1 > $O/c5_new.json
GDF_GBP_OLD=1 python tools/bench_c5.py 2>/dev/null | tail -1 > $O/c5_old.json
python tools/bench_c5.py 2>/dev/null | tail -1 > $O/c5_new2.json
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | head -5 > $O/pytest_gpu.txt
cat $O/pytest_groupby.txt $O/pytest_gpu.txt; for f in c5_new c5_old c5_new2; do cut -c1-600 $O/$f.json; echo; done
