O=gpurun_out/r2x; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_c5.py tests/test_gpu_groupby.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python tools/bench_c5.py 2>>$O/err.txt | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(round(d['ms'],2), round(d['frac_of_8TBps'],3), d['kernels_ms'], d['checks_pass'])"
tail -3 $O/err.txt
