ulimit -c 0
run() { echo "== $*"; env "$@" timeout 200 python tools/stress_join.py --seed 21 --case 237 2>&1 | grep -v amdgpu.ids | tail -2; }
run A=1
run GDF_JK_NO_SPARSE_OPT=1
run GDF_JK_NO_SKEW_SAMPLE=1
run GDF_JK_NO_SPEC=1
run GDF_JK_NO_DEFER=1
run GDF_JK_NO_FAST=1
run GDF_JK_DBG=512
echo "== case 57 seed 22"; timeout 200 python tools/stress_join.py --seed 22 --case 57 2>&1 | grep -v amdgpu.ids | tail -2
