ulimit -c 0
echo "seed 41 case 174 with stage clock"; GDF_JK_DBG=512 GDF_STRESS_VERBOSE=1 timeout 300 python tools/stress_join.py --seed 41 --case 174 --max-build 300000000 --max-probe 600000000 2>&1 | grep -v amdgpu | tail -25
echo "rc $?"
