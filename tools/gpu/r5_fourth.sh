set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_d
mkdir -p $O
cd $R
python tools/gpu/r5_c5_modes.py 10 > $O/c5_modes.jsonl 2>$O/c5_modes.err; cat $O/c5_modes.jsonl | cut -c1-300
LIBGDF_AMD_LAB=1 GDF_JK_SC_THREADS=512 GDF_JK_TRACE=1 timeout 900 python -m pytest tests/test_gpu_join.py -m gpu -q -k "headline_configuration_properties or six_byte_level1" 2>&1 | tail -60 > $O/pytest_half_tiles.txt; cut -c1-300 $O/pytest_half_tiles.txt
LIBGDF_AMD_LAB=1 timeout 900 python -m pytest tests/test_gpu_join.py -m gpu -q -k "headline_configuration_properties" 2>&1 | tail -5
python bench.py --force-distributed --strategy fused --steps 5 --warmup 2 --probe-rows 1000000000 --build-rows 125000000 --cpu-sample 0 --pandas-sample 0 2>$O/fd.err | grep '^{' | tail -1 > $O/bench_force_distributed_fused.json; python -c "
import json; d=json.load(open('$O/bench_force_distributed_fused.json')); print(d['ms_per_step'], d['config'].get('nranks'), d['kernels_ms_per_step'])"; tail -3 $O/fd.err
python tools/sim_c4_fused.py 2>/dev/null | tail -4 > $O/sim_c4_fused.txt; cat $O/sim_c4_fused.txt
