# round 5, third GPU call: (1) two half-tile workgroups per CU on the six-byte level-1 path (LAB: GDF_JK_SC_THREADS=512) against the
# one 1024-thread workgroup, alternating processes; (2) parity of that variant; (3) placement traces with the allocation budget; (4) tests
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5_c
mkdir -p $O
cd $R
B="python bench.py --steps 10 --warmup 6 --cpu-sample 0 --pandas-sample 0 --extra 0"
for i in 1 2 3; do
  for v in 1024 512; do
    LIBGDF_AMD_LAB=1 GDF_JK_SC_THREADS=$v $B 2>/dev/null | grep '^{' | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps({'sc_threads': $v, 'ms_per_step': d['ms_per_step'], 'probe_phase': d['roofline']['probe_phase']['kernels_ms'], 'kernels_ms_per_step': d['kernels_ms_per_step']}))" >> $O/half_tiles_ab.jsonl
  done
done
cut -c1-330 $O/half_tiles_ab.jsonl
LIBGDF_AMD_LAB=1 GDF_JK_SC_THREADS=512 timeout 900 python -m pytest tests/test_gpu_join.py -m gpu -q -x -k "six_byte_level1 or headline_configuration_properties or skewed_probe" 2>&1 | tail -4 > $O/pytest_half_tiles.txt; cat $O/pytest_half_tiles.txt
for i in 1 2; do
  python tools/gpu/r5_place.py 4 > $O/place_trace_$i.json 2>$O/place_trace_$i.err; python -c "
import json; d=json.load(open('$O/place_trace_$i.json')); print(d['first_calls_wall_ms'], d['settled_ms_per_join'], d['ms']); print('\n'.join(d['trace'][:26]))"
done
timeout 1500 python -m pytest tests/test_gpu_sort.py tests/test_gpu_c5.py tests/test_gpu_rmm.py -m gpu -q -x --durations=8 > $O/pytest_subset.txt 2>&1; tail -16 $O/pytest_subset.txt
