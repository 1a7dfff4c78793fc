# round 3, first call: the whole GPU suite on the hygiene + masked-read build, baselines of this box, the L2-probe lab.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3a
mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt
cat $O/pytest_gpu.txt
python bench.py --steps 10 --warmup 3 2>$O/bench.err | grep '^{' | tail -1 > $O/bench.json
cut -c1-900 $O/bench.json
python tools/bench_shapes.py --only c3_headline,c3_masked,c3_masked_99pct_valid,c3_materialise_2_payload_cols,c3_half_hit,c3_80pct_hit,dup4_build_keys > $O/bench_shapes.jsonl 2>$O/bench_shapes.err
cut -c1-700 $O/bench_shapes.jsonl
python tools/bench_c5.py 2>/dev/null | tail -1 > $O/bench_c5.json
python tools/bench_c5.py --null-keys 0.01 2>/dev/null | tail -1 > $O/bench_c5_nullkeys.json
cut -c1-500 $O/bench_c5.json $O/bench_c5_nullkeys.json
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/l2probe_lab.hip -o /tmp/l2probe_lab && timeout 600 /tmp/l2probe_lab > $O/l2probe_lab.txt 2>&1
cat $O/l2probe_lab.txt
