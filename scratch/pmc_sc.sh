# HBM traffic of jk_scatter1 with 1024- vs 512-thread tiles (is the excess over 8.8 GB per launch spill traffic?)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_sc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for t in 1024 512; do
  for c in FETCH_SIZE WRITE_SIZE; do
    GDF_JK_SC_THREADS=$t rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${t}_$c -o j -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 > $O/${t}_$c.log 2>&1
  done
  python $R/tools/pmc_hbm_json.py $(find $O/${t}_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find $O/${t}_WRITE_SIZE -name "*counter_collection.csv" | head -1) $O/pmc_$t.json
  python -c "
import json; d=json.load(open('$O/pmc_$t.json'))['kernels']
for k in ('jk_scatter1','jk_scatter2','jk_probe_write'): print($t, k, d[k]['fetch_kb_reported'], d[k]['write_kb_reported'])
"
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete
