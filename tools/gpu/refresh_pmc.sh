# after a kernel-source change: refresh what bench.py's roofline.traffic reads (profiles/*_pmc_hbm.json is stamped with the
# kernel build id) and the headline line.  usage: bash tools/gpu/refresh_pmc.sh r2zz
TAG=${1:-r2zz}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pandas-sample 0 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o join -- python $R/bench.py --steps 1 --warmup 0 --cpu-sample 0 --pandas-sample 0 > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o join -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --pandas-sample 0 > $O/trace.log 2>&1
cd $R
python tools/rocprof_summary.py $O/trace $O/kernel_stats.md
python tools/pmc_hbm_json.py $(find $O/pmc_fetch -name "*counter_collection.csv" | head -1) $(find $O/pmc_write -name "*counter_collection.csv" | head -1) $O/pmc_hbm.json
cp $O/pmc_hbm.json $R/profiles/zz_tmp_pmc_hbm.json          # so that THIS run's bench line already carries the traffic
python bench.py --steps 20 --warmup 5 2>$O/bench.err | grep '^{' | tail -1 > $O/bench.json
rm -f $R/profiles/zz_tmp_pmc_hbm.json
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" > $O/pytest_gpu.txt
rm -rf $O/trace/*/*.db; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*.db" -delete
cat $O/pytest_gpu.txt; cut -c1-400 $O/bench.json
