O=gpurun_out/r2ad; mkdir -p $O
for m in 2; do
  echo "LOOKBACK=$m"; GDF_SCAN_LOOKBACK=$m timeout 120 python - <<'PY'
import numpy as np, torch, libgdf_amd as gdf
from libgdf_amd import Column
rs = np.random.RandomState(3)
for dt in (np.int8, np.int32, np.int64):
    for n in (1, 4095, 4096, 4097, 10_000_019, 100_000_000):
        a = rs.randint(-100, 100, size=n).astype(dt)
        for inc in (True, False):
            got = gdf.api.prefixsum(Column(torch.from_numpy(a).cuda()), inc).cpu().numpy()
            exp = np.cumsum(a, dtype=dt)
            exp = exp if inc else (exp - a).astype(dt)
            assert np.array_equal(got, exp), (dt, n, inc)
print('parity ok')
PY
done
for m in 0 1 2; do
  echo "LOOKBACK=$m"; GDF_SCAN_LOOKBACK=$m timeout 300 python tools/bench_ops.py 2>>$O/err.txt | grep prefixsum | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"
done
for w in 2 4 6; do
  echo "spine, WGS_PER_CU=$w"; GDF_SCAN_WGS_PER_CU=$w GDF_SCAN_LOOKBACK=2 timeout 300 python tools/bench_ops.py 2>>$O/err.txt | grep prefixsum | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(round(d['ms'],3), round(d['frac_of_8TBps'],3), d['kernels_ms'])"
done
tail -3 $O/err.txt
