"""-m gpu: gpu_comparison*, gpu_apply_stencil, gdf_filter and the mask helpers vs numpy expectations."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from oracle import oracle
from util import ALL_DTYPES, gen_rand, random_valid

pytestmark = pytest.mark.gpu


def _col(a, v=None):
    from libgdf_amd.columns import column_from_numpy
    return column_from_numpy(a, v)


@pytest.mark.parametrize("ldt", ALL_DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("rdt", ALL_DTYPES, ids=lambda d: np.dtype(d).name)
def test_comparison_all_dtype_pairs(gdf, ldt, rdt):
    """tests/filterops_numeric/test_filterops.cu:79-178 covers GDF_EQUALS over the 36 pairs, sizes 0-9; we run all six operators."""
    for n in (1, 9, 1000):
        l = gen_rand(ldt, n, -5, 5) if np.dtype(ldt).kind == "i" else np.round(gen_rand(ldt, n) * 5).astype(ldt)
        r = gen_rand(rdt, n, -5, 5) if np.dtype(rdt).kind == "i" else np.round(gen_rand(rdt, n) * 5).astype(rdt)
        for op in range(6):
            out = gdf.api.comparison(_col(l), _col(r), op)
            np.testing.assert_array_equal(out.to_numpy(), oracle.comparison(l, r, op))
            assert out.valid_bits().all() and out.c.null_count == 0


@pytest.mark.parametrize("ldt", ALL_DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("sdt", ALL_DTYPES, ids=lambda d: np.dtype(d).name)
def test_comparison_static(gdf, ldt, sdt):
    n = 5000
    l = gen_rand(ldt, n, -20, 20) if np.dtype(ldt).kind == "i" else np.round(gen_rand(ldt, n) * 20).astype(ldt)
    s = np.dtype(sdt).type(3)
    for op in range(6):
        out = gdf.api.comparison(_col(l), s, op)
        np.testing.assert_array_equal(out.to_numpy(), oracle.comparison(l, s, op))


def test_comparison_masks(gdf):
    n = 1003
    l, r = gen_rand(np.int32, n), gen_rand(np.int32, n)
    lv, rv = random_valid(n), random_valid(n)
    out = gdf.api.comparison(_col(l, lv), _col(r, rv), 0)
    np.testing.assert_array_equal(out.valid_bits(), lv & rv)      # filterops.cu:139-153: AND + recount
    assert out.c.null_count == n - (lv & rv).sum()
    out = gdf.api.comparison(_col(l, lv), np.int32(0), 4)
    np.testing.assert_array_equal(out.valid_bits(), lv)
    assert out.c.null_count == n - lv.sum()


@pytest.mark.parametrize("dtype", ALL_DTYPES, ids=lambda d: np.dtype(d).name)
@pytest.mark.parametrize("n", [1, 7, 64, 1000, 300007])
def test_apply_stencil(gdf, dtype, n):
    a = gen_rand(dtype, n)
    st = (np.random.random(n) < 0.4).astype(np.int8)
    sv = random_valid(n)
    out = gdf.api.apply_stencil(_col(a), _col(st, sv))
    exp = oracle.apply_stencil(a, st, sv)
    assert out.size == len(exp)
    np.testing.assert_array_equal(out.to_numpy(), exp)           # stable: input order kept
    bits = out.valid_bits(n)
    assert bits[: len(exp)].all() and not bits[len(exp):].any()


@pytest.mark.parametrize("keep", [0.0, 0.03, 1.0])
@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.float32, np.int64], ids=lambda d: np.dtype(d).name)
def test_apply_stencil_selectivity_extremes_and_unaligned_slices(gdf, dtype, keep):
    """The LDS-staged write kernel (whole 4096-row tiles, a ragged last one) at no / few / all rows kept, and -- a column that starts in
    the middle of an allocation is not 16-byte aligned -- the kernels it falls back to."""
    n = 3 * 4096 * 7 + 1234
    a = gen_rand(dtype, n + 3)
    st = (np.random.random(n + 3) < keep).astype(np.int8)
    import torch
    from libgdf_amd.columns import Column
    ta, ts = torch.from_numpy(a).cuda(), torch.from_numpy(st).cuda()
    for off in (0, 3):
        av, sv_ = a[off:off + n], st[off:off + n]
        out = gdf.api.apply_stencil(Column(ta[off:off + n]), Column(ts[off:off + n]))
        exp = av[sv_ != 0]
        assert out.size == len(exp)
        np.testing.assert_array_equal(out.to_numpy(), exp)


@pytest.mark.parametrize("case", ["default", "two-passes", "bail-out"])
@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.float32, np.int64], ids=lambda d: np.dtype(d).name)
def test_apply_stencil_one_pass_in_lockstep_rounds(gdf, force_path, dtype, case):
    """From 2^22 rows on gpu_apply_stencil is ONE pass (csrc/filter.hip stencil_rounds_kernel: every resident workgroup reads the kept-row
    counts of all tiles of its round and adds them up itself; reference: streamcompactionops.cu:162-205, a stable copy_if).  Against the
    oracle at 2 % / 40 % / 100 % kept with a stencil validity mask, a ragged last tile, fewer tiles than workgroups in the last round; the
    same request through the two passes (GDF_FL_NO_ROUNDS) and through a forced bail-out (the flag set before the launch: the kernel leaves,
    the two passes start over).  The launches are checked.  (By default only 8-byte elements take the rounds; GDF_FL_ROUNDS_ANY_WIDTH makes
    every width take them here, and test_apply_stencil_default_kernel_by_width checks the default's choice.)"""
    from bench import read_profile
    lib = gdf._binding._gdf_cdll
    if case == "two-passes":
        force_path("GDF_FL_NO_ROUNDS")
    else:
        force_path("GDF_FL_ROUNDS_ANY_WIDTH")          # (the default takes the rounds for 8-byte elements only: the narrower ones lose there)
    if case == "bail-out":
        force_path("GDF_FL_FORCE_BAIL")
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    rs = np.random.RandomState(23)
    for n, keep in (((1 << 22), 0.4), ((1 << 22) + 4096 * 5 + 77, 0.02), (9_000_001, 1.0)):
        a = gen_rand(dtype, n)
        st = (rs.random_sample(n) < keep).astype(np.int8)
        sv = rs.random_sample(n) > 0.1 if keep < 1.0 else None
        out = gdf.api.apply_stencil(_col(a), _col(st, sv))
        exp = oracle.apply_stencil(a, st, sv if sv is not None else np.ones(n, dtype=bool))
        assert out.size == len(exp), (n, keep)
        np.testing.assert_array_equal(out.to_numpy(), exp)
    lib.gdf_amd_profile_enable(0)
    names = {k.split("@")[0] for k in read_profile(gdf)}
    assert ("compact_rounds" in names) == (case != "two-passes"), names
    assert ("compact_write" in names) == (case != "default"), names


@pytest.mark.parametrize("dtype", [np.int8, np.int16, np.int32, np.int64, np.float64], ids=lambda d: np.dtype(d).name)
def test_apply_stencil_default_kernel_by_width(gdf, dtype):
    """The default from 2^22 rows on: 8-byte elements take the lockstep rounds, narrower ones the two passes (measured: csrc/filter.hip compact)."""
    from bench import read_profile
    lib = gdf._binding._gdf_cdll
    n = (1 << 22) + 12345
    a = gen_rand(dtype, n)
    st = (np.random.RandomState(5).random_sample(n) < 0.3).astype(np.int8)
    lib.gdf_amd_profile_reset(); lib.gdf_amd_profile_enable(1)
    out = gdf.api.apply_stencil(_col(a), _col(st))
    lib.gdf_amd_profile_enable(0)
    np.testing.assert_array_equal(out.to_numpy(), a[st != 0])
    names = {k.split("@")[0] for k in read_profile(gdf)}
    assert ("compact_rounds" in names) == (np.dtype(dtype).itemsize == 8), names


@pytest.mark.parametrize("ldt", ALL_DTYPES, ids=lambda d: np.dtype(d).name)
def test_comparison_vector_kernel_and_its_tail(gdf, ldt):
    """16-byte-vector compare (round 5): whole vectors + a tail of n % (16 / width) rows, against a column of the same width and a scalar."""
    n = 70_001
    a, b = gen_rand(ldt, n), gen_rand(ldt, n)
    b[::3] = a[::3]
    for op, fn in enumerate([np.equal, np.not_equal, np.less, np.less_equal, np.greater, np.greater_equal]):
        got = gdf.api.comparison(_col(a), _col(b), op).to_numpy()
        np.testing.assert_array_equal(got.astype(bool), fn(a, b))
        got = gdf.api.comparison(_col(a), a[7], op).to_numpy()
        np.testing.assert_array_equal(got.astype(bool), fn(a, a[7]))


def test_filter_then_compact_pipeline_large(gdf):
    """SURVEY 8d micro-metric shape at 20M rows: col > v, then compaction; ~10 % selectivity."""
    n = 20_000_000
    a = np.random.randint(0, 1000, size=n).astype(np.int64)
    st = gdf.api.comparison(_col(a), np.int64(899), 4)
    out = gdf.api.apply_stencil(_col(a), st)
    exp = a[a > 899]
    assert out.size == len(exp)
    np.testing.assert_array_equal(out.to_numpy(), exp)


def test_apply_stencil_errors(gdf):
    from libgdf_amd import GDFError
    a = _col(gen_rand(np.int32, 10), np.ones(10, dtype=bool))
    st = _col(np.ones(10, dtype=np.int8))
    with pytest.raises(GDFError, match="GDF_VALIDITY_UNSUPPORTED"):
        gdf.api.apply_stencil(a, st)


def test_gdf_filter_known_answer_and_random(gdf):
    with open(os.path.join(os.path.dirname(__file__), "golden", "sqls_known_answers.json")) as f:
        fx = json.load(f)["filter"]
    cols = [np.array(fx["cols"][c]["values"], dtype=fx["cols"][c]["dtype"]) for c in ("c0", "c1", "c2")]
    idx = gdf.api.filter_rows([_col(c) for c in cols], fx["tuple"]).cpu().numpy()
    assert len(idx) == fx["expected_size"] and list(idx) == fx["expected_indices"]
    n = 200000
    cols = [gen_rand(np.int32, n, 0, 4), gen_rand(np.int64, n, 0, 3), np.round(gen_rand(np.float64, n) * 2)]
    vals = [2, 1, 1.0]
    idx = gdf.api.filter_rows([_col(c) for c in cols], vals).cpu().numpy()
    np.testing.assert_array_equal(idx.astype(np.uint64), oracle.filter_rows(cols, vals))     # ascending (stable copy_if)


@pytest.mark.parametrize("n", [1, 8, 31, 32, 33, 1000, 100003])
def test_count_nonzero_mask(gdf, n):
    """tests/validops/valids-tests.cu: device popcount vs host popcount."""
    import torch
    from libgdf_amd import libgdf
    from libgdf_amd.columns import mask_from_bools
    v = np.random.randint(0, 2, size=n).astype(bool)
    mask = mask_from_bools(v)
    mask[(n + 7) // 8:] = 0xFF               # garbage after the last row must not count
    if n % 8:
        mask[n // 8] |= np.uint8(0xFF << (n % 8) & 0xFF)
    d = torch.from_numpy(mask).cuda()
    cnt = C.c_int(0)
    libgdf.gdf_count_nonzero_mask(d.data_ptr(), n, C.byref(cnt))
    assert cnt.value == int(v.sum())


def test_validity_and(gdf):
    """python/tests/test_validity.py:19-74."""
    import torch
    from libgdf_amd import Column, libgdf
    n = 1001
    a, b = random_valid(n), random_valid(n)
    ca, cb = _col(gen_rand(np.int32, n), a), _col(gen_rand(np.int32, n), b)
    out = Column(torch.empty(n, dtype=torch.int32, device="cuda"), torch.zeros(128, dtype=torch.uint8, device="cuda"), 3)
    libgdf.gdf_validity_and(ca.ptr, cb.ptr, out.ptr)
    np.testing.assert_array_equal(out.valid_bits(), a & b)
    assert out.c.null_count == n - (a & b).sum()


@pytest.mark.parametrize("nl,nr", [(0, 5), (5, 0), (8, 8), (13, 70), (1000, 1), (129, 255)])
@pytest.mark.parametrize("dtype", [np.int8, np.int32, np.float64], ids=lambda d: np.dtype(d).name)
def test_gpu_concat(gdf, nl, nr, dtype):
    """tests/filterops_numeric/test_gpu_concat.cu: output = lhs ++ rhs, data and validity bits, for left sizes that
    end inside a mask byte and on a byte boundary."""
    import ctypes as C
    import torch
    from libgdf_amd import Column, GDFError, libgdf
    from libgdf_amd.columns import get_dtype
    l, r = gen_rand(dtype, nl), gen_rand(dtype, nr)
    lv, rv = random_valid(nl) if nl else np.zeros(0, dtype=bool), random_valid(nr) if nr else np.zeros(0, dtype=bool)
    cl, cr = _col(l, lv), _col(r, rv)
    tdt = {np.dtype(np.int8): torch.int8, np.dtype(np.int32): torch.int32, np.dtype(np.float64): torch.float64}[np.dtype(dtype)]
    out = Column(torch.empty(max(nl + nr, 1), dtype=tdt, device="cuda"), torch.zeros(192, dtype=torch.uint8, device="cuda"),
                 get_dtype(np.dtype(dtype)), size=nl + nr)
    libgdf.gpu_concat(cl.ptr, cr.ptr, out.ptr)
    np.testing.assert_array_equal(out.to_numpy(), np.concatenate([l, r]))
    np.testing.assert_array_equal(out.valid_bits(), np.concatenate([lv, rv]))
    bad = Column(torch.empty(nl + nr + 1, dtype=tdt, device="cuda"), None, get_dtype(np.dtype(dtype)))
    with pytest.raises(GDFError, match="GDF_COLUMN_SIZE_MISMATCH"):            # streamcompactionops.cu:392
        libgdf.gpu_concat(cl.ptr, cr.ptr, bad.ptr)


def test_column_concat(gdf):
    """tests/column/column-test.cu: data + masks, a missing mask counts as all valid."""
    import torch
    from libgdf_amd import Column, libgdf
    from libgdf_amd.columns import column_array
    parts = [gen_rand(np.int64, n) for n in (5, 64, 1, 1000)]
    valids = [random_valid(5), None, np.array([False]), random_valid(1000)]
    cols = [_col(p, v) for p, v in zip(parts, valids)]
    total = sum(len(p) for p in parts)
    out = Column(torch.empty(total, dtype=torch.int64, device="cuda"), torch.zeros(192, dtype=torch.uint8, device="cuda"), 4)
    libgdf.gdf_column_concat(out.ptr, column_array(cols), len(cols))
    np.testing.assert_array_equal(out.to_numpy(), np.concatenate(parts))
    exp_valid = np.concatenate([v if v is not None else np.ones(len(p), dtype=bool) for p, v in zip(parts, valids)])
    np.testing.assert_array_equal(out.valid_bits(), exp_valid)
    assert out.c.null_count == total - exp_valid.sum()


def test_rmm_roundtrip_and_log(gdf):
    """librmm ABI: alloc/free/realloc/getinfo, CSV log header (python/tests/test_rmm.py:52)."""
    from libgdf_amd import librmm
    p = C.c_void_p()
    librmm.rmmAlloc(C.byref(p), 1 << 20, None)
    assert p.value
    off = C.c_long(-1)
    librmm.rmmGetAllocationOffset(C.byref(off), p, None)
    assert off.value == 0
    librmm.rmmRealloc(C.byref(p), 1 << 21, None)
    f, t = C.c_size_t(0), C.c_size_t(0)
    librmm.rmmGetInfo(C.byref(f), C.byref(t), None)
    assert 0 < f.value <= t.value
    librmm.rmmFree(p, None)
    n = librmm.rmmLogSize()
    buf = C.create_string_buffer(n + 1)
    librmm.rmmGetLog(buf, n)
    assert buf.value.decode().startswith("Event Type,Device ID,Address,Stream,Size (bytes),Free Memory,Total Memory,Current Allocs,Start,End,Elapsed")
