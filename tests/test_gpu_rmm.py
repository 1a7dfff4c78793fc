"""-m gpu: librmm.so against the reference's tests/memory/memory_tests.cpp:52-192, in BOTH allocation modes
(CudaDefaultAllocation = hipMalloc per request, PoolAllocation = the caching pool that replaces cnmem)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

KB, MB, GB, TB, PB = 1 << 10, 1 << 20, 1 << 30, 1 << 40, 1 << 50
_MODE = [0]          # the allocation mode of the current `rmm` fixture (1 = PoolAllocation)


@pytest.fixture(params=[0, 1], ids=["default_allocation", "pool_allocation"])
def rmm(gdf, request):
    from libgdf_amd._binding import _rmm_cdll as lib, rmmOptions_t
    lib.rmmAlloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
    lib.rmmRealloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
    lib.rmmFree.argtypes = [C.c_void_p, C.c_void_p]
    lib.rmmGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_void_p]
    lib.rmmGetAllocationOffset.argtypes = [C.POINTER(C.c_long), C.c_void_p, C.c_void_p]
    lib.rmmInitialize.argtypes = [C.POINTER(rmmOptions_t)]
    lib.rmmGetErrorString.restype = C.c_char_p
    assert lib.rmmFinalize() == 0
    assert lib.rmmInitialize(C.byref(rmmOptions_t(request.param, 0, False))) == 0      # GdfTest fixture: init per case
    _MODE[0] = request.param
    yield lib
    assert lib.rmmFinalize() == 0
    assert lib.rmmInitialize(C.byref(rmmOptions_t(0, 0, False))) == 0                   # leave the session in default mode


def test_zero_sizes_and_bad_arguments(rmm):
    a = C.c_void_p()
    assert rmm.rmmAlloc(C.byref(a), 0, None) == 0                  # AllocateZeroBytes
    assert rmm.rmmAlloc(None, 0, None) == 0                        # NullPtrAllocateZeroBytes
    assert rmm.rmmAlloc(None, 4, None) == 2                        # NullPtrInvalidArgument -> RMM_ERROR_INVALID_ARGUMENT
    assert rmm.rmmGetErrorString(2) == b"RMM_ERROR_INVALID_ARGUMENT"
    assert rmm.rmmFree(None, None) == 0                            # FreeZero


@pytest.mark.parametrize("size", [4, KB, MB, GB], ids=["word", "KB", "MB", "GB"])
def test_allocate_and_free(rmm, size):
    a = C.c_void_p()
    assert rmm.rmmAlloc(C.byref(a), size, None) == 0 and a.value
    assert rmm.rmmFree(a, None) == 0


def test_allocate_too_much(rmm):
    free, total = C.c_size_t(), C.c_size_t()
    assert rmm.rmmGetInfo(C.byref(free), C.byref(total), None) == 0
    a = C.c_void_p()
    rc = rmm.rmmAlloc(C.byref(a), TB, None)                        # AllocateTB: fails unless the device really has it
    assert (rc != 0) == (TB > free.value)
    assert rmm.rmmFree(a, None) == 0
    a = C.c_void_p()
    assert rmm.rmmAlloc(C.byref(a), PB, None) != 0                 # AllocateTooMuch
    assert rmm.rmmFree(a, None) == 0


@pytest.mark.parametrize("first,second", [(MB, MB // 2), (GB, KB), (MB, 2 * MB), (KB, GB)],
                         ids=["smaller", "much_smaller", "larger", "much_larger"])
def test_reallocate(rmm, first, second):
    a = C.c_void_p()
    assert rmm.rmmAlloc(C.byref(a), first, None) == 0
    assert rmm.rmmRealloc(C.byref(a), second, None) == 0 and a.value
    assert rmm.rmmFree(a, None) == 0


def test_get_info_and_allocation_offset(rmm):
    fb, tb, fa, ta = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    assert rmm.rmmGetInfo(C.byref(fb), C.byref(tb), None) == 0
    a = C.c_void_p()
    assert rmm.rmmAlloc(C.byref(a), GB // 2, None) == 0
    assert rmm.rmmGetInfo(C.byref(fa), C.byref(ta), None) == 0
    assert ta.value >= tb.value and fa.value <= fb.value            # GetInfo: free memory goes down
    assert rmm.rmmFree(a, None) == 0
    a, b, off = C.c_void_p(), C.c_void_p(), C.c_long(-1)
    assert rmm.rmmAlloc(C.byref(a), KB, None) == 0 and rmm.rmmAlloc(C.byref(b), KB, None) == 0
    assert rmm.rmmGetAllocationOffset(C.byref(off), a, None) == 0 and off.value >= 0
    assert rmm.rmmGetAllocationOffset(C.byref(off), b, None) == 0 and off.value >= 0
    assert rmm.rmmFree(a, None) == 0 and rmm.rmmFree(b, None) == 0


def test_pool_reuses_freed_blocks(gdf):
    """PoolAllocation: a freed block serves the next request of its size class without a new hipMalloc."""
    from libgdf_amd._binding import _rmm_cdll as lib, rmmOptions_t
    lib.rmmAlloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p]
    lib.rmmFree.argtypes = [C.c_void_p, C.c_void_p]
    lib.rmmFinalize()
    lib.rmmInitialize(C.byref(rmmOptions_t(1, 0, False)))
    try:
        a, b = C.c_void_p(), C.c_void_p()
        assert lib.rmmAlloc(C.byref(a), 64 * MB, None) == 0
        first = a.value
        assert lib.rmmFree(a, None) == 0
        assert lib.rmmAlloc(C.byref(b), 64 * MB, None) == 0
        assert b.value == first
        assert lib.rmmFree(b, None) == 0
    finally:
        lib.rmmFinalize()
        lib.rmmInitialize(C.byref(rmmOptions_t(0, 0, False)))


def test_placed_blocks_keep_the_fastest_candidate(rmm):
    """Round 5: gdf_amd_rmm_place_* (include/memory.h, csrc/rmm.cpp) -- the pool that re-draws slow physical placements.  No counterpart
    in the reference.  The protocol with invented times: the first block of a (role, size) is the champion; while the pool is
    exploring, every allocation is a fresh challenger drawn while the champion (and the losers) are held, and the faster one stays;
    after `draws` challengers the champion serves every call and nothing is measured; a placed block may also come back through
    rmmFree; other sizes / the non-pool mode fall through to the plain allocator."""
    lib = rmm
    lib.gdf_amd_rmm_place_alloc.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.gdf_amd_rmm_place_free.argtypes = [C.c_int, C.c_void_p, C.c_float]
    lib.gdf_amd_rmm_place_draws.argtypes = [C.c_int]
    lib.gdf_amd_rmm_place_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    pool = _MODE[0] == 1
    lib.gdf_amd_rmm_place_draws(2)
    role, size = 9, GB + 5 * MB

    def alloc(max_draws=0):
        p, m = C.c_void_p(), C.c_int(-1)
        assert lib.gdf_amd_rmm_place_alloc(role, size, max_draws, C.byref(p), C.byref(m)) == 0
        return p.value, m.value
    if not pool:
        p, m = alloc()
        assert p and m == 0                              # CudaDefaultAllocation: a plain hipMalloc, nothing to measure
        assert lib.gdf_amd_rmm_place_free(role, p, -1.0) == 0
        lib.gdf_amd_rmm_place_draws(4)
        return
    stats = (C.c_ulonglong * 4)()
    lib.gdf_amd_rmm_place_stats(stats)
    drawn0, promoted0 = stats[0], stats[1]
    champ, m = alloc()
    assert champ and m == 1                              # the champion, to be timed
    assert lib.gdf_amd_rmm_place_free(role, champ, 10.0) == 0
    c1, m = alloc()
    assert c1 and c1 != champ and m == 1                 # challenger 1, drawn while the champion is held
    assert lib.gdf_amd_rmm_place_free(role, c1, 5.0) == 0            # faster: promoted
    c2, m = alloc()
    assert c2 not in (champ, c1) and m == 1              # challenger 2: neither the champion nor the held loser
    assert lib.gdf_amd_rmm_place_free(role, c2, 7.0) == 0            # slower than 5.0: dropped, exploration over
    for _ in range(3):
        p, m = alloc()
        assert p == c1 and m == 0                        # settled: the promoted block serves every call, unmeasured
        assert lib.rmmFree(p, None) == 0                 # ... and may come back through the plain entry point
    lib.gdf_amd_rmm_place_stats(stats)
    assert stats[0] - drawn0 == 2 and stats[1] - promoted0 == 1 and stats[3] == 0
    # a second shape of the same role is its own entry; a block below 1 GiB is the plain pool's
    small, ms = C.c_void_p(), C.c_int(-1)
    assert lib.gdf_amd_rmm_place_alloc(role, 64 * MB, 0, C.byref(small), C.byref(ms)) == 0 and ms.value == 0
    assert lib.gdf_amd_rmm_place_free(role, small, -1.0) == 0
    # the caller's own number of challengers (a calibration loop inside one call)
    role2 = 10
    seen = []
    for i in range(5):
        p, m = C.c_void_p(), C.c_int(-1)
        assert lib.gdf_amd_rmm_place_alloc(role2, size, 3, C.byref(p), C.byref(m)) == 0
        seen.append((p.value, m.value))
        assert lib.gdf_amd_rmm_place_free(role2, p, 4.0 - i if m.value else -1.0) == 0     # every candidate faster than the last
    assert [m for _, m in seen] == [1, 1, 1, 1, 0] and len({p for p, _ in seen[:4]}) == 4 and seen[4][0] == seen[3][0]
    # EARLY SETTLE: with at least four challengers drawn and a champion 7 % faster than the slowest candidate seen, the search ends
    # before its last draw (the join asks for up to sixteen: most searches end after four to six)
    role3 = 11
    times = [10.0, 10.1, 9.9, 10.0, 9.0, 9.5, 9.5]      # champion, then challengers; the fourth challenger is the fast one
    seen = []
    for i in range(7):
        p, m = C.c_void_p(), C.c_int(-1)
        assert lib.gdf_amd_rmm_place_alloc(role3, size, 12, C.byref(p), C.byref(m)) == 0
        seen.append((p.value, m.value))
        assert lib.gdf_amd_rmm_place_free(role3, p, times[i] if m.value else -1.0) == 0
    assert [m for _, m in seen] == [1, 1, 1, 1, 1, 0, 0] and seen[5][0] == seen[4][0] == seen[6][0]
    lib.gdf_amd_rmm_place_draws(4)
