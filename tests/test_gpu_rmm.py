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


def test_placed_blocks_size_classes_hold_and_bounded_searches(rmm):
    """Round 6: a champion serves a size CLASS -- requests of its role between six tenths of its block
    and all of it -- so a caller whose relations change size keeps it; a larger request makes a new entry (rounded up to an eighth of an
    octave) that replaces the smaller one; max_draws < 0 HOLDS: the champion, unmeasured, nothing drawn, the search stays open; a search
    holds a bounded number of losers next to champion and challenger; rmmGetInfo counts idle champions and held losers as free."""
    lib = rmm
    lib.gdf_amd_rmm_place_alloc.argtypes = [C.c_int, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]
    lib.gdf_amd_rmm_place_free.argtypes = [C.c_int, C.c_void_p, C.c_float]
    lib.gdf_amd_rmm_place_stats.argtypes = [C.POINTER(C.c_ulonglong)]
    lib.gdf_amd_rmm_place_min.argtypes = [C.c_size_t]
    if _MODE[0] != 1:
        pytest.skip("placed blocks exist in pool mode only")
    role = 21

    def alloc(size, max_draws):
        p, m = C.c_void_p(), C.c_int(-1)
        assert lib.gdf_amd_rmm_place_alloc(role, size, max_draws, C.byref(p), C.byref(m)) == 0
        return p.value, m.value

    def free_info():
        f, t = C.c_size_t(), C.c_size_t()
        assert lib.rmmGetInfo(C.byref(f), C.byref(t), None) == 0
        return f.value
    lib.gdf_amd_rmm_place_min(32 * MB)                       # the hook the callers' calibration tests use
    try:
        base = 96 * MB                                       # + an eighth of headroom, rounded up to 8 MiB: a 112 MiB block, serves 68 ... 112 MiB
        free0 = free_info()
        champ, m = alloc(base, 6)
        assert champ and m == 1
        assert lib.gdf_amd_rmm_place_free(role, champ, 10.0) == 0
        # HOLD: the champion, unmeasured, no challenger -- as often as asked
        for _ in range(3):
            p, m = alloc(base, -1)
            assert p == champ and m == 0
            assert lib.gdf_amd_rmm_place_free(role, p, -1.0) == 0
        # the size class: 90 MiB (>= 0.6 of 112) is served by the same entry and continues ITS search
        c1, m = alloc(90 * MB, 12)
        assert c1 and c1 != champ and m == 1
        assert lib.gdf_amd_rmm_place_free(role, c1, 10.1) == 0           # slower: a held loser
        held = [c1]
        for i in range(8):                                   # eight more slow challengers: at most five losers are held at any time
            c, m = alloc(100 * MB, 12)
            assert m == 1 and c != champ and c not in held[-4:]          # (not the champion, not one of the losers still held)
            held.append(c)
            assert lib.gdf_amd_rmm_place_free(role, c, 10.1 + 0.04 * i) == 0       # (never 7 % slower than the champion: no early settle)
            assert abs(free_info() - free0) < 32 * MB        # champion idle + losers held: all counted as free
        p, m = alloc(base, -1)
        assert p == champ and m == 0                         # the champion survived nine slower candidates
        assert lib.gdf_amd_rmm_place_free(role, p, -1.0) == 0
        # 60 MiB is below six tenths of the block: its own entry; 120 MiB does not fit: a new, larger entry (144 MiB) that takes
        # the 112 MiB entry's place
        stats = (C.c_ulonglong * 4)()
        lib.gdf_amd_rmm_place_stats(stats)
        n0 = stats[2]
        small, m = alloc(60 * MB, 6)
        assert small != champ and m == 1
        assert lib.gdf_amd_rmm_place_free(role, small, 5.0) == 0
        big, m = alloc(120 * MB, 6)
        assert big not in (champ, small) and m == 1
        assert lib.gdf_amd_rmm_place_free(role, big, 5.0) == 0
        lib.gdf_amd_rmm_place_stats(stats)
        assert stats[2] == n0 + 1                            # + the 60 MiB entry, + the 144 MiB entry, - the 112 MiB entry it replaced
        p, m = alloc(110 * MB, -1)
        assert p == big                                      # 110 MiB requests are the larger entry's now
        assert lib.gdf_amd_rmm_place_free(role, p, -1.0) == 0
    finally:
        lib.gdf_amd_rmm_place_min(0)

