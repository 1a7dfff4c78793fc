"""-m gpu: librmm.so against the reference's tests/memory/memory_tests.cpp:52-192, in BOTH allocation modes
(CudaDefaultAllocation = hipMalloc per request, PoolAllocation = the caching pool that replaces cnmem)."""
import ctypes as C

import pytest

pytestmark = pytest.mark.gpu

KB, MB, GB, TB, PB = 1 << 10, 1 << 20, 1 << 30, 1 << 40, 1 << 50


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
