// common.h -- internal helpers shared by the host dispatch code and the CDNA4 kernels.
//
// Nothing in here is part of the ABI.  Kernels are written for gfx950 only:
// 64-wide wavefronts (ballots are 64-bit), 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stddef.h>
#include <limits.h>
#include <new>

#include "gdf/gdf.h"
#include "memory.h"

// ---------------------------------------------------------------------------
// error plumbing (same contract as the reference's errorutils.h:8-30:
// a failed runtime call -> GDF_CUDA_ERROR, a failed allocation ->
// GDF_MEMORYMANAGER_ERROR; no exception ever crosses the C boundary)
// ---------------------------------------------------------------------------
#define HIP_TRY(call)                                                        \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) { gdf_amd::note_hip_error(_e, #call, __FILE__, __LINE__); return GDF_CUDA_ERROR; } \
  } while (0)
#define RMM_TRY(call)   do { if ((call) != RMM_SUCCESS) return GDF_MEMORYMANAGER_ERROR; } while (0)
#define GDF_TRY(call)   do { gdf_error _g = (call); if (_g != GDF_SUCCESS) return _g; } while (0)
#define GDF_REQUIRE(cond, err) do { if (!(cond)) return (err); } while (0)
#define HIP_CHECK_LAST() HIP_TRY(hipGetLastError())

namespace gdf_amd {

void note_hip_error(hipError_t e, const char *what, const char *file, int line);

// Every extern "C" entry point that can reach a host allocation (std::vector / new in the dispatch code) runs its body through
// this: the reference lets std::bad_alloc and thrust::system_error escape extern "C" (managed_allocator.cuh:34-45,
// thrust_rmm_allocator.h:44-49) -- a C caller (cffi / ctypes) cannot catch them.  Here an exception
// becomes GDF_MEMORYMANAGER_ERROR.  tests/test_abi.py / test_gpu_stress.py force one through gdf_amd_debug_force
// ("GDF_FORCE_HOST_ALLOC_FAILURE": make_key_table throws std::bad_alloc).
template <class F>
static inline gdf_error guarded(F &&body) noexcept {
  try {
    return body();
  } catch (...) {
    return GDF_MEMORYMANAGER_ERROR;
  }
}

constexpr int WAVE = 64;          // gfx950 wavefront width
constexpr int NUM_CU = 256;       // MI355X
constexpr int MAX_KEY_COLS = 16;  // key columns per relational call (reference tests use <= 5)

// all library work runs on the legacy default stream, like the reference
// (SURVEY.md 8b "Threading / streams").
static inline hipStream_t stream0() { return (hipStream_t)0; }

// ---- device scratch RAII over librmm ---------------------------------------
struct DevBuf {
  void *p = nullptr;
  // PLACED blocks (memory.h: gdf_amd_rmm_place_alloc): the multi-GB scratch of the regroup passes says what it is for, and -- while
  // the pool is still comparing physical placements for this (role, size) -- how long the kernels that scatter into it took:
  // clock_begin / clock_end bracket those launches with HIP events on the launch stream, reset() hands the sum to the pool.
  int role = -1;
  bool measure = false;
  bool borrowed = false;      // borrow(): p points into somebody else's allocation, nothing is freed here
  hipEvent_t ev[8];
  int nev = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  ~DevBuf() { reset(); }
  rmmError_t alloc(size_t bytes) { reset(); borrowed = false; return rmmAlloc(&p, bytes ? bytes : 1, (cudaStream_t)0); }
  rmmError_t alloc_placed(int role_, size_t bytes, int max_draws = 0) {
    reset();
    borrowed = false;
    int m = 0;
    const rmmError_t r = gdf_amd_rmm_place_alloc(role_, bytes ? bytes : 1, max_draws, &p, &m);
    if (r == RMM_SUCCESS) { role = role_; measure = m != 0; }
    return r;
  }
  void clock_mark(hipStream_t s) {
    if (!measure || nev >= 8) return;
    if (hipEventCreate(&ev[nev]) != hipSuccess) { (void)hipGetLastError(); measure = false; return; }
    (void)hipEventRecord(ev[nev++], s);
  }
  void clock_begin(hipStream_t s) { if (!(nev & 1)) clock_mark(s); }
  void clock_end(hipStream_t s) { if (nev & 1) clock_mark(s); }
  void borrow(void *q) { reset(); p = q; borrowed = q != nullptr; }
  void reset() {
    if (!p) return;
    if (borrowed) { p = nullptr; borrowed = false; return; }
    if (role >= 0) {
      float ms = -1.f;
      if (measure && nev >= 2 && !(nev & 1)) {
        ms = 0.f;
        for (int i = 0; i < nev; i += 2) {
          float t = 0.f;
          if (hipEventSynchronize(ev[i + 1]) != hipSuccess || hipEventElapsedTime(&t, ev[i], ev[i + 1]) != hipSuccess) { (void)hipGetLastError(); ms = -1.f; break; }
          ms += t;
        }
      }
      for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
      nev = 0;
      gdf_amd_rmm_place_free(role, p, ms);
      role = -1;
      measure = false;
    } else {
      rmmFree(p, (cudaStream_t)0);
    }
    p = nullptr;
  }
  // (a placed block handed on is freed through rmmFree, which knows it; its unreported events go with the hand-over)
  void *release() {
    for (int i = 0; i < nev; ++i) (void)hipEventDestroy(ev[i]);
    nev = 0;
    void *q = p;
    p = nullptr;
    role = -1;
    measure = false;
    borrowed = false;         // (a borrowed pointer handed on stays the lender's to free)
    return q;
  }
  template <class T> T *as() const { return static_cast<T *>(p); }
};

// ---- dtype helpers ----------------------------------------------------------
// Storage class of a column element: what the bytes are, ignoring date/time tags
// (gdf_table.cuh:704-854 maps DATE32->int32, DATE64/TIMESTAMP->int64 the same way).
enum ElemKind : int { K_I8 = 0, K_I16, K_I32, K_I64, K_F32, K_F64, K_BAD };

static inline ElemKind elem_kind(gdf_dtype t) {
  switch (t) {
    case GDF_INT8: return K_I8;
    case GDF_INT16: return K_I16;
    case GDF_INT32: case GDF_DATE32: return K_I32;
    case GDF_INT64: case GDF_DATE64: case GDF_TIMESTAMP: return K_I64;
    case GDF_FLOAT32: return K_F32;
    case GDF_FLOAT64: return K_F64;
    default: return K_BAD;
  }
}
static inline int kind_width(ElemKind k) {
  switch (k) { case K_I8: return 1; case K_I16: return 2; case K_I32: case K_F32: return 4;
               case K_I64: case K_F64: return 8; default: return -1; }
}
static inline int dtype_width(gdf_dtype t) { return kind_width(elem_kind(t)); }

static inline size_t mask_bytes(size_t rows) { return (rows + 7) / 8; }

// POD view of the key columns of one table, passed to kernels BY VALUE (the
// reference passed host-constructed C++ objects through unified memory,
// gdf_table.cuh:242; we never do that).
struct ColView {
  const void    *data;
  const uint8_t *valid;   // may be null
  int            kind;    // ElemKind
  int            width;   // bytes
};
struct KeyTable {
  int     ncols;
  int     any_valid;      // 1 if some column carries a mask
  int64_t nrows;
  ColView col[MAX_KEY_COLS];
};

gdf_error make_key_table(gdf_column **cols, int ncols, KeyTable *out);

// grid sizing for streaming kernels: enough blocks to fill 256 CUs several
// times over, capped so per-block partial state stays small.
static inline int stream_grid(size_t items, int per_block, int max_blocks = NUM_CU * 8) {
  size_t b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (b > (size_t)max_blocks) b = max_blocks;
  return (int)b;
}

#ifdef __HIPCC__
// ---------------------------------------------------------------------------
// wave64 primitives
// ---------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return __lane_id(); }

// Workgroup barrier that also drains this wave's outstanding LDS operations.
// hipcc (ROCm 7.2, gfx950) lowers __syncthreads() after NON-RETURNING LDS atomics
// (ds_add_u32 ...) to a bare s_barrier: no s_waitcnt lgkmcnt(0).  A wave can then
// pass the barrier with its last wave-instruction of LDS atomics still in flight and
// another wave reads the counters 64 increments short (seen as a ~1 % flaky
// histogram in jk_hist).  The explicit wait is free
// when nothing is outstanding.
__device__ __forceinline__ void block_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}

// number of set bits of `m` strictly below this lane
__device__ __forceinline__ int mask_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// match-any over the low `bits` bits of v: the live lanes of the wave that hold this lane's value (junk for a dead
// lane).  Every lane of the wave must call it.
__device__ __forceinline__ unsigned long long wave_match(uint32_t v, int bits, bool live) {
  unsigned long long peers = __ballot(live);
  for (int b = 0; b < bits; ++b) {
    const bool bit = (v >> b) & 1;
    const unsigned long long m = __ballot(bit);
    peers &= bit ? m : ~m;
  }
  return peers;
}

// One LDS atomic per DISTINCT counter per wave instead of one per lane: `counter[idx] += 1` for every live lane;
// returns the value a per-lane atomicAdd would have returned (lane order within the wave).  Few counters (a partition
// fan-out of 2..16) make 64 lanes queue on the same LDS address otherwise.
__device__ __forceinline__ uint32_t wave_aggregated_inc(uint32_t *counter, uint32_t idx, int bits, bool live) {
  const unsigned long long peers = wave_match(idx, bits, live);
  const int lane = lane_id();
  const int leader = __ffsll((long long)peers) - 1;
  uint32_t old = 0;
  if (live && lane == leader) old = atomicAdd(&counter[idx], (uint32_t)__popcll(peers));
  old = __shfl(old, live ? leader : lane);
  return old + (uint32_t)mask_rank(peers);
}

template <class T>
__device__ __forceinline__ T wave_reduce_add(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}

// inclusive scan across the 64 lanes of a wave, all lanes active.  DPP data movement (row_shr inside the rows of 16 lanes, then
// row_bcast:15 / row_bcast:31 across them -- gfx9 modes): six VALU steps instead of six ds_bpermute round trips through the LDS
// queue (~60 cycles each, behind whatever LDS traffic the wave has outstanding).  A lane without a source adds 0.
template <int CTRL, int ROW_MASK, bool BOUND>
__device__ __forceinline__ uint32_t dpp_move0(uint32_t x) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, BOUND);
}
template <int CTRL, int ROW_MASK, bool BOUND, class T>
__device__ __forceinline__ T dpp_move0_t(T x) {
  if constexpr (sizeof(T) == 8) {
    const uint64_t u = (uint64_t)x;
    const uint32_t lo = dpp_move0<CTRL, ROW_MASK, BOUND>((uint32_t)u), hi = dpp_move0<CTRL, ROW_MASK, BOUND>((uint32_t)(u >> 32));
    return (T)(((uint64_t)hi << 32) | lo);
  } else {
    return (T)dpp_move0<CTRL, ROW_MASK, BOUND>((uint32_t)x);
  }
}
// PRECONDITIONS: (1) T is an integer type -- the DPP moves copy value bits through uint32 halves, a float would be converted, not
// copied; (2) EVERY lane of the wave is active at the call (full EXEC): row_bcast:15 / :31 read lanes 15 / 31 / 47, and an inactive
// source lane contributes 0 (bound_ctrl), silently dropping its row's total.  All call sites sit in workgroup-uniform control flow of
// kernels whose block size is a multiple of 64.
template <class T>
__device__ __forceinline__ T wave_scan_incl(T v) {
  static_assert(std::is_integral<T>::value && (sizeof(T) == 4 || sizeof(T) == 8), "32- or 64-bit integers");
  v += dpp_move0_t<0x111, 0xf, true>(v);      // row_shr:1
  v += dpp_move0_t<0x112, 0xf, true>(v);      // row_shr:2
  v += dpp_move0_t<0x114, 0xf, true>(v);      // row_shr:4
  v += dpp_move0_t<0x118, 0xf, true>(v);      // row_shr:8  -> every row of 16 lanes holds its own inclusive scan
  v += dpp_move0_t<0x142, 0xa, false>(v);     // row_bcast:15 into rows 1 and 3
  v += dpp_move0_t<0x143, 0xc, false>(v);     // row_bcast:31 into rows 2 and 3
  return v;
}
// sum of wave_tot[0 .. my wave): NWAVES independent LDS reads and selects.  (A loop `for (w < my wave)` has a wave-dependent trip
// count: one LDS round trip per iteration, fifteen in a row for the last wave of a 1024-thread workgroup -- between two barriers
// of a regroup tile, so the whole workgroup waits for it.)
template <int NWAVES, class T>
__device__ __forceinline__ T waves_before_sum(const T *wave_tot, uint32_t tid) {
  const int mine = (int)(tid / WAVE);
  T v[NWAVES], sum = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) v[w] = wave_tot[w];
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) sum += w < mine ? v[w] : (T)0;
  return sum;
}

__device__ __forceinline__ bool bit_is_set(const uint8_t *mask, int64_t i) {
  return (mask[i >> 3] >> (i & 7)) & 1;   // LSB-first, include/gdf/utils.h:9-16
}

// row validity = AND over the key columns' masks (gdf_table.cuh:62-98), computed
// inline instead of materialising a row mask per call.
__device__ __forceinline__ bool row_valid(const KeyTable &t, int64_t i) {
  if (!t.any_valid) return true;
  bool ok = true;
  for (int c = 0; c < t.ncols; ++c)
    if (t.col[c].valid) ok = ok && bit_is_set(t.col[c].valid, i);
  return ok;
}

// raw element bits zero-extended to 64 (exact for equality on integer kinds)
__device__ __forceinline__ uint64_t load_bits(const ColView &c, int64_t i) {
  switch (c.width) {
    case 1: return ((const uint8_t *)c.data)[i];
    case 2: return ((const uint16_t *)c.data)[i];
    case 4: return ((const uint32_t *)c.data)[i];
    default: return ((const uint64_t *)c.data)[i];
  }
}

// typed equality of one element pair; floats compare with ==, so NaN never
// equals anything and -0.0 == +0.0 (gdf_table.cuh:580-691 rows_equal).
__device__ __forceinline__ bool elem_equal(const ColView &a, int64_t i, const ColView &b, int64_t j) {
  switch (a.kind) {
    case K_F32: return ((const float *)a.data)[i] == ((const float *)b.data)[j];
    case K_F64: return ((const double *)a.data)[i] == ((const double *)b.data)[j];
    default: return load_bits(a, i) == load_bits(b, j);
  }
}
__device__ __forceinline__ bool rows_equal(const KeyTable &a, int64_t i, const KeyTable &b, int64_t j) {
  for (int c = 0; c < a.ncols; ++c)
    if (!elem_equal(a.col[c], i, b.col[c], j)) return false;
  return true;
}

// 64-bit finaliser (bijective xorshift-multiply mix) used for INTERNAL partition
// ids and LDS slot numbers.  Not visible through the ABI: the public hash is
// Murmur3_32 in hash.cuh.
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
#endif  // __HIPCC__

}  // namespace gdf_amd
