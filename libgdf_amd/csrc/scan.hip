// scan.hip -- prefix sums for gdf_prefixsum_{i8,i32,i64,generic} and for the
// library's own histogram / offset scans.
//
// Replaces the reference's cub::DeviceScan calls (src/scan.cu:11-76).  Semantics
// kept: sum in the column's own dtype with wrap-around, inclusive or exclusive,
// equal size & dtype required, valid masks rejected.
//
// Shape: ONE pass with decoupled look-back (scan_lookback): 2*w bytes per element, the algorithmic minimum.
//   * a tile is 256 threads x 16 elements (i8: 4 KB, i32: 16 KB, i64: 32 KB); tile ids come from a ticket counter, so
//     every predecessor of a running tile is running or done (HIP promises no dispatch order);
//   * the tile is read with coalesced 16-byte loads, wave w owning 1024 consecutive elements, and scanned in that
//     (round, lane, element) order -- one wave scan per round of 64 vectors (16 consecutive elements per thread read
//     directly are 64 separate 64-byte requests per wave instruction: the three-pass kernels below cap at ~4.9 TB/s on
//     that; a transposition through LDS costs 37 KB per workgroup, i.e. half the occupancy);
//   * the tile publishes its AGGREGATE, then the whole workgroup looks back over 256 predecessors per round -- thread t
//     at tile - 1 - t -- for the nearest tile with a published INCLUSIVE prefix, summing the aggregates in between, and
//     publishes its own inclusive prefix.  A published value is one 8-byte word {flag, 32 data bits} written by a
//     single agent-scope store, so data and flag cannot be seen apart (64-bit sums are two such words).  Why 256-wide:
//     a look-back round costs one cross-XCD round trip (~1.5-2 us), so tiles can only resolve at window / round-trip --
//     with the usual one-wave window (64) that is ~35 tiles per us = 1.1 TB/s of 32 KB tiles; 256 lanes give 4x that.
// The reduce-then-scan shape of round 1 (three launches, 3*w bytes: scan_reduce -> scan_spine -> scan_apply) stays for
// unaligned columns and as the A/B reference (GDF_SCAN_3PASS=1).
#include "internal.h"

#include <cstdlib>

namespace gdf_amd {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_MAX_CHUNKS = 2048;

// ACC: accumulator type (uint32 for 1- and 4-byte columns, uint64 for 8-byte);
// ELEM: storage type.  Unsigned arithmetic gives the same bits as signed wrap.
template <class ACC, class ELEM, int ITEMS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce(const ELEM *__restrict__ in, ACC *__restrict__ chunk_sum,
                                                            size_t n, size_t chunk) {
  __shared__ ACC wsum[SCAN_THREADS / WAVE];
  const size_t begin = (size_t)blockIdx.x * chunk;
  const size_t end = begin + chunk < n ? begin + chunk : n;
  ACC acc = 0;
  size_t i = begin + threadIdx.x;
  for (; i + 7 * SCAN_THREADS < end; i += 8 * SCAN_THREADS) {      // 8 independent loads in flight per thread
    ELEM v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = in[i + (size_t)k * SCAN_THREADS];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += (ACC)v[k];
  }
  for (; i < end; i += SCAN_THREADS) acc += (ACC)in[i];
  acc = wave_reduce_add(acc);
  if (lane_id() == 0) wsum[threadIdx.x / WAVE] = acc;
  block_sync();
  if (threadIdx.x == 0) {
    ACC s = 0;
    for (int w = 0; w < SCAN_THREADS / WAVE; ++w) s += wsum[w];
    chunk_sum[blockIdx.x] = s;
  }
}

template <class ACC>
__global__ __launch_bounds__(SCAN_THREADS) void scan_spine(ACC *chunk_sum, int nchunks, ACC *running) {
  // exclusive scan of <= SCAN_MAX_CHUNKS values by one block, seeded with *running (the total of the
  // segments before this one), which is advanced by this segment's total
  __shared__ ACC wsum[SCAN_THREADS / WAVE];
  __shared__ ACC carry;
  if (threadIdx.x == 0) carry = *running;
  block_sync();
  for (int base = 0; base < nchunks; base += SCAN_THREADS) {
    const int i = base + threadIdx.x;
    ACC v = i < nchunks ? chunk_sum[i] : 0;
    ACC incl = wave_scan_incl(v);
    if (lane_id() == WAVE - 1) wsum[threadIdx.x / WAVE] = incl;
    block_sync();
    ACC woff = 0;
    for (int w = 0; w < (int)(threadIdx.x / WAVE); ++w) woff += wsum[w];
    const ACC c = carry;
    if (i < nchunks) chunk_sum[i] = c + woff + incl - v;
    block_sync();
    if (threadIdx.x == SCAN_THREADS - 1) carry = c + woff + incl;
    block_sync();
  }
  if (threadIdx.x == 0) *running = carry;
}

template <class ACC, class ELEM, int ITEMS>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const ELEM *in, ELEM *out,   // in == out allowed
                                                           const ACC *__restrict__ chunk_off, size_t n, size_t chunk,
                                                           int inclusive) {
  __shared__ ACC wsum[SCAN_THREADS / WAVE];
  const size_t begin = (size_t)blockIdx.x * chunk;
  const size_t end = begin + chunk < n ? begin + chunk : n;
  ACC carry = chunk_off[blockIdx.x];
  constexpr size_t TILE = (size_t)SCAN_THREADS * ITEMS;
  for (size_t tile = begin; tile < end; tile += TILE) {
    const size_t t0 = tile + (size_t)threadIdx.x * ITEMS;   // this thread owns ITEMS consecutive elements
    ACC v[ITEMS];
    ACC run = 0;
    const bool full = tile + TILE <= end;                  // block-uniform: whole tiles load and store without guards
    if (full) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) v[k] = (ACC)in[t0 + k];
    } else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) v[k] = (t0 + k < end) ? (ACC)in[t0 + k] : 0;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) run += v[k];
    const ACC incl = wave_scan_incl(run);
    if (lane_id() == WAVE - 1) wsum[threadIdx.x / WAVE] = incl;
    block_sync();
    ACC woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / WAVE; ++w) {
      if (w < (int)(threadIdx.x / WAVE)) woff += wsum[w];
      total += wsum[w];
    }
    ACC pre = carry + woff + incl - run;   // exclusive prefix of this thread's first element
    ELEM o[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      o[k] = (ELEM)(inclusive ? pre + v[k] : pre);
      pre += v[k];
    }
    if (full) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) out[t0 + k] = o[k];
    } else {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k)
        if (t0 + k < end) out[t0 + k] = o[k];
    }
    carry += total;
    block_sync();
  }
}


// ---------------------------------------------------------------------------
// single pass, decoupled look-back
// ---------------------------------------------------------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));     // native vector: non-temporal builtins take it
constexpr int LB_THREADS = 256;
constexpr int LB_ITEMS = 16;
constexpr int LB_TILE = LB_THREADS * LB_ITEMS;
constexpr unsigned long long LB_FLAG = 1ull << 32;

// tile state: NW = sizeof(ACC) / 4 words of aggregate, then NW words of inclusive prefix; word = LB_FLAG | 32 data bits
template <class ACC>
__device__ __forceinline__ void lb_publish(unsigned long long *slot, ACC v) {
#pragma unroll
  for (int i = 0; i < (int)sizeof(ACC) / 4; ++i)
    __hip_atomic_store(slot + i, LB_FLAG | (unsigned long long)(uint32_t)((uint64_t)v >> (32 * i)), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
template <class ACC>
__device__ __forceinline__ bool lb_read(const unsigned long long *slot, ACC &v) {
  unsigned long long w[sizeof(ACC) / 4];
#pragma unroll
  for (int i = 0; i < (int)sizeof(ACC) / 4; ++i) w[i] = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bool ok = true;
  uint64_t r = 0;
#pragma unroll
  for (int i = 0; i < (int)sizeof(ACC) / 4; ++i) {
    ok = ok && (w[i] & LB_FLAG);
    r |= (uint64_t)(uint32_t)w[i] << (32 * i);
  }
  v = (ACC)r;
  return ok;
}

template <class ACC, class ELEM>
__global__ __launch_bounds__(LB_THREADS) void scan_lookback(const ELEM *in, ELEM *out, size_t n, int inclusive,
                                                            unsigned long long *state, uint32_t *ticket) {
  constexpr int NW = sizeof(ACC) / 4;
  constexpr int VEC = 16 / sizeof(ELEM);           // elements per 16-byte vector
  constexpr int VPT = LB_ITEMS / VEC;              // vectors per thread (i8: 1, i32: 4, i64: 8)
  constexpr int NWAVES = LB_THREADS / WAVE;
  constexpr int SEG = WAVE * LB_ITEMS;             // elements per wave
  __shared__ ACC wsum[NWAVES];
  __shared__ unsigned long long s_m2[NWAVES], s_m0[NWAVES];
  __shared__ uint32_t s_tile;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u);
  block_sync();
  const uint32_t tile = s_tile;
  const size_t base = (size_t)tile * LB_TILE;
  const bool full = base + LB_TILE <= n;
  const int wave = threadIdx.x / WAVE, lane = lane_id();

  // ---- load + local scan.  No LDS transposition (its 37 KB per workgroup halved the occupancy of a kernel that lives on
  // bytes in flight): wave w owns SEG consecutive elements and reads them as 16-byte vectors, vector k * 64 + lane in
  // round k -- 1 KB contiguous per load instruction.  The order of the elements is then (round, lane, element within the
  // vector): one wave scan of the per-vector sums per ROUND instead of one per tile.  Partial tiles (the last one) take
  // plain guarded loads of 16 consecutive elements per thread, which is the same scheme with one round of 16-element
  // "vectors" per wave.
  union Vec { u32x4 q; ELEM e[VEC]; };
  Vec vv[VPT];
  ACC excl_in_wave[VPT];                            // prefix of vector (k, lane) inside the wave's segment
  ACC wave_total = 0;
  if (full) {
    const u32x4 *src = reinterpret_cast<const u32x4 *>(in + base + (size_t)wave * SEG);     // 16-byte aligned: the host checked
#pragma unroll
    for (int k = 0; k < VPT; ++k) vv[k].q = __builtin_nontemporal_load(src + k * WAVE + lane);
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      ACC sum = 0;
#pragma unroll
      for (int e = 0; e < VEC; ++e) sum += (ACC)vv[k].e[e];
      const ACC inc = wave_scan_incl(sum);
      excl_in_wave[k] = wave_total + inc - sum;
      wave_total += __shfl(inc, WAVE - 1, WAVE);
    }
  } else {
    const size_t t0 = base + (size_t)threadIdx.x * LB_ITEMS;      // vv holds this thread's 16 consecutive elements
    ACC sum = 0;
#pragma unroll
    for (int k = 0; k < VPT; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const size_t i = t0 + k * VEC + e;
        vv[k].e[e] = i < n ? in[i] : (ELEM)0;
        sum += (ACC)vv[k].e[e];
      }
    const ACC inc = wave_scan_incl(sum);
    excl_in_wave[0] = inc - sum;
    wave_total = __shfl(inc, WAVE - 1, WAVE);
  }
  if (lane == 0) wsum[wave] = wave_total;
  block_sync();
  ACC woff = 0, aggregate = 0;
#pragma unroll
  for (int w = 0; w < NWAVES; ++w) {
    if (w < wave) woff += wsum[w];
    aggregate += wsum[w];
  }
  unsigned long long *mine = state + (size_t)tile * 2 * NW;
  if (threadIdx.x == 0) lb_publish<ACC>(mine, aggregate);

  // ---- look-back, the whole workgroup: thread t examines tile - 1 - t (a tile before the first one has prefix 0) ----
  ACC exclusive = 0;
  long long nearest = (long long)tile - 1;
  for (;;) {
    const long long p = nearest - (long long)threadIdx.x;
    int st = 2;
    ACC val = 0;
    if (p >= 0) {
      const unsigned long long *theirs = state + (size_t)p * 2 * NW;
      if (!lb_read<ACC>(theirs + NW, val)) st = lb_read<ACC>(theirs, val) ? 1 : 0;
    }
    const unsigned long long m2 = __ballot(st == 2), m0 = __ballot(st == 0);
    block_sync();                 // the previous round's readers of s_m2 / s_m0 / wsum are done
    if (lane == 0) { s_m2[wave] = m2; s_m0[wave] = m0; }
    block_sync();
    int first_incl = -1;          // thread index of the nearest published inclusive prefix in this window
    bool wait = false;            // some tile nearer than that has published nothing yet
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) {
      if (first_incl < 0 && !wait) {
        const unsigned long long a = s_m2[w], z = s_m0[w];
        if (a) {
          const int c = __ffsll((long long)a) - 1;
          if (z & ((1ull << c) - 1ull)) wait = true;
          else first_incl = w * WAVE + c;
        } else if (z) {
          wait = true;
        }
      }
    }
    if (wait) { __builtin_amdgcn_s_sleep(4); continue; }      // workgroup-uniform
    const ACC c = (first_incl < 0 || (int)threadIdx.x <= first_incl) ? val : (ACC)0;
    const ACC part = wave_reduce_add(c);
    if (lane == 0) wsum[wave] = part;
    block_sync();
#pragma unroll
    for (int w = 0; w < NWAVES; ++w) exclusive += wsum[w];
    if (first_incl >= 0) break;
    nearest -= LB_THREADS;
  }
  if (threadIdx.x == 0) lb_publish<ACC>(mine + NW, (ACC)(exclusive + aggregate));

  // ---- outputs ----
  const ACC wave_base = exclusive + woff;
  if (full) {
    u32x4 *dst = reinterpret_cast<u32x4 *>(out + base + (size_t)wave * SEG);
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
      ACC pre = wave_base + excl_in_wave[k];
      Vec o;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const ACC x = (ACC)vv[k].e[e];
        o.e[e] = (ELEM)(inclusive ? pre + x : pre);
        pre += x;
      }
      __builtin_nontemporal_store(o.q, dst + k * WAVE + lane);
    }
  } else {
    const size_t t0 = base + (size_t)threadIdx.x * LB_ITEMS;
    ACC pre = wave_base + excl_in_wave[0];
#pragma unroll
    for (int k = 0; k < VPT; ++k)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const size_t i = t0 + k * VEC + e;
        const ACC x = (ACC)vv[k].e[e];
        if (i < n) out[i] = (ELEM)(inclusive ? pre + x : pre);
        pre += x;
      }
  }
}

template <class ACC, class ELEM>
static gdf_error device_scan_lookback(const ELEM *in, ELEM *out, size_t n, bool inclusive) {
  constexpr int NW = sizeof(ACC) / 4;
  const size_t ntiles = (n + LB_TILE - 1) / LB_TILE;
  DevBuf st;
  const size_t state_bytes = sizeof(unsigned long long) * ntiles * 2 * NW;
  RMM_TRY(st.alloc(state_bytes + sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(st.p, 0, state_bytes + sizeof(unsigned long long), stream0()));
  uint32_t *ticket = reinterpret_cast<uint32_t *>(st.as<unsigned char>() + state_bytes);
  GDF_LAUNCH("scan_lookback", (scan_lookback<ACC, ELEM>), dim3((unsigned)ntiles), dim3(LB_THREADS), 0, stream0(), in, out, n,
             inclusive ? 1 : 0, st.as<unsigned long long>(), ticket);
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));   // scratch is released on return
  return GDF_SUCCESS;
}

// GDF_SCAN_SEG_MB=<n> scans the input in segments of n MiB (reduce -> spine -> apply per segment) so that the apply
// pass re-reads what the reduce pass has just pulled through the 256 MiB Infinity Cache.  Measured on 1e8 int64:
// scan_apply drops from 0.33 to 0.27 ms at 128 MiB segments, but the smaller grids slow scan_reduce (0.17 -> 0.24 ms)
// and the extra launches add gaps -- 0.64 ms against 0.55 ms for the whole array in one go.  So the default is one
// segment; the switch stays for larger inputs / other parts.
template <class ACC, class ELEM>
gdf_error device_scan(const ELEM *in, ELEM *out, size_t n, bool inclusive) {
  if (n == 0) return GDF_SUCCESS;
  // one pass whenever the columns allow 16-byte accesses (a column may be a slice of a larger buffer) and the tile count
  // fits the 32-bit ticket
  static const bool three_pass = getenv("GDF_SCAN_3PASS") != nullptr;
  if (!three_pass && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && n / LB_TILE < 0x7fffffffULL)
    return device_scan_lookback<ACC, ELEM>(in, out, n, inclusive);
  constexpr int ITEMS = 16 / sizeof(ELEM) >= 4 ? 8 : 4;
  constexpr size_t TILE = (size_t)SCAN_THREADS * ITEMS;
  static const size_t seg_bytes = getenv("GDF_SCAN_SEG_MB") ? (size_t)atoll(getenv("GDF_SCAN_SEG_MB")) << 20 : 0;
  size_t seg = seg_bytes ? seg_bytes / sizeof(ELEM) / TILE * TILE : (n + TILE - 1) / TILE * TILE;
  if (seg < TILE) seg = TILE;
  DevBuf sums, running;
  RMM_TRY(sums.alloc(sizeof(ACC) * SCAN_MAX_CHUNKS));
  RMM_TRY(running.alloc(sizeof(ACC)));
  HIP_TRY(hipMemsetAsync(running.p, 0, sizeof(ACC), stream0()));
  for (size_t s0 = 0; s0 < n; s0 += seg) {
    const size_t m = n - s0 < seg ? n - s0 : seg;
    // chunks are whole tiles so that thread-contiguous loads stay aligned
    const size_t tiles = (m + TILE - 1) / TILE;
    const size_t tiles_per_chunk = (tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
    const size_t chunk = tiles_per_chunk * TILE;
    const int nchunks = (int)((m + chunk - 1) / chunk);
    GDF_LAUNCH("scan_reduce", (scan_reduce<ACC, ELEM, ITEMS>), dim3(nchunks), dim3(SCAN_THREADS), 0, stream0(), in + s0, sums.as<ACC>(), m, chunk);
    hipLaunchKernelGGL((scan_spine<ACC>), dim3(1), dim3(SCAN_THREADS), 0, stream0(), sums.as<ACC>(), nchunks, running.as<ACC>());
    GDF_LAUNCH("scan_apply", (scan_apply<ACC, ELEM, ITEMS>), dim3(nchunks), dim3(SCAN_THREADS), 0, stream0(), in + s0, out + s0, sums.as<ACC>(), m, chunk,
               inclusive ? 1 : 0);
  }
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));   // scratch is released on return
  return GDF_SUCCESS;
}

// internal entry points used by partition / join / group-by host code
gdf_error scan_u32(const uint32_t *in, uint32_t *out, size_t n, bool inclusive) {
  return device_scan<uint32_t, uint32_t>(in, out, n, inclusive);
}
gdf_error scan_u64(const uint64_t *in, uint64_t *out, size_t n, bool inclusive) {
  return device_scan<uint64_t, uint64_t>(in, out, n, inclusive);
}

}  // namespace gdf_amd

using namespace gdf_amd;

static gdf_error prefixsum_checked(gdf_column *inp, gdf_column *out, int inclusive, gdf_dtype expect) {
  (void)expect;
  GDF_REQUIRE(inp && out, GDF_DATASET_EMPTY);
  GDF_REQUIRE(inp->size == out->size, GDF_COLUMN_SIZE_MISMATCH);   // scan.cu:55
  GDF_REQUIRE(inp->dtype == out->dtype, GDF_UNSUPPORTED_DTYPE);     // scan.cu:56
  GDF_REQUIRE(!inp->valid, GDF_VALIDITY_UNSUPPORTED);               // scan.cu:57
  GDF_REQUIRE(!out->valid, GDF_VALIDITY_UNSUPPORTED);               // scan.cu:58
  return GDF_SUCCESS;
}

extern "C" {

gdf_error gdf_prefixsum_i8(gdf_column *inp, gdf_column *out, int inclusive) {
  GDF_TRY(prefixsum_checked(inp, out, inclusive, GDF_INT8));
  return device_scan<uint32_t, uint8_t>((const uint8_t *)inp->data, (uint8_t *)out->data, inp->size, inclusive != 0);
}
gdf_error gdf_prefixsum_i32(gdf_column *inp, gdf_column *out, int inclusive) {
  GDF_TRY(prefixsum_checked(inp, out, inclusive, GDF_INT32));
  return device_scan<uint32_t, uint32_t>((const uint32_t *)inp->data, (uint32_t *)out->data, inp->size,
                                         inclusive != 0);
}
gdf_error gdf_prefixsum_i64(gdf_column *inp, gdf_column *out, int inclusive) {
  GDF_TRY(prefixsum_checked(inp, out, inclusive, GDF_INT64));
  return device_scan<uint64_t, uint64_t>((const uint64_t *)inp->data, (uint64_t *)out->data, inp->size,
                                         inclusive != 0);
}
gdf_error gdf_prefixsum_generic(gdf_column *inp, gdf_column *out, int inclusive) {
  GDF_REQUIRE(inp, GDF_DATASET_EMPTY);
  switch (inp->dtype) {   // scan.cu:66-76: other dtypes silently succeed
    case GDF_INT8: return gdf_prefixsum_i8(inp, out, inclusive);
    case GDF_INT32: return gdf_prefixsum_i32(inp, out, inclusive);
    case GDF_INT64: return gdf_prefixsum_i64(inp, out, inclusive);
    default: return GDF_SUCCESS;
  }
}

}  // extern "C"
