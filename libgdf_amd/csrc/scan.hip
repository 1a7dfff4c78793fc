// scan.hip -- prefix sums for gdf_prefixsum_{i8,i32,i64,generic} and for the
// library's own histogram / offset scans.
//
// Replaces the reference's cub::DeviceScan calls (src/scan.cu:11-76).  Semantics
// kept: sum in the column's own dtype with wrap-around, inclusive or exclusive,
// equal size & dtype required, valid masks rejected.
//
// Shape (reduce-then-scan, three launches on the default stream):
//   1. scan_reduce : every block sums one contiguous chunk           (read N)
//   2. scan_spine  : one block scans the <= MAX_CHUNKS chunk sums
//   3. scan_apply  : every block re-reads its chunk and writes the scan seeded
//                    with its chunk offset                        (read N, write N)
// Inside a block a tile is 256 threads x ITEMS consecutive elements: per-thread
// serial scan, wave64 shuffle scan of the thread totals, one LDS hop across the
// four waves.  Algorithmic bytes are 2*w per element; this shape moves 3*w.
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

// GDF_SCAN_SEG_MB=<n> scans the input in segments of n MiB (reduce -> spine -> apply per segment) so that the apply
// pass re-reads what the reduce pass has just pulled through the 256 MiB Infinity Cache.  Measured on 1e8 int64:
// scan_apply drops from 0.33 to 0.27 ms at 128 MiB segments, but the smaller grids slow scan_reduce (0.17 -> 0.24 ms)
// and the extra launches add gaps -- 0.64 ms against 0.55 ms for the whole array in one go.  So the default is one
// segment; the switch stays for larger inputs / other parts.
template <class ACC, class ELEM>
gdf_error device_scan(const ELEM *in, ELEM *out, size_t n, bool inclusive) {
  if (n == 0) return GDF_SUCCESS;
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
