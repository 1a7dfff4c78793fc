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
  for (size_t i = begin + threadIdx.x; i < end; i += SCAN_THREADS) acc += (ACC)in[i];
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
__global__ __launch_bounds__(SCAN_THREADS) void scan_spine(ACC *chunk_sum, int nchunks) {
  // exclusive scan of <= SCAN_MAX_CHUNKS values by one block
  __shared__ ACC wsum[SCAN_THREADS / WAVE];
  __shared__ ACC carry;
  if (threadIdx.x == 0) carry = 0;
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
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      v[k] = (t0 + k < end) ? (ACC)in[t0 + k] : 0;
      run += v[k];
    }
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
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const ACC o = inclusive ? pre + v[k] : pre;
      if (t0 + k < end) out[t0 + k] = (ELEM)o;
      pre += v[k];
    }
    carry += total;
    block_sync();
  }
}

template <class ACC, class ELEM>
gdf_error device_scan(const ELEM *in, ELEM *out, size_t n, bool inclusive) {
  if (n == 0) return GDF_SUCCESS;
  constexpr int ITEMS = 16 / sizeof(ELEM) >= 4 ? 8 : 4;
  constexpr size_t TILE = (size_t)SCAN_THREADS * ITEMS;
  // chunks are whole tiles so that thread-contiguous loads stay aligned
  size_t tiles = (n + TILE - 1) / TILE;
  size_t tiles_per_chunk = (tiles + SCAN_MAX_CHUNKS - 1) / SCAN_MAX_CHUNKS;
  const size_t chunk = tiles_per_chunk * TILE;
  const int nchunks = (int)((n + chunk - 1) / chunk);
  DevBuf sums;
  RMM_TRY(sums.alloc(sizeof(ACC) * nchunks));
  GDF_LAUNCH("scan_reduce", (scan_reduce<ACC, ELEM, ITEMS>), dim3(nchunks), dim3(SCAN_THREADS), 0, stream0(), in,
                     sums.as<ACC>(), n, chunk);
  hipLaunchKernelGGL((scan_spine<ACC>), dim3(1), dim3(SCAN_THREADS), 0, stream0(), sums.as<ACC>(), nchunks);
  GDF_LAUNCH("scan_apply", (scan_apply<ACC, ELEM, ITEMS>), dim3(nchunks), dim3(SCAN_THREADS), 0, stream0(), in, out,
                     sums.as<ACC>(), n, chunk, inclusive ? 1 : 0);
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
