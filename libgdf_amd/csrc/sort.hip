// sort.hip -- gdf_order_by and the SORT-method group-by.
//
// Reference path being replaced: src/sqls_ops.cu:1134-1289 (gdf_group_by_single, GDF_SORT
// branch), :1373-1392 (gdf_order_by) over src/sqls_rtti_comp.hpp:299-320 (thrust::sort of
// a row permutation with the RTTI comparator LesserRTTI::less: one dtype switch per
// column per comparison, ~log2(N) dependent gathers per row) and :397-662
// (thrust::gather + thrust::reduce_by_key with LesserRTTI::equal).  Here:
//
//   order_rows    The key columns are folded, last column group first, into 64-bit
//                 ORDER-PRESERVING unsigned images (sign bit flipped for integers;
//                 IEEE total-order trick for floats with -0.0 folded onto +0.0 so that
//                 rows that compare == are adjacent) and the (image, row) pairs go
//                 through a stable LSD radix sort, 8 bits per pass.  Digits on which
//                 every key agrees are skipped (the OR of key ^ key[0], reduced while the
//                 images are built, decides), so int64 keys < 2^32 cost 4 passes, not 8.  Columns
//                 that do not fit one 64-bit image are handled as further stable sorts
//                 (LSD over column groups), gathering through the current permutation.
//   rs_scatter    one pass: every wave owns a CONTIGUOUS run of the tile (so stability
//                 is (wave, round, lane) order), ranks its keys with 8 ballots per key
//                 (match-any) against a per-wave LDS digit counter, the tile is
//                 regrouped by digit in LDS and leaves as runs of equal digit, at
//                 offsets from one exclusive scan over the (digit, tile) count matrix.
//   sg_*          group boundaries from adjacent-row equality (typed ==, so NaN rows
//                 stay singletons and -0.0 == +0.0, as LesserRTTI::equal), group ids by
//                 prefix sum, and a wave-level SEGMENTED reduction of the gathered
//                 aggregation column: segmented shuffle scan per 64 sorted rows, the open
//                 segment carried in registers across rounds, one atomic per
//                 (wave, group) pair.  COUNT and the AVG divisor are differences of
//                 group start offsets -- no accumulation at all.
//
// Semantics kept: output rows in ascending lexicographic key order; aggregation in the
// INPUT dtype (sum/static_cast<ValsT>(n) for AVG, sqls_rtti_comp.hpp:651-657); COUNT
// in the OUTPUT column's dtype (sqls_ops.cu:272-400); COUNT_DISTINCT (flag_distinct)
// writes the number of groups into element 0 and reports size 1 (:440-446 of the
// rtti header); out_col_indices receives size_t row numbers (quirk 7) and
// gdf_order_by's d_indx is size_t as well; valid masks -> GDF_VALIDITY_UNSUPPORTED.
// The reference's sort is unstable, so which row of a group out_col_indices names is
// unspecified there; this implementation names the LAST row of the group in input
// order, which is what the reference's own known-answer test expects
// (tests/sqls/sqls_g_tester.cu:250-256: indices {5,0,2,4}).  flag_sorted == 1 means the
// caller promises the rows are already ordered: the permutation is the identity.
// NaN keys order after +inf (the reference's comparator leaves them unordered).
#include "internal.h"

#include <algorithm>
#include <new>
#include <vector>

namespace gdf_amd {

constexpr int SG_THREADS = 256;
constexpr int SG_ROUNDS = 16;                        // 64 x 16 sorted rows per wave

enum SgOp : int { SG_SUM = 0, SG_MIN, SG_MAX, SG_AVG, SG_COUNT, SG_COUNT_DISTINCT };

// one group of adjacent key columns whose images together fit 64 bits.  An integer
// column contributes (value - bias) in `bits` bits (bias = the column minimum, from
// key_ranges); a float column (bits == 0) its full-width total-order image.
struct SortGroup {
  int ncols;
  const void *data[8];
  int kind[8];
  int shift[8];
  int bits[8];
  long long bias[8];
};

// order-preserving unsigned image of one float element
__device__ __forceinline__ uint64_t ordered_float_bits(const void *data, int kind, int64_t i) {
  if (kind == K_F32) {
    uint32_t b = ((const uint32_t *)data)[i];
    if ((b << 1) == 0) b = 0;                                   // -0.0 == +0.0
    if ((b & 0x7fffffffu) > 0x7f800000u) return 0xffffffffu;    // NaN: after +inf
    return (b >> 31) ? (uint32_t)~b : (b | 0x80000000u);
  }
  uint64_t b = ((const uint64_t *)data)[i];
  if ((b << 1) == 0) b = 0;
  if ((b & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL) return ~0ULL;
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ long long load_signed_kind(const void *data, int kind, int64_t i) {
  switch (kind) {
    case K_I8: return ((const int8_t *)data)[i];
    case K_I16: return ((const int16_t *)data)[i];
    case K_I32: return ((const int32_t *)data)[i];
    default: return ((const int64_t *)data)[i];
  }
}
__device__ __forceinline__ uint64_t group_image(const SortGroup &g, int64_t row) {
  uint64_t k = 0;
  for (int c = 0; c < g.ncols; ++c) {
    const uint64_t f = g.bits[c] ? ((uint64_t)(load_signed_kind(g.data[c], g.kind[c], row) - g.bias[c]) &
                                    (g.bits[c] >= 64 ? ~0ULL : ((1ULL << g.bits[c]) - 1ULL)))
                                 : ordered_float_bits(g.data[c], g.kind[c], row);
    k |= f << g.shift[c];
  }
  return k;
}

// per-column minimum / maximum over the rows whose element is valid: out[2c], out[2c+1]
// window_rows != 0: a strided SAMPLE -- sample row j is row (j / window_rows) * window_step + j % window_rows, `sample_rows` of them
__global__ __launch_bounds__(256) void rs_minmax(KeyTable t, long long *out, int64_t sample_rows, int64_t window_rows, int64_t window_step) {
  const int64_t rows = window_rows ? sample_rows : t.nrows;
  for (int c = 0; c < t.ncols; ++c) {
    if (t.col[c].kind == K_F32 || t.col[c].kind == K_F64) continue;
    long long lo = 0x7fffffffffffffffLL, hi = (long long)0x8000000000000000ULL;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < rows; j += (int64_t)gridDim.x * 256) {
      const int64_t i = window_rows ? (j / window_rows) * window_step + j % window_rows : j;
      if (t.col[c].valid && !bit_is_set(t.col[c].valid, i)) continue;
      const long long v = load_signed_kind(t.col[c].data, t.col[c].kind, i);
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
    for (int d = 1; d < WAVE; d <<= 1) {
      const long long l2 = ((long long)__shfl_xor((int)(lo >> 32), d) << 32) | (unsigned int)__shfl_xor((int)lo, d);
      const long long h2 = ((long long)__shfl_xor((int)(hi >> 32), d) << 32) | (unsigned int)__shfl_xor((int)hi, d);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    if (lane_id() == 0) { atomicMin(&out[2 * c], lo); atomicMax(&out[2 * c + 1], hi); }
  }
}

// the common shape -- an integer column without a mask -- with 8 independent loads per thread in flight (the generic
// kernel above has one dependent load per trip: 3.9 TB/s on C5's int64 + int32 key columns, this one 5.4)
template <class T>
__global__ __launch_bounds__(256) void rs_minmax_fast(const T *__restrict__ key, int64_t n, long long *out) {
  long long lo = 0x7fffffffffffffffLL, hi = (long long)0x8000000000000000ULL;
  const int64_t stride = (int64_t)gridDim.x * 256 * 8;
  for (int64_t base = (int64_t)blockIdx.x * 256 * 8; base < n; base += stride) {
    T v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = base + k * 256 + threadIdx.x;
      v[k] = key[i < n ? i : n - 1];          // clamped: a repeated element changes neither min nor max
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) { const long long x = (long long)v[k]; lo = x < lo ? x : lo; hi = x > hi ? x : hi; }
  }
  for (int d = 1; d < WAVE; d <<= 1) {
    const long long l2 = ((long long)__shfl_xor((int)(lo >> 32), d) << 32) | (unsigned int)__shfl_xor((int)lo, d);
    const long long h2 = ((long long)__shfl_xor((int)(hi >> 32), d) << 32) | (unsigned int)__shfl_xor((int)hi, d);
    lo = l2 < lo ? l2 : lo;
    hi = h2 > hi ? h2 : hi;
  }
  if (lane_id() == 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}

// host: lo_hi[2c], lo_hi[2c+1] = min / max of integer column c over its valid elements
// (lo > hi: no valid element); float columns are left at (max, min)
// `windows` > 1: a SAMPLE -- that many evenly spaced windows of `window_rows` rows instead of the whole table: a strided sample
// sees a sorted or clustered key column's whole range where a prefix sees a sliver of it (gb_plan_range_sampled).  One launch.
gdf_error key_ranges(const KeyTable &t, long long *lo_hi, int windows, int64_t window_rows) {
  for (int c = 0; c < t.ncols; ++c) { lo_hi[2 * c] = 0x7fffffffffffffffLL; lo_hi[2 * c + 1] = (long long)0x8000000000000000ULL; }
  DevBuf mm;
  RMM_TRY(mm.alloc(sizeof(long long) * 2 * t.ncols));
  HIP_TRY(hipMemcpyAsync(mm.p, lo_hi, sizeof(long long) * 2 * t.ncols, hipMemcpyHostToDevice, stream0()));
  bool plain = t.nrows > 0;               // every column an unmasked integer (floats are skipped by both kernels)
  for (int c = 0; c < t.ncols; ++c) plain = plain && !t.col[c].valid;
  if (windows > 1 && window_rows > 0 && (int64_t)windows * window_rows < t.nrows) {
    const int64_t step = (t.nrows - window_rows) / (windows - 1);
    const int64_t sample = (int64_t)windows * window_rows;
    GDF_LAUNCH("rs_minmax", rs_minmax, dim3(stream_grid((size_t)sample, 256 * 4)), dim3(256), 0, stream0(), t, mm.as<long long>(), sample, window_rows, step);
  } else if (plain) {
    const dim3 grid(stream_grid((size_t)t.nrows, 256 * 8 * 4));
    for (int c = 0; c < t.ncols; ++c) {
      long long *o = mm.as<long long>() + 2 * c;
      switch (t.col[c].kind) {
        case K_I64: GDF_LAUNCH("rs_minmax", rs_minmax_fast<long long>, grid, dim3(256), 0, stream0(), (const long long *)t.col[c].data, t.nrows, o); break;
        case K_I32: GDF_LAUNCH("rs_minmax", rs_minmax_fast<int32_t>, grid, dim3(256), 0, stream0(), (const int32_t *)t.col[c].data, t.nrows, o); break;
        case K_I16: GDF_LAUNCH("rs_minmax", rs_minmax_fast<int16_t>, grid, dim3(256), 0, stream0(), (const int16_t *)t.col[c].data, t.nrows, o); break;
        case K_I8: GDF_LAUNCH("rs_minmax", rs_minmax_fast<int8_t>, grid, dim3(256), 0, stream0(), (const int8_t *)t.col[c].data, t.nrows, o); break;
        default: break;
      }
    }
  } else if (t.nrows > 0) {
    GDF_LAUNCH("rs_minmax", rs_minmax, dim3(stream_grid((size_t)t.nrows, 256 * 16)), dim3(256), 0, stream0(), t, mm.as<long long>(), (int64_t)0, (int64_t)0, (int64_t)0);
  }
  HIP_TRY(read_back(lo_hi, mm.p, sizeof(long long) * 2 * t.ncols));
  return GDF_SUCCESS;
}

// perm may alias vals (row numbers are rewritten in place).  *varying collects the bits
// on which some key differs from key 0: a digit window with no varying bit needs no pass.
__global__ __launch_bounds__(256) void rs_make_keys(SortGroup g, const uint32_t *perm, uint64_t *__restrict__ keys, uint32_t *vals,
                                                    uint32_t n, unsigned long long *__restrict__ varying) {
  const uint64_t k0 = group_image(g, perm ? perm[0] : 0);
  uint64_t diff = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t row = perm ? perm[i] : i;
    const uint64_t k = group_image(g, row);
    keys[i] = k;
    vals[i] = row;
    diff |= k ^ k0;
  }
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)diff, d), hi = __shfl_xor((uint32_t)(diff >> 32), d);
    diff |= ((uint64_t)hi << 32) | lo;
  }
  if (lane_id() == 0 && diff) atomicOr(varying, (unsigned long long)diff);
}

// ---------------------------------------------------------------------------
// stable LSD radix sort of (uint64 key, V payload) pairs; V = uint32 (row numbers)
// or uint64 (aggregation values riding with their keys)
// ---------------------------------------------------------------------------
constexpr int RS_THREADS = 512;
constexpr int RS_WAVES = RS_THREADS / WAVE;
template <class V> struct RsGeom { static constexpr int ITEMS = sizeof(V) == 4 ? 8 : 6; static constexpr int TILE = RS_THREADS * ITEMS; };

// counts[v * ntiles + tile] = number of keys of the tile whose digit is v
template <class K, int BITS, int TILE>
__global__ __launch_bounds__(RS_THREADS) void rs_count(const K *__restrict__ keys, uint32_t n, int shift,
                                                       uint32_t *__restrict__ counts, uint32_t ntiles) {
  constexpr int BINS = 1 << BITS;
  constexpr int ITEMS = TILE / RS_THREADS;
  __shared__ uint32_t h[BINS];
  // same tile order as rs_scatter: XCD x counts the x-th eighth of the tiles.  counts[v][*] is written 4 bytes per
  // tile; 32 consecutive tiles fill one of its 128-byte lines, and they meet in one L2 instead of eight
  const uint32_t tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  for (int j = threadIdx.x; j < BINS; j += RS_THREADS) h[j] = 0;
  block_sync();
  // counting needs no order: a thread takes ITEMS CONSECUTIVE keys with wide loads (a 4-byte key per load kept one
  // wave at 256 B in flight and this kernel at 1.6 TB/s on 32-bit keys)
  const uint32_t first = tile * TILE + threadIdx.x * ITEMS;
  if (first + ITEMS <= n) {
    K k[ITEMS];
    constexpr int WORDS = ITEMS * (int)sizeof(K) / 8;              // ITEMS * sizeof(K) is a multiple of 8 for both geometries
    const uint2 *src = reinterpret_cast<const uint2 *>(keys + first);
    uint2 w[WORDS];
#pragma unroll
    for (int q = 0; q < WORDS; ++q) w[q] = src[q];
    __builtin_memcpy(k, w, sizeof(k));
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
      // skewed digits (C5: half of the rows share one partition digit) would serialise 64 same-address LDS
      // atomics per wave: the lanes that agree with lane 0's digit are counted with one ballot instead
      const uint32_t d = (uint32_t)(k[r] >> shift) & (BINS - 1);
      const uint32_t d0 = __shfl(d, 0);
      const unsigned long long m = __ballot(d == d0);
      if (d != d0) atomicAdd(&h[d], 1u);
      else if (lane_id() == 0) atomicAdd(&h[d0], (uint32_t)__popcll(m));
    }
  } else {
    for (int r = 0; r < ITEMS; ++r)
      if (first + r < n) atomicAdd(&h[(uint32_t)(keys[first + r] >> shift) & (BINS - 1)], 1u);
  }
  block_sync();
  for (int j = threadIdx.x; j < BINS; j += RS_THREADS) counts[(uint32_t)j * ntiles + tile] = h[j];
}

template <class K, class V, int BITS>
__global__ __launch_bounds__(RS_THREADS) void rs_scatter(const K *__restrict__ keys_in, const V *__restrict__ vals_in,
                                                         K *__restrict__ keys_out, V *__restrict__ vals_out,
                                                         uint32_t n, int shift, const uint32_t *__restrict__ offsets,
                                                         uint32_t ntiles) {
  constexpr int BINS = 1 << BITS;
  constexpr int ITEMS = RsGeom<V>::ITEMS;
  constexpr int TILE = RsGeom<V>::TILE;
  static_assert(BINS <= RS_THREADS, "one thread per bin");
  __shared__ K skey[TILE];
  __shared__ V sval[TILE];
  __shared__ uint32_t wcnt[RS_WAVES * BINS];   // per-wave digit counts, then exclusive prefix over waves
  __shared__ uint32_t binstart[BINS];          // first LDS position of a digit
  __shared__ uint32_t gbase[BINS];             // global position of LDS position 0 of a digit (mod 2^32)
  __shared__ uint32_t wtot[RS_WAVES];
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  // XCD x (= blockIdx.x % 8, MI355X_MICROARCH.md "Workgroup dispatch") takes the x-th eighth of the tiles, in order.  A
  // tile leaves a few keys per digit (4096 keys over 512 digits), so a 128-byte line of the output is filled by several
  // CONSECUTIVE tiles; in dispatch order those run on different XCDs, whose L2s each write their part of the line back
  // separately.  From one XCD the parts merge in its L2 and the line reaches HBM once.  The grid is 8 * ceil(ntiles / 8).
  const uint32_t tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  for (int j = threadIdx.x; j < RS_WAVES * BINS; j += RS_THREADS) wcnt[j] = 0;
  block_sync();
  const uint32_t tile_base = tile * TILE;
  const uint32_t wbase = tile_base + wave * (ITEMS * WAVE);
  const uint32_t last = n - 1;
  K key[ITEMS];
  V val[ITEMS];
  uint32_t rank[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {       // clamped, unconditional: the loads of a wave stay in flight together
    const uint32_t i = wbase + r * WAVE + lane;
    const uint32_t j = i < n ? i : last;
    key[r] = keys_in[j];
    val[r] = vals_in[j];
  }
  const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const bool live = wbase + r * WAVE + lane < n;
    const uint32_t digit = (uint32_t)(key[r] >> shift) & (BINS - 1);
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int b = 0; b < BITS; ++b) {
      const bool bit = (digit >> b) & 1;
      const unsigned long long m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int leader = __ffsll((long long)peers) - 1;   // peers of a dead lane is junk; it is never used
    uint32_t old = 0;
    if (live && lane == leader) old = atomicAdd(&wcnt[wave * BINS + digit], (uint32_t)__popcll(peers));
    old = __shfl(old, live ? leader : lane);
    rank[r] = old + (uint32_t)__popcll(peers & lt);
  }
  block_sync();
  uint32_t total = 0;
  if (threadIdx.x < BINS) {
    for (int w = 0; w < RS_WAVES; ++w) {
      const uint32_t c = wcnt[w * BINS + threadIdx.x];
      wcnt[w * BINS + threadIdx.x] = total;
      total += c;
    }
    const uint32_t incl = wave_scan_incl(total);
    if (lane == WAVE - 1) wtot[wave] = incl;
    binstart[threadIdx.x] = incl - total;        // wave-local for now
  }
  block_sync();
  if (threadIdx.x < BINS) {
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wtot[w];
    const uint32_t start = binstart[threadIdx.x] + before;
    binstart[threadIdx.x] = start;
    gbase[threadIdx.x] = offsets[threadIdx.x * ntiles + tile] - start;
  }
  block_sync();
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (wbase + r * WAVE + lane < n) {
      const uint32_t digit = (uint32_t)(key[r] >> shift) & (BINS - 1);
      const uint32_t pos = binstart[digit] + wcnt[wave * BINS + digit] + rank[r];
      skey[pos] = key[r];
      sval[pos] = val[r];
    }
  }
  block_sync();
  const uint32_t count = n - tile_base < (uint32_t)TILE ? n - tile_base : (uint32_t)TILE;
  for (uint32_t j = threadIdx.x; j < count; j += RS_THREADS) {
    const K k = skey[j];
    const uint32_t dst = gbase[(uint32_t)(k >> shift) & (BINS - 1)] + j;
    keys_out[dst] = k;
    vals_out[dst] = sval[j];
  }
}

// Sorts n pairs by the key bits set in `varying` (bits on which all keys agree need no
// pass).  The pairs ping-pong between (kin, vin) and (kout, vout); on return kin / vin
// point at the sorted data.  The varying bits [lo, hi) are covered by ceil(span / 9)
// windows of at most 9 bits (8-bit digits for wide keys, 9 when that saves a pass:
// 25 bits sort in 3 passes); windows may overlap upward, which a stable LSD sort tolerates.
template <class K, class V>
gdf_error radix_sort_pairs(K *&kin, K *&kout, V *&vin, V *&vout, uint32_t n, uint64_t varying) {
  if (varying == 0 || n < 2) return GDF_SUCCESS;
  constexpr int TILE = RsGeom<V>::TILE;
  const uint32_t ntiles = (n + TILE - 1) / TILE;
  const uint32_t xcd_grid = (ntiles + 7) / 8 * 8;        // rs_count / rs_scatter: every XCD takes a contiguous eighth of the tiles
  const int lo = __builtin_ctzll(varying), hi = 64 - __builtin_clzll(varying);
  const int span = hi - lo;
  const int passes = (span + 8) / 9;
  const int bpp = (span + passes - 1) / passes;          // <= 9
  DevBuf counts;
  RMM_TRY(counts.alloc(sizeof(uint32_t) * (size_t)ntiles * 512));
  for (int p = 0; p < passes; ++p) {
    const int shift = lo + p * bpp;
    const int bits = bpp > 8 ? 9 : 8;
    if (((varying >> shift) & ((1ULL << bits) - 1)) == 0) continue;
    if (bits == 9) {
      GDF_LAUNCH("rs_count", (rs_count<K, 9, TILE>), dim3(xcd_grid), dim3(RS_THREADS), 0, stream0(), (const K *)kin, n, shift, counts.as<uint32_t>(), ntiles);
      GDF_TRY(scan_u32(counts.as<uint32_t>(), counts.as<uint32_t>(), (size_t)ntiles * 512, false));
      GDF_LAUNCH("rs_scatter", (rs_scatter<K, V, 9>), dim3(xcd_grid), dim3(RS_THREADS), 0, stream0(), (const K *)kin, (const V *)vin, kout, vout, n,
                 shift, (const uint32_t *)counts.as<uint32_t>(), ntiles);
    } else {
      GDF_LAUNCH("rs_count", (rs_count<K, 8, TILE>), dim3(xcd_grid), dim3(RS_THREADS), 0, stream0(), (const K *)kin, n, shift, counts.as<uint32_t>(), ntiles);
      GDF_TRY(scan_u32(counts.as<uint32_t>(), counts.as<uint32_t>(), (size_t)ntiles * 256, false));
      GDF_LAUNCH("rs_scatter", (rs_scatter<K, V, 8>), dim3(xcd_grid), dim3(RS_THREADS), 0, stream0(), (const K *)kin, (const V *)vin, kout, vout, n,
                 shift, (const uint32_t *)counts.as<uint32_t>(), ntiles);
    }
    std::swap(kin, kout);
    std::swap(vin, vout);
  }
  HIP_CHECK_LAST();
  return GDF_SUCCESS;
}
gdf_error radix_sort_pairs_u64(uint64_t *&kin, uint64_t *&kout, uint64_t *&vin, uint64_t *&vout, uint32_t n, uint64_t varying) {
  return radix_sort_pairs<uint64_t, uint64_t>(kin, kout, vin, vout, n, varying);
}
gdf_error radix_sort_pairs_k32_u64(uint32_t *&kin, uint32_t *&kout, uint64_t *&vin, uint64_t *&vout, uint32_t n, uint64_t varying) {
  return radix_sort_pairs<uint32_t, uint64_t>(kin, kout, vin, vout, n, varying);
}

// ---------------------------------------------------------------------------
// HYBRID sort of (uint64 image, uint32 row) pairs: the TOP t bits by stable LSD passes over the whole array (rs_count /
// rs_scatter: 32 bytes of traffic per pair and pass), everything below inside LDS.  After the top passes the pairs of one
// value of the top bits -- a BUCKET -- are contiguous and in input order; t is chosen so that a bucket holds ~770 pairs, so a
// wave sorts its bucket by the remaining bits in LDS (the same stable digit ranking as rs_scatter, 8 bits per pass) and the
// pairs cross HBM once more instead of once per digit: 62-bit keys cost 2 + 1 array passes instead of 7.  A workgroup takes
// four consecutive buckets, one per wave; when one of them is larger than a wave's 1024 slots the four waves sort the
// workgroup's buckets together, one after the other (up to 4096 pairs).  Bucket sizes depend on the data: a strided sample
// predicts the largest one and the LSD path is taken when it would not fit; a bucket that outgrows 4096 pairs against the
// prediction raises a flag and the LSD sort runs over all varying bits from where the top passes left the pairs (stable
// passes on a stably pre-sorted sequence: same result, two passes wasted).
// ---------------------------------------------------------------------------
constexpr int HS_ITEMS = 16;
constexpr int HS_WAVE_CAP = HS_ITEMS * WAVE;        // 1024 pairs per wave
constexpr int HS_BLOCK_CAP = 4 * HS_WAVE_CAP;       // 4096 pairs per workgroup
constexpr int HS_BIN_CAP = 48;                       // pairs of one top digit a single thread still sorts by insertion
constexpr int HS_WINDOW = 1024;                     // sample: contiguous windows of 1024 rows

__global__ __launch_bounds__(256) void hs_sample(const uint64_t *__restrict__ keys, uint32_t n, int shift, uint32_t tmask,
                                                 uint32_t *__restrict__ hist, uint32_t nwin) {
  const uint32_t begin = (uint32_t)((uint64_t)blockIdx.x * (uint64_t)(n - (n < HS_WINDOW ? n : HS_WINDOW)) / (nwin > 1 ? nwin - 1 : 1));
#pragma unroll
  for (int k = 0; k < HS_WINDOW / 256; ++k) {
    const uint32_t i = begin + k * 256 + threadIdx.x;
    if (i < n) atomicAdd(&hist[(uint32_t)(keys[i] >> shift) & tmask], 1u);
  }
}
__global__ __launch_bounds__(256) void hs_max(const uint32_t *__restrict__ hist, uint32_t bins, uint32_t *out) {
  uint32_t m = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < bins; i += gridDim.x * 256) m = hist[i] > m ? hist[i] : m;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const uint32_t x = __shfl_xor(m, o, WAVE); m = x > m ? x : m; }
  if (lane_id() == 0 && m) atomicMax(out, m);
}
// first / one-past-last position of every bucket of the top-sorted pairs (both 0 for an empty bucket: zero-initialised)
__global__ __launch_bounds__(256) void hs_bounds(const uint64_t *__restrict__ keys, uint32_t n, int shift, uint32_t tmask,
                                                 uint32_t *__restrict__ first, uint32_t *__restrict__ last) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t t = (uint32_t)(keys[i] >> shift) & tmask;
    const uint32_t before = i ? (uint32_t)(keys[i - 1] >> shift) & tmask : ~0u;
    if (t != before) {
      first[t] = i;
      if (i) last[before] = i;
    }
    if (i == n - 1) last[t] = n;
  }
}

// GW waves sort the m pairs at [begin, begin + m) by key bits [lo, hi) through their LDS area; every wave of the WORKGROUP
// runs this (the barriers are the workgroup's): GW == 1 -- each wave with its own bucket and area; GW == 4 -- one bucket
struct HsArgs {
  const uint64_t *kin;
  const uint32_t *vin;
  uint64_t *kout;            // nullptr: nobody reads the sorted images
  uint32_t *vout;
  size_t *vout64;            // non-null: the row numbers leave as size_t (gdf_order_by's d_indx) and vout stays unwritten
  int lo, hi;
  uint64_t varying;
  int dbg;
};
// GW == 1: the LDS area belongs to one wave, whose LDS operations execute in order -- waiting for them is all the
// synchronisation there is (no s_barrier: the four waves of the workgroup run their buckets independently)
template <int GW>
__device__ __forceinline__ void hs_sync() {
  if (GW == 1) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
  } else {
    block_sync();
  }
}
template <int GW>
__device__ __forceinline__ void hs_sort_bucket(uint64_t *skey, uint32_t *sval, uint32_t *cnt, uint32_t *wtot, const HsArgs &a,
                                               uint32_t begin_v, uint32_t m_v) {
  const uint32_t begin = __builtin_amdgcn_readfirstlane(begin_v), m = __builtin_amdgcn_readfirstlane(m_v);   // wave-uniform
  const int lane = lane_id();
  const int wave = GW == 1 ? 0 : (int)(threadIdx.x / WAVE);
  const uint32_t jbase = (uint32_t)wave * HS_WAVE_CAP;
  uint32_t *mycnt = cnt + wave * 256;
  constexpr int NB = GW == 1 ? 4 : 1;                        // digits a thread owns: 4 * lane .. + 3 of its wave / digit threadIdx.x
  uint64_t key[HS_ITEMS];
  uint32_t val[HS_ITEMS];
#pragma unroll
  for (int r = 0; r < HS_ITEMS; ++r) {
    const uint32_t j = jbase + r * WAVE + lane;
    const uint32_t i = j < m ? begin + j : begin;           // clamped, unconditional (begin < n also for an empty bucket)
    key[r] = a.kin[i];
    val[r] = a.vin[i];
  }
  // One stable counting pass by the digit at `shift`, through the LDS tile and back into the registers (item order).
  // top == true: the digit is the TOP one of the bits still to sort.  Then the pairs of one digit value -- a bin, ~3 pairs of
  // a 770-pair bucket -- only need sorting among themselves: when no bin holds more than HS_BIN_CAP pairs the thread that owns
  // a bin finishes it with an insertion sort in LDS (stable; whole keys compare like their low parts inside a bin) and the
  // bucket is done after ONE ranking pass instead of one per digit.  Returns whether that happened.
  uint32_t outpos[HS_ITEMS];                                 // where the pair in item slot r leaves to (its slot, unless a finished top pass says otherwise)
#pragma unroll
  for (int r = 0; r < HS_ITEMS; ++r) outpos[r] = jbase + r * WAVE + lane;
  auto pass = [&](int shift, uint32_t mask, bool top) -> bool {
    for (int i = lane; i < 256; i += WAVE) mycnt[i] = 0;
    if (GW != 1 && threadIdx.x == 0) wtot[4] = 0;
    uint32_t rank[HS_ITEMS];
#pragma unroll
    for (int r = 0; r < HS_ITEMS; ++r) {
      rank[r] = 0;
      if (jbase + r * WAVE < m) {                            // (uniform) a round without a live lane ranks nothing
        const bool live = jbase + r * WAVE + lane < m;
        rank[r] = wave_aggregated_inc(mycnt, (uint32_t)(key[r] >> shift) & mask, 8, live);
      }
    }
    hs_sync<GW>();
    uint32_t bstart[NB], bcount[NB];
    if (GW == 1) {            // exclusive starts of the wave's 256 digits: four per lane
      uint32_t sum = 0;
#pragma unroll
      for (int q = 0; q < NB; ++q) { bcount[q] = mycnt[NB * lane + q]; sum += bcount[q]; }
      uint32_t at = wave_scan_incl(sum) - sum;
#pragma unroll
      for (int q = 0; q < NB; ++q) { bstart[q] = at; mycnt[NB * lane + q] = at; at += bcount[q]; }
    } else {                  // thread d owns digit d: prefix over the waves, then over the digits
      const uint32_t d = threadIdx.x;
      uint32_t c[GW], total = 0;
#pragma unroll
      for (int w = 0; w < GW; ++w) { c[w] = cnt[w * 256 + d]; total += c[w]; }
      const uint32_t incl = wave_scan_incl(total);
      if (lane == WAVE - 1) wtot[threadIdx.x / WAVE] = incl;
      block_sync();
      uint32_t start = incl - total + waves_before_sum<256 / WAVE>(wtot, threadIdx.x);
      bstart[0] = start;
      bcount[0] = total;
#pragma unroll
      for (int w = 0; w < GW; ++w) { cnt[w * 256 + d] = start; start += c[w]; }
    }
    bool finish = false;
    if (top) {
      bool big = false;
#pragma unroll
      for (int q = 0; q < NB; ++q) big |= bcount[q] > (uint32_t)HS_BIN_CAP;
      if (GW == 1) finish = __ballot(big) == 0ULL;
      else if (big) wtot[4] = 1;
    }
    hs_sync<GW>();
    if (GW != 1 && top) finish = wtot[4] == 0;
#pragma unroll
    for (int r = 0; r < HS_ITEMS; ++r) {
      if (jbase + r * WAVE + lane < m) {
        const uint32_t pos = mycnt[(uint32_t)(key[r] >> shift) & mask] + rank[r];
        skey[pos] = key[r];
        sval[pos] = val[r];
      }
    }
    hs_sync<GW>();
#pragma unroll
    for (int r = 0; r < HS_ITEMS; ++r) {
      const uint32_t j = jbase + r * WAVE + lane;
      if (j < m) { key[r] = skey[j]; val[r] = sval[j]; }
    }
    if (finish && !(a.dbg & 2)) {
      // every pair ranks itself inside its bin: pairs of the bin with a smaller key, or the same key and an earlier place
      // (independent LDS reads, no chain of dependent ones: an insertion sort by the bin's owner ran 1.3 ms per 1e8 pairs,
      // this 0.2); the bin is [start of its digit, start of the next one) in the tile
      const uint32_t *start0 = cnt;                          // GW == 1: the wave's row; GW == 4: row 0 = the bin's start
#pragma unroll
      for (int r = 0; r < HS_ITEMS; ++r) {
        if (jbase + r * WAVE < m) {
          const uint32_t j = jbase + r * WAVE + lane;
          const bool live = j < m;
          const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
          const uint32_t bs = live ? start0[d] : 0u;
          const uint32_t be = live ? (d == mask ? m : start0[d + 1]) : 0u;
          uint32_t before = 0;
          for (uint32_t t = bs; __ballot(t < be) != 0ULL; ++t) {
            if (t < be) {
              const uint64_t other = skey[t];
              before += (other < key[r] || (other == key[r] && t < j)) ? 1u : 0u;
            }
          }
          outpos[r] = bs + before;
        }
      }
    }
    hs_sync<GW>();             // (the next pass's counters and tile are written only behind this)
    return finish;
  };
  bool sorted = (a.dbg & 1) != 0;
  if (!sorted && a.hi - a.lo > 8 && !(a.dbg & 4)) sorted = pass(a.hi - 8, 0xffu, true);
  if (!sorted) {               // LSD over every digit below the bucket's bits (from the order the attempt left: it was stable)
    for (int shift = a.lo; shift < a.hi; shift += 8) {
      const int bits = a.hi - shift < 8 ? a.hi - shift : 8;
      const uint32_t mask = (1u << bits) - 1u;
      if (((a.varying >> shift) & mask) == 0) continue;     // every key of the ARRAY agrees on this digit (uniform)
      if (GW == 1) {                                        // ... or every key of this bucket does (duplicates)
        const uint32_t d0 = (uint32_t)(__builtin_amdgcn_readfirstlane((uint32_t)(key[0] >> shift))) & mask;
        bool differs = false;
#pragma unroll
        for (int r = 0; r < HS_ITEMS; ++r) differs |= r * WAVE + lane < m && ((uint32_t)(key[r] >> shift) & mask) != d0;
        if (__ballot(differs) == 0ULL) continue;
      }
      pass(shift, mask, false);
    }
  }
#pragma unroll
  for (int r = 0; r < HS_ITEMS; ++r) {
    const uint32_t j = jbase + r * WAVE + lane;
    if (j < m) {
      if (a.kout) a.kout[begin + outpos[r]] = key[r];
      if (a.vout64) a.vout64[begin + outpos[r]] = val[r];
      else a.vout[begin + outpos[r]] = val[r];
    }
  }
}

__global__ __launch_bounds__(256) void hs_local(HsArgs a, const uint32_t *__restrict__ first, const uint32_t *__restrict__ last,
                                                uint32_t *oversize) {
  __shared__ uint64_t skey[HS_BLOCK_CAP];
  __shared__ uint32_t sval[HS_BLOCK_CAP];
  __shared__ uint32_t cnt[4 * 256];
  __shared__ uint32_t wtot[8];              // [4]: "a bin is too large" flag of the workgroup-wide pass
  const uint32_t b0 = blockIdx.x * 4;
  uint32_t bg[4], mm[4];
  bool big = false;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    bg[q] = first[b0 + q];
    mm[q] = last[b0 + q] - bg[q];
    big |= mm[q] > (uint32_t)HS_WAVE_CAP;
  }
  if (!big) {
    const int wave = threadIdx.x / WAVE;
    const uint32_t mine = wave == 0 ? 0 : wave == 1 ? 1 : wave == 2 ? 2 : 3;
    uint32_t begin = bg[0], m = mm[0];
#pragma unroll
    for (int q = 1; q < 4; ++q) if (mine == (uint32_t)q) { begin = bg[q]; m = mm[q]; }
    if (m) hs_sort_bucket<1>(skey + wave * HS_WAVE_CAP, sval + wave * HS_WAVE_CAP, cnt + wave * 256, wtot, a, begin, m);
    return;
  }
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    if (mm[q] == 0) continue;
    if (mm[q] > (uint32_t)HS_BLOCK_CAP) {            // against the sample's prediction: the caller falls back
      if (threadIdx.x == 0) atomicExch(oversize, 1u);
      continue;
    }
    hs_sort_bucket<4>(skey, sval, cnt, wtot, a, bg[q], mm[q]);
    block_sync();
  }
}

// *done = false: the shape is not one for this path (nothing was touched) -- the caller runs radix_sort_pairs
static gdf_error hybrid_sort_pairs(uint64_t *&kin, uint64_t *&kout, uint32_t *&vin, uint32_t *&vout, uint32_t n, uint64_t varying,
                                   bool want_keys, size_t *perm64, bool *perm64_written, bool *done) {
  *done = false;
  if (lab::path_on("GDF_SORT_NO_HYBRID") || varying == 0) return GDF_SUCCESS;
  const uint32_t min_rows = (uint32_t)lab::path_int("GDF_HS_MIN_ROWS", 1 << 21);
  if (n < min_rows || n < 2 * HS_WINDOW) return GDF_SUCCESS;
  const int lo = __builtin_ctzll(varying), hi = 64 - __builtin_clzll(varying);
  const int span = hi - lo;
  int t = 0;
  while (t < 18 && ((uint64_t)768 << t) < (uint64_t)n) ++t;          // ~384 .. 768 pairs per bucket
  if (t < 4) t = 4;
  if (((uint64_t)HS_WAVE_CAP << t) < (uint64_t)n) return GDF_SUCCESS; // beyond 2^28 rows the buckets outgrow a wave on average
  const int lsd_passes = (span + 8) / 9, top_passes = (t + 8) / 9;
  if (span <= t || lsd_passes < top_passes + 2) return GDF_SUCCESS;   // too few bits below the top ones to pay for the extra pass
  const int shift = hi - t;
  const uint32_t tmask = (1u << t) - 1u, nb = 1u << t;
  const uint64_t top_bits = (uint64_t)tmask << shift;
  DevBuf tab, flags;
  RMM_TRY(tab.alloc(sizeof(uint32_t) * 2 * (size_t)nb));
  RMM_TRY(flags.alloc(sizeof(uint32_t) * 2));
  uint32_t *first = tab.as<uint32_t>(), *last = first + nb;
  // the largest bucket, predicted from <= 2^21 rows in 1024-row windows spread over the input
  const uint32_t nwin = std::min<uint32_t>(2048, n / HS_WINDOW);
  HIP_TRY(hipMemsetAsync(first, 0, sizeof(uint32_t) * nb, stream0()));
  HIP_TRY(hipMemsetAsync(flags.p, 0, sizeof(uint32_t) * 2, stream0()));
  GDF_LAUNCH("hs_sample", hs_sample, dim3(nwin), dim3(256), 0, stream0(), (const uint64_t *)kin, n, shift, tmask, first, nwin);
  GDF_LAUNCH("hs_max", hs_max, dim3(std::min<uint32_t>(256, (nb + 255) / 256)), dim3(256), 0, stream0(), (const uint32_t *)first, nb,
             flags.as<uint32_t>());
  uint32_t sample_max = 0;
  HIP_TRY(hipMemcpyAsync(&sample_max, flags.p, sizeof(uint32_t), hipMemcpyDeviceToHost, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  const double scale = (double)n / ((double)nwin * HS_WINDOW);
  if ((double)sample_max * scale > 0.75 * HS_BLOCK_CAP) return GDF_SUCCESS;
  // stable LSD passes over the top bits only, then the buckets' bounds
  GDF_TRY((radix_sort_pairs<uint64_t, uint32_t>(kin, kout, vin, vout, n, varying & top_bits)));
  HIP_TRY(hipMemsetAsync(first, 0, sizeof(uint32_t) * 2 * (size_t)nb, stream0()));
  GDF_LAUNCH("hs_bounds", hs_bounds, dim3(stream_grid(n, 256 * 8)), dim3(256), 0, stream0(), (const uint64_t *)kin, n, shift, tmask, first, last);
  HsArgs a{kin, vin, want_keys ? kout : nullptr, vout, perm64, lo, shift, varying, (int)lab::path_int("GDF_HS_DBG", 0)};   // (the sorted images leave only for a caller that reads them)
  GDF_LAUNCH("hs_local", hs_local, dim3(nb / 4), dim3(256), 0, stream0(), a, (const uint32_t *)first, (const uint32_t *)last,
             flags.as<uint32_t>() + 1);
  uint32_t oversize = 0;
  HIP_TRY(hipMemcpyAsync(&oversize, flags.as<uint32_t>() + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  if (oversize) {            // the pairs in (kin, vin) are still the top-sorted ones: sort them by all varying bits
    GDF_TRY((radix_sort_pairs<uint64_t, uint32_t>(kin, kout, vin, vout, n, varying)));
  } else {
    std::swap(kin, kout);
    std::swap(vin, vout);
    if (perm64 && perm64_written) *perm64_written = true;
  }
  *done = true;
  return GDF_SUCCESS;
}

__global__ __launch_bounds__(256) void rs_iota(uint32_t *p, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = i;
}
__global__ __launch_bounds__(256) void rs_widen(const uint32_t *__restrict__ in, size_t *__restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) out[i] = in[i];
}

static int bit_length(uint64_t v) { return v ? 64 - __builtin_clzll(v) : 0; }

// Sorted row permutation of table t (ascending, lexicographic, stable).  On return
// perm holds n uint32 row numbers; if sorted_keys is non-null and the whole key fitted
// one integer-only image, *sorted_keys keeps the sorted images (adjacent-equal test
// without gathers) and *keys_exact is set.
gdf_error order_rows(const KeyTable &t, uint32_t n, DevBuf &perm, DevBuf *sorted_keys, bool *keys_exact, size_t *perm64, bool *perm64_written) {
  if (keys_exact) *keys_exact = false;
  if (perm64_written) *perm64_written = false;
  // image width of every column: integers (value - min) in bit_length(max - min) bits, floats full width
  std::vector<long long> lo_hi(2 * t.ncols);
  bool any_float = false, any_int = false;
  for (int c = 0; c < t.ncols; ++c) {
    const bool f = (t.col[c].kind == K_F32 || t.col[c].kind == K_F64);
    any_float |= f;
    any_int |= !f;
  }
  KeyTable tn = t;
  tn.nrows = n;
  if (any_int) GDF_TRY(key_ranges(tn, lo_hi.data()));
  std::vector<int> width(t.ncols);
  for (int c = 0; c < t.ncols; ++c) {
    if (t.col[c].kind == K_F32 || t.col[c].kind == K_F64) { width[c] = t.col[c].width * 8; continue; }
    if (lo_hi[2 * c] > lo_hi[2 * c + 1]) lo_hi[2 * c] = lo_hi[2 * c + 1] = 0;
    width[c] = bit_length((uint64_t)lo_hi[2 * c + 1] - (uint64_t)lo_hi[2 * c]);
    if (width[c] == 0) width[c] = 1;                       // a constant column still owns one (never varying) bit
  }
  // column groups of <= 64 image bits, last columns first (LSD over groups)
  std::vector<SortGroup> groups;
  {
    int c = t.ncols - 1;
    while (c >= 0) {
      SortGroup g{};
      int bits = 0, first = c;
      while (first >= 0 && bits + width[first] <= 64 && c - first < 8) { bits += width[first]; --first; }
      ++first;
      int shift = bits;
      for (int k = first; k <= c; ++k) {
        shift -= width[k];
        const bool f = (t.col[k].kind == K_F32 || t.col[k].kind == K_F64);
        g.data[g.ncols] = t.col[k].data;
        g.kind[g.ncols] = t.col[k].kind;
        g.shift[g.ncols] = shift;
        g.bits[g.ncols] = f ? 0 : width[k];
        g.bias[g.ncols] = f ? 0 : lo_hi[2 * k];
        ++g.ncols;
      }
      groups.push_back(g);
      c = first - 1;
    }
  }

  DevBuf ka, kb, va, vb, vary;
  RMM_TRY(ka.alloc(sizeof(uint64_t) * (size_t)n));
  RMM_TRY(kb.alloc(sizeof(uint64_t) * (size_t)n));
  RMM_TRY(va.alloc(sizeof(uint32_t) * (size_t)n));
  RMM_TRY(vb.alloc(sizeof(uint32_t) * (size_t)n));
  RMM_TRY(vary.alloc(sizeof(unsigned long long)));
  uint64_t *kin = ka.as<uint64_t>(), *kout = kb.as<uint64_t>();
  uint32_t *vin = va.as<uint32_t>(), *vout = vb.as<uint32_t>();
  const int sgrid = stream_grid(n, 256 * 8);
  bool have_perm = false;
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    // after the first group the current permutation lives in vin; keys are rebuilt in kin
    // reading it, and the row numbers are rewritten in place (vals[i] = perm[i])
    HIP_TRY(hipMemsetAsync(vary.p, 0, sizeof(unsigned long long), stream0()));
    GDF_LAUNCH("rs_make_keys", rs_make_keys, dim3(sgrid), dim3(256), 0, stream0(), groups[gi],
               have_perm ? (const uint32_t *)vin : (const uint32_t *)nullptr, kin, vin, n, vary.as<unsigned long long>());
    have_perm = true;
    unsigned long long varying = 0;
    HIP_TRY(hipMemcpyAsync(&varying, vary.p, sizeof(varying), hipMemcpyDeviceToHost, stream0()));
    HIP_TRY(hipStreamSynchronize(stream0()));
    bool hybrid = false;
    // (the last group's sort may write the caller's size_t permutation itself: perm then holds nothing)
    GDF_TRY(hybrid_sort_pairs(kin, kout, vin, vout, n, varying, sorted_keys && groups.size() == 1 && !any_float,
                              gi + 1 == groups.size() ? perm64 : nullptr, perm64_written, &hybrid));
    if (!hybrid) GDF_TRY((radix_sort_pairs<uint64_t, uint32_t>(kin, kout, vin, vout, n, varying)));
  }
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  // hand the buffers that ended up holding the result to the caller
  if (vin == va.as<uint32_t>()) perm.p = va.release(); else perm.p = vb.release();
  if (sorted_keys && groups.size() == 1 && !any_float) {
    if (kin == ka.as<uint64_t>()) sorted_keys->p = ka.release(); else sorted_keys->p = kb.release();
    *keys_exact = true;
  }
  return GDF_SUCCESS;
}

// ---------------------------------------------------------------------------
// segmented reduction over the sorted rows
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sg_heads(KeyTable t, const uint32_t *__restrict__ perm, const uint64_t *__restrict__ sorted_keys,
                                                uint32_t *__restrict__ head, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    uint32_t hd = 1;
    if (i > 0) {
      if (sorted_keys) hd = sorted_keys[i] != sorted_keys[i - 1];
      else hd = !rows_equal(t, perm[i - 1], t, perm[i]);
    }
    head[i] = hd;
  }
}
// gid = inclusive scan of head (1-based); start[g] = first sorted position of group g
__global__ __launch_bounds__(256) void sg_starts(const uint32_t *__restrict__ gid, uint32_t *__restrict__ start, uint32_t n, uint32_t ngroups) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (i == 0 || gid[i] != gid[i - 1]) start[gid[i] - 1] = i;
  if (blockIdx.x == 0 && threadIdx.x == 0) start[ngroups] = n;
}

__device__ __forceinline__ uint64_t sg_ord_i64(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ULL; }
__device__ __forceinline__ uint64_t sg_ord_f64(double d) {
  const uint64_t b = (uint64_t)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double sg_unord_f64(uint64_t u) {
  const uint64_t b = (u >> 63) ? (u & 0x7fffffffffffffffULL) : ~u;
  return __longlong_as_double((long long)b);
}
__device__ __forceinline__ bool sg_is_flt(int kind) { return kind == K_F32 || kind == K_F64; }

// 64-bit image of one value: SUM/AVG integers as wrapped uint64, floats as double;
// MIN/MAX as an order-preserving unsigned image
__device__ __forceinline__ uint64_t sg_image(int op, const void *data, int kind, int64_t i) {
  if (sg_is_flt(kind)) {
    const double d = kind == K_F32 ? (double)((const float *)data)[i] : ((const double *)data)[i];
    return (op == SG_MIN || op == SG_MAX) ? sg_ord_f64(d) : (uint64_t)__double_as_longlong(d);
  }
  int64_t v;
  switch (kind) {
    case K_I8: v = ((const int8_t *)data)[i]; break;
    case K_I16: v = ((const int16_t *)data)[i]; break;
    case K_I32: v = ((const int32_t *)data)[i]; break;
    default: v = ((const int64_t *)data)[i]; break;
  }
  return (op == SG_MIN || op == SG_MAX) ? sg_ord_i64(v) : (uint64_t)v;
}
__device__ __forceinline__ uint64_t sg_fold(int op, bool flt, uint64_t a, uint64_t b) {
  if (op == SG_MIN) return a < b ? a : b;
  if (op == SG_MAX) return a > b ? a : b;
  if (flt) return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
  return a + b;
}
__device__ __forceinline__ void sg_flush(int op, bool flt, unsigned long long *acc, uint64_t v) {
  if (op == SG_MIN) atomicMin(acc, (unsigned long long)v);
  else if (op == SG_MAX) atomicMax(acc, (unsigned long long)v);
  else if (flt) atomicAdd((double *)acc, __longlong_as_double((long long)v));
  else atomicAdd(acc, (unsigned long long)v);
}
__device__ __forceinline__ uint64_t shfl_up64(uint64_t v, int d) {
  const uint32_t lo = __shfl_up((uint32_t)v, d), hi = __shfl_up((uint32_t)(v >> 32), d);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t readlane64(uint64_t v, int l) {
  const uint32_t lo = __shfl((uint32_t)v, l), hi = __shfl((uint32_t)(v >> 32), l);
  return ((uint64_t)hi << 32) | lo;
}

__global__ __launch_bounds__(256) void sg_fill(unsigned long long *p, unsigned long long v, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = v;
}

// acc[g] op= value of every sorted row of group g.  A wave walks 64 x SG_ROUNDS
// consecutive sorted rows; per round a segmented inclusive shuffle scan on the group
// id folds equal-id lanes, closed segments leave with one atomic, and the segment that
// is still open at lane 63 rides along in (carry_gid, carry) into the next round.
__global__ __launch_bounds__(SG_THREADS) void sg_reduce(const uint32_t *__restrict__ perm, const uint32_t *__restrict__ gid,
                                                        const void *__restrict__ val, int kind, int op,
                                                        unsigned long long *__restrict__ acc, uint32_t n) {
  const bool flt = sg_is_flt(kind);
  const int lane = lane_id();
  const uint64_t wave_global = (uint64_t)blockIdx.x * (SG_THREADS / WAVE) + threadIdx.x / WAVE;
  const uint64_t begin = wave_global * (uint64_t)(WAVE * SG_ROUNDS);
  if (begin >= n) return;
  uint32_t carry_gid = 0;      // 0 = no open segment (ids are 1-based)
  uint64_t carry = 0;
  for (int r = 0; r < SG_ROUNDS; ++r) {
    const uint64_t i = begin + (uint64_t)r * WAVE + lane;
    if (begin + (uint64_t)r * WAVE >= n) break;                 // wave-uniform
    const bool live = i < n;
    const uint32_t j = live ? (uint32_t)i : n - 1;
    const uint32_t g = live ? gid[j] : 0xffffffffu;             // dead lanes form their own trailing segment
    uint64_t v = sg_image(op, val, kind, perm[j]);
    // segmented inclusive scan: fold lane-d's value when it belongs to the same group
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint64_t up = shfl_up64(v, d);
      const uint32_t ug = __shfl_up(g, d);
      if (lane >= d && ug == g) v = sg_fold(op, flt, v, up);
    }
    const uint32_t gnext = __shfl_down(g, 1);
    const bool tail = (lane == WAVE - 1) || (gnext != g);       // last lane of its segment in this round
    const uint32_t g0 = __shfl(g, 0);
    // the carried segment either continues into lane 0's group or is finished
    if (carry_gid != 0 && carry_gid != g0) {
      if (lane == 0) sg_flush(op, flt, &acc[carry_gid - 1], carry);
      carry_gid = 0;
    }
    if (tail && carry_gid != 0 && g == carry_gid) v = sg_fold(op, flt, v, carry);
    const uint32_t glast = __shfl(g, WAVE - 1);
    const uint64_t vlast = readlane64(v, WAVE - 1);
    if (tail && live && lane != WAVE - 1) sg_flush(op, flt, &acc[g - 1], v);
    if (glast != 0xffffffffu) { carry_gid = glast; carry = vlast; }
    else carry_gid = 0;
  }
  if (carry_gid != 0 && lane == 0) sg_flush(op, flt, &acc[carry_gid - 1], carry);
}

struct SgOut {
  void *agg;            // aggregation output
  int agg_kind;         // dtype it is written in
  size_t *indices;      // out_col_indices->data or null
};

template <class T>
__device__ __forceinline__ void sg_store(void *out, uint32_t g, int op, bool flt, uint64_t a, uint32_t count) {
  T r;
  if (op == SG_COUNT || op == SG_COUNT_DISTINCT) r = (T)count;
  else if (op == SG_MIN || op == SG_MAX) r = flt ? (T)sg_unord_f64(a) : (T)(int64_t)(a ^ 0x8000000000000000ULL);
  else {
    const T s = flt ? (T)__longlong_as_double((long long)a) : (T)(int64_t)a;
    if (op == SG_AVG) { const T c = (T)count; r = (c != (T)0) ? (T)(s / c) : (T)0; }   // sum/static_cast<ValsT>(n)
    else r = s;
  }
  ((T *)out)[g] = r;
}

__global__ __launch_bounds__(256) void sg_finalize(const unsigned long long *__restrict__ acc, const uint32_t *__restrict__ start,
                                                   const uint32_t *__restrict__ perm, uint32_t ngroups, int op, int in_kind, SgOut o) {
  for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
    const uint32_t s = start[g], e = start[g + 1];
    if (o.indices) o.indices[g] = (size_t)perm[e - 1];
    const uint64_t a = acc ? acc[g] : 0;
    const bool flt = sg_is_flt(in_kind);
    switch (o.agg_kind) {
      case K_I8: sg_store<int8_t>(o.agg, g, op, flt, a, e - s); break;
      case K_I16: sg_store<int16_t>(o.agg, g, op, flt, a, e - s); break;
      case K_I32: sg_store<int32_t>(o.agg, g, op, flt, a, e - s); break;
      case K_I64: sg_store<int64_t>(o.agg, g, op, flt, a, e - s); break;
      case K_F32: sg_store<float>(o.agg, g, op, flt, a, e - s); break;
      default: sg_store<double>(o.agg, g, op, flt, a, e - s); break;
    }
  }
}

// out[g] = in[perm[start[g + 1] - 1]]: the multi_gather_host of sqls_ops.cu:232-252
__global__ __launch_bounds__(256) void sg_gather(const uint32_t *__restrict__ start, const uint32_t *__restrict__ perm, uint32_t ngroups,
                                                 int width, const void *__restrict__ in, void *__restrict__ out) {
  for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
    const uint32_t s = perm[start[g + 1] - 1];
    switch (width) {
      case 1: ((uint8_t *)out)[g] = ((const uint8_t *)in)[s]; break;
      case 2: ((uint16_t *)out)[g] = ((const uint16_t *)in)[s]; break;
      case 4: ((uint32_t *)out)[g] = ((const uint32_t *)in)[s]; break;
      default: ((uint64_t *)out)[g] = ((const uint64_t *)in)[s]; break;
    }
  }
}
__global__ void sg_store_count(void *out, int kind, uint32_t v) {
  switch (kind) {
    case K_I8: *(int8_t *)out = (int8_t)v; break;
    case K_I16: *(int16_t *)out = (int16_t)v; break;
    case K_I32: *(int32_t *)out = (int32_t)v; break;
    case K_I64: *(int64_t *)out = (int64_t)v; break;
    case K_F32: *(float *)out = (float)v; break;
    default: *(double *)out = (double)v; break;
  }
}

// sqls_ops.cu:1134-1289.  Inputs were validated by group_by_single (groupby.hip).
gdf_error group_by_sort(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                        gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt, int op) {
  KeyTable t;
  GDF_TRY(make_key_table(cols, ncols, &t));
  for (int c = 0; c < ncols; ++c) GDF_REQUIRE(cols[c]->size == cols[0]->size, GDF_COLUMN_SIZE_MISMATCH);
  GDF_REQUIRE(cols[0]->size < (size_t)0x7fffffff, GDF_COLUMN_SIZE_TOO_BIG);
  const uint32_t n = (uint32_t)cols[0]->size;
  const bool counting = (op == SG_COUNT || op == SG_COUNT_DISTINCT);
  const ElemKind in_kind = elem_kind(col_agg->dtype);
  // SUM/MIN/MAX/AVG dispatch on the input dtype and write that dtype (sqls_ops.cu:411-1083);
  // COUNT dispatches on the output column's dtype (:272-400)
  const ElemKind out_kind = counting ? elem_kind(out_col_agg->dtype) : in_kind;
  if (counting) GDF_REQUIRE(out_kind != K_BAD && out_col_agg->dtype <= GDF_FLOAT64, GDF_UNSUPPORTED_DTYPE);
  else {
    GDF_REQUIRE(in_kind != K_BAD && col_agg->dtype <= GDF_FLOAT64, GDF_UNSUPPORTED_DTYPE);
    GDF_REQUIRE(col_agg->size == cols[0]->size, GDF_COLUMN_SIZE_MISMATCH);
  }
  GDF_REQUIRE(out_col_agg->data != nullptr, GDF_DATASET_EMPTY);

  DevBuf perm, sorted_keys;
  bool keys_exact = false;
  if (ctxt->flag_sorted) {
    RMM_TRY(perm.alloc(sizeof(uint32_t) * (size_t)n));
    GDF_LAUNCH("rs_iota", rs_iota, dim3(stream_grid(n, 1024)), dim3(256), 0, stream0(), perm.as<uint32_t>(), n);
  } else {
    GDF_TRY(order_rows(t, n, perm, &sorted_keys, &keys_exact));
  }
  DevBuf gid, start, acc;
  RMM_TRY(gid.alloc(sizeof(uint32_t) * (size_t)n));
  const int grid = stream_grid(n, 256 * 4);
  GDF_LAUNCH("sg_heads", sg_heads, dim3(grid), dim3(256), 0, stream0(), t, perm.as<uint32_t>(),
             keys_exact ? sorted_keys.as<uint64_t>() : (const uint64_t *)nullptr, gid.as<uint32_t>(), n);
  GDF_TRY(scan_u32(gid.as<uint32_t>(), gid.as<uint32_t>(), n, true));
  uint32_t ngroups = 0;
  HIP_TRY(hipMemcpyAsync(&ngroups, gid.as<uint32_t>() + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  sorted_keys.reset();
  RMM_TRY(start.alloc(sizeof(uint32_t) * ((size_t)ngroups + 1)));
  GDF_LAUNCH("sg_starts", sg_starts, dim3(grid), dim3(256), 0, stream0(), gid.as<uint32_t>(), start.as<uint32_t>(), n, ngroups);
  if (!counting) {
    RMM_TRY(acc.alloc(sizeof(unsigned long long) * (size_t)ngroups));
    GDF_LAUNCH("sg_fill", sg_fill, dim3(stream_grid(ngroups, 1024)), dim3(256), 0, stream0(), acc.as<unsigned long long>(),
               op == SG_MIN ? ~0ULL : 0ULL, ngroups);
    const uint32_t per_block = WAVE * SG_ROUNDS * (SG_THREADS / WAVE);
    GDF_LAUNCH("sg_reduce", sg_reduce, dim3((n + per_block - 1) / per_block), dim3(SG_THREADS), 0, stream0(), perm.as<uint32_t>(),
               gid.as<uint32_t>(), (const void *)col_agg->data, (int)in_kind, op, acc.as<unsigned long long>(), n);
  }
  SgOut o{out_col_agg->data, (int)out_kind, out_col_indices ? (size_t *)out_col_indices->data : nullptr};
  const int ggrid = stream_grid(ngroups, 256);
  GDF_LAUNCH("sg_finalize", sg_finalize, dim3(ggrid), dim3(256), 0, stream0(),
             counting ? (const unsigned long long *)nullptr : acc.as<unsigned long long>(), start.as<uint32_t>(),
             perm.as<uint32_t>(), ngroups, op, (int)in_kind, o);
  size_t reported = ngroups;
  if (op == SG_COUNT_DISTINCT) {
    hipLaunchKernelGGL(sg_store_count, dim3(1), dim3(1), 0, stream0(), out_col_agg->data, (int)out_kind, ngroups);
    reported = 1;
  }
  if (out_col_values)
    for (int c = 0; c < ncols; ++c) {
      if (!out_col_values[c] || !out_col_values[c]->data) continue;
      GDF_LAUNCH("sg_gather", sg_gather, dim3(ggrid), dim3(256), 0, stream0(), start.as<uint32_t>(), perm.as<uint32_t>(),
                 (uint32_t)reported, t.col[c].width, t.col[c].data, out_col_values[c]->data);
      out_col_values[c]->size = reported;
    }
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  out_col_agg->size = reported;
  if (out_col_indices) out_col_indices->size = reported;
  return GDF_SUCCESS;
}

// ---------------------------------------------------------------------------
// gdf_radixsort_* / gdf_segmented_radixsort_* (reference src/sorting.cu, src/segmented_sorting.cu: thin wrappers
// over cub::DeviceRadixSort / DeviceSegmentedRadixSort with a plan object that owns the back buffers).  Here the
// plan only records its parameters: the keys' order-preserving images (complemented for a descending sort) and
// their row numbers go through the LSD radix sort above, then keys and values are gathered through the
// permutation.  Stable like CUB's (equal keys keep their input order, ascending and descending alike);
// begin_bit / end_bit restrict the sort to those bits of the image.  -0.0 and +0.0 compare equal and NaN orders
// after +inf (numpy's order; CUB orders by the raw bit pattern there).
// ---------------------------------------------------------------------------
struct RadixPlan {
  size_t num_items;
  int descending;
  unsigned begin_bit, end_bit;
  size_t sizeof_key, sizeof_val;
};

__device__ __forceinline__ uint64_t rsw_image(const void *key, int kind, int descending, const uint32_t *region,
                                              const uint8_t *region_sorted, uint32_t i) {
  const int bits = (kind == K_I8 ? 8 : (kind == K_I16 ? 16 : ((kind == K_I32 || kind == K_F32) ? 32 : 64)));
  const uint64_t mask = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL);
  uint64_t k;
  if (kind == K_F32 || kind == K_F64) k = ordered_float_bits(key, kind, i);
  else k = ((uint64_t)load_signed_kind(key, kind, i) ^ (1ULL << (bits - 1))) & mask;
  if (descending) k = ~k & mask;
  if (region && !region_sorted[region[i]]) k = 0;          // rows outside every segment keep their order
  return k;
}
__global__ __launch_bounds__(256) void rsw_images(const void *key, int kind, int descending, const uint32_t *__restrict__ region,
                                                  const uint8_t *__restrict__ region_sorted, uint64_t *__restrict__ img,
                                                  uint32_t *__restrict__ row, uint32_t n, unsigned long long *__restrict__ varying) {
  const uint64_t k0 = rsw_image(key, kind, descending, region, region_sorted, 0);
  uint64_t diff = 0;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint64_t k = rsw_image(key, kind, descending, region, region_sorted, i);
    img[i] = k;
    row[i] = i;
    diff |= k ^ k0;
  }
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)diff, d), hi = __shfl_xor((uint32_t)(diff >> 32), d);
    diff |= ((uint64_t)hi << 32) | lo;
  }
  if (lane_id() == 0 && diff) atomicOr(varying, (unsigned long long)diff);
}
__global__ __launch_bounds__(256) void rsw_region_keys(const uint32_t *__restrict__ region, const uint32_t *__restrict__ perm,
                                                       uint64_t *__restrict__ img, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) img[i] = region[perm[i]];
}
// region[i] = number of boundary points <= i (binary search over the sorted boundary list)
__global__ __launch_bounds__(256) void rsw_regions(const uint32_t *__restrict__ bounds, uint32_t nbounds, uint32_t *__restrict__ region, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    uint32_t lo = 0, hi = nbounds;
    while (lo < hi) { const uint32_t mid = (lo + hi) / 2; if (bounds[mid] <= i) lo = mid + 1; else hi = mid; }
    region[i] = lo;
  }
}
__global__ __launch_bounds__(256) void rsw_gather(const uint32_t *__restrict__ perm, uint32_t n, int width, const void *in, void *out) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const uint32_t s = perm[i];
    switch (width) {
      case 1: ((uint8_t *)out)[i] = ((const uint8_t *)in)[s]; break;
      case 2: ((uint16_t *)out)[i] = ((const uint16_t *)in)[s]; break;
      case 4: ((uint32_t *)out)[i] = ((const uint32_t *)in)[s]; break;
      default: ((uint64_t *)out)[i] = ((const uint64_t *)in)[s]; break;
    }
  }
}

// keycol (and valcol, int64) are sorted IN PLACE.  nseg < 0: whole column; otherwise only inside the nseg segments
// [begin[s], end[s]) given as DEVICE arrays of uint32.
static gdf_error radixsort_api(const RadixPlan *plan, gdf_column *keycol, gdf_column *valcol, gdf_dtype expect, int nseg,
                               const unsigned *d_begin, const unsigned *d_end) {
  GDF_REQUIRE(plan && keycol && valcol, GDF_DATASET_EMPTY);
  GDF_REQUIRE(!keycol->valid, GDF_VALIDITY_UNSUPPORTED);                    // sorting.cu:196-197
  GDF_REQUIRE(!valcol->valid, GDF_VALIDITY_UNSUPPORTED);
  GDF_REQUIRE(keycol->size == valcol->size, GDF_COLUMN_SIZE_MISMATCH);      // :199
  GDF_REQUIRE(plan->num_items == keycol->size, GDF_COLUMN_SIZE_MISMATCH);   // :202
  const int kw = dtype_width(expect);
  GDF_REQUIRE((size_t)kw == plan->sizeof_key && plan->sizeof_val == sizeof(int64_t), GDF_COLUMN_SIZE_MISMATCH);   // :204-207
  GDF_REQUIRE(plan->num_items < (size_t)0x7fffffff, GDF_COLUMN_SIZE_TOO_BIG);
  const uint32_t n = (uint32_t)plan->num_items;
  if (n < 2) return GDF_SUCCESS;
  GDF_REQUIRE(keycol->data && valcol->data, GDF_DATASET_EMPTY);
  const ElemKind kind = elem_kind(expect);
  const int bits = kw * 8;
  const unsigned b0 = plan->begin_bit < (unsigned)bits ? plan->begin_bit : bits, b1 = plan->end_bit < (unsigned)bits ? plan->end_bit : bits;
  const uint64_t range = (b1 <= b0) ? 0ULL : ((b1 >= 64 ? ~0ULL : ((1ULL << b1) - 1ULL)) & ~((1ULL << b0) - 1ULL));

  DevBuf region, rflags, dbounds;
  uint32_t nbounds = 0;
  if (nseg >= 0) {
    // regions: maximal runs between consecutive boundary points; a region is sorted iff it starts at the begin of a
    // non-empty segment.  Segments are disjoint (as DeviceSegmentedRadixSort requires).
    std::vector<unsigned> hb(nseg), he(nseg);
    if (nseg) {
      HIP_TRY(read_back(hb.data(), d_begin, sizeof(unsigned) * nseg));
      HIP_TRY(read_back(he.data(), d_end, sizeof(unsigned) * nseg));
    }
    std::vector<uint32_t> bounds;
    for (int s = 0; s < nseg; ++s) { bounds.push_back(hb[s]); bounds.push_back(he[s]); }
    std::sort(bounds.begin(), bounds.end());
    bounds.erase(std::unique(bounds.begin(), bounds.end()), bounds.end());
    nbounds = (uint32_t)bounds.size();
    std::vector<uint8_t> sorted_flag(nbounds + 1, 0);           // region r starts at bounds[r-1] (region 0 at row 0)
    for (int s = 0; s < nseg; ++s) {
      if (he[s] <= hb[s]) continue;
      const uint32_t r = (uint32_t)(std::upper_bound(bounds.begin(), bounds.end(), hb[s]) - bounds.begin());
      sorted_flag[r] = 1;
    }
    RMM_TRY(region.alloc(sizeof(uint32_t) * (size_t)n));
    RMM_TRY(rflags.alloc(sorted_flag.size()));
    RMM_TRY(dbounds.alloc(sizeof(uint32_t) * (nbounds ? nbounds : 1)));
    HIP_TRY(hipMemcpy(rflags.p, sorted_flag.data(), sorted_flag.size(), hipMemcpyHostToDevice));
    if (nbounds) HIP_TRY(hipMemcpy(dbounds.p, bounds.data(), sizeof(uint32_t) * nbounds, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(rsw_regions, dim3(stream_grid(n, 1024)), dim3(256), 0, stream0(), dbounds.as<uint32_t>(), nbounds, region.as<uint32_t>(), n);
  }

  DevBuf ka, kb, va, vb, vary, back;
  RMM_TRY(ka.alloc(sizeof(uint64_t) * (size_t)n));
  RMM_TRY(kb.alloc(sizeof(uint64_t) * (size_t)n));
  RMM_TRY(va.alloc(sizeof(uint32_t) * (size_t)n));
  RMM_TRY(vb.alloc(sizeof(uint32_t) * (size_t)n));
  RMM_TRY(vary.alloc(sizeof(unsigned long long)));
  HIP_TRY(hipMemsetAsync(vary.p, 0, sizeof(unsigned long long), stream0()));
  uint64_t *kin = ka.as<uint64_t>(), *kout = kb.as<uint64_t>();
  uint32_t *vin = va.as<uint32_t>(), *vout = vb.as<uint32_t>();
  const int grid = stream_grid(n, 1024);
  GDF_LAUNCH("rsw_images", rsw_images, dim3(grid), dim3(256), 0, stream0(), (const void *)keycol->data, (int)kind, plan->descending,
             (const uint32_t *)region.as<uint32_t>(), (const uint8_t *)rflags.as<uint8_t>(), kin, vin, n, vary.as<unsigned long long>());
  unsigned long long varying = 0;
  HIP_TRY(read_back(&varying, vary.p, sizeof(varying)));
  GDF_TRY((radix_sort_pairs<uint64_t, uint32_t>(kin, kout, vin, vout, n, varying & range)));
  if (nseg >= 0 && nbounds) {
    // second, stable sort on the region number puts every row back into its own region
    hipLaunchKernelGGL(rsw_region_keys, dim3(grid), dim3(256), 0, stream0(), (const uint32_t *)region.as<uint32_t>(), (const uint32_t *)vin, kin, n);
    uint64_t rbits = 0;
    for (uint32_t v = nbounds; v; v >>= 1) rbits = (rbits << 1) | 1ULL;
    GDF_TRY((radix_sort_pairs<uint64_t, uint32_t>(kin, kout, vin, vout, n, rbits)));
  }
  RMM_TRY(back.alloc((size_t)8 * n));
  hipLaunchKernelGGL(rsw_gather, dim3(grid), dim3(256), 0, stream0(), (const uint32_t *)vin, n, kw, (const void *)keycol->data, back.p);
  HIP_TRY(hipMemcpyAsync(keycol->data, back.p, (size_t)kw * n, hipMemcpyDeviceToDevice, stream0()));
  hipLaunchKernelGGL(rsw_gather, dim3(grid), dim3(256), 0, stream0(), (const uint32_t *)vin, n, 8, (const void *)valcol->data, back.p);
  HIP_TRY(hipMemcpyAsync(valcol->data, back.p, (size_t)8 * n, hipMemcpyDeviceToDevice, stream0()));
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
}

static gdf_error radixsort_generic(const RadixPlan *plan, gdf_column *keycol, gdf_column *valcol, int nseg, const unsigned *b, const unsigned *e) {
  GDF_REQUIRE(keycol && valcol, GDF_DATASET_EMPTY);
  GDF_REQUIRE(valcol->dtype == GDF_INT64, GDF_UNSUPPORTED_DTYPE);          // sorting.cu:224
  switch (keycol->dtype) {
    case GDF_INT8: case GDF_INT32: case GDF_INT64: case GDF_FLOAT32: case GDF_FLOAT64:
      return radixsort_api(plan, keycol, valcol, keycol->dtype, nseg, b, e);
    default: return GDF_UNSUPPORTED_DTYPE;
  }
}

}  // namespace gdf_amd

using namespace gdf_amd;

extern "C" {

gdf_radixsort_plan_type *gdf_radixsort_plan(size_t num_items, int descending, unsigned begin_bit, unsigned end_bit) {
  return reinterpret_cast<gdf_radixsort_plan_type *>(new (std::nothrow) RadixPlan{num_items, descending, begin_bit, end_bit, 0, 0});
}
gdf_error gdf_radixsort_plan_setup(gdf_radixsort_plan_type *hdl, size_t sizeof_key, size_t sizeof_val) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_REQUIRE(hdl, GDF_DATASET_EMPTY);
  RadixPlan *p = reinterpret_cast<RadixPlan *>(hdl);
  p->sizeof_key = sizeof_key;
  p->sizeof_val = sizeof_val;
  return GDF_SUCCESS;
  });
}
gdf_error gdf_radixsort_plan_free(gdf_radixsort_plan_type *hdl) {
  return gdf_amd::guarded([&]() -> gdf_error { delete reinterpret_cast<RadixPlan *>(hdl); return GDF_SUCCESS;
  });
}
gdf_segmented_radixsort_plan_type *gdf_segmented_radixsort_plan(size_t num_items, int descending, unsigned begin_bit, unsigned end_bit) {
  return reinterpret_cast<gdf_segmented_radixsort_plan_type *>(new (std::nothrow) RadixPlan{num_items, descending, begin_bit, end_bit, 0, 0});
}
gdf_error gdf_segmented_radixsort_plan_setup(gdf_segmented_radixsort_plan_type *hdl, size_t sizeof_key, size_t sizeof_val) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return gdf_radixsort_plan_setup(reinterpret_cast<gdf_radixsort_plan_type *>(hdl), sizeof_key, sizeof_val);
  });
}
gdf_error gdf_segmented_radixsort_plan_free(gdf_segmented_radixsort_plan_type *hdl) {
  return gdf_amd::guarded([&]() -> gdf_error {
  delete reinterpret_cast<RadixPlan *>(hdl);
  return GDF_SUCCESS;
  });
}

#define GDF_RSORT_IMPL(suffix, dtype_)                                                                                       \
  gdf_error gdf_radixsort_##suffix(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol) {                    \
    return radixsort_api(reinterpret_cast<RadixPlan *>(hdl), keycol, valcol, dtype_, -1, nullptr, nullptr);                   \
  }                                                                                                                           \
  gdf_error gdf_segmented_radixsort_##suffix(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol,  \
                                             unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets) {     \
    return radixsort_api(reinterpret_cast<RadixPlan *>(hdl), keycol, valcol, dtype_, (int)num_segments, d_begin_offsets,      \
                         d_end_offsets);                                                                                      \
  }
GDF_RSORT_IMPL(i8, GDF_INT8)
GDF_RSORT_IMPL(i32, GDF_INT32)
GDF_RSORT_IMPL(i64, GDF_INT64)
GDF_RSORT_IMPL(f32, GDF_FLOAT32)
GDF_RSORT_IMPL(f64, GDF_FLOAT64)
#undef GDF_RSORT_IMPL
gdf_error gdf_radixsort_generic(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return radixsort_generic(reinterpret_cast<RadixPlan *>(hdl), keycol, valcol, -1, nullptr, nullptr);
  });
}
gdf_error gdf_segmented_radixsort_generic(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol,
                                          unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return radixsort_generic(reinterpret_cast<RadixPlan *>(hdl), keycol, valcol, (int)num_segments, d_begin_offsets, d_end_offsets);
  });
}

// sqls_ops.cu:1373-1392.  `cols` is a host ARRAY of gdf_column (not pointers); d_cols /
// d_types are caller-provided device scratch that the reference fills with the data
// pointers / dtypes, and so do we; d_indx receives the sorted row numbers as size_t.
gdf_error gdf_order_by(size_t nrows, gdf_column *cols, size_t ncols, void **d_cols, int *d_types, size_t *d_indx) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_REQUIRE(cols != nullptr && ncols > 0, GDF_DATASET_EMPTY);
  GDF_REQUIRE(!cols->valid, GDF_VALIDITY_UNSUPPORTED);
  GDF_REQUIRE(ncols <= (size_t)MAX_KEY_COLS, GDF_JOIN_TOO_MANY_COLUMNS);
  GDF_REQUIRE(nrows < (size_t)0x7fffffff, GDF_COLUMN_SIZE_TOO_BIG);
  std::vector<gdf_column *> ptrs(ncols);
  std::vector<void *> h_cols(ncols);
  std::vector<int> h_types(ncols);
  for (size_t c = 0; c < ncols; ++c) {
    ptrs[c] = &cols[c];
    h_cols[c] = cols[c].data;
    h_types[c] = (int)cols[c].dtype;
  }
  if (d_cols) HIP_TRY(hipMemcpy(d_cols, h_cols.data(), sizeof(void *) * ncols, hipMemcpyHostToDevice));
  if (d_types) HIP_TRY(hipMemcpy(d_types, h_types.data(), sizeof(int) * ncols, hipMemcpyHostToDevice));
  if (nrows == 0) return GDF_SUCCESS;
  GDF_REQUIRE(d_indx != nullptr, GDF_DATASET_EMPTY);
  for (size_t c = 0; c < ncols; ++c) GDF_REQUIRE(cols[c].data != nullptr, GDF_DATASET_EMPTY);
  KeyTable t;
  GDF_TRY(make_key_table(ptrs.data(), (int)ncols, &t));
  DevBuf perm;
  bool widened = false;
  GDF_TRY(order_rows(t, (uint32_t)nrows, perm, nullptr, nullptr, d_indx, &widened));
  if (!widened) GDF_LAUNCH("rs_widen", rs_widen, dim3(stream_grid(nrows, 1024)), dim3(256), 0, stream0(), perm.as<uint32_t>(), d_indx, (uint32_t)nrows);
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
  });
}

}  // extern "C"
