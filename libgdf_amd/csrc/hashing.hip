// hashing.hip -- gdf_hash and gdf_hash_partition.
//
// Replaces the reference's src/hashing.cu: row hash via thrust::tabulate
// (:83-154) and the four-pass partitioner (:259-377, :401-536 -- hash+LDS
// histogram, two thrust scans, LDS-cursor offsets, one thrust::scatter + stream
// per column).  Here:
//   hash_rows_kernel     : one coalesced pass, 8 B in / 4 B out per row.
//   part_hist_kernel     : per-chunk partition histogram in LDS, written
//                          partition-major so that ONE exclusive scan yields every
//                          (partition, chunk) base.
//   part_scatter_tile_kernel (P <= 256): a tile of 2048 rows is ranked by partition
//                          with LDS atomics, and every column is staged through LDS in
//                          partition order so that a wave stores runs of consecutive
//                          addresses instead of one element per partition cursor.
//   part_scatter_kernel  : re-hashes the chunk (cheaper than storing and
//                          re-reading a 4 B partition id per row), claims
//                          destinations from LDS cursors and moves EVERY column
//                          (data + valid bit) in the same pass.
// Partition rule is the reference's: p = hash & (P-1) when P is a power of two,
// else hash % P (hashing.cu:193-237,434-468); partitions are laid out in
// increasing p; order inside a partition is unspecified.
#include "hash.cuh"
#include "internal.h"
#include "gdf/gdf_amd_ext.h"

#include <cstdlib>
#include <vector>

namespace gdf_amd {

constexpr int HP_THREADS = 256;
constexpr int HP_MAX_CHUNKS = 1024;
constexpr int HP_MAX_LDS_PARTS = 16384;     // 64 KiB of LDS counters
constexpr int HP_MAX_PAYLOAD_COLS = 32;     // columns moved per scatter launch

template <bool MURMUR>
__global__ __launch_bounds__(HP_THREADS) void hash_rows_kernel(KeyTable t, uint32_t *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * HP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * HP_THREADS)
    out[i] = hash_row<MURMUR>(t, i);
}

__device__ __forceinline__ uint32_t part_of(uint32_t h, uint32_t nparts, uint32_t pow2mask) {
  return pow2mask ? (h & pow2mask) : (h % nparts);
}
// TWO-LEVEL partitioning (gdf_hash_partition beyond 1024 partitions): a (tile, partition) run of the one-pass LDS-regrouped
// scatter shrinks to a row or two there (P = 12000: 5.1 ms per 1e8 rows of two int64 columns, 0.10 of the roofline), so the rows
// are first regrouped by SUPER-partition (partition >> kshift, at most 1024 of them) into a temporary table -- level A -- and
// every super-partition is then split into its <= 2^kshift partitions -- level B, whose chunks are pieces of ONE super-partition
// each (PartLevel::pieces).  The split is balanced (P = 12000: 94 x 128 bins) -- both levels then write runs of 32 rows or
// more.  Both levels are the kernels below with another bin function; mode 0 is the one-level call.
struct PartLevel {
  int mode;                   // 0: bin = partition; 1 (level A): bin = partition >> kshift; 2 (level B): bin = partition - (super-partition of the chunk << kshift)
  int kshift;
  uint32_t hashP, hashmask;   // mode != 0: the caller's partition count (the hash modulus); nparts is then the number of BINS
  uint32_t qstride, cstride;  // histogram / offsets index = bin * qstride + chunk * cstride; 0, 0: bin * nchunks + chunk
  // mode 2: level-B chunk c is PIECE c % pieces of super-partition c / pieces: the rows of that super-partition that came from the
  // level-A chunks [piece * group, (piece + 1) * group) -- contiguous in the level-A table, whose scanned histogram
  // bounds[s * nchunks_a + chunk] says where.  Nothing about level B is read back or uploaded: piece boundaries, the piece's
  // super-partition and its histogram column are arithmetic on c.
  const uint32_t *bounds;
  uint32_t pieces, group, nchunks_a, nbins_b;
  // mode 3 (level A's histogram pass): counts FULL partition ids, writes full[chunk * nparts + p] and, summed over the 2^kshift
  // partitions of a super-partition, hist[s * nchunks + chunk]
  uint32_t *full;
  // mode 0, xcd_groups != 0 (round 6, the pair kernel below): inside a partition the chunks are laid out XCD-MAJOR -- chunk c at
  // (c % 8) * xcd_groups + c / 8 of 8 * xcd_groups slots -- so that the runs that meet in a 128-byte line were written behind ONE L2
  // (chunk c runs on XCD c % 8; the join's level 1 does the same with its per-XCD regions)
  uint32_t xcd_groups;
};
__device__ __forceinline__ uint32_t level_bin(uint32_t h, uint32_t nparts, uint32_t pow2mask, const PartLevel &lv, int chunk) {
  if (lv.mode == 0) return part_of(h, nparts, pow2mask);
  const uint32_t p = part_of(h, lv.hashP, lv.hashmask);
  if (lv.mode == 3) return p;
  return lv.mode == 1 ? p >> lv.kshift : p - (((uint32_t)chunk / lv.pieces) << lv.kshift);
}
__device__ __forceinline__ size_t hist_index(uint32_t bin, int chunk, int nchunks, const PartLevel &lv) {
  if (lv.mode == 2) {
    const uint32_t sp = (uint32_t)chunk / lv.pieces, piece = (uint32_t)chunk - sp * lv.pieces;
    return ((size_t)sp * lv.nbins_b + bin) * lv.pieces + piece;
  }
  if (lv.xcd_groups) return ((size_t)bin * 8u + ((uint32_t)chunk & 7u)) * lv.xcd_groups + ((uint32_t)chunk >> 3);
  return lv.qstride ? (size_t)bin * lv.qstride + (size_t)chunk * lv.cstride : (size_t)bin * nchunks + chunk;
}
__device__ __forceinline__ void chunk_rows(int c, int64_t chunk, int64_t n, const PartLevel &lv, int64_t &begin, int64_t &end) {
  if (lv.mode == 2) {
    const uint32_t sp = (uint32_t)c / lv.pieces, piece = (uint32_t)c - sp * lv.pieces;
    const uint32_t c0 = piece * lv.group < lv.nchunks_a ? piece * lv.group : lv.nchunks_a;
    const uint32_t c1 = (piece + 1) * lv.group < lv.nchunks_a ? (piece + 1) * lv.group : lv.nchunks_a;
    begin = lv.bounds[(size_t)sp * lv.nchunks_a + c0];
    end = lv.bounds[(size_t)sp * lv.nchunks_a + c1];
  } else {
    begin = (int64_t)c * chunk;
    end = begin + chunk < n ? begin + chunk : n;
  }
}
// the histogram kernels' write-out of one chunk's LDS counters
__device__ __forceinline__ void hist_write(const uint32_t *lds_cnt, uint32_t nparts, int c, int nchunks, uint32_t *__restrict__ hist, const PartLevel &lv) {
  if (lv.mode == 3) {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) lv.full[(size_t)c * nparts + p] = lds_cnt[p];       // [chunk][partition]: coalesced
    const uint32_t K = 1u << lv.kshift;
    for (uint32_t sp = threadIdx.x; sp < (nparts >> lv.kshift); sp += HP_THREADS) {
      uint32_t sum = 0;
      for (uint32_t k = 0; k < K; ++k) sum += lds_cnt[(sp << lv.kshift) + ((k + sp) & (K - 1))];       // (rotated: neighbouring threads start on different banks)
      hist[(size_t)sp * nchunks + c] = sum;
    }
  } else {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) hist[hist_index(p, c, nchunks, lv)] = lds_cnt[p];
  }
}

// hist layout: hist[p * nchunks + chunk]
template <bool MURMUR>
__global__ __launch_bounds__(HP_THREADS) void part_hist_kernel(KeyTable t, int64_t n, int64_t chunk, int nchunks,
                                                               uint32_t nparts, uint32_t pow2mask,
                                                               uint32_t *__restrict__ hist, PartLevel lv) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cnt[];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) lds_cnt[p] = 0;
    block_sync();
    int64_t begin, end;
    chunk_rows(c, chunk, n, lv, begin, end);
    for (int64_t i = begin + threadIdx.x; i < end; i += HP_THREADS) {
      const uint32_t p = level_bin(hash_row<MURMUR>(t, i), nparts, pow2mask, lv, c);
      atomicAdd(&lds_cnt[p], 1u);
    }
    block_sync();
    hist_write(lds_cnt, nparts, c, nchunks, hist, lv);
    block_sync();
  }
}

// FAST variants (K = uint64_t / uint32_t): ONE 8- or 4-byte key column hashed with Murmur3 (the common shapes:
// an int64 / float64 / date64 or an int32 / float32 / date32 key).
// The key words of HP_BATCH rows per thread are requested together from clamped addresses -- the generic
// hash_row() walks the column list through a switch, which puts every load in its own basic block behind an
// s_waitcnt and leaves one load in flight per wave.
constexpr int HP_BATCH = 8;

template <class K>
__global__ __launch_bounds__(HP_THREADS) void part_hist_fast_kernel(const K *__restrict__ key, int64_t n, int64_t chunk,
                                                                     int nchunks, uint32_t nparts, uint32_t pow2mask,
                                                                     int agg_bits, uint32_t *__restrict__ hist, PartLevel lv) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cnt[];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) lds_cnt[p] = 0;
    block_sync();
    int64_t begin, end;
    chunk_rows(c, chunk, n, lv, begin, end);
    if (begin >= end) {                      // (an empty chunk: zeros)
      hist_write(lds_cnt, nparts, c, nchunks, hist, lv);
      block_sync();
      continue;
    }
    for (int64_t base = begin; base < end; base += HP_THREADS * HP_BATCH) {
      K k[HP_BATCH];
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const int64_t i = base + (int64_t)j * HP_THREADS + threadIdx.x;
        k[j] = key[i < end ? i : end - 1];
      }
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const bool live = base + (int64_t)j * HP_THREADS + threadIdx.x < end;
        const uint32_t part = level_bin(murmur3_32((uint64_t)k[j], (int)sizeof(K)), nparts, pow2mask, lv, c);
        if (agg_bits >= 0) wave_aggregated_inc(lds_cnt, part, agg_bits, live);     // agg_bits: see gdf_hash_partition
        else if (live) atomicAdd(&lds_cnt[part], 1u);
      }
    }
    block_sync();
    hist_write(lds_cnt, nparts, c, nchunks, hist, lv);
    block_sync();
  }
}

struct PayloadCols {
  int ncols;
  const void *in[HP_MAX_PAYLOAD_COLS];
  void *out[HP_MAX_PAYLOAD_COLS];
  const uint8_t *in_valid[HP_MAX_PAYLOAD_COLS];   // null -> skip mask
  uint32_t *out_valid[HP_MAX_PAYLOAD_COLS];       // zero-initialised words
  int width[HP_MAX_PAYLOAD_COLS];
  uint32_t *dst_map;                              // optional: destination of every row (wide tables)
};

__device__ __forceinline__ void move_elem(const void *in, void *out, int width, int64_t src, int64_t dst) {
  switch (width) {
    case 1: ((uint8_t *)out)[dst] = ((const uint8_t *)in)[src]; break;
    case 2: ((uint16_t *)out)[dst] = ((const uint16_t *)in)[src]; break;
    case 4: ((uint32_t *)out)[dst] = ((const uint32_t *)in)[src]; break;
    default: ((uint64_t *)out)[dst] = ((const uint64_t *)in)[src]; break;
  }
}

// offs: the scanned histogram (exclusive), same layout as hist
template <bool MURMUR>
__global__ __launch_bounds__(HP_THREADS) void part_scatter_kernel(KeyTable t, PayloadCols pc, int64_t n, int64_t chunk,
                                                                  int nchunks, uint32_t nparts, uint32_t pow2mask,
                                                                  const uint32_t *__restrict__ offs) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cur[];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) lds_cur[p] = offs[(size_t)p * nchunks + c];
    block_sync();
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < n ? begin + chunk : n;
    for (int64_t i = begin + threadIdx.x; i < end; i += HP_THREADS) {
      const uint32_t p = part_of(hash_row<MURMUR>(t, i), nparts, pow2mask);
      const int64_t dst = atomicAdd(&lds_cur[p], 1u);
      if (pc.dst_map) pc.dst_map[i] = (uint32_t)dst;
      for (int k = 0; k < pc.ncols; ++k) {
        move_elem(pc.in[k], pc.out[k], pc.width[k], i, dst);
        if (pc.out_valid[k]) {
          const bool v = pc.in_valid[k] ? bit_is_set(pc.in_valid[k], i) : true;
          if (v) atomicOr(&pc.out_valid[k][dst >> 5], 1u << (dst & 31));
        }
      }
    }
    block_sync();
  }
}

template <class K>
__global__ __launch_bounds__(HP_THREADS) void part_scatter_fast_kernel(const K *__restrict__ key, PayloadCols pc, int64_t n,
                                                                        int64_t chunk, int nchunks, uint32_t nparts,
                                                                        uint32_t pow2mask, int agg_bits, const uint32_t *__restrict__ offs) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cur[];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) lds_cur[p] = offs[(size_t)p * nchunks + c];
    block_sync();
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < n ? begin + chunk : n;
    for (int64_t base = begin; base < end; base += HP_THREADS * HP_BATCH) {
      K k[HP_BATCH];
      int64_t src[HP_BATCH];
      uint32_t dst[HP_BATCH];
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const int64_t i = base + (int64_t)j * HP_THREADS + threadIdx.x;
        src[j] = i < end ? i : end - 1;
        k[j] = key[src[j]];
      }
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const bool live = base + (int64_t)j * HP_THREADS + threadIdx.x < end;
        const uint32_t part = part_of(murmur3_32((uint64_t)k[j], (int)sizeof(K)), nparts, pow2mask);
        if (agg_bits >= 0) { const uint32_t d = wave_aggregated_inc(lds_cur, part, agg_bits, live); dst[j] = live ? d : 0xffffffffu; }
        else dst[j] = live ? atomicAdd(&lds_cur[part], 1u) : 0xffffffffu;
        if (live && pc.dst_map) pc.dst_map[src[j]] = dst[j];
      }
      for (int col = 0; col < pc.ncols; ++col) {
        uint64_t v[HP_BATCH];
        switch (pc.width[col]) {      // loads of the batch first, then its stores
          case 1:
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) v[j] = ((const uint8_t *)pc.in[col])[src[j]];
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) if (dst[j] != 0xffffffffu) ((uint8_t *)pc.out[col])[dst[j]] = (uint8_t)v[j];
            break;
          case 2:
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) v[j] = ((const uint16_t *)pc.in[col])[src[j]];
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) if (dst[j] != 0xffffffffu) ((uint16_t *)pc.out[col])[dst[j]] = (uint16_t)v[j];
            break;
          case 4:
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) v[j] = ((const uint32_t *)pc.in[col])[src[j]];
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) if (dst[j] != 0xffffffffu) ((uint32_t *)pc.out[col])[dst[j]] = (uint32_t)v[j];
            break;
          default:
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) v[j] = ((const uint64_t *)pc.in[col])[src[j]];
#pragma unroll
            for (int j = 0; j < HP_BATCH; ++j) if (dst[j] != 0xffffffffu) ((uint64_t *)pc.out[col])[dst[j]] = v[j];
        }
        if (pc.out_valid[col]) {
#pragma unroll
          for (int j = 0; j < HP_BATCH; ++j)
            if (dst[j] != 0xffffffffu && (!pc.in_valid[col] || bit_is_set(pc.in_valid[col], src[j])))
              atomicOr(&pc.out_valid[col][dst[j] >> 5], 1u << (dst[j] & 31));
        }
      }
    }
    block_sync();
  }
}

// LDS-regrouped scatter.  Same offsets contract as part_scatter_kernel.  Two shapes:
//   P <= 256  : 256 threads x 16 rows (4096-row tiles, 41 KB of LDS, several workgroups per CU);
//   P <= 1024 : 1024 threads x 12 rows (12288-row tiles, 137 KB, one workgroup per CU) -- a (tile, partition) run is still
//               12 rows = 96 bytes at P = 1024.  The direct scatter that served every fan-out beyond 256 in round 1 issues
//               one store request per row and column: 4.0 ms per 1e8 rows of (int64, float64, int8) at P = 1000.
constexpr int HPT_MAX_PARTS = 256;
constexpr int HPT_BIG_PARTS = 1024;
// TH threads, at most MAXP partitions, FI rows per thread with a directly read key column (half that with the generic row hash:
// that many generic hashes per thread spill)
template <int TH, int MAXP, int FI, bool FAST>
struct HptShape {
  static constexpr int ITEMS = FAST ? FI : FI / 2;
  static constexpr int TILE = TH * ITEMS;
  static constexpr size_t lds_bytes() { return 8 * (size_t)TILE + 2 * (size_t)TILE + 4 * (size_t)(4 * MAXP + 4 + TH / WAVE) + 16; }
};

template <bool MURMUR, int FASTW, int TH, int MAXP, int FI>      // FASTW = 8 / 4: one key column of that width read directly; 0: generic hash_row
__global__ __launch_bounds__(TH) void part_scatter_tile_kernel(KeyTable t, PayloadCols pc, int64_t n, int64_t chunk,
                                                               int nchunks, uint32_t nparts, uint32_t pow2mask,
                                                               const uint32_t *__restrict__ offs, PartLevel lv) {
  using Shape = HptShape<TH, MAXP, FI, FASTW != 0>;
  constexpr int HPT_ITEMS = Shape::ITEMS;
  constexpr int HPT_TILE = Shape::TILE;
  constexpr int HP_THREADS = TH;             // shadows the file-wide 256 inside this kernel
  constexpr int HPT_MAX_PARTS = MAXP;
  extern __shared__ __attribute__((aligned(16))) unsigned char hpt_lds[];
  uint64_t *stage = reinterpret_cast<uint64_t *>(hpt_lds);                 // [TILE]
  uint32_t *hist = reinterpret_cast<uint32_t *>(stage + HPT_TILE);         // [MAXP + 4]
  uint32_t *start = hist + MAXP + 4, *gbase = start + MAXP, *cursor = gbase + MAXP;
  uint32_t *wave_tot = cursor + MAXP;                                      // [TH / WAVE]
  uint16_t *bin_of = reinterpret_cast<uint16_t *>(wave_tot + TH / WAVE);   // [TILE]
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    if (threadIdx.x < nparts) cursor[threadIdx.x] = offs[hist_index(threadIdx.x, c, nchunks, lv)];
    int64_t begin, end;
    chunk_rows(c, chunk, n, lv, begin, end);
    for (int64_t tile = begin; tile < end; tile += HPT_TILE) {
      if (threadIdx.x < HPT_MAX_PARTS) hist[threadIdx.x] = 0;
      block_sync();
      uint32_t pr[HPT_ITEMS];          // partition << 16 | rank within (tile, partition)
      int64_t src[HPT_ITEMS];          // clamped row number: loads are unconditional and stay in flight together
      uint64_t kk[HPT_ITEMS];
#pragma unroll
      for (int k = 0; k < HPT_ITEMS; ++k) {
        const int64_t i = tile + (int64_t)k * HP_THREADS + threadIdx.x;
        src[k] = i < end ? i : end - 1;
        kk[k] = FASTW == 8 ? ((const uint64_t *)t.col[0].data)[src[k]] : (FASTW == 4 ? (uint64_t)((const uint32_t *)t.col[0].data)[src[k]] : 0);
      }
#pragma unroll
      for (int k = 0; k < HPT_ITEMS; ++k) {      // rows beyond the chunk are ranked on a trash counter: no branch around the
        const int64_t i = tile + (int64_t)k * HP_THREADS + threadIdx.x;       // atomic, all of them in flight together
        const uint32_t p = level_bin(FASTW ? murmur3_32(kk[k], FASTW) : hash_row<MURMUR>(t, src[k]), nparts, pow2mask, lv, c);
        pr[k] = i < end ? p : (uint32_t)HPT_MAX_PARTS;
      }
#pragma unroll
      for (int k = 0; k < HPT_ITEMS; ++k) {
        const uint32_t r = atomicAdd(&hist[pr[k]], 1u);
        pr[k] = pr[k] == (uint32_t)HPT_MAX_PARTS ? 0xffffffffu : (pr[k] << 16) | r;
      }
      block_sync();
      {   // exclusive scan of hist[0..nparts) by the 256 threads
        const uint32_t v = threadIdx.x < nparts ? hist[threadIdx.x] : 0;
        const uint32_t incl = wave_scan_incl(v);
        if (lane_id() == WAVE - 1) wave_tot[threadIdx.x / WAVE] = incl;
        block_sync();
        const uint32_t woff = waves_before_sum<HP_THREADS / WAVE>(wave_tot, threadIdx.x);
        if (threadIdx.x < nparts) {
          const uint32_t st = woff + incl - v;
          start[threadIdx.x] = st;
          gbase[threadIdx.x] = cursor[threadIdx.x] - st;
          cursor[threadIdx.x] += v;
        }
      }
      block_sync();
      const uint32_t total = (uint32_t)(end - tile < HPT_TILE ? end - tile : HPT_TILE);
      uint32_t pos[HPT_ITEMS];
#pragma unroll
      for (int k = 0; k < HPT_ITEMS; ++k) {
        pos[k] = 0;
        if (pr[k] != 0xffffffffu) {
          const uint32_t p = pr[k] >> 16;
          pos[k] = start[p] + (pr[k] & 0xffffu);
          bin_of[pos[k]] = (uint16_t)p;
          const int64_t i = tile + (int64_t)k * HP_THREADS + threadIdx.x;
          const uint32_t dst = gbase[p] + pos[k];
          if (pc.dst_map) pc.dst_map[i] = dst;
          for (int col = 0; col < pc.ncols; ++col)
            if (pc.out_valid[col]) {
              const bool v = pc.in_valid[col] ? bit_is_set(pc.in_valid[col], i) : true;
              if (v) atomicOr(&pc.out_valid[col][dst >> 5], 1u << (dst & 31));
            }
        }
      }
      for (int col = 0; col < pc.ncols; ++col) {
        const int width = pc.width[col];
        uint64_t v[HPT_ITEMS];
        switch (width) {
          case 1:
#pragma unroll
            for (int k = 0; k < HPT_ITEMS; ++k) v[k] = ((const uint8_t *)pc.in[col])[src[k]];
            break;
          case 2:
#pragma unroll
            for (int k = 0; k < HPT_ITEMS; ++k) v[k] = ((const uint16_t *)pc.in[col])[src[k]];
            break;
          case 4:
#pragma unroll
            for (int k = 0; k < HPT_ITEMS; ++k) v[k] = ((const uint32_t *)pc.in[col])[src[k]];
            break;
          default:
#pragma unroll
            for (int k = 0; k < HPT_ITEMS; ++k) v[k] = ((const uint64_t *)pc.in[col])[src[k]];
        }
#pragma unroll
        for (int k = 0; k < HPT_ITEMS; ++k)
          if (pr[k] != 0xffffffffu) stage[pos[k]] = v[k];
        block_sync();
        for (uint32_t j = threadIdx.x; j < total; j += HP_THREADS) {
          const uint32_t dst = gbase[bin_of[j]] + j;
          const uint64_t v = stage[j];
          switch (width) {
            case 1: ((uint8_t *)pc.out[col])[dst] = (uint8_t)v; break;
            case 2: ((uint16_t *)pc.out[col])[dst] = (uint16_t)v; break;
            case 4: ((uint32_t *)pc.out[col])[dst] = (uint32_t)v; break;
            default: ((uint64_t *)pc.out[col])[dst] = v; break;
          }
        }
        block_sync();
      }
    }
    block_sync();
  }
}

// ---- one or two 8-byte columns, no masks, one of them the (Murmur3-hashed) key, 16 < P <= 256: the shape of the join's level 1 ----
// (round 6, VERDICT r5 item 8: "the public entry should not be slower than the internal one".)  The generic tile kernel above moves
// column after column through one 8-byte stage, requests a column's words behind the barrier that follows the previous column's flush,
// branches around dead rows and walks its flush in a loop of unknown length: 7.8 - 9.2 ms for 1e9 rows of (int64, int64) at P = 256
// where jk_scatter1_pay regroups the same 32 GB in 6.2.  This kernel is that one's shape on the exact layout: BOTH columns staged
// together (8192 rows x 16 B), the next tile's words requested before the flush and consumed after it, every phase straight-line code
// (rows beyond the chunk rank on a trash counter and leave through per-thread dump slots), XCD-major chunk order inside a partition
// (PartLevel::xcd_groups).  KEYCOL: which of the two columns is hashed.
constexpr int HPP_THREADS = 1024, HPP_ITEMS = 8, HPP_TILE = HPP_THREADS * HPP_ITEMS, HPP_MAX_PARTS = 256;
struct HppLds {
  uint64_t a[HPP_TILE + 2];
  uint64_t b[HPP_TILE + 2];
  uint16_t bin_of[HPP_TILE + 4];
  uint32_t hist[HPP_MAX_PARTS + 64];        // + one trash counter per lane
  uint32_t start[HPP_MAX_PARTS], gbase[HPP_MAX_PARTS], cursor[HPP_MAX_PARTS];
  uint32_t wave_tot[HPP_THREADS / WAVE];
};
template <bool TWO, int KEYCOL>
__global__ __launch_bounds__(HPP_THREADS) void part_scatter_pairs_kernel(const uint64_t *__restrict__ in_a, const uint64_t *__restrict__ in_b,
                                                                         uint64_t *__restrict__ out_a, uint64_t *__restrict__ out_b,
                                                                         uint64_t *__restrict__ dump, int64_t n, int64_t chunk, int nchunks,
                                                                         uint32_t nparts, uint32_t pow2mask, const uint32_t *__restrict__ offs, PartLevel lv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hpp_raw[];
  HppLds &s = *reinterpret_cast<HppLds *>(hpp_raw);
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const uint32_t begin = (uint32_t)((int64_t)c * chunk);                         // rows < 2^31 (int offsets in the ABI)
    const uint32_t end = (int64_t)begin + chunk < n ? (uint32_t)(begin + chunk) : (uint32_t)n;
    if (threadIdx.x < nparts) s.cursor[threadIdx.x] = offs[hist_index(threadIdx.x, c, nchunks, lv)];
    if (threadIdx.x < HPP_MAX_PARTS) s.hist[threadIdx.x] = 0;
    // thread t owns the row PAIRS t, t + THREADS, ... of a tile: one 16-byte load per pair and column
    auto item_row = [](int k, uint32_t tid) -> uint32_t { return 2u * ((uint32_t)(k >> 1) * HPP_THREADS + tid) + (k & 1); };
    uint64_t na[HPP_ITEMS], nb[TWO ? HPP_ITEMS : 1];
    auto prefetch = [&](uint32_t tile) {                 // a pair that would cross `end` is read from the last two rows instead (chunks hold >= 2 rows)
#pragma unroll
      for (int k = 0; k < HPP_ITEMS; k += 2) {
        const uint32_t i = tile + item_row(k, threadIdx.x);
        const uint32_t ic = i + 2 <= end ? i : end - 2;
        na[k] = __builtin_nontemporal_load(in_a + ic); na[k + 1] = __builtin_nontemporal_load(in_a + ic + 1);
        if constexpr (TWO) { nb[k] = __builtin_nontemporal_load(in_b + ic); nb[k + 1] = __builtin_nontemporal_load(in_b + ic + 1); }
      }
    };
    prefetch(begin);                                     // (n >= 2^16: a pair read from [end - 2, end) is inside the column even for a one-row last chunk)
    block_sync();
    for (uint32_t tile = begin; tile < end; tile += HPP_TILE) {
      uint64_t a[HPP_ITEMS], b[TWO ? HPP_ITEMS : 1];
      uint32_t okmask = 0;
#pragma unroll
      for (int k = 0; k < HPP_ITEMS; ++k) {
        const uint32_t i = tile + item_row(k, threadIdx.x);
        // (the first row of a pair that was read from [end - 2, end) because it is row end - 1: the SECOND word loaded)
        const bool shifted = (k & 1) == 0 && i + 1 == end;
        a[k] = shifted ? na[k + 1] : na[k];
        if constexpr (TWO) b[k] = shifted ? nb[k + 1] : nb[k];
        okmask |= (uint32_t)(i < end) << k;
      }
      uint32_t pr[HPP_ITEMS];
#pragma unroll
      for (int k = 0; k < HPP_ITEMS; ++k) {
        const uint32_t p = part_of(murmur3_32(KEYCOL == 0 ? a[k] : b[TWO ? k : 0], 8), nparts, pow2mask);
        pr[k] = (okmask >> k) & 1u ? p : (uint32_t)HPP_MAX_PARTS + (threadIdx.x & 63u);
      }
#pragma unroll
      for (int k = 0; k < HPP_ITEMS; ++k) pr[k] = (pr[k] << 16) | atomicAdd(&s.hist[pr[k]], 1u);      // eight atomics in flight, one wait
      block_sync();
      {
        const uint32_t v = threadIdx.x < nparts ? s.hist[threadIdx.x] : 0;
        const uint32_t incl = wave_scan_incl(v);
        if (lane_id() == WAVE - 1) s.wave_tot[threadIdx.x / WAVE] = incl;
        block_sync();
        const uint32_t woff = waves_before_sum<HPP_THREADS / WAVE>(s.wave_tot, threadIdx.x);
        if (threadIdx.x < nparts) {
          const uint32_t st = woff + incl - v;
          s.start[threadIdx.x] = st;
          s.gbase[threadIdx.x] = s.cursor[threadIdx.x] - st;
          s.cursor[threadIdx.x] += v;
        }
        if (threadIdx.x < HPP_MAX_PARTS) s.hist[threadIdx.x] = 0;      // (nobody reads hist again before the next tile's ranking)
      }
      block_sync();
      {
        uint32_t st[HPP_ITEMS];
#pragma unroll
        for (int k = 0; k < HPP_ITEMS; ++k) st[k] = s.start[(pr[k] >> 16) & (HPP_MAX_PARTS - 1)];
#pragma unroll
        for (int k = 0; k < HPP_ITEMS; ++k) {
          const uint32_t pos = (okmask >> k) & 1u ? st[k] + (pr[k] & 0xffffu) : (uint32_t)HPP_TILE;      // dead rows: the trash slot
          s.a[pos] = a[k];
          if constexpr (TWO) s.b[pos] = b[k];
          s.bin_of[pos] = (uint16_t)(pr[k] >> 16);
        }
      }
      const bool more = tile + HPP_TILE < end;
      __builtin_amdgcn_sched_barrier(0);
      if (more) prefetch(tile + HPP_TILE);            // the next tile's words are on their way while this one is flushed
      block_sync();
      const uint32_t total = end - tile < (uint32_t)HPP_TILE ? end - tile : (uint32_t)HPP_TILE;
#pragma unroll
      for (int h = 0; h < HPP_ITEMS; h += 4) {
        uint64_t va[4], vb[TWO ? 4 : 1];
        uint32_t dst[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t j = threadIdx.x + (uint32_t)(h + k) * HPP_THREADS;
          va[k] = s.a[j];
          if constexpr (TWO) vb[k] = s.b[j];
          dst[k] = s.gbase[s.bin_of[j] & (HPP_MAX_PARTS - 1)] + j;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t j = threadIdx.x + (uint32_t)(h + k) * HPP_THREADS;
          // (a dead slot goes to this thread's own dump words: the stores stay unconditional, no branch per row)
          uint64_t *pa = j < total ? out_a + dst[k] : dump + threadIdx.x;
          *pa = va[k];
          if constexpr (TWO) {
            uint64_t *pb = j < total ? out_b + dst[k] : dump + HPP_THREADS + threadIdx.x;
            *pb = vb[k];
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      block_sync();                                   // the stage is free again; (the prefetched words are consumed at the top of the loop)
    }
    block_sync();
  }
}

// ---- up to four 8-byte columns, no masks, one of them the hashed key, up to 256 partitions: ONE 16384-row stage, column after column ----
// At 256 bins run length decides (the pair kernel's 32-row runs lose to the generic kernel's 48-row ones, above): this kernel regroups
// 16384-row tiles -- 64-row runs, 512 bytes, the join's level-1 tile -- through one 8-byte stage that the columns pass one after the
// other.  Hash + rank + scan happen once per tile; per column: values into the stage at the ranked positions, barrier, flush
// (sixteen unconditional stores per thread, dead slots to per-thread dump words), barrier.  While the key column is flushed the
// next column's words are already requested, while the last column is flushed the next tile's key words are: every load has a flush
// to hide behind.  Straight-line phases, XCD-major chunk order (PartLevel::xcd_groups) as in the pair kernel.
constexpr int HPC_THREADS = 1024, HPC_ITEMS = 16, HPC_TILE = HPC_THREADS * HPC_ITEMS, HPC_MAX_PARTS = 256, HPC_MAX_COLS = 4;
struct HpcCols { int ncols, keycol; const uint64_t *in[HPC_MAX_COLS]; uint64_t *out[HPC_MAX_COLS]; };
struct HpcLds {
  uint64_t stage[HPC_TILE + 2];
  uint8_t bin_of[HPC_TILE + 8];
  uint32_t hist[HPC_MAX_PARTS + 64];        // + one trash counter per lane
  uint32_t start[HPC_MAX_PARTS], gbase[HPC_MAX_PARTS], cursor[HPC_MAX_PARTS];
  uint32_t wave_tot[HPC_THREADS / WAVE];
};
// threadIdx.x through an opaque move: per-thread addresses derived from it are not hoisted out of the tile loops (hoisted, they are
// spilled: 26 VGPRs of scratch in the first version of the kernel below -- every reload a vmcnt(0) wait; join.hip does the same)
__device__ __forceinline__ uint32_t hp_opaque_tid() {
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  return tid;
}
__global__ __launch_bounds__(HPC_THREADS) void part_scatter_cols8_kernel(HpcCols cc, uint64_t *__restrict__ dump, int64_t n, int64_t chunk, int nchunks,
                                                                         uint32_t nparts, uint32_t pow2mask, const uint32_t *__restrict__ offs, PartLevel lv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hpc_raw[];
  HpcLds &s = *reinterpret_cast<HpcLds *>(hpc_raw);
  auto item_row = [](int k, uint32_t tid) -> uint32_t { return 2u * ((uint32_t)(k >> 1) * HPC_THREADS + tid) + (k & 1); };
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const uint32_t begin = (uint32_t)((int64_t)c * chunk);
    const uint32_t end = (int64_t)begin + chunk < n ? (uint32_t)(begin + chunk) : (uint32_t)n;
    if (threadIdx.x < nparts) s.cursor[threadIdx.x] = offs[hist_index(threadIdx.x, c, nchunks, lv)];
    if (threadIdx.x < HPC_MAX_PARTS) s.hist[threadIdx.x] = 0;
    uint64_t nxt[HPC_ITEMS];                               // the words of the column that is staged next (key column of a tile first)
    auto request = [&](const uint64_t *__restrict__ col, uint32_t tile) {      // a pair that would cross `end` is read from the last two rows (n >= 2^16)
      const uint32_t rtid = hp_opaque_tid();
#pragma unroll
      for (int k = 0; k < HPC_ITEMS; k += 2) {
        const uint32_t i = tile + item_row(k, rtid);
        const uint32_t ic = i + 2 <= end ? i : end - 2;
        nxt[k] = __builtin_nontemporal_load(col + ic);
        nxt[k + 1] = __builtin_nontemporal_load(col + ic + 1);
      }
    };
    request(cc.in[cc.keycol], begin);
    block_sync();
    for (uint32_t tile = begin; tile < end; tile += HPC_TILE) {
      const uint32_t total = end - tile < (uint32_t)HPC_TILE ? end - tile : (uint32_t)HPC_TILE;
      uint32_t pos2[HPC_ITEMS / 2];                         // LDS positions of this thread's rows, two per register (<= 16384: 15 bits)
      // ---- the key column: hash, rank, scan ----
      {
        uint32_t pr[HPC_ITEMS];
        const uint32_t htid = hp_opaque_tid();
#pragma unroll
        for (int h = 0; h < HPC_ITEMS; h += 4) {
#pragma unroll
          for (int k = h; k < h + 4; ++k) {
            const uint32_t i = tile + item_row(k, htid);
            const uint64_t key = ((k & 1) == 0 && i + 1 == end) ? nxt[k + 1] : nxt[k];
            nxt[k] = key;                                   // (the shifted word stays where the staging loop below expects it)
            const uint32_t p = part_of(murmur3_32(key, 8), nparts, pow2mask);
            pr[k] = i < end ? p : (uint32_t)HPC_MAX_PARTS + (htid & 63u);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = 0; k < HPC_ITEMS; ++k) pr[k] = (pr[k] << 16) | atomicAdd(&s.hist[pr[k]], 1u);
        block_sync();
        const uint32_t v = threadIdx.x < nparts ? s.hist[threadIdx.x] : 0;
        const uint32_t incl = wave_scan_incl(v);
        if (lane_id() == WAVE - 1) s.wave_tot[threadIdx.x / WAVE] = incl;
        block_sync();
        const uint32_t woff = waves_before_sum<HPC_THREADS / WAVE>(s.wave_tot, threadIdx.x);
        if (threadIdx.x < nparts) {
          const uint32_t st = woff + incl - v;
          s.start[threadIdx.x] = st;
          s.gbase[threadIdx.x] = s.cursor[threadIdx.x] - st;
          s.cursor[threadIdx.x] += v;
        }
        if (threadIdx.x < HPC_MAX_PARTS) s.hist[threadIdx.x] = 0;
        block_sync();
#pragma unroll
        for (int k = 0; k < HPC_ITEMS; ++k) {
          const uint32_t b = pr[k] >> 16;
          const uint32_t at = b < (uint32_t)HPC_MAX_PARTS ? s.start[b] + (pr[k] & 0xffffu) : (uint32_t)HPC_TILE;      // dead rows: the trash slot
          s.bin_of[at] = (uint8_t)b;
          if (k & 1) pos2[k >> 1] |= at << 16; else pos2[k >> 1] = at;
        }
      }
      // ---- column after column through the one stage (the key column first: its words are in nxt already) ----
      for (int ci = 0; ci < cc.ncols; ++ci) {
        const int col = ci == 0 ? cc.keycol : (ci <= cc.keycol ? ci - 1 : ci);
        if (ci > 0) {                                       // (the key column's pair shift was applied while hashing)
#pragma unroll
          for (int k = 0; k < HPC_ITEMS; k += 2) {
            const uint32_t i = tile + item_row(k, hp_opaque_tid());
            if (i + 1 == end) nxt[k] = nxt[k + 1];
          }
        }
#pragma unroll
        for (int k = 0; k < HPC_ITEMS; ++k) s.stage[(k & 1) ? pos2[k >> 1] >> 16 : pos2[k >> 1] & 0xffffu] = nxt[k];
        __builtin_amdgcn_sched_barrier(0);
        // what is staged next is requested now and travels under this column's flush
        if (ci + 1 < cc.ncols) request(cc.in[ci + 1 <= cc.keycol ? ci : ci + 1], tile);
        else if (tile + HPC_TILE < end) request(cc.in[cc.keycol], tile + HPC_TILE);
        block_sync();
        uint64_t *__restrict__ out = cc.out[col];
        const uint32_t ftid = hp_opaque_tid();
#pragma unroll
        for (int h = 0; h < HPC_ITEMS; h += 4) {
          uint64_t v[4];
          uint32_t dst[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t j = ftid + (uint32_t)(h + k) * HPC_THREADS;
            v[k] = s.stage[j];
            dst[k] = s.gbase[s.bin_of[j]] + j;
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const uint32_t j = ftid + (uint32_t)(h + k) * HPC_THREADS;
            uint64_t *at = j < total ? out + dst[k] : dump + ftid;
            *at = v[k];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        block_sync();                                       // the stage is free for the next column
      }
    }
    block_sync();
  }
}

// columns beyond the first HP_MAX_PAYLOAD_COLS follow the recorded row -> destination map
__global__ __launch_bounds__(HP_THREADS) void part_apply_map_kernel(PayloadCols pc, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * HP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * HP_THREADS) {
    const int64_t dst = pc.dst_map[i];
    for (int k = 0; k < pc.ncols; ++k) {
      move_elem(pc.in[k], pc.out[k], pc.width[k], i, dst);
      if (pc.out_valid[k]) {
        const bool v = pc.in_valid[k] ? bit_is_set(pc.in_valid[k], i) : true;
        if (v) atomicOr(&pc.out_valid[k][dst >> 5], 1u << (dst & 31));
      }
    }
  }
}

// gdf_amd_narrow_keys (include/gdf/gdf_amd_ext.h): 8 loads in flight per thread, 8 B in / 4 B out per row
__global__ __launch_bounds__(HP_THREADS) void narrow_keys_kernel(const long long *__restrict__ in, long long lo, unsigned long long span,
                                                                 int32_t *__restrict__ out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * HP_THREADS * 8;
  for (int64_t base = (int64_t)blockIdx.x * HP_THREADS * 8; base < n; base += stride) {
    long long v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = base + (int64_t)k * HP_THREADS + threadIdx.x;
      v[k] = in[i < n ? i : n - 1];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = base + (int64_t)k * HP_THREADS + threadIdx.x;
      const unsigned long long off = (unsigned long long)(v[k] - lo);
      if (i < n) out[i] = off <= span ? (int32_t)off : -1;
    }
  }
}

// gdf_amd_shuffle_partition (include/gdf/gdf_amd_ext.h): the sender side of the multi-GPU shuffle.  Same two passes and the
// same chunk / offsets layout as the FAST hash partition above, but the key is narrowed on the fly (KOUT narrower than
// KIN: the gdf_amd_narrow_keys image) and the travelling payload is the row NUMBER, which needs no input column:
// 8 B read per row in the histogram, 8 B read + 8 B written in the scatter, instead of narrow (12 B) + row-number
// column (4 B) + gdf_hash_partition over two 4-byte columns (4 B + 16 B).
template <class KIN, class KOUT>
__device__ __forceinline__ KOUT shuffle_key(KIN raw, long long lo, unsigned long long span) {
  if constexpr (sizeof(KOUT) < sizeof(KIN)) {
    const unsigned long long off = (unsigned long long)((long long)raw - lo);
    return off <= span ? (KOUT)off : (KOUT)0xffffffffu;
  } else {
    return (KOUT)raw;
  }
}

template <class KIN, class KOUT>
__global__ __launch_bounds__(HP_THREADS) void shuffle_hist_kernel(const KIN *__restrict__ key, long long lo, unsigned long long span,
                                                                   int64_t n, int64_t chunk, int nchunks, uint32_t nparts,
                                                                   uint32_t pow2mask, int agg_bits, uint32_t *__restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cnt[];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) lds_cnt[p] = 0;
    block_sync();
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < n ? begin + chunk : n;
    for (int64_t base = begin; base < end; base += HP_THREADS * HP_BATCH) {
      KIN k[HP_BATCH];
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const int64_t i = base + (int64_t)j * HP_THREADS + threadIdx.x;
        k[j] = key[i < end ? i : end - 1];
      }
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const bool live = base + (int64_t)j * HP_THREADS + threadIdx.x < end;
        const KOUT kk = shuffle_key<KIN, KOUT>(k[j], lo, span);
        const uint32_t part = part_of(murmur3_32((uint64_t)kk, (int)sizeof(KOUT)), nparts, pow2mask);
        if (agg_bits >= 0) wave_aggregated_inc(lds_cnt, part, agg_bits, live);
        else if (live) atomicAdd(&lds_cnt[part], 1u);
      }
    }
    block_sync();
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) hist[(size_t)p * nchunks + c] = lds_cnt[p];
    block_sync();
  }
}

template <class KIN, class KOUT>
__global__ __launch_bounds__(HP_THREADS) void shuffle_scatter_kernel(const KIN *__restrict__ key, long long lo, unsigned long long span,
                                                                      int32_t row_base, int64_t n, int64_t chunk, int nchunks,
                                                                      uint32_t nparts, uint32_t pow2mask, int agg_bits,
                                                                      const uint32_t *__restrict__ offs, KOUT *__restrict__ out_key,
                                                                      int32_t *__restrict__ out_row) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds_cur[];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t p = threadIdx.x; p < nparts; p += HP_THREADS) lds_cur[p] = offs[(size_t)p * nchunks + c];
    block_sync();
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < n ? begin + chunk : n;
    for (int64_t base = begin; base < end; base += HP_THREADS * HP_BATCH) {
      KIN k[HP_BATCH];
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const int64_t i = base + (int64_t)j * HP_THREADS + threadIdx.x;
        k[j] = key[i < end ? i : end - 1];
      }
#pragma unroll
      for (int j = 0; j < HP_BATCH; ++j) {
        const int64_t i = base + (int64_t)j * HP_THREADS + threadIdx.x;
        const bool live = i < end;
        const KOUT kk = shuffle_key<KIN, KOUT>(k[j], lo, span);
        const uint32_t part = part_of(murmur3_32((uint64_t)kk, (int)sizeof(KOUT)), nparts, pow2mask);
        uint32_t dst;
        if (agg_bits >= 0) dst = wave_aggregated_inc(lds_cur, part, agg_bits, live);
        else dst = live ? atomicAdd(&lds_cur[part], 1u) : 0u;
        if (live) {
          out_key[dst] = kk;
          out_row[dst] = row_base + (int32_t)i;
        }
      }
    }
    block_sync();
  }
}

// The same scatter with the batch regrouped by partition in LDS first (4-byte output keys: an LDS slot is the packed
// (key, row) pair): a wave then stores 64 consecutive slots of ONE partition per instruction instead of a handful of
// 4-byte pieces of every partition.  Same chunks and offsets as above.
constexpr int SHT_ITEMS = 8;
constexpr int SHT_TILE = HP_THREADS * SHT_ITEMS;
constexpr int SHT_MAX_PARTS = 64;         // one wave scans the per-partition counts
template <class KIN>
__global__ __launch_bounds__(HP_THREADS) void shuffle_scatter_tile_kernel(const KIN *__restrict__ key, long long lo, unsigned long long span,
                                                                           int32_t row_base, int64_t n, int64_t chunk, int nchunks,
                                                                           uint32_t nparts, uint32_t pow2mask, int agg_bits,
                                                                           const uint32_t *__restrict__ offs, uint32_t *__restrict__ out_key,
                                                                           int32_t *__restrict__ out_row) {
  __shared__ uint64_t stage[SHT_TILE];
  __shared__ uint8_t bin_of[SHT_TILE];
  __shared__ uint32_t cnt[SHT_MAX_PARTS], start[SHT_MAX_PARTS], gbase[SHT_MAX_PARTS], cursor[SHT_MAX_PARTS];
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    if (threadIdx.x < nparts) cursor[threadIdx.x] = offs[(size_t)threadIdx.x * nchunks + c];
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < n ? begin + chunk : n;
    for (int64_t base = begin; base < end; base += SHT_TILE) {
      if (threadIdx.x < SHT_MAX_PARTS) cnt[threadIdx.x] = 0;
      KIN k[SHT_ITEMS];
#pragma unroll
      for (int j = 0; j < SHT_ITEMS; ++j) {
        const int64_t i = base + (int64_t)j * HP_THREADS + threadIdx.x;
        k[j] = key[i < end ? i : end - 1];
      }
      block_sync();
      uint32_t kk[SHT_ITEMS], pr[SHT_ITEMS];          // narrowed key; partition << 16 | rank within (batch, partition)
#pragma unroll
      for (int j = 0; j < SHT_ITEMS; ++j) {
        const bool live = base + (int64_t)j * HP_THREADS + threadIdx.x < end;
        kk[j] = shuffle_key<KIN, uint32_t>(k[j], lo, span);
        const uint32_t part = part_of(murmur3_32((uint64_t)kk[j], 4), nparts, pow2mask);
        const uint32_t r = wave_aggregated_inc(cnt, part, agg_bits, live);
        pr[j] = live ? (part << 16) | r : 0xffffffffu;
      }
      block_sync();
      if (threadIdx.x < WAVE) {                       // exclusive scan of the counts by one wave
        const uint32_t v = threadIdx.x < nparts ? cnt[threadIdx.x] : 0;
        const uint32_t st = wave_scan_incl(v) - v;
        if (threadIdx.x < nparts) {
          start[threadIdx.x] = st;
          gbase[threadIdx.x] = cursor[threadIdx.x] - st;
          cursor[threadIdx.x] += v;
        }
      }
      block_sync();
#pragma unroll
      for (int j = 0; j < SHT_ITEMS; ++j) {
        if (pr[j] != 0xffffffffu) {
          const uint32_t part = pr[j] >> 16, pos = start[part] + (pr[j] & 0xffffu);
          const uint32_t row = (uint32_t)(row_base + (int32_t)(base + (int64_t)j * HP_THREADS + threadIdx.x));
          stage[pos] = ((uint64_t)kk[j] << 32) | row;
          bin_of[pos] = (uint8_t)part;
        }
      }
      block_sync();
      const uint32_t total = (uint32_t)(end - base < SHT_TILE ? end - base : SHT_TILE);
      for (uint32_t j = threadIdx.x; j < total; j += HP_THREADS) {
        const uint64_t tup = stage[j];
        const uint32_t dst = gbase[bin_of[j]] + j;
        out_key[dst] = (uint32_t)(tup >> 32);
        out_row[dst] = (int32_t)(uint32_t)tup;
      }
      // the next batch's first barrier (after its loads) separates this flush from the next regroup
    }
    block_sync();
  }
}

// ---------------------------------------------------------------------------
// gdf_amd_shuffle_partition_stable (include/gdf/gdf_amd_ext.h): the sender side of the multi-GPU shuffle WITHOUT a
// row-number column.  The partition is stable -- the keys of a partition keep their input order -- and for every
// partition a bitmap says which input rows it took, so the j-th key of partition p is row select(bitmap p, j): the
// receiver can name the original row of anything it joins from 1 bit per row instead of a 4-byte row number, and
// the exchange moves 4.125 bytes per row instead of 8.
// Tile = 4 waves x 8 rounds x 64 rows; wave w owns rows [512 w, 512 (w + 1)) of the tile, so input order is (wave,
// round, lane) order and ranks come from per-wave counts, as in the radix sort's rs_scatter.  Tiles are dealt to the XCDs
// in contiguous eighths (rs_scatter again): the partitions' output lines are filled by consecutive tiles.
// ---------------------------------------------------------------------------
constexpr int ST_ROUNDS = 8;
constexpr int ST_WAVES = HP_THREADS / WAVE;
constexpr int ST_TILE = HP_THREADS * ST_ROUNDS;

template <class KIN, class KOUT>
__global__ __launch_bounds__(HP_THREADS) void stable_count_kernel(const KIN *__restrict__ key, long long lo, unsigned long long span,
                                                                   int64_t n, uint32_t ntiles, uint32_t nparts, uint32_t pow2mask,
                                                                   uint32_t drop, uint32_t *__restrict__ counts) {
  __shared__ uint32_t cnt[SHT_MAX_PARTS];
  const uint32_t tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  if (threadIdx.x < SHT_MAX_PARTS) cnt[threadIdx.x] = 0;
  block_sync();
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int64_t wbase = (int64_t)tile * ST_TILE + (int64_t)wave * (ST_ROUNDS * WAVE);
  KIN k[ST_ROUNDS];
#pragma unroll
  for (int r = 0; r < ST_ROUNDS; ++r) {
    const int64_t i = wbase + r * WAVE + lane;
    k[r] = key[i < n ? i : n - 1];
  }
  // counting needs no order: one LDS atomic per row (one ballot per partition instead when there are only one or two
  // counters for 64 lanes to queue on -- see partition_agg_bits)
  uint32_t mine = 0;                     // lane p: rows of this wave that go to partition p (ballot variant)
#pragma unroll
  for (int r = 0; r < ST_ROUNDS; ++r) {
    const KOUT kk = shuffle_key<KIN, KOUT>(k[r], lo, span);
    // drop: a key outside the build range joins nothing -- it stays home (no partition, no bitmap bit)
    const bool live = wbase + r * WAVE + lane < n && !(drop && (uint32_t)kk == 0xffffffffu);
    const uint32_t part = live ? part_of(murmur3_32((uint64_t)kk, (int)sizeof(KOUT)), nparts, pow2mask) : 0xffffffffu;
    if (nparts <= 2) {
      for (uint32_t q = 0; q < nparts; ++q) {
        const unsigned long long m = __ballot(part == q);
        if ((uint32_t)lane == q) mine += (uint32_t)__popcll(m);
      }
    } else if (live) {
      atomicAdd(&cnt[part], 1u);
    }
  }
  if ((uint32_t)lane < nparts && mine) atomicAdd(&cnt[lane], mine);
  block_sync();
  if (threadIdx.x < nparts) counts[(size_t)threadIdx.x * ntiles + tile] = cnt[threadIdx.x];
}

template <class KIN, class KOUT>
__global__ __launch_bounds__(HP_THREADS) void stable_scatter_kernel(const KIN *__restrict__ key, long long lo, unsigned long long span,
                                                                     int64_t n, uint32_t ntiles, uint32_t nparts, uint32_t pow2mask,
                                                                     uint32_t drop, const uint32_t *__restrict__ offsets, KOUT *__restrict__ out_key,
                                                                     unsigned long long *__restrict__ bitmaps, uint64_t words) {
  __shared__ KOUT stage[ST_TILE];
  __shared__ uint8_t bin_of[ST_TILE];
  __shared__ uint32_t wtot[ST_WAVES * SHT_MAX_PARTS];     // rows of wave w for partition p, then: rows of earlier waves
  __shared__ uint32_t start[SHT_MAX_PARTS], gbase[SHT_MAX_PARTS];
  __shared__ uint32_t tile_total;
  extern __shared__ __attribute__((aligned(16))) unsigned long long bm[];   // [nparts][ST_TILE / 64]: bitmap words of the tile (dynamic:
                                                                            // 2 KB at fan-out 8; a static [64][32] cost most of the occupancy)
  const uint32_t tile = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  if (tile >= ntiles) return;
  const int wave = threadIdx.x / WAVE, lane = lane_id();
  const int64_t wbase = (int64_t)tile * ST_TILE + (int64_t)wave * (ST_ROUNDS * WAVE);
  KIN k[ST_ROUNDS];
#pragma unroll
  for (int r = 0; r < ST_ROUNDS; ++r) {
    const int64_t i = wbase + r * WAVE + lane;
    k[r] = key[i < n ? i : n - 1];
  }
  KOUT kk[ST_ROUNDS];
  uint32_t part[ST_ROUNDS];
  uint32_t before[ST_ROUNDS];             // lane p: rows of partition p in the EARLIER rounds of this wave
  unsigned long long same[ST_ROUNDS];     // the lanes of this round that share this lane's partition
  uint32_t run = 0;
#pragma unroll
  for (int r = 0; r < ST_ROUNDS; ++r) {
    const int64_t row0 = wbase + r * WAVE;
    kk[r] = shuffle_key<KIN, KOUT>(k[r], lo, span);
    const bool live = row0 + lane < n && !(drop && (uint32_t)kk[r] == 0xffffffffu);
    part[r] = live ? part_of(murmur3_32((uint64_t)kk[r], (int)sizeof(KOUT)), nparts, pow2mask) : 0xffffffffu;
    before[r] = run;
    same[r] = 0;
    unsigned long long word = 0;
    for (uint32_t q = 0; q < nparts; ++q) {
      const unsigned long long m = __ballot(part[r] == q);
      if ((uint32_t)lane == q) { word = m; run += (uint32_t)__popcll(m); }
      if (part[r] == q) same[r] = m;
    }
    // the tile's bitmap words meet in LDS and leave as 256-byte runs per partition below: written from here -- 8 bytes per
    // (wave, round, partition), 1.4e8 separate store requests per 1.125e9 rows at fan-out 8 -- they cost this kernel more
    // than its key traffic (3.7 TB/s on 12.1 B per row)
    if ((uint32_t)lane < nparts) bm[lane * (ST_TILE / WAVE) + wave * ST_ROUNDS + r] = word;
  }
  if ((uint32_t)lane < nparts) wtot[wave * SHT_MAX_PARTS + lane] = run;
  block_sync();
  if (threadIdx.x < WAVE) {               // one wave: per partition, rows of the earlier waves and the tile total; then the starts
    uint32_t total = 0;
    if ((uint32_t)lane < nparts)
      for (int w = 0; w < ST_WAVES; ++w) {
        const uint32_t c = wtot[w * SHT_MAX_PARTS + lane];
        wtot[w * SHT_MAX_PARTS + lane] = total;
        total += c;
      }
    const uint32_t incl = wave_scan_incl(total);
    const uint32_t st = incl - total;
    if ((uint32_t)lane < nparts) {
      start[lane] = st;
      gbase[lane] = offsets[(size_t)lane * ntiles + tile] - st;
    }
    if (lane == WAVE - 1) tile_total = incl;          // rows of this tile that travel (all of them unless some are dropped)
  }
  block_sync();
  const unsigned long long lt = lane ? (~0ULL >> (64 - lane)) : 0ULL;
#pragma unroll
  for (int r = 0; r < ST_ROUNDS; ++r) {
    const uint32_t q = part[r] == 0xffffffffu ? 0u : part[r];
    const uint32_t earlier = __shfl(before[r], (int)q);          // every lane takes part in the shuffle
    if (part[r] != 0xffffffffu) {
      const uint32_t pos = start[q] + wtot[wave * SHT_MAX_PARTS + q] + earlier + (uint32_t)__popcll(same[r] & lt);
      stage[pos] = kk[r];
      bin_of[pos] = (uint8_t)q;
    }
  }
  block_sync();
  const uint32_t total = tile_total;
  for (uint32_t j = threadIdx.x; j < total; j += HP_THREADS) out_key[gbase[bin_of[j]] + j] = stage[j];
  constexpr uint32_t WPT = ST_TILE / WAVE;                    // bitmap words per tile and partition
  const uint64_t word0 = (uint64_t)tile * WPT;
  for (uint32_t j = threadIdx.x; j < nparts * WPT; j += HP_THREADS) {
    const uint32_t q = j / WPT, w = j % WPT;
    if (word0 + w < words) bitmaps[(size_t)q * words + word0 + w] = bm[q * WPT + w];
  }
}

// gpu_hash_columns (src/hashops.cu:25-151): 64-bit FNV-1a over the little-endian bytes of every column's element,
// columns in order.  The reference XORs each byte as a (signed) `char`, so a byte >= 0x80 is sign-extended to 64
// bits before the XOR (hashops.cu:46-75) -- kept, it is what callers of the reference see.
struct FnvCols { int ncols; const void *data[MAX_KEY_COLS]; int width[MAX_KEY_COLS]; };
__global__ __launch_bounds__(HP_THREADS) void fnv_rows_kernel(FnvCols c, unsigned long long *__restrict__ out, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * HP_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * HP_THREADS) {
    unsigned long long h = 14695981039346656037ull;
    for (int k = 0; k < c.ncols; ++k) {
      uint64_t bits;
      switch (c.width[k]) {
        case 1: bits = ((const uint8_t *)c.data[k])[i]; break;
        case 2: bits = ((const uint16_t *)c.data[k])[i]; break;
        case 4: bits = ((const uint32_t *)c.data[k])[i]; break;
        default: bits = ((const uint64_t *)c.data[k])[i]; break;
      }
      for (int b = 0; b < c.width[k]; ++b) {
        h ^= (unsigned long long)(long long)(signed char)(bits >> (8 * b));
        h *= 1099511628211ull;
      }
    }
    out[i] = h;
  }
}

// One or two partitions put 64 lanes on one or two LDS counters: there the FAST kernels count / rank with
// wave_aggregated_inc (one ballot).  Measured at 2.5e8 rows, histogram pass, aggregated vs per-lane atomics:
// P=1 0.47 vs 0.84 ms, P=2 0.43 vs 0.52, P=4 0.47 vs 0.37, P=8 0.51 vs 0.36 -- from 4 partitions on the ballots cost
// more than the conflicts; the scatter pass does not care either way.  -1 = per-lane atomics.
static int partition_agg_bits(uint32_t P) {
  if (P > 2 || lab::knob_on("GDF_HP_NO_AGG")) return -1;
  int bits = 0;
  while ((1u << bits) < P) ++bits;
  return bits;
}

__global__ void gather_strided_u32(const uint32_t *in, uint32_t *out, int count, size_t stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = in[(size_t)i * stride];
}

}  // namespace gdf_amd

using namespace gdf_amd;


// level B's histogram from level A's full counts: out[p * pieces + j] = sum of full[c * nparts + p] over the chunks c of piece j
// (neighbouring threads = neighbouring partitions: every read is coalesced)
__global__ __launch_bounds__(256) void hpt_piece_counts(const uint32_t *__restrict__ full, uint32_t *__restrict__ out, uint32_t nparts, uint32_t nchunks,
                                                        uint32_t group, uint32_t pieces) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i == 0) out[(size_t)nparts * pieces] = 0;      // the entry behind the last one: the scan makes it the total
  if (i >= (size_t)nparts * pieces) return;
  const uint32_t j = (uint32_t)(i / nparts), p = (uint32_t)(i - (size_t)j * nparts);
  const uint32_t c0 = j * group, c1 = c0 + group < nchunks ? c0 + group : nchunks;
  uint32_t sum = 0;
  for (uint32_t c = c0; c < c1; ++c) sum += full[(size_t)c * nparts + p];
  out[(size_t)p * pieces + j] = sum;
}

// gdf_hash_partition beyond 1024 partitions (PartLevel): level A regroups every column by super-partition into a temporary table,
// level B splits each super-partition.  Same result contract as the one-level call (rows of a partition in a deterministic order,
// partition_offsets exact); costs one extra read + write of every column and a table-sized scratch.
namespace gdf_amd {
static gdf_error hash_partition_two_level(int ncols, gdf_column *input[], const int *columns_to_hash, int num_cols_to_hash, uint32_t P,
                                          gdf_column *output[], int partition_offsets[], bool murmur, bool *declined) {
  // *declined: the table-sized scratch (or a histogram) could not be allocated BEFORE anything was written -- the caller takes the
  // one-level pass, which needs no copy of the table (a request that fitted the device before the two-level path existed still does)
  *declined = false;
#define HP2_ALLOC(call) do { if ((call) != RMM_SUCCESS) { *declined = true; return GDF_SUCCESS; } } while (0)
  const int64_t n = (int64_t)input[0]->size;
  // the smallest shift that leaves at most 1024 super-partitions, raised to the balanced one (K ~ sqrt(P), at most 256 bins at level B)
  int kshift = 0;
  while (((P + (1u << kshift) - 1) >> kshift) > (uint32_t)HPT_BIG_PARTS) ++kshift;
  if (!lab::knob_on("GDF_HP_MIN_SHIFT"))
    while (kshift < 8 && (1u << (2 * kshift)) < P) ++kshift;
  const uint32_t K = 1u << kshift, S = (P + K - 1) >> kshift;
  const uint32_t pow2mask = (P & (P - 1)) == 0 ? P - 1 : 0;

  // the temporary table: data + (where both sides carry one) mask of every column
  std::vector<DevBuf> tmp_data(ncols), tmp_valid(ncols);
  std::vector<gdf_column> tmp_col(ncols);
  std::vector<gdf_column *> key_in(num_cols_to_hash), key_tmp(num_cols_to_hash);
  const size_t mask_words = (mask_bytes((size_t)n) + 3) / 4;
  for (int i = 0; i < ncols; ++i) {
    const int w = dtype_width(input[i]->dtype);
    HP2_ALLOC(tmp_data[i].alloc((size_t)w * (size_t)n));
    tmp_col[i] = *input[i];
    tmp_col[i].data = tmp_data[i].p;
    tmp_col[i].valid = nullptr;
    if (input[i]->valid) {                   // a key column's mask decides nothing here (hash_row ignores it) but travels with its column
      HP2_ALLOC(tmp_valid[i].alloc(mask_words * 4));
      HIP_TRY(hipMemsetAsync(tmp_valid[i].p, 0, mask_words * 4, stream0()));
      tmp_col[i].valid = (gdf_valid_type *)tmp_valid[i].p;
    }
  }
  for (int i = 0; i < num_cols_to_hash; ++i) { key_in[i] = input[columns_to_hash[i]]; key_tmp[i] = &tmp_col[columns_to_hash[i]]; }
  KeyTable t, t2;
  GDF_TRY(make_key_table(key_in.data(), num_cols_to_hash, &t));
  GDF_TRY(make_key_table(key_tmp.data(), num_cols_to_hash, &t2));
  const int fastw = (murmur && t.ncols == 1 && (t.col[0].width == 8 || t.col[0].width == 4)) ? t.col[0].width : 0;

  hipError_t clear_err = hipSuccess;
  auto payload = [&](gdf_column **in, gdf_column **out, bool clear_out_masks) -> PayloadCols {
    PayloadCols pc{};
    pc.ncols = ncols;
    for (int k = 0; k < ncols; ++k) {
      pc.in[k] = in[k]->data;
      pc.out[k] = out[k]->data;
      pc.width[k] = dtype_width(in[k]->dtype);
      const bool masks = in[k]->valid && out[k]->valid;
      pc.in_valid[k] = masks ? in[k]->valid : nullptr;
      pc.out_valid[k] = masks ? (uint32_t *)out[k]->valid : nullptr;
      if (masks && clear_out_masks) {
        const hipError_t e = hipMemsetAsync(out[k]->valid, 0, mask_words * 4, stream0());
        if (e != hipSuccess) clear_err = e;
      }
    }
    return pc;
  };
  std::vector<gdf_column *> tmp_ptr(ncols);
  for (int i = 0; i < ncols; ++i) tmp_ptr[i] = &tmp_col[i];

  // ---- level A.  Its histogram pass counts FULL partition ids per chunk (PartLevel mode 3): summed over a super-partition's K
  // partitions they are level A's own histogram, and summed over the chunks a level-B piece comes from they are level B's -- level B
  // needs no histogram pass, and nothing is read back between the levels ----
  int64_t chunk = (n + HP_MAX_CHUNKS - 1) / HP_MAX_CHUNKS;
  chunk = ((chunk + HP_THREADS * 8 - 1) / (HP_THREADS * 8)) * (HP_THREADS * 8);
  const int nchunks = (int)((n + chunk - 1) / chunk);
  const int grid = nchunks < NUM_CU * 4 ? nchunks : NUM_CU * 4;
  // a level-B piece: the rows a super-partition got from `group` consecutive level-A chunks, ~4 level-B tiles when keys are spread evenly
  // (GDF_HP_PIECE: test switch, the target size in rows)
  const double rows_per_cell = (double)chunk / (double)S;
  uint32_t group = (uint32_t)std::max(1.0, std::min((double)nchunks, (double)lab::path_int("GDF_HP_PIECE", 49152) / std::max(rows_per_cell, 1.0)));
  const uint32_t M = ((uint32_t)nchunks + group - 1) / group;
  const size_t full_words = (size_t)S * K * nchunks, histA_words = (size_t)S * nchunks + 1, histB_words = (size_t)S * K * M + 1;
  DevBuf histA, histB, histFull;
  DevBuf d_starts;
  HP2_ALLOC(histA.alloc(sizeof(uint32_t) * histA_words));
  HP2_ALLOC(histB.alloc(sizeof(uint32_t) * histB_words));
  HP2_ALLOC(histFull.alloc(sizeof(uint32_t) * full_words));
  HP2_ALLOC(d_starts.alloc(sizeof(uint32_t) * P));
#undef HP2_ALLOC
  HIP_TRY(hipMemsetAsync(histA.as<uint32_t>() + (histA_words - 1), 0, sizeof(uint32_t), stream0()));
  PartLevel lh{};
  lh.mode = 3; lh.kshift = kshift; lh.hashP = P; lh.hashmask = pow2mask; lh.full = histFull.as<uint32_t>();
  PartLevel la{};
  la.mode = 1; la.kshift = kshift; la.hashP = P; la.hashmask = pow2mask;
  const size_t ldsA = sizeof(uint32_t) * S * K;
#define HP2_HIST(NAME, LV, TAB, NBINS, NCH, CHUNK, GRID, LDS, OUT)                                                                              \
  do {                                                                                                                                    \
    if (fastw == 8) { HIP_TRY(hipFuncSetAttribute((const void *)part_hist_fast_kernel<uint64_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS))); \
      GDF_LAUNCH(NAME, part_hist_fast_kernel<uint64_t>, dim3(GRID), dim3(HP_THREADS), LDS, stream0(), (const uint64_t *)TAB.col[0].data, n, CHUNK, NCH, NBINS, 0u, -1, OUT, LV); } \
    else if (fastw == 4) { HIP_TRY(hipFuncSetAttribute((const void *)part_hist_fast_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS))); \
      GDF_LAUNCH(NAME, part_hist_fast_kernel<uint32_t>, dim3(GRID), dim3(HP_THREADS), LDS, stream0(), (const uint32_t *)TAB.col[0].data, n, CHUNK, NCH, NBINS, 0u, -1, OUT, LV); } \
    else if (murmur) { HIP_TRY(hipFuncSetAttribute((const void *)part_hist_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS))); \
      GDF_LAUNCH(NAME, part_hist_kernel<true>, dim3(GRID), dim3(HP_THREADS), LDS, stream0(), TAB, n, CHUNK, NCH, NBINS, 0u, OUT, LV); } \
    else { HIP_TRY(hipFuncSetAttribute((const void *)part_hist_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS))); \
      GDF_LAUNCH(NAME, part_hist_kernel<false>, dim3(GRID), dim3(HP_THREADS), LDS, stream0(), TAB, n, CHUNK, NCH, NBINS, 0u, OUT, LV); } \
  } while (0)
  HP2_HIST("part_hist", lh, t, S * K, nchunks, chunk, grid, ldsA, histA.as<uint32_t>());
  HIP_CHECK_LAST();
#undef HP2_HIST
  GDF_TRY(scan_u32(histA.as<uint32_t>(), histA.as<uint32_t>(), histA_words, false));      // [s][chunk] + the total: level-A offsets AND the pieces' row ranges
  // level B's histogram: (partition, piece) = the partition's counts over the piece's level-A chunks; index partition * M + piece
  GDF_LAUNCH("hpt_piece_counts", hpt_piece_counts, dim3((unsigned)((histB_words + 255) / 256)), dim3(256), 0, stream0(), (const uint32_t *)histFull.as<uint32_t>(),
             histB.as<uint32_t>(), (uint32_t)(S * K), (uint32_t)nchunks, group, M);
  HIP_CHECK_LAST();
  GDF_TRY(scan_u32(histB.as<uint32_t>(), histB.as<uint32_t>(), histB_words, false));
#define HP2_TILE(NAME, MUR, FW, TH, MAXP, FI, TAB, NBINS, NCH, CHUNK, GRID, OFFS, LV)                                                           \
  do {                                                                                                                                    \
    const size_t l2 = HptShape<TH, MAXP, FI, (FW) != 0>::lds_bytes();                                                                    \
    HIP_TRY(hipFuncSetAttribute((const void *)part_scatter_tile_kernel<MUR, FW, TH, MAXP, FI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2)); \
    GDF_LAUNCH(NAME, (part_scatter_tile_kernel<MUR, FW, TH, MAXP, FI>), dim3(GRID), dim3(TH), l2, stream0(), TAB, pc, n, CHUNK, NCH, NBINS, 0u, OFFS, LV); \
  } while (0)
  {
    PayloadCols pc = payload(input, tmp_ptr.data(), false);
    if (fastw == 8) HP2_TILE("part_scatter", true, 8, 1024, 1024, 12, t, S, nchunks, chunk, grid, histA.as<uint32_t>(), la);
    else if (fastw == 4) HP2_TILE("part_scatter", true, 4, 1024, 1024, 12, t, S, nchunks, chunk, grid, histA.as<uint32_t>(), la);
    else if (murmur) HP2_TILE("part_scatter", true, 0, 1024, 1024, 12, t, S, nchunks, chunk, grid, histA.as<uint32_t>(), la);
    else HP2_TILE("part_scatter", false, 0, 1024, 1024, 12, t, S, nchunks, chunk, grid, histA.as<uint32_t>(), la);
    HIP_CHECK_LAST();
  }
  // ---- level B: S x M pieces of the temporary table, K bins each, the level-A tile shape (1024 threads x 12 rows: one workgroup per
  // CU with sixteen waves; 256-thread tiles left a CU with two workgroups of four waves, 1.18 vs 0.86 ms).  The scan of
  // (partition, piece) walks a partition's pieces in row order: rows keep the order level A gave them ----
  PartLevel lb{};
  lb.mode = 2; lb.kshift = kshift; lb.hashP = P; lb.hashmask = pow2mask; lb.bounds = histA.as<uint32_t>();
  lb.pieces = M; lb.group = group; lb.nchunks_a = (uint32_t)nchunks; lb.nbins_b = K;
  const int nchunksB = (int)(S * M);
  const int gridB = nchunksB < NUM_CU * 2 ? nchunksB : NUM_CU * 2;
  {
    PayloadCols pc = payload(tmp_ptr.data(), output, true);
    HIP_TRY(clear_err);
    for (int k = 0; k < ncols; ++k)
      if (input[k]->valid && output[k]->valid) output[k]->null_count = input[k]->null_count;
    if (fastw == 8) HP2_TILE("part_scatter_b", true, 8, 1024, 1024, 12, t2, K, nchunksB, (int64_t)0, gridB, histB.as<uint32_t>(), lb);
    else if (fastw == 4) HP2_TILE("part_scatter_b", true, 4, 1024, 1024, 12, t2, K, nchunksB, (int64_t)0, gridB, histB.as<uint32_t>(), lb);
    else if (murmur) HP2_TILE("part_scatter_b", true, 0, 1024, 1024, 12, t2, K, nchunksB, (int64_t)0, gridB, histB.as<uint32_t>(), lb);
    else HP2_TILE("part_scatter_b", false, 0, 1024, 1024, 12, t2, K, nchunksB, (int64_t)0, gridB, histB.as<uint32_t>(), lb);
    HIP_CHECK_LAST();
  }
#undef HP2_TILE
  // partition_offsets (HOST array): every M-th entry of the scanned level-B histogram (index partition * M) is a partition's start
  hipLaunchKernelGGL(gather_strided_u32, dim3((P + 255) / 256), dim3(256), 0, stream0(), (const uint32_t *)histB.as<uint32_t>(), d_starts.as<uint32_t>(), (int)P, (size_t)M);
  HIP_CHECK_LAST();
  HIP_TRY(hipMemcpyAsync(partition_offsets, d_starts.p, sizeof(int) * P, hipMemcpyDeviceToHost, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
}

}  // namespace gdf_amd

extern "C" {

gdf_error gdf_hash(int num_cols, gdf_column **input, gdf_hash_func hash, gdf_column *output) {
  return gdf_amd::guarded([&]() -> gdf_error {
  // argument checks in the order of hashing.cu:85-110
  if (0 == num_cols || nullptr == input || nullptr == output) return GDF_DATASET_EMPTY;
  if (output->dtype != GDF_INT32) return GDF_UNSUPPORTED_DTYPE;
  if (nullptr != input[0] && 0 == input[0]->size) return GDF_SUCCESS;
  if (0 == output->size) return GDF_SUCCESS;
  if (nullptr == output->data) return GDF_DATASET_EMPTY;
  if (hash != GDF_HASH_MURMUR3 && hash != GDF_HASH_IDENTITY) return GDF_INVALID_HASH_FUNCTION;

  KeyTable t;
  GDF_TRY(make_key_table(input, num_cols, &t));
  const int64_t n = t.nrows;
  const int grid = stream_grid((size_t)n, HP_THREADS * 8);
  if (hash == GDF_HASH_MURMUR3)
    GDF_LAUNCH("hash_rows", hash_rows_kernel<true>, dim3(grid), dim3(HP_THREADS), 0, stream0(), t, (uint32_t *)output->data, n);
  else
    hipLaunchKernelGGL(hash_rows_kernel<false>, dim3(grid), dim3(HP_THREADS), 0, stream0(), t, (uint32_t *)output->data, n);
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
  });
}

gdf_error gpu_hash_columns(gdf_column **columns_to_hash, int num_columns, gdf_column *output_column, void *stream) {
  return gdf_amd::guarded([&]() -> gdf_error {
  (void)stream;      // a cudaStream_t* in the reference; the work is complete on return either way
  GDF_REQUIRE(columns_to_hash && num_columns > 0 && output_column && columns_to_hash[0], GDF_DATASET_EMPTY);
  GDF_REQUIRE(num_columns <= MAX_KEY_COLS, GDF_JOIN_TOO_MANY_COLUMNS);
  const int64_t n = (int64_t)columns_to_hash[0]->size;
  FnvCols c{};
  c.ncols = num_columns;
  for (int k = 0; k < num_columns; ++k) {
    GDF_REQUIRE(columns_to_hash[k] && (int64_t)columns_to_hash[k]->size == n, GDF_COLUMN_SIZE_MISMATCH);
    int w = 0;
    GDF_TRY(get_column_byte_width(columns_to_hash[k], &w));
    c.data[k] = columns_to_hash[k]->data;
    c.width[k] = w;
    GDF_REQUIRE(n == 0 || c.data[k], GDF_DATASET_EMPTY);
  }
  if (n == 0) return GDF_SUCCESS;
  GDF_REQUIRE(output_column->data, GDF_DATASET_EMPTY);
  GDF_LAUNCH("fnv_rows", fnv_rows_kernel, dim3(stream_grid((size_t)n, HP_THREADS * 8)), dim3(HP_THREADS), 0, stream0(), c,
             (unsigned long long *)output_column->data, n);
  HIP_CHECK_LAST();
  // output validity = AND of the masks of the inputs that have nulls (hashops.cu:124-134)
  output_column->null_count = 0;
  if (output_column->valid) {
    HIP_TRY(hipMemsetAsync(output_column->valid, 0xff, mask_bytes((size_t)n), stream0()));
    for (int k = 0; k < num_columns; ++k)
      if (columns_to_hash[k]->null_count > 0 && columns_to_hash[k]->valid) GDF_TRY(gdf_validity_and(output_column, columns_to_hash[k], output_column));
  }
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
  });
}

gdf_error gdf_amd_narrow_keys(gdf_column *in, int64_t lo, int64_t hi, gdf_column *out) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_REQUIRE(in && out, GDF_DATASET_EMPTY);
  GDF_REQUIRE(elem_kind(in->dtype) == K_I64 && out->dtype == GDF_INT32, GDF_UNSUPPORTED_DTYPE);
  GDF_REQUIRE(in->size == out->size, GDF_COLUMN_SIZE_MISMATCH);
  GDF_REQUIRE(!in->valid && !out->valid, GDF_VALIDITY_UNSUPPORTED);
  GDF_REQUIRE(hi >= lo && (uint64_t)hi - (uint64_t)lo < 0x7fffffffULL, GDF_INVALID_API_CALL);
  if (in->size == 0) return GDF_SUCCESS;
  GDF_REQUIRE(in->data && out->data, GDF_DATASET_EMPTY);
  const int64_t n = (int64_t)in->size;
  GDF_LAUNCH("narrow_keys", narrow_keys_kernel, dim3(stream_grid((size_t)n, HP_THREADS * 8 * 4)), dim3(HP_THREADS), 0, stream0(),
             (const long long *)in->data, (long long)lo, (unsigned long long)((uint64_t)hi - (uint64_t)lo), (int32_t *)out->data, n);
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
  });
}

gdf_error gdf_amd_shuffle_partition(gdf_column *keys, int narrow, int64_t lo, int64_t hi, int32_t row_base, int num_partitions,
                                    gdf_column *out_keys, gdf_column *out_rows, int partition_offsets[]) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_REQUIRE(keys && out_keys && out_rows && partition_offsets, GDF_DATASET_EMPTY);
  GDF_REQUIRE(num_partitions > 0 && num_partitions <= HP_MAX_LDS_PARTS, GDF_INVALID_API_CALL);
  GDF_REQUIRE(!keys->valid && !out_keys->valid && !out_rows->valid, GDF_VALIDITY_UNSUPPORTED);
  const int win = dtype_width(keys->dtype), wout = dtype_width(out_keys->dtype);
  GDF_REQUIRE((win == 8 || win == 4) && elem_kind(keys->dtype) != K_F32 && elem_kind(keys->dtype) != K_F64, GDF_UNSUPPORTED_DTYPE);
  if (narrow) {
    GDF_REQUIRE(elem_kind(keys->dtype) == K_I64 && out_keys->dtype == GDF_INT32, GDF_UNSUPPORTED_DTYPE);
    GDF_REQUIRE(hi >= lo && (uint64_t)hi - (uint64_t)lo < 0x7fffffffULL, GDF_INVALID_API_CALL);
  } else {
    GDF_REQUIRE(out_keys->dtype == keys->dtype, GDF_PARTITION_DTYPE_MISMATCH);
  }
  GDF_REQUIRE(out_rows->dtype == GDF_INT32, GDF_UNSUPPORTED_DTYPE);
  GDF_REQUIRE(keys->size == out_keys->size && keys->size == out_rows->size, GDF_COLUMN_SIZE_MISMATCH);
  const size_t num_rows = keys->size;
  GDF_REQUIRE(num_rows < (size_t)INT_MAX && (int64_t)row_base + (int64_t)num_rows <= (int64_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  const uint32_t P = (uint32_t)num_partitions;
  if (num_rows == 0) {
    for (uint32_t p = 0; p < P; ++p) partition_offsets[p] = 0;
    return GDF_SUCCESS;
  }
  GDF_REQUIRE(keys->data && out_keys->data && out_rows->data, GDF_DATASET_EMPTY);
  (void)wout;

  const int64_t n = (int64_t)num_rows;
  const uint32_t pow2mask = (P & (P - 1)) == 0 ? P - 1 : 0;
  int64_t chunk = (n + HP_MAX_CHUNKS - 1) / HP_MAX_CHUNKS;             // same chunking as gdf_hash_partition
  chunk = ((chunk + HP_THREADS * 8 - 1) / (HP_THREADS * 8)) * (HP_THREADS * 8);
  const int nchunks = (int)((n + chunk - 1) / chunk);
  const size_t lds = sizeof(uint32_t) * P;
  const int grid = nchunks < NUM_CU * 4 ? nchunks : NUM_CU * 4;
  DevBuf hist, starts;
  RMM_TRY(hist.alloc(sizeof(uint32_t) * (size_t)P * nchunks));
  RMM_TRY(starts.alloc(sizeof(uint32_t) * P));
  const int agg_bits = partition_agg_bits(P);
  const long long llo = narrow ? (long long)lo : 0;
  const unsigned long long span = narrow ? (unsigned long long)((uint64_t)hi - (uint64_t)lo) : 0;
#define SHUFFLE_PASSES(KIN, KOUT)                                                                                                  \
  GDF_LAUNCH("shuffle_hist", (shuffle_hist_kernel<KIN, KOUT>), dim3(grid), dim3(HP_THREADS), lds, stream0(), (const KIN *)keys->data, \
             llo, span, n, chunk, nchunks, P, pow2mask, agg_bits, hist.as<uint32_t>());                                             \
  HIP_CHECK_LAST();                                                                                                                \
  GDF_TRY(scan_u32(hist.as<uint32_t>(), hist.as<uint32_t>(), (size_t)P * nchunks, false));                                         \
  hipLaunchKernelGGL(gather_strided_u32, dim3((P + 255) / 256), dim3(256), 0, stream0(), hist.as<uint32_t>(), starts.as<uint32_t>(), \
                     (int)P, (size_t)nchunks);                                                                                     \
  GDF_LAUNCH("shuffle_scatter", (shuffle_scatter_kernel<KIN, KOUT>), dim3(grid), dim3(HP_THREADS), lds, stream0(),                   \
             (const KIN *)keys->data, llo, span, row_base, n, chunk, nchunks, P, pow2mask, agg_bits, hist.as<uint32_t>(),            \
             (KOUT *)out_keys->data, (int32_t *)out_rows->data);                                                                     \
  HIP_CHECK_LAST();
  // 4-byte output keys and a fan-out of at most 64: the LDS-regrouped scatter (see shuffle_scatter_tile_kernel)
  const bool tile_scatter = (narrow || win == 4) && P <= (uint32_t)SHT_MAX_PARTS && !lab::knob_on("GDF_HP_NO_SHUFFLE_TILE");
  int tile_bits = 0;
  while ((1u << tile_bits) < P) ++tile_bits;
#define SHUFFLE_TILE_PASSES(KIN)                                                                                                    \
  GDF_LAUNCH("shuffle_hist", (shuffle_hist_kernel<KIN, uint32_t>), dim3(grid), dim3(HP_THREADS), lds, stream0(),                     \
             (const KIN *)keys->data, llo, span, n, chunk, nchunks, P, pow2mask, agg_bits, hist.as<uint32_t>());                    \
  HIP_CHECK_LAST();                                                                                                                \
  GDF_TRY(scan_u32(hist.as<uint32_t>(), hist.as<uint32_t>(), (size_t)P * nchunks, false));                                         \
  hipLaunchKernelGGL(gather_strided_u32, dim3((P + 255) / 256), dim3(256), 0, stream0(), hist.as<uint32_t>(), starts.as<uint32_t>(), \
                     (int)P, (size_t)nchunks);                                                                                     \
  GDF_LAUNCH("shuffle_scatter", (shuffle_scatter_tile_kernel<KIN>), dim3(grid), dim3(HP_THREADS), 0, stream0(),                      \
             (const KIN *)keys->data, llo, span, row_base, n, chunk, nchunks, P, pow2mask, tile_bits, hist.as<uint32_t>(),          \
             (uint32_t *)out_keys->data, (int32_t *)out_rows->data);                                                               \
  HIP_CHECK_LAST();
  if (tile_scatter && narrow) { SHUFFLE_TILE_PASSES(uint64_t) }
  else if (tile_scatter) { SHUFFLE_TILE_PASSES(uint32_t) }
  else if (narrow) { SHUFFLE_PASSES(uint64_t, uint32_t) }
  else if (win == 8) { SHUFFLE_PASSES(uint64_t, uint64_t) }
  else { SHUFFLE_PASSES(uint32_t, uint32_t) }
#undef SHUFFLE_PASSES
#undef SHUFFLE_TILE_PASSES
  HIP_TRY(hipMemcpyAsync(partition_offsets, starts.p, sizeof(int) * P, hipMemcpyDeviceToHost, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
  });
}

gdf_error gdf_amd_shuffle_partition_stable(gdf_column *keys, int narrow, int64_t lo, int64_t hi, int num_partitions,
                                           gdf_column *out_keys, uint64_t *bitmaps, int partition_offsets[]) {
  return gdf_amd::guarded([&]() -> gdf_error {
  GDF_REQUIRE(keys && out_keys && bitmaps && partition_offsets, GDF_DATASET_EMPTY);
  GDF_REQUIRE(num_partitions > 0 && num_partitions <= SHT_MAX_PARTS, GDF_INVALID_API_CALL);
  GDF_REQUIRE(!keys->valid && !out_keys->valid, GDF_VALIDITY_UNSUPPORTED);
  const int win = dtype_width(keys->dtype);
  GDF_REQUIRE((win == 8 || win == 4) && elem_kind(keys->dtype) != K_F32 && elem_kind(keys->dtype) != K_F64, GDF_UNSUPPORTED_DTYPE);
  if (narrow) {
    GDF_REQUIRE(elem_kind(keys->dtype) == K_I64 && out_keys->dtype == GDF_INT32, GDF_UNSUPPORTED_DTYPE);
    GDF_REQUIRE(hi >= lo && (uint64_t)hi - (uint64_t)lo < 0x7fffffffULL, GDF_INVALID_API_CALL);
  } else {
    GDF_REQUIRE(out_keys->dtype == keys->dtype, GDF_PARTITION_DTYPE_MISMATCH);
  }
  GDF_REQUIRE(keys->size == out_keys->size, GDF_COLUMN_SIZE_MISMATCH);
  const size_t num_rows = keys->size;
  GDF_REQUIRE(num_rows < (size_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  const uint32_t P = (uint32_t)num_partitions;
  const uint32_t drop = narrow == 2 ? 1u : 0u;
  if (num_rows == 0) {
    for (uint32_t p = 0; p < P + drop; ++p) partition_offsets[p] = 0;
    return GDF_SUCCESS;
  }
  GDF_REQUIRE(keys->data && out_keys->data, GDF_DATASET_EMPTY);
  const int64_t n = (int64_t)num_rows;
  const uint32_t pow2mask = (P & (P - 1)) == 0 ? P - 1 : 0;
  const uint32_t ntiles = (uint32_t)((n + ST_TILE - 1) / ST_TILE);
  const uint32_t grid = (ntiles + 7) / 8 * 8;
  const uint64_t words = (uint64_t)((n + 63) / 64);
  DevBuf counts, starts;
  RMM_TRY(counts.alloc(sizeof(uint32_t) * ((size_t)P * ntiles + 1)));     // + the grand total, which the scan leaves behind the matrix
  RMM_TRY(starts.alloc(sizeof(uint32_t) * (P + 1)));
  const long long llo = narrow ? (long long)lo : 0;
  const unsigned long long span = narrow ? (unsigned long long)((uint64_t)hi - (uint64_t)lo) : 0;
#define STABLE_PASSES(KIN, KOUT)                                                                                                    \
  GDF_LAUNCH("stable_count", (stable_count_kernel<KIN, KOUT>), dim3(grid), dim3(HP_THREADS), 0, stream0(), (const KIN *)keys->data, llo, \
             span, n, ntiles, P, pow2mask, drop, counts.as<uint32_t>());                                                            \
  HIP_CHECK_LAST();                                                                                                                \
  HIP_TRY(hipMemsetAsync(counts.as<uint32_t>() + (size_t)P * ntiles, 0, sizeof(uint32_t), stream0()));                              \
  GDF_TRY(scan_u32(counts.as<uint32_t>(), counts.as<uint32_t>(), (size_t)P * ntiles + 1, false));                                  \
  hipLaunchKernelGGL(gather_strided_u32, dim3((P + 256) / 256), dim3(256), 0, stream0(), counts.as<uint32_t>(), starts.as<uint32_t>(), \
                     (int)P + 1, (size_t)ntiles);                                                                                  \
  GDF_LAUNCH("stable_scatter", (stable_scatter_kernel<KIN, KOUT>), dim3(grid), dim3(HP_THREADS), sizeof(unsigned long long) * P * (ST_TILE / WAVE), \
             stream0(), (const KIN *)keys->data, \
             llo, span, n, ntiles, P, pow2mask, drop, (const uint32_t *)counts.as<uint32_t>(), (KOUT *)out_keys->data,              \
             (unsigned long long *)bitmaps, words);                                                                                \
  HIP_CHECK_LAST();
  if (narrow) { STABLE_PASSES(uint64_t, uint32_t) }
  else if (win == 8) { STABLE_PASSES(uint64_t, uint64_t) }
  else { STABLE_PASSES(uint32_t, uint32_t) }
#undef STABLE_PASSES
  HIP_TRY(hipMemcpyAsync(partition_offsets, starts.p, sizeof(int) * (P + drop), hipMemcpyDeviceToHost, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
  });
}

gdf_error gdf_hash_partition(int num_input_cols, gdf_column *input[], int columns_to_hash[], int num_cols_to_hash,
                             int num_partitions, gdf_column *partitioned_output[], int partition_offsets[],
                             gdf_hash_func hash) {
  return gdf_amd::guarded([&]() -> gdf_error {
  // checks in the order of hashing.cu:573-607
  if (0 == num_input_cols || 0 == num_cols_to_hash || 0 == num_partitions || nullptr == input ||
      nullptr == partitioned_output || nullptr == columns_to_hash || nullptr == partition_offsets)
    return GDF_INVALID_API_CALL;
  const size_t num_rows = input[0]->size;
  if (0 == num_rows) return GDF_SUCCESS;
  for (int i = 0; i < num_input_cols; ++i) {
    if (nullptr == input[i]->data || nullptr == partitioned_output[i]->data) return GDF_DATASET_EMPTY;
    if (input[i]->dtype != partitioned_output[i]->dtype) return GDF_PARTITION_DTYPE_MISMATCH;
    if (num_rows != input[i]->size || num_rows != partitioned_output[i]->size) return GDF_COLUMN_SIZE_MISMATCH;
  }
  if (hash != GDF_HASH_MURMUR3 && hash != GDF_HASH_IDENTITY) return GDF_INVALID_HASH_FUNCTION;
  if (num_partitions < 0 || num_partitions > HP_MAX_LDS_PARTS) return GDF_INVALID_API_CALL;
  if (num_rows >= (size_t)INT_MAX) return GDF_COLUMN_SIZE_TOO_BIG;   // int offsets in the ABI

  gdf_nvtx_range_push("LIBGDF_HASH_PARTITION", GDF_PURPLE);   // hashing.cu:609
  struct Pop { ~Pop() { gdf_nvtx_range_pop(); } } pop;

  std::vector<gdf_column *> key_cols(num_cols_to_hash);
  for (int i = 0; i < num_cols_to_hash; ++i) {
    if (columns_to_hash[i] < 0 || columns_to_hash[i] >= num_input_cols) return GDF_INVALID_API_CALL;
    key_cols[i] = input[columns_to_hash[i]];
  }
  KeyTable t;
  GDF_TRY(make_key_table(key_cols.data(), num_cols_to_hash, &t));
  for (int i = 0; i < num_input_cols; ++i)
    if (dtype_width(input[i]->dtype) < 0) return GDF_UNSUPPORTED_DTYPE;

  const int64_t n = (int64_t)num_rows;
  const uint32_t P = (uint32_t)num_partitions;
  const uint32_t pow2mask = (P & (P - 1)) == 0 ? P - 1 : 0;   // P==1 -> mask 0 -> h % 1 == 0, same result
  // beyond the fan-out one LDS-regrouped pass serves: two levels (hash_partition_two_level)
  if (P > (uint32_t)HPT_BIG_PARTS && num_input_cols <= HP_MAX_PAYLOAD_COLS && n >= ((int64_t)1 << 18) && !lab::path_on("GDF_HP_ONE_LEVEL")) {
    bool declined = false;
    const gdf_error e = hash_partition_two_level(num_input_cols, input, columns_to_hash, num_cols_to_hash, P, partitioned_output,
                                                 partition_offsets, hash == GDF_HASH_MURMUR3, &declined);
    if (!declined) return e;           // (declined: no room for the temporary table -- the one-level pass below)
  }
  // chunking: at most HP_MAX_CHUNKS chunks, each a multiple of the block size
  int64_t chunk = (n + HP_MAX_CHUNKS - 1) / HP_MAX_CHUNKS;
  chunk = ((chunk + HP_THREADS * 8 - 1) / (HP_THREADS * 8)) * (HP_THREADS * 8);
  const int nchunks = (int)((n + chunk - 1) / chunk);
  const size_t lds = sizeof(uint32_t) * P;
  const int grid = nchunks < NUM_CU * 4 ? nchunks : NUM_CU * 4;

  DevBuf hist, starts;
  const bool murmur = hash == GDF_HASH_MURMUR3;
  const int agg_bits = partition_agg_bits(P);
  const int fastw = (murmur && t.ncols == 1 && (t.col[0].width == 8 || t.col[0].width == 4) && !lab::knob_on("GDF_HP_NO_FAST")) ? t.col[0].width : 0;
  // the PAIR kernel (part_scatter_pairs_kernel): one or two 8-byte columns without masks, one of them the hashed key, 16 < P <= 256 --
  // (key, value) tables, the partial aggregates of the distributed group-by.  Its chunks are laid out XCD-major inside a partition.
  // Measured at 1e9 rows x 2 int64 columns, alternating with the generic tile kernel in one process (profiles/r6_m_hash_partition_pairs_ab.jsonl):
  // P = 32 6.38 against 7.08 ms in the scatter kernel, P = 64 7.07 against 7.17, P = 256 8.70 against 8.18 -- at 256 bins its 8192-row tile
  // leaves 32-row runs where the generic kernel's 12288-row single-column stage leaves 48-row ones, and run length wins: up to 64 partitions.
  bool pairs = fastw == 8 && num_input_cols <= 2 && P > 16 && P <= (uint32_t)lab::knob_int("GDF_HP_PAIRS_MAX", 64) && P <= (uint32_t)HPP_MAX_PARTS &&
               n >= ((int64_t)1 << 16) && !lab::path_on("GDF_HP_NO_PAIRS");
  for (int i = 0; i < num_input_cols && pairs; ++i)
    pairs = dtype_width(input[i]->dtype) == 8 && !(input[i]->valid && partitioned_output[i]->valid);
  // the 16384-row single-stage kernel (part_scatter_cols8_kernel): up to four 8-byte columns without masks, up to 256 partitions, what the
  // pair kernel does not take
  bool cols8 = !pairs && fastw == 8 && num_input_cols <= HPC_MAX_COLS && P > 16 && P <= (uint32_t)HPC_MAX_PARTS && n >= ((int64_t)1 << 16) &&
               !lab::path_on("GDF_HP_NO_COLS8");
  for (int i = 0; i < num_input_cols && cols8; ++i)
    cols8 = dtype_width(input[i]->dtype) == 8 && !(input[i]->valid && partitioned_output[i]->valid);
  PartLevel lv0{};
  size_t hist_words = (size_t)P * nchunks, start_stride = (size_t)nchunks;
  if (pairs || cols8) {
    lv0.xcd_groups = (uint32_t)((nchunks + 7) / 8);
    start_stride = (size_t)8 * lv0.xcd_groups;
    hist_words = (size_t)P * start_stride;
  }
  RMM_TRY(hist.alloc(sizeof(uint32_t) * hist_words));
  RMM_TRY(starts.alloc(sizeof(uint32_t) * P));
  if ((pairs || cols8) && start_stride != (size_t)nchunks) HIP_TRY(hipMemsetAsync(hist.p, 0, sizeof(uint32_t) * hist_words, stream0()));      // (slots of chunks that do not exist)
  if (fastw == 8)
    GDF_LAUNCH("part_hist", part_hist_fast_kernel<uint64_t>, dim3(grid), dim3(HP_THREADS), lds, stream0(), (const uint64_t *)t.col[0].data, n, chunk,
               nchunks, P, pow2mask, agg_bits, hist.as<uint32_t>(), lv0);
  else if (fastw == 4)
    GDF_LAUNCH("part_hist", part_hist_fast_kernel<uint32_t>, dim3(grid), dim3(HP_THREADS), lds, stream0(), (const uint32_t *)t.col[0].data, n, chunk,
               nchunks, P, pow2mask, agg_bits, hist.as<uint32_t>(), PartLevel{});
  else if (murmur)
    GDF_LAUNCH("part_hist", part_hist_kernel<true>, dim3(grid), dim3(HP_THREADS), lds, stream0(), t, n, chunk, nchunks, P, pow2mask, hist.as<uint32_t>(), PartLevel{});
  else
    hipLaunchKernelGGL(part_hist_kernel<false>, dim3(grid), dim3(HP_THREADS), lds, stream0(), t, n, chunk, nchunks, P, pow2mask, hist.as<uint32_t>(), PartLevel{});
  HIP_CHECK_LAST();
  GDF_TRY(scan_u32(hist.as<uint32_t>(), hist.as<uint32_t>(), hist_words, false));
  hipLaunchKernelGGL(gather_strided_u32, dim3((P + 255) / 256), dim3(256), 0, stream0(), hist.as<uint32_t>(),
                     starts.as<uint32_t>(), (int)P, start_stride);
  HIP_CHECK_LAST();
  if (cols8) {
    HpcCols hc{};
    hc.ncols = num_input_cols;
    for (int i = 0; i < num_input_cols; ++i) {
      hc.in[i] = (const uint64_t *)input[i]->data;
      hc.out[i] = (uint64_t *)partitioned_output[i]->data;
      if (input[i] == key_cols[0]) hc.keycol = i;
    }
    DevBuf dump;
    RMM_TRY(dump.alloc(sizeof(uint64_t) * HPC_THREADS));
    const int pgrid = nchunks < NUM_CU * 4 ? nchunks : NUM_CU * 4;
    HIP_TRY(hipFuncSetAttribute((const void *)part_scatter_cols8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(HpcLds)));
    GDF_LAUNCH("part_scatter", part_scatter_cols8_kernel, dim3(pgrid), dim3(HPC_THREADS), sizeof(HpcLds), stream0(), hc, dump.as<uint64_t>(), n, chunk,
               nchunks, P, pow2mask, (const uint32_t *)hist.as<uint32_t>(), lv0);
    HIP_CHECK_LAST();
    HIP_TRY(hipMemcpyAsync(partition_offsets, starts.p, sizeof(int) * P, hipMemcpyDeviceToHost, stream0()));
    HIP_TRY(hipStreamSynchronize(stream0()));
    return GDF_SUCCESS;
  }
  if (pairs) {
    int keycol = 0;
    for (int i = 0; i < num_input_cols; ++i) if (input[i] == key_cols[0]) keycol = i;
    DevBuf dump;
    RMM_TRY(dump.alloc(sizeof(uint64_t) * 2 * HPP_THREADS));
    const bool two = num_input_cols == 2;
    const uint64_t *ia = (const uint64_t *)input[0]->data, *ib = two ? (const uint64_t *)input[1]->data : nullptr;
    uint64_t *oa = (uint64_t *)partitioned_output[0]->data, *ob = two ? (uint64_t *)partitioned_output[1]->data : nullptr;
    const int pgrid = nchunks < NUM_CU * 4 ? nchunks : NUM_CU * 4;          // (a multiple of 8 when it is not nchunks itself: chunk c runs on XCD c % 8)
#define HPP_LAUNCH(TWO, KC)                                                                                                            \
  do {                                                                                                                                 \
    HIP_TRY(hipFuncSetAttribute((const void *)part_scatter_pairs_kernel<TWO, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(HppLds))); \
    GDF_LAUNCH("part_scatter", (part_scatter_pairs_kernel<TWO, KC>), dim3(pgrid), dim3(HPP_THREADS), sizeof(HppLds), stream0(), ia, ib, oa, ob,   \
               dump.as<uint64_t>(), n, chunk, nchunks, P, pow2mask, (const uint32_t *)hist.as<uint32_t>(), lv0);                        \
  } while (0)
    if (!two) HPP_LAUNCH(false, 0);
    else if (keycol == 0) HPP_LAUNCH(true, 0);
    else HPP_LAUNCH(true, 1);
#undef HPP_LAUNCH
    HIP_CHECK_LAST();
    HIP_TRY(hipMemcpyAsync(partition_offsets, starts.p, sizeof(int) * P, hipMemcpyDeviceToHost, stream0()));
    HIP_TRY(hipStreamSynchronize(stream0()));
    return GDF_SUCCESS;
  }

  // move the columns, HP_MAX_PAYLOAD_COLS per launch; a wider table records the
  // row -> destination map in the first launch and replays it for the rest
  DevBuf dst_map;
  if (num_input_cols > HP_MAX_PAYLOAD_COLS) RMM_TRY(dst_map.alloc(sizeof(uint32_t) * num_rows));
  for (int first = 0; first < num_input_cols; first += HP_MAX_PAYLOAD_COLS) {
    PayloadCols pc{};
    pc.ncols = num_input_cols - first < HP_MAX_PAYLOAD_COLS ? num_input_cols - first : HP_MAX_PAYLOAD_COLS;
    for (int k = 0; k < pc.ncols; ++k) {
      gdf_column *ci = input[first + k], *co = partitioned_output[first + k];
      pc.in[k] = ci->data;
      pc.out[k] = co->data;
      pc.width[k] = dtype_width(ci->dtype);
      // masks travel only when both sides carry one (gdf_table.cuh:1101-1116)
      if (ci->valid && co->valid) {
        pc.in_valid[k] = ci->valid;
        pc.out_valid[k] = (uint32_t *)co->valid;
        // atomicOr works on 4-byte words: clear the mask rounded up to a word.  Arrow
        // buffers are padded to 64 bytes, so the trailing bytes belong to the column.
        HIP_TRY(hipMemsetAsync(co->valid, 0, ((mask_bytes(num_rows) + 3) / 4) * 4, stream0()));
        co->null_count = ci->null_count;
      } else {
        pc.in_valid[k] = nullptr;
        pc.out_valid[k] = nullptr;
      }
    }
    pc.dst_map = dst_map.as<uint32_t>();
    if (first > 0)
      hipLaunchKernelGGL(part_apply_map_kernel, dim3(stream_grid(num_rows, HP_THREADS * 4)), dim3(HP_THREADS), 0, stream0(), pc, n);
    else if (P > 16 && P <= (uint32_t)HPT_BIG_PARTS && !lab::knob_on("GDF_HP_NO_TILE")) {
      // measured at 1e8 rows x 2 int64 columns: P=256 1.47 ms vs 2.83 ms direct; at P=8 the direct kernel's runs are
      // long enough already (1.10 vs 1.19 ms), so small fan-outs keep it
#define HPT_LAUNCH(MUR, FW, TH, MAXP, FI)                                                                                              \
  do {                                                                                                                                 \
    const size_t tl = HptShape<TH, MAXP, FI, (FW) != 0>::lds_bytes();                                                                  \
    HIP_TRY(hipFuncSetAttribute((const void *)part_scatter_tile_kernel<MUR, FW, TH, MAXP, FI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tl)); \
    GDF_LAUNCH("part_scatter", (part_scatter_tile_kernel<MUR, FW, TH, MAXP, FI>), dim3(grid), dim3(TH), tl, stream0(), t, pc, n, chunk, \
               nchunks, P, pow2mask, hist.as<uint32_t>(), PartLevel{});                                                                \
  } while (0)
      // 12288-row tiles of a 1024-thread workgroup win over 4096-row tiles of 256 threads at EVERY fan-out they share (1e8 rows x
      // 2 int64 columns, scatter kernel: P = 64 0.73 vs 0.88 ms, P = 256 0.82 vs 1.07 ms): three times the run length.  The small
      // shape stays behind GDF_HP_BIG_FROM=256
      const uint32_t big_from = (uint32_t)lab::knob_int("GDF_HP_BIG_FROM", 16);
      if (P <= big_from && P <= (uint32_t)HPT_MAX_PARTS) {
        if (fastw == 8) HPT_LAUNCH(true, 8, 256, 256, 16);
        else if (fastw == 4) HPT_LAUNCH(true, 4, 256, 256, 16);
        else if (murmur) HPT_LAUNCH(true, 0, 256, 256, 16);
        else HPT_LAUNCH(false, 0, 256, 256, 16);
      } else {
        if (fastw == 8) HPT_LAUNCH(true, 8, 1024, 1024, 12);
        else if (fastw == 4) HPT_LAUNCH(true, 4, 1024, 1024, 12);
        else if (murmur) HPT_LAUNCH(true, 0, 1024, 1024, 12);
        else HPT_LAUNCH(false, 0, 1024, 1024, 12);
      }
#undef HPT_LAUNCH
    } else if (fastw == 8)
      GDF_LAUNCH("part_scatter", part_scatter_fast_kernel<uint64_t>, dim3(grid), dim3(HP_THREADS), lds, stream0(), (const uint64_t *)t.col[0].data, pc, n,
                 chunk, nchunks, P, pow2mask, agg_bits, hist.as<uint32_t>());
    else if (fastw == 4)
      GDF_LAUNCH("part_scatter", part_scatter_fast_kernel<uint32_t>, dim3(grid), dim3(HP_THREADS), lds, stream0(), (const uint32_t *)t.col[0].data, pc, n,
                 chunk, nchunks, P, pow2mask, agg_bits, hist.as<uint32_t>());
    else if (murmur)
      GDF_LAUNCH("part_scatter", part_scatter_kernel<true>, dim3(grid), dim3(HP_THREADS), lds, stream0(), t, pc, n, chunk, nchunks, P, pow2mask, hist.as<uint32_t>());
    else
      hipLaunchKernelGGL(part_scatter_kernel<false>, dim3(grid), dim3(HP_THREADS), lds, stream0(), t, pc, n, chunk, nchunks, P, pow2mask, hist.as<uint32_t>());
    HIP_CHECK_LAST();
  }
  // partition_offsets is a HOST array (hashing.cu:499-503)
  HIP_TRY(hipMemcpyAsync(partition_offsets, starts.p, sizeof(int) * P, hipMemcpyDeviceToHost, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
  });
}

}  // extern "C"
