// join.hip -- gdf_inner_join / gdf_left_join / gdf_full_join, HASH method.
//
// Reference path being replaced: src/join/joining.cu:282-653 ->
// join/joining.h:46-74 -> join/hash/join_compute_api.h:341-551 with the kernels of
// join/hash/join_kernels.cuh:46-455 over the global-memory multimap of
// src/hashmap/concurrent_unordered_multimap.cuh.  That design does one dependent
// random HBM read per probe step into a 2*N_build-slot table plus a second random
// read of the build key; here NO random HBM access remains:
//
//   1. jk_hist      both relations are turned into (key, row) tuples and radix
//   2. jk_scatter1  partitioned on the top FB <= 15 bits of hash_a(raw key) (a 32-bit
//   3. jk_scatter2  internal hash), so that one build partition fits LDS.  Two scatter
//                   levels (<= 256-way each); tiles are regrouped in LDS so a wave
//                   writes runs of consecutive addresses.  The histogram pass runs on
//                   the build side only: the probe side is laid out with per-partition
//                   slack and atomic fill counters (partition_side_spec), falling back
//                   to the exact histogram layout when a partition outgrows its room.
//   4. jk_probe     one workgroup per (partition, probe chunk): stages the build
//                   partition in LDS, builds a cuckoo table of POSITIONS over it
//                   (lookup = two independent reads, no data-dependent loop; linear
//                   probing per unit when the keys repeat), streams the probe tuples
//                   past it and writes the (probe row, build row) pairs with
//                   wave-ballot compaction into the unit's private output range.
//   Output sizing: foreign-key -> primary-key joins (one pair per probe tuple, checked
//   on a sample of units) run jk_probe<WRITE> once into slots laid out by the probe
//   counts; everything else runs jk_probe<COUNT>, scans the unit counts to exact
//   offsets, then jk_probe<WRITE>.
//
// Tuple formats.  NARROW (8 bytes): key32 << 32 | row -- used whenever the key fits 32
// bits: every key of <= 4 bytes, and 8-byte integer keys whose build-side range
// (max - min, one extra reduction over the build keys) is below 2^32; probe keys outside
// that range cannot match and are treated like null keys.  WIDE (12 bytes): key64 + row
// in two arrays.  key64 is the exact key for one <= 8-byte column (or several integer
// columns packed into 8 bytes); wider / mixed keys use a 64-bit hash and every hit is
// verified against the original columns.  On the main path (deferred two-level probe side,
// 2^15 partitions, indices only) the PROBE side's tuples shrink: NARROW keys travel as six
// bytes (hash remainder + row: L6 / P6 below), WIDE keys from one 8-byte column as ten
// (the same six bytes + the key's high word: W10 / P10, p10_key) -- exact in both cases,
// because the partition hash is a bijection of the key (NARROW) or of its low word for a
// given high word (WIDE).
//
// Semantics kept from the reference: rows with a null in any key column never match
// (join_kernels.cuh:58-66,314), float keys compare with == (NaN matches nothing), LEFT
// emits (l,-1) for unmatched probe rows, FULL appends (-1,r) for unmatched build rows
// (join_compute_api.h:54-186), INNER builds on the smaller side and flips
// (joining.h:58-66), outputs are library-allocated int32 columns of exactly the joined
// size, pair order unspecified.
#include "internal.h"
#include "gdf/gdf_amd_ext.h"

#include <chrono>
#include <cmath>
#include <cstdio>

#include <algorithm>
#include <cstdlib>
#include <deque>
#include <memory>
#include <type_traits>
#include <vector>

namespace gdf_amd {

// ---------------------------------------------------------------------------
// key construction
// ---------------------------------------------------------------------------
enum KeyMode : int { KM_RAW_INT = 0, KM_RAW_FLOAT, KM_PACKED, KM_HASHED };

struct KeyPlan {
  int mode;
  int verify;                 // 1: key64 is a hash, confirm hits with rows_equal
  int narrow;                 // 1: every joinable key is < 2^32 after subtracting kmin
  uint64_t kmin;              // subtracted from 8-byte integer keys in narrow mode (two's complement)
  uint64_t kspan;             // narrow mode with kmin != 0: the largest stored key (build max - min); decides whether hash_a is a
                              // bijection on the stored keys (six-byte level-2 tuples, p6_store)
  uint64_t klimit;            // the largest stored key that can join: 2^32 - 1 for NARROW keys, kspan once the build range is known
                              // (a probe key beyond the build maximum matches nothing, and the six-byte tuples compare hash
                              // remainders only: hash_a is a bijection on [0, kspan], not beyond it), ~0 for WIDE keys
  uint64_t kwindow;           // range-narrowed keys (else 0): the largest stored key that still lies in the 2^32 window of raw values hash_a
                              // is a bijection on.  An INNER join drops probe rows beyond kspan before they are partitioned (they match nothing,
                              // and half-hit joins run 1 ms faster for it); a LEFT / FULL join lets rows up to kwindow TRAVEL -- the probe kernel
                              // emits their (l, -1) for free, whereas dropped rows come back through the tail kernels (probe_prepared sets klimit)
  int shift[MAX_KEY_COLS];    // KM_PACKED bit offsets
  // KM_PACKED with ranged != 0: column c contributes (value - bias[c]) in bits[c] bits, the ranges taken from the BUILD
  // relation (plan_ranged); a probe value outside its column's range cannot match and makes the row unjoinable
  int ranged;
  int bits[MAX_KEY_COLS];
  long long bias[MAX_KEY_COLS];
};

static KeyPlan plan_keys(const KeyTable &t) {
  KeyPlan p{};
  bool all_int = true;
  int total = 0;
  for (int c = 0; c < t.ncols; ++c) {
    if (t.col[c].kind == K_F32 || t.col[c].kind == K_F64) all_int = false;
    p.shift[c] = total * 8;
    total += t.col[c].width;
  }
  if (t.ncols == 1) p.mode = all_int ? KM_RAW_INT : KM_RAW_FLOAT;
  else if (all_int && total <= 8) p.mode = KM_PACKED;
  else { p.mode = KM_HASHED; p.verify = 1; }
  // zero-extended raw bits of <= 4 bytes are below 2^32 by construction
  if (p.mode != KM_HASHED && total <= 4) p.narrow = 1;
  p.klimit = p.narrow ? 0xffffffffULL : ~0ULL;
  return p;
}

// canonical bits of a float element for equality-by-==: -0.0 -> +0.0; NaN -> not joinable
__device__ __forceinline__ bool float_bits(const ColView &c, int64_t i, uint64_t &bits) {
  if (c.kind == K_F32) {
    uint32_t b = ((const uint32_t *)c.data)[i];
    if ((b & 0x7fffffffu) > 0x7f800000u) return false;
    if ((b << 1) == 0) b = 0;
    bits = b;
  } else {
    uint64_t b = ((const uint64_t *)c.data)[i];
    if ((b & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL) return false;
    if ((b << 1) == 0) b = 0;
    bits = b;
  }
  return true;
}

__device__ __forceinline__ long long load_int_signed(const ColView &c, int64_t i) {
  switch (c.width) {
    case 1: return ((const int8_t *)c.data)[i];
    case 2: return ((const int16_t *)c.data)[i];
    case 4: return ((const int32_t *)c.data)[i];
    default: return ((const long long *)c.data)[i];
  }
}

// returns false when the row cannot match anything (null / NaN key, or outside the build range)
__device__ __forceinline__ bool make_key(const KeyTable &t, const KeyPlan &p, int64_t i, uint64_t &key) {
  if (!row_valid(t, i)) return false;
  switch (p.mode) {
    case KM_RAW_INT: {
      const uint64_t k = load_bits(t.col[0], i) - p.kmin;
      key = k;
      return k <= p.klimit;
    }
    case KM_RAW_FLOAT: return float_bits(t.col[0], i, key);
    case KM_PACKED: {
      uint64_t k = 0;
      if (p.ranged) {
        bool inside = true;
        for (int c = 0; c < t.ncols; ++c) {
          const uint64_t v = (uint64_t)(load_int_signed(t.col[c], i) - p.bias[c]);
          inside = inside && (p.bits[c] >= 64 || (v >> p.bits[c]) == 0);
          k |= v << p.shift[c];
        }
        key = k;
        return inside;
      }
      for (int c = 0; c < t.ncols; ++c) k |= load_bits(t.col[c], i) << p.shift[c];
      key = k;
      return true;
    }
    default: {
      uint64_t h = 0x9e3779b97f4a7c15ULL;
      for (int c = 0; c < t.ncols; ++c) {
        uint64_t b;
        if (t.col[c].kind == K_F32 || t.col[c].kind == K_F64) { if (!float_bits(t.col[c], i, b)) return false; }
        else b = load_bits(t.col[c], i);
        h = mix64(h ^ b) + 0x9e3779b97f4a7c15ULL * (uint64_t)(c + 1);
      }
      key = h;
      return true;
    }
  }
}

// FAST = 8 / 4: one 8-byte (the BASELINE configuration) / 4-byte integer column without a mask: the
// kernels read the column words directly (4-byte keys are zero-extended raw bits, always NARROW).  FAST = 0: generic.  All kernels below first issue a BATCH of
// independent loads, then consume them: with one dependent load per loop trip a wave has
// 512 B in flight and the kernels are latency-bound at ~25 % of HBM bandwidth
// (profiles/r1_a_kernel_stats.md).
template <int FAST>
__device__ __forceinline__ uint64_t fast_word(const void *col, int64_t i) {
  return FAST == 4 ? (uint64_t)((const uint32_t *)col)[i] : ((const uint64_t *)col)[i];
}
// two consecutive key words with one load (8-byte keys: 16 bytes, 4-byte keys: 8 bytes); only the element alignment
// is promised (a column may be a slice of a larger buffer)
struct __attribute__((packed, aligned(8))) KeyPair8 { uint64_t a, b; };
struct __attribute__((packed, aligned(4))) KeyPair4 { uint32_t a, b; };
template <int FAST>
__device__ __forceinline__ void fast_pair(const void *col, int64_t i, uint64_t &a, uint64_t &b) {
  // read-once data: non-temporal, so that the key stream does not push the partially written lines of the scatter's
  // write fronts out of L2 before the neighbouring run completes them
  if (FAST == 4) {
    const uint32_t *p = (const uint32_t *)col + i;
    a = __builtin_nontemporal_load(p);
    b = __builtin_nontemporal_load(p + 1);
  } else {
    const uint64_t *p = (const uint64_t *)col + i;
    a = __builtin_nontemporal_load(p);
    b = __builtin_nontemporal_load(p + 1);
  }
}
// FAST reads of a float key column (KM_RAW_FLOAT): the raw word becomes the canonical bits of float_bits() --
// -0.0 -> +0.0 -- and the return value says whether it can join at all (NaN cannot)
template <int FAST>
__device__ __forceinline__ bool fast_float_word(uint64_t &w) {
  if (FAST == 4) {
    if (((uint32_t)w & 0x7fffffffu) > 0x7f800000u) return false;
    if ((uint32_t)((uint32_t)w << 1) == 0) w = 0;
  } else {
    if ((w & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL) return false;
    if ((w << 1) == 0) w = 0;
  }
  return true;
}
template <int FAST>
__device__ __forceinline__ bool fetch_key(const KeyTable &t, const KeyPlan &p, int64_t i, uint64_t &key) {
  if (FAST) {
    uint64_t w = fast_word<FAST>(t.col[0].data, i);
    const bool joinable = p.mode != KM_RAW_FLOAT || fast_float_word<FAST>(w);
    const uint64_t k = w - p.kmin;
    key = k;
    return joinable && k <= p.klimit;
  }
  return make_key(t, p, i, key);
}

// Batched key fetch for rows i0 + k * stride (k < N), rows >= end are inactive.  FAST issues N
// UNCONDITIONAL loads from clamped addresses -- a load under `if (i < end)` gets its own basic block and
// its own s_waitcnt vmcnt(0) from hipcc, which serialises the batch (seen in the ISA of jk_hist).
template <int FAST, int N>
__device__ __forceinline__ void fetch_keys(const KeyTable &t, const KeyPlan &p, int64_t i0, int64_t stride, int64_t end,
                                           uint64_t (&key)[N], bool (&ok)[N]) {
  if (FAST) {
    const void *col = t.col[0].data;
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int64_t i = i0 + k * stride;
      key[k] = fast_word<FAST>(col, i < end ? i : end - 1);
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const bool joinable = p.mode != KM_RAW_FLOAT || fast_float_word<FAST>(key[k]);
      key[k] -= p.kmin;
      ok[k] = joinable && (i0 + k * stride < end) && key[k] <= p.klimit;
    }
    // the column's validity mask, paired with the data reads: one byte per row, requested together (a wave's 64 rows of
    // one k share 8 bytes -- one request); workgroup-uniform branch, the unmasked column pays nothing
    if (const uint8_t *valid = t.col[0].valid) {
      uint8_t m[N];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const int64_t i = i0 + k * stride;
        m[k] = valid[(i < end ? i : end - 1) >> 3];
      }
#pragma unroll
      for (int k = 0; k < N; ++k) ok[k] = ok[k] && ((m[k] >> ((i0 + k * stride) & 7)) & 1);
    }
  } else if (p.mode == KM_PACKED && p.ranged) {
    // several integer columns packed by range (plan_ranged): column by column, the N elements of a column requested
    // together from clamped row numbers -- make_key() row by row is one dependent load at a time
    int64_t row[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int64_t i = i0 + k * stride;
      ok[k] = i < end;
      row[k] = ok[k] ? i : end - 1;
      key[k] = 0;
    }
    for (int c = 0; c < t.ncols; ++c) {
      long long v[N];
      const void *data = t.col[c].data;
      switch (t.col[c].width) {
        case 1:
#pragma unroll
          for (int k = 0; k < N; ++k) v[k] = ((const int8_t *)data)[row[k]];
          break;
        case 2:
#pragma unroll
          for (int k = 0; k < N; ++k) v[k] = ((const int16_t *)data)[row[k]];
          break;
        case 4:
#pragma unroll
          for (int k = 0; k < N; ++k) v[k] = ((const int32_t *)data)[row[k]];
          break;
        default:
#pragma unroll
          for (int k = 0; k < N; ++k) v[k] = ((const long long *)data)[row[k]];
      }
      const int bits = p.bits[c], shift = p.shift[c];
      const long long bias = p.bias[c];
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const uint64_t u = (uint64_t)(v[k] - bias);
        ok[k] = ok[k] && (bits >= 64 || (u >> bits) == 0);
        key[k] |= u << shift;
      }
      if (t.col[c].valid) {
        uint8_t m[N];
#pragma unroll
        for (int k = 0; k < N; ++k) m[k] = t.col[c].valid[row[k] >> 3];
#pragma unroll
        for (int k = 0; k < N; ++k) ok[k] = ok[k] && ((m[k] >> (row[k] & 7)) & 1);
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const int64_t i = i0 + k * stride;
      ok[k] = i < end;
      key[k] = 0;
      if (ok[k]) ok[k] = make_key(t, p, i, key[k]);
    }
  }
}

// ---------------------------------------------------------------------------
// partition geometry and tuple storage
// ---------------------------------------------------------------------------
constexpr int JK_MAX_FB = 15;               // 32768 fine partitions: 128 KiB of LDS counters in jk_hist
constexpr int JK_HIST_THREADS = 1024;
constexpr int JK_HIST_ITEMS = 8;
constexpr int JK_SC_ITEMS = 16;             // tuples per thread per LDS tile; a tile is THREADS * 16 tuples
constexpr int JK_MAX_CHUNKS = 1 << 16;     // level-1 chunks (histogram columns) at most
constexpr int64_t JK_CHUNK_ROWS = 131072;  // rows per level-1 chunk (swept 8k..512k on C3: profiles/r1_c_sweeps.md)
constexpr int JK_PROBE_THREADS = 512;
constexpr int JK_PROBE_BATCH = 4;
constexpr int JK_TARGET_BUILD = 3072;       // build tuples per fine partition the geometry aims at
// largest build partition kept in LDS; larger ones take the global-table path.  (6144 until round 6: the general kernel's WIDE image of a
// partition of 6081 ... 6144 tuples -- 16 bytes per tuple + 2 x 8192 slots + 80 -- is 80 bytes more than a CU's 160 KiB; the launch failed
// with hipErrorInvalidValue.  Rare twice over: plain joins reach the general kernel only with the units whose cuckoo build did not settle.
// Found by tools/stress_join.py's wide keys; tests/test_gpu_join.py::test_wide_keys_largest_lds_partition)
constexpr int JK_MAX_BUILD = 6080;
constexpr uint32_t JK_PROBE_CHUNK = 1u << 17;   // probe tuples per work unit
constexpr int32_t JK_EMPTY = -1;
constexpr uint32_t JK_NOPOS = 0xffffffffu;
constexpr int JK_CUCKOO_MAX_MOVES = 32;
constexpr int JK_ROLE_LEVEL1 = 1, JK_ROLE_LEVEL2 = 2;      // placed blocks (DevBuf::alloc_placed): the probe side's level-1 / level-2 tuples
constexpr int JK_ROLE_OUT_PROBE = 3, JK_ROLE_OUT_BUILD = 4; // ... and the two index columns of a large dense join
constexpr int JK_ROLE_FUSED_LEVEL2 = 5;                     // ... and the level-2 tuples of a fused multi-GPU join's receiver
constexpr int JK_ROLE_LEVEL1_HI = 6;                        // ... and the high words of ten-byte level-1 tuples (WIDE keys, p10_key)
// challengers of the placement tournaments (partition_side_spec, probe_partitioned).  Level 1 has two MODES, about one fresh block in
// five is a fast one (profiles/r5_b_place_trace_*.json): 8 challengers.  Level 2 and the output columns spread over ~10 % without
// modes (r5_e_place_trace_*.json): fewer candidates get most of what there is, and every candidate is 4 - 7 GB of allocator churn.
constexpr int JK_PLACE_DRAWS = 16, JK_PLACE_DRAWS_L2 = 8, JK_PLACE_DRAWS_OUT = 6;

struct PartGeom {
  int fb, b1, b2;        // fine bits = b1 (level 1) + b2 (level 2)
  int nchunks;           // level-1 chunks (one histogram column each)
  int64_t chunk;         // rows per chunk (multiple of the scatter tile)
  int dbg;               // experiment switch (env GDF_JK_SDBG), 0 in production
  uint64_t kbias;        // added back to a stored key before hashing (= KeyPlan::kmin): partition ids and slots are
                         // functions of the RAW key, so the build-side histogram can run before kmin is known
  // SPECULATIVE layout (cap1 != 0): no histogram pass.  Coarse partition c owns tuples [c * cap1, (c+1) * cap1),
  // fine partition f owns [f * cap2, (f+1) * cap2); (tile, bin) runs claim their place with one atomicAdd on
  // the partition's fill counter.  A partition that outgrows its capacity raises *spec_flag and its run is
  // written to the dump area instead (memory safe); the host then repeats the side with the exact layout.
  uint32_t cap1, cap2, dump;
  // SKEWED probe keys (round 4): the speculative layout with PER-PARTITION room instead of one capacity for all -- region r of level 1
  // owns [rstart[r], rstart[r] + rcap[r]), fine partition f owns [fstart[f], fstart[f] + fcap[f]); sized from a 2^22-row sample of the
  // probe keys (probe_prepared / SkewCaps).  Null: the uniform layout (r * cap1, f * cap2).  cap1 / cap2 stay non-zero (the largest
  // capacity): they are what says "speculative" to the kernels.
  const uint32_t *rstart, *rcap, *fstart, *fcap;
  uint32_t *spec_cursor1;   // [2^b1 << xs] fill counters, zero-initialised
  uint32_t *spec_flag;
  // Level 1 of the speculative layout splits every coarse partition into 2^xs REGIONS of cap1 tuples, one per XCD
  // (block b runs on XCD b % 8, MI355X_MICROARCH.md "Workgroup dispatch"): the 512-byte (tile, bin) runs start at
  // arbitrary 8-byte offsets, so neighbouring runs share their first / last 128-byte line.  With one fill counter per
  // partition those neighbours come from different XCDs, whose L2s are not coherent: both write the shared line back
  // partially.  With one counter per (partition, XCD) the neighbours meet in ONE L2, which merges them into whole
  // lines before they reach HBM.  Level 2 walks the regions as 2^(b1+xs) segments.
  int xs;
  // Build relations beyond 2^15 x JK_TARGET_BUILD rows: b3 more partition bits, split off by a THIRD regrouping pass
  // over the level-2 output of both relations (refine_side); fb + b3 bits in all.  Host-side only.
  int b3;
  // added to the row number a level-1 tuple carries: a probe relation that arrives in slices (gdf_amd_join_probe_add)
  // is numbered across the slices
  int32_t row_base;
  // FUSED multi-GPU join (fj_*, gdf_amd_ext.h): `world` ranks share one hash space -- rank = mulhi(hash_a, world), and the
  // LOCAL partition id is taken from the low word of hash_a * world (uniform inside a rank).  0 / 1: single GPU, the id is
  // hash_a's own top bits.
  uint32_t world;
#ifdef GDF_AMD_LAB
  unsigned long long *lab_clock;     // LAB: jk_scatter1 stores the cycles the first / last wave of a workgroup spent in each phase, [chunk][2][8] (knob GDF_JK_CLOCK)
#endif
};
#ifdef GDF_AMD_LAB
#define LAB_PHASE(i)                                                          \
  if (g.lab_clock) {                                                          \
    const unsigned long long lab_now = clock64();                             \
    lab_ph[i] += (uint32_t)(lab_now - lab_prev);                              \
    lab_prev = lab_now;                                                       \
  }
#else
#define LAB_PHASE(i)
#endif

// NARROW: w[i] = key32 << 32 | row, idx unused.  WIDE: w[i] = key64, idx[i] = row.
// pay (NARROW probe sides of a join that materialises result_cols, see PayCarry): pay[i] = the row's PAYLOAD word -- the
// probe relation's non-key column(s) travel through the regroup passes next to their tuple, as a parallel array (the way WIDE
// tuples carry their row numbers), so that the probe kernel writes them out streaming instead of a later gather fetching
// one random 64-byte sector per 8-byte value (C3: 1e9 of them, 55 ms).  Kernels that only need keys (count pass, sample,
// cuckoo build) never touch it.
struct Tuples {
  uint64_t *w;
  int32_t *idx;
  uint64_t *pay;
};
template <bool NARROW>
__device__ __forceinline__ uint64_t tup_key(uint64_t w) { return NARROW ? (w >> 32) : w; }
template <bool NARROW>
__device__ __forceinline__ uint64_t tup_make(uint64_t key, int32_t row) { return NARROW ? ((key << 32) | (uint32_t)row) : key; }

// Internal hashes (never visible through the ABI -- the public row hash is Murmur3_32 in hash.cuh).
// All of them are functions of the RAW key.  32-bit arithmetic on purpose: gfx950 has no 64-bit integer
// multiplier, a 64-bit xorshift-multiply mixer costs ~4x the VALU cycles of this one, and the regroup
// passes are VALU/LDS-bound, not HBM-bound (profiles/r1_e_scatter_ablation.md).
//   key_fold : 64 -> 32 bits (injective on any key range below 2^32, i.e. on NARROW keys)
//   hash_a   : partition id = top fb bits; cuckoo table-0 slot = low bits (fb + log2 H <= 28: disjoint)
//   hash_b   : cuckoo table-1 slot, linear-probing slot, global-table slot
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t key_fold(uint64_t raw_key) {
  return (uint32_t)raw_key ^ ((uint32_t)(raw_key >> 32) * 0x9e3779b1u);
}
// a second, independent 64 -> 32 fold (also injective below 2^32: an odd multiplier permutes the low word).  Keys wider
// than 32 bits collide in ONE fold about n^2 / 2^33 times -- 1e8 keys spread over 2^60: a million pairs and ~9000 triples
// with the same key_fold, which land in the same partition AND, if both cuckoo tables hashed that fold, in the same two
// slots: a triple can never settle, a quarter of C3-sized WIDE joins' units fell back to linear probing
// (profiles/r2_b_bench_shapes.jsonl, c3_wide_keys).  Everything that has to tell keys of one partition apart uses this one.
__device__ __forceinline__ uint32_t key_fold2(uint64_t raw_key) {
  return (uint32_t)(raw_key >> 32) ^ ((uint32_t)raw_key * 0x85ebca6bu);
}
__device__ __forceinline__ uint32_t hash_a(uint64_t raw_key) { return lowbias32(key_fold(raw_key)); }
__device__ __forceinline__ uint32_t hash_b(uint64_t raw_key) { return lowbias32(key_fold2(raw_key) ^ 0x68e31da4u); }

__device__ __forceinline__ uint32_t fine_of(uint64_t raw_key, int fb) {
  return (uint32_t)((uint64_t)hash_a(raw_key) >> (32 - fb));      // 64-bit shift: fb == 0 gives 0 without a branch
}
// the same with the rank remap of a fused multi-GPU join (PartGeom::world)
struct PartGeom;
__device__ __forceinline__ uint32_t local_hash(uint32_t h, uint32_t world) {
  return world > 1 ? (uint32_t)((uint64_t)h * world) : h;
}
// slot of the global-table path (any table size)
__device__ __forceinline__ uint32_t slot_of(uint64_t key, uint32_t nslots) {
  return (uint32_t)(((uint64_t)hash_b(key) * nslots) >> 32);
}

// ---------------------------------------------------------------------------
// 1. histogram: fine histogram (global, LDS-accumulated) + per-chunk coarse histogram
//    H1[c * nchunks + chunk]
// ---------------------------------------------------------------------------
// minmax (may be null): signed min / max of the raw 8-byte keys of the joinable rows, gathered on the
// build side in the same pass so that the narrow tuple format can be decided without another read.
template <int FAST>
__global__ __launch_bounds__(JK_HIST_THREADS) void jk_hist(KeyTable t, KeyPlan plan, PartGeom g,
                                                           uint32_t *__restrict__ fine_hist,
                                                           uint32_t *__restrict__ H1, long long *minmax) {
  long long lo = LLONG_MAX, hi = LLONG_MIN;
  __shared__ long long wg_mm[2];          // this workgroup's key range; minmax[2 b], minmax[2 b + 1] get it (lo > hi: no key), the host merges
  if (threadIdx.x == 0) { wg_mm[0] = LLONG_MAX; wg_mm[1] = LLONG_MIN; }
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t *fine = lds;
  uint32_t *coarse = lds + (1u << g.fb);
  const uint32_t nfine = 1u << g.fb, ncoarse = 1u << g.b1;
  for (uint32_t f = threadIdx.x; f < nfine; f += JK_HIST_THREADS) fine[f] = 0;
  for (int chunk = blockIdx.x; chunk < g.nchunks; chunk += gridDim.x) {
    for (uint32_t c = threadIdx.x; c < ncoarse; c += JK_HIST_THREADS) coarse[c] = 0;
    block_sync();
    const int64_t begin = (int64_t)chunk * g.chunk;
    const int64_t end = begin + g.chunk < t.nrows ? begin + g.chunk : t.nrows;
    for (int64_t base = begin; base < end; base += (int64_t)JK_HIST_THREADS * JK_HIST_ITEMS) {
      uint64_t key[JK_HIST_ITEMS];
      bool ok[JK_HIST_ITEMS];
      fetch_keys<FAST, JK_HIST_ITEMS>(t, plan, base + threadIdx.x, JK_HIST_THREADS, end, key, ok);
#pragma unroll
      for (int k = 0; k < JK_HIST_ITEMS; ++k) {
        if (ok[k]) {
          const uint64_t raw = key[k] + plan.kmin;
          const uint32_t f = fine_of(raw, g.fb);
          atomicAdd(&fine[f], 1u);
          atomicAdd(&coarse[f >> g.b2], 1u);
          lo = (long long)raw < lo ? (long long)raw : lo;
          hi = (long long)raw > hi ? (long long)raw : hi;
        }
      }
    }
    block_sync();
    for (uint32_t c = threadIdx.x; c < ncoarse; c += JK_HIST_THREADS) H1[(size_t)c * g.nchunks + chunk] = coarse[c];
    block_sync();
  }
  for (uint32_t f = threadIdx.x; f < nfine; f += JK_HIST_THREADS)
    if (fine[f]) atomicAdd(&fine_hist[f], fine[f]);
  if (minmax) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const long long l2 = __shfl_xor(lo, o, WAVE), h2 = __shfl_xor(hi, o, WAVE);
      lo = l2 < lo ? l2 : lo;
      hi = h2 > hi ? h2 : hi;
    }
    if (lane_id() == 0 && lo <= hi) { atomicMin(&wg_mm[0], lo); atomicMax(&wg_mm[1], hi); }
    block_sync();
    if (threadIdx.x == 0) { minmax[2 * blockIdx.x] = wg_mm[0]; minmax[2 * blockIdx.x + 1] = wg_mm[1]; }
  }
}

// Skew probe for the histogram-free layout: JK_SKEW_SAMPLES evenly spaced rows of a FAST key column are binned by fine
// partition; *max_count = the fullest bin.  A uniform column puts ~2 samples in a bin; a key that holds 0.1 % of the rows
// puts 60 there.  (Zipf-distributed probe keys made the speculative level-1 pass overflow after ~8 % of the rows -- 1.4 ms
// thrown away per 1e9 rows before the exact layout took over.)
constexpr uint32_t JK_SKEW_SAMPLES = 1u << 16;
template <int FAST>
__global__ __launch_bounds__(256) void jk_sample_skew(const void *col, int64_t nrows, int fb, uint32_t *hist, uint32_t *max_count) {
  const uint32_t s = blockIdx.x * 256 + threadIdx.x;
  if (s >= JK_SKEW_SAMPLES) return;
  const int64_t i = (int64_t)(((unsigned __int128)s * (unsigned __int128)nrows) >> 16);
  const uint32_t f = fine_of(fast_word<FAST>(col, i), fb);
  const uint32_t old = atomicAdd(&hist[f], 1u);
  atomicMax(max_count, old + 1u);
}

// The capacity sample of a SKEWED probe column (SkewCaps): 2^22 evenly spaced rows binned by fine partition, LDS histograms merged with
// one global atomic per touched bin and workgroup.  counts[nfine] = rows sampled.
constexpr uint32_t JK_CAPS_SAMPLES = 1u << 22;
template <int FAST>
__global__ __launch_bounds__(1024) void jk_sample_caps(const void *col, int64_t nrows, int fb, uint32_t *__restrict__ counts) {
  extern __shared__ __attribute__((aligned(16))) uint32_t caps_lds[];
  const uint32_t nfine = 1u << fb;
  for (uint32_t f = threadIdx.x; f < nfine; f += 1024) caps_lds[f] = 0;
  block_sync();
  const uint32_t per_wg = JK_CAPS_SAMPLES / gridDim.x;          // a multiple of 8 * 1024 (the launch uses 256 workgroups)
  for (uint32_t j0 = threadIdx.x; j0 < per_wg; j0 += 8 * 1024) {
    uint64_t w[8];                                               // eight independent loads in flight per thread, then the atomics
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t sidx = blockIdx.x * per_wg + j0 + k * 1024;
      w[k] = fast_word<FAST>(col, (int64_t)(((unsigned __int128)sidx * (unsigned __int128)nrows) >> 22));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) atomicAdd(&caps_lds[fine_of(w[k], fb)], 1u);
  }
  block_sync();
  for (uint32_t f = threadIdx.x; f < nfine; f += 1024)
    if (caps_lds[f]) atomicAdd(&counts[f], caps_lds[f]);
  if (threadIdx.x == 0) atomicAdd(&counts[nfine], per_wg);
}
// what the host made of it: sampled rows per fine partition (the device arrays depend on the level-1 region split and are made by
// partition_side_spec)
struct SkewCaps {
  std::vector<uint32_t> fine;      // [nfine] sampled rows
  double rows_per_sample = 0;      // probe rows / sampled rows
};

// ---------------------------------------------------------------------------
// LDS tile regroup shared by both scatter levels.
//   phase A: every thread holds up to ITEMS tuples with bin + rank (rank from an
//            LDS atomic on hist[bin]);
//   phase B: exclusive scan of hist -> start; the global base of every bin is
//            fixed; tuples are written to LDS at start[bin] + rank;
//   phase C: LDS position j goes to global base[bin(j)] + (j - start[bin(j)]), so a
//            wave writes runs of consecutive addresses.
// ---------------------------------------------------------------------------
// PAY: payload words per tuple -- 0, 1 (PayCarry modes 1 - 3) or 2 (mode 4: two 8-byte columns, one 16-byte element)
template <bool NARROW, int THREADS, int PAY = 0, int ITEMS = JK_SC_ITEMS>
struct TileLds {
  // + a trash slot: tuples that do not travel are written there.  The 1024-thread level-1 tile also has room for the padding of
  // six-byte tuples (L6: every bin's run is padded to an even length, up to 256 dead tuples per tile)
  // (the WIDE 1024-thread tile is the ten-byte one, W10: 12 tuples per thread, padded the same way)
  static constexpr int PAD = (THREADS == 1024 && !PAY) ? 256 : 0;
  uint64_t w[THREADS * ITEMS + PAD + 2];
  int32_t idx[NARROW ? 4 : THREADS * ITEMS + PAD + 4];  // WIDE: the row numbers; W10 / P10 tiles: the key's high words
  alignas(16) uint64_t pay[PAY ? (THREADS * ITEMS + 2) * PAY : 2];      // payload words (PAY per tuple), regrouped with their tuples
  uint32_t hist[256 + 64];                                // + 64 trash counters, one per lane (never zeroed, never read): a probe relation with
                                                          // most of its keys outside the build range put 90 % of a tile's LDS atomics on ONE of them
  uint32_t start[256];
  uint32_t gbase[256];    // (global base - start[bin]) mod 2^32; destinations are < 2^31
  uint32_t wave_tot[THREADS / WAVE];
  uint32_t total;
  uint32_t total_abort;   // level 1, speculative layout: the overflow flag as thread 0 saw it during this tile
  // level 1 only, and LAST: a level-2 launch that is short of LDS asks for the bytes in front of them (level2_lds_bytes)
  uint32_t cursor[256];   // level 1: running global cursor of this chunk
  uint32_t odd[256];      // level 1, six-byte tuples (L6): the bin's run of this tile was padded to an even length
};

// block-wide exclusive scan of hist[0..nbins) (nbins <= 256 == blockDim) into start[]
template <bool NARROW, int THREADS, int PAY, int ITEMS>
__device__ __forceinline__ void tile_scan_bins(TileLds<NARROW, THREADS, PAY, ITEMS> &s, uint32_t nbins, uint32_t tid) {
  const uint32_t v = tid < nbins ? s.hist[tid] : 0;
  const uint32_t incl = wave_scan_incl(v);
  if (lane_id() == WAVE - 1) s.wave_tot[tid / WAVE] = incl;
  block_sync();
  uint32_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < THREADS / WAVE; ++w) {
    if (w < (int)(tid / WAVE)) woff += s.wave_tot[w];
    tot += s.wave_tot[w];
  }
  if (tid < nbins) s.start[tid] = woff + incl - v;
  if (tid == 0) s.total = tot;
}
template <bool NARROW, int THREADS, int PAY, int ITEMS>
__device__ __forceinline__ void tile_scan_bins(TileLds<NARROW, THREADS, PAY, ITEMS> &s, uint32_t nbins) {
  tile_scan_bins(s, nbins, threadIdx.x);
}
// threadIdx.x through an opaque move: addresses derived from the result cannot be hoisted out of the enclosing loop
// (hipcc hoists a dozen per-thread LDS addresses out of jk_scatter1's tile loop and then spills them; every reload is
// a scratch load, i.e. an s_waitcnt vmcnt(0) that also waits for the tile's stores)
__device__ __forceinline__ uint32_t opaque_tid() {
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  return tid;
}

// SIX-BYTE level-2 tuples (P6).  hash_a is a bijection on NARROW keys whose raw values do not straddle a 2^32 boundary
// (key_fold is then `low word ^ constant`, lowbias32 permutes 32 bits), and the fine partition IS the top fb bits of that hash:
// inside a partition a tuple is identified by the REMAINING 32 - fb bits.  With fb = 15 that is 17 bits; with a 31-bit row
// number a tuple fits 48 bits -- the level-2 output and the probe input shrink from 8 to 6 bytes per tuple (4 of the join's
// 48 bytes of HBM traffic per probe row).  Layout: bits 0..30 row, bits 31..47 hash remainder r; stored as a 4-byte and a
// 2-byte store at byte 6 * position, read back two tuples at a time as three dwords.  The probe compares remainders instead of
// keys: its LDS image of the build partition gets the keys replaced by their remainders when it is staged (p6_remainder), and
// equal (partition, remainder) means equal hash means equal key.  Indices-only INNER / LEFT joins on the main (deferred,
// two-level) path only: whoever needs the key VALUE back (carried key column) or other layouts keeps 8-byte tuples.
// (world > 1, a fused multi-GPU join: the partition id comes from the low word L of hash * world = rank * 2^32 + L.  An ODD world
// makes hash -> L a bijection of 32-bit words already.  A power-of-two world 2^k shifts the hash: L's low k bits are empty and
// the owner rank -- the k bits shifted out -- goes there, so the remainder still determines the hash whatever ranks' keys a
// receive buffer holds (the one-GPU emulations feed a sender's WHOLE buffer back).  Other worlds keep 8-byte tuples.)
__host__ __device__ __forceinline__ bool p6_world_ok(uint32_t world) { return world <= 1 || (world & 1u) || !(world & (world - 1u)); }
__device__ __forceinline__ uint32_t p6_low(uint32_t hash, uint32_t local, uint32_t world) {
  return (world > 1 && !(world & (world - 1u))) ? __builtin_rotateleft32(hash, 31 - __clz((int)world)) : local;     // = local | mulhi(hash, world)
}
__device__ __forceinline__ uint32_t p6_remainder(uint32_t key32, uint64_t kbias, int fb, uint32_t world) {
  const uint32_t q = hash_a((uint64_t)key32 + kbias);
  return p6_low(q, local_hash(q, world), world) & ((1u << (32 - fb)) - 1u);
}
__device__ __forceinline__ void p6_store(uint64_t *base, uint32_t pos, uint32_t r, uint32_t row) {
  unsigned char *at = reinterpret_cast<unsigned char *>(base) + (size_t)pos * 6u;
  const uint32_t lo = (row & 0x7fffffffu) | (r << 31);
  const uint16_t hi = (uint16_t)(r >> 1);
  __builtin_memcpy(at, &lo, 4);                    // 2-byte aligned: gfx950 global stores need no alignment
  __builtin_memcpy(at + 4, &hi, 2);
}
// tuples v and v + 1 (v even) of a P6 stream that starts at `base`: (remainder, row) each -- one 12-byte load
__device__ __forceinline__ void p6_load_pair(const uint64_t *base, uint32_t v, uint32_t &r0, uint32_t &row0, uint32_t &r1, uint32_t &row1) {
  const uint32_t *at = reinterpret_cast<const uint32_t *>(base) + (size_t)(v >> 1) * 3u;
  struct __attribute__((packed, aligned(4))) D3 { uint32_t a, b, c; };
  const D3 d = *reinterpret_cast<const D3 *>(at);
  row0 = d.a & 0x7fffffffu;
  r0 = (d.a >> 31) | ((d.b & 0xffffu) << 1);
  const uint32_t lo1 = (d.b >> 16) | (d.c << 16);
  row1 = lo1 & 0x7fffffffu;
  r1 = (lo1 >> 31) | ((d.c >> 16) << 1);
}

// SIX-BYTE LEVEL-1 tuples (L6, round 4).  hash_a is a bijection on the stored keys (the P6 condition) and after level 1 its top
// b1 = 8 bits are the tuple's place, so 24 bits identify the key; the row number gives up the bits its place implies as well:
// level 1 of the speculative layout splits every coarse partition into 64 REGIONS by chunk number (PartGeom::xs = 6: region =
// chunk % 64, chunks are 2^17 rows, chunk c runs on XCD c % 8 -- the per-XCD merging of write fronts is unchanged), so a tuple's
// row bits 17..22 ARE its region.  24 hash bits + (17 low + 7 high) explicit row bits = 48: a level-1 tuple is 6 bytes for up to
// 2^30 rows, 40 instead of 44 bytes of HBM traffic per probe row (write 8 -> 6 here, read 8 -> 6 at level 2).
// Layout of a tuple: bits 0..23 row24 = row[0..16] | row[23..29] << 17, bits 24..47 the hash remainder.  Tuples leave in PAIRS,
// one aligned 12-byte store per lane (the (tile, bin) runs are padded to even lengths with a dead tuple -- all ones, which no
// live tuple can be while rows stay below 2^30 - 2^23), and level 2 reads them back as pairs (jk_scatter2<IN6>).
constexpr uint32_t L6_ROW_MASK = 0xffffffu;
__device__ __forceinline__ uint32_t l6_row24(uint32_t row) { return (row & 0x1ffffu) | ((row >> 23) << 17); }
__device__ __forceinline__ uint32_t l6_row(uint32_t row24, uint32_t region) { return (row24 & 0x1ffffu) | (region << 17) | ((row24 >> 17) << 23); }
struct __attribute__((packed, aligned(4))) L6Pair { uint32_t a, b, c; };
// (hash remainder, row24) x 2 -> 12 bytes
__device__ __forceinline__ L6Pair l6_pack(uint32_t rem0, uint32_t row0, uint32_t rem1, uint32_t row1) {
  const uint32_t lo0 = row0 | (rem0 << 24), lo1 = row1 | (rem1 << 24);
  return L6Pair{lo0, (rem0 >> 8) | (lo1 << 16), (lo1 >> 16) | ((rem1 >> 8) << 16)};
}
__device__ __forceinline__ void l6_unpack(const L6Pair &d, uint32_t &rem0, uint32_t &row0, uint32_t &rem1, uint32_t &row1) {
  row0 = d.a & L6_ROW_MASK;
  rem0 = (d.a >> 24) | ((d.b & 0xffffu) << 8);
  const uint32_t lo1 = (d.b >> 16) | (d.c << 16);
  row1 = lo1 & L6_ROW_MASK;
  rem1 = (lo1 >> 24) | ((d.c >> 16) << 8);
}

// TEN-BYTE tuples for WIDE keys (W10 at level 1, P10 at level 2; round 6).  A key that does not fit 32 bits used to travel as
// key64 + row, 12 bytes in two arrays, on 8192-tuple tiles with ONE 512-thread workgroup per CU (the 16384 x 12-byte tile does not
// fit LDS): 15.1 ms for C3 with keys spread over 2^60 against 9.0 ms on NARROW keys.  The six-byte machinery above carries over
// because hash_a needs no help to become a bijection of 64-bit keys: key_fold(raw) = lo ^ hi * C is, for a FIXED high word, a
// permutation of the low word, so (hash_a(raw), hi) determines (lo, hi) -- equal partition + equal hash remainder + equal high word
// <=> equal key, exactly (reference semantics: a pair needs equal keys, join_kernels.cuh:259-455).  A WIDE tuple on the main path is
// therefore the NARROW six-byte tuple of its hash (l6_pack / p6_store, unchanged) plus the key's high word in a parallel array
// (Tuples::idx, at the same tuple position): 6 + 4 bytes at both levels, 56 instead of 64 bytes of HBM traffic per probe row, and --
// what matters more -- the level-1 tile is 12288 tuples of 12 bytes in LDS (hash word | row24, high word) under ONE 1024-thread
// workgroup, the shape the NARROW kernel is tuned on.  The build side keeps its 12-byte tuples: the probe kernels turn a staged
// build key into (remainder << 32 | high word) the way P6 turns it into its remainder (p10_key).  In the kernels this is simply
// "L6 / IN6 / P6 with NARROW = false".
__device__ __forceinline__ uint64_t p10_key(uint64_t raw_key, int fb) {
  return ((uint64_t)(hash_a(raw_key) & ((1u << (32 - fb)) - 1u)) << 32) | (uint32_t)(raw_key >> 32);
}
struct __attribute__((packed, aligned(4))) HiPair { uint32_t a, b; };      // the high words of tuples v, v + 1: one 8-byte access at any 4-byte offset
// one tuple of a P6 stream (the general kernels' WIDE loads: a 4-byte and a 2-byte load)
__device__ __forceinline__ void p6_load_one(const uint64_t *base, uint32_t pos, uint32_t &r, uint32_t &row) {
  const unsigned char *at = reinterpret_cast<const unsigned char *>(base) + (size_t)pos * 6u;
  uint32_t lo;
  uint16_t hi;
  __builtin_memcpy(&lo, at, 4);
  __builtin_memcpy(&hi, at + 4, 2);
  row = lo & 0x7fffffffu;
  r = (lo >> 31) | ((uint32_t)hi << 1);
}

// phase C: write the regrouped tile out.  LEVEL1 bins are the coarse id, LEVEL2 the sub id.
template <bool LEVEL1, bool NARROW, int THREADS, int PAY, int ITEMS, bool P6 = false>
__device__ __forceinline__ void tile_flush(TileLds<NARROW, THREADS, PAY, ITEMS> &s, const PartGeom &g, Tuples out) {
  const uint32_t total = s.total;
  const uint32_t submask = (1u << g.b2) - 1;
  if constexpr (P6) {
    // six-byte tuples leave in PAIRS: a lane takes LDS positions 2 i, 2 i + 1 (one 16-byte LDS read); inside a (tile, bin) run their
    // destinations are neighbours, and the two tuples go out as ONE 12-byte store instead of two 4-byte + two 2-byte ones (a wave
    // writes 768 contiguous bytes per instruction); only a pair that straddles a run boundary is stored tuple by tuple
    constexpr int UP = 2;
    const uint32_t rmask = (1u << (32 - g.fb)) - 1u;
    for (uint32_t i0 = threadIdx.x; 2 * i0 < total; i0 += THREADS * UP) {
      ulonglong2 ww[UP];
      uint2 hh[UP];                                    // WIDE (P10): the two tuples' high words, in step with them
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const uint32_t j = 2 * (i0 + u * THREADS);
        ww[u] = j < total ? *reinterpret_cast<const ulonglong2 *>(&s.w[j]) : ulonglong2{0, 0};      // (the slot behind an odd total is the tile's own, its content unused)
        hh[u] = (!NARROW && j < total) ? *reinterpret_cast<const uint2 *>(&s.idx[j]) : uint2{0, 0};
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const uint32_t j = 2 * (i0 + u * THREADS);
        if (j >= total) continue;
        const uint32_t h0 = (uint32_t)(ww[u].x >> 32), h1 = (uint32_t)(ww[u].y >> 32);      // jk_scatter2 staged the hash word (p6_low) in place of the key
        const uint32_t bin0 = (uint32_t)((uint64_t)h0 >> (32 - g.fb)) & submask, bin1 = (uint32_t)((uint64_t)h1 >> (32 - g.fb)) & submask;
        const uint32_t dst0 = s.gbase[bin0] + j, dst1 = s.gbase[bin1] + j + 1;
        const uint32_t r0 = h0 & rmask, r1 = h1 & rmask;
        if (j + 1 < total && dst1 == dst0 + 1) {
          const uint32_t lo0 = ((uint32_t)ww[u].x & 0x7fffffffu) | (r0 << 31), lo1 = ((uint32_t)ww[u].y & 0x7fffffffu) | (r1 << 31);
          struct __attribute__((packed, aligned(2))) D3 { uint32_t a, b, c; };
          D3 d{lo0, (r0 >> 1) | (lo1 << 16), (lo1 >> 16) | ((r1 >> 1) << 16)};
          *reinterpret_cast<D3 *>(reinterpret_cast<unsigned char *>(out.w) + (size_t)dst0 * 6u) = d;
          if constexpr (!NARROW) *reinterpret_cast<HiPair *>(out.idx + dst0) = HiPair{hh[u].x, hh[u].y};
        } else {
          p6_store(out.w, dst0, r0, (uint32_t)ww[u].x);
          if constexpr (!NARROW) out.idx[dst0] = (int32_t)hh[u].x;
          if (j + 1 < total) {
            p6_store(out.w, dst1, r1, (uint32_t)ww[u].y);
            if constexpr (!NARROW) out.idx[dst1] = (int32_t)hh[u].y;
          }
        }
      }
    }
    return;
  }
  constexpr int U = 4;
  for (uint32_t j0 = threadIdx.x; j0 < total; j0 += THREADS * U) {
    uint64_t ww[U], pp[U], pq[PAY == 2 ? U : 1];
    int32_t ii[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t j = j0 + u * THREADS;
      ww[u] = j < total ? s.w[j] : 0;
      ii[u] = (!NARROW && j < total) ? s.idx[j] : 0;
      if constexpr (PAY == 2) {             // two payload words: one 16-byte element
        const ulonglong2 e = j < total ? *reinterpret_cast<const ulonglong2 *>(&s.pay[2 * (size_t)j]) : ulonglong2{0, 0};
        pp[u] = e.x; pq[u] = e.y;
      } else {
        pp[u] = (PAY && j < total) ? s.pay[j] : 0;
      }
    }
    uint32_t dst[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t q = hash_a(tup_key<NARROW>(ww[u]) + g.kbias);
      const uint32_t h = local_hash(q, g.world);
      const uint32_t f = (uint32_t)((uint64_t)h >> (32 - g.fb));
      const uint32_t bin = LEVEL1 ? (f >> g.b2) : (f & submask);
      dst[u] = s.gbase[bin] + j0 + u * THREADS;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (LAB_BITS(g.dbg) & 4) dst[u] &= 0xffffu;        // experiment: all stores land in a 512 KiB window
      if (j0 + u * THREADS < total && !(LAB_BITS(g.dbg) & 1)) {
        if (LAB_BITS(g.dbg) & 16) __builtin_nontemporal_store(ww[u], out.w + dst[u]);      // experiment: streaming stores (level 2)
        else out.w[dst[u]] = ww[u];
        if (!NARROW) out.idx[dst[u]] = ii[u];
        if constexpr (PAY == 2) *reinterpret_cast<ulonglong2 *>(out.pay + 2 * (size_t)dst[u]) = ulonglong2{pp[u], pq[u]};
        else if (PAY) out.pay[dst[u]] = pp[u];
      }
    }
  }
}

// 2. level-1 scatter: raw key columns -> tuples grouped by coarse partition
// MASKED (FAST only): the key column carries a validity mask.  The mask is read PAIRED with the data: per tile one 4-byte
// word per lane (lanes 0..31 of a wave hold the 128 bytes that cover the wave's 16 rows per lane; all from at most two
// cache lines), prefetched with the key words of the next tile and spread to the lanes by ds_bpermute when the keys are
// consumed -- a null row is ranked on the trash counter like a row outside the narrow range.  No alignment is asked of
// the mask pointer and nothing is read behind its ceil(rows / 8) bytes (load_mask_word).
__device__ __forceinline__ uint32_t load_mask_word(const uint8_t *valid, uint32_t word_index, uint32_t mask_bytes) {
  // branch-free (a load under a branch is a basic block of its own with an s_waitcnt vmcnt(0), see fetch_keys): the word
  // that would reach past the mask is taken from its last four bytes and shifted down; mask_bytes >= 4
  const uint32_t at = word_index * 4u;
  const uint32_t from = at + 4u <= mask_bytes ? at : mask_bytes - 4u;
  uint32_t w;
  __builtin_memcpy(&w, valid + from, 4);               // gfx950 global loads need no alignment
  const uint32_t drop = (at - from) * 8u;                // bits of earlier bytes in front of ours; >= 32: every row is behind the end
  return drop < 32u ? w >> drop : 0u;
}
// tuples per thread of a level-1 tile: 16, and 12 where the tile holds 12 bytes per tuple under 1024 threads (WIDE keys)
__host__ __device__ constexpr int sc1_items(bool narrow, int threads) { return (!narrow && threads == 1024) ? 12 : JK_SC_ITEMS; }
template <int FAST, bool NARROW, int THREADS, bool MASKED = false, bool L6 = false>
__global__ __launch_bounds__(THREADS) void jk_scatter1(KeyTable t, KeyPlan plan, PartGeom g,
                                                             const uint32_t *__restrict__ H1off,   // scanned H1
                                                             Tuples out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  // L6 with WIDE keys = the TEN-BYTE tuples (W10, see p10_key): six bytes of (hash remainder, row24) as for NARROW keys + the key's
  // high word in out.idx; twelve tuples per thread (12 bytes per tuple in LDS)
  // (every WIDE 1024-thread tile holds twelve tuples per thread: the build side's (key64, row) tuples take the same 12 bytes in LDS)
  constexpr bool W10 = L6 && !NARROW;
  static_assert(!W10 || (FAST == 8 && THREADS == 1024), "ten-byte tuples: an 8-byte key column on the 1024-thread tile");
  constexpr int ITEMS = sc1_items(NARROW, THREADS);
  using Tile = TileLds<NARROW, THREADS, false, ITEMS>;
  Tile &s = *reinterpret_cast<Tile *>(tile_raw);
  constexpr int JK_TILE = THREADS * ITEMS;
  const int chunk = blockIdx.x;
  const uint32_t ncoarse = 1u << g.b1;
  if (!g.cap1 && threadIdx.x < ncoarse) s.cursor[threadIdx.x] = H1off[(size_t)threadIdx.x * g.nchunks + chunk];
  // row numbers are below 2^31 (positions are int32 in the ABI): 32-bit arithmetic throughout, half the registers
  const uint32_t begin = (uint32_t)((int64_t)chunk * g.chunk);
  const uint32_t end = (int64_t)begin + g.chunk < t.nrows ? (uint32_t)(begin + g.chunk) : (uint32_t)t.nrows;
  // FAST: the raw column words of the NEXT tile are requested while the current tile is flushed, so the
  // HBM read latency hides behind the LDS regroup + store phase (one workgroup per CU: nothing else would)
  // Row of item k of this thread within a tile.  FAST: thread t owns the row PAIRS t, t + THREADS, ... (one 16- / 8-byte
  // load per pair: half the address arithmetic and load instructions of one load per row); generic: rows t + k * THREADS.
  // (tid comes from opaque_tid() inside the tile loop: sixteen hoisted row numbers are sixteen registers)
  auto item_row = [](int k, uint32_t tid) -> uint32_t {
    return FAST ? 2u * ((uint32_t)(k >> 1) * THREADS + tid) + (k & 1) : (uint32_t)k * THREADS + tid;
  };
  uint64_t nxt[ITEMS];
  uint32_t nxtmask = 0;                          // MASKED: this lane's word of the tile's validity bits (see consume)
  const void *col = t.col[0].data;
  const uint8_t *vmask = t.col[0].valid;
  const uint32_t vbytes = (uint32_t)((t.nrows + 7) >> 3);
  auto prefetch = [&](uint32_t tile) {           // a pair that would cross `end` is read from the last two rows instead:
    const uint32_t tid = opaque_tid();           // its first row, if it is row end - 1, is then the SECOND word loaded (consume)
#pragma unroll
    for (int k = 0; k < ITEMS; k += 2) {
      const uint32_t i = tile + item_row(k, tid);
      fast_pair<FAST>(col, (int64_t)(i + 2 <= end ? i : end - 2), nxt[k], nxt[k + 1]);
    }
    if (MASKED) {
      // pair j of thread (wave w, lane L) is rows tile + 2 * (j * THREADS + 64 w + L) + {0, 1}: bit 2 L of the 128-bit group
      // (j, w), i.e. of mask word (tile / 32) + j * THREADS / 16 + 4 w + L / 16.  Lane l (and l + 32) fetches word
      // (j = l / 4 % 8, q = l % 4) of its wave; tile is a multiple of the largest tile, so of 32
      const uint32_t l = tid & 31u, w = tid >> 6;
      nxtmask = load_mask_word(vmask, (tile >> 5) + (uint32_t)(THREADS / 16) * (l >> 2) + 4u * w + (l & 3u), vbytes);
    }
  };
  if (FAST) prefetch(begin);
  if (threadIdx.x < 256) s.hist[threadIdx.x] = 0;
  block_sync();
  // Per tile: rank (LDS atomics) | scan | claim + regroup in LDS | flush.  Four barriers; the waves that finish their
  // share of a flush go on to rank the next tile (they touch only hist, which the flush does not read).
  // Everything between two barriers is straight-line code: a `if (ok) atomicAdd` per item compiled to sixteen
  // branches with an s_waitcnt lgkmcnt(0) each (one LDS round trip at a time), and a flush loop with a dynamic trip
  // count made the compiler wait for vmcnt(0) -- i.e. for the completion of the previous tile's STORES -- before the
  // prefetched keys could be used (gfx9 counts loads and stores in the one in-order vmcnt).  Rows that do not travel
  // (beyond the chunk, null, outside the narrow range) go to a trash counter / LDS slot / global dump slot instead.
  using KeyReg = typename std::conditional<NARROW, uint32_t, uint64_t>::type;    // NARROW: joinable keys are < 2^32
  const uint32_t l6_low = (uint32_t)g.kbias, l6_fold = (uint32_t)(g.kbias >> 32) * 0x9e3779b1u;
  KeyReg key[ITEMS];
  uint32_t okmask = 0;          // bit k: item k travels.  One VGPR; sixteen loop-carried bools cost 32 SGPRs and spills
  auto consume = [&](uint32_t tile) {
    const uint32_t tid = opaque_tid();
    okmask = 0;
    uint32_t validbits = 0xffffffffu;            // bit k: row of item k is not null
    if (MASKED) {
      validbits = 0;
      const uint32_t L = tid & 63u;
#pragma unroll
      for (int j = 0; j < ITEMS / 2; ++j) {
        const uint32_t wj = (uint32_t)__shfl((int)nxtmask, (int)(4u * j + (L >> 4)), WAVE);
        validbits |= ((wj >> ((2u * L) & 31u)) & 3u) << (2 * j);
      }
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      uint64_t raw = ((k & 1) == 0 && tile + item_row(k, tid) + 1 == end) ? nxt[k + 1] : nxt[k];
      const bool joinable = plan.mode != KM_RAW_FLOAT || fast_float_word<FAST ? FAST : 8>(raw);
      const uint64_t k64 = raw - plan.kmin;
      key[k] = (KeyReg)k64;
      okmask |= (uint32_t)(joinable && (tile + item_row(k, tid) < end) && k64 <= plan.klimit) << k;
    }
    okmask &= validbits;
  };
  if (FAST) consume(begin);
#ifdef GDF_AMD_LAB
  uint32_t lab_ph[6] = {0, 0, 0, 0, 0, 0};
  unsigned long long lab_prev = g.lab_clock ? clock64() : 0;
#endif
  for (uint32_t tile = begin; tile < end; tile += JK_TILE) {        // end + JK_TILE < 2^32
    if (!FAST) {
      uint64_t k64[ITEMS];
      bool ok[ITEMS];
      fetch_keys<FAST, ITEMS>(t, plan, (int64_t)tile + threadIdx.x, THREADS, (int64_t)end, k64, ok);   // all loads first
      okmask = 0;
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) { key[k] = (KeyReg)k64[k]; okmask |= (uint32_t)ok[k] << k; }
    }
    uint32_t binrank[ITEMS];             // bin << 16 | rank within (tile, bin); bin 256 = does not travel
#pragma unroll
    for (int h = 0; h < ITEMS; h += 4) {
#pragma unroll
      for (int k = h; k < h + 4; ++k) {
        if constexpr (W10) {
          // the tuple carries hash_a of its raw key and the key's high word from here on: together they determine the key (p10_key)
          const uint64_t raw = (uint64_t)key[k] + g.kbias;
          const uint32_t q = hash_a(raw);
          key[k] = (KeyReg)(((raw >> 32) << 32) | q);
          binrank[k] = (okmask >> k) & 1u ? q >> (32 - g.b1) : 256u + (threadIdx.x & 63u);
        } else if constexpr (L6) {
          // the tuple carries its HASH from here on (a bijection of the key, see L6 above): the flush does not hash again
          // (L6 keys do not straddle a 2^32 boundary of raw values: the high word of key + kbias is kbias's own, key_fold's multiply
          // a constant -- one add and one xor instead of an add-with-carry and a quarter-rate multiply; a row that does not travel
          // may hash to anything)
          const uint32_t q = lowbias32(((uint32_t)key[k] + l6_low) ^ l6_fold);
          key[k] = (KeyReg)q;
          binrank[k] = (okmask >> k) & 1u ? q >> (32 - g.b1) : 256u + (threadIdx.x & 63u);
        } else {
          const uint32_t b = fine_of((uint64_t)key[k] + g.kbias, g.fb) >> g.b2;     // hashed whether it travels or not: no branch
          binrank[k] = (okmask >> k) & 1u ? b : 256u + (threadIdx.x & 63u);
        }
      }
      __builtin_amdgcn_sched_barrier(0);       // four hashes at a time: sixteen interleaved ones spill
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)      // sixteen atomics in flight, one wait
      binrank[k] = (binrank[k] << 16) | atomicAdd(&s.hist[binrank[k]], 1u);
    block_sync();
    LAB_PHASE(0)       // hash + rank
    uint32_t claimed = 0;
    {
      const uint32_t tid = opaque_tid();
      // speculative layout: the claim (a returning global atomic, ~2 us) is issued as soon as the counts are final and
      // collected after the regroup, instead of sitting between two barriers on its own
      uint32_t base = 0;
      if (g.cap1 && tid < ncoarse) {
        uint32_t cnt = s.hist[tid];
        if constexpr (L6) {
          // runs of even length: a pair never straddles two bins (the odd one out is padded with a dead tuple below)
          s.odd[tid] = cnt & 1u;
          cnt = (cnt + 1u) & ~1u;
          s.hist[tid] = cnt;
        }
        if (cnt) base = atomicAdd(&g.spec_cursor1[(tid << g.xs) | (blockIdx.x & ((1u << g.xs) - 1u))], cnt);
      }
      tile_scan_bins(s, ncoarse, tid);
      block_sync();
      claimed = base;
    }
    LAB_PHASE(1)       // claim issue + scan
    const uint32_t wtid = opaque_tid();
    constexpr int RG = ITEMS % 8 == 0 ? 8 : 6;
#pragma unroll
    for (int h = 0; h < ITEMS; h += RG) {           // eight at a time: reads of start[] first (no branch), then the writes
      uint32_t st[RG];
#pragma unroll
      for (int k = 0; k < RG; ++k) st[k] = s.start[(binrank[h + k] >> 16) & 255u];
#pragma unroll
      for (int k = 0; k < RG; ++k) {
        const uint32_t pos = (okmask >> (h + k)) & 1u ? st[k] + (binrank[h + k] & 0xffffu) : (uint32_t)(JK_TILE + (L6 ? 256 : 0));
        const int32_t row = g.row_base + (int32_t)(tile + item_row(h + k, wtid));
        if constexpr (L6) s.w[pos] = ((uint64_t)(uint32_t)key[h + k] << 32) | l6_row24((uint32_t)row);
        else s.w[pos] = tup_make<NARROW>((uint64_t)key[h + k], row);
        if constexpr (W10) s.idx[pos] = (int32_t)(uint32_t)((uint64_t)key[h + k] >> 32);
        else if (!NARROW) s.idx[pos] = row;
      }
    }
    {
      // the claim's answer is needed only now, for the flush behind the next barrier: the returning global atomic (~2 us, and
      // behind the previous tile's stores in the in-order vmcnt) had the scan and the regroup to come back.  Consumed right
      // after the scan, the waves that own the bins sat out that latency in front of their share of the regroup
      const uint32_t tid = opaque_tid();
      if (tid < ncoarse) {
        if (g.cap1) {
          const uint32_t cnt = s.hist[tid];
          const uint32_t region = (tid << g.xs) | (blockIdx.x & ((1u << g.xs) - 1u));
          if constexpr (L6) {
            if (s.odd[tid]) s.w[s.start[tid] + cnt - 1u] = ~0ULL;        // the padding of an odd run: a dead tuple (nobody else writes this slot)
          }
          const uint32_t room = g.rcap ? g.rcap[region] : g.cap1, first = g.rstart ? g.rstart[region] : region * g.cap1;
          if (claimed + cnt > room) { atomicExch(g.spec_flag, 1u); s.gbase[tid] = g.dump - s.start[tid]; }
          else s.gbase[tid] = first + claimed - s.start[tid];
        } else {
          s.gbase[tid] = s.cursor[tid] - s.start[tid];
          s.cursor[tid] += s.hist[tid];
        }
      }
      if (tid < 256) s.hist[tid] = 0;      // nobody reads hist again before the next tile's ranking
      // speculative layout: once ANY workgroup has seen a partition outgrow its room the host will repeat the side with
      // the exact layout -- the rest of this pass is wasted work (and, with skewed keys, slow work: every overflowing run
      // of every workgroup lands on the same dump lines).  Thread 0 looks at the flag, everybody acts on it after the barrier.
      if (g.cap1 && tid == 0) s.total_abort = __hip_atomic_load(g.spec_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool more = tile + JK_TILE < end;
    __builtin_amdgcn_sched_barrier(0);         // keep the prefetch BELOW the regroup: hoisted, its 32 registers spill
    LAB_PHASE(2)       // regroup (LDS writes issued) + claim answer
    if (FAST && more) prefetch(tile + JK_TILE);
    block_sync();
    LAB_PHASE(3)       // prefetch issue + barrier (the regroup's LDS writes land here)
    if (g.cap1 && s.total_abort) return;       // workgroup-uniform (read after the barrier)
    // flush: JK_SC_ITEMS unconditional stores per thread (dead slots -> this thread's dump slot), four at a time to
    // keep the register count under the 128 a 1024-thread workgroup gets (the prefetched keys stay in registers)
    const uint32_t total = s.total;
    const uint32_t ftid = opaque_tid();
    constexpr int GROUP = W10 ? 3 : 4;
    if constexpr (L6) {
      // pairs: lane i of group h takes LDS positions 2 p, 2 p + 1 (p = ftid + (h * GROUP + k) * THREADS; one 16-byte LDS read) and
      // stores them as ONE aligned 12-byte word triple at tuple position gbase[bin] + 2 p -- both tuples sit in the same even-length
      // run.  Unconditional, like the 8-byte flush: pairs behind `total` go to this thread's dump slots
#pragma unroll
      for (int h = 0; h < ITEMS / 2 / GROUP; ++h) {
        ulonglong2 ww[GROUP];
        uint2 hh[W10 ? GROUP : 1];
        uint32_t gb[GROUP];
#pragma unroll
        for (int k = 0; k < GROUP; ++k) {
          ww[k] = *reinterpret_cast<const ulonglong2 *>(&s.w[2u * (ftid + (h * GROUP + k) * THREADS)]);
          if constexpr (W10) hh[k] = *reinterpret_cast<const uint2 *>(&s.idx[2u * (ftid + (h * GROUP + k) * THREADS)]);
        }
#pragma unroll
        for (int k = 0; k < GROUP; ++k) gb[k] = s.gbase[(uint32_t)(ww[k].x >> (64 - g.b1)) & 255u];
#pragma unroll
        for (int k = 0; k < GROUP; ++k) {
          const uint32_t j = 2u * (ftid + (h * GROUP + k) * THREADS);
          const uint32_t dst = j < total ? gb[k] + j : g.dump + 2u * ftid;
          const uint32_t rmask = (1u << (32 - g.b1)) - 1u;
          const L6Pair d = l6_pack((uint32_t)(ww[k].x >> 32) & rmask, (uint32_t)ww[k].x & L6_ROW_MASK, (uint32_t)(ww[k].y >> 32) & rmask,
                                   (uint32_t)ww[k].y & L6_ROW_MASK);
          *reinterpret_cast<L6Pair *>(reinterpret_cast<unsigned char *>(out.w) + (size_t)dst * 6u) = d;
          if constexpr (W10) *reinterpret_cast<uint2 *>(out.idx + dst) = hh[k];      // (dst is even: one aligned 8-byte store)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the padding can make a tile up to 256 tuples longer than its 16384 rows: the first two waves take those 128 pairs
      if (ftid < 128u) {
        const uint32_t j = (uint32_t)JK_TILE + 2u * ftid;
        const ulonglong2 wx = *reinterpret_cast<const ulonglong2 *>(&s.w[j]);
        const uint32_t gbx = s.gbase[(uint32_t)(wx.x >> (64 - g.b1)) & 255u];
        const uint32_t dst = j < total ? gbx + j : g.dump + 2u * ftid;
        const uint32_t rmask = (1u << (32 - g.b1)) - 1u;
        const L6Pair d = l6_pack((uint32_t)(wx.x >> 32) & rmask, (uint32_t)wx.x & L6_ROW_MASK, (uint32_t)(wx.y >> 32) & rmask, (uint32_t)wx.y & L6_ROW_MASK);
        *reinterpret_cast<L6Pair *>(reinterpret_cast<unsigned char *>(out.w) + (size_t)dst * 6u) = d;
        if constexpr (W10) *reinterpret_cast<uint2 *>(out.idx + dst) = *reinterpret_cast<const uint2 *>(&s.idx[j]);
      }
    } else {
#pragma unroll
    for (int h = 0; h < ITEMS / GROUP; ++h) {
      uint64_t ww[GROUP];
      int32_t ii[GROUP];
      uint32_t gb[GROUP];
#pragma unroll
      for (int k = 0; k < GROUP; ++k) {
        const uint32_t j = ftid + (h * GROUP + k) * THREADS;
        ww[k] = s.w[j];
        ii[k] = NARROW ? 0 : s.idx[j];
      }
#pragma unroll
      for (int k = 0; k < GROUP; ++k)
        gb[k] = s.gbase[(fine_of(tup_key<NARROW>(ww[k]) + g.kbias, g.fb) >> g.b2) & 255u];
#pragma unroll
      for (int k = 0; k < GROUP; ++k) {
        const uint32_t j = ftid + (h * GROUP + k) * THREADS;
        uint32_t dst = j < total ? gb[k] + j : g.dump + ftid;
        if (LAB_BITS(g.dbg) & 4) dst &= 0xffffu;           // experiment: all stores land in a 512 KiB window
        if (LAB_BITS(g.dbg) & 1) dst = g.dump + ftid;      // experiment: no useful stores
        if (LAB_BITS(g.dbg) & 8) __builtin_nontemporal_store(ww[k], out.w + dst);      // experiment: streaming stores (level 1)
        else out.w[dst] = ww[k];
        if (!NARROW) out.idx[dst] = ii[k];
      }
      __builtin_amdgcn_sched_barrier(0);       // one group's LDS reads at a time: hoisted together they spill
    }
    }
    // the stores above stay in flight: this waits for the LOADS only.  Unconditional: keeping the old keys alive for the
    // `no more tiles` case would cost 16 registers across the flush
    LAB_PHASE(4)       // flush: LDS reads + store issue
    if (FAST) consume(tile + JK_TILE);
    __builtin_amdgcn_sched_barrier(0);         // narrow the keys HERE: sunk into the next ranking, the 64-bit words stay live
    LAB_PHASE(5)       // wait for the next tile's keys
  }
#ifdef GDF_AMD_LAB
  if (g.lab_clock && (threadIdx.x == 0 || threadIdx.x == THREADS - 64))
    for (int i = 0; i < 6; ++i) g.lab_clock[((size_t)blockIdx.x * 2 + (threadIdx.x ? 1 : 0)) * 8 + i] = lab_ph[i];
#endif
}

// 2b. level-1 scatter of a probe side that CARRIES A PAYLOAD (Tuples::pay, PayCarry): one FAST key column without a mask,
// NARROW tuples, and per row one 64-bit payload word read from the relation's non-key column(s) in the same pass --
//   PMODE 1: one 8-byte column; 2: one 4-byte column (zero-extended); 3: two 4-byte columns (column 0 in the low half);
//   4 (round 6): TWO 8-byte columns -- a 16-byte payload element per tuple, 24 bytes per tuple in LDS, six tuples
//   per thread (6144-tuple tiles).  A second 8-byte column gathered behind the join costs one 64-byte sector per value (24 - 27 ms
//   per 1e9 pairs); carried, 8 more bytes through both regroup levels and the probe are 48 B of streaming traffic per row.
// The same phases as jk_scatter1 (rank | scan + claim | regroup in LDS | flush, next tile's words prefetched before the
// flush) on 8192-tuple tiles: the tile holds 16 bytes per tuple, and key + payload + their prefetch registers of 8 items
// fit the 128 VGPRs of a 1024-thread workgroup.  Tuples and payload words leave as two parallel streams of 256-byte runs.
struct PaySrc { const void *col[2]; };
constexpr int JK_PAY_ITEMS = 8;
constexpr int JK_PAY_THREADS = 1024;
__host__ __device__ constexpr int pay_words(int pmode) { return pmode == 4 ? 2 : (pmode ? 1 : 0); }
__host__ __device__ constexpr int pay_items1(int pmode) { return pmode == 4 ? 6 : JK_PAY_ITEMS; }      // level-1 tuples per thread
template <int FAST, int PMODE>
__global__ __launch_bounds__(JK_PAY_THREADS) void jk_scatter1_pay(KeyTable t, KeyPlan plan, PartGeom g, const uint32_t *__restrict__ H1off,
                                                                  PaySrc ps, Tuples out) {
  constexpr int THREADS = JK_PAY_THREADS, ITEMS = pay_items1(PMODE), PW = pay_words(PMODE);
  using Tile = TileLds<true, THREADS, PW, ITEMS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  Tile &s = *reinterpret_cast<Tile *>(tile_raw);
  constexpr int TILE = THREADS * ITEMS;
  const int chunk = blockIdx.x;
  const uint32_t ncoarse = 1u << g.b1;
  if (!g.cap1 && threadIdx.x < ncoarse) s.cursor[threadIdx.x] = H1off[(size_t)threadIdx.x * g.nchunks + chunk];
  const uint32_t begin = (uint32_t)((int64_t)chunk * g.chunk);
  const uint32_t end = (int64_t)begin + g.chunk < t.nrows ? (uint32_t)(begin + g.chunk) : (uint32_t)t.nrows;
  auto item_row = [](int k, uint32_t tid) -> uint32_t { return 2u * ((uint32_t)(k >> 1) * THREADS + tid) + (k & 1); };
  uint64_t nxt[ITEMS];                                   // raw key words of the next tile
  uint64_t nxp[(PMODE == 1 || PMODE == 4) ? ITEMS : 1];  // raw payload words, 8-byte column
  uint64_t nxq[PMODE == 4 ? ITEMS : 1];                  // ... and the second 8-byte column
  uint32_t nxa[PMODE != 1 ? ITEMS : 1];                  // 4-byte column 0
  uint32_t nxb[PMODE == 3 ? ITEMS : 1];                  // 4-byte column 1
  const void *col = t.col[0].data;
  auto prefetch = [&](uint32_t tile) {
    const uint32_t tid = opaque_tid();
#pragma unroll
    for (int k = 0; k < ITEMS; k += 2) {
      const uint32_t i = tile + item_row(k, tid);
      const int64_t at = (int64_t)(i + 2 <= end ? i : end - 2);
      fast_pair<FAST>(col, at, nxt[k], nxt[k + 1]);
      if (PMODE == 1 || PMODE == 4) {
        fast_pair<8>(ps.col[0], at, nxp[k], nxp[k + 1]);
        if (PMODE == 4) fast_pair<8>(ps.col[1], at, nxq[k], nxq[k + 1]);
      } else {
        uint64_t a0, a1;
        fast_pair<4>(ps.col[0], at, a0, a1);
        nxa[k] = (uint32_t)a0; nxa[k + 1] = (uint32_t)a1;
        if (PMODE == 3) {
          uint64_t b0, b1;
          fast_pair<4>(ps.col[1], at, b0, b1);
          nxb[k] = (uint32_t)b0; nxb[k + 1] = (uint32_t)b1;
        }
      }
    }
  };
  prefetch(begin);
  if (threadIdx.x < 256) s.hist[threadIdx.x] = 0;
  block_sync();
  uint32_t key[ITEMS];
  uint64_t pay[ITEMS], pay2[PMODE == 4 ? ITEMS : 1];
  uint32_t okmask = 0;
  auto consume = [&](uint32_t tile) {
    const uint32_t tid = opaque_tid();
    okmask = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const bool second = (k & 1) == 0 && tile + item_row(k, tid) + 1 == end;       // the pair was read one row early (prefetch)
      uint64_t raw = second ? nxt[k + 1] : nxt[k];
      const bool joinable = plan.mode != KM_RAW_FLOAT || fast_float_word<FAST>(raw);
      const uint64_t k64 = raw - plan.kmin;
      key[k] = (uint32_t)k64;
      okmask |= (uint32_t)(joinable && (tile + item_row(k, tid) < end) && k64 <= plan.klimit) << k;
      if constexpr (PMODE == 1 || PMODE == 4) {
        pay[k] = second ? nxp[k + 1] : nxp[k];
        if constexpr (PMODE == 4) pay2[k] = second ? nxq[k + 1] : nxq[k];
      } else if constexpr (PMODE == 2) pay[k] = second ? nxa[k + 1] : nxa[k];
      else pay[k] = (uint64_t)(second ? nxa[k + 1] : nxa[k]) | ((uint64_t)(second ? nxb[k + 1] : nxb[k]) << 32);
    }
  };
  consume(begin);
  for (uint32_t tile = begin; tile < end; tile += TILE) {
    uint32_t binrank[ITEMS];                    // bin << 16 | rank within (tile, bin); bin 256 = does not travel
    constexpr int HG = ITEMS % 4 == 0 ? 4 : 3;
#pragma unroll
    for (int h = 0; h < ITEMS; h += HG) {
#pragma unroll
      for (int k = h; k < h + HG; ++k) {
        const uint32_t b = fine_of((uint64_t)key[k] + g.kbias, g.fb) >> g.b2;
        binrank[k] = (okmask >> k) & 1u ? b : 256u + (threadIdx.x & 63u);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) binrank[k] = (binrank[k] << 16) | atomicAdd(&s.hist[binrank[k]], 1u);
    block_sync();
    uint32_t claimed = 0;
    {
      const uint32_t tid = opaque_tid();
      if (g.cap1 && tid < ncoarse) {
        const uint32_t cnt = s.hist[tid];
        if (cnt) claimed = atomicAdd(&g.spec_cursor1[(tid << g.xs) | (blockIdx.x & ((1u << g.xs) - 1u))], cnt);
      }
      tile_scan_bins(s, ncoarse, tid);
      block_sync();
    }
    const uint32_t wtid = opaque_tid();
    {
      uint32_t st[ITEMS];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) st[k] = s.start[(binrank[k] >> 16) & 255u];
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        const uint32_t pos = (okmask >> k) & 1u ? st[k] + (binrank[k] & 0xffffu) : (uint32_t)TILE;
        const int32_t row = g.row_base + (int32_t)(tile + item_row(k, wtid));
        s.w[pos] = ((uint64_t)key[k] << 32) | (uint32_t)row;
        if constexpr (PMODE == 4) *reinterpret_cast<ulonglong2 *>(&s.pay[2 * (size_t)pos]) = ulonglong2{pay[k], pay2[k]};
        else s.pay[pos] = pay[k];
      }
    }
    {   // the claim is looked at after the regroup (see jk_scatter1)
      const uint32_t tid = opaque_tid();
      if (tid < ncoarse) {
        if (g.cap1) {
          const uint32_t cnt = s.hist[tid];
          const uint32_t region = (tid << g.xs) | (blockIdx.x & ((1u << g.xs) - 1u));
          if (claimed + cnt > g.cap1) { atomicExch(g.spec_flag, 1u); s.gbase[tid] = g.dump - s.start[tid]; }
          else s.gbase[tid] = region * g.cap1 + claimed - s.start[tid];
        } else {
          s.gbase[tid] = s.cursor[tid] - s.start[tid];
          s.cursor[tid] += s.hist[tid];
        }
      }
      if (tid < 256) s.hist[tid] = 0;
      if (g.cap1 && tid == 0) s.total_abort = __hip_atomic_load(g.spec_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool more = tile + TILE < end;
    __builtin_amdgcn_sched_barrier(0);
    if (more) prefetch(tile + TILE);
    block_sync();
    if (g.cap1 && s.total_abort) return;
    const uint32_t total = s.total;
    const uint32_t ftid = opaque_tid();
    constexpr int GROUP = ITEMS % 4 == 0 ? 4 : 3;
#pragma unroll
    for (int h = 0; h < ITEMS / GROUP; ++h) {
      uint64_t ww[GROUP], pp[GROUP], pq[PMODE == 4 ? GROUP : 1];
      uint32_t gb[GROUP];
#pragma unroll
      for (int k = 0; k < GROUP; ++k) {
        const uint32_t j = ftid + (h * GROUP + k) * THREADS;
        ww[k] = s.w[j];
        if constexpr (PMODE == 4) { const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(&s.pay[2 * (size_t)j]); pp[k] = e.x; pq[k] = e.y; }
        else pp[k] = s.pay[j];
      }
#pragma unroll
      for (int k = 0; k < GROUP; ++k) gb[k] = s.gbase[(fine_of((ww[k] >> 32) + g.kbias, g.fb) >> g.b2) & 255u];
#pragma unroll
      for (int k = 0; k < GROUP; ++k) {
        const uint32_t j = ftid + (h * GROUP + k) * THREADS;
        const uint32_t dst = j < total ? gb[k] + j : g.dump + ftid;
        out.w[dst] = ww[k];
        if constexpr (PMODE == 4) *reinterpret_cast<ulonglong2 *>(out.pay + 2 * (size_t)dst) = ulonglong2{pp[k], pq[k]};
        else out.pay[dst] = pp[k];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    consume(tile + TILE);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// 3. level-2 scatter: tuples of one coarse partition -> fine partitions.  A tile
// never crosses a coarse boundary; bins claim their global range with one
// atomicAdd per (tile, non-empty bin) on the fine cursors.
struct Level2Map {                     // small host-built tables, device resident
  const uint32_t *coarse_off;          // [ncoarse] first tuple of each coarse partition
  const uint32_t *coarse_end;          // [ncoarse] one past its last tuple (== the next partition's first in the exact layout)
  const uint32_t *tile_prefix;         // [ncoarse+1] tiles before each coarse partition
  int xs;                              // the arrays describe 2^(b1+xs) SEGMENTS, segment >> xs = coarse partition (PartGeom::xs)
  uint32_t ntiles;                     // set by the launcher
  int xcd_order;                       // set by the launcher: the grid is 8 * ceil(ntiles / 8) blocks, see jk_scatter2
  const uint32_t *ntiles_dev;          // non-null: the map was built on the device (jk_make_l2map); ntiles is then an upper bound
                                       // for the grid and the real tile count is read from here
  uint32_t nseg;                       // 0: ncoarse << xs segments, segment >> xs = coarse partition.  Otherwise (fused multi-GPU
                                       // receive buffer): this many segments in the order (coarse partition, sender, XCD region) --
                                       // jk_make_l2map's permutation of the buffer's sender-major regions -- the coarse partition of
                                       // segment i is (i >> 3) / world
  const uint32_t *keys32;              // non-null: the input is this array of 4-byte keys (a receive buffer), tuple = key << 32 | position
  uint32_t calib_step;                 // > 1: a CALIBRATION run of the level-2 buffer's placement tournament -- only every calib_step-th tile
                                       // (of every XCD's eighth) is regrouped, the grid is that much smaller (sc2_grid)
};
// grid of a level-2 launch: one workgroup per tile, rounded up to whole rounds over the 8 XCDs; a calibration run takes every
// calib_step-th tile
static inline uint32_t sc2_grid(const Level2Map &m, uint32_t ntiles) {
  const uint32_t tiles = m.calib_step > 1 ? (ntiles + m.calib_step - 1) / m.calib_step : ntiles;
  return m.xcd_order ? ((tiles + 7) / 8) * 8 : tiles;
}

// K32: the input is a receive buffer of 4-byte keys (Level2Map::keys32, fused multi-GPU join) -- its own instantiations, so that the
// single-GPU kernels carry no trace of it
// IN6: the input is a stream of six-byte level-1 tuples (L6 above; P6 output only): pairs are read with one 12-byte load, the
// hash comes out of the tuple (coarse partition = the segment's, remainder = the tuple's) and the row number gets its region bits
// back from the segment -- this kernel then does not hash at all
template <bool NARROW, int THREADS, int PAY = 0, bool P6 = false, bool K32 = false, bool IN6 = false, int NITEMS = 0>
__global__ __launch_bounds__(THREADS) void jk_scatter2(PartGeom g, Level2Map m, Tuples in,
                                                             uint32_t *__restrict__ fine_cursor, Tuples out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  // a payload-carrying tile holds 16 bytes per tuple: half the tuples per thread keep it at four workgroups per CU
  constexpr int ITEMS = NITEMS ? NITEMS : (PAY ? JK_PAY_ITEMS : JK_SC_ITEMS);
  using Tile = TileLds<NARROW, THREADS, PAY, ITEMS>;
  Tile &s = *reinterpret_cast<Tile *>(tile_raw);
  constexpr int JK_TILE = THREADS * ITEMS;
  const uint32_t ncoarse = 1u << g.b1, nsub = 1u << g.b2;
  // locate the coarse partition that owns this tile (binary search over <= 257 entries)
  // XCD x (= blockIdx.x % 8) takes the x-th EIGHTH of the tiles, in order: every writer of a fine partition -- the tiles
  // of its coarse partition -- then shares one L2, which merges the partial first / last lines of neighbouring
  // (tile, bin) runs before they reach HBM (PartGeom::xs has the level-1 half of this)
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t tile_id = (m.xcd_order ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x) * (m.calib_step > 1 ? m.calib_step : 1u);
  if (tile_id >= (m.ntiles_dev ? *m.ntiles_dev : m.ntiles)) return;
  uint32_t lo = 0, hi = m.nseg ? m.nseg : (ncoarse << m.xs);
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (m.tile_prefix[mid] <= tile_id) lo = mid; else hi = mid;
  }
  const uint32_t p = m.nseg ? (lo >> 3) / g.world : (lo >> m.xs);          // fused receive buffer: segments are (coarse, sender, XCD)
  const uint32_t begin = m.coarse_off[lo] + (tile_id - m.tile_prefix[lo]) * JK_TILE;
  const uint32_t pend = m.coarse_end[lo];
  const uint32_t end = begin + JK_TILE < pend ? begin + JK_TILE : pend;
  const uint32_t submask = nsub - 1;

  if (threadIdx.x < 256) s.hist[threadIdx.x] = 0;
  block_sync();
  // straight-line phases as in jk_scatter1: tuples beyond the tile's end are ranked on a trash counter (bin 256) and
  // written to the trash slot of the LDS tile instead of being skipped by a branch per item
  uint64_t w[ITEMS], pay[PAY ? ITEMS : 1], pay2[PAY == 2 ? ITEMS : 1];
  // a receive buffer of 4-byte keys (fused multi-GPU join), full tile: four consecutive keys per 16-byte load -- a quarter of the load
  // instructions (which tuple of the tile a thread holds does not matter)
  const bool quads = K32 && end - begin == (uint32_t)JK_TILE;
  int32_t idx[ITEMS];
  uint32_t live6 = 0;                                 // IN6: bit k = tuple k of this thread is a live one
  if constexpr (IN6) {
    static_assert(!IN6 || (P6 && !PAY && !K32), "six-byte input: the main path only");
    const uint32_t region = lo & ((1u << m.xs) - 1u);
    L6Pair d[ITEMS / 2];
    uint2 hw[NARROW ? 1 : ITEMS / 2];                 // WIDE (ten-byte tuples, p10_key): the pair's high words from the parallel array
#pragma unroll
    for (int k = 0; k < ITEMS / 2; ++k) {             // all loads first: pair k of this thread = tuples begin + 2 (k THREADS + tid), + 1
      const uint32_t i = begin + 2u * (k * THREADS + threadIdx.x);
      const uint32_t ic = i < end ? i : end - 2u;     // (segments and tiles start and end at even positions)
      d[k] = *reinterpret_cast<const L6Pair *>(reinterpret_cast<const unsigned char *>(in.w) + (size_t)ic * 6u);
      if constexpr (!NARROW) hw[k] = *reinterpret_cast<const uint2 *>(in.idx + ic);
    }
#pragma unroll
    for (int k = 0; k < ITEMS / 2; ++k) {
      uint32_t r0, w0, r1, w1;
      l6_unpack(d[k], r0, w0, r1, w1);
      const bool in_tile = begin + 2u * (k * THREADS + threadIdx.x) < end;
      const uint32_t all = (1u << (32 - g.b1)) - 1u;
      live6 |= (uint32_t)(in_tile && !(r0 == all && w0 == L6_ROW_MASK)) << (2 * k);
      live6 |= (uint32_t)(in_tile && !(r1 == all && w1 == L6_ROW_MASK)) << (2 * k + 1);
      // the tuple as the rest of the kernel wants it: hash word | row
      w[2 * k] = ((uint64_t)((p << (32 - g.b1)) | r0) << 32) | (uint32_t)(g.row_base + (int32_t)l6_row(w0, region));
      w[2 * k + 1] = ((uint64_t)((p << (32 - g.b1)) | r1) << 32) | (uint32_t)(g.row_base + (int32_t)l6_row(w1, region));
      if constexpr (NARROW) idx[2 * k] = idx[2 * k + 1] = 0;
      else { idx[2 * k] = (int32_t)hw[k].x; idx[2 * k + 1] = (int32_t)hw[k].y; }
    }
  } else if (K32 && quads) {                          // workgroup-uniform
    if constexpr (K32) {
#pragma unroll
      for (int q = 0; q < ITEMS / 4; ++q) {
        const uint32_t i0 = begin + 4u * (q * THREADS + threadIdx.x);
        const uint4 kk = *reinterpret_cast<const uint4 *>(m.keys32 + i0);     // regions start at multiples of 64 keys, tiles at multiples of 4096
        w[4 * q + 0] = ((uint64_t)kk.x << 32) | (uint32_t)(g.row_base + (int32_t)i0);
        w[4 * q + 1] = ((uint64_t)kk.y << 32) | (uint32_t)(g.row_base + (int32_t)(i0 + 1));
        w[4 * q + 2] = ((uint64_t)kk.z << 32) | (uint32_t)(g.row_base + (int32_t)(i0 + 2));
        w[4 * q + 3] = ((uint64_t)kk.w << 32) | (uint32_t)(g.row_base + (int32_t)(i0 + 3));
        idx[4 * q] = idx[4 * q + 1] = idx[4 * q + 2] = idx[4 * q + 3] = 0;
      }
    }
  } else {
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {         // all loads first
    const uint32_t i = begin + k * THREADS + threadIdx.x;
    const uint32_t ic = i < end ? i : end - 1;        // clamped, unconditional: see fetch_keys
    if constexpr (K32) w[k] = ((uint64_t)m.keys32[ic] << 32) | (uint32_t)(g.row_base + (int32_t)ic);
    else w[k] = in.w[ic];                             // (non-temporal loads, which help jk_scatter1, cost 2 % here)
    idx[k] = NARROW ? 0 : in.idx[ic];
    if constexpr (PAY == 2) { const ulonglong2 e = *reinterpret_cast<const ulonglong2 *>(in.pay + 2 * (size_t)ic); pay[k] = e.x; pay2[k] = e.y; }
    else if (PAY) pay[k] = in.pay[ic];
  }
  }
  uint32_t binrank[ITEMS];
  // K32 (a fused join's receive buffer, world >= 1 ranks): hash_a(key32 + kbias) without 64-bit arithmetic -- key_fold's high word is
  // kbias's own or one more (the carry), two constants instead of an add-with-carry and a quarter-rate multiply -- and, for a
  // power-of-two world (decided once per kernel), the rank remap h * world as a shift: two of the four quarter-rate multiplies
  // per tuple are gone (half of this kernel's time is VALU issue, tools/kernel_blocks.py)
  const uint32_t kb_low = (uint32_t)g.kbias, kb_fold0 = (uint32_t)(g.kbias >> 32) * 0x9e3779b1u, kb_fold1 = kb_fold0 + 0x9e3779b1u;
  auto rank_tuples = [&](auto pow2_world) {
    constexpr bool POW2W = decltype(pow2_world)::value;
    const uint32_t wshift = POW2W ? (uint32_t)(31 - __clz((int)(g.world ? g.world : 1u))) : 0u;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t i = begin + k * THREADS + threadIdx.x;
      if constexpr (IN6) {
        const uint32_t q = (uint32_t)(w[k] >> 32);               // the tuple brought its hash along
        binrank[k] = ((live6 >> k) & 1u) ? ((uint32_t)((uint64_t)q >> (32 - g.fb)) & submask) : 256u + (threadIdx.x & 63u);
        continue;                                                // (w[k] already is hash word | row, what the P6 flush wants)
      }
      uint32_t q, lh;
      if constexpr (K32) {
        const uint32_t key32 = (uint32_t)(w[k] >> 32), low = key32 + kb_low;
        q = lowbias32(low ^ (low < key32 ? kb_fold1 : kb_fold0));
        lh = POW2W ? q << wshift : (uint32_t)((uint64_t)q * g.world);
      } else {
        q = hash_a(tup_key<NARROW>(w[k]) + g.kbias);
        lh = local_hash(q, g.world);
      }
      const uint32_t bin = (uint32_t)((uint64_t)lh >> (32 - g.fb)) & submask;
      binrank[k] = ((K32 && quads) || i < end) ? bin : 256u + (threadIdx.x & 63u);     // (quads: a full tile, every tuple is live whatever order they were fetched in)
      // six-byte tuples leave as (hash remainder, row): the key is not needed again, the LDS tile holds the hash word and the flush
      // does not hash a second time (two quarter-rate multiplies per tuple: the kernel's ALU work is not hidden at 4 waves per SIMD)
      if constexpr (P6 && !NARROW) {
        // WIDE tuples (key64, row) become ten-byte ones here: hash word | row in w, the raw key's high word in idx (p10_key)
        const uint32_t row = (uint32_t)idx[k];
        idx[k] = (int32_t)(uint32_t)((w[k] + g.kbias) >> 32);
        w[k] = ((uint64_t)p6_low(q, lh, g.world) << 32) | row;
      } else if constexpr (P6) {
        if constexpr (K32 && POW2W) w[k] = ((uint64_t)(g.world > 1 ? __builtin_rotateleft32(q, wshift) : lh) << 32) | (uint32_t)w[k];
        else w[k] = ((uint64_t)p6_low(q, lh, g.world) << 32) | (uint32_t)w[k];
      }
    }
  };
  if (K32 && (g.world & (g.world - 1u)) == 0) rank_tuples(std::true_type{});      // (workgroup-uniform)
  else rank_tuples(std::false_type{});
#pragma unroll
  for (int k = 0; k < ITEMS; ++k)            // sixteen atomics in flight, one wait
    binrank[k] = (binrank[k] << 16) | atomicAdd(&s.hist[binrank[k]], 1u);
  block_sync();
  // the claim -- a returning global atomic -- is issued as soon as the counts are final and looked at after the LDS regroup:
  // its answer is only needed by the flush (see jk_scatter1)
  uint32_t claimed = 0, mine = 0;
  if (threadIdx.x < nsub) {
    mine = s.hist[threadIdx.x];
    if (mine) claimed = atomicAdd(&fine_cursor[(p << g.b2) | threadIdx.x], mine);
  }
  tile_scan_bins(s, nsub);
  block_sync();
#pragma unroll
  for (int h = 0; h < ITEMS; h += 8) {
    uint32_t st[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) st[k] = s.start[(binrank[h + k] >> 16) & 255u];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t pos = (binrank[h + k] >> 24) ? (uint32_t)JK_TILE : st[k] + (binrank[h + k] & 0xffffu);
      s.w[pos] = w[h + k];
      if (!NARROW) s.idx[pos] = idx[h + k];
      if constexpr (PAY == 2) *reinterpret_cast<ulonglong2 *>(&s.pay[2 * (size_t)pos]) = ulonglong2{pay[h + k], pay2[h + k]};
      else if (PAY) s.pay[pos] = pay[h + k];
    }
  }
  if (threadIdx.x < nsub && mine) {
    const uint32_t f = (p << g.b2) | threadIdx.x;
    const uint32_t limit = g.fstart ? g.fstart[f] + g.fcap[f] : (f + 1) * g.cap2;
    if (g.cap2 && claimed + mine > limit) { atomicExch(g.spec_flag, 1u); s.gbase[threadIdx.x] = g.dump - s.start[threadIdx.x]; }
    else s.gbase[threadIdx.x] = claimed - s.start[threadIdx.x];
  }
  block_sync();
  tile_flush<false, NARROW, THREADS, PAY, ITEMS, P6>(s, g, out);
}

// ---------------------------------------------------------------------------
// Host-free bookkeeping of the histogram-free (speculative) layout.  Round 1 read the fill counters back after every
// scatter level, built the level-2 segment map / the work units / the output offsets in host loops and uploaded them:
// ~0.3 ms of idle GPU per C3 join in six gaps (tools/gpu/gaps.sh).  These single-workgroup kernels do the same on the device,
// so that the probe side runs scatter1 -> map -> scatter2 -> units -> sample without a host round trip; the host reads ONE
// small state block before the write pass.
// ---------------------------------------------------------------------------
constexpr int JK_BK_THREADS = 1024;
// exclusive block scan of one value per thread (JK_BK_THREADS threads); returns the exclusive prefix, *total = the sum
template <class T>
__device__ __forceinline__ T bk_block_scan(T v, T *lds_wave, T *total) {
  const T incl = wave_scan_incl(v);
  block_sync();
  if (lane_id() == WAVE - 1) lds_wave[threadIdx.x / WAVE] = incl;
  block_sync();
  T woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < JK_BK_THREADS / WAVE; ++w) {
    if (w < (int)(threadIdx.x / WAVE)) woff += lds_wave[w];
    tot += lds_wave[w];
  }
  *total = tot;
  return woff + incl - v;
}
// level-2 segment map from the level-1 fill counters: segment c = [c * cap1, c * cap1 + fill[c]), tile_prefix = tiles before it
// fj_world != 0 (receive buffer of a fused multi-GPU join): the buffer holds its regions sender-major -- region
// ((s * ncoarse + c) << 3) | x -- but they are PROCESSED coarse-partition-major, segment ((c * world + s) << 3) | x: all tiles
// of a coarse partition are then neighbours in the tile order, i.e. run on one XCD, whose L2 merges the short runs they
// write into the same fine partitions (the reason for jk_scatter2's XCD-ordered tiles)
constexpr uint32_t JK_L2MAP_LDS_SEGS = 16384;      // segments the staged version of jk_make_l2map holds (64 KB of LDS): C3's 2^(8 + 6)
__global__ __launch_bounds__(JK_BK_THREADS) void jk_make_l2map(const uint32_t *__restrict__ fill, uint32_t nseg, uint32_t cap1, uint32_t tile,
                                                               uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end,
                                                               uint32_t *__restrict__ tile_prefix, uint32_t *__restrict__ ntiles,
                                                               uint32_t fj_world, uint32_t ncoarse, const uint32_t *__restrict__ rstart = nullptr,
                                                               const uint32_t *__restrict__ rcap = nullptr) {
  __shared__ uint32_t lds_wave[JK_BK_THREADS / WAVE];
  extern __shared__ __attribute__((aligned(16))) uint32_t l2map_lds[];      // [nseg] (staged version; the launcher sizes it, 0 words otherwise)
  const uint32_t per = (nseg + JK_BK_THREADS - 1) / JK_BK_THREADS;
  const uint32_t c0 = threadIdx.x * per;
  auto region_of = [&](uint32_t seg) -> uint32_t {
    if (!fj_world) return seg;
    const uint32_t cs = seg >> 3, c = cs / fj_world, sender = cs - c * fj_world;
    return ((sender * ncoarse + c) << 3) | (seg & 7u);
  };
  // A region that overflowed (skewed keys: its runs went to the dump slot and the overflow flag is up, the host will repeat
  // the side with the exact layout) has a fill counter ABOVE its capacity: clamped, or the segments -- and level 2's reads --
  // run past the region and, for the last ones, past the buffer.  (Found by tools/stress_join.py: an illegal address on a
  // probe side with a tenth of its rows on one key.)
  auto filled = [&](uint32_t region) -> uint32_t { const uint32_t n = fill[region], room = rcap ? rcap[region] : cap1; return n < room ? n : room; };
  const bool pow2 = (tile & (tile - 1u)) == 0;
  const int shift = __ffs((int)tile) - 1;
  auto tiles_of = [&](uint32_t n) -> uint32_t { return pow2 ? (n + tile - 1u) >> shift : (n + tile - 1u) / tile; };
  if (nseg <= JK_L2MAP_LDS_SEGS) {
    // STAGED (round 5): this one workgroup stands between the two regroup kernels of a probe side -- 41 us when every thread walked its 16
    // segments twice through strided global reads.  Every global access is coalesced now (segment k * 1024 + thread), the counters are
    // read once, and the per-thread runs of the scan go through LDS.
    constexpr int MAXK = (int)(JK_L2MAP_LDS_SEGS / JK_BK_THREADS);
    uint32_t n_[MAXK], first_[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const uint32_t c = (uint32_t)k * JK_BK_THREADS + threadIdx.x;
      n_[k] = 0; first_[k] = 0;
      if (c < nseg) {
        const uint32_t r = region_of(c);
        n_[k] = filled(r);
        first_[k] = rstart ? rstart[r] : r * cap1;
        l2map_lds[c] = tiles_of(n_[k]);
      }
    }
    block_sync();
    uint32_t mine = 0;
    for (uint32_t c = c0; c < c0 + per && c < nseg; ++c) mine += l2map_lds[c];
    uint32_t total;
    uint32_t run = bk_block_scan<uint32_t>(mine, lds_wave, &total);
    for (uint32_t c = c0; c < c0 + per && c < nseg; ++c) { const uint32_t t = l2map_lds[c]; l2map_lds[c] = run; run += t; }
    block_sync();
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
      const uint32_t c = (uint32_t)k * JK_BK_THREADS + threadIdx.x;
      if (c < nseg) { seg_begin[c] = first_[k]; seg_end[c] = first_[k] + n_[k]; tile_prefix[c] = l2map_lds[c]; }
    }
    if (threadIdx.x == 0) { tile_prefix[nseg] = total; *ntiles = total; }
    return;
  }
  uint32_t mine = 0;
  for (uint32_t c = c0; c < c0 + per && c < nseg; ++c) mine += tiles_of(filled(region_of(c)));
  uint32_t total;
  uint32_t run = bk_block_scan<uint32_t>(mine, lds_wave, &total);
  for (uint32_t c = c0; c < c0 + per && c < nseg; ++c) {
    const uint32_t r = region_of(c), n = filled(r);
    const uint32_t first = rstart ? rstart[r] : r * cap1;
    seg_begin[c] = first;
    seg_end[c] = first + n;
    tile_prefix[c] = run;
    run += tiles_of(n);
  }
  if (threadIdx.x == 0) { tile_prefix[nseg] = total; *ntiles = total; }
}
static gdf_error l2map_prepare(uint32_t nseg, size_t *lds) {
  *lds = nseg <= JK_L2MAP_LDS_SEGS ? sizeof(uint32_t) * (size_t)nseg : 0;
  HIP_TRY(hipFuncSetAttribute((const void *)jk_make_l2map, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(uint32_t) * JK_L2MAP_LDS_SEGS)));
  return GDF_SUCCESS;
}
__global__ __launch_bounds__(256) void jk_init_cursor(uint32_t *cur, uint32_t nfine, uint32_t cap2, const uint32_t *__restrict__ fstart = nullptr) {
  for (uint32_t f = blockIdx.x * 256 + threadIdx.x; f <= nfine; f += gridDim.x * 256) cur[f] = f < nfine ? (fstart ? fstart[f] : f * cap2) : 0u;   // [nfine]: overflow flag
}

// level 2's map of an EXACT-layout side from the device copy of its partition index (begin[] = exclusive scan of the fine histogram, cnt[] the
// histogram): coarse_off[c] = first tuple of coarse partition c (ncoarse + 1 entries), tile_prefix = level-2 tiles before it, cursor[] = a
// copy of begin[].  One workgroup; what the host computed from the read-back histogram and sent in three copies (round 5: the copies
// stood between the histogram's read-back and level 1, and their host vectors forced a synchronisation behind level 2).
__global__ __launch_bounds__(JK_BK_THREADS) void jk_level2_index(const uint32_t *__restrict__ begin, const uint32_t *__restrict__ cnt, uint32_t nfine, int b2,
                                                                 uint32_t tile, uint32_t *__restrict__ coarse_off, uint32_t *__restrict__ tile_prefix,
                                                                 uint32_t *__restrict__ cursor) {
  __shared__ uint32_t lds_wave[JK_BK_THREADS / WAVE];
  const uint32_t ncoarse = nfine >> b2, total_rows = begin[nfine - 1] + cnt[nfine - 1];
  uint32_t run = 0;               // (ncoarse <= 2^8 in every geometry; the loop serves any)
  for (uint32_t base = 0; base < ncoarse; base += JK_BK_THREADS) {
    const uint32_t c = base + threadIdx.x;
    uint32_t first = 0, tiles = 0;
    if (c < ncoarse) {
      first = begin[(size_t)c << b2];
      const uint32_t next = c + 1 < ncoarse ? begin[(size_t)(c + 1) << b2] : total_rows;
      tiles = (next - first + tile - 1) / tile;
    }
    uint32_t total;
    const uint32_t before = bk_block_scan<uint32_t>(tiles, lds_wave, &total);
    if (c < ncoarse) { coarse_off[c] = first; tile_prefix[c] = run + before; }
    run += total;
    block_sync();
  }
  if (threadIdx.x == 0) { coarse_off[ncoarse] = total_rows; tile_prefix[ncoarse] = run; }
  for (uint32_t f = threadIdx.x; f < nfine; f += JK_BK_THREADS) cursor[f] = begin[f];
}

// ---------------------------------------------------------------------------
// 4. probe: one workgroup per work unit
// ---------------------------------------------------------------------------
struct Unit {
  uint32_t build_begin, build_count;   // tuple range of the fine partition on the build side
  uint32_t probe_begin, probe_count;   // this unit's slice of the partition on the probe side
};

// work units + output offsets of the optimistic pass from the fine fill counters of a speculative probe side: one thread
// per fine partition, JK_PROBE_CHUNK probe tuples per unit.  Unit numbers and output offsets are claimed per WAVE (a wave
// scan inside, one atomicAdd per wave on each of the two running totals), so the units come out in wave order rather than
// partition order -- nothing depends on it (the pair order of a join is unspecified).  A single workgroup walking the
// partitions in order took 135-200 us (dependent, uncoalesced reads); this takes a few microseconds.
// state[0] = units, [1] = sum of their probe counts, [2] = probe tuples in all fine partitions, [3] = an overflow flag was up
__global__ __launch_bounds__(256) void jk_make_units(uint32_t nfine, uint32_t cap2, const uint32_t *__restrict__ cursor,
                                                     const uint32_t *__restrict__ level1_flag,
                                                     const uint32_t *__restrict__ build_begin, const uint32_t *__restrict__ build_cnt,
                                                     int keep_probe, Unit *__restrict__ units, uint64_t *__restrict__ off,
                                                     unsigned long long *__restrict__ state, const uint32_t *__restrict__ fstart = nullptr,
                                                     const uint32_t *__restrict__ fcap = nullptr) {
  const uint32_t f = blockIdx.x * 256 + threadIdx.x;
  const uint32_t fc = f < nfine ? f : nfine - 1;
  const uint32_t cur = cursor[fc], bn = build_cnt[fc], bb = build_begin[fc];
  const uint32_t first = fstart ? fstart[fc] : fc * cap2;
  if (fcap) cap2 = fcap[fc];                       // (per-partition room: SkewCaps)
  // (a fine partition that outgrew its room -- overflow flag up, the host repeats with the exact layout -- is cut at its
  // capacity: its real count would make more units than the arrays hold)
  const uint32_t grown = f < nfine ? cur - first : 0u;
  const uint32_t all = grown < cap2 ? grown : cap2;
  const uint32_t pn = (all != 0 && (bn != 0 || keep_probe)) ? all : 0u;
  const uint32_t nun = (pn + JK_PROBE_CHUNK - 1) / JK_PROBE_CHUNK;
  const uint32_t incl_u = wave_scan_incl(nun);
  const unsigned long long incl_t = wave_scan_incl((unsigned long long)pn);
  const unsigned long long sum_all = wave_reduce_add((unsigned long long)all);
  unsigned long long base_u = 0, base_t = 0;
  if (lane_id() == WAVE - 1) {
    if (incl_u) { base_u = atomicAdd(&state[0], (unsigned long long)incl_u); base_t = atomicAdd(&state[1], incl_t); }
    if (sum_all) atomicAdd(&state[2], sum_all);
  }
  base_u = __shfl(base_u, WAVE - 1, WAVE);
  base_t = __shfl(base_t, WAVE - 1, WAVE);
  unsigned long long u = base_u + incl_u - nun, o = base_t + incl_t - pn;
  for (uint32_t at = 0; at < pn; at += JK_PROBE_CHUNK) {
    const uint32_t cnt = pn - at < JK_PROBE_CHUNK ? pn - at : JK_PROBE_CHUNK;
    units[u] = Unit{bb, bn, first + at, cnt};
    off[u] = o;
    ++u;
    o += cnt;
  }
  if (f == 0) state[3] = (unsigned long long)((cursor[nfine] ? 1u : 0u) | ((level1_flag && *level1_flag) ? 2u : 0u));     // 1: level 2 overflowed, 2: level 1
}

struct ProbeArgs {
  Tuples build, probe;          // fine-partitioned tuples
  const Unit *units;
  uint32_t nslots;              // LDS units: H (slots per cuckoo table, power of two); global-table path: slot count
  uint32_t cap;                 // LDS units: capacity of the staged build partition (multiple of 64)
  int keep_unmatched_probe;     // LEFT / FULL: emit (probe, -1)
  int verify;                   // confirm hits on the original columns
  uint8_t *build_matched;       // FULL: byte per build ROW, set when matched (may be null)
  uint64_t *counts;             // COUNT pass output / WRITE pass: exclusive offsets
  int32_t *out_probe; int32_t *out_build;
  int dbg;                      // experiment switch (env GDF_JK_DBG), 0 in production
  uint64_t kbias;               // see PartGeom::kbias
  int optimistic;               // WRITE pass without a count pass: unit u may write at most probe_count pairs
  unsigned long long *opt_state; // [0] = pairs written by all units, [1] = some unit needed more room,
                                 // [2] = units jk_probe_fast left to the general kernel (cuckoo build did not settle),
                                 // [3] = COUNT pass: units whose cuckoo build did not settle (linear probing)
  uint32_t *unit_todo;           // jk_probe_fast appends the ids of such units here (opt_state[2] = how many);
                                 // jk_probe: when non-null, workgroup b handles unit unit_todo[b]
  // COUNT pass as a SAMPLE over device-built units (jk_make_units): workgroup b of sample_n takes unit b * *nunits_dev / sample_n
  // and adds its pairs to opt_state[0] and its probe tuples to opt_state[1] instead of writing counts[]
  uint32_t sample_n;
  const unsigned long long *nunits_dev;
  uint32_t *unit_pairs;          // optimistic WRITE pass, non-null: unit u leaves the number of pairs it wrote here (sparse mode:
                                 // the host compacts the units' slot ranges afterwards, see jk_compact_units)
  // carried payload (PayCarry): pay_mode != 0 -> every pair's probe-side payload column value(s) go to pay_out[c][pos].
  // jk_probe_fast<PMODE> takes them from the payload word next to the probe tuple (probe.pay); the general kernels, which
  // see only the few units the lean kernel left over (and oversize partitions), fetch them from the source columns by row
  int pay_mode;                  // 0: none; 1: one 8-byte column; 2: one 4-byte column; 3: two 4-byte columns; 4: two 8-byte columns (16-byte elements in probe.pay)
  const void *pay_src[2];
  void *pay_out[2];
  // with a carried payload, an INNER join on one integer key column also writes the KEY column of the result (key_width = 8 /
  // 4, else 0): the value is the tuple's key + kbias, no gather; the general kernels read it from the probe column (key_src)
  int key_width;
  const void *key_src;
  void *key_out;
  // the BUILD relation's payload word (INNER joins): bpay_mode as pay_mode; jk_probe_bp takes it from the LDS image of the build
  // partition (build.pay staged next to the tuples), the general kernels from the source columns by build row.  bpay_mode 4: two
  // 8-byte columns -- the first travels with the build tuples as in mode 1, the second is read BY BUILD ROW when a unit stages its
  // partition (bpay_src[1]: ~3000 scattered reads per unit, 1e8 per C3 join, instead of 1e9 gathered values behind the join)
  int bpay_mode;
  const void *bpay_src[2];
  void *bpay_out[2];
  // six-byte probe tuples (p6_store): p6_fb != 0 -> probe.w is a P6 stream of (hash remainder, row) and the kernels (their <P6>
  // instantiations) replace the staged build keys by their remainders -- hash_a(key + p6_kbias) below the top p6_fb bits; kbias is
  // then 0: the "keys" the lookups hash and compare are the remainders
  int p6_fb;
  uint32_t p6_world;             // PartGeom::world of the build side (fused multi-GPU joins hash with the rank remap)
  uint64_t p6_kbias;
};
// the general kernels' payload write: a gather by probe row (rare units only)
__device__ __forceinline__ void pay_gather(const ProbeArgs &a, unsigned long long pos, int32_t prow) {
  if (a.pay_mode == 1 || a.pay_mode == 4) {
    ((uint64_t *)a.pay_out[0])[pos] = ((const uint64_t *)a.pay_src[0])[prow];
    if (a.pay_mode == 4) ((uint64_t *)a.pay_out[1])[pos] = ((const uint64_t *)a.pay_src[1])[prow];
  } else if (a.pay_mode) {
    ((uint32_t *)a.pay_out[0])[pos] = ((const uint32_t *)a.pay_src[0])[prow];
    if (a.pay_mode == 3) ((uint32_t *)a.pay_out[1])[pos] = ((const uint32_t *)a.pay_src[1])[prow];
  }
  if (a.key_width == 8) ((uint64_t *)a.key_out)[pos] = ((const uint64_t *)a.key_src)[prow];
  else if (a.key_width == 4) ((uint32_t *)a.key_out)[pos] = ((const uint32_t *)a.key_src)[prow];
}
__device__ __forceinline__ void bpay_gather(const ProbeArgs &a, unsigned long long pos, int32_t brow) {
  if (a.bpay_mode == 1 || a.bpay_mode == 4) {
    ((uint64_t *)a.bpay_out[0])[pos] = ((const uint64_t *)a.bpay_src[0])[brow];
    if (a.bpay_mode == 4) ((uint64_t *)a.bpay_out[1])[pos] = ((const uint64_t *)a.bpay_src[1])[brow];
  } else if (a.bpay_mode) {
    ((uint32_t *)a.bpay_out[0])[pos] = ((const uint32_t *)a.bpay_src[0])[brow];
    if (a.bpay_mode == 3) ((uint32_t *)a.bpay_out[1])[pos] = ((const uint32_t *)a.bpay_src[1])[brow];
  }
}

// LDS image of one work unit's build partition (dynamic region, every carve 16-byte aligned):
//   bw[cap]   build tuples staged linearly from HBM (position p = index inside the partition);
//             NARROW: the packed word, WIDE: the key
//   T[2*H]    table of POSITIONS.  Cuckoo mode: T[0..H) is table 0 (slot h0), T[H..2H) table 1 (slot h1);
//             linear-probing mode: one table of 2*H slots.
//   bi[cap]   WIDE only: build row numbers
//   misc      per-wave counters, the unit's output cursor, mode flag
//
// Why positions and two tables: a lookup in a cuckoo table is TWO independent reads at slots known up
// front -- straight-line code, no data-dependent loop.  With one lane per probe tuple the open-addressing
// walk of the first version ran, per wave, as many trips as the LONGEST chain among its 256 tuples and
// spent ~80 % of the kernel in that loop (profiles/r1_b_probe_ablation.md).  Cuckoo needs unique-ish keys
// (at most two copies of a key fit); partitions where the insertion does not settle (duplicate build
// keys, or plain bad luck) fall back, per unit, to linear probing over the same LDS image, which keeps
// the multimap semantics of the reference (join_kernels.cuh:259-455).
struct ProbeLds {
  uint64_t *bw;
  uint32_t *T;
  int32_t *bi;
  unsigned long long *wave_cnt;      // [JK_PROBE_THREADS / WAVE]
  unsigned long long *unit_cursor;
  unsigned int *cuckoo_failed;
  uint32_t *next;                    // [cap] general kernel only: chain of build tuples with the same key (JK_NOPOS ends it)
};

template <bool NARROW>
__device__ __forceinline__ ProbeLds carve_probe_lds(unsigned char *raw, uint32_t cap, uint32_t H) {
  ProbeLds l;
  l.bw = (uint64_t *)raw;
  l.T = (uint32_t *)(raw + (size_t)cap * 8);
  unsigned char *after = raw + (size_t)cap * 8 + (size_t)H * 8;
  l.bi = (int32_t *)after;
  if (!NARROW) after += (size_t)cap * 4;
  l.wave_cnt = (unsigned long long *)after;
  l.unit_cursor = l.wave_cnt + JK_PROBE_THREADS / WAVE;
  l.cuckoo_failed = (unsigned int *)(l.unit_cursor + 1);
  l.next = (uint32_t *)(l.unit_cursor + 2);      // present only when the launch asked for probe_lds_bytes(..., chained = true)
  return l;
}
static size_t probe_lds_bytes(bool narrow, uint32_t cap, uint32_t H, bool chained = true) {
  return (size_t)cap * (narrow ? 8 : 12) + (size_t)H * 8 + sizeof(unsigned long long) * (JK_PROBE_THREADS / WAVE + 2) +
         (chained ? (size_t)cap * 4 : 0);
}

template <bool NARROW>
__device__ __forceinline__ int32_t build_row(const ProbeLds &l, uint32_t p) {
  return NARROW ? (int32_t)(uint32_t)l.bw[p] : l.bi[p];
}

template <bool WRITE, bool NARROW, bool P6 = false>
__global__ __launch_bounds__(JK_PROBE_THREADS) void jk_probe(ProbeArgs a, KeyTable probe_t, KeyTable build_t) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t H = a.nslots, cap = a.cap;
  const ProbeLds l = carve_probe_lds<NARROW>(lds_raw, cap, H);
  uint32_t uid = a.unit_todo ? a.unit_todo[blockIdx.x] : blockIdx.x;     // second launch after jk_probe_fast: leftovers only
  if (a.sample_n) {
    const unsigned long long nunits = *a.nunits_dev;
    if (nunits == 0) return;
    uid = (uint32_t)(((unsigned long long)blockIdx.x * nunits) / a.sample_n);
  }
  const Unit u = a.units[uid];

  // ---- stage the build partition, clear the table ----
  for (uint32_t i = threadIdx.x; i < u.build_count; i += JK_PROBE_THREADS) {
    uint64_t w = a.build.w[u.build_begin + i];
    if constexpr (P6 && NARROW) w = ((uint64_t)p6_remainder((uint32_t)(w >> 32), a.p6_kbias, a.p6_fb, a.p6_world) << 32) | (uint32_t)w;
    else if constexpr (P6) w = p10_key(w + a.p6_kbias, a.p6_fb);      // ten-byte probe tuples: compare (hash remainder, high word)
    l.bw[i] = w;
    if (!NARROW) l.bi[i] = a.build.idx[u.build_begin + i];
  }
  for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_PROBE_THREADS) l.T[i] = JK_NOPOS;
  __shared__ uint32_t dup_seen;          // the multimap rebuild met a key a second time
  if (threadIdx.x == 0) { *l.unit_cursor = WRITE ? a.counts[uid] : 0ull; *l.cuckoo_failed = 0; dup_seen = 0; }
  block_sync();

  // ---- cuckoo build: exchange positions until an empty slot absorbs the chain ----
  for (uint32_t p0 = threadIdx.x; p0 < u.build_count; p0 += JK_PROBE_THREADS) {
    uint32_t cur = p0, table = 0;
    int moves = 0;
    for (; moves < JK_CUCKOO_MAX_MOVES; ++moves) {
      const uint64_t raw = tup_key<NARROW>(l.bw[cur]) + a.kbias;
      const uint32_t slot = table ? H + (hash_b(raw) & (H - 1)) : (hash_a(raw) & (H - 1));
      const uint32_t old = atomicExch(&l.T[slot], cur);
      if (old == JK_NOPOS) break;
      cur = old;          // the evicted tuple moves to its other table
      table ^= 1;
    }
    if (moves == JK_CUCKOO_MAX_MOVES) *l.cuckoo_failed = 1;   // `cur` is homeless: rebuild below
  }
  block_sync();
  const bool cuckoo = *l.cuckoo_failed == 0 && !(LAB_BITS(a.dbg) & 8);
  if (!cuckoo) {
    // ---- multimap rebuild over the same 2*H slots: open addressing over the DISTINCT keys, every key's tuples chained
    // behind the one that took the slot (next[]).  Round 1 gave every tuple a slot of its own: a key that occurs four times
    // made clusters four slots long, a lookup walked ~10 slots and, when it had more than one match, walked them again to
    // emit -- 22 ms of LDS chain walking for a join whose build keys all occur four times (profiles/r2_b_bench_shapes.jsonl).
    // Now a lookup finds its key's head in ~1 step (the table holds a quarter of the entries) and walks exactly its matches.
    block_sync();
    for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_PROBE_THREADS) l.T[i] = JK_NOPOS;
    for (uint32_t i = threadIdx.x; i < u.build_count; i += JK_PROBE_THREADS) l.next[i] = JK_NOPOS;
    block_sync();
    const uint32_t mask = 2 * H - 1;
    for (uint32_t p = threadIdx.x; p < u.build_count; p += JK_PROBE_THREADS) {
      const uint64_t kp = tup_key<NARROW>(l.bw[p]);
      uint32_t slot = hash_b(kp + a.kbias) & mask;
      for (;;) {
        uint32_t q = atomicCAS(&l.T[slot], JK_NOPOS, p);
        if (q == JK_NOPOS) break;                                   // p is the head of its key
        if (tup_key<NARROW>(l.bw[q]) == kp) {                        // same key: p goes right behind the head
          l.next[p] = atomicExch(&l.next[q], p);
          dup_seen = 1;
          break;
        }
        slot = (slot + 1) & mask;
      }
    }
    block_sync();
    // sample pass: units whose build keys REPEAT.  A cuckoo build of distinct keys also fails now and then -- at 3800 keys in 2 x 4096
    // slots (a 1.25e8-row build relation) one in four single attempts runs into a cycle; counted as "repeated keys", such joins
    // crossed the quarter-of-the-sample line in a third of the calls and took the general kernel for every unit (10.6 instead of
    // 2.9 ms, profiles/r3_zh_*).  The rebuild knows: it met a key twice, or it did not.
    if (!WRITE && a.opt_state && threadIdx.x == 0 && dup_seen) atomicAdd(&a.opt_state[3], 1ull);
  }

  if (LAB_BITS(a.dbg) & 128) return;      // experiment: build phase only
  // optimistic pass: the unit owns exactly probe_count output slots starting at its offset
  const unsigned long long unit_base = WRITE ? a.counts[uid] : 0ull;
  const unsigned long long unit_end = (WRITE && a.optimistic) ? unit_base + u.probe_count : ~0ull;
  const bool need_row = WRITE || a.verify;
  unsigned long long my_count = 0;
  // NARROW tuples are fetched two at a time (16-byte loads from an even tuple index: `lead` is 1 when the unit starts on an
  // odd one), which doubles the bytes every wave keeps in flight: with the loop streaming 8-byte words the kernel
  // sat at 2.8 TB/s on the read side alone -- latency-bound at two workgroups per CU (profiles/r1_g_probe_ablation.md).
  constexpr int VEC = NARROW ? 2 : 1;
  constexpr int NB = JK_PROBE_BATCH * VEC;
  const uint32_t lead = NARROW ? (u.probe_begin & 1u) : 0u;
  const uint32_t vbegin = u.probe_begin - lead;
  const uint32_t vtotal = lead + u.probe_count;
  for (uint32_t base = 0; base < vtotal; base += JK_PROBE_THREADS * NB) {
    uint64_t k[NB];
    int32_t prow[NB];
    bool act[NB];
#pragma unroll
    for (int b = 0; b < JK_PROBE_BATCH; ++b) {      // all HBM loads first
      if constexpr (NARROW) {
        const uint32_t v = base + (b * JK_PROBE_THREADS + threadIdx.x) * 2;
        const uint32_t last_pair = (vtotal - 1) & ~1u;
        ulonglong2 ww;
        if constexpr (P6) {
          uint32_t r0, row0, r1, row1;
          p6_load_pair(a.probe.w, vbegin + (v < last_pair ? v : last_pair), r0, row0, r1, row1);
          ww.x = ((uint64_t)r0 << 32) | row0;
          ww.y = ((uint64_t)r1 << 32) | row1;
        } else {
          ww = *reinterpret_cast<const ulonglong2 *>(a.probe.w + vbegin + (v < last_pair ? v : last_pair));   // clamped, unconditional
        }
        k[2 * b] = tup_key<NARROW>(ww.x);
        prow[2 * b] = (int32_t)(uint32_t)ww.x;
        act[2 * b] = v >= lead && v < vtotal;
        k[2 * b + VEC - 1] = tup_key<NARROW>(ww.y);
        prow[2 * b + VEC - 1] = (int32_t)(uint32_t)ww.y;
        act[2 * b + VEC - 1] = v + 1 < vtotal;
      } else {
        const uint32_t i = base + b * JK_PROBE_THREADS + threadIdx.x;
        const uint32_t ic = u.probe_begin + (i < u.probe_count ? i : u.probe_count - 1);   // clamped, unconditional
        if constexpr (P6) {            // ten-byte tuples: (remainder, row) from the six-byte stream, the high word from the parallel array
          uint32_t r, row;
          p6_load_one(a.probe.w, ic, r, row);
          k[b] = ((uint64_t)r << 32) | (uint32_t)a.probe.idx[ic];
          prow[b] = (int32_t)row;
        } else {
          k[b] = a.probe.w[ic];
          prow[b] = need_row ? a.probe.idx[ic] : 0;
        }
        act[b] = i < u.probe_count;
      }
    }
    uint32_t cnt[NB];
    uint32_t hit_a[NB], hit_b[NB];   // matching build positions (cuckoo mode: at most two)
    if (cuckoo) {
      uint32_t pa[NB], pb[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {    // 8 independent table reads
        const uint64_t raw = k[b] + a.kbias;
        pa[b] = l.T[hash_a(raw) & (H - 1)];
        pb[b] = l.T[H + (hash_b(raw) & (H - 1))];
        if (LAB_BITS(a.dbg) & 64) { pa[b] = (uint32_t)k[b] & 1023u; pb[b] = JK_NOPOS; }      // experiment: no table lookups
      }
      uint64_t ka[NB], kb[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {    // 8 independent key reads (position 0 stands in for "empty")
        ka[b] = tup_key<NARROW>(l.bw[pa[b] == JK_NOPOS ? 0 : pa[b]]);
        kb[b] = tup_key<NARROW>(l.bw[pb[b] == JK_NOPOS ? 0 : pb[b]]);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const bool active = act[b];
        bool ha = active && pa[b] != JK_NOPOS && ka[b] == k[b];
        bool hb = active && pb[b] != JK_NOPOS && kb[b] == k[b];
        if (a.verify) {
          if (ha) ha = rows_equal(probe_t, prow[b], build_t, build_row<NARROW>(l, pa[b]));
          if (hb) hb = rows_equal(probe_t, prow[b], build_t, build_row<NARROW>(l, pb[b]));
        }
        hit_a[b] = ha ? pa[b] : (hb ? pb[b] : JK_NOPOS);
        hit_b[b] = (ha && hb) ? pb[b] : JK_NOPOS;
        cnt[b] = (uint32_t)ha + (uint32_t)hb;
        if (a.build_matched) {
          if (ha) a.build_matched[build_row<NARROW>(l, pa[b])] = 1;
          if (hb) a.build_matched[build_row<NARROW>(l, pb[b])] = 1;
        }
      }
    } else {
      const uint32_t mask = 2 * H - 1;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        cnt[b] = 0;
        hit_a[b] = hit_b[b] = JK_NOPOS;
        if (act[b]) {
          uint32_t slot = hash_b(k[b] + a.kbias) & mask;
          uint32_t head = JK_NOPOS;
          for (;;) {                                     // the head of this key, if the partition holds it
            const uint32_t p = l.T[slot];
            if (p == JK_NOPOS) break;
            if (tup_key<NARROW>(l.bw[p]) == k[b]) { head = p; break; }
            slot = (slot + 1) & mask;
          }
          hit_b[b] = head;                               // the write pass starts its walk here
          for (uint32_t p = head; p != JK_NOPOS; p = l.next[p]) {
            if (!a.verify || rows_equal(probe_t, prow[b], build_t, build_row<NARROW>(l, p))) {
              if (cnt[b] == 0) hit_a[b] = p;
              ++cnt[b];
              if (a.build_matched) a.build_matched[build_row<NARROW>(l, p)] = 1;
            }
          }
        }
      }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool active = act[b];
      uint32_t c = cnt[b];
      const bool pad = active && c == 0 && a.keep_unmatched_probe;   // LEFT / FULL: (probe, -1)
      if (pad) c = 1;
      if (!WRITE) {
        my_count += c;
      } else {
        // wave-level compaction.  Common case (every lane emits 0 or 1 pair): ballot + popcount;
        // otherwise an exclusive scan of the per-lane counts.  One LDS atomic per wave claims the range.
        unsigned long long pos;
        if (__all(c <= 1)) {
          const unsigned long long mm = __ballot(c == 1);
          unsigned long long wbase = 0;
          if (lane_id() == 0 && mm) wbase = atomicAdd(l.unit_cursor, (unsigned long long)__popcll(mm));
          pos = __shfl(wbase, 0, WAVE) + mask_rank(mm);
        } else {
          const uint32_t incl = wave_scan_incl(c);
          const uint32_t wave_total = __shfl(incl, WAVE - 1, WAVE);
          unsigned long long wbase = 0;
          if (lane_id() == 0) wbase = atomicAdd(l.unit_cursor, (unsigned long long)wave_total);
          pos = __shfl(wbase, 0, WAVE) + incl - c;
        }
        if (pos + c > unit_end) {
          if (c) a.opt_state[1] = 1;     // would spill into the next unit's slots: the host redoes the join two-pass
        } else if (pad) {
          a.out_probe[pos] = prow[b];
          a.out_build[pos] = JK_EMPTY;
          pay_gather(a, pos, prow[b]);
        } else if (c >= 1 && (cuckoo || c == 1)) {
          if (LAB_BITS(a.dbg) & 32) continue;                  // experiment: no output stores
          a.out_probe[pos] = prow[b];
          a.out_build[pos] = build_row<NARROW>(l, hit_a[b]);
          pay_gather(a, pos, prow[b]);
          bpay_gather(a, pos, build_row<NARROW>(l, hit_a[b]));
          if (c == 2) {
            a.out_probe[pos + 1] = prow[b];
            a.out_build[pos + 1] = build_row<NARROW>(l, hit_b[b]);
            pay_gather(a, pos + 1, prow[b]);
            bpay_gather(a, pos + 1, build_row<NARROW>(l, hit_b[b]));
          }
        } else if (c > 1) {      // multimap mode with several matches: walk the key's chain again
          for (uint32_t p = hit_b[b]; p != JK_NOPOS; p = l.next[p]) {
            if (!a.verify || rows_equal(probe_t, prow[b], build_t, build_row<NARROW>(l, p))) {
              a.out_probe[pos] = prow[b];
              a.out_build[pos] = build_row<NARROW>(l, p);
              pay_gather(a, pos, prow[b]);
              bpay_gather(a, pos, build_row<NARROW>(l, p));
              ++pos;
            }
          }
        }
      }
    }
  }
  if (WRITE && a.optimistic) {
    block_sync();
    if (threadIdx.x == 0) {
      atomicAdd(&a.opt_state[0], *l.unit_cursor - unit_base);
      if (a.unit_pairs) a.unit_pairs[uid] = (uint32_t)(*l.unit_cursor - unit_base);
    }
  }
  if (!WRITE) {
    my_count = wave_reduce_add(my_count);
    if (lane_id() == 0) l.wave_cnt[threadIdx.x / WAVE] = my_count;
    block_sync();
    if (threadIdx.x == 0) {
      unsigned long long t = 0;
      for (int w = 0; w < JK_PROBE_THREADS / WAVE; ++w) t += l.wave_cnt[w];
      if (a.sample_n) { atomicAdd(&a.opt_state[0], t); atomicAdd(&a.opt_state[1], (unsigned long long)u.probe_count); }
      else a.counts[uid] = t;
    }
  }
}

// ---------------------------------------------------------------------------
// 4b. The same write pass for the PLAIN case -- INNER or LEFT join, NARROW tuples, exact keys (no verification on
// the original columns), no FULL-join marks -- which is what C3 and every foreign-key join on <= 32-bit-range
// keys runs.  jk_probe carries all the other cases in one body; per probe tuple it issued ~98 VALU + ~60 SALU
// instructions and the kernel was VALU-bound at 4.2 ms (profiles/r1_g_probe_ablation.md).  Here: 32-bit keys and
// unit-local 32-bit output positions (one scalar base pointer per array), the build tuple is read from LDS once
// (key and row in one 64-bit word), no runtime switches in the loop.  A unit whose cuckoo build does not settle
// is not handled here: it is flagged in unit_todo and the host runs jk_probe over the flagged units.
// ---------------------------------------------------------------------------
// POW2: H is a power of two and a slot is the top log2(H) bits of the hash product; otherwise H is any size
// (chosen by the host for a 40 % table load) and a slot is mulhi(hash product, H).
// KEEP: LEFT join -- a probe tuple without a match emits (probe row, -1).
// NARROW = false: the same lean pass over WIDE tuples (exact 64-bit keys: one 8-byte integer / float column, or several
// columns packed into 64 bits) -- keys spread over more than 2^32 used to take the general kernel at 8.9 ms per 1e9 probe
// tuples (profiles/r2_b_bench_shapes.jsonl, c3_wide_keys).
// PMODE != 0 (NARROW only): the probe tuples carry a payload word (Tuples::pay) and every pair writes its column value(s)
// next to its index pair -- the streaming replacement of the probe-side gather of a materialising join.
template <bool POW2, bool KEEP, bool NARROW, int PMODE = 0, bool P6 = false>
__global__ __launch_bounds__(JK_PROBE_THREADS) void jk_probe_fast(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t H = a.nslots, cap = a.cap;
  const ProbeLds l = carve_probe_lds<NARROW>(lds_raw, cap, H);
  unsigned int *lcur = (unsigned int *)l.unit_cursor;      // pairs written so far by this unit
  const Unit u = a.units[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < u.build_count; i += JK_PROBE_THREADS) {
    uint64_t w = a.build.w[u.build_begin + i];
    if constexpr (P6 && NARROW) w = ((uint64_t)p6_remainder((uint32_t)(w >> 32), a.p6_kbias, a.p6_fb, a.p6_world) << 32) | (uint32_t)w;      // six-byte probe tuples: compare remainders
    else if constexpr (P6) w = p10_key(w + a.p6_kbias, a.p6_fb);                                                                                        // ten-byte ones: (remainder, high word)
    l.bw[i] = w;
    if (!NARROW) l.bi[i] = a.build.idx[u.build_begin + i];
  }
  for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_PROBE_THREADS) l.T[i] = JK_NOPOS;
  if (threadIdx.x == 0) { *lcur = 0; *l.cuckoo_failed = 0; }
  block_sync();
  const uint32_t kb_lo = (uint32_t)a.kbias, kb_hi = (uint32_t)(a.kbias >> 32);
  // raw key = stored key + kbias; fold = lo ^ hi * C (key_fold).  NARROW keys are 32 bits, WIDE keys 64.
  using Key = typename std::conditional<NARROW, uint32_t, uint64_t>::type;
  auto fold_of = [&](Key key) -> uint32_t {
    if constexpr (P6 && !NARROW) {
      // ten-byte tuples: the "key" is (17-bit hash remainder) << 32 | the raw key's high word.  The remainder is spread as below;
      // xor-ed with the high word it tells the keys of a partition apart whether the high words vary (keys spread over 2^62) or
      // not (a dense range above 2^32) -- two of them collide in this fold with probability 2^-32
      uint32_t x = (uint32_t)(key >> 32);
      x ^= x << 13;
      x ^= x >> 7;
      return x ^ (uint32_t)key;
    } else if constexpr (P6) {
      // six-byte tuples: the "key" is a 17-bit hash remainder.  Two multiplicative slot hashes of such a small DENSE set (3 % of the
      // 2^17 values) share their bad differences -- 8 to 12 % of the partitions needed a second cuckoo attempt, and at 3800 keys per
      // partition a few units per join failed all four and went to the general kernel alone (0.13 ms of tail).  Two xor-shifts
      // spread the remainder over the word first: 0 failures in 600 simulated partitions.
      uint32_t x = (uint32_t)key;
      x ^= x << 13;
      x ^= x >> 7;
      return x;
    } else if constexpr (NARROW) {
      const uint32_t lo = key + kb_lo;
      const uint32_t hi = kb_hi + (lo < key ? 1u : 0u);
      return lo ^ (hi * 0x9e3779b1u);
    } else {
      return key_fold(key + a.kbias);
    }
  };
  // the fold table 1 hashes: NARROW keys are told apart by the one fold (it is injective on them), WIDE keys need the second
  auto fold2_of = [&](Key key, uint32_t f1) -> uint32_t {
    if constexpr (NARROW) return f1;
    else if constexpr (P6) {
      // the second fold of a ten-byte key: the same two words combined at another rotation (keys that share fold_of share this one
      // only if their remainders' spreads differ by a 16-bit-periodic word) -- no multiply, where the 12-byte tuples' two folds cost two
      uint32_t x = (uint32_t)(key >> 32);
      x ^= x << 13;
      x ^= x >> 7;
      return __builtin_rotateleft32(x, 16) + (uint32_t)key + ((uint32_t)key << 3);
    } else return key_fold2(key + a.kbias);
  };
  auto staged_key = [&](uint32_t p) -> Key {
    if constexpr (NARROW) return (uint32_t)(l.bw[p] >> 32);
    else return l.bw[p];
  };
  // Slot hashes: the TOP log2(H) bits of two multiplicative hashes of the folded key.  One quarter-rate 32-bit
  // multiply each instead of lowbias32's two multiplies and three xor-shifts: the keys of one partition are
  // already a pseudo-random subset (the partition id comes from lowbias32), so the tables only need two
  // different well-spread maps, and a build that does not settle is retried with another seed anyway.
  const int hshift = POW2 ? 32 - (__ffs((int)H) - 1) : 0;
  // A cuckoo build that runs into a cycle (17 of C3's 32768 partitions with one fixed pair of hash functions) is
  // repeated with another pair: `seed` perturbs the folded key before both slot hashes.  What still fails after
  // four attempts holds a key more than twice and belongs to the general kernel's linear probing.
  uint32_t seed = 0;
  for (int attempt = 0; attempt < 4; ++attempt) {
    for (uint32_t p0 = threadIdx.x; p0 < u.build_count; p0 += JK_PROBE_THREADS) {
      uint32_t cur = p0, table = 0;
      int moves = 0;
      for (; moves < JK_CUCKOO_MAX_MOVES; ++moves) {
        const Key ck = staged_key(cur);
        const uint32_t f = fold_of(ck) ^ seed, f2 = fold2_of(ck, fold_of(ck)) ^ seed;
        const uint32_t slot = table ? H + (POW2 ? (f2 * 0xc2b2ae35u) >> hshift : __umulhi(f2 * 0xc2b2ae35u, H))
                                    : (POW2 ? (f * 0x9e3779b1u) >> hshift : __umulhi(f * 0x9e3779b1u, H));
        const uint32_t old = atomicExch(&l.T[slot], cur);
        if (old == JK_NOPOS) break;
        cur = old;
        table ^= 1;
      }
      if (moves == JK_CUCKOO_MAX_MOVES) *l.cuckoo_failed = 1;
    }
    block_sync();
    if (!*l.cuckoo_failed || attempt == 3) break;
    block_sync();                                           // everyone has read the flag
    for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_PROBE_THREADS) l.T[i] = JK_NOPOS;
    if (threadIdx.x == 0) *l.cuckoo_failed = 0;
    seed += 0x9e3779b9u;
    block_sync();
  }
  if (*l.cuckoo_failed | (unsigned)(LAB_BITS(a.dbg) & 8)) {
    if (threadIdx.x == 0) a.unit_todo[atomicAdd(&a.opt_state[2], 1ull)] = blockIdx.x;
    return;
  }
  const unsigned long long unit_base = a.counts[blockIdx.x];
  const uint32_t unit_cap = a.optimistic ? u.probe_count : 0xffffffffu;
  int32_t *__restrict__ op = a.out_probe + unit_base;
  int32_t *__restrict__ ob = a.out_build + unit_base;
  constexpr int NB = JK_PROBE_BATCH * 2;
  const uint32_t lead = u.probe_begin & 1u;
  const uint64_t *__restrict__ src = a.probe.w + (u.probe_begin - lead);
  const int32_t *__restrict__ src_row = NARROW ? nullptr : a.probe.idx + (u.probe_begin - lead);
  const uint64_t *__restrict__ src_pay = PMODE ? a.probe.pay + (size_t)(u.probe_begin - lead) * (PMODE == 4 ? 2 : 1) : nullptr;
  uint64_t *__restrict__ po8 = (PMODE == 1 || PMODE == 4) ? (uint64_t *)a.pay_out[0] + unit_base : nullptr;
  uint64_t *__restrict__ po8b = PMODE == 4 ? (uint64_t *)a.pay_out[1] + unit_base : nullptr;
  uint32_t *__restrict__ po4a = (PMODE == 2 || PMODE == 3) ? (uint32_t *)a.pay_out[0] + unit_base : nullptr;
  uint32_t *__restrict__ po4b = PMODE == 3 ? (uint32_t *)a.pay_out[1] + unit_base : nullptr;
  uint64_t *__restrict__ ko8 = (PMODE && a.key_width == 8) ? (uint64_t *)a.key_out + unit_base : nullptr;
  uint32_t *__restrict__ ko4 = (PMODE && a.key_width == 4) ? (uint32_t *)a.key_out + unit_base : nullptr;
  const uint32_t vtotal = lead + u.probe_count;
  const uint32_t last_pair = (vtotal - 1) & ~1u;
  for (uint32_t base = 0; base < vtotal; base += JK_PROBE_THREADS * NB) {
    Key key[NB];
    uint32_t prow[NB];
    uint64_t pay[PMODE ? NB : 1], pay2[PMODE == 4 ? NB : 1];
    bool act[NB];
#pragma unroll
    for (int b = 0; b < JK_PROBE_BATCH; ++b) {      // all HBM loads first, 16 bytes each (WIDE: + 8 bytes of row numbers), clamped and unconditional
      const uint32_t v = base + (b * JK_PROBE_THREADS + threadIdx.x) * 2;
      const uint32_t vc = v < last_pair ? v : last_pair;
      ulonglong2 ww{};
      if constexpr (P6) {            // one 12-byte load: two (remainder, row) tuples
        uint32_t r0, row0, r1, row1;
        p6_load_pair(a.probe.w, (u.probe_begin - lead) + vc, r0, row0, r1, row1);
        if constexpr (NARROW) {
          key[2 * b] = r0; key[2 * b + 1] = r1;
        } else {                     // ten-byte tuples: + one 8-byte load of the two high words
          const HiPair hh = *reinterpret_cast<const HiPair *>(a.probe.idx + (u.probe_begin - lead) + vc);
          key[2 * b] = ((uint64_t)r0 << 32) | hh.a; key[2 * b + 1] = ((uint64_t)r1 << 32) | hh.b;
        }
        prow[2 * b] = row0; prow[2 * b + 1] = row1;
      } else {
        ww = *reinterpret_cast<const ulonglong2 *>(src + vc);
      }
      if constexpr (PMODE == 4) {             // two payload words per tuple: one 16-byte element each
        const ulonglong2 p0 = *reinterpret_cast<const ulonglong2 *>(src_pay + 2 * (size_t)vc);
        const ulonglong2 p1 = *reinterpret_cast<const ulonglong2 *>(src_pay + 2 * (size_t)vc + 2);
        pay[2 * b] = p0.x; pay2[2 * b] = p0.y; pay[2 * b + 1] = p1.x; pay2[2 * b + 1] = p1.y;
      } else if constexpr (PMODE != 0) {
        const ulonglong2 pp = *reinterpret_cast<const ulonglong2 *>(src_pay + vc);
        pay[2 * b] = pp.x; pay[2 * b + 1] = pp.y;
      }
      if constexpr (P6) {
      } else if constexpr (NARROW) {
        key[2 * b] = (uint32_t)(ww.x >> 32); prow[2 * b] = (uint32_t)ww.x;
        key[2 * b + 1] = (uint32_t)(ww.y >> 32); prow[2 * b + 1] = (uint32_t)ww.y;
      } else {
        const uint2 rr = *reinterpret_cast<const uint2 *>(src_row + vc);
        key[2 * b] = ww.x; prow[2 * b] = rr.x;
        key[2 * b + 1] = ww.y; prow[2 * b + 1] = rr.y;
      }
      act[2 * b] = v >= lead && v < vtotal;
      act[2 * b + 1] = v + 1 < vtotal;
    }
    uint32_t pa[NB], pb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {                  // 2 independent table reads per tuple
      const uint32_t f = fold_of(key[b]) ^ seed, f2 = fold2_of(key[b], fold_of(key[b])) ^ seed;
      pa[b] = l.T[POW2 ? (f * 0x9e3779b1u) >> hshift : __umulhi(f * 0x9e3779b1u, H)];
      pb[b] = l.T[H + (POW2 ? (f2 * 0xc2b2ae35u) >> hshift : __umulhi(f2 * 0xc2b2ae35u, H))];
    }
    uint64_t wa[NB], wb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {                  // 2 independent tuple reads (position 0 stands in for "empty")
      wa[b] = l.bw[pa[b] == JK_NOPOS ? 0 : pa[b]];
      wb[b] = l.bw[pb[b] == JK_NOPOS ? 0 : pb[b]];
    }
    // hits as bit masks (bit b: tuple b of this lane): table 0 / table 1 / no partner but kept (LEFT)
    uint32_t hamask = 0, hbmask = 0, padmask = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool ha = act[b] && pa[b] != JK_NOPOS && (NARROW ? (uint32_t)(wa[b] >> 32) == (uint32_t)key[b] : wa[b] == (uint64_t)key[b]);
      const bool hb = act[b] && pb[b] != JK_NOPOS && (NARROW ? (uint32_t)(wb[b] >> 32) == (uint32_t)key[b] : wb[b] == (uint64_t)key[b]);
      hamask |= (uint32_t)ha << b;
      hbmask |= (uint32_t)hb << b;
      padmask |= (uint32_t)(KEEP && act[b] && !ha && !hb) << b;
    }
    // WIDE: the build row of a tuple's hit comes from the staged row numbers -- NB independent LDS reads, issued together (round 6: read
    // inside emit(), under its per-tuple branch, every one of them was a round trip of its own: 4.6 - 4.7 ms per 1e9 probe tuples
    // against the NARROW kernel's 2.7).  A tuple with a hit in BOTH tables (a build key present twice) reads its second row in emit()
    int32_t hitrow[NARROW ? 1 : NB];
    if constexpr (!NARROW) {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const bool ha = (hamask >> b) & 1u, hb = (hbmask >> b) & 1u;
        hitrow[b] = l.bi[ha ? pa[b] : (hb ? pb[b] : 0u)];
      }
    }
    auto emit = [&](int b, uint32_t pos, uint32_t c) {
      const bool ha = (hamask >> b) & 1u, pad = (padmask >> b) & 1u;
      if (pos + c > unit_cap) a.opt_state[1] = 1;   // would spill into the next unit's slots: the host redoes the join two-pass
      else if (!(LAB_BITS(a.dbg) & 32)) {
        // build row of a hit: NARROW carries it in the low half of the staged word, WIDE reads it from the staged row numbers
        // (WIDE: the second row of a tuple with two hits is read under c == 2 only -- as a value selected next to hitrow[b] the
        // compiler issued the read for every emitted tuple, one LDS round trip each)
        int32_t first;
        if constexpr (NARROW) first = ha ? (int32_t)(uint32_t)wa[b] : (int32_t)(uint32_t)wb[b];
        else first = hitrow[b];
        op[pos] = (int32_t)prow[b];
        ob[pos] = pad ? JK_EMPTY : first;
        if (c == 2) {
          op[pos + 1] = (int32_t)prow[b];
          if constexpr (NARROW) ob[pos + 1] = (int32_t)(uint32_t)wb[b];
          else ob[pos + 1] = l.bi[pb[b]];
        }
        if constexpr (PMODE == 1 || PMODE == 4) { po8[pos] = pay[b]; if (c == 2) po8[pos + 1] = pay[b]; }
        if constexpr (PMODE == 4) { po8b[pos] = pay2[b]; if (c == 2) po8b[pos + 1] = pay2[b]; }
        if constexpr (PMODE == 2 || PMODE == 3) { po4a[pos] = (uint32_t)pay[b]; if (c == 2) po4a[pos + 1] = (uint32_t)pay[b]; }
        if constexpr (PMODE == 3) { po4b[pos] = (uint32_t)(pay[b] >> 32); if (c == 2) po4b[pos + 1] = (uint32_t)(pay[b] >> 32); }
        if constexpr (PMODE != 0 && NARROW) {          // the result's key column (workgroup-uniform branches)
          if (ko8) { ko8[pos] = (uint64_t)key[b] + a.kbias; if (c == 2) ko8[pos + 1] = (uint64_t)key[b] + a.kbias; }
          if (ko4) { ko4[pos] = (uint32_t)key[b]; if (c == 2) ko4[pos + 1] = (uint32_t)key[b]; }
        }
      }
    };
    if (__all((hamask & hbmask) == 0) && !(LAB_BITS(a.dbg) & 2048)) {      // (GDF_JK_DBG=2048: one claim per tuple, as before)
      // nobody has two pairs for one tuple (no key sits in both tables: the usual case).  ONE claim per wave and batch: the NB
      // ballots give every tuple its offset inside the wave's range, lane 0 claims the sum, and the base comes back through
      // readfirstlane -- per tuple a claim + a bpermute were two LDS round trips, sixteen per batch, on the critical path of a
      // kernel that otherwise streams
      const uint32_t onemask = hamask | hbmask | padmask;
      unsigned long long mm[NB];
      uint32_t off[NB], tot = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        mm[b] = __ballot((onemask >> b) & 1u);
        off[b] = tot;
        tot += (uint32_t)__popcll(mm[b]);
      }
      uint32_t wbase = 0;
      if (lane_id() == 0 && tot) wbase = atomicAdd(lcur, tot);
      wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if ((onemask >> b) & 1u) emit(b, wbase + off[b] + (uint32_t)mask_rank(mm[b]), 1u);
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint32_t c = ((hamask >> b) & 1u) + ((hbmask >> b) & 1u) + ((padmask >> b) & 1u);
        uint32_t pos;
        if (__all(c <= 1)) {
          const unsigned long long mm = __ballot(c == 1);
          uint32_t wbase = 0;
          if (lane_id() == 0 && mm) wbase = atomicAdd(lcur, (unsigned int)__popcll(mm));
          pos = __shfl(wbase, 0, WAVE) + mask_rank(mm);
        } else {
          const uint32_t incl = wave_scan_incl(c);
          const uint32_t wave_total = __shfl(incl, WAVE - 1, WAVE);
          uint32_t wbase = 0;
          if (lane_id() == 0) wbase = atomicAdd(lcur, wave_total);
          pos = __shfl(wbase, 0, WAVE) + incl - c;
        }
        if (c) emit(b, pos, c);
      }
    }
  }
  if (a.optimistic) {
    block_sync();
    if (threadIdx.x == 0) {
      atomicAdd(&a.opt_state[0], (unsigned long long)*lcur);
      if (a.unit_pairs) a.unit_pairs[blockIdx.x] = *lcur;
    }
  }
}

// The COUNT pass of the same plain case: jk_probe_fast's staging, cuckoo build and lookups, no output side (a unit that does not
// settle here or there goes to the general kernel for that pass; either way it is counted / written completely).  Joins between ~55 % and 100 % hits (and LEFT joins) need exact output sizes before
// they write; the general kernel's count pass is VALU-bound at 2.3 ms per 1e9 probe tuples, this one reads
// its 8 bytes per tuple at the HBM rate.  Units it cannot settle go to unit_todo / opt_state[2] like the write kernel's.
template <bool POW2, bool KEEP, bool NARROW, bool P6 = false>
__global__ __launch_bounds__(JK_PROBE_THREADS) void jk_count_fast(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t H = a.nslots, cap = a.cap;
  const ProbeLds l = carve_probe_lds<NARROW>(lds_raw, cap, H);
  const Unit u = a.units[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < u.build_count; i += JK_PROBE_THREADS) {
    uint64_t w = a.build.w[u.build_begin + i];
    if constexpr (P6 && NARROW) w = ((uint64_t)p6_remainder((uint32_t)(w >> 32), a.p6_kbias, a.p6_fb, a.p6_world) << 32) | (uint32_t)w;
    else if constexpr (P6) w = p10_key(w + a.p6_kbias, a.p6_fb);
    l.bw[i] = w;
    if (!NARROW) l.bi[i] = a.build.idx[u.build_begin + i];
  }
  for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_PROBE_THREADS) l.T[i] = JK_NOPOS;
  if (threadIdx.x == 0) *l.cuckoo_failed = 0;
  block_sync();
  const uint32_t kb_lo = (uint32_t)a.kbias, kb_hi = (uint32_t)(a.kbias >> 32);
  // raw key = stored key + kbias; fold = lo ^ hi * C (key_fold).  NARROW keys are 32 bits, WIDE keys 64.
  using Key = typename std::conditional<NARROW, uint32_t, uint64_t>::type;
  auto fold_of = [&](Key key) -> uint32_t {
    if constexpr (P6 && !NARROW) {
      // ten-byte tuples: the "key" is (17-bit hash remainder) << 32 | the raw key's high word.  The remainder is spread as below;
      // xor-ed with the high word it tells the keys of a partition apart whether the high words vary (keys spread over 2^62) or
      // not (a dense range above 2^32) -- two of them collide in this fold with probability 2^-32
      uint32_t x = (uint32_t)(key >> 32);
      x ^= x << 13;
      x ^= x >> 7;
      return x ^ (uint32_t)key;
    } else if constexpr (P6) {
      // six-byte tuples: the "key" is a 17-bit hash remainder.  Two multiplicative slot hashes of such a small DENSE set (3 % of the
      // 2^17 values) share their bad differences -- 8 to 12 % of the partitions needed a second cuckoo attempt, and at 3800 keys per
      // partition a few units per join failed all four and went to the general kernel alone (0.13 ms of tail).  Two xor-shifts
      // spread the remainder over the word first: 0 failures in 600 simulated partitions.
      uint32_t x = (uint32_t)key;
      x ^= x << 13;
      x ^= x >> 7;
      return x;
    } else if constexpr (NARROW) {
      const uint32_t lo = key + kb_lo;
      const uint32_t hi = kb_hi + (lo < key ? 1u : 0u);
      return lo ^ (hi * 0x9e3779b1u);
    } else {
      return key_fold(key + a.kbias);
    }
  };
  // the fold table 1 hashes: NARROW keys are told apart by the one fold (it is injective on them), WIDE keys need the second
  auto fold2_of = [&](Key key, uint32_t f1) -> uint32_t {
    if constexpr (NARROW) return f1;
    else if constexpr (P6) {
      // the second fold of a ten-byte key: the same two words combined at another rotation (keys that share fold_of share this one
      // only if their remainders' spreads differ by a 16-bit-periodic word) -- no multiply, where the 12-byte tuples' two folds cost two
      uint32_t x = (uint32_t)(key >> 32);
      x ^= x << 13;
      x ^= x >> 7;
      return __builtin_rotateleft32(x, 16) + (uint32_t)key + ((uint32_t)key << 3);
    } else return key_fold2(key + a.kbias);
  };
  auto staged_key = [&](uint32_t p) -> Key {
    if constexpr (NARROW) return (uint32_t)(l.bw[p] >> 32);
    else return l.bw[p];
  };
  // Slot hashes: the TOP log2(H) bits of two multiplicative hashes of the folded key.  One quarter-rate 32-bit
  // multiply each instead of lowbias32's two multiplies and three xor-shifts: the keys of one partition are
  // already a pseudo-random subset (the partition id comes from lowbias32), so the tables only need two
  // different well-spread maps, and a build that does not settle is retried with another seed anyway.
  const int hshift = POW2 ? 32 - (__ffs((int)H) - 1) : 0;
  // A cuckoo build that runs into a cycle (17 of C3's 32768 partitions with one fixed pair of hash functions) is
  // repeated with another pair: `seed` perturbs the folded key before both slot hashes.  What still fails after
  // four attempts holds a key more than twice and belongs to the general kernel's linear probing.
  uint32_t seed = 0;
  for (int attempt = 0; attempt < 4; ++attempt) {
    for (uint32_t p0 = threadIdx.x; p0 < u.build_count; p0 += JK_PROBE_THREADS) {
      uint32_t cur = p0, table = 0;
      int moves = 0;
      for (; moves < JK_CUCKOO_MAX_MOVES; ++moves) {
        const Key ck = staged_key(cur);
        const uint32_t f = fold_of(ck) ^ seed, f2 = fold2_of(ck, fold_of(ck)) ^ seed;
        const uint32_t slot = table ? H + (POW2 ? (f2 * 0xc2b2ae35u) >> hshift : __umulhi(f2 * 0xc2b2ae35u, H))
                                    : (POW2 ? (f * 0x9e3779b1u) >> hshift : __umulhi(f * 0x9e3779b1u, H));
        const uint32_t old = atomicExch(&l.T[slot], cur);
        if (old == JK_NOPOS) break;
        cur = old;
        table ^= 1;
      }
      if (moves == JK_CUCKOO_MAX_MOVES) *l.cuckoo_failed = 1;
    }
    block_sync();
    if (!*l.cuckoo_failed || attempt == 3) break;
    block_sync();                                           // everyone has read the flag
    for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_PROBE_THREADS) l.T[i] = JK_NOPOS;
    if (threadIdx.x == 0) *l.cuckoo_failed = 0;
    seed += 0x9e3779b9u;
    block_sync();
  }
  if (*l.cuckoo_failed | (unsigned)(LAB_BITS(a.dbg) & 8)) {
    if (threadIdx.x == 0) a.unit_todo[atomicAdd(&a.opt_state[2], 1ull)] = blockIdx.x;
    return;
  }
  // count: the write kernel's lookups without its output side
  constexpr int NB = JK_PROBE_BATCH * 2;
  const uint32_t lead = u.probe_begin & 1u;
  const uint64_t *__restrict__ src = a.probe.w + (u.probe_begin - lead);
  const uint32_t vtotal = lead + u.probe_count;
  const uint32_t last_pair = (vtotal - 1) & ~1u;
  uint32_t mine = 0;
  for (uint32_t base = 0; base < vtotal; base += JK_PROBE_THREADS * NB) {
    Key key[NB];
    bool act[NB];
#pragma unroll
    for (int b = 0; b < JK_PROBE_BATCH; ++b) {
      const uint32_t v = base + (b * JK_PROBE_THREADS + threadIdx.x) * 2;
      const uint32_t vc = v < last_pair ? v : last_pair;
      if constexpr (P6) {
        uint32_t r0, row0, r1, row1;
        p6_load_pair(a.probe.w, (u.probe_begin - lead) + vc, r0, row0, r1, row1);
        if constexpr (NARROW) {
          key[2 * b] = r0; key[2 * b + 1] = r1;
        } else {
          const HiPair hh = *reinterpret_cast<const HiPair *>(a.probe.idx + (u.probe_begin - lead) + vc);
          key[2 * b] = ((uint64_t)r0 << 32) | hh.a; key[2 * b + 1] = ((uint64_t)r1 << 32) | hh.b;
        }
      } else {
        const ulonglong2 ww = *reinterpret_cast<const ulonglong2 *>(src + vc);
        if constexpr (NARROW) { key[2 * b] = (uint32_t)(ww.x >> 32); key[2 * b + 1] = (uint32_t)(ww.y >> 32); }
        else { key[2 * b] = ww.x; key[2 * b + 1] = ww.y; }
      }
      act[2 * b] = v >= lead && v < vtotal;
      act[2 * b + 1] = v + 1 < vtotal;
    }
    uint32_t pa[NB], pb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t f = fold_of(key[b]) ^ seed, f2 = fold2_of(key[b], fold_of(key[b])) ^ seed;
      pa[b] = l.T[POW2 ? (f * 0x9e3779b1u) >> hshift : __umulhi(f * 0x9e3779b1u, H)];
      pb[b] = l.T[H + (POW2 ? (f2 * 0xc2b2ae35u) >> hshift : __umulhi(f2 * 0xc2b2ae35u, H))];
    }
    uint64_t wa[NB], wb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      wa[b] = l.bw[pa[b] == JK_NOPOS ? 0 : pa[b]];
      wb[b] = l.bw[pb[b] == JK_NOPOS ? 0 : pb[b]];
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool ha = act[b] && pa[b] != JK_NOPOS && (NARROW ? (uint32_t)(wa[b] >> 32) == (uint32_t)key[b] : wa[b] == (uint64_t)key[b]);
      const bool hb = act[b] && pb[b] != JK_NOPOS && (NARROW ? (uint32_t)(wb[b] >> 32) == (uint32_t)key[b] : wb[b] == (uint64_t)key[b]);
      mine += (uint32_t)ha + (uint32_t)hb + (uint32_t)(KEEP && act[b] && !ha && !hb);
    }
  }
  mine = wave_reduce_add(mine);
  if (lane_id() == 0) l.wave_cnt[threadIdx.x / WAVE] = mine;
  block_sync();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < JK_PROBE_THREADS / WAVE; ++w) t += l.wave_cnt[w];
    a.counts[blockIdx.x] = t;
  }
}

// The plain INNER write pass with the BUILD relation's payload word in the LDS image (PayCarry::bmode): the build partition is
// staged as (tuple, payload) pairs and every pair writes the build payload next to its index pair, the probe payload (PP: the
// probe tuples carry one too) and the key -- a materialising join without a single gather.  NARROW tuples.  The image is
// 16 bytes per build tuple: one 1024-thread workgroup per CU (86 KB at C3's partition size) instead of two 512-thread ones;
// a unit's staging + cuckoo build is ~2 % of its time, so little is lost to the missing overlap.
constexpr int JK_BP_THREADS = 1024;
constexpr int JK_BP_BATCH22 = 2;       // row pairs per thread and batch when two probe words AND two build words travel (3 spills: 128 VGPRs + 28 B of scratch)
// (PP: payload words a probe tuple carries, 0 / 1 / 2; BW2: the build relation has a SECOND 8-byte payload column, bpay_mode 4 -- staged by
// build row next to the carried word, 24 bytes per build tuple in LDS)
static size_t probe_bp_lds_bytes(uint32_t cap, uint32_t H, bool bw2 = false) { return (size_t)cap * (bw2 ? 24 : 16) + (size_t)H * 8 + 16; }
template <bool POW2, int PP, bool BW2 = false>
__global__ __launch_bounds__(JK_BP_THREADS) void jk_probe_bp(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t H = a.nslots, cap = a.cap;
  uint64_t *bw = (uint64_t *)lds_raw;                               // [cap] key32 << 32 | build row
  uint64_t *bp = bw + cap;                                          // [cap] build payload word
  uint64_t *bp2 = bp + cap;                                         // [cap] BW2: the second build payload column's values
  uint32_t *T = (uint32_t *)(bp + (BW2 ? 2 : 1) * (size_t)cap);     // [2 H] cuckoo tables of positions
  unsigned int *lcur = (unsigned int *)(T + 2 * (size_t)H);         // pairs written so far by this unit
  unsigned int *failed = lcur + 1;
  const Unit u = a.units[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < u.build_count; i += JK_BP_THREADS) {
    const uint64_t w = a.build.w[u.build_begin + i];
    bw[i] = w;
    bp[i] = a.build.pay[u.build_begin + i];
    if constexpr (BW2) bp2[i] = ((const uint64_t *)a.bpay_src[1])[(uint32_t)w];       // by build row: ~3000 scattered reads per unit
  }
  for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_BP_THREADS) T[i] = JK_NOPOS;
  if (threadIdx.x == 0) { *lcur = 0; *failed = 0; }
  block_sync();
  const uint32_t kb_lo = (uint32_t)a.kbias, kb_hi = (uint32_t)(a.kbias >> 32);
  auto fold_of = [&](uint32_t key) -> uint32_t {
    const uint32_t lo = key + kb_lo;
    const uint32_t hi = kb_hi + (lo < key ? 1u : 0u);
    return lo ^ (hi * 0x9e3779b1u);
  };
  const int hshift = POW2 ? 32 - (__ffs((int)H) - 1) : 0;
  uint32_t seed = 0;
  for (int attempt = 0; attempt < 4; ++attempt) {
    for (uint32_t p0 = threadIdx.x; p0 < u.build_count; p0 += JK_BP_THREADS) {
      uint32_t cur = p0, table = 0;
      int moves = 0;
      for (; moves < JK_CUCKOO_MAX_MOVES; ++moves) {
        const uint32_t f = fold_of((uint32_t)(bw[cur] >> 32)) ^ seed;
        const uint32_t slot = table ? H + (POW2 ? (f * 0xc2b2ae35u) >> hshift : __umulhi(f * 0xc2b2ae35u, H))
                                    : (POW2 ? (f * 0x9e3779b1u) >> hshift : __umulhi(f * 0x9e3779b1u, H));
        const uint32_t old = atomicExch(&T[slot], cur);
        if (old == JK_NOPOS) break;
        cur = old;
        table ^= 1;
      }
      if (moves == JK_CUCKOO_MAX_MOVES) *failed = 1;
    }
    block_sync();
    if (!*failed || attempt == 3) break;
    block_sync();
    for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_BP_THREADS) T[i] = JK_NOPOS;
    if (threadIdx.x == 0) *failed = 0;
    seed += 0x9e3779b9u;
    block_sync();
  }
  if (*failed) {
    if (threadIdx.x == 0) a.unit_todo[atomicAdd(&a.opt_state[2], 1ull)] = blockIdx.x;
    return;
  }
  const unsigned long long unit_base = a.counts[blockIdx.x];
  const uint32_t unit_cap = a.optimistic ? u.probe_count : 0xffffffffu;
  int32_t *__restrict__ op = a.out_probe + unit_base;
  int32_t *__restrict__ ob = a.out_build + unit_base;
  // output columns: widths are workgroup-uniform run-time values (one kernel for every payload mode)
  uint64_t *__restrict__ po8 = (PP && (a.pay_mode == 1 || a.pay_mode == 4)) ? (uint64_t *)a.pay_out[0] + unit_base : nullptr;
  uint64_t *__restrict__ po8b = PP == 2 ? (uint64_t *)a.pay_out[1] + unit_base : nullptr;
  uint32_t *__restrict__ po4a = (PP && (a.pay_mode == 2 || a.pay_mode == 3)) ? (uint32_t *)a.pay_out[0] + unit_base : nullptr;
  uint32_t *__restrict__ po4b = (PP && a.pay_mode == 3) ? (uint32_t *)a.pay_out[1] + unit_base : nullptr;
  uint64_t *__restrict__ bo8 = (a.bpay_mode == 1 || a.bpay_mode == 4) ? (uint64_t *)a.bpay_out[0] + unit_base : nullptr;
  uint64_t *__restrict__ bo8b = BW2 ? (uint64_t *)a.bpay_out[1] + unit_base : nullptr;
  uint32_t *__restrict__ bo4a = (a.bpay_mode == 2 || a.bpay_mode == 3) ? (uint32_t *)a.bpay_out[0] + unit_base : nullptr;
  uint32_t *__restrict__ bo4b = a.bpay_mode == 3 ? (uint32_t *)a.bpay_out[1] + unit_base : nullptr;
  uint64_t *__restrict__ ko8 = a.key_width == 8 ? (uint64_t *)a.key_out + unit_base : nullptr;
  uint32_t *__restrict__ ko4 = a.key_width == 4 ? (uint32_t *)a.key_out + unit_base : nullptr;
  // (with a probe payload as well a batch of 8 tuples per lane needs more than the 128 registers a 1024-thread workgroup gets)
  constexpr int BATCH = (PP == 2 && BW2) ? JK_BP_BATCH22 : ((PP == 2 || BW2) ? 2 : (PP ? 3 : JK_PROBE_BATCH)), NB = BATCH * 2;
  const uint32_t lead = u.probe_begin & 1u;
  const uint64_t *__restrict__ src = a.probe.w + (u.probe_begin - lead);
  const uint64_t *__restrict__ src_pay = PP ? a.probe.pay + (size_t)(u.probe_begin - lead) * (PP == 2 ? 2 : 1) : nullptr;
  const uint32_t vtotal = lead + u.probe_count;
  const uint32_t last_pair = (vtotal - 1) & ~1u;
  for (uint32_t base = 0; base < vtotal; base += JK_BP_THREADS * NB) {
    uint32_t key[NB], prow[NB];
    uint64_t pay[PP ? NB : 1], pay2[PP == 2 ? NB : 1];
    bool act[NB];
#pragma unroll
    for (int b = 0; b < BATCH; ++b) {
      const uint32_t v = base + (b * JK_BP_THREADS + threadIdx.x) * 2;
      const uint32_t vc = v < last_pair ? v : last_pair;
      const ulonglong2 ww = *reinterpret_cast<const ulonglong2 *>(src + vc);
      if constexpr (PP == 2) {
        const ulonglong2 p0 = *reinterpret_cast<const ulonglong2 *>(src_pay + 2 * (size_t)vc);
        const ulonglong2 p1 = *reinterpret_cast<const ulonglong2 *>(src_pay + 2 * (size_t)vc + 2);
        pay[2 * b] = p0.x; pay2[2 * b] = p0.y; pay[2 * b + 1] = p1.x; pay2[2 * b + 1] = p1.y;
      } else if constexpr (PP != 0) {
        const ulonglong2 pp = *reinterpret_cast<const ulonglong2 *>(src_pay + vc);
        pay[2 * b] = pp.x; pay[2 * b + 1] = pp.y;
      }
      key[2 * b] = (uint32_t)(ww.x >> 32); prow[2 * b] = (uint32_t)ww.x;
      key[2 * b + 1] = (uint32_t)(ww.y >> 32); prow[2 * b + 1] = (uint32_t)ww.y;
      act[2 * b] = v >= lead && v < vtotal;
      act[2 * b + 1] = v + 1 < vtotal;
    }
    uint32_t pa[NB], pb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const uint32_t f = fold_of(key[b]) ^ seed;
      pa[b] = T[POW2 ? (f * 0x9e3779b1u) >> hshift : __umulhi(f * 0x9e3779b1u, H)];
      pb[b] = T[H + (POW2 ? (f * 0xc2b2ae35u) >> hshift : __umulhi(f * 0xc2b2ae35u, H))];
    }
    uint64_t wa[NB], wb[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      wa[b] = bw[pa[b] == JK_NOPOS ? 0 : pa[b]];
      wb[b] = bw[pb[b] == JK_NOPOS ? 0 : pb[b]];
    }
    // per tuple ONE candidate survives in registers -- the position and build row of its hit (table 0 first); the rare tuple
    // with a hit in both tables (a build key present twice) re-reads the second one from LDS when it is written
    uint32_t hamask = 0, hbmask = 0, hpos[NB], hrow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const bool ha = act[b] && pa[b] != JK_NOPOS && (uint32_t)(wa[b] >> 32) == key[b];
      const bool hb = act[b] && pb[b] != JK_NOPOS && (uint32_t)(wb[b] >> 32) == key[b];
      hamask |= (uint32_t)ha << b;
      hbmask |= (uint32_t)hb << b;
      hpos[b] = ha ? pa[b] : (hb ? pb[b] : 0u);
      hrow[b] = ha ? (uint32_t)wa[b] : (uint32_t)wb[b];
    }
    uint64_t q[NB], q2[BW2 ? NB : 1];      // the build payload of the hit: one more LDS read per tuple (two with BW2), requested together
#pragma unroll
    for (int b = 0; b < NB; ++b) { q[b] = bp[hpos[b]]; if constexpr (BW2) q2[b] = bp2[hpos[b]]; }
    auto emit_one = [&](int b, uint32_t pos, uint32_t brow, uint64_t bpay, uint64_t bpay2) {
      if (pos >= unit_cap) { a.opt_state[1] = 1; return; }   // would spill into the next unit's slots: the host redoes the join two-pass
      op[pos] = (int32_t)prow[b];
      ob[pos] = (int32_t)brow;
      if (bo8) bo8[pos] = bpay;
      if constexpr (BW2) bo8b[pos] = bpay2;
      if (bo4a) bo4a[pos] = (uint32_t)bpay;
      if (bo4b) bo4b[pos] = (uint32_t)(bpay >> 32);
      if (ko8) ko8[pos] = (uint64_t)key[b] + a.kbias;
      if (ko4) ko4[pos] = key[b];
      if constexpr (PP == 2) po8b[pos] = pay2[b];
      if constexpr (PP != 0) {
        if (po8) po8[pos] = pay[b];
        if (po4a) po4a[pos] = (uint32_t)pay[b];
        if (po4b) po4b[pos] = (uint32_t)(pay[b] >> 32);
      }
    };
    if (__all((hamask & hbmask) == 0)) {
      // nobody has two pairs for one tuple: ONE claim per wave and batch (see jk_probe_fast)
      const uint32_t onemask = hamask | hbmask;
      unsigned long long mm[NB];
      uint32_t off[NB], tot = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        mm[b] = __ballot((onemask >> b) & 1u);
        off[b] = tot;
        tot += (uint32_t)__popcll(mm[b]);
      }
      uint32_t wbase = 0;
      if (lane_id() == 0 && tot) wbase = atomicAdd(lcur, tot);
      wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if ((onemask >> b) & 1u) emit_one(b, wbase + off[b] + (uint32_t)mask_rank(mm[b]), hrow[b], q[b], BW2 ? q2[b] : 0);
    } else {
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint32_t c = ((hamask >> b) & 1u) + ((hbmask >> b) & 1u);
        const uint32_t incl = wave_scan_incl(c);
        const uint32_t wave_total = __shfl(incl, WAVE - 1, WAVE);
        uint32_t wbase = 0;
        if (lane_id() == 0 && wave_total) wbase = atomicAdd(lcur, wave_total);
        uint32_t pos = __shfl(wbase, 0, WAVE) + incl - c;
        if (c) emit_one(b, pos++, hrow[b], q[b], BW2 ? q2[b] : 0);
        if (c == 2) emit_one(b, pos, (uint32_t)bw[pb[b]], bp[pb[b]], BW2 ? bp2[pb[b]] : 0);      // the second copy of the key, from table 1
      }
    }
  }
  if (a.optimistic) {
    block_sync();
    if (threadIdx.x == 0) {
      atomicAdd(&a.opt_state[0], (unsigned long long)*lcur);
      if (a.unit_pairs) a.unit_pairs[blockIdx.x] = *lcur;
    }
  }
}

// REPEATED BUILD KEYS in the plain case (INNER / LEFT, NARROW tuples, exact keys): a cuckoo table of positions holds a key at most
// twice, so such joins (many-to-many: the sample pass reports units that did not settle) took the general kernel for both
// passes -- 2.9 + 4.8 ms of a 10.7 ms join on 2.5e8 x 1e8 rows with every build key four times.  This is the lean kernel for
// them: an open-addressing table over the DISTINCT keys of the partition (2 H slots of positions, linear probing), the copies of
// a key chained behind its head (next[]: low 16 bits = next position, high 16 bits of a HEAD = the length of its chain, filled
// in once per partition after the build), the eight first probes of a batch in flight together, COUNT = one read of the head's
// length, WRITE = one output claim per wave and batch and a walk of exactly the key's copies.  Any multiplicity that fits LDS.
constexpr uint32_t JK_MM_END = 0xffffu;
template <bool WRITE, bool KEEP, bool P6>
__global__ __launch_bounds__(JK_PROBE_THREADS) void jk_probe_multi(ProbeArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const uint32_t H = a.nslots, cap = a.cap;
  const ProbeLds l = carve_probe_lds<true>(lds_raw, cap, H);
  unsigned int *lcur = (unsigned int *)l.unit_cursor;
  const Unit u = a.units[blockIdx.x];
  const uint32_t mask = 2 * H - 1;
  for (uint32_t i = threadIdx.x; i < u.build_count; i += JK_PROBE_THREADS) {
    uint64_t w = a.build.w[u.build_begin + i];
    if (P6) w = ((uint64_t)p6_remainder((uint32_t)(w >> 32), a.p6_kbias, a.p6_fb, a.p6_world) << 32) | (uint32_t)w;
    l.bw[i] = w;
    l.next[i] = JK_MM_END;
  }
  for (uint32_t i = threadIdx.x; i < 2 * H; i += JK_PROBE_THREADS) l.T[i] = JK_NOPOS;
  if (threadIdx.x == 0) *lcur = 0;
  block_sync();
  const uint32_t kb_lo = (uint32_t)a.kbias, kb_hi = (uint32_t)(a.kbias >> 32);
  auto slot_of_key = [&](uint32_t key) -> uint32_t {        // fold of the raw key, one multiply (the partition id took lowbias32's bits)
    const uint32_t lo = key + kb_lo;
    const uint32_t hi = kb_hi + (lo < key ? 1u : 0u);
    return (((lo ^ (hi * 0x9e3779b1u)) * 0xc2b2ae35u) >> 7) & mask;
  };
  // build: a position becomes the head of its key (CAS into an empty slot) or goes right behind the head
  for (uint32_t p = threadIdx.x; p < u.build_count; p += JK_PROBE_THREADS) {
    const uint32_t kp = (uint32_t)(l.bw[p] >> 32);
    uint32_t slot = slot_of_key(kp);
    for (;;) {
      const uint32_t q = atomicCAS(&l.T[slot], JK_NOPOS, p);
      if (q == JK_NOPOS) break;
      if ((uint32_t)(l.bw[q] >> 32) == kp) { l.next[p] = atomicExch(&l.next[q], p); break; }
      slot = (slot + 1) & mask;
    }
  }
  block_sync();
  // chain lengths, once per key: the thread that finds a head in its table slots walks the chain
  for (uint32_t sl = threadIdx.x; sl < 2 * H; sl += JK_PROBE_THREADS) {
    const uint32_t h = l.T[sl];
    if (h == JK_NOPOS) continue;
    uint32_t len = 1;
    for (uint32_t p = l.next[h] & 0xffffu; p != JK_MM_END; p = l.next[p] & 0xffffu) ++len;
    l.next[h] = (l.next[h] & 0xffffu) | (len << 16);
  }
  block_sync();
  const unsigned long long unit_base = WRITE ? a.counts[blockIdx.x] : 0ull;
  const uint32_t unit_cap = (WRITE && a.optimistic) ? u.probe_count : 0xffffffffu;
  int32_t *__restrict__ op = WRITE ? a.out_probe + unit_base : nullptr;
  int32_t *__restrict__ ob = WRITE ? a.out_build + unit_base : nullptr;
  constexpr int NB = JK_PROBE_BATCH * 2;
  const uint32_t lead = u.probe_begin & 1u;
  const uint64_t *__restrict__ src = a.probe.w + (u.probe_begin - lead);
  const uint32_t vtotal = lead + u.probe_count;
  const uint32_t last_pair = (vtotal - 1) & ~1u;
  unsigned long long mine = 0;
  for (uint32_t base = 0; base < vtotal; base += JK_PROBE_THREADS * NB) {
    uint32_t key[NB], prow[NB];
    bool act[NB];
#pragma unroll
    for (int b = 0; b < JK_PROBE_BATCH; ++b) {
      const uint32_t v = base + (b * JK_PROBE_THREADS + threadIdx.x) * 2;
      const uint32_t vc = v < last_pair ? v : last_pair;
      if constexpr (P6) {
        p6_load_pair(a.probe.w, (u.probe_begin - lead) + vc, key[2 * b], prow[2 * b], key[2 * b + 1], prow[2 * b + 1]);
      } else {
        const ulonglong2 ww = *reinterpret_cast<const ulonglong2 *>(src + vc);
        key[2 * b] = (uint32_t)(ww.x >> 32); prow[2 * b] = (uint32_t)ww.x;
        key[2 * b + 1] = (uint32_t)(ww.y >> 32); prow[2 * b + 1] = (uint32_t)ww.y;
      }
      act[2 * b] = v >= lead && v < vtotal;
      act[2 * b + 1] = v + 1 < vtotal;
    }
    // first probes of the batch together; the few that land on another key's head walk on one by one
    uint32_t slot[NB], head[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) { slot[b] = slot_of_key(key[b]); head[b] = l.T[slot[b]]; }
    uint64_t hw[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) hw[b] = l.bw[head[b] == JK_NOPOS ? 0 : head[b]];
    uint32_t cnt[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      while (head[b] != JK_NOPOS && (uint32_t)(hw[b] >> 32) != key[b]) {
        slot[b] = (slot[b] + 1) & mask;
        head[b] = l.T[slot[b]];
        hw[b] = l.bw[head[b] == JK_NOPOS ? 0 : head[b]];
      }
      const bool hit = act[b] && head[b] != JK_NOPOS;
      cnt[b] = hit ? l.next[head[b]] >> 16 : ((KEEP && act[b]) ? 1u : 0u);
      if (!hit) head[b] = JK_NOPOS;
    }
    if constexpr (!WRITE) {
#pragma unroll
      for (int b = 0; b < NB; ++b) mine += cnt[b];
    } else {
      // ONE claim per wave and batch: the lanes' pair counts of tuple b are scanned, the eight totals added up, lane 0 claims
      uint32_t off[NB], tot = 0;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const uint32_t incl = wave_scan_incl(cnt[b]);
        off[b] = tot + incl - cnt[b];
        tot += (uint32_t)__shfl((int)incl, WAVE - 1, WAVE);
      }
      uint32_t wbase = 0;
      if (lane_id() == 0 && tot) wbase = atomicAdd(lcur, tot);
      wbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)wbase);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        if (!cnt[b]) continue;
        uint32_t pos = wbase + off[b];
        if (pos + cnt[b] > unit_cap) { a.opt_state[1] = 1; continue; }      // would spill into the next unit's slots
        if (head[b] == JK_NOPOS) { op[pos] = (int32_t)prow[b]; ob[pos] = JK_EMPTY; continue; }      // LEFT: no partner
        op[pos] = (int32_t)prow[b];
        ob[pos] = (int32_t)(uint32_t)hw[b];
        for (uint32_t p = l.next[head[b]] & 0xffffu; p != JK_MM_END; p = l.next[p] & 0xffffu) {
          ++pos;
          op[pos] = (int32_t)prow[b];
          ob[pos] = (int32_t)(uint32_t)l.bw[p];
        }
      }
    }
  }
  if constexpr (!WRITE) {
    mine = wave_reduce_add(mine);
    if (lane_id() == 0) l.wave_cnt[threadIdx.x / WAVE] = mine;
    block_sync();
    if (threadIdx.x == 0) {
      unsigned long long t = 0;
      for (int w = 0; w < JK_PROBE_THREADS / WAVE; ++w) t += l.wave_cnt[w];
      a.counts[blockIdx.x] = t;
    }
  } else if (a.optimistic) {
    block_sync();
    if (threadIdx.x == 0) {
      atomicAdd(&a.opt_state[0], (unsigned long long)*lcur);
      if (a.unit_pairs) a.unit_pairs[blockIdx.x] = *lcur;
    }
  }
}

// Sparse optimistic pass: every unit wrote its pairs at the START of its own slot range (slot_off[u], room for one pair per
// probe tuple); this moves them to their final, dense places (pair_off[u] = exclusive scan of the units' pair counts).
// One workgroup per unit, 16 B per pair -- cheaper than a count pass (which reads every probe tuple and rebuilds every LDS
// table) as long as fewer than about half of the probe rows find a partner.
__global__ __launch_bounds__(256) void jk_widen_counts(const uint32_t *__restrict__ in, uint64_t *__restrict__ out, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i <= n; i += gridDim.x * 256) out[i] = i < n ? in[i] : 0;      // n + 1 entries
}
// (a carried payload column moves with its pairs: W = its element width, 0 = none)
template <class T>
__device__ __forceinline__ void compact_column(const void *src, void *dst, uint64_t from, uint64_t to, uint32_t n) {
  const T *s = (const T *)src + from;
  T *d = (T *)dst + to;
  for (uint32_t i = threadIdx.x; i < n; i += 256 * 4) {
    T v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = i + k * 256 < n ? s[i + k * 256] : T(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (i + k * 256 < n) d[i + k * 256] = v[k];
  }
}
struct PayMove { int ncols; int width[5]; const void *src[5]; void *dst[5]; };      // carried columns (probe payload, key, build payload): 8- or 4-byte elements
__global__ __launch_bounds__(256) void jk_compact_units(const uint64_t *__restrict__ slot_off, const uint64_t *__restrict__ pair_off,
                                                        const int32_t *__restrict__ sp, const int32_t *__restrict__ sb,
                                                        int32_t *__restrict__ dp, int32_t *__restrict__ db, PayMove pm) {
  const uint64_t from = slot_off[blockIdx.x], to = pair_off[blockIdx.x];
  const uint32_t n = (uint32_t)(pair_off[blockIdx.x + 1] - to);
  for (int c = 0; c < pm.ncols; ++c) {
    if (pm.width[c] == 8) compact_column<uint64_t>(pm.src[c], pm.dst[c], from, to, n);
    else compact_column<uint32_t>(pm.src[c], pm.dst[c], from, to, n);
  }
  for (uint32_t i = threadIdx.x; i < n; i += 256 * 4) {
    int32_t vp[4], vb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t j = i + k * 256;
      vp[k] = j < n ? sp[from + j] : 0;
      vb[k] = j < n ? sb[from + j] : 0;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t j = i + k * 256;
      if (j < n) { dp[to + j] = vp[k]; db[to + j] = vb[k]; }
    }
  }
}

// The same single pass for joins where MOST probe rows hit (50 % .. 100 %): closing every unit's hole by moving all pairs is
// 16 bytes per pair -- more than the count pass it would replace.  Instead only the pairs that sit BEHIND the final size N
// move, into the holes in front of it: with hit rate h that is h (1 - h) of the slots (80 % hits: 0.16 pairs per slot instead
// of 0.8), the pair order of a join being unspecified.  The columns keep their cap_pairs-slot allocation; their size is N.
//   jk_hole_counts: per unit, its free slots below N and its pairs at or above N (the two sums are equal);
//   jk_fill_holes:  after exclusive scans of both, tail pair number g goes to hole slot number g.
__global__ __launch_bounds__(256) void jk_hole_counts(const Unit *__restrict__ units, const uint64_t *__restrict__ slot_off,
                                                      const uint32_t *__restrict__ unit_pairs, uint32_t nunits, uint64_t N,
                                                      uint64_t *__restrict__ holes, uint64_t *__restrict__ tails) {
  const uint32_t u = blockIdx.x * 256 + threadIdx.x;
  if (u > nunits) return;
  uint64_t h = 0, t = 0;
  if (u < nunits) {
    const uint64_t o = slot_off[u], used_end = o + unit_pairs[u], slot_end = o + units[u].probe_count;
    if (used_end < N) h = (slot_end < N ? slot_end : N) - used_end;
    if (used_end > N) t = used_end - (o > N ? o : N);
  }
  holes[u] = h;
  tails[u] = t;
}
__global__ __launch_bounds__(256) void jk_fill_holes(const uint64_t *__restrict__ slot_off, const uint32_t *__restrict__ unit_pairs, uint32_t nunits,
                                                     uint64_t N, const uint64_t *__restrict__ hole_pre, const uint64_t *__restrict__ tail_pre,
                                                     int32_t *op, int32_t *ob, PayMove pm) {
  __shared__ uint32_t range[2];
  const uint32_t u = blockIdx.x;
  const uint64_t g0 = tail_pre[u];
  const uint32_t cnt = (uint32_t)(tail_pre[u + 1] - g0);
  if (!cnt) return;
  const uint64_t o = slot_off[u], src0 = o > N ? o : N;
  // last v with hole_pre[v] <= g (upper bound - 1): the unit whose hole holds slot number g
  auto unit_of = [&](uint64_t g, uint32_t lo, uint32_t hi) -> uint32_t {        // searches [lo, hi]
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo + 1) / 2;
      if (hole_pre[mid] <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
  };
  if (threadIdx.x < 2) range[threadIdx.x] = unit_of(threadIdx.x ? g0 + cnt - 1 : g0, 0, nunits - 1);
  __syncthreads();
  const uint32_t v0 = range[0], v1 = range[1];
  // the tail is a contiguous range of pairs and every hole a contiguous range of slots: one plain copy per (tail, hole) overlap --
  // a handful per unit -- instead of a search per pair (1.04 -> 0.6 ms at 80 % hits: the per-pair version was six dependent loads deep)
  for (uint32_t v = v0; v <= v1; ++v) {                 // workgroup-uniform
    const uint64_t hb = hole_pre[v], he = hole_pre[v + 1];
    const uint64_t ga = g0 > hb ? g0 : hb, gb = g0 + cnt < he ? g0 + cnt : he;
    if (ga >= gb) continue;
    const uint64_t to = slot_off[v] + unit_pairs[v] + (ga - hb), from = src0 + (ga - g0);
    const uint32_t n = (uint32_t)(gb - ga);
    compact_column<int32_t>(op, op, from, to, n);
    compact_column<int32_t>(ob, ob, from, to, n);
    for (int c = 0; c < pm.ncols; ++c) {
      if (pm.width[c] == 8) compact_column<uint64_t>(pm.src[c], pm.dst[c], from, to, n);
      else compact_column<uint32_t>(pm.src[c], pm.dst[c], from, to, n);
    }
  }
}

// ---------------------------------------------------------------------------
// global-table path for partitions whose build side exceeds the LDS image (heavy key
// duplication / build sides beyond 32768 * JK_MAX_BUILD rows).  Same semantics, table
// of (key, row) slots in HBM, one thread per tuple.
// ---------------------------------------------------------------------------
// The table holds DISTINCT keys; the build tuples of a key hang behind its slot as a chain of partition positions
// (thead[slot] -> next[pos] -> ...).  Round 1 inserted every tuple with linear probing: k copies of one key cost k^2 / 2
// probes to insert and k slots to walk for every probe row that hashes nearby -- a build side with 6e6 copies of one key
// (tools/stress_join.py --seed 41 --case 174) did not finish in five minutes.
constexpr unsigned long long GJ_NOKEY = ~0ull;        // never a NARROW key (< 2^32); a WIDE key with these bits uses slot S
__global__ void gj_fill(unsigned long long *tkey, int32_t *thead, uint32_t nslots) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i <= nslots; i += gridDim.x * blockDim.x) { tkey[i] = GJ_NOKEY; thead[i] = -1; }
}
template <bool NARROW>
__global__ void gj_build(Tuples b, uint32_t begin, uint32_t n, unsigned long long *tkey, int32_t *thead, int32_t *next, uint32_t S) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const unsigned long long k = tup_key<NARROW>(b.w[begin + i]);
    uint32_t slot = S;                              // the reserved slot of the key that looks like "no key"
    if (NARROW || k != GJ_NOKEY) {
      slot = slot_of(k, S);
      for (;;) {
        const unsigned long long old = atomicCAS(&tkey[slot], GJ_NOKEY, k);
        if (old == GJ_NOKEY || old == k) break;
        slot = (slot + 1 == S) ? 0 : slot + 1;
      }
    }
    next[i] = atomicExch(&thead[slot], (int32_t)i);
  }
}
template <bool WRITE, bool NARROW>
__global__ void gj_probe(ProbeArgs a, KeyTable probe_t, KeyTable build_t, const unsigned long long *tkey, const int32_t *thead,
                         const int32_t *next, uint32_t build_begin, uint32_t probe_begin, uint32_t probe_count,
                         unsigned long long *cursor) {
  const uint32_t S = a.nslots;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t rounds = (probe_count + stride - 1) / stride;
  auto build_row_at = [&](int32_t p) -> int32_t {
    return NARROW ? (int32_t)(uint32_t)a.build.w[build_begin + p] : a.build.idx[build_begin + p];
  };
  unsigned long long my_count = 0;
  for (uint32_t rnd = 0; rnd < rounds; ++rnd) {
    const uint32_t i = rnd * stride + blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t cnt = 0;
    int32_t prow = 0, head = -1;
    if (i < probe_count) {
      const uint64_t w = a.probe.w[probe_begin + i];
      const unsigned long long k = tup_key<NARROW>(w);
      prow = NARROW ? (int32_t)(uint32_t)w : a.probe.idx[probe_begin + i];
      if (!NARROW && k == GJ_NOKEY) head = thead[S];
      else {
        uint32_t slot = slot_of(k, S);
        for (;;) {
          const unsigned long long t = tkey[slot];
          if (t == k) { head = thead[slot]; break; }
          if (t == GJ_NOKEY) break;
          slot = (slot + 1 == S) ? 0 : slot + 1;
        }
      }
      for (int32_t p = head; p >= 0; p = next[p]) {
        const int32_t r = build_row_at(p);
        if (!a.verify || rows_equal(probe_t, prow, build_t, r)) {
          ++cnt;
          if (a.build_matched) a.build_matched[r] = 1;
        }
      }
    }
    const bool pad = (i < probe_count) && cnt == 0 && a.keep_unmatched_probe;
    if (!WRITE) {
      my_count += pad ? 1 : cnt;
    } else {
      const uint32_t emit = pad ? 1 : cnt;
      const uint32_t incl = wave_scan_incl(emit);
      const uint32_t wave_total = __shfl(incl, WAVE - 1, WAVE);
      unsigned long long base = 0;
      if (lane_id() == 0 && wave_total) base = atomicAdd(cursor, (unsigned long long)wave_total);
      base = __shfl(base, 0, WAVE);
      unsigned long long pos = base + incl - emit;
      if (pad) {
        a.out_probe[pos] = prow;
        a.out_build[pos] = JK_EMPTY;
        pay_gather(a, pos, prow);
      } else if (cnt) {
        for (int32_t p = head; p >= 0; p = next[p]) {
          const int32_t r = build_row_at(p);
          if (!a.verify || rows_equal(probe_t, prow, build_t, r)) {
            a.out_probe[pos] = prow;
            a.out_build[pos] = r;
            pay_gather(a, pos, prow);
            bpay_gather(a, pos, r);
            ++pos;
          }
        }
      }
    }
  }
  if (!WRITE) {
    my_count = wave_reduce_add(my_count);
    if (lane_id() == 0 && my_count) atomicAdd(cursor, my_count);
  }
}

// ---------------------------------------------------------------------------
// tails: rows that never entered the partitioned path
// ---------------------------------------------------------------------------
// rows of `t` that cannot match (null / NaN key, outside the build range), compacted
// behind *cursor as (row, -1) pairs (LEFT / FULL probe side)
// Output positions for the flagged items of a 256-thread tile (JK_TAIL_ITEMS rows per thread): ballots rank the flags
// inside every wave, the four wave totals meet in LDS and ONE atomicAdd per tile claims the range.  One atomic per
// wave and row round on the single cursor made the FULL join's tail 4 ms at 2e7 rows (312 k same-address atomics).
constexpr int JK_TAIL_ITEMS = 8;
__device__ __forceinline__ void tile_claim(const bool (&flag)[JK_TAIL_ITEMS], unsigned long long *cursor,
                                           unsigned long long (&pos)[JK_TAIL_ITEMS]) {
  __shared__ uint32_t wave_total[256 / WAVE];
  __shared__ unsigned long long tile_base;
  uint32_t running = 0, rank[JK_TAIL_ITEMS];
#pragma unroll
  for (int k = 0; k < JK_TAIL_ITEMS; ++k) {
    const unsigned long long m = __ballot(flag[k]);
    rank[k] = running + (uint32_t)mask_rank(m);
    running += (uint32_t)__popcll(m);
  }
  const int wave = threadIdx.x / WAVE;
  if (lane_id() == 0) wave_total[wave] = running;
  block_sync();
  if (threadIdx.x == 0) {
    uint32_t total = 0;
    for (int w = 0; w < 256 / WAVE; ++w) total += wave_total[w];
    tile_base = total ? atomicAdd(cursor, (unsigned long long)total) : 0ULL;
  }
  block_sync();
  uint32_t before = 0;
  for (int w = 0; w < wave; ++w) before += wave_total[w];
#pragma unroll
  for (int k = 0; k < JK_TAIL_ITEMS; ++k) pos[k] = tile_base + before + rank[k];
  block_sync();                           // the LDS words are reused by the next tile
}

// pm (LEFT joins that carry a payload): the tail rows' payload values, read by row -- they leave in row order
__global__ __launch_bounds__(256) void jk_emit_unjoinable(KeyTable t, KeyPlan plan, int32_t *out_row, int32_t *out_none,
                                                          unsigned long long *cursor, PayMove pm) {
  constexpr int64_t TILE = 256 * JK_TAIL_ITEMS;
  for (int64_t tile = (int64_t)blockIdx.x * TILE; tile < t.nrows; tile += (int64_t)gridDim.x * TILE) {
    bool emit[JK_TAIL_ITEMS];
    unsigned long long pos[JK_TAIL_ITEMS];
#pragma unroll
    for (int k = 0; k < JK_TAIL_ITEMS; ++k) {
      const int64_t i = tile + k * 256 + threadIdx.x;
      uint64_t key;
      emit[k] = i < t.nrows && !make_key(t, plan, i, key);
    }
    tile_claim(emit, cursor, pos);
#pragma unroll
    for (int k = 0; k < JK_TAIL_ITEMS; ++k)
      if (emit[k]) {
        const int64_t row = tile + k * 256 + threadIdx.x;
        out_row[pos[k]] = (int32_t)row;
        out_none[pos[k]] = JK_EMPTY;
        for (int c = 0; c < pm.ncols; ++c) {
          if (pm.width[c] == 8) ((uint64_t *)pm.dst[c])[pos[k]] = ((const uint64_t *)pm.src[c])[row];
          else ((uint32_t *)pm.dst[c])[pos[k]] = ((const uint32_t *)pm.src[c])[row];
        }
      }
  }
}
// FULL join: build rows no probe row matched -> (-1, row)
__global__ __launch_bounds__(256) void jk_emit_unmatched_build(const uint8_t *matched, int64_t nrows, int32_t *out_none,
                                                               int32_t *out_row, unsigned long long *cursor) {
  constexpr int64_t TILE = 256 * JK_TAIL_ITEMS;
  for (int64_t tile = (int64_t)blockIdx.x * TILE; tile < nrows; tile += (int64_t)gridDim.x * TILE) {
    bool emit[JK_TAIL_ITEMS];
    unsigned long long pos[JK_TAIL_ITEMS];
#pragma unroll
    for (int k = 0; k < JK_TAIL_ITEMS; ++k) {
      const int64_t i = tile + k * 256 + threadIdx.x;
      emit[k] = i < nrows && !matched[i < nrows ? i : nrows - 1];
    }
    tile_claim(emit, cursor, pos);
#pragma unroll
    for (int k = 0; k < JK_TAIL_ITEMS; ++k)
      if (emit[k]) {
        out_none[pos[k]] = JK_EMPTY;
        out_row[pos[k]] = (int32_t)(tile + k * 256 + threadIdx.x);
      }
  }
}
__global__ __launch_bounds__(256) void jk_count_unmatched(const uint8_t *matched, int64_t nrows, unsigned long long *count) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrows; i += (int64_t)gridDim.x * blockDim.x)
    c += matched[i] ? 0 : 1;
  c = wave_reduce_add(c);
  if (lane_id() == 0 && c) atomicAdd(count, c);
}
__global__ void jk_fill_pairs(int32_t *a, int32_t av, int32_t *b, int32_t bv, int64_t n, int iota_a, int iota_b) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    a[i] = iota_a ? (int32_t)i : av;
    b[i] = iota_b ? (int32_t)i : bv;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
enum JoinKind { JOIN_INNER, JOIN_LEFT, JOIN_FULL };
// internal only, never returned through the ABI: probe_prepared asks for a build side without the third level
constexpr gdf_error GDF_AMD_RETRY_WITHOUT_LEVEL3 = (gdf_error)31;     // above every code of gdf_error
constexpr gdf_error GDF_AMD_RETRY_EXACT_PROBE = (gdf_error)30;        // probe_partitioned: the deferred probe side overflowed

// A join that materialises result_cols may CARRY the probe relation's non-key column(s) through the partition passes as one
// 64-bit payload word per row (Tuples::pay) instead of gathering them afterwards -- set up by join_entry, honoured by
// probe_prepared when the join runs on NARROW tuples with exact keys from one FAST, unmasked key column (INNER / LEFT).
struct PayCarry {
  int mode = 0;                        // 1: one 8-byte column; 2: one 4-byte column; 3: two 4-byte columns; 4: TWO 8-byte columns (round 6)
  const void *src[2] = {nullptr, nullptr};
  void *dst[2] = {nullptr, nullptr};   // OUT: rmm allocations of *out_n elements each, the caller's to free -- set iff carried
  bool carried = false;                // OUT
  // INNER joins on ONE integer key column: the result's key column comes out of the probe kernel as well (tuple key + kmin)
  int key_width = 0;                   // 8 / 4; 0: the caller gathers the key column
  const void *key_src = nullptr;       // the probe relation's key column
  void *key_dst = nullptr;             // OUT, like dst[]
  // INNER joins: the BUILD relation's non-key column(s) as well (same modes): the word travels through the build side's
  // partition passes, sits next to the build tuple in LDS and is written per pair (jk_probe_bp) -- instead of a gather that,
  // L2-friendly or not, pays one request per value (8.9 ms per column and 1e9 pairs)
  int bmode = 0;
  const void *bsrc[2] = {nullptr, nullptr};
  void *bdst[2] = {nullptr, nullptr};  // OUT, set iff build_carried
  bool build_carried = false;          // OUT
  int belem_bytes() const { return (bmode == 1 || bmode == 4) ? 8 : 4; }
  int bncols() const { return (bmode == 3 || bmode == 4) ? 2 : (bmode ? 1 : 0); }
  int elem_bytes(int c) const { return (mode == 1 || mode == 4) ? 8 : 4; }
  int ncols() const { return (mode == 3 || mode == 4) ? 2 : (mode ? 1 : 0); }
};

struct SideBufs {            // partitioned tuples of one relation
  DevBuf w[2], idx[2], pay[2];
  int final_buf = 0;         // which ping-pong buffer holds the fine-partitioned tuples
  std::vector<uint32_t> fine_off;   // [nfine+1] on the host; EXACT layout only (partitions are contiguous)
  std::vector<uint32_t> fine_begin, fine_cnt;   // [nfine] first tuple / tuple count of every fine partition, both layouts
  bool speculative = false;
  uint32_t joinable = 0;     // tuples that entered the partitioned path
  // DEFERRED speculative layout (probe side of the main path): nothing was read back -- the fill counters, their overflow
  // flags and the capacity stay on the device and probe_partitioned builds its units there (jk_make_units); the host
  // vectors above are empty
  bool deferred = false;
  bool p6 = false;           // the fine-partitioned tuples of this (deferred probe) side are six-byte ones (p6_store)
  int pay_words = 1;         // 64-bit words per payload element in pay[] (2: PayCarry mode 4)
  uint32_t cap2 = 0;
  DevBuf d_level1;           // [nseg + 1] level-1 fill counters + overflow flag; behind them (8-byte aligned) the words of `zero`
  unsigned long long *zero = nullptr;     // 16 zeroed 8-byte words inside d_level1, cleared with the counters in front of level 1: the state
                                          // blocks of probe_partitioned (units + sample 8 | tails 4 | write pass 4) need no memset of their own
  DevBuf d_caps;             // per-partition room of a skewed probe side (SkewCaps): rstart | rcap | fstart | fcap, else empty
  const uint32_t *d_fstart = nullptr, *d_fcap = nullptr;
  uint32_t nseg = 0;
  DevBuf d_cursor;           // [nfine + 1] level-2 fill cursors (f * cap2 + fill) + overflow flag
  DevBuf d_map;              // the level-2 segment map jk_scatter2 may still be reading
  // device copies of fine_begin / fine_cnt (build side: uploaded once by prepare_build for jk_make_units)
  DevBuf d_begin, d_cnt;
  // a BUILD side of the main path leaves partition_side with its last launches still queued (round 5: the probe side's first launches
  // are queued behind them without a host round trip in between): the scratch those launches read or write -- the level-1 buffer, the
  // histograms' scans, the level-2 map -- waits here and is freed by settle(), behind a synchronisation
  std::vector<void *> hold;
  void keep(DevBuf &b) { if (b.p) hold.push_back(b.release()); }
  void settle() {
    if (hold.empty()) return;
    (void)hipStreamSynchronize(stream0());
    for (void *q : hold) rmmFree(q, (cudaStream_t)0);
    hold.clear();
  }
  ~SideBufs() { settle(); }
  Tuples tuples(int b) const { return Tuples{w[b].as<uint64_t>(), idx[b].as<int32_t>(), pay[b].as<uint64_t>()}; }
  Tuples final() const { return tuples(final_buf); }
};

static PartGeom choose_geometry(int64_t build_rows) {
  PartGeom g{};
  g.dbg = (int)lab::knob_int("GDF_JK_SDBG", 0);
  int fb = 0;
  while (fb < JK_MAX_FB && (build_rows >> fb) > JK_TARGET_BUILD) ++fb;
  // never fewer than 32 partitions: with one or a few, every tuple of a tile ranks on the same LDS counter and the
  // probe side has to take the histogram pass (5e8 x 3000 rows: 8.0 ms with one partition, see the notes in profiles/)
  if (fb < 5 && !lab::knob_on("GDF_JK_ALLOW_FEW_PARTS")) fb = 5;
  // (test hook: the geometry of a large build relation on a small one -- the six-byte tuples need 2^15 partitions)
  if (const long long forced = lab::path_int("GDF_JK_FORCE_FB", 0)) { if (forced >= 5 && forced <= JK_MAX_FB) fb = (int)forced; }
  g.fb = fb;
  g.b1 = fb <= 8 ? fb : (fb + 1) / 2;
  if (lab::knob_on("GDF_JK_B1") && fb > 8) { const int b1 = (int)lab::knob_int("GDF_JK_B1", 0); if (b1 >= fb - 8 && b1 <= 8) g.b1 = b1; }   // experiment switch
  g.b2 = fb - g.b1;
  // more rows than 2^15 LDS-sized partitions hold: a third level (at most 256-way), decided here, applied by refine_side
  g.b3 = 0;
  // (only when the 2^15 partitions would not fit LDS with some room for their spread: C4's 1.25e8-row shards stay two-level)
  if (fb == JK_MAX_FB && !lab::path_on("GDF_JK_NO_LEVEL3"))
    while (g.b3 < 8 && (build_rows >> (fb + g.b3)) > JK_MAX_BUILD - JK_MAX_BUILD / 5) ++g.b3;
  return g;
}

static inline int small_grid(int64_t n) { return stream_grid((size_t)(n > 0 ? n : 1), 256 * 8); }

template <int FAST, bool NARROW, int THREADS, bool MASKED>
static gdf_error launch_scatter1_t(const KeyTable &t, const KeyPlan &plan, const PartGeom &g, const uint32_t *H1off, Tuples out) {
  const size_t lds = sizeof(TileLds<NARROW, THREADS, false, sc1_items(NARROW, THREADS)>);
  HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter1<FAST, NARROW, THREADS, MASKED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  GDF_LAUNCH("jk_scatter1", (jk_scatter1<FAST, NARROW, THREADS, MASKED>), dim3(g.nchunks), dim3(THREADS), lds, stream0(), t, plan, g, H1off, out);
  HIP_CHECK_LAST();
  return GDF_SUCCESS;
}
template <int FAST, bool NARROW, bool MASKED>
static gdf_error launch_scatter1_n(int threads, const KeyTable &t, const KeyPlan &plan, const PartGeom &g, const uint32_t *H1off, Tuples out) {
  if (threads == 1024) { if constexpr (NARROW || FAST == 8) return launch_scatter1_t<FAST, NARROW, 1024, MASKED>(t, plan, g, H1off, out); }
  if (threads >= 512) return launch_scatter1_t<FAST, NARROW, 512, MASKED>(t, plan, g, H1off, out);
  if constexpr (MASKED) return launch_scatter1_t<FAST, NARROW, 512, MASKED>(t, plan, g, H1off, out);      // (masked: the two production tile sizes only)
  else return launch_scatter1_t<FAST, NARROW, 256, MASKED>(t, plan, g, H1off, out);
}
static gdf_error launch_scatter1(int fast, bool narrow, int threads, const KeyTable &t, const KeyPlan &plan, const PartGeom &g,
                                 const uint32_t *H1off, Tuples out, bool l6 = false) {
  const bool masked = fast != 0 && t.col[0].valid != nullptr;
  if (l6) {                          // six-byte level-1 tuples (L6): NARROW, a FAST key column, the 1024-thread tile, speculative layout
    // (TWO workgroups of half-size tiles per CU -- what a software-pipelined workgroup, VERDICT r4 item 1, comes down to in 135 KB of
    // LDS -- were built as a LAB variant in round 5 and lost: jk_scatter1 2.97 - 2.98 against 2.87 - 2.91 ms for C3's probe side in
    // alternating processes, profiles/r5_c_half_tile_workgroups_ab.jsonl; the variant is gone again)
    if (!(fast && threads == 1024 && g.cap1 && g.xs == 6 && g.b1 == 8) || (!narrow && fast != 8)) return GDF_INVALID_API_CALL;
    if (!narrow) {                   // WIDE keys: the ten-byte tuples (W10, p10_key) -- 12 tuples per thread, 12 bytes per tuple in LDS
      const size_t lds = sizeof(TileLds<false, 1024, false, 12>);
      if (masked) {
        HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter1<8, false, 1024, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GDF_LAUNCH("jk_scatter1_w10", (jk_scatter1<8, false, 1024, true, true>), dim3(g.nchunks), dim3(1024), lds, stream0(), t, plan, g, H1off, out);
      } else {
        HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter1<8, false, 1024, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GDF_LAUNCH("jk_scatter1_w10", (jk_scatter1<8, false, 1024, false, true>), dim3(g.nchunks), dim3(1024), lds, stream0(), t, plan, g, H1off, out);
      }
      HIP_CHECK_LAST();
      return GDF_SUCCESS;
    }
#define JK_SC1_L6(F, M)                                                                                                             \
    do {                                                                                                                            \
      const size_t lds = sizeof(TileLds<true, 1024>);                                                                               \
      HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter1<F, true, 1024, M, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      GDF_LAUNCH("jk_scatter1", (jk_scatter1<F, true, 1024, M, true>), dim3(g.nchunks), dim3(1024), lds, stream0(), t, plan, g, H1off, out); \
    } while (0)
    if (fast == 8 && masked) JK_SC1_L6(8, true);
    else if (fast == 8) JK_SC1_L6(8, false);
    else if (masked) JK_SC1_L6(4, true);
    else JK_SC1_L6(4, false);
#undef JK_SC1_L6
    HIP_CHECK_LAST();
    return GDF_SUCCESS;
  }
  if (masked) {
    if (fast == 4) return launch_scatter1_n<4, true, true>(threads, t, plan, g, H1off, out);
    return narrow ? launch_scatter1_n<8, true, true>(threads, t, plan, g, H1off, out)
                  : launch_scatter1_n<8, false, true>(threads, t, plan, g, H1off, out);
  }
  if (fast == 4) return launch_scatter1_n<4, true, false>(threads, t, plan, g, H1off, out);      // 4-byte keys are always narrow
  if (fast == 8) return narrow ? launch_scatter1_n<8, true, false>(threads, t, plan, g, H1off, out)
                               : launch_scatter1_n<8, false, false>(threads, t, plan, g, H1off, out);
  return narrow ? launch_scatter1_n<0, true, false>(threads, t, plan, g, H1off, out)
                : launch_scatter1_n<0, false, false>(threads, t, plan, g, H1off, out);
}
// 8 / 4: the relation is one raw integer / float column of that width, with or without a validity mask (direct column
// reads, the mask read paired with the data: jk_scatter1<MASKED>, fetch_keys); 0: generic key construction
static int fast_key_width(const KeyTable &t, const KeyPlan &plan) {
  if (t.ncols != 1 || (plan.mode != KM_RAW_INT && plan.mode != KM_RAW_FLOAT) || t.nrows < 2) return 0;   // jk_scatter1 loads row pairs
  if (t.col[0].valid && t.nrows < 64) return 0;                                                          // ... and whole mask words
  if (t.col[0].width == 8) return 8;
  if (t.col[0].width == 4 && plan.narrow && plan.kmin == 0) return 4;
  return 0;
}
// LDS of a level-2 tile without the level-1-only arrays at the end of TileLds: three 12-byte-per-tuple tiles per CU instead of two
template <bool NARROW, int THREADS>
static constexpr size_t level2_lds_bytes() { using Tile = TileLds<NARROW, THREADS>; return offsetof(Tile, cursor); }
template <bool NARROW, int THREADS>
static gdf_error launch_scatter2_t(uint32_t ntiles, const PartGeom &g, Level2Map m, Tuples in, uint32_t *cursor, Tuples out) {
  const size_t lds = sizeof(TileLds<NARROW, THREADS>);
  HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<NARROW, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  m.ntiles = ntiles;
  m.xcd_order = (ntiles >= 64 && !lab::path_on("GDF_JK_NO_XCD_ORDER")) ? 1 : 0;
  const uint32_t grid = sc2_grid(m, ntiles);
  GDF_LAUNCH("jk_scatter2", (jk_scatter2<NARROW, THREADS>), dim3(grid), dim3(THREADS), lds, stream0(), g, m, in, cursor, out);
  HIP_CHECK_LAST();
  return GDF_SUCCESS;
}
static gdf_error launch_scatter2(bool narrow, int threads, uint32_t ntiles, const PartGeom &g, Level2Map m, Tuples in,
                                 uint32_t *cursor, Tuples out, bool p6 = false, bool in6 = false, int pw = 1) {
  if (in6) {                         // six-byte level-1 tuples in, six-byte level-2 tuples out
    if (!(p6 && !m.keys32 && g.b1 == 8)) return GDF_INVALID_API_CALL;
    m.ntiles = ntiles;
    m.xcd_order = (ntiles >= 64 && !lab::path_on("GDF_JK_NO_XCD_ORDER")) ? 1 : 0;
    const uint32_t grid = sc2_grid(m, ntiles);
    if (!narrow && lab::knob_int("GDF_JK_W10_SC2_THREADS", 0) == 512) {      // experiment: the same 4096-tuple tile under 512 threads
      using Tile = TileLds<false, 512, false, 8>;
      const size_t lds = offsetof(Tile, cursor);
      HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<false, 512, false, true, false, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GDF_LAUNCH("jk_scatter2_w10", (jk_scatter2<false, 512, false, true, false, true, 8>), dim3(grid), dim3(512), lds, stream0(), g, m, in, cursor, out);
      HIP_CHECK_LAST();
      return GDF_SUCCESS;
    }
    if (!narrow) {                   // ten-byte tuples in and out (WIDE keys)
      const size_t lds = level2_lds_bytes<false, 256>();
      HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<false, 256, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GDF_LAUNCH("jk_scatter2_w10", (jk_scatter2<false, 256, false, true, false, true>), dim3(grid), dim3(256), lds, stream0(), g, m, in, cursor, out);
      HIP_CHECK_LAST();
      return GDF_SUCCESS;
    }
    const size_t lds = sizeof(TileLds<true, 256>);
    HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<true, 256, false, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GDF_LAUNCH("jk_scatter2", (jk_scatter2<true, 256, false, true, false, true>), dim3(grid), dim3(256), lds, stream0(), g, m, in, cursor, out);
    HIP_CHECK_LAST();
    return GDF_SUCCESS;
  }
  if (p6 && !narrow) {               // WIDE (key64, row) tuples in, ten-byte tuples out (p10_key): the production tile size
    if (m.keys32) return GDF_INVALID_API_CALL;
    m.ntiles = ntiles;
    m.xcd_order = (ntiles >= 64 && !lab::path_on("GDF_JK_NO_XCD_ORDER")) ? 1 : 0;
    const uint32_t grid = sc2_grid(m, ntiles);
    const size_t lds = level2_lds_bytes<false, 256>();
    HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<false, 256, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GDF_LAUNCH("jk_scatter2_w10", (jk_scatter2<false, 256, false, true>), dim3(grid), dim3(256), lds, stream0(), g, m, in, cursor, out);
    HIP_CHECK_LAST();
    return GDF_SUCCESS;
  }
  if (p6 || m.keys32) {              // six-byte output tuples (see p6_store) and / or a receive buffer of 4-byte keys as input:
    m.ntiles = ntiles;                 // NARROW, no payload, the production tile size
    m.xcd_order = (ntiles >= 64 && !lab::path_on("GDF_JK_NO_XCD_ORDER")) ? 1 : 0;
    const uint32_t grid = sc2_grid(m, ntiles);
#define JK_SC2_LAUNCH(T, SIX, K)                                                                                                       \
  do {                                                                                                                                \
    const size_t lds = sizeof(TileLds<true, T>);                                                                                      \
    HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<true, T, false, SIX, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    GDF_LAUNCH("jk_scatter2", (jk_scatter2<true, T, false, SIX, K>), dim3(grid), dim3(T), lds, stream0(), g, m, in, cursor, out);      \
  } while (0)
    if (p6 && m.keys32) JK_SC2_LAUNCH(256, true, true);
    else if (p6) JK_SC2_LAUNCH(256, true, false);
    else JK_SC2_LAUNCH(256, false, true);
#undef JK_SC2_LAUNCH
    HIP_CHECK_LAST();
    return GDF_SUCCESS;
  }
  if (narrow && in.pay && pw == 2) { // ... two payload words per tuple (PayCarry mode 4): 2048-tuple tiles of 24 bytes, three workgroups per CU
    using Tile = TileLds<true, 256, 2, JK_PAY_ITEMS>;
    const size_t lds = offsetof(Tile, cursor);
    HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<true, 256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    m.ntiles = ntiles;
    m.xcd_order = (ntiles >= 64 && !lab::path_on("GDF_JK_NO_XCD_ORDER")) ? 1 : 0;
    const uint32_t grid = sc2_grid(m, ntiles);
    GDF_LAUNCH("jk_scatter2", (jk_scatter2<true, 256, 2>), dim3(grid), dim3(256), lds, stream0(), g, m, in, cursor, out);
    HIP_CHECK_LAST();
    return GDF_SUCCESS;
  }
  if (narrow && in.pay) {            // a probe side that carries its payload words (PayCarry): the production tile size only
    const size_t lds = sizeof(TileLds<true, 256, true, JK_PAY_ITEMS>);
    HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter2<true, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    m.ntiles = ntiles;
    m.xcd_order = (ntiles >= 64 && !lab::path_on("GDF_JK_NO_XCD_ORDER")) ? 1 : 0;
    const uint32_t grid = sc2_grid(m, ntiles);
    GDF_LAUNCH("jk_scatter2", (jk_scatter2<true, 256, true>), dim3(grid), dim3(256), lds, stream0(), g, m, in, cursor, out);
    HIP_CHECK_LAST();
    return GDF_SUCCESS;
  }
  if (narrow) {
    if (threads == 1024) return launch_scatter2_t<true, 1024>(ntiles, g, m, in, cursor, out);
    if (threads == 512) return launch_scatter2_t<true, 512>(ntiles, g, m, in, cursor, out);
    return launch_scatter2_t<true, 256>(ntiles, g, m, in, cursor, out);
  }
  if (threads >= 512) return launch_scatter2_t<false, 512>(ntiles, g, m, in, cursor, out);
  return launch_scatter2_t<false, 256>(ntiles, g, m, in, cursor, out);
}

// partitions one relation into g.fb-bit fine partitions
// decide_narrow: this is the BUILD side of an 8-byte integer key: the histogram pass also returns the key
// range, and when it spans < 2^32 the plan switches to the narrow tuple format (for both relations).
// Level 2 regroups 4096-tuple tiles, four workgroups per CU whose load / LDS / store phases overlap: since every fine
// partition is written from one XCD (jk_scatter2), the partial lines of its short (tile, bin) runs merge in that L2.
// Measured on C3, jk_scatter2 with 1024 / 512 / 256 threads: 3.6 / 3.3 / 3.2 ms.  Level 1 goes the other way
// (3.7 vs 4.2 ms with 512 threads): its claims sit in a tile loop, it keeps one big tile per CU.
// level-1 scatter of a payload-carrying probe side (jk_scatter1_pay)
static gdf_error launch_scatter1_pay(int fast, int pmode, const KeyTable &t, const KeyPlan &plan, const PartGeom &g, const uint32_t *H1off,
                                     const PaySrc &ps, Tuples out) {
#define JK_PAY_LAUNCH(F, M)                                                                                                   \
  do {                                                                                                                        \
    const size_t lds = sizeof(TileLds<true, JK_PAY_THREADS, pay_words(M), pay_items1(M)>);                                   \
    HIP_TRY(hipFuncSetAttribute((const void *)jk_scatter1_pay<F, M>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    GDF_LAUNCH("jk_scatter1", (jk_scatter1_pay<F, M>), dim3(g.nchunks), dim3(JK_PAY_THREADS), lds, stream0(), t, plan, g, H1off, ps, out); \
  } while (0)
  if (fast == 8) { if (pmode == 1) JK_PAY_LAUNCH(8, 1); else if (pmode == 2) JK_PAY_LAUNCH(8, 2); else if (pmode == 3) JK_PAY_LAUNCH(8, 3); else JK_PAY_LAUNCH(8, 4); }
  else { if (pmode == 1) JK_PAY_LAUNCH(4, 1); else if (pmode == 2) JK_PAY_LAUNCH(4, 2); else if (pmode == 3) JK_PAY_LAUNCH(4, 3); else JK_PAY_LAUNCH(4, 4); }
#undef JK_PAY_LAUNCH
  HIP_CHECK_LAST();
  return GDF_SUCCESS;
}

static int level2_threads(int sc_threads) {
  const int env = (int)lab::knob_int("GDF_JK_SC2_THREADS", 0);
  if (env == 256 || env == 512 || env == 1024) return env < sc_threads ? env : sc_threads;
  return sc_threads > 256 ? 256 : sc_threads;
}

// pay (may be null) + pmode: the relation carries a payload word per row (PayCarry); NARROW tuples and a FAST key column only
// keep_running (with device_index: a BUILD side of the main path): nothing waits for the last launches -- the scratch they use goes to
// sb->hold, the caller settles it (SideBufs::settle) once the stream has been synchronised for some other reason
static gdf_error partition_side(const KeyTable &t, KeyPlan &plan, PartGeom g, SideBufs *sb, bool decide_narrow, const PaySrc *pay = nullptr,
                                int pmode = 0, bool device_index = false, bool want_p6 = false, bool keep_running = false) {
  const int64_t n = t.nrows;
  bool narrow = plan.narrow != 0;
  // Chunks are SMALL (a few tiles) and processed in blockIdx order, so that the workgroups resident at
  // any moment write into a narrow window of every partition's output range: with a few thousand
  // 2 MiB pages live the scatter ran ~1.5x slower per row at 1e9 rows than at 5e8 (TLB reach).
  const int64_t chunk_rows_env = lab::knob_int("GDF_JK_CHUNK_ROWS", 0);
  int64_t chunk = chunk_rows_env ? chunk_rows_env : JK_CHUNK_ROWS;
  if (n / chunk > JK_MAX_CHUNKS) chunk = (n + JK_MAX_CHUNKS - 1) / JK_MAX_CHUNKS;
  constexpr int64_t MAX_TILE = 1024 * JK_SC_ITEMS;      // chunks are whole tiles for every tile size in use
  chunk = ((chunk + MAX_TILE - 1) / MAX_TILE) * MAX_TILE;
  g.chunk = chunk;
  g.nchunks = (int)((n + chunk - 1) / chunk);
  if (g.nchunks == 0) g.nchunks = 1;
  const uint32_t nfine = 1u << g.fb, ncoarse = 1u << g.b1;

  // the fine histogram and, behind it, one (min, max) pair per workgroup of jk_hist: one buffer, one read-back
  const int hist_grid = g.nchunks < NUM_CU ? g.nchunks : NUM_CU;
  const size_t hist_bytes = sizeof(uint32_t) * nfine + sizeof(long long) * 2 * (size_t)hist_grid;
  DevBuf fine_hist, H1;
  RMM_TRY(fine_hist.alloc(hist_bytes));
  RMM_TRY(H1.alloc(sizeof(uint32_t) * (size_t)ncoarse * g.nchunks));
  HIP_TRY(hipMemsetAsync(fine_hist.p, 0, sizeof(uint32_t) * nfine, stream0()));
  const size_t hist_lds = sizeof(uint32_t) * (nfine + ncoarse);
  // FAST: one 8-byte integer key column, no mask -> kernels read the column words directly
  const int fast = fast_key_width(t, plan);
  HIP_TRY(hipFuncSetAttribute((const void *)jk_hist<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds));
  HIP_TRY(hipFuncSetAttribute((const void *)jk_hist<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds));
  HIP_TRY(hipFuncSetAttribute((const void *)jk_hist<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hist_lds));
  long long *d_mm = nullptr;
  if (decide_narrow && !lab::knob_on("GDF_JK_WIDE"))         // GDF_JK_WIDE: experiment switch, force 12-byte tuples
    d_mm = reinterpret_cast<long long *>(fine_hist.as<uint32_t>() + nfine);
  if (fast == 8)
    GDF_LAUNCH("jk_hist", jk_hist<8>, dim3(hist_grid), dim3(JK_HIST_THREADS), hist_lds, stream0(), t, plan, g,
               fine_hist.as<uint32_t>(), H1.as<uint32_t>(), d_mm);
  else if (fast == 4)
    GDF_LAUNCH("jk_hist", jk_hist<4>, dim3(hist_grid), dim3(JK_HIST_THREADS), hist_lds, stream0(), t, plan, g,
               fine_hist.as<uint32_t>(), H1.as<uint32_t>(), d_mm);
  else
    GDF_LAUNCH("jk_hist", jk_hist<0>, dim3(hist_grid), dim3(JK_HIST_THREADS), hist_lds, stream0(), t, plan, g,
               fine_hist.as<uint32_t>(), H1.as<uint32_t>(), d_mm);
  HIP_CHECK_LAST();
  // the histogram's read-back is queued FIRST and the two scans behind it: they run while the host waits for the copy and lays the
  // partitions out (round 5: scan, synchronise, read back, scan, synchronise stood in a row here -- 0.2 ms from jk_hist's last wave to
  // level 1's first on C3's build side, most of it an idle GPU)
  std::vector<uint32_t> fh(hist_bytes / sizeof(uint32_t));
  ReadTicket hist_ticket;
  HIP_TRY(read_back_begin(&hist_ticket, fine_hist.p, d_mm ? hist_bytes : sizeof(uint32_t) * nfine, 0));
  DevBuf scan_a, scan_b, d_coarse, d_tiles, cursor;
  // (declared behind every buffer a queued launch may touch, i.e. destroyed in front of them: an early return waits for the stream first)
  struct Quiesce { bool armed = true; ~Quiesce() { if (armed) (void)hipStreamSynchronize(stream0()); } } quiesce;
  GDF_TRY(scan_u32_async(H1.as<uint32_t>(), H1.as<uint32_t>(), (size_t)ncoarse * g.nchunks, false, &scan_a));
  if (device_index) {
    // a BUILD side: jk_make_units wants the partition index on the device -- the histogram is there already and its exclusive scan is
    // the partitions' first tuples; uploading the two host vectors cost 0.1 ms of idle GPU behind the build side's last kernel
    RMM_TRY(sb->d_begin.alloc(sizeof(uint32_t) * nfine));
    GDF_TRY(scan_u32_async(fine_hist.as<uint32_t>(), sb->d_begin.as<uint32_t>(), nfine, false, &scan_b));
  }
  HIP_TRY(read_back_end(&hist_ticket, fh.data()));
  long long h[2] = {LLONG_MAX, LLONG_MIN};
  if (d_mm) {
    const long long *wg = reinterpret_cast<const long long *>(fh.data() + nfine);
    for (int b = 0; b < hist_grid; ++b) { h[0] = std::min(h[0], wg[2 * b]); h[1] = std::max(h[1], wg[2 * b + 1]); }
  }
  fh.resize(nfine);
  sb->fine_off.assign(nfine + 1, 0);
  for (uint32_t f = 0; f < nfine; ++f) sb->fine_off[f + 1] = sb->fine_off[f] + fh[f];
  sb->joinable = sb->fine_off[nfine];
  sb->fine_begin.assign(sb->fine_off.begin(), sb->fine_off.begin() + nfine);
  sb->fine_cnt = fh;
  sb->speculative = false;
  if (device_index) {
    sb->d_cnt.reset();
    sb->d_cnt.p = fine_hist.release();
  }
  const size_t cap = sb->joinable ? sb->joinable : 1;
  if (d_mm) {
    if (h[0] <= h[1] && (uint64_t)h[1] - (uint64_t)h[0] < 0xffffffffULL) {
      plan.narrow = 1;
      plan.kmin = (uint64_t)h[0];
      plan.kspan = (uint64_t)h[1] - (uint64_t)h[0];
      plan.klimit = plan.kspan;
      plan.kwindow = (plan.kmin & 0xffffffffULL) + plan.kspan < (1ULL << 32) ? 0xffffffffULL - (plan.kmin & 0xffffffffULL) : 0xffffffffULL;
      narrow = true;
    }
  }
  g.kbias = plan.kmin;
  // scatter tile: THREADS * 16 tuples regrouped in LDS per step.  Bigger tiles mean longer runs per
  // (tile, bin) -- DRAM-friendlier writes -- at the price of fewer resident workgroups.
  const int sc_threads_env = (int)lab::knob_int("GDF_JK_SC_THREADS", 0);
  // (WIDE tuples from one 8-byte key column: the 1024-thread tile with twelve tuples per thread, sc1_items -- 16384 x 12 B would not
  // fit 160 KiB; round 6, 512 threads before: C3's wide build side 0.64 ms in jk_scatter1)
  int sc_threads = sc_threads_env ? sc_threads_env : ((narrow || (fast == 8 && !lab::knob_on("GDF_JK_WIDE_512"))) ? 1024 : 512);   // swept on C3: profiles/r1_c_sweeps.md
  if (sc_threads != 256 && sc_threads != 512 && sc_threads != 1024) sc_threads = 256;
  if (!narrow && fast != 8 && sc_threads == 1024) sc_threads = 512;
  // a payload word only travels with NARROW tuples from a FAST, unmasked key column (jk_scatter1_pay); a build side that turns out
  // otherwise simply does not carry it (sb->pay stays empty and the caller gathers)
  if (pay && !(narrow && fast && !t.col[0].valid)) pay = nullptr;
  const int sc2_threads = pay ? 256 : level2_threads(sc_threads);
  const int64_t JK_TILE2 = (int64_t)sc2_threads * (pay ? JK_PAY_ITEMS : JK_SC_ITEMS);

  // + 2: the probe kernel's last 16-byte load may touch one tuple past the end; + 1024: jk_scatter1's per-thread dump slots
  g.dump = (uint32_t)(cap + 2);
  RMM_TRY(sb->w[0].alloc(sizeof(uint64_t) * (cap + 2 + 1024)));
  if (!narrow) RMM_TRY(sb->idx[0].alloc(sizeof(int32_t) * (cap + 2 + 1024)));
  const int pw = pay ? pay_words(pmode) : 1;
  sb->pay_words = pw;
  if (pay) RMM_TRY(sb->pay[0].alloc(sizeof(uint64_t) * pw * (cap + 2 + 1024)));
  const Tuples t0 = sb->tuples(0);
  // level 2's map and cursors are made and sent BEFORE level 1 is launched: the host has everything, and behind the level-1 kernel
  // every staged upload was a stall between the two levels
  const bool level2 = g.b2 > 0 && sb->joinable > 0;
  bool p6 = false;
  uint32_t ntiles = 0;
  std::vector<uint32_t> coarse_off, tile_prefix;
  if (level2) {
    coarse_off.resize(ncoarse + 1);
    tile_prefix.resize(ncoarse + 1);
    for (uint32_t c = 0; c <= ncoarse; ++c) coarse_off[c] = sb->fine_off[(size_t)c << g.b2];
    tile_prefix[0] = 0;
    for (uint32_t c = 0; c < ncoarse; ++c)
      tile_prefix[c + 1] = tile_prefix[c] + (coarse_off[c + 1] - coarse_off[c] + JK_TILE2 - 1) / JK_TILE2;
    ntiles = tile_prefix[ncoarse];
    RMM_TRY(d_coarse.alloc(sizeof(uint32_t) * (ncoarse + 1)));
    RMM_TRY(d_tiles.alloc(sizeof(uint32_t) * (ncoarse + 1)));
    RMM_TRY(cursor.alloc(sizeof(uint32_t) * nfine));
    if (device_index) {
      // the map and the cursors from the device copy of the index (the host keeps ntiles for the grid): no upload, no host vector in flight
      hipLaunchKernelGGL(jk_level2_index, dim3(1), dim3(JK_BK_THREADS), 0, stream0(), (const uint32_t *)sb->d_begin.as<uint32_t>(),
                         (const uint32_t *)sb->d_cnt.as<uint32_t>(), nfine, g.b2, (uint32_t)JK_TILE2, d_coarse.as<uint32_t>(), d_tiles.as<uint32_t>(),
                         cursor.as<uint32_t>());
      HIP_CHECK_LAST();
    } else {
      HIP_TRY(hipMemcpyAsync(d_coarse.p, coarse_off.data(), sizeof(uint32_t) * (ncoarse + 1), hipMemcpyHostToDevice, stream0()));
      HIP_TRY(hipMemcpyAsync(d_tiles.p, tile_prefix.data(), sizeof(uint32_t) * (ncoarse + 1), hipMemcpyHostToDevice, stream0()));
      // the cursors start at the partitions' first tuples: the host's prefix sums
      HIP_TRY(hipMemcpyAsync(cursor.p, sb->fine_off.data(), sizeof(uint32_t) * nfine, hipMemcpyHostToDevice, stream0()));
    }
    // (a probe side on the exact layout -- skewed probe keys -- writes six-byte tuples as the speculative layout does, see p6_store)
    // (WIDE keys: ten-byte tuples, the key's high words in idx[] -- p10_key)
    p6 = want_p6 && !pay && sc2_threads == 256;
    RMM_TRY(sb->w[1].alloc(p6 ? 6 * (cap + 2) + 16 : sizeof(uint64_t) * (cap + 2)));
    if (!narrow) RMM_TRY(sb->idx[1].alloc(sizeof(int32_t) * (cap + 2)));     // + 2: the lean probe kernel reads row numbers in pairs
    if (pay) RMM_TRY(sb->pay[1].alloc(sizeof(uint64_t) * pw * (cap + 2)));
  }
  if (pay) GDF_TRY(launch_scatter1_pay(fast, pmode, t, plan, g, H1.as<uint32_t>(), *pay, t0));
  else GDF_TRY(launch_scatter1(fast, narrow, sc_threads, t, plan, g, H1.as<uint32_t>(), t0));
  HIP_CHECK_LAST();
  sb->final_buf = 0;
  if (level2) {
    Level2Map m{d_coarse.as<uint32_t>(), d_coarse.as<uint32_t>() + 1, d_tiles.as<uint32_t>()};
    if (ntiles) GDF_TRY(launch_scatter2(narrow, sc2_threads, ntiles, g, m, sb->tuples(0), cursor.as<uint32_t>(), sb->tuples(1), p6, false, pw));
    HIP_CHECK_LAST();
    if (keep_running && device_index) {          // (no host vector is in flight on this path)
      for (DevBuf *b : {&sb->w[0], &sb->idx[0], &sb->pay[0], &d_coarse, &d_tiles, &cursor, &H1, &fine_hist, &scan_a, &scan_b}) sb->keep(*b);
      quiesce.armed = false;
    } else {
      HIP_TRY(hipStreamSynchronize(stream0()));   // host vectors + scratch go out of scope
      sb->w[0].reset();
      sb->idx[0].reset();
      sb->pay[0].reset();
    }
    sb->final_buf = 1;
    sb->p6 = p6;
  } else if (keep_running && device_index) {
    for (DevBuf *b : {&H1, &fine_hist, &scan_a, &scan_b}) sb->keep(*b);
    quiesce.armed = false;
  } else {
    HIP_TRY(hipStreamSynchronize(stream0()));
  }
  return GDF_SUCCESS;
}

// Histogram-free partitioning of one relation (PartGeom's SPECULATIVE layout): the 8 B/row histogram read is
// skipped, every partition gets room for its expected size + 8 standard deviations under a uniform hash.
// *ok = false: some partition outgrew its room (skewed keys); the caller repeats the side with
// partition_side().  The plan's tuple format must be final (the build side decides it).
// dup: expected rows per distinct key (probe rows / build rows): a partition's load is a sum over its keys, so
// its variance grows with the multiplicity -- measured on C3 (10 probe rows per key) before this term existed:
// the plain sqrt(mean) slack overflowed.
// A probe relation accumulated slice by slice (gdf_amd_join_probe_*): the level-2 buffer, its fill counters and the
// capacity of a fine partition persist across the slices; every slice gets its own level-1 pass.
struct SpecAppend {
  DevBuf cursor;               // [nfine] fill counters of the fine partitions
  uint32_t cap2 = 0;           // room per fine partition, from the EXPECTED total
  bool started = false;
  int64_t rows = 0;            // rows added so far = row number of the next slice's first row
};

static gdf_error partition_side_spec(const KeyTable &t, const KeyPlan &plan, PartGeom g, double dup, SideBufs *sb, bool *ok,
                                     SpecAppend *app = nullptr, bool defer = false, const PaySrc *pay = nullptr, int pmode = 0,
                                     bool want_p6 = false, const SkewCaps *caps = nullptr) {
  *ok = false;
  if (app) g.row_base = (int32_t)app->rows;
  const int64_t n = t.nrows;
  const bool narrow = plan.narrow != 0;
  const int64_t chunk_rows_env = lab::knob_int("GDF_JK_CHUNK_ROWS", 0);
  int64_t chunk = chunk_rows_env ? chunk_rows_env : JK_CHUNK_ROWS;
  if (n / chunk > JK_MAX_CHUNKS) chunk = (n + JK_MAX_CHUNKS - 1) / JK_MAX_CHUNKS;
  constexpr int64_t MAX_TILE = 1024 * JK_SC_ITEMS;
  chunk = ((chunk + MAX_TILE - 1) / MAX_TILE) * MAX_TILE;
  g.chunk = chunk;
  g.nchunks = (int)((n + chunk - 1) / chunk);
  if (g.nchunks == 0) g.nchunks = 1;
  const uint32_t nfine = 1u << g.fb, ncoarse = 1u << g.b1;
  const int fast = fast_key_width(t, plan);
  const int sc_threads_env = (int)lab::knob_int("GDF_JK_SC_THREADS", 0);
  // TEN-BYTE level-1 tuples for WIDE keys (W10, see p10_key): the conditions of the six-byte ones below on an 8-byte key column; the
  // level-1 tile is then 1024 threads x 12 tuples (decided here: the tile size goes into the buffer sizes)
  const bool w10 = want_p6 && defer && !app && !narrow && !pay && fast == 8 && !sc_threads_env && g.b1 == 8 && g.b2 > 0 && chunk == ((int64_t)1 << 17) &&
                   n < (((int64_t)1 << 30) - ((int64_t)1 << 23)) && g.row_base == 0 &&
                   ((n >= ((int64_t)1 << 26) && !lab::path_on("GDF_JK_NO_XCD_SPLIT")) || lab::path_on("GDF_JK_FORCE_L6")) && !lab::path_on("GDF_JK_NO_L6");
  int sc_threads = sc_threads_env ? sc_threads_env : ((narrow || w10 || (fast == 8 && !lab::knob_on("GDF_JK_WIDE_512"))) ? 1024 : 512);
  if (sc_threads != 256 && sc_threads != 512 && sc_threads != 1024) sc_threads = 256;
  if (!narrow && fast != 8 && sc_threads == 1024) sc_threads = 512;
  const int64_t JK_TILE = (int64_t)sc_threads * sc1_items(narrow, sc_threads);
  const int sc2_threads = pay ? 256 : level2_threads(sc_threads);
  const int64_t JK_TILE2 = (int64_t)sc2_threads * (pay ? JK_PAY_ITEMS : JK_SC_ITEMS);
  auto room = [dup](double mean, uint32_t align) {
    const double c = mean + 8.0 * std::sqrt(mean * (1.0 + dup)) + 64.0;
    return (uint32_t)(((uint64_t)c + align - 1) / align * align);
  };
  // one region per (coarse partition, XCD) when a second level follows (PartGeom::xs); a single level needs its
  // partitions contiguous for the probe units
  g.xs = (g.b2 > 0 && n >= ((int64_t)1 << 26) && !lab::path_on("GDF_JK_NO_XCD_SPLIT")) ? 3 : 0;
  // SIX-BYTE LEVEL-1 tuples (L6, see l6_pack): the deferred main path with six-byte level-2 tuples, a FAST key column on the
  // 1024-thread tile, 256 coarse partitions (24 hash bits left), chunks of exactly 2^17 rows and 64 regions per coarse partition
  // (the row number's bits 17..22), rows below 2^30 - 2^23 (seven explicit high bits, and the all-ones tuple stays free for padding).
  // GDF_JK_FORCE_L6: test switch, small relations too (their few chunks number the regions all the same); GDF_JK_NO_L6: off
  const bool l6 = w10 || (want_p6 && defer && !app && narrow && !pay && fast != 0 && sc_threads == 1024 && sc2_threads == 256 && g.b1 == 8 && g.b2 > 0 &&
                          chunk == ((int64_t)1 << 17) && n < (((int64_t)1 << 30) - ((int64_t)1 << 23)) && g.row_base == 0 &&
                          (g.xs == 3 || lab::path_on("GDF_JK_FORCE_L6")) && !lab::path_on("GDF_JK_NO_L6"));
  if (l6) g.xs = 6;
  const uint32_t nseg = ncoarse << g.xs;
  uint32_t cap1 = room((double)n / nseg, 64), cap2 = app ? app->cap2 : (g.b2 ? room((double)n / nfine, 8) : 0);
  uint64_t size1 = (uint64_t)nseg * cap1 + JK_TILE, size2 = (uint64_t)nfine * cap2 + JK_TILE;
  // SKEWED probe keys (SkewCaps): per-region / per-partition room from the capacity sample -- the estimate + 6 sigma of it, and for a
  // level-1 region 8 sigma of its own share on top (a region holds 1 / 2^xs of its coarse partition's rows).  Deferred path only.
  std::vector<uint32_t> cap_words;
  if (caps) {
    if (!(defer && !app && g.b2 > 0 && !pay && caps->fine.size() == nfine)) return GDF_SUCCESS;
    cap_words.resize(2 * (size_t)nseg + 2 * (size_t)nfine);
    uint32_t *rstart = cap_words.data(), *rcap = rstart + nseg, *fstart = rcap + nseg, *fcap = fstart + nfine;
    const double scale = caps->rows_per_sample;
    uint64_t run1 = 0, run2 = 0;
    uint32_t max1 = 0, max2 = 0;
    const uint32_t nreg = 1u << g.xs;
    for (uint32_t c = 0; c < ncoarse; ++c) {
      double sc = 0;
      for (uint32_t f = c << g.b2; f < ((c + 1) << g.b2); ++f) {
        const double sf = (double)caps->fine[f];
        sc += sf;
        const double U = (sf + 6.0 * std::sqrt(sf + 1.0) + 4.0) * scale;
        const uint64_t cf = ((uint64_t)(U + 64.0) + 7) / 8 * 8;
        fstart[f] = (uint32_t)std::min<uint64_t>(run2, 0xffffffffULL);
        fcap[f] = (uint32_t)std::min<uint64_t>(cf, 0x7fffffffULL);
        run2 += cf;
        max2 = std::max(max2, fcap[f]);
      }
      const double Uc = (sc + 6.0 * std::sqrt(sc + 1.0) + 4.0) * scale / (double)nreg;
      const uint64_t cr = ((uint64_t)(Uc + 8.0 * std::sqrt(Uc) + 64.0) + 63) / 64 * 64;
      for (uint32_t r = 0; r < nreg; ++r) {
        rstart[c * nreg + r] = (uint32_t)std::min<uint64_t>(run1, 0xffffffffULL);
        rcap[c * nreg + r] = (uint32_t)std::min<uint64_t>(cr, 0x7fffffffULL);
        run1 += cr;
        max1 = std::max(max1, rcap[c * nreg + r]);
      }
    }
    size1 = run1 + JK_TILE;
    size2 = run2 + JK_TILE;
    cap1 = max1;
    cap2 = max2;
  }
  if (size1 >= 0x7fffffffULL || size2 >= 0x7fffffffULL) return GDF_SUCCESS;      // tuple positions are 31-bit

  DevBuf spec;
  constexpr size_t ZERO_WORDS = 32;                            // SideBufs::zero
  const size_t zero_at = ((size_t)nseg + 2) / 2 * 2;          // (the first even word index behind the nseg + 1 counters)
  RMM_TRY(spec.alloc(sizeof(uint32_t) * (zero_at + ZERO_WORDS)));
  HIP_TRY(hipMemsetAsync(spec.p, 0, sizeof(uint32_t) * (zero_at + ZERO_WORDS), stream0()));
  g.kbias = plan.kmin;
  g.cap1 = cap1;
  g.cap2 = 0;
  g.dump = caps ? (uint32_t)(size1 - JK_TILE) : nseg * cap1;
  if (caps) {
    RMM_TRY(sb->d_caps.alloc(sizeof(uint32_t) * cap_words.size()));
    HIP_TRY(hipMemcpyAsync(sb->d_caps.p, cap_words.data(), sizeof(uint32_t) * cap_words.size(), hipMemcpyHostToDevice, stream0()));
    HIP_TRY(hipStreamSynchronize(stream0()));                  // (cap_words is a local)
    g.rstart = sb->d_caps.as<uint32_t>();
    g.rcap = g.rstart + nseg;
    g.fstart = g.rcap + nseg;
    g.fcap = g.fstart + nfine;
    sb->d_fstart = g.fstart;
    sb->d_fcap = g.fcap;
  }
  g.spec_cursor1 = spec.as<uint32_t>();
  g.spec_flag = spec.as<uint32_t>() + nseg;
  // the deferred main path allocates its two tuple buffers as PLACED blocks (DevBuf::alloc_placed; memory.h): the pool re-draws a
  // physical placement the regroup kernels run slowly on, judged by the times reported here
  // (every shape of the deferred path: WIDE tuples -- their row numbers, idx[], stay plain allocations -- and payload-carrying ones too)
  const bool placed = defer && !app && g.b2 > 0;
  // TEN-byte tuples (W10): the 4 GB of high words are a placed block of their own role, drawn, timed and kept or dropped in step with
  // the six-byte stream's (both see the same calibration times, like the two output columns).  (Behind the six-byte stream in ONE
  // 10 GB block, jk_scatter1 took 4.4 ms for C3's probe side in three processes of three, 3.9 - 4.0 in two plain allocations:
  // profiles/r6_e_*; level 2 does not care and keeps its high words inside its block.)
  const bool hi_placed1 = placed && l6 && !narrow;
  const size_t bytes1 = l6 ? 6 * ((size_t)size1 + 2048) + 16 : sizeof(uint64_t) * size1;
  const size_t bytes1_hi = sizeof(int32_t) * ((size_t)size1 + (l6 ? 2048 : 0));      // (W10: the high words, dump slots as in w[0])
  // (place_draws_now: the call's budget for candidates, internal.h -- once it is spent the pool is asked to HOLD its champions)
  const bool calibrate = !lab::knob_on("GDF_JK_NO_CALIBRATE");
  if (placed) RMM_TRY(sb->w[0].alloc_placed(JK_ROLE_LEVEL1, bytes1, calibrate ? place_draws_now(JK_PLACE_DRAWS) : JK_PLACE_DRAWS));
  else RMM_TRY(sb->w[0].alloc(bytes1));      // (L6: + two dump slots per thread behind the regions)
  if (hi_placed1) RMM_TRY(sb->idx[0].alloc_placed(JK_ROLE_LEVEL1_HI, bytes1_hi, -1));      // (held while the six-byte stream's block is chosen)
  else if (!narrow) RMM_TRY(sb->idx[0].alloc(bytes1_hi));
  const int pw = pay ? pay_words(pmode) : 1;
  sb->pay_words = pw;
  if (pay) RMM_TRY(sb->pay[0].alloc(sizeof(uint64_t) * pw * size1));
#ifdef GDF_AMD_LAB
  DevBuf lab_clk;
  if (lab::knob_on("GDF_JK_CLOCK") && !pay) {
    RMM_TRY(lab_clk.alloc(sizeof(unsigned long long) * 16 * (size_t)g.nchunks));
    HIP_TRY(hipMemsetAsync(lab_clk.p, 0, sizeof(unsigned long long) * 16 * (size_t)g.nchunks, stream0()));
    g.lab_clock = lab_clk.as<unsigned long long>();
  }
#endif
  // PLACEMENT TOURNAMENT of the level-1 buffer (round 5).  jk_scatter1 runs in one of two modes on a given physical placement of this
  // buffer (2.9 or 3.3 ms for C3's probe side), and most fresh blocks are slow ones.  While the pool is still comparing
  // placements for this (role, size) -- the first call of a shape -- every candidate block is timed on a CALIBRATION run, the real kernel
  // over the first quarter of the chunks (every region's write front opens, ~0.8 ms), and handed back with that time; the pool keeps
  // the fastest of JK_PLACE_DRAWS + 1 and this call, and every later one, runs on it.  ~2 ms per candidate, once per shape.
  if (placed && calibrate) {
    for (int round = 0; round <= JK_PLACE_DRAWS && sb->w[0].measure; ++round) {
      PlaceRound charge;
      PartGeom gc = g;
      gc.nchunks = std::max(1, g.nchunks / 4);
      sb->w[0].clock_begin(stream0());
      if (pay) GDF_TRY(launch_scatter1_pay(fast, pmode, t, plan, gc, nullptr, *pay, sb->tuples(0)));
      else GDF_TRY(launch_scatter1(fast, narrow, sc_threads, t, plan, gc, nullptr, sb->tuples(0), l6));
      sb->w[0].clock_end(stream0());
      HIP_TRY(hipMemsetAsync(spec.p, 0, sizeof(uint32_t) * (nseg + 1), stream0()));
      RMM_TRY(sb->w[0].alloc_placed(JK_ROLE_LEVEL1, bytes1, place_draws_now(JK_PLACE_DRAWS)));      // (reset() reports the time; the champion or the next challenger comes back)
    }
    // ... then the high words' block, with the six-byte stream on its champion: one coordinate at a time.  (Both kinds of block have
    // their fast and slow placements, about one fresh block in five a fast one; candidates drawn and judged in PAIRS kept a slow
    // block of one kind or the other in two processes of three, 4.4 instead of 3.8 ms in jk_scatter1: profiles/r6_f_*)
    // (only once the six-byte stream's search is over -- its block is then the same in every run that times a candidate here)
    if (hi_placed1 && !sb->w[0].measure && place_budget_left()) RMM_TRY(sb->idx[0].alloc_placed(JK_ROLE_LEVEL1_HI, bytes1_hi, JK_PLACE_DRAWS));
    for (int round = 0; hi_placed1 && round <= JK_PLACE_DRAWS && sb->idx[0].measure; ++round) {
      PlaceRound charge;
      PartGeom gc = g;
      gc.nchunks = std::max(1, g.nchunks / 4);
      sb->idx[0].clock_begin(stream0());
      GDF_TRY(launch_scatter1(fast, narrow, sc_threads, t, plan, gc, nullptr, sb->tuples(0), l6));
      sb->idx[0].clock_end(stream0());
      HIP_TRY(hipMemsetAsync(spec.p, 0, sizeof(uint32_t) * (nseg + 1), stream0()));
      RMM_TRY(sb->idx[0].alloc_placed(JK_ROLE_LEVEL1_HI, bytes1_hi, place_draws_now(JK_PLACE_DRAWS)));
    }
  }
  // DEFERRED: level 2's fill cursors are set IN FRONT of level 1 (nothing of level 1 is in them): one launch less between the two kernels
  DevBuf d_map, cursor;
  if (defer && !app && g.b2 > 0) {
    RMM_TRY(cursor.alloc(sizeof(uint32_t) * ((size_t)nfine + 2)));             // fill cursors | level-2 overflow flag | tile count
    hipLaunchKernelGGL(jk_init_cursor, dim3(32), dim3(256), 0, stream0(), cursor.as<uint32_t>(), nfine, cap2, g.fstart);
    HIP_CHECK_LAST();
  }
  sb->w[0].clock_begin(stream0());
  if (pay) GDF_TRY(launch_scatter1_pay(fast, pmode, t, plan, g, nullptr, *pay, sb->tuples(0)));
  else GDF_TRY(launch_scatter1(fast, narrow, sc_threads, t, plan, g, nullptr, sb->tuples(0), l6));
  sb->w[0].clock_end(stream0());
#ifdef GDF_AMD_LAB
  if (g.lab_clock) {
    std::vector<unsigned long long> h(16 * (size_t)g.nchunks);
    HIP_TRY(read_back(h.data(), lab_clk.p, sizeof(unsigned long long) * h.size()));
    static const char *names[6] = {"hash + rank + barrier", "claim issue + scan + barrier", "regroup issue + claim answer", "prefetch issue + barrier",
                                   "flush (LDS reads, store issue)", "wait for the next tile's keys"};
    for (int wv = 0; wv < 2; ++wv) {
      double sum[6] = {0, 0, 0, 0, 0, 0}, all = 0;
      for (int c = 0; c < g.nchunks; ++c) for (int i = 0; i < 6; ++i) { sum[i] += (double)h[((size_t)c * 2 + wv) * 8 + i]; all += (double)h[((size_t)c * 2 + wv) * 8 + i]; }
      fprintf(stderr, "jk_scatter1 phase clock, %s wave of every workgroup (%d chunks, %.0f cycles per chunk):\n", wv ? "last" : "first", g.nchunks, all / g.nchunks);
      for (int i = 0; i < 6; ++i) fprintf(stderr, "  %-34s %5.1f %%\n", names[i], 100.0 * sum[i] / all);
    }
    g.lab_clock = nullptr;
  }
#endif
#ifdef GDF_AMD_LAB
  if (lab::knob_on("GDF_JK_TRACE")) {
    std::vector<uint32_t> c1(nseg + 1);
    HIP_TRY(read_back(c1.data(), spec.p, sizeof(uint32_t) * (nseg + 1)));
    uint64_t sum = 0; uint32_t mx = 0, nz = 0;
    for (uint32_t i = 0; i < nseg; ++i) { sum += c1[i]; mx = std::max(mx, c1[i]); nz += c1[i] != 0; }
    fprintf(stderr, "level 1: nseg %u cap1 %u xs %d l6 %d chunks %d: fill sum %llu max %u nonzero %u flag %u\n", nseg, cap1, g.xs, (int)l6, g.nchunks,
            (unsigned long long)sum, mx, nz, c1[nseg]);
  }
#endif
  if (defer && !app && g.b2 > 0) {
    // DEFERRED: the level-2 map and the fill cursors are made on the device, nothing is read back here; the overflow flags
    // of both levels are looked at once, with the work units (jk_make_units / probe_partitioned)
    RMM_TRY(d_map.alloc(sizeof(uint32_t) * (3 * (size_t)nseg + 2)));          // segment begins | segment ends | tile prefix [nseg + 1]
    uint32_t *seg_begin = d_map.as<uint32_t>(), *seg_end = seg_begin + nseg, *tile_prefix = seg_end + nseg;
    uint32_t *ntiles_dev = cursor.as<uint32_t>() + nfine + 1;
    size_t map_lds = 0;
    GDF_TRY(l2map_prepare(nseg, &map_lds));
    hipLaunchKernelGGL(jk_make_l2map, dim3(1), dim3(JK_BK_THREADS), map_lds, stream0(), (const uint32_t *)spec.as<uint32_t>(), nseg, cap1,
                       (uint32_t)JK_TILE2, seg_begin, seg_end, tile_prefix, ntiles_dev, 0u, 0u, g.rstart, g.rcap);
    HIP_CHECK_LAST();
    const bool p6 = want_p6 && !pay && sc2_threads == 256;
    const bool hi_inside2 = placed && p6 && !narrow;          // (ten-byte tuples: the high words behind the six-byte stream, as at level 1)
    const size_t six2 = (6 * (size_t)size2 + 16 + 255) & ~(size_t)255;
    const size_t bytes2 = p6 ? (hi_inside2 ? six2 + sizeof(int32_t) * ((size_t)size2 + 2) : 6 * (size_t)size2 + 16) : sizeof(uint64_t) * size2;
    if (placed) RMM_TRY(sb->w[1].alloc_placed(JK_ROLE_LEVEL2, bytes2, calibrate ? place_draws_now(JK_PLACE_DRAWS_L2) : JK_PLACE_DRAWS_L2));
    else RMM_TRY(sb->w[1].alloc(bytes2));
    if (hi_inside2) sb->idx[1].borrow(sb->w[1].as<char>() + six2);
    else if (!narrow) RMM_TRY(sb->idx[1].alloc(sizeof(int32_t) * size2));
    if (pay) RMM_TRY(sb->pay[1].alloc(sizeof(uint64_t) * pw * size2));
    PartGeom g2 = g;
    g2.cap2 = cap2;
    g2.dump = caps ? (uint32_t)(size2 - JK_TILE) : nfine * cap2;
    g2.spec_flag = cursor.as<uint32_t>() + nfine;
    Level2Map m{seg_begin, seg_end, tile_prefix, g.xs};
    m.ntiles_dev = ntiles_dev;
    // every segment ends in at most one partial tile: an upper bound of the tile count sizes the grid
    const uint32_t tile_bound = (uint32_t)((uint64_t)n / (uint64_t)JK_TILE2) + nseg + 1;
    // PLACEMENT TOURNAMENT of the level-2 buffer, as for level 1 above: every candidate is timed on a calibration run of the real
    // kernel over every fourth tile (all 2^15 write fronts open), the fill cursors are set back, the pool keeps the fastest
    if (placed && calibrate) {
      for (int round = 0; round <= JK_PLACE_DRAWS_L2 && sb->w[1].measure; ++round) {
        PlaceRound charge;
        Level2Map mc = m;
        mc.calib_step = 4;
        sb->w[1].clock_begin(stream0());
        GDF_TRY(launch_scatter2(narrow, sc2_threads, tile_bound, g2, mc, sb->tuples(0), cursor.as<uint32_t>(), sb->tuples(1), p6, l6, pw));
        sb->w[1].clock_end(stream0());
        hipLaunchKernelGGL(jk_init_cursor, dim3(32), dim3(256), 0, stream0(), cursor.as<uint32_t>(), nfine, cap2, g.fstart);      // (+ the overflow flag)
        HIP_CHECK_LAST();
        RMM_TRY(sb->w[1].alloc_placed(JK_ROLE_LEVEL2, bytes2, place_draws_now(JK_PLACE_DRAWS_L2)));
        if (hi_inside2) sb->idx[1].borrow(sb->w[1].as<char>() + six2);
      }
    }
    sb->w[1].clock_begin(stream0());
    GDF_TRY(launch_scatter2(narrow, sc2_threads, tile_bound, g2, m, sb->tuples(0), cursor.as<uint32_t>(), sb->tuples(1), p6, l6, pw));
    sb->w[1].clock_end(stream0());
    sb->p6 = p6;
    // no synchronisation: the map and the level-1 tuples stay allocated until probe_partitioned has read its state block
    sb->d_map.p = d_map.release();
    sb->final_buf = 1;
    sb->fine_off.clear(); sb->fine_begin.clear(); sb->fine_cnt.clear();
    sb->deferred = true;
    sb->speculative = true;
    sb->cap2 = cap2;
    sb->nseg = nseg;
    sb->zero = reinterpret_cast<unsigned long long *>(spec.as<uint32_t>() + zero_at);
    sb->d_level1.p = spec.release();
    sb->d_cursor.p = cursor.release();
    *ok = true;
    return GDF_SUCCESS;
  }
  std::vector<uint32_t> c1(nseg + 1);
  HIP_TRY(read_back(c1.data(), spec.p, sizeof(uint32_t) * (nseg + 1)));
  if (c1[nseg]) return GDF_SUCCESS;
  sb->final_buf = 0;
  sb->fine_off.clear();
  if (app && g.b2 == 0) return GDF_SUCCESS;      // (the caller never asks: accumulation needs the two-level layout)
  if (g.b2 == 0) {
    sb->fine_begin.resize(nfine);
    for (uint32_t f = 0; f < nfine; ++f) sb->fine_begin[f] = f * cap1;
    sb->fine_cnt.assign(c1.begin(), c1.begin() + nfine);
  } else {
    std::vector<uint32_t> coarse_off(2 * nseg), tile_prefix(nseg + 1), cur(nfine);     // per SEGMENT (Level2Map::xs)
    tile_prefix[0] = 0;
    for (uint32_t c = 0; c < nseg; ++c) {
      coarse_off[c] = c * cap1;
      coarse_off[nseg + c] = c * cap1 + c1[c];
      tile_prefix[c + 1] = tile_prefix[c] + (uint32_t)((c1[c] + JK_TILE2 - 1) / JK_TILE2);
    }
    for (uint32_t f = 0; f < nfine; ++f) cur[f] = f * cap2;
    const uint32_t ntiles = tile_prefix[nseg];
    DevBuf d_coarse, d_tiles, own_cursor;
    DevBuf &cursor = app ? app->cursor : own_cursor;
    RMM_TRY(d_coarse.alloc(sizeof(uint32_t) * 2 * nseg));
    RMM_TRY(d_tiles.alloc(sizeof(uint32_t) * (nseg + 1)));
    HIP_TRY(hipMemcpyAsync(d_coarse.p, coarse_off.data(), sizeof(uint32_t) * 2 * nseg, hipMemcpyHostToDevice, stream0()));
    HIP_TRY(hipMemcpyAsync(d_tiles.p, tile_prefix.data(), sizeof(uint32_t) * (nseg + 1), hipMemcpyHostToDevice, stream0()));
    if (!app || !app->started) {        // the first (or only) slice sets up the level-2 buffer and its fill counters
      RMM_TRY(cursor.alloc(sizeof(uint32_t) * ((size_t)nfine + 1)));      // + the level-2 overflow flag: one read-back for both
      cur.push_back(0u);
      HIP_TRY(hipMemcpyAsync(cursor.p, cur.data(), sizeof(uint32_t) * ((size_t)nfine + 1), hipMemcpyHostToDevice, stream0()));
      cur.pop_back();
      RMM_TRY(sb->w[1].alloc(sizeof(uint64_t) * size2));
      if (!narrow) RMM_TRY(sb->idx[1].alloc(sizeof(int32_t) * size2));
      if (pay) RMM_TRY(sb->pay[1].alloc(sizeof(uint64_t) * pw * size2));
      if (app) app->started = true;
    }
    PartGeom g2 = g;
    g2.cap2 = cap2;
    g2.dump = nfine * cap2;
    g2.spec_flag = cursor.as<uint32_t>() + nfine;
    Level2Map m{d_coarse.as<uint32_t>(), d_coarse.as<uint32_t>() + nseg, d_tiles.as<uint32_t>(), g.xs};
    if (ntiles) GDF_TRY(launch_scatter2(narrow, sc2_threads, ntiles, g2, m, sb->tuples(0), cursor.as<uint32_t>(), sb->tuples(1), false, false, pw));
    uint32_t flag = 0;
    if (!app) {
      cur.resize((size_t)nfine + 1);
      HIP_TRY(read_back(cur.data(), cursor.p, sizeof(uint32_t) * ((size_t)nfine + 1)));
      flag = cur[nfine];
      cur.pop_back();
    } else {
      HIP_TRY(read_back(&flag, cursor.as<uint32_t>() + nfine, sizeof(uint32_t)));
    }
    sb->w[0].reset();
    sb->idx[0].reset();
    sb->pay[0].reset();
    if (flag) return GDF_SUCCESS;
    if (app) {                          // the fill counters are read once, by spec_append_finish
      app->rows += n;
      *ok = true;
      return GDF_SUCCESS;
    }
    sb->fine_begin.resize(nfine);
    sb->fine_cnt.resize(nfine);
    for (uint32_t f = 0; f < nfine; ++f) { sb->fine_begin[f] = f * cap2; sb->fine_cnt[f] = cur[f] - f * cap2; }
    sb->final_buf = 1;
  }
  uint64_t total = 0;
  for (uint32_t c : sb->fine_cnt) total += c;
  sb->joinable = (uint32_t)total;
  sb->speculative = true;
  *ok = true;
  return GDF_SUCCESS;
}

// the fine partitions of an accumulated probe relation, once all slices are in
static gdf_error spec_append_finish(const PartGeom &g, SpecAppend *app, SideBufs *sb) {
  const uint32_t nfine = 1u << g.fb;
  std::vector<uint32_t> cur(nfine);
  HIP_TRY(read_back(cur.data(), app->cursor.p, sizeof(uint32_t) * nfine));
  sb->fine_off.clear();
  sb->fine_begin.resize(nfine);
  sb->fine_cnt.resize(nfine);
  uint64_t total = 0;
  for (uint32_t f = 0; f < nfine; ++f) {
    sb->fine_begin[f] = f * app->cap2;
    sb->fine_cnt[f] = cur[f] - f * app->cap2;
    total += sb->fine_cnt[f];
  }
  sb->final_buf = 1;
  sb->joinable = (uint32_t)total;
  sb->speculative = true;
  return GDF_SUCCESS;
}

template <bool NARROW>
static gdf_error run_probe(bool write, const char *name, size_t nunits, size_t lds, const ProbeArgs &a, const KeyTable &probe_t,
                           const KeyTable &build_t) {
  if (!nunits) return GDF_SUCCESS;
  if constexpr (!NARROW) {
    if (a.p6_fb) {                 // ten-byte probe tuples
      if (write) {
        HIP_TRY(hipFuncSetAttribute((const void *)jk_probe<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GDF_LAUNCH(name, (jk_probe<true, false, true>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), lds, stream0(), a, probe_t, build_t);
      } else {
        HIP_TRY(hipFuncSetAttribute((const void *)jk_probe<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GDF_LAUNCH(name, (jk_probe<false, false, true>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), lds, stream0(), a, probe_t, build_t);
      }
      HIP_CHECK_LAST();
      return GDF_SUCCESS;
    }
  }
  if constexpr (NARROW) {
    if (a.p6_fb) {                 // six-byte probe tuples
      if (write) {
        HIP_TRY(hipFuncSetAttribute((const void *)jk_probe<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GDF_LAUNCH(name, (jk_probe<true, true, true>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), lds, stream0(), a, probe_t, build_t);
      } else {
        HIP_TRY(hipFuncSetAttribute((const void *)jk_probe<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GDF_LAUNCH(name, (jk_probe<false, true, true>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), lds, stream0(), a, probe_t, build_t);
      }
      HIP_CHECK_LAST();
      return GDF_SUCCESS;
    }
  }
  if (write) {
    HIP_TRY(hipFuncSetAttribute((const void *)jk_probe<true, NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GDF_LAUNCH(name, (jk_probe<true, NARROW>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), lds, stream0(), a, probe_t, build_t);
  } else {
    HIP_TRY(hipFuncSetAttribute((const void *)jk_probe<false, NARROW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GDF_LAUNCH(name, (jk_probe<false, NARROW>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), lds, stream0(), a, probe_t, build_t);
  }
  HIP_CHECK_LAST();
  return GDF_SUCCESS;
}
static gdf_error run_probe(bool narrow, bool write, const char *name, size_t nunits, size_t lds, const ProbeArgs &a,
                           const KeyTable &probe_t, const KeyTable &build_t) {
  return narrow ? run_probe<true>(write, name, nunits, lds, a, probe_t, build_t)
                : run_probe<false>(write, name, nunits, lds, a, probe_t, build_t);
}

// WRITE pass over all units.  PLAIN joins (see jk_probe_fast) run the lean kernel first and hand the units whose
// cuckoo build did not settle to the general one.  a.opt_state must point at 3 zeroed counters.
static gdf_error run_write_pass(bool narrow, bool plain, size_t nunits, size_t lds, ProbeArgs a, uint32_t max_build,
                                const KeyTable &probe_t, const KeyTable &build_t) {
  if (!nunits) return GDF_SUCCESS;
  if (!plain || lab::knob_on("GDF_JK_NO_FAST") || (!narrow && lab::knob_on("GDF_JK_NO_FAST_WIDE")))
    return run_probe(narrow, true, "jk_probe_write", nunits, lds, a, probe_t, build_t);
  DevBuf todo;
  RMM_TRY(todo.alloc(sizeof(uint32_t) * nunits));
  a.unit_todo = todo.as<uint32_t>();
  // The general kernel's H is the power of two >= the largest build partition: between 25 % and 50 % table load.
  // A cuckoo build above ~42 % runs into cycles often (measured at 1.25e8 build rows: 3814 tuples per partition in
  // 2 x 4096 slots), so the lean kernel then sizes its tables for 40 % and takes slots with a mulhi.
  ProbeArgs fa = a;
  const bool pow2 = (double)max_build <= 0.42 * 2.0 * (double)a.nslots;
  if (!pow2) fa.nslots = ((uint32_t)(max_build * 1.25) + 63) & ~63u;
  // (the lean kernels never chain: without the general kernel's next[] a WIDE image of C3's partitions is 73 KB instead of 86 -- TWO
  // workgroups per CU instead of one; round 6, the 12-byte tuples' probe pass had run at half occupancy since round 2)
  const size_t flds = probe_lds_bytes(narrow, a.cap, fa.nslots, false);
#define JK_FAST_LAUNCH(...)                                                                                                      \
  do {                                                                                                                           \
    HIP_TRY(hipFuncSetAttribute((const void *)jk_probe_fast<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds)); \
    GDF_LAUNCH("jk_probe_write", (jk_probe_fast<__VA_ARGS__>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), flds, stream0(), fa); \
  } while (0)
  const bool keep = a.keep_unmatched_probe != 0;
  if (narrow && a.bpay_mode && !keep) {        // the build relation's payload word rides in the LDS image (jk_probe_bp)
    const bool bw2 = a.bpay_mode == 4;
    const size_t blds = probe_bp_lds_bytes(a.cap, fa.nslots, bw2);
#define JK_BP_LAUNCH(P2, PPAY, B2)                                                                                                \
  do {                                                                                                                            \
    HIP_TRY(hipFuncSetAttribute((const void *)jk_probe_bp<P2, PPAY, B2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)blds)); \
    GDF_LAUNCH("jk_probe_write", (jk_probe_bp<P2, PPAY, B2>), dim3((unsigned)nunits), dim3(JK_BP_THREADS), blds, stream0(), fa);  \
  } while (0)
#define JK_BP_PICK(P2)                                                                                                            \
  do {                                                                                                                            \
    if (a.pay_mode == 4 && bw2) JK_BP_LAUNCH(P2, 2, true); else if (a.pay_mode == 4) JK_BP_LAUNCH(P2, 2, false);                   \
    else if (a.pay_mode && bw2) JK_BP_LAUNCH(P2, 1, true); else if (a.pay_mode) JK_BP_LAUNCH(P2, 1, false);                        \
    else if (bw2) JK_BP_LAUNCH(P2, 0, true); else JK_BP_LAUNCH(P2, 0, false);                                                      \
  } while (0)
    if (pow2) JK_BP_PICK(true); else JK_BP_PICK(false);
#undef JK_BP_PICK
#undef JK_BP_LAUNCH
  } else if (narrow && a.p6_fb) {      // six-byte probe tuples (never together with carried columns)
    if (pow2 && keep) JK_FAST_LAUNCH(true, true, true, 0, true);
    else if (pow2) JK_FAST_LAUNCH(true, false, true, 0, true);
    else if (keep) JK_FAST_LAUNCH(false, true, true, 0, true);
    else JK_FAST_LAUNCH(false, false, true, 0, true);
  } else if (narrow && a.pay_mode) {
#define JK_FAST_PAY(M)                                                                                                             \
  do {                                                                                                                             \
    if (pow2 && keep) JK_FAST_LAUNCH(true, true, true, M); else if (pow2) JK_FAST_LAUNCH(true, false, true, M);                    \
    else if (keep) JK_FAST_LAUNCH(false, true, true, M); else JK_FAST_LAUNCH(false, false, true, M);                                \
  } while (0)
    if (a.pay_mode == 1) JK_FAST_PAY(1); else if (a.pay_mode == 2) JK_FAST_PAY(2); else if (a.pay_mode == 3) JK_FAST_PAY(3); else JK_FAST_PAY(4);
#undef JK_FAST_PAY
  } else if (narrow) {
    if (pow2 && keep) JK_FAST_LAUNCH(true, true, true);
    else if (pow2) JK_FAST_LAUNCH(true, false, true);
    else if (keep) JK_FAST_LAUNCH(false, true, true);
    else JK_FAST_LAUNCH(false, false, true);
  } else if (a.p6_fb) {                // WIDE keys on ten-byte probe tuples
    if (pow2 && keep) JK_FAST_LAUNCH(true, true, false, 0, true);
    else if (pow2) JK_FAST_LAUNCH(true, false, false, 0, true);
    else if (keep) JK_FAST_LAUNCH(false, true, false, 0, true);
    else JK_FAST_LAUNCH(false, false, false, 0, true);
  } else {
    if (pow2 && keep) JK_FAST_LAUNCH(true, true, false);
    else if (pow2) JK_FAST_LAUNCH(true, false, false);
    else if (keep) JK_FAST_LAUNCH(false, true, false);
    else JK_FAST_LAUNCH(false, false, false);
  }
#undef JK_FAST_LAUNCH
  HIP_CHECK_LAST();
  unsigned long long left = 0;
  HIP_TRY(read_back(&left, a.opt_state + 2, sizeof(left)));
  if (LAB_BITS(a.dbg) & 256) fprintf(stderr, "jk_probe_fast: %llu of %zu units left to the general kernel\n", left, nunits);
  if (left) GDF_TRY(run_probe(narrow, true, "jk_probe_write_general", (size_t)left, lds, a, probe_t, build_t));
  HIP_TRY(hipStreamSynchronize(stream0()));      // `todo` goes out of scope
  return GDF_SUCCESS;
}

// both passes of a plain join whose build keys repeat (jk_probe_multi): NARROW tuples, no carried columns, no FULL-join marks
static gdf_error run_multi_pass(bool write, size_t nunits, size_t lds, const ProbeArgs &a) {
  if (!nunits) return GDF_SUCCESS;
  const bool keep = a.keep_unmatched_probe != 0, p6 = a.p6_fb != 0;
  const char *name = write ? "jk_probe_write" : "jk_probe_count";
#define JK_MM_LAUNCH(W, KP, SIX)                                                                                                   \
  do {                                                                                                                            \
    HIP_TRY(hipFuncSetAttribute((const void *)jk_probe_multi<W, KP, SIX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    GDF_LAUNCH(name, (jk_probe_multi<W, KP, SIX>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), lds, stream0(), a);            \
  } while (0)
  if (write) {
    if (keep && p6) JK_MM_LAUNCH(true, true, true); else if (keep) JK_MM_LAUNCH(true, true, false);
    else if (p6) JK_MM_LAUNCH(true, false, true); else JK_MM_LAUNCH(true, false, false);
  } else {
    if (keep && p6) JK_MM_LAUNCH(false, true, true); else if (keep) JK_MM_LAUNCH(false, true, false);
    else if (p6) JK_MM_LAUNCH(false, false, true); else JK_MM_LAUNCH(false, false, false);
  }
#undef JK_MM_LAUNCH
  HIP_CHECK_LAST();
  return GDF_SUCCESS;
}

// COUNT pass over all units: the lean kernel for plain joins (jk_count_fast), the units it could not settle and every other
// case through the general kernel.  a.counts must be zeroed; a.opt_state must point at 3 zeroed counters.
static gdf_error run_count_pass(bool narrow, bool plain, size_t nunits, size_t lds, ProbeArgs a, uint32_t max_build,
                                const KeyTable &probe_t, const KeyTable &build_t) {
  if (!nunits) return GDF_SUCCESS;
  if (!plain || a.build_matched || lab::knob_on("GDF_JK_NO_FAST") || lab::knob_on("GDF_JK_NO_FAST_COUNT"))
    return run_probe(narrow, false, "jk_probe_count", nunits, lds, a, probe_t, build_t);
  DevBuf todo;
  RMM_TRY(todo.alloc(sizeof(uint32_t) * nunits));
  a.unit_todo = todo.as<uint32_t>();
  ProbeArgs fa = a;
  const bool pow2 = (double)max_build <= 0.42 * 2.0 * (double)a.nslots;      // as run_write_pass
  if (!pow2) fa.nslots = ((uint32_t)(max_build * 1.25) + 63) & ~63u;
  const size_t flds = probe_lds_bytes(narrow, a.cap, fa.nslots, false);
#define JK_COUNT_LAUNCH(...)                                                                                                     \
  do {                                                                                                                            \
    HIP_TRY(hipFuncSetAttribute((const void *)jk_count_fast<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)flds)); \
    GDF_LAUNCH("jk_probe_count", (jk_count_fast<__VA_ARGS__>), dim3((unsigned)nunits), dim3(JK_PROBE_THREADS), flds, stream0(), fa); \
  } while (0)
  const bool keep = a.keep_unmatched_probe != 0;
  if (narrow && a.p6_fb) {
    if (pow2 && keep) JK_COUNT_LAUNCH(true, true, true, true); else if (pow2) JK_COUNT_LAUNCH(true, false, true, true);
    else if (keep) JK_COUNT_LAUNCH(false, true, true, true); else JK_COUNT_LAUNCH(false, false, true, true);
  } else if (narrow) {
    if (pow2 && keep) JK_COUNT_LAUNCH(true, true, true); else if (pow2) JK_COUNT_LAUNCH(true, false, true);
    else if (keep) JK_COUNT_LAUNCH(false, true, true); else JK_COUNT_LAUNCH(false, false, true);
  } else if (a.p6_fb) {
    if (pow2 && keep) JK_COUNT_LAUNCH(true, true, false, true); else if (pow2) JK_COUNT_LAUNCH(true, false, false, true);
    else if (keep) JK_COUNT_LAUNCH(false, true, false, true); else JK_COUNT_LAUNCH(false, false, false, true);
  } else {
    if (pow2 && keep) JK_COUNT_LAUNCH(true, true, false); else if (pow2) JK_COUNT_LAUNCH(true, false, false);
    else if (keep) JK_COUNT_LAUNCH(false, true, false); else JK_COUNT_LAUNCH(false, false, false);
  }
#undef JK_COUNT_LAUNCH
  HIP_CHECK_LAST();
  unsigned long long left = 0;
  HIP_TRY(read_back(&left, a.opt_state + 2, sizeof(left)));
  if (left) GDF_TRY(run_probe(narrow, false, "jk_probe_count_general", (size_t)left, lds, a, probe_t, build_t));
  HIP_TRY(hipStreamSynchronize(stream0()));      // `todo` goes out of scope
  return GDF_SUCCESS;
}

// The join proper.  probe_t / build_t already reflect the INNER-join swap.
// On success *out_probe / *out_build own rmm allocations of *out_n int32 each.
// host-side stage clock (GDF_JK_DBG & 512): where the time between the kernels goes
struct StageClock {
  bool on;
  std::chrono::steady_clock::time_point t0, last;
  explicit StageClock(bool enable) : on(enable), t0(std::chrono::steady_clock::now()), last(t0) {}
  void mark(const char *what) {
    if (!on) return;
    (void)hipStreamSynchronize(stream0());
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  [join] %-28s %8.3f ms (total %8.3f)\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
            std::chrono::duration<double, std::milli>(now - t0).count());
    last = now;
  }
};

// Third level: every one of the 2^fb partitions of `sb` (either layout) is split 2^b3-way by the next hash bits into a
// new buffer in the speculative layout (partition f owns [f * cap3, (f + 1) * cap3), fill counters); jk_scatter2 does
// the work, with the level-2 partitions as its segments.  *ok = false leaves `sb` untouched (a partition outgrew its
// room: skewed keys -- the caller continues with the 2^fb partitions and the global-table path).
static gdf_error refine_side(const PartGeom &g, bool narrow, double dup, SideBufs *sb, bool *ok) {
  *ok = false;
  const uint32_t nseg = 1u << g.fb, nsub = 1u << g.b3;
  const uint64_t nfine3 = (uint64_t)nseg << g.b3;
  const int64_t n = sb->joinable;
  if (n == 0) return GDF_SUCCESS;
  const int threads = sb->pay[sb->final_buf].p ? 256 : level2_threads(narrow ? 1024 : 512);
  const int64_t TILE = (int64_t)threads * (sb->pay[sb->final_buf].p ? JK_PAY_ITEMS : JK_SC_ITEMS);
  const double mean = (double)n / (double)nfine3;
  const uint32_t cap3 = (uint32_t)(((uint64_t)(mean + 8.0 * std::sqrt(mean * (1.0 + dup)) + 64.0) + 7) / 8 * 8);
  const uint64_t size3 = nfine3 * cap3 + TILE;
  if (size3 >= 0x7fffffffULL) return GDF_SUCCESS;                      // tuple positions are 31-bit
  std::vector<uint32_t> seg(2 * (size_t)nseg), tile_prefix((size_t)nseg + 1), cur((size_t)nfine3);
  tile_prefix[0] = 0;
  for (uint32_t c = 0; c < nseg; ++c) {
    seg[c] = sb->fine_begin[c];
    seg[nseg + c] = sb->fine_begin[c] + sb->fine_cnt[c];
    tile_prefix[c + 1] = tile_prefix[c] + (uint32_t)((sb->fine_cnt[c] + TILE - 1) / TILE);
  }
  for (uint64_t f = 0; f < nfine3; ++f) cur[f] = (uint32_t)(f * cap3);
  const uint32_t ntiles = tile_prefix[nseg];
  DevBuf d_seg, d_tiles, cursor, flag, nw, nidx, npay;
  const bool carries = sb->pay[sb->final_buf].p != nullptr;
  RMM_TRY(d_seg.alloc(sizeof(uint32_t) * 2 * nseg));
  RMM_TRY(d_tiles.alloc(sizeof(uint32_t) * ((size_t)nseg + 1)));
  RMM_TRY(cursor.alloc(sizeof(uint32_t) * nfine3));
  RMM_TRY(flag.alloc(sizeof(uint32_t)));
  RMM_TRY(nw.alloc(sizeof(uint64_t) * size3));
  if (!narrow) RMM_TRY(nidx.alloc(sizeof(int32_t) * size3));
  if (carries) RMM_TRY(npay.alloc(sizeof(uint64_t) * sb->pay_words * size3));
  HIP_TRY(hipMemcpyAsync(d_seg.p, seg.data(), sizeof(uint32_t) * 2 * nseg, hipMemcpyHostToDevice, stream0()));
  HIP_TRY(hipMemcpyAsync(d_tiles.p, tile_prefix.data(), sizeof(uint32_t) * ((size_t)nseg + 1), hipMemcpyHostToDevice, stream0()));
  HIP_TRY(hipMemcpyAsync(cursor.p, cur.data(), sizeof(uint32_t) * nfine3, hipMemcpyHostToDevice, stream0()));
  HIP_TRY(hipMemsetAsync(flag.p, 0, sizeof(uint32_t), stream0()));
  PartGeom g3 = g;                      // jk_scatter2 sees 2^fb "coarse" segments and 2^b3 sub-bins of an (fb + b3)-bit id
  g3.b1 = g.fb;
  g3.b2 = g.b3;
  g3.fb = g.fb + g.b3;
  g3.cap1 = 0;
  g3.cap2 = cap3;
  g3.dump = (uint32_t)(nfine3 * cap3);
  g3.spec_flag = flag.as<uint32_t>();
  g3.xs = 0;
  Level2Map m{d_seg.as<uint32_t>(), d_seg.as<uint32_t>() + nseg, d_tiles.as<uint32_t>(), 0};
  if (ntiles) GDF_TRY(launch_scatter2(narrow, threads, ntiles, g3, m, sb->final(), cursor.as<uint32_t>(), Tuples{nw.as<uint64_t>(), nidx.as<int32_t>(), npay.as<uint64_t>()},
                                      false, false, sb->pay_words));
  uint32_t overflow = 0;
  HIP_TRY(read_back(cur.data(), cursor.p, sizeof(uint32_t) * nfine3));
  HIP_TRY(read_back(&overflow, flag.p, sizeof(uint32_t)));
  if (overflow) return GDF_SUCCESS;
  (void)nsub;
  // commit: the refined tuples replace the level-2 ones
  const int o = sb->final_buf ^ 1;
  sb->w[0].reset(); sb->w[1].reset(); sb->idx[0].reset(); sb->idx[1].reset(); sb->pay[0].reset(); sb->pay[1].reset();
  sb->w[o].p = nw.release();
  if (!narrow) sb->idx[o].p = nidx.release();
  if (carries) sb->pay[o].p = npay.release();
  sb->final_buf = o;
  sb->fine_off.clear();
  sb->fine_begin.resize(nfine3);
  sb->fine_cnt.resize(nfine3);
  for (uint64_t f = 0; f < nfine3; ++f) { sb->fine_begin[f] = (uint32_t)(f * cap3); sb->fine_cnt[f] = cur[f] - (uint32_t)(f * cap3); }
  sb->speculative = true;
  *ok = true;
  return GDF_SUCCESS;
}

// The build relation after partitioning: everything a probe pass needs besides the probe relation itself.  Made
// once per gdf_*_join call, or once per gdf_amd_join_build (include/gdf/gdf_amd_ext.h) and probed many times.
struct BuildSide {
  KeyPlan plan;              // key format BOTH relations are brought into (the build side decides narrow / kmin)
  PartGeom g;
  SideBufs B;
};

// Several integer key columns: pack (value - min) of every column, ranges from the build relation, when the fields fit
// 64 bits together.  The packed key is exact (no hashed 64-bit key + row comparison against the original columns: that
// path reads both relations' rows at random and ran 8x slower on an (int64, int32) key), and when the fields fit 32
// bits the join takes the NARROW tuples and the lean probe kernel of the single-column case.
static gdf_error plan_ranged(const KeyTable &build_t, KeyPlan *plan) {
  if (build_t.ncols < 2 || (plan->mode != KM_HASHED && plan->mode != KM_PACKED) || lab::path_on("GDF_JK_NO_RANGED")) return GDF_SUCCESS;
  for (int c = 0; c < build_t.ncols; ++c)
    if (build_t.col[c].kind == K_F32 || build_t.col[c].kind == K_F64) return GDF_SUCCESS;
  std::vector<long long> h(2 * build_t.ncols);
  GDF_TRY(key_ranges(build_t, h.data()));
  KeyPlan p = *plan;
  int total = 0;
  for (int c = 0; c < build_t.ncols; ++c) {
    long long lo = h[2 * c], hi = h[2 * c + 1];
    if (lo > hi) lo = hi = 0;                                 // no valid element: nothing will join anyway
    const uint64_t span = (uint64_t)hi - (uint64_t)lo;
    int bits = 0;
    while (bits < 64 && (span >> bits) != 0) ++bits;
    p.bits[c] = bits;
    p.bias[c] = lo;
    p.shift[c] = total;
    total += bits;
  }
  if (total > 64) return GDF_SUCCESS;                         // keep the hashed / byte-packed plan
  p.mode = KM_PACKED;
  p.ranged = 1;
  p.verify = 0;
  p.narrow = total <= 32 ? 1 : 0;
  p.klimit = p.narrow ? 0xffffffffULL : ~0ULL;
  p.kmin = 0;
  *plan = p;
  return GDF_SUCCESS;
}

// bpay / bmode (PayCarry::bmode): the build relation's payload word travels with its tuples when the side ends up on NARROW
// tuples from one FAST, unmasked key column (partition_side decides; B.pay stays empty otherwise)
// keep_running: the call returns with the side's last launches still queued (partition_side); the caller settles bs->B
static gdf_error prepare_build(const KeyTable &build_t, BuildSide *bs, bool no_level3 = false, const PaySrc *bpay = nullptr, int bmode = 0,
                               bool keep_running = false) {
  if (bmode == 4) bmode = 1;               // (two 8-byte build columns: the FIRST travels with the tuples, the second is staged by row -- jk_probe_bp<.., BW2>)
  bs->plan = plan_keys(build_t);           // a function of the key dtypes only: the probe relation has the same ones
  GDF_TRY(plan_ranged(build_t, &bs->plan));
  bs->g = choose_geometry(build_t.nrows);
  const bool range_candidate = !bs->plan.narrow && bs->plan.mode == KM_RAW_INT && build_t.col[0].width == 8;
  const bool stays_two_level = !(bs->g.b3 > 0 && !no_level3);          // (a third level rewrites the index on the host)
  GDF_TRY(partition_side(build_t, bs->plan, bs->g, &bs->B, range_candidate, bpay, bmode, stays_two_level, false,
                         keep_running && stays_two_level));   // may switch plan to the narrow format
  if (bs->g.b3 > 0 && !no_level3) {
    bool ok = false;
    GDF_TRY(refine_side(bs->g, bs->plan.narrow != 0, 0.0, &bs->B, &ok));
    uint32_t largest = 0;
    if (ok) for (uint32_t c : bs->B.fine_cnt) largest = std::max(largest, c);
    // a refined partition that still does not fit LDS (skew) needs the global-table path, which wants the exact layout:
    // such a relation is partitioned again without the third level
    if (ok && largest > (uint32_t)JK_MAX_BUILD) {
      bs->B.w[0].reset(); bs->B.w[1].reset(); bs->B.idx[0].reset(); bs->B.idx[1].reset(); bs->B.pay[0].reset(); bs->B.pay[1].reset();
      bs->g.b3 = 0;
      GDF_TRY(partition_side(build_t, bs->plan, bs->g, &bs->B, false, bpay, bmode));
    } else if (!ok) {
      bs->g.b3 = 0;
    }
  } else {
    bs->g.b3 = 0;
  }
  // the partition index once more on the device: jk_make_units builds the work units of a deferred probe side from it
  const size_t nparts = bs->B.fine_cnt.size();
  if (nparts && !(stays_two_level && bs->B.d_cnt.p && bs->B.d_begin.p)) {
    RMM_TRY(bs->B.d_begin.alloc(sizeof(uint32_t) * nparts));
    RMM_TRY(bs->B.d_cnt.alloc(sizeof(uint32_t) * nparts));
    HIP_TRY(hipMemcpyAsync(bs->B.d_begin.p, bs->B.fine_begin.data(), sizeof(uint32_t) * nparts, hipMemcpyHostToDevice, stream0()));
    HIP_TRY(hipMemcpyAsync(bs->B.d_cnt.p, bs->B.fine_cnt.data(), sizeof(uint32_t) * nparts, hipMemcpyHostToDevice, stream0()));
  }
  return GDF_SUCCESS;
}

static gdf_error probe_partitioned(const KeyTable &probe_t, const KeyTable &build_t, const BuildSide &bs, const KeyPlan &plan, SideBufs &P, JoinKind kind,
                                   int32_t **out_probe, int32_t **out_build, int64_t *out_n, StageClock &clk, PayCarry *pc = nullptr);

// The probe side's skew sample (jk_sample_skew) needs nothing from the build pass: hash_join_core launches it BEFORE the build side is
// partitioned, and its 4-byte answer is read when the probe side's turn comes -- no kernel + read-back round trip between the two sides.
struct SkewProbe {
  DevBuf sh;
  size_t words = 0;
  ReadTicket answer;          // the copy of the 4-byte answer is queued right behind the sample: whoever asks later waits for nothing
};
static bool skew_probe_wanted(const KeyTable &probe_t, const KeyPlan &plan, const PartGeom &g) {
  return g.fb >= 10 && probe_t.nrows >= ((int64_t)1 << 24) && fast_key_width(probe_t, plan) && plan.mode == KM_RAW_INT &&
         !lab::knob_on("GDF_JK_NO_SKEW_SAMPLE");
}
static gdf_error skew_probe_launch(const KeyTable &probe_t, const KeyPlan &plan, const PartGeom &g, SkewProbe *sp) {
  sp->words = ((size_t)1 << g.fb) + 1;
  RMM_TRY(sp->sh.alloc(sizeof(uint32_t) * sp->words));
  HIP_TRY(hipMemsetAsync(sp->sh.p, 0, sizeof(uint32_t) * sp->words, stream0()));
  if (fast_key_width(probe_t, plan) == 8)
    hipLaunchKernelGGL(jk_sample_skew<8>, dim3(JK_SKEW_SAMPLES / 256), dim3(256), 0, stream0(), probe_t.col[0].data, probe_t.nrows, g.fb,
                       sp->sh.as<uint32_t>(), sp->sh.as<uint32_t>() + (sp->words - 1));
  else
    hipLaunchKernelGGL(jk_sample_skew<4>, dim3(JK_SKEW_SAMPLES / 256), dim3(256), 0, stream0(), probe_t.col[0].data, probe_t.nrows, g.fb,
                       sp->sh.as<uint32_t>(), sp->sh.as<uint32_t>() + (sp->words - 1));
  HIP_CHECK_LAST();
  HIP_TRY(read_back_begin(&sp->answer, sp->sh.as<uint32_t>() + (sp->words - 1), sizeof(uint32_t), 1));
  return GDF_SUCCESS;
}

static gdf_error probe_prepared(const KeyTable &probe_t, const KeyTable &build_t, const BuildSide &bs, JoinKind kind,
                                int32_t **out_probe, int32_t **out_build, int64_t *out_n, StageClock &clk, PayCarry *pc = nullptr,
                                SkewProbe *early_skew = nullptr) {
  // which probe rows travel (KeyPlan::kwindow): decided per probe call, on this call's COPY of the plan -- a prepared build side
  // serves INNER and LEFT probes alike, possibly from several threads at once
  KeyPlan plan = bs.plan;
  const PartGeom &g = bs.g;
  const SideBufs &B = bs.B;
  place_budget_begin();
  ProfTag probe_tag("@probe");          // (profiling only: the probe side's launches are reported apart from the build side's)
  if (plan.kwindow) plan.klimit = kind == JOIN_INNER ? plan.kspan : plan.kwindow;

  SideBufs P;
  // The probe side is the big one (C3: 10x the build side): it is partitioned WITHOUT a histogram pass
  // when the build partitions all fit LDS (the global-table path wants contiguous partition runs).
  uint32_t largest_build = 0;
  for (uint32_t c : B.fine_cnt) largest_build = std::max(largest_build, c);
  const int64_t spec_min = lab::path_int("GDF_JK_SPEC_MIN", (int64_t)1 << 22);   // test switch
  bool spec_ok = false;
  // skewed probe keys would overflow the speculative layout: ask a sample first (one small kernel and a 4-byte read-back)
  bool skew = false;
  const int probe_fast = fast_key_width(probe_t, plan);
  if (skew_probe_wanted(probe_t, plan, g)) {
    SkewProbe local;
    SkewProbe *sp = (early_skew && early_skew->sh.p && early_skew->words == ((size_t)1 << g.fb) + 1) ? early_skew : &local;
    if (sp == &local) GDF_TRY(skew_probe_launch(probe_t, plan, g, sp));
    uint32_t fullest = 0;
    HIP_TRY(read_back_end(&sp->answer, &fullest));
    // expected samples per bin: 2^16 / 2^fb (2 at fb = 15); a Poisson(2) bin reaches 16 with probability ~1e-10
    const double expect = (double)JK_SKEW_SAMPLES / (double)((uint64_t)1 << g.fb);
    skew = (double)fullest > 8.0 * expect + 12.0;
  }
  // the main path -- two-level speculative probe side, every build partition in LDS, no FULL-join marks -- keeps its
  // bookkeeping on the device (see jk_make_units)
  const bool defer = g.b2 > 0 && g.b3 == 0 && kind != JOIN_FULL && B.d_cnt.p != nullptr && !lab::knob_on("GDF_JK_NO_DEFER");
  // the probe relation's payload column(s) ride along (PayCarry) when the join runs on NARROW tuples with exact keys from
  // one FAST, unmasked key column; everything else leaves pc->carried false and the caller gathers as before
  PaySrc pay_src{};
  const bool carry_any = pc && plan.narrow && !plan.verify && kind != JOIN_FULL && !lab::path_on("GDF_JK_NO_CARRY");
  const bool carry = carry_any && pc->mode && probe_fast && !probe_t.col[0].valid;
  // (what probe_partitioned gets: the carry request when the probe payload travels or the build payload did, see prepare_build)
  PayCarry *pc_eff = (carry || (carry_any && kind == JOIN_INNER && pc->bmode && B.pay[B.final_buf].p)) ? pc : nullptr;
  if (carry) { pay_src.col[0] = pc->src[0]; pay_src.col[1] = pc->src[1]; }
  const PaySrc *pay = carry ? &pay_src : nullptr;
  const int pmode = carry ? pc->mode : 0;
  // six-byte level-2 tuples (p6_store): the main path, 2^15 fine partitions (17 hash bits left), nothing carried, and stored keys
  // on which hash_a is a bijection -- their raw values raw = key + kmin must not straddle a 2^32 boundary
  const bool bijective = plan.kmin == 0 || (plan.kmin & 0xffffffffULL) + plan.kspan < (1ULL << 32);
  const bool p6_ok = g.fb == JK_MAX_FB && g.b2 > 0 && g.b3 == 0 && kind != JOIN_FULL && g.world <= 1 && plan.narrow && !plan.verify && pc_eff == nullptr &&
                     bijective && !lab::path_on("GDF_JK_NO_P6");
  // TEN-byte tuples for WIDE keys (p10_key): the same conditions on one exact 8-byte key column (integer, or float by canonical bits) --
  // the partition hash plus the raw key's high word identify the key, no range is asked of it.  GDF_JK_NO_W10: the 12-byte tuples
  const bool w10_ok = g.fb == JK_MAX_FB && g.b2 > 0 && g.b3 == 0 && kind != JOIN_FULL && g.world <= 1 && !plan.narrow && !plan.verify && pc_eff == nullptr &&
                      (plan.mode == KM_RAW_INT || plan.mode == KM_RAW_FLOAT) && probe_fast == 8 && fast_key_width(build_t, plan) == 8 &&
                      !lab::path_on("GDF_JK_NO_P6") && !lab::path_on("GDF_JK_NO_W10");
  const bool want_p6 = defer && (p6_ok || w10_ok);
  // (the exact layout's probe side too -- skewed probe keys -- as long as every build partition is an LDS unit: the global-table path reads 8-byte tuples)
  const bool want_p6_exact = (p6_ok || w10_ok) && largest_build <= (uint32_t)JK_MAX_BUILD && !lab::path_on("GDF_JK_NO_P6_EXACT");
  if (!skew && g.fb > 0 && probe_t.nrows >= spec_min && largest_build <= (uint32_t)JK_MAX_BUILD && !lab::path_on("GDF_JK_NO_SPEC"))
    GDF_TRY(partition_side_spec(probe_t, plan, g, std::max(1.0, (double)probe_t.nrows / std::max<uint32_t>(B.joinable, 1)), &P, &spec_ok,
                                nullptr, defer, pay, pmode, want_p6));
  // SKEWED probe keys keep the histogram-free layout too (round 4): a second, larger sample of the probe column
  // (2^22 rows, binned by fine partition) sizes every level-1 region and every fine partition individually (SkewCaps) -- the exact
  // layout's histogram pass over the probe relation (1.5 of a Zipf join's 13.7 ms) is not needed, and the side stays on the
  // deferred path with its six-byte tuples.  A partition that outgrows its room anyway raises the usual flag: exact layout then.
  if (skew && defer && !pay && probe_fast && g.fb > 0 && probe_t.nrows >= spec_min && largest_build <= (uint32_t)JK_MAX_BUILD &&
      !lab::path_on("GDF_JK_NO_SPEC") && !lab::path_on("GDF_JK_NO_SKEW_CAPS")) {
    const uint32_t nfine_p = 1u << g.fb;
    DevBuf d_counts;
    RMM_TRY(d_counts.alloc(sizeof(uint32_t) * ((size_t)nfine_p + 1)));
    HIP_TRY(hipMemsetAsync(d_counts.p, 0, sizeof(uint32_t) * ((size_t)nfine_p + 1), stream0()));
    const size_t lds = sizeof(uint32_t) * nfine_p;
    if (probe_fast == 8) {
      HIP_TRY(hipFuncSetAttribute((const void *)jk_sample_caps<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GDF_LAUNCH("jk_sample_caps", jk_sample_caps<8>, dim3(256), dim3(1024), lds, stream0(), probe_t.col[0].data, probe_t.nrows, g.fb, d_counts.as<uint32_t>());
    } else {
      HIP_TRY(hipFuncSetAttribute((const void *)jk_sample_caps<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GDF_LAUNCH("jk_sample_caps", jk_sample_caps<4>, dim3(256), dim3(1024), lds, stream0(), probe_t.col[0].data, probe_t.nrows, g.fb, d_counts.as<uint32_t>());
    }
    HIP_CHECK_LAST();
    SkewCaps caps;
    caps.fine.resize((size_t)nfine_p + 1);
    HIP_TRY(read_back(caps.fine.data(), d_counts.p, sizeof(uint32_t) * caps.fine.size()));
    const uint32_t sampled = caps.fine[nfine_p];
    caps.fine.resize(nfine_p);
    if (sampled) {
      caps.rows_per_sample = (double)probe_t.nrows / (double)sampled;
      GDF_TRY(partition_side_spec(probe_t, plan, g, 1.0, &P, &spec_ok, nullptr, defer, pay, pmode, want_p6, &caps));
    }
  }
  if (!spec_ok) {
    P.w[0].reset(); P.w[1].reset(); P.idx[0].reset(); P.idx[1].reset(); P.pay[0].reset(); P.pay[1].reset();
    KeyPlan probe_plan = plan;             // partition_side only rewrites the plan when asked to decide the format
    GDF_TRY(partition_side(probe_t, probe_plan, g, &P, false, pay, pmode, false, want_p6_exact));
  }
  const bool narrow = plan.narrow != 0;
  if (g.b3 > 0) {
    bool ok = false;
    GDF_TRY(refine_side(g, narrow, std::max(1.0, (double)probe_t.nrows / std::max<uint32_t>(B.joinable, 1)), &P, &ok));
    if (!ok) return GDF_AMD_RETRY_WITHOUT_LEVEL3;     // skewed probe keys: the caller repeats with a 2^fb-partition build side
  }
  clk.mark("partition probe side");
  gdf_error e = probe_partitioned(probe_t, build_t, bs, plan, P, kind, out_probe, out_build, out_n, clk, pc_eff);
  if (e != GDF_AMD_RETRY_EXACT_PROBE) return e;
  // a deferred speculative probe side turned out to have overflowed (skewed keys): the exact layout, host bookkeeping
  SideBufs Q;
  KeyPlan probe_plan = plan;
  GDF_TRY(partition_side(probe_t, probe_plan, g, &Q, false, pay, pmode, false, want_p6_exact));
  return probe_partitioned(probe_t, build_t, bs, plan, Q, kind, out_probe, out_build, out_n, clk, pc_eff);
}

// the part of a join after both relations are partitioned: work units, optimistic single pass or count + write, tails
static gdf_error probe_partitioned(const KeyTable &probe_t, const KeyTable &build_t, const BuildSide &bs, const KeyPlan &plan, SideBufs &P, JoinKind kind,
                                   int32_t **out_probe, int32_t **out_build, int64_t *out_n, StageClock &clk, PayCarry *pc) {
  const PartGeom &g = bs.g;
  const SideBufs &B = bs.B;
  const uint32_t nfine = 1u << (g.fb + g.b3);
  const bool keep_probe = kind != JOIN_INNER;
  const bool narrow = plan.narrow != 0;

  // ---- work units ----
  // DEFERRED probe side: units, output offsets and the sample are made on the device from the fill counters and ONE state
  // block comes back; otherwise the host walks the partition index it already holds.
  const bool deferred = P.deferred;
  std::vector<Unit> units;
  struct Run { uint32_t f0, f1; };    // [f0, f1): consecutive fine partitions that need the global-table path
  std::vector<Run> oversize;
  uint32_t max_build = 0;
  size_t nunits = 0;
  DevBuf d_units, d_counts, d_matched, d_tail, d_off, d_bk;
  unsigned long long bk[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // [0..3] jk_make_units state, [4..7] sample state (pairs, tuples, -, linear units)
  constexpr size_t NSAMPLE = 64;
  if (deferred) {
    for (uint32_t f = 0; f < nfine; ++f) max_build = std::max(max_build, B.fine_cnt[f]);      // <= JK_MAX_BUILD: the caller checked
    const size_t unit_bound = (size_t)nfine + (size_t)(probe_t.nrows / JK_PROBE_CHUNK) + 2;
    RMM_TRY(d_units.alloc(sizeof(Unit) * unit_bound));
    RMM_TRY(d_off.alloc(sizeof(uint64_t) * (unit_bound + 1)));
    if (P.zero) d_bk.borrow(P.zero);
    else {
      RMM_TRY(d_bk.alloc(sizeof(bk)));
      HIP_TRY(hipMemsetAsync(d_bk.p, 0, sizeof(bk), stream0()));
    }
    hipLaunchKernelGGL(jk_make_units, dim3((nfine + 255) / 256), dim3(256), 0, stream0(), nfine, P.cap2, (const uint32_t *)P.d_cursor.as<uint32_t>(),
                       (const uint32_t *)(P.d_level1.as<uint32_t>() + P.nseg), (const uint32_t *)B.d_begin.as<uint32_t>(),
                       (const uint32_t *)B.d_cnt.as<uint32_t>(), keep_probe ? 1 : 0, d_units.as<Unit>(), d_off.as<uint64_t>(),
                       d_bk.as<unsigned long long>(), P.d_fstart, P.d_fcap);
    HIP_CHECK_LAST();
  } else {
    units.reserve((size_t)nfine + (size_t)(probe_t.nrows / JK_PROBE_CHUNK) + 1);     // the GPU idles while this list is made
    for (uint32_t f = 0; f < nfine; ++f) {
      const uint32_t bn = B.fine_cnt[f];
      const uint32_t pn = P.fine_cnt[f];
      if (pn == 0) continue;
      if (bn == 0 && !keep_probe) continue;
      if (bn > (uint32_t)JK_MAX_BUILD) {
        if (!oversize.empty() && oversize.back().f1 == f) oversize.back().f1 = f + 1;   // one table per RUN, not per partition
        else oversize.push_back(Run{f, f + 1});
        continue;
      }
      max_build = std::max(max_build, bn);
      for (uint32_t off = 0; off < pn; off += JK_PROBE_CHUNK)
        units.push_back(Unit{B.fine_begin[f], bn, P.fine_begin[f] + off, std::min(JK_PROBE_CHUNK, pn - off)});
    }
    nunits = units.size();
    RMM_TRY(d_units.alloc(sizeof(Unit) * (nunits ? nunits : 1)));
    if (nunits) HIP_TRY(hipMemcpyAsync(d_units.p, units.data(), sizeof(Unit) * nunits, hipMemcpyHostToDevice, stream0()));
  }
  // LDS geometry shared by all units: room for the largest in-LDS build partition, H = slots per cuckoo table
  const uint32_t cap_lds = (std::max<uint32_t>(max_build, 64) + 63) & ~63u;
  uint32_t H_lds = 64;
  while (H_lds < max_build) H_lds <<= 1;

  if (deferred && P.zero) d_tail.borrow(P.zero + 8);
  else {
    RMM_TRY(d_tail.alloc(sizeof(unsigned long long) * 4));
    HIP_TRY(hipMemsetAsync(d_tail.p, 0, sizeof(unsigned long long) * 4, stream0()));
  }
  if (kind == JOIN_FULL) {
    RMM_TRY(d_matched.alloc((size_t)(build_t.nrows ? build_t.nrows : 1)));
    HIP_TRY(hipMemsetAsync(d_matched.p, 0, (size_t)(build_t.nrows ? build_t.nrows : 1), stream0()));
  }

  // carried columns (PayCarry): probe payload (<= 2), key, build payload (<= 2) -- sized when the pair count is known
  struct Carried { int width; DevBuf buf; void **commit; };
  Carried cc[5];
  int ncc = 0, probe_cc = 0, key_cc = -1, build_cc = -1;
  // (the BP image is 16 bytes per build tuple + the tables: with very full partitions it does not fit one CU's 160 KiB -- the
  // build payload is then gathered like any other column)
  const bool bp_pow2 = (double)max_build <= 0.42 * 2.0 * (double)H_lds;
  const uint32_t bp_slots = bp_pow2 ? H_lds : (((uint32_t)(max_build * 1.25) + 63) & ~63u);
  const bool build_pay = pc && kind == JOIN_INNER && pc->bmode && B.pay[B.final_buf].p != nullptr && narrow && !plan.verify &&
                         oversize.empty() && probe_bp_lds_bytes(cap_lds, bp_slots, pc->bmode == 4) <= (size_t)160 * 1024;
  const bool probe_pay = pc && pc->mode && P.pay[P.final_buf].p != nullptr;
  if (pc) {
    if (probe_pay) for (int c = 0; c < pc->ncols(); ++c) { cc[ncc].width = pc->elem_bytes(c); cc[ncc++].commit = &pc->dst[c]; }
    probe_cc = ncc;
    if (kind == JOIN_INNER && pc->key_width && (probe_pay || build_pay)) { key_cc = ncc; cc[ncc].width = pc->key_width; cc[ncc++].commit = &pc->key_dst; }
    if (build_pay) { build_cc = ncc; for (int c = 0; c < pc->bncols(); ++c) { cc[ncc].width = pc->belem_bytes(); cc[ncc++].commit = &pc->bdst[c]; } }
  }
  auto alloc_pay = [&](uint64_t total) -> gdf_error {
    for (int c = 0; c < ncc; ++c) RMM_TRY(cc[c].buf.alloc((size_t)cc[c].width * (size_t)(total ? total : 1)));
    return GDF_SUCCESS;
  };
  auto set_pay = [&](ProbeArgs &x) {
    if (!pc) return;
    if (probe_pay) {
      x.pay_mode = pc->mode;
      for (int c = 0; c < pc->ncols(); ++c) { x.pay_src[c] = pc->src[c]; x.pay_out[c] = cc[c].buf.p; }
    }
    if (key_cc >= 0) { x.key_width = pc->key_width; x.key_src = pc->key_src; x.key_out = cc[key_cc].buf.p; }
    if (build_cc >= 0) {
      x.bpay_mode = pc->bmode;
      for (int c = 0; c < pc->bncols(); ++c) { x.bpay_src[c] = pc->bsrc[c]; x.bpay_out[c] = cc[build_cc + c].buf.p; }
    }
  };
  auto pay_move_at = [&](uint64_t first) -> PayMove {        // the probe payload columns from output position `first` on, filled from the source by row
    PayMove pm{};                                            // (LEFT-join tails; key and build payload are only carried by INNER joins, which have none)
    for (int c = 0; c < probe_cc; ++c) {
      pm.width[c] = cc[c].width;
      pm.src[c] = pc->src[c];
      pm.dst[c] = cc[c].buf.as<char>() + first * (uint64_t)cc[c].width;
    }
    pm.ncols = probe_cc;
    return pm;
  };
  auto pay_move_all = [&](DevBuf (*dense)[5]) -> PayMove {   // every carried column: from its buffer into dense[] (compaction) or in place (null)
    PayMove pm{};
    for (int c = 0; c < ncc; ++c) { pm.width[c] = cc[c].width; pm.src[c] = cc[c].buf.p; pm.dst[c] = dense ? (*dense)[c].p : cc[c].buf.p; }
    pm.ncols = ncc;
    return pm;
  };
  auto commit_pay = [&]() {
    if (!pc) return;
    for (int c = 0; c < ncc; ++c) *cc[c].commit = cc[c].buf.release();
    pc->carried = probe_pay;
    pc->build_carried = build_cc >= 0;
  };
  auto drop_pay = [&]() { for (int c = 0; c < ncc; ++c) cc[c].buf.reset(); };

  ProbeArgs a{};
  a.build = B.final();
  a.probe = P.final();
  a.units = d_units.as<Unit>();
  a.nslots = H_lds;
  a.cap = cap_lds;
  a.keep_unmatched_probe = keep_probe ? 1 : 0;
  a.verify = plan.verify;
  a.build_matched = d_matched.as<uint8_t>();
  a.dbg = (int)lab::knob_int("GDF_JK_DBG", 0);
  a.kbias = plan.kmin;
  if (P.p6) {                       // six-byte probe tuples: the kernels hash and compare hash remainders (ProbeArgs::p6_fb)
    a.p6_fb = g.fb;
    a.p6_world = g.world;
    a.p6_kbias = plan.kmin;
    a.kbias = 0;
  }
  const size_t probe_lds = probe_lds_bytes(narrow, cap_lds, H_lds);
  const bool plain = kind != JOIN_FULL && !plan.verify;       // INNER and LEFT with exact keys: see jk_probe_fast

  // set by the sample below: most sampled units hold repeated build keys (their cuckoo build fell back to linear probing).
  // The lean write kernel would give up on nearly every unit after four cuckoo attempts (5.7 of 31 ms on a join whose
  // build keys all occur four times, profiles/r2_b_bench_shapes.jsonl) -- such joins go to the general kernel directly.
  bool dup_heavy = false;
  clk.mark("units + argument setup");
  // ---- optimistic single pass ----
  // A foreign-key -> primary-key join whose every probe row finds its key emits exactly one pair per
  // probe tuple.  Then unit u's output is its probe_count slots at the prefix sum of the probe counts
  // and no count pass is needed.  Tried when a count over a sample of units shows one pair per tuple;
  // any surprise during the pass (a unit short of its slots, or needing more) falls back to count + write.
  bool try_optimistic = false;
  double sample_hit = 1.0;                 // pairs per probe tuple over the sampled units
  uint64_t cap_pairs = 0;
  if (deferred) {
    // the sample runs over the device-built units (workgroup b takes unit b * units / 64) and leaves its sums next to
    // jk_make_units' state: one read-back for everything
    ProbeArgs sa = a;
    sa.build_matched = nullptr;
    sa.sample_n = (uint32_t)NSAMPLE;
    sa.nunits_dev = d_bk.as<unsigned long long>();
    sa.opt_state = d_bk.as<unsigned long long>() + 4;
    if (!(LAB_BITS(a.dbg) & 16)) GDF_TRY(run_probe(narrow, false, "jk_probe_sample", NSAMPLE, probe_lds, sa, probe_t, build_t));
    HIP_TRY(read_back(bk, d_bk.p, sizeof(bk)));
    P.w[0].reset();                        // level-1 tuples and the segment map: everything that read them has run
    P.idx[0].reset();
    P.d_map.reset();
#ifdef GDF_AMD_LAB
    if (lab::knob_on("GDF_JK_TRACE")) fprintf(stderr, "deferred probe side: overflow flags %llu (1 = level 2, 2 = level 1), units %llu, tuples %llu\n", bk[3], bk[0], bk[2]);
#endif
    if (bk[3]) return GDF_AMD_RETRY_EXACT_PROBE;
    nunits = (size_t)bk[0];
    cap_pairs = bk[1];
    P.joinable = (uint32_t)bk[2];
    dup_heavy = bk[7] * 4 >= NSAMPLE;
    try_optimistic = nunits && !(LAB_BITS(a.dbg) & 16) && bk[4] == bk[5];
    sample_hit = bk[5] ? (double)bk[4] / (double)bk[5] : 1.0;
    clk.mark("sample count");
  } else if (nunits && oversize.empty() && kind != JOIN_FULL && !(LAB_BITS(a.dbg) & 16)) {
    const size_t nsample = std::min<size_t>(nunits, NSAMPLE);
    std::vector<Unit> sample(nsample);
    uint64_t sample_tuples = 0;
    for (size_t i = 0; i < nsample; ++i) { sample[i] = units[i * nunits / nsample]; sample_tuples += sample[i].probe_count; }
    DevBuf d_sample, d_scount, d_sstate;
    RMM_TRY(d_sample.alloc(sizeof(Unit) * nsample));
    RMM_TRY(d_scount.alloc(sizeof(uint64_t) * nsample));
    HIP_TRY(hipMemcpyAsync(d_sample.p, sample.data(), sizeof(Unit) * nsample, hipMemcpyHostToDevice, stream0()));
    // the slots of the optimistic pass are laid out and uploaded now, under the sample kernel, not after its read-back
    std::vector<uint64_t> off(nunits + 1);
    off[0] = 0;
    for (size_t i = 0; i < nunits; ++i) off[i + 1] = off[i] + units[i].probe_count;
    RMM_TRY(d_off.alloc(sizeof(uint64_t) * (nunits + 1)));
    ProbeArgs sa = a;
    sa.units = d_sample.as<Unit>();
    sa.counts = d_scount.as<uint64_t>();
    sa.build_matched = nullptr;
    RMM_TRY(d_sstate.alloc(sizeof(unsigned long long) * 4));
    HIP_TRY(hipMemsetAsync(d_sstate.p, 0, sizeof(unsigned long long) * 4, stream0()));
    sa.opt_state = d_sstate.as<unsigned long long>();
    GDF_TRY(run_probe(narrow, false, "jk_probe_sample", nsample, probe_lds, sa, probe_t, build_t));
    HIP_TRY(hipMemcpyAsync(d_off.p, off.data(), sizeof(uint64_t) * (nunits + 1), hipMemcpyHostToDevice, stream0()));
    std::vector<uint64_t> scount(nsample);
    HIP_TRY(read_back(scount.data(), d_scount.p, sizeof(uint64_t) * nsample));
    unsigned long long sstate[4] = {0, 0, 0, 0};
    HIP_TRY(read_back(sstate, d_sstate.p, sizeof(sstate)));
    dup_heavy = sstate[3] * 4 >= nsample;
    clk.mark("sample count");
    uint64_t sample_pairs = 0;
    for (uint64_t c : scount) sample_pairs += c;
    try_optimistic = sample_pairs == sample_tuples;
    sample_hit = sample_tuples ? (double)sample_pairs / (double)sample_tuples : 1.0;
    cap_pairs = off[nunits];
  }
  // SPARSE optimistic pass: the sample says fewer than one pair per probe tuple (some probe rows miss) and no repeated build
  // keys.  The single write pass still works -- a unit's pairs fit the slots of its probe tuples -- it just leaves a hole at the
  // end of every unit's range.  Closing the holes is cheaper than the count pass it replaces (8 B per probe TUPLE plus a second
  // build of every LDS table) at EVERY hit rate: below a third by packing all pairs into exact-size columns (jk_compact_units,
  // 16 B per pair), from a third on by moving only the pairs behind the final size into the holes in front of it (jk_fill_holes:
  // h (1 - h) of the slots; the columns keep their allocation of one slot per probe tuple).  A unit that runs out of slots
  // (a build key present more than once after all) sends the call to count + write.
  const bool try_sparse = !try_optimistic && d_off.p && nunits && oversize.empty() && kind == JOIN_INNER && !dup_heavy && !(LAB_BITS(a.dbg) & 16) &&
                          sample_hit <= lab::knob_float("GDF_JK_SPARSE_MAX", 1.0) && !lab::path_on("GDF_JK_NO_SPARSE_OPT");
  if (try_optimistic || try_sparse) {
    DevBuf d_state, d_upairs;
    if (deferred && P.zero) d_state.borrow(P.zero + 12);
    else {
      RMM_TRY(d_state.alloc(sizeof(unsigned long long) * 4));
      HIP_TRY(hipMemsetAsync(d_state.p, 0, sizeof(unsigned long long) * 4, stream0()));
    }
    const uint64_t probe_tail = keep_probe ? (uint64_t)probe_t.nrows - P.joinable : 0;
    const uint64_t total = cap_pairs + probe_tail;
    if (total >= (uint64_t)INT_MAX) return GDF_COLUMN_SIZE_TOO_BIG;
    // the two index columns of a large dense join are PLACED blocks too (the probe kernel's 2 x 4 GB of writes have their fast and
    // slow placements like the regroup passes'): tournament below; the caller gets the winners and frees them through rmmFree as ever
    const bool place_out = try_optimistic && nunits >= 64 && !lab::knob_on("GDF_JK_NO_CALIBRATE");
    const size_t out_bytes = sizeof(int32_t) * (size_t)(total ? total : 1);
    DevBuf op, ob;
    if (place_out) {
      const int d = place_draws_now(JK_PLACE_DRAWS_OUT);
      RMM_TRY(op.alloc_placed(JK_ROLE_OUT_PROBE, out_bytes, d));
      RMM_TRY(ob.alloc_placed(JK_ROLE_OUT_BUILD, out_bytes, d));
    } else {
      RMM_TRY(op.alloc(out_bytes));
      RMM_TRY(ob.alloc(out_bytes));
    }
    GDF_TRY(alloc_pay(total));

    ProbeArgs oa = a;
    set_pay(oa);
    oa.counts = d_off.as<uint64_t>();
    oa.out_probe = op.as<int32_t>();
    oa.out_build = ob.as<int32_t>();
    oa.build_matched = nullptr;
    oa.optimistic = 1;
    oa.opt_state = d_state.as<unsigned long long>();
    if (try_sparse) {
      RMM_TRY(d_upairs.alloc(sizeof(uint32_t) * (nunits + 1)));
      HIP_TRY(hipMemsetAsync(d_upairs.p, 0, sizeof(uint32_t) * (nunits + 1), stream0()));
      oa.unit_pairs = d_upairs.as<uint32_t>();
    }
    // PLACEMENT TOURNAMENT of the output columns (both roles see the same times, so they keep and drop their candidates together):
    // a calibration run is the write pass over the first quarter of the units; the pass's state words are cleared behind it
    for (int round = 0; place_out && round <= JK_PLACE_DRAWS_OUT && (op.measure || ob.measure); ++round) {
      PlaceRound charge;
      op.clock_begin(stream0());
      ob.clock_begin(stream0());
      GDF_TRY(run_write_pass(narrow, plain && !dup_heavy, nunits / 4, probe_lds, oa, max_build, probe_t, build_t));
      op.clock_end(stream0());
      ob.clock_end(stream0());
      HIP_TRY(hipMemsetAsync(d_state.p, 0, sizeof(unsigned long long) * 4, stream0()));
      if (try_sparse) HIP_TRY(hipMemsetAsync(d_upairs.p, 0, sizeof(uint32_t) * (nunits + 1), stream0()));
      const int d = place_draws_now(JK_PLACE_DRAWS_OUT);
      RMM_TRY(op.alloc_placed(JK_ROLE_OUT_PROBE, out_bytes, d));
      RMM_TRY(ob.alloc_placed(JK_ROLE_OUT_BUILD, out_bytes, d));
      oa.out_probe = op.as<int32_t>();
      oa.out_build = ob.as<int32_t>();
    }
    clk.mark("output allocation");
    P.w[P.final_buf].clock_begin(stream0());      // (a placed level-2 block: the probe kernel reads it)
    GDF_TRY(run_write_pass(narrow, plain && !dup_heavy, nunits, probe_lds, oa, max_build, probe_t, build_t));
    P.w[P.final_buf].clock_end(stream0());
    unsigned long long st[2] = {0, 0};
    HIP_TRY(read_back(st, d_state.p, sizeof(st)));
    clk.mark("write pass");
    if (st[1] == 0 && st[0] == cap_pairs) {       // dense: every slot of every unit was written
      if (probe_tail) {
        GDF_LAUNCH("jk_emit_unjoinable", jk_emit_unjoinable, dim3(small_grid(probe_t.nrows)), dim3(256), 0, stream0(), probe_t, plan,
                           oa.out_probe + cap_pairs, oa.out_build + cap_pairs, d_tail.as<unsigned long long>(), pay_move_at(cap_pairs));
        HIP_CHECK_LAST();
      }
      HIP_TRY(hipStreamSynchronize(stream0()));
      *out_n = (int64_t)total;
      if (total == 0) { *out_probe = nullptr; *out_build = nullptr; return GDF_SUCCESS; }
      *out_probe = (int32_t *)op.release();
      *out_build = (int32_t *)ob.release();
      commit_pay();
      return GDF_SUCCESS;
    }
    if (try_sparse && st[1] == 0 && st[0] * 3 >= cap_pairs && !lab::knob_on("GDF_JK_NO_HOLE_FILL")) {
      // most slots are taken: move only the pairs behind the final size into the holes in front of it (jk_fill_holes)
      const uint64_t pairs = st[0];
      DevBuf d_holes, d_tails;
      RMM_TRY(d_holes.alloc(sizeof(uint64_t) * (nunits + 1)));
      RMM_TRY(d_tails.alloc(sizeof(uint64_t) * (nunits + 1)));
      GDF_LAUNCH("jk_hole_counts", jk_hole_counts, dim3((unsigned)(nunits / 256 + 1)), dim3(256), 0, stream0(), (const Unit *)d_units.as<Unit>(),
                 (const uint64_t *)d_off.as<uint64_t>(), (const uint32_t *)d_upairs.as<uint32_t>(), (uint32_t)nunits, pairs, d_holes.as<uint64_t>(),
                 d_tails.as<uint64_t>());
      GDF_TRY(scan_u64(d_holes.as<uint64_t>(), d_holes.as<uint64_t>(), nunits + 1, false));
      GDF_TRY(scan_u64(d_tails.as<uint64_t>(), d_tails.as<uint64_t>(), nunits + 1, false));
      const PayMove pm = pay_move_all(nullptr);
      GDF_LAUNCH("jk_fill_holes", jk_fill_holes, dim3((unsigned)nunits), dim3(256), 0, stream0(), (const uint64_t *)d_off.as<uint64_t>(),
                 (const uint32_t *)d_upairs.as<uint32_t>(), (uint32_t)nunits, pairs, (const uint64_t *)d_holes.as<uint64_t>(),
                 (const uint64_t *)d_tails.as<uint64_t>(), op.as<int32_t>(), ob.as<int32_t>(), pm);
      HIP_CHECK_LAST();
      HIP_TRY(hipStreamSynchronize(stream0()));
      clk.mark("hole filling");
      *out_n = (int64_t)pairs;
      *out_probe = (int32_t *)op.release();        // (allocated for cap_pairs slots, `pairs` of them in use)
      *out_build = (int32_t *)ob.release();
      commit_pay();
      return GDF_SUCCESS;
    }
    if (try_sparse && st[1] == 0) {               // no unit ran out of slots: close the holes
      const uint64_t pairs = st[0];
      *out_n = (int64_t)pairs;
      if (pairs == 0) { *out_probe = nullptr; *out_build = nullptr; return GDF_SUCCESS; }
      DevBuf d_poff, fp, fb;
      RMM_TRY(d_poff.alloc(sizeof(uint64_t) * (nunits + 1)));
      GDF_LAUNCH("jk_widen_counts", jk_widen_counts, dim3(small_grid(nunits + 1)), dim3(256), 0, stream0(), (const uint32_t *)d_upairs.as<uint32_t>(),
                 d_poff.as<uint64_t>(), (uint32_t)nunits);
      GDF_TRY(scan_u64(d_poff.as<uint64_t>(), d_poff.as<uint64_t>(), nunits + 1, false));
      RMM_TRY(fp.alloc(sizeof(int32_t) * pairs));
      RMM_TRY(fb.alloc(sizeof(int32_t) * pairs));
      DevBuf dense_pay[5];
      for (int c = 0; c < ncc; ++c) RMM_TRY(dense_pay[c].alloc((size_t)cc[c].width * pairs));
      const PayMove pm = pay_move_all(&dense_pay);
      GDF_LAUNCH("jk_compact_units", jk_compact_units, dim3((unsigned)nunits), dim3(256), 0, stream0(), (const uint64_t *)d_off.as<uint64_t>(),
                 (const uint64_t *)d_poff.as<uint64_t>(), (const int32_t *)op.as<int32_t>(), (const int32_t *)ob.as<int32_t>(), fp.as<int32_t>(),
                 fb.as<int32_t>(), pm);
      HIP_CHECK_LAST();
      HIP_TRY(hipStreamSynchronize(stream0()));
      clk.mark("compaction");
      *out_probe = (int32_t *)fp.release();
      *out_build = (int32_t *)fb.release();
      for (int c = 0; c < ncc; ++c) { cc[c].buf.reset(); cc[c].buf.p = dense_pay[c].release(); }
      commit_pay();
      return GDF_SUCCESS;
    }
    drop_pay();      // the attempt is discarded: the two-pass path sizes its own columns
    // otherwise: fall through to the exact two-pass path (buffers above are released here)
  }
  const size_t nslots_all = nunits + oversize.size();     // one count slot per LDS unit + one per oversize partition
  RMM_TRY(d_counts.alloc(sizeof(uint64_t) * (nslots_all + 1)));
  HIP_TRY(hipMemsetAsync(d_counts.p, 0, sizeof(uint64_t) * (nslots_all + 1), stream0()));
  a.counts = d_counts.as<uint64_t>();

  // repeated build keys in a plain NARROW join without carried columns: the lean multimap kernel serves both passes
  const bool multi = plain && dup_heavy && narrow && ncc == 0 && !lab::path_on("GDF_JK_NO_MULTI");
  // ---- count pass ----
  {
    DevBuf d_cstate;
    RMM_TRY(d_cstate.alloc(sizeof(unsigned long long) * 4));
    HIP_TRY(hipMemsetAsync(d_cstate.p, 0, sizeof(unsigned long long) * 4, stream0()));
    ProbeArgs ca = a;
    ca.opt_state = d_cstate.as<unsigned long long>();
    if (multi) GDF_TRY(run_multi_pass(false, nunits, probe_lds, ca));
    else GDF_TRY(run_count_pass(narrow, plain && !dup_heavy, nunits, probe_lds, ca, max_build, probe_t, build_t));
  }
  // oversize partitions: one global table each, kept for the write pass
  struct GTable { DevBuf key, idx, next; uint32_t nslots; };      // idx: chain heads per slot (+ the reserved slot), next: per build tuple
  std::vector<GTable> gt(oversize.size());
  for (size_t o = 0; o < oversize.size(); ++o) {
    const uint32_t f = oversize[o].f0, fe = oversize[o].f1;
    const uint32_t bn = B.fine_off[fe] - B.fine_off[f], pn = P.fine_off[fe] - P.fine_off[f];
    gt[o].nslots = bn * 2;
    RMM_TRY(gt[o].key.alloc(sizeof(uint64_t) * ((size_t)gt[o].nslots + 1)));
    RMM_TRY(gt[o].idx.alloc(sizeof(int32_t) * ((size_t)gt[o].nslots + 1)));
    RMM_TRY(gt[o].next.alloc(sizeof(int32_t) * (size_t)(bn ? bn : 1)));
    hipLaunchKernelGGL(gj_fill, dim3(small_grid(gt[o].nslots + 1)), dim3(256), 0, stream0(), gt[o].key.as<unsigned long long>(),
                       gt[o].idx.as<int32_t>(), gt[o].nslots);
    ProbeArgs ga = a;
    ga.nslots = gt[o].nslots;
    unsigned long long *cnt = (unsigned long long *)(d_counts.as<uint64_t>() + nunits + o);
    if (narrow) {
      hipLaunchKernelGGL(gj_build<true>, dim3(small_grid(bn)), dim3(256), 0, stream0(), a.build, B.fine_off[f], bn,
                         gt[o].key.as<unsigned long long>(), gt[o].idx.as<int32_t>(), gt[o].next.as<int32_t>(), gt[o].nslots);
      hipLaunchKernelGGL((gj_probe<false, true>), dim3(small_grid(pn)), dim3(256), 0, stream0(), ga, probe_t, build_t,
                         (const unsigned long long *)gt[o].key.as<unsigned long long>(), (const int32_t *)gt[o].idx.as<int32_t>(),
                         (const int32_t *)gt[o].next.as<int32_t>(), B.fine_off[f], P.fine_off[f], pn, cnt);
    } else {
      hipLaunchKernelGGL(gj_build<false>, dim3(small_grid(bn)), dim3(256), 0, stream0(), a.build, B.fine_off[f], bn,
                         gt[o].key.as<unsigned long long>(), gt[o].idx.as<int32_t>(), gt[o].next.as<int32_t>(), gt[o].nslots);
      hipLaunchKernelGGL((gj_probe<false, false>), dim3(small_grid(pn)), dim3(256), 0, stream0(), ga, probe_t, build_t,
                         (const unsigned long long *)gt[o].key.as<unsigned long long>(), (const int32_t *)gt[o].idx.as<int32_t>(),
                         (const int32_t *)gt[o].next.as<int32_t>(), B.fine_off[f], P.fine_off[f], pn, cnt);
    }
    HIP_CHECK_LAST();
  }

  // ---- sizes ----
  GDF_TRY(scan_u64(d_counts.as<uint64_t>(), d_counts.as<uint64_t>(), nslots_all + 1, false));
  uint64_t matched_total = 0;
  HIP_TRY(read_back(&matched_total, d_counts.as<uint64_t>() + nslots_all, sizeof(uint64_t)));
  const uint64_t probe_tail = keep_probe ? (uint64_t)probe_t.nrows - P.joinable : 0;   // rows that cannot match
  uint64_t build_tail = 0;
  if (kind == JOIN_FULL) {
    unsigned long long *d_cnt = d_tail.as<unsigned long long>() + 2;
    GDF_LAUNCH("jk_count_unmatched", jk_count_unmatched, dim3(small_grid(build_t.nrows)), dim3(256), 0, stream0(), d_matched.as<uint8_t>(),
                       build_t.nrows, d_cnt);
    HIP_CHECK_LAST();
    unsigned long long h = 0;
    HIP_TRY(read_back(&h, d_cnt, sizeof(h)));
    build_tail = h;
  }
  const uint64_t total = matched_total + probe_tail + build_tail;
  if (total >= (uint64_t)INT_MAX) return GDF_COLUMN_SIZE_TOO_BIG;   // int32 index columns
  *out_n = (int64_t)total;
  if (total == 0) { *out_probe = nullptr; *out_build = nullptr; return GDF_SUCCESS; }

  DevBuf op, ob;
  RMM_TRY(op.alloc(sizeof(int32_t) * total));
  RMM_TRY(ob.alloc(sizeof(int32_t) * total));
  GDF_TRY(alloc_pay(total));
  set_pay(a);
  a.out_probe = op.as<int32_t>();
  a.out_build = ob.as<int32_t>();
  a.build_matched = nullptr;   // marks were taken in the count pass

  // ---- write pass ----
  DevBuf d_wstate;
  RMM_TRY(d_wstate.alloc(sizeof(unsigned long long) * 4));
  HIP_TRY(hipMemsetAsync(d_wstate.p, 0, sizeof(unsigned long long) * 4, stream0()));
  a.opt_state = d_wstate.as<unsigned long long>();
  if (multi) GDF_TRY(run_multi_pass(true, nunits, probe_lds, a));
  else GDF_TRY(run_write_pass(narrow, plain && !dup_heavy, nunits, probe_lds, a, max_build, probe_t, build_t));
  for (size_t o = 0; o < oversize.size(); ++o) {
    const uint32_t f = oversize[o].f0, fe = oversize[o].f1;
    const uint32_t pn = P.fine_off[fe] - P.fine_off[f];
    ProbeArgs ga = a;
    ga.nslots = gt[o].nslots;
    // the exclusive offset of this partition doubles as its write cursor
    unsigned long long *cur = (unsigned long long *)(d_counts.as<uint64_t>() + nunits + o);
    if (narrow)
      hipLaunchKernelGGL((gj_probe<true, true>), dim3(small_grid(pn)), dim3(256), 0, stream0(), ga, probe_t, build_t,
                         (const unsigned long long *)gt[o].key.as<unsigned long long>(), (const int32_t *)gt[o].idx.as<int32_t>(),
                         (const int32_t *)gt[o].next.as<int32_t>(), B.fine_off[f], P.fine_off[f], pn, cur);
    else
      hipLaunchKernelGGL((gj_probe<true, false>), dim3(small_grid(pn)), dim3(256), 0, stream0(), ga, probe_t, build_t,
                         (const unsigned long long *)gt[o].key.as<unsigned long long>(), (const int32_t *)gt[o].idx.as<int32_t>(),
                         (const int32_t *)gt[o].next.as<int32_t>(), B.fine_off[f], P.fine_off[f], pn, cur);
    HIP_CHECK_LAST();
  }
  if (probe_tail) {
    unsigned long long *cur = d_tail.as<unsigned long long>();
    GDF_LAUNCH("jk_emit_unjoinable", jk_emit_unjoinable, dim3(small_grid(probe_t.nrows)), dim3(256), 0, stream0(), probe_t, plan,
                       a.out_probe + matched_total, a.out_build + matched_total, cur, pay_move_at(matched_total));
    HIP_CHECK_LAST();
  }
  if (build_tail) {
    unsigned long long *cur = d_tail.as<unsigned long long>() + 1;
    GDF_LAUNCH("jk_emit_unmatched_build", jk_emit_unmatched_build, dim3(small_grid(build_t.nrows)), dim3(256), 0, stream0(),
                       d_matched.as<uint8_t>(), build_t.nrows, a.out_probe + matched_total + probe_tail,
                       a.out_build + matched_total + probe_tail, cur);
    HIP_CHECK_LAST();
  }
  HIP_TRY(hipStreamSynchronize(stream0()));
  *out_probe = (int32_t *)op.release();
  *out_build = (int32_t *)ob.release();
  commit_pay();
  return GDF_SUCCESS;
}

static gdf_error hash_join_core(const KeyTable &probe_t, const KeyTable &build_t, JoinKind kind, int32_t **out_probe,
                                int32_t **out_build, int64_t *out_n, PayCarry *pc = nullptr) {
  StageClock clk((lab::knob_int("GDF_JK_DBG", 0) & 512) != 0);
  BuildSide bs;
  PaySrc bsrc{};
  const bool bcarry = pc && pc->mode >= 0 && pc->bmode && kind == JOIN_INNER && !lab::path_on("GDF_JK_NO_CARRY");
  if (bcarry) { bsrc.col[0] = pc->bsrc[0]; bsrc.col[1] = pc->bsrc[1]; }
  SkewProbe early_skew;
  {
    const KeyPlan p0 = plan_keys(build_t);         // (what prepare_build starts from; the sample hashes RAW keys: no range, no tuple format in it)
    const PartGeom g0 = choose_geometry(build_t.nrows);
    if (build_t.ncols == 1 && skew_probe_wanted(probe_t, p0, g0)) GDF_TRY(skew_probe_launch(probe_t, p0, g0, &early_skew));
  }
  // (the build side's last launches are still queued when this returns: the probe side's first ones follow without a host round trip)
  GDF_TRY(prepare_build(build_t, &bs, false, bcarry ? &bsrc : nullptr, bcarry ? pc->bmode : 0, true));
  clk.mark("partition build side");
  gdf_error e = probe_prepared(probe_t, build_t, bs, kind, out_probe, out_build, out_n, clk, pc, &early_skew);
  bs.B.settle();
  if (e != GDF_AMD_RETRY_WITHOUT_LEVEL3) return e;
  BuildSide plain;
  GDF_TRY(prepare_build(build_t, &plain, true, bcarry ? &bsrc : nullptr, bcarry ? pc->bmode : 0));
  return probe_prepared(probe_t, build_t, plain, kind, out_probe, out_build, out_n, clk, pc, &early_skew);
}

// FULL join with an empty side (joining.cu:214-280 trivial_full_join): every row of
// the non-empty side paired with -1.
static gdf_error trivial_full_join(int64_t left_rows, int64_t right_rows, gdf_column *left_result, gdf_column *right_result) {
  const int64_t n = left_rows > 0 ? left_rows : right_rows;
  DevBuf l, r;
  RMM_TRY(l.alloc(sizeof(int32_t) * (size_t)n));
  RMM_TRY(r.alloc(sizeof(int32_t) * (size_t)n));
  hipLaunchKernelGGL(jk_fill_pairs, dim3(small_grid(n)), dim3(256), 0, stream0(), l.as<int32_t>(), -1, r.as<int32_t>(), -1, n,
                     left_rows > 0 ? 1 : 0, left_rows > 0 ? 0 : 1);
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  gdf_column_view(left_result, l.release(), nullptr, (gdf_size_type)n, GDF_INT32);
  gdf_column_view(right_result, r.release(), nullptr, (gdf_size_type)n, GDF_INT32);
  return GDF_SUCCESS;
}

// joining.cu:282-373 join_call: validation + method dispatch
// pc (may be null): payload columns of the PROBE relation to carry (join_entry decides which relation that is, by the same
// rule as below); *flipped tells the caller whether the probe relation was the right one
static gdf_error join_call(JoinKind kind, int num_cols, gdf_column **leftcol, gdf_column **rightcol,
                           gdf_column *left_result, gdf_column *right_result, gdf_context *ctx, PayCarry *pc = nullptr) {
  if (0 == num_cols || nullptr == leftcol || nullptr == rightcol) return GDF_DATASET_EMPTY;
  if (nullptr == ctx) return GDF_INVALID_API_CALL;
  const size_t left_size = leftcol[0]->size, right_size = rightcol[0]->size;
  if (left_size >= (size_t)INT_MAX) return GDF_COLUMN_SIZE_TOO_BIG;
  if (right_size >= (size_t)INT_MAX) return GDF_COLUMN_SIZE_TOO_BIG;
  if (0 == left_size && 0 == right_size) return GDF_SUCCESS;
  if (kind == JOIN_LEFT && 0 == left_size) return GDF_SUCCESS;
  if (kind == JOIN_INNER && (0 == left_size || 0 == right_size)) return GDF_SUCCESS;
  // LEFT with an empty right relation has no early return in the reference (joining.cu:304-323): every left row pairs
  // with -1, which is the FULL join's answer too; no kernel here ever sees a zero-row relation (its data may be null)
  if ((kind == JOIN_FULL && (0 == left_size || 0 == right_size)) || (kind == JOIN_LEFT && 0 == right_size))
    return trivial_full_join((int64_t)left_size, (int64_t)right_size, left_result, right_result);
  for (int i = 0; i < num_cols; ++i) {
    if (right_size > 0 && nullptr == rightcol[i]->data) return GDF_DATASET_EMPTY;
    if (left_size > 0 && nullptr == leftcol[i]->data) return GDF_DATASET_EMPTY;
    if (rightcol[i]->dtype != leftcol[i]->dtype) return GDF_JOIN_DTYPE_MISMATCH;
    if (left_size != leftcol[i]->size) return GDF_COLUMN_SIZE_MISMATCH;
    if (right_size != rightcol[i]->size) return GDF_COLUMN_SIZE_MISMATCH;
  }
  const bool sort_method = ctx->flag_method == GDF_SORT;
  if (sort_method) {
    // GDF_SORT (joining.cu:100-159, 352-365): one key column, no masks, floats compared by BIT
    // PATTERN (sort_join dispatches FLOAT32/64 to int32_t/int64_t).  The set of index pairs
    // an equi-join produces does not depend on the algorithm, so the request runs on the same
    // partitioned join as GDF_HASH; only the pair ORDER differs from a merge join's, and no
    // caller contract covers it.  FULL has no sort implementation in the reference: the
    // generic SortJoin returns two empty columns and success (joining.cu:66-75).
    if (num_cols != 1) return GDF_JOIN_TOO_MANY_COLUMNS;
    GDF_REQUIRE(!leftcol[0]->valid && !rightcol[0]->valid, GDF_VALIDITY_UNSUPPORTED);
    if (kind == JOIN_FULL) {
      gdf_column_view(left_result, nullptr, nullptr, 0, N_GDF_TYPES);
      gdf_column_view(right_result, nullptr, nullptr, 0, N_GDF_TYPES);
      return GDF_SUCCESS;
    }
  } else if (ctx->flag_method != GDF_HASH) {
    return GDF_UNSUPPORTED_METHOD;
  }

  gdf_nvtx_range_push("LIBGDF_JOIN", GDF_CYAN);   // joining.cu:343
  struct Pop { ~Pop() { gdf_nvtx_range_pop(); } } pop;

  KeyTable lt, rt;
  GDF_TRY(make_key_table(leftcol, num_cols, &lt));
  GDF_TRY(make_key_table(rightcol, num_cols, &rt));
  if (sort_method)
    for (KeyTable *t : {&lt, &rt}) {
      if (t->col[0].kind == K_F32) t->col[0].kind = K_I32;
      if (t->col[0].kind == K_F64) t->col[0].kind = K_I64;
    }
  // compute_hash_join starts by clearing both outputs (join_compute_api.h:353-354)
  gdf_column_view(left_result, nullptr, nullptr, 0, N_GDF_TYPES);
  gdf_column_view(right_result, nullptr, nullptr, 0, N_GDF_TYPES);

  // the table is built on the RIGHT relation; INNER builds on the smaller one (joining.h:58-66)
  const bool flip = kind == JOIN_INNER && right_size > left_size;
  int32_t *o_probe = nullptr, *o_build = nullptr;
  int64_t n = 0;
  GDF_TRY(hash_join_core(flip ? rt : lt, flip ? lt : rt, kind, &o_probe, &o_build, &n, sort_method ? nullptr : pc));
  if (n == 0) return GDF_SUCCESS;   // size-0 outputs with dtype N_GDF_TYPES (join_compute_api.h:439-441)
  gdf_column_view(left_result, flip ? o_build : o_probe, nullptr, (gdf_size_type)n, GDF_INT32);
  gdf_column_view(right_result, flip ? o_probe : o_build, nullptr, (gdf_size_type)n, GDF_INT32);
  return GDF_SUCCESS;
}

// ---------------------------------------------------------------------------
// optional materialisation of the joined rows (joining.cu:375-479 + the gathers of
// gdf_table.cuh:873-963): out = [left non-key..., key..., right non-key...]
// ---------------------------------------------------------------------------
// One launch gathers up to GS_MAX_COLS columns through ONE index map (the map is read once per row, not once per
// column), GS_ROWS rows per thread so that every column has GS_ROWS independent random reads in flight per lane, and the
// validity words come from ballots over 64 consecutive rows (two whole 32-bit words per wave, no atomics).
// A column may name a second source (FULL join key columns): rows whose map entry is negative take alt[alt_map[i]].
constexpr int GS_MAX_COLS = 8;
constexpr int GS_ROWS = 4;
struct GatherSet {
  int ncols;
  int width[GS_MAX_COLS];
  const void *in[GS_MAX_COLS];
  const uint8_t *in_valid[GS_MAX_COLS];      // may be null: every source row valid
  const void *alt[GS_MAX_COLS];              // may be null
  const uint8_t *alt_valid[GS_MAX_COLS];
  void *out[GS_MAX_COLS];
  uint32_t *out_valid[GS_MAX_COLS];
};
template <class T>
__device__ __forceinline__ void gather_one_column(const GatherSet &s, int c, const int32_t (&src)[GS_ROWS], const int32_t (&asrc)[GS_ROWS],
                                                  int64_t i0, int64_t n) {
  const T *in = (const T *)s.in[c], *alt = (const T *)s.alt[c];
  T v[GS_ROWS];
  bool valid[GS_ROWS];
#pragma unroll
  for (int r = 0; r < GS_ROWS; ++r) {       // all random reads first (clamped row 0 stands in for "no source row")
    const bool from_alt = src[r] < 0 && alt != nullptr && asrc[r] >= 0;
    const T *base = from_alt ? alt : in;
    const int32_t row = from_alt ? asrc[r] : src[r];
    v[r] = base[row >= 0 ? row : 0];
    const uint8_t *vb = from_alt ? s.alt_valid[c] : s.in_valid[c];
    valid[r] = row >= 0 && (vb == nullptr || ((vb[row >> 3] >> (row & 7)) & 1));
  }
  T *out = (T *)s.out[c];
#pragma unroll
  for (int r = 0; r < GS_ROWS; ++r) {
    const int64_t i = i0 + (int64_t)r * 256;
    const bool live = i < n;
    if (live && (src[r] >= 0 || (alt != nullptr && asrc[r] >= 0))) out[i] = v[r];
    const unsigned long long m = __ballot(live && valid[r]);
    if (live) {
      if (lane_id() == 0) s.out_valid[c][i >> 5] = (uint32_t)m;
      if (lane_id() == 32) s.out_valid[c][i >> 5] = (uint32_t)(m >> 32);
    }
  }
}
__global__ __launch_bounds__(256) void jk_gather_multi(GatherSet s, const int32_t *__restrict__ map, const int32_t *__restrict__ alt_map,
                                                       int64_t n) {
  // XCD x (= blockIdx % 8, round-robin dispatch) takes the x-th contiguous EIGHTH of the output: the pairs of one build
  // partition are neighbours in a join's output and name the same few thousand build rows, so the gather of a build-side
  // column re-reads each 64-byte sector ~10 times within a window of ~30 k pairs -- from ONE L2 when that window belongs to
  // one XCD.  In dispatch order the window's blocks were dealt to all eight XCDs and every one of them fetched the
  // partition's rows from HBM for itself (C3, two build-side columns: 28.9 ms of gathers)
  const uint32_t per_xcd = gridDim.x >> 3;
  const uint32_t block = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
  const int64_t i0 = (int64_t)block * (256 * GS_ROWS) + threadIdx.x;
  int32_t src[GS_ROWS], asrc[GS_ROWS];
#pragma unroll
  for (int r = 0; r < GS_ROWS; ++r) {
    const int64_t i = i0 + (int64_t)r * 256;
    src[r] = map[i < n ? i : n - 1];
    asrc[r] = alt_map ? alt_map[i < n ? i : n - 1] : -1;
  }
  for (int c = 0; c < s.ncols; ++c) {
    switch (s.width[c]) {
      case 1: gather_one_column<uint8_t>(s, c, src, asrc, i0, n); break;
      case 2: gather_one_column<uint16_t>(s, c, src, asrc, i0, n); break;
      case 4: gather_one_column<uint32_t>(s, c, src, asrc, i0, n); break;
      default: gather_one_column<uint64_t>(s, c, src, asrc, i0, n); break;
    }
  }
}

// what one output column of the materialisation is made from
struct GatherJob {
  const gdf_column *src;
  const gdf_column *alt;      // FULL join key columns: the right key where there is no left row
  gdf_column *dst;
};
// allocates the outputs of `jobs` and gathers them through `map` (and `alt_map`), GS_MAX_COLS columns per launch; no
// synchronisation -- the caller waits once for all sides
static gdf_error gather_columns(const std::vector<GatherJob> &jobs, const int32_t *map, const int32_t *alt_map, int64_t n) {
  for (size_t j0 = 0; j0 < jobs.size(); j0 += GS_MAX_COLS) {
    GatherSet s{};
    for (size_t j = j0; j < jobs.size() && j < j0 + GS_MAX_COLS; ++j) {
      const GatherJob &job = jobs[j];
      const int w = dtype_width(job.src->dtype);
      if (w < 0) return GDF_UNSUPPORTED_DTYPE;
      DevBuf data, valid;
      RMM_TRY(data.alloc((size_t)w * (size_t)(n ? n : 1)));
      const size_t vbytes = ((mask_bytes((size_t)n) + 7) / 8) * 8;   // whole 64-bit groups
      RMM_TRY(valid.alloc(vbytes ? vbytes : 8));
      HIP_TRY(hipMemsetAsync(valid.p, 0, vbytes ? vbytes : 8, stream0()));
      const int c = s.ncols++;
      s.width[c] = w;
      s.in[c] = job.src->data;
      s.in_valid[c] = job.src->valid;
      s.alt[c] = job.alt ? job.alt->data : nullptr;
      s.alt_valid[c] = job.alt ? job.alt->valid : nullptr;
      s.out[c] = data.p;
      s.out_valid[c] = valid.as<uint32_t>();
      gdf_column_view(job.dst, data.release(), (gdf_valid_type *)valid.release(), (gdf_size_type)n, job.src->dtype);
      job.dst->dtype_info = job.src->dtype_info;
    }
    if (n) {
      unsigned grid = (unsigned)((n + 256 * GS_ROWS - 1) / (256 * GS_ROWS));
      if (grid >= 64) grid = (grid + 7u) & ~7u;          // a multiple of 8: the kernel's XCD-contiguous block order (blocks past the end do nothing)
      GDF_LAUNCH("jk_gather_multi", jk_gather_multi, dim3(grid), dim3(256), 0, stream0(), s, map, alt_map, n);
      HIP_CHECK_LAST();
    }
  }
  return GDF_SUCCESS;
}

static gdf_error join_entry(JoinKind kind, gdf_column **left_cols, int num_left_cols, int left_join_cols[],
                            gdf_column **right_cols, int num_right_cols, int right_join_cols[], int num_cols_to_join,
                            int result_num_cols, gdf_column **result_cols, gdf_column *left_indices,
                            gdf_column *right_indices, gdf_context *ctx) {
  // joining.cu:495-511
  if (nullptr == left_cols || nullptr == right_cols) return GDF_DATASET_EMPTY;
  if (0 == num_cols_to_join) return GDF_SUCCESS;
  if (nullptr == left_join_cols || nullptr == right_join_cols) return GDF_DATASET_EMPTY;
  const bool compute_df = result_cols != nullptr;
  if ((nullptr == left_indices || nullptr == right_indices) && !compute_df) return GDF_DATASET_EMPTY;
  if (nullptr == ctx) return GDF_INVALID_API_CALL;

  gdf_column tmp_l{}, tmp_r{};
  gdf_column *lout = left_indices ? left_indices : &tmp_l;
  gdf_column *rout = right_indices ? right_indices : &tmp_r;
  struct Cleanup {   // temporaries created for the materialisation only
    gdf_column *l, *r;
    ~Cleanup() { if (l) gdf_column_free(l); if (r) gdf_column_free(r); }
  } cleanup{left_indices ? nullptr : &tmp_l, right_indices ? nullptr : &tmp_r};

  std::vector<gdf_column *> lj(num_cols_to_join), rj(num_cols_to_join);
  for (int i = 0; i < num_cols_to_join; ++i) {
    lj[i] = left_cols[left_join_cols[i]];
    rj[i] = right_cols[right_join_cols[i]];
  }
  // ---- payload carrying (PayCarry): when result_cols are wanted and the PROBE relation -- the left one, or the larger one
  // of an INNER join (join_call's rule) -- has one 8-byte, one 4-byte or two 4-byte non-key columns without masks, their
  // values travel through the join's partition passes next to the (key, row) tuples and the probe kernel writes the
  // result columns streaming.  The gather they replace reads one random 64-byte sector per value: 1e9 of them for C3.
  const int expect = num_left_cols + num_right_cols - num_cols_to_join;
  PayCarry pc;
  int carried_col[2] = {-1, -1};       // column numbers in the probe relation
  int bcarried_col[2] = {-1, -1};      // ... and in the build relation
  bool probe_is_right = false;
  if (compute_df && result_num_cols == expect && kind != JOIN_FULL && ctx->flag_method == GDF_HASH && num_left_cols > 0 && num_right_cols > 0 &&
      left_cols[0] && right_cols[0]) {
    probe_is_right = kind == JOIN_INNER && lj[0] && rj[0] && rj[0]->size > lj[0]->size;
    gdf_column **pcols = probe_is_right ? right_cols : left_cols;
    const int npcols = probe_is_right ? num_right_cols : num_left_cols;
    const int *pkeys = probe_is_right ? right_join_cols : left_join_cols;
    std::vector<int> nonkey;
    for (int c = 0; c < npcols; ++c) {
      bool is_key = false;
      for (int i = 0; i < num_cols_to_join; ++i) is_key = is_key || pkeys[i] == c;
      if (!is_key) nonkey.push_back(c);
    }
    // mode of a relation's non-key columns: 1 = one 8-byte column, 2 = one 4-byte, 3 = two 4-byte, 0 = none carried.  One
    // 64-bit word per side travels: of several candidates (unmasked, 8 or 4 bytes wide) the first 8-byte column is taken,
    // else the first two 4-byte ones; the relation's other non-key columns (more of them, masked ones, other widths) are
    // gathered through the index columns as before.  (A second word per side was priced and not built: carrying 8 bytes
    // through both partition levels moves 48 bytes per row, the gather it replaces one 64-byte sector + 12 -- DESIGN 3.7.)
    auto payload_mode = [&](gdf_column **cols, const std::vector<int> &which, const gdf_column *keycol, int (&slot)[2], const void *(&src)[2]) -> int {
      int wide = -1, wide2 = -1, narrow[2] = {-1, -1}, nn = 0;
      for (size_t j = 0; j < which.size(); ++j) {
        const gdf_column *col = cols[which[j]];
        const int w = col ? dtype_width(col->dtype) : -1;
        if (!col || !col->data || col->valid || col->size != keycol->size) continue;
        if (w == 8 && wide < 0) wide = which[j];
        else if (w == 8 && wide2 < 0) wide2 = which[j];
        if (w == 4 && nn < 2) narrow[nn++] = which[j];
      }
      // (round 6, mode 4: a relation with two unmasked 8-byte non-key columns carries BOTH -- a 16-byte payload element)
      if (wide >= 0 && wide2 >= 0 && !lab::path_on("GDF_JK_NO_CARRY2")) {
        src[0] = cols[wide]->data; slot[0] = wide; src[1] = cols[wide2]->data; slot[1] = wide2;
        return 4;
      }
      if (wide >= 0) { src[0] = cols[wide]->data; slot[0] = wide; return 1; }
      for (int j = 0; j < nn; ++j) { src[j] = cols[narrow[j]]->data; slot[j] = narrow[j]; }
      return nn == 2 ? 3 : (nn == 1 ? 2 : 0);
    };
    pc.mode = payload_mode(pcols, nonkey, pcols[pkeys[0]], carried_col, pc.src);
    if (kind == JOIN_INNER) {            // the build relation's non-key columns (jk_probe_bp)
      gdf_column **bcols = probe_is_right ? left_cols : right_cols;
      const int nbcols = probe_is_right ? num_left_cols : num_right_cols;
      const int *bkeys = probe_is_right ? left_join_cols : right_join_cols;
      std::vector<int> bnonkey;
      for (int c = 0; c < nbcols; ++c) {
        bool is_key = false;
        for (int i = 0; i < num_cols_to_join; ++i) is_key = is_key || bkeys[i] == c;
        if (!is_key) bnonkey.push_back(c);
      }
      pc.bmode = payload_mode(bcols, bnonkey, bcols[bkeys[0]], bcarried_col, pc.bsrc);
    }
    const bool plain = pc.mode != 0 || pc.bmode != 0;
    if (plain) {
      // the result's key column: an INNER join's matched pair holds the same key bits on both sides when the key is ONE
      // integer column of one dtype without nulls -- the probe kernel then writes it from the tuple (no gather at all)
      if (kind == JOIN_INNER && num_cols_to_join == 1 && lj[0] && rj[0]) {
        const gdf_column *lk = lj[0], *rk = rj[0];
        const ElemKind ek = elem_kind(lk->dtype);
        if (lk->dtype == rk->dtype && lk->dtype_info.time_unit == rk->dtype_info.time_unit && (ek == K_I32 || ek == K_I64) &&
            !lk->valid && !rk->valid) {
          pc.key_width = kind_width(ek);
          pc.key_src = (probe_is_right ? rk : lk)->data;
        }
      }
    }
  }
  struct PayFree {     // carried columns that were not handed to result_cols (an error below): released here
    PayCarry &pc;
    ~PayFree() {
      for (int c = 0; c < 2; ++c) if (pc.dst[c]) rmmFree(pc.dst[c], (cudaStream_t)0);
      for (int c = 0; c < 2; ++c) if (pc.bdst[c]) rmmFree(pc.bdst[c], (cudaStream_t)0);
      if (pc.key_dst) rmmFree(pc.key_dst, (cudaStream_t)0);
    }
  } pay_free{pc};

  gdf_error err = join_call(kind, num_cols_to_join, lj.data(), rj.data(), lout, rout, ctx, (pc.mode || pc.bmode) ? &pc : nullptr);
  if (!compute_df || err != GDF_SUCCESS) return err;

  // ---- materialise: [left non-key..., key columns..., right non-key...] ----
  if (result_num_cols != expect) return GDF_INVALID_API_CALL;
  gdf_nvtx_range_push("LIBGDF_JOIN_OUTPUT", GDF_CYAN);   // joining.cu:391
  struct Pop { ~Pop() { gdf_nvtx_range_pop(); } } pop;
  const int64_t n = (int64_t)lout->size;
  const int32_t *lmap = (const int32_t *)lout->data, *rmap = (const int32_t *)rout->data;
  std::vector<char> l_is_key(num_left_cols, 0), r_is_key(num_right_cols, 0);
  for (int i = 0; i < num_cols_to_join; ++i) { l_is_key[left_join_cols[i]] = 1; r_is_key[right_join_cols[i]] = 1; }
  // Two launches (per 8 columns): everything that follows the LEFT index map -- the left non-key columns and the key
  // columns, whose values come from the left row when there is one, else from the right one (FULL join tail) -- and
  // the right non-key columns.  Round 1 ran one gather kernel + one stream synchronisation per output column and a
  // second gather + merge pass per FULL-join key column.
  std::vector<GatherJob> left_jobs, right_jobs;
  // a carried column is complete already: its data came out of the probe kernel, every row of it is valid (no input mask;
  // the probe row of a pair always exists in an INNER / LEFT join)
  auto adopt = [&](void *&data, const gdf_column *src, gdf_column *dst) -> int {     // 1: dst now owns `data`, all rows valid; < 0: -(error)
    {
      DevBuf valid;
      const size_t vbytes = ((mask_bytes((size_t)n) + 7) / 8) * 8;
      if (valid.alloc(vbytes ? vbytes : 8) != RMM_SUCCESS) return -(int)GDF_MEMORYMANAGER_ERROR;
      if (hipMemsetAsync(valid.p, 0, vbytes ? vbytes : 8, stream0()) != hipSuccess) return -(int)GDF_CUDA_ERROR;
      if (n / 8 && hipMemsetAsync(valid.p, 0xff, (size_t)(n / 8), stream0()) != hipSuccess) return -(int)GDF_CUDA_ERROR;
      if (n % 8) {
        const uint8_t tail = (uint8_t)((1u << (n % 8)) - 1u);
        if (hipMemcpy(valid.as<uint8_t>() + n / 8, &tail, 1, hipMemcpyHostToDevice) != hipSuccess) return -(int)GDF_CUDA_ERROR;
      }
      gdf_column_view(dst, data, (gdf_valid_type *)valid.release(), (gdf_size_type)n, src->dtype);
      dst->dtype_info = src->dtype_info;
      data = nullptr;                            // owned by the result column now
      return 1;
    }
  };
  auto take_carried = [&](bool right_side, int c, gdf_column *dst) -> int {      // 1: taken, 0: not a carried column, < 0: -(error)
    if (right_side == probe_is_right) {
      if (!pc.carried) return 0;
      for (int j = 0; j < pc.ncols(); ++j)
        if (carried_col[j] == c && pc.dst[j]) return adopt(pc.dst[j], (right_side ? right_cols : left_cols)[c], dst);
    } else {
      if (!pc.build_carried) return 0;
      for (int j = 0; j < pc.bncols(); ++j)
        if (bcarried_col[j] == c && pc.bdst[j]) return adopt(pc.bdst[j], (right_side ? right_cols : left_cols)[c], dst);
    }
    return 0;
  };
  int o = 0;
  for (int c = 0; c < num_left_cols; ++c)
    if (!l_is_key[c]) {
      const int took = take_carried(false, c, result_cols[o]);
      if (took < 0) return (gdf_error)(-took);
      if (!took) left_jobs.push_back(GatherJob{left_cols[c], nullptr, result_cols[o]});
      ++o;
    }
  // INNER join, integer keys of one dtype: a matched pair's two key values are the same bits, so the key column may be read
  // from EITHER side -- and the rows of the smaller relation are the ones the pair list revisits (all pairs of a build
  // partition are neighbours in the output and name the same few thousand build rows: their gather mostly hits L2, while
  // the other side's row numbers are a random permutation -- one 64-byte sector fetched per 8-byte value)
  const bool right_is_smaller = num_right_cols > 0 && num_left_cols > 0 && right_cols[0]->size < left_cols[0]->size;
  for (int i = 0; i < num_cols_to_join; ++i) {
    const gdf_column *lk = left_cols[left_join_cols[i]], *rk = right_cols[right_join_cols[i]];
    const ElemKind ek = elem_kind(lk->dtype);
    const bool same_bits = kind == JOIN_INNER && lk->dtype == rk->dtype && lk->dtype_info.time_unit == rk->dtype_info.time_unit &&
                           (ek == K_I8 || ek == K_I16 || ek == K_I32 || ek == K_I64) &&
                           (lk->valid == nullptr || lk->null_count == 0) && (rk->valid == nullptr || rk->null_count == 0);
    if (pc.key_dst) {                             // came out of the probe kernel (one key column: i == 0)
      const int took = adopt(pc.key_dst, lk, result_cols[o++]);
      if (took < 0) return (gdf_error)(-took);
    }
    else if (same_bits && right_is_smaller) right_jobs.push_back(GatherJob{rk, nullptr, result_cols[o++]});
    else left_jobs.push_back(GatherJob{lk, kind == JOIN_FULL ? rk : nullptr, result_cols[o++]});
  }
  for (int c = 0; c < num_right_cols; ++c)
    if (!r_is_key[c]) {
      const int took = take_carried(true, c, result_cols[o]);
      if (took < 0) return (gdf_error)(-took);
      if (!took) right_jobs.push_back(GatherJob{right_cols[c], nullptr, result_cols[o]});
      ++o;
    }
  GDF_TRY(gather_columns(left_jobs, lmap, kind == JOIN_FULL ? rmap : nullptr, n));
  GDF_TRY(gather_columns(right_jobs, rmap, nullptr, n));
  HIP_TRY(hipStreamSynchronize(stream0()));
  HIP_CHECK_LAST();
  return GDF_SUCCESS;
}

// ---------------------------------------------------------------------------
// test hook: run the radix partitioner alone and hand back the fine-partitioned tuples
// (tests/test_gpu_join_internals.py checks them against a numpy restatement of fine_of).
// out_info[0] = 1 when the narrow 8-byte tuple format was used, out_info[1] = kmin; the
// returned keys are the STORED keys (raw bits minus kmin).
// ---------------------------------------------------------------------------
__global__ void jk_unpack_narrow(const uint64_t *w, uint32_t n, uint64_t *key, int32_t *idx) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    key[i] = w[i] >> 32;
    idx[i] = (int32_t)(uint32_t)w[i];
  }
}

gdf_error debug_partition(gdf_column *col, int fb, uint64_t *out_key, int32_t *out_idx, uint32_t *out_fine_off,
                          uint32_t *out_joinable, uint64_t *out_info) {
  gdf_column *cols[1] = {col};
  KeyTable t;
  GDF_TRY(make_key_table(cols, 1, &t));
  KeyPlan plan = plan_keys(t);
  PartGeom g{};
  g.dbg = (int)lab::knob_int("GDF_JK_SDBG", 0);
  g.fb = fb;
  g.b1 = fb <= 8 ? fb : (fb + 1) / 2;
  if (lab::knob_on("GDF_JK_B1") && fb > 8) { const int b1 = (int)lab::knob_int("GDF_JK_B1", 0); if (b1 >= fb - 8 && b1 <= 8) g.b1 = b1; }   // experiment switch
  g.b2 = fb - g.b1;
  SideBufs sb;
  GDF_TRY(partition_side(t, plan, g, &sb, !plan.narrow && plan.mode == KM_RAW_INT && t.col[0].width == 8));
  if (sb.joinable) {
    const Tuples f = sb.final();
    if (plan.narrow) {
      hipLaunchKernelGGL(jk_unpack_narrow, dim3(small_grid(sb.joinable)), dim3(256), 0, stream0(), f.w, sb.joinable, out_key, out_idx);
      HIP_CHECK_LAST();
      HIP_TRY(hipStreamSynchronize(stream0()));
    } else {
      HIP_TRY(hipMemcpy(out_key, f.w, sizeof(uint64_t) * sb.joinable, hipMemcpyDeviceToDevice));
      HIP_TRY(hipMemcpy(out_idx, f.idx, sizeof(int32_t) * sb.joinable, hipMemcpyDeviceToDevice));
    }
  }
  for (size_t f = 0; f < sb.fine_off.size(); ++f) out_fine_off[f] = sb.fine_off[f];
  *out_joinable = sb.joinable;
  out_info[0] = (uint64_t)plan.narrow;
  out_info[1] = plan.kmin;
  return GDF_SUCCESS;
}

// ---------------------------------------------------------------------------
// gdf_amd_join_build_* (include/gdf/gdf_amd_ext.h): partition the build relation once, probe it many times.
// The multi-GPU join receives the probe relation in slices while the build relation is already complete.
// ---------------------------------------------------------------------------
struct PreparedBuild {
  int ncols = 0;
  gdf_column cols[MAX_KEY_COLS];        // copies of the caller's structs; the DATA stays the caller's and must outlive this
  gdf_column *colp[MAX_KEY_COLS];
  KeyTable table;
  BuildSide side;
  bool partitioned = false;             // false: empty build relation, probes take the generic entry point
  bool fj = false;                      // made by gdf_amd_fj_build_create from a receive buffer: there are no key columns to re-read
};

static gdf_error build_create(gdf_column **build_cols, int num_cols, PreparedBuild **out) {
  GDF_REQUIRE(build_cols && out && num_cols > 0, GDF_DATASET_EMPTY);
  GDF_REQUIRE(num_cols <= MAX_KEY_COLS, GDF_JOIN_TOO_MANY_COLUMNS);
  for (int i = 0; i < num_cols; ++i) {
    GDF_REQUIRE(build_cols[i], GDF_DATASET_EMPTY);
    GDF_REQUIRE(build_cols[i]->size == build_cols[0]->size, GDF_COLUMN_SIZE_MISMATCH);
    GDF_REQUIRE(build_cols[i]->size == 0 || build_cols[i]->data, GDF_DATASET_EMPTY);
  }
  GDF_REQUIRE(build_cols[0]->size < (size_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  std::unique_ptr<PreparedBuild> pb(new PreparedBuild());
  pb->ncols = num_cols;
  for (int i = 0; i < num_cols; ++i) { pb->cols[i] = *build_cols[i]; pb->colp[i] = &pb->cols[i]; }
  if (build_cols[0]->size > 0) {
    GDF_TRY(make_key_table(pb->colp, num_cols, &pb->table));
    GDF_TRY(prepare_build(pb->table, &pb->side));
    HIP_TRY(hipStreamSynchronize(stream0()));
    pb->partitioned = true;
  }
  *out = pb.release();
  return GDF_SUCCESS;
}

static gdf_error build_probe(PreparedBuild *pb, int left_join, gdf_column **probe_cols, int num_cols, gdf_column *probe_indices,
                             gdf_column *build_indices) {
  GDF_REQUIRE(pb && probe_cols && probe_indices && build_indices, GDF_DATASET_EMPTY);
  GDF_REQUIRE(!pb->fj, GDF_INVALID_API_CALL);
  GDF_REQUIRE(num_cols == pb->ncols, GDF_JOIN_DTYPE_MISMATCH);
  const JoinKind kind = left_join ? JOIN_LEFT : JOIN_INNER;
  const size_t probe_size = probe_cols[0] ? probe_cols[0]->size : 0;
  if (!pb->partitioned || probe_size == 0) {      // an empty side: nothing to reuse, same results as gdf_{inner,left}_join
    gdf_context ctx{0, GDF_HASH, 0, 0, 0};
    return join_call(kind, num_cols, probe_cols, pb->colp, probe_indices, build_indices, &ctx);
  }
  GDF_REQUIRE(probe_size < (size_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  for (int i = 0; i < num_cols; ++i) {
    GDF_REQUIRE(probe_cols[i] && probe_cols[i]->data, GDF_DATASET_EMPTY);
    GDF_REQUIRE(probe_cols[i]->dtype == pb->cols[i].dtype, GDF_JOIN_DTYPE_MISMATCH);
    GDF_REQUIRE(probe_cols[i]->size == probe_size, GDF_COLUMN_SIZE_MISMATCH);
  }
  KeyTable pt;
  GDF_TRY(make_key_table(probe_cols, num_cols, &pt));
  gdf_column_view(probe_indices, nullptr, nullptr, 0, N_GDF_TYPES);
  gdf_column_view(build_indices, nullptr, nullptr, 0, N_GDF_TYPES);
  StageClock clk((lab::knob_int("GDF_JK_DBG", 0) & 512) != 0);
  int32_t *o_probe = nullptr, *o_build = nullptr;
  int64_t n = 0;
  gdf_error e = probe_prepared(pt, pb->table, pb->side, kind, &o_probe, &o_build, &n, clk);
  if (e == GDF_AMD_RETRY_WITHOUT_LEVEL3) {           // skewed probe keys against a three-level build side: rebuild it plainly, once
    BuildSide plain;
    GDF_TRY(prepare_build(pb->table, &plain, true));
    e = probe_prepared(pt, pb->table, plain, kind, &o_probe, &o_build, &n, clk);
  }
  GDF_TRY(e);
  if (n == 0) return GDF_SUCCESS;
  gdf_column_view(probe_indices, o_probe, nullptr, (gdf_size_type)n, GDF_INT32);
  gdf_column_view(build_indices, o_build, nullptr, (gdf_size_type)n, GDF_INT32);
  return GDF_SUCCESS;
}

// ---------------------------------------------------------------------------
// gdf_amd_join_probe_* (include/gdf/gdf_amd_ext.h): a probe relation that arrives in slices is partitioned slice by
// slice into ONE set of fine partitions and probed once.  (Probing every slice on its own re-inserts the LDS tables of
// all build partitions per slice: 0.65 ms per slice at C4's shard sizes.)  INNER joins on plain NARROW keys with a
// two-level build side only; everything else reports GDF_UNSUPPORTED_METHOD and the caller probes slice by slice.
// ---------------------------------------------------------------------------
struct ProbeAccum {
  PreparedBuild *pb = nullptr;
  SideBufs P;
  SpecAppend app;
  double dup = 1.0;
  bool failed = false;
  std::deque<DevBuf> keep;              // fj_probe_add: segment maps the queued level-2 kernels still read
};

static gdf_error accum_begin(PreparedBuild *pb, size_t expected_rows, ProbeAccum **out) {
  GDF_REQUIRE(pb && out, GDF_DATASET_EMPTY);
  place_budget_begin();
  const KeyPlan &plan = pb->side.plan;
  const PartGeom &g = pb->side.g;
  uint32_t largest_build = 0;
  for (uint32_t c : pb->side.B.fine_cnt) largest_build = std::max(largest_build, c);
  if (!pb->partitioned || plan.verify || !plan.narrow || g.b2 == 0 || g.b3 != 0 || largest_build > (uint32_t)JK_MAX_BUILD ||
      (!pb->fj && (expected_rows < ((size_t)1 << 22) || lab::knob_on("GDF_JK_NO_ACCUM"))))
    return GDF_UNSUPPORTED_METHOD;
  std::unique_ptr<ProbeAccum> a(new ProbeAccum());
  a->pb = pb;
  a->dup = std::max(1.0, (double)expected_rows / std::max<uint32_t>(pb->side.B.joinable, 1));
  const double mean = (double)expected_rows / (double)(1u << g.fb);
  // room per fine partition for the EXPECTED total plus 3 % (the slices' sizes are the senders' business) and the usual spread
  a->app.cap2 = (uint32_t)(((uint64_t)(mean * 1.03 + 8.0 * std::sqrt(mean * (1.0 + a->dup)) + 64.0) + 7) / 8 * 8);
  if ((uint64_t)(1u << g.fb) * a->app.cap2 + 16384 >= 0x7fffffffULL) return GDF_UNSUPPORTED_METHOD;
  *out = a.release();
  return GDF_SUCCESS;
}

static gdf_error accum_add(ProbeAccum *a, gdf_column **probe_cols, int num_cols) {
  GDF_REQUIRE(a && probe_cols, GDF_DATASET_EMPTY);
  GDF_REQUIRE(!a->pb->fj, GDF_INVALID_API_CALL);                      // a build side made from a receive buffer takes gdf_amd_fj_probe_add
  GDF_REQUIRE(num_cols == a->pb->ncols, GDF_JOIN_DTYPE_MISMATCH);
  if (a->failed) return GDF_UNSUPPORTED_METHOD;
  const size_t n = probe_cols[0] ? probe_cols[0]->size : 0;
  if (n == 0) return GDF_SUCCESS;
  for (int i = 0; i < num_cols; ++i) {
    GDF_REQUIRE(probe_cols[i] && probe_cols[i]->data, GDF_DATASET_EMPTY);
    GDF_REQUIRE(probe_cols[i]->dtype == a->pb->cols[i].dtype, GDF_JOIN_DTYPE_MISMATCH);
    GDF_REQUIRE(probe_cols[i]->size == n, GDF_COLUMN_SIZE_MISMATCH);
    GDF_REQUIRE(!probe_cols[i]->valid, GDF_VALIDITY_UNSUPPORTED);
  }
  GDF_REQUIRE((uint64_t)a->app.rows + n < (uint64_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  KeyTable pt;
  GDF_TRY(make_key_table(probe_cols, num_cols, &pt));
  bool ok = false;
  GDF_TRY(partition_side_spec(pt, a->pb->side.plan, a->pb->side.g, a->dup, &a->P, &ok, &a->app));
  if (!ok) { a->failed = true; return GDF_UNSUPPORTED_METHOD; }     // a partition outgrew its room: probe the slices one by one
  return GDF_SUCCESS;
}

static gdf_error accum_finish(ProbeAccum *a, gdf_column *probe_indices, gdf_column *build_indices) {
  std::unique_ptr<ProbeAccum> own(a);
  GDF_REQUIRE(a && probe_indices && build_indices, GDF_DATASET_EMPTY);
  if (a->failed) return GDF_UNSUPPORTED_METHOD;
  gdf_column_view(probe_indices, nullptr, nullptr, 0, N_GDF_TYPES);
  gdf_column_view(build_indices, nullptr, nullptr, 0, N_GDF_TYPES);
  if (!a->app.started) return GDF_SUCCESS;                            // no rows were added
  // A fused receiver's probe side goes the DEFERRED way of the single-GPU main path (round 5): its fill cursors and their overflow flag
  // are what jk_make_units reads -- units, output offsets and the sample are made on the device and ONE state block comes back, instead of
  // the flag, then 2^15 cursors, then a unit list walked and uploaded by the host (0.25 ms of idle GPU in front of the probe kernel)
  if (a->pb->fj && a->pb->side.B.d_cnt.p && a->pb->side.B.d_begin.p && !lab::knob_on("GDF_JK_NO_DEFER")) {
    a->P.deferred = true;
    a->P.speculative = true;
    a->P.cap2 = a->app.cap2;
    a->P.nseg = 0;
    a->P.final_buf = 1;
    a->P.d_cursor.reset();
    a->P.d_cursor.p = a->app.cursor.release();
    KeyTable all = a->pb->table;
    all.nrows = a->app.rows;
    for (int c = 0; c < all.ncols; ++c) { all.col[c].data = nullptr; all.col[c].valid = nullptr; }
    all.any_valid = 0;
    StageClock clk(false);
    int32_t *o_probe = nullptr, *o_build = nullptr;
    int64_t n = 0;
    const gdf_error e = probe_partitioned(all, a->pb->table, a->pb->side, a->pb->side.plan, a->P, JOIN_INNER, &o_probe, &o_build, &n, clk);
    if (e == GDF_AMD_RETRY_EXACT_PROBE) return GDF_UNSUPPORTED_METHOD;          // a fine partition outgrew its room (the flag, as below)
    GDF_TRY(e);
    if (n == 0) return GDF_SUCCESS;
    gdf_column_view(probe_indices, o_probe, nullptr, (gdf_size_type)n, GDF_INT32);
    gdf_column_view(build_indices, o_build, nullptr, (gdf_size_type)n, GDF_INT32);
    return GDF_SUCCESS;
  }
  if (a->pb->fj) {                  // the level-2 passes of gdf_amd_fj_probe_add were only queued: their overflow flag is looked at now
    uint32_t flag = 0;
    HIP_TRY(read_back(&flag, a->app.cursor.as<uint32_t>() + (1u << a->pb->side.g.fb), sizeof(flag)));
    a->keep.clear();
    if (flag) return GDF_UNSUPPORTED_METHOD;
  }
  GDF_TRY(spec_append_finish(a->pb->side.g, &a->app, &a->P));
  KeyTable all = a->pb->table;            // stands for the accumulated relation: same key columns, no data (never read on this path)
  all.nrows = a->app.rows;
  for (int c = 0; c < all.ncols; ++c) { all.col[c].data = nullptr; all.col[c].valid = nullptr; }
  all.any_valid = 0;
  StageClock clk(false);
  int32_t *o_probe = nullptr, *o_build = nullptr;
  int64_t n = 0;
  GDF_TRY(probe_partitioned(all, a->pb->table, a->pb->side, a->pb->side.plan, a->P, JOIN_INNER, &o_probe, &o_build, &n, clk));
  if (n == 0) return GDF_SUCCESS;
  gdf_column_view(probe_indices, o_probe, nullptr, (gdf_size_type)n, GDF_INT32);
  gdf_column_view(build_indices, o_build, nullptr, (gdf_size_type)n, GDF_INT32);
  return GDF_SUCCESS;
}


// ---------------------------------------------------------------------------
// FUSED multi-GPU join (gdf_amd_fj_*, include/gdf/gdf_amd_ext.h): the SENDER runs the join's level-1 regroup, the receiver
// continues at level 2.
//
// All ranks share one hash space: h = hash_a(key); rank = mulhi(h, world) owns the key; inside a rank the partition ids come
// from the low word of h * world (uniform again).  The sender regroups its rows into world << c1 bins (bin = rank << c1 |
// coarse partition on that rank; at most 1024), writing the NARROWED 4-byte keys into a send buffer laid out as regions of
// `cap` keys per (bin, XCD) -- so everything for rank r is one contiguous, fixed-size block and no count exchange precedes
// the data -- and, per input row, the POSITION its key went to (out_pos, in row order; never travels).  The receiver's buffer
// (world blocks, sender-major) is exactly a level-1 output in the speculative layout with world << (c1 + 3) segments: the
// level-2 regroup reads its 4-byte keys, numbers the tuples by their position in that buffer and drops them into the fine
// partitions the LDS probe works on.  Per row: sender 8 B in + 4 + 4 B out, receiver 4 B in + 8 B out, probe 8 + 8 -- against
// 8 + 8 + 4.125 (stable split, two passes) and 4 + 8, 8 + 8, 8 + 8 for the key-only shuffle of round 1 -- and 4 B on the links.
// Global row ids: a result index is a position in a receive buffer, i.e. (sender, region, offset); the sender knows which
// row it put there (out_pos inverted; libgdf_amd/multigpu.py resolves them lazily, with a second exchange outside the timed path).
// ---------------------------------------------------------------------------
constexpr int FJ_THREADS = 1024;
constexpr int FJ_ITEMS = 32;          // 32768-key tiles: a (tile, bin) run is 32 keys = 128 bytes at 1024 bins
constexpr int FJ_ROUND = 8;           // keys loaded and ranked per round (64-bit raw keys cost two registers each: sixteen per round spilled)
constexpr int FJ_TILE = FJ_THREADS * FJ_ITEMS;
constexpr int FJ_MAX_BINS = 1024;
constexpr int64_t FJ_CHUNK = ((int64_t)131072 + FJ_TILE - 1) / FJ_TILE * FJ_TILE;     // rows per sender workgroup

struct FjSend {
  const void *keys;          // int64 or int32 column
  uint32_t n;
  long long lo;              // keys travel as (key - lo); a key outside [lo, lo + span] joins nothing and is dropped
  unsigned long long span;
  uint32_t world;
  int c1;                    // coarse bits per rank: bins = world << c1
  uint32_t cap;              // keys per (bin, XCD) region
  uint32_t chunk;            // rows per workgroup (a multiple of the tile)
  uint32_t *out_keys;
  uint32_t *out_pos;         // [n] where row i's key went in out_keys (0xffffffff: nowhere)
  uint32_t *fill;            // [bins * 8] zero-initialised fill counters
  uint32_t *overflow;        // set when a region outgrew `cap` (its run goes to the dump area behind the regions)
};

// The first version kept (key, row) pairs in the LDS tile and scattered BOTH as 4-byte arrays: two 64-byte runs per
// (tile, bin), 5.05 ms per 1.125e9 rows (3.7 TB/s) -- the store-REQUEST wall of profiles/r2_a_single_pass_ablation.md.  Row
// numbers do not need the scatter: a row's DESTINATION is known when its rank is, and goes out in row order as a plain
// streaming store (out_pos); whoever wants "which row sits at position p" inverts that array later (global ids are
// resolved lazily, outside the timed path).  The tile then holds 4-byte keys only: twice the keys per tile, 128-byte runs,
// and half the scattered arrays.
// base + a 32-bit BYTE offset (element index below 2^32 / sizeof(T)): the access is emitted with the uniform base in scalar
// registers and one vector register of offset, instead of a 64-bit vector address
template <class T>
__device__ __forceinline__ T *at32(T *base, uint32_t index) {
  using Byte = typename std::conditional<std::is_const<T>::value, const char, char>::type;
  return reinterpret_cast<T *>(reinterpret_cast<Byte *>(base) + (uint32_t)(index * (uint32_t)sizeof(T)));
}

// POW2: the world is a power of two 2^k with k + c1 >= 1 -- rank = mulhi(h, world) is then h's top k bits and the coarse id the next
// c1, i.e. the bin is ONE shift of the hash instead of a 32 x 32 -> 64 multiply (two quarter-rate instructions; the kernel hashes
// every key three times and spends ~70 % of its time issuing VALU work, tools/kernel_blocks.py).
// (Round 5 shipped the HASH instead of the narrowed key where hash_a is a bijection of it -- no second hash at the receiver, no second and
// third one in this kernel's flush -- and gained nothing: fused local passes 13.4 - 13.9 against 13.3 - 13.4 ms, the receiver's level 2
// 3.41 against 3.45, profiles/r5_g_sim_c4_fused_{hash,key}_travels.txt.  Once more the VALU work is hidden; what the receiver's level 2
// pays for is its 256 bins -- runs of 16 six-byte tuples.  Removed again.)
template <class K, bool POW2>
__global__ __launch_bounds__(FJ_THREADS) void fj_scatter(FjSend a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fj_lds[];
  uint32_t *tk = reinterpret_cast<uint32_t *>(fj_lds);                  // [TILE + 4]: narrowed keys regrouped by bin; [TILE] = trash slot
  uint32_t *hist = tk + FJ_TILE + 4;                                    // [MAX_BINS + 4] counts, then exclusive starts; [MAX_BINS] = trash bin
  uint32_t *gbase = hist + FJ_MAX_BINS + 4;                             // [MAX_BINS + 4] destination of LDS position 0 of a bin
  uint32_t *wave_tot = gbase + FJ_MAX_BINS + 4;                         // [THREADS / WAVE]
  const uint32_t nbins = a.world << a.c1;
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t begin = blockIdx.x * a.chunk;
  const uint32_t end = begin + a.chunk < a.n ? begin + a.chunk : a.n;
  const uint32_t dump = (nbins << 3) * a.cap;
  const K *col = (const K *)a.keys;
  // hash_a(key32 + lo) without 64-bit arithmetic: key_fold is `low word ^ high word * C`, and the high word of key32 + lo is lo's own
  // or one more (the carry) -- two constants to choose from instead of an add-with-carry and a quarter-rate multiply per hash
  const uint32_t lo_word = (uint32_t)(unsigned long long)a.lo;
  const uint32_t fold0 = (uint32_t)((unsigned long long)a.lo >> 32) * 0x9e3779b1u, fold1 = fold0 + 0x9e3779b1u;
  const uint32_t pow2_shift = 32u - ((uint32_t)a.c1 + (uint32_t)(31 - __clz((int)a.world)));
  auto bin_of_key = [&](uint32_t key32) -> uint32_t {
    const uint32_t low = key32 + lo_word;
    const uint32_t h = lowbias32(low ^ (low < key32 ? fold1 : fold0));
    if constexpr (POW2) return h >> pow2_shift;
    const uint64_t u = (uint64_t)h * a.world;
    return ((uint32_t)(u >> 32) << a.c1) | (uint32_t)((uint64_t)(uint32_t)u >> (32 - a.c1));
  };
  for (uint32_t b = threadIdx.x; b < FJ_MAX_BINS + 4; b += FJ_THREADS) hist[b] = 0;
  block_sync();
  // Every address is the tile's (uniform) base plus a 32-bit offset below 2^18: one register per access instead of a 64-bit
  // pointer each (with clamped 64-bit addresses for 32 loads and 32 stores the kernel spilled 380 bytes per lane).
  // The first round of the NEXT tile is requested before this tile's flush (pre[]): its latency hides behind the stores,
  // the one workgroup of a CU has nothing else to overlap it with.
  K pre[FJ_ROUND];
  // (UNCONDITIONAL: behind `if (there is a next tile)` pre[] stays live across the whole tile body -- the skipped case
  // carries the old values around the loop -- and the allocator spills it, i.e. waits for each load right after issuing it.
  // Without a next tile every lane re-reads the chunk's first key: one cache line.)
  auto prefetch = [&](uint32_t tile) {
    const uint32_t tid = opaque_tid();
    const bool real = tile < end;
    const K *base = col + (real ? tile : begin);
    const uint32_t live = !real ? 1u : (end - tile < (uint32_t)FJ_TILE ? end - tile : (uint32_t)FJ_TILE);
#pragma unroll
    for (int k = 0; k < FJ_ROUND; ++k) {
      const uint32_t o = k * FJ_THREADS + tid;
      pre[k] = __builtin_nontemporal_load(at32(base, o < live ? o : live - 1));
    }
  };
  if (begin >= end) return;
  prefetch(begin);
  for (uint32_t tile = begin; tile < end; tile += FJ_TILE) {
    constexpr bool FULL = false;                      // (a separate instantiation for full tiles made the allocator spill pre[])
    const K *tcol = col + tile;
    uint32_t *tpos = a.out_pos + tile;
    const uint32_t live = FULL ? (uint32_t)FJ_TILE : end - tile;
    // per item: the narrowed key (32 registers), its rank within (tile, bin) as 16 bits (16 registers) and one bit "travels";
    // the bin is hashed again where it is needed -- kept next to key and rank it is 32 more registers and the kernel spills
    uint32_t key[FJ_ITEMS], rk[FJ_ITEMS / 2], okmask = 0;
#pragma unroll
    for (int h = 0; h < FJ_ITEMS; h += FJ_ROUND) {
      const uint32_t tid = opaque_tid();
      K raw[FJ_ROUND];
#pragma unroll
      for (int k = 0; k < FJ_ROUND; ++k) {             // all loads of the round first, unconditional (round 0: prefetched)
        const uint32_t o = (h + k) * FJ_THREADS + tid;
        if (h == 0) raw[k] = pre[k];
        else raw[k] = __builtin_nontemporal_load(at32(tcol, FULL || o < live ? o : live - 1));
      }
      uint32_t bin[FJ_ROUND];
#pragma unroll
      for (int q = 0; q < FJ_ROUND; q += 4) {          // four hashes at a time (interleaved by the dozen they spill)
#pragma unroll
        for (int k = q; k < q + 4; ++k) {
          const uint32_t o = (h + k) * FJ_THREADS + tid;
          const unsigned long long off = (unsigned long long)((long long)raw[k] - a.lo);
          const bool travels = (FULL || o < live) && off <= a.span;
          key[h + k] = (uint32_t)off;
          okmask |= (uint32_t)travels << (h + k);
          const uint32_t b = bin_of_key(key[h + k]);                 // hashed whether it travels or not: no branch
          bin[k] = travels ? b : (uint32_t)FJ_MAX_BINS;              // MAX_BINS: the trash counter
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int k = 0; k < FJ_ROUND; ++k) bin[k] = atomicAdd(&hist[bin[k]], 1u);      // a round in flight, one wait
#pragma unroll
      for (int k = 0; k < FJ_ROUND; k += 2) {
        rk[(h + k) / 2] = bin[k] | (bin[k + 1] << 16);               // a rank is below 32768
        asm volatile("" : "+v"(rk[(h + k) / 2]));                    // packed HERE, not after the next round's loads
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    block_sync();
    {   // claim the runs, exclusive scan of the counts (thread t owns bin t)
      const uint32_t tid = opaque_tid();
      const uint32_t cnt = tid < nbins ? hist[tid] : 0;
      uint32_t gb = 0;
      if (cnt) {
        const uint32_t region = (tid << 3) | xcd;
        const uint32_t at = atomicAdd(&a.fill[region], cnt);
        if (at + cnt > a.cap) { atomicExch(a.overflow, 1u); gb = dump; }
        else gb = region * a.cap + at;
      }
      const uint32_t incl = wave_scan_incl(cnt);
      if (lane_id() == WAVE - 1) wave_tot[tid / WAVE] = incl;
      block_sync();
      const uint32_t start = incl - cnt + waves_before_sum<FJ_THREADS / WAVE>(wave_tot, tid);
      hist[tid] = start;
      gbase[tid] = gb - start;
    }
    block_sync();
    uint32_t total = 0;
    for (int w = 0; w < FJ_THREADS / WAVE; ++w) total += wave_tot[w];
    // regroup in LDS; the row's destination leaves right away, in row order (coalesced, nobody re-reads it soon)
#pragma unroll
    for (int h = 0; h < FJ_ITEMS; h += 8) {
      const uint32_t tid = opaque_tid();
      uint32_t st[8], gbv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t b = bin_of_key(key[h + k]) & (FJ_MAX_BINS - 1);
        st[k] = hist[b];
        gbv[k] = gbase[b];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t o = (h + k) * FJ_THREADS + tid;
        const bool travels = (okmask >> (h + k)) & 1u;
        const uint32_t rank = (rk[(h + k) / 2] >> (16 * (k & 1))) & 0xffffu;
        const uint32_t pos = travels ? st[k] + rank : (uint32_t)FJ_TILE;
        tk[pos] = key[h + k];
        if (FULL || o < live) __builtin_nontemporal_store(travels ? pos + gbv[k] : 0xffffffffu, at32(tpos, o));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    block_sync();
    // the starts are used up: clear the counters for the next tile's ranking here instead of behind one more barrier pair
    for (uint32_t b = threadIdx.x; b < FJ_MAX_BINS + 4; b += FJ_THREADS) hist[b] = 0;
    prefetch(tile + FJ_TILE);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < FJ_ITEMS; h += 8) {              // unconditional stores: dead slots go to this thread's dump slot
      const uint32_t tid = opaque_tid();
      uint32_t w[8], gbv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) w[k] = tk[tid + (h + k) * FJ_THREADS];
#pragma unroll
      for (int k = 0; k < 8; ++k) gbv[k] = gbase[bin_of_key(w[k]) & (FJ_MAX_BINS - 1)];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t j = tid + (h + k) * FJ_THREADS;
        a.out_keys[j < total ? gbv[k] + j : dump + tid] = w[k];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    block_sync();                                        // the tile and gbase are free again
  }
}
static constexpr size_t fj_scatter_lds() { return 4 * (size_t)(FJ_TILE + 4) + 4 * (size_t)(2 * (FJ_MAX_BINS + 4) + FJ_THREADS / WAVE) + 16; }

// layout agreed by all ranks from global numbers only
static gdf_error fj_plan(int world, int64_t build_rows_total, int64_t rows_max, double dup, int *fine_bits, int *coarse_bits, uint32_t *cap) {
  GDF_REQUIRE(world >= 1 && fine_bits && coarse_bits && cap, GDF_INVALID_API_CALL);
  const PartGeom g = choose_geometry((build_rows_total + world - 1) / world);
  if (g.b3 != 0) return GDF_UNSUPPORTED_METHOD;                       // a rank's share needs a third level: not on this path
  int cmax = 0;
  while ((world << (cmax + 1)) <= FJ_MAX_BINS && cmax + 1 <= 8) ++cmax;
  const int cmin = g.fb > 8 ? g.fb - 8 : 0;                            // the receiver's level 2 is at most 256-way
  if (cmin > cmax || (world << cmax) > FJ_MAX_BINS) return GDF_UNSUPPORTED_METHOD;
  int c1 = g.b1 < cmin ? cmin : (g.b1 > cmax ? cmax : g.b1);
  if (c1 > g.fb - 1) c1 = g.fb - 1;                                    // the receiver's level 2 needs at least one bit
  if (c1 < cmin || c1 < 0) return GDF_UNSUPPORTED_METHOD;
  *fine_bits = g.fb;
  *coarse_bits = c1;
  // rows per (bin, XCD) region: workgroup b fills the regions of XCD b % 8, so a call with few workgroups spreads its rows
  // over fewer than eight regions per bin (one workgroup: a single one)
  const double nwg = std::ceil((double)(rows_max > 0 ? rows_max : 1) / (double)FJ_CHUNK);
  const double per_xcd = std::max((double)rows_max / 8.0 * (1.0 + 8.0 / nwg), (double)std::min<int64_t>(rows_max, FJ_CHUNK));
  const double mean = per_xcd / (double)((uint64_t)world << c1);
  // + 6 standard deviations; `dup` = expected rows per distinct key (all copies of a key land in one region, so the
  // variance grows with the multiplicity -- same term as the join's own speculative layout).  The slack travels: 6 sigma
  // at C4's ten probe rows per key is +10 % on the links
  if (!(dup >= 1.0)) dup = 1.0;
  *cap = (uint32_t)(((uint64_t)(mean + 6.0 * std::sqrt(mean * (1.0 + dup)) + 64.0) + 63) / 64 * 64);
  if ((uint64_t)(((uint64_t)world << c1) * 8) * *cap + FJ_TILE >= 0x7fffffffULL) return GDF_UNSUPPORTED_METHOD;
  return GDF_SUCCESS;
}

static gdf_error fj_send(gdf_column *keys, int64_t lo, int64_t hi, int world, int coarse_bits, uint32_t cap,
                         uint32_t *out_keys, uint32_t *out_pos, uint32_t *out_fill, int *overflowed) {
  GDF_REQUIRE(keys && out_keys && out_pos && out_fill && overflowed, GDF_DATASET_EMPTY);
  GDF_REQUIRE(!keys->valid, GDF_VALIDITY_UNSUPPORTED);
  const ElemKind kind = elem_kind(keys->dtype);
  GDF_REQUIRE(kind == K_I64 || kind == K_I32, GDF_UNSUPPORTED_DTYPE);
  GDF_REQUIRE(world >= 1 && coarse_bits >= 0 && ((uint64_t)world << coarse_bits) <= (uint64_t)FJ_MAX_BINS, GDF_INVALID_API_CALL);
  GDF_REQUIRE(hi >= lo && (uint64_t)hi - (uint64_t)lo < 0xffffffffULL, GDF_INVALID_API_CALL);
  GDF_REQUIRE(keys->size < (size_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  const uint32_t nregions = ((uint32_t)world << coarse_bits) << 3;
  // the kernel computes region offsets in 32 bits: what gdf_amd_fj_plan guarantees is checked again for a caller's own numbers
  GDF_REQUIRE(cap >= 64 && cap % 64 == 0 && (uint64_t)nregions * cap + FJ_TILE < 0x7fffffffULL, GDF_INVALID_API_CALL);
  HIP_TRY(hipMemsetAsync(out_fill, 0, sizeof(uint32_t) * ((size_t)nregions + 1), stream0()));      // [nregions]: the overflow flag
  *overflowed = 0;
  if (keys->size == 0) { HIP_TRY(hipStreamSynchronize(stream0())); return GDF_SUCCESS; }
  GDF_REQUIRE(keys->data, GDF_DATASET_EMPTY);
  FjSend a{};
  a.keys = keys->data;
  a.n = (uint32_t)keys->size;
  a.lo = lo;
  a.span = (unsigned long long)((uint64_t)hi - (uint64_t)lo);
  a.world = (uint32_t)world;
  a.c1 = coarse_bits;
  a.cap = cap;
  a.chunk = (uint32_t)FJ_CHUNK;
  a.out_keys = out_keys;
  a.out_pos = out_pos;
  a.fill = out_fill;
  a.overflow = out_fill + nregions;
  const unsigned grid = (unsigned)(((uint64_t)a.n + a.chunk - 1) / a.chunk);
  const size_t lds = fj_scatter_lds();
  const bool pow2 = (world & (world - 1)) == 0 && (world > 1 || coarse_bits > 0) && !lab::path_on("GDF_FJ_NO_POW2");
  auto launch = [&](auto kernel) -> gdf_error {
    HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    GDF_LAUNCH("fj_scatter", kernel, dim3(grid), dim3(FJ_THREADS), lds, stream0(), a);
    return GDF_SUCCESS;
  };
  if (kind == K_I64) GDF_TRY(pow2 ? launch(fj_scatter<long long, true>) : launch(fj_scatter<long long, false>));
  else GDF_TRY(pow2 ? launch(fj_scatter<int, true>) : launch(fj_scatter<int, false>));
  HIP_CHECK_LAST();
  uint32_t flag = 0;
  HIP_TRY(read_back(&flag, out_fill + nregions, sizeof(flag)));
  *overflowed = flag ? 1 : 0;
  return GDF_SUCCESS;
}

// one level-2 pass over a receive buffer: `keys` = world blocks of (1 << (c1 + 3)) regions of `cap` keys, `fill` (device) the
// matching fill counters, sender-major.  Appends into sb's level-2 buffer through `cursor` ([nfine + 2]: cursors | overflow flag
// | tile count).  Launches only; `keep` receives the scratch that must outlive the kernels.
static gdf_error fj_level2(const uint32_t *keys, const uint32_t *fill, uint32_t nseg, uint32_t cap, const PartGeom &g, uint32_t cap2,
                           int32_t row_base, uint32_t *cursor, Tuples out, std::deque<DevBuf> *keep, bool p6 = false, uint32_t calib_step = 0) {
  const uint32_t nfine = 1u << g.fb;
  // (world x coarse partitions = 1024 leaves 8 bits to level 2: 256 bins per 4096-tuple tile, (tile, bin) runs of 16 tuples.  8192-tuple
  // tiles -- 512 threads, runs of 32 as on the single-GPU path -- were slower: 4.15 vs 3.52 ms per 1e9 keys, profiles/r3_o_fused_level2.txt)
  const int sc2_threads = 256;
  const int64_t TILE2 = (int64_t)sc2_threads * JK_SC_ITEMS;
  keep->emplace_back();
  DevBuf &d_map = keep->back();
  RMM_TRY(d_map.alloc(sizeof(uint32_t) * (3 * (size_t)nseg + 2)));
  uint32_t *seg_begin = d_map.as<uint32_t>(), *seg_end = seg_begin + nseg, *tile_prefix = seg_end + nseg;
  uint32_t *ntiles_dev = cursor + nfine + 1;
  size_t map_lds = 0;
  GDF_TRY(l2map_prepare(nseg, &map_lds));
  hipLaunchKernelGGL(jk_make_l2map, dim3(1), dim3(JK_BK_THREADS), map_lds, stream0(), fill, nseg, cap, (uint32_t)TILE2, seg_begin, seg_end, tile_prefix,
                     ntiles_dev, g.world, 1u << g.b1);
  HIP_CHECK_LAST();
  PartGeom g2 = g;
  g2.cap1 = 0;
  g2.cap2 = cap2;
  g2.dump = nfine * cap2;
  g2.spec_flag = cursor + nfine;
  g2.row_base = row_base;
  Level2Map m{seg_begin, seg_end, tile_prefix, 0};
  m.ntiles_dev = ntiles_dev;
  m.nseg = nseg;
  m.keys32 = keys;
  m.calib_step = calib_step;          // (> 1: a calibration run of the level-2 buffer's placement tournament, fj_probe_add)
  const uint32_t tile_bound = (uint32_t)(((uint64_t)nseg * cap) / (uint64_t)TILE2) + nseg + 1;
  GDF_TRY(launch_scatter2(true, sc2_threads, tile_bound, g2, m, Tuples{nullptr, nullptr, nullptr}, cursor, out, p6));
  return GDF_SUCCESS;
}

static PartGeom fj_geometry(int world, int fine_bits, int coarse_bits, int64_t lo) {
  PartGeom g{};
  g.fb = fine_bits;
  g.b1 = coarse_bits;
  g.b2 = fine_bits - coarse_bits;
  g.kbias = (uint64_t)lo;
  g.world = (uint32_t)world;
  return g;
}

// the received build relation -> a PreparedBuild whose partitions live in the speculative layout
static gdf_error fj_build_create(const uint32_t *recv_keys, const uint32_t *recv_fill, int world, int64_t lo, int fine_bits, int coarse_bits,
                                 uint32_t cap, int64_t expected_rows, PreparedBuild **out) {
  GDF_REQUIRE(recv_keys && recv_fill && out, GDF_DATASET_EMPTY);
  GDF_REQUIRE(fine_bits >= 1 && fine_bits <= JK_MAX_FB && coarse_bits >= 0 && fine_bits - coarse_bits >= 1 && fine_bits - coarse_bits <= 8,
              GDF_INVALID_API_CALL);
  GDF_REQUIRE(world >= 1 && ((uint64_t)world << coarse_bits) <= (uint64_t)FJ_MAX_BINS && cap >= 64 && cap % 64 == 0 &&
              (((uint64_t)world << coarse_bits) << 3) * cap + FJ_TILE < 0x7fffffffULL, GDF_INVALID_API_CALL);      // positions are 31-bit (fj_send)
  std::unique_ptr<PreparedBuild> pb(new PreparedBuild());
  pb->ncols = 1;
  pb->fj = true;
  gdf_column_view(&pb->cols[0], nullptr, nullptr, 0, GDF_INT32);
  pb->colp[0] = &pb->cols[0];
  pb->table = KeyTable{};
  pb->table.ncols = 1;
  pb->table.col[0] = ColView{nullptr, nullptr, (int)K_I32, 4};
  BuildSide &bs = pb->side;
  bs.plan = KeyPlan{};
  bs.plan.mode = KM_RAW_INT;
  bs.plan.narrow = 1;
  bs.plan.kmin = (uint64_t)lo;
  bs.plan.klimit = 0x7ffffffeULL;
  bs.plan.kspan = 0x7ffffffeULL;          // narrowed keys are below 2^31 - 1 (gdf_amd_fj_send): the largest span the range can have
  bs.g = fj_geometry(world, fine_bits, coarse_bits, lo);
  const uint32_t nfine = 1u << fine_bits, nseg = ((uint32_t)world << coarse_bits) << 3;
  const double mean = (double)(expected_rows > 0 ? expected_rows : 1) / (double)nfine;
  const uint32_t cap2 = (uint32_t)(((uint64_t)(mean * 1.03 + 8.0 * std::sqrt(mean * 2.0) + 64.0) + 7) / 8 * 8);
  const uint64_t size2 = (uint64_t)nfine * cap2 + 16384;
  if (size2 >= 0x7fffffffULL) return GDF_UNSUPPORTED_METHOD;
  DevBuf cursor;
  RMM_TRY(cursor.alloc(sizeof(uint32_t) * ((size_t)nfine + 2)));
  hipLaunchKernelGGL(jk_init_cursor, dim3(32), dim3(256), 0, stream0(), cursor.as<uint32_t>(), nfine, cap2);
  HIP_TRY(hipMemsetAsync(cursor.as<uint32_t>() + nfine + 1, 0, sizeof(uint32_t), stream0()));
  RMM_TRY(bs.B.w[1].alloc(sizeof(uint64_t) * size2));
  std::deque<DevBuf> keep;
  GDF_TRY(fj_level2(recv_keys, recv_fill, nseg, cap, bs.g, cap2, 0, cursor.as<uint32_t>(), bs.B.tuples(1), &keep));
  std::vector<uint32_t> cur((size_t)nfine + 1);
  HIP_TRY(read_back(cur.data(), cursor.p, sizeof(uint32_t) * ((size_t)nfine + 1)));
  if (cur[nfine]) return GDF_UNSUPPORTED_METHOD;                      // a fine partition outgrew its room (skewed build keys)
  bs.B.final_buf = 1;
  bs.B.fine_begin.resize(nfine);
  bs.B.fine_cnt.resize(nfine);
  uint64_t total = 0;
  uint32_t largest = 0;
  for (uint32_t f = 0; f < nfine; ++f) {
    bs.B.fine_begin[f] = f * cap2;
    bs.B.fine_cnt[f] = cur[f] - f * cap2;
    total += bs.B.fine_cnt[f];
    largest = std::max(largest, bs.B.fine_cnt[f]);
  }
  if (largest > (uint32_t)JK_MAX_BUILD) return GDF_UNSUPPORTED_METHOD;  // would need the global-table path, which wants the exact layout
  bs.B.joinable = (uint32_t)total;
  bs.B.speculative = true;
  pb->table.nrows = (int64_t)total;
  RMM_TRY(bs.B.d_begin.alloc(sizeof(uint32_t) * nfine));
  RMM_TRY(bs.B.d_cnt.alloc(sizeof(uint32_t) * nfine));
  HIP_TRY(hipMemcpyAsync(bs.B.d_begin.p, bs.B.fine_begin.data(), sizeof(uint32_t) * nfine, hipMemcpyHostToDevice, stream0()));
  HIP_TRY(hipMemcpyAsync(bs.B.d_cnt.p, bs.B.fine_cnt.data(), sizeof(uint32_t) * nfine, hipMemcpyHostToDevice, stream0()));
  HIP_TRY(hipStreamSynchronize(stream0()));
  pb->partitioned = true;
  *out = pb.release();
  return GDF_SUCCESS;
}

static gdf_error fj_probe_add(ProbeAccum *a, const uint32_t *recv_keys, const uint32_t *recv_fill, uint32_t cap, int64_t position_base,
                              int64_t buffer_elems) {
  GDF_REQUIRE(a && recv_keys && recv_fill, GDF_DATASET_EMPTY);
  GDF_REQUIRE(a->pb->fj, GDF_INVALID_API_CALL);
  if (a->failed) return GDF_UNSUPPORTED_METHOD;
  GDF_REQUIRE(position_base >= 0 && position_base + buffer_elems < (int64_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  const PartGeom &g = a->pb->side.g;
  const uint32_t nfine = 1u << g.fb, nseg = (g.world << g.b1) << 3;
  // the receive buffer must be the one this build side's plan describes: nseg regions of cap keys
  GDF_REQUIRE(cap >= 64 && cap % 64 == 0 && (uint64_t)nseg * cap + FJ_TILE < 0x7fffffffULL && (uint64_t)buffer_elems >= (uint64_t)nseg * cap,
              GDF_INVALID_API_CALL);
  if (!a->app.started) {
    const uint64_t size2 = (uint64_t)nfine * a->app.cap2 + 16384;
    RMM_TRY(a->app.cursor.alloc(sizeof(uint32_t) * ((size_t)nfine + 2)));
    hipLaunchKernelGGL(jk_init_cursor, dim3(32), dim3(256), 0, stream0(), a->app.cursor.as<uint32_t>(), nfine, a->app.cap2);
    // six-byte level-2 tuples (p6_store) as on the single-GPU main path: 2^15 partitions per rank, narrowed keys whose raw values
    // (key + lo) cannot straddle a 2^32 boundary, and a world whose rank remap keeps the remainder injective (p6_low)
    const KeyPlan &plan = a->pb->side.plan;
    a->P.p6 = g.fb == JK_MAX_FB && g.b3 == 0 && (plan.kmin & 0xffffffffULL) + plan.kspan < (1ULL << 32) && p6_world_ok(g.world) &&
              !lab::path_on("GDF_JK_NO_P6");
    // the accumulator's level-2 buffer is a PLACED block like the single-GPU path's (partition_side_spec): every candidate is timed on
    // every fourth tile of this first slice, the cursors are set back, the pool keeps the fastest
    const size_t bytes2 = a->P.p6 ? 6 * size2 + 16 : sizeof(uint64_t) * size2;
    RMM_TRY(a->P.w[1].alloc_placed(JK_ROLE_FUSED_LEVEL2, bytes2, JK_PLACE_DRAWS_L2));
    for (int round = 0; round <= JK_PLACE_DRAWS_L2 && a->P.w[1].measure && !lab::knob_on("GDF_JK_NO_CALIBRATE"); ++round) {
      a->P.w[1].clock_begin(stream0());
      GDF_TRY(fj_level2(recv_keys, recv_fill, nseg, cap, g, a->app.cap2, (int32_t)position_base, a->app.cursor.as<uint32_t>(), a->P.tuples(1), &a->keep,
                        a->P.p6, 4));
      a->P.w[1].clock_end(stream0());
      hipLaunchKernelGGL(jk_init_cursor, dim3(32), dim3(256), 0, stream0(), a->app.cursor.as<uint32_t>(), nfine, a->app.cap2);
      HIP_CHECK_LAST();
      RMM_TRY(a->P.w[1].alloc_placed(JK_ROLE_FUSED_LEVEL2, bytes2, JK_PLACE_DRAWS_L2));
    }
    a->app.started = true;
  }
  GDF_TRY(fj_level2(recv_keys, recv_fill, nseg, cap, g, a->app.cap2, (int32_t)position_base, a->app.cursor.as<uint32_t>(), a->P.tuples(1), &a->keep,
                    a->P.p6));
  a->app.rows = position_base + buffer_elems;       // positions are numbered across the slices' receive buffers
  return GDF_SUCCESS;
}


// ---------------------------------------------------------------------------
// gdf_amd_dist_inner_join: the fused multi-GPU join end to end behind the C ABI (include/gdf/gdf_amd_ext.h).  The orchestration
// libgdf_amd/multigpu.py fused_inner_join used to do over torch.distributed, over a gdf_amd_transport instead: a host in any
// language gets the distributed join from libgdf.so alone.
// ---------------------------------------------------------------------------
namespace {
struct DistTicket { void *keys = nullptr, *fill = nullptr; };

struct DistExchange {          // one relation slice on the wire: send buffers (alive until the exchange is done), receive buffers
  DevBuf sk, sf, rk, rf;
  DistTicket t;
  bool posted_keys = false, posted_fill = false;      // per ticket: a failure between the two all_to_alls leaves ONE of them on the wire
};

// All ranks learn the worst of their `level`s: 0 fine, 1 "this shape does not fit, decline together", 2 "a rank hit a hard local
// error" (a rank that returned on a failed allocation left its peers blocked in the next collective -- RCCL has no
// timeout of its own; local errors are now carried to the next agreement and every rank leaves there).
static gdf_error dist_agree(gdf_amd_transport *tr, int level, int *worst) {
  int64_t v = level;
  if (tr->all_reduce_i64(tr->ctx, &v, 1, 1) != 0) return GDF_C_ERROR;
  *worst = (int)v;
  return GDF_SUCCESS;
}
static gdf_error dist_alloc(gdf_amd_transport *tr, DistExchange *x, size_t block_elems, size_t regions_per_rank) {
  const size_t world = (size_t)tr->world;
  RMM_TRY(x->sk.alloc(sizeof(uint32_t) * (world * block_elems + FJ_TILE)));
  RMM_TRY(x->sf.alloc(sizeof(uint32_t) * (world * regions_per_rank + 1)));
  RMM_TRY(x->rk.alloc(sizeof(uint32_t) * world * block_elems));
  RMM_TRY(x->rf.alloc(sizeof(uint32_t) * (world * regions_per_rank + 2)));
  return GDF_SUCCESS;
}
static gdf_error dist_post(gdf_amd_transport *tr, DistExchange *x, size_t block_elems, size_t regions_per_rank) {
  if (tr->all_to_all(tr->ctx, x->sk.p, x->rk.p, sizeof(uint32_t) * block_elems, &x->t.keys) != 0) return GDF_C_ERROR;
  x->posted_keys = true;
  if (tr->all_to_all(tr->ctx, x->sf.p, x->rf.p, sizeof(uint32_t) * regions_per_rank, &x->t.fill) != 0) return GDF_C_ERROR;
  x->posted_fill = true;
  return GDF_SUCCESS;
}
static gdf_error dist_wait(gdf_amd_transport *tr, DistExchange *x) {
  gdf_error e = GDF_SUCCESS;
  if (x->posted_keys) { x->posted_keys = false; if (tr->wait(tr->ctx, x->t.keys) != 0) e = GDF_C_ERROR; }
  if (x->posted_fill) { x->posted_fill = false; if (tr->wait(tr->ctx, x->t.fill) != 0) e = GDF_C_ERROR; }
  return e;
}
}  // namespace

static gdf_error dist_inner_join(gdf_column *probe_keys, gdf_column *build_keys, gdf_amd_transport *tr, int chunks, uint32_t *probe_pos,
                                 uint32_t *build_pos, gdf_column *probe_indices, gdf_column *build_indices, gdf_amd_dist_info *info,
                                 int *declined) {
  GDF_REQUIRE(probe_keys && build_keys && tr && probe_indices && build_indices && info && declined, GDF_DATASET_EMPTY);
  GDF_REQUIRE(tr->all_to_all && tr->wait && tr->all_reduce_i64 && tr->world >= 1 && tr->rank >= 0 && tr->rank < tr->world, GDF_INVALID_API_CALL);
  GDF_REQUIRE(!probe_keys->valid && !build_keys->valid, GDF_VALIDITY_UNSUPPORTED);
  GDF_REQUIRE(probe_keys->dtype == build_keys->dtype, GDF_JOIN_DTYPE_MISMATCH);
  const ElemKind kind = elem_kind(probe_keys->dtype);
  GDF_REQUIRE(kind == K_I64 || kind == K_I32, GDF_UNSUPPORTED_DTYPE);
  GDF_REQUIRE(probe_keys->size < (size_t)INT_MAX && build_keys->size < (size_t)INT_MAX, GDF_COLUMN_SIZE_TOO_BIG);
  GDF_REQUIRE((probe_keys->size == 0 || (probe_keys->data && probe_pos)) && (build_keys->size == 0 || (build_keys->data && build_pos)), GDF_DATASET_EMPTY);
  *declined = 1;
  *info = gdf_amd_dist_info{};
  gdf_column_view(probe_indices, nullptr, nullptr, 0, N_GDF_TYPES);
  gdf_column_view(build_indices, nullptr, nullptr, 0, N_GDF_TYPES);
  const int world = tr->world;
  const int64_t n_p = (int64_t)probe_keys->size, n_b = (int64_t)build_keys->size;

  // A hard LOCAL error (allocation, kernel launch, an internal inconsistency) is remembered, not returned: this rank keeps taking
  // part in every collective its peers will enter, says so at the next agreement, and all ranks leave there together -- this one
  // with its error, the others with GDF_C_ERROR.  Only a failing TRANSPORT returns at once (there is nothing left to agree over).
  gdf_error hard = GDF_SUCCESS;
  auto note = [&](gdf_error e) { if (e != GDF_SUCCESS && hard == GDF_SUCCESS) hard = e; return e; };
  auto leave = [&](int worst) -> gdf_error { return hard != GDF_SUCCESS ? hard : (worst >= 2 ? GDF_C_ERROR : GDF_SUCCESS); };

  // ---- numbers every rank must agree on: the global build-side key range, the largest and the total shard sizes ----
  long long mm[2] = {LLONG_MAX, LLONG_MIN};
  if (n_b) {
    KeyTable bt;
    gdf_column *bc = build_keys;
    if (note(make_key_table(&bc, 1, &bt)) == GDF_SUCCESS) note(key_ranges(bt, mm));
  }
  int64_t mins[4] = {mm[0] <= mm[1] ? (int64_t)mm[0] : INT64_MAX, mm[0] <= mm[1] ? -(int64_t)mm[1] : INT64_MAX, -n_p, -n_b};
  if (tr->all_reduce_i64(tr->ctx, mins, 4, 0) != 0) return GDF_C_ERROR;
  int64_t sums[3] = {n_p, n_b, hard != GDF_SUCCESS ? 1 : 0};
  if (tr->all_reduce_i64(tr->ctx, sums, 3, 2) != 0) return GDF_C_ERROR;
  if (sums[2] != 0) return leave(2);
  const int64_t lo = mins[0], hi = mins[1] == INT64_MAX ? INT64_MIN : -mins[1], p_max = -mins[2], b_max = -mins[3];
  const int64_t p_total = sums[0], b_total = sums[1];
  // (everything up to the first exchange is decided from these shared numbers: every rank takes the same exits)
  if (b_total == 0 || p_total == 0 || lo > hi || (uint64_t)hi - (uint64_t)lo >= (uint64_t)0x7ffffffe) return GDF_SUCCESS;
  if (chunks < 1) chunks = 1;
  if ((int64_t)chunks > p_max) chunks = (int)p_max;
  const int64_t step_max = (p_max + chunks - 1) / chunks;
  int fb_b = 0, cb_b = 0, fb_p = 0, cb_p = 0;
  uint32_t cap_b = 0, cap_p = 0;
  gdf_error e = fj_plan(world, b_total, b_max, 1.0, &fb_b, &cb_b, &cap_b);      // (a pure function of the shared numbers)
  if (e == GDF_UNSUPPORTED_METHOD) return GDF_SUCCESS;
  GDF_TRY(e);
  e = fj_plan(world, b_total, step_max, std::max(1.0, (double)p_total / (double)std::max<int64_t>(b_total, 1)), &fb_p, &cb_p, &cap_p);
  if (e == GDF_UNSUPPORTED_METHOD) return GDF_SUCCESS;
  GDF_TRY(e);
  const size_t rpr_b = (size_t)8 << cb_b, rpr_p = (size_t)8 << cb_p;
  const size_t block_b = rpr_b * cap_b, block_p = rpr_p * cap_p;
  // result positions are 31-bit and count the blocks' room and empty regions
  if ((uint64_t)chunks * world * block_p >= 0x7fffffffULL || (uint64_t)world * block_b >= 0x7fffffffULL) return GDF_SUCCESS;
  info->world = world; info->chunks = chunks; info->lo = lo; info->hi = hi;
  info->fine_bits_p = fb_p; info->coarse_bits_p = cb_p; info->cap_p = cap_p; info->block_p = (int64_t)block_p;
  info->fine_bits_b = fb_b; info->coarse_bits_b = cb_b; info->cap_b = cap_b; info->block_b = (int64_t)block_b;

  // a failure between a posted exchange and its wait must not leave the peers' blocks on the wire: settle what is posted first
  std::deque<DistExchange> wire;
  auto settle = [&]() { for (DistExchange &x : wire) (void)dist_wait(tr, &x); };
  struct Settle { decltype(settle) &f; ~Settle() { f(); } } settle_on_exit{settle};

  // ---- every buffer that will go on the wire is allocated BEFORE the first agreement: an out-of-memory rank says so there ----
  // (they all stay alive until the call ends anyway -- queued level-2 kernels read the receive buffers)
  wire.emplace_back();
  DistExchange &bx = wire.back();
  note(dist_alloc(tr, &bx, block_b, rpr_b));
  for (int c = 0; c < chunks && hard == GDF_SUCCESS; ++c) {
    wire.emplace_back();
    note(dist_alloc(tr, &wire.back(), block_p, rpr_p));
  }
  DevBuf dummy_pos, dummy_ppos;
  if (!build_pos && note(dummy_pos.alloc(sizeof(uint32_t)) == RMM_SUCCESS ? GDF_SUCCESS : GDF_MEMORYMANAGER_ERROR) == GDF_SUCCESS) build_pos = dummy_pos.as<uint32_t>();
  if (!probe_pos && note(dummy_ppos.alloc(sizeof(uint32_t)) == RMM_SUCCESS ? GDF_SUCCESS : GDF_MEMORYMANAGER_ERROR) == GDF_SUCCESS) probe_pos = dummy_ppos.as<uint32_t>();

  // ---- build relation ----
  int over = 0;
  if (hard == GDF_SUCCESS) note(fj_send(build_keys, lo, hi, world, cb_b, cap_b, bx.sk.as<uint32_t>(), build_pos, bx.sf.as<uint32_t>(), &over));
  int worst = 0;
  GDF_TRY(dist_agree(tr, hard != GDF_SUCCESS ? 2 : (over != 0 ? 1 : 0), &worst));
  if (worst) return leave(worst);          // a region overflowed somewhere (skewed build keys) or a rank failed: every rank leaves
  GDF_TRY(dist_post(tr, &bx, block_b, rpr_b));

  // ---- probe relation: `chunks` slices, software-pipelined ----
  const int64_t step = (n_p + chunks - 1) / chunks;
  info->slice_rows = step;
  const int width = kind == K_I64 ? 8 : 4;
  std::unique_ptr<PreparedBuild> build;
  ProbeAccum *acc = nullptr;               // owned by accum_finish once it is called
  // (queued level-2 kernels still read the receive buffers and write the accumulator's: drain the stream before anything is freed)
  struct AccGuard { ProbeAccum *&a; ~AccGuard() { (void)hipStreamSynchronize(stream0()); delete a; } } acc_guard{acc};
  bool failed = false;
  DistExchange *pending = nullptr;
  int pending_index = 0;
  auto add = [&](DistExchange *x, int index) -> gdf_error {
    GDF_TRY(dist_wait(tr, x));             // (a transport failure: returned at once)
    if (!acc || failed || hard != GDF_SUCCESS) return GDF_SUCCESS;
    const gdf_error ea = fj_probe_add(acc, x->rk.as<uint32_t>(), x->rf.as<uint32_t>(), cap_p, (int64_t)index * (int64_t)world * (int64_t)block_p,
                                      (int64_t)world * (int64_t)block_p);
    if (ea == GDF_UNSUPPORTED_METHOD || ea == GDF_COLUMN_SIZE_TOO_BIG) { failed = true; return GDF_SUCCESS; }     // a plan change, settled below
    note(ea);
    return GDF_SUCCESS;
  };
  for (int c = 0; c < chunks; ++c) {
    const int64_t a = std::min<int64_t>(n_p, (int64_t)c * step), b = std::min<int64_t>(n_p, (int64_t)(c + 1) * step);
    gdf_column slice = *probe_keys;
    slice.data = probe_keys->data ? (char *)probe_keys->data + (size_t)a * width : nullptr;
    slice.size = (gdf_size_type)(b - a);
    DistExchange &px = wire[(size_t)c + 1];
    if (hard == GDF_SUCCESS) note(fj_send(&slice, lo, hi, world, cb_p, cap_p, px.sk.as<uint32_t>(), probe_pos + a, px.sf.as<uint32_t>(), &over));
    failed = failed || over != 0;
    // the slice goes on the wire BEFORE anybody asks whether it overflowed (the agreement used to sit in front of the
    // first exchange, one blocking all-reduce on the happy path of every join); an overflowed buffer is memory safe, just useless --
    // and so is the buffer of a rank that has failed locally: it is posted all the same, the peers are waiting for it
    GDF_TRY(dist_post(tr, &px, block_p, rpr_p));
    if (c == 0) {
      // a region overflowed on some rank's FIRST slice (skewed probe keys are usually skewed everywhere), or a rank failed: every
      // rank leaves now, before three more slices are regrouped, shipped and partitioned for nothing
      GDF_TRY(dist_agree(tr, hard != GDF_SUCCESS ? 2 : (over != 0 ? 1 : 0), &worst));
      if (worst) return leave(worst);
    }
    if (!build) {
      GDF_TRY(dist_wait(tr, &bx));
      if (hard == GDF_SUCCESS) {
        PreparedBuild *pb = nullptr;
        gdf_error eb = fj_build_create(bx.rk.as<uint32_t>(), bx.rf.as<uint32_t>(), world, lo, fb_b, cb_b, cap_b, b_total / world + 1, &pb);
        build.reset(pb);
        if (eb == GDF_SUCCESS) {
          eb = accum_begin(build.get(), (size_t)(p_total / world + 1), &acc);
          if (eb != GDF_SUCCESS) acc = nullptr;
        }
        if (eb == GDF_UNSUPPORTED_METHOD || eb == GDF_COLUMN_SIZE_TOO_BIG) failed = true;
        else note(eb);
      }
      if (!build) build.reset(new PreparedBuild());       // (so that the build exchange is not waited for again)
    }
    if (pending) GDF_TRY(add(pending, pending_index));
    pending = &px;
    pending_index = c;
  }
  if (pending) GDF_TRY(add(pending, pending_index));
  bool have = false;
  if (acc && !failed && hard == GDF_SUCCESS) {
    ProbeAccum *fin = acc;
    acc = nullptr;                                           // accum_finish owns it from here
    const gdf_error ef = accum_finish(fin, probe_indices, build_indices);
    if (ef == GDF_UNSUPPORTED_METHOD || ef == GDF_COLUMN_SIZE_TOO_BIG) failed = true;
    else if (note(ef) == GDF_SUCCESS) have = true;
  }
  GDF_TRY(dist_agree(tr, hard != GDF_SUCCESS ? 2 : ((failed || !have) ? 1 : 0), &worst));
  if (worst) {
    if (have) { gdf_column_free(probe_indices); gdf_column_free(build_indices); }
    gdf_column_view(probe_indices, nullptr, nullptr, 0, N_GDF_TYPES);
    gdf_column_view(build_indices, nullptr, nullptr, 0, N_GDF_TYPES);
    return leave(worst);
  }
  HIP_TRY(hipStreamSynchronize(stream0()));                  // the receive buffers go out of scope with this call
  *declined = 0;
  return GDF_SUCCESS;
}

}  // namespace gdf_amd

using namespace gdf_amd;

extern "C" {

// non-reference export, test hook only (out_fine_off is a HOST array of 2^fb + 1 entries)
__attribute__((visibility("default"))) gdf_error gdf_amd_debug_partition(gdf_column *col, int fb, uint64_t *out_key, int32_t *out_idx,
                                                                        uint32_t *out_fine_off, uint32_t *out_joinable,
                                                                        uint64_t *out_info) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return debug_partition(col, fb, out_key, out_idx, out_fine_off, out_joinable, out_info);
  });
}

// non-reference exports (include/gdf/gdf_amd_ext.h): a build relation partitioned once and probed many times
__attribute__((visibility("default"))) gdf_error gdf_amd_join_build_create(gdf_column **build_cols, int num_cols, gdf_amd_join_build **out) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return build_create(build_cols, num_cols, reinterpret_cast<PreparedBuild **>(out));
  });
}
__attribute__((visibility("default"))) gdf_error gdf_amd_join_build_probe(gdf_amd_join_build *build, int left_join, gdf_column **probe_cols,
                                                                         int num_cols, gdf_column *probe_indices, gdf_column *build_indices) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return build_probe(reinterpret_cast<PreparedBuild *>(build), left_join, probe_cols, num_cols, probe_indices, build_indices);
  });
}
__attribute__((visibility("default"))) void gdf_amd_join_build_free(gdf_amd_join_build *build) {
  delete reinterpret_cast<PreparedBuild *>(build);
}
__attribute__((visibility("default"))) gdf_error gdf_amd_join_probe_begin(gdf_amd_join_build *build, size_t expected_rows, gdf_amd_join_probe **out) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return accum_begin(reinterpret_cast<PreparedBuild *>(build), expected_rows, reinterpret_cast<ProbeAccum **>(out));
  });
}
__attribute__((visibility("default"))) gdf_error gdf_amd_join_probe_add(gdf_amd_join_probe *probe, gdf_column **probe_cols, int num_cols) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return accum_add(reinterpret_cast<ProbeAccum *>(probe), probe_cols, num_cols);
  });
}
__attribute__((visibility("default"))) gdf_error gdf_amd_join_probe_finish(gdf_amd_join_probe *probe, gdf_column *probe_indices,
                                                                          gdf_column *build_indices) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return accum_finish(reinterpret_cast<ProbeAccum *>(probe), probe_indices, build_indices);
  });
}

// fused multi-GPU join (include/gdf/gdf_amd_ext.h)
__attribute__((visibility("default"))) gdf_error gdf_amd_fj_plan(int world, int64_t build_rows_total, int64_t rows_max, double rows_per_key,
                                                                int *fine_bits, int *coarse_bits, uint32_t *cap) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return fj_plan(world, build_rows_total, rows_max, rows_per_key, fine_bits, coarse_bits, cap);
  });
}
__attribute__((visibility("default"))) gdf_error gdf_amd_fj_send(gdf_column *keys, int64_t lo, int64_t hi, int world, int coarse_bits, uint32_t cap,
                                                                uint32_t *out_keys, uint32_t *out_pos, uint32_t *out_fill, int *overflowed) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return fj_send(keys, lo, hi, world, coarse_bits, cap, out_keys, out_pos, out_fill, overflowed);
  });
}
__attribute__((visibility("default"))) gdf_error gdf_amd_fj_build_create(const uint32_t *recv_keys, const uint32_t *recv_fill, int world, int64_t lo,
                                                                        int fine_bits, int coarse_bits, uint32_t cap, int64_t expected_rows,
                                                                        gdf_amd_join_build **out) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return fj_build_create(recv_keys, recv_fill, world, lo, fine_bits, coarse_bits, cap, expected_rows, reinterpret_cast<PreparedBuild **>(out));
  });
}
__attribute__((visibility("default"))) gdf_error gdf_amd_fj_probe_add(gdf_amd_join_probe *probe, const uint32_t *recv_keys, const uint32_t *recv_fill,
                                                                     uint32_t cap, int64_t position_base, int64_t buffer_elems) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return fj_probe_add(reinterpret_cast<ProbeAccum *>(probe), recv_keys, recv_fill, cap, position_base, buffer_elems);
  });
}

// the multi-GPU join behind the C ABI (include/gdf/gdf_amd_ext.h)
__attribute__((visibility("default"))) gdf_error gdf_amd_dist_inner_join(gdf_column *probe_keys, gdf_column *build_keys, gdf_amd_transport *transport,
                                                                        int chunks, uint32_t *probe_pos, uint32_t *build_pos, gdf_column *probe_indices,
                                                                        gdf_column *build_indices, gdf_amd_dist_info *info, int *declined) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return dist_inner_join(probe_keys, build_keys, transport, chunks, probe_pos, build_pos, probe_indices, build_indices, info, declined);
  });
}

gdf_error gdf_inner_join(gdf_column **left_cols, int num_left_cols, int left_join_cols[], gdf_column **right_cols,
                         int num_right_cols, int right_join_cols[], int num_cols_to_join, int result_num_cols,
                         gdf_column **result_cols, gdf_column *left_indices, gdf_column *right_indices,
                         gdf_context *join_context) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return join_entry(JOIN_INNER, left_cols, num_left_cols, left_join_cols, right_cols, num_right_cols, right_join_cols,
                    num_cols_to_join, result_num_cols, result_cols, left_indices, right_indices, join_context);
  });
}

gdf_error gdf_left_join(gdf_column **left_cols, int num_left_cols, int left_join_cols[], gdf_column **right_cols,
                        int num_right_cols, int right_join_cols[], int num_cols_to_join, int result_num_cols,
                        gdf_column **result_cols, gdf_column *left_indices, gdf_column *right_indices,
                        gdf_context *join_context) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return join_entry(JOIN_LEFT, left_cols, num_left_cols, left_join_cols, right_cols, num_right_cols, right_join_cols,
                    num_cols_to_join, result_num_cols, result_cols, left_indices, right_indices, join_context);
  });
}

gdf_error gdf_full_join(gdf_column **left_cols, int num_left_cols, int left_join_cols[], gdf_column **right_cols,
                        int num_right_cols, int right_join_cols[], int num_cols_to_join, int result_num_cols,
                        gdf_column **result_cols, gdf_column *left_indices, gdf_column *right_indices,
                        gdf_context *join_context) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return join_entry(JOIN_FULL, left_cols, num_left_cols, left_join_cols, right_cols, num_right_cols, right_join_cols,
                    num_cols_to_join, result_num_cols, result_cols, left_indices, right_indices, join_context);
  });
}

}  // extern "C"
