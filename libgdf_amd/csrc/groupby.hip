// groupby.hip -- gdf_group_by_{sum,min,max,avg,count}, HASH method.
//
// Reference path being replaced: src/sqls_ops.cu:1085-1363 ->
// groupby/groupby.cuh:86-419 -> groupby/hash/groupby_compute_api.h:140-225 with the
// kernels of groupby_kernels.cuh:42-160 over concurrent_unordered_map.cuh (a
// 2*N-slot global table keyed by ROW INDEX: every row pays a CAS on a hot slot
// plus a rows_equal gather of the group's first row, and init + extract sweep
// 2*N slots).  Here:
//
//   gb_aggregate  every workgroup pre-aggregates its rows in an LDS
//                 open-addressing table keyed by the packed 64-bit key
//                 (ds_cmpst_b64 claim, ds_add/min/max on the accumulator), so hot
//                 keys never leave the CU; LDS entries are merged into a global
//                 table sized to the number of GROUPS, not rows, when the
//                 workgroup finishes.  Keys that do not fit 64 bits (or float
//                 keys, which must compare with ==) use a global table keyed by
//                 first-row index like the reference, without the LDS stage.
//   gb_extract    wave-ballot compaction of the occupied slots into the caller's
//                 preallocated outputs, finishing AVG = SUM / COUNT in the same pass.
//   sort_result_rows  optional lexicographic sort of the result rows through sort.hip's
//                 radix sort (flag_sort_result, and always for AVG as in groupby.cuh:345-386).
//
// Semantics kept: aggregation in the INPUT dtype with wrap-around for integers
// (aggregation_operations.cuh:30-86), COUNT in the OUTPUT column's dtype
// (groupby.cuh:102-109), AVG = sum_in_input_dtype / (avg_type)count with avg_type =
// output dtype (groupby.cuh:308-328), any valid mask -> GDF_VALIDITY_UNSUPPORTED
// (sqls_ops.cu:1103-1106), empty input -> all output sizes 0, out_col_indices ignored.
#include "internal.h"

#include <cstdlib>
#include <vector>

namespace gdf_amd {

constexpr int GB_THREADS = 512;
constexpr uint32_t GB_LDS_SLOTS = 4096;            // keys 32 KiB + acc 32 KiB (+ cnt 32 KiB for AVG)
constexpr uint32_t GB_LDS_LIMIT = GB_LDS_SLOTS * 3 / 4;
constexpr uint64_t GB_EMPTY_KEY = 0x8000000000000000ULL;   // reserved packed key; a real key with these bits uses slot T
constexpr int32_t GB_EMPTY_ROW = -1;

enum GbOp : int { OP_SUM = 0, OP_MIN, OP_MAX, OP_AVG, OP_COUNT, OP_COUNT_DISTINCT };   // sort.hip's SgOp has the same order

// Column c occupies bits [shift, shift + bits) of the packed key and holds (value - bias).
// Natural layout: bias 0, bits = 8 * width (the raw element bits).  Range layout (integer
// key columns whose widths sum to more than 8 bytes): bias = column minimum, bits =
// bit length of (max - min), found by one min/max pass -- C5's (int64, int32) key with
// 1e6 x 16 distinct values packs into 24 bits this way and keeps the LDS / dense paths.
struct GbKeyPlan {
  int packed;                  // 1: exact 64-bit packed key, 0: first-row table + rows_equal
  int ordered;                 // 1: unsigned order of the packed key == lexicographic typed order of the rows
  int total_bits;
  int shift[MAX_KEY_COLS];
  int bits[MAX_KEY_COLS];
  int64_t bias[MAX_KEY_COLS];
};

static GbKeyPlan gb_plan_keys(const KeyTable &t) {
  GbKeyPlan p{};
  int total = 0;
  bool all_int = true;
  for (int c = 0; c < t.ncols; ++c) {
    if (t.col[c].kind == K_F32 || t.col[c].kind == K_F64) all_int = false;
    p.shift[c] = total * 8;
    p.bits[c] = t.col[c].width * 8;
    p.bias[c] = 0;
    total += t.col[c].width;
  }
  p.packed = (all_int && total <= 8) ? 1 : 0;
  p.total_bits = total * 8;
  return p;
}

__device__ __forceinline__ int64_t load_signed(const ColView &c, int64_t i) {
  switch (c.width) {
    case 1: return ((const int8_t *)c.data)[i];
    case 2: return ((const int16_t *)c.data)[i];
    case 4: return ((const int32_t *)c.data)[i];
    default: return ((const int64_t *)c.data)[i];
  }
}
__device__ __forceinline__ uint64_t low_mask(int bits) { return bits >= 64 ? ~0ULL : ((1ULL << bits) - 1ULL); }

// Float key columns take part in the packed-key paths through a signed integer IMAGE that orders like the values
// and equals iff the values are == : -0.0 is folded onto +0.0, negative values have their magnitude bits flipped.
// NaN has no such image (NaN != NaN: every NaN row is its own group): gb_plan_range declines when it meets one.
__host__ __device__ __forceinline__ long long f64_image(uint64_t b) {
  if ((b << 1) == 0) b = 0;
  const long long s = (long long)b;
  return s ^ ((s >> 63) & 0x7fffffffffffffffLL);
}
__host__ __device__ __forceinline__ long long f32_image(uint32_t b) {
  if ((uint32_t)(b << 1) == 0) b = 0;
  const int32_t s = (int32_t)b;
  return (long long)(s ^ ((s >> 31) & 0x7fffffff));
}
// the element of key column c at row i as a signed 64-bit number (integers: the value; floats: the image)
__device__ __forceinline__ long long load_key_image(const ColView &c, int64_t i) {
  if (c.kind == K_F64) return f64_image(((const uint64_t *)c.data)[i]);
  if (c.kind == K_F32) return f32_image(((const uint32_t *)c.data)[i]);
  return load_signed(c, i);
}

__device__ __forceinline__ uint64_t gb_pack(const KeyTable &t, const GbKeyPlan &p, int64_t i) {
  uint64_t k = 0;
  for (int c = 0; c < t.ncols; ++c)
    k |= ((uint64_t)(load_key_image(t.col[c], i) - p.bias[c]) & low_mask(p.bits[c])) << p.shift[c];
  return k;
}
// inverse of gb_pack for column c, stored at the column's width
__device__ __forceinline__ void gb_unpack_store(const KeyTable &t, const GbKeyPlan &p, uint64_t key, int c, void *out, int64_t pos) {
  uint64_t bits = ((key >> p.shift[c]) & low_mask(p.bits[c])) + (uint64_t)p.bias[c];
  if (t.col[c].kind == K_F64) { const long long img = (long long)bits; bits = (uint64_t)(img ^ ((img >> 63) & 0x7fffffffffffffffLL)); }
  if (t.col[c].kind == K_F32) { const int32_t img = (int32_t)(long long)bits; bits = (uint32_t)(img ^ ((img >> 31) & 0x7fffffff)); }
  switch (t.col[c].width) {
    case 1: ((uint8_t *)out)[pos] = (uint8_t)bits; break;
    case 2: ((uint16_t *)out)[pos] = (uint16_t)bits; break;
    case 4: ((uint32_t *)out)[pos] = (uint32_t)bits; break;
    default: ((uint64_t *)out)[pos] = bits; break;
  }
}

// hash of a row for the first-row table: equal rows (under ==) must hash equally, so
// -0.0 is folded onto +0.0; NaN rows hash by bit pattern and never compare equal.
__device__ __forceinline__ uint64_t gb_hash_row(const KeyTable &t, int64_t i) {
  uint64_t h = 0x9e3779b97f4a7c15ULL;
  for (int c = 0; c < t.ncols; ++c) {
    uint64_t b = load_bits(t.col[c], i);
    if (t.col[c].kind == K_F32 && (uint32_t)(b << 1) == 0) b = 0;
    if (t.col[c].kind == K_F64 && (b << 1) == 0) b = 0;
    h = mix64(h ^ b) + 0x9e3779b97f4a7c15ULL * (uint64_t)(c + 1);
  }
  return h;
}

// ---- 64-bit accumulator encoding ---------------------------------------------
// SUM/AVG: integers as wrapped uint64, floats as double.  MIN/MAX: order-preserving
// unsigned image so one unsigned atomic serves every dtype.  COUNT: uint64.
__device__ __forceinline__ uint64_t ord_i64(int64_t v) { return (uint64_t)v ^ 0x8000000000000000ULL; }
__device__ __forceinline__ uint64_t ord_f64(double d) {
  const uint64_t b = (uint64_t)__double_as_longlong(d);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double unord_f64(uint64_t u) {
  const uint64_t b = (u >> 63) ? (u & 0x7fffffffffffffffULL) : ~u;
  return __longlong_as_double((long long)b);
}

struct GbVal { const void *data; int kind; const uint8_t *valid; };   // valid: null values are skipped (C5 semantics)

__device__ __forceinline__ int64_t load_int(const GbVal &v, int64_t i) {
  switch (v.kind) {
    case K_I8: return ((const int8_t *)v.data)[i];
    case K_I16: return ((const int16_t *)v.data)[i];
    case K_I32: return ((const int32_t *)v.data)[i];
    default: return ((const int64_t *)v.data)[i];
  }
}
__device__ __forceinline__ double load_flt(const GbVal &v, int64_t i) {
  return v.kind == K_F32 ? (double)((const float *)v.data)[i] : ((const double *)v.data)[i];
}
__host__ __device__ __forceinline__ bool is_flt(int kind) { return kind == K_F32 || kind == K_F64; }

__host__ __device__ __forceinline__ uint64_t acc_identity(int op) { return op == OP_MIN ? ~0ULL : 0ULL; }
static inline uint64_t acc_identity_host(int op) { return op == OP_MIN ? ~0ULL : 0ULL; }

// image of row i's value that is folded into the accumulator
__device__ __forceinline__ uint64_t acc_image(int op, const GbVal &v, int64_t i) {
  switch (op) {
    case OP_COUNT: return 1;
    case OP_MIN: case OP_MAX: return is_flt(v.kind) ? ord_f64(load_flt(v, i)) : ord_i64(load_int(v, i));
    default: return is_flt(v.kind) ? (uint64_t)__double_as_longlong(load_flt(v, i)) : (uint64_t)load_int(v, i);
  }
}

// fold `img` into *acc (LDS or global; the compiler picks ds_ / global_ atomics)
__device__ __forceinline__ void acc_fold(int op, bool flt, unsigned long long *acc, uint64_t img) {
  switch (op) {
    case OP_MIN: atomicMin(acc, (unsigned long long)img); break;
    case OP_MAX: atomicMax(acc, (unsigned long long)img); break;
    case OP_COUNT: atomicAdd(acc, (unsigned long long)img); break;
    default:
      if (flt) atomicAdd((double *)acc, __longlong_as_double((long long)img));
      else atomicAdd(acc, (unsigned long long)img);
  }
}
// merging two partial accumulators uses the same fold (sum of sums, min of mins, ...)

struct GbTable {            // global table; slot T is reserved for the key GB_EMPTY_KEY (packed mode)
  uint32_t T;               // power of two
  unsigned long long *keys; // packed mode
  int32_t *first;           // first-row mode
  unsigned long long *acc;
  unsigned long long *cnt;  // AVG only
  unsigned int *occupied;   // number of claimed slots
  unsigned int *overflow;   // set when the table passes its fill limit
  unsigned int *special;    // set when some row carries the reserved key (slot T is live)
  uint32_t limit;
};

// returns slot index, or 0xffffffff on overflow
__device__ __forceinline__ uint32_t gtable_find_packed(const GbTable &g, uint64_t key) {
  if (key == GB_EMPTY_KEY) { *g.special = 1u; return g.T; }
  uint32_t slot = (uint32_t)(mix64(key) >> 32) & (g.T - 1);
  for (uint32_t probes = 0; probes < g.T; ++probes) {
    // a table that overflowed keeps filling until every workgroup has noticed: do not walk a full table
    if ((probes & 63u) == 63u && *(volatile unsigned int *)g.overflow) return 0xffffffffu;
    unsigned long long cur = g.keys[slot];
    if (cur == key) return slot;
    if (cur == GB_EMPTY_KEY) {
      const unsigned long long old = atomicCAS(&g.keys[slot], (unsigned long long)GB_EMPTY_KEY, (unsigned long long)key);
      if (old == GB_EMPTY_KEY) {
        if (atomicAdd(g.occupied, 1u) >= g.limit) atomicExch(g.overflow, 1u);
        return slot;
      }
      if (old == key) return slot;
    }
    slot = (slot + 1) & (g.T - 1);
  }
  return 0xffffffffu;
}

__device__ __forceinline__ uint32_t gtable_find_rows(const GbTable &g, const KeyTable &t, int64_t row) {
  uint32_t slot = (uint32_t)(gb_hash_row(t, row) >> 32) & (g.T - 1);
  for (uint32_t probes = 0; probes < g.T; ++probes) {
    if ((probes & 63u) == 63u && *(volatile unsigned int *)g.overflow) return 0xffffffffu;
    int32_t cur = g.first[slot];
    if (cur == GB_EMPTY_ROW) {
      const int32_t old = atomicCAS(&g.first[slot], GB_EMPTY_ROW, (int32_t)row);
      if (old == GB_EMPTY_ROW) {
        if (atomicAdd(g.occupied, 1u) >= g.limit) atomicExch(g.overflow, 1u);
        return slot;
      }
      cur = old;
    }
    if (rows_equal(t, row, t, cur)) return slot;
    slot = (slot + 1) & (g.T - 1);
  }
  return 0xffffffffu;
}

__global__ __launch_bounds__(256) void gb_init_table(GbTable g, int op, int packed) {
  const uint32_t n = g.T + 1;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (packed) g.keys[i] = GB_EMPTY_KEY; else g.first[i] = GB_EMPTY_ROW;
    g.acc[i] = acc_identity(op);
    if (g.cnt) g.cnt[i] = 0;
  }
}

// ---------------------------------------------------------------------------
// aggregation
// ---------------------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(GB_THREADS) void gb_aggregate(KeyTable t, GbKeyPlan plan, GbVal val, int op, GbTable g,
                                                           int64_t chunk) {
  // g.cnt != null: the number of (valid) values per group is tracked -- AVG, or any op over a masked column
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  unsigned long long *lkey = (unsigned long long *)gb_lds;
  unsigned long long *lacc = lkey + GB_LDS_SLOTS;
  unsigned long long *lcnt = lacc + GB_LDS_SLOTS;            // AVG only
  const bool counted = g.cnt != nullptr;
  unsigned int *lfill = (unsigned int *)(lcnt + (counted ? GB_LDS_SLOTS : 0));
  const bool flt = is_flt(val.kind);
  const bool avg = counted;            // below, "avg" means: keep a per-group count next to the accumulator
  const int fold_op = op == OP_AVG ? OP_SUM : op;

  if (PACKED) {
    for (uint32_t i = threadIdx.x; i < GB_LDS_SLOTS; i += GB_THREADS) {
      lkey[i] = GB_EMPTY_KEY;
      lacc[i] = acc_identity(op);
      if (avg) lcnt[i] = 0;
    }
    if (threadIdx.x == 0) *lfill = 0;
    block_sync();
  }

  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
  for (int64_t i = begin + threadIdx.x; i < end; i += GB_THREADS) {
    if (*(volatile unsigned int *)g.overflow) break;     // another workgroup filled the table: the host retries larger
    if (!row_valid(t, i)) continue;                       // a null in any key column drops the row
    const bool vok = !val.valid || bit_is_set(val.valid, i);   // a null value keeps its group but adds nothing
    const uint64_t img = acc_image(fold_op, val, i);
    if (PACKED) {
      const uint64_t key = gb_pack(t, plan, i);
      bool done = false;
      if (key != GB_EMPTY_KEY) {
        uint32_t slot = (uint32_t)mix64(key) & (GB_LDS_SLOTS - 1);
        for (int probes = 0; probes < 32; ++probes) {
          unsigned long long cur = lkey[slot];
          if (cur == GB_EMPTY_KEY) {
            if (*(volatile unsigned int *)lfill >= GB_LDS_LIMIT) break;      // LDS table full: new keys go to HBM
            const unsigned long long old = atomicCAS(&lkey[slot], (unsigned long long)GB_EMPTY_KEY, (unsigned long long)key);
            if (old == GB_EMPTY_KEY) { atomicAdd(lfill, 1u); cur = key; } else cur = old;
          }
          if (cur == key) {
            if (vok) {
              acc_fold(fold_op, flt, &lacc[slot], img);
              if (avg) atomicAdd(&lcnt[slot], 1ULL);
            }
            done = true;
            break;
          }
          slot = (slot + 1) & (GB_LDS_SLOTS - 1);
        }
      }
      if (!done) {
        const uint32_t s = gtable_find_packed(g, key);
        if (s == 0xffffffffu) { atomicExch(g.overflow, 1u); break; }
        if (vok) {
          acc_fold(fold_op, flt, &g.acc[s], img);
          if (avg) atomicAdd(&g.cnt[s], 1ULL);
        }
      }
    } else {
      const uint32_t s = gtable_find_rows(g, t, i);
      if (s == 0xffffffffu) { atomicExch(g.overflow, 1u); break; }
      if (vok) {
        acc_fold(fold_op, flt, &g.acc[s], img);
        if (avg) atomicAdd(&g.cnt[s], 1ULL);
      }
    }
  }

  if (PACKED) {
    block_sync();
    // merge this workgroup's partial aggregates into the global table
    for (uint32_t i = threadIdx.x; i < GB_LDS_SLOTS; i += GB_THREADS) {
      const uint64_t key = lkey[i];
      if (key == GB_EMPTY_KEY) continue;
      const uint32_t s = gtable_find_packed(g, key);
      if (s == 0xffffffffu) { atomicExch(g.overflow, 1u); continue; }
      acc_fold(fold_op, flt, &g.acc[s], lacc[i]);
      if (avg) atomicAdd(&g.cnt[s], lcnt[i]);
    }
  }
}

// ---------------------------------------------------------------------------
// extraction
// ---------------------------------------------------------------------------
struct GbOut {
  int ncols;
  void *key_out[MAX_KEY_COLS];
  void *agg_out;
  int agg_kind;      // kind the aggregate is WRITTEN as
  int in_kind;       // kind of the input values (decides the accumulator encoding)
  uint8_t *agg_ok;   // optional: 1 byte per group, 0 = the group had no valid value (output is null)
  int counted;       // the count passed to store_result is the number of VALID values
  // SORT-method calls served by the direct path (group_by_single): out_col_indices = every group's LAST row in input order
  // (sqls_ops.cu:1154-1156 / sqls_g_tester.cu:250-256), taken from last_rows[id] (row + 1; gb_direct_last_rows)
  size_t *indices;
  const unsigned int *last_rows;
};

__device__ __forceinline__ void store_int(void *out, int kind, int64_t pos, int64_t v) {
  switch (kind) {
    case K_I8: ((int8_t *)out)[pos] = (int8_t)v; break;
    case K_I16: ((int16_t *)out)[pos] = (int16_t)v; break;
    case K_I32: ((int32_t *)out)[pos] = (int32_t)v; break;
    case K_I64: ((int64_t *)out)[pos] = v; break;
    case K_F32: ((float *)out)[pos] = (float)v; break;
    default: ((double *)out)[pos] = (double)v; break;
  }
}
__device__ __forceinline__ void store_flt(void *out, int kind, int64_t pos, double v) {
  switch (kind) {
    case K_I8: ((int8_t *)out)[pos] = (int8_t)v; break;
    case K_I16: ((int16_t *)out)[pos] = (int16_t)v; break;
    case K_I32: ((int32_t *)out)[pos] = (int32_t)v; break;
    case K_I64: ((int64_t *)out)[pos] = (int64_t)v; break;
    case K_F32: ((float *)out)[pos] = (float)v; break;
    default: ((double *)out)[pos] = v; break;
  }
}
// wrap a 64-bit integer sum to the width of `kind` (sum_op adds in the input dtype)
__device__ __forceinline__ int64_t wrap_int(int kind, uint64_t v) {
  switch (kind) {
    case K_I8: return (int8_t)v;
    case K_I16: return (int16_t)v;
    case K_I32: return (int32_t)v;
    default: return (int64_t)v;
  }
}
// (avg_type)count, then the division in the usual-arithmetic-conversion type of
// (sum_type, avg_type), then the cast to avg_type: groupby.cuh:308-328.
__device__ __forceinline__ void store_avg(const GbOut &o, int64_t pos, uint64_t acc, uint64_t count) {
  const int sk = o.in_kind, ak = o.agg_kind;
  if (!is_flt(sk) && !is_flt(ak)) {
    const int64_t s = wrap_int(sk, acc);
    const int64_t c = wrap_int(ak, count);            // static_cast<avg_type>(count)
    // both operands promote to int (narrow types) or to the wider of the two: an int64 divide covers every case
    store_int(o.agg_out, ak, pos, c == 0 ? 0 : s / c);
    return;
  }
  if (is_flt(sk)) {
    double s = __longlong_as_double((long long)acc);
    if (sk == K_F32) s = (double)(float)s;             // the sum lives in a float column
    if (is_flt(ak)) {
      const double c = ak == K_F32 ? (double)(float)count : (double)count;
      const bool in_float = (sk == K_F32 && ak == K_F32);
      store_flt(o.agg_out, ak, pos, in_float ? (double)((float)s / (float)c) : s / c);
    } else {
      const int64_t c = wrap_int(ak, count);
      // float / integer -> computed in the float type of the sum
      const double q = sk == K_F32 ? (double)((float)s / (float)c) : s / (double)c;
      store_flt(o.agg_out, ak, pos, q);
    }
    return;
  }
  // integer sum, float avg_type: computed in avg_type
  const int64_t s = wrap_int(sk, acc);
  if (ak == K_F32) store_flt(o.agg_out, ak, pos, (double)((float)s / (float)count));
  else store_flt(o.agg_out, ak, pos, (double)s / (double)count);
}

__device__ __forceinline__ void store_result(const GbOut &o, int op, int64_t pos, uint64_t acc, uint64_t cnt) {
  const bool empty = o.counted && cnt == 0 && op != OP_COUNT;    // every value of the group was null
  if (o.agg_ok) o.agg_ok[pos] = empty ? 0 : 1;
  if (empty) { store_int(o.agg_out, o.agg_kind, pos, 0); return; }
  switch (op) {
    case OP_COUNT: store_int(o.agg_out, o.agg_kind, pos, (int64_t)acc); break;   // count_op<out dtype>
    case OP_AVG: store_avg(o, pos, acc, cnt); break;
    case OP_SUM:
      if (is_flt(o.in_kind)) store_flt(o.agg_out, o.in_kind, pos, __longlong_as_double((long long)acc));
      else store_int(o.agg_out, o.in_kind, pos, wrap_int(o.in_kind, acc));
      break;
    default:   // MIN / MAX
      if (is_flt(o.in_kind)) store_flt(o.agg_out, o.in_kind, pos, unord_f64(acc));
      else store_int(o.agg_out, o.in_kind, pos, (int64_t)(acc ^ 0x8000000000000000ULL));
  }
}

template <bool PACKED>
__global__ __launch_bounds__(256) void gb_extract(KeyTable t, GbKeyPlan plan, GbTable g, GbOut o, int op,
                                                  unsigned int special_used, unsigned long long *out_count) {
  const uint32_t n = g.T + 1;
  const uint32_t stride = gridDim.x * blockDim.x;
  const uint32_t rounds = (n + stride - 1) / stride;
  for (uint32_t rnd = 0; rnd < rounds; ++rnd) {
    const uint32_t i = rnd * stride + blockIdx.x * blockDim.x + threadIdx.x;
    bool live = false;
    uint64_t key = 0;
    int32_t row = 0;
    if (i < n) {
      if (PACKED) {
        if (i == g.T) { live = special_used != 0; key = GB_EMPTY_KEY; }
        else { key = g.keys[i]; live = key != GB_EMPTY_KEY; }
      } else if (i < g.T) {
        row = g.first[i];
        live = row != GB_EMPTY_ROW;
      }
    }
    const unsigned long long m = __ballot(live);
    unsigned long long base = 0;
    if (lane_id() == 0 && m) base = atomicAdd(out_count, (unsigned long long)__popcll(m));
    base = __shfl(base, 0, WAVE);
    if (live) {
      const int64_t pos = (int64_t)(base + mask_rank(m));
      for (int c = 0; c < t.ncols; ++c) {
        if (PACKED) { gb_unpack_store(t, plan, key, c, o.key_out[c], pos); continue; }
        const uint64_t bits = load_bits(t.col[c], row);
        switch (t.col[c].width) {
          case 1: ((uint8_t *)o.key_out[c])[pos] = (uint8_t)bits; break;
          case 2: ((uint16_t *)o.key_out[c])[pos] = (uint16_t)bits; break;
          case 4: ((uint32_t *)o.key_out[c])[pos] = (uint32_t)bits; break;
          default: ((uint64_t *)o.key_out[c])[pos] = bits; break;
        }
      }
      store_result(o, op, pos, g.acc[i], g.cnt ? g.cnt[i] : 0);
    }
  }
}

// ---------------------------------------------------------------------------
// result sort: the OUTPUT key columns are ordered lexicographically with typed <
// (LesserRTTI, sqls_rtti_comp.hpp:33-279) by sort.hip's radix sort of a row
// permutation, then every output column is gathered through it.
// ---------------------------------------------------------------------------
__global__ void gb_sort_gather(const uint32_t *perm, uint32_t n, int width, const void *in, void *out) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t s = perm[i];
    switch (width) {
      case 1: ((uint8_t *)out)[i] = ((const uint8_t *)in)[s]; break;
      case 2: ((uint16_t *)out)[i] = ((const uint16_t *)in)[s]; break;
      case 4: ((uint32_t *)out)[i] = ((const uint32_t *)in)[s]; break;
      default: ((uint64_t *)out)[i] = ((const uint64_t *)in)[s]; break;
    }
  }
}

static gdf_error sort_result_rows(int ncols, gdf_column **key_cols, const int *key_kind, void *agg, int agg_width, uint32_t n,
                                  uint8_t *agg_ok = nullptr) {
  if (n < 2) return GDF_SUCCESS;
  KeyTable t{};
  t.ncols = ncols;
  t.nrows = n;
  for (int c = 0; c < ncols; ++c) {
    t.col[c].data = key_cols[c]->data;
    t.col[c].valid = nullptr;
    t.col[c].kind = key_kind[c];
    t.col[c].width = kind_width((ElemKind)key_kind[c]);
  }
  DevBuf perm, tmp;
  GDF_TRY(order_rows(t, n, perm, nullptr, nullptr));
  RMM_TRY(tmp.alloc((size_t)8 * n));
  const int grid = stream_grid(n, 256 * 4);
  auto permute = [&](void *data, int width) -> gdf_error {
    hipLaunchKernelGGL(gb_sort_gather, dim3(grid), dim3(256), 0, stream0(), perm.as<uint32_t>(), n, width, data, tmp.p);
    HIP_TRY(hipMemcpyAsync(data, tmp.p, (size_t)width * n, hipMemcpyDeviceToDevice, stream0()));
    return GDF_SUCCESS;
  };
  for (int c = 0; c < ncols; ++c) GDF_TRY(permute(key_cols[c]->data, kind_width((ElemKind)key_kind[c])));
  GDF_TRY(permute(agg, agg_width));
  if (agg_ok) GDF_TRY(permute(agg_ok, 1));
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
}

// ---------------------------------------------------------------------------
// dense path (packed keys, few groups): dictionary -> dense group ids -> LDS accumulators
//
//   gb_dict_build     find-or-insert every row's key in the global table (almost all rows only READ:
//                     the key is already there; the table is a few hundred KB and L2 resident);
//   gb_dict_number    numbers the occupied slots 0..G-1 (ballot compaction);
//   gb_dense_aggregate  one workgroup per CU keeps ALL G accumulators in LDS (8 B each, +4 B row count
//                     for AVG), looks each row's key up read-only and folds the value with one LDS
//                     atomic; partial results are merged into the global accumulators once per workgroup;
//   gb_dense_extract  writes group g's key and finished aggregate at output row g.
// The first version sent every key beyond the 3072 that fit its LDS key table to global atomics: C2
// (1e8 rows, 1e4 groups) ran at 354 GB/s.  Dense ids need no keys in LDS, so 16384 groups fit.
// ---------------------------------------------------------------------------
constexpr int GB_DENSE_THREADS = 1024;
constexpr int GB_DENSE_BATCH = 8;
constexpr uint32_t GB_DENSE_MAX_GROUPS = 16384;       // 128 KiB of 8-byte accumulators
constexpr uint32_t GB_DENSE_MAX_GROUPS_AVG = 12288;   // 144 KiB of 8-byte sums + 4-byte counts

// dictionary entry of the dense path: one 16-byte word so that a lookup is ONE L2 read
struct __attribute__((aligned(16))) GbDictEntry {
  unsigned long long key;
  uint32_t id;
  uint32_t pad;
};
struct GbDict {
  uint32_t T;                 // power of two; entry T is the reserved key's
  GbDictEntry *e;
  unsigned int *occupied, *overflow, *special;
  uint32_t limit;
};

// returns 0: gave up (table overflowed), 1: key present, 2: key newly inserted (the caller counts these,
// one atomic per wave instead of one per key)
__device__ __forceinline__ int dict_insert(const GbDict &d, uint64_t key) {
  if (key == GB_EMPTY_KEY) { *d.special = 1u; return 1; }
  uint32_t slot = (uint32_t)(mix64(key) >> 32) & (d.T - 1);
  for (uint32_t probes = 0; probes < d.T; ++probes) {
    if ((probes & 7u) == 7u && *(volatile unsigned int *)d.overflow) return 0;
    const unsigned long long cur = d.e[slot].key;
    if (cur == key) return 1;
    if (cur == GB_EMPTY_KEY) {
      const unsigned long long old = atomicCAS(&d.e[slot].key, (unsigned long long)GB_EMPTY_KEY, (unsigned long long)key);
      if (old == GB_EMPTY_KEY) return 2;
      if (old == key) return 1;
    }
    slot = (slot + 1) & (d.T - 1);
  }
  return 0;
}

// FASTKEY: one 8-byte integer key column -> the key IS the column word, loaded without any branch.
// (An LDS "already seen" set in front of the global table was tried and was 2x SLOWER: at 10k keys in
// 16k LDS slots a third of the rows leave the home slot and walk a dependent LDS chain; the L2-resident
// table at 4 % load answers almost every row with one independent read.)
constexpr int GB_DICT_THREADS = 512;

// MASKED: rows with a null key element are skipped (their lanes re-find the reserved key, which costs nothing)
template <bool FASTKEY, bool MASKED>
__global__ __launch_bounds__(GB_DICT_THREADS) void gb_dict_build(KeyTable t, GbKeyPlan plan, GbDict g, int64_t chunk, int64_t stride) {
  // stride > 1: a strided SAMPLE -- t.nrows counts the sampled rows, row i of the sample is row i * stride of the table
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
  for (int64_t base = begin; base < end; base += GB_DICT_THREADS * GB_DENSE_BATCH) {
    if (*(volatile unsigned int *)g.overflow) return;
    uint64_t key[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      const int64_t i = base + (int64_t)k * GB_DICT_THREADS + threadIdx.x;
      const int64_t ic = (i < end ? i : end - 1) * stride;       // clamped: finished lanes re-find a real key
      key[k] = FASTKEY ? ((const uint64_t *)t.col[0].data)[ic] : gb_pack(t, plan, ic);
    }
    bool skip[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      const int64_t i = base + (int64_t)k * GB_DICT_THREADS + threadIdx.x;
      skip[k] = MASKED && !row_valid(t, (i < end ? i : end - 1) * stride);
    }
    // almost every row finds its key already present in its home slot: probe all BATCH home slots
    // first (independent reads), insert only the misses
    unsigned long long home[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) home[k] = g.e[(uint32_t)(mix64(key[k]) >> 32) & (g.T - 1)].key;
    unsigned int fresh = 0;
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k)
      if (!skip[k] && (home[k] != key[k] || key[k] == GB_EMPTY_KEY)) {
        const int r = dict_insert(g, key[k]);
        if (r == 0) atomicExch(g.overflow, 1u);
        fresh += (r == 2);
      }
    fresh = wave_reduce_add(fresh);
    if (lane_id() == 0 && fresh && atomicAdd(g.occupied, fresh) + fresh > g.limit) atomicExch(g.overflow, 1u);
  }
}

// ids[slot] = dense id, group_slot[id] = slot; slot T stands for the reserved key
__global__ __launch_bounds__(256) void gb_dict_number(GbDict g, unsigned int special_used, uint32_t *group_slot,
                                                      unsigned int *counter) {
  if (special_used == 0xffffffffu) special_used = *g.special;        // not read back yet (LDS dictionary: no host round trip)
  // ONE claim on the counter per workgroup: a claim per wave was 4096 atomics on one word = 45 us of a 2^18-entry table's
  // numbering (the same-word atomic rate, ~90 per us), whatever the number of live entries
  __shared__ uint32_t wave_tot[256 / WAVE];
  __shared__ uint32_t block_base;
  const uint32_t n = g.T + 1;
  const uint32_t per = (n + gridDim.x * 256 - 1) / (gridDim.x * 256);      // consecutive entries per thread
  const uint32_t first = (blockIdx.x * 256 + threadIdx.x) * per;
  auto live = [&](uint32_t i) { return i < n && (i == g.T ? special_used != 0 : g.e[i].key != GB_EMPTY_KEY); };
  uint32_t mine = 0;
  for (uint32_t k = 0; k < per; ++k) mine += live(first + k);
  const uint32_t incl = wave_scan_incl(mine);
  if (lane_id() == WAVE - 1) wave_tot[threadIdx.x / WAVE] = incl;
  block_sync();
  uint32_t before = 0, total = 0;
  for (int w = 0; w < 256 / WAVE; ++w) { if (w < (int)(threadIdx.x / WAVE)) before += wave_tot[w]; total += wave_tot[w]; }
  if (threadIdx.x == 0) block_base = total ? atomicAdd(counter, total) : 0;
  block_sync();
  uint32_t id = block_base + before + incl - mine;
  for (uint32_t k = 0; k < per; ++k) {
    const uint32_t i = first + k;
    if (live(i)) {
      g.e[i].id = id;
      if (id <= g.limit) group_slot[id] = i;        // (a table that overflowed its limit is abandoned by the caller)
      ++id;
    }
  }
}

__global__ __launch_bounds__(256) void gb_fill_u64(unsigned long long *p, unsigned long long v, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
// the direct path's three clears in one launch: accumulators to the identity, row counts and the two result words to zero
__global__ __launch_bounds__(256) void gb_direct_init(unsigned long long *acc, unsigned long long identity, unsigned long long *cnt, uint32_t n,
                                                      unsigned int *two_words) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { acc[i] = identity; cnt[i] = 0; }
  if (blockIdx.x == 0 && threadIdx.x < 2) two_words[threadIdx.x] = 0;
}
__global__ __launch_bounds__(256) void gb_dict_clear(GbDictEntry *e, uint32_t n) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) e[i] = GbDictEntry{GB_EMPTY_KEY, 0u, 0u};
}

// read-only lookup of a key that is known to be in the table
__device__ __forceinline__ uint32_t dict_lookup(const GbDict &d, uint64_t key) {
  if (key == GB_EMPTY_KEY) return d.e[d.T].id;
  uint32_t slot = (uint32_t)(mix64(key) >> 32) & (d.T - 1);
  for (;;) {
    const uint4 w = *reinterpret_cast<const uint4 *>(&d.e[slot]);      // one 16-byte read: key + id
    if ((((unsigned long long)w.y << 32) | w.x) == key) return w.z;
    slot = (slot + 1) & (d.T - 1);
  }
}

// FASTVAL: 8-byte values (int64 / float64): the raw word is loaded branch-free and turned into the
// accumulator image afterwards
template <bool FASTKEY, bool FASTVAL, bool MASKED>
__global__ __launch_bounds__(GB_DENSE_THREADS) void gb_dense_aggregate(KeyTable t, GbKeyPlan plan, GbVal val, int op, GbDict g,
                                                                       uint32_t ngroups,
                                                                       unsigned long long *gacc, unsigned long long *gcnt,
                                                                       int64_t chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  unsigned long long *lacc = (unsigned long long *)gb_lds;
  unsigned int *lcnt = (unsigned int *)(lacc + ngroups);            // counted only
  const bool flt = is_flt(val.kind);
  const bool avg = gcnt != nullptr;     // "avg" = a count of (valid) values is kept per group: AVG, or a masked value column
  const int fold_op = op == OP_AVG ? OP_SUM : op;
  for (uint32_t i = threadIdx.x; i < ngroups; i += GB_DENSE_THREADS) {
    lacc[i] = acc_identity(fold_op);
    if (avg) lcnt[i] = 0;
  }
  block_sync();
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
  for (int64_t base = begin; base < end; base += (int64_t)GB_DENSE_THREADS * GB_DENSE_BATCH) {
    uint64_t key[GB_DENSE_BATCH], img[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {                       // all HBM loads first, from clamped addresses
      const int64_t i = base + (int64_t)k * GB_DENSE_THREADS + threadIdx.x;
      const int64_t ic = i < end ? i : end - 1;
      key[k] = FASTKEY ? ((const uint64_t *)t.col[0].data)[ic] : gb_pack(t, plan, ic);
      img[k] = FASTVAL ? ((const uint64_t *)val.data)[ic] : acc_image(fold_op, val, ic);
    }
    if (FASTVAL) {
#pragma unroll
      for (int k = 0; k < GB_DENSE_BATCH; ++k) {
        // same images as acc_image(): COUNT -> 1, MIN/MAX -> order-preserving, SUM -> the raw word (int64 / double bits)
        if (fold_op == OP_COUNT) img[k] = 1;
        else if (fold_op == OP_MIN || fold_op == OP_MAX)
          img[k] = flt ? ord_f64(__longlong_as_double((long long)img[k])) : ord_i64((int64_t)img[k]);
      }
    }
    // first probe of all BATCH keys issued together (independent 16-byte L2 reads); the rare key that
    // is not in its home slot finishes with the scalar walk
    uint32_t gid[GB_DENSE_BATCH];
    uint4 w[GB_DENSE_BATCH];
    bool use[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      const int64_t i = base + (int64_t)k * GB_DENSE_THREADS + threadIdx.x;
      use[k] = i < end;
      if (MASKED) {
        const int64_t ic = i < end ? i : end - 1;
        if (!row_valid(t, ic)) { use[k] = false; key[k] = GB_EMPTY_KEY; }      // never looked up in the table proper
        else if (val.valid && !bit_is_set(val.valid, ic)) use[k] = false;      // null value: the group exists, nothing to add
      }
    }
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      const uint32_t slot = (uint32_t)(mix64(key[k]) >> 32) & (g.T - 1);
      w[k] = *reinterpret_cast<const uint4 *>(&g.e[slot]);
    }
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      if ((((unsigned long long)w[k].y << 32) | w[k].x) == key[k] && key[k] != GB_EMPTY_KEY) gid[k] = w[k].z;
      else if (MASKED && !use[k]) gid[k] = 0;
      else gid[k] = dict_lookup(g, key[k]);
    }
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      if (use[k]) {
        acc_fold(fold_op, flt, &lacc[gid[k]], img[k]);
        if (avg) atomicAdd(&lcnt[gid[k]], 1u);
      }
    }
  }
  block_sync();
  for (uint32_t i = threadIdx.x; i < ngroups; i += GB_DENSE_THREADS) {
    const unsigned long long v = lacc[i];
    if (avg) {
      const unsigned int c = lcnt[i];
      if (c) { acc_fold(fold_op, flt, &gacc[i], v); atomicAdd(&gcnt[i], (unsigned long long)c); }
    } else if (v != acc_identity(fold_op) || fold_op == OP_SUM) {
      // an untouched MIN/MAX/COUNT cell equals the identity and contributes nothing; SUM cells are folded
      // unconditionally only when non-zero (adding 0 is a no-op, skip the atomic)
      if (!(fold_op == OP_SUM && v == 0 && !flt)) acc_fold(fold_op, flt, &gacc[i], v);
    }
  }
}

__global__ __launch_bounds__(256) void gb_dense_extract(KeyTable t, GbKeyPlan plan, GbDict g, const uint32_t *group_slot,
                                                        uint32_t ngroups, GbOut o, int op, const unsigned long long *gacc,
                                                        const unsigned long long *gcnt) {
  for (uint32_t gi = blockIdx.x * blockDim.x + threadIdx.x; gi < ngroups; gi += gridDim.x * blockDim.x) {
    const uint32_t slot = group_slot[gi];
    const uint64_t key = slot == g.T ? GB_EMPTY_KEY : g.e[slot].key;
    for (int c = 0; c < t.ncols; ++c) gb_unpack_store(t, plan, key, c, o.key_out[c], gi);
    store_result(o, op, gi, gacc[gi], gcnt ? gcnt[gi] : 0);
  }
}

// ---------------------------------------------------------------------------
// LDS dictionary (few groups under SPARSE keys -- C2 with its 10 k keys scattered over 2^62).
// The dense path above asks the L2-resident table twice per row (build: 0.62 ms, aggregate: 0.80 ms per 1e8 rows).  Here
// the dictionary comes from a strided SAMPLE, is numbered, and its image -- a 2-choice, 2-slot-bucket cuckoo table of
// 4-byte words (18-bit fingerprint << 14 | id) plus the 8-byte keys by id -- is copied into the LDS of every workgroup of
//   gb_ld_encode     one pass over the KEY column(s): two independent 8-byte LDS reads give four candidate words, a
//                    fingerprint match is confirmed against the full key (exact: no false merge), the row's 2-byte group
//                    id is written out.  A key the sample never saw is inserted into the global table the slow way,
//                    its rows get id 0xffff and the host numbers the (now complete) dictionary again and repeats the pass;
//   gb_ld_aggregate  one pass over ids + values: LDS accumulators indexed by id, as in the direct path.
// HBM bytes per row with 8-byte keys and values: 8 + 2 + 2 + 8 = 20 (dense path: 8 + 16).  A divergence-free lookup is
// the point: a linear-probing LDS table at this load (a third of the keys off their home slot) was 2x slower than the L2
// table (see gb_dict_build).
// ---------------------------------------------------------------------------
constexpr uint32_t GB_LD_BUCKETS = 8192;             // x 2 slots x 4 B = 64 KiB
constexpr uint32_t GB_LD_SLOTS = 2 * GB_LD_BUCKETS;
constexpr uint32_t GB_LD_MAX_GROUPS = 10922;         // two thirds of the slots; + 8 B per key = 150 KiB of LDS in gb_ld_encode
constexpr uint32_t GB_LD_EMPTY = 0xffffffffu;        // id 16383 never exists
constexpr uint32_t GB_LD_ID_MASK = 16383u;
constexpr int GB_LD_THREADS = 1024;

struct GbLdHash { uint32_t b1, b2, fp; };
__device__ __forceinline__ GbLdHash ld_hash(uint64_t key) {
  const uint64_t h = mix64(key);
  return GbLdHash{(uint32_t)h & (GB_LD_BUCKETS - 1), (uint32_t)(h >> 13) & (GB_LD_BUCKETS - 1), (uint32_t)(h >> 46)};
}

// one workgroup: keys by id + the cuckoo table over them, built in LDS, written out as the image every encode workgroup loads
// state: [1] set by gb_ld_encode when a row's key is not in the image, [2] this kernel gave up (too many groups for the
// image, or the cuckoo walk did not end), [3] the group count the image was built for
__global__ __launch_bounds__(GB_LD_THREADS) void gb_ld_image(GbDict g, const uint32_t *__restrict__ group_slot,
                                                             uint32_t *tabimg, unsigned long long *keyimg, unsigned int *state) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  uint32_t *tab = (uint32_t *)gb_lds;
  unsigned long long *keys = (unsigned long long *)(tab + GB_LD_SLOTS);
  unsigned int *fail = state + 2;
  // the group count comes from the device: the host has not read the sample's result back (one round trip less per call)
  const uint32_t ngroups = *g.occupied + (*g.special ? 1u : 0u);
  if (threadIdx.x == 0) state[3] = ngroups;
  if (*(volatile unsigned int *)g.overflow || ngroups == 0 || ngroups > GB_LD_MAX_GROUPS) {
    if (threadIdx.x == 0) *fail = 1u;
    return;
  }
  for (uint32_t i = threadIdx.x; i < GB_LD_SLOTS; i += GB_LD_THREADS) tab[i] = GB_LD_EMPTY;
  {   // the keys by id: two dependent gathers (id -> slot -> key), each as ONE round of loads (in a rolled loop they were
      // eleven round trips in a row: most of this kernel's 33 us)
    constexpr int PER = (GB_LD_MAX_GROUPS + GB_LD_THREADS - 1) / GB_LD_THREADS;
    uint32_t slot[PER];
    unsigned long long kv[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t i = k * GB_LD_THREADS + threadIdx.x;
      slot[k] = i < ngroups ? group_slot[i] : g.T;
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) kv[k] = slot[k] == g.T ? GB_EMPTY_KEY : g.e[slot[k]].key;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const uint32_t i = k * GB_LD_THREADS + threadIdx.x;
      if (i < ngroups) keys[i] = kv[k];
    }
  }
  block_sync();
  bool bad = false;
  for (uint32_t i = threadIdx.x; i < ngroups; i += GB_LD_THREADS) {
    GbLdHash h = ld_hash(keys[i]);
    uint32_t cur = (h.fp << 14) | i, b = h.b1;
    bool placed = false;
    for (int it = 0; it < 256; ++it) {
      if (atomicCAS(&tab[2 * b], GB_LD_EMPTY, cur) == GB_LD_EMPTY || atomicCAS(&tab[2 * b + 1], GB_LD_EMPTY, cur) == GB_LD_EMPTY) { placed = true; break; }
      // both slots taken: evict one (slots never become empty again, so the exchange returns an entry) and move IT to its
      // other bucket -- the random walk of cuckoo hashing, every token held by exactly one thread or one slot
      const uint32_t pick = ((cur * 2654435761u) >> 31) ^ (uint32_t)(it & 1);
      cur = atomicExch(&tab[2 * b + pick], cur);
      h = ld_hash(keys[cur & GB_LD_ID_MASK]);
      b = h.b1 == b ? h.b2 : h.b1;
    }
    bad = bad || !placed;
  }
  if (bad) *fail = 1u;
  block_sync();
  for (uint32_t i = threadIdx.x; i < GB_LD_SLOTS; i += GB_LD_THREADS) tabimg[i] = tab[i];
  for (uint32_t i = threadIdx.x; i < ngroups; i += GB_LD_THREADS) keyimg[i] = keys[i];
}

template <bool FASTKEY>
__global__ __launch_bounds__(GB_LD_THREADS) void gb_ld_encode(KeyTable t, GbKeyPlan plan, GbDict g, const uint32_t *__restrict__ tabimg,
                                                              const unsigned long long *__restrict__ keyimg,
                                                              uint16_t *__restrict__ ids, int64_t chunk, unsigned int *state) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  uint2 *tab2 = (uint2 *)gb_lds;                                              // bucket = two slot words
  unsigned long long *keys = (unsigned long long *)(gb_lds + sizeof(uint32_t) * GB_LD_SLOTS);
  if (state[2]) return;                                                       // no image: the host takes the dense path
  const uint32_t ngroups = state[3];
  unsigned int *missed = state + 1;
  {
    uint4 *dst = (uint4 *)gb_lds;
    const uint4 *src = (const uint4 *)tabimg;
    for (uint32_t i = threadIdx.x; i < GB_LD_SLOTS / 4; i += GB_LD_THREADS) dst[i] = src[i];
    for (uint32_t i = threadIdx.x; i < ngroups; i += GB_LD_THREADS) keys[i] = keyimg[i];
  }
  block_sync();
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
  unsigned int fresh = 0;
  for (int64_t base = begin; base < end; base += (int64_t)GB_LD_THREADS * GB_DENSE_BATCH) {
    uint64_t key[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {                       // all HBM loads first, from clamped addresses
      const int64_t i = base + (int64_t)k * GB_LD_THREADS + threadIdx.x;
      const int64_t ic = i < end ? i : end - 1;
      key[k] = FASTKEY ? ((const uint64_t *)t.col[0].data)[ic] : gb_pack(t, plan, ic);
    }
    uint2 c1[GB_DENSE_BATCH], c2[GB_DENSE_BATCH];
    uint32_t fp[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {                       // sixteen independent LDS reads, one wait
      const GbLdHash h = ld_hash(key[k]);
      fp[k] = h.fp;
      c1[k] = tab2[h.b1];
      c2[k] = tab2[h.b2];
    }
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      const int64_t i = base + (int64_t)k * GB_LD_THREADS + threadIdx.x;
      const uint32_t cand[4] = {c1[k].x, c1[k].y, c2[k].x, c2[k].y};
      uint32_t sel = 0, nmatch = 0;
#pragma unroll
      for (int c = 0; c < 4; ++c) {                                  // (the empty word carries id 16383 >= ngroups)
        const bool m = (cand[c] >> 14) == fp[k] && (cand[c] & GB_LD_ID_MASK) < ngroups;
        sel = m ? (cand[c] & GB_LD_ID_MASK) : sel;
        nmatch += m;
      }
      uint32_t id = 0xffffu;
      if (nmatch == 1) {                                             // the common case: ONE dependent read confirms the key
        if (keys[sel] == key[k]) id = sel;
      } else if (nmatch > 1) {                                       // two fingerprints agree (or both buckets coincide): check them all
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t cid = cand[c] & GB_LD_ID_MASK;
          if ((cand[c] >> 14) == fp[k] && cid < ngroups && keys[cid] == key[k]) id = cid;
        }
      }
      if (i < end) {
        if (id == 0xffffu) {                                       // the sample never saw this key: the global table learns it
          const int r = dict_insert(g, key[k]);
          if (r == 0) atomicExch(g.overflow, 1u);
          fresh += (r == 2);
          *missed = 1u;
        }
        ids[i] = (uint16_t)id;
      }
    }
  }
  fresh = wave_reduce_add(fresh);
  if (lane_id() == 0 && fresh && atomicAdd(g.occupied, fresh) + fresh > g.limit) atomicExch(g.overflow, 1u);
}

// FASTVAL = 8 / 4: an 8- / 4-byte value column read directly and widened in registers, 0: acc_image() (gb_direct_aggregate)
template <int FASTVAL>
__global__ __launch_bounds__(GB_LD_THREADS) void gb_ld_aggregate(GbVal val, int op, const uint16_t *__restrict__ ids, int64_t nrows,
                                                                 uint32_t ngroups, unsigned long long *gacc, unsigned long long *gcnt,
                                                                 int64_t chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  unsigned long long *lacc = (unsigned long long *)gb_lds;
  unsigned int *lcnt = (unsigned int *)(lacc + ngroups);            // counted only
  const bool flt = is_flt(val.kind);
  const bool avg = gcnt != nullptr;
  const int fold_op = op == OP_AVG ? OP_SUM : op;
  for (uint32_t i = threadIdx.x; i < ngroups; i += GB_LD_THREADS) {
    lacc[i] = acc_identity(fold_op);
    if (avg) lcnt[i] = 0;
  }
  block_sync();
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = begin + chunk < nrows ? begin + chunk : nrows;
  for (int64_t base = begin; base < end; base += (int64_t)GB_LD_THREADS * GB_DENSE_BATCH) {
    uint32_t id[GB_DENSE_BATCH];
    uint64_t img[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      const int64_t i = base + (int64_t)k * GB_LD_THREADS + threadIdx.x;
      const int64_t ic = i < end ? i : end - 1;
      id[k] = ids[ic];
      img[k] = FASTVAL == 8 ? ((const uint64_t *)val.data)[ic] : (FASTVAL == 4 ? (uint64_t)((const uint32_t *)val.data)[ic] : acc_image(fold_op, val, ic));
    }
    if (FASTVAL) {
#pragma unroll
      for (int k = 0; k < GB_DENSE_BATCH; ++k) {
        if (FASTVAL == 4)     // widen: float -> the bits of its double, int32 -> sign-extended
          img[k] = flt ? (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)img[k])) : (uint64_t)(int64_t)(int32_t)(uint32_t)img[k];
        if (fold_op == OP_COUNT) img[k] = 1;
        else if (fold_op == OP_MIN || fold_op == OP_MAX)
          img[k] = flt ? ord_f64(__longlong_as_double((long long)img[k])) : ord_i64((int64_t)img[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      if (base + (int64_t)k * GB_LD_THREADS + threadIdx.x < end) {
        acc_fold(fold_op, flt, &lacc[id[k]], img[k]);
        if (avg) atomicAdd(&lcnt[id[k]], 1u);
      }
    }
  }
  block_sync();
  for (uint32_t i = threadIdx.x; i < ngroups; i += GB_LD_THREADS) {
    const unsigned long long v = lacc[i];
    if (avg) {
      const unsigned int c = lcnt[i];
      if (c) { acc_fold(fold_op, flt, &gacc[i], v); atomicAdd(&gcnt[i], (unsigned long long)c); }
    } else if (v != acc_identity(fold_op) || fold_op == OP_SUM) {
      if (!(fold_op == OP_SUM && v == 0 && !flt)) acc_fold(fold_op, flt, &gacc[i], v);
    }
  }
}

// out mask byte b covers groups 8b..8b+7 (LSB first); ok == null means every group is valid.  The eight flags of a mask byte are ONE
// 8-byte load where they are all there (rmm allocations are aligned), and the missing groups are counted per wave: one atomic per
// wave instead of one per mask byte with a null group -- C5's averages have millions of all-null groups, and 2e6 atomics on one word
// took 0.24 ms of a 9.2 ms call (profiles/r5_l_c5_timeline_before.md)
__global__ __launch_bounds__(256) void gb_write_mask(const uint8_t *ok, uint32_t n, uint8_t *mask, unsigned int *nulls) {
  const uint32_t nbytes = (n + 7) / 8;
  const bool words = ok && ((uintptr_t)ok & 7u) == 0;
  unsigned int missing = 0;
  for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < nbytes; b += gridDim.x * 256) {
    uint8_t m = 0;
    if (!ok) {
      const uint32_t left = n - b * 8;
      m = left >= 8 ? 0xffu : (uint8_t)((1u << left) - 1u);
    } else if (words && b * 8 + 8 <= n) {
      unsigned long long w = reinterpret_cast<const unsigned long long *>(ok)[b];
      // every non-zero flag byte -> its top bit (the carry out of its low seven bits, or the top bit itself), moved to bit 0 of the
      // byte; the multiplication gathers bit 8k into bit 56 + k (no two partial products meet)
      w = ((((w & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | w) & 0x8080808080808080ull) >> 7;
      m = (uint8_t)((w * 0x0102040810204080ull) >> 56);
      missing += 8u - (unsigned int)__popc((unsigned int)m);
    } else {
      for (int k = 0; k < 8; ++k) {
        const uint32_t g = b * 8 + k;
        if (g >= n) break;
        if (ok[g]) m |= (uint8_t)(1u << k); else ++missing;
      }
    }
    mask[b] = m;
  }
  if (ok) {         // (kernel-uniform) one atomic per WORKGROUP: atomics on one word retire at ~3 ns each, whoever sends them
    __shared__ unsigned int wg_missing;
    if (threadIdx.x == 0) wg_missing = 0;
    block_sync();
    const unsigned int total = wave_reduce_add(missing);
    if (lane_id() == 0 && total) atomicAdd(&wg_missing, total);
    block_sync();
    if (threadIdx.x == 0 && wg_missing) atomicAdd(nulls, wg_missing);
  }
}

// write the output validity masks the caller supplied buffers for: keys are never null
// (rows with a null key were dropped), the aggregate is null for an all-null group
static gdf_error write_output_masks(int ncols, gdf_column **out_keys, gdf_column *out_agg, const uint8_t *agg_ok, uint32_t ngroups) {
  // the null counter is only ever raised for an aggregate with all-null groups (agg_ok): without one, nothing is allocated,
  // cleared or read back (one host round trip of a 0.4 ms C2 call)
  DevBuf nulls;
  if (agg_ok) {
    RMM_TRY(nulls.alloc(sizeof(unsigned int)));
    HIP_TRY(hipMemsetAsync(nulls.p, 0, sizeof(unsigned int), stream0()));
  }
  const int grid = stream_grid((ngroups + 7) / 8 + 1, 256);
  for (int c = 0; c < ncols; ++c) {
    out_keys[c]->null_count = 0;
    if (out_keys[c]->valid && ngroups)
      hipLaunchKernelGGL(gb_write_mask, dim3(grid), dim3(256), 0, stream0(), (const uint8_t *)nullptr, ngroups,
                         (uint8_t *)out_keys[c]->valid, nulls.as<unsigned int>());
  }
  out_agg->null_count = 0;
  if (out_agg->valid && ngroups) {
    hipLaunchKernelGGL(gb_write_mask, dim3(agg_ok && grid > 1024 ? 1024 : grid), dim3(256), 0, stream0(), agg_ok, ngroups, (uint8_t *)out_agg->valid,
                       nulls.as<unsigned int>());
    unsigned int h = 0;
    if (agg_ok) HIP_TRY(read_back(&h, nulls.p, sizeof(h)));
    out_agg->null_count = h;
  }
  HIP_CHECK_LAST();
  HIP_TRY(hipStreamSynchronize(stream0()));
  return GDF_SUCCESS;
}

// integer key columns too wide for the natural layout: try (value - min) in bit_length(max - min) bits
// min / max of the key IMAGES of every column over the rows whose element is valid, and whether a NaN was seen
__global__ __launch_bounds__(256) void gb_image_ranges(KeyTable t, long long *out, unsigned int *has_nan) {
  for (int c = 0; c < t.ncols; ++c) {
    long long lo = 0x7fffffffffffffffLL, hi = (long long)0x8000000000000000ULL;
    bool nan = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.nrows; i += (int64_t)gridDim.x * 256) {
      if (t.col[c].valid && !bit_is_set(t.col[c].valid, i)) continue;
      if (t.col[c].kind == K_F64) nan = nan || (((const uint64_t *)t.col[c].data)[i] & 0x7fffffffffffffffULL) > 0x7ff0000000000000ULL;
      if (t.col[c].kind == K_F32) nan = nan || (((const uint32_t *)t.col[c].data)[i] & 0x7fffffffu) > 0x7f800000u;
      const long long v = load_key_image(t.col[c], i);
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
    if (__ballot(nan) && lane_id() == 0) atomicExch(has_nan, 1u);
  }
}

static gdf_error gb_plan_range(const KeyTable &t, GbKeyPlan *plan, std::vector<long long> *ranges_out = nullptr) {
  bool any_float = false;
  for (int c = 0; c < t.ncols; ++c) any_float = any_float || t.col[c].kind == K_F32 || t.col[c].kind == K_F64;
  std::vector<long long> h(2 * t.ncols);
  if (any_float) {
    if (lab::path_on("GDF_GB_NO_FLOAT_IMAGE") || t.nrows == 0) return GDF_SUCCESS;
    for (int c = 0; c < t.ncols; ++c) { h[2 * c] = 0x7fffffffffffffffLL; h[2 * c + 1] = (long long)0x8000000000000000ULL; }
    DevBuf mm;
    RMM_TRY(mm.alloc(sizeof(long long) * 2 * t.ncols + sizeof(unsigned int)));
    HIP_TRY(hipMemcpyAsync(mm.p, h.data(), sizeof(long long) * 2 * t.ncols, hipMemcpyHostToDevice, stream0()));
    unsigned int *d_nan = (unsigned int *)(mm.as<long long>() + 2 * t.ncols);
    HIP_TRY(hipMemsetAsync(d_nan, 0, sizeof(unsigned int), stream0()));
    GDF_LAUNCH("gb_image_ranges", gb_image_ranges, dim3(stream_grid((size_t)t.nrows, 256 * 16)), dim3(256), 0, stream0(), t, mm.as<long long>(), d_nan);
    HIP_CHECK_LAST();
    unsigned int has_nan = 0;
    HIP_TRY(read_back(h.data(), mm.p, sizeof(long long) * 2 * t.ncols));
    HIP_TRY(read_back(&has_nan, d_nan, sizeof(has_nan)));
    if (has_nan) return GDF_SUCCESS;        // NaN keys: every NaN row is a group of its own, only the row-comparing path does that
  } else {
    GDF_TRY(key_ranges(t, h.data()));
    if (ranges_out) *ranges_out = h;        // the direct path wants the same numbers: one pass over the keys, not two
  }
  int total = 0;
  GbKeyPlan p{};
  for (int c = 0; c < t.ncols; ++c) {
    long long lo = h[2 * c], hi = h[2 * c + 1];
    if (lo > hi) lo = hi = 0;                                 // the column has no valid element
    const uint64_t span = (uint64_t)hi - (uint64_t)lo;
    int bits = 0;
    while (bits < 64 && (span >> bits) != 0) ++bits;
    p.bits[c] = bits;
    p.bias[c] = lo;
    total += bits;
    if (total > 63) return GDF_SUCCESS;                       // does not fit: keep what the caller had
  }
  for (int c = 0, below = total; c < t.ncols; ++c) { below -= p.bits[c]; p.shift[c] = below; }   // column 0 on top
  p.packed = 1;                                               // <= 63 bits: the reserved key 1 << 63 cannot occur
  p.ordered = 1;
  p.total_bits = total;
  *plan = p;
  return GDF_SUCCESS;
}


// The same layout GUESSED from a 65536-row prefix: every column's span is rounded up to the next power of two above the
// sample's span (the same number of key bits as the exact plan unless the prefix is unrepresentative), its minimum taken from
// the sample.  The full min / max pass (12 B per row: 2.1 of C5's 24 ms) is skipped; whoever packs keys with this plan
// must check every value against it (gbp_count does) and fall back to the exact plan on a violation.
static gdf_error gb_plan_range_sampled(const KeyTable &t, GbKeyPlan *plan, bool *ok) {
  *ok = false;
  for (int c = 0; c < t.ncols; ++c)
    if (t.col[c].kind == K_F32 || t.col[c].kind == K_F64) return GDF_SUCCESS;
  if (t.nrows < (1 << 16)) return GDF_SUCCESS;
  // sixteen windows of 4096 rows spread over the whole table (a 65536-row PREFIX saw a sliver of a sorted or time-ordered key
  // column's range: the guess was violated only after the count and the full scatter pass had run on it)
  std::vector<long long> h(2 * t.ncols);
  GDF_TRY(key_ranges(t, h.data(), 16, 4096));
  int total = 0;
  GbKeyPlan p{};
  for (int c = 0; c < t.ncols; ++c) {
    long long lo = h[2 * c], hi = h[2 * c + 1];
    if (lo > hi) return GDF_SUCCESS;                          // no valid element in the sample: no guess
    const uint64_t span = (uint64_t)hi - (uint64_t)lo;
    int bits = 0;
    while (bits < 64 && (span >> bits) != 0) ++bits;
    p.bits[c] = bits;
    p.bias[c] = lo;
    total += bits;
    if (total > 63) return GDF_SUCCESS;
  }
  for (int c = 0, below = total; c < t.ncols; ++c) { below -= p.bits[c]; p.shift[c] = below; }
  p.packed = 1;
  p.ordered = 1;
  p.total_bits = total;
  *plan = p;
  *ok = true;
  return GDF_SUCCESS;
}
// ---------------------------------------------------------------------------
// direct path (integer keys whose value RANGE is small -- C2: keys in [0, 10000)): no table at all.  The group id
// is the mixed-radix number sum_c (key_c - min_c) * stride_c (column 0 most significant, so ids ascend in
// lexicographic key order); one min/max pass finds the ranges, then every workgroup folds its rows into LDS
// accumulators indexed by that id -- one LDS atomic per row, no L2 dictionary lookups (the dense path spends
// ~1.3 of its 1.55 ms on C2 in those) -- and one workgroup writes the non-empty ids out in order, which is
// already the sorted result.
// ---------------------------------------------------------------------------
struct GbDirect {
  int ncols;
  long long lo[MAX_KEY_COLS];
  uint32_t span[MAX_KEY_COLS];
  uint32_t stride[MAX_KEY_COLS];
  uint32_t total;
};
constexpr uint32_t GB_DIRECT_MAX_IDS = 12288;          // 8-byte accumulator + 4-byte row count per id: 144 KiB of LDS

// id of row i, or 0xffffffff when some key element lies outside its column's [lo, lo + span) window (possible only
// when the windows were guessed from a sample)
__device__ __forceinline__ uint32_t direct_id(const KeyTable &t, const GbDirect &d, int64_t i) {
  uint32_t id = 0;
  bool inside = true;
  for (int c = 0; c < d.ncols; ++c) {
    const uint64_t off = (uint64_t)(load_signed(t.col[c], i) - d.lo[c]);
    inside = inside && off < d.span[c];
    id += (uint32_t)off * d.stride[c];
  }
  return inside ? id : 0xffffffffu;
}

// FASTKEY = 8 / 4: one int64 / int32 key column read directly, 0: direct_id().  FASTVAL = 8 / 4: an 8- / 4-byte value
// column read directly (integer or float by val.kind) and widened to the 64-bit accumulator image in registers, 0: acc_image()
template <int FASTKEY, int FASTVAL>
__global__ __launch_bounds__(GB_DENSE_THREADS) void gb_direct_aggregate(KeyTable t, GbDirect d, GbVal val, int op,
                                                                        unsigned long long *gacc, unsigned long long *gcnt,
                                                                        int64_t chunk, unsigned int *outside) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  unsigned long long *lacc = (unsigned long long *)gb_lds;
  unsigned int *lcnt = (unsigned int *)(lacc + d.total);
  const bool flt = is_flt(val.kind);
  const int fold_op = op == OP_AVG ? OP_SUM : op;
  for (uint32_t i = threadIdx.x; i < d.total; i += GB_DENSE_THREADS) { lacc[i] = acc_identity(fold_op); lcnt[i] = 0; }
  block_sync();
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
  const long long lo0 = d.lo[0];
  for (int64_t base = begin; base < end; base += (int64_t)GB_DENSE_THREADS * GB_DENSE_BATCH) {
    uint32_t id[GB_DENSE_BATCH];
    uint64_t img[GB_DENSE_BATCH];
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {                       // all HBM loads first, from clamped addresses
      const int64_t i = base + (int64_t)k * GB_DENSE_THREADS + threadIdx.x;
      const int64_t ic = i < end ? i : end - 1;
      if (FASTKEY) {
        const long long kv = FASTKEY == 8 ? ((const long long *)t.col[0].data)[ic] : (long long)((const int32_t *)t.col[0].data)[ic];
        const uint64_t off = (uint64_t)(kv - lo0);
        id[k] = off < d.total ? (uint32_t)off : 0xffffffffu;
      } else {
        id[k] = direct_id(t, d, ic);
      }
      img[k] = FASTVAL == 8 ? ((const uint64_t *)val.data)[ic] : (FASTVAL == 4 ? (uint64_t)((const uint32_t *)val.data)[ic] : acc_image(fold_op, val, ic));
    }
    if (FASTVAL) {
#pragma unroll
      for (int k = 0; k < GB_DENSE_BATCH; ++k) {
        if (FASTVAL == 4)     // widen: float -> the bits of its double, int32 -> sign-extended
          img[k] = flt ? (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)img[k])) : (uint64_t)(int64_t)(int32_t)(uint32_t)img[k];
        if (fold_op == OP_COUNT) img[k] = 1;
        else if (fold_op == OP_MIN || fold_op == OP_MAX)
          img[k] = flt ? ord_f64(__longlong_as_double((long long)img[k])) : ord_i64((int64_t)img[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < GB_DENSE_BATCH; ++k) {
      if (base + (int64_t)k * GB_DENSE_THREADS + threadIdx.x < end) {
        if (id[k] == 0xffffffffu) { *outside = 1u; continue; }     // the guessed window was too small: the host repeats with exact ranges
        acc_fold(fold_op, flt, &lacc[id[k]], img[k]);
        atomicAdd(&lcnt[id[k]], 1u);
      }
    }
  }
  block_sync();
  for (uint32_t i = threadIdx.x; i < d.total; i += GB_DENSE_THREADS) {
    const unsigned int c = lcnt[i];
    if (c) { acc_fold(fold_op, flt, &gacc[i], lacc[i]); atomicAdd(&gcnt[i], (unsigned long long)c); }
  }
}

// one workgroup: compacts the non-empty ids in ascending order (= lexicographic key order) and finishes the aggregates
// last[id] = max(row + 1) over the rows of group id (0: no row): LDS atomicMax per row, one global atomicMax per touched id and
// workgroup.  Only for SORT-method calls that asked for out_col_indices; reads the key columns once more (8 of C2's 16 B per row).
__global__ __launch_bounds__(GB_DENSE_THREADS) void gb_direct_last_rows(KeyTable t, GbDirect d, unsigned int *__restrict__ glast, int64_t chunk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  unsigned int *llast = (unsigned int *)gb_lds;
  for (uint32_t i = threadIdx.x; i < d.total; i += GB_DENSE_THREADS) llast[i] = 0;
  block_sync();
  const int64_t begin = (int64_t)blockIdx.x * chunk, end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
  for (int64_t i = begin + threadIdx.x; i < end; i += GB_DENSE_THREADS) {
    const uint32_t id = direct_id(t, d, i);
    if (id != 0xffffffffu) atomicMax(&llast[id], (unsigned int)i + 1u);
  }
  block_sync();
  for (uint32_t i = threadIdx.x; i < d.total; i += GB_DENSE_THREADS)
    if (llast[i]) atomicMax(&glast[i], llast[i]);
}

__global__ __launch_bounds__(1024) void gb_direct_extract(KeyTable t, GbDirect d, GbOut o, int op, const unsigned long long *gacc,
                                                          const unsigned long long *gcnt, unsigned int *out_groups) {
  // Workgroup b writes the non-empty ids of [1024 b, 1024 b + 1024), one id per thread.  Where its output starts -- the number
  // of non-empty ids before its block -- it counts itself: at most 11 more loads per thread, all in flight together.  (As ONE
  // workgroup walking twelve ids per thread this kernel took 30-38 us of C2's 440.)
  __shared__ uint32_t wave_tot[1024 / WAVE], wave_own[1024 / WAVE];
  constexpr uint32_t MAXB = (GB_DIRECT_MAX_IDS + 1023) / 1024;
  const uint32_t nblocks = (d.total + 1023) / 1024;
  uint32_t before_blocks = 0;
#pragma unroll
  for (uint32_t k = 0; k < MAXB; ++k) {
    const uint32_t id = k * 1024 + threadIdx.x;
    // blocks before mine count towards my base (the last workgroup's base + its own count is the number of groups)
    const bool wanted = k < blockIdx.x && id < d.total;
    before_blocks += (wanted && gcnt[id] != 0) ? 1u : 0u;
  }
  const uint32_t id = blockIdx.x * 1024 + threadIdx.x;
  const unsigned long long cnt = id < d.total ? gcnt[id] : 0ULL;
  const unsigned long long acc = id < d.total ? gacc[id] : 0ULL;
  const unsigned long long m = __ballot(cnt != 0);
  const uint32_t bsum = wave_reduce_add(before_blocks);
  if (lane_id() == 0) { wave_tot[threadIdx.x / WAVE] = bsum; wave_own[threadIdx.x / WAVE] = (uint32_t)__popcll(m); }
  block_sync();
  uint32_t base = 0, own_before = 0, own_total = 0;
  for (int w = 0; w < 1024 / WAVE; ++w) {
    base += wave_tot[w];
    if (w < (int)(threadIdx.x / WAVE)) own_before += wave_own[w];
    own_total += wave_own[w];
  }
  if (cnt) {
    const uint32_t pos = base + own_before + mask_rank(m);
    for (int c = 0; c < d.ncols; ++c) {
      const uint64_t bits = (uint64_t)(d.lo[c] + (long long)((id / d.stride[c]) % d.span[c]));
      switch (t.col[c].width) {
        case 1: ((uint8_t *)o.key_out[c])[pos] = (uint8_t)bits; break;
        case 2: ((uint16_t *)o.key_out[c])[pos] = (uint16_t)bits; break;
        case 4: ((uint32_t *)o.key_out[c])[pos] = (uint32_t)bits; break;
        default: ((uint64_t *)o.key_out[c])[pos] = bits; break;
      }
    }
    store_result(o, op, pos, acc, cnt);
    if (o.indices) o.indices[pos] = (size_t)(o.last_rows[id] - 1u);
  }
  if (blockIdx.x == nblocks - 1 && threadIdx.x == 0) *out_groups = base + own_total;
}

// ---------------------------------------------------------------------------
// sorted path (packed keys, MANY groups -- C5 has ~1.6e7): a global hash table of that
// size lives in HBM and every row pays dependent random atomics on it (measured:
// 85 ms per 1e8 rows at 1e7 groups).  Instead the (packed key, value image) pairs are
// radix sorted on the few key bits the range layout needs (C5: 25 bits = 3 passes of
// streaming HBM traffic) and reduced by segments; nothing is random-access.
//   key     = packed << vbit | value_valid      (vbit = 1 only when the value column has a mask;
//             a row with a null key gets the bit just above the packed field and sorts behind every real key)
//   payload = accumulator image of the value (identity if the value is null)
// ---------------------------------------------------------------------------
template <class K>      // K = uint32_t when key bits + valid bit + null bit fit 32 (12-byte pairs), else uint64_t
__global__ __launch_bounds__(256) void gb_sorted_make_pairs(KeyTable t, GbKeyPlan plan, GbVal val, int fold_op, int vbit,
                                                            uint64_t null_key, K *__restrict__ keys, uint64_t *__restrict__ payload,
                                                            unsigned long long *__restrict__ varying, unsigned int *__restrict__ dropped) {
  uint64_t diff = 0;
  unsigned int drop = 0;
  const uint64_t k0 = gb_pack(t, plan, 0) << vbit;      // any common reference value serves the OR-reduction
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < t.nrows; i += (int64_t)gridDim.x * 256) {
    uint64_t k, p;
    if (!row_valid(t, i)) { k = null_key; p = 0; ++drop; }
    else {
      const bool vok = !val.valid || bit_is_set(val.valid, i);
      k = (gb_pack(t, plan, i) << vbit) | (uint64_t)(vbit && vok);
      p = vok ? acc_image(fold_op, val, i) : acc_identity(fold_op);
    }
    keys[i] = (K)k;
    payload[i] = p;
    diff |= k ^ k0;
  }
  for (int d = 1; d < WAVE; d <<= 1) {
    const uint32_t lo = __shfl_xor((uint32_t)diff, d), hi = __shfl_xor((uint32_t)(diff >> 32), d);
    diff |= ((uint64_t)hi << 32) | lo;
  }
  drop = wave_reduce_add(drop);
  if (lane_id() == 0) {
    if (diff) atomicOr(varying, (unsigned long long)diff);
    if (drop) atomicAdd(dropped, drop);
  }
}

__global__ __launch_bounds__(256) void gb_sorted_heads(const uint64_t *__restrict__ keys, int vbit, uint32_t *__restrict__ head, uint32_t n) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    head[i] = (i == 0 || (keys[i] >> vbit) != (keys[i - 1] >> vbit)) ? 1u : 0u;
}
__global__ __launch_bounds__(256) void gb_sorted_starts(const uint32_t *__restrict__ gid, uint32_t *__restrict__ start, uint32_t n, uint32_t ngroups) {
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
    if (i == 0 || gid[i] != gid[i - 1]) start[gid[i] - 1] = i;
  if (blockIdx.x == 0 && threadIdx.x == 0) start[ngroups] = n;
}

__device__ __forceinline__ uint64_t img_merge(int op, bool flt, uint64_t a, uint64_t b) {
  if (op == OP_MIN) return a < b ? a : b;
  if (op == OP_MAX) return a > b ? a : b;
  if (flt && op != OP_COUNT) return (uint64_t)__double_as_longlong(__longlong_as_double((long long)a) + __longlong_as_double((long long)b));
  return a + b;
}
__device__ __forceinline__ uint64_t shfl_up_u64(uint64_t v, int d) {
  return ((uint64_t)__shfl_up((uint32_t)(v >> 32), d) << 32) | __shfl_up((uint32_t)v, d);
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, int l) {
  return ((uint64_t)__shfl((uint32_t)(v >> 32), l) << 32) | __shfl((uint32_t)v, l);
}

// acc[g] (and cnt[g] = number of valid values, COUNTED) over the sorted pairs: every wave walks
// 64 x GB_SEG_ROUNDS consecutive pairs; per round a segmented shuffle scan on the group id folds the
// lanes of one group, closed segments leave with one atomic, the open one rides in registers.
constexpr int GB_SEG_ROUNDS = 16;
template <bool COUNTED>
__global__ __launch_bounds__(256) void gb_sorted_reduce(const uint64_t *__restrict__ keys, const uint64_t *__restrict__ payload,
                                                        const uint32_t *__restrict__ gid, int op, bool flt,
                                                        unsigned long long *__restrict__ acc, unsigned long long *__restrict__ cnt,
                                                        uint32_t n) {
  const int lane = lane_id();
  const uint64_t begin = ((uint64_t)blockIdx.x * 4 + threadIdx.x / WAVE) * (uint64_t)(WAVE * GB_SEG_ROUNDS);
  if (begin >= n) return;
  uint32_t carry_gid = 0, carry_c = 0;      // gid 0 = no open segment (ids are 1-based)
  uint64_t carry = 0;
  for (int r = 0; r < GB_SEG_ROUNDS; ++r) {
    if (begin + (uint64_t)r * WAVE >= n) break;                 // wave-uniform
    const uint64_t i = begin + (uint64_t)r * WAVE + lane;
    const bool live = i < n;
    const uint32_t j = live ? (uint32_t)i : n - 1;
    const uint32_t g = live ? gid[j] : 0xffffffffu;             // dead lanes form their own trailing segment
    uint64_t v = payload[j];
    uint32_t c = COUNTED ? (uint32_t)(keys[j] & 1ULL) : 0u;
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const uint64_t uv = shfl_up_u64(v, d);
      const uint32_t ug = __shfl_up(g, d);
      const uint32_t uc = COUNTED ? __shfl_up(c, d) : 0u;
      if (lane >= d && ug == g) { v = img_merge(op, flt, v, uv); c += uc; }
    }
    const uint32_t gnext = __shfl_down(g, 1);
    const bool tail = (lane == WAVE - 1) || (gnext != g);
    const uint32_t g0 = __shfl(g, 0);
    if (carry_gid != 0 && carry_gid != g0) {                    // the open segment ended with the previous round
      if (lane == 0) {
        acc_fold(op, flt && op != OP_COUNT, &acc[carry_gid - 1], carry);
        if (COUNTED && carry_c) atomicAdd(&cnt[carry_gid - 1], (unsigned long long)carry_c);
      }
      carry_gid = 0;
    }
    if (tail && carry_gid != 0 && g == carry_gid) { v = img_merge(op, flt, v, carry); c += carry_c; }
    const uint32_t glast = __shfl(g, WAVE - 1);
    const uint64_t vlast = shfl_u64(v, WAVE - 1);
    const uint32_t clast = COUNTED ? __shfl(c, WAVE - 1) : 0u;
    if (tail && live && lane != WAVE - 1) {
      acc_fold(op, flt && op != OP_COUNT, &acc[g - 1], v);
      if (COUNTED && c) atomicAdd(&cnt[g - 1], (unsigned long long)c);
    }
    if (glast != 0xffffffffu) { carry_gid = glast; carry = vlast; carry_c = clast; }
    else carry_gid = 0;
  }
  if (carry_gid != 0 && lane == 0) {
    acc_fold(op, flt && op != OP_COUNT, &acc[carry_gid - 1], carry);
    if (COUNTED && carry_c) atomicAdd(&cnt[carry_gid - 1], (unsigned long long)carry_c);
  }
}

__global__ __launch_bounds__(256) void gb_sorted_extract(KeyTable t, GbKeyPlan plan, const uint64_t *__restrict__ keys, int vbit,
                                                         const uint32_t *__restrict__ start, uint32_t ngroups, GbOut o, int op,
                                                         const unsigned long long *__restrict__ acc,
                                                         const unsigned long long *__restrict__ cnt) {
  for (uint32_t g = blockIdx.x * 256 + threadIdx.x; g < ngroups; g += gridDim.x * 256) {
    const uint32_t s = start[g], e = start[g + 1];
    const uint64_t key = keys[s] >> vbit;
    for (int c = 0; c < t.ncols; ++c) gb_unpack_store(t, plan, key, c, o.key_out[c], g);
    store_result(o, op, g, acc[g], cnt ? cnt[g] : (unsigned long long)(e - s));
  }
}

// ---------------------------------------------------------------------------
// partitioned direct path: the sorted path's cheaper sibling for ORDERED packed keys (range layout).  Only the
// HIGH bits of the packed key are radix sorted -- just enough that what is left, the low GB_PART_ID_BITS bits,
// indexes LDS accumulators directly.  C5's 24-bit key: 11 high bits = 2 radix passes instead of 3, and the
// heads / scan / segmented-reduce stages (16 of its 73 ms) become one streaming pass with an LDS atomic per row.
// Keys of 14..22 bits (1e5 - 4e6 groups) need a single 8/9-bit pass.
// ---------------------------------------------------------------------------
constexpr int GB_PART_ID_BITS = 13;            // 8192 ids: 8 B accumulator + 4 B row count + 4 B valid count = 128 KiB of LDS
constexpr int GB_PART_MAX_BITS = 13;           // at most 8192 partitions (1 GiB of global cells)
constexpr uint32_t GB_PART_UNIT_ROWS = 1u << 18;
constexpr uint32_t GB_PART_TRASH = 64;          // LDS slots behind the ids for records that are not folded (gb_part_aggregate)
struct GbPartUnit { uint32_t begin, count, part, pad; };

// pstart[p] = first sorted position whose partition id (key >> low) is >= p, for p in [0, P]
template <class K>
__global__ __launch_bounds__(256) void gb_part_bounds(const K *__restrict__ keys, uint32_t n, int low, uint32_t P,
                                                      uint32_t *__restrict__ pstart) {
  for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p <= P; p += gridDim.x * 256) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
      const uint32_t mid = lo + (hi - lo) / 2;
      if (((uint64_t)keys[mid] >> low) < (uint64_t)p) lo = mid + 1; else hi = mid;
    }
    pstart[p] = lo;
  }
}

// SPECULATIVE record layout (round 4): no count pass.  Partition q owns G segments of cap[q] records, one per scatter workgroup,
// at record index (qprefix[q] * G + w * cap[q]); a workgroup appends to its own segments only -- no atomics, no histogram -- and
// leaves its fill counts in fill[q * G + w].  The capacities come from the strided sample (expected share + 5 sigma of the
// estimate + 6 sigma of a workgroup's own fluctuation).  A segment that would overflow raises flags[2]; every workgroup sees the
// flag at its next tile and stops, and the host repeats the call on the exact layout (gbp_count + scan) -- clustered input, e.g.
// rows sorted by key, ends there after a few tiles.  gb_part_aggregate streams a partition's whole region and masks the slack.
struct GbSpec {
  const uint32_t *qprefix;         // [P + 1] exclusive prefix of cap[] (in records PER WORKGROUP); nullptr: the exact layout
  const uint32_t *cap;             // [P]
  uint32_t *fill;                  // [P * G]
  uint32_t G;                      // workgroups of the scatter kernel (= segments per partition)
  // xcd != 0: ONE segment per partition and XCD (G = 8) instead of one per workgroup.  The workgroups of an XCD append to it
  // together: a (tile, partition) run claims its place with ONE returning atomic on fill[q * 8 + xcc] -- relaxed, AGENT scope (the
  // counter is shared across workgroups; no fence: an acquire / release pair around it cost +3 ms in round 3) -- where xcc is read
  // from HW_REG_XCC_ID, so that the counter a workgroup uses is one only its own XCD touches, whatever the dispatcher does.  Why: a
  // tail partition gets ~1 record per tile and workgroup; with 32 workgroups behind one write front its 128-byte line fills in
  // microseconds, inside the L2, instead of leaving as ten partial lines.  (Where the atomics execute: the L2 counters of C5 show
  // every one of the 1.28e8 forwarded to the memory side as an atomic request, TCC_EA0_ATOMIC = TCC_ATOMIC,
  // profiles/r4_u_c5_l2_atomic_counters.txt -- they are resolved where any XCD would see them; choosing the counter by XCD is what
  // keeps an XCD's appends together, not what makes them atomic.)
  int xcd;
};

// one (32-bit packed key, 64-bit accumulator image) pair as the fused partition pass writes it: 12 bytes, ONE store per row
// (image first: words 0..1 of a 12-byte load or store are an even-aligned register pair, what 64-bit LDS / VALU operands need on
// gfx950 -- with the key in front every record cost two register copies on either side)
struct __attribute__((aligned(4))) GbRec { uint32_t lo, hi, key; };

// REC: `keys` points to GbRec records (the fused partition pass), `payload` is unused
template <bool VBIT, class K, bool REC = false>
__global__ __launch_bounds__(GB_DENSE_THREADS) void gb_part_aggregate(const K *__restrict__ keys, const uint64_t *__restrict__ payload,
                                                                      const GbPartUnit *__restrict__ units, int id_bits, int op, bool flt,
                                                                      unsigned long long *gacc, unsigned int *grows, unsigned int *gvalid,
                                                                      GbSpec spec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gb_lds[];
  __shared__ uint32_t seg_fill[NUM_CU];                             // speculative layout: the fill counts of this partition's segments
  const uint32_t ids = 1u << id_bits;
  // + GB_PART_TRASH slots behind the ids, one per lane: a record that is not folded (a dead slot of the speculative layout, a null
  // value) goes there instead of around a branch -- the loop body is straight-line code, so the next batch's loads stay in flight
  // under this batch's LDS atomics (a branch per record is a basic block per record, each with its own wait for every load)
  unsigned long long *lacc = (unsigned long long *)gb_lds;
  unsigned int *lrows = (unsigned int *)(lacc + ids + GB_PART_TRASH);
  unsigned int *lvalid = lrows + ids + GB_PART_TRASH;               // VBIT only
  const GbPartUnit u = units[blockIdx.x];
  for (uint32_t i = threadIdx.x; i < ids; i += GB_DENSE_THREADS) {
    lacc[i] = acc_identity(op);
    lrows[i] = 0;
    if (VBIT) lvalid[i] = 0;
  }
  // speculative layout (REC only): the unit is a slice of the partition's REGION -- G segments of `cap` slots, the first fill[w] of
  // segment w hold records.  magic = ceil(2^32 / cap): slot / cap by one multiply and a correction
  uint32_t sp_cap = 0, sp_magic = 0, sp_first = 0;
  if (REC && spec.qprefix) {
    for (uint32_t w = threadIdx.x; w < spec.G; w += GB_DENSE_THREADS) seg_fill[w] = spec.fill[(size_t)u.part * spec.G + w];
    sp_cap = spec.cap[u.part];
    sp_magic = (uint32_t)((((uint64_t)1 << 32) + sp_cap - 1) / sp_cap);
    sp_first = u.begin - spec.qprefix[u.part] * spec.G;             // the unit's first slot, counted from the region's start
  }
  block_sync();
  const uint32_t mask = ids - 1;
  const uint32_t trash = ids + (threadIdx.x & (GB_PART_TRASH - 1u));
  // Software pipeline (round 4): batch n + 1 is requested before batch n is folded into the LDS accumulators, so the HBM round trip
  // of the one resident workgroup hides behind its LDS atomics instead of alternating with them.
  // a batch as it is loaded: a record stays the three words of its 12-byte load until it is folded (the 64-bit image is an ALIGNED
  // register pair for ds_add_u64, words 1..2 of a load are not: built in front of the scheduling barrier below, the copy that
  // aligns it waits for the load that has only just been issued)
  struct Batch {
    uint32_t live;
    uint32_t rk[REC ? GB_DENSE_BATCH : 1], rlo[REC ? GB_DENSE_BATCH : 1], rhi[REC ? GB_DENSE_BATCH : 1];
    K k[REC ? 1 : GB_DENSE_BATCH];
    uint64_t v[REC ? 1 : GB_DENSE_BATCH];
  };
  auto fetch = [&](auto slack, uint32_t base, Batch &bt) {      // slack: the speculative layout's dead slots are masked
    constexpr bool SLACK = decltype(slack)::value;
    uint32_t livemask = 0;
#pragma unroll
    for (int b = 0; b < GB_DENSE_BATCH; ++b) {
      bool live = base + b * GB_DENSE_THREADS + threadIdx.x < u.count;
      if constexpr (SLACK) {                      // a slot behind its segment's fill count holds nothing
        const uint32_t o = sp_first + base + b * GB_DENSE_THREADS + threadIdx.x;
        uint32_t w = __umulhi(o, sp_magic);
        uint32_t off = o - w * sp_cap;
        if ((int32_t)off < 0) { --w; off += sp_cap; }          // the estimate is at most one too large (o < 2^31)
        const uint32_t have = seg_fill[w < spec.G ? w : 0];     // read whether the slot is live or not: no branch
        live = live & (off < have);
      }
      livemask |= (uint32_t)live << b;
    }
    bt.live = livemask;
#pragma unroll
    for (int b = 0; b < GB_DENSE_BATCH; ++b) {                       // all HBM loads first; a dead slot re-reads the unit's first one
      const uint32_t i = base + b * GB_DENSE_THREADS + threadIdx.x;  // (the slack of the speculative layout costs no HBM traffic)
      const uint32_t ic = u.begin + (((livemask >> b) & 1u) ? i : 0u);
      if constexpr (REC) {
        const GbRec r = reinterpret_cast<const GbRec *>(keys)[ic];
        bt.rk[b] = r.key;
        bt.rlo[b] = r.lo;
        bt.rhi[b] = r.hi;
      } else {
        bt.k[b] = keys[ic];
        bt.v[b] = payload[ic];
      }
    }
  };
  auto run = [&](auto fold_one) {
    auto fold = [&](const Batch &bt) {
#pragma unroll
      for (int b = 0; b < GB_DENSE_BATCH; ++b) {
        K key;
        uint64_t img;
        if constexpr (REC) { key = (K)bt.rk[b]; img = ((uint64_t)bt.rhi[b] << 32) | bt.rlo[b]; }
        else { key = bt.k[b]; img = bt.v[b]; }
        const bool live = (bt.live >> b) & 1u;
        const uint32_t id = live ? (uint32_t)(key >> (VBIT ? 1 : 0)) & mask : trash;
        const uint32_t idv = (!VBIT || (key & 1)) ? id : trash;
        atomicAdd(&lrows[id], 1u);
        fold_one(&lacc[idv], img);
        if (VBIT) atomicAdd(&lvalid[idv], 1u);
      }
    };
    constexpr uint32_t STEP = GB_DENSE_THREADS * GB_DENSE_BATCH;
    if (!u.count) return;
    auto loop = [&](auto slack) {
      Batch A, B;
      fetch(slack, 0, A);
      for (uint32_t base = 0; base < u.count; base += 2 * STEP) {
        // (a batch behind the unit's end is all dead slots: every lane re-reads the unit's first record, one cached line)
        fetch(slack, base + STEP, B);
        __builtin_amdgcn_sched_barrier(0);           // the requests leave BEFORE the fold below (the scheduler sinks them to their uses)
        fold(A);
        __builtin_amdgcn_sched_barrier(0);
        fetch(slack, base + 2 * STEP, A);
        __builtin_amdgcn_sched_barrier(0);
        fold(B);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (REC && sp_cap) loop(std::true_type{});     // (workgroup-uniform; decided once, not once per record)
    else loop(std::false_type{});
  };
  switch (op) {        // the fold is chosen once per kernel, not once per record
    case OP_MIN: run([](unsigned long long *a, uint64_t x) { atomicMin(a, (unsigned long long)x); }); break;
    case OP_MAX: run([](unsigned long long *a, uint64_t x) { atomicMax(a, (unsigned long long)x); }); break;
    case OP_COUNT: run([](unsigned long long *a, uint64_t x) { atomicAdd(a, (unsigned long long)x); }); break;
    default:
      if (flt) run([](unsigned long long *a, uint64_t x) { atomicAdd((double *)a, __longlong_as_double((long long)x)); });
      else run([](unsigned long long *a, uint64_t x) { atomicAdd(a, (unsigned long long)x); });
  }
  block_sync();
  const size_t cell0 = (size_t)u.part << id_bits;
  if (u.pad == 1u) {
    // the ONLY unit of its partition, and nobody else writes these cells (no hot window in them): plain coalesced stores into the
    // pre-filled cells instead of up to three global atomics per id -- a tail partition of C5 is one unit of ~36 k records that
    // touches most of its 8192 ids, 0.67 atomics per record
    for (uint32_t i = threadIdx.x; i < ids; i += GB_DENSE_THREADS) {
      const unsigned int r = lrows[i];
      if (!r) continue;
      grows[cell0 + i] = r;
      if (VBIT) {
        const unsigned int c = lvalid[i];
        if (c) { gvalid[cell0 + i] = c; gacc[cell0 + i] = lacc[i]; }
      } else {
        gacc[cell0 + i] = lacc[i];
      }
    }
    return;
  }
  for (uint32_t i = threadIdx.x; i < ids; i += GB_DENSE_THREADS) {
    const unsigned int r = lrows[i];
    if (!r) continue;
    atomicAdd(&grows[cell0 + i], r);
    if (VBIT) {
      const unsigned int c = lvalid[i];
      if (c) { atomicAdd(&gvalid[cell0 + i], c); acc_fold(op, flt, &gacc[cell0 + i], lacc[i]); }
    } else {
      acc_fold(op, flt, &gacc[cell0 + i], lacc[i]);
    }
  }
}

// ---------------------------------------------------------------------------
// Fused partitioning for the partitioned path (32-bit sort keys, <= 2^GBP_MAX_PART_BITS partitions): ONE pass that reads the
// raw key / value columns, builds the (packed key, accumulator image) pair and drops it into the partition named by the high
// key bits -- instead of gb_sorted_make_pairs (20 B in, 12 B out) followed by two stable radix passes (2 x 24 B per row plus
// their digit counts).  The pairs of a partition need no order: gb_part_aggregate indexes its LDS accumulators with the LOW
// key bits.  Per row: gbp_count 12 B (keys), gbp_scatter 20.1 B in + 12 B out, aggregation 12 B = 56 B, against 116 B for C5
// through the radix sort (24 -> ~12 ms).
//   gbp_count    per-chunk partition histogram hist[p * nchunks + chunk] (LDS counters; lanes that share the partition id of their
//                wave's first live lane are counted with one ballot -- C5's Zipf keys put 45 % of the rows into
//                partition 0), the number of rows dropped for a null key, and a flag when a key falls outside the plan's
//                ranges (possible only with sample-guessed ranges);
//   gbp_scatter  12288-row tiles, ranks from wave-wide match-any ballots (one LDS atomic per distinct partition and wave: the
//                hot partition would otherwise serialise thousands of same-address atomics per tile), the key and then the
//                payload staged through LDS in partition order so that a wave stores runs of consecutive addresses.
// ---------------------------------------------------------------------------
constexpr int GBP_MAX_PART_BITS = 11;
constexpr int GBP_MAX_PARTS = 1 << GBP_MAX_PART_BITS;
constexpr int GBP_RANK_TRASH = 64;            // counters behind hist[MAX_PARTS], one per lane: where a row that is not ranked adds its 1 (gbp_rank_plain)
constexpr int GBP_THREADS = 1024;
constexpr int GBP_ITEMS = 8;
constexpr int GBP_TILE = GBP_THREADS * GBP_ITEMS;
// the scatter kernel's shape: 1024 threads x 8 rows = 8192-row tiles, 115 KB of LDS, one workgroup per CU.  Two 512-thread
// workgroups per CU on 4096-row tiles (72 KB each) overlap their phases but halve the run lengths: 13.9 instead of 11.9 ms on
// C5 -- the store-REQUEST rate of the short (tile, partition) runs is the wall, not the phase structure; 12 rows per thread
// (longer runs) spill
constexpr int GBP_SC_THREADS = 1024;
constexpr int GBP_SC_TILE = GBP_SC_THREADS * GBP_ITEMS;
constexpr int GBP_MAX_CHUNKS = 1024;       // hist columns at most (chunks are whole tiles)

// packed keys of N rows (clamped row numbers in src[]) with the loads of one column issued together; ok[k] = false when the
// row has a null key element or a value outside the plan's range for its column
template <int N>
__device__ __forceinline__ void gbp_pack(const KeyTable &t, const GbKeyPlan &p, const uint32_t (&src)[N], uint64_t (&key)[N], bool (&ok)[N],
                                         bool (&inside)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) { key[k] = 0; ok[k] = true; inside[k] = true; }
  for (int c = 0; c < t.ncols; ++c) {
    long long v[N];
    const void *data = t.col[c].data;
    switch (t.col[c].kind) {
      case K_I8:
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = ((const int8_t *)data)[src[k]];
        break;
      case K_I16:
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = ((const int16_t *)data)[src[k]];
        break;
      case K_I32:
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = ((const int32_t *)data)[src[k]];
        break;
      case K_F32:
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = f32_image(((const uint32_t *)data)[src[k]]);
        break;
      case K_F64:
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = f64_image(((const uint64_t *)data)[src[k]]);
        break;
      default:
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = ((const long long *)data)[src[k]];
    }
    const int bits = p.bits[c], shift = p.shift[c];
    const long long bias = p.bias[c];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const uint64_t u = (uint64_t)(v[k] - bias);
      inside[k] = inside[k] && (bits >= 64 || (u >> bits) == 0);
      key[k] |= (u & low_mask(bits)) << shift;
    }
    if (t.col[c].valid) {
      uint8_t m[N];
#pragma unroll
      for (int k = 0; k < N; ++k) m[k] = t.col[c].valid[src[k] >> 3];
#pragma unroll
      for (int k = 0; k < N; ++k) ok[k] = ok[k] && ((m[k] >> (src[k] & 7)) & 1);
    }
  }
}

// raw 64-bit images of N rows of one key column (sign-extended integers, ordered float images), all loads issued together
template <int N>
__device__ __forceinline__ void gbp_load_col(const ColView &col, const uint32_t (&src)[N], long long (&v)[N]) {
  const void *data = col.data;
  switch (col.kind) {
    case K_I8:
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = ((const int8_t *)data)[src[k]];
      break;
    case K_I16:
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = ((const int16_t *)data)[src[k]];
      break;
    case K_I32:
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = ((const int32_t *)data)[src[k]];
      break;
    case K_F32:
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = f32_image(((const uint32_t *)data)[src[k]]);
      break;
    case K_F64:
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = f64_image(((const uint64_t *)data)[src[k]]);
      break;
    default:
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = ((const long long *)data)[src[k]];
  }
}
// gbp_pack for plans whose packed key fits 32 bits (the fused scatter's case), shaped for a kernel that sits at its register
// limit: the words of the first TWO key columns are requested together (a second column otherwise waits for the first one's
// round trip to HBM: the column loop cannot be unrolled), the key is accumulated in 32 bits, and the per-row flags are two bit
// masks instead of sixteen bools.  okmask bit k: no null key element in row k; outside bit k: a value of row k lies outside the plan's range for its column
// K0 / K1: the kinds of the first two key columns when the caller knows them at compile time (K_I32 or K_I64; K1 = -2: there
// is no second column), -1: read t.col[c].kind.  The type switch of the dynamic form costs more than its branches: the
// registers its cases load into are shared, so the waitcnt pass sees "maybe pending" loads on every path and makes the
// requests of column 0 wait for everything issued before them (the value column) -- one HBM round trip per tile that the
// static form does not have (all of a tile's requests leave together).
template <int N, int K0 = -1, int K1 = -1>
__device__ __forceinline__ void gbp_pack32(const KeyTable &t, const GbKeyPlan &p, const uint32_t (&src)[N], uint32_t (&key)[N], uint32_t &okmask,
                                           uint32_t &outside) {
#pragma unroll
  for (int k = 0; k < N; ++k) key[k] = 0;
  okmask = (1u << N) - 1u;
  outside = 0;
  long long v0[N], v1[N];
  if constexpr (K0 == K_I32) {
#pragma unroll
    for (int k = 0; k < N; ++k) v0[k] = ((const int32_t *)t.col[0].data)[src[k]];
  } else if constexpr (K0 == K_I64) {
#pragma unroll
    for (int k = 0; k < N; ++k) v0[k] = ((const long long *)t.col[0].data)[src[k]];
  } else {
    gbp_load_col<N>(t.col[0], src, v0);
  }
  if constexpr (K1 == K_I32) {
#pragma unroll
    for (int k = 0; k < N; ++k) v1[k] = ((const int32_t *)t.col[1].data)[src[k]];
  } else if constexpr (K1 == K_I64) {
#pragma unroll
    for (int k = 0; k < N; ++k) v1[k] = ((const long long *)t.col[1].data)[src[k]];
  } else if constexpr (K1 == -1) {
    if (t.ncols > 1) gbp_load_col<N>(t.col[1], src, v1);
  }
  auto fold = [&](int c, const long long (&v)[N]) {
    const int bits = p.bits[c], shift = p.shift[c];
    const long long bias = p.bias[c];
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const uint64_t u = (uint64_t)(v[k] - bias);
      outside |= (uint32_t)(bits < 64 && (u >> bits) != 0) << k;
      key[k] |= (uint32_t)(u & low_mask(bits)) << shift;
    }
    if (t.col[c].valid) {
      uint8_t m[N];
#pragma unroll
      for (int k = 0; k < N; ++k) m[k] = t.col[c].valid[src[k] >> 3];
#pragma unroll
      for (int k = 0; k < N; ++k) okmask &= ~((uint32_t)(((m[k] >> (src[k] & 7)) & 1) ^ 1) << k);
    }
  };
  fold(0, v0);
  if constexpr (K1 >= 0) fold(1, v1);
  else if constexpr (K1 == -1) {
    if (t.ncols > 1) fold(1, v1);
    for (int c = 2; c < t.ncols; ++c) {
      gbp_load_col<N>(t.col[c], src, v0);
      fold(c, v0);
    }
  }
}

// ---------------------------------------------------------------------------
// HOT WINDOW (round 4).  Skewed keys (C5: Zipf(1) over 1e6 values x 16) put a large share of the rows on a few thousand groups:
// the 4096 ids of ONE aligned key window -- the densest one of a strided sample -- are aggregated by the scatter kernel itself,
// in per-workgroup LDS accumulators (sum / row count / valid count, the layout gb_part_aggregate keeps per partition), and
// merged into the global cells with one atomic per touched id and workgroup at the end of the kernel.  Those rows are never
// counted (gbp_count skips them), never staged, never written as records and never read by the aggregation: C5's window
// (k0 < 256) holds 42 % of the rows.  The window is a performance guess only -- any window gives the same result -- and the
// plain kernels run when the sample finds no window worth the LDS (uniform keys).
// LDS: 64 KB of accumulators leave room for GBP_HOT_CAP staged records per 8192-row tile; a tile with more cold rows than that
// (the sample mispredicted) is regrouped and flushed in rounds.
// ---------------------------------------------------------------------------
constexpr int GBP_HOT_BITS = 12;
constexpr uint32_t GBP_HOT_IDS = 1u << GBP_HOT_BITS;
constexpr uint32_t GBP_NO_HOT = 0xffffffffu;
struct GbHot {
  uint32_t window;                 // key >> GBP_HOT_BITS of the hot ids (the key WITHOUT its validity bit)
  unsigned long long *gacc;        // the global cells the partials are merged into (indexed by key)
  unsigned int *grows, *gvalid;
  int dbg;                         // LAB build: ablation bits (knob GDF_GBP_HOT_DBG), 0 in production
  int plain_rank;                  // the sample found no partition with a large share of the rows that are ranked: gbp_rank_plain
};

// Strided sample of the packed keys: counts[w] += sampled rows whose key lies in window w (GBP_HOT_IDS ids each, nwin <= 4096
// windows), counts[nwin] += sampled rows with a valid, in-range key.  GBP_SAMPLE_WINDOWS windows of 1024 rows spread evenly over
// the table (2^21 rows: 25 MB of C5's 20 GB), sixteen windows per workgroup.  The host reads the counts back and decides two
// things from them: the hot window (the densest one, when it holds enough of the rows) and, for the SPECULATIVE record layout
// (GbSpec below), how much room every partition gets.
constexpr int GBP_SAMPLE_WINDOWS = 2048;
constexpr int GBP_SAMPLE_PER_WG = 16;
__global__ __launch_bounds__(1024) void gbp_sample_hist(KeyTable t, GbKeyPlan plan, uint32_t nwin, unsigned int *__restrict__ counts) {
  __shared__ uint32_t cnt[4096];
  for (uint32_t q = threadIdx.x; q < 4096; q += 1024) cnt[q] = 0;
  block_sync();
  const double stride = (double)t.nrows / (double)GBP_SAMPLE_WINDOWS;
  uint32_t good = 0;
  for (int w0 = 0; w0 < GBP_SAMPLE_PER_WG; w0 += 8) {
    uint32_t src[8], key[8], okmask, outmask;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = (int64_t)((double)(blockIdx.x * GBP_SAMPLE_PER_WG + w0 + k) * stride) + threadIdx.x;
      src[k] = (uint32_t)(i < t.nrows ? i : t.nrows - 1);
    }
    gbp_pack32<8, -1, -1>(t, plan, src, key, okmask, outmask);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t w = key[k] >> GBP_HOT_BITS;
      if (((okmask >> k) & 1u) && !((outmask >> k) & 1u) && w < nwin) { atomicAdd(&cnt[w], 1u); ++good; }
    }
  }
  block_sync();
  for (uint32_t q = threadIdx.x; q < nwin; q += 1024)
    if (cnt[q]) atomicAdd(&counts[q], cnt[q]);
  good = wave_reduce_add(good);
  if (lane_id() == 0 && good) atomicAdd(&counts[nwin], good);
}

// flags[0] += rows dropped for a null key, flags[1] = 1 when a key lies outside the plan's ranges
template <int K0 = -1, int K1 = -1>
__global__ __launch_bounds__(GBP_THREADS) void gbp_count(KeyTable t, GbKeyPlan plan, int low, int vbit, uint32_t nparts, int64_t chunk,
                                                         int nchunks, uint32_t *__restrict__ hist, unsigned int *__restrict__ flags,
                                                         uint32_t qstride, uint32_t cstride, uint32_t hot_window) {
  __shared__ uint32_t cnt[GBP_MAX_PARTS];
  constexpr int B = (K0 >= 0 && K1 == -2) ? 12 : 8;       // one key column to read: half as many rows again in flight per thread (77 VGPRs at 8; 16 spill)
  unsigned int dropped = 0, outside = 0;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t q = threadIdx.x; q < nparts; q += GBP_THREADS) cnt[q] = 0;
    block_sync();
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
    for (int64_t base = begin; base < end; base += (int64_t)GBP_THREADS * B) {
      uint32_t src[B];                 // row numbers fit 31 bits (the entry point refuses INT_MAX rows)
      uint32_t key[B], okmask, outmask;          // (the fused path's keys fit 32 bits with their validity bit)
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const int64_t i = base + (int64_t)k * GBP_THREADS + threadIdx.x;
        src[k] = (uint32_t)(i < end ? i : end - 1);
      }
      if constexpr (K0 >= 0) {
        gbp_pack32<B, K0, K1>(t, plan, src, key, okmask, outmask);
      } else {        // any number of columns of any kind: the column loop (two columns' words in flight at once cost it three spills)
        uint64_t key64[B];
        bool ok[B], inside[B];
        gbp_pack<B>(t, plan, src, key64, ok, inside);
        okmask = outmask = 0;
#pragma unroll
        for (int k = 0; k < B; ++k) { key[k] = (uint32_t)key64[k]; okmask |= (uint32_t)ok[k] << k; outmask |= (uint32_t)!inside[k] << k; }
      }
#pragma unroll
      for (int k = 0; k < B; ++k) {
        const bool live = base + (int64_t)k * GBP_THREADS + threadIdx.x < end;
        const bool okk = (okmask >> k) & 1u;
        if (live && !okk) ++dropped;
        if (live && okk && ((outmask >> k) & 1u)) outside = 1;
        // (rows of the hot window are aggregated by the scatter kernel, GbHot: they get no place among the records)
        bool mine = live && okk && (key[k] >> GBP_HOT_BITS) != hot_window;
        const uint32_t part = (key[k] << vbit) >> low;
        // the partition id of the wave's first live lane by ballot, the rest one LDS atomic each.  (One round: C5's hot partition holds
        // 45 % of the rows, the next one 5 %; a second ballot round cost more than the same-address atomics it saved -- 1.71 -> 1.50 ms,
        // none at all 1.68, three 2.1, profiles/r3_zq_gbp_count_rounds.txt)
#pragma unroll
        for (int round = 0; round < 1; ++round) {
          const unsigned long long todo = __ballot(mine);
          if (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t lp = __shfl(part, leader);
            const unsigned long long same = __ballot(mine && part == lp);
            if (lane_id() == leader) atomicAdd(&cnt[lp], (uint32_t)__popcll(same));
            if (part == lp) mine = false;
          }
        }
        if (mine) atomicAdd(&cnt[part], 1u);
      }
    }
    block_sync();
    for (uint32_t q = threadIdx.x; q < nparts; q += GBP_THREADS) hist[(size_t)q * qstride + (size_t)c * cstride] = cnt[q];
    block_sync();
  }
  dropped = wave_reduce_add(dropped);
  if (lane_id() == 0 && dropped) atomicAdd(&flags[0], dropped);
  if (outside) flags[1] = 1u;
}

// Ranks of N rows per lane within their (tile, partition) groups, hist[partition] += rows -- the fused scatter's replacement for
// N x wave_aggregated_inc.  The 11-bit match-any behind that costs ~90 VALU instructions per row and lane (a quarter of the
// kernel's time on C5) to protect against ONE hot counter; here the partitions of the first two still-unranked lanes of the wave
// are settled by a ballot each (gbp_count's trick: a hot partition is almost always one of them), everybody else takes a plain
// returning LDS atomic.  Phase 1 is ballots only, phase 2 issues every atomic of the N rows before anything waits.
template <int N>
__device__ __forceinline__ void gbp_rank(uint32_t *hist, const uint32_t (&part)[N], uint32_t livemask, uint32_t (&rank)[N]) {
  const int lane = lane_id();
  uint32_t inrank[N];            // lanes settled by a ballot: position within their group
  uint32_t caught = 0;           // two bits per row: 0 = plain atomic, 1 / 2 = settled with the first / second leader's group
  int lead[N][2];                // wave-uniform: the leaders' lanes (64: nobody)
#pragma unroll
  for (int k = 0; k < N; ++k) {
    bool mine = (livemask >> k) & 1u;
    inrank[k] = 0;
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      const unsigned long long todo = __ballot(mine);
      const int leader = __builtin_amdgcn_readfirstlane(todo ? __ffsll((long long)todo) - 1 : 64);
      const uint32_t lp = __builtin_amdgcn_readlane(part[k], leader & 63);
      const bool grp = mine && leader < 64 && part[k] == lp;
      const unsigned long long same = __ballot(grp);
      lead[k][round] = leader;
      if (grp) { inrank[k] = (uint32_t)mask_rank(same) | ((uint32_t)__popcll(same) << 16); caught |= (uint32_t)(round + 1) << (2 * k); mine = false; }
    }
  }
  uint32_t res[N];               // leaders: their group's base; plain lanes: their own rank
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint32_t c = (caught >> (2 * k)) & 3u;
    res[k] = 0;
    // ONE atomic instruction per row: leaders add their group's size (a leader's inrank is 0 | size << 16), plain lanes add 1
    const bool leads = lane == lead[k][0] || lane == lead[k][1];
    if (leads || (c == 0 && ((livemask >> k) & 1u))) res[k] = atomicAdd(&hist[part[k]], leads ? inrank[k] >> 16 : 1u);
  }
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const uint32_t c = (caught >> (2 * k)) & 3u;
    const uint32_t b0 = __builtin_amdgcn_readlane(res[k], lead[k][0] & 63), b1 = __builtin_amdgcn_readlane(res[k], lead[k][1] & 63);
    rank[k] = c == 0 ? res[k] : (c == 1 ? b0 : b1) + (inrank[k] & 0xffffu);
  }
}

// The same without the ballots: one returning LDS atomic per row, a row that is not ranked adds to its lane's trash counter -- no
// branch, eight atomics in flight.  gbp_rank's two leader rounds cost ~150 SIMD cycles per row and wave to protect against a hot
// counter (same-address LDS atomics are served one lane after the other); once the hot key window is aggregated where it is read,
// the busiest partition of C5 holds 8 % of the ranked rows -- three lanes of a wave -- and the kernel, which issues VALU work for two
// thirds of its time (tools/kernel_blocks.py), is better off with the plain atomic.  The host decides from the sample (GbHot::plain_rank).
template <int N>
__device__ __forceinline__ void gbp_rank_plain(uint32_t *hist, const uint32_t (&part)[N], uint32_t livemask, uint32_t (&rank)[N]) {
  const uint32_t trash = (uint32_t)GBP_MAX_PARTS + (uint32_t)lane_id();
  uint32_t at[N];
#pragma unroll
  for (int k = 0; k < N; ++k) at[k] = ((livemask >> k) & 1u) ? part[k] & (uint32_t)(GBP_MAX_PARTS - 1) : trash;
#pragma unroll
  for (int k = 0; k < N; ++k) rank[k] = atomicAdd(&hist[at[k]], 1u);
}

// The fused scatter for ANY key / value shape: the column loop with its type switches (gbp_pack), match-any ranks, six barriers
// per tile.  Shapes with a static signature take gbp_scatter_static below.
template <bool VBIT>
__global__ __launch_bounds__(GBP_SC_THREADS) void gbp_scatter(KeyTable t, GbKeyPlan plan, GbVal val, int fold_op, int low, int part_bits,
                                                           uint32_t nparts, int64_t chunk, int nchunks, const uint32_t *__restrict__ offs,
                                                           GbRec *__restrict__ rec_out, uint32_t qstride, uint32_t cstride) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gbp_lds[];
  uint64_t *stage = reinterpret_cast<uint64_t *>(gbp_lds);                   // [TILE] accumulator images, regrouped by partition
  uint32_t *stage_k = reinterpret_cast<uint32_t *>(stage + GBP_SC_TILE);        // [TILE] their keys (the partition is key >> low)
  uint32_t *hist = stage_k + GBP_SC_TILE;                                       // [MAX_PARTS + 4]
  uint32_t *start = hist + GBP_MAX_PARTS + GBP_RANK_TRASH, *gbase = start + GBP_MAX_PARTS, *cursor = gbase + GBP_MAX_PARTS;
  uint32_t *wave_tot = cursor + GBP_MAX_PARTS;                               // [THREADS / WAVE]
  constexpr int PER = GBP_MAX_PARTS / GBP_SC_THREADS;                           // partitions per thread in the scan (2)
  constexpr int vbit = VBIT ? 1 : 0;
  for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
    for (uint32_t q = threadIdx.x; q < nparts; q += GBP_SC_THREADS) cursor[q] = offs[(size_t)q * qstride + (size_t)c * cstride];
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
    for (int64_t tile = begin; tile < end; tile += GBP_SC_TILE) {
      for (uint32_t q = threadIdx.x; q < GBP_MAX_PARTS; q += GBP_SC_THREADS) hist[q] = 0;
      block_sync();
      uint32_t src[GBP_ITEMS];
      uint64_t key[GBP_ITEMS];
      bool ok[GBP_ITEMS], inside[GBP_ITEMS];
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) {
        const int64_t i = tile + (int64_t)k * GBP_SC_THREADS + threadIdx.x;
        src[k] = (uint32_t)(i < end ? i : end - 1);
      }
      // the value column is requested NOW, with the keys: one round of HBM latency per tile instead of two (as a second load
      // phase behind the key flush this kernel ran at 2.9 TB/s on its 32 B per row)
      uint64_t img[GBP_ITEMS];
      switch (fold_op == OP_COUNT ? -1 : val.kind) {
        case -1:
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) img[k] = 1;
          break;
        case K_I8:
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) img[k] = (uint64_t)(int64_t)((const int8_t *)val.data)[src[k]];
          break;
        case K_I16:
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) img[k] = (uint64_t)(int64_t)((const int16_t *)val.data)[src[k]];
          break;
        case K_I32:
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) img[k] = (uint64_t)(int64_t)((const int32_t *)val.data)[src[k]];
          break;
        case K_F32:
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) img[k] = (uint64_t)__double_as_longlong((double)((const float *)val.data)[src[k]]);
          break;
        default:      // K_I64 and K_F64: the raw 64-bit word is the SUM image of both
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) img[k] = ((const uint64_t *)val.data)[src[k]];
      }
      if (fold_op == OP_MIN || fold_op == OP_MAX) {
        const bool flt = is_flt(val.kind);
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k)
          img[k] = flt ? ord_f64(__longlong_as_double((long long)img[k])) : ord_i64((int64_t)img[k]);
      }
      gbp_pack<GBP_ITEMS>(t, plan, src, key, ok, inside);
      // validity of the VALUE: rides as the key's lowest bit when the aggregation counts valid values (VBIT), and a null
      // value always contributes the identity (COUNT of a masked column has no such bit but still must not count nulls)
      uint32_t vmask = 0;               // bit k: the value of item k is valid
      if (val.valid) {
        uint8_t vb[GBP_ITEMS];
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) vb[k] = val.valid[src[k] >> 3];
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) vmask |= (uint32_t)((vb[k] >> (src[k] & 7)) & 1) << k;
      } else {
        vmask = 0xffffffffu;
      }
      uint32_t pr[GBP_ITEMS];          // partition << 16 | rank within (tile, partition); 0xffffffff: the row does not travel
      uint32_t k32[GBP_ITEMS];
      {
        uint32_t part[GBP_ITEMS], rk[GBP_ITEMS], livemask = 0;
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) {
          const bool live = tile + (int64_t)k * GBP_SC_THREADS + threadIdx.x < end && ok[k];
          k32[k] = (uint32_t)((key[k] << vbit) | (uint64_t)(VBIT && ((vmask >> k) & 1u)));
          part[k] = k32[k] >> low;
          livemask |= (uint32_t)live << k;
        }
        gbp_rank<GBP_ITEMS>(hist, part, livemask, rk);
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) pr[k] = ((livemask >> k) & 1u) ? (part[k] << 16) | rk[k] : 0xffffffffu;
      }
      block_sync();
      {   // exclusive scan of hist[0..MAX_PARTS) by the 1024 threads, PER consecutive partitions each
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { v[q] = hist[threadIdx.x * PER + q]; sum += v[q]; }
        const uint32_t incl = wave_scan_incl(sum);
        if (lane_id() == WAVE - 1) wave_tot[threadIdx.x / WAVE] = incl;
        block_sync();
        uint32_t run = incl - sum + waves_before_sum<GBP_SC_THREADS / WAVE>(wave_tot, threadIdx.x);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const uint32_t b = threadIdx.x * PER + q;
          start[b] = run;
          gbase[b] = cursor[b] - run;          // (only partitions < nparts are ever looked up)
          cursor[b] += v[q];
          run += v[q];
        }
      }
      block_sync();
      uint32_t total = 0;
      for (int w = 0; w < GBP_SC_THREADS / WAVE; ++w) total += wave_tot[w];
      // key and image are regrouped together and leave as ONE 12-byte record per row: written as a 4-byte and an 8-byte array
      // (two LDS rounds, two flushes) every short (tile, partition) run cost two store requests -- and the request rate, not
      // the bytes, bounds this kernel on C5, where half of the rows sit in runs of a few rows
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) {
        if (pr[k] != 0xffffffffu) {
          const uint32_t pos = start[pr[k] >> 16] + (pr[k] & 0xffffu);
          stage_k[pos] = k32[k];
          stage[pos] = ((vmask >> k) & 1u) ? img[k] : acc_identity(fold_op);
        }
      }
      block_sync();
      for (uint32_t j = threadIdx.x; j < total; j += GBP_SC_THREADS) {
        const uint32_t kk = stage_k[j];
        const uint64_t vv = stage[j];
        rec_out[gbase[kk >> low] + j] = GbRec{(uint32_t)vv, (uint32_t)(vv >> 32), kk};
      }
      block_sync();
    }
  }
}

// threadIdx.x through an opaque move: values derived from the result cannot be hoisted out of the enclosing loop (hipcc hoists a
// dozen per-thread row offsets / LDS addresses out of the tile loop and then spills what has to stay live across it)
__device__ __forceinline__ uint32_t gbp_opaque_tid() {
  uint32_t tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  return tid;
}
// The fused scatter for one or two 4- / 8-byte integer key columns (K0, K1 as in gbp_pack32) and an 8-byte value column
// (K_I64 / K_F64, not COUNT); VMASK: the value carries a validity mask.  What it does differently, and what each step bought on
// C5 (1e9 rows, gbp_scatter 10.2 ms at the start; tools/gpu/r2bb.sh .. r2bg.sh):
//   * no type switch between a tile's HBM requests (see gbp_pack32): value words, mask bytes and both key columns leave together;
//   * four barriers per tile instead of six: the tile's counters are cleared by the scan that reads them (every thread its own
//     two), and nothing closes the flush -- a wave that has stored its share goes on with the next tile's rows, touching only
//     hist[], which the flush does not read, and stops at that tile's first barrier until every wave has left the flush
//     (these two together: 10.2 -> 9.9 ms);
//   * gbp_rank instead of an 11-bit match-any per row (9.7 -> 8.05 ms: the kernel was a quarter VALU);
//   * the LDS reads of the regroup and of the flush are batched (one round trip per phase instead of one or two per row);
// Tried and dropped (tools/gpu/r2be.sh, r2bg.sh): touching every 128-byte line of the next tile with a 4-byte load during the flush
// (8.1 -> 12.9 ms: 64 lines per wave instruction are 64 requests), and jk_scatter1's pipeline -- the next tile's key words requested
// before the flush and held in registers across it (9.34 against 9.40 ms, for 6 spilled registers).
//   * HOT (round 4, GbHot above): the rows of the hot key window are folded into LDS accumulators instead of being staged; the
//     stage then holds GBP_HOT_CAP records and a tile with more cold rows is regrouped and flushed in rounds.
constexpr int GBP_HOT_CAP = 5 * GBP_SC_THREADS;         // 5120 records: 60 KB next to 64 KB of accumulators and 32 KB of counters
//   * SPEC (round 4, GbSpec): no count pass in front -- every workgroup appends to its own segment of every partition; the scan
//     step checks the segment's room, a workgroup that runs out raises flags[2] and all of them stop at their next tile.
constexpr int GB_ROLE_RECORDS = 16, GB_PLACE_DRAWS = 6;      // the placed record buffer of the fused partition pass (gb_sorted_partitioned)
constexpr uint32_t GBP_SPEC_SKIP = 0xA0000000u;          // a destination at or beyond 2^31: the flush does not store there
template <bool VBIT, int K0, int K1, bool VMASK, bool HOT = false, bool SPEC = false>
__global__ __launch_bounds__(GBP_SC_THREADS) void gbp_scatter_static(KeyTable t, GbKeyPlan plan, GbVal val, int fold_op, int low,
                                                                  uint32_t nparts, int64_t chunk, int nchunks, const uint32_t *__restrict__ offs,
                                                                  GbRec *__restrict__ rec_out, unsigned int *__restrict__ flags,
                                                                  uint32_t qstride, uint32_t cstride, GbHot hot, GbSpec spec) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gbp_lds[];
  constexpr int CAP = HOT ? GBP_HOT_CAP : GBP_SC_TILE;                       // staged records per round
  constexpr int FLUSH_ITEMS = CAP / GBP_SC_THREADS;
  uint64_t *stage = reinterpret_cast<uint64_t *>(gbp_lds);
  unsigned long long *hacc = reinterpret_cast<unsigned long long *>(stage + CAP);      // HOT: [GBP_HOT_IDS] accumulators
  uint32_t *stage_k = reinterpret_cast<uint32_t *>(hacc + (HOT ? GBP_HOT_IDS : 0));
  uint32_t *hist = stage_k + CAP;
  uint32_t *start = hist + GBP_MAX_PARTS + GBP_RANK_TRASH, *gbase = start + GBP_MAX_PARTS, *cursor = gbase + GBP_MAX_PARTS;
  uint32_t *wave_tot = cursor + GBP_MAX_PARTS;
  unsigned int *hrows = wave_tot + GBP_SC_THREADS / WAVE, *hvalid = hrows + GBP_HOT_IDS;   // HOT: rows / valid values per hot id
  constexpr int PER = GBP_MAX_PARTS / GBP_SC_THREADS;
  constexpr int vbit = VBIT ? 1 : 0;
  using W0 = typename std::conditional<K0 == K_I64, long long, int32_t>::type;
  using W1 = typename std::conditional<K1 == K_I64, long long, int32_t>::type;
  // the raw words of one tile: what is in flight between the request and the first use
  W0 r0[GBP_ITEMS];
  W1 r1[GBP_ITEMS];
  uint64_t img[GBP_ITEMS];
  uint8_t vb[GBP_ITEMS];
  auto row_of = [&](int64_t tile, int64_t end, int k, uint32_t tid) -> uint32_t {        // clamped: requests are unconditional
    const int64_t i = tile + (int64_t)k * GBP_SC_THREADS + tid;
    return (uint32_t)(i < end ? i : end - 1);
  };
  auto request = [&](int64_t tile, int64_t end) {
    const uint32_t tid = gbp_opaque_tid();
#ifdef GDF_AMD_LAB
    if (HOT && (LAB_BITS(hot.dbg) & 16)) {            // LAB bit 16: read-once input as non-temporal loads (do the streams push the write fronts out of L2?)
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) if (VMASK) vb[k] = __builtin_nontemporal_load(val.valid + (row_of(tile, end, k, tid) >> 3));
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) img[k] = __builtin_nontemporal_load((const uint64_t *)val.data + row_of(tile, end, k, tid));
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) r0[k] = __builtin_nontemporal_load((const W0 *)t.col[0].data + row_of(tile, end, k, tid));
      if (K1 >= 0) {
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) r1[k] = __builtin_nontemporal_load((const W1 *)t.col[1].data + row_of(tile, end, k, tid));
      }
      return;
    }
#endif
#pragma unroll
    for (int k = 0; k < GBP_ITEMS; ++k) if (VMASK) vb[k] = val.valid[row_of(tile, end, k, tid) >> 3];
#pragma unroll
    for (int k = 0; k < GBP_ITEMS; ++k) img[k] = ((const uint64_t *)val.data)[row_of(tile, end, k, tid)];
#pragma unroll
    for (int k = 0; k < GBP_ITEMS; ++k) r0[k] = ((const W0 *)t.col[0].data)[row_of(tile, end, k, tid)];
    if (K1 >= 0) {
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) r1[k] = ((const W1 *)t.col[1].data)[row_of(tile, end, k, tid)];
    }
  };
  for (uint32_t q = threadIdx.x; q < GBP_MAX_PARTS; q += GBP_SC_THREADS) hist[q] = 0;
  if constexpr (HOT) {
    for (uint32_t i = threadIdx.x; i < GBP_HOT_IDS; i += GBP_SC_THREADS) { hacc[i] = acc_identity(fold_op); hrows[i] = 0; hvalid[i] = 0; }
  }
  // SPEC: the scan thread of partitions b = 2 tid, 2 tid + 1 keeps their segments' first record and room in registers; cursor[]
  // counts what the segment holds so far (never reset: one segment per partition for the whole kernel)
  uint32_t seg_base[PER], seg_cap[PER];
  __shared__ uint32_t spec_abort;
  uint32_t xcc = 0;
  if constexpr (SPEC) {
    // HW_REG_XCC_ID (hardware register 20), bits 3..0: the XCD this workgroup runs on
    if (spec.xcd) xcc = (uint32_t)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 7u;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const uint32_t b = threadIdx.x * PER + q;
      const bool in = b < nparts;
      seg_cap[q] = in ? spec.cap[b] : 0u;
      seg_base[q] = in ? spec.qprefix[b] * spec.G + (spec.xcd ? xcc : blockIdx.x) * seg_cap[q] : 0u;
      cursor[b] = 0;
    }
    if (threadIdx.x == 0) spec_abort = 0;
  }
  block_sync();
  // chunk -> workgroup: XCD x (workgroups x, x + 8, ...: MI355X_MICROARCH.md "Workgroup dispatch") takes the x-th EIGHTH of the chunks, its
  // workgroups round-robin inside it.  The regions of chunks c and c + 1 are neighbours inside every partition and share their
  // boundary line; with c -> workgroup c % grid they sat behind two different, non-coherent L2s (the trick of jk_scatter2's tiles)
  const bool xcd_map = (gridDim.x & 7u) == 0 && !(LAB_BITS(hot.dbg) & 32);
  const int per_xcd = (nchunks + 7) / 8, first = xcd_map ? (int)(blockIdx.x >> 3) : (int)blockIdx.x, step = xcd_map ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  for (int ci = first; xcd_map ? ci < per_xcd : ci < nchunks; ci += step) {
    const int c = xcd_map ? (int)(blockIdx.x & 7u) * per_xcd + ci : ci;
    if (c >= nchunks) break;
    if constexpr (!SPEC) {
      for (uint32_t q = threadIdx.x; q < nparts; q += GBP_SC_THREADS) cursor[q] = offs[(size_t)q * qstride + (size_t)c * cstride];
    }
    const int64_t begin = (int64_t)c * chunk;
    const int64_t end = begin + chunk < t.nrows ? begin + chunk : t.nrows;
    for (int64_t tile = begin; tile < end; tile += GBP_SC_TILE) {
      request(tile, end);
      // SPEC: somebody ran out of room -- the host repeats the call on the exact layout, the rest of this pass is wasted work
      // (thread 0 looks at the flag once per tile; everybody acts on it behind the tile's first barrier)
      if constexpr (SPEC) {
        if (threadIdx.x == 0) spec_abort = __hip_atomic_load(&flags[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // ---- consume: packed 32-bit key, flags as bit masks ----
      const uint32_t tid = gbp_opaque_tid();
      uint32_t k32[GBP_ITEMS], okmask = (1u << GBP_ITEMS) - 1u, outside = 0, inrange = 0, vmask = VMASK ? 0u : 0xffffffffu;
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) {
        const uint64_t u0 = (uint64_t)((long long)r0[k] - plan.bias[0]);
        outside |= (uint32_t)(plan.bits[0] < 64 && (u0 >> plan.bits[0]) != 0) << k;
        k32[k] = (uint32_t)(u0 & low_mask(plan.bits[0])) << plan.shift[0];
        if (K1 >= 0) {
          const uint64_t u1 = (uint64_t)((long long)r1[k] - plan.bias[1]);
          outside |= (uint32_t)(plan.bits[1] < 64 && (u1 >> plan.bits[1]) != 0) << k;
          k32[k] |= (uint32_t)(u1 & low_mask(plan.bits[1])) << plan.shift[1];
        }
        inrange |= (uint32_t)(tile + (int64_t)k * GBP_SC_THREADS + tid < end) << k;
        if (VMASK) vmask |= (uint32_t)((vb[k] >> (row_of(tile, end, k, tid) & 7)) & 1) << k;
      }
      // null key elements (rare: a late request of mask bytes)
#pragma unroll
      for (int c2 = 0; c2 < (K1 >= 0 ? 2 : 1); ++c2) {
        if (t.col[c2].valid) {
          uint8_t m[GBP_ITEMS];
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) m[k] = t.col[c2].valid[row_of(tile, end, k, tid) >> 3];
#pragma unroll
          for (int k = 0; k < GBP_ITEMS; ++k) okmask &= ~((uint32_t)(((m[k] >> (row_of(tile, end, k, tid) & 7)) & 1) ^ 1) << k);
        }
      }
      // gbp_count leaves out a last key column that cannot change the partition id (see the launch): every column's values are
      // checked against the plan's ranges HERE as well, flags[1] as in gbp_count (the caller looks at it after this kernel)
      if (outside & okmask & inrange) flags[1] = 1u;
      uint64_t acc[GBP_ITEMS];         // the accumulator image that travels (the identity for a null value)
      {
        const bool minmax = fold_op == OP_MIN || fold_op == OP_MAX, flt = is_flt(val.kind);
        const uint64_t ident = acc_identity(fold_op);
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) {
          uint64_t x = img[k];
          if (minmax) x = flt ? ord_f64(__longlong_as_double((long long)x)) : ord_i64((int64_t)x);
          acc[k] = ((vmask >> k) & 1u) ? x : ident;
        }
      }
      uint32_t part[GBP_ITEMS], rk[GBP_ITEMS];
      uint32_t livemask = okmask & inrange;
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) {
        k32[k] = (k32[k] << vbit) | (uint32_t)(VBIT && ((vmask >> k) & 1u));
        part[k] = k32[k] >> low;
      }
      if constexpr (HOT) {
        // rows of the hot window: folded into this workgroup's accumulators (exactly what gb_part_aggregate does with a record)
        // and taken out of the tile
        const bool flt = is_flt(val.kind);
        uint32_t hotmask = 0;
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) hotmask |= (uint32_t)((k32[k] >> (GBP_HOT_BITS + vbit)) == hot.window) << k;
        hotmask &= livemask;
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) {
          if (((hotmask >> k) & 1u) && !(LAB_BITS(hot.dbg) & 1)) {          // (LAB bit 1: hot rows vanish without their atomics)
            const uint32_t id = (k32[k] >> vbit) & (GBP_HOT_IDS - 1u);
            atomicAdd(&hrows[id], 1u);
            if (!VBIT || (k32[k] & 1u)) {
              acc_fold(fold_op, flt, &hacc[id], acc[k]);
              if (VBIT) atomicAdd(&hvalid[id], 1u);
            }
          }
        }
        livemask &= ~hotmask;
        if (LAB_BITS(hot.dbg) & 4) livemask = 0;            // (LAB bit 4: no cold row travels)
      }
      if (hot.plain_rank) gbp_rank_plain<GBP_ITEMS>(hist, part, livemask, rk);      // (uniform)
      else gbp_rank<GBP_ITEMS>(hist, part, livemask, rk);
      block_sync();
      if constexpr (SPEC) {
        if (spec_abort) return;                     // workgroup-uniform (written before the barrier above); no hot merge: the result is discarded
      }
      uint32_t claimed[PER];            // XCD-shared segments: where this thread's partitions' runs start in their segments
#pragma unroll
      for (int q = 0; q < PER; ++q) claimed[q] = 0;
      {   // exclusive scan of hist[0..MAX_PARTS) by the 1024 threads, PER consecutive partitions each; clears hist for the next tile
        uint32_t v[PER], sum = 0;
#pragma unroll
        for (int q = 0; q < PER; ++q) { v[q] = hist[threadIdx.x * PER + q]; sum += v[q]; hist[threadIdx.x * PER + q] = 0; }
        // XCD-shared segments: the runs' places are claimed NOW (returning atomics in the XCD's L2) and looked at behind the regroup
        if constexpr (SPEC) {
          if (spec.xcd) {
#pragma unroll
            for (int q = 0; q < PER; ++q) {
              const uint32_t b = threadIdx.x * PER + q;
              // AGENT scope, relaxed (round 5): the counter is shared by DIFFERENT workgroups, which only an agent-scope
              // atomic is defined for in the HIP memory model.  It costs nothing: on gfx950 a relaxed returning atomic is the SAME
              // instruction at workgroup and at agent scope (`global_atomic_add ... sc0`; only system scope adds sc1) -- what cost 3 ms
              // in round 3 was the acquire / release fence pair around it, not the scope.
              claimed[q] = (v[q] && b < nparts) ? __hip_atomic_fetch_add(&spec.fill[(size_t)b * 8u + xcc], v[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            }
          }
        }
        const uint32_t incl = wave_scan_incl(sum);
        if (lane_id() == WAVE - 1) wave_tot[threadIdx.x / WAVE] = incl;
        block_sync();
        uint32_t run = incl - sum + waves_before_sum<GBP_SC_THREADS / WAVE>(wave_tot, threadIdx.x);
#pragma unroll
        for (int q = 0; q < PER; ++q) {
          const uint32_t b = threadIdx.x * PER + q;
          start[b] = run;
          if constexpr (SPEC) {
            if (spec.xcd) {
              // XCD-shared segments: the claim's answer -- a returning L2 atomic, ~1 - 2 us -- is only needed by the flush.  It is
              // looked at behind the regroup (settle_claims below); the run's length waits in cursor[], which this layout does not use
              cursor[b] = v[q];
            } else {
              const uint32_t have = cursor[b];
              const bool fits = have + v[q] <= seg_cap[q];
              if (!fits) flags[2] = 1u;              // (the records of this run go nowhere: GBP_SPEC_SKIP)
              gbase[b] = fits ? seg_base[q] + have - run : GBP_SPEC_SKIP - run;
              cursor[b] = fits ? have + v[q] : have;
            }
          } else {
            gbase[b] = cursor[b] - run;
            cursor[b] += v[q];
          }
          run += v[q];
        }
      }
      block_sync();
      uint32_t total = 0;
      for (int w = 0; w < GBP_SC_THREADS / WAVE; ++w) total += wave_tot[w];
      uint32_t st[GBP_ITEMS];          // all reads of start[] first (one LDS round trip, not one per row), then the writes
#pragma unroll
      for (int k = 0; k < GBP_ITEMS; ++k) st[k] = start[part[k] & (GBP_MAX_PARTS - 1)] + rk[k];
      // one round: the records at tile positions [round0, round0 + CAP) are regrouped in the stage and flushed
      auto regroup = [&](uint32_t round0) {
#pragma unroll
        for (int k = 0; k < GBP_ITEMS; ++k) {
          const uint32_t pos = st[k] - round0;               // (unsigned: positions below round0 wrap beyond CAP)
          if (((livemask >> k) & 1u) && (!HOT || pos < (uint32_t)CAP)) {
            stage_k[pos] = k32[k];
            stage[pos] = acc[k];
          }
        }
      };
#ifdef GDF_AMD_LAB
      const uint32_t lab_span = (uint32_t)(t.nrows / gridDim.x) - (uint32_t)GBP_SC_TILE;
      const uint32_t lab_stream = blockIdx.x * (uint32_t)(t.nrows / gridDim.x) + (uint32_t)(((tile - begin) / GBP_SC_TILE) * CAP) % (lab_span ? lab_span : 1u);
#endif
      auto flush = [&](uint32_t round0) {
        // every LDS read first (the record, then its partition's base), then the stores; slots beyond the round re-read slot 0
        uint32_t kk[FLUSH_ITEMS], gb[FLUSH_ITEMS];
        uint64_t vv[FLUSH_ITEMS];
        const uint32_t ftid = gbp_opaque_tid();
        const uint32_t left = total - round0, cnt = (HOT && left > (uint32_t)CAP) ? (uint32_t)CAP : left;
#pragma unroll
        for (int k = 0; k < FLUSH_ITEMS; ++k) {
          const uint32_t j = ftid + k * GBP_SC_THREADS;
          const uint32_t jc = j < cnt ? j : 0u;
          kk[k] = stage_k[jc];
          vv[k] = stage[jc];
        }
#pragma unroll
        for (int k = 0; k < FLUSH_ITEMS; ++k) gb[k] = gbase[(kk[k] >> low) & (GBP_MAX_PARTS - 1)];
#pragma unroll
        for (int k = 0; k < FLUSH_ITEMS; ++k) {
          const uint32_t j = ftid + k * GBP_SC_THREADS;
          uint32_t dst = gb[k] + round0 + j;
#ifdef GDF_AMD_LAB
          // LAB bit 8: the same records as one contiguous stream per workgroup (what would the stores cost without the short runs?)
          if (LAB_BITS(hot.dbg) & 8) { dst = lab_stream + j; }
#endif
          if (j < cnt && (!SPEC || (int32_t)dst >= 0) && !(LAB_BITS(hot.dbg) & 2)) rec_out[dst] = GbRec{(uint32_t)vv[k], (uint32_t)(vv[k] >> 32), kk[k]};   // (LAB bit 2: no stores)
        }
      };
      // HOT: the stage holds CAP < TILE records; a tile with more cold rows than that (the sample mispredicted the window) sends its
      // LATER positions first, round by round, then the first CAP as every tile does (`total` is workgroup-uniform).
      // (Requesting the NEXT tile's words in front of the last flush and holding them in registers across it -- the five store
      // instructions leave room, no spill -- changed nothing: 8.34 - 8.78 against 8.39 - 8.49 ms on C5.  The stores are not a
      // latency the loads could hide behind: the same records written as one contiguous stream per workgroup cost 6.7 ms, no
      // stores at all 6.1, the short (tile, partition) runs 8.5 -- profiles/r4_c_c5_scatter_ablation.txt.  Requesting the next
      // tile right after this one's words are consumed -- in flight under ranking, scan, regroup and flush -- needs the raw and
      // the processed words of a tile at once: 128 VGPRs + 53 spilled, 12.0 against 8.46 ms.)
      // XCD-shared segments: the claims' answers become the partitions' bases (the scan left start[] and, in cursor[], the runs' lengths)
      auto settle_claims = [&]() {
        if constexpr (SPEC) {
          if (spec.xcd) {
#pragma unroll
            for (int q = 0; q < PER; ++q) {
              const uint32_t b = threadIdx.x * PER + q;
              const uint32_t have = claimed[q], len = cursor[b], at = start[b];
              const bool fits = have + len <= seg_cap[q];
              if (!fits) flags[2] = 1u;              // (the records of this run go nowhere: GBP_SPEC_SKIP)
              gbase[b] = fits ? seg_base[q] + have - at : GBP_SPEC_SKIP - at;
            }
          }
        }
      };
      const bool rounds = HOT && total > (uint32_t)CAP;       // (workgroup-uniform, rare: the sample mispredicted the hot window)
      if (rounds) settle_claims();
      if constexpr (HOT) {
        for (uint32_t round0 = (total - 1u) / (uint32_t)CAP * (uint32_t)CAP; total && round0 > 0; round0 -= (uint32_t)CAP) {
          regroup(round0);
          block_sync();
          flush(round0);
          block_sync();                  // the next round overwrites the stage
        }
      }
      regroup(0);
      if (!rounds) settle_claims();
      block_sync();
      flush(0);
    }
  }
  if constexpr (SPEC) {
    // what this workgroup's segments hold (every partition, also the untouched ones: the aggregation reads all G counts);
    // XCD-shared segments: the claim counters ARE the fill counts
    if (!spec.xcd) {
      block_sync();
      for (uint32_t q = threadIdx.x; q < nparts; q += GBP_SC_THREADS) spec.fill[(size_t)q * spec.G + blockIdx.x] = cursor[q];
    }
  }
  if constexpr (HOT) {
    // merge this workgroup's partials into the cells that own the keys (cell index = key), as gb_part_aggregate merges a unit
    block_sync();
    const bool flt = is_flt(val.kind);
    for (uint32_t i = threadIdx.x; i < GBP_HOT_IDS; i += GBP_SC_THREADS) {
      const unsigned int r = hrows[i];
      if (!r) continue;
      const size_t cell = ((size_t)hot.window << GBP_HOT_BITS) | i;
      atomicAdd(&hot.grows[cell], r);
      if (VBIT) {
        const unsigned int cv = hvalid[i];
        if (cv) { atomicAdd(&hot.gvalid[cell], cv); acc_fold(fold_op, flt, &hot.gacc[cell], hacc[i]); }
      } else {
        acc_fold(fold_op, flt, &hot.gacc[cell], hacc[i]);
      }
    }
  }
}
static constexpr size_t gbp_scatter_lds(bool hot = false) {
  return hot ? 12 * (size_t)GBP_HOT_CAP + 16 * (size_t)GBP_HOT_IDS + 4 * (size_t)(4 * GBP_MAX_PARTS + GBP_RANK_TRASH + GBP_SC_THREADS / WAVE) + 16
             : 12 * (size_t)GBP_SC_TILE + 4 * (size_t)(4 * GBP_MAX_PARTS + GBP_RANK_TRASH + GBP_SC_THREADS / WAVE) + 16;
}

// number of non-empty cells per block of 1024 cells
__global__ __launch_bounds__(1024) void gb_part_count(const unsigned int *__restrict__ grows, uint32_t *__restrict__ block_count) {
  __shared__ uint32_t wcnt[1024 / WAVE];
  const size_t cell = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const unsigned long long m = __ballot(grows[cell] != 0);
  if (lane_id() == 0) wcnt[threadIdx.x / WAVE] = (uint32_t)__popcll(m);
  block_sync();
  if (threadIdx.x == 0) {
    uint32_t tot = 0;
    for (int w = 0; w < 1024 / WAVE; ++w) tot += wcnt[w];
    block_count[blockIdx.x] = tot;
  }
}
// block_base = exclusive scan of block_count.  Cells are visited in ascending order, so the output is sorted by key.
__global__ __launch_bounds__(1024) void gb_part_extract(KeyTable t, GbKeyPlan plan, GbOut o, int op, const unsigned long long *__restrict__ gacc,
                                                        const unsigned int *__restrict__ grows, const unsigned int *__restrict__ gvalid,
                                                        const uint32_t *__restrict__ block_base, unsigned int *out_groups) {
  __shared__ uint32_t wcnt[1024 / WAVE];
  const size_t cell = (size_t)blockIdx.x * 1024 + threadIdx.x;
  const unsigned int rows = grows[cell];
  const unsigned long long m = __ballot(rows != 0);
  if (lane_id() == 0) wcnt[threadIdx.x / WAVE] = (uint32_t)__popcll(m);
  block_sync();
  uint32_t before = 0, tot = 0;
  for (int w = 0; w < 1024 / WAVE; ++w) { if (w < (int)(threadIdx.x / WAVE)) before += wcnt[w]; tot += wcnt[w]; }
  if (rows) {
    const uint32_t pos = block_base[blockIdx.x] + before + mask_rank(m);
    for (int c = 0; c < t.ncols; ++c) gb_unpack_store(t, plan, (uint64_t)cell, c, o.key_out[c], pos);
    store_result(o, op, pos, gacc[cell], gvalid ? gvalid[cell] : rows);
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *out_groups = block_base[blockIdx.x] + tot;
}

// Everything the four aggregation paths share about one gdf_group_by_* call.
__global__ void gb_strided_u32(const uint32_t *in, uint32_t *out, int count, size_t stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = in[(size_t)i * stride];
}

struct GbJob {
  bool range_violated = false;     // set by a path that ran on sample-guessed key ranges and met a key outside them
  int ncols;
  gdf_column **out_keys;
  gdf_column *out_agg;
  int op;
  bool sort_result;
  KeyTable t;
  ElemKind in_kind, out_kind;
  GbKeyPlan plan;
  GbVal val;
  bool masked;      // some key column or the value column carries a validity mask
  bool counted;     // a per-group count of (valid) values is kept: AVG, or a masked value column
  bool want_ok;     // the caller supplied out_col_agg->valid and groups can come out null
  DevBuf agg_ok;
  std::vector<long long> ranges;   // [2 * ncols] exact min / max per key column when gb_plan_range already took them
  size_t *sort_indices = nullptr;  // a SORT-method call on the direct path: out_col_indices->data (every group's last row), if asked for
  bool sort_method = false;        // a SORT-method call on the direct path (with or without out_col_indices)
};

// Path 1 -- direct index: integer keys with a small value range, no masks.  *done = false: not applicable.
static gdf_error gb_path_direct(GbJob &j, bool *done) {
  *done = false;
  [[maybe_unused]] const int ncols = j.ncols;
  [[maybe_unused]] gdf_column **out_keys = j.out_keys;
  [[maybe_unused]] gdf_column *out_agg = j.out_agg;
  [[maybe_unused]] const int op = j.op;
  [[maybe_unused]] const bool sort_result = j.sort_result;
  [[maybe_unused]] const KeyTable &t = j.t;
  [[maybe_unused]] const int64_t n = j.t.nrows;
  [[maybe_unused]] const ElemKind in_kind = j.in_kind, out_kind = j.out_kind;
  [[maybe_unused]] const GbKeyPlan &plan = j.plan;
  [[maybe_unused]] const GbVal &val = j.val;
  [[maybe_unused]] const bool masked = j.masked, avg = j.counted, want_ok = j.want_ok;
  [[maybe_unused]] DevBuf &agg_ok = j.agg_ok;
  bool all_int = true;
  for (int c = 0; c < ncols; ++c) all_int = all_int && t.col[c].kind != K_F32 && t.col[c].kind != K_F64;
  if (all_int && !masked && n >= 4096 && !lab::path_on("GDF_GB_NO_DIRECT")) {
    // The exact ranges cost a pass over the keys (0.25 of C2's 0.66 ms).  With ONE key column the window is first
    // guessed from a 65536-row prefix and widened to the whole id space; a row outside it raises a flag and the
    // attempt is repeated with the exact range.
    const bool guess_first = ncols == 1 && n > (1 << 20) && !lab::knob_on("GDF_GB_NO_GUESS");
    for (int attempt = guess_first ? 0 : 1; attempt < 2; ++attempt) {
      std::vector<long long> h(2 * ncols);
      KeyTable tr = t;
      if (attempt == 0) tr.nrows = 1 << 16;
      if (attempt == 1 && j.ranges.size() == h.size()) h = j.ranges;
      else GDF_TRY(key_ranges(tr, h.data()));
      GbDirect d{};
      d.ncols = ncols;
      uint64_t total = 1;
      for (int c = 0; c < ncols && total <= GB_DIRECT_MAX_IDS; ++c) {
        const uint64_t span = (uint64_t)h[2 * c + 1] - (uint64_t)h[2 * c] + 1;       // 0 on a full 64-bit range: caught below
        d.lo[c] = h[2 * c];
        d.span[c] = (uint32_t)span;
        total = (span == 0 || span > GB_DIRECT_MAX_IDS) ? GB_DIRECT_MAX_IDS + 1 : total * span;
      }
      if (total > GB_DIRECT_MAX_IDS) break;                     // the range is too wide even for the sample: dictionary paths
      if (attempt == 0) {
        // widen the guessed window symmetrically to the full id space (saturating at the int64 limits)
        const uint64_t room = (GB_DIRECT_MAX_IDS - total) / 2;
        const long long lo = d.lo[0], hi = h[1];
        const long long new_lo = (lo < (long long)(0x8000000000000000ULL + room)) ? (long long)0x8000000000000000ULL : lo - (long long)room;
        const long long new_hi = (hi > (long long)(0x7fffffffffffffffULL - room)) ? 0x7fffffffffffffffLL : hi + (long long)room;
        d.lo[0] = new_lo;
        d.span[0] = (uint32_t)((uint64_t)new_hi - (uint64_t)new_lo + 1);
        total = d.span[0];
      }
      for (int c = 0, below = (int)total; c < ncols; ++c) { below /= (int)d.span[c]; d.stride[c] = (uint32_t)below; }
      d.total = (uint32_t)total;
      DevBuf gacc, gcnt, ng;
      RMM_TRY(gacc.alloc(sizeof(uint64_t) * d.total));
      RMM_TRY(gcnt.alloc(sizeof(uint64_t) * d.total));
      RMM_TRY(ng.alloc(sizeof(unsigned int) * 2));
      const int fold_op = op == OP_AVG ? OP_SUM : op;
      GDF_LAUNCH("gb_fill", gb_direct_init, dim3(stream_grid(d.total, 256)), dim3(256), 0, stream0(), gacc.as<unsigned long long>(),
                 (unsigned long long)acc_identity_host(fold_op), gcnt.as<unsigned long long>(), d.total, ng.as<unsigned int>());
      const int agrid = stream_grid((size_t)n, GB_DENSE_THREADS * GB_DENSE_BATCH, NUM_CU);
      const int64_t achunk = (((n + agrid - 1) / agrid) + GB_DENSE_THREADS - 1) / GB_DENSE_THREADS * GB_DENSE_THREADS;
      const size_t dlds = (size_t)d.total * 12 + 16;
      const int fastkey = ncols == 1 && (t.col[0].width == 8 || t.col[0].width == 4) ? t.col[0].width : 0;
      const int fastval = (op != OP_COUNT && (in_kind == K_I64 || in_kind == K_F64)) ? 8 : ((op != OP_COUNT && (in_kind == K_I32 || in_kind == K_F32)) ? 4 : 0);
#define GB_DIRECT_LAUNCH(FK, FV)                                                                                             \
  do {                                                                                                                       \
    HIP_TRY(hipFuncSetAttribute((const void *)gb_direct_aggregate<FK, FV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds)); \
    GDF_LAUNCH("gb_direct_aggregate", (gb_direct_aggregate<FK, FV>), dim3(agrid), dim3(GB_DENSE_THREADS), dlds, stream0(), t, d, val, op, \
               gacc.as<unsigned long long>(), gcnt.as<unsigned long long>(), achunk, ng.as<unsigned int>() + 1);             \
  } while (0)
      if (fastkey == 8 && fastval == 8) GB_DIRECT_LAUNCH(8, 8);
      else if (fastkey == 8 && fastval == 4) GB_DIRECT_LAUNCH(8, 4);
      else if (fastkey == 8) GB_DIRECT_LAUNCH(8, 0);
      else if (fastkey == 4 && fastval == 8) GB_DIRECT_LAUNCH(4, 8);
      else if (fastkey == 4 && fastval == 4) GB_DIRECT_LAUNCH(4, 4);
      else if (fastkey == 4) GB_DIRECT_LAUNCH(4, 0);
      else if (fastval == 8) GB_DIRECT_LAUNCH(0, 8);
      else if (fastval == 4) GB_DIRECT_LAUNCH(0, 4);
      else GB_DIRECT_LAUNCH(0, 0);
#undef GB_DIRECT_LAUNCH
      GbOut o{};
      o.ncols = ncols;
      for (int c = 0; c < ncols; ++c) o.key_out[c] = out_keys[c]->data;
      o.agg_out = out_agg->data;
      o.in_kind = (int)in_kind;
      o.agg_kind = (int)((op == OP_COUNT || op == OP_AVG) ? out_kind : in_kind);
      DevBuf glast;
      if (j.sort_indices) {
        RMM_TRY(glast.alloc(sizeof(unsigned int) * d.total));
        HIP_TRY(hipMemsetAsync(glast.p, 0, sizeof(unsigned int) * d.total, stream0()));
        const size_t llds = (size_t)d.total * 4 + 16;
        HIP_TRY(hipFuncSetAttribute((const void *)gb_direct_last_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)llds));
        GDF_LAUNCH("gb_direct_last_rows", gb_direct_last_rows, dim3(agrid), dim3(GB_DENSE_THREADS), llds, stream0(), t, d, glast.as<unsigned int>(), achunk);
        o.indices = j.sort_indices;
        o.last_rows = glast.as<unsigned int>();
      }
      GDF_LAUNCH("gb_extract", gb_direct_extract, dim3((d.total + 1023) / 1024), dim3(1024), 0, stream0(), t, d, o, op, (const unsigned long long *)gacc.as<unsigned long long>(),
                 (const unsigned long long *)gcnt.as<unsigned long long>(), ng.as<unsigned int>());
      HIP_CHECK_LAST();
      unsigned int res[2] = {0, 0};                                                   // {groups, some row outside the window}
      HIP_TRY(read_back(res, ng.p, sizeof(res)));
      if (res[1]) continue;                                                          // guessed window too small: exact ranges next
      for (int c = 0; c < ncols; ++c) out_keys[c]->size = (gdf_size_type)res[0];
      out_agg->size = (gdf_size_type)res[0];
      *done = true;
      // (a SORT-method call leaves the caller's validity masks alone, as group_by_sort does: that method rejects
      // masks on the way in, sqls_ops.cu:1103-1106, and never writes one on the way out)
      if (j.sort_method) { HIP_TRY(hipStreamSynchronize(stream0())); return GDF_SUCCESS; }
      return write_output_masks(ncols, out_keys, out_agg, nullptr, res[0]);          // ids ascend: the output is already sorted
    }
  }

  return GDF_SUCCESS;
}

// Path 2 -- dense ids: packed keys and few enough groups for per-workgroup LDS accumulators.
static gdf_error gb_path_dense(GbJob &j, bool *done) {
  *done = false;
  [[maybe_unused]] const int ncols = j.ncols;
  [[maybe_unused]] gdf_column **out_keys = j.out_keys;
  [[maybe_unused]] gdf_column *out_agg = j.out_agg;
  [[maybe_unused]] const int op = j.op;
  [[maybe_unused]] const bool sort_result = j.sort_result;
  [[maybe_unused]] const KeyTable &t = j.t;
  [[maybe_unused]] const int64_t n = j.t.nrows;
  [[maybe_unused]] const ElemKind in_kind = j.in_kind, out_kind = j.out_kind;
  [[maybe_unused]] const GbKeyPlan &plan = j.plan;
  [[maybe_unused]] const GbVal &val = j.val;
  [[maybe_unused]] const bool masked = j.masked, avg = j.counted, want_ok = j.want_ok;
  [[maybe_unused]] DevBuf &agg_ok = j.agg_ok;
  // The table is sized by GROUPS.  Start small (the common case) and grow x256 on
  // overflow, up to the 2*N slots the reference always allocates.
  uint64_t cap_max = 1;
  while (cap_max < 2 * (uint64_t)n) cap_max <<= 1;
  // 2^18 entries of 16 bytes: at most 6 % load with the 16384 groups this path takes.  Smaller tables were tried for L2
  // locality and LOST -- C2's sparse twin, dict build + aggregate: 2^15 0.82 + 0.92 ms, 2^16 0.70 + 0.84, 2^17 0.64 + 0.80,
  // 2^18 0.61 + 0.77 -- a key that is not in its home slot costs a dependent walk, which matters more than the footprint.
  // GDF_GB_DICT_BITS: experiment switch.
  const int dict_bits = (int)lab::knob_int("GDF_GB_DICT_BITS", 18);
  const uint64_t tmax = 1ull << (dict_bits >= 15 && dict_bits <= 20 ? dict_bits : 18);
  uint64_t T = cap_max < tmax ? cap_max : tmax;
  if (plan.packed && !lab::knob_on("GDF_GB_NO_DENSE")) {
    DevBuf dict, flags, group_slot;
    RMM_TRY(dict.alloc(sizeof(GbDictEntry) * (T + 1)));
    RMM_TRY(flags.alloc(sizeof(unsigned int) * 8));           // [0..2] the dictionary's, [4..7] the LDS dictionary's state (gb_ld_image)
    HIP_TRY(hipMemsetAsync(flags.p, 0, sizeof(unsigned int) * 8, stream0()));
    GbDict g{};
    g.T = (uint32_t)T;
    g.e = dict.as<GbDictEntry>();
    g.occupied = flags.as<unsigned int>();
    g.overflow = flags.as<unsigned int>() + 1;
    g.special = flags.as<unsigned int>() + 2;
    const uint32_t max_groups = avg ? GB_DENSE_MAX_GROUPS_AVG : GB_DENSE_MAX_GROUPS;
    const bool fastkey = !masked && t.ncols == 1 && t.col[0].kind == K_I64 && plan.bits[0] == 64 && plan.bias[0] == 0;
    g.limit = max_groups;                       // more distinct keys than this: stop early, use the general path
    GDF_LAUNCH("gb_fill", gb_dict_clear, dim3(stream_grid(T + 1, 256 * 8)), dim3(256), 0, stream0(), g.e, (uint32_t)(T + 1));
    const int bgrid = stream_grid((size_t)n, GB_DICT_THREADS * GB_DENSE_BATCH * 4, NUM_CU * 8);
    const int64_t bchunk = (((n + bgrid - 1) / bgrid) + GB_DICT_THREADS - 1) / GB_DICT_THREADS * GB_DICT_THREADS;
    auto dict_build = [&](const KeyTable &tt, int grid_, int64_t chunk_, int64_t stride_ = 1) -> gdf_error {
      if (fastkey) GDF_LAUNCH("gb_dict_build", (gb_dict_build<true, false>), dim3(grid_), dim3(GB_DICT_THREADS), 0, stream0(), tt, plan, g, chunk_, stride_);
      else if (!t.any_valid) GDF_LAUNCH("gb_dict_build", (gb_dict_build<false, false>), dim3(grid_), dim3(GB_DICT_THREADS), 0, stream0(), tt, plan, g, chunk_, stride_);
      else GDF_LAUNCH("gb_dict_build", (gb_dict_build<false, true>), dim3(grid_), dim3(GB_DICT_THREADS), 0, stream0(), tt, plan, g, chunk_, stride_);
      return GDF_SUCCESS;
    };
    unsigned int h_flags[3] = {0, 0, 0};
    // LDS dictionary (see gb_ld_encode): unmasked rows only, and enough of them to pay for four small kernels and a read-back.
    // On success the rows' group ids are in `ids`, the dictionary is complete and numbered (group_slot).
    DevBuf ids;
    unsigned int *ld_state = flags.as<unsigned int>() + 4;
    bool have_ids = false;
    const bool try_ld = !masked && n >= ((int64_t)1 << 22) && !lab::path_on("GDF_GB_NO_LDS_DICT");
    const int64_t sample = 1 << 16;      // a quarter of the table's slots: the prefix cannot crowd it
    if (try_ld) {
      // sample -> number -> image -> encode, ONE read-back at the end: the kernels take the group count from the device and
      // stand down by themselves when the sample overflows the table or holds more groups than the image can
      const int64_t strided = 1 << 18;                 // every key that owns >= 1e-4 of the rows is in here with p > 1 - e^-26
      KeyTable ts = t;
      ts.nrows = strided;
      GDF_TRY(dict_build(ts, (int)(strided / (GB_DICT_THREADS * GB_DENSE_BATCH)), GB_DICT_THREADS * GB_DENSE_BATCH, n / strided));
      DevBuf tabimg, keyimg;
      RMM_TRY(tabimg.alloc(sizeof(uint32_t) * GB_LD_SLOTS));
      RMM_TRY(keyimg.alloc(sizeof(uint64_t) * GB_LD_MAX_GROUPS));
      RMM_TRY(ids.alloc(sizeof(uint16_t) * (size_t)n));
      RMM_TRY(group_slot.alloc(sizeof(uint32_t) * (size_t)(g.limit + 1)));     // the sample may hold up to `limit` groups (+ the reserved key)
      const int egrid = stream_grid((size_t)n, GB_LD_THREADS * GB_DENSE_BATCH, NUM_CU);
      const int64_t echunk = (((n + egrid - 1) / egrid) + GB_LD_THREADS - 1) / GB_LD_THREADS * GB_LD_THREADS;
      const size_t ilds = sizeof(uint32_t) * GB_LD_SLOTS + sizeof(uint64_t) * GB_LD_MAX_GROUPS;
      HIP_TRY(hipFuncSetAttribute((const void *)gb_ld_image, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds));
      HIP_TRY(hipFuncSetAttribute((const void *)gb_ld_encode<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds));
      HIP_TRY(hipFuncSetAttribute((const void *)gb_ld_encode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ilds));
      for (int round = 0; round < 2; ++round) {                    // round 1 only if round 0 met keys the sample had not
        if (round) HIP_TRY(hipMemsetAsync(ld_state, 0, sizeof(unsigned int) * 4, stream0()));      // [4] = the numbering counter
        GDF_LAUNCH("gb_dict_number", gb_dict_number, dim3(stream_grid(T + 1, 256 * 4)), dim3(256), 0, stream0(), g, 0xffffffffu,
                   group_slot.as<uint32_t>(), ld_state);
        GDF_LAUNCH("gb_ld_image", gb_ld_image, dim3(1), dim3(GB_LD_THREADS), ilds, stream0(), g, group_slot.as<uint32_t>(),
                   tabimg.as<uint32_t>(), keyimg.as<unsigned long long>(), ld_state);
        if (fastkey)
          GDF_LAUNCH("gb_ld_encode", gb_ld_encode<true>, dim3(egrid), dim3(GB_LD_THREADS), ilds, stream0(), t, plan, g, tabimg.as<uint32_t>(),
                     keyimg.as<unsigned long long>(), ids.as<uint16_t>(), echunk, ld_state);
        else
          GDF_LAUNCH("gb_ld_encode", gb_ld_encode<false>, dim3(egrid), dim3(GB_LD_THREADS), ilds, stream0(), t, plan, g, tabimg.as<uint32_t>(),
                     keyimg.as<unsigned long long>(), ids.as<uint16_t>(), echunk, ld_state);
        HIP_CHECK_LAST();
        unsigned int all[8];
        HIP_TRY(read_back(all, flags.p, sizeof(all)));
        for (int k = 0; k < 3; ++k) h_flags[k] = all[k];
        const unsigned int *st = all + 4;
        if (h_flags[1] || st[2]) break;          // table overflow: the general path.  No image (too many groups): the dense path.
        if (!st[1]) { have_ids = true; break; }
        // keys beyond the sample: every one of them is in the global table now (inserted by the rows that missed)
      }
    } else if (n > 16 * sample) {
      // a 65536-row prefix already tells "far too many groups" apart (C5) without paying for a
      // full pass that fills the table and gives up
      KeyTable ts = t;
      ts.nrows = sample;
      GDF_TRY(dict_build(ts, (int)(sample / (GB_DICT_THREADS * GB_DENSE_BATCH)), GB_DICT_THREADS * GB_DENSE_BATCH));
      HIP_TRY(read_back(h_flags, flags.p, sizeof(h_flags)));
    }
    if (!have_ids && !h_flags[1]) {
      GDF_TRY(dict_build(t, bgrid, bchunk));
      HIP_CHECK_LAST();
      HIP_TRY(read_back(h_flags, flags.p, sizeof(h_flags)));
    }
    const uint32_t ngroups = h_flags[0] + (h_flags[2] ? 1u : 0u);
    if (!h_flags[1] && ngroups <= max_groups) {
      DevBuf gacc, gcnt;
      RMM_TRY(gacc.alloc(sizeof(uint64_t) * (ngroups ? ngroups : 1)));
      if (avg) RMM_TRY(gcnt.alloc(sizeof(uint64_t) * (ngroups ? ngroups : 1)));
      if (!have_ids) {
        RMM_TRY(group_slot.alloc(sizeof(uint32_t) * (ngroups ? ngroups : 1)));
        HIP_TRY(hipMemsetAsync(flags.p, 0, sizeof(unsigned int), stream0()));     // reuse [0] as the numbering counter
        GDF_LAUNCH("gb_dict_number", gb_dict_number, dim3(stream_grid(T + 1, 256 * 4)), dim3(256), 0, stream0(), g, h_flags[2],
                   group_slot.as<uint32_t>(), flags.as<unsigned int>());
      }
      GDF_LAUNCH("gb_fill", gb_fill_u64, dim3(stream_grid(ngroups, 256)), dim3(256), 0, stream0(), gacc.as<unsigned long long>(),
                 (unsigned long long)(op == OP_MIN ? ~0ULL : 0ULL), ngroups);
      if (avg) HIP_TRY(hipMemsetAsync(gcnt.p, 0, sizeof(uint64_t) * ngroups, stream0()));
      const int agrid = stream_grid((size_t)n, GB_DENSE_THREADS * GB_DENSE_BATCH, NUM_CU);
      const int64_t achunk = (((n + agrid - 1) / agrid) + GB_DENSE_THREADS - 1) / GB_DENSE_THREADS * GB_DENSE_THREADS;
      const size_t dlds = (size_t)ngroups * (avg ? 12 : 8) + 16;
      const bool fastval = !masked && op != OP_COUNT && kind_width(in_kind) == 8;
#define GB_DENSE_LAUNCH(FK, FV, MK)                                                                                          \
  do {                                                                                                                       \
    HIP_TRY(hipFuncSetAttribute((const void *)gb_dense_aggregate<FK, FV, MK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds)); \
    GDF_LAUNCH("gb_dense_aggregate", (gb_dense_aggregate<FK, FV, MK>), dim3(agrid), dim3(GB_DENSE_THREADS), dlds, stream0(), t, plan, val, \
               op, g, ngroups, gacc.as<unsigned long long>(), gcnt.as<unsigned long long>(), achunk);                        \
  } while (0)
      if (have_ids) {
        const int fv = op == OP_COUNT ? 0 : (kind_width(in_kind) == 8 ? 8 : (kind_width(in_kind) == 4 ? 4 : 0));
        const int lgrid = stream_grid((size_t)n, GB_LD_THREADS * GB_DENSE_BATCH, NUM_CU);
        const int64_t lchunk = (((n + lgrid - 1) / lgrid) + GB_LD_THREADS - 1) / GB_LD_THREADS * GB_LD_THREADS;
#define GB_LD_LAUNCH(FV)                                                                                                     \
  do {                                                                                                                       \
    HIP_TRY(hipFuncSetAttribute((const void *)gb_ld_aggregate<FV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dlds)); \
    GDF_LAUNCH("gb_ld_aggregate", gb_ld_aggregate<FV>, dim3(lgrid), dim3(GB_LD_THREADS), dlds, stream0(), val, op, ids.as<uint16_t>(), n, \
               ngroups, gacc.as<unsigned long long>(), gcnt.as<unsigned long long>(), lchunk);                               \
  } while (0)
        if (fv == 8) GB_LD_LAUNCH(8);
        else if (fv == 4) GB_LD_LAUNCH(4);
        else GB_LD_LAUNCH(0);
#undef GB_LD_LAUNCH
      } else if (masked) GB_DENSE_LAUNCH(false, false, true);
      else if (fastkey && fastval) GB_DENSE_LAUNCH(true, true, false);
      else if (fastkey) GB_DENSE_LAUNCH(true, false, false);
      else if (fastval) GB_DENSE_LAUNCH(false, true, false);
      else GB_DENSE_LAUNCH(false, false, false);
#undef GB_DENSE_LAUNCH
      GbOut o{};
      o.ncols = ncols;
      for (int c = 0; c < ncols; ++c) o.key_out[c] = out_keys[c]->data;
      o.agg_out = out_agg->data;
      o.in_kind = (int)in_kind;
      o.agg_kind = (int)((op == OP_COUNT || op == OP_AVG) ? out_kind : in_kind);
      if (want_ok) RMM_TRY(agg_ok.alloc(ngroups ? ngroups : 1));
      o.agg_ok = agg_ok.as<uint8_t>();
      o.counted = val.valid != nullptr;
      GDF_LAUNCH("gb_extract", gb_dense_extract, dim3(stream_grid(ngroups ? ngroups : 1, 256)), dim3(256), 0, stream0(), t, plan, g,
                 group_slot.as<uint32_t>(), ngroups, o, op, gacc.as<unsigned long long>(), gcnt.as<unsigned long long>());
      HIP_CHECK_LAST();
      HIP_TRY(hipStreamSynchronize(stream0()));
      for (int c = 0; c < ncols; ++c) out_keys[c]->size = (gdf_size_type)ngroups;
      out_agg->size = (gdf_size_type)ngroups;
      if (sort_result || op == OP_AVG) {
        int kinds[MAX_KEY_COLS];
        for (int c = 0; c < ncols; ++c) kinds[c] = t.col[c].kind;
        GDF_TRY(sort_result_rows(ncols, out_keys, kinds, out_agg->data, kind_width((ElemKind)o.agg_kind), ngroups, o.agg_ok));
      }
      *done = true;
      return write_output_masks(ncols, out_keys, out_agg, o.agg_ok, ngroups);
    }
    // too many groups for LDS accumulators
  }

  return GDF_SUCCESS;
}

// The partitioned variant of the sorted path (see the kernels above), for key type K = uint32_t (12-byte pairs,
// when key bits + valid bit + null bit fit 32) or uint64_t.
template <class K>
static gdf_error gb_sorted_partitioned(GbJob &j, const GbKeyPlan &sp, int vbit, int null_bit, bool *done, bool guessed = false, bool allow_spec = true) {
  [[maybe_unused]] const int ncols = j.ncols;
  [[maybe_unused]] gdf_column **out_keys = j.out_keys;
  [[maybe_unused]] gdf_column *out_agg = j.out_agg;
  [[maybe_unused]] const int op = j.op;
  [[maybe_unused]] const bool sort_result = j.sort_result;
  [[maybe_unused]] const KeyTable &t = j.t;
  [[maybe_unused]] const int64_t n = j.t.nrows;
  [[maybe_unused]] const ElemKind in_kind = j.in_kind, out_kind = j.out_kind;
  [[maybe_unused]] const GbKeyPlan &plan = j.plan;
  [[maybe_unused]] const GbVal &val = j.val;
  [[maybe_unused]] const bool masked = j.masked, avg = j.counted, want_ok = j.want_ok;
  [[maybe_unused]] DevBuf &agg_ok = j.agg_ok;
  const uint32_t nn = (uint32_t)n;
  const int id_bits = sp.total_bits < GB_PART_ID_BITS ? sp.total_bits : GB_PART_ID_BITS;
  const int part_bits = sp.total_bits - id_bits;
  const int fold_op = op == OP_AVG ? OP_SUM : op;
  const bool flt = is_flt(val.kind);
  // FUSED: one partition pass straight from the raw columns (gbp_count / gbp_scatter) instead of pair build + radix sort
  bool fused = false;
  if constexpr (sizeof(K) == 4)
    fused = part_bits >= 1 && part_bits <= GBP_MAX_PART_BITS && n >= ((int64_t)1 << 20) && !lab::knob_on("GDF_GB_NO_FUSED");
  if (guessed && !fused) { *done = false; return GDF_SUCCESS; }      // only the fused kernels check keys against a guessed plan
  DevBuf ka, kb, pa, pb, fl;
  if (fused) {
    // (12-byte records, key and image together: allocated below, once the layout -- exact or speculative -- is known)
  } else {
    RMM_TRY(ka.alloc(sizeof(K) * (size_t)nn));
    RMM_TRY(pa.alloc(sizeof(uint64_t) * (size_t)nn));
  }
  if (!fused) {
    RMM_TRY(kb.alloc(sizeof(K) * (size_t)nn));
    RMM_TRY(pb.alloc(sizeof(uint64_t) * (size_t)nn));
  }
  RMM_TRY(fl.alloc(16));
  HIP_TRY(hipMemsetAsync(fl.p, 0, 16, stream0()));
  const uint64_t null_key = null_bit ? (1ULL << (sp.total_bits + vbit)) : 0ULL;
  K *kin = ka.as<K>(), *kout = kb.as<K>();
  uint64_t *pin = pa.as<uint64_t>(), *pout = pb.as<uint64_t>();
  struct { unsigned long long varying; unsigned int dropped, pad; } hf{};
  std::vector<uint32_t> hp;             // partition starts (fused: from the scanned histogram)
  DevBuf gacc, grows, gvalid;           // the global cells (one per key: accumulator, rows, valid values)
  bool cells_ready = false;             // made before the scatter kernel when it aggregates a hot window (GbHot)
  uint32_t hot_window_cells = GBP_NO_HOT;   // the hot window the scatter kernel merged into the cells itself, if any
  GbSpec aggregate_spec{};              // the speculative record layout, when the fused pass ran on it (the plan's device arrays: keep_spec)
  DevBuf keep_spec;
  // speculative layout: the scatter kernel's flags are looked at LATE -- their copy is queued right behind the kernel, the aggregation
  // and its counting passes behind the copy, and the host comes back for the flags before it launches the extraction (round 5: the
  // read-back and the unit list stood between the two big kernels, 0.06 ms of idle GPU, and a synchronising scan behind them)
  DevBuf late_flags;
  ReadTicket late_ticket;
  bool late = false;
  if (fused) {
    if constexpr (sizeof(K) == 4) {
      const uint32_t P = 1u << part_bits;
      const int low = vbit + id_bits;
      int max_chunks = GBP_MAX_CHUNKS;
      if (lab::knob_int("GDF_GBP_CHUNKS", 0) > 0) max_chunks = (int)lab::knob_int("GDF_GBP_CHUNKS", 0);      // experiment switch
      int64_t chunk = (n + max_chunks - 1) / max_chunks;
      chunk = (chunk + GBP_TILE - 1) / GBP_TILE * GBP_TILE;
      const int nchunks = (int)((n + chunk - 1) / chunk);
      DevBuf hist, d_start, d_flags;
      RMM_TRY(d_start.alloc(sizeof(uint32_t) * ((size_t)P + 1)));
      RMM_TRY(d_flags.alloc(sizeof(unsigned int) * 4));          // rows dropped for a null key | range violated | speculative layout overflowed
      HIP_TRY(hipMemsetAsync(d_flags.p, 0, sizeof(unsigned int) * 4, stream0()));

      // static key signature (gbp_pack32): one or two 4- / 8-byte integer key columns
      const bool no_static = lab::path_on("GDF_GBP_DYNAMIC");          // (read per call: the tests flip it)
      auto int_kind = [](int k) { return k == K_I32 || k == K_I64; };
      const bool key_sig = !no_static && (t.ncols == 1 || t.ncols == 2) && int_kind(t.col[0].kind) && (t.ncols == 1 || int_kind(t.col[1].kind));
      const int k0 = key_sig ? t.col[0].kind : -1, k1 = !key_sig ? -1 : (t.ncols == 2 ? t.col[1].kind : -2);
      // The count only needs the partition id.  A second key column whose bit field lies entirely below the id bits that select the
      // partition (C5: int32 values 0..15 under 2^13 ids) and that has no nulls cannot change a row's partition or drop the row:
      // the count does not read it (8 instead of 12 B per row on C5) and the scatter kernel, which reads every column anyway,
      // checks its values against a guessed range.  Only with the statically typed scatter kernels (they carry that check).
      const bool val_sig = (val.kind == K_I64 || val.kind == K_F64) && fold_op != OP_COUNT && (!vbit || val.valid) && !lab::knob_on("GDF_GBP_OLD");
      const bool skip_low = key_sig && val_sig && t.ncols == 2 && !t.col[1].valid && sp.shift[1] + sp.bits[1] + vbit <= low &&
                            !lab::knob_on("GDF_GBP_COUNT_ALL");
      const int ck1 = skip_low ? -2 : k1;
      // record layout: partition-major -- hist[p][chunk], one scan gives every (partition, chunk) its place inside the partition's
      // contiguous range.  (LAB knob GDF_GBP_CHUNK_MAJOR: [chunk][partition] segments, every workgroup's 2048 write fronts inside
      // its own ~12 MB window -- the layout experiment of profiles/r3_*_c5_layout.md; the aggregation does not read that layout,
      // the knob only times the scatter.)
      const bool chunk_major = lab::knob_on("GDF_GBP_CHUNK_MAJOR");
      const uint32_t qstride = chunk_major ? 1u : (uint32_t)nchunks, cstride = chunk_major ? P : 1u;
      // HOT WINDOW (GbHot): the densest aligned window of GBP_HOT_IDS ids in a strided sample, when it holds enough of the rows to
      // pay for 64 KB of LDS in the scatter kernel.  Statically typed scatter kernels only; a key column the count does not read
      // (skip_low) must lie below the window bits, so that the count can tell hot rows from their first column alone.
      uint32_t hot_window = GBP_NO_HOT;
      const bool lean_sig = key_sig && val_sig && !lab::knob_on("GDF_GBP_OLD");
      const dim3 sgrid(nchunks < NUM_CU ? nchunks : NUM_CU);
      const bool hot_ok = lean_sig && id_bits == GB_PART_ID_BITS && !chunk_major && n >= ((int64_t)1 << 22) && !lab::path_on("GDF_GBP_NO_HOT");
      // SPECULATIVE layout (GbSpec): no count pass; GDF_GBP_SPEC_MIN_ROWS / GDF_GBP_NO_SPEC: test switches
      const bool spec_wanted = allow_spec && lean_sig && id_bits == GB_PART_ID_BITS && !chunk_major && !lab::path_on("GDF_GBP_NO_SPEC") &&
                               n >= lab::path_int("GDF_GBP_SPEC_MIN_ROWS", (long long)1 << 24);
      GbSpec spec{};
      int plain_rank = 0;                 // GbHot::plain_rank, decided from the sample below
      DevBuf d_spec;                      // cap [P] | qprefix [P + 1] | fill [P * G]
      const size_t hot_cells_pad = (((size_t)P << id_bits) + 1023) / 1024 * 1024;
      auto fill_cells = [&]() -> gdf_error {          // (again behind every calibration run of the placement tournament below)
        GDF_LAUNCH("gb_fill", gb_fill_u64, dim3(stream_grid(hot_cells_pad, 1024)), dim3(256), 0, stream0(), gacc.as<unsigned long long>(),
                   (unsigned long long)acc_identity_host(fold_op), (uint32_t)hot_cells_pad);
        HIP_TRY(hipMemsetAsync(grows.p, 0, sizeof(unsigned int) * hot_cells_pad, stream0()));
        if (vbit) HIP_TRY(hipMemsetAsync(gvalid.p, 0, sizeof(unsigned int) * hot_cells_pad, stream0()));
        return GDF_SUCCESS;
      };
      auto make_cells = [&]() -> gdf_error {
        RMM_TRY(gacc.alloc(sizeof(uint64_t) * hot_cells_pad));
        RMM_TRY(grows.alloc(sizeof(unsigned int) * hot_cells_pad));
        if (vbit) RMM_TRY(gvalid.alloc(sizeof(unsigned int) * hot_cells_pad));
        GDF_TRY(fill_cells());
        cells_ready = true;
        return GDF_SUCCESS;
      };
      if (hot_ok || spec_wanted) {
        const uint32_t nwin = 1u << (sp.total_bits - GBP_HOT_BITS);
        const uint32_t wpp = 1u << (id_bits - GBP_HOT_BITS);                  // sample windows per partition
        DevBuf d_cnt;
        RMM_TRY(d_cnt.alloc(sizeof(unsigned int) * ((size_t)nwin + 1)));
        HIP_TRY(hipMemsetAsync(d_cnt.p, 0, sizeof(unsigned int) * ((size_t)nwin + 1), stream0()));
        GDF_LAUNCH("gbp_sample_hist", gbp_sample_hist, dim3(GBP_SAMPLE_WINDOWS / GBP_SAMPLE_PER_WG), dim3(1024), 0, stream0(), t, sp, nwin,
                   d_cnt.as<unsigned int>());
        HIP_CHECK_LAST();
        // the sample's copy is queued, and behind it the cells every outcome of this pass needs: they are cleared while the host
        // reads the sample and lays the segments out (round 5; they used to follow that, in front of the scatter kernel)
        std::vector<unsigned int> cnt((size_t)nwin + 1);
        ReadTicket sample_ticket;
        HIP_TRY(read_back_begin(&sample_ticket, d_cnt.p, sizeof(unsigned int) * cnt.size(), 0));
        GDF_TRY(make_cells());
        HIP_TRY(read_back_end(&sample_ticket, cnt.data()));
        const double S = (double)cnt[nwin];
        uint32_t best = 0;
        for (uint32_t w = 1; w < nwin; ++w) if (cnt[w] > cnt[best]) best = w;
        // the hot window: worth its 64 KB of LDS from a fifth of the rows (a key column the count pass does not read must lie
        // below the window bits -- only the exact layout has a count pass)
        const long long forced = lab::path_int("GDF_GBP_HOT_WINDOW", -1);       // test switch: any window gives the same result
        if (hot_ok && forced >= 0) hot_window = (uint32_t)forced < nwin ? (uint32_t)forced : nwin - 1;
        else if (hot_ok && S >= 1024.0 && (double)cnt[best] >= 0.2 * S) hot_window = best;
        // ranking without the leader ballots (gbp_rank_plain) when no partition holds more than an eighth of the rows that are
        // ranked, i.e. a wave's worst same-address LDS atomic serves ~8 lanes (GDF_GBP_PLAIN_RANK = 0 / 1: test switch)
        if (S >= 65536.0) {
          double ranked = 0, busiest_part = 0;
          for (uint32_t q = 0; q < P; ++q) {
            double sq = 0;
            for (uint32_t i = 0; i < wpp; ++i) {
              const uint32_t w = q * wpp + i;
              if (w < nwin && w != hot_window) sq += (double)cnt[w];
            }
            ranked += sq;
            busiest_part = std::max(busiest_part, sq);
          }
          plain_rank = busiest_part <= 0.125 * ranked ? 1 : 0;
        }
        plain_rank = (int)lab::path_int("GDF_GBP_PLAIN_RANK", plain_rank);
        if (spec_wanted && S >= 65536.0) {
          // room per (partition, workgroup): the sample's estimate of the partition's cold rows + 5 sigma of that estimate, shared
          // out over G workgroups, + 6 sigma of a workgroup's own share (Poisson) -- a segment overflows about once in 1e8
          // XCD-shared segments (GbSpec::xcd): the DEFAULT since the ranks inside a tile are plain atomics (C5 in alternating processes of
          // one box: 9.45 - 9.54 against 10.12 - 10.19 ms, the scatter kernel 7.33 against 7.95, profiles/r4_o_c5_xcd_shared_ab.txt; it
          // was 10.67 against 10.9 - 11.4 when it was measured first, r4_k).  A run's place is claimed with a relaxed AGENT-scope atomic
          // (round 5: defined for counters shared across workgroups; round 4 shipped workgroup scope, which only the part made right) on
          // a counter chosen by HW_REG_XCC_ID, so that an XCD's appends stay together; the kernel boundary writes the L2s back before
          // the aggregation reads fill counts and records.  GDF_GBP_NO_XCD keeps the per-workgroup segments (no atomics at all), and the
          // tests run both layouts against the oracle.
          const bool xcd_mode = (sgrid.x & 7u) == 0 && !lab::path_on("GDF_GBP_NO_XCD");
          const uint32_t G = xcd_mode ? 8u : sgrid.x;
          // the rows the BUSIEST workgroup (XCD) gets (the kernel's chunk -> workgroup map: XCD x takes the x-th eighth of the chunks, its
          // workgroups round-robin inside it; chunks are whole tiles, so a small table leaves some workgroups a chunk more)
          int64_t busiest = 0;
          {
            const uint32_t WG = sgrid.x;
            std::vector<int64_t> rows_of(WG, 0);
            const bool xcd_map = (WG & 7u) == 0;
            const int per_xcd = (nchunks + 7) / 8;
            for (int c = 0; c < nchunks; ++c) {
              const int64_t r = std::min<int64_t>(chunk, n - (int64_t)c * chunk);
              const uint32_t wg = xcd_map ? (uint32_t)(c / per_xcd) + 8u * (uint32_t)((c % per_xcd) % (int)(WG >> 3)) : (uint32_t)c % WG;
              rows_of[xcd_mode ? (wg & 7u) : wg] += r;               // (XCD-shared segments: workgroup b runs on XCD b % 8)
            }
            for (uint32_t w = 0; w < (xcd_mode ? 8u : WG); ++w) busiest = std::max(busiest, rows_of[w]);
          }
          const double scale = (double)busiest / S;
          std::vector<uint32_t> plan_words(2 * (size_t)P + 1);
          uint32_t *capv = plan_words.data(), *pre = capv + P;
          uint64_t total = 0;
          for (uint32_t q = 0; q < P; ++q) {
            double sq = 0;
            for (uint32_t i = 0; i < wpp; ++i) {
              const uint32_t w = q * wpp + i;
              if (w < nwin && w != hot_window) sq += (double)cnt[w];
            }
            const double U = (sq + 5.0 * std::sqrt(sq + 1.0) + 3.0) * scale;
            const uint64_t c = (uint64_t)(U + 6.0 * std::sqrt(U) + 8.0);
            capv[q] = (uint32_t)std::min<uint64_t>(c, 0x7fffffffULL);
            pre[q] = (uint32_t)std::min<uint64_t>(total, 0xffffffffULL);
            total += capv[q];
          }
          pre[P] = (uint32_t)std::min<uint64_t>(total, 0xffffffffULL);
          // record positions are 31-bit in the kernels; the buffer must not dwarf the relation either (a flat sample: 2.5 x)
          if (total * G < 0x7fffffffULL && total * G <= (uint64_t)(2.5 * (double)nn) + ((uint64_t)P * G * 64)) {
            RMM_TRY(d_spec.alloc(sizeof(uint32_t) * (plan_words.size() + (size_t)P * G)));
            HIP_TRY(hipMemcpyAsync(d_spec.p, plan_words.data(), sizeof(uint32_t) * plan_words.size(), hipMemcpyHostToDevice, stream0()));
            HIP_TRY(hipStreamSynchronize(stream0()));                  // (plan_words is a local)
            spec.cap = d_spec.as<uint32_t>();
            spec.qprefix = spec.cap + P;
            spec.fill = d_spec.as<uint32_t>() + plan_words.size();
            spec.G = G;
            spec.xcd = xcd_mode ? 1 : 0;
            if (xcd_mode) HIP_TRY(hipMemsetAsync(spec.fill, 0, sizeof(uint32_t) * (size_t)P * G, stream0()));
            // (a PLACED block, DevBuf::alloc_placed: the scatter kernel's thousands of write fronts have their faster and slower physical
            // placements of this buffer -- 6.98 to 7.35 ms for C5 across re-allocations, profiles/r5_d_c5_scatter_after_reallocation.jsonl;
            // tournament below)
            place_budget_begin();          // (the call's budget for candidate blocks, internal.h)
            RMM_TRY(ka.alloc_placed(GB_ROLE_RECORDS, sizeof(GbRec) * (size_t)(total * G), place_draws_now(GB_PLACE_DRAWS)));
            kin = ka.as<K>();
            hp.resize((size_t)P + 1);
            for (uint32_t q = 0; q <= P; ++q) hp[q] = pre[q] * G;
          }
        }
      }
      const bool is_spec = spec.qprefix != nullptr;
      if (!is_spec) {
        RMM_TRY(ka.alloc(sizeof(GbRec) * (size_t)nn));
        kin = ka.as<K>();
        RMM_TRY(hist.alloc(sizeof(uint32_t) * ((size_t)P * nchunks + 1)));
        HIP_TRY(hipMemsetAsync(hist.as<uint32_t>() + (size_t)P * nchunks, 0, sizeof(uint32_t), stream0()));
      }
      // (exact layout: a key column the count pass skips must lie below the window bits, so that it can tell hot rows from the first column alone)
      if (!is_spec && skip_low && sp.shift[1] + sp.bits[1] > GBP_HOT_BITS) hot_window = GBP_NO_HOT;
      // the cells the partial aggregates are merged into: made BEFORE the scatter kernel, which merges the hot window's (make_cells above)
      if (hot_window != GBP_NO_HOT && !cells_ready) GDF_TRY(make_cells());
      const GbHot hot{hot_window, gacc.as<unsigned long long>(), grows.as<unsigned int>(), gvalid.as<unsigned int>(),
                      (int)lab::knob_int("GDF_GBP_HOT_DBG", 0), plain_rank};
      auto count = [&](auto kernel) {
        GDF_LAUNCH("gbp_count", kernel, dim3(nchunks < NUM_CU * 2 ? nchunks : NUM_CU * 2), dim3(GBP_THREADS), 0, stream0(), t, sp, low, vbit, P,
                   chunk, nchunks, hist.as<uint32_t>(), d_flags.as<unsigned int>(), qstride, cstride, hot_window);
      };
      if (is_spec) {}                     // no count pass: the scatter kernel appends to per-workgroup segments
      else if (k0 == K_I32 && ck1 == -2) count(gbp_count<K_I32, -2>);
      else if (k0 == K_I64 && ck1 == -2) count(gbp_count<K_I64, -2>);
      else if (k0 == K_I32 && ck1 == K_I32) count(gbp_count<K_I32, K_I32>);
      else if (k0 == K_I32 && ck1 == K_I64) count(gbp_count<K_I32, K_I64>);
      else if (k0 == K_I64 && ck1 == K_I32) count(gbp_count<K_I64, K_I32>);
      else if (k0 == K_I64 && ck1 == K_I64) count(gbp_count<K_I64, K_I64>);
      else count(gbp_count<-1, -1>);
      if (!is_spec) GDF_TRY(scan_u32(hist.as<uint32_t>(), hist.as<uint32_t>(), (size_t)P * nchunks + 1, false));
      const bool is_hot = hot_window != GBP_NO_HOT;
      const size_t slds = gbp_scatter_lds(is_hot);
      const bool lean = !lab::knob_on("GDF_GBP_OLD");              // A/B switch: the scatter kernel with the type switches for every shape
      const bool sig = lean && key_sig && val_sig;
      int launch_chunks = nchunks;       // (a calibration run of the placement tournament takes the first quarter of the chunks)
      auto run_scatter = [&]() -> gdf_error {
      if (sig) {
        const int vm = vbit ? 2 : (val.valid ? 1 : 0);        // 0: no mask, 1: mask, 2: mask + validity bit in the key
        auto scatter = [&](auto kernel) -> gdf_error {
          HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
          GDF_LAUNCH(is_hot ? "gbp_scatter_hot" : "gbp_scatter", kernel, sgrid, dim3(GBP_SC_THREADS), slds, stream0(), t, sp, val, fold_op, low, P, chunk,
                     launch_chunks, (const uint32_t *)hist.as<uint32_t>(), ka.as<GbRec>(), d_flags.as<unsigned int>(), qstride, cstride, hot, spec);
          return GDF_SUCCESS;
        };
#define GBP_SIG(K0, K1)                                                                                                          \
        if (k0 == K0 && k1 == K1) {                                                                                                 \
          if (vm == 2 && is_hot && is_spec) GDF_TRY(scatter(gbp_scatter_static<true, K0, K1, true, true, true>));                  \
          else if (vm == 0 && is_hot && is_spec) GDF_TRY(scatter(gbp_scatter_static<false, K0, K1, false, true, true>));           \
          else if (vm == 2 && is_spec) GDF_TRY(scatter(gbp_scatter_static<true, K0, K1, true, false, true>));                      \
          else if (vm == 0 && is_spec) GDF_TRY(scatter(gbp_scatter_static<false, K0, K1, false, false, true>));                    \
          else if (vm == 2 && is_hot) GDF_TRY(scatter(gbp_scatter_static<true, K0, K1, true, true>));                              \
          else if (vm == 0 && is_hot) GDF_TRY(scatter(gbp_scatter_static<false, K0, K1, false, true>));                            \
          else if (vm == 2) GDF_TRY(scatter(gbp_scatter_static<true, K0, K1, true>));                                              \
          else if (vm == 1) GDF_TRY(scatter(gbp_scatter_static<false, K0, K1, true>));                                             \
          else GDF_TRY(scatter(gbp_scatter_static<false, K0, K1, false>));                                                         \
        }
        GBP_SIG(K_I32, -2) GBP_SIG(K_I64, -2) GBP_SIG(K_I32, K_I32) GBP_SIG(K_I32, K_I64) GBP_SIG(K_I64, K_I32) GBP_SIG(K_I64, K_I64)
#undef GBP_SIG
      } else {
        auto scatter = [&](auto kernel) -> gdf_error {
          HIP_TRY(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds));
          GDF_LAUNCH("gbp_scatter", kernel, sgrid, dim3(GBP_SC_THREADS), slds, stream0(), t, sp, val, fold_op, low, part_bits, P, chunk, launch_chunks,
                     (const uint32_t *)hist.as<uint32_t>(), ka.as<GbRec>(), qstride, cstride);
          return GDF_SUCCESS;
        };
        if (vbit) GDF_TRY(scatter(gbp_scatter<true>));
        else GDF_TRY(scatter(gbp_scatter<false>));
      }
      return GDF_SUCCESS;
      };
      // PLACEMENT TOURNAMENT of the speculative record buffer (as for the join's tuple buffers, join.hip): while the pool is comparing
      // placements for this size, every candidate is timed on the real kernel over the first quarter of the chunks; the fill counters,
      // the flags and the hot window's cells are set back behind each run
      if (is_spec && sig && !lab::knob_on("GDF_GBP_NO_CALIBRATE")) {
        const size_t rec_bytes = sizeof(GbRec) * (size_t)(hp[P]);
        for (int round = 0; round <= GB_PLACE_DRAWS && ka.measure; ++round) {
          PlaceRound charge;
          launch_chunks = std::max(1, nchunks / 4);
          ka.clock_begin(stream0());
          GDF_TRY(run_scatter());
          ka.clock_end(stream0());
          HIP_TRY(hipMemsetAsync(d_flags.p, 0, sizeof(unsigned int) * 4, stream0()));
          HIP_TRY(hipMemsetAsync(spec.fill, 0, sizeof(uint32_t) * (size_t)P * spec.G, stream0()));
          if (is_hot) GDF_TRY(fill_cells());
          RMM_TRY(ka.alloc_placed(GB_ROLE_RECORDS, rec_bytes, place_draws_now(GB_PLACE_DRAWS)));
          kin = ka.as<K>();
        }
        launch_chunks = nchunks;
      }
      GDF_TRY(run_scatter());
      if (chunk_major) {                 // (LAB) the scatter has been timed; the records are not in the layout the aggregation reads
        HIP_TRY(hipStreamSynchronize(stream0()));
        *done = false;
        return GDF_SUCCESS;
      }
      if (!is_spec) {
        hipLaunchKernelGGL(gb_strided_u32, dim3((P + 256) / 256), dim3(256), 0, stream0(), (const uint32_t *)hist.as<uint32_t>(), d_start.as<uint32_t>(),
                           (int)P + 1, (size_t)nchunks);
        HIP_CHECK_LAST();
        hp.resize((size_t)P + 1);
        HIP_TRY(read_back(hp.data(), d_start.p, sizeof(uint32_t) * ((size_t)P + 1)));
      }
      if (is_spec) {
        HIP_TRY(read_back_begin(&late_ticket, d_flags.p, sizeof(unsigned int) * 3, 0));
        late_flags.p = d_flags.release();
        late = true;
        hf.dropped = 0;                 // (nobody counted them; 0 makes nvalid an upper bound, which is all it is used for)
      } else {
        unsigned int hfl[3] = {0, 0, 0};
        HIP_TRY(read_back(hfl, d_flags.p, sizeof(hfl)));
        hf.dropped = hfl[0];
        j.range_violated = hfl[1] != 0;
        if (j.range_violated) { *done = false; return GDF_SUCCESS; }      // sample-guessed ranges did not hold: the caller retries exactly
      }
      aggregate_spec = spec;
      keep_spec.p = d_spec.release();
      hot_window_cells = hot_window;
    }
  } else {
    GDF_LAUNCH("gb_sorted_make_pairs", gb_sorted_make_pairs<K>, dim3(stream_grid((size_t)n, 256 * 8)), dim3(256), 0, stream0(), t, sp, val,
               fold_op, vbit, null_key, kin, pin, fl.as<unsigned long long>(), (unsigned int *)(fl.as<unsigned long long>() + 1));
    HIP_TRY(read_back(&hf, fl.p, 16));
  }
  // 0: go on; 1: the sample-guessed ranges did not hold (the caller retries exactly); 2: a segment of the speculative layout ran out of
  // room (clustered input, or one chance in ~1e8 per segment): everything again on the exact layout.  Whatever was queued behind the
  // scatter kernel ran on what it left -- inside its buffers (gb_part_aggregate masks slots by the segments' capacities) -- and is waited for
  auto late_verdict = [&](int *verdict) -> gdf_error {
    *verdict = 0;
    if (!late) return GDF_SUCCESS;
    late = false;
    unsigned int hfl[3] = {0, 0, 0};
    HIP_TRY(read_back_end(&late_ticket, hfl));
    j.range_violated = hfl[1] != 0;
    if (j.range_violated) *verdict = 1;
    else if (hfl[2]) *verdict = 2;
    if (*verdict) HIP_TRY(hipStreamSynchronize(stream0()));
    return GDF_SUCCESS;
  };
  auto late_exit = [&](int verdict) -> gdf_error {
    if (verdict == 1) { *done = false; return GDF_SUCCESS; }
    ka.reset(); gacc.reset(); grows.reset(); gvalid.reset(); keep_spec.reset();
    return gb_sorted_partitioned<K>(j, sp, vbit, null_bit, done, guessed, false);
  };
  const uint32_t nvalid = nn - hf.dropped;
  uint32_t ngroups = 0;
  GbOut o{};
  o.ncols = ncols;
  for (int c = 0; c < ncols; ++c) o.key_out[c] = out_keys[c]->data;
  o.agg_out = out_agg->data;
  o.in_kind = (int)in_kind;
  o.agg_kind = (int)((op == OP_COUNT || op == OP_AVG) ? out_kind : in_kind);
  o.counted = val.valid != nullptr;
  if (nvalid) {
    const int low = vbit + id_bits;
    const uint32_t P = 1u << part_bits;
    const size_t cells = (size_t)P << id_bits;
    const size_t cells_pad = (cells + 1023) / 1024 * 1024;
    DevBuf pstart, d_units, bcnt, ng;
    if (!fused) {
      const uint64_t himask = low >= 64 ? 0ULL : ~((1ULL << low) - 1ULL);
      if constexpr (sizeof(K) == 4) GDF_TRY(radix_sort_pairs_k32_u64(kin, kout, pin, pout, nn, hf.varying & himask));
      else GDF_TRY(radix_sort_pairs_u64(kin, kout, pin, pout, nn, hf.varying & himask));
      RMM_TRY(pstart.alloc(sizeof(uint32_t) * ((size_t)P + 1)));
      GDF_LAUNCH("gb_part_bounds", gb_part_bounds<K>, dim3(stream_grid((size_t)P + 1, 256)), dim3(256), 0, stream0(), (const K *)kin, nvalid, low, P,
                 pstart.as<uint32_t>());
      hp.resize((size_t)P + 1);
      HIP_TRY(read_back(hp.data(), pstart.p, sizeof(uint32_t) * ((size_t)P + 1)));
    }
    std::vector<GbPartUnit> units;
    for (uint32_t p = 0; p < P; ++p) {
      // pad = 1: the partition's only unit and no hot window inside its cells -- it stores its aggregates instead of adding them
      const bool alone = hp[p + 1] - hp[p] <= GB_PART_UNIT_ROWS &&
                         (hot_window_cells == GBP_NO_HOT || (hot_window_cells >> (id_bits - GBP_HOT_BITS)) != p);
      for (uint32_t b = hp[p]; b < hp[p + 1]; b += GB_PART_UNIT_ROWS)
        units.push_back(GbPartUnit{b, std::min(GB_PART_UNIT_ROWS, hp[p + 1] - b), p, alone ? 1u : 0u});
    }
    RMM_TRY(d_units.alloc(sizeof(GbPartUnit) * (units.size() ? units.size() : 1)));
    HIP_TRY(hipMemcpyAsync(d_units.p, units.data(), sizeof(GbPartUnit) * units.size(), hipMemcpyHostToDevice, stream0()));
    RMM_TRY(bcnt.alloc(sizeof(uint32_t) * (cells_pad / 1024)));
    RMM_TRY(ng.alloc(sizeof(unsigned int)));
    if (!cells_ready) {
      RMM_TRY(gacc.alloc(sizeof(uint64_t) * cells_pad));
      RMM_TRY(grows.alloc(sizeof(unsigned int) * cells_pad));
      if (vbit) RMM_TRY(gvalid.alloc(sizeof(unsigned int) * cells_pad));
      GDF_LAUNCH("gb_fill", gb_fill_u64, dim3(stream_grid(cells_pad, 1024)), dim3(256), 0, stream0(), gacc.as<unsigned long long>(),
                 (unsigned long long)acc_identity_host(fold_op), (uint32_t)cells_pad);
      HIP_TRY(hipMemsetAsync(grows.p, 0, sizeof(unsigned int) * cells_pad, stream0()));
      if (vbit) HIP_TRY(hipMemsetAsync(gvalid.p, 0, sizeof(unsigned int) * cells_pad, stream0()));
    }
    const size_t plds = (((size_t)1 << id_bits) + GB_PART_TRASH) * (vbit ? 16 : 12) + 16;
    bool launched = units.empty();          // (every row in the hot window: the scatter kernel aggregated them all)
    if constexpr (sizeof(K) == 4) {
      if (fused && !launched) {
        launched = true;
        if (vbit) {
          HIP_TRY(hipFuncSetAttribute((const void *)gb_part_aggregate<true, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
          GDF_LAUNCH("gb_part_aggregate", (gb_part_aggregate<true, K, true>), dim3((unsigned)units.size()), dim3(GB_DENSE_THREADS), plds, stream0(),
                     (const K *)kin, (const uint64_t *)nullptr, (const GbPartUnit *)d_units.as<GbPartUnit>(), id_bits, fold_op, flt,
                     gacc.as<unsigned long long>(), grows.as<unsigned int>(), gvalid.as<unsigned int>(), aggregate_spec);
        } else {
          HIP_TRY(hipFuncSetAttribute((const void *)gb_part_aggregate<false, K, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
          GDF_LAUNCH("gb_part_aggregate", (gb_part_aggregate<false, K, true>), dim3((unsigned)units.size()), dim3(GB_DENSE_THREADS), plds, stream0(),
                     (const K *)kin, (const uint64_t *)nullptr, (const GbPartUnit *)d_units.as<GbPartUnit>(), id_bits, fold_op, flt,
                     gacc.as<unsigned long long>(), grows.as<unsigned int>(), (unsigned int *)nullptr, aggregate_spec);
        }
      }
    }
    if (launched) {
    } else if (vbit) {
      HIP_TRY(hipFuncSetAttribute((const void *)gb_part_aggregate<true, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
      GDF_LAUNCH("gb_part_aggregate", (gb_part_aggregate<true, K>), dim3((unsigned)units.size()), dim3(GB_DENSE_THREADS), plds, stream0(), (const K *)kin,
                 (const uint64_t *)pin, (const GbPartUnit *)d_units.as<GbPartUnit>(), id_bits, fold_op, flt, gacc.as<unsigned long long>(),
                 grows.as<unsigned int>(), gvalid.as<unsigned int>(), GbSpec{});
    } else {
      HIP_TRY(hipFuncSetAttribute((const void *)gb_part_aggregate<false, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)plds));
      GDF_LAUNCH("gb_part_aggregate", (gb_part_aggregate<false, K>), dim3((unsigned)units.size()), dim3(GB_DENSE_THREADS), plds, stream0(), (const K *)kin,
                 (const uint64_t *)pin, (const GbPartUnit *)d_units.as<GbPartUnit>(), id_bits, fold_op, flt, gacc.as<unsigned long long>(),
                 grows.as<unsigned int>(), (unsigned int *)nullptr, GbSpec{});
    }
    const unsigned nblocks = (unsigned)(cells_pad / 1024);
    GDF_LAUNCH("gb_part_count", gb_part_count, dim3(nblocks), dim3(1024), 0, stream0(), (const unsigned int *)grows.as<unsigned int>(), bcnt.as<uint32_t>());
    DevBuf scan_scratch;
    GDF_TRY(scan_u32_async(bcnt.as<uint32_t>(), bcnt.as<uint32_t>(), nblocks, false, &scan_scratch));       // (the read-back below waits for it)
    {
      int verdict = 0;
      GDF_TRY(late_verdict(&verdict));
      if (verdict) return late_exit(verdict);
    }
    // the number of groups is needed before the outputs can be written only for the optional ok-bytes
    if (want_ok) RMM_TRY(agg_ok.alloc(cells_pad < (size_t)nvalid ? cells_pad : (size_t)nvalid));
    o.agg_ok = agg_ok.as<uint8_t>();
    GDF_LAUNCH("gb_extract", gb_part_extract, dim3(nblocks), dim3(1024), 0, stream0(), t, sp, o, op, (const unsigned long long *)gacc.as<unsigned long long>(),
               (const unsigned int *)grows.as<unsigned int>(), (const unsigned int *)gvalid.as<unsigned int>(), (const uint32_t *)bcnt.as<uint32_t>(),
               ng.as<unsigned int>());
    HIP_CHECK_LAST();
    HIP_TRY(read_back(&ngroups, ng.p, sizeof(ngroups)));
  }
  {
    int verdict = 0;                    // (no valid row: nothing was queued behind the flags)
    GDF_TRY(late_verdict(&verdict));
    if (verdict) return late_exit(verdict);
  }
  for (int c = 0; c < ncols; ++c) out_keys[c]->size = (gdf_size_type)ngroups;
  out_agg->size = (gdf_size_type)ngroups;
  *done = true;
  return write_output_masks(ncols, out_keys, out_agg, o.agg_ok, ngroups);     // cells ascend: already sorted
}

// Path 3 -- sorted: packed keys, many groups.
static gdf_error gb_path_sorted(GbJob &j, bool *done) {
  *done = false;
  [[maybe_unused]] const int ncols = j.ncols;
  [[maybe_unused]] gdf_column **out_keys = j.out_keys;
  [[maybe_unused]] gdf_column *out_agg = j.out_agg;
  [[maybe_unused]] const int op = j.op;
  [[maybe_unused]] const bool sort_result = j.sort_result;
  [[maybe_unused]] const KeyTable &t = j.t;
  [[maybe_unused]] const int64_t n = j.t.nrows;
  [[maybe_unused]] const ElemKind in_kind = j.in_kind, out_kind = j.out_kind;
  [[maybe_unused]] const GbKeyPlan &plan = j.plan;
  [[maybe_unused]] const GbVal &val = j.val;
  [[maybe_unused]] const bool masked = j.masked, avg = j.counted, want_ok = j.want_ok;
  [[maybe_unused]] DevBuf &agg_ok = j.agg_ok;
  if (plan.packed && !lab::path_on("GDF_GB_NO_SORTED")) {
    GbKeyPlan sp = plan;
    if (!sp.ordered) GDF_TRY(gb_plan_range(t, &sp));          // fewer key bits = fewer radix passes, and sorted output for free
    const int vbit = (val.valid != nullptr && op != OP_COUNT) ? 1 : 0;
    const int null_bit = t.any_valid ? 1 : 0;
    // partitioned variant: sort the high key bits only, index LDS accumulators with the low ones
    {
      const int id_bits = sp.total_bits < GB_PART_ID_BITS ? sp.total_bits : GB_PART_ID_BITS;
      if (sp.ordered && sp.total_bits - id_bits <= GB_PART_MAX_BITS && !lab::path_on("GDF_GB_NO_PART")) {
        if (sp.total_bits + vbit + null_bit <= 32 && !lab::knob_on("GDF_GB_NO_K32")) return gb_sorted_partitioned<uint32_t>(j, sp, vbit, null_bit, done);
        return gb_sorted_partitioned<uint64_t>(j, sp, vbit, null_bit, done);
      }
    }
    if (sp.total_bits + vbit + null_bit <= 64) {
      const uint32_t nn = (uint32_t)n;
      DevBuf ka, kb, pa, pb, fl, gid, start, acc, cnt;
      RMM_TRY(ka.alloc(sizeof(uint64_t) * (size_t)nn));
      RMM_TRY(kb.alloc(sizeof(uint64_t) * (size_t)nn));
      RMM_TRY(pa.alloc(sizeof(uint64_t) * (size_t)nn));
      RMM_TRY(pb.alloc(sizeof(uint64_t) * (size_t)nn));
      RMM_TRY(fl.alloc(16));
      HIP_TRY(hipMemsetAsync(fl.p, 0, 16, stream0()));
      const int fold_op = op == OP_AVG ? OP_SUM : op;
      const bool flt = is_flt(val.kind);
      const uint64_t null_key = null_bit ? (1ULL << (sp.total_bits + vbit)) : 0ULL;
      uint64_t *kin = ka.as<uint64_t>(), *kout = kb.as<uint64_t>(), *pin = pa.as<uint64_t>(), *pout = pb.as<uint64_t>();
      GDF_LAUNCH("gb_sorted_make_pairs", gb_sorted_make_pairs<uint64_t>, dim3(stream_grid((size_t)n, 256 * 8)), dim3(256), 0, stream0(), t, sp, val,
                 fold_op, vbit, null_key, kin, pin, fl.as<unsigned long long>(), (unsigned int *)(fl.as<unsigned long long>() + 1));
      struct { unsigned long long varying; unsigned int dropped, pad; } hf;
      HIP_TRY(read_back(&hf, fl.p, 16));
      const uint32_t nvalid = nn - hf.dropped;
      uint32_t ngroups = 0;
      GbOut o{};
      o.ncols = ncols;
      for (int c = 0; c < ncols; ++c) o.key_out[c] = out_keys[c]->data;
      o.agg_out = out_agg->data;
      o.in_kind = (int)in_kind;
      o.agg_kind = (int)((op == OP_COUNT || op == OP_AVG) ? out_kind : in_kind);
      o.counted = val.valid != nullptr;
      GDF_TRY(radix_sort_pairs_u64(kin, kout, pin, pout, nn, hf.varying));
      if (nvalid) {
        RMM_TRY(gid.alloc(sizeof(uint32_t) * (size_t)nvalid));
        const int hgrid = stream_grid(nvalid, 256 * 4);
        GDF_LAUNCH("gb_sorted_heads", gb_sorted_heads, dim3(hgrid), dim3(256), 0, stream0(), (const uint64_t *)kin, vbit, gid.as<uint32_t>(), nvalid);
        GDF_TRY(scan_u32(gid.as<uint32_t>(), gid.as<uint32_t>(), nvalid, true));
        HIP_TRY(read_back(&ngroups, gid.as<uint32_t>() + (nvalid - 1), sizeof(uint32_t)));
        RMM_TRY(start.alloc(sizeof(uint32_t) * ((size_t)ngroups + 1)));
        RMM_TRY(acc.alloc(sizeof(uint64_t) * (size_t)ngroups));
        if (vbit) RMM_TRY(cnt.alloc(sizeof(uint64_t) * (size_t)ngroups));
        GDF_LAUNCH("gb_sorted_starts", gb_sorted_starts, dim3(hgrid), dim3(256), 0, stream0(), (const uint32_t *)gid.as<uint32_t>(), start.as<uint32_t>(),
                   nvalid, ngroups);
        GDF_LAUNCH("gb_fill", gb_fill_u64, dim3(stream_grid(ngroups, 1024)), dim3(256), 0, stream0(), acc.as<unsigned long long>(),
                   (unsigned long long)acc_identity_host(fold_op), ngroups);
        if (vbit) HIP_TRY(hipMemsetAsync(cnt.p, 0, sizeof(uint64_t) * (size_t)ngroups, stream0()));
        const uint32_t per_block = WAVE * GB_SEG_ROUNDS * 4;
        const dim3 rgrid((nvalid + per_block - 1) / per_block);
        if (vbit) GDF_LAUNCH("gb_sorted_reduce", gb_sorted_reduce<true>, rgrid, dim3(256), 0, stream0(), (const uint64_t *)kin, (const uint64_t *)pin,
                             (const uint32_t *)gid.as<uint32_t>(), fold_op, flt, acc.as<unsigned long long>(), cnt.as<unsigned long long>(), nvalid);
        else GDF_LAUNCH("gb_sorted_reduce", gb_sorted_reduce<false>, rgrid, dim3(256), 0, stream0(), (const uint64_t *)kin, (const uint64_t *)pin,
                        (const uint32_t *)gid.as<uint32_t>(), fold_op, flt, acc.as<unsigned long long>(), cnt.as<unsigned long long>(), nvalid);
        if (want_ok) RMM_TRY(agg_ok.alloc(ngroups));
        o.agg_ok = agg_ok.as<uint8_t>();
        GDF_LAUNCH("gb_extract", gb_sorted_extract, dim3(stream_grid(ngroups, 256)), dim3(256), 0, stream0(), t, sp, (const uint64_t *)kin, vbit,
                   (const uint32_t *)start.as<uint32_t>(), ngroups, o, op, (const unsigned long long *)acc.as<unsigned long long>(),
                   (const unsigned long long *)cnt.as<unsigned long long>());
        HIP_CHECK_LAST();
        HIP_TRY(hipStreamSynchronize(stream0()));
      }
      for (int c = 0; c < ncols; ++c) out_keys[c]->size = (gdf_size_type)ngroups;
      out_agg->size = (gdf_size_type)ngroups;
      if ((sort_result || op == OP_AVG) && !sp.ordered) {
        int kinds[MAX_KEY_COLS];
        for (int c = 0; c < ncols; ++c) kinds[c] = t.col[c].kind;
        GDF_TRY(sort_result_rows(ncols, out_keys, kinds, out_agg->data, kind_width((ElemKind)o.agg_kind), ngroups, o.agg_ok));
      }
      *done = true;
      return write_output_masks(ncols, out_keys, out_agg, o.agg_ok, ngroups);
    }
  }
  return GDF_SUCCESS;
}

// Path 4 -- hash table: LDS pre-aggregation in front of a global table (float keys, keys that do not pack).
static gdf_error gb_path_table(GbJob &j) {
  [[maybe_unused]] const int ncols = j.ncols;
  [[maybe_unused]] gdf_column **out_keys = j.out_keys;
  [[maybe_unused]] gdf_column *out_agg = j.out_agg;
  [[maybe_unused]] const int op = j.op;
  [[maybe_unused]] const bool sort_result = j.sort_result;
  [[maybe_unused]] const KeyTable &t = j.t;
  [[maybe_unused]] const int64_t n = j.t.nrows;
  [[maybe_unused]] const ElemKind in_kind = j.in_kind, out_kind = j.out_kind;
  [[maybe_unused]] const GbKeyPlan &plan = j.plan;
  [[maybe_unused]] const GbVal &val = j.val;
  [[maybe_unused]] const bool masked = j.masked, avg = j.counted, want_ok = j.want_ok;
  [[maybe_unused]] DevBuf &agg_ok = j.agg_ok;
  // The table is sized by GROUPS.  Start small (the common case) and grow x256 on
  // overflow, up to the 2*N slots the reference always allocates.
  uint64_t cap_max = 1;
  while (cap_max < 2 * (uint64_t)n) cap_max <<= 1;
  uint64_t T = cap_max < (1u << 18) ? cap_max : (1u << 18);
  // workgroup geometry: one contiguous chunk of rows per workgroup
  const int grid = stream_grid((size_t)n, GB_THREADS * 32, NUM_CU * 4);
  int64_t chunk = (n + grid - 1) / grid;
  const size_t lds = plan.packed ? (size_t)GB_LDS_SLOTS * 8 * (avg ? 3 : 2) + 16 : 0;
  for (;;) {
    DevBuf keys, first, acc, cnt, flags, out_count;
    if (plan.packed) RMM_TRY(keys.alloc(sizeof(uint64_t) * (T + 1))); else RMM_TRY(first.alloc(sizeof(int32_t) * (T + 1)));
    RMM_TRY(acc.alloc(sizeof(uint64_t) * (T + 1)));
    if (avg) RMM_TRY(cnt.alloc(sizeof(uint64_t) * (T + 1)));
    RMM_TRY(flags.alloc(sizeof(unsigned int) * 4));
    RMM_TRY(out_count.alloc(sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(flags.p, 0, sizeof(unsigned int) * 4, stream0()));
    HIP_TRY(hipMemsetAsync(out_count.p, 0, sizeof(unsigned long long), stream0()));
    GbTable g{};
    g.T = (uint32_t)T;
    g.keys = keys.as<unsigned long long>();
    g.first = first.as<int32_t>();
    g.acc = acc.as<unsigned long long>();
    g.cnt = cnt.as<unsigned long long>();
    g.occupied = flags.as<unsigned int>();
    g.overflow = flags.as<unsigned int>() + 1;
    g.special = flags.as<unsigned int>() + 2;
    g.limit = T >= cap_max ? 0xffffffffu : (uint32_t)(T / 2);   // at 2*N slots the table can never fill
    GDF_LAUNCH("gb_init_table", gb_init_table, dim3(stream_grid(T + 1, 256 * 8)), dim3(256), 0, stream0(), g, op == OP_AVG ? OP_SUM : op, plan.packed);
    if (plan.packed) {
      HIP_TRY(hipFuncSetAttribute((const void *)gb_aggregate<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      GDF_LAUNCH("gb_aggregate_packed", gb_aggregate<true>, dim3(grid), dim3(GB_THREADS), lds, stream0(), t, plan, val, op, g, chunk);
    } else {
      GDF_LAUNCH("gb_aggregate_rows", gb_aggregate<false>, dim3(grid), dim3(GB_THREADS), 0, stream0(), t, plan, val, op, g, chunk);
    }
    HIP_CHECK_LAST();
    unsigned int h_flags[3] = {0, 0, 0};
    HIP_TRY(read_back(h_flags, flags.p, sizeof(h_flags)));
    if (h_flags[1]) {                       // too many groups for this table
      if (T >= cap_max) return GDF_HASH_TABLE_INSERT_FAILURE;
      T = T * 256 < cap_max ? T * 256 : cap_max;     // 2^18 -> 2^26 -> 2N: at most two retries
      continue;
    }
    const unsigned int special_used = h_flags[2];   // slot T (reserved key) is live iff some row used it
    GbOut o{};
    o.ncols = ncols;
    for (int c = 0; c < ncols; ++c) o.key_out[c] = out_keys[c]->data;
    o.agg_out = out_agg->data;
    o.in_kind = (int)in_kind;
    o.agg_kind = (int)((op == OP_COUNT || op == OP_AVG) ? out_kind : in_kind);
    if (want_ok) RMM_TRY(agg_ok.alloc((size_t)h_flags[0] + 2));
    o.agg_ok = agg_ok.as<uint8_t>();
    o.counted = val.valid != nullptr;
    const int egrid = stream_grid(T + 1, 256 * 4);
    if (plan.packed)
      GDF_LAUNCH("gb_extract", gb_extract<true>, dim3(egrid), dim3(256), 0, stream0(), t, plan, g, o, op, special_used, out_count.as<unsigned long long>());
    else
      GDF_LAUNCH("gb_extract", gb_extract<false>, dim3(egrid), dim3(256), 0, stream0(), t, plan, g, o, op, special_used, out_count.as<unsigned long long>());
    HIP_CHECK_LAST();
    unsigned long long ngroups = 0;
    HIP_TRY(read_back(&ngroups, out_count.p, sizeof(ngroups)));
    for (int c = 0; c < ncols; ++c) out_keys[c]->size = (gdf_size_type)ngroups;   // gdf_table.cuh:334-342
    out_agg->size = (gdf_size_type)ngroups;
    if (sort_result || op == OP_AVG) {
      int kinds[MAX_KEY_COLS];
      for (int c = 0; c < ncols; ++c) kinds[c] = t.col[c].kind;
      GDF_TRY(sort_result_rows(ncols, out_keys, kinds, out_agg->data, kind_width((ElemKind)o.agg_kind), (uint32_t)ngroups, o.agg_ok));
    }
    return write_output_masks(ncols, out_keys, out_agg, o.agg_ok, (uint32_t)ngroups);
  }
}

// ---------------------------------------------------------------------------
// host driver
// ---------------------------------------------------------------------------
// sort_indices / direct_done: a SORT-method call (group_by_single) that only wants the direct path -- *direct_done says whether it ran
static gdf_error group_by_hash(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column **out_keys,
                               gdf_column *out_agg, int op, bool sort_result, size_t *sort_indices = nullptr, bool *direct_done = nullptr) {
  // groupby.cuh:218-238
  if (0 == ncols || nullptr == cols || nullptr == col_agg) return GDF_DATASET_EMPTY;
  if (nullptr == out_keys || nullptr == out_agg) return GDF_DATASET_EMPTY;
  if (0 == cols[0]->size || 0 == col_agg->size) return GDF_SUCCESS;
  KeyTable t;
  GDF_TRY(make_key_table(cols, ncols, &t));
  const int64_t n = t.nrows;
  if (n >= (int64_t)INT_MAX) return GDF_COLUMN_SIZE_TOO_BIG;
  // every kernel below reads rows [0, n) of EVERY key column and (COUNT aside) of the aggregation column: unequal sizes are an error
  // here, not an out-of-bounds read (the reference's gdf_table asserts equal column sizes, gdf_table.cuh:249-322; the SORT method
  // answers GDF_COLUMN_SIZE_MISMATCH, sort.hip group_by_sort, and so does its direct-path shortcut, which enters through here)
  for (int c = 0; c < ncols; ++c) GDF_REQUIRE(cols[c] && cols[c]->size == cols[0]->size, GDF_COLUMN_SIZE_MISMATCH);
  if (op != OP_COUNT) GDF_REQUIRE(col_agg->size == cols[0]->size, GDF_COLUMN_SIZE_MISMATCH);

  // dtype dispatch (groupby.cuh:86-190): COUNT is typed by the OUTPUT column, the rest by the input
  const ElemKind in_kind = elem_kind(col_agg->dtype);
  const ElemKind out_kind = elem_kind(out_agg->dtype);
  if (op == OP_COUNT) { if (out_kind == K_BAD) return GDF_UNSUPPORTED_DTYPE; }
  else if (in_kind == K_BAD) return GDF_UNSUPPORTED_DTYPE;
  if (op == OP_AVG && (out_kind == K_BAD || out_agg->dtype == GDF_DATE32 || out_agg->dtype == GDF_DATE64 ||
                       out_agg->dtype == GDF_TIMESTAMP || col_agg->dtype == GDF_DATE32 ||
                       col_agg->dtype == GDF_DATE64 || col_agg->dtype == GDF_TIMESTAMP))
    return GDF_UNSUPPORTED_DTYPE;   // groupby.cuh:376-385,409-418 list only the six numeric types
  for (int c = 0; c < ncols; ++c)
    if (!out_keys[c] || !out_keys[c]->data) return GDF_DATASET_EMPTY;
  if (!out_agg->data) return GDF_DATASET_EMPTY;

  GbJob j{};
  j.sort_indices = sort_indices;
  j.ncols = ncols;
  j.out_keys = out_keys;
  j.out_agg = out_agg;
  j.op = op;
  j.sort_result = sort_result;
  j.t = t;
  j.in_kind = in_kind;
  j.out_kind = out_kind;
  j.plan = gb_plan_keys(t);
  if (direct_done) {                 // the SORT method's fast route: the direct path or nothing
    j.sort_method = true;
    j.val = GbVal{col_agg->data, (int)(op == OP_COUNT ? K_I8 : in_kind), nullptr};
    j.masked = false;
    j.counted = op == OP_AVG;
    j.want_ok = false;
    return gb_path_direct(j, direct_done);
  }
  GbKeyPlan guess_plan{};
  bool guess_plan_ok = false;
  if (!j.plan.packed && n >= ((int64_t)1 << 24) && !lab::knob_on("GDF_GB_NO_GUESS_RANGES")) {
    GDF_TRY(gb_plan_range_sampled(t, &guess_plan, &guess_plan_ok));
    const int vb = (col_agg->valid != nullptr && op != OP_COUNT) ? 1 : 0, nb = t.any_valid ? 1 : 0;
    guess_plan_ok = guess_plan_ok && guess_plan.total_bits >= 18 && guess_plan.total_bits - GB_PART_ID_BITS <= GBP_MAX_PART_BITS &&
                    guess_plan.total_bits + vb + nb <= 32;
  }
  if (!j.plan.packed && !guess_plan_ok) GDF_TRY(gb_plan_range(t, &j.plan, &j.ranges));
  j.val = GbVal{col_agg->data, (int)(op == OP_COUNT ? K_I8 : in_kind), (const uint8_t *)col_agg->valid};
  // Validity masks (beyond the reference, which rejects them: sqls_ops.cu:1103-1106; semantics of
  // SURVEY.md 8d C5 = pandas dropna): a row with a null in any key column is dropped; a null value is
  // skipped, so SUM/MIN/MAX/AVG/COUNT run over the valid values of each group; a group without any
  // valid value is reported with value 0 and, if the caller gave out_col_agg a mask, a cleared bit.
  j.masked = t.any_valid || j.val.valid != nullptr;
  j.counted = op == OP_AVG || (j.val.valid != nullptr && op != OP_COUNT);
  j.want_ok = j.val.valid != nullptr && op != OP_COUNT && out_agg->valid != nullptr;

  bool done = false;
  // Many rows, keys that only pack by range: try the partitioned path on ranges GUESSED from a prefix before paying for the
  // exact min / max pass.  Taken only when the guessed layout says "more ids than LDS accumulators hold" (fewer: the direct /
  // dictionary paths are the better ones and want exact ranges); its count kernel checks every key and reports a violation.
  if (guess_plan_ok) {
    const int vbit = (j.val.valid != nullptr && op != OP_COUNT) ? 1 : 0;
    const int null_bit = t.any_valid ? 1 : 0;
    const GbKeyPlan exact_later = j.plan;
    j.plan = guess_plan;
    j.range_violated = false;
    GDF_TRY(gb_sorted_partitioned<uint32_t>(j, guess_plan, vbit, null_bit, &done, true));
    if (done) return GDF_SUCCESS;
    j.plan = exact_later;
    GDF_TRY(gb_plan_range(t, &j.plan, &j.ranges));              // the guess did not hold (or did not qualify): exact ranges after all
  }
  // the four paths, cheapest first; each declines (done == false) what it cannot take
  GDF_TRY(gb_path_direct(j, &done));
  if (done) return GDF_SUCCESS;
  if (j.plan.packed) {
    GDF_TRY(gb_path_dense(j, &done));
    if (done) return GDF_SUCCESS;
    GDF_TRY(gb_path_sorted(j, &done));
    if (done) return GDF_SUCCESS;
  }
  return gb_path_table(j);
}

// sqls_ops.cu:1085-1363 gdf_group_by_single
static gdf_error group_by_single(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                                 gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt, int op) {
  if (0 == ncols || nullptr == cols || nullptr == col_agg || nullptr == out_col_agg || nullptr == ctxt)
    return GDF_DATASET_EMPTY;
  // The reference rejects every mask here (sqls_ops.cu:1103-1106).  The HASH method accepts them with the
  // semantics documented in group_by_hash (BASELINE config C5); the SORT method keeps the reference's answer.
  if (ctxt->flag_method != GDF_HASH) {
    for (int i = 0; i < ncols; ++i) GDF_REQUIRE(!cols[i]->valid, GDF_VALIDITY_UNSUPPORTED);
    GDF_REQUIRE(!col_agg->valid, GDF_VALIDITY_UNSUPPORTED);
  }
  if (0 == cols[0]->size || 0 == col_agg->size) {
    out_col_agg->size = 0;
    if (out_col_indices) out_col_indices->size = 0;
    if (out_col_values)
      for (int c = 0; c < ncols; ++c)
        if (out_col_values[c]) out_col_values[c]->size = 0;
    return GDF_SUCCESS;
  }
  if (ctxt->flag_method != GDF_HASH && ctxt->flag_method != GDF_SORT) return GDF_UNSUPPORTED_METHOD;
  gdf_nvtx_range_push("LIBGDF_GROUPBY", GDF_ORANGE);   // sqls_ops.cu:1132
  struct Pop { ~Pop() { gdf_nvtx_range_pop(); } } pop;
  if (ctxt->flag_method == GDF_SORT) {                  // sort.hip; sqls_ops.cu:1134-1289
    // Integer keys with a small value range (C2's shape) need no sort to come out sorted: the direct path numbers the groups in
    // lexicographic key order and aggregates in LDS -- the SORT method's contract (ascending groups, aggregation in the input
    // dtype, COUNT in the output column's, out_col_indices = every group's last row) is met by it at a tenth of the cost of
    // sort + segmented reduce (4.8 -> 0.7 ms per 1e8 rows, 1e4 groups).  Everything else -- wide or float keys, COUNT_DISTINCT,
    // presorted input, an AVG whose output column is typed differently from its input -- takes the sort.
    bool all_out = out_col_values != nullptr;
    for (int c = 0; c < ncols && all_out; ++c) all_out = out_col_values[c] && out_col_values[c]->data;
    // (the sort's own dtype rules stay in force: no date / timestamp aggregation columns there, sqls_ops.cu:411-1083)
    const bool plain_dtypes = col_agg->dtype <= GDF_FLOAT64 && out_col_agg->dtype <= GDF_FLOAT64 &&
                              (op == OP_COUNT || out_col_agg->dtype == col_agg->dtype || op != OP_AVG);
    const bool avg_typed = plain_dtypes && (op != OP_AVG || out_col_agg->dtype == col_agg->dtype);
    if (all_out && avg_typed && op != OP_COUNT_DISTINCT && !ctxt->flag_sorted && out_col_agg->data && col_agg->size == cols[0]->size &&
        (!out_col_indices || out_col_indices->data) && !lab::path_on("GDF_SORT_NO_DIRECT")) {
      bool done = false;
      GDF_TRY(group_by_hash(ncols, cols, col_agg, out_col_values, out_col_agg, op, true,
                            out_col_indices ? (size_t *)out_col_indices->data : nullptr, &done));
      if (done) {
        if (out_col_indices) out_col_indices->size = out_col_agg->size;
        return GDF_SUCCESS;
      }
    }
    return group_by_sort(ncols, cols, col_agg, out_col_indices, out_col_values, out_col_agg, ctxt, op);
  }
  if (op == OP_COUNT_DISTINCT) return GDF_UNSUPPORTED_METHOD;   // hash branch's default case, sqls_ops.cu:1347-1349
  return group_by_hash(ncols, cols, col_agg, out_col_values, out_col_agg, op, ctxt->flag_sort_result == 1);
}

}  // namespace gdf_amd

using namespace gdf_amd;

extern "C" {

gdf_error gdf_group_by_sum(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                           gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return group_by_single(ncols, cols, col_agg, out_col_indices, out_col_values, out_col_agg, ctxt, OP_SUM);
  });
}
gdf_error gdf_group_by_min(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                           gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return group_by_single(ncols, cols, col_agg, out_col_indices, out_col_values, out_col_agg, ctxt, OP_MIN);
  });
}
gdf_error gdf_group_by_max(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                           gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return group_by_single(ncols, cols, col_agg, out_col_indices, out_col_values, out_col_agg, ctxt, OP_MAX);
  });
}
gdf_error gdf_group_by_avg(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                           gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt) {
  return gdf_amd::guarded([&]() -> gdf_error {
  return group_by_single(ncols, cols, col_agg, out_col_indices, out_col_values, out_col_agg, ctxt, OP_AVG);
  });
}
gdf_error gdf_group_by_count(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                             gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt) {
  return gdf_amd::guarded([&]() -> gdf_error {
  if (nullptr == ctxt) return GDF_DATASET_EMPTY;
  // flag_distinct selects COUNT_DISTINCT (sqls_ops.cu:1483-1486), which only the SORT method implements
  return group_by_single(ncols, cols, col_agg, out_col_indices, out_col_values, out_col_agg, ctxt,
                         ctxt->flag_distinct ? OP_COUNT_DISTINCT : OP_COUNT);
  });
}

}  // extern "C"
