/*
 * gdf_oracle.c -- CPU restatement of the reference's relational hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under libgdf_amd/ links, imports or calls this
 * file; it is used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * as the CHECKER for the HIP library (and, in bench.py, as the timed CPU baseline,
 * kind "port").  Plain C, single-threaded, written from the reference's semantics;
 * every function cites the reference lines it restates (paths relative to
 * /root/reference/libgdf).
 *
 * Pinning: orc_murmur3_32 / orc_hash_combine / orc_identity_hash are checked against
 * (a) the golden values captured from the reference's own header
 * (tests/golden/murmur3_32.json) and (b) oracle/_ref/libref_hash.so --
 * the reference's src/hashmap/hash_functions.cuh compiled for the host -- whenever
 * /root/reference is present (tests/test_oracle_pinning.py).  Group-by and filter
 * semantics are checked against the known-answer vectors of the reference's sqls
 * tests (tests/golden/sqls_known_answers.json).  Join / partition results have no
 * golden vectors in the reference (its tests compare against an in-test CPU
 * solution, tests/join/join-tests.cu:260-356, tests/hashing/hash-partition-test.cu:
 * 166-246); the join and partition oracles restate exactly those CPU solutions.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* element kinds: the storage class behind a gdf_dtype (gdf_table.cuh:704-854 maps
 * DATE32->int32, DATE64/TIMESTAMP->int64) */
enum { K_I8 = 0, K_I16, K_I32, K_I64, K_F32, K_F64 };

static int kind_width(int k) {
  switch (k) { case K_I8: return 1; case K_I16: return 2; case K_I32: case K_F32: return 4; default: return 8; }
}

/* ---------------------------------------------------------------------------
 * hashing: src/hashmap/hash_functions.cuh:30-164
 * ------------------------------------------------------------------------- */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

/* MurmurHash3_x86_32, seed 0, over `len` raw bytes (hash_functions.cuh:80-118) */
uint32_t orc_murmur3_32(const uint8_t *data, int len) {
  const int nblocks = len / 4;
  uint32_t h1 = 0;
  const uint32_t c1 = 0xcc9e2d51u, c2 = 0x1b873593u;
  for (int i = 0; i < nblocks; ++i) {
    uint32_t k1;
    memcpy(&k1, data + 4 * i, 4);
    k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2;
    h1 ^= k1; h1 = rotl32(h1, 13); h1 = h1 * 5 + 0xe6546b64u;
  }
  const uint8_t *tail = data + nblocks * 4;
  uint32_t k1 = 0;
  switch (len & 3) {
    case 3: k1 ^= (uint32_t)tail[2] << 16; /* fallthrough */
    case 2: k1 ^= (uint32_t)tail[1] << 8;  /* fallthrough */
    case 1: k1 ^= tail[0];
            k1 *= c1; k1 = rotl32(k1, 15); k1 *= c2; h1 ^= k1;
  }
  h1 ^= (uint32_t)len;
  h1 ^= h1 >> 16; h1 *= 0x85ebca6bu; h1 ^= h1 >> 13; h1 *= 0xc2b2ae35u; h1 ^= h1 >> 16;   /* fmix32 :48-56 */
  return h1;
}

/* hash_functions.cuh:71-78 */
uint32_t orc_hash_combine(uint32_t lhs, uint32_t rhs) { return lhs ^ (rhs + 0x9e3779b9u + (lhs << 6) + (lhs >> 2)); }

/* IdentityHash: static_cast<uint32_t>(key) (hash_functions.cuh:157-163).  For float
 * kinds the conversion of a negative / out-of-range / NaN value is undefined in C++;
 * GPUs saturate (negative, NaN -> 0; too large -> UINT32_MAX), which is what we state. */
static uint32_t sat_u32(double d) {
  if (!(d > 0.0)) return 0;
  if (d >= 4294967295.0) return 4294967295u;
  return (uint32_t)d;
}
uint32_t orc_identity_hash(const void *p, int kind) {
  switch (kind) {
    case K_I8: return (uint32_t)*(const int8_t *)p;
    case K_I16: return (uint32_t)*(const int16_t *)p;
    case K_I32: return (uint32_t)*(const int32_t *)p;
    case K_I64: return (uint32_t)*(const int64_t *)p;
    case K_F32: return sat_u32(*(const float *)p);
    default: return sat_u32(*(const double *)p);
  }
}

/* gdf_table::hash_row (gdf_table.cuh:704-854): fold column hashes left to right, the
 * first column is not combined */
static uint32_t hash_row(int ncols, const void *const *data, const int *kinds, int64_t i, int identity) {
  uint32_t h = 0;
  for (int c = 0; c < ncols; ++c) {
    const int w = kind_width(kinds[c]);
    const uint8_t *p = (const uint8_t *)data[c] + (size_t)i * w;
    const uint32_t k = identity ? orc_identity_hash(p, kinds[c]) : orc_murmur3_32(p, w);
    h = c == 0 ? k : orc_hash_combine(h, k);
  }
  return h;
}

/* gdf_hash (hashing.cu:83-154): ignores valid masks */
void orc_hash_rows(int ncols, const void *const *data, const int *kinds, int64_t n, int identity, uint32_t *out) {
  for (int64_t i = 0; i < n; ++i) out[i] = hash_row(ncols, data, kinds, i, identity);
}

/* partitioners of hashing.cu:193-237: & (P-1) for powers of two, % otherwise (:434-468) */
void orc_partition_ids(int ncols, const void *const *data, const int *kinds, int64_t n, int identity, uint32_t nparts,
                       uint32_t *out) {
  const int pow2 = (nparts & (nparts - 1)) == 0;
  for (int64_t i = 0; i < n; ++i) {
    const uint32_t h = hash_row(ncols, data, kinds, i, identity);
    out[i] = pow2 ? (h & (nparts - 1)) : (h % nparts);
  }
}

/* ---------------------------------------------------------------------------
 * rows
 * ------------------------------------------------------------------------- */
static int bit_valid(const uint8_t *mask, int64_t i) { return mask ? (mask[i >> 3] >> (i & 7)) & 1 : 1; }   /* utils.h:9-16 */

/* row validity = AND of the column masks (gdf_table.cuh:62-98) */
static int row_valid(int ncols, const uint8_t *const *valid, int64_t i) {
  for (int c = 0; c < ncols; ++c)
    if (!bit_valid(valid ? valid[c] : NULL, i)) return 0;
  return 1;
}

/* typed == per column (gdf_table.cuh:580-691); floats: NaN != NaN, -0.0 == +0.0 */
static int elem_equal(int kind, const void *a, int64_t i, const void *b, int64_t j) {
  switch (kind) {
    case K_I8: return ((const int8_t *)a)[i] == ((const int8_t *)b)[j];
    case K_I16: return ((const int16_t *)a)[i] == ((const int16_t *)b)[j];
    case K_I32: return ((const int32_t *)a)[i] == ((const int32_t *)b)[j];
    case K_I64: return ((const int64_t *)a)[i] == ((const int64_t *)b)[j];
    case K_F32: return ((const float *)a)[i] == ((const float *)b)[j];
    default: return ((const double *)a)[i] == ((const double *)b)[j];
  }
}
static int rows_equal(int ncols, const int *kinds, const void *const *a, int64_t i, const void *const *b, int64_t j) {
  for (int c = 0; c < ncols; ++c)
    if (!elem_equal(kinds[c], a[c], i, b[c], j)) return 0;
  return 1;
}
static int elem_less(int kind, const void *a, int64_t i, int64_t j) {   /* LesserRTTI::less, sqls_rtti_comp.hpp:99-140 */
  switch (kind) {
    case K_I8: return ((const int8_t *)a)[i] < ((const int8_t *)a)[j];
    case K_I16: return ((const int16_t *)a)[i] < ((const int16_t *)a)[j];
    case K_I32: return ((const int32_t *)a)[i] < ((const int32_t *)a)[j];
    case K_I64: return ((const int64_t *)a)[i] < ((const int64_t *)a)[j];
    case K_F32: return ((const float *)a)[i] < ((const float *)a)[j];
    default: return ((const double *)a)[i] < ((const double *)a)[j];
  }
}

/* a row hash that respects == (used only to bucket rows inside the oracle) */
static uint64_t eq_hash(int ncols, const int *kinds, const void *const *data, int64_t i) {
  uint64_t h = 1469598103934665603ULL;
  for (int c = 0; c < ncols; ++c) {
    uint64_t b = 0;
    const int w = kind_width(kinds[c]);
    memcpy(&b, (const uint8_t *)data[c] + (size_t)i * w, (size_t)w);
    if (kinds[c] == K_F32 && (uint32_t)(b << 1) == 0) b = 0;
    if (kinds[c] == K_F64 && (b << 1) == 0) b = 0;
    h = (h ^ b) * 1099511628211ULL;
    h ^= h >> 29;
  }
  return h;
}

/* ---------------------------------------------------------------------------
 * join: the multimap CPU solution of tests/join/join-tests.cu:260-356 with the kernel
 * rules of join_kernels.cuh:46-78,259-455 -- build on the right relation, rows with a
 * null key never match, LEFT adds (l,-1), FULL adds (-1,r) (join_compute_api.h:54-186).
 * Output pairs are sorted by (l, r): order is not part of the contract (:342-345).
 * ------------------------------------------------------------------------- */
typedef struct { int32_t l, r; } pair_t;
static int pair_cmp(const void *a, const void *b) {
  const pair_t *x = a, *y = b;
  if (x->l != y->l) return x->l < y->l ? -1 : 1;
  if (x->r != y->r) return x->r < y->r ? -1 : 1;
  return 0;
}

/* kind: 0 inner, 1 left, 2 full.  Returns the number of pairs; *out_l / *out_r are
 * malloc'ed (free with orc_free). */
int64_t orc_join(int join_kind, int ncols, const int *kinds, const void *const *ldata, const uint8_t *const *lvalid,
                 int64_t nl, const void *const *rdata, const uint8_t *const *rvalid, int64_t nr, int32_t **out_l,
                 int32_t **out_r) {
  /* chained hash table over the valid right rows */
  int64_t nb = 1;
  while (nb < 2 * nr) nb <<= 1;
  int64_t *head = malloc(sizeof(int64_t) * (size_t)nb), *next = malloc(sizeof(int64_t) * (size_t)(nr ? nr : 1));
  uint8_t *matched = calloc((size_t)(nr ? nr : 1), 1);
  for (int64_t b = 0; b < nb; ++b) head[b] = -1;
  for (int64_t j = nr - 1; j >= 0; --j) {
    if (!row_valid(ncols, rvalid, j)) continue;
    const int64_t b = (int64_t)(eq_hash(ncols, kinds, rdata, j) & (uint64_t)(nb - 1));
    next[j] = head[b];
    head[b] = j;
  }
  size_t cap = (size_t)(nl + nr + 16), n = 0;
  pair_t *out = malloc(sizeof(pair_t) * cap);
#define PUSH(L, R) do { if (n == cap) { cap *= 2; out = realloc(out, sizeof(pair_t) * cap); } out[n].l = (L); out[n].r = (R); ++n; } while (0)
  for (int64_t i = 0; i < nl; ++i) {
    int found = 0;
    if (row_valid(ncols, lvalid, i)) {
      const int64_t b = (int64_t)(eq_hash(ncols, kinds, ldata, i) & (uint64_t)(nb - 1));
      for (int64_t j = head[b]; j >= 0; j = next[j])
        if (rows_equal(ncols, kinds, ldata, i, rdata, j)) { PUSH((int32_t)i, (int32_t)j); matched[j] = 1; found = 1; }
    }
    if (!found && join_kind != 0) PUSH((int32_t)i, -1);
  }
  if (join_kind == 2)
    for (int64_t j = 0; j < nr; ++j)
      if (!matched[j]) PUSH(-1, (int32_t)j);
#undef PUSH
  qsort(out, n, sizeof(pair_t), pair_cmp);
  *out_l = malloc(sizeof(int32_t) * (n ? n : 1));
  *out_r = malloc(sizeof(int32_t) * (n ? n : 1));
  for (size_t k = 0; k < n; ++k) { (*out_l)[k] = out[k].l; (*out_r)[k] = out[k].r; }
  free(out); free(head); free(next); free(matched);
  return (int64_t)n;
}

void orc_free(void *p) { free(p); }

/* ---------------------------------------------------------------------------
 * group-by: map<key tuple -> aggregate> (tests/groupby/groupby-test.cu:227-259) with the
 * functors of aggregation_operations.cuh:30-86 -- the aggregate lives in the INPUT
 * dtype (integer sums wrap), COUNT lives in the OUTPUT dtype (groupby.cuh:102-109),
 * AVG = sum / static_cast<avg_type>(count) with count a size_t (groupby.cuh:308-386).
 * Output rows are sorted lexicographically by key (what flag_sort_result / AVG give).
 * ------------------------------------------------------------------------- */
enum { OP_SUM = 0, OP_MIN, OP_MAX, OP_AVG, OP_COUNT };

typedef union { int8_t i8; int16_t i16; int32_t i32; int64_t i64; float f32; double f64; } cell_t;

static void cell_load(cell_t *c, int kind, const void *data, int64_t i) {
  switch (kind) {
    case K_I8: c->i8 = ((const int8_t *)data)[i]; break;
    case K_I16: c->i16 = ((const int16_t *)data)[i]; break;
    case K_I32: c->i32 = ((const int32_t *)data)[i]; break;
    case K_I64: c->i64 = ((const int64_t *)data)[i]; break;
    case K_F32: c->f32 = ((const float *)data)[i]; break;
    default: c->f64 = ((const double *)data)[i]; break;
  }
}
static void cell_store(const cell_t *c, int kind, void *data, int64_t i) {
  switch (kind) {
    case K_I8: ((int8_t *)data)[i] = c->i8; break;
    case K_I16: ((int16_t *)data)[i] = c->i16; break;
    case K_I32: ((int32_t *)data)[i] = c->i32; break;
    case K_I64: ((int64_t *)data)[i] = c->i64; break;
    case K_F32: ((float *)data)[i] = c->f32; break;
    default: ((double *)data)[i] = c->f64; break;
  }
}
/* acc = op(acc, v) in the type `kind`; integer + wraps like the GPU's two's complement add */
static void cell_fold(cell_t *acc, const cell_t *v, int kind, int op) {
#define FOLD(F, U)                                                                   \
  switch (op) {                                                                      \
    case OP_SUM: acc->F = (__typeof__(acc->F))((U)acc->F + (U)v->F); break;          \
    case OP_MIN: if (v->F < acc->F) acc->F = v->F; break;                            \
    case OP_MAX: if (v->F > acc->F) acc->F = v->F; break;                            \
  }
  switch (kind) {
    case K_I8: FOLD(i8, uint8_t) break;
    case K_I16: FOLD(i16, uint16_t) break;
    case K_I32: FOLD(i32, uint32_t) break;
    case K_I64: FOLD(i64, uint64_t) break;
    case K_F32: FOLD(f32, float) break;
    default: FOLD(f64, double) break;
  }
#undef FOLD
}
/* count_op in the OUTPUT dtype: ++ in that type */
static void cell_count(cell_t *acc, int kind) {
  switch (kind) {
    case K_I8: acc->i8 = (int8_t)((uint8_t)acc->i8 + 1); break;
    case K_I16: acc->i16 = (int16_t)((uint16_t)acc->i16 + 1); break;
    case K_I32: acc->i32 = (int32_t)((uint32_t)acc->i32 + 1); break;
    case K_I64: acc->i64 = (int64_t)((uint64_t)acc->i64 + 1); break;
    case K_F32: acc->f32 += 1.0f; break;
    default: acc->f64 += 1.0; break;
  }
}
/* avg[i] = (avg_type)(sum / (avg_type)count), compute_average (groupby.cuh:308-328) */
static void cell_avg(cell_t *out, int avg_kind, const cell_t *sum, int sum_kind, size_t count) {
#define AVG_INNER(SF, AF, AT) { AT c = (AT)count; if (c == 0) out->AF = 0; else out->AF = (AT)(sum->SF / c); } break;
#define AVG_OUTER(SF)                           \
  switch (avg_kind) {                           \
    case K_I8: AVG_INNER(SF, i8, int8_t)        \
    case K_I16: AVG_INNER(SF, i16, int16_t)     \
    case K_I32: AVG_INNER(SF, i32, int32_t)     \
    case K_I64: AVG_INNER(SF, i64, int64_t)     \
    case K_F32: AVG_INNER(SF, f32, float)       \
    default: AVG_INNER(SF, f64, double)         \
  }
  switch (sum_kind) {
    case K_I8: AVG_OUTER(i8) break;
    case K_I16: AVG_OUTER(i16) break;
    case K_I32: AVG_OUTER(i32) break;
    case K_I64: AVG_OUTER(i64) break;
    case K_F32: AVG_OUTER(f32) break;
    default: AVG_OUTER(f64) break;
  }
#undef AVG_OUTER
#undef AVG_INNER
}

typedef struct { int ncols; const int *kinds; const void *const *data; } sort_ctx_t;
static sort_ctx_t g_sort;
static int first_cmp(const void *a, const void *b) {
  const int64_t i = *(const int64_t *)a, j = *(const int64_t *)b;
  for (int c = 0; c < g_sort.ncols; ++c) {
    if (elem_less(g_sort.kinds[c], g_sort.data[c], i, j)) return -1;
    if (elem_less(g_sort.kinds[c], g_sort.data[c], j, i)) return 1;
  }
  return i < j ? -1 : (i > j ? 1 : 0);
}

/* returns the number of groups; out_keys[c] / out_agg must hold n rows */
int64_t orc_group_by(int op, int ncols, const int *kinds, const void *const *keys, int64_t n, const void *vals,
                     int val_kind, int out_kind, void *const *out_keys, void *out_agg) {
  if (n == 0) return 0;
  int64_t nb = 1;
  while (nb < 2 * n) nb <<= 1;
  int64_t *slot_first = malloc(sizeof(int64_t) * (size_t)nb);   /* first row of the group in this slot */
  int64_t *slot_group = malloc(sizeof(int64_t) * (size_t)nb);
  for (int64_t b = 0; b < nb; ++b) slot_first[b] = -1;
  int64_t *first = malloc(sizeof(int64_t) * (size_t)n);
  cell_t *acc = calloc((size_t)n, sizeof(cell_t));
  size_t *cnt = calloc((size_t)n, sizeof(size_t));
  int64_t ngroups = 0;
  const int acc_kind = op == OP_COUNT ? out_kind : val_kind;
  for (int64_t i = 0; i < n; ++i) {
    int64_t b = (int64_t)(eq_hash(ncols, kinds, keys, i) & (uint64_t)(nb - 1));
    int64_t g = -1;
    /* concurrent_unordered_map::insert (:481-544): walk until an empty slot or an equal row */
    for (;;) {
      if (slot_first[b] < 0) { slot_first[b] = i; slot_group[b] = g = ngroups; first[ngroups++] = i; break; }
      if (rows_equal(ncols, kinds, keys, i, keys, slot_first[b])) { g = slot_group[b]; break; }
      b = (b + 1) & (nb - 1);
    }
    cell_t v;
    memset(&v, 0, sizeof v);
    if (op != OP_COUNT) cell_load(&v, val_kind, vals, i);
    if (cnt[g] == 0 && op != OP_COUNT) acc[g] = v;            /* identity then fold == first value */
    else if (op == OP_COUNT) cell_count(&acc[g], acc_kind);
    else cell_fold(&acc[g], &v, acc_kind, op == OP_AVG ? OP_SUM : op);
    if (op == OP_COUNT && cnt[g] == 0) { /* first row already counted above */ }
    cnt[g]++;
  }
  /* order groups by key */
  g_sort.ncols = ncols; g_sort.kinds = kinds; g_sort.data = keys;
  int64_t *order = malloc(sizeof(int64_t) * (size_t)ngroups);
  /* sort the first-row indices, then map back to group ids through a lookup */
  int64_t *group_of_first = malloc(sizeof(int64_t) * (size_t)n);
  for (int64_t g = 0; g < ngroups; ++g) { order[g] = first[g]; group_of_first[first[g]] = g; }
  qsort(order, (size_t)ngroups, sizeof(int64_t), first_cmp);
  for (int64_t k = 0; k < ngroups; ++k) {
    const int64_t row = order[k], g = group_of_first[row];
    for (int c = 0; c < ncols; ++c) {
      const int w = kind_width(kinds[c]);
      memcpy((uint8_t *)out_keys[c] + (size_t)k * w, (const uint8_t *)keys[c] + (size_t)row * w, (size_t)w);
    }
    if (op == OP_AVG) {
      cell_t r;
      memset(&r, 0, sizeof r);
      cell_avg(&r, out_kind, &acc[g], val_kind, cnt[g]);
      cell_store(&r, out_kind, out_agg, k);
    } else {
      cell_store(&acc[g], acc_kind, out_agg, k);
    }
  }
  free(slot_first); free(slot_group); free(first); free(acc); free(cnt); free(order); free(group_of_first);
  return ngroups;
}

/* ---------------------------------------------------------------------------------------------------------------------
 * orc_join_parallel_i64 -- an ALL-CORES CPU baseline for the inner join on one int64 key column ("optional second
 * baseline: our own multi-threaded C++ CPU oracle (OpenMP, all cores) for a fairer rows/s comparison").  Test / bench
 * infrastructure like the rest of this file; not a restatement of reference code (the reference has no CPU join besides the
 * std::multimap of its tests, join-tests.cu:260-356, which orc_join restates) -- it is what a competent CPU implementation of
 * the same operation looks like: both relations are radix-partitioned on the low bits of a multiplicative hash (one parallel
 * histogram + scatter each), then every partition builds a small open-addressing table of its build tuples and streams its
 * probe tuples past it, partitions in parallel.  Multimap semantics: every (probe row, build row) with equal keys.
 * Checked against orc_join in tests/test_oracle_cpu.py.  Returns the number of pairs (pairs beyond `cap` are counted, not stored).
 * ------------------------------------------------------------------------------------------------------------------- */
#ifdef _OPENMP
#include <omp.h>
#endif
static inline uint32_t par_hash(int64_t k) { return (uint32_t)(((uint64_t)k * 0x9E3779B97F4A7C15ULL) >> 32); }

typedef struct { int64_t key; int32_t row; } par_tuple;

/* tuples of `keys` grouped by partition: out[off[p] .. off[p + 1]) */
static void par_partition(const int64_t *keys, int64_t n, int bits, par_tuple *out, int64_t *off, int threads) {
  const int64_t P = (int64_t)1 << bits;
  int64_t *hist = (int64_t *)calloc((size_t)P * threads, sizeof(int64_t));
#pragma omp parallel num_threads(threads)
  {
#ifdef _OPENMP
    const int t = omp_get_thread_num();
#else
    const int t = 0;
#endif
    const int64_t a = n * t / threads, b = n * (t + 1) / threads;
    int64_t *h = hist + (size_t)t * P;
    for (int64_t i = a; i < b; ++i) ++h[par_hash(keys[i]) & (P - 1)];
#pragma omp barrier
#pragma omp single
    {
      int64_t run = 0;
      for (int64_t p = 0; p < P; ++p) {
        off[p] = run;
        for (int q = 0; q < threads; ++q) { const int64_t c = hist[(size_t)q * P + p]; hist[(size_t)q * P + p] = run; run += c; }
      }
      off[P] = run;
    }
    for (int64_t i = a; i < b; ++i) {
      const int64_t at = h[par_hash(keys[i]) & (P - 1)]++;
      out[at].key = keys[i];
      out[at].row = (int32_t)i;
    }
  }
  free(hist);
}

int64_t orc_join_parallel_i64(const int64_t *probe, int64_t np, const int64_t *build, int64_t nb, int32_t *out_probe, int32_t *out_build,
                              int64_t cap, int threads) {
#ifdef _OPENMP
  if (threads <= 0) threads = omp_get_max_threads();
#else
  threads = 1;
#endif
  int bits = 0;
  while (bits < 16 && ((int64_t)1 << bits) * 2048 < nb) ++bits;          /* ~2048 build tuples per partition: the table stays in L1 / L2 */
  const int64_t P = (int64_t)1 << bits;
  par_tuple *pt = (par_tuple *)malloc(sizeof(par_tuple) * (size_t)(np ? np : 1));
  par_tuple *bt = (par_tuple *)malloc(sizeof(par_tuple) * (size_t)(nb ? nb : 1));
  int64_t *poff = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P + 1)), *boff = (int64_t *)malloc(sizeof(int64_t) * (size_t)(P + 1));
  par_partition(probe, np, bits, pt, poff, threads);
  par_partition(build, nb, bits, bt, boff, threads);
  /* pass 1: pairs per partition; pass 2: write at exact offsets (so that the output needs no per-thread buffers) */
  int64_t *cnt = (int64_t *)calloc((size_t)(P + 1), sizeof(int64_t));
  for (int pass = 0; pass < 2; ++pass) {
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads)
    for (int64_t p = 0; p < P; ++p) {
      const int64_t b0 = boff[p], b1 = boff[p + 1], p0 = poff[p], p1 = poff[p + 1];
      if (b1 == b0 || p1 == p0) continue;
      int64_t slots = 16;
      while (slots < 2 * (b1 - b0)) slots <<= 1;
      int32_t *head = (int32_t *)malloc(sizeof(int32_t) * (size_t)slots);         /* slot -> build tuple index (in bt), -1 empty */
      int32_t *next = (int32_t *)malloc(sizeof(int32_t) * (size_t)(b1 - b0));     /* chain of tuples with the SAME key */
      for (int64_t s = 0; s < slots; ++s) head[s] = -1;
      for (int64_t i = b0; i < b1; ++i) {
        int64_t s = (par_hash(bt[i].key) >> bits) & (slots - 1);
        while (head[s] >= 0 && bt[b0 + head[s]].key != bt[i].key) s = (s + 1) & (slots - 1);
        next[i - b0] = head[s];
        head[s] = (int32_t)(i - b0);
      }
      int64_t found = 0, at = pass ? cnt[p] : 0;
      for (int64_t i = p0; i < p1; ++i) {
        int64_t s = (par_hash(pt[i].key) >> bits) & (slots - 1);
        while (head[s] >= 0 && bt[b0 + head[s]].key != pt[i].key) s = (s + 1) & (slots - 1);
        for (int32_t j = head[s]; j >= 0; j = next[j]) {
          if (pass && at + found < cap) { out_probe[at + found] = pt[i].row; out_build[at + found] = bt[b0 + j].row; }
          ++found;
        }
      }
      if (!pass) cnt[p] = found;
      free(head);
      free(next);
    }
    if (!pass) {
      int64_t run = 0;
      for (int64_t p = 0; p <= P; ++p) { const int64_t c = p < P ? cnt[p] : 0; cnt[p] = run; run += c; }
    }
  }
  const int64_t total = cnt[P];
  free(cnt); free(poff); free(boff); free(pt); free(bt);
  return total;
}

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
