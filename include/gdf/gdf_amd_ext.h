/*
 * gdf_amd_ext.h -- exports of libgdf.so that the reference does NOT have.  Nothing here is needed by a
 * caller of the reference API; they exist for measurement (bench.py, tools/), for tests, and for the multi-GPU
 * layer (libgdf_amd/multigpu.py), which sits above the unchanged gdf_* ABI.
 */
#ifndef GDF_AMD_EXT_H
#define GDF_AMD_EXT_H
#include <gdf/gdf.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* per-kernel HIP-event timing of everything the library launches (csrc/prof.cpp) */
void gdf_amd_profile_enable(int on);
void gdf_amd_profile_reset(void);
int  gdf_amd_profile_read(char (*names)[64], double *total_ms, int *launches, int capacity);

/* test hook: the join's radix partitioner alone (csrc/join.hip, tests/test_gpu_join_internals.py) */
gdf_error gdf_amd_debug_partition(gdf_column *col, int fb, uint64_t *out_key, int32_t *out_idx, uint32_t *out_fine_off,
                                  uint32_t *out_joinable, uint64_t *out_info);

/* (the path-forcing test hook gdf_amd_debug_force is not exported from libgdf.so any more: include/gdf/gdf_amd_testhook.h,
 * libgdf_testhook.so -- test infrastructure, loaded by the tests only) */

/*
 * out[i] = (int32)(in[i] - lo) when lo <= in[i] <= hi, else -1.   in: GDF_INT64 (or DATE64 / TIMESTAMP), no mask;
 * out: caller-preallocated GDF_INT32 of the same size; requires 0 <= hi - lo < 2^31 - 1.
 * The multi-GPU join ships 4-byte keys instead of 8-byte ones when the global build-key range allows it: probe keys
 * outside the range cannot match and become -1, which no narrowed build key equals.
 */
gdf_error gdf_amd_narrow_keys(gdf_column *in, int64_t lo, int64_t hi, gdf_column *out);

/*
 * A build relation partitioned ONCE and probed many times.  gdf_amd_join_build_probe(b, 0, probe...) returns exactly what
 * gdf_inner_join(probe, build) with GDF_HASH returns (probe_indices = left, build_indices = right; library-allocated
 * int32 columns, gdf_column_free them), left_join != 0 what gdf_left_join returns -- except that the table is always on
 * the prepared relation, never on the smaller one.  The build columns' DATA is not copied: it must stay alive and
 * unchanged until gdf_amd_join_build_free.  The multi-GPU join probes the received build relation once per slice of
 * the probe relation.
 */
typedef struct gdf_amd_join_build gdf_amd_join_build;
gdf_error gdf_amd_join_build_create(gdf_column **build_cols, int num_cols, gdf_amd_join_build **out);
gdf_error gdf_amd_join_build_probe(gdf_amd_join_build *build, int left_join, gdf_column **probe_cols, int num_cols,
                                   gdf_column *probe_indices, gdf_column *build_indices);
void gdf_amd_join_build_free(gdf_amd_join_build *build);

/*
 * A probe relation that arrives in slices (the multi-GPU join receives it that way), partitioned slice by slice and
 * probed ONCE: probe_indices number the rows across the slices in the order they were added; the result is what
 * gdf_amd_join_build_probe(build, 0, all slices concatenated) returns.  INNER joins on one unmasked integer key column
 * whose values fit the 32-bit tuple format only, and expected_rows >= 2^22 (an estimate of the total; a few percent of
 * head-room are built in): anything else -- and a slice sequence so skewed that a partition outgrows its room --
 * returns GDF_UNSUPPORTED_METHOD from _begin / _add / _finish, and the caller probes its slices one by one instead.
 * _finish releases the object whatever it returns.
 */
typedef struct gdf_amd_join_probe gdf_amd_join_probe;
gdf_error gdf_amd_join_probe_begin(gdf_amd_join_build *build, size_t expected_rows, gdf_amd_join_probe **out);
gdf_error gdf_amd_join_probe_add(gdf_amd_join_probe *probe, gdf_column **probe_cols, int num_cols);
gdf_error gdf_amd_join_probe_finish(gdf_amd_join_probe *probe, gdf_column *probe_indices, gdf_column *build_indices);

/*
 * The sender side of the multi-GPU shuffle in ONE pass over the keys (instead of narrow + row-number column +
 * gdf_hash_partition): partitions (key', row) on Murmur3(key') into num_partitions partitions exactly as
 * gdf_hash_partition(GDF_HASH_MURMUR3) places them, where row = row_base + i and key' = keys[i], or, with narrow != 0
 * (keys GDF_INT64-like, out_keys GDF_INT32), the gdf_amd_narrow_keys(lo, hi) image of it.  out_keys / out_rows
 * (GDF_INT32) are caller-preallocated with keys->size rows; partition_offsets is a HOST array of num_partitions ints.
 */
gdf_error gdf_amd_shuffle_partition(gdf_column *keys, int narrow, int64_t lo, int64_t hi, int32_t row_base, int num_partitions,
                                    gdf_column *out_keys, gdf_column *out_rows, int partition_offsets[]);

/*
 * The same split as gdf_amd_shuffle_partition without a row-number column: the partition is STABLE (the keys of a
 * partition keep their input order) and bitmap p -- ceil(rows / 64) 64-bit words at bitmaps + p * ceil(rows / 64), DEVICE
 * memory, bit i of word i / 64 -- has bit i set iff input row i went to partition p.  The j-th key of partition p is
 * therefore input row select(bitmap p, j): a receiver names the original row of a joined key from one bit per row
 * instead of a 4-byte row number (4.125 instead of 8 bytes per row on the links).  num_partitions <= 64.
 * narrow = 2: as narrow = 1, and a row whose key lies outside [lo, hi] -- it cannot match any build key of an inner
 * join -- is DROPPED: it goes to no partition and sets no bitmap bit (with narrow = 1 all such rows become the key -1 and
 * pile up on one rank).  partition_offsets then has num_partitions + 1 entries, the last one = rows that travel.
 */
gdf_error gdf_amd_shuffle_partition_stable(gdf_column *keys, int narrow, int64_t lo, int64_t hi, int num_partitions,
                                           gdf_column *out_keys, uint64_t *bitmaps, int partition_offsets[]);

/*
 * FUSED multi-GPU join: the sender runs the join's level-1 regroup, the receiver continues at level 2 (csrc/join.hip "FUSED
 * multi-GPU join"; libgdf_amd/multigpu.py fused_inner_join).  All ranks share one hash space; rank = mulhi(hash, world).
 *
 * gdf_amd_fj_plan    layout every rank derives from GLOBAL numbers only: fine / coarse partition bits of a rank's share of
 *                    `build_rows_total`, and the room `cap` (keys per (bin, XCD) region) for calls of at most `rows_max` rows
 *                    whose keys repeat `rows_per_key` times on average (all copies of a key share a region).
 *                    GDF_UNSUPPORTED_METHOD: this world size / relation size does not fit the path (use the key shuffle).
 * gdf_amd_fj_send    int64 / int32 keys -> out_keys: (world << coarse_bits) bins x 8 regions x cap 4-byte keys (key - lo; rows
 *                    outside [lo, hi] are dropped), rank r's keys in the r-th contiguous block of (8 << coarse_bits) * cap
 *                    elements (+ one tile = 32768 elements of dump space behind the last block); out_pos[i]: the position in
 *                    out_keys that row i's key went to (0xffffffff: dropped) -- it stays with the sender, who can tell from it
 *                    which row a result position names; out_fill: DEVICE array of (world << coarse_bits) * 8 fill counters
 *                    + 1 word.  *overflowed = 1: some region outgrew cap (skewed keys), the buffers are unusable.
 * gdf_amd_fj_build_create   receive buffer of the build relation (the blocks every sender made for this rank, sender-major,
 *                    with their fill counters in the same order, both in DEVICE memory) -> a build handle
 *                    (gdf_amd_join_probe_begin / gdf_amd_fj_probe_add / gdf_amd_join_probe_finish / gdf_amd_join_build_free).
 * gdf_amd_fj_probe_add      one receive buffer of the probe relation; result indices are POSITIONS: position_base + offset in
 *                    this buffer for the probe side, offset in the build receive buffer for the build side.
 */
gdf_error gdf_amd_fj_plan(int world, int64_t build_rows_total, int64_t rows_max, double rows_per_key, int *fine_bits, int *coarse_bits,
                          uint32_t *cap);
gdf_error gdf_amd_fj_send(gdf_column *keys, int64_t lo, int64_t hi, int world, int coarse_bits, uint32_t cap,
                          uint32_t *out_keys, uint32_t *out_pos, uint32_t *out_fill, int *overflowed);
gdf_error gdf_amd_fj_build_create(const uint32_t *recv_keys, const uint32_t *recv_fill, int world, int64_t lo, int fine_bits,
                                  int coarse_bits, uint32_t cap, int64_t expected_rows, gdf_amd_join_build **out);
gdf_error gdf_amd_fj_probe_add(gdf_amd_join_probe *probe, const uint32_t *recv_keys, const uint32_t *recv_fill, uint32_t cap,
                               int64_t position_base, int64_t buffer_elems);

/*
 * gdf_amd_dist_inner_join -- the multi-GPU inner join behind the C ABI (one process per GPU; BASELINE.json north_star: "host code
 * stays C++ behind the same C ABI ... radix-partitioned on key and shuffled with RCCL all-to-all over xGMI").  No counterpart in
 * the reference, which is single-GPU.  It runs the FUSED join above end to end: global key
 * range and sizes (three small all-reduces), gdf_amd_fj_plan, the sender's level-1 regroup of the build relation and of the probe
 * relation in `chunks` slices, the all-to-all of EQUAL blocks + fill counters per slice (software-pipelined: slice c travels while
 * slice c + 1 is regrouped and slice c - 1 is partitioned at the receiver), level 2 + LDS probe at the receiver, and the
 * collective agreement that decides between result and decline.
 *
 * The wire is a gdf_amd_transport: four function pointers and an optional fifth (all_to_all_v).  gdf_amd_rccl_transport_create gives the RCCL one (ncclSend / ncclRecv
 * groups on a stream of its own, librccl resolved with dlopen at that moment -- libgdf.so itself does not link against it); a host
 * in another language, or a test that moves the blocks through host memory between processes (tests/multirank_common.py), fills
 * the struct itself.
 *
 * Every rank of the transport calls with its shard: one GDF_INT64 (or GDF_INT32) key column each, no validity masks.
 *   probe_pos / build_pos   caller-allocated DEVICE arrays, one uint32 per local probe / build row: the position that row's key
 *                           went to in its send buffer (0xffffffff: dropped, the key lies outside the global build range) -- they
 *                           stay with the sender, who can tell from them which of its rows a result position names
 *   probe_indices / build_indices   library-allocated GDF_INT32 columns (gdf_column_free), exactly as gdf_inner_join's: this
 *                           rank's pairs as POSITIONS in its receive buffers -- probe: slice * world * block_p + offset in slice's
 *                           buffer, build: offset in the build receive buffer; offset / block = the sender's rank
 *   info                    the layout both sides were exchanged in (what turns positions into (rank, row))
 *   *declined = 1           ON EVERY RANK: the shape does not fit the fused path (keys that do not narrow to 31 bits, a relation
 *                           or world size outside gdf_amd_fj_plan's range, skewed keys that overflow the fixed-size regions, the
 *                           31-bit position space) -- nothing is returned and the caller takes the key shuffle
 *                           (libgdf_amd/multigpu.py distributed_inner_join) on all ranks together.
 * A hard LOCAL error (out of memory, a HIP failure) does not make its rank drop out of the collectives: every wire buffer is
 * allocated before the first agreement, the rank keeps posting its (useless) blocks, reports the failure at the next agreement and
 * ALL ranks leave there -- the rank that met the error with its code, its peers with GDF_C_ERROR.  A failing TRANSPORT call
 * (all_to_all / wait / all_reduce_i64 returning non-zero) is returned at once; the peers then see their own transport fail or time out.
 */
typedef struct gdf_amd_transport {
  void *ctx;
  int rank, world;
  /* DEVICE buffers: block r of `send` (bytes_per_rank bytes) goes to rank r, block s of `recv` arrives from rank s.  Ordered behind
     the work the library has issued so far; *ticket names the exchange.  0 = success. */
  int (*all_to_all)(void *ctx, const void *send, void *recv, size_t bytes_per_rank, void **ticket);
  /* orders the library's stream behind the exchange (recv may be read, send reused, afterwards); releases the ticket */
  int (*wait)(void *ctx, void *ticket);
  /* HOST values, reduced in place over all ranks; op 0 = min, 1 = max, 2 = sum */
  int (*all_reduce_i64)(void *ctx, int64_t *values, int count, int op);
  /* releases ctx (may be NULL) */
  void (*destroy)(void *ctx);
  /* OPTIONAL, may be NULL (round 6; the all-to-all-v SURVEY 8(e) names): bytes [send_off[r], send_off[r + 1]) of the DEVICE buffer `send`
     go to rank r, bytes [recv_off[s], recv_off[s + 1]) of `recv` arrive from rank s.  The offset arrays are HOST arrays of world + 1
     entries; what rank a sends to rank b is what b expects from a (the callers exchange their counts first).  Ordering, ticket and
     return value as all_to_all.  With it gdf_amd_dist_group_by* and gdf_amd_dist_shuffle_join ship EXACT sizes straight from their
     partitioned columns into the result columns; without it they pad to equal blocks and stage them.  LAST member: a host that
     fills the struct itself and was written against the four-function version must zero it. */
  int (*all_to_all_v)(void *ctx, const void *send, const size_t *send_off, void *recv, const size_t *recv_off, void **ticket);
} gdf_amd_transport;

typedef struct gdf_amd_dist_info {
  int world, chunks;
  int64_t lo, hi;                      /* global build-side key range: keys travel as (key - lo) */
  int64_t slice_rows;                  /* probe rows per slice (the last one may be shorter) */
  int fine_bits_p, coarse_bits_p;      /* gdf_amd_fj_plan of the probe slices ...                */
  uint32_t cap_p;
  int fine_bits_b, coarse_bits_b;      /* ... and of the build relation                          */
  uint32_t cap_b;
  int64_t block_p, block_b;            /* elements every sender makes for one rank: (8 << coarse_bits) * cap */
} gdf_amd_dist_info;

gdf_error gdf_amd_dist_inner_join(gdf_column *probe_keys, gdf_column *build_keys, gdf_amd_transport *transport, int chunks,
                                  uint32_t *probe_pos, uint32_t *build_pos, gdf_column *probe_indices, gdf_column *build_indices,
                                  gdf_amd_dist_info *info, int *declined);

/* The KEY SHUFFLE join behind the C ABI (csrc/dist_ops.hip) -- what all ranks take TOGETHER when gdf_amd_dist_inner_join says
 * *declined = 1 (keys that do not narrow to 31 bits, skewed keys, a shape outside gdf_amd_fj_plan's range).  COLLECTIVE; any
 * int32 / int64 key column without a mask, any shard sizes (also none).  Every rank splits (key, local row number) of both
 * relations by owner rank = Murmur3(key) % world (gdf_amd_shuffle_partition), the partitions travel as equal blocks over
 * transport->all_to_all (block size and failures agreed by one all-reduce per relation), the owner joins what it received with
 * gdf_inner_join (src/join/joining.cu:571-653 per rank) and resolves the pairs to GLOBAL row ids:
 *   out_probe_ids / out_build_ids   library-allocated GDF_INT64 columns (gdf_column_free): (owner rank << 40) | local row of the
 *                                   probe / build row of every pair this rank produced; every pair of the global join comes out
 *                                   on exactly one rank, in no particular order.
 * One exchange per relation, not pipelined: the fallback trades the fused path's overlap for generality.  Local errors travel
 * into the agreements as in gdf_amd_dist_inner_join. */
gdf_error gdf_amd_dist_shuffle_join(gdf_column *probe_keys, gdf_column *build_keys, gdf_amd_transport *transport,
                                    gdf_column *out_probe_ids, gdf_column *out_build_ids);
/* ... and as a LEFT / FULL join (round 6; reference: gdf_left_join / gdf_full_join, src/join/joining.cu:571-653, per owner rank): every row
 * of either relation reaches exactly one owner, so a probe row without a partner comes out once as (global id, -1) and -- FULL -- a
 * build row without a partner once as (-1, global id).  Semantics of unmatched rows and of the pair order as the single-GPU calls. */
gdf_error gdf_amd_dist_shuffle_left_join(gdf_column *probe_keys, gdf_column *build_keys, gdf_amd_transport *transport,
                                         gdf_column *out_probe_ids, gdf_column *out_build_ids);
gdf_error gdf_amd_dist_shuffle_full_join(gdf_column *probe_keys, gdf_column *build_keys, gdf_amd_transport *transport,
                                         gdf_column *out_probe_ids, gdf_column *out_build_ids);

/* DISTRIBUTED MATERIALISATION (round 6): the multi-GPU face of the joins' result_cols step (src/join/joining.cu:375-479 gathers the
 * relations' columns by the index columns; across ranks the index is a global row id).  COLLECTIVE.  `ids` is a GDF_INT64 column
 * without a mask of global row ids as the gdf_amd_dist_shuffle_*join entries produce them -- (owner rank << 40) | local row, or -1
 * for the missing side of an unmatched row -- in any number and any order (also none); `columns` are 1 ... 16 columns of THIS rank's
 * shard of the relation the ids name (equal sizes, any fixed-width dtype, validity masks honoured).  Every id is asked of its owner
 * (the local rows travel over the transport, split by owner), the owner reads its shard and the values travel back:
 *   outs[c]   library-allocated column (gdf_column_free) of ids->size rows in the dtype of columns[c], ALWAYS with a validity mask:
 *             row i is columns[c]'s row named by ids[i] on its owner -- null where ids[i] is -1 or the source row is null -- and
 *             null_count says how many are.
 * Errors: an id that names no rank or no row of its owner's shard is GDF_INVALID_API_CALL; local errors are carried into the next
 * agreement as in the entries above, every rank returns together. */
gdf_error gdf_amd_dist_gather(gdf_column *ids, int ncols, gdf_column **columns, gdf_amd_transport *transport, gdf_column **outs);

/* MULTI-GPU GROUP-BY behind the C ABI (csrc/dist_ops.hip; no counterpart in the reference, which is single-GPU -- per rank it
 * composes gdf_group_by_<op>, src/sqls_ops.cu:1426-1487, and gdf_hash_partition, src/hashing.cu:559-654).  COLLECTIVE: every rank
 * of the transport calls it with its row shard of (keys, values) -- one int32 / int64 key column, one numeric value column, no
 * validity masks, possibly no rows.  Every rank pre-aggregates its shard, the partial aggregates travel to the rank
 * Murmur3(key) % world owns (equal blocks over transport->all_to_all, their size agreed by one all-reduce) and are combined there:
 * partial sums / minima / maxima by the same operator, partial counts (int64) by a sum, AVG as the quotient of the combined sums
 * (accumulated in int64 / float64) and counts -- a GDF_FLOAT64 column.
 *   out_keys / out_agg   library-allocated columns (gdf_column_free) with THIS RANK'S groups, sorted by key; every group of the
 *                        global relation comes out on exactly one rank.  SUM / MIN / MAX keep the value dtype (a sum wraps as the
 *                        single-GPU gdf_group_by_sum does), COUNT is GDF_INT64, AVG GDF_FLOAT64.
 * Errors: a local error (bad arguments on one rank, out of memory) is carried into the next agreement -- the block size's, or the one
 * behind the wire buffers' allocation (round 6: two all-reduces per exchange) -- and that rank returns its code, the others GDF_C_ERROR:
 * nobody is left waiting in a collective.  A rank whose local work fails BEHIND the last exchange returns its code alone (the others have
 * their results).  op: GDF_SUM, GDF_MIN, GDF_MAX, GDF_COUNT, GDF_AVG. */
gdf_error gdf_amd_dist_group_by(gdf_agg_op op, gdf_column *keys, gdf_column *values, gdf_amd_transport *transport,
                                gdf_column *out_keys, gdf_column *out_agg);
gdf_error gdf_amd_dist_group_by_sum(gdf_column *keys, gdf_column *values, gdf_amd_transport *transport, gdf_column *out_keys, gdf_column *out_agg);
gdf_error gdf_amd_dist_group_by_min(gdf_column *keys, gdf_column *values, gdf_amd_transport *transport, gdf_column *out_keys, gdf_column *out_agg);
gdf_error gdf_amd_dist_group_by_max(gdf_column *keys, gdf_column *values, gdf_amd_transport *transport, gdf_column *out_keys, gdf_column *out_agg);
gdf_error gdf_amd_dist_group_by_count(gdf_column *keys, gdf_column *values, gdf_amd_transport *transport, gdf_column *out_keys, gdf_column *out_agg);
gdf_error gdf_amd_dist_group_by_avg(gdf_column *keys, gdf_column *values, gdf_amd_transport *transport, gdf_column *out_keys, gdf_column *out_agg);

/* The same over SEVERAL key columns and WITH validity masks (round 6; BASELINE configuration C5 -- a two-key, masked AVG -- across ranks).
 * Reference shape: gdf_group_by_* over ncols key columns (src/sqls_ops.cu:1085-1363, src/groupby/groupby.cuh:208-250); a group's owner
 * is its ROW HASH % world -- the Murmur3 fold over the key columns of src/gdf_table.cuh:704-854, i.e. gdf_hash_partition on all key
 * columns.  Key columns: any dtype the local group-by takes; values: numeric.  Mask semantics are the local HASH group-by's extension
 * (the reference rejects masks, sqls_ops.cu:1103-1106): a row with a null in ANY key column is dropped, a null value is skipped, a group
 * without a valid value reports 0 and a cleared validity bit (COUNT: 0, valid).  Per rank: the partial aggregate (AVG: the sum of the
 * widened values) and the number of valid values of every local group travel to the owner, which combines the partials that had a valid
 * value by the same operator and the counts by a sum; AVG = combined sum / combined count (src/groupby/groupby.cuh:308-419 per rank).
 *   out_keys[ncols], out_agg   library-allocated columns (gdf_column_free) with THIS RANK'S groups in ascending lexicographic key order;
 *                              out_agg carries a validity mask (and null_count) unless op is GDF_COUNT.  Dtypes as the single-key entry.
 * Errors travel into the agreements as above; two small all-reduces (block size; "every rank has its wire buffers") precede the exchange. */
gdf_error gdf_amd_dist_group_by_multi(gdf_agg_op op, int ncols, gdf_column **keys, gdf_column *values, gdf_amd_transport *transport,
                                      gdf_column **out_keys, gdf_column *out_agg);

/* RCCL transport.  id: the 128 bytes of an ncclUniqueId -- made by ONE rank with gdf_amd_rccl_unique_id and handed to the others by
   whatever the host has (MPI, a file, torch.distributed's store); every rank then calls gdf_amd_rccl_transport_create (collective:
   ncclCommInitRank) with the device it computes on current.  gdf_amd_transport_free destroys communicator and stream. */
gdf_error gdf_amd_rccl_unique_id(char id[128]);
gdf_error gdf_amd_rccl_transport_create(const char id[128], int world, int rank, gdf_amd_transport **out);
void gdf_amd_transport_free(gdf_amd_transport *transport);
/* the communicator's own answer (ncclCommCount / ncclCommUserRank) for a transport made by gdf_amd_rccl_transport_create */
gdf_error gdf_amd_rccl_transport_ranks(gdf_amd_transport *transport, int *nranks, int *rank);
/* synchronous copy on the library's stream for transports that stage blocks through host memory; direction 0: device -> host,
   1: host -> device, 2: device -> device */
gdf_error gdf_amd_copy(void *dst, const void *src, size_t bytes, int direction);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GDF_AMD_EXT_H */
