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

/*
 * test hook: force one of the library's alternative code paths (csrc/lab.h "path" switches, e.g. "GDF_JK_NO_SPEC") for
 * the calls that follow in this process; value NULL clears the name.  libgdf.so reads NO environment variable -- the
 * parity tests that push one request through two code paths select the second one here (tests/conftest.py force_path).
 */
gdf_error gdf_amd_debug_force(const char *name, const char *value);

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

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GDF_AMD_EXT_H */
