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
 * out[i] = (int32)(in[i] - lo) when lo <= in[i] <= hi, else -1.   in: GDF_INT64 (or DATE64 / TIMESTAMP), no mask;
 * out: caller-preallocated GDF_INT32 of the same size; requires 0 <= hi - lo < 2^31 - 1.
 * The multi-GPU join ships 4-byte keys instead of 8-byte ones when the global build-key range allows it: probe keys
 * outside the range cannot match and become -1, which no narrowed build key equals.
 */
gdf_error gdf_amd_narrow_keys(gdf_column *in, int64_t lo, int64_t hi, gdf_column *out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GDF_AMD_EXT_H */
