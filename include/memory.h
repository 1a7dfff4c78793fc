/*
 * memory.h -- C ABI of librmm.so, the device-memory manager libgdf.so allocates
 * its outputs and scratch from.  Binary compatible with the reference's
 * libgdf/include/memory.h:25-184 (same enum values, same rmmOptions_t layout,
 * same eleven entry points), so librmm_cffi can dlopen it unchanged.
 *
 * Backing store is HIP: "CudaDefaultAllocation" maps to hipMalloc/hipFree and
 * "PoolAllocation" to a size-binned free-list pool carved from hipMalloc (the
 * reference used cnmem, an un-vendored submodule).  The stream argument is an
 * opaque pointer in the reference ABI; a hipStream_t has the same width.
 */
#ifndef GDF_AMD_MEMORY_H
#define GDF_AMD_MEMORY_H

#include <stddef.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library itself is built with -fvisibility=hidden */
#endif

typedef struct CUstream_st *cudaStream_t;   /* memory.h:25: opaque; == hipStream_t here */
typedef long int offset_t;

typedef enum {                              /* memory.h:29-39 */
  RMM_SUCCESS = 0,
  RMM_ERROR_CUDA_ERROR,          /* a HIP runtime error                        */
  RMM_ERROR_INVALID_ARGUMENT,
  RMM_ERROR_NOT_INITIALIZED,
  RMM_ERROR_OUT_OF_MEMORY,
  RMM_ERROR_UNKNOWN,
  RMM_ERROR_IO,
  N_RMM_ERROR
} rmmError_t;

typedef enum { CudaDefaultAllocation = 0, PoolAllocation = 1 } rmmAllocationMode_t;   /* memory.h:41-45 */

typedef struct {                            /* memory.h:50-56 */
  rmmAllocationMode_t allocation_mode;
  size_t              initial_pool_size;   /* 0 = half of the free device memory   */
  bool                enable_logging;      /* record every alloc/realloc/free      */
} rmmOptions_t;

rmmError_t  rmmInitialize(rmmOptions_t *options);           /* memory.h:65  */
rmmError_t  rmmFinalize(void);                              /* memory.h:72  */
const char *rmmGetErrorString(rmmError_t errcode);          /* memory.h:80  */
rmmError_t  rmmAlloc(void **ptr, size_t size, cudaStream_t stream);          /* memory.h:96  */
rmmError_t  rmmRealloc(void **ptr, size_t new_size, cudaStream_t stream);    /* memory.h:112 */
rmmError_t  rmmFree(void *ptr, cudaStream_t stream);                         /* memory.h:124 */
rmmError_t  rmmGetAllocationOffset(offset_t *offset, void *ptr, cudaStream_t stream);  /* memory.h:138 */
rmmError_t  rmmGetInfo(size_t *freeSize, size_t *totalSize, cudaStream_t stream);      /* memory.h:152 */
rmmError_t  rmmWriteLog(const char *filename);              /* memory.h:164 */
size_t      rmmLogSize(void);                               /* memory.h:171 */
rmmError_t  rmmGetLog(char *buffer, size_t buffer_size);    /* memory.h:184 */

/* Extension, no counterpart in the reference: pool blocks of 64 MiB and more as physically contiguous allocations
   (hipDeviceMallocContiguous).  Off by default -- measured slower for every kernel that scatters into such a block,
   DESIGN.md 3.8 -- and kept for A/B runs. */
void        gdf_amd_rmm_contiguous(int on);

/* Extension, no counterpart in the reference: PLACED blocks -- a pool that re-draws slow physical placements (rmm.cpp, place_alloc).
   libgdf.so allocates the multi-GB scratch of its regroup passes through these: `role` says what the block is for, `*measure` comes
   back non-zero while the pool is still comparing placements for this (role, size) and wants to be told on _place_free how long the
   kernels that scatter into the block took (milliseconds; < 0: unknown).  Blocks below 1 GiB and non-pool modes fall through to
   rmmAlloc / rmmFree.  _place_draws: challengers per (role, size), default 4; 0: never re-draw; < 0: plain pool.  max_draws > 0: this
   caller's own number of challengers (one that times a short calibration run per candidate inside ONE call can afford more).  A search ends
   before its last draw once four challengers are drawn and the champion is 7 % faster than the slowest candidate timed (early settle). */
/* Round 6: (a) SIZE CLASSES -- a new entry's block is rounded up to one of eight steps per octave and serves every later request of its
   role between six tenths of the block and all of it (the block itself: the first request + an eighth, rounded up), so callers whose relations change size from call to call keep their
   champions; (b) a search holds at most champion + challenger + 2 losers, draws only blocks of at most a quarter of the free device
   memory, and gives its losers back when it stops moving; (c) max_draws < 0 = HOLD: the champion of the class as it stands, unmeasured,
   nothing drawn, the search stays open -- what libgdf.so asks for once a call has spent its time budget on candidates (the search goes
   on with the next call); (d) on out-of-memory the idle champions and losers of every entry go before a request fails.
   _place_min: test hook, the size from which a request is a placed block (0: the default, 1 GiB). */
rmmError_t  gdf_amd_rmm_place_alloc(int role, size_t size, int max_draws, void **ptr, int *measure);
void        gdf_amd_rmm_place_min(size_t bytes);
rmmError_t  gdf_amd_rmm_place_free(int role, void *ptr, float ms);
void        gdf_amd_rmm_place_draws(int draws);
void        gdf_amd_rmm_place_stats(unsigned long long out[4]);
size_t      gdf_amd_rmm_place_trace(char *buf, size_t cap);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GDF_AMD_MEMORY_H */
