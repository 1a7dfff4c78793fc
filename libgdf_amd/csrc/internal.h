// internal.h -- host-side entry points shared between the translation units of
// libgdf.so (not exported: the Makefile builds with -fvisibility=hidden and only
// the extern "C" ABI of include/gdf/gdf.h is marked default).
#pragma once
#include "common.h"
#include "lab.h"
#include "prof.h"

namespace gdf_amd {

// plumbing.cpp: blocking device -> host copy of a few bytes to a few hundred KB through a pinned staging buffer
hipError_t read_back(void *host_dst, const void *dev_src, size_t bytes);
// ... and in two halves: _begin queues the copy on the launch stream, _end waits for THAT copy only (what was queued behind it keeps running)
struct ReadTicket { bool pending = false; int lane = 0; size_t bytes = 0; const void *dev_src = nullptr; };
hipError_t read_back_begin(ReadTicket *t, const void *dev_src, size_t bytes, int lane);      // lane 0 / 1: one ticket in flight per lane and thread
hipError_t read_back_end(ReadTicket *t, void *host_dst);

// PLACEMENT SEARCH BUDGET of one public call (round 6).  The callers of the placed blocks (join.hip, groupby.hip)
// time candidate blocks on calibration runs INSIDE their call; round 5 ran every search to its end in the first call of a shape --
// 30 - 55 ms instead of ~20, with up to 17 + 9 + 7 fresh multi-GB blocks.  Now a call spends at most its budget on candidates and asks
// the pool to HOLD (DevBuf::alloc_placed with place_draws_now() < 0) once it is spent; the searches go on with the next call.  The first
// searching call of a process gets a small budget (it also pays the cold allocations and the kernels' first launches), later ones more.
void place_budget_begin();                 // at the top of a public entry point that may search
bool place_budget_left();
int place_draws_now(int max_draws);        // max_draws while the budget lasts, then -1 (hold)
struct PlaceRound {                        // charges its own lifetime (one candidate: allocation + calibration run + its report) to the budget
  double t0;
  PlaceRound();
  ~PlaceRound();
};

// plumbing.cpp: the CUs of the CURRENT device, asked once per device (NUM_CU is the MI355X's 256 and only sizes grids of kernels that do
// not care; the lockstep-rounds kernels of scan.hip / filter.hip need every workgroup RESIDENT and size their grids from this -- a
// partitioned or masked device has fewer)
int device_cu_count();

// scan.hip: device-wide prefix sums (in == out allowed)
gdf_error scan_u32(const uint32_t *in, uint32_t *out, size_t n, bool inclusive);
gdf_error scan_u64(const uint64_t *in, uint64_t *out, size_t n, bool inclusive);
// the same without the closing synchronisation: the launches are queued, *scratch (allocated here) must outlive them
gdf_error scan_u32_async(const uint32_t *in, uint32_t *out, size_t n, bool inclusive, DevBuf *scratch);

// sort.hip: stable ascending lexicographic row order of t's first n rows -> perm (n x uint32).
// sorted_keys / keys_exact are optional (see sort.hip).
gdf_error order_rows(const KeyTable &t, uint32_t n, DevBuf &perm, DevBuf *sorted_keys, bool *keys_exact, size_t *perm64 = nullptr,
                     bool *perm64_written = nullptr);
// sort.hip: lo_hi[2c] / lo_hi[2c+1] = min / max (as signed 64-bit) of integer key column c over its valid
// elements of rows [0, t.nrows); lo > hi when the column has none; float columns are not touched
gdf_error key_ranges(const KeyTable &t, long long *lo_hi, int windows = 1, int64_t window_rows = 0);      // windows > 1: a strided sample
// sort.hip: stable LSD radix sort of n (key, 64-bit payload) pairs on the key bits set in `varying`; the
// pairs ping-pong between the two buffer sets and kin / vin point at the sorted data on return
gdf_error radix_sort_pairs_u64(uint64_t *&kin, uint64_t *&kout, uint64_t *&vin, uint64_t *&vout, uint32_t n, uint64_t varying);
// the same with 32-bit keys (12-byte pairs): a quarter less traffic per pass when the key bits fit
gdf_error radix_sort_pairs_k32_u64(uint32_t *&kin, uint32_t *&kout, uint64_t *&vin, uint64_t *&vout, uint32_t n, uint64_t varying);
// sort.hip: SORT-method group-by (op = GbOp of groupby.hip, 5 = COUNT_DISTINCT)
gdf_error group_by_sort(int ncols, gdf_column **cols, gdf_column *col_agg, gdf_column *out_col_indices,
                        gdf_column **out_col_values, gdf_column *out_col_agg, gdf_context *ctxt, int op);

}  // namespace gdf_amd
