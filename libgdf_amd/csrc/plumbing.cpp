// plumbing.cpp -- host-only ABI helpers: column / context views, error names,
// runtime error passthrough and profiler ranges.
//
// Behaviour follows the reference's src/column.cpp:160-275, src/context.cpp:3-11,
// src/errorhandling.cpp:5-35, src/cudautils.cu:4-14 and src/nvtx_utils.cpp:19-71
// (ranges are forwarded to roctx so they show up in rocprofv3 --marker-trace).
#include "internal.h"

#include <cstdio>
#include <cstring>
#include <atomic>
#include <chrono>
#include <map>
#include <set>
#include <mutex>
#include <string>
#include <roctracer/roctx.h>

// (csrc/testhook.cpp; weak: absent from every process that has not loaded libgdf_testhook.so in front of this library)
extern "C" __attribute__((weak, visibility("default"))) const char *gdf_amd_testhook_forced(const char *name);

namespace gdf_amd {

void note_hip_error(hipError_t e, const char *what, const char *file, int line) {
  std::fprintf(stderr, "ERROR: HIP runtime call %s in line %d of file %s failed with %s (%d).\n", what, line, file,
               hipGetErrorString(e), (int)e);
}

gdf_error make_key_table(gdf_column **cols, int ncols, KeyTable *out) {
  // test hook: stands in for a host allocation of the dispatch code failing (every relational entry point comes through here
  // right after its argument checks and allocates std::vectors afterwards); the entry points turn it into GDF_MEMORYMANAGER_ERROR
  if (lab::path_on("GDF_FORCE_HOST_ALLOC_FAILURE")) throw std::bad_alloc();
  if (ncols > MAX_KEY_COLS) return GDF_JOIN_TOO_MANY_COLUMNS;
  out->ncols = ncols;
  out->any_valid = 0;
  out->nrows = ncols > 0 ? (int64_t)cols[0]->size : 0;
  for (int c = 0; c < ncols; ++c) {
    const ElemKind k = elem_kind(cols[c]->dtype);
    if (k == K_BAD) return GDF_UNSUPPORTED_DTYPE;
    out->col[c].data = cols[c]->data;
    out->col[c].valid = cols[c]->valid;
    out->col[c].kind = (int)k;
    out->col[c].width = kind_width(k);
    if (cols[c]->valid) out->any_valid = 1;
  }
  return GDF_SUCCESS;
}

int device_cu_count() {
  static thread_local int cached_dev = -1, cached = NUM_CU;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return NUM_CU; }
  if (dev != cached_dev) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cached = n;
    else { (void)hipGetLastError(); cached = NUM_CU; }
    cached_dev = dev;
  }
  return cached;
}

// Small device -> host read-backs (counters, flags, histograms) through a pinned staging buffer: hipMemcpy into
// pageable memory takes the runtime's slow path, and a C3 join does eight of them between its kernels.
hipError_t read_back(void *host_dst, const void *dev_src, size_t bytes) {
  static thread_local void *pinned = nullptr;
  static thread_local size_t capacity = 0;
  if (bytes == 0) return hipSuccess;
  if (bytes > capacity) {
    if (pinned) (void)hipHostFree(pinned);
    pinned = nullptr;
    capacity = 0;
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    hipError_t e = hipHostMalloc(&pinned, want, hipHostMallocPortable);
    if (e != hipSuccess) { pinned = nullptr; return hipMemcpy(host_dst, dev_src, bytes, hipMemcpyDeviceToHost); }   // still correct
    capacity = want;
  }
  hipError_t e = hipMemcpyAsync(pinned, dev_src, bytes, hipMemcpyDeviceToHost, stream0());
  if (e != hipSuccess) return e;
  e = hipStreamSynchronize(stream0());
  if (e != hipSuccess) return e;
  std::memcpy(host_dst, pinned, bytes);
  return hipSuccess;
}

// The same in two halves (round 5): the copy is queued where the data is ready, the host comes back for it later -- what it queues in
// between (the scans behind jk_hist, the whole build side behind the skew sample) runs while it waits only for the COPY, not for the
// stream.  Two independent staging areas per thread (`lane`), one ticket in flight per lane.
namespace {
struct ReadLane { void *pinned = nullptr; size_t capacity = 0; hipEvent_t ev = nullptr; };
thread_local ReadLane g_read_lane[2];
}
hipError_t read_back_begin(ReadTicket *t, const void *dev_src, size_t bytes, int lane) {
  ReadLane &l = g_read_lane[lane & 1];
  t->pending = false;
  t->lane = lane & 1;
  t->bytes = bytes;
  t->dev_src = dev_src;
  if (bytes == 0) return hipSuccess;
  if (!l.ev && hipEventCreateWithFlags(&l.ev, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); l.ev = nullptr; return hipSuccess; }   // _end falls back to read_back
  if (bytes > l.capacity) {
    if (l.pinned) (void)hipHostFree(l.pinned);
    l.pinned = nullptr;
    l.capacity = 0;
    const size_t want = bytes < (1u << 18) ? (1u << 18) : bytes;
    if (hipHostMalloc(&l.pinned, want, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); l.pinned = nullptr; return hipSuccess; }
    l.capacity = want;
  }
  hipError_t e = hipMemcpyAsync(l.pinned, dev_src, bytes, hipMemcpyDeviceToHost, stream0());
  if (e != hipSuccess) return e;
  e = hipEventRecord(l.ev, stream0());
  // (the lane's event belongs to the device that was current when it was made; a thread that comes back with another
  // device current cannot record it there -- the ticket then stays unset and _end takes the blocking copy)
  if (e != hipSuccess) { (void)hipGetLastError(); return hipSuccess; }
  t->pending = true;
  return hipSuccess;
}
hipError_t read_back_end(ReadTicket *t, void *host_dst) {
  if (t->bytes == 0) return hipSuccess;
  if (!t->pending) return read_back(host_dst, t->dev_src, t->bytes);        // no event / no pinned memory: the blocking copy
  t->pending = false;
  ReadLane &l = g_read_lane[t->lane];
  const hipError_t e = hipEventSynchronize(l.ev);
  if (e != hipSuccess) return e;
  std::memcpy(host_dst, l.pinned, t->bytes);
  return hipSuccess;
}

// the placement-search budget of one call (internal.h)
namespace {
thread_local double g_place_spent_ms = 0.0, g_place_limit_ms = 0.0;
std::atomic<int> g_place_calls{0};         // public calls that charged a candidate so far (process-wide)
thread_local bool g_place_charged = false;
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}
void place_budget_begin() {
  g_place_spent_ms = 0.0;
  g_place_charged = false;
  // the first searching calls of a process: ~6 ms of candidates next to the cold allocations; then up to 24 ms per call until the
  // searches have settled (a C3 join's three searches need ~20 - 40 ms of candidates in all)
  g_place_limit_ms = g_place_calls.load(std::memory_order_relaxed) < 1 ? 6.0 : 24.0;
  if (const long long forced = lab::path_int("GDF_PLACE_BUDGET_MS", -1); forced >= 0) g_place_limit_ms = (double)forced;      // (test hook: 0 = always hold, large = round 5's behaviour)
}
bool place_budget_left() { return g_place_spent_ms < g_place_limit_ms; }
int place_draws_now(int max_draws) { return place_budget_left() ? max_draws : -1; }
PlaceRound::PlaceRound() : t0(now_ms()) {}
PlaceRound::~PlaceRound() {
  g_place_spent_ms += now_ms() - t0;
  if (!g_place_charged) { g_place_charged = true; g_place_calls.fetch_add(1, std::memory_order_relaxed); }
}

// lab.h: forced paths.  The registry lives in libgdf_testhook.so (csrc/testhook.cpp, test infrastructure); this library only holds a
// WEAK reference to its lookup, bound when libgdf.so is loaded -- null, i.e. one pointer test per lookup, in every process that did not
// load the hook library first.
namespace lab {
const char *forced(const char *name) { return gdf_amd_testhook_forced ? gdf_amd_testhook_forced(name) : nullptr; }
}  // namespace lab

}  // namespace gdf_amd

extern "C" {

gdf_size_type gdf_column_sizeof(void) { return sizeof(gdf_column); }

gdf_error gdf_column_view(gdf_column *column, void *data, gdf_valid_type *valid, gdf_size_type size,
                          gdf_dtype dtype) {
  // dtype_info and col_name are deliberately left untouched (column.cpp:176-188)
  column->data = data;
  column->valid = valid;
  column->size = size;
  column->dtype = dtype;
  column->null_count = 0;
  return GDF_SUCCESS;
}

gdf_error gdf_column_view_augmented(gdf_column *column, void *data, gdf_valid_type *valid, gdf_size_type size,
                                    gdf_dtype dtype, gdf_size_type null_count) {
  column->data = data;
  column->valid = valid;
  column->size = size;
  column->dtype = dtype;
  column->null_count = null_count;
  return GDF_SUCCESS;
}

gdf_error gdf_column_free(gdf_column *column) {
  RMM_TRY(rmmFree(column->data, (cudaStream_t)0));
  RMM_TRY(rmmFree(column->valid, (cudaStream_t)0));
  return GDF_SUCCESS;
}

gdf_error get_column_byte_width(gdf_column *col, int *width) {
  const int w = gdf_amd::dtype_width(col->dtype);
  *width = w;   // -1 for unsupported, as column.cpp:268-271
  return w > 0 ? GDF_SUCCESS : GDF_UNSUPPORTED_DTYPE;
}

gdf_error gdf_context_view(gdf_context *context, int flag_sorted, gdf_method flag_method, int flag_distinct,
                           int flag_sort_result, int flag_sort_inplace) {
  context->flag_sorted = flag_sorted;
  context->flag_method = flag_method;
  context->flag_distinct = flag_distinct;
  context->flag_sort_result = flag_sort_result;
  context->flag_sort_inplace = flag_sort_inplace;
  return GDF_SUCCESS;
}

const char *gdf_error_get_name(gdf_error errcode) {
  static const char *const names[N_GDF_ERRORS] = {
      "GDF_SUCCESS", "GDF_CUDA_ERROR", "GDF_UNSUPPORTED_DTYPE", "GDF_COLUMN_SIZE_MISMATCH",
      "GDF_COLUMN_SIZE_TOO_BIG", "GDF_DATASET_EMPTY", "GDF_VALIDITY_MISSING", "GDF_VALIDITY_UNSUPPORTED",
      "GDF_INVALID_API_CALL", "GDF_JOIN_DTYPE_MISMATCH", "GDF_JOIN_TOO_MANY_COLUMNS", "GDF_DTYPE_MISMATCH",
      "GDF_UNSUPPORTED_METHOD", "GDF_INVALID_AGGREGATOR", "GDF_INVALID_HASH_FUNCTION",
      "GDF_PARTITION_DTYPE_MISMATCH", "GDF_HASH_TABLE_INSERT_FAILURE", "GDF_UNSUPPORTED_JOIN_TYPE", "GDF_C_ERROR",
      "GDF_FILE_ERROR", "GDF_MEMORYMANAGER_ERROR", "GDF_UNDEFINED_NVTX_COLOR", "GDF_NULL_NVTX_NAME"};
  if ((int)errcode < 0 || (int)errcode >= (int)N_GDF_ERRORS) return "Internal error. Unknown error code.";
  return names[(int)errcode];
}

int gdf_cuda_last_error(void) { return (int)hipGetLastError(); }
const char *gdf_cuda_error_string(int cuda_error) { return hipGetErrorString((hipError_t)cuda_error); }
const char *gdf_cuda_error_name(int cuda_error) { return hipGetErrorName((hipError_t)cuda_error); }

gdf_error gdf_nvtx_range_push(char const *const name, gdf_color color) {
  if ((int)color < 0 || (int)color > (int)GDF_NUM_COLORS) return GDF_UNDEFINED_NVTX_COLOR;
  if (!name) return GDF_NULL_NVTX_NAME;
  roctxRangePushA(name);
  return GDF_SUCCESS;
}

gdf_error gdf_nvtx_range_push_hex(char const *const name, unsigned int) {
  if (!name) return GDF_NULL_NVTX_NAME;
  roctxRangePushA(name);
  return GDF_SUCCESS;
}

gdf_error gdf_nvtx_range_pop(void) {
  roctxRangePop();
  return GDF_SUCCESS;
}

}  // extern "C"

