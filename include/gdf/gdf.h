/*
 * gdf.h -- the C ABI of the MI355X-native libgdf.so (drop-in boundary).
 *
 * Every struct, enum value and entry point below is binary compatible with the
 * interface the reference's cffi bindings dlopen (citations are relative to
 * /root/reference/libgdf):
 *   structs / enums ......... include/gdf/cffi/types.h:1-221
 *   entry points ............ include/gdf/cffi/functions.h:1-785
 *   csv / csr argument PODs . include/gdf/cffi/io_types.h, convert_types.h
 * Layout facts pinned by tests/test_abi.py: sizeof(gdf_column)==56 with
 * data@0 valid@8 size@16 dtype@24 null_count@32 dtype_info@40 col_name@48;
 * sizeof(gdf_context)==20; enums are positional ints.
 *
 * The header is plain C (usable from cgo / JNI / ctypes / cffi) and is also
 * what the C++ host code in libgdf_amd/csrc compiles against.  Entry points
 * that are outside the relational hot path are declared
 * through gdf_unsupported.def; they are exported and return
 * GDF_UNSUPPORTED_METHOD (or a null handle).
 */
#ifndef GDF_AMD_GDF_H
#define GDF_AMD_GDF_H

#include <stddef.h>
#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library itself is built with -fvisibility=hidden */
#endif

/* ---- scalar typedefs (types.h:3-8) ------------------------------------- */
typedef size_t         gdf_size_type;
typedef gdf_size_type  gdf_index_type;
typedef unsigned char  gdf_valid_type;   /* 8 rows per mask byte, LSB first   */
typedef long           gdf_date64;
typedef int            gdf_date32;
typedef int            gdf_category;

#define GDF_VALID_BITSIZE 8              /* gdf.h:10 in the reference        */

/* ---- column element types (types.h:15-29) ------------------------------ */
typedef enum {
  GDF_invalid = 0,
  GDF_INT8 = 1, GDF_INT16 = 2, GDF_INT32 = 3, GDF_INT64 = 4,
  GDF_FLOAT32 = 5, GDF_FLOAT64 = 6,
  GDF_DATE32 = 7,      /* int32 days since epoch                             */
  GDF_DATE64 = 8,      /* int64 ms since epoch                               */
  GDF_TIMESTAMP = 9,   /* int64, unit in dtype_info                          */
  GDF_CATEGORY = 10, GDF_STRING = 11,
  N_GDF_TYPES = 12
} gdf_dtype;

/* ---- status codes (types.h:39-64); names via gdf_error_get_name -------- */
typedef enum {
  GDF_SUCCESS = 0,
  GDF_CUDA_ERROR = 1,                /* a HIP runtime call failed            */
  GDF_UNSUPPORTED_DTYPE = 2,
  GDF_COLUMN_SIZE_MISMATCH = 3,
  GDF_COLUMN_SIZE_TOO_BIG = 4,
  GDF_DATASET_EMPTY = 5,
  GDF_VALIDITY_MISSING = 6,
  GDF_VALIDITY_UNSUPPORTED = 7,
  GDF_INVALID_API_CALL = 8,
  GDF_JOIN_DTYPE_MISMATCH = 9,
  GDF_JOIN_TOO_MANY_COLUMNS = 10,
  GDF_DTYPE_MISMATCH = 11,
  GDF_UNSUPPORTED_METHOD = 12,
  GDF_INVALID_AGGREGATOR = 13,
  GDF_INVALID_HASH_FUNCTION = 14,
  GDF_PARTITION_DTYPE_MISMATCH = 15,
  GDF_HASH_TABLE_INSERT_FAILURE = 16,
  GDF_UNSUPPORTED_JOIN_TYPE = 17,
  GDF_C_ERROR = 18,
  GDF_FILE_ERROR = 19,
  GDF_MEMORYMANAGER_ERROR = 20,
  GDF_UNDEFINED_NVTX_COLOR = 21,
  GDF_NULL_NVTX_NAME = 22,
  N_GDF_ERRORS = 23
} gdf_error;

typedef enum { GDF_HASH_MURMUR3 = 0, GDF_HASH_IDENTITY = 1 } gdf_hash_func;   /* types.h:66-69 */

typedef enum {                                                               /* types.h:71-77 */
  TIME_UNIT_NONE = 0, TIME_UNIT_s, TIME_UNIT_ms, TIME_UNIT_us, TIME_UNIT_ns
} gdf_time_unit;

typedef struct { gdf_time_unit time_unit; } gdf_dtype_extra_info;            /* types.h:79-82 */

/* ---- the Arrow-layout column descriptor (types.h:84-92) ----------------
 * A host-side POD; `data` and `valid` are DEVICE pointers.  valid==NULL means
 * "no nulls".  Bit i of the mask is (valid[i/8] >> (i%8)) & 1, 1 = not null
 * (include/gdf/utils.h:9-16). */
typedef struct gdf_column_ {
  void                 *data;
  gdf_valid_type       *valid;
  gdf_size_type         size;
  gdf_dtype             dtype;
  gdf_size_type         null_count;
  gdf_dtype_extra_info  dtype_info;
  char                 *col_name;      /* host string, never touched here    */
} gdf_column;

typedef enum { GDF_SORT = 0, GDF_HASH = 1, N_GDF_METHODS = 2 } gdf_method;   /* types.h:101-105 */

typedef enum {                                                               /* types.h:107-114 */
  GDF_QUANT_LINEAR = 0, GDF_QUANT_LOWER, GDF_QUANT_HIGHER, GDF_QUANT_MIDPOINT,
  GDF_QUANT_NEAREST, N_GDF_QUANT_METHODS
} gdf_quantile_method;

typedef enum {                                                               /* types.h:123-131 */
  GDF_SUM = 0, GDF_MIN, GDF_MAX, GDF_AVG, GDF_COUNT, GDF_COUNT_DISTINCT, N_GDF_AGG_OPS
} gdf_agg_op;

typedef enum {                                                               /* types.h:142-153 */
  GDF_GREEN = 0, GDF_BLUE, GDF_YELLOW, GDF_PURPLE, GDF_CYAN, GDF_RED, GDF_WHITE,
  GDF_DARK_GREEN, GDF_ORANGE, GDF_NUM_COLORS
} gdf_color;

/* ---- per-call options (types.h:161-167) -------------------------------- */
typedef struct gdf_context_ {
  int        flag_sorted;        /* input already sorted? (unused by HASH)   */
  gdf_method flag_method;        /* GDF_HASH selects everything in this lib  */
  int        flag_distinct;
  int        flag_sort_result;   /* HASH group-by: 1 = sort output by key    */
  int        flag_sort_inplace;
} gdf_context;

/* opaque handles of out-of-scope subsystems (types.h:169-182) */
typedef struct _OpaqueIpcParser              gdf_ipc_parser_type;
typedef struct _OpaqueRadixsortPlan          gdf_radixsort_plan_type;
typedef struct _OpaqueSegmentedRadixsortPlan gdf_segmented_radixsort_plan_type;

typedef enum { GDF_ORDER_ASC = 0, GDF_ORDER_DESC } order_by_type;            /* types.h:183-186 */

typedef enum {                                                               /* types.h:188-195 */
  GDF_EQUALS = 0, GDF_NOT_EQUALS, GDF_LESS_THAN, GDF_LESS_THAN_OR_EQUALS,
  GDF_GREATER_THAN, GDF_GREATER_THAN_OR_EQUALS
} gdf_comparison_operator;

typedef enum { GDF_WINDOW_RANGE = 0, GDF_WINDOW_ROW } window_function_type;  /* types.h:197-200 */
typedef enum {                                                               /* types.h:202-210 */
  GDF_WINDOW_AVG = 0, GDF_WINDOW_SUM, GDF_WINDOW_MAX, GDF_WINDOW_MIN, GDF_WINDOW_COUNT,
  GDF_WINDOW_STDDEV, GDF_WINDOW_VAR
} window_reduction_type;

/* argument PODs of the csv reader / csr converter (io_types.h:26-60,
 * convert_types.h:33-41) -- present only so the stubs have the right shape. */
typedef struct {
  int num_cols_out; int num_rows_out; gdf_column **data;
  char *file_path; char lineterminator; char delimiter; bool delim_whitespace; bool skipinitialspace;
  int num_cols; const char **names; const char **dtype;
  int skiprows; int skipfooter; bool dayfirst;
} csv_read_arg;
typedef struct csr_gdf_ {
  void *A; gdf_size_type *IA; int64_t *JA; gdf_dtype dtype; int64_t nnz;
  gdf_size_type rows; gdf_size_type cols;
} csr_gdf;

/* ======================================================================== *
 *  Hot-path entry points                            *
 * ======================================================================== */

/* --- column / context / error plumbing ---------------------------------- *
 * replaces src/column.cpp:160-275, src/context.cpp:3-11,
 * src/errorhandling.cpp:5-35, src/cudautils.cu:4-14 (functions.h:33-107)   */
gdf_size_type gdf_column_sizeof(void);
gdf_error gdf_column_view(gdf_column *column, void *data, gdf_valid_type *valid,
                          gdf_size_type size, gdf_dtype dtype);
gdf_error gdf_column_view_augmented(gdf_column *column, void *data, gdf_valid_type *valid,
                                    gdf_size_type size, gdf_dtype dtype, gdf_size_type null_count);
gdf_error gdf_column_free(gdf_column *column);           /* rmmFree(data), rmmFree(valid) */
gdf_error gdf_column_concat(gdf_column *output, gdf_column *columns_to_concat[], int num_columns);
gdf_error get_column_byte_width(gdf_column *col, int *width);
gdf_error gdf_context_view(gdf_context *context, int flag_sorted, gdf_method flag_method,
                           int flag_distinct, int flag_sort_result, int flag_sort_inplace);
const char *gdf_error_get_name(gdf_error errcode);
int         gdf_cuda_last_error(void);                   /* hipGetLastError()              */
const char *gdf_cuda_error_string(int cuda_error);       /* hipGetErrorString              */
const char *gdf_cuda_error_name(int cuda_error);         /* hipGetErrorName                */

/* --- profiler ranges: src/nvtx_utils.cpp:19-71 -> roctx (functions.h:18-31) */
gdf_error gdf_nvtx_range_push(char const *const name, gdf_color color);
gdf_error gdf_nvtx_range_push_hex(char const *const name, unsigned int color);
gdf_error gdf_nvtx_range_pop(void);

/* --- valid-mask helpers: src/validops.cu:86-256, src/binaryops.cu (functions.h:32,674) */
gdf_error gdf_count_nonzero_mask(gdf_valid_type const *masks, int num_rows, int *count);
gdf_error gdf_validity_and(gdf_column *lhs, gdf_column *rhs, gdf_column *output);

/* --- hash join: src/join/joining.cu:571-653 (functions.h:226-318) --------
 * left_indices/right_indices receive LIBRARY-allocated GDF_INT32 device
 * arrays of exactly the number of joined pairs (free with gdf_column_free);
 * pair order is unspecified.  result_cols (optional) receives the gathered
 * rows: [left non-key..., key..., right non-key...]. */
gdf_error gdf_inner_join(gdf_column **left_cols, int num_left_cols, int left_join_cols[],
                         gdf_column **right_cols, int num_right_cols, int right_join_cols[],
                         int num_cols_to_join, int result_num_cols, gdf_column **result_cols,
                         gdf_column *left_indices, gdf_column *right_indices,
                         gdf_context *join_context);
gdf_error gdf_left_join(gdf_column **left_cols, int num_left_cols, int left_join_cols[],
                        gdf_column **right_cols, int num_right_cols, int right_join_cols[],
                        int num_cols_to_join, int result_num_cols, gdf_column **result_cols,
                        gdf_column *left_indices, gdf_column *right_indices,
                        gdf_context *join_context);
gdf_error gdf_full_join(gdf_column **left_cols, int num_left_cols, int left_join_cols[],
                        gdf_column **right_cols, int num_right_cols, int right_join_cols[],
                        int num_cols_to_join, int result_num_cols, gdf_column **result_cols,
                        gdf_column *left_indices, gdf_column *right_indices,
                        gdf_context *join_context);

/* --- hash group-by: src/sqls_ops.cu:1426-1487 (functions.h:727-772) ------
 * Outputs are caller-preallocated; ->size of every output is set to the
 * number of groups.  out_col_indices is ignored by the HASH method. */
gdf_error gdf_group_by_sum(int ncols, gdf_column **cols, gdf_column *col_agg,
                           gdf_column *out_col_indices, gdf_column **out_col_values,
                           gdf_column *out_col_agg, gdf_context *ctxt);
gdf_error gdf_group_by_min(int ncols, gdf_column **cols, gdf_column *col_agg,
                           gdf_column *out_col_indices, gdf_column **out_col_values,
                           gdf_column *out_col_agg, gdf_context *ctxt);
gdf_error gdf_group_by_max(int ncols, gdf_column **cols, gdf_column *col_agg,
                           gdf_column *out_col_indices, gdf_column **out_col_values,
                           gdf_column *out_col_agg, gdf_context *ctxt);
gdf_error gdf_group_by_avg(int ncols, gdf_column **cols, gdf_column *col_agg,
                           gdf_column *out_col_indices, gdf_column **out_col_values,
                           gdf_column *out_col_agg, gdf_context *ctxt);
gdf_error gdf_group_by_count(int ncols, gdf_column **cols, gdf_column *col_agg,
                             gdf_column *out_col_indices, gdf_column **out_col_values,
                             gdf_column *out_col_agg, gdf_context *ctxt);

/* --- row hash + hash partition: src/hashing.cu:83-154,559-654 (functions.h:344-378) */
gdf_error gdf_hash(int num_cols, gdf_column **input, gdf_hash_func hash, gdf_column *output);
gdf_error gdf_hash_partition(int num_input_cols, gdf_column *input[], int columns_to_hash[],
                             int num_cols_to_hash, int num_partitions,
                             gdf_column *partitioned_output[], int partition_offsets[],
                             gdf_hash_func hash);

/* --- prefix sum: src/scan.cu:53-76 (functions.h:355-358) ----------------- */
gdf_error gdf_prefixsum_generic(gdf_column *inp, gdf_column *out, int inclusive);
gdf_error gdf_prefixsum_i8(gdf_column *inp, gdf_column *out, int inclusive);
gdf_error gdf_prefixsum_i32(gdf_column *inp, gdf_column *out, int inclusive);
gdf_error gdf_prefixsum_i64(gdf_column *inp, gdf_column *out, int inclusive);

/* --- filter predicates + stream compaction: src/filterops.cu:162-662,
 *     src/streamcompactionops.cu:208-339, src/sqls_ops.cu:1401-1424
 *     (functions.h:677-690,711-725) */
gdf_error gpu_comparison_static_i8 (gdf_column *lhs, int8_t  value, gdf_column *output, gdf_comparison_operator operation);
gdf_error gpu_comparison_static_i16(gdf_column *lhs, int16_t value, gdf_column *output, gdf_comparison_operator operation);
gdf_error gpu_comparison_static_i32(gdf_column *lhs, int32_t value, gdf_column *output, gdf_comparison_operator operation);
gdf_error gpu_comparison_static_i64(gdf_column *lhs, int64_t value, gdf_column *output, gdf_comparison_operator operation);
gdf_error gpu_comparison_static_f32(gdf_column *lhs, float   value, gdf_column *output, gdf_comparison_operator operation);
gdf_error gpu_comparison_static_f64(gdf_column *lhs, double  value, gdf_column *output, gdf_comparison_operator operation);
gdf_error gpu_comparison(gdf_column *lhs, gdf_column *rhs, gdf_column *output, gdf_comparison_operator operation);
gdf_error gpu_apply_stencil(gdf_column *lhs, gdf_column *stencil, gdf_column *output);
gdf_error gdf_filter(size_t nrows, gdf_column *cols, size_t ncols, void **d_cols, int *d_types,
                     void **d_vals, size_t *d_indx, size_t *new_sz);

/* ======================================================================== *
 *  Out-of-scope entry points: exported, return GDF_UNSUPPORTED_METHOD.     *
 * ======================================================================== */
#define GDF_DECL_UNARY(name)        gdf_error name(gdf_column *input, gdf_column *output);
#define GDF_DECL_UNARY_TU(name)     gdf_error name(gdf_column *input, gdf_column *output, gdf_time_unit time_unit);
#define GDF_DECL_BINARY(name)       gdf_error name(gdf_column *lhs, gdf_column *rhs, gdf_column *output);
#define GDF_DECL_REDUCE(name, T)    gdf_error name(gdf_column *col, T *dev_result, gdf_size_type dev_result_size);
#define GDF_DECL_RSORT(name)        gdf_error name(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol);
#define GDF_DECL_SEGSORT(name)      gdf_error name(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol, \
                                                   unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets);
#include "gdf_unsupported.def"
#undef GDF_DECL_UNARY
#undef GDF_DECL_UNARY_TU
#undef GDF_DECL_BINARY
#undef GDF_DECL_REDUCE
#undef GDF_DECL_RSORT
#undef GDF_DECL_SEGSORT

/* one-off shapes (functions.h:108-224,692-705,774-785; io_functions.h) */
gdf_ipc_parser_type *gdf_ipc_parser_open(const uint8_t *schema, size_t length);
void        gdf_ipc_parser_open_recordbatches(gdf_ipc_parser_type *handle, const uint8_t *recordbatches, size_t length);
void        gdf_ipc_parser_close(gdf_ipc_parser_type *handle);
int         gdf_ipc_parser_failed(gdf_ipc_parser_type *handle);
const char *gdf_ipc_parser_to_json(gdf_ipc_parser_type *handle);
const char *gdf_ipc_parser_get_error(gdf_ipc_parser_type *handle);
const void *gdf_ipc_parser_get_data(gdf_ipc_parser_type *handle);
int64_t     gdf_ipc_parser_get_data_offset(gdf_ipc_parser_type *handle);
const char *gdf_ipc_parser_get_schema_json(gdf_ipc_parser_type *handle);
const char *gdf_ipc_parser_get_layout_json(gdf_ipc_parser_type *handle);
/* radix-sort wrappers (reference functions.h:108-224, src/sorting.cu, src/segmented_sorting.cu): csrc/sort.hip */
gdf_error   gdf_radixsort_i8(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol);
gdf_error   gdf_radixsort_i32(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol);
gdf_error   gdf_radixsort_i64(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol);
gdf_error   gdf_radixsort_f32(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol);
gdf_error   gdf_radixsort_f64(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol);
gdf_error   gdf_radixsort_generic(gdf_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol);
gdf_error   gdf_segmented_radixsort_i8(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol, unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets);
gdf_error   gdf_segmented_radixsort_i32(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol, unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets);
gdf_error   gdf_segmented_radixsort_i64(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol, unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets);
gdf_error   gdf_segmented_radixsort_f32(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol, unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets);
gdf_error   gdf_segmented_radixsort_f64(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol, unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets);
gdf_error   gdf_segmented_radixsort_generic(gdf_segmented_radixsort_plan_type *hdl, gdf_column *keycol, gdf_column *valcol, unsigned num_segments, unsigned *d_begin_offsets, unsigned *d_end_offsets);
gdf_radixsort_plan_type *gdf_radixsort_plan(size_t num_items, int descending, unsigned begin_bit, unsigned end_bit);
gdf_error   gdf_radixsort_plan_setup(gdf_radixsort_plan_type *hdl, size_t sizeof_key, size_t sizeof_val);
gdf_error   gdf_radixsort_plan_free(gdf_radixsort_plan_type *hdl);
gdf_segmented_radixsort_plan_type *gdf_segmented_radixsort_plan(size_t num_items, int descending, unsigned begin_bit, unsigned end_bit);
gdf_error   gdf_segmented_radixsort_plan_setup(gdf_segmented_radixsort_plan_type *hdl, size_t sizeof_key, size_t sizeof_val);
gdf_error   gdf_segmented_radixsort_plan_free(gdf_segmented_radixsort_plan_type *hdl);
unsigned int gdf_reduce_optimal_output_size(void);
gdf_error   gpu_concat(gdf_column *lhs, gdf_column *rhs, gdf_column *output);
gdf_error   gpu_hash_columns(gdf_column **columns_to_hash, int num_columns, gdf_column *output_column, void *stream);
gdf_error   gdf_order_by(size_t nrows, gdf_column *cols, size_t ncols, void **d_cols, int *d_types, size_t *d_indx);
gdf_error   gdf_quantile_exact(gdf_column *col_in, gdf_quantile_method prec, double q, void *t_erased_res, gdf_context *ctxt);
gdf_error   gdf_quantile_aprrox(gdf_column *col_in, double q, void *t_erased_res, gdf_context *ctxt);
gdf_error   read_csv(csv_read_arg *args);
gdf_error   gdf_to_csr(gdf_column **gdfData, int num_cols, csr_gdf *csrReturn);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}  /* extern "C" */
#endif
#endif /* GDF_AMD_GDF_H */
