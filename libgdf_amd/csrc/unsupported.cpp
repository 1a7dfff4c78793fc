// unsupported.cpp -- exported stubs for the reference entry points that are
// outside the relational hot path.  The cffi
// binding resolves symbols lazily (python/libgdf_cffi/wrapper.py:13-34), so a
// caller only meets these when it actually calls one; it then gets
// GDF_UNSUPPORTED_METHOD (or a null handle) instead of a missing-symbol crash.
#include "gdf/gdf.h"

extern "C" {

#define GDF_DECL_UNARY(name)     gdf_error name(gdf_column *, gdf_column *) { return GDF_UNSUPPORTED_METHOD; }
#define GDF_DECL_UNARY_TU(name)  gdf_error name(gdf_column *, gdf_column *, gdf_time_unit) { return GDF_UNSUPPORTED_METHOD; }
#define GDF_DECL_BINARY(name)    gdf_error name(gdf_column *, gdf_column *, gdf_column *) { return GDF_UNSUPPORTED_METHOD; }
#define GDF_DECL_REDUCE(name, T) gdf_error name(gdf_column *, T *, gdf_size_type) { return GDF_UNSUPPORTED_METHOD; }
#define GDF_DECL_RSORT(name)     gdf_error name(gdf_radixsort_plan_type *, gdf_column *, gdf_column *) { return GDF_UNSUPPORTED_METHOD; }
#define GDF_DECL_SEGSORT(name)                                                                        \
  gdf_error name(gdf_segmented_radixsort_plan_type *, gdf_column *, gdf_column *, unsigned, unsigned *, \
                 unsigned *) { return GDF_UNSUPPORTED_METHOD; }
#include "gdf/gdf_unsupported.def"

unsigned int gdf_reduce_optimal_output_size(void) { return 0; }
gdf_error gdf_quantile_exact(gdf_column *, gdf_quantile_method, double, void *, gdf_context *) { return GDF_UNSUPPORTED_METHOD; }
gdf_error gdf_quantile_aprrox(gdf_column *, double, void *, gdf_context *) { return GDF_UNSUPPORTED_METHOD; }
gdf_error read_csv(csv_read_arg *) { return GDF_UNSUPPORTED_METHOD; }
gdf_error gdf_to_csr(gdf_column **, int, csr_gdf *) { return GDF_UNSUPPORTED_METHOD; }

}  // extern "C"
