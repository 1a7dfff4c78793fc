/* gdf_amd_testhook.h -- libgdf_testhook.so (csrc/testhook.cpp): TEST INFRASTRUCTURE, no counterpart in the reference and not part of
 * the drop-in library.  libgdf.so exports no switch that changes which algorithm later calls run; a test process that wants one
 * request pushed through two code paths loads libgdf_testhook.so with RTLD_GLOBAL BEFORE libgdf.so (whose weak reference to
 * gdf_amd_testhook_forced is bound at load time) and names the path here.  tests/conftest.py::force_path is the caller. */
#ifndef GDF_AMD_TESTHOOK_H
#define GDF_AMD_TESTHOOK_H
#include "gdf/gdf.h"
#ifdef __cplusplus
extern "C" {
#endif
/* force one of the library's alternative code paths (csrc/lab.h "path" switches, e.g. "GDF_JK_NO_SPEC") for the calls that follow in
 * this process; value NULL clears the name */
gdf_error gdf_amd_debug_force(const char *name, const char *value);
/* what libgdf.so asks: the value forced for `name`, or NULL */
const char *gdf_amd_testhook_forced(const char *name);
#ifdef __cplusplus
}
#endif
#endif
