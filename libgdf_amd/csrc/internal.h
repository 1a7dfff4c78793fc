// internal.h -- host-side entry points shared between the translation units of
// libgdf.so (not exported: the Makefile builds with -fvisibility=hidden and only
// the extern "C" ABI of include/gdf/gdf.h is marked default).
#pragma once
#include "common.h"
#include "prof.h"

namespace gdf_amd {

// scan.hip: device-wide prefix sums (in == out allowed)
gdf_error scan_u32(const uint32_t *in, uint32_t *out, size_t n, bool inclusive);
gdf_error scan_u64(const uint64_t *in, uint64_t *out, size_t n, bool inclusive);

}  // namespace gdf_amd
