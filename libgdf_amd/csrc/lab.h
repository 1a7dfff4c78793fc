// lab.h -- the ONLY place this library consults anything but its arguments.
//
// The shipped libgdf.so reads no environment variable: a stray GDF_* variable in a caller's environment must
// not change which algorithm a drop-in library runs.  Two kinds of switches exist in the sources:
//
//   path(name)  a path selector the parity tests need (the same join / group-by through two different code
//               paths must give the same answer).  Shipped build: set ONLY through the exported test hook
//               gdf_amd_debug_force(name, value) (include/gdf/gdf_amd_ext.h), never from the environment.
//   knob(name)  an experiment knob (tile sizes, ablation bit masks, A/B switches of a tuning session).
//               Shipped build: a compile-time nullptr -- the code behind it folds away.
//
// A LAB build (make lab: -DGDF_AMD_LAB, lib/lab/libgdf.so, never loaded unless LIBGDF_AMD_LAB=1 asks the
// Python binding for it) also reads both kinds from the environment, per call, which is what the tuning
// scripts under tools/gpu/ use.  Kernel-side ablation branches are written `if (LAB_BITS(a.dbg) & 32)`:
// constant 0 in the shipped build.
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>

namespace gdf_amd {
namespace lab {

// plumbing.cpp: value set through gdf_amd_debug_force, or nullptr.  The returned pointer is an interned string that is
// never freed (the library itself never writes the registry).
const char *forced(const char *name);

#ifdef GDF_AMD_LAB
static inline const char *path(const char *name) {
  if (const char *v = forced(name)) return v;
  return std::getenv(name);
}
static inline const char *knob(const char *name) { return path(name); }
#define LAB_BITS(x) (x)
#else
static inline const char *path(const char *name) { return forced(name); }
static inline constexpr const char *knob(const char *) { return nullptr; }
__host__ __device__ static inline constexpr int no_bits() { return 0; }
#define LAB_BITS(x) gdf_amd::lab::no_bits()
#endif

static inline bool path_on(const char *name) { return path(name) != nullptr; }
static inline bool knob_on(const char *name) { return knob(name) != nullptr; }
static inline long long path_int(const char *name, long long dflt) { const char *v = path(name); return v ? atoll(v) : dflt; }
static inline long long knob_int(const char *name, long long dflt) { const char *v = knob(name); return v ? atoll(v) : dflt; }
static inline double knob_float(const char *name, double dflt) { const char *v = knob(name); return v ? atof(v) : dflt; }

}  // namespace lab
}  // namespace gdf_amd
