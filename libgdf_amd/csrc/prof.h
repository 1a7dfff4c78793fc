// prof.h -- opt-in per-kernel timing with HIP events on the library's own stream.
//
// bench.py needs the average duration of the dominant kernel measured live, on the
// stream the kernel is launched on (the legacy default stream here).  When
// gdf_amd_profile_enable(1) has been called, every GDF_LAUNCH brackets the launch
// with a hipEvent pair; gdf_amd_profile_read() synchronises, folds the pairs into
// per-name totals and reports them.  Disabled (the default) it costs one branch.
#pragma once
#include <hip/hip_runtime.h>

namespace gdf_amd {
bool prof_enabled();
void prof_begin(const char *name);
void prof_end();
// appended to the names of the launches that follow ("" / nullptr: none): the join tags its PROBE-side launches "@probe", so that
// bench.py can price the probe phase (16 B per probe row over the probe-side kernels) apart from the build side, which
// runs the same kernels
void prof_set_tag(const char *tag);
struct ProfTag {
  bool on;
  explicit ProfTag(const char *tag) : on(prof_enabled()) { if (on) prof_set_tag(tag); }
  ~ProfTag() { if (on) prof_set_tag(nullptr); }
};
}  // namespace gdf_amd

#define GDF_LAUNCH(name, ...)                                   \
  do {                                                          \
    const bool _p = gdf_amd::prof_enabled();                    \
    if (_p) gdf_amd::prof_begin(name);                          \
    hipLaunchKernelGGL(__VA_ARGS__);                            \
    if (_p) gdf_amd::prof_end();                                \
  } while (0)

extern "C" {
#pragma GCC visibility push(default)
// extra (non-reference) exports, used only by bench.py / profiling scripts
void gdf_amd_profile_enable(int on);
void gdf_amd_profile_reset(void);
// writes up to `cap` records; returns the number of distinct kernel names
int gdf_amd_profile_read(char names[][64], double *total_ms, int *launches, int cap);
#pragma GCC visibility pop
}
