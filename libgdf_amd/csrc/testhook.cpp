// testhook.cpp -- libgdf_testhook.so: TEST INFRASTRUCTURE, not part of the product.
//
// The shipped libgdf.so reads no environment variable and (round 6) exports no switch either: its alternative code
// paths (csrc/lab.h "path" selectors) are looked up through a WEAK reference to gdf_amd_testhook_forced, which stays null in every
// process that has not loaded THIS library (RTLD_GLOBAL, in front of libgdf.so).  The parity tests that push one request through two code
// paths load it (tests/conftest.py sets LIBGDF_AMD_TESTHOOK=1 for the Python binding, libgdf_amd/_binding.py) and set names through
// gdf_amd_debug_force; nothing else ever does.  include/gdf/gdf_amd_testhook.h declares the two entry points.
#include <map>
#include <mutex>
#include <set>
#include <string>

#include "gdf/gdf_amd_testhook.h"

namespace {
std::mutex g_mutex;
// name -> interned value.  Values are interned in a set that only grows, so a pointer handed out by the lookup stays valid for the
// life of the process even when another thread forces the same name again or clears it (c_str() of a map entry that a
// concurrent gdf_amd_debug_force erased was a use-after-free); a test process sets a handful of distinct values.
std::map<std::string, const char *> &forced_map() { static std::map<std::string, const char *> m; return m; }
const char *intern(const char *value) { static std::set<std::string> pool; return pool.insert(value).first->c_str(); }
int g_count = 0;
}  // namespace

extern "C" {

__attribute__((visibility("default"))) const char *gdf_amd_testhook_forced(const char *name) {
  if (__atomic_load_n(&g_count, __ATOMIC_RELAXED) == 0) return nullptr;       // nothing forced: one relaxed load per lookup
  std::lock_guard<std::mutex> lock(g_mutex);
  auto it = forced_map().find(name);
  return it == forced_map().end() ? nullptr : it->second;
}

__attribute__((visibility("default"))) gdf_error gdf_amd_debug_force(const char *name, const char *value) {
  if (!name) return GDF_INVALID_API_CALL;
  try {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto &m = forced_map();
    if (value) m[name] = intern(value); else m.erase(name);
    __atomic_store_n(&g_count, (int)m.size(), __ATOMIC_RELAXED);
  } catch (...) {
    return GDF_MEMORYMANAGER_ERROR;
  }
  return GDF_SUCCESS;
}

}  // extern "C"
