// prof.cpp -- see prof.h
#include "prof.h"
#include "gdf/gdf_amd_ext.h"

#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Pair { std::string name; hipEvent_t a, b; };
std::mutex mu;
bool enabled = false;
std::vector<Pair> pending;
std::map<std::string, std::pair<double, int>> totals;
hipEvent_t cur_start;
std::string cur_name;
std::string cur_tag;

void fold_locked() {
  if (pending.empty()) return;
  (void)hipStreamSynchronize((hipStream_t)0);
  for (auto &p : pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      auto &t = totals[p.name];
      t.first += ms;
      t.second += 1;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  pending.clear();
}
}  // namespace

namespace gdf_amd {
bool prof_enabled() { return enabled; }
void prof_set_tag(const char *tag) {
  std::lock_guard<std::mutex> g(mu);
  cur_tag = tag ? tag : "";
}
void prof_begin(const char *name) {
  std::lock_guard<std::mutex> g(mu);
  cur_name = name;
  cur_name += cur_tag;
  (void)hipEventCreate(&cur_start);
  (void)hipEventRecord(cur_start, (hipStream_t)0);
}
void prof_end() {
  std::lock_guard<std::mutex> g(mu);
  hipEvent_t stop;
  (void)hipEventCreate(&stop);
  (void)hipEventRecord(stop, (hipStream_t)0);
  pending.push_back({cur_name, cur_start, stop});
  if (pending.size() > 4096) fold_locked();
}
}  // namespace gdf_amd

extern "C" {
void gdf_amd_profile_enable(int on) { std::lock_guard<std::mutex> g(mu); enabled = on != 0; }
void gdf_amd_profile_reset(void) {
  std::lock_guard<std::mutex> g(mu);
  fold_locked();
  totals.clear();
}
int gdf_amd_profile_read(char names[][64], double *total_ms, int *launches, int cap) {
  std::lock_guard<std::mutex> g(mu);
  fold_locked();
  int i = 0;
  for (auto &kv : totals) {
    if (i < cap) {
      std::strncpy(names[i], kv.first.c_str(), 63);
      names[i][63] = 0;
      total_ms[i] = kv.second.first;
      launches[i] = kv.second.second;
    }
    ++i;
  }
  return i;
}
}
