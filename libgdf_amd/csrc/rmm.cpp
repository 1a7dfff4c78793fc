// rmm.cpp -- librmm.so: the device-memory manager behind include/memory.h.
//
// Same C ABI and error behaviour as the reference's src/memory/memory.cpp:138-298
// (+ memory_manager.cpp:31-70 for the CSV event log), re-built on HIP:
//   * CudaDefaultAllocation -> hipMalloc / hipFree per call;
//   * PoolAllocation        -> a caching allocator: freed blocks are kept in a
//     size-ordered free list and handed back to the next request that fits
//     (the reference used cnmem, an absent submodule).  A relational call
//     allocates and frees multi-GB scratch (partition buffers, join indices);
//     with 288 GB of HBM per GPU holding on to those blocks is the right trade.
// Thread safety: one mutex around the pool and one around the log, as the
// reference's Manager (memory_manager.h:119-144).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <string>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <set>
#include <sstream>
#include <unordered_map>
#include <vector>

#include "memory.h"

namespace {

using Clock = std::chrono::system_clock;

struct Event {
  int kind;  // 0 alloc, 1 realloc, 2 free
  int device;
  void *ptr;
  size_t size;
  void *stream;
  size_t free_mem, total_mem, live;
  Clock::time_point t0, t1;
};

struct Manager {
  std::mutex mu;
  rmmOptions_t opt{CudaDefaultAllocation, 0, false};
  bool initialized = false;

  // pool state
  std::multimap<size_t, void *> free_blocks;        // capacity -> block
  std::unordered_map<void *, size_t> live_blocks;   // block -> capacity
  size_t cached_bytes = 0, live_bytes = 0;

  // placed blocks (gdf_amd_rmm_place_*): see the comment above place_alloc
  struct Placed {
    int role = 0;
    size_t want = 0;
    void *champ = nullptr, *chall = nullptr;
    float champ_ms = -1.f;
    int draws = 0;                 // challengers drawn so far
    int max_draws = 0;             // ... of at most this many (the last caller's word)
    double explore_ms = 0;         // host time the challengers' hipMalloc calls have cost so far: exploration stops at PLACE_BUDGET_MS
    float worst_ms = 0.f;          // the slowest candidate timed so far (EARLY SETTLE: see place_free)
    double explore_max = 0;        // the single most expensive of those hipMalloc calls: not counted against the budget (see place_alloc)
    int stale = 0;                 // allocations in a row that found losers held and drew nothing (PLACE_STALE_CALLS)
    bool busy = false;             // a block of this entry is out with a caller
    std::vector<void *> losers;    // held until the exploration ends: a freed loser's pages would come straight back as the next draw
    unsigned long long stamp = 0;  // last use, for eviction
  };
  std::vector<Placed> placed;
  int place_draws = 4;
  unsigned long long place_clock = 0;
  size_t placed_idle_bytes = 0;     // champions not in use: available to the next call of their role, counted as free by rmmGetInfo
  size_t placed_loser_bytes = 0;    // losers a running search holds: given back when memory is short, counted as free likewise
  // counters for tests / profiles (gdf_amd_rmm_place_stats)
  unsigned long long place_drawn = 0, place_promoted = 0;
  std::string place_trace;          // one line per decision, bounded (gdf_amd_rmm_place_trace)

  // log state
  std::mutex log_mu;
  std::vector<Event> events;
  std::set<void *> current;
  Clock::time_point base = Clock::now();

  static Manager &get() { static Manager m; return m; }
};

inline bool pool_mode(Manager &m) { return m.opt.allocation_mode == PoolAllocation; }

// round requests so that near-equal sizes recycle the same cached blocks
inline size_t round_size(size_t n) {
  if (n < 256) return 256;
  if (n < (1u << 20)) return (n + 255) & ~size_t(255);          // 256 B granules below 1 MiB
  return (n + ((size_t(1) << 20) - 1)) & ~((size_t(1) << 20) - 1);  // 1 MiB granules above
}

rmmError_t map_hip(hipError_t e) {
  if (e == hipSuccess) return RMM_SUCCESS;
  (void)hipGetLastError();   // do not leave a sticky error behind
  return e == hipErrorOutOfMemory ? RMM_ERROR_OUT_OF_MEMORY : RMM_ERROR_CUDA_ERROR;
}

// PHYSICALLY CONTIGUOUS pool blocks -- an experiment that LOST, kept behind gdf_amd_rmm_contiguous(1) for the record.  The join's regroup
// kernels keep thousands of write fronts open across a multi-GB scratch buffer (jk_scatter1: 16384 regions over 6.3 GB), and with plain
// hipMalloc the same kernel on the same VIRTUAL addresses runs in one of two modes -- 2.85 or 3.25 ms for C3's probe side -- decided anew
// by every re-allocation of the buffer (profiles/r4_q_scatter1_modes_reallocation.jsonl): what changes is the physical backing the driver
// hands out.  Asking for ONE physical range (hipDeviceMallocContiguous) does not pick the fast mode: every kernel that writes short runs
// gets much slower (C3 9.0 - 9.9 -> 13.4 - 14.8 ms per join, jk_scatter2 2.7 -> 5.0 - 5.3, profiles/r4_q_contiguous_scratch_ab.jsonl), and
// the allocation itself can take seconds.
int g_contiguous = 0;
constexpr size_t CONTIGUOUS_MIN = size_t(64) << 20;
hipError_t pool_hip_malloc(void **p, size_t want) {
  if (__atomic_load_n(&g_contiguous, __ATOMIC_RELAXED) && want >= CONTIGUOUS_MIN) {
    if (hipExtMallocWithFlags(p, want, hipDeviceMallocContiguous) == hipSuccess) return hipSuccess;
    (void)hipGetLastError();
    *p = nullptr;
  }
  return hipMalloc(p, want);
}

void release_cache_locked(Manager &m) {
  for (auto &kv : m.free_blocks) (void)hipFree(kv.second);
  m.free_blocks.clear();
  m.cached_bytes = 0;
}

bool place_release_idle_locked(Manager &m);      // (placed blocks, below)

rmmError_t pool_alloc(Manager &m, void **ptr, size_t size) {
  const size_t want = round_size(size);
  std::lock_guard<std::mutex> g(m.mu);
  auto it = m.free_blocks.lower_bound(want);
  // accept a cached block only if it wastes at most a fifth of itself.  (Up to half was accepted at first: a 4 GB request
  // then took the 7.6 GB block another buffer of the same call needs a moment later, that one went to hipMalloc -- milliseconds
  // for a block of this size -- and a join's steady state took several calls of such swaps to settle, if it ever did:
  // tools/bench_shapes.py saw 13 - 16 ms per C3 join around 10.6 ms of kernels.)
  // The tight window is for the multi-GB blocks that made that trouble; below 1 GiB a block up to twice the request is taken
  // (slices of varying size, speculative capacities: a 25 % window sent most of them to hipMalloc and let the misfits pile up).
  const size_t window = want >= (size_t(1) << 30) ? want + want / 4 : 2 * want;
  if (it != m.free_blocks.end() && it->first <= window) {
    *ptr = it->second;
    m.live_blocks[*ptr] = it->first;
    m.cached_bytes -= it->first;
    m.live_bytes += it->first;
    m.free_blocks.erase(it);
    return RMM_SUCCESS;
  }
  // a long-running process whose request sizes drift must not grow the cache without bound: past 64 GiB of cached blocks
  // (more than any single relational call of the benchmarks keeps) the cache is returned to the runtime before growing further
  if (m.cached_bytes > (size_t(64) << 30)) release_cache_locked(m);
  void *p = nullptr;
  hipError_t e = pool_hip_malloc(&p, want);
  if (e == hipErrorOutOfMemory) {       // give cached blocks back and retry once
    (void)hipGetLastError();
    release_cache_locked(m);
    e = pool_hip_malloc(&p, want);
  }
  if (e == hipErrorOutOfMemory && place_release_idle_locked(m)) {      // ... and the placed blocks nobody is using
    (void)hipGetLastError();
    e = pool_hip_malloc(&p, want);
  }
  if (e != hipSuccess) return map_hip(e);
  m.live_blocks[p] = want;
  m.live_bytes += want;
  *ptr = p;
  return RMM_SUCCESS;
}

rmmError_t pool_free(Manager &m, void *ptr) {
  if (!ptr) return RMM_SUCCESS;
  std::lock_guard<std::mutex> g(m.mu);
  auto it = m.live_blocks.find(ptr);
  if (it == m.live_blocks.end()) return RMM_ERROR_INVALID_ARGUMENT;   // not ours
  m.free_blocks.emplace(it->second, ptr);
  m.cached_bytes += it->second;
  m.live_bytes -= it->second;
  m.live_blocks.erase(it);
  return RMM_SUCCESS;
}

// PLACED BLOCKS -- a pool that re-draws slow physical placements.
//
// The join's regroup kernels keep thousands of write fronts open across a multi-GB scratch block, and the SAME kernel on the SAME virtual
// addresses runs in one of two modes (C3's probe side: jk_scatter1 3.20 or 3.65 ms), decided anew by every hipMalloc of the block: what
// changes is the physical backing the driver hands out (profiles/r4_q_scatter1_modes_reallocation.jsonl; nothing in user space selects
// it, a physically contiguous range is far worse).  So the pool lets the caller say what a block is FOR (a small integer role) and how
// long the kernels that scatter into it took (HIP events on the caller's stream):
//   * the first call of a (role, size) gets a fresh block, the CHAMPION, and reports its time when it gives the block back;
//   * the next `place_draws` allocations (or `max_draws`, the caller's own number) each get a CHALLENGER -- a fresh hipMalloc made while the champion (and every earlier loser) is
//     still held, so that it cannot be the same physical pages -- and the faster of the two stays champion;
//   * after that the champion serves every call, the losers go back to the runtime, and nothing is measured any more.
// A caller that repeats a join shape pays a few multi-GB hipMalloc / hipFree pairs (~2 ms each) over its first calls and then runs on
// the best of `place_draws + 1` placements; a one-off call pays nothing (its block is simply cached here instead of in the free list).
// Only in pool mode, only for blocks of PLACE_MIN bytes and more; anything else falls through to the plain pool.
constexpr size_t PLACE_MIN_DEFAULT = size_t(1) << 30;
size_t g_place_min = PLACE_MIN_DEFAULT;       // (test hook gdf_amd_rmm_place_min: the calibration loops of the callers on small inputs)
constexpr size_t PLACE_MAX_ENTRIES = 8;      // (a WIDE-key join holds five: level 1, its high words, level 2, the two output columns)
// a fresh multi-GB hipMalloc usually takes ~1 ms, but the driver can take SECONDS for one when it has to wait for memory another
// process released a moment ago (profiles/r5_b_place_trace_*.json: 1.8 s for nine of them): the search for a better placement ends
// when its allocations have cost this much
constexpr double PLACE_BUDGET_MS = 60.0;
// EARLY SETTLE: a search ends before its last draw once it has seen both kinds of placement and holds the fast one -- at least
// PLACE_SETTLE_DRAWS challengers drawn and the champion's time at most PLACE_SETTLE_GAIN of the slowest candidate's.  The callers ask for
// up to 16 draws: one fresh block in five is a fast one for the join's level-1 buffer, eight draws missed them all in one process of
// six on some boxes (9.5 instead of 9.05 ms per join), sixteen miss in 3 % -- and most searches end after four to six.
constexpr int PLACE_SETTLE_DRAWS = 4;
constexpr float PLACE_SETTLE_GAIN = 0.93f;       // (0.95 settled on a SLOW level-1 block once: the slow kind alone spans 0.86 - 0.905 ms, a fast block sits at 0.78 - 0.82)
// Round 6: what a search may HOLD.  Round 5 kept every loser until the search settled -- up to 16 blocks
// of 6 GB next to the champion, > 150 GB transient beside 9 GB of inputs.  Now at most PLACE_MAX_LOSERS losers are held (the oldest
// goes back to the runtime right AFTER the next challenger has been drawn, so that draw cannot be its pages again), a challenger
// is only drawn while it is at most a quarter of the free device memory, and an entry whose search makes no progress for
// PLACE_STALE_CALLS allocations gives its losers back.
constexpr size_t PLACE_MAX_LOSERS = 4;       // (fewer, and a search starts drawing its own freed losers again: one fresh level-1 block in five is a fast one)
constexpr int PLACE_STALE_CALLS = 8;

// SIZE CLASSES (round 6): round 5 keyed a champion on the exact (rounded) size, so a caller whose relations change
// size from query to query searched anew every time.  A new entry's block is rounded UP to one of eight steps per octave (at most
// 12.5 % more than asked, on top of an eighth of headroom), and an entry serves every request of its role between six tenths of
// its block and the whole of it (a relation 0.9 ... 1.1 times the first one's size finds its champion whatever the rounding did).
inline size_t place_class(size_t want) {
  size_t oct = size_t(1) << 20;
  while ((oct << 1) <= want) oct <<= 1;
  const size_t step = oct >> 3;
  return (want + step - 1) / step * step;
}
inline bool place_serves(size_t block, size_t want) { return block >= want && want >= block - block / 5 * 2; }      // 0.6 ... 1.0 of the block

void place_note(Manager &m, const char *what, int role, size_t bytes, double ms) {      // one trace line for a slow runtime call (> 5 ms)
  if (ms < 5.0 || m.place_trace.size() >= 16384) return;
  char line[120];
  snprintf(line, sizeof line, "role %d %s of %zu MiB took %.1f ms\n", role, what, bytes >> 20, ms);
  m.place_trace += line;
}
double place_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void place_free_loser_locked(Manager &m, Manager::Placed &e, size_t i) {
  const double t0 = place_now_ms();
  (void)hipFree(e.losers[i]);
  place_note(m, "hipFree(loser)", e.role, e.want, place_now_ms() - t0);
  e.losers.erase(e.losers.begin() + (long)i);
  m.placed_loser_bytes -= e.want;
}
void place_drop_losers_locked(Manager &m, Manager::Placed &e) {
  while (!e.losers.empty()) place_free_loser_locked(m, e, e.losers.size() - 1);
}
void place_drop_entry_locked(Manager &m, Manager::Placed &e) {      // (the entry is not busy)
  place_drop_losers_locked(m, e);
  if (e.chall) { (void)hipFree(e.chall); e.chall = nullptr; }
  if (e.champ) { (void)hipFree(e.champ); e.champ = nullptr; m.placed_idle_bytes -= e.want; }
}
// everything the placed cache can give back without touching a block that is out with a caller; returns whether anything was freed
bool place_release_idle_locked(Manager &m) {
  bool any = false;
  for (auto it = m.placed.begin(); it != m.placed.end();) {
    if (!it->losers.empty()) { place_drop_losers_locked(m, *it); it->draws = 1 << 20; any = true; }      // memory is tight: stop exploring
    if (!it->busy) { if (it->champ) any = true; place_drop_entry_locked(m, *it); it = m.placed.erase(it); }
    else ++it;
  }
  return any;
}

// max_draws > 0: the caller's own number of challengers; 0: the pool's default; < 0: HOLD -- the champion of the class as it stands,
// unmeasured, no challenger is drawn and the search (its losers, its count) stays as it is: what a caller asks for once it has spent
// its per-call time budget on candidates (the search goes on with its next call)
rmmError_t place_alloc(Manager &m, int role, size_t size, int max_draws, void **ptr, int *measure) {
  *measure = 0;
  const size_t want = round_size(size);
  const bool hold = max_draws < 0;
  {
    std::lock_guard<std::mutex> g(m.mu);
    if (pool_mode(m) && want >= g_place_min && m.place_draws >= 0) {
      // (a caller that calibrates candidates inside ONE call can afford more of them than one that spends a call on each)
      const int draws = m.place_draws == 0 ? 0 : (max_draws > 0 ? std::min(max_draws, 16) : m.place_draws);
      Manager::Placed *e = nullptr;
      for (auto &x : m.placed)                // the smallest idle block of this role that serves the request (a busy one only if nothing else does)
        if (x.role == role && place_serves(x.want, want) && (!e || (e->busy && !x.busy) || (e->busy == x.busy && x.want < e->want))) e = &x;
      if (!e) {
        // an eighth of headroom on top, then the class: the first request of a role is served by a block that also holds relations up
        // to ~1.15x its size -- bench.py's sweep over 0.9 ... 1.1e9 probe rows runs on the champions its first call chose
        const size_t block = place_class(want + want / 8);
        // entries of this role the new, larger block serves as well go: their requests are this entry's from now on
        for (auto it = m.placed.begin(); it != m.placed.end();) {
          if (it->role == role && !it->busy && it->want <= block && place_serves(block, it->want)) { place_drop_entry_locked(m, *it); it = m.placed.erase(it); }
          else ++it;
        }
        if (m.placed.size() >= PLACE_MAX_ENTRIES) {          // evict the entry used longest ago (never one that is out)
          size_t victim = m.placed.size();
          for (size_t i = 0; i < m.placed.size(); ++i)
            if (!m.placed[i].busy && (victim == m.placed.size() || m.placed[i].stamp < m.placed[victim].stamp)) victim = i;
          if (victim < m.placed.size()) { place_drop_entry_locked(m, m.placed[victim]); m.placed.erase(m.placed.begin() + victim); }
        }
        if (m.placed.size() < PLACE_MAX_ENTRIES) {
          m.placed.emplace_back();
          e = &m.placed.back();
          e->role = role;
          e->want = block;
        }
      }
      if (e && !e->busy) {
        e->stamp = ++m.place_clock;
        if (!e->champ) {
          void *p = nullptr;
          const double t0 = place_now_ms();
          hipError_t err = hipMalloc(&p, e->want);
          place_note(m, "hipMalloc(champion)", role, e->want, place_now_ms() - t0);
          if (err == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            release_cache_locked(m);
            err = hipMalloc(&p, e->want);
          }
          if (err == hipErrorOutOfMemory) {
            //  the idle champions and the losers of OTHER entries go before this request fails: the plain pool would have
            // had that memory.  The release erases idle entries -- this one among them: the request is the plain pool's now
            (void)hipGetLastError();
            (void)place_release_idle_locked(m);
            e = nullptr;
          } else if (err != hipSuccess) {
            return map_hip(err);
          } else {
            e->champ = p;
            e->champ_ms = -1.f;
            e->worst_ms = 0.f;
            e->explore_ms = 0;
            e->explore_max = 0;
            e->draws = 0;
            e->stale = 0;
            e->busy = true;
            // (an entry that starts life HELD has a search ahead of it: it counts as exploring -- gdf_amd_rmm_place_stats, what a
            // benchmark's warm-up waits for -- until a call with budget left has run that search)
            e->max_draws = hold ? (draws > 0 ? 1 : 0) : draws;
            *ptr = p;
            *measure = !hold && draws > 0;
            return RMM_SUCCESS;
          }
        }
      }
      if (e && !e->busy) {
        m.placed_idle_bytes -= e->want;
        e->busy = true;
        if (!hold) e->max_draws = draws;
        bool drew = false;
        if (!hold && e->champ_ms > 0.f && e->draws < draws) {        // a challenger, drawn while the champion is held
          // ... if the device has room for it: a search holds at most champion + challenger + PLACE_MAX_LOSERS blocks, and a block
          // is drawn only while it is a quarter of the free memory at most
          size_t free_b = 0, total_b = 0;
          const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && e->want <= free_b / 4;
          void *p = nullptr;
          const auto t0 = std::chrono::steady_clock::now();
          const hipError_t drawn = room ? hipMalloc(&p, e->want) : hipErrorOutOfMemory;
          const double took = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
          e->explore_ms += took;
          if (took > e->explore_max) e->explore_max = took;
          place_note(m, "hipMalloc(challenger)", role, e->want, took);
          // the budget forgives ONE stall: behind another process's exit the driver takes seconds for one multi-GB hipMalloc, once per
          // process -- a search that gave up there kept whatever it had drawn so far, a slow block in one process of eight
          // on such a box (9.47 instead of 9.05 ms per join); the allocations after the stall cost 0.3 ms again
          if (drawn == hipSuccess && e->explore_ms - e->explore_max > PLACE_BUDGET_MS) e->draws = draws - 1;      // this one is the last
          if (drawn == hipSuccess) {
            while (e->losers.size() > PLACE_MAX_LOSERS) place_free_loser_locked(m, *e, 0);      // (the new block cannot be these pages)
            e->chall = p;
            ++e->draws;
            e->stale = 0;
            ++m.place_drawn;
            *ptr = p;
            *measure = 1;
            return RMM_SUCCESS;
          }
          (void)hipGetLastError();
          e->draws = draws;                                         // no room for a second block of this size: settle
          place_drop_losers_locked(m, *e);
          drew = true;
        }
        // a search that is not moving (held calls, a caller that never reports times) does not sit on its losers for ever
        if (!drew && !e->losers.empty() && ++e->stale >= PLACE_STALE_CALLS) { place_drop_losers_locked(m, *e); e->draws = 1 << 20; }
        *ptr = e->champ;
        *measure = !hold && e->champ_ms <= 0.f && draws > 0;         // (a champion whose first call could not be timed)
        return RMM_SUCCESS;
      }
    }
  }
  return pool_mode(m) ? pool_alloc(m, ptr, size) : map_hip(hipMalloc(ptr, size));
}

rmmError_t place_free(Manager &m, int role, void *ptr, float ms) {
  if (!ptr) return RMM_SUCCESS;
  {
    std::lock_guard<std::mutex> g(m.mu);
    for (auto &e : m.placed) {
      if (e.role != role || !e.busy || (ptr != e.champ && ptr != e.chall)) continue;
      if (m.place_trace.size() < 16384) {
        char line[160];
        snprintf(line, sizeof line, "role %d MiB %zu %s draw %d ms %.3f champion_ms %.3f hipMalloc_ms_so_far %.1f\n", role, e.want >> 20,
                 ptr == e.chall ? "challenger" : "champion", e.draws, ms, e.champ_ms, e.explore_ms);
        m.place_trace += line;
      }
      if (ms > 0.f && ms > e.worst_ms) e.worst_ms = ms;
      if (ptr == e.chall) {
        // the challenger takes over when it was measurably faster (2 %: the event times of one kernel repeat within ~1 %)
        if (ms > 0.f && e.champ_ms > 0.f && ms < 0.98f * e.champ_ms) {
          e.losers.push_back(e.champ);
          e.champ = e.chall;
          e.champ_ms = ms;
          ++m.place_promoted;
        } else {
          e.losers.push_back(e.chall);
        }
        m.placed_loser_bytes += e.want;
        e.chall = nullptr;
      } else if (ms > 0.f && e.champ_ms <= 0.f) {
        e.champ_ms = ms;        // FIRST-use time against first-use time: a challenger is only ever measured on its first call
      }
      if (e.draws >= PLACE_SETTLE_DRAWS && e.draws < e.max_draws && e.champ_ms > 0.f && e.champ_ms <= PLACE_SETTLE_GAIN * e.worst_ms) {
        if (m.place_trace.size() < 16384) {
          char line[120];
          snprintf(line, sizeof line, "role %d settles early after %d draws: champion %.3f ms, slowest %.3f ms\n", role, e.draws, e.champ_ms, e.worst_ms);
          m.place_trace += line;
        }
        e.draws = e.max_draws;
      }
      if (e.draws >= e.max_draws && !e.losers.empty()) {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t nl = e.losers.size();
        place_drop_losers_locked(m, e);
        if (m.place_trace.size() < 16384) {
          char line[120];
          snprintf(line, sizeof line, "role %d settled: %zu losers freed in %.1f ms\n", role, nl,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
          m.place_trace += line;
        }
      }
      e.busy = false;
      m.placed_idle_bytes += e.want;
      return RMM_SUCCESS;
    }
  }
  if (pool_mode(m)) {
    rmmError_t r = pool_free(m, ptr);
    if (r != RMM_ERROR_INVALID_ARGUMENT) return r;
  }
  return map_hip(hipFree(ptr));
}

struct LogScope {   // mirrors the reference's rmm::LogIt (memory.cpp:52-108)
  Manager &m; int kind; void *ptr; size_t size; void *stream; int dev = 0; Clock::time_point t0;
  LogScope(Manager &m_, int k, void *p, size_t s, void *st) : m(m_), kind(k), ptr(p), size(s), stream(st) {
    if (m.opt.enable_logging) { (void)hipGetDevice(&dev); t0 = Clock::now(); }
  }
  ~LogScope() {
    if (!m.opt.enable_logging) return;
    auto t1 = Clock::now();
    size_t f = 0, t = 0;
    (void)hipMemGetInfo(&f, &t);
    std::lock_guard<std::mutex> g(m.log_mu);
    if (kind == 0) m.current.insert(ptr); else if (kind == 2) m.current.erase(ptr);
    m.events.push_back({kind, dev, ptr, size, stream, f, t, m.current.size(), t0, t1});
  }
};

void write_csv(Manager &m, std::ostream &os) {
  // header string is pinned by the reference's python/tests/test_rmm.py:52
  os << "Event Type,Device ID,Address,Stream,Size (bytes),Free Memory,Total Memory,Current Allocs,Start,End,Elapsed\n";
  std::lock_guard<std::mutex> g(m.log_mu);
  for (auto &e : m.events) {
    const char *name = e.kind == 0 ? "Alloc" : (e.kind == 1 ? "Realloc" : "Free");
    std::chrono::duration<double> a = e.t0 - m.base, b = e.t1 - m.base, d = e.t1 - e.t0;
    os << name << "," << e.device << "," << e.ptr << "," << e.stream << "," << e.size << "," << e.free_mem << ","
       << e.total_mem << "," << e.live << "," << a.count() << "," << b.count() << "," << d.count() << std::endl;
  }
}

}  // namespace

// no exception crosses the C boundary (the bookkeeping containers allocate): a failed host allocation is an out-of-memory answer
template <class F>
static inline rmmError_t rmm_guarded(F &&body) noexcept {
  try {
    return body();
  } catch (...) {
    return RMM_ERROR_OUT_OF_MEMORY;
  }
}

extern "C" {

// A-B hook (not part of the reference's memory.h): pool blocks of 64 MiB and more as physically contiguous allocations (default OFF)
__attribute__((visibility("default"))) void gdf_amd_rmm_contiguous(int on) { __atomic_store_n(&g_contiguous, on ? 1 : 0, __ATOMIC_RELAXED); }

// Placed blocks (see place_alloc): what libgdf.so allocates its multi-GB regroup scratch through.  `measure` tells the caller whether
// the pool wants to hear, on gdf_amd_rmm_place_free, how many milliseconds the kernels that scatter into the block took (< 0: unknown).
__attribute__((visibility("default"))) rmmError_t gdf_amd_rmm_place_alloc(int role, size_t size, int max_draws, void **ptr, int *measure) {
  return rmm_guarded([&]() -> rmmError_t {
  if (!ptr || !measure) return RMM_ERROR_INVALID_ARGUMENT;
  Manager &m = Manager::get();
  LogScope log(m, 0, nullptr, size, nullptr);
  const rmmError_t r = place_alloc(m, role, size ? size : 1, max_draws, ptr, measure);
  if (r == RMM_SUCCESS) log.ptr = *ptr;
  return r;
  });
}
__attribute__((visibility("default"))) rmmError_t gdf_amd_rmm_place_free(int role, void *ptr, float ms) {
  return rmm_guarded([&]() -> rmmError_t {
  Manager &m = Manager::get();
  LogScope log(m, 2, ptr, 0, nullptr);
  return place_free(m, role, ptr, ms);
  });
}
// test hook: the size from which a request is a placed block (0: the default, 1 GiB) -- the callers' calibration loops on small inputs
__attribute__((visibility("default"))) void gdf_amd_rmm_place_min(size_t bytes) {
  Manager &m = Manager::get();
  std::lock_guard<std::mutex> g(m.mu);
  g_place_min = bytes ? bytes : PLACE_MIN_DEFAULT;
}
// challengers drawn per (role, size); 0: placed blocks are cached but never re-drawn; < 0: the plain pool serves placed requests
__attribute__((visibility("default"))) void gdf_amd_rmm_place_draws(int draws) {
  Manager &m = Manager::get();
  std::lock_guard<std::mutex> g(m.mu);
  m.place_draws = draws > 16 ? 16 : draws;
}
// the decisions so far, one text line each; returns the length needed (incl. the terminating 0)
__attribute__((visibility("default"))) size_t gdf_amd_rmm_place_trace(char *buf, size_t cap) {
  Manager &m = Manager::get();
  std::lock_guard<std::mutex> g(m.mu);
  if (buf && cap) {
    const size_t n = std::min(cap - 1, m.place_trace.size());
    std::memcpy(buf, m.place_trace.data(), n);
    buf[n] = 0;
  }
  return m.place_trace.size() + 1;
}
// out[0] challengers drawn, out[1] challengers promoted, out[2] entries, out[3] entries still exploring
__attribute__((visibility("default"))) void gdf_amd_rmm_place_stats(unsigned long long out[4]) {
  Manager &m = Manager::get();
  std::lock_guard<std::mutex> g(m.mu);
  out[0] = m.place_drawn;
  out[1] = m.place_promoted;
  out[2] = m.placed.size();
  out[3] = 0;
  for (auto &e : m.placed) out[3] += e.draws < e.max_draws;
}

rmmError_t rmmInitialize(rmmOptions_t *options) {
  return rmm_guarded([&]() -> rmmError_t {
  Manager &m = Manager::get();
  std::lock_guard<std::mutex> g(m.mu);
  if (options) m.opt = *options;
  m.initialized = true;
  if (pool_mode(m) && m.opt.initial_pool_size > 0) {
    // pre-warm the cache with one block of the requested size (cnmem reserved it up front)
    void *p = nullptr;
    const size_t want = round_size(m.opt.initial_pool_size);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return map_hip(e);
    m.free_blocks.emplace(want, p);
    m.cached_bytes += want;
  }
  return RMM_SUCCESS;
  });
}

rmmError_t rmmFinalize(void) {
  return rmm_guarded([&]() -> rmmError_t {
  Manager &m = Manager::get();
  {
    std::lock_guard<std::mutex> g(m.mu);
    release_cache_locked(m);
    for (auto &e : m.placed) {                                // placed blocks, in use or not, go with the pool
      place_drop_losers_locked(m, e);
      if (e.chall) (void)hipFree(e.chall);
      if (e.champ) (void)hipFree(e.champ);
    }
    m.placed.clear();
    m.placed_idle_bytes = 0;
    m.placed_loser_bytes = 0;
    for (auto &kv : m.live_blocks) (void)hipFree(kv.first);   // leaked by the caller; the pool dies with us
    m.live_blocks.clear();
    m.live_bytes = 0;
    m.initialized = false;
  }
  std::lock_guard<std::mutex> g(m.log_mu);
  m.events.clear();
  m.current.clear();
  return RMM_SUCCESS;
  });
}

const char *rmmGetErrorString(rmmError_t errcode) {
  switch (errcode) {
    case RMM_SUCCESS: return "RMM_SUCCESS";
    case RMM_ERROR_CUDA_ERROR: return "RMM_ERROR_CUDA_ERROR";
    case RMM_ERROR_INVALID_ARGUMENT: return "RMM_ERROR_INVALID_ARGUMENT";
    case RMM_ERROR_NOT_INITIALIZED: return "RMM_ERROR_NOT_INITIALIZED";
    case RMM_ERROR_OUT_OF_MEMORY: return "RMM_ERROR_OUT_OF_MEMORY";
    case RMM_ERROR_UNKNOWN: return "RMM_ERROR_UNKNOWN";
    case RMM_ERROR_IO: return "RMM_ERROR_IO";
    default: return "Internal error. Unknown error code.";
  }
}

rmmError_t rmmAlloc(void **ptr, size_t size, cudaStream_t stream) {
  return rmm_guarded([&]() -> rmmError_t {
  Manager &m = Manager::get();
  LogScope log(m, 0, nullptr, size, stream);
  if (!ptr && !size) return RMM_SUCCESS;
  if (!ptr) return RMM_ERROR_INVALID_ARGUMENT;
  rmmError_t r;
  if (pool_mode(m)) r = pool_alloc(m, ptr, size);
  else r = map_hip(hipMalloc(ptr, size));
  if (r == RMM_SUCCESS) log.ptr = *ptr;
  return r;
  });
}

rmmError_t rmmFree(void *ptr, cudaStream_t stream) {
  return rmm_guarded([&]() -> rmmError_t {
  Manager &m = Manager::get();
  LogScope log(m, 2, ptr, 0, stream);
  {
    // a placed block that comes back through the plain entry point (handed on by its owner): no measurement
    int role = -1;
    {
      std::lock_guard<std::mutex> g(m.mu);
      for (auto &e : m.placed) if (e.busy && ptr && (ptr == e.champ || ptr == e.chall)) role = e.role;
    }
    if (role >= 0) return place_free(m, role, ptr, -1.f);
  }
  if (pool_mode(m)) {
    rmmError_t r = pool_free(m, ptr);
    if (r != RMM_ERROR_INVALID_ARGUMENT) return r;
    // a pointer allocated before the pool was switched on: release it directly
  }
  return map_hip(hipFree(ptr));
  });
}

rmmError_t rmmRealloc(void **ptr, size_t new_size, cudaStream_t stream) {
  return rmm_guarded([&]() -> rmmError_t {
  Manager &m = Manager::get();
  LogScope log(m, 1, nullptr, new_size, stream);
  if (!ptr && !new_size) return RMM_SUCCESS;
  if (!ptr) return RMM_ERROR_INVALID_ARGUMENT;
  // like the reference (memory.cpp:207-231): free then allocate, contents are NOT preserved
  rmmError_t r = pool_mode(m) ? pool_free(m, *ptr) : map_hip(hipFree(*ptr));
  if (r == RMM_ERROR_INVALID_ARGUMENT && pool_mode(m)) r = map_hip(hipFree(*ptr));
  if (r != RMM_SUCCESS) return r;
  r = pool_mode(m) ? pool_alloc(m, ptr, new_size) : map_hip(hipMalloc(ptr, new_size));
  if (r == RMM_SUCCESS) log.ptr = *ptr;
  return r;
  });
}

rmmError_t rmmGetAllocationOffset(offset_t *offset, void *ptr, cudaStream_t) {
  return rmm_guarded([&]() -> rmmError_t {
  if (!offset) return RMM_ERROR_INVALID_ARGUMENT;
  hipDeviceptr_t base = nullptr;
  size_t extent = 0;
  if (hipMemGetAddressRange(&base, &extent, (hipDeviceptr_t)ptr) != hipSuccess) {
    (void)hipGetLastError();
    return RMM_ERROR_INVALID_ARGUMENT;
  }
  *offset = (offset_t)((char *)ptr - (char *)base);
  return RMM_SUCCESS;
  });
}

rmmError_t rmmGetInfo(size_t *freeSize, size_t *totalSize, cudaStream_t) {
  return rmm_guarded([&]() -> rmmError_t {
  if (!freeSize || !totalSize) return RMM_ERROR_INVALID_ARGUMENT;
  Manager &m = Manager::get();
  hipError_t e = hipMemGetInfo(freeSize, totalSize);
  if (e != hipSuccess) return map_hip(e);
  if (pool_mode(m)) {   // cached blocks are available to the next rmmAlloc
    std::lock_guard<std::mutex> g(m.mu);
    *freeSize += m.cached_bytes + m.placed_idle_bytes + m.placed_loser_bytes;
  }
  return RMM_SUCCESS;
  });
}

rmmError_t rmmWriteLog(const char *filename) {
  return rmm_guarded([&]() -> rmmError_t {
  if (!filename) return RMM_ERROR_IO;
  std::ofstream f(filename);
  if (!f.good()) return RMM_ERROR_IO;
  write_csv(Manager::get(), f);
  return f.good() ? RMM_SUCCESS : RMM_ERROR_IO;
  });
}

size_t rmmLogSize(void) {
  std::ostringstream s;
  write_csv(Manager::get(), s);
  return s.str().size();
}

rmmError_t rmmGetLog(char *buffer, size_t buffer_size) {
  return rmm_guarded([&]() -> rmmError_t {
  if (!buffer) return RMM_ERROR_INVALID_ARGUMENT;
  std::ostringstream s;
  write_csv(Manager::get(), s);
  const std::string str = s.str();
  std::memcpy(buffer, str.data(), std::min(buffer_size, str.size()));
  return RMM_SUCCESS;
  });
}

}  // extern "C"
