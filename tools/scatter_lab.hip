// scatter_lab.hip -- standalone experiment: can the PROBE side of the join be radix-partitioned in ONE pass with a
// fan-out of 2^11 .. 2^14 (instead of two <= 256-way LDS-regrouped passes), relying on the per-XCD L2 to merge the
// short (tile, bin) runs into whole lines?  (VERDICT r1 "next" item 3.)  Not part of libgdf.so.
//
//   build:  hipcc -O3 --offload-arch=gfx950 tools/scatter_lab.hip -o tools/scatter_lab
//   run:    tools/scatter_lab [rows=1000000000] [variant ...]
//
// Input: rows int64 keys = splitmix64(i) % 1e8 (the C3 probe column).  Output tuples key32 << 32 | row, laid out in
// regions of `cap` tuples per (bin, XCD) with one fill counter each (the speculative layout of join.hip).
// Variants (all read 8 B and write 8 B per row):
//   D<fb>  DIRECT: rank by LDS atomic, one claim per (tile, non-empty bin), tuples stored from registers
//   R<fb>  REGROUP: the same, but the tile is regrouped by bin in LDS first (lanes of a wave store consecutive addresses)
// suffix a = claims with agent-scope atomics (default), w = workgroup-scope atomics (executed in the XCD's own L2;
//            regions are selected by the PHYSICAL XCC id, so every user of a counter sits behind that L2),
//        n = no tuple stores (claims only),  t<k> = threads (512/1024), i<k> = items per thread
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du;
  x ^= x >> 15; x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}
__device__ __forceinline__ void block_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
}
__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 7u;
}

__global__ void gen_keys(uint64_t *k, uint32_t n, uint64_t space) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint64_t z = (uint64_t)i + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k[i] = (z >> 1) % space;
  }
}

struct Params {
  const uint64_t *keys;
  uint32_t n;
  uint64_t *out;
  uint32_t *cursor;      // [F * 8]
  uint32_t cap;          // tuples per (bin, xcd) region
  uint32_t chunk;        // rows per workgroup (multiple of the tile)
  uint32_t *flag;
  uint32_t dump;         // first tuple of a TILE-sized dump area for runs that outgrow their region
  int wg_scope, no_store;
};

template <bool WG>
__device__ __forceinline__ uint32_t claim(uint32_t *p, uint32_t v) {
  return WG ? __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
            : __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- DIRECT -------------------------------------------------------------------------------------------------------
template <int FB, int THREADS, int ITEMS, bool WG>
__global__ __launch_bounds__(THREADS) void scat_direct(Params p) {
  constexpr uint32_t F = 1u << FB;
  constexpr uint32_t TILE = THREADS * ITEMS;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint32_t *hist = lds, *gbase = lds + F + 4;
  const uint32_t xcd = WG ? xcc_id() : (blockIdx.x & 7u);
  const uint32_t begin = blockIdx.x * p.chunk;
  const uint32_t end = begin + p.chunk < p.n ? begin + p.chunk : p.n;
  for (uint32_t b = threadIdx.x; b < F; b += THREADS) hist[b] = 0;
  block_sync();
  for (uint32_t tile = begin; tile < end; tile += TILE) {
    uint64_t raw[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k += 2) {       // row pairs: one 16-byte load each
      const uint32_t i = tile + 2u * ((uint32_t)(k >> 1) * THREADS + threadIdx.x);
      const uint32_t ic = i + 2 <= end ? i : end - 2;
      const uint64_t *q = p.keys + ic;
      raw[k] = __builtin_nontemporal_load(q);
      raw[k + 1] = __builtin_nontemporal_load(q + 1);
    }
    uint32_t key[ITEMS], br[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t i = tile + 2u * ((uint32_t)(k >> 1) * THREADS + threadIdx.x) + (k & 1);
      key[k] = (uint32_t)raw[k];
      const uint32_t bin = lowbias32(key[k]) >> (32 - FB);
      br[k] = i < end ? bin : F;            // F = trash counter
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t r = atomicAdd(&hist[br[k]], 1u);
      br[k] = (br[k] << 16) | r;            // FB <= 15, rank < 2^16
    }
    block_sync();
    for (uint32_t b = threadIdx.x; b < F; b += THREADS) {
      const uint32_t cnt = hist[b];
      if (cnt) {
        const uint32_t region = b * 8u + xcd;
        const uint32_t base = claim<WG>(&p.cursor[region], cnt);
        if (base + cnt > p.cap) { *p.flag = 1; gbase[b] = p.dump; }
        else gbase[b] = region * p.cap + base;
        hist[b] = 0;
      }
    }
    block_sync();
    if (!p.no_store) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        const uint32_t i = tile + 2u * ((uint32_t)(k >> 1) * THREADS + threadIdx.x) + (k & 1);
        const uint32_t bin = br[k] >> 16;
        if (bin < F) {
          p.out[gbase[bin] + (br[k] & 0xffffu)] = ((uint64_t)key[k] << 32) | i;
        }
      }
    }
    // gbase is rewritten only after the next tile's first barrier, which every wave reaches after its stores were issued
  }
}

// ---- REGROUP ------------------------------------------------------------------------------------------------------
template <int FB, int THREADS, int ITEMS, bool WG>
__global__ __launch_bounds__(THREADS) void scat_regroup(Params p) {
  constexpr uint32_t F = 1u << FB;
  constexpr uint32_t TILE = THREADS * ITEMS;
  constexpr uint32_t PER = F / THREADS > 0 ? F / THREADS : 1;      // bins per thread in the scan
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  uint64_t *tw = (uint64_t *)lds;                 // [TILE + 1]
  uint32_t *hist = lds + 2 * (TILE + 2);          // [F + 1] counts, then exclusive starts
  uint32_t *gbase = hist + F + 4;                 // [F] global base - start
  uint32_t *wave_tot = gbase + F;                 // [THREADS / 64]
  const uint32_t xcd = WG ? xcc_id() : (blockIdx.x & 7u);
  const uint32_t begin = blockIdx.x * p.chunk;
  const uint32_t end = begin + p.chunk < p.n ? begin + p.chunk : p.n;
  for (uint32_t b = threadIdx.x; b <= F; b += THREADS) hist[b] = 0;
  block_sync();
  for (uint32_t tile = begin; tile < end; tile += TILE) {
    uint64_t raw[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; k += 2) {
      const uint32_t i = tile + 2u * ((uint32_t)(k >> 1) * THREADS + threadIdx.x);
      const uint32_t ic = i + 2 <= end ? i : end - 2;
      const uint64_t *q = p.keys + ic;
      raw[k] = __builtin_nontemporal_load(q);
      raw[k + 1] = __builtin_nontemporal_load(q + 1);
    }
    uint32_t key[ITEMS], br[ITEMS];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t i = tile + 2u * ((uint32_t)(k >> 1) * THREADS + threadIdx.x) + (k & 1);
      key[k] = (uint32_t)raw[k];
      const uint32_t bin = lowbias32(key[k]) >> (32 - FB);
      br[k] = i < end ? bin : F;
    }
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t r = atomicAdd(&hist[br[k]], 1u);
      br[k] = (br[k] << 16) | r;
    }
    block_sync();
    // claim + exclusive scan of the F counts (thread t owns bins [t * PER, (t + 1) * PER))
    uint32_t cnt[PER], gb[PER], sum = 0;
    const uint32_t b0 = threadIdx.x * PER;
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) {
      cnt[j] = b0 + j < F ? hist[b0 + j] : 0;
      gb[j] = 0;
      if (cnt[j]) {
        const uint32_t region = (b0 + j) * 8u + xcd;
        const uint32_t base = claim<WG>(&p.cursor[region], cnt[j]);
        if (base + cnt[j] > p.cap) { *p.flag = 1; gb[j] = p.dump; }
        else gb[j] = region * p.cap + base;
      }
      sum += cnt[j];
    }
    uint32_t incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const uint32_t nb = __shfl_up(incl, o, 64);
      if ((int)(threadIdx.x & 63) >= o) incl += nb;
    }
    if ((threadIdx.x & 63) == 63) wave_tot[threadIdx.x >> 6] = incl;
    block_sync();
    uint32_t woff = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) if (w < (int)(threadIdx.x >> 6)) woff += wave_tot[w];
    uint32_t run = woff + incl - sum;
#pragma unroll
    for (uint32_t j = 0; j < PER; ++j) {
      if (b0 + j < F) { hist[b0 + j] = run; gbase[b0 + j] = gb[j] - run; }
      run += cnt[j];
    }
    block_sync();
#pragma unroll
    for (int k = 0; k < ITEMS; ++k) {
      const uint32_t i = tile + 2u * ((uint32_t)(k >> 1) * THREADS + threadIdx.x) + (k & 1);
      const uint32_t bin = br[k] >> 16;
      const uint32_t pos = bin < F ? hist[bin] + (br[k] & 0xffffu) : TILE;
      tw[pos] = ((uint64_t)key[k] << 32) | i;
    }
    block_sync();
    const uint32_t total = end - tile < TILE ? end - tile : TILE;
    if (!p.no_store) {
#pragma unroll
      for (int k = 0; k < ITEMS; ++k) {
        const uint32_t j = threadIdx.x + k * THREADS;
        const uint64_t w = tw[j < total ? j : 0];
        const uint32_t bin = lowbias32((uint32_t)(w >> 32)) >> (32 - FB);
        if (j < total) p.out[gbase[bin] + j] = w;
      }
    }
    block_sync();
    for (uint32_t b = threadIdx.x; b <= F; b += THREADS) hist[b] = 0;
    block_sync();
  }
}

// ---- checks ---------------------------------------------------------------------------------------------------------
__global__ void check_regions(const uint64_t *out, const uint32_t *cursor, uint32_t nregions, uint32_t cap, int fb,
                              unsigned long long *acc) {
  unsigned long long ksum = 0, rsum = 0, bad = 0, cnt = 0;
  for (uint32_t r = blockIdx.x; r < nregions; r += gridDim.x) {
    const uint32_t fill = cursor[r] < cap ? cursor[r] : cap;
    const uint32_t bin = r >> 3;
    for (uint32_t i = threadIdx.x; i < fill; i += blockDim.x) {
      const uint64_t w = out[(size_t)r * cap + i];
      ksum += w >> 32; rsum += (uint32_t)w; ++cnt;
      if ((lowbias32((uint32_t)(w >> 32)) >> (32 - fb)) != bin) ++bad;
    }
  }
  atomicAdd(&acc[0], ksum); atomicAdd(&acc[1], rsum); atomicAdd(&acc[2], bad); atomicAdd(&acc[3], cnt);
}
__global__ void check_input(const uint64_t *keys, uint32_t n, unsigned long long *acc) {
  unsigned long long ksum = 0, rsum = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) { ksum += (uint32_t)keys[i]; rsum += i; }
  atomicAdd(&acc[4], ksum); atomicAdd(&acc[5], rsum);
}

template <int FB, int THREADS, int ITEMS, bool WG>
static void launch(bool regroup, Params p, int grid, hipStream_t s) {
  constexpr uint32_t F = 1u << FB;
  if (regroup) {
    const size_t lds = 8 * (size_t)(THREADS * ITEMS + 2) + 4 * (size_t)(2 * F + 4 + THREADS / 64 + 4);
    CHECK(hipFuncSetAttribute((const void *)scat_regroup<FB, THREADS, ITEMS, WG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    scat_regroup<FB, THREADS, ITEMS, WG><<<grid, THREADS, lds, s>>>(p);
  } else {
    const size_t lds = 4 * (size_t)(2 * F + 8);
    CHECK(hipFuncSetAttribute((const void *)scat_direct<FB, THREADS, ITEMS, WG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    scat_direct<FB, THREADS, ITEMS, WG><<<grid, THREADS, lds, s>>>(p);
  }
}

template <int FB, int THREADS, int ITEMS>
static void launch_s(bool regroup, bool wg, Params p, int grid, hipStream_t s) {
  if (wg) launch<FB, THREADS, ITEMS, true>(regroup, p, grid, s);
  else launch<FB, THREADS, ITEMS, false>(regroup, p, grid, s);
}

template <int THREADS, int ITEMS>
static bool launch_fb(int fb, bool regroup, bool wg, Params p, int grid, hipStream_t s) {
  constexpr size_t tile_bytes = 8 * (size_t)THREADS * ITEMS;
  switch (fb) {
    case 8: launch_s<8, THREADS, ITEMS>(regroup, wg, p, grid, s); return true;
    case 10: launch_s<10, THREADS, ITEMS>(regroup, wg, p, grid, s); return true;
    case 11: launch_s<11, THREADS, ITEMS>(regroup, wg, p, grid, s); return true;
    case 12: if (regroup && tile_bytes + 8 * 4096 > 160 * 1024 - 256) return false; launch_s<12, THREADS, ITEMS>(regroup, wg, p, grid, s); return true;
    case 13: if (regroup && tile_bytes + 8 * 8192 > 160 * 1024 - 256) return false; launch_s<13, THREADS, ITEMS>(regroup, wg, p, grid, s); return true;
    case 14: if (regroup) return false; launch_s<14, THREADS, ITEMS>(regroup, wg, p, grid, s); return true;
    default: return false;
  }
}

static bool dispatch(int fb, int threads, int items, bool regroup, bool wg, Params p, int grid, hipStream_t s) {
  if (threads == 1024 && items == 16) return launch_fb<1024, 16>(fb, regroup, wg, p, grid, s);
  if (threads == 1024 && items == 12) return launch_fb<1024, 12>(fb, regroup, wg, p, grid, s);
  if (threads == 1024 && items == 8) return launch_fb<1024, 8>(fb, regroup, wg, p, grid, s);
  if (threads == 512 && items == 16) return launch_fb<512, 16>(fb, regroup, wg, p, grid, s);
  if (threads == 512 && items == 8) return launch_fb<512, 8>(fb, regroup, wg, p, grid, s);
  if (threads == 256 && items == 16) return launch_fb<256, 16>(fb, regroup, wg, p, grid, s);
  return false;
}

int main(int argc, char **argv) {
  uint32_t n = argc > 1 ? (uint32_t)atoll(argv[1]) : 1000000000u;
  std::vector<std::string> variants;
  for (int i = 2; i < argc; ++i) variants.push_back(argv[i]);
  if (variants.empty()) variants = {"R8", "D8", "D11", "D12", "D13", "D14", "R11", "R12i12", "R13i8", "D13w", "D13n", "D13wn", "R12i12w", "D13t512", "D13t256"};
  uint64_t *keys, *out;
  CHECK(hipMalloc(&keys, (size_t)n * 8));
  const size_t out_cap = (size_t)n + (size_t)n / 4 + (1u << 22);      // + a dump area of one tile behind the regions
  CHECK(hipMalloc(&out, out_cap * 8));
  uint32_t *cursor, *flag;
  CHECK(hipMalloc(&cursor, (size_t)(1 << 14) * 8 * 4));
  CHECK(hipMalloc(&flag, 4));
  unsigned long long *acc;
  CHECK(hipMalloc(&acc, 8 * 8));
  gen_keys<<<2048, 256>>>(keys, n, 100000000ull);
  CHECK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (const std::string &v : variants) {
    const bool regroup = v[0] == 'R';
    int fb = atoi(v.c_str() + 1);
    size_t pos = 1; while (pos < v.size() && isdigit(v[pos])) ++pos;
    bool wg = false, nostore = false; int threads = 1024, items = 16;
    while (pos < v.size()) {
      const char c = v[pos++];
      if (c == 'w') wg = true;
      else if (c == 'a') wg = false;
      else if (c == 'n') nostore = true;
      else if (c == 't') { threads = atoi(v.c_str() + pos); while (pos < v.size() && isdigit(v[pos])) ++pos; }
      else if (c == 'i') { items = atoi(v.c_str() + pos); while (pos < v.size() && isdigit(v[pos])) ++pos; }
    }
    const uint32_t F = 1u << fb, nreg = F * 8;
    const double mean = (double)n / nreg;
    const uint32_t cap = (uint32_t)(mean + 8 * std::sqrt(mean) + 64);
    if ((size_t)nreg * cap + 32768 > out_cap) { printf("%-10s skipped (layout needs %zu tuples)\n", v.c_str(), (size_t)nreg * cap); continue; }
    const uint32_t tile = threads * items;
    const uint32_t chunk = ((131072 + tile - 1) / tile) * tile;
    const int grid = (int)((n + chunk - 1) / chunk);
    Params p{keys, n, out, cursor, cap, chunk, flag, nreg * cap, wg ? 1 : 0, nostore ? 1 : 0};
    float best = 1e9f, sum = 0; int reps = 4; bool ok = true;
    for (int r = 0; r < reps; ++r) {
      CHECK(hipMemsetAsync(cursor, 0, (size_t)nreg * 4, 0));
      CHECK(hipMemsetAsync(flag, 0, 4, 0));
      CHECK(hipEventRecord(e0, 0));
      ok = dispatch(fb, threads, items, regroup, wg, p, grid, 0);
      if (!ok) break;
      CHECK(hipEventRecord(e1, 0));
      CHECK(hipEventSynchronize(e1));
      CHECK(hipGetLastError());
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (r) { sum += ms; best = ms < best ? ms : best; }
    }
    if (!ok) { printf("%-10s not instantiated\n", v.c_str()); continue; }
    uint32_t hflag; CHECK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
    unsigned long long h[8] = {0};
    if (!nostore) {
      CHECK(hipMemset(acc, 0, 64));
      check_regions<<<4096, 256>>>(out, cursor, nreg, cap, fb, acc);
      check_input<<<4096, 256>>>(keys, n, acc);
      CHECK(hipMemcpy(h, acc, 64, hipMemcpyDeviceToHost));
    }
    const bool good = nostore || (h[0] == h[4] && h[1] == h[5] && h[2] == 0 && h[3] == n && !hflag);
    printf("%-10s F=%5u T=%5u cap=%6u  best %.3f ms  avg %.3f ms  %.2f TB/s (16 B/row)  %s%s\n", v.c_str(), F, tile, cap, best,
           sum / (reps - 1), 16.0 * n / best / 1e9, good ? "OK" : "MISMATCH", hflag ? " overflow" : "");
    fflush(stdout);
  }
  return 0;
}
