// placement_api.hip -- standalone (round 5): does the WAY a multi-GB block is obtained change its streaming-write bandwidth (the property the
// scatter kernels' fast and slow placements come down to, tools/placement_lab.hip)?  Six 8 GiB blocks each (all held within a kind) from
//   hipMalloc | hipMallocAsync (stream-ordered pool) | virtual-memory API: hipMemCreate chunks of 1 GiB / 64 MiB / the recommended granularity
//   mapped into one reserved range | hipExtMallocWithFlags(hipDeviceMallocContiguous)
// per block: streaming write GB/s, streaming read GB/s, 768-byte runs at 16384 fronts (ms).
//   build: hipcc -O3 --offload-arch=gfx950 tools/placement_api.hip -o tools/placement_api
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void w(uint4 *out, size_t n16) {
  const uint4 v{1u, 2u, 3u, 4u};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void r(const uint4 *in, size_t n16, unsigned long long *acc) {
  unsigned long long s = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = in[i]; s += v.x ^ v.w; }
  if (s == 0x123456789abcdefULL) atomicAdd(acc, s);
}
__global__ __launch_bounds__(1024) void fronts(uint32_t *out, size_t block_dwords, uint32_t nfronts, uint32_t rounds) {
  const uint32_t lane = threadIdx.x & 63u, gw = blockIdx.x * 16u + (threadIdx.x >> 6), nw = gridDim.x * 16u;
  const size_t region = block_dwords / nfronts;
  const uint32_t run = 64u * 3u, slots = (uint32_t)(region / run);
  for (uint32_t i = 0; i < rounds; ++i) {
    uint32_t h = (gw + i * nw) * 2654435761u;
    h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 13;
    const size_t at = (size_t)(h % nfronts) * region + (size_t)((i * 7919u + gw) % slots) * run + lane * 3u;
    out[at] = h; out[at + 1] = h + 1; out[at + 2] = h + 2;
  }
}
static hipEvent_t e0, e1;
static unsigned long long *acc;
template <class F> static float best_ms(F &&f) {
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CHECK(hipEventRecord(e0, 0)); f(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  return best;
}
static void score(const char *kind, int b, void *p, size_t bytes) {
  const float tw = best_ms([&] { w<<<2048, 256>>>((uint4 *)p, bytes / 16); });
  const float tr = best_ms([&] { r<<<2048, 256>>>((const uint4 *)p, bytes / 16, acc); });
  const float tf = best_ms([&] { fronts<<<256, 1024>>>((uint32_t *)p, bytes / 4, 16384u, 1907u); });
  printf("%-28s %2d  write %5.0f GB/s  read %5.0f GB/s  fronts %6.3f ms   %p\n", kind, b, bytes / tw / 1e6, bytes / tr / 1e6, tf, p);
  fflush(stdout);
}
static void *vmm_block(size_t bytes, size_t chunk, std::vector<hipMemGenericAllocationHandle_t> *handles) {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  void *va = nullptr;
  if (hipMemAddressReserve(&va, bytes, 0, nullptr, 0) != hipSuccess) return nullptr;
  for (size_t o = 0; o < bytes; o += chunk) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) return nullptr;
    handles->push_back(h);
    if (hipMemMap((char *)va + o, chunk, 0, h, 0) != hipSuccess) return nullptr;
  }
  hipMemAccessDesc d{};
  d.location = prop.location;
  d.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(va, bytes, &d, 1) != hipSuccess) return nullptr;
  return va;
}
int main(int argc, char **argv) {
  const int per = argc > 1 ? atoi(argv[1]) : 6;
  const size_t bytes = (size_t)8 << 30;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipMalloc(&acc, 8));
  {
    std::vector<void *> held;
    for (int b = 0; b < per; ++b) { void *p; CHECK(hipMalloc(&p, bytes)); held.push_back(p); score("hipMalloc", b, p, bytes); }
    for (void *p : held) CHECK(hipFree(p));
  }
  {
    std::vector<void *> held;
    for (int b = 0; b < per; ++b) {
      void *p = nullptr;
      if (hipMallocAsync(&p, bytes, 0) != hipSuccess) { printf("hipMallocAsync failed\n"); (void)hipGetLastError(); break; }
      CHECK(hipStreamSynchronize(0));
      held.push_back(p); score("hipMallocAsync", b, p, bytes);
    }
    for (void *p : held) (void)hipFreeAsync(p, 0);
    CHECK(hipDeviceSynchronize());
  }
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  size_t gran_min = 0, gran_rec = 0;
  (void)hipMemGetAllocationGranularity(&gran_min, &prop, hipMemAllocationGranularityMinimum);
  (void)hipMemGetAllocationGranularity(&gran_rec, &prop, hipMemAllocationGranularityRecommended);
  printf("# VMM granularity: minimum %zu, recommended %zu bytes\n", gran_min, gran_rec);
  for (size_t chunk : {(size_t)1 << 30, (size_t)64 << 20, gran_rec ? gran_rec : (size_t)2 << 20}) {
    if (chunk < ((size_t)1 << 20)) chunk = (size_t)2 << 20;
    char kind[64];
    snprintf(kind, sizeof kind, "VMM chunks of %zu MiB", chunk >> 20);
    std::vector<hipMemGenericAllocationHandle_t> handles;
    std::vector<void *> held;
    for (int b = 0; b < per; ++b) {
      void *p = vmm_block(bytes, chunk, &handles);
      if (!p) { printf("%s: failed (%s)\n", kind, hipGetErrorString(hipGetLastError())); break; }
      held.push_back(p); score(kind, b, p, bytes);
    }
    for (void *p : held) { (void)hipMemUnmap(p, bytes); (void)hipMemAddressFree(p, bytes); }
    for (auto h : handles) (void)hipMemRelease(h);
  }
  {
    std::vector<void *> held;
    for (int b = 0; b < 3; ++b) {
      void *p = nullptr;
      if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) != hipSuccess) { printf("contiguous: failed\n"); (void)hipGetLastError(); break; }
      held.push_back(p); score("hipDeviceMallocContiguous", b, p, bytes);
    }
    for (void *p : held) CHECK(hipFree(p));
  }
  return 0;
}
