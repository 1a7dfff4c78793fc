// placement_va.hip -- standalone (round 5): is a block's write bandwidth a property of its PHYSICAL memory or of its VIRTUAL address?
//   (A) ONE set of physical chunks (8 x 1 GiB, hipMemCreate) mapped at many virtual addresses at once (aliases): same memory, different VA;
//   (B) many DIFFERENT physical sets mapped, one after the other, at ONE virtual address.
// per mapping: streaming write GB/s, streaming read GB/s, 768-byte runs at 16384 fronts (ms).
//   build: hipcc -O3 --offload-arch=gfx950 tools/placement_va.hip -o tools/placement_va
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
  printf("%-22s %3d  write %5.0f GB/s  read %5.0f GB/s  fronts %6.3f ms   va %p\n", kind, b, bytes / tw / 1e6, bytes / tr / 1e6, tf, p);
  fflush(stdout);
}
int main(int argc, char **argv) {
  const int nva = argc > 1 ? atoi(argv[1]) : 24;
  const size_t chunk = (size_t)1 << 30, bytes = (size_t)8 << 30;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipMalloc(&acc, 8));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc d{};
  d.location = prop.location;
  d.flags = hipMemAccessFlagsProtReadWrite;
  auto make_set = [&]() { std::vector<hipMemGenericAllocationHandle_t> hs(bytes / chunk); for (auto &h : hs) CHECK(hipMemCreate(&h, chunk, &prop, 0)); return hs; };
  auto map_at = [&](void *va, const std::vector<hipMemGenericAllocationHandle_t> &hs) {
    for (size_t i = 0; i < hs.size(); ++i) CHECK(hipMemMap((char *)va + i * chunk, chunk, 0, hs[i], 0));
    CHECK(hipMemSetAccess(va, bytes, &d, 1));
  };
  // (A) one physical set, many virtual addresses (all reservations held: distinct addresses)
  {
    auto hs = make_set();
    std::vector<void *> vas;
    for (int i = 0; i < nva; ++i) {
      void *va = nullptr;
      CHECK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
      vas.push_back(va);
      map_at(va, hs);
      score("A same memory", i, va, bytes);
    }
    // the first few once more: is the number a stable property of the mapping?
    for (int i = 0; i < 4 && i < nva; ++i) score("A again", i, vas[i], bytes);
    for (void *va : vas) { CHECK(hipMemUnmap(va, bytes)); CHECK(hipMemAddressFree(va, bytes)); }
    for (auto h : hs) CHECK(hipMemRelease(h));
  }
  // (B) one virtual address, many physical sets (all sets held: distinct memory)
  {
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> sets;
    for (int i = 0; i < 12; ++i) {
      sets.push_back(make_set());
      map_at(va, sets.back());
      score("B same address", i, va, bytes);
      CHECK(hipMemUnmap(va, bytes));
    }
    CHECK(hipMemAddressFree(va, bytes));
    for (auto &hs : sets) for (auto h : hs) CHECK(hipMemRelease(h));
  }
  return 0;
}
