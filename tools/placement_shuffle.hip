// placement_shuffle.hip -- standalone (round 5): does the ORDER in which physical chunks are laid out in a block matter?  A pool of physical
// chunks (hipMemCreate, chunk size argv[1] MiB, 16 GiB in all) is mapped into 8 GiB blocks in creation order, reversed, interleaved and in random
// permutations; each block is scored by a streaming write and by the write-fronts pattern; the first block also over growing footprints.
//   build: hipcc -O3 --offload-arch=gfx950 tools/placement_shuffle.hip -o tools/placement_shuffle
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
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
__global__ __launch_bounds__(256) void w(uint4 *out, size_t n16) {
  const uint4 v{1u, 2u, 3u, 4u};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}
static hipEvent_t e0, e1;
template <class F> static float best_ms(F &&f) {
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CHECK(hipEventRecord(e0, 0)); f(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  return best;
}
int main(int argc, char **argv) {
  const size_t chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 2) << 20;
  const size_t bytes = (size_t)8 << 30;
  const int per = (int)(bytes / chunk), pool = 2 * per;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc d{};
  d.location = prop.location;
  d.flags = hipMemAccessFlagsProtReadWrite;
  void *va = nullptr;
  CHECK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
  using H = hipMemGenericAllocationHandle_t;
  std::vector<H> hs(pool);
  for (auto &h : hs) CHECK(hipMemCreate(&h, chunk, &prop, 0));
  printf("# %d chunks of %zu MiB, 8 GiB blocks of %d chunks at va %p\n", pool, chunk >> 20, per, va);
  auto show = [&](const char *name, const std::vector<int> &ix, bool curve) {
    for (int i = 0; i < per; ++i) CHECK(hipMemMap((char *)va + (size_t)i * chunk, chunk, 0, hs[ix[i]], 0));
    CHECK(hipMemSetAccess(va, bytes, &d, 1));
    const float g = (float)(bytes / best_ms([&] { w<<<2048, 256>>>((uint4 *)va, bytes / 16); }) / 1e6);
    const float f = best_ms([&] { fronts<<<256, 1024>>>((uint32_t *)va, bytes / 4, 16384u, 1907u); });
    printf("%-44s write %5.0f GB/s  fronts %6.3f ms", name, g, f);
    if (curve) {
      printf("   fronts over the first 1 / 2 / 4 GiB (512 KiB regions):");
      for (int gib = 1; gib <= 4; gib *= 2) printf(" %5.3f", best_ms([&] { fronts<<<256, 1024>>>((uint32_t *)va, ((size_t)gib << 30) / 4, 2048u * gib, 1907u); }));
    }
    printf("\n");
    fflush(stdout);
    CHECK(hipMemUnmap(va, bytes));
  };
  std::vector<int> ix(per);
  for (int i = 0; i < per; ++i) ix[i] = i;
  show("creation order, first half of the pool", ix, true);
  for (int i = 0; i < per; ++i) ix[i] = per + i;
  show("creation order, second half", ix, true);
  for (int i = 0; i < per; ++i) ix[i] = per - 1 - i;
  show("first half reversed", ix, false);
  for (int i = 0; i < per; ++i) ix[i] = 2 * i;
  show("every second chunk", ix, false);
  for (int i = 0; i < per; ++i) ix[i] = (i % 2) * per + i / 2;
  show("halves interleaved", ix, false);
  std::mt19937 rng(12345);
  for (int trial = 0; trial < 6; ++trial) {
    std::vector<int> all(pool);
    for (int i = 0; i < pool; ++i) all[i] = i;
    std::shuffle(all.begin(), all.end(), rng);
    all.resize(per);
    char name[64];
    snprintf(name, sizeof name, "random permutation %d", trial);
    show(name, all, trial == 0);
  }
  for (int i = 0; i < per; ++i) ix[i] = i;
  show("creation order, first half, again", ix, false);
  return 0;
}
