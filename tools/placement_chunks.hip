// placement_chunks.hip -- standalone (round 5): the streaming-write rate of individual PHYSICAL chunks made with the virtual-memory API
// (hipMemCreate, mapped one by one), each scored by five write sweeps inside one timing; then blocks COMPOSED of the fastest / the slowest
// chunks are scored as wholes.  Can a fast block be built on purpose?
//   build: hipcc -O3 --offload-arch=gfx950 tools/placement_chunks.hip -o tools/placement_chunks     run: tools/placement_chunks [chunk MiB=1024] [chunks=160]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void w(uint4 *out, size_t n16, int sweeps) {
  const uint4 v{1u, 2u, 3u, 4u};
  for (int s = 0; s < sweeps; ++s)
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = uint4{v.x + (unsigned)s, v.y, v.z, v.w};
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
template <class F> static float best_ms(F &&f, int reps = 3) {
  float best = 1e9f;
  for (int rep = 0; rep < reps; ++rep) {
    float ms;
    CHECK(hipEventRecord(e0, 0)); f(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); CHECK(hipGetLastError());
    CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  return best;
}
int main(int argc, char **argv) {
  const size_t chunk = (size_t)(argc > 1 ? atoll(argv[1]) : 1024) << 20;
  const int n = argc > 2 ? atoi(argv[2]) : 160;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc d{};
  d.location = prop.location;
  d.flags = hipMemAccessFlagsProtReadWrite;
  void *va1 = nullptr;
  CHECK(hipMemAddressReserve(&va1, chunk, 0, nullptr, 0));
  struct C { hipMemGenericAllocationHandle_t h; float gbs; int idx; };
  std::vector<C> cs;
  printf("# chunk MiB %zu: write GB/s of every chunk alone (5 sweeps per timing, best of 3), in creation order\n", chunk >> 20);
  for (int c = 0; c < n; ++c) {
    hipMemGenericAllocationHandle_t h;
    if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { printf("# out of memory at %d\n", c); (void)hipGetLastError(); break; }
    CHECK(hipMemMap(va1, chunk, 0, h, 0));
    CHECK(hipMemSetAccess(va1, chunk, &d, 1));
    const float ms = best_ms([&] { w<<<2048, 256>>>((uint4 *)va1, chunk / 16, 5); });
    CHECK(hipMemUnmap(va1, chunk));
    cs.push_back({h, (float)(5.0 * chunk / ms / 1e6), c});
    printf("%5.0f%s", cs.back().gbs, (c % 16 == 15) ? "\n" : " ");
  }
  printf("\n");
  // compose 8-chunk blocks: the fastest chunks, the slowest chunks, and chunks in creation order
  std::vector<C> sorted = cs;
  std::sort(sorted.begin(), sorted.end(), [](const C &a, const C &b) { return a.gbs > b.gbs; });
  const int per = (int)(((size_t)8 << 30) / chunk);
  auto compose = [&](const char *name, const std::vector<C> &pick) {
    const size_t bytes = chunk * pick.size();
    void *va = nullptr;
    CHECK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
    for (size_t i = 0; i < pick.size(); ++i) CHECK(hipMemMap((char *)va + i * chunk, chunk, 0, pick[i].h, 0));
    CHECK(hipMemSetAccess(va, bytes, &d, 1));
    const float tw = best_ms([&] { w<<<2048, 256>>>((uint4 *)va, bytes / 16, 1); });
    const float tf = best_ms([&] { fronts<<<256, 1024>>>((uint32_t *)va, bytes / 4, 16384u, 1907u); });
    float mean = 0; for (auto &c : pick) mean += c.gbs; mean /= pick.size();
    printf("%-34s write %5.0f GB/s  fronts %6.3f ms   (mean of its chunks alone %5.0f)\n", name, bytes / tw / 1e6, tf, mean);
    CHECK(hipMemUnmap(va, bytes)); CHECK(hipMemAddressFree(va, bytes));
  };
  if ((int)cs.size() >= 4 * per) {
    compose("8 GiB of the FASTEST chunks", std::vector<C>(sorted.begin(), sorted.begin() + per));
    compose("8 GiB of the next fastest", std::vector<C>(sorted.begin() + per, sorted.begin() + 2 * per));
    compose("8 GiB of the SLOWEST chunks", std::vector<C>(sorted.end() - per, sorted.end()));
    compose("8 GiB of median chunks", std::vector<C>(sorted.begin() + sorted.size() / 2 - per / 2, sorted.begin() + sorted.size() / 2 - per / 2 + per));
    for (int b = 0; b < 4; ++b) compose("8 GiB in creation order", std::vector<C>(cs.begin() + b * per, cs.begin() + (b + 1) * per));
    // interleaved: fast and slow chunks alternating
    std::vector<C> mix;
    for (int i = 0; i < per / 2; ++i) { mix.push_back(sorted[i]); mix.push_back(sorted[sorted.size() - 1 - i]); }
    compose("8 GiB fast / slow alternating", mix);
  }
  return 0;
}
