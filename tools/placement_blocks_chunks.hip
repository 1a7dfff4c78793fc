// placement_blocks_chunks.hip -- standalone (round 5): blocks of 8 x 1 GiB physical chunks (hipMemCreate) are scored as wholes; then every
// chunk is scored alone, and blocks are re-composed from the chunks of fast and of slow blocks.  Is a block's speed the sum of its chunks'?
//   build: hipcc -O3 --offload-arch=gfx950 tools/placement_blocks_chunks.hip -o tools/placement_blocks_chunks
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
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
  const int nblocks = argc > 1 ? atoi(argv[1]) : 12;
  const size_t chunk = (size_t)1 << 30;
  const int per = 8;
  const size_t bytes = chunk * per;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc d{};
  d.location = prop.location;
  d.flags = hipMemAccessFlagsProtReadWrite;
  void *va = nullptr, *va1 = nullptr;
  CHECK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
  CHECK(hipMemAddressReserve(&va1, chunk, 0, nullptr, 0));
  using H = hipMemGenericAllocationHandle_t;
  auto score_block = [&](const std::vector<H> &hs, float *gbs, float *fr) {
    for (int i = 0; i < per; ++i) CHECK(hipMemMap((char *)va + (size_t)i * chunk, chunk, 0, hs[i], 0));
    CHECK(hipMemSetAccess(va, bytes, &d, 1));
    *gbs = (float)(bytes / best_ms([&] { w<<<2048, 256>>>((uint4 *)va, bytes / 16); }) / 1e6);
    *fr = best_ms([&] { fronts<<<256, 1024>>>((uint32_t *)va, bytes / 4, 16384u, 1907u); });
    CHECK(hipMemUnmap(va, bytes));
  };
  std::vector<std::vector<H>> blocks;
  std::vector<float> bscore;
  // some other allocations first, so that the blocks do not all come from the very start of a fresh device
  void *filler = nullptr;
  if (argc > 2) CHECK(hipMalloc(&filler, (size_t)atoll(argv[2]) << 30));
  for (int b = 0; b < nblocks; ++b) {
    std::vector<H> hs(per);
    for (auto &h : hs) CHECK(hipMemCreate(&h, chunk, &prop, 0));
    float g, f;
    score_block(hs, &g, &f);
    printf("block %2d  write %5.0f GB/s  fronts %6.3f ms   chunks alone (fronts over 1 GiB, 2048 fronts, ms):", b, g, f);
    for (int i = 0; i < per; ++i) {
      CHECK(hipMemMap(va1, chunk, 0, hs[i], 0));
      CHECK(hipMemSetAccess(va1, chunk, &d, 1));
      const float t = best_ms([&] { fronts<<<256, 1024>>>((uint32_t *)va1, chunk / 4, 2048u, 1907u); });
      CHECK(hipMemUnmap(va1, chunk));
      printf(" %5.3f", t);
    }
    printf("\n");
    fflush(stdout);
    blocks.push_back(hs);
    bscore.push_back(f);
  }
  // re-compose: the chunks of the fastest two blocks interleaved with those of the slowest two
  std::vector<int> order(nblocks);
  for (int i = 0; i < nblocks; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return bscore[a] < bscore[b]; });
  const int f0 = order[0], f1 = order[1], s0 = order[nblocks - 1], s1 = order[nblocks - 2];
  auto show = [&](const char *name, std::vector<H> hs) { float g, f; score_block(hs, &g, &f); printf("%-46s write %5.0f GB/s  fronts %6.3f ms\n", name, g, f); };
  printf("# fastest blocks %d %d, slowest %d %d\n", f0, f1, s0, s1);
  show("fastest block again", blocks[f0]);
  show("slowest block again", blocks[s0]);
  { std::vector<H> m; for (int i = 0; i < 4; ++i) m.push_back(blocks[f0][i]); for (int i = 0; i < 4; ++i) m.push_back(blocks[s0][i]); show("4 chunks of the fastest + 4 of the slowest", m); }
  { std::vector<H> m; for (int i = 0; i < 4; ++i) { m.push_back(blocks[f0][i]); m.push_back(blocks[s0][i]); } show("... alternating", m); }
  { std::vector<H> m; for (int i = 0; i < 4; ++i) { m.push_back(blocks[f0][i]); m.push_back(blocks[f1][i]); } show("4 + 4 chunks of the two fastest, alternating", m); }
  { std::vector<H> m; for (int i = 0; i < 4; ++i) { m.push_back(blocks[s0][i]); m.push_back(blocks[s1][i]); } show("4 + 4 chunks of the two slowest, alternating", m); }
  { std::vector<H> m(blocks[s0].rbegin(), blocks[s0].rend()); show("slowest block, chunks in reverse order", m); }
  { std::vector<H> m; for (int i = 0; i < per; ++i) m.push_back(blocks[(i * 5) % nblocks][i]); show("one chunk from each of eight blocks", m); }
  return 0;
}
