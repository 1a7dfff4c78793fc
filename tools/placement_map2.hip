// placement_map2.hip -- standalone (round 5): at which block size do slow-WRITE placements appear, and is a slow block slow everywhere?
// For block sizes 1 / 2 / 4 / 8 GiB: ten fresh blocks each (held), the streaming-write rate of the whole block and of each of its
// 512 MiB sub-ranges (GB/s).   build: hipcc -O3 --offload-arch=gfx950 tools/placement_map2.hip -o tools/placement_map2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void w(uint4 *out, size_t n16) {
  const uint4 v{1u, 2u, 3u, 4u};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}
static float wr(void *p, size_t bytes, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    float ms;
    CHECK(hipEventRecord(e0, 0)); w<<<2048, 256>>>((uint4 *)p, bytes / 16); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
  }
  return (float)(bytes / best / 1e6);
}
int main(int argc, char **argv) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<void *> held;
  const int per = argc > 1 ? atoi(argv[1]) : 10;
  for (size_t gib : {1, 2, 4, 8}) {
    const size_t bytes = gib << 30, sub = (size_t)512 << 20;
    printf("# %zu GiB blocks: whole-block write GB/s | per 512 MiB sub-range\n", gib);
    for (int b = 0; b < per; ++b) {
      void *p = nullptr;
      if (hipMalloc(&p, bytes) != hipSuccess) { printf("# out of memory\n"); break; }
      held.push_back(p);
      printf("%5.0f |", wr(p, bytes, e0, e1));
      for (size_t o = 0; o < bytes; o += sub) printf(" %4.0f", wr((char *)p + o, sub, e0, e1));
      printf("   %p\n", p);
      fflush(stdout);
    }
  }
  return 0;
}
