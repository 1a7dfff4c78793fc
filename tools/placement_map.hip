// placement_map.hip -- standalone (round 5, follows tools/placement_lab.hip): the lab showed that a block's speed under the scatter kernels is
// its plain streaming-WRITE bandwidth (reads do not differ).  How is the fast memory laid out?  Allocate many small chunks (all held), time a
// streaming write and a streaming read of each, print them in allocation order with their virtual addresses.
//   build: hipcc -O3 --offload-arch=gfx950 tools/placement_map.hip -o tools/placement_map      run: tools/placement_map [chunk MiB=512] [chunks=256]
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
int main(int argc, char **argv) {
  const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 512;
  const int n = argc > 2 ? atoi(argv[2]) : 256;
  const size_t bytes = mib << 20, n16 = bytes / 16;
  unsigned long long *acc;
  CHECK(hipMalloc(&acc, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  std::vector<void *> held;
  printf("# chunk MiB %zu; per chunk: streaming write GB/s, streaming read GB/s (best of 4), virtual address\n", mib);
  for (int c = 0; c < n; ++c) {
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("# out of memory at chunk %d\n", c); break; }
    held.push_back(p);
    float bw = 1e9f, br = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      float ms;
      CHECK(hipEventRecord(e0, 0)); w<<<2048, 256>>>((uint4 *)p, n16); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1)); bw = ms < bw ? ms : bw;
      CHECK(hipEventRecord(e0, 0)); r<<<2048, 256>>>((const uint4 *)p, n16, acc); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      CHECK(hipEventElapsedTime(&ms, e0, e1)); br = ms < br ? ms : br;
    }
    printf("%4d %8.0f %8.0f  %p\n", c, bytes / bw / 1e6, bytes / br / 1e6, p);
  }
  return 0;
}
