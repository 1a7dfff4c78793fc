// Round 6 lab: what does giving multi-GB blocks back to the runtime cost, and who pays?  (The placed pool's searches free their losers
// when they settle; one later hipMalloc of the process then took 1.7 - 2.5 s in every process, profiles/r6_i_place_*.json.)
// hipcc --offload-arch=gfx950 -O2 tools/lab/free_stall_lab.hip -o tools/lab/free_stall_lab
// argv: blocks (10), variant: 0 plain, 1 sleep 3 s behind the frees, 2 re-allocate another size (7 GiB), 3 free in REVERSE order,
// 4 blocks never touched by a kernel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void touch(unsigned long long *p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = i;
}
int main(int argc, char **argv) {
  const size_t GB = size_t(1) << 30;
  const int nblk = argc > 1 ? atoi(argv[1]) : 10, variant = argc > 2 ? atoi(argv[2]) : 0;
  size_t sz = 6 * GB;
  for (int round = 0; round < 3; ++round) {
    std::vector<void *> b(nblk);
    printf("round %d (variant %d), %d x %zu GiB, ms per hipMalloc:", round, variant, nblk, sz >> 30);
    for (auto &p : b) {
      const double t0 = now();
      if (hipMalloc(&p, sz) != hipSuccess) { printf("malloc failed\n"); return 1; }
      printf(" %.1f", now() - t0);
    }
    double t1 = now();
    if (variant != 4) for (auto p : b) hipLaunchKernelGGL(touch, dim3(2048), dim3(256), 0, 0, (unsigned long long *)p, sz / 8);
    (void)hipDeviceSynchronize();
    double t2 = now();
    if (variant == 3) for (int i = nblk - 1; i >= 0; --i) (void)hipFree(b[i]);
    else for (auto p : b) (void)hipFree(p);
    double t3 = now();
    printf(" | touch %.1f ms, free all %.1f ms\n", t2 - t1, t3 - t2);
    if (variant == 1) std::this_thread::sleep_for(std::chrono::seconds(3));
    if (variant == 2) sz += GB;
  }
  return 0;
}
