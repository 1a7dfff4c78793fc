// How long does a small device -> host read-back take after a tiny kernel?  (the library reads a few words back two or three times per call;
// a 0.4 ms C2 group-by spends ~0.1 ms outside its kernels)   hipcc --offload-arch=gfx950 -O2 tools/readback_lab.hip -o /tmp/readback_lab
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(unsigned int *p, unsigned int v) { if (threadIdx.x == 0) p[0] = v; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  unsigned int *d, *pinned, *mapped_dev, *mapped;
  hipStream_t s;
  hipStreamCreate(&s);
  hipMalloc(&d, 64);
  hipHostMalloc(&pinned, 64, hipHostMallocDefault);
  hipHostMalloc(&mapped, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostGetDevicePointer((void **)&mapped_dev, mapped, 0);
  const int reps = 2000;
  unsigned int h[16];
  for (int mode = 0; mode < 4; ++mode) {
    double best = 1e30, sum = 0;
    for (int r = 0; r < reps; ++r) {
      const double t0 = now();
      if (mode == 3) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, mapped_dev, (unsigned)r);
      else hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (unsigned)r);
      if (mode == 0) hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);                                     // pageable destination (what the library does)
      if (mode == 1) { hipMemcpyAsync(h, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
      if (mode == 2) { hipMemcpyAsync(pinned, d, 8, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); h[0] = pinned[0]; }
      if (mode == 3) { hipStreamSynchronize(s); h[0] = mapped[0]; }
      const double dt = now() - t0;
      if (h[0] != (unsigned)r) { printf("mode %d: wrong value\n", mode); return 1; }
      if (r >= 100) { sum += dt; best = dt < best ? dt : best; }
    }
    const char *name[] = {"hipMemcpy to pageable", "hipMemcpyAsync to pageable + sync", "hipMemcpyAsync to pinned + sync", "kernel writes mapped host memory + sync"};
    printf("%-45s  avg %6.1f us  min %6.1f us  (launch + read-back)\n", name[mode], sum / (reps - 100), best);
  }
  // the other direction: a small table uploaded before a kernel that reads it (64 B .. 128 KB)
  unsigned int *dbig, *pin_big;
  hipMalloc(&dbig, 1 << 17);
  hipHostMalloc(&pin_big, 1 << 17, hipHostMallocDefault);
  static unsigned int pageable[1 << 15];
  for (size_t bytes : {64ul, 1024ul, 4096ul, 131072ul}) {
    for (int mode = 0; mode < 2; ++mode) {
      double sum = 0;
      for (int r = 0; r < reps; ++r) {
        pageable[0] = pin_big[0] = (unsigned)r;
        const double t0 = now();
        hipMemcpyAsync(dbig, mode ? (void *)pin_big : (void *)pageable, bytes, hipMemcpyHostToDevice, s);
        const double t1 = now();
        hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, d, (unsigned)r);
        hipStreamSynchronize(s);
        const double t2 = now();
        if (r >= 100) sum += (mode == 0 ? t2 - t0 : t2 - t0), (void)t1;
      }
      printf("upload %6zu B from %-8s + tiny kernel + sync: avg %6.1f us\n", bytes, mode ? "pinned" : "pageable", sum / (reps - 100));
    }
  }
  return 0;
}
