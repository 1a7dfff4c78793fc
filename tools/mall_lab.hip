// mall_lab.hip -- experiment: does the 256 MiB Infinity Cache keep a REUSED intermediate buffer out of HBM?
// (idea: run the join's two regroup levels batch by batch, level 1 writing a buffer that level 2 reads right away.)
//   build:  hipcc -O3 --offload-arch=gfx950 tools/mall_lab.hip -o tools/mall_lab
// Test A: ping-pong copies inside a working set of 2 x S bytes.
// Test B: big input --k1--> staging buffer --k2--> big output, staging buffer either REUSED every batch (S bytes) or a
//         fresh slice of a big buffer; reports total time per 8 GB of input for several batch sizes.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT_LOAD, bool NT_STORE>
__global__ __launch_bounds__(256) void copy16(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n16; i += stride) {
    uint4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t j = i + k * 256 < n16 ? i + k * 256 : n16 - 1;
      if (NT_LOAD) {
        const unsigned long long *p = (const unsigned long long *)(in + j);
        unsigned long long a = __builtin_nontemporal_load(p), b = __builtin_nontemporal_load(p + 1);
        v[k] = make_uint4((uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32));
      } else v[k] = in[j];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (i + k * 256 < n16) {
        if (NT_STORE) {
          unsigned long long *p = (unsigned long long *)(out + i + k * 256);
          __builtin_nontemporal_store(((unsigned long long)v[k].y << 32) | v[k].x, p);
          __builtin_nontemporal_store(((unsigned long long)v[k].w << 32) | v[k].z, p + 1);
        } else out[i + k * 256] = v[k];
      }
  }
}

int main() {
  const size_t BIG = (size_t)8 << 30;
  char *in, *out, *stage;
  CHECK(hipMalloc(&in, BIG)); CHECK(hipMalloc(&out, BIG)); CHECK(hipMalloc(&stage, BIG));
  CHECK(hipMemset(in, 1, BIG)); CHECK(hipMemset(out, 2, BIG)); CHECK(hipMemset(stage, 3, BIG));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int grid = 256 * 8;
  printf("Test A: ping-pong copy inside 2 x S\n");
  for (size_t mb : {8, 16, 32, 64, 96, 128, 192, 256, 512, 2048}) {
    const size_t S = mb << 20, n16 = S / 16;
    const int reps = (int)((BIG * 2) / S); // move 16 GB in total
    for (int w = 0; w < 2; ++w) {
      CHECK(hipEventRecord(e0, 0));
      for (int r = 0; r < reps; ++r) {
        if (r & 1) copy16<false, false><<<grid, 256>>>((const uint4 *)(stage + S), (uint4 *)stage, n16);
        else copy16<false, false><<<grid, 256>>>((const uint4 *)stage, (uint4 *)(stage + S), n16);
      }
      CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (w) printf("  S = %5zu MB  %8.3f ms for %d copies  %.2f TB/s (read+write)  %.1f us/launch\n", mb, ms, reps, 2.0 * S * reps / ms / 1e9, 1000.0 * ms / reps);
    }
  }
  printf("Test B: 8 GB in -> stage -> 8 GB out, per batch two kernels\n");
  for (int nt = 0; nt < 2; ++nt)
  for (size_t mb : {16, 32, 64, 128, 256, 1024}) {
    const size_t S = mb << 20, n16 = S / 16, nb = BIG / S;
    for (int reuse = 1; reuse >= 0; --reuse) {
      float best = 1e9;
      for (int w = 0; w < 3; ++w) {
        CHECK(hipEventRecord(e0, 0));
        for (size_t b = 0; b < nb; ++b) {
          char *st = reuse ? stage + (b & 1) * S : stage + b * S;
          if (nt) {
            copy16<true, false><<<grid, 256>>>((const uint4 *)(in + b * S), (uint4 *)st, n16);
            copy16<false, true><<<grid, 256>>>((const uint4 *)st, (uint4 *)(out + b * S), n16);
          } else {
            copy16<false, false><<<grid, 256>>>((const uint4 *)(in + b * S), (uint4 *)st, n16);
            copy16<false, false><<<grid, 256>>>((const uint4 *)st, (uint4 *)(out + b * S), n16);
          }
        }
        CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("  nt=%d batch %5zu MB  stage %s  %7.3f ms  (%.2f TB/s over 32 GB moved)\n", nt, mb, reuse ? "REUSED (2 x S)" : "fresh slices ", best, 4.0 * BIG / best / 1e9);
    }
  }
  return 0;
}
