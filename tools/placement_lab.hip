// placement_lab.hip -- standalone experiment (round 5): what distinguishes a FAST physical placement of a multi-GB scatter buffer from a
// slow one?  jk_scatter1 runs 12 % faster on about one fresh hipMalloc in five; the pool picks by timing the real kernel.
// This lab allocates a series of fresh 8 GB blocks (all held, so that they are physically distinct), runs the level-1-like regroup
// scatter of tools/scatter_lab.hip (R8: 256 bins x 8 XCD regions, 16384-tuple tiles) into each of them and, on the SAME block, a set of
// synthetic patterns -- streaming write, streaming read, 768-byte runs at 16384 write fronts, random 64-byte writes -- to see whether any
// cheap pattern tells the blocks apart the way the real kernel does.  Not part of libgdf.so.
//   build:  hipcc -O3 --offload-arch=gfx950 tools/placement_lab.hip -o tools/placement_lab
//   run:    tools/placement_lab [blocks=12] [rows=1000000000]
#define main scatter_lab_main
#include "scatter_lab.hip"
#undef main

__global__ __launch_bounds__(256) void m_stream_write(uint4 *out, size_t n16) {
  const uint4 v{1u, 2u, 3u, 4u};
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) out[i] = v;
}
__global__ __launch_bounds__(256) void m_stream_read(const uint4 *in, size_t n16, unsigned long long *acc) {
  unsigned long long s = 0;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { const uint4 v = in[i]; s += v.x ^ v.w; }
  if (s == 0x123456789abcdefULL) atomicAdd(acc, s);
}
// FRONTS write fronts spread evenly over the block; wave g in round r appends one run of RUN_BYTES (64 lanes x 12 B) to front
// hash(g, r) % FRONTS at that front's r-th run slot: many short contiguous runs at many open fronts, no LDS, no atomics
template <int RUN_DWORDS_PER_LANE>
__global__ __launch_bounds__(1024) void m_fronts(uint32_t *out, size_t block_dwords, uint32_t fronts, uint32_t rounds) {
  const uint32_t lane = threadIdx.x & 63u, gw = blockIdx.x * 16u + (threadIdx.x >> 6), nw = gridDim.x * 16u;
  const size_t region = block_dwords / fronts;                              // dwords per front
  const uint32_t run = 64u * RUN_DWORDS_PER_LANE;
  const uint32_t slots = (uint32_t)(region / run);
  for (uint32_t r = 0; r < rounds; ++r) {
    uint32_t h = (gw + r * nw) * 2654435761u;
    h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 13;
    const uint32_t f = h % fronts;
    const size_t at = (size_t)f * region + (size_t)((r * 7919u + gw) % slots) * run + lane * RUN_DWORDS_PER_LANE;
#pragma unroll
    for (int k = 0; k < RUN_DWORDS_PER_LANE; ++k) out[at + k] = h + k;
  }
}
__global__ __launch_bounds__(256) void m_random64(uint4 *out, size_t n16, uint32_t per_thread) {
  uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
  for (uint32_t i = 0; i < per_thread; ++i) {
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    const size_t line = ((size_t)x * (n16 / 4)) >> 32;                       // a 64-byte sector
    const uint4 v{x, x, x, x};
    out[line * 4 + (threadIdx.x & 3u)] = v;
  }
}

template <class F>
static float time_ms(F &&launch, int reps = 3) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(e0, 0));
    launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return best;
}

int main(int argc, char **argv) {
  const int nblocks = argc > 1 ? atoi(argv[1]) : 12;
  const uint32_t n = argc > 2 ? (uint32_t)atoll(argv[2]) : 1000000000u;
  uint64_t *keys;
  CHECK(hipMalloc(&keys, (size_t)n * 8));
  gen_keys<<<2048, 256>>>(keys, n, 100000000ull);
  uint32_t *cursor, *flag;
  unsigned long long *acc;
  CHECK(hipMalloc(&cursor, (size_t)(1 << 14) * 8 * 4));
  CHECK(hipMalloc(&flag, 4));
  CHECK(hipMalloc(&acc, 64));
  CHECK(hipDeviceSynchronize());
  const int fb = 8;
  const uint32_t F = 1u << fb, nreg = F * 8;
  const double mean = (double)n / nreg;
  const uint32_t cap = (uint32_t)(mean + 8 * std::sqrt(mean) + 64);
  const size_t tuples = (size_t)nreg * cap + 32768;
  const size_t bytes = tuples * 8;
  const uint32_t tile = 1024 * 16, chunk = ((131072 + tile - 1) / tile) * tile;
  const int grid = (int)((n + chunk - 1) / chunk);
  std::vector<uint64_t *> blocks;
  printf("# block bytes %.2f GB; scatter R8 = scat_regroup<8, 1024, 16> over %u rows (ms, best of 3); synthetic patterns on the same block\n", bytes / 1e9, n);
  printf("# %5s %10s %12s %12s %14s %14s %12s\n", "block", "scatter", "stream_wr", "stream_rd", "fronts16k_768B", "fronts2k_768B", "random64");
  for (int b = 0; b < nblocks; ++b) {
    uint64_t *out = nullptr;
    if (hipMalloc(&out, bytes) != hipSuccess) { printf("# block %d: out of memory\n", b); break; }
    blocks.push_back(out);
    Params p{keys, n, out, cursor, cap, chunk, flag, nreg * cap, 0, 0};
    const float sc = time_ms([&] {
      CHECK(hipMemsetAsync(cursor, 0, (size_t)nreg * 4, 0));
      CHECK(hipMemsetAsync(flag, 0, 4, 0));
      launch<8, 1024, 16, false>(true, p, grid, 0);
    });
    const size_t n16 = bytes / 16;
    const float wr = time_ms([&] { m_stream_write<<<2048, 256>>>((uint4 *)out, n16); });
    const float rd = time_ms([&] { m_stream_read<<<2048, 256>>>((const uint4 *)out, n16, acc); });
    // 1e9 tuples x 6 B in 768-byte runs = 7.8e6 runs: 256 workgroups x 16 waves x 1907 rounds
    const float f16 = time_ms([&] { m_fronts<3><<<256, 1024>>>((uint32_t *)out, bytes / 4, 16384u, 1907u); });
    const float f2 = time_ms([&] { m_fronts<3><<<256, 1024>>>((uint32_t *)out, bytes / 4, 2048u, 1907u); });
    const float rn = time_ms([&] { m_random64<<<2048, 256>>>((uint4 *)out, n16, 64u); });
    printf("  %5d %10.3f %12.3f %12.3f %14.3f %14.3f %12.3f   %p\n", b, sc, wr, rd, f16, f2, rn, (void *)out);
    fflush(stdout);
  }
  // the same blocks once more, in reverse order: is a block's time a property of the block?
  printf("# second visit, reverse order (scatter only)\n");
  for (int b = (int)blocks.size() - 1; b >= 0; --b) {
    Params p{keys, n, blocks[b], cursor, cap, chunk, flag, nreg * cap, 0, 0};
    const float sc = time_ms([&] {
      CHECK(hipMemsetAsync(cursor, 0, (size_t)nreg * 4, 0));
      CHECK(hipMemsetAsync(flag, 0, 4, 0));
      launch<8, 1024, 16, false>(true, p, grid, 0);
    });
    printf("  %5d %10.3f\n", b, sc);
  }
  return 0;
}
