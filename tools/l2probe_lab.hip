// l2probe_lab.hip -- can ONE regroup level be dropped by probing an L2-resident table instead of an LDS-resident one?
//
// Today both relations are regrouped twice (256-way, then 128-way) so that a build partition (~3 k
// tuples) fits LDS: 48 B of HBM traffic per probe row.  With ONE level (P = 512 .. 2048 partitions) a build partition is
// 50 k .. 200 k tuples: a 1 - 2 MB open-addressing table that fits the 4 MB L2 of ONE XCD -- if every workgroup that probes
// partition p runs on the same XCD (blockIdx % 8) and the XCD works on few partitions at a time.  Per probe row that is
// 8 + 8 (level 1) + 8 + 8 (probe in, pairs out) = 32 B of HBM traffic plus one random 16-byte L2 hit.
//
// This lab measures only the probe kernel of that design, on synthetic pre-partitioned tuples:
//   1e9 probe tuples (key32 << 32 | row), NP / P per partition, every key present in the partition's table;
//   table = buckets of two 8-byte entries (key32 << 32 | build row), linear probing over buckets.
// Kill criterion: the kernel has to beat jk_scatter2 + jk_probe_fast = 5.9 ms by a margin that pays for a 512-way (instead
// of 256-way) level 1, i.e. run in <= 4.5 ms.
//
// build: hipcc --offload-arch=gfx950 -O3 tools/l2probe_lab.hip -o /tmp/l2probe_lab ; run: /tmp/l2probe_lab
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x;
}
__device__ __forceinline__ uint32_t part_of(uint32_t key, int pbits) { return pbits ? lowbias32(key) >> (32 - pbits) : 0u; }
__device__ __forceinline__ uint32_t bucket_of(uint32_t key, uint32_t nbuckets) { return (uint32_t)(((uint64_t)(key * 0x9e3779b1u) * nbuckets) >> 32); }

// ---- set-up kernels (not timed) ----
__global__ void count_parts(uint32_t nb, int pbits, uint32_t *cnt) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) atomicAdd(&cnt[part_of(i, pbits)], 1u);
}
__global__ void fill_parts(uint32_t nb, int pbits, const uint32_t *off, uint32_t *cur, uint32_t *keys) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
    const uint32_t p = part_of(i, pbits);
    keys[off[p] + atomicAdd(&cur[p], 1u)] = i;
  }
}
__global__ void build_tables(uint32_t nb, int pbits, uint32_t nbuckets, unsigned long long *table) {
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < nb; i += gridDim.x * blockDim.x) {
    const uint32_t p = part_of(i, pbits);
    unsigned long long *t = table + (size_t)p * nbuckets * 2;
    uint32_t b = bucket_of(i, nbuckets);
    const unsigned long long e = ((unsigned long long)i << 32) | (i ^ 0x55555555u);      // build row = a function of the key (checked by the probe)
    for (;;) {
      if (atomicCAS(&t[2 * b], ~0ull, e) == ~0ull) break;
      if (atomicCAS(&t[2 * b + 1], ~0ull, e) == ~0ull) break;
      b = b + 1 == nbuckets ? 0 : b + 1;
    }
  }
}
__device__ __forceinline__ uint64_t splitmix(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
__global__ void make_probe(uint64_t np_per, uint32_t P, const uint32_t *off, const uint32_t *cnt, const uint32_t *keys, unsigned long long *tuples) {
  const uint64_t total = np_per * P;
  for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t p = (uint32_t)(j / np_per);
    const uint32_t k = keys[off[p] + (uint32_t)(splitmix(j) % cnt[p])];
    tuples[j] = ((unsigned long long)k << 32) | (uint32_t)j;
  }
}

// ---- the probe kernel ----
// unit u: partition and chunk.  XCD-affine order: workgroup b runs on XCD b % 8 (round-robin dispatch); XCD x takes the
// partitions p with p % 8 == x, all chunks of one partition before the next, so that an XCD's resident workgroups share
// one or two tables.
template <bool NT>
__global__ __launch_bounds__(512) void probe(const unsigned long long *__restrict__ tuples, const unsigned long long *__restrict__ table,
                                             uint32_t nbuckets, uint64_t np_per, uint32_t chunk, uint32_t chunks_per_part, uint32_t P, int affine,
                                             int32_t *__restrict__ out_probe, int32_t *__restrict__ out_build, unsigned long long *bad) {
  uint32_t p, c;
  if (affine) {
    const uint32_t x = blockIdx.x & 7u, s = blockIdx.x >> 3;       // s-th workgroup of XCD x
    p = (s / chunks_per_part) * 8u + x;
    c = s % chunks_per_part;
  } else {
    p = blockIdx.x / chunks_per_part;
    c = blockIdx.x % chunks_per_part;
  }
  if (p >= P) return;
  const unsigned long long *t = table + (size_t)p * nbuckets * 2;
  const uint64_t begin = (uint64_t)p * np_per + (uint64_t)c * chunk;
  const uint64_t end = begin + chunk < (uint64_t)(p + 1) * np_per ? begin + chunk : (uint64_t)(p + 1) * np_per;
  unsigned long long wrong = 0;
  constexpr int B = 4;                                              // 16-byte loads per thread and batch: 8 tuples
  for (uint64_t base = begin; base < end; base += 512 * 2 * B) {
    unsigned long long w[2 * B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      uint64_t i = base + (uint64_t)(b * 512 + threadIdx.x) * 2;
      if (i + 2 > end) i = end - 2;
      const ulonglong2 v = NT ? ulonglong2{__builtin_nontemporal_load(tuples + i), __builtin_nontemporal_load(tuples + i + 1)}
                              : *reinterpret_cast<const ulonglong2 *>(tuples + i);
      w[2 * b] = v.x; w[2 * b + 1] = v.y;
    }
    ulonglong2 e[2 * B];
    uint32_t bk[2 * B];
#pragma unroll
    for (int k = 0; k < 2 * B; ++k) {                               // 8 independent random L2 reads
      bk[k] = bucket_of((uint32_t)(w[k] >> 32), nbuckets);
      e[k] = *reinterpret_cast<const ulonglong2 *>(t + 2 * (size_t)bk[k]);
    }
#pragma unroll
    for (int k = 0; k < 2 * B; ++k) {
      const uint32_t key = (uint32_t)(w[k] >> 32);
      uint32_t row;
      for (;;) {
        if ((uint32_t)(e[k].x >> 32) == key) { row = (uint32_t)e[k].x; break; }
        if ((uint32_t)(e[k].y >> 32) == key) { row = (uint32_t)e[k].y; break; }
        if (e[k].y == ~0ull) { row = 0xffffffffu; break; }
        bk[k] = bk[k] + 1 == nbuckets ? 0 : bk[k] + 1;
        e[k] = *reinterpret_cast<const ulonglong2 *>(t + 2 * (size_t)bk[k]);
      }
      wrong += row != (key ^ 0x55555555u);
      const uint64_t i = base + (uint64_t)((k >> 1) * 512 + threadIdx.x) * 2 + (k & 1);
      if (i < end) {
        if (NT) { __builtin_nontemporal_store((int32_t)(uint32_t)w[k], out_probe + i); __builtin_nontemporal_store((int32_t)row, out_build + i); }
        else { out_probe[i] = (int32_t)(uint32_t)w[k]; out_build[i] = (int32_t)row; }
      }
    }
  }
  if (wrong) atomicAdd(bad, wrong);
}

int main(int argc, char **argv) {
  const uint64_t NP = argc > 1 ? strtoull(argv[1], 0, 10) : 1000000000ull;
  const uint32_t NB = argc > 2 ? (uint32_t)strtoul(argv[2], 0, 10) : 100000000u;
  unsigned long long *tuples, *bad;
  int32_t *op, *ob;
  CHECK(hipMalloc(&tuples, NP * 8 + 64));
  CHECK(hipMalloc(&op, NP * 4 + 64));
  CHECK(hipMalloc(&ob, NP * 4 + 64));
  CHECK(hipMalloc(&bad, 8));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%6s %9s %8s %7s %6s %3s | %8s %8s\n", "P", "buckets", "tableMB", "chunk", "affine", "nt", "ms", "bad");
  for (int pbits : {8, 9, 10, 11}) {
    const uint32_t P = 1u << pbits;
    const uint64_t np_per = NP / P;
    uint32_t *cnt, *off, *cur, *keys;
    CHECK(hipMalloc(&cnt, P * 4)); CHECK(hipMalloc(&off, P * 4)); CHECK(hipMalloc(&cur, P * 4)); CHECK(hipMalloc(&keys, (size_t)NB * 4));
    CHECK(hipMemset(cnt, 0, P * 4)); CHECK(hipMemset(cur, 0, P * 4));
    count_parts<<<2048, 256>>>(NB, pbits, cnt);
    std::vector<uint32_t> h(P), o(P);
    CHECK(hipMemcpy(h.data(), cnt, P * 4, hipMemcpyDeviceToHost));
    uint32_t run = 0, mx = 0;
    for (uint32_t p = 0; p < P; ++p) { o[p] = run; run += h[p]; mx = h[p] > mx ? h[p] : mx; }
    CHECK(hipMemcpy(off, o.data(), P * 4, hipMemcpyHostToDevice));
    fill_parts<<<2048, 256>>>(NB, pbits, off, cur, keys);
    make_probe<<<8192, 256>>>(np_per, P, off, cnt, keys, tuples);
    CHECK(hipDeviceSynchronize());
    for (int loadsel = 0; loadsel < 2; ++loadsel) {
      // buckets per partition: a power of two is not required (mulhi slots); load 0.75 and 0.5 of 2-entry buckets
      const uint32_t nbuckets = (uint32_t)((double)mx / 2.0 / (loadsel ? 0.5 : 0.75)) + 16;
      unsigned long long *table;
      CHECK(hipMalloc(&table, (size_t)P * nbuckets * 16));
      CHECK(hipMemset(table, 0xff, (size_t)P * nbuckets * 16));
      build_tables<<<4096, 256>>>(NB, pbits, nbuckets, table);
      CHECK(hipDeviceSynchronize());
      for (uint32_t chunk : {32768u, 131072u}) {
        const uint32_t cpp = (uint32_t)((np_per + chunk - 1) / chunk);
        for (int affine = 1; affine >= 0; --affine) {
          for (int nt = 1; nt >= 0; --nt) {
            if (!affine && !nt) continue;
            const uint32_t grid = affine ? ((P + 7) / 8) * cpp * 8 : P * cpp;
            float best = 1e9f;
            unsigned long long hb = 0;
            for (int rep = 0; rep < 3; ++rep) {
              CHECK(hipMemset(bad, 0, 8));
              CHECK(hipEventRecord(e0));
              if (nt) probe<true><<<grid, 512>>>(tuples, table, nbuckets, np_per, chunk, cpp, P, affine, op, ob, bad);
              else probe<false><<<grid, 512>>>(tuples, table, nbuckets, np_per, chunk, cpp, P, affine, op, ob, bad);
              CHECK(hipEventRecord(e1));
              CHECK(hipEventSynchronize(e1));
              float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
              best = ms < best ? ms : best;
              CHECK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
            }
            printf("%6u %9u %8.2f %7u %6d %3d | %8.3f %8llu\n", P, nbuckets, nbuckets * 16.0 / 1048576.0, chunk, affine, nt, best, hb);
            fflush(stdout);
          }
        }
      }
      CHECK(hipFree(table));
    }
    CHECK(hipFree(cnt)); CHECK(hipFree(off)); CHECK(hipFree(cur)); CHECK(hipFree(keys));
  }
  return 0;
}
