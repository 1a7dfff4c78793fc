// hash.cuh -- the PUBLIC row hash (visible through gdf_hash and
// gdf_hash_partition), which therefore has to be bit-exact with the reference:
//   MurmurHash3_x86_32, seed 0, over the sizeof(T) raw bytes of each element
//   (src/hashmap/hash_functions.cuh:30-121), columns folded left to right with
//   the Boost-style combine l ^ (r + 0x9e3779b9 + (l<<6) + (l>>2)) and the first
//   column NOT combined (src/gdf_table.cuh:704-854);
//   IdentityHash = static_cast<uint32_t>(value) (hash_functions.cuh:129-164).
// Golden values: tests/golden/murmur3_32.json (captured from the reference header).
#pragma once
#include "common.h"

namespace gdf_amd {

__host__ __device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ __forceinline__ uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85ebca6bu;
  h ^= h >> 13; h *= 0xc2b2ae35u;
  h ^= h >> 16;
  return h;
}

__host__ __device__ __forceinline__ uint32_t murmur_block(uint32_t h, uint32_t k) {
  k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
  h ^= k; h = rotl32(h, 13);
  return h * 5u + 0xe6546b64u;
}

// Murmur3_32 of an element of `width` bytes whose little-endian bytes are the
// low bytes of `bits`.
__host__ __device__ __forceinline__ uint32_t murmur3_32(uint64_t bits, int width) {
  uint32_t h = 0;   // seed
  if (width == 8) {
    h = murmur_block(h, (uint32_t)bits);
    h = murmur_block(h, (uint32_t)(bits >> 32));
  } else if (width == 4) {
    h = murmur_block(h, (uint32_t)bits);
  } else {            // 1 or 2 tail bytes, no body block
    uint32_t k = (uint32_t)bits & (width == 2 ? 0xffffu : 0xffu);
    k *= 0xcc9e2d51u; k = rotl32(k, 15); k *= 0x1b873593u;
    h ^= k;
  }
  h ^= (uint32_t)width;
  return fmix32(h);
}

__host__ __device__ __forceinline__ uint32_t hash_combine(uint32_t l, uint32_t r) {
  return l ^ (r + 0x9e3779b9u + (l << 6) + (l >> 2));
}

#ifdef __HIPCC__
// static_cast<uint32_t>(value) for every element kind.  Integer kinds wrap; float
// kinds use the hardware conversion (v_cvt_u32_f32 / v_cvt_u32_f64: truncates,
// saturates, NaN and negatives -> 0), the AMD analogue of what the reference's
// cast compiles to on its GPU.
__device__ __forceinline__ uint32_t identity_hash(const ColView &c, int64_t i) {
  switch (c.kind) {
    case K_I8:  return (uint32_t)((const int8_t *)c.data)[i];
    case K_I16: return (uint32_t)((const int16_t *)c.data)[i];
    case K_I32: return (uint32_t)((const int32_t *)c.data)[i];
    case K_I64: return (uint32_t)((const int64_t *)c.data)[i];
    case K_F32: return (uint32_t)((const float *)c.data)[i];
    default:    return (uint32_t)((const double *)c.data)[i];
  }
}

template <bool MURMUR>
__device__ __forceinline__ uint32_t hash_row(const KeyTable &t, int64_t i) {
  uint32_t h = 0;
  for (int c = 0; c < t.ncols; ++c) {
    const uint32_t k = MURMUR ? murmur3_32(load_bits(t.col[c], i), t.col[c].width) : identity_hash(t.col[c], i);
    h = (c == 0) ? k : hash_combine(h, k);
  }
  return h;
}
#endif

}  // namespace gdf_amd
