// ref_hash_shim.cpp -- exposes the REFERENCE's own hash functors to the oracle tests.
//
// Compiles /root/reference/libgdf/src/hashmap/hash_functions.cuh where it lies (nothing
// is copied into this repository) for the host, with the CUDA qualifiers defined away.
// The resulting oracle/_ref/libref_hash.so exists only in the build container (the
// reference tree is not present on the GPU box) and is used to (a) validate
// oracle/gdf_oracle.c's restatement of Murmur3_32 / hash_combine / IdentityHash and
// (b) regenerate tests/golden/murmur3_32.json (tests/golden/make_golden.py).
#include <cstdint>
#include <cstring>
#define __host__
#define __device__
#define __forceinline__ inline
#include REF_HASH_HEADER

extern "C" {
#define REF_MURMUR(NAME, T) \
  uint32_t ref_murmur_##NAME(const void *p) { T v; std::memcpy(&v, p, sizeof(T)); return MurmurHash3_32<T>()(v); }
REF_MURMUR(i8, int8_t) REF_MURMUR(i16, int16_t) REF_MURMUR(i32, int32_t) REF_MURMUR(i64, int64_t)
REF_MURMUR(f32, float) REF_MURMUR(f64, double)
uint32_t ref_hash_combine(uint32_t l, uint32_t r) { return MurmurHash3_32<int32_t>().hash_combine(l, r); }
uint32_t ref_identity_i32(int32_t v) { return IdentityHash<int32_t>()(v); }
uint32_t ref_identity_i64(int64_t v) { return IdentityHash<int64_t>()(v); }
uint32_t ref_identity_i8(int8_t v) { return IdentityHash<int8_t>()(v); }
}
