// Shared helpers for the MANSY gfx950 kernels (device + host).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define MANSY_OK 0
#define MANSY_EINVAL (-1)
#define MANSY_EHIP (-2)
#define MANSY_EUNSUPPORTED (-3)

extern "C" void mansy_set_error(const char* fmt, ...);

#define MANSY_HIP_CHECK(expr)                                                         \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      mansy_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      return MANSY_EHIP;                                                              \
    }                                                                                 \
  } while (0)

#define MANSY_LAUNCH_CHECK() MANSY_HIP_CHECK(hipGetLastError())

#define MANSY_REQUIRE(cond, ...)        \
  do {                                  \
    if (!(cond)) {                      \
      mansy_set_error(__VA_ARGS__);     \
      return MANSY_EINVAL;              \
    }                                   \
  } while (0)

// ---- counter hash shared bit-for-bit with oracle/rng.py -------------------------------
__host__ __device__ __forceinline__ uint32_t mansy_hash_u32(uint32_t seed, uint32_t site, uint32_t idx) {
  uint32_t h = seed ^ (site * 0x9E3779B9u);
  h ^= idx + 0x7F4A7C15u + (h << 6) + (h >> 2);
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  h += idx * 0x27D4EB2Fu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
  return h;
}
__host__ __device__ __forceinline__ float mansy_uniform01(uint32_t seed, uint32_t site, uint32_t idx) {
  return (float)(mansy_hash_u32(seed, site, idx) >> 8) * (1.0f / 16777216.0f);
}
// dropout keep decision: keep iff u >= p
__host__ __device__ __forceinline__ bool mansy_keep(uint32_t seed, uint32_t site, uint32_t idx, float p) {
  return mansy_uniform01(seed, site, idx) >= p;
}

struct MansyDrop {   // a dropout site; p == 0 disables
  float p;
  uint32_t seed;
  uint32_t site;
};
static inline MansyDrop mansy_no_drop() { MansyDrop d; d.p = 0.f; d.seed = 0; d.site = 0; return d; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline int mansy_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }
