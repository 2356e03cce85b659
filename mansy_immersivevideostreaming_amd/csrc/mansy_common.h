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

// Every kernel launch of the library goes through MANSY_LAUNCH (or MANSY_GEMM_LAUNCH, gemm_tile.h): it counts the launch in a process
// counter that bench.py reads live (mansy_prof_launch_count; launches enqueued during a hipGraph capture count once, at capture).
// A measurement hook only: nothing in the library reads it.
extern "C" unsigned long long g_mansy_launch_count;
#define MANSY_LAUNCH(kern, grid, block, shmem, st, ...)               \
  do {                                                                \
    __atomic_fetch_add(&g_mansy_launch_count, 1ull, __ATOMIC_RELAXED); \
    hipLaunchKernelGGL(kern, grid, block, shmem, st, __VA_ARGS__);    \
  } while (0)

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
  uint32_t base;     // added to the kernel's element index: a launch over the rows [b0, b0 + n) of a tensor draws the mask the launch over
                     // all rows would draw there (base = b0 x the kernel's elements per row)
};
static inline MansyDrop mansy_no_drop() { MansyDrop d; d.p = 0.f; d.seed = 0; d.site = 0; d.base = 0; return d; }

// Cross-lane reductions on the VALU (DPP row rotations inside each 16-lane row, v_readlane across the four rows) instead of
// __shfl_xor, which hipcc lowers to ds_bpermute_b32 -- an LDS-pipe round trip per step.
template <int CTRL>
__device__ __forceinline__ float mansy_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float mansy_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// sum / max over each aligned group of 16 lanes, result in all 16 (row_ror:8,4,2,1)
__device__ __forceinline__ float row16_sum(float v) {
  v += mansy_dpp<0x128>(v); v += mansy_dpp<0x124>(v); v += mansy_dpp<0x122>(v); v += mansy_dpp<0x121>(v);
  return v;
}
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, mansy_dpp<0x128>(v)); v = fmaxf(v, mansy_dpp<0x124>(v)); v = fmaxf(v, mansy_dpp<0x122>(v)); v = fmaxf(v, mansy_dpp<0x121>(v));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  return (mansy_lane(v, 0) + mansy_lane(v, 16)) + (mansy_lane(v, 32) + mansy_lane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  return fmaxf(fmaxf(mansy_lane(v, 0), mansy_lane(v, 16)), fmaxf(mansy_lane(v, 32), mansy_lane(v, 48)));
}

static inline int mansy_ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- bf16 images of activations (MANSY_PREC_BF16 with bf16 storage, round 6): a row-wise kernel that produces an operand of a dense product also
// stores its bf16 image (round to nearest even, what v_cvt_pk_bf16_f32 does), which the product then stages by LDS-DMA as it is (csrc/gemm_bf16a.hip).
// `p16` is the image's address for the SAME element index as the float store; null = no image wanted.
typedef __bf16 mansy_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void mansy_st_bf16x4(unsigned short* p16, float x, float y, float z, float w) {
  mansy_bf16x4 o; o[0] = (__bf16)x; o[1] = (__bf16)y; o[2] = (__bf16)z; o[3] = (__bf16)w;
  *reinterpret_cast<mansy_bf16x4*>(p16) = o;
}
__device__ __forceinline__ void mansy_st_bf16(unsigned short* p16, float x) { *reinterpret_cast<__bf16*>(p16) = (__bf16)x; }
