// Pieces shared by the GEMM main loops (gemm_f32.hip: exact-fp32 MFMA; gemm_bf16s.hip: split-bf16 MFMA): launch parameters and
// the fused epilogue.  Both loops end with 32x32 MFMA accumulator blocks in the same register layout.
#pragma once
#include <hip/hip_ext.h>
#include "mansy_kernels.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace mansy_gemm {

constexpr int BK = 32;
constexpr int KC_LD = BK + 4;   // K-contiguous LDS row stride (floats)
constexpr int NT = 256;

// Launch timing for bench.py's roofline leg: when the launcher has set a start / stop event pair, the product's kernel is dispatched
// with hipExtLaunchKernelGGL, which stamps the two events with the kernel's own begin / end (what rocprofv3 reports as its duration);
// events recorded around the launch would add the queue's per-packet overhead (~1 us) to every one of the ~285 products of a step.
extern hipEvent_t g_ev_start, g_ev_stop;
#define MANSY_GEMM_LAUNCH(kern, grid, block, st, params)                                                                        \
  do {                                                                                                                          \
    __atomic_fetch_add(&g_mansy_launch_count, 1ull, __ATOMIC_RELAXED);                                                          \
    if (mansy_gemm::g_ev_start) hipExtLaunchKernelGGL(kern, grid, block, 0, st, mansy_gemm::g_ev_start, mansy_gemm::g_ev_stop, 0, params); \
    else hipLaunchKernelGGL(kern, grid, block, 0, st, params);                                                                  \
  } while (0)

struct GemmParams {
  const float* A; const float* B; float* C;
  int lda, ldb, ldc;
  int M, N, K;
  int k_per_split;
  int vec_ok;
  int c_vec_ok;      // 16-byte epilogue legal: C, bias, mask, resid 16-B aligned, their lds and N multiples of 4
  GemmEpilogue ep;
  // optional second problem of identical shape in the same launch (LDS-DMA loop; grid.z = 2 x splits, problem slowest)
  const float* A2 = nullptr; const float* B2 = nullptr; float* C2 = nullptr; float* a_rowsum2 = nullptr; int splits_pp = 1;
  // > 0: the XCD-aware tile order walks the column tiles in groups of col_group (all row panels of a group before the next group), so that an
  // XCD's L2 holds ONE group's slice of B next to the A panels it streams (wide-N products: B alone is 3 MB of the 4 MB L2 at N = 1536)
  int col_group = 0;
  int c_rmw_ok = 0;  // C (and C2) 16-byte aligned, ldc % 4 == 0, N % 4 == 0: an accumulating product whose blocks have one owner may read-add-write float4
};

// Row-major pass of the fused epilogue: the C tile sits in LDS as [BM][BN + 4] floats (`smem`, written by the caller, who has also
// synchronised the workgroup); bias / activation / mask / dropout / residual and the store run on float4 rows, NTHR threads (tid in
// [0, NTHR)).  Shared by every main loop, whatever its wave layout, so that a product's value does not depend on the loop it ran on.
template <int BM, int BN, int NTHR>
__device__ __forceinline__ void gemm_epilogue_rows(const GemmParams& p, const float* smem, int m0, int n0, int tid, float* Cz) {
  constexpr int CLD = BN + 4;
  const GemmEpilogue& ep = p.ep;
  const float drop_scale = ep.drop.p > 0.f ? 1.f / (1.f - ep.drop.p) : 1.f;
  constexpr int C4 = BN / 4;
  constexpr int IT = BM * C4 / NTHR, GRP = IT < 4 ? IT : 4;
  static_assert(BM * C4 % NTHR == 0 && IT % GRP == 0, "epilogue groups");
  // Groups of GRP float4 per thread: the residual / mask loads of a whole group are issued before its first store (C may
  // alias the residual, so the compiler will not move a load above a store by itself: one exposed L2/HBM round trip per
  // float4 otherwise -- ~3 us on the [4096, 512] decoder products).
#pragma unroll
  for (int g0 = 0; g0 < IT; g0 += GRP) {
    float4 rr[GRP], mk[GRP];
    long long off_r[GRP], off_m[GRP];
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int idx = tid + (g0 + u) * NTHR;
      const int lr = idx / C4, c4 = idx % C4;
      const int row = min(m0 + lr, p.M - 1), col = min(n0 + c4 * 4, p.N - 4);      // clamped: out-of-range float4 are not stored
      off_r[u] = (long long)row * ep.resid_ld + col; off_m[u] = (long long)row * ep.mask_ld + col;
      rr[u] = make_float4(0.f, 0.f, 0.f, 0.f); mk[u] = make_float4(1.f, 1.f, 1.f, 1.f);
    }
    if (ep.resid16) {        // bf16 residual stream (bf16-storage mode): 8-byte loads of the image
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const mansy_bf16x4 t = *reinterpret_cast<const mansy_bf16x4*>(ep.resid16 + off_r[u]);
        rr[u] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
      }
    } else if (ep.resid) {   // one uniform branch around the group's loads, not one per load
#pragma unroll
      for (int u = 0; u < GRP; ++u) rr[u] = *reinterpret_cast<const float4*>(ep.resid + off_r[u]);
    }
    if (ep.mask16) {
#pragma unroll
      for (int u = 0; u < GRP; ++u) {
        const mansy_bf16x4 t = *reinterpret_cast<const mansy_bf16x4*>(ep.mask16 + off_m[u]);
        mk[u] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
      }
    } else if (ep.mask_src) {
#pragma unroll
      for (int u = 0; u < GRP; ++u) mk[u] = *reinterpret_cast<const float4*>(ep.mask_src + off_m[u]);
    }
#pragma unroll
    for (int u = 0; u < GRP; ++u) {
      const int idx = tid + (g0 + u) * NTHR;
      const int lr = idx / C4, c4 = idx % C4;
      const int row = m0 + lr, col = n0 + c4 * 4;
      if (row >= p.M || col >= p.N) continue;           // N % 4 == 0 on this path: a float4 is entirely in or out
      float4 v = *reinterpret_cast<const float4*>(smem + lr * CLD + c4 * 4);
      if (ep.pre_a && col >= ep.pre_col0) {        // pre_col0 % 4 == 0 on this path: a float4 is entirely in or out
        const long long po = (long long)row * ep.pre_ld + (col - ep.pre_col0);
        const float4 a = *reinterpret_cast<const float4*>(ep.pre_a + po);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
        if (ep.pre_b) { const float4 b2 = *reinterpret_cast<const float4*>(ep.pre_b + po); v.x += b2.x; v.y += b2.y; v.z += b2.z; v.w += b2.w; }
      }
      if (ep.bias) { const float4 b = *reinterpret_cast<const float4*>(ep.bias + col); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
      if (ep.relu) {
        v.x = v.x > 0.f ? v.x : v.x * ep.relu_slope; v.y = v.y > 0.f ? v.y : v.y * ep.relu_slope;
        v.z = v.z > 0.f ? v.z : v.z * ep.relu_slope; v.w = v.w > 0.f ? v.w : v.w * ep.relu_slope;
      }
      if (ep.mask_src || ep.mask16) {
        v.x = mk[u].x > 0.f ? v.x * ep.mask_scale : v.x * ep.mask_neg; v.y = mk[u].y > 0.f ? v.y * ep.mask_scale : v.y * ep.mask_neg;
        v.z = mk[u].z > 0.f ? v.z * ep.mask_scale : v.z * ep.mask_neg; v.w = mk[u].w > 0.f ? v.w * ep.mask_scale : v.w * ep.mask_neg;
      }
      if (ep.drop.p > 0.f) {
        const uint32_t base = (uint32_t)row * (uint32_t)p.N + (uint32_t)col;
        v.x = mansy_keep(ep.drop.seed, ep.drop.site, ep.drop.base + base + 0, ep.drop.p) ? v.x * drop_scale : 0.f;
        v.y = mansy_keep(ep.drop.seed, ep.drop.site, ep.drop.base + base + 1, ep.drop.p) ? v.y * drop_scale : 0.f;
        v.z = mansy_keep(ep.drop.seed, ep.drop.site, ep.drop.base + base + 2, ep.drop.p) ? v.z * drop_scale : 0.f;
        v.w = mansy_keep(ep.drop.seed, ep.drop.site, ep.drop.base + base + 3, ep.drop.p) ? v.w * drop_scale : 0.f;
      }
      v.x += rr[u].x; v.y += rr[u].y; v.z += rr[u].z; v.w += rr[u].w;
      if (Cz) *reinterpret_cast<float4*>(Cz + (long long)row * p.ldc + col) = v;
      if (ep.c16) {          // bf16 image of the output (round 6: the bf16-storage mode's consumers read this one)
        typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
        bf16x4_t o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
        *reinterpret_cast<bf16x4_t*>(ep.c16 + (long long)row * ep.c16_ld + col) = o;
      }
    }
  }
}

// Epilogue shared by both main loops.  C/D layout of the 32x32 MFMA block: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
// `smem` is the (now idle) staging LDS, at least BM*(BN+4) floats; SMEM_FLOATS is its size.
template <int BM, int BN, int SMEM_FLOATS>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&acc)[BM / 64][BN / 64 > 0 ? BN / 64 : 1], float* smem, int m0, int n0,
                                              int tid, int split, float* Cbase) {
  constexpr int TM = BM / 64, TN = BN / 64;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const GemmEpilogue& ep = p.ep;
  const bool atomic = ep.accumulate || (gridDim.z > 1 && ep.split_slab == 0);
  float* const Cz = Cbase + (long long)split * ep.split_slab;             // own slab per K split in slab mode
  const float drop_scale = ep.drop.p > 0.f ? 1.f / (1.f - ep.drop.p) : 1.f;
  if (!atomic && p.c_vec_ok) {
    // Row-major 16-byte epilogue: the accumulators (one column per lane, 16 scattered rows) are transposed through the LDS
    // staging buffers (free after the last barrier) so that bias / activation / mask / dropout / residual and the store all
    // run on float4 rows: 16 global_store_dwordx4 per thread instead of 64 global_store_dword, coalesced 512-B row pieces.
    constexpr int CLD = BN + 4;
    static_assert(BM * CLD <= SMEM_FLOATS, "C tile must fit the staging LDS");
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          smem[(wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wn * (BN / 2) + j * 32 + r] = acc[i][j][e];
    __syncthreads();
    gemm_epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid, Cz);
    return;
  }
  // Scalar path (atomics / unaligned): loads (mask / residual) hoisted into unconditional clamped-address batches.
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + r;
      const int colc = min(col, p.N - 1);
      const int row_base = m0 + wm * (BM / 2) + i * 32 + 4 * h;
      const float bias = ep.bias ? ep.bias[colc] : 0.f;
      float v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        float pre = 0.f;
        if (ep.pre_a && col >= ep.pre_col0 && col < p.N) {
          const long long po = (long long)min(row_base + (e & 3) + 8 * (e >> 2), p.M - 1) * ep.pre_ld + (col - ep.pre_col0);
          pre = ep.pre_a[po] + (ep.pre_b ? ep.pre_b[po] : 0.f);
        }
        v[e] = acc[i][j][e] + pre + bias;
        if (ep.relu) v[e] = v[e] > 0.f ? v[e] : v[e] * ep.relu_slope;
      }
      if (ep.mask_src) {
        float mk[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
          mk[e] = ep.mask_src[(long long)min(row_base + (e & 3) + 8 * (e >> 2), p.M - 1) * ep.mask_ld + colc];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = mk[e] > 0.f ? v[e] * ep.mask_scale : v[e] * ep.mask_neg;
      }
      if (ep.drop.p > 0.f) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const uint32_t row = (uint32_t)(row_base + (e & 3) + 8 * (e >> 2));
          v[e] = mansy_keep(ep.drop.seed, ep.drop.site, ep.drop.base + row * (uint32_t)p.N + (uint32_t)col, ep.drop.p) ? v[e] * drop_scale : 0.f;
        }
      }
      if (ep.resid) {
        float rr[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
          rr[e] = ep.resid[(long long)min(row_base + (e & 3) + 8 * (e >> 2), p.M - 1) * ep.resid_ld + colc];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += rr[e];
      }
      if (col < p.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = row_base + (e & 3) + 8 * (e >> 2);
          if (row < p.M) {
            float* dst = Cz + (long long)row * p.ldc + col;
            if (atomic) atomicAdd(dst, v[e]); else *dst = v[e];
          }
        }
      }
    }
  }
}

// One LDS-DMA piece (global_load_lds_dwordx4): per-lane source = sbase (SGPR pair, wave-uniform) + voff (VGPR, bytes); destination = LDS
// byte address lds_dst (wave-uniform, via M0) + lane * 16 -- 1 KiB of LDS written contiguously per wave instruction.  Issued through
// inline asm (hipcc would put s_waitcnt vmcnt(0) in front of every ds_read that follows an LDS-DMA); completion is counted by hand.
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

}  // namespace mansy_gemm

// bf16-storage products (gemm_bf16a.hip): A16 / B planes K-contiguous (NN) and the K-major weight-gradient form (TN)
int mansy_gemm_bf16a_nn(const mansy_gemm::GemmParams& p, int tile, hipStream_t st);
int mansy_gemm_bf16a_tn(const mansy_gemm::GemmParams& p, int splits, int kgroups, hipStream_t st);
// capture of wave-split-K launches (gemm_f32.hip): see there
int mansy_gemm_capture_begin();
int mansy_gemm_capture_end(mansy_gemm::GemmParams* out, int* variant, int* gx, int* gy, int* gz, int max_n);
// split-bf16 main loop (gemm_bf16s.hip); tile 128 -> 128x128, else 64x64; prec 3 = bf16x3, 6 = bf16x6
int mansy_gemm_bf16s_dispatch(const mansy_gemm::GemmParams& p, int tile, int prec, int a_kmajor, int b_kmajor, int splits, hipStream_t st);
// B operand pre-split into bf16 planes (GemmEpilogue::b_planes), A K-contiguous, no split-K
int mansy_gemm_bf16p_dispatch(const mansy_gemm::GemmParams& p, int tile, int prec, hipStream_t st);
