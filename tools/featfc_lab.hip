// LAB (not part of the library; round-3 experiment): to try it, copy into mansy_immersivevideostreaming_amd/csrc/ and call
// mansy_launch_featfc from PEng::head() / head_pair() in place of the FeatureNet + fc products for batches <= 1024 rows.  Measured: correct, and
// slower than the two launches it replaces (rollout step 31 -> 35 us).
// FeatureNet + head fc as ONE launch for the small batches of the PPO loop (rollout: 256 rows; minibatch: 512-588): the two dependent
// products of every policy forward
//     F   = LeakyReLU(obs[B, K] Wbd^T + b)          block-diagonal: ten branches of 128 features on disjoint column windows
//     A1' = F [B, 1280] Wfc^T                       (Wfc: one head's [128, 1280] or the stacked [actor ; critic] [256, 1280])
// were two launches on the ~7 us floor of a dependent product each.  The block-diagonal structure makes them chainable without any
// grid-wide synchronisation: A1' = sum_j F_j Wfc[:, 128 j : 128 j + 128]^T, so the workgroup that forms the 64 x 128 tile F_j of one
// branch for 64 rows can multiply it straight away by that branch's 128-column slice of Wfc and emit a PARTIAL A1' -- one slab per
// branch, summed by head_out_kernel exactly like the K-split slabs it already sums.  Grid = 10 branches x ceil(B / 64) row blocks.
//   phase 1: the exact-fp32 LDS-DMA loop of gemm_f32.hip on a 64 x 128 tile over the branch's K window (1-11 K-tiles), then bias +
//            LeakyReLU in registers and the shared epilogue, which stores the F tile to HBM (the backward needs F) and leaves it in
//            LDS, row-major [64][132];
//   phase 2: A fragments straight from that LDS image (132-float rows: the 16 rows of a ds_read_b128 group land on 16 different
//            4-bank groups), B = the Wfc slice by LDS-DMA, 4 K-tiles, then the shared epilogue into the branch's slab.
// Same fp32 operations as the two launches (v_mfma_f32_32x32x2_f32 = an fmaf chain); the grouping of the 1 280-term sums differs
// (ten 128-term slabs instead of fourteen 96-term ones), i.e. float32 rounding only.
#include "gemm_tile.h"
#include "../../include/mansy_hip.h"

using namespace mansy_gemm;

namespace {

constexpr int FF_NB = 10, FF_HID = 128, FF_FEAT = FF_NB * FF_HID;

struct FeatFcParams {
  const float* obs; int obs_ld; int rows;
  const float* Wbd; int kp; const float* bbd;
  const float* Wfc; int fc_ld;
  float* F; float* slabs; long long slab_stride;
  int win_off[FF_NB], win_len[FF_NB];
  float slope;
};

// fragment of the staged K-contiguous image [R][32] (gemm_f32.hip::read_frag_dma): lane (r, h), chunk c -> k = 16 h + 8 c + 0..7
__device__ __forceinline__ void ff_frag(const float* __restrict__ lds, int row, int h, int chunk, float (&out)[8]) {
  const int c0 = h * 4 + chunk * 2, sw = (row >> 1) & 7;
  const float4 v0 = *reinterpret_cast<const float4*>(lds + row * BK + ((c0 + 0) ^ sw) * 4);
  const float4 v1 = *reinterpret_cast<const float4*>(lds + row * BK + ((c0 + 1) ^ sw) * 4);
  out[0] = v0.x; out[1] = v0.y; out[2] = v0.z; out[3] = v0.w; out[4] = v1.x; out[5] = v1.y; out[6] = v1.z; out[7] = v1.w;
}

template <int N2>
__global__ __launch_bounds__(NT) void featfc_kernel(FeatFcParams q) {
  constexpr int BM = 64, BN1 = FF_HID, TN1 = BN1 / 64, TN2 = N2 / 64;
  constexpr int CLD = BN1 + 4;                                           // row stride of the F tile the first epilogue leaves in LDS
  constexpr int F_FLOATS = (BM * CLD + 255) / 256 * 256;                 // 1 KiB-aligned end: the phase-2 staging starts there
  constexpr int ST1 = (BM + BN1) * BK;                                   // phase-1 stage (floats): A [64][32] + B [128][32]
  constexpr int ST2 = N2 * BK;                                           // phase-2 stage: B [N2][32]
  constexpr int C2_FLOATS = BM * (N2 + 4);
  constexpr int NEED_A = 2 * ST1, NEED_B = F_FLOATS + 2 * ST2;
  constexpr int SMEM_FLOATS = (NEED_A > NEED_B ? NEED_A : NEED_B) > C2_FLOATS ? (NEED_A > NEED_B ? NEED_A : NEED_B) : C2_FLOATS;
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // heavy branches (the two 320-column ones: 11 K-tiles) are dealt first
  const int order[FF_NB] = {1, 2, 3, 0, 4, 5, 6, 7, 8, 9};
  const int j = order[blockIdx.x % FF_NB];
  const int m0 = (blockIdx.x / FF_NB) * BM;
  const int k_begin = q.win_off[j], nk = q.win_len[j] / BK;

  // ---------------------------------------------------------------- phase 1: F_j tile = obs[m0.., window] Wbd[128 j.., window]^T
  unsigned voa[BM / 32], vob[BN1 / 32];
#pragma unroll
  for (int i = 0; i < BM / 32; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voa[i] = (unsigned)((min(m0 + row, q.rows - 1) - m0) * q.obs_ld + c * 4) * 4u;
  }
#pragma unroll
  for (int i = 0; i < BN1 / 32; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    vob[i] = (unsigned)(row * q.kp + c * 4) * 4u;
  }
  const float* sa = q.obs + (long long)m0 * q.obs_ld + k_begin;
  const float* sb = q.Wbd + (long long)(j * FF_HID) * q.kp + k_begin;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * 1024u);

  f32x16 acc1[1][TN1];
#pragma unroll
  for (int jj = 0; jj < TN1; ++jj)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc1[0][jj][e] = 0.f;
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < BM / 32; ++i) glds16(voa[i], sa, lds_wave + i * 4096u);
#pragma unroll
    for (int i = 0; i < BN1 / 32; ++i) glds16(vob[i], sb, lds_wave + BM * BK * 4u + i * 4096u);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    sa += BK; sb += BK;
    const unsigned lds_next = lds_wave + (unsigned)(cur ^ 1) * (ST1 * 4u);
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < BM / 32; ++i) glds16(voa[i], sa, lds_next + i * 4096u);
#pragma unroll
      for (int i = 0; i < BN1 / 32; ++i) glds16(vob[i], sb, lds_next + BM * BK * 4u + i * 4096u);
    }
    const float* a_l = smem + cur * ST1;
    const float* b_l = a_l + BM * BK;
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      float af[8], bf[TN1][8];
      ff_frag(a_l, wm * 32 + r, h, chunk, af);
#pragma unroll
      for (int jj = 0; jj < TN1; ++jj) ff_frag(b_l, wn * (BN1 / 2) + jj * 32 + r, h, chunk, bf[jj]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int jj = 0; jj < TN1; ++jj) acc1[0][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[jj][kk], acc1[0][jj], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // bias + LeakyReLU in registers (column = wn * 64 + jj * 32 + r of this branch), then the shared epilogue: F tile -> HBM and -> LDS
#pragma unroll
  for (int jj = 0; jj < TN1; ++jj) {
    const float b = q.bbd[j * FF_HID + wn * (BN1 / 2) + jj * 32 + r];
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float v = acc1[0][jj][e] + b; acc1[0][jj][e] = v > 0.f ? v : v * q.slope; }
  }
  GemmParams p1;
  p1.A = nullptr; p1.B = nullptr; p1.C = q.F + j * FF_HID; p1.lda = p1.ldb = 0; p1.ldc = FF_FEAT; p1.M = q.rows; p1.N = BN1; p1.K = 0;
  p1.k_per_split = 0; p1.vec_ok = 1; p1.c_vec_ok = 1;
  // (BM / 64 == 1 here; the epilogue's accumulator parameter is f32x16 [BM / 64][BN / 64])
  gemm_epilogue<BM, BN1, SMEM_FLOATS>(p1, acc1, smem, m0, 0, tid, 0, p1.C);
  __syncthreads();                                                       // every wave sees the whole F tile in smem[row * CLD + col]

  // ---------------------------------------------------------------- phase 2: slab_j[m0.., :] = F_j tile (LDS) Wfc[:, 128 j ..]^T
  unsigned vo2[N2 / 32];
#pragma unroll
  for (int i = 0; i < N2 / 32; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    vo2[i] = (unsigned)(row * q.fc_ld + c * 4) * 4u;
  }
  const float* s2 = q.Wfc + j * FF_HID;
  const unsigned lds2 = lds_wave + (unsigned)(F_FLOATS * 4);
  f32x16 acc2[1][TN2];
#pragma unroll
  for (int jj = 0; jj < TN2; ++jj)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[0][jj][e] = 0.f;
#pragma unroll
  for (int i = 0; i < N2 / 32; ++i) glds16(vo2[i], s2, lds2 + i * 4096u);
  constexpr int NK2 = FF_HID / BK;
  const float* frow = smem + (wm * 32 + r) * CLD;
  for (int kt = 0; kt < NK2; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    s2 += BK;
    if (kt + 1 < NK2) {
#pragma unroll
      for (int i = 0; i < N2 / 32; ++i) glds16(vo2[i], s2, lds2 + (unsigned)(cur ^ 1) * (ST2 * 4u) + i * 4096u);
    }
    const float* b_l = smem + F_FLOATS + cur * ST2;
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      float af[8], bf[TN2][8];
      const float4 v0 = *reinterpret_cast<const float4*>(frow + kt * BK + h * 16 + chunk * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(frow + kt * BK + h * 16 + chunk * 8 + 4);
      af[0] = v0.x; af[1] = v0.y; af[2] = v0.z; af[3] = v0.w; af[4] = v1.x; af[5] = v1.y; af[6] = v1.z; af[7] = v1.w;
#pragma unroll
      for (int jj = 0; jj < TN2; ++jj) ff_frag(b_l, wn * (N2 / 2) + jj * 32 + r, h, chunk, bf[jj]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int jj = 0; jj < TN2; ++jj) acc2[0][jj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[jj][kk], acc2[0][jj], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();                                                       // F tile and staging idle: the second epilogue reuses the LDS
  GemmParams p2;
  p2.A = nullptr; p2.B = nullptr; p2.C = q.slabs + (long long)j * q.slab_stride; p2.lda = p2.ldb = 0; p2.ldc = N2; p2.M = q.rows; p2.N = N2; p2.K = 0;
  p2.k_per_split = 0; p2.vec_ok = 1; p2.c_vec_ok = 1;
  gemm_epilogue<BM, N2, SMEM_FLOATS>(p2, acc2, smem, m0, 0, tid, 0, p2.C);
}

}  // namespace

// n2: 128 (one head's fc weight [128, 1280]) or 256 (two heads stacked).  win_off / win_len: the ten branches' K windows in whole
// 32-wide K-tiles inside the packed image (the pack launch writes zeros around a branch inside its window).  slabs: [10][rows][n2].
int mansy_launch_featfc(const float* obs, int obs_ld, int rows, const float* Wbd, int kp, const float* bbd, const float* Wfc, int n2, float* F,
                        float* slabs, const int* win_off, const int* win_len, float slope, hipStream_t st) {
  MANSY_REQUIRE(obs && Wbd && bbd && Wfc && F && slabs && rows >= 1 && (n2 == 128 || n2 == 256), "featfc: bad arguments");
  auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  MANSY_REQUIRE(al16(obs) && al16(Wbd) && al16(Wfc) && al16(F) && al16(slabs) && obs_ld % 4 == 0 && kp % 32 == 0, "featfc: operands must be 16-byte aligned");
  FeatFcParams q;
  q.obs = obs; q.obs_ld = obs_ld; q.rows = rows; q.Wbd = Wbd; q.kp = kp; q.bbd = bbd; q.Wfc = Wfc; q.fc_ld = FF_FEAT; q.F = F; q.slabs = slabs;
  q.slab_stride = (long long)rows * n2; q.slope = slope;
  for (int j = 0; j < FF_NB; ++j) {
    MANSY_REQUIRE(win_off[j] % 32 == 0 && win_len[j] % 32 == 0 && win_len[j] >= 32 && win_off[j] + win_len[j] <= kp, "featfc: bad K window %d", j);
    q.win_off[j] = win_off[j]; q.win_len[j] = win_len[j];
  }
  const dim3 grid(FF_NB * mansy_ceil_div(rows, 64));
  if (n2 == 128) hipLaunchKernelGGL(featfc_kernel<128>, grid, dim3(NT), 0, st, q);
  else hipLaunchKernelGGL(featfc_kernel<256>, grid, dim3(NT), 0, st, q);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
