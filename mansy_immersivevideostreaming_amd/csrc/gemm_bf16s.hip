// Split-bf16 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x16_bf16): the opt-in precision modes of the dense products.
// Operands stay fp32 in HBM; on their way into LDS every element is split into bf16 terms by round-to-nearest-even
// (v_cvt_pk_bf16_f32):   a = a0 + a1 (+ a2),  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)
// and the product is formed from bf16 x bf16 MFMAs accumulated in fp32:
//   bf16x3 (NP = 2 planes):  a0 b0 + a0 b1 + a1 b0                          dropped terms ~2^-16 |a||b|
//   bf16x6 (NP = 3 planes):  + a1 b1 + a0 b2 + a2 b0                        dropped terms ~2^-24 |a||b|  (fp32-equivalent)
// at 1/16 of the fp32-matrix cycle cost per MFMA, i.e. 16/3 and 16/6 of the fp32-matrix peak.  Parity study of both variants
// against the reference goldens: tools/bf16_split_study.py (bf16x6 is indistinguishable from fp32 on every output and gradient;
// bf16x3 keeps outputs / loss within 1.3e-5 and all tile decisions, gradients of small-magnitude parameters to ~1e-2).
//
// Structure: 256 threads = 4 waves (2x2), tile 128x128 or 64x64, BK = 32 (two 16-k MFMA steps), register-staged: the fp32 tile
// of K-tile t+1 is loaded into registers while the MFMAs of tile t run, then split and written to ONE set of LDS planes
// (two barriers per K-tile; two workgroups per CU cover each other's staging phase -- LDS stays <= 68 KB).
// LDS image per plane and operand, both global layouts: [rows][32 bf16], row stride 80 B (5 x 16 B: the 16 lanes of a
// ds_read_b128 group read 16 different rows, 5*row mod 16 is a bijection -> conflict-free; the ds_write_b128 groups of 8 lanes
// likewise).  Fragment of lane (r = lane & 31, h = lane >> 5) for MFMA step s: the 8 bf16 at k = 16 s + 8 h of row r -- A and
// B agree, and the sum over k is order-free.
//   K-contiguous operand: a thread stages 8 consecutive k of one row (2 x global_load_dwordx4)
//   K-major operand:      a thread stages 8 k of one row with the 64 lanes of a wave on 64 consecutive rows
//                         (8 x global_load_dword, 256 B contiguous per instruction)
// Whole K-tiles only (K % 32 == 0, 16-byte aligned operands): the dispatcher (gemm_f32.hip) routes everything else to the
// fp32 loops.  The epilogue is the shared one (gemm_tile.h): the 32x32 accumulator blocks have the fp32 kernels' layout.
#ifndef MANSY_BF16S_SCHED
#define MANSY_BF16S_SCHED 0
#endif
#include "gemm_tile.h"

using namespace mansy_gemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

// LDS image per plane and operand: [rows][32 bf16], element offset of the 16-B chunk q of a row:
//   compact (the two-stage loop): 64-B rows, chunk q at slot q ^ ((row >> 2) & 3) -- the 16 lanes of a ds_read_b128 group read
//            four aligned row quadruples with four different keys: 16 different 16-B slots
//   padded  (the one-stage loop): 80-B rows (5 slots: 5 * row mod 16 is a bijection), no swizzle; measured 7 % faster there
template <bool PAD>
__device__ __forceinline__ int lds_off(int row, int q) { return PAD ? row * 40 + q * 8 : row * 32 + ((q ^ ((row >> 2) & 3)) << 3); }

template <int R>
struct Stg { static constexpr int ITEMS = R / 64; };      // (row, 8-k chunk) items per thread and operand

// Per-thread loop-invariant source offsets (floats, relative to the tile corner of the current K-tile).  Rows beyond the
// matrix are clamped onto the last valid one (their accumulators are never stored).
template <int R, bool KMAJ, bool PAD>
__device__ __forceinline__ void stage_offsets(int ld, int row0, int nrows, unsigned (&off)[Stg<R>::ITEMS], int (&ldso)[Stg<R>::ITEMS],
                                              int (&srow)[Stg<R>::ITEMS], int tid) {
#pragma unroll
  for (int i = 0; i < Stg<R>::ITEMS; ++i) {
    int kc;
    if (!KMAJ) {
      const int idx = tid + i * NT;
      srow[i] = idx >> 2; kc = idx & 3;
      off[i] = (unsigned)((min(row0 + srow[i], nrows - 1) - row0) * ld + kc * 8);
    } else {
      srow[i] = (tid & 63) + 64 * i; kc = tid >> 6;
      off[i] = (unsigned)(kc * 8 * ld + (min(row0 + srow[i], nrows - 1) - row0));
    }
    ldso[i] = lds_off<PAD>(srow[i], kc);
  }
}

template <int R, bool KMAJ>
__device__ __forceinline__ void stage_load(const float* __restrict__ corner, int ld, const unsigned (&off)[Stg<R>::ITEMS],
                                           float (&v)[Stg<R>::ITEMS][8]) {
#pragma unroll
  for (int i = 0; i < Stg<R>::ITEMS; ++i) {
    if (!KMAJ) {
      const float4* s = reinterpret_cast<const float4*>(corner + off[i]);
      const float4 a = s[0], b = s[1];
      v[i][0] = a.x; v[i][1] = a.y; v[i][2] = a.z; v[i][3] = a.w; v[i][4] = b.x; v[i][5] = b.y; v[i][6] = b.z; v[i][7] = b.w;
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) v[i][kk] = (corner + (long long)kk * ld)[off[i]];
    }
  }
}

// Split 8 staged floats into NP bf16 planes and store them (one ds_write_b128 per plane).
template <int NP>
__device__ __forceinline__ void split_store(__bf16* __restrict__ dst, int plane_elems, const float (&v)[8]) {
  bf16x8 p0, p1, p2;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 t0 = (__bf16)v[e];
    p0[e] = t0;
    if (NP >= 2) {
      const float r1 = v[e] - (float)t0;
      const __bf16 t1 = (__bf16)r1;
      p1[e] = t1;
      if (NP == 3) { const float r2 = r1 - (float)t1; p2[e] = (__bf16)r2; }
    }
  }
  *reinterpret_cast<bf16x8*>(dst) = p0;
  if (NP >= 2) *reinterpret_cast<bf16x8*>(dst + plane_elems) = p1;
  if (NP == 3) *reinterpret_cast<bf16x8*>(dst + 2 * plane_elems) = p2;
}

// DB: two LDS stages -- the split + LDS store of K-tile t+1 is issued between the MFMAs of tile t (one barrier per K-tile, the
// global loads of tile t+2 fly across a whole tile); !DB: one stage, two barriers per K-tile (less LDS: the 3-plane mode keeps
// two workgroups per CU that way).
template <int BM, int BN, bool AKM, bool BKM, int NP, bool DB>
__global__ __launch_bounds__(NT, 2) void gemm_bf16s_kernel(GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int SROW = DB ? 32 : 40;
  constexpr int A_PLANE = BM * SROW, B_PLANE = BN * SROW;                 // bf16 elements
  constexpr int STAGE = NP * (A_PLANE + B_PLANE);                         // bf16 elements per stage
  constexpr int STAGE_FLOATS = (DB ? 2 : 1) * STAGE / 2;
  constexpr int C_FLOATS = BM * (BN + 4);
  constexpr int SMEM_FLOATS = STAGE_FLOATS > C_FLOATS ? STAGE_FLOATS : C_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  __bf16* const planes = reinterpret_cast<__bf16*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y, split;
  {   // XCD-aware bijective remap over the whole 3-D grid, K split slowest (see gemm_f32_dma_kernel)
    const int per_split = gridDim.x * gridDim.y, nwg = per_split * gridDim.z;
    const int orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    split = t / per_split; t -= split * per_split;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  int k_begin = split * p.k_per_split;
  int k_end = min(p.K, k_begin + p.k_per_split);
  if (BN == 64 && p.ep.tile_krange) {             // block-diagonal weights: structurally-zero K-tiles of this column tile are skipped
    k_begin = max(k_begin, p.ep.tile_krange[2 * tile_x]);
    k_end = min(k_end, p.ep.tile_krange[2 * tile_x + 1]);
  }
  const int nk = max(0, (k_end - k_begin) / BK);

  unsigned offa[Stg<BM>::ITEMS], offb[Stg<BN>::ITEMS];
  int lda_[Stg<BM>::ITEMS], ldb_[Stg<BN>::ITEMS], rowa[Stg<BM>::ITEMS], rowb[Stg<BN>::ITEMS];
  stage_offsets<BM, AKM, !DB>(p.lda, m0, p.M, offa, lda_, rowa, tid);
  stage_offsets<BN, BKM, !DB>(p.ldb, n0, p.N, offb, ldb_, rowb, tid);
  const float* ca = AKM ? p.A + (long long)k_begin * p.lda + m0 : p.A + (long long)m0 * p.lda + k_begin;
  const float* cb = BKM ? p.B + (long long)k_begin * p.ldb + n0 : p.B + (long long)n0 * p.ldb + k_begin;
  const long long step_a = AKM ? (long long)BK * p.lda : BK, step_b = BKM ? (long long)BK * p.ldb : BK;
  // fragment offsets of this lane (bf16 elements inside a plane): MFMA step s reads chunk q = 2 s + h of row r
  int fa[TM][2], fb[TN][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i][s] = lds_off<!DB>(wm * (BM / 2) + i * 32 + r, 2 * s + h);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j][s] = lds_off<!DB>(wn * (BN / 2) + j * 32 + r, 2 * s + h);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // bias gradient riding on a dW product (K-major A): row sums of the fp32 A tile, taken by the first n-tile from its staged
  // registers (each thread owns 8 k of its rows; the 4 waves hold the 4 k-chunks of a row)
  const bool do_rowsum = AKM && p.ep.a_rowsum && tile_x == 0;
  float rowsum[Stg<BM>::ITEMS];
#pragma unroll
  for (int i = 0; i < Stg<BM>::ITEMS; ++i) rowsum[i] = 0.f;

  float va[Stg<BM>::ITEMS][8], vb[Stg<BN>::ITEMS][8];
  // `real`: the staged registers hold a K-tile that has not been staged before.  The branch-free two-stage loop re-stages the
  // final tile in its last iteration (nobody reads those planes); the bias-gradient rider must count every tile exactly once.
  auto stage_store = [&](int stage, bool real) {          // staged registers -> bf16 planes of `stage`
    if (do_rowsum) {
#pragma unroll
      for (int i = 0; i < Stg<BM>::ITEMS; ++i) {
        float t = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) t += va[i][e];
        rowsum[i] += real ? t : 0.f;
      }
    }
    __bf16* const a_pl = planes + stage * STAGE;
    __bf16* const b_pl = a_pl + NP * A_PLANE;
#pragma unroll
    for (int i = 0; i < Stg<BM>::ITEMS; ++i) split_store<NP>(a_pl + lda_[i], A_PLANE, va[i]);
#pragma unroll
    for (int i = 0; i < Stg<BN>::ITEMS; ++i) split_store<NP>(b_pl + ldb_[i], B_PLANE, vb[i]);
  };
  auto mfma_step = [&](int stage, int s) {
    const __bf16* const a_pl = planes + stage * STAGE;
    const __bf16* const b_pl = a_pl + NP * A_PLANE;
    bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(a_pl + pl * A_PLANE + fa[i][s]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(b_pl + pl * B_PLANE + fb[j][s]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        // small terms first: they are added to each other before they meet the large partial sum of a0 b0
        if (NP == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][NP - 1], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][NP - 1], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][NP >= 2 ? 1 : 0], bf[j][NP >= 2 ? 1 : 0], acc[i][j], 0, 0, 0);
        }
        if (NP >= 2) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][NP >= 2 ? 1 : 0], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][NP >= 2 ? 1 : 0], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);      // (NP == 1, plain bf16: this product alone)
      }
  };

  if (DB) {
    if (nk > 0) {
      stage_load<BM, AKM>(ca, p.lda, offa, va);
      stage_load<BN, BKM>(cb, p.ldb, offb, vb);
      stage_store(0, true);
      if (nk > 1) {                                        // (nk == 1: the corners stay on tile 0 -- the loop body's unconditional
        ca += step_a; cb += step_b;                        //  re-load must never leave the operand)
        stage_load<BM, AKM>(ca, p.lda, offa, va);
        stage_load<BN, BKM>(cb, p.ldb, offb, vb);
      }
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      // Branch-free body (one basic block, so that the scheduler may place the split VALU work and the LDS stores between the
      // MFMAs): the last iterations re-stage / re-load the final tile instead of skipping -- harmless, nobody reads it.
      const int cur = kt & 1;
      mfma_step(cur, 0);
      stage_store(cur ^ 1, kt + 1 < nk);                   // tile kt+1: registers -> the other stage
      mfma_step(cur, 1);
      const bool more = kt + 2 < nk;                       // tile kt+2: in flight across the whole next iteration
      ca += more ? step_a : 0; cb += more ? step_b : 0;
      stage_load<BM, AKM>(ca, p.lda, offa, va);
      stage_load<BN, BKM>(cb, p.ldb, offb, vb);
#if MANSY_BF16S_SCHED
#pragma unroll
      for (int g = 0; g < 2 * TM * TN * (NP == 3 ? 6 : (NP == 2 ? 3 : 1)); ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, NP == 3 ? 4 : 4, 0);   // a few split VALU ops in its shadow
      }
#endif
      __syncthreads();
    }
  } else {
    if (nk > 0) {
      stage_load<BM, AKM>(ca, p.lda, offa, va);
      stage_load<BN, BKM>(cb, p.ldb, offb, vb);
    }
    for (int kt = 0; kt < nk; ++kt) {
      stage_store(0, true);
      __syncthreads();
      ca += step_a; cb += step_b;
      if (kt + 1 < nk) {                                   // next tile's global loads fly during the MFMAs
        stage_load<BM, AKM>(ca, p.lda, offa, va);
        stage_load<BN, BKM>(cb, p.ldb, offb, vb);
      }
      mfma_step(0, 0);
      mfma_step(0, 1);
      __syncthreads();                                     // every wave is done reading the planes
    }
  }

  if (do_rowsum) {
#pragma unroll
    for (int i = 0; i < Stg<BM>::ITEMS; ++i)
      if (m0 + rowa[i] < p.M) atomicAdd(p.ep.a_rowsum + m0 + rowa[i], rowsum[i]);
  }
  gemm_epilogue<BM, BN, SMEM_FLOATS>(p, acc, smem, m0, n0, tid, split, p.C);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x6, 128 x 128 tile, HALF K-tiles (16 k = one MFMA step) so that TWO stages of three planes fit beside a second workgroup:
// 2 x 3 x (128 + 128) rows x 32 B = 48 KB (the 32-k two-stage image is 96 KB: one workgroup per CU, measured slower than one
// stage with two).  Same pipeline as the two-stage loop above: split + ds_write of tile t+1 between the MFMAs of tile t, one
// barrier per (half) K-tile, global loads of tile t+2 across a whole iteration.  LDS rows of 32 B, chunk q (0 / 1) at slot
// q ^ ((row >> 3) & 1): the 16 lanes of a ds_read_b128 group then hit 16 different 16-B slots.
__device__ __forceinline__ int lds_off16(int row, int q) { return row * 16 + ((q ^ ((row >> 3) & 1)) << 3); }

template <bool AKM, bool BKM>
__global__ __launch_bounds__(NT, 2) void gemm_bf16x6_k16_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, NP = 3, KT = 16, TM = 2, TN = 2;
  constexpr int A_PLANE = BM * KT, B_PLANE = BN * KT;                     // bf16 elements
  constexpr int STAGE = NP * (A_PLANE + B_PLANE);
  constexpr int C_FLOATS = BM * (BN + 4);
  constexpr int SMEM_FLOATS = (2 * STAGE / 2) > C_FLOATS ? (2 * STAGE / 2) : C_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SMEM_FLOATS];
  __bf16* const planes = reinterpret_cast<__bf16*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y, split;
  {   // XCD-aware bijective remap over the whole 3-D grid, K split slowest (see gemm_f32_dma_kernel)
    const int per_split = gridDim.x * gridDim.y, nwg = per_split * gridDim.z;
    const int orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    split = t / per_split; t -= split * per_split;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int k_begin = split * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int nk = max(0, (k_end - k_begin) / KT);

  // one (row, 8-k chunk) item per thread and operand
  auto item = [&](bool kmaj, int ld, int row0, int nrows, unsigned& off, int& ldso, int& srow) {
    int kc;
    if (!kmaj) { srow = tid >> 1; kc = tid & 1; off = (unsigned)((min(row0 + srow, nrows - 1) - row0) * ld + kc * 8); }
    else { srow = lane + 64 * (wave >> 1); kc = wave & 1; off = (unsigned)(kc * 8 * ld + (min(row0 + srow, nrows - 1) - row0)); }
    ldso = lds_off16(srow, kc);
  };
  unsigned offa, offb; int lda_, ldb_, rowa, rowb;
  item(AKM, p.lda, m0, p.M, offa, lda_, rowa);
  item(BKM, p.ldb, n0, p.N, offb, ldb_, rowb);
  const float* ca = AKM ? p.A + (long long)k_begin * p.lda + m0 : p.A + (long long)m0 * p.lda + k_begin;
  const float* cb = BKM ? p.B + (long long)k_begin * p.ldb + n0 : p.B + (long long)n0 * p.ldb + k_begin;
  const long long step_a = AKM ? (long long)KT * p.lda : KT, step_b = BKM ? (long long)KT * p.ldb : KT;
  int fa[TM], fb[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) fa[i] = lds_off16(wm * (BM / 2) + i * 32 + r, h);
#pragma unroll
  for (int j = 0; j < TN; ++j) fb[j] = lds_off16(wn * (BN / 2) + j * 32 + r, h);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const bool do_rowsum = AKM && p.ep.a_rowsum && tile_x == 0;
  float rowsum = 0.f;
  float va[8], vb[8];
  auto load8 = [&](bool kmaj, const float* corner, int ld, unsigned off, float (&v)[8]) {
    if (!kmaj) {
      const float4* s = reinterpret_cast<const float4*>(corner + off);
      const float4 a = s[0], b = s[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) v[kk] = (corner + (long long)kk * ld)[off];
    }
  };
  auto stage_store = [&](int stage, bool real) {          // real: see gemm_bf16s_kernel (the last iteration re-stages the final tile)
    if (do_rowsum) {
      float t = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) t += va[e];
      rowsum += real ? t : 0.f;
    }
    __bf16* const a_pl = planes + stage * STAGE;
    __bf16* const b_pl = a_pl + NP * A_PLANE;
    split_store<NP>(a_pl + lda_, A_PLANE, va);
    split_store<NP>(b_pl + ldb_, B_PLANE, vb);
  };
  auto mfma_step = [&](int stage) {
    const __bf16* const a_pl = planes + stage * STAGE;
    const __bf16* const b_pl = a_pl + NP * A_PLANE;
    bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(a_pl + pl * A_PLANE + fa[i]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(b_pl + pl * B_PLANE + fb[j]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
  };

  if (nk > 0) {
    load8(AKM, ca, p.lda, offa, va);
    load8(BKM, cb, p.ldb, offb, vb);
    stage_store(0, true);
    if (nk > 1) {                                          // (nk == 1: the corners stay on tile 0)
      ca += step_a; cb += step_b;
      load8(AKM, ca, p.lda, offa, va);
      load8(BKM, cb, p.ldb, offb, vb);
    }
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {                        // branch-free body: the last iterations re-stage / re-load the final tile
    const int cur = kt & 1;
    mfma_step(cur);
    stage_store(cur ^ 1, kt + 1 < nk);
    const bool more = kt + 2 < nk;
    ca += more ? step_a : 0; cb += more ? step_b : 0;
    load8(AKM, ca, p.lda, offa, va);
    load8(BKM, cb, p.ldb, offb, vb);
    __syncthreads();
  }
  if (do_rowsum && m0 + rowa < p.M) atomicAdd(p.ep.a_rowsum + m0 + rowa, rowsum);
  gemm_epilogue<BM, BN, SMEM_FLOATS>(p, acc, smem, m0, n0, tid, split, p.C);
}

// ---------------------------------------------------------------------------------------------------------------------
// Pre-split B: the weights' bf16 planes exist in HBM (mansy_launch_weight_planes, once per step), so the B tiles go HBM/L2 -> LDS by
// LDS-DMA -- no split VALU work, no staging registers and no ds_write for that operand (the ds_write path, ~80 B/clk/CU, is what
// bounds the in-loop split: removing B's writes measured -17 % on the forward / dX shapes).  A (activations, K-contiguous) is
// split in the loop as above.  Two LDS stages in the compact swizzled layout; the DMA of tile t+1 is issued first thing in
// iteration t (its stage was released by the previous barrier) and has the whole iteration to land.
template <int BM, int BN, int NP>
__global__ __launch_bounds__(NT, 2) void gemm_bf16p_kernel(GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64, PB = BN / 16;                 // PB: 1-KiB DMA pieces (16 rows x 64 B) per B plane
  constexpr int A_PLANE = BM * 32, B_PLANE = BN * 32;                     // bf16 elements
  constexpr int STAGE = NP * (A_PLANE + B_PLANE);
  constexpr int STAGE_FLOATS = 2 * STAGE / 2;
  constexpr int C_FLOATS = BM * (BN + 4);
  constexpr int SMEM_FLOATS = STAGE_FLOATS > C_FLOATS ? STAGE_FLOATS : C_FLOATS;
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];
  __bf16* const planes = reinterpret_cast<__bf16*>(smem);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y;
  {   // XCD-aware bijective remap (no split-K on this path: forward and dX products only)
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int nk = p.K / BK;

  unsigned offa[Stg<BM>::ITEMS];
  int lda_[Stg<BM>::ITEMS], rowa[Stg<BM>::ITEMS];
  stage_offsets<BM, false, false>(p.lda, m0, p.M, offa, lda_, rowa, tid);
  const float* ca = p.A + (long long)m0 * p.lda;
  // B planes: this wave's pieces wave, wave + 4, ... of every plane; lane -> (row, 16-byte slot), source chunk = slot ^ key(row)
  const unsigned short* cb = p.ep.b_planes + (long long)n0 * p.ep.b_planes_ld;
  unsigned vob[(PB + 3) / 4];
#pragma unroll
  for (int i = 0; i < (PB + 3) / 4; ++i) {
    const int row = (wave + 4 * i) * 16 + (lane >> 2), slot = lane & 3;
    const int q = slot ^ ((row >> 2) & 3);
    vob[i] = (unsigned)(((min(n0 + row, p.N - 1) - n0) * p.ep.b_planes_ld + q * 8) * 2);
  }
  const unsigned lds_b0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)(NP * A_PLANE * 2) + (unsigned)wave * 1024u);
  auto dma_b = [&](int stage, const unsigned short* corner) {
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int i = 0; i < (PB + 3) / 4; ++i)
        if (wave + 4 * i < PB)
          glds16(vob[i], corner + (long long)pl * p.ep.b_plane_stride, lds_b0 + (unsigned)(stage * STAGE * 2) + (unsigned)(pl * B_PLANE * 2) + (unsigned)i * 4096u);
  };
  int fa[TM][2], fb[TN][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i][s] = lds_off<false>(wm * (BM / 2) + i * 32 + r, 2 * s + h);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j][s] = lds_off<false>(wn * (BN / 2) + j * 32 + r, 2 * s + h);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float va[Stg<BM>::ITEMS][8];
  auto stage_store_a = [&](int stage) {
    __bf16* const a_pl = planes + stage * STAGE;
#pragma unroll
    for (int i = 0; i < Stg<BM>::ITEMS; ++i) split_store<NP>(a_pl + lda_[i], A_PLANE, va[i]);
  };
  auto mfma_step = [&](int stage, int s) {
    const __bf16* const a_pl = planes + stage * STAGE;
    const __bf16* const b_pl = a_pl + NP * A_PLANE;
    bf16x8 af[TM][NP], bf[TN][NP];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) af[i][pl] = *reinterpret_cast<const bf16x8*>(a_pl + pl * A_PLANE + fa[i][s]);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int pl = 0; pl < NP; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(b_pl + pl * B_PLANE + fb[j][s]);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (NP == 3) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
        }
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
      }
  };

  if (nk > 0) {
    dma_b(0, cb);
    stage_load<BM, false>(ca, p.lda, offa, va);
    stage_store_a(0);
    if (nk > 1) { ca += BK; stage_load<BM, false>(ca, p.lda, offa, va); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    cb += (kt + 1 < nk) ? BK : 0;                          // B tile kt+1 (the last iteration re-fetches its own tile: harmless)
    dma_b(cur ^ 1, cb);
    mfma_step(cur, 0);
    stage_store_a(cur ^ 1);                                // A tile kt+1: registers -> the other stage, between the MFMAs
    mfma_step(cur, 1);
    ca += (kt + 2 < nk) ? BK : 0;                          // A tile kt+2: in flight across the whole next iteration
    stage_load<BM, false>(ca, p.lda, offa, va);
    // the DMA pieces are older than the A loads just issued (Stg<BM>::ITEMS x 2 dwordx4 per lane): wait for all but those
    if (Stg<BM>::ITEMS == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    __syncthreads();
  }
  gemm_epilogue<BM, BN, SMEM_FLOATS>(p, acc, smem, m0, n0, tid, 0, p.C);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3, pre-split B, and A split AT FRAGMENT READ (round 3): the fp32 A tile goes HBM/L2 -> LDS by LDS-DMA exactly like the exact-fp32
// loop's (gemm_f32.hip: [rows][32 floats] image, 128-B rows, XOR swizzle applied on the source address), so the loop has no staging
// registers, no global_load -> VGPR -> ds_write pass and no cooperative split phase at all; each wave converts the 8 floats of a
// fragment into its two bf16 planes in registers right before the MFMAs that consume them (v_cvt_pk_bf16_f32 + one subtraction per
// element, issued in the shadow of the MFMAs).  The conversion is done once per wave that needs the fragment (2 x redundant over the
// 2 x 2 wave grid) -- the price of removing the ds_write path (~80 B/clk/CU), which is what bounded the in-loop split.  One barrier
// per K-tile; the DMA of tile t+1 is issued right after the barrier that retires tile t-1 and has the whole MFMA phase to land.
template <int BM, int BN>
__global__ __launch_bounds__(NT, 2) void gemm_bf16f_kernel(GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64, PA = BM / 32, PB = BN / 16, PBW = (PB + 3) / 4;
  constexpr int A_BYTES = BM * BK * 4, B_PLANE = BN * 32, B_PLANE_BYTES = B_PLANE * 2;      // B_PLANE in bf16 elements
  constexpr int STAGE_BYTES = A_BYTES + 2 * B_PLANE_BYTES;
  constexpr int C_FLOATS = BM * (BN + 4);
  constexpr int SMEM_FLOATS = (2 * STAGE_BYTES / 4) > C_FLOATS ? (2 * STAGE_BYTES / 4) : C_FLOATS;
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y;
  {   // XCD-aware bijective remap (no split-K on this path: forward and dX products only)
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int nk = p.K / BK;

  // A: fp32 image [BM][32], piece i of this wave = rows i*32 + wave*8 .. +8 (128 B each); k-chunk c of a row sits in slot c ^ ((row >> 1) & 7)
  unsigned voa[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voa[i] = (unsigned)((min(m0 + row, p.M - 1) - m0) * p.lda + c * 4) * 4u;
  }
  // B planes: image [BN][32 bf16] per plane (64-B rows), piece = 16 rows; 16-byte chunk q of a row in slot q ^ ((row >> 2) & 3)
  unsigned vob[PBW];
#pragma unroll
  for (int i = 0; i < PBW; ++i) {
    const int row = (wave + 4 * i) * 16 + (lane >> 2), slot = lane & 3;
    const int q = slot ^ ((row >> 2) & 3);
    vob[i] = (unsigned)(((min(n0 + row, p.N - 1) - n0) * p.ep.b_planes_ld + q * 8) * 2);
  }
  const float* ca = p.A + (long long)m0 * p.lda;
  const unsigned short* cb = p.ep.b_planes + (long long)n0 * p.ep.b_planes_ld;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * 1024u);
  auto dma = [&](int stage, const float* a_corner, const unsigned short* b_corner) {
    const unsigned base = lds0 + (unsigned)(stage * STAGE_BYTES);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(voa[i], a_corner, base + (unsigned)i * 4096u);
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int i = 0; i < PBW; ++i)
        if (PB % 4 == 0 || wave + 4 * i < PB)
          glds16(vob[i], b_corner + (long long)pl * p.ep.b_plane_stride, base + (unsigned)(A_BYTES + pl * B_PLANE_BYTES) + (unsigned)i * 4096u);
  };
  // fragment addresses: A in floats inside the A image (two 16-byte slots per k-step), B in bf16 elements inside a plane
  int fa[TM][2][2], fb[TN][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (BM / 2) + i * 32 + r, sw = (row >> 1) & 7, c0 = 4 * s + 2 * h;
      fa[i][s][0] = row * BK + ((c0 + 0) ^ sw) * 4;
      fa[i][s][1] = row * BK + ((c0 + 1) ^ sw) * 4;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j][s] = lds_off<false>(wn * (BN / 2) + j * 32 + r, 2 * s + h);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nk > 0) dma(0, ca, cb);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt have landed ...
    __builtin_amdgcn_s_barrier();                        // ... and everyone's; everyone is done reading tile kt-1
    asm volatile("" ::: "memory");
    ca += BK; cb += BK;
    if (kt + 1 < nk) dma(cur ^ 1, ca, cb);
    const float* a_l = smem + cur * (STAGE_BYTES / 4);
    const __bf16* b_l = reinterpret_cast<const __bf16*>(a_l) + A_BYTES / 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(b_l + fb[j][s]);
        bl[j] = *reinterpret_cast<const bf16x8*>(b_l + B_PLANE + fb[j][s]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 v0 = *reinterpret_cast<const float4*>(a_l + fa[i][s][0]);
        const float4 v1 = *reinterpret_cast<const float4*>(a_l + fa[i][s][1]);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 t0 = (__bf16)v[e];
          ah[i][e] = t0;
          al[i][e] = (__bf16)(v[e] - (float)t0);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads retired before the barrier that frees this buffer
  }
  __syncthreads();                                        // staging LDS idle: the epilogue reuses it
  gemm_epilogue<BM, BN, SMEM_FLOATS>(p, acc, smem, m0, n0, tid, 0, p.C);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3 / bf16x6 (NP = 2 / 3 planes), pre-split B, A split at fragment read, on an NS-stage ring (late round 3): gemm_bf16f_kernel's loop with NS - 1 K-tiles in
// flight instead of one.  For the 64 x 64 tiles of the [4 096-row] decoder products: one K-tile of such a tile is 6 MFMAs per wave
// (~200 cycles), a cold fetch (the activations were written by the previous launch, usually on another XCD) is ~2 000, so with one
// tile in flight the loop runs at one fetch latency per K-tile (in the step: 20 us per product against 14 us with warm operands).
// 16 KB (20 KB with three planes) per stage: three stages and two workgroups per CU fit the LDS.  Measured in the step (tools/bf16_ring_ab.py,
// variants interleaved in one process): bf16x3 train step 15.87 -> 14.53 ms with three stages (14.69 with four); results bit-identical to
// gemm_bf16p_kernel's.
//   wait for this wave's pieces of tile t (counted vmcnt: the younger tiles' pieces stay in flight) -> raw s_barrier (everyone's pieces
//   of t have landed, everyone is done reading t - 1) -> DMA of tile t + NS - 1 into the stage t - 1 occupied -> fragments + MFMAs of t.
template <int BM, int BN, int NS, int NP>
__global__ __launch_bounds__(NT, 2) void gemm_bf16h_kernel(GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64, PA = BM / 32, PB = BN / 16, PBW = (PB + 3) / 4, D = NS - 1;
  constexpr int A_BYTES = BM * BK * 4, B_PLANE = BN * 32, B_PLANE_BYTES = B_PLANE * 2;      // B_PLANE in bf16 elements
  constexpr int STAGE_BYTES = A_BYTES + NP * B_PLANE_BYTES;
  constexpr int C_FLOATS = BM * (BN + 4);
  constexpr int SMEM_FLOATS = (NS * STAGE_BYTES / 4) > C_FLOATS ? (NS * STAGE_BYTES / 4) : C_FLOATS;
  constexpr int PPT = PA + NP * PBW;                                                         // DMA pieces per wave and K-tile
  static_assert(NP >= 1 && NP <= 3 && PB % 4 == 0 && NS >= 2 && NS <= 4 && PPT * (D - 1 > 0 ? D - 1 : 0) <= 15, "piece counts must fit the counted waits");
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y;
  {   // XCD-aware bijective remap (no split-K on this path: forward and dX products only)
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int nk = p.K / BK;

  unsigned voa[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voa[i] = (unsigned)((min(m0 + row, p.M - 1) - m0) * p.lda + c * 4) * 4u;
  }
  unsigned vob[PBW];
#pragma unroll
  for (int i = 0; i < PBW; ++i) {
    const int row = (wave + 4 * i) * 16 + (lane >> 2), slot = lane & 3;
    const int q = slot ^ ((row >> 2) & 3);
    vob[i] = (unsigned)(((min(n0 + row, p.N - 1) - n0) * p.ep.b_planes_ld + q * 8) * 2);
  }
  const float* const ca = p.A + (long long)m0 * p.lda;
  const unsigned short* const cb = p.ep.b_planes + (long long)n0 * p.ep.b_planes_ld;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * 1024u);
  auto dma = [&](int stage, int kt) {
    const unsigned base = lds0 + (unsigned)(stage * STAGE_BYTES);
    const float* a_corner = ca + (long long)kt * BK;
    const unsigned short* b_corner = cb + (long long)kt * BK;
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(voa[i], a_corner, base + (unsigned)i * 4096u);
#pragma unroll
    for (int pl = 0; pl < NP; ++pl)
#pragma unroll
      for (int i = 0; i < PBW; ++i)
        glds16(vob[i], b_corner + (long long)pl * p.ep.b_plane_stride, base + (unsigned)(A_BYTES + pl * B_PLANE_BYTES) + (unsigned)i * 4096u);
  };
  int fa[TM][2][2], fb[TN][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * (BM / 2) + i * 32 + r, sw = (row >> 1) & 7, c0 = 4 * s + 2 * h;
      fa[i][s][0] = row * BK + ((c0 + 0) ^ sw) * 4;
      fa[i][s][1] = row * BK + ((c0 + 1) ^ sw) * 4;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j][s] = lds_off<false>(wn * (BN / 2) + j * 32 + r, 2 * s + h);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) dma(d, d);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // tiles still wanted in flight after tile kt has landed: min(D - 1, nk - 1 - kt) of them, PPT pieces each (the wait count is an immediate)
    const int ahead = nk - 1 - kt;
    if (D >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
    else if (D >= 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + D < nk) dma(cur == 0 ? NS - 1 : cur - 1, kt + D);       // into the stage tile kt - 1 occupied
    const float* a_l = smem + cur * (STAGE_BYTES / 4);
    const __bf16* b_l = reinterpret_cast<const __bf16*>(a_l) + A_BYTES / 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af[TM][NP], bf[TN][NP];                      // plane 0 = leading bf16 part, 1 / 2 = first / second remainder
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int pl = 0; pl < NP; ++pl) bf[j][pl] = *reinterpret_cast<const bf16x8*>(b_l + pl * B_PLANE + fb[j][s]);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 v0 = *reinterpret_cast<const float4*>(a_l + fa[i][s][0]);
        const float4 v1 = *reinterpret_cast<const float4*>(a_l + fa[i][s][1]);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {                     // split_store's arithmetic, in registers
          const __bf16 t0 = (__bf16)v[e];
          af[i][0][e] = t0;
          if (NP >= 2) {
            const float r1 = v[e] - (float)t0;
            const __bf16 t1 = (__bf16)r1;
            af[i][NP >= 2 ? 1 : 0][e] = t1;
            if (NP == 3) af[i][NP - 1][e] = (__bf16)(r1 - (float)t1);
          }
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {                    // smallest products first, the order of gemm_bf16p_kernel (results bit-identical to it)
          if (NP == 3) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][NP - 1], bf[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][NP - 1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][NP >= 2 ? 1 : 0], bf[j][NP >= 2 ? 1 : 0], acc[i][j], 0, 0, 0);
          }
          if (NP >= 2) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][NP >= 2 ? 1 : 0], bf[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][NP >= 2 ? 1 : 0], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);      // (NP == 1, plain bf16: this product alone)
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads retired before the barrier that frees this buffer
    cur = cur == NS - 1 ? 0 : cur + 1;
  }
  __syncthreads();                                        // staging LDS idle (every DMA was waited for): the epilogue reuses it
  gemm_epilogue<BM, BN, SMEM_FLOATS>(p, acc, smem, m0, n0, tid, 0, p.C);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3, 256 x 128 tile, EIGHT waves, THREE LDS stages (round 3).  The 128 x 128 loops above are bound by the LDS fill: with two
// stages per workgroup one K-tile per workgroup is in flight (2 x 32 KB per CU), and in-flight bytes / loaded round trip is all the
// fill rate there is.  This loop is two of those workgroups fused -- waves 0-3 own rows 0..127, waves 4-7 rows
// 128..255 -- sharing ONE B tile (0.75 x the bytes per flop) on a three-stage ring with TWO K-tiles in flight (96 KB per CU):
//   wait for this wave's pieces of tile t (s_waitcnt vmcnt(6): tile t+1's six pieces stay in flight) -> raw s_barrier (everyone's
//   pieces of t have landed; everyone is done reading t-1) -> DMA of tile t+2 into the stage t-1 occupied -> fragments + MFMAs of t.
// No __syncthreads() inside the loop (its fence would drain the DMA queue); 144 KB of LDS, one workgroup (2 waves per SIMD) per CU.
// A is staged in fp32 and split at fragment read, B comes pre-split, as in gemm_bf16f_kernel.
__global__ __launch_bounds__(512) void gemm_bf16g_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 128, TM = 2, TN = 2, NW = 8, PA = BM / (8 * NW), NS = 3;      // PA: A pieces (8 rows x 128 B) per wave and K-tile
  constexpr int A_BYTES = BM * BK * 4, B_PLANE = BN * 32, B_PLANE_BYTES = B_PLANE * 2;
  constexpr int STAGE_BYTES = A_BYTES + 2 * B_PLANE_BYTES;                                   // 48 KB
  constexpr int C_HALF_FLOATS = 128 * (BN + 4);
  constexpr int SMEM_FLOATS = NS * STAGE_BYTES / 4;
  static_assert(2 * C_HALF_FLOATS <= SMEM_FLOATS && BN / 16 == NW, "two half-tile epilogues must fit the ring; one B piece per wave and plane");
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, half = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y;
  {
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int nk = p.K / BK;

  unsigned voa[PA];
#pragma unroll
  for (int i = 0; i < PA; ++i) {          // piece q = i * 8 + wave: rows q * 8 .. + 8
    const int row = (i * NW + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voa[i] = (unsigned)((min(m0 + row, p.M - 1) - m0) * p.lda + c * 4) * 4u;
  }
  unsigned vob;
  {
    const int row = wave * 16 + (lane >> 2), slot = lane & 3;
    const int q = slot ^ ((row >> 2) & 3);
    vob = (unsigned)(((min(n0 + row, p.N - 1) - n0) * p.ep.b_planes_ld + q * 8) * 2);
  }
  const float* ca = p.A + (long long)m0 * p.lda;
  const unsigned short* cb = p.ep.b_planes + (long long)n0 * p.ep.b_planes_ld;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * 1024u);
  auto dma = [&](int stage, const float* a_corner, const unsigned short* b_corner) {
    const unsigned base = lds0 + (unsigned)(stage * STAGE_BYTES);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(voa[i], a_corner, base + (unsigned)i * (NW * 1024u));
    glds16(vob, b_corner, base + (unsigned)A_BYTES);
    glds16(vob, b_corner + p.ep.b_plane_stride, base + (unsigned)(A_BYTES + B_PLANE_BYTES));
  };
  int fa[TM][2][2], fb[TN][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = half * 128 + wm * 64 + i * 32 + r, sw = (row >> 1) & 7, c0 = 4 * s + 2 * h;
      fa[i][s][0] = row * BK + ((c0 + 0) ^ sw) * 4;
      fa[i][s][1] = row * BK + ((c0 + 1) ^ sw) * 4;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j][s] = lds_off<false>(wn * 64 + j * 32 + r, 2 * s + h);
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  if (nk > 0) dma(0, ca, cb);
  if (nk > 1) dma(1, ca + BK, cb + BK);
  int cur = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");       // tile kt landed (this wave's pieces); tile kt+1's 6 stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + 2 < nk) dma(cur == 0 ? 2 : cur - 1, ca + (long long)(kt + 2) * BK, cb + (long long)(kt + 2) * BK);     // into the stage tile kt-1 occupied
    const float* a_l = smem + cur * (STAGE_BYTES / 4);
    const __bf16* b_l = reinterpret_cast<const __bf16*>(a_l) + A_BYTES / 2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = *reinterpret_cast<const bf16x8*>(b_l + fb[j][s]);
        bl[j] = *reinterpret_cast<const bf16x8*>(b_l + B_PLANE + fb[j][s]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 v0 = *reinterpret_cast<const float4*>(a_l + fa[i][s][0]);
        const float4 v1 = *reinterpret_cast<const float4*>(a_l + fa[i][s][1]);
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 t0 = (__bf16)v[e];
          ah[i][e] = t0;
          al[i][e] = (__bf16)(v[e] - (float)t0);
        }
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    cur = cur == 2 ? 0 : cur + 1;
  }
  __syncthreads();
  // each half of the workgroup is a 128 x 128 tile of the shared epilogue, with its own piece of the (now idle) ring
  gemm_epilogue<128, 128, C_HALF_FLOATS>(p, acc, smem + half * C_HALF_FLOATS, m0 + half * 128, n0, tid & 255, 0, p.C);
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16x3, 256 x 128 tile, TWELVE waves with fixed roles (round 4): 8 consumer waves (ds_read + split + MFMA only) and 4 loader waves
// (LDS-DMA only) on a three-stage ring with two K-tiles in flight.
// Why: in the eight-wave loop above (gemm_bf16g_kernel) every wave issues its own DMA pieces between its MFMAs (an LDS-DMA piece costs
// 60-185 issue cycles of the issuing wave: six per wave and K-tile stall that SIMD's matrix pipe) and every A fragment is read and split
// by the two waves of the 2 x 2 grid that need it.  Here
//   * loader waves issue all 48 pieces of a K-tile (12 each) and park at `s_waitcnt vmcnt` / `s_barrier`: the consumers' instruction
//     streams contain no vector-memory instruction at all;
//   * consumer wave w owns rows 32 w .. 32 w + 31 and ALL 128 columns (1 x 4 blocks of 32 x 32): each A fragment is read and split
//     once per workgroup; two consumer waves per SIMD cover each other's ds_read latency (the loader is the third wave there).
// Ring protocol (one raw s_barrier per K-tile, all 12 waves):
//   loader:   wait until its pieces of tile t have landed (vmcnt(12): tile t+1's twelve stay in flight) -> barrier t -> DMA of tile t+2
//             into the stage tile t-1 occupied (its readers have passed barrier t, i.e. are done with it)
//   consumer: barrier t -> fragments + MFMAs of tile t (every read retired before it reaches barrier t+1)
// Same products in the same order per output element as gemm_bf16f / gemm_bf16g / gemm_bf16p (lo x hi, hi x lo, hi x hi per 16-k step):
// results bit-identical to them.  Epilogue: the C tile goes through the idle ring in the row-major image of gemm_epilogue_rows.
// Measured (profiles/r04_gemm_bench_bf16k.txt, r04_bf16k_sq_pmc.txt; same box, interleaved): [40 960, 512, 512] 97 -> 82 us, [40 960, 1536, 512]
// 271 -> 223 us, matrix pipes busy 0.28 -> 0.34 / 0.33 -> 0.41 of the elapsed cycles at the 2.0-2.2 GHz the chip holds under this load.
// Tried and dropped (same files): an L2 prefetch stream of the A lines 4 / 6 K-tiles ahead (one LDS-DMA dword per consumer wave and
// K-tile into a scratch pad): 3-8 % SLOWER -- the staging-only form of this loop already fills at 70 GB/s per CU at K = 512, the fill is
// not the bound; starting half of the first round's workgroups half a tile late (to de-phase the epilogue store bursts): no effect.
__global__ __launch_bounds__(768) void gemm_bf16k_kernel(GemmParams p) {
  constexpr int BM = 256, BN = 128, NCW = 8, NS = 3;
  constexpr int A_BYTES = BM * BK * 4, B_PLANE = BN * 32, B_PLANE_BYTES = B_PLANE * 2;
  constexpr int STAGE_BYTES = A_BYTES + 2 * B_PLANE_BYTES;                                   // 48 KB
  constexpr int RING_FLOATS = NS * STAGE_BYTES / 4;
  constexpr int CLD = BN + 4;
  static_assert(BM * CLD <= RING_FLOATS, "the C image must fit the ring");
  __shared__ __attribute__((aligned(1024))) float smem[RING_FLOATS];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int tile_x, tile_y;
  {
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int nk = p.K / BK;
  const float* const ca = p.A + (long long)m0 * p.lda;
  const unsigned short* const cb = p.ep.b_planes + (long long)n0 * p.ep.b_planes_ld;
  const unsigned lds_base = (unsigned)(uintptr_t)smem;

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  if (wave >= NCW) {
    // ------------------------------------------------------------------ loader: pieces lw, lw + 4, ... of every image
    const int lw = wave - NCW;
    unsigned voa[8], vob[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) {           // A piece q = 4 i + lw: rows 8 q .. 8 q + 7 (128 B each); k-chunk c of a row sits in slot c ^ ((row >> 1) & 7)
      const int row = (i * 4 + lw) * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      voa[i] = (unsigned)((min(m0 + row, p.M - 1) - m0) * p.lda + c * 4) * 4u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {           // B piece b = lw + 4 i of a plane: rows 16 b .. 16 b + 15 (64 B each); chunk q in slot q ^ ((row >> 2) & 3)
      const int row = (lw + 4 * i) * 16 + (lane >> 2), slot = lane & 3;
      const int q = slot ^ ((row >> 2) & 3);
      vob[i] = (unsigned)(((min(n0 + row, p.N - 1) - n0) * p.ep.b_planes_ld + q * 8) * 2);
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)lw * 1024u);
    auto dma = [&](int stage, int kt) {
      const unsigned base = lds0 + (unsigned)(stage * STAGE_BYTES);
      const float* a_corner = ca + (long long)kt * BK;
      const unsigned short* b_corner = cb + (long long)kt * BK;
#pragma unroll
      for (int i = 0; i < 8; ++i) glds16(voa[i], a_corner, base + (unsigned)i * 4096u);
#pragma unroll
      for (int pl = 0; pl < 2; ++pl)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          glds16(vob[i], b_corner + (long long)pl * p.ep.b_plane_stride, base + (unsigned)(A_BYTES + pl * B_PLANE_BYTES) + (unsigned)i * 4096u);
    };
    if (nk > 0) dma(0, 0);
    if (nk > 1) dma(1, 1);
    int st = 2;                                                     // stage of tile t + 2
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");    // this wave's pieces of tile kt have landed; tile kt+1's 12 stay in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                 // barrier kt: tile kt readable; tile kt-1 no longer read
      asm volatile("" ::: "memory");
      if (kt + 2 < nk) dma(st, kt + 2);
      st = st == 2 ? 0 : st + 1;
    }
  } else {
    // ------------------------------------------------------------------ consumer: rows 32 wave .. + 31, all 128 columns
    const int r = lane & 31, h = lane >> 5;
    int fa[2][2], fb[4][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int row = wave * 32 + r, sw = (row >> 1) & 7, c0 = 4 * s + 2 * h;
      fa[s][0] = row * BK + ((c0 + 0) ^ sw) * 4;
      fa[s][1] = row * BK + ((c0 + 1) ^ sw) * 4;
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j][s] = lds_off<false>(j * 32 + r, 2 * s + h);
    }
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      __builtin_amdgcn_s_barrier();                                 // barrier kt
      asm volatile("" ::: "memory");
      const float* a_l = smem + cur * (STAGE_BYTES / 4);
      const __bf16* b_l = reinterpret_cast<const __bf16*>(a_l) + A_BYTES / 2;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 ah, al, bh[4], bl[4];
        const float4 v0 = *reinterpret_cast<const float4*>(a_l + fa[s][0]);
        const float4 v1 = *reinterpret_cast<const float4*>(a_l + fa[s][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          bh[j] = *reinterpret_cast<const bf16x8*>(b_l + fb[j][s]);
          bl[j] = *reinterpret_cast<const bf16x8*>(b_l + B_PLANE + fb[j][s]);
        }
        const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const __bf16 t0 = (__bf16)v[e];
          ah[e] = t0;
          al[e] = (__bf16)(v[e] - (float)t0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[j], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // LDS reads retired before the barrier that frees this stage
      cur = cur == 2 ? 0 : cur + 1;
    }
  }
  __syncthreads();                                                  // ring idle (every DMA waited for, every fragment read)
  if (wave < NCW) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        smem[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + j * 32 + r] = acc[j][e];
  }
  __syncthreads();
  if (wave >= NCW) return;
  gemm_epilogue_rows<BM, BN, NCW * 64>(p, smem, m0, n0, (int)threadIdx.x, p.C);
}

// ---- weights -> bf16 planes (and planes of the transpose), 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void weight_planes_kernel(MansyWPlaneTab tab, unsigned short* __restrict__ out, unsigned short* __restrict__ out_t,
                                                           long long plane_stride, int n_planes) {
  __shared__ float tile[32][33];
  int t = 0, b = blockIdx.x;
  for (; t < tab.n; ++t) {
    const int tiles = ((tab.N[t] + 31) / 32) * ((tab.K[t] + 31) / 32);
    if (b < tiles) break;
    b -= tiles;
  }
  if (t >= tab.n) return;
  const int N = tab.N[t], K = tab.K[t], tk = (K + 31) / 32;
  const int n0 = (b / tk) * 32, k0 = (b % tk) * 32;
  const float* W = tab.w[t];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;               // 32 x 8 threads, 4 rows each
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, k = k0 + tx;
    tile[ty + 8 * i][tx] = (n < N && k < K) ? W[(long long)n * K + k] : 0.f;
  }
  __syncthreads();
  auto emit = [&](float x, unsigned short* dst) {
    const __bf16 t0 = (__bf16)x;
    const float r1 = x - (float)t0;
    const __bf16 t1 = (__bf16)r1;
    dst[0] = __builtin_bit_cast(unsigned short, t0);
    dst[plane_stride] = __builtin_bit_cast(unsigned short, t1);
    if (n_planes == 3) { const float r2 = r1 - (float)t1; dst[2 * plane_stride] = __builtin_bit_cast(unsigned short, (__bf16)r2); }
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + ty + 8 * i, k = k0 + tx;                           // W planes: row n, k contiguous across the 32 lanes
    if (n < N && k < K) emit(tile[ty + 8 * i][tx], out + tab.off[t] + (long long)n * K + k);
    const int kk = k0 + ty + 8 * i, nn = n0 + tx;                         // W^T planes: row k, n contiguous across the 32 lanes
    if (kk < K && nn < N) emit(tile[tx][ty + 8 * i], out_t + tab.off[t] + (long long)kk * N + nn);
  }
}

template <int BM, int BN, int NP, bool DB>
int launch_layouts(const GemmParams& p, int a_kmajor, int b_kmajor, int splits, hipStream_t st) {
  dim3 grid(mansy_ceil_div(p.N, BN), mansy_ceil_div(p.M, BM), splits);
  dim3 block(NT);
  if (!a_kmajor && !b_kmajor) MANSY_GEMM_LAUNCH((gemm_bf16s_kernel<BM, BN, false, false, NP, DB>), grid, block, st, p);
  else if (!a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_bf16s_kernel<BM, BN, false, true, NP, DB>), grid, block, st, p);
  else if (a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_bf16s_kernel<BM, BN, true, true, NP, DB>), grid, block, st, p);
  else MANSY_GEMM_LAUNCH((gemm_bf16s_kernel<BM, BN, true, false, NP, DB>), grid, block, st, p);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

}  // namespace

// tile: 128 -> 128x128, anything else -> 64x64; prec: 1 (plain bf16), 3 (bf16x3) or 6 (bf16x6).  Preconditions as the LDS-DMA loop's.
// bf16x3 runs the two-stage loop (64 KB of planes, two workgroups per CU); bf16x6: 128 x 128 tiles on the half-K-tile two-stage
// loop (48 KB), 64 x 64 tiles on the one-stage loop (two 32-k stages of three planes are 96 KB, i.e. one workgroup per CU, and
// measured 10-20 % slower than one stage with two).
int mansy_gemm_bf16s_dispatch(const GemmParams& p, int tile, int prec, int a_kmajor, int b_kmajor, int splits, hipStream_t st) {
  if (prec == 1) {         // plain bf16 (MANSY_PREC_BF16): one plane per operand, one product -- the two-stage loop with half the LDS of bf16x3
    if (tile == 128) return launch_layouts<128, 128, 1, true>(p, a_kmajor, b_kmajor, splits, st);
    return launch_layouts<64, 64, 1, true>(p, a_kmajor, b_kmajor, splits, st);
  }
  if (prec == 3) {
    if (tile == 128) return launch_layouts<128, 128, 2, true>(p, a_kmajor, b_kmajor, splits, st);
    return launch_layouts<64, 64, 2, true>(p, a_kmajor, b_kmajor, splits, st);
  }
  if (tile == 128) {       // half K-tiles, two stages, two workgroups per CU: 3-14 % faster than one 32-k stage (tools/gemm_bench.py)
    dim3 grid(mansy_ceil_div(p.N, 128), mansy_ceil_div(p.M, 128), splits), block(NT);
    if (!a_kmajor && !b_kmajor) MANSY_GEMM_LAUNCH((gemm_bf16x6_k16_kernel<false, false>), grid, block, st, p);
    else if (!a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_bf16x6_k16_kernel<false, true>), grid, block, st, p);
    else if (a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_bf16x6_k16_kernel<true, true>), grid, block, st, p);
    else MANSY_GEMM_LAUNCH((gemm_bf16x6_k16_kernel<true, false>), grid, block, st, p);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  return launch_layouts<64, 64, 3, false>(p, a_kmajor, b_kmajor, splits, st);
}

// B pre-split into planes (weights): forward / dX products with a K-contiguous A
// Loop variant of the bf16x3 products with pre-split weights (GemmEpilogue::variant low byte, MANSY_VARIANT_BF16(v); default 1):
//   1: A by LDS-DMA, split at fragment read (256 x 128 tiles: gemm_bf16k_kernel's twelve waves with fixed roles; 128 x 128: gemm_bf16f; 64 x 64 tiles:
//      gemm_bf16h_kernel's three-stage ring); 8: as 1 with the round-3 eight-wave loop (gemm_bf16g) on the 256 x 128 tiles; 4: as 1 without any 256 x 128
//      loop; 7: as 1 with the round-2 loop on the 64 x 64 tiles; 0: the round-2 loop.  Every selectable loop computes the same products in the same
//      order (bit-identical results); the timing-only staging / math forms of rounds 3-4 (variants 2 / 3 / 11 / 12, results wrong) and the four-stage
//      ring (6) are gone from the sources (round 6: nothing in the release library may produce a wrong product).

int mansy_gemm_bf16p_dispatch(const GemmParams& p, int tile, int prec, hipStream_t st) {
  const int bvar = mansy_var_bf16(p.ep.variant);
  if (prec == 1) {
    // plain bf16 with the weight's leading plane pre-converted: the ring loop (A staged in fp32 by LDS-DMA and rounded at fragment read, B's ONE plane by
    // LDS-DMA), 128 x 128 or 64 x 64 tiles (gemm_dispatch sends operands it cannot take to the general staging loop)
    MANSY_REQUIRE((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && p.lda % 4 == 0, "bf16 planes path: A must be 16-byte aligned with lda %% 4 == 0 (the dispatcher checks)");
    dim3 block(NT);
    if (tile >= 128) { dim3 grid(mansy_ceil_div(p.N, 128), mansy_ceil_div(p.M, 128), 1); MANSY_GEMM_LAUNCH((gemm_bf16h_kernel<128, 128, 3, 1>), grid, block, st, p); }
    else { dim3 grid(mansy_ceil_div(p.N, 64), mansy_ceil_div(p.M, 64), 1); MANSY_GEMM_LAUNCH((gemm_bf16h_kernel<64, 64, 3, 1>), grid, block, st, p); }
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  if (prec == 3 && bvar >= 1 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && p.lda % 4 == 0) {
    if (tile == 256 && p.K < 3 * BK) tile = 128;
    dim3 block(NT);
    // measured per shape (tools/gemm_bench.py --planes, profiles/r03_gemm_bench_bf16f.txt): 128 x 128 tiles 3-9 % faster than the
    // round-2 loop on the [40 960-row] products; the two-stage 64 x 64 instance ([4 096-row] decoder products) 6 % slower IN ISOLATION
    // (one K-tile of a 64 x 64 tile is 6 MFMAs per wave: the second, redundant fragment conversion is no longer hidden) -- but inside
    // the step those products are bound by the latency of their cold operands, which the three-stage ring below (gemm_bf16h_kernel)
    // covers; the 128 x 64 instance loses to both at every shape and is reachable only as force_tile 96
    const bool big = tile == 256 || (tile == 128 && (bvar == 1 || bvar >= 8) && (long long)mansy_ceil_div(p.M, 256) * mansy_ceil_div(p.N, 128) >= 256);
    if (big && bvar != 8 && p.c_vec_ok && !p.ep.accumulate && p.ep.split_slab == 0) {
      // round 4: twelve waves with fixed roles (8 consumers + 4 loaders); variant 8 = the eight-wave loop below (A/B runs)
      dim3 grid(mansy_ceil_div(p.N, 128), mansy_ceil_div(p.M, 256), 1);
      MANSY_GEMM_LAUNCH(gemm_bf16k_kernel, grid, dim3(768), st, p);
      MANSY_LAUNCH_CHECK(); return MANSY_OK;
    }
    if (big) {
      // enough 256 x 128 tiles for every CU: the eight-wave three-stage loop (one workgroup per CU, two K-tiles in flight)
      dim3 grid(mansy_ceil_div(p.N, 128), mansy_ceil_div(p.M, 256), 1);
      MANSY_GEMM_LAUNCH(gemm_bf16g_kernel, grid, dim3(512), st, p); MANSY_LAUNCH_CHECK(); return MANSY_OK;
    }
    if (tile == 128) { dim3 grid(mansy_ceil_div(p.N, 128), mansy_ceil_div(p.M, 128), 1); MANSY_GEMM_LAUNCH((gemm_bf16f_kernel<128, 128>), grid, block, st, p); MANSY_LAUNCH_CHECK(); return MANSY_OK; }
    if (tile == 96) { dim3 grid(mansy_ceil_div(p.N, 64), mansy_ceil_div(p.M, 128), 1); MANSY_GEMM_LAUNCH((gemm_bf16f_kernel<128, 64>), grid, block, st, p); MANSY_LAUNCH_CHECK(); return MANSY_OK; }
  }
  // 64 x 64 tiles (the [4 096-row] decoder products): the ring loop, both modes (variant 0 / 7: the round-2 loop, for A/B runs)
  if (tile == 64 && bvar != 0 && bvar != 7 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && p.lda % 4 == 0) {
    dim3 grid(mansy_ceil_div(p.N, 64), mansy_ceil_div(p.M, 64), 1), block(NT);
    if (prec == 3) {
      MANSY_GEMM_LAUNCH((gemm_bf16h_kernel<64, 64, 3, 2>), grid, block, st, p);
    } else MANSY_GEMM_LAUNCH((gemm_bf16h_kernel<64, 64, 3, 3>), grid, block, st, p);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  if (tile == 96) tile = 64;
  const int BMN = tile == 128 ? 128 : 64;
  dim3 grid(mansy_ceil_div(p.N, BMN), mansy_ceil_div(p.M, BMN), 1), block(NT);
  if (prec == 3) {
    if (tile == 128) MANSY_GEMM_LAUNCH((gemm_bf16p_kernel<128, 128, 2>), grid, block, st, p);
    else MANSY_GEMM_LAUNCH((gemm_bf16p_kernel<64, 64, 2>), grid, block, st, p);
  } else {
    if (tile == 128) MANSY_GEMM_LAUNCH((gemm_bf16p_kernel<128, 128, 3>), grid, block, st, p);
    else MANSY_GEMM_LAUNCH((gemm_bf16p_kernel<64, 64, 3>), grid, block, st, p);
  }
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_weight_planes(const MansyWPlaneTab& tab, unsigned short* out, unsigned short* out_t, long long plane_stride, int n_planes,
                               hipStream_t st) {
  MANSY_REQUIRE(tab.n >= 0 && tab.n <= MANSY_WPLANE_MAX && out && out_t && (n_planes == 2 || n_planes == 3), "weight_planes: bad arguments");
  long long tiles = 0;
  for (int t = 0; t < tab.n; ++t) tiles += (long long)((tab.N[t] + 31) / 32) * ((tab.K[t] + 31) / 32);
  if (tiles == 0) return MANSY_OK;
  MANSY_LAUNCH(weight_planes_kernel, dim3((unsigned)tiles), dim3(256), 0, st, tab, out, out_t, plane_stride, n_planes);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// resident workgroups per CU (diagnostic)
extern "C" int mansy_gemm_bf16s_occupancy(int prec, int tile) {
  int n = -1;
  hipError_t e;
#define OCC(BMN, NPL, D) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_bf16s_kernel<BMN, BMN, false, false, NPL, D>, NT, 0)
  if (prec == 3 && tile == 128) OCC(128, 2, true);
  else if (prec == 3) OCC(64, 2, true);
  else if (tile == 128) OCC(128, 3, false);
  else OCC(64, 3, false);
#undef OCC
  return e == hipSuccess ? n : -1;
}
