// The wave-split-K loop body of the small fp32 products (gemm_f32_wsk_kernel / gemm_f32_wsk_dual_kernel of gemm_f32.hip) as a header, so that the
// persistent rollout kernel of ppo_engine.hip (round 5) can run the SAME body -- same tiles, same per-wave K-tile assignment, same order of sums: the
// same bits -- on the blocks it places itself.
#pragma once
#include "gemm_tile.h"

namespace mansy_gemm {

__device__ __forceinline__ void glds16_sc1(unsigned voff, const void* sbase, unsigned lds_dst) {      // as glds16, the load misses this CU's L1
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

template <int R, bool KMAJ>
__device__ __forceinline__ void read_frag_dma(const float* __restrict__ lds, int rb, int r, int h, int chunk, float (&out)[8]) {
  if (!KMAJ) {
    const int row = rb + r, c0 = h * 4 + chunk * 2;
    const int sw = (row >> 1) & 7;
    const float4 v0 = *reinterpret_cast<const float4*>(lds + row * BK + ((c0 + 0) ^ sw) * 4);
    const float4 v1 = *reinterpret_cast<const float4*>(lds + row * BK + ((c0 + 1) ^ sw) * 4);
    out[0] = v0.x; out[1] = v0.y; out[2] = v0.z; out[3] = v0.w;
    out[4] = v1.x; out[5] = v1.y; out[6] = v1.z; out[7] = v1.w;
  } else {
    const float* q = lds + (h * 16 + chunk * 8) * R + rb + r;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) out[kk] = q[kk * R];
  }
}

// Wave-split-K loop for SMALL products (round 4): a launch that cannot fill the chip -- the PPO cycle's FeatureNet / head / dF products, 80-160
// tiles of 64 x 64 -- runs as long as ONE workgroup's K chain: measured 3.8 us + 0.62 us per K-tile (tools/gemm_small_ksweep.py; a lone wave per
// SIMD hides neither its fragment-read latency nor the 64-cycle dependent MFMA chain).  Here a workgroup owns a 32 x 32 output block and its four
// waves SPLIT THE K-TILES between them (wave w takes tiles w, w + 4, ...): the chain is a quarter as long and four times as many workgroups fill the
// chip.  Each wave stages its own tiles into its own LDS stage (A and B by LDS-DMA, the images and fragment reads of the loop above; no workgroup
// barrier inside the loop: a wave only reads what it staged itself, ordered by its own vmcnt; the next tile is requested as soon as the current one
// sits in registers and flies under its MFMAs; 32 KB of LDS per workgroup, so five fit a CU), the four partial blocks are summed
// through LDS in wave order (deterministic), then the shared row-major epilogue runs.  A K-contiguous; B K-contiguous or K-major.
// Round 5 (in-kernel phase stamps of the lab build, since deleted; profiles/r05_wsk_phase_lab_two_stage.txt): inside a [32, 512, 512] launch a wave spends 0.44 us before its first DMA is out,
// 0.36 us until the tile has landed, 4 x 0.68 us on its K-tiles (16 dependent MFMAs ~0.47 us -- the block's 256 MFMAs are 1.7 us of the CU's four matrix pipes
// however the waves are arranged --, fragment reads ~0.12, DMA issue ~0.13) and 0.44 us on the cross-wave sum and the epilogue.  A second stage per wave with the
// fragment reads and the DMA issue interleaved BEHIND the MFMAs (built, bit-identical) made the K-tile no shorter (0.72 us: the dependent MFMA chain is the period)
// and the preamble 0.3 us longer; requesting the row epilogue's residual / mask / bias float4 under the K loop and handing the summed block over in registers
// changed nothing inside the launch and cost 0.8 us per launch in the B = 512 step (rocprofv3 averages): all removed again.
constexpr int WSK_T = 32, WSK_WAVE_FLOATS = 2 * WSK_T * BK, WSK_SMEM_FLOATS = 4 * WSK_WAVE_FLOATS;
// The loop as a device function of (problem, workgroup index `orig` inside the problem's gx x gy x gz grid, the workgroup's LDS): one problem per
// launch (gemm_f32_wsk_kernel) or two independent problems side by side in one launch (gemm_f32_wsk_dual_kernel).
// A_SC1: the A operand was written by OTHER workgroups of the SAME launch (the persistent rollout of ppo_engine.hip): its LDS-DMA loads carry sc1
// (they miss this CU's L1, which no other CU's store ever refreshes).  DIRECT: `orig` IS the block index t of the gx x gy x gz grid (no XCD remap: the
// caller places the work itself).
template <bool AK, bool BKM, bool A_SC1 = false, bool DIRECT = false>
__device__ __forceinline__ void gemm_f32_wsk_body(const GemmParams& p, int orig, int gx, int gy, int gz, float* smem) {
  constexpr int T = WSK_T, TILE_FLOATS = T * BK, WAVE_FLOATS = WSK_WAVE_FLOATS, CLD = T + 4;       // ONE stage per wave: 8 KB (32 KB per workgroup, five per CU)
  static_assert(T * CLD + 64 <= WAVE_FLOATS, "a partial block and its row sums must fit a wave's stage");

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y, split;
  {   // XCD-aware bijective remap over the whole 3-D grid, K split slowest (see gemm_f32_dma_kernel)
    const int per_split = gx * gy, nwg = per_split * gz;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    if (DIRECT) t = orig;
    split = t / per_split; t -= split * per_split;
    tile_y = t / gx; tile_x = t - tile_y * gx;
    if (AK && p.ep.tile_list) {          // the list names 64 x 64 tiles (column tile, row tile): four 32 x 32 blocks each
      const int e = t >> 2, sub = t & 3;
      tile_x = 2 * p.ep.tile_list[2 * e] + (sub & 1); tile_y = 2 * p.ep.tile_list[2 * e + 1] + (sub >> 1);
    }
  }
  const float* Ap = p.A; const float* Bp = p.B; float* Cp = p.C; float* rowsum_dst = p.ep.a_rowsum;
  if (AK && p.A2) {                                 // two same-shape problems in one launch: the upper half of the splits is problem 2
    const int prob = split / p.splits_pp;
    split -= prob * p.splits_pp;
    if (prob) { Ap = p.A2; Bp = p.B2; Cp = p.C2; rowsum_dst = p.a_rowsum2; }
  }
  const int m0 = tile_y * T, n0 = tile_x * T;
  int rs_first = 0;                                 // first column block of this row panel that runs
  if (AK && p.ep.tile_nrange) {                     // the table is per 64-row tile of C
    const int lo = p.ep.tile_nrange[2 * (m0 / 64)], hi = p.ep.tile_nrange[2 * (m0 / 64) + 1];
    if (n0 >= hi || n0 + T <= lo) return;
    rs_first = lo / T;
  }
  int k_begin = split * p.k_per_split;
  int k_end = min(p.K, k_begin + p.k_per_split);
  if (p.ep.tile_krange) {                          // block-diagonal weights: the table is per 64-column tile
    k_begin = max(k_begin, p.ep.tile_krange[2 * (n0 / 64)]);
    k_end = min(k_end, p.ep.tile_krange[2 * (n0 / 64) + 1]);
  }
  const int nk = max(0, (k_end - k_begin) / BK);

  // this WAVE's pieces of a tile.  K-contiguous operand: image [32][32] (128-B rows, k-chunk c of a row in slot c ^ ((row >> 1) & 7));
  // K-major operand: image [32 k][32 rows] linear
  unsigned voa[4], vob[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    if (!AK) voa[i] = (unsigned)((min(m0 + row, p.M - 1) - m0) * p.lda + c * 4) * 4u;
    else voa[i] = (unsigned)(row * p.lda + (min(m0 + (lane & 7) * 4, p.M - 4) - m0)) * 4u;       // row = k index of the tile here
    if (!BKM) vob[i] = (unsigned)((min(n0 + row, p.N - 1) - n0) * p.ldb + c * 4) * 4u;
    else vob[i] = (unsigned)(row * p.ldb + (min(n0 + (lane & 7) * 4, p.N - 4) - n0)) * 4u;
  }
  const float* const sa = AK ? Ap + (long long)k_begin * p.lda + m0 : Ap + (long long)m0 * p.lda + k_begin;
  const float* const sb = BKM ? Bp + (long long)k_begin * p.ldb + n0 : Bp + (long long)n0 * p.ldb + k_begin;
  const long long step_a = AK ? (long long)BK * p.lda : BK, step_b = BKM ? (long long)BK * p.ldb : BK;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * (WAVE_FLOATS * 4u));
  auto dma = [&](int kt) {
    const float* ca = sa + (long long)kt * step_a;
    const float* cb = sb + (long long)kt * step_b;
    const unsigned dst = lds_w;
#pragma unroll
    for (int i = 0; i < 4; ++i) { if (A_SC1) glds16_sc1(voa[i], ca, dst + i * 1024u); else glds16(voa[i], ca, dst + i * 1024u); }
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(vob[i], cb, dst + TILE_FLOATS * 4u + i * 1024u);
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  // accumulating product whose blocks have one owner (no K split over workgroups): the old values of C are requested NOW, so that their round
  // trip (the gradient buffer was zero-filled by another launch: cold) runs under the K loop instead of behind it
  const bool rmw = AK && p.ep.accumulate && p.splits_pp == 1 && p.c_rmw_ok;
  float4 c_old = make_float4(0.f, 0.f, 0.f, 0.f);
  float4* c_dst = nullptr;
  if (rmw) {
    const int row = m0 + (tid >> 3), col = n0 + (tid & 7) * 4;
    if (row < p.M && col < p.N) {                        // N % 4 == 0 on this path: a float4 is entirely in or out
      c_dst = reinterpret_cast<float4*>(Cp + (long long)row * p.ldc + col);
      c_old = *c_dst;
    }
  }
  float* const mine = smem + wave * WAVE_FLOATS;
  const float* const a_l = mine;
  const float* const b_l = mine + TILE_FLOATS;
  // bias-gradient rider: ONE column block per row panel (the first that runs) sums the k-rows of the staged (K-major) A tiles -- every column block
  // stages the same tiles.  (Round 4, first form: the panel's column blocks shared the k-rows and each added its 32 sums atomically -- up to 32
  // workgroups x 4 waves on the same 32 addresses at the end of a 10 us launch.)
  const bool do_rowsum = AK && rowsum_dst != nullptr && tile_x == rs_first;
  float rowsum = 0.f;
  int kt = wave;
  if (kt < nk) dma(kt);
  for (; kt < nk; kt += 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's tile has landed (it staged it itself: no barrier)
    float af[2][8], bf[2][8];
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      read_frag_dma<T, AK>(a_l, 0, r, h, chunk, af[chunk]);
      read_frag_dma<T, BKM>(b_l, 0, r, h, chunk, bf[chunk]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the whole tile is in registers: the stage may be staged again ...
    if (kt + 4 < nk) dma(kt + 4);                        // ... and the wave's next tile flies under this tile's 16 MFMAs
    if (do_rowsum) {                                     // lane (r, h): row r of the block, k-rows 16 h .. 16 h + 15 of the tile (= this lane's A fragments)
#pragma unroll
      for (int chunk = 0; chunk < 2; ++chunk)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) rowsum += af[chunk][kk];
    }
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[chunk][kk], bf[chunk][kk], acc, 0, 0, 0);
  }
  // partial block of this wave -> its own (now idle) stage, [32][36]; its row sums behind it
#pragma unroll
  for (int e = 0; e < 16; ++e) mine[((e & 3) + 8 * (e >> 2) + 4 * h) * CLD + r] = acc[e];
  if (do_rowsum) mine[T * CLD + lane] = rowsum;
  __syncthreads();
  if (do_rowsum && tid < T && m0 + tid < p.M) {            // the four waves' sums in wave order, the two k-halves of each; then one add per row
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) t += smem[w * WAVE_FLOATS + T * CLD + tid] + smem[w * WAVE_FLOATS + T * CLD + 32 + tid];
    if (p.splits_pp == 1) rowsum_dst[m0 + tid] += t;      // the only workgroup of the launch that owns these rows' sums
    else atomicAdd(rowsum_dst + m0 + tid, t);             // K split over workgroups: one add per split
  }
  const int off = (tid >> 3) * CLD + (tid & 7) * 4;      // thread -> (row tid / 8, float4 tid % 8): the mapping of gemm_epilogue_rows<32, 32, 256>
  float4 v = *reinterpret_cast<const float4*>(smem + off);
#pragma unroll
  for (int w = 1; w < 4; ++w) {                          // the four partial blocks, in wave order
    const float4 q = *reinterpret_cast<const float4*>(smem + w * WAVE_FLOATS + off);
    v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
  }
  float* const Cz = Cp + (long long)split * p.ep.split_slab;
  if (rmw) {                                             // this workgroup is the only owner of its block: plain read-add-write (the read was issued up front)
    if (c_dst) *c_dst = make_float4(c_old.x + v.x, c_old.y + v.y, c_old.z + v.z, c_old.w + v.w);
    return;
  }
  if (p.ep.accumulate || (p.splits_pp > 1 && p.ep.split_slab == 0)) {
    // accumulating product with a K split over workgroups / K split without slabs: plain epilogue by construction, partial sums added atomically
    const int row = m0 + (tid >> 3), col = n0 + (tid & 7) * 4;
    if (row < p.M) {
      float* dst = Cz + (long long)row * p.ldc + col;
      if (col + 0 < p.N) atomicAdd(dst + 0, v.x);
      if (col + 1 < p.N) atomicAdd(dst + 1, v.y);
      if (col + 2 < p.N) atomicAdd(dst + 2, v.z);
      if (col + 3 < p.N) atomicAdd(dst + 3, v.w);
    }
    return;
  }
  *reinterpret_cast<float4*>(smem + off) = v;             // read back by the same thread below
  __syncthreads();
  gemm_epilogue_rows<T, T, NT>(p, smem, m0, n0, tid, Cz);
}


}  // namespace mansy_gemm
