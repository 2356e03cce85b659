// Exact-fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), LDS-tiled, with fused
// epilogues.  This is the MFMA-bound kernel of the viewport-prediction Transformer
// (reference arithmetic: torch.nn.Transformer Linear layers) and of the PPO nets.
//
// Two main loops share one epilogue (gemm_epilogue):
//   gemm_f32_dma_kernel  the hot path: operand tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), unpadded
//                        XOR-swizzled LDS images, tiles 128x64 / 64x64 (128x128 when forced), whole K-tiles only.
//   gemm_f32_kernel      register-staged fallback for everything else (K % 32 != 0, unaligned / odd leading
//                        dimensions): padded LDS rows, K tail zero-filled on the way into LDS.
// Common structure: 256 threads = 4 waves (2x2), BK = 32, each wave owns a (BM/2) x (BN/2) sub-tile = TM x TN MFMA blocks
// of 32x32, one barrier per K-tile, two LDS buffers.  LDS images keep the global orientation of each operand:
//   K-contiguous operand  -> rows of BK floats (fragments: 2 x ds_read_b128 per 8 k)
//   K-major operand       -> BK rows of `rows` floats (fragments: ds_read_b32, conflict-free)
// The 32x32x2 MFMA consumes k = {0,1} from lane halves h = lane>>5; because the sum over k is order-free as long as A and
// B agree, lane half h is given k in [16h, 16h+16) of the BK block, so a K-contiguous fragment is 16 consecutive floats
// of one LDS row.  Split-K: atomics (dW products) or per-split slabs (skinny products); workgroups are remapped so that
// the tiles sharing an operand panel -- and whole K splits -- land on one XCD's L2.
#include <vector>
#include "gemm_tile.h"
#include "gemm_wsk.h"

using namespace mansy_gemm;

namespace {

template <int R, bool KMAJ>
struct TileGeom {
  static constexpr int LOADS = R * BK / 4 / NT;                    // float4 per thread
  static constexpr int LDS_FLOATS = KMAJ ? BK * (R + 4) : R * KC_LD;
};

// Load one operand tile (R rows x BK k) into registers.  Branch-free: every address is clamped into the operand,
// so the loads of a K-tile issue back to back and ONE vmcnt wait sits in front of the LDS stores (guards written
// as branches made hipcc serialise every load).  Rows/columns beyond the matrix read a duplicate of the last valid
// one -- harmless, those accumulators are never stored.  Only k >= k_end must contribute exact zeros; that select is
// applied in store_tile (after the MFMAs of the current tile) so it does not pull the vmcnt wait forward.
// VEC: 16-byte loads (ld % 4 == 0, 16-byte aligned base, contiguous-axis extent % 4 == 0).
template <int R, bool KMAJ, bool VEC>
__device__ __forceinline__ void load_tile(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int k_end,
                                          float4 (&reg)[TileGeom<R, KMAJ>::LOADS], int tid) {
#pragma unroll
  for (int i = 0; i < TileGeom<R, KMAJ>::LOADS; ++i) {
    const int idx = tid + i * NT;
    int outer, outer_lim, inner0, lim_inner;
    if (!KMAJ) {
      outer = row0 + (idx >> 3); outer_lim = nrows; inner0 = k0 + (idx & 7) * 4; lim_inner = k_end;
    } else {
      constexpr int C4 = R / 4;
      outer = k0 + idx / C4; outer_lim = k_end; inner0 = row0 + (idx % C4) * 4; lim_inner = nrows;
    }
    const long long base = (long long)min(outer, outer_lim - 1) * ld;
    float4 v;
    if (VEC) {
      v = *reinterpret_cast<const float4*>(P + base + min(inner0, lim_inner - 4));
    } else {
      v.x = P[base + min(inner0 + 0, lim_inner - 1)];
      v.y = P[base + min(inner0 + 1, lim_inner - 1)];
      v.z = P[base + min(inner0 + 2, lim_inner - 1)];
      v.w = P[base + min(inner0 + 3, lim_inner - 1)];
    }
    reg[i] = v;
  }
}

// Write the staged registers to the LDS image, zeroing k >= k_end (K tail of the last tile / split).
template <int R, bool KMAJ>
__device__ __forceinline__ void store_tile(float* __restrict__ lds, const float4 (&reg)[TileGeom<R, KMAJ>::LOADS], int tid, int k0,
                                           int k_end) {
#pragma unroll
  for (int i = 0; i < TileGeom<R, KMAJ>::LOADS; ++i) {
    const int idx = tid + i * NT;
    float4 v = reg[i];
    if (!KMAJ) {
      const int r = idx >> 3, c4 = idx & 7;
      const int k = k0 + c4 * 4;
      if (k + 0 >= k_end) v.x = 0.f;
      if (k + 1 >= k_end) v.y = 0.f;
      if (k + 2 >= k_end) v.z = 0.f;
      if (k + 3 >= k_end) v.w = 0.f;
      *reinterpret_cast<float4*>(lds + r * KC_LD + c4 * 4) = v;
    } else {
      constexpr int C4 = R / 4;
      const int kr = idx / C4, c4 = idx % C4;
      if (k0 + kr >= k_end) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(lds + kr * (R + 4) + c4 * 4) = v;
    }
  }
}

// Read an 8-k fragment chunk for one 32-row MFMA block: out[kk] = tile(row = rb + r, k = 16h + 8*chunk + kk)
template <int R, bool KMAJ>
__device__ __forceinline__ void read_frag(const float* __restrict__ lds, int rb, int r, int h, int chunk, float (&out)[8]) {
  if (!KMAJ) {
    const float4* p = reinterpret_cast<const float4*>(lds + (rb + r) * KC_LD + h * 16 + chunk * 8);
    const float4 v0 = p[0], v1 = p[1];
    out[0] = v0.x; out[1] = v0.y; out[2] = v0.z; out[3] = v0.w;
    out[4] = v1.x; out[5] = v1.y; out[6] = v1.z; out[7] = v1.w;
  } else {
    const float* p = lds + (h * 16 + chunk * 8) * (R + 4) + rb + r;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) out[kk] = p[kk * (R + 4)];
  }
}


template <int BM, int BN, bool AK, bool BKM, bool VEC>
__global__ __launch_bounds__(NT) void gemm_f32_kernel(GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_FLOATS = TileGeom<BM, AK>::LDS_FLOATS;
  constexpr int B_FLOATS = TileGeom<BN, BKM>::LDS_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[2 * (A_FLOATS + B_FLOATS)];
  constexpr int STAGE = A_FLOATS + B_FLOATS;   // stage s: A at smem + s*STAGE, B right after it

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile mapping (speed only, never correctness): workgroups are dealt round-robin over the 8 XCDs by linear id,
  // so give XCD x the CONTIGUOUS logical tile range [x*nwg/8, (x+1)*nwg/8): the n-tiles that share an A row panel then hit
  // one XCD's L2 instead of eight (PMC: 4.9x operand over-fetch before this remap).  Bijective for any nwg.
  int tile_x = blockIdx.x, tile_y = blockIdx.y;
  {
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
    if (BM == 64 && BN == 64 && p.ep.tile_list) { tile_x = p.ep.tile_list[2 * t]; tile_y = p.ep.tile_list[2 * t + 1]; }
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  int first_tile = 0;
  if (BM == 64 && p.ep.tile_nrange) {             // only the column tiles that meet this row tile's wanted range run
    const int lo = p.ep.tile_nrange[2 * tile_y], hi = p.ep.tile_nrange[2 * tile_y + 1];
    if (n0 >= hi || n0 + BN <= lo) return;
    first_tile = lo / BN;
  }
  const bool first_n_tile = tile_x == first_tile;
  const int k_begin = blockIdx.z * p.k_per_split;
  const int k_end = min(p.K, k_begin + p.k_per_split);
  const int nk = (k_end - k_begin + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float rowsum = 0.f;
  float4 ra[TileGeom<BM, AK>::LOADS], rb[TileGeom<BN, BKM>::LOADS];
  if (nk > 0) {
    load_tile<BM, AK, VEC>(p.A, p.lda, m0, p.M, k_begin, k_end, ra, tid);
    load_tile<BN, BKM, VEC>(p.B, p.ldb, n0, p.N, k_begin, k_end, rb, tid);
    store_tile<BM, AK>(smem, ra, tid, k_begin, k_end);
    store_tile<BN, BKM>(smem + A_FLOATS, rb, tid, k_begin, k_end);
  }
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      const int k0 = k_begin + (kt + 1) * BK;
      load_tile<BM, AK, VEC>(p.A, p.lda, m0, p.M, k0, k_end, ra, tid);
      load_tile<BN, BKM, VEC>(p.B, p.ldb, n0, p.N, k0, k_end, rb, tid);
    }
    const float* a_l = smem + cur * STAGE;
    const float* b_l = a_l + A_FLOATS;
    if (AK && p.ep.a_rowsum && first_n_tile && tid < BM) {       // bias gradient: row sums of the staged A tile (zeros beyond k_end)
#pragma unroll
      for (int kk = 0; kk < BK; ++kk) rowsum += a_l[kk * (BM + 4) + tid];
    }
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      float af[TM][8], bf[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) read_frag<BM, AK>(a_l, wm * (BM / 2) + i * 32, r, h, chunk, af[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_frag<BN, BKM>(b_l, wn * (BN / 2) + j * 32, r, h, chunk, bf[j]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      const int k0n = k_begin + (kt + 1) * BK;
      store_tile<BM, AK>(smem + (cur ^ 1) * STAGE, ra, tid, k0n, k_end);
      store_tile<BN, BKM>(smem + (cur ^ 1) * STAGE + A_FLOATS, rb, tid, k0n, k_end);
    }
    __syncthreads();
  }

  if (AK && p.ep.a_rowsum && first_n_tile && tid < BM && m0 + tid < p.M) atomicAdd(p.ep.a_rowsum + m0 + tid, rowsum);

  gemm_epilogue<BM, BN, 2 * (A_FLOATS + B_FLOATS)>(p, acc, smem, m0, n0, tid, blockIdx.z, p.C);
}

// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA main loop (global_load_lds_dwordx4): the operand tiles go HBM/L2 -> LDS without passing through VGPRs, so there
// is no staging register set, no ds_write pass and no select in the K loop (+6..13 % over the register-staged loop on the
// VP shapes, tools/gemm_lab.hip).  Preconditions (checked by the dispatcher): 16-byte aligned operands with ld % 4 == 0,
// K % 32 == 0 (no K tail to zero-fill: the DMA cannot), and for K-major operands the row count % 4 == 0.
// One DMA wave-instruction writes 64 lanes x 16 B = 1 KiB of LDS *contiguously* (wave-uniform base in M0 + lane*16); only
// the global source address is per lane.  LDS images are therefore unpadded:
//   K-contiguous operand -> [R][32]: 128-B rows, 8 rows per instruction.  Bank conflicts of the ds_read_b128 fragment reads
//                           are avoided by an XOR swizzle applied on the SOURCE side: slot c' of row holds k-chunk
//                           c' ^ ((row >> 1) & 7).  (A ds_read_b128 is served in four 16-lane groups -- {0-3,12-15,20-27},
//                           {4-11,16-19,28-31}, ... -- and a 128-B row spans half of the 64 banks, so the 16-B slot of a lane is
//                           (row & 1) * 8 + slot; with (row & 7) as the key rows 12 and 20, 13 and 21, ... of a group met on
//                           one slot: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.48.  (row >> 1) & 7 is distinct over
//                           the even and over the odd rows of every group.)
//   K-major operand      -> [32][R]: fragment reads are ds_read_b32 along the row axis, conflict-free as is.
// The DMA is issued through inline asm: hipcc would otherwise put s_waitcnt vmcnt(0) in front of every ds_read that
// follows an LDS-DMA; completion is counted by hand (one vmcnt(0) + barrier per K-tile, the loads of tile t+1 having the
// whole MFMA phase of tile t to land).
// One DMA piece: per-lane source = sbase (SGPR pair, wave-uniform) + voff (VGPR, bytes); destination = LDS byte address
// lds_dst (wave-uniform, via M0) + lane*16.  Keeping the tile corner in SGPRs and the per-lane offsets loop-invariant makes a
// piece 3 scalar instructions + the load (per-lane 64-bit address arithmetic in the K loop cost ~8 % of the MFMA rate).
// Byte offsets of this lane's R/32 DMA pieces relative to the tile corner (row0, k0); piece i lands at LDS byte
// (tile image) + i*4096 + wave*1024 + lane*16 for both layouts.
template <int R, bool KMAJ>
__device__ __forceinline__ void dma_offsets(int ld, int row0, int nrows, unsigned (&voff)[R / 32], int tid) {
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < R / 32; ++i) {
    if (!KMAJ) {            // image [R][32]: 8 rows x 128 B per piece, k-chunk c of a row in slot c ^ ((row >> 1) & 7)
      const int row = i * 32 + wave * 8 + (lane >> 3);
      const int c = (lane & 7) ^ ((row >> 1) & 7);
      voff[i] = (unsigned)((min(row0 + row, nrows - 1) - row0) * ld + c * 4) * 4u;
    } else {                // image [32][R], linear
      constexpr int C4 = R / 4;
      const int idx = (i * 4 + wave) * 64 + lane;
      const int kr = idx / C4, c4 = idx % C4;
      voff[i] = (unsigned)(kr * ld + (min(row0 + c4 * 4, nrows - 4) - row0)) * 4u;
    }
  }
}

// PLAIN (compile time): the launch has none of the optional forms -- one problem, no K split over workgroups, no tile list / ranges, no row-sum rider, no
// column-group order, the row-major epilogue.  The decoder recurrence is a chain of ~240 such launches whose fixed cost (prologue, first-tile latency,
// epilogue) is a quarter of their time: their instance carries none of the other forms' scalar loads and branches in front of the first LDS-DMA.
template <int BM, int BN, bool AK, bool BKM, bool PLAIN = false>
__global__ __launch_bounds__(NT) void gemm_f32_dma_kernel(GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64, PA = BM / 32, PB = BN / 32;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, STAGE = A_FLOATS + B_FLOATS;
  constexpr int C_FLOATS = BM * (BN + 4);
  constexpr int SMEM_FLOATS = 2 * STAGE > C_FLOATS ? 2 * STAGE : C_FLOATS;
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y;
  int split;
  {   // XCD-aware bijective remap (see gemm_f32_kernel), here over the whole 3-D grid with the K split slowest: XCD x gets the
      // contiguous logical range [x*nwg/8, (x+1)*nwg/8), i.e. whole K splits when there are >= 8 of them -- every tile of a
      // split then streams the same operand rows through ONE L2 (dW products: operands fetched once instead of 2-4x).
    const int per_split = gridDim.x * gridDim.y, nwg = per_split * gridDim.z;
    const int orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    if (PLAIN) split = 0;
    else { split = t / per_split; t -= split * per_split; }
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
    if (!PLAIN && p.col_group > 0 && (int)gridDim.x > p.col_group) {      // column-group order (bijective for any grid): see GemmParams::col_group
      const int G = p.col_group, gy = gridDim.y, full = ((int)gridDim.x / G) * G;
      if (t < full * gy) { const int g = t / (G * gy), rr = t - g * G * gy; tile_y = rr / G; tile_x = g * G + (rr - tile_y * G); }
      else { const int W = gridDim.x - full, rr = t - full * gy; tile_y = rr / W; tile_x = full + (rr - tile_y * W); }
    }
    if (!PLAIN && BM == 64 && BN == 64 && p.ep.tile_list) { tile_x = p.ep.tile_list[2 * t]; tile_y = p.ep.tile_list[2 * t + 1]; }
  }
  const float* Ap = p.A; const float* Bp = p.B; float* Cp = p.C; float* rowsum_dst = PLAIN ? nullptr : p.ep.a_rowsum;
  if (!PLAIN && p.A2) {                                       // two same-shape problems in one launch: the upper half of the splits is problem 2
    const int prob = split / p.splits_pp;
    split -= prob * p.splits_pp;
    if (prob) { Ap = p.A2; Bp = p.B2; Cp = p.C2; rowsum_dst = p.a_rowsum2; }
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  // the column tiles of this row panel that run (they share the row-sum work): counted from N, not from gridDim.x -- with a
  // tile list the grid's x extent is the length of the list, not the number of column tiles of C
  const int n_col_tiles = (p.N + BN - 1) / BN;
  int rs_first = 0, rs_cnt = n_col_tiles;
  if (!PLAIN && BM == 64 && p.ep.tile_nrange) {
    const int lo = p.ep.tile_nrange[2 * tile_y], hi = p.ep.tile_nrange[2 * tile_y + 1];
    if (n0 >= hi || n0 + BN <= lo) return;
    rs_first = lo / BN;
    rs_cnt = min(n_col_tiles, (hi + BN - 1) / BN) - rs_first;
  }
  int k_begin = PLAIN ? 0 : split * p.k_per_split;
  int k_end = PLAIN ? p.K : min(p.K, k_begin + p.k_per_split);
  if (!PLAIN && BN == 64 && p.ep.tile_krange) {             // structurally-zero K-tiles of this column tile are skipped
    k_begin = max(k_begin, p.ep.tile_krange[2 * tile_x]);
    k_end = min(k_end, p.ep.tile_krange[2 * tile_x + 1]);
  }
  const int nk = max(0, (k_end - k_begin) / BK);  // K % BK == 0 and k_per_split % BK == 0 on this path

  unsigned voa[PA], vob[PB];
  dma_offsets<BM, AK>(p.lda, m0, p.M, voa, tid);
  dma_offsets<BN, BKM>(p.ldb, n0, p.N, vob, tid);
  // tile corners (wave-uniform -> SGPRs) and their per-K-tile strides
  const float* sa = AK ? Ap + (long long)k_begin * p.lda + m0 : Ap + (long long)m0 * p.lda + k_begin;
  const float* sb = BKM ? Bp + (long long)k_begin * p.ldb + n0 : Bp + (long long)n0 * p.ldb + k_begin;
  const long long step_a = AK ? (long long)BK * p.lda : BK, step_b = BKM ? (long long)BK * p.ldb : BK;
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * 1024u);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float rowsum = 0.f;
  if (nk > 0) {
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(voa[i], sa, lds_wave + i * 4096u);
#pragma unroll
    for (int i = 0; i < PB; ++i) glds16(vob[i], sb, lds_wave + A_FLOATS * 4u + i * 4096u);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's pieces of tile kt have landed ...
    __builtin_amdgcn_s_barrier();                        // ... and everyone's; everyone is done reading tile kt-1
    asm volatile("" ::: "memory");
    sa += step_a; sb += step_b;                          // corners of tile kt+1
    const unsigned lds_next = lds_wave + (unsigned)(cur ^ 1) * (STAGE * 4u);
    if (kt + 1 < nk) {
#pragma unroll
      for (int i = 0; i < PA; ++i) glds16(voa[i], sa, lds_next + i * 4096u);
#pragma unroll
      for (int i = 0; i < PB; ++i) glds16(vob[i], sb, lds_next + A_FLOATS * 4u + i * 4096u);
    }
    const float* a_l = smem + cur * STAGE;
    const float* b_l = a_l + A_FLOATS;
    if (AK && rowsum_dst && tid < BM) {          // bias gradient: row sums of the staged A tile.  Every n-tile of this
      // row panel stages the same A tile, so they share the BK rows (a single n-tile doing all of them ran ~1.3x longer
      // than its neighbours and set the kernel's tail)
      for (int kk = tile_x - rs_first; kk < BK; kk += rs_cnt) rowsum += a_l[kk * BM + tid];
    }
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      float af[TM][8], bf[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) read_frag_dma<BM, AK>(a_l, wm * (BM / 2) + i * 32, r, h, chunk, af[i]);
#pragma unroll
      for (int j = 0; j < TN; ++j) read_frag_dma<BN, BKM>(b_l, wn * (BN / 2) + j * 32, r, h, chunk, bf[j]);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads retired before the barrier that frees this buffer
  }
  __syncthreads();                                        // staging LDS idle: the epilogue reuses it

  if (AK && rowsum_dst && tile_x - rs_first < BK && tid < BM && m0 + tid < p.M) atomicAdd(rowsum_dst + m0 + tid, rowsum);
  if (PLAIN) {      // the row-major pass by construction (the dispatcher checked): gemm_epilogue's first branch, without its tests
    constexpr int CLD = BN + 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          smem[(wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wn * (BN / 2) + j * 32 + r] = acc[i][j][e];
    __syncthreads();
    gemm_epilogue_rows<BM, BN, NT>(p, smem, m0, n0, tid, Cp);
    return;
  }
  gemm_epilogue<BM, BN, SMEM_FLOATS>(p, acc, smem, m0, n0, tid, split, Cp);
}

// ---------------------------------------------------------------------------------------------------------------------
template <bool AK, bool BKM>
__global__ __launch_bounds__(NT) void gemm_f32_wsk_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(1024))) float smem[WSK_SMEM_FLOATS];
  gemm_f32_wsk_body<AK, BKM>(p, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, gridDim.x, gridDim.y, gridDim.z, smem);
}
// Two INDEPENDENT small products in one launch (round 4): the workgroups [0, n1) run problem 1 (weight-gradient form, both operands K-major), the rest
// problem 2 (A K-contiguous, B K-major).  The PPO minibatch step's fc weight-gradient pair and its dF product both read dA1 and feed different
// consumers; as two dependent launches each cost its own ~8 us of launch / fill / drain in a cycle that is a chain of such launches.
struct WskGrid { int x, y, z; };
__global__ __launch_bounds__(NT) void gemm_f32_wsk_dual_kernel(GemmParams p1, WskGrid g1, GemmParams p2, WskGrid g2) {
  __shared__ __attribute__((aligned(1024))) float smem[WSK_SMEM_FLOATS];
  const int n1 = g1.x * g1.y * g1.z, orig = blockIdx.x;
  if (orig < n1) gemm_f32_wsk_body<true, true>(p1, orig, g1.x, g1.y, g1.z, smem);
  else gemm_f32_wsk_body<false, true>(p2, orig - n1, g2.x, g2.y, g2.z, smem);
}

template <int BM, int BN>
int launch_dma(const GemmParams& p, int a_kmajor, int b_kmajor, int splits, hipStream_t st) {
  dim3 grid(mansy_ceil_div(p.N, BN), mansy_ceil_div(p.M, BM), splits);
  if (BM == 64 && BN == 64 && p.ep.tile_list) grid = dim3(p.ep.tile_list_n, 1, splits);      // only the listed tiles
  dim3 block(NT);
  // the plain form (the decoder-step products on 64 x 64 tiles, the [40 960-row] forward / dX products on 128 x 64): its own instance (see the kernel).
  // Measured on the VP step (round 4's lab-build A/B): 22.49 -> 22.34 ms with the 64 x 64 instance, -0.05 ms more with the 128 x 64 one
  const bool plain_form = ((BM == 64 && BN == 64) || (BM == 128 && BN == 64)) && !mansy_var_no_plain(p.ep.variant) && !a_kmajor && splits == 1 && !p.A2 && !p.ep.tile_list && !p.ep.tile_nrange && !p.ep.tile_krange &&
                          !p.ep.a_rowsum && !p.ep.accumulate && p.ep.split_slab == 0 && p.c_vec_ok && !(p.col_group > 0 && (int)grid.x > p.col_group);
  if (plain_form) {
    constexpr int PBM = (BM == 128 && BN == 64) ? 128 : 64;      // (instantiated for the two tile shapes that have a plain form)
    if (b_kmajor) MANSY_GEMM_LAUNCH((gemm_f32_dma_kernel<PBM, 64, false, true, true>), grid, block, st, p);
    else MANSY_GEMM_LAUNCH((gemm_f32_dma_kernel<PBM, 64, false, false, true>), grid, block, st, p);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  if (!a_kmajor && !b_kmajor) MANSY_GEMM_LAUNCH((gemm_f32_dma_kernel<BM, BN, false, false>), grid, block, st, p);
  else if (!a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_f32_dma_kernel<BM, BN, false, true>), grid, block, st, p);
  else if (a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_f32_dma_kernel<BM, BN, true, true>), grid, block, st, p);
  else MANSY_GEMM_LAUNCH((gemm_f32_dma_kernel<BM, BN, true, false>), grid, block, st, p);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

template <int BM, int BN, bool VEC>
int launch_cfg(const GemmParams& p, int a_kmajor, int b_kmajor, int splits, hipStream_t st) {
  dim3 grid(mansy_ceil_div(p.N, BN), mansy_ceil_div(p.M, BM), splits);
  if (BM == 64 && BN == 64 && p.ep.tile_list) grid = dim3(p.ep.tile_list_n, 1, splits);      // only the listed tiles
  dim3 block(NT);
  if (!a_kmajor && !b_kmajor) MANSY_GEMM_LAUNCH((gemm_f32_kernel<BM, BN, false, false, VEC>), grid, block, st, p);
  else if (!a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_f32_kernel<BM, BN, false, true, VEC>), grid, block, st, p);
  else if (a_kmajor && b_kmajor) MANSY_GEMM_LAUNCH((gemm_f32_kernel<BM, BN, true, true, VEC>), grid, block, st, p);
  else MANSY_GEMM_LAUNCH((gemm_f32_kernel<BM, BN, true, false, VEC>), grid, block, st, p);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// ---- optional launch timing (bench.py roofline leg): a HIP event pair attached to every GEMM dispatch (gemm_tile.h: MANSY_GEMM_LAUNCH)
struct ProfState {
  bool on = false;
  std::vector<hipEvent_t> ev;     // pairs
  size_t used = 0;
  double flops = 0.0;
};
ProfState g_prof;
}  // namespace
namespace mansy_gemm { hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr; }
// (ABI 8: no process-wide precision, no kernel-selection knobs.  The precision travels with the call; the loop a product runs on follows from
// its shape and from GemmEpilogue::variant -- mansy_kernels.h, MANSY_VARIANT_* in include/mansy_hip.h.  The column-group width of the XCD-aware
// tile order defaults to 12: profiles/r04_gemm_colgroup.txt, [40 960, 1 536, 512] fetch 578 -> 211 MB per launch at the same duration.)

extern "C" int mansy_prof_gemm_enable(int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  g_prof.flops = 0.0;
  return MANSY_OK;
}
// Synchronises the device, returns the summed duration (ms) of the GEMM launches recorded since enable,
// their count and their exact FLOPs (2*M*N*K each); resets the recorder.
extern "C" int mansy_prof_gemm_collect(double* total_ms, long long* launches, double* flops) {
  MANSY_HIP_CHECK(hipDeviceSynchronize());
  double ms = 0.0;
  for (size_t i = 0; i + 1 < g_prof.used; i += 2) {
    float t = 0.f;
    MANSY_HIP_CHECK(hipEventElapsedTime(&t, g_prof.ev[i], g_prof.ev[i + 1]));
    ms += t;
  }
  if (total_ms) *total_ms = ms;
  if (launches) *launches = (long long)(g_prof.used / 2);
  if (flops) *flops = g_prof.flops;
  g_prof.used = 0;
  g_prof.flops = 0.0;
  return MANSY_OK;
}

// Small products (at most WSK_MAX_TILES output tiles of 64 x 64 and <= 256 workgroups incl. K splits) run on the wave-split-K loop: the PPO cycle's
// products have 80-160 tiles; at 256 -- the half-batch decoder products of the VP step, two of them in flight on two streams -- the 64 x 64 loop
// wins inside the step (profiles/r04_f32_wsk_threshold.txt).  MANSY_VARIANT_NO_WSK / _NO_WSK_TN (per call) put one product back on the 64 x 64 loop.
constexpr int WSK_MAX_TILES = 200;
int mansy_gemm_wsk_tn_enabled() { return !mansy_var_no_wsk_tn(0); }

// ---- two independent small products as ONE launch (gemm_f32_wsk_dual_kernel).  Between mansy_gemm_pair_begin() and mansy_gemm_pair_end() the products
// that resolve to the wave-split-K loop are collected instead of launched (everything else launches at once: the caller states that the products
// between the two calls do not depend on each other); pair_end launches a (TN, NN) couple as one grid, anything else one by one.
struct WskPending { GemmParams p; int variant; dim3 grid; };       // variant: 0 = <false, false>, 1 = <false, true>, 2 = <true, true>
static thread_local bool g_pair_open = false;
static thread_local std::vector<WskPending> g_pair;
static int wsk_launch_one(const WskPending& w, hipStream_t st) {
  if (w.variant == 0) MANSY_GEMM_LAUNCH((gemm_f32_wsk_kernel<false, false>), w.grid, dim3(NT), st, w.p);
  else if (w.variant == 1) MANSY_GEMM_LAUNCH((gemm_f32_wsk_kernel<false, true>), w.grid, dim3(NT), st, w.p);
  else MANSY_GEMM_LAUNCH((gemm_f32_wsk_kernel<true, true>), w.grid, dim3(NT), st, w.p);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
static int wsk_launch(const GemmParams& p, int variant, dim3 grid, hipStream_t st) {
  WskPending w{p, variant, grid};
  if (g_pair_open) { g_pair.push_back(w); return MANSY_OK; }
  return wsk_launch_one(w, st);
}
// Capture instead of launch (round 5, the persistent rollout of ppo_engine.hip): between capture_begin and capture_end the products that resolve to
// the wave-split-K loop are collected exactly as they would be launched -- parameters, instance, grid -- and handed to the caller, who runs the same
// loop body on the same blocks inside a kernel of its own.  Returns the number collected (products that resolved to another loop were launched
// normally and are not in the list: the caller checks the count).
int mansy_gemm_capture_begin() {
  if (g_pair_open || g_prof.on) return 0;      // (the launch recorder stamps event pairs on real dispatches: nothing to capture while it runs)
  g_pair_open = true; g_pair.clear();
  return 1;
}
int mansy_gemm_capture_end(mansy_gemm::GemmParams* out, int* variant, int* gx, int* gy, int* gz, int max_n) {
  g_pair_open = false;
  std::vector<WskPending> w;
  w.swap(g_pair);
  int n = 0;
  for (const WskPending& x : w) {
    if (n >= max_n) break;
    out[n] = x.p; variant[n] = x.variant; gx[n] = (int)x.grid.x; gy[n] = (int)x.grid.y; gz[n] = (int)x.grid.z; ++n;
  }
  return (int)w.size() <= max_n ? n : -1;
}
int mansy_gemm_pair_begin() {
  if (mansy_var_no_pair(0) || g_prof.on || g_pair_open) return 0;      // (the launch recorder stamps one event pair per product: no pairing while it runs)
  g_pair_open = true; g_pair.clear();
  return 1;
}
int mansy_gemm_pair_end(hipStream_t st) {
  g_pair_open = false;
  std::vector<WskPending> w;
  w.swap(g_pair);
  if (w.size() == 2 && w[0].variant + w[1].variant == 3 && (w[0].variant == 2 || w[1].variant == 2)) {
    const WskPending& tn = w[0].variant == 2 ? w[0] : w[1];
    const WskPending& nn = w[0].variant == 2 ? w[1] : w[0];
    const WskGrid g1{(int)tn.grid.x, (int)tn.grid.y, (int)tn.grid.z}, g2{(int)nn.grid.x, (int)nn.grid.y, (int)nn.grid.z};
    const dim3 grid(g1.x * g1.y * g1.z + g2.x * g2.y * g2.z);
    __atomic_fetch_add(&g_mansy_launch_count, 1ull, __ATOMIC_RELAXED);
    hipLaunchKernelGGL(gemm_f32_wsk_dual_kernel, grid, dim3(NT), 0, st, tn.p, g1, nn.p, g2);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  for (const WskPending& x : w) { const int rc = wsk_launch_one(x, st); if (rc) return rc; }
  return MANSY_OK;
}

// tile codes: 128 -> 128x128, 96 -> 128x64 (LDS-DMA loop only), 64 -> 64x64
static int gemm_dispatch(const GemmParams& p, int tile, bool dma, int bf, int a_kmajor, int b_kmajor, int splits, hipStream_t st) {
  if (bf && p.ep.b_planes && !a_kmajor && splits == 1 && !p.ep.tile_krange && (reinterpret_cast<uintptr_t>(p.ep.b_planes) & 15) == 0 &&
      p.ep.b_planes_ld % 8 == 0 && p.ep.b_plane_stride % 8 == 0 && (bf != 1 || ((reinterpret_cast<uintptr_t>(p.A) & 15) == 0 && p.lda % 4 == 0)))
    return mansy_gemm_bf16p_dispatch(p, tile, bf, st);            // weights pre-split into planes: B by LDS-DMA
  if (bf) return mansy_gemm_bf16s_dispatch(p, tile, bf, a_kmajor, b_kmajor, splits, st);
  if (dma && !mansy_var_no_wsk(p.ep.variant) && tile == 64 && (p.c_vec_ok || p.ep.accumulate)) {
    // a launch that cannot fill the chip: 32 x 32 blocks, the K-tiles split over the workgroup's four waves.  NT / NN: plain or slab-split stores,
    // any fused epilogue; TN (the weight-gradient products): accumulating (atomics) or slab-split, row-sum rider, tile list / ranges, paired problems
    const long long tiles64 = p.ep.tile_list ? (long long)p.ep.tile_list_n : (long long)mansy_ceil_div(p.M, 64) * mansy_ceil_div(p.N, 64);
    const bool small = tiles64 <= WSK_MAX_TILES && tiles64 * splits <= 256;      // few output tiles, and the K splits do not fill the chip either
    const bool store_ok = !p.ep.accumulate && (splits == 1 || p.ep.split_slab != 0);
    if (small && !a_kmajor && p.c_vec_ok && store_ok && !p.ep.tile_list && !p.ep.tile_nrange && !p.ep.a_rowsum && !p.A2) {
      dim3 grid(mansy_ceil_div(p.N, 32), mansy_ceil_div(p.M, 32), splits);
      return wsk_launch(p, b_kmajor ? 1 : 0, grid, st);
    }
    if (small && !mansy_var_no_wsk_tn(p.ep.variant) && a_kmajor && b_kmajor && !p.ep.tile_krange && (p.ep.accumulate || splits == p.splits_pp * (p.A2 ? 2 : 1)) &&
        (p.ep.accumulate || ((p.splits_pp == 1 || p.ep.split_slab != 0) && p.c_vec_ok)) && (!p.ep.tile_list || p.ep.tile_nrange)) {
      dim3 grid(mansy_ceil_div(p.N, 32), mansy_ceil_div(p.M, 32), splits);
      if (p.ep.tile_list) grid = dim3(4 * p.ep.tile_list_n, 1, splits);
      return wsk_launch(p, 2, grid, st);
    }
  }
  if (dma) {
    if (tile == 128) return launch_dma<128, 128>(p, a_kmajor, b_kmajor, splits, st);
    if (tile == 96) return launch_dma<128, 64>(p, a_kmajor, b_kmajor, splits, st);
    return launch_dma<64, 64>(p, a_kmajor, b_kmajor, splits, st);
  }
  if (p.vec_ok) {
    if (tile == 128) return launch_cfg<128, 128, true>(p, a_kmajor, b_kmajor, splits, st);
    return launch_cfg<64, 64, true>(p, a_kmajor, b_kmajor, splits, st);
  }
  if (tile == 128) return launch_cfg<128, 128, false>(p, a_kmajor, b_kmajor, splits, st);
  return launch_cfg<64, 64, false>(p, a_kmajor, b_kmajor, splits, st);
}

static long long tile_count(int M, int N, int tile) {
  const int bm = tile == 64 ? 64 : 128, bn = tile == 128 ? 128 : 64;
  return (long long)mansy_ceil_div(M, bm) * mansy_ceil_div(N, bn);
}

// force_tile: 0 = heuristics; 64 / 96 / 128 = that tile; negative = that tile on the register-staged loop (A/B tests).
int mansy_launch_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor, float* C, int ldc,
                          int M, int N, int K, const GemmEpilogue& ep, int force_tile, int force_splitk, hipStream_t st) {
  MANSY_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
  if (M == 0 || N == 0) return MANSY_OK;          // empty product (pointers of empty buffers may be null)
  MANSY_REQUIRE(A && B && C, "gemm: null pointer");
  GemmParams p;
  p.A = A; p.B = B; p.C = C; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.ep = ep;
  p.col_group = (ep.tile_nrange || ep.tile_list || ep.tile_krange) ? 0 : mansy_var_col_group(ep.variant);
  p.c_rmw_ok = (reinterpret_cast<uintptr_t>(C) & 15) == 0 && (!ep.pair_C || (reinterpret_cast<uintptr_t>(ep.pair_C) & 15) == 0) && ldc % 4 == 0 && N % 4 == 0;
  p.vec_ok = ((lda % 4) == 0) && ((ldb % 4) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
             ((reinterpret_cast<uintptr_t>(B) & 15) == 0) && (K % 4 == 0) && (!a_kmajor || M % 4 == 0) &&
             (!b_kmajor || N % 4 == 0);
  auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  p.c_vec_ok = al16(C) && (ldc % 4 == 0) && (N % 4 == 0) && (!ep.bias || al16(ep.bias)) &&
               (!ep.mask_src || (al16(ep.mask_src) && ep.mask_ld % 4 == 0)) && (!ep.resid || (al16(ep.resid) && ep.resid_ld % 4 == 0)) &&
               (!ep.pre_a || (al16(ep.pre_a) && (!ep.pre_b || al16(ep.pre_b)) && ep.pre_ld % 4 == 0 && ep.pre_col0 % 4 == 0));
  const bool plain = !ep.bias && !ep.relu && !ep.mask_src && ep.drop.p == 0.f && !ep.resid && !ep.pre_a;
  // LDS-DMA loop: whole 16-byte chunks and whole K-tiles only (it cannot zero-fill a K tail)
  const bool dma = p.vec_ok && K >= BK && K % BK == 0 && force_tile >= 0;
  // split-bf16 modes: same preconditions as the LDS-DMA loop; everything else stays on the fp32 loops
  const int bf = dma ? ep.prec : 0;
  if (!dma && p.vec_ok && force_tile >= 0 && plain && K % BK != 0 && K >= 8 * BK && ep.split_slab == 0 && !ep.tile_krange) {
    // long reduce dimension that is not a multiple of the K-tile (e.g. a dW over 3 276 rows): whole K-tiles on the LDS-DMA
    // loop, the < 32 leftover as a second, accumulating launch of the register-staged loop (sums and row sums are additive)
    const int Kmain = K / BK * BK;
    const int rc = mansy_launch_gemm_f32(A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, Kmain, ep, force_tile, force_splitk, st);
    if (rc) return rc;
    GemmEpilogue tail = ep; tail.accumulate = 1;
    const float* A2 = a_kmajor ? A + (long long)Kmain * lda : A + Kmain;
    const float* B2 = b_kmajor ? B + (long long)Kmain * ldb : B + Kmain;
    if (tail.pair_A) {      // the second problem of a paired launch has the same layout: its leftover rows start at Kmain too
      tail.pair_A = a_kmajor ? ep.pair_A + (long long)Kmain * lda : ep.pair_A + Kmain;
      tail.pair_B = b_kmajor ? ep.pair_B + (long long)Kmain * ldb : ep.pair_B + Kmain;
    }
    return mansy_launch_gemm_f32(A2, lda, a_kmajor, B2, ldb, b_kmajor, C, ldc, M, N, K - Kmain, tail, -64, 1, st);
  }
  if (ep.pair_A) {
    MANSY_REQUIRE(ep.pair_B && ep.pair_C && plain && ep.accumulate && ep.split_slab == 0, "gemm: a paired product needs a plain accumulating epilogue");
    auto al = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
    if (!dma || bf || !al(ep.pair_A) || !al(ep.pair_B)) {  // no second-problem support off the fp32 LDS-DMA loop: two launches
      GemmEpilogue e1 = ep; e1.pair_A = e1.pair_B = nullptr; e1.pair_C = e1.pair_rowsum = nullptr;
      const int rc = mansy_launch_gemm_f32(A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, e1, force_tile, force_splitk, st);
      if (rc) return rc;
      e1.a_rowsum = ep.pair_rowsum;
      return mansy_launch_gemm_f32(ep.pair_A, lda, a_kmajor, ep.pair_B, ldb, b_kmajor, ep.pair_C, ldc, M, N, K, e1, force_tile, force_splitk, st);
    }
    p.A2 = ep.pair_A; p.B2 = ep.pair_B; p.C2 = ep.pair_C; p.a_rowsum2 = ep.pair_rowsum;
    p.c_vec_ok = 0;                                        // accumulating: the scalar (atomic) epilogue is used anyway
  }
  const int n_prob = p.A2 ? 2 : 1;
  const bool can_split = plain && ep.accumulate && K >= 4096;
  int tile;
  if (force_tile) {
    tile = force_tile < 0 ? -force_tile : force_tile;
    const bool bf_planes = bf == 3 && ep.b_planes && !a_kmajor;      // the fragment-split loop has a 128x64 instance
    if ((!dma || (bf && !bf_planes)) && tile == 96) tile = 64;
  } else if (bf) {
    // split-bf16 loop: 128x128 (least operand traffic and split work per MFMA) once it fills the chip, else 64x64
    tile = (can_split || tile_count(M, N, 128) >= 256) ? 128 : 64;
  } else if (dma) {
    // LDS-DMA loop: 128x64 and 64x64 run the K loop at the same rate as 128x128 (tools/gemm_lab.hip) and quantise better
    // over 256 CUs.  cost ~ (workgroups per CU, rounded up) x tile area / relative efficiency; a lone workgroup per CU
    // (one wave per SIMD, nothing to overlap its barriers with) pays ~10 %.
    const int cand[2] = {96, 64};
    const double area_over_eff[2] = {8192.0 / 1.00, 4096.0 / 0.97};
    double best = 0.0;
    tile = 96;
    for (int c = 0; c < 2; ++c) {
      const long long per_cu = (tile_count(M, N, cand[c]) * (can_split ? 256 : 1) + 255) / 256;   // split-K fills the chip anyway
      const double cost = (double)per_cu * area_over_eff[c] * (per_cu == 1 ? 1.1 : 1.0);
      if (c == 0 || cost < best) { best = cost; tile = cand[c]; }
    }
  } else if (can_split) {
    tile = 128;     // dW products: split-K makes up the workgroups; small tiles re-stream both operands through L2
  } else {
    // register-staged loop: 128x128 when it alone gives two full rounds of resident workgroups, else 64x64
    tile = tile_count(M, N, 128) >= 512 ? 128 : 64;
  }
  // tile_krange is honoured by the 64-column tiles of the LDS-DMA / split-bf16 loops (ppo_engine.hip packs B only inside the ranges and
  // makes sure its operands qualify for those loops)
  if (ep.tile_krange && dma && tile == 128) tile = bf ? 64 : 96;
  if (tile != 64 || bf) p.ep.tile_nrange = nullptr;        // (fp32 loops with 64-row tiles only; without it every tile is computed -- still correct)
  if (tile != 64 || !p.ep.tile_nrange || bf || ep.tile_list_n < 1) { p.ep.tile_list = nullptr; p.ep.tile_list_n = 0; }
  const long long tiles = tile_count(M, N, tile) * n_prob;
  // split-K only for plain accumulating products (dW = dY^T X): partial sums are atomically added
  int splits = 1;
  // small accumulating weight-gradient product on the wave-split-K loop: the K-tiles are split inside the workgroup, so no split over workgroups
  // (every element of C then has ONE owner: a plain read-add-write instead of float atomics)
  const bool wsk_tn = dma && !bf && !mansy_var_no_wsk_tn(ep.variant) && tile == 64 && a_kmajor && b_kmajor && plain && ep.accumulate && !p.ep.tile_list &&
                      tiles <= WSK_MAX_TILES && ep.split_slab == 0;      // (any K: at K = 3264 the split-K 64 x 64 loop is 3 us faster WITHOUT the bias-gradient
                                                                               // rider -- 17.5 vs 20.6 us -- but its rider adds (column tiles x splits)-way atomics per row)
  if (force_splitk > 0) splits = force_splitk;
  else if (wsk_tn) splits = 1;
  else if (plain && ep.accumulate && tiles < 256) {
    // fill ONE round of resident workgroups, never spill a few into a second: 2 per CU for the 128x128 / 64x64 loops as
    // dispatched here, 3 per CU for the 128x64 LDS-DMA loop (48 KB LDS, 136 VGPRs) -- tools/dw_split_sweep.py
    splits = (int)(((dma && !bf && tile == 96) ? 768 : 512) / tiles);
    if (splits < 1) splits = 1;
    const int max_splits = K / (BK * 8) > 0 ? K / (BK * 8) : 1;
    if (splits > max_splits) splits = max_splits;
  }
  MANSY_REQUIRE(splits == 1 || plain, "gemm: split-K requires a plain epilogue");
  MANSY_REQUIRE(ep.split_slab == 0 || (!ep.accumulate && (ep.split_slab % 4) == 0), "gemm: slab split-K stores, it does not accumulate");
  int kps = mansy_ceil_div(mansy_ceil_div(K, splits), BK) * BK;
  if (kps <= 0) kps = BK;
  splits = K > 0 ? mansy_ceil_div(K, kps) : 1;
  p.k_per_split = kps;
  p.splits_pp = splits;
  splits *= n_prob;                                        // grid.z: problem-major
  if (!g_prof.on) return gemm_dispatch(p, tile, dma, bf, a_kmajor, b_kmajor, splits, st);
  if (g_prof.used + 2 > g_prof.ev.size()) {
    for (int i = 0; i < 2; ++i) { hipEvent_t e; MANSY_HIP_CHECK(hipEventCreate(&e)); g_prof.ev.push_back(e); }
  }
  mansy_gemm::g_ev_start = g_prof.ev[g_prof.used]; mansy_gemm::g_ev_stop = g_prof.ev[g_prof.used + 1];     // stamped by the dispatch itself
  const int rc = gemm_dispatch(p, tile, dma, bf, a_kmajor, b_kmajor, splits, st);
  mansy_gemm::g_ev_start = mansy_gemm::g_ev_stop = nullptr;
  g_prof.used += 2;
  g_prof.flops += 2.0 * (double)M * (double)N * (double)K * n_prob * (p.ep.tile_nrange ? (double)ep.flops_frac : 1.0);
  return rc;
}


// bf16-storage products (round 6; csrc/gemm_bf16a.hip): operands are bf16 images in HBM (ep.a16 + a weight plane ep.b_planes for the forward / dX
// form, ep.a16 + ep.b16 K-major for the weight-gradient form); C fp32 (nullable when ep.c16 takes the output).  force_tile: 0 = by shape, 64 / 96 / 128.
int mansy_launch_gemm_bf16a(int a_kmajor, int b_kmajor, float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep, int force_tile, int force_splitk,
                            hipStream_t st) {
  MANSY_REQUIRE(M >= 0 && N >= 0 && K >= 0, "gemm: negative dimension");
  if (M == 0 || N == 0) return MANSY_OK;
  MANSY_REQUIRE(ep.a16 && (C || ep.c16), "bf16-storage product: needs a16 and an output (C or c16)");
  GemmParams p;
  p.A = nullptr; p.B = nullptr; p.C = C; p.lda = ep.a16_ld; p.ldb = 0; p.ldc = ldc; p.M = M; p.N = N; p.K = K; p.ep = ep;
  auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  auto al8 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 7) == 0; };
  p.vec_ok = 1;
  p.c_vec_ok = al16(C) && (!C || ldc % 4 == 0) && (N % 4 == 0) && (!ep.bias || al16(ep.bias)) && (!ep.c16 || (al8(ep.c16) && ep.c16_ld % 4 == 0)) &&
               (!ep.mask_src || (al16(ep.mask_src) && ep.mask_ld % 4 == 0)) && (!ep.resid || (al16(ep.resid) && ep.resid_ld % 4 == 0)) && !ep.pre_a;
  int rc;
  double flops = 2.0 * (double)M * (double)N * (double)K;
  const bool timed = g_prof.on;
  if (timed) {
    if (g_prof.used + 2 > g_prof.ev.size()) { for (int i = 0; i < 2; ++i) { hipEvent_t e; MANSY_HIP_CHECK(hipEventCreate(&e)); g_prof.ev.push_back(e); } }
    mansy_gemm::g_ev_start = g_prof.ev[g_prof.used]; mansy_gemm::g_ev_stop = g_prof.ev[g_prof.used + 1];
  }
  if (a_kmajor && b_kmajor) {
    MANSY_REQUIRE(ep.b16 && C && !ep.c16 && !ep.bias && !ep.relu && !ep.mask_src && ep.drop.p == 0.f && !ep.resid && ep.accumulate,
                  "bf16-storage weight-gradient product: plain accumulating epilogue, K-major b16");
    // split-K over the rows.  Eight-wave workgroups (two K groups: gemm_bf16a.hip), one per CU, at least 8 K-tiles of 64 per group; a reduce dimension too
    // short for that (or force_tile == 64): four-wave workgroups, two per CU, at least 8 K-tiles per split
    const long long tiles = (long long)mansy_ceil_div(M, 128) * mansy_ceil_div(N, 128);
    int kgroups = (force_tile == 64 || K < 64 * 16) ? 1 : 2;
    int splits = force_splitk > 0 ? force_splitk : (int)((kgroups == 2 ? 256 : 512) / tiles);
    if (splits < 1) splits = 1;
    const int max_splits = K / (64 * 8 * kgroups) > 0 ? K / (64 * 8 * kgroups) : 1;
    if (splits > max_splits) splits = max_splits;
    int kps = mansy_ceil_div(mansy_ceil_div(K, splits), 64) * 64;
    if (kps <= 0) kps = 64;
    splits = mansy_ceil_div(K, kps);
    p.k_per_split = kps; p.splits_pp = splits;
    rc = mansy_gemm_bf16a_tn(p, splits, kgroups, st);
  } else {
    MANSY_REQUIRE(!a_kmajor && ep.b_planes && !ep.accumulate && ep.split_slab == 0 && p.c_vec_ok,
                  "bf16-storage forward / dX product: K-contiguous a16, a weight plane, a storing row-major epilogue");
    int tile = force_tile;
    if (!tile) {
      // per-shape timing: profiles/r06_gemm_bf16a_lab.txt.  [40 960-row] products: 128 x 128 (least L2 -> LDS traffic per flop) once that fills the chip;
      // the [4 096-row] decoder-step products: 128 x 64 for wide outputs (N >= 1024: 13.7 us against 15.7 on 64 x 64), 64 x 64 otherwise (7.9 against 9.6)
      const long long t128 = (long long)mansy_ceil_div(M, 128) * mansy_ceil_div(N, 128);
      const long long t96 = (long long)mansy_ceil_div(M, 128) * mansy_ceil_div(N, 64);
      tile = t128 >= 512 ? 128 : ((t96 >= 192 && (N >= 1024 || t96 > 512)) ? 96 : 64);
    }
    p.k_per_split = K; p.splits_pp = 1;
    rc = mansy_gemm_bf16a_nn(p, tile, st);
  }
  if (timed) { mansy_gemm::g_ev_start = mansy_gemm::g_ev_stop = nullptr; g_prof.used += 2; g_prof.flops += flops; }
  return rc;
}

int mansy_gemm_effective_splits(int K, int requested) {
  if (requested <= 1 || K <= 0) return 1;
  int kps = mansy_ceil_div(mansy_ceil_div(K, requested), BK) * BK;
  if (kps <= 0) kps = BK;
  return mansy_ceil_div(K, kps);
}
