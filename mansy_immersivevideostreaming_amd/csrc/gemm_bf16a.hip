// bf16-STORAGE products of the MANSY_PREC_BF16 perf mode (round 6): both operands are bf16 IN HBM and are staged by LDS-DMA as they are
// -- no conversion on the way into LDS, half the HBM and L2 -> LDS bytes of the fp32-operand loops (gemm_bf16s.hip rounds fp32 operands at
// fragment read: same arithmetic, twice the traffic) -- one v_mfma_f32_32x32x16_bf16 product, fp32 accumulate, the shared fused epilogue
// (gemm_tile.h) which can store C as fp32, as bf16 (GemmEpilogue::c16), or both.
//
//   NN  (forward and dX products):  A16 [M, K] K-contiguous (an activation's bf16 image), B = a weight's leading bf16 plane [N, K]
//       K-contiguous (GemmEpilogue::b_planes: W for the forward, W^T for dX).  K-tile = 64 bf16 = 128-byte rows: the LDS image, its XOR
//       swizzle and the DMA piece geometry are those of the fp32 LDS-DMA loop (gemm_f32.hip); NS-stage ring with counted vmcnt waits and one
//       raw s_barrier per K-tile as gemm_bf16h_kernel.  Fragment of lane (r = lane & 31, h = lane >> 5) at k-step s: the 16 bytes at chunk
//       2 s + h of row r -- one ds_read_b128 per operand block and k-step.
//   TN  (weight gradients dW = dY^T X): A16 = dY [K = rows, M] and B16 = X [K = rows, N], both K-MAJOR (rows of the activation slabs as
//       they lie in HBM).  A K-tile is 64 rows of 128 columns = 256-byte LDS rows; the MFMA wants 8 consecutive k of ONE column per lane,
//       i.e. a column of the image: ds_read_b64_tr_b16 (gfx950's transposing LDS read: per 16-lane group a 4-row x 16-column block,
//       delivered column-major) -- two per operand block and k-step, on the guide's conflict-free image (b): 16-byte chunk ch of row k at
//       256 k + 16 (ch ^ (((k & 3) << 2) | ((k >> 2) & 3))).  Split-K over the rows with atomic accumulation (as the fp32 dW products);
//       the bias-gradient rider (row sums of dY^T) is one more MFMA per A block against a fragment of ones, in the first column tile only.
#include <type_traits>
#include "gemm_tile.h"

using namespace mansy_gemm;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int BK16 = 64;          // bf16 elements per K-tile (128 bytes per K-contiguous row)

// ------------------------------------------------------------------------------------------------------------------------------ NN
// workgroups of one CU's 160 KB of LDS (ring or the epilogue's C staging, whichever is larger), at most 4
constexpr int nn_wgs_per_cu(int BM, int BN, int NS) {
  const int ring = NS * (BM + BN) * 128, c = BM * (BN + 4) * 4, lds = ring > c ? ring : c;
  return 160 * 1024 / lds > 4 ? 4 : 160 * 1024 / lds;
}

template <int BM, int BN, int NS>
__global__ __launch_bounds__(NT, nn_wgs_per_cu(BM, BN, NS)) void gemm_bf16a_nn_kernel(GemmParams p) {
  constexpr int TM = BM / 64, TN = BN / 64 > 0 ? BN / 64 : 1, PA = BM / 32, PB = BN / 32, D = NS - 1;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE_BYTES = A_BYTES + B_BYTES;
  constexpr int C_FLOATS = BM * (BN + 4);
  constexpr int SMEM_FLOATS = (NS * STAGE_BYTES / 4) > C_FLOATS ? (NS * STAGE_BYTES / 4) : C_FLOATS;
  constexpr int PPT = PA + PB;                                                                  // DMA pieces per wave and K-tile
  static_assert(BN >= 64 && NS >= 2 && NS <= 6 && (D - 1) * PPT <= 63, "piece counts must fit the counted waits");
  __shared__ __attribute__((aligned(1024))) float smem[SMEM_FLOATS];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int tile_x, tile_y;
  {   // XCD-aware bijective remap: consecutive tiles (row panel by row panel) stay on one XCD, whose L2 then holds ONE A panel and the weight plane
    const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int nk = p.K / BK16;
  const unsigned short* const A16 = p.ep.a16;
  const int lda = p.ep.a16_ld, ldb = p.ep.b_planes_ld;

  // per-lane source byte offsets of this wave's DMA pieces (lane L of a piece lands at LDS piece base + 16 L: row L >> 3, slot L & 7 of a
  // 128-byte row; the XOR swizzle is applied on the SOURCE side: slot s of row `row` receives chunk s ^ ((row >> 1) & 7))
  unsigned voa[PA], vob[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    voa[i] = (unsigned)(((min(m0 + row, p.M - 1) - m0) * lda + c * 8) * 2);
  }
#pragma unroll
  for (int i = 0; i < PB; ++i) {
    const int row = i * 32 + wave * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((row >> 1) & 7);
    vob[i] = (unsigned)(((min(n0 + row, p.N - 1) - n0) * ldb + c * 8) * 2);
  }
  const unsigned short* const ca = A16 + (long long)m0 * lda;
  const unsigned short* const cb = p.ep.b_planes + (long long)n0 * ldb;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * 1024u);
  auto dma = [&](int stage, int kt) {
    const unsigned base = lds0 + (unsigned)(stage * STAGE_BYTES);
    const unsigned short* a_corner = ca + (long long)kt * BK16;
    const unsigned short* b_corner = cb + (long long)kt * BK16;
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(voa[i], a_corner, base + (unsigned)i * 4096u);
#pragma unroll
    for (int i = 0; i < PB; ++i) glds16(vob[i], b_corner, base + (unsigned)A_BYTES + (unsigned)i * 4096u);
  };
  // fragment byte offsets inside a stage: block row (i or j), k-step s
  int fa[TM][4], fb[TN][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * (BM / 2) + i * 32 + r; fa[i][s] = row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4); }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = wn * (BN / 2) + j * 32 + r; fb[j][s] = A_BYTES + row * 128 + (((2 * s + h) ^ ((row >> 1) & 7)) << 4); }
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
    if (D >= 5 && ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D >= 5 ? 4 * PPT : 0) : "memory");
    else if (D >= 4 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D >= 4 ? 3 * PPT : 0) : "memory");
    else if (D >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D >= 3 ? 2 * PPT : 0) : "memory");
    else if (D >= 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + D < nk) dma(cur == 0 ? NS - 1 : cur - 1, kt + D);       // into the stage tile kt - 1 occupied
    const char* const st_l = reinterpret_cast<const char*>(smem) + cur * STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(st_l + fa[i][s]);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(st_l + fb[j][s]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads retired before the barrier that frees this buffer
    cur = cur == NS - 1 ? 0 : cur + 1;
  }
  __syncthreads();                                        // staging LDS idle (every DMA was waited for): the epilogue reuses it
  gemm_epilogue<BM, BN, SMEM_FLOATS>(p, acc, smem, m0, n0, tid, 0, p.C);
}

// ------------------------------------------------------------------------------------------------------------------------------ TN
// ds_read_b64_tr_b16 (through its builtin, so that the compiler counts lgkmcnt for it): lane 4 q + p of a 16-lane group supplies the address of row q,
// columns 4 p .. 4 p + 3 of the group's 4 x 16 block; lane i of the group receives column i, row q in element q.  EXEC must be all ones (no divergence
// around it).  (Round 6 first issued it by inline asm with a separate `s_waitcnt lgkmcnt(0)` statement: nothing tied the results to the wait, the compiler
// scheduled the consuming MFMAs above it, and the kernel produced NaN weight gradients once two processes shared the device -- tools/vp_dp2_probe.py.)
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read(unsigned lds_byte_address) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(uintptr_t)lds_byte_address);
}

// byte offset of 16-byte chunk ch (0..15) of row k in a [64][128 x bf16] image with 256-byte rows (guide T10, image (b))
__device__ __forceinline__ unsigned tn_off(int k, int ch) { return (unsigned)(256 * k + 16 * (ch ^ (((k & 3) << 2) | ((k >> 2) & 3)))); }

// KG = 2 ("K groups"): a workgroup of EIGHT waves; waves 0-3 take the even K-tiles of the workgroup's K range, waves 4-7 the odd ones, each group on a ring
// of its own, and the two partial tiles are added through LDS before the atomics.  The atomics are the expensive part of a split-K product here (fp32
// atomic adds of all XCDs on one gradient run memory-side: tools/fill_probe_tn.hip -- [512, 512, 40960]: K loop 26 us at 32 splits / 36 us at 16, atomics
// 30 us at 32 splits / 14 us at 16): one 8-wave workgroup per CU has the K loop of 32 four-wave splits and the atomics of 16.
template <int NS, int KG>
__global__ __launch_bounds__(NT * KG, KG == 1 ? 2 : 1) void gemm_bf16a_tn_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, D = NS - 1;
  static_assert(NS == 2 && (KG == 1 || KG == 2), "two-stage ring; one or two K groups");
  constexpr int A_BYTES = BK16 * 256, B_BYTES = BK16 * 256, STAGE_BYTES = A_BYTES + B_BYTES;      // 32 KB per stage
  constexpr int PA = 4, PB = 4, PPT = PA + PB;                                                     // 16 pieces of 1 KB per operand tile, 4 per wave
  __shared__ __attribute__((aligned(1024))) char smem_all[KG * NS * STAGE_BYTES];

  const int tid = threadIdx.x & (NT - 1), kg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NT));      // position inside the K group; the group
  char* const smem = smem_all + kg * (NS * STAGE_BYTES);
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  int split, tile_x, tile_y;
  {   // XCD-aware bijective remap (workgroup ids go round-robin over the 8 XCDs): ALL output tiles of one K split run on ONE XCD, so a split's rows of dY and X
      // are fetched from HBM / the Infinity Cache into ONE L2 and every other tile of the split hits there (as dispatched, the tiles sharing a panel sat on 4
      // (dY) and 2 (X) different XCDs and every panel crossed the fabric that many times)
    const int per = gridDim.x * gridDim.y, nwg = per * gridDim.z, orig = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    const int q = nwg >> 3, rem = nwg & 7, xcd = orig & 7, local = orig >> 3;
    const int t = (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + local;
    split = t / per;
    const int tt = t - split * per;
    tile_y = tt / gridDim.x; tile_x = tt - tile_y * gridDim.x;
  }
  const int m0 = tile_y * BM, n0 = tile_x * BN;
  const int k_begin = split * p.k_per_split, k_end = min(p.K, k_begin + p.k_per_split);
  const int nk_all = (k_end - k_begin) / BK16;
  const int nk = (nk_all - kg + KG - 1) / KG, nit = (nk_all + KG - 1) / KG;      // this group's K-tiles (tile it of the group = K-tile it * KG + kg); iterations (barriers) of the workgroup
  const unsigned short* const A16 = p.ep.a16;           // [K][M] (dY rows), lda
  const unsigned short* const B16 = p.ep.b16;           // [K][N] (X rows), ldb
  const int lda = p.ep.a16_ld, ldb = p.ep.b16_ld;

  // DMA: piece i of wave w = rows (i * 16 + w * 4) .. + 4 of the tile (4 rows x 256 B = 1 KB); lane L -> row L >> 4, slot L & 15; source chunk =
  // slot ^ key(row).  Columns beyond M / N are clamped onto the last whole 8-column chunk (their accumulators are never stored).
  unsigned voa[PA], vob[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int k = i * 16 + wave * 4 + (lane >> 4);
    const int ch = (lane & 15) ^ (((k & 3) << 2) | ((k >> 2) & 3));
    voa[i] = (unsigned)((k * lda + min(ch * 8, p.M - m0 - 8)) * 2);
    vob[i] = (unsigned)((k * ldb + min(ch * 8, p.N - n0 - 8)) * 2);
  }
  const unsigned short* const ca = A16 + (long long)k_begin * lda + m0;
  const unsigned short* const cb = B16 + (long long)k_begin * ldb + n0;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + (unsigned)wave * 1024u);
  auto dma = [&](int stage, int kt) {
    const unsigned base = lds0 + (unsigned)(stage * STAGE_BYTES);
    const unsigned short* a_corner = ca + (long long)(kt * KG + kg) * BK16 * lda;
    const unsigned short* b_corner = cb + (long long)(kt * KG + kg) * BK16 * ldb;
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16(voa[i], a_corner, base + (unsigned)i * 4096u);
#pragma unroll
    for (int i = 0; i < PB; ++i) glds16(vob[i], b_corner, base + (unsigned)A_BYTES + (unsigned)i * 4096u);
  };
  // transposed-read addresses: 16-lane group g = lane >> 4: column block cb16 = g & 1 (columns 16 cb16 ..), k half hh = g >> 1 (k = 8 hh ..);
  // within the group lane 4 q + pp supplies row q, columns 4 pp ..: chunk = (column >> 3), + 8 bytes for the odd half of the chunk
  const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
  const int cb16 = g & 1, hh = g >> 1;
  unsigned ta[2][2], tb[2][2];        // [block][t]: byte offset at k-step 0 (k-step s adds s * 16 rows); t = which 4 of the lane's 8 k
#pragma unroll
  for (int blk = 0; blk < 2; ++blk)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int k = 8 * hh + 4 * t + q;
      const int col_a = wm * 64 + blk * 32 + 16 * cb16 + 4 * pp, col_b = wn * 64 + blk * 32 + 16 * cb16 + 4 * pp;
      ta[blk][t] = tn_off(k, col_a >> 3) + 8u * ((col_a >> 2) & 1);
      tb[blk][t] = (unsigned)A_BYTES + tn_off(k, col_b >> 3) + 8u * ((col_b >> 2) & 1);
    }
  // (row k + 16 s has the same swizzle key as row k -- the key depends on k & 15 only -- so k-step s is a constant + 4096 s bytes)

  f32x16 acc[2][2], rs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int e = 0; e < 16; ++e) rs[i][e] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }
  const bool want_rs = __builtin_amdgcn_readfirstlane((int)(p.ep.a_rowsum != nullptr && tile_x == 0 && wn == 0)) != 0;   // wave-uniform, in an SGPR       // bias gradient: row sums of dY^T, taken once per row panel
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nk) dma(d, d);
  const unsigned smem_base = (unsigned)(uintptr_t)smem;
  // (the bias-gradient rider is decided OUTSIDE the K loop -- two copies of the loop -- so that each k-step is one basic block the scheduler can pipeline)
  auto k_loop = [&](auto with_rs) {
  constexpr bool RS = decltype(with_rs)::value;
  int cur = 0;
  for (int kt = 0; kt < nit; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (two stages: the one tile in flight is the one wanted)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + D < nk) dma(cur == 0 ? NS - 1 : cur - 1, kt + D);
    if (KG > 1 && kt >= nk) continue;                         // (an odd K-tile count: the second group sits out the last iteration, barrier included above)
    const unsigned st_l = smem_base + (unsigned)(cur * STAGE_BYTES);
    // k-step s + 1's eight transposed reads are issued before k-step s's MFMAs (the compiler tracks lgkmcnt for the builtin: the MFMAs of step s wait for
    // exactly their own reads)
    bf16x8 af[2][2], bf[2][2];
    auto frag = [&](int buf, int s) {
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        const bf16x4 a0 = tr_read(st_l + ta[blk][0] + 4096u * s), a1 = tr_read(st_l + ta[blk][1] + 4096u * s);
        const bf16x4 b0 = tr_read(st_l + tb[blk][0] + 4096u * s), b1 = tr_read(st_l + tb[blk][1] + 4096u * s);
        af[buf][blk] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        bf[buf][blk] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    frag(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) frag((s + 1) & 1, s + 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0);
        if (RS) rs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], ones, rs[i], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads retired before the barrier that frees this buffer
    cur = cur == NS - 1 ? 0 : cur + 1;
  }
  };
  if (want_rs) k_loop(std::true_type{}); else k_loop(std::false_type{});
  // accumulate: C[m][n] += acc (fp32 atomics: the K splits and the steps' other products add into the same gradient)
  const int r = lane & 31, h = lane >> 5;
  auto add_block_row = [&](auto which) {
    constexpr int i = decltype(which)::value;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + r;
      if (col < p.N) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          if (row < p.M) atomicAdd(p.C + (long long)row * p.ldc + col, acc[i][j][e]);
        }
      }
    }
    if (want_rs && r == 0) {          // every column of rs holds the row sums: column 0's lanes (0 and 32) add them
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (row < p.M) atomicAdd(p.ep.a_rowsum + row, rs[i][e]);
      }
    }
  };
  if (KG == 2) {
    // the two groups' partial tiles: group g keeps block row i == g, hands block row 1 - g to the other group through LDS (the rings are idle), and issues
    // the atomics of its own block row only -- 32 staged floats per thread (+ 16 of the row sums), each read back by the thread at the same position.
    // (Block rows are indexed by CONSTANTS in each branch: a run-time index into the accumulator arrays would move them to scratch memory.)
    float* const xch = reinterpret_cast<float*>(smem_all);
    auto hand_over = [&](auto send_c) {
      constexpr int SEND = decltype(send_c)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[((SEND * 8 + wave * 2 + j) * 16 + e) * 64 + lane] = acc[SEND][j][e];
      if (want_rs) {
#pragma unroll
        for (int e = 0; e < 16; ++e) xch[16384 + ((SEND * 4 + wave) * 16 + e) * 64 + lane] = rs[SEND][e];
      }
    };
    auto take = [&](auto keep_c) {
      constexpr int KEEP = decltype(keep_c)::value;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[KEEP][j][e] += xch[((KEEP * 8 + wave * 2 + j) * 16 + e) * 64 + lane];
      if (want_rs) {
#pragma unroll
        for (int e = 0; e < 16; ++e) rs[KEEP][e] += xch[16384 + ((KEEP * 4 + wave) * 16 + e) * 64 + lane];
      }
    };
    __syncthreads();
    if (kg == 0) hand_over(std::integral_constant<int, 1>{}); else hand_over(std::integral_constant<int, 0>{});
    __syncthreads();
    if (kg == 0) { take(std::integral_constant<int, 0>{}); add_block_row(std::integral_constant<int, 0>{}); }
    else { take(std::integral_constant<int, 1>{}); add_block_row(std::integral_constant<int, 1>{}); }
  } else {
    add_block_row(std::integral_constant<int, 0>{});
    add_block_row(std::integral_constant<int, 1>{});
  }
}

}  // namespace

// NN: A16 [M, K] bf16 (ep.a16, ep.a16_ld), B = ep.b_planes [N, K] bf16.  tile: 64 (64 x 64, 3 stages), 96 (128 x 64, 2 stages) or 128 (128 x 128, 2 stages).
int mansy_gemm_bf16a_nn(const GemmParams& p, int tile, hipStream_t st) {
  auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  MANSY_REQUIRE(p.ep.a16 && p.ep.b_planes && al16(p.ep.a16) && al16(p.ep.b_planes) && p.ep.a16_ld % 8 == 0 && p.ep.b_planes_ld % 8 == 0 && p.K % BK16 == 0 &&
                    p.K >= BK16, "bf16-storage product: operands must be 16-byte aligned with leading dimensions %% 8 == 0 and K %% 64 == 0");
  dim3 block(NT);
  // TWO-stage rings: what bounds these loops is how many workgroups a CU holds (each one's cold first fetch, epilogue and store drain overlap the others' K
  // loops), not how deep one workgroup prefetches -- profiles/r06_gemm_bf16a_lab.txt: 128 x 128 with 2 stages (2 per CU) 47 us on [40960, 512, 512] against
  // 57 with 3 stages (1 per CU) and 60 with 4; tools/fill_probe.hip: the K loop alone is 20-26 us of it, the 42 MB of stores 15 us, nothing overlapped.
  // (Round 6 also tried PERSISTENT workgroups for these shapes -- two per CU walking tile lists, the next tile's first K-tile in flight under the epilogue,
  // staggered starts: 10-14 % faster launch by launch on rotating buffers (profiles/r06_gemm_bf16a_nnp_lab.txt), nothing in the step; not kept.)
  if (tile == 128) { dim3 grid(mansy_ceil_div(p.N, 128), mansy_ceil_div(p.M, 128), 1); MANSY_GEMM_LAUNCH((gemm_bf16a_nn_kernel<128, 128, 2>), grid, block, st, p); }
  else if (tile == 96) { dim3 grid(mansy_ceil_div(p.N, 64), mansy_ceil_div(p.M, 128), 1); MANSY_GEMM_LAUNCH((gemm_bf16a_nn_kernel<128, 64, 2>), grid, block, st, p); }
  else { dim3 grid(mansy_ceil_div(p.N, 64), mansy_ceil_div(p.M, 64), 1); MANSY_GEMM_LAUNCH((gemm_bf16a_nn_kernel<64, 64, 3>), grid, block, st, p); }
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// TN: C[M, N] += A16^T B16 over K rows; A16 [K, M], B16 [K, N] bf16; k_per_split % 64 == 0; M, N >= 8 and multiples of 8.
int mansy_gemm_bf16a_tn(const GemmParams& p, int splits, int kgroups, hipStream_t st) {
  auto al16 = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  MANSY_REQUIRE(p.ep.a16 && p.ep.b16 && al16(p.ep.a16) && al16(p.ep.b16) && p.ep.a16_ld % 8 == 0 && p.ep.b16_ld % 8 == 0 && p.M % 8 == 0 && p.N % 8 == 0 &&
                    p.M >= 8 && p.N >= 8 && p.K % BK16 == 0 && p.k_per_split % BK16 == 0,
                "bf16-storage weight-gradient product: 16-byte aligned operands, M, N multiples of 8, K and the split length multiples of 64");
  dim3 grid(mansy_ceil_div(p.N, 128), mansy_ceil_div(p.M, 128), splits);
  if (kgroups == 2) { dim3 block(2 * NT); MANSY_GEMM_LAUNCH((gemm_bf16a_tn_kernel<2, 2>), grid, block, st, p); }
  else { dim3 block(NT); MANSY_GEMM_LAUNCH((gemm_bf16a_tn_kernel<2, 1>), grid, block, st, p); }
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
