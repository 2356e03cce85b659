// GEMM structure lab (tuning aid, not product code): NT fp32-MFMA main-loop variants, to see where the gap to the
// 155 TFLOP/s sustained MFMA rate goes.  Shapes must be tile multiples.  Operands are random (zeros read high).
//   VAR 0: register staging, padded LDS rows, one barrier per K-tile (the csrc/gemm_f32.hip structure)
//   VAR 1: VAR 0 without global loads / LDS writes (LDS read + MFMA + barrier only)
//   VAR 2: LDS-DMA (global_load_lds_dwordx4), XOR-swizzled 128-B rows, 2 buffers, vmcnt(0)+barrier per K-tile
//   VAR 3: LDS-DMA, NBUF buffers, counted vmcnt + raw s_barrier: loads stay in flight across the barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, NT = 256;

__device__ __forceinline__ void tile_coords(int& tile_x, int& tile_y) {
  const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
  const int q = nwg >> 3, rr = nwg & 7, xcd = orig & 7, local = orig >> 3;
  const int t = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + local;
  tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
}

template <int BM, int BN>
__device__ __forceinline__ void store_c(float* smem, f32x16 (&acc)[BM / 64][BN / 64], float* __restrict__ C, int N, int m0, int n0, int tid) {
  constexpr int TM = BM / 64, TN = BN / 64, CLD = BN + 4, C4 = BN / 4;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) smem[(wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wn * (BN / 2) + j * 32 + r] = acc[i][j][e];
  __syncthreads();
#pragma unroll 4
  for (int idx = tid; idx < BM * C4; idx += NT) {
    const int lr = idx / C4, c4 = idx % C4;
    *reinterpret_cast<float4*>(C + (long long)(m0 + lr) * N + n0 + c4 * 4) = *reinterpret_cast<const float4*>(smem + lr * CLD + c4 * 4);
  }
}

// ---------------------------------------------------------------- VAR 0 / 1: register staging
template <int R>
__device__ __forceinline__ void rload(const float* __restrict__ P, int ld, int row0, int k0, float4 (&reg)[R * BK / 4 / NT], int tid) {
#pragma unroll
  for (int i = 0; i < R * BK / 4 / NT; ++i) {
    const int idx = tid + i * NT;
    reg[i] = *reinterpret_cast<const float4*>(P + (long long)(row0 + (idx >> 3)) * ld + k0 + (idx & 7) * 4);
  }
}
template <int R>
__device__ __forceinline__ void rstore(float* __restrict__ lds, const float4 (&reg)[R * BK / 4 / NT], int tid) {
#pragma unroll
  for (int i = 0; i < R * BK / 4 / NT; ++i) {
    const int idx = tid + i * NT;
    *reinterpret_cast<float4*>(lds + (idx >> 3) * (BK + 4) + (idx & 7) * 4) = reg[i];
  }
}

template <int BM, int BN, int VAR>
__global__ __launch_bounds__(NT) void lab_reg(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = BM / 64, TN = BN / 64, LD = BK + 4;
  constexpr int AF = BM * LD, BF = BN * LD, STAGE = AF + BF;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int tile_x, tile_y; tile_coords(tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN, nk = K / BK;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float4 ra[BM * BK / 4 / NT], rb[BN * BK / 4 / NT];
  rload<BM>(A, K, m0, 0, ra, tid); rload<BN>(B, K, n0, 0, rb, tid);
  rstore<BM>(smem, ra, tid); rstore<BN>(smem + AF, rb, tid);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (VAR == 0 && kt + 1 < nk) { rload<BM>(A, K, m0, (kt + 1) * BK, ra, tid); rload<BN>(B, K, n0, (kt + 1) * BK, rb, tid); }
    const float* a_l = smem + cur * STAGE;
    const float* b_l = a_l + AF;
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      float af[TM][8], bf[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4* p = reinterpret_cast<const float4*>(a_l + (wm * (BM / 2) + i * 32 + r) * LD + h * 16 + chunk * 8);
        const float4 v0 = p[0], v1 = p[1];
        af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w; af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float4* p = reinterpret_cast<const float4*>(b_l + (wn * (BN / 2) + j * 32 + r) * LD + h * 16 + chunk * 8);
        const float4 v0 = p[0], v1 = p[1];
        bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w; bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    }
    if (VAR == 0 && kt + 1 < nk) { rstore<BM>(smem + (cur ^ 1) * STAGE, ra, tid); rstore<BN>(smem + (cur ^ 1) * STAGE + AF, rb, tid); }
    __syncthreads();
  }
  store_c<BM, BN>(smem, acc, C, N, m0, n0, tid);
}

// ---------------------------------------------------------------- VAR 2 / 3: LDS-DMA staging
// LDS image of an operand tile: [R rows][8 chunks of 16 B], unpadded (one wave-instruction fills 8 rows x 128 B,
// lane-linear).  Slot (row, c') holds the global k-chunk c = c' ^ (row & 7): the swizzle is applied on the SOURCE address.
__device__ __forceinline__ void glds16(const float* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// Generalised: WM x WN waves, K-tile BKT floats (32 or 64), NBUF LDS buffers, PRIO: s_setprio(1) around the MFMA cluster.
// LDS row = BKT floats (128 or 256 B); 16-B chunk c of row `row` sits in slot c ^ (row & 7).
template <int R, int BKT, int NW>
__device__ __forceinline__ void dma_tile(const float* __restrict__ P, int ld, int row0, int k0, float* lds, int tid) {
  constexpr int CPR = BKT / 4;            // 16-B chunks per row
  constexpr int RPI = 64 / CPR;           // rows per wave-instruction
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int i = 0; i < R / (RPI * NW); ++i) {
    const int rbase = (i * NW + wave) * RPI;
    const int row = rbase + lane / CPR;
    const int c = (lane % CPR) ^ (row & 7);
    const float* src = P + (long long)(row0 + row) * ld + k0 + c * 4;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds + rbase * BKT));
    glds16(src, dst);
  }
}

template <int BM, int BN, int BKT, int NBUF, int WM, int WN, int PRIO>
__global__ __launch_bounds__(64 * WM * WN) void lab_dma(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int NW = WM * WN, NTH = 64 * NW;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int AF = BM * BKT, BF = BN * BKT, STAGE = AF + BF;
  constexpr int CF = BM * (BN + 4);
  constexpr int SM = NBUF * STAGE > CF ? NBUF * STAGE : CF;
  __shared__ __attribute__((aligned(1024))) float smem[SM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  int tile_x, tile_y; tile_coords(tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN, nk = K / BKT;
  constexpr int PER_TILE = (BM + BN) / ((256 / BKT) * NW);     // DMA instructions per wave per K-tile
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s) {
    if (s < nk) { dma_tile<BM, BKT, NW>(A, K, m0, s * BKT, smem + s * STAGE, tid); dma_tile<BN, BKT, NW>(B, K, n0, s * BKT, smem + s * STAGE + AF, tid); }
  }
  for (int kt = 0; kt < nk; ++kt) {
    if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else { if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory");
    if (kt + NBUF - 1 < nk) {
      const int s = (kt + NBUF - 1) % NBUF;
      dma_tile<BM, BKT, NW>(A, K, m0, (kt + NBUF - 1) * BKT, smem + s * STAGE, tid);
      dma_tile<BN, BKT, NW>(B, K, n0, (kt + NBUF - 1) * BKT, smem + s * STAGE + AF, tid);
    }
    const float* a_l = smem + (kt % NBUF) * STAGE;
    const float* b_l = a_l + AF;
    // lane half h takes k in [h*BKT/2, (h+1)*BKT/2) of the K-tile: chunks of 8 k = 2 x 16 B
#pragma unroll
    for (int chunk = 0; chunk < BKT / 16; ++chunk) {
      float af[TM][8], bf[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / WM) + i * 32 + r;
        const int c0 = h * (BKT / 8) + chunk * 2;
        const float4 v0 = *reinterpret_cast<const float4*>(a_l + row * BKT + ((c0 + 0) ^ (row & 7)) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(a_l + row * BKT + ((c0 + 1) ^ (row & 7)) * 4);
        af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w; af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / WN) + j * 32 + r;
        const int c0 = h * (BKT / 8) + chunk * 2;
        const float4 v0 = *reinterpret_cast<const float4*>(b_l + row * BKT + ((c0 + 0) ^ (row & 7)) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(b_l + row * BKT + ((c0 + 1) ^ (row & 7)) * 4);
        bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w; bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
      }
      if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
      if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // epilogue (generic wave grid)
  constexpr int CLD = BN + 4, C4 = BN / 4;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) smem[(wm * (BM / WM) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wn * (BN / WN) + j * 32 + r] = acc[i][j][e];
  __syncthreads();
#pragma unroll 4
  for (int idx = tid; idx < BM * C4; idx += NTH) {
    const int lr = idx / C4, c4 = idx % C4;
    *reinterpret_cast<float4*>(C + (long long)(m0 + lr) * N + n0 + c4 * 4) = *reinterpret_cast<const float4*>(smem + lr * CLD + c4 * 4);
  }
}

// ---------------------------------------------------------------- lab_dma2: cheap DMA issue (SGPR base + loop-invariant VGPR offsets)
// ORDER 0: DMA block, then fragment reads + MFMA.  1: chunk-0 fragment reads, DMA block, MFMA.  2: DMA pieces interleaved
// between the MFMA groups of chunk 0 (sched_barrier pins the order).
__device__ __forceinline__ void glds16s(unsigned voff, const float* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
template <int BM, int BN, int ORDER>
__global__ __launch_bounds__(256) void lab_dma2(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = BM / 64, TN = BN / 64, PA = BM / 32, PB = BN / 32;
  constexpr int AF = BM * BK, BF = BN * BK, STAGE = AF + BF;
  constexpr int CF = BM * (BN + 4);
  constexpr int SM = 2 * STAGE > CF ? 2 * STAGE : CF;
  __shared__ __attribute__((aligned(1024))) float smem[SM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int tile_x, tile_y; tile_coords(tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN, nk = K / BK;
  // loop-invariant per-lane byte offsets of the DMA pieces, relative to the tile's (row0, k0) corner
  unsigned voa[PA], vob[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) { const int row = i * 32 + wave * 8 + (lane >> 3); voa[i] = (unsigned)(row * K + (((lane & 7) ^ (row & 7)) * 4)) * 4u; }
#pragma unroll
  for (int i = 0; i < PB; ++i) { const int row = i * 32 + wave * 8 + (lane >> 3); vob[i] = (unsigned)(row * K + (((lane & 7) ^ (row & 7)) * 4)) * 4u; }
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + wave * 1024u);
  const float* sa = A + (long long)m0 * K;
  const float* sb = B + (long long)n0 * K;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
  for (int i = 0; i < PA; ++i) glds16s(voa[i], sa, lds_wave + i * 4096u);
#pragma unroll
  for (int i = 0; i < PB; ++i) glds16s(vob[i], sb, lds_wave + AF * 4u + i * 4096u);
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* a_l = smem + cur * STAGE;
    const float* b_l = a_l + AF;
    const bool more = kt + 1 < nk;
    const float* san = sa + (kt + 1) * BK;
    const float* sbn = sb + (kt + 1) * BK;
    const unsigned ldn = lds_wave + (cur ^ 1) * (STAGE * 4u);
    float af[2][TM][8], bf[2][TN][8];
    auto read_chunk = [&](int chunk, float (&fa)[TM][8], float (&fb)[TN][8]) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / 2) + i * 32 + r, c0 = h * 4 + chunk * 2;
        const float4 v0 = *reinterpret_cast<const float4*>(a_l + row * BK + ((c0 + 0) ^ (row & 7)) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(a_l + row * BK + ((c0 + 1) ^ (row & 7)) * 4);
        fa[i][0] = v0.x; fa[i][1] = v0.y; fa[i][2] = v0.z; fa[i][3] = v0.w; fa[i][4] = v1.x; fa[i][5] = v1.y; fa[i][6] = v1.z; fa[i][7] = v1.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / 2) + j * 32 + r, c0 = h * 4 + chunk * 2;
        const float4 v0 = *reinterpret_cast<const float4*>(b_l + row * BK + ((c0 + 0) ^ (row & 7)) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(b_l + row * BK + ((c0 + 1) ^ (row & 7)) * 4);
        fb[j][0] = v0.x; fb[j][1] = v0.y; fb[j][2] = v0.z; fb[j][3] = v0.w; fb[j][4] = v1.x; fb[j][5] = v1.y; fb[j][6] = v1.z; fb[j][7] = v1.w;
      }
    };
    if (ORDER == 0) {
      if (more) {
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16s(voa[i], san, ldn + i * 4096u);
#pragma unroll
        for (int i = 0; i < PB; ++i) glds16s(vob[i], sbn, ldn + AF * 4u + i * 4096u);
      }
      read_chunk(0, af[0], bf[0]);
    } else {
      read_chunk(0, af[0], bf[0]);
      if (ORDER == 1 && more) {
#pragma unroll
        for (int i = 0; i < PA; ++i) glds16s(voa[i], san, ldn + i * 4096u);
#pragma unroll
        for (int i = 0; i < PB; ++i) glds16s(vob[i], sbn, ldn + AF * 4u + i * 4096u);
      }
    }
    if (ORDER == 2) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][i][kk], bf[0][j][kk], acc[i][j], 0, 0, 0);
      if (ORDER == 2) {
        if (more) {
          if (kk < PA) glds16s(voa[kk < PA ? kk : 0], san, ldn + kk * 4096u);
          else if (kk - PA < PB) glds16s(vob[kk - PA < PB ? kk - PA : 0], sbn, ldn + AF * 4u + (kk - PA) * 4096u);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    read_chunk(1, af[1], bf[1]);
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][i][kk], bf[1][j][kk], acc[i][j], 0, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  constexpr int CLD = BN + 4, C4 = BN / 4;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) smem[(wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wn * (BN / 2) + j * 32 + r] = acc[i][j][e];
  __syncthreads();
#pragma unroll 4
  for (int idx = tid; idx < BM * C4; idx += 256) {
    const int lr = idx / C4, c4 = idx % C4;
    *reinterpret_cast<float4*>(C + (long long)(m0 + lr) * N + n0 + c4 * 4) = *reinterpret_cast<const float4*>(smem + lr * CLD + c4 * 4);
  }
}

// lab_dma3: lab_dma2 (ORDER 0) with the K-tile depth BKT (32 / 64) and the number of LDS buffers NBUF (2 / 3) as parameters.
// Row image = BKT floats; 16-B chunk c of a row in slot c ^ (row & 7) (low 3 bits).
template <int BM, int BN, int BKT, int NBUF>
__global__ __launch_bounds__(256) void lab_dma3(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int M, int N, int K) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int CPR = BKT / 4, RPI = 64 / CPR;            // chunks per row, rows per DMA piece
  constexpr int PA = BM / (RPI * 4), PB = BN / (RPI * 4);  // pieces per wave
  constexpr int AF = BM * BKT, BF = BN * BKT, STAGE = AF + BF;
  constexpr int CF = BM * (BN + 4);
  constexpr int SM = NBUF * STAGE > CF ? NBUF * STAGE : CF;
  __shared__ __attribute__((aligned(1024))) float smem[SM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  int tile_x, tile_y; tile_coords(tile_x, tile_y);
  const int m0 = tile_y * BM, n0 = tile_x * BN, nk = K / BKT;
  unsigned voa[PA], vob[PB];
#pragma unroll
  for (int i = 0; i < PA; ++i) { const int row = (i * 4 + wave) * RPI + lane / CPR; voa[i] = (unsigned)(row * K + (((lane % CPR) ^ (row & 7)) * 4)) * 4u; }
#pragma unroll
  for (int i = 0; i < PB; ++i) { const int row = (i * 4 + wave) * RPI + lane / CPR; vob[i] = (unsigned)(row * K + (((lane % CPR) ^ (row & 7)) * 4)) * 4u; }
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + wave * 1024u);
  const float* sa = A + (long long)m0 * K;
  const float* sb = B + (long long)n0 * K;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  auto issue = [&](int kt) {
    const float* san = sa + kt * BKT;
    const float* sbn = sb + kt * BKT;
    const unsigned ldn = lds_wave + (kt % NBUF) * (STAGE * 4u);
#pragma unroll
    for (int i = 0; i < PA; ++i) glds16s(voa[i], san, ldn + i * 4096u);
#pragma unroll
    for (int i = 0; i < PB; ++i) glds16s(vob[i], sbn, ldn + AF * 4u + i * 4096u);
  };
#pragma unroll
  for (int s = 0; s < NBUF - 1; ++s) if (s < nk) issue(s);
  for (int kt = 0; kt < nk; ++kt) {
    if (NBUF == 2 || kt + 1 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NBUF - 1 < nk) issue(kt + NBUF - 1);
    const float* a_l = smem + (kt % NBUF) * STAGE;
    const float* b_l = a_l + AF;
#pragma unroll
    for (int chunk = 0; chunk < BKT / 16; ++chunk) {
      float af[TM][8], bf[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wm * (BM / 2) + i * 32 + r, c0 = h * (BKT / 8) + chunk * 2;
        const float4 v0 = *reinterpret_cast<const float4*>(a_l + row * BKT + ((c0 + 0) ^ (row & 7)) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(a_l + row * BKT + ((c0 + 1) ^ (row & 7)) * 4);
        af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w; af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wn * (BN / 2) + j * 32 + r, c0 = h * (BKT / 8) + chunk * 2;
        const float4 v0 = *reinterpret_cast<const float4*>(b_l + row * BKT + ((c0 + 0) ^ (row & 7)) * 4);
        const float4 v1 = *reinterpret_cast<const float4*>(b_l + row * BKT + ((c0 + 1) ^ (row & 7)) * 4);
        bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w; bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
      }
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  constexpr int CLD = BN + 4, C4 = BN / 4;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) smem[(wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wn * (BN / 2) + j * 32 + r] = acc[i][j][e];
  __syncthreads();
#pragma unroll 4
  for (int idx = tid; idx < BM * C4; idx += 256) {
    const int lr = idx / C4, c4 = idx % C4;
    *reinterpret_cast<float4*>(C + (long long)(m0 + lr) * N + n0 + c4 * 4) = *reinterpret_cast<const float4*>(smem + lr * CLD + c4 * 4);
  }
}

__global__ void fill_rand(float* p, long long n, unsigned seed) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13;
    p[i] = (float)(x & 0xffff) / 65536.f - 0.5f;
  }
}
__global__ void ref_check(const float* A, const float* B, const float* C, int M, int N, int K, int samples, float* maxerr) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= samples) return;
  const int m = (int)(((long long)s * 7919) % M), n = (int)(((long long)s * 104729) % N);
  double acc = 0.0;
  for (int k = 0; k < K; ++k) acc += (double)A[(long long)m * K + k] * (double)B[(long long)n * K + k];
  const float e = fabsf((float)acc - C[(long long)m * N + n]);
  atomicMax(reinterpret_cast<int*>(maxerr), __float_as_int(e));
}

template <typename F>
void bench(const char* name, int BM, int BN, const float* A, const float* B, float* C, int M, int N, int K, F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipMemset(C, 0, (size_t)M * N * 4);
  for (int i = 0; i < 3; ++i) launch();
  float* d_err; (void)hipMalloc(&d_err, 4); (void)hipMemset(d_err, 0, 4);
  hipLaunchKernelGGL(ref_check, dim3(16), dim3(256), 0, 0, A, B, C, M, N, K, 4096, d_err);
  float err; (void)hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost); (void)hipFree(d_err);
  (void)hipEventRecord(e0);
  const int n = 20;
  for (int i = 0; i < n; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double us = ms / n * 1e3;
  printf("%-28s tile %3dx%-3d M=%6d N=%5d K=%5d: %8.1f us %6.1f TF  maxerr %.2e\n", name, BM, BN, M, N, K, us, 2.0 * M * N * K / us / 1e6, err);
  fflush(stdout);
}

#define LAUNCH(KERN, BM, BN, NTH) [&] { hipLaunchKernelGGL((KERN), dim3(N / BN, M / BM), dim3(NTH), 0, 0, A, B, C, M, N, K); }
#define DMA(BM, BN, BKT, NBUF, WM, WN, PRIO) \
  bench("dma " #BM "x" #BN " bk" #BKT " nbuf" #NBUF " w" #WM "x" #WN " prio" #PRIO, BM, BN, A, B, C, M, N, K, LAUNCH((lab_dma<BM, BN, BKT, NBUF, WM, WN, PRIO>), BM, BN, 64 * WM * WN))

#define DMA3(BM, BN, BKT, NBUF) bench("dma3 bk" #BKT " nbuf" #NBUF, BM, BN, A, B, C, M, N, K, LAUNCH((lab_dma3<BM, BN, BKT, NBUF>), BM, BN, 256))
void suite(const float* A, const float* B, float* C, int M, int N, int K) {
  DMA3(64, 64, 32, 2); DMA3(64, 64, 64, 2); DMA3(64, 64, 32, 3); DMA3(64, 64, 64, 3);
  DMA3(128, 64, 32, 2); DMA3(128, 64, 64, 2); DMA3(128, 64, 32, 3);
}

int main() {
  float *A, *B, *C;
  const long long na = 40960ll * 4096, nb = 1536ll * 4096;
  (void)hipMalloc(&A, na * 4); (void)hipMalloc(&B, nb * 4); (void)hipMalloc(&C, 40960ll * 1536 * 4);
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, A, na, 1u);
  hipLaunchKernelGGL(fill_rand, dim3(4096), dim3(256), 0, 0, B, nb, 2u);
  (void)hipDeviceSynchronize();
  const int shapes[][3] = {{4096, 512, 512}, {4096, 512, 1536}, {4096, 1536, 512}, {40960, 512, 512}, {40960, 1536, 512}};
  for (auto& sh : shapes) { suite(A, B, C, sh[0], sh[1], sh[2]); printf("\n"); }
  return 0;
}
