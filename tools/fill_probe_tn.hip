// Where do the weight-gradient (TN) bf16-storage product's cycles go (round 6)?  The loop of csrc/gemm_bf16a.hip's gemm_bf16a_tn_kernel on [512, 512, 40960] with
// its parts switched on one at a time (1 = LDS-DMA fill, 2 = transposed fragment reads, 4 = MFMA, 8 = atomic epilogue).  Build on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/fill_probe_tn.hip -o /tmp/fill_probe_tn
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <type_traits>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NT = 256, BK16 = 64;
struct P { const unsigned short* a16; const unsigned short* b16; int lda, ldb, M, N, K, k_per_split, ldc; float* C; float* a_rowsum; };
__device__ __forceinline__ void glds16(unsigned voff, const void* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;
__device__ __forceinline__ bf16x4 tr_read(unsigned lds_byte_address) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(uintptr_t)lds_byte_address);
}

// byte offset of 16-byte chunk ch (0..15) of row k in a [64][128 x bf16] image with 256-byte rows (guide T10, image (b))
__device__ __forceinline__ unsigned tn_off(int k, int ch) { return (unsigned)(256 * k + 16 * (ch ^ (((k & 3) << 2) | ((k >> 2) & 3)))); }

template <int NS, int MODE>
__global__ __launch_bounds__(NT, NS <= 2 ? 2 : 1) void tn(P p) {
  constexpr int BM = 128, BN = 128, D = NS - 1;
  constexpr int A_BYTES = BK16 * 256, B_BYTES = BK16 * 256, STAGE_BYTES = A_BYTES + B_BYTES;      // 32 KB per stage
  constexpr int PA = 4, PB = 4, PPT = PA + PB;                                                     // 16 pieces of 1 KB per operand tile, 4 per wave
  __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE_BYTES];

  const int tid = threadIdx.x;
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
  const int nk = (k_end - k_begin) / BK16;
  const unsigned short* const A16 = p.a16;           // [K][M] (dY rows), lda
  const unsigned short* const B16 = p.b16;           // [K][N] (X rows), ldb
  const int lda = p.lda, ldb = p.ldb;

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
    const unsigned short* a_corner = ca + (long long)kt * BK16 * lda;
    const unsigned short* b_corner = cb + (long long)kt * BK16 * ldb;
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
  const bool want_rs = __builtin_amdgcn_readfirstlane((int)(p.a_rowsum != nullptr && tile_x == 0 && wn == 0)) != 0;   // wave-uniform, in an SGPR       // bias gradient: row sums of dY^T, taken once per row panel
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
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;
    if (D >= 3 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPT) : "memory");
    else if (D >= 2 && ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + D < nk) dma(cur == 0 ? NS - 1 : cur - 1, kt + D);
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
    if (MODE & 2) {
    frag(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) frag((s + 1) & 1, s + 1);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) { if (MODE & 4) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], bf[s & 1][j], acc[i][j], 0, 0, 0); else { acc[i][j][0] += (float)af[s & 1][i][0] + (float)bf[s & 1][j][0] + (float)af[s & 1][i][4] + (float)bf[s & 1][j][4]; } }
        if (RS) rs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s & 1][i], ones, rs[i], 0, 0, 0);
      }
    }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS reads retired before the barrier that frees this buffer
    cur = cur == NS - 1 ? 0 : cur + 1;
  }
  };
  if (want_rs) k_loop(std::true_type{}); else k_loop(std::false_type{});
  if (!(MODE & 8)) { float sum = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) sum += acc[i][j][e];
    if (sum == 12345.678f) p.C[tid] = sum;
    return; }
  // accumulate: C[m][n] += acc (fp32 atomics: the K splits and the steps' other products add into the same gradient)
  const int r = lane & 31, h = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
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
        if (row < p.M) atomicAdd(p.a_rowsum + row, rs[i][e]);
      }
    }
  }
}


template <int NS, int MODE>
void run(const char* name, std::vector<unsigned short*>& A, std::vector<unsigned short*>& B, P p, int splits) {
  p.k_per_split = p.K / splits;
  dim3 grid(p.N / 128, p.M / 128, splits), block(NT);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 6; ++i) { p.a16 = A[i % A.size()]; p.b16 = B[i % B.size()]; hipLaunchKernelGGL((tn<NS, MODE>), grid, block, 0, 0, p); }
  hipEventRecord(e0);
  const int it = 30;
  for (int i = 0; i < it; ++i) { p.a16 = A[i % A.size()]; p.b16 = B[i % B.size()]; hipLaunchKernelGGL((tn<NS, MODE>), grid, block, 0, 0, p); }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fill = (double)(p.M / 128) * (p.N / 128) * (p.K / 64) * 32768.0, hbm = (double)(p.M + p.N) * p.K * 2;
  printf("%-52s splits %2d  %7.1f us   LDS fill %5.2f TB/s   operands from HBM %5.2f TB/s\n", name, splits, ms / it * 1e3, fill / (ms / it * 1e-3) / 1e12, hbm / (ms / it * 1e-3) / 1e12);
}
int main() {
  P p; p.M = 512; p.N = 512; p.K = 40960; p.lda = p.M; p.ldb = p.N; p.ldc = p.N;
  std::vector<unsigned short*> A(6), B(6);
  for (auto& a : A) { hipMalloc(&a, (size_t)p.M * p.K * 2); hipMemset(a, 0x3c, (size_t)p.M * p.K * 2); }
  for (auto& a : B) { hipMalloc(&a, (size_t)p.N * p.K * 2); hipMemset(a, 0x3c, (size_t)p.N * p.K * 2); }
  hipMalloc(&p.C, (size_t)p.M * p.N * 4); hipMemset(p.C, 0, (size_t)p.M * p.N * 4); hipMalloc(&p.a_rowsum, p.M * 4); hipMemset(p.a_rowsum, 0, p.M * 4);
  for (int splits : {16, 32}) {
    run<2, 1>("DMA only", A, B, p, splits);
    run<2, 3>("DMA + transposed reads", A, B, p, splits);
    run<2, 7>("DMA + transposed reads + MFMA", A, B, p, splits);
    run<2, 15>("DMA + transposed reads + MFMA + atomics", A, B, p, splits);
  }
  std::vector<unsigned short*> A1(A.begin(), A.begin() + 1), B1(B.begin(), B.begin() + 1);
  run<2, 7>("same operands every launch (Infinity Cache): loop", A1, B1, p, 16);
  return 0;
}
