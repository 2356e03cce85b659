// Feasibility lab (tuning aid, not product code) for the row-owning workgroup of DESIGN section 8: a chain of P dependent
// [4096, 512] x [512, 512] products (ReLU between them) computed by 256 workgroups that each OWN 16 rows for the whole chain.
// Activations never leave the workgroup (two [16][512] LDS images, ping-pong); every weight matrix streams from L2 straight
// into registers (float4 per lane = the B fragments of four v_mfma_f32_16x16x4_f32 k-steps, software-prefetched); no
// inter-workgroup synchronisation.  Compared with the product's 64x64 LDS-DMA tile kernel launched P times (tools/chain_lab.hip).
//   hipcc -O3 --offload-arch=gfx950 -o tools/_bin/rowown_lab tools/rowown_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef NWAVES
#define NWAVES 8
#endif
constexpr int D = 512, ROWS = 16, NW = NWAVES, NT = NW * 64, COLS_W = D / NW, TILES = COLS_W / 16, KB = 32, NKB = D / KB, LDA = D + 4, PF = 4;
constexpr int KBS = D * KB;   // floats per k-block of the packed weight image
static_assert(NKB % PF == 0, "the k loop is unrolled by the prefetch depth");

// W[512][512] (row = output column, PyTorch Linear layout) -> the fragment-order image own_product streams: every load instruction of a
// wave reads 1 KB contiguous, a k-block of the four waves 64 KB contiguous (all L2 channels), instead of sixteen 64-B pieces 2 KB apart
__global__ void pack_weight(const float* __restrict__ W, float* __restrict__ Wp) {
  const int idx = blockIdx.x * 256 + threadIdx.x;            // one float4 each
  if (idx >= D * D / 4) return;
  const int lane = idx & 63, half = (idx >> 6) & 1, t = (idx >> 7) % TILES, wave = (idx >> 7) / TILES % NW, kb = idx / (D * KB / 4);
  const int n = lane & 15, g = lane >> 4;
  *reinterpret_cast<float4*>(Wp + (long long)idx * 4) =
      *reinterpret_cast<const float4*>(W + (long long)(wave * COLS_W + t * 16 + n) * D + kb * KB + 8 * g + 4 * half);
}


// the chain for this workgroup's 16 rows.  The weight stream is continuous ACROSS products: block kb of product p refills its stage
// with block kb+PF, which is block kb+PF-NKB of product p+1 once kb+PF passes the end, so the pipeline never drains at the
// epilogue / barrier between products.  in / out are LDS images (row stride LDA).
__global__ __launch_bounds__(NT) void rowown_chain(const float* __restrict__ X, const float* __restrict__ W1, const float* __restrict__ W2, float* __restrict__ Y, int P, int kbs) {
  __shared__ __attribute__((aligned(16))) float act[2][ROWS * LDA];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row0 = blockIdx.x * ROWS;
  const int n = lane & 15, g = lane >> 4;
  const int woff = wave * TILES * 2 * 256 + lane * 4;          // packed image: [kb][wave][tile][half][lane][4]
  float4 bq[PF][TILES][2];
#pragma unroll
  for (int s = 0; s < PF - 1; ++s)
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
      bq[s][t][0] = *reinterpret_cast<const float4*>(W1 + woff + s * kbs + t * 512);
      bq[s][t][1] = *reinterpret_cast<const float4*>(W1 + woff + s * kbs + t * 512 + 256);
    }
  for (int i = tid; i < ROWS * (D / 4); i += NT) {
    const int r = i / (D / 4), c4 = i % (D / 4);
    *reinterpret_cast<float4*>(&act[0][r * LDA + c4 * 4]) = *reinterpret_cast<const float4*>(X + (long long)(row0 + r) * D + c4 * 4);
  }
  __syncthreads();
#pragma unroll 1
  for (int p = 0; p < P; ++p) {
    const float* Wc = ((p & 1) ? W2 : W1) + woff;
    const float* Wn = ((p & 1) ? W1 : W2) + woff;
    const bool last = p + 1 == P;
    const float* in = act[p & 1] + n * LDA + 8 * g;
    f32x4 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 a0 = *reinterpret_cast<const float4*>(in), a1 = *reinterpret_cast<const float4*>(in + 4);
#pragma unroll 1
    for (int kb0 = 0; kb0 < NKB; kb0 += PF) {
      const float* Wx = last ? Wc : Wn;                         // past the end of the chain the stream re-reads the last product's first blocks
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        const int kb = kb0 + s, j = kb + PF - 1;                // this stage consumes block kb and fetches block j into the stage consumed last
        const float* src = j < NKB ? Wc + j * kbs : Wx + (j - NKB) * kbs;
        float4 (&b)[TILES][2] = bq[s];
        float4 (&f)[TILES][2] = bq[(s + PF - 1) % PF];
        const float4 c0 = a0, c1 = a1;
        if (kb + 1 < NKB) {                                     // the next block's A fragment, under this block's MFMAs
          a0 = *reinterpret_cast<const float4*>(in + (kb + 1) * KB);
          a1 = *reinterpret_cast<const float4*>(in + (kb + 1) * KB + 4);
        }
        // (unconditional fetch: a branch around it makes the compiler's vmcnt bookkeeping assume the short path and wait for everything)
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
          f[t][0] = *reinterpret_cast<const float4*>(src + t * 512);
          f[t][1] = *reinterpret_cast<const float4*>(src + t * 512 + 256);
        }
        // k-step outermost: eight independent accumulators between two uses of the same one
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.x, b[t][0].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.y, b[t][0].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.z, b[t][0].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0.w, b[t][0].w, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.x, b[t][1].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.y, b[t][1].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.z, b[t][1].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1.w, b[t][1].w, acc[t], 0, 0, 0);
        // one fetch per four MFMAs: sixteen loads issued back to back leave the matrix pipe idle while the wave sits in VMEM issue
#pragma unroll
        for (int i = 0; i < TILES * 2; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
      }
    }
    float* out = act[(p + 1) & 1];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = fmaxf(acc[t][j], 0.f);
        const int m = 4 * g + j, col = wave * COLS_W + t * 16 + n;
        if (last) Y[(long long)(row0 + m) * D + col] = v;
        else out[m * LDA + col] = v;
      }
    __syncthreads();
  }
}

// reference: one [4096,512]x[512,512] product per launch, plain fp32 (correctness only)
__global__ void ref_product(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)M * D) return;
  const int m = (int)(idx / D), n = (int)(idx % D);
  float s = 0.f;
  for (int k = 0; k < D; ++k) s = fmaf(A[(long long)m * D + k], W[(long long)n * D + k], s);
  C[idx] = fmaxf(s, 0.f);
}
__global__ void fill_rand(float* p, long long n, unsigned seed, float scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13;
    p[i] = ((float)(x & 0xffff) / 65536.f - 0.5f) * scale;
  }
}
__global__ void maxdiff(const float* a, const float* b, long long n, float* out) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    atomicMax(reinterpret_cast<int*>(out), __float_as_int(fabsf(a[i] - b[i])));
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : 0;        // rowown_lab 8: just the chain of 8 (for rocprofv3 --pmc)
  const int M = 4096;
  float *X, *W1, *W2, *P1, *P2, *Y, *R0, *R1, *d_err;
  (void)hipMalloc(&X, (size_t)M * D * 4); (void)hipMalloc(&W1, (size_t)D * D * 4); (void)hipMalloc(&W2, (size_t)D * D * 4); (void)hipMalloc(&P1, (size_t)D * D * 4); (void)hipMalloc(&P2, (size_t)D * D * 4);
  (void)hipMalloc(&Y, (size_t)M * D * 4); (void)hipMalloc(&R0, (size_t)M * D * 4); (void)hipMalloc(&R1, (size_t)M * D * 4); (void)hipMalloc(&d_err, 4);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, X, (long long)M * D, 1u, 1.f);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, W1, (long long)D * D, 2u, 0.12f);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, W2, (long long)D * D, 3u, 0.12f);
  hipLaunchKernelGGL(pack_weight, dim3(D * D / 4 / 256), dim3(256), 0, 0, W1, P1);
  hipLaunchKernelGGL(pack_weight, dim3(D * D / 4 / 256), dim3(256), 0, 0, W2, P2);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int kbs : {KBS, 0})
  for (int P : {1, 2, 8}) {
    if (only && (P != only || !kbs)) continue;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rowown_chain, dim3(M / ROWS), dim3(NT), 0, 0, X, P1, P2, Y, P, kbs);
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(rowown_chain, dim3(M / ROWS), dim3(NT), 0, 0, X, P1, P2, Y, P, kbs);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
    // correctness against the naive chain
    const float* in = X;
    for (int p = 0; p < P; ++p) {
      float* out = (p & 1) ? R1 : R0;
      hipLaunchKernelGGL(ref_product, dim3((M * D + 255) / 256), dim3(256), 0, 0, in, (p & 1) ? W2 : W1, out, M);
      in = out;
    }
    (void)hipMemset(d_err, 0, 4);
    hipLaunchKernelGGL(maxdiff, dim3(1024), dim3(256), 0, 0, Y, in, (long long)M * D, d_err);
    float err; (void)hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost);
    const double us = ms / 20 * 1e3;
    printf("%s row-owning chain of %d products: %8.2f us (%6.2f us per product, %6.1f TF)  max |diff| vs naive %.2e\n", kbs ? "            " : "(L1-resident)", P, us, us / P,
           2.0 * M * D * D * P / us / 1e6, err);
  }
  return 0;
}
