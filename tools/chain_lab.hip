// Feasibility lab for the persistent decoder recurrence (tuning aid, not product code): two DEPENDENT [4096,512]x[512,512]
// products (the decoder FFN: H = relu(X W1^T), Y = H W2^T) as ONE launch.  The 8 workgroups that own the 8 column tiles of a 64-row
// block form a team on one XCD; after its H tile a workgroup bumps the team's counter and waits (bounded spin) until all 8 tiles
// of its row block are in L2, then runs the second product on them.  Compared against the same two products as two launches.
//   hipcc -O3 --offload-arch=gfx950 -o tools/_bin/chain_lab tools/chain_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, BM = 64, BN = 64, AF = BM * BK, BF = BN * BK, STAGE = AF + BF, CLD = BN + 4;
constexpr int SMEM = 2 * STAGE > BM * CLD ? 2 * STAGE : BM * CLD;

__device__ __forceinline__ void glds16s(unsigned voff, const float* sbase, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16s_sc1(unsigned voff, const float* sbase, unsigned lds_dst) {    // agent-scope load: misses this CU's L1
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void tile_coords(int& tile_x, int& tile_y) {      // XCD x owns a contiguous range of tiles (8 row blocks)
  const int nwg = gridDim.x * gridDim.y, orig = blockIdx.y * gridDim.x + blockIdx.x;
  const int q = nwg >> 3, xcd = orig & 7, local = orig >> 3;
  const int t = xcd * q + local;
  tile_y = t / gridDim.x; tile_x = t - tile_y * gridDim.x;
}

// one 64x64 tile of C = act(A[M,K] * B[N,K]^T), LDS-DMA double-buffered loop of csrc/gemm_f32.hip
template <bool RELU, bool A_COHERENT = false>
__device__ __forceinline__ void tile_product(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int N, int K, int m0, int n0,
                                             float* smem, int tid) {
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
  unsigned vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int row = i * 32 + wave * 8 + (lane >> 3); vo[i] = (unsigned)(row * K + (((lane & 7) ^ ((row >> 1) & 7)) * 4)) * 4u; }
  const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)smem + wave * 1024u);
  const float* sa = A + (long long)m0 * K;
  const float* sb = B + (long long)n0 * K;
  const int nk = K / BK;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (A_COHERENT) glds16s_sc1(vo[i], sa, lds_wave + i * 4096u); else glds16s(vo[i], sa, lds_wave + i * 4096u);
    glds16s(vo[i], sb, lds_wave + AF * 4u + i * 4096u);
  }
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* a_l = smem + cur * STAGE;
    const float* b_l = a_l + AF;
    if (kt + 1 < nk) {
      const unsigned ldn = lds_wave + (cur ^ 1) * (STAGE * 4u);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (A_COHERENT) glds16s_sc1(vo[i], sa + (kt + 1) * BK, ldn + i * 4096u); else glds16s(vo[i], sa + (kt + 1) * BK, ldn + i * 4096u);
        glds16s(vo[i], sb + (kt + 1) * BK, ldn + AF * 4u + i * 4096u);
      }
    }
#pragma unroll
    for (int chunk = 0; chunk < 2; ++chunk) {
      const int ra = wm * 32 + r, rb = wn * 32 + r, c0 = h * 4 + chunk * 2;
      const float4 a0 = *reinterpret_cast<const float4*>(a_l + ra * BK + ((c0 + 0) ^ ((ra >> 1) & 7)) * 4);
      const float4 a1 = *reinterpret_cast<const float4*>(a_l + ra * BK + ((c0 + 1) ^ ((ra >> 1) & 7)) * 4);
      const float4 b0 = *reinterpret_cast<const float4*>(b_l + rb * BK + ((c0 + 0) ^ ((rb >> 1) & 7)) * 4);
      const float4 b1 = *reinterpret_cast<const float4*>(b_l + rb * BK + ((c0 + 1) ^ ((rb >> 1) & 7)) * 4);
      const float af[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, bf[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf[kk], acc, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float v = acc[e];
    smem[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * CLD + wn * 32 + r] = RELU ? fmaxf(v, 0.f) : v;
  }
  __syncthreads();
#pragma unroll 4
  for (int idx = tid; idx < BM * (BN / 4); idx += 256) {
    const int lr = idx / (BN / 4), c4 = idx % (BN / 4);
    *reinterpret_cast<float4*>(C + (long long)(m0 + lr) * N + n0 + c4 * 4) = *reinterpret_cast<const float4*>(smem + lr * CLD + c4 * 4);
  }
  __syncthreads();
}

template <bool RELU>
__global__ __launch_bounds__(256) void single(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int N, int K) {
  __shared__ __attribute__((aligned(1024))) float smem[SMEM];
  int tx, ty; tile_coords(tx, ty);
  tile_product<RELU>(A, B, C, N, K, ty * BM, tx * BN, smem, threadIdx.x);
}

// both products in one launch; cnt[row block] counts finished H tiles; `epoch` makes the counters reusable without a reset
__global__ __launch_bounds__(256) void chain2(const float* __restrict__ X, const float* __restrict__ W1, const float* __restrict__ W2, float* __restrict__ Hbuf,
                                              float* __restrict__ Y, unsigned* __restrict__ cnt, unsigned epoch, int N, int K, unsigned* __restrict__ timeouts) {
  __shared__ __attribute__((aligned(1024))) float smem[SMEM];
  int tx, ty; tile_coords(tx, ty);
  tile_product<true>(X, W1, Hbuf, N, K, ty * BM, tx * BN, smem, threadIdx.x);
  // the team lives on one XCD: its H tiles only have to reach that XCD's L2, which the (write-through) stores have done once
  // vmcnt drains -- an agent-scope release fence would write the whole L2 back (buffer_wbl2), tens of microseconds
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(cnt + ty, 1u);
    const unsigned want = epoch * gridDim.x;
    int spins = 0;
    while (__hip_atomic_load(cnt + ty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 20)) { atomicAdd(timeouts, 1u); break; }      // bounded: a lost team-mate cannot hang the GPU
    }
  }
  __syncthreads();
  tile_product<false, true>(Hbuf, W2, Y, N, N, ty * BM, tx * BN, smem, threadIdx.x);      // H is read with sc1 loads: straight from L2
}

// P dependent products in one launch (activations ping-pong between two buffers), optional start stagger of the odd row blocks:
// do teams that run out of phase fill each other's pipeline bubbles?
__global__ __launch_bounds__(256) void chainP(const float* __restrict__ X, const float* __restrict__ W1, const float* __restrict__ W2, float* __restrict__ buf0,
                                              float* __restrict__ buf1, unsigned* __restrict__ cnt, unsigned base, int P, int N, int stagger_sleeps,
                                              unsigned* __restrict__ timeouts) {
  __shared__ __attribute__((aligned(1024))) float smem[SMEM];
  int tx, ty; tile_coords(tx, ty);
  if (stagger_sleeps > 0 && (ty & 1)) for (int i = 0; i < stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
  for (int p = 0; p < P; ++p) {
    const float* in = p == 0 ? X : ((p & 1) ? buf0 : buf1);
    float* out = (p & 1) ? buf1 : buf0;
    const float* W = (p & 1) ? W2 : W1;
    if (p == 0) tile_product<true, false>(in, W, out, N, N, ty * BM, tx * BN, smem, threadIdx.x);
    else tile_product<true, true>(in, W, out, N, N, ty * BM, tx * BN, smem, threadIdx.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(cnt + ty, 1u);
      const unsigned want = (base + p + 1) * gridDim.x;
      int spins = 0;
      while (__hip_atomic_load(cnt + ty, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 20)) { atomicAdd(timeouts, 1u); break; }
      }
    }
    __syncthreads();
  }
}

__global__ void fill_rand(float* p, long long n, unsigned seed, float scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13;
    p[i] = ((float)(x & 0xffff) / 65536.f - 0.5f) * scale;
  }
}
__global__ void diff(const float* a, const float* b, long long n, unsigned* bad) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    if (a[i] != b[i]) atomicAdd(bad, 1u);
}

int main() {
  const int M = 4096, N = 512, K = 512;
  float *X, *W1, *W2, *H, *Y, *H2, *Y2; unsigned *cnt, *flags;
  (void)hipMalloc(&X, (size_t)M * K * 4); (void)hipMalloc(&W1, (size_t)N * K * 4); (void)hipMalloc(&W2, (size_t)N * N * 4);
  (void)hipMalloc(&H, (size_t)M * N * 4); (void)hipMalloc(&Y, (size_t)M * N * 4); (void)hipMalloc(&H2, (size_t)M * N * 4); (void)hipMalloc(&Y2, (size_t)M * N * 4);
  (void)hipMalloc(&cnt, 64 * 4); (void)hipMalloc(&flags, 8); (void)hipMemset(cnt, 0, 64 * 4); (void)hipMemset(flags, 0, 8);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, X, (long long)M * K, 1u, 1.f);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, W1, (long long)N * K, 2u, 0.1f);
  hipLaunchKernelGGL(fill_rand, dim3(1024), dim3(256), 0, 0, W2, (long long)N * N, 3u, 0.1f);
  const dim3 grid(N / BN, M / BM);
  unsigned epoch = 0;
  auto two = [&] {
    hipLaunchKernelGGL(single<true>, grid, dim3(256), 0, 0, X, W1, H2, N, K);
    hipLaunchKernelGGL(single<false>, grid, dim3(256), 0, 0, H2, W2, Y2, N, N);
  };
  auto one = [&] { ++epoch; hipLaunchKernelGGL(chain2, grid, dim3(256), 0, 0, X, W1, W2, H, Y, cnt, epoch, N, K, flags); };
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    for (int i = 0; i < 5; ++i) { two(); one(); }
    (void)hipDeviceSynchronize();
    float ms;
    (void)hipEventRecord(e0); for (int i = 0; i < 50; ++i) two(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); printf("two launches : %7.2f us per pair\n", ms / 50 * 1e3);
    (void)hipEventRecord(e0); for (int i = 0; i < 50; ++i) one(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); printf("one launch   : %7.2f us per pair\n", ms / 50 * 1e3);
  }
  {   // a chain of P = 8 dependent products: 8 launches vs one launch, without and with a start stagger of the odd row blocks
    const int P = 8;
    unsigned *cnt2; (void)hipMalloc(&cnt2, 64 * 4); (void)hipMemset(cnt2, 0, 64 * 4);
    unsigned base = 0;
    auto sep = [&] {
      for (int p = 0; p < P; ++p) {
        const float* in = p == 0 ? X : ((p & 1) ? H2 : Y2);
        hipLaunchKernelGGL(single<true>, grid, dim3(256), 0, 0, in, (p & 1) ? W2 : W1, (p & 1) ? Y2 : H2, N, N);
      }
    };
    for (int st : {0, 1, 2, 4}) {          // one s_sleep(127) = 8 128 cycles = 3.4 us
      auto fused = [&] { hipLaunchKernelGGL(chainP, grid, dim3(256), 0, 0, X, W1, W2, H, Y, cnt2, base, P, N, st, flags); base += P; };
      for (int i = 0; i < 3; ++i) { sep(); fused(); }
      (void)hipDeviceSynchronize();
      float ms;
      (void)hipEventRecord(e0); for (int i = 0; i < 20; ++i) sep(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1); const float t_sep = ms / 20 * 1e3;
      (void)hipEventRecord(e0); for (int i = 0; i < 20; ++i) fused(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
      printf("chain of %d: %d launches %7.2f us | one launch, stagger %3d sleeps: %7.2f us\n", P, P, t_sep, st, ms / 20 * 1e3);
    }
    unsigned *bad; (void)hipMalloc(&bad, 4); (void)hipMemset(bad, 0, 4);
    hipLaunchKernelGGL(diff, dim3(1024), dim3(256), 0, 0, Y, Y2, (long long)M * N, bad);
    unsigned b; (void)hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
    printf("chain result differs from the launch-by-launch result in %u elements\n", b);
    two(); one();                 // restore H2 / Y2 / H / Y of the two-product comparison below
  }
  hipLaunchKernelGGL(diff, dim3(1024), dim3(256), 0, 0, Y, Y2, (long long)M * N, flags + 1);
  unsigned f[2]; (void)hipMemcpy(f, flags, 8, hipMemcpyDeviceToHost);
  printf("spin timeouts %u, elements differing from the two-launch result %u of %d\n", f[0], f[1], M * N);
  return 0;
}
