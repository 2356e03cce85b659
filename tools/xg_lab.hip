// Lab (tuning aid, not product code): where do the ~28 us of one peer-memory gradient average go at world 1?  The collective kernel of csrc/xgmi.hip in
// pieces, 1.7 MB (427 072 floats), each variant launched back to back 300 times between two events, alone and followed by a "neighbour" kernel that
// re-reads 8 MB of ordinary device memory it has read before (what the cycle's next launches do with their weights: a cache invalidate shows up THERE).
//   v0  copy into the slot + per-workgroup system release + ticket + flag + every-wave system acquire + reduce   (round-4 kernel)
//   v1  no copy                                                                                                   (round-5 slot form, first cut)
//   v2  no copy, ONE lane stores the flag (no fence: the gradient was written by EARLIER launches), one wave acquires
//   v3  v2 with agent-scope acquire instead of system scope
//   v4  v2 without any acquire (what a reduce of already-visible memory costs)
//   v5  v4 on ordinary (coarse-grained) memory instead of the fine-grained slot
//   grid: 64 / 128 / 256 workgroups
//   hipcc -O3 --offload-arch=gfx950 -o tools/_bin/xg_lab tools/xg_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int V>
__global__ __launch_bounds__(256) void xg(float* __restrict__ g, long long n4, float* __restrict__ slot, unsigned* flag, unsigned* counter, unsigned target, unsigned epoch,
                                          double* __restrict__ parts) {
  const long long gtid = (long long)blockIdx.x * 256 + threadIdx.x, gsize = (long long)gridDim.x * 256;
  if (V == 0) {
    for (long long i = gtid; i < n4; i += gsize) reinterpret_cast<float4*>(slot)[i] = reinterpret_cast<const float4*>(g)[i];
  }
  if (V <= 1) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __atomic_thread_fence(__ATOMIC_RELEASE);
      const unsigned prev = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (prev + 1u == target) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  } else {
    if (blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (V == 2 || V == 3) {
      if (threadIdx.x < 64) {
        if (V == 2) __atomic_thread_fence(__ATOMIC_ACQUIRE); else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
  }
  double sq = 0.0;
  for (long long i0 = gtid; i0 < n4; i0 += gsize * 4) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const long long i = i0 + u * gsize; v[u] = reinterpret_cast<const float4*>(slot)[i < n4 ? i : n4 - 1]; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long long i = i0 + u * gsize;
      if (i >= n4) continue;
      reinterpret_cast<float4*>(g)[i] = v[u];
      sq += ((double)v[u].x * v[u].x + (double)v[u].y * v[u].y) + ((double)v[u].z * v[u].z + (double)v[u].w * v[u].w);
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = sq;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) parts[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void neighbour(const float4* __restrict__ w, long long n4, float* out) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) { const float4 v = w[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.678f) out[0] = s;
}
// what produces the gradient in the cycle: a read-add-write pass over the buffer (the accumulate epilogues) -- on the slot vs on ordinary memory
__global__ __launch_bounds__(256) void producer(float4* __restrict__ g, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) { float4 v = g[i]; v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f; g[i] = v; }
}

template <int V>
static void run(const char* name, int blocks, float* g, float* slot, long long n4, unsigned* flag, unsigned* counter, double* parts, const float4* w, float* out, bool with_nb, bool with_prod) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 300;
  CK(hipMemset(counter, 0, 4));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(counter, 0, 4));
    CK(hipEventRecord(e0));
    for (int it = 0; it < iters; ++it) {
      if (with_prod) hipLaunchKernelGGL(producer, dim3(512), dim3(256), 0, 0, (float4*)slot, n4);
      hipLaunchKernelGGL(xg<V>, dim3(blocks), dim3(256), 0, 0, g, n4, slot, flag, counter, (unsigned)((it + 1) * blocks), (unsigned)(it + 1), parts);
      if (with_nb) hipLaunchKernelGGL(neighbour, dim3(512), dim3(256), 0, 0, w, (long long)(8 << 20) / 16, out);
    }
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s blocks %3d%s%s: %.2f us per iteration\n", name, blocks, with_prod ? " +producer" : "", with_nb ? " +neighbour" : "", ms * 1e3 / iters);
}

int main() {
  const long long n = 427072, n4 = n / 4;
  float *g, *slot_fg, *slot_cg, *out; float4* w; unsigned *flag, *counter; double* parts;
  CK(hipMalloc(&g, n * 4)); CK(hipMalloc(&slot_cg, n * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&w, 8 << 20)); CK(hipMalloc(&counter, 256)); CK(hipMalloc(&parts, 8 * 256));
  CK(hipExtMallocWithFlags((void**)&slot_fg, n * 4 + 256, hipDeviceMallocFinegrained));
  flag = reinterpret_cast<unsigned*>(slot_fg + n);
  CK(hipMemset(g, 0, n * 4)); CK(hipMemset(slot_fg, 0, n * 4 + 256)); CK(hipMemset(slot_cg, 0, n * 4)); CK(hipMemset(w, 0, 8 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // baselines: the neighbour and the producers alone
  for (int which = 0; which < 3; ++which) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 300; ++it) {
        if (which == 0) hipLaunchKernelGGL(neighbour, dim3(512), dim3(256), 0, 0, w, (long long)(8 << 20) / 16, out);
        if (which == 1) hipLaunchKernelGGL(producer, dim3(512), dim3(256), 0, 0, (float4*)slot_fg, n4);
        if (which == 2) hipLaunchKernelGGL(producer, dim3(512), dim3(256), 0, 0, (float4*)slot_cg, n4);
      }
      CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%s alone: %.2f us per launch\n", which == 0 ? "neighbour (8 MB re-read)" : which == 1 ? "producer on the fine-grained slot" : "producer on ordinary memory", ms * 1e3 / 300);
  }
  for (int nb = 0; nb < 2; ++nb) {
    run<0>("v0 copy + release/ticket + acquire", 64, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<1>("v1 no copy", 64, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<2>("v2 flag only, one-wave sys acquire", 64, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<3>("v3 flag only, agent acquire", 64, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<4>("v4 flag only, no acquire", 64, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<4>("v4 flag only, no acquire", 128, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<4>("v4 flag only, no acquire", 256, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<2>("v2 flag only, one-wave sys acquire", 256, g, slot_fg, n4, flag, counter, parts, w, out, nb, false);
    run<4>("v5 = v4 on ordinary memory", 64, g, slot_cg, n4, flag, counter, parts, w, out, nb, false);
    run<4>("v5 = v4 on ordinary memory", 256, g, slot_cg, n4, flag, counter, parts, w, out, nb, false);
  }
  run<2>("v2 (fine-grained slot)", 256, g, slot_fg, n4, flag, counter, parts, w, out, true, true);
  run<4>("v5 (ordinary memory)", 256, g, slot_cg, n4, flag, counter, parts, w, out, true, true);
  return 0;
}
