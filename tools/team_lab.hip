// Feasibility lab for the small-batch decoder recurrence (tuning aid, not product code): what does ONE phase seam cost when the
// workgroups that exchange data all sit on ONE XCD and synchronise through that XCD's L2?
//   * 256 workgroups are launched (one per CU); each reads its XCC id (s_getreg HW_REG_XCC_ID) and claims a member slot of that XCD's team --
//     placement is READ, never assumed;
//   * per iteration every member writes its slice of a [rows, 512] activation image, the team meets at a counter, every member re-reads the whole
//     image (what the next product's A operand needs) and checks every word;
//   * mode 0: barrier only; mode 1: plain stores + workgroup-scope (L2) atomic + sc1 loads  (valid only because the team shares one L2);
//     mode 2: sc1 write-through stores + agent-scope atomic + sc1 loads (valid at any placement); mode 3: as 2 but ONE team of all 256 workgroups
//     (the chip-wide seam the guide prices at 4-7 us).
//   hipcc -O3 --offload-arch=gfx950 -o tools/_bin/team_lab tools/team_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Ctl {
  unsigned claim[8];          // members claimed per XCC
  unsigned started;           // workgroups that have claimed
  unsigned pad0[7];
  unsigned ctr[8][32];        // one barrier counter per team, on lines of their own
  unsigned err;               // mismatching words
  unsigned timeout;           // a spin gave up
  unsigned long long t0[8], t1[8];
  unsigned size[8];
};

typedef unsigned v4u __attribute__((ext_vector_type(4)));
// sc1 (aux = 16) 16-byte accesses the compiler counts in its own s_waitcnt bookkeeping (guide: Guideline 16, R1)
__device__ __forceinline__ float4 ld_sc1(__amdgpu_buffer_rsrc_t r, int i) {
  const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, i * 16, 0, 16);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void st_sc1(__amdgpu_buffer_rsrc_t r, int i, float4 q) {
  v4u v; v.x = __float_as_uint(q.x); v.y = __float_as_uint(q.y); v.z = __float_as_uint(q.z); v.w = __float_as_uint(q.w);
  __builtin_amdgcn_raw_buffer_store_b128(v, r, i * 16, 0, 16);
}

__device__ __forceinline__ bool spin_ge(const unsigned* p, unsigned target, Ctl* c) {
  const long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    if (wall_clock64() - t0 > 200000000LL) { atomicAdd(&c->timeout, 1u); return false; }      // 2 s at 100 MHz
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}

__global__ __launch_bounds__(256) void team_lab(Ctl* c, float* buf, int n_iter, int mode, int teams_wanted, int rows) {
  __shared__ unsigned sh[4];
  const int tid = threadIdx.x;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    if (mode == 3) xcc = 0;
    sh[0] = xcc;
    sh[1] = __hip_atomic_fetch_add(&c->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&c->started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[3] = spin_ge(&c->started, gridDim.x, c) ? 1u : 0u;
    sh[2] = __hip_atomic_load(&c->claim[xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int team = sh[0], member = sh[1], size = sh[2];
  if (!sh[3] || team >= teams_wanted) return;
  if (tid == 0 && member == 0) { c->size[team] = size; c->t0[team] = wall_clock64(); }
  const int P4 = rows * 512 / 4;                       // float4s of one image
  float* img = buf + (size_t)team * 2 * rows * 512;
  const int lo = (int)((long long)P4 * member / size), hi = (int)((long long)P4 * (member + 1) / size);
  unsigned bad = 0;
  for (int it = 1; it <= n_iter; ++it) {
    float4* cur = reinterpret_cast<float4*>(img + (size_t)(it & 1) * rows * 512);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(cur, 0, rows * 512 * 4, 0x00020000);
    if (mode != 0) {
      for (int i = lo + tid; i < hi; i += 256) {
        const float v = (float)(it * 7 + (i & 1023));
        const float4 q = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
        if (mode == 1) cur[i] = q; else st_sc1(rs, i, q);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      if (mode == 1) __hip_atomic_fetch_add(&c->ctr[team][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      else __hip_atomic_fetch_add(&c->ctr[team][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sh[3] = spin_ge(&c->ctr[team][0], (unsigned)it * (unsigned)size, c) ? 1u : 0u;
    }
    __syncthreads();
    if (!sh[3]) return;
    if (mode != 0) {
      for (int i0 = tid; i0 < P4; i0 += 256 * 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; v[u] = ld_sc1(rs, i < P4 ? i : P4 - 1); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int i = i0 + 256 * u;
          if (i < P4) { const float w = (float)(it * 7 + (i & 1023)); bad += (v[u].x != w) + (v[u].y != w + 1.f) + (v[u].z != w + 2.f) + (v[u].w != w + 3.f); }
        }
      }
    }
  }
  if (bad) atomicAdd(&c->err, bad);
  __syncthreads();
  if (tid == 0 && member == 0) c->t1[team] = wall_clock64();
}

int main(int argc, char** argv) {
  const int n_iter = argc > 1 ? atoi(argv[1]) : 2000;
  Ctl* c; float* buf;
  CK(hipMalloc(&c, sizeof(Ctl)));
  CK(hipMalloc(&buf, sizeof(float) * 8 * 2 * 1024 * 512));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int rows_list[] = {32, 64, 128};
  for (int mode = 0; mode <= 3; ++mode)
    for (int teams : {1, 8})
      for (int rows : rows_list) {
        if (mode == 0 && rows != 32) continue;
        if (mode == 3 && teams != 1) continue;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipMemset(c, 0, sizeof(Ctl)));
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(team_lab, dim3(256), dim3(256), 0, 0, c, buf, n_iter, mode, teams, rows);
          CK(hipEventRecord(e1));
          CK(hipDeviceSynchronize());
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          Ctl h; CK(hipMemcpy(&h, c, sizeof(Ctl), hipMemcpyDeviceToHost));
          if (rep == 0) continue;
          double in_us = 0; int nt = 0;
          for (int t = 0; t < 8; ++t) if (h.t1[t] > h.t0[t]) { in_us += (double)(h.t1[t] - h.t0[t]) / 100.0; ++nt; }
          printf("mode %d teams %d rows %3d: %.3f us per seam (event), %.3f in-kernel; team sizes", mode, teams, rows, ms * 1e3 / n_iter, nt ? in_us / nt / n_iter : 0.0);
          for (int t = 0; t < 8; ++t) printf(" %u", h.claim[t]);
          printf("  err %u timeout %u\n", h.err, h.timeout);
        }
      }
  return 0;
}
