// LayerNorm (+ fused residual add) forward/backward and the DistillLayer tail
// (BatchNorm1d -> ELU -> MaxPool1d(3,2,1)) forward/backward.
// Reference arithmetic: nn.LayerNorm inside nn.Transformer layers (post-norm: norm(x + sublayer(x)));
// DistillLayer: viewport_prediction/models/customized_transformer.py:13-36.
// All HBM-bound: one wavefront per row with 16-byte accesses, column sums reduced in registers
// -> LDS -> one atomic per column per workgroup.
#include <cstdlib>
#include "mansy_kernels.h"

#define RC_HOOK(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

namespace {

constexpr int LN_MAXV = 4;     // float4 per lane -> C <= 1024

// NV = C / 256 float4 per lane.  One wave per row (several rows per wave when there are more rows than waves); every load of a
// row is issued before its first store (z_out may alias a: the compiler keeps loads behind earlier stores).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ z_out, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int rows, int C,
                                                            float eps, unsigned short* __restrict__ y16, const unsigned short* __restrict__ a16) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  float4 wv[NV], bv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) { wv[i] = *reinterpret_cast<const float4*>(w + (i * 64 + lane) * 4); bv[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
  if (bias) {
#pragma unroll
    for (int i = 0; i < NV; ++i) bv[i] = *reinterpret_cast<const float4*>(bias + (i * 64 + lane) * 4);
  }
  for (int row = wave_global; row < rows; row += nwaves) {
    const long long base = (long long)row * C + lane * 4;
    float4 v[NV], rv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (a16) {           // bf16 residual stream: the input row is read from its image
        const mansy_bf16x4 t = *reinterpret_cast<const mansy_bf16x4*>(a16 + base + i * 256);
        v[i] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
      } else v[i] = *reinterpret_cast<const float4*>(a + base + i * 256);
      rv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (b) {
#pragma unroll
      for (int i = 0; i < NV; ++i) rv[i] = *reinterpret_cast<const float4*>(b + base + i * 256);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (b) { v[i].x += rv[i].x; v[i].y += rv[i].y; v[i].z += rv[i].z; v[i].w += rv[i].w; }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    if (z_out) {
#pragma unroll
      for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(z_out + base + i * 256) = v[i];
    }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
    const float var = wave_sum(q) / (float)C;
    const float rs = 1.0f / sqrtf(var + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float4 o;
      o.x = (v[i].x - mu) * rs * wv[i].x; o.y = (v[i].y - mu) * rs * wv[i].y;
      o.z = (v[i].z - mu) * rs * wv[i].z; o.w = (v[i].w - mu) * rs * wv[i].w;
      if (bias) { o.x += bv[i].x; o.y += bv[i].y; o.z += bv[i].z; o.w += bv[i].w; }
      if (y) *reinterpret_cast<float4*>(y + base + i * 256) = o;
      if (y16) mansy_st_bf16x4(y16 + base + i * 256, o.x, o.y, o.z, o.w);
    }
  }
}

// Generic (any C) one-wave-per-row fallback used when C is not a multiple of 256.
__global__ __launch_bounds__(256) void layernorm_fwd_generic(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ z_out, float* __restrict__ y,
                                                             float* __restrict__ mean, float* __restrict__ rstd, int rows, int C,
                                                             float eps) {
  const int lane = threadIdx.x & 63;
  const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  for (int row = wave_global; row < rows; row += nwaves) {
    const long long base = (long long)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) { float x = a[base + c]; if (b) x += b[base + c]; s += x; }
    const float mu = wave_sum(s) / (float)C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { float x = a[base + c]; if (b) x += b[base + c]; const float d = x - mu; q += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
    for (int c = lane; c < C; c += 64) {
      float x = a[base + c]; if (b) x += b[base + c];
      if (z_out) z_out[base + c] = x;
      float o = (x - mu) * rs * w[c];
      if (bias) o += bias[c];
      y[base + c] = o;
    }
  }
}

// dz = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy * w
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ w, float* __restrict__ dz,
                                                            float* __restrict__ dz_drop, MansyDrop drop, float* __restrict__ dw,
                                                            float* __restrict__ dbias, int rows, int C) {
  extern __shared__ float red[];      // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  const float dsc = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float* my_dw = red + (wave * 2 + 0) * C;
  float* my_db = red + (wave * 2 + 1) * C;
  for (int c = lane; c < C; c += 64) { my_dw[c] = 0.f; my_db[c] = 0.f; }
  for (int row = wave_global; row < rows; row += nwaves) {
    const long long base = (long long)row * C;
    const float mu = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float g = dy[base + c] * w[c];
      const float xh = (z[base + c] - mu) * rs;
      s1 += g; s2 += g * xh;
    }
    s1 = wave_sum(s1) / (float)C; s2 = wave_sum(s2) / (float)C;
    for (int c = lane; c < C; c += 64) {
      const float d = dy[base + c];
      const float xh = (z[base + c] - mu) * rs;
      const float o = rs * (d * w[c] - s1 - xh * s2);
      dz[base + c] = o;
      if (dz_drop) {
        float od = o;
        if (drop.p > 0.f) od = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(base + c), drop.p) ? o * dsc : 0.f;
        dz_drop[base + c] = od;
      }
      my_dw[c] += d * xh;
      my_db[c] += d;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) { a0 += red[(wv * 2 + 0) * C + c]; a1 += red[(wv * 2 + 1) * C + c]; }
    if (dw) atomicAdd(dw + c, a0);
    if (dbias) atomicAdd(dbias + c, a1);
  }
}

// Vectorised backward for C % 256 == 0, C <= 1024: float4 per lane, column sums kept in registers.
// NV = C / 256 float4 per lane and row.  A wave takes RB = 4 rows per iteration and issues all their dy / z loads before
// touching any (16 KiB in flight per wave): with one row at a time a wave spends two dependent HBM round trips per row.
// PART: the workgroup's column sums go to its own slot of `dw` = partials[gridDim.x][2][C] (stored, or added when
// dbias != nullptr is used as the "accumulate" flag) instead of gridDim.x-way contended atomics on the C weight-gradient
// addresses -- those serialise in L2 and were half of the kernel's time on [4096, 512] inputs.
template <int NV, bool PART, int RB>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ w, float* __restrict__ dz,
                                                                float* __restrict__ dz_drop, MansyDrop drop, float* __restrict__ dw,
                                                                float* __restrict__ dbias, int rows, int C, unsigned short* __restrict__ dz_drop16,
                                                                const unsigned short* __restrict__ z16) {
  extern __shared__ float red[];      // [4 waves][2][C]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * 256) >> 6;
  const float dsc = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  const float invC = 1.f / (float)C;
  float4 ww[NV], adw[NV], adb[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    adw[i] = make_float4(0.f, 0.f, 0.f, 0.f); adb[i] = adw[i];
    ww[i] = *reinterpret_cast<const float4*>(w + (i * 64 + lane) * 4);
  }
  for (int row0 = wave_global * RB; row0 < rows; row0 += nwaves * RB) {
    float4 d[RB][NV], zz[RB][NV];
    float mu[RB], rs[RB];
    // (ONE uniform branch around all loads of the iteration, not one per load: every dy / z load of the RB rows is issued before any is used)
    if (z16) {           // bf16 residual stream: the LayerNorm's input was kept as its image only
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int row = min(row0 + u, rows - 1);          // rows past the end re-read the last row; they are not written / summed
        mu[u] = mean[row]; rs[u] = rstd[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const long long off = (long long)row * C + (i * 64 + lane) * 4;
          d[u][i] = *reinterpret_cast<const float4*>(dy + off);
          const mansy_bf16x4 t = *reinterpret_cast<const mansy_bf16x4*>(z16 + off);
          zz[u][i] = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int row = min(row0 + u, rows - 1);
        mu[u] = mean[row]; rs[u] = rstd[row];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const long long off = (long long)row * C + (i * 64 + lane) * 4;
          d[u][i] = *reinterpret_cast<const float4*>(dy + off);
          zz[u][i] = *reinterpret_cast<const float4*>(z + off);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const bool live = row0 + u < rows;
      const long long base = (long long)(row0 + u) * C;
      float4 xh[NV];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const float4 t = zz[u][i];
        xh[i] = make_float4((t.x - mu[u]) * rs[u], (t.y - mu[u]) * rs[u], (t.z - mu[u]) * rs[u], (t.w - mu[u]) * rs[u]);
        const float gx = d[u][i].x * ww[i].x, gy = d[u][i].y * ww[i].y, gz = d[u][i].z * ww[i].z, gw = d[u][i].w * ww[i].w;
        s1 += (gx + gy) + (gz + gw);
        s2 += (gx * xh[i].x + gy * xh[i].y) + (gz * xh[i].z + gw * xh[i].w);
      }
      s1 = wave_sum(s1) * invC; s2 = wave_sum(s2) * invC;
      if (live) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const long long off = base + (i * 64 + lane) * 4;
          const float4 di = d[u][i];
          float4 o;
          o.x = rs[u] * (di.x * ww[i].x - s1 - xh[i].x * s2); o.y = rs[u] * (di.y * ww[i].y - s1 - xh[i].y * s2);
          o.z = rs[u] * (di.z * ww[i].z - s1 - xh[i].z * s2); o.w = rs[u] * (di.w * ww[i].w - s1 - xh[i].w * s2);
          *reinterpret_cast<float4*>(dz + off) = o;
          if (dz_drop || dz_drop16) {
            float4 od = o;
            if (drop.p > 0.f) {
              od.x = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(off + 0), drop.p) ? o.x * dsc : 0.f;
              od.y = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(off + 1), drop.p) ? o.y * dsc : 0.f;
              od.z = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(off + 2), drop.p) ? o.z * dsc : 0.f;
              od.w = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(off + 3), drop.p) ? o.w * dsc : 0.f;
            }
            if (dz_drop) *reinterpret_cast<float4*>(dz_drop + off) = od;
            if (dz_drop16) mansy_st_bf16x4(dz_drop16 + off, od.x, od.y, od.z, od.w);
          }
          adw[i].x += di.x * xh[i].x; adw[i].y += di.y * xh[i].y; adw[i].z += di.z * xh[i].z; adw[i].w += di.w * xh[i].w;
          adb[i].x += di.x; adb[i].y += di.y; adb[i].z += di.z; adb[i].w += di.w;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    *reinterpret_cast<float4*>(red + (wave * 2 + 0) * C + (i * 64 + lane) * 4) = adw[i];
    *reinterpret_cast<float4*>(red + (wave * 2 + 1) * C + (i * 64 + lane) * 4) = adb[i];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) { a0 += red[(wv * 2 + 0) * C + c]; a1 += red[(wv * 2 + 1) * C + c]; }
    if (PART) {
      float* slot = dw + (size_t)blockIdx.x * 2 * C;
      if (dbias) { a0 += slot[c]; a1 += slot[C + c]; }
      slot[c] = a0; slot[C + c] = a1;
    } else {
      if (dw) atomicAdd(dw + c, a0);
      if (dbias) atomicAdd(dbias + c, a1);
    }
  }
}

// dw[c] += sum_p partials[p][0][c], dbias[c] += sum_p partials[p][1][c]; grid (ceil(C/64), 2, ceil(nparts/64)): a workgroup
// sums 64 slots x 64 columns (4 slot-groups of 16, all 16 loads of a thread in flight) and adds its share atomically.
__global__ __launch_bounds__(256) void ln_partials_reduce_kernel(const float* __restrict__ partials, int nparts, int C,
                                                                 float* __restrict__ dw, float* __restrict__ dbias) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, which = blockIdx.y;
  float* dst = which == 0 ? dw : dbias;
  if (!dst) return;
  float acc = 0.f;
  if (c < C) {
    const float* src = partials + (size_t)which * C + c;
    const int p0 = blockIdx.z * 64 + grp * 16;
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(p0 + u, nparts - 1) * 2 * C];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += p0 + u < nparts ? v[u] : 0.f;
  }
  red[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && c < C) atomicAdd(dst + c, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// the same for up to LN_MULTI_MAX slot sets in ONE launch (round 5: the decoder's 7 / 14 sets and the encoder's 5 at the end of a backward were
// twelve launches of ~4 us each at the reference's batch sizes): blockIdx.y = 2 * set + which
struct LnMulti { const float* partials[LN_MULTI_MAX]; int nparts[LN_MULTI_MAX]; float* dw[LN_MULTI_MAX]; float* dbias[LN_MULTI_MAX]; };
__global__ __launch_bounds__(256) void ln_partials_reduce_multi_kernel(LnMulti t, int C) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane, set = blockIdx.y >> 1, which = blockIdx.y & 1;
  float* dst = which == 0 ? t.dw[set] : t.dbias[set];
  const int nparts = t.nparts[set];
  if (!dst || (int)blockIdx.z * 64 >= nparts) return;
  float acc = 0.f;
  if (c < C) {
    const float* src = t.partials[set] + (size_t)which * C + c;
    const int p0 = blockIdx.z * 64 + grp * 16;
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = src[(size_t)min(p0 + u, nparts - 1) * 2 * C];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += p0 + u < nparts ? v[u] : 0.f;
  }
  red[grp][lane] = acc;
  __syncthreads();
  if (grp == 0 && c < C) atomicAdd(dst + c, (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]));
}

// ------------------------------------------------------------------ DistillLayer tail
// column sums of x and x^2 over rows -> doubles
__global__ __launch_bounds__(256) void colstats_kernel(const float* __restrict__ x, int rows, int C, double* __restrict__ stats, double* __restrict__ part) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int r0 = blockIdx.y, rstep = gridDim.y;
  double s = 0.0, q = 0.0;
  constexpr int U = 8;                                    // loads of U rows issued before their sums (double adds are sequential per thread)
  for (int r = r0; r < rows; r += U * rstep) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = x[(long long)min(r + u * rstep, rows - 1) * C + c];
#pragma unroll
    for (int u = 0; u < U; ++u) if (r + u * rstep < rows) { const double d = v[u]; s += d; q += d * d; }
  }
  if (part) {                                             // per-workgroup partial sums, added up in order by col_parts_reduce_kernel
    part[((long long)blockIdx.y * 2 + 0) * C + c] = s;
    part[((long long)blockIdx.y * 2 + 1) * C + c] = q;
    return;
  }
  atomicAdd(stats + c, s);
  atomicAdd(stats + C + c, q);
}
// dst[i] = sum_p part[p][i], i < n (= 2 C), p in order: the second half of a column reduction whose first half left one partial per workgroup.
// (Round 4: the first half used to add its sums with double atomics -- 512 to 1024 workgroups on the same 2 C addresses, resolved at the memory side
// across the eight XCDs: ~80 us of a 125 us kernel.)
__global__ __launch_bounds__(256) void col_parts_reduce_kernel(const double* __restrict__ part, int nparts, int n, double* __restrict__ dst) {
  // grid (ceil(n / 256), ceil(nparts / 32)): a thread sums up to 32 partials of its column and adds the sum to the (zeroed) destination -- at most
  // 16 adds per address.  (First form: one thread per column over all 512 partials: 30 us for 4 MB.)
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int p0 = blockIdx.y * 32, p1 = min(p0 + 32, nparts);
  double v[32];
#pragma unroll
  for (int u = 0; u < 32; ++u) v[u] = part[(long long)min(p0 + u, nparts - 1) * n + i];
  double t = 0.0;
#pragma unroll
  for (int u = 0; u < 32; ++u) if (p0 + u < p1) t += v[u];
  atomicAdd(dst + i, t);
}

__global__ void bn_finalize_kernel(const double* __restrict__ stats, int n, int C, float* run_mean, float* run_var,
                                   long long* num_batches, float* mean_out, float* rstd_out, int train, float eps, float momentum) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && train && num_batches) *num_batches += 1;
  if (c >= C) return;
  if (train) {
    const double mu = stats[c] / n;
    double var = stats[C + c] / n - mu * mu;
    if (var < 0) var = 0;
    mean_out[c] = (float)mu;
    rstd_out[c] = (float)(1.0 / sqrt(var + (double)eps));
    const double var_u = n > 1 ? var * ((double)n / (double)(n - 1)) : var;
    run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mu;
    run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)var_u;
  } else {
    mean_out[c] = run_mean[c];
    rstd_out[c] = 1.0f / sqrtf(run_var[c] + eps);
  }
}

__device__ __forceinline__ float elu1(float v) { return v > 0.f ? v : expm1f(v); }

__global__ __launch_bounds__(256) void bn_elu_pool_kernel(const float* __restrict__ conv, const float* __restrict__ bn_w,
                                                          const float* __restrict__ bn_b, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ mem,
                                                          unsigned char* __restrict__ argmax, DistillShape s, unsigned short* __restrict__ mem16) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)s.B * s.M * s.C;
  if (idx >= total) return;
  const int c = (int)(idx % s.C);
  const int m = (int)((idx / s.C) % s.M);
  const int b = (int)(idx / ((long long)s.C * s.M));
  const float mu = mean[c], sc = rstd[c] * bn_w[c], sh = bn_b[c];
  float best = -INFINITY; int bi = 0;
  for (int t = -1; t <= 1; ++t) {
    const int sp = 2 * m + t;
    if (sp < 0 || sp >= s.S) continue;
    const float v = elu1((conv[((long long)b * s.S + sp) * s.C + c] - mu) * sc + sh);
    if (v > best) { best = v; bi = sp; }     // first maximum wins (torch max_pool1d tie rule)
  }
  mem[idx] = best;
  if (mem16) mansy_st_bf16(mem16 + idx, best);      // bf16-storage mode: the memory is the A operand of every decoder layer's K/V projection
  if (argmax) argmax[idx] = (unsigned char)bi;
}

// stage 1: g[b,s,c] = dL/d(BN output) ; column sums of g and g*xhat -> doubles (stats[2C..4C))
// g = dL/d(BN output) through the max-pool scatter and the ELU; per-channel [sum g, sum g*xhat] in double.
// V = 4: a thread owns 4 channels (16-byte accesses) and every 2*gridDim.y-th row (V = 1: 1 channel, every gridDim.y-th);
// the pooled-gradient loads are unconditional (a predicate around them made each row a dependent round trip).
template <int V>
__global__ __launch_bounds__(256) void distill_bwd_stage1(const float* __restrict__ conv, const float* __restrict__ dmem,
                                                          const unsigned char* __restrict__ argmax, const float* __restrict__ bn_w,
                                                          const float* __restrict__ bn_b, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ g, double* __restrict__ stats,
                                                          DistillShape s, double* __restrict__ part) {
  const int CV = s.C / V;
  const int tpr = CV < 256 ? CV : 256;                  // threads per row (CV a multiple of 256, or a divisor of it)
  const int rsub = threadIdx.x / tpr, rpw = 256 / tpr;  // row sub-stream of this thread, row streams per workgroup
  const int c = (blockIdx.x * tpr + threadIdx.x % tpr) * V;
  __shared__ double sh_part[256][2 * V + 1];
  if (c >= s.C || rsub >= rpw) return;                  // (never true on the partial-sum path: CV is a multiple or a divisor of 256 there)
  float mu[V], rs[V], w[V], sh[V];
  double sg[V], sgx[V];
#pragma unroll
  for (int j = 0; j < V; ++j) { mu[j] = mean[c + j]; rs[j] = rstd[c + j]; w[j] = bn_w[c + j]; sh[j] = bn_b[c + j]; sg[j] = 0.0; sgx[j] = 0.0; }
  const int rows = s.B * s.S;
  constexpr int U = 4;                                    // rows in flight per thread: every load of the U rows is issued before the first use (one row at a
                                                          // time made the [40 960-row] call 80 dependent round trips per thread: 129 us for 220 MB)
  const int rstep = gridDim.y * rpw;
  for (int r0 = blockIdx.y * rpw + rsub; r0 < rows; r0 += U * rstep) {
    float d0[U][V], d1[U][V], xv[U][V];
    unsigned char a0[U][V], a1[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(r0 + u * rstep, rows - 1);        // clamped: rows beyond the end are recomputed and dropped
      const int b = r / s.S, sp = r % s.S;
      // pooling windows containing sp: 2m-1 <= sp <= 2m+1  <=>  m in {sp/2, (sp+1)/2} (the same window twice when sp is even)
      const int m0 = min(sp / 2, s.M - 1), m1 = min((sp + 1) / 2, s.M - 1);
      const long long i0 = ((long long)b * s.M + m0) * s.C + c, i1 = ((long long)b * s.M + m1) * s.C + c;
      if (V == 4) {
        *reinterpret_cast<float4*>(d0[u]) = *reinterpret_cast<const float4*>(dmem + i0);
        *reinterpret_cast<float4*>(d1[u]) = *reinterpret_cast<const float4*>(dmem + i1);
        *reinterpret_cast<float4*>(xv[u]) = *reinterpret_cast<const float4*>(conv + (long long)r * s.C + c);
        *reinterpret_cast<uchar4*>(a0[u]) = *reinterpret_cast<const uchar4*>(argmax + i0);
        *reinterpret_cast<uchar4*>(a1[u]) = *reinterpret_cast<const uchar4*>(argmax + i1);
      } else {
        d0[u][0] = dmem[i0]; d1[u][0] = dmem[i1]; xv[u][0] = conv[(long long)r * s.C + c]; a0[u][0] = argmax[i0]; a1[u][0] = argmax[i1];
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * rstep;
      if (r >= rows) break;
      const int sp = r % s.S;
      const bool two = (sp + 1) / 2 != sp / 2 && (sp + 1) / 2 < s.M;
      const bool one = sp / 2 < s.M;
      float gg[V];
#pragma unroll
      for (int j = 0; j < V; ++j) {
        float up = 0.f;
        if (one && a0[u][j] == (unsigned char)sp) up += d0[u][j];
        if (two && a1[u][j] == (unsigned char)sp) up += d1[u][j];
        const float xh = (xv[u][j] - mu[j]) * rs[j];
        const float ypre = xh * w[j] + sh[j];
        gg[j] = up * (ypre > 0.f ? 1.f : expf(ypre));
        sg[j] += gg[j]; sgx[j] += (double)gg[j] * xh;
      }
      if (V == 4) *reinterpret_cast<float4*>(g + (long long)r * s.C + c) = *reinterpret_cast<const float4*>(gg);
      else g[(long long)r * s.C + c] = gg[0];
    }
  }
  if (part) {                                             // the workgroup's row streams combined through LDS, one partial per workgroup and column
#pragma unroll
    for (int j = 0; j < V; ++j) { sh_part[threadIdx.x][j] = sg[j]; sh_part[threadIdx.x][V + j] = sgx[j]; }
    __syncthreads();
    if (rsub == 0) {
      for (int q = 1; q < rpw; ++q)
#pragma unroll
        for (int j = 0; j < V; ++j) { sg[j] += sh_part[q * tpr + threadIdx.x][j]; sgx[j] += sh_part[q * tpr + threadIdx.x][V + j]; }
#pragma unroll
      for (int j = 0; j < V; ++j) {
        part[((long long)blockIdx.y * 2 + 0) * s.C + c + j] = sg[j];
        part[((long long)blockIdx.y * 2 + 1) * s.C + c + j] = sgx[j];
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < V; ++j) {
    atomicAdd(stats + 2 * s.C + c + j, sg[j]);
    atomicAdd(stats + 3 * s.C + c + j, sgx[j]);
  }
}

__global__ __launch_bounds__(256) void distill_bwd_stage2(const float* __restrict__ conv, const float* __restrict__ g,
                                                          const float* __restrict__ bn_w, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const double* __restrict__ stats,
                                                          float* __restrict__ dconv, float* __restrict__ dbn_w, float* __restrict__ dbn_b,
                                                          DistillShape s, unsigned short* __restrict__ dconv16) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long total = (long long)s.B * s.S * s.C;
  if (idx >= total) return;
  const int c = (int)(idx % s.C);
  const double n = (double)s.B * s.S * (s.sync_world > 1 ? s.sync_world : 1);
  const float sg = (float)(stats[2 * s.C + c] / n), sgx = (float)(stats[3 * s.C + c] / n);
  const float rs = rstd[c];
  const float xh = (conv[idx] - mean[c]) * rs;
  const float dc = bn_w[c] * rs * (g[idx] - sg - xh * sgx);
  dconv[idx] = dc;
  if (dconv16) mansy_st_bf16(dconv16 + idx, dc);    // bf16-storage mode: operand of the conv's dW and dX products
  if (idx < s.C) {   // first row's threads publish the parameter gradients
    atomicAdd(dbn_w + c, (float)stats[5 * s.C + c]);
    atomicAdd(dbn_b + c, (float)stats[4 * s.C + c]);
  }
}

}  // namespace

int mansy_launch_layernorm_fwd(const float* a, const float* b, const float* w, const float* bias, float* z_out, float* y,
                               float* mean, float* rstd, int rows, int C, float eps, hipStream_t st, unsigned short* y16, const unsigned short* a16) {
  MANSY_REQUIRE((a || a16) && w && (y || y16), "layernorm_fwd: null pointer");
  MANSY_REQUIRE(!a16 || (!b && !z_out && (C % 256) == 0 && C <= 256 * LN_MAXV), "layernorm_fwd: a bf16 input needs the vectorised kernel and no second addend");
  MANSY_REQUIRE(!y16 || ((C % 256) == 0 && C <= 256 * LN_MAXV), "layernorm_fwd: the bf16 image needs the vectorised kernel (C %% 256 == 0)");
  if (rows <= 0) return MANSY_OK;
  const int grid = min(mansy_ceil_div(rows, 4), 2048);
  if ((C % 256) == 0 && C <= 256 * LN_MAXV) {
    switch (C / 256) {
      case 1: MANSY_LAUNCH(layernorm_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, a, b, w, bias, z_out, y, mean, rstd, rows, C, eps, y16, a16); break;
      case 2: MANSY_LAUNCH(layernorm_fwd_kernel<2>, dim3(grid), dim3(256), 0, st, a, b, w, bias, z_out, y, mean, rstd, rows, C, eps, y16, a16); break;
      case 3: MANSY_LAUNCH(layernorm_fwd_kernel<3>, dim3(grid), dim3(256), 0, st, a, b, w, bias, z_out, y, mean, rstd, rows, C, eps, y16, a16); break;
      default: MANSY_LAUNCH(layernorm_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, a, b, w, bias, z_out, y, mean, rstd, rows, C, eps, y16, a16); break;
    }
  } else
    MANSY_LAUNCH(layernorm_fwd_generic, dim3(grid), dim3(256), 0, st, a, b, w, bias, z_out, y, mean, rstd, rows, C, eps);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// rows per wave of the partial-sum backward: 2 for the decoder-step calls (<= 8192 rows: one wave per SIMD carries the whole ~1500-instruction body of its
// rows, so half the rows per wave is a shorter launch: VP step 22.32 -> 22.20 ms; 1 row per wave gives it back), 4 for the [40 960-row] encoder calls
static int ln_rows_per_wave(int rows) { return rows <= 8192 ? 2 : 4; }
int mansy_launch_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                               float* dz, float* dz_drop, MansyDrop drop, float* dw, float* dbias, int rows, int C,
                               hipStream_t st) {
  MANSY_REQUIRE(dy && z && mean && rstd && w && dz, "layernorm_bwd: null pointer");
  if (rows <= 0) return MANSY_OK;
  const int grid = min(mansy_ceil_div(rows, 16), 1024);     // 4 rows per wave: more workgroups only multiply the per-column atomics (measured slower)
  const size_t lds = (size_t)4 * 2 * C * sizeof(float);
  MANSY_REQUIRE(lds <= 64 * 1024, "layernorm_bwd: C=%d too large", C);
  if ((C % 256) == 0 && C <= 256 * LN_MAXV) {
    switch (C / 256) {
      case 1: MANSY_LAUNCH((layernorm_bwd_vec_kernel<1, false, 4>), dim3(grid), dim3(256), lds, st, dy, z, mean, rstd, w, dz, dz_drop, drop, dw, dbias, rows, C, (unsigned short*)nullptr, (const unsigned short*)nullptr); break;
      case 2:      // (the same rows-per-wave rule as the partial-sum form: the two forms stay bit-identical per row)
        if (ln_rows_per_wave(rows) == 2) MANSY_LAUNCH((layernorm_bwd_vec_kernel<2, false, 2>), dim3(grid), dim3(256), lds, st, dy, z, mean, rstd, w, dz, dz_drop, drop, dw, dbias, rows, C, (unsigned short*)nullptr, (const unsigned short*)nullptr);
        else MANSY_LAUNCH((layernorm_bwd_vec_kernel<2, false, 4>), dim3(grid), dim3(256), lds, st, dy, z, mean, rstd, w, dz, dz_drop, drop, dw, dbias, rows, C, (unsigned short*)nullptr, (const unsigned short*)nullptr);
        break;
      case 3: MANSY_LAUNCH((layernorm_bwd_vec_kernel<3, false, 4>), dim3(grid), dim3(256), lds, st, dy, z, mean, rstd, w, dz, dz_drop, drop, dw, dbias, rows, C, (unsigned short*)nullptr, (const unsigned short*)nullptr); break;
      default: MANSY_LAUNCH((layernorm_bwd_vec_kernel<4, false, 4>), dim3(grid), dim3(256), lds, st, dy, z, mean, rstd, w, dz, dz_drop, drop, dw, dbias, rows, C, (unsigned short*)nullptr, (const unsigned short*)nullptr); break;
    }
  } else
    MANSY_LAUNCH(layernorm_bwd_kernel, dim3(grid), dim3(256), lds, st, dy, z, mean, rstd, w, dz, dz_drop, drop, dw, dbias,
                       rows, C);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_distill_fwd(const float* conv, const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                             long long* num_batches, float* mean_out, float* rstd_out, float* mem, unsigned char* argmax,
                             double* stats_d, const DistillShape& s, int train, float eps, float momentum, hipStream_t st, double* part, unsigned short* mem16) {
  MANSY_REQUIRE(conv && bn_w && bn_b && run_mean && run_var && mean_out && rstd_out && mem && stats_d, "distill_fwd: null pointer");
  MANSY_REQUIRE(s.M == (s.S - 1) / 2 + 1, "distill_fwd: M must be floor((S-1)/2)+1");
  MANSY_REQUIRE(s.S <= 255, "distill_fwd: S too large");
  const int rows = s.B * s.S;
  if (train) {
    MANSY_HIP_CHECK(hipMemsetAsync(stats_d, 0, sizeof(double) * 6 * s.C, st));
    dim3 grid(mansy_ceil_div(s.C, 256), min(rows, MANSY_DISTILL_PARTS));
    MANSY_LAUNCH(colstats_kernel, grid, dim3(256), 0, st, conv, rows, s.C, stats_d, part);
    if (part) MANSY_LAUNCH(col_parts_reduce_kernel, dim3(mansy_ceil_div(2 * s.C, 256), mansy_ceil_div((int)grid.y, 32)), dim3(256), 0, st, part, (int)grid.y, 2 * s.C, stats_d);
    if (s.sync_world > 1) RC_HOOK(mansy_bn_sync_invoke(0, s.hook, s.hook_user));     // SyncBN: all-reduce [sum, sumsq] (2C doubles) over the data-parallel ranks
  }
  const int n_glob = rows * (train && s.sync_world > 1 ? s.sync_world : 1);
  MANSY_LAUNCH(bn_finalize_kernel, dim3(mansy_ceil_div(s.C, 256)), dim3(256), 0, st, stats_d, n_glob, s.C, run_mean, run_var,
                     num_batches, mean_out, rstd_out, train, eps, momentum);
  const long long total = (long long)s.B * s.M * s.C;
  MANSY_LAUNCH(bn_elu_pool_kernel, dim3(mansy_ceil_div(total, 256)), dim3(256), 0, st, conv, bn_w, bn_b, mean_out, rstd_out,
                     mem, argmax, s, mem16);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_distill_bwd(const float* conv, const float* dmem, const unsigned char* argmax, const float* bn_w,
                             const float* bn_b, const float* mean, const float* rstd, float* g_tmp, float* dconv, float* dbn_w,
                             float* dbn_b, double* stats_d, const DistillShape& s, hipStream_t st, double* part, unsigned short* dconv16) {
  MANSY_REQUIRE(conv && dmem && argmax && bn_w && bn_b && mean && rstd && g_tmp && dconv && dbn_w && dbn_b && stats_d,
                "distill_bwd: null pointer");
  const int rows = s.B * s.S;
  MANSY_HIP_CHECK(hipMemsetAsync(stats_d + 2 * s.C, 0, sizeof(double) * 2 * s.C, st));
  auto al = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  const int cv = s.C / 4;
  if (s.C % 4 == 0 && (cv % 256 == 0 || 256 % cv == 0) && al(conv) && al(dmem) && al(g_tmp) && (reinterpret_cast<uintptr_t>(argmax) & 3) == 0) {
    const int tpr = cv < 256 ? cv : 256;
    dim3 grid1(mansy_ceil_div(cv, tpr), min(mansy_ceil_div(rows, 4 * (256 / tpr)), MANSY_DISTILL_PARTS));
    MANSY_LAUNCH(distill_bwd_stage1<4>, grid1, dim3(256), 0, st, conv, dmem, argmax, bn_w, bn_b, mean, rstd, g_tmp, stats_d, s, part);
    if (part) MANSY_LAUNCH(col_parts_reduce_kernel, dim3(mansy_ceil_div(2 * s.C, 256), mansy_ceil_div((int)grid1.y, 32)), dim3(256), 0, st, part, (int)grid1.y, 2 * s.C, stats_d + 2 * s.C);
  } else {
    dim3 grid1(mansy_ceil_div(s.C, 256), min(rows, 512));
    MANSY_LAUNCH(distill_bwd_stage1<1>, grid1, dim3(256), 0, st, conv, dmem, argmax, bn_w, bn_b, mean, rstd, g_tmp, stats_d, s, nullptr);
  }
  // parameter gradients use THIS rank's sums (the gradient all-reduce averages them); the input gradient needs the global ones
  MANSY_HIP_CHECK(hipMemcpyAsync(stats_d + 4 * s.C, stats_d + 2 * s.C, sizeof(double) * 2 * s.C, hipMemcpyDeviceToDevice, st));
  if (s.sync_world > 1) RC_HOOK(mansy_bn_sync_invoke(1, s.hook, s.hook_user));       // SyncBN backward: all-reduce [sum g, sum g*xhat]
  const long long total = (long long)rows * s.C;
  MANSY_LAUNCH(distill_bwd_stage2, dim3(mansy_ceil_div(total, 256)), dim3(256), 0, st, conv, g_tmp, bn_w, mean, rstd, stats_d,
                     dconv, dbn_w, dbn_b, s, dconv16);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// ---- LayerNorm backward with per-workgroup partial weight-gradient sums (no atomics): the engine's form.
int mansy_ln_bwd_parts(int rows) { return min(mansy_ceil_div(rows, 4 * ln_rows_per_wave(rows)), 1024); }
bool mansy_ln_bwd_partial_ok(int C) { return (C % 256) == 0 && C <= 256 * LN_MAXV; }
// partials: [mansy_ln_bwd_parts(rows)][2][C]; accumulate != 0 adds to what the slots hold (decoder: one LayerNorm
// applied at T steps), else the slots are overwritten.
int mansy_launch_layernorm_bwd_partial(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                                       float* dz, float* dz_drop, MansyDrop drop, float* partials, int accumulate, int rows, int C,
                                       hipStream_t st, unsigned short* dz_drop16, const unsigned short* z16) {
  MANSY_REQUIRE(dy && (z || z16) && mean && rstd && w && dz && partials, "layernorm_bwd_partial: null pointer");
  MANSY_REQUIRE(mansy_ln_bwd_partial_ok(C), "layernorm_bwd_partial: C=%d unsupported", C);
  if (rows <= 0) return MANSY_OK;
  const int grid = mansy_ln_bwd_parts(rows);
  const size_t lds = (size_t)4 * 2 * C * sizeof(float);
  float* flag = accumulate ? partials : nullptr;      // the kernel only tests it for null
#define LNB(NV, RBV) MANSY_LAUNCH((layernorm_bwd_vec_kernel<NV, true, RBV>), dim3(grid), dim3(256), lds, st, dy, z, mean, rstd, w, dz, dz_drop, drop, partials, flag, rows, C, dz_drop16, z16)
  if (C / 256 == 2) { if (ln_rows_per_wave(rows) == 2) LNB(2, 2); else LNB(2, 4); }
  else if (C / 256 == 1) LNB(1, 4);
  else if (C / 256 == 3) LNB(3, 4);
  else LNB(4, 4);
#undef LNB
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_ln_partials_reduce_multi(const MansyLnReduce* sets, int n, int C, hipStream_t st) {
  MANSY_REQUIRE(sets && n >= 1 && n <= LN_MULTI_MAX, "ln_partials_reduce_multi: 1..%d sets", LN_MULTI_MAX);
  LnMulti t; int maxp = 1;
  for (int i = 0; i < LN_MULTI_MAX; ++i) {
    const MansyLnReduce& s = sets[i < n ? i : 0];
    t.partials[i] = s.partials; t.nparts[i] = i < n ? s.nparts : 0; t.dw[i] = i < n ? s.dw : nullptr; t.dbias[i] = i < n ? s.dbias : nullptr;
    if (i < n) { MANSY_REQUIRE(s.partials && s.nparts >= 1, "ln_partials_reduce_multi: bad set %d", i); maxp = s.nparts > maxp ? s.nparts : maxp; }
  }
  MANSY_LAUNCH(ln_partials_reduce_multi_kernel, dim3(mansy_ceil_div(C, 64), 2 * n, mansy_ceil_div(maxp, 64)), dim3(256), 0, st, t, C);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_ln_partials_reduce(const float* partials, int nparts, int C, float* dw, float* dbias, hipStream_t st) {
  MANSY_REQUIRE(partials && nparts >= 1, "ln_partials_reduce: bad arguments");
  MANSY_LAUNCH(ln_partials_reduce_kernel, dim3(mansy_ceil_div(C, 64), 2, mansy_ceil_div(nparts, 64)), dim3(256), 0, st, partials, nparts, C, dw, dbias);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
