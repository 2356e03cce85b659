// HBM-bound glue kernels of the viewport-prediction path: ViewportEmbedding + PositionalEncoding
// (mtio.py:10-44), predictor Linear(d->6)+Sigmoid (mtio.py:60), MTIO loss (mtio.py:94-104,
// utils/common.py:73-80), MTIO channel mix (mtio.py:72-90), ensemble mean + wrap
// (mtio.py:125-133, utils/common.py:61-70), im2col for the circular conv, AdamW
// (run_models.py:29: torch.optim.AdamW defaults) and Adam-with-L2 (run_mansy.py:216).
#include "mansy_kernels.h"

namespace {

constexpr int MAX_IN = 8;   // in_channel * num_head = 6 in the reference

__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, int in_ch, const float* __restrict__ W,
                                                        const float* __restrict__ b, const float* __restrict__ pe,
                                                        float* __restrict__ out, int rows, int C, int S, int pos_fixed,
                                                        MansyDrop drop) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)rows * C) return;
  const int c = (int)(idx % C);
  const int r = (int)(idx / C);
  const int pos = pos_fixed >= 0 ? pos_fixed : (r % S);
  float acc = 0.f;
  for (int k = 0; k < in_ch; ++k) acc = fmaf(x[(long long)r * in_ch + k], W[c * in_ch + k], acc);
  if (b) acc += b[c];
  acc += pe[(long long)pos * C + c];
  if (drop.p > 0.f) acc = mansy_keep(drop.seed, drop.site, (uint32_t)idx, drop.p) ? acc * (1.f / (1.f - drop.p)) : 0.f;
  out[idx] = acc;
}

// one wave per row: dE = dX*mask ; dtok[r,k] = sum_c dE[r,c] W[c,k]
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dX, const float* __restrict__ W,
                                                        float* __restrict__ dE, float* __restrict__ dtok, int in_ch, int rows,
                                                        int C, MansyDrop drop) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float dsc = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float acc[MAX_IN];
#pragma unroll
  for (int k = 0; k < MAX_IN; ++k) acc[k] = 0.f;
  for (int c = lane; c < C; c += 64) {
    const long long idx = (long long)row * C + c;
    float g = dX[idx];
    if (drop.p > 0.f) g = mansy_keep(drop.seed, drop.site, (uint32_t)idx, drop.p) ? g * dsc : 0.f;
    if (dE) dE[idx] = g;
#pragma unroll
    for (int k = 0; k < MAX_IN; ++k) if (k < in_ch) acc[k] = fmaf(g, W[c * in_ch + k], acc[k]);
  }
  if (dtok) {
#pragma unroll
    for (int k = 0; k < MAX_IN; ++k) {
      if (k < in_ch) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) dtok[(long long)row * in_ch + k] = s;
      }
    }
  }
}

// out += sum_r small[r,k] * big[r,c]; grid (C/256, row-chunks)
__global__ __launch_bounds__(256) void outer_reduce_kernel(const float* __restrict__ small_, int small_n,
                                                           const float* __restrict__ big, int rows, int C,
                                                           float* __restrict__ out, int c_major, float* __restrict__ bsum_big,
                                                           float* __restrict__ bsum_small) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int r0 = blockIdx.y, rstep = gridDim.y;
  float acc[MAX_IN];
  float sb = 0.f;
#pragma unroll
  for (int k = 0; k < MAX_IN; ++k) acc[k] = 0.f;
  if (c < C) {
    for (int r = r0; r < rows; r += rstep) {
      const float v = big[(long long)r * C + c];
      sb += v;
#pragma unroll
      for (int k = 0; k < MAX_IN; ++k) if (k < small_n) acc[k] = fmaf(small_[(long long)r * small_n + k], v, acc[k]);
    }
#pragma unroll
    for (int k = 0; k < MAX_IN; ++k)
      if (k < small_n) atomicAdd(out + (c_major ? (long long)c * small_n + k : (long long)k * C + c), acc[k]);
    if (bsum_big) atomicAdd(bsum_big + c, sb);
  }
  if (bsum_small && blockIdx.x == 0 && threadIdx.x < small_n) {
    float s = 0.f;
    for (int r = r0; r < rows; r += rstep) s += small_[(long long)r * small_n + threadIdx.x];
    atomicAdd(bsum_small + threadIdx.x, s);
  }
}

__global__ __launch_bounds__(256) void predictor_fwd_kernel(const float* __restrict__ h, const float* __restrict__ W,
                                                            const float* __restrict__ b, float* __restrict__ y_a, long long sa,
                                                            float* __restrict__ y_b, long long sb, int rows, int C, int out_ch) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  float acc[MAX_IN];
#pragma unroll
  for (int k = 0; k < MAX_IN; ++k) acc[k] = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = h[(long long)row * C + c];
#pragma unroll
    for (int k = 0; k < MAX_IN; ++k) if (k < out_ch) acc[k] = fmaf(v, W[(long long)k * C + c], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < MAX_IN; ++k) {
    if (k < out_ch) {
      float s = wave_sum(acc[k]);
      if (lane == 0) {
        if (b) s += b[k];
        const float y = 1.f / (1.f + expf(-s));
        y_a[row * sa + k] = y;
        if (y_b) y_b[row * sb + k] = y;
      }
    }
  }
}

__global__ __launch_bounds__(256) void predictor_bwd_kernel(const float* __restrict__ dy_a, long long sa, const float* __restrict__ dy_b,
                                                            long long sb, const float* __restrict__ y, long long sy,
                                                            const float* __restrict__ W, float* __restrict__ dz, float* __restrict__ dh,
                                                            int rows, int C, int out_ch) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  float g[MAX_IN];
#pragma unroll
  for (int k = 0; k < MAX_IN; ++k) {
    g[k] = 0.f;
    if (k < out_ch) {
      float d = dy_a[row * sa + k];
      if (dy_b) d += dy_b[row * sb + k];
      const float yy = y[row * sy + k];
      g[k] = d * yy * (1.f - yy);
      if (lane == 0) dz[(long long)row * out_ch + k] = g[k];
    }
  }
  for (int c = lane; c < C; c += 64) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < MAX_IN; ++k) if (k < out_ch) acc = fmaf(g[k], W[(long long)k * C + c], acc);
    dh[(long long)row * C + c] = acc;
  }
}

// periodic distance e = min(|a-b|, |a+1-b|, |a-1-b|) ; loss += inv_2bt * e^2 ; d/da = inv_2bt * 2 e * sign(arg)
__global__ __launch_bounds__(256) void mtio_loss_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long long n,
                                                        float inv_2bt, double* __restrict__ accum, float* __restrict__ dpred) {
  __shared__ double part[4];
  double local = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = pred[i], b = gt[i];
    const float d0 = a - b, d1 = a + 1.f - b, d2 = a - 1.f - b;
    float e = fabsf(d0), arg = d0;
    // torch.minimum keeps the first operand on ties: strict < below
    if (fabsf(d1) < e) { e = fabsf(d1); arg = d1; }
    if (fabsf(d2) < e) { e = fabsf(d2); arg = d2; }
    local += (double)(e * e);
    if (dpred) dpred[i] = inv_2bt * 2.f * e * (arg > 0.f ? 1.f : (arg < 0.f ? -1.f : 0.f));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = local;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(accum, (part[0] + part[1]) + (part[2] + part[3]));
}
__global__ void mtio_loss_finish(const double* accum, float inv_2bt, float* loss_out) { *loss_out = (float)(*accum * (double)inv_2bt); }

// decoupled (AdamW, torch single-tensor math) or L2-coupled (Adam weight_decay) update, float4 per thread
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float sqrt_bc2, int decoupled) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  const int cnt = (int)min((long long)4, n - i4);
  float pp[4], gg[4], mm[4], vv[4];
  if (cnt == 4) {
    *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(p + i4);
    *reinterpret_cast<float4*>(gg) = *reinterpret_cast<const float4*>(g + i4);
    *reinterpret_cast<float4*>(mm) = *reinterpret_cast<const float4*>(m + i4);
    *reinterpret_cast<float4*>(vv) = *reinterpret_cast<const float4*>(v + i4);
  } else {
    for (int e = 0; e < cnt; ++e) { pp[e] = p[i4 + e]; gg[e] = g[i4 + e]; mm[e] = m[i4 + e]; vv[e] = v[i4 + e]; }
  }
  const float step_size = lr / bc1;
  for (int e = 0; e < cnt; ++e) {
    float grad = gg[e];
    if (decoupled) pp[e] = pp[e] * (1.f - lr * wd);
    else grad = grad + wd * pp[e];
    mm[e] = mm[e] + (grad - mm[e]) * (1.f - b1);            // torch: exp_avg.lerp_(grad, 1-beta1)
    vv[e] = vv[e] * b2 + (1.f - b2) * grad * grad;          // torch: mul_(beta2).addcmul_(grad, grad, 1-beta2)
    const float denom = sqrtf(vv[e]) / sqrt_bc2 + eps;
    pp[e] = pp[e] - step_size * (mm[e] / denom);
  }
  if (cnt == 4) {
    *reinterpret_cast<float4*>(p + i4) = *reinterpret_cast<float4*>(pp);
    *reinterpret_cast<float4*>(m + i4) = *reinterpret_cast<float4*>(mm);
    *reinterpret_cast<float4*>(v + i4) = *reinterpret_cast<float4*>(vv);
  } else {
    for (int e = 0; e < cnt; ++e) { p[i4 + e] = pp[e]; m[i4 + e] = mm[e]; v[i4 + e] = vv[e]; }
  }
}

__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int S, int C) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // over [B*S, C*3]
  const long long total = (long long)B * S * C * 3;
  if (idx >= total) return;
  const int kk = (int)(idx % (3 * C));
  const long long row = idx / (3 * C);
  const int ci = kk / 3, t = kk % 3;
  const int s = (int)(row % S);
  const long long b = row / S;
  int sp = s + t - 1; if (sp < 0) sp += S; if (sp >= S) sp -= S;
  col[idx] = x[(b * S + sp) * C + ci];
}
__global__ __launch_bounds__(256) void col2im3_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int S, int C) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // over [B*S, C]
  const long long total = (long long)B * S * C;
  if (idx >= total) return;
  const int ci = (int)(idx % C);
  const long long row = idx / C;
  const int s = (int)(row % S);
  const long long b = row / S;
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    int so = s - t + 1; if (so < 0) so += S; if (so >= S) so -= S;       // output position that read x[s] with tap t
    acc += dcol[(b * S + so) * (3LL * C) + ci * 3 + t];
  }
  dx[idx] = acc;
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ld, int rows, int C, float* __restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) s += x[(long long)r * ld + c];
  atomicAdd(out + c, s);
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = a[i] + b[i];
}

__global__ __launch_bounds__(256) void mtio_mix_kernel(const float* __restrict__ x, const int* __restrict__ perm1,
                                                       const int* __restrict__ perm2, float* __restrict__ out, int B, int L, int c) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // over [B, L, 3c]
  const long long total = (long long)B * L * 3 * c;
  if (idx >= total) return;
  const int ch = (int)(idx % (3 * c));
  const int l = (int)((idx / (3 * c)) % L);
  const int b = (int)(idx / ((long long)3 * c * L));
  const int k = ch / c, j = ch % c;
  int src = b;
  if (k == 1 && perm1) src = perm1[b];
  if (k == 2 && perm2) src = perm2[b];
  out[idx] = x[((long long)src * L + l) * c + j];
}

__global__ __launch_bounds__(256) void ensemble_wrap_kernel(const float* __restrict__ pred, float* __restrict__ out, long long rows,
                                                            int heads, int c) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // over [rows, c]
  if (idx >= rows * c) return;
  const int j = (int)(idx % c);
  const long long r = idx / c;
  float s = 0.f;
  for (int k = 0; k < heads; ++k) s += pred[r * heads * c + k * c + j];   // torch.sum over the gathered heads, in order
  float v = s / (float)heads;
  if (v < 0.f) v = v - (float)(int)v + 1.f;
  else if (v > 1.f) v = v - (float)(int)v;
  out[idx] = v;
}

__global__ __launch_bounds__(256) void tb_to_bt_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int B, int C) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // dst index over [B,T,C]
  if (idx >= (long long)T * B * C) return;
  const int c = (int)(idx % C);
  const int t = (int)((idx / C) % T);
  const long long b = idx / ((long long)C * T);
  dst[idx] = src[((long long)t * B + b) * C + c];
}

// sliding-window gather (load_dataset.py:43-52): table [n_trace, L, c]; idx [B,2] = (trace slot, timestep)
__global__ __launch_bounds__(256) void traj_gather_kernel(const float* __restrict__ table, int L, int c, const int* __restrict__ idx, int B, int S,
                                                          int T, float* __restrict__ hist, float* __restrict__ cur, float* __restrict__ fut) {
  const int W = S + 1 + T;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * W * c) return;
  const int j = (int)(t % c);
  const int w = (int)((t / c) % W);
  const int b = (int)(t / ((long long)c * W));
  const int slot = idx[2 * b], ts = idx[2 * b + 1];
  const float v = table[((long long)slot * L + (ts - S + w)) * c + j];
  if (w < S) hist[((long long)b * S + w) * c + j] = v;
  else if (w == S) cur[(long long)b * c + j] = v;
  else fut[((long long)b * T + (w - S - 1)) * c + j] = v;
}

// utils/common.py:73-80 per row: sum_j min(|a-b|,|a+1-b|,|a-1-b|)^2 / c
__global__ __launch_bounds__(256) void periodic_mse_kernel(const float* __restrict__ a, const float* __restrict__ b, long long rows, int c,
                                                           float* __restrict__ out) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int j = 0; j < c; ++j) {
    const float x = a[r * c + j], y = b[r * c + j];
    float e = fabsf(x - y);
    e = fminf(e, fabsf(x + 1.f - y));
    e = fminf(e, fabsf(x - 1.f - y));
    s += e * e;
  }
  out[r] = s / (float)c;
}

inline dim3 g1(long long n) { return dim3(mansy_ceil_div(n, 256)); }

}  // namespace

int mansy_launch_traj_gather(const float* table, int L, int c, const int* idx, int B, int S, int T, float* hist, float* cur, float* fut,
                             hipStream_t st) {
  MANSY_REQUIRE(table && idx && hist && cur && fut && L >= S + 1 + T, "traj_gather: bad arguments");
  if (B <= 0) return MANSY_OK;
  hipLaunchKernelGGL(traj_gather_kernel, g1((long long)B * (S + 1 + T) * c), dim3(256), 0, st, table, L, c, idx, B, S, T, hist, cur, fut);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_periodic_mse(const float* a, const float* b, long long rows, int c, float* out, hipStream_t st) {
  MANSY_REQUIRE(a && b && out && c >= 1, "periodic_mse: bad arguments");
  if (rows <= 0) return MANSY_OK;
  hipLaunchKernelGGL(periodic_mse_kernel, g1(rows), dim3(256), 0, st, a, b, rows, c, out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_embed_fwd(const float* x, int in_ch, const float* W, const float* b, const float* pe, float* out, int rows,
                           int C, int S, int pos_fixed, MansyDrop drop, hipStream_t st) {
  MANSY_REQUIRE(x && W && pe && out, "embed_fwd: null pointer");
  MANSY_REQUIRE(in_ch >= 1 && in_ch <= MAX_IN, "embed_fwd: in_ch %d unsupported", in_ch);
  if (rows <= 0) return MANSY_OK;
  hipLaunchKernelGGL(embed_fwd_kernel, g1((long long)rows * C), dim3(256), 0, st, x, in_ch, W, b, pe, out, rows, C, S, pos_fixed, drop);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_embed_bwd(const float* dX, const float* W, float* dE, float* dtok, int in_ch, int rows, int C, MansyDrop drop,
                           hipStream_t st) {
  MANSY_REQUIRE(dX && W, "embed_bwd: null pointer");
  MANSY_REQUIRE(in_ch >= 1 && in_ch <= MAX_IN, "embed_bwd: in_ch %d unsupported", in_ch);
  if (rows <= 0) return MANSY_OK;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(mansy_ceil_div(rows, 4)), dim3(256), 0, st, dX, W, dE, dtok, in_ch, rows, C, drop);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_outer_reduce(const float* small_, int small_n, const float* big, int rows, int C, float* out, int c_major,
                              float* bsum_big, float* bsum_small, hipStream_t st) {
  MANSY_REQUIRE(small_ && big && out, "outer_reduce: null pointer");
  MANSY_REQUIRE(small_n >= 1 && small_n <= MAX_IN, "outer_reduce: small_n %d unsupported", small_n);
  if (rows <= 0) return MANSY_OK;
  dim3 grid(mansy_ceil_div(C, 256), min(rows, 256));
  hipLaunchKernelGGL(outer_reduce_kernel, grid, dim3(256), 0, st, small_, small_n, big, rows, C, out, c_major, bsum_big, bsum_small);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_predictor_fwd(const float* h, const float* W, const float* b, float* y_a, long long ya_stride, float* y_b,
                               long long yb_stride, int rows, int C, int out_ch, hipStream_t st) {
  MANSY_REQUIRE(h && W && y_a, "predictor_fwd: null pointer");
  MANSY_REQUIRE(out_ch >= 1 && out_ch <= MAX_IN, "predictor_fwd: out_ch %d unsupported", out_ch);
  if (rows <= 0) return MANSY_OK;
  hipLaunchKernelGGL(predictor_fwd_kernel, dim3(mansy_ceil_div(rows, 4)), dim3(256), 0, st, h, W, b, y_a, ya_stride, y_b, yb_stride,
                     rows, C, out_ch);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_predictor_bwd(const float* dy_a, long long sa, const float* dy_b, long long sb, const float* y, long long sy,
                               const float* W, float* dz, float* dh, int rows, int C, int out_ch, hipStream_t st) {
  MANSY_REQUIRE(dy_a && y && W && dz && dh, "predictor_bwd: null pointer");
  MANSY_REQUIRE(out_ch >= 1 && out_ch <= MAX_IN, "predictor_bwd: out_ch %d unsupported", out_ch);
  if (rows <= 0) return MANSY_OK;
  hipLaunchKernelGGL(predictor_bwd_kernel, dim3(mansy_ceil_div(rows, 4)), dim3(256), 0, st, dy_a, sa, dy_b, sb, y, sy, W, dz, dh,
                     rows, C, out_ch);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_mtio_loss(const float* pred, const float* gt, long long n, float inv_2bt, double* loss_accum, float* loss_out,
                           float* dpred, hipStream_t st) {
  MANSY_REQUIRE(pred && gt && loss_accum && loss_out, "mtio_loss: null pointer");
  MANSY_HIP_CHECK(hipMemsetAsync(loss_accum, 0, sizeof(double), st));
  if (n > 0) {
    const int grid = min(mansy_ceil_div(n, 256), 1024);
    hipLaunchKernelGGL(mtio_loss_kernel, dim3(grid), dim3(256), 0, st, pred, gt, n, inv_2bt, loss_accum, dpred);
  }
  hipLaunchKernelGGL(mtio_loss_finish, dim3(1), dim3(1), 0, st, loss_accum, inv_2bt, loss_out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                       float wd, int step, int decoupled, hipStream_t st) {
  MANSY_REQUIRE(p && g && m && v, "adamw: null pointer");
  MANSY_REQUIRE(step >= 1, "adamw: step must be >= 1");
  if (n <= 0) return MANSY_OK;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  hipLaunchKernelGGL(adamw_kernel, g1((n + 3) / 4), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1,
                     (float)sqrt(bc2), decoupled);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_im2col3(const float* x, float* col, int B, int S, int C, hipStream_t st) {
  MANSY_REQUIRE(x && col, "im2col3: null pointer");
  const long long total = (long long)B * S * C * 3;
  if (total <= 0) return MANSY_OK;
  hipLaunchKernelGGL(im2col3_kernel, g1(total), dim3(256), 0, st, x, col, B, S, C);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_col2im3(const float* dcol, float* dx, int B, int S, int C, hipStream_t st) {
  MANSY_REQUIRE(dcol && dx, "col2im3: null pointer");
  const long long total = (long long)B * S * C;
  if (total <= 0) return MANSY_OK;
  hipLaunchKernelGGL(col2im3_kernel, g1(total), dim3(256), 0, st, dcol, dx, B, S, C);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_colsum(const float* x, int ld, int rows, int C, float* out, hipStream_t st) {
  MANSY_REQUIRE(x && out, "colsum: null pointer");
  if (rows <= 0) return MANSY_OK;
  dim3 grid(mansy_ceil_div(C, 256), min(rows, 256));
  hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, st, x, ld, rows, C, out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_add(const float* a, const float* b, float* y, long long n, hipStream_t st) {
  MANSY_REQUIRE(a && b && y, "add: null pointer");
  if (n <= 0) return MANSY_OK;
  hipLaunchKernelGGL(add_kernel, g1(n), dim3(256), 0, st, a, b, y, n);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_mtio_mix(const float* x, const int* perm1, const int* perm2, float* out, int B, int L, int c, hipStream_t st) {
  MANSY_REQUIRE(x && out, "mtio_mix: null pointer");
  const long long total = (long long)B * L * 3 * c;
  if (total <= 0) return MANSY_OK;
  hipLaunchKernelGGL(mtio_mix_kernel, g1(total), dim3(256), 0, st, x, perm1, perm2, out, B, L, c);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_ensemble_wrap(const float* pred, float* out, long long rows, int heads, int c, hipStream_t st) {
  MANSY_REQUIRE(pred && out, "ensemble_wrap: null pointer");
  if (rows <= 0) return MANSY_OK;
  hipLaunchKernelGGL(ensemble_wrap_kernel, g1(rows * c), dim3(256), 0, st, pred, out, rows, heads, c);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_tb_to_bt(const float* src, float* dst, int T, int B, int C, hipStream_t st) {
  MANSY_REQUIRE(src && dst, "tb_to_bt: null pointer");
  const long long total = (long long)T * B * C;
  if (total <= 0) return MANSY_OK;
  hipLaunchKernelGGL(tb_to_bt_kernel, g1(total), dim3(256), 0, st, src, dst, T, B, C);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
