// HBM-bound glue kernels of the viewport-prediction path: ViewportEmbedding + PositionalEncoding
// (mtio.py:10-44), predictor Linear(d->6)+Sigmoid (mtio.py:60), MTIO loss (mtio.py:94-104,
// utils/common.py:73-80), MTIO channel mix (mtio.py:72-90), ensemble mean + wrap
// (mtio.py:125-133, utils/common.py:61-70), im2col for the circular conv, AdamW
// (run_models.py:29: torch.optim.AdamW defaults) and Adam-with-L2 (run_mansy.py:216).
#include "mansy_kernels.h"

namespace {

constexpr int MAX_IN = 8;   // in_channel * num_head = 6 in the reference
// The small dimension (in_channel = 2, in_channel * num_head = 6) is a template parameter NS where it is one of the
// reference's values, so its loops unroll without predicates; NS = 0 is the generic form: MAX_IN iterations, loads at a
// clamped index and a 0/1 weight.  (A runtime `if (k < n)` around each load makes hipcc branch and wait vmcnt(0) per load.)
template <int NS> __device__ __forceinline__ constexpr int small_bound() { return NS > 0 ? NS : MAX_IN; }
template <int NS> __device__ __forceinline__ int small_idx(int k, int n) { return NS > 0 ? k : min(k, n - 1); }
template <int NS> __device__ __forceinline__ float small_on(int k, int n) { return (NS > 0 || k < n) ? 1.f : 0.f; }

// V consecutive channels per thread (V = 4: 16-byte stores; needs C % 4 == 0 and 16-byte aligned pe / out)
template <int V, int NS>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const float* __restrict__ x, int in_ch, const float* __restrict__ W,
                                                        const float* __restrict__ b, const float* __restrict__ pe,
                                                        float* __restrict__ out, int rows, int C, int S, int pos_fixed,
                                                        MansyDrop drop, unsigned short* __restrict__ out16) {
  const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
  if (idx >= (long long)rows * C) return;
  const int c = (int)(idx % C);
  const int r = (int)(idx / C);
  const int pos = pos_fixed >= 0 ? pos_fixed : (r % S);
  constexpr int NK = small_bound<NS>();
  float xin[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) xin[k] = x[(long long)r * in_ch + small_idx<NS>(k, in_ch)] * small_on<NS>(k, in_ch);
  float acc[V], pev[V], bv[V];
  if (V == 4) *reinterpret_cast<float4*>(pev) = *reinterpret_cast<const float4*>(pe + (long long)pos * C + c);
  else pev[0] = pe[(long long)pos * C + c];
#pragma unroll
  for (int j = 0; j < V; ++j) bv[j] = b ? b[c + j] : 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) a = fmaf(xin[k], W[(c + j) * in_ch + small_idx<NS>(k, in_ch)], a);
    a += bv[j];
    a += pev[j];
    if (drop.p > 0.f) a = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(idx + j), drop.p) ? a * (1.f / (1.f - drop.p)) : 0.f;
    acc[j] = a;
  }
  if (V == 4) *reinterpret_cast<float4*>(out + idx) = *reinterpret_cast<const float4*>(acc);
  else out[idx] = acc[0];
  if (out16) { if (V == 4) mansy_st_bf16x4(out16 + idx, acc[0], acc[1], acc[2], acc[3]); else mansy_st_bf16(out16 + idx, acc[0]); }
}

// Many-row form (encoder: rows = B*S): a thread keeps its 4 channels' weights / bias in registers and walks rows, so a
// row costs it the in_ch input loads (one cache line for the whole row), one positional float4 and one store -- the
// per-element form above re-loads the 4 x in_ch weights for every output (35 load instructions per 4 outputs).
// Needs C % 4 == 0 and (C/4) dividing 256; grid.x workgroups take contiguous row ranges.
template <int NS>
__global__ __launch_bounds__(256) void embed_fwd_rows_kernel(const float* __restrict__ x, int in_ch, const float* __restrict__ W,
                                                             const float* __restrict__ b, const float* __restrict__ pe,
                                                             float* __restrict__ out, int rows, int C, int S, int pos_fixed,
                                                             MansyDrop drop, int rows_per_wg, unsigned short* __restrict__ out16) {
  constexpr int NK = small_bound<NS>();
  const int C4 = C >> 2, rpi = 256 / C4;                 // rows per iteration of the workgroup
  const int c = (threadIdx.x % C4) * 4, rsub = threadIdx.x / C4;
  float wv[4][NK], bv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    bv[j] = b ? b[c + j] : 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) wv[j][k] = W[(c + j) * in_ch + small_idx<NS>(k, in_ch)] * small_on<NS>(k, in_ch);
  }
  const float dsc = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  const int r_begin = blockIdx.x * rows_per_wg, r_end = min(rows, r_begin + rows_per_wg);
  for (int r = r_begin + rsub; r < r_end; r += rpi) {
    const int pos = pos_fixed >= 0 ? pos_fixed : (r % S);
    float xin[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) xin[k] = x[(long long)r * in_ch + small_idx<NS>(k, in_ch)];
    const float4 pv = *reinterpret_cast<const float4*>(pe + (long long)pos * C + c);
    const float pev[4] = {pv.x, pv.y, pv.z, pv.w};
    const long long idx = (long long)r * C + c;
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < NK; ++k) a = fmaf(xin[k], wv[j][k], a);
      a += bv[j];
      a += pev[j];
      if (drop.p > 0.f) a = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(idx + j), drop.p) ? a * dsc : 0.f;
      acc[j] = a;
    }
    *reinterpret_cast<float4*>(out + idx) = *reinterpret_cast<const float4*>(acc);
    if (out16) mansy_st_bf16x4(out16 + idx, acc[0], acc[1], acc[2], acc[3]);
  }
}

// one wave per row: dE = dX*mask ; dtok[r,k] = sum_c dE[r,c] W[c,k]
template <int V, int NS>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dX, const float* __restrict__ W,
                                                        float* __restrict__ dE, float* __restrict__ dtok, int in_ch, int rows,
                                                        int C, MansyDrop drop) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float dsc = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  constexpr int NK = small_bound<NS>();
  float acc[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) acc[k] = 0.f;
  for (int c = lane * V; c < C; c += 64 * V) {
    const long long idx = (long long)row * C + c;
    float g[V];
    if (V == 4) *reinterpret_cast<float4*>(g) = *reinterpret_cast<const float4*>(dX + idx);
    else g[0] = dX[idx];
#pragma unroll
    for (int j = 0; j < V; ++j)
      if (drop.p > 0.f) g[j] = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(idx + j), drop.p) ? g[j] * dsc : 0.f;
    if (dE) {
      if (V == 4) *reinterpret_cast<float4*>(dE + idx) = *reinterpret_cast<const float4*>(g);
      else dE[idx] = g[0];
    }
    if (dtok) {      // (the encoder side has no token gradient: the pass is then a masked copy)
#pragma unroll
      for (int j = 0; j < V; ++j)
#pragma unroll
        for (int k = 0; k < NK; ++k) acc[k] = fmaf(g[j], W[(c + j) * in_ch + small_idx<NS>(k, in_ch)], acc[k]);
    }
  }
  if (dtok) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float t = wave_sum(acc[k]);
      if (lane == 0 && (NS > 0 || k < in_ch)) dtok[(long long)row * in_ch + k] = t;
    }
  }
}

// out += sum_r small[r,k] * big[r,c]  (the dW of the K<=8 linears: embedding, predictor) + optional column / small sums.
// grid (ceil(C/256), row-chunks): a wave owns 256 columns (float4 per lane, V = 4) or 64 (V = 1) and every 4*gridDim.y-th
// row; rows are unrolled by 4 so that four 1-KiB row pieces are in flight per wave; the 4 waves of a workgroup are combined
// through LDS before the atomics.
template <int V, int NS>
__global__ __launch_bounds__(256) void outer_reduce_kernel(const float* __restrict__ small_, int small_n,
                                                           const float* __restrict__ big, int rows, int C,
                                                           float* __restrict__ out, int c_major, float* __restrict__ bsum_big,
                                                           float* __restrict__ bsum_small) {
  constexpr int NK = small_bound<NS>();
  __shared__ float red[4][NK + 1][64 * V];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + lane) * V;
  const bool live = c < C;
  float acc[NK][V], sb[V];
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) acc[k][j] = 0.f;
#pragma unroll
  for (int j = 0; j < V; ++j) sb[j] = 0.f;
  constexpr int U = 4;                                   // rows in flight per wave (8: 57 -> 65 us on the [40 960-row] calls)
  const int rstep = gridDim.y * 4;
  for (int r0 = blockIdx.y * 4 + wave; r0 < rows; r0 += U * rstep) {
    float v[U][V];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = r0 + u * rstep;
      if (live && r < rows) {
        if (V == 4) *reinterpret_cast<float4*>(v[u]) = *reinterpret_cast<const float4*>(big + (long long)r * C + c);
        else v[u][0] = big[(long long)r * C + c];
      } else {
#pragma unroll
        for (int j = 0; j < V; ++j) v[u][j] = 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = min(r0 + u * rstep, rows - 1);      // v[u] is zero beyond the last row
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const float sv = small_[(long long)r * small_n + small_idx<NS>(k, small_n)];
#pragma unroll
        for (int j = 0; j < V; ++j) acc[k][j] = fmaf(sv, v[u][j], acc[k][j]);
      }
#pragma unroll
      for (int j = 0; j < V; ++j) sb[j] += v[u][j];
    }
  }
#pragma unroll
  for (int k = 0; k < NK; ++k)
#pragma unroll
    for (int j = 0; j < V; ++j) red[wave][k][lane * V + j] = acc[k][j];
#pragma unroll
  for (int j = 0; j < V; ++j) red[wave][NK][lane * V + j] = sb[j];
  __syncthreads();
  // 256 threads over (NK + 1) x 64*V sums
  for (int i = threadIdx.x; i < (NK + 1) * 64 * V; i += 256) {
    const int k = i / (64 * V), cc = i % (64 * V);
    const int col = blockIdx.x * 64 * V + cc;
    if (col >= C) continue;
    const float t = red[0][k][cc] + red[1][k][cc] + red[2][k][cc] + red[3][k][cc];
    if (k < small_n) atomicAdd(out + (c_major ? (long long)col * small_n + k : (long long)k * C + col), t);
    else if (k == NK && bsum_big) atomicAdd(bsum_big + col, t);
  }
  if (bsum_small && blockIdx.x == 0 && threadIdx.x < small_n) {
    float t = 0.f;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) t += small_[(long long)r * small_n + threadIdx.x];
    atomicAdd(bsum_small + threadIdx.x, t);
  }
}

template <int V, int NS>
__global__ __launch_bounds__(256) void predictor_fwd_kernel(const float* __restrict__ h, const float* __restrict__ W,
                                                            const float* __restrict__ b, float* __restrict__ y_a, long long sa,
                                                            float* __restrict__ y_b, long long sb, int rows, int C, int out_ch) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  constexpr int NK = small_bound<NS>();
  float acc[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) acc[k] = 0.f;
  for (int c = lane * V; c < C; c += 64 * V) {
    float v[V], w[NK][V];
    if (V == 4) *reinterpret_cast<float4*>(v) = *reinterpret_cast<const float4*>(h + (long long)row * C + c);
    else v[0] = h[(long long)row * C + c];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float* wr = W + (long long)small_idx<NS>(k, out_ch) * C + c;
      if (V == 4) *reinterpret_cast<float4*>(w[k]) = *reinterpret_cast<const float4*>(wr);
      else w[k][0] = wr[0];
    }
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[k] = fmaf(v[j], w[k][j], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    float t = wave_sum(acc[k]);
    if (lane == 0 && (NS > 0 || k < out_ch)) {
      if (b) t += b[k];
      const float y = 1.f / (1.f + expf(-t));
      y_a[row * sa + k] = y;
      if (y_b) y_b[row * sb + k] = y;
    }
  }
}

template <int V, int NS>
__global__ __launch_bounds__(256) void predictor_bwd_kernel(const float* __restrict__ dy_a, long long sa, const float* __restrict__ dy_b,
                                                            long long sb, const float* __restrict__ y, long long sy,
                                                            const float* __restrict__ W, float* __restrict__ dz, float* __restrict__ dh,
                                                            int rows, int C, int out_ch) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  constexpr int NK = small_bound<NS>();
  float g[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    const int kc = small_idx<NS>(k, out_ch);
    float d = dy_a[row * sa + kc];
    if (dy_b) d += dy_b[row * sb + kc];
    const float yy = y[row * sy + kc];
    g[k] = d * yy * (1.f - yy) * small_on<NS>(k, out_ch);
    if (lane == 0 && (NS > 0 || k < out_ch)) dz[(long long)row * out_ch + k] = g[k];
  }
  for (int c = lane * V; c < C; c += 64 * V) {
    float acc[V], w[NK][V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float* wr = W + (long long)small_idx<NS>(k, out_ch) * C + c;
      if (V == 4) *reinterpret_cast<float4*>(w[k]) = *reinterpret_cast<const float4*>(wr);
      else w[k][0] = wr[0];
    }
#pragma unroll
    for (int k = 0; k < NK; ++k)
#pragma unroll
      for (int j = 0; j < V; ++j) acc[j] = fmaf(g[k], w[k][j], acc[j]);
    if (V == 4) *reinterpret_cast<float4*>(dh + (long long)row * C + c) = *reinterpret_cast<const float4*>(acc);
    else dh[(long long)row * C + c] = acc[0];
  }
}

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
// one Adam element (torch.optim.Adam / AdamW, foreach=False arithmetic order)
__device__ __forceinline__ void adamw_elem(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps, float wd,
                                           float step_size, float sqrt_bc2, int decoupled) {
  if (decoupled) p = p * (1.f - lr * wd);
  else g = g + wd * p;
  m = m + (g - m) * (1.f - b1);               // torch: exp_avg.lerp_(grad, 1-beta1)
  v = v * b2 + (1.f - b2) * g * g;            // torch: mul_(beta2).addcmul_(grad, grad, 1-beta2)
  const float denom = sqrtf(v) / sqrt_bc2 + eps;
  p = p - step_size * (m / denom);
}
// (four elements per lane in named registers: indexing small per-lane arrays with a runtime count made the compiler place them in
//  LDS -- 16 KB per workgroup and a 29 us launch for 270 k elements, found in profiles/r02c_ppo_kernel_stats.csv)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                    float wd, float bc1, float sqrt_bc2, int decoupled, const float* __restrict__ bias) {
  const long long i4 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (bias) { bc1 = bias[0]; sqrt_bc2 = bias[1]; }      // graph-replayed steps: the bias corrections of THIS replay's step count, written by the host before the replay
  const float step_size = lr / bc1;
  if (i4 + 4 <= n) {
    float4 pp = *reinterpret_cast<const float4*>(p + i4);
    const float4 gg = *reinterpret_cast<const float4*>(g + i4);
    float4 mm = *reinterpret_cast<const float4*>(m + i4);
    float4 vv = *reinterpret_cast<const float4*>(v + i4);
    adamw_elem(pp.x, gg.x, mm.x, vv.x, lr, b1, b2, eps, wd, step_size, sqrt_bc2, decoupled);
    adamw_elem(pp.y, gg.y, mm.y, vv.y, lr, b1, b2, eps, wd, step_size, sqrt_bc2, decoupled);
    adamw_elem(pp.z, gg.z, mm.z, vv.z, lr, b1, b2, eps, wd, step_size, sqrt_bc2, decoupled);
    adamw_elem(pp.w, gg.w, mm.w, vv.w, lr, b1, b2, eps, wd, step_size, sqrt_bc2, decoupled);
    *reinterpret_cast<float4*>(p + i4) = pp;
    *reinterpret_cast<float4*>(m + i4) = mm;
    *reinterpret_cast<float4*>(v + i4) = vv;
    return;
  }
  for (long long e = i4; e < n; ++e) {          // the last, partial group of four
    float pe = p[e], me = m[e], ve = v[e];
    adamw_elem(pe, g[e], me, ve, lr, b1, b2, eps, wd, step_size, sqrt_bc2, decoupled);
    p[e] = pe; m[e] = me; v[e] = ve;
  }
}

// circular k=3 im2col for the DistillLayer conv: col[row, ci*3 + t] = x[(b, (s + t - 1) mod S), ci].
// V = 4: a thread takes 4 channels: three 16-byte row reads (s-1, s, s+1), three 16-byte stores (12 contiguous floats).
template <int V>
__global__ __launch_bounds__(256) void im2col3_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int S, int C, unsigned short* __restrict__ col16) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // over [B*S, C/V]
  const int CV = C / V;
  if (idx >= (long long)B * S * CV) return;
  const int ci = (int)(idx % CV) * V;
  const long long row = idx / CV;
  const int s = (int)(row % S);
  const long long b = row / S;
  const int sm = s == 0 ? S - 1 : s - 1, sp = s == S - 1 ? 0 : s + 1;
  float r0[V], r1[V], r2[V], o[3 * V];
  if (V == 4) {
    *reinterpret_cast<float4*>(r0) = *reinterpret_cast<const float4*>(x + (b * S + sm) * C + ci);
    *reinterpret_cast<float4*>(r1) = *reinterpret_cast<const float4*>(x + (b * S + s) * C + ci);
    *reinterpret_cast<float4*>(r2) = *reinterpret_cast<const float4*>(x + (b * S + sp) * C + ci);
  } else {
    r0[0] = x[(b * S + sm) * C + ci]; r1[0] = x[(b * S + s) * C + ci]; r2[0] = x[(b * S + sp) * C + ci];
  }
#pragma unroll
  for (int j = 0; j < V; ++j) { o[3 * j] = r0[j]; o[3 * j + 1] = r1[j]; o[3 * j + 2] = r2[j]; }
  float* dst = col + row * (3LL * C) + 3 * ci;
  if (V == 4) {
#pragma unroll
    for (int q = 0; q < 3; ++q) *reinterpret_cast<float4*>(dst + 4 * q) = *reinterpret_cast<const float4*>(o + 4 * q);
    if (col16) {
#pragma unroll
      for (int q = 0; q < 3; ++q) mansy_st_bf16x4(col16 + row * (3LL * C) + 3 * ci + 4 * q, o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
    }
  } else {
    dst[0] = o[0]; dst[1] = o[1]; dst[2] = o[2];
  }
}
// dx[(b,s), ci] = sum_t dcol[(b, (s - t + 1) mod S), ci*3 + t].  V = 4: the 12 contiguous floats of each of the three
// source rows are read as 3 float4 (a third of each is used; the other two thirds are the neighbours' and hit L2).
template <int V>
__global__ __launch_bounds__(256) void col2im3_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int S, int C) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // over [B*S, C/V]
  const int CV = C / V;
  if (idx >= (long long)B * S * CV) return;
  const int ci = (int)(idx % CV) * V;
  const long long row = idx / CV;
  const int s = (int)(row % S);
  const long long b = row / S;
  float acc[V];
#pragma unroll
  for (int j = 0; j < V; ++j) acc[j] = 0.f;
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    int so = s - t + 1; if (so < 0) so += S; if (so >= S) so -= S;       // output position that read x[s] with tap t
    const float* src = dcol + (b * S + so) * (3LL * C) + 3 * ci;
    float v[3 * V];
    if (V == 4) {
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<float4*>(v + 4 * q) = *reinterpret_cast<const float4*>(src + 4 * q);
    } else {
      v[0] = src[0]; v[1] = src[1]; v[2] = src[2];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] += v[3 * j + t];
  }
  if (V == 4) *reinterpret_cast<float4*>(dx + row * C + ci) = *reinterpret_cast<const float4*>(acc);
  else dx[row * C + ci] = acc[0];
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

// the three MTIO mixes of a train step (history, current, future: mtio.py:77-87) as ONE launch: segment s covers [B, L_s, 3c]
struct MtioMix3 { const float* x[3]; float* out[3]; int L[3]; long long end[3]; };
__global__ __launch_bounds__(256) void mtio_mix3_kernel(MtioMix3 a, const int* __restrict__ perm1, const int* __restrict__ perm2, int B, int c) {
  long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.end[2]) return;
  const int s = idx < a.end[0] ? 0 : (idx < a.end[1] ? 1 : 2);
  if (s) idx -= a.end[s - 1];
  const int L = a.L[s];
  const int ch = (int)(idx % (3 * c));
  const int l = (int)((idx / (3 * c)) % L);
  const int b = (int)(idx / ((long long)3 * c * L));
  const int k = ch / c, j = ch % c;
  int src = b;
  if (k == 1 && perm1) src = perm1[b];
  if (k == 2 && perm2) src = perm2[b];
  a.out[s][idx] = a.x[s][((long long)src * L + l) * c + j];
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

// LinearRegression.sample (viewport_prediction/models/linear_regression.py:18-36, `run_models.py --model regression`): one thread per
// (trajectory, coordinate) fits the least-squares line through the S + 1 past samples over t = 0..S and extrapolates T steps.
// scikit-learn's arithmetic, in float64 like it: centre t and y on their means, coef = <tc, yc> / <tc, tc>,
// intercept = y_mean - t_mean * coef, prediction = t * coef + intercept, rounded to float32 on the store (the reference assigns the
// float64 prediction into a float32 tensor).
__global__ __launch_bounds__(256) void linreg_sample_kernel(const float* __restrict__ hist, const float* __restrict__ cur, int B, int S, int T,
                                                            int c, float* __restrict__ out) {
#pragma clang fp contract(off)
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;     // over [B, c]
  if (idx >= (long long)B * c) return;
  const int j = (int)(idx % c);
  const long long b = idx / c;
  const int L = S + 1;
  const float* h = hist + b * S * c + j;
  const double last = (double)cur[b * c + j];
  double sy = 0.0;
  for (int l = 0; l < S; ++l) sy += (double)h[(long long)l * c];
  sy += last;
  const double y_mean = sy / (double)L, t_mean = (double)(L - 1) * 0.5;
  double sxy = 0.0, sxx = 0.0;
  for (int l = 0; l < L; ++l) {
    const double tc = (double)l - t_mean, yc = (l < S ? (double)h[(long long)l * c] : last) - y_mean;
    sxy += tc * yc; sxx += tc * tc;
  }
  const double coef = sxy / sxx, intercept = y_mean - t_mean * coef;
  for (int k = 0; k < T; ++k) out[(b * T + k) * c + j] = (float)((double)(L + k) * coef + intercept);
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

// c == 2 (the reference's (x, y) samples): one thread per SAMPLE, 8-byte loads and stores -- half the threads, twice the bytes per
// access of the per-float form (2.6 -> see profiles/r03_hbm_kernels_bench*.txt)
__global__ __launch_bounds__(256) void traj_gather2_kernel(const float2* __restrict__ table, int L, const int* __restrict__ idx, int B, int S, int T,
                                                           float2* __restrict__ hist, float2* __restrict__ cur, float2* __restrict__ fut) {
  const int W = S + 1 + T;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long long)B * W) return;
  const int b = (int)(t / W), w = (int)(t - (long long)b * W);
  const int slot = idx[2 * b], ts = idx[2 * b + 1];
  const float2 v = table[(long long)slot * L + (ts - S + w)];
  if (w < S) hist[(long long)b * S + w] = v;
  else if (w == S) cur[b] = v;
  else fut[(long long)b * T + (w - S - 1)] = v;
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
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// KERNEL<V, NS>: V = 4 when `vec`, NS = the small dimension when it is one of the reference's (2, 6), else generic
#define MANSY_SMALL_DISPATCH(KERNEL, vec, ns, grid_v4, grid_v1, ...)                                                          \
  do {                                                                                                                        \
    if (vec) {                                                                                                                \
      if ((ns) == 2) MANSY_LAUNCH((KERNEL<4, 2>), grid_v4, dim3(256), 0, st, __VA_ARGS__);                              \
      else if ((ns) == 6) MANSY_LAUNCH((KERNEL<4, 6>), grid_v4, dim3(256), 0, st, __VA_ARGS__);                         \
      else MANSY_LAUNCH((KERNEL<4, 0>), grid_v4, dim3(256), 0, st, __VA_ARGS__);                                        \
    } else {                                                                                                                  \
      if ((ns) == 2) MANSY_LAUNCH((KERNEL<1, 2>), grid_v1, dim3(256), 0, st, __VA_ARGS__);                              \
      else if ((ns) == 6) MANSY_LAUNCH((KERNEL<1, 6>), grid_v1, dim3(256), 0, st, __VA_ARGS__);                         \
      else MANSY_LAUNCH((KERNEL<1, 0>), grid_v1, dim3(256), 0, st, __VA_ARGS__);                                        \
    }                                                                                                                         \
  } while (0)

}  // namespace

int mansy_launch_traj_gather(const float* table, int L, int c, const int* idx, int B, int S, int T, float* hist, float* cur, float* fut,
                             hipStream_t st) {
  MANSY_REQUIRE(table && idx && hist && cur && fut && L >= S + 1 + T, "traj_gather: bad arguments");
  if (B <= 0) return MANSY_OK;
  auto al8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7) == 0; };
  if (c == 2 && al8(table) && al8(hist) && al8(cur) && al8(fut))
    MANSY_LAUNCH(traj_gather2_kernel, g1((long long)B * (S + 1 + T)), dim3(256), 0, st, reinterpret_cast<const float2*>(table), L, idx, B, S, T,
                       reinterpret_cast<float2*>(hist), reinterpret_cast<float2*>(cur), reinterpret_cast<float2*>(fut));
  else
    MANSY_LAUNCH(traj_gather_kernel, g1((long long)B * (S + 1 + T) * c), dim3(256), 0, st, table, L, c, idx, B, S, T, hist, cur, fut);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_periodic_mse(const float* a, const float* b, long long rows, int c, float* out, hipStream_t st) {
  MANSY_REQUIRE(a && b && out && c >= 1, "periodic_mse: bad arguments");
  if (rows <= 0) return MANSY_OK;
  MANSY_LAUNCH(periodic_mse_kernel, g1(rows), dim3(256), 0, st, a, b, rows, c, out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_embed_fwd(const float* x, int in_ch, const float* W, const float* b, const float* pe, float* out, int rows,
                           int C, int S, int pos_fixed, MansyDrop drop, hipStream_t st, unsigned short* out16) {
  MANSY_REQUIRE(x && W && pe && out, "embed_fwd: null pointer");
  MANSY_REQUIRE(in_ch >= 1 && in_ch <= MAX_IN, "embed_fwd: in_ch %d unsupported", in_ch);
  if (rows <= 0) return MANSY_OK;
  const bool vec = C % 4 == 0 && al16(pe) && al16(out);
  if (vec && rows >= 8192 && 256 % (C / 4) == 0) {
    const int rpw = 8 * (256 / (C / 4));                 // 8 iterations per workgroup
    const dim3 grid(mansy_ceil_div(rows, rpw));
    if (in_ch == 2) MANSY_LAUNCH(embed_fwd_rows_kernel<2>, grid, dim3(256), 0, st, x, in_ch, W, b, pe, out, rows, C, S, pos_fixed, drop, rpw, out16);
    else if (in_ch == 6) MANSY_LAUNCH(embed_fwd_rows_kernel<6>, grid, dim3(256), 0, st, x, in_ch, W, b, pe, out, rows, C, S, pos_fixed, drop, rpw, out16);
    else MANSY_LAUNCH(embed_fwd_rows_kernel<0>, grid, dim3(256), 0, st, x, in_ch, W, b, pe, out, rows, C, S, pos_fixed, drop, rpw, out16);
  } else {
    MANSY_SMALL_DISPATCH(embed_fwd_kernel, vec, in_ch, g1((long long)rows * C / 4), g1((long long)rows * C),
                         x, in_ch, W, b, pe, out, rows, C, S, pos_fixed, drop, out16);
  }
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_embed_bwd(const float* dX, const float* W, float* dE, float* dtok, int in_ch, int rows, int C, MansyDrop drop,
                           hipStream_t st) {
  MANSY_REQUIRE(dX && W, "embed_bwd: null pointer");
  MANSY_REQUIRE(in_ch >= 1 && in_ch <= MAX_IN, "embed_bwd: in_ch %d unsupported", in_ch);
  if (rows <= 0) return MANSY_OK;
  MANSY_SMALL_DISPATCH(embed_bwd_kernel, C % 4 == 0 && al16(dX) && (!dE || al16(dE)), in_ch, dim3(mansy_ceil_div(rows, 4)),
                       dim3(mansy_ceil_div(rows, 4)), dX, W, dE, dtok, in_ch, rows, C, drop);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_outer_reduce(const float* small_, int small_n, const float* big, int rows, int C, float* out, int c_major,
                              float* bsum_big, float* bsum_small, hipStream_t st) {
  MANSY_REQUIRE(small_ && big && out, "outer_reduce: null pointer");
  MANSY_REQUIRE(small_n >= 1 && small_n <= MAX_IN, "outer_reduce: small_n %d unsupported", small_n);
  if (rows <= 0) return MANSY_OK;
  // (measured, round 4: 64 chunks instead of 256 -- a quarter of the atomics per output element -- DOUBLED the [40 960-row] calls of the VP step,
  // 57 -> 116 us: the streaming, not the tail of float atomics, sets this kernel's time)
  const int chunks = max(1, min(mansy_ceil_div(rows, 16), 256));
  MANSY_SMALL_DISPATCH(outer_reduce_kernel, C % 4 == 0 && al16(big), small_n, dim3(mansy_ceil_div(C, 256), chunks),
                       dim3(mansy_ceil_div(C, 64), chunks), small_, small_n, big, rows, C, out, c_major, bsum_big, bsum_small);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_predictor_fwd(const float* h, const float* W, const float* b, float* y_a, long long ya_stride, float* y_b,
                               long long yb_stride, int rows, int C, int out_ch, hipStream_t st) {
  MANSY_REQUIRE(h && W && y_a, "predictor_fwd: null pointer");
  MANSY_REQUIRE(out_ch >= 1 && out_ch <= MAX_IN, "predictor_fwd: out_ch %d unsupported", out_ch);
  if (rows <= 0) return MANSY_OK;
  MANSY_SMALL_DISPATCH(predictor_fwd_kernel, C % 4 == 0 && al16(h) && al16(W), out_ch, dim3(mansy_ceil_div(rows, 4)),
                       dim3(mansy_ceil_div(rows, 4)), h, W, b, y_a, ya_stride, y_b, yb_stride, rows, C, out_ch);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_predictor_bwd(const float* dy_a, long long sa, const float* dy_b, long long sb, const float* y, long long sy,
                               const float* W, float* dz, float* dh, int rows, int C, int out_ch, hipStream_t st) {
  MANSY_REQUIRE(dy_a && y && W && dz && dh, "predictor_bwd: null pointer");
  MANSY_REQUIRE(out_ch >= 1 && out_ch <= MAX_IN, "predictor_bwd: out_ch %d unsupported", out_ch);
  if (rows <= 0) return MANSY_OK;
  MANSY_SMALL_DISPATCH(predictor_bwd_kernel, C % 4 == 0 && al16(W) && al16(dh), out_ch, dim3(mansy_ceil_div(rows, 4)),
                       dim3(mansy_ceil_div(rows, 4)), dy_a, sa, dy_b, sb, y, sy, W, dz, dh, rows, C, out_ch);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_mtio_loss(const float* pred, const float* gt, long long n, float inv_2bt, double* loss_accum, float* loss_out,
                           float* dpred, hipStream_t st) {
  MANSY_REQUIRE(pred && gt && loss_accum && loss_out, "mtio_loss: null pointer");
  MANSY_HIP_CHECK(hipMemsetAsync(loss_accum, 0, sizeof(double), st));
  if (n > 0) {
    const int grid = min(mansy_ceil_div(n, 256), 128);      // every workgroup ends with ONE double atomic on the same address: 128-way, not 1024-way (14 -> ~6 us)
    MANSY_LAUNCH(mtio_loss_kernel, dim3(grid), dim3(256), 0, st, pred, gt, n, inv_2bt, loss_accum, dpred);
  }
  MANSY_LAUNCH(mtio_loss_finish, dim3(1), dim3(1), 0, st, loss_accum, inv_2bt, loss_out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps,
                       float wd, int step, int decoupled, hipStream_t st, const float* bias_dev) {
  MANSY_REQUIRE(p && g && m && v, "adamw: null pointer");
  MANSY_REQUIRE(step >= 1, "adamw: step must be >= 1");
  if (n <= 0) return MANSY_OK;
  const double bc1 = 1.0 - pow((double)b1, (double)step);
  const double bc2 = 1.0 - pow((double)b2, (double)step);
  MANSY_LAUNCH(adamw_kernel, g1((n + 3) / 4), dim3(256), 0, st, p, g, m, v, n, lr, b1, b2, eps, wd, (float)bc1,
                     (float)sqrt(bc2), decoupled, bias_dev);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_im2col3(const float* x, float* col, int B, int S, int C, hipStream_t st, unsigned short* col16) {
  MANSY_REQUIRE(x && col, "im2col3: null pointer");
  if ((long long)B * S * C <= 0) return MANSY_OK;
  if (C % 4 == 0 && al16(x) && al16(col)) MANSY_LAUNCH(im2col3_kernel<4>, g1((long long)B * S * C / 4), dim3(256), 0, st, x, col, B, S, C, col16);
  else MANSY_LAUNCH(im2col3_kernel<1>, g1((long long)B * S * C), dim3(256), 0, st, x, col, B, S, C, (unsigned short*)nullptr);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_col2im3(const float* dcol, float* dx, int B, int S, int C, hipStream_t st) {
  MANSY_REQUIRE(dcol && dx, "col2im3: null pointer");
  const long long total = (long long)B * S * C;
  if (total <= 0) return MANSY_OK;
  if (C % 4 == 0 && al16(dcol) && al16(dx)) MANSY_LAUNCH(col2im3_kernel<4>, g1(total / 4), dim3(256), 0, st, dcol, dx, B, S, C);
  else MANSY_LAUNCH(col2im3_kernel<1>, g1(total), dim3(256), 0, st, dcol, dx, B, S, C);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_colsum(const float* x, int ld, int rows, int C, float* out, hipStream_t st) {
  MANSY_REQUIRE(x && out, "colsum: null pointer");
  if (rows <= 0) return MANSY_OK;
  dim3 grid(mansy_ceil_div(C, 256), min(rows, 256));
  MANSY_LAUNCH(colsum_kernel, grid, dim3(256), 0, st, x, ld, rows, C, out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_add(const float* a, const float* b, float* y, long long n, hipStream_t st) {
  MANSY_REQUIRE(a && b && y, "add: null pointer");
  if (n <= 0) return MANSY_OK;
  MANSY_LAUNCH(add_kernel, g1(n), dim3(256), 0, st, a, b, y, n);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_mtio_mix(const float* x, const int* perm1, const int* perm2, float* out, int B, int L, int c, hipStream_t st) {
  MANSY_REQUIRE(x && out, "mtio_mix: null pointer");
  const long long total = (long long)B * L * 3 * c;
  if (total <= 0) return MANSY_OK;
  MANSY_LAUNCH(mtio_mix_kernel, g1(total), dim3(256), 0, st, x, perm1, perm2, out, B, L, c);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_mtio_mix3(const float* hist, const float* cur, const float* fut, const int* perm1, const int* perm2, float* src6, float* cur6, float* fut6,
                           int B, int S, int T, int c, hipStream_t st) {
  MANSY_REQUIRE(hist && cur && fut && src6 && cur6 && fut6, "mtio_mix3: null pointer");
  MtioMix3 a;
  a.x[0] = hist; a.x[1] = cur; a.x[2] = fut; a.out[0] = src6; a.out[1] = cur6; a.out[2] = fut6; a.L[0] = S; a.L[1] = 1; a.L[2] = T;
  a.end[0] = (long long)B * S * 3 * c; a.end[1] = a.end[0] + (long long)B * 3 * c; a.end[2] = a.end[1] + (long long)B * T * 3 * c;
  if (a.end[2] <= 0) return MANSY_OK;
  MANSY_LAUNCH(mtio_mix3_kernel, g1(a.end[2]), dim3(256), 0, st, a, perm1, perm2, B, c);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_ensemble_wrap(const float* pred, float* out, long long rows, int heads, int c, hipStream_t st) {
  MANSY_REQUIRE(pred && out, "ensemble_wrap: null pointer");
  if (rows <= 0) return MANSY_OK;
  MANSY_LAUNCH(ensemble_wrap_kernel, g1(rows * c), dim3(256), 0, st, pred, out, rows, heads, c);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_linreg_sample(const float* hist, const float* cur, int B, int S, int T, int c, float* out, hipStream_t st) {
  MANSY_REQUIRE(S >= 1 && T >= 0 && c >= 1, "linreg_sample: need S >= 1 (two points fix a line), T >= 0, c >= 1");
  if (B <= 0 || T == 0) return MANSY_OK;
  MANSY_REQUIRE(hist && cur && out, "linreg_sample: null pointer");
  MANSY_LAUNCH(linreg_sample_kernel, g1((long long)B * c), dim3(256), 0, st, hist, cur, B, S, T, c, out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
int mansy_launch_tb_to_bt(const float* src, float* dst, int T, int B, int C, hipStream_t st) {
  MANSY_REQUIRE(src && dst, "tb_to_bt: null pointer");
  const long long total = (long long)T * B * C;
  if (total <= 0) return MANSY_OK;
  MANSY_LAUNCH(tb_to_bt_kernel, g1(total), dim3(256), 0, st, src, dst, T, B, C);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
