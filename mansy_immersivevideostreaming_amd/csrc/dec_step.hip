// Fused row-wise tail / head of one autoregressive decoder step of the viewport Transformer (mtio.py:150-166 +
// customized_transformer.py:79-83).  Between the last product of step i and the first product of step
// i+1 the reference runs four row-wise ops on [B, d] tensors:
//     LayerNorm3 of the last decoder layer -> final decoder LayerNorm -> predictor Linear(d->6)+Sigmoid -> embedding of the
//     fed-back token (+ positional row, dropout)
// and the mirror image in the backward pass.  Each is one wave per row here already; chained in one kernel they cost one
// launch instead of four and the intermediate rows never leave registers (they are still stored: the backward needs them).
// Arithmetic is the same, operation for operation, as layernorm_fwd_kernel / predictor_fwd_kernel / embed_fwd_kernel
// (norm.hip, elementwise.hip), so the fused and unfused paths agree bit for bit.
#include "mansy_kernels.h"

namespace {

constexpr int C6 = 6;            // in_channel * num_head of the reference model (other values take the unfused path)

template <int NV>
__global__ __launch_bounds__(256) void dec_tail_fwd_kernel(MansyDecTailFwd p) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= p.rows) return;
  const int C = p.C;
  const long long base = (long long)row * C + lane * 4;
  // ---- z3 = a + b ; y3 = LN3(z3)
  float4 v[NV], t[NV], w3[NV], b3[NV], wd[NV], bd[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (p.a16) { const mansy_bf16x4 u = *reinterpret_cast<const mansy_bf16x4*>(p.a16 + base + i * 256); v[i] = make_float4((float)u[0], (float)u[1], (float)u[2], (float)u[3]); }
    else v[i] = *reinterpret_cast<const float4*>(p.a + base + i * 256);
    t[i] = *reinterpret_cast<const float4*>(p.b + base + i * 256);
    w3[i] = *reinterpret_cast<const float4*>(p.n3_w + lane * 4 + i * 256);
    wd[i] = *reinterpret_cast<const float4*>(p.dn_w + lane * 4 + i * 256);
    b3[i] = make_float4(0.f, 0.f, 0.f, 0.f); bd[i] = b3[i];
  }
  if (p.n3_b) {            // (the bias-free LayerNorm layout of torch >= 2.1 checkpoints passes null)
#pragma unroll
    for (int i = 0; i < NV; ++i) b3[i] = *reinterpret_cast<const float4*>(p.n3_b + lane * 4 + i * 256);
  }
  if (p.dn_b) {
#pragma unroll
    for (int i = 0; i < NV; ++i) bd[i] = *reinterpret_cast<const float4*>(p.dn_b + lane * 4 + i * 256);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x += t[i].x; v[i].y += t[i].y; v[i].z += t[i].z; v[i].w += t[i].w;
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (!p.img_only) *reinterpret_cast<float4*>(p.z3 + base + i * 256) = v[i];
    if (p.z3_16) mansy_st_bf16x4(p.z3_16 + base + i * 256, v[i].x, v[i].y, v[i].z, v[i].w);
  }
  float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float dx = v[i].x - mu, dy = v[i].y - mu, dz = v[i].z - mu, dw = v[i].w - mu;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  float rs = 1.0f / sqrtf(wave_sum(q) / (float)C + p.eps);
  if (lane == 0) { p.m3[row] = mu; p.r3[row] = rs; }
  float4 y[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    y[i].x = (v[i].x - mu) * rs * w3[i].x + b3[i].x; y[i].y = (v[i].y - mu) * rs * w3[i].y + b3[i].y;
    y[i].z = (v[i].z - mu) * rs * w3[i].z + b3[i].z; y[i].w = (v[i].w - mu) * rs * w3[i].w + b3[i].w;
    if (!p.img_only) *reinterpret_cast<float4*>(p.y3 + base + i * 256) = y[i];
    if (p.y3_16) mansy_st_bf16x4(p.y3_16 + base + i * 256, y[i].x, y[i].y, y[i].z, y[i].w);
  }
  // ---- dec_out = LN_dec(y3)
  s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (y[i].x + y[i].y) + (y[i].z + y[i].w);
  mu = wave_sum(s) / (float)C;
  q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float dx = y[i].x - mu, dy = y[i].y - mu, dz = y[i].z - mu, dw = y[i].w - mu;
    q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
  }
  rs = 1.0f / sqrtf(wave_sum(q) / (float)C + p.eps);
  if (lane == 0) { p.md[row] = mu; p.rd[row] = rs; }
  float4 o[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    o[i].x = (y[i].x - mu) * rs * wd[i].x + bd[i].x; o[i].y = (y[i].y - mu) * rs * wd[i].y + bd[i].y;
    o[i].z = (y[i].z - mu) * rs * wd[i].z + bd[i].z; o[i].w = (y[i].w - mu) * rs * wd[i].w + bd[i].w;
    *reinterpret_cast<float4*>(p.dec_out + base + i * 256) = o[i];
  }
  // ---- predictor: tok[k] = sigmoid(dec_out . Wp[k] + bp[k])
  float acc[C6];
#pragma unroll
  for (int k = 0; k < C6; ++k) acc[k] = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float ov[4] = {o[i].x, o[i].y, o[i].z, o[i].w};
    float wk[C6][4];
#pragma unroll
    for (int k = 0; k < C6; ++k) *reinterpret_cast<float4*>(wk[k]) = *reinterpret_cast<const float4*>(p.pw + (long long)k * C + lane * 4 + i * 256);
#pragma unroll
    for (int k = 0; k < C6; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[k] = fmaf(ov[j], wk[k][j], acc[k]);
  }
  float tok[C6];
#pragma unroll
  for (int k = 0; k < C6; ++k) {
    float u = wave_sum(acc[k]);
    if (p.pb) u += p.pb[k];
    tok[k] = 1.f / (1.f + expf(-u));
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < C6; ++k) {
      p.tok_next[(long long)row * C6 + k] = tok[k];
      if (p.pred_bt) p.pred_bt[(long long)row * p.pred_stride + k] = tok[k];
    }
  }
  // ---- embedding of the fed-back token for step i+1
  if (p.emb_next) {
    const float dsc = p.edrop.p > 0.f ? 1.f / (1.f - p.edrop.p) : 1.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int c = lane * 4 + i * 256;
      float pev[4], ebv[4], e[4];
      *reinterpret_cast<float4*>(pev) = *reinterpret_cast<const float4*>(p.pe_row + c);
      if (p.eb) *reinterpret_cast<float4*>(ebv) = *reinterpret_cast<const float4*>(p.eb + c);
      else { ebv[0] = ebv[1] = ebv[2] = ebv[3] = 0.f; }
      float ew[4][C6];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < C6; ++k) ew[j][k] = p.ew[(c + j) * C6 + k];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < C6; ++k) a = fmaf(tok[k], ew[j][k], a);
        a += ebv[j];
        a += pev[j];
        if (p.edrop.p > 0.f) a = mansy_keep(p.edrop.seed, p.edrop.site, p.edrop.base + (uint32_t)((long long)row * C + c + j), p.edrop.p) ? a * dsc : 0.f;
        e[j] = a;
      }
      if (!p.img_only) *reinterpret_cast<float4*>(p.emb_next + base + i * 256) = *reinterpret_cast<const float4*>(e);
      if (p.emb_next16) mansy_st_bf16x4(p.emb_next16 + base + i * 256, e[0], e[1], e[2], e[3]);
    }
  }
}

// ---- backward mirror, run at the START of backward step i (steps go T-1 -> 0):
//   embedding backward of step i+1 (its input gradient gx_next is what the layer stack of step i+1 just produced):
//       dE_{i+1} = gx_next * dropmask ; dtok[k] = sum_c dE_{i+1}[c] We[c][k]          (skipped on the last step)
//   predictor backward:  g[k] = (dpred_i[k] + dtok[k]) * y (1 - y) ; dh[c] = sum_k g[k] Wp[k][c]
//   final decoder LayerNorm backward (dy = dh, z = y3 of the last layer) -> gradient wrt y3
//   LayerNorm3 backward of the last layer -> gz (residual-path gradient) and dbr3 = gz * dropmask (the lin2 branch)
// LayerNorm weight-gradient column sums go to per-workgroup slots like layernorm_bwd_vec_kernel<.., PART> (accumulated over
// the steps): gridDim.x must be the slot count.
constexpr int HEAD_WAVES = 8;     // 2 rows per wave at B = 4096 with the 256 LayerNorm slot workgroups: the per-row chain is ten dependent wave reductions
template <int NV>
__global__ __launch_bounds__(64 * HEAD_WAVES) void dec_head_bwd_kernel(MansyDecHeadBwd p) {
  extern __shared__ float red[];      // [HEAD_WAVES][4][C]: (dw, db) of the final norm, (dw, db) of LayerNorm3
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wave_global = blockIdx.x * HEAD_WAVES + wave;
  const int nwaves = gridDim.x * HEAD_WAVES;
  const int C = p.C;
  const float invC = 1.f / (float)C;
  const float esc = p.edrop.p > 0.f ? 1.f / (1.f - p.edrop.p) : 1.f;
  const float dsc = p.drop3.p > 0.f ? 1.f / (1.f - p.drop3.p) : 1.f;
  float4 wd[NV], w3[NV], adw_d[NV], adb_d[NV], adw_3[NV], adb_3[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    wd[i] = *reinterpret_cast<const float4*>(p.dn_w + lane * 4 + i * 256);
    w3[i] = *reinterpret_cast<const float4*>(p.n3_w + lane * 4 + i * 256);
    adw_d[i] = make_float4(0.f, 0.f, 0.f, 0.f); adb_d[i] = adw_d[i]; adw_3[i] = adw_d[i]; adb_3[i] = adw_d[i];
  }
  for (int row = wave_global; row < p.rows; row += nwaves) {
    const long long base = (long long)row * C + lane * 4;
    // every row-sized load of this row up front
    float4 gn[NV], zy[NV], zz[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      gn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.y3_16) {           // bf16 residual stream: the forward kept y3 / z3 as images only
        const mansy_bf16x4 u = *reinterpret_cast<const mansy_bf16x4*>(p.y3_16 + base + i * 256), w = *reinterpret_cast<const mansy_bf16x4*>(p.z3_16 + base + i * 256);
        zy[i] = make_float4((float)u[0], (float)u[1], (float)u[2], (float)u[3]); zz[i] = make_float4((float)w[0], (float)w[1], (float)w[2], (float)w[3]);
      } else {
        zy[i] = *reinterpret_cast<const float4*>(p.y3 + base + i * 256);
        zz[i] = *reinterpret_cast<const float4*>(p.z3 + base + i * 256);
      }
    }
    if (p.gx_next) {
#pragma unroll
      for (int i = 0; i < NV; ++i) gn[i] = *reinterpret_cast<const float4*>(p.gx_next + base + i * 256);
    }
    const float mud = p.md[row], rsd = p.rd[row], mu3 = p.m3[row], rs3 = p.r3[row];
    float dpv[C6], yv[C6];
#pragma unroll
    for (int k = 0; k < C6; ++k) { dpv[k] = p.dpred[(long long)row * p.dpred_stride + k]; yv[k] = p.pred[(long long)row * C6 + k]; }
    // ---- embedding backward of step i+1
    float dtok[C6];
#pragma unroll
    for (int k = 0; k < C6; ++k) dtok[k] = 0.f;
    if (p.gx_next) {
      float acc[C6];
#pragma unroll
      for (int k = 0; k < C6; ++k) acc[k] = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int c = lane * 4 + i * 256;
        float g[4] = {gn[i].x, gn[i].y, gn[i].z, gn[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (p.edrop.p > 0.f) g[j] = mansy_keep(p.edrop.seed, p.edrop.site, p.edrop.base + (uint32_t)((long long)row * C + c + j), p.edrop.p) ? g[j] * esc : 0.f;
        *reinterpret_cast<float4*>(p.dE_next + base + i * 256) = *reinterpret_cast<const float4*>(g);
        float ew[4][C6];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < C6; ++k) ew[j][k] = p.ew[(c + j) * C6 + k];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int k = 0; k < C6; ++k) acc[k] = fmaf(g[j], ew[j][k], acc[k]);
      }
#pragma unroll
      for (int k = 0; k < C6; ++k) dtok[k] = wave_sum(acc[k]);
    }
    // ---- predictor backward
    float gk[C6];
#pragma unroll
    for (int k = 0; k < C6; ++k) {
      float dd = dpv[k];
      if (p.gx_next) dd += dtok[k];
      gk[k] = dd * yv[k] * (1.f - yv[k]);
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < C6; ++k) p.dz[(long long)row * C6 + k] = gk[k];
    }
    float4 dh[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float a[4] = {0.f, 0.f, 0.f, 0.f};
      float wk[C6][4];
#pragma unroll
      for (int k = 0; k < C6; ++k) *reinterpret_cast<float4*>(wk[k]) = *reinterpret_cast<const float4*>(p.pw + (long long)k * C + lane * 4 + i * 256);
#pragma unroll
      for (int k = 0; k < C6; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaf(gk[k], wk[k][j], a[j]);
      dh[i] = make_float4(a[0], a[1], a[2], a[3]);
    }
    // ---- final decoder LayerNorm backward: dy = dh, z = y3
    float4 xh[NV], dy3[NV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xh[i] = make_float4((zy[i].x - mud) * rsd, (zy[i].y - mud) * rsd, (zy[i].z - mud) * rsd, (zy[i].w - mud) * rsd);
      const float gx = dh[i].x * wd[i].x, gy = dh[i].y * wd[i].y, gz = dh[i].z * wd[i].z, gw = dh[i].w * wd[i].w;
      s1 += (gx + gy) + (gz + gw);
      s2 += (gx * xh[i].x + gy * xh[i].y) + (gz * xh[i].z + gw * xh[i].w);
    }
    s1 = wave_sum(s1) * invC; s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      dy3[i].x = rsd * (dh[i].x * wd[i].x - s1 - xh[i].x * s2); dy3[i].y = rsd * (dh[i].y * wd[i].y - s1 - xh[i].y * s2);
      dy3[i].z = rsd * (dh[i].z * wd[i].z - s1 - xh[i].z * s2); dy3[i].w = rsd * (dh[i].w * wd[i].w - s1 - xh[i].w * s2);
      adw_d[i].x += dh[i].x * xh[i].x; adw_d[i].y += dh[i].y * xh[i].y; adw_d[i].z += dh[i].z * xh[i].z; adw_d[i].w += dh[i].w * xh[i].w;
      adb_d[i].x += dh[i].x; adb_d[i].y += dh[i].y; adb_d[i].z += dh[i].z; adb_d[i].w += dh[i].w;
    }
    // ---- LayerNorm3 backward of the last layer: dy = dy3, z = z3
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      xh[i] = make_float4((zz[i].x - mu3) * rs3, (zz[i].y - mu3) * rs3, (zz[i].z - mu3) * rs3, (zz[i].w - mu3) * rs3);
      const float gx = dy3[i].x * w3[i].x, gy = dy3[i].y * w3[i].y, gz = dy3[i].z * w3[i].z, gw = dy3[i].w * w3[i].w;
      s1 += (gx + gy) + (gz + gw);
      s2 += (gx * xh[i].x + gy * xh[i].y) + (gz * xh[i].z + gw * xh[i].w);
    }
    s1 = wave_sum(s1) * invC; s2 = wave_sum(s2) * invC;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const long long off = base + i * 256;
      float4 o;
      o.x = rs3 * (dy3[i].x * w3[i].x - s1 - xh[i].x * s2); o.y = rs3 * (dy3[i].y * w3[i].y - s1 - xh[i].y * s2);
      o.z = rs3 * (dy3[i].z * w3[i].z - s1 - xh[i].z * s2); o.w = rs3 * (dy3[i].w * w3[i].w - s1 - xh[i].w * s2);
      *reinterpret_cast<float4*>(p.gz + off) = o;
      float4 od = o;
      if (p.drop3.p > 0.f) {
        od.x = mansy_keep(p.drop3.seed, p.drop3.site, p.drop3.base + (uint32_t)(off + 0), p.drop3.p) ? o.x * dsc : 0.f;
        od.y = mansy_keep(p.drop3.seed, p.drop3.site, p.drop3.base + (uint32_t)(off + 1), p.drop3.p) ? o.y * dsc : 0.f;
        od.z = mansy_keep(p.drop3.seed, p.drop3.site, p.drop3.base + (uint32_t)(off + 2), p.drop3.p) ? o.z * dsc : 0.f;
        od.w = mansy_keep(p.drop3.seed, p.drop3.site, p.drop3.base + (uint32_t)(off + 3), p.drop3.p) ? o.w * dsc : 0.f;
      }
      if (!p.dbr3_img_only) *reinterpret_cast<float4*>(p.dbr3 + off) = od;
      if (p.dbr3_16) mansy_st_bf16x4(p.dbr3_16 + off, od.x, od.y, od.z, od.w);
      adw_3[i].x += dy3[i].x * xh[i].x; adw_3[i].y += dy3[i].y * xh[i].y; adw_3[i].z += dy3[i].z * xh[i].z; adw_3[i].w += dy3[i].w * xh[i].w;
      adb_3[i].x += dy3[i].x; adb_3[i].y += dy3[i].y; adb_3[i].z += dy3[i].z; adb_3[i].w += dy3[i].w;
    }
  }
  // ---- the two LayerNorms' column sums: waves -> workgroup -> this workgroup's slots (accumulated over the steps)
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = lane * 4 + i * 256;
    *reinterpret_cast<float4*>(red + (wave * 4 + 0) * C + c) = adw_d[i];
    *reinterpret_cast<float4*>(red + (wave * 4 + 1) * C + c) = adb_d[i];
    *reinterpret_cast<float4*>(red + (wave * 4 + 2) * C + c) = adw_3[i];
    *reinterpret_cast<float4*>(red + (wave * 4 + 3) * C + c) = adb_3[i];
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 4 * C; idx += 64 * HEAD_WAVES) {
    const int which = idx / C, c = idx % C;
    float a = 0.f;
#pragma unroll
    for (int wv = 0; wv < HEAD_WAVES; ++wv) a += red[(wv * 4 + which) * C + c];
    float* slot = (which < 2 ? p.part_dn : p.part_n3) + (size_t)blockIdx.x * 2 * C + (which & 1) * C + c;
    *slot += a;
  }
}

}  // namespace

// C in {256, 512}: the backward head keeps HEAD_WAVES x 4 x C floats of LDS (64 KiB at C = 512)
int mansy_dec_tail_ok(int C, int c6) { return (C == 256 || C == 512) && c6 == C6; }

int mansy_launch_dec_tail_fwd(const MansyDecTailFwd& p, hipStream_t st) {
  MANSY_REQUIRE(mansy_dec_tail_ok(p.C, p.C6), "dec_tail_fwd: unsupported width (C=%d, tokens of %d)", p.C, p.C6);
  MANSY_REQUIRE(p.a && p.b && p.n3_w && p.z3 && p.y3 && p.m3 && p.r3 && p.dn_w && p.dec_out && p.md && p.rd && p.pw &&
                p.tok_next && (!p.emb_next || (p.ew && p.pe_row)), "dec_tail_fwd: null pointer");
  if (p.rows <= 0) return MANSY_OK;
  const dim3 grid(mansy_ceil_div(p.rows, 4));
  switch (p.C / 256) {
    case 1: MANSY_LAUNCH(dec_tail_fwd_kernel<1>, grid, dim3(256), 0, st, p); break;
    case 2: MANSY_LAUNCH(dec_tail_fwd_kernel<2>, grid, dim3(256), 0, st, p); break;
    case 3: MANSY_LAUNCH(dec_tail_fwd_kernel<3>, grid, dim3(256), 0, st, p); break;
    default: MANSY_LAUNCH(dec_tail_fwd_kernel<4>, grid, dim3(256), 0, st, p); break;
  }
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_dec_head_bwd(const MansyDecHeadBwd& p, int n_slots, hipStream_t st) {
  MANSY_REQUIRE(mansy_dec_tail_ok(p.C, p.C6), "dec_head_bwd: unsupported width (C=%d, tokens of %d)", p.C, p.C6);
  MANSY_REQUIRE(p.dpred && p.pred && p.pw && p.dz && p.y3 && p.md && p.rd && p.dn_w && p.part_dn && p.z3 && p.m3 && p.r3 && p.n3_w && p.part_n3 &&
                p.gz && p.dbr3 && (!p.gx_next || (p.ew && p.dE_next)) && n_slots >= 1, "dec_head_bwd: null pointer");
  if (p.rows <= 0) return MANSY_OK;
  const size_t lds = (size_t)HEAD_WAVES * 4 * p.C * sizeof(float);
  const dim3 block(64 * HEAD_WAVES);
  switch (p.C / 256) {
    case 1: MANSY_LAUNCH(dec_head_bwd_kernel<1>, dim3(n_slots), block, lds, st, p); break;
    case 2: MANSY_LAUNCH(dec_head_bwd_kernel<2>, dim3(n_slots), block, lds, st, p); break;
    case 3: MANSY_LAUNCH(dec_head_bwd_kernel<3>, dim3(n_slots), block, lds, st, p); break;
    default: MANSY_LAUNCH(dec_head_bwd_kernel<4>, dim3(n_slots), block, lds, st, p); break;
  }
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
