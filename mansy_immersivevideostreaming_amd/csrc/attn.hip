// Fused small-sequence multi-head attention (forward + backward) for the viewport Transformer:
// encoder self-attention (S x S, S <= 16), KV-cached decoder self-attention (1 x (i+1)) and
// cross-attention over the distilled memory (1 x M).  Reference arithmetic:
// torch.nn.MultiheadAttention inside nn.Transformer{Encoder,Decoder}Layer:
// softmax(Q K^T / sqrt(dh)) with attention-probability dropout, then P V.
//
// One 64-lane wavefront per (batch, head): everything lives in registers / a private LDS slice,
// QK^T + softmax + dropout + PV in one kernel, no HBM round trip for the score matrix.
// These blocks (<= 16x16x64) cannot fill an MFMA tile; the kernel is HBM/latency-bound and its
// job is coalesced 256-byte row reads and zero extra traffic.
#include "mansy_kernels.h"

namespace {

constexpr int LMAX = 16;
constexpr int DH_LD = 65;            // padded row (floats) -> conflict-free column-strided reads
constexpr int WAVES = 4;

struct AttnPtrs {
  const float* Q; const float* K; const float* V; float* O; float* P;
  const float* dO; float* dQ; float* dK; float* dV;
  float* dS_out; float* Pk_out;     // q1x4 backward, deferred dK/dV: [nb*H, Lk] score gradients / dropped probabilities
  // bf16 images of the OUTPUT rows (bf16-storage mode, round 6; 4-heads-per-wave kernels only): the float4 stored at address a also goes, rounded, to
  // img_s + (a - img_f) -- one mapping for O / dQ / dK / dV (they are rows of one slab).  Null: no image.
  const float* img_f; unsigned short* img_s;
  int img_only;                     // != 0 (with img_s): the rows are operands of dense products and nothing else -- the float4 store is skipped
  int kv16;                         // != 0 (with img_s; q1x4 kernels): the K / V rows (and the pull form's Q rows) are READ from their bf16 images -- the
                                    // K/V cache of the bf16-storage mode is bf16 in HBM (written by the projection's epilogue): half the cache traffic
};
// a 4-float piece of a K / V / Q row: from the float slab, or (kv16) from its bf16 image at the same element index
__device__ __forceinline__ float4 attn_ld4(const AttnPtrs& p, const float* addr) {
  if (p.kv16) {
    const mansy_bf16x4 t = *reinterpret_cast<const mansy_bf16x4*>(p.img_s + (addr - p.img_f));
    return make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
  }
  return *reinterpret_cast<const float4*>(addr);
}
#define ATTN_ST(p, addr, v) do { if (!(p).img_only) *reinterpret_cast<float4*>(addr) = (v); ATTN_IMG(p, addr, v); } while (0)
#define ATTN_IMG(p, addr, v) do { if ((p).img_s) mansy_st_bf16x4((p).img_s + ((addr) - (p).img_f), (v).x, (v).y, (v).z, (v).w); } while (0)

__device__ __forceinline__ void load_rows(const float* base, long long rs, int L, int dh, int lane, float* lds) {
  for (int i = 0; i < L; ++i)
    if (lane < dh) lds[i * DH_LD + lane] = base[i * rs + lane];
}

__global__ __launch_bounds__(64 * WAVES) void attn_fwd_kernel(AttnPtrs p, AttnShape s, MansyDrop drop) {
  __shared__ float sQ[WAVES][LMAX * DH_LD];
  __shared__ float sK[WAVES][LMAX * DH_LD];
  __shared__ float sP[WAVES][LMAX * (LMAX + 1)];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long bh = (long long)blockIdx.x * WAVES + wave;
  const bool active = bh < (long long)s.nb * s.H;
  const int b = active ? (int)(bh / s.H) : 0, h = active ? (int)(bh % s.H) : 0;
  const int Lq = s.Lq, Lk = s.Lk, dh = s.dh;
  const float* Qb = p.Q + b * s.q_bs + h * dh;
  const float* Kb = p.K + b * s.k_bs + h * dh;
  const float* Vb = p.V + b * s.v_bs + h * dh;
  float* q_l = sQ[wave]; float* k_l = sK[wave]; float* p_l = sP[wave];
  if (active) {
    load_rows(Qb, s.q_rs, Lq, dh, lane, q_l);
    load_rows(Kb, s.k_rs, Lk, dh, lane, k_l);
  }
  __syncthreads();
  if (active) {
    for (int pr = lane; pr < Lq * Lk; pr += 64) {
      const int i = pr / Lk, j = pr % Lk;
      float acc = 0.f;
      for (int d = 0; d < dh; ++d) acc = fmaf(q_l[i * DH_LD + d], k_l[j * DH_LD + d], acc);
      p_l[i * (LMAX + 1) + j] = acc * s.scale;
    }
  }
  __syncthreads();
  if (active && lane < Lq) {
    const int i = lane;
    float m = -INFINITY;
    for (int j = 0; j < Lk; ++j) m = fmaxf(m, p_l[i * (LMAX + 1) + j]);
    float sum = 0.f;
    for (int j = 0; j < Lk; ++j) { const float e = expf(p_l[i * (LMAX + 1) + j] - m); p_l[i * (LMAX + 1) + j] = e; sum += e; }
    const float inv = 1.f / sum;
    const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
    for (int j = 0; j < Lk; ++j) {
      float pv = p_l[i * (LMAX + 1) + j] * inv;
      const long long pidx = (bh * Lq + i) * Lk + j;
      if (p.P) p.P[pidx] = pv;
      if (drop.p > 0.f) pv = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)pidx, drop.p) ? pv * ds : 0.f;
      p_l[i * (LMAX + 1) + j] = pv;
    }
  }
  __syncthreads();
  if (active && lane < dh) {
    float v[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) v[j] = j < Lk ? Vb[j * s.v_rs + lane] : 0.f;
    float* Ob = p.O + b * s.o_bs + h * dh;
    for (int i = 0; i < Lq; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < LMAX; ++j) if (j < Lk) acc = fmaf(p_l[i * (LMAX + 1) + j], v[j], acc);
      Ob[i * s.o_rs + lane] = acc;
    }
  }
}

__global__ __launch_bounds__(64 * WAVES) void attn_bwd_kernel(AttnPtrs p, AttnShape s, MansyDrop drop, int accum_kv) {
  __shared__ float sA[WAVES][LMAX * DH_LD];   // dO rows
  __shared__ float sB[WAVES][LMAX * DH_LD];   // V rows
  __shared__ float sP[WAVES][LMAX * (LMAX + 1)];    // P (pre-dropout)
  __shared__ float sD[WAVES][LMAX * (LMAX + 1)];    // dropped P, later dS
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long bh = (long long)blockIdx.x * WAVES + wave;
  const bool active = bh < (long long)s.nb * s.H;
  const int b = active ? (int)(bh / s.H) : 0, h = active ? (int)(bh % s.H) : 0;
  const int Lq = s.Lq, Lk = s.Lk, dh = s.dh;
  const float* Qb = p.Q + b * s.q_bs + h * dh;
  const float* Kb = p.K + b * s.k_bs + h * dh;
  const float* Vb = p.V + b * s.v_bs + h * dh;
  const float* dOb = p.dO + b * s.o_bs + h * dh;
  float* do_l = sA[wave]; float* v_l = sB[wave]; float* p_l = sP[wave]; float* d_l = sD[wave];
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  if (active) {
    load_rows(dOb, s.o_rs, Lq, dh, lane, do_l);
    load_rows(Vb, s.v_rs, Lk, dh, lane, v_l);
    for (int pr = lane; pr < Lq * Lk; pr += 64) {
      const int i = pr / Lk, j = pr % Lk;
      const long long pidx = (bh * Lq + i) * Lk + j;
      const float pv = p.P[pidx];
      p_l[i * (LMAX + 1) + j] = pv;
      float keepf = 1.f;
      if (drop.p > 0.f) keepf = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)pidx, drop.p) ? ds : 0.f;
      d_l[i * (LMAX + 1) + j] = pv * keepf;    // dropped probabilities (for dV)
    }
  }
  __syncthreads();
  // dV[j][d] = sum_i Pd[i][j] dO[i][d]
  if (active && lane < dh) {
    float* dVb = p.dV + b * s.v_bs + h * dh;
    for (int j = 0; j < Lk; ++j) {
      float acc = 0.f;
      for (int i = 0; i < Lq; ++i) acc = fmaf(d_l[i * (LMAX + 1) + j], do_l[i * DH_LD + lane], acc);
      float* dst = dVb + j * s.v_rs + lane;
      *dst = accum_kv ? *dst + acc : acc;
    }
  }
  __syncthreads();
  // dPd[i][j] = sum_d dO[i][d] V[j][d] ; dP = dPd * keep
  if (active) {
    for (int pr = lane; pr < Lq * Lk; pr += 64) {
      const int i = pr / Lk, j = pr % Lk;
      float acc = 0.f;
      for (int d = 0; d < dh; ++d) acc = fmaf(do_l[i * DH_LD + d], v_l[j * DH_LD + d], acc);
      float keepf = 1.f;
      if (drop.p > 0.f) keepf = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)((bh * Lq + i) * Lk + j), drop.p) ? ds : 0.f;
      d_l[i * (LMAX + 1) + j] = acc * keepf;   // dP (wrt pre-dropout probabilities)
    }
  }
  __syncthreads();
  if (active && lane < Lq) {
    const int i = lane;
    float delta = 0.f;
    for (int j = 0; j < Lk; ++j) delta = fmaf(p_l[i * (LMAX + 1) + j], d_l[i * (LMAX + 1) + j], delta);
    for (int j = 0; j < Lk; ++j)
      d_l[i * (LMAX + 1) + j] = p_l[i * (LMAX + 1) + j] * (d_l[i * (LMAX + 1) + j] - delta) * s.scale;   // dS
  }
  __syncthreads();
  if (active && lane < dh) {
    float kq[LMAX];
    // dQ[i][d] = sum_j dS[i][j] K[j][d]
#pragma unroll
    for (int j = 0; j < LMAX; ++j) kq[j] = j < Lk ? Kb[j * s.k_rs + lane] : 0.f;
    float* dQb = p.dQ + b * s.q_bs + h * dh;
    for (int i = 0; i < Lq; ++i) {
      float acc = 0.f;
#pragma unroll
      for (int j = 0; j < LMAX; ++j) if (j < Lk) acc = fmaf(d_l[i * (LMAX + 1) + j], kq[j], acc);
      dQb[i * s.q_rs + lane] = acc;
    }
    // dK[j][d] = sum_i dS[i][j] Q[i][d]
#pragma unroll
    for (int i = 0; i < LMAX; ++i) kq[i] = i < Lq ? Qb[i * s.q_rs + lane] : 0.f;
    float* dKb = p.dK + b * s.k_bs + h * dh;
    for (int j = 0; j < Lk; ++j) {
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < LMAX; ++i) if (i < Lq) acc = fmaf(d_l[i * (LMAX + 1) + j], kq[i], acc);
      float* dst = dKb + j * s.k_rs + lane;
      *dst = accum_kv ? *dst + acc : acc;
    }
  }
}

// ---- Lq == 1 (KV-cached decoder self-attention and cross-attention): one wave per (batch, head), lane = head-dim
// element; scores via wave reductions, everything in registers, no LDS and no barriers.  Pure HBM streaming:
// forward reads (2 Lk + 1) rows of 256 B and writes one; backward additionally read-modify-writes 2 Lk rows.
__global__ __launch_bounds__(64 * WAVES) void attn_fwd_q1_kernel(AttnPtrs p, AttnShape s, MansyDrop drop) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long bh = (long long)blockIdx.x * WAVES + wave;
  if (bh >= (long long)s.nb * s.H) return;
  const int b = (int)(bh / s.H), h = (int)(bh % s.H);
  const int Lk = s.Lk, dh = s.dh;
  const bool on = lane < dh;
  const float* Kb = p.K + b * s.k_bs + h * dh + lane;
  const float* Vb = p.V + b * s.v_bs + h * dh + lane;
  const float q = on ? p.Q[b * s.q_bs + h * dh + lane] : 0.f;
  float sc[LMAX], v[LMAX];
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    const float kj = (on && j < Lk) ? Kb[j * s.k_rs] : 0.f;
    v[j] = (on && j < Lk) ? Vb[j * s.v_rs] : 0.f;
    sc[j] = q * kj;
  }
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    if (j < Lk) { sc[j] = wave_sum(sc[j]) * s.scale; m = fmaxf(m, sc[j]); }
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) if (j < Lk) { sc[j] = expf(sc[j] - m); sum += sc[j]; }
  const float inv = 1.f / sum;
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float o = 0.f;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    if (j < Lk) {
      float pv = sc[j] * inv;
      const long long pidx = bh * Lk + j;
      if (p.P && lane == j) p.P[pidx] = pv;
      if (drop.p > 0.f) pv = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)pidx, drop.p) ? pv * ds : 0.f;
      o = fmaf(pv, v[j], o);
    }
  }
  if (on) p.O[b * s.o_bs + h * dh + lane] = o;
}

__global__ __launch_bounds__(64 * WAVES) void attn_bwd_q1_kernel(AttnPtrs p, AttnShape s, MansyDrop drop, int accum_kv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long bh = (long long)blockIdx.x * WAVES + wave;
  if (bh >= (long long)s.nb * s.H) return;
  const int b = (int)(bh / s.H), h = (int)(bh % s.H);
  const int Lk = s.Lk, dh = s.dh;
  const bool on = lane < dh;
  const float* Kb = p.K + b * s.k_bs + h * dh + lane;
  const float* Vb = p.V + b * s.v_bs + h * dh + lane;
  float* dKb = p.dK + b * s.k_bs + h * dh + lane;
  float* dVb = p.dV + b * s.v_bs + h * dh + lane;
  const float q = on ? p.Q[b * s.q_bs + h * dh + lane] : 0.f;
  const float dO = on ? p.dO[b * s.o_bs + h * dh + lane] : 0.f;
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float k[LMAX], dP[LMAX], P[LMAX], keepf[LMAX], dk_old[LMAX], dv_old[LMAX];
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    const bool in = on && j < Lk;
    k[j] = in ? Kb[j * s.k_rs] : 0.f;
    const float vj = in ? Vb[j * s.v_rs] : 0.f;
    dk_old[j] = (in && accum_kv) ? dKb[j * s.k_rs] : 0.f;
    dv_old[j] = (in && accum_kv) ? dVb[j * s.v_rs] : 0.f;
    dP[j] = dO * vj;
    P[j] = j < Lk ? p.P[bh * Lk + j] : 0.f;
    keepf[j] = 1.f;
    if (drop.p > 0.f && j < Lk) keepf[j] = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(bh * Lk + j), drop.p) ? ds : 0.f;
  }
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    if (j < Lk) { dP[j] = wave_sum(dP[j]) * keepf[j]; delta = fmaf(P[j], dP[j], delta); }
  }
  float dq = 0.f;
#pragma unroll
  for (int j = 0; j < LMAX; ++j) {
    if (j < Lk) {
      const float dS = P[j] * (dP[j] - delta) * s.scale;
      dq = fmaf(dS, k[j], dq);
      if (on) {
        dKb[j * s.k_rs] = dk_old[j] + dS * q;
        dVb[j * s.v_rs] = dv_old[j] + P[j] * keepf[j] * dO;
      }
    }
  }
  if (on) p.dQ[b * s.q_bs + h * dh + lane] = dq;
}

// ---- Lq == 1, dh == 64, H % 4 == 0: FOUR heads per wave, 16 lanes x float4 per head.  A (batch, 4-head) group is 1 KiB of
// contiguous q/k/v, so every load/store is a full-width 16-B-per-lane instruction (4x fewer memory instructions and waves
// than one head per wave) and the score reductions stay inside 16-lane groups (4 shuffle steps).
__device__ __forceinline__ float group16_sum(float v) { return row16_sum(v); }
__device__ __forceinline__ float dot4(const float4& a, const float4& b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// LKT = compile-time bound on the key rows (>= s.Lk): every K / V / P load of a wave is issued up front, unconditionally
// (rows >= Lk re-read row Lk-1 and are masked out of the arithmetic).  With a runtime `if (j < Lk)` around each load hipcc
// branches and waits per row: Lk dependent HBM round trips per wave, 3.5 us per cached step in the decode loop.
template <int LKT>
__global__ __launch_bounds__(64 * WAVES) void attn_fwd_q1x4_kernel(AttnPtrs p, AttnShape s, MansyDrop drop) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hg = s.H >> 2;                                  // head groups per batch entry
  const long long g = (long long)blockIdx.x * WAVES + wave;
  if (g >= (long long)s.nb * hg) return;
  const int b = (int)(g / hg), h = (int)(g % hg) * 4 + (lane >> 4), c = lane & 15;
  const long long bh = (long long)b * s.H + h;
  const int Lk = s.Lk;
  const int col = h * 64 + c * 4;
  const float4 q = *reinterpret_cast<const float4*>(p.Q + b * s.q_bs + col);
  const float* Kb = p.K + b * s.k_bs + col;
  const float* Vb = p.V + b * s.v_bs + col;
  float sc[LKT];
  float4 k[LKT], v[LKT];
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    const int jc = min(j, Lk - 1);
    k[j] = attn_ld4(p, Kb + jc * s.k_rs);
    v[j] = attn_ld4(p, Vb + jc * s.v_rs);
  }
  float m = -INFINITY;
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    sc[j] = group16_sum(dot4(q, k[j])) * s.scale;
    if (j < Lk) m = fmaxf(m, sc[j]);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < LKT; ++j) { sc[j] = j < Lk ? expf(sc[j] - m) : 0.f; sum += sc[j]; }
  const float inv = 1.f / sum;
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    float pv = sc[j] * inv;                      // 0 for j >= Lk
    const long long pidx = bh * Lk + j;
    if (p.P && c == j && j < Lk) p.P[pidx] = pv;
    if (drop.p > 0.f) pv = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)pidx, drop.p) ? pv * ds : 0.f;
    o.x = fmaf(pv, v[j].x, o.x); o.y = fmaf(pv, v[j].y, o.y); o.z = fmaf(pv, v[j].z, o.z); o.w = fmaf(pv, v[j].w, o.w);
  }
  ATTN_ST(p, p.O + b * s.o_bs + col, o);
}

template <int LKT>
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_q1x4_kernel(AttnPtrs p, AttnShape s, MansyDrop drop, int accum_kv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hg = s.H >> 2;
  const long long g = (long long)blockIdx.x * WAVES + wave;
  if (g >= (long long)s.nb * hg) return;
  const int b = (int)(g / hg), h = (int)(g % hg) * 4 + (lane >> 4);
  const long long bh = (long long)b * s.H + h;
  const int Lk = s.Lk;
  const int col = h * 64 + (lane & 15) * 4;
  const float4 q = *reinterpret_cast<const float4*>(p.Q + b * s.q_bs + col);
  const float4 dO = *reinterpret_cast<const float4*>(p.dO + b * s.o_bs + col);
  const float* Kb = p.K + b * s.k_bs + col;
  const float* Vb = p.V + b * s.v_bs + col;
  float* dKb = p.dK + b * s.k_bs + col;
  float* dVb = p.dV + b * s.v_bs + col;
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float4 k[LKT], vv[LKT];
  float dP[LKT], P[LKT], keepf[LKT];
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    const int jc = min(j, Lk - 1);
    k[j] = attn_ld4(p, Kb + jc * s.k_rs);
    vv[j] = attn_ld4(p, Vb + jc * s.v_rs);
    P[j] = p.P[bh * Lk + jc];
  }
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    keepf[j] = 1.f;
    if (drop.p > 0.f) keepf[j] = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(bh * Lk + j), drop.p) ? ds : 0.f;
    if (j >= Lk) P[j] = 0.f;
    dP[j] = group16_sum(dot4(dO, vv[j])) * keepf[j];
    delta = fmaf(P[j], dP[j], delta);
  }
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
  float dS[LKT];
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    dS[j] = P[j] * (dP[j] - delta) * s.scale;      // 0 for j >= Lk
    dq.x = fmaf(dS[j], k[j].x, dq.x); dq.y = fmaf(dS[j], k[j].y, dq.y); dq.z = fmaf(dS[j], k[j].z, dq.z); dq.w = fmaf(dS[j], k[j].w, dq.w);
  }
  ATTN_ST(p, p.dQ + b * s.q_bs + col, dq);
  if (p.dS_out) {          // deferred dK / dV (mansy_launch_attn_kvgrad): keep this step's coefficients, touch no K/V gradient row
    const int c = lane & 15;
#pragma unroll
    for (int j = 0; j < LKT; ++j) {
      if (c == j && j < Lk) { p.dS_out[bh * Lk + j] = dS[j]; p.Pk_out[bh * Lk + j] = P[j] * keepf[j]; }
    }
    return;
  }
  // dK / dV rows: read-modify-write in groups of 4 rows (8 loads in flight, then 8 stores)
#pragma unroll
  for (int j0 = 0; j0 < LKT; j0 += 4) {
    float4 ok[4], ov[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ok[u] = make_float4(0.f, 0.f, 0.f, 0.f); ov[u] = ok[u]; }
    if (accum_kv) {                              // uniform: one branch around the whole group's loads
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (j0 + u < LKT) {
          const int jc = min(j0 + u, Lk - 1);
          ok[u] = *reinterpret_cast<const float4*>(dKb + jc * s.k_rs);
          ov[u] = *reinterpret_cast<const float4*>(dVb + jc * s.v_rs);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
      if (j < LKT && j < Lk) {
        const float pd = P[j] * keepf[j];
        const float4 dk = make_float4(ok[u].x + dS[j] * q.x, ok[u].y + dS[j] * q.y, ok[u].z + dS[j] * q.z, ok[u].w + dS[j] * q.w);
        const float4 dv = make_float4(ov[u].x + pd * dO.x, ov[u].y + pd * dO.y, ov[u].z + pd * dO.z, ov[u].w + pd * dO.w);
        *reinterpret_cast<float4*>(dKb + j * s.k_rs) = dk;
        *reinterpret_cast<float4*>(dVb + j * s.v_rs) = dv;
      }
    }
  }
}

// KV-cached decoder SELF-attention backward at step `step` (keys/values = steps 0..step, Lk = step+1), "pull" form.
// The steps run T-1 -> 0, so when step i is processed every later step i' > i has already recorded its coefficients
// dS_{i'}[.] / Pk_{i'}[.] (this kernel stores row i for the earlier steps).  Row i of the K/V gradient is then complete:
//   dK_i = sum_{i' >= i} dS_{i'}[i] q_{i'}        dV_i = sum_{i' >= i} Pk_{i'}[i] dO_{i'}
// and is written ONCE, instead of every step read-modify-writing rows 0..i (2 x 2 x (i+1) row accesses -> 2 x (T-i) reads).
struct SelfPull {
  const float* Q_all; long long q_ts;      // query rows of all steps: step i at Q_all + i*q_ts (+ b*q_bs)
  const float* dO_all; long long o_ts;     // attention-output gradients of all steps (steps > `step` already written)
  float* dS_all; float* Pk_all;            // [T][nb*H][T] coefficient rows
  int T, step;
};
template <int LKT>
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_selfpull_q1x4_kernel(AttnPtrs p, AttnShape s, MansyDrop drop, SelfPull sp) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hg = s.H >> 2;
  const long long g = (long long)blockIdx.x * WAVES + wave;
  if (g >= (long long)s.nb * hg) return;
  const int b = (int)(g / hg), h = (int)(g % hg) * 4 + (lane >> 4), c = lane & 15;
  const long long bh = (long long)b * s.H + h, nbh = (long long)s.nb * s.H;
  const int Lk = s.Lk;                       // == sp.step + 1
  const int col = h * 64 + c * 4;
  const float4 q = *reinterpret_cast<const float4*>(sp.Q_all + sp.step * sp.q_ts + b * s.q_bs + col);
  const float4 dO = *reinterpret_cast<const float4*>(sp.dO_all + sp.step * sp.o_ts + b * s.o_bs + col);
  const float* Kb = p.K + b * s.k_bs + col;
  const float* Vb = p.V + b * s.v_bs + col;
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float4 k[LKT], vv[LKT];
  float dP[LKT], P[LKT], keepf[LKT];
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    const int jc = min(j, Lk - 1);
    k[j] = attn_ld4(p, Kb + jc * s.k_rs);
    vv[j] = attn_ld4(p, Vb + jc * s.v_rs);
    P[j] = p.P[bh * Lk + jc];
  }
  float delta = 0.f;
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    keepf[j] = 1.f;
    if (drop.p > 0.f) keepf[j] = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(bh * Lk + j), drop.p) ? ds : 0.f;
    if (j >= Lk) P[j] = 0.f;
    dP[j] = group16_sum(dot4(dO, vv[j])) * keepf[j];
    delta = fmaf(P[j], dP[j], delta);
  }
  float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
  float dS_own = 0.f, Pk_own = 0.f;          // coefficients (step, step): this step's term of its own K/V row
  float* dS_row = sp.dS_all + ((long long)sp.step * nbh + bh) * sp.T;
  float* Pk_row = sp.Pk_all + ((long long)sp.step * nbh + bh) * sp.T;
#pragma unroll
  for (int j = 0; j < LKT; ++j) {
    const float dSj = P[j] * (dP[j] - delta) * s.scale;      // 0 for j >= Lk
    const float Pkj = P[j] * keepf[j];
    dq.x = fmaf(dSj, k[j].x, dq.x); dq.y = fmaf(dSj, k[j].y, dq.y); dq.z = fmaf(dSj, k[j].z, dq.z); dq.w = fmaf(dSj, k[j].w, dq.w);
    if (j == Lk - 1) { dS_own = dSj; Pk_own = Pkj; }
    if (c == j && j < Lk) { dS_row[j] = dSj; Pk_row[j] = Pkj; }
  }
  ATTN_ST(p, p.dQ + b * s.q_bs + col, dq);
  float4 dk = make_float4(dS_own * q.x, dS_own * q.y, dS_own * q.z, dS_own * q.w);
  float4 dv = make_float4(Pk_own * dO.x, Pk_own * dO.y, Pk_own * dO.z, Pk_own * dO.w);
  for (int i0 = sp.step + 1; i0 < sp.T; i0 += 4) {          // later steps, four at a time: 8 row loads in flight
    float4 qn[4], gn[4];
    float a[4], bb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int ic = min(i0 + u, sp.T - 1);
      qn[u] = attn_ld4(p, sp.Q_all + ic * sp.q_ts + b * s.q_bs + col);
      gn[u] = *reinterpret_cast<const float4*>(sp.dO_all + ic * sp.o_ts + b * s.o_bs + col);
      const long long ci = ((long long)ic * nbh + bh) * sp.T + sp.step;
      a[u] = sp.dS_all[ci]; bb[u] = sp.Pk_all[ci];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float au = i0 + u < sp.T ? a[u] : 0.f, bu = i0 + u < sp.T ? bb[u] : 0.f;
      dk.x = fmaf(au, qn[u].x, dk.x); dk.y = fmaf(au, qn[u].y, dk.y); dk.z = fmaf(au, qn[u].z, dk.z); dk.w = fmaf(au, qn[u].w, dk.w);
      dv.x = fmaf(bu, gn[u].x, dv.x); dv.y = fmaf(bu, gn[u].y, dv.y); dv.z = fmaf(bu, gn[u].z, dv.z); dv.w = fmaf(bu, gn[u].w, dv.w);
    }
  }
  ATTN_ST(p, p.dK + b * s.k_bs + sp.step * s.k_rs + col, dk);
  ATTN_ST(p, p.dV + b * s.v_bs + sp.step * s.v_rs + col, dv);
}

// Deferred K/V gradients of a Lq == 1 attention evaluated at TT query steps against the SAME K/V rows (decoder
// cross-attention: every step attends to the distilled memory):
//   dK[j] = sum_i dS_i[j] * q_i        dV[j] = sum_i Pk_i[j] * dO_i
// replaces TT read-modify-write passes over the K/V gradient rows (2 x Lk x 1 KiB per wave and step) by one pass that
// reads the TT query / output-gradient rows once.  Same wave layout as the q1x4 kernels (4 heads x 16 lanes x float4).
template <int TT>
__global__ __launch_bounds__(64 * WAVES) void attn_kvgrad_q1x4_kernel(const float* __restrict__ Q_all, long long q_ts,
                                                                      const float* __restrict__ dO_all, long long o_ts,
                                                                      const float* __restrict__ dS_all, const float* __restrict__ Pk_all,
                                                                      float* __restrict__ dK, float* __restrict__ dV, AttnShape s, int T,
                                                                      int accum, const float* img_f, unsigned short* img_s) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hg = s.H >> 2;
  const long long g = (long long)blockIdx.x * WAVES + wave;
  if (g >= (long long)s.nb * hg) return;
  const int b = (int)(g / hg), h = (int)(g % hg) * 4 + (lane >> 4), c = lane & 15;
  const long long bh = (long long)b * s.H + h;
  const int Lk = s.Lk;
  const int col = h * 64 + c * 4;
  const long long coef_ts = (long long)s.nb * s.H * Lk;
  float4 q[TT], go[TT];
  float aS[TT], aP[TT];
#pragma unroll
  for (int i = 0; i < TT; ++i) {
    const int ic = min(i, T - 1);
    q[i] = *reinterpret_cast<const float4*>(Q_all + ic * q_ts + b * s.q_bs + col);
    go[i] = *reinterpret_cast<const float4*>(dO_all + ic * o_ts + b * s.o_bs + col);
    const long long ci = ic * coef_ts + bh * Lk + min(c, Lk - 1);
    aS[i] = i < T ? dS_all[ci] : 0.f;      // lane c of a head group holds coefficient j = c
    aP[i] = i < T ? Pk_all[ci] : 0.f;
  }
  float* dKb = dK + b * s.k_bs + col;
  float* dVb = dV + b * s.v_bs + col;
  for (int j = 0; j < Lk; ++j) {
    float4 dk = make_float4(0.f, 0.f, 0.f, 0.f), dv = dk;
    if (accum) {
      dk = *reinterpret_cast<const float4*>(dKb + j * s.k_rs);
      dv = *reinterpret_cast<const float4*>(dVb + j * s.v_rs);
    }
    const int src = (lane & 48) + j;
#pragma unroll
    for (int i = 0; i < TT; ++i) {
      const float a = __shfl(aS[i], src, 64), bb = __shfl(aP[i], src, 64);
      dk.x = fmaf(a, q[i].x, dk.x); dk.y = fmaf(a, q[i].y, dk.y); dk.z = fmaf(a, q[i].z, dk.z); dk.w = fmaf(a, q[i].w, dk.w);
      dv.x = fmaf(bb, go[i].x, dv.x); dv.y = fmaf(bb, go[i].y, dv.y); dv.z = fmaf(bb, go[i].z, dv.z); dv.w = fmaf(bb, go[i].w, dv.w);
    }
    *reinterpret_cast<float4*>(dKb + j * s.k_rs) = dk;
    *reinterpret_cast<float4*>(dVb + j * s.v_rs) = dv;
    if (img_s) {      // bf16-storage mode: the K/V gradient rows are operands of the memory projection's dW and dX products
      mansy_st_bf16x4(img_s + (dKb + j * s.k_rs - img_f), dk.x, dk.y, dk.z, dk.w);
      mansy_st_bf16x4(img_s + (dVb + j * s.v_rs - img_f), dv.x, dv.y, dv.z, dv.w);
    }
  }
}

// ---- Lq == Lk == S <= 10, dh == 64, H % 4 == 0 (encoder self-attention): four heads per wave, 16 lanes x float4 per head,
// every Q / K / V (/ dO) row of the (batch, 4-head) group loaded up front as full 1-KiB wave accesses.  Lane c of a head
// group owns score column j = c (softmax output, dropout mask, dS); columns are broadcast inside the 16-lane group when
// a sum over j needs them.  ST = compile-time S bound (rows >= S re-read row S-1 and are masked).
template <int ST>
__global__ __launch_bounds__(64 * WAVES) void attn_fwd_sx4_kernel(AttnPtrs p, AttnShape s, MansyDrop drop) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hg = s.H >> 2;
  const long long g = (long long)blockIdx.x * WAVES + wave;
  if (g >= (long long)s.nb * hg) return;
  const int b = (int)(g / hg), h = (int)(g % hg) * 4 + (lane >> 4), c = lane & 15, gbase = lane & 48;
  const long long bh = (long long)b * s.H + h;
  const int S = s.Lk;
  const int col = h * 64 + c * 4;
  const float* Qb = p.Q + b * s.q_bs + col;
  const float* Kb = p.K + b * s.k_bs + col;
  const float* Vb = p.V + b * s.v_bs + col;
  float* Ob = p.O + b * s.o_bs + col;
  float4 q[ST], k[ST], v[ST];
#pragma unroll
  for (int j = 0; j < ST; ++j) {
    const int jc = min(j, S - 1);
    q[j] = *reinterpret_cast<const float4*>(Qb + jc * s.q_rs);
    k[j] = *reinterpret_cast<const float4*>(Kb + jc * s.k_rs);
    v[j] = *reinterpret_cast<const float4*>(Vb + jc * s.v_rs);
  }
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
#pragma unroll
  for (int i = 0; i < ST; ++i) {
    if (i < S) {
      float sc[ST];
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < ST; ++j) {
        sc[j] = group16_sum(dot4(q[i], k[j])) * s.scale;
        if (j < S) m = fmaxf(m, sc[j]);
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < ST; ++j) { sc[j] = j < S ? expf(sc[j] - m) : 0.f; sum += sc[j]; }
      const float inv = 1.f / sum;
      float pc = 0.f;                               // this lane's column
#pragma unroll
      for (int j = 0; j < ST; ++j) if (c == j) pc = sc[j] * inv;
      const long long prow = (bh * S + i) * S;
      if (c < S && p.P) p.P[prow + c] = pc;
      if (drop.p > 0.f) pc = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)(prow + c), drop.p) ? pc * ds : 0.f;
      float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < ST; ++j) {
        const float pv = __shfl(pc, gbase + j, 64);      // 0 for j >= S
        o.x = fmaf(pv, v[j].x, o.x); o.y = fmaf(pv, v[j].y, o.y); o.z = fmaf(pv, v[j].z, o.z); o.w = fmaf(pv, v[j].w, o.w);
      }
      ATTN_ST(p, Ob + i * s.o_rs, o);
    }
  }
}

template <int ST>
__global__ __launch_bounds__(64 * WAVES) void attn_bwd_sx4_kernel(AttnPtrs p, AttnShape s, MansyDrop drop, int accum_kv) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int hg = s.H >> 2;
  const long long g = (long long)blockIdx.x * WAVES + wave;
  if (g >= (long long)s.nb * hg) return;
  const int b = (int)(g / hg), h = (int)(g % hg) * 4 + (lane >> 4), c = lane & 15, gbase = lane & 48;
  const long long bh = (long long)b * s.H + h;
  const int S = s.Lk;
  const int col = h * 64 + c * 4;
  const float* Qb = p.Q + b * s.q_bs + col;
  const float* Kb = p.K + b * s.k_bs + col;
  const float* Vb = p.V + b * s.v_bs + col;
  const float* Gb = p.dO + b * s.o_bs + col;
  float4 k[ST], v[ST], go[ST];
  float pc[ST];                                     // P[i][c]: this lane's column of every row
  const int cc = min(c, S - 1);
#pragma unroll
  for (int j = 0; j < ST; ++j) {
    const int jc = min(j, S - 1);
    k[j] = *reinterpret_cast<const float4*>(Kb + jc * s.k_rs);
    v[j] = *reinterpret_cast<const float4*>(Vb + jc * s.v_rs);
    go[j] = *reinterpret_cast<const float4*>(Gb + jc * s.o_rs);
    pc[j] = p.P[(bh * S + jc) * S + cc];
  }
  const float ds = drop.p > 0.f ? 1.f / (1.f - drop.p) : 1.f;
  float dSc[ST], Pdc[ST];                           // dS[i][c], dropped P[i][c]
  float* dQb = p.dQ + b * s.q_bs + col;
#pragma unroll
  for (int i = 0; i < ST; ++i) {
    dSc[i] = 0.f; Pdc[i] = 0.f;
    if (i < S) {
      float dpc = 0.f;
#pragma unroll
      for (int j = 0; j < ST; ++j) {
        const float t = group16_sum(dot4(go[i], v[j]));
        if (c == j) dpc = t;
      }
      float kc = 1.f;
      if (drop.p > 0.f) kc = mansy_keep(drop.seed, drop.site, drop.base + (uint32_t)((bh * S + i) * S + c), drop.p) ? ds : 0.f;
      const float Pc = c < S ? pc[i] : 0.f;
      const float dPk = dpc * kc;
      const float delta = group16_sum(Pc * dPk);
      dSc[i] = Pc * (dPk - delta) * s.scale;
      Pdc[i] = Pc * kc;
      float4 dq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int j = 0; j < ST; ++j) {
        const float a = __shfl(dSc[i], gbase + j, 64);
        dq.x = fmaf(a, k[j].x, dq.x); dq.y = fmaf(a, k[j].y, dq.y); dq.z = fmaf(a, k[j].z, dq.z); dq.w = fmaf(a, k[j].w, dq.w);
      }
      ATTN_ST(p, dQb + i * s.q_rs, dq);
    }
  }
  // K / V rows are dead from here; the Q rows take their registers (fetched only now: all five row sets at once would
  // need > 256 VGPRs at S = 10 and halve the occupancy)
  asm volatile("" ::: "memory");
  float4 q[ST];
#pragma unroll
  for (int j = 0; j < ST; ++j) q[j] = *reinterpret_cast<const float4*>(Qb + min(j, S - 1) * s.q_rs);
  float* dKb = p.dK + b * s.k_bs + col;
  float* dVb = p.dV + b * s.v_bs + col;
#pragma unroll
  for (int j = 0; j < ST; ++j) {
    if (j < S) {
      float4 dk = make_float4(0.f, 0.f, 0.f, 0.f), dv = dk;
      if (accum_kv) {
        dk = *reinterpret_cast<const float4*>(dKb + j * s.k_rs);
        dv = *reinterpret_cast<const float4*>(dVb + j * s.v_rs);
      }
#pragma unroll
      for (int i = 0; i < ST; ++i) {               // rows i >= S carry zeros
        const float a = __shfl(dSc[i], gbase + j, 64), bb = __shfl(Pdc[i], gbase + j, 64);
        dk.x = fmaf(a, q[i].x, dk.x); dk.y = fmaf(a, q[i].y, dk.y); dk.z = fmaf(a, q[i].z, dk.z); dk.w = fmaf(a, q[i].w, dk.w);
        dv.x = fmaf(bb, go[i].x, dv.x); dv.y = fmaf(bb, go[i].y, dv.y); dv.z = fmaf(bb, go[i].z, dv.z); dv.w = fmaf(bb, go[i].w, dv.w);
      }
      ATTN_ST(p, dKb + j * s.k_rs, dk);
      ATTN_ST(p, dVb + j * s.v_rs, dv);
    }
  }
}

constexpr int SX4_MAX = 10;     // 4 x S float4 row sets + coefficients fit the register file up to here
#define MANSY_SX4_DISPATCH(KERNEL, S_, ...)                                                                \
  switch (S_) {                                                                                            \
    case 2: MANSY_LAUNCH(KERNEL<2>, __VA_ARGS__); break;                                             \
    case 3: MANSY_LAUNCH(KERNEL<3>, __VA_ARGS__); break;                                             \
    case 4: MANSY_LAUNCH(KERNEL<4>, __VA_ARGS__); break;                                             \
    case 5: MANSY_LAUNCH(KERNEL<5>, __VA_ARGS__); break;                                             \
    case 6: MANSY_LAUNCH(KERNEL<6>, __VA_ARGS__); break;                                             \
    case 7: MANSY_LAUNCH(KERNEL<7>, __VA_ARGS__); break;                                             \
    case 8: MANSY_LAUNCH(KERNEL<8>, __VA_ARGS__); break;                                             \
    case 9: MANSY_LAUNCH(KERNEL<9>, __VA_ARGS__); break;                                             \
    default: MANSY_LAUNCH(KERNEL<10>, __VA_ARGS__); break;                                           \
  }
static bool sx4_ok(const AttnShape& s, const void* a, const void* b, const void* c, const void* d) {
  auto al = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  return s.Lq == s.Lk && s.Lk >= 2 && s.Lk <= SX4_MAX && s.dh == 64 && (s.H % 4) == 0 && (s.q_bs % 4) == 0 && (s.q_rs % 4) == 0 &&
         (s.k_bs % 4) == 0 && (s.k_rs % 4) == 0 && (s.v_bs % 4) == 0 && (s.v_rs % 4) == 0 && (s.o_bs % 4) == 0 && (s.o_rs % 4) == 0 &&
         al(a) && al(b) && al(c) && al(d);
}

// LKT buckets: exact for the decode lengths the engine produces (1..10), then 12 and 16
#define MANSY_Q1X4_DISPATCH(KERNEL, Lk, ...)                                                               \
  switch (Lk) {                                                                                            \
    case 1: MANSY_LAUNCH(KERNEL<1>, __VA_ARGS__); break;                                             \
    case 2: MANSY_LAUNCH(KERNEL<2>, __VA_ARGS__); break;                                             \
    case 3: MANSY_LAUNCH(KERNEL<3>, __VA_ARGS__); break;                                             \
    case 4: MANSY_LAUNCH(KERNEL<4>, __VA_ARGS__); break;                                             \
    case 5: MANSY_LAUNCH(KERNEL<5>, __VA_ARGS__); break;                                             \
    case 6: MANSY_LAUNCH(KERNEL<6>, __VA_ARGS__); break;                                             \
    case 7: MANSY_LAUNCH(KERNEL<7>, __VA_ARGS__); break;                                             \
    case 8: MANSY_LAUNCH(KERNEL<8>, __VA_ARGS__); break;                                             \
    case 9: MANSY_LAUNCH(KERNEL<9>, __VA_ARGS__); break;                                             \
    case 10: MANSY_LAUNCH(KERNEL<10>, __VA_ARGS__); break;                                           \
    case 11: case 12: MANSY_LAUNCH(KERNEL<12>, __VA_ARGS__); break;                                  \
    default: MANSY_LAUNCH(KERNEL<16>, __VA_ARGS__); break;                                           \
  }

static bool q1x4_ok(const AttnShape& s, const void* a, const void* b, const void* c, const void* d) {
  auto al = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  return s.Lq == 1 && s.dh == 64 && (s.H % 4) == 0 && (s.q_bs % 4) == 0 && (s.k_bs % 4) == 0 && (s.k_rs % 4) == 0 && (s.v_bs % 4) == 0 &&
         (s.v_rs % 4) == 0 && (s.o_bs % 4) == 0 && al(a) && al(b) && al(c) && al(d);
}

int check_shape(const AttnShape& s) {
  MANSY_REQUIRE(s.Lq >= 1 && s.Lq <= LMAX && s.Lk >= 1 && s.Lk <= LMAX, "attn: sequence length %d x %d outside [1,%d]", s.Lq, s.Lk, LMAX);
  MANSY_REQUIRE(s.dh >= 1 && s.dh <= 64, "attn: head dim %d outside [1,64]", s.dh);
  MANSY_REQUIRE(s.nb >= 0 && s.H >= 1, "attn: bad batch/head count");
  return MANSY_OK;
}

}  // namespace

int mansy_launch_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* P_save, const AttnShape& s,
                          MansyDrop drop, hipStream_t st, const float* img_f, unsigned short* img_s, int img_only, int kv16) {
  int rc = check_shape(s); if (rc) return rc;
  MANSY_REQUIRE(Q && K && V && O, "attn_fwd: null pointer");
  MANSY_REQUIRE(!kv16 || q1x4_ok(s, Q, K, V, O), "attn_fwd: the bf16 K/V cache is read by the 4-heads-per-wave Lq = 1 kernel only");
  const long long n = (long long)s.nb * s.H;
  if (n == 0) return MANSY_OK;
  AttnPtrs p = {Q, K, V, O, P_save, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, img_f, img_s, img_s ? img_only : 0, img_s ? kv16 : 0};
  MANSY_REQUIRE(!img_s || q1x4_ok(s, Q, K, V, O) || sx4_ok(s, Q, K, V, O), "attn_fwd: the bf16 image needs the 4-heads-per-wave kernels");
  if (q1x4_ok(s, Q, K, V, O))
    MANSY_Q1X4_DISPATCH(attn_fwd_q1x4_kernel, s.Lk, dim3(mansy_ceil_div(n / 4, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop)
  else if (sx4_ok(s, Q, K, V, O))
    MANSY_SX4_DISPATCH(attn_fwd_sx4_kernel, s.Lk, dim3(mansy_ceil_div(n / 4, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop)
  else if (s.Lq == 1) MANSY_LAUNCH(attn_fwd_q1_kernel, dim3(mansy_ceil_div(n, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop);
  else MANSY_LAUNCH(attn_fwd_kernel, dim3(mansy_ceil_div(n, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_launch_attn_bwd(const float* Q, const float* K, const float* V, const float* P_save, const float* dO, float* dQ,
                          float* dK, float* dV, const AttnShape& s, MansyDrop drop, int accum_kv, hipStream_t st, const float* img_f, unsigned short* img_s, int img_only) {
  int rc = check_shape(s); if (rc) return rc;
  MANSY_REQUIRE(Q && K && V && P_save && dO && dQ && dK && dV, "attn_bwd: null pointer");
  const long long n = (long long)s.nb * s.H;
  if (n == 0) return MANSY_OK;
  AttnPtrs p = {Q, K, V, nullptr, const_cast<float*>(P_save), dO, dQ, dK, dV, nullptr, nullptr, img_f, img_s, (img_s && !accum_kv) ? img_only : 0, 0};
  MANSY_REQUIRE(!img_s || (sx4_ok(s, Q, K, V, dO) && sx4_ok(s, dQ, dK, dV, dO) && !q1x4_ok(s, Q, K, V, dO)), "attn_bwd: the bf16 image is kept by the encoder (S x S) kernel only");
  if (q1x4_ok(s, Q, K, V, dO) && q1x4_ok(s, dQ, dK, dV, dO))
    MANSY_Q1X4_DISPATCH(attn_bwd_q1x4_kernel, s.Lk, dim3(mansy_ceil_div(n / 4, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop, accum_kv)
  else if (sx4_ok(s, Q, K, V, dO) && sx4_ok(s, dQ, dK, dV, dO))
    MANSY_SX4_DISPATCH(attn_bwd_sx4_kernel, s.Lk, dim3(mansy_ceil_div(n / 4, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop, accum_kv)
  else if (s.Lq == 1) MANSY_LAUNCH(attn_bwd_q1_kernel, dim3(mansy_ceil_div(n, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop, accum_kv);
  else MANSY_LAUNCH(attn_bwd_kernel, dim3(mansy_ceil_div(n, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop, accum_kv);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// ---- deferred K/V gradients (decoder cross-attention).  Shape-only test; pointers must be 16-byte aligned.
int mansy_attn_deferred_kv_ok(const AttnShape& s, int T) {
  return s.Lq == 1 && s.dh == 64 && (s.H % 4) == 0 && s.Lk >= 1 && s.Lk <= LMAX && T >= 1 && T <= LMAX && (s.q_bs % 4) == 0 &&
         (s.k_bs % 4) == 0 && (s.k_rs % 4) == 0 && (s.v_bs % 4) == 0 && (s.v_rs % 4) == 0 && (s.o_bs % 4) == 0;
}
// One step: dQ only; dS_out / Pk_out [nb*H, Lk] receive the coefficients mansy_launch_attn_kvgrad sums over the steps.
int mansy_launch_attn_bwd_dq(const float* Q, const float* K, const float* V, const float* P_save, const float* dO, float* dQ,
                             float* dS_out, float* Pk_out, const AttnShape& s, MansyDrop drop, hipStream_t st, const float* img_f, unsigned short* img_s, int img_only, int kv16) {
  int rc = check_shape(s); if (rc) return rc;
  MANSY_REQUIRE(Q && K && V && P_save && dO && dQ && dS_out && Pk_out, "attn_bwd_dq: null pointer");
  MANSY_REQUIRE(q1x4_ok(s, Q, K, V, dO) && q1x4_ok(s, dQ, K, V, dO), "attn_bwd_dq: shape / alignment not on the 4-heads-per-wave path");
  const long long n = (long long)s.nb * s.H;
  if (n == 0) return MANSY_OK;
  AttnPtrs p = {Q, K, V, nullptr, const_cast<float*>(P_save), dO, dQ, nullptr, nullptr, dS_out, Pk_out, img_f, img_s, img_s ? img_only : 0, img_s ? kv16 : 0};
  MANSY_Q1X4_DISPATCH(attn_bwd_q1x4_kernel, s.Lk, dim3(mansy_ceil_div(n / 4, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop, 0)
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
// Q_all / dO_all: step i at + i*q_ts / + i*o_ts (batch strides from s); dS_all / Pk_all: [T][nb*H][Lk].
int mansy_launch_attn_kvgrad(const float* Q_all, long long q_ts, const float* dO_all, long long o_ts, const float* dS_all,
                             const float* Pk_all, float* dK, float* dV, const AttnShape& s, int T, int accum, hipStream_t st, const float* img_f,
                             unsigned short* img_s) {
  MANSY_REQUIRE(Q_all && dO_all && dS_all && Pk_all && dK && dV, "attn_kvgrad: null pointer");
  MANSY_REQUIRE(mansy_attn_deferred_kv_ok(s, T) && (q_ts % 4) == 0 && (o_ts % 4) == 0, "attn_kvgrad: unsupported shape");
  auto al = [](const void* x) { return (reinterpret_cast<uintptr_t>(x) & 15) == 0; };
  MANSY_REQUIRE(al(Q_all) && al(dO_all) && al(dK) && al(dV), "attn_kvgrad: pointers must be 16-byte aligned");
  const long long n = (long long)s.nb * s.H;
  if (n == 0) return MANSY_OK;
#define MANSY_KVGRAD_ARGS dim3(mansy_ceil_div(n / 4, WAVES)), dim3(64 * WAVES), 0, st, Q_all, q_ts, dO_all, o_ts, dS_all, Pk_all, dK, dV, s, T, accum, img_f, img_s
  MANSY_Q1X4_DISPATCH(attn_kvgrad_q1x4_kernel, T, MANSY_KVGRAD_ARGS)
#undef MANSY_KVGRAD_ARGS
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// ---- decoder self-attention backward, pull form (see attn_bwd_selfpull_q1x4_kernel).  s.Lk must equal step + 1.
// Q_all / dO_all: step i at + i*q_ts / + i*o_ts floats; dS_all / Pk_all: [T][nb*H][T] (rows > step already written by the
// calls for the later steps; this call writes row `step`).  dQ: this step's [nb, H*64]; dK / dV: K/V gradient slabs, row `step`
// is overwritten.
int mansy_attn_selfpull_ok(const AttnShape& s, int T) { return mansy_attn_deferred_kv_ok(s, T); }
int mansy_launch_attn_bwd_selfpull(const float* Q_all, long long q_ts, const float* K, const float* V, const float* P_save,
                                   const float* dO_all, long long o_ts, float* dQ, float* dK, float* dV, float* dS_all, float* Pk_all,
                                   const AttnShape& s, int T, int step, MansyDrop drop, hipStream_t st, const float* img_f, unsigned short* img_s, int img_only, int kv16) {
  int rc = check_shape(s); if (rc) return rc;
  MANSY_REQUIRE(Q_all && K && V && P_save && dO_all && dQ && dK && dV && dS_all && Pk_all, "attn_bwd_selfpull: null pointer");
  MANSY_REQUIRE(step >= 0 && step < T && s.Lk == step + 1 && mansy_attn_selfpull_ok(s, T) && (q_ts % 4) == 0 && (o_ts % 4) == 0,
                "attn_bwd_selfpull: unsupported shape (Lk=%d step=%d T=%d)", s.Lk, step, T);
  MANSY_REQUIRE(q1x4_ok(s, Q_all, K, V, dO_all) && q1x4_ok(s, dQ, dK, dV, dO_all), "attn_bwd_selfpull: alignment");
  const long long n = (long long)s.nb * s.H;
  if (n == 0) return MANSY_OK;
  AttnPtrs p = {nullptr, K, V, nullptr, const_cast<float*>(P_save), nullptr, dQ, dK, dV, nullptr, nullptr, img_f, img_s, img_s ? img_only : 0, img_s ? kv16 : 0};
  SelfPull sp = {Q_all, q_ts, dO_all, o_ts, dS_all, Pk_all, T, step};
  MANSY_Q1X4_DISPATCH(attn_bwd_selfpull_q1x4_kernel, s.Lk, dim3(mansy_ceil_div(n / 4, WAVES)), dim3(64 * WAVES), 0, st, p, s, drop, sp)
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}
