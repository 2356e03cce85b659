// A2C baseline engine: the "simple RL" actor-critic of the reference's comparison tables, each C-ABI call one
// kernel sequence on one HIP stream.
//
// Reference semantics: FeatureNet/Actor/Critic (bitrate_selection/models/simple_rl.py:9-63), SimpleRLEnv's observation
// (envs/simple_rl_env.py:85-170), tianshou==0.4.8 A2CPolicy.learn + torch.optim.RMSprop behind run_simple_rl.py:190-211
// (T2, restated in oracle/a2c_oracle.py).
//
// Structure: an observation is one 416-float row [throughput 0:8 | chunk_sizes 8:328 | rebuffer 328 | last_bitrates 329:331 |
// pred_viewport 331:395 | zero pad]; the five branches are dense layers on disjoint column ranges, so the FeatureNet is ONE
// block-diagonal product F = LeakyReLU(obs[B,416] * Wbd[640,416]^T + b) on the fp32 MFMA GEMM (structurally-zero K-tiles
// skipped); actor and critic share F and run their fc layers as one stacked [640 -> 256] product; the 128 -> {15, 1} output
// layers, softmax, sampling, loss and its gradient are row-wise kernels; every weight gradient is a dY^T X product.
#include <string>
#include <vector>
#include "mansy_kernels.h"
#include "../../include/mansy_hip.h"

namespace {

constexpr int HID = 128, NB = 5, FEAT = HID * NB, LD = MANSY_A2C_OBS_LD, NACT = 15, MAXOUT = 16;
constexpr float SLOPE = 0.01f;
constexpr int NPARAM = 2 * NB + 8;
static_assert(LD % 32 == 0, "packed K must be whole K-tiles");

struct Branch { int off, len; };
__host__ __device__ inline Branch branch_geom(int j) {
  const int off[NB] = {MANSY_A2C_O_THROUGHPUT, MANSY_A2C_O_SIZE, MANSY_A2C_O_REBUFFER, MANSY_A2C_O_LAST_RATES, MANSY_A2C_O_PRED_VP};
  const int len[NB] = {8, 320, 1, 2, 64};
  Branch b; b.off = off[j]; b.len = len[j];
  return b;
}

struct Net {
  const float* bw[NB]; const float* bb[NB];
  const float* fc_w[2]; const float* fc_b[2]; const float* out_w[2]; const float* out_b[2];      // [0] actor, [1] critic
  float* gbw[NB]; float* gbb[NB]; float* gfc_w[2]; float* gfc_b[2]; float* gout_w[2]; float* gout_b[2];
};
void bind(const float* const* p, float* const* g, Net& n) {
  for (int j = 0; j < NB; ++j) { n.bw[j] = p[2 * j]; n.bb[j] = p[2 * j + 1]; n.gbw[j] = g ? g[2 * j] : nullptr; n.gbb[j] = g ? g[2 * j + 1] : nullptr; }
  for (int h = 0; h < 2; ++h) {
    const int b = 2 * NB + 4 * h;
    n.fc_w[h] = p[b]; n.fc_b[h] = p[b + 1]; n.out_w[h] = p[b + 2]; n.out_b[h] = p[b + 3];
    n.gfc_w[h] = g ? g[b] : nullptr; n.gfc_b[h] = g ? g[b + 1] : nullptr; n.gout_w[h] = g ? g[b + 2] : nullptr; n.gout_b[h] = g ? g[b + 3] : nullptr;
  }
}

struct ParamInfo { std::string name; long long numel; int ndim; long long shape[4]; };
std::vector<ParamInfo> table() {
  std::vector<ParamInfo> v;
  auto add = [&](const std::string& n, long long a, long long b = 0, long long c = 0) {
    ParamInfo p; p.name = n; p.shape[0] = a; p.shape[1] = b; p.shape[2] = c; p.shape[3] = 0; p.ndim = c ? 3 : (b ? 2 : 1);
    p.numel = a * (b ? b : 1) * (c ? c : 1); v.push_back(p);
  };
  const std::string f = "actor.feature_net.";
  add(f + "conv1d_1.0.weight", HID, 1, 8); add(f + "conv1d_1.0.bias", HID);
  add(f + "conv1d_2.0.weight", HID, 1, 320); add(f + "conv1d_2.0.bias", HID);
  add(f + "fc1.0.weight", HID, 1); add(f + "fc1.0.bias", HID);
  add(f + "fc2.0.weight", HID, 2); add(f + "fc2.0.bias", HID);
  add(f + "fc3.0.weight", HID, 64); add(f + "fc3.0.bias", HID);
  add("actor.fc.0.weight", HID, FEAT); add("actor.fc.0.bias", HID); add("actor.out.weight", NACT, HID); add("actor.out.bias", NACT);
  add("critic.fc.0.weight", HID, FEAT); add("critic.fc.0.bias", HID); add("critic.out.weight", 1, HID); add("critic.out.bias", 1);
  return v;
}

// ------------------------------------------------------------------------------------ kernels
struct PackArgs {
  const float* bw[NB]; const float* bb[NB]; const float* fc_w[2]; const float* fc_b[2];
  const float* g_src; const int* g_idx; float* g_dst; int g_rows; float* zero_ptr; long long zero_n;     // riders: row gather, zero-fill
};
// Wbd [FEAT, LD] (zero off the block diagonal), bbd [FEAT], per 64-feature column tile the K range holding its weights,
// Wfc2 [2*HID, FEAT] / bfc2 [2*HID] = actor.fc stacked on critic.fc.
__global__ __launch_bounds__(256) void a2c_pack_kernel(PackArgs a, float* __restrict__ Wbd, float* __restrict__ bbd, int* __restrict__ krange,
                                                       float* __restrict__ Wfc2, float* __restrict__ bfc2) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < FEAT / 64) {
    const Branch g = branch_geom((int)idx * 64 / HID);
    krange[2 * idx] = g.off / 32 * 32;
    krange[2 * idx + 1] = min(LD, (g.off + g.len + 31) / 32 * 32);
  }
  if (idx < 2 * HID) bfc2[idx] = a.fc_b[idx / HID][idx % HID];
  if (idx < (long long)FEAT * LD) {
    const int col = (int)(idx % LD), row = (int)(idx / LD);
    const int j = row / HID, r = row % HID;
    const Branch g = branch_geom(j);
    Wbd[idx] = (col >= g.off && col < g.off + g.len) ? a.bw[j][r * g.len + (col - g.off)] : 0.f;
    if (col == 0) bbd[row] = a.bb[j][r];
    return;
  }
  long long i2 = idx - (long long)FEAT * LD;
  if (i2 < 2LL * HID * FEAT) { Wfc2[i2] = a.fc_w[i2 / ((long long)HID * FEAT)][i2 % ((long long)HID * FEAT)]; return; }
  i2 -= 2LL * HID * FEAT;
  const long long n_g = (long long)a.g_rows * (LD / 4);
  if (i2 < n_g) {
    const int r = (int)(i2 / (LD / 4)), c4 = (int)(i2 % (LD / 4));
    reinterpret_cast<float4*>(a.g_dst)[(size_t)r * (LD / 4) + c4] = reinterpret_cast<const float4*>(a.g_src)[(size_t)a.g_idx[r] * (LD / 4) + c4];
    return;
  }
  i2 -= n_g;
  if (i2 < (a.zero_n + 3) / 4) {
    const long long e0 = i2 * 4;
    if (e0 + 4 <= a.zero_n) *reinterpret_cast<float4*>(a.zero_ptr + e0) = make_float4(0.f, 0.f, 0.f, 0.f);
    else for (long long e = e0; e < a.zero_n; ++e) a.zero_ptr[e] = 0.f;
  }
}

// Output layers on H2 [B, 2*HID] (actor half | critic half): probs = softmax(H_a Wa^T + ba) [B,16] (col 15 zero), value [B];
// optional Categorical(probs).sample() by inverse CDF on the renormalised probabilities + log_prob of the sample.
__global__ __launch_bounds__(256) void a2c_out_kernel(const float* __restrict__ H2, const float* __restrict__ Wa, const float* __restrict__ ba,
                                                      const float* __restrict__ Wc, const float* __restrict__ bc, int rows,
                                                      float* __restrict__ probs, float* __restrict__ value, const float* __restrict__ u_ext,
                                                      uint32_t seed, uint32_t site, int* __restrict__ act, float* __restrict__ logp) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  const float* h = H2 + (size_t)row * 2 * HID;
  const float a0 = h[lane], a1 = h[64 + lane], c0 = h[HID + lane], c1 = h[HID + 64 + lane];
  float w0[NACT], w1[NACT];
#pragma unroll
  for (int k = 0; k < NACT; ++k) { w0[k] = Wa[k * HID + lane]; w1[k] = Wa[k * HID + 64 + lane]; }
  const float v = wave_sum(c0 * Wc[lane] + c1 * Wc[64 + lane]) + bc[0];
  float z[NACT], m = -INFINITY;
#pragma unroll
  for (int k = 0; k < NACT; ++k) { z[k] = wave_sum(a0 * w0[k] + a1 * w1[k]) + ba[k]; m = fmaxf(m, z[k]); }
  float p[NACT], s = 0.f;
#pragma unroll
  for (int k = 0; k < NACT; ++k) { p[k] = expf(z[k] - m); s += p[k]; }
#pragma unroll
  for (int k = 0; k < NACT; ++k) p[k] = p[k] / s;
  if (probs && lane < MAXOUT) {
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) if (k == lane) mine = p[k];
    probs[(size_t)row * MAXOUT + lane] = mine;
  }
  if (value && lane == 0) value[row] = v;
  if (act) {
    float ps = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) ps += p[k];
    const float u = u_ext ? u_ext[row] : mansy_uniform01(seed, site, (uint32_t)row);
    float c = 0.f, pa = 0.f; int a = NACT - 1; bool found = false;
#pragma unroll
    for (int k = 0; k < NACT; ++k) { c += p[k] / ps; if (!found && c > u) { a = k; found = true; } }
#pragma unroll
    for (int k = 0; k < NACT; ++k) if (k == a) pa = p[k] / ps;
    if (lane == 0) {
      act[row] = a;
      if (logp) logp[row] = logf(fminf(fmaxf(pa, 1.1920929e-07f), 1.f - 1.1920929e-07f));
    }
  }
}

// A2CPolicy.learn loss for one minibatch (T2) with dist = torch.distributions.Categorical(probs):
//   pn = p / sum(p); L = log(clamp(pn, eps, 1 - eps)); log_prob = L[act]; entropy = -sum(L * pn)
//   loss = -(log_prob * adv).mean() + vf_coef * mean((ret - v)^2) - ent_coef * entropy.mean()
// and its gradient wrt the PRE-softmax logits (g[:, 0:15]) and the value (g[:, 15]).  stats: [loss, actor, vf, entropy].
struct LossArgs {
  const float* probs; const float* value; const int* act; const float* adv; const float* ret; const int* idx; int n; float vf_coef, ent_coef;
  float* g; float* stats;
};
__device__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}
__global__ __launch_bounds__(1024) void a2c_loss_kernel(LossArgs a) {
  __shared__ float sh[16];
  const float EPS = 1.1920929e-07f;
  const int n = a.n;
  float l_act = 0.f, l_vf = 0.f, l_ent = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int bi = a.idx ? a.idx[i] : i;
    float pr[MAXOUT];
#pragma unroll
    for (int q4 = 0; q4 < MAXOUT / 4; ++q4)
      *reinterpret_cast<float4*>(pr + 4 * q4) = *reinterpret_cast<const float4*>(a.probs + (size_t)i * MAXOUT + 4 * q4);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) s += pr[k];
    const int act = a.act[bi];
    const float adv = a.adv[bi];
    float pn[NACT], L[NACT], in[NACT], ent = 0.f, lp = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) {
      pn[k] = pr[k] / s;
      in[k] = (pn[k] >= EPS && pn[k] <= 1.f - EPS) ? 1.f : 0.f;       // clamp passes the gradient inside [eps, 1 - eps]
      L[k] = logf(fminf(fmaxf(pn[k], EPS), 1.f - EPS));
      ent -= L[k] * pn[k];
      if (k == act) lp = L[k];
    }
    l_act += -lp * adv;
    l_ent += ent;
    // d loss / d pn_k
    float gpn[NACT], dot = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) {
      float g = 0.f;
      if (k == act) g += -adv * in[k] / fmaxf(pn[k], EPS);
      g += a.ent_coef * (L[k] + in[k] * pn[k] / fmaxf(pn[k], EPS));     // -ent_coef * dH/dpn_k, dH/dpn_k = -(L_k + pn_k dL_k/dpn_k)
      gpn[k] = g / (float)n;
      dot += gpn[k] * pn[k];
    }
    // pn = p / s  ->  d/dp_j = (gpn_j - sum_k gpn_k pn_k) / s ; then softmax backward with p (the stored probabilities)
    float gp[NACT], dot2 = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) { gp[k] = (gpn[k] - dot) / s; dot2 += gp[k] * pr[k]; }
    float gout[MAXOUT];
#pragma unroll
    for (int k = 0; k < NACT; ++k) gout[k] = pr[k] * (gp[k] - dot2);
    const float v = a.value[i], ret = a.ret[bi];
    l_vf += (ret - v) * (ret - v);
    gout[NACT] = a.vf_coef * 2.f * (v - ret) / (float)n;
#pragma unroll
    for (int q4 = 0; q4 < MAXOUT / 4; ++q4)
      *reinterpret_cast<float4*>(a.g + (size_t)i * MAXOUT + 4 * q4) = *reinterpret_cast<const float4*>(gout + 4 * q4);
  }
  const float al = block_sum(l_act, sh) / (float)n;
  const float vf = block_sum(l_vf, sh) / (float)n;
  const float em = block_sum(l_ent, sh) / (float)n;
  if (threadIdx.x == 0 && a.stats) { a.stats[0] = al + a.vf_coef * vf - a.ent_coef * em; a.stats[1] = al; a.stats[2] = vf; a.stats[3] = em; }
}

// dH2[r, c] = (c < HID ? sum_k g[r,k] Wa[k,c] : g[r,15] Wc[c-HID]) * LeakyReLU'(H2[r,c])
__global__ __launch_bounds__(256) void a2c_out_bwd_kernel(const float* __restrict__ g, const float* __restrict__ H2, const float* __restrict__ Wa,
                                                          const float* __restrict__ Wc, int rows, float* __restrict__ dH2) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)rows * 2 * HID) return;
  const int r = (int)(idx / (2 * HID)), c = (int)(idx % (2 * HID));
  const float* gr = g + (size_t)r * MAXOUT;
  float d;
  if (c < HID) {
    d = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) d += gr[k] * Wa[k * HID + c];
  } else d = gr[NACT] * Wc[c - HID];
  dH2[idx] = H2[idx] > 0.f ? d : d * SLOPE;
}

// scatter of the packed gradients into the reference-layout parameter gradients:
//   dWbd [FEAT, LD] block-diagonal entries + dbbd [FEAT] -> the five branches; Gout [16, 2*HID] = g^T H2 and its row sums
//   gsum [16] -> actor.out (rows 0..14, columns 0..127) and critic.out (row 15, columns 128..255).
struct UnpackArgs { float* gbw[NB]; float* gbb[NB]; float* gout_w[2]; float* gout_b[2]; };
__global__ __launch_bounds__(256) void a2c_unpack_kernel(const float* __restrict__ dWbd, const float* __restrict__ dbbd, const float* __restrict__ Gout,
                                                         const float* __restrict__ gsum, UnpackArgs a) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < FEAT) a.gbb[idx / HID][idx % HID] += dbbd[idx];
  if (idx < NACT) a.gout_b[0][idx] += gsum[idx];
  if (idx == NACT) a.gout_b[1][0] += gsum[NACT];
  if (idx < (long long)MAXOUT * 2 * HID) {
    const int k = (int)(idx / (2 * HID)), c = (int)(idx % (2 * HID));
    if (k < NACT && c < HID) a.gout_w[0][k * HID + c] += Gout[idx];
    if (k == NACT && c >= HID) a.gout_w[1][c - HID] += Gout[idx];
  }
  if (idx >= (long long)FEAT * LD) return;
  const int col = (int)(idx % LD), row = (int)(idx / LD);
  const int j = row / HID, r = row % HID;
  const Branch g = branch_geom(j);
  if (col >= g.off && col < g.off + g.len) a.gbw[j][r * g.len + (col - g.off)] += dWbd[idx];
}

constexpr int NORM_PARTS = MANSY_CLIP_SCRATCH_DOUBLES;
__global__ __launch_bounds__(256) void a2c_sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ parts) {
  __shared__ double red[256];
  double local = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) { const double t = g[i]; local += t * t; }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = red[0];
}
// clip_grad_norm_(max_norm) folded into torch.optim.RMSprop(lr, alpha, eps): sq = alpha sq + (1 - alpha) g^2 ; p -= lr g / (sqrt(sq) + eps).
// apply == 0: only scale the stored gradient (parity tests read the clipped gradient).
__global__ __launch_bounds__(256) void a2c_clip_rmsprop_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ sq, long long n, float lr,
                                                               float alpha, float eps, const double* __restrict__ parts, float max_norm, int apply) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float coef = 1.f;
  if (max_norm > 0.f) {
    double t = 0.0;
    for (int k = 0; k < NORM_PARTS; ++k) t += parts[k];
    coef = max_norm / ((float)sqrt(t) + 1e-6f);
    coef = coef < 1.f ? coef : 1.f;
  }
  const float grad = g[i] * coef;
  if (!apply) { g[i] = grad; return; }
  const float s = sq[i] * alpha + (1.f - alpha) * grad * grad;
  sq[i] = s;
  p[i] = p[i] - lr * (grad / (sqrtf(s) + eps));
}

// SimpleRLEnv observation rows from the MANSYEnv kernel's outputs of the same step (envs/simple_rl_env.py:112-118,148-165)
__global__ __launch_bounds__(256) void a2c_obs_kernel(const float* __restrict__ obs, const float* __restrict__ qoe_parts, const int* __restrict__ actions,
                                                      const unsigned char* __restrict__ fresh, int n, int r0, int r1, int r2, int r3, int r4,
                                                      float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)n * LD) return;
  const int e = (int)(idx / LD), c = (int)(idx % LD);
  const float* row = obs + (size_t)e * MANSY_OBS_LD;
  const bool is_fresh = !actions || (fresh && fresh[e]);
  float v = 0.f;
  if (c < MANSY_A2C_O_SIZE) v = row[MANSY_O_THROUGHPUT + c];
  else if (c < MANSY_A2C_O_REBUFFER) v = row[MANSY_O_SIZE + (c - MANSY_A2C_O_SIZE)];
  else if (c >= MANSY_A2C_O_PRED_VP && c < MANSY_A2C_O_PRED_VP + 64) v = row[MANSY_O_PRED_VP + (c - MANSY_A2C_O_PRED_VP)];
  else if (c < MANSY_A2C_O_PRED_VP && !is_fresh) {
    if (c == MANSY_A2C_O_REBUFFER) v = qoe_parts[4 * e + 2];
    else {
      const int A2R[15][2] = {{1,0},{2,0},{3,0},{4,0},{2,1},{3,1},{4,1},{3,2},{4,2},{4,3},{0,0},{1,1},{2,2},{3,3},{4,4}};
      const int a = actions[e];
      const int ri = (a >= 0 && a < 15) ? A2R[a][c - MANSY_A2C_O_LAST_RATES] : 0;
      const int rate = ri == 0 ? r0 : ri == 1 ? r1 : ri == 2 ? r2 : ri == 3 ? r3 : r4;
      v = (float)rate / (float)r4;
    }
  }
  out[idx] = v;
}

// ------------------------------------------------------------------------------------ workspace + engine
struct Work {
  float *Wbd, *bbd, *Wfc2, *bfc2, *obs_mb, *F, *H2, *probs, *value, *g, *dH2, *dF, *dWbd, *dbbd, *Gout, *gsum; int* krange; double* acc;
};
size_t layout(int maxB, char* base, Work& W) {
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  W.Wbd = (float*)take(sizeof(float) * FEAT * LD); W.bbd = (float*)take(sizeof(float) * FEAT);
  W.Wfc2 = (float*)take(sizeof(float) * 2 * HID * FEAT); W.bfc2 = (float*)take(sizeof(float) * 2 * HID);
  W.obs_mb = (float*)take(sizeof(float) * (size_t)maxB * LD);
  W.F = (float*)take(sizeof(float) * (size_t)maxB * FEAT); W.H2 = (float*)take(sizeof(float) * (size_t)maxB * 2 * HID);
  W.probs = (float*)take(sizeof(float) * (size_t)maxB * MAXOUT); W.value = (float*)take(sizeof(float) * (size_t)maxB);
  W.g = (float*)take(sizeof(float) * (size_t)maxB * MAXOUT); W.dH2 = (float*)take(sizeof(float) * (size_t)maxB * 2 * HID);
  W.dF = (float*)take(sizeof(float) * (size_t)maxB * FEAT);
  W.dWbd = (float*)take(sizeof(float) * FEAT * LD); W.dbbd = (float*)take(sizeof(float) * FEAT);
  W.Gout = (float*)take(sizeof(float) * MAXOUT * 2 * HID); W.gsum = (float*)take(sizeof(float) * MAXOUT);
  W.krange = (int*)take(sizeof(int) * 2 * (FEAT / 64)); W.acc = (double*)take(sizeof(double) * NORM_PARTS);
  return off;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

struct Eng {
  hipStream_t st; Work W;
  int prec = 0;      // MANSY_PREC_* of this call's products (the entry point's `precision` argument)
  int pack(const Net& n, const float* g_src, const int* g_idx, int g_rows, float* zero_ptr, long long zero_n) {
    PackArgs a;
    for (int j = 0; j < NB; ++j) { a.bw[j] = n.bw[j]; a.bb[j] = n.bb[j]; }
    for (int h = 0; h < 2; ++h) { a.fc_w[h] = n.fc_w[h]; a.fc_b[h] = n.fc_b[h]; }
    a.g_src = g_src; a.g_idx = g_idx; a.g_dst = W.obs_mb; a.g_rows = g_src ? g_rows : 0; a.zero_ptr = zero_ptr; a.zero_n = zero_ptr ? zero_n : 0;
    MANSY_REQUIRE(!zero_ptr || (reinterpret_cast<uintptr_t>(zero_ptr) & 15) == 0, "a2c pack: gradient buffer must be 16-byte aligned");
    const long long threads = (long long)FEAT * LD + 2LL * HID * FEAT + (long long)a.g_rows * (LD / 4) + (a.zero_n + 3) / 4;
    MANSY_LAUNCH(a2c_pack_kernel, dim3(mansy_ceil_div(threads, 256)), dim3(256), 0, st, a, W.Wbd, W.bbd, W.krange, W.Wfc2, W.bfc2);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  int forward(const Net& n, const float* obs, int B, float* probs, float* value, const float* u, uint32_t seed, uint32_t site, int* act, float* logp) {
    GemmEpilogue e1; e1.prec = prec; e1.bias = W.bbd; e1.relu = 1; e1.relu_slope = SLOPE; e1.tile_krange = W.krange;
    RC(mansy_launch_gemm_f32(obs, LD, 0, W.Wbd, LD, 0, W.F, FEAT, B, FEAT, LD, e1, 0, 0, st));
    GemmEpilogue e2; e2.prec = prec; e2.bias = W.bfc2; e2.relu = 1; e2.relu_slope = SLOPE;
    RC(mansy_launch_gemm_f32(W.F, FEAT, 0, W.Wfc2, FEAT, 0, W.H2, 2 * HID, B, 2 * HID, FEAT, e2, 0, 0, st));
    MANSY_LAUNCH(a2c_out_kernel, dim3(mansy_ceil_div(B, 4)), dim3(256), 0, st, W.H2, n.out_w[0], n.out_b[0], n.out_w[1], n.out_b[1], B, probs, value,
                       u, seed, site, act, logp);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  // g [B,16] = dL/d(logits | value) -> all parameter gradients (accumulated into zeroed buffers)
  int backward(const Net& n, const float* obs, int B) {
    MANSY_LAUNCH(a2c_out_bwd_kernel, dim3(mansy_ceil_div((long long)B * 2 * HID, 256)), dim3(256), 0, st, W.g, W.H2, n.out_w[0], n.out_w[1], B, W.dH2);
    MANSY_LAUNCH_CHECK();
    MANSY_HIP_CHECK(hipMemsetAsync(W.gsum, 0, sizeof(float) * MAXOUT, st));
    GemmEpilogue eo; eo.prec = prec; eo.a_rowsum = W.gsum;                                                                    // Gout = g^T H2, gsum = column sums of g
    RC(mansy_launch_gemm_f32(W.g, MAXOUT, 1, W.H2, 2 * HID, 1, W.Gout, 2 * HID, MAXOUT, 2 * HID, B, eo, 0, 1, st));
    GemmEpilogue acc; acc.prec = prec; acc.accumulate = 1; acc.a_rowsum = n.gfc_b[0];                                         // fc weight / bias gradients, both heads
    acc.pair_A = W.dH2 + HID; acc.pair_B = W.F; acc.pair_C = n.gfc_w[1]; acc.pair_rowsum = n.gfc_b[1];
    RC(mansy_launch_gemm_f32(W.dH2, 2 * HID, 1, W.F, FEAT, 1, n.gfc_w[0], FEAT, HID, FEAT, B, acc, 0, 0, st));
    GemmEpilogue ef; ef.prec = prec; ef.mask_src = W.F; ef.mask_ld = FEAT; ef.mask_scale = 1.f; ef.mask_neg = SLOPE;        // dPre = (dH2 Wfc2) * LeakyReLU'(F)
    RC(mansy_launch_gemm_f32(W.dH2, 2 * HID, 0, W.Wfc2, FEAT, 1, W.dF, FEAT, B, FEAT, 2 * HID, ef, 0, 0, st));
    MANSY_HIP_CHECK(hipMemsetAsync(W.dbbd, 0, sizeof(float) * FEAT, st));
    GemmEpilogue ew; ew.prec = prec; ew.a_rowsum = W.dbbd;                                                                   // dWbd = dPre^T obs
    RC(mansy_launch_gemm_f32(W.dF, FEAT, 1, obs, LD, 1, W.dWbd, LD, FEAT, LD, B, ew, 0, 1, st));
    UnpackArgs u;
    for (int j = 0; j < NB; ++j) { u.gbw[j] = n.gbw[j]; u.gbb[j] = n.gbb[j]; }
    for (int h = 0; h < 2; ++h) { u.gout_w[h] = n.gout_w[h]; u.gout_b[h] = n.gout_b[h]; }
    MANSY_LAUNCH(a2c_unpack_kernel, dim3(mansy_ceil_div((long long)FEAT * LD, 256)), dim3(256), 0, st, W.dWbd, W.dbbd, W.Gout, W.gsum, u);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  int clip_rmsprop(float* p, float* g, float* sq, long long n, float max_norm, float lr, float alpha, float eps, int apply) {
    if (max_norm > 0.f) MANSY_LAUNCH(a2c_sumsq_kernel, dim3(NORM_PARTS), dim3(256), 0, st, g, n, W.acc);
    if (!apply && max_norm <= 0.f) return MANSY_OK;
    MANSY_LAUNCH(a2c_clip_rmsprop_kernel, dim3(mansy_ceil_div(n, 256)), dim3(256), 0, st, p, g, sq, n, lr, alpha, eps, W.acc, max_norm, apply);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
};

int setup(void* ws, int maxB, int precision, hipStream_t st, Eng& e) {
  MANSY_REQUIRE(precision == 0 || precision == 1 || precision == 3 || precision == 6, "precision must be MANSY_PREC_F32 (0), _BF16 (1), _BF16X3 (3) or _BF16X6 (6), got %d", precision);
  e.prec = precision;
  MANSY_REQUIRE(ws && maxB >= 1, "a2c: bad workspace / batch");
  e.st = st;
  layout(maxB, (char*)ws, e.W);
  return MANSY_OK;
}

}  // namespace

extern "C" {

int mansy_a2c_num_params(void) { return NPARAM; }
int mansy_a2c_param_info(int idx, char* name, int name_len, long long* numel, int* ndim, long long shape[4]) {
  const std::vector<ParamInfo> t = table();
  MANSY_REQUIRE(idx >= 0 && idx < (int)t.size(), "a2c_param_info: index %d out of range", idx);
  if (name && name_len > 0) { strncpy(name, t[idx].name.c_str(), name_len - 1); name[name_len - 1] = 0; }
  if (numel) *numel = t[idx].numel;
  if (ndim) *ndim = t[idx].ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = t[idx].shape[i];
  return MANSY_OK;
}
size_t mansy_a2c_workspace_bytes(int max_batch) { Work W; return max_batch >= 1 ? layout(max_batch, nullptr, W) : 0; }

int mansy_a2c_obs(const float* obs, const float* qoe_parts, const int* actions, const unsigned char* fresh, int n, const int video_rates[5],
                  float* out, void* stream) {
  MANSY_REQUIRE(obs && out && video_rates && n >= 1 && (!actions || qoe_parts), "a2c_obs: bad arguments");
  MANSY_LAUNCH(a2c_obs_kernel, dim3(mansy_ceil_div((long long)n * LD, 256)), dim3(256), 0, (hipStream_t)stream, obs, qoe_parts, actions, fresh, n,
                     video_rates[0], video_rates[1], video_rates[2], video_rates[3], video_rates[4], out);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_a2c_forward(const float* const* params, const float* obs, int B, float* probs, float* value, int* act, float* logp, const float* u,
                      uint32_t seed, uint32_t site, int reuse_packed, void* workspace, int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && obs && B >= 1 && B <= max_batch, "a2c_forward: bad arguments (B=%d, max_batch=%d)", B, max_batch);
  Eng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  Net n; bind(params, nullptr, n);
  if (!reuse_packed) RC(e.pack(n, nullptr, nullptr, 0, nullptr, 0));
  return e.forward(n, obs, B, probs ? probs : e.W.probs, value, u, seed, site, act, logp);
}

// One A2C minibatch update (T2: A2CPolicy.learn body): gather rows idx[0..mb), forward, loss, backward, clip_grad_norm_,
// RMSprop.  apply == 0: gradients only (clipped when max_grad_norm > 0) -- parity tests and data-parallel callers.
int mansy_a2c_minibatch_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_sq, long long n_flat,
                             const float* obs_all, const int* idx, const int* act_all, const float* adv_all, const float* ret_all, int mb,
                             float vf_coef, float ent_coef, float max_grad_norm, float lr, float alpha, float eps, int apply, float* stats,
                             void* workspace, int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && grads && flat_p && flat_g && flat_sq && obs_all && act_all && adv_all && ret_all, "a2c_minibatch_step: null pointer");
  MANSY_REQUIRE(mb >= 1 && mb <= max_batch, "a2c_minibatch_step: bad minibatch size");
  Eng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  Net n; bind(params, grads, n);
  const float* obs = idx ? e.W.obs_mb : obs_all;
  RC(e.pack(n, idx ? obs_all : nullptr, idx, mb, flat_g, n_flat));
  RC(e.forward(n, obs, mb, e.W.probs, e.W.value, nullptr, 0, 0, nullptr, nullptr));
  LossArgs la;
  la.probs = e.W.probs; la.value = e.W.value; la.act = act_all; la.adv = adv_all; la.ret = ret_all; la.idx = idx; la.n = mb; la.vf_coef = vf_coef;
  la.ent_coef = ent_coef; la.g = e.W.g; la.stats = stats;
  MANSY_LAUNCH(a2c_loss_kernel, dim3(1), dim3(1024), 0, e.st, la);
  MANSY_LAUNCH_CHECK();
  RC(e.backward(n, obs, mb));
  return e.clip_rmsprop(flat_p, flat_g, flat_sq, n_flat, max_grad_norm, lr, alpha, eps, apply);
}

int mansy_clip_grad_rmsprop(float* flat_p, float* flat_g, float* flat_sq, long long n_flat, float max_grad_norm, float lr, float alpha, float eps,
                            double* scratch, void* stream) {
  MANSY_REQUIRE(flat_p && flat_g && flat_sq && scratch && n_flat >= 1, "clip_grad_rmsprop: bad arguments");
  Eng e; e.st = (hipStream_t)stream; e.W.acc = scratch;
  return e.clip_rmsprop(flat_p, flat_g, flat_sq, n_flat, max_grad_norm, lr, alpha, eps, 1);
}

}  // extern "C"
