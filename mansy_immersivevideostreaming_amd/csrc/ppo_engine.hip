// Bitrate-selection engine: policy/value/identifier networks, rollout action sampling, identifier training and
// reward relabel, GAE, and the PPO minibatch update -- each a single C-ABI call that enqueues its whole kernel
// sequence on one HIP stream (no host sync; hipGraph-capturable).
//
// Reference semantics: FeatureNet/Actor/Critic/QoEIdentifier (bitrate_selection/models/mansy.py:5-155),
// calculate_indentifier_reward + train_identifier (utils/mansy_utils.py:9-49), relabel loop (models/mansy_ppo.py:41-51),
// tianshou==0.4.8 PPOPolicy.process_fn/learn + A2CPolicy._compute_returns (T2, restated in oracle/ppo_oracle.py).
//
// MI355X-first structure:
//   * the ten branches of a FeatureNet are dense layers on disjoint column ranges of the 780-float observation row,
//     so the whole net is ONE block-diagonal product  F = LeakyReLU(obs[B,K] * Wbd[1280,K]^T + b)  on the fp32 MFMA GEMM
//     (K = 748 policy / 764 identifier); Wbd is re-packed from the compact reference-layout parameters after every
//     optimiser step (3.8 MB) and its gradient is un-packed from one dWbd = dPre^T obs product;
//   * actor and critic share F inside a minibatch (the reference evaluates the shared FeatureNet twice);
//   * 128->{15,1,3} output layers, residual add, softmax/sampling, PPO loss and its gradient are fused row-wise kernels.
#include <algorithm>
#include <cstdlib>
#include <string>
#include <vector>
#include "mansy_kernels.h"
#include "gemm_wsk.h"
#include "../../include/mansy_hip.h"

using mansy_gemm::GemmParams;

namespace {

// the streaming environment's device code (bit-exact double / sequential-float arithmetic: contraction off), for the fused
// policy + environment rollout launch
namespace envdev {
#pragma clang fp contract(off)
#include "env_device.h"
#pragma clang fp contract(fast)
}  // namespace envdev

constexpr int HID = 128, NB = 10, FEAT = HID * NB, OBS_LD = MANSY_OBS_LD, NACT = 15, MAXOUT = 16;
constexpr float SLOPE = 0.01f;
constexpr int K_POLICY = 748, K_IDENT = 764;
constexpr int KP = 768;                     // K of the packed block-diagonal image: padded to whole 32-wide K-tiles (LDS-DMA GEMM loop)
static_assert(KP <= MANSY_OBS_LD && KP % 32 == 0 && KP >= 764, "packed K must cover both nets and stay inside an observation row");
constexpr int RESID_COL = FEAT - HID;      // 10th branch output is the residual of every head
constexpr int NORM_PARTS_C = MANSY_CLIP_SCRATCH_DOUBLES;   // gradient-norm partial sums
constexpr int HB_BLOCKS = 64;               // workgroups of the output-layer backward (each ends with n_out x 128 global atomics)
constexpr int MAX_SLABS = 16;              // K splits of a head's fc product (head_split_request)
constexpr int DW_SLABS = 24;               // K (= batch) splits of the packed FeatureNet weight-gradient product (featnet_bwd)
constexpr int DW_TILES_MAX = 64;           // 64 x 64 tiles of that product that meet a branch's window (42 for both nets)

struct Branch { int off, len; };
__host__ __device__ inline Branch branch_geom(int j, int identifier) {
  const int off[NB] = {0, 8, 328, 648, 712, 720, 728, 736, 744, identifier ? MANSY_O_ACT_1HOT : MANSY_O_QOE_W};
  const int len[NB] = {8, 320, 320, 64, 8, 8, 8, 8, 1, identifier ? 15 : 3};
  Branch b; b.off = off[j]; b.len = len[j];
  return b;
}

// K window of branch j in whole 32-wide K-tiles: what the FeatureNet product reads for that branch's 128 features (tile_krange)
__host__ __device__ inline Branch window_geom(int j, int identifier) {
  const Branch g = branch_geom(j, identifier);
  Branch w; w.off = g.off / 32 * 32;
  const int hi = (g.off + g.len + 31) / 32 * 32;
  w.len = (hi < KP ? hi : KP) - w.off;
  return w;
}
// the 64 x 64 tiles of dWbd [FEAT, K] that meet the block diagonal, as (column tile, row tile) pairs in row-tile order
__host__ __device__ inline int active_tiles(int identifier, int K, int* list) {
  int n = 0;
  for (int t = 0; t < FEAT / 64; ++t) {
    const Branch w = window_geom(t * 64 / HID, identifier);
    const int c1 = (K + 63) / 64 < (w.off + w.len + 63) / 64 ? (K + 63) / 64 : (w.off + w.len + 63) / 64;
    for (int c = w.off / 64; c < c1; ++c) { if (list) { list[2 * n] = c; list[2 * n + 1] = t; } ++n; }
  }
  return n;
}
// input widths of the ten branches added up = columns of the compact (reference-layout) weight gradients
__host__ __device__ inline int compact_cols(int identifier) { int n = 0; for (int j = 0; j < NB; ++j) n += branch_geom(j, identifier).len; return n; }
inline int window_cols(int identifier) { int n = 0; for (int j = 0; j < NB; ++j) n += window_geom(j, identifier).len; return n; }

struct NetP {                  // one network = feature net + head
  const float* bw[NB]; const float* bb[NB]; const float* fc_w; const float* fc_b; const float* out_w; const float* out_b;
  float* gbw[NB]; float* gbb[NB]; float* gfc_w; float* gfc_b; float* gout_w; float* gout_b;
};

struct ParamInfo { std::string name; long long numel; int ndim; long long shape[4]; };
void addp(std::vector<ParamInfo>& v, const std::string& n, long long a, long long b = 0, long long c = 0) {
  ParamInfo p; p.name = n; p.shape[0] = a; p.shape[1] = b; p.shape[2] = c; p.shape[3] = 0; p.ndim = c ? 3 : (b ? 2 : 1);
  p.numel = a * (b ? b : 1) * (c ? c : 1); v.push_back(p);
}
void add_fnet(std::vector<ParamInfo>& v, const std::string& p, int identifier) {
  static const char* names[8] = {"conv1d1", "conv1d2", "conv1d3", "conv1d4", "conv1d5", "conv1d6", "conv1d7", "conv1d8"};
  static const int cin[8] = {1, 5, 5, 1, 1, 1, 1, 1};
  static const int kk[8] = {8, 64, 64, 64, 8, 8, 8, 8};
  for (int j = 0; j < 8; ++j) { addp(v, p + names[j] + ".0.weight", HID, cin[j], kk[j]); addp(v, p + names[j] + ".0.bias", HID); }
  addp(v, p + "fc1.0.weight", HID, 1); addp(v, p + "fc1.0.bias", HID);
  addp(v, p + "fc2.0.weight", HID, identifier ? 15 : 3); addp(v, p + "fc2.0.bias", HID);
}
// kind 0: actor-critic (unique tensors: shared feature net, actor head, critic head) ; kind 1: identifier
std::vector<ParamInfo> net_table(int kind) {
  std::vector<ParamInfo> v;
  if (kind == 0) {
    add_fnet(v, "actor.feature_net.", 0);
    addp(v, "actor.fc.0.weight", HID, FEAT); addp(v, "actor.fc.0.bias", HID); addp(v, "actor.out.weight", NACT, HID); addp(v, "actor.out.bias", NACT);
    addp(v, "critic.fc.0.weight", HID, FEAT); addp(v, "critic.fc.0.bias", HID); addp(v, "critic.out.weight", 1, HID); addp(v, "critic.out.bias", 1);
  } else {
    add_fnet(v, "identifier.feature_net.", 1);
    addp(v, "identifier.fc.0.weight", HID, FEAT); addp(v, "identifier.fc.0.bias", HID); addp(v, "identifier.out.weight", 3, HID);
    addp(v, "identifier.out.bias", 3);
  }
  return v;
}
// bind: params[0..19] feature net, then head(s)
void bind_net(const float* const* params, float* const* grads, int head_base, NetP& n) {
  for (int j = 0; j < NB; ++j) {
    n.bw[j] = params[2 * j]; n.bb[j] = params[2 * j + 1];
    n.gbw[j] = grads ? grads[2 * j] : nullptr; n.gbb[j] = grads ? grads[2 * j + 1] : nullptr;
  }
  n.fc_w = params[head_base]; n.fc_b = params[head_base + 1]; n.out_w = params[head_base + 2]; n.out_b = params[head_base + 3];
  n.gfc_w = grads ? grads[head_base] : nullptr; n.gfc_b = grads ? grads[head_base + 1] : nullptr;
  n.gout_w = grads ? grads[head_base + 2] : nullptr; n.gout_b = grads ? grads[head_base + 3] : nullptr;
}

// ------------------------------------------------------------------------------------ kernels
struct PackArgs {
  const float* bw[NB]; const float* bb[NB]; const float* fc_a; const float* fc_c; float* Wfc2;
  // riders on the same launch (independent prologue work of a PPO minibatch step): gather the minibatch's observation rows,
  // zero the flat gradient buffer
  const float* g_src; const int* g_idx; float* g_dst; int g_rows; float* zero_ptr; long long zero_n;
  // one more rider (an extra workgroup at the end of the grid): mean / unbiased std of the minibatch's advantages -> adv_stats[0..1]
  // (the fused loss in head_out_kernel normalises with them)
  const float* adv; int adv_n; float* adv_stats;
  // two small zero-fills riding on every pack launch: the packed bias-gradient accumulator dbbd [FEAT] (the dWbd product adds
  // its row sums there) and the gradient-norm / barrier slots (doubles, zeroed as pairs of floats)
  float* z2; int z2n; float* z3; int z3n;
};
// Wbd [FEAT, KP], bbd [FEAT], and per 64-feature column tile of the product the K range that holds its branch's weights
// (GemmEpilogue::tile_krange).  Only the K windows are written (branch weights, zeros around them up to the 32-wide K-tile
// borders): the product runs on the 64-column LDS-DMA / split-bf16 loops, which never read a row outside its window (featnet()
// insists on that path), so 131 k of the image's 983 k floats are touched per pack.  n_win = HID * window_cols().
// one 256-thread workgroup: mean / unbiased std of the minibatch's advantages (two-pass, like torch: mean, then variance)
__device__ __forceinline__ void adv_stats_block(const float* __restrict__ adv, const int* __restrict__ idx, int n, float* __restrict__ out) {
  __shared__ float sh_adv[8];
  float s1 = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s1 += adv[idx ? idx[i] : i];
  s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) sh_adv[threadIdx.x >> 6] = s1;
  __syncthreads();
  const float mean = (sh_adv[0] + sh_adv[1] + sh_adv[2] + sh_adv[3]) / (float)n;
  float q = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) { const float d = adv[idx ? idx[i] : i] - mean; q += d * d; }
  q = wave_sum(q);
  if ((threadIdx.x & 63) == 0) sh_adv[4 + (threadIdx.x >> 6)] = q;
  __syncthreads();
  if (threadIdx.x == 0) {
    out[0] = mean;
    out[1] = sqrtf((sh_adv[4] + sh_adv[5] + sh_adv[6] + sh_adv[7]) / (float)(n > 1 ? n - 1 : 1));
  }
}
__global__ __launch_bounds__(256) void pack_wbd_kernel(PackArgs a, int identifier, int K, long long n_win, float* __restrict__ Wbd,
                                                       float* __restrict__ bbd, int* __restrict__ krange, int* __restrict__ tlist) {
  if (a.adv && blockIdx.x == gridDim.x - 1) { adv_stats_block(a.adv, a.g_idx, a.adv_n, a.adv_stats); return; }
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx < a.z2n) a.z2[idx] = 0.f;
  if (idx < a.z3n) a.z3[idx] = 0.f;
  if (idx < FEAT / 64) {
    const Branch g = branch_geom((int)idx * 64 / HID, identifier);
    krange[2 * idx] = g.off / 32 * 32;
    krange[2 * idx + 1] = min(KP, (g.off + g.len + 31) / 32 * 32);
  }
  if (idx < FEAT) bbd[idx] = a.bb[idx / HID][idx % HID];
  if (idx == FEAT) active_tiles(identifier, K, tlist);          // tile list of the weight-gradient product (featnet_bwd)
  if (idx >= n_win) {                           // tail of the grid: [actor.fc | critic.fc] stacked to one [2*HID, FEAT] operand, then the riders
    long long i2 = idx - n_win;
    const long long n_fc = a.Wfc2 ? 2LL * HID * FEAT : 0;
    if (i2 < n_fc) { a.Wfc2[i2] = i2 < (long long)HID * FEAT ? a.fc_a[i2] : a.fc_c[i2 - (long long)HID * FEAT]; return; }
    i2 -= n_fc;
    const long long n_g = (long long)a.g_rows * (OBS_LD / 4);
    if (i2 < n_g) {
      const int r = (int)(i2 / (OBS_LD / 4)), c4 = (int)(i2 % (OBS_LD / 4));
      reinterpret_cast<float4*>(a.g_dst)[(size_t)r * (OBS_LD / 4) + c4] = reinterpret_cast<const float4*>(a.g_src)[(size_t)a.g_idx[r] * (OBS_LD / 4) + c4];
      return;
    }
    i2 -= n_g;
    if (i2 < (a.zero_n + 3) / 4) {
      const long long e0 = i2 * 4;
      if (e0 + 4 <= a.zero_n) *reinterpret_cast<float4*>(a.zero_ptr + e0) = make_float4(0.f, 0.f, 0.f, 0.f);
      else for (long long e = e0; e < a.zero_n; ++e) a.zero_ptr[e] = 0.f;
    }
    return;
  }
  int j = 0, rem = (int)idx;
  Branch w = window_geom(0, identifier);
  while (rem >= HID * w.len) { rem -= HID * w.len; w = window_geom(++j, identifier); }
  const int r = rem / w.len, col = w.off + rem % w.len;
  const Branch g = branch_geom(j, identifier);
  float v = 0.f;
  if (col < K && col >= g.off && col < g.off + g.len) v = a.bw[j][r * g.len + (col - g.off)];
  Wbd[(long long)(j * HID + r) * KP + col] = v;
}
// packed gradients -> the compact reference-layout parameter gradients: block-diagonal entries of dWbd [FEAT, K] and the
// packed bias gradient dbbd [FEAT] (first FEAT threads) are added to the ten branches' weight / bias gradients.
struct UnpackArgs { float* gbw[NB]; float* gbb[NB]; };
// Optional rider (PPO minibatch step with clipping): the squared gradient norm.  The branch gradients are exactly the values
// this kernel writes (the buffers were zeroed by the prologue), the head gradients [tail_g, tail_g + tail_n) were completed by
// earlier launches and are scanned by extra workgroups at the end of the grid; per-workgroup sums are added into the
// NORM_PARTS slots that clip_adam_kernel adds up (zeroed by the pack launch's riders).
struct NormRider { double* parts; const float* tail_g; long long tail_n; };
// Riders of the PPO minibatch step (round 3): the output layers' weight gradients gWout_h[k, c] = sum_r g_h[r, k] H_h[r, c] and
// gbout_h[k] = sum_r g_h[r, k] from what head_out_kernel wrote (rows summed in index order: deterministic, no atomics), stored
// into the (zeroed) flat gradient, their squares added to the norm; and the loss statistics from the per-row terms.  The norm
// scan of the head gradients skips the two output-layer ranges [skip0, skip0 + skip0_n), [skip1, ...) these riders own.
struct OutGradRider {
  int on; int rows;
  const float* g[2]; const float* H[2]; int n_out[2]; float* gW[2]; float* gb[2];
  const float* lossrows; float vf_coef, ent_coef; float* stats; int mse;
  const float* skip0; long long skip0_n; const float* skip1; long long skip1_n;
};
constexpr int OG_BLOCKS = 257;         // 16 logits (actor: 15 + the critic's one value) x 16 groups of 8 columns + 1 for biases / statistics
// dWbd arrives as nsplit K-split slabs (stride `slab` floats) of which only the block-diagonal windows were written
// (GemmEpilogue::tile_nrange): summed here in slab order.
__global__ __launch_bounds__(256) void unpack_dwbd_kernel(const float* __restrict__ dWbd, int nsplit, long long slab, const float* __restrict__ dbbd,
                                                          int identifier, int K, UnpackArgs a, NormRider nr, OutGradRider og, int og_first) {
  // one thread per element of the ten branch weights (HID x KC with KC = the branches' input widths added up: the index space of
  // the compact gradients, 8 x smaller than the packed image's)
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const int KC = compact_cols(identifier);                      // 745 state columns + 15 (identifier) / 3 (policy)
  const long long n_main = (long long)HID * KC;
  double sq = 0.0;
  if (og.on && (int)blockIdx.x >= og_first) {                   // output-layer gradient riders (the last OG_BLOCKS workgroups)
    const int ob = blockIdx.x - og_first;
    if (ob < OG_BLOCKS - 1) {
      // workgroup -> (logit k = ob / 16, 8 columns from 8 (ob % 16)); thread -> (column t % 8, row lane t / 8): 32 row lanes stride
      // the rows (16 each at a 512-row minibatch, 8 loads in flight), then one LDS reduction.  k = 15 is the critic's value column.
      __shared__ float sh_w[32][9];
      const int k = ob >> 4, cl = threadIdx.x & 7, c = 8 * (ob & 15) + cl, rl = threadIdx.x >> 3;
      if (!(k < og.n_out[0] || (k == MAXOUT - 1 && og.n_out[1] > 0))) return;       // (a head with fewer logits: the identifier's 3)
      const int hd = k < og.n_out[0] ? 0 : 1, kk = hd ? 0 : k;
      const float* __restrict__ g = og.g[hd]; const float* __restrict__ H = og.H[hd];
      float acc = 0.f, gacc = 0.f;
      for (int r0 = rl; r0 < og.rows; r0 += 32 * 8) {
        float gv[8], hv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int r = min(r0 + 32 * u, og.rows - 1); gv[u] = g[(size_t)r * MAXOUT + kk]; hv[u] = H[(size_t)r * HID + c]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool live = r0 + 32 * u < og.rows;
          acc = live ? fmaf(gv[u], hv[u], acc) : acc;
          gacc = live ? gacc + gv[u] : gacc;
        }
      }
      sh_w[rl][cl] = acc;
      if (cl == 0) sh_w[rl][8] = gacc;                           // the logit's bias gradient: the row lanes' sums of g[:, k]
      __syncthreads();
      if (threadIdx.x < 8) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += sh_w[i][threadIdx.x];
        og.gW[hd][kk * HID + 8 * (ob & 15) + threadIdx.x] = t;
        sq = (double)t * (double)t;
      } else if (threadIdx.x == 8 && (ob & 15) == 0) {           // one of the logit's 16 workgroups also stores its bias gradient
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += sh_w[i][8];
        og.gb[hd][kk] = t;
        sq = (double)t * (double)t;
      }
    } else {
      // the loss statistics
      __shared__ float sh_og[4][4];
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      if (og.stats && og.mse) {             // identifier: loss = sum of the rows' squared errors / (3 rows), summed in double like ident_mse_kernel
        __shared__ double sh_d[4];
        double t = 0.0;
        for (int i = threadIdx.x; i < og.rows; i += 256) t += (double)og.lossrows[4 * i];
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) t += __shfl_xor(t, o2, 64);
        if (lane == 0) sh_d[wv] = t;
        __syncthreads();
        if (threadIdx.x == 0) og.stats[0] = (float)(((sh_d[0] + sh_d[1]) + (sh_d[2] + sh_d[3])) / (double)(og.rows * 3));
      } else if (og.stats) {
        float c = 0.f, v = 0.f, e = 0.f;
        for (int i = threadIdx.x; i < og.rows; i += 256) { c += og.lossrows[4 * i]; v += og.lossrows[4 * i + 1]; e += og.lossrows[4 * i + 2]; }
        c = wave_sum(c); v = wave_sum(v); e = wave_sum(e);
        if (lane == 0) { sh_og[0][wv] = c; sh_og[1][wv] = v; sh_og[2][wv] = e; }
        __syncthreads();
        if (threadIdx.x == 0) {
          const float cm = (sh_og[0][0] + sh_og[0][1] + sh_og[0][2] + sh_og[0][3]) / (float)og.rows;
          const float vm = (sh_og[1][0] + sh_og[1][1] + sh_og[1][2] + sh_og[1][3]) / (float)og.rows;
          const float em = (sh_og[2][0] + sh_og[2][1] + sh_og[2][2] + sh_og[2][3]) / (float)og.rows;
          og.stats[0] = cm + og.vf_coef * vm - og.ent_coef * em; og.stats[1] = cm; og.stats[2] = vm; og.stats[3] = em;
        }
      }
    }
  } else {
  if (idx < FEAT) { const float b = dbbd[idx]; a.gbb[idx / HID][idx % HID] += b; sq += (double)b * (double)b; }
  if (idx < n_main) {
    const int r = (int)(idx / KC);
    int c = (int)(idx % KC), j = 0;
    Branch g = branch_geom(0, identifier);
    while (c >= g.len) { c -= g.len; g = branch_geom(++j, identifier); }
    const long long src = (long long)(j * HID + r) * K + g.off + c;
    float v = dWbd[src];
    for (int z = 1; z < nsplit; ++z) v += dWbd[(long long)z * slab + src];
    a.gbw[j][r * g.len + c] += v; sq += (double)v * (double)v;
  } else if (nr.parts) {
    const long long t4 = (idx - (n_main + 255) / 256 * 256) * 4;          // tail workgroups start on a workgroup boundary
    // (the output-layer ranges belong to the riders above when they run: whole 16-byte groups, every tensor is 256-byte aligned)
    const float* q = nr.tail_g + t4;
    const bool skip = og.on && ((q >= og.skip0 && q < og.skip0 + og.skip0_n) || (q >= og.skip1 && q < og.skip1 + og.skip1_n));
    if (skip) {
    } else if (t4 >= 0 && t4 + 4 <= nr.tail_n) {
      const float4 v = *reinterpret_cast<const float4*>(nr.tail_g + t4);
      sq += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    } else if (t4 >= 0) {
      for (long long e = t4; e < nr.tail_n; ++e) sq += (double)nr.tail_g[e] * (double)nr.tail_g[e];
    }
  }
  }
  if (!nr.parts) return;
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    if (t != 0.0) atomicAdd(nr.parts + (blockIdx.x % NORM_PARTS_C), t);
  }
}

// one wave per (row, head): H = A1 + F[:, resid] ; out[k] = H . Wout[k] + b[k] (k < n_out <= 16), optional sigmoid;
// optional categorical sample (inverse CDF on softmax(out)) with log-prob.  blockIdx.y selects the head (actor / critic share
// one launch in the PPO update).
// nsplit > 0: A1 is not yet formed -- sum the nsplit K-split slabs of the fc product (slab stride `slab` floats, row stride
// pre_ld, this head's columns from pre_col), add the fc bias, apply the LeakyReLU and store A1 (the backward reads it);
// nsplit == 0: A1 holds the activated fc output already.
struct HeadOut {
  float* A1; const float* fc_b; const float* Wout; const float* bout; int n_out; int sigmoid; float* H; float* out; int pre_col;
  int* act; float* logp;
  int out_ld;      // row stride of `out` for this head (0: the launch-wide out_ld); 1 writes a value head straight into a [B] vector
  // riders that used to be launches of their own (round 3):
  const int* act_given; int n_given;     // log-probability of GIVEN actions for the rows < n_given -> logp (process_fn's logp_old; was logp_kernel)
  float* relabel_rew; float* relabel_idrew; const float* relabel_obs; float relabel_lamb;      // identifier reward + relabel (was relabel_kernel)
};
// PPO minibatch loss fused into the output-layer launch (T2: tianshou 0.4.8 PPOPolicy.learn): head 0 (actor) turns its row's
// logits into the clipped-surrogate + entropy terms and their gradient wrt the logits, head 1 (critic) its value into the
// (clipped) value loss and its gradient; per-row loss terms go to lossrows [rows, 4] = (clip, vf, ent, -) and are summed in a
// fixed order by head_out_bwd_kernel.  adv_stats: minibatch mean / unbiased std of the advantages (pack_wbd_kernel rider).
struct LossFuse {
  int on;
  const int* act; const float* adv; const float* logp_old; const float* v_old; const float* ret; const int* idx;
  int n; float eps_clip, vf_coef, ent_coef; int norm_adv, value_clip; float adv_eps;
  float dual_clip;           // T2: PPOPolicy dual_clip (> 1) or 0: for negative advantages the surrogate is bounded below by dual_clip * adv
  const float* adv_stats; float* dlogits; float* dvalue; int dvalue_ld; float* lossrows;
  // round 3: the output layer's input-side backward in the same launch (was head_out_bwd_kernel's row loop): per head h,
  //   dH_h[row, c] = sum_k g[k] Wout_h[k, c]      (residual branch of the head; joins dF in the dF product's epilogue)
  //   dA1[row, h * HID + c] = dH * leaky'(A1)     (operand of the fc weight-gradient and dF products)
  // The output layer's own weight gradients (a [<= 15, B] x [B, 128] reduction over the rows) are taken by riders of the
  // unpack launch from the dlogits / dvalue and H this kernel writes -- fixed summation order, no atomics.  Null: off.
  float* bwd_dH[2]; float* bwd_dA1; int bwd_dA1_ld;
  const float* mse_obs;      // on == 2 (identifier): observation rows holding the regression target at MANSY_O_QOE_W
};
// Rollout fusion: the wave that sampled row e's action goes on to step environment e (lane = tile) in the same launch --
// MANSYEnv.step with the action it just drew; the observation rows it writes are the next policy input.
struct EnvFuse {
  int on; mansy_env_tables T; envdev::EnvState* st; float* obs_next; float* obs_cur; float* reward; unsigned char* done; float* qoe_parts;
  mansy_env_episode_log elog;
};
struct HeadOutArgs { HeadOut h[2]; };
// MODE 0: every path by its run-time flag.  The two launches that make up most of a PPO cycle have their flags fixed at compile time, so that the
// paths they never take -- and the scalar loads of their arguments -- are not in their instruction stream: MODE 1 = the rollout launch (one head, fc
// slabs, sampling + environment step; no loss, no riders, no sigmoid), MODE 2 = the minibatch step (two heads, fc slabs, fused PPO loss + input-side
// backward; no sampling, no environment, no riders), MODE 3 = forward-only launches (process_fn's evaluation pass over 2 x 4096 rows, the identifier's
// validation / relabel passes: no loss, no sampling, no environment; the log-probability / relabel riders and the sigmoid stay run-time flags).
template <int MODE>
__global__ __launch_bounds__(256) void head_out_kernel(HeadOutArgs args, const float* __restrict__ A1pre, int nsplit_rt, long long slab, int pre_ld,
                                                       const float* __restrict__ F, int out_ld, int rows, const float* __restrict__ u_ext,
                                                       uint32_t seed, uint32_t site, LossFuse lf, EnvFuse ef) {
  const HeadOut& d = args.h[blockIdx.y];
  float* __restrict__ A1 = d.A1; const float* __restrict__ Wout = d.Wout; const float* __restrict__ bout = d.bout;
  float* __restrict__ H = d.H; float* __restrict__ out = d.out; int* __restrict__ act = (MODE == 2 || MODE == 3) ? nullptr : d.act; float* __restrict__ logp = d.logp;
  const int n_out = d.n_out, sigmoid = (MODE == 0 || MODE == 3) ? d.sigmoid : 0;
  const int lf_on = (MODE == 1 || MODE == 3) ? 0 : (MODE == 2 ? 1 : lf.on);
  const bool env_on = MODE == 1 ? true : ((MODE == 2 || MODE == 3) ? false : ef.on != 0);
  const int nsplit = (MODE == 0 || MODE == 3) ? nsplit_rt : max(nsplit_rt, 1);          // (modes 1 / 2 always sum slabs)
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if (row >= rows) return;
  // fused-loss operands (a gather through idx): requested first, ahead of this kernel's stores, so the two dependent round
  // trips overlap the slab sums
  int l_bi = 0, l_act = 0; float l_adv = 0.f, l_logp_old = 0.f, l_ret = 0.f, l_vold = 0.f, l_mean = 0.f, l_std = 1.f;
  if (lf_on == 1) {
    l_bi = lf.idx ? lf.idx[row] : row;
    if (blockIdx.y == 0) { l_adv = lf.adv[l_bi]; l_act = lf.act[l_bi]; l_logp_old = lf.logp_old[l_bi]; l_mean = lf.adv_stats[0]; l_std = lf.adv_stats[1]; }
    else { l_ret = lf.ret[l_bi]; l_vold = lf.value_clip ? lf.v_old[l_bi] : 0.f; }
  }
  // Every load that does not depend on this kernel's own arithmetic is requested here, before the first wait: the residual branch of F, the
  // output layer's weights (at clamped indices: a predicate around each would serialise them), the sampling uniform, the fc slabs -- and, in
  // the rollout launch, the environment's state record, whose table rows (env_step_requests below) then fly under the output layer.
  const float f0 = F[(size_t)row * FEAT + RESID_COL + lane];
  const float f1 = F[(size_t)row * FEAT + RESID_COL + 64 + lane];
  float w0[MAXOUT], w1[MAXOUT], bo[MAXOUT];
#pragma unroll
  for (int k = 0; k < MAXOUT; ++k) {
    const int kc = min(k, n_out - 1);
    w0[k] = Wout[kc * HID + lane]; w1[k] = Wout[kc * HID + 64 + lane]; bo[k] = bout[kc];
  }
  const float u_row = (act && u_ext) ? u_ext[row] : 0.f;
  float a0, a1;
  float p0[MAX_SLABS], p1[MAX_SLABS];
  if (nsplit > 0) {
    a0 = d.fc_b[lane]; a1 = d.fc_b[64 + lane];
    const float* pre = A1pre + (size_t)row * pre_ld + d.pre_col;
    // all slabs requested up front at clamped indices (a run-time trip count made hipcc wait for every pair of loads: up to 16
    // dependent L2 round trips); summed in slab order
#pragma unroll
    for (int z = 0; z < MAX_SLABS; ++z) {
      const long long zz = (long long)min(z, nsplit - 1) * slab;
      p0[z] = pre[zz + lane]; p1[z] = pre[zz + 64 + lane];
    }
  } else {
    a0 = A1[(size_t)row * HID + lane]; a1 = A1[(size_t)row * HID + 64 + lane];
  }
  const int erow = __builtin_amdgcn_readfirstlane(row);
  envdev::EnvRegs es = {};
  if (env_on && act) envdev::load_state(es, ef.st[erow], lane);
  if (nsplit > 0) {
#pragma unroll
    for (int z = 0; z < MAX_SLABS; ++z) { a0 = z < nsplit ? a0 + p0[z] : a0; a1 = z < nsplit ? a1 + p1[z] : a1; }
    a0 = a0 > 0.f ? a0 : a0 * SLOPE; a1 = a1 > 0.f ? a1 : a1 * SLOPE;
    A1[(size_t)row * HID + lane] = a0; A1[(size_t)row * HID + 64 + lane] = a1;
  }
  envdev::EnvPre eq = {};
  if (env_on && act) eq = envdev::env_step_requests(ef.T, es, lane);
  const float h0 = a0 + f0;
  const float h1 = a1 + f1;
  if (H) { H[(size_t)row * HID + lane] = h0; H[(size_t)row * HID + 64 + lane] = h1; }
  float o[MAXOUT];
#pragma unroll
  for (int k = 0; k < MAXOUT; ++k) {
    float p = wave_sum(h0 * w0[k] + h1 * w1[k]) + bo[k];
    if (sigmoid) p = 1.f / (1.f + expf(-p));
    o[k] = k < n_out ? p : 0.f;
  }
  if (out && lane < n_out) {
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < MAXOUT; ++k) if (k == lane) mine = o[k];
    out[(size_t)row * (d.out_ld ? d.out_ld : out_ld) + lane] = mine;
  }
  if ((MODE == 0 || MODE == 3) && d.act_given && row < d.n_given) {      // log pi(a | obs) of the given action (logp_kernel's arithmetic: max, sequential sum of exp, log)
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NACT; ++k) m = fmaxf(m, o[k]);
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) se += expf(o[k] - m);
    const int a = d.act_given[row];
    float oa = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) if (k == a) oa = o[k];
    if (lane == 0) logp[row] = (oa - m) - logf(se);
  }
  if ((MODE == 0 || MODE == 3) && d.relabel_rew && lane == 0) {          // rew <- (1 - lamb) rew + lamb (1 - mean_k (pred_k - w_k)^2)   (mansy_ppo.py:43-48)
    float sq = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) { const float dd = o[k] - d.relabel_obs[(size_t)row * OBS_LD + MANSY_O_QOE_W + k]; sq += dd * dd; }
    const float ir = 1.f - sq / 3.f;
    if (d.relabel_idrew) d.relabel_idrew[row] = ir;
    d.relabel_rew[row] = (1.f - d.relabel_lamb) * d.relabel_rew[row] + d.relabel_lamb * ir;
  }
  float gk[MAXOUT];                // dL/d(output pre-activation) of this row and head (fused loss)
#pragma unroll
  for (int k = 0; k < MAXOUT; ++k) gk[k] = 0.f;
  if (lf_on == 1 && blockIdx.y == 0) {  // actor: clipped surrogate + entropy of this row, gradient wrt the logits (every lane computes the same scalars)
    float adv = l_adv;
    if (lf.norm_adv) adv = (adv - l_mean) / (l_std + lf.adv_eps);
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NACT; ++k) m = fmaxf(m, o[k]);
    float e[NACT], se = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) { e[k] = expf(o[k] - m); se += e[k]; }
    const float lse = logf(se);
    const int a = l_act;
    float lg_act = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) if (k == a) lg_act = o[k];
    const float lp_a = (lg_act - m) - lse;
    const float ratio = expf(lp_a - l_logp_old);
    const float surr1 = ratio * adv;
    const float rc = fminf(fmaxf(ratio, 1.f - lf.eps_clip), 1.f + lf.eps_clip);
    const float surr2 = rc * adv;
    float dlogp;
    if (surr1 <= surr2) dlogp = -ratio * adv;
    else dlogp = (ratio > 1.f - lf.eps_clip && ratio < 1.f + lf.eps_clip) ? -ratio * adv : 0.f;
    float clip_term = fminf(surr1, surr2);
    if (lf.dual_clip > 0.f && adv < 0.f) {                  // T2 (ppo.py): clip2 = max(min(surr1, surr2), dual_clip * adv) where adv < 0
      const float bound = lf.dual_clip * adv;
      if (clip_term < bound) { clip_term = bound; dlogp = 0.f; }      // the bound does not depend on the logits
    }
    dlogp /= (float)lf.n;
    float ent = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) { const float p = e[k] / se; const float lp = (o[k] - m) - lse; ent -= p * lp; }
    float mine = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) {
      const float p = e[k] / se, lp = (o[k] - m) - lse;
      float g = dlogp * ((k == a ? 1.f : 0.f) - p);
      g += -lf.ent_coef * (-p * (lp + ent)) / (float)lf.n;      // d(-ent_coef * mean H)/dlogit_k = ent_coef * p_k (log p_k + H) / n
      if (k == lane) mine = g;
      gk[k] = g;
    }
    if (lane < MAXOUT) lf.dlogits[(size_t)row * MAXOUT + lane] = mine;       // column 15 (lane 15) is 0
    if (lane == 0) { lf.lossrows[4 * (size_t)row + 0] = -clip_term; lf.lossrows[4 * (size_t)row + 2] = ent; }
  }
  if (lf_on == 1 && blockIdx.y == 1) {   // critic: (clipped) value loss of this row and its gradient (every lane computes the same scalars)
    const float v = o[0], ret = l_ret;
    float dv, lv;
    if (lf.value_clip) {
      const float vo = l_vold;
      const float diff = v - vo;
      const float vc = vo + fminf(fmaxf(diff, -lf.eps_clip), lf.eps_clip);
      const float vf1 = (ret - v) * (ret - v), vf2 = (ret - vc) * (ret - vc);
      if (vf1 >= vf2) { lv = vf1; dv = -2.f * (ret - v); }
      else { lv = vf2; dv = (diff > -lf.eps_clip && diff < lf.eps_clip) ? -2.f * (ret - vc) : 0.f; }
    } else { lv = (ret - v) * (ret - v); dv = -2.f * (ret - v); }
    gk[0] = lf.vf_coef * dv / (float)lf.n;
    if (lane == 0) { lf.dvalue[(size_t)row * lf.dvalue_ld] = gk[0]; lf.lossrows[4 * (size_t)row + 1] = lv; }
  }
  if (lf_on == 2) {                 // identifier: MSE(sigmoid outputs, the observation's normalised QoE weights) of this row, gradient wrt the pre-sigmoid
    float sq = 0.f, mine = 0.f;      // (train_identifier, mansy_utils.py:20-31; ident_mse_kernel's arithmetic per element)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float p = o[k];
      const float dd = p - lf.mse_obs[(size_t)row * OBS_LD + MANSY_O_QOE_W + k];
      sq += dd * dd;
      gk[k] = (2.f * dd / (float)(lf.n * 3)) * p * (1.f - p);
      if (k == lane) mine = gk[k];
    }
    if (lane < MAXOUT) lf.dlogits[(size_t)row * MAXOUT + lane] = mine;
    if (lane == 0) lf.lossrows[4 * (size_t)row] = sq;
  }
  if (lf_on && lf.bwd_dA1) {        // the output layer's input-side backward for this row (see LossFuse)
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int k = 0; k < MAXOUT; ++k) {
      const float g = k < n_out ? gk[k] : 0.f;
      d0 = fmaf(g, w0[k], d0); d1 = fmaf(g, w1[k], d1);
    }
    float* dH = lf.bwd_dH[blockIdx.y];
    dH[(size_t)row * HID + lane] = d0; dH[(size_t)row * HID + 64 + lane] = d1;
    float* dA = lf.bwd_dA1 + (size_t)row * lf.bwd_dA1_ld + blockIdx.y * HID;
    dA[lane] = a0 > 0.f ? d0 : d0 * SLOPE; dA[64 + lane] = a1 > 0.f ? d1 : d1 * SLOPE;
  }
  if (act) {                       // Categorical(logits).sample() by inverse CDF; log_prob of the sample
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXOUT; ++k) if (k < n_out) m = fmaxf(m, o[k]);
    float e[MAXOUT], s = 0.f;
#pragma unroll
    for (int k = 0; k < MAXOUT; ++k) { e[k] = k < n_out ? expf(o[k] - m) : 0.f; s += e[k]; }
    const float u = u_ext ? u_row : mansy_uniform01(seed, site, (uint32_t)row);
    float c = 0.f; int a = n_out - 1; bool found = false;
#pragma unroll
    for (int k = 0; k < MAXOUT; ++k) {
      if (k < n_out) { c += e[k] / s; if (!found && c > u) { a = k; found = true; } }
    }
    if (lane == 0) {
      act[row] = a;
      float oa = 0.f;
#pragma unroll
      for (int k = 0; k < MAXOUT; ++k) if (k == a) oa = o[k];
      if (logp) logp[row] = (oa - m) - logf(s);
    }
    if (env_on) envdev::env_step_finish(ef.T, ef.st, erow, lane, a, es, eq, ef.obs_next, ef.obs_cur, ef.reward, ef.done, ef.qoe_parts, ef.elog);
  }
}


// ------------------------------------------------------------------------------------ persistent rollout on XCD teams (round 5)
// A collect of T vector steps as ONE launch.  The three launches of a rollout step (FeatureNet product, heads' fc product in K-split slabs, output layer +
// sampling + environment step) become three PHASES of a persistent kernel; what separates them is not a kernel boundary (1.5 us + each launch's fill and
// drain) but a barrier among the workgroups of ONE XCD through that XCD's L2: 0.77 us (tools/team_lab.hip, profiles/r05_team_seam_lab.txt).  One workgroup
// per CU reads the XCD it runs on (HW_REG_XCC_ID -- placement is READ, never assumed) and joins that XCD's team; the environments are dealt to the teams in
// chunks of 32 rows, and a chunk never leaves its team: trajectories are independent, so no byte ever crosses an XCD inside the launch.  Per step and chunk:
//   A  the members share the 40 column blocks of F = LeakyReLU(obs Wbd^T + b)   (gemm_f32_wsk_body, the SAME body, blocks and K order as the launch it replaces)
//   B  ... and the 4 x nsplit blocks of the fc product's K-split slabs
//   C  row r's owner sums the slabs, forms the logits, samples and steps environment r (the row logic of head_out_kernel<1>); the observation it writes
//      is phase A's operand of the next step.
// Operands written by another workgroup of the launch are read past this CU's L1 (sc1 LDS-DMA / sc1 loads; the L2 is the team's coherence point: plain
// stores, drained before the barrier); weights, tables and uniforms do not change inside the launch.  Results are bit-identical to the three-launch path.
struct RolloutCtl { unsigned claim[8]; unsigned started; unsigned err; unsigned pad[6]; unsigned bar[8][16]; unsigned long long reserved[8]; };
static_assert(sizeof(RolloutCtl) <= MANSY_ROLLOUT_CTL_BYTES, "rollout control block");
struct RolloutArgs {
  GemmParams p1, p2; int g1x, g2x, g2z;
  const float* fc_b; const float* Wout; const float* bout;
  const float* F; const float* A1s; long long slab; int nsplit;
  int n_env, T;
  float* obs; long long obs_ts; float* obs_next; long long obsn_ts; float* carry;
  const float* u; int* act; float* logp; float* rew; unsigned char* done;
  mansy_env_tables Tb; envdev::EnvState* st; float* qoe_parts; mansy_env_episode_log elog;
  long long timeout_ticks; int* err_host;
};
__device__ __forceinline__ float ld_sc1_f(const float* p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ bool spin_until(const unsigned* p, unsigned target, long long ticks) {
  const long long t0 = wall_clock64();
  while ((int)(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
    if (wall_clock64() - t0 > ticks) return false;
    __builtin_amdgcn_s_sleep(1);
  }
  return true;
}
__global__ __launch_bounds__(256) void rollout_team_kernel(RolloutArgs a, RolloutCtl* __restrict__ ctl) {
  __shared__ __attribute__((aligned(1024))) float smem[mansy_gemm::WSK_SMEM_FLOATS];
  __shared__ unsigned sh[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    xcc &= 7u;
    sh[0] = xcc;
    sh[1] = __hip_atomic_fetch_add(&ctl->claim[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(&ctl->started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    sh[2] = spin_until(&ctl->started, gridDim.x, a.timeout_ticks) ? 1u : 0u;      // every workgroup of the launch is resident and has joined its team
    for (int i = 0; i < 8; ++i) sh[4 + i] = __hip_atomic_load(&ctl->claim[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int xcc = sh[0], m = sh[1], s = sh[4 + xcc];
  int nteams = 0, ti = 0;
  for (int i = 0; i < 8; ++i) { if (sh[4 + i]) { if (i < xcc) ++ti; ++nteams; } }
  if (!sh[2]) { if (tid == 0) { ctl->err = 1u; __hip_atomic_store(a.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } return; }
  unsigned nbar = 0;
  unsigned* const bar = &ctl->bar[xcc][0];
  // a barrier in two halves, so that loads which do not depend on the phase just finished can be REQUESTED between the arrival and the wait and fly
  // under it (the wait's own poll returns behind them: loads return in order)
  auto barrier_arrive = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's stores have left for L2
    __syncthreads();
    nbar += (unsigned)s;
    if (tid == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto barrier_wait = [&]() -> bool {
    if (tid == 0) sh[3] = spin_until(bar, nbar, a.timeout_ticks) ? 1u : 0u;
    __syncthreads();
    return sh[3] != 0u;
  };
  const int chunks = (a.n_env + 31) / 32;
  bool ok = true;
  for (int chunk = ti; chunk < chunks && ok; chunk += nteams) {
    const int row0 = chunk * 32, rows = min(32, a.n_env - row0);
    // phase C: member m's wave w owns the rows m + s (w + 4 j) of the chunk (with a full team of 32: row m, wave 0)
    for (int t = 0; t < a.T && ok; ++t) {
      // ---- A: FeatureNet blocks of this chunk's rows
      {
        GemmParams p = a.p1;
        p.A = a.obs + (long long)t * a.obs_ts + (long long)row0 * p.lda; p.C = a.p1.C + (long long)row0 * p.ldc; p.M = rows;
        bool first = true;
        for (int blk = m; blk < a.g1x; blk += s) {
          if (!first) __syncthreads();
          first = false;
          mansy_gemm::gemm_f32_wsk_body<false, false, true, true>(p, blk, a.g1x, 1, 1, smem);
        }
      }
      barrier_arrive();
      ok = barrier_wait();
      if (!ok) break;
      // ---- B: the fc product's slabs
      {
        GemmParams p = a.p2;
        p.A = a.p2.A + (long long)row0 * p.lda; p.C = a.p2.C + (long long)row0 * p.ldc; p.M = rows;
        bool first = true;
        for (int blk = m; blk < a.g2x * a.g2z; blk += s) {
          if (!first) __syncthreads();
          first = false;
          mansy_gemm::gemm_f32_wsk_body<false, false, true, true>(p, blk, a.g2x, 1, a.g2z, smem);
        }
      }
      barrier_arrive();
      // ---- C: one wave per row: slab sums, output layer, sampling, environment step.  Everything that does not depend on phase B -- the output
      // layer's weights, the uniform, the environment's record and the table rows it names -- is requested HERE, between arrival and wait.
      const int rr0 = m + s * wave;                          // this wave's first row of the chunk
      const bool have0 = rr0 < rows;
      const int row_c = row0 + (have0 ? rr0 : 0);
      float w0[MAXOUT], w1[MAXOUT], bo[MAXOUT];
#pragma unroll
      for (int k = 0; k < MAXOUT; ++k) { const int kc = min(k, NACT - 1); w0[k] = a.Wout[kc * HID + lane]; w1[k] = a.Wout[kc * HID + 64 + lane]; bo[k] = a.bout[kc]; }
      const float fcb0 = a.fc_b[lane], fcb1 = a.fc_b[64 + lane];
      float u_pre = 0.f;
      envdev::EnvRegs es_pre = {};
      envdev::EnvPre eq_pre = {};
      if (have0) {
        u_pre = a.u[(size_t)t * a.n_env + row_c];
        envdev::load_state(es_pre, a.st[__builtin_amdgcn_readfirstlane(row_c)], lane);
        eq_pre = envdev::env_step_requests(a.Tb, es_pre, lane);
      }
      ok = barrier_wait();
      if (!ok) break;
      for (int j = wave; ; j += 4) {
        const int rr = m + s * j;
        if (rr >= rows) break;
        const int row = row0 + rr;
        const float f0 = ld_sc1_f(a.F + (size_t)row * FEAT + RESID_COL + lane);
        const float f1 = ld_sc1_f(a.F + (size_t)row * FEAT + RESID_COL + 64 + lane);
        float a0 = fcb0, a1 = fcb1;
        float p0[MAX_SLABS], p1[MAX_SLABS];
        const float* pre = a.A1s + (size_t)row * HID;
#pragma unroll
        for (int z = 0; z < MAX_SLABS; ++z) { const long long zz = (long long)min(z, a.nsplit - 1) * a.slab; p0[z] = ld_sc1_f(pre + zz + lane); p1[z] = ld_sc1_f(pre + zz + 64 + lane); }
        const int erow = __builtin_amdgcn_readfirstlane(row);
        envdev::EnvRegs es = es_pre;
        envdev::EnvPre eq = eq_pre;
        float u_row = u_pre;
        if (j != wave) {                                     // (a team smaller than the chunk: further rows of this wave, nothing requested ahead)
          u_row = a.u[(size_t)t * a.n_env + row];
          envdev::load_state(es, a.st[erow], lane);
          eq = envdev::env_step_requests(a.Tb, es, lane);
        }
#pragma unroll
        for (int z = 0; z < MAX_SLABS; ++z) { a0 = z < a.nsplit ? a0 + p0[z] : a0; a1 = z < a.nsplit ? a1 + p1[z] : a1; }
        a0 = a0 > 0.f ? a0 : a0 * SLOPE; a1 = a1 > 0.f ? a1 : a1 * SLOPE;
        const float h0 = a0 + f0, h1 = a1 + f1;
        float o[MAXOUT];
#pragma unroll
        for (int k = 0; k < MAXOUT; ++k) { const float pk = wave_sum(h0 * w0[k] + h1 * w1[k]) + bo[k]; o[k] = k < NACT ? pk : 0.f; }
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXOUT; ++k) if (k < NACT) mx = fmaxf(mx, o[k]);
        float e[MAXOUT], se = 0.f;
#pragma unroll
        for (int k = 0; k < MAXOUT; ++k) { e[k] = k < NACT ? expf(o[k] - mx) : 0.f; se += e[k]; }
        float c = 0.f; int act = NACT - 1; bool found = false;
#pragma unroll
        for (int k = 0; k < MAXOUT; ++k) { if (k < NACT) { c += e[k] / se; if (!found && c > u_row) { act = k; found = true; } } }
        if (lane == 0) {
          a.act[(size_t)t * a.n_env + row] = act;
          float oa = 0.f;
#pragma unroll
          for (int k = 0; k < MAXOUT; ++k) if (k == act) oa = o[k];
          a.logp[(size_t)t * a.n_env + row] = (oa - mx) - logf(se);
        }
        float* obs_cur = t + 1 < a.T ? a.obs + (long long)(t + 1) * a.obs_ts : a.carry;
        envdev::env_step_finish(a.Tb, a.st, erow, lane, act, es, eq, a.obs_next + (long long)t * a.obsn_ts, obs_cur, a.rew + (size_t)t * a.n_env,
                                a.done + (size_t)t * a.n_env, a.qoe_parts, a.elog);
      }
      barrier_arrive();
      ok = barrier_wait();
    }
  }
  if (!ok && tid == 0) { ctl->err = 1u; __hip_atomic_store(a.err_host, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
}

// backward of the output layer + residual: dH[r,c] = sum_k g[r,k] Wout[k,c] ; dA1 = dH * leaky'(A1) ;
// gWout[k,c] += sum_r g[r,k] H[r,c] ; gbout[k] += sum_r g[r,k]   (g already includes sigmoid' when needed)
struct HeadBwd {
  const float* g; int g_ld; const float* A1; const float* H; const float* Wout; int n_out; float* dH; float* dA1; int dA1_ld; float* gWout;
  float* gbout;
};
struct HeadBwdArgs { HeadBwd h[2]; };
// loss statistics of the fused PPO loss: stats = [loss, clip, vf, ent] from the per-row terms (fixed summation order)
struct LossFinish { const float* lossrows; int n; float vf_coef, ent_coef; float* stats; };
__global__ __launch_bounds__(256) void head_out_bwd_kernel(HeadBwdArgs args, int rows, LossFinish fin) {
  if (fin.stats && blockIdx.x == 0 && blockIdx.y == 0) {
    __shared__ float sh_fin[3][4];
    float c = 0.f, v = 0.f, e = 0.f;
    for (int i = threadIdx.x; i < fin.n; i += 256) { c += fin.lossrows[4 * i]; v += fin.lossrows[4 * i + 1]; e += fin.lossrows[4 * i + 2]; }
    c = wave_sum(c); v = wave_sum(v); e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) { sh_fin[0][threadIdx.x >> 6] = c; sh_fin[1][threadIdx.x >> 6] = v; sh_fin[2][threadIdx.x >> 6] = e; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float cm = (sh_fin[0][0] + sh_fin[0][1] + sh_fin[0][2] + sh_fin[0][3]) / (float)fin.n;
      const float vm = (sh_fin[1][0] + sh_fin[1][1] + sh_fin[1][2] + sh_fin[1][3]) / (float)fin.n;
      const float em = (sh_fin[2][0] + sh_fin[2][1] + sh_fin[2][2] + sh_fin[2][3]) / (float)fin.n;
      fin.stats[0] = cm + fin.vf_coef * vm - fin.ent_coef * em; fin.stats[1] = cm; fin.stats[2] = vm; fin.stats[3] = em;
    }
    __syncthreads();
  }
  const HeadBwd& d = args.h[blockIdx.y];
  const float* __restrict__ g = d.g; const int g_ld = d.g_ld; const float* __restrict__ A1 = d.A1; const float* __restrict__ H = d.H;
  const float* __restrict__ Wout = d.Wout; const int n_out = d.n_out; float* __restrict__ dH = d.dH; float* __restrict__ dA1 = d.dA1;
  const int dA1_ld = d.dA1_ld; float* __restrict__ gWout = d.gWout; float* __restrict__ gbout = d.gbout;
  __shared__ float sw[MAXOUT * HID];
  __shared__ float sb[MAXOUT];
  for (int i = threadIdx.x; i < MAXOUT * HID; i += 256) sw[i] = 0.f;
  if (threadIdx.x < MAXOUT) sb[threadIdx.x] = 0.f;
  __syncthreads();
  const int c = threadIdx.x & 127;                 // two rows per pass
  const int half = threadIdx.x >> 7;
  float aw[MAXOUT], ab[MAXOUT];
#pragma unroll
  for (int k = 0; k < MAXOUT; ++k) { aw[k] = 0.f; ab[k] = 0.f; }
  float wk[MAXOUT];                                  // this thread's column of the output weights (row-invariant)
#pragma unroll
  for (int k = 0; k < MAXOUT; ++k) wk[k] = Wout[min(k, n_out - 1) * HID + c];
  for (int row = blockIdx.x * 2 + half; row < rows; row += gridDim.x * 2) {
    float acc = 0.f;
    const float hv = H[(size_t)row * HID + c];
    float gk[MAXOUT];
#pragma unroll
    for (int k = 0; k < MAXOUT; ++k) gk[k] = k < g_ld ? g[(size_t)row * g_ld + k] : 0.f;      // g rows are g_ld (>= n_out) wide
#pragma unroll
    for (int k = 0; k < MAXOUT; ++k) {
      const float gm = k < n_out ? gk[k] : 0.f;
      acc = fmaf(gm, wk[k], acc);
      aw[k] = fmaf(gm, hv, aw[k]);
      if (c == 0) ab[k] += gm;
    }
    dH[(size_t)row * HID + c] = acc;
    const float a1 = A1[(size_t)row * HID + c];
    dA1[(size_t)row * dA1_ld + c] = a1 > 0.f ? acc : acc * SLOPE;
  }
#pragma unroll
  for (int k = 0; k < MAXOUT; ++k) {
    if (k < n_out) { atomicAdd(&sw[k * HID + c], aw[k]); if (c == 0) atomicAdd(&sb[k], ab[k]); }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_out * HID; i += 256) atomicAdd(gWout + i, sw[i]);
  if (threadIdx.x < n_out) atomicAdd(gbout + threadIdx.x, sb[threadIdx.x]);
}

__device__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// Behaviour cloning: mean cross entropy of the expert actions minus ent_coef * mean entropy, and its gradient wrt the
// logits (the value head receives a zero gradient).
__global__ __launch_bounds__(1024) void bc_loss_kernel(const float* __restrict__ logits, const int* __restrict__ act, int n, float ent_coef,
                                                       float* __restrict__ dlogits, float* __restrict__ dvalue, float* __restrict__ stats) {
  __shared__ float sh[16];
  float l_ce = 0.f, l_ent = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float lg[MAXOUT];
#pragma unroll
    for (int q4 = 0; q4 < MAXOUT / 4; ++q4)
      *reinterpret_cast<float4*>(lg + 4 * q4) = *reinterpret_cast<const float4*>(logits + (size_t)i * MAXOUT + 4 * q4);
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < NACT; ++k) m = fmaxf(m, lg[k]);
    float e[NACT], se = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) { e[k] = expf(lg[k] - m); se += e[k]; }
    const float lse = logf(se);
    const int a = act[i];
    float ent = 0.f, lp_act = 0.f;
#pragma unroll
    for (int k = 0; k < NACT; ++k) {
      const float p = e[k] / se, lp = (lg[k] - m) - lse;
      ent -= p * lp;
      if (k == a) lp_act = lp;
    }
    l_ce += -lp_act;
    l_ent += ent;
    float gout[MAXOUT];
#pragma unroll
    for (int k = 0; k < NACT; ++k) {
      const float p = e[k] / se, lp = (lg[k] - m) - lse;
      gout[k] = (p - (k == a ? 1.f : 0.f)) / (float)n + ent_coef * p * (lp + ent) / (float)n;
    }
    gout[NACT] = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < MAXOUT / 4; ++q4)
      *reinterpret_cast<float4*>(dlogits + (size_t)i * MAXOUT + 4 * q4) = *reinterpret_cast<const float4*>(gout + 4 * q4);
    dvalue[(size_t)i * MAXOUT] = 0.f;
  }
  const float ce = block_sum(l_ce, sh) / (float)n;
  const float em = block_sum(l_ent, sh) / (float)n;
  if (threadIdx.x == 0) { stats[0] = ce - ent_coef * em; stats[1] = ce; stats[2] = em; }
}

// MSE(pred[B,3], obs[:,745:748]) forward + gradient wrt the PRE-sigmoid output (pred = sigmoid(z))
__global__ __launch_bounds__(256) void ident_mse_kernel(const float* __restrict__ pred, const float* __restrict__ obs, int B, float* __restrict__ dz,
                                                        double* __restrict__ acc, float* __restrict__ loss) {
  double local = 0.0;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < B * 3; i += gridDim.x * 256) {
    const int r = i / 3, k = i % 3;
    const float p = pred[(size_t)r * MAXOUT + k];
    const float d = p - obs[(size_t)r * OBS_LD + MANSY_O_QOE_W + k];
    local += (double)(d * d);
    if (dz) dz[(size_t)r * MAXOUT + k] = (2.f * d / (float)(B * 3)) * p * (1.f - p);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o, 64);
  if ((threadIdx.x & 63) == 0) atomicAdd(acc, local);
  // the last workgroup to arrive turns the sum into the loss (was a second, one-thread launch): acc[1] is the arrival counter,
  // zeroed with acc[0] by the pack launch's riders
  __shared__ int last;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(reinterpret_cast<unsigned*>(acc + 1), 1u) + 1u == gridDim.x;
  __syncthreads();
  if (last && threadIdx.x == 0) {
    const double tot = __hip_atomic_load(acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *loss = (float)(tot / (double)(B * 3));
  }
}

__global__ __launch_bounds__(256) void gae_kernel(const float* __restrict__ rew, const float* __restrict__ v_s, const float* __restrict__ v_next,
                                                  const unsigned char* __restrict__ done, int T, int N, double gamma, double lam,
                                                  const double* __restrict__ rms, int rew_norm, double eps, double* __restrict__ ret_unnorm,
                                                  float* __restrict__ adv) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= N) return;
  const double v_scale = rew_norm ? sqrt(rms[1] + eps) : 1.0;
  double gae = 0.0;
  for (int t = T - 1; t >= 0; --t) {
    const size_t i = (size_t)t * N + e;
    const double vs = (double)v_s[i] * v_scale;
    const double dn = done[i] ? 1.0 : 0.0;
    const double vn = (double)v_next[i] * v_scale * (1.0 - dn);
    const double end = (done[i] || t == T - 1) ? 1.0 : 0.0;       // last collected index of an unfinished episode
    const double delta = (double)rew[i] + gamma * vn - vs;
    gae = delta + (1.0 - end) * gamma * lam * gae;
    adv[i] = (float)gae;
    ret_unnorm[i] = gae + vs;
  }
}
// running return statistics (tianshou RunningMeanStd) updated on device: rms = [mean, var, count] doubles
__global__ __launch_bounds__(1024) void ret_stats_kernel(const double* __restrict__ x, long long n, double* __restrict__ part) {
  __shared__ double sh[2][16];
  double s = 0.0, q = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) { s += x[i]; q += x[i] * x[i]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tq = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { ts += sh[0][i]; tq += sh[1][i]; }
    part[0] = ts; part[1] = tq;
  }
}
__global__ __launch_bounds__(256) void ret_finish_kernel(const double* __restrict__ ret_unnorm, long long n, const double* __restrict__ rms,
                                                         int rew_norm, double eps, float* __restrict__ ret_out) {
  // normalise with the OLD variance, then (thread 0 of block 0, after a grid-wide read of rms[1]) merge the batch
  const double scale = rew_norm ? sqrt(rms[1] + eps) : 1.0;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) ret_out[i] = (float)(ret_unnorm[i] / scale);
}
// The four kernels below as ONE single-workgroup launch for N <= 1024 environments (the rollout sizes of run_mansy.py: 256):
// scan, statistics of the un-normalised returns, normalisation with the OLD variance, then the running-moment merge -- the
// phases are separated by workgroup barriers instead of kernel boundaries (4 launches -> 1).  Same arithmetic, same order.
__global__ __launch_bounds__(1024) void gae_fused_kernel(const float* __restrict__ rew, const float* __restrict__ v_s, const float* __restrict__ v_next,
                                                        const unsigned char* __restrict__ done, int T, int N, double gamma, double lam,
                                                        double* __restrict__ rms, int rew_norm, double eps, double* __restrict__ ret_unnorm,
                                                        float* __restrict__ adv, float* __restrict__ ret_out) {
  __shared__ double sh[2][16];
  __shared__ double sh_scale;
  const long long n = (long long)T * N;
  const double v_scale = rew_norm ? sqrt(rms[1] + eps) : 1.0;        // every thread reads the OLD variance before anyone merges
  const int e = threadIdx.x;
  if (e < N) {
    double gae = 0.0;
    for (int t = T - 1; t >= 0; --t) {
      const size_t i = (size_t)t * N + e;
      const double vs = (double)v_s[i] * v_scale;
      const double dn = done[i] ? 1.0 : 0.0;
      const double vn = (double)v_next[i] * v_scale * (1.0 - dn);
      const double end = (done[i] || t == T - 1) ? 1.0 : 0.0;
      const double delta = (double)rew[i] + gamma * vn - vs;
      gae = delta + (1.0 - end) * gamma * lam * gae;
      adv[i] = (float)gae;
      ret_unnorm[i] = gae + vs;
    }
  }
  __syncthreads();
  double s = 0.0, q = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) { const double x = ret_unnorm[i]; s += x; q += x * x; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = q; }
  if (threadIdx.x == 0) sh_scale = v_scale;
  __syncthreads();
  for (long long i = threadIdx.x; i < n; i += blockDim.x) ret_out[i] = (float)(ret_unnorm[i] / sh_scale);
  if (threadIdx.x == 0 && rew_norm) {
    double ts = 0.0, tq = 0.0;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { ts += sh[0][i]; tq += sh[1][i]; }
    ret_unnorm[n] = ts; ret_unnorm[n + 1] = tq;                    // (the `part` slots of the four-launch form)
    const double bm = ts / (double)n, bv = tq / (double)n - bm * bm;
    const double delta = bm - rms[0], tot = rms[2] + (double)n;
    const double new_mean = rms[0] + delta * (double)n / tot;
    const double m2 = rms[1] * rms[2] + bv * (double)n + delta * delta * rms[2] * (double)n / tot;
    rms[0] = new_mean; rms[1] = m2 / tot; rms[2] = tot;
  }
}
__global__ void rms_merge_kernel(double* rms, const double* part, long long n) {
  const double bm = part[0] / (double)n, bv = part[1] / (double)n - bm * bm;
  const double delta = bm - rms[0], tot = rms[2] + (double)n;
  const double new_mean = rms[0] + delta * (double)n / tot;
  const double m2 = rms[1] * rms[2] + bv * (double)n + delta * delta * rms[2] * (double)n / tot;
  rms[0] = new_mean; rms[1] = m2 / tot; rms[2] = tot;
}

// global L2-norm gradient clipping (torch.nn.utils.clip_grad_norm_): g *= max_norm / (norm + 1e-6) if that is < 1
// squared-gradient partial sums: NORM_PARTS workgroups, one double each (no zero-fill, no atomics); consumers add them up
constexpr int NORM_PARTS = MANSY_CLIP_SCRATCH_DOUBLES;
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long long n, double* __restrict__ parts) {
  __shared__ double red[256];
  double local = 0.0;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    local += ((double)v.x * (double)v.x + (double)v.y * (double)v.y) + ((double)v.z * (double)v.z + (double)v.w * (double)v.w);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float t = g[(n4 << 2) + threadIdx.x]; local += (double)t * (double)t; }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) parts[blockIdx.x] = red[0];
}
__device__ __forceinline__ float clip_coef(const double* __restrict__ parts, float max_norm) {
  double t = 0.0;
  for (int i = 0; i < NORM_PARTS; ++i) t += parts[i];
  const float coef = max_norm / ((float)sqrt(t) + 1e-6f);
  return coef < 1.f ? coef : 1.f;
}
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, long long n, const double* __restrict__ parts, float max_norm) {
  const float coef = clip_coef(parts, max_norm);
  if (coef >= 1.f) return;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) g[i] *= coef;
}
// Adam with L2 weight decay (torch.optim.Adam(weight_decay=wd)) on gradients scaled by the global-norm clip coefficient:
// clip_grad_norm_ + step in one pass (the stored gradient is left unscaled; it is zeroed before its next use).
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                       float wd, float bc1, float sqrt_bc2, const double* __restrict__ parts,
                                                       float max_norm, const float* __restrict__ bias) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (bias) { bc1 = bias[0]; sqrt_bc2 = bias[1]; }
  const float coef = clip_coef(parts, max_norm);
  const float pp = p[i];
  const float grad = g[i] * coef + wd * pp;
  const float mm = m[i] + (grad - m[i]) * (1.f - b1);
  const float vv = v[i] * b2 + (1.f - b2) * grad * grad;
  const float denom = sqrtf(vv) / sqrt_bc2 + eps;
  p[i] = pp - (lr / bc1) * (mm / denom);
  m[i] = mm; v[i] = vv;
}
// Last launch of a PPO minibatch step and, at the same time, the prologue of the NEXT one (round 3: one launch fewer per step).
//  * clip + Adam(L2) exactly as clip_adam_kernel, then the thread ZEROES its gradient element (the next step accumulates into it) and
//    SCATTERS the parameter it has just written into the packed images the next forward reads: branch weights into the
//    block-diagonal Wbd [FEAT, KP] (the zero columns around a branch inside its K window never change), branch biases into bbd, the
//    two fc weights into the stacked [2 HID, FEAT] operand -- what pack_wbd_kernel rebuilt from scratch before every step;
//  * rider workgroups at the end of the grid do the data-dependent part of that prologue for the next minibatch, when the caller
//    names it: gather its observation rows, mean / unbiased std of its advantages, zero the packed bias-gradient accumulator and
//    the OTHER set of gradient-norm slots (the sets alternate with the Adam step count, so this launch can still read its own).
struct TailTab { long long off[28]; int numel[28]; };       // flat offsets / sizes of the 28 unique actor-critic tensors, table order
struct TailNext { const float* g_src; const int* g_idx; float* g_dst; int g_rows; const float* adv; float* adv_stats; float* dbbd; double* parts_next; };
__global__ __launch_bounds__(256) void step_tail_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ gz, float* __restrict__ m, float* __restrict__ v,
                                                       long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float sqrt_bc2,
                                                       const double* __restrict__ parts, float max_norm, TailTab tab, float* __restrict__ Wbd,
                                                       float* __restrict__ bbd, float* __restrict__ Wfc2, TailNext nx, int adam_blocks, const float* __restrict__ bias) {
  if (bias) { bc1 = bias[0]; sqrt_bc2 = bias[1]; }      // a graph-replayed step: this replay's bias corrections from device memory (include/mansy_hip.h, adam_bias)
  if ((int)blockIdx.x >= adam_blocks) {
    const int rb = blockIdx.x - adam_blocks;
    if (rb == 0) {
      for (int i = threadIdx.x; i < FEAT; i += 256) nx.dbbd[i] = 0.f;
      if (threadIdx.x < NORM_PARTS_C) nx.parts_next[threadIdx.x] = 0.0;
      if (nx.adv && nx.g_rows >= 1) adv_stats_block(nx.adv, nx.g_idx, nx.g_rows, nx.adv_stats);
      return;
    }
    const long long i2 = (long long)(rb - 1) * 256 + threadIdx.x;
    if (nx.g_src && i2 < (long long)nx.g_rows * (OBS_LD / 4)) {
      const int r = (int)(i2 / (OBS_LD / 4)), c4 = (int)(i2 % (OBS_LD / 4));
      reinterpret_cast<float4*>(nx.g_dst)[(size_t)r * (OBS_LD / 4) + c4] = reinterpret_cast<const float4*>(nx.g_src)[(size_t)nx.g_idx[r] * (OBS_LD / 4) + c4];
    }
    return;
  }
  // four consecutive elements per thread (16-byte accesses; the tensors start on 256-byte boundaries, so the four share one tensor's slot)
  const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= n) return;
  const float coef = clip_coef(parts, max_norm);
  float pp[4], gg[4], mo[4], vo[4], pn[4];
  const bool full = i + 4 <= n;
  if (full) {
    *reinterpret_cast<float4*>(pp) = *reinterpret_cast<const float4*>(p + i); *reinterpret_cast<float4*>(gg) = *reinterpret_cast<const float4*>(g + i);
    *reinterpret_cast<float4*>(mo) = *reinterpret_cast<const float4*>(m + i); *reinterpret_cast<float4*>(vo) = *reinterpret_cast<const float4*>(v + i);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) { const bool in = i + q < n; pp[q] = in ? p[i + q] : 0.f; gg[q] = in ? g[i + q] : 0.f; mo[q] = in ? m[i + q] : 0.f; vo[q] = in ? v[i + q] : 0.f; }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float grad = gg[q] * coef + wd * pp[q];
    const float mm = mo[q] + (grad - mo[q]) * (1.f - b1);
    const float vv = vo[q] * b2 + (1.f - b2) * grad * grad;
    const float denom = sqrtf(vv) / sqrt_bc2 + eps;
    pn[q] = pp[q] - (lr / bc1) * (mm / denom);
    mo[q] = mm; vo[q] = vv;
  }
  if (full) {
    *reinterpret_cast<float4*>(p + i) = *reinterpret_cast<const float4*>(pn); *reinterpret_cast<float4*>(m + i) = *reinterpret_cast<const float4*>(mo);
    *reinterpret_cast<float4*>(v + i) = *reinterpret_cast<const float4*>(vo); *reinterpret_cast<float4*>(gz + i) = make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i + q < n) { p[i + q] = pn[q]; m[i + q] = mo[q]; v[i + q] = vo[q]; gz[i + q] = 0.f; }
  }
  // which tensor? (offsets ascending; the 256-byte alignment gaps between tensors belong to none)
  int lo = 0, hi = 27;
#pragma unroll
  for (int it = 0; it < 5; ++it) { const int mid = (lo + hi + 1) >> 1; if (tab.off[mid] <= i) lo = mid; else hi = mid - 1; }
  const long long e0 = i - tab.off[lo];
  const int numel = tab.numel[lo];
  if (e0 >= numel) return;
  if (lo < 2 * NB) {
    const int j = lo >> 1;
    if (lo & 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (e0 + q < numel) bbd[j * HID + (int)e0 + q] = pn[q];
      return;
    }
    const Branch gm = branch_geom(j, 0);
    int r = (int)e0 / gm.len, c = (int)e0 % gm.len;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (e0 + q < numel) Wbd[(long long)(j * HID + r) * KP + gm.off + c] = pn[q];
      if (++c == gm.len) { c = 0; ++r; }
    }
  } else if (lo == 2 * NB || lo == 2 * NB + 4) {          // the two fc weights: [HID, FEAT] each, numel % 4 == 0
    float* dst = Wfc2 + (lo == 2 * NB ? 0 : (long long)HID * FEAT) + e0;
    if (e0 + 4 <= numel) *reinterpret_cast<float4*>(dst) = *reinterpret_cast<const float4*>(pn);
    else for (int q = 0; q < 4 && e0 + q < numel; ++q) dst[q] = pn[q];
  }
}

// ------------------------------------------------------------------------------------ workspace
struct PWork {
  float *Wbd, *bbd, *F, *A1a, *Ha, *A1c, *Hc, *outa, *outc, *dHa, *dHc, *dA1a, *dA1c, *dF, *dWbd, *dbbd, *obs_mb, *gout, *gout_c;
  float* A1s;      // split-K slabs of a head's fc product (head(), head_pair())
  float *Wfc2, *dA1p;   // [actor.fc | critic.fc] stacked [2*HID, FEAT]; their dA1 side by side [B, 2*HID]
  int* krange;     // per 64-feature tile K range of the packed block-diagonal image
  int* tlist;      // (column tile, row tile) pairs of the weight-gradient product's tiles on the block diagonal
  double* acc;
  float *adv_stats, *lossrows;   // fused PPO loss: minibatch advantage mean / std; per-row (clip, vf, ent, -) terms
};
// The head's fc product is [B,1280] x [1280,128]: 2 column tiles however large K is, so for B <= 2048 it is split over K
// into slabs (GemmEpilogue::split_slab) that head_out_kernel sums -- 16x fewer K-tiles on the critical path at B = 256.
inline int head_split_request(int B, int n_cols = HID) {
  const int tiles = mansy_ceil_div(B, 64) * (n_cols / 64);
  const int req = 256 / tiles;
  return req < 2 ? 1 : (req > 16 ? 16 : req);
}
inline int head_slab_rows(int maxB) {
  int rows = maxB;
  for (int B = 1; B <= maxB; B = B < 64 ? 64 : B + 64) {
    const int r = mansy_gemm_effective_splits(FEAT, head_split_request(B)) * B;
    if (r > rows) rows = r;
  }
  return rows + 64;
}
size_t ppo_layout(int maxB, char* base, PWork& W) {
  size_t tot = 0;
  auto f = [&](size_t n) { const size_t off = (tot + 255) & ~size_t(255); tot = off + n * sizeof(float); return (float*)(base ? base + off : nullptr); };
  W.Wbd = f((size_t)FEAT * KP); W.bbd = f(FEAT); W.krange = (int*)f(2 * FEAT / 64); W.tlist = (int*)f(2 * DW_TILES_MAX); W.F = f((size_t)maxB * FEAT);
  W.A1a = f((size_t)maxB * HID); W.Ha = f((size_t)maxB * HID); W.A1c = f((size_t)maxB * HID); W.Hc = f((size_t)maxB * HID);
  W.outa = f((size_t)maxB * MAXOUT); W.outc = f((size_t)maxB * MAXOUT);
  W.dHa = f((size_t)maxB * HID); W.dHc = f((size_t)maxB * HID); W.dA1a = f((size_t)maxB * HID); W.dA1c = f((size_t)maxB * HID);
  W.dF = f((size_t)maxB * FEAT); W.dWbd = f((size_t)DW_SLABS * FEAT * K_IDENT); W.dbbd = f(FEAT); W.obs_mb = f((size_t)maxB * OBS_LD);
  W.gout = f((size_t)maxB * MAXOUT); W.gout_c = f((size_t)maxB * MAXOUT);
  W.A1s = f((size_t)head_slab_rows(maxB) * 2 * HID);
  W.Wfc2 = f((size_t)2 * HID * FEAT); W.dA1p = f((size_t)maxB * 2 * HID);
  W.acc = (double*)f(4 * 64);      // 2 x NORM_PARTS doubles (gradient-norm partial sums, the sets alternate with the Adam step count in a
                                   // chained PPO update; set 0 is also the identifier-loss accumulator)
  W.adv_stats = f(8); W.lossrows = f((size_t)maxB * 4);
  return tot + 256;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// the launches that make up most of a cycle run compile-time instances of head_out_kernel (MODE 1 / 2 / 3); a lab build (csrc/lab/) can send
// them all to the generic instance (variant bit 0x1000) for A/B timing -- a release build reads no switch
static bool head_out_modes() { return (mansy_variant_of(0) & 0x1000) == 0; }

struct PEng {
  hipStream_t st; PWork W;
  int prec = 0;      // MANSY_PREC_* of this call's products (the entry point's `precision` argument)
  // pair != nullptr: also stack the fc weights of (n, *pair) for head_pair()
  int pack(const NetP& n, int identifier, const NetP* pair = nullptr, const float* g_src = nullptr, const int* g_idx = nullptr, int g_rows = 0,
           float* zero_ptr = nullptr, long long zero_n = 0) {
    PackArgs a; for (int j = 0; j < NB; ++j) { a.bw[j] = n.bw[j]; a.bb[j] = n.bb[j]; }
    a.adv = nullptr; a.adv_n = 0; a.adv_stats = nullptr;
    a.fc_a = n.fc_w; a.fc_c = pair ? pair->fc_w : nullptr; a.Wfc2 = pair ? W.Wfc2 : nullptr;
    a.g_src = g_src; a.g_idx = g_idx; a.g_dst = W.obs_mb; a.g_rows = g_src ? g_rows : 0;
    a.zero_ptr = zero_ptr; a.zero_n = zero_ptr ? zero_n : 0;
    a.z2 = W.dbbd; a.z2n = FEAT; a.z3 = reinterpret_cast<float*>(W.acc); a.z3n = 4 * NORM_PARTS_C;
    MANSY_REQUIRE(!zero_ptr || (reinterpret_cast<uintptr_t>(zero_ptr) & 15) == 0, "pack: gradient buffer must be 16-byte aligned");
    const int K = identifier ? K_IDENT : K_POLICY;
    const long long n_win = (long long)HID * window_cols(identifier);
    const long long threads = n_win + (pair ? 2LL * HID * FEAT : 0) + (long long)a.g_rows * (OBS_LD / 4) + (a.zero_n + 3) / 4;
    MANSY_LAUNCH(pack_wbd_kernel, dim3(mansy_ceil_div(threads, 256)), dim3(256), 0, st, a, identifier, K, n_win, W.Wbd, W.bbd, W.krange, W.tlist);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  // PPO minibatch prologue: re-pack, gather rows (idx != null), zero gradients, advantage statistics
  int pack_mb(const NetP& a, const NetP& c, const float* g_src, const int* idx, int mb, float* zero_ptr, long long zero_n, const float* adv) {
    PackArgs pa; for (int j = 0; j < NB; ++j) { pa.bw[j] = a.bw[j]; pa.bb[j] = a.bb[j]; }
    pa.fc_a = a.fc_w; pa.fc_c = c.fc_w; pa.Wfc2 = W.Wfc2;
    pa.g_src = g_src; pa.g_idx = idx; pa.g_dst = W.obs_mb; pa.g_rows = g_src ? mb : 0;
    pa.zero_ptr = zero_ptr; pa.zero_n = zero_ptr ? zero_n : 0;
    pa.adv = adv; pa.adv_n = mb; pa.adv_stats = W.adv_stats;
    pa.z2 = W.dbbd; pa.z2n = FEAT; pa.z3 = reinterpret_cast<float*>(W.acc); pa.z3n = 4 * NORM_PARTS_C;
    MANSY_REQUIRE(!zero_ptr || (reinterpret_cast<uintptr_t>(zero_ptr) & 15) == 0, "pack: gradient buffer must be 16-byte aligned");
    const long long n_win = (long long)HID * window_cols(0);
    const long long threads = n_win + 2LL * HID * FEAT + (long long)pa.g_rows * (OBS_LD / 4) + (pa.zero_n + 3) / 4;
    MANSY_LAUNCH(pack_wbd_kernel, dim3(mansy_ceil_div(threads, 256) + 1), dim3(256), 0, st, pa, 0, K_POLICY, n_win, W.Wbd, W.bbd, W.krange, W.tlist);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  int featnet(const float* obs, int B, int identifier) {
    const int K = identifier ? K_IDENT : K_POLICY;
    (void)K;      // the packed image spans KP columns (zero beyond K); obs rows are OBS_LD >= KP floats
    GemmEpilogue ep; ep.prec = prec; ep.bias = W.bbd; ep.relu = 1; ep.relu_slope = SLOPE; ep.tile_krange = W.krange;
    // the packed image is only defined inside the K windows: the product must run on a loop that honours tile_krange -- the LDS-DMA
    // loop or its split-bf16 twin, i.e. K a multiple of 32 (KP), leading dimensions multiples of 4, 16-byte aligned operands
    static_assert(KP % 32 == 0 && OBS_LD % 4 == 0, "FeatureNet product must qualify for the LDS-DMA loop");
    MANSY_REQUIRE((reinterpret_cast<uintptr_t>(obs) & 15) == 0 && (reinterpret_cast<uintptr_t>(W.Wbd) & 15) == 0,
                  "featnet: observation rows and workspace must be 16-byte aligned");
    return mansy_launch_gemm_f32(obs, OBS_LD, 0, W.Wbd, KP, 0, W.F, FEAT, B, FEAT, KP, ep, 0, 0, st);
  }
  struct HeadRiders { const int* act_given = nullptr; int n_given = 0; float* logp = nullptr;
                      float* relabel_rew = nullptr; float* relabel_idrew = nullptr; const float* relabel_obs = nullptr; float relabel_lamb = 0.f; };
  int head(const NetP& n, int B, int n_out, int sigmoid, float* A1, float* H, float* out, const float* u, uint32_t seed, uint32_t site, int* act,
           float* logp, const EnvFuse* env = nullptr, int out_ld = 0, const HeadRiders* rd = nullptr, const LossFuse* fuse = nullptr) {
    const int req = head_split_request(B);
    int nsplit = 0;
    if (req > 1) {
      nsplit = mansy_gemm_effective_splits(FEAT, req);
      MANSY_REQUIRE(nsplit <= MAX_SLABS, "head: %d K splits exceed the slab sum's unroll", nsplit);
      GemmEpilogue ep; ep.prec = prec; ep.split_slab = (long long)B * HID;
      RC(mansy_launch_gemm_f32(W.F, FEAT, 0, n.fc_w, FEAT, 0, W.A1s, HID, B, HID, FEAT, ep, 0, req, st));
    } else {
      GemmEpilogue ep; ep.prec = prec; ep.bias = n.fc_b; ep.relu = 1; ep.relu_slope = SLOPE;
      RC(mansy_launch_gemm_f32(W.F, FEAT, 0, n.fc_w, FEAT, 0, A1, HID, B, HID, FEAT, ep, 0, 0, st));
    }
    HeadOutArgs ha;
    ha.h[0] = {A1, n.fc_b, n.out_w, n.out_b, n_out, sigmoid, H, out, 0, act, logp, out_ld};
    if (rd) {
      ha.h[0].act_given = rd->act_given; ha.h[0].n_given = rd->n_given; if (rd->logp) ha.h[0].logp = rd->logp;
      ha.h[0].relabel_rew = rd->relabel_rew; ha.h[0].relabel_idrew = rd->relabel_idrew; ha.h[0].relabel_obs = rd->relabel_obs; ha.h[0].relabel_lamb = rd->relabel_lamb;
    }
    ha.h[1] = ha.h[0];
    LossFuse none; memset(&none, 0, sizeof(none));
    if (fuse) none = *fuse;
    EnvFuse ef; memset(&ef, 0, sizeof(ef));
    if (env) ef = *env;
    // the rollout launch has its own instance of the kernel (MODE 1)
    if (head_out_modes() && env && ef.on && act && nsplit > 0 && !sigmoid && !rd && !fuse)
      MANSY_LAUNCH(head_out_kernel<1>, dim3(mansy_ceil_div(B, 4), 1), dim3(256), 0, st, ha, W.A1s, nsplit, (long long)B * HID, HID, W.F, MAXOUT, B, u, seed, site, none, ef);
    else if (head_out_modes() && !env && !act && !fuse)      // forward only (+ riders): MODE 3
      MANSY_LAUNCH(head_out_kernel<3>, dim3(mansy_ceil_div(B, 4), 1), dim3(256), 0, st, ha, W.A1s, nsplit, (long long)B * HID, HID, W.F, MAXOUT, B, u, seed, site, none, ef);
    else
      MANSY_LAUNCH(head_out_kernel<0>, dim3(mansy_ceil_div(B, 4), 1), dim3(256), 0, st, ha, W.A1s, nsplit, (long long)B * HID, HID, W.F, MAXOUT, B, u, seed, site, none, ef);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  // actor + critic on the shared features in one product ([B,1280] x [1280,256], K-split slabs) and one head_out launch
  // value != nullptr: the critic's output goes straight into that [B] vector (no strided copy afterwards)
  int head_pair(const NetP& a, const NetP& c, int B, const LossFuse* fuse = nullptr, float* value = nullptr, const HeadRiders* rd = nullptr) {
    const int req = head_split_request(B, 2 * HID);
    const int nsplit = mansy_gemm_effective_splits(FEAT, req);
    MANSY_REQUIRE(nsplit <= MAX_SLABS, "head_pair: %d K splits exceed the slab sum's unroll", nsplit);
    GemmEpilogue ep; ep.prec = prec; ep.split_slab = (long long)B * 2 * HID;
    RC(mansy_launch_gemm_f32(W.F, FEAT, 0, W.Wfc2, FEAT, 0, W.A1s, 2 * HID, B, 2 * HID, FEAT, ep, 0, req, st));
    HeadOutArgs ha;
    ha.h[0] = {W.A1a, a.fc_b, a.out_w, a.out_b, NACT, 0, W.Ha, W.outa, 0, nullptr, nullptr, 0};
    ha.h[1] = {W.A1c, c.fc_b, c.out_w, c.out_b, 1, 0, W.Hc, value ? value : W.outc, HID, nullptr, nullptr, value ? 1 : 0};
    if (rd) { ha.h[0].act_given = rd->act_given; ha.h[0].n_given = rd->n_given; ha.h[0].logp = rd->logp; }
    LossFuse lf; memset(&lf, 0, sizeof(lf));
    if (fuse) lf = *fuse;
    EnvFuse noenv; memset(&noenv, 0, sizeof(noenv));
    // the minibatch step's launch has its own instance of the kernel (MODE 2)
    if (head_out_modes() && fuse && lf.on == 1 && !rd)
      MANSY_LAUNCH(head_out_kernel<2>, dim3(mansy_ceil_div(B, 4), 2), dim3(256), 0, st, ha, W.A1s, nsplit, (long long)B * 2 * HID, 2 * HID, W.F, MAXOUT, B, nullptr, 0u, 0u, lf, noenv);
    else if (head_out_modes() && !fuse)                       // forward only (+ the log-probability rider): MODE 3
      MANSY_LAUNCH(head_out_kernel<3>, dim3(mansy_ceil_div(B, 4), 2), dim3(256), 0, st, ha, W.A1s, nsplit, (long long)B * 2 * HID, 2 * HID, W.F, MAXOUT, B, nullptr, 0u, 0u, lf, noenv);
    else
      MANSY_LAUNCH(head_out_kernel<0>, dim3(mansy_ceil_div(B, 4), 2), dim3(256), 0, st, ha, W.A1s, nsplit, (long long)B * 2 * HID, 2 * HID, W.F, MAXOUT, B, nullptr, 0u, 0u, lf, noenv);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  // backward of one head given g = dL/d(out pre-sigmoid) [B,MAXOUT]: param grads + dF contribution (accumulate)
  int head_bwd(const NetP& n, int B, int n_out, const float* g, const float* A1, const float* H, float* dH, float* dA1, bool accumulate_dF) {
    HeadBwdArgs hb;
    hb.h[0] = {g, MAXOUT, A1, H, n.out_w, n_out, dH, dA1, HID, n.gout_w, n.gout_b};
    hb.h[1] = hb.h[0];
    LossFinish nofin; memset(&nofin, 0, sizeof(nofin));
    MANSY_LAUNCH(head_out_bwd_kernel, dim3(min(mansy_ceil_div(B, 2), 128), 1), dim3(256), 0, st, hb, B, nofin);
    MANSY_LAUNCH_CHECK();
    GemmEpilogue acc; acc.prec = prec; acc.accumulate = 1; acc.a_rowsum = n.gfc_b;                                           // gfc_b += column sums of dA1
    RC(mansy_launch_gemm_f32(dA1, HID, 1, W.F, FEAT, 1, n.gfc_w, FEAT, HID, FEAT, B, acc, 0, 0, st));      // gfc_w += dA1^T F
    MANSY_REQUIRE(!accumulate_dF, "head_bwd: accumulating dF is not supported with the fused LeakyReLU-derivative epilogue");
    // dPre = (dA1 Wfc + [dH in the residual columns]) * leaky'(F): residual join + derivative in the product's epilogue
    GemmEpilogue ep; ep.prec = prec; ep.pre_a = dH; ep.pre_ld = HID; ep.pre_col0 = RESID_COL; ep.mask_src = W.F; ep.mask_ld = FEAT; ep.mask_scale = 1.f; ep.mask_neg = SLOPE;
    return mansy_launch_gemm_f32(dA1, HID, 0, n.fc_w, FEAT, 1, W.dF, FEAT, B, FEAT, HID, ep, 0, 0, st);
  }
  // both heads' backward: one output-layer launch, the two fc weight gradients, ONE dF = [dA1a | dA1c] [Wfc_a ; Wfc_c] product
  int head_bwd_pair(const NetP& a, const NetP& c, int B, const LossFinish* finish = nullptr) {
    HeadBwdArgs hb;
    hb.h[0] = {W.gout, MAXOUT, W.A1a, W.Ha, a.out_w, NACT, W.dHa, W.dA1p, 2 * HID, a.gout_w, a.gout_b};
    hb.h[1] = {W.gout_c, MAXOUT, W.A1c, W.Hc, c.out_w, 1, W.dHc, W.dA1p + HID, 2 * HID, c.gout_w, c.gout_b};
    LossFinish fin; memset(&fin, 0, sizeof(fin));
    if (finish) fin = *finish;
    MANSY_LAUNCH(head_out_bwd_kernel, dim3(min(mansy_ceil_div(B, 2), HB_BLOCKS), 2), dim3(256), 0, st, hb, B, fin);
    MANSY_LAUNCH_CHECK();
    GemmEpilogue acc; acc.prec = prec; acc.accumulate = 1;           // gfc_w_{a,c} += dA1_{a,c}^T F, gfc_b_{a,c} += column sums: two products, one launch
    acc.a_rowsum = a.gfc_b;
    acc.pair_A = W.dA1p + HID; acc.pair_B = W.F; acc.pair_C = c.gfc_w; acc.pair_rowsum = c.gfc_b;
    RC(mansy_launch_gemm_f32(W.dA1p, 2 * HID, 1, W.F, FEAT, 1, a.gfc_w, FEAT, HID, FEAT, B, acc, 0, 0, st));
    // dPre = ([dA1a | dA1c] [Wfc_a ; Wfc_c] + [dHa + dHc in the residual columns]) * leaky'(F) -- the former featgrad_finish launch
    GemmEpilogue ep; ep.prec = prec; ep.pre_a = W.dHa; ep.pre_b = W.dHc; ep.pre_ld = HID; ep.pre_col0 = RESID_COL;
    ep.mask_src = W.F; ep.mask_ld = FEAT; ep.mask_scale = 1.f; ep.mask_neg = SLOPE;
    return mansy_launch_gemm_f32(W.dA1p, 2 * HID, 0, W.Wfc2, FEAT, 1, W.dF, FEAT, B, FEAT, 2 * HID, ep, 0, 0, st);
  }
  // head_bwd without the output-layer launch (the identifier's training step): dA1 / dH were written by head_out_kernel
  int fc_bwd_single(const NetP& n, int B, const float* dA1, const float* dH) {
    // the fc weight gradient and the dF product both read dA1 and feed different consumers: one launch where both are small (mansy_gemm_pair_*)
    const int paired = mansy_gemm_pair_begin();
    GemmEpilogue acc; acc.prec = prec; acc.accumulate = 1; acc.a_rowsum = n.gfc_b;
    int rc = mansy_launch_gemm_f32(dA1, HID, 1, W.F, FEAT, 1, n.gfc_w, FEAT, HID, FEAT, B, acc, 0, 0, st);
    GemmEpilogue ep; ep.prec = prec; ep.pre_a = dH; ep.pre_ld = HID; ep.pre_col0 = RESID_COL; ep.mask_src = W.F; ep.mask_ld = FEAT; ep.mask_scale = 1.f; ep.mask_neg = SLOPE;
    if (!rc) rc = mansy_launch_gemm_f32(dA1, HID, 0, n.fc_w, FEAT, 1, W.dF, FEAT, B, FEAT, HID, ep, 0, 0, st);
    if (paired) { const int rc2 = mansy_gemm_pair_end(st); if (!rc) rc = rc2; }
    return rc;
  }
  // head_bwd_pair without the output-layer launch: head_out_kernel wrote dH / dA1 itself (LossFuse::bwd_*), the output layers' weight
  // gradients are riders of the unpack launch (OutGradRider) -- the PPO minibatch step's form
  int fc_bwd_pair(const NetP& a, const NetP& c, int B) {
    const int paired = mansy_gemm_pair_begin();      // the two fc weight gradients and the dF product: independent readers of dA1 (see fc_bwd_single)
    GemmEpilogue acc; acc.prec = prec; acc.accumulate = 1;
    acc.a_rowsum = a.gfc_b;
    acc.pair_A = W.dA1p + HID; acc.pair_B = W.F; acc.pair_C = c.gfc_w; acc.pair_rowsum = c.gfc_b;
    int rc = mansy_launch_gemm_f32(W.dA1p, 2 * HID, 1, W.F, FEAT, 1, a.gfc_w, FEAT, HID, FEAT, B, acc, 0, 0, st);
    GemmEpilogue ep; ep.prec = prec; ep.pre_a = W.dHa; ep.pre_b = W.dHc; ep.pre_ld = HID; ep.pre_col0 = RESID_COL;
    ep.mask_src = W.F; ep.mask_ld = FEAT; ep.mask_scale = 1.f; ep.mask_neg = SLOPE;
    if (!rc) rc = mansy_launch_gemm_f32(W.dA1p, 2 * HID, 0, W.Wfc2, FEAT, 1, W.dF, FEAT, B, FEAT, 2 * HID, ep, 0, 0, st);
    if (paired) { const int rc2 = mansy_gemm_pair_end(st); if (!rc) rc = rc2; }
    return rc;
  }
  // norm_tail != nullptr: also leave the squared norm of ALL gradients (branches + [norm_tail, norm_tail + norm_tail_n)) in W.acc
  int featnet_bwd(const NetP& n, const float* obs, int B, int identifier, const float* dHa, const float* dHb, const float* norm_tail = nullptr,
                  long long norm_tail_n = 0, double* norm_parts = nullptr, const OutGradRider* out_grad = nullptr) {
    const int K = identifier ? K_IDENT : K_POLICY;
    (void)dHa; (void)dHb;      // joined inside the dF product's epilogue (head_bwd / head_bwd_pair); dbbd and the norm slots were zeroed by pack
    // dWbd = dPre^T obs, wanted on the block diagonal only: the launch runs the 42 of 240 output tiles that meet a branch's window
    // (GemmEpilogue::tile_list / tile_nrange, tables written by the pack launch) and splits the batch (= K of this product) into
    // slabs that the unpack launch sums.  dbbd = column sums of dPre ride on the staged A tiles as before.
    const int Bmain = (B % 32 != 0 && B >= 256) ? B / 32 * 32 : B;          // whole K-tiles on the LDS-DMA loop, the rest added below
    // ~3 K-tiles per split: the 42 running tiles are one workgroup each, and a lone workgroup per CU waits a full memory round trip
    // per K-tile (2 us at K = 3 264 from HBM), so the reduce dimension is what fills the chip
    int req = std::min(DW_SLABS, std::max(1, (Bmain / 32 + 2) / 3));
    if (prec != 0) req = std::min(req, 6);            // the split-bf16 loops run (and store) every tile of every slab
    else if (mansy_gemm_wsk_tn_enabled()) {
      // minibatch-sized products run on the wave-split-K loop, which splits the K-tiles over a workgroup's four waves itself: ~12 K-tiles per slab
      // there (3 per wave) -- fewer, larger workgroups and fewer slabs for the unpack launch to add
      const int req_w = std::max(1, (Bmain / 32 + 8) / 12);
      if ((long long)active_tiles(identifier, K, nullptr) * req_w <= 256) req = req_w;
    }
    const int nsplit = mansy_gemm_effective_splits(Bmain, req);
    const long long slab = (long long)FEAT * K;
    MANSY_REQUIRE(nsplit <= DW_SLABS, "featnet_bwd: %d slabs exceed the workspace", nsplit);
    const int active = active_tiles(identifier, K, nullptr);             // host copy of the geometry pack_wbd_kernel writes
    MANSY_REQUIRE(active <= DW_TILES_MAX, "featnet_bwd: %d tiles exceed the list", active);
    GemmEpilogue ep; ep.prec = prec; ep.a_rowsum = W.dbbd; ep.tile_nrange = W.krange; ep.split_slab = nsplit > 1 ? slab : 0;
    ep.tile_list = W.tlist; ep.tile_list_n = active;
    ep.flops_frac = (float)active / (float)(FEAT / 64 * mansy_ceil_div(K, 64));
    RC(mansy_launch_gemm_f32(W.dF, FEAT, 1, obs, OBS_LD, 1, W.dWbd, K, FEAT, K, Bmain, ep, 64, nsplit > 1 ? req : 1, st));
    if (Bmain < B) {                                                        // the < 32 leftover rows: added into slab 0
      GemmEpilogue tail; tail.prec = prec; tail.a_rowsum = W.dbbd; tail.tile_nrange = W.krange; tail.accumulate = 1; tail.flops_frac = ep.flops_frac;
      tail.tile_list = W.tlist; tail.tile_list_n = active;
      RC(mansy_launch_gemm_f32(W.dF + (size_t)Bmain * FEAT, FEAT, 1, obs + (size_t)Bmain * OBS_LD, OBS_LD, 1, W.dWbd, K, FEAT, K, B - Bmain, tail,
                               -64, 1, st));
    }
    UnpackArgs u; for (int j = 0; j < NB; ++j) { u.gbw[j] = n.gbw[j]; u.gbb[j] = n.gbb[j]; }
    NormRider nr; nr.parts = norm_tail ? (norm_parts ? norm_parts : W.acc) : nullptr; nr.tail_g = norm_tail; nr.tail_n = norm_tail_n;
    const long long main_blocks = mansy_ceil_div((long long)HID * compact_cols(identifier), 256);
    const long long tail_blocks = norm_tail ? mansy_ceil_div(mansy_ceil_div(norm_tail_n, 4), 256) : 0;
    MANSY_REQUIRE(!norm_tail || (reinterpret_cast<uintptr_t>(norm_tail) & 15) == 0, "featnet_bwd: gradient tail must be 16-byte aligned");
    OutGradRider og; memset(&og, 0, sizeof(og));
    if (out_grad) og = *out_grad;
    MANSY_LAUNCH(unpack_dwbd_kernel, dim3(main_blocks + tail_blocks + (og.on ? OG_BLOCKS : 0)), dim3(256), 0, st, W.dWbd, nsplit, slab, W.dbbd,
                       identifier, K, u, nr, og, (int)(main_blocks + tail_blocks));
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  // clip + Adam + gradient zero-fill + re-pack of the updated parameters + (next != null) the next minibatch's gather / statistics
  // g_zero: the buffer the NEXT step accumulates its gradients into (zero-filled here); null = flat_g itself.  The data-parallel peer form hands over
  // the other exchange slot: this step's gradients were produced in one slot and averaged into flat_g, the next step's go into the other slot.
  int step_tail(const float* const* params, float* flat_p, float* flat_g, float* m, float* v, long long n, float max_norm, float lr, float wd, int step,
                double* parts_cur, double* parts_next, const float* obs_all, const int* next_idx, int next_mb, const float* adv_all, float* g_zero = nullptr,
                const float* bias_dev = nullptr) {
    TailTab tab;
    static const std::vector<ParamInfo> t = net_table(0);       // (built once: 28 entries with std::string names -- this runs in every minibatch step)
    MANSY_REQUIRE(t.size() == 28, "step_tail: parameter table changed");
    for (int k = 0; k < 28; ++k) {
      tab.off[k] = params[k] - flat_p; tab.numel[k] = (int)t[k].numel;
      MANSY_REQUIRE(tab.off[k] >= 0 && tab.off[k] + tab.numel[k] <= n && (k == 0 || tab.off[k] >= tab.off[k - 1] + tab.numel[k - 1]),
                    "step_tail: params[] must be ascending views of flat_p");
    }
    for (int k = 0; k < 28; ++k) MANSY_REQUIRE(tab.off[k] % 4 == 0, "step_tail: tensor %d does not start on a 16-byte boundary of flat_p", k);
    MANSY_REQUIRE(((reinterpret_cast<uintptr_t>(flat_p) | reinterpret_cast<uintptr_t>(flat_g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                  "step_tail: flat buffers must be 16-byte aligned");
    TailNext nx; nx.g_src = next_idx ? obs_all : nullptr; nx.g_idx = next_idx; nx.g_dst = W.obs_mb; nx.g_rows = next_mb;
    nx.adv = next_mb > 0 ? adv_all : nullptr; nx.adv_stats = W.adv_stats; nx.dbbd = W.dbbd; nx.parts_next = parts_next;
    const int adam_blocks = (int)mansy_ceil_div(mansy_ceil_div(n, 4), 256);       // four elements per thread
    const int rider_blocks = 1 + (next_idx ? (int)mansy_ceil_div((long long)next_mb * (OBS_LD / 4), 256) : 0);
    const double bc1 = 1.0 - pow(0.9, (double)step), bc2 = 1.0 - pow(0.999, (double)step);
    MANSY_REQUIRE(!g_zero || (reinterpret_cast<uintptr_t>(g_zero) & 15) == 0, "step_tail: the next gradient buffer must be 16-byte aligned");
    MANSY_LAUNCH(step_tail_kernel, dim3(adam_blocks + rider_blocks), dim3(256), 0, st, flat_p, (const float*)flat_g, g_zero ? g_zero : flat_g, m, v, n, lr, 0.9f, 0.999f, 1e-8f, wd,
                       (float)bc1, (float)sqrt(bc2), parts_cur, max_norm, tab, W.Wbd, W.bbd, W.Wfc2, nx, adam_blocks, bias_dev);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  // tail_from >= 0: elements [tail_from, n) carry Adam step count `tail_step` instead of `step` (parameters that received
  // their first gradient later than the rest: torch keeps one step counter per parameter); tail_step <= 0 never happens
  // with gradients present, so it is rejected.  The clip coefficient is the global one for both ranges.
  int clip_and_adam(float* flat_p, float* flat_g, float* m, float* v, long long n, float max_norm, float lr, float wd, int step,
                    long long tail_from = -1, int tail_step = 0, bool have_sumsq = false, const float* bias_dev = nullptr) {
    if (max_norm > 0.f) {
      if (!have_sumsq) MANSY_LAUNCH(sumsq_kernel, dim3(NORM_PARTS), dim3(256), 0, st, flat_g, n, W.acc);
      if (step <= 0) MANSY_LAUNCH(clip_scale_kernel, dim3(256), dim3(256), 0, st, flat_g, n, W.acc, max_norm);   // parity tests: clipped gradients
      MANSY_LAUNCH_CHECK();
    }
    if (step <= 0) return MANSY_OK;
    const bool lag = tail_from >= 0 && tail_from < n && tail_step != step;
    if (lag) MANSY_REQUIRE(tail_step >= 1 && tail_from % 4 == 0, "adam: lagged tail needs tail_step >= 1 and a 16-byte aligned start");
    MANSY_REQUIRE(!(lag && bias_dev), "adam: device-side bias corrections (adam_bias) and a lagged tail exclude each other");
    const long long n_head = lag ? tail_from : n;
    for (int part = 0; part < (lag ? 2 : 1); ++part) {
      const long long o = part ? tail_from : 0, cnt = part ? n - tail_from : n_head;
      const int stp = part ? tail_step : step;
      if (cnt <= 0) continue;
      if (max_norm <= 0.f) {                     // Adam with L2 (run_mansy.py:216,226)
        int rc = mansy_launch_adamw(flat_p + o, flat_g + o, m + o, v + o, cnt, lr, 0.9f, 0.999f, 1e-8f, wd, stp, 0, st, bias_dev);
        if (rc) return rc;
        continue;
      }
      const double bc1 = 1.0 - pow(0.9, (double)stp), bc2 = 1.0 - pow(0.999, (double)stp);
      MANSY_LAUNCH(clip_adam_kernel, dim3(mansy_ceil_div(cnt, 256)), dim3(256), 0, st, flat_p + o, flat_g + o, m + o, v + o, cnt, lr, 0.9f,
                         0.999f, 1e-8f, wd, (float)bc1, (float)sqrt(bc2), W.acc, max_norm, bias_dev);
    }
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
};

int setup(void* ws, int maxB, int precision, hipStream_t st, PEng& e) {
  MANSY_REQUIRE(precision == 0 || precision == 1 || precision == 3 || precision == 6, "precision must be MANSY_PREC_F32 (0), _BF16 (1), _BF16X3 (3) or _BF16X6 (6), got %d", precision);
  e.prec = precision;
  MANSY_REQUIRE(ws && maxB >= 1, "ppo: bad workspace / batch");
  e.st = st;
  ppo_layout(maxB, (char*)ws, e.W);
  return MANSY_OK;
}

}  // namespace

extern "C" {

int mansy_net_num_params(int kind) { return (int)net_table(kind != 0).size(); }
int mansy_net_param_info(int kind, int idx, char* name, int name_len, long long* numel, int* ndim, long long shape[4]) {
  const std::vector<ParamInfo> t = net_table(kind != 0);
  MANSY_REQUIRE(idx >= 0 && idx < (int)t.size(), "net_param_info: index %d out of range", idx);
  if (name && name_len > 0) { strncpy(name, t[idx].name.c_str(), name_len - 1); name[name_len - 1] = 0; }
  if (numel) *numel = t[idx].numel;
  if (ndim) *ndim = t[idx].ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = t[idx].shape[i];
  return MANSY_OK;
}
size_t mansy_ppo_workspace_bytes(int max_batch) { PWork W; return max_batch >= 1 ? ppo_layout(max_batch, nullptr, W) : 0; }

// logits [B,16] (15 used), value [B] (nullable => actor only), optional sampling (act/logp; u nullable => hash RNG)
int mansy_policy_forward(const float* const* params, const float* obs, int B, float* logits, float* value, int* act, float* logp,
                         const float* u, uint32_t seed, uint32_t site, int reuse_packed, void* workspace, int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && obs && B >= 1 && B <= max_batch, "policy_forward: bad arguments (B=%d, max_batch=%d)", B, max_batch);
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP a, c; bind_net(params, nullptr, 20, a); bind_net(params, nullptr, 24, c);
  if (!reuse_packed) RC(e.pack(a, 0));      // rollouts: the block-diagonal image of the (unchanged) parameters is packed once per collect
  RC(e.featnet(obs, B, 0));
  RC(e.head(a, B, NACT, 0, e.W.A1a, e.W.Ha, logits ? logits : e.W.outa, u, seed, site, act, logp));
  if (value) RC(e.head(c, B, 1, 0, e.W.A1c, e.W.Hc, value, nullptr, 0, 0, nullptr, nullptr, nullptr, 1));     // written as a [B] vector
  return MANSY_OK;
}

// Rollout step as one call: policy forward + Categorical sample for the B = n_env observation rows, then (inside the same
// output-layer launch) the environment step of every environment with the action just drawn.
int mansy_policy_env_step(const float* const* params, const float* obs, int n_env, float* logits, int* act, float* logp, const float* u,
                          uint32_t seed, uint32_t site, int reuse_packed, void* workspace, int max_batch, const mansy_env_tables* T, void* env_state,
                          float* obs_next, float* obs_cur, float* reward, unsigned char* done, float* qoe_parts, const mansy_env_episode_log* elog,
                          int precision, void* stream) {
  MANSY_REQUIRE(params && obs && act && n_env >= 1 && n_env <= max_batch, "policy_env_step: bad arguments (n_env=%d, max_batch=%d)", n_env, max_batch);
  MANSY_REQUIRE(T && env_state && obs_next && reward && done, "policy_env_step: null environment pointer");
  MANSY_REQUIRE(T->size && T->quality && T->video_len && T->vp_gt && T->vp_pred && T->vp_acc && T->vp_start && T->vp_end && T->trace_bw &&
                    T->trace_len && T->samples && T->qoe_w && T->n_sample >= 1, "policy_env_step: incomplete tables");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP a; bind_net(params, nullptr, 20, a);
  if (!reuse_packed) RC(e.pack(a, 0));
  RC(e.featnet(obs, n_env, 0));
  EnvFuse ef; memset(&ef, 0, sizeof(ef));
  ef.on = 1; ef.T = *T; ef.st = (envdev::EnvState*)env_state; ef.obs_next = obs_next; ef.obs_cur = obs_cur; ef.reward = reward; ef.done = done;
  ef.qoe_parts = qoe_parts;
  if (elog) ef.elog = *elog;
  return e.head(a, n_env, NACT, 0, e.W.A1a, e.W.Ha, logits ? logits : e.W.outa, u, seed, site, act, logp, &ef);
}

// A whole collect -- T vector steps of policy forward + Categorical sample + environment step for n_env environments -- as ONE persistent launch on XCD
// teams (rollout_team_kernel above).  obs_slab [T][n_env][OBS_LD]: row block 0 holds the observations the collect starts from (the caller copied its
// carry there), block t + 1 receives step t's auto-reset observation, `carry` the last one; obs_next_slab / reward / done / act / logp as the per-step
// call writes them.  Same values, bit for bit, as T calls of mansy_policy_env_step.  MANSY_EINVAL (message: "rollout_team: ...") where the form does not
// apply -- another precision than fp32, a batch whose two products do not resolve to the wave-split-K loop, the launch recorder running -- so that the
// caller can take the per-step path.
int mansy_policy_rollout(const float* const* params, float* obs_slab, int n_env, int T, const float* u, int* act, float* logp, float* obs_next_slab,
                         float* carry, float* reward, unsigned char* done, float* qoe_parts, const mansy_env_tables* Tb, void* env_state,
                         const mansy_env_episode_log* elog, int reuse_packed, void* ctl, int* err_host, void* workspace, int max_batch, int precision,
                         void* stream) {
  MANSY_REQUIRE(params && obs_slab && u && act && logp && obs_next_slab && carry && reward && done && Tb && env_state && ctl && err_host && n_env >= 1 &&
                    n_env <= max_batch && T >= 1, "policy_rollout: bad arguments");
  MANSY_REQUIRE(precision == 0, "rollout_team: fp32 only (precision %d)", precision);
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP a; bind_net(params, nullptr, 20, a);
  const int req = head_split_request(n_env);
  MANSY_REQUIRE(req > 1, "rollout_team: the heads' fc product is not K-split at this batch");
  const int nsplit = mansy_gemm_effective_splits(FEAT, req);
  MANSY_REQUIRE(nsplit <= MAX_SLABS, "rollout_team: %d K splits exceed the slab sum's unroll", nsplit);
  if (!reuse_packed) RC(e.pack(a, 0));
  MANSY_REQUIRE(mansy_gemm_capture_begin(), "rollout_team: products cannot be captured now (launch recorder or a paired launch open)");
  int rc = e.featnet(obs_slab, n_env, 0);
  if (!rc) { GemmEpilogue ep; ep.prec = 0; ep.split_slab = (long long)n_env * HID; rc = mansy_launch_gemm_f32(e.W.F, FEAT, 0, a.fc_w, FEAT, 0, e.W.A1s, HID, n_env, HID, FEAT, ep, 0, req, e.st); }
  GemmParams gp[2]; int var[2], gx[2], gy[2], gz[2];
  const int ncap = mansy_gemm_capture_end(gp, var, gx, gy, gz, 2);
  RC(rc);
  MANSY_REQUIRE(ncap == 2 && var[0] == 0 && var[1] == 0 && gz[0] == 1 && gz[1] == nsplit && gx[0] == FEAT / 32 && gx[1] == HID / 32,
                "rollout_team: the two products of a step do not both resolve to the wave-split-K loop at n_env = %d", n_env);
  RolloutArgs ra;
  ra.p1 = gp[0]; ra.p2 = gp[1]; ra.g1x = gx[0]; ra.g2x = gx[1]; ra.g2z = gz[1];
  ra.fc_b = a.fc_b; ra.Wout = a.out_w; ra.bout = a.out_b;
  ra.F = e.W.F; ra.A1s = e.W.A1s; ra.slab = (long long)n_env * HID; ra.nsplit = nsplit;
  ra.n_env = n_env; ra.T = T;
  ra.obs = obs_slab; ra.obs_ts = (long long)n_env * OBS_LD; ra.obs_next = obs_next_slab; ra.obsn_ts = (long long)n_env * OBS_LD; ra.carry = carry;
  ra.u = u; ra.act = act; ra.logp = logp; ra.rew = reward; ra.done = done;
  ra.Tb = *Tb; ra.st = (envdev::EnvState*)env_state; ra.qoe_parts = qoe_parts;
  memset(&ra.elog, 0, sizeof(ra.elog));
  if (elog) ra.elog = *elog;
  // per DEVICE, not per process (a process that drives a second device or another SKU must not inherit the first one's CU count / clock rate)
  int wall_khz = 100000, n_cu = 256;
  {
    static thread_local int cached_dev = -1, cached_khz = 0, cached_cu = 0;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess) {
      if (dev != cached_dev) {
        int khz = 0, n = 0;
        cached_khz = (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0) ? khz : 100000;
        cached_cu = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        cached_dev = dev;
      }
      wall_khz = cached_khz; n_cu = cached_cu;
    }
  }
  ra.timeout_ticks = 2000LL * wall_khz;          // 2 s: a workgroup that never became resident (the device is shared) gives up loudly
  ra.err_host = err_host;
  MANSY_HIP_CHECK(hipMemsetAsync(ctl, 0, MANSY_ROLLOUT_CTL_BYTES, e.st));
  // one workgroup per CU (241 VGPRs + 16 AGPRs: a second one per CU does not fit; forced to fit it spills and the step takes 43 us: profiles/r05_rollout_team_ab.txt)
  MANSY_LAUNCH(rollout_team_kernel, dim3(n_cu < 256 ? n_cu : 256), dim3(256), 0, e.st, ra, (RolloutCtl*)ctl);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_identifier_forward(const float* const* params, const float* obs, int B, float* pred, void* workspace, int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && obs && pred && B >= 1 && B <= max_batch, "identifier_forward: bad arguments");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP n; bind_net(params, nullptr, 20, n);
  RC(e.pack(n, 1));
  RC(e.featnet(obs, B, 1));
  return e.head(n, B, 3, 1, e.W.A1a, e.W.Ha, pred, nullptr, 0, 0, nullptr, nullptr);
}

// one full-batch step of train_identifier (mansy_utils.py:20-31): MSE fwd + bwd + Adam(L2).
// step == 0: loss only (validation); step < 0: loss + gradients into flat_g, no optimiser step.
// idx != NULL: the B rows are obs[idx[0..B)] (train_identifier's shuffled 80 / 20 split: gathered by a rider of the pack launch, like
// the PPO minibatch step gathers its rows -- no separate gather launch, no shuffled copy of the buffer)
int mansy_identifier_train_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_m, float* flat_v,
                                long long n_flat, const float* obs_all, const int* idx, int B, float lr, float weight_decay, int step, float* loss_out,
                                void* workspace, int max_batch, void* xg_ctx, const float* adam_bias, int precision, void* stream) {
  MANSY_REQUIRE(params && obs_all && loss_out && B >= 1 && B <= max_batch, "identifier_train_step: bad arguments");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  const bool train = step != 0;
  MANSY_REQUIRE(!train || (grads && flat_p && flat_g && flat_m && flat_v), "identifier_train_step: null optimiser buffers");
  // xg_ctx != NULL (step > 0): the data-parallel step as one call -- gradients into the exchange slot, one launch averages them into flat_g, Adam
  std::vector<float*> slot_grads;
  float* const g_avg = flat_g;
  const int sync_kind = xg_ctx ? *reinterpret_cast<const int*>(xg_ctx) : 0;
  MANSY_REQUIRE(!xg_ctx || sync_kind == MANSY_SYNC_XG || sync_kind == MANSY_SYNC_RCCL, "identifier_train_step: the sync context is neither a peer-memory nor a communicator context");
  void* const comm = sync_kind == MANSY_SYNC_RCCL ? xg_ctx : nullptr;
  if (comm) { MANSY_REQUIRE(step > 0, "identifier_train_step: the averaged form is a training step (step > 0)"); xg_ctx = nullptr; }
  if (xg_ctx) {
    MANSY_REQUIRE(step > 0, "identifier_train_step: the peer-averaged form is a training step (step > 0)");
    float* s0 = nullptr; float* s1 = nullptr;
    RC(mansy_xg_slot_ptrs(xg_ctx, &s0, &s1));
    const int cur = mansy_xg_next_slot(xg_ctx);
    MANSY_REQUIRE(cur == 0 || cur == 1, "identifier_train_step: bad exchange context");
    float* slot = cur ? s1 : s0;
    const int np = mansy_net_num_params(1);
    slot_grads.resize(np);
    for (int k = 0; k < np; ++k) {
      MANSY_REQUIRE(grads[k] >= flat_g && grads[k] < flat_g + n_flat, "identifier_train_step: grads[] must point into flat_g");
      slot_grads[k] = slot + (grads[k] - flat_g);
    }
    grads = slot_grads.data(); flat_g = slot;
  }
  NetP n; bind_net(params, grads, 20, n);
  RC(e.pack(n, 1, nullptr, idx ? obs_all : nullptr, idx, B, train ? flat_g : nullptr, n_flat));      // row gather + gradient zero-fill ride on the pack launch
  const float* obs = idx ? e.W.obs_mb : obs_all;
  RC(e.featnet(obs, B, 1));
  if (!train) {          // validation: loss only (W.acc[0..1], the accumulator and the arrival counter, were zeroed by the pack launch's riders)
    RC(e.head(n, B, 3, 1, e.W.A1a, e.W.Ha, e.W.outa, nullptr, 0, 0, nullptr, nullptr));
    MANSY_LAUNCH(ident_mse_kernel, dim3(min(mansy_ceil_div(B * 3, 256), 256)), dim3(256), 0, e.st, e.W.outa, obs, B, nullptr, e.W.acc, loss_out);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  // training (step < 0: gradients only -- data-parallel callers all-reduce, then mansy_clip_grad_adam): the MSE, its gradient and the
  // output layer's input-side backward ride on the output-layer launch, the output layer's weight gradients and the loss on the unpack
  // launch (round 3: ident_mse_kernel and head_out_bwd_kernel are gone from this path, as from the PPO step)
  LossFuse lf; memset(&lf, 0, sizeof(lf));
  lf.on = 2; lf.n = B; lf.mse_obs = obs; lf.dlogits = e.W.gout; lf.lossrows = e.W.lossrows;
  lf.bwd_dH[0] = e.W.dHa; lf.bwd_dH[1] = e.W.dHa; lf.bwd_dA1 = e.W.dA1a; lf.bwd_dA1_ld = HID;
  RC(e.head(n, B, 3, 1, e.W.A1a, e.W.Ha, e.W.outa, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, nullptr, &lf));
  RC(e.fc_bwd_single(n, B, e.W.dA1a, e.W.dHa));
  OutGradRider og; memset(&og, 0, sizeof(og));
  og.on = 1; og.rows = B; og.g[0] = e.W.gout; og.g[1] = e.W.gout; og.H[0] = e.W.Ha; og.H[1] = e.W.Ha; og.n_out[0] = 3; og.n_out[1] = 0;
  og.gW[0] = n.gout_w; og.gb[0] = n.gout_b; og.gW[1] = n.gout_w; og.gb[1] = n.gout_b;
  og.lossrows = e.W.lossrows; og.stats = loss_out; og.mse = 1;
  RC(e.featnet_bwd(n, obs, B, 1, e.W.dHa, nullptr, nullptr, 0, nullptr, &og));
  if (step < 0) return MANSY_OK;
  if (xg_ctx) RC(mansy_xg_reduce_avg(xg_ctx, g_avg, n_flat, nullptr, stream));
  if (comm) RC(mansy_allreduce_avg_f32(comm, g_avg, n_flat, stream));
  return e.clip_and_adam(flat_p, g_avg, flat_m, flat_v, n_flat, 0.f, lr, weight_decay, step, -1, 0, false, adam_bias);
}

int mansy_identifier_relabel(const float* const* params, const float* obs, float* rew, float* id_rew, int B, float lamb, void* workspace,
                             int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && obs && rew && B >= 1 && B <= max_batch, "identifier_relabel: bad arguments");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP n; bind_net(params, nullptr, 20, n);
  RC(e.pack(n, 1));
  RC(e.featnet(obs, B, 1));
  PEng::HeadRiders rd; rd.relabel_rew = rew; rd.relabel_idrew = id_rew; rd.relabel_obs = obs; rd.relabel_lamb = lamb;      // the relabel rides on the output-layer launch
  RC(e.head(n, B, 3, 1, e.W.A1a, e.W.Ha, e.W.outa, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, &rd));
  return MANSY_OK;
}

// log-prob of given actions under logits [B,16]
// n_logp: log-probabilities are wanted for the first n_logp rows only (process_fn evaluates obs and obs_next -- 2 x 4096 rows of
// one buffer -- in ONE pass: values for all rows, log-probabilities of the taken actions for the obs half)
int mansy_policy_evaluate(const float* const* params, const float* obs, int B, const int* act, int n_logp, float* logp, float* value, void* workspace,
                          int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && obs && B >= 1 && B <= max_batch, "policy_evaluate: bad arguments");
  MANSY_REQUIRE(!logp || (n_logp >= 1 && n_logp <= B), "policy_evaluate: n_logp must be in [1, B] when log-probabilities are wanted");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP a, c; bind_net(params, nullptr, 20, a); bind_net(params, nullptr, 24, c);
  const bool both = logp && value;
  RC(e.pack(a, 0, both ? &c : nullptr));
  RC(e.featnet(obs, B, 0));
  if (logp) MANSY_REQUIRE(act, "policy_evaluate: logp needs actions");
  PEng::HeadRiders rd; rd.act_given = logp ? act : nullptr; rd.n_given = logp ? n_logp : 0; rd.logp = logp;      // logp_old rides on the output-layer launch
  if (both) RC(e.head_pair(a, c, B, nullptr, value, &rd));      // actor + critic in one stacked product; the value lands in `value`
  else if (logp) RC(e.head(a, B, NACT, 0, e.W.A1a, e.W.Ha, e.W.outa, nullptr, 0, 0, nullptr, nullptr, nullptr, 0, &rd));
  if (value && !both) RC(e.head(c, B, 1, 0, e.W.A1c, e.W.Hc, value, nullptr, 0, 0, nullptr, nullptr, nullptr, 1));
  return MANSY_OK;
}

// rew / v_s / v_next / done: [T][N] step-major (T steps of N environments).  rms = device doubles [mean, var, count]
// (tianshou RunningMeanStd).  scratch: T*N + 2 doubles.  T2 semantics: critic outputs are un-normalised by sqrt(var+eps)
// before the scan, returns are re-normalised with the OLD variance, then the statistics absorb the un-normalised returns.
int mansy_gae_returns(const float* rew, const float* v_s, const float* v_next, const unsigned char* done, int T, int N, double gamma, double gae_lambda,
                      int rew_norm, double* rms, double* scratch, float* returns, float* adv, void* stream) {
  MANSY_REQUIRE(rew && v_s && v_next && done && rms && scratch && returns && adv && T >= 1 && N >= 1, "gae_returns: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const long long n = (long long)T * N;
  double* part = scratch + n;
  if (N <= 1024) {
    MANSY_LAUNCH(gae_fused_kernel, dim3(1), dim3(1024), 0, st, rew, v_s, v_next, done, T, N, (double)gamma, (double)gae_lambda, rms, rew_norm, 1e-8,
                       scratch, adv, returns);
    MANSY_LAUNCH_CHECK();
    return MANSY_OK;
  }
  MANSY_LAUNCH(gae_kernel, dim3(mansy_ceil_div(N, 256)), dim3(256), 0, st, rew, v_s, v_next, done, T, N, (double)gamma, (double)gae_lambda,
                     rms, rew_norm, 1e-8, scratch, adv);
  MANSY_LAUNCH(ret_stats_kernel, dim3(1), dim3(1024), 0, st, scratch, n, part);
  MANSY_LAUNCH(ret_finish_kernel, dim3(mansy_ceil_div(n, 256)), dim3(256), 0, st, scratch, n, rms, rew_norm, 1e-8, returns);
  if (rew_norm) MANSY_LAUNCH(rms_merge_kernel, dim3(1), dim3(1), 0, st, rms, part, n);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

// One PPO minibatch update (T2: PPOPolicy.learn body): gather rows idx[0..mb) of the collected buffer, forward shared
// FeatureNet + both heads, clipped loss / value loss / entropy, backward, global grad-norm clip, Adam(L2).
int mansy_ppo_minibatch_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_m, float* flat_v,
                             long long n_flat, const float* obs_all, const int* idx, const int* act_all, const float* adv_all,
                             const float* logp_old_all, const float* v_old_all, const float* ret_all, int mb, float eps_clip, float vf_coef,
                             float ent_coef, int norm_adv, int value_clip, float dual_clip, float max_grad_norm, float lr, float weight_decay, int step,
                             long long tail_from, int tail_step, float* stats, void* workspace, int max_batch, int chain_in,
                             const int* next_idx, int next_mb, void* xg_ctx, const float* adam_bias, int precision, void* stream) {
  MANSY_REQUIRE(params && grads && flat_p && flat_g && flat_m && flat_v && obs_all && act_all && adv_all && logp_old_all && v_old_all && ret_all,
                "ppo_minibatch_step: null pointer");
  // xg_ctx != NULL (round 5): the data-parallel step as ONE call -- this rank's raw gradients are produced straight in its exchange slot
  // (csrc/xgmi.hip), one launch publishes it / waits for the peers / sums all ranks' slots in rank order into flat_g (+ the partial sums of
  // squares), then the chained tail (clip + Adam + re-pack + next prologue) runs on the average and zeroes the OTHER slot for the next step.
  // Same launches as the single-process step with one more (the average) and without the norm rider; the host does nothing per step that the
  // single-process form does not do.  Needs the chained form: step > 0, max_grad_norm > 0, no lagged tail.
  float* slot_g[2] = {nullptr, nullptr};
  int slot_cur = 0;
  std::vector<float*> slot_grads;
  // a communicator context instead (MANSY_SYNC_RCCL): raw gradients in flat_g, ncclAllReduce(avg) on the stream, a norm launch, the same tail
  void* const sync = xg_ctx;
  const int sync_kind = sync ? *reinterpret_cast<const int*>(sync) : 0;
  MANSY_REQUIRE(!sync || sync_kind == MANSY_SYNC_XG || sync_kind == MANSY_SYNC_RCCL, "ppo_minibatch_step: the sync context is neither a peer-memory nor a communicator context");
  void* const comm = sync_kind == MANSY_SYNC_RCCL ? sync : nullptr;
  if (comm) {
    MANSY_REQUIRE(max_grad_norm > 0.f && step > 0 && !(tail_from >= 0 && tail_from < n_flat && tail_step != step), "ppo_minibatch_step: the averaged form needs the clipped step without a lagged tail");
    xg_ctx = nullptr;
  }
  if (xg_ctx) {
    MANSY_REQUIRE(max_grad_norm > 0.f && step > 0 && !(tail_from >= 0 && tail_from < n_flat && tail_step != step), "ppo_minibatch_step: the peer-averaged form needs the clipped step without a lagged tail");
    RC(mansy_xg_slot_ptrs(xg_ctx, &slot_g[0], &slot_g[1]));
    slot_cur = mansy_xg_next_slot(xg_ctx);
    MANSY_REQUIRE(slot_cur == 0 || slot_cur == 1, "ppo_minibatch_step: bad exchange context");
    const int np = mansy_net_num_params(0);
    slot_grads.resize(np);
    for (int k = 0; k < np; ++k) {
      MANSY_REQUIRE(grads[k] >= flat_g && grads[k] < flat_g + n_flat, "ppo_minibatch_step: grads[] must point into flat_g");
      slot_grads[k] = slot_g[slot_cur] + (grads[k] - flat_g);
    }
  }
  float* const g_avg = flat_g;                               // where the (averaged) gradient ends up for the tail
  if (xg_ctx) { grads = slot_grads.data(); flat_g = slot_g[slot_cur]; }
  MANSY_REQUIRE(mb >= 2 && mb <= max_batch, "ppo_minibatch_step: bad minibatch size");
  MANSY_REQUIRE(dual_clip == 0.f || dual_clip > 1.f, "ppo_minibatch_step: dual_clip must be 0 (off) or > 1 (tianshou asserts the same)");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP a, c; bind_net(params, grads, 20, a); bind_net(params, grads, 24, c);
  // one prologue launch: re-pack the block-diagonal / stacked weights, gather the minibatch rows, zero the gradient buffer
  const float* obs = idx ? e.W.obs_mb : obs_all;
  // loss fused into the output-layer launches: advantage statistics ride on the prologue, per-row terms come out of head_out,
  // their sums out of head_out_bwd (13 -> 12 launches per minibatch step)
  // chain_in: the previous call (same workspace, stream, parameters) already did all of that for exactly this minibatch in its
  // last launch (step_tail) -- 16 minibatch steps of an update are then 1 + 16 x 9 launches instead of 16 x 10
  MANSY_REQUIRE(next_mb >= 0 && next_mb <= max_batch && (next_mb == 0 || !idx == !next_idx), "ppo_minibatch_step: bad next minibatch");
  const bool lagged = tail_from >= 0 && tail_from < n_flat && tail_step != step;
  const bool chain_ok = max_grad_norm > 0.f && step > 0 && !lagged;       // the step that ends in step_tail
  // (step == 0 -- gradients only, the data-parallel form -- may be chained INTO: mansy_ppo_dp_tail, called after the gradient
  // average, is then the launch that prepared this minibatch)
  MANSY_REQUIRE(next_mb == 0 || chain_ok, "ppo_minibatch_step: preparing the next minibatch needs the clipped single-process step (no lagged tail)");
  MANSY_REQUIRE(!chain_in || chain_ok || step == 0, "ppo_minibatch_step: chain_in needs the clipped step or the gradients-only form");
  double* const parts_cur = e.W.acc + (chain_ok && (step & 1) ? NORM_PARTS_C : 0);
  double* const parts_next = e.W.acc + (chain_ok && (step & 1) ? 0 : NORM_PARTS_C);
  if (!chain_in) RC(e.pack_mb(a, c, idx ? obs_all : nullptr, idx, mb, flat_g, n_flat, adv_all));
  RC(e.featnet(obs, mb, 0));
  LossFuse lf; memset(&lf, 0, sizeof(lf));
  lf.on = 1; lf.act = act_all; lf.adv = adv_all; lf.logp_old = logp_old_all; lf.v_old = v_old_all; lf.ret = ret_all; lf.idx = idx; lf.n = mb;
  lf.eps_clip = eps_clip; lf.vf_coef = vf_coef; lf.ent_coef = ent_coef; lf.norm_adv = norm_adv; lf.value_clip = value_clip; lf.dual_clip = dual_clip;
  lf.adv_eps = 0.f;   // T2: tianshou 0.4.8 ppo.py divides by the bare unbiased std (`(adv - mean) / std  # per-batch norm`); `+ self._eps` is 0.5.0's
  lf.adv_stats = e.W.adv_stats; lf.dlogits = e.W.gout; lf.dvalue = e.W.gout_c; lf.dvalue_ld = MAXOUT; lf.lossrows = e.W.lossrows;
  // the output layers' backward rides on the launches around it (round 3: head_out_bwd_kernel's launch is gone from this step): the
  // input side (dH, dA1) in head_out_kernel, the weight gradients and the loss statistics in the unpack launch
  lf.bwd_dH[0] = e.W.dHa; lf.bwd_dH[1] = e.W.dHc; lf.bwd_dA1 = e.W.dA1p; lf.bwd_dA1_ld = 2 * HID;
  RC(e.head_pair(a, c, mb, &lf));
  RC(e.fc_bwd_pair(a, c, mb));
  OutGradRider og; memset(&og, 0, sizeof(og));
  og.on = 1; og.rows = mb; og.g[0] = e.W.gout; og.g[1] = e.W.gout_c; og.H[0] = e.W.Ha; og.H[1] = e.W.Hc; og.n_out[0] = NACT; og.n_out[1] = 1;
  og.gW[0] = a.gout_w; og.gb[0] = a.gout_b; og.gW[1] = c.gout_w; og.gb[1] = c.gout_b;
  og.lossrows = e.W.lossrows; og.vf_coef = vf_coef; og.ent_coef = ent_coef; og.stats = stats;
  og.skip0 = a.gout_w; og.skip0_n = c.gfc_w - a.gout_w; og.skip1 = c.gout_w; og.skip1_n = (flat_g + n_flat) - c.gout_w;
  MANSY_REQUIRE(og.skip0_n > 0 && og.skip0_n % 4 == 0 && og.skip1_n > 0 && a.gout_b > a.gout_w && a.gout_b < c.gfc_w && c.gout_b > c.gout_w,
                "ppo_minibatch_step: unexpected order of the head gradients in flat_g");
  // squared gradient norm as a rider on the last gradient-writing launch (12 -> 11 launches): the head gradients are the
  // contiguous tail of the flat buffer, starting at actor.fc.0.weight
  const bool ride = max_grad_norm > 0.f && step > 0 && !xg_ctx && !comm;      // (averaged forms: the norm is the AVERAGE's)
  const float* tail = grads[2 * NB];
  const long long tail_n = (flat_g + n_flat) - tail;
  MANSY_REQUIRE(!ride || (tail >= flat_g && tail_n > 0 && tail_n <= n_flat), "ppo_minibatch_step: grads[] must point into flat_g");
  // (Tried and dropped, round 2: unpack + norm + clip + Adam as ONE launch with a device-scope barrier between the norm and the
  // update -- 64..256 resident workgroups, two returning atomics per workgroup, gradients kept in registers across the barrier.
  // 22-32 us per launch against 12 us for the two launches it replaced: a kernel boundary is cheaper than a device-scope
  // rendezvous on this chip, as tools/chain_lab.hip found for the GEMM chain.)
  RC(e.featnet_bwd(a, obs, mb, 0, e.W.dHa, e.W.dHc, ride ? tail : nullptr, ride ? tail_n : 0, parts_cur, &og));
  if (comm) {
    RC(mansy_allreduce_avg_f32(comm, flat_g, n_flat, stream));
    MANSY_LAUNCH(sumsq_kernel, dim3(NORM_PARTS), dim3(256), 0, e.st, flat_g, n_flat, parts_cur);
    MANSY_LAUNCH_CHECK();
    return e.step_tail(params, flat_p, flat_g, flat_m, flat_v, n_flat, max_grad_norm, lr, weight_decay, step, parts_cur, parts_next, obs_all,
                       next_idx, next_mb, adv_all, nullptr, adam_bias);
  }
  if (xg_ctx) {
    RC(mansy_xg_reduce_avg(xg_ctx, g_avg, n_flat, parts_cur, stream));      // slot -> average in flat_g, sums of squares in this step's norm slots
    return e.step_tail(params, flat_p, g_avg, flat_m, flat_v, n_flat, max_grad_norm, lr, weight_decay, step, parts_cur, parts_next, obs_all,
                       next_idx, next_mb, adv_all, slot_g[slot_cur ^ 1], adam_bias);
  }
  if (chain_ok)
    return e.step_tail(params, flat_p, flat_g, flat_m, flat_v, n_flat, max_grad_norm, lr, weight_decay, step, parts_cur, parts_next, obs_all,
                       next_idx, next_mb, adv_all, nullptr, adam_bias);
  return e.clip_and_adam(flat_p, flat_g, flat_m, flat_v, n_flat, max_grad_norm, lr, weight_decay, step, tail_from, tail_step, ride, adam_bias);
}

// Behaviour-cloning step (utils/mansy_utils.py:52-69): loss = CrossEntropy(actor logits, expert action) - ent_coef * mean
// entropy; backward through the actor head and the shared feature net; Adam(L2) over the first n_update elements of the
// flat buffers only -- the critic head (the tail of the buffer) has no gradient there and torch.optim.Adam skips
// parameters whose .grad is None (no weight decay, no state).  step <= 0: forward + loss only (the validation pass,
// :71-78; stats[1] is the plain cross entropy).  stats: [loss, cross entropy, mean entropy].
int mansy_bc_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_m, float* flat_v,
                  long long n_flat, long long n_update, const float* obs, const int* act, int B, float ent_coef, float lr,
                  float weight_decay, int step, float* stats, void* workspace, int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && obs && act && stats && B >= 1 && B <= max_batch, "bc_step: bad arguments");
  MANSY_REQUIRE(step <= 0 || (grads && flat_p && flat_g && flat_m && flat_v && n_update >= 1 && n_update <= n_flat), "bc_step: bad buffers");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  NetP a, c; bind_net(params, grads, 20, a); bind_net(params, grads, 24, c);
  RC(e.pack(a, 0, &c, nullptr, nullptr, B, step > 0 ? flat_g : nullptr, step > 0 ? n_flat : 0));
  RC(e.featnet(obs, B, 0));
  RC(e.head_pair(a, c, B));
  MANSY_LAUNCH(bc_loss_kernel, dim3(1), dim3(1024), 0, e.st, e.W.outa, act, B, ent_coef, e.W.gout, e.W.gout_c, stats);
  MANSY_LAUNCH_CHECK();
  if (step <= 0) return MANSY_OK;
  RC(e.head_bwd_pair(a, c, B));
  RC(e.featnet_bwd(a, obs, B, 0, e.W.dHa, e.W.dHc));
  return e.clip_and_adam(flat_p, flat_g, flat_m, flat_v, n_update, 0.f, lr, weight_decay, step);
}

// Data-parallel counterpart of the chained step's last launch: the ranks ran mansy_ppo_minibatch_step with step = 0 (raw local
// gradients) and averaged flat_g; this call clips by the global norm (have_sumsq: `scratch` already holds the partial sums of squares
// -- mansy_xg_allreduce_avg leaves them), applies Adam(L2), zeroes flat_g, re-packs the updated parameters and, with next_mb > 0,
// gathers the next minibatch and takes its advantage statistics, so that the next mansy_ppo_minibatch_step passes chain_in = 1.
int mansy_ppo_dp_tail(const float* const* params, float* flat_p, float* flat_g, float* flat_m, float* flat_v, long long n_flat, float max_grad_norm,
                      float lr, float weight_decay, int step, double* scratch, int have_sumsq, const float* obs_all, const float* adv_all,
                      const int* next_idx, int next_mb, float* next_flat_g, const float* adam_bias, void* workspace, int max_batch, int precision, void* stream) {
  MANSY_REQUIRE(params && flat_p && flat_g && flat_m && flat_v && scratch && step >= 1 && max_grad_norm > 0.f, "ppo_dp_tail: bad arguments");
  MANSY_REQUIRE(next_mb >= 0 && next_mb <= max_batch && (next_mb == 0 || (obs_all && adv_all)), "ppo_dp_tail: bad next minibatch");
  PEng e; RC(setup(workspace, max_batch, precision, (hipStream_t)stream, e));
  if (!have_sumsq) { MANSY_LAUNCH(sumsq_kernel, dim3(NORM_PARTS), dim3(256), 0, e.st, flat_g, n_flat, scratch); MANSY_LAUNCH_CHECK(); }
  return e.step_tail(params, flat_p, flat_g, flat_m, flat_v, n_flat, max_grad_norm, lr, weight_decay, step, scratch, e.W.acc + NORM_PARTS_C, obs_all,
                     next_idx, next_mb, adv_all, next_flat_g, adam_bias);
}

// Global-norm clip (torch clip_grad_norm_ semantics; max_norm <= 0 disables) followed by Adam with L2 weight decay over
// flat buffers.  Data-parallel callers run the minibatch step with step = 0 and max_grad_norm = 0 (raw gradients),
// all-reduce flat_g over RCCL, then call this.  scratch: MANSY_CLIP_SCRATCH_DOUBLES doubles.
int mansy_clip_grad_adam(float* flat_p, float* flat_g, float* flat_m, float* flat_v, long long n_flat, float max_grad_norm, float lr,
                         float weight_decay, int step, long long tail_from, int tail_step, double* scratch, int have_sumsq, const float* adam_bias,
                         void* stream) {
  MANSY_REQUIRE(flat_p && flat_g && flat_m && flat_v && scratch && step >= 1, "clip_grad_adam: bad arguments");
  PEng e; e.st = (hipStream_t)stream; e.W.acc = scratch;
  return e.clip_and_adam(flat_p, flat_g, flat_m, flat_v, n_flat, max_grad_norm, lr, weight_decay, step, tail_from, tail_step, have_sumsq != 0, adam_bias);
}

}  // extern "C"
