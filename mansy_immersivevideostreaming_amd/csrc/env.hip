// Device-resident vectorised streaming environment: one 64-lane wavefront per environment, lane = tile
// (the reference's 8x8 tiling is exactly one wave).  Reference semantics:
//   MANSYEnv.reset / step                bitrate_selection/envs/mansy_env.py:99-248
//   action2rates, allocate_tile_rates    bitrate_selection/utils/common.py:101-119, 142-193
//   Simulator / NetworkTrace / PlaybackBuffer / HMDTrace  bitrate_selection/simulators/*.py
//   QoEModel.calculate_qoe               bitrate_selection/utils/qoe.py:22-34
// Bit-exactness contract (checked against oracle/env.c, itself pinned on reference trajectories):
//   * ring index of the pyramid allocation = 8-neighbour BFS depth on the torus == toroidal Chebyshev distance,
//     computed by <= 4 bit-parallel dilations of the 64-bit predicted-viewport mask (rows are bytes);
//   * chunk size = exact integer wave reduction; download time / buffer / rebuffer in IEEE double with FMA
//     contraction OFF (Python floats); QoE sums are SEQUENTIAL float32 adds in tile order (Python sum() over a
//     float32 array), done with v_readlane in a fixed order -- never a tree reduction.
// Tables (manifests, viewport maps, traces) stay resident in HBM; an env step touches 2 x 1280 B of manifest rows,
// 2 x 64 B of viewport maps, a few trace bins and writes one 3 120-byte observation row (coalesced 256-B pieces).
#include "mansy_kernels.h"
#include "../../include/mansy_hip.h"

#pragma clang fp contract(off)

namespace {

#include "env_device.h"

__global__ __launch_bounds__(256) void env_init_kernel(EnvState* st, int n_env, int index_offset, int worker_num, int seed) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_env) return;
  EnvRegs s = {};
  s.worker_num = worker_num;
  s.worker_id = (seed + index_offset + e) % worker_num;      // tianshou venv.seed(seed): env i gets seed + i (mansy_env.py:253-256)
  copy_scalars(st[e], s);
#pragma unroll
  for (int k = 0; k < PAST_K; ++k) store_state(st[e], s, k);  // (lane 0 repeats the scalars; the rings get their eight zeros)
}

__global__ __launch_bounds__(256) void env_reset_kernel(mansy_env_tables T, EnvState* st, int n_env, float* obs) {
  const int e = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (e >= n_env) return;
  EnvRegs s;
  load_state(s, st[e], lane);            // every lane holds the (uniform) scalars, lanes 0..7 the ring elements
  do_reset(T, s);
  write_obs(T, s, s.next_chunk, -1, lane, obs + (size_t)e * OBS_LD);
  store_state(st[e], s, lane);
}

__global__ __launch_bounds__(256) void env_step_kernel(mansy_env_tables T, EnvState* st, int n_env, const int* __restrict__ actions,
                                                       float* obs_next, float* obs_cur, float* reward, unsigned char* done,
                                                       float* qoe_parts, mansy_env_episode_log elog) {
  const int e = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;   // wave-uniform
  if (e >= n_env) return;
  env_step_wave(T, st, e, lane, actions[e], obs_next, obs_cur, reward, done, qoe_parts, elog);
}

__global__ __launch_bounds__(256) void alloc_rates_kernel(const float* __restrict__ pred_vp, const int* __restrict__ actions, int n,
                                                          mansy_env_tables T, int* __restrict__ versions) {
  const int e = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (e >= n) return;
  const int action = actions[e];
  const int rin = (action >= 0 && action < N_ACTION) ? A2R[action][0] : 0;
  const int rout = (action >= 0 && action < N_ACTION) ? A2R[action][1] : 0;
  unsigned long long m = __ballot(pred_vp[(size_t)e * NTL + lane] == 1.0f);
  int dist = ((m >> lane) & 1ull) ? 0 : -1;
  if (m == 0ull) dist = 0;
  else {
#pragma unroll
    for (int sidx = 1; sidx <= 4; ++sidx) { m = dilate8(m); if (dist < 0 && ((m >> lane) & 1ull)) dist = sidx; }
  }
  const Rates rates = load_rates(T);
  versions[(size_t)e * NTL + lane] = dist == 0 ? rin : closest_rate_version(rates, pick_rate(rates, rout) / (dist > 0 ? dist : 1));
}


// ---- MPC expert (bitrate_selection/envs/expert_env.py) ------------------------------------------------------------
// Profile cache (expert_env.py:126-181): one wave per (viewport trace, chunk, action, gt|pred map).  The tile-rate
// allocation runs on the chosen map, size and quality are gathered for the 64 tiles, and quality / intra-viewport
// variance are the reference's sequential float32 sums over the GROUND-TRUTH viewport (simulator.py:146-158).
__global__ __launch_bounds__(256) void expert_profile_kernel(mansy_env_tables T, const int* __restrict__ vp_video, int n_vp,
                                                             float* gt_quality, float* pred_quality, float* gt_var, float* pred_var,
                                                             int* gt_size, int* pred_size) {
  const long long w = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long long total = (long long)n_vp * T.n_vpchunk_max * N_ACTION * 2;
  if (w >= total) return;
  const int which = (int)(w & 1);
  const int action = (int)((w >> 1) % N_ACTION);
  const long long cell = (w >> 1) / N_ACTION;            // vp * n_vpchunk_max + j
  const int vp = (int)(cell / T.n_vpchunk_max), j = (int)(cell % T.n_vpchunk_max);
  const int video = vp_video[vp], chunk = T.vp_start[vp] + j;
  const int vlen1 = T.video_len[video] - 1;
  const int end_chunk = T.vp_end[vp] < vlen1 ? T.vp_end[vp] : vlen1;
  float* out_q = which ? pred_quality : gt_quality;
  float* out_v = which ? pred_var : gt_var;
  int* out_s = which ? pred_size : gt_size;
  const size_t o = (size_t)cell * N_ACTION + action;
  if (chunk < T.startup_download + 1 || chunk > end_chunk) {       // never visited by an episode
    if (lane == 0) { out_q[o] = 0.f; out_v[o] = 0.f; out_s[o] = 0; }
    return;
  }
  const size_t vrow = (size_t)cell * NTL;
  const float gv = (float)T.vp_gt[vrow + lane];
  const bool in_map = (which ? T.vp_pred[vrow + lane] : T.vp_gt[vrow + lane]) == 1;
  unsigned long long m = __ballot(in_map);
  int dist = ((m >> lane) & 1ull) ? 0 : -1;
  if (m == 0ull) dist = 0;
  else {
#pragma unroll
    for (int sidx = 1; sidx <= 4; ++sidx) { m = dilate8(m); if (dist < 0 && ((m >> lane) & 1ull)) dist = sidx; }
  }
  const Rates rates = load_rates(T);
  const int rin = A2R[action][0], rout = A2R[action][1];
  const int ver = dist == 0 ? rin : closest_rate_version(rates, pick_rate(rates, rout) / (dist > 0 ? dist : 1));
  const size_t mrow = ((size_t)video * T.n_chunk_max + chunk) * NR * NTL;
  const int chunk_size = wave_isum(T.size[mrow + ver * NTL + lane]);
  const float tq = T.quality[mrow + ver * NTL + lane];
  const ViewportSums vs = viewport_sums(gv, tq);
  const float s_v = vs.s_v;
  const float vq = vs.s_vq / s_v;
  const float s_var = var_sum(vs, gv, tq, vq);
  if (lane == 0) { out_q[o] = vq; out_v[o] = s_var / s_v; out_s[o] = chunk_size; }
}

// Exhaustive look-ahead (expert_env.py:358-422).  Plan i = sum_t a_t * 15^t; its score is the float32 sum over
// t < horizon = min(H, chunks left) of the QoE of a virtual download of the cached chunk size (network.py:22-35 /
// buffer.py:8-15 in doubles, qoe.py:49-59 in float32).  The reference keeps the FIRST plan with the strictly largest
// score; plans that differ only in digits >= horizon tie, so the winner has i < 15^horizon.  One thread scores the 15
// plans that share a (horizon-1)-step prefix; (score, index) pairs are packed into an order-preserving 64-bit key
// (larger score wins, then the smaller index) and merged with atomicMax.
__device__ __forceinline__ unsigned long long expert_key(float v, unsigned idx) {
  if (v != v) return 0ull;                                         // NaN never beats the incumbent
  unsigned u = __float_as_uint(v == 0.f ? 0.f : v);                // -0.0 == 0.0 in the reference's comparison
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - idx);
}

struct PlanState { double cur_time, buf; int cur_idx, has_prev; float prev, sum; };

__device__ __forceinline__ void plan_step(PlanState& ps, const double* __restrict__ bw, int tlen, double chunk_length, float w0, float w1,
                                          float w2, float max_rate, float q, float var, int size_i) {
  double size = (double)size_i;
  const double start = ps.cur_time;
  while (size > 0) {
    const double fl = floor(ps.cur_time + 1);
    const double remain = (fl - ps.cur_time) * bw[ps.cur_idx];
    if (size >= remain) { ps.cur_idx = ps.cur_idx + 1 == tlen ? 0 : ps.cur_idx + 1; ps.cur_time = fl; size -= remain; }
    else { ps.cur_time += size / bw[ps.cur_idx]; size = 0; }
  }
  const double download_time = ps.cur_time - start;
  double rebuf = 0.0;
  if (download_time > ps.buf) { rebuf = download_time - ps.buf; ps.buf = chunk_length; }
  else ps.buf = ps.buf - download_time + chunk_length;
  const float vq = q / max_rate, intra = var / max_rate;
  const float inter = ps.has_prev ? fabsf(vq - ps.prev) : 0.f;
  ps.prev = vq; ps.has_prev = 1;
  const float qoe3 = intra + inter;
  ps.sum = ps.sum + (w0 * vq - w1 * (float)rebuf - w2 * qoe3);
}

__global__ __launch_bounds__(256) void expert_keys_init_kernel(unsigned long long* keys, int n_env) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n_env) keys[e] = expert_key(-INFINITY, 0u);
}

__global__ __launch_bounds__(256) void expert_search_kernel(mansy_env_tables T, const EnvState* __restrict__ st, int horizon_cfg,
                                                            const float* __restrict__ pred_quality, const float* __restrict__ pred_var,
                                                            const int* __restrict__ pred_size, unsigned long long* keys) {
  const int e = blockIdx.y;
  const EnvState& s = st[e];
  int horizon = s.end_chunk - s.next_chunk + 1;
  horizon = horizon < horizon_cfg ? horizon : horizon_cfg;
  const unsigned p = blockIdx.x * 256 + threadIdx.x;
  unsigned long long key = 0ull;
  if (horizon <= 0) {                                               // nothing left to download: every plan scores 0
    if (p == 0) key = expert_key(0.f, 0u);
  } else {
    unsigned n_prefix = 1;
    for (int t = 0; t < horizon - 1; ++t) n_prefix *= N_ACTION;
    if (p < n_prefix) {
      const double* bw = T.trace_bw + (size_t)s.trace * T.trace_len_max;
      const int tlen = T.trace_len[s.trace];
      const float* w = T.qoe_w + 3 * s.qoe;
      const float w0 = w[0], w1 = w[1], w2 = w[2], max_rate = (float)T.video_rates[NR - 1];
      const double chunk_length = (double)T.chunk_length;
      const size_t row0 = ((size_t)s.vp * T.n_vpchunk_max + (s.next_chunk - T.vp_start[s.vp])) * N_ACTION;
      PlanState ps = {s.cur_time, s.buf_size, s.cur_idx, s.has_prev, s.prev_vq, 0.f};
      unsigned tmp = p;
      for (int t = 0; t < horizon - 1; ++t) {
        const size_t k = row0 + (size_t)t * N_ACTION + tmp % N_ACTION;
        tmp /= N_ACTION;
        plan_step(ps, bw, tlen, chunk_length, w0, w1, w2, max_rate, pred_quality[k], pred_var[k], pred_size[k]);
      }
      const size_t klast = row0 + (size_t)(horizon - 1) * N_ACTION;
      for (int a = 0; a < N_ACTION; ++a) {
        PlanState q = ps;
        plan_step(q, bw, tlen, chunk_length, w0, w1, w2, max_rate, pred_quality[klast + a], pred_var[klast + a], pred_size[klast + a]);
        const unsigned long long k2 = expert_key(q.sum, p + (unsigned)a * n_prefix);
        key = k2 > key ? k2 : key;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o, 64);
    key = other > key ? other : key;
  }
  if ((threadIdx.x & 63) == 0 && key != 0ull) atomicMax(keys + e, key);
}

__global__ __launch_bounds__(256) void expert_pick_kernel(const unsigned long long* __restrict__ keys, int n_env, int* actions,
                                                          float* best_value, long long* best_index) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n_env) return;
  const unsigned long long key = keys[e];
  const unsigned idx = 0xFFFFFFFFu - (unsigned)(key & 0xFFFFFFFFull);
  actions[e] = (int)(idx % N_ACTION);         // rates2action(action2rates(a)) == a for all 15 actions
  if (best_value) {
    unsigned u = (unsigned)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
    best_value[e] = __uint_as_float(u);
  }
  if (best_index) best_index[e] = (long long)idx;
}

int check_tables(const mansy_env_tables* T) {
  MANSY_REQUIRE(T, "env: null tables");
  MANSY_REQUIRE(T->size && T->quality && T->video_len && T->vp_gt && T->vp_pred && T->vp_acc && T->vp_start && T->vp_end && T->trace_bw &&
                    T->trace_len && T->samples && T->qoe_w, "env: null table pointer");
  MANSY_REQUIRE(T->n_sample >= 1 && T->n_chunk_max >= 1 && T->n_vpchunk_max >= 1 && T->trace_len_max >= 1, "env: empty tables");
  return MANSY_OK;
}

}  // namespace

extern "C" {

int mansy_env_state_bytes(void) { return (int)sizeof(EnvState); }

int mansy_env_init(void* state, int n_env, int index_offset, int worker_num, int seed, void* stream) {
  MANSY_REQUIRE(state && n_env >= 1 && worker_num >= 1, "env_init: bad arguments");
  MANSY_LAUNCH(env_init_kernel, dim3(mansy_ceil_div(n_env, 256)), dim3(256), 0, (hipStream_t)stream, (EnvState*)state, n_env,
                     index_offset, worker_num, seed);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_env_reset(const mansy_env_tables* T, void* state, int n_env, float* obs, void* stream) {
  int rc = check_tables(T); if (rc) return rc;
  MANSY_REQUIRE(state && obs && n_env >= 1, "env_reset: bad arguments");
  MANSY_LAUNCH(env_reset_kernel, dim3(mansy_ceil_div((long long)n_env * 64, 256)), dim3(256), 0, (hipStream_t)stream, *T,
                     (EnvState*)state, n_env, obs);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_env_step(const mansy_env_tables* T, void* state, int n_env, const int* actions, float* obs_next, float* obs_cur, float* reward,
                   unsigned char* done, float* qoe_parts, const mansy_env_episode_log* elog, void* stream) {
  int rc = check_tables(T); if (rc) return rc;
  MANSY_REQUIRE(state && actions && obs_next && reward && done && n_env >= 1, "env_step: bad arguments");
  mansy_env_episode_log el = {nullptr, nullptr, 0};
  if (elog) el = *elog;
  MANSY_LAUNCH(env_step_kernel, dim3(mansy_ceil_div((long long)n_env * 64, 256)), dim3(256), 0, (hipStream_t)stream, *T,
                     (EnvState*)state, n_env, actions, obs_next, obs_cur, reward, done, qoe_parts, el);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_allocate_tile_rates(const float* pred_viewport, const int* actions, int n, const int video_rates[5], int* versions, void* stream) {
  MANSY_REQUIRE(pred_viewport && actions && versions && video_rates && n >= 0, "allocate_tile_rates: bad arguments");
  if (n == 0) return MANSY_OK;
  mansy_env_tables T;
  memset(&T, 0, sizeof(T));
  for (int i = 0; i < 5; ++i) T.video_rates[i] = video_rates[i];
  MANSY_LAUNCH(alloc_rates_kernel, dim3(mansy_ceil_div((long long)n * 64, 256)), dim3(256), 0, (hipStream_t)stream, pred_viewport,
                     actions, n, T, versions);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_expert_profile(const mansy_env_tables* T, const int* vp_video, int n_vp, float* gt_quality, float* pred_quality, float* gt_var,
                         float* pred_var, int* gt_size, int* pred_size, void* stream) {
  int rc = check_tables(T); if (rc) return rc;
  MANSY_REQUIRE(vp_video && n_vp >= 1 && gt_quality && pred_quality && gt_var && pred_var && gt_size && pred_size,
                "expert_profile: bad arguments");
  const long long waves = (long long)n_vp * T->n_vpchunk_max * N_ACTION * 2;
  MANSY_LAUNCH(expert_profile_kernel, dim3((unsigned)mansy_ceil_div(waves * 64, 256)), dim3(256), 0, (hipStream_t)stream, *T,
                     vp_video, n_vp, gt_quality, pred_quality, gt_var, pred_var, gt_size, pred_size);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

int mansy_expert_choose_action(const mansy_env_tables* T, const void* state, int n_env, int horizon, const float* pred_quality,
                               const float* pred_var, const int* pred_size, unsigned long long* keys, int* actions, float* best_value,
                               long long* best_index, void* stream) {
  int rc = check_tables(T); if (rc) return rc;
  MANSY_REQUIRE(state && n_env >= 1 && pred_quality && pred_var && pred_size && keys && actions, "expert_choose_action: bad arguments");
  MANSY_REQUIRE(horizon >= 1 && horizon <= MANSY_EXPERT_MAX_HORIZON, "expert_choose_action: horizon must be in [1, %d]",
                MANSY_EXPERT_MAX_HORIZON);
  unsigned n_prefix = 1;
  for (int t = 0; t < horizon - 1; ++t) n_prefix *= N_ACTION;
  const dim3 eb(mansy_ceil_div(n_env, 256));
  MANSY_LAUNCH(expert_keys_init_kernel, eb, dim3(256), 0, (hipStream_t)stream, keys, n_env);
  MANSY_LAUNCH(expert_search_kernel, dim3(mansy_ceil_div(n_prefix, 256u), n_env), dim3(256), 0, (hipStream_t)stream, *T,
                     (const EnvState*)state, horizon, pred_quality, pred_var, pred_size, keys);
  MANSY_LAUNCH(expert_pick_kernel, eb, dim3(256), 0, (hipStream_t)stream, keys, n_env, actions, best_value, best_index);
  MANSY_LAUNCH_CHECK();
  return MANSY_OK;
}

}  // extern "C"
