// Device side of the streaming environment (one 64-lane wavefront per environment, lane = tile), shared by csrc/env.hip and the
// fused rollout launch in csrc/ppo_engine.hip.  See env.hip for the reference citations and the bit-exactness contract.
// Include inside an anonymous namespace, after `#pragma clang fp contract(off)`-sensitive code has been considered: every function
// here that does double arithmetic relies on the including file's `#pragma clang fp contract(off)`.
#pragma once

constexpr int NTL = 64, NR = 5, PAST_K = 8, N_ACTION = 15;
constexpr int OBS_LD = MANSY_OBS_LD;
static __constant__ int A2R[N_ACTION][2] = {{1,0},{2,0},{3,0},{4,0},{2,1},{3,1},{4,1},{3,2},{4,2},{4,3},{0,0},{1,1},{2,2},{3,3},{4,4}};

typedef float RingMem[PAST_K];

// Per-environment record. EnvState (history rings as float[8], AoS, 256 bytes) is the layout in HBM; EnvRegs is the copy a wave
// works on: the scalars are held by every lane (uniform), each history ring is spread over lanes 0..7 (lane k holds element k),
// so a ring costs one register instead of eight, pushing a value is one row-shift and the observation's history slots are
// written straight from the lanes that hold them.  (A local record with array members would live in scratch memory: a
// lane-indexed ring read becomes a dynamically indexed private load, which pins the stack object.)
template <typename Ring>
struct EnvRec {
  int worker_id, worker_num, sample_id;
  int video, vp, trace, qoe;
  int next_chunk, end_chunk;
  int cur_idx, has_prev, log_n;
  double cur_time, buf_size, last_chunk_accuracy;
  double log_qoe, log_qoe1, log_qoe2, log_qoe3;
  float prev_vq, buffer0;
  Ring past_throughput, past_acc, past_in, past_out, past_q, past_var, past_rebuf;
};
using EnvState = EnvRec<RingMem>;
using EnvRegs = EnvRec<float>;

template <typename D, typename S>
__device__ __forceinline__ void copy_scalars(D& d, const S& s) {   // member-wise: a whole-struct copy is a memcpy via the stack
  d.worker_id = s.worker_id; d.worker_num = s.worker_num; d.sample_id = s.sample_id;
  d.video = s.video; d.vp = s.vp; d.trace = s.trace; d.qoe = s.qoe;
  d.next_chunk = s.next_chunk; d.end_chunk = s.end_chunk;
  d.cur_idx = s.cur_idx; d.has_prev = s.has_prev; d.log_n = s.log_n;
  d.cur_time = s.cur_time; d.buf_size = s.buf_size; d.last_chunk_accuracy = s.last_chunk_accuracy;
  d.log_qoe = s.log_qoe; d.log_qoe1 = s.log_qoe1; d.log_qoe2 = s.log_qoe2; d.log_qoe3 = s.log_qoe3;
  d.prev_vq = s.prev_vq; d.buffer0 = s.buffer0;
}
__device__ __forceinline__ void load_state(EnvRegs& d, const EnvState& s, int lane) {   // every lane of the wave calls this
  copy_scalars(d, s);
  const int k = lane & (PAST_K - 1);
  d.past_throughput = s.past_throughput[k]; d.past_acc = s.past_acc[k]; d.past_in = s.past_in[k]; d.past_out = s.past_out[k];
  d.past_q = s.past_q[k]; d.past_var = s.past_var[k]; d.past_rebuf = s.past_rebuf[k];
}
__device__ __forceinline__ void store_state(EnvState& d, const EnvRegs& s, int lane) {  // every lane of the wave calls this
  if (lane == 0) copy_scalars(d, s);
  if (lane < PAST_K) {
    d.past_throughput[lane] = s.past_throughput; d.past_acc[lane] = s.past_acc; d.past_in[lane] = s.past_in; d.past_out[lane] = s.past_out;
    d.past_q[lane] = s.past_q; d.past_var[lane] = s.past_var; d.past_rebuf[lane] = s.past_rebuf;
  }
}

__device__ __forceinline__ unsigned long long dilate8(unsigned long long m) {
  // columns +-1 with wrap inside each row byte, then rows +-1 with wrap (rotate by 8 bits)
  const unsigned long long l = ((m << 1) & 0xFEFEFEFEFEFEFEFEull) | ((m >> 7) & 0x0101010101010101ull);
  const unsigned long long r = ((m >> 1) & 0x7F7F7F7F7F7F7F7Full) | ((m << 7) & 0x8080808080808080ull);
  const unsigned long long h = m | l | r;
  return h | (h << 8) | (h >> 56) | (h >> 8) | (h << 56);
}

// The five bitrates as named scalars (a local int[5] with a runtime index would live in scratch memory).
struct Rates { int r0, r1, r2, r3, r4; };
__device__ __forceinline__ Rates load_rates(const mansy_env_tables& T) {
  Rates r = {T.video_rates[0], T.video_rates[1], T.video_rates[2], T.video_rates[3], T.video_rates[4]};
  return r;
}
__device__ __forceinline__ int pick_rate(const Rates& r, int i) {
  return i == 4 ? r.r4 : i == 3 ? r.r3 : i == 2 ? r.r2 : i == 1 ? r.r1 : r.r0;
}
__device__ __forceinline__ void closer(int cand, int cand_rate, int rate, int& ver, int& ver_rate, int& gap) {
  const int g = abs(cand_rate - rate);
  if (g < gap || (g == gap && cand_rate < ver_rate)) { ver = cand; ver_rate = cand_rate; gap = g; }
}
__device__ __forceinline__ int closest_rate_version(const Rates& r, int rate) {   // nearest bitrate, ties to the lower one
  int ver = 0, ver_rate = r.r0, gap = abs(r.r0 - rate);
  closer(1, r.r1, rate, ver, ver_rate, gap);
  closer(2, r.r2, rate, ver, ver_rate, gap);
  closer(3, r.r3, rate, ver, ver_rate, gap);
  closer(4, r.r4, rate, ver, ver_rate, gap);
  return ver;
}

__device__ __forceinline__ float seq_sum64(float x) {      // ((..(0 + x0) + x1) ..) + x63, float32, tile order
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < NTL; ++t) s = s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), t));
  return s;
}
// The same sequential sum restricted to the lanes of `mask` (tile order).  Every skipped term must be +0.0f: s + 0.0f == s
// bit for bit for the non-negative partial sums that occur here, so skipping them changes nothing -- a viewport covers
// 9..20 of the 64 tiles, which makes the dependent-add chain 3-7x shorter.
__device__ __forceinline__ float seq_sum_masked(float x, unsigned long long mask) {
  float s = 0.f;
  while (mask) {
    const int t = __builtin_ctzll(mask);
    s = s + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), t));
    mask &= mask - 1;
  }
  return s;
}
// Quality sums over the ground-truth viewport gv (tile-order float32 sums of the reference, qoe.py:23-24 / simulator.py:156):
// returns s_v = sum(gv), s_vq = sum(gv * tq); var_sum(...) then gives sum(gv * |tq - vq|).  For 0/1 maps with finite
// non-negative qualities (the data format) only the viewport's tiles are visited; anything else takes the full 64-term chain.
struct ViewportSums { float s_v, s_vq; unsigned long long mask; bool fast; };
__device__ __forceinline__ ViewportSums viewport_sums(float gv, float tq) {
  ViewportSums r;
  r.mask = __ballot(gv != 0.f);
  r.fast = r.mask != 0ull && __ballot(!(gv == 0.f || gv == 1.f) || !(tq >= 0.f && tq < 3.0e38f)) == 0ull;   // (empty viewport: NaN path, keep it literal)
  if (r.fast) { r.s_v = (float)__popcll(r.mask); r.s_vq = seq_sum_masked(gv * tq, r.mask); }
  else { r.s_vq = seq_sum64(gv * tq); r.s_v = seq_sum64(gv); }
  return r;
}
__device__ __forceinline__ float var_sum(const ViewportSums& r, float gv, float tq, float vq) {
  const float term = gv * fabsf(tq - vq);
  return r.fast ? seq_sum_masked(term, r.mask) : seq_sum64(term);
}

__device__ __forceinline__ int wave_isum(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// The table rows an observation shows (next chunk's sizes / qualities for all five versions, predicted viewport), loaded
// separately from their use so that the step kernel can issue them before its dependent chain of simulator loads.
struct ObsRows { int size[NR]; float quality[NR]; unsigned char pred; };
__device__ __forceinline__ ObsRows load_obs_rows(const mansy_env_tables& T, int video, int vp, int chunk, int lane) {
  ObsRows o;
  const size_t mrow = ((size_t)video * T.n_chunk_max + chunk) * NR * NTL;
#pragma unroll
  for (int r = 0; r < NR; ++r) { o.size[r] = T.size[mrow + r * NTL + lane]; o.quality[r] = T.quality[mrow + r * NTL + lane]; }
  o.pred = T.vp_pred[((size_t)vp * T.n_vpchunk_max + (chunk - T.vp_start[vp])) * NTL + lane];
  return o;
}
__device__ __forceinline__ void store_obs(const mansy_env_tables& T, const EnvRegs& s, const ObsRows& rows, int action, int lane,
                                          float* __restrict__ obs) {
  const float inv_rate = (float)T.video_rates[NR - 1];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    obs[MANSY_O_SIZE + r * NTL + lane] = (float)rows.size[r] / (float)T.max_size;
    obs[MANSY_O_QUALITY + r * NTL + lane] = rows.quality[r] / inv_rate;
  }
  obs[MANSY_O_PRED_VP + lane] = (float)rows.pred;
  // the 68 scalar slots: 0..7 throughput | 712..743 acc,q,var,rebuf | 744 buffer | 745..747 qoe_w | 748..762 one-hot |
  // 763..778 rates in/out | 779 pad
  if (lane < PAST_K) {
    obs[MANSY_O_THROUGHPUT + lane] = s.past_throughput;
    obs[MANSY_O_VP_ACC + lane] = s.past_acc;
    obs[MANSY_O_PAST_Q + lane] = s.past_q;
    obs[MANSY_O_PAST_VAR + lane] = s.past_var;
    obs[MANSY_O_PAST_REBUF + lane] = s.past_rebuf;
    obs[MANSY_O_RATES_IN + lane] = s.past_in;
    obs[MANSY_O_RATES_OUT + lane] = s.past_out;
  }
  const float* w = T.qoe_w + 3 * s.qoe;
  const float wsum = (w[0] + w[1]) + w[2];
  if (lane < 3) obs[MANSY_O_QOE_W + lane] = w[lane] / wsum;
  if (lane < N_ACTION) obs[MANSY_O_ACT_1HOT + lane] = (lane == action) ? 1.f : 0.f;
  if (lane == 0) { obs[MANSY_O_BUFFER] = s.buffer0 / (float)T.startup_download; obs[MANSY_OBS_DIM] = 0.f; }
}
__device__ __forceinline__ void write_obs(const mansy_env_tables& T, const EnvRegs& s, int chunk, int action, int lane, float* __restrict__ obs) {
  store_obs(T, s, load_obs_rows(T, s.video, s.vp, chunk, lane), action, lane, obs);
}

__device__ __forceinline__ void do_reset(const mansy_env_tables& T, EnvRegs& s) {
  s.sample_id = s.worker_id % T.n_sample;   // (reference: IndexError if worker_id >= len(samples); wrap instead)
  s.worker_id = (s.worker_id + s.worker_num) % T.n_sample;
  const int* sm = T.samples + 4 * s.sample_id;
  s.video = sm[0]; s.vp = sm[1]; s.trace = sm[2]; s.qoe = sm[3];
  s.buf_size = (double)(T.chunk_length * 3);
  s.cur_time = 0.0; s.cur_idx = 0;
  const int end_chunk = T.vp_end[s.vp], vlen1 = T.video_len[s.video] - 1;
  s.end_chunk = end_chunk < vlen1 ? end_chunk : vlen1;
  s.next_chunk = T.startup_download + 1;
  s.has_prev = 0; s.prev_vq = 0.f;
  s.last_chunk_accuracy = T.vp_acc[(size_t)s.vp * T.n_vpchunk_max + (s.next_chunk - T.vp_start[s.vp])];
  s.past_throughput = 0.f; s.past_acc = 0.f; s.past_in = 0.f; s.past_out = 0.f; s.past_q = 0.f; s.past_var = 0.f; s.past_rebuf = 0.f;
  s.buffer0 = (float)s.buf_size;
  s.log_qoe = s.log_qoe1 = s.log_qoe2 = s.log_qoe3 = 0.0; s.log_n = 0;
}

// np.roll(ring, 1); ring[0] = v on a lane-distributed ring: lane k takes lane k-1's element (row shift right by one; lane 0, which
// has no source lane, keeps the `old` operand = v).  v is uniform over the wave.
__device__ __forceinline__ void roll_push(float& ring, float v) {
  ring = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, ring), 0x111, 0xf, 0xf, false));
}

// One environment step by one wavefront (lane = tile): MANSYEnv.step (mansy_env.py:154-248).  obs_next: the post-action
// observation row (terminal one when the episode ends); obs_cur (optional): the same, except that a finished environment is reset and
// shows the first observation of its next episode.  Used by env_step_kernel and by the fused policy + environment rollout launch.
//
// The step is split where the ACTION enters: env_step_requests() issues every load that depends on the state alone, env_step_finish()
// does the rest.  env_step_kernel calls them back to back; the fused rollout launch (ppo_engine.hip head_out_kernel) loads the state and
// issues the requests BEFORE the policy's output layer, so that the two dependent round trips (state -> table rows) fly under the
// logits / softmax / sampling arithmetic instead of behind it.
struct EnvPre {
  ObsRows rows_next, rows_cur; float gv, w0, w1, w2; const double* bw; int tlen; double bwc, acc_next; Rates rates;
};
__device__ __forceinline__ EnvPre env_step_requests(const mansy_env_tables& T, const EnvRegs& s, int lane) {
  EnvPre q;
  const int chunk = s.next_chunk;
  // the observation after this step shows chunk + 1 (or, when the episode ends, this chunk again): known now, so its rows are
  // requested first and arrive while the simulator's dependent loads are in flight
  const bool over_pre = chunk + 1 > s.end_chunk;
  q.rows_next = load_obs_rows(T, s.video, s.vp, over_pre ? chunk : chunk + 1, lane);
  // everything else that depends on the state alone is requested up front too: this chunk's five versions (the allocated version
  // is selected from registers instead of a load that waits for the allocation), the preference weights, the trace bin the
  // download starts in, the next chunk's prediction accuracy
  q.rows_cur = load_obs_rows(T, s.video, s.vp, chunk, lane);
  const size_t vrow = ((size_t)s.vp * T.n_vpchunk_max + (chunk - T.vp_start[s.vp])) * NTL;
  q.gv = (float)T.vp_gt[vrow + lane];
  const float* w = T.qoe_w + 3 * s.qoe;
  q.w0 = w[0]; q.w1 = w[1]; q.w2 = w[2];
  q.bw = T.trace_bw + (size_t)s.trace * T.trace_len_max;
  q.tlen = T.trace_len[s.trace];
  q.bwc = q.bw[s.cur_idx];                                  // always bw[s.cur_idx]
  q.acc_next = over_pre ? 0.0 : T.vp_acc[(size_t)s.vp * T.n_vpchunk_max + (chunk + 1 - T.vp_start[s.vp])];
  q.rates = load_rates(T);
  return q;
}
__device__ __forceinline__ void env_step_finish(const mansy_env_tables& T, EnvState* st, int e, int lane, int action, EnvRegs& s, const EnvPre& q,
                                                float* obs_next, float* obs_cur, float* reward, unsigned char* done, float* qoe_parts,
                                                const mansy_env_episode_log& elog) {
  const int rin = (action >= 0 && action < N_ACTION) ? A2R[action][0] : 0;
  const int rout = (action >= 0 && action < N_ACTION) ? A2R[action][1] : 0;
  const int chunk = s.next_chunk;
  const ObsRows& rows_next = q.rows_next; const ObsRows& rows_cur = q.rows_cur;
  const float gv = q.gv, w0 = q.w0, w1 = q.w1, w2 = q.w2;
  const double* bw = q.bw; const int tlen = q.tlen; double bwc = q.bwc; const double acc_next = q.acc_next;
  const bool in_pred = rows_cur.pred == 1;
  // ---- pyramid tile-rate allocation
  unsigned long long m = __ballot(in_pred);
  int dist = ((m >> lane) & 1ull) ? 0 : -1;
  if (m == 0ull) dist = 0;                              // empty prediction: BFS queue empty => every scale stays 0
  else {
#pragma unroll
    for (int sidx = 1; sidx <= 4; ++sidx) {
      m = dilate8(m);
      if (dist < 0 && ((m >> lane) & 1ull)) dist = sidx;
    }
  }
  const Rates rates = q.rates;
  const int ver = dist == 0 ? rin : closest_rate_version(rates, pick_rate(rates, rout) / (dist > 0 ? dist : 1));
  // ---- Simulator.simulate_download
  int my_size = rows_cur.size[0]; float tq = rows_cur.quality[0];
#pragma unroll
  for (int r = 1; r < NR; ++r) { my_size = ver == r ? rows_cur.size[r] : my_size; tq = ver == r ? rows_cur.quality[r] : tq; }
  const int chunk_size = wave_isum(my_size);
  const double start = s.cur_time;
  double size = (double)chunk_size;
  while (size > 0) {
    const double fl = floor(s.cur_time + 1);
    const double remain = (fl - s.cur_time) * bwc;
    if (size >= remain) { s.cur_idx = s.cur_idx + 1 == tlen ? 0 : s.cur_idx + 1; bwc = bw[s.cur_idx]; s.cur_time = fl; size -= remain; }
    else { s.cur_time += size / bwc; size = 0; }
  }
  const double download_time = s.cur_time - start;
  double rebuf = 0.0;
  if (download_time > s.buf_size) { rebuf = download_time - s.buf_size; s.buf_size = (double)T.chunk_length; }
  else s.buf_size = s.buf_size - download_time + (double)T.chunk_length;
  s.next_chunk += 1;
  const bool over = s.next_chunk > s.end_chunk;
  // ---- QoE (sequential float32 sums in tile order)
  const ViewportSums vs = viewport_sums(gv, tq);
  const float s_v = vs.s_v;
  float vq = vs.s_vq / s_v;
  const float s_var = var_sum(vs, gv, tq, vq);
  const float max_rate = (float)rates.r4;
  const float intra = (s_var / s_v) / max_rate;
  vq = vq / max_rate;
  const float inter = s.has_prev ? fabsf(vq - s.prev_vq) : 0.f;
  s.prev_vq = vq; s.has_prev = 1;
  const float qoe1 = vq, qoe3 = intra + inter;
  const float qoe = w0 * qoe1 - w1 * (float)rebuf - w2 * qoe3;
  const float wsum = (w0 + w1) + w2;
  const float rew = T.train_identifier_reward ? qoe / wsum : qoe;
  s.log_qoe += (double)qoe; s.log_qoe1 += (double)qoe1; s.log_qoe2 += rebuf; s.log_qoe3 += (double)qoe3; s.log_n += 1;
  // ---- history rings
  roll_push(s.past_throughput, (float)(((double)chunk_size / download_time) / T.max_throughput));
  roll_push(s.past_acc, (float)s.last_chunk_accuracy);
  roll_push(s.past_in, (float)((double)pick_rate(rates, rin) / (double)rates.r4));
  roll_push(s.past_out, (float)((double)pick_rate(rates, rout) / (double)rates.r4));
  s.buffer0 = (float)s.buf_size;
  roll_push(s.past_q, qoe1);
  roll_push(s.past_rebuf, (float)(rebuf / (double)T.startup_download));
  roll_push(s.past_var, qoe3);
  if (!over) s.last_chunk_accuracy = acc_next;
  store_obs(T, s, rows_next, action, lane, obs_next + (size_t)e * OBS_LD);
  if (lane == 0) {
    reward[e] = rew;
    done[e] = over ? 1 : 0;
    if (qoe_parts) { qoe_parts[4 * e + 0] = qoe; qoe_parts[4 * e + 1] = qoe1; qoe_parts[4 * e + 2] = (float)rebuf; qoe_parts[4 * e + 3] = qoe3; }
  }
  if (over) {
    if (lane == 0 && elog.records && elog.count) {       // episode summary for the CSV log (mansy_env.py:271-290)
      const unsigned slot = atomicAdd(elog.count, 1u);
      if (slot < (unsigned)elog.capacity) {
        double* r = elog.records + (size_t)slot * 8;
        r[0] = (double)s.sample_id; r[1] = (double)e; r[2] = (double)s.log_n; r[3] = s.log_qoe; r[4] = s.log_qoe1; r[5] = s.log_qoe2;
        r[6] = s.log_qoe3; r[7] = (double)s.qoe;
      }
    }
    if (obs_cur) {            // auto-reset: the observation the policy sees next comes from the new episode
      do_reset(T, s);
      write_obs(T, s, s.next_chunk, -1, lane, obs_cur + (size_t)e * OBS_LD);
    }
  } else if (obs_cur && obs_cur != obs_next) {
    store_obs(T, s, rows_next, action, lane, obs_cur + (size_t)e * OBS_LD);
  }
  store_state(st[e], s, lane);
}
__device__ __forceinline__ void env_step_wave(const mansy_env_tables& T, EnvState* st, int e, int lane, int action, float* obs_next, float* obs_cur,
                                              float* reward, unsigned char* done, float* qoe_parts, const mansy_env_episode_log& elog) {
  EnvRegs s;
  load_state(s, st[e], lane);
  const EnvPre q = env_step_requests(T, s, lane);
  env_step_finish(T, st, e, lane, action, s, q, obs_next, obs_cur, reward, done, qoe_parts, elog);
}

