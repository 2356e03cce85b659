/* TEST INFRASTRUCTURE ONLY (oracle). Not linked into the product library.
 *
 * Plain-C, strictly sequential restatement of the reference's trace-driven streaming environment:
 *   MANSYEnv.reset / step                bitrate_selection/envs/mansy_env.py:99-248
 *   action2rates, allocate_tile_rates    bitrate_selection/utils/common.py:101-119, 142-193
 *   normalize_*                          bitrate_selection/utils/common.py:40-57
 *   Simulator.simulate_download          bitrate_selection/simulators/simulator.py:88-108
 *   NetworkTrace.simulate_download       bitrate_selection/simulators/network.py:22-35
 *   PlaybackBuffer.push_chunk            bitrate_selection/simulators/buffer.py:8-15
 *   HMDTrace.get_viewport                bitrate_selection/simulators/hmdtrace.py:16-23
 *   QoEModel.calculate_qoe               bitrate_selection/utils/qoe.py:22-34
 *   ExpertEnv cache profile / choose_action   bitrate_selection/envs/expert_env.py:126-181, 358-422
 *   ExpertSimulator.virtual_simulate_download_with_chunk_size, calculate_chunk_size_and_quality
 *                                        bitrate_selection/simulators/simulator.py:127-158
 *   QoEModelExpert.calculate_qoe_with_given_quality   bitrate_selection/utils/qoe.py:49-59
 *
 * Number semantics are those of THIS container (Python 3.10, numpy 2.2 / NEP 50), pinned by
 * tests/golden/env_reference.npz and expert_reference.npz which were produced by running the imported reference:
 *   - chunk size: exact integer sum; download time / buffer / rebuffer: Python floats (IEEE double,
 *     no FMA contraction -- build with -ffp-contract=off);
 *   - QoE: Python sum() over float32 arrays = sequential float32 accumulation in tile order;
 *     np.float32 (op) python-scalar stays float32;
 *   - BFS ring distance on the 8-neighbour torus is restated literally (queue), the HIP kernel uses
 *     the equivalent toroidal Chebyshev distance.
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -shared -fPIC (oracle/build.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NT 64            /* tiles */
#define NR 5             /* rate versions */
#define PAST_K 8
#define N_ACTION 15
#define OBS_DIM 779
#define OBS_LD 780

/* obs vector offsets (float32) -- shared with csrc/env.hip and the host mirrors */
#define O_THROUGHPUT 0
#define O_SIZE 8
#define O_QUALITY 328
#define O_PRED_VP 648
#define O_VP_ACC 712
#define O_PAST_Q 720
#define O_PAST_VAR 728
#define O_PAST_REBUF 736
#define O_BUFFER 744
#define O_QOE_W 745
#define O_ACT_1HOT 748
#define O_RATES_IN 763
#define O_RATES_OUT 771

typedef struct {
    /* video manifests: [n_video][n_chunk_max][NR][NT] */
    const int32_t *size;  const float *quality;  const int32_t *video_len;  int n_chunk_max;
    /* viewports: [n_vp][n_vpchunk_max][NT] u8, accuracy [n_vp][n_vpchunk_max] f64, first/last chunk ids */
    const uint8_t *vp_gt; const uint8_t *vp_pred; const double *vp_acc; const int32_t *vp_start; const int32_t *vp_end;
    int n_vpchunk_max;
    /* network traces: [n_trace][trace_len_max] bytes/s (double), lengths */
    const double *trace_bw; const int32_t *trace_len; int trace_len_max;
    /* episode catalogue: [n_sample][4] = (video slot, viewport slot, trace slot, qoe index); weights [n_qoe][3] */
    const int32_t *samples; int n_sample; const float *qoe_w;
    /* config.yml constants */
    int video_rates[NR]; int startup_download; int chunk_length; double max_size; double max_throughput;
    int train_identifier_reward;     /* 1: reward = qoe / sum(w)  (mode == train && use_identifier) */
} oracle_env_tables;

typedef struct {
    int worker_id, worker_num, sample_id;
    int video, vp, trace, qoe;
    int next_chunk, end_chunk;
    double cur_time; int cur_idx;        /* NetworkTrace */
    double buf_size;                     /* PlaybackBuffer */
    int has_prev; float prev_vq;         /* QoEModel.prev_viewport_quality */
    double last_chunk_accuracy;
    float past_throughput[PAST_K], past_acc[PAST_K], past_in[PAST_K], past_out[PAST_K], past_q[PAST_K], past_var[PAST_K],
        past_rebuf[PAST_K];
    float buffer0;                       /* self.buffer[0] (float32 array element) */
    double log_qoe, log_qoe1, log_qoe2, log_qoe3; int log_n;   /* per-episode CSV accumulators (Python float sums) */
} oracle_env_state;

static void action2rates(int a, int *rin, int *rout) {
    static const int t[N_ACTION][2] = {{1,0},{2,0},{3,0},{4,0},{2,1},{3,1},{4,1},{3,2},{4,2},{4,3},{0,0},{1,1},{2,2},{3,3},{4,4}};
    *rin = 0; *rout = 0;
    if (a >= 0 && a < N_ACTION) { *rin = t[a][0]; *rout = t[a][1]; }
}

static int pyfloordiv(int a, int b) { int q = a / b; if ((a % b != 0) && ((a < 0) != (b < 0))) q--; return q; }

static int closest_rate_version(const int *rates, int rate) {
    int ver = 0, gap = abs(rates[0] - rate);
    for (int i = 0; i < NR; ++i) {
        int g = abs(rates[i] - rate);
        if (g < gap) { ver = i; gap = g; }
        else if (g == gap && rates[i] < rates[ver]) ver = i;
    }
    return ver;
}

/* utils/common.py:142-193 -- literal BFS */
void oracle_allocate_tile_rates(int rate_in, int rate_out, const float *pred_viewport, const int *rates, int32_t *versions) {
    int scales[8][8], visited[8][8];
    int qr[64], qc[64], head = 0, tail = 0;
    memset(scales, 0, sizeof(scales)); memset(visited, 0, sizeof(visited));
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c)
        if (pred_viewport[r * 8 + c] == 1.0f) { visited[r][c] = 1; qr[tail] = r; qc[tail] = c; ++tail; }
    static const int dir[8][2] = {{1,0},{-1,0},{0,1},{0,-1},{1,-1},{-1,-1},{1,1},{-1,1}};
    while (head < tail) {
        int r = qr[head], c = qc[head]; ++head;
        for (int d = 0; d < 8; ++d) {
            int nr = ((r + dir[d][0]) % 8 + 8) % 8, nc = ((c + dir[d][1]) % 8 + 8) % 8;
            if (!visited[nr][nc]) { scales[nr][nc] = scales[r][c] + 1; qr[tail] = nr; qc[tail] = nc; ++tail; visited[nr][nc] = 1; }
        }
    }
    int max_scale = 0;
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) if (scales[r][c] > max_scale) max_scale = scales[r][c];
    for (int i = 0; i < 64; ++i) versions[i] = 0;
    for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) if (scales[r][c] == 0) versions[r * 8 + c] = rate_in;
    for (int s = 1; s <= max_scale; ++s) {
        int v = closest_rate_version(rates, pyfloordiv(rates[rate_out], s));
        for (int r = 0; r < 8; ++r) for (int c = 0; c < 8; ++c) if (scales[r][c] == s) versions[r * 8 + c] = v;
    }
}

static void roll_push(float *ring, float v) {       /* np.roll(x, 1); x[0,0] = v */
    for (int i = PAST_K - 1; i > 0; --i) ring[i] = ring[i - 1];
    ring[0] = v;
}

static const uint8_t *vp_row(const oracle_env_tables *T, const uint8_t *base, int vp, int chunk) {
    return base + ((size_t)vp * T->n_vpchunk_max + (chunk - T->vp_start[vp])) * NT;
}

static void write_obs(const oracle_env_tables *T, const oracle_env_state *s, int action, float *obs) {
    memset(obs, 0, sizeof(float) * OBS_LD);
    const int32_t *sz = T->size + ((size_t)s->video * T->n_chunk_max + s->next_chunk) * NR * NT;
    const float *ql = T->quality + ((size_t)s->video * T->n_chunk_max + s->next_chunk) * NR * NT;
    for (int i = 0; i < PAST_K; ++i) {
        obs[O_THROUGHPUT + i] = s->past_throughput[i]; obs[O_VP_ACC + i] = s->past_acc[i]; obs[O_PAST_Q + i] = s->past_q[i];
        obs[O_PAST_VAR + i] = s->past_var[i]; obs[O_PAST_REBUF + i] = s->past_rebuf[i];
        obs[O_RATES_IN + i] = s->past_in[i]; obs[O_RATES_OUT + i] = s->past_out[i];
    }
    /* normalize_size(...).astype(float32): float32 array / python int -> float32 division */
    for (int i = 0; i < NR * NT; ++i) obs[O_SIZE + i] = (float)sz[i] / (float)T->max_size;
    for (int i = 0; i < NR * NT; ++i) obs[O_QUALITY + i] = ql[i] / (float)T->video_rates[NR - 1];
    const uint8_t *pv = vp_row(T, T->vp_pred, s->vp, s->next_chunk);
    for (int i = 0; i < NT; ++i) obs[O_PRED_VP + i] = (float)pv[i];
    obs[O_BUFFER] = s->buffer0 / (float)T->startup_download;
    const float *w = T->qoe_w + 3 * s->qoe;
    float wsum = (w[0] + w[1]) + w[2];                 /* python sum(): ((0 + w0) + w1) + w2 in float32 */
    for (int i = 0; i < 3; ++i) obs[O_QOE_W + i] = w[i] / wsum;
    if (action >= 0) obs[O_ACT_1HOT + action] = 1.0f;
}

/* mansy_env.py:99-152 */
void oracle_env_reset(const oracle_env_tables *T, oracle_env_state *s, float *obs) {
    s->sample_id = s->worker_id % T->n_sample;   /* reference: IndexError if worker_id >= len(samples); wrap instead */
    s->worker_id = (s->worker_id + s->worker_num) % T->n_sample;
    const int32_t *sm = T->samples + 4 * s->sample_id;
    s->video = sm[0]; s->vp = sm[1]; s->trace = sm[2]; s->qoe = sm[3];
    /* Simulator.__init__ (simulator.py:28-45) */
    s->buf_size = (double)(T->chunk_length * 3);
    s->cur_time = 0.0; s->cur_idx = 0;
    int end_chunk = T->vp_end[s->vp];
    int vlen1 = T->video_len[s->video] - 1;
    s->end_chunk = end_chunk < vlen1 ? end_chunk : vlen1;
    s->next_chunk = T->startup_download + 1;
    s->has_prev = 0; s->prev_vq = 0.f;
    s->last_chunk_accuracy = T->vp_acc[(size_t)s->vp * T->n_vpchunk_max + (s->next_chunk - T->vp_start[s->vp])];
    memset(s->past_throughput, 0, sizeof(float) * PAST_K); memset(s->past_acc, 0, sizeof(float) * PAST_K);
    memset(s->past_in, 0, sizeof(float) * PAST_K); memset(s->past_out, 0, sizeof(float) * PAST_K);
    memset(s->past_q, 0, sizeof(float) * PAST_K); memset(s->past_var, 0, sizeof(float) * PAST_K);
    memset(s->past_rebuf, 0, sizeof(float) * PAST_K);
    s->buffer0 = (float)s->buf_size;
    write_obs(T, s, -1, obs);
}

/* mansy_env.py:154-248.  Returns done; *reward is float32 like the reference's np.float32 result. */
int oracle_env_step(const oracle_env_tables *T, oracle_env_state *s, int action, float *obs, float *reward, float *qoe_parts) {
    int rin, rout; action2rates(action, &rin, &rout);
    const uint8_t *pv = vp_row(T, T->vp_pred, s->vp, s->next_chunk);
    const uint8_t *gv = vp_row(T, T->vp_gt, s->vp, s->next_chunk);
    float pvf[NT]; for (int i = 0; i < NT; ++i) pvf[i] = (float)pv[i];
    int32_t ver[NT];
    oracle_allocate_tile_rates(rin, rout, pvf, T->video_rates, ver);
    /* Simulator.simulate_download */
    const int32_t *sz = T->size + ((size_t)s->video * T->n_chunk_max + s->next_chunk) * NR * NT;
    const float *ql = T->quality + ((size_t)s->video * T->n_chunk_max + s->next_chunk) * NR * NT;
    long long chunk_size = 0; float tq[NT];
    for (int t = 0; t < NT; ++t) { chunk_size += sz[ver[t] * NT + t]; tq[t] = ql[ver[t] * NT + t]; }
    /* NetworkTrace.simulate_download (network.py:22-35), Python floats */
    const double *bw = T->trace_bw + (size_t)s->trace * T->trace_len_max;
    const int tlen = T->trace_len[s->trace];
    double start = s->cur_time, size = (double)chunk_size;
    while (size > 0) {
        double remain = (floor(s->cur_time + 1) - s->cur_time) * bw[s->cur_idx];
        if (size >= remain) { s->cur_idx = (s->cur_idx + 1) % tlen; s->cur_time = floor(s->cur_time + 1); size -= remain; }
        else { s->cur_time += size / bw[s->cur_idx]; size = 0; }
    }
    double download_time = s->cur_time - start;
    /* PlaybackBuffer.push_chunk */
    double rebuf = 0.0;
    if (download_time > s->buf_size) { rebuf = download_time - s->buf_size; s->buf_size = (double)T->chunk_length; }
    else s->buf_size = s->buf_size - download_time + (double)T->chunk_length;
    s->next_chunk += 1;
    int over = s->next_chunk > s->end_chunk;
    /* QoEModel.calculate_qoe (qoe.py:22-34): sequential float32 sums over the GT viewport */
    float s_vq = 0.f, s_v = 0.f;
    for (int t = 0; t < NT; ++t) { s_vq += (float)gv[t] * tq[t]; s_v += (float)gv[t]; }
    float vq = s_vq / s_v;
    float s_var = 0.f;
    for (int t = 0; t < NT; ++t) s_var += (float)gv[t] * fabsf(tq[t] - vq);
    float intra = (s_var / s_v) / (float)T->video_rates[NR - 1];
    vq = vq / (float)T->video_rates[NR - 1];
    float inter = s->has_prev ? fabsf(vq - s->prev_vq) : 0.0f;
    s->prev_vq = vq; s->has_prev = 1;
    const float *w = T->qoe_w + 3 * s->qoe;
    float qoe1 = vq, qoe2f = (float)rebuf, qoe3 = intra + inter;
    float qoe = w[0] * qoe1 - w[1] * qoe2f - w[2] * qoe3;
    float wsum = (w[0] + w[1]) + w[2];
    *reward = T->train_identifier_reward ? qoe / wsum : qoe;
    if (qoe_parts) { qoe_parts[0] = qoe; qoe_parts[1] = qoe1; qoe_parts[2] = (float)rebuf; qoe_parts[3] = qoe3; }
    s->log_qoe += (double)qoe; s->log_qoe1 += (double)qoe1; s->log_qoe2 += rebuf; s->log_qoe3 += (double)qoe3; s->log_n += 1;
    /* history rings (mansy_env.py:192-206) */
    roll_push(s->past_throughput, (float)(((double)chunk_size / download_time) / T->max_throughput));
    roll_push(s->past_acc, (float)s->last_chunk_accuracy);
    roll_push(s->past_in, (float)((double)T->video_rates[rin] / (double)T->video_rates[NR - 1]));
    roll_push(s->past_out, (float)((double)T->video_rates[rout] / (double)T->video_rates[NR - 1]));
    s->buffer0 = (float)s->buf_size;
    roll_push(s->past_q, qoe1);
    roll_push(s->past_rebuf, (float)(rebuf / (double)T->startup_download));
    roll_push(s->past_var, qoe3);
    if (!over)
        s->last_chunk_accuracy = T->vp_acc[(size_t)s->vp * T->n_vpchunk_max + (s->next_chunk - T->vp_start[s->vp])];
    /* when over, the reference keeps the previous chunk's size/quality/viewport in the state */
    {
        oracle_env_state tmp = *s;
        if (over) tmp.next_chunk = s->next_chunk - 1;
        write_obs(T, &tmp, action, obs);
    }
    return over;
}

int oracle_env_state_size(void) { return (int)sizeof(oracle_env_state); }
int oracle_env_tables_size(void) { return (int)sizeof(oracle_env_tables); }

/* ---------------------------------------------------------------------------------------------------------------
 * MPC expert (ExpertEnv).  Its reset/step are MANSYEnv's with reward = qoe (expert_env.py:184-330), i.e.
 * oracle_env_reset / oracle_env_step with train_identifier_reward = 0; what it adds is the per-chunk profile cache and
 * the exhaustive look-ahead search.
 * Cache arrays: [n_vp][n_vpchunk_max][N_ACTION], chunk index relative to vp_start, filled for the chunks an episode can
 * visit (startup_download + 1 .. min(vp_end, video_len - 1)), zero elsewhere.  vp_video[vp] = manifest slot of that
 * viewport trace's video. */
static void chunk_size_and_quality(const oracle_env_tables *T, int video, int chunk, const int32_t *ver, const uint8_t *gv,
                                   long long *chunk_size, float *quality, float *var) {
    /* simulator.py:146-158 + expert_env.py:162,172: exact integer size, float32 sequential sums in tile order */
    const int32_t *sz = T->size + ((size_t)video * T->n_chunk_max + chunk) * NR * NT;
    const float *ql = T->quality + ((size_t)video * T->n_chunk_max + chunk) * NR * NT;
    long long cs = 0; float tq[NT];
    for (int t = 0; t < NT; ++t) { cs += sz[ver[t] * NT + t]; tq[t] = ql[ver[t] * NT + t]; }
    float s_vq = 0.f, s_v = 0.f;
    for (int t = 0; t < NT; ++t) { s_vq += (float)gv[t] * tq[t]; s_v += (float)gv[t]; }
    const float vq = s_vq / s_v;
    float s_var = 0.f;
    for (int t = 0; t < NT; ++t) s_var += (float)gv[t] * fabsf(tq[t] - vq);
    *chunk_size = cs; *quality = vq; *var = s_var / s_v;
}

void oracle_expert_profile(const oracle_env_tables *T, const int32_t *vp_video, int n_vp, float *gt_quality, float *pred_quality,
                           float *gt_var, float *pred_var, int64_t *gt_size, int64_t *pred_size) {
    for (int vp = 0; vp < n_vp; ++vp) {
        const int video = vp_video[vp];
        int end_chunk = T->vp_end[vp];
        if (T->video_len[video] - 1 < end_chunk) end_chunk = T->video_len[video] - 1;
        for (int chunk = T->startup_download + 1; chunk <= end_chunk; ++chunk) {
            const uint8_t *gv = vp_row(T, T->vp_gt, vp, chunk), *pv = vp_row(T, T->vp_pred, vp, chunk);
            float gvf[NT], pvf[NT];
            for (int i = 0; i < NT; ++i) { gvf[i] = (float)gv[i]; pvf[i] = (float)pv[i]; }
            const size_t row = ((size_t)vp * T->n_vpchunk_max + (chunk - T->vp_start[vp])) * N_ACTION;
            for (int a = 0; a < N_ACTION; ++a) {
                int rin, rout; action2rates(a, &rin, &rout);
                int32_t ver[NT]; long long cs;
                oracle_allocate_tile_rates(rin, rout, gvf, T->video_rates, ver);
                chunk_size_and_quality(T, video, chunk, ver, gv, &cs, gt_quality + row + a, gt_var + row + a);
                gt_size[row + a] = cs;
                oracle_allocate_tile_rates(rin, rout, pvf, T->video_rates, ver);
                chunk_size_and_quality(T, video, chunk, ver, gv, &cs, pred_quality + row + a, pred_var + row + a);
                pred_size[row + a] = cs;
            }
        }
    }
}

/* expert_env.py:358-422 -- literal: every one of the action_space**horizon plans is scored over the first
 * min(horizon, chunks left) steps; the first plan with the strictly largest float32 QoE sum wins; its first action is
 * returned.  `best_value` / `best_index` (optional) expose the winning sum and plan index. */
int oracle_expert_choose(const oracle_env_tables *T, const oracle_env_state *s, int horizon_cfg, const float *pred_quality,
                         const float *pred_var, const int64_t *pred_size, float *best_value, long long *best_index) {
    long long n_plans = 1;
    for (int j = 0; j < horizon_cfg; ++j) n_plans *= N_ACTION;
    int horizon = s->end_chunk - s->next_chunk + 1;
    if (horizon_cfg < horizon) horizon = horizon_cfg;
    const double *bw = T->trace_bw + (size_t)s->trace * T->trace_len_max;
    const int tlen = T->trace_len[s->trace];
    const float *w = T->qoe_w + 3 * s->qoe;
    const float max_rate = (float)T->video_rates[NR - 1];
    const size_t row0 = ((size_t)s->vp * T->n_vpchunk_max + (s->next_chunk - T->vp_start[s->vp])) * N_ACTION;
    float best = -INFINITY; long long best_i = 0;
    for (long long i = 0; i < n_plans; ++i) {
        double cur_time = s->cur_time, buf = s->buf_size; int cur_idx = s->cur_idx;      /* record / restore */
        int has_prev = s->has_prev; float prev = s->prev_vq;
        float qoe_sum = 0.f; long long tmp = i;
        for (int t = 0; t < horizon; ++t) {
            const int a = (int)(tmp % N_ACTION); tmp /= N_ACTION;
            const size_t k = row0 + (size_t)t * N_ACTION + a;
            double size = (double)pred_size[k];
            const double start = cur_time;
            while (size > 0) {                                  /* network.py:22-35 */
                double remain = (floor(cur_time + 1) - cur_time) * bw[cur_idx];
                if (size >= remain) { cur_idx = (cur_idx + 1) % tlen; cur_time = floor(cur_time + 1); size -= remain; }
                else { cur_time += size / bw[cur_idx]; size = 0; }
            }
            const double download_time = cur_time - start;
            double rebuf = 0.0;                                 /* buffer.py:8-15 */
            if (download_time > buf) { rebuf = download_time - buf; buf = (double)T->chunk_length; }
            else buf = buf - download_time + (double)T->chunk_length;
            const float vq = pred_quality[k] / max_rate, intra = pred_var[k] / max_rate;   /* qoe.py:49-59 */
            const float inter = has_prev ? fabsf(vq - prev) : 0.0f;
            prev = vq; has_prev = 1;
            const float qoe3 = intra + inter;
            const float qoe = w[0] * vq - w[1] * (float)rebuf - w[2] * qoe3;
            qoe_sum = qoe_sum + qoe;
        }
        if (best < qoe_sum) { best = qoe_sum; best_i = i; }
    }
    if (best_value) *best_value = best;
    if (best_index) *best_index = best_i;
    return (int)(best_i % N_ACTION);      /* rates2action(action2rates(a)) == a for all 15 actions */
}
