/* libmansy_hip.so -- C ABI of the MI355X-native MANSY hot path (gfx950).
 *
 * The reference (duowuyms/MANSY_ImmersiveVideoStreaming) is pure Python and has no FFI; its seams
 * are Python classes.  Each entry point below replaces the arithmetic behind one of those seams and
 * is what the host-side mirrors in mansy_immersivevideostreaming_amd/ bind through ctypes:
 *
 *   mansy_vp_forward / mansy_vp_backward / mansy_vp_train_step / mansy_vp_sample
 *        ViewportTransformerMTIO.forward/_process_src_current/sample
 *        (viewport_prediction/models/mtio.py:65-166, models/customized_transformer.py:13-83)
 *        + the train-loop body (viewport_prediction/run_models.py:37-44)
 *   mansy_mtio_loss_fwd_bwd   ViewportTransformerMTIO.loss_function (mtio.py:94-104, utils/common.py:73-80)
 *   mansy_adamw_step          torch.optim.AdamW as used at run_models.py:29,44; Adam+L2 at run_mansy.py:216
 *   mansy_tilemap / _iou / _or_groups
 *        find_tiles_covered_by_viewport + IoU (viewport_prediction/utils/common.py:37-58,83-127;
 *        utils/results.py:13-31; predict.py:36-47)
 *
 * Conventions: plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 * (PyTorch's allocator in the shipped host code); no ownership transfer; workspaces are sized by the
 * *_workspace_bytes query; every call is asynchronous on `stream` (a hipStream_t passed as void*);
 * return 0 on success, negative MANSY_E* otherwise, message via mansy_last_error().
 */
#ifndef MANSY_HIP_H
#define MANSY_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* mansy_last_error(void);
int mansy_abi_version(void);

/* ------------------------------------------------------------------ viewport predictor */
/* Precision of the dense products (torch.nn.Linear / Conv1d arithmetic behind mtio.py, customized_transformer.py and
 * bitrate_selection/models/mansy.py) of ONE call: MANSY_PREC_F32 = exact fp32 on v_mfma_f32_32x32x2_f32 (the parity mode);
 * MANSY_PREC_BF16X3 / _BF16X6: operands split into bf16 terms, 3 / 6 bf16 MFMA products accumulated in fp32 (BASELINE.json configs[4]
 * "bf16 MFMA").  Products with K % 32 != 0 or unaligned operands stay fp32.  A launch-time property: a captured hipGraph keeps the
 * mode it was captured in.  Carried by mansy_vp_config::precision, by the `precision` argument of the PPO / A2C entry points and by
 * mansy_gemm_epilogue::prec.  ABI 8: the library holds NO process-wide mutable state -- no precision setter, no hook registration, no
 * kernel-selection knob, no environment variable; any other value than the four below is MANSY_EINVAL.  (What is process-wide and
 * read-only after first use: the error string of the calling thread, the launch counter and launch recorder of the measurement hooks.) */
#define MANSY_PREC_F32 0
#define MANSY_PREC_BF16 1      /* plain bf16: ONE bf16 MFMA product per fp32 product, fp32 accumulate (errors ~1e-3 of the operands' scale: the class of the
                                * reference's own GPU setting, torch.set_float32_matmul_precision('high'); a perf mode, never the parity mode) */
#define MANSY_PREC_BF16X3 3
#define MANSY_PREC_BF16X6 6

/* SyncBN hook (mansy_vp_config::bn_sync_fn / bn_sync_user): with bn_sync_world > 1 the engine calls fn(which, user) after enqueuing
 * the per-channel partial sums (which = 0: forward [sum, sumsq]; 1: backward [sum g, sum g*xhat]); the hook must all-reduce (SUM) the
 * 2*d_model doubles at workspace slot "dis.stats" (+ 0 / + 2*d_model doubles) over the ranks, ordered on the same stream.
 * which = 2 (mansy_vp_backward / _train_step, between the DistillLayer backward and the encoder backward): the gradients of every
 * parameter from transformer.decoder.layers.0.* to the end of the table are final -- the hook may start their all-reduce on
 * another stream while the encoder backward runs (it must not touch the stream's pending work); returning 0 without doing
 * anything is fine. */
typedef int (*mansy_bn_sync_fn)(int which, void* user);

typedef struct mansy_vp_config {
  int B, S, T;                 /* batch, history window, future window */
  int d_model, n_head, d_ff;   /* 512, 8 (nn.Transformer default), 512 */
  int n_enc, n_dec;            /* --block-num */
  int in_ch;                   /* in_channel * num_head = 6 */
  int has_bias;                /* 1: torch<=2.0 layout (biases), 0: torch>=2.1 layout (bias-free) */
  float p_pe, p_drop;          /* 0.2 (mtio.py:49), 0.1 (nn.Transformer default); 0 disables */
  float ln_eps, bn_eps, bn_momentum;
  int max_len;                 /* rows of the positional table (5000) */
  int bn_sync_world;           /* data-parallel ranks sharing DistillLayer BatchNorm statistics (<= 1: local) */
  int two_stream;              /* the decoder recurrence as two half-batches on two streams (products of one half under the attention /
                                * LayerNorm passes of the other): 0 = never, 1 = where it pays (even B >= 2048: sample() +4 %, train step
                                * +2 % at B = 4096; below that the step is a chain of latency-bound launches and the split only doubles
                                * them), 2 = wherever the halves are whole (even B >= 256).  Same function, bit-identical forward.
                                * Per-kernel timings are taken with it off (concurrent kernels stretch each other's durations). */
  int precision;               /* MANSY_PREC_* of this call's dense products */
  mansy_bn_sync_fn bn_sync_fn; /* SyncBN / gradient-ready hook of this call; NULL with bn_sync_world > 1 is MANSY_EINVAL (ABI 8: there is */
  void* bn_sync_user;          /*   no process-wide registration to fall back to) */
} mansy_vp_config;

/* ordered parameter table (names are the reference state_dict keys) */
int mansy_vp_num_params(const mansy_vp_config* cfg);
int mansy_vp_param_info(const mansy_vp_config* cfg, int idx, char* name, int name_len, long long* numel,
                        int* ndim, long long shape[4]);
size_t mansy_vp_workspace_bytes(const mansy_vp_config* cfg);
/* named activation slabs inside the workspace (for parity tests / debugging) */
int mansy_vp_ws_lookup(const mansy_vp_config* cfg, const char* name, long long* offset_bytes, long long* numel);

/* src [B,S,in_ch], cur [B,in_ch] -> pred [B,T,in_ch].  train != 0: dropout (hash RNG keyed by seed)
 * + BatchNorm batch statistics and running-stat update; train == 0: eval. */
int mansy_vp_forward(const mansy_vp_config* cfg, const float* const* params, const float* pe, float* bn_running_mean,
                     float* bn_running_var, long long* bn_num_batches, const float* src, const float* cur, float* pred,
                     void* workspace, int train, uint32_t seed, void* stream);
/* Accumulates (+=) parameter gradients of sum(pred * dpred) into grads[]; needs the workspace of the
 * preceding train-mode forward with the same seed. */
int mansy_vp_backward(const mansy_vp_config* cfg, const float* const* params, float* const* grads, const float* src,
                      const float* dpred, void* workspace, uint32_t seed, void* stream);
/* history [B,S,c], current [B,1,c] (c = in_ch/3) -> out [B,T,c]: heads replicated, eval forward,
 * 3-head mean, wrap to [0,1] (mtio.py:106-133). */
int mansy_vp_sample(const mansy_vp_config* cfg, const float* const* params, const float* pe, float* bn_running_mean,
                    float* bn_running_var, const float* history, const float* current, float* out, void* workspace,
                    void* stream);
/* One training step (run_models.py:37-44): MTIO mix (perm1/perm2 device int32[B], NULL => replicate branch),
 * zero grads, forward, MTIO loss, backward, AdamW over the flat parameter buffer.  params/grads point into
 * flat_p/flat_g.  loss_out: device float.  step <= 0 skips the AdamW update (data-parallel callers all-reduce
 * flat_g over RCCL first and then call mansy_adamw_step). */
int mansy_vp_train_step(const mansy_vp_config* cfg, const float* const* params, float* const* grads, float* flat_p,
                        float* flat_g, float* flat_m, float* flat_v, long long n_flat, const float* pe,
                        float* bn_running_mean, float* bn_running_var, long long* bn_num_batches, const float* history,
                        const float* current, const float* future, const int* perm1, const int* perm2, float lr,
                        float beta1, float beta2, float eps, float weight_decay, int step, float* loss_out,
                        void* workspace, uint32_t seed, void* stream);

int mansy_mtio_mix(const float* x, const int* perm1, const int* perm2, float* out, int B, int L, int c, void* stream);
/* loss = sum_heads mean_{B,T}( sum_xy e^2 / 2 ); dpred may be NULL.  scratch: 8 bytes device. */
int mansy_mtio_loss_fwd_bwd(const float* pred, const float* gt, int B, int T, int C, double* scratch, float* loss_out,
                            float* dpred, void* stream);
int mansy_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, int decoupled, void* stream);
int mansy_ensemble_wrap(const float* pred, float* out, long long rows, int heads, int c, void* stream);
/* LinearRegression.sample (viewport_prediction/models/linear_regression.py:18-36; `run_models.py --model regression`, :101-102): the
 * comparison baseline -- per trajectory and coordinate a least-squares line through history [B,S,c] + current [B,1,c] over
 * t = 0..S (scikit-learn LinearRegression(fit_intercept=True), float64 arithmetic), extrapolated to t = S+1..S+T -> out [B,T,c]. */
int mansy_linreg_sample(const float* history, const float* current, int B, int S, int T, int c, float* out, void* stream);

/* ViewportDataset.__getitem__ batched (viewport_prediction/utils/load_dataset.py:43-52): table [n_trace, L, c] resident
 * in HBM, idx int32 [B,2] = (trace slot, timestep) -> history [B,S,c], current [B,1,c], future [B,T,c] */
int mansy_traj_gather(const float* table, int L, int c, const int* idx, int B, int S, int T, float* hist, float* cur, float* fut,
                      void* stream);
/* mean_square_error (viewport_prediction/utils/common.py:73-80): per row of c coordinates */
int mansy_periodic_mse(const float* a, const float* b, long long rows, int c, float* out, void* stream);

/* ------------------------------------------------------------------ tile hit map
 * find_tiles_covered_by_viewport / _find_regions_covered_by_fov / find_block_covered_by_point
 * (viewport_prediction/utils/common.py:37-58, 83-127) for n points at once: xy [n,2] float32 normalised centres, pixel =
 * (int)(x * W), (int)(y * H) (float32 product, truncation toward zero: utils/results.py:15-18, predict.py:40-43); one uint64 per
 * point, bit row * tile_num_w + col.  Centres OUTSIDE the frame behave like the reference's Python: floor division on negative
 * pixels, numpy slice semantics on negative tile indices (a negative bound counts from the end of the axis). */
int mansy_tilemap(const float* xy, long long n, int W, int H, int tile_num_w, int tile_num_h, int fov_w, int fov_h,
                  uint64_t* maps, void* stream);
int mansy_tilemap_iou(const uint64_t* a, const uint64_t* b, long long n, double* iou, void* stream);
int mansy_tilemap_or_groups(const uint64_t* maps, long long ngroups, int group, uint64_t* out, void* stream);
/* accuracy (IoU), recall, precision, f1 per point (viewport_prediction/utils/results.py:21-31): out double [n,4] */
int mansy_tilemap_metrics(const uint64_t* gt, const uint64_t* pred, long long n, double* out, void* stream);

/* ------------------------------------------------------------------ vectorised streaming environment
 * Replaces MANSYEnv.reset/step (bitrate_selection/envs/mansy_env.py:99-248) + Simulator/NetworkTrace/PlaybackBuffer/
 * HMDTrace (bitrate_selection/simulators/*.py) + QoEModel.calculate_qoe (utils/qoe.py:22-34) + action2rates /
 * allocate_tile_rates (utils/common.py:101-193) for N environments at once.  One observation = one row of
 * MANSY_OBS_LD floats with the reference's dict keys at fixed offsets: */
#define MANSY_OBS_DIM 779
#define MANSY_OBS_LD 780
#define MANSY_O_THROUGHPUT 0   /* 'throughput'               [8]   */
#define MANSY_O_SIZE 8         /* 'next_chunk_size'          [5,64] */
#define MANSY_O_QUALITY 328    /* 'next_chunk_quality'       [5,64] */
#define MANSY_O_PRED_VP 648    /* 'pred_viewport'            [64]  */
#define MANSY_O_VP_ACC 712     /* 'viewport_acc'             [8]   */
#define MANSY_O_PAST_Q 720     /* 'past_viewport_qualities'  [8]   */
#define MANSY_O_PAST_VAR 728   /* 'past_quality_variances'   [8]   */
#define MANSY_O_PAST_REBUF 736 /* 'past_rebuffering'         [8]   */
#define MANSY_O_BUFFER 744     /* 'buffer'                   [1]   */
#define MANSY_O_QOE_W 745      /* 'qoe_weight'               [3]   */
#define MANSY_O_ACT_1HOT 748   /* 'action_one_hot'           [15]  */
#define MANSY_O_RATES_IN 763   /* 'rates_inside'             [8]   */
#define MANSY_O_RATES_OUT 771  /* 'rates_outside'            [8]   */

typedef struct mansy_env_tables {
  const int32_t* size; const float* quality; const int32_t* video_len; int n_chunk_max;     /* [n_video][n_chunk_max][5][64] */
  const uint8_t* vp_gt; const uint8_t* vp_pred; const double* vp_acc;                        /* [n_vp][n_vpchunk_max][64] / [..] */
  const int32_t* vp_start; const int32_t* vp_end; int n_vpchunk_max;
  const double* trace_bw; const int32_t* trace_len; int trace_len_max;                       /* [n_trace][trace_len_max] bytes/s */
  const int32_t* samples; int n_sample; const float* qoe_w;                                  /* [n_sample][4], [n_qoe][3] */
  int video_rates[5]; int startup_download; int chunk_length; double max_size; double max_throughput;
  int train_identifier_reward;   /* 1: reward = qoe / sum(w) (mode == 'train' and use_identifier, mansy_env.py:168-177) */
} mansy_env_tables;
/* finished-episode records: 8 doubles each = (sample_id, env, n_steps, sum qoe, sum qoe1, sum qoe2, sum qoe3, qoe index) */
typedef struct mansy_env_episode_log { double* records; unsigned int* count; int capacity; } mansy_env_episode_log;

int mansy_env_state_bytes(void);
/* env i starts at sample (seed + index_offset + i) % worker_num and strides by worker_num (mansy_env.py:55-56,100-101) */
int mansy_env_init(void* state, int n_env, int index_offset, int worker_num, int seed, void* stream);
int mansy_env_reset(const mansy_env_tables* T, void* state, int n_env, float* obs, void* stream);
/* obs_next: state after the action (terminal observation when done); obs_cur (optional): same, except that finished
 * environments are reset and show the first observation of their next episode (vector-env auto-reset). */
int mansy_env_step(const mansy_env_tables* T, void* state, int n_env, const int* actions, float* obs_next, float* obs_cur,
                   float* reward, unsigned char* done, float* qoe_parts, const mansy_env_episode_log* elog, void* stream);
/* pred_viewport [n,64] (0/1 floats), actions [n] -> rate version per tile [n,64] */
int mansy_allocate_tile_rates(const float* pred_viewport, const int* actions, int n, const int video_rates[5], int* versions,
                              void* stream);

/* ------------------------------------------------------------------ MPC expert (demonstrations for behaviour cloning)
 * Replaces ExpertEnv._profile_viewport_qualities_sizes and ExpertEnv.choose_action (bitrate_selection/envs/
 * expert_env.py:126-181, 358-422) + ExpertSimulator.virtual_simulate_download_with_chunk_size /
 * calculate_chunk_size_and_quality (simulators/simulator.py:127-158) + QoEModelExpert.calculate_qoe_with_given_quality
 * (utils/qoe.py:49-59).  ExpertEnv.reset/step are mansy_env_reset/_step with train_identifier_reward = 0.
 * Cache arrays are [n_vp][n_vpchunk_max][15] (chunk index relative to vp_start; entries of chunks no episode visits are
 * zero); vp_video[vp] = manifest slot of that viewport trace's video.  Chunk sizes are int32 (64 tiles x < 2^25 bytes). */
#define MANSY_EXPERT_MAX_HORIZON 6
int mansy_expert_profile(const mansy_env_tables* T, const int* vp_video, int n_vp, float* gt_quality, float* pred_quality, float* gt_var,
                         float* pred_var, int* gt_size, int* pred_size, void* stream);
/* For each of n_env environments (state = the mansy_env_* records): first action of the best of the 15^horizon plans
 * (first plan with the strictly largest float32 QoE sum).  keys: n_env uint64 workspace.  best_value / best_index
 * (optional): winning score and plan index. */
int mansy_expert_choose_action(const mansy_env_tables* T, const void* state, int n_env, int horizon, const float* pred_quality,
                               const float* pred_var, const int* pred_size, unsigned long long* keys, int* actions, float* best_value,
                               long long* best_index, void* stream);

/* ------------------------------------------------------------------ bitrate-selection networks + PPO
 * Replaces FeatureNet/Actor/Critic/QoEIdentifier.forward (bitrate_selection/models/mansy.py:26-155),
 * calculate_indentifier_reward / train_identifier (utils/mansy_utils.py:9-49), the relabel loop (models/mansy_ppo.py:41-51)
 * and tianshou==0.4.8's PPOPolicy.process_fn/learn arithmetic behind mansy_ppo.py:53-55.
 * kind 0 = actor-critic (28 unique tensors: shared feature net, actor head, critic head), kind 1 = identifier (24).
 * `params`/`grads`: device pointers in mansy_net_param_info order.  obs rows are MANSY_OBS_LD floats. */
int mansy_net_num_params(int kind);
int mansy_net_param_info(int kind, int idx, char* name, int name_len, long long* numel, int* ndim, long long shape[4]);
size_t mansy_ppo_workspace_bytes(int max_batch);
/* logits [B,16] (15 used; nullable), value [B] (nullable), optional Categorical sampling: act int32 [B], logp [B];
 * u [B] external uniforms in [0,1) or NULL => counter hash (seed, site, row) */
int mansy_policy_forward(const float* const* params, const float* obs, int B, float* logits, float* value, int* act, float* logp,
                         const float* u, uint32_t seed, uint32_t site, int reuse_packed /* 1: parameters unchanged since the
                         previous policy call on this workspace */, void* workspace, int max_batch, int precision, void* stream);
/* One rollout step as ONE call and one launch fewer: mansy_policy_forward (actor only, sampling on) for the n_env observation rows,
 * then mansy_env_step of every environment with the action just drawn, inside the same output-layer launch (the wave that sampled
 * row e goes on to step environment e).  Same outputs as the two calls. */
int mansy_policy_env_step(const float* const* params, const float* obs, int n_env, float* logits, int* act, float* logp, const float* u,
                          uint32_t seed, uint32_t site, int reuse_packed, void* workspace, int max_batch, const mansy_env_tables* T,
                          void* env_state, float* obs_next, float* obs_cur, float* reward, unsigned char* done, float* qoe_parts,
                          const mansy_env_episode_log* elog, int precision, void* stream);
/* A whole collect -- T vector steps of mansy_policy_env_step -- as ONE persistent launch (round 5): the three launches of a step become three phases of a
 * kernel whose workgroups (one per CU) form one team per XCD and synchronise through that XCD's L2; the environments are dealt to the teams in chunks of
 * 32 and never leave them.  obs_slab [T][n_env][MANSY_OBS_LD]: block 0 = the observations the collect starts from, block t + 1 <- step t's auto-reset
 * observation, `carry` <- the last one; u / act / logp / reward / done: [T][n_env]; obs_next_slab [T][n_env][MANSY_OBS_LD].  Bit-identical to T calls of
 * mansy_policy_env_step.  ctl: MANSY_ROLLOUT_CTL_BYTES of device memory (zeroed by the call); err_host: a host-MAPPED int (hipHostMalloc) that the kernel
 * sets to 1 if a workgroup or team-mate never showed up within 2 s (a shared device): the outputs are then garbage and the caller must raise.
 * Returns MANSY_EINVAL ("rollout_team: ...") where the form does not apply (precision != fp32, n_env whose products do not run on the wave-split-K loop,
 * launch recorder on): take the per-step call then. */
#define MANSY_ROLLOUT_CTL_BYTES 1024
int mansy_policy_rollout(const float* const* params, float* obs_slab, int n_env, int T, const float* u, int* act, float* logp, float* obs_next_slab,
                         float* carry, float* reward, unsigned char* done, float* qoe_parts, const mansy_env_tables* tables, void* env_state,
                         const mansy_env_episode_log* elog, int reuse_packed, void* ctl, int* err_host, void* workspace, int max_batch, int precision,
                         void* stream);
/* logits' log-probabilities of the given actions for the first n_logp rows (logp != NULL) and / or the critic's value for all B rows.
 * process_fn calls it once on the 2 x 4096 rows [obs ; obs_next] of one rollout buffer: v_s, v_s_ and logp_old in one pass. */
int mansy_policy_evaluate(const float* const* params, const float* obs, int B, const int* act, int n_logp, float* logp, float* value,
                          void* workspace, int max_batch, int precision, void* stream);
int mansy_identifier_forward(const float* const* params, const float* obs, int B, float* pred /* [B,16], 3 used */, void* workspace,
                             int max_batch, int precision, void* stream);
int mansy_identifier_train_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_m,
                                float* flat_v, long long n_flat, const float* obs, const int* idx /* NULL: rows 0..B of obs; else obs[idx[r]] */,
                                int B, float lr, float weight_decay, int step, float* loss_out, void* workspace, int max_batch,
                                void* xg_ctx /* NULL, or (step > 0) a mansy_xg context: see mansy_ppo_minibatch_step */,
                                const float* adam_bias /* NULL, or device [2]: see mansy_ppo_minibatch_step */, int precision, void* stream);
int mansy_identifier_relabel(const float* const* params, const float* obs, float* rew, float* id_rew, int B, float lamb,
                             void* workspace, int max_batch, int precision, void* stream);
int mansy_gae_returns(const float* rew, const float* v_s, const float* v_next, const unsigned char* done, int T, int N, double gamma,
                      double gae_lambda, int rew_norm, double* rms, double* scratch, float* returns, float* adv, void* stream);
int mansy_ppo_minibatch_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_m,
                             float* flat_v, long long n_flat, const float* obs_all, const int* idx, const int* act_all,
                             const float* adv_all, const float* logp_old_all, const float* v_old_all, const float* ret_all, int mb,
                             float eps_clip, float vf_coef, float ent_coef, int norm_adv, int value_clip, float dual_clip,
                             float max_grad_norm, float lr, float weight_decay, int step, long long tail_from, int tail_step, float* stats,
                             void* workspace, int max_batch, int chain_in, const int* next_idx, int next_mb, void* xg_ctx, const float* adam_bias,
                             int precision, void* stream);
/* adam_bias (ABI 9; nullable): device pointer to two floats [1 - 0.9^t, sqrt(1 - 0.999^t)] that the Adam launch READS instead of the values the
 * host derives from `step` -- what makes the call replayable from a captured hipGraph: the kernel arguments of a replay are frozen, the caller
 * rewrites the two floats (stream-ordered) before every replay.  `step` still selects the step's form and its scratch parity: a graph must hold
 * an EVEN number of steps per flat buffer.  Not with a lagged tail.  Accepted by mansy_identifier_train_step, mansy_ppo_dp_tail and
 * mansy_clip_grad_adam as well. */
/* xg_ctx = the step's `sync` context (ABI 8; NULL = single process) -- the DATA-PARALLEL step as ONE call.  Needs the clipped chained form
 * (max_grad_norm > 0, step > 0, no lagged tail); every rank makes the same sequence of calls on its context.
 *  - a mansy_xg context of n_flat floats (MANSY_SYNC_XG): the rank's raw gradients are produced straight in its exchange slot, ONE more launch
 *    than the single-process step publishes the slot, waits (bounded) for every peer's, sums all ranks' slots in rank order into flat_g and
 *    leaves the sums of squares of the average; clip + Adam + the chained prologue follow as in the single-process step;
 *  - a communicator context (MANSY_SYNC_RCCL, mansy_comm_create): raw gradients in flat_g, ncclAllReduce(avg) on the stream, a gradient-norm
 *    launch, then the same tail.
 * (The three-call form stays for callers with a collective of their own: step = 0 here, their all-reduce, mansy_ppo_dp_tail.) */
/* Layout contract of the chained forms (step > 0 with max_grad_norm > 0, and mansy_ppo_dp_tail): their last launch updates four consecutive
 * elements per thread and scatters them into the packed images, so flat_p / flat_g / flat_m / flat_v must be 16-byte aligned and every
 * params[k] must be an ascending view of flat_p that starts at a multiple of 4 floats (the host mirror aligns tensors to 256 bytes);
 * anything else is MANSY_EINVAL. */
/* dual_clip: tianshou PPOPolicy's dual_clip (> 1; for negative advantages the clipped surrogate is bounded below by dual_clip * adv,
 * run_mansy.py --dual-clip) or 0 = off (the reference's default None). */
/* Chaining (the clipped single-process step only: max_grad_norm > 0, step > 0, no lagged tail): the step's last launch -- clip + Adam --
 * also zeroes the gradient buffer, writes the updated parameters into the packed images the next forward reads and, when next_mb > 0,
 * gathers the rows next_idx[0..next_mb) of obs_all and takes the statistics of their advantages; the NEXT call on the same workspace /
 * stream / parameters for exactly that minibatch passes chain_in = 1 and skips its prologue launch (16 steps of PPOPolicy.learn:
 * 1 + 16 x 9 launches instead of 16 x 10).  chain_in = 0, next_mb = 0: the self-contained step.  After a step with step > 0 flat_g is
 * all zeros. */
/* Behaviour cloning on expert demonstrations (behavior_cloning_pretraining, utils/mansy_utils.py:52-93): loss =
 * CrossEntropy(actor logits, act) - ent_coef * mean entropy; Adam(L2) over the first n_update elements of the flat buffers
 * (the critic head -- the tail -- has no gradient here and torch.optim.Adam skips it).  step <= 0: forward + loss only.
 * stats: [loss, cross entropy, mean entropy]. */
int mansy_bc_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_m, float* flat_v,
                  long long n_flat, long long n_update, const float* obs, const int* act, int B, float ent_coef, float lr,
                  float weight_decay, int step, float* stats, void* workspace, int max_batch, int precision, void* stream);

/* clip_grad_norm_ (max_norm <= 0: off) + Adam with L2 weight decay over flat buffers (data-parallel second half).
 * scratch: MANSY_CLIP_SCRATCH_DOUBLES doubles of device memory (gradient-norm partial sums; need not be zeroed).
 * tail_from >= 0 (here and in mansy_ppo_minibatch_step): elements [tail_from, n_flat) use Adam step count tail_step
 * instead of step -- torch keeps one counter per parameter, and the critic head sees its first gradient only after the
 * behaviour-cloning steps; pass -1 / 0 otherwise. */
#define MANSY_CLIP_SCRATCH_DOUBLES 64
/* have_sumsq != 0: scratch already holds the MANSY_CLIP_SCRATCH_DOUBLES partial sums of squares of flat_g (mansy_xg_allreduce_avg
 * leaves them there) -- the norm launch is skipped. */
int mansy_clip_grad_adam(float* flat_p, float* flat_g, float* flat_m, float* flat_v, long long n_flat, float max_grad_norm, float lr,
                         float weight_decay, int step, long long tail_from, int tail_step, double* scratch, int have_sumsq,
                         const float* adam_bias /* nullable: see mansy_ppo_minibatch_step */, void* stream);

/* Data-parallel form of the chained step's last launch (after the ranks averaged the raw gradients of a step = 0 call): clip by the
 * global norm, Adam(L2), zero flat_g, re-pack the updated parameters into the workspace's packed images and prepare the next
 * minibatch (next_mb > 0) -- the next mansy_ppo_minibatch_step(step = 0) then passes chain_in = 1.  Actor-critic buffers only. */
/* next_flat_g (ABI 8; nullable = flat_g): the buffer the NEXT mansy_ppo_minibatch_step(step = 0) will use as its flat_g / grads[] -- zero-filled
 * here instead of flat_g.  The peer-memory form (mansy_xg_reduce_avg) produces each step's gradients in one of the rank's two exchange slots and
 * averages them into flat_g; the next step's gradients go into the other slot. */
int mansy_ppo_dp_tail(const float* const* params, float* flat_p, float* flat_g, float* flat_m, float* flat_v, long long n_flat,
                      float max_grad_norm, float lr, float weight_decay, int step, double* scratch, int have_sumsq, const float* obs_all,
                      const float* adv_all, const int* next_idx, int next_mb, float* next_flat_g, const float* adam_bias /* nullable */, void* workspace,
                      int max_batch, int precision, void* stream);

/* ------------------------------------------------------------------ thin wrappers over RCCL communicators (round 5)
 * The three collectives of the data-parallel hot path as C-ABI calls: explicit communicator, explicit stream, device pointers.  RCCL is bound at
 * run time (dlopen of librccl.so -- torch's copy when the process already holds one), so the library has no link-time dependency on it.
 * mansy_comm_unique_id on rank 0 -> the host moves the 128 bytes to every rank -> mansy_comm_create on every rank (collective) -> any number of
 * collectives, the same sequence on every rank -> mansy_comm_destroy.  A communicator context is also a `sync` context of
 * mansy_ppo_minibatch_step / mansy_identifier_train_step (below), like the peer-memory context of the next section. */
typedef struct mansy_comm_id { unsigned char bytes[128]; } mansy_comm_id;
int mansy_comm_unique_id(mansy_comm_id* out);
int mansy_comm_create(const mansy_comm_id* id, int world, int rank, void** comm_out);
int mansy_comm_destroy(void* comm);
int mansy_allreduce_avg_f32(void* comm, float* buf /* in place */, long long n, void* stream);      /* flat gradient buffers (ncclAvg) */
int mansy_allreduce_sum_f64(void* comm, double* buf /* in place */, long long n, void* stream);     /* SyncBN statistics (the hook of mansy_vp_config) */
int mansy_allgather_f64(void* comm, const double* send, double* recv /* [world][n_per_rank] */, long long n_per_rank, void* stream);   /* return normaliser */
/* first int of every sync context (what the engine entry points dispatch on) */
#define MANSY_SYNC_XG 0x5847      /* peer-memory context (mansy_xg_create) */
#define MANSY_SYNC_RCCL 0x5243    /* RCCL communicator context (mansy_comm_create) */

/* ------------------------------------------------------------------ one-shot gradient all-reduce over peer-mapped memory (xGMI)
 * The data-parallel PPO update averages a 1.7 MB / 1.05 MB flat gradient 16 + 2 times per 2.6 ms cycle, every time on
 * the critical path.  Instead of a library all-reduce launch + a separate gradient-norm launch, each rank runs ONE kernel that
 * publishes its gradient in fine-grained device memory the peers have mapped (hipIpc), waits -- bounded -- for every peer's epoch
 * flag, sums all ranks' copies in rank order straight over the point-to-point links, writes the average over g and leaves the
 * partial sums of squares for mansy_clip_grad_adam(have_sumsq = 1).  Every rank computes bit-identical averages.
 * Protocol: every rank mansy_xg_create(n, world, rank) -> mansy_xg_export(handle) -> the host exchanges the 64-byte handles
 * (torch.distributed all_gather_object) -> mansy_xg_import(all handles, rank order) -> any number of mansy_xg_allreduce_avg calls,
 * one per gradient, the same sequence on every rank.  A wait that does not meet its peers within the timeout (default 2 s)
 * poisons the call's output with NaN and raises a sticky error (mansy_xg_status); it never hangs. */
typedef struct mansy_xg_handle { unsigned char bytes[64]; } mansy_xg_handle;
int mansy_xg_create(long long n_floats, int world, int rank, void** ctx_out);
int mansy_xg_export(void* ctx, mansy_xg_handle* out);
int mansy_xg_import(void* ctx, const mansy_xg_handle* all /* [world], rank order */);
int mansy_xg_set_timeout_ms(void* ctx, double ms);
int mansy_xg_allreduce_avg(void* ctx, float* g /* in place */, long long n, double* sumsq_parts /* [MANSY_CLIP_SCRATCH_DOUBLES] or NULL */,
                           void* stream);
/* Round-5 form, no copy in front of the flag: the rank's two exchange slots ARE its flat gradient buffers.  The gradient kernels of a step use
 * slot mansy_xg_next_slot(ctx) (0 / 1; pointers from mansy_xg_slot_ptrs, n floats each, 16-byte aligned, fine-grained device memory) as flat_g /
 * grads[]; mansy_xg_reduce_avg then publishes that slot, waits (bounded) for every peer's, sums all ranks' slots in rank order and writes the
 * average to g_out (ordinary device memory) + the partial sums of squares; the slots alternate with every call.  Measured on one MI355X at
 * world 1 (everything but wire time): DESIGN.md section 6. */
int mansy_xg_slot_ptrs(void* ctx, float** slot0, float** slot1);
int mansy_xg_next_slot(void* ctx);
int mansy_xg_reduce_avg(void* ctx, float* g_out, long long n, double* sumsq_parts /* [MANSY_CLIP_SCRATCH_DOUBLES] or NULL */, void* stream);
int mansy_xg_status(void* ctx);
int mansy_xg_destroy(void* ctx);

/* ------------------------------------------------------------------ A2C baseline ("simple RL", the comparison agent)
 * Replaces FeatureNet/Actor/Critic.forward (bitrate_selection/models/simple_rl.py:9-63), SimpleRLEnv's observation
 * (envs/simple_rl_env.py:85-170: reset/step are MANSYEnv's simulator + QoE with a five-key observation) and
 * tianshou==0.4.8's A2CPolicy.learn + torch.optim.RMSprop behind run_simple_rl.py:190-211.  Returns / advantages:
 * mansy_gae_returns (A2CPolicy._compute_returns is the function PPO inherits).
 * One observation = one row of MANSY_A2C_OBS_LD floats: */
#define MANSY_A2C_OBS_LD 416
#define MANSY_A2C_O_THROUGHPUT 0   /* 'throughput'    [1,8]  */
#define MANSY_A2C_O_SIZE 8         /* 'chunk_sizes'   [5,64] */
#define MANSY_A2C_O_REBUFFER 328   /* 'rebuffer'      [1]    */
#define MANSY_A2C_O_LAST_RATES 329 /* 'last_bitrates' [2]    */
#define MANSY_A2C_O_PRED_VP 331    /* 'pred_viewport' [64]   (395..415: zero padding) */
/* 18 unique tensors in state_dict order: shared feature net (10), actor head (4), critic head (4) */
int mansy_a2c_num_params(void);
int mansy_a2c_param_info(int idx, char* name, int name_len, long long* numel, int* ndim, long long shape[4]);
size_t mansy_a2c_workspace_bytes(int max_batch);
/* rows of the five-key observation from the outputs of mansy_env_reset / mansy_env_step for the same step: obs [n,780],
 * qoe_parts [n,4] and actions [n] of that step (actions == NULL: observations right after reset), fresh [n] (optional):
 * rows whose environment was auto-reset (rebuffer / last_bitrates zero). */
int mansy_a2c_obs(const float* obs, const float* qoe_parts, const int* actions, const unsigned char* fresh, int n, const int video_rates[5],
                  float* out, void* stream);
/* probs [B,16] (15 used: the reference's Actor returns softmax outputs as "logits"), value [B] (nullable), optional
 * Categorical(probs) sampling: act int32 [B], logp [B]; u [B] external uniforms or NULL => counter hash (seed, site, row) */
int mansy_a2c_forward(const float* const* params, const float* obs, int B, float* probs, float* value, int* act, float* logp, const float* u,
                      uint32_t seed, uint32_t site, int reuse_packed, void* workspace, int max_batch, int precision, void* stream);
/* loss = -(log_prob * adv).mean() + vf_coef * mse(ret, value) - ent_coef * entropy.mean(); clip_grad_norm_; RMSprop(lr, alpha,
 * eps) over flat buffers.  apply == 0: gradients only.  stats: [loss, actor loss, value loss, entropy]. */
int mansy_a2c_minibatch_step(const float* const* params, float* const* grads, float* flat_p, float* flat_g, float* flat_sq, long long n_flat,
                             const float* obs_all, const int* idx, const int* act_all, const float* adv_all, const float* ret_all, int mb,
                             float vf_coef, float ent_coef, float max_grad_norm, float lr, float alpha, float eps, int apply, float* stats,
                             void* workspace, int max_batch, int precision, void* stream);
/* data-parallel second half: clip_grad_norm_ + RMSprop over (all-reduced) flat gradients; scratch as mansy_clip_grad_adam */
int mansy_clip_grad_rmsprop(float* flat_p, float* flat_g, float* flat_sq, long long n_flat, float max_grad_norm, float lr, float alpha,
                            float eps, double* scratch, void* stream);

/* ------------------------------------------------------------------ single kernels (unit-test surface) */
typedef struct mansy_gemm_epilogue {
  const float* bias; int relu; const float* mask_src; int mask_ld; float mask_scale;
  float drop_p; uint32_t drop_seed; uint32_t drop_site; const float* resid; int resid_ld; int accumulate;
  /* bias-gradient rider of a dW product (a_kmajor != 0): a_rowsum[m] += sum_k A[m][k] (zero it first); the `+= dy.sum(0)` of a
   * Linear's bias gradient at mtio.py / mansy.py autograd.  prec: MANSY_PREC_* of this product.
   * SINGLE WRITER: `accumulate` and `a_rowsum` update C / a_rowsum with plain read-add-write on the small-product
   * loops (one owner per block, no atomics): two accumulating products into the SAME C or a_rowsum must be ordered on one stream. */
  float* a_rowsum; int prec;
  /* kernel-selection override of THIS call (0 = defaults; diagnostics / parity tests: the same product on two loops).  OR of
   * MANSY_VARIANT_* below.  Per call, thread-safe: nothing about it is remembered. */
  int variant;
} mansy_gemm_epilogue;
#define MANSY_VARIANT_BF16(v) (((v) + 1) & 0xFF) /* loop variant v of the bf16x3 products with pre-split weights (0, 1 default, 4, 7, 8: real loops with
                                                  * bit-identical results -- csrc/gemm_bf16s.hip; any other code runs the default.  Round 6: the timing-only
                                                  * forms 2 / 3 / 11 / 12, whose results were wrong, no longer exist) */
#define MANSY_VARIANT_NO_WSK 0x100               /* fp32 products too small to fill the chip: the 64 x 64 LDS-DMA loop instead of the wave-split-K loop
                                                  * (same products, different summation order) */
#define MANSY_VARIANT_NO_WSK_TN 0x200            /* the same for the small weight-gradient (TN) products only */
#define MANSY_VARIANT_NO_PLAIN 0x400             /* no compile-time "plain" instance of the LDS-DMA loop (bit-identical) */
#define MANSY_VARIANT_COL_GROUP(g) ((((g) + 1) & 0xFF) << 16) /* column-group width g of the XCD-aware tile order (default 12; 0 = row-panel-major); tile ORDER only */
int mansy_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor, float* C, int ldc,
                   int M, int N, int K, const mansy_gemm_epilogue* ep, int force_tile, int force_splitk, void* stream);
/* Split-bf16 modes, weights split ahead of the products (what the viewport engine does once per step for every Linear / Conv1d weight):
 * mansy_weight_planes writes n_planes (2: bf16x3, 3: bf16x6) bf16 planes of W [N, K] (plane t at out + t * plane_stride, row-major [N, K])
 * and of its transpose (out_t + t * plane_stride, row-major [K, N]); mansy_gemm_planes is mansy_gemm_f32 with a K-contiguous A and those
 * planes as the B operand (planes of W for C = A W^T, planes of W^T -- planes_ld = N of W -- for C = A W): in the split modes B then
 * reaches LDS by LDS-DMA, with no split work in the loop.  B / ldb / b_kmajor describe the same operand in fp32 (used in fp32 mode). */
int mansy_weight_planes(const float* W, int N, int K, uint16_t* out, uint16_t* out_t, long long plane_stride, int n_planes, void* stream);
int mansy_gemm_planes(const float* A, int lda, const float* B, int ldb, int b_kmajor, const uint16_t* planes, long long plane_stride,
                      int planes_ld, float* C, int ldc, int M, int N, int K, const mansy_gemm_epilogue* ep, int force_tile, void* stream);
/* bf16-STORAGE product (round 6; the MANSY_PREC_BF16 perf mode's products when its operands live in HBM as bf16): both operands are bf16 images,
 * staged by LDS-DMA without conversion, one v_mfma_f32_32x32x16_bf16 product, fp32 accumulate.  Two forms:
 *   a_kmajor = b_kmajor = 0: C[M, N] = A16[M, K] * B16[N, K]^T (forward with B16 = bf16(W), dX with B16 = bf16(W^T)); full fused epilogue; the output
 *     goes to C (fp32, nullable) and / or C16 (bf16 image of the final value, nullable);
 *   a_kmajor = b_kmajor = 1: C[M, N] += A16[K, M]^T * B16[K, N] (weight gradient dW = dY^T X over the K rows of two activation slabs, split-K with
 *     atomic accumulation; ep->accumulate must be set, ep->a_rowsum as in mansy_gemm_f32).
 * resid16 / mask16 (forward form, nullable): bf16 images of the residual / the mask source, read INSTEAD of ep->resid / ep->mask_src with the same leading
 * dimensions ep->resid_ld / ep->mask_ld (the bf16-storage mode's residual streams and ReLU masks are bf16 images: csrc/vp_engine.hip).
 * force_tile: 0 = by shape; forward form: 64, 96 (128 x 64), 128 = that tile; weight-gradient form: 64 = four-wave workgroups (two per CU) instead of the
 * eight-wave ones with two K groups (same sums in another order).  force_splitk (weight-gradient form): number of K splits, 0 = by shape.
 * K % 64 == 0, 16-byte aligned operands, leading dimensions % 8 == 0.  The reference's analogue: torch.set_float32_matmul_precision('high')
 * (viewport_prediction/run_models.py:135). */
int mansy_gemm_bf16(const uint16_t* A16, int lda, int a_kmajor, const uint16_t* B16, int ldb, int b_kmajor, float* C, int ldc, uint16_t* C16, int ldc16,
                    int M, int N, int K, const mansy_gemm_epilogue* ep, const uint16_t* resid16, const uint16_t* mask16, int force_tile, int force_splitk, void* stream);
typedef struct mansy_attn_shape {
  int nb, H, Lq, Lk, dh;
  long long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  float scale;
} mansy_attn_shape;
int mansy_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* P_save, const mansy_attn_shape* s,
                   float drop_p, uint32_t seed, uint32_t site, void* stream);
int mansy_attn_bwd(const float* Q, const float* K, const float* V, const float* P_save, const float* dO, float* dQ,
                   float* dK, float* dV, const mansy_attn_shape* s, float drop_p, uint32_t seed, uint32_t site,
                   int accum_kv, void* stream);
int mansy_layernorm_fwd(const float* a, const float* b, const float* w, const float* bias, float* z_out, float* y,
                        float* mean, float* rstd, int rows, int C, float eps, void* stream);
int mansy_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* w, float* dz,
                        float* dz_drop, float drop_p, uint32_t seed, uint32_t site, float* dw, float* dbias, int rows,
                        int C, void* stream);

/* Engine forms of the two kernels above (same arithmetic, restructured traffic), exposed for the parity tests:
 *  - LayerNorm backward with per-workgroup partial weight-gradient sums: partials[mansy_layernorm_bwd_parts(rows)][2][C]
 *    (overwritten, or added to when accumulate != 0); mansy_ln_partials_reduce adds them into dw / dbias (either may be null).
 *  - Lq == 1 attention backward with deferred K/V gradients: per step mansy_attn_bwd_dq writes dQ and the step's
 *    coefficients dS_out / Pk_out [nb*H, Lk]; mansy_attn_kvgrad forms dK[j] = sum_i dS_i[j] q_i, dV[j] = sum_i Pk_i[j] dO_i
 *    from the T steps (Q_all / dO_all: step i at + i*q_ts / + i*o_ts floats; dS_all / Pk_all: [T][nb*H][Lk]). */
int mansy_layernorm_bwd_parts(int rows);
int mansy_layernorm_bwd_partial(const float* dy, const float* z, const float* mean, const float* rstd, const float* w, float* dz,
                                float* dz_drop, float drop_p, uint32_t seed, uint32_t site, float* partials, int accumulate,
                                int rows, int C, void* stream);
int mansy_ln_partials_reduce(const float* partials, int nparts, int C, float* dw, float* dbias, void* stream);
int mansy_attn_bwd_dq(const float* Q, const float* K, const float* V, const float* P_save, const float* dO, float* dQ,
                      float* dS_out, float* Pk_out, const mansy_attn_shape* s, float drop_p, uint32_t seed, uint32_t site,
                      void* stream);
int mansy_attn_kvgrad(const float* Q_all, long long q_ts, const float* dO_all, long long o_ts, const float* dS_all,
                      const float* Pk_all, float* dK, float* dV, const mansy_attn_shape* s, int T, int accum, void* stream);
/* KV-cached self-attention backward, "pull" form (steps T-1 -> 0; s->Lk == step + 1): writes dQ of the step, its coefficient
 * rows in dS_all / Pk_all [T][nb*H][T], and row `step` of the K/V gradient slabs complete (own term + later steps' terms). */
int mansy_attn_bwd_selfpull(const float* Q_all, long long q_ts, const float* K, const float* V, const float* P_save,
                            const float* dO_all, long long o_ts, float* dQ, float* dK, float* dV, float* dS_all, float* Pk_all,
                            const mansy_attn_shape* s, int T, int step, float drop_p, uint32_t seed, uint32_t site, void* stream);

/* ------------------------------------------------------------------ measurement hooks (bench.py) */
/* A HIP event pair attached to every GEMM dispatch on its own stream (the kernel's begin / end); collect() = device sync + summed ms, count, FLOPs. */
int mansy_prof_gemm_enable(int on);
int mansy_prof_gemm_collect(double* total_ms, long long* launches, double* flops);
/* (ABI 8: the round-4 A/B knobs mansy_gemm_bf16_variant / mansy_gemm_col_group / mansy_gemm_f32_wsk are gone from the library.  One product's
 * loop can be chosen per call through mansy_gemm_epilogue::variant; whole-engine A/B timings use a -DMANSY_LAB build of the same sources
 * (libmansy_hip_lab.so, tools/ only), which alone exports mansy_lab_set_variant.) */
/* Kernel launches this library has enqueued since the process started (every launch site counts; a launch enqueued during a hipGraph
 * capture counts once, at capture time -- a replay of the graph adds nothing).  bench.py reads the difference around a cycle. */
unsigned long long mansy_prof_launch_count(void);

#ifdef __cplusplus
}
#endif
#endif
