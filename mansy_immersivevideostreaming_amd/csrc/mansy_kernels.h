// Internal C++ launch API shared by the engine (vp_engine.hip, ppo_engine.hip) and the
// C-ABI test shims (capi.hip).  Every function enqueues on `st` and returns a MANSY_* status.
#pragma once
#include "mansy_common.h"

// ---------------------------------------------------------------- GEMM (gemm_f32.hip)
// C[M,N] (+)= A[M,K] * B[K,N] in exact fp32 on v_mfma_f32_32x32x2_f32.
//   a_kmajor == 0: A(m,k) at A[m*lda + k]      (row-major [M,K],  "K-contiguous")
//   a_kmajor == 1: A(m,k) at A[k*lda + m]      (row-major [K,M],  i.e. A^T stored)
//   b_kmajor == 0: B(k,n) at B[n*ldb + k]      (row-major [N,K],  torch Linear weight)
//   b_kmajor == 1: B(k,n) at B[k*ldb + n]      (row-major [K,N])
// Epilogue order: +pre_a[m, n - pre_col0] (+pre_b) for n >= pre_col0 -> +bias[n] -> (leaky) relu
//                 -> mask (mask_src[m,n] > 0 ? v*mask_scale : v*mask_neg) -> dropout(site, idx = m*N+n) -> +resid[m,n]
//                 -> store / atomicAdd.
struct GemmEpilogue {
  const float* bias = nullptr;
  int relu = 0;
  float relu_slope = 0.f;   // LeakyReLU negative slope (0 = ReLU)
  const float* mask_src = nullptr;
  int mask_ld = 0;
  float mask_scale = 1.f;
  float mask_neg = 0.f;
  MansyDrop drop = {0.f, 0u, 0u};
  const float* resid = nullptr;
  int resid_ld = 0;
  int accumulate = 0;   // C += result (atomicAdd); forced when split-K > 1
  float* a_rowsum = nullptr;   // K-major A only: a_rowsum[m] += sum_k A(m,k)  (bias gradient riding on the dW product)
  // > 0 (with force_splitk > 1 and an otherwise plain epilogue): K split z stores its partial product to its own slab
  // C + z*split_slab (floats) with ordinary stores -- no atomics, no zero-fill; the consumer sums
  // mansy_gemm_effective_splits(K, force_splitk) slabs (skinny products: few output tiles, long K).
  long long split_slab = 0;
  // optional, LDS-DMA loop with 64-column tiles only (a pure speed hint: B must be zero outside the ranges): per 64-column
  // tile t the K range [tile_krange[2t], tile_krange[2t+1]) (multiples of 32) that holds all of B's non-zeros for those
  // columns -- block-diagonal weights (FeatureNet) skip the K-tiles that are structurally zero.
  const int* tile_krange = nullptr;
  // optional, 64-row tiles only (force_tile 64 / -64; ignored otherwise and by the split-bf16 loops): per 64-row tile t of C the
  // column range [tile_nrange[2t], tile_nrange[2t+1]) that the consumer reads; workgroups whose column tile lies outside it
  // exit at once and leave their part of C UNWRITTEN (gradient of a block-diagonal weight: only the diagonal blocks are wanted).
  const int* tile_nrange = nullptr;
  float flops_frac = 1.f;       // with tile_nrange: the fraction of C's tiles that run (only for the launch profiler's flop count)
  // optional companion of tile_nrange (64 x 64 tiles): the tiles that run, as (column tile, row tile) pairs -- the launch then has
  // tile_list_n workgroups per K split instead of one per tile of C (thousands of workgroups that exit at once cost ~1 ns each).
  const int* tile_list = nullptr; int tile_list_n = 0;
  // optional second product of identical shape / layout / leading dimensions in the SAME launch (plain accumulating
  // epilogue only, e.g. two dW products sharing an operand): C2 += A2 * B2, a_rowsum2 like a_rowsum.  Falls back to two
  // launches off the LDS-DMA loop.
  const float* pair_A = nullptr; const float* pair_B = nullptr; float* pair_C = nullptr; float* pair_rowsum = nullptr;
  // precision of THIS product (MANSY_PREC_*): 0 fp32, 1 bf16, 3 bf16x3, 6 bf16x6 -- there is no process-wide mode (ABI 8)
  int prec = 0;
  // kernel-selection override of THIS product (mansy_gemm_epilogue::variant, MANSY_VARIANT_*; 0 = the defaults).  The engines never
  // set it; the parity tests use it to run the same product on two loops, tools/ for A/B timings.
  int variant = 0;
  // optional addend applied BEFORE the mask stage to the columns n >= pre_col0: v += pre_a[m*pre_ld + n - pre_col0] (+ pre_b[..]).
  // (FeatureNet backward: the heads' residual gradients join the last 128 feature columns ahead of the LeakyReLU derivative,
  // which is the mask stage -- the former featgrad_finish launch.)
  const float* pre_a = nullptr; const float* pre_b = nullptr; int pre_ld = 0; int pre_col0 = 0;
  // optional, split-bf16 modes only: the B operand already split into bf16 planes (weights, split once per step by
  // mansy_launch_weight_planes): plane t of B(k, n) at b_planes[t * b_plane_stride + n * b_planes_ld + k] (K-contiguous rows, 16-byte
  // aligned, b_planes_ld % 8 == 0).  With a K-contiguous A the product then stages B by LDS-DMA: no split work and no ds_write for it.
  const unsigned short* b_planes = nullptr; long long b_plane_stride = 0; int b_planes_ld = 0;
  // bf16-STORAGE products (MANSY_PREC_BF16 only, round 6; csrc/gemm_bf16a.hip): the operands as bf16 images in HBM, staged by LDS-DMA without any
  // conversion.  a16: the A operand in A's own logical layout (K-contiguous [M, K] or K-major [K, M]), leading dimension a16_ld (elements);
  // b16: a K-major B ([K, N]: the X of a weight-gradient product), b16_ld.  With a16 set the float A / B pointers are not read.
  const unsigned short* a16 = nullptr; int a16_ld = 0;
  const unsigned short* b16 = nullptr; int b16_ld = 0;
  // optional bf16 image of the OUTPUT (after the whole epilogue), written next to -- or, with a null C, instead of -- the fp32 store
  // (row-major float4 epilogue only: c_vec_ok and no accumulation)
  unsigned short* c16 = nullptr; int c16_ld = 0;
  // the residual / the mask source read from their bf16 IMAGES instead of floats (bf16-storage mode with bf16 residual streams; row-major epilogue only;
  // same leading dimensions resid_ld / mask_ld).  The mask only needs the sign of its source.
  const unsigned short* resid16 = nullptr; const unsigned short* mask16 = nullptr;
};

// ---- decoding of GemmEpilogue::variant (bit layout: include/mansy_hip.h, MANSY_VARIANT_*).  A release build has NO process-wide knob: a call
// that passes 0 gets the compiled-in defaults.  Only a -DMANSY_LAB build (tools/: build_ext.build(lab=True) -> libmansy_hip_lab.so) has a
// settable default for calls that pass 0 (mansy_lab_set_variant), so that whole engine steps can be A/B-timed.
#ifdef MANSY_LAB
extern int g_mansy_lab_variant;
inline int mansy_variant_of(int v) { return v ? v : g_mansy_lab_variant; }
#else
inline int mansy_variant_of(int v) { return v; }
#endif
inline int mansy_var_bf16(int v) { const int b = mansy_variant_of(v) & 0xFF; return b ? b - 1 : 1; }                 // bf16x3 loop variant (default 1)
inline bool mansy_var_no_wsk(int v) { return (mansy_variant_of(v) & 0x100) != 0; }                                 // small products on the 64 x 64 loop
inline bool mansy_var_no_wsk_tn(int v) { return (mansy_variant_of(v) & 0x300) != 0; }                              // small weight-gradient products likewise
inline bool mansy_var_no_plain(int v) { return (mansy_variant_of(v) & 0x400) != 0; }                               // no compile-time plain instance
inline bool mansy_var_no_pair(int v) { return (mansy_variant_of(v) & 0x800) != 0; }                                // (lab default only) no paired launch
inline int mansy_var_col_group(int v) { const int g = (mansy_variant_of(v) >> 16) & 0xFF; return g ? g - 1 : 12; }   // column-group width (default 12)

// Weights -> bf16 planes for the split-bf16 products (gemm_bf16s.hip): for each listed [N, K] fp32 matrix W (leading dimension K)
// n_planes planes of W (plane t at out + t * plane_stride + off + n * K + k) and of its transpose (at out_t + t * plane_stride + off +
// k * N + n), a = a0 + a1 (+ a2) by round-to-nearest-even.  One launch for up to MANSY_WPLANE_MAX matrices.
constexpr int MANSY_WPLANE_MAX = 32;
struct MansyWPlaneTab { const float* w[MANSY_WPLANE_MAX]; int N[MANSY_WPLANE_MAX]; int K[MANSY_WPLANE_MAX]; long long off[MANSY_WPLANE_MAX]; int n; };
int mansy_launch_weight_planes(const MansyWPlaneTab& tab, unsigned short* out, unsigned short* out_t, long long plane_stride, int n_planes,
                               hipStream_t st);
int mansy_gemm_effective_splits(int K, int requested);
// bf16-storage products (MANSY_PREC_BF16 with bf16 images of the operands: GemmEpilogue::a16 + b_planes, or a16 + b16 for dW = dY^T X)
int mansy_launch_gemm_bf16a(int a_kmajor, int b_kmajor, float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep, int force_tile, int force_splitk,
                            hipStream_t st);
int mansy_launch_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor,
                          float* C, int ldc, int M, int N, int K, const GemmEpilogue& ep, int force_tile,
                          int force_splitk, hipStream_t st);

// ---------------------------------------------------------------- attention (attn.hip)
struct AttnShape {
  int nb;          // batch entries
  int H;           // heads
  int Lq, Lk;      // <= 16 each
  int dh;          // <= 64
  // strides in floats: element (b, row, h, d) at base + b*bs + row*rs + h*dh + d
  long long q_bs, q_rs, k_bs, k_rs, v_bs, v_rs, o_bs, o_rs;
  float scale;
};
// P_save: [nb*H, Lq, Lk] softmax probabilities BEFORE dropout (needed by backward); may be null in eval.
// img_f / img_s (all attention launchers; bf16-storage mode): the output rows' bf16 image -- the float4 stored at address a also goes to img_s + (a - img_f)
int mansy_launch_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* P_save,
                          const AttnShape& s, MansyDrop drop, hipStream_t st, const float* img_f = nullptr, unsigned short* img_s = nullptr, int img_only = 0,
                          int kv16 = 0);      // kv16 (Lq = 1 kernels): K / V rows are read from their bf16 images (bf16 K/V cache)
// dQ is overwritten; dK/dV are accumulated (+=) when accum_kv != 0, else overwritten.
// dq/dk/dv use the q/k/v strides of `s`; dO uses the o strides.
int mansy_launch_attn_bwd(const float* Q, const float* K, const float* V, const float* P_save, const float* dO,
                          float* dQ, float* dK, float* dV, const AttnShape& s, MansyDrop drop, int accum_kv,
                          hipStream_t st, const float* img_f = nullptr, unsigned short* img_s = nullptr, int img_only = 0);

// Deferred K/V gradients for a Lq == 1 attention evaluated at T steps against the same K/V rows (decoder cross-attention):
// per step mansy_launch_attn_bwd_dq writes dQ and the step's coefficients (dS, dropped P: [nb*H, Lk] each); one
// mansy_launch_attn_kvgrad then forms dK[j] = sum_i dS_i[j] q_i, dV[j] = sum_i Pk_i[j] dO_i (overwrites, or += when accum).
// Q_all / dO_all hold step i at + i*q_ts / + i*o_ts; dS_all / Pk_all are [T][nb*H][Lk].
int mansy_attn_deferred_kv_ok(const AttnShape& s, int T);
int mansy_launch_attn_bwd_dq(const float* Q, const float* K, const float* V, const float* P_save, const float* dO, float* dQ,
                             float* dS_out, float* Pk_out, const AttnShape& s, MansyDrop drop, hipStream_t st,
                             const float* img_f = nullptr, unsigned short* img_s = nullptr, int img_only = 0, int kv16 = 0);
int mansy_launch_attn_kvgrad(const float* Q_all, long long q_ts, const float* dO_all, long long o_ts, const float* dS_all,
                             const float* Pk_all, float* dK, float* dV, const AttnShape& s, int T, int accum, hipStream_t st,
                             const float* img_f = nullptr, unsigned short* img_s = nullptr);      // + bf16 images of dK / dV at img_s + (p - img_f)

// KV-cached decoder self-attention backward, "pull" form: the steps run T-1 -> 0; the call for step i records its coefficient
// rows and writes row i of the K/V gradient complete (own term + the terms of the later steps, read from the Q / dO slabs),
// so no K/V gradient row is ever read-modify-written and the slab needs no zero-fill.  s.Lk == step + 1.
int mansy_attn_selfpull_ok(const AttnShape& s, int T);
int mansy_launch_attn_bwd_selfpull(const float* Q_all, long long q_ts, const float* K, const float* V, const float* P_save,
                                   const float* dO_all, long long o_ts, float* dQ, float* dK, float* dV, float* dS_all, float* Pk_all,
                                   const AttnShape& s, int T, int step, MansyDrop drop, hipStream_t st,
                                   const float* img_f = nullptr, unsigned short* img_s = nullptr, int img_only = 0, int kv16 = 0);

// ---------------------------------------------------------------- norms (norm.hip)
// z = a (+ b);  y = LN(z) * w (+ bias).  z_out may be null (not saved) or alias a when b == null.
int mansy_launch_layernorm_fwd(const float* a, const float* b, const float* w, const float* bias, float* z_out,
                               float* y, float* mean, float* rstd, int rows, int C, float eps, hipStream_t st,
                               unsigned short* y16 = nullptr, const unsigned short* a16 = nullptr);      // y16: also (or, with y null, only) store y's bf16 image;
                                                                                                      // a16: read the input rows from their bf16 image (bf16 residual stream)
// dz = LN'(dy); dz_drop (optional) = dz * dropout-mask(drop) ; dw += sum dy*xhat ; dbias += sum dy.
// add_to (optional): dz += add_to (an extra gradient flowing into z's consumers' sum), applied before outputs.
int mansy_launch_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                               float* dz, float* dz_drop, MansyDrop drop, float* dw, float* dbias, int rows, int C,
                               hipStream_t st);

// Engine form of the LayerNorm backward: weight-gradient column sums go to per-workgroup slots
// partials[mansy_ln_bwd_parts(rows)][2][C] (overwritten, or added to when accumulate != 0) instead of atomics on
// dw/dbias; mansy_launch_ln_partials_reduce then adds the slots into dw / dbias (either may be null).
int mansy_ln_bwd_parts(int rows);
bool mansy_ln_bwd_partial_ok(int C);
int mansy_launch_layernorm_bwd_partial(const float* dy, const float* z, const float* mean, const float* rstd, const float* w,
                                       float* dz, float* dz_drop, MansyDrop drop, float* partials, int accumulate, int rows, int C,
                                       hipStream_t st, unsigned short* dz_drop16 = nullptr, const unsigned short* z16 = nullptr);      // dz_drop16: bf16 image of dz_drop; z16: z read from its image
int mansy_launch_ln_partials_reduce(const float* partials, int nparts, int C, float* dw, float* dbias, hipStream_t st);
// several slot sets in one launch (the end of a VP backward: every LayerNorm's weight / bias gradient)
constexpr int LN_MULTI_MAX = 64;      // >= 2 halves x (3 x 8 layers + 1)
struct MansyLnReduce { const float* partials; int nparts; float* dw; float* dbias; };
int mansy_launch_ln_partials_reduce_multi(const MansyLnReduce* sets, int n, int C, hipStream_t st);

// BatchNorm1d(train) + ELU + MaxPool1d(3,2,1) of the DistillLayer on conv output [B*S, C].
struct DistillShape { int B, S, M, C; int sync_world = 1; int (*hook)(int, void*) = nullptr; void* hook_user = nullptr; };
// Invokes the call's data-parallel hook (mansy_vp_config::bn_sync_fn; fn == nullptr: the deprecated process-wide registration of
// capi.hip: mansy_set_bn_sync_hook); which = 0 forward stats, 1 backward stats, 2 decoder-side gradients final.
int mansy_bn_sync_invoke(int which, int (*fn)(int, void*), void* user);
// the wave-split-K loop serves small fp32 weight-gradient (TN) products (gemm_f32.hip; callers that pick a K split for such a product ask first)
int mansy_gemm_wsk_tn_enabled();
// Two INDEPENDENT products as one launch where both resolve to the wave-split-K loop (gemm_f32.hip: gemm_f32_wsk_dual_kernel): the products launched
// between begin (returns 1 if pairing is on) and end must not depend on each other; end launches what was collected (as one grid if it can, else one by one).
int mansy_gemm_pair_begin();
int mansy_gemm_pair_end(hipStream_t st);
// stats_d: device scratch of 6*C doubles ([sum, sumsq] forward, [sum g, sum g*xhat] backward global + local copy).
// part (optional): scratch of MANSY_DISTILL_PARTS * 2 * C doubles -- the column sums then go through one partial per workgroup and a small reduce
// launch (32 partials per thread, <= 16 adds per address) instead of double atomics on 2 C addresses from every workgroup.
constexpr int MANSY_DISTILL_PARTS = 512;
int mansy_launch_distill_fwd(const float* conv, const float* bn_w, const float* bn_b, float* run_mean, float* run_var,
                             long long* num_batches, float* mean_out, float* rstd_out, float* mem, unsigned char* argmax,
                             double* stats_d, const DistillShape& s, int train, float eps, float momentum,
                             hipStream_t st, double* part = nullptr, unsigned short* mem16 = nullptr);
// dconv [B*S,C] from dmem [B*M,C]; dbn_w/dbn_b accumulated (+=).  g_tmp: [B*S,C] scratch.
int mansy_launch_distill_bwd(const float* conv, const float* dmem, const unsigned char* argmax, const float* bn_w,
                             const float* bn_b, const float* mean, const float* rstd, float* g_tmp, float* dconv,
                             float* dbn_w, float* dbn_b, double* stats_d, const DistillShape& s, hipStream_t st, double* part = nullptr,
                             unsigned short* dconv16 = nullptr);      // mem16 / dconv16: bf16 images of the outputs (bf16-storage mode)

// ---------------------------------------------------------------- fused decoder-step tail (dec_step.hip)
// One launch for the four row-wise ops between the last product of decoder step i and the first of step i+1:
//   z3 = a + b ; y3 = LN3(z3) ; dec_out = LN_dec(y3) ; tok = sigmoid(dec_out Wp^T + bp) ; emb_next = drop(tok We^T + be + pe_row)
// (emb_next == nullptr on the last step).  Same arithmetic as the separate kernels.
struct MansyDecTailFwd {
  const float* a; const float* b; const float* n3_w; const float* n3_b; float* z3; float* y3; float* m3; float* r3;
  const float* dn_w; const float* dn_b; float* dec_out; float* md; float* rd;
  const float* pw; const float* pb; float* tok_next; float* pred_bt; long long pred_stride;
  const float* ew; const float* eb; const float* pe_row; float* emb_next; MansyDrop edrop;
  int rows, C, C6; float eps;
  unsigned short* y3_16 = nullptr; unsigned short* emb_next16 = nullptr;      // bf16 images of y3 / emb_next (bf16-storage mode: operands of the next products)
  // bf16 residual stream (round 6): a16 = the first addend read from its image; z3_16 = image of z3; img_only != 0: z3 / y3 / emb_next are kept as images ONLY
  const unsigned short* a16 = nullptr; unsigned short* z3_16 = nullptr; int img_only = 0;
};
int mansy_dec_tail_ok(int C, int c6);
int mansy_launch_dec_tail_fwd(const MansyDecTailFwd& p, hipStream_t st);
// Backward mirror, at the start of backward step i: embedding backward of step i+1 (gx_next null on the last step), predictor
// backward, final-norm backward, LayerNorm3 backward of the last layer.  part_dn / part_n3: LayerNorm weight-gradient slot sets
// [n_slots][2][C] (accumulated; n_slots workgroups are launched).
struct MansyDecHeadBwd {
  const float* gx_next; const float* ew; float* dE_next; MansyDrop edrop;
  const float* dpred; long long dpred_stride; const float* pred; const float* pw; float* dz;
  const float* y3; const float* md; const float* rd; const float* dn_w; float* part_dn;
  const float* z3; const float* m3; const float* r3; const float* n3_w; float* part_n3; float* gz; float* dbr3; MansyDrop drop3;
  int rows, C, C6;
  unsigned short* dbr3_16 = nullptr;       // bf16 image of dbr3 (bf16-storage mode)
  int dbr3_img_only = 0;                   // != 0 (with dbr3_16): dbr3 is an operand of dense products only -- no float store
  const unsigned short* y3_16 = nullptr; const unsigned short* z3_16 = nullptr;      // (both or neither) y3 / z3 read from their images (bf16 residual stream)
};
int mansy_launch_dec_head_bwd(const MansyDecHeadBwd& p, int n_slots, hipStream_t st);

// ---------------------------------------------------------------- elementwise (elementwise.hip)
// out[r, c] = sum_k x[r,k] W[c,k] + b[c] + pe[pos(r), c], dropout(site, r*C+c).
// pos(r) = pos_fixed if pos_fixed >= 0 else (r % S).
int mansy_launch_embed_fwd(const float* x, int in_ch, const float* W, const float* b, const float* pe, float* out,
                           int rows, int C, int S, int pos_fixed, MansyDrop drop, hipStream_t st, unsigned short* out16 = nullptr);      // out16: bf16 image of out
// dE = dX * dropmask (stored, for the deferred dW) ; dtok[r,k] = sum_c dE[r,c] W[c,k]  (dtok may be null)
int mansy_launch_embed_bwd(const float* dX, const float* W, float* dE, float* dtok, int in_ch, int rows, int C,
                           MansyDrop drop, hipStream_t st);
// out[c*small_n + k] (c_major=1) or out[k*C + c] (c_major=0)  +=  sum_r small[r,k] * big[r,c]; bsum[...] += sums.
//   embed:     small = tok [R,6],  big = dE [R,C]   -> dW[C,6] (c_major=1), db[C]  = sum_r big[r,c]
//   predictor: small = dz  [R,6],  big = h  [R,C]   -> dW[6,C] (c_major=0), db[6]  = sum_r small[r,k]
int mansy_launch_outer_reduce(const float* small_, int small_n, const float* big, int rows, int C, float* out,
                              int c_major, float* bsum_big, float* bsum_small, hipStream_t st);
// y[r,k] = sigmoid(sum_c h[r,c] W[k,c] + b[k]); written to y_a[r*ya_stride + k] and (optional) y_b likewise.
int mansy_launch_predictor_fwd(const float* h, const float* W, const float* b, float* y_a, long long ya_stride,
                               float* y_b, long long yb_stride, int rows, int C, int out_ch, hipStream_t st);
// dz[r,k] = (dy_a[r*sa+k] + (dy_b ? dy_b[r*sb+k] : 0)) * y(1-y) ; dh[r,c] = sum_k dz[r,k] W[k,c]
int mansy_launch_predictor_bwd(const float* dy_a, long long sa, const float* dy_b, long long sb, const float* y,
                               long long sy, const float* W, float* dz, float* dh, int rows, int C, int out_ch,
                               hipStream_t st);
// MTIO loss (mtio.py:94-104): loss = 1/(2*B*T) * sum e^2 over [n] elements, e = periodic distance; dpred optional.
int mansy_launch_mtio_loss(const float* pred, const float* gt, long long n, float inv_2bt, double* loss_accum,
                           float* loss_out, float* dpred, hipStream_t st);
int mansy_launch_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                       float eps, float wd, int step, int decoupled, hipStream_t st, const float* bias_dev = nullptr);      // bias_dev: device [1 - b1^t, sqrt(1 - b2^t)] read by the kernel instead of the host's (hipGraph replays)
// im2col for the circular k=3 conv: col[b*S+s, ci*3+t] = x[b, (s+t-1) mod S, ci]
int mansy_launch_im2col3(const float* x, float* col, int B, int S, int C, hipStream_t st, unsigned short* col16 = nullptr);      // col16: bf16 image of col
// dx[b,s,ci] = sum_t dcol[b, (s-t+1) mod S, ci*3+t]
int mansy_launch_col2im3(const float* dcol, float* dx, int B, int S, int C, hipStream_t st);
// colsum: out[c] += sum_r x[r*ld + c]
int mansy_launch_colsum(const float* x, int ld, int rows, int C, float* out, hipStream_t st);
// y = a + b (n floats)
int mansy_launch_add(const float* a, const float* b, float* y, long long n, hipStream_t st);
// MTIO channel mix (mtio.py:72-90): out[b, l, k*c + j] = x[perm_k[b], l, j] ; perm null => identity
int mansy_launch_mtio_mix(const float* x, const int* perm1, const int* perm2, float* out, int B, int L, int c,
                          hipStream_t st);
int mansy_launch_mtio_mix3(const float* hist, const float* cur, const float* fut, const int* perm1, const int* perm2, float* src6, float* cur6, float* fut6,
                           int B, int S, int T, int c, hipStream_t st);
// ensemble mean over heads + wrap to [0,1] (mtio.py:125-133, utils/common.py:61-70): pred [B,T,heads*c] -> [B,T,c]
int mansy_launch_ensemble_wrap(const float* pred, float* out, long long rows, int heads, int c, hipStream_t st);
int mansy_launch_linreg_sample(const float* hist, const float* cur, int B, int S, int T, int c, float* out, hipStream_t st);
// transpose-copy [T,B,C] <-> [B,T,C]
int mansy_launch_tb_to_bt(const float* src, float* dst, int T, int B, int C, hipStream_t st);

int mansy_launch_traj_gather(const float* table, int L, int c, const int* idx, int B, int S, int T, float* hist, float* cur, float* fut,
                             hipStream_t st);
int mansy_launch_periodic_mse(const float* a, const float* b, long long rows, int c, float* out, hipStream_t st);

// ---------------------------------------------------------------- tile map (tilemap.hip)
int mansy_launch_tilemap_metrics(const unsigned long long* gt, const unsigned long long* pred, long long n, double* out, hipStream_t st);
int mansy_launch_tilemap(const float* xy, long long n, int W, int H, int nw, int nh, int fov_w, int fov_h,
                         unsigned long long* maps, hipStream_t st);
int mansy_launch_tilemap_iou(const unsigned long long* a, const unsigned long long* b, long long n, double* iou,
                             hipStream_t st);
int mansy_launch_tilemap_or_groups(const unsigned long long* maps, long long ngroups, int group, unsigned long long* out,
                                   hipStream_t st);
