// Viewport-prediction engine: the whole MTIO Transformer forward, hand-derived backward and the
// fused train step, sequenced on one HIP stream with no host synchronisation (hipGraph-capturable).
//
// What it computes (reference): ViewportTransformerMTIO._process_src_current / sample / loss and
// the train-loop body -- viewport_prediction/models/mtio.py:65-166, models/customized_transformer.py
// :13-83, run_models.py:37-44.  How (MI355X-first, not the reference's structure):
//   * decoder = KV-cached incremental decoder: step i computes ONE new token per sample instead of
//     re-running all i+1 positions (55 -> 10 token-positions per trajectory); identical function and
//     gradients when dropout is off (causal mask => earlier positions never change).
//   * every per-step activation lives in a step-major slab [T][B][C]; the recurrent backward only runs
//     the dX-type products per step, and ALL weight gradients of the decoder are deferred to one
//     dW = dY^T X GEMM per weight over the stacked [T*B] rows (reduce dim 40 960 at B = 4096).
//   * all dense products run on the fp32 MFMA kernel (gemm_f32.hip) with bias/ReLU/dropout/mask/
//     residual fused in the epilogue; attention is the fused one-wave-per-(batch,head) kernel.
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <vector>
#include "mansy_kernels.h"
#include "../../include/mansy_hip.h"

namespace {

constexpr int MAXL = 8;

inline uint32_t site_pe_src() { return 1u; }
inline uint32_t site_enc(int l, int k) { return 100u + l * 8 + k; }
inline uint32_t site_pe_tgt(int i) { return 1000u + i; }
inline uint32_t site_dec(int l, int i, int k) { return 10000u + (l * 64 + i) * 8 + k; }

struct LinearP { const float* w = nullptr; const float* b = nullptr; float* gw = nullptr; float* gb = nullptr; };
typedef LinearP NormP;
struct EncLayerP { LinearP in_proj, out_proj, lin1, lin2; NormP n1, n2; };
struct DecLayerP { LinearP sa_in, sa_out, ca_in, ca_out, lin1, lin2; NormP n1, n2, n3; };
struct VPParams {
  LinearP emb; EncLayerP enc[MAXL]; NormP enc_norm; DecLayerP dec[MAXL]; NormP dec_norm;
  LinearP conv; NormP bn; LinearP pred;
};

struct ParamInfo { std::string name; long long numel; int ndim; long long shape[4]; };

void add_param(std::vector<ParamInfo>& v, const std::string& name, long long a, long long b = 0, long long c = 0) {
  ParamInfo p; p.name = name; p.shape[0] = a; p.shape[1] = b; p.shape[2] = c; p.shape[3] = 0;
  p.ndim = c ? 3 : (b ? 2 : 1);
  p.numel = a * (b ? b : 1) * (c ? c : 1);
  v.push_back(p);
}

// Order == the reference model's state_dict order (parameters only).
std::vector<ParamInfo> param_table(const mansy_vp_config& c) {
  std::vector<ParamInfo> v;
  const long long d = c.d_model, f = c.d_ff;
  const bool hb = c.has_bias != 0;
  add_param(v, "embedding.linear.weight", d, c.in_ch);
  add_param(v, "embedding.linear.bias", d);
  auto mha = [&](const std::string& p) {
    add_param(v, p + "in_proj_weight", 3 * d, d);
    if (hb) add_param(v, p + "in_proj_bias", 3 * d);
    add_param(v, p + "out_proj.weight", d, d);
    if (hb) add_param(v, p + "out_proj.bias", d);
  };
  auto lin = [&](const std::string& p, long long o, long long i) {
    add_param(v, p + ".weight", o, i);
    if (hb) add_param(v, p + ".bias", o);
  };
  auto norm = [&](const std::string& p) {
    add_param(v, p + ".weight", d);
    if (hb) add_param(v, p + ".bias", d);
  };
  for (int l = 0; l < c.n_enc; ++l) {
    const std::string p = "transformer.encoder.layers." + std::to_string(l) + ".";
    mha(p + "self_attn.");
    lin(p + "linear1", f, d); lin(p + "linear2", d, f);
    norm(p + "norm1"); norm(p + "norm2");
  }
  norm("transformer.encoder.norm");
  for (int l = 0; l < c.n_dec; ++l) {
    const std::string p = "transformer.decoder.layers." + std::to_string(l) + ".";
    mha(p + "self_attn."); mha(p + "multihead_attn.");
    lin(p + "linear1", f, d); lin(p + "linear2", d, f);
    norm(p + "norm1"); norm(p + "norm2"); norm(p + "norm3");
  }
  norm("transformer.decoder.norm");
  add_param(v, "transformer.distill_layer.downConv.weight", d, d, 3);
  add_param(v, "transformer.distill_layer.downConv.bias", d);
  add_param(v, "transformer.distill_layer.norm.weight", d);
  add_param(v, "transformer.distill_layer.norm.bias", d);
  add_param(v, "predictor.0.weight", c.in_ch, d);
  add_param(v, "predictor.0.bias", c.in_ch);
  return v;
}

void bind_params(const mansy_vp_config& c, const float* const* params, float* const* grads, VPParams& P) {
  int k = 0;
  const bool hb = c.has_bias != 0;
  auto nextw = [&](LinearP& L, bool bias) {
    L.w = params[k]; L.gw = grads ? grads[k] : nullptr; ++k;
    if (bias) { L.b = params[k]; L.gb = grads ? grads[k] : nullptr; ++k; }
  };
  nextw(P.emb, true);
  for (int l = 0; l < c.n_enc; ++l) {
    EncLayerP& e = P.enc[l];
    nextw(e.in_proj, hb); nextw(e.out_proj, hb); nextw(e.lin1, hb); nextw(e.lin2, hb); nextw(e.n1, hb); nextw(e.n2, hb);
  }
  nextw(P.enc_norm, hb);
  for (int l = 0; l < c.n_dec; ++l) {
    DecLayerP& e = P.dec[l];
    nextw(e.sa_in, hb); nextw(e.sa_out, hb); nextw(e.ca_in, hb); nextw(e.ca_out, hb);
    nextw(e.lin1, hb); nextw(e.lin2, hb); nextw(e.n1, hb); nextw(e.n2, hb); nextw(e.n3, hb);
  }
  nextw(P.dec_norm, hb);
  nextw(P.conv, true);
  nextw(P.bn, true);
  nextw(P.pred, true);
}

// ------------------------------------------------------------------------------ workspace layout
struct EncBuf { float *qkv, *P, *ao, *z1, *m1, *r1, *y1, *h, *z2, *m2, *r2, *y2; };
struct DecBuf {
  float *memkv, *qkv, *P1, *ao1, *z1, *m1, *r1, *y1, *qc, *P2, *ao2, *z2, *m2, *r2, *y2, *h, *z3, *m3, *r3, *y3;
  float *dqkv, *dbr1, *dqc, *dbr2, *da, *dbr3, *dmemkv;   // backward slabs (deferred dW operands)
  float *dao2, *dS2, *Pk2;                                // deferred cross-attention K/V gradient operands (attn.hip kvgrad)
  float *dao1, *dS1, *Pk1;                                // self-attention backward, pull form (attn.hip selfpull)
};
struct Work {
  float *src6, *cur6, *fut6, *pred_bt, *dpred_bt;           // train_step staging
  float *x0; EncBuf enc[MAXL]; float *enc_out, *me, *re;
  float *col, *conv, *bn_mean, *bn_rstd, *mem; unsigned char* argmax; double* stats; double* dis_part;
  float *tok_all, *emb_all; DecBuf dec[MAXL]; float *dec_out, *md, *rd;
  float *t_enc, *t_dec;                                     // GEMM-out temporaries
  float *g_a, *g_b, *g_c, *g_wide, *g_ff;                   // encoder backward temporaries [N, .]
  float *s_a, *s_b, *s_c, *s_tok;                           // decoder step backward temporaries [B, .]
  float *dE_all, *dz_all, *dmem;
  float *lnp_dec, *lnp_enc;                                 // LayerNorm weight-gradient partial sums (norm.hip)
  double* loss_acc;
  unsigned short *wpl, *wpl_t;                              // bf16 planes of the GEMM weights and of their transposes (split-bf16 modes)
  long long wpl_stride;                                     // elements per plane (sum of the listed weights)
  // bf16-storage mode (MANSY_PREC_BF16, round 6): ONE image arena mirroring the float workspace element for element -- the bf16 image of the float at
  // fbase + i lives at img + i.  Producers of dense-product operands store the image next to the float; the products stage it by LDS-DMA as it is.
  const float* fbase; unsigned short* img;
};
struct BufInfo { std::string name; size_t off; size_t bytes; };

struct Layout {
  std::vector<BufInfo> bufs;
  size_t total = 0;
  char* base = nullptr;
  void* add(const std::string& name, size_t bytes) {
    const size_t off = (total + 255) & ~size_t(255);
    total = off + bytes;
    bufs.push_back({name, off, bytes});
    return base ? (void*)(base + off) : nullptr;
  }
  float* f(const std::string& name, size_t n) { return (float*)add(name, n * sizeof(float)); }
};

void build_layout(const mansy_vp_config& c, Layout& L, Work& W) {
  const size_t B = c.B, S = c.S, T = c.T, d = c.d_model, f = c.d_ff, H = c.n_head, C6 = c.in_ch;
  const size_t M = (S - 1) / 2 + 1, N = B * S, TB = T * B;
  W.src6 = L.f("src6", N * C6); W.cur6 = L.f("cur6", B * C6); W.fut6 = L.f("fut6", TB * C6);
  W.pred_bt = L.f("pred_bt", TB * C6); W.dpred_bt = L.f("dpred_bt", TB * C6);
  W.x0 = L.f("enc.x0", N * d);
  for (int l = 0; l < c.n_enc; ++l) {
    const std::string p = "enc" + std::to_string(l) + ".";
    EncBuf& e = W.enc[l];
    e.qkv = L.f(p + "qkv", N * 3 * d); e.P = L.f(p + "P", B * H * S * S); e.ao = L.f(p + "ao", N * d);
    e.z1 = L.f(p + "z1", N * d); e.m1 = L.f(p + "m1", N); e.r1 = L.f(p + "r1", N); e.y1 = L.f(p + "y1", N * d);
    e.h = L.f(p + "h", N * f);
    e.z2 = L.f(p + "z2", N * d); e.m2 = L.f(p + "m2", N); e.r2 = L.f(p + "r2", N); e.y2 = L.f(p + "y2", N * d);
  }
  W.enc_out = L.f("enc.out", N * d); W.me = L.f("enc.me", N); W.re = L.f("enc.re", N);
  W.col = L.f("dis.col", N * 3 * d); W.conv = L.f("dis.conv", N * d);
  W.bn_mean = L.f("dis.bn_mean", d); W.bn_rstd = L.f("dis.bn_rstd", d);
  W.mem = L.f("mem", B * M * d);
  W.argmax = (unsigned char*)L.add("dis.argmax", B * M * d);
  W.stats = (double*)L.add("dis.stats", 6 * d * sizeof(double));
  W.dis_part = (double*)L.add("dis.part", (size_t)MANSY_DISTILL_PARTS * 2 * d * sizeof(double));      // per-workgroup partial column sums (norm.hip)
  W.tok_all = L.f("tok_all", (T + 1) * B * C6);
  W.emb_all = L.f("dec.emb", TB * d);
  for (int l = 0; l < c.n_dec; ++l) {
    const std::string p = "dec" + std::to_string(l) + ".";
    DecBuf& e = W.dec[l];
    e.memkv = L.f(p + "memkv", B * M * 2 * d);
    e.qkv = L.f(p + "qkv", TB * 3 * d); e.P1 = L.f(p + "P1", TB * H * T); e.ao1 = L.f(p + "ao1", TB * d);
    e.z1 = L.f(p + "z1", TB * d); e.m1 = L.f(p + "m1", TB); e.r1 = L.f(p + "r1", TB); e.y1 = L.f(p + "y1", TB * d);
    e.qc = L.f(p + "qc", TB * d); e.P2 = L.f(p + "P2", TB * H * M); e.ao2 = L.f(p + "ao2", TB * d);
    e.z2 = L.f(p + "z2", TB * d); e.m2 = L.f(p + "m2", TB); e.r2 = L.f(p + "r2", TB); e.y2 = L.f(p + "y2", TB * d);
    e.h = L.f(p + "h", TB * f);
    e.z3 = L.f(p + "z3", TB * d); e.m3 = L.f(p + "m3", TB); e.r3 = L.f(p + "r3", TB); e.y3 = L.f(p + "y3", TB * d);
    e.dqkv = L.f(p + "dqkv", TB * 3 * d); e.dbr1 = L.f(p + "dbr1", TB * d); e.dqc = L.f(p + "dqc", TB * d);
    e.dbr2 = L.f(p + "dbr2", TB * d); e.da = L.f(p + "da", TB * f); e.dbr3 = L.f(p + "dbr3", TB * d);
    e.dmemkv = L.f(p + "dmemkv", B * M * 2 * d);
    e.dao2 = L.f(p + "dao2", TB * d); e.dS2 = L.f(p + "dS2", TB * H * M); e.Pk2 = L.f(p + "Pk2", TB * H * M);
    e.dao1 = L.f(p + "dao1", TB * d); e.dS1 = L.f(p + "dS1", TB * H * T); e.Pk1 = L.f(p + "Pk1", TB * H * T);
  }
  W.dec_out = L.f("dec.out", TB * d); W.md = L.f("dec.md", TB); W.rd = L.f("dec.rd", TB);
  W.t_enc = L.f("tmp.t_enc", N * d); W.t_dec = L.f("tmp.t_dec", B * d);
  W.g_a = L.f("tmp.g_a", N * d); W.g_b = L.f("tmp.g_b", N * d); W.g_c = L.f("tmp.g_c", N * d);
  W.g_wide = L.f("tmp.g_wide", N * 3 * d); W.g_ff = L.f("tmp.g_ff", N * f);
  W.s_a = L.f("tmp.s_a", B * d); W.s_b = L.f("tmp.s_b", B * d); W.s_c = L.f("tmp.s_c", B * d);
  W.s_tok = L.f("tmp.s_tok", B * C6);
  W.dE_all = L.f("dec.dE", TB * d); W.dz_all = L.f("dec.dz", TB * C6); W.dmem = L.f("dmem", B * M * d);
  // decoder: one slot set per LayerNorm (3 per layer + the final norm), accumulated over the T steps; encoder: one scratch set
  // (x 2: when the decoder runs as two half-batches on two streams each half accumulates into its own slot sets)
  W.lnp_dec = L.f("lnp.dec", (size_t)2 * (3 * c.n_dec + 1) * mansy_ln_bwd_parts((int)B) * 2 * d);
  W.lnp_enc = L.f("lnp.enc", (size_t)(2 * c.n_enc + 1) * mansy_ln_bwd_parts((int)N) * 2 * d);      // one set per encoder LayerNorm (round 5: reduced together at the end)
  W.loss_acc = (double*)L.add("loss_acc", 64);
  // bf16x3 mode: 2 planes x (W, W^T) of every GEMM weight, 2 bytes each (73 MB at d = 512, 1 % of the B = 4096 workspace; filled
  // once per step).  Only bf16x3 pre-splits its weights (prepare_planes), so the third plane round 2 reserved is gone; the region
  // stays part of the one workspace in every mode so that a model may change its precision between calls without re-sizing it.
  size_t wtot = 0;
  for (const ParamInfo& pi : param_table(c))
    if (pi.ndim >= 2 && pi.shape[0] * (pi.numel / pi.shape[0]) == pi.numel && pi.numel / pi.shape[0] >= 32 && pi.shape[0] >= 32)
      wtot += ((size_t)pi.numel + 63) / 64 * 64;
  W.wpl_stride = (long long)wtot;
  W.wpl = (unsigned short*)L.add("wplanes", wtot * 2 * sizeof(unsigned short));
  W.wpl_t = (unsigned short*)L.add("wplanes_t", wtot * 2 * sizeof(unsigned short));
  // image arena: 2 bytes per float of everything above (+ 50 % workspace, in every mode: a model may change its precision between calls)
  const size_t floats_above = (L.total + 3) / 4;
  W.fbase = (const float*)L.base;
  W.img = (unsigned short*)L.add("bf16_images", floats_above * sizeof(unsigned short));
}

int check_cfg(const mansy_vp_config* c) {
  MANSY_REQUIRE(c, "vp: null config");
  MANSY_REQUIRE(c->B >= 1 && c->S >= 1 && c->S <= 16 && c->T >= 1 && c->T <= 16, "vp: need B>=1, 1<=S,T<=16 (got B=%d S=%d T=%d)", c->B, c->S, c->T);
  MANSY_REQUIRE(c->n_head >= 1 && c->d_model % c->n_head == 0 && c->d_model / c->n_head <= 64, "vp: d_model/n_head must be an integer <= 64");
  MANSY_REQUIRE(c->d_model % 4 == 0 && c->d_ff % 4 == 0, "vp: d_model and d_ff must be multiples of 4");
  MANSY_REQUIRE(c->n_enc >= 1 && c->n_enc <= MAXL && c->n_dec >= 1 && c->n_dec <= MAXL, "vp: 1..%d layers supported", MAXL);
  MANSY_REQUIRE(c->in_ch >= 1 && c->in_ch <= 8 && c->in_ch % 3 == 0, "vp: in_ch must be 3*in_channel <= 8");
  MANSY_REQUIRE(c->max_len >= c->S && c->max_len >= c->T, "vp: positional table too short");
  MANSY_REQUIRE(c->p_pe >= 0.f && c->p_pe < 1.f && c->p_drop >= 0.f && c->p_drop < 1.f, "vp: dropout p outside [0,1)");
  MANSY_REQUIRE(c->precision == 0 || c->precision == 1 || c->precision == 3 || c->precision == 6, "vp: precision must be MANSY_PREC_F32 (0), _BF16 (1), _BF16X3 (3) or _BF16X6 (6), got %d", c->precision);
  return MANSY_OK;
}

#define RC(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

struct Eng {
  const mansy_vp_config& c;
  VPParams P;
  Work W;
  hipStream_t st;
  bool train;
  uint32_t seed;
  int B, S, T, d, f, H, dh, M, N, TB, C6;
  float drop_scale;

  Eng(const mansy_vp_config& cfg, hipStream_t s, bool tr, uint32_t sd) : c(cfg), st(s), train(tr), seed(sd) {
    wtab.n = 0;
    prec = c.precision;
    B = c.B; S = c.S; T = c.T; d = c.d_model; f = c.d_ff; H = c.n_head; dh = d / H; M = (S - 1) / 2 + 1;
    N = B * S; TB = T * B; C6 = c.in_ch;
    drop_scale = (train && c.p_drop > 0.f) ? 1.f / (1.f - c.p_drop) : 1.f;
  }
  // ---- split-bf16 modes: the GEMM weights' planes (gemm_bf16s.hip, pre-split B)
  MansyWPlaneTab wtab; int prec = 0;
  // ---- bf16-storage form of the MANSY_PREC_BF16 mode (round 6): every operand of a dense product has a bf16 image in HBM (written by its producer:
  // LayerNorm, attention, embedding, the fused decoder tail / head, a product's own epilogue), the products stage the images by LDS-DMA without any
  // conversion (csrc/gemm_bf16a.hip), the K/V cache is read as bf16 by the decoder attention.  Needs the vectorised row kernels and whole 64-wide
  // K-tiles: d %% 256 == 0, d_ff %% 64 == 0, B %% 64 == 0; any other shape keeps the fp32-operand loops (same arithmetic, twice the traffic).
  bool s16 = false;
  unsigned short* im(const float* p) const { return s16 ? W.img + (p - W.fbase) : nullptr; }
  bool s16_ok() const {
    return prec == 1 && d % 256 == 0 && d <= 1024 && f % 64 == 0 && B % 64 == 0 && S <= 10 && dh == 64 && H % 4 == 0 && mansy_dec_tail_ok(d, C6) != 0 &&
           mansy_ln_bwd_partial_ok(d) &&
           mansy_attn_deferred_kv_ok(cross_shape(), T) != 0 && mansy_attn_selfpull_ok(self_shape(T - 1), T) != 0;
  }
  MansyLnReduce ln_pending[LN_MULTI_MAX]; int ln_n_pending = 0;      // LayerNorm weight-gradient slot sets waiting for their reduce launch
  int flush_ln() {
    if (!ln_n_pending) return MANSY_OK;
    const int n = ln_n_pending; ln_n_pending = 0;
    return mansy_launch_ln_partials_reduce_multi(ln_pending, n, d, st);
  }
  void wtab_add(const LinearP& L, int Nout, int K) {
    if (!L.w || Nout < 32 || K < 32 || wtab.n >= MANSY_WPLANE_MAX) return;
    long long off = 0;
    for (int t = 0; t < wtab.n; ++t) off += ((long long)wtab.N[t] * wtab.K[t] + 63) / 64 * 64;
    wtab.w[wtab.n] = L.w; wtab.N[wtab.n] = Nout; wtab.K[wtab.n] = K; wtab.off[wtab.n] = off; ++wtab.n;
  }
  // once per engine call that runs products: split every GEMM weight into planes (order == param_table order, so the offsets
  // add up to the workspace size computed there)
  int prepare_planes() {
    wtab.n = 0;
    // bf16x3 only: measured +4 % on the forward / dX products (tools/gemm_bench.py --planes).  bf16x6 stays on the in-loop split:
    // two LDS stages of three planes leave one workgroup per CU and ran 8-15 % SLOWER with pre-split weights.
    if (prec != 3 && prec != 1) return MANSY_OK;      // (plain bf16 reads the leading plane only)
    for (int l = 0; l < c.n_enc; ++l) { const EncLayerP& p = P.enc[l]; wtab_add(p.in_proj, 3 * d, d); wtab_add(p.out_proj, d, d); wtab_add(p.lin1, f, d); wtab_add(p.lin2, d, f); }
    for (int l = 0; l < c.n_dec; ++l) {
      const DecLayerP& p = P.dec[l];
      wtab_add(p.sa_in, 3 * d, d); wtab_add(p.sa_out, d, d); wtab_add(p.ca_in, 3 * d, d); wtab_add(p.ca_out, d, d); wtab_add(p.lin1, f, d); wtab_add(p.lin2, d, f);
    }
    wtab_add(P.conv, d, 3 * d);
    long long tot = 0;
    for (int t = 0; t < wtab.n; ++t) tot += ((long long)wtab.N[t] * wtab.K[t] + 63) / 64 * 64;
    if (tot > W.wpl_stride) { wtab.n = 0; return MANSY_OK; }       // (cannot happen: same table as build_layout) -> in-loop split
    s16 = s16_ok();
    return mansy_launch_weight_planes(wtab, W.wpl, W.wpl_t, W.wpl_stride, 2, st);
  }
  // planes of the sub-matrix starting at `w` (rows r0.. of a listed weight) for the forward (transposed = false: B = W [Nout, K]) or
  // the dX product (transposed = true: B = W^T [K, Nout_total], columns r0..)
  void attach_planes(GemmEpilogue& ep, const float* w, bool transposed) const {
    for (int t = 0; t < wtab.n; ++t) {
      const long long e = w - wtab.w[t];
      if (e < 0 || e >= (long long)wtab.N[t] * wtab.K[t] || e % wtab.K[t]) continue;
      const long long r0 = e / wtab.K[t];
      ep.b_plane_stride = W.wpl_stride;
      if (!transposed) { ep.b_planes = W.wpl + wtab.off[t] + r0 * wtab.K[t]; ep.b_planes_ld = wtab.K[t]; }
      else { ep.b_planes = W.wpl_t + wtab.off[t] + r0; ep.b_planes_ld = wtab.N[t]; }
      return;
    }
  }
  // base: first element index of the launch inside the full tensor (a launch over the rows [b0, ..) of a [B, C] tensor passes b0 * C)
  MansyDrop dr(uint32_t site, float p, size_t base = 0) const {
    MansyDrop x; x.p = train ? p : 0.f; x.seed = seed; x.site = site; x.base = (uint32_t)base; return x;
  }

  // Y[rows,N] = X[rows,K] W[N,K]^T (+b) with epilogue
  // resid != nullptr: Y = resid + drop(X W^T + b) -- the sub-layer's residual sum z, written by the product's epilogue, so that
  // the LayerNorm that follows reads ONE tensor and writes one (it used to read x and the product, and write z and y).
  // Same fp32 operations in the same order as drop(product) stored and then added by the LayerNorm kernel: bit-identical.
  // y_img (bf16-storage mode): 0 = the output as floats, 1 = floats + bf16 image (it is also an operand of a later product / the K/V cache), 2 = the image
  // ONLY.  A product with a residual forms a sub-layer's z = resid + drop(product): in the bf16-storage mode the residual stream is bf16 -- the residual
  // is read from its image and z is kept as an image only (the LayerNorm that follows reads it there)
  int lin_fwd(const float* X, int rows, int K, const float* w, const float* b, int Nout, float* Y, int relu, MansyDrop drop,
              const float* resid = nullptr, int y_img = 0, bool x_img = true) {
    GemmEpilogue ep; ep.prec = prec; ep.bias = b; ep.relu = relu; ep.drop = drop; ep.resid = resid; ep.resid_ld = Nout;
    if (prec) attach_planes(ep, w, false);
    if (s16 && resid) { ep.resid16 = im(resid); ep.resid = nullptr; y_img = 2; }
    if (s16 && y_img) { ep.c16 = im(Y); ep.c16_ld = Nout; }          // (the shared epilogue stores it on either loop)
    if (s16 && x_img && ep.b_planes && K % 64 == 0) {
      ep.a16 = im(X); ep.a16_ld = K;
      return mansy_launch_gemm_bf16a(0, 0, y_img == 2 ? nullptr : Y, Nout, rows, Nout, K, ep, 0, 0, st);
    }
    return mansy_launch_gemm_f32(X, K, 0, w, K, 0, Y, Nout, rows, Nout, K, ep, 0, 0, st);
  }
  // y = LN(z) for a z already formed by lin_fwd(..., resid)
  int ln_of(const float* z, const NormP& n, float* y, float* m, float* r, int rows) {
    if (s16) return mansy_launch_layernorm_fwd(nullptr, nullptr, n.w, n.b, nullptr, nullptr, m, r, rows, d, c.ln_eps, st, im(y), im(z));      // bf16 residual stream: image in, image out
    return mansy_launch_layernorm_fwd(z, nullptr, n.w, n.b, nullptr, y, m, r, rows, d, c.ln_eps, st, im(y));
  }
  // dX[rows,K] = dY[rows,N] W[N,K] (+resid) (mask)
  // dx_img: 0 = dX as floats only (it feeds a row-wise kernel), 1 = floats + bf16 image, 2 = the image ONLY (dX is nothing but an operand of later
  // products: no float copy is written) -- bf16-storage mode; otherwise floats.  img_in = false: dY has no image (its producer keeps floats only).
  int lin_dx(const float* dY, int rows, int Nout, const float* w, int K, float* dX, const float* resid, const float* mask_src,
             float mask_scale, int dx_img = 0, bool img_in = true) {
    GemmEpilogue ep; ep.prec = prec; ep.resid = resid; ep.resid_ld = K; ep.mask_src = mask_src; ep.mask_ld = K; ep.mask_scale = mask_scale;
    if (prec) attach_planes(ep, w, true);
    if (s16 && img_in && ep.b_planes && Nout % 64 == 0) {
      ep.a16 = im(dY); ep.a16_ld = Nout;
      if (mask_src) { ep.mask16 = im(mask_src); ep.mask_src = nullptr; }      // (the FFN hidden is kept as an image only: the ReLU mask needs its sign)
      if (dx_img) { ep.c16 = im(dX); ep.c16_ld = K; }
      return mansy_launch_gemm_bf16a(0, 0, dx_img == 2 ? nullptr : dX, K, rows, K, Nout, ep, 0, 0, st);
    }
    return mansy_launch_gemm_f32(dY, Nout, 0, w, K, 1, dX, K, rows, K, Nout, ep, 0, 0, st);
  }
  // gw[N,K] += dY[rows,N]^T X[rows,K] ; gb[N] += colsum(dY)
  // img_in = false: one of the operands has no bf16 image (floats only): the fp32-operand loop
  int lin_dw(const float* dY, const float* X, int rows, int Nout, int K, float* gw, float* gb, bool img_in = true) {
    GemmEpilogue ep; ep.prec = prec; ep.accumulate = 1; ep.a_rowsum = gb;      // bias gradient = row sums of dY^T, taken from the staged A tiles
    if (s16 && img_in && rows % 64 == 0 && Nout % 8 == 0 && K % 8 == 0) {
      ep.a16 = im(dY); ep.a16_ld = Nout; ep.b16 = im(X); ep.b16_ld = K;
      return mansy_launch_gemm_bf16a(1, 1, gw, K, Nout, K, rows, ep, 0, 0, st);
    }
    return mansy_launch_gemm_f32(dY, Nout, 1, X, K, 1, gw, K, Nout, K, rows, ep, 0, 0, st);
  }
  int ln_fwd(const float* a, const float* b, const NormP& n, float* z, float* y, float* m, float* r, int rows) {
    if (s16 && !b && !z) return mansy_launch_layernorm_fwd(nullptr, nullptr, n.w, n.b, nullptr, y, m, r, rows, d, c.ln_eps, st, im(y), im(a));   // (the input is a LayerNorm output: image only)
    return mansy_launch_layernorm_fwd(a, b, n.w, n.b, z, y, m, r, rows, d, c.ln_eps, st, im(y));
  }
  // part_slot < 0: encoder LayerNorm (applied once): partial sums into the scratch set, reduced into the gradient right away.
  // part_slot >= 0: decoder LayerNorm #part_slot (applied at every step): accumulate into its slot set, reduced after the loop.
  int ln_bwd(const float* dy, const float* z, const float* m, const float* r, const NormP& n, float* dz, float* dz_drop,
             MansyDrop drop, int rows, int part_slot = -1) {
    if (!mansy_ln_bwd_partial_ok(d)) return mansy_launch_layernorm_bwd(dy, z, m, r, n.w, dz, dz_drop, drop, n.gw, n.gb, rows, d, st);
    if (part_slot < 0) {          // encoder LayerNorm -(part_slot + 1): its own scratch set, overwritten; added into the gradient by the one reduce launch at the end
      const int e = -(part_slot + 1);
      float* set = W.lnp_enc + (size_t)e * mansy_ln_bwd_parts(rows) * 2 * d;
      RC(mansy_launch_layernorm_bwd_partial(dy, s16 ? nullptr : z, m, r, n.w, dz, s16 ? nullptr : dz_drop, drop, set, 0, rows, d, st, dz_drop ? im(dz_drop) : nullptr,
                                            im(z)));      // (bf16 storage: z is read from its image, dz_drop feeds products only -- image only)
      ln_pending[ln_n_pending++] = MansyLnReduce{set, mansy_ln_bwd_parts(rows), n.gw, n.gb};
      return MANSY_OK;
    }
    float* slots = W.lnp_dec + (size_t)part_slot * mansy_ln_bwd_parts(B) * 2 * d;
    return mansy_launch_layernorm_bwd_partial(dy, s16 ? nullptr : z, m, r, n.w, dz, s16 ? nullptr : dz_drop, drop, slots, 1, rows, d, st, dz_drop ? im(dz_drop) : nullptr, im(z));
  }

  AttnShape enc_shape() const {
    AttnShape s; s.nb = B; s.H = H; s.Lq = S; s.Lk = S; s.dh = dh;
    s.q_bs = s.k_bs = s.v_bs = (long long)S * 3 * d; s.q_rs = s.k_rs = s.v_rs = 3 * d;
    s.o_bs = (long long)S * d; s.o_rs = d; s.scale = 1.f / sqrtf((float)dh);
    return s;
  }
  AttnShape self_shape(int i, int nb = -1) const {   // query = slab i, keys = slabs 0..i; nb rows of the batch (default: all)
    AttnShape s; s.nb = nb < 0 ? B : nb; s.H = H; s.Lq = 1; s.Lk = i + 1; s.dh = dh;
    s.q_bs = 3 * d; s.q_rs = 0; s.k_bs = s.v_bs = 3 * d; s.k_rs = s.v_rs = (long long)B * 3 * d;
    s.o_bs = d; s.o_rs = 0; s.scale = 1.f / sqrtf((float)dh);
    return s;
  }
  AttnShape cross_shape(int nb = -1) const {
    AttnShape s; s.nb = nb < 0 ? B : nb; s.H = H; s.Lq = 1; s.Lk = M; s.dh = dh;
    s.q_bs = d; s.q_rs = 0; s.k_bs = s.v_bs = (long long)M * 2 * d; s.k_rs = s.v_rs = 2 * d;
    s.o_bs = d; s.o_rs = 0; s.scale = 1.f / sqrtf((float)dh);
    return s;
  }

  // ------------------------------------------------------------------ forward
  int forward(const float* src, const float* cur, const float* pe, float* bn_rm, float* bn_rv, long long* bn_nbt, float* pred_bt) {
    RC(prepare_planes());
    RC(mansy_launch_embed_fwd(src, C6, P.emb.w, P.emb.b, pe, W.x0, N, d, S, -1, dr(site_pe_src(), c.p_pe), st, im(W.x0)));
    const float* x = W.x0;
    for (int l = 0; l < c.n_enc; ++l) {
      const EncLayerP& p = P.enc[l]; EncBuf& e = W.enc[l];
      RC(lin_fwd(x, N, d, p.in_proj.w, p.in_proj.b, 3 * d, e.qkv, 0, mansy_no_drop()));
      RC(mansy_launch_attn_fwd(e.qkv, e.qkv + d, e.qkv + 2 * d, e.ao, e.P, enc_shape(), dr(site_enc(l, 0), c.p_drop), st, W.fbase, im(W.fbase), 1));
      RC(lin_fwd(e.ao, N, d, p.out_proj.w, p.out_proj.b, d, e.z1, 0, dr(site_enc(l, 1), c.p_drop), x));          // z1 = x + drop(out_proj(ao))
      RC(ln_of(e.z1, p.n1, e.y1, e.m1, e.r1, N));
      RC(lin_fwd(e.y1, N, d, p.lin1.w, p.lin1.b, f, e.h, 1, dr(site_enc(l, 2), c.p_drop), nullptr, 2));
      RC(lin_fwd(e.h, N, f, p.lin2.w, p.lin2.b, d, e.z2, 0, dr(site_enc(l, 3), c.p_drop), e.y1));                // z2 = y1 + drop(lin2(h))
      RC(ln_of(e.z2, p.n2, e.y2, e.m2, e.r2, N));
      x = e.y2;
    }
    RC(ln_fwd(x, nullptr, P.enc_norm, nullptr, W.enc_out, W.me, W.re, N));
    // DistillLayer: circular conv k=3 as one K=3d GEMM on the im2col image, then BN+ELU+maxpool
    RC(mansy_launch_im2col3(W.enc_out, W.col, B, S, d, st, im(W.col)));
    RC(lin_fwd(W.col, N, 3 * d, P.conv.w, P.conv.b, d, W.conv, 0, mansy_no_drop()));
    DistillShape ds = {B, S, M, d, c.bn_sync_world > 1 ? c.bn_sync_world : 1, c.bn_sync_fn, c.bn_sync_user};
    RC(mansy_launch_distill_fwd(W.conv, P.bn.w, P.bn.b, bn_rm, bn_rv, bn_nbt, W.bn_mean, W.bn_rstd, W.mem, W.argmax, W.stats, ds,
                                train ? 1 : 0, c.bn_eps, c.bn_momentum, st, W.dis_part, im(W.mem)));
    for (int l = 0; l < c.n_dec; ++l) {
      const DecLayerP& p = P.dec[l];
      RC(lin_fwd(W.mem, B * M, d, p.ca_in.w + (size_t)d * d, p.ca_in.b ? p.ca_in.b + d : nullptr, 2 * d, W.dec[l].memkv, 0, mansy_no_drop(), nullptr, 1));
      // (the projected K/V rows get an image: the cross-attention cache)
    }
    MANSY_HIP_CHECK(hipMemcpyAsync(W.tok_all, cur, sizeof(float) * B * C6, hipMemcpyDeviceToDevice, st));
    // the four row-wise ops between the last product of step i and the first of step i+1 run as one launch (dec_step.hip)
    const bool fuse_tail = mansy_dec_tail_ok(d, C6) != 0;
    // Trajectories are independent through the whole recurrence, so the batch can run as two halves on two streams: the products
    // of one half (MFMA-bound) execute under the attention / LayerNorm passes of the other (HBM-bound).  Submission is interleaved
    // step by step; every launch indexes the full slabs at its row offset and draws the dropout masks of those rows.
    const bool split = split_ok();
    const int h0 = split ? B / 2 : B;
    if (split) RC(fork());
    for (int i = 0; i < T; ++i) {
      RC(dec_fwd_step(i, 0, h0, pe, pred_bt, fuse_tail));
      if (split) { hipStream_t keep = st; st = st2; int rc = dec_fwd_step(i, h0, B - h0, pe, pred_bt, fuse_tail); st = keep; RC(rc); }
    }
    if (split) RC(join());
    return MANSY_OK;
  }

  // ---- two-stream plumbing (the second stream and its events are created once per process)
  hipStream_t st2 = nullptr;
  // mansy_vp_config::two_stream.  Measured at B = 4096: train step 23.08 -> 22.67 ms (+1.8 %), sample() 507 -> 529 k trajectories/s
  // (+4 %), identical losses.  The host mirror sets it by default; per-kernel timings (bench.py's roofline leg, rocprof kernel
  // stats) are taken with it off, because concurrent kernels stretch each other's durations.
  // Round 5 (profiles/r05_two_stream_sweep.txt): below B = 2048 the step is a latency-bound chain of ~5 us launches whatever the row count, and a second
  // stream only doubles the launches (B = 512, hist 5 / pred 15: 7.07 ms with the split, 6.11 without; sample() 3.0 vs 2.2 ms); at 1024 it is even; from
  // 2048 it pays (+4 %).  two_stream = 1: the engine decides (B >= 2048); 2: forced wherever the halves are whole (B >= 256, even: the equivalence test).
  bool split_ok() const { return B % 2 == 0 && ((c.two_stream == 1 && B >= 2048) || (c.two_stream == 2 && B >= 256)); }
  int fork() {
    static thread_local hipStream_t s2 = nullptr; static thread_local hipEvent_t ev_f = nullptr;      // one side stream per host thread
    if (!s2) { MANSY_HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking)); MANSY_HIP_CHECK(hipEventCreateWithFlags(&ev_f, hipEventDisableTiming)); }
    st2 = s2;
    MANSY_HIP_CHECK(hipEventRecord(ev_f, st));
    MANSY_HIP_CHECK(hipStreamWaitEvent(st2, ev_f, 0));
    return MANSY_OK;
  }
  int join() {
    static thread_local hipEvent_t ev_j = nullptr;
    if (!ev_j) MANSY_HIP_CHECK(hipEventCreateWithFlags(&ev_j, hipEventDisableTiming));
    MANSY_HIP_CHECK(hipEventRecord(ev_j, st2));
    MANSY_HIP_CHECK(hipStreamWaitEvent(st, ev_j, 0));
    return MANSY_OK;
  }

  // decoder step i for the rows [b0, b0 + nb) of the batch, on the current stream
  int dec_fwd_step(int i, int b0, int nb, const float* pe, float* pred_bt, bool fuse_tail) {
    const size_t o = (size_t)i * B + b0;                       // row of this launch inside the step-major [T][B][.] slabs
    const float* tok = W.tok_all + o * C6;
    float* emb = W.emb_all + o * d;
    float* t_dec = W.t_dec + (size_t)b0 * d;
    if (i == 0 || !fuse_tail) RC(mansy_launch_embed_fwd(tok, C6, P.emb.w, P.emb.b, pe, emb, nb, d, 1, i, dr(site_pe_tgt(i), c.p_pe, (size_t)b0 * d), st, im(emb)));
    const float* xi = emb;
    for (int l = 0; l < c.n_dec; ++l) {
      const DecLayerP& p = P.dec[l]; DecBuf& e = W.dec[l];
      float* qkv_i = e.qkv + o * 3 * d;
      const float* kv0 = e.qkv + (size_t)b0 * 3 * d;           // keys / values of step 0 for these rows (step stride B * 3d)
      const float* mkv = e.memkv + (size_t)b0 * M * 2 * d;
      RC(lin_fwd(xi, nb, d, p.sa_in.w, p.sa_in.b, 3 * d, qkv_i, 0, mansy_no_drop(), nullptr, 1));      // + bf16 image: the K/V cache of the bf16-storage mode
      RC(mansy_launch_attn_fwd(qkv_i, kv0 + d, kv0 + 2 * d, e.ao1 + o * d, e.P1 + o * H * T, self_shape(i, nb),
                               dr(site_dec(l, i, 0), c.p_drop, (size_t)b0 * H * (i + 1)), st, W.fbase, im(W.fbase), 1, s16 ? 1 : 0));
      RC(lin_fwd(e.ao1 + o * d, nb, d, p.sa_out.w, p.sa_out.b, d, e.z1 + o * d, 0, dr(site_dec(l, i, 1), c.p_drop, (size_t)b0 * d), xi));
      RC(ln_of(e.z1 + o * d, p.n1, e.y1 + o * d, e.m1 + o, e.r1 + o, nb));
      RC(lin_fwd(e.y1 + o * d, nb, d, p.ca_in.w, p.ca_in.b, d, e.qc + o * d, 0, mansy_no_drop()));
      RC(mansy_launch_attn_fwd(e.qc + o * d, mkv, mkv + d, e.ao2 + o * d, e.P2 + o * H * M, cross_shape(nb),
                               dr(site_dec(l, i, 2), c.p_drop, (size_t)b0 * H * M), st, W.fbase, im(W.fbase), 1, s16 ? 1 : 0));
      RC(lin_fwd(e.ao2 + o * d, nb, d, p.ca_out.w, p.ca_out.b, d, e.z2 + o * d, 0, dr(site_dec(l, i, 3), c.p_drop, (size_t)b0 * d), e.y1 + o * d));
      RC(ln_of(e.z2 + o * d, p.n2, e.y2 + o * d, e.m2 + o, e.r2 + o, nb));
      RC(lin_fwd(e.y2 + o * d, nb, d, p.lin1.w, p.lin1.b, f, e.h + o * f, 1, dr(site_dec(l, i, 4), c.p_drop, (size_t)b0 * f), nullptr, 2));
      if (fuse_tail && l == c.n_dec - 1) {                  // LayerNorm3 of the last layer is the head of the fused tail (a + b form)
        RC(lin_fwd(e.h + o * f, nb, f, p.lin2.w, p.lin2.b, d, t_dec, 0, dr(site_dec(l, i, 5), c.p_drop, (size_t)b0 * d)));
        break;
      }
      RC(lin_fwd(e.h + o * f, nb, f, p.lin2.w, p.lin2.b, d, e.z3 + o * d, 0, dr(site_dec(l, i, 5), c.p_drop, (size_t)b0 * d), e.y2 + o * d));
      RC(ln_of(e.z3 + o * d, p.n3, e.y3 + o * d, e.m3 + o, e.r3 + o, nb));
      xi = e.y3 + o * d;
    }
    if (fuse_tail) {
      const DecLayerP& p = P.dec[c.n_dec - 1]; DecBuf& e = W.dec[c.n_dec - 1];
      MansyDecTailFwd tp;
      tp.a = e.y2 + o * d; tp.b = t_dec; tp.n3_w = p.n3.w; tp.n3_b = p.n3.b; tp.z3 = e.z3 + o * d; tp.y3 = e.y3 + o * d;
      tp.m3 = e.m3 + o; tp.r3 = e.r3 + o;
      tp.dn_w = P.dec_norm.w; tp.dn_b = P.dec_norm.b; tp.dec_out = W.dec_out + o * d; tp.md = W.md + o; tp.rd = W.rd + o;
      tp.pw = P.pred.w; tp.pb = P.pred.b; tp.tok_next = W.tok_all + ((size_t)(i + 1) * B + b0) * C6;
      tp.pred_bt = pred_bt ? pred_bt + ((size_t)b0 * T + i) * C6 : nullptr; tp.pred_stride = (long long)T * C6;
      const bool more = i + 1 < T;
      tp.ew = P.emb.w; tp.eb = P.emb.b; tp.pe_row = pe + (size_t)(i + 1) * d; tp.emb_next = more ? W.emb_all + ((size_t)(i + 1) * B + b0) * d : nullptr;
      tp.edrop = dr(site_pe_tgt(i + 1), c.p_pe, (size_t)b0 * d);
      tp.rows = nb; tp.C = d; tp.C6 = C6; tp.eps = c.ln_eps;
      tp.y3_16 = im(tp.y3); tp.emb_next16 = tp.emb_next ? im(tp.emb_next) : nullptr;
      tp.a16 = im(tp.a); tp.z3_16 = im(tp.z3); tp.img_only = s16 ? 1 : 0;          // bf16 residual stream: y2 read from its image, z3 / y3 / emb_next kept as images only
      return mansy_launch_dec_tail_fwd(tp, st);
    }
    RC(ln_fwd(xi, nullptr, P.dec_norm, nullptr, W.dec_out + o * d, W.md + o, W.rd + o, nb));
    return mansy_launch_predictor_fwd(W.dec_out + o * d, P.pred.w, P.pred.b, W.tok_all + ((size_t)(i + 1) * B + b0) * C6, C6,
                                      pred_bt ? pred_bt + ((size_t)b0 * T + i) * C6 : nullptr, (long long)T * C6, nb, d, C6, st);
  }

  // ------------------------------------------------------------------ backward
  // backward of decoder step i for the rows [b0, b0 + nb) on the current stream; `half` selects the LayerNorm slot sets
  int dec_bwd_step(int i, int b0, int nb, int half, const float* dpred_bt, bool defer_cross, bool pull_self, bool ln_parts, bool fuse_head) {
    const float ms = drop_scale;
    const size_t lnp_set = (size_t)mansy_ln_bwd_parts(B) * 2 * d;
    float* const lnp = W.lnp_dec + (size_t)half * (3 * c.n_dec + 1) * lnp_set;
    const size_t o = (size_t)i * B + b0;
    const float* pred_tb = W.tok_all + (size_t)B * C6;
    float* gx = W.s_b + (size_t)b0 * d;     // gradient wrt the current layer's output
    float* gz = W.s_a + (size_t)b0 * d;     // scratch for residual-path gradients
    float* gt = W.s_c + (size_t)b0 * d;
    float* s_tok = W.s_tok + (size_t)b0 * C6;
    const size_t bd = (size_t)b0 * d;        // dropout index base of a [.., d]-wide row kernel
    auto ln_bwd_dec = [&](const float* dy, const float* z, const float* m, const float* r, const NormP& n, float* dz, float* dz_drop, MansyDrop drop,
                          int slot) {
      if (!ln_parts) return mansy_launch_layernorm_bwd(dy, z, m, r, n.w, dz, dz_drop, drop, n.gw, n.gb, nb, d, st);
      return mansy_launch_layernorm_bwd_partial(dy, s16 ? nullptr : z, m, r, n.w, dz, s16 ? nullptr : dz_drop, drop, lnp + (size_t)slot * lnp_set, 1, nb, d, st,
                                                dz_drop ? im(dz_drop) : nullptr, im(z));
    };
    if (fuse_head) {
      const int L = c.n_dec - 1;
      DecBuf& e = W.dec[L];
      MansyDecHeadBwd hp;
      const bool has_next = i < T - 1;           // gx still holds d/d(embedding input) of step i+1
      hp.gx_next = has_next ? gx : nullptr; hp.ew = P.emb.w; hp.dE_next = has_next ? W.dE_all + (o + B) * d : nullptr;
      hp.edrop = dr(site_pe_tgt(i + 1), c.p_pe, bd);
      hp.dpred = dpred_bt + ((size_t)b0 * T + i) * C6; hp.dpred_stride = (long long)T * C6; hp.pred = pred_tb + o * C6; hp.pw = P.pred.w;
      hp.dz = W.dz_all + o * C6;
      hp.y3 = e.y3 + o * d; hp.md = W.md + o; hp.rd = W.rd + o; hp.dn_w = P.dec_norm.w; hp.part_dn = lnp + (size_t)(3 * c.n_dec) * lnp_set;
      hp.z3 = e.z3 + o * d; hp.m3 = e.m3 + o; hp.r3 = e.r3 + o; hp.n3_w = P.dec[L].n3.w; hp.part_n3 = lnp + (size_t)(3 * L + 2) * lnp_set;
      hp.gz = gz; hp.dbr3 = e.dbr3 + o * d; hp.drop3 = dr(site_dec(L, i, 5), c.p_drop, bd);
      hp.rows = nb; hp.C = d; hp.C6 = C6; hp.dbr3_16 = im(hp.dbr3); hp.dbr3_img_only = s16 ? 1 : 0;
      hp.y3_16 = im(hp.y3); hp.z3_16 = im(hp.z3);
      // (its own grid rule: 16 rows per workgroup as before -- 8 made this fused kernel 17.9 -> 22 us; it fills the first slots of its sets, the rest stay
      // zero from the memset and the reduce launches add them as such)
      RC(mansy_launch_dec_head_bwd(hp, std::min(mansy_ln_bwd_parts(nb), (nb + 15) / 16), st));
    } else {
      // predictor + final decoder LayerNorm
      RC(mansy_launch_predictor_bwd(dpred_bt + ((size_t)b0 * T + i) * C6, (long long)T * C6, i < T - 1 ? s_tok : nullptr, C6,
                                    pred_tb + o * C6, C6, P.pred.w, W.dz_all + o * C6, gz, nb, d, C6, st));
      const float* last_y = W.dec[c.n_dec - 1].y3 + o * d;
      RC(ln_bwd_dec(gz, last_y, W.md + o, W.rd + o, P.dec_norm, gx, nullptr, mansy_no_drop(), 3 * c.n_dec));
    }
    for (int l = c.n_dec - 1; l >= 0; --l) {
      const DecLayerP& p = P.dec[l]; DecBuf& e = W.dec[l];
      const float* mkv = e.memkv + (size_t)b0 * M * 2 * d;
      // norm3( y2 + drop(lin2(h)) )
      if (!(fuse_head && l == c.n_dec - 1))
        RC(ln_bwd_dec(gx, e.z3 + o * d, e.m3 + o, e.r3 + o, p.n3, gz, e.dbr3 + o * d, dr(site_dec(l, i, 5), c.p_drop, bd), 3 * l + 2));
      RC(lin_dx(e.dbr3 + o * d, nb, d, p.lin2.w, f, e.da + o * f, nullptr, e.h + o * f, ms, 2));      // da is an operand of two products and nothing else: image only
      RC(lin_dx(e.da + o * f, nb, f, p.lin1.w, d, gt, gz, nullptr, 1.f));                 // gt = d/dy2
      // norm2( y1 + drop(ca_out(ao2)) )
      RC(ln_bwd_dec(gt, e.z2 + o * d, e.m2 + o, e.r2 + o, p.n2, gz, e.dbr2 + o * d, dr(site_dec(l, i, 3), c.p_drop, bd), 3 * l + 1));
      if (defer_cross) {
        float* dao2_i = e.dao2 + o * d;
        RC(lin_dx(e.dbr2 + o * d, nb, d, p.ca_out.w, d, dao2_i, nullptr, nullptr, 1.f));   // d/dao2, kept for the deferred dV
        RC(mansy_launch_attn_bwd_dq(e.qc + o * d, mkv, mkv + d, e.P2 + o * H * M, dao2_i, e.dqc + o * d, e.dS2 + o * H * M,
                                    e.Pk2 + o * H * M, cross_shape(nb), dr(site_dec(l, i, 2), c.p_drop, (size_t)b0 * H * M), st, W.fbase, im(W.fbase), 1, s16 ? 1 : 0));
      } else {
        float* dmkv = e.dmemkv + (size_t)b0 * M * 2 * d;
        RC(lin_dx(e.dbr2 + o * d, nb, d, p.ca_out.w, d, gt, nullptr, nullptr, 1.f));       // gt = d/dao2
        RC(mansy_launch_attn_bwd(e.qc + o * d, mkv, mkv + d, e.P2 + o * H * M, gt, e.dqc + o * d, dmkv, dmkv + d,
                                 cross_shape(nb), dr(site_dec(l, i, 2), c.p_drop, (size_t)b0 * H * M), 1, st));
      }
      RC(lin_dx(e.dqc + o * d, nb, d, p.ca_in.w, d, gt, gz, nullptr, 1.f));                // gt = d/dy1
      // norm1( x + drop(sa_out(ao1)) )
      RC(ln_bwd_dec(gt, e.z1 + o * d, e.m1 + o, e.r1 + o, p.n1, gz, e.dbr1 + o * d, dr(site_dec(l, i, 1), c.p_drop, bd), 3 * l + 0));
      float* dqkv_i = e.dqkv + o * 3 * d;
      const float* kv0 = e.qkv + (size_t)b0 * 3 * d;
      float* dkv0 = e.dqkv + (size_t)b0 * 3 * d;
      if (pull_self) {
        RC(lin_dx(e.dbr1 + o * d, nb, d, p.sa_out.w, d, e.dao1 + o * d, nullptr, nullptr, 1.f));   // d/dao1, kept: later rows of dV pull it
        // coefficient rows of this half: [T][nb*H][T], its own region of the [T][B*H][T] buffers
        const size_t co = (size_t)b0 * T * H * T;
        RC(mansy_launch_attn_bwd_selfpull(kv0, (long long)B * 3 * d, kv0 + d, kv0 + 2 * d, e.P1 + o * H * T, e.dao1 + (size_t)b0 * d, (long long)B * d,
                                          dqkv_i, dkv0 + d, dkv0 + 2 * d, e.dS1 + co, e.Pk1 + co, self_shape(i, nb), T, i,
                                          dr(site_dec(l, i, 0), c.p_drop, (size_t)b0 * H * (i + 1)), st, W.fbase, im(W.fbase), 1, s16 ? 1 : 0));
      } else {
        RC(lin_dx(e.dbr1 + o * d, nb, d, p.sa_out.w, d, gt, nullptr, nullptr, 1.f));       // gt = d/dao1
        RC(mansy_launch_attn_bwd(qkv_at(e, o), kv0 + d, kv0 + 2 * d, e.P1 + o * H * T, gt, dqkv_i, dkv0 + d, dkv0 + 2 * d,
                                 self_shape(i, nb), dr(site_dec(l, i, 0), c.p_drop, (size_t)b0 * H * (i + 1)), 1, st));
      }
      RC(lin_dx(dqkv_i, nb, 3 * d, p.sa_in.w, d, gx, gz, nullptr, 1.f));                   // gx = d/d(layer input)
    }
    // embedding of the fed-back token: dE slab (masked) + gradient wrt pred_{i-1}; fused into the head of step i-1, except
    // for step 0 (its token is the observed current position: only the masked gradient for the deferred dW is needed)
    if (!fuse_head) return mansy_launch_embed_bwd(gx, P.emb.w, W.dE_all + o * d, s_tok, C6, nb, d, dr(site_pe_tgt(i), c.p_pe, bd), st);
    if (i == 0) return mansy_launch_embed_bwd(gx, P.emb.w, W.dE_all + o * d, nullptr, C6, nb, d, dr(site_pe_tgt(0), c.p_pe, bd), st);
    return MANSY_OK;
  }
  const float* qkv_at(const DecBuf& e, size_t o) const { return e.qkv + o * 3 * d; }

  int backward(const float* src, const float* dpred_bt) {
    if (wtab.n == 0) RC(prepare_planes());      // separate mansy_vp_backward call: the weights have not changed since the forward
    const float ms = drop_scale;
    // Cross-attention K/V gradients: every step attends to the same memory rows, so (where the 4-heads-per-wave kernels
    // apply) the steps only record their coefficients and ONE pass per layer forms dK/dV -- instead of T read-modify-write
    // passes over the [B*M, 2d] gradient rows.
    const bool defer_cross = mansy_attn_deferred_kv_ok(cross_shape(), T) != 0;
    // Self-attention K/V gradients in pull form (same condition): row i of the gradient slab is written once, complete, by
    // the call for step i -- no read-modify-write of rows 0..i at every step, no zero-fill of the slab.
    const bool pull_self = mansy_attn_selfpull_ok(self_shape(T - 1), T) != 0;
    for (int l = 0; l < c.n_dec; ++l) {
      if (!pull_self) MANSY_HIP_CHECK(hipMemsetAsync(W.dec[l].dqkv, 0, sizeof(float) * (size_t)TB * 3 * d, st));
      if (!defer_cross) MANSY_HIP_CHECK(hipMemsetAsync(W.dec[l].dmemkv, 0, sizeof(float) * (size_t)B * M * 2 * d, st));
    }
    const bool ln_parts = mansy_ln_bwd_partial_ok(d);
    const size_t lnp_set = (size_t)mansy_ln_bwd_parts(B) * 2 * d;
    const int n_ln = 3 * c.n_dec + 1;
    // two halves on two streams (see forward): each half accumulates its LayerNorm weight-gradient sums in its own slot sets
    const bool split = split_ok();
    const int h0 = split ? B / 2 : B;
    if (ln_parts) MANSY_HIP_CHECK(hipMemsetAsync(W.lnp_dec, 0, sizeof(float) * (size_t)(split ? 2 : 1) * n_ln * lnp_set, st));
    // fused head of a backward step (dec_step.hip): embedding backward of step i+1 + predictor backward + final norm backward
    // + LayerNorm3 backward of the last layer in one launch
    const bool fuse_head = ln_parts && mansy_dec_tail_ok(d, C6) != 0;
    if (split) RC(fork());
    for (int i = T - 1; i >= 0; --i) {
      RC(dec_bwd_step(i, 0, h0, 0, dpred_bt, defer_cross, pull_self, ln_parts, fuse_head));
      if (split) {
        hipStream_t keep = st; st = st2;
        const int rc = dec_bwd_step(i, h0, B - h0, 1, dpred_bt, defer_cross, pull_self, ln_parts, fuse_head);
        st = keep; RC(rc);
      }
    }
    if (split) RC(join());
    if (ln_parts) {   // decoder LayerNorm weight gradients: slot sets -> gradients (both halves' sets add into the same gradient); queued for the ONE
      // reduce launch at the end of the backward (with the encoder's sets: round 5, was 7 / 14 + 5 launches)
      for (int half = 0; half < (split ? 2 : 1); ++half) {
        const int np = mansy_ln_bwd_parts(half ? B - h0 : h0);
        float* base = W.lnp_dec + (size_t)half * n_ln * lnp_set;
        for (int l = 0; l < c.n_dec; ++l) {
          ln_pending[ln_n_pending++] = MansyLnReduce{base + (size_t)(3 * l + 0) * lnp_set, np, P.dec[l].n1.gw, P.dec[l].n1.gb};
          ln_pending[ln_n_pending++] = MansyLnReduce{base + (size_t)(3 * l + 1) * lnp_set, np, P.dec[l].n2.gw, P.dec[l].n2.gb};
          ln_pending[ln_n_pending++] = MansyLnReduce{base + (size_t)(3 * l + 2) * lnp_set, np, P.dec[l].n3.gw, P.dec[l].n3.gb};
        }
        ln_pending[ln_n_pending++] = MansyLnReduce{base + (size_t)(3 * c.n_dec) * lnp_set, np, P.dec_norm.gw, P.dec_norm.gb};
      }
      RC(flush_ln());      // here: the decoder-side gradients must be final before the hook below (which = 2) hands them to the all-reduce
    }
    // ---- deferred decoder weight gradients: one reduce-dim-(T*B) GEMM per weight
    // They are LEAVES of the backward (only the optimizer -- or, data parallel, the gradient hook below -- reads them), while the memory gradient, the
    // DistillLayer and the encoder backward that follow are one dependent chain: with the two-stream schedule on and no hook to feed, the leaves go to the
    // side stream and run beside that chain (their split-K atomics' tails and the chain's launch gaps fill each other); joined at the end of the backward.
    // Every product still adds into a gradient buffer of its own, so nothing about the sums changes.
    const bool side_dw = split && c.bn_sync_world <= 1;
    if (side_dw) RC(fork());
    {
      hipStream_t keep = st;
      if (side_dw) st = st2;
      int rc = mansy_launch_outer_reduce(W.dz_all, C6, W.dec_out, TB, d, P.pred.gw, 0, nullptr, P.pred.gb, st);
      if (!rc) rc = mansy_launch_outer_reduce(W.tok_all, C6, W.dE_all, TB, d, P.emb.gw, 1, P.emb.gb, nullptr, st);
      for (int l = 0; l < c.n_dec && !rc; ++l) {
        const DecLayerP& p = P.dec[l]; DecBuf& e = W.dec[l];
        const float* x_all = l == 0 ? W.emb_all : W.dec[l - 1].y3;
        rc = lin_dw(e.dqkv, x_all, TB, 3 * d, d, p.sa_in.gw, p.sa_in.gb);
        if (!rc) rc = lin_dw(e.dbr1, e.ao1, TB, d, d, p.sa_out.gw, p.sa_out.gb);
        if (!rc) rc = lin_dw(e.dqc, e.y1, TB, d, d, p.ca_in.gw, p.ca_in.gb);
        if (!rc) rc = lin_dw(e.dbr2, e.ao2, TB, d, d, p.ca_out.gw, p.ca_out.gb);
        if (!rc) rc = lin_dw(e.da, e.y2, TB, f, d, p.lin1.gw, p.lin1.gb);
        if (!rc) rc = lin_dw(e.dbr3, e.h, TB, d, f, p.lin2.gw, p.lin2.gb);
      }
      st = keep;
      RC(rc);
    }
    for (int l = 0; l < c.n_dec; ++l) {      // the chain: K/V gradients of the memory -> their projection's gradients -> the memory gradient
      const DecLayerP& p = P.dec[l]; DecBuf& e = W.dec[l];
      if (defer_cross)
        RC(mansy_launch_attn_kvgrad(e.qc, (long long)B * d, e.dao2, (long long)B * d, e.dS2, e.Pk2, e.dmemkv, e.dmemkv + d, cross_shape(), T, 0, st, W.fbase, im(W.fbase)));
      RC(lin_dw(e.dmemkv, W.mem, B * M, 2 * d, d, p.ca_in.gw + (size_t)d * d, p.ca_in.gb ? p.ca_in.gb + d : nullptr, defer_cross));      // (the per-step accumulating form of the K/V gradients keeps floats only)
      RC(lin_dx(e.dmemkv, B * M, 2 * d, p.ca_in.w + (size_t)d * d, d, W.dmem, l == 0 ? nullptr : W.dmem, nullptr, 1.f, 0, defer_cross));
    }
    // ---- DistillLayer
    DistillShape ds = {B, S, M, d, c.bn_sync_world > 1 ? c.bn_sync_world : 1, c.bn_sync_fn, c.bn_sync_user};
    RC(mansy_launch_distill_bwd(W.conv, W.dmem, W.argmax, P.bn.w, P.bn.b, W.bn_mean, W.bn_rstd, W.g_a, W.g_b, P.bn.gw, P.bn.gb, W.stats,
                                ds, st, W.dis_part, im(W.g_b)));
    RC(lin_dw(W.g_b, W.col, N, d, 3 * d, P.conv.gw, P.conv.gb));
    RC(lin_dx(W.g_b, N, d, P.conv.w, 3 * d, W.g_wide, nullptr, nullptr, 1.f));
    RC(mansy_launch_col2im3(W.g_wide, W.g_a, B, S, d, st));
    // Data parallel: every gradient from the first decoder layer to the end of the parameter table (decoder layers, decoder
    // norm, DistillLayer conv + BatchNorm, predictor -- two thirds of the flat buffer) is final here.  The host hook (which = 2)
    // starts their all-reduce on its side stream / second communicator, under the encoder backward that follows.
    if (c.bn_sync_world > 1) RC(mansy_bn_sync_invoke(2, c.bn_sync_fn, c.bn_sync_user));
    // ---- encoder
    const float* last = W.enc[c.n_enc - 1].y2;
    RC(ln_bwd(W.g_a, last, W.me, W.re, P.enc_norm, W.g_b, nullptr, mansy_no_drop(), N, -1));
    float* gx = W.g_b; float* gz = W.g_a; float* gt = W.g_c;
    for (int l = c.n_enc - 1; l >= 0; --l) {
      const EncLayerP& p = P.enc[l]; EncBuf& e = W.enc[l];
      const float* x_in = l == 0 ? W.x0 : W.enc[l - 1].y2;
      RC(ln_bwd(gx, e.z2, e.m2, e.r2, p.n2, gz, gt, dr(site_enc(l, 3), c.p_drop), N, -(2 + 2 * l)));        // gt = d/d(lin2 out)
      RC(lin_dw(gt, e.h, N, d, f, p.lin2.gw, p.lin2.gb));
      RC(lin_dx(gt, N, d, p.lin2.w, f, W.g_ff, nullptr, e.h, ms, 2));                          // g_ff = d/d(lin1 pre-act): an operand of two products only (image only)
      RC(lin_dw(W.g_ff, e.y1, N, f, d, p.lin1.gw, p.lin1.gb));
      RC(lin_dx(W.g_ff, N, f, p.lin1.w, d, gt, gz, nullptr, 1.f));                             // gt = d/dy1
      RC(ln_bwd(gt, e.z1, e.m1, e.r1, p.n1, gz, gx, dr(site_enc(l, 1), c.p_drop), N, -(3 + 2 * l)));         // gx = d/d(out_proj out)
      RC(lin_dw(gx, e.ao, N, d, d, p.out_proj.gw, p.out_proj.gb));
      RC(lin_dx(gx, N, d, p.out_proj.w, d, gt, nullptr, nullptr, 1.f));                        // gt = d/dao
      RC(mansy_launch_attn_bwd(e.qkv, e.qkv + d, e.qkv + 2 * d, e.P, gt, W.g_wide, W.g_wide + d, W.g_wide + 2 * d, enc_shape(),
                               dr(site_enc(l, 0), c.p_drop), 0, st, W.fbase, im(W.fbase), 1));
      RC(lin_dw(W.g_wide, x_in, N, 3 * d, d, p.in_proj.gw, p.in_proj.gb));
      RC(lin_dx(W.g_wide, N, 3 * d, p.in_proj.w, d, gx, gz, nullptr, 1.f));                    // gx = d/d(layer input)
    }
    RC(mansy_launch_embed_bwd(gx, P.emb.w, gt, nullptr, C6, N, d, dr(site_pe_src(), c.p_pe), st));
    RC(mansy_launch_outer_reduce(src, C6, gt, N, d, P.emb.gw, 1, P.emb.gb, nullptr, st));
    RC(flush_ln());      // the encoder's five LayerNorm weight / bias gradients: one launch
    if (side_dw) RC(join());      // the decoder's weight gradients, taken beside everything since the decoder recurrence
    return MANSY_OK;
  }
};

int setup(const mansy_vp_config* cfg, void* workspace, Layout& L, Work& W) {
  RC(check_cfg(cfg));
  L.base = (char*)workspace;
  build_layout(*cfg, L, W);
  return MANSY_OK;
}

}  // namespace

extern "C" {

int mansy_vp_num_params(const mansy_vp_config* cfg) {
  if (check_cfg(cfg)) return MANSY_EINVAL;
  return (int)param_table(*cfg).size();
}

int mansy_vp_param_info(const mansy_vp_config* cfg, int idx, char* name, int name_len, long long* numel, int* ndim, long long shape[4]) {
  RC(check_cfg(cfg));
  const std::vector<ParamInfo> t = param_table(*cfg);
  MANSY_REQUIRE(idx >= 0 && idx < (int)t.size(), "vp_param_info: index %d out of range", idx);
  if (name && name_len > 0) { strncpy(name, t[idx].name.c_str(), name_len - 1); name[name_len - 1] = 0; }
  if (numel) *numel = t[idx].numel;
  if (ndim) *ndim = t[idx].ndim;
  if (shape) for (int i = 0; i < 4; ++i) shape[i] = t[idx].shape[i];
  return MANSY_OK;
}

size_t mansy_vp_workspace_bytes(const mansy_vp_config* cfg) {
  Layout L; Work W;
  if (setup(cfg, nullptr, L, W)) return 0;
  return L.total + 256;
}

int mansy_vp_ws_lookup(const mansy_vp_config* cfg, const char* name, long long* offset_bytes, long long* numel) {
  Layout L; Work W;
  RC(setup(cfg, nullptr, L, W));
  for (const BufInfo& b : L.bufs)
    if (b.name == name) { if (offset_bytes) *offset_bytes = (long long)b.off; if (numel) *numel = (long long)(b.bytes / 4); return MANSY_OK; }
  mansy_set_error("vp_ws_lookup: unknown buffer '%s'", name);
  return MANSY_EINVAL;
}

int mansy_vp_forward(const mansy_vp_config* cfg, const float* const* params, const float* pe, float* bn_running_mean,
                     float* bn_running_var, long long* bn_num_batches, const float* src, const float* cur, float* pred,
                     void* workspace, int train, uint32_t seed, void* stream) {
  MANSY_REQUIRE(params && pe && bn_running_mean && bn_running_var && src && cur && pred && workspace, "vp_forward: null pointer");
  Layout L; Work W;
  RC(setup(cfg, workspace, L, W));
  Eng e(*cfg, (hipStream_t)stream, train != 0, seed);
  e.W = W;
  bind_params(*cfg, params, nullptr, e.P);
  return e.forward(src, cur, pe, bn_running_mean, bn_running_var, bn_num_batches, pred);
}

int mansy_vp_backward(const mansy_vp_config* cfg, const float* const* params, float* const* grads, const float* src,
                      const float* dpred, void* workspace, uint32_t seed, void* stream) {
  MANSY_REQUIRE(params && grads && src && dpred && workspace, "vp_backward: null pointer");
  Layout L; Work W;
  RC(setup(cfg, workspace, L, W));
  Eng e(*cfg, (hipStream_t)stream, true, seed);
  e.W = W;
  bind_params(*cfg, params, grads, e.P);
  return e.backward(src, dpred);
}

int mansy_vp_sample(const mansy_vp_config* cfg, const float* const* params, const float* pe, float* bn_running_mean,
                    float* bn_running_var, const float* history, const float* current, float* out, void* workspace, void* stream) {
  MANSY_REQUIRE(params && pe && history && current && out && workspace, "vp_sample: null pointer");
  Layout L; Work W;
  RC(setup(cfg, workspace, L, W));
  hipStream_t st = (hipStream_t)stream;
  Eng e(*cfg, st, false, 0u);
  e.W = W;
  bind_params(*cfg, params, nullptr, e.P);
  const int c = cfg->in_ch / 3;
  RC(mansy_launch_mtio_mix(history, nullptr, nullptr, W.src6, cfg->B, cfg->S, c, st));
  RC(mansy_launch_mtio_mix(current, nullptr, nullptr, W.cur6, cfg->B, 1, c, st));
  RC(e.forward(W.src6, W.cur6, pe, bn_running_mean, bn_running_var, nullptr, W.pred_bt));
  return mansy_launch_ensemble_wrap(W.pred_bt, out, (long long)cfg->B * cfg->T, 3, c, st);
}

int mansy_vp_train_step(const mansy_vp_config* cfg, const float* const* params, float* const* grads, float* flat_p, float* flat_g,
                        float* flat_m, float* flat_v, long long n_flat, const float* pe, float* bn_running_mean,
                        float* bn_running_var, long long* bn_num_batches, const float* history, const float* current,
                        const float* future, const int* perm1, const int* perm2, float lr, float beta1, float beta2, float eps,
                        float weight_decay, int step, float* loss_out, void* workspace, uint32_t seed, void* stream) {
  MANSY_REQUIRE(params && grads && flat_p && flat_g && flat_m && flat_v && pe && history && current && future && loss_out && workspace,
                "vp_train_step: null pointer");
  MANSY_REQUIRE((perm1 == nullptr) == (perm2 == nullptr), "vp_train_step: perm1/perm2 must both be set or both NULL");
  Layout L; Work W;
  RC(setup(cfg, workspace, L, W));
  hipStream_t st = (hipStream_t)stream;
  Eng e(*cfg, st, true, seed);
  e.W = W;
  bind_params(*cfg, params, grads, e.P);
  const int c = cfg->in_ch / 3;
  RC(mansy_launch_mtio_mix3(history, current, future, perm1, perm2, W.src6, W.cur6, W.fut6, cfg->B, cfg->S, cfg->T, c, st));      // one launch (was three)
  MANSY_HIP_CHECK(hipMemsetAsync(flat_g, 0, sizeof(float) * (size_t)n_flat, st));
  RC(e.forward(W.src6, W.cur6, pe, bn_running_mean, bn_running_var, bn_num_batches, W.pred_bt));
  const long long n = (long long)cfg->B * cfg->T * cfg->in_ch;
  RC(mansy_launch_mtio_loss(W.pred_bt, W.fut6, n, 1.f / (2.f * (float)cfg->B * (float)cfg->T), W.loss_acc, loss_out, W.dpred_bt, st));
  RC(e.backward(W.src6, W.dpred_bt));
  if (step <= 0) return MANSY_OK;
  return mansy_launch_adamw(flat_p, flat_g, flat_m, flat_v, n_flat, lr, beta1, beta2, eps, weight_decay, step, 1, st);
}

}  // extern "C"
