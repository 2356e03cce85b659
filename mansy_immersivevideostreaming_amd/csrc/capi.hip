// C-ABI shims (include/mansy_hip.h) over the C++ launch API, plus error reporting.
#include <stdarg.h>
#include "mansy_kernels.h"
#include "../../include/mansy_hip.h"

static thread_local char g_err[1024] = "";

extern "C" void mansy_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" { unsigned long long g_mansy_launch_count = 0; }
extern "C" unsigned long long mansy_prof_launch_count(void) { return __atomic_load_n(&g_mansy_launch_count, __ATOMIC_RELAXED); }

int mansy_bn_sync_invoke(int which, int (*fn)(int, void*), void* user) {
  MANSY_REQUIRE(fn, "bn_sync_world > 1 but the call carries no hook (mansy_vp_config::bn_sync_fn)");
  const int rc = fn(which, user);
  MANSY_REQUIRE(rc == 0, "bn sync hook failed (%d)", rc);
  return MANSY_OK;
}

extern "C" {


const char* mansy_last_error(void) { return g_err; }
int mansy_abi_version(void) { return 9; }   // == _lib.py ABI_VERSION

int mansy_gemm_f32(const float* A, int lda, int a_kmajor, const float* B, int ldb, int b_kmajor, float* C, int ldc, int M, int N,
                   int K, const mansy_gemm_epilogue* ep, int force_tile, int force_splitk, void* stream) {
  GemmEpilogue e;
  if (ep) {
    e.bias = ep->bias; e.relu = ep->relu; e.mask_src = ep->mask_src; e.mask_ld = ep->mask_ld; e.mask_scale = ep->mask_scale;
    e.drop.p = ep->drop_p; e.drop.seed = ep->drop_seed; e.drop.site = ep->drop_site;
    e.resid = ep->resid; e.resid_ld = ep->resid_ld; e.accumulate = ep->accumulate;
    e.a_rowsum = ep->a_rowsum; e.prec = ep->prec; e.variant = ep->variant;
    MANSY_REQUIRE(!ep->a_rowsum || a_kmajor, "gemm: a_rowsum rides on a K-major A (dW = dY^T X)");
    MANSY_REQUIRE(ep->prec == 0 || ep->prec == 1 || ep->prec == 3 || ep->prec == 6, "gemm: prec must be MANSY_PREC_F32 (0), _BF16 (1), _BF16X3 (3) or _BF16X6 (6), got %d", ep->prec);
  }
  MANSY_REQUIRE(force_tile == 0 || force_tile == 64 || force_tile == 96 || force_tile == 128 || force_tile == -64 || force_tile == -128,
                "gemm: force_tile must be 0, 64, 96 (128x64), 128, or -64 / -128 (register-staged loop)");
  return mansy_launch_gemm_f32(A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, e, force_tile, force_splitk, (hipStream_t)stream);
}

int mansy_weight_planes(const float* W, int N, int K, uint16_t* out, uint16_t* out_t, long long plane_stride, int n_planes, void* stream) {
  MANSY_REQUIRE(W && N >= 1 && K >= 1 && plane_stride >= (long long)N * K, "weight_planes: bad arguments");
  MansyWPlaneTab tab; tab.n = 1; tab.w[0] = W; tab.N[0] = N; tab.K[0] = K; tab.off[0] = 0;
  return mansy_launch_weight_planes(tab, out, out_t, plane_stride, n_planes, (hipStream_t)stream);
}

int mansy_gemm_planes(const float* A, int lda, const float* B, int ldb, int b_kmajor, const uint16_t* planes, long long plane_stride, int planes_ld,
                      float* C, int ldc, int M, int N, int K, const mansy_gemm_epilogue* ep, int force_tile, void* stream) {
  GemmEpilogue e;
  if (ep) {
    e.bias = ep->bias; e.relu = ep->relu; e.mask_src = ep->mask_src; e.mask_ld = ep->mask_ld; e.mask_scale = ep->mask_scale;
    e.drop.p = ep->drop_p; e.drop.seed = ep->drop_seed; e.drop.site = ep->drop_site;
    e.resid = ep->resid; e.resid_ld = ep->resid_ld; e.accumulate = ep->accumulate; e.prec = ep->prec; e.variant = ep->variant;
    MANSY_REQUIRE(ep->prec == 0 || ep->prec == 1 || ep->prec == 3 || ep->prec == 6, "gemm_planes: prec must be MANSY_PREC_F32 (0), _BF16 (1), _BF16X3 (3) or _BF16X6 (6), got %d", ep->prec);
  }
  MANSY_REQUIRE(force_tile == 0 || force_tile == 64 || force_tile == 96 || force_tile == 128 || force_tile == 256,
                "gemm_planes: force_tile must be 0, 64, 96 (128x64), 128 or 256 (256x128, bf16x3)");
  e.b_planes = planes; e.b_plane_stride = plane_stride; e.b_planes_ld = planes_ld;
  return mansy_launch_gemm_f32(A, lda, 0, B, ldb, b_kmajor, C, ldc, M, N, K, e, force_tile, 0, (hipStream_t)stream);
}

int mansy_gemm_bf16(const uint16_t* A16, int lda, int a_kmajor, const uint16_t* B16, int ldb, int b_kmajor, float* C, int ldc, uint16_t* C16, int ldc16,
                    int M, int N, int K, const mansy_gemm_epilogue* ep, const uint16_t* resid16, const uint16_t* mask16, int force_tile, int force_splitk, void* stream) {
  GemmEpilogue e;
  if (ep) {
    e.bias = ep->bias; e.relu = ep->relu; e.mask_src = ep->mask_src; e.mask_ld = ep->mask_ld; e.mask_scale = ep->mask_scale;
    e.drop.p = ep->drop_p; e.drop.seed = ep->drop_seed; e.drop.site = ep->drop_site;
    e.resid = ep->resid; e.resid_ld = ep->resid_ld; e.accumulate = ep->accumulate; e.a_rowsum = ep->a_rowsum;
  }
  MANSY_REQUIRE(A16 && B16, "gemm_bf16: null operand");
  MANSY_REQUIRE((!resid16 && !mask16) || (ep && !a_kmajor), "gemm_bf16: resid16 / mask16 belong to the forward form and take their leading dimensions from ep");
  if (resid16) { MANSY_REQUIRE((reinterpret_cast<uintptr_t>(resid16) & 7) == 0 && ep->resid_ld % 4 == 0, "gemm_bf16: resid16 must be 8-byte aligned, resid_ld %% 4 == 0"); e.resid16 = resid16; e.resid = nullptr; }
  if (mask16) { MANSY_REQUIRE((reinterpret_cast<uintptr_t>(mask16) & 7) == 0 && ep->mask_ld % 4 == 0, "gemm_bf16: mask16 must be 8-byte aligned, mask_ld %% 4 == 0"); e.mask16 = mask16; e.mask_src = nullptr; }
  MANSY_REQUIRE(force_tile == 0 || force_tile == 64 || force_tile == 96 || force_tile == 128, "gemm_bf16: force_tile must be 0, 64, 96 (128x64) or 128");
  e.prec = 1; e.a16 = A16; e.a16_ld = lda; e.c16 = C16; e.c16_ld = ldc16;
  if (a_kmajor && b_kmajor) { e.b16 = B16; e.b16_ld = ldb; }
  else { MANSY_REQUIRE(!a_kmajor && !b_kmajor, "gemm_bf16: forms are (K-contiguous A, B [N, K] K-contiguous) and (K-major A, K-major B)"); e.b_planes = B16; e.b_planes_ld = ldb; e.b_plane_stride = 0; }
  return mansy_launch_gemm_bf16a(a_kmajor, b_kmajor, C, ldc, M, N, K, e, force_tile, force_splitk, (hipStream_t)stream);
}

static AttnShape to_shape(const mansy_attn_shape* s) {
  AttnShape a;
  a.nb = s->nb; a.H = s->H; a.Lq = s->Lq; a.Lk = s->Lk; a.dh = s->dh;
  a.q_bs = s->q_bs; a.q_rs = s->q_rs; a.k_bs = s->k_bs; a.k_rs = s->k_rs; a.v_bs = s->v_bs; a.v_rs = s->v_rs;
  a.o_bs = s->o_bs; a.o_rs = s->o_rs; a.scale = s->scale;
  return a;
}

int mansy_attn_fwd(const float* Q, const float* K, const float* V, float* O, float* P_save, const mansy_attn_shape* s, float drop_p,
                   uint32_t seed, uint32_t site, void* stream) {
  MANSY_REQUIRE(s, "attn_fwd: null shape");
  MansyDrop d = {drop_p, seed, site};
  return mansy_launch_attn_fwd(Q, K, V, O, P_save, to_shape(s), d, (hipStream_t)stream);
}
int mansy_attn_bwd(const float* Q, const float* K, const float* V, const float* P_save, const float* dO, float* dQ, float* dK,
                   float* dV, const mansy_attn_shape* s, float drop_p, uint32_t seed, uint32_t site, int accum_kv, void* stream) {
  MANSY_REQUIRE(s, "attn_bwd: null shape");
  MansyDrop d = {drop_p, seed, site};
  return mansy_launch_attn_bwd(Q, K, V, P_save, dO, dQ, dK, dV, to_shape(s), d, accum_kv, (hipStream_t)stream);
}
int mansy_layernorm_fwd(const float* a, const float* b, const float* w, const float* bias, float* z_out, float* y, float* mean,
                        float* rstd, int rows, int C, float eps, void* stream) {
  return mansy_launch_layernorm_fwd(a, b, w, bias, z_out, y, mean, rstd, rows, C, eps, (hipStream_t)stream);
}
int mansy_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* w, float* dz,
                        float* dz_drop, float drop_p, uint32_t seed, uint32_t site, float* dw, float* dbias, int rows, int C,
                        void* stream) {
  MansyDrop d = {drop_p, seed, site};
  return mansy_launch_layernorm_bwd(dy, z, mean, rstd, w, dz, dz_drop, d, dw, dbias, rows, C, (hipStream_t)stream);
}

int mansy_layernorm_bwd_parts(int rows) { return mansy_ln_bwd_parts(rows); }
int mansy_layernorm_bwd_partial(const float* dy, const float* z, const float* mean, const float* rstd, const float* w, float* dz,
                                float* dz_drop, float drop_p, uint32_t seed, uint32_t site, float* partials, int accumulate,
                                int rows, int C, void* stream) {
  MansyDrop d = {drop_p, seed, site};
  return mansy_launch_layernorm_bwd_partial(dy, z, mean, rstd, w, dz, dz_drop, d, partials, accumulate, rows, C, (hipStream_t)stream);
}
int mansy_ln_partials_reduce(const float* partials, int nparts, int C, float* dw, float* dbias, void* stream) {
  return mansy_launch_ln_partials_reduce(partials, nparts, C, dw, dbias, (hipStream_t)stream);
}
int mansy_attn_bwd_dq(const float* Q, const float* K, const float* V, const float* P_save, const float* dO, float* dQ,
                      float* dS_out, float* Pk_out, const mansy_attn_shape* s, float drop_p, uint32_t seed, uint32_t site,
                      void* stream) {
  MANSY_REQUIRE(s, "attn_bwd_dq: null shape");
  MansyDrop d = {drop_p, seed, site};
  return mansy_launch_attn_bwd_dq(Q, K, V, P_save, dO, dQ, dS_out, Pk_out, to_shape(s), d, (hipStream_t)stream);
}
int mansy_attn_kvgrad(const float* Q_all, long long q_ts, const float* dO_all, long long o_ts, const float* dS_all,
                      const float* Pk_all, float* dK, float* dV, const mansy_attn_shape* s, int T, int accum, void* stream) {
  MANSY_REQUIRE(s, "attn_kvgrad: null shape");
  return mansy_launch_attn_kvgrad(Q_all, q_ts, dO_all, o_ts, dS_all, Pk_all, dK, dV, to_shape(s), T, accum, (hipStream_t)stream);
}

int mansy_attn_bwd_selfpull(const float* Q_all, long long q_ts, const float* K, const float* V, const float* P_save,
                            const float* dO_all, long long o_ts, float* dQ, float* dK, float* dV, float* dS_all, float* Pk_all,
                            const mansy_attn_shape* s, int T, int step, float drop_p, uint32_t seed, uint32_t site, void* stream) {
  MANSY_REQUIRE(s, "attn_bwd_selfpull: null shape");
  MansyDrop d = {drop_p, seed, site};
  return mansy_launch_attn_bwd_selfpull(Q_all, q_ts, K, V, P_save, dO_all, o_ts, dQ, dK, dV, dS_all, Pk_all, to_shape(s), T, step, d,
                                        (hipStream_t)stream);
}

int mansy_mtio_mix(const float* x, const int* perm1, const int* perm2, float* out, int B, int L, int c, void* stream) {
  return mansy_launch_mtio_mix(x, perm1, perm2, out, B, L, c, (hipStream_t)stream);
}
int mansy_mtio_loss_fwd_bwd(const float* pred, const float* gt, int B, int T, int C, double* scratch, float* loss_out, float* dpred,
                            void* stream) {
  MANSY_REQUIRE(B >= 1 && T >= 1 && C >= 1, "mtio_loss: bad shape");
  return mansy_launch_mtio_loss(pred, gt, (long long)B * T * C, 1.f / (2.f * (float)B * (float)T), scratch, loss_out, dpred,
                                (hipStream_t)stream);
}
int mansy_adamw_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, int step, int decoupled, void* stream) {
  return mansy_launch_adamw(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, decoupled, (hipStream_t)stream);
}
int mansy_linreg_sample(const float* history, const float* current, int B, int S, int T, int c, float* out, void* stream) {
  return mansy_launch_linreg_sample(history, current, B, S, T, c, out, (hipStream_t)stream);
}
int mansy_ensemble_wrap(const float* pred, float* out, long long rows, int heads, int c, void* stream) {
  return mansy_launch_ensemble_wrap(pred, out, rows, heads, c, (hipStream_t)stream);
}

int mansy_traj_gather(const float* table, int L, int c, const int* idx, int B, int S, int T, float* hist, float* cur, float* fut, void* stream) {
  return mansy_launch_traj_gather(table, L, c, idx, B, S, T, hist, cur, fut, (hipStream_t)stream);
}
int mansy_periodic_mse(const float* a, const float* b, long long rows, int c, float* out, void* stream) {
  return mansy_launch_periodic_mse(a, b, rows, c, out, (hipStream_t)stream);
}
int mansy_tilemap_metrics(const uint64_t* gt, const uint64_t* pred, long long n, double* out, void* stream) {
  return mansy_launch_tilemap_metrics((const unsigned long long*)gt, (const unsigned long long*)pred, n, out, (hipStream_t)stream);
}

int mansy_tilemap(const float* xy, long long n, int W, int H, int tile_num_w, int tile_num_h, int fov_w, int fov_h, uint64_t* maps,
                  void* stream) {
  return mansy_launch_tilemap(xy, n, W, H, tile_num_w, tile_num_h, fov_w, fov_h, (unsigned long long*)maps, (hipStream_t)stream);
}
int mansy_tilemap_iou(const uint64_t* a, const uint64_t* b, long long n, double* iou, void* stream) {
  return mansy_launch_tilemap_iou((const unsigned long long*)a, (const unsigned long long*)b, n, iou, (hipStream_t)stream);
}
int mansy_tilemap_or_groups(const uint64_t* maps, long long ngroups, int group, uint64_t* out, void* stream) {
  return mansy_launch_tilemap_or_groups((const unsigned long long*)maps, ngroups, group, (unsigned long long*)out, (hipStream_t)stream);
}

}  // extern "C"
