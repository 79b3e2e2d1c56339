// gpsr_api.hip -- C ABI (include/gpsr.h) over the gfx950 kernels.  No torch types,
// no allocation: every buffer, the workspace included, belongs to the caller.
#include "gpsr_kernels.hpp"
#include "../../include/gpsr.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[256] = "";

static int hip_fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return -2;
}
static int arg_fail(const char* what) {
  snprintf(g_err, sizeof(g_err), "bad argument: %s", what);
  return -1;
}
#define CHECK_LAUNCH(where)                                 \
  do {                                                      \
    hipError_t e_ = hipGetLastError();                      \
    if (e_ != hipSuccess) return hip_fail(e_, where);       \
  } while (0)

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// ---------------------------------------------------------------- kernel timing (bench.py roofline)
// Optional HIP-event brackets around tagged launches, on the stream the kernel is
// launched on.  Off by default; costs two hipEventRecord per tagged launch when on.
#include <vector>
enum { T_UPDATE = 0, T_PANEL, T_PIVOT, T_BUILD, T_GRAD1, T_GRAD2, T_PREDICT, T_UPDATE_LAST, T_UPDATE2, T_UPDATE_SEL, T_NTAGS };
struct ProfRec { hipEvent_t a, b; int tag; };
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;
static bool g_prof_on = false;

static inline ProfRec* prof_start(int tag, hipStream_t st) {
  if (!g_prof_on || g_prof_used >= g_prof.size()) return nullptr;
  ProfRec* r = &g_prof[g_prof_used++];
  r->tag = tag;
  (void)hipEventRecord(r->a, st);
  return r;
}
static inline void prof_stop(ProfRec* r, hipStream_t st) { if (r) (void)hipEventRecord(r->b, st); }

extern "C" int gpsr_profile_begin(int max_records) {
  if (max_records < 1 || max_records > (1 << 20)) return arg_fail("max_records");
  while ((int)g_prof.size() < max_records) {
    ProfRec r; r.tag = -1;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return hip_fail(hipGetLastError(), "hipEventCreate");
    g_prof.push_back(r);
  }
  g_prof_used = 0;
  g_prof_on = true;
  return 0;
}

// Stops recording, waits for the recorded events and returns per-tag launch counts and
// total milliseconds (arrays of GPSR_PROFILE_NTAGS entries).
extern "C" int gpsr_profile_end(int32_t* counts, double* total_ms) {
  g_prof_on = false;
  if (!counts || !total_ms) return arg_fail("null pointer");
  for (int t = 0; t < T_NTAGS; ++t) { counts[t] = 0; total_ms[t] = 0.0; }
  for (size_t i = 0; i < g_prof_used; ++i) {
    hipError_t e = hipEventSynchronize(g_prof[i].b);
    if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b);
    if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
    counts[g_prof[i].tag] += 1;
    total_ms[g_prof[i].tag] += (double)ms;
  }
  g_prof_used = 0;
  return 0;
}

extern "C" const char* gpsr_version(void) { return "gpsr 0.1 gfx950"; }
extern "C" const char* gpsr_last_error(void) { return g_err; }
extern "C" int gpsr_padded_n(int n) { return ((n + NB - 1) / NB) * NB; }

// ---------------------------------------------------------------- gather
extern "C" int gpsr_gather(const double* train_in, const double* train_tgt, const double* test_in,
                           int H, int W, int s, int p, const int32_t* locs, int B,
                           const int32_t* tidx, int nt, double* X, double* y, double* Xs, void* stream) {
  if (B <= 0) return 0;
  if (!train_in || !train_tgt || !test_in || !locs || !tidx || !X || !y || !Xs) return arg_fail("null pointer");
  if (p < 3 || s < 1 || p > H || p > W || nt < 1 || nt > p - 2) return arg_fail("patch geometry");
  const int n = nt * nt, m = (p * s - 2) * (p * s - 2);
  const int total = 9 * n + 8 * m;
  int gx = (total + 255) / 256; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_gather, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, train_in, train_tgt, test_in,
                     H, W, s, p, locs, tidx, nt, X, y, Xs);
  CHECK_LAUNCH("k_gather");
  return 0;
}

// ---------------------------------------------------------------- init
static float inv_softplus_host(float x) { return x + logf(-expm1f(-x)); }

extern "C" int gpsr_init_hypers(const double* y, int B, int n, int kernel_id, float init_raw_os,
                                float init_raw_ls, const float* carry, float* raw, void* stream) {
  if (B <= 0) return 0;
  if (!y || !raw || n < 2) return arg_fail("init_hypers");
  if (kernel_id < 0 || kernel_id > 4) return arg_fail("kernel_id");
  const float raw_noise0 = inv_softplus_host(0.001f - 1e-4f);
  const float raw_ls_rbf = inv_softplus_host(0.223f);
  hipLaunchKernelGGL(k_init_hypers, dim3(B), dim3(256), 0, (hipStream_t)stream, y, n, kernel_id, raw_noise0,
                     raw_ls_rbf, init_raw_os, init_raw_ls, carry, raw);
  CHECK_LAUNCH("k_init_hypers");
  return 0;
}

// ---------------------------------------------------------------- covariance
static size_t build_lds(int d) { return (size_t)2 * NB * (d + 1) * sizeof(double); }

static int launch_build(const double* X, int B, int n, int N, int d, int kid, const float* raw, double* A,
                        double* logdet, hipStream_t st, int full_diag) {
  const int nblk = N / NB, ntri = nblk * (nblk + 1) / 2, total = ntri * B;
  const size_t lds = build_lds(d);
#define BUILD(K)                                                                                              \
  do {                                                                                                        \
    if (d == 8) hipLaunchKernelGGL((k_build8<K>), dim3(total), dim3(MM_THREADS), 0, st, X, raw, A, n, N, ntri, total, logdet, full_diag); \
    else hipLaunchKernelGGL((k_build<K>), dim3(total), dim3(256), lds, st, X, raw, A, n, N, d, ntri, total, logdet);           \
  } while (0)
  ProfRec* pr = prof_start(T_BUILD, st);
  switch (kid) {
    case K_RBF: BUILD(K_RBF); break;
    case K_M12: BUILD(K_M12); break;
    case K_M32: BUILD(K_M32); break;
    case K_M52: BUILD(K_M52); break;
    default: BUILD(K_LIN); break;
  }
  prof_stop(pr, st);
#undef BUILD
  CHECK_LAUNCH("k_build");
  return 0;
}

extern "C" int gpsr_covariance(const double* X, int B, int n, int d, int kernel_id, const float* raw,
                               double* K, void* stream) {
  if (B <= 0) return 0;
  if (!X || !raw || !K) return arg_fail("null pointer");
  if (n < 1 || d < 1 || d > 64 || kernel_id < 0 || kernel_id > 4) return arg_fail("covariance shape");
  return launch_build(X, B, n, gpsr_padded_n(n), d, kernel_id, raw, K, nullptr, (hipStream_t)stream, 1);
}

// ---------------------------------------------------------------- inversion
// Wo2 / U[1] / M[1]: second pivot of a two-pivot step (k_update2)
struct InvWs { double *Wo, *Wo2, *U[2], *M[2]; };

static size_t inv_ws_bytes(int B, int N) {
  return 2 * align256((size_t)B * NB * N * 8) + 4 * align256((size_t)B * NB * NB * 8);
}
static InvWs inv_ws_carve(char* base, int B, int N) {
  InvWs w;
  w.Wo = (double*)base; base += align256((size_t)B * NB * N * 8);
  w.Wo2 = (double*)base; base += align256((size_t)B * NB * N * 8);
  for (int i = 0; i < 2; ++i) {
    w.U[i] = (double*)base; base += align256((size_t)B * NB * NB * 8);
    w.M[i] = (double*)base; base += align256((size_t)B * NB * NB * 8);
  }
  return w;
}

static const bool g_pair_steps = getenv("GPSR_SINGLE_STEPS") == nullptr;   // debugging switch: one pivot per pass

static int launch_pivot(double* A, int B, int N, int k, double* U, double* M, double* logdet, int32_t* flags,
                        hipStream_t st) {
  ProfRec* pr = prof_start(T_PIVOT, st);
  hipLaunchKernelGGL(k_pivot, dim3(B), dim3(MM_THREADS), PIVOT_LDS_BYTES, st, A, N, k, U, M, logdet, flags);
  prof_stop(pr, st);
  CHECK_LAUNCH("k_pivot");
  return 0;
}
static int launch_panel(const double* A, int B, int N, int k, const double* U, double* Wo, hipStream_t st) {
  const int nblk = N / NB, tp = (nblk - 1) * B;
  ProfRec* pr = prof_start(T_PANEL, st);
  hipLaunchKernelGGL(k_panel, dim3(tp), dim3(MM_THREADS), 0, st, A, N, nblk, k, U, Wo, tp);
  prof_stop(pr, st);
  CHECK_LAUNCH("k_panel");
  return 0;
}

// y / raw / part non-null: the last update launch also leaves the mat-vec partials of
// alpha = K^-1 (y - c) (grad1_partials), n = number of real rows.
// sweeps every pivot block of B matrices; logdet must have been zeroed
static int launch_sweeps(double* A, int B, int N, const InvWs& w, double* logdet, int32_t* flags, hipStream_t st,
                         const double* y = nullptr, const float* raw = nullptr, int n = 0, double* part = nullptr) {
  const int nblk = N / NB;
  static bool lds_attr_set = false;
  if (!lds_attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_pivot, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)PIVOT_LDS_BYTES);
    if (e != hipSuccess) return hip_fail(e, "pivot lds attr");
    lds_attr_set = true;
  }
  const int ntri1 = (nblk - 1) * nblk / 2;
  // pivots are taken two at a time while at least one single step remains for the end (the alpha epilogue
  // lives in the single-step kernel): nblk odd -> (nblk-1)/2 pairs + 1 single, even -> (nblk-2)/2 pairs + 2
  const int npairs = (g_pair_steps && nblk >= 3) ? (nblk - 1 - (nblk % 2 == 0 ? 1 : 0)) / 2 : 0;
  int rc;
  int k = 0;
  for (int m = 0; m < npairs; ++m, k += 2) {
    const int k1 = k, k2 = k + 1;
    if ((rc = launch_pivot(A, B, N, k1, w.U[0], w.M[0], logdet, flags, st))) return rc;
    if ((rc = launch_panel(A, B, N, k1, w.U[0], w.Wo, st))) return rc;
    {
      const int tu = nblk * B;   // the tiles of block row / column k2 only
      ProfRec* pr = prof_start(T_UPDATE_SEL, st);
      hipLaunchKernelGGL(k_update<false>, dim3(tu), dim3(MM_THREADS), 0, st, A, N, nblk, k1, w.Wo, w.M[0], ntri1, tu,
                         (const double*)nullptr, (const float*)nullptr, 0, (double*)nullptr, k2);
      prof_stop(pr, st);
      CHECK_LAUNCH("k_update(sel)");
    }
    if ((rc = launch_pivot(A, B, N, k2, w.U[1], w.M[1], logdet, flags, st))) return rc;
    if ((rc = launch_panel(A, B, N, k2, w.U[1], w.Wo2, st))) return rc;
    {
      const int nb2 = nblk - 2, per = nb2 * (nb2 + 1) / 2 + nb2 + 1 + nblk - 1, tu = per * B;
      ProfRec* pr = prof_start(T_UPDATE2, st);
      hipLaunchKernelGGL(k_update2, dim3(tu), dim3(MM_THREADS), 0, st, A, N, nblk, k1, w.Wo, w.Wo2, w.M[0], w.M[1], tu);
      prof_stop(pr, st);
      CHECK_LAUNCH("k_update2");
    }
  }
  for (; k < nblk; ++k) {
    if ((rc = launch_pivot(A, B, N, k, w.U[0], w.M[0], logdet, flags, st))) return rc;
    if (nblk > 1) {
      if ((rc = launch_panel(A, B, N, k, w.U[0], w.Wo, st))) return rc;
      const int tu = (ntri1 + nblk - 1) * B;
      const bool last = (k == nblk - 1) && part;
      ProfRec* pr = prof_start(last ? T_UPDATE_LAST : T_UPDATE, st);
      if (last)
        hipLaunchKernelGGL(k_update<true>, dim3(tu), dim3(MM_THREADS), 0, st, A, N, nblk, k, w.Wo, w.M[0], ntri1, tu,
                           y, raw, n, part, -1);
      else
        hipLaunchKernelGGL(k_update<false>, dim3(tu), dim3(MM_THREADS), 0, st, A, N, nblk, k, w.Wo, w.M[0], ntri1, tu,
                           (const double*)nullptr, (const float*)nullptr, 0, (double*)nullptr, -1);
      prof_stop(pr, st);
      CHECK_LAUNCH("k_update");
    }
  }
  return 0;
}

extern "C" size_t gpsr_invert_workspace_bytes(int B, int n) {
  if (B <= 0 || n <= 0) return 0;
  return inv_ws_bytes(B, gpsr_padded_n(n));
}

extern "C" int gpsr_invert(double* A, int B, int n, double* logdet, int32_t* flags, void* workspace,
                           size_t workspace_bytes, void* stream) {
  if (B <= 0) return 0;
  if (!A || !logdet || !flags || !workspace) return arg_fail("null pointer");
  if (n % NB != 0) return arg_fail("gpsr_invert needs n == gpsr_padded_n(n)");
  if (workspace_bytes < inv_ws_bytes(B, n)) return arg_fail("workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(logdet, 0, (size_t)B * 8, st);
  if (e != hipSuccess) return hip_fail(e, "memset logdet");
  e = hipMemsetAsync(flags, 0, (size_t)B * 4, st);
  if (e != hipSuccess) return hip_fail(e, "memset flags");
  InvWs w = inv_ws_carve((char*)workspace, B, n);
  return launch_sweeps(A, B, n, w, logdet, flags, st);
}

// ---------------------------------------------------------------- fit
struct FitWs {
  double* A; InvWs inv; double *stats1, *stats2, *logdet, *part; float *m1, *m2;
};
// per-tile mat-vec partials of the last update launch: [B][ntri(nblk-1) + nblk-1][2][NB]
static size_t part_bytes(int B, int nblk) {
  const int per = (nblk - 1) * nblk / 2 + nblk - 1;
  return align256((size_t)B * (per > 0 ? per : 1) * 2 * NB * 8);
}
static size_t fit_ws_bytes(int B, int N) {
  const int nblk = N / NB, ntri = nblk * (nblk + 1) / 2;
  return align256((size_t)B * N * N * 8) + inv_ws_bytes(B, N) + align256((size_t)B * nblk * 4 * 8) +
         align256((size_t)B * ntri * 2 * 8) + align256((size_t)B * 8) + part_bytes(B, nblk) +
         2 * align256((size_t)B * 4 * 4);
}
static FitWs fit_ws_carve(char* base, int B, int N) {
  const int nblk = N / NB, ntri = nblk * (nblk + 1) / 2;
  FitWs w;
  w.A = (double*)base; base += align256((size_t)B * N * N * 8);
  w.inv = inv_ws_carve(base, B, N); base += inv_ws_bytes(B, N);
  w.stats1 = (double*)base; base += align256((size_t)B * nblk * 4 * 8);
  w.stats2 = (double*)base; base += align256((size_t)B * ntri * 2 * 8);
  w.logdet = (double*)base; base += align256((size_t)B * 8);
  w.part = (double*)base; base += part_bytes(B, nblk);
  w.m1 = (float*)base; base += align256((size_t)B * 4 * 4);
  w.m2 = (float*)base;
  return w;
}

extern "C" size_t gpsr_fit_workspace_bytes(int B, int n, int d) {
  (void)d;
  if (B <= 0 || n <= 0) return 0;
  return fit_ws_bytes(B, gpsr_padded_n(n));
}

static int launch_grad2(const FitWs& w, const double* X, const float* raw, const double* alpha, int B, int n,
                        int N, int d, int kid, hipStream_t st) {
  const int nblk = N / NB, ntri = nblk * (nblk + 1) / 2, total = ntri * B;
  const size_t lds = build_lds(d) + 2 * NB * sizeof(double);
#define G2(K)                                                                                                 \
  do {                                                                                                        \
    if (d == 8) hipLaunchKernelGGL((k_grad2_8<K>), dim3(total), dim3(MM_THREADS), 0, st, w.A, X, raw, alpha, n, N, ntri, total, w.stats2); \
    else hipLaunchKernelGGL((k_grad2<K>), dim3(total), dim3(256), lds, st, w.A, X, raw, alpha, n, N, d, ntri, total, w.stats2);            \
  } while (0)
  ProfRec* pr = prof_start(T_GRAD2, st);
  switch (kid) {
    case K_RBF: G2(K_RBF); break;
    case K_M12: G2(K_M12); break;
    case K_M32: G2(K_M32); break;
    case K_M52: G2(K_M52); break;
    default: G2(K_LIN); break;
  }
  prof_stop(pr, st);
#undef G2
  CHECK_LAUNCH("k_grad2");
  return 0;
}

// K build + inversion + alpha (+ gradient statistics when with_grad)
static int eval_once(const FitWs& w, const double* X, const double* y, const float* raw, double* alpha,
                     int32_t* flags, int B, int n, int N, int d, int kid, bool with_grad, hipStream_t st) {
  int rc = launch_build(X, B, n, N, d, kid, raw, w.A, w.logdet, st, 0);   // diagonal tiles: lower sub-blocks only
  if (rc) return rc;
  const int nblk = N / NB;
  const bool fused = nblk > 1;   // alpha's mat-vec rides on the last update launch; one block: nothing to ride on
  rc = fused ? launch_sweeps(w.A, B, N, w.inv, w.logdet, flags, st, y, raw, n, w.part)
             : launch_sweeps(w.A, B, N, w.inv, w.logdet, flags, st);
  if (rc) return rc;
  ProfRec* pr = prof_start(T_GRAD1, st);
  if (fused) {
    hipLaunchKernelGGL(k_grad1r, dim3(nblk, B), dim3(MM_THREADS), 0, st, w.A, N, n, nblk, nblk - 1, y, raw, w.part,
                       (nblk - 1) * nblk / 2, alpha, w.stats1);
  } else {
    const size_t lds1 = (size_t)(N + 5 * NB + 32) * sizeof(double);
    hipLaunchKernelGGL(k_grad1, dim3(nblk, B), dim3(MM_THREADS), lds1, st, w.A, N, n, nblk, y, raw, alpha, w.stats1);
  }
  prof_stop(pr, st);
  CHECK_LAUNCH("k_grad1");
  if (with_grad) return launch_grad2(w, X, raw, alpha, B, n, N, d, kid, st);
  return 0;
}

static int check_fit_args(const double* X, const double* y, int B, int n, int d, int kid, const void* ws,
                          size_t wsb) {
  if (!X || !y || !ws) return arg_fail("null pointer");
  if (n < 2 || d < 1 || d > 64 || kid < 0 || kid > 4) return arg_fail("fit shape / kernel_id");
  if (gpsr_padded_n(n) > 16 * NB) return arg_fail("n too large (max 1792)");
  if (wsb < fit_ws_bytes(B, gpsr_padded_n(n))) return arg_fail("workspace too small");
  return 0;
}

extern "C" int gpsr_fit(const double* X, const double* y, int B, int n, int d, int kernel_id, float* raw,
                        int iters, float lr, double* alpha, double* loss, int32_t* flags, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (B <= 0) return 0;
  int rc = check_fit_args(X, y, B, n, d, kernel_id, workspace, workspace_bytes);
  if (rc) return rc;
  if (!raw || !alpha || !flags || iters < 0) return arg_fail("fit outputs");
  hipStream_t st = (hipStream_t)stream;
  const int N = gpsr_padded_n(n), nblk = N / NB, ntri = nblk * (nblk + 1) / 2;
  FitWs w = fit_ws_carve((char*)workspace, B, N);
  hipError_t e = hipMemsetAsync(w.m1, 0, 2 * align256((size_t)B * 16), st);
  if (e != hipSuccess) return hip_fail(e, "memset adam state");
  e = hipMemsetAsync(flags, 0, (size_t)B * 4, st);
  if (e != hipSuccess) return hip_fail(e, "memset flags");
  const double b1 = 0.9, b2 = 0.999;
  for (int it = 1; it <= iters; ++it) {
    rc = eval_once(w, X, y, raw, alpha, flags, B, n, N, d, kernel_id, true, st);
    if (rc) return rc;
    const double bc1 = 1.0 - pow(b1, it), bc2 = 1.0 - pow(b2, it);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    hipLaunchKernelGGL(k_adam, dim3((B + 63) / 64), dim3(64), 0, st, raw, w.m1, w.m2, w.stats1, w.stats2, w.logdet,
                       B, n, nblk, ntri, loss, (float*)nullptr, 1, step_size, bc2_sqrt, (float)(1.0 - b1),
                       (float)b2, (float)(1.0 - b2), 1e-8f);
    CHECK_LAUNCH("k_adam");
  }
  // final factorisation at the post-step hyper-parameters -> alpha for the predictive mean
  return eval_once(w, X, y, raw, alpha, flags, B, n, N, d, kernel_id, false, st);
}

extern "C" int gpsr_loss_grad(const double* X, const double* y, int B, int n, int d, int kernel_id,
                              const float* raw, double* loss, float* grad, int32_t* flags, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (B <= 0) return 0;
  int rc = check_fit_args(X, y, B, n, d, kernel_id, workspace, workspace_bytes);
  if (rc) return rc;
  if (!raw || !loss || !grad || !flags) return arg_fail("loss_grad outputs");
  hipStream_t st = (hipStream_t)stream;
  const int N = gpsr_padded_n(n), nblk = N / NB, ntri = nblk * (nblk + 1) / 2;
  FitWs w = fit_ws_carve((char*)workspace, B, N);
  hipError_t e = hipMemsetAsync(flags, 0, (size_t)B * 4, st);
  if (e != hipSuccess) return hip_fail(e, "memset flags");
  double* alpha = w.inv.Wo;  // [B][N] fits in [B][NB][N]; Wo is dead after the sweeps
  rc = eval_once(w, X, y, raw, alpha, flags, B, n, N, d, kernel_id, true, st);
  if (rc) return rc;
  hipLaunchKernelGGL(k_adam, dim3((B + 63) / 64), dim3(64), 0, st, (float*)raw, w.m1, w.m2, w.stats1, w.stats2,
                     w.logdet, B, n, nblk, ntri, loss, grad, 0, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f);
  CHECK_LAUNCH("k_adam(eval)");
  return 0;
}

// ---------------------------------------------------------------- predict
extern "C" int gpsr_predict(const double* X, const double* Xs, const double* alpha, const float* raw, int B,
                            int n, int m, int kernel_id, double* out, void* stream) {
  if (B <= 0 || m <= 0) return 0;
  if (!X || !Xs || !alpha || !raw || !out) return arg_fail("null pointer");
  if (kernel_id < 0 || kernel_id > 4) return arg_fail("kernel_id");
  if (n < 1 || (size_t)n * 9 * 8 > 150 * 1024) return arg_fail("n too large for predict");
  const int N = gpsr_padded_n(n);
  const size_t lds = (size_t)n * 9 * sizeof(double);
  dim3 grid((m + 256 * PRED_TP - 1) / (256 * PRED_TP), B);
  hipStream_t st = (hipStream_t)stream;
#define PR(K)                                                                                               \
  do {                                                                                                      \
    if (lds > 48 * 1024) {                                                                                  \
      hipError_t e = hipFuncSetAttribute((const void*)k_predict<K>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                         (int)lds);                                                         \
      if (e != hipSuccess) return hip_fail(e, "predict lds attr");                                          \
    }                                                                                                       \
    hipLaunchKernelGGL((k_predict<K>), grid, dim3(256), lds, st, X, Xs, alpha, raw, n, N, m, out);          \
  } while (0)
  ProfRec* pr = prof_start(T_PREDICT, st);
  switch (kernel_id) {
    case K_RBF: PR(K_RBF); break;
    case K_M12: PR(K_M12); break;
    case K_M32: PR(K_M32); break;
    case K_M52: PR(K_M52); break;
    default: PR(K_LIN); break;
  }
  prof_stop(pr, st);
#undef PR
  CHECK_LAUNCH("k_predict");
  return 0;
}

// ---------------------------------------------------------------- combine
extern "C" int gpsr_combine(const double* pred, const double* test_in, int H, int W, int s, int p,
                            const int32_t* rows, int nr, const int32_t* cols, int nc, int mode, int clip,
                            double* out, void* stream) {
  if (!pred || !rows || !cols || !out) return arg_fail("null pointer");
  if (nr < 1 || nc < 1 || p < 3 || s < 1 || (mode != 0 && mode != 1)) return arg_fail("combine geometry");
  const size_t total = (size_t)H * s * W * s;
  hipLaunchKernelGGL(k_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred,
                     test_in, H, W, s, p, rows, nr, cols, nc, mode, clip, out);
  CHECK_LAUNCH("k_combine");
  return 0;
}

// ---------------------------------------------------------------- fused predict_pixels
struct PixWs { int32_t* locs; float* raw; double *pred, *X, *y, *Xs, *alpha; char* fit; };

static size_t pix_ws_carve(char* base, int B, int Bc, int n, int m, PixWs* w) {
  const int N = gpsr_padded_n(n);
  size_t off = 0;
#define CARVE(field, type, bytes) do { if (w) w->field = (type)(base + off); off += align256(bytes); } while (0)
  CARVE(locs, int32_t*, (size_t)B * 2 * 4);
  CARVE(raw, float*, (size_t)B * 4 * 4);
  CARVE(pred, double*, (size_t)B * m * 8);
  CARVE(X, double*, (size_t)Bc * n * 8 * 8);
  CARVE(y, double*, (size_t)Bc * n * 8);
  CARVE(Xs, double*, (size_t)Bc * m * 8 * 8);
  CARVE(alpha, double*, (size_t)Bc * N * 8);
#undef CARVE
  if (w) w->fit = base + off;
  return off + fit_ws_bytes(Bc, N);
}

extern "C" size_t gpsr_predict_pixels_workspace_bytes(int s, int p, int nt, int B, int chunk) {
  if (s < 1 || p < 3 || nt < 1 || B < 1) return 0;
  const int Bc = (chunk <= 0 || chunk > B) ? B : chunk;
  return pix_ws_carve(nullptr, B, Bc, nt * nt, (p * s - 2) * (p * s - 2), nullptr);
}

extern "C" int gpsr_predict_pixels(const double* train_in, const double* train_tgt, const double* test_in,
                                   int H, int W, int s, int p, const int32_t* rows, int nr,
                                   const int32_t* cols, int nc, const int32_t* tidx, int nt,
                                   int kernel_id, float init_raw_os, float init_raw_ls, int iters, float lr,
                                   int mode, int chunk, double* out, float* raw_out, int32_t* flags,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!train_in || !train_tgt || !test_in || !rows || !cols || !tidx || !out || !flags || !workspace)
    return arg_fail("null pointer");
  if (nr < 1 || nc < 1 || p < 3 || s < 1 || p > H || p > W || nt < 1 || nt > p - 2) return arg_fail("patch geometry");
  if (kernel_id < 0 || kernel_id > 4 || iters < 0 || (mode != 0 && mode != 1)) return arg_fail("kernel_id / iters / mode");
  const int B = nr * nc, Bc = (chunk <= 0 || chunk > B) ? B : chunk;
  const int n = nt * nt, m = (p * s - 2) * (p * s - 2), N = gpsr_padded_n(n);
  if (n < 2 || N > 16 * NB) return arg_fail("n out of range (2..1792)");
  if (workspace_bytes < gpsr_predict_pixels_workspace_bytes(s, p, nt, B, chunk)) return arg_fail("workspace too small");
  PixWs w;
  pix_ws_carve((char*)workspace, B, Bc, n, m, &w);
  float* raw = raw_out ? raw_out : w.raw;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_make_locs, dim3((B + 255) / 256), dim3(256), 0, st, rows, cols, nr, nc, w.locs);
  CHECK_LAUNCH("k_make_locs");
  const size_t fit_bytes = fit_ws_bytes(Bc, N);
  for (int b0 = 0; b0 < B; b0 += Bc) {
    const int bn = (B - b0 < Bc) ? B - b0 : Bc;
    int rc = gpsr_gather(train_in, train_tgt, test_in, H, W, s, p, w.locs + 2 * b0, bn, tidx, nt, w.X, w.y, w.Xs, stream);
    if (rc) return rc;
    rc = gpsr_init_hypers(w.y, bn, n, kernel_id, init_raw_os, init_raw_ls, nullptr, raw + 4 * b0, stream);
    if (rc) return rc;
    rc = gpsr_fit(w.X, w.y, bn, n, 8, kernel_id, raw + 4 * b0, iters, lr, w.alpha, nullptr, flags + b0, w.fit,
                  fit_bytes, stream);
    if (rc) return rc;
    rc = gpsr_predict(w.X, w.Xs, w.alpha, raw + 4 * b0, bn, n, m, kernel_id, w.pred + (size_t)b0 * m, stream);
    if (rc) return rc;
  }
  return gpsr_combine(w.pred, test_in, H, W, s, p, rows, nr, cols, nc, mode, 1, out, stream);
}

// ---------------------------------------------------------------- AGPR
extern "C" int gpsr_agpr_weights(const double* X, const double* alpha, const float* raw, int n, int d,
                                 double* w, void* stream) {
  if (!X || !alpha || !raw || !w || d < 1 || d > 64 || n < 1) return arg_fail("agpr_weights");
  hipLaunchKernelGGL(k_agpr_weights, dim3(1), dim3(64), 0, (hipStream_t)stream, X, alpha, raw, n, d, w);
  CHECK_LAUNCH("k_agpr_weights");
  return 0;
}

extern "C" int gpsr_agpr_filter(const double* img, int Hs, int Ws, int L, const double* w, const float* raw,
                                double* out, void* stream) {
  if (!img || !w || !raw || !out || L < 1 || (L & 1) == 0 || L > 8 || Hs < L || Ws < L) return arg_fail("agpr_filter");
  const size_t total = (size_t)Hs * Ws;
  hipLaunchKernelGGL(k_agpr_filter, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img,
                     Hs, Ws, L, w, raw, out);
  CHECK_LAUNCH("k_agpr_filter");
  return 0;
}

extern "C" int gpsr_selftest_mfma(const double* A, const double* Bm, double* D, void* stream) {
  if (!A || !Bm || !D) return arg_fail("null pointer");
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, (hipStream_t)stream, A, Bm, D);
  CHECK_LAUNCH("k_selftest_mfma");
  return 0;
}

// ---------------------------------------------------------------- image stages
#define GRID1D(n) dim3((unsigned)(((size_t)(n) + 255) / 256)), dim3(256), 0, (hipStream_t)stream

extern "C" int gpsr_u8_to_unit(const uint8_t* src, double* dst, size_t n, void* stream) {
  if (!src || !dst) return arg_fail("null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_u8_to_unit, GRID1D(n), src, dst, n);
  CHECK_LAUNCH("k_u8_to_unit");
  return 0;
}

extern "C" int gpsr_unit_to_u8(const double* src, uint8_t* dst, size_t n, double scale, void* stream) {
  if (!src || !dst) return arg_fail("null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_unit_to_u8, GRID1D(n), src, dst, n, scale);
  CHECK_LAUNCH("k_unit_to_u8");
  return 0;
}

extern "C" int gpsr_resample(const double* src, int Hs, int Ws, double* dst, int Hd, int Wd, const int32_t* xi,
                             const double* xw, const int32_t* yi, const double* yw, int taps, void* stream) {
  if (!src || !dst || !xi || !xw || !yi || !yw) return arg_fail("null pointer");
  if (Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1 || taps < 1 || taps > 8) return arg_fail("resample shape");
  hipLaunchKernelGGL(k_resample, GRID1D((size_t)Hd * Wd), src, Hs, Ws, dst, Hd, Wd, xi, xw, yi, yw, taps);
  CHECK_LAUNCH("k_resample");
  return 0;
}

extern "C" int gpsr_decimate2(const double* src, int Hs, int Ws, double* dst, void* stream) {
  if (!src || !dst || Hs < 2 || Ws < 2) return arg_fail("decimate2");
  const int Hd = Hs / 2, Wd = Ws / 2;
  hipLaunchKernelGGL(k_decimate2, GRID1D((size_t)Hd * Wd), src, Ws, dst, Hd, Wd);
  CHECK_LAUNCH("k_decimate2");
  return 0;
}

extern "C" int gpsr_sepfilter_sym(const double* src, int H, int W, double* dst, const double* kern, int ksize,
                                  void* stream) {
  if (!src || !dst || !kern || H < 1 || W < 1) return arg_fail("sepfilter");
  if ((ksize & 1) == 0 || ksize < 1 || ksize > 7) return arg_fail("sepfilter ksize (odd, <= 7)");
  hipLaunchKernelGGL(k_sepfilter_sym, GRID1D((size_t)H * W), src, H, W, dst, kern, ksize / 2);
  CHECK_LAUNCH("k_sepfilter_sym");
  return 0;
}

extern "C" int gpsr_filter3x3(const double* src, int H, int W, double* dst, const double* kern, void* stream) {
  if (!src || !dst || !kern || H < 1 || W < 1) return arg_fail("filter3x3");
  hipLaunchKernelGGL(k_filter3x3, GRID1D((size_t)H * W), src, H, W, dst, kern);
  CHECK_LAUNCH("k_filter3x3");
  return 0;
}

extern "C" int gpsr_axpy(const double* a, const double* b, double alpha, double* dst, size_t n, void* stream) {
  if (!a || !b || !dst) return arg_fail("null pointer");
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_axpy, GRID1D(n), a, b, alpha, dst, n);
  CHECK_LAUNCH("k_axpy");
  return 0;
}
