// gpsr_api.hip -- C ABI (include/gpsr.h) over the gfx950 kernels.  No torch types; the only allocation is
// the workspace of an optional gpsr_ctx (gpsr_create): every other buffer belongs to the caller.
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
enum { T_UPDATE = 0, T_PANEL, T_PIVOT, T_BUILD, T_GRAD1, T_GRAD2, T_PREDICT, T_UPDATE_LAST, T_UPDATE2, T_UPDATE_SEL, T_FUSED, T_NTAGS };
struct ProfRec { hipEvent_t a, b; int tag; };
static std::vector<ProfRec> g_prof;
static size_t g_prof_used = 0;
static bool g_prof_on = false;

static unsigned g_prof_mask = ~0u;   // tags that are bracketed (gpsr_profile_select)
static inline ProfRec* prof_start(int tag, hipStream_t st) {
  if (!g_prof_on || !((g_prof_mask >> tag) & 1u) || g_prof_used >= g_prof.size()) return nullptr;
  ProfRec* r = &g_prof[g_prof_used++];
  r->tag = tag;
  (void)hipEventRecord(r->a, st);
  return r;
}
static inline void prof_stop(ProfRec* r, hipStream_t st) { if (r) (void)hipEventRecord(r->b, st); }

// Which tags gpsr_profile_begin brackets (bit t = tag t; default all).  Two hipEventRecord per launch are not free: on
// the chained warm-start modes (188 000 small launches per Set14 x3 step) bracketing every launch costs a quarter of
// the step, so bench.py brackets only the kernel its roofline object describes inside the timed region.
extern "C" int gpsr_profile_select(unsigned tag_mask) {
  g_prof_mask = tag_mask;
  return 0;
}

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
  if ((((uintptr_t)X) | ((uintptr_t)Xs)) & 15) return arg_fail("X and Xs must be 16-byte aligned");
  const int n = nt * nt, m = (p * s - 2) * (p * s - 2);
  const int total = 5 * n + 4 * m;   // 16-byte work units, see k_gather
  int gx = (total + 255) / 256; if (gx > 64) gx = 64;
  hipLaunchKernelGGL(k_gather, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, train_in, train_tgt, test_in,
                     H, W, s, p, locs, tidx, nt, X, y, Xs);
  CHECK_LAUNCH("k_gather");
  return 0;
}

// ---------------------------------------------------------------- init
static float inv_softplus_host(float x) { return x + logf(-expm1f(-x)); }

extern "C" int gpsr_init_hypers(const double* y, int B, int n, int kernel_id, float init_raw_os,
                                float init_raw_ls, const float* carry, int carry_stride, float* raw, void* stream) {
  if (B <= 0) return 0;
  if (!y || !raw || n < 2) return arg_fail("init_hypers");
  if (carry && carry_stride != 0 && carry_stride < 4) return arg_fail("carry_stride (0: one carry for all GPs, >= 4: floats between rows)");
  if (kernel_id < 0 || kernel_id > 4) return arg_fail("kernel_id");
  const float raw_noise0 = inv_softplus_host(0.001f - 1e-4f);
  const float raw_ls_rbf = inv_softplus_host(0.223f);
  hipLaunchKernelGGL(k_init_hypers, dim3(B), dim3(256), 0, (hipStream_t)stream, y, n, kernel_id, raw_noise0,
                     raw_ls_rbf, init_raw_os, init_raw_ls, carry, carry_stride, raw);
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

extern "C" int gpsr_covariance_grad(const double* X, int B, int n, int d, int kernel_id, const float* raw,
                                    double* K, double* dK_dl, void* stream) {
  if (B <= 0) return 0;
  if (!X || !raw || !dK_dl) return arg_fail("null pointer");
  if (n < 1 || d < 1 || d > 64 || kernel_id < 0 || kernel_id > 4) return arg_fail("covariance shape");
  const int N = gpsr_padded_n(n), nblk = N / NB, ntri = nblk * (nblk + 1) / 2, total = ntri * B;
  const size_t lds = build_lds(d);
  hipStream_t st = (hipStream_t)stream;
#define BDL(KK) hipLaunchKernelGGL((k_build_dl<KK>), dim3(total), dim3(256), lds, st, X, raw, K, dK_dl, n, N, d, ntri, total)
  switch (kernel_id) {
    case K_RBF: BDL(K_RBF); break;
    case K_M12: BDL(K_M12); break;
    case K_M32: BDL(K_M32); break;
    case K_M52: BDL(K_M52); break;
    default: BDL(K_LIN); break;
  }
#undef BDL
  CHECK_LAUNCH("k_build_dl");
  return 0;
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
                        int with_p, hipStream_t st) {
  ProfRec* pr = prof_start(T_PIVOT, st);
  hipLaunchKernelGGL(k_pivot, dim3(B), dim3(MM_THREADS), PIVOT_LDS_BYTES, st, A, N, k, U, M, logdet, flags, with_p);
  prof_stop(pr, st);
  CHECK_LAUNCH("k_pivot");
  return 0;
}
// Small batches run the tile products in strip mode (gpsr_kernels.hpp): one wave per workgroup, seven workgroups per tile.
// Used while the launch in strip form still fits the GPU at once (48 KB of LDS each: three such workgroups per CU, 768
// slots): measured on n = 784, B = 1, 2 every launch is 10-35 % shorter, at B = 4 the panel and the restricted update
// still are, the full updates (756 single-wave workgroups then) no longer.
static const bool g_strip = getenv("GPSR_NO_STRIP") == nullptr;   // debugging switch
static inline bool strip_mode(int tiles) { return g_strip && tiles * NBT <= 700; }

static int launch_panel(const double* A, int B, int N, int k, const double* U, double* Wo, hipStream_t st) {
  const int nblk = N / NB, tp = (nblk - 1) * B;
  ProfRec* pr = prof_start(T_PANEL, st);
  if (strip_mode(tp)) hipLaunchKernelGGL(k_panel<true>, dim3(tp * NBT), dim3(64), 0, st, A, N, nblk, k, U, Wo, tp * NBT);
  else hipLaunchKernelGGL(k_panel<false>, dim3(tp), dim3(MM_THREADS), 0, st, A, N, nblk, k, U, Wo, tp);
  prof_stop(pr, st);
  CHECK_LAUNCH("k_panel");
  return 0;
}

// ---------------------------------------------------------------- fit workspace
struct FitWs {
  double* A; InvWs inv; double *stats1, *stats2, *logdet, *part; float *m1, *m2;
};
// per-tile mat-vec partials of the last update launch: [B][ntri(nblk-1) + nblk-1][2][NB]
static size_t part_bytes(int B, int nblk) {
  const int per = (nblk - 1) * nblk / 2 + nblk - 1;
  return align256((size_t)B * (per > 0 ? per : 1) * 2 * NB * 8);
}
static size_t fit_ws_bytes1(int B, int N) {
  const int nblk = N / NB, ntri = nblk * (nblk + 1) / 2;
  return align256((size_t)B * N * N * 8) + inv_ws_bytes(B, N) + align256((size_t)B * nblk * 4 * 8) +
         align256((size_t)B * ntri * 2 * 8) + align256((size_t)B * 8) + part_bytes(B, nblk) +
         2 * align256((size_t)B * 4 * 4);
}
// ---- fused per-GP route (k_fit_fused; opt-in): ONE launch per fit, one workgroup per GP, a persistent grid of two
// workgroups per CU; the workspace is per SLOT (resident workgroup), not per GP (0.9 GiB instead of 19 GiB for 10 800 GPs
// of n = 324).  Built to take the n = 324 path (BASELINE config 4) off HBM; measured on MI355X it does neither that
// (7.9 MB of HBM traffic per GP-evaluation against 4.5 MB, profiles/r03_pmc_fit1*_10800x324.json) nor run faster (2.0 us
// per GP-evaluation against 1.585): gpsr_kernels.hpp and DESIGN.md section 9 say why.  Kept as a tested, bit-identical,
// opt-in route: gpsr_set_fused(1) / GPSR_FUSED=1; GPSR_FUSED_STAGGER is a tuning switch.
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static int g_fused_mode = env_int("GPSR_FUSED", 0) ? 1 : 0;
static const int g_fused_stagger = env_int("GPSR_FUSED_STAGGER", 24);

extern "C" int gpsr_set_fused(int on) {
  const int old = g_fused_mode;
  g_fused_mode = on ? 1 : 0;
  return old;
}
static bool fused_applies(int B, int N, int d) {
  (void)B;
  return g_fused_mode != 0 && d == 8 && N / NB >= 2;
}
static int fused_slots() {   // two workgroups per CU (76 KB of LDS each)
  static int slots[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
  if (!slots[dev]) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    slots[dev] = 2 * cus;
  }
  return slots[dev];
}
static size_t fused_ws_bytes(int B, int N) {
  const int S = B < fused_slots() ? B : fused_slots();
  return fit_ws_bytes1(S, N) + 2 * align256((size_t)B * 16);   // + the Adam moments of every GP
}
// the fit may run its batch as two halves with a workspace each (run_dual)
static size_t fit_ws_bytes(int B, int N, int d) {
  if (fused_applies(B, N, d)) return fused_ws_bytes(B, N);
  const size_t one = fit_ws_bytes1(B, N), two = fit_ws_bytes1((B + 1) / 2, N) + fit_ws_bytes1(B / 2 > 0 ? B / 2 : 1, N);
  return one > two ? one : two;
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
  if (B <= 0 || n <= 0) return 0;
  return fit_ws_bytes(B, gpsr_padded_n(n), d);
}

// ---------------------------------------------------------------- launch sequence
// One evaluation (covariance, blocked sweep, alpha, gradient statistics) and one fit are sequences of
// launches over the same batch.  They are written down as a list of operations first and issued
// afterwards, so that the same list can drive one batch on one stream or two half-batches on two streams.
enum { OP_BUILD, OP_PIVOT, OP_PANEL, OP_UPD_SEL, OP_UPD2, OP_UPD, OP_GRAD1, OP_GRAD2, OP_ADAM };
struct Op { int kind, k, set, arg; bool heavy; };   // set: which (U, M, Wo) triple; arg: OP_UPD last flag / OP_ADAM iteration /
                                                    // OP_PIVOT: the pivot forms tile (k,k) = -M^T M itself (mark_pivots)

struct Batch {   // one batch (or half-batch) of GPs and its buffers
  int B;
  int Bdec;   // 0, or the size of the whole batch this one is a slice of: launch-shape decisions that two launches of
              // one pivot step must agree on (who forms tile (k,k)) are taken on it
  const double *X, *y;
  float* raw;
  double *alpha, *loss;
  float* grad;
  int32_t* flags;
  FitWs w;
  hipStream_t st;
};
struct FitShape { int n, N, d, kid; float lr; bool with_part; };

// Tile (k,k) = -M^T M of a pivot step is formed either by k_pivot itself (its phase 3) or, as one more tile product, by
// the update launch that follows.  The second takes 12-19 % off the latency-bound pivot launch and is free while the
// update launch fits the GPU in one round of workgroups; in a large batch the extra workgroup slot costs more than the
// pivot saves (633 x n=784: update2 +5 %, 10 800 x n=324: +20 %).  The alpha-epilogue instantiation stays free of the
// extra role, single-block matrices have no update.  OP_PIVOT.arg = kind of the update launch that follows (0: none).
enum { PT_NONE = 0, PT_UPD = 1, PT_UPD2 = 2 };
static void mark_pivots(Op* ops, int nops) {
  for (int i = 0; i < nops; ++i) {
    if (ops[i].kind != OP_PIVOT) continue;
    int j = i + 1;
    while (j < nops && ops[j].kind != OP_UPD && ops[j].kind != OP_UPD2) ++j;
    ops[i].arg = (j == nops || (ops[j].kind == OP_UPD && ops[j].arg)) ? PT_NONE : (ops[j].kind == OP_UPD ? PT_UPD : PT_UPD2);
  }
}
static const bool g_ptile = getenv("GPSR_NO_PTILE") == nullptr;   // debugging switch
static inline bool update_forms_ptile(int pt_kind, int B, int nblk) {
  if (!g_ptile || pt_kind == PT_NONE) return false;
  const int nb2 = nblk - 2;
  const int per = pt_kind == PT_UPD ? (nblk - 1) * nblk / 2 + nblk : nb2 * (nb2 + 1) / 2 + nb2 + 1 + nblk;
  return per * B <= 512;
}
// pivots are taken two at a time while at least one single step remains for the end (the alpha epilogue
// lives in the single-step kernel): nblk odd -> (nblk-1)/2 pairs + 1 single, even -> (nblk-2)/2 pairs + 2
static void append_sweep_ops(std::vector<Op>& ops, int nblk, bool with_part) {
  const int npairs = (g_pair_steps && nblk >= 3) ? (nblk - 1 - (nblk % 2 == 0 ? 1 : 0)) / 2 : 0;
  int k = 0;
  for (int m = 0; m < npairs; ++m, k += 2) {
    ops.push_back({OP_PIVOT, k, 0, 0, false});
    ops.push_back({OP_PANEL, k, 0, 0, false});
    ops.push_back({OP_UPD_SEL, k, 0, 0, false});
    ops.push_back({OP_PIVOT, k + 1, 1, 0, false});
    ops.push_back({OP_PANEL, k + 1, 1, 0, false});
    ops.push_back({OP_UPD2, k, 0, 0, true});
  }
  for (; k < nblk; ++k) {
    ops.push_back({OP_PIVOT, k, 0, 0, false});
    if (nblk > 1) {
      ops.push_back({OP_PANEL, k, 0, 0, false});
      ops.push_back({OP_UPD, k, 0, (k == nblk - 1 && with_part) ? 1 : 0, true});
    }
  }
  mark_pivots(ops.data(), (int)ops.size());
}
static void append_eval_ops(std::vector<Op>& ops, int nblk, bool with_grad) {
  ops.push_back({OP_BUILD, 0, 0, 0, true});
  append_sweep_ops(ops, nblk, nblk > 1);   // alpha's mat-vec rides on the last update launch; one block: nothing to ride on
  ops.push_back({OP_GRAD1, 0, 0, 0, false});
  if (with_grad) ops.push_back({OP_GRAD2, 0, 0, 0, true});
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

static int launch_op(const Op& op, const Batch& h, const FitShape& c) {
  const int B = h.B, N = c.N, nblk = N / NB, ntri1 = (nblk - 1) * nblk / 2, ntri = nblk * (nblk + 1) / 2;
  const int Bd = h.Bdec ? h.Bdec : h.B;
  const FitWs& w = h.w;
  hipStream_t st = h.st;
  double* Wo = op.set ? w.inv.Wo2 : w.inv.Wo;
  ProfRec* pr;
  switch (op.kind) {
    case OP_BUILD:
      return launch_build(h.X, B, c.n, N, c.d, c.kid, h.raw, w.A, w.logdet, st, 0);   // diagonal tiles: lower sub-blocks only
    case OP_PIVOT:
      return launch_pivot(w.A, B, N, op.k, w.inv.U[op.set], w.inv.M[op.set], w.logdet, h.flags, update_forms_ptile(op.arg, Bd, nblk) ? 0 : (op.k == nblk - 1 ? 2 : 1), st);
    case OP_PANEL:
      return launch_panel(w.A, B, N, op.k, w.inv.U[op.set], Wo, st);
    case OP_UPD_SEL: {
      const int tu = nblk * B;   // the tiles of block row / column k+1 only
      pr = prof_start(T_UPDATE_SEL, st);
      if (strip_mode(tu))
        hipLaunchKernelGGL((k_update<false, true>), dim3(tu * NBT), dim3(64), 0, st, w.A, N, nblk, op.k, w.inv.Wo, w.inv.M[0],
                           ntri1, tu * NBT, (const double*)nullptr, (const float*)nullptr, 0, (double*)nullptr, op.k + 1, 0);
      else
        hipLaunchKernelGGL((k_update<false, false>), dim3(tu), dim3(MM_THREADS), 0, st, w.A, N, nblk, op.k, w.inv.Wo, w.inv.M[0],
                           ntri1, tu, (const double*)nullptr, (const float*)nullptr, 0, (double*)nullptr, op.k + 1, 0);
      prof_stop(pr, st);
      CHECK_LAUNCH("k_update(sel)");
      return 0;
    }
    case OP_UPD2: {
      const int pt = update_forms_ptile(PT_UPD2, Bd, nblk) ? 1 : 0;
      const int nb2 = nblk - 2, per = nb2 * (nb2 + 1) / 2 + nb2 + 1 + nblk - 1 + pt, tu = per * B;
      pr = prof_start(T_UPDATE2, st);
      if (strip_mode(tu))
        hipLaunchKernelGGL(k_update2<true>, dim3(tu * NBT), dim3(64), 0, st, w.A, N, nblk, op.k, w.inv.Wo, w.inv.Wo2,
                           w.inv.M[0], w.inv.M[1], tu * NBT, pt);
      else
        hipLaunchKernelGGL(k_update2<false>, dim3(tu), dim3(MM_THREADS), 0, st, w.A, N, nblk, op.k, w.inv.Wo, w.inv.Wo2,
                           w.inv.M[0], w.inv.M[1], tu, pt);
      prof_stop(pr, st);
      CHECK_LAUNCH("k_update2");
      return 0;
    }
    case OP_UPD: {
      const int pt = (!op.arg && update_forms_ptile(PT_UPD, Bd, nblk)) ? 1 : 0;   // + tile (k,k), see mark_pivots
      const int tu = (ntri1 + nblk - 1 + pt) * B;
      pr = prof_start(op.arg ? T_UPDATE_LAST : T_UPDATE, st);
      if (op.arg)
        hipLaunchKernelGGL((k_update<true, false>), dim3(tu), dim3(MM_THREADS), 0, st, w.A, N, nblk, op.k, w.inv.Wo, w.inv.M[0],
                           ntri1, tu, h.y, (const float*)h.raw, c.n, w.part, -1, 0);
      else if (strip_mode(tu))
        hipLaunchKernelGGL((k_update<false, true>), dim3(tu * NBT), dim3(64), 0, st, w.A, N, nblk, op.k, w.inv.Wo, w.inv.M[0],
                           ntri1, tu * NBT, (const double*)nullptr, (const float*)nullptr, 0, (double*)nullptr, -1, pt);
      else
        hipLaunchKernelGGL((k_update<false, false>), dim3(tu), dim3(MM_THREADS), 0, st, w.A, N, nblk, op.k, w.inv.Wo, w.inv.M[0],
                           ntri1, tu, (const double*)nullptr, (const float*)nullptr, 0, (double*)nullptr, -1, pt);
      prof_stop(pr, st);
      CHECK_LAUNCH("k_update");
      return 0;
    }
    case OP_GRAD1: {
      pr = prof_start(T_GRAD1, st);
      if (nblk > 1) {
        hipLaunchKernelGGL(k_grad1r, dim3(nblk, B), dim3(MM_THREADS), 0, st, w.A, N, c.n, nblk, nblk - 1, h.y,
                           (const float*)h.raw, w.part, ntri1, h.alpha, w.stats1);
      } else {
        const size_t lds1 = (size_t)(N + 5 * NB + 32) * sizeof(double);
        hipLaunchKernelGGL(k_grad1, dim3(nblk, B), dim3(MM_THREADS), lds1, st, w.A, N, c.n, nblk, h.y,
                           (const float*)h.raw, h.alpha, w.stats1);
      }
      prof_stop(pr, st);
      CHECK_LAUNCH("k_grad1");
      return 0;
    }
    case OP_GRAD2:
      return launch_grad2(w, h.X, h.raw, h.alpha, B, c.n, N, c.d, c.kid, st);
    case OP_ADAM: {
      if (op.arg > 0) {   // Adam step number op.arg
        const double b1 = 0.9, b2 = 0.999;
        const double bc1 = 1.0 - pow(b1, op.arg), bc2 = 1.0 - pow(b2, op.arg);
        const float step_size = (float)((double)c.lr / bc1);
        const float bc2_sqrt = (float)sqrt(bc2);
        hipLaunchKernelGGL(k_adam, dim3((B + 63) / 64), dim3(64), 0, st, h.raw, w.m1, w.m2, w.stats1, w.stats2, w.logdet,
                           B, c.n, nblk, ntri, h.loss, (float*)nullptr, 1, step_size, bc2_sqrt, (float)(1.0 - b1),
                           (float)b2, (float)(1.0 - b2), 1e-8f);
      } else {            // evaluation only: loss and gradient out, no step
        hipLaunchKernelGGL(k_adam, dim3((B + 63) / 64), dim3(64), 0, st, h.raw, w.m1, w.m2, w.stats1, w.stats2, w.logdet,
                           B, c.n, nblk, ntri, h.loss, h.grad, 0, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f);
      }
      CHECK_LAUNCH("k_adam");
      return 0;
    }
  }
  return arg_fail("unknown op");
}

static int ensure_lds_attrs() {
  static bool done[64] = {};   // the attribute belongs to the kernel's code object on ONE device: once per device
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  if (dev < 0 || dev >= 64 || !done[dev]) {
    e = hipFuncSetAttribute((const void*)k_pivot, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIVOT_LDS_BYTES);
    if (e != hipSuccess) return hip_fail(e, "pivot lds attr");
    if (dev >= 0 && dev < 64) done[dev] = true;
  }
  return 0;
}

static int run_single(const std::vector<Op>& ops, const Batch& h, const FitShape& c) {
  int rc = ensure_lds_attrs();
  for (size_t i = 0; i < ops.size() && !rc; ++i) rc = launch_op(ops[i], h, c);
  return rc;
}

// Two half-batches on two streams (opt-in: gpsr_set_dual_stream).  The launches of one evaluation alternate
// between kernels that fill the chip (covariance, the big updates, the gradient pass) and kernels that cannot
// (k_pivot is a latency-bound chain on 2 workgroups per CU; k_panel and the 7-tile update are short): back to
// back on one stream the latter leave part of the machine idle.  Two independent halves on two streams fill
// some of those holes: measured 1030 -> 975 ms per Set14 x3 step.  Chaining the big kernels A, B, A, B with
// events so that only small ones run underneath was tried and is slower (1009 ms): the small kernels starve
// under a kernel that owns every workgroup slot.  Half A uses the caller's stream, half B a side stream forked
// from and joined to it with events: the call stays asynchronous and ordered on the caller's stream as a whole.
// Not the default: with overlapping launches a per-launch duration no longer measures one kernel (bench.py's
// roofline), and the results are bit-identical either way.
struct SideStream { hipStream_t s; hipEvent_t fork, join; bool ok; };
static thread_local SideStream g_side[64] = {};
static int g_dual_stream = 0;

extern "C" int gpsr_set_dual_stream(int on) {
  const int old = g_dual_stream;
  g_dual_stream = on ? 1 : 0;
  return old;
}

static SideStream* side_stream() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  SideStream* ss = &g_side[dev];
  if (!ss->ok) {   // all three or nothing: what was created is destroyed again if a later creation fails
    if (hipStreamCreateWithFlags(&ss->s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    if (hipEventCreateWithFlags(&ss->fork, hipEventDisableTiming) != hipSuccess) {
      (void)hipStreamDestroy(ss->s);
      return nullptr;
    }
    if (hipEventCreateWithFlags(&ss->join, hipEventDisableTiming) != hipSuccess) {
      (void)hipEventDestroy(ss->fork);
      (void)hipStreamDestroy(ss->s);
      return nullptr;
    }
    ss->ok = true;   // lives until the thread's process ends (one side stream per thread and device)
  }
  return ss;
}

static int run_dual(const std::vector<Op>& ops, const Batch& a, Batch b, const FitShape& c, SideStream* ss) {
  int rc = ensure_lds_attrs();
  if (rc) return rc;
  hipError_t e;
  b.st = ss->s;
  if ((e = hipEventRecord(ss->fork, a.st)) != hipSuccess) return hip_fail(e, "fork record");
  if ((e = hipStreamWaitEvent(b.st, ss->fork, 0)) != hipSuccess) return hip_fail(e, "fork wait");
  // the host feeds the two streams alternately, one big launch (and the small ones before it) at a time
  const size_t n = ops.size();
  size_t ia = 0, ib = 0;
  while (!rc && (ia < n || ib < n)) {
    while (!rc && ia < n) {
      const Op& op = ops[ia++];
      rc = launch_op(op, a, c);
      if (op.heavy) break;
    }
    while (!rc && ib < n) {
      const Op& op = ops[ib++];
      rc = launch_op(op, b, c);
      if (op.heavy) break;
    }
  }
  // also after a failed launch: the side stream may still be working on the caller's buffers, so the caller's stream
  // must not run (or free) past this call before it has drained -- and a capture must not be left forked
  if ((e = hipEventRecord(ss->join, b.st)) != hipSuccess) return rc ? rc : hip_fail(e, "join record");
  if ((e = hipStreamWaitEvent(a.st, ss->join, 0)) != hipSuccess) return rc ? rc : hip_fail(e, "join wait");
  return rc;
}

// Default schedule of a large batch: the launches that fill the chip (covariance, the big updates, the gradient pass)
// take the whole batch on the caller's stream, so their durations measure one kernel; the runs of small launches in
// between -- pivot, panel, restricted update, pivot, panel: latency-bound workgroups that end in half-empty rounds
// (633 pivots on 512 slots) -- are issued for the two halves of the batch on two streams, forked from and joined
// to the caller's stream around each run, so that one half's panel fills the slots the other half's pivots leave.
// Same kernels on the same per-GP data: bit-identical.  gpsr_set_split_small(0) / GPSR_NO_SPLIT_SMALL=1 turn it off.
static int g_split_small = getenv("GPSR_NO_SPLIT_SMALL") == nullptr ? 1 : 0;

extern "C" int gpsr_set_split_small(int on) {
  const int old = g_split_small;
  g_split_small = on ? 1 : 0;
  return old;
}

static Batch slice_batch(const Batch& h, int b0, int nb, const FitShape& c) {
  const int N = c.N, nblk = N / NB, ntri = nblk * (nblk + 1) / 2, per = (nblk - 1) * nblk / 2 + nblk - 1;
  Batch s = h;
  s.B = nb; s.Bdec = h.Bdec ? h.Bdec : h.B;
  const size_t o = (size_t)b0;
  if (h.X) s.X = h.X + o * c.n * c.d;
  if (h.y) s.y = h.y + o * c.n;
  if (h.raw) s.raw = h.raw + 4 * o;
  if (h.alpha) s.alpha = h.alpha + o * N;
  if (h.loss) s.loss = h.loss + o;
  if (h.grad) s.grad = h.grad + 4 * o;
  if (h.flags) s.flags = h.flags + o;
  s.w.A = h.w.A + o * N * N;
  s.w.inv.Wo = h.w.inv.Wo + o * NB * N;
  s.w.inv.Wo2 = h.w.inv.Wo2 + o * NB * N;
  for (int i = 0; i < 2; ++i) { s.w.inv.U[i] = h.w.inv.U[i] + o * NB * NB; s.w.inv.M[i] = h.w.inv.M[i] + o * NB * NB; }
  if (h.w.stats1) s.w.stats1 = h.w.stats1 + o * nblk * 4;
  if (h.w.stats2) s.w.stats2 = h.w.stats2 + o * ntri * 2;
  if (h.w.logdet) s.w.logdet = h.w.logdet + o;
  if (h.w.part) s.w.part = h.w.part + o * (per > 0 ? per : 1) * 2 * NB;
  if (h.w.m1) s.w.m1 = h.w.m1 + 4 * o;
  if (h.w.m2) s.w.m2 = h.w.m2 + 4 * o;
  return s;
}

static int run_split_small(const std::vector<Op>& ops, const Batch& full, const FitShape& c, SideStream* ss) {
  int rc = ensure_lds_attrs();
  if (rc) return rc;
  const int Ba = (full.B + 1) / 2;   // uneven cuts (512 + 121, 384 + 249, 256 + 377 of 633) measured the same or worse
  Batch a = slice_batch(full, 0, Ba, c), b = slice_batch(full, Ba, full.B - Ba, c);
  b.st = ss->s;
  hipError_t e;
  const size_t n = ops.size();
  size_t i = 0;
  while (i < n) {
    size_t j = i;
    bool has_pivot = false;
    while (j < n && !ops[j].heavy) { has_pivot |= ops[j].kind == OP_PIVOT; ++j; }
    if (j == i) { if ((rc = launch_op(ops[i], full, c))) return rc; ++i; continue; }
    if (!has_pivot) {   // grad1 / adam alone: nothing to overlap
      for (; i < j; ++i) if ((rc = launch_op(ops[i], full, c))) return rc;
      continue;
    }
    if ((e = hipEventRecord(ss->fork, full.st)) != hipSuccess) return hip_fail(e, "fork record");
    if ((e = hipStreamWaitEvent(b.st, ss->fork, 0)) != hipSuccess) return hip_fail(e, "fork wait");
    // op by op, first half then second half: the host alternates, the two streams drain side by side
    for (size_t t = i; t < j && !rc; ++t) {
      rc = launch_op(ops[t], a, c);
      if (!rc) rc = launch_op(ops[t], b, c);
    }
    // the join is recorded and waited for even when a launch failed: the side stream may hold work on the caller's
    // workspace, and a stream capture must not be left with an unjoined fork
    if ((e = hipEventRecord(ss->join, b.st)) != hipSuccess) return rc ? rc : hip_fail(e, "join record");
    if ((e = hipStreamWaitEvent(full.st, ss->join, 0)) != hipSuccess) return rc ? rc : hip_fail(e, "join wait");
    if (rc) return rc;
    i = j;
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
  Batch h = {};
  h.B = B; h.flags = flags; h.st = st;
  h.w.A = A; h.w.inv = inv_ws_carve((char*)workspace, B, n); h.w.logdet = logdet;
  FitShape c = {n, n, 0, 0, 0.f, false};
  std::vector<Op> ops;
  append_sweep_ops(ops, n / NB, false);
  return run_single(ops, h, c);
}

// ---------------------------------------------------------------- fit
static int check_fit_args(const double* X, const double* y, int B, int n, int d, int kid, const void* ws,
                          size_t wsb) {
  if (!X || !y || !ws) return arg_fail("null pointer");
  if (n < 2 || d < 1 || d > 64 || kid < 0 || kid > 4) return arg_fail("fit shape / kernel_id");
  if (gpsr_padded_n(n) > 16 * NB) return arg_fail("n too large (max 1792)");
  if (wsb < fit_ws_bytes(B, gpsr_padded_n(n), d)) return arg_fail("workspace too small");
  return 0;
}

// One launch per <= FUSED_MAX_IT Adam iterations (the per-iteration constants travel as a kernel argument), the last
// one also runs the evaluation at the final hyper-parameters.  The Adam moments of every GP live behind the slot
// workspace so that a fit of more iterations can continue across launches.
template <int KID>
static int launch_fused(const double* X, const double* y, int B, int n, int N, float* raw, double* alpha, double* loss,
                        int32_t* flags, float* m1g, float* m2g, int nit, int final_eval, const AdamTab& tab,
                        const FusedWs& w, int slots, hipStream_t st) {
  static bool attr_done[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  if (dev < 0 || dev >= 64 || !attr_done[dev]) {
    e = hipFuncSetAttribute((const void*)k_fit_fused<KID>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)PIVOT_LDS_BYTES);
    if (e != hipSuccess) return hip_fail(e, "fused lds attr");
    if (dev >= 0 && dev < 64) attr_done[dev] = true;
  }
  ProfRec* pr = prof_start(T_FUSED, st);
  hipLaunchKernelGGL((k_fit_fused<KID>), dim3(slots), dim3(MM_THREADS), PIVOT_LDS_BYTES, st, X, y, raw, alpha, loss, flags,
                     m1g, m2g, B, n, N, nit, final_eval, tab, w, g_pair_steps ? 1 : 0, g_fused_stagger, (float)(1.0 - 0.9),
                     (float)0.999, (float)(1.0 - 0.999), 1e-8f);
  prof_stop(pr, st);
  CHECK_LAUNCH("k_fit_fused");
  return 0;
}

static int run_fused(const double* X, const double* y, int B, int n, int N, int kid, float* raw, int iters, float lr,
                     double* alpha, double* loss, int32_t* flags, void* workspace, hipStream_t st) {
  const int slots = B < fused_slots() ? B : fused_slots();
  const FitWs fw = fit_ws_carve((char*)workspace, slots, N);
  FusedWs w = {fw.A, fw.inv.Wo, fw.inv.Wo2, fw.inv.U[0], fw.inv.M[0], fw.inv.U[1], fw.inv.M[1],
               fw.stats1, fw.stats2, fw.logdet, fw.part};
  float* m1g = (float*)((char*)workspace + fit_ws_bytes1(slots, N));
  float* m2g = (float*)((char*)m1g + align256((size_t)B * 16));
  hipError_t e = hipMemsetAsync(m1g, 0, 2 * align256((size_t)B * 16), st);
  if (e != hipSuccess) return hip_fail(e, "memset adam state");
  int it = 0;
  do {
    const int nit = iters - it < FUSED_MAX_IT ? iters - it : FUSED_MAX_IT;
    AdamTab tab;
    for (int q = 0; q < nit; ++q) {   // as OP_ADAM: step number it + q + 1
      const double b1 = 0.9, b2 = 0.999;
      const double bc1 = 1.0 - pow(b1, it + q + 1), bc2 = 1.0 - pow(b2, it + q + 1);
      tab.step[q] = (float)((double)lr / bc1);
      tab.bc2s[q] = (float)sqrt(bc2);
    }
    for (int q = nit; q < FUSED_MAX_IT; ++q) { tab.step[q] = 0.f; tab.bc2s[q] = 1.f; }
    const int fin = it + nit >= iters ? 1 : 0;
    int rc;
#define FUSED(K) rc = launch_fused<K>(X, y, B, n, N, raw, alpha, loss, flags, m1g, m2g, nit, fin, tab, w, slots, st)
    switch (kid) {
      case K_RBF: FUSED(K_RBF); break;
      case K_M12: FUSED(K_M12); break;
      case K_M32: FUSED(K_M32); break;
      case K_M52: FUSED(K_M52); break;
      default: FUSED(K_LIN); break;
    }
#undef FUSED
    if (rc) return rc;
    it += nit;
  } while (it < iters);
  return 0;
}

#define DUAL_MIN_B 128   // below this a half-batch no longer fills the chip by itself

extern "C" int gpsr_fit(const double* X, const double* y, int B, int n, int d, int kernel_id, float* raw,
                        int iters, float lr, double* alpha, double* loss, int32_t* flags, void* workspace,
                        size_t workspace_bytes, void* stream) {
  if (B <= 0) return 0;
  int rc = check_fit_args(X, y, B, n, d, kernel_id, workspace, workspace_bytes);
  if (rc) return rc;
  if (!raw || !alpha || !flags || iters < 0) return arg_fail("fit outputs");
  hipStream_t st = (hipStream_t)stream;
  const int N = gpsr_padded_n(n), nblk = N / NB;
  hipError_t e = hipMemsetAsync(flags, 0, (size_t)B * 4, st);
  if (e != hipSuccess) return hip_fail(e, "memset flags");
  if (fused_applies(B, N, d))
    return run_fused(X, y, B, n, N, kernel_id, raw, iters, lr, alpha, loss, flags, workspace, st);
  // `iters` Adam steps, then the final factorisation at the post-step hyper-parameters -> alpha
  std::vector<Op> ops;
  for (int it = 1; it <= iters; ++it) {
    append_eval_ops(ops, nblk, true);
    ops.push_back({OP_ADAM, 0, 0, it, false});
  }
  append_eval_ops(ops, nblk, false);
  FitShape c = {n, N, d, kernel_id, lr, nblk > 1};
  SideStream* ss = (g_dual_stream && nblk > 1 && B >= 2 * DUAL_MIN_B) ? side_stream() : nullptr;
  const int Ba = ss ? (B + 1) / 2 : B;
  Batch a = {};
  a.B = Ba; a.X = X; a.y = y; a.raw = raw; a.alpha = alpha; a.loss = loss; a.flags = flags; a.st = st;
  a.w = fit_ws_carve((char*)workspace, Ba, N);
  e = hipMemsetAsync(a.w.m1, 0, 2 * align256((size_t)Ba * 16), st);
  if (e != hipSuccess) return hip_fail(e, "memset adam state");
  if (!ss) {
    SideStream* sp = (g_split_small && nblk > 1 && B >= 2 * DUAL_MIN_B) ? side_stream() : nullptr;
    return sp ? run_split_small(ops, a, c, sp) : run_single(ops, a, c);
  }
  Batch b = {};
  b.B = B - Ba; b.X = X + (size_t)Ba * n * d; b.y = y + (size_t)Ba * n; b.raw = raw + 4 * (size_t)Ba;
  b.alpha = alpha + (size_t)Ba * N; b.loss = loss ? loss + Ba : nullptr; b.flags = flags + Ba;
  b.w = fit_ws_carve((char*)workspace + fit_ws_bytes1(Ba, N), b.B, N);
  e = hipMemsetAsync(b.w.m1, 0, 2 * align256((size_t)b.B * 16), st);   // on the caller's stream: before the fork
  if (e != hipSuccess) return hip_fail(e, "memset adam state");
  return run_dual(ops, a, b, c, ss);
}

// ---------------------------------------------------------------- fit without any allocation
// For callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
// that forbid it): gpsr_fit with its workspace in a file-scope __device__ array instead of a caller / context
// allocation.  Each GPU the library runs on holds its own copy of the array, at its own address, until the
// process ends; the host-side name `g_noalloc_scratch` is only a handle for hipGetSymbolAddress -- the device
// address it returns (per device, looked up once with that device current) is what kernels and memsets get.
#define GPSR_NOALLOC_BYTES ((size_t)64 << 20)
static __device__ __align__(256) unsigned char g_noalloc_scratch[GPSR_NOALLOC_BYTES];
static void* g_noalloc_ptr[64] = {};
#include <mutex>
static std::mutex g_noalloc_mu;

// The array the kernels of this call must use is the copy on the device the STREAM belongs to (the launches go
// there), which need not be the thread's current device: a non-null stream is asked for its device, and a stream of
// another device than the current one is refused (hipGetSymbolAddress answers for the current device only, and the
// launch itself would fail).  The null stream is the current device's.
static int noalloc_scratch(hipStream_t st, void** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
  if (st) {
    hipDevice_t sdev = -1;
    e = hipStreamGetDevice(st, &sdev);
    if (e != hipSuccess) return hip_fail(e, "hipStreamGetDevice");
    if ((int)sdev != dev) {
      snprintf(g_err, sizeof(g_err), "gpsr_fit_noalloc: the stream belongs to device %d, the current device is %d "
               "(hipSetDevice to the stream's device first)", (int)sdev, dev);
      return -1;
    }
  }
  if (dev < 0 || dev >= 64) return arg_fail("device index out of range for the static scratch table");
  std::lock_guard<std::mutex> lock(g_noalloc_mu);
  if (!g_noalloc_ptr[dev]) {
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
    if (e != hipSuccess || !p) return hip_fail(e, "hipGetSymbolAddress(static scratch)");
    g_noalloc_ptr[dev] = p;
  }
  *out = g_noalloc_ptr[dev];
  return 0;
}

extern "C" size_t gpsr_noalloc_scratch_bytes(void) { return GPSR_NOALLOC_BYTES; }

extern "C" int gpsr_fit_noalloc(const double* X, const double* y, int B, int n, int d, int kernel_id, float* raw,
                                int iters, float lr, double* alpha, double* loss, int32_t* flags, void* stream) {
  if (B <= 0) return 0;
  if (!X || !y) return arg_fail("null pointer");
  if (n < 2 || d < 1 || d > 64 || kernel_id < 0 || kernel_id > 4) return arg_fail("fit shape / kernel_id");
  const int N = gpsr_padded_n(n), nblk = N / NB;
  if (N > 16 * NB) return arg_fail("n too large (max 1792)");
  if (!raw || !alpha || !flags || iters < 0) return arg_fail("fit outputs");
  const size_t need = fit_ws_bytes1(B, N);
  if (need > GPSR_NOALLOC_BYTES) {
    snprintf(g_err, sizeof(g_err), "gpsr_fit_noalloc: %d GPs of n=%d need %zu bytes of scratch, the static array holds %zu "
             "(use gpsr_fit with a caller workspace)", B, n, need, (size_t)GPSR_NOALLOC_BYTES);
    return -1;
  }
  void* ws = nullptr;
  hipStream_t st = (hipStream_t)stream;
  int rc = noalloc_scratch(st, &ws);
  if (rc) return rc;
  hipError_t e = hipMemsetAsync(flags, 0, (size_t)B * 4, st);
  if (e != hipSuccess) return hip_fail(e, "memset flags");
  Batch a = {};
  a.B = B; a.X = X; a.y = y; a.raw = raw; a.alpha = alpha; a.loss = loss; a.flags = flags; a.st = st;
  a.w = fit_ws_carve((char*)ws, B, N);
  // the array keeps the last call's contents: the Adam moments must start at zero (everything else the
  // kernels read is written earlier in the same call: A by k_build, logdet by k_build, panels by k_panel ...)
  e = hipMemsetAsync(a.w.m1, 0, 2 * align256((size_t)B * 16), st);
  if (e != hipSuccess) return hip_fail(e, "memset adam state");
  FitShape c = {n, N, d, kernel_id, lr, nblk > 1};
  // the launch list of ONE evaluation on the stack (no heap): <= 1 + 6*(nblk/2) + 3*2 + 2 entries
  Op ops[64];
  int nops = 0;
  {
    const int npairs = (g_pair_steps && nblk >= 3) ? (nblk - 1 - (nblk % 2 == 0 ? 1 : 0)) / 2 : 0;
    ops[nops++] = {OP_BUILD, 0, 0, 0, true};
    int k = 0;
    for (int mth = 0; mth < npairs; ++mth, k += 2) {
      ops[nops++] = {OP_PIVOT, k, 0, 0, false};
      ops[nops++] = {OP_PANEL, k, 0, 0, false};
      ops[nops++] = {OP_UPD_SEL, k, 0, 0, false};
      ops[nops++] = {OP_PIVOT, k + 1, 1, 0, false};
      ops[nops++] = {OP_PANEL, k + 1, 1, 0, false};
      ops[nops++] = {OP_UPD2, k, 0, 0, true};
    }
    for (; k < nblk; ++k) {
      ops[nops++] = {OP_PIVOT, k, 0, 0, false};
      if (nblk > 1) {
        ops[nops++] = {OP_PANEL, k, 0, 0, false};
        ops[nops++] = {OP_UPD, k, 0, (k == nblk - 1) ? 1 : 0, true};
      }
    }
    ops[nops++] = {OP_GRAD1, 0, 0, 0, false};
    mark_pivots(ops, nops);
  }
  if ((rc = ensure_lds_attrs())) return rc;
  const Op grad2 = {OP_GRAD2, 0, 0, 0, true};
  for (int it = 1; it <= iters + 1; ++it) {
    for (int i = 0; i < nops; ++i)
      if ((rc = launch_op(ops[i], a, c))) return rc;
    if (it <= iters) {
      if ((rc = launch_op(grad2, a, c))) return rc;
      const Op adam = {OP_ADAM, 0, 0, it, false};
      if ((rc = launch_op(adam, a, c))) return rc;
    }
  }
  return 0;
}

extern "C" int gpsr_loss_grad(const double* X, const double* y, int B, int n, int d, int kernel_id,
                              const float* raw, double* loss, float* grad, int32_t* flags, void* workspace,
                              size_t workspace_bytes, void* stream) {
  if (B <= 0) return 0;
  int rc = check_fit_args(X, y, B, n, d, kernel_id, workspace, workspace_bytes);
  if (rc) return rc;
  if (!raw || !loss || !grad || !flags) return arg_fail("loss_grad outputs");
  hipStream_t st = (hipStream_t)stream;
  const int N = gpsr_padded_n(n), nblk = N / NB;
  hipError_t e = hipMemsetAsync(flags, 0, (size_t)B * 4, st);
  if (e != hipSuccess) return hip_fail(e, "memset flags");
  Batch h = {};
  h.B = B; h.X = X; h.y = y; h.raw = (float*)raw; h.loss = loss; h.grad = grad; h.flags = flags; h.st = st;
  h.w = fit_ws_carve((char*)workspace, B, N);
  h.alpha = h.w.inv.Wo;  // [B][N] fits in [B][NB][N]; Wo is dead after the sweeps
  std::vector<Op> ops;
  append_eval_ops(ops, nblk, true);
  ops.push_back({OP_ADAM, 0, 0, 0, false});
  FitShape c = {n, N, d, kernel_id, 0.f, nblk > 1};
  return run_single(ops, h, c);
}

// ---------------------------------------------------------------- context + flag count
struct gpsr_ctx { int device; void* ws; size_t bytes; };

static size_t pix_ws_carve(char* base, int B, int Bc, int n, int m, struct PixWs* w);

extern "C" gpsr_ctx* gpsr_create(int device_id, int max_B, int max_n, int max_m) {
  if (max_B < 1 || max_n < 2 || max_m < 0 || gpsr_padded_n(max_n) > 16 * NB) { arg_fail("gpsr_create sizes"); return nullptr; }
  hipError_t e = hipSetDevice(device_id);
  if (e != hipSuccess) { hip_fail(e, "hipSetDevice"); return nullptr; }
  size_t bytes = gpsr_fit_workspace_bytes(max_B, max_n, 8);
  if (max_m > 0) {   // gpsr_predict_pixels: locs, raw, pred, X, y, Xs, alpha on top of the fit workspace (chunk = all patches)
    const size_t pix = pix_ws_carve(nullptr, max_B, max_B, max_n, max_m, nullptr);
    if (pix > bytes) bytes = pix;
  }
  gpsr_ctx* c = new gpsr_ctx{device_id, nullptr, bytes};
  e = hipMalloc(&c->ws, c->bytes);
  if (e != hipSuccess) { hip_fail(e, "hipMalloc(workspace)"); delete c; return nullptr; }
  return c;
}
extern "C" void gpsr_destroy(gpsr_ctx* ctx) {
  if (!ctx) return;
  (void)hipFree(ctx->ws);
  delete ctx;
}
extern "C" void* gpsr_ctx_workspace(const gpsr_ctx* ctx) { return ctx ? ctx->ws : nullptr; }
extern "C" size_t gpsr_ctx_workspace_bytes(const gpsr_ctx* ctx) { return ctx ? ctx->bytes : 0; }

extern "C" int gpsr_flag_count(const int32_t* flags, int B, void* stream) {
  if (!flags || B < 0) return arg_fail("gpsr_flag_count");
  if (B == 0) return 0;
  std::vector<int32_t> host((size_t)B);
  hipError_t e = hipMemcpyAsync(host.data(), flags, (size_t)B * 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "flag copy");
  e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return hip_fail(e, "flag sync");
  int cnt = 0;
  for (int b = 0; b < B; ++b) cnt += host[(size_t)b] != 0;
  return cnt;
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
  if (nr < 1 || nc < 1 || p < 3 || s < 1 || H < 1 || W < 1 || (mode != 0 && mode != 1)) return arg_fail("combine geometry");
  const int Hs = H * s, Ws = W * s;
  const size_t lds = (size_t)((nr + 31) / 32 + (nc + 31) / 32) * sizeof(uint32_t);
  if ((Hs + CMB_TH - 1) / CMB_TH > 65535 || lds > 48 * 1024) return arg_fail("combine: image or patch grid too large");
  hipLaunchKernelGGL(k_combine, dim3((Ws + CMB_TW - 1) / CMB_TW, (Hs + CMB_TH - 1) / CMB_TH), dim3(CMB_TH * CMB_TW), lds,
                     (hipStream_t)stream, pred, test_in, H, W, s, p, rows, nr, cols, nc, mode, clip, out);
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
  return off + fit_ws_bytes(Bc, N, 8);
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
  const size_t fit_bytes = fit_ws_bytes(Bc, N, 8);
  for (int b0 = 0; b0 < B; b0 += Bc) {
    const int bn = (B - b0 < Bc) ? B - b0 : Bc;
    int rc = gpsr_gather(train_in, train_tgt, test_in, H, W, s, p, w.locs + 2 * b0, bn, tidx, nt, w.X, w.y, w.Xs, stream);
    if (rc) return rc;
    rc = gpsr_init_hypers(w.y, bn, n, kernel_id, init_raw_os, init_raw_ls, nullptr, 0, raw + 4 * b0, stream);
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

// ---------------------------------------------------------------- uint8 colour stages + metrics
extern "C" int gpsr_bgr2gray(const uint8_t* bgr, uint8_t* gray, size_t npx, void* stream) {
  if (!bgr || !gray) return arg_fail("null pointer");
  if (npx == 0) return 0;
  hipLaunchKernelGGL(k_bgr2gray, GRID1D(npx), bgr, gray, npx);
  CHECK_LAUNCH("k_bgr2gray");
  return 0;
}

static int launch_bgr2ycc(const uint8_t* bgr, uint8_t* Y, uint8_t* CR, uint8_t* CB, size_t npx, int c_r, int c_b,
                          void* stream) {
  if (!bgr || !Y || !CR || !CB) return arg_fail("null pointer");
  if (npx == 0) return 0;
  hipLaunchKernelGGL(k_bgr2ycc, GRID1D(npx), bgr, Y, CR, CB, npx, c_r, c_b);
  CHECK_LAUNCH("k_bgr2ycc");
  return 0;
}
static int launch_ycc2bgr(const uint8_t* Y, const uint8_t* CR, const uint8_t* CB, uint8_t* bgr, size_t npx, int c0,
                          int c1, int c2, int c3, void* stream) {
  if (!bgr || !Y || !CR || !CB) return arg_fail("null pointer");
  if (npx == 0) return 0;
  hipLaunchKernelGGL(k_ycc2bgr, GRID1D(npx), Y, CR, CB, bgr, npx, c0, c1, c2, c3);
  CHECK_LAUNCH("k_ycc2bgr");
  return 0;
}
extern "C" int gpsr_bgr2yuv(const uint8_t* bgr, uint8_t* Y, uint8_t* U, uint8_t* V, size_t npx, void* stream) {
  return launch_bgr2ycc(bgr, Y, V, U, npx, 14369, 8061, stream);       // V = .877 (R-Y), U = .492 (B-Y)
}
extern "C" int gpsr_yuv2bgr(const uint8_t* Y, const uint8_t* U, const uint8_t* V, uint8_t* bgr, size_t npx, void* stream) {
  return launch_ycc2bgr(Y, V, U, bgr, npx, 18678, -9519, -6472, 33292, stream);
}
extern "C" int gpsr_bgr2ycrcb(const uint8_t* bgr, uint8_t* Y, uint8_t* Cr, uint8_t* Cb, size_t npx, void* stream) {
  return launch_bgr2ycc(bgr, Y, Cr, Cb, npx, 11682, 9241, stream);
}
extern "C" int gpsr_ycrcb2bgr(const uint8_t* Y, const uint8_t* Cr, const uint8_t* Cb, uint8_t* bgr, size_t npx, void* stream) {
  return launch_ycc2bgr(Y, Cr, Cb, bgr, npx, 22987, -11698, -5636, 29049, stream);
}

extern "C" int gpsr_resample_u8(const uint8_t* src, int Hs, int Ws, uint8_t* dst, int Hd, int Wd, const int32_t* xi,
                                const float* xw, const int32_t* yi, const float* yw, void* stream) {
  if (!src || !dst || !xi || !xw || !yi || !yw) return arg_fail("null pointer");
  if (Hs < 1 || Ws < 1 || Hd < 1 || Wd < 1) return arg_fail("resample_u8 shape");
  hipLaunchKernelGGL(k_resample_u8, GRID1D((size_t)Hd * Wd), src, Ws, dst, Hd, Wd, xi, xw, yi, yw);
  CHECK_LAUNCH("k_resample_u8");
  return 0;
}

extern "C" int gpsr_sepfilter_u8(const uint8_t* src, int H, int W, uint8_t* dst, const int32_t* kern, int ksize,
                                 void* stream) {
  if (!src || !dst || !kern || src == dst) return arg_fail("sepfilter_u8: null pointer or in-place call");
  if (H < 1 || W < 1 || ksize < 1 || ksize > 7 || ksize % 2 == 0) return arg_fail("sepfilter_u8: odd ksize <= 7");
  hipLaunchKernelGGL(k_sepfilter_u8, GRID1D((size_t)H * W), src, H, W, dst, kern, ksize);
  CHECK_LAUNCH("k_sepfilter_u8");
  return 0;
}

static size_t metrics_parts(int H, int W, int crop) {
  const size_t Hc = (size_t)(H - 2 * crop), Wc = (size_t)(W - 2 * crop);
  return (Hc * Wc + MET_THREADS - 1) / MET_THREADS;
}
extern "C" size_t gpsr_metrics_workspace_bytes(int H, int W, int crop) {
  if (crop < 0 || H - 2 * crop < 1 || W - 2 * crop < 1) return 0;
  return 2 * align256(metrics_parts(H, W, crop) * 8);
}
extern "C" int gpsr_metrics(const uint8_t* a, const uint8_t* b, int H, int W, int crop, double* out, void* workspace,
                            size_t workspace_bytes, void* stream) {
  if (!a || !b || !out || !workspace) return arg_fail("null pointer");
  if (crop < 0 || H - 2 * crop < 1 || W - 2 * crop < 1) return arg_fail("metrics: image smaller than the crop");
  if (workspace_bytes < gpsr_metrics_workspace_bytes(H, W, crop)) return arg_fail("workspace too small");
  const int Hc = H - 2 * crop, Wc = W - 2 * crop;
  const size_t parts = metrics_parts(H, W, crop);
  if (parts > 0x7fffffff) return arg_fail("metrics: image too large");
  unsigned long long* sse = (unsigned long long*)workspace;
  double* ssp = (double*)((char*)workspace + align256(parts * 8));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(k_metrics_partial, dim3((unsigned)parts), dim3(MET_THREADS), 0, st, a, b, W, crop, Hc, Wc, sse, ssp);
  CHECK_LAUNCH("k_metrics_partial");
  const double nwin = (Hc >= 7 && Wc >= 7) ? (double)(Hc - 6) * (double)(Wc - 6) : 0.0;
  hipLaunchKernelGGL(k_metrics_final, dim3(1), dim3(MET_THREADS), 0, st, sse, ssp, (int)parts, (double)Hc * (double)Wc,
                     nwin, out);
  CHECK_LAUNCH("k_metrics_final");
  return 0;
}

// ---------------------------------------------------------------- AGPR training-set selection
static void as_carve(void* workspace, int N, double** dist, double** rep, double** dv, int32_t** orig) {
  char* w = (char*)workspace;
  *dist = (double*)w; w += align256((size_t)N * N * 8);
  *rep = (double*)w;  w += align256((size_t)N * 8);
  *dv = (double*)w;   w += align256((size_t)N * 8);
  *orig = (int32_t*)w;
}
extern "C" size_t gpsr_active_sample_workspace_bytes(int N) {
  if (N < 2) return 0;
  return align256((size_t)N * N * 8) + 2 * align256((size_t)N * 8) + align256((size_t)N * 4);
}
extern "C" int gpsr_active_sample(const double* data, int N, int D, int K, int n_pick, double bandwidth, double w_rep,
                                  double w_div, int32_t* out_idx, double* rep_out, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  if (n_pick == 0) return 0;
  if (!data || !out_idx || !workspace) return arg_fail("null pointer");
  if (N < 2 || N > 46340) return arg_fail("active_sample: N must be in 2..46340");
  if (D < 1 || D > AS_MAXD) return arg_fail("active_sample: D must be in 1..128 (numpy's unrolled summation block)");
  if (K < 1 || K > AS_MAXK || K >= N) return arg_fail("active_sample: K must be in 1..64 and below N");
  if (n_pick < 0 || n_pick > N) return arg_fail("active_sample: n_pick must be in 0..N");
  if (!(bandwidth > 0.0)) return arg_fail("active_sample: bandwidth must be positive");
  if (workspace_bytes < gpsr_active_sample_workspace_bytes(N)) return arg_fail("workspace too small");
  double *dist, *rep, *dv; int32_t* orig;
  as_carve(workspace, N, &dist, &rep, &dv, &orig);
  hipStream_t st = (hipStream_t)stream;
  const unsigned g = (unsigned)((N + 15) / 16);
  hipLaunchKernelGGL(k_as_dist, dim3(g, g), dim3(256), (size_t)(2 * 16 * (D | 1)) * sizeof(double), st, data, N, D, dist);
  CHECK_LAUNCH("k_as_dist");
  const double two_bw = 2 * bandwidth;   // agpr.py:128,222 divide by (2*self.bandwidth)
  hipLaunchKernelGGL(k_as_rep, dim3((unsigned)N), dim3(256), 0, st, dist, N, K, two_bw, rep);
  CHECK_LAUNCH("k_as_rep");
  hipLaunchKernelGGL(k_as_greedy, dim3(1), dim3(AS_THREADS), 0, st, dist, rep, N, n_pick, two_bw, w_rep, w_div, orig, dv,
                     out_idx);
  CHECK_LAUNCH("k_as_greedy");
  if (rep_out) {
    hipError_t e = hipMemcpyAsync(rep_out, rep, (size_t)N * 8, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return hip_fail(e, "active_sample: copy of the scores");
  }
  return 0;
}

#ifdef GPSR_UPD_TRACE
// debug build only (tools/upd_trace.py): read (and clear) the k_update2 cycle counters
extern "C" int gpsr_debug_upd_trace(unsigned long long* out8) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_upd_trace), 8 * sizeof(unsigned long long));
  if (e != hipSuccess) return hip_fail(e, "trace read");
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_upd_trace), z, sizeof(z));
  return e == hipSuccess ? 0 : hip_fail(e, "trace clear");
}
#endif

#ifdef GPSR_FUSED_TRACE
// debug build only (tools/fused_trace.py): read (and clear) the k_fit_fused role cycle counters
extern "C" int gpsr_debug_fused_trace(unsigned long long* out16) {
  hipError_t e = hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_fused_trace), 16 * sizeof(unsigned long long));
  if (e != hipSuccess) return hip_fail(e, "trace read");
  unsigned long long z[16] = {};
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_fused_trace), z, sizeof(z));
  return e == hipSuccess ? 0 : hip_fail(e, "trace clear");
}
#endif

#ifdef GPSR_PIV_TRACE
// debug build only (tools/piv_trace.py): read (and clear) the k_pivot phase cycle counters
extern "C" int gpsr_debug_piv_trace(unsigned long long* out8) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_piv_trace), 8 * sizeof(unsigned long long));
  if (e != hipSuccess) return hip_fail(e, "trace read");
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_piv_trace), z, sizeof(z));
  return e == hipSuccess ? 0 : hip_fail(e, "trace clear");
}
extern "C" int gpsr_debug_piv_trace2(unsigned long long* out8) {
  hipError_t e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_piv_trace2), 8 * sizeof(unsigned long long));
  if (e != hipSuccess) return hip_fail(e, "trace2 read");
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_piv_trace2), z, sizeof(z));
  return e == hipSuccess ? 0 : hip_fail(e, "trace2 clear");
}
#endif
