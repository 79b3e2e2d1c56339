/*
 * gpsr.h -- C ABI of the MI355X (gfx950) batched Gaussian-process
 * super-resolution engine.
 *
 * The reference (fynsta/Super-resolution) is pure Python on GPyTorch and has no
 * FFI of its own; each entry point below names the reference code it replaces
 * (file:line in the reference tree).  SURVEY.md 8(b) is the contract:
 *   - extern "C", plain pointers and sizes, no torch types;
 *   - every function returns int: 0 ok (launches enqueued), <0 bad argument / HIP error
 *     (text in gpsr_last_error()).  Numerical failures are per GP and asynchronous: a GP
 *     whose factorisation hits a non-positive pivot gets flags[b] != 0 in the caller's
 *     flags[B]; gpsr_flag_count() is the one call that waits for the stream and returns
 *     how many (the "positive return" of SURVEY.md 8(b));
 *   - the caller owns every buffer (device pointers), including the workspace -- either
 *     its own allocation or the one an opaque gpsr_ctx (gpsr_create / gpsr_destroy) holds;
 *   - every launch goes on the hipStream_t passed in (as void*); calls are
 *     asynchronous and, gpsr_flag_count / gpsr_profile_end aside, never synchronise.
 *
 * Data are float64 images / features; hyper-parameters are the float32 RAW
 * (unconstrained) values GPyTorch keeps:  raw[b][0..3] =
 *   { likelihood.noise_covar.raw_noise, mean_module.raw_constant,
 *     covar_module.raw_outputscale, covar_module.base_kernel.raw_lengthscale
 *     (raw_variance for the linear kernel) }
 * with noise = softplus(raw)+1e-4 and the others softplus(raw).
 */
#ifndef GPSR_H
#define GPSR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* kernel_id: reference kernels.py:7-13 (get_kernel_name tags kernels.py:26-47) */
enum {
  GPSR_KERNEL_RBF = 0,      /* rbf_kernel          "RBF"      */
  GPSR_KERNEL_MATERN12 = 1, /* exponential_kernel  "matern12" */
  GPSR_KERNEL_MATERN32 = 2, /* matern32_kernel     "matern32" */
  GPSR_KERNEL_MATERN52 = 3, /* matern52_kernel     "matern52" */
  GPSR_KERNEL_LINEAR = 4    /* linear_kernel       "linear"   */
};

/* PatchInterpolationMode, image_patches.py:4-6 */
enum { GPSR_COMBINE_AVERAGE = 0, GPSR_COMBINE_BILINEAR = 1 };

#define GPSR_BLOCK 112 /* pivot-block / tile edge of the batched inversion */
/*
 * Matrix layout of K / A / dK_dl below: TILE-BLOCKED.  A matrix of padded order N = nblk * GPSR_BLOCK is stored as
 * [nblk][nblk][GPSR_BLOCK][GPSR_BLOCK] doubles: tile (I, J) is one contiguous row-major 112 x 112 block at offset
 * (I * nblk + J) * 112 * 112, element (i, j) of the matrix sits in tile (i / 112, j / 112) at [i % 112][j % 112].
 * (N * N doubles per matrix, as for row-major; for N = 112 the two layouts coincide.)  Only tiles with I >= J are
 * written / maintained ("lower block-triangle"), diagonal tiles full.
 */

/* Library version string ("gpsr <x.y> gfx950"). */
const char* gpsr_version(void);

/* Last HIP error text seen by this thread's most recent failing call. */
const char* gpsr_last_error(void);

/*
 * Opaque context = a device and one workspace allocation, for callers that do not bring
 * their own allocator.  It is sized for gpsr_fit / gpsr_loss_grad /
 * gpsr_invert of up to max_B GPs of up to max_n training points (d <= 64 features) and,
 * when max_m > 0, also for gpsr_predict_pixels of up to max_B patches (any chunk) with
 * n <= max_n training and m <= max_m test points per patch (m = (p*s-2)^2); max_m = 0
 * leaves gpsr_predict_pixels out (it would answer "workspace too small").  gpsr_create makes
 * device_id current and is the only allocating call of the library (one hipMalloc);
 * gpsr_destroy frees it.  Pass gpsr_ctx_workspace() / gpsr_ctx_workspace_bytes() as the
 * `workspace` arguments below.  One context per (device, stream): calls that share a
 * workspace must be ordered on one stream.  Returns NULL on failure (gpsr_last_error()).
 * No counterpart in the reference (GPyTorch allocates through torch).
 */
typedef struct gpsr_ctx gpsr_ctx;
gpsr_ctx* gpsr_create(int device_id, int max_B, int max_n, int max_m);
void gpsr_destroy(gpsr_ctx* ctx);
void* gpsr_ctx_workspace(const gpsr_ctx* ctx);
size_t gpsr_ctx_workspace_bytes(const gpsr_ctx* ctx);

/* Waits for `stream` and returns the number of non-zero entries of flags[B] (device
 * pointer): the per-GP numerical failures of the calls enqueued before; <0 on error. */
int gpsr_flag_count(const int32_t* flags, int B, void* stream);

/* Padded matrix order used for n training points: n rounded up to GPSR_BLOCK. */
int gpsr_padded_n(int n);

/*
 * K1 gather.  Replaces PatchIterator.__next__ (image_patches.py:92-107) plus the
 * feature list-comprehensions of gprsr.py:148,150,192 / extract_neighbors
 * (gprsr.py:34-35) for B patches at once.
 *   train_in, train_tgt : [H][W] f64      test_in : [H*s][W*s] f64
 *   locs  : [B][2] int32 (row, col) LR patch corners (PatchHandler.patch_locations)
 *   tidx  : [nt] int32 per-axis training indices inside the LR patch
 *           (range(1,p-1), or the grid of gprsr.py:140); n = nt*nt
 *   X  : [B][n][8]   y : [B][n]   Xs : [B][m][8],  m = (p*s-2)^2   (X, Xs 16-byte aligned)
 * Feature order is that of gprsr.py:35.
 */
int gpsr_gather(const double* train_in, const double* train_tgt, const double* test_in,
                int H, int W, int s, int p, const int32_t* locs, int B,
                const int32_t* tidx, int nt,
                double* X, double* y, double* Xs, void* stream);

/*
 * Per-patch initial hyper-parameters, gprsr.py:162-174: noise=1e-3,
 * constant=mean(y.float()); RBF: lengthscale=.223, outputscale=var(y.float());
 * other kernels: raw_outputscale/raw_lengthscale = init_raw_os/init_raw_ls, or, when
 * carry != NULL, the two floats at carry[b*carry_stride + 2], carry[b*carry_stride + 3]
 * (device pointer to previous GPs' raw[4] rows -- the reference's warm-start chain through
 * the kernel singletons, kernels.py:7-13 / SURVEY.md F5).  carry_stride 0: every GP of the
 * call continues the same chain link; carry_stride >= 4: GP b continues the chain whose last
 * raw row is carry + b*carry_stride (B chains stepped at once, one GP each).
 */
int gpsr_init_hypers(const double* y, int B, int n, int kernel_id,
                     float init_raw_os, float init_raw_ls, const float* carry, int carry_stride,
                     float* raw, void* stream);

/*
 * K2 covariance.  K[b] = outputscale * k(X_b, X_b) + noise * I  as the lower
 * block-triangle (GPSR_BLOCK tiles, diagonal tiles full) of a tile-blocked matrix of order
 * N = gpsr_padded_n(n); rows/cols >= n are identity.  Replaces GPyTorch's
 * kernel forward + likelihood noise reached from kernels.py:21-24 and
 * gprsr.py:183-184.   X : [B][n][d]   raw : [B][4]   K : [B][N][N]
 */
int gpsr_covariance(const double* X, int B, int n, int d, int kernel_id,
                    const float* raw, double* K, void* stream);

/*
 * K2 with derivative: K as above (or NULL) and dK_dl[b] = outputscale * d k / d lengthscale
 * (d k / d variance for the linear kernel), same layout, zero in the padding: the matrix
 * whose trace against (K^-1 - alpha alpha^T) is the lengthscale gradient GPyTorch's autograd
 * produces at gprsr.py:185.  The fit itself never materialises it (k_grad2 fuses it).
 */
int gpsr_covariance_grad(const double* X, int B, int n, int d, int kernel_id,
                         const float* raw, double* K, double* dK_dl, void* stream);

/*
 * In-place batched SPD inversion by blocked symmetric Gauss-Jordan sweeps
 * (MFMA f64 trailing updates).  A: tile-blocked (see above).  On return A[b] holds -inv(A[b]) in the lower
 * block-triangle, logdet[b] = log det A[b].  Workspace: gpsr_invert_workspace_bytes.
 * This is the factorisation the fit uses instead of GPyTorch's dense Cholesky
 * (linear_operator psd_safe_cholesky reached from gprsr.py:184-185).
 */
size_t gpsr_invert_workspace_bytes(int B, int n);
int gpsr_invert(double* A, int B, int n, double* logdet, int32_t* flags,
                void* workspace, size_t workspace_bytes, void* stream);

/*
 * K3+K4 fit.  Replaces the training loop gprsr.py:176-186 (agpr.py:75-83):
 * `iters` Adam steps (lr, betas .9/.999, eps 1e-8, float32 state) on
 * -ExactMarginalLogLikelihood/n, then one more factorisation at the final
 * hyper-parameters giving alpha = (K+noise I)^-1 (y - c) for gprsr.py:195.
 *   X : [B][n][d]  y : [B][n]  raw : [B][4] in/out
 *   alpha : [B][N] out (entries >= n are 0)   loss : [B] out or NULL (last evaluated loss)
 *   flags : [B] int32 out (non-zero: a pivot was <= 0 at some iteration)
 */
size_t gpsr_fit_workspace_bytes(int B, int n, int d);
int gpsr_fit(const double* X, const double* y, int B, int n, int d, int kernel_id,
             float* raw, int iters, float lr, double* alpha, double* loss, int32_t* flags,
             void* workspace, size_t workspace_bytes, void* stream);

/*
 * gpsr_fit that allocates nothing.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  Same arguments and results (bit-identical) as gpsr_fit minus the workspace: the scratch
 * (covariance matrices, panels, pivot factors, statistics, Adam moments) lives in a file-scope __device__
 * array of gpsr_noalloc_scratch_bytes() = 64 MiB that every GPU the library runs on holds until the process
 * ends.  Limit: gpsr_fit_workspace_bytes-like need of the batch <= 64 MiB, i.e. about 9 GPs of n = 784, 33 of
 * n = 324, 130 of n <= 112; larger batches are rejected (<0, text in gpsr_last_error()).  The call makes no
 * hipMalloc / hipFree (no allocating routine at all: its launch list lives on the stack), never runs
 * dual-stream, and is asynchronous on `stream`.  All calls on one GPU share that GPU's array: calls that
 * overlap on one GPU (two streams, two threads) race -- order them on one stream.
 */
size_t gpsr_noalloc_scratch_bytes(void);
int gpsr_fit_noalloc(const double* X, const double* y, int B, int n, int d, int kernel_id,
                     float* raw, int iters, float lr, double* alpha, double* loss, int32_t* flags,
                     void* stream);

/*
 * Opt-in: gpsr_fit runs a batch of >= 256 GPs as two half-batches on two streams (the caller's and an
 * internal one, forked from and joined to it with events), which fills part of the machine that the small
 * latency-bound launches of one batch leave idle (+5 % throughput on Set14 x3).  Results are bit-identical.
 * Off by default: overlapping launches make per-launch timings (gpsr_profile_*) ambiguous.  Process-wide,
 * returns the previous setting.  No counterpart in the reference.
 */
int gpsr_set_dual_stream(int on);

/*
 * On by default: gpsr_fit on a batch of >= 256 GPs issues the runs of SMALL launches between the chip-filling ones
 * (pivot, panel, restricted update, pivot, panel: latency-bound workgroups that end in half-empty rounds) for the two
 * halves of the batch on two streams (the caller's and an internal one, forked from and joined to it with events
 * around each run); the big launches -- covariance, the full updates, the gradient pass -- take the whole batch on the
 * caller's stream alone, so their per-launch durations (gpsr_profile_*) still measure one kernel.  Results are
 * bit-identical.  gpsr_set_split_small(0) (or GPSR_NO_SPLIT_SMALL=1 in the environment) turns it off; process-wide,
 * returns the previous setting.  No counterpart in the reference.
 */
int gpsr_set_split_small(int on);

/*
 * Opt-in, off by default (GPSR_FUSED=1 in the environment turns it on): the fused per-GP route of gpsr_fit
 * (k_fit_fused; no counterpart in the reference).  ONE launch runs the whole fit -- one workgroup per GP through
 * covariance, sweep, alpha, gradient statistics and the Adam step of every iteration, a persistent grid of two
 * workgroups per CU walking the batch, the matrix and panels in a per-workgroup-slot workspace (min(B, 2 x CUs) slots
 * instead of B) -- instead of ~650 launches over the whole batch.  Same role functions on the same data in the same
 * order: bit-identical to the launch-per-step route.  Used when on and d == 8, N >= 224.  Measured on MI355X it is
 * slower and moves MORE bytes through HBM than the default route, hence opt-in: its use is the
 * small workspace.  Process-wide, returns the
 * previous setting.  gpsr_fit_workspace_bytes / gpsr_predict_pixels_workspace_bytes follow the current setting (the
 * fused route needs min(B, 2 x CUs) matrix slots + 32 bytes per GP): size the workspace after setting it.
 */
int gpsr_set_fused(int on);

/*
 * One evaluation of the loss and of d loss / d raw (no Adam step), for tests:
 * what loss.backward() leaves in .grad at gprsr.py:185.   grad : [B][4] float32
 */
int gpsr_loss_grad(const double* X, const double* y, int B, int n, int d, int kernel_id,
                   const float* raw, double* loss, float* grad, int32_t* flags,
                   void* workspace, size_t workspace_bytes, void* stream);

/*
 * K5 predictive mean, fused (K(X*,X) is never materialised):
 * out[b][t] = c + outputscale * sum_i k(Xs[b][t], X[b][i]) alpha[b][i].
 * Replaces likelihood(model(test_x)).mean, gprsr.py:195.  d must be 8.
 *   alpha : [B][N] (N = gpsr_padded_n(n))   out : [B][m]
 */
int gpsr_predict(const double* X, const double* Xs, const double* alpha, const float* raw,
                 int B, int n, int m, int kernel_id, double* out, void* stream);

/*
 * K6 recombination.  Replaces the write-back gprsr.py:199-202,
 * PatchHandler.combine_patches (image_patches.py:44-76) and the clip gprsr.py:208.
 * Patches lie on the grid rows[nr] x cols[nc] (LR corners, row-major patch
 * order, duplicates allowed); pred holds their predicted interiors [nr*nc][m];
 * the 1-px rim of a patch is the test image itself and is only used where it
 * lies on the image border.  With test_in == NULL, pred instead holds whole
 * patches [nr*nc][P][P] (rim included), which is the shape combine_patches
 * itself is handed.  Additions happen in patch order, so the result is
 * bit-identical to the reference loop.   out : [H*s][W*s], clipped to [0,1]
 * when clip != 0 (gprsr.py:208; combine_patches itself does not clip).
 */
int gpsr_combine(const double* pred, const double* test_in, int H, int W, int s, int p,
                 const int32_t* rows, int nr, const int32_t* cols, int nc, int mode, int clip,
                 double* out, void* stream);

/*
 * Fused K1..K6: GPRSR.predict_pixels (gprsr.py:126-210) for one image in one call --
 * gather -> initial hyper-parameters -> `iters` Adam steps -> predictive mean ->
 * ordered recombination (+ clip to [0,1], gprsr.py:208).  Every patch is fitted from
 * the fresh initial values (warm_start "independent"; the serial chain of SURVEY.md F5
 * is composed from the separate entry points with `carry`).
 *   train_in, train_tgt : [H][W] f64     test_in : [H*s][W*s] f64
 *   rows[nr], cols[nc]  : LR patch-corner grid (PatchHandler.patch_locations is its
 *                         row-major product, duplicates kept), B = nr*nc patches
 *   tidx[nt]            : per-axis training indices, n = nt*nt
 *   chunk               : patches fitted per pass (1..B; 0 = all at once) -- bounds the workspace
 *   out : [H*s][W*s] f64        raw_out : [B][4] f32 final raw hyper-parameters, or NULL
 *   flags : [B] int32 (non-zero: a pivot was <= 0 at some iteration)
 * Workspace size: gpsr_predict_pixels_workspace_bytes (same s, p, nt, B, chunk).
 */
size_t gpsr_predict_pixels_workspace_bytes(int s, int p, int nt, int B, int chunk);
int gpsr_predict_pixels(const double* train_in, const double* train_tgt, const double* test_in,
                        int H, int W, int s, int p, const int32_t* rows, int nr,
                        const int32_t* cols, int nc, const int32_t* tidx, int nt,
                        int kernel_id, float init_raw_os, float init_raw_ls, int iters, float lr,
                        int mode, int chunk, double* out, float* raw_out, int32_t* flags,
                        void* workspace, size_t workspace_bytes, void* stream);

/*
 * K7 AGPR: the linear-kernel posterior mean is affine in the 7x7 window:
 * w[49] = outputscale * variance * X^T alpha.  gpsr_agpr_weights builds w from
 * a fitted model (agpr.py:58,176), gpsr_agpr_filter applies
 * out[i][j] = bias + sum w * window(img, i, j) to the interior of a [Hs][Ws]
 * image and copies the `pad`-px rim (agpr.py:165-185).  Windows are cast to
 * float32 first, as agpr.py:170 does.
 */
int gpsr_agpr_weights(const double* X, const double* alpha, const float* raw, int n, int d,
                      double* w, void* stream);
int gpsr_agpr_filter(const double* img, int Hs, int Ws, int L, const double* w, const float* raw,
                     double* out, void* stream);

/*
 * Image stages between the two GP phases, so the image never leaves the GPU
 * (SURVEY.md 8(f) rank 1).  OpenCV is restated, not linked:
 *   gpsr_u8_to_unit    img.astype(double)/255            gprsr.py:91, agpr.py:161
 *   gpsr_unit_to_u8    (img*scale).astype(uint8), truncating   gprsr.py:100
 *   gpsr_resample      cv.resize INTER_CUBIC (taps=4) / INTER_LINEAR (taps=2) on CV_64F:
 *                      horizontal then vertical pass with the index/coefficient
 *                      tables the host computes the way resize.cpp does
 *                      (gprsr.py:106,115; agpr.py:163,205,246)
 *   gpsr_decimate2     the INTER_AREA fast path cv.resize takes for fx=fy=1/2  (gprsr.py:115)
 *   gpsr_sepfilter_sym cv.GaussianBlur on CV_64F: symmetric separable filter,
 *                      BORDER_REFLECT_101 (gprsr.py:114; agpr.py:245)
 *   gpsr_filter3x3     cv.filter2D 3x3 correlation, BORDER_REFLECT_101 (agpr.py:206)
 *   gpsr_axpy          dst = a + alpha*b (agpr.py:207)
 */
int gpsr_u8_to_unit(const uint8_t* src, double* dst, size_t n, void* stream);
int gpsr_unit_to_u8(const double* src, uint8_t* dst, size_t n, double scale, void* stream);
int gpsr_resample(const double* src, int Hs, int Ws, double* dst, int Hd, int Wd,
                  const int32_t* xi, const double* xw, const int32_t* yi, const double* yw,
                  int taps, void* stream);
int gpsr_decimate2(const double* src, int Hs, int Ws, double* dst, void* stream);
int gpsr_sepfilter_sym(const double* src, int H, int W, double* dst, const double* kern, int ksize,
                       void* stream);
int gpsr_filter3x3(const double* src, int H, int W, double* dst, const double* kern, void* stream);
int gpsr_axpy(const double* a, const double* b, double alpha, double* dst, size_t n, void* stream);

/*
 * uint8 colour stages, so that a colour image goes to the GPU once.  OpenCV's
 * fixed-point paths restated in integer arithmetic (bit-exact by construction):
 *   gpsr_bgr2gray    cv.cvtColor(COLOR_BGR2GRAY): 15-bit table 3735/19235/9798     gprsr.py:53, evaluation.py:96
 *   gpsr_bgr2yuv     cv.cvtColor(COLOR_BGR2YUV)  + cv.split -> planes Y, U, V      gprsr.py:57-58
 *   gpsr_yuv2bgr     cv.merge + cv.cvtColor(COLOR_YUV2BGR)                         gprsr.py:62-63
 *   gpsr_bgr2ycrcb / gpsr_ycrcb2bgr   the same for YCrCb                           agpr.py:150-156
 *   gpsr_resample_u8 cv.resize(INTER_CUBIC) on a uint8 plane as the reference's (IPP-backed) OpenCV ran it: float32
 *                    taps xw, yw (host tables [n][4], not quantised), float32 passes, ties to even -- fitted to and
 *                    pinned by the reference's own uint8 planes (oracle/cvops.py)    gprsr.py:60-61, agpr.py:152-153, 246
 *   gpsr_sepfilter_u8 cv.GaussianBlur on a uint8 plane: OpenCV's bit-exact fixed-point path, 8-bit taps kern[ksize]
 *                    (sum 256, host: getGaussianKernelFixedPoint_ED), one rounding at the end, BORDER_REFLECT_101;
 *                    out of place                                                              agpr.py:245 via :286-287
 * bgr: [npx][3] interleaved (cv.imread layout); planes: [npx].
 */
int gpsr_bgr2gray(const uint8_t* bgr, uint8_t* gray, size_t npx, void* stream);
int gpsr_bgr2yuv(const uint8_t* bgr, uint8_t* Y, uint8_t* U, uint8_t* V, size_t npx, void* stream);
int gpsr_yuv2bgr(const uint8_t* Y, const uint8_t* U, const uint8_t* V, uint8_t* bgr, size_t npx, void* stream);
int gpsr_bgr2ycrcb(const uint8_t* bgr, uint8_t* Y, uint8_t* Cr, uint8_t* Cb, size_t npx, void* stream);
int gpsr_ycrcb2bgr(const uint8_t* Y, const uint8_t* Cr, const uint8_t* Cb, uint8_t* bgr, size_t npx, void* stream);
int gpsr_resample_u8(const uint8_t* src, int Hs, int Ws, uint8_t* dst, int Hd, int Wd,
                     const int32_t* xi, const float* xw, const int32_t* yi, const float* yw, void* stream);
int gpsr_sepfilter_u8(const uint8_t* src, int H, int W, uint8_t* dst, const int32_t* kern, int ksize, void* stream);

/*
 * Evaluator metrics (evaluation.py:83-98): two uint8 grey images [H][W], cropped by `crop`
 * px per side (preprocess_image, evaluation.py:95-98), skimage's definitions restated:
 *   out[0] = peak_signal_noise_ratio(data_range=255) in dB (+inf when identical)   evaluation.py:87
 *   out[1] = structural_similarity(data_range=255): 7x7 uniform window, sample covariance, K1 .01, K2 .03,
 *            mean over the windows inside the image                                evaluation.py:93
 *   out[2] = mean squared error
 * out: device double[3].  Deterministic (integer window sums, fixed-order reductions).
 */
size_t gpsr_metrics_workspace_bytes(int H, int W, int crop);
int gpsr_metrics(const uint8_t* a, const uint8_t* b, int H, int W, int crop, double* out,
                 void* workspace, size_t workspace_bytes, void* stream);

/*
 * AGPR training-set selection ("active sampling", agpr.py:120-137 with get_characteristic_score / get_diversity
 * :211-222).  data: [N][D] f64 candidate rows (49 bicubic pixels + the HR centre pixel,
 * np.unique'd, agpr.py:113-118).  Three launches: the [N][N] squared distances (:123-124), the representativeness
 * score of every row from its K nearest neighbours (:126-128), and the greedy loop that picks n_pick rows by
 * argmax of w_rep * representativeness + w_div * diversity (:132-137; w_rep = CHARACTERISTIC_TRADEOFF, w_div =
 * 1 - CHARACTERISTIC_TRADEOFF as the host evaluates it).  bandwidth is agpr.py:125's value, which the host computes
 * (it is a median of the INDICES np.where returns: 0.2 * (N - 1) / 2 for finite data).  Kept from the reference
 * because they decide the choice: the score array is indexed by the position in the shrinking pool, sums run in
 * numpy's order (distances bit-identical to numpy's), np.argmax's first-maximum rule.
 *   out_idx : [n_pick] int32, indices into data in the order chosen     rep_out : [N] scores, or NULL
 * D <= 128, K <= 64, N <= 46340.  Workspace: gpsr_active_sample_workspace_bytes(N) (N = 5000: 200 MB).
 */
size_t gpsr_active_sample_workspace_bytes(int N);
int gpsr_active_sample(const double* data, int N, int D, int K, int n_pick, double bandwidth, double w_rep,
                       double w_div, int32_t* out_idx, double* rep_out, void* workspace, size_t workspace_bytes,
                       void* stream);

/*
 * Kernel timing for bench.py's roofline line: HIP-event brackets around the
 * tagged launches, on the stream they are launched on.  No counterpart in the
 * reference (its only instrumentation is time.time() per image, gprsr.py:49-72).
 * gpsr_profile_end waits for the recorded events and fills counts[] / total_ms[]
 * (GPSR_PROFILE_NTAGS entries each).  Not thread-safe; one profiling session at a time.
 */
enum {
  GPSR_PROF_UPDATE = 0, GPSR_PROF_PANEL, GPSR_PROF_PIVOT, GPSR_PROF_BUILD,
  GPSR_PROF_GRAD1, GPSR_PROF_GRAD2, GPSR_PROF_PREDICT,
  GPSR_PROF_UPDATE_LAST, /* k_update<true>: the last pivot step's launch, which also leaves the alpha partials */
  GPSR_PROF_UPDATE2,     /* k_update2: both rank-112 updates of a two-pivot step in one pass over A */
  GPSR_PROF_UPDATE_SEL,  /* k_update restricted to the next pivot's block row/column (first half of a two-pivot step) */
  GPSR_PROF_FUSED,       /* k_fit_fused: a whole fit (<= 128 Adam iterations + the final evaluation) of the batch */
  GPSR_PROFILE_NTAGS
};
int gpsr_profile_select(unsigned tag_mask);   /* bit t set: launches with tag t are bracketed (default: all tags) */
int gpsr_profile_begin(int max_records);
int gpsr_profile_end(int32_t* counts, double* total_ms);

/* Self-test of the f64 MFMA operand/accumulator lane maps: D = A(16x4) B(4x16),
 * plain row-major inputs and output.  Used by tests only. */
int gpsr_selftest_mfma(const double* A, const double* Bm, double* D, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GPSR_H */
