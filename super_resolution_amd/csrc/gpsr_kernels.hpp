// gpsr_kernels.hpp -- device code of the batched GP super-resolution engine (gfx950 only).
//
// Layout conventions
//   A      [B][nblk][nblk][NB][NB] f64, N = nblk*NB: TILE-BLOCKED -- tile (I,J) of a matrix is one contiguous
//          row-major [NB][NB] block (100 KB), so the 16-row strip a wave reads / writes and every LDS-DMA chunk
//          is one contiguous range instead of 16 row segments N*8 bytes apart (round 1 measured -6.5 % on the
//          update kernels for contiguous C accesses).  Only tiles (I,J) with I>=J are maintained ("lower
//          block-triangle"); diagonal tiles are kept full.
//   Wo     [B][nblk][NB][NB]  k-major panel of the current pivot step, W = A_.k L^-T, one contiguous
//          [kk][r] block per row block I: Wo[I][kk][r] = W[I*NB + r][kk]
//   U, M   [B][NB][NB] U = L^-T (upper), M = L^-1 (lower) of the current pivot block A_kk = L L^T
//   raw    [B][4] f32  raw hyper-parameters (noise, const, outputscale, lengthscale)
//
// Inversion = blocked symmetric Gauss-Jordan sweep (Goodnight's SWP operator) in
// Cholesky form:  for each pivot block k, with A_kk = L L^T:
//   W_i = A_ik L^-T (i != k) ; A_ij <- A_ij - W_i W_j^T (i,j != k) ;
//   A_ik <- W_i L^-1 ; A_kk <- -(L L^T)^-1
// after all pivots A = -A^-1 and sum log(pivots) = log det A.  Total work ~n^3
// flops, all of it in 112x112x112 f64 MFMA tile products.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NB 112          // pivot block == tile edge
#define NBT 7           // NB / 16 : MFMA tiles per edge == waves per workgroup
#define MM_THREADS 448  // 64 * NBT
#define KC 16           // k-chunk staged through LDS
#define TILE_OFF(I, J, nblk) ((size_t)((I) * (nblk) + (J)) * (size_t)(NB * NB))   // tile (I,J) inside one matrix

typedef double v4d __attribute__((ext_vector_type(4)));

enum { K_RBF = 0, K_M12 = 1, K_M32 = 2, K_M52 = 3, K_LIN = 4 };

struct Hyp { double noise, mean, os, ls, il, il2; };   // il = 1/ls, il2 = 1/ls^2 (once per thread: the pair loops multiply)

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ Hyp load_hyp(const float* __restrict__ raw) {
  Hyp h;
  h.noise = (double)(softplus_f(raw[0]) + 1e-4f);
  h.mean = (double)raw[1];
  h.os = (double)softplus_f(raw[2]);
  h.ls = (double)softplus_f(raw[3]);
  h.il = 1.0 / h.ls;
  h.il2 = 1.0 / (h.ls * h.ls);
  return h;
}

// The same, computed by ONE wave of the workgroup and handed to the others through LDS (hs: 6 doubles; the caller's
// next __syncthreads() publishes it, load_hyp_shared_get reads it after that barrier).  The constraint transforms are
// ~450 instructions (three softplus: expf, log1pf; two divisions) against ~2200 in the pair loop of a k_build8 /
// k_grad2_8 thread: done by every wave they were a fifth of those launches.
__device__ __forceinline__ void load_hyp_shared_put(const float* __restrict__ raw, double* __restrict__ hs, int tid) {
  if (tid < 64) {
    const Hyp h = load_hyp(raw);
    if (tid == 0) { hs[0] = h.noise; hs[1] = h.mean; hs[2] = h.os; hs[3] = h.ls; hs[4] = h.il; hs[5] = h.il2; }
  }
}
__device__ __forceinline__ Hyp load_hyp_shared_get(const double* __restrict__ hs) {
  Hyp h;
  h.noise = hs[0]; h.mean = hs[1]; h.os = hs[2]; h.ls = hs[3]; h.il = hs[4]; h.il2 = hs[5];
  return h;
}

// ---- lean f64 sqrt / exp for the pair loops (k_build8, k_grad2_8, k_predict are bound by the f64 VALU issue
// rate; libm's versions spend a third of their instructions on range scaling and special cases that cannot occur
// here, and the compiler turns every Horner step on a constant into v_mov_b64 + v_fmac_f64).
// d = a*b + c as one VOP3 instruction, c (a constant) read from an SGPR pair.
__device__ __forceinline__ double fma_sc(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c));
  return d;
}

// sqrt(x) for x in [1e-30, 1e300]: v_rsq_f64 seed (2^-23 relative), one coupled Newton step (2^-46) and one
// residual correction (< 1 ulp); libm's exponent scaling and its second correction (exact rounding) are left out.
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  return fma(d, h, g);
}

// exp(-s) for s >= 0 (anything above 800 gives 0): n = rint(-s log2 e), r = -s - n ln2 in two pieces,
// degree-13 Taylor polynomial on |r| <= ln2/2 (truncation 4e-18 relative), ldexp.  < 1 ulp.
__device__ __forceinline__ double exp_neg(double s) {
  s = fmin(s, 800.0);
  const double n = __builtin_rint(s * -1.4426950408889634);
  double r = fma(n, -6.93147180369123816490e-01, -s);
  r = fma(n, -1.90821492927058770002e-10, r);
  double p = fma_sc(r, 1.6059043836821613e-10, 2.08767569878681e-09);   // 1/13!, 1/12!
  p = fma_sc(p, r, 2.505210838544172e-08);    // 1/11!
  p = fma_sc(p, r, 2.755731922398589e-07);    // 1/10!
  p = fma_sc(p, r, 2.7557319223985893e-06);   // 1/9!
  p = fma_sc(p, r, 2.48015873015873e-05);     // 1/8!
  p = fma_sc(p, r, 1.984126984126984e-04);    // 1/7!
  p = fma_sc(p, r, 1.388888888888889e-03);    // 1/6!
  p = fma_sc(p, r, 8.333333333333333e-03);    // 1/5!
  p = fma_sc(p, r, 4.1666666666666664e-02);   // 1/4!
  p = fma_sc(p, r, 1.6666666666666666e-01);   // 1/3!
  p = fma_sc(p, r, 0.5);
  p = fma_sc(p, r, 1.0);
  p = fma_sc(p, r, 1.0);
  return ldexp(p, (int)n);
}

// base kernel value from the pair statistic (squared distance, or dot product for linear)
template <int KID>
__device__ __forceinline__ double kern_val(double st, const Hyp& h) {
  if (KID == K_LIN) return h.ls * st;
  if (KID == K_RBF) return exp_neg(0.5 * st * h.il2);
  double r = sqrt_pos(fmin(fmax(st * h.il2, 1e-30), 1e300));
  if (KID == K_M12) return exp_neg(r);
  if (KID == K_M32) { double s = 1.7320508075688772 * r; return (s + 1.0) * exp_neg(s); }
  double s = 2.23606797749979 * r;
  return (s + 1.0 + (5.0 / 3.0) * r * r) * exp_neg(s);
}

// value and d/d(lengthscale) (d/d(variance) for linear)
template <int KID>
__device__ __forceinline__ void kern_val_dl(double st, const Hyp& h, double& k, double& dk) {
  if (KID == K_LIN) { k = h.ls * st; dk = st; return; }
  if (KID == K_RBF) { k = exp_neg(0.5 * st * h.il2); dk = k * st * h.il2 * h.il; return; }
  double r = sqrt_pos(fmin(fmax(st * h.il2, 1e-30), 1e300));
  if (KID == K_M12) { k = exp_neg(r); dk = k * r * h.il; return; }
  if (KID == K_M32) {
    double s = 1.7320508075688772 * r, e = exp_neg(s);
    k = (s + 1.0) * e; dk = 3.0 * r * r * e * h.il; return;
  }
  double s = 2.23606797749979 * r, e = exp_neg(s);
  k = (s + 1.0 + (5.0 / 3.0) * r * r) * e;
  dk = (5.0 / 3.0) * r * r * (1.0 + s) * e * h.il;
}

template <int KID>
__device__ __forceinline__ double pair_stat(const double* __restrict__ a, const double* __restrict__ b, int d) {
  double s = 0.0;
  if (KID == K_LIN) {
    for (int f = 0; f < d; ++f) s = fma(a[f], b[f], s);
  } else {
    for (int f = 0; f < d; ++f) { double t = a[f] - b[f]; s = fma(t, t, s); }
  }
  return s;
}

// lower-triangular tile enumeration: t -> (I, J), I >= J
__device__ __forceinline__ void tri_decode(int t, int& I, int& J) {
  int i = 0;
  while ((i + 1) * (i + 2) / 2 <= t) ++i;
  I = i; J = t - i * (i + 1) / 2;
}

// XCD-aware work-item remap: blocks b and b+8 share an XCD (speed only); give each
// XCD a contiguous range of the virtual index so the tiles of one GP share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int total) {
  int q = total >> 3, r = total & 7, x = bid & 7, s = bid >> 3;
  int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
  return base + s;
}

// locs[b] = (rows[b / nc], cols[b % nc]): PatchHandler.patch_locations order (image_patches.py:31-41)
__global__ void k_make_locs(const int* __restrict__ rows, const int* __restrict__ cols, int nr, int nc,
                            int* __restrict__ locs) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nr * nc) return;
  locs[2 * b] = rows[b / nc];
  locs[2 * b + 1] = cols[b % nc];
}

// ---------------------------------------------------------------- K1 gather
// One 16-byte store per lane: a work unit is a PAIR of neighbour features of one point (the
// reference's order gprsr.py:35 is four pairs of opposite neighbours: (+1,0)/(-1,0), (0,+1)/(0,-1),
// (+1,+1)/(-1,-1), (+1,-1)/(-1,+1)), or one training target.  Units per patch: 4n + n + 4m.
__global__ void k_gather(const double* __restrict__ tin, const double* __restrict__ ttg,
                         const double* __restrict__ tst, int H, int W, int s, int p,
                         const int32_t* __restrict__ locs, const int32_t* __restrict__ tidx, int nt,
                         double* __restrict__ X, double* __restrict__ y, double* __restrict__ Xs) {
  const int b = blockIdx.y;
  const int r0 = locs[2 * b], c0 = locs[2 * b + 1];
  const int n = nt * nt, P = p * s, ms = P - 2, m = ms * ms, Ws = W * s;
  const int total = 5 * n + 4 * m;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < total; u += gridDim.x * blockDim.x) {
    if (u >= 4 * n && u < 5 * n) {
      const int q = u - 4 * n;
      const int yy = tidx[q / nt], xx = tidx[q % nt];
      y[(size_t)b * n + q] = ttg[(size_t)(r0 + yy) * W + c0 + xx];
      continue;
    }
    const bool test = u >= 5 * n;
    const int v = test ? u - 5 * n : u;
    const int q = v >> 2, h = v & 3;
    const int dyA = (h != 1) ? 1 : 0, dxA = (h == 0) ? 0 : (h == 3 ? -1 : 1);
    if (!test) {
      const int yy = tidx[q / nt], xx = tidx[q % nt];
      const double* c = tin + (size_t)(r0 + yy) * W + c0 + xx;
      const int o = dyA * W + dxA;
      *(double2*)(X + (size_t)b * n * 8 + 2 * (size_t)v) = make_double2(c[o], c[-o]);
    } else {
      const int yy = 1 + q / ms, xx = 1 + q % ms;
      const double* c = tst + (size_t)(r0 * s + yy) * Ws + c0 * s + xx;
      const int o = dyA * Ws + dxA;
      *(double2*)(Xs + (size_t)b * m * 8 + 2 * (size_t)v) = make_double2(c[o], c[-o]);
    }
  }
}

// ------------------------------------------------------- initial hypers
__global__ void k_init_hypers(const double* __restrict__ y, int n, int kid, float raw_noise0,
                              float raw_ls_rbf, float init_os, float init_ls,
                              const float* __restrict__ carry, int carry_stride, float* __restrict__ raw) {
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* yb = y + (size_t)b * n;
  double s = 0.0;
  for (int i = tid; i < n; i += 256) s += (double)(float)yb[i];
  red[tid] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const double mean = red[0] / n;
  __syncthreads();
  double v = 0.0;
  for (int i = tid; i < n; i += 256) { double t = (double)(float)yb[i] - mean; v += t * t; }
  red[tid] = v; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) {
    float* r = raw + 4 * b;
    r[0] = raw_noise0;
    r[1] = (float)mean;
    if (kid == K_RBF) {
      float var = (float)(red[0] / (n - 1));
      r[2] = var + logf(-expm1f(-var));  // inv_softplus
      r[3] = raw_ls_rbf;
    } else if (carry) {
      r[2] = carry[(size_t)b * carry_stride + 2]; r[3] = carry[(size_t)b * carry_stride + 3];
    } else {
      r[2] = init_os; r[3] = init_ls;
    }
  }
}

// ---------------------------------------------------------- K2 covariance
// grid.x = ntri * B (xcd-remapped), 256 threads, dynamic LDS = 2*NB*(d+1) doubles
template <int KID>
__global__ void k_build(const double* __restrict__ X, const float* __restrict__ raw,
                        double* __restrict__ A, int n, int N, int d, int ntri, int total,
                        double* __restrict__ logdet) {
  extern __shared__ double sm[];
  const int v = xcd_remap(blockIdx.x, total);
  const int b = v / ntri, t = v % ntri;
  int I, J; tri_decode(t, I, J);
  const int ldx = d + 1;
  double* xi = sm; double* xj = sm + NB * ldx;
  const double* Xb = X + (size_t)b * n * d;
  for (int e = threadIdx.x; e < NB * d; e += blockDim.x) {
    int r = e / d, f = e % d;
    int gi = I * NB + r, gj = J * NB + r;
    xi[r * ldx + f] = gi < n ? Xb[(size_t)gi * d + f] : 0.0;
    xj[r * ldx + f] = gj < n ? Xb[(size_t)gj * d + f] : 0.0;
  }
  const Hyp h = load_hyp(raw + 4 * b);
  if (t == 0 && threadIdx.x == 0 && logdet) logdet[b] = 0.0;
  __syncthreads();
  double* Ab = A + (size_t)b * N * N;
  for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
    int i = e / NB, j = e % NB;
    int gi = I * NB + i, gj = J * NB + j;
    double val;
    if (gi < n && gj < n) {
      double st = pair_stat<KID>(xi + i * ldx, xj + j * ldx, d);
      val = h.os * kern_val<KID>(st, h);
      if (gi == gj) val += h.noise;
    } else {
      val = (gi == gj) ? 1.0 : 0.0;
    }
    Ab[TILE_OFF(I, J, N / NB) + (size_t)i * NB + j] = val;
  }
}

// Covariance and its derivative with respect to the lengthscale (variance for the linear kernel), for
// callers and tests of the gradient: K as k_build writes it (may be null), dK[b] = outputscale * dk/dl in
// the same lower block-triangle layout, zero in the identity padding.  Same launch shape as k_build.
template <int KID>
__global__ void k_build_dl(const double* __restrict__ X, const float* __restrict__ raw, double* __restrict__ A,
                           double* __restrict__ dA, int n, int N, int d, int ntri, int total) {
  extern __shared__ double sm[];
  const int v = xcd_remap(blockIdx.x, total);
  const int b = v / ntri, t = v % ntri;
  int I, J; tri_decode(t, I, J);
  const int ldx = d + 1;
  double* xi = sm; double* xj = sm + NB * ldx;
  const double* Xb = X + (size_t)b * n * d;
  for (int e = threadIdx.x; e < NB * d; e += blockDim.x) {
    int r = e / d, f = e % d;
    int gi = I * NB + r, gj = J * NB + r;
    xi[r * ldx + f] = gi < n ? Xb[(size_t)gi * d + f] : 0.0;
    xj[r * ldx + f] = gj < n ? Xb[(size_t)gj * d + f] : 0.0;
  }
  const Hyp h = load_hyp(raw + 4 * b);
  __syncthreads();
  for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
    int i = e / NB, j = e % NB;
    int gi = I * NB + i, gj = J * NB + j;
    double kv = (gi == gj) ? 1.0 : 0.0, dk = 0.0;
    if (gi < n && gj < n) {
      const double st = pair_stat<KID>(xi + i * ldx, xj + j * ldx, d);
      kern_val_dl<KID>(st, h, kv, dk);
      kv = h.os * kv + (gi == gj ? h.noise : 0.0);
      dk = h.os * dk;
    }
    const size_t off = (size_t)b * N * N + TILE_OFF(I, J, N / NB) + (size_t)i * NB + j;
    if (A) A[off] = kv;
    dA[off] = dk;
  }
}

// d == 8 fast path (the SRGPR feature vector): 448 threads; thread (g, j) = (tid / NB, tid % NB)
// keeps column point x_j in registers and walks rows i = g, g+4, ...; the row point x_i
// is an LDS broadcast (one address per wave, up to the g boundary).  Same per-element
// arithmetic as k_build, element order only.
// Role form (one tile by one 448-thread workgroup; LDS handed in) + the launch wrapper: the fused per-GP fit kernel
// (k_fit_fused) runs the same role bodies one after the other, so both routes execute the same instructions.
template <int KID>
__device__ __forceinline__ void role_build8(const double* __restrict__ Xb, const float* __restrict__ rawb,
                                            double* __restrict__ Agp, int n, int N, int I, int J, int full_diag,
                                            double* __restrict__ xi, double* __restrict__ hs, int tid) {
  // Diagonal tiles inside the fit: the sweep reads only the 16x16 sub-blocks on and below the diagonal
  // (k_pivot, the SKIP_DIAG updates), so only those 28 of 49 are evaluated: 16 threads per sub-block, one
  // column each.  gpsr_covariance (full_diag) keeps whole diagonal tiles.
  const bool sub = (I == J) && !full_diag;
  int g, j, i0 = 0, istep = 4, iend = NB;
  if (!sub) { g = tid / NB; j = tid - g * NB; i0 = g; }
  else {
    int bi, bj; tri_decode(tid >> 4, bi, bj);
    j = 16 * bj + (tid & 15); i0 = 16 * bi; istep = 1; iend = i0 + 16; g = 0;
  }
  const int gj = J * NB + j;
  double xj[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) xj[f] = gj < n ? Xb[(size_t)gj * 8 + f] : 0.0;
  for (int e = tid; e < NB * 8; e += MM_THREADS) {
    const int gi = I * NB + (e >> 3);
    xi[e] = gi < n ? Xb[(size_t)I * NB * 8 + e] : 0.0;
  }
  load_hyp_shared_put(rawb, hs, tid);
  __syncthreads();
  const Hyp h = load_hyp_shared_get(hs);
  double* Ab = Agp + TILE_OFF(I, J, N / NB) + j;
  // Rows of real points first, padding rows (identity) after; an off-diagonal tile has no diagonal element, so its
  // loop carries neither the noise term nor any per-element test (same arithmetic per element; the tests and the
  // select cost a fifth of the instructions of the loop).
  const int ireal = min(iend, n - I * NB);            // rows i < ireal are real points (may be <= i0)
  if (gj < n) {
    if (I != J) {
#pragma unroll 2
      for (int i = i0; i < ireal; i += istep) {
        const double st = pair_stat<KID>(xi + i * 8, xj, 8);
        Ab[(size_t)i * NB] = h.os * kern_val<KID>(st, h);
      }
    } else {
#pragma unroll 2
      for (int i = i0; i < ireal; i += istep) {
        const double st = pair_stat<KID>(xi + i * 8, xj, 8);
        double val = h.os * kern_val<KID>(st, h);
        if (i == j) val += h.noise;
        Ab[(size_t)i * NB] = val;
      }
    }
  }
  int ipad = i0;                                       // first row of this thread's walk that is padding
  if (gj < n && ireal > i0) ipad = i0 + ((ireal - i0 + istep - 1) / istep) * istep;
  for (int i = ipad; i < iend; i += istep) Ab[(size_t)i * NB] = (I == J && i == j) ? 1.0 : 0.0;
}

template <int KID>
__global__ void __launch_bounds__(MM_THREADS)
k_build8(const double* __restrict__ X, const float* __restrict__ raw, double* __restrict__ A, int n, int N,
         int ntri, int total, double* __restrict__ logdet, int full_diag) {
  __shared__ __attribute__((aligned(16))) double xi[NB * 8];
  __shared__ double hs[6];
  const int v = xcd_remap(blockIdx.x, total);
  const int b = v / ntri, t = v % ntri;
  int I, J; tri_decode(t, I, J);
  const int tid = threadIdx.x;
  if (t == 0 && tid == 0 && logdet) logdet[b] = 0.0;
  role_build8<KID>(X + (size_t)b * n * 8, raw + 4 * b, A + (size_t)b * N * N, n, N, I, J, full_diag, xi, hs, tid);
}

// ------------------------------------------------------------ pivot block
// One workgroup (7 waves) per GP factors the NBxNB pivot block S = A_kk in LDS:
//   phase 1  blocked Cholesky S = L L^T (16x16 sub-blocks; the wave that owns diagonal sub-block q+1 updates it
//            with panel column q and factors + inverts it in registers at once (look-ahead, chol16_inv), publishing
//            inv(L16) in LDS for the next sub-panel product, while the others finish the trailing update of step q)
//   phase 2  M = L^-1 by block forward substitution down the columns, row q computed DURING step q of phase 1 by the
//            waves that have run out of factorisation work (in place of L); only the last row is left afterwards
//   phase 3  P = M^T M = U U^T  ->  A_kk <- -P   (the pivot block of -A^-1; with_p, else left to the update launch)
//   U and M go to global for the panel / update kernels block by block, as each becomes final.
// Everything outside the 16x16 register kernels is f64 MFMA.  Cholesky form on both
// levels (W = T L^-T, A -= W W^T, T <- W L^-1) keeps the error at the level of an
// unblocked sweep; forming A -= (T P) T^T with an explicit block inverse loses
// sqrt(cond) digits per level (measured: DESIGN.md "Numerics of the inversion").
#define PV_DLD 17

// 1/sqrt(x), x > 0: v_rsq_f64 seed + one third-order correction (relative error ~ e^3 of the seed's e <= 2^-23)
__device__ __forceinline__ double rsqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);
  return fma(y * e, fma(0.375, e, 0.5), y);
}

// broadcast of one accumulator element to the whole wave through scalar registers (lane index is a compile-time constant)
__device__ __forceinline__ double bcast_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// 16x16 SPD block in registers, accumulator lane layout (lr = lane&15, lk = lane>>4; a[c] = D[lk+4c][lr]):
// m <- inv(chol(D)) in the same layout, *ld += log det D, *bad on a non-positive pivot.  Four columns per step:
// the k-depth of v_mfma_f64_16x16x4_f64 is 4, so the rank-4 elimination of the block is ONE instruction and no
// cross-lane shuffle is needed.  Step J (columns 4J..4J+3),
// D = current Schur complement (symmetric, accumulator layout a[c](lr,lk) = D[lk+4c][lr]), M accumulates L^-1:
//     L4 = chol(D_JJ), M4 = L4^-1        4x4 on ten values broadcast through scalar registers, every lane the same
//     P  = D[:, 4J..4J+3] M4^T           (16 x 4)  one MFMA: A operand = M4 (lanes lr < 4), B operand = a[J], which
//                                        by symmetry of D is D[:, 4J..4J+3] in B-operand layout; P comes out in
//                                        A/B-operand layout (lane (lr,lk) = P[lr][lk])
//     R  = M4 M[4J..4J+3, :]             (4 x 16)  one MFMA: A = M4, B = m[J]; these are the final rows of M
//     D -= P P^T ;  M[rows > 4J+3, :] -= P R                                     one MFMA each
// This is the serial chain of k_pivot (7 calls, 65 % of the workgroup's time), and it is bound by the NUMBER of f64
// instructions of its one wave, not by their dependences (tools/chol16_bench.hip + tools/valu_lat.hip: a lone wave
// issues one f64 VALU instruction per ~7.5 cycles dependent or not, v_rsq_f64 ~17, an MFMA 64, and MFMA and f64 VALU
// share the pipe): per step ~75 f64 instructions + 4 rsq + 20 readlane + 20 cndmask + 4 MFMA = ~1000 cycles.  Hence
// the 2x2-minor form below (fewer instructions than four sequential pivots), selects instead of a branch tree for the
// A operand, one log per block: 4896 -> 4032 cycles per call.  Forcing the Schur update ahead of the R product
// (sched_barrier, or a false dependence) made it slower: 4312 / 4448.
__device__ __forceinline__ void chol16_inv(double a[4], double m[4], int lane, double* ld, int* bad) {
  const int lr = lane & 15, lk = lane >> 4;
  v4d D = (v4d){a[0], a[1], a[2], a[3]};
  v4d M;
#pragma unroll
  for (int c = 0; c < 4; ++c) M[c] = (lk + 4 * c == lr) ? 1.0 : 0.0;
  double prod = 1.0;
  const v4d zero4 = (v4d){0.0, 0.0, 0.0, 0.0};
  // the ten lanes (lr + 16 lk, lk <= lr < 4) that carry M4 as the A operand of the P / R products
  const bool L00 = lane == 0, L10 = lane == 1, L20 = lane == 2, L30 = lane == 3, L11 = lane == 17, L21 = lane == 18,
             L31 = lane == 19, L22 = lane == 34, L32 = lane == 35, L33 = lane == 51;
#pragma unroll
  for (int J = 0; J < 4; ++J) {
    // D[4J+r][4J+c] sits in D[J] at lane (lr = 4J+c, lk = r)
    const double dj = D[J];
#define DJ(r, c) bcast_lane(dj, 4 * J + (c) + 16 * (r))
    const double d00 = DJ(0, 0), d10 = DJ(1, 0), d11 = DJ(1, 1), d20 = DJ(2, 0), d21 = DJ(2, 1), d22 = DJ(2, 2),
                 d30 = DJ(3, 0), d31 = DJ(3, 1), d32 = DJ(3, 2), d33 = DJ(3, 3);
#undef DJ
    // The 4x4 block as two 2x2 blocks.  Second pivot of a 2x2 block [[x, y], [y, z]]: p = z - y^2/x = q/x with the minor
    // q = x z - y^2, so 1/sqrt(p) = sqrt(x) / sqrt(q): the two v_rsq_f64 + correction chains run side by side instead of
    // one after the other (the rounding error of q is eps (xz + y^2), i.e. eps (z + y^2/x) on p: that of the usual form).
    const double q1 = fma(-d10, d10, d00 * d11);
    const double r0 = rsqrt_pos(d00), rq1 = rsqrt_pos(q1);
    const double r1 = rq1 * (d00 * r0);
    const double l10 = d10 * r0, l20 = d20 * r0, l30 = d30 * r0;
    const double l21 = fma(-l20, l10, d21) * r1, l31 = fma(-l30, l10, d31) * r1;
    const double e22 = fma(-l21, l21, fma(-l20, l20, d22));
    const double e32 = fma(-l31, l21, fma(-l30, l20, d32));
    const double e33 = fma(-l31, l31, fma(-l30, l30, d33));
    const double q3 = fma(-e32, e32, e22 * e33);
    const double r2 = rsqrt_pos(e22), rq3 = rsqrt_pos(q3);
    const double r3 = rq3 * (e22 * r2);
    const double l32 = e32 * r2;
    if (!(d00 > 0.0) || !(q1 > 0.0) || !(e22 > 0.0) || !(q3 > 0.0)) *bad = 1;
    prod *= q1 * q3;                    // = the four pivots d00 (q1/d00) e22 (q3/e22); 16 pivots in [1e-4, 1e3]: one log per block
    const double m10 = -(l10 * r0) * r1;
    const double m21 = -(l21 * r1) * r2;
    const double m32 = -(l32 * r2) * r3;
    const double m20 = -fma(l21, m10, l20 * r0) * r2;
    const double m31 = -fma(l32, m21, l31 * r1) * r3;
    const double m30 = -fma(l32, m20, fma(l31, m10, l30 * r0)) * r3;
    // A operand of the P / R products: lane (lr, lk) holds M4[lr][lk] for lr < 4, zero elsewhere (selects, no branches)
    double m4a = L00 ? r0 : 0.0;
    m4a = L10 ? m10 : m4a; m4a = L11 ? r1 : m4a;
    m4a = L20 ? m20 : m4a; m4a = L21 ? m21 : m4a; m4a = L22 ? r2 : m4a;
    m4a = L30 ? m30 : m4a; m4a = L31 ? m31 : m4a; m4a = L32 ? m32 : m4a; m4a = L33 ? r3 : m4a;
    const v4d po = __builtin_amdgcn_mfma_f64_16x16x4f64(m4a, D[J], zero4, 0, 0, 0);
    const double P = (lr < 4 * J) ? 0.0 : po[0];          // rows of blocks already eliminated contribute nothing
    const double Pm = (lr < 4 * J + 4) ? 0.0 : po[0];     // only rows below the block are eliminated in M
    if (J < 3) D = __builtin_amdgcn_mfma_f64_16x16x4f64(-P, P, D, 0, 0, 0);
    const v4d ro = __builtin_amdgcn_mfma_f64_16x16x4f64(m4a, M[J], zero4, 0, 0, 0);
    const double R = ro[0];
    M = __builtin_amdgcn_mfma_f64_16x16x4f64(-Pm, R, M, 0, 0, 0);
    M[J] = R;                                             // rows 4J..4J+3 of M are final
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) m[c] = M[c];
  *ld += log(prod);
}

// Block-packed LDS image of the pivot block: the 28 lower 16x16 blocks, each [16][17];
// (i,j), i >= j, at block index i(i+1)/2 + j.  60.9 KB + 15.2 KB of per-wave parking blocks = 76 KB -> two workgroups per CU (the 16x16 register factorisations are a serial,
// latency-bound chain: a second resident workgroup fills the bubbles).
#define PV_BLK (16 * PV_DLD)
#define PV_NBLK (NBT * (NBT + 1) / 2)
#define SB(i, j) (S + (((i) * ((i) + 1)) / 2 + (j)) * PV_BLK)

#ifdef GPSR_PIV_TRACE
// Debug build only (tools/piv_trace.py): cycles of a k_pivot workgroup by phase, summed over workgroups (stamped by
// thread 0 right after the barrier that ends each phase): [0] workgroups, [1] load, [2] blocked Cholesky,
// [3] M = L^-1, [4] P = M^T M + stores of A_kk, [5] U / M stores, [6] entry -> exit.
__device__ unsigned long long g_piv_trace[8];
// g_piv_trace2: the look-ahead (critical) wave q+1 inside step q of phase 1, summed over steps and workgroups:
// [0] steps, [1] sub-panel product, [2] wait at the barrier after it, [3] diagonal block update, [4] chol16_inv + publish,
// [5] wait at the step's last barrier, [6] wave 6's trailing update (steps where it is not the look-ahead wave), [7] its count
__device__ unsigned long long g_piv_trace2[8];
#define PIV2_T() __builtin_readcyclecounter()
#define PIV2_ADD(i, v) do { if (lane == 0) atomicAdd(&g_piv_trace2[i], (unsigned long long)(v)); } while (0)
#define PIV_STAMP(i) do { if (tid == 0) { const unsigned long long t_ = __builtin_readcyclecounter(); atomicAdd(&g_piv_trace[i], t_ - piv_t); piv_t = t_; } } while (0)
#else
#define PIV_STAMP(i)
#endif

// Row block r of column block c of M = L^-1 (forward substitution down the column, c < r):
//   X[r] = -inv(L16_r) * sum_{t=c..r-1} L[r,t] X[t],   X[c] = inv(L16_c),
// from the LDS image S: L[r,t] at SB(r,t) (t < r), inv(L16_r) at SB(r,r), the finished X[t] of this column at SB(t,c)
// (in place of L[t,c]).  The sum comes out in accumulator layout (reg cc of lane (lr,lk) = [lk+4cc][lr]), which IS the
// MFMA B-operand layout once the k index is read as lk+4cc instead of 4ks+lk; the A operands are indexed the same way:
// no transposition between the two products.
#define SB_(S, i, j) ((S) + (((i) * ((i) + 1)) / 2 + (j)) * (16 * PV_DLD))
__device__ __forceinline__ v4d minv_col_step(const double* __restrict__ S, int r, int c, int lr, int lk) {
  v4d sum = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int t = c; t < r; ++t) {
    const double* Lrt = SB_(S, r, t);
    const double* Xtc = SB_(S, t, c);
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
      sum = __builtin_amdgcn_mfma_f64_16x16x4f64(Lrt[lr * PV_DLD + lk + 4 * cc], Xtc[(lk + 4 * cc) * PV_DLD + lr], sum, 0, 0, 0);
  }
  const double* Mrr = SB_(S, r, r);
  v4d xr = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int cc = 0; cc < 4; ++cc)
    xr = __builtin_amdgcn_mfma_f64_16x16x4f64(-Mrr[lr * PV_DLD + lk + 4 * cc], sum[cc], xr, 0, 0, 0);
  return xr;
}

// Which column block of row q of M a wave computes during step q of the factorisation (-1: none).  Row q has q blocks
// (columns 0..q-1, block (q,c) costs q-c+1 products), waves 0..q are out of factorisation work, wave q+1 runs the
// serial 16x16 factorisation.  Waves w and w+4 share a SIMD, and a worker on the factorising wave's SIMD costs that
// wave ~1000 cycles per step (measured), so the blocks are dealt to idle waves on the OTHER SIMDs, longest column to
// the least loaded SIMD; q = 6 is the row that is left after the last step (everybody idle).  The blocks live in LDS,
// so the owner of a column may change from step to step (a barrier lies between a block's write and its next read).
__device__ __forceinline__ int pv_col_of(int q, int wave) {
  // nibble `wave` of entry q = column + 1
  const unsigned tab = q == 1 ? 0x0000001u : q == 2 ? 0x0000021u : q == 3 ? 0x0003210u : q == 4 ? 0x0043201u
                     : q == 5 ? 0x0451032u : q == 6 ? 0x0561432u : 0u;
  return (int)((tab >> (4 * wave)) & 15u) - 1;
}

// 16x16 block (bi,bj) of M = L^-1 from the accumulator layout (v[c] = block[lk+4c][lr]) to global: M[bi][bj] and
// U[bj][bi] = its transpose.  Issued as soon as a block is final, so the stores drain behind the remaining steps
// instead of after them (the zero blocks are never written: the panel / update kernels skip the products that would
// read them).
__device__ __forceinline__ void store_mu_block(double* __restrict__ Mb, double* __restrict__ Ub, int bi, int bj,
                                               const double v[4], int lr, int lk) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    Mb[(16 * bi + lk + 4 * c) * NB + 16 * bj + lr] = v[c];
    Ub[(16 * bj + lr) * NB + 16 * bi + lk + 4 * c] = v[c];
  }
}

// Akk: the pivot tile (contiguous [NB][NB]); Ub, Mb, logdet_b, flags_b: this GP's; sm: PIVOT_LDS_BYTES of LDS
__device__ __forceinline__ void role_pivot(double* __restrict__ Akk, double* __restrict__ Ub, double* __restrict__ Mb,
                                           double* __restrict__ logdet_b, int32_t* __restrict__ flags_b, int with_p,
                                           double* __restrict__ sm) {
#ifdef GPSR_PIV_TRACE
  unsigned long long piv_t = __builtin_readcyclecounter();
  const unsigned long long piv_t0 = piv_t;
#endif
  double* S = sm;                           // [PV_NBLK][16][PV_DLD]
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  // 28 blocks x 16 rows x 8 units of 16 B = 3584 units = 8 per thread: all 8 loads are issued
  // before the first LDS store, so the block arrives in one memory latency instead of sixteen
  {
    double2 v[PV_NBLK * 128 / MM_THREADS];
    int dst[PV_NBLK * 128 / MM_THREADS];
#pragma unroll
    for (int i = 0; i < PV_NBLK * 128 / MM_THREADS; ++i) {
      const int u = tid + MM_THREADS * i;
      const int blk = u >> 7, r = (u >> 3) & 15, c = (u & 7) * 2;
      int bi = 0;
      while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
      const int bj = blk - bi * (bi + 1) / 2;
      v[i] = *(const double2*)(Akk + (size_t)(16 * bi + r) * NB + 16 * bj + c);
      dst[i] = blk * PV_BLK + r * PV_DLD + c;
    }
#pragma unroll
    for (int i = 0; i < PV_NBLK * 128 / MM_THREADS; ++i) { S[dst[i]] = v[i].x; S[dst[i] + 1] = v[i].y; }
  }
  __syncthreads();
  PIV_STAMP(1);
  double ld = 0.0;
  int bad = 0;
  double* stg = sm + PV_NBLK * PV_BLK + wave * PV_BLK;   // per-wave parking block
  // ---------------- phase 1: blocked Cholesky; afterwards SB(q,q) = inv(L16_q), SB(w,q) = L[w,q]
  // The 16x16 factor+inverse is a serial register kernel: ONE wave (the owner of the diagonal
  // block) runs it and publishes inv(L16_q) in place.
  // Look-ahead: the wave that owns diagonal block q+1 updates it with panel column q and factors
  // it at once, in registers (the MFMA accumulator layout is chol16_inv's input layout), while
  // the waves below finish their part of the trailing update of step q.
  if (wave == 0) {
    double a[4], m[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) a[c] = SB(0, 0)[(lk + 4 * c) * PV_DLD + lr];
    chol16_inv(a, m, lane, &ld, &bad);
#pragma unroll
    for (int c = 0; c < 4; ++c) SB(0, 0)[(lk + 4 * c) * PV_DLD + lr] = m[c];
    store_mu_block(Mb, Ub, 0, 0, m, lr, lk);
  }
  __syncthreads();
  for (int q = 0; q < NBT - 1; ++q) {
#ifdef GPSR_PIV_TRACE
    const bool crit = wave == q + 1;
    unsigned long long p2a = PIV2_T(), p2b = 0, p2c = 0, p2d = 0, p2e = 0;
#endif
    double an[4] = {0.0, 0.0, 0.0, 0.0};
    if (wave > q) {
      // L_wq = S[w,q] * inv(L16)^T, computed TRANSPOSED (A = inv(L16), B = S[w,q]: same products, same order): the
      // accumulator then holds L_wq[lr][lk+4c], which is both the [row][k] image the other waves read from LDS and this
      // wave's own A operand of the trailing update -- it stays in registers, no LDS read-back after the barrier
      const double* Mqq = SB(q, q);
      double* Swq = SB(wave, q);
      v4d cw = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        cw = __builtin_amdgcn_mfma_f64_16x16x4f64(Mqq[lr * PV_DLD + lk + 4 * ks], Swq[lr * PV_DLD + lk + 4 * ks], cw,
                                                  0, 0, 0);
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int c = 0; c < 4; ++c) { Swq[lr * PV_DLD + lk + 4 * c] = cw[c]; an[c] = -cw[c]; }
    }
#ifdef GPSR_PIV_TRACE
    p2b = PIV2_T();
#endif
    __syncthreads();   // panel column q written
#ifdef GPSR_PIV_TRACE
    p2c = PIV2_T();
#endif
    if (wave > q) {
      for (int j = q + 1; j <= wave; ++j) {
        double* Swj = SB(wave, j);
        const double* Sjq = SB(j, q);
        v4d acc;
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = Swj[(lk + 4 * c) * PV_DLD + lr];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(an[ks], Sjq[lr * PV_DLD + lk + 4 * ks], acc, 0, 0, 0);
        if (wave == q + 1) {   // j == wave: the next diagonal block, final -> factor it now
#ifdef GPSR_PIV_TRACE
          asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[3]));
          p2d = PIV2_T();
#endif
          double a[4], m[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) a[c] = acc[c];
          chol16_inv(a, m, lane, &ld, &bad);
#pragma unroll
          for (int c = 0; c < 4; ++c) Swj[(lk + 4 * c) * PV_DLD + lr] = m[c];
          store_mu_block(Mb, Ub, wave, wave, m, lr, lk);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) Swj[(lk + 4 * c) * PV_DLD + lr] = acc[c];
        }
      }
    }
    // M = L^-1 rides along: waves 0..q have no part in step q of the factorisation (their rows are done), row q of L and
    // inv(L16_q) are final, so they compute the blocks (q, c), c < q, of M now (minv_col_step), send them to global
    // (M and U = M^T), park them in LDS and, once everybody is done reading L[q,.] (the barrier), put them in place of
    // L[q,c].  After the last step only row 6 is left.  Which wave takes which column: pv_col_of.
    const int colq = pv_col_of(q, wave);
    if (colq >= 0) {
      const v4d xq = minv_col_step(S, q, colq, lr, lk);
      const double xv[4] = {xq[0], xq[1], xq[2], xq[3]};
      store_mu_block(Mb, Ub, q, colq, xv, lr, lk);
#pragma unroll
      for (int c = 0; c < 4; ++c) stg[(lk + 4 * c) * PV_DLD + lr] = xq[c];
    }
#ifdef GPSR_PIV_TRACE
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    p2e = PIV2_T();
#endif
    __syncthreads();
    if (colq >= 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) SB(q, colq)[(lk + 4 * c) * PV_DLD + lr] = stg[(lk + 4 * c) * PV_DLD + lr];
    }
#ifdef GPSR_PIV_TRACE
    {
      const unsigned long long p2f = PIV2_T();
      if (crit) {
        PIV2_ADD(0, 1); PIV2_ADD(1, p2b - p2a); PIV2_ADD(2, p2c - p2b); PIV2_ADD(3, p2d - p2c); PIV2_ADD(4, p2e - p2d);
        PIV2_ADD(5, p2f - p2e);
      } else if (wave == NBT - 1) { PIV2_ADD(6, p2e - p2c); PIV2_ADD(7, 1); }
    }
#endif
  }
  PIV_STAMP(2);
  // every wave factored one diagonal block: combine log-determinant and flags
  {
    __shared__ double ldsum[NBT];
    __shared__ int badsum[NBT];
    if (lane == 0) { ldsum[wave] = ld; badsum[wave] = bad; }
    __syncthreads();
    if (tid == 0) {
      ld = 0.0; bad = 0;
      for (int w = 0; w < NBT; ++w) { ld += ldsum[w]; bad |= badsum[w]; }
    }
  }
  // ---------------- phase 2 (what is left of it): the last row of M = L^-1; rows 1..5 were computed by the idle waves
  // during the factorisation.  Wave c < 6 owns column c; the last row costs it 7 - c block products.
  {
    v4d x6;
    const int col6 = pv_col_of(NBT - 1, wave);
    if (col6 >= 0) {
      x6 = minv_col_step(S, NBT - 1, col6, lr, lk);
      const double xv[4] = {x6[0], x6[1], x6[2], x6[3]};
      store_mu_block(Mb, Ub, NBT - 1, col6, xv, lr, lk);
    }
    if (with_p) {   // phase 3 reads M from LDS
      __syncthreads();
      if (col6 >= 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) SB(NBT - 1, col6)[(lk + 4 * c) * PV_DLD + lr] = x6[c];
      }
      __syncthreads();
    }
  }
  PIV_STAMP(3);
  // ---------------- phase 3: P = M^T M, wave a computes P[a, 0..a]; A_kk <- -P (with_p > 1: both triangles -- only the
  // last pivot of a sweep needs the mirror, every other pivot tile is updated and mirrored again by a later step).
  // Only when nothing follows that can do it (with_p: single-block matrices, the step before the alpha epilogue):
  // otherwise the update launch of the same step forms -M^T M as one more tile product (form_ptile: same products in
  // the same order, bit-identical), next to its other tiles instead of on this latency-bound workgroup's path.
  if (with_p) {
    // t outermost: the A operand (block (t,a) of M) is read once per t and feeds the a+1 independent accumulators, so the
    // MFMAs issue back to back; with bb outermost every MFMA waited for its own pair of LDS reads (runtime trip counts,
    // nothing hoisted): 8.6 us of a 37 us launch for 2.6 us of MFMA time.  Same order of additions per block.
    const int a = wave;
    v4d pacc[NBT];
#pragma unroll
    for (int bb = 0; bb < NBT; ++bb) pacc[bb] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NBT; ++t) {
      if (t < a) continue;
      const double* Mta = SB(t, a);
      double av[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) av[ks] = Mta[(lk + 4 * ks) * PV_DLD + lr];
#pragma unroll
      for (int bb = 0; bb < NBT; ++bb) {
        if (bb > a || bb > t) continue;
        const double* Mtb = SB(t, bb);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          pacc[bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[ks], Mtb[(lk + 4 * ks) * PV_DLD + lr], pacc[bb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int bb = 0; bb < NBT; ++bb) {
      if (bb > a) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        Akk[(size_t)(16 * a + lk + 4 * c) * NB + 16 * bb + lr] = -pacc[bb][c];
        if (bb != a && with_p > 1) Akk[(size_t)(16 * bb + lr) * NB + 16 * a + lk + 4 * c] = -pacc[bb][c];
      }
    }
  }
  __syncthreads();
  PIV_STAMP(4);
  if (tid == 0) {
    *logdet_b += ld;
    if (bad) *flags_b = 1;
  }
#ifdef GPSR_PIV_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  PIV_STAMP(5);
  if (tid == 0) { atomicAdd(&g_piv_trace[0], 1ull); atomicAdd(&g_piv_trace[6], __builtin_readcyclecounter() - piv_t0); }
#endif
}
#define PIVOT_LDS_BYTES ((PV_NBLK * PV_BLK + NBT * PV_BLK) * sizeof(double))

__global__ void __launch_bounds__(MM_THREADS, 4)
k_pivot(double* __restrict__ A, int N, int k, double* __restrict__ Ubuf, double* __restrict__ Mbuf,
        double* __restrict__ logdet, int32_t* __restrict__ flags, int with_p) {
  extern __shared__ double sm[];
  const int b = blockIdx.x;
  role_pivot(A + (size_t)b * N * N + TILE_OFF(k, k, N / NB), Ubuf + (size_t)b * NB * NB, Mbuf + (size_t)b * NB * NB,
             logdet + b, flags + b, with_p, sm);
}


// ------------------------------------------------- shared MFMA tile-product pieces
// Tile products accumulate acc[t] (16 rows of this wave x 16 cols of block t) over kk in
// [0, NB) from "k-major" LDS chunk images (element (kk, c) at [kk*NB + c]) that arrive by
// global_load_lds.  SKIP: the B operand is block-triangular at 16x16 granularity, the zero
// products are skipped:
//   SKIP_UPPER  Bop[kk][c] == 0 unless tile(c) >= chunk(kk)
//   SKIP_LOWER  Bop[kk][c] == 0 unless tile(c) <= chunk(kk)
enum { SKIP_NONE = 0, SKIP_UPPER = 1, SKIP_LOWER = 2, SKIP_DIAG = 3, SKIP_PTP = 4 };   // SKIP_DIAG: symmetric result, wave w computes blocks t <= w only
// SKIP_PTP: M^T M of the lower block-triangular M (both operands = M, k-major): SKIP_DIAG, and strip w starts at chunk w
// (block (chunk, w) of M is zero above the diagonal and was never written: k_pivot stores the non-zero blocks only)

// store a wave's 16x112 accumulator strip transposed: dst[(16*t + col) * ld + 16*wave + row] = sign*acc
// (per-wave 16x17 LDS scratch; the staging buffers must be free)
__device__ __forceinline__ void store_strip_transposed(const v4d acc[NBT], double* __restrict__ dst, size_t ld,
                                                       double* __restrict__ scratch, int wave, int lane,
                                                       double sign, double* __restrict__ dst2, int tmax = NBT) {
  const int lr = lane & 15, lk = lane >> 4;
  double* tr = scratch + wave * (16 * 17);
#pragma unroll
  for (int t = 0; t < NBT; ++t) {
    if (t >= tmax) break;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) tr[(lk + 4 * rr) * 17 + lr] = acc[t][rr];  // [row][col]
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int cl = lk + 4 * rr;  // col local
      const double v = tr[lr * 17 + cl];   // element (row = lr, col = cl)
      const size_t off = (size_t)(16 * t + cl) * ld + 16 * wave + lr;
      dst[off] = sign * v;
      if (dst2) dst2[off] = -sign * v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ----------------------------------------------------------------- update
// grid.x = (ntri(nblk-1) + nblk-1) * B.  Two kinds of 112x112 tile products:
//   trailing tile (I,J), I >= J, I,J != k:   A_IJ -= W_I W_J^T      (Aop = -Wo_I, Bop = Wo_J)
//   pivot-column tile I != k:                A_Ik  = W_I L^-1 = W_I M (Aop = Wo_I, Bop = M lower: SKIP_LOWER)
//     stored row-major into block (I,k) when I > k, transposed into block (k,I) when I < k.
// The 16-deep operand chunks go global -> LDS directly (global_load_lds_dwordx4: no VGPR
// staging, no LDS store pass), double-buffered, one barrier per chunk (9 % faster than a
// register-staged loop).  LDS image of a chunk = [16][112] doubles, unpadded (the DMA writes
// wave-uniform base + lane*16 B, so the image must be lane-linear); the MFMA operand reads
// (16 consecutive doubles per 16-lane group, rows 896 B apart) are bank-conflict free.
#define DMA_CHUNK (KC * NB)  // doubles per operand chunk
// experiment switch (tools/fused_time.py): the chunk loops of the workgroup-form tile products rolled instead of unrolled
#ifdef GPSR_ROLL_CHUNKS
#define CHUNK_LOOP_U _Pragma("nounroll")
#define CHUNK_LOOP_D _Pragma("nounroll")
#else
#define CHUNK_LOOP_U _Pragma("unroll")
#define CHUNK_LOOP_D
#endif

__device__ __forceinline__ void dma_chunk(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop,
                                          int ldb, int chunk, double* __restrict__ bufA, double* __restrict__ bufB,
                                          int wave, int lane) {
  // 14 KiB per operand = 14 wave-instructions of 1 KiB; wave w issues pieces 2w and 2w+1.
  // Address = wave-uniform chunk base (scalar registers) + a 32-bit per-lane byte offset that does not depend on
  // the chunk: the compiler keeps the four offsets in registers for the whole tile and the per-chunk address
  // arithmetic is scalar (measured with GPSR_UPD_TRACE=3: the 64-bit vector multiplies of the naive form cost
  // ~500 cycles per chunk between the barrier and the first MFMA).
  const char* pa = (const char*)(Aop + (size_t)(chunk * KC) * lda);
  const char* pb = (const char*)(Bop + (size_t)(chunk * KC) * ldb);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int piece = 2 * wave + u;
    const int e = piece * 128 + lane * 2;
    const int kk = e / NB, c = e % NB;
    const unsigned oa = (unsigned)(kk * lda + c) * 8u, ob = (unsigned)(kk * ldb + c) * 8u;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + oa),
                                     (__attribute__((address_space(3))) void*)(bufA + piece * 128), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + ob),
                                     (__attribute__((address_space(3))) void*)(bufB + piece * 128), 16, 0, 0);
  }
}

template <int SKIP, bool NEG_A>
__device__ __forceinline__ void chunk_mfma_dma(const double* __restrict__ As, const double* __restrict__ Bs,
                                               v4d acc[NBT], int wave, int lane, int ch) {
  const int lr = lane & 15, lk = lane >> 4;
  if (SKIP == SKIP_PTP && ch < wave) return;
#pragma unroll
  for (int ks = 0; ks < KC / 4; ++ks) {
    const int row = (ks * 4 + lk) * NB;
    const double a0 = As[row + 16 * wave + lr];
    const double av = NEG_A ? -a0 : a0;
#pragma unroll
    for (int t = 0; t < NBT; ++t) {
      if (SKIP == SKIP_LOWER && t > ch) continue;
      if ((SKIP == SKIP_DIAG || SKIP == SKIP_PTP) && t > wave) continue;
      const double bv = Bs[row + 16 * t + lr];
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
    }
  }
}

template <int SKIP, bool NEG_A>
__device__ __forceinline__ void tile_product_dma(const double* __restrict__ Aop, int lda,
                                                 const double* __restrict__ Bop, int ldb, double* __restrict__ lds,
                                                 v4d acc[NBT], int wave, int lane) {
  // lds: [2 buffers][A chunk | B chunk]
  dma_chunk(Aop, lda, Bop, ldb, 0, lds, lds + DMA_CHUNK, wave, lane);
  CHUNK_LOOP_D
  for (int ch = 0; ch < NB / KC; ++ch) {
    double* cur = lds + (ch & 1) * 2 * DMA_CHUNK;
    double* nxt = lds + ((ch + 1) & 1) * 2 * DMA_CHUNK;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of chunk ch have landed
    __syncthreads();                                   // everyone's have; everyone is done reading `nxt`
    if (ch + 1 < NB / KC) dma_chunk(Aop, lda, Bop, ldb, ch + 1, nxt, nxt + DMA_CHUNK, wave, lane);
    chunk_mfma_dma<SKIP, NEG_A>(cur, cur + DMA_CHUNK, acc, wave, lane, ch);
  }
}

// Same product for a trailing tile C -= ..., with the read of C taken off the critical
// path: chunk 0's DMA is issued first, then the 28 loads of the wave's C strip; chunk 0
// runs on zeroed accumulators as soon as the DMA (older in vmcnt order) has landed, and
// C is added when the loads have (before chunk 1).
__device__ __forceinline__ void tile_product_dma_latec(const double* __restrict__ Aop, int lda,
                                                       const double* __restrict__ Bop, int ldb,
                                                       double* __restrict__ lds, v4d acc[NBT],
                                                       const double* __restrict__ Ct, size_t ldc, int wave, int lane) {
  dma_chunk(Aop, lda, Bop, ldb, 0, lds, lds + DMA_CHUNK, wave, lane);
  v4d c0[NBT];
#pragma unroll
  for (int t = 0; t < NBT; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) c0[t][rr] = Ct[(size_t)(4 * rr) * ldc + 16 * t];
#pragma unroll
  for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  asm volatile("s_waitcnt vmcnt(28)" ::: "memory");    // chunk 0's 4 DMA pieces (older than the 28 C loads) have landed
  __builtin_amdgcn_s_barrier();                        // bare barrier: __syncthreads' fence would drain vmcnt to 0
  dma_chunk(Aop, lda, Bop, ldb, 1, lds + 2 * DMA_CHUNK, lds + 3 * DMA_CHUNK, wave, lane);
  chunk_mfma_dma<SKIP_NONE, true>(lds, lds + DMA_CHUNK, acc, wave, lane, 0);
#pragma unroll
  for (int t = 0; t < NBT; ++t) acc[t] += c0[t];   // the compiler folds these in as each acc[t] finishes chunk 0
  CHUNK_LOOP_D
  for (int ch = 1; ch < NB / KC; ++ch) {
    double* cur = lds + (ch & 1) * 2 * DMA_CHUNK;
    double* nxt = lds + ((ch + 1) & 1) * 2 * DMA_CHUNK;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ch + 1 < NB / KC) dma_chunk(Aop, lda, Bop, ldb, ch + 1, nxt, nxt + DMA_CHUNK, wave, lane);
    chunk_mfma_dma<SKIP_NONE, true>(cur, cur + DMA_CHUNK, acc, wave, lane, ch);
  }
}

// ---------------------------------------------------------------- strip mode (small batches)
// With a handful of GPs per launch (the reference's warm-start chain: one GP, or one per image) a 7-wave workgroup
// per 112x112 tile leaves most of the 256 CUs idle and the launch takes as long as ONE tile product on ONE CU.  In
// strip mode a tile is seven workgroups of ONE wave: each owns a 16-row strip, loads the whole B chunk (14 KB) and
// only its own 16 x 16 corner of the A chunk (2 KB), and, being alone, needs no barrier -- only s_waitcnt on its own
// LDS-DMA -- so it prefetches two chunks ahead (three stages of 16 KB).  Same per-strip arithmetic in the same order
// as the workgroup form: bit-identical results.  B-operand traffic is 7x (L2 hits), irrelevant at these sizes.
#define SCH (KC * NB + KC * 16)   // doubles per stage: B chunk [16][112] | A strip chunk [16][16]
#define S_STAGES 3

// ROWMAJOR_A: the A operand lies row-major ([row][kk], the tile below the pivot in k_panel): stored XOR-swizzled as in
// dma_chunk_panel (unit u of row r at slot 8r + (u ^ ((r>>1)&7))); otherwise k-major ([kk][row]) -> [kk][16].
template <bool ROWMAJOR_A>
__device__ __forceinline__ void strip_dma(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop, int ldb,
                                          int chunk, double* __restrict__ buf, int strip, int lane) {
  const char* pb = (const char*)(Bop + (size_t)(chunk * KC) * ldb);
#pragma unroll
  for (int u = 0; u < 14; ++u) {
    const int e = u * 128 + lane * 2;
    const int kk = e / NB, c = e % NB;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + (unsigned)(kk * ldb + c) * 8u),
                                     (__attribute__((address_space(3))) void*)(buf + u * 128), 16, 0, 0);
  }
  double* bufA = buf + KC * NB;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int q = u * 8 + (lane >> 3), j = lane & 7;   // q: kk (k-major) or row (row-major); j: 16-byte unit
    const char* pa;
    if (!ROWMAJOR_A) pa = (const char*)(Aop + (size_t)(chunk * KC + q) * lda + 16 * strip + 2 * j);
    else pa = (const char*)(Aop + (size_t)(16 * strip + q) * lda + chunk * KC + 2 * (j ^ ((q >> 1) & 7)));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)pa,
                                     (__attribute__((address_space(3))) void*)(bufA + u * 128), 16, 0, 0);
  }
}

template <int SKIP, bool NEG_A, bool ROWMAJOR_A, bool SWAP = false>   // SWAP: operands exchanged, acc[t] holds the block TRANSPOSED
__device__ __forceinline__ void strip_mfma(const double* __restrict__ buf, v4d acc[NBT], int strip, int lane, int ch) {
  const int lr = lane & 15, lk = lane >> 4;
  const double* As = buf + KC * NB;
  if (SKIP == SKIP_PTP && ch < strip) return;
#pragma unroll
  for (int ks = 0; ks < KC / 4; ++ks) {
    const int kk = ks * 4 + lk;
    const double a0 = ROWMAJOR_A ? As[(lr * 8 + ((kk >> 1) ^ ((lr >> 1) & 7))) * 2 + (kk & 1)] : As[kk * 16 + lr];
    const double av = NEG_A ? -a0 : a0;
#pragma unroll
    for (int t = 0; t < NBT; ++t) {
      if (SKIP == SKIP_LOWER && t > ch) continue;
      if (SKIP == SKIP_UPPER && t < ch) continue;
      if ((SKIP == SKIP_DIAG || SKIP == SKIP_PTP) && t > strip) continue;
      const double bv = buf[kk * NB + 16 * t + lr];
      acc[t] = SWAP ? __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc[t], 0, 0, 0)
                    : __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
    }
  }
}

// Wait until at most N_ of this wave's vector-memory operations are outstanding AND all of its LDS reads have returned:
// the DMA issued next overwrites the stage whose operands the previous iteration read, and an LDS read that has been
// issued but is still queued behind other workgroups' LDS traffic can be overtaken by DMA data that hits in L2 (seen
// as garbage operands -> non-positive pivots once a second process shared the GPU; the workgroup form is safe because
// __syncthreads() drains lgkmcnt before its barrier).
template <int N_> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N_) : "memory"); }

// acc (+)= (+-)A1 B1 [+ (+-)A2 B2]: NPH phases of 7 chunks, one continuous stream.  LATEC: acc starts at zero and the
// strip of C (Ct, ldc) is loaded behind chunk 0's DMA and added after chunk 0's products (as tile_product_dma_latec).
// Every strip_dma is 16 LDS-DMA instructions, the 28 loads of C sit between chunk 0's and chunk 1's: waiting for
// chunk g leaves exactly the instructions issued after it outstanding.
template <int SKIP1, bool NEG1, int SKIP2, bool NEG2, bool LATEC, int NPH, bool ROWMAJOR_A1, bool SWAP = false>
__device__ __forceinline__ void strip_product(const double* __restrict__ A1, int lda1, const double* __restrict__ B1, int ldb1,
                                              const double* __restrict__ A2, int lda2, const double* __restrict__ B2, int ldb2,
                                              double* __restrict__ lds, v4d acc[NBT], const double* __restrict__ Ct, size_t ldc,
                                              int strip, int lane) {
  constexpr int NC = NB / KC, G = NPH * NC;
  strip_dma<ROWMAJOR_A1>(A1, lda1, B1, ldb1, 0, lds, strip, lane);
  v4d c0[NBT];
  if (LATEC) {
#pragma unroll
    for (int t = 0; t < NBT; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) c0[t][rr] = Ct[(size_t)(4 * rr) * ldc + 16 * t];
#pragma unroll
    for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  strip_dma<ROWMAJOR_A1>(A1, lda1, B1, ldb1, 1, lds + SCH, strip, lane);
#pragma unroll
  for (int g = 0; g < G; ++g) {
    double* cur = lds + (g % S_STAGES) * SCH;
    if (g == 0 && LATEC) wait_vmcnt<16 + 28>();
    else if (g + 1 < G) wait_vmcnt<16>();
    else wait_vmcnt<0>();
    if (g + 2 < G) {   // into the stage read at iteration g-1: its LDS reads were consumed by that iteration's MFMAs
      double* nxt = lds + ((g + 2) % S_STAGES) * SCH;
      if (g + 2 < NC) strip_dma<ROWMAJOR_A1>(A1, lda1, B1, ldb1, g + 2, nxt, strip, lane);
      else strip_dma<false>(A2, lda2, B2, ldb2, g + 2 - NC, nxt, strip, lane);
    }
    if (g < NC) strip_mfma<SKIP1, NEG1, ROWMAJOR_A1, SWAP>(cur, acc, strip, lane, g);
    else strip_mfma<SKIP2, NEG2, false, SWAP>(cur, acc, strip, lane, g - NC);
    if (LATEC && g == 0) {
#pragma unroll
      for (int t = 0; t < NBT; ++t) acc[t] += c0[t];
    }
  }
}

// one interface for both forms
template <int SKIP, bool NEG_A, bool STRIP>
__device__ __forceinline__ void tile_product(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop, int ldb,
                                             double* __restrict__ lds, v4d acc[NBT], int wave, int lane) {
  if constexpr (STRIP) strip_product<SKIP, NEG_A, SKIP_NONE, false, false, 1, false>(Aop, lda, Bop, ldb, nullptr, 0, nullptr, 0, lds, acc, nullptr, 0, wave, lane);
  else tile_product_dma<SKIP, NEG_A>(Aop, lda, Bop, ldb, lds, acc, wave, lane);
}
template <bool STRIP>
__device__ __forceinline__ void tile_product_latec(const double* __restrict__ Aop, int lda, const double* __restrict__ Bop, int ldb,
                                                   double* __restrict__ lds, v4d acc[NBT], const double* __restrict__ Ct, size_t ldc,
                                                   int wave, int lane) {
  if constexpr (STRIP) strip_product<SKIP_NONE, true, SKIP_NONE, false, true, 1, false>(Aop, lda, Bop, ldb, nullptr, 0, nullptr, 0, lds, acc, Ct, ldc, wave, lane);
  else tile_product_dma_latec(Aop, lda, Bop, ldb, lds, acc, Ct, ldc, wave, lane);
}

// ---- alpha = K^-1 (y - c) without re-reading the inverse -------------------------------
// At the last pivot step every tile of -K^-1 except the pivot block is final in the
// accumulators of k_update.  Each workgroup leaves the two mat-vec partials of its tile,
//   part[0][r] = sum_c T[r][c] r_J[c]   (-> rows of block I)
//   part[1][c] = sum_r T[r][c] r_I[r]   (-> rows of block J, by symmetry; for a diagonal
//                tile: the blocks left of the diagonal only, and the target is block I too)
// and k_grad1r adds them up in a fixed order.  acc[t][rr] = T[16 wave + lk + 4 rr][16 t + lr].
#define G1_COLBUF (NBT * 16 * 17)   // doubles: behind the per-wave transposition scratch
// rv: LDS, [0, NB) = r of row block I, [NB, 2NB) = r of column block J (staged at kernel start by grad1_stage_r,
// so the epilogue does not wait on global memory).
__device__ __forceinline__ void grad1_stage_r(double* __restrict__ rv, const double* __restrict__ yb, double mean,
                                              int n, int I, int J, int tid) {
  if (tid < 2 * NB) {
    const int g = (tid < NB ? I : J) * NB + (tid < NB ? tid : tid - NB);
    rv[tid] = g < n ? yb[g] - mean : 0.0;
  }
}

// x + (x of the lane's partner under a DPP row permutation): quad_perm [1,0,3,2] (0xB1), [2,3,0,1] (0x4E), row_half_mirror
// (0x141), row_mirror (0x140) in turn sum the 16 lanes of a row into every lane -- the same pairs, hence the same bits,
// as the xor-1/2/4/8 butterfly of __shfl_xor, without its eight ds_bpermute round trips through the LDS pipeline.
template <int CTRL>
__device__ __forceinline__ double dpp_add(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, false);
  return x + __hiloint2double(hi, lo);
}

__device__ __forceinline__ void grad1_partials(const v4d acc[NBT], bool diag, const double* __restrict__ rv,
                                               double* __restrict__ lds, double* __restrict__ part, int wave,
                                               int lane, int tid) {
  const int lr = lane & 15, lk = lane >> 4;
  double rI[4], rJ[NBT];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) rI[rr] = rv[16 * wave + lk + 4 * rr];
#pragma unroll
  for (int t = 0; t < NBT; ++t) rJ[t] = rv[NB + 16 * t + lr];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    double p = 0.0;
#pragma unroll
    for (int t = 0; t < NBT; ++t)
      if (!diag || t <= wave) p = fma(acc[t][rr], rJ[t], p);
    p = dpp_add<0xB1>(p); p = dpp_add<0x4E>(p); p = dpp_add<0x141>(p); p = dpp_add<0x140>(p);
    if (lr == 0) part[16 * wave + lk + 4 * rr] = p;
  }
  // column sums: the four row groups (lk) of a wave leave their partials side by side in LDS and the final pass adds
  // them as the xor-16 / xor-32 butterfly would, (q0 + q1) + (q2 + q3), then the waves in order: same bits, no
  // cross-row shuffles
  double* cb = lds + G1_COLBUF + (wave * 4 + lk) * NB;
#pragma unroll
  for (int t = 0; t < NBT; ++t) {
    double q = 0.0;
    if (!diag || t < wave) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) q = fma(acc[t][rr], rI[rr], q);
    }
    cb[16 * t + lr] = q;
  }
  __syncthreads();
  if (tid < NB) {
    double sacc = 0.0;
#pragma unroll
    for (int w = 0; w < NBT; ++w) {
      const double* c4 = lds + G1_COLBUF + (w * 4) * NB + tid;
      sacc += (c4[0] + c4[NB]) + (c4[2 * NB] + c4[3 * NB]);
    }
    part[NB + tid] = sacc;
  }
}

#ifdef GPSR_UPD_TRACE
// Debug build only (tools/upd_trace.py): where the cycles of a k_update2 compute wave go.  Sums over all
// wave-tiles: [0] wave-tiles, [1] cycles kernel entry -> exit, [2] waiting for this wave's DMA (s_waitcnt vmcnt),
// [3] waiting at the chunk barrier, [4] issuing the next chunk's DMA, [5] the chunk's LDS reads + MFMAs,
// [6] kernel entry -> first chunk wait, [7] last MFMA issued -> kernel exit (C stores, mirror).
__device__ unsigned long long g_upd_trace[8];
#define TRC_CLK() __builtin_readcyclecounter()
#endif

// Two-phase product for the fused update of TWO pivot steps: acc += (+-)A1 B1 + (+-)A2 B2 with one
// pass over C (k-depth 224), the 14 chunks double-buffered as one continuous stream.  LATEC as in
// tile_product_dma_latec (C read overlapped with chunk 0, which runs on zeroed accumulators).
template <int SKIP1, bool NEG1, int SKIP2, bool NEG2, bool LATEC>
__device__ __forceinline__ void tile_product_dma2(const double* __restrict__ A1, int lda1, const double* __restrict__ B1,
                                                  int ldb1, const double* __restrict__ A2, int lda2,
                                                  const double* __restrict__ B2, int ldb2, double* __restrict__ lds,
                                                  v4d acc[NBT], const double* __restrict__ Ct, size_t ldc, int wave,
                                                  int lane) {
  constexpr int NC = NB / KC;
  dma_chunk(A1, lda1, B1, ldb1, 0, lds, lds + DMA_CHUNK, wave, lane);
  v4d c0[NBT];
  if (LATEC) {
#pragma unroll
    for (int t = 0; t < NBT; ++t)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) c0[t][rr] = Ct[(size_t)(4 * rr) * ldc + 16 * t];
#pragma unroll
    for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
#ifdef GPSR_UPD_TRACE
  unsigned long long trc = 0;   // GPSR_UPD_TRACE == 1: s_waitcnt + barrier; == 2: LDS reads + MFMAs; == 3: DMA issue
#endif
  CHUNK_LOOP_U
  for (int g = 0; g < 2 * NC; ++g) {
    double* cur = lds + (g & 1) * 2 * DMA_CHUNK;
    double* nxt = lds + ((g + 1) & 1) * 2 * DMA_CHUNK;
#if GPSR_UPD_TRACE == 1
    const unsigned long long ta = TRC_CLK();
#endif
    if (LATEC && g == 0) {
      asm volatile("s_waitcnt vmcnt(28)" ::: "memory");   // chunk 0's DMA pieces are older than the 28 C loads
      __builtin_amdgcn_s_barrier();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
#if GPSR_UPD_TRACE == 1
    trc += TRC_CLK() - ta;
#endif
#if GPSR_UPD_TRACE == 3
    const unsigned long long tb = TRC_CLK();
#endif
    if (g + 1 < NC) dma_chunk(A1, lda1, B1, ldb1, g + 1, nxt, nxt + DMA_CHUNK, wave, lane);
    else if (g + 1 < 2 * NC) dma_chunk(A2, lda2, B2, ldb2, g + 1 - NC, nxt, nxt + DMA_CHUNK, wave, lane);
#if GPSR_UPD_TRACE == 3
    trc += TRC_CLK() - tb;
#endif
#if GPSR_UPD_TRACE == 2
    const unsigned long long tc = TRC_CLK();
#endif
    if (g < NC) chunk_mfma_dma<SKIP1, NEG1>(cur, cur + DMA_CHUNK, acc, wave, lane, g);
    else chunk_mfma_dma<SKIP2, NEG2>(cur, cur + DMA_CHUNK, acc, wave, lane, g - NC);
    if (LATEC && g == 0) {
#pragma unroll
      for (int t = 0; t < NBT; ++t) acc[t] += c0[t];
    }
#if GPSR_UPD_TRACE == 2
    trc += TRC_CLK() - tc;
#endif
  }
#ifdef GPSR_UPD_TRACE
  if (LATEC && lane == 0) atomicAdd(&g_upd_trace[1 + GPSR_UPD_TRACE], trc);   // off-diagonal trailing tiles only
#endif
}

// tile (k,k) of -A^-1 after pivot step k: -P = -M^T M (what k_pivot's phase 3 computes when asked to)
template <bool STRIP>
__device__ __forceinline__ void form_ptile(double* __restrict__ Tkk, const double* __restrict__ M, double* __restrict__ lds,
                                           int wave, int lane, bool mirror) {
  const int lr = lane & 15, lk = lane >> 4;
  v4d acc[NBT];
#pragma unroll
  for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  tile_product<SKIP_PTP, true, STRIP>(M, NB, M, NB, lds, acc, wave, lane);
  double* Ct = Tkk + (size_t)(16 * wave + lk) * NB + lr;
#pragma unroll
  for (int t = 0; t < NBT; ++t) {
    if (t > wave) break;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
  }
  if (mirror) {
    __syncthreads();
    store_strip_transposed(acc, Tkk, (size_t)NB, lds, wave, lane, 1.0, nullptr, wave);
  }
}

template <int SKIP1, bool NEG1, int SKIP2, bool NEG2, bool LATEC, bool STRIP>
__device__ __forceinline__ void tile_product2(const double* __restrict__ A1, int lda1, const double* __restrict__ B1, int ldb1,
                                              const double* __restrict__ A2, int lda2, const double* __restrict__ B2, int ldb2,
                                              double* __restrict__ lds, v4d acc[NBT], const double* __restrict__ Ct, size_t ldc,
                                              int wave, int lane) {
  if constexpr (STRIP) strip_product<SKIP1, NEG1, SKIP2, NEG2, LATEC, 2, false>(A1, lda1, B1, ldb1, A2, lda2, B2, ldb2, lds, acc, Ct, ldc, wave, lane);
  else tile_product_dma2<SKIP1, NEG1, SKIP2, NEG2, LATEC>(A1, lda1, B1, ldb1, A2, lda2, B2, ldb2, lds, acc, Ct, ldc, wave, lane);
}

// Role form of one tile of a pivot step (k_update below is the launch wrapper).  Ab, Wob, Mb, yb, rawb, partb: this
// GP's matrix, panel, pivot factor, targets, raw hyper-parameters and mat-vec partials; t0: the tile (see k_update);
// lds: STRIP ? S_STAGES*SCH : 4*DMA_CHUNK doubles; rv: 2*NB doubles (G1 only).
template <bool G1, bool STRIP>
__device__ __forceinline__ void role_update(double* __restrict__ Ab, int nblk, int k, const double* __restrict__ Wob,
                                            const double* __restrict__ Mb, int ntri1, int t0,
                                            const double* __restrict__ yb, const float* __restrict__ rawb, int n,
                                            double* __restrict__ partb, int sel, double* __restrict__ lds,
                                            double* __restrict__ rv, int wave, int lane, int tid) {
  if constexpr (!G1) {
    if (sel < 0 && t0 == ntri1 + nblk - 1) {
      form_ptile<STRIP>(Ab + TILE_OFF(k, k, nblk), Mb, lds, wave, lane, true);
      return;
    }
  }
  const int lr = lane & 15, lk = lane >> 4;
  v4d acc[NBT];
  int I, J;
  bool trailing;
  if (sel < 0) {
    trailing = t0 < ntri1;
    if (trailing) { tri_decode(t0, I, J); if (I >= k) ++I; if (J >= k) ++J; }
    else { I = t0 - ntri1; if (I >= k) ++I; J = k; }
  } else {
    trailing = t0 < nblk - 1;
    if (trailing) { const int other = t0 + (t0 >= k ? 1 : 0); I = other > sel ? other : sel; J = other > sel ? sel : other; }
    else { I = sel; J = k; }
  }
  if (trailing) {
    if (G1) grad1_stage_r(rv, yb, (double)rawb[1], n, I, J, tid);
    const double* Aop = Wob + (size_t)I * (NB * NB);   // used negated
    const double* Bop = Wob + (size_t)J * (NB * NB);
    double* Ct = Ab + TILE_OFF(I, J, nblk) + (size_t)(16 * wave + lk) * NB + lr;
    if (I == J) {
      // symmetric diagonal tile: wave w computes the 16x16 blocks t <= w and mirrors t < w
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[t][rr] = t <= wave ? Ct[(size_t)(4 * rr) * NB + 16 * t] : 0.0;
      tile_product<SKIP_DIAG, true, STRIP>(Aop, NB, Bop, NB, lds, acc, wave, lane);
#pragma unroll
      for (int t = 0; t < NBT; ++t) {
        if (t > wave) break;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
      }
      // The blocks above the diagonal of a diagonal tile are a mirror nobody reads during the sweep (k_pivot, the
      // C reads of later updates and k_grad2_8 take the blocks on and below it): only a launch that can be the last one
      // to touch the tile writes them (for gpsr_invert's callers, k_grad1r and the generic k_grad2); the restricted
      // first half of a pivot pair and k_update2 are always followed by a single step and skip the mirror.
      if (G1 || sel < 0) {
        __syncthreads();   // chunk buffers are free: per-wave transposition scratch
        store_strip_transposed(acc, Ab + TILE_OFF(I, I, nblk), (size_t)NB, lds, wave, lane, 1.0, nullptr, wave);
      }
      if (G1) grad1_partials(acc, true, rv, lds, partb + (size_t)t0 * (2 * NB), wave, lane, tid);
    } else {
      tile_product_latec<STRIP>(Aop, NB, Bop, NB, lds, acc, Ct, (size_t)NB, wave, lane);
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
      if (G1) {
        __syncthreads();   // every wave is out of the chunk loop: the LDS buffers are free
        grad1_partials(acc, false, rv, lds, partb + (size_t)t0 * (2 * NB), wave, lane, tid);
      }
    }
  } else {
    if (G1) grad1_stage_r(rv, yb, (double)rawb[1], n, I, k, tid);
    const double* Aop = Wob + (size_t)I * (NB * NB);
    const double* Bop = Mb;
#pragma unroll
    for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    tile_product<SKIP_LOWER, false, STRIP>(Aop, NB, Bop, NB, lds, acc, wave, lane);
    if (I > k) {
      double* Ct = Ab + TILE_OFF(I, k, nblk) + (size_t)(16 * wave + lk) * NB + lr;
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
    } else {
      __syncthreads();
      store_strip_transposed(acc, Ab + TILE_OFF(k, I, nblk), (size_t)NB, lds, wave, lane, 1.0, nullptr);
    }
    if (G1) {
      __syncthreads();
      grad1_partials(acc, false, rv, lds, partb + (size_t)t0 * (2 * NB), wave, lane, tid);
    }
  }
}


template <bool G1, bool STRIP>   // G1: last pivot step of a fit evaluation, leaves the alpha mat-vec partials (own
                     // instantiation: the epilogue's registers must not weigh on the six launches that do not need it).
                     // STRIP: one wave per workgroup, seven workgroups per tile (small batches; never with G1)
__global__ void __launch_bounds__(STRIP ? 64 : MM_THREADS, STRIP ? 2 : 4)
k_update(double* __restrict__ A, int N, int nblk, int k,
             const double* __restrict__ Wo, const double* __restrict__ Mbuf, int ntri1, int total,
             const double* __restrict__ y, const float* __restrict__ raw, int n, double* __restrict__ part,
             int sel, int ptile) {   // ptile (sel < 0, never with G1): one more tile per GP, (k,k) = -M^T M
  static_assert(!(G1 && STRIP), "the alpha epilogue reduces across the waves of a workgroup");
  __shared__ __attribute__((aligned(16))) double lds[STRIP ? S_STAGES * SCH : 4 * DMA_CHUNK];   // strip form: 48 KB, three workgroups per CU
  __shared__ double rv[G1 ? 2 * NB : 1];   // G1 only: r = y - c of the tile's row and column blocks
  const int tid = threadIdx.x, lane = tid & 63;
  const int vv = xcd_remap(blockIdx.x, total);
  const int wave = STRIP ? vv % NBT : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int v = STRIP ? vv / NBT : vv;
  // sel < 0: every tile of the step.  sel >= 0 (first half of a two-pivot step): only the nblk tiles that
  // the NEXT pivot's panel reads, i.e. those of block row / column `sel`: the nblk-1 trailing tiles
  // (sel, other) / (other, sel) and the pivot-column tile of row block sel.
  const int per = sel < 0 ? ntri1 + nblk - 1 + ptile : nblk;
  const int b = v / per, t0 = v % per;
  role_update<G1, STRIP>(A + (size_t)b * N * N, nblk, k, Wo + (size_t)b * NB * N, Mbuf + (size_t)b * NB * NB, ntri1, t0,
                         G1 ? y + (size_t)b * n : nullptr, G1 ? raw + 4 * b : nullptr, n,
                         G1 ? part + (size_t)b * per * (2 * NB) : nullptr, sel, lds, rv, wave, lane, tid);
}

// ------------------------------------------------- update of two pivot steps at once
// After pivot k1, panel k1 (-> Wo1), the nblk tiles of block row/column k2 = k1+1 (k_update with
// sel = k2), pivot k2 and panel k2 (-> Wo2), every other tile takes both rank-112 updates in ONE
// pass over A (k-depth 224: C is read and written once instead of twice):
//   (a) trailing (I,J), I,J not in {k1,k2}:  A_IJ  -= W1_I W1_J^T + W2_I W2_J^T
//   (b) I not in {k1,k2}:                    A_Ik1  = W1_I M1 - W2_I W2_k1^T   (pivot column of step k1, then trailing of k2)
//   (c) tile (k1,k1) (= -P1 from pivot k1):  A_k1k1 -= W2_k1 W2_k1^T;  with ptile (c'): A_k1k1 = -M1^T M1 - W2_k1 W2_k1^T
//   (d) I != k2:                             A_Ik2  = W2_I M2                  (pivot column of step k2)
//   (e) tile (k2,k2), with ptile:            A_k2k2 = -M2^T M2
// Diagonal tiles: blocks on and below the diagonal only (no mirror: a single step always follows, see k_update).
// grid.x = (ntri(nblk-2) + (nblk-2) + 1 + (nblk-1) + ptile) * B.
// Role form of one tile of the fused update (k_update2 below is the launch wrapper): Ab, W1, W2, M1, M2 are this GP's.
template <bool STRIP>
__device__ __forceinline__ void role_update2(double* __restrict__ Ab, int nblk, int k1, const double* __restrict__ W1,
                                             const double* __restrict__ W2, const double* __restrict__ M1,
                                             const double* __restrict__ M2, int t0, int ptile,
                                             double* __restrict__ lds, int wave, int lane) {
#ifdef GPSR_UPD_TRACE
  const unsigned long long trc_t0 = TRC_CLK();
#endif
  const int lr = lane & 15, lk = lane >> 4;
  const int k2 = k1 + 1, nb2 = nblk - 2, ntri2 = nb2 * (nb2 + 1) / 2;
  const int per = ntri2 + nb2 + 1 + nblk - 1 + ptile;
#define WT(I) ((size_t)(I) * (NB * NB))   // row block I of a tile-blocked panel
  v4d acc[NBT];
  if (t0 < ntri2) {                                   // (a)
    int I, J; tri_decode(t0, I, J);
    if (I >= k1) I += 2;
    if (J >= k1) J += 2;
    double* Ct = Ab + TILE_OFF(I, J, nblk) + (size_t)(16 * wave + lk) * NB + lr;
    if (I == J) {
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[t][rr] = t <= wave ? Ct[(size_t)(4 * rr) * NB + 16 * t] : 0.0;
      tile_product2<SKIP_DIAG, true, SKIP_DIAG, true, false, STRIP>(W1 + WT(I), NB, W1 + WT(J), NB, W2 + WT(I), NB,
                                                                    W2 + WT(J), NB, lds, acc, Ct, (size_t)NB, wave, lane);
#pragma unroll
      for (int t = 0; t < NBT; ++t) {
        if (t > wave) break;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
      }
    } else {
#ifdef GPSR_UPD_TRACE
      const unsigned long long trc_t1 = TRC_CLK();
#endif
      tile_product2<SKIP_NONE, true, SKIP_NONE, true, true, STRIP>(W1 + WT(I), NB, W1 + WT(J), NB, W2 + WT(I), NB,
                                                                   W2 + WT(J), NB, lds, acc, Ct, (size_t)NB, wave, lane);
#ifdef GPSR_UPD_TRACE
      const unsigned long long trc_t2 = TRC_CLK();
#endif
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
#ifdef GPSR_UPD_TRACE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stores retired: what the workgroup slot really costs
      const unsigned long long trc_t3 = TRC_CLK();
      if (lane == 0) {
        atomicAdd(&g_upd_trace[0], 1ull);
        atomicAdd(&g_upd_trace[1], trc_t3 - trc_t0);
        atomicAdd(&g_upd_trace[6], trc_t1 - trc_t0);
        atomicAdd(&g_upd_trace[7], trc_t3 - trc_t2);
      }
#endif
    }
  } else if (t0 < ntri2 + nb2) {                      // (b)
    int I = t0 - ntri2; if (I >= k1) I += 2;
#pragma unroll
    for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    tile_product2<SKIP_LOWER, false, SKIP_NONE, true, false, STRIP>(W1 + WT(I), NB, M1, NB,
                                                                    W2 + WT(I), NB, W2 + WT(k1), NB, lds, acc, nullptr,
                                                                    0, wave, lane);
    if (I > k1) {
      double* Ct = Ab + TILE_OFF(I, k1, nblk) + (size_t)(16 * wave + lk) * NB + lr;
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
    } else {
      __syncthreads();
      store_strip_transposed(acc, Ab + TILE_OFF(k1, I, nblk), (size_t)NB, lds, wave, lane, 1.0, nullptr);
    }
  } else if (t0 == ntri2 + nb2) {                     // (c)
    double* Ct = Ab + TILE_OFF(k1, k1, nblk) + (size_t)(16 * wave + lk) * NB + lr;
    if (ptile) {
#pragma unroll
      for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
      tile_product2<SKIP_PTP, true, SKIP_DIAG, true, false, STRIP>(M1, NB, M1, NB, W2 + WT(k1), NB, W2 + WT(k1), NB, lds, acc,
                                                                   nullptr, 0, wave, lane);
    } else {   // -P1 is there (k_pivot with_p)
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) acc[t][rr] = t <= wave ? Ct[(size_t)(4 * rr) * NB + 16 * t] : 0.0;
      tile_product<SKIP_DIAG, true, STRIP>(W2 + WT(k1), NB, W2 + WT(k1), NB, lds, acc, wave, lane);
    }
#pragma unroll
    for (int t = 0; t < NBT; ++t) {
      if (t > wave) break;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
    }
  } else if (ptile && t0 == per - 1) {                // (e)
    form_ptile<STRIP>(Ab + TILE_OFF(k2, k2, nblk), M2, lds, wave, lane, false);
  } else {                                            // (d)
    int I = t0 - (ntri2 + nb2 + 1); if (I >= k2) ++I;
#pragma unroll
    for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    tile_product<SKIP_LOWER, false, STRIP>(W2 + WT(I), NB, M2, NB, lds, acc, wave, lane);
    if (I > k2) {
      double* Ct = Ab + TILE_OFF(I, k2, nblk) + (size_t)(16 * wave + lk) * NB + lr;
#pragma unroll
      for (int t = 0; t < NBT; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ct[(size_t)(4 * rr) * NB + 16 * t] = acc[t][rr];
    } else {
      __syncthreads();
      store_strip_transposed(acc, Ab + TILE_OFF(k2, I, nblk), (size_t)NB, lds, wave, lane, 1.0, nullptr);
    }
  }
}

#undef WT

template <bool STRIP>
__global__ void __launch_bounds__(STRIP ? 64 : MM_THREADS, STRIP ? 2 : 4)
k_update2(double* __restrict__ A, int N, int nblk, int k1, const double* __restrict__ Wo1,
          const double* __restrict__ Wo2, const double* __restrict__ M1buf, const double* __restrict__ M2buf,
          int total, int ptile) {   // ptile: the pivots left tiles (k1,k1), (k2,k2) to this launch (roles (c'), (e))
  __shared__ __attribute__((aligned(16))) double lds[STRIP ? S_STAGES * SCH : 4 * DMA_CHUNK];   // strip form: 48 KB, three workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int vv = xcd_remap(blockIdx.x, total);
  const int wave = STRIP ? vv % NBT : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int v = STRIP ? vv / NBT : vv;
  const int nb2 = nblk - 2;
  const int per = nb2 * (nb2 + 1) / 2 + nb2 + 1 + nblk - 1 + ptile;
  const int b = v / per, t0 = v % per;
  role_update2<STRIP>(A + (size_t)b * N * N, nblk, k1, Wo1 + (size_t)b * NB * N, Wo2 + (size_t)b * NB * N,
                      M1buf + (size_t)b * NB * NB, M2buf + (size_t)b * NB * NB, t0, ptile, lds, wave, lane);
}


// ------------------------------------------------------------------ panel
// grid.x = (nblk-1) * B.  For row block i != k, T = A_ik:  W = T L^-T = T U, stored k-major,
// Wo[k'][i0 + r] = W[r][k'].  Wave w owns rows [16w, 16w+16) of the tile; U is upper
// block-triangular (blocks t >= chunk).  Both operands arrive by global_load_lds
// (double-buffered, one barrier per chunk):
//   B operand  U   : the chunk [16 kk][112 k'] as it lies in memory
//   A operand  T   : i < k -> block (k,i) holds T^T: chunk [16 kk][112 r] as it lies (k-major);
//                    i > k -> block (i,k) holds T row-major: the chunk is 112 rows x 128 B.
//                             The DMA writes LDS lane-linearly but each lane picks its own 16 B
//                             of global memory, so the 16-byte units of a row are stored
//                             XOR-swizzled (unit u of row r at slot 8r + (u ^ ((r>>1)&7))):
//                             the MFMA A reads (16 lanes = 16 rows, same kk) are conflict-free.
__device__ __forceinline__ void dma_chunk_panel(const double* __restrict__ Tsrc, size_t N, bool below,
                                                const double* __restrict__ U, int chunk, double* __restrict__ bufA,
                                                double* __restrict__ bufB, int wave, int lane) {
  // wave-uniform chunk bases + chunk-independent 32-bit per-lane byte offsets (see dma_chunk)
  const char* pb = (const char*)(U + (size_t)(chunk * KC) * NB);
  const char* pa = (const char*)(below ? Tsrc + chunk * KC : Tsrc + (size_t)(chunk * KC) * N);
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int piece = 2 * wave + u;
    const int e = piece * 128 + lane * 2;
    const int kk = e / NB, c = e % NB;
    const unsigned ob = (unsigned)(kk * NB + c) * 8u;
    unsigned oa;
    if (!below) oa = (unsigned)(kk * (int)N + c) * 8u;
    else {
      const int slot = piece * 64 + lane, r = slot >> 3, j = slot & 7;
      oa = (unsigned)(r * (int)N + 2 * (j ^ ((r >> 1) & 7))) * 8u;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pa + oa),
                                     (__attribute__((address_space(3))) void*)(bufA + piece * 128), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pb + ob),
                                     (__attribute__((address_space(3))) void*)(bufB + piece * 128), 16, 0, 0);
  }
}

// Ab, U, Wob: this GP's matrix, pivot factor and panel; i: the row block (!= k); lds: STRIP ? S_STAGES*SCH : 4*DMA_CHUNK doubles
template <bool STRIP>
__device__ __forceinline__ void role_panel(const double* __restrict__ Ab, int nblk, int k, int i,
                                           const double* __restrict__ U, double* __restrict__ Wob,
                                           double* __restrict__ lds, int wave, int lane) {
  const int lr = lane & 15, lk = lane >> 4;
  const bool below = i > k;
  const double* Tsrc = below ? Ab + TILE_OFF(i, k, nblk) : Ab + TILE_OFF(k, i, nblk);
  v4d acc[NBT];
#pragma unroll
  for (int t = 0; t < NBT; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  if constexpr (STRIP) {
    if (below) strip_product<SKIP_UPPER, false, SKIP_NONE, false, false, 1, true, true>(Tsrc, NB, U, NB, nullptr, 0, nullptr, 0, lds, acc, nullptr, 0, wave, lane);
    else strip_product<SKIP_UPPER, false, SKIP_NONE, false, false, 1, false, true>(Tsrc, NB, U, NB, nullptr, 0, nullptr, 0, lds, acc, nullptr, 0, wave, lane);
  } else {
  dma_chunk_panel(Tsrc, (size_t)NB, below, U, 0, lds, lds + DMA_CHUNK, wave, lane);
  const int r = 16 * wave + lr, rsw = (r >> 1) & 7;
  CHUNK_LOOP_U
  for (int ch = 0; ch < NB / KC; ++ch) {
    const double* As = lds + (ch & 1) * 2 * DMA_CHUNK;
    const double* Bs = As + DMA_CHUNK;
    double* nxt = lds + ((ch + 1) & 1) * 2 * DMA_CHUNK;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ch + 1 < NB / KC) dma_chunk_panel(Tsrc, (size_t)NB, below, U, ch + 1, nxt, nxt + DMA_CHUNK, wave, lane);
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int kk = ks * 4 + lk;
      const double av = below ? As[(r * 8 + ((kk >> 1) ^ rsw)) * 2 + (kk & 1)] : As[kk * NB + r];
#pragma unroll
      for (int t = 0; t < NBT; ++t) {
        if (t < ch) continue;
        const double bv = Bs[kk * NB + 16 * t + lr];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(bv, av, acc[t], 0, 0, 0);
      }
    }
  }
  }
  // The product is computed TRANSPOSED (U as the A operand, T as the B operand: same LDS reads, same products in the
  // same order), so the accumulators hold the k-major panel image itself,
  //   acc[t][rr] = W[r = 16*wave + lr][k' = 16t + lk + 4rr]  ->  Wo[k'][i0 + r],
  // and go out as 128-byte row segments with no transposition through LDS.
  double* Wt = Wob + (size_t)i * (NB * NB) + 16 * wave + lr;
#pragma unroll
  for (int t = 0; t < NBT; ++t)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) Wt[(size_t)(16 * t + lk + 4 * rr) * NB] = acc[t][rr];
}

template <bool STRIP>
__global__ void __launch_bounds__(STRIP ? 64 : MM_THREADS, STRIP ? 2 : 4)
k_panel(const double* __restrict__ A, int N, int nblk, int k, const double* __restrict__ Ubuf,
         double* __restrict__ Wo, int total) {
  __shared__ __attribute__((aligned(16))) double lds[STRIP ? S_STAGES * SCH : 4 * DMA_CHUNK];   // strip form: 48 KB, three workgroups per CU
  const int tid = threadIdx.x, lane = tid & 63;
  const int vv = xcd_remap(blockIdx.x, total);
  const int wave = STRIP ? vv % NBT : __builtin_amdgcn_readfirstlane(tid >> 6);
  const int v = STRIP ? vv / NBT : vv;
  const int b = v / (nblk - 1);
  int i = v % (nblk - 1); if (i >= k) ++i;
  role_panel<STRIP>(A + (size_t)b * N * N, nblk, k, i, Ubuf + (size_t)b * NB * NB, Wo + (size_t)b * NB * N, lds, wave, lane);
}

// ------------------------------------------------------------------ grad1
// grid (nblk, B), 448 threads.  With A = -Kinv (lower block-triangle):
//   alpha_i = sum_j Kinv_ij r_j,  r = y - mean (0 for padded rows)
// stats1[b][I][0..3] = { sum_i Kinv_ii, sum alpha^2, sum alpha r, sum alpha }  over rows of block I, i < n
__global__ void __launch_bounds__(MM_THREADS)
k_grad1(const double* __restrict__ A, int N, int n, int nblk, const double* __restrict__ y,
        const float* __restrict__ raw, double* __restrict__ alpha, double* __restrict__ stats1) {
  extern __shared__ double sm[];  // r[N] | part[4][NB] | rowacc[NB] | red[8*4]
  double* r = sm; double* part = sm + N; double* rowacc = part + 4 * NB; double* red = rowacc + NB;
  const int I = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const double mean = (double)raw[4 * b + 1];
  const double* yb = y + (size_t)b * n;
  for (int i = tid; i < N; i += MM_THREADS) r[i] = i < n ? yb[i] - mean : 0.0;
  __syncthreads();
  const double* Ab = A + (size_t)b * N * N;
  const int I0 = I * NB;
  // part A: rows of block I, columns [0, I0+NB)
  const int ncols = I0 + NB;
  for (int q = 0; q < 16; ++q) {
    const int row = 16 * wave + q;
    const double* ar = Ab + (size_t)(I0 + row) * N;
    double s = 0.0;
    for (int j = lane; j < ncols; j += 64) s = fma(ar[j], r[j], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) rowacc[row] = s;
  }
  // part B: columns of block I in the blocks below
  {
    const int c = tid % NB, g = tid / NB;
    double s = 0.0;
    for (int ip = I0 + NB + g; ip < N; ip += 4) s = fma(Ab[(size_t)ip * N + I0 + c], r[ip], s);
    part[g * NB + c] = s;
  }
  __syncthreads();
  double st[4] = {0.0, 0.0, 0.0, 0.0};
  if (tid < NB) {
    const int gi = I0 + tid;
    double s = rowacc[tid] + ((part[tid] + part[NB + tid]) + (part[2 * NB + tid] + part[3 * NB + tid]));
    double al = gi < n ? -s : 0.0;
    alpha[(size_t)b * N + gi] = al;
    if (gi < n) {
      st[0] = -Ab[(size_t)gi * N + gi];
      st[1] = al * al; st[2] = al * r[gi]; st[3] = al;
    }
  }
  if (tid < 128) {  // waves 0,1 reduce
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double s = st[u];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
      if (lane == 0) red[wave * 4 + u] = s;
    }
  }
  __syncthreads();
  if (tid < 4) stats1[((size_t)b * nblk + I) * 4 + tid] = red[tid] + red[4 + tid];
}

// ------------------------------------------------------------------ grad1 (reduce form)
// grid (nblk, B), 448 threads: same outputs as k_grad1 (alpha, stats1), from the per-tile
// mat-vec partials the last k_update launch left (grad1_partials) plus the pivot block of
// that step, the one tile k_update does not own.  Fixed summation order: deterministic.
#define G1R_LDS (NB + 4 * NB + 4 * NB + 32)   // doubles of LDS: rk | pk[4][NB] | pt4[4][NB] | red[32]
// Role form: row block R of one GP (Ab, yb, rawb, pb = its partials, alphab, stats1b = its outputs); g1s: G1R_LDS doubles
__device__ __forceinline__ void role_grad1r(const double* __restrict__ Ab, int n, int nblk, int k, int R,
                                            const double* __restrict__ yb, const float* __restrict__ rawb,
                                            const double* __restrict__ pb, int ntri1, double* __restrict__ alphab,
                                            double* __restrict__ stats1b, double* __restrict__ g1s, int tid) {
  double* rk = g1s;
  double (*pk)[NB] = (double (*)[NB])(g1s + NB);
  double (*pt4)[NB] = (double (*)[NB])(g1s + 5 * NB);
  double* red = g1s + 9 * NB;
  const int wave = tid >> 6, lane = tid & 63;
  const int per = ntri1 + nblk - 1;
  const double mean = (double)rawb[1];
  if (R == k) {   // pivot block: alpha_k += A_kk r_k  (A_kk symmetric, both triangles stored: read columns, coalesced)
    if (tid < NB) { const int gj = k * NB + tid; rk[tid] = gj < n ? yb[gj] - mean : 0.0; }
    __syncthreads();
    const int c = tid % NB, g = tid / NB;
    const double* Akk = Ab + TILE_OFF(k, k, nblk);
    double sacc = 0.0;
    for (int j = g; j < NB; j += 4) sacc = fma(Akk[(size_t)j * NB + c], rk[j], sacc);
    pk[g][c] = sacc;
    __syncthreads();
  }
  // the per-tile partials that land in row block R, added in a fixed order: four groups of threads take the
  // tiles t0 = g, g+4, ... (their loads are in flight together: a single thread walking all tiles waits for one
  // dependent global load after the other, 16 us per launch when the batch is small), then group sums 0..3
  {
    const int c = tid % NB, g = tid / NB;
    double sg = 0.0;
    for (int t0 = g; t0 < per; t0 += 4) {
      int I, J;
      if (t0 < ntri1) { tri_decode(t0, I, J); if (I >= k) ++I; if (J >= k) ++J; }
      else { I = t0 - ntri1; if (I >= k) ++I; J = k; }
      const double* pt = pb + (size_t)t0 * (2 * NB);
      if (I == R) sg += pt[c];
      if ((I == J ? I : J) == R) sg += pt[NB + c];
    }
    pt4[g][c] = sg;
    __syncthreads();
  }
  double st[4] = {0.0, 0.0, 0.0, 0.0};
  if (tid < NB) {
    double sacc = (R == k) ? (pk[0][tid] + pk[1][tid]) + (pk[2][tid] + pk[3][tid]) : 0.0;
    sacc += (pt4[0][tid] + pt4[1][tid]) + (pt4[2][tid] + pt4[3][tid]);
    const int gi = R * NB + tid;
    const double al = gi < n ? -sacc : 0.0;
    alphab[gi] = al;
    if (gi < n) {
      const double ri = yb[gi] - mean;
      st[0] = -Ab[TILE_OFF(R, R, nblk) + (size_t)tid * NB + tid];
      st[1] = al * al; st[2] = al * ri; st[3] = al;
    }
  }
  if (tid < 128) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double sacc = st[u];
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o);
      if (lane == 0) red[wave * 4 + u] = sacc;
    }
  }
  __syncthreads();
  if (tid < 4) stats1b[R * 4 + tid] = red[tid] + red[4 + tid];
}


__global__ void __launch_bounds__(MM_THREADS)
k_grad1r(const double* __restrict__ A, int N, int n, int nblk, int k, const double* __restrict__ y,
         const float* __restrict__ raw, const double* __restrict__ part, int ntri1,
         double* __restrict__ alpha, double* __restrict__ stats1) {
  __shared__ double g1s[G1R_LDS];
  const int R = blockIdx.x, b = blockIdx.y;
  role_grad1r(A + (size_t)b * N * N, n, nblk, k, R, y + (size_t)b * n, raw + 4 * b,
              part + (size_t)b * (ntri1 + nblk - 1) * (2 * NB), ntri1, alpha + (size_t)b * N, stats1 + (size_t)b * nblk * 4,
              g1s, threadIdx.x);
}

// ------------------------------------------------------------------ grad2
// grid.x = ntri*B, 256 threads, dynamic LDS = 2*NB*(d+1) + 2*NB doubles.
// stats2[b][t][0..1] = sum over tile of (Kinv_ij - a_i a_j) * {k_ij, dk_ij/dl}, off-diagonal tiles x2
template <int KID>
__global__ void k_grad2(const double* __restrict__ A, const double* __restrict__ X,
                        const float* __restrict__ raw, const double* __restrict__ alpha,
                        int n, int N, int d, int ntri, int total, double* __restrict__ stats2) {
  extern __shared__ double sm[];
  const int v = xcd_remap(blockIdx.x, total);
  const int b = v / ntri, t = v % ntri;
  int I, J; tri_decode(t, I, J);
  const int ldx = d + 1;
  double* xi = sm; double* xj = sm + NB * ldx; double* ai = xj + NB * ldx; double* aj = ai + NB;
  __shared__ double red[2][4];
  const double* Xb = X + (size_t)b * n * d;
  for (int e = threadIdx.x; e < NB * d; e += blockDim.x) {
    int r = e / d, f = e % d;
    int gi = I * NB + r, gj = J * NB + r;
    xi[r * ldx + f] = gi < n ? Xb[(size_t)gi * d + f] : 0.0;
    xj[r * ldx + f] = gj < n ? Xb[(size_t)gj * d + f] : 0.0;
  }
  for (int e = threadIdx.x; e < NB; e += blockDim.x) {
    ai[e] = alpha[(size_t)b * N + I * NB + e];
    aj[e] = alpha[(size_t)b * N + J * NB + e];
  }
  const Hyp h = load_hyp(raw + 4 * b);
  __syncthreads();
  const double* Ab = A + (size_t)b * N * N;
  double s0 = 0.0, s1 = 0.0;
  for (int e = threadIdx.x; e < NB * NB; e += blockDim.x) {
    int i = e / NB, j = e % NB;
    int gi = I * NB + i, gj = J * NB + j;
    if (gi < n && gj < n) {
      double w = -Ab[TILE_OFF(I, J, N / NB) + (size_t)i * NB + j] - ai[i] * aj[j];
      double st = pair_stat<KID>(xi + i * ldx, xj + j * ldx, d);
      double kv, dk;
      kern_val_dl<KID>(st, h, kv, dk);
      s0 = fma(w, kv, s0);
      s1 = fma(w, dk, s1);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    double s = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
    if (I != J) s *= 2.0;
    stats2[((size_t)b * ntri + t) * 2 + threadIdx.x] = s;
  }
}

// d == 8 fast path, same thread layout as k_build8 (x_j, alpha_j in registers; x_i, alpha_i LDS broadcasts).
#define G28_LDS (NB * 8 + NB + 2 * NBT + 2 + 6)   // doubles of LDS: xi | ai | red[2][NBT] | (pad) | hs
// Role form: tile t = (I,J) of one GP; g2s: G28_LDS doubles (16-byte aligned)
template <int KID>
__device__ __forceinline__ void role_grad2_8(const double* __restrict__ Agp, const double* __restrict__ Xb,
                                             const float* __restrict__ rawb, const double* __restrict__ alphab, int n,
                                             int N, int I, int J, double* __restrict__ stats2t,
                                             double* __restrict__ g2s, int tid) {
  double* xi = g2s;
  double* ai = g2s + NB * 8;
  double (*red)[NBT] = (double (*)[NBT])(g2s + NB * 9);
  double* hs = g2s + NB * 9 + 2 * NBT + 2;
  // diagonal tiles: only the 28 sub-blocks on and below the diagonal (16 threads each, one column per
  // thread); the strictly-lower ones count twice (symmetry), as the off-diagonal tiles do
  const bool sub = (I == J);
  int j, i0, istep = 4, iend = NB;
  double wsym = 1.0;
  if (!sub) { const int g = tid / NB; j = tid - g * NB; i0 = g; }
  else {
    int bi, bj; tri_decode(tid >> 4, bi, bj);
    j = 16 * bj + (tid & 15); i0 = 16 * bi; istep = 1; iend = i0 + 16;
    if (bi != bj) wsym = 2.0;
  }
  const int gj = J * NB + j;
  double xj[8];
#pragma unroll
  for (int f = 0; f < 8; ++f) xj[f] = gj < n ? Xb[(size_t)gj * 8 + f] : 0.0;
  const double aj = alphab[gj];
  for (int e = tid; e < NB * 8; e += MM_THREADS) {
    const int gi = I * NB + (e >> 3);
    xi[e] = gi < n ? Xb[(size_t)I * NB * 8 + e] : 0.0;
  }
  if (tid < NB) ai[tid] = alphab[I * NB + tid];
  load_hyp_shared_put(rawb, hs, tid);
  __syncthreads();
  const Hyp h = load_hyp_shared_get(hs);
  const double* Ab = Agp + TILE_OFF(I, J, N / NB) + j;
  double s0 = 0.0, s1 = 0.0;
  if (gj < n) {
    const int ireal = min(iend, n - I * NB);   // padding rows contribute nothing
#pragma unroll 2
    for (int i = i0; i < ireal; i += istep) {
      const double w = -Ab[(size_t)i * NB] - ai[i] * aj;
      const double st = pair_stat<KID>(xi + i * 8, xj, 8);
      double kv, dk;
      kern_val_dl<KID>(st, h, kv, dk);
      s0 = fma(w, kv, s0);
      s1 = fma(w, dk, s1);
    }
    s0 *= wsym; s1 *= wsym;   // 1 or 2: exact, and it commutes with every rounding of the sums above (bit-identical
                              // to scaling each weight)
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
  if (lane == 0) { red[0][wave] = s0; red[1][wave] = s1; }
  __syncthreads();
  if (tid < 2) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NBT; ++w) s += red[tid][w];
    if (I != J) s *= 2.0;
    stats2t[tid] = s;
  }
}


template <int KID>
__global__ void __launch_bounds__(MM_THREADS)
k_grad2_8(const double* __restrict__ A, const double* __restrict__ X, const float* __restrict__ raw,
          const double* __restrict__ alpha, int n, int N, int ntri, int total, double* __restrict__ stats2) {
  __shared__ __attribute__((aligned(16))) double g2s[G28_LDS];
  const int v = xcd_remap(blockIdx.x, total);
  const int b = v / ntri, t = v % ntri;
  int I, J; tri_decode(t, I, J);
  role_grad2_8<KID>(A + (size_t)b * N * N, X + (size_t)b * n * 8, raw + 4 * b, alpha + (size_t)b * N, n, N, I, J,
                    stats2 + ((size_t)b * ntri + t) * 2, g2s, threadIdx.x);
}

// ------------------------------------------------------------- loss/grad/Adam
// One thread per GP.  do_step: torch.optim.Adam single-tensor update in float32.
// One GP: r = its 4 raw parameters, m1 / m2 = its Adam moments, s1 [nblk][4], s2 [ntri][2] = its statistics.
__device__ __forceinline__ void adam_gp(float* __restrict__ r, float* __restrict__ m1, float* __restrict__ m2,
                                        const double* __restrict__ s1, const double* __restrict__ s2, double logdet_b,
                                        int n, int nblk, int ntri, double* __restrict__ loss_b,
                                        float* __restrict__ grad_b, int do_step, float step_size, float bc2_sqrt,
                                        float one_minus_b1, float b2, float one_minus_b2, float eps) {
  double tr = 0.0, aa = 0.0, ar = 0.0, sa = 0.0;
  for (int I = 0; I < nblk; ++I) {
    const double* s = s1 + (size_t)I * 4;
    tr += s[0]; aa += s[1]; ar += s[2]; sa += s[3];
  }
  double sk = 0.0, sd = 0.0;
  for (int t = 0; t < ntri; ++t) { sk += s2[(size_t)t * 2]; sd += s2[(size_t)t * 2 + 1]; }
  const double os = (double)softplus_f(r[2]);
  const double dn = (double)n;
  const double loss = 0.5 * (ar + logdet_b + dn * 1.8378770664093453) / dn;
  if (loss_b) *loss_b = loss;
  float g[4];
  g[0] = (float)(0.5 * (tr - aa) / dn) * sigmoid_f(r[0]);
  g[1] = (float)(-sa / dn);
  g[2] = (float)(0.5 * sk / dn) * sigmoid_f(r[2]);
  g[3] = (float)(0.5 * os * sd / dn) * sigmoid_f(r[3]);
  if (grad_b) { for (int u = 0; u < 4; ++u) grad_b[u] = g[u]; }
  if (do_step) {
    for (int u = 0; u < 4; ++u) {
      float gu = g[u];
      float ea = m1[u];
      ea = ea + one_minus_b1 * (gu - ea);               // lerp_
      float es = m2[u] * b2 + one_minus_b2 * gu * gu;  // mul_().addcmul_()
      m1[u] = ea; m2[u] = es;
      float denom = sqrtf(es) / bc2_sqrt + eps;
      r[u] = r[u] - step_size * (ea / denom);
    }
  }
}

__global__ void k_adam(float* __restrict__ raw, float* __restrict__ m1, float* __restrict__ m2,
                       const double* __restrict__ stats1, const double* __restrict__ stats2,
                       const double* __restrict__ logdet, int B, int n, int nblk, int ntri,
                       double* __restrict__ loss_out, float* __restrict__ grad_out, int do_step,
                       float step_size, float bc2_sqrt, float one_minus_b1, float b2,
                       float one_minus_b2, float eps) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  adam_gp(raw + 4 * b, m1 + 4 * b, m2 + 4 * b, stats1 + (size_t)b * nblk * 4, stats2 + (size_t)b * ntri * 2, logdet[b], n,
          nblk, ntri, loss_out ? loss_out + b : nullptr, grad_out ? grad_out + 4 * b : nullptr, do_step, step_size,
          bc2_sqrt, one_minus_b1, b2, one_minus_b2, eps);
}

// ---------------------------------------------------------------- fused per-GP fit (opt-in experiment)
// The launch-per-step route above passes the WHOLE batch through HBM a dozen times per evaluation (n = 324: ~4.5 MB of
// counted traffic per GP-evaluation against 0.6 MB for the lower triangle itself) and its launches are, one after the
// other, HBM-bound (build, panel, updates), latency-bound (pivot) or VALU-bound (build, gradient).  Here ONE workgroup
// owns one GP for its whole fit -- covariance, sweep, alpha, gradient statistics and the Adam step of every iteration, no
// launch boundary and no inter-workgroup dependence anywhere -- and a persistent grid of two workgroups per CU walks the
// batch; the matrix, panels and pivot factors live in a per-SLOT workspace (slot = resident workgroup: 512 x 1.9 MB at
// N = 336 instead of B x 1.9 MB).  The idea was that a workgroup re-reads within microseconds what it wrote, out of L2 /
// Infinity Cache.  MEASURED (MI355X, 10 800 GPs of n = 324): it does not -- 7.9 MB of HBM traffic per GP-evaluation
// against 4.5 MB (the 0.9 GB of live slots exceed the 256 MB Infinity Cache, every store goes through to HBM, and the
// out-of-line roles add their callee-saved-register traffic) -- and it is slower, 2.0 against 1.585 us per
// GP-evaluation, because two workgroups per CU hide none of a role's own latencies.  Kept as a
// tested, bit-identical option and as the reason every kernel body is a role function.
// The body is the sequence of role functions the launch wrappers above execute, on the same data in the same order:
// results are bit-identical to the launch-per-step route (tests/test_gpu_kernels.py::test_fit_fused_bit_identical).
// Within a workgroup a role's global stores are ordered before the next role's loads by __syncthreads() (workgroup-
// scope release / acquire: one CU, one vector L1); nothing the kernel writes is read through a const __restrict__
// kernel argument, so no read of it can be turned into a scalar load of stale data.
#define FUSED_MAX_IT 128
struct AdamTab { float step[FUSED_MAX_IT], bc2s[FUSED_MAX_IT]; };   // per-iteration Adam constants (host doubles -> float, as k_adam gets them)
struct FusedWs { double *A, *Wo, *Wo2, *U0, *M0, *U1, *M1, *stats1, *stats2, *logdet, *part; };   // [slots][...]

// Out-of-line copies of the roles for the fused kernel: inlined, the compiler hoists every role's lane-invariant address
// arithmetic to the top of the kernel and keeps it live across all roles (1.4 KB of scratch per lane); as calls each role
// gets its own register allocation, and a call per 112x112 tile product costs nothing measurable.
#define FZ __device__ __attribute__((noinline))
template <int KID>
FZ void fz_build8(const double* Xb, const float* rawb, double* Agp, int n, int N, int I, int J, double* xi, double* hs) {
  role_build8<KID>(Xb, rawb, Agp, n, N, I, J, 0, xi, hs, threadIdx.x);
}
FZ void fz_pivot(double* Akk, double* Ub, double* Mb, double* logdet_b, int32_t* flags_b, int with_p, double* sm) {
  role_pivot(Akk, Ub, Mb, logdet_b, flags_b, with_p, sm);
}
FZ void fz_panel(const double* Ab, int nblk, int k, int i, const double* U, double* Wob, double* lds) {
  const int tid = threadIdx.x;
  role_panel<false>(Ab, nblk, k, i, U, Wob, lds, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63);
}
FZ void fz_update(double* Ab, int nblk, int k, const double* Wob, const double* Mb, int ntri1, int t0, int sel, double* lds) {
  const int tid = threadIdx.x;
  role_update<false, false>(Ab, nblk, k, Wob, Mb, ntri1, t0, nullptr, nullptr, 0, nullptr, sel, lds, nullptr,
                            __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63, tid);
}
FZ void fz_update_last(double* Ab, int nblk, int k, const double* Wob, const double* Mb, int ntri1, int t0, const double* yb,
                       const float* rawb, int n, double* partb, double* lds, double* rv) {
  const int tid = threadIdx.x;
  role_update<true, false>(Ab, nblk, k, Wob, Mb, ntri1, t0, yb, rawb, n, partb, -1, lds, rv,
                           __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63, tid);
}
FZ void fz_update2(double* Ab, int nblk, int k1, const double* W1, const double* W2, const double* M1, const double* M2,
                   int t0, double* lds) {
  const int tid = threadIdx.x;
  role_update2<false>(Ab, nblk, k1, W1, W2, M1, M2, t0, 0, lds, __builtin_amdgcn_readfirstlane(tid >> 6), tid & 63);
}
FZ void fz_grad1r(const double* Ab, int n, int nblk, int k, int R, const double* yb, const float* rawb, const double* pb,
                  int ntri1, double* alphab, double* stats1b, double* g1s) {
  role_grad1r(Ab, n, nblk, k, R, yb, rawb, pb, ntri1, alphab, stats1b, g1s, threadIdx.x);
}
template <int KID>
FZ void fz_grad2_8(const double* Agp, const double* Xb, const float* rawb, const double* alphab, int n, int N, int I, int J,
                   double* stats2t, double* g2s) {
  role_grad2_8<KID>(Agp, Xb, rawb, alphab, n, N, I, J, stats2t, g2s, threadIdx.x);
}
#undef FZ

#ifdef GPSR_FUSED_TRACE
// Debug build only (tools/fused_trace.py): cycles of a fused workgroup by role, summed over evaluations and workgroups:
// [0] evaluations, [1] covariance, [2] pivots, [3] panels, [4] single-step / restricted updates, [5] two-pivot updates,
// [6] last update (+ alpha partials), [7] alpha reduce, [8] gradient statistics, [9] whole evaluation
__device__ unsigned long long g_fused_trace[16];
#define FTR_T() __builtin_readcyclecounter()
#define FTR_ADD(i) do { if (tid == 0) { const unsigned long long t_ = FTR_T(); atomicAdd(&g_fused_trace[i], t_ - ftr_t); ftr_t = t_; } } while (0)
#else
#define FTR_ADD(i)
#endif

template <int KID>
__device__ __forceinline__ void fused_eval(const double* __restrict__ Xb, const double* __restrict__ yb, float* rawb,
                                           double* alphab, int32_t* flagsb, int n, int N, int nblk, bool with_grad,
                                           int pair_steps, const FusedWs& s, double* sm, int tid, int wave, int lane) {
  const int ntri = nblk * (nblk + 1) / 2, ntri1 = (nblk - 1) * nblk / 2;
#ifdef GPSR_FUSED_TRACE
  unsigned long long ftr_t = FTR_T();
  const unsigned long long ftr_t0 = ftr_t;
#endif
  double* lds = sm;
  double* rv = sm + 4 * DMA_CHUNK;
  for (int t = 0; t < ntri; ++t) {
    int I, J; tri_decode(t, I, J);
    fz_build8<KID>(Xb, rawb, s.A, n, N, I, J, sm, sm + NB * 8);
    __syncthreads();
  }
  if (tid == 0) *s.logdet = 0.0;
  FTR_ADD(1);
  const int npairs = (pair_steps && nblk >= 3) ? (nblk - 1 - (nblk % 2 == 0 ? 1 : 0)) / 2 : 0;   // as append_sweep_ops
  const int nb2 = nblk - 2, per2 = nb2 * (nb2 + 1) / 2 + nb2 + 1 + nblk - 1, per1 = ntri1 + nblk - 1;
  for (int j = 0; j < nblk; ++j) {
    const bool inpair = j < 2 * npairs, second = inpair && (j & 1), first = inpair && !second, last = j == nblk - 1;
    double* U = second ? s.U1 : s.U0;
    double* M = second ? s.M1 : s.M0;
    double* W = second ? s.Wo2 : s.Wo;
    fz_pivot(s.A + TILE_OFF(j, j, nblk), U, M, s.logdet, flagsb, last ? 2 : 1, sm);
    __syncthreads();
    FTR_ADD(2);
    for (int q = 0; q < nblk - 1; ++q) {
      fz_panel(s.A, nblk, j, q + (q >= j ? 1 : 0), U, W, lds);
      __syncthreads();
    }
    FTR_ADD(3);
    if (second) {
      for (int t0 = 0; t0 < per2; ++t0) {
        fz_update2(s.A, nblk, j - 1, s.Wo, s.Wo2, s.M0, s.M1, t0, lds);
        __syncthreads();
      }
      FTR_ADD(5);
    } else if (!first && last) {
      for (int t0 = 0; t0 < per1; ++t0) {
        fz_update_last(s.A, nblk, j, W, M, ntri1, t0, yb, rawb, n, s.part, lds, rv);
        __syncthreads();
      }
      FTR_ADD(6);
    } else {
      const int sel = first ? j + 1 : -1, cnt = first ? nblk : per1;
      for (int t0 = 0; t0 < cnt; ++t0) {
        fz_update(s.A, nblk, j, W, M, ntri1, t0, sel, lds);
        __syncthreads();
      }
      FTR_ADD(4);
    }
  }
  for (int R = 0; R < nblk; ++R) {
    fz_grad1r(s.A, n, nblk, nblk - 1, R, yb, rawb, s.part, ntri1, alphab, s.stats1, sm);
    __syncthreads();
  }
  FTR_ADD(7);
  if (with_grad) {
    for (int t = 0; t < ntri; ++t) {
      int I, J; tri_decode(t, I, J);
      fz_grad2_8<KID>(s.A, Xb, rawb, alphab, n, N, I, J, s.stats2 + 2 * t, sm);
      __syncthreads();
    }
  }
  FTR_ADD(8);
#ifdef GPSR_FUSED_TRACE
  if (tid == 0) { atomicAdd(&g_fused_trace[0], 1ull); atomicAdd(&g_fused_trace[9], FTR_T() - ftr_t0); }
#endif
}

// grid = slots (<= 2 per CU), 448 threads, dynamic LDS = PIVOT_LDS_BYTES.  Iterations it0 .. it0+nit-1 (0-based) of the
// Adam loop, then (final_eval) the evaluation at the final hyper-parameters that leaves alpha.  nblk >= 2, d == 8.
template <int KID>
__global__ void __launch_bounds__(MM_THREADS, 4)
k_fit_fused(const double* __restrict__ X, const double* __restrict__ y, float* raw, double* alpha, double* loss,
            int32_t* flags, float* m1g, float* m2g, int B, int n, int N, int nit, int final_eval, AdamTab tab,
            FusedWs w, int pair_steps, int stagger, float one_minus_b1, float b2, float one_minus_b2, float eps) {
  extern __shared__ __attribute__((aligned(16))) double fsm[];
  __shared__ float mom[8];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int nblk = N / NB, ntri = nblk * (nblk + 1) / 2, ntri1 = (nblk - 1) * nblk / 2;
  const int slot = blockIdx.x;
  FusedWs s;
  s.A = w.A + (size_t)slot * N * N;
  s.Wo = w.Wo + (size_t)slot * NB * N;
  s.Wo2 = w.Wo2 + (size_t)slot * NB * N;
  s.U0 = w.U0 + (size_t)slot * NB * NB; s.M0 = w.M0 + (size_t)slot * NB * NB;
  s.U1 = w.U1 + (size_t)slot * NB * NB; s.M1 = w.M1 + (size_t)slot * NB * NB;
  s.stats1 = w.stats1 + (size_t)slot * nblk * 4;
  s.stats2 = w.stats2 + (size_t)slot * ntri * 2;
  s.logdet = w.logdet + slot;
  s.part = w.part + (size_t)slot * (ntri1 + nblk - 1) * (2 * NB);
  // the two workgroups of a CU would otherwise run in lock step (same phase = same bottleneck at the same time):
  // the second half of the grid starts `stagger` x ~3.4 us late
  if (blockIdx.x >= (gridDim.x + 1) / 2)
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const double* Xb = X + (size_t)b * n * 8;
    const double* yb = y + (size_t)b * n;
    float* rawb = raw + 4 * (size_t)b;
    double* alphab = alpha + (size_t)b * N;
    if (tid < 4) { mom[tid] = m1g[4 * (size_t)b + tid]; mom[4 + tid] = m2g[4 * (size_t)b + tid]; }
    __syncthreads();
    for (int e = 0; e < nit + final_eval; ++e) {
      const bool with_grad = e < nit;
      fused_eval<KID>(Xb, yb, rawb, alphab, flags + b, n, N, nblk, with_grad, pair_steps, s, fsm, tid, wave, lane);
      if (with_grad && tid == 0)
        adam_gp(rawb, mom, mom + 4, s.stats1, s.stats2, *s.logdet, n, nblk, ntri, loss ? loss + b : nullptr, nullptr, 1,
                tab.step[e], tab.bc2s[e], one_minus_b1, b2, one_minus_b2, eps);
      __syncthreads();
    }
    if (tid < 4) { m1g[4 * (size_t)b + tid] = mom[tid]; m2g[4 * (size_t)b + tid] = mom[4 + tid]; }
    __syncthreads();
  }
}

// ---------------------------------------------------------------- K5 predict
// grid (ceil(m / (256*TP)), B), 256 threads, dynamic LDS = n*9 doubles.  d == 8.
#define PRED_TP 4
template <int KID>
__global__ void __launch_bounds__(256)
k_predict(const double* __restrict__ X, const double* __restrict__ Xs, const double* __restrict__ alpha,
          const float* __restrict__ raw, int n, int N, int m, double* __restrict__ out) {
  extern __shared__ double sm[];
  double* xs = sm; double* al = sm + (size_t)n * 8;
  const int b = blockIdx.y, tid = threadIdx.x;
  const double* Xb = X + (size_t)b * n * 8;
  for (int e = tid; e < n * 8; e += 256) xs[e] = Xb[e];
  for (int e = tid; e < n; e += 256) al[e] = alpha[(size_t)b * N + e];
  const Hyp h = load_hyp(raw + 4 * b);
  __syncthreads();
  const int base = blockIdx.x * 256 * PRED_TP;
  double q[PRED_TP][8];
  double acc[PRED_TP];
  const double* Xsb = Xs + (size_t)b * m * 8;
#pragma unroll
  for (int u = 0; u < PRED_TP; ++u) {
    int t = base + u * 256 + tid;
    acc[u] = 0.0;
#pragma unroll
    for (int f = 0; f < 8; ++f) q[u][f] = t < m ? Xsb[(size_t)t * 8 + f] : 0.0;
  }
  for (int i = 0; i < n; ++i) {
    double xr[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) xr[f] = xs[i * 8 + f];
    const double a = al[i];
#pragma unroll
    for (int u = 0; u < PRED_TP; ++u) {
      double st = 0.0;
#pragma unroll
      for (int f = 0; f < 8; ++f) {
        if (KID == K_LIN) st = fma(q[u][f], xr[f], st);
        else { double dlt = q[u][f] - xr[f]; st = fma(dlt, dlt, st); }
      }
      acc[u] = fma(kern_val<KID>(st, h), a, acc[u]);
    }
  }
#pragma unroll
  for (int u = 0; u < PRED_TP; ++u) {
    int t = base + u * 256 + tid;
    if (t < m) out[(size_t)b * m + t] = h.mean + h.os * acc[u];
  }
}

// ---------------------------------------------------------------- K6 combine
// One thread per HR pixel, one workgroup per 8 x 32 pixel tile.  The workgroup first marks, in two
// LDS bit masks, the patch rows / patch columns whose window touches its tile (any order of
// rows[] / cols[] is accepted); a pixel then walks only the set bits, in ascending patch index,
// so contributions are added in patch order (row-major over rows x cols) like the reference's loop.
// dynamic LDS = (ceil(nr/32) + ceil(nc/32)) words.
#define CMB_TH 8
#define CMB_TW 32
__global__ void __launch_bounds__(CMB_TH * CMB_TW)
k_combine(const double* __restrict__ pred, const double* __restrict__ tst, int H, int W,
          int s, int p, const int32_t* __restrict__ rows, int nr,
          const int32_t* __restrict__ cols, int nc, int mode, int clip,
          double* __restrict__ out) {
  extern __shared__ uint32_t cmask_sm[];
  const int Hs = H * s, Ws = W * s, P = p * s, ms = P - 2;
  const int nrw = (nr + 31) >> 5, ncw = (nc + 31) >> 5;
  uint32_t* rmask = cmask_sm;
  uint32_t* cmask = cmask_sm + nrw;
  const int tid = threadIdx.x;
  const int R0 = blockIdx.y * CMB_TH, C0 = blockIdx.x * CMB_TW;
  for (int w = tid; w < nrw + ncw; w += CMB_TH * CMB_TW) cmask_sm[w] = 0u;
  __syncthreads();
  for (int e = tid; e < nr; e += CMB_TH * CMB_TW) {
    const int x = rows[e] * s;
    if (x < R0 + CMB_TH && x + P > R0) atomicOr(&rmask[e >> 5], 1u << (e & 31));
  }
  for (int e = tid; e < nc; e += CMB_TH * CMB_TW) {
    const int y = cols[e] * s;
    if (y < C0 + CMB_TW && y + P > C0) atomicOr(&cmask[e >> 5], 1u << (e & 31));
  }
  __syncthreads();
  const int R = R0 + tid / CMB_TW, C = C0 + tid % CMB_TW;
  if (R >= Hs || C >= Ws) return;
  const size_t idx = (size_t)R * Ws + C;
  double acc = 0.0, wsum = 0.0;
  const bool full = (tst == nullptr);  // pred holds whole [P][P] patches, rim included
  const double pix = full ? 0.0 : tst[idx];
  const size_t m = (size_t)ms * ms;
  for (int wr = 0; wr < nrw; ++wr) {
    uint32_t rb = rmask[wr];
    while (rb) {
      const int ir = (wr << 5) + __ffs(rb) - 1;
      rb &= rb - 1;
      const int x = rows[ir] * s;
      const int i = R - x;
      if (i < 0 || i >= P) continue;
      const int xs_ = (x == 0) ? 0 : 1;
      const int xe = (x + P == Hs) ? P : P - 1;
      if (i < xs_ || i >= xe) continue;
      for (int wc = 0; wc < ncw; ++wc) {
        uint32_t cb = cmask[wc];
        while (cb) {
          const int ic = (wc << 5) + __ffs(cb) - 1;
          cb &= cb - 1;
          const int y = cols[ic] * s;
          const int j = C - y;
          if (j < 0 || j >= P) continue;
          const int ys_ = (y == 0) ? 0 : 1;
          const int ye = (y + P == Ws) ? P : P - 1;
          if (j < ys_ || j >= ye) continue;
          double v;
          if (full)
            v = pred[(size_t)(ir * nc + ic) * P * P + (size_t)i * P + j];
          else if (i >= 1 && i <= P - 2 && j >= 1 && j <= P - 2)
            v = pred[(size_t)(ir * nc + ic) * m + (size_t)(i - 1) * ms + (j - 1)];
          else
            v = pix;
          if (mode == 0) {
            acc = __dadd_rn(acc, v);
            wsum = __dadd_rn(wsum, 1.0);
          } else {
            double vw = 1.0 - fabs(0.5 - (double)i / (double)P);
            double hw = 1.0 - fabs(0.5 - (double)j / (double)P);
            double w = __dmul_rn(vw, hw);
            acc = __dadd_rn(acc, __dmul_rn(v, w));
            wsum = __dadd_rn(wsum, w);
          }
        }
      }
    }
  }
  double r = acc / wsum;
  out[idx] = clip ? fmin(fmax(r, 0.0), 1.0) : r;
}

// ---------------------------------------------------------------- K7 AGPR
// w[f] = os * var * sum_i X[i][f] alpha[i]    (one block, d <= 64 threads active)
__global__ void k_agpr_weights(const double* __restrict__ X, const double* __restrict__ alpha,
                               const float* __restrict__ raw, int n, int d, double* __restrict__ w) {
  const int f = threadIdx.x;
  if (f >= d) return;
  const Hyp h = load_hyp(raw);
  double s = 0.0;
  for (int i = 0; i < n; ++i) s = fma(X[(size_t)i * d + f], alpha[i], s);
  w[f] = h.os * h.ls * s;
}

__global__ void k_agpr_filter(const double* __restrict__ img, int Hs, int Ws, int L,
                              const double* __restrict__ w, const float* __restrict__ raw,
                              double* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Hs * Ws) return;
  const int pad = L / 2;
  const int i = (int)(idx / Ws), j = (int)(idx % Ws);
  if (i < pad || i >= Hs - pad || j < pad || j >= Ws - pad) { out[idx] = img[idx]; return; }
  double s = 0.0;
  for (int a = 0; a < L; ++a)
    for (int c = 0; c < L; ++c)
      s = fma((double)(float)img[(size_t)(i - pad + a) * Ws + (j - pad + c)], w[a * L + c], s);
  out[idx] = (double)raw[1] + s;
}

// ---------------------------------------------------------------- self-test
__global__ void k_selftest_mfma(const double* __restrict__ A, const double* __restrict__ Bm, double* __restrict__ D) {
  const int lane = threadIdx.x;
  const double a = A[(lane & 15) * 4 + (lane >> 4)];   // A[i][k], 16x4 row-major
  const double b = Bm[(lane >> 4) * 16 + (lane & 15)];  // B[k][j], 4x16 row-major
  v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int rr = 0; rr < 4; ++rr) D[((lane >> 4) + 4 * rr) * 16 + (lane & 15)] = acc[rr];
}

// ------------------------------------------------ image stages between the phases
// (cv.resize / cv.GaussianBlur / cv.filter2D restated; tables come from the host)

// uint8 -> float64 / 255   (gprsr.py:91)   and   float64 -> uint8 by truncation (gprsr.py:100)
__global__ void k_u8_to_unit(const uint8_t* __restrict__ src, double* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (double)src[i] / 255.0;
}
__global__ void k_unit_to_u8(const double* __restrict__ src, uint8_t* __restrict__ dst, size_t n, double scale) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (uint8_t)(int)(src[i] * scale);
}

// Separable resampling with T taps per axis (cubic T=4, linear T=2): horizontal pass
// then vertical pass, taps added left to right without contraction -- the order of
// OpenCV's HResize*/VResize* for CV_64F.  xi/yi: source indices (border already
// replicated), xw/yw: float32 coefficients widened to float64.
__global__ void k_resample(const double* __restrict__ src, int Hs, int Ws, double* __restrict__ dst, int Hd,
                           int Wd, const int32_t* __restrict__ xi, const double* __restrict__ xw,
                           const int32_t* __restrict__ yi, const double* __restrict__ yw, int T) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Hd * Wd) return;
  const int r = (int)(idx / Wd), c = (int)(idx % Wd);
  double acc = 0.0;
  for (int k = 0; k < T; ++k) {
    const double* row = src + (size_t)yi[r * T + k] * Ws;
    double h = __dmul_rn(row[xi[c * T]], xw[c * T]);
    for (int j = 1; j < T; ++j) h = __dadd_rn(h, __dmul_rn(row[xi[c * T + j]], xw[c * T + j]));
    const double t = __dmul_rn(h, yw[r * T + k]);
    acc = (k == 0) ? t : __dadd_rn(acc, t);
  }
  dst[idx] = acc;
}

// exact 2x2 decimation (cv.resize INTER_LINEAR with fx=fy=1/2 takes the INTER_AREA fast path)
__global__ void k_decimate2(const double* __restrict__ src, int Ws, double* __restrict__ dst, int Hd, int Wd) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Hd * Wd) return;
  const int r = (int)(idx / Wd), c = (int)(idx % Wd);
  const double* p0 = src + (size_t)(2 * r) * Ws + 2 * c;
  const double* p1 = p0 + Ws;
  dst[idx] = __dmul_rn(__dadd_rn(__dadd_rn(__dadd_rn(p0[0], p0[1]), p1[0]), p1[1]), 0.25);
}

__device__ __forceinline__ int reflect101(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
  return i;
}

// symmetric separable filter, radius h <= 3, BORDER_REFLECT_101, rows then columns
// (cv.GaussianBlur for CV_64F -> sepFilter2D with SymmRow/SymmColumn filters)
__global__ void k_sepfilter_sym(const double* __restrict__ src, int H, int W, double* __restrict__ dst,
                                const double* __restrict__ kern, int h) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W) return;
  const int r = (int)(idx / W), c = (int)(idx % W);
  double rowv[7];
  for (int d = -h; d <= h; ++d) {
    const double* row = src + (size_t)reflect101(r + d, H) * W;
    double t = __dmul_rn(row[c], kern[h]);
    for (int j = 1; j <= h; ++j)
      t = __dadd_rn(t, __dmul_rn(__dadd_rn(row[reflect101(c + j, W)], row[reflect101(c - j, W)]), kern[h + j]));
    rowv[d + h] = t;
  }
  double o = __dmul_rn(rowv[h], kern[h]);
  for (int j = 1; j <= h; ++j) o = __dadd_rn(o, __dmul_rn(__dadd_rn(rowv[h + j], rowv[h - j]), kern[h + j]));
  dst[idx] = o;
}

// 3x3 correlation, BORDER_REFLECT_101, non-zero taps added in row-major order (cv.filter2D)
__global__ void k_filter3x3(const double* __restrict__ src, int H, int W, double* __restrict__ dst,
                            const double* __restrict__ kern) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W) return;
  const int r = (int)(idx / W), c = (int)(idx % W);
  double o = 0.0;
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      const double kv = kern[a * 3 + b];
      if (kv != 0.0)
        o = __dadd_rn(o, __dmul_rn(kv, src[(size_t)reflect101(r + a - 1, H) * W + reflect101(c + b - 1, W)]));
    }
  dst[idx] = o;
}

// dst = a + alpha * b    /   dst = clip(a, 0, 1)    (IBP loop of agpr.py:199-209, clip agpr.py:188)
__global__ void k_axpy(const double* __restrict__ a, const double* __restrict__ b, double alpha,
                       double* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = __dadd_rn(a[i], __dmul_rn(b[i], alpha));
}

// ------------------------------------------------ uint8 colour stages (cv.cvtColor / cv.resize on CV_8U, restated)
// OpenCV's fixed-point paths: RGB2Gray<uchar> uses the 15-bit table (BY15 3735, GY15 19235, RY15 9798, >> 15);
// RGB2YCrCb_i / YCrCb2RGB_i use the 14-bit yuv_shift constants.  Integer arithmetic: bit-exact by construction.
__device__ __forceinline__ int descale_i(int x, int n) { return (x + (1 << (n - 1))) >> n; }
__device__ __forceinline__ uint8_t sat_u8(int x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }

// gprsr.py:53 / evaluation.py:96: BGR interleaved -> grey
__global__ void k_bgr2gray(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ gray, size_t npx) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx) return;
  const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
  gray[i] = (uint8_t)descale_i(b * 3735 + g * 19235 + r * 9798, 15);
}

// gprsr.py:57-58 (COLOR_BGR2YUV: c_r = 14369 (V), c_b = 8061 (U)) and agpr.py:150 (COLOR_BGR2YCrCb: 11682, 9241):
// BGR interleaved -> three planes Y, Cr-like (from R-Y), Cb-like (from B-Y)
__global__ void k_bgr2ycc(const uint8_t* __restrict__ bgr, uint8_t* __restrict__ Y, uint8_t* __restrict__ CR,
                          uint8_t* __restrict__ CB, size_t npx, int c_r, int c_b) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx) return;
  const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
  const int y = descale_i(b * 1868 + g * 9617 + r * 4899, 14);
  const int half = 128 << 14;
  Y[i] = sat_u8(y);
  CR[i] = sat_u8(descale_i((r - y) * c_r + half, 14));
  CB[i] = sat_u8(descale_i((b - y) * c_b + half, 14));
}

// gprsr.py:62-63 (COLOR_YUV2BGR: 18678, -9519, -6472, 33292) and agpr.py:155-156 (COLOR_YCrCb2BGR: 22987, -11698,
// -5636, 29049): planes -> BGR interleaved
__global__ void k_ycc2bgr(const uint8_t* __restrict__ Y, const uint8_t* __restrict__ CR, const uint8_t* __restrict__ CB,
                          uint8_t* __restrict__ bgr, size_t npx, int c0, int c1, int c2, int c3) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npx) return;
  const int y = Y[i], cr = (int)CR[i] - 128, cb = (int)CB[i] - 128;
  bgr[3 * i] = sat_u8(y + descale_i(cb * c3, 14));
  bgr[3 * i + 1] = sat_u8(y + descale_i(cb * c2 + cr * c1, 14));
  bgr[3 * i + 2] = sat_u8(y + descale_i(cr * c0, 14));
}

// cv.resize INTER_CUBIC on a CV_8U plane (gprsr.py:60-61, agpr.py:152-153, 246: chroma planes, the uint8 downscale) as the
// reference's OpenCV ran it: float32 taps (host tables, not quantised), float32 passes, horizontal then vertical, each
// (t0 w0 + t1 w1) + (t2 w2 + t3 w3) with the first product of a pair fused, ties to even -- the arithmetic that
// reproduces the reference's own uint8 planes pixel for pixel (oracle/cvops.py::_resize_cubic_u8 has the evidence;
// resize.cpp's 11-bit fixed-point code does not: the opencv-python wheels route this call through IPP).
__global__ void k_resample_u8(const uint8_t* __restrict__ src, int Ws, uint8_t* __restrict__ dst, int Hd, int Wd,
                              const int32_t* __restrict__ xi, const float* __restrict__ xw,
                              const int32_t* __restrict__ yi, const float* __restrict__ yw) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)Hd * Wd) return;
  const int r = (int)(idx / Wd), c = (int)(idx % Wd);
  const float w0 = xw[4 * c], w1 = xw[4 * c + 1], w2 = xw[4 * c + 2], w3 = xw[4 * c + 3];
  float h[4];
  for (int k = 0; k < 4; ++k) {
    const uint8_t* row = src + (size_t)yi[4 * r + k] * Ws;
    const float t0 = (float)row[xi[4 * c]], t1 = (float)row[xi[4 * c + 1]], t2 = (float)row[xi[4 * c + 2]],
                t3 = (float)row[xi[4 * c + 3]];
    h[k] = __builtin_fmaf(t0, w0, t1 * w1) + __builtin_fmaf(t2, w2, t3 * w3);   // nothing else contracts (-ffp-contract=off)
  }
  const float f = __builtin_fmaf(h[0], yw[4 * r], h[1] * yw[4 * r + 1]) + __builtin_fmaf(h[2], yw[4 * r + 2], h[3] * yw[4 * r + 3]);
  const float v = rintf(f);
  dst[idx] = (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));
}

// cv.GaussianBlur on a CV_8U plane (agpr.py:245 on the uint8 HR image, agpr.py:286-287): OpenCV's bit-exact fixed-point
// path -- 8-bit taps (host: getGaussianKernelFixedPoint_ED, sum 256), horizontal pass exact in 8.8, vertical in 16.16,
// one rounding (half up) at the end, BORDER_REFLECT_101.  Integer arithmetic: any evaluation order gives the same bits.
__global__ void k_sepfilter_u8(const uint8_t* __restrict__ src, int H, int W, uint8_t* __restrict__ dst,
                               const int32_t* __restrict__ kern, int ksize) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)H * W) return;
  const int r = (int)(idx / W), c = (int)(idx % W), rad = ksize / 2;
  int v = 0;
  for (int a = 0; a < ksize; ++a) {
    const uint8_t* row = src + (size_t)reflect101(r + a - rad, H) * W;
    int h = 0;
    for (int b = 0; b < ksize; ++b) h += kern[b] * (int)row[reflect101(c + b - rad, W)];
    v += kern[a] * h;
  }
  dst[idx] = (uint8_t)((v + (1 << 15)) >> 16);
}

// ------------------------------------------------ PSNR / SSIM (evaluation.py:83-98, skimage.metrics restated)
// Two uint8 grey images, cropped by `crop` px on every side (Evaluator.preprocess_image).  PSNR needs the sum of
// squared errors: integers, exact in 64 bits whatever the order.  SSIM = mean over all 7x7 windows that lie inside
// the cropped image of S(window) (skimage: uniform filter, then crop 3 px = exactly those windows); the five
// window sums are integers too, S is float64, and the mean is reduced in a fixed order (per-workgroup tree ->
// one partial per workgroup -> k_metrics_final sums them serially per lane, then a tree): deterministic.
#define MET_THREADS 256
__global__ void __launch_bounds__(MET_THREADS)
k_metrics_partial(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int W, int crop, int Hc, int Wc,
                  unsigned long long* __restrict__ sse_part, double* __restrict__ ssim_part) {
  __shared__ unsigned long long sq[MET_THREADS];
  __shared__ double ss[MET_THREADS];
  const int tid = threadIdx.x;
  const size_t idx = (size_t)blockIdx.x * MET_THREADS + tid;
  unsigned long long e2 = 0;
  double S = 0.0;
  if (idx < (size_t)Hc * Wc) {
    const int i = (int)(idx / Wc), j = (int)(idx % Wc);
    const uint8_t* pa = a + (size_t)(crop + i) * W + crop + j;
    const uint8_t* pb = b + (size_t)(crop + i) * W + crop + j;
    const int d = (int)pa[0] - (int)pb[0];
    e2 = (unsigned long long)(d * d);
    if (i + 7 <= Hc && j + 7 <= Wc) {   // window with top-left corner (i, j)
      int sa = 0, sb = 0, saa = 0, sbb = 0, sab = 0;
      for (int u = 0; u < 7; ++u)
        for (int v = 0; v < 7; ++v) {
          const int x = pa[(size_t)u * W + v], y = pb[(size_t)u * W + v];
          sa += x; sb += y; saa += x * x; sbb += y * y; sab += x * y;
        }
      const double ux = sa / 49.0, uy = sb / 49.0, uxx = saa / 49.0, uyy = sbb / 49.0, uxy = sab / 49.0;
      const double cn = 49.0 / 48.0;
      const double vx = cn * (uxx - ux * ux), vy = cn * (uyy - uy * uy), vxy = cn * (uxy - ux * uy);
      const double C1 = (0.01 * 255) * (0.01 * 255), C2 = (0.03 * 255) * (0.03 * 255);
      S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
  }
  sq[tid] = e2; ss[tid] = S;
  __syncthreads();
  for (int o = MET_THREADS / 2; o > 0; o >>= 1) {
    if (tid < o) { sq[tid] += sq[tid + o]; ss[tid] += ss[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) { sse_part[blockIdx.x] = sq[0]; ssim_part[blockIdx.x] = ss[0]; }
}

// out[0] = PSNR (dB, data_range 255; +inf for identical images), out[1] = mean SSIM, out[2] = MSE
__global__ void __launch_bounds__(MET_THREADS)
k_metrics_final(const unsigned long long* __restrict__ sse_part, const double* __restrict__ ssim_part, int nparts,
                double npix, double nwin, double* __restrict__ out) {
  __shared__ unsigned long long sq[MET_THREADS];
  __shared__ double ss[MET_THREADS];
  const int tid = threadIdx.x;
  unsigned long long e2 = 0;
  double S = 0.0;
  for (int p = tid; p < nparts; p += MET_THREADS) { e2 += sse_part[p]; S += ssim_part[p]; }
  sq[tid] = e2; ss[tid] = S;
  __syncthreads();
  for (int o = MET_THREADS / 2; o > 0; o >>= 1) {
    if (tid < o) { sq[tid] += sq[tid + o]; ss[tid] += ss[tid + o]; }
    __syncthreads();
  }
  if (tid == 0) {
    const double mse = (double)sq[0] / npix;
    out[0] = 10.0 * log10(255.0 * 255.0 / mse);
    out[1] = nwin > 0 ? ss[0] / nwin : 0.0;
    out[2] = mse;
  }
}

// ------------------------------------------------ AGPR training-set selection (agpr.py:120-137, 211-222)
// The reference builds an [N][N] matrix of squared distances between candidate rows (bicubic 7x7 patch + HR centre
// pixel, D = 50), scores every row by the mean of exp(-d / 2bw) over its K nearest neighbours ("representativeness"),
// then picks floor(ratio * N) rows greedily: argmax of w_rep * rep + w_div * div, div = min over the rows chosen so far
// of -exp(-d / 2bw).  Two habits of the reference are kept because they decide which rows are chosen: rep is indexed
// by a row's POSITION IN THE SHRINKING POOL (np.delete shifts the pool, the score array stays), and the sums run in
// numpy's order (np_sum below), so the distances -- and through them every tie -- are those numpy produces.
//
// numpy's float64 add.reduce over a contiguous axis of n <= 128 elements (pairwise_sum's unrolled block): eight
// interleaved partial sums, a fixed tree over them, the n % 8 tail added one by one; plain loop below 8 elements.
template <class F>
__device__ __forceinline__ double np_sum(int n, F term) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += term(i);
    return r;
  }
  double r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = term(j);
  int i = 8;
  for (; i < n - (n % 8); i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] += term(i + j);
  }
  double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
  for (; i < n; ++i) res += term(i);
  return res;
}

// dist[i][j] = sum_k (x_ik - x_jk)^2 (agpr.py:123), +inf on the diagonal (:124).  16 x 16 pairs per workgroup, the
// 2 x 16 rows staged in LDS (row length padded to an odd number of doubles: conflict-free column walks).
#define AS_MAXD 128
__global__ void __launch_bounds__(256)
k_as_dist(const double* __restrict__ data, int N, int D, double* __restrict__ dist) {
  extern __shared__ double as_rows[];
  const int ld = D | 1;
  double* ri = as_rows;
  double* rj = as_rows + 16 * ld;
  const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16;
  for (int e = threadIdx.x; e < 16 * D; e += 256) {
    const int r = e / D, k = e - r * D;
    ri[r * ld + k] = (i0 + r < N) ? data[(size_t)(i0 + r) * D + k] : 0.0;
    rj[r * ld + k] = (j0 + r < N) ? data[(size_t)(j0 + r) * D + k] : 0.0;
  }
  __syncthreads();
  const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
  const int i = i0 + ti, j = j0 + tj;
  if (i >= N || j >= N) return;
  const double* a = ri + ti * ld;
  const double* b = rj + tj * ld;
  const double s = np_sum(D, [&](int k) { const double t = a[k] - b[k]; return t * t; });
  dist[(size_t)i * N + j] = (i == j) ? __builtin_inf() : s;
}

// rep[i] = mean over the K smallest entries of row i of exp(-d / (2 bw)) (agpr.py:126-128): K rounds of "smallest
// (value, column) pair above the previous one", so equal distances are taken in column order (their values, and
// with them the mean, do not depend on the order numpy's unstable argsort would give them).
#define AS_MAXK 64
__global__ void __launch_bounds__(256)
k_as_rep(const double* __restrict__ dist, int N, int K, double two_bw, double* __restrict__ rep) {
  __shared__ double sv[256];
  __shared__ int sc[256];
  __shared__ double nd[AS_MAXK];
  const int i = blockIdx.x, tid = threadIdx.x;
  const double* row = dist + (size_t)i * N;
  double pv = -1.0;   // distances are >= 0: nothing is excluded in the first round
  int pc = -1;
  for (int r = 0; r < K; ++r) {
    double bv = __builtin_inf();
    int bc = 0x7fffffff;
    for (int c = tid; c < N; c += 256) {
      const double v = row[c];
      const bool after = v > pv || (v == pv && c > pc);
      if (after && (v < bv || (v == bv && c < bc))) { bv = v; bc = c; }
    }
    sv[tid] = bv; sc[tid] = bc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if (tid < o) {
        const double v = sv[tid + o];
        const int c = sc[tid + o];
        if (v < sv[tid] || (v == sv[tid] && c < sc[tid])) { sv[tid] = v; sc[tid] = c; }
      }
      __syncthreads();
    }
    pv = sv[0]; pc = sc[0];
    if (tid == 0) nd[r] = pv;
    __syncthreads();
  }
  if (tid == 0) {
    const double s = np_sum(K, [&](int k) { return exp(-nd[k] / two_bw); });
    rep[i] = s / (double)K;
  }
}

// The greedy loop (agpr.py:132-137) as ONE workgroup: the pool lives in global memory as orig[pos] (row index) and
// dv[pos] (running diversity), both shifted left past the chosen position every round exactly as np.delete does.
#define AS_THREADS 1024
__global__ void __launch_bounds__(AS_THREADS)
k_as_greedy(const double* __restrict__ dist, const double* __restrict__ rep, int N, int n_pick, double two_bw,
            double w_rep, double w_div, int32_t* __restrict__ orig, double* __restrict__ dv,
            int32_t* __restrict__ out_idx) {
  __shared__ double sv[AS_THREADS];
  __shared__ int sp[AS_THREADS];
  const int tid = threadIdx.x;
  for (int p = tid; p < N; p += AS_THREADS) { orig[p] = p; dv[p] = 0.0; }   // get_diversity: 0 while nothing is chosen
  __syncthreads();
  for (int it = 0; it < n_pick; ++it) {
    const int len = N - it;
    // np.argmax: the first position holding the maximum
    double bv = -__builtin_inf();
    int bp = 0x7fffffff;
    for (int p = tid; p < len; p += AS_THREADS) {
      const double s = w_rep * rep[p] + w_div * dv[p];
      if (s > bv) { bv = s; bp = p; }
    }
    sv[tid] = bv; sp[tid] = bp;
    __syncthreads();
    for (int o = AS_THREADS / 2; o > 0; o >>= 1) {
      if (tid < o) {
        const double v = sv[tid + o];
        const int p = sp[tid + o];
        if (v > sv[tid] || (v == sv[tid] && p < sp[tid])) { sv[tid] = v; sp[tid] = p; }
      }
      __syncthreads();
    }
    int j = sp[0];
    if (j >= len) j = 0;   // every score NaN: np.argmax answers with the first NaN; position 0 keeps the loop finite
    const int pick = orig[j];
    __syncthreads();        // everyone has read sp[0] and orig[j] before the pool moves
    if (tid == 0) out_idx[it] = pick;
    const double* drow = dist + (size_t)pick * N;
    // np.delete + the new minimum, in rounds of AS_THREADS positions: a round reads positions q (+1 past j) and
    // writes positions q; its reads of q + 1 all happen before the barrier, the next round's writes after it
    for (int base = 0; base < len - 1; base += AS_THREADS) {
      const int p = base + tid;
      int o = 0;
      double d = 0.0;
      if (p < len - 1) {
        const int q = p + (p >= j ? 1 : 0);
        o = orig[q];
        const double cand = -exp(-drow[o] / two_bw);
        d = (it == 0) ? cand : fmin(dv[q], cand);
      }
      __syncthreads();
      if (p < len - 1) { orig[p] = o; dv[p] = d; }
      __syncthreads();
    }
  }
}
