// chol16_inv (the 16x16 factor + inverse on the serial chain of k_pivot) in isolation: one wave, ITER dependent calls,
// cycles per call, and a check of the result against a host Cholesky + inverse.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -I super_resolution_amd/csrc -o /tmp/chol16_bench tools/chol16_bench.hip
#include "gpsr_kernels.hpp"
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(64) k_bench(const double* D, double* Mout, unsigned long long* cyc, int iters, double* ldout) {
  const int lane = threadIdx.x, lr = lane & 15, lk = lane >> 4;
  double a0[4], a[4], m[4] = {0, 0, 0, 0};
  for (int c = 0; c < 4; ++c) a0[c] = D[(lk + 4 * c) * 16 + lr];
  double ld = 0.0;
  int bad = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    for (int c = 0; c < 4; ++c) a[c] = a0[c] + 1e-300 * m[c];   // dependent on the previous call: nothing overlaps
    chol16_inv(a, m, lane, &ld, &bad);
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  for (int c = 0; c < 4; ++c) Mout[(lk + 4 * c) * 16 + lr] = m[c];
  if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = (unsigned long long)bad; ldout[0] = ld / iters; }
}

int main() {
  const int n = 16, iters = 2000;
  std::vector<double> D(n * n), L(n * n, 0.0), Mi(n * n, 0.0), Mg(n * n);
  // a Matern-like SPD block with a small nugget: condition ~1e5
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      const double d = std::fabs((double)(i - j)) * 0.15;
      D[i * n + j] = (1.0 + 2.236 * d + 1.6667 * d * d) * std::exp(-2.236 * d) + (i == j ? 1e-4 : 0.0);
    }
  double ldref = 0.0;
  for (int j = 0; j < n; ++j) {
    double s = D[j * n + j];
    for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
    L[j * n + j] = std::sqrt(s); ldref += std::log(s);
    for (int i = j + 1; i < n; ++i) {
      double t = D[i * n + j];
      for (int k = 0; k < j; ++k) t -= L[i * n + k] * L[j * n + k];
      L[i * n + j] = t / L[j * n + j];
    }
  }
  for (int j = 0; j < n; ++j) {   // Mi = L^-1
    Mi[j * n + j] = 1.0 / L[j * n + j];
    for (int i = j + 1; i < n; ++i) {
      double t = 0.0;
      for (int k = j; k < i; ++k) t += L[i * n + k] * Mi[k * n + j];
      Mi[i * n + j] = -t / L[i * n + i];
    }
  }
  double *dD, *dM, *dld; unsigned long long* dc;
  CK(hipMalloc(&dD, n * n * 8)); CK(hipMalloc(&dM, n * n * 8)); CK(hipMalloc(&dc, 16)); CK(hipMalloc(&dld, 8));
  CK(hipMemcpy(dD, D.data(), n * n * 8, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k_bench, dim3(1), dim3(64), 0, 0, dD, dM, dc, iters, dld);
    CK(hipDeviceSynchronize());
  }
  unsigned long long c[2]; double ld;
  CK(hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost)); CK(hipMemcpy(Mg.data(), dM, n * n * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(&ld, dld, 8, hipMemcpyDeviceToHost));
  double err = 0.0, ref = 0.0;
  for (int i = 0; i < n * n; ++i) { err = std::fmax(err, std::fabs(Mg[i] - Mi[i])); ref = std::fmax(ref, std::fabs(Mi[i])); }
  printf("chol16_inv: %.0f cycles per call (s_memtime ticks), bad=%llu, max |M - ref| / max |ref| = %.3e, logdet %.12f (ref %.12f)\n",
         (double)c[0] / iters, c[1], err / ref, ld, ldref);
  return 0;
}
