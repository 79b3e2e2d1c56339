// Practical f64 MFMA ceiling of the box: waves that do nothing but independent
// v_mfma_f64_16x16x4_f64 chains.  Usage: mfma_peak
// Prints TFLOP/s for several (waves per workgroup, workgroups per CU) shapes so the
// k_update roofline fraction can be judged against what the silicon sustains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(1024) k_peak(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  double* out; CK(hipMalloc(&out, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 4000, NACC = 7;
  const int shapes[][2] = {{4, 1}, {4, 2}, {4, 4}, {7, 1}, {7, 2}, {7, 3}, {8, 1}, {8, 2}, {16, 1}};
  for (auto& sh : shapes) {
    const int waves = sh[0], wgpc = sh[1];
    const int grid = 256 * wgpc * 8;   // 8 rounds of full occupancy
    hipLaunchKernelGGL(k_peak<NACC>, dim3(grid), dim3(64 * waves), 0, 0, out, 10, 1.0, 2.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_peak<NACC>, dim3(grid), dim3(64 * waves), 0, 0, out, iters, 1.0, 2.0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = (double)grid * waves * iters * NACC * 2048.0;
    printf("waves/wg %2d  grid %5d (%d wg/CU/round)  %.3f ms  %.1f TFLOP/s\n", waves, grid, wgpc, ms, flops / ms * 1e-9);
  }
  return 0;
}
