// f64 VALU issue rate of a lone wave on gfx950 (dependent / independent chains, full / partial EXEC) and v_rsq_f64.
// Measured: 7.5 cycles per v_fma_f64 dependent or not, no gain from 16 active lanes, rsq ~17.
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/valu_lat tools/valu_lat.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k(double* out, unsigned long long* cyc, int mode) {
  const int lane = threadIdx.x;
  double x = out[lane], y = 1.0000001, z = 0.5;
  unsigned long long t0 = 0, t1 = 0;
  if (mode == 0) {          // dependent fma chain, full exec
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = fma(x, y, z);
    }
    t1 = __builtin_readcyclecounter();
  } else if (mode == 1) {   // same, 16 lanes active
    if (lane < 16) {
      t0 = __builtin_readcyclecounter();
#pragma unroll 1
      for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) x = fma(x, y, z);
      }
      t1 = __builtin_readcyclecounter();
    }
  } else if (mode == 2) {   // 4 independent chains, full exec (throughput)
    double a = x, b = x + 1, c = x + 2, d = x + 3;
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { a = fma(a, y, z); b = fma(b, y, z); c = fma(c, y, z); d = fma(d, y, z); }
    }
    t1 = __builtin_readcyclecounter();
    x = a + b + c + d;
  } else if (mode == 3) {   // dependent rsq chain
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < 1000; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __builtin_amdgcn_rsq(x + 2.0);
    }
    t1 = __builtin_readcyclecounter();
  } else if (mode == 4) {   // 4 independent chains, 16 lanes
    if (lane < 16) {
      double a = x, b = x + 1, c = x + 2, d = x + 3;
      t0 = __builtin_readcyclecounter();
#pragma unroll 1
      for (int i = 0; i < 1000; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { a = fma(a, y, z); b = fma(b, y, z); c = fma(c, y, z); d = fma(d, y, z); }
      }
      t1 = __builtin_readcyclecounter();
      x = a + b + c + d;
    }
  }
  out[lane] = x;
  if (lane == 0) cyc[0] = t1 - t0;
}
int main() {
  double* d; unsigned long long* c; hipMalloc(&d, 512); hipMalloc(&c, 8); hipMemset(d, 0, 512);
  const char* names[] = {"dependent fma, 64 lanes", "dependent fma, 16 lanes", "4 independent fma chains, 64 lanes", "dependent rsq(+add), 64 lanes", "4 independent fma chains, 16 lanes"};
  for (int m = 0; m < 5; ++m) {
    for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c, m); hipDeviceSynchronize(); }
    unsigned long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
    printf("%-40s %.2f cycles per op\n", names[m], (double)h / 16000.0);
  }
  return 0;
}
