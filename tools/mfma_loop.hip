// Inner loop of k_update in isolation: operands resident in LDS (no global traffic at all),
// 7 waves x (16x112 strip = 7 accumulators), 16-deep chunks, optional barrier per chunk.
// Tells how much of the f64 MFMA peak the LDS-fed loop itself can reach.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define NB 112
#define KC 16
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int WAVES, bool BARRIER, int NACC>
__global__ void __launch_bounds__(64 * WAVES, 4) k_loop(double* out, int chunks) {
  __shared__ double lds[4 * KC * NB];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 4 * KC * NB; e += 64 * WAVES) lds[e] = 1e-3 * (e % 97);
  __syncthreads();
  v4d acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int ch = 0; ch < chunks; ++ch) {
    const double* As = lds + (ch & 1) * 2 * KC * NB;
    const double* Bs = As + KC * NB;
    if (BARRIER) __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
      const int row = (ks * 4 + lk) * NB;
      const double av = As[row + 16 * (wave % 7) + lr];
#pragma unroll
      for (int t = 0; t < NACC; ++t) {
        const double bv = Bs[row + 16 * t + lr];
        acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678) out[0] = s;
}


template <int WAVES, bool BARRIER, int NACC>
int run(double* out, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int chunks = 700, grid = 256 * 2 * 8;
  hipLaunchKernelGGL((k_loop<WAVES, BARRIER, NACC>), dim3(grid), dim3(64 * WAVES), 0, 0, out, 7);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_loop<WAVES, BARRIER, NACC>), dim3(grid), dim3(64 * WAVES), 0, 0, out, chunks);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * WAVES * chunks * 4 * NACC * 2048.0;
  printf("%-34s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
  return 0;
}


// 14 waves per workgroup: each 16-row strip split into two waves (column blocks 0-3 and 4-6)
template <bool BARRIER>
__global__ void __launch_bounds__(896, 7) k_loop14(double* out, int chunks) {
  __shared__ double lds[4 * KC * NB];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 4 * KC * NB; e += 896) lds[e] = 1e-3 * (e % 97);
  __syncthreads();
  const int strip = wave % 7, half = wave / 7;
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int ch = 0; ch < chunks; ++ch) {
    const double* As = lds + (ch & 1) * 2 * KC * NB;
    const double* Bs = As + KC * NB;
    if (BARRIER) __syncthreads();
    if (half == 0) {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        const int row = (ks * 4 + lk) * NB;
        const double av = As[row + 16 * strip + lr];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[row + 16 * t + lr], acc[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        const int row = (ks * 4 + lk) * NB;
        const double av = As[row + 16 * strip + lr];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[row + 16 * (4 + t) + lr], acc[t], 0, 0, 0);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678) out[0] = s;
}

template <bool BARRIER>
int run14(double* out, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int chunks = 700, grid = 256 * 2 * 8;
  hipLaunchKernelGGL((k_loop14<BARRIER>), dim3(grid), dim3(896), 0, 0, out, 7);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_loop14<BARRIER>), dim3(grid), dim3(896), 0, 0, out, chunks);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * 49 * chunks * 4 * 2048.0;
  printf("%-34s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
  return 0;
}


// 14 waves, block-balanced: waves land on SIMD (wave % 4) (checked by k_simd_probe), so SIMD0 holds 4 waves and must
// get 13 of the 49 blocks (4+3+3+3), SIMDs 1..3 get 12 (3+3+3+3 / 4+4+4 / 4+4+4).  Strip s is shared by one
// 4-block wave (column blocks 0-3) and one 3-block wave (4-6).
__constant__ int c_strip[14] = {0, 0, 1, 2, 1, 2, 3, 4, 3, 4, 5, 6, 5, 6};
__constant__ int c_wide[14]  = {1, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0};
template <bool BARRIER>
__global__ void __launch_bounds__(896, 7) k_loop14b(double* out, int chunks) {
  __shared__ double lds[4 * KC * NB];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 4 * KC * NB; e += 896) lds[e] = 1e-3 * (e % 97);
  __syncthreads();
  const int strip = c_strip[wave], wide = c_wide[wave];
  v4d acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int ch = 0; ch < chunks; ++ch) {
    const double* As = lds + (ch & 1) * 2 * KC * NB;
    const double* Bs = As + KC * NB;
    if (BARRIER) __syncthreads();
    if (wide) {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        const int row = (ks * 4 + lk) * NB;
        const double av = As[row + 16 * strip + lr];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[row + 16 * t + lr], acc[t], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        const int row = (ks * 4 + lk) * NB;
        const double av = As[row + 16 * strip + lr];
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[row + 16 * (4 + t) + lr], acc[t], 0, 0, 0);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678) out[0] = s;
}

template <bool BARRIER>
int run14b(double* out, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int chunks = 700, grid = 256 * 2 * 8;
  hipLaunchKernelGGL((k_loop14b<BARRIER>), dim3(grid), dim3(896), 0, 0, out, 7);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL((k_loop14b<BARRIER>), dim3(grid), dim3(896), 0, 0, out, chunks);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * 49 * chunks * 4 * 2048.0;
  printf("%-34s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
  return 0;
}

// which SIMD does wave w of a workgroup land on?  HW_ID (hwreg 4) bits [5:4] = SIMD_ID.
__global__ void k_simd_probe(int* simd) {
  const int wave = threadIdx.x >> 6;
  const unsigned id = __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
  if ((threadIdx.x & 63) == 0) simd[blockIdx.x * (blockDim.x >> 6) + wave] = (int)id;
}

int probe(int waves) {
  const int nblk = 1024;
  int* d; CK(hipMalloc(&d, nblk * waves * sizeof(int)));
  hipLaunchKernelGGL(k_simd_probe, dim3(nblk), dim3(64 * waves), 0, 0, d);
  CK(hipDeviceSynchronize());
  int* h = new int[nblk * waves];
  CK(hipMemcpy(h, d, nblk * waves * sizeof(int), hipMemcpyDeviceToHost));
  int hist[16][4] = {};
  int rr = 0;
  for (int b = 0; b < nblk; ++b) {
    bool ok = true;
    for (int w = 0; w < waves; ++w) { hist[w][h[b * waves + w] & 3]++; if (((h[b * waves + w] - h[b * waves]) & 3) != (w & 3)) ok = false; }
    rr += ok;
  }
  printf("SIMD of wave w over %d workgroups of %d waves (round-robin from wave 0's SIMD in %d of them):\n", nblk, waves, rr);
  for (int w = 0; w < waves; ++w) printf("  wave %2d: %4d %4d %4d %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
  delete[] h; CK(hipFree(d));
  return 0;
}

// Producer/consumer variant: 7 compute waves + 1 wave that would issue the DMA.  No s_barrier in the loop:
// ready[buf] (set by the producer) and done[buf] (ds_add by each consumer) are LDS flags; a consumer never waits
// for another consumer.  Spins are bounded (a stuck flag ends the loop instead of hanging the GPU).
__global__ void __launch_bounds__(512, 4) k_loop_pc(double* out, int chunks) {
  __shared__ double lds[4 * KC * NB];
  __shared__ int ready[2], done[2];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lr = lane & 15, lk = lane >> 4;
  for (int e = tid; e < 4 * KC * NB; e += 512) lds[e] = 1e-3 * (e % 97);
  if (tid < 2) { ready[tid] = 0; done[tid] = 0; }
  __syncthreads();
  volatile int* vready = ready;
  volatile int* vdone = done;
  v4d acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
  if (wave == 7) {
    for (int ch = 0; ch < chunks; ++ch) {
      const int b = ch & 1, need = 7 * (ch >> 1);
      int spins = 0;
      while (vdone[b] < need && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
      // (the DMA of chunk ch into buffer b and its s_waitcnt vmcnt(0) would go here)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) vready[b] = ch + 1;
    }
  } else {
    for (int ch = 0; ch < chunks; ++ch) {
      const int b = ch & 1;
      int spins = 0;
      while (vready[b] < ch + 1 && ++spins < (1 << 20)) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const double* As = lds + b * 2 * KC * NB;
      const double* Bs = As + KC * NB;
#pragma unroll
      for (int ks = 0; ks < KC / 4; ++ks) {
        const int row = (ks * 4 + lk) * NB;
        const double av = As[row + 16 * wave + lr];
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Bs[row + 16 * t + lr], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) atomicAdd(&done[b], 1);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int t = 0; t < 7; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  if (s == 12345.678) out[0] = s;
}

int run_pc(double* out, const char* name) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int chunks = 700, grid = 256 * 2 * 8;
  hipLaunchKernelGGL(k_loop_pc, dim3(grid), dim3(512), 0, 0, out, 7);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_loop_pc, dim3(grid), dim3(512), 0, 0, out, chunks);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flops = (double)grid * 7 * chunks * 4 * 7 * 2048.0;
  printf("%-34s %.3f ms  %.1f TFLOP/s\n", name, ms, flops / ms * 1e-9);
  return 0;
}

int main() {
  double* out; CK(hipMalloc(&out, 8));
  run<7, true, 7>(out, "7 waves, barrier/chunk, 7 acc");
  run<7, false, 7>(out, "7 waves, no barrier, 7 acc");
  run<4, true, 7>(out, "4 waves, barrier/chunk, 7 acc");
  run<4, false, 7>(out, "4 waves, no barrier, 7 acc");
  run<8, true, 7>(out, "8 waves, barrier/chunk, 7 acc");
  run<8, false, 7>(out, "8 waves, no barrier, 7 acc");
  run14<true>(out, "14 half-strip waves, barrier");
  run14<false>(out, "14 half-strip waves, no barrier");
  run14b<true>(out, "14 waves block-balanced, barrier");
  run14b<false>(out, "14 waves block-balanced, no barrier");
  run_pc(out, "7+1 waves, LDS flags, no barrier");
  probe(7);
  probe(14);
  return 0;
}
