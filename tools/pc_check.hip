// Hardware check of the producer/consumer chunk pipeline of the update kernels: one wave streams chunks
// global -> LDS by global_load_lds into a ring of NBUF buffers and never waits for them: the "landed" flag
// of a chunk is itself written by a trailing 4-byte global_load_lds (loads of a wave return in order), from
// a table of sequence numbers.  Seven waves read every chunk and compare against the known pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#define NB 112
#define KC 8
#define NBT 7
#define NBUF 4
#define DMA_CHUNK (KC * NB)
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef volatile __attribute__((address_space(3))) int* pc_flag_t;
__device__ const int pc_seq[64] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32,
                                   33, 34, 35, 36, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64};

__device__ __forceinline__ void pc_wait(pc_flag_t f, int need) {
  int spins = 0;
  while (*f < need && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void pc_issue(const double* Aop, int lda, int chunk, double* bufA, int* flag, int lane) {
#pragma unroll
  for (int piece = 0; piece < NBT; ++piece) {     // KC*NB*8 B = 7 KiB = 7 wave-instructions of 1 KiB
    const int e = piece * 128 + lane * 2;
    const int kk = e / NB, c = e % NB;
    const double* ga = Aop + (size_t)(chunk * KC + kk) * lda + c;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)ga,
                                     (__attribute__((address_space(3))) void*)(bufA + piece * 128), 16, 0, 0);
  }
  if (lane == 0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(pc_seq + chunk),
                                     (__attribute__((address_space(3))) void*)flag, 4, 0, 0);
}

__global__ void __launch_bounds__(512) k_check(const double* G, int lda, int chunks, int* bad, int mode) {
  __shared__ __attribute__((aligned(16))) double lds[NBUF * DMA_CHUNK];
  __shared__ int flg[2 * NBUF];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  G += (size_t)blockIdx.x * chunks * KC * lda;      // every workgroup streams its own 1.4 MB: HBM misses, while the
  const double base = (double)((size_t)blockIdx.x * chunks * KC * lda);   // flag loads hit the cache
  if (tid < 2 * NBUF) flg[tid] = 0;
  __syncthreads();
  pc_flag_t vf = (pc_flag_t)flg;
  if (wave == NBT) {
    for (int g = 0; g < chunks; ++g) {
      const int b = g % NBUF;
      if (g >= NBUF) pc_wait(vf + NBUF + b, NBT * (g / NBUF));   // the seven readers have left chunk g - NBUF
      pc_issue(G, lda, g, lds + b * DMA_CHUNK, flg + b, lane);
    }
    return;     // outstanding loads still complete after s_endpgm?  checked by the readers below
  }
  int nbad = 0;
  for (int g = 0; g < chunks; ++g) {
    const int b = g % NBUF;
    pc_wait(vf + b, g + 1);
    const double* cur = lds + b * DMA_CHUNK;
    for (int e = lane; e < DMA_CHUNK; e += 64) {
      const int kk = e / NB, c = e % NB;
      const double want = base + (double)((g * KC + kk) * lda + c);
      if (cur[e] != want) ++nbad;
    }
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(flg + NBUF + b, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  if (mode == 1) __syncthreads();   // barrier among the seven survivors
  if (nbad) atomicAdd(bad, nbad);
  if (mode == 1 && tid == 0) atomicAdd(bad + 1, 1);
}

__global__ void k_fill(double* d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = (double)i;
}

int main() {
  const int lda = 784, chunks = 28, rows = chunks * KC;
  const int nwg = 4096;
  const size_t total = (size_t)nwg * rows * lda;
  double* d; int* bad;
  CK(hipMalloc(&d, sizeof(double) * total)); CK(hipMalloc(&bad, 8));
  hipLaunchKernelGGL(k_fill, dim3(65536), dim3(256), 0, 0, d, total);
  CK(hipDeviceSynchronize());
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(bad, 0, 8));
    hipLaunchKernelGGL(k_check, dim3(nwg), dim3(512), 0, 0, d, lda, chunks, bad, mode);
    CK(hipDeviceSynchronize());
    int hb[2]; CK(hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost));
    printf("mode %d: mismatching elements %d, workgroups past the survivor barrier %d\n", mode, hb[0], hb[1]);
  }
  return 0;
}
