// Probe: can the LDS-DMA writes (global_load_lds_dwordx4) of one workgroup land in the LDS of ANOTHER workgroup on the same CU?
// Background: with 256-thread GroupNorm blocks (7 KB of static LDS, fp64 accumulators) co-resident with the weight-gradient GEMM of
// the side stream, the GroupNorm sums came out wrong -- and right again when that GEMM staged through registers.
//   writer<<<many, 256, W bytes>>>: every wave keeps filling its block's whole dynamic LDS by DMA from a buffer of 0xA5 bytes
//                                   (mode 0: waits for its DMAs before it exits; mode 1: exits with DMAs in flight)
//   victim<<<many, 256>>> (other stream, 8 KB static LDS): writes a pattern, re-reads it for a few microseconds, counts mismatches.
// Prints the mismatch count for several writer LDS sizes (below / above 64 KB).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(g) : "memory", "m0");
}

__global__ __launch_bounds__(256) void writer(const uint4* __restrict__ src, int lds_bytes, int iters, int mode, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const int ninstr = lds_bytes / 1024;                  // 1 KB per wave-instruction
  for (int it = 0; it < iters; it++) {
    for (int i = (int)wave_u; i < ninstr; i += 4) dma16(src + ((it * 131 + i * 64 + lane) & 4095), lds0 + i * 1024);
    if (mode == 0 || it + 1 < iters) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (mode == 0) {
    __syncthreads();
    unsigned bad = 0;
    for (int c = tid; c < lds_bytes / 16; c += 256) { const uint4 v = *(const uint4*)(smem + c * 16); bad += (v.x != 0xA5A5A5A5u) + (v.w != 0xA5A5A5A5u); }
    if (bad) atomicAdd(sink + 1, bad);                 // the writer's own view of its LDS
  }
}

__global__ __launch_bounds__(256) void victim(int spins, unsigned* __restrict__ sink) {
  __shared__ unsigned pat[2048];                        // 8 KB
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += 256) pat[i] = 0x10000u * blockIdx.x + i;
  __syncthreads();
  unsigned bad = 0;
  for (int s = 0; s < spins; s++) {
    for (int i = tid; i < 2048; i += 256) bad += (pat[i] != 0x10000u * blockIdx.x + i);
    __builtin_amdgcn_s_sleep(8);
  }
  if (bad) atomicAdd(sink, bad);
}

// victim 2: the GroupNorm pattern -- fp64 LDS atomics into a few accumulators, then a check of the exact sums
__global__ __launch_bounds__(256) void victim_atomic(int rounds, unsigned* __restrict__ sink) {
  __shared__ double red[896];                           // 7 KB like gn_bwd_resident_kernel
  const int tid = threadIdx.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; r++) {
    for (int i = tid; i < 896; i += 256) red[i] = 0.0;
    __syncthreads();
    for (int k = 0; k < 16; k++) atomicAdd(&red[(tid & 15) * 4 + (k & 3)], (double)(1 + (tid >> 4)));
    __syncthreads();
    // column c = (tid & 15) * 4 + j receives, from each of the 16 row lanes t = tid >> 4, four adds of (1 + t): 4 * (16 * 17 / 2) = 544
    if (tid < 64) bad += (red[tid] != 544.0);
    __syncthreads();
  }
  if (bad) atomicAdd(sink, bad);
}

int main() {
  uint4* src; unsigned* sink;
  CHECK(hipMalloc(&src, 4096 * 16)); CHECK(hipMemset(src, 0xA5, 4096 * 16)); CHECK(hipMalloc(&sink, 8));
  hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
  CHECK(hipFuncSetAttribute((const void*)writer, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  const int sizes[] = {32 * 1024, 60 * 1024, 64 * 1024, 66 * 1024, 67584, 100 * 1024, 150 * 1024};
  for (int mode = 0; mode < 2; mode++)
    for (int si = 0; si < 7; si++) {
      CHECK(hipMemset(sink, 0, 8));
      for (int rep = 0; rep < 20; rep++) {
        hipLaunchKernelGGL(writer, dim3(2048), dim3(256), sizes[si], s1, src, sizes[si], 40, mode, sink);
        hipLaunchKernelGGL(victim, dim3(4096), dim3(256), 0, s2, 200, sink);
      }
      CHECK(hipDeviceSynchronize());
      unsigned h[2]; CHECK(hipMemcpy(h, sink, 8, hipMemcpyDeviceToHost));
      printf("writer LDS %6d bytes, mode %d (%s): victim mismatches %u, writer's own mismatches %u\n", sizes[si], mode,
             mode ? "exits with DMAs in flight" : "waits before exit", h[0], h[1]);
    }
  for (int si = 0; si < 7; si++) {
    CHECK(hipMemset(sink, 0, 8));
    for (int rep = 0; rep < 20; rep++) {
      hipLaunchKernelGGL(writer, dim3(2048), dim3(256), sizes[si], s1, src, sizes[si], 40, 0, sink);
      hipLaunchKernelGGL(victim_atomic, dim3(4096), dim3(256), 0, s2, 50, sink);
    }
    CHECK(hipDeviceSynchronize());
    unsigned h[2]; CHECK(hipMemcpy(h, sink, 8, hipMemcpyDeviceToHost));
    printf("writer LDS %6d bytes beside fp64 LDS atomics: victim wrong sums %u, writer's own mismatches %u\n", sizes[si], h[0], h[1]);
  }
  return 0;
}
