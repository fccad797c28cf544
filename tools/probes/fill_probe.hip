// Probe: sustained L2 -> LDS fill rate per CU for the two staging engines of gemm.hip, on an L2-resident source (a few MB re-read
// by every block), 2 blocks x 4 waves per CU as in the conv kernels, no MFMA work:
//   mode 0: LDS-DMA  (global_load_lds_dwordx4, 64 KB ring per block, vmcnt-throttled)
//   mode 1: register staging (global_load_dwordx4 -> VGPR -> ds_write_b128), 12 loads in flight per thread
//   mode 2: both at once (half the bytes each)
// Prints bytes / clock / CU (wall clock and shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_void_ptr;

__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(g) : "memory", "m0");
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void fill(const uint4* __restrict__ src, long src_chunks, int iters, unsigned* __restrict__ sink, unsigned long long* __restrict__ cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 64 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  // every block walks the same 4 MB window (L2-resident after the first touch), starting at a block-dependent offset
  long pos = ((long)blockIdx.x * 4099) % (src_chunks - 8192);
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
    // one "stage" = 64 KB = 4096 chunks: 16 chunks per thread
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) dma16(src + pos + (wave * 16 + i) * 64 + lane, lds0 + ((wave_u * 16 + i) * 1024));
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (MODE == 1) {
      uint4 r[16];
#pragma unroll
      for (int i = 0; i < 16; i++) r[i] = src[pos + (wave * 16 + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 16; i++) *(uint4*)(smem + (wave * 16 + i) * 1024 + lane * 16) = r[i];
    } else {
      uint4 r[8];
#pragma unroll
      for (int i = 0; i < 8; i++) dma16(src + pos + (wave * 16 + i) * 64 + lane, lds0 + ((wave_u * 16 + i) * 1024));
#pragma unroll
      for (int i = 0; i < 8; i++) r[i] = src[pos + (wave * 16 + 8 + i) * 64 + lane];
#pragma unroll
      for (int i = 0; i < 8; i++) *(uint4*)(smem + (wave * 16 + 8 + i) * 1024 + lane * 16) = r[i];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    acc += *(const unsigned*)(smem + ((tid * 52 + it) & 0xfffc));
    __syncthreads();
    pos += 4096; if (pos + 8192 > src_chunks) pos = 0;
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345678u) sink[0] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE> int run(const uint4* src, long chunks, unsigned* sink, unsigned long long* cyc, const char* name) {
  const int blocks = 512, iters = 200;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipFuncSetAttribute((const void*)fill<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipLaunchKernelGGL(fill<MODE>, dim3(blocks), dim3(256), 65536, 0, src, chunks, 20, sink, cyc);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(fill<MODE>, dim3(blocks), dim3(256), 65536, 0, src, chunks, iters, sink, cyc);
  CHECK(hipEventRecord(b, 0)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h(blocks); CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
  double mean = 0; for (auto v : h) mean += (double)v / blocks;
  const double bytes_per_cu = 2.0 * iters * 65536;         // 2 blocks per CU
  printf("%-28s %7.1f us  chip %6.2f TB/s  %5.1f B/shader-clk/CU (block mean %.0f clk)\n", name, ms * 1e3, blocks * (double)iters * 65536 / ms / 1e9, bytes_per_cu / mean, mean);
  return 0;
}

int main() {
  const long chunks = (4l << 20) / 16 + 8192;
  uint4* src; unsigned* sink; unsigned long long* cyc;
  CHECK(hipMalloc(&src, chunks * 16)); CHECK(hipMemset(src, 1, chunks * 16)); CHECK(hipMalloc(&sink, 16)); CHECK(hipMalloc(&cyc, 512 * 8));
  run<0>(src, chunks, sink, cyc, "LDS-DMA");
  run<1>(src, chunks, sink, cyc, "register staging");
  run<2>(src, chunks, sink, cyc, "half DMA + half registers");
  return 0;
}
