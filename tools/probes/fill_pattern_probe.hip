// Probe: L2 -> LDS fill rate per CU (LDS-DMA, 2 blocks x 4 waves per CU, no MFMA work) as a function of the ACCESS PATTERN of one
// 1 KB wave-instruction: `seg` contiguous bytes per row, 1024 / seg rows, rows `ld` bytes apart -- the tile shapes of gemm.hip
// (weight gradient: 256-B / 128-B row segments of [rows][channels] activations; conv forward: 64-B segments).  Each block streams its
// own column panel of a [rows][ld] matrix like a GEMM block does; `share` blocks read the same panel (L2 reuse).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_void_ptr;
__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(g) : "memory", "m0");
}
// stage = 16 instructions (16 KB) per block, double buffered: issue stage s+1, wait stage s (vmcnt counted), barrier
__global__ __launch_bounds__(256, 2) void fill(const char* __restrict__ src, long rows, int ld, int seg, int share, int stages, unsigned* sink, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 x 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int rpi = 1024 / seg;                 // rows per instruction
  const int panels = ld / seg;                // column panels
  // blocks sharing a panel run on the same XCD (blockIdx % 8), as the XCD-aware tile mapping of gemm.hip arranges
  const int b = (blockIdx.x % 8) * (64 / share) + (blockIdx.x / 8) / share;
  const int panel = b % panels;
  const long rows_per_block = (long)stages * 16 * rpi;
  long row0 = ((long)(b / panels) * rows_per_block) % (rows - rows_per_block);
  const int lrow = lane / (seg / 16), lcol = (lane % (seg / 16)) * 16;
  const char* base = src + (long)panel * seg + lcol;
  unsigned acc = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
  auto issue = [&](int s, int buf) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const long r = row0 + ((long)s * 16 + wave * 4 + i) * rpi + lrow;
      dma16(base + r * ld, lds0 + buf * 16384 + (wave_u * 4 + i) * 1024);
    }
  };
  issue(0, 0);
  for (int s = 0; s < stages; s++) {
    if (s + 1 < stages) { issue(s + 1, (s + 1) & 1); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc += *(const unsigned*)(smem + (s & 1) * 16384 + ((tid * 52 + s) & 0x3ffc));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 0x12345678u) sink[0] = acc;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const long rows = 196608; const int ldmax = 2048;
  char* src; unsigned* sink; unsigned long long* cyc;
  CHECK(hipMalloc(&src, rows * ldmax)); CHECK(hipMemset(src, 1, rows * ldmax)); CHECK(hipMalloc(&sink, 16)); CHECK(hipMalloc(&cyc, 512 * 8));
  CHECK(hipFuncSetAttribute((const void*)fill, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int blocks = 512, stages = 96;
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  struct Cfg { int ld, seg, share; const char* name; };
  const Cfg cfgs[] = {
    {1024, 1024, 8, "contiguous 1 KB rows, 8 share"}, {1024, 1024, 64, "contiguous 1 KB rows, 64 share"},
    {1024, 256, 8, "256-B segments, ld 1 KB (wgrad dY, 512 ch), 8 share"}, {1024, 256, 64, "256-B segments, ld 1 KB, 64 share"},
    {1024, 128, 8, "128-B segments, ld 1 KB (wgrad X, 512 ch), 8 share"}, {1024, 128, 64, "128-B segments, ld 1 KB, 64 share"},
    {1024, 64, 8, "64-B segments, ld 1 KB (conv A, K32), 8 share"}, {1024, 64, 64, "64-B segments, ld 1 KB, 64 share"},
    {2048, 256, 8, "256-B segments, ld 2 KB, 8 share"}, {256, 256, 8, "256-B segments, ld 256 B (128 ch), 8 share"},
    {512, 128, 8, "128-B segments, ld 512 B (256 ch), 8 share"}, {256, 64, 8, "64-B segments, ld 256 B, 8 share"}, {256, 64, 64, "64-B segments, ld 256 B, 64 share"},
  };
  for (const Cfg& c : cfgs) {
    for (int rep = 0; rep < 2; rep++) {
      CHECK(hipEventRecord(a, 0));
      hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 32768, 0, src, rows, c.ld, c.seg, c.share, stages, sink, cyc);
      CHECK(hipEventRecord(b, 0)); CHECK(hipDeviceSynchronize());
    }
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h(blocks); CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
    double mean = 0; for (auto v : h) mean += (double)v / blocks;
    printf("%-48s %7.1f us  chip %6.2f TB/s  %5.1f B/shader-clk/CU (stage %.0f clk)\n", c.name, ms * 1e3, blocks * (double)stages * 16384 / ms / 1e9, 2.0 * stages * 16384 / mean, mean / stages);
  }
  return 0;
}
