// Shared-library form of the synthetic aggressor of tools/probes/pk_f32_probe.hip (an LDS-DMA fill loop, optionally with MFMAs between the
// DMAs), so that tools/debug/gn_hazard_diag.py can run it beside the REAL GroupNorm kernel:  which side carries the trigger of DESIGN.md 3.3?
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probes/dma_writer_lib.hip -o tools/probes/libdma_writer.so
#include <hip/hip_runtime.h>
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds_off)), "v"(g) : "memory", "m0");
}
__global__ __launch_bounds__(256, 2) void writer(const uint4* __restrict__ src, int lds_bytes, int iters, int mode, float* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const int ninstr = lds_bytes / 1024;
  f32x4 acc[8];
  for (int i = 0; i < 8; i++) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  uint4 a = src[lane], b = src[lane + 64];
  for (int it = 0; it < iters; it++) {
    for (int i = (int)wave_u; i < ninstr; i += 4) {
      if (mode & 1) dma16(src + ((it * 131 + i * 64 + lane) & 4095), lds0 + i * 1024);          // bit 0: LDS-DMA fill
      else *(uint4*)(smem + i * 1024 + lane * 16) = src[(it * 131 + i * 64 + lane) & 4095];      //        else load + ds_write (register staging)
      if (mode & 2) {                                                                            // bit 1: MFMAs between the fills
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[k], 0, 0, 0);
      }
      if (mode & 4) { const uint4 r = *(const uint4*)(smem + ((i * 1024 + lane * 16 + 4096) % lds_bytes)); a.x ^= r.x & 1u; }   // bit 2: LDS reads of the ring
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s + (float)a.x;
}
static uint4* g_src = nullptr; static float* g_sink = nullptr;
extern "C" int dma_writer_launch(void* stream, int lds_kb, int iters, int mode, int nblocks) {
  if (!g_src) {
    if (hipMalloc(&g_src, 4096 * 16) != hipSuccess || hipMemset(g_src, 0x3C, 4096 * 16) != hipSuccess || hipMalloc(&g_sink, 16) != hipSuccess) return -1;
    if (hipFuncSetAttribute((const void*)writer, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return -2;
  }
  hipLaunchKernelGGL(writer, dim3(nblocks), dim3(256), lds_kb * 1024, (hipStream_t)stream, g_src, lds_kb * 1024, iters, mode, g_sink);
  return hipGetLastError() == hipSuccess ? 0 : -3;
}
