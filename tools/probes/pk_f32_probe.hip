// Probe for the concurrent-kernel hazard of DESIGN.md 3.3 (round 4): are the results of PACKED fp32 VALU instructions (v_pk_fma_f32,
// v_pk_add_f32, v_pk_mul_f32 -- what hipcc's SLP vectoriser makes of adjacent scalar f32 math) of one wave disturbed by LDS-DMA traffic
// (global_load_lds_dwordx4) of OTHER workgroups on the same CU?
// Finding that led here (tools/debug/gn_hazard*.sh): the 256-thread GroupNorm backward returns wrong sums in EVEN channels only (= the
// low lane of every v_pk_*_f32) beside the LDS-DMA weight-gradient GEMM of another stream; built with -fno-slp-vectorize it is exact.
//   victim : every thread runs the same chain twice -- packed (inline asm v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32) and scalar
//            (v_fma_f32 / v_add_f32 / v_mul_f32 on the two halves) -- and counts bitwise mismatches of the low / high lane.
//            mode bit 0: operands come from global loads (dwordx2) instead of registers; mode bit 1: 7 KB of LDS + fp64 LDS atomics beside it
//   writer : keeps filling its dynamic LDS by DMA (optionally with MFMAs between the DMAs), on another stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds_off)), "v"(g) : "memory", "m0");
}

__global__ __launch_bounds__(256, 2) void writer(const uint4* __restrict__ src, int lds_bytes, int iters, int mfma, float* __restrict__ sink) {
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
      dma16(src + ((it * 131 + i * 64 + lane) & 4095), lds0 + i * 1024);
      if (mfma) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[k], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float s = 0.f;
  for (int i = 0; i < 8; i++) s += acc[i][0];
  if (s == 12345.678f) sink[0] = s;
}

__global__ __launch_bounds__(256) void victim(const float2* __restrict__ data, int n_data, int iters, int mode, unsigned* __restrict__ bad) {
  __shared__ double red[896];
  const int tid = threadIdx.x;
  if (mode & 2) { for (int i = tid; i < 896; i += 256) red[i] = 0.0; __syncthreads(); }
  v2f accp = {0.f, 0.f}; float a0 = 0.f, a1 = 0.f;
  const v2f c = {1.0009765625f, 0.99951171875f};
  unsigned bad_lo = 0, bad_hi = 0;
  for (int it = 0; it < iters; it++) {
    v2f x;
    if (mode & 1) { const float2 t = data[(blockIdx.x * 256 + tid + it * 8191) % n_data]; x = v2f{t.x, t.y}; }
    else { const unsigned h = (unsigned)(tid * 2654435761u + it * 40503u); x = v2f{(float)(h & 1023) * 0.001f - 0.5f, (float)((h >> 10) & 1023) * 0.001f - 0.5f}; }
    // packed: acc = fma(x, c, acc); acc = acc + x; acc = acc * c
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n\tv_pk_add_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %2" : "+v"(accp) : "v"(x), "v"(c));
    // scalar twin
    asm volatile("v_fma_f32 %0, %2, %4, %0\n\tv_fma_f32 %1, %3, %5, %1\n\tv_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %3\n\tv_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %5"
                 : "+v"(a0), "+v"(a1) : "v"(x.x), "v"(x.y), "v"(c.x), "v"(c.y));
    if (__float_as_uint(accp.x) != __float_as_uint(a0)) { bad_lo++; accp.x = a0; }
    if (__float_as_uint(accp.y) != __float_as_uint(a1)) { bad_hi++; accp.y = a1; }
    if ((mode & 2) && (it & 15) == 0) atomicAdd(&red[(tid & 15) * 4 + (it & 3)], (double)a0);
  }
  if (bad_lo) atomicAdd(bad, bad_lo);
  if (bad_hi) atomicAdd(bad + 1, bad_hi);
  if ((mode & 2) && red[tid & 63] == 1.2345) atomicAdd(bad + 2, 1u);
}

int main(int argc, char** argv) {
  uint4* src; float2* data; unsigned* bad; float* sink;
  const int n_data = 1 << 22;
  CHECK(hipMalloc(&src, 4096 * 16)); CHECK(hipMemset(src, 0x3C, 4096 * 16)); CHECK(hipMalloc(&bad, 16)); CHECK(hipMalloc(&sink, 16));
  CHECK(hipMalloc(&data, (size_t)n_data * 8));
  { float2* h = (float2*)malloc((size_t)n_data * 8); for (int i = 0; i < n_data; i++) { h[i].x = (float)(rand() % 2001) * 0.001f - 1.f; h[i].y = (float)(rand() % 2001) * 0.001f - 1.f; }
    CHECK(hipMemcpy(data, h, (size_t)n_data * 8, hipMemcpyHostToDevice)); free(h); }
  hipStream_t s1, s2; CHECK(hipStreamCreate(&s1)); CHECK(hipStreamCreate(&s2));
  CHECK(hipFuncSetAttribute((const void*)writer, hipFuncAttributeMaxDynamicSharedMemorySize, 110 * 1024));
  const int lds_sizes[] = {0, 36 * 1024, 72 * 1024, 108 * 1024};
  for (int mfma = 0; mfma < 2; mfma++)
    for (int li = 0; li < 4; li++)
      for (int mode = 0; mode < 4; mode++) {
        CHECK(hipMemset(bad, 0, 16));
        for (int rep = 0; rep < 10; rep++) {
          if (lds_sizes[li]) hipLaunchKernelGGL(writer, dim3(1024), dim3(256), lds_sizes[li], s1, src, lds_sizes[li], 60, mfma, sink);
          hipLaunchKernelGGL(victim, dim3(4096), dim3(256), 0, s2, data, n_data, 400, mode, bad);
        }
        CHECK(hipDeviceSynchronize());
        unsigned h[4]; CHECK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
        printf("writer LDS %3d KB%s | victim mode %d (%s operands%s): mismatching packed results  low lane %u  high lane %u\n", lds_sizes[li] / 1024,
               lds_sizes[li] ? (mfma ? " + MFMA" : "       ") : " (none)", mode, (mode & 1) ? "loaded" : "register", (mode & 2) ? ", LDS atomics" : "", h[0], h[1]);
        fflush(stdout);
      }
  return 0;
}
