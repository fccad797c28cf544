// Probe 3: do plain global loads of a narrow-slab kernel (the access pattern of the 256-thread GroupNorm backward: 8 bytes per lane,
// 32-byte row segments, 12 rows per thread, everything loaded up front) return the right data while the library's LDS-DMA
// weight-gradient GEMM runs on another stream?  victim: y = x through registers, with a block barrier and an LDS fp64 atomic in
// between (as in the GroupNorm kernel); checked bitwise on the device afterwards.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "eegldm.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define ECHECK(x) do { int r_ = (x); if (r_) { printf("%s -> %s\n", #x, eegldm_last_error()); return 1; } } while (0)

// one block = one sample x 16 channels (32-byte rows of a [rows][C] bf16 tensor), L rows: thread (tx, ty): tx = 8-byte column, ty = row lane
__global__ __launch_bounds__(256) void victim_copy(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int L, int C, int xcd) {
  __shared__ double red[64];
  const int nchunk = C / 16, lin = blockIdx.x;
  int chunk, b;
  if (xcd) { const int xx = lin & 7, j = lin >> 3; b = (j / nchunk) * 8 + xx; chunk = j % nchunk; } else { chunk = lin % nchunk; b = lin / nchunk; }
  const int tx = threadIdx.x & 3, ty = threadIdx.x >> 2;            // 4 x 64
  if (threadIdx.x < 64) red[threadIdx.x] = 0.0;
  uint2 raw[12];
  const char* xs = (const char*)(x + (long)b * L * C) + (chunk * 16 + tx * 4) * 2;
  for (int k = 0; k < 12; k++) { const int l = k * 64 + ty; if (l < L) raw[k] = *(const uint2*)(xs + (long)l * C * 2); }
  __syncthreads();
  double s = 0.0;
  for (int k = 0; k < 12; k++) { const int l = k * 64 + ty; if (l < L) s += (double)(raw[k].x & 0xffff); }
  atomicAdd(&red[tx], s);
  __syncthreads();
  char* ys = (char*)(y + (long)b * L * C) + (chunk * 16 + tx * 4) * 2;
  const unsigned salt = red[tx] < 0.0 ? 1u : 0u;                   // (keeps the atomic alive; always 0)
  for (int k = 0; k < 12; k++) { const int l = k * 64 + ty; if (l < L) { uint2 v = raw[k]; v.x ^= salt; *(uint2*)(ys + (long)l * C * 2) = v; } }
}
__global__ void compare(const uint4* a, const uint4* b, long n, unsigned* bad) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const uint4 u = a[i], v = b[i];
    if (u.x != v.x || u.y != v.y || u.z != v.z || u.w != v.w) atomicAdd(bad, 1u);
  }
}

int main() {
  const int B = 256, L = 768, C = 128;          // the level-0 GroupNorm of the UNet
  const int Lw = 192, Cw = 512;
  eegldm_ctx* ctx; ECHECK(eegldm_ctx_create(0, nullptr, 1, &ctx));
  void *xw, *dyw; float *dw, *db; unsigned short *x, *y; unsigned* bad;
  CHECK(hipMalloc(&xw, (size_t)B * Lw * Cw * 2)); CHECK(hipMalloc(&dyw, (size_t)B * Lw * Cw * 2));
  CHECK(hipMemset(xw, 0x3c, (size_t)B * Lw * Cw * 2)); CHECK(hipMemset(dyw, 0x3d, (size_t)B * Lw * Cw * 2));
  CHECK(hipMalloc(&dw, (size_t)3 * Cw * Cw * 4)); CHECK(hipMalloc(&db, Cw * 4)); CHECK(hipMemset(dw, 0, (size_t)3 * Cw * Cw * 4)); CHECK(hipMemset(db, 0, Cw * 4));
  const size_t n = (size_t)B * L * C;
  CHECK(hipMalloc(&x, n * 2)); CHECK(hipMalloc(&y, n * 2)); CHECK(hipMalloc(&bad, 4));
  { unsigned short* h = new unsigned short[n]; for (size_t i = 0; i < n; i++) h[i] = (unsigned short)(i * 2654435761u >> 13); CHECK(hipMemcpy(x, h, n * 2, hipMemcpyHostToDevice)); delete[] h; }
  hipStream_t s2; CHECK(hipStreamCreate(&s2));
  for (int xcd = 0; xcd < 2; xcd++)
    for (int noise = 0; noise < 2; noise++) {
      unsigned tot = 0;
      for (int rep = 0; rep < 10; rep++) {
        CHECK(hipMemset(y, 0, n * 2)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipDeviceSynchronize());
        if (noise) for (int k = 0; k < 6; k++) ECHECK(eegldm_conv1d_bwd_weight(ctx, xw, Cw, dyw, Cw, dw, db, B, Lw, Cw, Cw, 3, 1, 1, 1, EEGLDM_BF16));
        for (int k = 0; k < 6; k++) hipLaunchKernelGGL(victim_copy, dim3(B * C / 16), dim3(256), 0, s2, x, y, L, C, xcd);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(compare, dim3(1024), dim3(256), 0, s2, (const uint4*)x, (const uint4*)y, (long)(n / 8), bad);
        unsigned h; CHECK(hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost)); tot += h;
      }
      printf("victim copy (xcd order %d) %s: %u mismatching 16-byte chunks in 10 runs\n", xcd, noise ? "beside the weight-gradient GEMM" : "quiet", tot);
    }
  return 0;
}
