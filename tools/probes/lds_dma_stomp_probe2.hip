// Probe 2: the same victims as lds_dma_stomp_probe.hip, but beside the LIBRARY's fused 3-tap weight-gradient GEMM (LDS-DMA ring,
// or register staging with EEGLDM_WGRAD_NO_DMA=1) running on a second stream.  Mismatching LDS words are dumped (index, value).
//   hipcc --offload-arch=gfx950 -O2 -I../../include lds_dma_stomp_probe2.hip -L<package dir> -leegldm -Wl,-rpath,<package dir>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "eegldm.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#define ECHECK(x) do { int r_ = (x); if (r_) { printf("%s -> %s\n", #x, eegldm_last_error()); return 1; } } while (0)

__global__ __launch_bounds__(256) void victim(int spins, unsigned* __restrict__ sink, unsigned* __restrict__ dump) {
  __shared__ unsigned pat[1792];                        // 7 KB
  const int tid = threadIdx.x;
  for (int i = tid; i < 1792; i += 256) pat[i] = 0x5A000000u + i;
  __syncthreads();
  unsigned bad = 0;
  for (int s = 0; s < spins; s++) {
    for (int i = tid; i < 1792; i += 256) {
      const unsigned v = pat[i];
      if (v != 0x5A000000u + i) {
        bad++;
        const unsigned k = atomicAdd(sink + 2, 1u);
        if (k < 64) { dump[4 * k] = blockIdx.x; dump[4 * k + 1] = i; dump[4 * k + 2] = v; dump[4 * k + 3] = s; }
        pat[i] = 0x5A000000u + i;
      }
    }
    __builtin_amdgcn_s_sleep(8);
  }
  if (bad) atomicAdd(sink, bad);
}
__global__ __launch_bounds__(256) void victim_atomic(int rounds, unsigned* __restrict__ sink) {
  __shared__ double red[896];
  const int tid = threadIdx.x;
  unsigned bad = 0;
  for (int r = 0; r < rounds; r++) {
    for (int i = tid; i < 896; i += 256) red[i] = 0.0;
    __syncthreads();
    for (int k = 0; k < 16; k++) atomicAdd(&red[(tid & 15) * 4 + (k & 3)], (double)(1 + (tid >> 4)));
    __syncthreads();
    if (tid < 64) bad += (red[tid] != 544.0);
    __syncthreads();
  }
  if (bad) atomicAdd(sink + 1, bad);
}
// victim 3: registers only (no LDS): a long dependent integer chain whose result is known
__global__ __launch_bounds__(256) void victim_regs(int rounds, unsigned* __restrict__ sink) {
  unsigned v[32];
  for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 33 + i;
  unsigned bad = 0;
  for (int r = 0; r < rounds; r++) {
    for (int i = 0; i < 32; i++) bad += (v[i] != threadIdx.x * 33 + i + r);
    for (int i = 0; i < 32; i++) v[i] += 1;
    __builtin_amdgcn_s_sleep(4);
  }
  if (bad) atomicAdd(sink + 3, bad);
}

int main() {
  const int B = 256, L = 192, C = 512;
  eegldm_ctx* ctx; ECHECK(eegldm_ctx_create(0, nullptr, 1, &ctx));
  void *x, *dy; float *dw, *db; unsigned *sink, *dump;
  CHECK(hipMalloc(&x, (size_t)B * L * C * 2)); CHECK(hipMalloc(&dy, (size_t)B * L * C * 2));
  CHECK(hipMemset(x, 0x3c, (size_t)B * L * C * 2)); CHECK(hipMemset(dy, 0x3d, (size_t)B * L * C * 2));    // bf16 0x3c3c / 0x3d3d
  CHECK(hipMalloc(&dw, (size_t)3 * C * C * 4)); CHECK(hipMalloc(&db, C * 4)); CHECK(hipMemset(dw, 0, (size_t)3 * C * C * 4)); CHECK(hipMemset(db, 0, C * 4));
  CHECK(hipMalloc(&sink, 16)); CHECK(hipMalloc(&dump, 64 * 16));
  hipStream_t s2; CHECK(hipStreamCreate(&s2));
  for (int noise = 0; noise < 2; noise++) {
    CHECK(hipMemset(sink, 0, 16)); CHECK(hipMemset(dump, 0, 64 * 16));
    for (int rep = 0; rep < 10; rep++) {
      if (noise) for (int k = 0; k < 6; k++) ECHECK(eegldm_conv1d_bwd_weight(ctx, x, C, dy, C, dw, db, B, L, C, C, 3, 1, 1, 1, EEGLDM_BF16));
      hipLaunchKernelGGL(victim, dim3(8192), dim3(256), 0, s2, 100, sink, dump);
      hipLaunchKernelGGL(victim_atomic, dim3(8192), dim3(256), 0, s2, 30, sink);
      hipLaunchKernelGGL(victim_regs, dim3(8192), dim3(256), 0, s2, 100, sink);
      CHECK(hipDeviceSynchronize());
    }
    unsigned h[4], d[256]; CHECK(hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(d, dump, 1024, hipMemcpyDeviceToHost));
    printf("%s: LDS pattern mismatches %u (events %u), wrong fp64 atomic sums %u, register mismatches %u\n", noise ? "beside the weight-gradient GEMM" : "quiet", h[0], h[2], h[1], h[3]);
    for (unsigned k = 0; k < (h[2] < 24 ? h[2] : 24); k++) printf("   block %u word %u value %08x spin %u\n", d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
  }
  return 0;
}
