// Probe: throughput of fp32 atomic adds as the split-K weight-gradient epilogue issues them (512 blocks x 256 threads, each block adds
// a 128 x 64 x 3 fp32 tile = 24576 values, natural MFMA fragment layout: a wave instruction touches 4 rows x 16 consecutive floats)
//   mode 0: device(agent)-scope atomics, ONE shared tile (what gemm.hip does today)
//   mode 1: agent-scope atomics, one tile per XCD (index = HW_REG_XCC_ID)
//   mode 2: workgroup-scope atomics (no sc1: executed in the XCD's own L2), one tile per XCD
//   mode 3: plain stores of every block's partial tile to a workspace (for comparison)
// plus a correctness check of mode 2 (sum over the 8 XCD tiles == number of blocks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

template <int MODE>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, int tiles, int reps) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, q = lane >> 4, wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x % tiles;
  constexpr int TILE = 3 * 128 * 64;
  float* base;
  if (MODE == 0) base = out + (size_t)tile * TILE;
  else if (MODE == 3) base = out + (size_t)blockIdx.x * TILE;
  else base = out + ((size_t)xcc_id() * tiles + tile) * TILE;
  for (int rep = 0; rep < reps; rep++)
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < 4; i++)
        for (int r = 0; r < 4; r++) {
          const int m = wm * 64 + i * 16 + q * 4 + r;
          for (int j = 0; j < 2; j++) {
            const int n = wn * 32 + j * 16 + lm;
            float* p = base + (size_t)a * 128 * 64 + m * 64 + n;
            if (MODE == 0 || MODE == 1) atomicAdd(p, 1.0f);
            else if (MODE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else *p = 1.0f;
          }
        }
}

template <int MODE> int run(float* buf, size_t bytes, int blocks, int tiles, const char* name) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  CHECK(hipMemset(buf, 0, bytes));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, tiles, 1);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(buf, 0, bytes));
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, buf, tiles, 1);
  CHECK(hipEventRecord(b, 0)); CHECK(hipDeviceSynchronize());
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  const double n = (double)blocks * 24576;
  printf("%-44s blocks %4d tiles %3d: %8.1f us  %7.1f G adds/s\n", name, blocks, tiles, ms * 1e3, n / ms / 1e6);
  return 0;
}

int main() {
  const size_t bytes = (size_t)512 * 24576 * 4 + (8 * 32 * 24576 * 4);
  float* buf; CHECK(hipMalloc(&buf, bytes));
  for (int tiles : {2, 8, 32}) {
    run<0>(buf, bytes, 512, tiles, "agent atomics, shared tile");
    run<1>(buf, bytes, 512, tiles, "agent atomics, per-XCD tile");
    run<2>(buf, bytes, 512, tiles, "workgroup-scope atomics, per-XCD tile");
    run<3>(buf, bytes, 512, tiles, "plain stores, per-block partial");
  }
  // correctness of mode 2
  CHECK(hipMemset(buf, 0, bytes));
  hipLaunchKernelGGL(k<2>, dim3(512), dim3(256), 0, 0, buf, 2, 1);
  CHECK(hipDeviceSynchronize());
  std::vector<float> h((size_t)8 * 2 * 24576);
  CHECK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
  double bad = 0; double per_xcd[8] = {0};
  for (int t = 0; t < 2; t++) for (int e = 0; e < 24576; e++) {
    double s = 0; for (int x = 0; x < 8; x++) { s += h[((size_t)x * 2 + t) * 24576 + e]; if (e == 0 && t == 0) per_xcd[x] = h[((size_t)x * 2 + t) * 24576]; }
    if (s != 256.0) bad++;
  }
  printf("mode 2 correctness: %s (%g elements with a wrong total); blocks per XCD seen on tile 0:", bad ? "FAIL" : "OK", bad);
  for (int x = 0; x < 8; x++) printf(" %g", per_xcd[x]);
  printf("\n");
  return 0;
}
