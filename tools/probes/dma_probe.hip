// Probe: __builtin_amdgcn_global_load_lds (16-byte LDS-DMA) semantics used by csrc/gemm.hip:
// LDS destination = wave-uniform pointer + lane*16; per-lane global source; completion via vmcnt
// (plain __syncthreads() must cover it).  Prints OK/FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

__global__ void dma_probe(const unsigned* __restrict__ src, unsigned* __restrict__ out, const unsigned* __restrict__ zeros) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 256 threads move 2 x 4 KB: chunk c (16 B) of LDS <- src chunk perm(c), odd chunks of the 2nd half <- zero page
  for (int it = 0; it < 2; it++) {
    const int c = it * 256 + tid;                     // LDS chunk index
    const int sc = (c ^ 5) ;                          // some per-lane permutation of the source chunk
    const unsigned* g = (it == 1 && (c & 1)) ? zeros : src + sc * 4;
    char* dst_wave = smem + (it * 256 + wave * 64) * 16;     // wave-uniform base
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lds_ptr_t)dst_wave, 16, 0, 0);
  }
  __syncthreads();
  for (int i = tid; i < 2048; i += 256) out[i] = ((const unsigned*)smem)[i];
}

int main() {
  std::vector<unsigned> h(2048), o(2048), z(4, 0);
  for (int i = 0; i < 2048; i++) h[i] = 1000 + i;
  unsigned *d, *dout, *dz;
  hipMalloc(&d, 8192); hipMalloc(&dout, 8192); hipMalloc(&dz, 16);
  hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(dz, z.data(), 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(dma_probe, dim3(1), dim3(256), 8192, 0, d, dout, dz);
  hipMemcpy(o.data(), dout, 8192, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int c = 0; c < 512; c++) for (int j = 0; j < 4; j++) {
    unsigned want = (c >= 256 && (c & 1)) ? 0u : 1000 + ((c ^ 5) * 4 + j);
    if (o[c * 4 + j] != want) { if (bad < 5) printf("chunk %d word %d: got %u want %u\n", c, j, o[c * 4 + j], want); bad++; }
  }
  printf("LDS-DMA probe: %s (%d mismatches), status %s\n", bad ? "FAIL" : "OK", bad, hipGetErrorString(hipDeviceSynchronize()));
  return 0;
}
