// Hardware probe (developer tool, not part of the product path).
// Prints the lane->element maps of ds_read_b64_tr_b16 and of the two MFMA
// shapes csrc/gemm.hip relies on, so the fragment code there can be checked
// against silicon rather than against recollection.
//   hipcc --offload-arch=gfx950 -O2 layout_probe.hip -o layout_probe && ./layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ void tr_probe(short* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 4];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
  __syncthreads();
  // lane p supplies the address of chunk p (4 shorts = 8 bytes, values 4p..4p+3)
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
  for (int j = 0; j < 4; j++) out[threadIdx.x * 4 + j] = v[j];
}

// D = A*B, A[m][k] = m + k/64 style unique small values, B = selector.
__global__ void mfma_bf16_probe(float* out, int ksel0) {
  int l = threadIdx.x;
  bf16x8 a, b;
  f32x4 acc = {0, 0, 0, 0};
  for (int j = 0; j < 8; j++) {
    int k = (l >> 4) * 8 + j;  // hypothesis under test
    int m = l & 15;
    a[j] = (__bf16)(float)(m * 8 + (k & 7) + ((k >> 3) * 0));  // exact in bf16 (<256)
    // store k>>3 separately in a second run (ksel picks a 16-wide k window)
    b[j] = (__bf16)((k == ksel0 + (l & 15)) ? 1.0f : 0.0f);  // B[k][n=l&15]
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; i++) out[l * 4 + i] = acc[i];
}

__global__ void mfma_f32_probe(float* out) {
  int l = threadIdx.x;
  float a = (float)((l & 15) * 4 + (l >> 4));          // A[m=l&15][k=l>>4] = 4m+k
  float b = ((l >> 4) == (l & 15)) ? 1.0f : 0.0f;      // B[k=l>>4][n=l&15] = (k==n)
  f32x4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  for (int i = 0; i < 4; i++) out[l * 4 + i] = acc[i];
}

int main() {
  short* ds; float* df;
  hipMalloc(&ds, 256 * sizeof(short)); hipMalloc(&df, 256 * sizeof(float));
  std::vector<short> hs(256); std::vector<float> hf(256);

  tr_probe<<<1, 64>>>(ds);
  hipMemcpy(hs.data(), ds, 256 * 2, hipMemcpyDeviceToHost);
  printf("TR16_B64: lane -> 4 source element indices (chunk p = elems 4p..4p+3)\n");
  for (int l = 0; l < 64; l++)
    printf("  lane %2d: %3d %3d %3d %3d\n", l, hs[l * 4], hs[l * 4 + 1], hs[l * 4 + 2], hs[l * 4 + 3]);
  // expectation: lane i (group g=i>>4, i'=i&15), elem j = 64g + 16j + i'
  int bad = 0;
  for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++)
    if (hs[l * 4 + j] != 64 * (l >> 4) + 16 * j + (l & 15)) bad++;
  printf("TR16 hypothesis (lane i elem j = 64*(i>>4)+16*j+(i&15)): %s (%d mismatches)\n", bad ? "FALSE" : "TRUE", bad);

  for (int ks = 0; ks < 32; ks += 16) {
    mfma_bf16_probe<<<1, 64>>>(df, ks);
    hipMemcpy(hf.data(), df, 256 * 4, hipMemcpyDeviceToHost);
    // with the hypothesis, D[m][n] = A[m][k=ks+n] = m*8 + ((ks+n)&7); D map: col=lane&15,row=(lane>>4)*4+reg
    int badm = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
      int n = l & 15, m = (l >> 4) * 4 + r;
      float exp = (float)(m * 8 + ((ks + n) & 7));
      if (hf[l * 4 + r] != exp) badm++;
    }
    printf("MFMA 16x16x32 bf16 A/B[k=(l>>4)*8+j], D[row=(l>>4)*4+r][col=l&15], ksel=%d: %s (%d)\n", ks, badm ? "FALSE" : "TRUE", badm);
  }
  mfma_f32_probe<<<1, 64>>>(df);
  hipMemcpy(hf.data(), df, 256 * 4, hipMemcpyDeviceToHost);
  int badf = 0;
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
    int n = l & 15, m = (l >> 4) * 4 + r;
    float exp = (n < 4) ? (float)(m * 4 + n) : 0.0f;
    if (hf[l * 4 + r] != exp) badf++;
  }
  printf("MFMA 16x16x4 f32 A[m=l&15][k=l>>4] B[k=l>>4][n=l&15]: %s (%d)\n", badf ? "FALSE" : "TRUE", badf);
  hipError_t e = hipDeviceSynchronize();
  printf("status: %s\n", hipGetErrorString(e));
  return 0;
}
