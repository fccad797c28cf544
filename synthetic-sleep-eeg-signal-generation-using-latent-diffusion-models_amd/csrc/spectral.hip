// Spectral ("Jukebox") loss, forward and gradient in one launch:
//   L = sum_k ( |FFT_ortho(recon)[k]| - |FFT_ortho(target)[k]| )^2      (two-sided, reduction="sum")
// JukeboxLoss(spatial_dims=1, reduction="sum") at /root/reference/src/train_autoencoderkl.py:158,208
// (MONAI; fftn over dims (1,2) -- with one channel that is a 1-D FFT of the 3072-sample window).
//
// One workgroup per window, everything LDS-resident (3 x 24 KB):
//   * recon and target are transformed TOGETHER as one complex FFT of z = recon + i*target;
//     the two real spectra are separated with the Hermitian identities.
//   * N = 3072 = 48 x 64 is done as a four-step FFT (48-point DFTs down the columns, twiddle,
//     64-point DFTs along the rows) with a sin/cos table of the N-th roots in LDS -- ~112 complex
//     MACs per output, negligible next to the autoencoder, exact to fp32 rounding, no bit reversal.
//   * gradient: dL/d recon[n] = Re FFT_ortho(G)[n], G[k] = 2(A_k - B_k) conj(R_k)/A_k
//     (G_k := 0 where A_k is zero to rounding, A_k^2 <= 1e-10 x the window's mean bin power: the two spectra are separated from
//      ONE packed fp32 FFT, so an exactly-zero A_k comes out as rounding noise of ~1e-7 of the spectrum's rms amplitude with a
//      meaningless phase; torch produces NaN there --
//      README.md:17 notes that instability.  Deviation stated in INTEGRATION.md, tested in tests/test_gpu_aekl.py).
// HBM traffic = the two windows in, one gradient window out: HBM-bound, ~36 KB per window.
#include "common.h"
#include "internal.h"

namespace {
constexpr int NT = 256;

template <int N1, int N2>
struct FFT {
  static constexpr int N = N1 * N2;
  static constexpr int YP = N2 + 1;                    // padded row pitch of the intermediate
  static constexpr int LDS_BYTES = (N + N + N1 * YP) * (int)sizeof(float2);

  // forward DFT (e^{-2 pi i nk/N}) of a[0..N) -> x[0..N), natural order both sides; y is scratch [N1][YP]
  __device__ static void run(const float2* __restrict__ a, float2* __restrict__ x, float2* __restrict__ y, const float2* __restrict__ tw) {
    // step 1+2: Y[k1][n2] = W_N^{n2 k1} * sum_{n1} a[N2 n1 + n2] W_N1^{n1 k1}
    for (int idx = threadIdx.x; idx < N; idx += NT) {
      const int k1 = idx / N2, n2 = idx - k1 * N2;
      float re = 0.f, im = 0.f;
      int t = 0; const int step = (N2 * k1) % N;
#pragma unroll 8
      for (int n1 = 0; n1 < N1; n1++) {
        const float2 v = a[N2 * n1 + n2], w = tw[t];
        re += v.x * w.x - v.y * w.y; im += v.x * w.y + v.y * w.x;
        t += step; if (t >= N) t -= N;
      }
      const float2 w = tw[(n2 * k1) % N];
      y[k1 * YP + n2] = make_float2(re * w.x - im * w.y, re * w.y + im * w.x);
    }
    __syncthreads();
    // step 3: X[k1 + N1 k2] = sum_{n2} Y[k1][n2] W_N2^{n2 k2}
    for (int idx = threadIdx.x; idx < N; idx += NT) {
      const int k2 = idx / N1, k1 = idx - k2 * N1;
      float re = 0.f, im = 0.f;
      int t = 0; const int step = (N1 * k2) % N;
#pragma unroll 8
      for (int n2 = 0; n2 < N2; n2++) {
        const float2 v = y[k1 * YP + n2], w = tw[t];
        re += v.x * w.x - v.y * w.y; im += v.x * w.y + v.y * w.x;
        t += step; if (t >= N) t -= N;
      }
      x[idx] = make_float2(re, im);
    }
    __syncthreads();
  }
};

template <int N1, int N2>
__global__ __launch_bounds__(NT) void spectral_kernel(const float* __restrict__ recon, const float* __restrict__ target, float* __restrict__ loss,
                                                      float* __restrict__ drecon, float gweight, int loss_stride) {
  using F = FFT<N1, N2>;
  constexpr int N = F::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float2* bufA = (float2*)smem;
  float2* bufX = bufA + N;
  float2* bufY = bufX + N;
  __shared__ float2 tw[N];        // N-th roots of unity, e^{-2 pi i j/N}
  __shared__ float red[4];
  const long base = (long)blockIdx.x * N;
  const float scale = rsqrtf((float)N);
  float en = 0.f;
  for (int j = threadIdx.x; j < N; j += NT) {
    float s, c; sincospif(-2.0f * (float)j / (float)N, &s, &c);
    tw[j] = make_float2(c, s);
    const float2 z = make_float2(recon[base + j], target[base + j]);
    bufA[j] = z; en += z.x * z.x + z.y * z.y;
  }
  en = wave_sum(en);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = en;
  __syncthreads();
  // Parseval (ortho norm): sum_k |Z_k|^2 = sum_n |z_n|^2, so this is 1e-10 x the mean power of a bin
  const float zero_thr = 1e-10f * (red[0] + red[1] + red[2] + red[3]) / (float)N;
  F::run(bufA, bufX, bufY, tw);
  float part = 0.f;
  for (int k = threadIdx.x; k < N; k += NT) {
    const float2 zk = bufX[k], zm = bufX[k ? N - k : 0];
    const float rr = 0.5f * scale * (zk.x + zm.x), ri = 0.5f * scale * (zk.y - zm.y);      // R_k
    const float tr = 0.5f * scale * (zk.y + zm.y), ti = -0.5f * scale * (zk.x - zm.x);     // T_k
    const float A = sqrtf(rr * rr + ri * ri), Bm = sqrtf(tr * tr + ti * ti);
    const float d = A - Bm;
    part += d * d;
    const float g = (A * A > zero_thr) ? 2.0f * d / A : 0.f;
    bufA[k] = make_float2(g * rr, -g * ri);      // G_k = 2(A-B) conj(R_k)/A
  }
  part = wave_sum(part);
  __syncthreads();                       // red[] was read for zero_thr above
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss + (size_t)blockIdx.x * loss_stride, red[0] + red[1] + red[2] + red[3]);
  if (drecon) {
    F::run(bufA, bufX, bufY, tw);
    for (int n = threadIdx.x; n < N; n += NT) drecon[base + n] += gweight * scale * bufX[n].x;
  }
}

template <int N1, int N2>
int launch(eegldm_ctx* ctx, const float* recon, const float* target, float* loss, float* drecon, int B, float w) {
  auto kern = spectral_kernel<N1, N2>;
  constexpr int lds = FFT<N1, N2>::LDS_BYTES;
  static DevOnce attr;
  if (attr.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  float* dst = loss; int stride = 0;
  if (eeg_deterministic()) {      // a partial per window (zeroed: one writer each), then the windows in order
    EEG_TRY(eeg_det_buffer(ctx, (size_t)B * sizeof(float), &dst)); stride = 1;
    HIP_TRY(hipMemsetAsync(dst, 0, (size_t)B * sizeof(float), ctx->stream));
  }
  hipLaunchKernelGGL(kern, dim3(B), dim3(NT), lds, ctx->stream, recon, target, dst, drecon, w, stride);
  LAUNCH_CHECK();
  if (stride) EEG_TRY(ew_fold_partials_det(ctx, dst, B, 1, 0, 1, loss));
  return 0;
}
}  // namespace

// loss (device scalar) = sum over the B windows; d_recon_accum (nullable) += grad_weight * dL/d recon
extern "C" int eegldm_spectral_loss(eegldm_ctx* ctx, const float* recon, const float* target, float* loss, float* d_recon_accum,
                                    int B, int C, int L, float grad_weight) {
  EEG_CHECK(ctx && recon && target && loss && B > 0, "bad argument");
  EEG_CHECK(C == 1, "spectral loss is implemented for single-channel windows (the reference's 1ch x 3072 EEG); got C=%d", C);
  if (!ctx->loss_prezeroed) HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), ctx->stream));
  if (L == 3072) return launch<48, 64>(ctx, recon, target, loss, d_recon_accum, B, grad_weight);
  if (L == 768) return launch<24, 32>(ctx, recon, target, loss, d_recon_accum, B, grad_weight);
  if (L == 256) return launch<16, 16>(ctx, recon, target, loss, d_recon_accum, B, grad_weight);
  if (L == 96) return launch<8, 12>(ctx, recon, target, loss, d_recon_accum, B, grad_weight);
  EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "spectral loss window length %d (supported: 3072, 768, 256, 96)", L);
}
