// Quality metrics of generated / reconstructed windows, on the device:
//   * 1-D multi-scale SSIM: the reference's local 1-D adaptation of MONAI's MultiScaleSSIMMetric
//     (/root/reference/src/compute_mmds.py:214-408; called with spatial_dims=1, data_range=1.0, kernel_size=7 at :487).
//   * multitaper power spectral density up to fmax (mne Epochs.compute_psd(fmax=18) as used at
//     /root/reference/src/sample_trials.py:172-181): DPSS tapers and their weights are supplied by the host (a few KB,
//     computed once), the per-window work -- de-mean, taper, DFT bins 0..n_bins-1, weighted |.|^2 average -- runs here.
// Both are one workgroup per window with the window(s) resident in LDS: HBM traffic = the windows in, a few numbers out.
#include "common.h"

namespace {
constexpr int NT = 256;
constexpr int MAX_TAPS = 17, MAX_SCALES = 8;
struct SsimParams { float k[MAX_TAPS]; float w[MAX_SCALES]; int ksize, n_scales; float c1, c2; };

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// out[b] = prod_s relu(mean cs_s)^w_s (last scale: relu(mean ssim)); mean over channels and valid positions (compute_mmds.py:370-397)
__global__ __launch_bounds__(NT) void ms_ssim_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                     int C, int L, SsimParams P) {
  extern __shared__ float sm[];
  float* X = sm; float* Y = sm + L;
  __shared__ float red[4];
  const long base = (long)blockIdx.x * C * L;
  float cs_acc[MAX_SCALES], ss_last = 0.f;
#pragma unroll
  for (int s = 0; s < MAX_SCALES; s++) cs_acc[s] = 0.f;
  for (int c = 0; c < C; c++) {
    __syncthreads();
    for (int i = threadIdx.x; i < L; i += NT) { X[i] = a[base + (long)c * L + i]; Y[i] = b[base + (long)c * L + i]; }
    __syncthreads();
    int Ls = L;
    for (int s = 0; s < P.n_scales; s++) {
      const int Lo = Ls - P.ksize + 1;
      float cs_sum = 0.f, ss_sum = 0.f;
      for (int p = threadIdx.x; p < Lo; p += NT) {
        float mx = 0.f, my = 0.f, mxx = 0.f, myy = 0.f, mxy = 0.f;
        for (int t = 0; t < P.ksize; t++) {
          const float k = P.k[t], x = X[p + t], y = Y[p + t];
          mx += k * x; my += k * y; mxx += k * x * x; myy += k * y * y; mxy += k * x * y;
        }
        const float sx = mxx - mx * mx, sy = myy - my * my, sxy = mxy - mx * my;
        const float cs = (2.0f * sxy + P.c2) / (sx + sy + P.c2);
        cs_sum += cs;
        ss_sum += (2.0f * mx * my + P.c1) / (mx * mx + my * my + P.c1) * cs;
      }
      cs_sum = block_sum(cs_sum, red);
      cs_acc[s] += cs_sum / (float)Lo;
      if (s == P.n_scales - 1) { ss_sum = block_sum(ss_sum, red); ss_last += ss_sum / (float)Lo; }
      // avg_pool1d(kernel_size=2): pairs, a trailing odd element is dropped (compute_mmds.py:388-389)
      const int Ln = Ls / 2;
      float px[8], py[8];                         // L <= 4096 -> at most 8 outputs per thread
      int cnt = 0;
      __syncthreads();
      for (int i = threadIdx.x; i < Ln; i += NT, cnt++) { px[cnt] = 0.5f * (X[2 * i] + X[2 * i + 1]); py[cnt] = 0.5f * (Y[2 * i] + Y[2 * i + 1]); }
      __syncthreads();
      cnt = 0;
      for (int i = threadIdx.x; i < Ln; i += NT, cnt++) { X[i] = px[cnt]; Y[i] = py[cnt]; }
      __syncthreads();
      Ls = Ln;
    }
  }
  if (threadIdx.x == 0) {
    float r = 1.f;
    for (int s = 0; s < P.n_scales; s++) {
      float v = (s == P.n_scales - 1 ? ss_last : cs_acc[s]) / (float)C;
      v = fmaxf(v, 0.f);
      r *= powf(v, P.w[s]);
    }
    out[blockIdx.x] = r;
  }
}

// psd[b][k] = scale_k * sum_t |w_t * DFT_k(taper_t * (x_b - mean))|^2 / sum_t w_t^2 / sfreq, one-sided:
// scale_k = 2 except DC (and Nyquist for even L) = 1  (mne.time_frequency.multitaper: _mt_spectra / _psd_from_mt, normalization="length")
__global__ __launch_bounds__(NT) void psd_kernel(const float* __restrict__ x, const float* __restrict__ tapers, const float* __restrict__ tw2,
                                                 float* __restrict__ psd, int L, int K, int n_bins, float inv_norm) {
  extern __shared__ float sm[];
  float* xs = sm;                         // [L] de-meaned window
  float2* tw = (float2*)(sm + ((L + 1) & ~1));     // [L] e^{-2 pi i j / L}
  __shared__ float red[4];
  const float* xb = x + (long)blockIdx.x * L;
  float s = 0.f;
  for (int i = threadIdx.x; i < L; i += NT) s += xb[i];
  const float mean = block_sum(s, red) / (float)L;
  for (int i = threadIdx.x; i < L; i += NT) {
    xs[i] = xb[i] - mean;
    float sn, cs; sincospif(-2.0f * (float)i / (float)L, &sn, &cs);
    tw[i] = make_float2(cs, sn);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_bins; k += NT) {
    float acc = 0.f;
    for (int t = 0; t < K; t++) {
      const float* tp = tapers + (long)t * L;
      float re = 0.f, im = 0.f;
      int idx = 0;
      for (int n = 0; n < L; n++) {
        const float v = xs[n] * tp[n]; const float2 w = tw[idx];
        re += v * w.x; im += v * w.y;
        idx += k; if (idx >= L) idx -= L;
      }
      acc += tw2[t] * (re * re + im * im);
    }
    const bool edge = k == 0 || ((L & 1) == 0 && k == L / 2);
    psd[(long)blockIdx.x * n_bins + k] = acc * inv_norm * (edge ? 1.0f : 2.0f);
  }
}
}  // namespace

extern "C" int eegldm_ms_ssim_1d(eegldm_ctx* ctx, const float* a, const float* b, float* out, int B, int C, int L, const float* kernel_host,
                                 int ksize, const float* weights_host, int n_scales, float data_range, float k1, float k2) {
  EEG_CHECK(ctx && a && b && out && kernel_host && weights_host, "null argument");
  EEG_CHECK(B >= 1 && C >= 1 && L >= 1 && L <= 4096, "window length %d outside [1, 4096]", L);
  EEG_CHECK(ksize >= 1 && ksize <= MAX_TAPS && n_scales >= 1 && n_scales <= MAX_SCALES, "kernel_size <= %d and <= %d scales", MAX_TAPS, MAX_SCALES);
  const int div = (n_scales - 1) > 1 ? (n_scales - 1) * (n_scales - 1) : 1;         // compute_mmds.py:361-369
  EEG_CHECK(L / div > ksize - 1, "for %d scales and kernel size %d the window must be longer than %d", n_scales, ksize, (ksize - 1) * div);
  SsimParams P = {};
  for (int i = 0; i < ksize; i++) P.k[i] = kernel_host[i];
  for (int i = 0; i < n_scales; i++) P.w[i] = weights_host[i];
  P.ksize = ksize; P.n_scales = n_scales; P.c1 = (k1 * data_range) * (k1 * data_range); P.c2 = (k2 * data_range) * (k2 * data_range);
  hipLaunchKernelGGL(ms_ssim_kernel, dim3(B), dim3(NT), sizeof(float) * 2 * L, ctx->stream, a, b, out, C, L, P);
  LAUNCH_CHECK();
  return 0;
}

extern "C" int eegldm_psd_multitaper(eegldm_ctx* ctx, const float* x, const float* tapers, const float* weights_host, int n_tapers, float sfreq,
                                     int n_bins, float* psd, int B, int L) {
  EEG_CHECK(ctx && x && tapers && weights_host && psd, "null argument");
  EEG_CHECK(B >= 1 && L >= 2 && L <= 4096 && n_tapers >= 1 && n_tapers <= 16, "bad sizes");
  EEG_CHECK(n_bins >= 1 && n_bins <= L / 2 + 1, "n_bins %d outside the one-sided spectrum [1, %d]", n_bins, L / 2 + 1);
  float w2[16]; double den = 0.0;
  for (int t = 0; t < n_tapers; t++) { w2[t] = weights_host[t] * weights_host[t]; den += w2[t]; }
  float* w2d = (float*)((char*)ctx->scratch + (4u << 20));         // 64 bytes of the context scratch (main-stream only)
  HIP_TRY(hipMemcpyAsync(w2d, w2, sizeof(float) * n_tapers, hipMemcpyHostToDevice, ctx->stream));
  const size_t lds = sizeof(float) * ((L + 1) & ~1) + sizeof(float2) * L;
  hipLaunchKernelGGL(psd_kernel, dim3(B), dim3(NT), lds, ctx->stream, x, tapers, w2d, psd, L, n_tapers, n_bins, (float)(1.0 / (den * (double)sfreq)));
  LAUNCH_CHECK();
  return 0;
}
