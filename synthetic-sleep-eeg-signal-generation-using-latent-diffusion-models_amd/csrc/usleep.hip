// U-Sleep feature extractor + feature statistics for the FID of /root/reference/src/compute_fid.py:341-419
// (model: /root/reference/src/models/usleep.py:20-287).  Evaluation only: forward, fp32, (B, C, L) layout as the
// reference holds it -- the channel counts are 2, 6, 9, 11, ... 214, 302 (nothing divides by 4 or 8) and the lengths run
// 3000, 1500, 750, 375, 188, ... 3, 2, 1, so the contraction is a direct convolution over flattened (sample, position) rows, not an MFMA
// GEMM: 6.7 MMAC per window through the encoder, HBM / launch bound.
//
//   usleep_conv_kernel<K>   conv1d 'same' (odd K: symmetric; K = 2: the extra zero on the RIGHT, as torch pads) over a virtual input
//                           [A (optionally nearest x2) | skip], both cropped to the common length (usleep.py:5-17,91-96) -> + bias -> ELU
//                           -> eval: folded BatchNorm affine; train: per-channel sum / sum-of-squares (fp64) for the batch statistics
//   usleep_bn_*             fold (eval: gamma / sqrt(running_var + eps) ...), finalize (train: batch statistics -> affine, running
//                           statistics with momentum 0.1 and the unbiased variance, num_batches_tracked), apply
//   usleep_maxpool_kernel   ConstantPad1d(1, 0) on BOTH sides when the length is odd, then MaxPool1d(2) (usleep.py:44-51): the zeros take
//                           part in the max
//   usleep_clf_kernel       conv1 -> tanh -> AvgPool1d(input_size) -> conv1 -> ELU -> conv1 (usleep.py:218-247)
//   feat_accumulate_kernel  sum x and sum x x^T of a feature batch in fp64 (mean / unbiased covariance of the Frechet distance; the
//                           302 x 302 matrix square root is host linear algebra, as in the reference's FIDMetric)
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "net.h"

namespace {
constexpr int UT = 256;            // 4 waves: wave = channel group (4 output channels), lane = one of 64 flattened (sample, position) rows
constexpr int UROWS = 64;
constexpr int UCO = 16;            // output channels per block
constexpr int UCI = 32;            // input channels per LDS weight stage
constexpr int UKMAX = 16;

struct ConvSrc {
  const float* a; int Ca, La, ups;     // channels [0, Ca): a[b][ci][ups ? p / 2 : p], p < L
  const float* b2; int Cb, Lb;         // channels [Ca, Ca + Cb): b2[b][ci - Ca][p]
  int L;                               // common (cropped) length of the virtual input == output length
};

__device__ __forceinline__ float elu_f(float v) { return v > 0.f ? v : expm1f(v); }

template <int KT>
__global__ __launch_bounds__(UT) void usleep_conv_kernel(ConvSrc s, const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ fold, float* __restrict__ y, double* __restrict__ sums,
                                                         long rows, int Cout, int kk, int left) {
  __shared__ float ws[UCI * UKMAX * UCO];                 // [ci][k][co16]: the four weights a thread needs are one 16-byte LDS read
  const int K = KT ? KT : kk;
  const int Cin = s.Ca + s.Cb;
  const int lane = threadIdx.x & 63, cg = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * UROWS + lane;
  const int co_blk = blockIdx.y * UCO;
  const bool live = row < rows;
  const int b = live ? (int)(row / s.L) : 0, l = live ? (int)(row % s.L) : 0;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ci0 = 0; ci0 < Cin; ci0 += UCI) {
    const int nci = Cin - ci0 < UCI ? Cin - ci0 : UCI;
    __syncthreads();
    for (int i = threadIdx.x; i < nci * K * UCO; i += UT) {
      const int co = i % UCO, ck = i / UCO, c = ck / K, k = ck % K;
      ws[i] = co_blk + co < Cout ? w[((long)(co_blk + co) * Cin + ci0 + c) * K + k] : 0.f;
    }
    __syncthreads();
    if (!live) continue;
    for (int c = 0; c < nci; c++) {
      const int ci = ci0 + c;
      const float* src; int ups = 0;
      if (ci < s.Ca) { src = s.a + ((long)b * s.Ca + ci) * s.La; ups = s.ups; }
      else src = s.b2 + ((long)b * s.Cb + (ci - s.Ca)) * s.Lb;
#pragma unroll
      for (int k = 0; k < (KT ? KT : UKMAX); k++) {
        if (!KT && k >= K) break;
        const int p = l + k - left;
        const float xv = (p >= 0 && p < s.L) ? src[ups ? (p >> 1) : p] : 0.f;
        const float4 wv = *(const float4*)&ws[(c * K + k) * UCO + cg * 4];
        acc[0] = fmaf(wv.x, xv, acc[0]); acc[1] = fmaf(wv.y, xv, acc[1]); acc[2] = fmaf(wv.z, xv, acc[2]); acc[3] = fmaf(wv.w, xv, acc[3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int co = co_blk + cg * 4 + j;                    // wave-uniform
    if (co >= Cout) break;
    float v = 0.f;
    if (live) {
      v = elu_f(acc[j] + bias[co]);
      if (fold) v = fmaf(v, fold[co], fold[Cout + co]);
      y[((long)b * Cout + co) * s.L + l] = v;
    }
    if (sums) {                                            // train-mode BatchNorm: the wave's 64 rows all belong to channel `co`
      const double sv = wave_sum_d(live ? (double)v : 0.0), sq = wave_sum_d(live ? (double)v * (double)v : 0.0);
      if (lane == 0) { atomicAdd(&sums[co], sv); atomicAdd(&sums[Cout + co], sq); }
    }
  }
}

struct BnDesc { long w, b, rm, rv, nbt, fold; int C; };

// eval: fold[layer] = {gamma / sqrt(running_var + eps), beta - running_mean * that}
__global__ void usleep_bn_fold_kernel(const BnDesc* __restrict__ tab, const float* __restrict__ params, const float* __restrict__ buffers,
                                      float* __restrict__ fold, float eps) {
  const BnDesc d = tab[blockIdx.x];
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float sc = params[d.w + c] / sqrtf(buffers[d.rv + c] + eps);
    fold[d.fold + c] = sc; fold[d.fold + d.C + c] = params[d.b + c] - buffers[d.rm + c] * sc;
  }
}
// train: batch statistics of one layer -> affine; running statistics (momentum, unbiased variance); sums re-zeroed for the next layer
__global__ void usleep_bn_finalize_kernel(BnDesc d, const float* __restrict__ params, float* __restrict__ buffers, float* __restrict__ fold,
                                          double* __restrict__ sums, double n, float eps, float momentum) {
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const double mean = sums[c] / n;
    double var = sums[d.C + c] / n - mean * mean; if (var < 0.0) var = 0.0;
    const float sc = params[d.w + c] / sqrtf((float)var + eps);
    fold[d.fold + c] = sc; fold[d.fold + d.C + c] = params[d.b + c] - (float)mean * sc;
    const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
    buffers[d.rm + c] = (1.f - momentum) * buffers[d.rm + c] + momentum * (float)mean;
    buffers[d.rv + c] = (1.f - momentum) * buffers[d.rv + c] + momentum * (float)unb;
    sums[c] = 0.0; sums[d.C + c] = 0.0;
  }
  if (threadIdx.x == 0) buffers[d.nbt] += 1.f;
}
__global__ void usleep_affine_kernel(float* __restrict__ y, const float* __restrict__ fold, long n, int C, int L) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i / L) % C);
    y[i] = fmaf(y[i], fold[c], fold[C + c]);
  }
}
__global__ void usleep_maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, long nrows, int L, int Lo) {
  const int odd = L & 1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nrows * Lo; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Lo; const int o = (int)(i % Lo);
    const int p0 = 2 * o - odd, p1 = p0 + 1;             // positions in the un-padded row; out of range = the zero pad
    const float a = (p0 >= 0 && p0 < L) ? x[r * L + p0] : 0.f, b = (p1 >= 0 && p1 < L) ? x[r * L + p1] : 0.f;
    y[i] = fmaxf(a, b);
  }
}
// one block per (sample, pooling window): mean over the window of tanh(W0 x + b0), then the two 1x1 convs on that vector
__global__ __launch_bounds__(UT) void usleep_clf_kernel(const float* __restrict__ x, const float* __restrict__ w0, const float* __restrict__ b0,
                                                        const float* __restrict__ w3, const float* __restrict__ b3, const float* __restrict__ w5,
                                                        const float* __restrict__ b5, float* __restrict__ y, int C1, int L, int win, int S, int ncls) {
  __shared__ float sw[16 * 16 + 16];
  __shared__ double red[4][16];
  const int b = blockIdx.x / S, sidx = blockIdx.x % S;
  for (int i = threadIdx.x; i < C1 * C1; i += UT) sw[i] = w0[i];
  for (int i = threadIdx.x; i < C1; i += UT) sw[256 + i] = b0[i];
  __syncthreads();
  double acc[16];
  for (int c = 0; c < 16; c++) acc[c] = 0.0;
  for (int p = sidx * win + threadIdx.x; p < (sidx + 1) * win; p += UT) {
    float xv[16];
    for (int c = 0; c < C1; c++) xv[c] = x[((long)b * C1 + c) * L + p];
    for (int co = 0; co < C1; co++) {
      float v = sw[256 + co];
      for (int c = 0; c < C1; c++) v = fmaf(sw[co * C1 + c], xv[c], v);
      acc[co] += (double)tanhf(v);
    }
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int co = 0; co < C1; co++) { const double t = wave_sum_d(acc[co]); if (lane == 0) red[wv][co] = t; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float m[16], z[16];
    for (int co = 0; co < C1; co++) m[co] = (float)((red[0][co] + red[1][co] + red[2][co] + red[3][co]) / (double)win);
    for (int k = 0; k < ncls; k++) { float v = b3[k]; for (int c = 0; c < C1; c++) v = fmaf(w3[k * C1 + c], m[c], v); z[k] = elu_f(v); }
    for (int k = 0; k < ncls; k++) { float v = b5[k]; for (int c = 0; c < ncls; c++) v = fmaf(w5[k * ncls + c], z[c], v); y[((long)b * ncls + k) * S + sidx] = v; }
  }
}

// sum[i] += sum_n f[n][i];  outer[i][j] += sum_n f[n][i] f[n][j]   (fp64; one 16 x 16 tile of `outer` per block, rows split over blockIdx.z)
__global__ __launch_bounds__(UT) void feat_accumulate_kernel(const float* __restrict__ f, long N, int D, double* __restrict__ sum, double* __restrict__ outer,
                                                             long rows_per_z) {
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4), j = blockIdx.y * 16 + (threadIdx.x & 15);
  const long n0 = (long)blockIdx.z * rows_per_z, n1 = n0 + rows_per_z < N ? n0 + rows_per_z : N;
  if (i >= D || j >= D) return;
  double acc = 0.0, s = 0.0;
  for (long n = n0; n < n1; n++) { const double a = (double)f[n * D + i]; acc = fma(a, (double)f[n * D + j], acc); s += a; }
  atomicAdd(&outer[(long)i * D + j], acc);
  if (blockIdx.y == 0 && (threadIdx.x & 15) == 0) atomicAdd(&sum[i], s);
}
}  // namespace

struct UConv { long w, b; int cin, cout, k; };
struct eegldm_usleep {
  eegldm_ctx* ctx = nullptr;
  eegldm_usleep_cfg cfg;
  std::vector<int> ch;
  std::vector<Entry> entries; std::vector<int> kinds;     // reference state_dict order; kind 0 = parameter, 1 = buffer
  long nparams = 0, nbuffers = 0, nfold = 0;
  float* params = nullptr; float* buffers = nullptr;
  std::vector<UConv> enc_c, pre_c, post_c; UConv bot_c, clf0, clf3, clf5;
  std::vector<BnDesc> bns;                                  // order: encoder 0..d-1, bottom, decoder i: preskip, postskip
  BnDesc* d_bns = nullptr; float* fold = nullptr; double* sums = nullptr;
  Arena arena;
  ~eegldm_usleep() { if (d_bns) (void)hipFree(d_bns); if (fold) (void)hipFree(fold); if (sums) (void)hipFree(sums); }
};

namespace {
void add_param(eegldm_usleep* u, const std::string& name, long off, int ndim, int s0, int s1 = 0, int s2 = 0, int kind = 0) {
  Entry e; e.name = name; e.offset = off; e.ndim = ndim; e.shape[0] = s0; e.shape[1] = s1; e.shape[2] = s2;
  e.numel = ndim == 0 ? 1 : (long)s0 * (ndim > 1 ? s1 : 1) * (ndim > 2 ? s2 : 1);
  u->entries.push_back(e); u->kinds.push_back(kind);
}
UConv add_conv(eegldm_usleep* u, const std::string& p, int cout, int cin, int k) {
  UConv c; c.cin = cin; c.cout = cout; c.k = k;
  c.w = u->nparams; add_param(u, p + ".weight", c.w, 3, cout, cin, k); u->nparams += (long)cout * cin * k;
  c.b = u->nparams; add_param(u, p + ".bias", c.b, 1, cout); u->nparams += cout;
  return c;
}
void add_bn(eegldm_usleep* u, const std::string& p, int C) {
  BnDesc d; d.C = C;
  d.w = u->nparams; add_param(u, p + ".weight", d.w, 1, C); u->nparams += C;
  d.b = u->nparams; add_param(u, p + ".bias", d.b, 1, C); u->nparams += C;
  d.rm = u->nbuffers; add_param(u, p + ".running_mean", d.rm, 1, C, 0, 0, 1); u->nbuffers += C;
  d.rv = u->nbuffers; add_param(u, p + ".running_var", d.rv, 1, C, 0, 0, 1); u->nbuffers += C;
  d.nbt = u->nbuffers; add_param(u, p + ".num_batches_tracked", d.nbt, 0, 0, 0, 0, 1); u->nbuffers += 1;
  d.fold = u->nfold; u->nfold += 2 * C;
  u->bns.push_back(d);
}

int launch_conv(eegldm_usleep* u, const ConvSrc& s, const UConv& c, int left, const float* fold, float* y, double* sums, int B) {
  const long rows = (long)B * s.L;
  if (rows == 0) return 0;
  const dim3 grid((unsigned)((rows + UROWS - 1) / UROWS), (unsigned)((c.cout + UCO - 1) / UCO));
  const float* w = u->params + c.w; const float* bias = u->params + c.b;
#define ULAUNCH(KT) hipLaunchKernelGGL((usleep_conv_kernel<KT>), grid, dim3(UT), 0, u->ctx->stream, s, w, bias, fold, y, sums, rows, c.cout, c.k, left)
  switch (c.k) {
    case 1: ULAUNCH(1); break;
    case 2: ULAUNCH(2); break;
    case 3: ULAUNCH(3); break;
    case 5: ULAUNCH(5); break;
    case 7: ULAUNCH(7); break;
    case 9: ULAUNCH(9); break;
    case 11: ULAUNCH(11); break;
    default: ULAUNCH(0); break;
  }
#undef ULAUNCH
  LAUNCH_CHECK();
  return 0;
}
// conv -> ELU -> BatchNorm of one layer; bn_idx indexes u->bns
int conv_elu_bn(eegldm_usleep* u, const ConvSrc& s, const UConv& c, int left, int bn_idx, float* y, int B, int training) {
  const BnDesc& d = u->bns[bn_idx];
  if (!training) return launch_conv(u, s, c, left, u->fold + d.fold, y, nullptr, B);
  EEG_TRY(launch_conv(u, s, c, left, nullptr, y, u->sums, B));
  const long n = (long)B * c.cout * s.L;
  hipLaunchKernelGGL(usleep_bn_finalize_kernel, dim3(1), dim3(256), 0, u->ctx->stream, d, u->params, u->buffers, u->fold, u->sums,
                     (double)B * (double)s.L, 1e-5f, 0.1f);
  LAUNCH_CHECK();
  if (n > 0) {
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(usleep_affine_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, u->ctx->stream, y, u->fold + d.fold, n, c.cout, s.L);
    LAUNCH_CHECK();
  }
  return 0;
}
}  // namespace

extern "C" int eegldm_usleep_create(eegldm_ctx* ctx, const eegldm_usleep_cfg* cfg, eegldm_usleep** out) {
  EEG_CHECK(ctx && cfg && out, "null argument");
  EEG_CHECK(cfg->in_chans >= 1 && cfg->depth >= 1 && cfg->depth <= 16 && cfg->n_classes >= 1 && cfg->n_classes <= 16, "bad U-Sleep configuration");
  EEG_CHECK(cfg->kernel_size >= 1 && cfg->kernel_size <= UKMAX && (cfg->kernel_size & 1), "time_conv_size must be odd and <= %d (usleep.py:157-163)", UKMAX);
  EEG_CHECK(cfg->input_size >= 1, "input_size");
  eegldm_usleep* u = new eegldm_usleep();
  u->ctx = ctx; u->cfg = *cfg;
  // usleep.py:165-172: channels c_0 = in_chans, c_{i+1} = int(f_i * sqrt(complexity_factor)), f_{i+1} = int(f_i * sqrt(2)) (float64 there)
  u->ch.push_back(cfg->in_chans);
  int f = cfg->n_time_filters;
  for (int i = 0; i <= cfg->depth; i++) { u->ch.push_back((int)((double)f * std::sqrt((double)cfg->complexity_factor))); f = (int)((double)f * std::sqrt(2.0)); }
  const int d = cfg->depth, k = cfg->kernel_size;
  const std::vector<int>& ch = u->ch;
  if (ch[1] > 16) { delete u; EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "classifier width %d > 16", ch[1]); }
  for (int i = 0; i < d; i++) {
    const std::string p = "encoder." + std::to_string(i) + ".block_prepool";
    u->enc_c.push_back(add_conv(u, p + ".0", ch[i + 1], ch[i], k)); add_bn(u, p + ".2", ch[i + 1]);
  }
  u->bot_c = add_conv(u, "bottom.0", ch[d + 1], ch[d], k); add_bn(u, "bottom.2", ch[d + 1]);
  for (int i = 0; i < d; i++) {
    const int ci = ch[d + 1 - i], co = ch[d - i];
    const std::string p = "decoder." + std::to_string(i);
    u->pre_c.push_back(add_conv(u, p + ".block_preskip.1", co, ci, 2)); add_bn(u, p + ".block_preskip.3", co);
    u->post_c.push_back(add_conv(u, p + ".block_postskip.0", co, (cfg->with_skip_connection ? 2 : 1) * co, k)); add_bn(u, p + ".block_postskip.2", co);
  }
  u->clf0 = add_conv(u, "clf.0", ch[1], ch[1], 1);
  u->clf3 = add_conv(u, "clf.3", cfg->n_classes, ch[1], 1);
  u->clf5 = add_conv(u, "clf.5", cfg->n_classes, cfg->n_classes, 1);
  int maxc = 0; for (int c : ch) maxc = c > maxc ? c : maxc;
  auto fail = [&](const char* what) { delete u; eegldm_set_error(std::string("eegldm_usleep_create: ") + what); return EEGLDM_ERR_HIP; };
  if (hipMalloc(&u->d_bns, sizeof(BnDesc) * u->bns.size()) != hipSuccess) return fail("hipMalloc");
  if (hipMemcpy(u->d_bns, u->bns.data(), sizeof(BnDesc) * u->bns.size(), hipMemcpyHostToDevice) != hipSuccess) return fail("hipMemcpy");
  if (hipMalloc(&u->fold, sizeof(float) * u->nfold) != hipSuccess) return fail("hipMalloc");
  if (hipMalloc(&u->sums, sizeof(double) * 2 * maxc) != hipSuccess) return fail("hipMalloc");
  if (hipMemset(u->sums, 0, sizeof(double) * 2 * maxc) != hipSuccess) return fail("hipMemset");
  *out = u;
  return 0;
}
extern "C" int eegldm_usleep_destroy(eegldm_usleep* u) { delete u; return 0; }
extern "C" long eegldm_usleep_num_params(const eegldm_usleep* u) { return u ? u->nparams : 0; }
extern "C" long eegldm_usleep_num_buffers(const eegldm_usleep* u) { return u ? u->nbuffers : 0; }
extern "C" int eegldm_usleep_num_entries(const eegldm_usleep* u) { return u ? (int)u->entries.size() : 0; }
extern "C" int eegldm_usleep_channel(const eegldm_usleep* u, int i) { return (u && i >= 0 && i < (int)u->ch.size()) ? u->ch[i] : -1; }
extern "C" int eegldm_usleep_entry(const eegldm_usleep* u, int i, char* name, int cap, int* kind, long* offset, long* numel, int* ndim, int shape[3]) {
  EEG_CHECK(u && i >= 0 && i < (int)u->entries.size(), "entry index %d out of range", i);
  const Entry& e = u->entries[i];
  if (name && cap > 0) { strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (kind) *kind = u->kinds[i];
  if (offset) *offset = e.offset;
  if (numel) *numel = e.numel;
  if (ndim) *ndim = e.ndim;
  if (shape) { shape[0] = e.shape[0]; shape[1] = e.shape[1]; shape[2] = e.shape[2]; }
  return 0;
}
extern "C" int eegldm_usleep_bind(eegldm_usleep* u, float* params, float* buffers) {
  EEG_CHECK(u && params && buffers, "null argument");
  u->params = params; u->buffers = buffers;
  return 0;
}

// x (B, in_chans, T) fp32 -> y_pred (B, n_classes, T / input_size), decoder output (B, c_1, T), bottleneck (B, c_{depth+1}, Lb); every
// output is optional, and with neither y_pred nor decoder_out the decoder is not run (the FID feature needs the bottleneck only).
// training != 0: BatchNorm on batch statistics + running-statistics update (what compute_fid.py runs: it never calls model.eval()).
extern "C" int eegldm_usleep_forward(eegldm_usleep* u, const float* x, float* y_pred, float* decoder_out, float* bottom, int B, int T, int training) {
  EEG_CHECK(u && x && u->params && u->buffers, "null argument / unbound parameters");
  EEG_CHECK(y_pred || decoder_out || bottom, "nothing to return");
  EEG_CHECK(B >= 1 && T >= 1, "bad sizes");
  EEG_CHECK(!y_pred || T >= u->cfg.input_size, "T=%d is shorter than the classifier's pooling window %d", T, u->cfg.input_size);
  eegldm_ctx* ctx = u->ctx;
  const int d = u->cfg.depth, k = u->cfg.kernel_size, left = (k - 1) / 2;
  const std::vector<int>& ch = u->ch;
  u->arena.reset();
  auto falloc = [&](long n) { return (float*)u->arena.alloc(sizeof(float) * (size_t)(n > 0 ? n : 1)); };
  if (!training) {
    hipLaunchKernelGGL(usleep_bn_fold_kernel, dim3((unsigned)u->bns.size()), dim3(64), 0, ctx->stream, u->d_bns, u->params, u->buffers, u->fold, 1e-5f);
    LAUNCH_CHECK();
  }
  std::vector<const float*> res(d); std::vector<int> resL(d);
  const float* cur = x; int Lc = T, Cc = ch[0];
  for (int i = 0; i < d; i++) {
    float* y; ALLOC_OR_FAIL(y, falloc((long)B * ch[i + 1] * Lc));
    ConvSrc s = {cur, Cc, Lc, 0, nullptr, 0, 0, Lc};
    EEG_TRY(conv_elu_bn(u, s, u->enc_c[i], left, i, y, B, training));
    res[i] = y; resL[i] = Lc;
    const int Lo = (Lc + ((Lc & 1) ? 2 : 0)) / 2;
    float* p; ALLOC_OR_FAIL(p, falloc((long)B * ch[i + 1] * Lo));
    const long n = (long)B * ch[i + 1] * Lo, blocks = (n + 255) / 256;
    hipLaunchKernelGGL(usleep_maxpool_kernel, dim3((unsigned)(blocks < 4096 ? (blocks < 1 ? 1 : blocks) : 4096)), dim3(256), 0, ctx->stream, y, p, (long)B * ch[i + 1], Lc, Lo);
    LAUNCH_CHECK();
    cur = p; Lc = Lo; Cc = ch[i + 1];
  }
  float* bot; ALLOC_OR_FAIL(bot, falloc((long)B * ch[d + 1] * Lc));
  { ConvSrc s = {cur, Cc, Lc, 0, nullptr, 0, 0, Lc}; EEG_TRY(conv_elu_bn(u, s, u->bot_c, left, d, bot, B, training)); }
  if (bottom) HIP_TRY(hipMemcpyAsync(bottom, bot, sizeof(float) * (size_t)B * ch[d + 1] * Lc, hipMemcpyDeviceToDevice, ctx->stream));
  if (!y_pred && !decoder_out) return 0;
  cur = bot; Cc = ch[d + 1];
  for (int i = 0; i < d; i++) {
    const int co = ch[d - i], Lr = resL[d - 1 - i];
    float* pre; ALLOC_OR_FAIL(pre, falloc((long)B * co * 2 * Lc));
    { ConvSrc s = {cur, Cc, Lc, 1, nullptr, 0, 0, 2 * Lc}; EEG_TRY(conv_elu_bn(u, s, u->pre_c[i], 0, d + 1 + 2 * i, pre, B, training)); }
    int n = 2 * Lc;
    ConvSrc s2 = {pre, co, 2 * Lc, 0, nullptr, 0, 0, n};
    if (u->cfg.with_skip_connection) { n = n < Lr ? n : Lr; s2.b2 = res[d - 1 - i]; s2.Cb = co; s2.Lb = Lr; s2.L = n; }   // _crop_tensors_to_match + cat([x, residual])
    float* post; ALLOC_OR_FAIL(post, falloc((long)B * co * n));
    EEG_TRY(conv_elu_bn(u, s2, u->post_c[i], left, d + 2 + 2 * i, post, B, training));
    cur = post; Lc = n; Cc = co;
  }
  if (decoder_out) HIP_TRY(hipMemcpyAsync(decoder_out, cur, sizeof(float) * (size_t)B * Cc * Lc, hipMemcpyDeviceToDevice, ctx->stream));
  if (y_pred) {
    const int S = Lc / u->cfg.input_size;
    EEG_CHECK(S >= 1, "decoder output length %d is shorter than the pooling window %d", Lc, u->cfg.input_size);
    const float* P = u->params;
    hipLaunchKernelGGL(usleep_clf_kernel, dim3((unsigned)(B * S)), dim3(UT), 0, ctx->stream, cur, P + u->clf0.w, P + u->clf0.b, P + u->clf3.w, P + u->clf3.b,
                       P + u->clf5.w, P + u->clf5.b, y_pred, ch[1], Lc, u->cfg.input_size, S, u->cfg.n_classes);
    LAUNCH_CHECK();
  }
  return 0;
}

// Streaming first and second moments of a feature batch: sum[D] += sum_n f[n], outer[D][D] += sum_n f[n] f[n]^T (fp64 device buffers,
// zeroed by the caller before the first batch).  mean = sum / N, covariance = (outer - N mean mean^T) / (N - 1): the inputs of the
// Frechet distance (compute_fid.py:412-414; FIDMetric's torch.mean / _cov(rowvar=False)).
extern "C" int eegldm_feature_moments(eegldm_ctx* ctx, const float* feats, long N, int D, double* sum, double* outer) {
  EEG_CHECK(ctx && feats && sum && outer && N >= 1 && D >= 1, "bad argument");
  const int nt = (D + 15) / 16;
  long nz = N / 64; if (nz < 1) nz = 1; if (nz > 64) nz = 64;
  const long rpz = (N + nz - 1) / nz;
  hipLaunchKernelGGL(feat_accumulate_kernel, dim3(nt, nt, (unsigned)((N + rpz - 1) / rpz)), dim3(UT), 0, ctx->stream, feats, N, D, sum, outer, rpz);
  LAUNCH_CHECK();
  return 0;
}
