// HBM-bound kernels of the AutoencoderKL / PatchDiscriminator train step that are not
// shared with the UNet: BatchNorm1d(train)+LeakyReLU, LeakyReLU, nearest x2 up-sampling,
// KL reparameterisation, L1 and least-squares adversarial losses.
// Reference call sites: /root/reference/src/train_autoencoderkl.py:204-234; MONAI
// PatchDiscriminator / PatchAdversarialLoss / AutoencoderKL.sampling.
#include "common.h"
#include "internal.h"

namespace {
constexpr int NT = 256;
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)
inline int grid1d(long n, eegldm_ctx* ctx) {
  long blocks = (n + NT - 1) / NT, cap = (long)ctx->num_cu * 16;
  if (blocks < 1) blocks = 1;
  return (int)(blocks < cap ? blocks : cap);
}

// ------------------------------------------------------------------ BatchNorm statistics
// sums[c][0..1] += sum x, sum x^2 over a chunk of rows (double atomics); grid (ceil(C/64), RSPLIT)
template <typename T>
__global__ __launch_bounds__(NT) void bn_stats_kernel(const T* __restrict__ x, long ldx, double* __restrict__ sums, long rows, int C, long rows_per_block) {
  __shared__ float r1[4][64], r2[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
  const long l0 = (long)blockIdx.y * rows_per_block, l1 = min(rows, l0 + rows_per_block);
  float s1 = 0.f, s2 = 0.f;
  if (c < C) for (long l = l0 + ry; l < l1; l += 4) { const float v = ld_f32(x + l * ldx + c); s1 += v; s2 += v * v; }
  r1[ry][threadIdx.x & 63] = s1; r2[ry][threadIdx.x & 63] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    const int i = threadIdx.x;
    atomicAdd(&sums[2 * c], (double)(r1[0][i] + r1[1][i] + r1[2][i] + r1[3][i]));
    atomicAdd(&sums[2 * c + 1], (double)(r2[0][i] + r2[1][i] + r2[2][i] + r2[3][i]));
  }
}
// stats[c] = (mean, rstd); running stats: momentum 0.1, unbiased variance (torch BatchNorm1d defaults)
// (repeats: that many momentum updates with the same batch statistics -- eegldm_ctx::bn_running_repeats)
__device__ __forceinline__ void bn_finalize_one(double s1, double s2, int c, float* __restrict__ stats, float* __restrict__ rmean, float* __restrict__ rvar,
                                                float* __restrict__ nbt, double n, float eps, float momentum, int repeats) {
  const double mean = s1 / n;
  double var = s2 / n - mean * mean;
  if (var < 0) var = 0;
  stats[2 * c] = (float)mean; stats[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
  if (rmean) {
    float rm = rmean[c], rv = rvar[c];
    const float ub = (float)(var * n / (n > 1 ? n - 1 : 1));
    for (int k = 0; k < repeats; k++) { rm = (1.0f - momentum) * rm + momentum * (float)mean; rv = (1.0f - momentum) * rv + momentum * ub; }
    rmean[c] = rm; rvar[c] = rv;
    if (c == 0 && nbt) nbt[0] += (float)repeats;
  }
}
__global__ void bn_finalize_kernel(const double* __restrict__ sums, float* __restrict__ stats, float* __restrict__ rmean,
                                   float* __restrict__ rvar, float* __restrict__ nbt, int C, double n, float eps, float momentum, int repeats) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  bn_finalize_one(sums[2 * c], sums[2 * c + 1], c, stats, rmean, rvar, nbt, n, eps, momentum, repeats);
}
// fold of the per-block partial rows parts[nparts][2 C] (interleaved sum, sum of squares) AND the finalisation in one launch: a 1024-thread block
// owns 64 consecutive values = 32 channels, 16 row lanes add their rows in fp64, the lanes meet in LDS in a fixed order (no atomics, no sum
// area), the even lanes finalise their channel.  Replaces bn_fold_kernel + bn_finalize_kernel on the forward path (two ~5 us launches per layer).
__global__ __launch_bounds__(1024) void bn_fold_finalize_kernel(const float* __restrict__ parts, int nparts, int C, float* __restrict__ stats, float* __restrict__ rmean,
                                                                float* __restrict__ rvar, float* __restrict__ nbt, double n, float eps, float momentum, int repeats) {
  __shared__ double red[16][64];
  const int col = threadIdx.x & 63, seg = threadIdx.x >> 6, i = blockIdx.x * 64 + col, n2c = 2 * C;
  double s = 0.0;
  if (i < n2c) {
#pragma unroll 4
    for (int r = seg; r < nparts; r += 16) s += (double)parts[(size_t)r * n2c + i];
  }
  red[seg][col] = s;
  __syncthreads();
  if (seg == 0 && (col & 1) == 0 && i < n2c) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) { s1 += red[k][col]; s2 += red[k][col + 1]; }
    bn_finalize_one(s1, s2, i >> 1, stats, rmean, rvar, nbt, n, eps, momentum, repeats);
  }
}
__global__ void bn_eval_stats_kernel(const float* __restrict__ rmean, const float* __restrict__ rvar, float* __restrict__ stats, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) { stats[2 * c] = rmean[c]; stats[2 * c + 1] = rsqrtf(rvar[c] + eps); }
}
// y = lrelu(gamma * (x - mean) * rstd + beta, slope); gamma == null -> plain lrelu(x)
template <typename T>
__global__ void bn_lrelu_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ stats, T* __restrict__ y, long ldy, long rows, int C, float slope) {
  GRID_STRIDE(i, rows * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    float z = ld_f32(x + r * ldx + c);
    if (gamma) z = (z - stats[2 * c]) * stats[2 * c + 1] * gamma[c] + beta[c];
    st_f32(y + r * ldy + c, z > 0.f ? z : slope * z);
  }
}
// per-channel sums of dz and dz*xhat (double atomics) -> sums[c][0..1]
template <typename T>
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stats,
                                                           const T* __restrict__ dy, long lddy, double* __restrict__ sums, long rows, int C,
                                                           long rows_per_block, float slope) {
  __shared__ float r1[4][64], r2[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), ry = threadIdx.x >> 6;
  const long l0 = (long)blockIdx.y * rows_per_block, l1 = min(rows, l0 + rows_per_block);
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float mean = stats[2 * c], rstd = stats[2 * c + 1], ga = gamma[c], be = beta[c];
    for (long l = l0 + ry; l < l1; l += 4) {
      const float xh = (ld_f32(x + l * ldx + c) - mean) * rstd;
      float dz = ld_f32(dy + l * lddy + c);
      if (ga * xh + be <= 0.f) dz *= slope;
      s1 += dz; s2 += dz * xh;
    }
  }
  r1[ry][threadIdx.x & 63] = s1; r2[ry][threadIdx.x & 63] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    const int i = threadIdx.x;
    atomicAdd(&sums[2 * c], (double)(r1[0][i] + r1[1][i] + r1[2][i] + r1[3][i]));
    atomicAdd(&sums[2 * c + 1], (double)(r2[0][i] + r2[1][i] + r2[2][i] + r2[3][i]));
  }
}
// dx = gamma*rstd*(dz - S1/n - xhat*S2/n); plain lrelu when gamma == null; dgamma/dbeta accumulated by one block
template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ stats, const T* __restrict__ dy, long lddy, const double* __restrict__ sums,
                                    T* __restrict__ dx, long lddx, float* __restrict__ dgamma, float* __restrict__ dbeta, long rows, int C, float slope) {
  const float inv_n = 1.0f / (float)rows;
  if (gamma && dgamma && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) { atomicAdd(&dbeta[c], (float)sums[2 * c]); atomicAdd(&dgamma[c], (float)sums[2 * c + 1]); }
  GRID_STRIDE(i, rows * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    const float xv = ld_f32(x + r * ldx + c);
    float dz = ld_f32(dy + r * lddy + c);
    if (!gamma) { st_f32(dx + r * lddx + c, xv > 0.f ? dz : slope * dz); continue; }
    const float rstd = stats[2 * c + 1], ga = gamma[c], xh = (xv - stats[2 * c]) * rstd;
    if (ga * xh + beta[c] <= 0.f) dz *= slope;
    st_f32(dx + r * lddx + c, ga * rstd * (dz - (float)sums[2 * c] * inv_n - xh * (float)sums[2 * c + 1] * inv_n));
  }
}

// ------------------------------------------------------------------ 4-channel-vector BatchNorm kernels (C % 4 == 0, ld % 4 == 0)
// Thread (tx, ty): tx = 4-channel column (fixed for the thread: scale/shift hoisted, no integer division per element),
// ty = row lane; a block walks `rows_per_block` rows.  The scalar kernels above remain for odd shapes.
template <typename T> __device__ __forceinline__ void ldv4(const T* p, float v[4]);
template <> __device__ __forceinline__ void ldv4<float>(const float* p, float v[4]) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
template <> __device__ __forceinline__ void ldv4<bf16_t>(const bf16_t* p, float v[4]) {
  const uint2 t = *(const uint2*)p;
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u); v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void ldv4<f16_t>(const f16_t* p, float v[4]) {
  const uint2 t = *(const uint2*)p;
  v[0] = w16_lo<f16_t>(t.x); v[1] = w16_hi<f16_t>(t.x); v[2] = w16_lo<f16_t>(t.y); v[3] = w16_hi<f16_t>(t.y);
}
template <typename T> __device__ __forceinline__ void stv4(T* p, const float v[4]);
template <> __device__ __forceinline__ void stv4<f16_t>(f16_t* p, const float v[4]) { uint2 t; t.x = pack_f16x2(v[0], v[1]); t.y = pack_f16x2(v[2], v[3]); *(uint2*)p = t; }
template <> __device__ __forceinline__ void stv4<float>(float* p, const float v[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void stv4<bf16_t>(bf16_t* p, const float v[4]) { uint2 t; t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]); *(uint2*)p = t; }

struct BnMap { int TX, TY, tx, ty; bool act; };
__device__ __forceinline__ BnMap bnmap(int C) {
  BnMap m; const int ncols = C / 4;
  m.TX = ncols < NT ? ncols : NT; m.TY = NT / m.TX;
  m.tx = threadIdx.x % m.TX; m.ty = threadIdx.x / m.TX; m.act = threadIdx.x < m.TX * m.TY;
  return m;
}
// MODE 0: sums x, x^2;  MODE 1: sums dz, dz*xhat (backward)
template <typename T, int MODE>
__global__ __launch_bounds__(NT) void bn_reduce4_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ stats, const T* __restrict__ dy, long lddy,
                                                        void* __restrict__ parts, long rows, int C, long rows_per_block, float slope) {
  __shared__ double red[2 * 1024];   // fp64 LDS atomics: order-independent sums, and faster than contended fp32 LDS atomics (see norm.hip)
  const BnMap m = bnmap(C);
  for (int i = threadIdx.x; i < 2 * C; i += NT) red[i] = 0.0;
  __syncthreads();
  const long l0 = (long)blockIdx.x * rows_per_block, l1 = min(rows, l0 + rows_per_block);
  if (m.act) {
    for (int col = m.tx; col < C / 4; col += m.TX) {
      const int c = col * 4;
      float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
      float mean[4], rstd[4], ga[4], be[4];
      if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 4; j++) { mean[j] = stats[2 * (c + j)]; rstd[j] = stats[2 * (c + j) + 1]; ga[j] = gamma[c + j]; be[j] = beta[c + j]; }
      }
#pragma unroll 4
      for (long l = l0 + m.ty; l < l1; l += m.TY) {
        float v[4]; ldv4<T>(x + l * ldx + c, v);
        if (MODE == 0) {
#pragma unroll
          for (int j = 0; j < 4; j++) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
        } else {
          float d[4]; ldv4<T>(dy + l * lddy + c, d);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const float xh = (v[j] - mean[j]) * rstd[j];
            const float dz = (ga[j] * xh + be[j] <= 0.f) ? d[j] * slope : d[j];
            s1[j] += dz; s2[j] += dz * xh;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) { atomicAdd(&red[2 * (c + j)], (double)s1[j]); atomicAdd(&red[2 * (c + j) + 1], (double)s2[j]); }
    }
  }
  __syncthreads();
  // written partials + a folding pass: a thousand blocks adding atomically into the same 2C addresses serialise in L2
  float* part = (float*)parts + (size_t)blockIdx.x * 2 * C;
  for (int i = threadIdx.x; i < 2 * C; i += NT) part[i] = (float)red[i];
}
__global__ void bn_fold_kernel(const float* __restrict__ parts, int nparts, int n2c, double* __restrict__ sums, double* __restrict__ zero_next, int n_zero) {
  __shared__ double red[NT];
  // the OTHER sum area (used by the previous BatchNorm call, whose consumers precede this kernel in stream order) is re-zeroed here
  // for the next call: no memset launch per call
  if (blockIdx.y == 0 && (threadIdx.x >> 6) == 0) for (int j = blockIdx.x * 64 + (threadIdx.x & 63); j < n_zero; j += gridDim.x * 64) zero_next[j] = 0.0;
  const int i = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6;
  const int per = (nparts + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(nparts, r0 + per);    // sums pre-zeroed
  double s = 0.0;
  if (i < n2c) for (int r = r0 + seg; r < r1; r += NT / 64) s += (double)parts[(size_t)r * n2c + i];
  red[threadIdx.x] = s;
  __syncthreads();
  if (seg == 0 && i < n2c) atomicAdd(&sums[i], red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}
// deterministic mode: the same fold with a fixed two-level shape (16 lanes per element add the rows q, q + 16, ... in order, then the lane
// sums in lane order) and no atomics -- fp64 sums of fp32 partials are order-independent only as long as nothing is rounded
__global__ __launch_bounds__(NT) void bn_fold_det_kernel(const float* __restrict__ parts, int nparts, int n2c, double* __restrict__ sums, double* __restrict__ zero_next, int n_zero) {
  __shared__ double red[16][17];
  for (int j = blockIdx.x * NT + threadIdx.x; j < n_zero; j += gridDim.x * NT) zero_next[j] = 0.0;
  const int e = threadIdx.x & 15, q = threadIdx.x >> 4, i = blockIdx.x * 16 + e;
  double s = 0.0;
  if (i < n2c) {
#pragma unroll 4
    for (int r = q; r < nparts; r += 16) s += (double)parts[(size_t)r * n2c + i];
  }
  red[q][e] = s;
  __syncthreads();
  if (q == 0 && i < n2c) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][e];
    sums[i] = t;
  }
}
// MODE 0: y = lrelu(bn(x));  MODE 1: dx of the same (sums = per-channel S1, S2)
template <typename T, int MODE>
__global__ __launch_bounds__(NT) void bn_apply4_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ stats, const T* __restrict__ dy, long lddy, const double* __restrict__ sums,
                                                       T* __restrict__ out, long ldo, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                       long rows, int C, long rows_per_block, float slope) {
  const BnMap m = bnmap(C);
  if (MODE == 1 && gamma && dgamma && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += NT) { atomicAdd(&dbeta[c], (float)sums[2 * c]); atomicAdd(&dgamma[c], (float)sums[2 * c + 1]); }
  if (!m.act) return;
  const float inv_n = 1.0f / (float)rows;
  const long l0 = (long)blockIdx.x * rows_per_block, l1 = min(rows, l0 + rows_per_block);
  for (int col = m.tx; col < C / 4; col += m.TX) {
    const int c = col * 4;
    float sc[4], sh[4], mean[4], rstd[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (gamma) {
        mean[j] = stats[2 * (c + j)]; rstd[j] = stats[2 * (c + j) + 1];
        sc[j] = gamma[c + j] * rstd[j]; sh[j] = beta[c + j] - mean[j] * sc[j];
        if (MODE == 1) { k1[j] = (float)sums[2 * (c + j)] * inv_n; k2[j] = (float)sums[2 * (c + j) + 1] * inv_n; }
      } else { sc[j] = 1.f; sh[j] = 0.f; mean[j] = 0.f; rstd[j] = 1.f; k1[j] = 0.f; k2[j] = 0.f; }
    }
#pragma unroll 4
    for (long l = l0 + m.ty; l < l1; l += m.TY) {
      float v[4], o[4]; ldv4<T>(x + l * ldx + c, v);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 4; j++) { const float z = v[j] * sc[j] + sh[j]; o[j] = z > 0.f ? z : slope * z; }
      } else {
        float d[4]; ldv4<T>(dy + l * lddy + c, d);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float z = v[j] * sc[j] + sh[j];
          const float dz = z > 0.f ? d[j] : slope * d[j];
          if (gamma) { const float xh = (v[j] - mean[j]) * rstd[j]; o[j] = sc[j] * (dz - k1[j] - xh * k2[j]); }
          else o[j] = dz;
        }
      }
      stv4<T>(out + l * ldo + c, o);
    }
  }
}
inline void bn_split(long rows, eegldm_ctx* ctx, int per_cu, int* blocks, long* rpb) {
  long want = (long)ctx->num_cu * per_cu, maxb = (rows + 15) / 16;
  if (want > maxb) want = maxb;
  if (want < 1) want = 1;
  *rpb = (rows + want - 1) / want;
  *blocks = (int)((rows + *rpb - 1) / *rpb);
}

// ------------------------------------------------------------------ nearest x2 (MONAI Upsample; twin ae_kl.py:27-30)
template <typename T>
__global__ void upsample2_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, long rows_in, int C) {
  GRID_STRIDE(i, rows_in * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    const T v = x[r * ldx + c];
    y[(2 * r) * ldy + c] = v; y[(2 * r + 1) * ldy + c] = v;
  }
}
template <typename T>
__global__ void upsample2_bwd_kernel(const T* __restrict__ dy, long lddy, T* __restrict__ dx, long lddx, long rows_in, int C) {
  GRID_STRIDE(i, rows_in * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    st_f32(dx + r * lddx + c, ld_f32(dy + (2 * r) * lddy + c) + ld_f32(dy + (2 * r + 1) * lddy + c));
  }
}

// ------------------------------------------------------------------ reparameterisation + KL (train_autoencoderkl.py:210-211)
// mu, lv: conv outputs (T, [rows][C]); eps fp32 [rows][C] or null (z = mu).  Writes z (T), sigma (fp32 rows x C),
// accumulates KL = 0.5 * sum(mu^2 + sigma^2 - log sigma^2 - 1) / B into *kl.
template <typename T>
__global__ __launch_bounds__(NT) void reparam_kernel(const T* __restrict__ mu, const T* __restrict__ lv, const float* __restrict__ eps,
                                                     T* __restrict__ z, float* __restrict__ sigma, float* __restrict__ kl, long n, float inv_B, float* __restrict__ parts) {
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float m = ld_f32(mu + i);
    const float l = fminf(20.0f, fmaxf(-30.0f, ld_f32(lv + i)));
    const float sg = expf(0.5f * l);
    sigma[i] = sg;
    st_f32(z + i, eps ? m + eps[i] * sg : m);
    s += 0.5f * (m * m + sg * sg - logf(sg * sg) - 1.0f);
  }
  if (kl) {
    s = wave_sum(s);
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { const float v = (red[0] + red[1] + red[2] + red[3]) * inv_B; if (parts) parts[blockIdx.x] = v; else atomicAdd(kl, v); }
  }
}
// dmu = dz + klw*mu/B [+ dmu_ext] ; dlv = [ -30 < lv < 20 ] * (dz*eps + klw*(sigma - 1/sigma)/B [+ dsigma_ext]) * sigma/2
// dmu_ext / dsg_ext (nullable): gradients a caller's own loss put on the returned z_mu / z_sigma tensors, fp32 (B, lat, Ll) -- the
// autograd bridge (eegldm_aekl_backward_ex); element i of the NLC tensors is (row = i / lat, channel = i % lat)
template <typename T>
__global__ void reparam_bwd_kernel(const T* __restrict__ mu, const T* __restrict__ lv, const float* __restrict__ eps, const float* __restrict__ sigma,
                                   const T* __restrict__ dz, T* __restrict__ dmu, T* __restrict__ dlv, long n, float klw_over_B,
                                   const float* __restrict__ dmu_ext, const float* __restrict__ dsg_ext, int lat, int Ll) {
  GRID_STRIDE(i, n) {
    const float d = dz ? ld_f32(dz + i) : 0.f, m = ld_f32(mu + i), l = ld_f32(lv + i), sg = sigma[i];
    float em = 0.f, es = 0.f;
    if (dmu_ext || dsg_ext) {
      const long row = i / lat; const int c = (int)(i - row * lat); const long b = row / Ll; const int pos = (int)(row - b * Ll);
      const long e = (b * lat + c) * Ll + pos;
      if (dmu_ext) em = dmu_ext[e];
      if (dsg_ext) es = dsg_ext[e];
    }
    st_f32(dmu + i, d + klw_over_B * m + em);
    const float dsg = d * (eps ? eps[i] : 0.f) + klw_over_B * (sg - 1.0f / sg) + es;
    st_f32(dlv + i, (l > -30.0f && l < 20.0f) ? dsg * sg * 0.5f : 0.f);
  }
}

// ------------------------------------------------------------------ losses on fp32 NCL tensors
// L1: loss += w_loss * mean|a-b| ; da += w_grad * sign(a-b)/n
__global__ __launch_bounds__(NT) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ loss,
                                                float* __restrict__ da, long n, float inv_n, float wgrad, int overwrite, float* __restrict__ parts) {
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float d = a[i] - b[i];
    s += fabsf(d);
    if (da) { const float g = wgrad * inv_n * (d > 0.f ? 1.0f : (d < 0.f ? -1.0f : 0.f)); da[i] = overwrite ? g : da[i] + g; }
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float v = (red[0] + red[1] + red[2] + red[3]) * inv_n; if (parts) parts[blockIdx.x] = v; else atomicAdd(loss, v); }
}
// least-squares GAN: a = lrelu(logit, 0.05); loss = mean((a - target)^2); dlogit = w * 2(a-target)/n * lrelu'
__global__ __launch_bounds__(NT) void lsgan_kernel(const float* __restrict__ lg, float target, float* __restrict__ loss, float* __restrict__ dlg,
                                                   long n, float inv_n, float wgrad, float* __restrict__ parts) {
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float x = lg[i], a = x > 0.f ? x : 0.05f * x, d = a - target;
    s += d * d;
    if (dlg) dlg[i] = wgrad * 2.0f * d * inv_n * (x > 0.f ? 1.0f : 0.05f);
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float v = (red[0] + red[1] + red[2] + red[3]) * inv_n; if (parts) parts[blockIdx.x] = v; else atomicAdd(loss, v); }
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long n) { GRID_STRIDE(i, n) y[i] += a * x[i]; }

inline void pick_rsplit(long rows, int C, eegldm_ctx* ctx, int* rsplit, long* rpb) {
  long want = ((long)ctx->num_cu * 8) / ((C + 63) / 64); if (want < 1) want = 1;
  long maxs = (rows + 63) / 64; if (want > maxs) want = maxs;
  if (eeg_deterministic()) want = 1;      // the row splits add with fp64 atomics: one split = one writer per channel
  *rpb = (rows + want - 1) / want; *rsplit = (int)((rows + *rpb - 1) / *rpb);
}
}  // namespace

#define DISPATCH_T(dtype, ...)                                            \
  do {                                                                    \
    if ((dtype) == EEGLDM_F32) { typedef float T; __VA_ARGS__; }          \
    else if ((dtype) == EEGLDM_BF16) { typedef bf16_t T; __VA_ARGS__; }   \
    else if ((dtype) == EEGLDM_F16) { typedef f16_t T; __VA_ARGS__; }     \
    else EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "dtype %d", (int)(dtype));      \
  } while (0)

// Sum areas in the context scratch: two alternating ones for the partials + fold path (the fold kernel of a call re-zeroes the area
// the PREVIOUS call used -- all of that call's consumers precede it in stream order), a third, memset per call, for the atomic path.
static double* bn_area(eegldm_ctx* ctx, int i) { return (double*)((char*)ctx->scratch + (2u << 20) + (size_t)i * (256u << 10)); }
// parts: [nb][nvals] fp32 partial sums -> sums[nvals] (fp64) in the current sum area; shared with the fused discriminator tail (disc_tail.hip)
int ls_bn_fold(eegldm_ctx* ctx, const void* parts, int nb, int nvals, double** sums_out) {
  EEG_CHECK((size_t)nvals * sizeof(double) <= (256u << 10), "BatchNorm sum area too small for %d values", nvals);
  const int cur = ctx->bn_flip, oth = cur ^ 1;
  double* sums = bn_area(ctx, cur);
  if (eeg_deterministic())
    hipLaunchKernelGGL(bn_fold_det_kernel, dim3((nvals + 15) / 16), dim3(NT), 0, ctx->stream, (const float*)parts, nb, nvals, sums, bn_area(ctx, oth), ctx->bn_dirty[oth]);
  else
    hipLaunchKernelGGL(bn_fold_kernel, dim3((nvals + 63) / 64, 32), dim3(NT), 0, ctx->stream, (const float*)parts, nb, nvals, sums, bn_area(ctx, oth), ctx->bn_dirty[oth]);
  LAUNCH_CHECK();
  ctx->bn_dirty[cur] = nvals; ctx->bn_dirty[oth] = 0; ctx->bn_flip = oth;
  *sums_out = sums;
  return 0;
}
static int bn_fold_launch(eegldm_ctx* ctx, const void* parts, int nb, int C, double** sums_out) { return ls_bn_fold(ctx, parts, nb, 2 * C, sums_out); }
// the statistics half of the forward: stats[C][2] = (mean, rstd) of the batch (training; running statistics updated when rmean != null) or of
// the running statistics (eval)
int ls_bn_stats(eegldm_ctx* ctx, const void* x, long ldx, float* stats, float* rmean, float* rvar, float* nbt, long rows, int C, int training, int dtype) {
  if (training) {
    double* sums = bn_area(ctx, 2);   // BatchNorm region of the context scratch (GroupNorm owns [0, 1 MiB) self-cleaning)
    EEG_CHECK((size_t)C * 2 * sizeof(double) <= (256u << 10), "scratch too small");
    if (C % 4 == 0 && ldx % 4 == 0 && C <= 1024) {
      int nb; long rpb4; bn_split(rows, ctx, 8, &nb, &rpb4);
      void* parts = (char*)ctx->scratch + (8u << 20);
      DISPATCH_T(dtype, hipLaunchKernelGGL((bn_reduce4_kernel<T, 0>), dim3(nb), dim3(NT), 0, ctx->stream, (const T*)x, ldx, nullptr, nullptr, nullptr,
                                           (const T*)nullptr, 0, parts, rows, C, rpb4, 0.f));
      LAUNCH_CHECK();
      if (!eeg_deterministic()) {      // fold + finalise in one launch (the deterministic mode keeps its ordered fold)
        hipLaunchKernelGGL(bn_fold_finalize_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, ctx->stream, (const float*)parts, nb, C, stats, rmean, rvar, nbt,
                           (double)rows, 1e-5f, 0.1f, ctx->bn_running_repeats);
        LAUNCH_CHECK();
        return 0;
      }
      EEG_TRY(bn_fold_launch(ctx, parts, nb, C, &sums));
    } else {
      HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, ctx->stream));
      int rs; long rpb; pick_rsplit(rows, C, ctx, &rs, &rpb);
      DISPATCH_T(dtype, hipLaunchKernelGGL((bn_stats_kernel<T>), dim3((C + 63) / 64, rs), dim3(NT), 0, ctx->stream, (const T*)x, ldx, sums, rows, C, rpb));
    }
    LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, ctx->stream, sums, stats, rmean, rvar, nbt, C, (double)rows, 1e-5f, 0.1f,
                       ctx->bn_running_repeats);
    LAUNCH_CHECK();
  } else {
    EEG_CHECK(rmean && rvar, "eval-mode BatchNorm needs running statistics");
    hipLaunchKernelGGL(bn_eval_stats_kernel, dim3((C + 255) / 256), dim3(256), 0, ctx->stream, rmean, rvar, stats, C, 1e-5f);
    LAUNCH_CHECK();
  }
  return 0;
}
// statistics from the per-block column partials the producing conv left (conv_ws.hip ST kernels): parts[nb][2 C] interleaved (sum, sum of squares)
int ls_bn_stats_from_parts(eegldm_ctx* ctx, const float* parts, int nb, float* stats, float* rmean, float* rvar, float* nbt, long rows, int C) {
  if (!eeg_deterministic()) {
    hipLaunchKernelGGL(bn_fold_finalize_kernel, dim3((2 * C + 63) / 64), dim3(1024), 0, ctx->stream, parts, nb, C, stats, rmean, rvar, nbt, (double)rows, 1e-5f, 0.1f,
                       ctx->bn_running_repeats);
    LAUNCH_CHECK();
    return 0;
  }
  double* sums;
  EEG_TRY(ls_bn_fold(ctx, parts, nb, 2 * C, &sums));
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, ctx->stream, sums, stats, rmean, rvar, nbt, C, (double)rows, 1e-5f, 0.1f,
                     ctx->bn_running_repeats);
  LAUNCH_CHECK();
  return 0;
}
// the apply half: y = lrelu(gamma * (x - mean) * rstd + beta) from given statistics (gamma == null: plain LeakyReLU)
int ls_bn_apply(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta, const float* stats, void* y, long ldy, long rows, int C,
                float slope, int dtype) {
  if (C % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0) {
    int nb; long rpb4; bn_split(rows, ctx, 16, &nb, &rpb4);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply4_kernel<T, 0>), dim3(nb), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats, (const T*)nullptr, 0,
                                         nullptr, (T*)y, ldy, nullptr, nullptr, rows, C, rpb4, slope));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_lrelu_apply_kernel<T>), dim3(grid1d(rows * C, ctx)), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats, (T*)y, ldy, rows, C, slope));
  }
  LAUNCH_CHECK();
  return 0;
}
// BatchNorm1d + LeakyReLU forward.  training: batch statistics (and running-stat update when rmean != null);
// eval: running statistics.  stats: [C][2] fp32 out.  gamma == null: plain LeakyReLU (stats unused).
int ls_bn_lrelu_fwd(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta, float* stats, float* rmean, float* rvar,
                    float* nbt, void* y, long ldy, long rows, int C, float slope, int training, int dtype) {
  if (gamma) EEG_TRY(ls_bn_stats(ctx, x, ldx, stats, rmean, rvar, nbt, rows, C, training, dtype));
  return ls_bn_apply(ctx, x, ldx, gamma, beta, stats, y, ldy, rows, C, slope, dtype);
}
int ls_bn_lrelu_bwd(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta, const float* stats, const void* dy, long lddy,
                    void* dx, long lddx, float* dgamma, float* dbeta, long rows, int C, float slope, int dtype) {
  double* sums = bn_area(ctx, 2);
  if (gamma) {
    if (C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && C <= 1024) {
      int nb; long rpb4; bn_split(rows, ctx, 8, &nb, &rpb4);
      void* parts = (char*)ctx->scratch + (8u << 20);
      DISPATCH_T(dtype, hipLaunchKernelGGL((bn_reduce4_kernel<T, 1>), dim3(nb), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats,
                                           (const T*)dy, lddy, parts, rows, C, rpb4, slope));
      EEG_TRY(bn_fold_launch(ctx, parts, nb, C, &sums));
    } else {
      HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, ctx->stream));
      int rs; long rpb; pick_rsplit(rows, C, ctx, &rs, &rpb);
      DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3((C + 63) / 64, rs), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats,
                                           (const T*)dy, lddy, sums, rows, C, rpb, slope));
    }
    LAUNCH_CHECK();
  }
  if (C % 4 == 0 && ldx % 4 == 0 && lddy % 4 == 0 && lddx % 4 == 0) {
    int nb; long rpb4; bn_split(rows, ctx, 16, &nb, &rpb4);
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_apply4_kernel<T, 1>), dim3(nb), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats, (const T*)dy, lddy,
                                         sums, (T*)dx, lddx, dgamma, dbeta, rows, C, rpb4, slope));
  } else {
    DISPATCH_T(dtype, hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid1d(rows * C, ctx)), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats,
                                         (const T*)dy, lddy, sums, (T*)dx, lddx, dgamma, dbeta, rows, C, slope));
  }
  LAUNCH_CHECK();
  return 0;
}
int ls_upsample2(eegldm_ctx* ctx, const void* x, long ldx, void* y, long ldy, long rows_in, int C, int dtype) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2_kernel<T>), dim3(grid1d(rows_in * C, ctx)), dim3(NT), 0, ctx->stream, (const T*)x, ldx, (T*)y, ldy, rows_in, C));
  LAUNCH_CHECK(); return 0;
}
int ls_upsample2_bwd(eegldm_ctx* ctx, const void* dy, long lddy, void* dx, long lddx, long rows_in, int C, int dtype) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((upsample2_bwd_kernel<T>), dim3(grid1d(rows_in * C, ctx)), dim3(NT), 0, ctx->stream, (const T*)dy, lddy, (T*)dx, lddx, rows_in, C));
  LAUNCH_CHECK(); return 0;
}
int ls_reparam(eegldm_ctx* ctx, const void* mu, const void* lv, const float* eps, void* z, float* sigma, float* kl, long n, int B, int dtype) {
  const int nb = grid1d(n, ctx);
  float* parts = nullptr;      // deterministic mode: a KL partial per block + an ordered fold
  if (kl && eeg_deterministic()) EEG_TRY(eeg_det_buffer(ctx, (size_t)nb * sizeof(float), &parts));
  DISPATCH_T(dtype, hipLaunchKernelGGL((reparam_kernel<T>), dim3(nb), dim3(NT), 0, ctx->stream, (const T*)mu, (const T*)lv, eps, (T*)z, sigma, kl, n, 1.0f / (float)B, parts));
  LAUNCH_CHECK();
  if (parts) EEG_TRY(ew_fold_partials_det(ctx, parts, nb, 1, 0, 1, kl));
  return 0;
}
int ls_reparam_bwd(eegldm_ctx* ctx, const void* mu, const void* lv, const float* eps, const float* sigma, const void* dz, void* dmu, void* dlv, long n,
                   float klw_over_B, int dtype, const float* dmu_ext, const float* dsg_ext, int lat, int Ll) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((reparam_bwd_kernel<T>), dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, (const T*)mu, (const T*)lv, eps, sigma, (const T*)dz,
                                       (T*)dmu, (T*)dlv, n, klw_over_B, dmu_ext, dsg_ext, lat, Ll));
  LAUNCH_CHECK(); return 0;
}

// ================================================================== C ABI (primitives of the AutoencoderKL / PatchDiscriminator path)
// BatchNorm1d (train: batch statistics + running-statistics update; eval: running statistics) + LeakyReLU(slope), NLC rows
extern "C" int eegldm_batchnorm_lrelu_fwd(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta, float* stats,
                                          float* running_mean, float* running_var, float* num_batches_tracked, void* y, long ldy,
                                          long rows, int C, float slope, int training, int dtype) {
  EEG_CHECK(ctx && x && y && rows > 0 && C > 0, "bad argument");
  EEG_CHECK(!gamma || (beta && stats), "BatchNorm needs gamma, beta and the statistics buffer");
  return ls_bn_lrelu_fwd(ctx, x, ldx, gamma, beta, stats, running_mean, running_var, num_batches_tracked, y, ldy, rows, C, slope, training, dtype);
}
extern "C" int eegldm_batchnorm_lrelu_bwd(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta, const float* stats,
                                          const void* dy, long lddy, void* dx, long lddx, float* dgamma, float* dbeta, long rows, int C,
                                          float slope, int dtype) {
  EEG_CHECK(ctx && x && dy && dx && rows > 0 && C > 0, "bad argument");
  EEG_CHECK(!gamma || (beta && stats && dgamma && dbeta), "BatchNorm backward needs gamma, beta, the forward statistics and both gradient buffers");
  return ls_bn_lrelu_bwd(ctx, x, ldx, gamma, beta, stats, dy, lddy, dx, lddx, dgamma, dbeta, rows, C, slope, dtype);
}
// z = mu + eps * sigma, sigma = exp(clamp(log_var, -30, 20) / 2); kl (nullable) += KL(N(mu, sigma) || N(0, 1)) summed over elements / B
extern "C" int eegldm_kl_reparam_fwd(eegldm_ctx* ctx, const void* mu, const void* log_var, const float* eps, void* z, float* sigma, float* kl,
                                     long n, int B, int dtype) {
  EEG_CHECK(ctx && mu && log_var && z && sigma && n > 0 && B > 0, "bad argument");
  return ls_reparam(ctx, mu, log_var, eps, z, sigma, kl, n, B, dtype);
}
extern "C" int eegldm_kl_reparam_bwd(eegldm_ctx* ctx, const void* mu, const void* log_var, const float* eps, const float* sigma, const void* dz,
                                     void* dmu, void* dlog_var, long n, float kl_weight_over_B, int dtype) {
  EEG_CHECK(ctx && mu && log_var && sigma && dmu && dlog_var && n > 0, "bad argument");
  return ls_reparam_bwd(ctx, mu, log_var, eps, sigma, dz, dmu, dlog_var, n, kl_weight_over_B, dtype, nullptr, nullptr, 1, 1);
}

// ================================================================== C ABI (losses)
extern "C" int eegldm_l1_loss(eegldm_ctx* ctx, const float* a, const float* b, float* loss, float* da_accum, long n, float grad_weight) {
  EEG_CHECK(ctx && a && b && loss && n > 0, "bad argument");
  if (!ctx->loss_prezeroed) HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), ctx->stream));
  const int nb = grid1d(n / 4 + 1, ctx);
  float* parts = nullptr;
  if (eeg_deterministic()) EEG_TRY(eeg_det_buffer(ctx, (size_t)nb * sizeof(float), &parts));
  hipLaunchKernelGGL(l1_kernel, dim3(nb), dim3(NT), 0, ctx->stream, a, b, loss, da_accum, n, 1.0f / (float)n, grad_weight, ctx->l1_overwrite ? 1 : 0, parts);
  LAUNCH_CHECK();
  if (parts) EEG_TRY(ew_fold_partials_det(ctx, parts, nb, 1, 0, 1, loss));
  return 0;
}
extern "C" int eegldm_lsgan_loss(eegldm_ctx* ctx, const float* logits, int target_is_real, float* loss, float* dlogits, long n, float grad_weight) {
  EEG_CHECK(ctx && logits && loss && n > 0, "bad argument");
  if (!ctx->loss_prezeroed) HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), ctx->stream));
  const int nb = grid1d(n, ctx);
  float* parts = nullptr;
  if (eeg_deterministic()) EEG_TRY(eeg_det_buffer(ctx, (size_t)nb * sizeof(float), &parts));
  hipLaunchKernelGGL(lsgan_kernel, dim3(nb), dim3(NT), 0, ctx->stream, logits, target_is_real ? 1.0f : 0.0f, loss, dlogits, n, 1.0f / (float)n, grad_weight, parts);
  LAUNCH_CHECK();
  if (parts) EEG_TRY(ew_fold_partials_det(ctx, parts, nb, 1, 0, 1, loss));
  return 0;
}
extern "C" int eegldm_axpy(eegldm_ctx* ctx, float* y, const float* x, float a, long n) {
  EEG_CHECK(ctx && y && x, "null argument");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, y, x, a, n);
  LAUNCH_CHECK(); return 0;
}
