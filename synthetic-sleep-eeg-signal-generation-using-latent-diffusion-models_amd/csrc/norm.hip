// GroupNorm (+SiLU, + fused AvgPool2 / nearest-x2 resampling) forward and backward on the
// NLC layout.  Reference: nn.GroupNorm(32, C, eps=1e-6) at /root/reference/src/models/unet.py:71-74
// used by ResBlock (:260-262, :286-288) and AttentionBlock (:157); MONAI AutoencoderKL uses
// G = norm_num_groups = 1 (config_aekl_eeg.yaml:25).  Up/down ResBlocks resample AFTER the
// activation, on both h and the raw x (unet.py:308-313).
//
// HBM-bound.  Statistics are reduced per (sample, group) with fp32 thread partials (a few
// dozen elements each), LDS float atomics per block, then fp64 global atomics -- biased
// variance from fp64 sums, so E[x^2]-mean^2 cancellation stays below fp32 rounding.
#include "common.h"
#include "internal.h"
#include <stdlib.h>

namespace {

constexpr int NT = 256;
constexpr int MAXG_LDS = 1024;   // LDS accumulators per block (groups / channels)
constexpr int GN_NSLOT = 64;     // partial dgamma/dbeta buffers
constexpr int GN_FOLD_MAX = 96;                          // batched folds per flush (the config_ldm UNet has 42 + 7)
constexpr size_t GN_REGION_FLOATS = (size_t)GN_NSLOT * 2 * 1024;   // one launch's slot region (C <= 1024): 512 KiB
constexpr size_t GN_SLOT_OFFSET = 1u << 20;   // byte offset of the slot area inside ctx->scratch (64 slots x 2C floats <= 512 KiB)
constexpr size_t GN_SLOT_OFFSET_B = (1u << 20) + (512u << 10);   // second slot area: partials whose fold the caller runs later (side stream)
// third / fourth slot areas (alternating): folds the caller runs one ResBlock later, inside that block's side-stream section
constexpr size_t GN_SLOT_OFFSET_C[2] = {24u << 20, (24u << 20) + (512u << 10)};
static inline size_t gn_slot_region(int region) { return region == 0 ? GN_SLOT_OFFSET_B : GN_SLOT_OFFSET_C[(region - 1) & 1]; }

template <typename T, int V> struct Vec;
template <> struct Vec<float, 4> { typedef float4 type; };
template <> struct Vec<bf16_t, 4> { typedef uint2 type; };
template <> struct Vec<f16_t, 4> { typedef uint2 type; };

template <typename T> __device__ __forceinline__ void load4(const T* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
  float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
  uint2 t = *(const uint2*)p;
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void load4<f16_t>(const f16_t* p, float v[4]) {
  uint2 t = *(const uint2*)p;
  v[0] = w16_lo<f16_t>(t.x); v[1] = w16_hi<f16_t>(t.x); v[2] = w16_lo<f16_t>(t.y); v[3] = w16_hi<f16_t>(t.y);
}
template <typename T> __device__ __forceinline__ void unpack4(const typename Vec<T, 4>::type& t, float v[4]);
template <> __device__ __forceinline__ void unpack4<f16_t>(const uint2& t, float v[4]) {
  v[0] = w16_lo<f16_t>(t.x); v[1] = w16_hi<f16_t>(t.x); v[2] = w16_lo<f16_t>(t.y); v[3] = w16_hi<f16_t>(t.y);
}
template <> __device__ __forceinline__ void unpack4<float>(const float4& t, float v[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
template <> __device__ __forceinline__ void unpack4<bf16_t>(const uint2& t, float v[4]) {
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
  *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
  uint2 t;
  t.x = pack_bf16x2(v[0], v[1]);
  t.y = pack_bf16x2(v[2], v[3]);
  *(uint2*)p = t;
}
template <> __device__ __forceinline__ void store4<f16_t>(f16_t* p, const float v[4]) {
  uint2 t; t.x = pack_f16x2(v[0], v[1]); t.y = pack_f16x2(v[2], v[3]); *(uint2*)p = t;
}
template <typename T, int V> __device__ __forceinline__ void loadv(const T* p, float v[V]) {
  if constexpr (V == 4) load4<T>(p, v); else v[0] = ld_f32(p);
}
template <typename T, int V> __device__ __forceinline__ void storev(T* p, const float v[V]) {
  if constexpr (V == 4) store4<T>(p, v); else st_f32(p, v[0]);
}

// thread -> (column-vector, row-lane) decomposition of a block
struct ColMap { int TX, TY, ncols; };
__device__ __forceinline__ ColMap colmap(int C, int V) {
  ColMap m; m.ncols = C / V;
  if (m.ncols >= NT) { m.TX = NT; m.TY = 1; } else { m.TX = m.ncols; m.TY = NT / m.ncols; }
  return m;
}

// ------------------------------------------------------------------ forward statistics
// grid (LSPLIT, B); sums: double [B][G][2] (pre-zeroed)
template <typename T, int V>
__global__ __launch_bounds__(NT) void gn_stats_kernel(const T* __restrict__ x, long ldx, double* __restrict__ sums,
                                                      int L, int C, int G, int rows_per_block) {
  __shared__ double acc[2 * MAXG_LDS];   // fp64 like the global sums: the arrival order of the threads must not show in the statistics (reproducible forward)
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G;
  for (int i = tid; i < 2 * G; i += NT) acc[i] = 0.0;
  __syncthreads();
  const ColMap cm = colmap(C, V);
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  if (tid < cm.TX * cm.TY) {
    const int tx = tid % cm.TX, ty = tid / cm.TX;
    for (int col = tx; col < cm.ncols; col += cm.TX) {
      const int c = col * V;
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int l = l0 + ty; l < l1; l += cm.TY) {
        float v[V];
        loadv<T, V>(x + ((long)b * L + l) * ldx + c, v);
#pragma unroll
        for (int k = 0; k < V; k++) { s1 += v[k]; s2 += v[k] * v[k]; }
      }
      const int g = c / cpg;
      atomicAdd(&acc[2 * g], (double)s1);
      atomicAdd(&acc[2 * g + 1], (double)s2);
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * G; i += NT) atomicAdd(&sums[(long)b * G * 2 + i], acc[i]);
}

// (self-cleaning: the accumulators are zeroed again for the next GroupNorm, so no memset launches are needed)
__global__ void gn_finalize_kernel(double* __restrict__ sums, float* __restrict__ stats, int BG, double n, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= BG) return;
  const double mean = sums[2 * i] / n;
  double var = sums[2 * i + 1] / n - mean * mean;
  sums[2 * i] = 0.0; sums[2 * i + 1] = 0.0;
  if (var < 0) var = 0;
  stats[2 * i] = (float)mean;
  stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// ------------------------------------------------------------------ forward apply
// grid (row chunks, B): block = one sample, a run of WORK rows (work row = output row for avgpool, input row
// otherwise).  Threads tile (column vector, row lane) like the statistics kernel, so the element loop has no
// integer division; per-thread scale/shift are hoisted out of the row loop.
// resample: 0 none, 1 avgpool2 (pairs of input rows -> one output row), 2 nearest x2 (one input row -> two output rows)
template <typename T, int V>
__global__ __launch_bounds__(NT) void gn_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ stats,
                                                      T* __restrict__ y, long ldy, T* __restrict__ xr, long ldxr,
                                                      int L, int C, int G, int silu, int resample, int rows_per_block) {
  const int b = blockIdx.y, tid = threadIdx.x, cpg = C / G;
  const ColMap cm = colmap(C, V);
  if (tid >= cm.TX * cm.TY) return;
  const int tx = tid % cm.TX, ty = tid / cm.TX;
  const int Lw = (resample == 1) ? L / 2 : L;
  const int l0 = blockIdx.x * rows_per_block, l1 = min(Lw, l0 + rows_per_block);
  const T* xb = x + (long)b * L * ldx;
  for (int col = tx; col < cm.ncols; col += cm.TX) {
    const int c = col * V, g = c / cpg;
    const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    float ga[V], be[V];
#pragma unroll
    for (int k = 0; k < V; k++) { ga[k] = gamma[c + k] * rstd; be[k] = beta[c + k] - mean * ga[k]; }
    if (resample == 1) {
      T* yb = y + (long)b * Lw * ldy; T* xrb = xr ? xr + (long)b * Lw * ldxr : nullptr;
      for (int lw = l0 + ty; lw < l1; lw += cm.TY) {
        float v0[V], v1[V], o[V], r[V];
        loadv<T, V>(xb + (long)(2 * lw) * ldx + c, v0);
        loadv<T, V>(xb + (long)(2 * lw + 1) * ldx + c, v1);
#pragma unroll
        for (int k = 0; k < V; k++) {
          float z0 = v0[k] * ga[k] + be[k], z1 = v1[k] * ga[k] + be[k];
          if (silu) { z0 = silu_f(z0); z1 = silu_f(z1); }
          o[k] = 0.5f * (z0 + z1); r[k] = 0.5f * (v0[k] + v1[k]);
        }
        storev<T, V>(yb + (long)lw * ldy + c, o);
        if (xrb) storev<T, V>(xrb + (long)lw * ldxr + c, r);
      }
    } else if (resample == 0) {
      T* yb = y + (long)b * L * ldy;
#pragma unroll 4
      for (int lw = l0 + ty; lw < l1; lw += cm.TY) {
        float v[V], o[V];
        loadv<T, V>(xb + (long)lw * ldx + c, v);
#pragma unroll
        for (int k = 0; k < V; k++) { float z = v[k] * ga[k] + be[k]; o[k] = silu ? silu_f(z) : z; }
        storev<T, V>(yb + (long)lw * ldy + c, o);
      }
    } else {
      T* yb = y + (long)b * 2 * L * ldy; T* xrb = xr ? xr + (long)b * 2 * L * ldxr : nullptr;
      for (int lw = l0 + ty; lw < l1; lw += cm.TY) {
        float v[V], o[V];
        loadv<T, V>(xb + (long)lw * ldx + c, v);
#pragma unroll
        for (int k = 0; k < V; k++) { float z = v[k] * ga[k] + be[k]; o[k] = silu ? silu_f(z) : z; }
        storev<T, V>(yb + (long)(2 * lw) * ldy + c, o); storev<T, V>(yb + (long)(2 * lw + 1) * ldy + c, o);
        if (xrb) { storev<T, V>(xrb + (long)(2 * lw) * ldxr + c, v); storev<T, V>(xrb + (long)(2 * lw + 1) * ldxr + c, v); }
      }
    }
  }
}

// ------------------------------------------------------------------ backward
// effective upstream gradient at input row (b,l), channels c..c+V-1
// (resample mode as a compile-time constant inside streaming loops: a run-time branch around the load makes hipcc wait
//  for every load separately instead of keeping an unrolled batch in flight)
template <typename T, int V, int RS>
__device__ __forceinline__ void load_dy_eff_c(const T* dy, long lddy, int b, int l, int L, int c, float d[V]) {
  if constexpr (RS == 0) {
    loadv<T, V>(dy + ((long)b * L + l) * lddy + c, d);
  } else if constexpr (RS == 1) {
    loadv<T, V>(dy + ((long)b * (L / 2) + (l >> 1)) * lddy + c, d);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] *= 0.5f;
  } else {
    float e[V];
    loadv<T, V>(dy + ((long)b * 2 * L + 2 * l) * lddy + c, d);
    loadv<T, V>(dy + ((long)b * 2 * L + 2 * l + 1) * lddy + c, e);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] += e[k];
  }
}
template <typename T, int V>
__device__ __forceinline__ void load_dy_eff(const T* dy, long lddy, int b, int l, int L, int c, int resample, float d[V]) {
  if (resample == 0) {
    loadv<T, V>(dy + ((long)b * L + l) * lddy + c, d);
  } else if (resample == 1) {
    loadv<T, V>(dy + ((long)b * (L / 2) + (l >> 1)) * lddy + c, d);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] *= 0.5f;
  } else {
    float e[V];
    loadv<T, V>(dy + ((long)b * 2 * L + 2 * l) * lddy + c, d);
    loadv<T, V>(dy + ((long)b * 2 * L + 2 * l + 1) * lddy + c, e);
#pragma unroll
    for (int k = 0; k < V; k++) d[k] += e[k];
  }
}

// grid (LSPLIT, B): group sums S1 = sum dz*gamma, S2 = sum dz*gamma*xhat -> gsums double [B][G][2];
// dgamma[c] += sum dz*xhat, dbeta[c] += sum dz (fp32 atomics)
template <typename T, int V, int RS>
__global__ __launch_bounds__(NT) void gn_bwd_reduce_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ stats,
                                                           const T* __restrict__ dy, long lddy, double* __restrict__ gsums,
                                                           float* __restrict__ slots,
                                                           int L, int C, int G, int silu, int rows_per_block, int nslot) {
  __shared__ double accg[2 * MAXG_LDS];  // group sums feed dx: fp64 so that the thread arrival order does not show (see gn_stats_kernel)
  __shared__ double accc[2 * MAXG_LDS];
  const int b = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / G;
  for (int i = tid; i < 2 * G; i += NT) accg[i] = 0.0;
  for (int i = tid; i < 2 * C; i += NT) accc[i] = 0.0;
  __syncthreads();
  const ColMap cm = colmap(C, V);
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  if (tid < cm.TX * cm.TY) {
    const int tx = tid % cm.TX, ty = tid / cm.TX;
    for (int col = tx; col < cm.ncols; col += cm.TX) {
      const int c = col * V, g = c / cpg;
      const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
      float ga[V], be[V], dg[V], db[V];
#pragma unroll
      for (int k = 0; k < V; k++) { ga[k] = gamma[c + k]; be[k] = beta[c + k]; dg[k] = 0.f; db[k] = 0.f; }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
      for (int l = l0 + ty; l < l1; l += cm.TY) {
        float v[V], d[V];
        loadv<T, V>(x + ((long)b * L + l) * ldx + c, v);
        load_dy_eff_c<T, V, RS>(dy, lddy, b, l, L, c, d);
#pragma unroll
        for (int k = 0; k < V; k++) {
          const float xh = (v[k] - mean) * rstd;
          float dz = d[k];
          if (silu) dz *= silu_grad_f(ga[k] * xh + be[k]);
          dg[k] += dz * xh; db[k] += dz;
          s1 += dz * ga[k]; s2 += dz * ga[k] * xh;
        }
      }
      atomicAdd(&accg[2 * g], (double)s1); atomicAdd(&accg[2 * g + 1], (double)s2);
#pragma unroll
      for (int k = 0; k < V; k++) { atomicAdd(&accc[c + k], (double)dg[k]); atomicAdd(&accc[C + c + k], (double)db[k]); }
    }
  }
  __syncthreads();
  for (int i = tid; i < 2 * G; i += NT) atomicAdd(&gsums[(long)b * G * 2 + i], accg[i]);
  // per-channel sums go to one of NSLOT partial buffers (thousands of blocks hammering 2C addresses serialise in L2)
  if (slots) {
    float* sl = slots + (size_t)((blockIdx.y * gridDim.x + blockIdx.x) % nslot) * 2 * C;
    for (int i = tid; i < 2 * C; i += NT) atomicAdd(&sl[i], (float)accc[i]);
  }
}

// runs AFTER the backward apply kernel: folds the slots into dgamma/dbeta and re-zeroes slots and group sums
__global__ void gn_slot_reduce_kernel(float* __restrict__ slots, float* __restrict__ dgamma, float* __restrict__ dbeta, int C,
                                      double* __restrict__ gsums, int n_gsums, int nslot = GN_NSLOT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = i; k < n_gsums; k += gridDim.x * blockDim.x) gsums[k] = 0.0;
  if (i >= 2 * C || !slots) return;
  float s = 0.f;
  for (int k = 0; k < nslot; k++) { s += slots[(size_t)k * 2 * C + i]; slots[(size_t)k * 2 * C + i] = 0.f; }
  if (i < C) dgamma[i] += s; else dbeta[i - C] += s;
}

// dx = rstd * (dz*gamma - S1/n - xhat*S2/n) [+ resample^T(dxr)];  grid (row chunks, B), same tiling as the forward
template <typename T, int V, int RS>
__global__ __launch_bounds__(NT) void gn_bwd_apply_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ stats,
                                                          const T* __restrict__ dy, long lddy, const double* __restrict__ gsums,
                                                          T* __restrict__ dx, long lddx, const T* __restrict__ dxr, long lddxr,
                                                          int L, int C, int G, int silu, int rows_per_block) {
  const int b = blockIdx.y, tid = threadIdx.x, cpg = C / G;
  const ColMap cm = colmap(C, V);
  if (tid >= cm.TX * cm.TY) return;
  const int tx = tid % cm.TX, ty = tid / cm.TX;
  const float inv_n = 1.0f / ((float)cpg * (float)L);
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  for (int col = tx; col < cm.ncols; col += cm.TX) {
    const int c = col * V, g = c / cpg;
    const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
    const float m1 = (float)gsums[((long)b * G + g) * 2] * inv_n, m2 = (float)gsums[((long)b * G + g) * 2 + 1] * inv_n;
    float ga[V], be[V];
#pragma unroll
    for (int k = 0; k < V; k++) { ga[k] = gamma[c + k]; be[k] = beta[c + k]; }
#pragma unroll 2
    for (int l = l0 + ty; l < l1; l += cm.TY) {
      const long row = (long)b * L + l;
      float v[V], d[V], o[V];
      loadv<T, V>(x + row * ldx + c, v);
      load_dy_eff_c<T, V, RS>(dy, lddy, b, l, L, c, d);
#pragma unroll
      for (int k = 0; k < V; k++) {
        const float xh = (v[k] - mean) * rstd;
        float dz = d[k];
        if (silu) dz *= silu_grad_f(ga[k] * xh + be[k]);
        o[k] = rstd * (dz * ga[k] - m1 - xh * m2);
      }
      if (dxr) {
        float e[V];
        load_dy_eff_c<T, V, RS>(dxr, lddxr, b, l, L, c, e);
#pragma unroll
        for (int k = 0; k < V; k++) o[k] += e[k];
      }
      storev<T, V>(dx + row * lddx + c, o);
    }
  }
}

// effective upstream gradient from a wave-uniform per-sample base and 32-bit byte offsets (the one-pass kernels: per-lane
// 64-bit pointer arithmetic was ~30 % of their VALU instructions, and they are VALU-bound)
template <typename T>
__device__ __forceinline__ void load_dy_eff32(const char* base, unsigned ldb, unsigned cb, int l, int resample, float d[4]) {
  if (resample == 0) {
    load4<T>((const T*)(base + ((unsigned)l * ldb + cb)), d);
  } else if (resample == 1) {
    load4<T>((const T*)(base + ((unsigned)(l >> 1) * ldb + cb)), d);
#pragma unroll
    for (int k = 0; k < 4; k++) d[k] *= 0.5f;
  } else {
    float e[4];
    load4<T>((const T*)(base + ((unsigned)(2 * l) * ldb + cb)), d);
    load4<T>((const T*)(base + ((unsigned)(2 * l + 1) * ldb + cb)), e);
#pragma unroll
    for (int k = 0; k < 4; k++) d[k] += e[k];
  }
}
__device__ __forceinline__ long dy_rows(int L, int resample) { return resample == 1 ? L / 2 : (resample == 2 ? 2L * L : L); }

// ------------------------------------------------------------------ register-resident one-pass GroupNorm
// A 1024-thread block owns ONE sample x a chunk of CC channels (whole groups) over the full length L and keeps its
// rows in registers: x (and dy) are read from HBM exactly once, statistics / group sums are reduced inside the
// block (no global atomics, no finalize launch), and the normalised output / input gradient is produced from the
// registers.  HBM traffic: forward 1R+1W (was 2R+1W), backward 2R+1W (was 4R+1W).
// Thread (tx, ty): tx = 4-channel vector column of the chunk, ty = row lane; row k of a thread is k*TY + ty, or in
// pair mode (avgpool) rows 2*((k/2)*TY + ty) + k%2 so that both rows of a pooling pair live in one thread.
constexpr int NTB = 1024;
constexpr int RES_MAXG = 64;     // local groups per chunk
constexpr int RES_MAXC = 256;    // channels per chunk

struct ResMap { int TX, TY, tx, ty, c, gl, b; bool act; };
// The grid is 1-D: linear id -> (channel chunk, sample).  Plain order = chunk fastest.  XCD-aware order (`xcd` != 0, B % 8 == 0): the
// hardware deals consecutive workgroup ids round-robin to the 8 XCDs, so ids = x (mod 8) run on XCD x in increasing order; the j-th of
// them is chunk j % nchunk of sample (j / nchunk) * 8 + x -- all chunks of one sample run back to back on ONE XCD, so a 128-byte line
// that several narrow chunks share is fetched from HBM once and served to its other readers by that XCD's L2.
template <int NTH>
__device__ __forceinline__ ResMap resmap(int CC, int C, int cpg, int xcd) {
  ResMap m; m.TX = CC / 4; m.TY = NTH / m.TX;
  const int tid = threadIdx.x;
  const int nchunk = C / CC, lin = blockIdx.x;
  int chunk, b;
  if (xcd) { const int x = lin & 7, j = lin >> 3; b = (j / nchunk) * 8 + x; chunk = j % nchunk; }
  else { chunk = lin % nchunk; b = lin / nchunk; }
  m.b = b;
  m.tx = tid % m.TX; m.ty = tid / m.TX;
  m.c = chunk * CC + m.tx * 4; m.gl = (m.tx * 4) / cpg;
  m.act = tid < m.TX * m.TY && m.c < C;
  return m;
}
__device__ __forceinline__ int res_row(int k, int ty, int TY, bool pair) {
  return pair ? 2 * ((k >> 1) * TY + ty) + (k & 1) : k * TY + ty;
}

#ifdef EEG_STAGE_TIMING
__device__ unsigned long long gn_tlog[4096 * 8];
#define GN_TSTAMP(k) do { if (threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); gn_tlog[((size_t)blockIdx.y * gridDim.x + blockIdx.x) % 4096 * 8 + (k)] = __builtin_readcyclecounter(); } } while (0)
#else
#define GN_TSTAMP(k) do {} while (0)
#endif
template <typename T, int RPT, int NTH>
__global__ __launch_bounds__(NTH) void gn_fwd_resident_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, T* __restrict__ y, long ldy,
                                                              T* __restrict__ xr, long ldxr, float* __restrict__ stats,
                                                              int L, int C, int G, float eps, int silu, int resample, int CC, int xcd) {
  // fp64 LDS accumulators: the order in which the waves arrive no longer shows in the fp32 mean / rstd, so the forward (and with it
  // seeded sampling) is reproducible run to run -- fp32 atomics differed in the last bit and bf16 roundings downstream flipped
  __shared__ double red[2 * RES_MAXG];
  GN_TSTAMP(0);
  const int cpg = C / G;
  const ResMap m = resmap<NTH>(CC, C, cpg, xcd);
  const int b = m.b;
  const bool pair = resample == 1;
  for (int i = threadIdx.x; i < 2 * RES_MAXG; i += NTH) red[i] = 0.0;
  typename Vec<T, 4>::type raw[RPT];
  const T* xb = x + (long)b * L * ldx + m.c;
  if (m.act) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int l = res_row(k, m.ty, m.TY, pair);
      if (l < L) raw[k] = *(const typename Vec<T, 4>::type*)(xb + (long)l * ldx);
    }
  }
  GN_TSTAMP(1);
  __syncthreads();
  GN_TSTAMP(2);
  const float inv_n = 1.0f / ((float)cpg * (float)L);
  float mean = 0.f, rstd = 0.f;
  if constexpr (sizeof(T) == 2) {
    // 16-bit storage: ONE reduction round.  Sum and sum of squares are taken about a per-thread-independent shift of zero in fp32
    // per thread (<= 48 elements), accumulated in fp64 across the block, and the variance is E[x^2] - mean^2 in fp64: the relative
    // error of the fp32 partials (~1e-7) is amplified by 1 + mean^2 / var, harmless for activations whose mean is within tens of
    // standard deviations, and far below the bf16 rounding of the output.  Saves the second barrier round of the two-pass form
    // (mean first, then centred squares: 5.3 k of the block's 28 k cycles, tools/debug/gn_timing.py).  The fp32 engine keeps two passes.
    float s1 = 0.f, s2 = 0.f;
    if (m.act) {
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        if (res_row(k, m.ty, m.TY, pair) < L) {
          float v[4]; unpack4<T>(raw[k], v);
          s1 += (v[0] + v[1]) + (v[2] + v[3]);
          s2 = fmaf(v[0], v[0], s2); s2 = fmaf(v[1], v[1], s2); s2 = fmaf(v[2], v[2], s2); s2 = fmaf(v[3], v[3], s2);
        }
      }
      atomicAdd(&red[m.gl], (double)s1); atomicAdd(&red[RES_MAXG + m.gl], (double)s2);
    }
    __syncthreads();
    GN_TSTAMP(3);
    GN_TSTAMP(4);
    if (!m.act) return;
    const double mu = red[m.gl] * (double)inv_n;
    double var = red[RES_MAXG + m.gl] * (double)inv_n - mu * mu; if (var < 0.0) var = 0.0;
    mean = (float)mu;
    rstd = rsqrtf((float)var + eps);
  } else {
  float s = 0.f;
  if (m.act) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      if (res_row(k, m.ty, m.TY, pair) < L) { float v[4]; unpack4<T>(raw[k], v); s += (v[0] + v[1]) + (v[2] + v[3]); }
    }
    atomicAdd(&red[m.gl], (double)s);
  }
  __syncthreads();
  GN_TSTAMP(3);
  if (m.act) {
    mean = (float)(red[m.gl] * (double)inv_n);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      if (res_row(k, m.ty, m.TY, pair) < L) {
        float v[4]; unpack4<T>(raw[k], v);
#pragma unroll
        for (int j = 0; j < 4; j++) { const float d = v[j] - mean; q += d * d; }
      }
    }
    atomicAdd(&red[RES_MAXG + m.gl], (double)q);
  }
  __syncthreads();
  GN_TSTAMP(4);
  if (!m.act) return;
  rstd = rsqrtf((float)(red[RES_MAXG + m.gl] * (double)inv_n) + eps);
  }
  if (m.ty == 0 && (m.tx * 4) % cpg == 0) { float* st = stats + ((long)b * G + m.c / cpg) * 2; st[0] = mean; st[1] = rstd; }
  float ga[4], be[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { ga[j] = gamma[m.c + j] * rstd; be[j] = beta[m.c + j] - mean * ga[j]; }
  if (resample == 1) {
    T* yb = y + (long)b * (L / 2) * ldy + m.c; T* xrb = xr ? xr + (long)b * (L / 2) * ldxr + m.c : nullptr;
#pragma unroll
    for (int k = 0; k + 1 < RPT; k += 2) {
      const int l = res_row(k, m.ty, m.TY, true);
      if (l + 1 < L) {
        float v0[4], v1[4], o[4], r[4];
        unpack4<T>(raw[k], v0); unpack4<T>(raw[k + 1], v1);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          float z0 = v0[j] * ga[j] + be[j], z1 = v1[j] * ga[j] + be[j];
          if (silu) { z0 = silu_f(z0); z1 = silu_f(z1); }
          o[j] = 0.5f * (z0 + z1); r[j] = 0.5f * (v0[j] + v1[j]);
        }
        store4<T>(yb + (long)(l >> 1) * ldy, o);
        if (xrb) store4<T>(xrb + (long)(l >> 1) * ldxr, r);
      }
    }
  } else {
    const int up = resample == 2 ? 2 : 1;
    T* yb = y + (long)b * up * L * ldy + m.c; T* xrb = (xr && up == 2) ? xr + (long)b * 2 * L * ldxr + m.c : nullptr;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int l = res_row(k, m.ty, m.TY, false);
      if (l < L) {
        float v[4], o[4];
        unpack4<T>(raw[k], v);
#pragma unroll
        for (int j = 0; j < 4; j++) { const float z = v[j] * ga[j] + be[j]; o[j] = silu ? silu_f(z) : z; }
        if (up == 1) store4<T>(yb + (long)l * ldy, o);
        else {
          store4<T>(yb + (long)(2 * l) * ldy, o); store4<T>(yb + (long)(2 * l + 1) * ldy, o);
          if (xrb) { store4<T>(xrb + (long)(2 * l) * ldxr, v); store4<T>(xrb + (long)(2 * l + 1) * ldxr, v); }
        }
      }
    }
  }
  GN_TSTAMP(5);
}

// backward: dgamma/dbeta partials go to the slot buffers (folded by gn_slot_reduce_kernel); colsum_ps (optional) receives
// the per-sample column sums of the written dx -- the block owns (sample, channels) over all of L, so no atomics.
// RAW0: resample == 0 specialisation that fetches the gradient rows packed, in the same predicated block as the x rows (the
// conversion in place makes hipcc wait for every load separately: 12 serialised HBM latencies, 19 k of the block's 57 k cycles)
#ifdef EEG_GN_W4      // developer build (tools/debug/gn_hazard.sh): at most 128 VGPRs, whatever else the build flags do
#define GN_BWD_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define GN_BWD_ATTR
#endif
template <typename T, int RPT, bool RAW0, bool SILU, int NTH>
__global__ __launch_bounds__(NTH) GN_BWD_ATTR void gn_bwd_resident_kernel(const T* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ stats,
                                                              const T* __restrict__ dy, long lddy, T* __restrict__ dx, long lddx,
                                                              const T* __restrict__ dxr, long lddxr, float* __restrict__ slots,
                                                              float* __restrict__ colsum_ps, long ldps,
                                                              int L, int C, int G, int silu, int resample, int CC,
                                                              const T* __restrict__ dxr2, long lddxr2, int xcd, int nslot) {
  // fp64 LDS accumulators (as in the forward): dx must not depend on the order in which the waves arrive -- an fp32 one-ulp
  // difference in the group sums flips bf16 roundings of dx and the flips compound through the remaining layers
  __shared__ double redg[2 * RES_MAXG];
  __shared__ double redc[3 * RES_MAXC];
#ifdef EEG_GN_BWD_NO_LDS_ATOMICS
  __shared__ float gn_part[NTH * 8];
#endif
  GN_TSTAMP(0);
  const int cpg = C / G;
  const ResMap m = resmap<NTH>(CC, C, cpg, xcd);
  const int b = m.b;
  for (int i = threadIdx.x; i < 2 * RES_MAXG + 3 * RES_MAXC; i += NTH) { if (i < 2 * RES_MAXG) redg[i] = 0.0; else redc[i - 2 * RES_MAXG] = 0.0; }
  typename Vec<T, 4>::type raw[RPT];
  float d[RPT][4];
  // wave-uniform per-sample bases + 32-bit byte offsets (scalar-base addressing: one VALU add per access)
  const char* xs = (const char*)(x + (long)b * L * ldx);
  const char* dys = (const char*)(dy + (long)b * dy_rows(L, resample) * lddy);
  const unsigned cb = (unsigned)m.c * (unsigned)sizeof(T);
  const unsigned ldxb = (unsigned)ldx * (unsigned)sizeof(T), lddyb = (unsigned)lddy * (unsigned)sizeof(T);
  typename Vec<T, 4>::type dr[RAW0 ? RPT : 1];
  if (m.act) {
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int l = k * m.TY + m.ty;
      if (l < L) {
        raw[k] = *(const typename Vec<T, 4>::type*)(xs + ((unsigned)l * ldxb + cb));
        if constexpr (RAW0) dr[k] = *(const typename Vec<T, 4>::type*)(dys + ((unsigned)l * lddyb + cb));
      }
    }
    if constexpr (!RAW0) {
#pragma unroll
      for (int k = 0; k < RPT; k++) {
        const int l = k * m.TY + m.ty;
        if (l < L) load_dy_eff32<T>(dys, lddyb, cb, l, resample, d[k]);
      }
    }
  }
  GN_TSTAMP(1);
  __syncthreads();
  GN_TSTAMP(2);
  float mean = 0.f, rstd = 0.f, ga[4], be[4];
  if (m.act) {
    const float* st = stats + ((long)b * G + m.c / cpg) * 2;
    mean = st[0]; rstd = st[1];
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; j++) { ga[j] = gamma[m.c + j]; be[j] = beta[m.c + j]; }
    // pass 1 is VALU issue-bound (profiles/r01_gemm_stage_timing.txt): xhat as one FMA, SiLU' evaluated unconditionally and
    // selected (no per-element branch), and the group sums S1 = sum dz*gamma, S2 = sum dz*gamma*xhat are NOT accumulated per
    // element -- they follow from the per-channel sums after the block reduction (S1 = sum_c gamma_c db_c, S2 = sum_c gamma_c dg_c)
    const float nmr = -mean * rstd;
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      if (k * m.TY + m.ty < L) {
        float v[4]; unpack4<T>(raw[k], v);
        if constexpr (RAW0) unpack4<T>(dr[k], d[k]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float xh = fmaf(v[j], rstd, nmr);
          float dz = d[k][j];
          if constexpr (SILU) dz *= silu_grad_f(fmaf(ga[j], xh, be[j]));   // (compile-time: a run-time select cost one v_cndmask per element)
          d[k][j] = dz;
          dg[j] = fmaf(dz, xh, dg[j]); db[j] += dz;
        }
      }
    }
#ifdef EEG_GN_BWD_NO_LDS_ATOMICS
    // developer build (tools/debug/gn_hazard.sh): no LDS atomics anywhere in this kernel -- per-thread partials to LDS, one writer per sum
#pragma unroll
    for (int j = 0; j < 4; j++) { gn_part[(m.ty * m.TX + m.tx) * 8 + j] = dg[j]; gn_part[(m.ty * m.TX + m.tx) * 8 + 4 + j] = db[j]; }
#ifdef EEG_GN_HAZ_VERIFY      // (1) does a thread read back what it just wrote?
    {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bool mism = false;
#pragma unroll
      for (int j = 0; j < 4; j++) mism = mism || ((volatile float*)gn_part)[(m.ty * m.TX + m.tx) * 8 + j] != dg[j] || ((volatile float*)gn_part)[(m.ty * m.TX + m.tx) * 8 + 4 + j] != db[j];
      if (mism) printf("HAZ1 own write not read back: block %d thread %d\n", (int)blockIdx.x, (int)threadIdx.x);
    }
#endif
#else
#pragma unroll
    for (int j = 0; j < 4; j++) { atomicAdd(&redc[m.tx * 4 + j], (double)dg[j]); atomicAdd(&redc[RES_MAXC + m.tx * 4 + j], (double)db[j]); }
#endif
  }
#ifdef EEG_GN_BWD_NO_LDS_ATOMICS
  __syncthreads();
  if (m.act && m.ty == 0) {
    double sg[4] = {0.0, 0.0, 0.0, 0.0}, sb[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = 0; r < m.TY; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) { sg[j] += (double)gn_part[(r * m.TX + m.tx) * 8 + j]; sb[j] += (double)gn_part[(r * m.TX + m.tx) * 8 + 4 + j]; }
#pragma unroll
    for (int j = 0; j < 4; j++) { redc[m.tx * 4 + j] = sg[j]; redc[RES_MAXC + m.tx * 4 + j] = sb[j]; }
#ifdef EEG_GN_HAZ_VERIFY      // (2) do the partials still read the same a little later?  (3) were they all written: own row 0 equals the registers
    __builtin_amdgcn_s_sleep(64);
    double sg2[4] = {0.0, 0.0, 0.0, 0.0}, sb2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int r = 0; r < m.TY; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) { sg2[j] += (double)((volatile float*)gn_part)[(r * m.TX + m.tx) * 8 + j]; sb2[j] += (double)((volatile float*)gn_part)[(r * m.TX + m.tx) * 8 + 4 + j]; }
    bool diff = false;
#pragma unroll
    for (int j = 0; j < 4; j++) diff = diff || sg2[j] != sg[j] || sb2[j] != sb[j];
    if (diff) printf("HAZ2 partials changed after the barrier: block %d column %d (TY %d): %.9g -> %.9g\n", (int)blockIdx.x, m.tx, m.TY, sg[0], sg2[0]);
#endif
  }
#endif
  // the residual-path addend(s) of dx are fetched HERE, packed and unconditionally (clamped row), so their round trip runs under the
  // two barriers and the group-sum phase: loaded inside pass 2 they were twelve load-wait-use chains per thread (the first GroupNorm of
  // every ResBlock has such an addend: those launches took ~70 us against 45 us without)
  typename Vec<T, 4>::type er[RAW0 ? RPT : 1];      // (the rarer second addend stays an in-loop load: both arrays spill at 12 rows per thread)
  if constexpr (RAW0) {
    if (m.act && dxr) {
      const char* q = (const char*)(dxr + (long)b * L * lddxr);
      const unsigned ldb = (unsigned)lddxr * (unsigned)sizeof(T);
#pragma unroll
      for (int k = 0; k < RPT; k++) { const int l = k * m.TY + m.ty, lc = l < L ? l : L - 1; er[k] = *(const typename Vec<T, 4>::type*)(q + ((unsigned)lc * ldb + cb)); }
    }
  }
  __syncthreads();
#ifdef EEG_GN_BWD_NO_LDS_ATOMICS
  if (m.act && m.ty == 0 && (m.tx * 4) % cpg == 0) {       // one writer per group: its channels' sums in channel order
    double p1 = 0.0, p2 = 0.0;
    for (int i = 0; i < cpg; i++) { const double g = (double)gamma[m.c + i]; p1 += g * redc[RES_MAXC + m.tx * 4 + i]; p2 += g * redc[m.tx * 4 + i]; }
    redg[2 * m.gl] = p1; redg[2 * m.gl + 1] = p2;
  }
#else
  if (m.act && m.ty == 0) {
    double p1 = 0.0, p2 = 0.0;
#pragma unroll
    for (int j = 0; j < 4; j++) { p1 += (double)ga[j] * redc[RES_MAXC + m.tx * 4 + j]; p2 += (double)ga[j] * redc[m.tx * 4 + j]; }
    atomicAdd(&redg[2 * m.gl], p1); atomicAdd(&redg[2 * m.gl + 1], p2);
  }
#endif
  GN_TSTAMP(3);
  __syncthreads();
  GN_TSTAMP(4);
  if (m.act) {
    const float inv_n = 1.0f / ((float)cpg * (float)L);
    const float m1 = (float)(redg[2 * m.gl] * (double)inv_n), m2 = (float)(redg[2 * m.gl + 1] * (double)inv_n);
    if (slots && m.ty == 0) {
      float* sl = slots + (size_t)(b % nslot) * 2 * C;
#pragma unroll
      for (int j = 0; j < 4; j++) { atomicAdd(&sl[m.c + j], (float)redc[m.tx * 4 + j]); atomicAdd(&sl[C + m.c + j], (float)redc[RES_MAXC + m.tx * 4 + j]); }
    }
    // dx = rstd * (dz*gamma - m1 - xhat*m2) with the per-thread constants folded
    const float rm1 = rstd * m1, rm2 = rstd * m2;
    float gr[4];
#pragma unroll
    for (int j = 0; j < 4; j++) gr[j] = ga[j] * rstd;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    char* dxs = (char*)(dx + (long)b * L * lddx);
    const char* dxrs = dxr ? (const char*)(dxr + (long)b * dy_rows(L, resample) * lddxr) : nullptr;
    const unsigned lddxb = (unsigned)lddx * (unsigned)sizeof(T), lddxrb = (unsigned)lddxr * (unsigned)sizeof(T);
    const char* dxr2s = dxr2 ? (const char*)(dxr2 + (long)b * L * lddxr2) : nullptr;
    const unsigned lddxr2b = (unsigned)lddxr2 * (unsigned)sizeof(T);
#pragma unroll
    for (int k = 0; k < RPT; k++) {
      const int l = k * m.TY + m.ty;
      if (l < L) {
        float v[4], o[4]; unpack4<T>(raw[k], v);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float xh = fmaf(v[j], rstd, -mean * rstd);
          o[j] = fmaf(d[k][j], gr[j], fmaf(xh, -rm2, -rm1));
        }
        if (dxr) {
          float e[4];
          if constexpr (RAW0) unpack4<T>(er[k], e); else load_dy_eff32<T>(dxrs, lddxrb, cb, l, resample, e);
#pragma unroll
          for (int j = 0; j < 4; j++) o[j] += e[j];
        }
        if (dxr2) {      // second addend at dx's own resolution (UNet skip gradient): saves a separate read-modify-write pass over dx
          float e[4];
          load4<T>((const T*)(dxr2s + ((unsigned)l * lddxr2b + cb)), e);
#pragma unroll
          for (int j = 0; j < 4; j++) o[j] += e[j];
        }
#pragma unroll
        for (int j = 0; j < 4; j++) cs[j] += o[j];
        store4<T>((T*)(dxs + ((unsigned)l * lddxb + cb)), o);
      }
    }
    if (colsum_ps) {
#pragma unroll
      for (int j = 0; j < 4; j++) atomicAdd(&redc[2 * RES_MAXC + m.tx * 4 + j], (double)cs[j]);
    }
  }
  GN_TSTAMP(5);
  if (colsum_ps) {
    __syncthreads();
    if (m.act && m.ty == 0) {
#pragma unroll
      for (int j = 0; j < 4; j++) colsum_ps[(long)b * ldps + m.c + j] = (float)redc[2 * RES_MAXC + m.tx * 4 + j];
    }
  }
}

// ------------------------------------------------------------------ pipelined form of the one-pass backward (round 4; bf16, resample 0)
// gn_bwd_resident_kernel is a chain per workgroup -- load the slab, pass 1, two barrier rounds, pass 2, store -- and with one 1024-thread
// workgroup per CU nothing overlaps the chain: HBM idles while the VALU runs and the other way round (in the LDM step 47 us per 50 MB
// tensor, 3.2 TB/s for x + dy + dx).  Here ONE persistent workgroup per CU walks slabs of the same size in threads (6 rows of 4 channels
// per thread = 48 KB per tensor), and the NEXT slab's x, dy, residual-path addend and (mean, rstd) pairs are fetched by LDS-DMA into a
// 3 x 48 KB landing area while the current slab is processed from registers: the loads of slab k+1 and the stores of slab k both run
// under the VALU phases.  gfx950 retires loads and stores through one in-order vmcnt, so "slab k+1 has landed" is vmcnt(#stores of slab k)
// = vmcnt(6): every vector-memory instruction a wave issues behind its DMAs is a store or a no-return atomic (gamma / beta are loaded
// once in front of the loop -- a block keeps its channel chunk -- and the statistics arrive by DMA as well), so the compiler never waits on
// vmcnt inside the loop and extra stores of some waves (column sums, slot atomics) only make the counted wait stricter.  Barriers are
// raw s_barrier with lgkmcnt(0): __syncthreads() carries a release fence = s_waitcnt vmcnt(0) = wait for the prefetch.
// Same arithmetic and the same fp64 LDS accumulators as the resident kernel: dx and the per-sample column sums are bit-identical to it.
constexpr int PIPE_RPT = 6;
constexpr int PIPE_SLAB = PIPE_RPT * NTB * 8;                 // bytes of one tensor's slab: 49 152
constexpr int PIPE_OFF_ST = 3 * PIPE_SLAB;                    // 1 KB landing area of the statistics DMA
constexpr int PIPE_OFF_RED = PIPE_OFF_ST + 1024;              // redg[2 * RES_MAXG] | redc[3 * RES_MAXC] (doubles)
constexpr int PIPE_LDS = PIPE_OFF_RED + (2 * RES_MAXG + 3 * RES_MAXC) * 8;      // 155 648 of 163 840

__device__ __forceinline__ void pipe_dma(const void* base, unsigned voff, unsigned lds_off) {      // as gemm_big.hip's dma16s
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  const unsigned long long b = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds_off)), "v"(voff), "s"(b) : "memory", "m0");
}
#define PIPE_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#ifdef EEG_STAGE_TIMING      // (make dbg, tools/debug/gn_pipe_timing.py) eight stamps per slab, thread 0, NO memory waits of their own
#define PIPE_TSTAMP(k, i) do { if (threadIdx.x == 0 && (k) < 8) gn_tlog[(size_t)(blockIdx.x % 512) * 64 + (k) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define PIPE_TSTAMP(k, i) do {} while (0)
#endif

template <bool SILU, bool HAS_ER, typename T16 = bf16_t>
__global__ __launch_bounds__(NTB) void gn_bwd_pipe_kernel(const bf16_t* __restrict__ x, long ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ stats,
                                                          const bf16_t* __restrict__ dy, long lddy, bf16_t* __restrict__ dx, long lddx,
                                                          const bf16_t* __restrict__ dxr, long lddxr, float* __restrict__ slots,
                                                          float* __restrict__ colsum_ps, long ldps, int B, int L, int C, int G, int CC, int ndg) {
  extern __shared__ __attribute__((aligned(16))) char psm[];
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  double* const redg = (double*)(psm + PIPE_OFF_RED);
  double* const redc = redg + 2 * RES_MAXG;
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane((unsigned)(tid >> 6));
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)psm;
  const int cpg = C / G, nchunk = C / CC;
  const int TX = CC >> 2, TY = NTB / TX;                  // L == PIPE_RPT * TY: every thread owns exactly PIPE_RPT rows
  const int tx = tid % TX, ty = tid / TX;
  // slab list: workgroup ids = xc (mod 8) run on XCD xc; within an XCD the chunks of one sample are neighbours (they share 128-byte lines
  // when a chunk row is 64 bytes) and a workgroup keeps its chunk for all its samples
  const int xc = blockIdx.x & 7, mb = blockIdx.x >> 3, MB = gridDim.x >> 3;
  const int chunk = mb % nchunk, slot = mb / nchunk, nslot = MB / nchunk;
  const int c = chunk * CC + tx * 4, gl = (tx * 4) / cpg;
  const int noct = B >> 3;
  const int nk = slot < noct ? (noct - slot + nslot - 1) / nslot : 0;
  float ga[4], gl2[4], bl2[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { ga[j] = gamma[c + j]; gl2[j] = ga[j] * 1.4426950408889634f; bl2[j] = beta[c + j] * 1.4426950408889634f; }
  for (int i = tid; i < 2 * RES_MAXG + 3 * RES_MAXC; i += NTB) redg[i] = 0.0;
  // the loads above have to be back before the first DMA: a compiler wait for them behind it would wait for the DMA too
  asm volatile("" :: "v"(ga[0]), "v"(ga[1]), "v"(ga[2]), "v"(ga[3]), "v"(bl2[0]), "v"(bl2[1]), "v"(bl2[2]), "v"(bl2[3]));
  // DMA lane map: 16-byte piece q = (i * 16 + wave) * 64 + lane of the slab, row q / LPR, piece q % LPR; LDS keeps the pieces in q order
  const int LPR = CC >> 3;
  const int drow = (int)(wave_u * 64 + lane) / LPR, dcol = lane % LPR;
  const unsigned vx = (unsigned)drow * (unsigned)ldx * 2u + (unsigned)dcol * 16u, sx = (unsigned)(1024 / LPR) * (unsigned)ldx * 2u;
  const unsigned vd = (unsigned)drow * (unsigned)lddy * 2u + (unsigned)dcol * 16u, sd = (unsigned)(1024 / LPR) * (unsigned)lddy * 2u;
  const unsigned ve = (unsigned)drow * (unsigned)lddxr * 2u + (unsigned)dcol * 16u, se = (unsigned)(1024 / LPR) * (unsigned)lddxr * 2u;
  const unsigned st_bytes = (unsigned)B * (unsigned)G * 8u;
  const unsigned st_sub = ((unsigned)(chunk * (CC / cpg)) * 8u) & 15u;        // (b G + g0) * 8 mod 16: G = 32 makes it independent of b
  // x, dy and the (mean, rstd) pairs of a slab: 7 DMA instructions per wave (every wave fetches the same statistics pieces to the same
  // place, so that all waves count alike); the residual-path addend goes separately: it is read from LDS in pass 2 only
  auto issue_xd = [&](int b) __attribute__((always_inline)) {
    const char* xs = (const char*)(x + (long)b * L * ldx + chunk * CC);
    const char* ds = (const char*)(dy + (long)b * L * lddy + chunk * CC);
#pragma unroll
    for (int i = 0; i < 3; i++) pipe_dma(xs, vx + i * sx, lds0 + (unsigned)(i * 16 + wave_u) * 1024u);
#pragma unroll
    for (int i = 0; i < 3; i++) pipe_dma(ds, vd + i * sd, lds0 + PIPE_SLAB + (unsigned)(i * 16 + wave_u) * 1024u);
    {       // 16-byte pieces from the aligned-down address of the chunk's first pair, clamped to the buffer (the needed pieces never are)
      const unsigned a0 = (((unsigned)b * (unsigned)G + (unsigned)(chunk * (CC / cpg))) * 8u) & ~15u;
      unsigned a = a0 + (unsigned)lane * 16u; if (a > st_bytes - 16u) a = st_bytes - 16u;
      pipe_dma(stats, a, lds0 + PIPE_OFF_ST);
    }
  };
  auto issue_e = [&](int b) __attribute__((always_inline)) {
    const char* es = (const char*)(dxr + (long)b * L * lddxr + chunk * CC);
#pragma unroll
    for (int i = 0; i < 3; i++) pipe_dma(es, ve + i * se, lds0 + 2 * PIPE_SLAB + (unsigned)(i * 16 + wave_u) * 1024u);
  };
  if (nk > 0) { issue_xd(slot * 8 + xc); if constexpr (HAS_ER) issue_e(slot * 8 + xc); }
  const float inv_n = 1.0f / ((float)cpg * (float)L);
  const unsigned rowb = (unsigned)CC * 2u;
  const unsigned lrd0 = (unsigned)ty * rowb + (unsigned)tx * 8u, lrds = (unsigned)TY * rowb;      // this thread's piece of row ty; rows step by TY
  const unsigned lddxb = (unsigned)lddx * 2u;
  const unsigned st0 = (unsigned)ty * lddxb + (unsigned)c * 2u, sts = (unsigned)TY * lddxb;      // the same for the dx rows (32-bit offsets off a uniform base)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int k = 0; k < nk; k++) {
    const int b = (slot + k * nslot) * 8 + xc;
    PIPE_TSTAMP(k, 0);
    PIPE_BARRIER();                                           // every wave's pieces of slab k have landed (counted wait at the bottom of slab k-1)
    PIPE_TSTAMP(k, 1);
    uint2 raw[PIPE_RPT], dr[PIPE_RPT];
#pragma unroll
    for (int r = 0; r < PIPE_RPT; r++) {
      const unsigned off = lrd0 + (unsigned)r * lrds;
      raw[r] = *(const uint2*)(psm + off); dr[r] = *(const uint2*)(psm + PIPE_SLAB + off);
    }
    const float2 mr = *(const float2*)(psm + PIPE_OFF_ST + st_sub + (unsigned)gl * 8u);
    PIPE_BARRIER();                                           // everybody holds its copy: the x / dy landing areas may be refilled
    PIPE_TSTAMP(k, 2);
    if (k + 1 < nk) issue_xd((slot + (k + 1) * nslot) * 8 + xc);
    PIPE_TSTAMP(k, 3);
    const float mean = mr.x, rstd = mr.y, nmr = -mean * rstd;
    float d[PIPE_RPT][4];
    {
      // the kernel is VALU-bound (tools/debug/gn_pipe_timing.py: pass 1 is 8 k of a slab's 12.5 k cycles), so SiLU' is taken from
      // t = z log2(e) directly: s = 1 / (1 + 2^-t), SiLU' = s (1 + z (1 - s)) = s fma(t, ln2 (1 - s), 1) -- one multiply less per element
      float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int r = 0; r < PIPE_RPT; r++) {
        float v[4]; unpack4<T16>(raw[r], v); unpack4<T16>(dr[r], d[r]);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float xh = fmaf(v[j], rstd, nmr);
          float dz = d[r][j];
          if constexpr (SILU) {
            const float t = fmaf(gl2[j], xh, bl2[j]);
            const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-t));
            dz *= sg * fmaf(t, fmaf(-0.6931471805599453f, sg, 0.6931471805599453f), 1.0f);
          }
          d[r][j] = dz;
          dg[j] = fmaf(dz, xh, dg[j]); db[j] += dz;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; j++) { atomicAdd(&redc[tx * 4 + j], (double)dg[j]); atomicAdd(&redc[RES_MAXC + tx * 4 + j], (double)db[j]); }
    }
    PIPE_TSTAMP(k, 4);
    PIPE_BARRIER();
    if (ty == 0) {
      double p1 = 0.0, p2 = 0.0;
#pragma unroll
      for (int j = 0; j < 4; j++) { p1 += (double)ga[j] * redc[RES_MAXC + tx * 4 + j]; p2 += (double)ga[j] * redc[tx * 4 + j]; }
      atomicAdd(&redg[2 * gl], p1); atomicAdd(&redg[2 * gl + 1], p2);
    }
    if constexpr (HAS_ER) {       // this slab's addend (issued at the end of slab k-1) is older than the 7 DMAs of slab k+1
      if (k + 1 < nk) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    PIPE_BARRIER();
    PIPE_TSTAMP(k, 5);
    {
      const float m1 = (float)(redg[2 * gl] * (double)inv_n), m2 = (float)(redg[2 * gl + 1] * (double)inv_n);
      if (slots && ty == 0) {
        float* sl = slots + (size_t)(b % ndg) * 2 * C;
#pragma unroll
        for (int j = 0; j < 4; j++) { atomicAdd(&sl[c + j], (float)redc[tx * 4 + j]); atomicAdd(&sl[C + c + j], (float)redc[RES_MAXC + tx * 4 + j]); }
      }
      // dx = rstd (dz gamma - m1 - xhat m2), xhat = x rstd + nmr, as two FMAs per element on the raw x: the per-thread constants carry the rest
      const float rm1 = rstd * m1, rm2 = rstd * m2;
      const float a2 = -rm2 * rstd, b2 = fmaf(-rm2, nmr, -rm1);
      float gr[4];
#pragma unroll
      for (int j = 0; j < 4; j++) gr[j] = ga[j] * rstd;
      float cs[4] = {0.f, 0.f, 0.f, 0.f};
      char* dxs = (char*)(dx + (long)b * L * lddx);
#pragma unroll
      for (int r = 0; r < PIPE_RPT; r++) {
        float v[4], o[4]; unpack4<T16>(raw[r], v);
#pragma unroll
        for (int j = 0; j < 4; j++) o[j] = fmaf(d[r][j], gr[j], fmaf(v[j], a2, b2));
        if constexpr (HAS_ER) {
          float e[4]; unpack4<T16>(*(const uint2*)(psm + 2 * PIPE_SLAB + lrd0 + (unsigned)r * lrds), e);
#pragma unroll
          for (int j = 0; j < 4; j++) o[j] += e[j];
        }
        if (colsum_ps) {
#pragma unroll
          for (int j = 0; j < 4; j++) cs[j] += o[j];
        }
        store4<T16>((T16*)(dxs + (st0 + (unsigned)r * sts)), o);
      }
      if (colsum_ps) {
#pragma unroll
        for (int j = 0; j < 4; j++) atomicAdd(&redc[2 * RES_MAXC + tx * 4 + j], (double)cs[j]);
      }
    }
    PIPE_TSTAMP(k, 6);
    PIPE_BARRIER();
    if (ty == 0) {       // column sums out, this thread's accumulators back to zero (next touched behind the two barriers at the top of slab k+1)
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (colsum_ps) colsum_ps[(long)b * ldps + c + j] = (float)redc[2 * RES_MAXC + tx * 4 + j];
        redc[tx * 4 + j] = 0.0; redc[RES_MAXC + tx * 4 + j] = 0.0; redc[2 * RES_MAXC + tx * 4 + j] = 0.0;
      }
    }
    if (tid < 2 * RES_MAXG) redg[tid] = 0.0;
    // x / dy of slab k+1 have landed when at most this slab's PIPE_RPT stores and the three addend DMAs, all issued behind them, are in flight
    if (k + 1 < nk) {
      if constexpr (HAS_ER) { issue_e((slot + (k + 1) * nslot) * 8 + xc); asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    }
    PIPE_TSTAMP(k, 7);
  }
}


// chunk width for the resident kernels: the widest whole-group chunk (<= 256 channels, dividing C) whose rows fit the
// per-thread register budget; 0 = not eligible (fall back to the split kernels)
int resident_chunk(int L, int C, int G, int resample_pair, int rpt_max, int* rpt_out, int nth = NTB) {
  const int cpg = C / G;
  if (C % 4 != 0 || cpg % 4 != 0) return 0;
  int best = 0, best_rpt = 0;
  for (int mlt = 1; mlt * cpg <= RES_MAXC && mlt <= RES_MAXG; mlt++) {
    const int cc = mlt * cpg;
    if (C % cc != 0) continue;
    const int tx = cc / 4, ty = nth / tx;
    if (ty < 1) break;
    int rpt = resample_pair ? 2 * ((L / 2 + ty - 1) / ty) : (L + ty - 1) / ty;
    if (rpt > rpt_max) break;
    best = cc; best_rpt = rpt;
  }
  *rpt_out = best_rpt;
  return best;
}

int grid_for(long total_threads, eegldm_ctx* ctx) {
  long blocks = (total_threads + NT - 1) / NT;
  long cap = (long)ctx->num_cu * 16;
  return (int)(blocks < cap ? (blocks < 1 ? 1 : blocks) : cap);
}

// row-chunk split of one sample: `per_cu` blocks per CU in total, at least `min_rows` rows per block.
// Reductions want few long chunks (fewer atomics), streaming apply kernels want many short ones.
int pick_lsplit(int B, int L, int C, eegldm_ctx* ctx, int* rows_per_block, int per_cu = 8, int min_rows = 16) {
  int want = (ctx->num_cu * per_cu + B - 1) / B;
  int maxsplit = (L + min_rows - 1) / min_rows;
  int ls = want < 1 ? 1 : (want > maxsplit ? maxsplit : want);
  int rpb = (L + ls - 1) / ls;
  *rows_per_block = rpb;
  return (L + rpb - 1) / rpb;
}

template <typename T, int V>
int gn_fwd_t(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta, void* y, long ldy,
             float* stats, int B, int L, int C, int G, float eps, int silu, int resample, void* xr, long ldxr) {
  if constexpr (V == 4) {
    EEG_ENV_VAR(bool, off, getenv("EEGLDM_GN_NO_RESIDENT") != nullptr);
    // 12 rows per thread (56 VGPRs: two 1024-thread blocks per CU) measured faster than 24 (one block per CU): 24 vs 35 us
    constexpr int fwd_rpt_max = 12;
    // threads per block: 1024 = one (sample, 64-channel) slab per block; 512 / 256 = narrower slabs (32 / 16 channels), 2 / 4x as many
    // independent blocks per CU whose load / reduce / store phases interleave (EEGLDM_GN_FWD_NTH; narrow slabs use the XCD-aware order)
    EEG_ENV_VAR(int, fwd_nth, getenv("EEGLDM_GN_FWD_NTH") ? atoi(getenv("EEGLDM_GN_FWD_NTH")) : 1024);
    int nth = (fwd_nth == 512 || fwd_nth == 256) ? fwd_nth : 1024;
    int rpt = 0; int cc = off ? 0 : resident_chunk(L, C, G, resample == 1, fwd_rpt_max, &rpt, nth);
    // a launch that leaves most of the chip idle (sampling one window per call: 1-8 slabs) is a latency chain, not a bandwidth problem:
    // 16-channel slabs on 256 threads are 4x as many, shorter blocks (DDIM-50 at B = 1: 71.5 -> 64.5 ms).  The fp64 group sums make
    // the statistics independent of the slab shape, so the outputs do not change.
    EEG_ENV_VAR(bool, no_few, getenv("EEGLDM_GN_NO_FEW_SLAB_NARROW") != nullptr);
    if (!no_few && cc && nth == 1024 && (long)(C / cc) * B <= ctx->num_cu / 4) {
      int rpt2 = 0; const int cc2 = resident_chunk(L, C, G, resample == 1, fwd_rpt_max, &rpt2, 256);
      if (cc2 && cc2 * (int)sizeof(T) >= 32) { nth = 256; cc = cc2; rpt = rpt2; }
    }
    // measured (tools/debug/gn_bench.py): wins while the rows are at least a 128-byte line and the blocks fit two rounds
    // (1024 blocks = the 100 MB concat tensors of the up path: 47-48 us one-pass vs 51 us split)
    constexpr long fwd_bpc = 4;
    constexpr bool no_xcd = false;
    constexpr int narrow_min = 32;    // 32-byte rows (16-channel slabs) are fine under the XCD-aware order: L = 3072 runs one-pass (pixel-space step 22.84 -> 22.64 ms); 128 = round-2 behaviour
    const bool can_xcd = !no_xcd && B % 8 == 0;
    const int min_row = nth == 1024 ? (can_xcd ? narrow_min : 128) : 32;
    if (cc && (cc * (int)sizeof(T) < min_row || (long)(C / cc) * B > fwd_bpc * ctx->num_cu * (1024 / nth))) cc = 0;
    if (cc) {
      const int xcd = ((nth < 1024 || cc * (int)sizeof(T) < 128) && can_xcd) ? 1 : 0;
      const dim3 grid((unsigned)((long)(C / cc) * B));
#define GN_FWD_RES1(R, N) hipLaunchKernelGGL((gn_fwd_resident_kernel<T, R, N>), grid, dim3(N), 0, ctx->stream, (const T*)x, ldx, gamma, beta, \
                                         (T*)y, ldy, (T*)xr, ldxr, stats, L, C, G, eps, silu, resample, cc, xcd)
#define GN_FWD_RES(R) do { if (nth == 1024) GN_FWD_RES1(R, 1024); else if (nth == 512) GN_FWD_RES1(R, 512); else GN_FWD_RES1(R, 256); } while (0)
      if (rpt <= 6) GN_FWD_RES(6); else if (rpt <= 12) GN_FWD_RES(12); else { if constexpr (sizeof(T) == 2) GN_FWD_RES(24); }
#undef GN_FWD_RES
#undef GN_FWD_RES1
      LAUNCH_CHECK();
      return 0;
    }
  }
  double* sums = (double*)ctx->scratch;      // zero on entry (context creation / previous finalize)
  int rpb; int ls = pick_lsplit(B, L, C, ctx, &rpb);
  hipLaunchKernelGGL((gn_stats_kernel<T, V>), dim3(ls, B), dim3(NT), 0, ctx->stream, (const T*)x, ldx, sums, L, C, G, rpb);
  LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((B * G + 255) / 256), dim3(256), 0, ctx->stream, sums, stats, B * G,
                     (double)(C / G) * (double)L, eps);
  LAUNCH_CHECK();
  int rpb2; int ls2 = pick_lsplit(B, resample == 1 ? L / 2 : L, C, ctx, &rpb2, 16, 8);
  hipLaunchKernelGGL((gn_apply_kernel<T, V>), dim3(ls2, B), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma,
                     beta, stats, (T*)y, ldy, (T*)xr, ldxr, L, C, G, silu, resample, rpb2);
  LAUNCH_CHECK();
  return 0;
}

template <typename T, int V>
int gn_bwd_t(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta, const float* stats,
             const void* dy, long lddy, void* dx, long lddx, float* dgamma, float* dbeta, int B, int L, int C, int G,
             int silu, int resample, const void* dxr, long lddxr, float* colsum_ps, long ldps, int* colsum_done,
             const void* dxr2, long lddxr2, int* dxr2_done, int* slots_deferred, int defer_region) {
  if (colsum_done) *colsum_done = 0;
  if (dxr2_done) *dxr2_done = 0;
  if (slots_deferred) *slots_deferred = 0;
  if (!dxr2_done || lddxr2 % 4 != 0) dxr2 = nullptr;      // only a caller that can fall back may hand over a second addend
  if constexpr (V == 4) {
    EEG_ENV_VAR(bool, off, getenv("EEGLDM_GN_NO_RESIDENT") != nullptr);
    EEG_ENV_VAR(int, bwd_nth, getenv("EEGLDM_GN_BWD_NTH") ? atoi(getenv("EEGLDM_GN_BWD_NTH")) : 1024);      // see gn_fwd_t
    // 256-thread blocks (EEGLDM_GN_BWD_NTH=256) measure 42-44 vs 47-49 us on the 50 MB tensors at L <= 384 (tools/debug/gn_nth.py); at step
    // level they gain nothing (19.76 vs 19.67 ms), so the default stays at 1024 threads.
    // HAZARD: until round 4 these blocks returned wrong sums (~1e-3, bf16 rounding flips over whole slabs) whenever they
    // shared a CU with the fused 3-tap weight-gradient GEMM (LDS-DMA build) of another stream -- 1024-thread blocks own a CU and never
    // share it.  Round-4 diagnosis (tools/debug/gn_hazard.sh, gn_hazard_diag.py): the errors sit in EVEN channels only = the low lane of
    // the v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32 instructions hipcc's SLP vectoriser makes of the per-channel math; LDS atomics,
    // LDS stomping and late DMA are not involved (the sums are wrong with the atomics replaced by single-writer stores; right with
    // the packed instructions gone).  The library is now built with -fno-slp-vectorize -fno-vectorize (Makefile), which removes the error
    // (0 of 108 noisy runs against 27 of 27).  The fence below stays as a second line: narrow blocks are refused whenever anything of this
    // library can run beside them -- the opt-in side stream, or a second live context in this process (g_eeg_live_ctx, api.hip).
    // EEGLDM_GN_NARROW_UNFENCED=1 (developer / regression switch): narrow blocks even beside a second context -- the configuration that
    // returned wrong sums until the library was built without packed-fp32 instructions (tests/test_gpu_concurrency.py runs it)
    EEG_ENV_VAR(bool, unfenced, getenv("EEGLDM_GN_NARROW_UNFENCED") != nullptr);
#ifdef EEG_GN_NO_FENCE      // developer build (tools/debug/gn_hazard.sh)
    const int nth = (bwd_nth == 512 || bwd_nth == 256) ? bwd_nth : 1024;
#else
    const int nth = ((bwd_nth == 512 || bwd_nth == 256) && (unfenced || (!ctx->side_on && g_eeg_live_ctx <= 1))) ? bwd_nth : 1024;
#endif
    if constexpr (sizeof(T) == 2) {
      // pipelined persistent form (gn_bwd_pipe_kernel): slabs of exactly 6 rows x 4 channels per thread, L * CC = 24 576, CC a power of two
      EEG_ENV_VAR(bool, no_pipe, getenv("EEGLDM_GN_NO_PIPE") != nullptr);
      constexpr int pipe_min_row = 64;
      EEG_ENV_VAR(int, pipe_min_slabs, getenv("EEGLDM_GN_PIPE_MIN_SLABS") ? atoi(getenv("EEGLDM_GN_PIPE_MIN_SLABS")) : 2);
      EEG_ENV_VAR(int, pipe_max_slot, getenv("EEGLDM_GN_PIPE_MAX_SLOT") ? atoi(getenv("EEGLDM_GN_PIPE_MAX_SLOT")) : 1 << 20);
      // with a residual-path addend (a fourth 48 KB stream per slab) the kernel is bandwidth-bound like the resident one and measures
      // 3-5 % slower than it (tools/debug/gn_pipe_bench.py): those launches keep the resident kernel unless EEGLDM_GN_PIPE_ADDEND=1
      EEG_ENV_VAR(bool, pipe_addend, getenv("EEGLDM_GN_PIPE_ADDEND") != nullptr);
      const int cpg = C / G;
      const int pcc = (L > 0 && (PIPE_RPT * NTB * 4) % L == 0) ? (PIPE_RPT * NTB * 4) / L : 0;
      const bool pow2 = pcc >= 16 && pcc <= RES_MAXC && (pcc & (pcc - 1)) == 0;
      auto al16 = [](const void* p) { return ((size_t)p & 15) == 0; };
      if (!no_pipe && !off && resample == 0 && !dxr2 && (!dxr || pipe_addend) && pow2 && pcc * 2 >= pipe_min_row && C % pcc == 0 && C % G == 0 && pcc % cpg == 0 &&
          pcc / cpg <= RES_MAXG && cpg % 4 == 0 && (G * 8) % 16 == 0 && B % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 && lddx % 4 == 0 &&
          (!dxr || lddxr % 8 == 0) && al16(x) && al16(dy) && al16(stats) && (!dxr || al16(dxr)) && (long)L * (ldx > lddy ? ldx : lddy) * 2 < (1l << 31)) {
        const int nchunk = C / pcc, per_xcd = ctx->num_cu / 8;
        int nslot = per_xcd / nchunk; if (nslot > B / 8) nslot = B / 8; if (nslot > pipe_max_slot) nslot = pipe_max_slot;
        if (nslot >= 1 && (long)nchunk * B >= (long)pipe_min_slabs * 8 * nslot * nchunk) {
          float* slots = nullptr;
          constexpr bool no_defer_p = false;
          const bool det = eeg_deterministic() && dgamma;      // a slot per sample (one writer each), folded right behind the launch in sample order
          const bool defer = slots_deferred && dgamma && !no_defer_p && !det;
          int ndg = GN_NSLOT;
          if (dgamma) slots = (float*)((char*)ctx->scratch + (defer ? gn_slot_region(defer_region) : GN_SLOT_OFFSET));
          if (det) {
            ndg = B;
            EEG_TRY(eeg_det_buffer(ctx, (size_t)ndg * 2 * C * sizeof(float), &slots));
            HIP_TRY(hipMemsetAsync(slots, 0, (size_t)ndg * 2 * C * sizeof(float), ctx->stream));
          }
          constexpr bool no_batch_p = false;
          const bool batched = defer && ctx->defer_wgrad && !no_batch_p && C <= 1024 && ctx->gn_fold_count < GN_FOLD_MAX;
          if (batched) {
            if (!ctx->gn_slot_arena) {
              HIP_TRY(hipMalloc(&ctx->gn_slot_arena, (size_t)GN_FOLD_MAX * GN_REGION_FLOATS * sizeof(float)));
              HIP_TRY(hipMemsetAsync(ctx->gn_slot_arena, 0, (size_t)GN_FOLD_MAX * GN_REGION_FLOATS * sizeof(float), ctx->stream));
            }
            slots = ctx->gn_slot_arena + (size_t)ctx->gn_fold_count * GN_REGION_FLOATS;
            ctx->gn_fold_pending.push_back({slots, dgamma, dbeta, C});
            ctx->gn_fold_count++;
          }
          static DevOnce attr_once;
          if (attr_once.need(ctx->device)) {
            HIP_TRY(hipFuncSetAttribute((const void*)gn_bwd_pipe_kernel<true, true, T>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS));
            HIP_TRY(hipFuncSetAttribute((const void*)gn_bwd_pipe_kernel<true, false, T>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS));
            HIP_TRY(hipFuncSetAttribute((const void*)gn_bwd_pipe_kernel<false, true, T>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS));
            HIP_TRY(hipFuncSetAttribute((const void*)gn_bwd_pipe_kernel<false, false, T>, hipFuncAttributeMaxDynamicSharedMemorySize, PIPE_LDS));
          }
          const dim3 grid((unsigned)(8 * nslot * nchunk));
#define GN_BWD_PIPE(SL, ER) hipLaunchKernelGGL((gn_bwd_pipe_kernel<SL, ER, T>), grid, dim3(NTB), PIPE_LDS, ctx->stream, (const bf16_t*)x, ldx, gamma, beta, stats, \
                                           (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, (const bf16_t*)dxr, lddxr, slots, colsum_ps, ldps, B, L, C, G, pcc, ndg)
          if (silu) { if (dxr) GN_BWD_PIPE(true, true); else GN_BWD_PIPE(true, false); }
          else { if (dxr) GN_BWD_PIPE(false, true); else GN_BWD_PIPE(false, false); }
#undef GN_BWD_PIPE
          LAUNCH_CHECK();
          if (det) {
            EEG_TRY(ew_fold_partials_det(ctx, slots, ndg, 2 * C, 0, C, dgamma)); EEG_TRY(ew_fold_partials_det(ctx, slots, ndg, 2 * C, C, C, dbeta));
          } else if (slots && !defer) {
            hipLaunchKernelGGL(gn_slot_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, ctx->stream, slots, dgamma, dbeta, C, (double*)ctx->scratch, 0, ndg);
            LAUNCH_CHECK();
          }
          if (defer) *slots_deferred = batched ? 2 : 1;
          if (colsum_done && colsum_ps) *colsum_done = 1;
          return 0;
        }
      }
    }
    int rpt = 0; int cc = off ? 0 : resident_chunk(L, C, G, 0, sizeof(T) == 2 ? 12 : 8, &rpt, nth);
    // measured (tools/debug/gn_bench.py): the one-pass kernel wins while its blocks fit two rounds of one block per CU
    // (1024 blocks: 112 us one-pass vs 117-120 us split on the 100 MB tensors; 2048 blocks of 96-channel chunks lose)
    constexpr long bwd_bpc = 8;
    constexpr int bwd_minrow = 128, narrow_min = 32;
    constexpr bool no_xcd = false;
    const bool can_xcd = !no_xcd && B % 8 == 0;
    const int min_row = nth == 1024 ? (can_xcd && narrow_min < bwd_minrow ? narrow_min : bwd_minrow) : 32;
    if (cc && (cc * (int)sizeof(T) < min_row || (long)(C / cc) * B > bwd_bpc * ctx->num_cu * (1024 / nth))) cc = 0;
    if (cc) {
      const int xcd = ((nth < 1024 || cc * (int)sizeof(T) < 128) && can_xcd) ? 1 : 0;
      const dim3 grid((unsigned)((long)(C / cc) * B));
      // a caller that can run the 7-us fold of the dgamma / dbeta partial slots elsewhere (side stream) gets them in the second slot
      // region and calls op_gn_slot_reduce_deferred itself
      constexpr bool no_defer = false;
      const bool det = eeg_deterministic() && dgamma;      // see the pipelined launch above
      const bool defer = slots_deferred && dgamma && !no_defer && !det;
      float* slots = dgamma ? (float*)((char*)ctx->scratch + (defer ? gn_slot_region(defer_region) : GN_SLOT_OFFSET)) : nullptr;
      int nslot = GN_NSLOT;
      if (det) {
        nslot = B;
        EEG_TRY(eeg_det_buffer(ctx, (size_t)nslot * 2 * C * sizeof(float), &slots));
        HIP_TRY(hipMemsetAsync(slots, 0, (size_t)nslot * 2 * C * sizeof(float), ctx->stream));
      }
      // batched mode (the UNet backward with grouped weight gradients): this launch gets its OWN slot region and is folded together with
      // all the others by op_gn_fold_flush -- 49 folds of 7 us become one launch
      constexpr bool no_batch = false;
      const bool batched = defer && ctx->defer_wgrad && !no_batch && C <= 1024 && ctx->gn_fold_count < GN_FOLD_MAX;
      if (batched) {
        if (!ctx->gn_slot_arena) {
          HIP_TRY(hipMalloc(&ctx->gn_slot_arena, (size_t)GN_FOLD_MAX * GN_REGION_FLOATS * sizeof(float)));
          HIP_TRY(hipMemsetAsync(ctx->gn_slot_arena, 0, (size_t)GN_FOLD_MAX * GN_REGION_FLOATS * sizeof(float), ctx->stream));
        }
        slots = ctx->gn_slot_arena + (size_t)ctx->gn_fold_count * GN_REGION_FLOATS;
        ctx->gn_fold_pending.push_back({slots, dgamma, dbeta, C});
        ctx->gn_fold_count++;
      }
#define GN_BWD_RES3(R, RAW, SL, N) hipLaunchKernelGGL((gn_bwd_resident_kernel<T, R, RAW, SL, N>), grid, dim3(N), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats, \
                                         (const T*)dy, lddy, (T*)dx, lddx, (const T*)dxr, lddxr, slots, colsum_ps, ldps, L, C, G, silu, resample, cc, (const T*)dxr2, lddxr2, xcd, nslot)
#define GN_BWD_RES2(R, RAW, SL) do { if (nth == 1024) GN_BWD_RES3(R, RAW, SL, 1024); else if (nth == 512) GN_BWD_RES3(R, RAW, SL, 512); else GN_BWD_RES3(R, RAW, SL, 256); } while (0)
      constexpr bool raw0 = true;
#define GN_BWD_RES1(R, RAW) do { if (silu) GN_BWD_RES2(R, RAW, true); else GN_BWD_RES2(R, RAW, false); } while (0)
#define GN_BWD_RES(R) do { if (resample == 0 && raw0) GN_BWD_RES1(R, true); else GN_BWD_RES1(R, false); } while (0)
      constexpr int RLO = sizeof(T) == 2 ? 6 : 4, RHI = sizeof(T) == 2 ? 12 : 8;
      if (rpt <= RLO) GN_BWD_RES(RLO); else GN_BWD_RES(RHI);
#undef GN_BWD_RES
#undef GN_BWD_RES1
#undef GN_BWD_RES2
#undef GN_BWD_RES3
      LAUNCH_CHECK();
      if (det) {
        EEG_TRY(ew_fold_partials_det(ctx, slots, nslot, 2 * C, 0, C, dgamma)); EEG_TRY(ew_fold_partials_det(ctx, slots, nslot, 2 * C, C, C, dbeta));
      } else if (slots && !defer) {
        hipLaunchKernelGGL(gn_slot_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, ctx->stream, slots, dgamma, dbeta, C, (double*)ctx->scratch, 0, nslot);
        LAUNCH_CHECK();
      }
      if (defer) *slots_deferred = batched ? 2 : 1;       // 2: nothing left for the caller to fold
      if (colsum_done && colsum_ps) *colsum_done = 1;
      if (dxr2) *dxr2_done = 1;
      return 0;
    }
  }
  double* gsums = (double*)ctx->scratch;     // zero on entry; shared with the forward sums (stream-ordered)
  int rpb; int ls = pick_lsplit(B, L, C, ctx, &rpb);
  float* slots = dgamma ? (float*)((char*)ctx->scratch + GN_SLOT_OFFSET) : nullptr;
  int nslot = GN_NSLOT;
  if (slots && eeg_deterministic()) {      // a slot per block: every partial row has one writer, the fold adds the rows in order
    nslot = ls * B;
    EEG_TRY(eeg_det_buffer(ctx, (size_t)nslot * 2 * C * sizeof(float), &slots));
    HIP_TRY(hipMemsetAsync(slots, 0, (size_t)nslot * 2 * C * sizeof(float), ctx->stream));
  }
#define GN_BWD_SPLIT(RS) do { \
    hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, V, RS>), dim3(ls, B), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, beta, stats, \
                       (const T*)dy, lddy, gsums, slots, L, C, G, silu, rpb, nslot); \
    hipLaunchKernelGGL((gn_bwd_apply_kernel<T, V, RS>), dim3(ls2, B), dim3(NT), 0, ctx->stream, (const T*)x, ldx, gamma, \
                       beta, stats, (const T*)dy, lddy, gsums, (T*)dx, lddx, (const T*)dxr, lddxr, L, C, G, silu, rpb2); } while (0)
  int rpb2; int ls2 = pick_lsplit(B, L, C, ctx, &rpb2, 16, 8);
  if (resample == 0) GN_BWD_SPLIT(0); else if (resample == 1) GN_BWD_SPLIT(1); else GN_BWD_SPLIT(2);
#undef GN_BWD_SPLIT
  LAUNCH_CHECK();
  const int n_gs = 2 * B * G;
  hipLaunchKernelGGL(gn_slot_reduce_kernel, dim3(((2 * C > n_gs ? 2 * C : n_gs) + 255) / 256), dim3(256), 0, ctx->stream, slots, dgamma, dbeta, C, gsums, n_gs, nslot);
  LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ thin AutoencoderKL layers: G = 1, C in {1, 2, 4, 8}, contiguous samples
// (config_aekl_eeg_2_2_4_spec.yaml: num_channels [2, 2, 4], norm_num_groups 1).  A sample is one flat run of L*C <= 24576 elements:
// one 256-thread block keeps it in registers as 16-byte chunks of 8 elements (element j of every chunk is channel j % C), so the
// statistics, the normalisation and the whole backward are ONE launch each instead of stats + finalize + apply /
// reduce + apply (2048 blocks of one row per thread and 256-way LDS atomics on these shapes: 40 us for a 1.5 MB tensor).
constexpr int FLAT_NT = 256;
template <typename T> struct Chunk8;
template <> struct Chunk8<bf16_t> {
  typedef uint4 raw_t;
  static __device__ __forceinline__ raw_t load(const bf16_t* p) { return *(const uint4*)p; }
  static __device__ __forceinline__ void unpack(const raw_t& r, float v[8]) {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u); v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u); v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
  }
  // opaque to the optimiser: keeps the packed chunk (not its eight unpacked floats) live between two passes
  static __device__ __forceinline__ void touch(raw_t& r) { asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)); }
  static __device__ __forceinline__ void store(bf16_t* p, const float v[8]) {
    uint4 t; t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]); t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
    *(uint4*)p = t;
  }
};
template <> struct Chunk8<f16_t> {
  typedef uint4 raw_t;
  static __device__ __forceinline__ raw_t load(const f16_t* p) { return *(const uint4*)p; }
  static __device__ __forceinline__ void unpack(const raw_t& r, float v[8]) {
    v[0] = w16_lo<f16_t>(r.x); v[1] = w16_hi<f16_t>(r.x); v[2] = w16_lo<f16_t>(r.y); v[3] = w16_hi<f16_t>(r.y);
    v[4] = w16_lo<f16_t>(r.z); v[5] = w16_hi<f16_t>(r.z); v[6] = w16_lo<f16_t>(r.w); v[7] = w16_hi<f16_t>(r.w);
  }
  static __device__ __forceinline__ void touch(raw_t& r) { asm volatile("" : "+v"(r.x), "+v"(r.y), "+v"(r.z), "+v"(r.w)); }
  static __device__ __forceinline__ void store(f16_t* p, const float v[8]) {
    uint4 t; t.x = pack_f16x2(v[0], v[1]); t.y = pack_f16x2(v[2], v[3]); t.z = pack_f16x2(v[4], v[5]); t.w = pack_f16x2(v[6], v[7]);
    *(uint4*)p = t;
  }
};
template <> struct Chunk8<float> {
  struct raw_t { float4 a, b; };
  static __device__ __forceinline__ raw_t load(const float* p) { raw_t r; r.a = *(const float4*)p; r.b = *(const float4*)(p + 4); return r; }
  static __device__ __forceinline__ void unpack(const raw_t& r, float v[8]) {
    v[0] = r.a.x; v[1] = r.a.y; v[2] = r.a.z; v[3] = r.a.w; v[4] = r.b.x; v[5] = r.b.y; v[6] = r.b.z; v[7] = r.b.w;
  }
  static __device__ __forceinline__ void touch(raw_t&) {}
  static __device__ __forceinline__ void store(float* p, const float v[8]) {
    *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); *(float4*)(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
};
// sum of K per-thread values over the block (4 waves): result in every thread
template <int K, int NW = 4> __device__ __forceinline__ void flat_block_sum(float (&v)[K], float* sm) {
#pragma unroll
  for (int k = 0; k < K; k++)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; k++) sm[wave * K + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; k++) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 4) t += (sm[w * K + k] + sm[(w + 1) * K + k]) + (sm[(w + 2) * K + k] + sm[(w + 3) * K + k]);
    v[k] = t;
  }
  __syncthreads();
}

template <typename T, int C, int MAXCH>
__global__ __launch_bounds__(FLAT_NT) void gn_flat_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              T* __restrict__ y, float* __restrict__ stats, int n, float eps, int silu) {
  __shared__ float sm[4];
  const int b = blockIdx.x, tid = threadIdx.x, nch = n >> 3;
  const T* xs = x + (long)b * n; T* ys = y + (long)b * n;
  typename Chunk8<T>::raw_t raw[MAXCH];
#pragma unroll
  for (int k = 0; k < MAXCH; k++) { const int ci = k * FLAT_NT + tid; if (ci < nch) raw[k] = Chunk8<T>::load(xs + (long)ci * 8); }
  float s[1] = {0.f};
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    if (k * FLAT_NT + tid < nch) {
      float v[8]; Chunk8<T>::unpack(raw[k], v);
      s[0] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  flat_block_sum<1>(s, sm);
  const float inv_n = 1.0f / (float)n, mean = s[0] * inv_n;
  float q[1] = {0.f};
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    if (k * FLAT_NT + tid < nch) {
      float v[8]; Chunk8<T>::unpack(raw[k], v);
#pragma unroll
      for (int j = 0; j < 8; j++) { const float d = v[j] - mean; q[0] = fmaf(d, d, q[0]); }
    }
  }
  flat_block_sum<1>(q, sm);
  const float rstd = rsqrtf(q[0] * inv_n + eps);
  if (tid == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
  float ga[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { ga[j] = gamma[j % C]; be[j] = beta[j % C]; }
  const float nmr = -mean * rstd;
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    const int ci = k * FLAT_NT + tid;
    if (ci < nch) {
      float v[8], o[8]; Chunk8<T>::unpack(raw[k], v);
#pragma unroll
      for (int j = 0; j < 8; j++) { const float z = fmaf(fmaf(v[j], rstd, nmr), ga[j], be[j]); o[j] = silu ? silu_f(z) : z; }
      Chunk8<T>::store(ys + (long)ci * 8, o);
    }
  }
}

template <typename T, int C, int MAXCH>
__global__ __launch_bounds__(FLAT_NT) void gn_flat_bwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const float* __restrict__ stats, const T* __restrict__ dy, T* __restrict__ dx,
                                                              const T* __restrict__ dxr, float* __restrict__ slots, int n, int silu, int nslot) {
  __shared__ float sm[4 * 2 * C];
  const int b = blockIdx.x, tid = threadIdx.x, nch = n >> 3;
  const T* xs = x + (long)b * n; const T* dys = dy + (long)b * n; T* dxs = dx + (long)b * n;
  typename Chunk8<T>::raw_t raw[MAXCH], rd[MAXCH];
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    const int ci = k * FLAT_NT + tid;
    if (ci < nch) { raw[k] = Chunk8<T>::load(xs + (long)ci * 8); rd[k] = Chunk8<T>::load(dys + (long)ci * 8); }
  }
  const float mean = stats[2 * b], rstd = stats[2 * b + 1], nmr = -mean * rstd;
  float ga[8], be[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { ga[j] = gamma[j % C]; be[j] = beta[j % C]; }
  float dz[MAXCH][8];
  float dg[8], db[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { dg[j] = 0.f; db[j] = 0.f; }
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    if (k * FLAT_NT + tid < nch) {
      float v[8], e[8]; Chunk8<T>::unpack(raw[k], v); Chunk8<T>::unpack(rd[k], e);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const float xh = fmaf(v[j], rstd, nmr);
        const float z = silu ? e[j] * silu_grad_f(fmaf(ga[j], xh, be[j])) : e[j];
        dz[k][j] = z; dg[j] = fmaf(z, xh, dg[j]); db[j] += z;
      }
    }
  }
  float r[2 * C];
#pragma unroll
  for (int c = 0; c < C; c++) {
    float a = 0.f, bb = 0.f;
#pragma unroll
    for (int j = c; j < 8; j += C) { a += dg[j]; bb += db[j]; }
    r[c] = a; r[C + c] = bb;
  }
  flat_block_sum<2 * C>(r, sm);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < C; c++) { s1 = fmaf(ga[c], r[C + c], s1); s2 = fmaf(ga[c], r[c], s2); }
  if (slots && tid == 0) {
    float* sl = slots + (size_t)(b % nslot) * 2 * C;
#pragma unroll
    for (int c = 0; c < 2 * C; c++) atomicAdd(&sl[c], r[c]);
  }
  const float inv_n = 1.0f / (float)n;
  const float rm1 = rstd * s1 * inv_n, rm2 = rstd * s2 * inv_n;
  float gr[8];
#pragma unroll
  for (int j = 0; j < 8; j++) gr[j] = ga[j] * rstd;
  const T* dxrs = dxr ? dxr + (long)b * n : nullptr;
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    const int ci = k * FLAT_NT + tid;
    if (ci < nch) {
      float v[8], o[8]; Chunk8<T>::unpack(raw[k], v);
#pragma unroll
      for (int j = 0; j < 8; j++) o[j] = fmaf(dz[k][j], gr[j], fmaf(fmaf(v[j], rstd, nmr), -rm2, -rm1));
      if (dxrs) {
        float e[8]; const typename Chunk8<T>::raw_t t = Chunk8<T>::load(dxrs + (long)ci * 8); Chunk8<T>::unpack(t, e);
#pragma unroll
        for (int j = 0; j < 8; j++) o[j] += e[j];
      }
      Chunk8<T>::store(dxs + (long)ci * 8, o);
    }
  }
}

// Same idea for the [32,32,64] AutoencoderKL (frozen encoder in front of every LDM step, decoder after sampling): G = 1, C a
// multiple of 8 that divides 8192, a sample of up to 98 304 contiguous elements resident in 1024 threads x 12 chunks.  With
// 1024 * 8 a multiple of C every thread always sees the same 8 channels.  1R + 1W instead of the split kernels' 2R + 1W.
constexpr int WIDE_NT = 1024;
template <typename T, int MAXCH>
__global__ __launch_bounds__(WIDE_NT) void gn_flat_fwd_wide_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   T* __restrict__ y, float* __restrict__ stats, int n, int C, float eps, int silu) {
  __shared__ float sm[WIDE_NT / 64];
  const int b = blockIdx.x, tid = threadIdx.x, nch = n >> 3;
  const T* xs = x + (long)b * n; T* ys = y + (long)b * n;
  typename Chunk8<T>::raw_t raw[MAXCH];
#pragma unroll
  for (int k = 0; k < MAXCH; k++) { const int ci = k * WIDE_NT + tid; if (ci < nch) raw[k] = Chunk8<T>::load(xs + (long)ci * 8); }
  float s[1] = {0.f};
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    if (k * WIDE_NT + tid < nch) {
      float v[8]; Chunk8<T>::unpack(raw[k], v);
      s[0] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
  }
  flat_block_sum<1, WIDE_NT / 64>(s, sm);
#pragma unroll
  for (int k = 0; k < MAXCH; k++) Chunk8<T>::touch(raw[k]);
  const float inv_n = 1.0f / (float)n, mean = s[0] * inv_n;
  float q[1] = {0.f};
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    if (k * WIDE_NT + tid < nch) {
      float v[8]; Chunk8<T>::unpack(raw[k], v);
#pragma unroll
      for (int j = 0; j < 8; j++) { const float d = v[j] - mean; q[0] = fmaf(d, d, q[0]); }
    }
  }
  flat_block_sum<1, WIDE_NT / 64>(q, sm);
#pragma unroll
  for (int k = 0; k < MAXCH; k++) Chunk8<T>::touch(raw[k]);
  const float rstd = rsqrtf(q[0] * inv_n + eps);
  if (tid == 0) { stats[2 * b] = mean; stats[2 * b + 1] = rstd; }
  const int c0 = (tid * 8) % C;
  float ga[8], be[8];
  { const float4 g0 = *(const float4*)(gamma + c0), g1 = *(const float4*)(gamma + c0 + 4), b0 = *(const float4*)(beta + c0), b1 = *(const float4*)(beta + c0 + 4);
    ga[0] = g0.x; ga[1] = g0.y; ga[2] = g0.z; ga[3] = g0.w; ga[4] = g1.x; ga[5] = g1.y; ga[6] = g1.z; ga[7] = g1.w;
    be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w; }
  const float nmr = -mean * rstd;
#pragma unroll
  for (int k = 0; k < MAXCH; k++) {
    const int ci = k * WIDE_NT + tid;
    if (ci < nch) {
      float v[8], o[8]; Chunk8<T>::unpack(raw[k], v);
#pragma unroll
      for (int j = 0; j < 8; j++) { const float z = fmaf(fmaf(v[j], rstd, nmr), ga[j], be[j]); o[j] = silu ? silu_f(z) : z; }
      Chunk8<T>::store(ys + (long)ci * 8, o);
    }
  }
}
bool gn_flat_wide_ok(int L, int C, int G, int resample, long l0, long l1) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_GN_NO_FLAT") != nullptr);
  const long n = (long)L * C;
  return !off && G == 1 && resample == 0 && C >= 16 && C % 8 == 0 && (WIDE_NT * 8) % C == 0 && n % 8 == 0 &&
         n <= (long)WIDE_NT * 8 * 12 && l0 == C && l1 == C;
}
template <typename T>
int gn_flat_fwd_wide(eegldm_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, float* stats, int B, int L, int C, float eps, int silu) {
  const int n = L * C;
  if (n <= WIDE_NT * 8 * 6) hipLaunchKernelGGL((gn_flat_fwd_wide_kernel<T, 6>), dim3(B), dim3(WIDE_NT), 0, ctx->stream, (const T*)x, gamma, beta, (T*)y, stats, n, C, eps, silu);
  else hipLaunchKernelGGL((gn_flat_fwd_wide_kernel<T, 12>), dim3(B), dim3(WIDE_NT), 0, ctx->stream, (const T*)x, gamma, beta, (T*)y, stats, n, C, eps, silu);
  LAUNCH_CHECK();
  return 0;
}

bool gn_flat_ok(int L, int C, int G, int resample, long l0, long l1, long l2, long l3) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_GN_NO_FLAT") != nullptr);
  const long n = (long)L * C;
  return !off && G == 1 && resample == 0 && (C == 1 || C == 2 || C == 4 || C == 8) && n % 8 == 0 && n <= (long)FLAT_NT * 8 * 12 &&
         l0 == C && l1 == C && (l2 == 0 || l2 == C) && (l3 == 0 || l3 == C);
}
template <typename T, int C>
int gn_flat_fwd_c(eegldm_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, float* stats, int B, int n, float eps, int silu) {
  if (n <= FLAT_NT * 8 * 3) hipLaunchKernelGGL((gn_flat_fwd_kernel<T, C, 3>), dim3(B), dim3(FLAT_NT), 0, ctx->stream, (const T*)x, gamma, beta, (T*)y, stats, n, eps, silu);
  else hipLaunchKernelGGL((gn_flat_fwd_kernel<T, C, 12>), dim3(B), dim3(FLAT_NT), 0, ctx->stream, (const T*)x, gamma, beta, (T*)y, stats, n, eps, silu);
  LAUNCH_CHECK();
  return 0;
}
template <typename T>
int gn_flat_fwd(eegldm_ctx* ctx, const void* x, const float* gamma, const float* beta, void* y, float* stats, int B, int L, int C, float eps, int silu) {
  switch (C) {
    case 1: return gn_flat_fwd_c<T, 1>(ctx, x, gamma, beta, y, stats, B, L * C, eps, silu);
    case 2: return gn_flat_fwd_c<T, 2>(ctx, x, gamma, beta, y, stats, B, L * C, eps, silu);
    case 4: return gn_flat_fwd_c<T, 4>(ctx, x, gamma, beta, y, stats, B, L * C, eps, silu);
    default: return gn_flat_fwd_c<T, 8>(ctx, x, gamma, beta, y, stats, B, L * C, eps, silu);
  }
}
template <typename T, int C>
int gn_flat_bwd_c(eegldm_ctx* ctx, const void* x, const float* gamma, const float* beta, const float* stats, const void* dy, void* dx, const void* dxr,
                  float* slots, int B, int n, int silu, int nslot) {
  if (n <= FLAT_NT * 8 * 3) hipLaunchKernelGGL((gn_flat_bwd_kernel<T, C, 3>), dim3(B), dim3(FLAT_NT), 0, ctx->stream, (const T*)x, gamma, beta, stats, (const T*)dy, (T*)dx, (const T*)dxr, slots, n, silu, nslot);
  else hipLaunchKernelGGL((gn_flat_bwd_kernel<T, C, 12>), dim3(B), dim3(FLAT_NT), 0, ctx->stream, (const T*)x, gamma, beta, stats, (const T*)dy, (T*)dx, (const T*)dxr, slots, n, silu, nslot);
  LAUNCH_CHECK();
  return 0;
}
template <typename T>
int gn_flat_bwd(eegldm_ctx* ctx, const void* x, const float* gamma, const float* beta, const float* stats, const void* dy, void* dx, const void* dxr,
                float* dgamma, float* dbeta, int B, int L, int C, int silu) {
  float* slots = dgamma ? (float*)((char*)ctx->scratch + GN_SLOT_OFFSET) : nullptr;
  int nslot = GN_NSLOT;
  if (slots && eeg_deterministic()) {
    nslot = B;
    EEG_TRY(eeg_det_buffer(ctx, (size_t)nslot * 2 * C * sizeof(float), &slots));
    HIP_TRY(hipMemsetAsync(slots, 0, (size_t)nslot * 2 * C * sizeof(float), ctx->stream));
  }
  int rc;
  switch (C) {
    case 1: rc = gn_flat_bwd_c<T, 1>(ctx, x, gamma, beta, stats, dy, dx, dxr, slots, B, L * C, silu, nslot); break;
    case 2: rc = gn_flat_bwd_c<T, 2>(ctx, x, gamma, beta, stats, dy, dx, dxr, slots, B, L * C, silu, nslot); break;
    case 4: rc = gn_flat_bwd_c<T, 4>(ctx, x, gamma, beta, stats, dy, dx, dxr, slots, B, L * C, silu, nslot); break;
    default: rc = gn_flat_bwd_c<T, 8>(ctx, x, gamma, beta, stats, dy, dx, dxr, slots, B, L * C, silu, nslot); break;
  }
  if (rc) return rc;
  if (slots) {
    hipLaunchKernelGGL(gn_slot_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, slots, dgamma, dbeta, C, (double*)ctx->scratch, 0, nslot);
    LAUNCH_CHECK();
  }
  return 0;
}

int gn_check(eegldm_ctx* ctx, int B, int L, int C, int G, int resample, long ldx) {
  EEG_CHECK(B > 0 && L > 0 && C > 0 && G > 0 && C % G == 0, "bad shape B=%d L=%d C=%d G=%d", B, L, C, G);
  EEG_CHECK(G <= MAXG_LDS && C <= MAXG_LDS, "C, G must be <= %d", MAXG_LDS);
  EEG_CHECK(resample >= 0 && resample <= 2, "resample must be 0/1/2");
  EEG_CHECK(resample != 1 || L % 2 == 0, "avgpool needs even L");
  EEG_CHECK((size_t)B * G * 2 * sizeof(double) <= GN_SLOT_OFFSET, "scratch too small for B*G=%d", B * G);
  return 0;
}
bool vec4_ok(int C, int G, long a, long b, long c, long d) {
  return (C % 4 == 0) && ((C / G) % 4 == 0) && a % 4 == 0 && b % 4 == 0 && c % 4 == 0 && d % 4 == 0;
}

}  // namespace

extern "C" int eegldm_groupnorm_fwd(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta,
                                    void* y, long ldy, float* stats, int B, int L, int C, int G, float eps,
                                    int fuse_silu, int resample, void* xr, long ldxr, int dtype) {
  EEG_TRY(gn_check(ctx, B, L, C, G, resample, ldx));
  if (gn_flat_ok(L, C, G, resample, ldx, ldy, 0, 0)) {
    if (dtype == EEGLDM_F32) return gn_flat_fwd<float>(ctx, x, gamma, beta, y, stats, B, L, C, eps, fuse_silu);
    if (dtype == EEGLDM_BF16) return gn_flat_fwd<bf16_t>(ctx, x, gamma, beta, y, stats, B, L, C, eps, fuse_silu);
    if (dtype == EEGLDM_F16) return gn_flat_fwd<f16_t>(ctx, x, gamma, beta, y, stats, B, L, C, eps, fuse_silu);
  }
  if (gn_flat_wide_ok(L, C, G, resample, ldx, ldy) && (dtype != EEGLDM_F32 || (long)L * C <= (long)WIDE_NT * 8 * 6)) {   // fp32: 12 chunks of 8 floats would spill
    if (dtype == EEGLDM_F32) return gn_flat_fwd_wide<float>(ctx, x, gamma, beta, y, stats, B, L, C, eps, fuse_silu);
    if (dtype == EEGLDM_BF16) return gn_flat_fwd_wide<bf16_t>(ctx, x, gamma, beta, y, stats, B, L, C, eps, fuse_silu);
    if (dtype == EEGLDM_F16) return gn_flat_fwd_wide<f16_t>(ctx, x, gamma, beta, y, stats, B, L, C, eps, fuse_silu);
  }
  const bool v4 = vec4_ok(C, G, ldx, ldy, xr ? ldxr : 0, 0);
  if (dtype == EEGLDM_F32) {
    return v4 ? gn_fwd_t<float, 4>(ctx, x, ldx, gamma, beta, y, ldy, stats, B, L, C, G, eps, fuse_silu, resample, xr, ldxr)
              : gn_fwd_t<float, 1>(ctx, x, ldx, gamma, beta, y, ldy, stats, B, L, C, G, eps, fuse_silu, resample, xr, ldxr);
  } else if (dtype == EEGLDM_BF16) {
    return v4 ? gn_fwd_t<bf16_t, 4>(ctx, x, ldx, gamma, beta, y, ldy, stats, B, L, C, G, eps, fuse_silu, resample, xr, ldxr)
              : gn_fwd_t<bf16_t, 1>(ctx, x, ldx, gamma, beta, y, ldy, stats, B, L, C, G, eps, fuse_silu, resample, xr, ldxr);
  } else if (dtype == EEGLDM_F16) {
    return v4 ? gn_fwd_t<f16_t, 4>(ctx, x, ldx, gamma, beta, y, ldy, stats, B, L, C, G, eps, fuse_silu, resample, xr, ldxr)
              : gn_fwd_t<f16_t, 1>(ctx, x, ldx, gamma, beta, y, ldy, stats, B, L, C, G, eps, fuse_silu, resample, xr, ldxr);
  }
  EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "dtype %d", dtype);
}

// colsum_ps (optional): per-sample column sums of dx [B][ldps] fp32; *colsum_done tells the caller whether the
// one-pass kernel produced them (otherwise it runs ew_colsum itself)
int op_groupnorm_bwd(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta,
                     const float* stats, const void* dy, long lddy, void* dx, long lddx,
                     float* dgamma, float* dbeta, int B, int L, int C, int G, int fuse_silu,
                     int resample, const void* dxr, long lddxr, int dtype, float* colsum_ps, long ldps, int* colsum_done,
                     const void* dxr2, long lddxr2, int* dxr2_done, int* slots_deferred, int defer_region) {
  if (dxr2_done) *dxr2_done = 0;
  if (slots_deferred) *slots_deferred = 0;
  EEG_TRY(gn_check(ctx, B, L, C, G, resample, ldx));
  if (gn_flat_ok(L, C, G, resample, ldx, lddy, lddx, dxr ? lddxr : 0)) {
    if (colsum_done) *colsum_done = 0;
    if (dtype == EEGLDM_F32) return gn_flat_bwd<float>(ctx, x, gamma, beta, stats, dy, dx, dxr, dgamma, dbeta, B, L, C, fuse_silu);
    if (dtype == EEGLDM_BF16) return gn_flat_bwd<bf16_t>(ctx, x, gamma, beta, stats, dy, dx, dxr, dgamma, dbeta, B, L, C, fuse_silu);
    if (dtype == EEGLDM_F16) return gn_flat_bwd<f16_t>(ctx, x, gamma, beta, stats, dy, dx, dxr, dgamma, dbeta, B, L, C, fuse_silu);
  }
  const bool v4 = vec4_ok(C, G, ldx, lddy, lddx, dxr ? lddxr : 0);
#define GN_BWD_ARGS ctx, x, ldx, gamma, beta, stats, dy, lddy, dx, lddx, dgamma, dbeta, B, L, C, G, fuse_silu, resample, dxr, lddxr, colsum_ps, ldps, colsum_done, dxr2, lddxr2, dxr2_done, slots_deferred, defer_region
  if (dtype == EEGLDM_F32) return v4 ? gn_bwd_t<float, 4>(GN_BWD_ARGS) : gn_bwd_t<float, 1>(GN_BWD_ARGS);
  if (dtype == EEGLDM_BF16) return v4 ? gn_bwd_t<bf16_t, 4>(GN_BWD_ARGS) : gn_bwd_t<bf16_t, 1>(GN_BWD_ARGS);
  if (dtype == EEGLDM_F16) return v4 ? gn_bwd_t<f16_t, 4>(GN_BWD_ARGS) : gn_bwd_t<f16_t, 1>(GN_BWD_ARGS);
#undef GN_BWD_ARGS
  EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "dtype %d", dtype);
}

// folds the second slot area into dgamma / dbeta (and re-zeroes it): the deferred half of op_groupnorm_bwd(..., slots_deferred)
namespace {
__global__ void gn_slot_reduce_multi_kernel(const eegldm_ctx::GnFoldRec* __restrict__ tab) {
  const eegldm_ctx::GnFoldRec r = tab[blockIdx.y];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * r.C) return;
  float s = 0.f;
  for (int k = 0; k < GN_NSLOT; k++) { s += r.slots[(size_t)k * 2 * r.C + i]; r.slots[(size_t)k * 2 * r.C + i] = 0.f; }
  if (i < r.C) r.dgamma[i] += s; else r.dbeta[i - r.C] += s;
}
}  // namespace
// folds every slot region recorded since the last flush into its dgamma / dbeta (one launch); the device copy of the table is cached
// (the same layers write the same regions step after step) and re-uploaded, after a stream drain, only when it changes
int op_gn_fold_flush(eegldm_ctx* ctx) {
  std::vector<eegldm_ctx::GnFoldRec>& pend = ctx->gn_fold_pending;
  if (pend.empty()) return 0;
  const int first = ctx->gn_fold_count - (int)pend.size();      // table position of this flush's first record
  if (!ctx->gn_fold_dev) HIP_TRY(hipMalloc(&ctx->gn_fold_dev, sizeof(eegldm_ctx::GnFoldRec) * GN_FOLD_MAX));
  if ((int)ctx->gn_fold_host.size() < GN_FOLD_MAX) ctx->gn_fold_host.resize(GN_FOLD_MAX, {nullptr, nullptr, nullptr, 0});
  bool same = true; int maxc = 0;
  for (size_t i = 0; i < pend.size(); i++) {
    const eegldm_ctx::GnFoldRec &a = pend[i], &b = ctx->gn_fold_host[first + i];
    same = same && a.slots == b.slots && a.dgamma == b.dgamma && a.dbeta == b.dbeta && a.C == b.C;
    maxc = a.C > maxc ? a.C : maxc;
  }
  if (!same) {
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(ctx->gn_fold_dev + first, pend.data(), sizeof(eegldm_ctx::GnFoldRec) * pend.size(), hipMemcpyHostToDevice));
    for (size_t i = 0; i < pend.size(); i++) ctx->gn_fold_host[first + i] = pend[i];
  }
  hipLaunchKernelGGL(gn_slot_reduce_multi_kernel, dim3((2 * maxc + 255) / 256, (unsigned)pend.size()), dim3(256), 0, ctx->stream, ctx->gn_fold_dev + first);
  LAUNCH_CHECK();
  pend.clear();
  return 0;
}

int op_gn_slot_reduce_deferred(eegldm_ctx* ctx, float* dgamma, float* dbeta, int C, int region) {
  float* slots = (float*)((char*)ctx->scratch + gn_slot_region(region));
  hipLaunchKernelGGL(gn_slot_reduce_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, ctx->stream, slots, dgamma, dbeta, C, (double*)ctx->scratch, 0);
  LAUNCH_CHECK();
  return 0;
}

extern "C" int eegldm_groupnorm_bwd(eegldm_ctx* ctx, const void* x, long ldx, const float* gamma, const float* beta,
                                    const float* stats, const void* dy, long lddy, void* dx, long lddx,
                                    float* dgamma, float* dbeta, int B, int L, int C, int G, int fuse_silu,
                                    int resample, const void* dxr, long lddxr, int dtype) {
  return op_groupnorm_bwd(ctx, x, ldx, gamma, beta, stats, dy, lddy, dx, lddx, dgamma, dbeta, B, L, C, G, fuse_silu, resample,
                          dxr, lddxr, dtype, nullptr, 0, nullptr);
}

#ifdef EEG_STAGE_TIMING
extern "C" int eegldm_debug_read_gn_tlog(unsigned long long* dst_host, long n_words) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(dst_host, HIP_SYMBOL(gn_tlog), n_words * 8));
  return 0;
}
#endif
