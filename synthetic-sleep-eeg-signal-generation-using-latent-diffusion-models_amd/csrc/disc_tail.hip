// Fused tail of the PatchDiscriminator: BatchNorm1d + LeakyReLU of the last hidden layer and the final 3-tap conv to ONE logit channel
// (MONAI PatchDiscriminator: `Convolution(.., norm=BATCH, act=LEAKYRELU(0.2))` x num_layers_d, then `final_conv` with out_channels = 1;
//  config/config_aekl_eeg.yaml:30-40; twin /root/reference/src/models/discriminator.py:47-84; step body train_autoencoderkl.py:213-228).
//
// Layer-by-layer the tail of one forward + backward moved, per B = 256 pass over y = conv3's output (98 MB in bf16):
//   forward   apply (read y, write a)                 + final conv (read a)                                            = 3 |y|
//   backward  final dgrad (write da) + final wgrad (read a, da) + BatchNorm reduce (read y, da) + apply (read y, da, write dy) = 9 |y|
// The final conv has ONE output channel, so its data gradient is a rank-3 outer product that costs nothing to recompute,
//   da[b, l, c] = dl[b, l + 1] w0[c] + dl[b, l] w1[c] + dl[b, l - 1] w2[c]        (dl = gradient of the logits, zero outside the sample)
// and its forward / weight gradient are per-row dot products / per-channel sums that fit in the passes BatchNorm makes anyway:
//   forward   logits[b, l] = bias + sum_t sum_c w_t[c] a[b, l + t - 1, c],   a = lrelu(gamma (y - mean) rstd + beta)     reads y          = 1 |y|
//   backward  reduce: S1 = sum dz, S2 = sum dz xhat, dW_t[c] = sum_rows a[row, c] dl[row - t + 1], dbias = sum dl        reads y          = 1 |y|
//             apply:  dy = gamma rstd (dz - S1 / N - xhat S2 / N),   dz = lrelu'(z) da                                  reads y, writes dy = 2 |y|
// `a` and `da` never exist.  One wave covers whole rows (G lanes per row, 16 bytes per lane and chunk); per-row dot products are reduced
// with DPP adds inside 16 lanes and two cross-row shuffles; per-channel sums are kept in registers over the rows a wave walks.
#include "common.h"
#include "internal.h"

int ls_bn_fold(eegldm_ctx* ctx, const void* parts, int nb, int nvals, double** sums_out);      // losses.hip

namespace {
constexpr int NT = 256;
constexpr int MAXR = 128;      // output rows per block of the forward kernel (+ 2 halo rows)

template <typename T> struct Ch { static constexpr int E = 8; };
template <> struct Ch<float> { static constexpr int E = 4; };

template <typename T> __device__ __forceinline__ void unpack16(const uint4& v, float* o) {
  if constexpr (sizeof(T) == 4) { o[0] = __uint_as_float(v.x); o[1] = __uint_as_float(v.y); o[2] = __uint_as_float(v.z); o[3] = __uint_as_float(v.w); }
  else {
    o[0] = w16_lo<T>(v.x); o[1] = w16_hi<T>(v.x); o[2] = w16_lo<T>(v.y); o[3] = w16_hi<T>(v.y);
    o[4] = w16_lo<T>(v.z); o[5] = w16_hi<T>(v.z); o[6] = w16_lo<T>(v.w); o[7] = w16_hi<T>(v.w);
  }
}
template <typename T> __device__ __forceinline__ uint4 pack16(const float* o) {
  uint4 v;
  if constexpr (sizeof(T) == 4) { v.x = __float_as_uint(o[0]); v.y = __float_as_uint(o[1]); v.z = __float_as_uint(o[2]); v.w = __float_as_uint(o[3]); }
  else { v.x = pack16x2<T>(o[0], o[1]); v.y = pack16x2<T>(o[2], o[3]); v.z = pack16x2<T>(o[4], o[5]); v.w = pack16x2<T>(o[6], o[7]); }
  return v;
}
// sum over the G lanes (G a power of two, 4 .. 64) that share a row; every lane of the group ends up with the total
template <int G> __device__ __forceinline__ float group_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  if constexpr (G >= 16) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  if constexpr (G >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

struct TailArgs {
  const void* y; long ldy;
  const float* gamma; const float* beta; const float* stats;      // BatchNorm: [C], [C], [C][2] = (mean, rstd)
  const float* w; const float* bias;                              // final conv, fp32 master weights [3][C]; bias [1] or null
  float slope;
  int B, L, C;
};

// per-lane constants of one 16-byte chunk (E channels starting at c0)
template <int E> struct ChunkK { float sc[E], sh[E], w0[E], w1[E], w2[E]; };
template <int E> __device__ __forceinline__ void load_k(const TailArgs& p, int c0, ChunkK<E>& k) {
#pragma unroll
  for (int e = 0; e < E; e++) {
    const int c = c0 + e;
    const float mean = p.stats[2 * c], rstd = p.stats[2 * c + 1];
    k.sc[e] = p.gamma[c] * rstd; k.sh[e] = p.beta[c] - mean * k.sc[e];
    k.w0[e] = p.w[c]; k.w1[e] = p.w[p.C + c]; k.w2[e] = p.w[2 * p.C + c];
  }
}

// ---------------------------------------------------------------- forward: logits (fp32, [B][L]) from y
// block = RBLK output rows of one sample; the RBLK + 2 rows l0 - 1 .. l0 + RBLK each give three partial logits p_t = <w_t, a[row]>, which
// land in LDS; logits[l] = bias + p_0[l - 1] + p_1[l] + p_2[l + 1]
template <typename T, int G, int NJ>
__global__ __launch_bounds__(NT) void tail_fwd_kernel(const TailArgs p, float* __restrict__ logits, int RBLK) {
  constexpr int E = Ch<T>::E, RPW = 64 / G;
  __shared__ float ps[MAXR + 2][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane % G, rg = lane / G;
  const int b = blockIdx.y, l0 = blockIdx.x * RBLK, nrow = min(RBLK, p.L - l0) + 2;
  ChunkK<E> k[NJ];
#pragma unroll
  for (int j = 0; j < NJ; j++) load_k<E>(p, (g + G * j) * E, k[j]);
  const char* yb = (const char*)p.y + (size_t)b * p.L * p.ldy * sizeof(T);
  for (int rb = wave * RPW; rb < nrow; rb += 4 * RPW) {
    const int r = rb + rg, l = l0 - 1 + r;
    const bool ok = r < nrow && l >= 0 && l < p.L;
    float p0 = 0.f, p1 = 0.f, p2 = 0.f;
    if (ok) {
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const uint4 v = *(const uint4*)(yb + ((size_t)l * p.ldy + (g + G * j) * E) * sizeof(T));
        float x[E]; unpack16<T>(v, x);
#pragma unroll
        for (int e = 0; e < E; e++) {
          const float z = x[e] * k[j].sc[e] + k[j].sh[e];
          const float a = z > 0.f ? z : p.slope * z;
          p0 = fmaf(a, k[j].w0[e], p0); p1 = fmaf(a, k[j].w1[e], p1); p2 = fmaf(a, k[j].w2[e], p2);
        }
      }
    }
    p0 = group_sum<G>(p0); p1 = group_sum<G>(p1); p2 = group_sum<G>(p2);
    if (g == 0 && r < nrow) { ps[r][0] = p0; ps[r][1] = p1; ps[r][2] = p2; }
  }
  __syncthreads();
  const float bias = p.bias ? p.bias[0] : 0.f;
  for (int i = tid; i < nrow - 2; i += NT)      // output row l0 + i = LDS row i + 1: tap t reads a[l + t - 1] = LDS row i + t
    logits[(size_t)b * p.L + l0 + i] = bias + ps[i][0] + ps[i + 1][1] + ps[i + 2][2];
}

// ---------------------------------------------------------------- backward, shared per-row arithmetic
struct RowDl { float dn, d0, dp; };      // dl[l + 1], dl[l], dl[l - 1] of this row (zero outside the sample)
// l = row % L (kept by the caller: no division per row).  Branch-free: every load is unconditional on a clamped index and zeroed by a select, so
// that the U rows a lane requests per iteration are all in flight together (a load under a branch is a basic block with its own wait)
__device__ __forceinline__ RowDl load_dl(const float* __restrict__ dl, long row, int l, int L, long nrows, bool ok) {
  const long rc = row < nrows ? row : nrows - 1;
  const float d0 = dl[rc], dn = dl[rc + 1 < nrows ? rc + 1 : rc], dp = dl[rc > 0 ? rc - 1 : 0];
  RowDl r; r.d0 = ok ? d0 : 0.f; r.dn = (ok && l + 1 < L) ? dn : 0.f; r.dp = (ok && l > 0) ? dp : 0.f;
  return r;
}

// sums layout (planar): S1 [0, C), S2 [C, 2C), dW0 [2C, 3C), dW1 [3C, 4C), dW2 [4C, 5C), dbias [5C]
template <typename T, int G, int NJ, bool PG>
__global__ __launch_bounds__(NT) void tail_bwd_reduce_kernel(const TailArgs p, const float* __restrict__ dl, float* __restrict__ parts, long rows_per_block) {
  constexpr int E = Ch<T>::E, RPW = 64 / G, NV = PG ? 5 : 2;
  extern __shared__ double red[];      // [5 C + 1] (fp64 LDS atomics: order-independent)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane % G, rg = lane / G;
  const int nvals = 5 * p.C + 1;
  for (int i = tid; i < nvals; i += NT) red[i] = 0.0;
  __syncthreads();
  const long rows = (long)p.B * p.L, r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float sc[NJ][E], sh[NJ][E], mean[NJ][E], rstd[NJ][E], w0[NJ][E], w1[NJ][E], w2[NJ][E];
  float acc[NJ][E][NV];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int c = (g + G * j) * E + e;
      mean[j][e] = p.stats[2 * c]; rstd[j][e] = p.stats[2 * c + 1];
      sc[j][e] = p.gamma[c] * rstd[j][e]; sh[j][e] = p.beta[c] - mean[j][e] * sc[j][e];
      w0[j][e] = p.w[c]; w1[j][e] = p.w[p.C + c]; w2[j][e] = p.w[2 * p.C + c];
#pragma unroll
      for (int v = 0; v < NV; v++) acc[j][e][v] = 0.f;
    }
  float sdl = 0.f;
  // (U rows of a lane in flight per iteration: measured 1 -> 4 on this kernel, 28 -> 29 us without and 44 -> 57 us with the weight-gradient
  // sums -- it is VALU-bound, 12 / 17 operations per element, and the extra live registers cost occupancy; the first layer's kernel below,
  // 9 operations per element, went 39 -> 21 us with U = 4)
  constexpr int U = 1;
  constexpr int STEP = 4 * RPW;
  int l = (int)((r0 + wave * RPW + rg) % p.L);
  for (long row = r0 + wave * RPW + rg; row < r1; row += STEP * U) {
    uint4 v[U][NJ]; RowDl d[U];
    int lu = l;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long rr = row + (long)u * STEP; const bool ok = rr < r1; const long rc = ok ? rr : r1 - 1;
      d[u] = load_dl(dl, rr, lu, p.L, rows, ok);
#pragma unroll
      for (int j = 0; j < NJ; j++) v[u][j] = *(const uint4*)((const char*)p.y + ((size_t)rc * p.ldy + (g + G * j) * E) * sizeof(T));
      lu += STEP; while (lu >= p.L) lu -= p.L;
    }
    l = lu;
#pragma unroll
    for (int u = 0; u < U; u++) {
      if (g == 0) sdl += d[u].d0;
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        float x[E]; unpack16<T>(v[u][j], x);
#pragma unroll
        for (int e = 0; e < E; e++) {
          const float z = x[e] * sc[j][e] + sh[j][e];
          const float xh = (x[e] - mean[j][e]) * rstd[j][e];
          const float da = fmaf(d[u].dn, w0[j][e], fmaf(d[u].d0, w1[j][e], d[u].dp * w2[j][e]));
          const float dz = z > 0.f ? da : p.slope * da;
          acc[j][e][0] += dz; acc[j][e][1] = fmaf(dz, xh, acc[j][e][1]);
          if constexpr (PG) {
            const float a = z > 0.f ? z : p.slope * z;
            acc[j][e][2] = fmaf(a, d[u].dn, acc[j][e][2]); acc[j][e][3] = fmaf(a, d[u].d0, acc[j][e][3]); acc[j][e][4] = fmaf(a, d[u].dp, acc[j][e][4]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int c = (g + G * j) * E + e;
#pragma unroll
      for (int v = 0; v < NV; v++) atomicAdd(&red[v * p.C + c], (double)acc[j][e][v]);
    }
  if (PG && g == 0) atomicAdd(&red[5 * p.C], (double)sdl);
  __syncthreads();
  float* part = parts + (size_t)blockIdx.x * nvals;
  for (int i = tid; i < nvals; i += NT) part[i] = (float)red[i];
}

// dy = gamma rstd (dz - S1 / N - xhat S2 / N); block 0 also adds the parameter gradients from the folded sums
template <typename T, int G, int NJ>
__global__ __launch_bounds__(NT) void tail_bwd_apply_kernel(const TailArgs p, const float* __restrict__ dl, const double* __restrict__ sums,
                                                            T* __restrict__ dy, long lddy, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            float* __restrict__ dw, float* __restrict__ dbias, long rows_per_block) {
  constexpr int E = Ch<T>::E, RPW = 64 / G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane % G, rg = lane / G;
  const long rows = (long)p.B * p.L, r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  if (blockIdx.x == 0 && dgamma) {
    for (int c = tid; c < p.C; c += NT) {
      atomicAdd(&dbeta[c], (float)sums[c]); atomicAdd(&dgamma[c], (float)sums[p.C + c]);
      if (dw) { atomicAdd(&dw[c], (float)sums[2 * p.C + c]); atomicAdd(&dw[p.C + c], (float)sums[3 * p.C + c]); atomicAdd(&dw[2 * p.C + c], (float)sums[4 * p.C + c]); }
    }
    if (tid == 0 && dbias) atomicAdd(dbias, (float)sums[5 * p.C]);
  }
  const float inv_n = 1.0f / (float)rows;
  float sc[NJ][E], sh[NJ][E], mean[NJ][E], rstd[NJ][E], w0[NJ][E], w1[NJ][E], w2[NJ][E], k1[NJ][E], k2[NJ][E];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int c = (g + G * j) * E + e;
      mean[j][e] = p.stats[2 * c]; rstd[j][e] = p.stats[2 * c + 1];
      sc[j][e] = p.gamma[c] * rstd[j][e]; sh[j][e] = p.beta[c] - mean[j][e] * sc[j][e];
      w0[j][e] = p.w[c]; w1[j][e] = p.w[p.C + c]; w2[j][e] = p.w[2 * p.C + c];
      k1[j][e] = (float)sums[c] * inv_n; k2[j][e] = (float)sums[p.C + c] * inv_n;
    }
  int l = (int)((r0 + wave * RPW + rg) % p.L);
  for (long row = r0 + wave * RPW + rg; row < r1; row += 4 * RPW) {
    const RowDl d = load_dl(dl, row, l, p.L, rows, true);
    l += 4 * RPW; while (l >= p.L) l -= p.L;
#pragma unroll
    for (int j = 0; j < NJ; j++) {
      const uint4 v = *(const uint4*)((const char*)p.y + ((size_t)row * p.ldy + (g + G * j) * E) * sizeof(T));
      float x[E], o[E]; unpack16<T>(v, x);
#pragma unroll
      for (int e = 0; e < E; e++) {
        const float z = x[e] * sc[j][e] + sh[j][e];
        const float xh = (x[e] - mean[j][e]) * rstd[j][e];
        const float da = fmaf(d.dn, w0[j][e], fmaf(d.d0, w1[j][e], d.dp * w2[j][e]));
        const float dz = z > 0.f ? da : p.slope * da;
        o[e] = sc[j][e] * (dz - k1[j][e] - xh * k2[j][e]);
      }
      *(uint4*)((char*)dy + ((size_t)row * lddy + (g + G * j) * E) * sizeof(T)) = pack16<T>(o);
    }
  }
}

// ================================================================ fused HEAD backward: LeakyReLU + the first conv (1 -> C0 channels, k 3)
// Layer by layer the first layer's backward was: LeakyReLU backward (read a0, da0, write dy0), conv0 weight gradient (read dy0), bias column
// sums (read dy0), and -- generator pass -- conv0 data gradient (read dy0) + an NLC -> NCL pass: 4-5 |a0| of traffic for an input with ONE
// channel.  Everything the layer needs besides da0 can be recomputed from the window itself: the pre-activation z = b + w0 x[sl-1] + w1 x[sl]
// + w2 x[sl+1] (three FMAs per element, in the forward kernel's own order, so the mask is the forward's), hence
//   dy0 = da0 (z > 0 ? 1 : slope),   dW_t[c] = sum dy0[b, l, c] x[b, s l + t - 1],   db[c] = sum dy0,   dx[b, i] = sum_{t, l: s l + t - 1 = i} <dy0[b, l, :], w_t>
// in ONE pass over da0 each (parameter gradients: the discriminator passes; dx: the generator pass).
struct HeadArgs {
  const void* da; long ldda;          // gradient of the activated first-layer output, [B * Lo][C0]
  const void* x;                      // the layer's input as the forward saw it: [B * L] elements of the engine dtype (one channel, ld 1)
  const void* w; const float* bias;   // [3][C0][1] in the engine dtype (the copy the forward conv read), bias fp32 [C0] or null
  float slope;
  int B, L, Lo, C0, stride;           // input / output positions per sample; pad_l = 1
};
template <typename T> __device__ __forceinline__ float head_x(const HeadArgs& p, long base, int i) {      // x[b][i], zero outside the sample (branch-free: clamped load + select)
  const int ic = i < 0 ? 0 : (i >= p.L ? p.L - 1 : i);
  const float v = ld_f32((const T*)p.x + base + ic);
  return (i >= 0 && i < p.L) ? v : 0.f;
}

// parameter gradients: partial sums per block, planar [dW0 | dW1 | dW2 | db] (4 C0 values)
template <typename T, int G, int NJ>
__global__ __launch_bounds__(NT) void head_bwd_pg_kernel(const HeadArgs p, float* __restrict__ parts, long rows_per_block) {
  constexpr int E = Ch<T>::E, RPW = 64 / G;
  extern __shared__ double red[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane % G, rg = lane / G;
  const int nvals = 4 * p.C0;
  for (int i = tid; i < nvals; i += NT) red[i] = 0.0;
  __syncthreads();
  const long rows = (long)p.B * p.Lo, r0 = (long)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  float w0[NJ][E], w1[NJ][E], w2[NJ][E], bs[NJ][E], acc[NJ][E][4];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int c = (g + G * j) * E + e;
      w0[j][e] = ld_f32((const T*)p.w + c); w1[j][e] = ld_f32((const T*)p.w + p.C0 + c); w2[j][e] = ld_f32((const T*)p.w + 2 * p.C0 + c);
      bs[j][e] = p.bias ? p.bias[c] : 0.f;
#pragma unroll
      for (int v = 0; v < 4; v++) acc[j][e][v] = 0.f;
    }
  // U rows per lane in flight (branch-free loads on clamped addresses); rows past the range carry a zero gradient chunk and contribute nothing
  constexpr int U = 4;
  constexpr int STEP = 4 * RPW;
  long row = r0 + wave * RPW + rg;
  int b = (int)(row / p.Lo), l = (int)(row - (long)b * p.Lo);
  for (; row < r1; row += STEP * U) {
    uint4 v[U][NJ]; float xm[U], x0[U], xp[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long rr = row + (long)u * STEP; const bool ok = rr < r1; const long rc = ok ? rr : r1 - 1;
      const int bc = b < p.B ? b : p.B - 1;
      const long xb = (long)bc * p.L; const int i0 = l * p.stride - 1;
      xm[u] = head_x<T>(p, xb, i0); x0[u] = head_x<T>(p, xb, i0 + 1); xp[u] = head_x<T>(p, xb, i0 + 2);
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const uint4 t = *(const uint4*)((const char*)p.da + ((size_t)rc * p.ldda + (g + G * j) * E) * sizeof(T));
        v[u][j] = ok ? t : make_uint4(0u, 0u, 0u, 0u);
      }
      l += STEP; while (l >= p.Lo) { l -= p.Lo; b++; }
    }
#pragma unroll
    for (int u = 0; u < U; u++)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        float d[E]; unpack16<T>(v[u][j], d);
#pragma unroll
        for (int e = 0; e < E; e++) {
          const float z = fmaf(xp[u], w2[j][e], fmaf(x0[u], w1[j][e], fmaf(xm[u], w0[j][e], bs[j][e])));
          const float dy = z > 0.f ? d[e] : p.slope * d[e];
          acc[j][e][0] = fmaf(dy, xm[u], acc[j][e][0]); acc[j][e][1] = fmaf(dy, x0[u], acc[j][e][1]); acc[j][e][2] = fmaf(dy, xp[u], acc[j][e][2]);
          acc[j][e][3] += dy;
        }
      }
  }
  // the 64 / G row groups of a wave hold partial sums of the SAME channels: add them with shuffles first (lanes g, g + G, ...), so that only
  // one lane per channel and wave reaches the LDS accumulator (with G = 8 the direct form was 32 lanes serialising on every address)
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int c = (g + G * j) * E + e;
#pragma unroll
      for (int v = 0; v < 4; v++) {
        float a = acc[j][e][v];
#pragma unroll
        for (int d = G; d < 64; d <<= 1) a += __shfl_xor(a, d, 64);
        if (rg == 0) atomicAdd(&red[v * p.C0 + c], (double)a);
      }
    }
  __syncthreads();
  float* part = parts + (size_t)blockIdx.x * nvals;
  for (int i = tid; i < nvals; i += NT) part[i] = (float)red[i];
}
__global__ void head_finish_kernel(const double* __restrict__ sums, float* __restrict__ dw, float* __restrict__ db, int C0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 3 * C0) dw[i] += (float)sums[i];
  else if (i < 4 * C0 && db) db[i - 3 * C0] += (float)sums[i];
}

// input gradient, fp32 [B][L] (NCL with one channel): block = RBLK output rows l0 .. of one sample (+ one halo row); row l gives the three dot
// products q_t[l] = <dy0[l, :], w_t>, and dx[s l + t - 1] collects q_t[l].  stride 2: dx[2l] = q_1[l], dx[2l + 1] = q_2[l] + q_0[l + 1];
// stride 1: dx[l] = q_0[l + 1] + q_1[l] + q_2[l - 1]  (two halo rows).
template <typename T, int G, int NJ>
__global__ __launch_bounds__(NT) void head_bwd_dx_kernel(const HeadArgs p, float* __restrict__ dx, int RBLK) {
  constexpr int E = Ch<T>::E, RPW = 64 / G;
  __shared__ float qs[MAXR + 2][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane % G, rg = lane / G;
  const int b = blockIdx.y, l0 = blockIdx.x * RBLK, nout = min(RBLK, p.Lo - l0), nrow = nout + 2;      // LDS row r <-> output row l0 - 1 + r
  float w0[NJ][E], w1[NJ][E], w2[NJ][E], bs[NJ][E];
#pragma unroll
  for (int j = 0; j < NJ; j++)
#pragma unroll
    for (int e = 0; e < E; e++) {
      const int c = (g + G * j) * E + e;
      w0[j][e] = ld_f32((const T*)p.w + c); w1[j][e] = ld_f32((const T*)p.w + p.C0 + c); w2[j][e] = ld_f32((const T*)p.w + 2 * p.C0 + c);
      bs[j][e] = p.bias ? p.bias[c] : 0.f;
    }
  const long xb = (long)b * p.L;
  for (int rb = wave * RPW; rb < nrow; rb += 4 * RPW) {
    const int r = rb + rg, l = l0 - 1 + r;
    const bool ok = r < nrow && l >= 0 && l < p.Lo;
    const int lc = l < 0 ? 0 : (l >= p.Lo ? p.Lo - 1 : l);      // (branch-free loads: a row outside the sample reads a clamped one and is zeroed)
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
    {
      const int i0 = lc * p.stride - 1;
      const float xm = head_x<T>(p, xb, i0), x0 = head_x<T>(p, xb, i0 + 1), xp = head_x<T>(p, xb, i0 + 2);
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const uint4 t = *(const uint4*)((const char*)p.da + (((size_t)b * p.Lo + lc) * p.ldda + (g + G * j) * E) * sizeof(T));
        const uint4 v = ok ? t : make_uint4(0u, 0u, 0u, 0u);
        float d[E]; unpack16<T>(v, d);
#pragma unroll
        for (int e = 0; e < E; e++) {
          const float z = fmaf(xp, w2[j][e], fmaf(x0, w1[j][e], fmaf(xm, w0[j][e], bs[j][e])));
          const float dy = z > 0.f ? d[e] : p.slope * d[e];
          q0 = fmaf(dy, w0[j][e], q0); q1 = fmaf(dy, w1[j][e], q1); q2 = fmaf(dy, w2[j][e], q2);
        }
      }
    }
    q0 = group_sum<G>(q0); q1 = group_sum<G>(q1); q2 = group_sum<G>(q2);
    if (g == 0 && r < nrow) { qs[r][0] = q0; qs[r][1] = q1; qs[r][2] = q2; }
  }
  __syncthreads();
  float* dxb = dx + xb;
  if (p.stride == 2) {
    for (int i = tid; i < nout; i += NT) {      // output row l = l0 + i = LDS row i + 1
      const int l = l0 + i;
      if (2 * l < p.L) dxb[2 * l] = qs[i + 1][1];
      if (2 * l + 1 < p.L) dxb[2 * l + 1] = qs[i + 1][2] + qs[i + 2][0];      // (row l + 1 beyond the sample: its q is zero)
    }
    // the position in front of the sample's first output row (index -1) does not exist; odd lengths end on an even index (handled above)
  } else {
    for (int i = tid; i < nout; i += NT) dxb[l0 + i] = qs[i + 2][0] + qs[i + 1][1] + qs[i][2];
  }
}

// lanes per row / chunks per lane for a channel count: C / E chunks per row, at most 64 lanes per row
inline bool tail_shape(int C, int E, int* G, int* NJ) {
  if (C % E != 0) return false;
  const int ch = C / E;
  if (ch >= 64) { if (ch % 64 != 0 || ch / 64 > 2) return false; *G = 64; *NJ = ch / 64; return true; }
  if (ch != 4 && ch != 8 && ch != 16 && ch != 32) return false;
  *G = ch; *NJ = 1;
  return true;
}
}  // namespace

bool disc_tail_ok(int dtype, int C, long ldy) {
  int G, NJ;
  const int E = dtype == EEGLDM_F32 ? 4 : 8;
  return tail_shape(C, E, &G, &NJ) && ldy % E == 0 && C <= 1024;
}

#define TAIL_DISPATCH(dtype, G, NJ, ...)                                                                     \
  do {                                                                                                         \
    if ((dtype) == EEGLDM_F32) { typedef float T; TAIL_G(G, NJ, __VA_ARGS__); }                                       \
    else if ((dtype) == EEGLDM_BF16) { typedef bf16_t T; TAIL_G(G, NJ, __VA_ARGS__); }                                \
    else { typedef f16_t T; TAIL_G(G, NJ, __VA_ARGS__); }                                                             \
  } while (0)
#define TAIL_G(G, NJ, ...)                                                                                     \
  do {                                                                                                         \
    if (G == 64 && NJ == 1) { constexpr int G_ = 64, NJ_ = 1; __VA_ARGS__; }                                          \
    else if (G == 64 && NJ == 2) { constexpr int G_ = 64, NJ_ = 2; __VA_ARGS__; }                                     \
    else if (G == 32) { constexpr int G_ = 32, NJ_ = 1; __VA_ARGS__; }                                                \
    else if (G == 16) { constexpr int G_ = 16, NJ_ = 1; __VA_ARGS__; }                                                \
    else if (G == 8) { constexpr int G_ = 8, NJ_ = 1; __VA_ARGS__; }                                                  \
    else { constexpr int G_ = 4, NJ_ = 1; __VA_ARGS__; }                                                              \
  } while (0)

// logits: fp32 [B][L] (= NCL with one channel).  stats as left by ls_bn_stats.
int disc_tail_fwd(eegldm_ctx* ctx, int dtype, const void* y, long ldy, const float* gamma, const float* beta, const float* stats, const float* w3,
                  const float* bias, float slope, float* logits, int B, int L, int C) {
  int G, NJ;
  EEG_CHECK(tail_shape(C, dtype == EEGLDM_F32 ? 4 : 8, &G, &NJ), "fused discriminator tail: unsupported channel count %d", C);
  TailArgs p = {y, ldy, gamma, beta, stats, w3, bias, slope, B, L, C};
  // rows per block: enough blocks to keep ~12 per CU in flight, 2 halo rows per block
  long want = ((long)B * L + (long)ctx->num_cu * 12 - 1) / ((long)ctx->num_cu * 12);
  int rblk = (int)(want < 16 ? 16 : (want > MAXR ? MAXR : want));
  if (rblk > L) rblk = L;
  const dim3 grid((L + rblk - 1) / rblk, B);
  TAIL_DISPATCH(dtype, G, NJ, hipLaunchKernelGGL((tail_fwd_kernel<T, G_, NJ_>), grid, dim3(NT), 0, ctx->stream, p, logits, rblk));
  LAUNCH_CHECK();
  return 0;
}

// dlogits: fp32 [B][L].  dy: [B * L][lddy] in `dtype`.  dgamma / dbeta / dw3 ([3][C]) / dbias are ACCUMULATED; all four NULL = no parameter gradients.
int disc_tail_bwd(eegldm_ctx* ctx, int dtype, const void* y, long ldy, const float* gamma, const float* beta, const float* stats, const float* w3,
                  float slope, const float* dlogits, void* dy, long lddy, float* dgamma, float* dbeta, float* dw3, float* dbias, int B, int L, int C) {
  int G, NJ;
  EEG_CHECK(tail_shape(C, dtype == EEGLDM_F32 ? 4 : 8, &G, &NJ), "fused discriminator tail: unsupported channel count %d", C);
  TailArgs p = {y, ldy, gamma, beta, stats, w3, nullptr, slope, B, L, C};
  const long rows = (long)B * L;
  const int nvals = 5 * C + 1;
  long nb = (long)ctx->num_cu * 4; if (nb > (rows + 63) / 64) nb = (rows + 63) / 64;
  while ((size_t)nb * nvals * sizeof(float) > (16u << 20)) nb /= 2;
  if (nb < 1) nb = 1;
  const long rpb = (rows + nb - 1) / nb; nb = (rows + rpb - 1) / rpb;
  float* parts = (float*)((char*)ctx->scratch + (8u << 20));
  const size_t sh = (size_t)nvals * sizeof(double);
  const bool pg = dgamma != nullptr;
  if (pg) TAIL_DISPATCH(dtype, G, NJ, hipLaunchKernelGGL((tail_bwd_reduce_kernel<T, G_, NJ_, true>), dim3((unsigned)nb), dim3(NT), sh, ctx->stream, p, dlogits, parts, rpb));
  else TAIL_DISPATCH(dtype, G, NJ, hipLaunchKernelGGL((tail_bwd_reduce_kernel<T, G_, NJ_, false>), dim3((unsigned)nb), dim3(NT), sh, ctx->stream, p, dlogits, parts, rpb));
  LAUNCH_CHECK();
  double* sums;
  EEG_TRY(ls_bn_fold(ctx, parts, (int)nb, nvals, &sums));
  long nba = (long)ctx->num_cu * 16; if (nba > (rows + 15) / 16) nba = (rows + 15) / 16;
  if (nba < 1) nba = 1;
  const long rpa = (rows + nba - 1) / nba; nba = (rows + rpa - 1) / rpa;
  TAIL_DISPATCH(dtype, G, NJ, hipLaunchKernelGGL((tail_bwd_apply_kernel<T, G_, NJ_>), dim3((unsigned)nba), dim3(NT), 0, ctx->stream, p, dlogits, sums, (T*)dy, lddy,
                                                 dgamma, dbeta, dw3, dbias, rpa));
  LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------- fused head (first layer) backward
bool disc_head_ok(int dtype, int C0, long ldda, int stride, int L, int Lo) {
  int G, NJ;
  const int E = dtype == EEGLDM_F32 ? 4 : 8;
  if (!tail_shape(C0, E, &G, &NJ) || ldda % E != 0 || C0 > 1024) return false;
  if (stride == 2) return Lo == (L + 2 - 3) / 2 + 1 && L == 2 * Lo;      // every input position 0 .. 2 Lo - 1 is written by exactly one output row
  return stride == 1 && Lo == L;
}
// da: [B * Lo][ldda] gradient of the ACTIVATED first-layer output; x: [B * L] input elements (engine dtype); w: the engine-dtype weight copy
// [3][C0]; dw / db (fp32, ACCUMULATED, both NULL = none); dx (fp32 [B][L], WRITTEN, NULL = none)
int disc_head_bwd(eegldm_ctx* ctx, int dtype, const void* da, long ldda, const void* x, const void* w, const float* bias, float slope,
                  float* dw, float* db, float* dx, int B, int L, int Lo, int C0, int stride) {
  int G, NJ;
  EEG_CHECK(tail_shape(C0, dtype == EEGLDM_F32 ? 4 : 8, &G, &NJ), "fused discriminator head: unsupported channel count %d", C0);
  HeadArgs p = {da, ldda, x, w, bias, slope, B, L, Lo, C0, stride};
  const long rows = (long)B * Lo;
  if (dw) {
    const int nvals = 4 * C0;
    long nb = (long)ctx->num_cu * 4; if (nb > (rows + 63) / 64) nb = (rows + 63) / 64;
    if (nb < 1) nb = 1;
    const long rpb = (rows + nb - 1) / nb; nb = (rows + rpb - 1) / rpb;
    float* parts = (float*)((char*)ctx->scratch + (8u << 20));
    EEG_CHECK((size_t)nb * nvals * sizeof(float) <= (16u << 20), "partials area too small");
    TAIL_DISPATCH(dtype, G, NJ, hipLaunchKernelGGL((head_bwd_pg_kernel<T, G_, NJ_>), dim3((unsigned)nb), dim3(NT), (size_t)nvals * sizeof(double), ctx->stream, p, parts, rpb));
    LAUNCH_CHECK();
    double* sums;
    EEG_TRY(ls_bn_fold(ctx, parts, (int)nb, nvals, &sums));
    hipLaunchKernelGGL(head_finish_kernel, dim3((nvals + 255) / 256), dim3(256), 0, ctx->stream, sums, dw, db, C0);
    LAUNCH_CHECK();
  }
  if (dx) {
    long want = (rows + (long)ctx->num_cu * 12 - 1) / ((long)ctx->num_cu * 12);
    int rblk = (int)(want < 16 ? 16 : (want > MAXR ? MAXR : want));
    if (rblk > Lo) rblk = Lo;
    const dim3 grid((Lo + rblk - 1) / rblk, B);
    TAIL_DISPATCH(dtype, G, NJ, hipLaunchKernelGGL((head_bwd_dx_kernel<T, G_, NJ_>), grid, dim3(NT), 0, ctx->stream, p, dx, rblk));
    LAUNCH_CHECK();
  }
  return 0;
}
