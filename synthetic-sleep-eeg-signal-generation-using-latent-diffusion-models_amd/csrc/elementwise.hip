// HBM-bound elementwise / small-reduction kernels of the hot path: layout conversion,
// weight packing, timestep embedding, SiLU, per-sample column sums, softmax, scheduler
// arithmetic, MSE, Adam, Philox RNG.  One pass over the data each, 16-byte accesses where
// the layout allows.  Reference call sites are cited at each entry point.
#include "common.h"
#include "internal.h"

namespace {
constexpr int NT = 256;

inline int grid1d(long n, eegldm_ctx* ctx, int per_thread = 1) {
  long blocks = (n + (long)NT * per_thread - 1) / ((long)NT * per_thread);
  long cap = (long)ctx->num_cu * 16;
  if (blocks < 1) blocks = 1;
  return (int)(blocks < cap ? blocks : cap);
}
#define GRID_STRIDE(i, n) for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

// ------------------------------------------------------------------ layout
// (B,C,L) fp32 -> rows (b,l) x C in T.  One block handles a 64(l) x 64(c) tile via LDS so both
// sides are coalesced; for tiny C (1..4) the tile degenerates gracefully.
template <typename T>
__global__ __launch_bounds__(NT) void ncl_to_nlc_kernel(const float* __restrict__ src, T* __restrict__ dst, long ld, int C, int L) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, l0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += NT) {
    const int cc = i / 64, ll = i % 64;
    if (c0 + cc < C && l0 + ll < L) tile[cc][ll] = src[((long)b * C + c0 + cc) * L + l0 + ll];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += NT) {
    const int ll = i / 64, cc = i % 64;
    if (c0 + cc < C && l0 + ll < L) st_f32(dst + ((long)b * L + l0 + ll) * ld + c0 + cc, tile[cc][ll]);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void nlc_to_ncl_kernel(const T* __restrict__ src, long ld, float* __restrict__ dst, int C, int L) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z, l0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += NT) {
    const int ll = i / 64, cc = i % 64;
    if (c0 + cc < C && l0 + ll < L) tile[cc][ll] = ld_f32(src + ((long)b * L + l0 + ll) * ld + c0 + cc);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += NT) {
    const int cc = i / 64, ll = i % 64;
    if (c0 + cc < C && l0 + ll < L) dst[((long)b * C + c0 + cc) * L + l0 + ll] = tile[cc][ll];
  }
}
// Few channels (C <= 8: the 1-channel windows, logits and latents at the models' ends): the 64 x 64 tile above would run 4096 iterations
// per block for 64 * C elements (24 us for the 3 MB of a (256, 1, 3072) batch).  Here a thread owns one position (b, l) and walks its C
// channels: reads are coalesced along l for every channel, writes are C consecutive elements per thread.
template <typename T>
__global__ __launch_bounds__(NT) void ncl_to_nlc_small_kernel(const float* __restrict__ src, T* __restrict__ dst, long ld, int C, int L, long n) {
  GRID_STRIDE(i, n) {
    const long b = i / L; const int l = (int)(i - b * L);
    for (int c = 0; c < C; c++) st_f32(dst + i * ld + c, src[(b * C + c) * L + l]);
  }
}
template <typename T>
__global__ __launch_bounds__(NT) void nlc_to_ncl_small_kernel(const T* __restrict__ src, long ld, float* __restrict__ dst, int C, int L, long n) {
  GRID_STRIDE(i, n) {
    const long b = i / L; const int l = (int)(i - b * L);
    for (int c = 0; c < C; c++) dst[(b * C + c) * L + l] = ld_f32(src + i * ld + c);
  }
}

__global__ void pack_w_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int K, int unpack) {
  const long n = (long)Cout * Cin * K;
  GRID_STRIDE(i, n) {  // i indexes the packed layout [K][Cout][Cin]
    const int ci = (int)(i % Cin); const long r = i / Cin; const int co = (int)(r % Cout); const int k = (int)(r / Cout);
    const long ref = ((long)co * Cin + ci) * K + k;
    if (unpack) p[ref] = w[i]; else p[i] = w[ref];
  }
}
template <typename T> __global__ void cast_kernel(const float* __restrict__ s, T* __restrict__ d, long n) {
  GRID_STRIDE(i, n) st_f32(d + i, s[i]);
}
__global__ void fill_kernel(float* p, long n, float v) { GRID_STRIDE(i, n) p[i] = v; }

// ------------------------------------------------------------------ timestep embedding (unet.py:12-36)
template <typename T>
__global__ void temb_kernel(const int64_t* __restrict__ t, T* __restrict__ out, int B, int dim) {
  const int half = dim / 2;
  GRID_STRIDE(i, (long)B * dim) {
    const int b = (int)(i / dim), j = (int)(i % dim);
    float v = 0.f;
    if (j < 2 * half) {
      const int f = j < half ? j : j - half;
      const float freq = expf(-logf(10000.0f) * (float)f / (float)half);      // fp32 like the reference (unet.py:26-28); one ulp of it is 6e-5 in cos / sin at t ~ 1000
      const float a = (float)t[b] * freq;
      v = j < half ? cosf(a) : sinf(a);
    }
    st_f32(out + i, v);
  }
}
// y = silu(x) ; x fp32 [n] -> T
template <typename T> __global__ void silu_kernel(const float* __restrict__ x, T* __restrict__ y, long n) {
  GRID_STRIDE(i, n) st_f32(y + i, silu_f(x[i]));
}
// dx = dy * silu'(x): dy fp32, x fp32 -> T
template <typename T> __global__ void silu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, T* __restrict__ dx, long n) {
  GRID_STRIDE(i, n) st_f32(dx + i, dy[i] * silu_grad_f(x[i]));
}

// ------------------------------------------------------------------ column sums
// out[b][c] = sum_l X[b][l][c] (per sample, written) and/or total[c] += sum over everything (fp32 atomics).
// grid (LSPLIT, B); 4-channel vectors per thread, (column vector, row lane) tiling, LDS reduction over row lanes.
template <typename T, int V>
__global__ __launch_bounds__(NT) void colsum_kernel(const T* __restrict__ x, long ldx, float* __restrict__ out, long ldo,
                                                    float* __restrict__ total, float* __restrict__ parts, int L, int Cfull, int rows_per_block) {
  __shared__ float acc[1024];
  const int b = blockIdx.y, tid = threadIdx.x;
  // grid.z tiles the channels in chunks of 1024 (qkv biases have 1536, the batched embedding bias ~7k)
  const int c0 = blockIdx.z * 1024, C = min(1024, Cfull - c0);
  x += c0; if (out) out += c0; if (total) total += c0;
  for (int i = tid; i < C; i += NT) acc[i] = 0.f;
  __syncthreads();
  const int ncols = C / V;
  const int TX = ncols >= NT ? NT : ncols, TY = ncols >= NT ? 1 : NT / ncols;
  const int l0 = blockIdx.x * rows_per_block, l1 = min(L, l0 + rows_per_block);
  if (tid < TX * TY) {
    const int tx = tid % TX, ty = tid / TX;
    for (int col = tx; col < ncols; col += TX) {
      const int c = col * V;
      float s[V];
#pragma unroll
      for (int k = 0; k < V; k++) s[k] = 0.f;
#pragma unroll 8
      for (int l = l0 + ty; l < l1; l += TY) {
        const T* p = x + ((long)b * L + l) * ldx + c;
        if constexpr (V == 4 && sizeof(T) == 2) {
          const uint2 t = *(const uint2*)p;
          s[0] += w16_lo<T>(t.x); s[1] += w16_hi<T>(t.x);
          s[2] += w16_lo<T>(t.y); s[3] += w16_hi<T>(t.y);
        } else if constexpr (V == 4) {
          const float4 t = *(const float4*)p; s[0] += t.x; s[1] += t.y; s[2] += t.z; s[3] += t.w;
        } else {
          s[0] += ld_f32(p);
        }
      }
#pragma unroll
      for (int k = 0; k < V; k++) atomicAdd(&acc[c + k], s[k]);
    }
  }
  __syncthreads();
  for (int i = tid; i < C; i += NT) {
    if (out) out[(long)b * ldo + i] = acc[i];     // written, not accumulated: single L split only
    if (parts) parts[((long)b * gridDim.x + blockIdx.x) * Cfull + c0 + i] = acc[i];
    else if (total) atomicAdd(total + i, acc[i]);
  }
}
// second stage of the two-stage total: total[c] += sum over the nparts written partial rows (one thread per channel;
// thousands of blocks adding atomically into the same C addresses serialised in L2 and cost more than the streaming pass)
__global__ void colsum_finish_kernel(const float* __restrict__ parts, int nparts, int C, float* __restrict__ total) {
  __shared__ float red[NT];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), seg = threadIdx.x >> 6;      // 64 channels x 4 row lanes per block; grid.y row ranges
  const int per = (nparts + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * per, r1 = min(nparts, r0 + per);
  float s = 0.f;
  if (c < C) for (int r = r0 + seg; r < r1; r += NT / 64) s += parts[(long)r * C + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (seg == 0 && c < C) atomicAdd(total + c, red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]);
}

// ------------------------------------------------------------------ softmax over rows (unet.py:123)
// register-resident variants (n % 4 == 0, n <= 256 * K): a lane owns K float4 runs of the row, read ONCE with unconditional loads (clamped
// offset), one exp per element, packed stores -- the three-pass versions below re-read the row from the L2 for max, sum and output with
// 4-byte lane-strided loads (98 / 95 us on the 151 MB logits of the T = 768 attention, 2.3 / 3.2 TB/s)
template <typename T> __device__ __forceinline__ void sm_ld4(const T* p, float v[4]) {
  if constexpr (sizeof(T) == 4) { const float4 t = *(const float4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
  else { const uint2 t = *(const uint2*)p; v[0] = w16_lo<T>(t.x); v[1] = w16_hi<T>(t.x); v[2] = w16_lo<T>(t.y); v[3] = w16_hi<T>(t.y); }
}
template <typename T> __device__ __forceinline__ void sm_st4(T* p, const float v[4]) {
  if constexpr (sizeof(T) == 4) *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
  else { uint2 t; t.x = pack16x2<T>(v[0], v[1]); t.y = pack16x2<T>(v[2], v[3]); *(uint2*)p = t; }
}
template <typename T, int K>
__global__ __launch_bounds__(NT) void softmax_reg_kernel(const float* __restrict__ S, T* __restrict__ P, long rows, int n) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* s = S + row * n;
  float4 v[K];
#pragma unroll
  for (int k = 0; k < K; k++) { const int i = (k * 64 + lane) * 4; v[k] = *(const float4*)(s + (i < n ? i : n - 4)); }
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < K; k++) if ((k * 64 + lane) * 4 < n) mx = fmaxf(mx, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < K; k++) {
    v[k].x = expf(v[k].x - mx); v[k].y = expf(v[k].y - mx); v[k].z = expf(v[k].z - mx); v[k].w = expf(v[k].w - mx);
    if ((k * 64 + lane) * 4 < n) sum += (v[k].x + v[k].y) + (v[k].z + v[k].w);
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = (k * 64 + lane) * 4;
    if (i < n) { const float o[4] = {v[k].x * inv, v[k].y * inv, v[k].z * inv, v[k].w * inv}; sm_st4<T>(P + row * n + i, o); }
  }
}
template <typename T, int K>
__global__ __launch_bounds__(NT) void softmax_bwd_reg_kernel(const float* __restrict__ dP, const T* __restrict__ P, T* __restrict__ dS,
                                                             long rows, int n, float alpha) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float4 g[K]; float pv[K][4];
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = (k * 64 + lane) * 4, ic = i < n ? i : n - 4;
    g[k] = *(const float4*)(dP + row * n + ic);
    sm_ld4<T>(P + row * n + ic, pv[k]);
  }
  float dot = 0.f;
#pragma unroll
  for (int k = 0; k < K; k++) if ((k * 64 + lane) * 4 < n) dot += (g[k].x * pv[k][0] + g[k].y * pv[k][1]) + (g[k].z * pv[k][2] + g[k].w * pv[k][3]);
  dot = wave_sum(dot);
#pragma unroll
  for (int k = 0; k < K; k++) {
    const int i = (k * 64 + lane) * 4;
    if (i < n) {
      const float o[4] = {alpha * pv[k][0] * (g[k].x - dot), alpha * pv[k][1] * (g[k].y - dot), alpha * pv[k][2] * (g[k].z - dot), alpha * pv[k][3] * (g[k].w - dot)};
      sm_st4<T>(dS + row * n + i, o);
    }
  }
}
// one wave per row; S fp32 [rows][n] -> P (T) [rows][n]
template <typename T>
__global__ __launch_bounds__(NT) void softmax_kernel(const float* __restrict__ S, T* __restrict__ P, long rows, int n) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* s = S + row * n;
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, s[i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
  for (int i = lane; i < n; i += 64) sum += expf(s[i] - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int i = lane; i < n; i += 64) st_f32(P + row * n + i, expf(s[i] - mx) * inv);
}
// dS = alpha * P o (dP - sum(dP o P)) ; dP fp32, P (T) -> dS (T)
template <typename T>
__global__ __launch_bounds__(NT) void softmax_bwd_kernel(const float* __restrict__ dP, const T* __restrict__ P, T* __restrict__ dS,
                                                         long rows, int n, float alpha) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  float dot = 0.f;
  for (int i = lane; i < n; i += 64) dot += dP[row * n + i] * ld_f32(P + row * n + i);
  dot = wave_sum(dot);
  for (int i = lane; i < n; i += 64) {
    const float p = ld_f32(P + row * n + i);
    st_f32(dS + row * n + i, alpha * p * (dP[row * n + i] - dot));
  }
}

// dst[r][c] += src[r][c] (both T, own leading dims); 4-channel vectors, 2-D grid (no per-element division)
template <typename T>
__global__ void add_rows_kernel(T* __restrict__ dst, long ldd, const T* __restrict__ src, long lds, long rows, int C) {
  const int nc = C / 4;
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    for (int cv = threadIdx.x; cv < nc; cv += blockDim.x) {
      T* d = dst + r * ldd + cv * 4; const T* sp = src + r * lds + cv * 4;
      if constexpr (sizeof(T) == 2) {
        const uint2 a = *(const uint2*)d, b = *(const uint2*)sp;
        uint2 o;
        o.x = pack16x2<T>(w16_lo<T>(a.x) + w16_lo<T>(b.x), w16_hi<T>(a.x) + w16_hi<T>(b.x));
        o.y = pack16x2<T>(w16_lo<T>(a.y) + w16_lo<T>(b.y), w16_hi<T>(a.y) + w16_hi<T>(b.y));
        *(uint2*)d = o;
      } else {
        float4 a = *(const float4*)d; const float4 b = *(const float4*)sp;
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; *(float4*)d = a;
      }
    }
    for (int c = nc * 4 + threadIdx.x; c < C; c += blockDim.x) { T* d = dst + r * ldd + c; st_f32(d, ld_f32(d) + ld_f32(src + r * lds + c)); }
  }
}
// ------------------------------------------------------------------ use_scale_shift_norm (unet.py:318-322)
// a[r][c] = silu(u), u = hn[r][c] * (1 + scale[b][c]) + shift[b][c], b = r / L; one block column per sample, threads along the channels
template <typename T>
__global__ void film_silu_fwd_kernel(const T* __restrict__ hn, long ldh, const float* __restrict__ emb, long lde, T* __restrict__ a, long lda, int L, int C) {
  const int b = blockIdx.y;
  const float* e = emb + (long)b * lde;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float s1 = 1.0f + e[c], sh = e[C + c];
    for (int l = blockIdx.x; l < L; l += gridDim.x) {
      const long r = (long)b * L + l;
      const float u = fmaf(ld_f32(hn + r * ldh + c), s1, sh);
      st_f32(a + r * lda + c, u / (1.0f + expf(-u)));
    }
  }
}
// backward: g = da * silu'(u); dhn = g * (1 + scale); demb[b] = [sum_l g * hn | sum_l g].  One block = 64 channels x 4 row lanes of one sample.
template <typename T>
__global__ __launch_bounds__(256) void film_silu_bwd_kernel(const T* __restrict__ hn, long ldh, const float* __restrict__ emb, long lde, const T* __restrict__ da,
                                                            long ldda, T* __restrict__ dhn, long lddh, float* __restrict__ demb, long ldde, int L, int C) {
  __shared__ float r1[4][64], r2[4][64];
  const int b = blockIdx.y, cx = threadIdx.x & 63, ry = threadIdx.x >> 6, c = blockIdx.x * 64 + cx;
  float s1 = 0.f, s2 = 0.f;
  if (c < C) {
    const float sc = 1.0f + emb[(long)b * lde + c], sh = emb[(long)b * lde + C + c];
    for (int l = ry; l < L; l += 4) {
      const long r = (long)b * L + l;
      const float h = ld_f32(hn + r * ldh + c), u = fmaf(h, sc, sh);
      const float sg = 1.0f / (1.0f + expf(-u));
      const float g = ld_f32(da + r * ldda + c) * sg * (1.0f + u * (1.0f - sg));
      st_f32(dhn + r * lddh + c, g * sc);
      s1 = fmaf(g, h, s1); s2 += g;
    }
  }
  r1[ry][cx] = s1; r2[ry][cx] = s2;
  __syncthreads();
  if (ry == 0 && c < C) {
    demb[(long)b * ldde + c] = r1[0][cx] + r1[1][cx] + r1[2][cx] + r1[3][cx];
    demb[(long)b * ldde + C + c] = r2[0][cx] + r2[1][cx] + r2[2][cx] + r2[3][cx];
  }
}
template <typename T>
__global__ void copy_rows_kernel(T* __restrict__ dst, long ldd, const T* __restrict__ src, long lds, long rows, int C) {
  GRID_STRIDE(i, rows * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    dst[r * ldd + c] = src[r * lds + c];
  }
}

// ------------------------------------------------------------------ stand-alone resampling (unet.py:177-224: Downsample / Upsample with use_conv = False)
// The executors never launch these: inside a ResBlock the resampling rides the GroupNorm kernels (norm.hip `resample`).  They are the
// primitive-granularity form of the same two ops behind the C ABI.  Rows are (sample, position) flattened; L is even, so the pair
// (2r, 2r + 1) never straddles two samples.  MODE 0: y[r] = (x[2r] + x[2r+1]) / 2   (AvgPool1d(2, 2) forward)
//                                            MODE 1: y[r] = x[2r] + x[2r+1]           (nearest x 2 backward)
//                                            MODE 2: y[2r] = y[2r+1] = x[r] / 2       (AvgPool1d(2, 2) backward)
//                                            MODE 3: y[2r] = y[2r+1] = x[r]           (nearest x 2 forward)
template <typename T, int MODE>
__global__ __launch_bounds__(NT) void resample2_kernel(const T* __restrict__ x, long ldx, T* __restrict__ y, long ldy, long rows_small, int C) {
  GRID_STRIDE(i, rows_small * C) {
    const long r = i / C; const int c = (int)(i - r * C);
    if constexpr (MODE <= 1) {
      const float a = ld_f32(x + (2 * r) * ldx + c), b = ld_f32(x + (2 * r + 1) * ldx + c);
      st_f32(y + r * ldy + c, MODE == 0 ? (a + b) * 0.5f : a + b);
    } else {
      const float a = ld_f32(x + r * ldx + c), v = MODE == 2 ? a * 0.5f : a;
      st_f32(y + (2 * r) * ldy + c, v); st_f32(y + (2 * r + 1) * ldy + c, v);
    }
  }
}

// ------------------------------------------------------------------ schedulers (training.py:429-436, sample_trials.py:163)
__global__ void add_noise_kernel(const float* __restrict__ x, const float* __restrict__ nz, const int64_t* __restrict__ t,
                                 const float* __restrict__ acp, float* __restrict__ out, long n, long per, int velocity) {
  GRID_STRIDE(i, n) {
    const float a = acp[t[i / per]];
    const float sa = sqrtf(a), sb = sqrtf(1.0f - a);
    out[i] = velocity ? (sa * nz[i] - sb * x[i]) : (sa * x[i] + sb * nz[i]);
  }
}
__global__ void ddim_step_kernel(const float* __restrict__ mo, const float* __restrict__ x, float a_t, float a_prev, int pred,
                                 int clip, float* __restrict__ prev, float* __restrict__ x0o, long n) {
  const float sa = sqrtf(a_t), sb = sqrtf(1.0f - a_t), sap = sqrtf(a_prev), sbp = sqrtf(1.0f - a_prev);
  GRID_STRIDE(i, n) {
    const float o = mo[i], s = x[i];
    float x0, e;
    if (pred == EEGLDM_PRED_EPSILON) { x0 = (s - sb * o) / sa; e = o; }
    else if (pred == EEGLDM_PRED_V) { x0 = sa * s - sb * o; e = sa * o + sb * s; }
    else { x0 = o; e = (s - sa * x0) / sb; }
    if (clip) x0 = fminf(1.0f, fmaxf(-1.0f, x0));
    prev[i] = sap * x0 + sbp * e;
    if (x0o) x0o[i] = x0;
  }
}

// DDIMScheduler.step with eta > 0 (Song et al. eq. 12 / 16): sigma = eta sqrt((1 - a_prev) / (1 - a_t) (1 - a_t / a_prev)),
// prev = sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma^2) e + sigma noise; eta = 0 is ddim_step_kernel
__global__ void ddim_step_eta_kernel(const float* __restrict__ mo, const float* __restrict__ x, const float* __restrict__ nz, float a_t, float a_prev,
                                     float sigma, float dir, int pred, int clip, float* __restrict__ prev, float* __restrict__ x0o, long n) {
  const float sa = sqrtf(a_t), sb = sqrtf(1.0f - a_t), sap = sqrtf(a_prev);
  GRID_STRIDE(i, n) {
    const float o = mo[i], s = x[i];
    float x0, e;
    if (pred == EEGLDM_PRED_EPSILON) { x0 = (s - sb * o) / sa; e = o; }
    else if (pred == EEGLDM_PRED_V) { x0 = sa * s - sb * o; e = sa * o + sb * s; }
    else { x0 = o; e = (s - sa * x0) / sb; }
    if (clip) x0 = fminf(1.0f, fmaxf(-1.0f, x0));
    prev[i] = fmaf(sigma, nz[i], fmaf(sap, x0, dir * e));
    if (x0o) x0o[i] = x0;
  }
}

// DDPM ancestral step (DDPMScheduler.step, variance_type fixed_small: the 1000-step logging sampler of util.py:241-243,261-285 and
// sample_trials_ddpm.py:99-102; same arithmetic as DDPM.p_sample, /root/reference/src/models/ldm.py:311-357):
//   x0 from the prediction type, optional clamp, mean = c0 * x0 + ct * x_t, plus sigma * noise when t > 0 (sigma = 0 at t = 0)
__global__ void ddpm_step_kernel(const float* __restrict__ mo, const float* __restrict__ x, const float* __restrict__ nz, float sa, float sb,
                                 float c0, float ct, float sigma, int pred, int clip, float* __restrict__ prev, float* __restrict__ x0o, long n) {
  GRID_STRIDE(i, n) {
    const float o = mo[i], s = x[i];
    float x0;
    if (pred == EEGLDM_PRED_EPSILON) x0 = (s - sb * o) / sa;
    else if (pred == EEGLDM_PRED_V) x0 = sa * s - sb * o;
    else x0 = o;
    if (clip) x0 = fminf(1.0f, fmaxf(-1.0f, x0));
    float m = c0 * x0 + ct * s;
    if (sigma != 0.0f) m += sigma * nz[i];
    prev[i] = m;
    if (x0o) x0o[i] = x0;
  }
}

// ------------------------------------------------------------------ MSE (training.py:437)
__global__ __launch_bounds__(NT) void mse_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ loss,
                                                 float* __restrict__ dp, long n, float inv_n, float gscale, float* __restrict__ parts) {
  float s = 0.f;
  GRID_STRIDE(i, n) {
    const float d = p[i] - t[i];
    s += d * d;
    if (dp) dp[i] = 2.0f * d * inv_n * gscale;
  }
  s = wave_sum(s);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { const float v = (red[0] + red[1] + red[2] + red[3]) * inv_n; if (parts) parts[blockIdx.x] = v; else atomicAdd(loss, v); }
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam defaults, train_ldm.py:208)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float ginv) {
  GRID_STRIDE(i, n) {
    const float gi = g[i] * ginv;
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
  }
}

// ------------------------------------------------------------------ Philox4x32-10 (perf-path RNG; parity runs pass noise in)
__device__ __forceinline__ void philox_round(unsigned& c0, unsigned& c1, unsigned& c2, unsigned& c3, unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}
__device__ __forceinline__ void philox(unsigned long long seed, unsigned long long ctr, unsigned r[4]) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0, c3 = 0;
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; i++) { philox_round(c0, c1, c2, c3, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
__global__ void randn_kernel(float* __restrict__ out, long n, unsigned long long seed, unsigned long long offset) {
  const long nq = (n + 3) / 4;
  GRID_STRIDE(i, nq) {
    unsigned r[4]; philox(seed, offset + (unsigned long long)i, r);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float u1 = ((float)r[2 * h] + 1.0f) * 2.3283064365386963e-10f;   // (0,1]
      const float u2 = (float)r[2 * h + 1] * 2.3283064365386963e-10f;
      const float rad = sqrtf(-2.0f * logf(u1));
      z[2 * h] = rad * cosf(6.283185307179586f * u2); z[2 * h + 1] = rad * sinf(6.283185307179586f * u2);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) if (i * 4 + k < n) out[i * 4 + k] = z[k];
  }
}
// nn.Dropout(p) on a [rows][C] tensor in place (ResBlock.out_layers[2], unet.py:289): x <- keep ? x / (1 - p) : 0, keep from Philox(seed, offset + e / 4)
// word e % 4 of element e = r * C + c.  The backward applies the SAME call (same seed / offset) to the incoming gradient: the mask is
// regenerated, never stored.
template <typename T>
__global__ void dropout_rows_kernel(T* __restrict__ x, long ld, long rows, int C, float p, float inv_keep, unsigned long long seed, unsigned long long offset) {
  const long nq = (rows * C + 3) / 4;
  GRID_STRIDE(i, nq) {
    unsigned r[4]; philox(seed, offset + (unsigned long long)i, r);
    const long e0 = i * 4;
    if constexpr (sizeof(T) == 2) {
      // C % 4 == 0 and ld % 4 == 0 (checked by the launcher for this path): the four elements of a counter sit in one row, 8 bytes apart from nothing
      if ((C & 3) == 0 && (ld & 3) == 0) {
        const long row = e0 / C; const int c = (int)(e0 - row * C);
        uint2* q = (uint2*)(x + row * ld + c);
        const uint2 v = *q;
        const bool k0 = (float)r[0] * 2.3283064365386963e-10f >= p, k1 = (float)r[1] * 2.3283064365386963e-10f >= p;
        const bool k2 = (float)r[2] * 2.3283064365386963e-10f >= p, k3 = (float)r[3] * 2.3283064365386963e-10f >= p;
        uint2 o;
        o.x = pack16x2<T>(k0 ? w16_lo<T>(v.x) * inv_keep : 0.0f, k1 ? w16_hi<T>(v.x) * inv_keep : 0.0f);
        o.y = pack16x2<T>(k2 ? w16_lo<T>(v.y) * inv_keep : 0.0f, k3 ? w16_hi<T>(v.y) * inv_keep : 0.0f);
        *q = o;
        continue;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const long e = e0 + k;
      if (e < rows * C) {
        const long row = e / C; const int c = (int)(e - row * C);
        T* q = x + row * ld + c;
        const bool keep = (float)r[k] * 2.3283064365386963e-10f >= p;
        st_f32(q, keep ? ld_f32(q) * inv_keep : 0.0f);
      }
    }
  }
}
__global__ void randint_kernel(int64_t* __restrict__ out, long n, int64_t high, unsigned long long seed, unsigned long long offset) {
  GRID_STRIDE(i, n) {
    unsigned r[4]; philox(seed, offset + (unsigned long long)i, r);
    const unsigned long long v = ((unsigned long long)r[0] << 32) | r[1];
    out[i] = (int64_t)(v % (unsigned long long)high);
  }
}
}  // namespace

// ================================================================== internal launchers (used by the executors)
#define DISPATCH_T(dtype, ...)                                            \
  do {                                                                    \
    if ((dtype) == EEGLDM_F32) { typedef float T; __VA_ARGS__; }          \
    else if ((dtype) == EEGLDM_BF16) { typedef bf16_t T; __VA_ARGS__; }   \
    else if ((dtype) == EEGLDM_F16) { typedef f16_t T; __VA_ARGS__; }     \
    else EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "dtype %d", (int)(dtype));      \
  } while (0)

int ew_temb(eegldm_ctx* ctx, const int64_t* t, void* out, int B, int dim, int dtype) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((temb_kernel<T>), dim3(grid1d((long)B * dim, ctx)), dim3(NT), 0, ctx->stream, t, (T*)out, B, dim));
  LAUNCH_CHECK(); return 0;
}
int ew_silu(eegldm_ctx* ctx, const float* x, void* y, long n, int dtype) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((silu_kernel<T>), dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, x, (T*)y, n));
  LAUNCH_CHECK(); return 0;
}
int ew_silu_bwd(eegldm_ctx* ctx, const float* dy, const float* x, void* dx, long n, int dtype) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((silu_bwd_kernel<T>), dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, dy, x, (T*)dx, n));
  LAUNCH_CHECK(); return 0;
}
// out_ps: per-sample sums [B][ldo] fp32 (written; single L split) or NULL; total: fp32 [C] accumulated (+=) or NULL
// ---- deterministic column sums (EEGLDM_DETERMINISTIC=1): no atomics anywhere.  Stage 1: thread = channel, block = (row segment, sample):
// the rows of the segment are added in order and the partial row is written.  Stage 2: thread = channel, the partial rows are added in order
// (segments of a sample, then samples) in fp64.
template <typename T>
__global__ __launch_bounds__(NT) void colsum_det_kernel(const T* __restrict__ x, long ldx, float* __restrict__ parts, int L, int C, int rows_per_seg) {
  const int c = blockIdx.z * NT + threadIdx.x;
  if (c >= C) return;
  const int b = blockIdx.y, l0 = blockIdx.x * rows_per_seg, l1 = min(L, l0 + rows_per_seg);
  float s = 0.f;
  for (int l = l0; l < l1; l++) s += ld_f32(x + ((long)b * L + l) * ldx + c);
  parts[((long)b * gridDim.x + blockIdx.x) * C + c] = s;
}
// out_ps[b][c] = sum over the nseg partial rows of sample b, in segment order (thread = (sample, channel))
__global__ __launch_bounds__(NT) void colsum_det_ps_kernel(const float* __restrict__ parts, int nseg, int C, float* __restrict__ out_ps, long ldo) {
  const int c = blockIdx.x * NT + threadIdx.x, b = blockIdx.y;
  if (c >= C) return;
  double s = 0.0;
  for (int g = 0; g < nseg; g++) s += (double)parts[((long)b * nseg + g) * C + c];
  out_ps[(long)b * ldo + c] = (float)s;
}
// total[i] += sum_p parts[p * stride + off + i] with a FIXED two-level shape: 16 lanes per element, lane q adds the rows q, q + 16, ... in
// order (fp64), then the 16 lane sums are added in lane order.  A block serves 16 elements; the result does not depend on timing.
__global__ __launch_bounds__(NT) void fold_partials_det_kernel(const float* __restrict__ parts, int nparts, long stride, int off, int n, float* __restrict__ total) {
  __shared__ double red[16][17];
  const int e = threadIdx.x & 15, q = threadIdx.x >> 4, i = blockIdx.x * 16 + e;
  double s = 0.0;
  if (i < n) {
#pragma unroll 4
    for (int p = q; p < nparts; p += 16) s += (double)parts[(long)p * stride + off + i];
  }
  red[q][e] = s;
  __syncthreads();
  if (q == 0 && i < n) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 16; k++) t += red[k][e];
    total[i] += (float)t;
  }
}
int ew_fold_partials_det(eegldm_ctx* ctx, const float* parts, int nparts, long stride, int off, int n, float* total) {
  hipLaunchKernelGGL(fold_partials_det_kernel, dim3((n + 15) / 16), dim3(NT), 0, ctx->stream, parts, nparts, stride, off, n, total);
  LAUNCH_CHECK(); return 0;
}
int ew_colsum(eegldm_ctx* ctx, const void* x, long ldx, float* out_ps, long ldo, float* total, int B, int L, int C, int dtype) {
  if (eeg_deterministic()) {
    long nseg = ((size_t)16 << 20) / ((size_t)B * C * sizeof(float)); if (nseg > (L + 7) / 8) nseg = (L + 7) / 8; if (nseg > 64) nseg = 64; if (nseg < 1) nseg = 1;
    int rps = (int)((L + nseg - 1) / nseg); nseg = (L + rps - 1) / rps;
    float* parts = nullptr; EEG_TRY(eeg_det_buffer(ctx, (size_t)B * nseg * C * sizeof(float), &parts));
    const dim3 grid((unsigned)nseg, (unsigned)B, (unsigned)((C + NT - 1) / NT));
    DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_det_kernel<T>), grid, dim3(NT), 0, ctx->stream, (const T*)x, ldx, parts, L, C, rps));
    LAUNCH_CHECK();
    if (out_ps) {
      hipLaunchKernelGGL(colsum_det_ps_kernel, dim3((C + NT - 1) / NT, B), dim3(NT), 0, ctx->stream, parts, (int)nseg, C, out_ps, ldo);
      LAUNCH_CHECK();
      if (total) EEG_TRY(ew_fold_partials_det(ctx, out_ps, B, ldo, 0, C, total));
    } else if (total) EEG_TRY(ew_fold_partials_det(ctx, parts, (int)(B * nseg), C, 0, C, total));
    return 0;
  }
  int lsplit = 1, rpb = L;
  if (!out_ps) {  // free to split L when only the fp32 atomic total is wanted
    int want = (ctx->num_cu * 8 + B - 1) / B; if (want < 1) want = 1;   // 8 blocks (2048 threads) per CU: enough loads in flight to stream
    int maxs = (L + 31) / 32; lsplit = want > maxs ? maxs : want; rpb = (L + lsplit - 1) / lsplit; lsplit = (L + rpb - 1) / rpb;
  }
  dim3 grid(lsplit, B, (C + 1023) / 1024);
  const bool v4 = (C % 4 == 0) && (ldx % 4 == 0);
  // totals over many blocks: written partials + a finishing pass instead of same-address atomics
  float* parts = nullptr;
  const long nparts = (long)lsplit * B;
  if (total && nparts >= 64 && (size_t)nparts * C * sizeof(float) <= (16u << 20)) parts = (float*)((char*)ctx->scratch + (8u << 20));
  if (v4) { DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T, 4>), grid, dim3(NT), 0, ctx->stream, (const T*)x, ldx, out_ps, ldo, total, parts, L, C, rpb)); }
  else { DISPATCH_T(dtype, hipLaunchKernelGGL((colsum_kernel<T, 1>), grid, dim3(NT), 0, ctx->stream, (const T*)x, ldx, out_ps, ldo, total, parts, L, C, rpb)); }
  LAUNCH_CHECK();
  if (parts) { hipLaunchKernelGGL(colsum_finish_kernel, dim3((C + 63) / 64, 32), dim3(NT), 0, ctx->stream, parts, (int)nparts, C, total); LAUNCH_CHECK(); }
  return 0;
}
// total[i] += sum over nparts rows of parts[r][i] (i < n): the finishing pass of the written-partials reductions
int ew_fold_partials(eegldm_ctx* ctx, const float* parts, int nparts, int n, float* total) {
  if (eeg_deterministic()) return ew_fold_partials_det(ctx, parts, nparts, n, 0, n, total);
  hipLaunchKernelGGL(colsum_finish_kernel, dim3((n + 63) / 64, 32), dim3(NT), 0, ctx->stream, parts, nparts, n, total);
  LAUNCH_CHECK(); return 0;
}
int ew_softmax(eegldm_ctx* ctx, const float* S, void* P, long rows, int n, int dtype) {
  if (n % 4 == 0 && n >= 4 && n <= 1024) {
    const dim3 g((unsigned)((rows + 3) / 4));
#define SMX(K_) DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_reg_kernel<T, K_>), g, dim3(NT), 0, ctx->stream, S, (T*)P, rows, n))
    if (n <= 256) SMX(1); else if (n <= 512) SMX(2); else if (n <= 768) SMX(3); else SMX(4);
#undef SMX
    LAUNCH_CHECK(); return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0, ctx->stream, S, (T*)P, rows, n));
  LAUNCH_CHECK(); return 0;
}
int ew_softmax_bwd(eegldm_ctx* ctx, const float* dP, const void* P, void* dS, long rows, int n, float alpha, int dtype) {
  if (n % 4 == 0 && n >= 4 && n <= 1024) {
    const dim3 g((unsigned)((rows + 3) / 4));
#define SMB(K_) DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_bwd_reg_kernel<T, K_>), g, dim3(NT), 0, ctx->stream, dP, (const T*)P, (T*)dS, rows, n, alpha))
    if (n <= 256) SMB(1); else if (n <= 512) SMB(2); else if (n <= 768) SMB(3); else SMB(4);
#undef SMB
    LAUNCH_CHECK(); return 0;
  }
  DISPATCH_T(dtype, hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(NT), 0, ctx->stream, dP, (const T*)P, (T*)dS, rows, n, alpha));
  LAUNCH_CHECK(); return 0;
}
int ew_add_rows(eegldm_ctx* ctx, void* dst, long ldd, const void* src, long lds, long rows, int C, int dtype) {
  EEG_CHECK(ldd % 4 == 0 && lds % 4 == 0, "add_rows: leading dimensions must be multiples of 4");
  const long cap = (long)ctx->num_cu * 32;
  DISPATCH_T(dtype, hipLaunchKernelGGL((add_rows_kernel<T>), dim3((unsigned)(rows < cap ? rows : cap)), dim3(C >= 512 ? 128 : 64), 0, ctx->stream, (T*)dst, ldd, (const T*)src, lds, rows, C));
  LAUNCH_CHECK(); return 0;
}
int ew_film_silu_fwd(eegldm_ctx* ctx, const void* hn, long ldh, const float* emb, long lde, void* a, long lda, int B, int L, int C, int dtype) {
  const int gx = L < 64 ? L : 64;
  DISPATCH_T(dtype, hipLaunchKernelGGL((film_silu_fwd_kernel<T>), dim3(gx, B), dim3(C >= 256 ? 256 : (C >= 128 ? 128 : 64)), 0, ctx->stream, (const T*)hn, ldh, emb, lde, (T*)a, lda, L, C));
  LAUNCH_CHECK(); return 0;
}
int ew_film_silu_bwd(eegldm_ctx* ctx, const void* hn, long ldh, const float* emb, long lde, const void* da, long ldda, void* dhn, long lddh,
                     float* demb, long ldde, int B, int L, int C, int dtype) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((film_silu_bwd_kernel<T>), dim3((C + 63) / 64, B), dim3(256), 0, ctx->stream, (const T*)hn, ldh, emb, lde, (const T*)da, ldda,
                                       (T*)dhn, lddh, demb, ldde, L, C));
  LAUNCH_CHECK(); return 0;
}
int ew_dropout_rows(eegldm_ctx* ctx, void* x, long ld, long rows, int C, float p, uint64_t seed, uint64_t offset, int dtype) {
  EEG_CHECK(p >= 0.0f && p < 1.0f, "dropout probability %g outside [0, 1)", (double)p);
  const long nq = (rows * C + 3) / 4;
  DISPATCH_T(dtype, hipLaunchKernelGGL((dropout_rows_kernel<T>), dim3(grid1d(nq, ctx)), dim3(NT), 0, ctx->stream, (T*)x, ld, rows, C, p, 1.0f / (1.0f - p),
                                       (unsigned long long)seed, (unsigned long long)offset));
  LAUNCH_CHECK(); return 0;
}
int ew_copy_rows(eegldm_ctx* ctx, void* dst, long ldd, const void* src, long lds, long rows, int C, int dtype) {
  DISPATCH_T(dtype, hipLaunchKernelGGL((copy_rows_kernel<T>), dim3(grid1d(rows * C, ctx)), dim3(NT), 0, ctx->stream, (T*)dst, ldd, (const T*)src, lds, rows, C));
  LAUNCH_CHECK(); return 0;
}

// ================================================================== C ABI
extern "C" int eegldm_ncl_to_nlc(eegldm_ctx* ctx, const float* src, void* dst, long ld, int B, int C, int L, int dtype) {
  EEG_CHECK(B > 0 && C > 0 && L > 0 && ld >= C, "bad shape");
  if (C <= 8) {
    const long n = (long)B * L;
    DISPATCH_T(dtype, hipLaunchKernelGGL((ncl_to_nlc_small_kernel<T>), dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, src, (T*)dst, ld, C, L, n));
    LAUNCH_CHECK(); return 0;
  }
  dim3 grid((L + 63) / 64, (C + 63) / 64, B);
  DISPATCH_T(dtype, hipLaunchKernelGGL((ncl_to_nlc_kernel<T>), grid, dim3(NT), 0, ctx->stream, src, (T*)dst, ld, C, L));
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_nlc_to_ncl(eegldm_ctx* ctx, const void* src, long ld, float* dst, int B, int C, int L, int dtype) {
  EEG_CHECK(B > 0 && C > 0 && L > 0 && ld >= C, "bad shape");
  if (C <= 8) {
    const long n = (long)B * L;
    DISPATCH_T(dtype, hipLaunchKernelGGL((nlc_to_ncl_small_kernel<T>), dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, (const T*)src, ld, dst, C, L, n));
    LAUNCH_CHECK(); return 0;
  }
  dim3 grid((L + 63) / 64, (C + 63) / 64, B);
  DISPATCH_T(dtype, hipLaunchKernelGGL((nlc_to_ncl_kernel<T>), grid, dim3(NT), 0, ctx->stream, (const T*)src, ld, dst, C, L));
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_pack_conv_weight(eegldm_ctx* ctx, const float* w, float* p, int Cout, int Cin, int K) {
  hipLaunchKernelGGL(pack_w_kernel, dim3(grid1d((long)Cout * Cin * K, ctx)), dim3(NT), 0, ctx->stream, w, p, Cout, Cin, K, 0);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_unpack_conv_weight(eegldm_ctx* ctx, const float* p, float* w, int Cout, int Cin, int K) {
  hipLaunchKernelGGL(pack_w_kernel, dim3(grid1d((long)Cout * Cin * K, ctx)), dim3(NT), 0, ctx->stream, p, w, Cout, Cin, K, 1);
  LAUNCH_CHECK(); return 0;
}
// [3][Cout][Cin] -> [3][Cin / 32][Cout][32] for every table entry; one 16-byte chunk (8 elements) per thread
__global__ void kblk_pack_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const KbDesc* __restrict__ tab, int n, long total) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= total) return;
  int e = 0;
  while (e + 1 < n && tab[e + 1].chunk0 <= c) e++;
  const KbDesc d = tab[e];
  const long r = c - d.chunk0;                 // chunk index inside the weight: (tap, co, ci / 8)
  const int cpr = d.cin / 8;
  const int ci8 = (int)(r % cpr); const long tc = r / cpr;
  const int co = (int)(tc % d.cout), t = (int)(tc / d.cout);
  const long o = (((long)t * (d.cin / 32) + ci8 / 4) * d.cout + co) * 4 + (ci8 & 3);
  dst[d.off / 8 + o] = src[d.off / 8 + r];
}
__global__ void kblk_pack1_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, int cout, int cin, long total) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= total) return;
  const int cpr = cin / 8;
  const int ci8 = (int)(r % cpr); const long tc = r / cpr;
  const int co = (int)(tc % cout), t = (int)(tc / cout);
  dst[(((long)t * (cin / 32) + ci8 / 4) * cout + co) * 4 + (ci8 & 3)] = src[r];
}
// [3][Cout][Cin] -> [3][Cout / 32][Cin][32]: one 16-byte OUTPUT chunk (8 consecutive co of one ci) per thread, threads consecutive in ci
// (the eight 2-byte reads of a thread are each coalesced across the wave)
__device__ __forceinline__ void kblk_t_chunk(const bf16_t* __restrict__ src, uint4* __restrict__ dst, int cout, int cin, long r) {
  const int ci = (int)(r % cin); long x = r / cin;
  const int c4 = (int)(x & 3); x >>= 2;
  const int cb = (int)(x % (cout / 32)), t = (int)(x / (cout / 32));
  const bf16_t* s = src + ((long)t * cout + cb * 32 + c4 * 8) * cin + ci;
  unsigned v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) v[k] = (unsigned)s[(long)(2 * k) * cin] | ((unsigned)s[(long)(2 * k + 1) * cin] << 16);
  dst[(((long)t * (cout / 32) + cb) * cin + ci) * 4 + c4] = make_uint4(v[0], v[1], v[2], v[3]);
}
__global__ void kblk_pack_t_kernel(const bf16_t* __restrict__ src, uint4* __restrict__ dst, const KbDesc* __restrict__ tab, int n, long total) {
  const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= total) return;
  int e = 0;
  while (e + 1 < n && tab[e + 1].chunk0 <= c) e++;
  const KbDesc d = tab[e];
  kblk_t_chunk(src + d.off, dst + d.off / 8, d.cout, d.cin, c - d.chunk0);
}
__global__ void kblk_pack_t1_kernel(const bf16_t* __restrict__ src, uint4* __restrict__ dst, int cout, int cin, long total) {
  const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < total) kblk_t_chunk(src, dst, cout, cin, r);
}
int kblk_pack_t_one(eegldm_ctx* ctx, const void* w_plain, void* w_packed, int Cout, int Cin, int taps) {
  const long total = (long)taps * Cout * Cin / 8;
  hipLaunchKernelGGL(kblk_pack_t1_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, ctx->stream, (const bf16_t*)w_plain, (uint4*)w_packed, Cout, Cin, total);
  LAUNCH_CHECK(); return 0;
}
int kblk_pack_t(eegldm_ctx* ctx, const void* w_plain, void* w_packed, const KbDesc* d_table, int n, long total_chunks) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(kblk_pack_t_kernel, dim3((unsigned)((total_chunks + NT - 1) / NT)), dim3(NT), 0, ctx->stream, (const bf16_t*)w_plain, (uint4*)w_packed,
                     d_table, n, total_chunks);
  LAUNCH_CHECK(); return 0;
}
int kblk_pack_one(eegldm_ctx* ctx, const void* w_plain, void* w_packed, int Cout, int Cin, int taps) {
  const long total = (long)taps * Cout * Cin / 8;
  hipLaunchKernelGGL(kblk_pack1_kernel, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, ctx->stream, (const uint4*)w_plain, (uint4*)w_packed, Cout, Cin, total);
  LAUNCH_CHECK(); return 0;
}
int kblk_pack(eegldm_ctx* ctx, const void* w_plain, void* w_packed, const KbDesc* d_table, int n, long total_chunks) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(kblk_pack_kernel, dim3((unsigned)((total_chunks + NT - 1) / NT)), dim3(NT), 0, ctx->stream, (const uint4*)w_plain, (uint4*)w_packed,
                     d_table, n, total_chunks);
  LAUNCH_CHECK(); return 0;
}
// ---- stride-2 3-tap convs of 64 input channels as STRIDE-1 convs of the weight-stationary kernel (conv_ws.hip), round 6.
// Two consecutive input rows of an NLC tensor with ld = Cin are one row of 2 Cin channels, so with x'[m] = [x[2m] | x[2m + 1]]
//   forward   y[m]   = W0 x[2m - 1] + W1 x[2m] + W2 x[2m + 1]            = [0 | W0] x'[m - 1] + [W1 | W2] x'[m]
//   backward  [dx[2m] | dx[2m + 1]] = [W1^T | W2^T] dy[m] + [0 | W0^T] dy[m + 1]                         (dx' = the paired view of dx)
// i.e. both are 3-tap stride-1 convs with one structurally empty tap and 128 reduction channels when Cin = 64, Cout = 128 -- the shape
// conv3_ws_kernel keeps in registers.  wf / wd: [3][128][128] 16-bit, element (tap, n, k); w: the packed conv weight [3][Cout][Cin].
__global__ void s2ws_pack_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, int Cout, int Cin) {
  const int N = 2 * Cin;      // = Cout = 128 (checked by the launcher): both repacked weights are [3][N][N]
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * N * N) return;
  const int tp = i / (N * N), n = (i / N) % N, k = i % N;
  auto W = [&](int t, int co, int ci) { return w[((long)t * Cout + co) * Cin + ci]; };
  bf16_t f = 0, d = 0;
  // forward: n = output channel co, k = channel of the paired input row
  if (tp == 0) { if (k >= Cin) f = W(0, n, k - Cin); }
  else if (tp == 1) f = k < Cin ? W(1, n, k) : W(2, n, k - Cin);
  // data gradient: n = column of the paired dx row (ci, or Cin + ci), k = output channel co
  if (tp == 1) d = n < Cin ? W(1, k, n) : W(2, k, n - Cin);
  else if (tp == 2) { if (n >= Cin) d = W(0, k, n - Cin); }
  wf[i] = f; wd[i] = d;
}
int s2ws_pack(eegldm_ctx* ctx, const void* w, void* wf, void* wd, int Cout, int Cin) {
  EEG_CHECK(Cin == 64 && Cout == 128, "stride-2 -> weight-stationary mapping: Cin 64, Cout 128");
  const int n = 3 * 128 * 128;
  hipLaunchKernelGGL(s2ws_pack_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, ctx->stream, (const bf16_t*)w, (bf16_t*)wf, (bf16_t*)wd, Cout, Cin);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_cast(eegldm_ctx* ctx, const float* s, void* d, long n, int dtype) {
  if (n <= 0) return 0;
  DISPATCH_T(dtype, hipLaunchKernelGGL((cast_kernel<T>), dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, s, (T*)d, n));
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_fill(eegldm_ctx* ctx, float* p, long n, float v) {
  if (n <= 0) return 0;
  if (v == 0.0f) { HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(float), ctx->stream)); return 0; }
  hipLaunchKernelGGL(fill_kernel, dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, p, n, v);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_add_noise(eegldm_ctx* ctx, const float* x, const float* nz, const int64_t* t, const float* acp, float* out, int B, long per) {
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid1d((long)B * per, ctx)), dim3(NT), 0, ctx->stream, x, nz, t, acp, out, (long)B * per, per, 0);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_get_velocity(eegldm_ctx* ctx, const float* x, const float* nz, const int64_t* t, const float* acp, float* out, int B, long per) {
  hipLaunchKernelGGL(add_noise_kernel, dim3(grid1d((long)B * per, ctx)), dim3(NT), 0, ctx->stream, x, nz, t, acp, out, (long)B * per, per, 1);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_ddim_step(eegldm_ctx* ctx, const float* mo, const float* x, float a_t, float a_prev, int pred, int clip,
                                float* prev, float* x0, long n) {
  EEG_CHECK(pred >= 0 && pred <= 2, "prediction type %d", pred);
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, mo, x, a_t, a_prev, pred, clip, prev, x0, n);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_ddim_step_eta(eegldm_ctx* ctx, const float* mo, const float* x, const float* noise, float a_t, float a_prev, float eta,
                                    int pred, int clip, float* prev, float* x0, long n) {
  EEG_CHECK(ctx && mo && x && prev, "null argument");
  EEG_CHECK(pred >= 0 && pred <= 2, "prediction type %d", pred);
  EEG_CHECK(eta >= 0.0f && a_t > 0.0f && a_t < 1.0f && a_prev > 0.0f && a_prev <= 1.0f, "bad eta / schedule values");
  if (eta == 0.0f) return eegldm_ddim_step(ctx, mo, x, a_t, a_prev, pred, clip, prev, x0, n);
  EEG_CHECK(noise, "eta > 0 needs a noise tensor");
  // sigma_t(eta) and the direction coefficient in double on the host, like the schedulers' tables
  const double var = (1.0 - (double)a_prev) / (1.0 - (double)a_t) * (1.0 - (double)a_t / (double)a_prev);
  const double sigma = (double)eta * sqrt(var > 0.0 ? var : 0.0);
  const double d2 = 1.0 - (double)a_prev - sigma * sigma;
  hipLaunchKernelGGL(ddim_step_eta_kernel, dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, mo, x, noise, a_t, a_prev, (float)sigma,
                     (float)sqrt(d2 > 0.0 ? d2 : 0.0), pred, clip, prev, x0, n);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_ddpm_step(eegldm_ctx* ctx, const float* mo, const float* x, const float* noise, float a_t, float a_prev, float beta_t,
                                int pred, int clip, float* prev, float* x0, long n) {
  return eegldm_ddpm_step_var(ctx, mo, x, noise, a_t, a_prev, beta_t, 0, pred, clip, prev, x0, n);
}
// variance_large != 0: DDPMScheduler(variance_type="fixed_large"): sigma^2 = beta_t instead of the posterior variance
extern "C" int eegldm_ddpm_step_var(eegldm_ctx* ctx, const float* mo, const float* x, const float* noise, float a_t, float a_prev, float beta_t,
                                    int variance_large, int pred, int clip, float* prev, float* x0, long n) {
  EEG_CHECK(ctx && mo && x && prev, "null argument");
  EEG_CHECK(pred >= 0 && pred <= 2, "prediction type %d", pred);
  EEG_CHECK(a_t > 0.0f && a_t < 1.0f && a_prev > 0.0f && a_prev <= 1.0f && beta_t > 0.0f && beta_t < 1.0f, "bad schedule values");
  // posterior q(x_{t-1} | x_t, x_0): coefficients in double on the host, as the schedulers build their tables
  const double bt = 1.0 - (double)a_t, bp = 1.0 - (double)a_prev;
  const double c0 = sqrt((double)a_prev) * (double)beta_t / bt, ct = sqrt(1.0 - (double)beta_t) * bp / bt;
  double var = variance_large ? (double)beta_t : bp / bt * (double)beta_t;
  const bool last = a_prev >= 1.0f;                  // t == 0: no noise
  if (var < 1e-20) var = 1e-20;
  const float sigma = last ? 0.0f : (float)sqrt(var);
  EEG_CHECK(last || noise, "noise is required for t > 0");
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, mo, x, noise, (float)sqrt((double)a_t), (float)sqrt(bt),
                     (float)c0, (float)ct, sigma, pred, clip, prev, x0, n);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_mse_loss(eegldm_ctx* ctx, const float* p, const float* t, float* loss, float* dp, long n, float gscale) {
  HIP_TRY(hipMemsetAsync(loss, 0, sizeof(float), ctx->stream));
  // (deterministic mode: a written partial per block and an ordered fold instead of a race of per-block atomics)
  const int nb = grid1d(n, ctx, 4);
  float* parts = nullptr;
  if (eeg_deterministic()) EEG_TRY(eeg_det_buffer(ctx, (size_t)nb * sizeof(float), &parts));
  hipLaunchKernelGGL(mse_kernel, dim3(nb), dim3(NT), 0, ctx->stream, p, t, loss, dp, n, 1.0f / (float)n, gscale, parts);
  LAUNCH_CHECK();
  if (parts) EEG_TRY(ew_fold_partials_det(ctx, parts, nb, 1, 0, 1, loss));
  return 0;
}
extern "C" int eegldm_adam_step(eegldm_ctx* ctx, float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                                float eps, int step, float ginv) {
  EEG_CHECK(step >= 1, "step starts at 1");
  const float bc1 = 1.0f - powf(b1, (float)step), bc2s = sqrtf(1.0f - powf(b2, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(grid1d(n, ctx, 2)), dim3(NT), 0, ctx->stream, p, g, m, v, n, lr, b1, b2, eps, bc1, bc2s, ginv);
  LAUNCH_CHECK(); return 0;
}
// GradScaler.unscale_/step support (training.py:334,441-443 use torch.cuda.amp.GradScaler): found_inf[0] = 1 if any gradient is
// inf / nan, else 0.  One pass over the flat gradient buffer, 16-byte loads; the flag stays on the device until the host reads it.
__global__ void finite_check_kernel(const float* __restrict__ g, long n, float* __restrict__ found_inf) {
  bool bad = false;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = ((const float4*)g)[i];
    // (x - x) is 0 for finite x and nan for inf / nan
    bad |= ((v.x - v.x) + (v.y - v.y) + (v.z - v.z) + (v.w - v.w)) != 0.0f;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) bad |= (g[i] - g[i]) != 0.0f;
  if (__any(bad) && (threadIdx.x & 63) == 0) found_inf[0] = 1.0f;
}
extern "C" int eegldm_grad_check_finite(eegldm_ctx* ctx, const float* g, long n, float* found_inf) {
  EEG_CHECK(g && found_inf && n >= 0, "grad_check_finite: null argument");
  EEG_CHECK(((uintptr_t)g & 15) == 0, "grad_check_finite: gradient buffer must be 16-byte aligned");
  HIP_TRY(hipMemsetAsync(found_inf, 0, sizeof(float), ctx->stream));
  if (n > 0) { hipLaunchKernelGGL(finite_check_kernel, dim3(grid1d((n + 3) / 4, ctx, 4)), dim3(NT), 0, ctx->stream, g, n, found_inf); LAUNCH_CHECK(); }
  return 0;
}
extern "C" int eegldm_randn(eegldm_ctx* ctx, float* out, long n, uint64_t seed, uint64_t offset) {
  hipLaunchKernelGGL(randn_kernel, dim3(grid1d((n + 3) / 4, ctx)), dim3(NT), 0, ctx->stream, out, n, seed, offset);
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_dropout(eegldm_ctx* ctx, void* x, long ld, long rows, int C, float p, uint64_t seed, uint64_t offset, int dtype) {
  EEG_CHECK(ctx && x && rows >= 0 && C > 0 && ld >= C, "bad argument");
  if (rows == 0 || p == 0.0f) return 0;
  return ew_dropout_rows(ctx, x, ld, rows, C, p, seed, offset, dtype);
}
extern "C" int eegldm_randint(eegldm_ctx* ctx, int64_t* out, long n, int64_t high, uint64_t seed, uint64_t offset) {
  EEG_CHECK(high > 0, "high must be positive");
  hipLaunchKernelGGL(randint_kernel, dim3(grid1d(n, ctx)), dim3(NT), 0, ctx->stream, out, n, high, seed, offset);
  LAUNCH_CHECK(); return 0;
}

// ---- stand-alone AvgPool1d(2,2) / nearest x 2 (include/eegldm.h)
static int resample2(eegldm_ctx* ctx, int mode, const void* x, long ldx, void* y, long ldy, int B, int L_small, int C, int dtype) {
  EEG_CHECK(ctx && x && y && B > 0 && L_small > 0 && C > 0 && ldx >= C && ldy >= C, "bad argument");
  const long rows = (long)B * L_small;
#define RS2(M) DISPATCH_T(dtype, hipLaunchKernelGGL((resample2_kernel<T, M>), dim3(grid1d(rows * C, ctx)), dim3(NT), 0, ctx->stream, (const T*)x, ldx, (T*)y, ldy, rows, C))
  if (mode == 0) RS2(0); else if (mode == 1) RS2(1); else if (mode == 2) RS2(2); else RS2(3);
#undef RS2
  LAUNCH_CHECK(); return 0;
}
extern "C" int eegldm_avgpool2_fwd(eegldm_ctx* ctx, const void* x, long ldx, void* y, long ldy, int B, int L, int C, int dtype) {
  EEG_CHECK(L % 2 == 0, "AvgPool1d(2, 2): even length expected, got %d", L);
  return resample2(ctx, 0, x, ldx, y, ldy, B, L / 2, C, dtype);
}
extern "C" int eegldm_avgpool2_bwd(eegldm_ctx* ctx, const void* dy, long lddy, void* dx, long lddx, int B, int L, int C, int dtype) {
  EEG_CHECK(L % 2 == 0, "AvgPool1d(2, 2): even length expected, got %d", L);
  return resample2(ctx, 2, dy, lddy, dx, lddx, B, L / 2, C, dtype);
}
extern "C" int eegldm_nearest2_fwd(eegldm_ctx* ctx, const void* x, long ldx, void* y, long ldy, int B, int L, int C, int dtype) {
  return resample2(ctx, 3, x, ldx, y, ldy, B, L, C, dtype);
}
extern "C" int eegldm_nearest2_bwd(eegldm_ctx* ctx, const void* dy, long lddy, void* dx, long lddx, int B, int L, int C, int dtype) {
  return resample2(ctx, 1, dy, lddy, dx, lddx, B, L, C, dtype);
}
