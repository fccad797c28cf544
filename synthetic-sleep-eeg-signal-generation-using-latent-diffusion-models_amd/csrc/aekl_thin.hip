// Whole-network kernels for THIN AutoencoderKL configurations: ONE workgroup per window runs the entire autoencoder forward
// (encoder, heads, reparameterisation + KL, decoder) and ONE the entire backward, with every activation resident in LDS.
//
// Why: config/config_aekl_eeg_2_2_4_spec.yaml (BASELINE configs[1]) has num_channels [2,2,4] -- 934 parameters, 0.9 MMAC per window,
// activations of at most 2 x 3072 values.  As a layer-by-layer sequence it is ~300 launches per train step of 4-6 us each
// (profiles/r02_aekl_gan_step_bf16_B256_kernel_stats_v1.txt: 1.9 ms of the 5.5 ms step for 0.05 ms worth of HBM traffic): pure
// dispatch latency.  Here the host compiles the block list of the model (MONAI AutoencoderKL as configured at
// /root/reference/src/train_autoencoderkl.py:129-133; structural twin /root/reference/src/models/ae_kl.py:123-291) into a list of
// micro-ops over four LDS tensors (aekl_thin.h) and a 512-thread workgroup interprets it for its window.  All arithmetic is
// fp32 on the fp32 master parameters whatever the engine dtype (the tensors never leave the CU, so there is no storage format to
// choose); tensors needed again by the backward pass (block inputs, pre-norm activations, head outputs) go to a per-window tape
// in global memory, conv inputs that are GroupNorm+SiLU outputs are recomputed from it.  Parameter gradients are reduced inside
// the workgroup and added to the flat gradient buffer with one atomic per parameter and window.
#include "aekl_thin.h"
#include "internal.h"

#include <stdlib.h>

namespace {
#ifndef EEG_THIN_NT
#define EEG_THIN_NT 512
#endif
constexpr int NT = EEG_THIN_NT, NWAVE = NT / 64;      // threads per window (developer builds: -DEEG_THIN_NT=1024)
constexpr int MC = THIN_MAXC;
constexpr float GN_EPS_T = 1e-6f;
constexpr int RED_ROWS = NWAVE * 4;                // 16-lane rows per workgroup: one partial per row and value
constexpr int RED_FLOATS = RED_ROWS * 56 + 64;

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is fence + s_barrier and the fence waits for EVERY outstanding
// global access (s_waitcnt vmcnt(0)): the tape prefetch of the backward kernel, the tape stores of the forward kernel and the
// gradient atomics would all be waited for at the next barrier.  Nothing in these kernels reads global memory written earlier in the
// same launch (tape and statistics are produced by the forward launch and consumed by the backward launch), so LDS ordering suffices.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct BufSel { float* base; int maxt; __device__ __forceinline__ float* operator[](int i) const { return base + (size_t)i * maxt; } };
struct Bufs { BufSel b; float* red; };

// sum of NV per-thread values over the workgroup; results in red[RED_ROWS*NV + i] (valid for every thread after the call).
// Inside a 16-lane row the sum is four DPP adds (quad swaps, half-row mirror, row mirror: full-rate VALU, no LDS); the 32 row
// partials per value then meet in LDS.  The first version reduced every value over the whole wave with six `__shfl_xor` = six
// ds_bpermute round trips each: 312 LDS-pipe instructions for the 52 gradients of a 4 -> 4 conv, ~3.5 k cycles per call, 60 calls
// per window in the backward kernel.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));     // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));     // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));    // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));    // row_mirror
  return v;
}
template <int NV> __device__ __forceinline__ void block_reduce(float (&v)[NV], float* red) {
  static_assert(NV <= 56, "row-partial area holds 56 values per row");
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = row16_sum(v[i]);
  const int row = threadIdx.x >> 4;
  lds_barrier();
  if ((threadIdx.x & 15) == 0) {
#pragma unroll
    for (int i = 0; i < NV; i++) red[row * NV + i] = v[i];
  }
  lds_barrier();
  if (threadIdx.x < NV) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < RED_ROWS; w++) s += red[w * NV + threadIdx.x];
    red[RED_ROWS * NV + threadIdx.x] = s;
  }
  lds_barrier();
}

// The channel counts are compile-time (1, 2 or 4 each): the kernels are VALU-issue bound -- one 8-wave workgroup per CU, every
// wave64 VALU instruction occupies its SIMD for 4 cycles -- so a loop predicated on a run-time channel count (the first version:
// 4 x 4 x 3 predicated FMAs per position for a 2 -> 2 conv) cost 3x the instructions of the specialised one.
template <int CIN, int COUT>
__device__ __forceinline__ void load_w(const ThinOp& o, const float* P, int woff, int boff, float (&w)[COUT][CIN][3], float (&bias)[COUT]) {
#pragma unroll
  for (int co = 0; co < COUT; co++) {
    bias[co] = boff >= 0 ? P[boff + co] : 0.f;
#pragma unroll
    for (int ci = 0; ci < CIN; ci++)
#pragma unroll
      for (int k = 0; k < 3; k++) w[co][ci][k] = k < o.k ? P[woff + (k * COUT + co) * CIN + ci] : 0.f;
  }
}

// window of one channel row around 4 consecutive positions: v[0..5] = X[l0-1 .. l0+4] (zero outside the row)
__device__ __forceinline__ void row_window(const float* __restrict__ row, int l0, int L, bool halo, float (&v)[6]) {
  const float4 c = *(const float4*)(row + l0);
  v[1] = c.x; v[2] = c.y; v[3] = c.z; v[4] = c.w;
  v[0] = (halo && l0 > 0) ? row[l0 - 1] : 0.f;
  v[5] = (halo && l0 + 4 < L) ? row[l0 + 4] : 0.f;
}

// The kernels are LDS-latency / VALU-issue bound (one 8-wave workgroup per CU = 2 waves per SIMD): a scalar loop of the form
// load, wait ~100 cycles, use ran at ~400 cycles per 64 positions.  Every hot loop therefore handles FOUR consecutive positions
// per thread (ds_read_b128 + two halo words per channel row; 4 x the FMAs per LDS instruction, independent accumulators).
template <int CIN, int COUT>
__device__ void f_conv_t(const ThinOp& o, const float* P, const Bufs& B) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst]; const float* A = o.add >= 0 ? B.b[o.add] : nullptr;
  float w[COUT][CIN][3], bias[COUT];
  load_w<CIN, COUT>(o, P, o.w, o.b, w, bias);
  const int Lin = o.Lin, Lout = o.Lout, stride = o.stride, nk = o.k, pad = o.pad_l;
  if (stride == 1 && ((nk == 3 && pad == 1) || (nk == 1 && pad == 0)) && (Lout & 3) == 0) {
    const bool k3 = nk == 3;
    for (int q = threadIdx.x; q < (Lout >> 2); q += NT) {
      const int l0 = q << 2;
      float acc[COUT][4];
#pragma unroll
      for (int co = 0; co < COUT; co++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[co][j] = bias[co];
#pragma unroll
      for (int ci = 0; ci < CIN; ci++) {
        float v[6];
        row_window(X + ci * Lin, l0, Lin, k3, v);
#pragma unroll
        for (int co = 0; co < COUT; co++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (k3) { acc[co][j] = fmaf(w[co][ci][0], v[j], acc[co][j]); acc[co][j] = fmaf(w[co][ci][1], v[j + 1], acc[co][j]); acc[co][j] = fmaf(w[co][ci][2], v[j + 2], acc[co][j]); }
            else acc[co][j] = fmaf(w[co][ci][0], v[j + 1], acc[co][j]);
          }
      }
#pragma unroll
      for (int co = 0; co < COUT; co++) {
        float4 r = make_float4(acc[co][0], acc[co][1], acc[co][2], acc[co][3]);
        if (A) { const float4 a = *(const float4*)(A + co * Lout + l0); r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w; }
        *(float4*)(Y + co * Lout + l0) = r;
      }
    }
    lds_barrier();
    return;
  }
  for (int lo = threadIdx.x; lo < Lout; lo += NT) {
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; co++) acc[co] = bias[co];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int li = lo * stride + k - pad;
      if (k < nk && li >= 0 && li < Lin) {
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) {
          const float xv = X[ci * Lin + li];
#pragma unroll
          for (int co = 0; co < COUT; co++) acc[co] = fmaf(w[co][ci][k], xv, acc[co]);
        }
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; co++) Y[co * Lout + lo] = acc[co] + (A ? A[co * Lout + lo] : 0.f);
  }
  lds_barrier();
}
#define THIN_CC(fn, ...)                                                                      \
  switch (o.cin * 8 + o.cout) {                                                               \
    case 1 * 8 + 1: fn<1, 1>(__VA_ARGS__); break; case 1 * 8 + 2: fn<1, 2>(__VA_ARGS__); break; case 1 * 8 + 4: fn<1, 4>(__VA_ARGS__); break; \
    case 2 * 8 + 1: fn<2, 1>(__VA_ARGS__); break; case 2 * 8 + 2: fn<2, 2>(__VA_ARGS__); break; case 2 * 8 + 4: fn<2, 4>(__VA_ARGS__); break; \
    case 4 * 8 + 1: fn<4, 1>(__VA_ARGS__); break; case 4 * 8 + 2: fn<4, 2>(__VA_ARGS__); break; case 4 * 8 + 4: fn<4, 4>(__VA_ARGS__); break; \
    default: break;                                                                           \
  }
__device__ void f_conv(const ThinOp& o, const float* P, const Bufs& B) { THIN_CC(f_conv_t, o, P, B) }

// channel parameter of element group q (4 consecutive elements never straddle a channel row: L % 4 == 0)
template <int C> __device__ __forceinline__ float sel(const float (&a)[C], int c) {
  float r = a[0];
#pragma unroll
  for (int i = 1; i < C; i++) r = c == i ? a[i] : r;
  return r;
}
template <int C>
__device__ __forceinline__ void gn_apply_t(const ThinOp& o, const float* P, const float* X, float* Y, float mean, float rstd) {
  float g[C], bb[C];
#pragma unroll
  for (int c = 0; c < C; c++) { g[c] = P[o.gw + c] * rstd; bb[c] = P[o.gb + c] - mean * g[c]; }
  const int L = o.Lin, nq = (C * L) >> 2; const bool silu = o.silu != 0;
  for (int q = threadIdx.x; q < nq; q += NT) {
    const int c = (q << 2) / L;
    const float gc = sel<C>(g, c), bc = sel<C>(bb, c);
    const float4 x = *(const float4*)(X + (q << 2));
    float4 z = make_float4(fmaf(x.x, gc, bc), fmaf(x.y, gc, bc), fmaf(x.z, gc, bc), fmaf(x.w, gc, bc));
    if (silu) { z.x = silu_f(z.x); z.y = silu_f(z.y); z.z = silu_f(z.z); z.w = silu_f(z.w); }
    *(float4*)(Y + (q << 2)) = z;
  }
  lds_barrier();
}
template <int C>
__device__ void f_gn_t(const ThinOp& o, const float* P, const Bufs& B, float* __restrict__ stats) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst];
  const int n = C * o.Lin, nq = n >> 2;
  // one pass: sums of (x - K) and (x - K)^2 with K = the tensor's first element (removes the cancellation of E[x^2] - mean^2)
  const float K = X[0];
  float sq[2] = {0.f, 0.f};
  for (int q = threadIdx.x; q < nq; q += NT) {
    const float4 x = *(const float4*)(X + (q << 2));
    const float a = x.x - K, b = x.y - K, c = x.z - K, d = x.w - K;
    sq[0] += (a + b) + (c + d);
    sq[1] = fmaf(a, a, fmaf(b, b, fmaf(c, c, fmaf(d, d, sq[1]))));
  }
  block_reduce<2>(sq, B.red);
  const float m1 = B.red[RED_ROWS * 2] / (float)n, m2 = B.red[RED_ROWS * 2 + 1] / (float)n;
  const float mean = K + m1, rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + GN_EPS_T);
  if (threadIdx.x == 0) { stats[2 * o.stat] = mean; stats[2 * o.stat + 1] = rstd; }
  gn_apply_t<C>(o, P, X, Y, mean, rstd);
}
__device__ void f_gn(const ThinOp& o, const float* P, const Bufs& B, float* stats) {
  if (o.cin == 1) f_gn_t<1>(o, P, B, stats); else if (o.cin == 2) f_gn_t<2>(o, P, B, stats); else if (o.cin == 4) f_gn_t<4>(o, P, B, stats);
}
// GroupNorm apply with saved statistics (backward: recompute a conv's input)
__device__ void b_recomp(const ThinOp& o, const float* P, const Bufs& B, const float* __restrict__ stats) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst];
  const float mean = stats[2 * o.stat], rstd = stats[2 * o.stat + 1];
  if (o.cin == 1) gn_apply_t<1>(o, P, X, Y, mean, rstd); else if (o.cin == 2) gn_apply_t<2>(o, P, X, Y, mean, rstd); else if (o.cin == 4) gn_apply_t<4>(o, P, X, Y, mean, rstd);
}

__device__ void d_ups(const ThinOp& o, const Bufs& B) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst];
  for (int i = threadIdx.x; i < o.cin * o.Lout; i += NT) { const int c = i / o.Lout, l = i - c * o.Lout; Y[i] = X[c * o.Lin + (l >> 1)]; }
  lds_barrier();
}

template <int LAT>
__device__ void f_heads_t(const ThinOp& o, const float* P, const Bufs& B, float* __restrict__ tape, const int* __restrict__ tape_off,
                          const float* __restrict__ eps, float* __restrict__ z_mu, float* __restrict__ z_sigma, float* __restrict__ kl, float inv_B) {
  const float* H = B.b[o.src]; float* Z = B.b[o.dst];
  const int L = o.Lin;
  float* tmu = tape + tape_off[o.save]; float* tlv = tape + tape_off[o.save + 1];
  float part[1] = {0.f};
  for (int l = threadIdx.x; l < L; l += NT) {
    float h[LAT];
#pragma unroll
    for (int ci = 0; ci < LAT; ci++) h[ci] = H[ci * L + l];
#pragma unroll
    for (int co = 0; co < LAT; co++) {
      float mu = P[o.b + co], lv = P[o.b2 + co];
#pragma unroll
      for (int ci = 0; ci < LAT; ci++) { mu = fmaf(P[o.w + co * LAT + ci], h[ci], mu); lv = fmaf(P[o.w2 + co * LAT + ci], h[ci], lv); }
      const float lvc = fminf(20.f, fmaxf(-30.f, lv));
      const float sg = __expf(0.5f * lvc);
      const float e = eps ? eps[co * L + l] : 0.f;
      Z[co * L + l] = fmaf(e, sg, mu);
      tmu[co * L + l] = mu; tlv[co * L + l] = lv;
      if (z_mu) z_mu[co * L + l] = mu;
      if (z_sigma) z_sigma[co * L + l] = sg;
      part[0] += 0.5f * (mu * mu + sg * sg - lvc - 1.0f);
    }
  }
  block_reduce<1>(part, B.red);
  if (kl && threadIdx.x == 0) atomicAdd(kl, B.red[RED_ROWS] * inv_B);
}
__device__ void f_heads(const ThinOp& o, const float* P, const Bufs& B, float* tape, const int* tape_off, const float* eps, float* z_mu, float* z_sigma,
                        float* kl, float inv_B) {
  if (o.cin == 1) f_heads_t<1>(o, P, B, tape, tape_off, eps, z_mu, z_sigma, kl, inv_B);
  else if (o.cin == 2) f_heads_t<2>(o, P, B, tape, tape_off, eps, z_mu, z_sigma, kl, inv_B);
  else if (o.cin == 4) f_heads_t<4>(o, P, B, tape, tape_off, eps, z_mu, z_sigma, kl, inv_B);
}

// ------------------------------------------------------------------ backward pieces
template <int CIN, int COUT>
__device__ void b_conv_t(const ThinOp& o, const float* P, float* __restrict__ G, const Bufs& B) {
  const float* dY = B.b[o.src]; float* A = B.b[o.act];
  float w[COUT][CIN][3], bias[COUT];
  load_w<CIN, COUT>(o, P, o.w, -1, w, bias);
  const int Lin = o.Lin, Lout = o.Lout, stride = o.stride, nk = o.k, pad = o.pad_l;
  const bool vec = stride == 1 && ((nk == 3 && pad == 1) || (nk == 1 && pad == 0)) && (Lout & 3) == 0;
  const bool k3 = nk == 3;
  // ---- dW / db
  constexpr int NW_ = COUT * CIN * 3, NV = NW_ + COUT;
  float acc[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) acc[i] = 0.f;
  if (vec) {
    for (int q = threadIdx.x; q < (Lout >> 2); q += NT) {
      const int l0 = q << 2;
      float dy[COUT][4];
#pragma unroll
      for (int co = 0; co < COUT; co++) {
        const float4 d = *(const float4*)(dY + co * Lout + l0);
        dy[co][0] = d.x; dy[co][1] = d.y; dy[co][2] = d.z; dy[co][3] = d.w;
        acc[NW_ + co] += (d.x + d.y) + (d.z + d.w);
      }
#pragma unroll
      for (int ci = 0; ci < CIN; ci++) {
        float v[6];
        row_window(A + ci * Lin, l0, Lin, k3, v);
#pragma unroll
        for (int co = 0; co < COUT; co++)
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (k3) {
#pragma unroll
              for (int k = 0; k < 3; k++) acc[(co * CIN + ci) * 3 + k] = fmaf(dy[co][j], v[j + k], acc[(co * CIN + ci) * 3 + k]);
            } else acc[(co * CIN + ci) * 3] = fmaf(dy[co][j], v[j + 1], acc[(co * CIN + ci) * 3]);
          }
      }
    }
  } else {
    for (int lo = threadIdx.x; lo < Lout; lo += NT) {
      float dy[COUT];
#pragma unroll
      for (int co = 0; co < COUT; co++) { dy[co] = dY[co * Lout + lo]; acc[NW_ + co] += dy[co]; }
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int li = lo * stride + k - pad;
        if (k < nk && li >= 0 && li < Lin) {
#pragma unroll
          for (int ci = 0; ci < CIN; ci++) {
            const float xv = A[ci * Lin + li];
#pragma unroll
            for (int co = 0; co < COUT; co++) acc[(co * CIN + ci) * 3 + k] = fmaf(dy[co], xv, acc[(co * CIN + ci) * 3 + k]);
          }
        }
      }
    }
  }
  block_reduce<NV>(acc, B.red);       // ends with a barrier: every read of A above is done
  const float* R = B.red + RED_ROWS * NV;
  if (threadIdx.x < NW_) {
    const int co = threadIdx.x / (CIN * 3), ci = (threadIdx.x / 3) % CIN, k = threadIdx.x % 3;
    if (k < nk) G[o.w + (k * COUT + co) * CIN + ci] += R[threadIdx.x];
  } else if (threadIdx.x < NV) {
    if (o.b >= 0) G[o.b + (threadIdx.x - NW_)] += R[threadIdx.x];
  }
  // ---- dX -> overwrites the activation buffer
  if (o.need_dx) {
    if (vec) {
      // dX[ci][l] = sum_co sum_k w[co][ci][k] dY[co][l + 1 - k]  (k = 3, pad 1)  |  sum_co w[co][ci][0] dY[co][l]  (k = 1)
      for (int q = threadIdx.x; q < (Lin >> 2); q += NT) {
        const int l0 = q << 2;
        float dx[CIN][4];
#pragma unroll
        for (int ci = 0; ci < CIN; ci++)
#pragma unroll
          for (int j = 0; j < 4; j++) dx[ci][j] = 0.f;
#pragma unroll
        for (int co = 0; co < COUT; co++) {
          float v[6];
          row_window(dY + co * Lout, l0, Lout, k3, v);
#pragma unroll
          for (int ci = 0; ci < CIN; ci++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (k3) { dx[ci][j] = fmaf(w[co][ci][0], v[j + 2], dx[ci][j]); dx[ci][j] = fmaf(w[co][ci][1], v[j + 1], dx[ci][j]); dx[ci][j] = fmaf(w[co][ci][2], v[j], dx[ci][j]); }
              else dx[ci][j] = fmaf(w[co][ci][0], v[j + 1], dx[ci][j]);
            }
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) *(float4*)(A + ci * Lin + l0) = make_float4(dx[ci][0], dx[ci][1], dx[ci][2], dx[ci][3]);
      }
    } else {
      for (int li = threadIdx.x; li < Lin; li += NT) {
        float dx[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) dx[ci] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; k++) {
          const int t = li + pad - k;
          if (k < nk && t >= 0 && (stride == 1 || (t & 1) == 0)) {
            const int lo = stride == 1 ? t : t >> 1;
            if (lo < Lout) {
#pragma unroll
              for (int co = 0; co < COUT; co++) {
                const float dy = dY[co * Lout + lo];
#pragma unroll
                for (int ci = 0; ci < CIN; ci++) dx[ci] = fmaf(w[co][ci][k], dy, dx[ci]);
              }
            }
          }
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ci++) A[ci * Lin + li] = dx[ci];
      }
    }
  }
  lds_barrier();
}
__device__ void b_conv(const ThinOp& o, const float* P, float* G, const Bufs& B) { THIN_CC(b_conv_t, o, P, G, B) }

template <int C>
__device__ void b_gn_t(const ThinOp& o, const float* P, float* __restrict__ G, const Bufs& B, const float* __restrict__ stats) {
  const float* X = B.b[o.act]; const float* dY = B.b[o.src]; float* dX = B.b[o.dst]; const float* ADD = o.add >= 0 ? B.b[o.add] : nullptr;
  const float mean = stats[2 * o.stat], rstd = stats[2 * o.stat + 1];
  const int L = o.Lin, nq = (C * L) >> 2; const bool silu = o.silu != 0;
  float acc[2 * C];
#pragma unroll
  for (int i = 0; i < 2 * C; i++) acc[i] = 0.f;
  float ga[C], be[C];
#pragma unroll
  for (int c = 0; c < C; c++) { ga[c] = P[o.gw + c]; be[c] = P[o.gb + c]; }
  const float nmr = -mean * rstd;
  for (int q = threadIdx.x; q < nq; q += NT) {
    const int c = (q << 2) / L;
    const float gc = sel<C>(ga, c), bc = sel<C>(be, c);
    const float4 x = *(const float4*)(X + (q << 2)), d = *(const float4*)(dY + (q << 2));
    const float xs[4] = {x.x, x.y, x.z, x.w}, ds[4] = {d.x, d.y, d.z, d.w};
    float sg = 0.f, sb = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float xh = fmaf(xs[j], rstd, nmr);
      const float dz = ds[j] * (silu ? silu_grad_f(fmaf(xh, gc, bc)) : 1.0f);
      sg = fmaf(dz, xh, sg); sb += dz;
    }
#pragma unroll
    for (int i = 0; i < C; i++) { acc[i] += c == i ? sg : 0.f; acc[C + i] += c == i ? sb : 0.f; }
  }
  block_reduce<2 * C>(acc, B.red);
  const float* R = B.red + RED_ROWS * 2 * C;
  if (threadIdx.x < C) G[o.gw + threadIdx.x] += R[threadIdx.x];
  else if (threadIdx.x < 2 * C) G[o.gb + threadIdx.x - C] += R[threadIdx.x];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < C; c++) { s1 = fmaf(ga[c], R[C + c], s1); s2 = fmaf(ga[c], R[c], s2); }
  const float inv_n = 1.0f / (float)(C * L);
  const float m1 = s1 * inv_n, m2 = s2 * inv_n;
  for (int q = threadIdx.x; q < nq; q += NT) {
    const int c = (q << 2) / L;
    const float gc = sel<C>(ga, c), bc = sel<C>(be, c);
    const float4 x = *(const float4*)(X + (q << 2)), d = *(const float4*)(dY + (q << 2));
    const float xs[4] = {x.x, x.y, x.z, x.w}, ds[4] = {d.x, d.y, d.z, d.w};
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const float xh = fmaf(xs[j], rstd, nmr);
      const float dz = ds[j] * (silu ? silu_grad_f(fmaf(xh, gc, bc)) : 1.0f);
      r[j] = rstd * (dz * gc - m1 - xh * m2);
    }
    float4 out = make_float4(r[0], r[1], r[2], r[3]);
    if (ADD) { const float4 a = *(const float4*)(ADD + (q << 2)); out.x += a.x; out.y += a.y; out.z += a.z; out.w += a.w; }
    *(float4*)(dX + (q << 2)) = out;
  }
  lds_barrier();
}
__device__ void b_gn(const ThinOp& o, const float* P, float* G, const Bufs& B, const float* stats) {
  if (o.cin == 1) b_gn_t<1>(o, P, G, B, stats); else if (o.cin == 2) b_gn_t<2>(o, P, G, B, stats); else if (o.cin == 4) b_gn_t<4>(o, P, G, B, stats);
}

template <int LAT>
__device__ void b_heads_t(const ThinOp& o, const float* P, float* __restrict__ G, const Bufs& B, const float* __restrict__ tape,
                          const int* __restrict__ tape_off, const float* __restrict__ eps, float klw_over_B,
                          const float* __restrict__ dme, const float* __restrict__ dse) {      // dme / dse: external gradients on z_mu / z_sigma of this window (nullable)
  float* DZ = B.b[o.src]; float* DH = B.b[o.dst]; const float* H = B.b[o.act];
  const int L = o.Lin;
  const float* tmu = tape + tape_off[o.save]; const float* tlv = tape + tape_off[o.save + 1];
  constexpr int NV = 2 * (LAT * LAT + LAT);
  float acc[NV];            // [dWmu co][ci], [dbmu], [dWlv], [dblv]
#pragma unroll
  for (int i = 0; i < NV; i++) acc[i] = 0.f;
  for (int l = threadIdx.x; l < L; l += NT) {
    float h[LAT], dmu[LAT], dlv[LAT], dh[LAT];
#pragma unroll
    for (int c = 0; c < LAT; c++) { h[c] = H[c * L + l]; dh[c] = 0.f; }
#pragma unroll
    for (int co = 0; co < LAT; co++) {
      const float dz = DZ[co * L + l], mu = tmu[co * L + l], lv = tlv[co * L + l];
      const bool inside = lv > -30.f && lv < 20.f;             // clamp passes the gradient strictly inside (torch.clamp)
      const float lvc = fminf(20.f, fmaxf(-30.f, lv));
      const float sg = __expf(0.5f * lvc);
      const float e = eps ? eps[co * L + l] : 0.f;
      dmu[co] = fmaf(klw_over_B, mu, dz) + (dme ? dme[co * L + l] : 0.f);       // z = mu + eps sigma ; KL: d/dmu = mu
      // d/dlv: z -> dz * eps * sigma / 2 ; KL 0.5 (sigma^2 - lv - 1) -> 0.5 (sigma^2 - 1) ; external d/dsigma -> * sigma / 2
      dlv[co] = inside ? fmaf(dz * e + (dse ? dse[co * L + l] : 0.f), 0.5f * sg, klw_over_B * 0.5f * (sg * sg - 1.0f)) : 0.f;
    }
#pragma unroll
    for (int co = 0; co < LAT; co++) {
      acc[LAT * LAT + co] += dmu[co]; acc[LAT * LAT + LAT + LAT * LAT + co] += dlv[co];
#pragma unroll
      for (int ci = 0; ci < LAT; ci++) {
        acc[co * LAT + ci] = fmaf(dmu[co], h[ci], acc[co * LAT + ci]);
        acc[LAT * LAT + LAT + co * LAT + ci] = fmaf(dlv[co], h[ci], acc[LAT * LAT + LAT + co * LAT + ci]);
        dh[ci] = fmaf(P[o.w + co * LAT + ci], dmu[co], dh[ci]);
        dh[ci] = fmaf(P[o.w2 + co * LAT + ci], dlv[co], dh[ci]);
      }
    }
#pragma unroll
    for (int ci = 0; ci < LAT; ci++) DH[ci * L + l] = dh[ci];
  }
  block_reduce<NV>(acc, B.red);
  const float* R = B.red + RED_ROWS * NV;
  const int t = threadIdx.x;
  if (t < LAT * LAT) G[o.w + t] += R[t];
  else if (t < LAT * LAT + LAT) G[o.b + (t - LAT * LAT)] += R[t];
  else if (t < 2 * LAT * LAT + LAT) G[o.w2 + (t - LAT * LAT - LAT)] += R[t];
  else if (t < NV) G[o.b2 + (t - 2 * LAT * LAT - LAT)] += R[t];
  lds_barrier();
}
__device__ void b_heads(const ThinOp& o, const float* P, float* G, const Bufs& B, const float* tape, const int* tape_off, const float* eps, float klw_over_B,
                        const float* dme, const float* dse) {
  if (o.cin == 1) b_heads_t<1>(o, P, G, B, tape, tape_off, eps, klw_over_B, dme, dse);
  else if (o.cin == 2) b_heads_t<2>(o, P, G, B, tape, tape_off, eps, klw_over_B, dme, dse);
  else if (o.cin == 4) b_heads_t<4>(o, P, G, B, tape, tape_off, eps, klw_over_B, dme, dse);
}

__device__ __forceinline__ Bufs make_bufs(char* smem, int maxt) {
  Bufs B;
  B.b.base = (float*)smem; B.b.maxt = maxt;
  B.red = (float*)smem + (size_t)THIN_NBUF * maxt;
  return B;
}

__global__ __launch_bounds__(NT) void thin_fwd_kernel(const ThinOp* ops, int nops, const int* __restrict__ tape_off, int tape_stride, int nstat,
                                                      int maxt, const float* P, const float* __restrict__ x, const float* __restrict__ eps,
                                                      float* __restrict__ recon, float* __restrict__ z_mu, float* __restrict__ z_sigma, float* __restrict__ kl,
                                                      float* __restrict__ tape_all, float* __restrict__ stats_all, int lat, int Ll, float inv_B, int nparams, unsigned long long* __restrict__ prof,
                                                      int kl_stride) {      // deterministic mode: 1 = a KL partial per window (folded in order afterwards), else 0
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Bufs B = make_bufs(smem, maxt);
  // the whole parameter set (934 values for [2,2,4]) is copied to LDS once: every micro-op starts by reading its weights, and a
  // dependent global load there exposed ~2 us of latency per op (90 ops: most of the first version's 310 us)
  float* PL = B.red + RED_FLOATS;
  for (int j = threadIdx.x; j < nparams; j += NT) PL[j] = P[j];
  // ... and so is the micro-op table (each op used to start with a dependent scalar load of its descriptor)
  int* OL = (int*)(PL + ((nparams + 3) & ~3));
  for (int j = threadIdx.x; j < nops * (int)(sizeof(ThinOp) / 4); j += NT) OL[j] = ((const int*)ops)[j];
  lds_barrier();
  P = PL; ops = (const ThinOp*)OL;
  const int b = blockIdx.x;
  float* tape = tape_all + (size_t)b * tape_stride; float* stats = stats_all + (size_t)b * nstat * 2;
  for (int i = 0; i < nops; i++) {
    const ThinOp o = ops[i];
    const unsigned long long t0 = prof ? __builtin_readcyclecounter() : 0ull;
    switch (o.kind) {
      case TF_LOAD: {
        const float* src = x + (size_t)b * o.cin * o.Lin; float* D = B.b[o.dst];
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = src[j];
        lds_barrier();
      } break;
      case TF_CONV: f_conv(o, P, B); break;
      case TF_GN: f_gn(o, P, B, stats); break;
      case TF_UPS: d_ups(o, B); break;
      case TF_SAVE: {
        const float4* S = (const float4*)B.b[o.src]; float4* D = (float4*)(tape + tape_off[o.save]);
        for (int j = threadIdx.x; j < (o.cin * o.Lin) >> 2; j += NT) D[j] = S[j];
        lds_barrier();       // the next op may overwrite the saved buffer
      } break;
      case TF_HEADS:
        f_heads(o, P, B, tape, tape_off, eps ? eps + (size_t)b * lat * Ll : nullptr, z_mu ? z_mu + (size_t)b * lat * Ll : nullptr,
                z_sigma ? z_sigma + (size_t)b * lat * Ll : nullptr, kl ? kl + (size_t)b * kl_stride : nullptr, inv_B);
        break;
      case TF_STORE: {
        const float* S = B.b[o.src]; float* D = recon + (size_t)b * o.cin * o.Lin;
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = S[j];
      } break;
    }
    if (prof && blockIdx.x == 0 && threadIdx.x == 0) prof[i] = __builtin_readcyclecounter() - t0;
  }
}

__global__ __launch_bounds__(NT) void thin_bwd_kernel(const ThinOp* ops, int nops, const int* __restrict__ tape_off, int tape_stride, int nstat,
                                                      int maxt, const float* P, float* __restrict__ G, const float* __restrict__ d_recon,
                                                      const float* __restrict__ eps, float* __restrict__ dx_out, const float* __restrict__ tape_all,
                                                      const float* __restrict__ stats_all, int lat, int Ll, float klw_over_B, int nparams, unsigned long long* __restrict__ prof,
                                                      int sp_off, int nspare, const float* __restrict__ dmu_ext, const float* __restrict__ dsg_ext,
                                                      int g_stride) {       // deterministic mode: nparams = a gradient row per window (zeroed, folded in order afterwards), else 0
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Bufs B = make_bufs(smem, maxt);
  float* PL = B.red + RED_FLOATS;
  for (int j = threadIdx.x; j < nparams; j += NT) PL[j] = P[j];
  // The window's parameter gradients accumulate in LDS (every parameter is produced by exactly one thread of one op) and go out as
  // ONE pass of global atomics at the end: issued per op (round 2) they were in flight -- ~2 us each -- whenever the next tape load
  // needed `s_waitcnt vmcnt(0)` for its LDS-DMA, and the whole workgroup waited at that op's barrier for wave 0's atomics.
  float* GL = PL + ((nparams + 3) & ~3);
  for (int j = threadIdx.x; j < nparams; j += NT) GL[j] = 0.f;
  // the window's GroupNorm statistics (written by the forward launch) also come to LDS once: read from global at the top of every
  // GroupNorm-backward / recompute op they were a dependent ~2 us round trip in front of 50 ops
  float* SL = GL + ((nparams + 3) & ~3);
  for (int j = threadIdx.x; j < 2 * nstat; j += NT) SL[j] = stats_all[(size_t)blockIdx.x * nstat * 2 + j];
  int* OL = (int*)(SL + ((2 * nstat + 3) & ~3));
  for (int j = threadIdx.x; j < nops * (int)(sizeof(ThinOp) / 4); j += NT) OL[j] = ((const int*)ops)[j];
  lds_barrier();
  P = PL; ops = (const ThinOp*)OL;
  float* const Gglobal = G; G = GL;
  const int b = blockIdx.x;
  const float* tape = tape_all + (size_t)b * tape_stride; const float* stats = SL;
  // Tape prefetch by LDS-DMA into spare LDS tensors (round 3).  A tape load is a pure round trip (~2.3 us: only 256 workgroups of
  // 512 threads run) and cost 7 200 cycles on average, 38 of them = 21 % of the kernel (EEGLDM_THIN_PROF=1) -- the round-2 prefetch
  // into REGISTERS did not survive the calls to the op functions (values live across a call are saved to scratch, which first
  // waits for the load).  global_load_lds needs no registers: the next `nspare` (<= 2) tape tensors stream into spare LDS buffers
  // while the ops before them run; the TB_LOADT op then waits (vmcnt(0): also the one other prefetch in flight, issued earlier),
  // and copies LDS -> LDS.
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  const int sp_stride = (maxt + 256) * 4;                          // bytes per spare tensor (whole 1 KB DMA instructions)
  int pf0 = -1, pf1 = -1;                                        // op index whose tape tensor spare slot 0 / 1 holds (or is receiving)
  auto next_loadt = [&](int j) __attribute__((always_inline)) {
    while (j < nops && !(ops[j].kind == TB_LOADT && ops[j].cin * ops[j].Lin >= 4 && ops[j].cin * ops[j].Lin <= maxt)) j++;
    return j;
  };
  auto dma_tape = [&](const int slot, const int j) __attribute__((always_inline)) {
    const int nq = (ops[j].cin * ops[j].Lin) >> 2;                 // 16-byte chunks
    const float4* src = (const float4*)(tape + tape_off[ops[j].save]);
    const unsigned dst = lds_base + (unsigned)sp_off + (unsigned)(slot * sp_stride);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int ii = wave; ii * 64 < nq; ii += NWAVE) {
      const int q = ii * 64 + lane;
      const unsigned off_s = __builtin_amdgcn_readfirstlane(dst + (unsigned)ii * 1024u);
      const float4* g = src + (q < nq ? q : nq - 1);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(off_s), "v"(g) : "memory", "m0");
    }
  };
  if (nspare > 0) {
    const int j0 = next_loadt(0);
    if (j0 < nops) { dma_tape(0, j0); pf0 = j0; if (nspare > 1) { const int j1 = next_loadt(j0 + 1); if (j1 < nops) { dma_tape(1, j1); pf1 = j1; } } }
  }
  for (int i = 0; i < nops; i++) {
    const ThinOp o = ops[i];
    const unsigned long long t0 = prof ? __builtin_readcyclecounter() : 0ull;
    switch (o.kind) {
      case TB_LOADDY: {
        const float* src = d_recon + (size_t)b * o.cin * o.Lin; float* D = B.b[o.dst];
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = src[j];
        lds_barrier();
      } break;
      case TB_LOADT: {
        float4* D = (float4*)B.b[o.dst];
        const int nq = (o.cin * o.Lin) >> 2;
        if (pf0 == i || pf1 == i) {
          const int slot = pf0 == i ? 0 : 1;
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA instructions have landed ...
          lds_barrier();                                         // ... and so have everybody else's
          const float4* S = (const float4*)(smem + sp_off + slot * sp_stride);
          for (int j = threadIdx.x; j < nq; j += NT) D[j] = S[j];
          lds_barrier();                                         // the spare tensor is free again: refill it with the load after the other slot's
          const int other = slot == 0 ? pf1 : pf0;
          const int jn = next_loadt((other > i ? other : i) + 1);
          const int nv = jn < nops ? jn : -1;
          if (slot == 0) pf0 = nv; else pf1 = nv;
          if (jn < nops && (slot == 0 || nspare > 1)) dma_tape(slot, jn);
        } else {
          const float4* src = (const float4*)(tape + tape_off[o.save]);
          for (int j = threadIdx.x; j < nq; j += NT) D[j] = src[j];
        }
        lds_barrier();
      } break;
      case TB_RECOMP: b_recomp(o, P, B, stats); break;
      case TB_UPS: d_ups(o, B); break;
      case TB_CONV: b_conv(o, P, G, B); break;
      case TB_GN: b_gn(o, P, G, B, stats); break;
      case TB_UPSBWD: {
        const float* S = B.b[o.src]; float* D = B.b[o.dst];      // src: cin x Lin (= 2 Lout), dst: cin x Lout
        for (int j = threadIdx.x; j < o.cin * o.Lout; j += NT) { const int c = j / o.Lout, l = j - c * o.Lout; D[j] = S[c * o.Lin + 2 * l] + S[c * o.Lin + 2 * l + 1]; }
        lds_barrier();
      } break;
      case TB_COPY: {
        const float4* S = (const float4*)B.b[o.src]; float4* D = (float4*)B.b[o.dst];
        for (int j = threadIdx.x; j < (o.cin * o.Lin) >> 2; j += NT) D[j] = S[j];
        lds_barrier();
      } break;
      case TB_HEADS: b_heads(o, P, G, B, tape, tape_off, eps ? eps + (size_t)b * lat * Ll : nullptr, klw_over_B,
                             dmu_ext ? dmu_ext + (size_t)b * lat * Ll : nullptr, dsg_ext ? dsg_ext + (size_t)b * lat * Ll : nullptr); break;
      case TB_STOREDX: {
        if (dx_out) {
          const float* S = B.b[o.src]; float* D = dx_out + (size_t)b * o.cin * o.Lin;
          for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = S[j];
        }
      } break;
    }
    if (prof && blockIdx.x == 0 && threadIdx.x == 0) prof[i] = __builtin_readcyclecounter() - t0;
  }
  lds_barrier();
  for (int j = threadIdx.x; j < nparams; j += NT) atomicAdd(Gglobal + (size_t)blockIdx.x * g_stride + j, GL[j]);
}

size_t lds_bytes(const ThinProgram& p) {
  const size_t nops = p.fwd.size() > p.bwd.size() ? p.fwd.size() : p.bwd.size();
  return sizeof(float) * ((size_t)THIN_NBUF * p.maxt + RED_FLOATS + 2 * (size_t)p.nparams + 2 * (size_t)p.nstat + 72) + nops * sizeof(ThinOp);   // (2 x: parameters, and the backward kernel's gradient accumulators)
}
// backward kernel: up to two spare LDS tensors behind everything else for the tape prefetch (as many as the 160 KB allow)
size_t lds_bytes_bwd(const ThinProgram& p, int* sp_off, int* nspare) {
  const size_t base = (lds_bytes(p) + 15) & ~(size_t)15, one = sizeof(float) * ((size_t)p.maxt + 256);
  int n = 2;
  while (n > 0 && base + n * one > 160 * 1024) n--;
  *sp_off = (int)base; *nspare = n;
  return base + n * one;
}
}  // namespace

// developer aid (EEGLDM_THIN_PROF=1): per-op shader cycles of workgroup 0, summed per op kind, printed after every launch
static unsigned long long* prof_buf() {
  EEG_ENV_VAR(bool, on, getenv("EEGLDM_THIN_PROF") != nullptr);
  static unsigned long long* buf = nullptr;
  if (on && !buf) (void)hipMalloc(&buf, 8 * 1024);
  return on ? buf : nullptr;
}
static void prof_dump(eegldm_ctx* ctx, const std::vector<ThinOp>& ops, unsigned long long* prof, const char* tag) {
  (void)hipStreamSynchronize(ctx->stream);
  std::vector<unsigned long long> h(ops.size());
  (void)hipMemcpy(h.data(), prof, 8 * ops.size(), hipMemcpyDeviceToHost);
  unsigned long long kind[16] = {0}, tot = 0; int cnt[16] = {0};
  for (size_t i = 0; i < ops.size(); i++) { kind[ops[i].kind] += h[i]; cnt[ops[i].kind]++; tot += h[i]; }
  fprintf(stderr, "thin %s: %zu ops, %llu cycles;", tag, ops.size(), tot);
  for (int k = 0; k < 16; k++) if (cnt[k]) fprintf(stderr, " kind%d x%d: %llu", k, cnt[k], kind[k]);
  fprintf(stderr, "\n");
}

int thin_upload(ThinProgram* p) {
  EEG_CHECK(p->maxt > 0 && p->maxt <= THIN_MAX_FLOATS, "thin autoencoder: tensor of %d values does not fit the LDS buffers", p->maxt);
  EEG_CHECK(p->nparams > 0 && lds_bytes(*p) <= 160 * 1024, "thin autoencoder: %d parameters + tensors exceed the LDS", p->nparams);
  thin_free(p);
  HIP_TRY(hipMalloc(&p->d_fwd, sizeof(ThinOp) * p->fwd.size()));
  HIP_TRY(hipMalloc(&p->d_bwd, sizeof(ThinOp) * p->bwd.size()));
  HIP_TRY(hipMalloc(&p->d_tape_off, sizeof(int) * p->tape_off.size()));
  HIP_TRY(hipMemcpy(p->d_fwd, p->fwd.data(), sizeof(ThinOp) * p->fwd.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_bwd, p->bwd.data(), sizeof(ThinOp) * p->bwd.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_tape_off, p->tape_off.data(), sizeof(int) * p->tape_off.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipFuncSetAttribute((const void*)thin_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(*p)));
  { int so, ns; HIP_TRY(hipFuncSetAttribute((const void*)thin_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_bwd(*p, &so, &ns))); }
  return 0;
}
void thin_free(ThinProgram* p) {
  if (p->d_fwd) (void)hipFree(p->d_fwd);
  if (p->d_bwd) (void)hipFree(p->d_bwd);
  if (p->d_tape_off) (void)hipFree(p->d_tape_off);
  p->d_fwd = p->d_bwd = nullptr; p->d_tape_off = nullptr;
}

int thin_forward(eegldm_ctx* ctx, const ThinProgram& p, const float* params, const float* x, const float* eps, float* recon, float* z_mu,
                 float* z_sigma, float* kl, int B) {
  EEG_CHECK(p.d_fwd && p.tape && p.stats, "thin program not prepared");
  unsigned long long* prof = prof_buf();
  float* kl_dst = kl; int kl_stride = 0;
  if (kl && eeg_deterministic()) {
    EEG_TRY(eeg_det_buffer(ctx, (size_t)B * sizeof(float), &kl_dst)); kl_stride = 1;
    HIP_TRY(hipMemsetAsync(kl_dst, 0, (size_t)B * sizeof(float), ctx->stream));
  }
  hipLaunchKernelGGL(thin_fwd_kernel, dim3(B), dim3(NT), lds_bytes(p), ctx->stream, p.d_fwd, (int)p.fwd.size(), p.d_tape_off, p.tape_stride, p.nstat, p.maxt,
                     params, x, eps, recon, z_mu, z_sigma, kl_dst, p.tape, p.stats, p.lat, p.Ll, 1.0f / (float)B, p.nparams, prof, kl_stride);
  LAUNCH_CHECK();
  if (kl_stride) EEG_TRY(ew_fold_partials_det(ctx, kl_dst, B, 1, 0, 1, kl));
  if (prof) prof_dump(ctx, p.fwd, prof, "fwd");
  return 0;
}
int thin_backward(eegldm_ctx* ctx, const ThinProgram& p, const float* params, float* grads, const float* d_recon, const float* eps, float klw_over_B,
                  float* dx, int B, const float* dmu_ext, const float* dsg_ext) {
  EEG_CHECK(p.d_bwd && p.tape && p.stats, "thin program not prepared");
  unsigned long long* prof = prof_buf();
  int sp_off = 0, nspare = 0;
  const size_t lds = lds_bytes_bwd(p, &sp_off, &nspare);
  float* g_dst = grads; int g_stride = 0;
  if (eeg_deterministic()) {
    EEG_TRY(eeg_det_buffer(ctx, (size_t)B * p.nparams * sizeof(float), &g_dst)); g_stride = p.nparams;
    HIP_TRY(hipMemsetAsync(g_dst, 0, (size_t)B * p.nparams * sizeof(float), ctx->stream));
  }
  hipLaunchKernelGGL(thin_bwd_kernel, dim3(B), dim3(NT), lds, ctx->stream, p.d_bwd, (int)p.bwd.size(), p.d_tape_off, p.tape_stride, p.nstat, p.maxt,
                     params, g_dst, d_recon, eps, dx, p.tape, p.stats, p.lat, p.Ll, klw_over_B, p.nparams, prof, sp_off, nspare, dmu_ext, dsg_ext, g_stride);
  LAUNCH_CHECK();
  if (g_stride) EEG_TRY(ew_fold_partials_det(ctx, g_dst, B, p.nparams, 0, p.nparams, grads));
  if (prof) prof_dump(ctx, p.bwd, prof, "bwd");
  return 0;
}
