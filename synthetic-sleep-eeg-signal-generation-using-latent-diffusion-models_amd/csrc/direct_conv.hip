// Direct (non-MFMA) 1-D convolution kernels for "thin" layers where one side has <= 8
// channels: UNet conv_in / conv_out (unet.py:385, :504), every layer of the [2,2,4]
// AutoencoderKL (config_aekl_eeg_2_2_4_spec.yaml), the latent heads, the first and last
// PatchDiscriminator convs.  These are HBM/launch-bound (a few MACs per byte): the kernels
// keep global accesses coalesced along the wide channel side and broadcast the thin side.
// Layout NLC; weights packed [K][Cout][Cin] in the activation dtype; bias/grad fp32.
#include "common.h"
#include "internal.h"

namespace {
constexpr int NT = 256;

struct DArgs {
  const void* in; long ldin;      // input rows (x for fwd, dy for dgrad)
  const void* w;                  // [K][Cout][Cin]
  const float* bias;
  const void* resid; long ldr;
  void* out; long ldout;
  int B, Lo, Li;                  // rows per sample of OUTPUT of this launch / of its input
  int Co, Ci;                     // channels of this launch's output / input
  int Cout, Cin;                  // the conv's true Cout/Cin (weight strides)
  int K, stride, pad_l;
  int dgrad;                      // 0: fwd (in row = lo*stride + t - pad_l), 1: dgrad (lo = (li + pad_l - t)/stride)
  float act_slope;                // > 0: LeakyReLU(act_slope) applied to the output (fused activation; thin-input register kernel only)
};

// weight element for (tap t, launch-output channel o, launch-input channel i)
template <typename T>
__device__ __forceinline__ float wsel(const DArgs& a, int t, int o, int i) {
  const T* w = (const T*)a.w;
  return a.dgrad ? ld_f32(w + ((long)t * a.Cout + i) * a.Cin + o) : ld_f32(w + ((long)t * a.Cout + o) * a.Cin + i);
}
// input position for output position l and tap t; -1 if it contributes nothing
__device__ __forceinline__ int in_pos(const DArgs& a, int l, int t) {
  if (!a.dgrad) { const int v = l * a.stride + t - a.pad_l; return (v >= 0 && v < a.Li) ? v : -1; }
  const int u = l + a.pad_l - t;
  if (u < 0 || (u % a.stride) != 0) return -1;
  const int lo = u / a.stride;
  return lo < a.Li ? lo : -1;
}

// thread per (row, out channel); loops taps x (thin) input channels
template <typename T>
__global__ __launch_bounds__(NT) void dconv_point_kernel(const DArgs a) {
  const long total = (long)a.B * a.Lo * a.Co;
  for (long idx = (long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long)gridDim.x * NT) {
    const long r = idx / a.Co; const int o = (int)(idx - r * a.Co);
    const int b = (int)(r / a.Lo), l = (int)(r - (long)b * a.Lo);
    float acc = a.bias ? a.bias[o] : 0.f;
    for (int t = 0; t < a.K; t++) {
      const int v = in_pos(a, l, t);
      if (v < 0) continue;
      const T* xin = (const T*)a.in + ((long)b * a.Li + v) * a.ldin;
      for (int i = 0; i < a.Ci; i++) acc += ld_f32(xin + i) * wsel<T>(a, t, o, i);
    }
    if (a.resid) acc += ld_f32((const T*)a.resid + r * a.ldr + o);
    st_f32((T*)a.out + r * a.ldout + o, acc);
  }
}

// thread per output ROW when both sides are thin (<= 8 channels, padded up to CI / CO in {1,2,4,8}): weights and bias
// sit in LDS (broadcast reads), the input taps are read once per row as one vector when the layout allows, all outputs of
// the row leave as one vector.  The per-(row, channel) kernel above paid two integer divisions and up to 24 scalar
// 2-byte loads per OUTPUT ELEMENT (25 us for a 3 MB tensor).
template <typename T, int N> struct RowVec { T v[N]; };
template <typename T, int CI, int CO>
__global__ __launch_bounds__(NT) void dconv_row_kernel(const DArgs a) {
  __shared__ float wl[3 * 8 * 8];
  __shared__ float bl[8];
  const int nw = a.K * a.Co * a.Ci;
  for (int i = threadIdx.x; i < a.K * CO * CI; i += NT) {
    const int t = i / (CO * CI), o = (i / CI) % CO, ci = i % CI;
    wl[i] = (o < a.Co && ci < a.Ci) ? wsel<T>(a, t, o, ci) : 0.f;
  }
  (void)nw;
  if (threadIdx.x < CO) bl[threadIdx.x] = (a.bias && threadIdx.x < a.Co) ? a.bias[threadIdx.x] : 0.f;
  __syncthreads();
  const long rows = (long)a.B * a.Lo;
  const bool vin = (a.Ci == CI) && (a.ldin % CI == 0), vout = (a.Co == CO) && (a.ldout % CO == 0) && (!a.resid || a.ldr % CO == 0);
  for (long r = (long)blockIdx.x * NT + threadIdx.x; r < rows; r += (long)gridDim.x * NT) {
    const int b = (int)(r / a.Lo), l = (int)(r - (long)b * a.Lo);
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; o++) acc[o] = bl[o];
    for (int t = 0; t < a.K; t++) {
      const int v = in_pos(a, l, t);
      if (v < 0) continue;
      const T* xin = (const T*)a.in + ((long)b * a.Li + v) * a.ldin;
      float xv[CI];
      if (vin) {
        const RowVec<T, CI> rv = *(const RowVec<T, CI>*)xin;
#pragma unroll
        for (int i = 0; i < CI; i++) xv[i] = ld_f32(&rv.v[i]);
      } else {
#pragma unroll
        for (int i = 0; i < CI; i++) xv[i] = i < a.Ci ? ld_f32(xin + i) : 0.f;
      }
      const float* wt = wl + t * CO * CI;
#pragma unroll
      for (int o = 0; o < CO; o++)
#pragma unroll
        for (int i = 0; i < CI; i++) acc[o] += xv[i] * wt[o * CI + i];
    }
    T* op = (T*)a.out + r * a.ldout;
    if (vout) {
      if (a.resid) {
        const RowVec<T, CO> rr = *(const RowVec<T, CO>*)((const T*)a.resid + r * a.ldr);
#pragma unroll
        for (int o = 0; o < CO; o++) acc[o] += ld_f32(&rr.v[o]);
      }
      RowVec<T, CO> ov;
#pragma unroll
      for (int o = 0; o < CO; o++) st_f32(&ov.v[o], acc[o]);
      *(RowVec<T, CO>*)op = ov;
    } else {
#pragma unroll
      for (int o = 0; o < CO; o++) {
        if (o < a.Co) {
          float s = acc[o];
          if (a.resid) s += ld_f32((const T*)a.resid + r * a.ldr + o);
          st_f32(op + o, s);
        }
      }
    }
  }
}

// ---- wide side handled as 16-byte channel groups (G = 8 bf16 / 4 fp32 channels per thread) ----------------------------
// thin input (Ci <= 8), wide output (Co % G == 0): thread per (row, channel group); weights for the group in registers
template <typename T>
__global__ __launch_bounds__(NT) void dconv_thin_in_kernel(const DArgs a) {
  constexpr int G = 16 / sizeof(T);
  extern __shared__ float wl[];                                  // [K][Ci][Co] (output channel fastest) followed by bias [Co]
  for (int i = threadIdx.x; i < a.K * a.Ci * a.Co; i += NT) {
    const int o = i % a.Co, ci = (i / a.Co) % a.Ci, t = i / (a.Co * a.Ci);
    wl[i] = wsel<T>(a, t, o, ci);
  }
  float* bl = wl + a.K * a.Ci * a.Co;
  for (int i = threadIdx.x; i < a.Co; i += NT) bl[i] = a.bias ? a.bias[i] : 0.f;
  __syncthreads();
  const int gpr = a.Co / G;                                    // channel groups per row
  const long total = (long)a.B * a.Lo * gpr;
  for (long idx = (long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long)gridDim.x * NT) {
    const long r = idx / gpr;
    const int o0 = (int)(idx - r * gpr) * G;
    const int b = (int)(r / a.Lo), l = (int)(r - (long)b * a.Lo);
    float acc[G];
#pragma unroll
    for (int k = 0; k < G; k++) acc[k] = bl[o0 + k];
    for (int t = 0; t < a.K; t++) {
      const int v = in_pos(a, l, t);
      if (v < 0) continue;
      const T* xin = (const T*)a.in + ((long)b * a.Li + v) * a.ldin;
      for (int i = 0; i < a.Ci; i++) {
        const float xv = ld_f32(xin + i);
        const float* wp = wl + ((long)t * a.Ci + i) * a.Co + o0;
#pragma unroll
        for (int k = 0; k < G; k++) acc[k] += xv * wp[k];
      }
    }
    if (a.resid) {
      const RowVec<T, G> rr = *(const RowVec<T, G>*)((const T*)a.resid + r * a.ldr + o0);
#pragma unroll
      for (int k = 0; k < G; k++) acc[k] += ld_f32(&rr.v[k]);
    }
    RowVec<T, G> ov;
#pragma unroll
    for (int k = 0; k < G; k++) st_f32(&ov.v[k], acc[k]);
    *(RowVec<T, G>*)((T*)a.out + r * a.ldout + o0) = ov;
  }
}
// Same shape class with the weights of the thread's channel group in REGISTERS and (channel group, row lane) threads: the kernel
// above re-reads 8 weights from LDS per tap and input channel for every output vector and pays two 64-bit divisions per vector
// (57 us for the discriminator's 1 -> 64 stride-2 conv = 0.9 TB/s on a 50 MB output).  CI <= 4, K <= 3, Co/G divides the block.
template <typename T, int CI>
__global__ __launch_bounds__(NT) void dconv_thin_in_reg_kernel(const DArgs a) {
  constexpr int G = 16 / sizeof(T);
  const int gpr = a.Co / G, tx = threadIdx.x % gpr, ty = threadIdx.x / gpr, rpb = NT / gpr, o0 = tx * G;
  float w[3][CI][G], bias[G];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int i = 0; i < CI; i++)
#pragma unroll
      for (int k = 0; k < G; k++) w[t][i][k] = (t < a.K && i < a.Ci) ? wsel<T>(a, t, o0 + k, i) : 0.f;
#pragma unroll
  for (int k = 0; k < G; k++) bias[k] = a.bias ? a.bias[o0 + k] : 0.f;
  const int rows = a.B * a.Lo;
  for (int r = blockIdx.x * rpb + ty; r < rows; r += gridDim.x * rpb) {
    const int b = r / a.Lo, l = r - b * a.Lo;
    float acc[G];
#pragma unroll
    for (int k = 0; k < G; k++) acc[k] = bias[k];
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int v = t < a.K ? in_pos(a, l, t) : -1;
      if (v >= 0) {
        const T* xin = (const T*)a.in + ((long)b * a.Li + v) * a.ldin;
#pragma unroll
        for (int i = 0; i < CI; i++) {
          if (i < a.Ci) {
            const float xv = ld_f32(xin + i);
#pragma unroll
            for (int k = 0; k < G; k++) acc[k] = fmaf(xv, w[t][i][k], acc[k]);
          }
        }
      }
    }
    if (a.resid) {
      const RowVec<T, G> rr = *(const RowVec<T, G>*)((const T*)a.resid + (long)r * a.ldr + o0);
#pragma unroll
      for (int k = 0; k < G; k++) acc[k] += ld_f32(&rr.v[k]);
    }
    if (a.act_slope > 0.f) {      // fused LeakyReLU (the discriminator's first layer): saves a 100 MB elementwise pass per forward
#pragma unroll
      for (int k = 0; k < G; k++) acc[k] = acc[k] > 0.f ? acc[k] : acc[k] * a.act_slope;
    }
    RowVec<T, G> ov;
#pragma unroll
    for (int k = 0; k < G; k++) st_f32(&ov.v[k], acc[k]);
    *(RowVec<T, G>*)((T*)a.out + (long)r * a.ldout + o0) = ov;
  }
}
// Single-channel input (the discriminator's first layer 1 -> 64, stride 2, on 3072-sample windows; stride-1 forward; and the
// stride-1 dgrad of a conv with ONE output channel, the discriminator's last layer, whose gradient input has one channel):
// one block = SEG consecutive output rows of ONE sample.  The input samples the segment needs (<= SEG * stride + 2) are staged once in
// LDS as fp32; the row loop then has no division, no global load and no dependent address arithmetic -- the generic register kernel
// above spent its time in a per-row chain of (r / Lo, three 2-byte global loads, wait) and reached 1.1 TB/s on the 50 MB output.
template <typename T, int SEG>
__global__ __launch_bounds__(NT) void dconv_in1_seg_kernel(const DArgs a) {
  constexpr int G = 16 / sizeof(T);
  __shared__ float xs[SEG * 2 + 8];
  const int gpr = a.Co / G, tx = threadIdx.x % gpr, ty = threadIdx.x / gpr, rpb = NT / gpr, o0 = tx * G;
  const int b = blockIdx.y, l0 = blockIdx.x * SEG;
  const int nseg = min(SEG, a.Lo - l0);
  // input positions v0 .. v0 + nin - 1; tap t of output row l reads xs[l * sm + toff[t]]  (dgrad, stride 1: position l + pad_l - t)
  const bool dg = a.dgrad != 0;
  const int v0 = dg ? l0 + a.pad_l - (a.K - 1) : l0 * a.stride - a.pad_l, nin = (nseg - 1) * a.stride + a.K;
  const int sm = a.stride;
  int toff[3];
#pragma unroll
  for (int t = 0; t < 3; t++) toff[t] = t < a.K ? (dg ? a.K - 1 - t : t) : 0;
  const T* xin = (const T*)a.in + (long)b * a.Li * a.ldin;
  for (int i = threadIdx.x; i < SEG * 2 + 8; i += NT) { const int v = v0 + i; xs[i] = (i < nin && v >= 0 && v < a.Li) ? ld_f32(xin + (long)v * a.ldin) : 0.f; }   // (zeros behind the segment: a K < 3 conv multiplies them by zero weights)
  float w[3][G], bias[G];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int k = 0; k < G; k++) w[t][k] = t < a.K ? wsel<T>(a, t, o0 + k, 0) : 0.f;
#pragma unroll
  for (int k = 0; k < G; k++) bias[k] = a.bias ? a.bias[o0 + k] : 0.f;
  __syncthreads();
  T* out = (T*)a.out + ((long)b * a.Lo + l0) * a.ldout + o0;
  const T* res = a.resid ? (const T*)a.resid + ((long)b * a.Lo + l0) * a.ldr + o0 : nullptr;
#pragma unroll 4
  for (int l = ty; l < nseg; l += rpb) {
    const float x0 = xs[l * sm + toff[0]], x1 = xs[l * sm + toff[1]], x2 = xs[l * sm + toff[2]];
    float acc[G];
#pragma unroll
    for (int k = 0; k < G; k++) acc[k] = fmaf(x2, w[2][k], fmaf(x1, w[1][k], fmaf(x0, w[0][k], bias[k])));
    if (res) {
      const RowVec<T, G> rr = *(const RowVec<T, G>*)(res + (long)l * a.ldr);
#pragma unroll
      for (int k = 0; k < G; k++) acc[k] += ld_f32(&rr.v[k]);
    }
    if (a.act_slope > 0.f) {
#pragma unroll
      for (int k = 0; k < G; k++) acc[k] = acc[k] > 0.f ? acc[k] : acc[k] * a.act_slope;
    }
    RowVec<T, G> ov;
#pragma unroll
    for (int k = 0; k < G; k++) st_f32(&ov.v[k], acc[k]);
    *(RowVec<T, G>*)(out + (long)l * a.ldout) = ov;
  }
}
// wide input (Ci % G == 0, Ci/G a power of two <= 64), thin output (Co <= 8): Ci/G lanes per row, shuffle reduction
template <typename T>
__global__ __launch_bounds__(NT) void dconv_thin_out_kernel(const DArgs a) {
  constexpr int G = 16 / sizeof(T);
  extern __shared__ float wl[];                                  // [K][Co][Ci]
  for (int i = threadIdx.x; i < a.K * a.Co * a.Ci; i += NT) {
    const int t = i / (a.Co * a.Ci), o = (i / a.Ci) % a.Co, ci = i % a.Ci;
    wl[i] = wsel<T>(a, t, o, ci);
  }
  __syncthreads();
  const int lpr = a.Ci / G, lane = threadIdx.x % lpr, rpb = NT / lpr;
  const long rows = (long)a.B * a.Lo;
  for (long r = (long)blockIdx.x * rpb + threadIdx.x / lpr; r < rows + rpb; r += (long)gridDim.x * rpb) {   // uniform trip count for the shuffles
    const bool ok = r < rows;
    const long rc = ok ? r : rows - 1;
    const int b = (int)(rc / a.Lo), l = (int)(rc - (long)b * a.Lo);
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; o++) acc[o] = 0.f;
    for (int t = 0; t < a.K; t++) {
      const int v = in_pos(a, l, t);
      if (v < 0) continue;
      const RowVec<T, G> rv = *(const RowVec<T, G>*)((const T*)a.in + ((long)b * a.Li + v) * a.ldin + lane * G);
      float xv[G];
#pragma unroll
      for (int k = 0; k < G; k++) xv[k] = ld_f32(&rv.v[k]);
#pragma unroll
      for (int o = 0; o < 8; o++) {
        if (o < a.Co) {
          const float* wp = wl + ((long)t * a.Co + o) * a.Ci + lane * G;
#pragma unroll
          for (int k = 0; k < G; k++) acc[o] += xv[k] * wp[k];
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 8; o++) {
      if (o >= a.Co) break;
      float s = acc[o];
      for (int d = lpr >> 1; d > 0; d >>= 1) s += __shfl_xor(s, d, 64);
      if (lane == 0 && ok) {
        if (a.bias) s += a.bias[o];
        if (a.resid) s += ld_f32((const T*)a.resid + r * a.ldr + o);
        st_f32((T*)a.out + r * a.ldout + o, s);
      }
    }
  }
}
// Co == 1, stride 1, forward: a lane group walks a RUN of R consecutive output rows of one sample with a sliding window -- every
// input row is loaded once (R + K - 1 independent 16-byte loads per lane instead of K * R), the tap weights of the lane's channels
// sit in registers, and the R row sums are shuffle-reduced together.  (One row per group-iteration with three dependent loads and
// LDS weight reads ran the discriminator's 512 -> 1 conv at 1.3 TB/s.)
template <typename T, int R>
__global__ __launch_bounds__(NT) void dconv_thin_out_run_kernel(const DArgs a) {
  constexpr int G = 16 / sizeof(T);
  const int lpr = a.Ci / G, lane = threadIdx.x % lpr, gpb = NT / lpr;
  float w[3][G];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int k = 0; k < G; k++) w[t][k] = t < a.K ? wsel<T>(a, t, 0, lane * G + k) : 0.f;
  const int rps = a.Lo / R;                                      // runs per sample
  const long nruns = (long)a.B * rps;
  const float bias = a.bias ? a.bias[0] : 0.f;
  for (long g = (long)blockIdx.x * gpb + threadIdx.x / lpr; g < nruns + gpb; g += (long)gridDim.x * gpb) {   // uniform trip count for the shuffles
    const bool ok = g < nruns;
    const long gc = ok ? g : nruns - 1;
    const int b = (int)((unsigned)gc / (unsigned)rps), l0 = ((int)gc - b * rps) * R;     // (the launcher keeps nruns < 2^31)
    float acc[R];
#pragma unroll
    for (int i = 0; i < R; i++) acc[i] = 0.f;
    // all R + 2 row loads are issued unconditionally (clamped row index): predicated loads made hipcc wait for each one separately,
    // a chain of ten L2 / HBM round trips per run (63 us for the discriminator's 512 -> 1 layer = 0.8 TB/s on a 50 MB input)
    RowVec<T, G> rv[R + 2];
    const T* xb = (const T*)a.in + (long)b * a.Li * a.ldin + lane * G;
#pragma unroll
    for (int j = 0; j < R + 2; j++) {
      const int v = l0 - a.pad_l + j, vc = v < 0 ? 0 : (v >= a.Li ? a.Li - 1 : v);
      rv[j] = *(const RowVec<T, G>*)(xb + (long)vc * a.ldin);
    }
#pragma unroll
    for (int j = 0; j < R + 2; j++) {
      const int v = l0 - a.pad_l + j;
      if (j < R + a.K - 1 && v >= 0 && v < a.Li) {
        float xv[G];
#pragma unroll
        for (int k = 0; k < G; k++) xv[k] = ld_f32(&rv[j].v[k]);
#pragma unroll
        for (int t = 0; t < 3; t++) {
          if (t < a.K && j - t >= 0 && j - t < R) {              // input row l0 - pad + j is tap t of output row l0 + j - t
            float d = 0.f;
#pragma unroll
            for (int k = 0; k < G; k++) d = fmaf(xv[k], w[t][k], d);
            acc[j - t] += d;
          }
        }
      }
    }
    if (R == 8 && lpr == 64) {
      // eight sums over the 64 lanes of the wave with 10 shuffles instead of 48: each halving step trades half of the values with the
      // partner lane (bit 5, 4, 3 of the lane select rows 4.., 2.., 1..), then three single-value steps over the remaining 8 lanes
      float h4[4], h2[2], h1;
      const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
#pragma unroll
      for (int i = 0; i < 4; i++) { const float keep = b5 ? acc[4 + i] : acc[i], send = b5 ? acc[i] : acc[4 + i]; h4[i] = keep + __shfl_xor(send, 32, 64); }
#pragma unroll
      for (int i = 0; i < 2; i++) { const float keep = b4 ? h4[2 + i] : h4[i], send = b4 ? h4[i] : h4[2 + i]; h2[i] = keep + __shfl_xor(send, 16, 64); }
      { const float keep = b3 ? h2[1] : h2[0], send = b3 ? h2[0] : h2[1]; h1 = keep + __shfl_xor(send, 8, 64); }
      h1 += __shfl_xor(h1, 4, 64); h1 += __shfl_xor(h1, 2, 64); h1 += __shfl_xor(h1, 1, 64);
      if ((lane & 7) == 0 && ok) {
        const int i = (b5 ? 4 : 0) + (b4 ? 2 : 0) + (b3 ? 1 : 0);
        const long r = (long)b * a.Lo + l0 + i;
        float sres = h1 + bias;
        if (a.resid) sres += ld_f32((const T*)a.resid + r * a.ldr);
        st_f32((T*)a.out + r * a.ldout, sres);
      }
    } else {
#pragma unroll
    for (int i = 0; i < R; i++)
      for (int d = lpr >> 1; d > 0; d >>= 1) acc[i] += __shfl_xor(acc[i], d, 64);
    if (lane == 0 && ok) {
#pragma unroll
      for (int i = 0; i < R; i++) {
        const long r = (long)b * a.Lo + l0 + i;
        float sres = acc[i] + bias;
        if (a.resid) sres += ld_f32((const T*)a.resid + r * a.ldr);
        st_f32((T*)a.out + r * a.ldout, sres);
      }
    }
    }
  }
}
// Weight gradient of a (wide Cin) -> ONE output channel conv, stride 1 (the discriminator's last layer, 512 -> 1 at 192 rows per sample):
// a few blocks per sample (segr rows each); the sample's dy (one value per row) sits zero-padded in LDS, every x row is read ONCE and feeds all taps
// (dW[t][ci] += x[v][ci] * dy[v - t + pad_l]) -- the generic kernel above reads each x row once per tap (3 x 50 MB through the L2).
template <typename T>
__global__ __launch_bounds__(NT) void dconv_wgrad_in1out_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                                                float* __restrict__ parts, int Lo, int Cin, int K, int pad_l, int segr) {
  constexpr int G = 16 / sizeof(T);
  constexpr int MAXL = 1024;
  __shared__ float dys[MAXL + 4];
  __shared__ float red[3 * G][NT];
  const int lpr = Cin / G, lane = threadIdx.x % lpr, rpb = NT / lpr, rl = threadIdx.x / lpr;
  const int b = blockIdx.x, v_lo = blockIdx.y * segr, v_hi = min(Lo, v_lo + segr);      // this block's x rows of sample b
  for (int i = threadIdx.x; i < Lo + 4; i += NT) { const int l = i - 2; dys[i] = (l >= 0 && l < Lo) ? ld_f32(dy + ((long)b * Lo + l) * lddy) : 0.f; }
  __syncthreads();
  float acc[3][G];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int k = 0; k < G; k++) acc[t][k] = 0.f;
  const T* xb = x + (long)b * Lo * ldx + lane * G;
#pragma unroll 8
  for (int v = v_lo + rl; v < v_hi; v += rpb) {
    const RowVec<T, G> xv = *(const RowVec<T, G>*)(xb + (long)v * ldx);
    float d[3];
#pragma unroll
    for (int t = 0; t < 3; t++) d[t] = t < K ? dys[v - t + pad_l + 2] : 0.f;
#pragma unroll
    for (int k = 0; k < G; k++) { const float xf = ld_f32(&xv.v[k]);
#pragma unroll
      for (int t = 0; t < 3; t++) acc[t][k] = fmaf(xf, d[t], acc[t][k]); }
  }
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int k = 0; k < G; k++) red[t * G + k][threadIdx.x] = acc[t][k];
  __syncthreads();
  for (int o = threadIdx.x; o < 3 * G * lpr; o += NT) {
    const int pl = o / lpr, ln = o - pl * lpr, t = pl / G, k = pl % G;
    if (t >= K) continue;
    float sum = 0.f;
    for (int q = 0; q < rpb; q++) sum += red[pl][q * lpr + ln];
    parts[((long)b * gridDim.y + blockIdx.y) * ((long)K * Cin) + (long)t * Cin + ln * G + k] = sum;      // [block][t][co = 0][ci]
  }
}
// weight gradient of a (thin <= 2) x (wide % G == 0) conv: dW[t][co][ci] += sum_r dy[r][co] * x[in_row(r,t)][ci].
// WIDE_OUT: the wide side is Cout (dy rows are wide, x is thin) else Cin (x rows wide, dy thin).
template <typename T, bool WIDE_OUT>
__global__ __launch_bounds__(NT) void dconv_wgrad_wt_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                                            float* __restrict__ dw, float* __restrict__ parts, int B, int Lo, int Li, int Cout, int Cin, int K,
                                                            int stride, int pad_l) {
  constexpr int G = 16 / sizeof(T);
  __shared__ float red[3 * 2 * G][NT];       // every accumulator plane of the block at once: ONE barrier in the epilogue (it was 48)
  const int wide = WIDE_OUT ? Cout : Cin, thin = WIDE_OUT ? Cin : Cout;
  const int lpr = wide / G, lane = threadIdx.x % lpr, rpb = NT / lpr, rl = threadIdx.x / lpr;
  float acc[3][2][G];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < G; k++) acc[t][j][k] = 0.f;
  const int rows = B * Lo;                    // (the launcher keeps B * Lo < 2^31: 32-bit index arithmetic in the row loop)
  // loads are unconditional (clamped row, zero factor for taps outside the sample): predicated loads are waited for one by one
#pragma unroll 2
  for (int r = blockIdx.x * rpb + rl; r < rows; r += gridDim.x * rpb) {
    const int b = (int)((unsigned)r / (unsigned)Lo), lo = r - b * Lo;
    if (WIDE_OUT) {
      const RowVec<T, G> dv = *(const RowVec<T, G>*)(dy + (long)r * lddy + lane * G);
      float xv[3][2];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int v = lo * stride + t - pad_l;
        const bool ok = t < K && v >= 0 && v < Li;
        const T* xr = x + ((long)b * Li + (ok ? v : 0)) * ldx;
#pragma unroll
        for (int j = 0; j < 2; j++) xv[t][j] = (j < thin) ? (ok ? ld_f32(xr + j) : 0.f) : 0.f;
      }
      float d[G];
#pragma unroll
      for (int k = 0; k < G; k++) d[k] = ld_f32(&dv.v[k]);
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
          if (j < thin) {
#pragma unroll
            for (int k = 0; k < G; k++) acc[t][j][k] = fmaf(d[k], xv[t][j], acc[t][j][k]);
          }
        }
    } else {
      RowVec<T, G> xvv[3]; float f[3];
#pragma unroll
      for (int t = 0; t < 3; t++) {
        const int v = lo * stride + t - pad_l;
        const bool ok = t < K && v >= 0 && v < Li;
        f[t] = ok ? 1.f : 0.f;
        xvv[t] = *(const RowVec<T, G>*)(x + ((long)b * Li + (ok ? v : 0)) * ldx + lane * G);
      }
      float d[2];
#pragma unroll
      for (int j = 0; j < 2; j++) d[j] = j < thin ? ld_f32(dy + (long)r * lddy + j) : 0.f;
#pragma unroll
      for (int t = 0; t < 3; t++)
#pragma unroll
        for (int k = 0; k < G; k++) { const float xv = ld_f32(&xvv[t].v[k]) * f[t];
#pragma unroll
          for (int j = 0; j < 2; j++) acc[t][j][k] = fmaf(d[j], xv, acc[t][j][k]); }
    }
  }
  // reduce over the rpb row lanes of the block: all planes go to LDS, one barrier, then (plane, channel group) threads sum their column
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int k = 0; k < G; k++) red[(t * 2 + j) * G + k][threadIdx.x] = acc[t][j][k];
  __syncthreads();
  for (int o = threadIdx.x; o < 3 * 2 * G * lpr; o += NT) {
    const int pl = o / lpr, ln = o - pl * lpr;
    const int t = pl / (2 * G), j = (pl / G) % 2, k = pl % G;
    if (t >= K || j >= thin) continue;
    float sum = 0.f;
    for (int q = 0; q < rpb; q++) sum += red[pl][q * lpr + ln];
    const int wch = ln * G + k;
    const int co = WIDE_OUT ? wch : j, ci = WIDE_OUT ? j : wch;
    // written partial per block + a folding pass (hundreds of blocks adding atomically into the same few hundred
    // addresses serialise in L2); plain atomics only when no workspace was given
    const long e = ((long)t * Cout + co) * Cin + ci;
    if (parts) parts[(long)blockIdx.x * ((long)K * Cout * Cin) + e] = sum; else atomicAdd(dw + e, sum);
  }
}

// wave per row, lanes over the (wide) input channels, <= 8 output channels
template <typename T>
__global__ __launch_bounds__(NT) void dconv_rowdot_kernel(const DArgs a) {
  const int lane = threadIdx.x & 63;
  const long rows = (long)a.B * a.Lo;
  for (long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long)gridDim.x * 4) {
    const int b = (int)(r / a.Lo), l = (int)(r - (long)b * a.Lo);
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; o++) acc[o] = 0.f;
    for (int t = 0; t < a.K; t++) {
      const int v = in_pos(a, l, t);
      if (v < 0) continue;
      const T* xin = (const T*)a.in + ((long)b * a.Li + v) * a.ldin;
      for (int i = lane; i < a.Ci; i += 64) {
        const float xv = ld_f32(xin + i);
#pragma unroll
        for (int o = 0; o < 8; o++) if (o < a.Co) acc[o] += xv * wsel<T>(a, t, o, i);
      }
    }
#pragma unroll
    for (int o = 0; o < 8; o++) {
      if (o >= a.Co) break;
      float s = wave_sum(acc[o]);
      if (lane == 0) {
        if (a.bias) s += a.bias[o];
        if (a.resid) s += ld_f32((const T*)a.resid + r * a.ldr + o);
        st_f32((T*)a.out + r * a.ldout + o, s);
      }
    }
  }
}

// dW[t][co][ci] += sum_r dy[r][co] * x[in_row(r,t)][ci].
// "wide" variant (one side up to 512 channels): thread per weight element, each block grid-strides over row chunks
// and issues ONE atomic per element at the end (atomics = blocks x E, not threads x chunks).
template <typename T>
__global__ __launch_bounds__(NT) void dconv_wgrad_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                                         float* __restrict__ dw, int B, int Lo, int Li, int Cout, int Cin, int K,
                                                         int stride, int pad_l, int rows_per_chunk) {
  const int E = K * Cout * Cin;
  const long rows = (long)B * Lo;
  for (int e = threadIdx.x; e < E; e += NT) {
    const int ci = e % Cin, co = (e / Cin) % Cout, t = e / (Cin * Cout);
    float acc = 0.f;
    for (long r0 = (long)blockIdx.x * rows_per_chunk; r0 < rows; r0 += (long)gridDim.x * rows_per_chunk) {
      const long r1 = min(rows, r0 + rows_per_chunk);
      for (long r = r0; r < r1; r++) {
        const int b = (int)(r / Lo), lo = (int)(r - (long)b * Lo);
        const int v = lo * stride + t - pad_l;
        if (v >= 0 && v < Li) acc += ld_f32(dy + r * lddy + co) * ld_f32(x + ((long)b * Li + v) * ldx + ci);
      }
    }
    atomicAdd(dw + e, acc);
  }
}

// vector-load version of the tiny weight gradient: exact channel counts as template constants, one load per row and tap
template <typename T, int CI, int CO>
__global__ __launch_bounds__(NT) void dconv_wgrad_tinyv_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                                               float* __restrict__ parts, int B, int Lo, int Li, int K, int stride, int pad_l) {
  float acc[3][CO][CI];
  float dbs[CO];
#pragma unroll
  for (int o = 0; o < CO; o++) dbs[o] = 0.f;
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int o = 0; o < CO; o++)
#pragma unroll
      for (int i = 0; i < CI; i++) acc[t][o][i] = 0.f;
  const long rows = (long)B * Lo;
#pragma unroll 2
  for (long r = (long)blockIdx.x * NT + threadIdx.x; r < rows; r += (long)gridDim.x * NT) {
    const int b = (int)(r / Lo), lo = (int)(r - (long)b * Lo);
    const RowVec<T, CO> dv = *(const RowVec<T, CO>*)(dy + r * lddy);
    float d[CO];
#pragma unroll
    for (int o = 0; o < CO; o++) { d[o] = ld_f32(&dv.v[o]); dbs[o] += d[o]; }
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int v = lo * stride + t - pad_l;
      if (t < K && v >= 0 && v < Li) {
        const RowVec<T, CI> xv = *(const RowVec<T, CI>*)(x + ((long)b * Li + v) * ldx);
#pragma unroll
        for (int i = 0; i < CI; i++) {
          const float xf = ld_f32(&xv.v[i]);
#pragma unroll
          for (int o = 0; o < CO; o++) acc[t][o][i] += d[o] * xf;
        }
      }
    }
  }
  // one row of 3*CO*CI weight-gradient sums (dw is [K][CO][CI]: the flat index) + CO bias-gradient sums per block, WRITTEN to the
  // partial buffer and folded by dconv_tiny_fold_kernel: 512 blocks adding atomically into the same <= 52 addresses serialised
  // in L2 for 15-35 us per launch, and the bias gradient was a separate column-sum pass over dy
  constexpr int EW = 3 * CO * CI, EP = EW + CO;
  __shared__ float red[4][EP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int o = 0; o < CO; o++)
#pragma unroll
      for (int i = 0; i < CI; i++) {
        const float s = wave_sum(acc[t][o][i]);
        if (lane == 0) red[wave][(t * CO + o) * CI + i] = s;
      }
#pragma unroll
  for (int o = 0; o < CO; o++) {
    const float s = wave_sum(dbs[o]);
    if (lane == 0) red[wave][EW + o] = s;
  }
  __syncthreads();
  const int e = threadIdx.x;
  if (e < EP) parts[(long)blockIdx.x * EP + e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// dw[e] += sum_p parts[p][e] (e < ew), dbias[o] += sum_p parts[p][bias_off + o]: lanes = elements, (block, wave) = part subsets
__global__ __launch_bounds__(NT) void dconv_tiny_fold_kernel(const float* __restrict__ parts, int nparts, int stride, int ew, int bias_off, int co,
                                                             float* __restrict__ dw, float* __restrict__ dbias) {
  const int e = threadIdx.x & 63, w = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nw = gridDim.x * (NT / 64);
  const bool is_w = e < ew, is_b = dbias && e >= bias_off && e < bias_off + co;
  if (!is_w && !is_b) return;
  float s = 0.f;
#pragma unroll 4
  for (int p = w; p < nparts; p += nw) s += parts[(long)p * stride + e];
  atomicAdd(is_w ? dw + e : dbias + (e - bias_off), s);
}

// "tiny" variant (Cin, Cout <= 4; every layer of the [2,2,4] autoencoder): thread per output row with all K*Cout*Cin
// partial sums in registers, wave-shuffle + LDS reduction, one atomic per element per block.
template <typename T>
__global__ __launch_bounds__(NT) void dconv_wgrad_tiny_kernel(const T* __restrict__ x, long ldx, const T* __restrict__ dy, long lddy,
                                                              float* __restrict__ dw, int B, int Lo, int Li, int Cout, int Cin, int K,
                                                              int stride, int pad_l) {
  float acc[3][4][4];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
      for (int i = 0; i < 4; i++) acc[t][o][i] = 0.f;
  const long rows = (long)B * Lo;
  for (long r = (long)blockIdx.x * NT + threadIdx.x; r < rows; r += (long)gridDim.x * NT) {
    const int b = (int)(r / Lo), lo = (int)(r - (long)b * Lo);
    float d[4];
#pragma unroll
    for (int o = 0; o < 4; o++) d[o] = o < Cout ? ld_f32(dy + r * lddy + o) : 0.f;
#pragma unroll
    for (int t = 0; t < 3; t++) {
      const int v = lo * stride + t - pad_l;
      if (t < K && v >= 0 && v < Li) {
        const T* xr = x + ((long)b * Li + v) * ldx;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const float xv = i < Cin ? ld_f32(xr + i) : 0.f;
#pragma unroll
          for (int o = 0; o < 4; o++) acc[t][o][i] += d[o] * xv;
        }
      }
    }
  }
  __shared__ float red[4][48];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int o = 0; o < 4; o++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const float s = wave_sum(acc[t][o][i]);
        if (lane == 0) red[wave][(t * 4 + o) * 4 + i] = s;
      }
  __syncthreads();
  const int e = threadIdx.x;
  if (e < 48) {
    const int t = e / 16, o = (e / 4) % 4, i = e % 4;
    if (t < K && o < Cout && i < Cin) atomicAdd(dw + ((long)t * Cout + o) * Cin + i, red[0][e] + red[1][e] + red[2][e] + red[3][e]);
  }
}

inline int grid_cap(long blocks, eegldm_ctx* ctx) {
  long cap = (long)ctx->num_cu * 16;
  if (blocks < 1) blocks = 1;
  return (int)(blocks < cap ? blocks : cap);
}
}  // namespace

// is this conv handled by the direct kernels (vs the MFMA implicit GEMM)?
bool conv_is_thin(int Cin, int Cout, int dtype) {
  const int epc = dtype == EEGLDM_F32 ? 4 : 8;
  return Cin < 16 || Cout < 16 || (Cin % epc) != 0 || (Cout % epc) != 0;
}

// true when dconv_run takes the register-resident thin-input kernel for this forward conv (the one that can fuse a LeakyReLU)
bool dconv_fuses_act(int dtype, int Cin, int Cout, int K, long ldout) {
  const int G = dtype == EEGLDM_F32 ? 4 : 8;
  constexpr bool reg_ok = true, fuse_ok = true;
  const int gpr = Cout / G;
  return reg_ok && fuse_ok && Cin <= 4 && Cout % G == 0 && Cout >= 16 && K <= 3 && ldout % G == 0 && ((size_t)K * Cin * Cout + Cout) * 4 <= 48 * 1024 &&
         gpr <= NT && NT % gpr == 0;
}
int dconv_run(eegldm_ctx* ctx, int dtype, bool dgrad, const void* in, long ldin, const void* w, const float* bias,
              const void* resid, long ldr, void* out, long ldout, int B, int Lin, int Lout, int Cin, int Cout, int K,
              int stride, int pad_l, float act_slope) {
  DArgs a;
  a.act_slope = act_slope;
  EEG_CHECK(act_slope <= 0.f || (!dgrad && dconv_fuses_act(dtype, Cin, Cout, K, ldout) && (long)B * Lout < (1L << 30)), "fused activation is not available for this conv");
  a.in = in; a.ldin = ldin; a.w = w; a.bias = bias; a.resid = resid; a.ldr = ldr; a.out = out; a.ldout = ldout;
  a.B = B; a.Cout = Cout; a.Cin = Cin; a.K = K; a.stride = stride; a.pad_l = pad_l; a.dgrad = dgrad ? 1 : 0;
  if (!dgrad) { a.Lo = Lout; a.Li = Lin; a.Co = Cout; a.Ci = Cin; }
  else        { a.Lo = Lin;  a.Li = Lout; a.Co = Cin;  a.Ci = Cout; }
  const bool rowdot = a.Ci > 8 && a.Co <= 8;
  EEG_CHECK(a.Ci <= 8 || a.Co <= 8, "direct conv expects a thin side (Cin=%d Cout=%d)", Cin, Cout);
  const long rows = (long)B * a.Lo;
  if (a.Ci <= 8 && a.Co <= 8 && K <= 3) {
    const int ci = a.Ci <= 1 ? 1 : (a.Ci <= 2 ? 2 : (a.Ci <= 4 ? 4 : 8)), co = a.Co <= 1 ? 1 : (a.Co <= 2 ? 2 : (a.Co <= 4 ? 4 : 8));
    const dim3 g(grid_cap((rows + NT - 1) / NT, ctx));
#define DROW(T_, CI_, CO_) hipLaunchKernelGGL((dconv_row_kernel<T_, CI_, CO_>), g, dim3(NT), 0, ctx->stream, a)
#define DROW_CO(T_, CI_) do { if (co == 1) DROW(T_, CI_, 1); else if (co == 2) DROW(T_, CI_, 2); else if (co == 4) DROW(T_, CI_, 4); else DROW(T_, CI_, 8); } while (0)
#define DROW_T(T_) do { if (ci == 1) DROW_CO(T_, 1); else if (ci == 2) DROW_CO(T_, 2); else if (ci == 4) DROW_CO(T_, 4); else DROW_CO(T_, 8); } while (0)
    if (dtype == EEGLDM_F32) DROW_T(float); else if (dtype == EEGLDM_F16) DROW_T(f16_t); else DROW_T(bf16_t);
#undef DROW_T
#undef DROW_CO
#undef DROW
    LAUNCH_CHECK();
    return 0;
  }
  {
    const int G = dtype == EEGLDM_F32 ? 4 : 8;
    const bool al = (ldin % G == 0) && (ldout % G == 0) && (!resid || ldr % G == 0);
    if (a.Ci <= 8 && a.Co % G == 0 && a.Co >= 16 && K <= 3 && ldout % G == 0 && (!resid || ldr % G == 0) && ((size_t)K * a.Ci * a.Co + a.Co) * 4 <= 48 * 1024) {
      const long total = rows * (a.Co / G);
      const dim3 g(grid_cap((total + NT - 1) / NT, ctx));
      constexpr bool reg_ok = true;
      const int gpr = a.Co / G;
      constexpr bool seg_ok = true;
      if (seg_ok && a.Ci == 1 && K <= 3 && (dgrad ? stride == 1 : stride <= 2) && gpr <= NT && NT % gpr == 0 && a.Lo >= 64 && a.B <= 65535) {
        constexpr int SEG = 128;
        const dim3 gs((a.Lo + SEG - 1) / SEG, a.B);
        if (dtype == EEGLDM_F32) hipLaunchKernelGGL((dconv_in1_seg_kernel<float, SEG>), gs, dim3(NT), 0, ctx->stream, a);
        else if (dtype == EEGLDM_F16) hipLaunchKernelGGL((dconv_in1_seg_kernel<f16_t, SEG>), gs, dim3(NT), 0, ctx->stream, a);
        else hipLaunchKernelGGL((dconv_in1_seg_kernel<bf16_t, SEG>), gs, dim3(NT), 0, ctx->stream, a);
        LAUNCH_CHECK();
        return 0;
      }
      if (reg_ok && a.Ci <= 4 && gpr <= NT && NT % gpr == 0 && rows < (1L << 30)) {
#define DTI(T_, CI_) hipLaunchKernelGGL((dconv_thin_in_reg_kernel<T_, CI_>), g, dim3(NT), 0, ctx->stream, a)
#define DTI_T(T_) do { if (a.Ci == 1) DTI(T_, 1); else if (a.Ci == 2) DTI(T_, 2); else DTI(T_, 4); } while (0)
        if (dtype == EEGLDM_F32) DTI_T(float); else if (dtype == EEGLDM_F16) DTI_T(f16_t); else DTI_T(bf16_t);
#undef DTI_T
#undef DTI
        LAUNCH_CHECK();
        return 0;
      }
      const size_t sh = ((size_t)K * a.Ci * a.Co + a.Co) * sizeof(float);
      if (dtype == EEGLDM_F32) hipLaunchKernelGGL((dconv_thin_in_kernel<float>), g, dim3(NT), sh, ctx->stream, a);
      else if (dtype == EEGLDM_F16) hipLaunchKernelGGL((dconv_thin_in_kernel<f16_t>), g, dim3(NT), sh, ctx->stream, a);
      else hipLaunchKernelGGL((dconv_thin_in_kernel<bf16_t>), g, dim3(NT), sh, ctx->stream, a);
      LAUNCH_CHECK();
      return 0;
    }
    const int lpr = a.Ci / G;
    if (a.Co <= 8 && a.Ci % G == 0 && lpr >= 1 && lpr <= 64 && (lpr & (lpr - 1)) == 0 && ldin % G == 0 && (size_t)K * a.Co * a.Ci * 4 <= 48 * 1024 && al == al) {
      const int rpb = NT / lpr;
      constexpr bool run_ok = true;
      constexpr int RUN = 8;
      if (run_ok && !dgrad && a.Co == 1 && stride == 1 && K <= 3 && a.Lo % RUN == 0 && a.Li == a.Lo && pad_l <= K - 1 && rows < (1L << 31)) {
        const long nruns = rows / RUN;
        const dim3 gr(grid_cap((nruns + rpb - 1) / rpb, ctx));
        if (dtype == EEGLDM_F32) hipLaunchKernelGGL((dconv_thin_out_run_kernel<float, RUN>), gr, dim3(NT), 0, ctx->stream, a);
        else if (dtype == EEGLDM_F16) hipLaunchKernelGGL((dconv_thin_out_run_kernel<f16_t, RUN>), gr, dim3(NT), 0, ctx->stream, a);
        else hipLaunchKernelGGL((dconv_thin_out_run_kernel<bf16_t, RUN>), gr, dim3(NT), 0, ctx->stream, a);
        LAUNCH_CHECK();
        return 0;
      }
      const dim3 g(grid_cap((rows + rpb - 1) / rpb, ctx));
      const size_t sh = (size_t)K * a.Co * a.Ci * sizeof(float);
      if (dtype == EEGLDM_F32) hipLaunchKernelGGL((dconv_thin_out_kernel<float>), g, dim3(NT), sh, ctx->stream, a);
      else if (dtype == EEGLDM_F16) hipLaunchKernelGGL((dconv_thin_out_kernel<f16_t>), g, dim3(NT), sh, ctx->stream, a);
      else hipLaunchKernelGGL((dconv_thin_out_kernel<bf16_t>), g, dim3(NT), sh, ctx->stream, a);
      LAUNCH_CHECK();
      return 0;
    }
  }
  if (dtype == EEGLDM_F32) {
    if (rowdot) hipLaunchKernelGGL((dconv_rowdot_kernel<float>), dim3(grid_cap((rows + 3) / 4, ctx)), dim3(NT), 0, ctx->stream, a);
    else hipLaunchKernelGGL((dconv_point_kernel<float>), dim3(grid_cap((rows * a.Co + NT - 1) / NT, ctx)), dim3(NT), 0, ctx->stream, a);
  } else if (dtype == EEGLDM_F16) {
    if (rowdot) hipLaunchKernelGGL((dconv_rowdot_kernel<f16_t>), dim3(grid_cap((rows + 3) / 4, ctx)), dim3(NT), 0, ctx->stream, a);
    else hipLaunchKernelGGL((dconv_point_kernel<f16_t>), dim3(grid_cap((rows * a.Co + NT - 1) / NT, ctx)), dim3(NT), 0, ctx->stream, a);
  } else {
    if (rowdot) hipLaunchKernelGGL((dconv_rowdot_kernel<bf16_t>), dim3(grid_cap((rows + 3) / 4, ctx)), dim3(NT), 0, ctx->stream, a);
    else hipLaunchKernelGGL((dconv_point_kernel<bf16_t>), dim3(grid_cap((rows * a.Co + NT - 1) / NT, ctx)), dim3(NT), 0, ctx->stream, a);
  }
  LAUNCH_CHECK();
  return 0;
}

bool dconv_wgrad_tinyv_ok(int Cin, int Cout, int K) {
  return K <= 3 && (Cin == 1 || Cin == 2 || Cin == 4) && (Cout == 1 || Cout == 2 || Cout == 4);
}

// *bias_done (optional) is set when the kernel also produced dbias (the tiny vector kernel); otherwise the caller sums dy itself
int dconv_wgrad(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* dy, long lddy, float* dw, int B, int Lin,
                int Lout, int Cin, int Cout, int K, int stride, int pad_l, float* dbias, int* bias_done) {
  const long rows = (long)B * Lout;
  if (bias_done) *bias_done = 0;
  if (Cin <= 4 && Cout <= 4 && K <= 3) {
    long nb = (rows + NT - 1) / NT; if (nb > 2L * ctx->num_cu) nb = 2L * ctx->num_cu;
    const int blocks = (int)(nb < 1 ? 1 : nb);
    if (dconv_wgrad_tinyv_ok(Cin, Cout, K) && ldx % Cin == 0 && lddy % Cout == 0) {
      // partial sums live in their own slice of the context scratch ([4, 5) MiB): this path also runs on the side stream
      float* parts = (float*)((char*)ctx->scratch + (4u << 20));
      const int EW = 3 * Cout * Cin, EP = EW + Cout;
#define DWV(T_, CI_, CO_) hipLaunchKernelGGL((dconv_wgrad_tinyv_kernel<T_, CI_, CO_>), dim3(blocks), dim3(NT), 0, ctx->stream, (const T_*)x, ldx, (const T_*)dy, lddy, \
                                             parts, B, Lout, Lin, K, stride, pad_l)
#define DWV_CO(T_, CI_) do { if (Cout == 1) DWV(T_, CI_, 1); else if (Cout == 2) DWV(T_, CI_, 2); else DWV(T_, CI_, 4); } while (0)
#define DWV_T(T_) do { if (Cin == 1) DWV_CO(T_, 1); else if (Cin == 2) DWV_CO(T_, 2); else DWV_CO(T_, 4); } while (0)
      if (dtype == EEGLDM_F32) DWV_T(float); else if (dtype == EEGLDM_F16) DWV_T(f16_t); else DWV_T(bf16_t);
#undef DWV_T
#undef DWV_CO
#undef DWV
      LAUNCH_CHECK();
      if (eeg_deterministic()) {      // partial rows in block order, one thread per element
        EEG_TRY(ew_fold_partials_det(ctx, parts, blocks, EP, 0, K * Cout * Cin, dw));
        if (dbias) EEG_TRY(ew_fold_partials_det(ctx, parts, blocks, EP, EW, Cout, dbias));
      } else {
        hipLaunchKernelGGL(dconv_tiny_fold_kernel, dim3(blocks >= 64 ? 8 : 1), dim3(NT), 0, ctx->stream, parts, blocks, EP, K * Cout * Cin, EW, Cout, dw, dbias);
        LAUNCH_CHECK();
      }
      if (bias_done && dbias) *bias_done = 1;
      return 0;
    }
    const int tb = eeg_deterministic() ? 1 : blocks;      // one atomic per element and block: a single block is a single writer
    if (dtype == EEGLDM_F32)
      hipLaunchKernelGGL((dconv_wgrad_tiny_kernel<float>), dim3(tb), dim3(NT), 0, ctx->stream, (const float*)x, ldx, (const float*)dy, lddy,
                         dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l);
    else if (dtype == EEGLDM_F16)
      hipLaunchKernelGGL((dconv_wgrad_tiny_kernel<f16_t>), dim3(tb), dim3(NT), 0, ctx->stream, (const f16_t*)x, ldx, (const f16_t*)dy, lddy,
                         dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l);
    else
      hipLaunchKernelGGL((dconv_wgrad_tiny_kernel<bf16_t>), dim3(tb), dim3(NT), 0, ctx->stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy,
                         dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l);
    LAUNCH_CHECK();
    return 0;
  }
  {
    const int G = dtype == EEGLDM_F32 ? 4 : 8;
    const bool wide_out = Cin <= 2 && Cout % G == 0 && Cout >= 16 && lddy % G == 0;
    const bool wide_in = Cout <= 2 && Cin % G == 0 && Cin >= 16 && ldx % G == 0;
    const int wide = wide_out ? Cout : Cin, lpr = wide / G;
    if ((wide_out || wide_in) && K <= 3 && lpr <= NT && NT % lpr == 0 && rows < (1L << 31)) {
      const int rpb = NT / lpr;
      // few blocks: every block ends with one atomic per weight element on the SAME addresses (2048 blocks cost 460 us in atomics)
      const long E = (long)K * Cout * Cin;
      constexpr bool in1out_ok = true;
      // blocks per sample: enough blocks to fill the chip four times over (each thread then has <= 8-16 independent row loads in flight)
      int nsegs = (int)((4L * ctx->num_cu + B - 1) / B); if (nsegs < 1) nsegs = 1;
      int segr = (Lout + nsegs - 1) / nsegs; segr = (segr + rpb - 1) / rpb * rpb; nsegs = (Lout + segr - 1) / segr;
      if (in1out_ok && wide_in && Cout == 1 && stride == 1 && Lin == Lout && Lout <= 1024 && pad_l <= 2 && K - 1 - pad_l <= 2 && dtype == EEGLDM_BF16 && B <= 65535 &&
          (size_t)B * nsegs * K * Cin * sizeof(float) <= (16u << 20)) {
        float* parts1 = (float*)((char*)ctx->scratch + (8u << 20));
        hipLaunchKernelGGL((dconv_wgrad_in1out_kernel<bf16_t>), dim3(B, nsegs), dim3(NT), 0, ctx->stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy, parts1, Lout, Cin, K, pad_l, segr);
        LAUNCH_CHECK();
        return ew_fold_partials(ctx, parts1, B * nsegs, (int)((long)K * Cin), dw);
      }
      long nb = (rows + rpb - 1) / rpb; const long capb = (long)ctx->num_cu * 4; if (nb > capb) nb = capb;   // 48 KB of LDS per block: 3 per CU
      float* parts = ((size_t)nb * E * sizeof(float) <= (16u << 20)) ? (float*)((char*)ctx->scratch + (8u << 20)) : nullptr;
      if (!parts && eeg_deterministic()) EEG_TRY(eeg_det_buffer(ctx, (size_t)nb * E * sizeof(float), &parts));
      if (!parts) { const long cap2 = (long)ctx->num_cu * 2; if (nb > cap2) nb = cap2; }
#define DWT(T_, WO_) hipLaunchKernelGGL((dconv_wgrad_wt_kernel<T_, WO_>), dim3((unsigned)nb), dim3(NT), 0, ctx->stream, (const T_*)x, ldx, (const T_*)dy, lddy, \
                                        dw, parts, B, Lout, Lin, Cout, Cin, K, stride, pad_l)
      if (dtype == EEGLDM_F32) { if (wide_out) DWT(float, true); else DWT(float, false); }
      else if (dtype == EEGLDM_F16) { if (wide_out) DWT(f16_t, true); else DWT(f16_t, false); }
      else { if (wide_out) DWT(bf16_t, true); else DWT(bf16_t, false); }
#undef DWT
      LAUNCH_CHECK();
      if (parts) EEG_TRY(ew_fold_partials(ctx, parts, (int)nb, (int)E, dw));
      return 0;
    }
  }
  const int rpc = 64;
  long blocks = (rows + rpc - 1) / rpc;
  const long cap = (long)ctx->num_cu * 4;
  if (blocks > cap) blocks = cap;
  if (eeg_deterministic()) blocks = 1;      // one atomic per element and block (generic fallback shape: rare, slow, ordered)
  if (dtype == EEGLDM_F32)
    hipLaunchKernelGGL((dconv_wgrad_kernel<float>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, (const float*)x, ldx,
                       (const float*)dy, lddy, dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l, rpc);
  else if (dtype == EEGLDM_F16)
    hipLaunchKernelGGL((dconv_wgrad_kernel<f16_t>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, (const f16_t*)x, ldx,
                       (const f16_t*)dy, lddy, dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l, rpc);
  else
    hipLaunchKernelGGL((dconv_wgrad_kernel<bf16_t>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, lddy, dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l, rpc);
  LAUNCH_CHECK();
  return 0;
}
