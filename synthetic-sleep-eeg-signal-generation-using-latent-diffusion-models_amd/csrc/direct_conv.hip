// Direct (non-MFMA) 1-D convolution kernels for "thin" layers where one side has <= 8
// channels: UNet conv_in / conv_out (unet.py:385, :504), every layer of the [2,2,4]
// AutoencoderKL (config_aekl_eeg_2_2_4_spec.yaml), the latent heads, the first and last
// PatchDiscriminator convs.  These are HBM/launch-bound (a few MACs per byte): the kernels
// keep global accesses coalesced along the wide channel side and broadcast the thin side.
// Layout NLC; weights packed [K][Cout][Cin] in the activation dtype; bias/grad fp32.
#include "common.h"

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

int dconv_run(eegldm_ctx* ctx, int dtype, bool dgrad, const void* in, long ldin, const void* w, const float* bias,
              const void* resid, long ldr, void* out, long ldout, int B, int Lin, int Lout, int Cin, int Cout, int K,
              int stride, int pad_l) {
  DArgs a;
  a.in = in; a.ldin = ldin; a.w = w; a.bias = bias; a.resid = resid; a.ldr = ldr; a.out = out; a.ldout = ldout;
  a.B = B; a.Cout = Cout; a.Cin = Cin; a.K = K; a.stride = stride; a.pad_l = pad_l; a.dgrad = dgrad ? 1 : 0;
  if (!dgrad) { a.Lo = Lout; a.Li = Lin; a.Co = Cout; a.Ci = Cin; }
  else        { a.Lo = Lin;  a.Li = Lout; a.Co = Cin;  a.Ci = Cout; }
  const bool rowdot = a.Ci > 8 && a.Co <= 8;
  EEG_CHECK(a.Ci <= 8 || a.Co <= 8, "direct conv expects a thin side (Cin=%d Cout=%d)", Cin, Cout);
  const long rows = (long)B * a.Lo;
  if (dtype == EEGLDM_F32) {
    if (rowdot) hipLaunchKernelGGL((dconv_rowdot_kernel<float>), dim3(grid_cap((rows + 3) / 4, ctx)), dim3(NT), 0, ctx->stream, a);
    else hipLaunchKernelGGL((dconv_point_kernel<float>), dim3(grid_cap((rows * a.Co + NT - 1) / NT, ctx)), dim3(NT), 0, ctx->stream, a);
  } else {
    if (rowdot) hipLaunchKernelGGL((dconv_rowdot_kernel<bf16_t>), dim3(grid_cap((rows + 3) / 4, ctx)), dim3(NT), 0, ctx->stream, a);
    else hipLaunchKernelGGL((dconv_point_kernel<bf16_t>), dim3(grid_cap((rows * a.Co + NT - 1) / NT, ctx)), dim3(NT), 0, ctx->stream, a);
  }
  LAUNCH_CHECK();
  return 0;
}

int dconv_wgrad(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* dy, long lddy, float* dw, int B, int Lin,
                int Lout, int Cin, int Cout, int K, int stride, int pad_l) {
  const long rows = (long)B * Lout;
  if (Cin <= 4 && Cout <= 4 && K <= 3) {
    const int blocks = grid_cap((rows + NT - 1) / NT, ctx) / 4 + 1;
    if (dtype == EEGLDM_F32)
      hipLaunchKernelGGL((dconv_wgrad_tiny_kernel<float>), dim3(blocks), dim3(NT), 0, ctx->stream, (const float*)x, ldx, (const float*)dy, lddy,
                         dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l);
    else
      hipLaunchKernelGGL((dconv_wgrad_tiny_kernel<bf16_t>), dim3(blocks), dim3(NT), 0, ctx->stream, (const bf16_t*)x, ldx, (const bf16_t*)dy, lddy,
                         dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l);
    LAUNCH_CHECK();
    return 0;
  }
  const int rpc = 64;
  long blocks = (rows + rpc - 1) / rpc;
  const long cap = (long)ctx->num_cu * 4;
  if (blocks > cap) blocks = cap;
  if (dtype == EEGLDM_F32)
    hipLaunchKernelGGL((dconv_wgrad_kernel<float>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, (const float*)x, ldx,
                       (const float*)dy, lddy, dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l, rpc);
  else
    hipLaunchKernelGGL((dconv_wgrad_kernel<bf16_t>), dim3((unsigned)blocks), dim3(NT), 0, ctx->stream, (const bf16_t*)x, ldx,
                       (const bf16_t*)dy, lddy, dw, B, Lout, Lin, Cout, Cin, K, stride, pad_l, rpc);
  LAUNCH_CHECK();
  return 0;
}
