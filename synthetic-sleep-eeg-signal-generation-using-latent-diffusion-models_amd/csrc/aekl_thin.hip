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

namespace {
constexpr int NT = 512, NWAVE = NT / 64;
constexpr int MC = THIN_MAXC;
constexpr float GN_EPS_T = 1e-6f;
constexpr int RED_FLOATS = NWAVE * 56 + 64;

struct Bufs { float* b[THIN_NBUF]; float* red; };

// sum of NV per-thread values over the workgroup; results in red[NWAVE*NV + i] (valid for every thread after the call)
template <int NV> __device__ __forceinline__ void block_reduce(float (&v)[NV], float* red) {
#pragma unroll
  for (int i = 0; i < NV; i++) v[i] = wave_sum(v[i]);
  const int wave = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < NV; i++) red[wave * NV + i] = v[i];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWAVE; w++) s += red[w * NV + threadIdx.x];
    red[NWAVE * NV + threadIdx.x] = s;
  }
  __syncthreads();
}

__device__ __forceinline__ void load_w(const ThinOp& o, const float* __restrict__ P, int woff, int boff, float (&w)[MC][MC][3], float (&bias)[MC]) {
#pragma unroll
  for (int co = 0; co < MC; co++) {
    bias[co] = (boff >= 0 && co < o.cout) ? P[boff + co] : 0.f;
#pragma unroll
    for (int ci = 0; ci < MC; ci++)
#pragma unroll
      for (int k = 0; k < 3; k++) w[co][ci][k] = (co < o.cout && ci < o.cin && k < o.k) ? P[woff + (k * o.cout + co) * o.cin + ci] : 0.f;
  }
}

__device__ void f_conv(const ThinOp& o, const float* __restrict__ P, const Bufs& B) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst]; const float* A = o.add >= 0 ? B.b[o.add] : nullptr;
  float w[MC][MC][3], bias[MC];
  load_w(o, P, o.w, o.b, w, bias);
  for (int lo = threadIdx.x; lo < o.Lout; lo += NT) {
    float acc[MC];
#pragma unroll
    for (int co = 0; co < MC; co++) acc[co] = bias[co];
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int li = lo * o.stride + k - o.pad_l;
      if (k < o.k && li >= 0 && li < o.Lin) {
#pragma unroll
        for (int ci = 0; ci < MC; ci++) {
          if (ci < o.cin) {
            const float xv = X[ci * o.Lin + li];
#pragma unroll
            for (int co = 0; co < MC; co++) acc[co] = fmaf(w[co][ci][k], xv, acc[co]);
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < MC; co++) if (co < o.cout) Y[co * o.Lout + lo] = acc[co] + (A ? A[co * o.Lout + lo] : 0.f);
  }
  __syncthreads();
}

__device__ void f_gn(const ThinOp& o, const float* __restrict__ P, const Bufs& B, float* __restrict__ stats) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst];
  const int n = o.cin * o.Lin;
  float s[1] = {0.f};
  for (int i = threadIdx.x; i < n; i += NT) s[0] += X[i];
  block_reduce<1>(s, B.red);
  const float mean = B.red[NWAVE] / (float)n;
  float q[1] = {0.f};
  for (int i = threadIdx.x; i < n; i += NT) { const float d = X[i] - mean; q[0] = fmaf(d, d, q[0]); }
  block_reduce<1>(q, B.red);
  const float rstd = rsqrtf(B.red[NWAVE] / (float)n + GN_EPS_T);
  if (threadIdx.x == 0) { stats[2 * o.stat] = mean; stats[2 * o.stat + 1] = rstd; }
  for (int c = 0; c < o.cin; c++) {
    const float g = P[o.gw + c] * rstd, bb = P[o.gb + c] - mean * g;
    for (int l = threadIdx.x; l < o.Lin; l += NT) {
      const float z = fmaf(X[c * o.Lin + l], g, bb);
      Y[c * o.Lin + l] = o.silu ? silu_f(z) : z;
    }
  }
  __syncthreads();
}

// GroupNorm apply with saved statistics (backward: recompute a conv's input)
__device__ void b_recomp(const ThinOp& o, const float* __restrict__ P, const Bufs& B, const float* __restrict__ stats) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst];
  const float mean = stats[2 * o.stat], rstd = stats[2 * o.stat + 1];
  for (int c = 0; c < o.cin; c++) {
    const float g = P[o.gw + c] * rstd, bb = P[o.gb + c] - mean * g;
    for (int l = threadIdx.x; l < o.Lin; l += NT) {
      const float z = fmaf(X[c * o.Lin + l], g, bb);
      Y[c * o.Lin + l] = o.silu ? silu_f(z) : z;
    }
  }
  __syncthreads();
}

__device__ void d_ups(const ThinOp& o, const Bufs& B) {
  const float* X = B.b[o.src]; float* Y = B.b[o.dst];
  for (int i = threadIdx.x; i < o.cin * o.Lout; i += NT) { const int c = i / o.Lout, l = i - c * o.Lout; Y[i] = X[c * o.Lin + (l >> 1)]; }
  __syncthreads();
}

__device__ void f_heads(const ThinOp& o, const float* __restrict__ P, const Bufs& B, float* __restrict__ tape, const int* __restrict__ tape_off,
                        const float* __restrict__ eps, float* __restrict__ z_mu, float* __restrict__ z_sigma, float* __restrict__ kl, float inv_B) {
  const float* H = B.b[o.src]; float* Z = B.b[o.dst];
  const int lat = o.cin, L = o.Lin;
  float* tmu = tape + tape_off[o.save]; float* tlv = tape + tape_off[o.save + 1];
  float part[1] = {0.f};
  for (int l = threadIdx.x; l < L; l += NT) {
    float h[MC];
#pragma unroll
    for (int ci = 0; ci < MC; ci++) h[ci] = ci < lat ? H[ci * L + l] : 0.f;
#pragma unroll
    for (int co = 0; co < MC; co++) {
      if (co < lat) {
        float mu = P[o.b + co], lv = P[o.b2 + co];
#pragma unroll
        for (int ci = 0; ci < MC; ci++) if (ci < lat) { mu = fmaf(P[o.w + co * lat + ci], h[ci], mu); lv = fmaf(P[o.w2 + co * lat + ci], h[ci], lv); }
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
  }
  block_reduce<1>(part, B.red);
  if (kl && threadIdx.x == 0) atomicAdd(kl, B.red[NWAVE] * inv_B);
}

// ------------------------------------------------------------------ backward pieces
__device__ void b_conv(const ThinOp& o, const float* __restrict__ P, float* __restrict__ G, const Bufs& B) {
  const float* dY = B.b[o.src]; float* A = B.b[o.act];
  float w[MC][MC][3], bias[MC];
  load_w(o, P, o.w, -1, w, bias);
  // ---- dW / db
  float acc[MC * MC * 3 + MC];
#pragma unroll
  for (int i = 0; i < MC * MC * 3 + MC; i++) acc[i] = 0.f;
  for (int lo = threadIdx.x; lo < o.Lout; lo += NT) {
    float dy[MC];
#pragma unroll
    for (int co = 0; co < MC; co++) { dy[co] = co < o.cout ? dY[co * o.Lout + lo] : 0.f; acc[MC * MC * 3 + co] += dy[co]; }
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int li = lo * o.stride + k - o.pad_l;
      if (k < o.k && li >= 0 && li < o.Lin) {
#pragma unroll
        for (int ci = 0; ci < MC; ci++) {
          if (ci < o.cin) {
            const float xv = A[ci * o.Lin + li];
#pragma unroll
            for (int co = 0; co < MC; co++) acc[(co * MC + ci) * 3 + k] = fmaf(dy[co], xv, acc[(co * MC + ci) * 3 + k]);
          }
        }
      }
    }
  }
  block_reduce<MC * MC * 3 + MC>(acc, B.red);       // ends with a barrier: every read of A above is done
  const float* R = B.red + NWAVE * (MC * MC * 3 + MC);
  if (threadIdx.x < MC * MC * 3) {
    const int co = threadIdx.x / (MC * 3), ci = (threadIdx.x / 3) % MC, k = threadIdx.x % 3;
    if (co < o.cout && ci < o.cin && k < o.k) atomicAdd(G + o.w + (k * o.cout + co) * o.cin + ci, R[threadIdx.x]);
  } else if (threadIdx.x < MC * MC * 3 + MC) {
    const int co = threadIdx.x - MC * MC * 3;
    if (o.b >= 0 && co < o.cout) atomicAdd(G + o.b + co, R[threadIdx.x]);
  }
  // ---- dX -> overwrites the activation buffer
  if (o.need_dx) {
    for (int li = threadIdx.x; li < o.Lin; li += NT) {
      float dx[MC];
#pragma unroll
      for (int ci = 0; ci < MC; ci++) dx[ci] = 0.f;
#pragma unroll
      for (int k = 0; k < 3; k++) {
        const int t = li + o.pad_l - k;
        if (k < o.k && t >= 0 && (o.stride == 1 || (t & 1) == 0)) {
          const int lo = o.stride == 1 ? t : t >> 1;
          if (lo < o.Lout) {
#pragma unroll
            for (int co = 0; co < MC; co++) {
              if (co < o.cout) {
                const float dy = dY[co * o.Lout + lo];
#pragma unroll
                for (int ci = 0; ci < MC; ci++) dx[ci] = fmaf(w[co][ci][k], dy, dx[ci]);
              }
            }
          }
        }
      }
#pragma unroll
      for (int ci = 0; ci < MC; ci++) if (ci < o.cin) A[ci * o.Lin + li] = dx[ci];
    }
  }
  __syncthreads();
}

__device__ void b_gn(const ThinOp& o, const float* __restrict__ P, float* __restrict__ G, const Bufs& B, const float* __restrict__ stats) {
  const float* X = B.b[o.act]; const float* dY = B.b[o.src]; float* dX = B.b[o.dst]; const float* ADD = o.add >= 0 ? B.b[o.add] : nullptr;
  const float mean = stats[2 * o.stat], rstd = stats[2 * o.stat + 1];
  const int L = o.Lin, C = o.cin;
  float acc[2 * MC];
#pragma unroll
  for (int i = 0; i < 2 * MC; i++) acc[i] = 0.f;
  float ga[MC], be[MC];
#pragma unroll
  for (int c = 0; c < MC; c++) { ga[c] = c < C ? P[o.gw + c] : 0.f; be[c] = c < C ? P[o.gb + c] : 0.f; }
#pragma unroll
  for (int c = 0; c < MC; c++) {
    if (c < C) {
      for (int l = threadIdx.x; l < L; l += NT) {
        const float xh = (X[c * L + l] - mean) * rstd;
        const float dz = dY[c * L + l] * (o.silu ? silu_grad_f(fmaf(xh, ga[c], be[c])) : 1.0f);
        acc[c] = fmaf(dz, xh, acc[c]); acc[MC + c] += dz;
      }
    }
  }
  block_reduce<2 * MC>(acc, B.red);
  const float* R = B.red + NWAVE * 2 * MC;
  if (threadIdx.x < MC) { if (threadIdx.x < C) atomicAdd(G + o.gw + threadIdx.x, R[threadIdx.x]); }
  else if (threadIdx.x < 2 * MC) { if (threadIdx.x - MC < C) atomicAdd(G + o.gb + threadIdx.x - MC, R[threadIdx.x]); }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int c = 0; c < MC; c++) { s1 = fmaf(ga[c], R[MC + c], s1); s2 = fmaf(ga[c], R[c], s2); }
  const float inv_n = 1.0f / (float)(C * L);
  const float m1 = s1 * inv_n, m2 = s2 * inv_n;
#pragma unroll
  for (int c = 0; c < MC; c++) {
    if (c < C) {
      for (int l = threadIdx.x; l < L; l += NT) {
        const float xh = (X[c * L + l] - mean) * rstd;
        const float dz = dY[c * L + l] * (o.silu ? silu_grad_f(fmaf(xh, ga[c], be[c])) : 1.0f);
        dX[c * L + l] = rstd * (dz * ga[c] - m1 - xh * m2) + (ADD ? ADD[c * L + l] : 0.f);
      }
    }
  }
  __syncthreads();
}

__device__ void b_heads(const ThinOp& o, const float* __restrict__ P, float* __restrict__ G, const Bufs& B, const float* __restrict__ tape,
                        const int* __restrict__ tape_off, const float* __restrict__ eps, float klw_over_B) {
  float* DZ = B.b[o.src]; float* DH = B.b[o.dst]; const float* H = B.b[o.act];
  const int lat = o.cin, L = o.Lin;
  const float* tmu = tape + tape_off[o.save]; const float* tlv = tape + tape_off[o.save + 1];
  constexpr int NV = 2 * (MC * MC + MC);
  float acc[NV];            // [dWmu co][ci], [dbmu], [dWlv], [dblv]
#pragma unroll
  for (int i = 0; i < NV; i++) acc[i] = 0.f;
  for (int l = threadIdx.x; l < L; l += NT) {
    float h[MC], dmu[MC], dlv[MC], dh[MC];
#pragma unroll
    for (int c = 0; c < MC; c++) { h[c] = c < lat ? H[c * L + l] : 0.f; dh[c] = 0.f; dmu[c] = 0.f; dlv[c] = 0.f; }
#pragma unroll
    for (int co = 0; co < MC; co++) {
      if (co < lat) {
        const float dz = DZ[co * L + l], mu = tmu[co * L + l], lv = tlv[co * L + l];
        const bool inside = lv > -30.f && lv < 20.f;             // clamp passes the gradient strictly inside (torch.clamp)
        const float lvc = fminf(20.f, fmaxf(-30.f, lv));
        const float sg = __expf(0.5f * lvc);
        const float e = eps ? eps[co * L + l] : 0.f;
        dmu[co] = fmaf(klw_over_B, mu, dz);                       // z = mu + eps sigma ; KL: d/dmu = mu
        // d/dlv: z -> dz * eps * sigma / 2 ; KL 0.5 (sigma^2 - lv - 1) -> 0.5 (sigma^2 - 1)
        dlv[co] = inside ? fmaf(dz * e, 0.5f * sg, klw_over_B * 0.5f * (sg * sg - 1.0f)) : 0.f;
      }
    }
#pragma unroll
    for (int co = 0; co < MC; co++) {
      if (co < lat) {
        acc[MC * MC + co] += dmu[co]; acc[MC * MC + MC + MC * MC + co] += dlv[co];
#pragma unroll
        for (int ci = 0; ci < MC; ci++) {
          if (ci < lat) {
            acc[co * MC + ci] = fmaf(dmu[co], h[ci], acc[co * MC + ci]);
            acc[MC * MC + MC + co * MC + ci] = fmaf(dlv[co], h[ci], acc[MC * MC + MC + co * MC + ci]);
            dh[ci] = fmaf(P[o.w + co * lat + ci], dmu[co], dh[ci]);
            dh[ci] = fmaf(P[o.w2 + co * lat + ci], dlv[co], dh[ci]);
          }
        }
      }
    }
#pragma unroll
    for (int ci = 0; ci < MC; ci++) if (ci < lat) DH[ci * L + l] = dh[ci];
  }
  block_reduce<NV>(acc, B.red);
  const float* R = B.red + NWAVE * NV;
  const int t = threadIdx.x;
  if (t < MC * MC) { const int co = t / MC, ci = t % MC; if (co < lat && ci < lat) atomicAdd(G + o.w + co * lat + ci, R[t]); }
  else if (t < MC * MC + MC) { const int co = t - MC * MC; if (co < lat) atomicAdd(G + o.b + co, R[t]); }
  else if (t < 2 * MC * MC + MC) { const int u = t - MC * MC - MC, co = u / MC, ci = u % MC; if (co < lat && ci < lat) atomicAdd(G + o.w2 + co * lat + ci, R[t]); }
  else if (t < NV) { const int co = t - 2 * MC * MC - MC; if (co < lat) atomicAdd(G + o.b2 + co, R[t]); }
  __syncthreads();
}

__device__ __forceinline__ Bufs make_bufs(char* smem, int maxt) {
  Bufs B;
#pragma unroll
  for (int i = 0; i < THIN_NBUF; i++) B.b[i] = (float*)smem + (size_t)i * maxt;
  B.red = (float*)smem + (size_t)THIN_NBUF * maxt;
  return B;
}

__global__ __launch_bounds__(NT) void thin_fwd_kernel(const ThinOp* __restrict__ ops, int nops, const int* __restrict__ tape_off, int tape_stride, int nstat,
                                                      int maxt, const float* __restrict__ P, const float* __restrict__ x, const float* __restrict__ eps,
                                                      float* __restrict__ recon, float* __restrict__ z_mu, float* __restrict__ z_sigma, float* __restrict__ kl,
                                                      float* __restrict__ tape_all, float* __restrict__ stats_all, int lat, int Ll, float inv_B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Bufs B = make_bufs(smem, maxt);
  const int b = blockIdx.x;
  float* tape = tape_all + (size_t)b * tape_stride; float* stats = stats_all + (size_t)b * nstat * 2;
  for (int i = 0; i < nops; i++) {
    const ThinOp o = ops[i];
    switch (o.kind) {
      case TF_LOAD: {
        const float* src = x + (size_t)b * o.cin * o.Lin; float* D = B.b[o.dst];
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = src[j];
        __syncthreads();
      } break;
      case TF_CONV: f_conv(o, P, B); break;
      case TF_GN: f_gn(o, P, B, stats); break;
      case TF_UPS: d_ups(o, B); break;
      case TF_SAVE: {
        const float* S = B.b[o.src]; float* D = tape + tape_off[o.save];
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = S[j];
        __syncthreads();       // the next op may overwrite the saved buffer
      } break;
      case TF_HEADS:
        f_heads(o, P, B, tape, tape_off, eps ? eps + (size_t)b * lat * Ll : nullptr, z_mu ? z_mu + (size_t)b * lat * Ll : nullptr,
                z_sigma ? z_sigma + (size_t)b * lat * Ll : nullptr, kl, inv_B);
        break;
      case TF_STORE: {
        const float* S = B.b[o.src]; float* D = recon + (size_t)b * o.cin * o.Lin;
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = S[j];
      } break;
    }
  }
}

__global__ __launch_bounds__(NT) void thin_bwd_kernel(const ThinOp* __restrict__ ops, int nops, const int* __restrict__ tape_off, int tape_stride, int nstat,
                                                      int maxt, const float* __restrict__ P, float* __restrict__ G, const float* __restrict__ d_recon,
                                                      const float* __restrict__ eps, float* __restrict__ dx_out, const float* __restrict__ tape_all,
                                                      const float* __restrict__ stats_all, int lat, int Ll, float klw_over_B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const Bufs B = make_bufs(smem, maxt);
  const int b = blockIdx.x;
  const float* tape = tape_all + (size_t)b * tape_stride; const float* stats = stats_all + (size_t)b * nstat * 2;
  for (int i = 0; i < nops; i++) {
    const ThinOp o = ops[i];
    switch (o.kind) {
      case TB_LOADDY: {
        const float* src = d_recon + (size_t)b * o.cin * o.Lin; float* D = B.b[o.dst];
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = src[j];
        __syncthreads();
      } break;
      case TB_LOADT: {
        const float* src = tape + tape_off[o.save]; float* D = B.b[o.dst];
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = src[j];
        __syncthreads();
      } break;
      case TB_RECOMP: b_recomp(o, P, B, stats); break;
      case TB_UPS: d_ups(o, B); break;
      case TB_CONV: b_conv(o, P, G, B); break;
      case TB_GN: b_gn(o, P, G, B, stats); break;
      case TB_UPSBWD: {
        const float* S = B.b[o.src]; float* D = B.b[o.dst];      // src: cin x Lin (= 2 Lout), dst: cin x Lout
        for (int j = threadIdx.x; j < o.cin * o.Lout; j += NT) { const int c = j / o.Lout, l = j - c * o.Lout; D[j] = S[c * o.Lin + 2 * l] + S[c * o.Lin + 2 * l + 1]; }
        __syncthreads();
      } break;
      case TB_COPY: {
        const float* S = B.b[o.src]; float* D = B.b[o.dst];
        for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = S[j];
        __syncthreads();
      } break;
      case TB_HEADS: b_heads(o, P, G, B, tape, tape_off, eps ? eps + (size_t)b * lat * Ll : nullptr, klw_over_B); break;
      case TB_STOREDX: {
        if (dx_out) {
          const float* S = B.b[o.src]; float* D = dx_out + (size_t)b * o.cin * o.Lin;
          for (int j = threadIdx.x; j < o.cin * o.Lin; j += NT) D[j] = S[j];
        }
      } break;
    }
  }
}

size_t lds_bytes(const ThinProgram& p) { return sizeof(float) * ((size_t)THIN_NBUF * p.maxt + RED_FLOATS + 64); }
}  // namespace

int thin_upload(ThinProgram* p) {
  EEG_CHECK(p->maxt > 0 && p->maxt <= THIN_MAX_FLOATS, "thin autoencoder: tensor of %d values does not fit the LDS buffers", p->maxt);
  thin_free(p);
  HIP_TRY(hipMalloc(&p->d_fwd, sizeof(ThinOp) * p->fwd.size()));
  HIP_TRY(hipMalloc(&p->d_bwd, sizeof(ThinOp) * p->bwd.size()));
  HIP_TRY(hipMalloc(&p->d_tape_off, sizeof(int) * p->tape_off.size()));
  HIP_TRY(hipMemcpy(p->d_fwd, p->fwd.data(), sizeof(ThinOp) * p->fwd.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_bwd, p->bwd.data(), sizeof(ThinOp) * p->bwd.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(p->d_tape_off, p->tape_off.data(), sizeof(int) * p->tape_off.size(), hipMemcpyHostToDevice));
  HIP_TRY(hipFuncSetAttribute((const void*)thin_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(*p)));
  HIP_TRY(hipFuncSetAttribute((const void*)thin_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes(*p)));
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
  hipLaunchKernelGGL(thin_fwd_kernel, dim3(B), dim3(NT), lds_bytes(p), ctx->stream, p.d_fwd, (int)p.fwd.size(), p.d_tape_off, p.tape_stride, p.nstat, p.maxt,
                     params, x, eps, recon, z_mu, z_sigma, kl, p.tape, p.stats, p.lat, p.Ll, 1.0f / (float)B);
  LAUNCH_CHECK();
  return 0;
}
int thin_backward(eegldm_ctx* ctx, const ThinProgram& p, const float* params, float* grads, const float* d_recon, const float* eps, float klw_over_B,
                  float* dx, int B) {
  EEG_CHECK(p.d_bwd && p.tape && p.stats, "thin program not prepared");
  hipLaunchKernelGGL(thin_bwd_kernel, dim3(B), dim3(NT), lds_bytes(p), ctx->stream, p.d_bwd, (int)p.bwd.size(), p.d_tape_off, p.tape_stride, p.nstat, p.maxt,
                     params, grads, d_recon, eps, dx, p.tape, p.stats, p.lat, p.Ll, klw_over_B);
  LAUNCH_CHECK();
  return 0;
}
