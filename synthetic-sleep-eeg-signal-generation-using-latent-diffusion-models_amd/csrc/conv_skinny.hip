// Few-row convolution / linear layer (bf16 or fp16 operands, 1 or 3 taps, stride 1): the forward GEMMs of the UNet when a launch has only a few
// hundred rows -- sampling ONE window per call, as the reference's sampler does (sample_trials.py:149-163: 50 DDIM steps on a
// (1, 1, 768) latent).
//
// At B = 1 the general implicit-GEMM kernel (gemm.hip) has 2-12 tiles for 256 CUs and each tile walks its whole reduction dimension
// alone: a 512 -> 512 k3 conv over 192 rows is 8 blocks x 48 K stages = 18 us, a 1024 -> 512 one 30 us, where a trivial kernel takes
// 4.7 us from launch to completion (rocprofv3 trace of the B = 1 chain: 42 + 23 such launches per UNet forward = 0.9 of 1.66 ms).
// Here the launch is cut the other way:
//   * one block = a (16 RF rows) x (16 CF columns) output tile, EIGHT waves that split the REDUCTION dimension: wave w takes the
//     32-channel chunks w, w + 8, ... of every tap; a 3 x 512-long reduction is 6 MFMA steps per wave instead of 48 stages per block;
//   * operands go straight from L2 / HBM into MFMA fragments (16-byte loads in the fragment layout, every load of a wave's units
//     issued before the first MFMA): no LDS staging, no barrier inside the reduction;
//   * the eight partial tiles meet in LDS (one barrier), bias + time-embedding row + residual are added once, one rounding to bf16.
// 192-288 blocks per launch instead of 2-12.  Weights are re-read by every row tile and activations by every column tile (from L2:
// 20-50 MB per launch); that is the price of the short dependency chain and why this kernel is only chosen for launches the
// general kernel cannot fill (conv_skinny_try).
#include <stdlib.h>

#include "common.h"
#include "internal.h"

namespace {
constexpr int SK_WAVES = 8;
constexpr int SK_LOADS = 24;                // 16-byte loads in flight per lane and round (96 VGPRs); a unit is RF + CF of them

typedef unsigned sk_u32x2 __attribute__((ext_vector_type(2)));

struct SkArgs {
  const bf16_t* x; long ldx;
  const bf16_t* w; long sWt; int Cin;       // w[tap * sWt + n * Cin + k]   (the plain packed layout [tap][Cout][Cin])
  const float* bias; const float* rowvec; long ld_rowvec;
  const bf16_t* resid; long ldr;
  bf16_t* y; long ldy;
  int M, L, N;
  // GroupNorm statistics out (producer side): every 16-row x 4-channel lane quad of the output leaves (sum, sum of squares) of its bf16-ROUNDED
  // values -- what a GroupNorm kernel reading the stored tensor would see -- in its own slot part_out[(sample * L/16 + row fragment) * N/4 + quad]:
  // plain stores, no zeroing, no contention (fp64 atomics on the 64 (sample, group) sums cost the producer +6 us: ~125 ns per same-address atomic)
  float2* part_out;
  // GroupNorm (+ SiLU) on the activation operand (consumer side, GN kernels): x is the RAW tensor, gn_part its producer's slots
  // (a concatenated operand [h | skip] has two producers: quads below gn_nqa come from gn_part, the others from gn_part_b)
  const float2* gn_part; const float2* gn_part_b; int gn_nqa; const float* gn_gamma; const float* gn_beta; int gn_cpg; float gn_eps; int gn_silu;
  // K extension (round 5): + sum_k x2[r][k] * w2[n][k] + bias2[n] -- the ResBlock's skip_connection 1 x 1 conv inside its second conv's launch
  // (h = skip_connection(x) + out_layers(h), unet.py:302,327): a second, one-tap reduction over a RAW second operand into the same accumulators.
  // One launch less per skip block in the 72-83-launch chain of a one-window forward; no intermediate tensor, one rounding.
  const bf16_t* x2; long ldx2; const bf16_t* w2; int Cin2; const float* bias2;
};

// statistics slot of (16-row fragment rt, 4-channel quad qd) of sample `b` of a tensor whose channels [0, 4 nqa) were written by one producer
// (slots pa, nqa quads per row fragment) and the rest by another (pb, nqb)
__device__ __forceinline__ float2 sk_slot(const float2* pa, const float2* pb, int nqa, int nqb, long b, int rts, int rt, int qd) {
  return qd < nqa ? pa[(b * rts + rt) * nqa + qd] : pb[(b * rts + rt) * nqb + (qd - nqa)];
}

constexpr int SK_GN_MAXC = 1024;            // widest normalised operand (scale / shift table in LDS)

// 8 bf16 activations -> GroupNorm scale / shift (+ SiLU) -> 8 bf16, the rounding the stand-alone GroupNorm kernel applies to its output
template <typename T16>
__device__ __forceinline__ f32x4 sk_mma(const uint4& a, const uint4& b, const f32x4& c) {
  if constexpr (Is16<T16>::f16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <typename T16>
__device__ __forceinline__ uint4 sk_norm8(uint4 a, const float* sc, const float* sh, bool silu) {
  const unsigned in[4] = {a.x, a.y, a.z, a.w}; unsigned out[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    float z0 = w16_lo<T16>(in[j]) * sc[2 * j] + sh[2 * j], z1 = w16_hi<T16>(in[j]) * sc[2 * j + 1] + sh[2 * j + 1];
    if (silu) { z0 = silu_f(z0); z1 = silu_f(z1); }
    out[j] = pack16x2<T16>(z0, z1);
  }
  return make_uint4(out[0], out[1], out[2], out[3]);
}

// rotate a fragment by one lane inside every 16-lane row (the 16 rows of a fragment): CTRL 0x121 = row_ror:1 (lane i <- lane i - 1),
// 0x12F = row_ror:15 (lane i <- lane i + 1).  A 3-tap conv's shifted operand fragments are rotations of the centre ones.
template <int CTRL>
__device__ __forceinline__ uint4 sk_rot(uint4 v) {
  return make_uint4((unsigned)__builtin_amdgcn_update_dpp(0, (int)v.x, CTRL, 0xf, 0xf, false), (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.y, CTRL, 0xf, 0xf, false),
                    (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.z, CTRL, 0xf, 0xf, false), (unsigned)__builtin_amdgcn_update_dpp(0, (int)v.w, CTRL, 0xf, 0xf, false));
}

template <int TAPS, int RF, int CF, bool GN, typename T16 = bf16_t>
__global__ __launch_bounds__(64 * SK_WAVES) void conv_skinny_kernel(const SkArgs p) {
  __shared__ f32x4 part[SK_WAVES][RF * CF][64];
  __shared__ float2 gn_ss[GN ? SK_GN_MAXC : 1];            // (scale, shift) per input channel of this block's sample
  __shared__ float2 gn_mr[GN ? 64 : 1];                    // (mean, rstd) per group
  const int tid = threadIdx.x, lane = tid & 63, lm = lane & 15, q = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);            // scalar: the chunk walk below is wave-uniform
  const int m0 = blockIdx.x * (16 * RF), n0 = blockIdx.y * (16 * CF);
  const int kchunks = p.Cin >> 5;

  // ---- the epilogue's addends first (waves 0 .. RF*CF-1 own one output fragment each): their latency hides under the reduction.
  // Unconditional loads from addresses that are always legal; what a null pointer or an out-of-range lane fetched is never used.
  const bool owner = wave < RF * CF;
  const int f_own = owner ? wave : 0, rf_own = f_own / CF, cf_own = f_own - rf_own * CF;
  const int m_own = m0 + rf_own * 16 + lm, n_own = n0 + cf_own * 16 + q * 4;
  const bool own_ok = owner && m_own < p.M && n_own < p.N;               // N % 4 == 0 (checked by the host)
  const int mc = m_own < p.M ? m_own : p.M - 1, nc = n_own < p.N ? n_own : 0;
  // issued as inline assembly AFTER the first round's operand loads: as compiler-visible loads in front of the reduction they were
  // retired with a vmcnt(0) before the first operand load went out (one full memory latency).  Loads retire in order, so hipcc's
  // counted waits for its own (older) loads stay sufficient with these three behind them; the explicit wait is before the epilogue.
  f32x4 e_bias, e_row, e_bias2; sk_u32x2 e_res;
  const void* pb2 = p.bias2 ? (const void*)(p.bias2 + nc) : (const void*)p.x;
  const void* pb = p.bias ? (const void*)(p.bias + nc) : (const void*)p.x;
  const void* pr = p.rowvec ? (const void*)(p.rowvec + (long)(mc / p.L) * p.ld_rowvec + nc) : (const void*)p.x;
  const void* ps = p.resid ? (const void*)(p.resid + (long)mc * p.ldr + nc) : (const void*)p.x;

  // row bookkeeping per row fragment: the lane's (clamped) output row and its position inside its sample
  int row[RF], pos[RF];
#pragma unroll
  for (int rf = 0; rf < RF; rf++) {
    const int m = m0 + rf * 16 + lm;
    row[rf] = m < p.M ? m : p.M - 1;
    pos[rf] = row[rf] % p.L;
  }
  const bf16_t* wrow[CF];
#pragma unroll
  for (int cf = 0; cf < CF; cf++) { const int n = n0 + cf * 16 + lm; wrow[cf] = p.w + (long)(n < p.N ? n : p.N - 1) * p.Cin + q * 8; }
  const bf16_t* xrow[TAPS][RF]; bool xok[TAPS][RF];
#pragma unroll
  for (int t = 0; t < TAPS; t++)
#pragma unroll
    for (int rf = 0; rf < RF; rf++) {
      const int d = TAPS == 3 ? t - 1 : 0, l = pos[rf] + d;
      xok[t][rf] = l >= 0 && l < p.L;                                    // else: the conv's zero padding at the sample's ends
      int r = row[rf] + d; r = r < 0 ? 0 : (r >= p.M ? p.M - 1 : r);
      xrow[t][rf] = p.x + (long)r * p.ldx + q * 8;
    }

  // 3 taps: only the CENTRE rows and one halo fragment are loaded (lane 0 of every 16-lane row: the row above the tile, lane 15: the
  // row below it); the two shifted operand fragments of a tap are lane rotations of the centre ones (a third of the activation
  // loads, and of the GroupNorm / SiLU work when the operand is normalised on load)
  const bf16_t* xhalo;
  {
    int r = lm == 0 ? row[0] - 1 : (lm == 15 ? row[RF - 1] + 1 : row[0]); r = r < 0 ? 0 : (r >= p.M ? p.M - 1 : r);
    xhalo = p.x + (long)r * p.ldx + q * 8;
  }
  f32x4 acc[RF][CF];
#pragma unroll
  for (int rf = 0; rf < RF; rf++)
#pragma unroll
    for (int cf = 0; cf < CF; cf++) acc[rf][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // wave w reduces the 32-channel chunks w, w + 8, ... (all taps of a chunk); CH chunks per round
  constexpr int LPC = TAPS == 3 ? RF + 1 + 3 * CF : RF + CF;           // loads per chunk
  // (at most 2 chunks per round: with 16 chunks -- 512 channels -- a wave owns exactly two, and a longer round would only fetch
  // clamped duplicates: the loads are unconditional.  1 x 1 kernels ran rounds of 6-8 chunks before: 7.4 -> see DESIGN.md 3.3)
  constexpr int CH0 = SK_LOADS / LPC < 1 ? 1 : SK_LOADS / LPC, CH = CH0 > 2 ? 2 : CH0;
  // GroupNorm on load: gamma / beta of this thread's channels now, so that only ONE memory latency (the statistics slots, overlapped
  // with the operand loads) lies between the loads and the first MFMA
  float g_ga[2] = {0.f, 0.f}, g_be[2] = {0.f, 0.f};
  // ... and the first 48 statistics slots of this thread's group (16 threads per group, 3 each: all of them for this UNet) go out
  // BEFORE the operand loads: they come back first, and the fold runs while the operands are still in flight
  float2 pv[3] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
  if (GN) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int ch = tid + j * 64 * SK_WAVES;
      if (ch < p.Cin) { g_ga[j] = p.gn_gamma[ch]; g_be[j] = p.gn_beta[ch]; }
    }
    const int G = p.Cin / p.gn_cpg, qpg = p.gn_cpg >> 2, rts = p.L >> 4, S = rts * qpg, nq = p.Cin >> 2, g = tid >> 4;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const int i = (tid & 15) + 16 * k;
      if (g < G && i < S) { const int rt = i / qpg; pv[k] = sk_slot(p.gn_part, p.gn_part_b, p.gn_nqa, nq - p.gn_nqa, m0 / p.L, rts, rt, g * qpg + (i - rt * qpg)); }
    }
  }
  // Rounds: the first one is straight-line code (a loop header costs a conservative vmcnt(0) before the loads), the rest -- only
  // reductions longer than 8 * CH chunks have any -- is a loop.
#define SK_ROUND(C0, FIRST)                                                                                                          \
  {                                                                                                                                  \
    const int c0 = (C0);                                                                                                             \
    uint4 xc[CH][RF], xh[CH], wb[CH][TAPS][CF];                                                                                      \
    _Pragma("unroll") for (int i = 0; i < CH; i++) {                                                                                 \
      int c = c0 + i * SK_WAVES; if (c >= kchunks) c = kchunks - 1; if (c < 0) c = 0; /* clamped: surplus products are skipped */    \
      _Pragma("unroll") for (int rf = 0; rf < RF; rf++) xc[i][rf] = *(const uint4*)(xrow[TAPS / 2][rf] + c * 32);                    \
      if (TAPS == 3) xh[i] = *(const uint4*)(xhalo + c * 32);                                                                        \
      _Pragma("unroll") for (int t = 0; t < TAPS; t++)                                                                               \
        _Pragma("unroll") for (int cf = 0; cf < CF; cf++) wb[i][t][cf] = *(const uint4*)(wrow[cf] + (long)t * p.sWt + c * 32);      \
    }                                                                                                                                \
    if (FIRST) {                                                                                                                     \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_bias) : "v"(pb) : "memory");                                          \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_row) : "v"(pr) : "memory");                                           \
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(e_res) : "v"(ps) : "memory");                                           \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_bias2) : "v"(pb2) : "memory");                                        \
    }                                                                                                                                \
    __builtin_amdgcn_sched_barrier(0); /* all loads of the round ahead of its first MFMA */                                          \
    if (GN && FIRST) { /* the sample's group statistics from the producer's slots, then scale / shift per input channel; all of it    \
                          under the operand loads' latency */                                                                        \
      const int G = p.Cin / p.gn_cpg, qpg = p.gn_cpg >> 2, rts = p.L >> 4, S = rts * qpg, nq = p.Cin >> 2;                           \
      const double cnt = (double)p.gn_cpg * (double)p.L;                                                                             \
      for (int g = tid >> 4; g < G; g += 4 * SK_WAVES) { /* 16 threads per group */                                                   \
        const bool pre = g == (tid >> 4);            /* first pass: the three prefetched slots (zeros where there was none) */        \
        double s1 = pre ? (double)pv[0].x + (double)pv[1].x + (double)pv[2].x : 0.0;                                                 \
        double s2 = pre ? (double)pv[0].y + (double)pv[1].y + (double)pv[2].y : 0.0;                                                 \
        for (int i = (tid & 15) + (pre ? 48 : 0); i < S; i += 16) {                                                                  \
          const int rt = i / qpg;                                                                                                    \
          const float2 v = sk_slot(p.gn_part, p.gn_part_b, p.gn_nqa, nq - p.gn_nqa, m0 / p.L, rts, rt, g * qpg + (i - rt * qpg));    \
          s1 += (double)v.x; s2 += (double)v.y;                                                                                      \
        }                                                                                                                            \
        _Pragma("unroll") for (int o = 1; o < 16; o <<= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }                     \
        if ((tid & 15) == 0) {                                                                                                       \
          const double mu = s1 / cnt; double var = s2 / cnt - mu * mu; if (var < 0.0) var = 0.0;                                     \
          gn_mr[g] = make_float2((float)mu, rsqrtf((float)var + p.gn_eps));                                                          \
        }                                                                                                                            \
      }                                                                                                                              \
      __syncthreads();                                                                                                               \
      _Pragma("unroll") for (int j = 0; j < 2; j++) { /* Cin <= SK_GN_MAXC = 2 x 512 */                                              \
        const int ch = tid + j * 64 * SK_WAVES;                                                                                      \
        if (ch < p.Cin) {                                                                                                            \
          const float2 mr = gn_mr[ch / p.gn_cpg];                                                                                    \
          const float ga = g_ga[j] * mr.y;                                                                                           \
          gn_ss[ch] = make_float2(ga, g_be[j] - mr.x * ga);                                                                          \
        }                                                                                                                            \
      }                                                                                                                              \
      __syncthreads();                                                                                                               \
    }                                                                                                                                \
    _Pragma("unroll") for (int i = 0; i < CH; i++) {                                                                                 \
      if (c0 + i * SK_WAVES < kchunks) { /* wave-uniform */                                                                          \
        float nsc[8], nsh[8];                                                                                                        \
        if (GN) {                                                                                                                    \
          const float4* sp = (const float4*)&gn_ss[(c0 + i * SK_WAVES) * 32 + q * 8];                                                \
          _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                                            \
            const float4 v = sp[j]; nsc[2 * j] = v.x; nsh[2 * j] = v.y; nsc[2 * j + 1] = v.z; nsh[2 * j + 1] = v.w;                  \
          }                                                                                                                          \
        }                                                                                                                            \
        uint4 ctr[RF], up[RF], dn[RF], hal = make_uint4(0u, 0u, 0u, 0u);                                                             \
        _Pragma("unroll") for (int rf = 0; rf < RF; rf++) ctr[rf] = GN ? sk_norm8<T16>(xc[i][rf], nsc, nsh, p.gn_silu != 0) : xc[i][rf];  \
        if (TAPS == 3) {                                                                                                             \
          hal = GN ? sk_norm8<T16>(xh[i], nsc, nsh, p.gn_silu != 0) : xh[i];                                                              \
          _Pragma("unroll") for (int rf = 0; rf < RF; rf++) { up[rf] = sk_rot<0x121>(ctr[rf]); dn[rf] = sk_rot<0x12F>(ctr[rf]); }    \
        }                                                                                                                            \
        _Pragma("unroll") for (int t = 0; t < TAPS; t++)                                                                             \
          _Pragma("unroll") for (int rf = 0; rf < RF; rf++) {                                                                        \
            uint4 a = ctr[rf];                                                                                                       \
            if (TAPS == 3 && t == 0) { a = up[rf]; if (lm == 0) a = rf == 0 ? hal : up[rf > 0 ? rf - 1 : 0]; }   /* row above */     \
            if (TAPS == 3 && t == 2) { a = dn[rf]; if (lm == 15) a = rf == RF - 1 ? hal : dn[rf + 1 < RF ? rf + 1 : RF - 1]; } /* below */ \
            if (TAPS == 3 && !xok[t][rf]) a = make_uint4(0u, 0u, 0u, 0u);                                                            \
            _Pragma("unroll") for (int cf = 0; cf < CF; cf++)                                                                        \
              acc[rf][cf] = sk_mma<T16>(wb[i][t][cf], a, acc[rf][cf]);            \
          }                                                                                                                          \
      }                                                                                                                              \
    }                                                                                                                                \
  }
  SK_ROUND(wave, true)
  for (int cr = wave + SK_WAVES * CH; cr < kchunks; cr += SK_WAVES * CH) SK_ROUND(cr, false)
#undef SK_ROUND
  // ---- K extension: the second (one-tap, un-normalised) reduction; wave w takes chunks w, w + 8, ... of Cin2, two per round
  if (p.x2) {
    const int kc2 = p.Cin2 >> 5;
    const bf16_t* x2row[RF]; const bf16_t* w2row[CF];
#pragma unroll
    for (int rf = 0; rf < RF; rf++) x2row[rf] = p.x2 + (long)row[rf] * p.ldx2 + q * 8;
#pragma unroll
    for (int cf = 0; cf < CF; cf++) { const int n = n0 + cf * 16 + lm; w2row[cf] = p.w2 + (long)(n < p.N ? n : p.N - 1) * p.Cin2 + q * 8; }
    for (int c0 = wave; c0 < kc2; c0 += 2 * SK_WAVES) {
      uint4 xa[2][RF], wb2[2][CF];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        int c = c0 + i * SK_WAVES; if (c >= kc2) c = kc2 - 1;
#pragma unroll
        for (int rf = 0; rf < RF; rf++) xa[i][rf] = *(const uint4*)(x2row[rf] + c * 32);
#pragma unroll
        for (int cf = 0; cf < CF; cf++) wb2[i][cf] = *(const uint4*)(w2row[cf] + c * 32);
      }
#pragma unroll
      for (int i = 0; i < 2; i++) {
        if (c0 + i * SK_WAVES < kc2) {
#pragma unroll
          for (int rf = 0; rf < RF; rf++)
#pragma unroll
            for (int cf = 0; cf < CF; cf++)
              acc[rf][cf] = sk_mma<T16>(wb2[i][cf], xa[i][rf], acc[rf][cf]);
        }
      }
    }
  }
#pragma unroll
  for (int rf = 0; rf < RF; rf++)
#pragma unroll
    for (int cf = 0; cf < CF; cf++) part[wave][rf * CF + cf][lane] = acc[rf][cf];
  __syncthreads();
  // ---- fold the eight partial tiles; lane (lm, q) of fragment (rf, cf) owns row m0 + rf*16 + lm, channels n0 + cf*16 + q*4 .. +4
  if (owner) {
    f32x4 s = part[0][f_own][lane];
#pragma unroll
    for (int w = 1; w < SK_WAVES; w++) { const f32x4 v = part[w][f_own][lane]; s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3]; }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_bias), "+v"(e_row), "+v"(e_res), "+v"(e_bias2) :: "memory");
    if (own_ok) {
      if (p.bias) { s[0] += e_bias[0]; s[1] += e_bias[1]; s[2] += e_bias[2]; s[3] += e_bias[3]; }
      if (p.bias2) { s[0] += e_bias2[0]; s[1] += e_bias2[1]; s[2] += e_bias2[2]; s[3] += e_bias2[3]; }
      if (p.rowvec) { s[0] += e_row[0]; s[1] += e_row[1]; s[2] += e_row[2]; s[3] += e_row[3]; }
      if (p.resid) {
        s[0] += w16_lo<T16>(e_res[0]); s[1] += w16_hi<T16>(e_res[0]);
        s[2] += w16_lo<T16>(e_res[1]); s[3] += w16_hi<T16>(e_res[1]);
      }
      *(uint2*)(p.y + (long)m_own * p.ldy + n_own) = make_uint2(pack16x2<T16>(s[0], s[1]), pack16x2<T16>(s[2], s[3]));
    }
    if (p.part_out) {       // GroupNorm statistics of the tensor just written (rounded values): one slot per 16-row x 4-channel lane quad
      const unsigned lo = pack16x2<T16>(s[0], s[1]), hi = pack16x2<T16>(s[2], s[3]);
      const float r0 = w16_lo<T16>(lo), r1 = w16_hi<T16>(lo), r2 = w16_lo<T16>(hi), r3 = w16_hi<T16>(hi);
      float a1 = own_ok ? (r0 + r1) + (r2 + r3) : 0.f, a2 = own_ok ? (r0 * r0 + r1 * r1) + (r2 * r2 + r3 * r3) : 0.f;
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a1 += __shfl_xor(a1, o); a2 += __shfl_xor(a2, o); }
      const int r16 = m0 + rf_own * 16;                                   // first row of this fragment: L % 16 == 0, so inside one sample
      if (lm == 0 && n_own < p.N && r16 < p.M)
        p.part_out[((long)(r16 / p.L) * (p.L >> 4) + ((r16 % p.L) >> 4)) * (p.N >> 2) + (n_own >> 2)] = make_float2(a1, a2);
    }
  }
}

template <int TAPS, bool GN, typename T16>
void sk_launch_t(eegldm_ctx* ctx, const SkArgs& a, int rf, int cf) {
  const dim3 blk(64 * SK_WAVES);
  if (rf == 2 && cf == 2) hipLaunchKernelGGL((conv_skinny_kernel<TAPS, 2, 2, GN, T16>), dim3((a.M + 31) / 32, (a.N + 31) / 32), blk, 0, ctx->stream, a);
  else if (rf == 2) hipLaunchKernelGGL((conv_skinny_kernel<TAPS, 2, 1, GN, T16>), dim3((a.M + 31) / 32, (a.N + 15) / 16), blk, 0, ctx->stream, a);
  else hipLaunchKernelGGL((conv_skinny_kernel<TAPS, 1, 1, GN, T16>), dim3((a.M + 15) / 16, (a.N + 15) / 16), blk, 0, ctx->stream, a);
}
template <int TAPS, bool GN>
void sk_launch(eegldm_ctx* ctx, const SkArgs& a, int rf, int cf, int dtype) {
  if (dtype == EEGLDM_F16) sk_launch_t<TAPS, GN, f16_t>(ctx, a, rf, cf); else sk_launch_t<TAPS, GN, bf16_t>(ctx, a, rf, cf);
}
}  // namespace

static long sk_max_tiles() {
  return 32;
}
// shape test shared by conv_skinny_try and the callers that want to plan a fused GroupNorm around it (net.hip)
bool conv_skinny_takes(int dtype, int Cin, int Cout, int taps, int B, int L) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_NO_CONV_SKINNY") != nullptr);
  const long M = (long)B * L;
  if (off || (dtype != EEGLDM_BF16 && dtype != EEGLDM_F16) || (taps != 1 && taps != 3) || Cin % 32 != 0 || Cout % 4 != 0 || Cout < 16) return false;
  // only launches the general kernel cannot spread over the chip: at most `max_tiles` of its 128 x 128 tiles
  return ((M + 127) / 128) * (((long)Cout + 127) / 128) <= sk_max_tiles();
}

// Y[r][n] = sum_t sum_k A[r + t - pad][k] * w[t][n][k] (+ bias[n] + rowvec[sample(r)][n] + resid[r][n]); rows flattened (sample, position).
// A = X, or with `gn`: A = SiLU?(GroupNorm(X)) rounded to bf16, from gn->stats = (sum, sum of squares) per (sample, group) of X.
// part_out (optional): per-quad statistics of Y for the next GroupNorm ((B * L / 16) * (Cout / 4) float2 slots, all written).
// Returns 1 when this kernel took the launch, 0 when the shape is not its (the caller goes on to the general kernels), < 0 on error.
int conv_skinny_ex(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, int Cin, int Cout, int taps, const float* bias,
                   const float* rowvec, long ld_rowvec, const void* resid, long ldr, void* y, long ldy, int B, int L,
                   const SkinnyGn* gn, float2* part_out, const SkinnyExt* ext) {
  const long M = (long)B * L;
  if (!conv_skinny_takes(dtype, Cin, Cout, taps, B, L)) return 0;
  if (ext && (!ext->x2 || !ext->w2 || ext->Cin2 % 32 != 0 || ext->ldx2 % 8 != 0 || (((size_t)ext->x2 | (size_t)ext->w2) % 16) != 0 ||
              (ext->bias2 && (size_t)ext->bias2 % 16 != 0) || resid)) return 0;
  if (ldx % 8 != 0 || ldy % 4 != 0 || (resid && ldr % 4 != 0) || (rowvec && ld_rowvec % 4 != 0)) return 0;
  if (((size_t)x | (size_t)w) % 16 != 0 || (size_t)y % 8 != 0 || (resid && (size_t)resid % 8 != 0) || (bias && (size_t)bias % 16 != 0) ||
      (rowvec && (size_t)rowvec % 16 != 0)) return 0;
  // the fused forms need whole 32-row tiles inside one sample and 4-channel lane quads inside one group
  if (gn && (L % 32 != 0 || Cin > SK_GN_MAXC || gn->cpg < 4 || gn->cpg % 4 != 0 || Cin % gn->cpg != 0 || Cin / gn->cpg > 64)) return 0;
  if (part_out && L % 32 != 0) return 0;
  SkArgs a = {};
  a.x = (const bf16_t*)x; a.ldx = ldx; a.w = (const bf16_t*)w; a.sWt = (long)Cout * Cin; a.Cin = Cin;
  a.bias = bias; a.rowvec = rowvec; a.ld_rowvec = ld_rowvec; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  a.y = (bf16_t*)y; a.ldy = ldy; a.M = (int)M; a.L = L; a.N = Cout;
  a.part_out = part_out;
  if (ext) { a.x2 = (const bf16_t*)ext->x2; a.ldx2 = ext->ldx2; a.w2 = (const bf16_t*)ext->w2; a.Cin2 = ext->Cin2; a.bias2 = ext->bias2; }
  if (gn) { a.gn_part = gn->part; a.gn_part_b = gn->part_b ? gn->part_b : gn->part; a.gn_nqa = gn->part_b ? gn->nqa : Cin / 4; a.gn_gamma = gn->gamma; a.gn_beta = gn->beta; a.gn_cpg = gn->cpg; a.gn_eps = gn->eps; a.gn_silu = gn->silu; }
  // widest register tile that still gives every CU most of a block (fewer re-reads of the operands through L2)
  EEG_ENV_VAR(int, force, getenv("EEGLDM_CONV_SKINNY_TILE") ? atoi(getenv("EEGLDM_CONV_SKINNY_TILE")) : 0);   // 11 / 21 / 22
  const long want = ctx->num_cu * 3 / 4;
  int rf = 1, cf = 1;
  if (((M + 31) / 32) * ((Cout + 31) / 32) >= want) { rf = 2; cf = 2; }
  else if (((M + 31) / 32) * ((Cout + 15) / 16) >= want) { rf = 2; cf = 1; }
  if (force) { rf = force / 10; cf = force % 10; if (rf < 1 || rf > 2 || cf < 1 || cf > rf) { rf = 1; cf = 1; } }
  ProfRec rec; const bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = taps == 3 ? PROF_CONV_FWD : PROF_GEMM_NT; rec.flops = 2.0 * (double)M * Cout * ((double)Cin * taps + (ext ? ext->Cin2 : 0));
    rec.M = (int)M; rec.N = Cout; rec.K = Cin; rec.taps = taps; rec.splitk = 1;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b));
    HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  if (gn && taps == 3) sk_launch<3, true>(ctx, a, rf, cf, dtype); else if (gn) sk_launch<1, true>(ctx, a, rf, cf, dtype);
  else if (taps == 3) sk_launch<3, false>(ctx, a, rf, cf, dtype); else sk_launch<1, false>(ctx, a, rf, cf, dtype);
  LAUNCH_CHECK();
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return 1;
}

int conv_skinny_try(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, int Cin, int Cout, int taps, const float* bias,
                    const float* rowvec, long ld_rowvec, const void* resid, long ldr, void* y, long ldy, int B, int L) {
  return conv_skinny_ex(ctx, dtype, x, ldx, w, Cin, Cout, taps, bias, rowvec, ld_rowvec, resid, ldr, y, ldy, B, L, nullptr, nullptr, nullptr);
}
