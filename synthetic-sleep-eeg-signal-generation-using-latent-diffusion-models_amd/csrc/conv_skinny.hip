// Few-row convolution / linear layer (bf16, 1 or 3 taps, stride 1): the forward GEMMs of the UNet when a launch has only a few
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
};

template <int TAPS, int RF, int CF>
__global__ __launch_bounds__(64 * SK_WAVES) void conv_skinny_kernel(const SkArgs p) {
  __shared__ f32x4 part[SK_WAVES][RF * CF][64];
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
  f32x4 e_bias, e_row; sk_u32x2 e_res;
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

  f32x4 acc[RF][CF];
#pragma unroll
  for (int rf = 0; rf < RF; rf++)
#pragma unroll
    for (int cf = 0; cf < CF; cf++) acc[rf][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // wave w reduces the 32-channel chunks w, w + 8, ... (all taps of a chunk); CH chunks = CH * TAPS * (RF + CF) loads per round
  constexpr int CH = SK_LOADS / (TAPS * (RF + CF)) < 1 ? 1 : SK_LOADS / (TAPS * (RF + CF));
  // Rounds: the first one is straight-line code (a loop header costs a conservative vmcnt(0) before the loads), the rest -- only
  // reductions longer than 8 * CH chunks have any -- is a loop.
#define SK_ROUND(C0, FIRST)                                                                                                          \
  {                                                                                                                                  \
    const int c0 = (C0);                                                                                                             \
    uint4 xa[CH][TAPS][RF], wb[CH][TAPS][CF];                                                                                        \
    _Pragma("unroll") for (int i = 0; i < CH; i++) {                                                                                 \
      int c = c0 + i * SK_WAVES; if (c >= kchunks) c = kchunks - 1; if (c < 0) c = 0; /* clamped: surplus products are skipped */    \
      _Pragma("unroll") for (int t = 0; t < TAPS; t++) {                                                                             \
        _Pragma("unroll") for (int rf = 0; rf < RF; rf++) xa[i][t][rf] = *(const uint4*)(xrow[t][rf] + c * 32);                      \
        _Pragma("unroll") for (int cf = 0; cf < CF; cf++) wb[i][t][cf] = *(const uint4*)(wrow[cf] + (long)t * p.sWt + c * 32);      \
      }                                                                                                                              \
    }                                                                                                                                \
    if (FIRST) {                                                                                                                     \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_bias) : "v"(pb) : "memory");                                          \
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(e_row) : "v"(pr) : "memory");                                           \
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(e_res) : "v"(ps) : "memory");                                           \
    }                                                                                                                                \
    __builtin_amdgcn_sched_barrier(0); /* all loads of the round ahead of its first MFMA */                                          \
    _Pragma("unroll") for (int i = 0; i < CH; i++) {                                                                                 \
      if (c0 + i * SK_WAVES < kchunks) { /* wave-uniform */                                                                          \
        _Pragma("unroll") for (int t = 0; t < TAPS; t++)                                                                             \
          _Pragma("unroll") for (int rf = 0; rf < RF; rf++) {                                                                        \
            uint4 a = xa[i][t][rf];                                                                                                  \
            if (TAPS == 3 && !xok[t][rf]) a = make_uint4(0u, 0u, 0u, 0u);                                                            \
            _Pragma("unroll") for (int cf = 0; cf < CF; cf++)                                                                        \
              acc[rf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[i][t][cf]),                        \
                                                                    __builtin_bit_cast(bf16x8, a), acc[rf][cf], 0, 0, 0);            \
          }                                                                                                                          \
      }                                                                                                                              \
    }                                                                                                                                \
  }
  SK_ROUND(wave, true)
  for (int cr = wave + SK_WAVES * CH; cr < kchunks; cr += SK_WAVES * CH) SK_ROUND(cr, false)
#undef SK_ROUND
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
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_bias), "+v"(e_row), "+v"(e_res) :: "memory");
    if (own_ok) {
      if (p.bias) { s[0] += e_bias[0]; s[1] += e_bias[1]; s[2] += e_bias[2]; s[3] += e_bias[3]; }
      if (p.rowvec) { s[0] += e_row[0]; s[1] += e_row[1]; s[2] += e_row[2]; s[3] += e_row[3]; }
      if (p.resid) {
        s[0] += __uint_as_float(e_res[0] << 16); s[1] += __uint_as_float(e_res[0] & 0xffff0000u);
        s[2] += __uint_as_float(e_res[1] << 16); s[3] += __uint_as_float(e_res[1] & 0xffff0000u);
      }
      *(uint2*)(p.y + (long)m_own * p.ldy + n_own) = make_uint2(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], s[3]));
    }
  }
}

template <int TAPS>
void sk_launch(eegldm_ctx* ctx, const SkArgs& a, int rf, int cf) {
  const dim3 blk(64 * SK_WAVES);
  if (rf == 2 && cf == 2) hipLaunchKernelGGL((conv_skinny_kernel<TAPS, 2, 2>), dim3((a.M + 31) / 32, (a.N + 31) / 32), blk, 0, ctx->stream, a);
  else if (rf == 2) hipLaunchKernelGGL((conv_skinny_kernel<TAPS, 2, 1>), dim3((a.M + 31) / 32, (a.N + 15) / 16), blk, 0, ctx->stream, a);
  else hipLaunchKernelGGL((conv_skinny_kernel<TAPS, 1, 1>), dim3((a.M + 15) / 16, (a.N + 15) / 16), blk, 0, ctx->stream, a);
}
}  // namespace

// Y[r][n] = sum_t sum_k X[r + t - pad][k] * w[t][n][k] (+ bias[n] + rowvec[sample(r)][n] + resid[r][n]); rows flattened (sample, position).
// Returns 1 when this kernel took the launch, 0 when the shape is not its (the caller goes on to the general kernels), < 0 on error.
int conv_skinny_try(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, int Cin, int Cout, int taps, const float* bias,
                    const float* rowvec, long ld_rowvec, const void* resid, long ldr, void* y, long ldy, int B, int L) {
  static const bool off = getenv("EEGLDM_NO_CONV_SKINNY") != nullptr;
  static const long max_tiles = getenv("EEGLDM_CONV_SKINNY_MAX_TILES") ? atol(getenv("EEGLDM_CONV_SKINNY_MAX_TILES")) : 32;
  const long M = (long)B * L;
  if (off || dtype != EEGLDM_BF16 || (taps != 1 && taps != 3) || Cin % 32 != 0 || Cout % 4 != 0 || Cout < 16) return 0;
  // only launches the general kernel cannot spread over the chip: at most `max_tiles` of its 128 x 128 tiles
  if (((M + 127) / 128) * (((long)Cout + 127) / 128) > max_tiles) return 0;
  if (ldx % 8 != 0 || ldy % 4 != 0 || (resid && ldr % 4 != 0) || (rowvec && ld_rowvec % 4 != 0)) return 0;
  if (((size_t)x | (size_t)w) % 16 != 0 || (size_t)y % 8 != 0 || (resid && (size_t)resid % 8 != 0) || (bias && (size_t)bias % 16 != 0) ||
      (rowvec && (size_t)rowvec % 16 != 0)) return 0;
  SkArgs a = {};
  a.x = (const bf16_t*)x; a.ldx = ldx; a.w = (const bf16_t*)w; a.sWt = (long)Cout * Cin; a.Cin = Cin;
  a.bias = bias; a.rowvec = rowvec; a.ld_rowvec = ld_rowvec; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  a.y = (bf16_t*)y; a.ldy = ldy; a.M = (int)M; a.L = L; a.N = Cout;
  // widest register tile that still gives every CU most of a block (fewer re-reads of the operands through L2)
  static const int force = getenv("EEGLDM_CONV_SKINNY_TILE") ? atoi(getenv("EEGLDM_CONV_SKINNY_TILE")) : 0;   // 11 / 21 / 22
  const long want = ctx->num_cu * 3 / 4;
  int rf = 1, cf = 1;
  if (((M + 31) / 32) * ((Cout + 31) / 32) >= want) { rf = 2; cf = 2; }
  else if (((M + 31) / 32) * ((Cout + 15) / 16) >= want) { rf = 2; cf = 1; }
  if (force) { rf = force / 10; cf = force % 10; if (rf < 1 || rf > 2 || cf < 1 || cf > rf) { rf = 1; cf = 1; } }
  ProfRec rec; const bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = taps == 3 ? PROF_CONV_FWD : PROF_GEMM_NT; rec.flops = 2.0 * (double)M * Cout * Cin * taps;
    rec.M = (int)M; rec.N = Cout; rec.K = Cin; rec.taps = taps; rec.splitk = 1;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b));
    HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  if (taps == 3) sk_launch<3>(ctx, a, rf, cf); else sk_launch<1>(ctx, a, rf, cf);
  LAUNCH_CHECK();
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return 1;
}
