// 192 x 256 output tile, 8 waves, ONE workgroup per CU: the implicit-GEMM 3-tap convolution (forward and data gradient) and the
// 1-tap NT products of the UNet's 256- and 512-channel levels (bf16 storage, fp32 accumulate).
//   Y[r][n] = sum_t sum_k A[r + t - pad][k] * W_t[n][k]   (+ bias[n] + rowvec[sample(r)][n] + resid[r][n])
// (reference ops: nn.Conv1d of /root/reference/src/models/unet.py:263,291,302 -- in_layers / out_layers / skip_connection -- and
//  their input gradients; the data gradient runs as the same product over dY with the taps flipped and a [tap][Cin][Cout] weight copy)
//
// Why a second GEMM kernel (gemm.hip keeps every other shape): the 128 x 128 tile of gemm.hip moves 1 byte from L2 to LDS per 48 MACs
// and its K loop is bound by exactly that path (~29 B/clk/CU, MFMA pipe 62 % busy in the loop, 0.32 of the MFMA peak over
// the step).  This tile moves 1 byte per 77 (3 taps) / 110 (1 tap) MACs in 128-byte row segments (the L2 -> LDS path runs 64-byte
// segments at half the rate, tools/probes/fill_pattern_probe.hip), and 192 rows divide every sequence length of the model (192 / 384 /
// 768; 768 / 1536 / 3072), so a tile never straddles two samples: the conv's zero padding is two halo rows per tile, decided per tile,
// instead of a per-lane row mask.  M tiles x N tiles = 512 workgroups for every 256- / 512-channel layer at B = 256: two full rounds of
// the 256 CUs (a 256 x 256 tile would give 384 = 1.5 rounds).
//
// Structure (one K stage = 64 reduction channels, 128 bytes per tile row):
//   LDS      2 x A tile  (192 rows + 2 halo rows + DMA padding, 26 KB each)  +  3 x B piece (256 rows of ONE tap, 32 KB each)  = 148 KB
//   pieces   p = 3 * stage + tap;  piece p = B(stage, tap)  [+ A(stage) when tap == 0]  filled by LDS-DMA (global_load_lds_dwordx4),
//            lane-linear 1 KB per wave instruction, XOR-swizzled on the SOURCE side (gemm.hip nt_swz<2>: conflict-free ds_read_b128)
//   phase p  48 MFMAs per wave (6 x 4 fragments x 2 k-steps of 32) with the fragment reads running ONE k-step ahead, across the phase
//            boundary; ONE barrier per phase, between the two k-steps, in front of which a wave waits (counted vmcnt) for ITS part of
//            piece p+1 and behind which the DMAs of piece p+3 are issued: every piece has two full phases to land (see the kernel).
//   epilogue bias / embedding row / residual added in the fragment layout (fp32), packed to bf16, transposed through LDS, 16-byte
//            row-contiguous stores.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "internal.h"

namespace {

// developer ablation builds (tools/debug/gemm_big_ablate.sh): -DEEG_BIG_DBG=<bits>, compile-time so that the loop keeps its shape
#ifdef EEG_BIG_DBG
#define BIG_DBG(bit) ((EEG_BIG_DBG) & (bit))
#else
#define BIG_DBG(bit) 0
#endif

constexpr int BM = 192, BN = 256, BK = 64, NWAVE = 8, NTHR = 512;
constexpr int FM = 6, FN = 4;                       // 16 x 16 fragments per wave: 96 rows x 64 columns (waves: 2 along M x 4 along N)
constexpr int ROWB = BK * 2;                        // bytes per tile row
constexpr int A_MAIN = BM * ROWB;                   // 24 KB = 24 wave instructions (3 per wave)
// A tile in LDS: [896 B pad][halo row above the tile][192 rows][halo row below][896 B pad] -- the 194 rows are LINEAR (tap t of tile row
// r is physical row r + t, so a fragment address is one per-lane base + immediates), and each halo row is the tail / head of a 1 KB
// DMA instruction whose other lanes land in the padding (wave 0: the row above, wave 1: the row below)
constexpr int A_PAD = 896;
constexpr int A_ALLOC = A_PAD + (BM + 2) * ROWB + A_PAD;      // 26 624
constexpr int B_ALLOC = BN * ROWB;                  // 32 KB = 32 wave instructions (4 per wave)
constexpr int RING = 2 * A_ALLOC + 3 * B_ALLOC;     // 151 552 bytes
constexpr int PITCH16 = BN * 2 + 16;                // packed bf16 epilogue tile
constexpr int EPI_BYTES = BM * PITCH16;
constexpr int LDS_BYTES = RING > EPI_BYTES ? RING : EPI_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the 160 KiB LDS");

struct BigArgs {
  const bf16_t* A; long lda;
  const bf16_t* B; long ldb; long sBt;     // plain: [tap][N][K] (ldb = K stride of a row); K-blocked: [tap][K/32][N][32]
  bf16_t* C; long ldc;
  int M, N, K, L;                         // L = rows per sample (conv zero padding at its edges); M % 192 == 0, L % 192 == 0, N % 256 == 0, K % 64 == 0
  const float* bias; const float* rowvec; long ld_rowvec; int rows_per_vec;
  const bf16_t* resid; long ldr;
  const void* zero_page;
  int tiles_m, tiles_n;
  // K extension (gemm_bigp_kernel<.., XT = true>): K2 / 64 further 1-tap stages over a SECOND operand pair behind the 3-tap stages --
  //   Y[r][n] += sum_k A2[r][k] * W2[n][k]  (+ bias2[n])
  // = the ResBlock's skip_connection 1 x 1 conv folded into out_layers' conv (h = skip(x) + conv2(a2), unet.py:302,327): one launch, one
  // fp32 accumulator, one rounding, no intermediate tensor.  W2 is K-blocked [K2 / 32][N][32]; K2 % 64 == 0, K2 >= 128.
  const bf16_t* A2; long lda2; const bf16_t* B2; int K2; const float* bias2;
  // operand transform (gemm_big_kernel<.., XF != 0>, round 6): the A tile of every K stage is rewritten IN LDS, once it has landed, as
  //   a[r][k] <- act(a[r][k] * xf_scale[sample(r)][k] + xf_shift[sample(r)][k])          (act: XF 1 = SiLU, 2 = LeakyReLU(xf_slope))
  // = GroupNorm(+SiLU) / BatchNorm + LeakyReLU applied on the consuming conv's operand load: the normalised tensor is never written.
  // Tables are fp32 [M / L][xf_ld] (xf_ld = 0: one row for every sample); zero-padding rows of the conv stay zero.
  const float* xf_scale = nullptr; const float* xf_shift = nullptr; long xf_ld = 0; float xf_slope = 0.f;
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;

__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {       // see gemm.hip: inline asm so that hipcc does not order ds_reads behind it
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(__builtin_amdgcn_readfirstlane(lds_off)), "v"(g) : "memory", "m0");
}
// the same with a scalar base and a 32-bit per-lane byte offset
__device__ __forceinline__ void dma16s(const void* base, unsigned voff, unsigned lds_off) {
  // (readfirstlane: the "s" constraint needs a value the compiler KNOWS to be wave-uniform; the tile origin comes from blockIdx arithmetic it does not always prove)
  const unsigned long long u = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  const unsigned long long b = ((unsigned long long)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(__builtin_amdgcn_readfirstlane(lds_off)), "v"(voff), "s"(b) : "memory", "m0");
}
// one 16 x 16 x 32 MFMA step in the operand format of storage type T (bf16_t: v_mfma_f32_16x16x32_bf16; f16_t, round 5: ..._f16)
typedef _Float16 big_f16x8 __attribute__((ext_vector_type(8)));
template <typename T> __device__ __forceinline__ f32x4 big_mma(const uint4& a, const uint4& b, const f32x4& c) {
  if constexpr (Is16<T>::f16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(big_f16x8, a), __builtin_bit_cast(big_f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int swz2(int row) { return 2 * ((row >> 1) & 3); }     // 16-byte slot ^= swz2(row) (128-byte rows)

// ---- epilogue shared by both kernels: acc[i][j][r] = C[wm*96 + i*16 + lm][wn*64 + j*16 + q*4 + r] (operands were swapped: 4 consecutive
// columns per lane).  bias / embedding row / residual are added in fp32 in the fragment layout, the sum is rounded to bf16 ONCE, the
// packed tile goes through LDS and leaves as 16-byte row-contiguous stores.
template <typename T> __device__ __forceinline__ void big_epilogue(const BigArgs& p, f32x4 (&acc)[FM][FN], char* smem, int m0, int n0, int tid, int lm, int q, int wm, int wn) {

  float4 add[FN];
#pragma unroll
  for (int j = 0; j < FN; j++) {
    const int nj = n0 + wn * 64 + j * 16 + q * 4;
    add[j] = p.bias ? *(const float4*)(p.bias + nj) : make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.rowvec) {           // a tile lies inside one sample (rows_per_vec % 192 == 0)
      const float4 e = *(const float4*)(p.rowvec + (long)(m0 / p.rows_per_vec) * p.ld_rowvec + nj);
      add[j].x += e.x; add[j].y += e.y; add[j].z += e.z; add[j].w += e.w;
    }
  }
  uint2 res[2][FN];
  auto load_res = [&](int i, uint2 (&r)[FN]) __attribute__((always_inline)) {
    const bf16_t* rp = p.resid + (long)(m0 + wm * (FM * 16) + i * 16 + lm) * p.ldr + n0 + wn * 64 + q * 4;
#pragma unroll
    for (int j = 0; j < FN; j++) r[j] = *(const uint2*)(rp + j * 16);
  };
  if (p.resid) load_res(0, res[0]);
#pragma unroll
  for (int i = 0; i < FM; i++) {
    if (p.resid && i + 1 < FM) load_res(i + 1, res[(i + 1) & 1]);
#pragma unroll
    for (int j = 0; j < FN; j++) {
      float4 a = add[j];
      if (p.resid) {
        const uint2 r = res[i & 1][j];
        a.x += w16_lo<T>(r.x); a.y += w16_hi<T>(r.x);
        a.z += w16_lo<T>(r.y); a.w += w16_hi<T>(r.y);
      }
      uint2 o;
      o.x = pack16x2<T>(acc[i][j][0] + a.x, acc[i][j][1] + a.y);
      o.y = pack16x2<T>(acc[i][j][2] + a.z, acc[i][j][3] + a.w);
      *(uint2*)(smem + (wm * (FM * 16) + i * 16 + lm) * PITCH16 + (wn * 64 + j * 16 + q * 4) * 2) = o;
    }
  }
  __syncthreads();
  constexpr int CH8 = BN / 8, RSTEP = NTHR / CH8, NIT = BM / RSTEP;      // 32 chunks per row, 16 rows per pass, 12 passes
  const int cs8 = tid % CH8, r0 = tid / CH8;
  uint4 v8[NIT];
#pragma unroll
  for (int cc = 0; cc < NIT; cc++) v8[cc] = *(const uint4*)(smem + (r0 + cc * RSTEP) * PITCH16 + cs8 * 16);
#pragma unroll
  for (int cc = 0; cc < NIT; cc++) *(uint4*)(p.C + (long)(m0 + r0 + cc * RSTEP) * p.ldc + n0 + cs8 * 8) = v8[cc];
}

// XF (round 6, measured prototype of normalise-on-operand-load): 0 = none; 1 / 2 = the A tile of stage s + 1 is transformed in
// LDS (scale, shift, SiLU / LeakyReLU) between the barriers of phases (s, 1) and (s, 2).  That needs the tile landed one barrier earlier
// than the plain schedule asks (vmcnt(0) at the barrier of phase (s, 1): one phase of DMA lead for the A-carrying piece instead of two).
template <int TAPS, bool KBLK, bool FLIP, typename T = bf16_t, int XF = 0>      // FLIP: tap t reads weight slice 2 - t (data gradient; also tells the two apart in a kernel trace).  TAPS = 3 only: with one tap every piece carries an A tile and the two-buffer A ring would be overwritten while it is read
__global__ __launch_bounds__(NTHR, 2) void gemm_big_kernel(const BigArgs p) {
  static_assert(TAPS == 3, "ring layout assumes three pieces per A tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  // XCD-aware tile order (workgroups are dealt round-robin to the 8 XCDs): XCD x walks M tiles x, x + 8, ... and runs all N tiles of
  // an M tile back to back, so an A row panel is fetched into ONE L2
  int tile_m, tile_n;
  {
    const int w = blockIdx.x;
    if ((p.tiles_m & 7) == 0) { const int xcd = w & 7, slot = w >> 3; tile_n = slot % p.tiles_n; tile_m = (slot / p.tiles_n) * 8 + xcd; }
    else { tile_n = w % p.tiles_n; tile_m = w / p.tiles_n; }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int S = p.K / BK;
  const bf16_t* zeros = (const bf16_t*)p.zero_page;

  // ---- XF: scale / shift rows of this tile's sample into LDS behind the ring (2 K floats), before any DMA is in flight
  float* const xs = (float*)(smem + RING);
  bool xf_top = false, xf_bot = false;
  if constexpr (XF != 0) {
    const long smp = m0 / p.L;
    const float* gsc = p.xf_scale + smp * p.xf_ld; const float* gsh = p.xf_shift + smp * p.xf_ld;
    for (int i = tid; i < p.K; i += NTHR) { xs[i] = gsc[i]; xs[p.K + i] = gsh[i]; }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    xf_top = m0 % p.L != 0; xf_bot = (m0 + BM) % p.L != 0;      // halo rows that are real rows of the sample (else: the conv's zero padding, left alone)
  }
  // chunk k of this thread (16 bytes = 8 channels of one tile row) of A buffer `buf`, K stage s
  // developer ablations of the transform (tools/r06/gn_onload_ablate.sh): -DEEG_BIG_XF_ABL=1 the LDS round trip without the arithmetic,
  // =2 no transform at all (what the earlier "tile landed" wait costs by itself); -DEEG_BIG_DBG=4 removes the MFMAs as for the plain kernel
#ifndef EEG_BIG_XF_ABL
#define EEG_BIG_XF_ABL 0
#endif
  auto xform = [&](const int buf, const int s, const int k) __attribute__((always_inline)) {
    if constexpr (XF != 0 && EEG_BIG_XF_ABL != 2) {
      const int c = tid + NTHR * k;
      if (c < (BM + 2) * 8) {
        const int pr = c >> 3;
        if (!((pr == 0 && !xf_top) || (pr == BM + 1 && !xf_bot))) {
          char* ptr = smem + buf * A_ALLOC + A_PAD + c * 16;
          const uint4 v = *(const uint4*)ptr;
#if EEG_BIG_XF_ABL == 1
          *(uint4*)ptr = v;      // ablation: the LDS round trip of the transform without its arithmetic
#else
          const int ch = s * BK + (((c & 7) ^ swz2(pr)) << 3);
          const float4 s0 = *(const float4*)(xs + ch), s1 = *(const float4*)(xs + ch + 4);
          const float4 h0 = *(const float4*)(xs + p.K + ch), h1 = *(const float4*)(xs + p.K + ch + 4);
          const float x[8] = {w16_lo<T>(v.x), w16_hi<T>(v.x), w16_lo<T>(v.y), w16_hi<T>(v.y), w16_lo<T>(v.z), w16_hi<T>(v.z), w16_lo<T>(v.w), w16_hi<T>(v.w)};
          const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
          float y[8];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const float z = fmaf(x[e], scv[e], shv[e]);
            y[e] = XF == 1 ? silu_f(z) : (z > 0.f ? z : p.xf_slope * z);
          }
          uint4 o; o.x = pack16x2<T>(y[0], y[1]); o.y = pack16x2<T>(y[2], y[3]); o.z = pack16x2<T>(y[4], y[5]); o.w = pack16x2<T>(y[6], y[7]);
          *(uint4*)ptr = o;
#endif
        }
      }
    }
  };

  // ---- LDS-DMA sources, decoded once: LDS chunk c of a tile receives the 16 bytes its swizzled position stands for.
  // 32-bit per-lane byte offsets against a SCALAR base that the K loop advances (global_load_lds with an SGPR base): one VGPR per
  // DMA instruction instead of a 64-bit pointer, no per-lane address arithmetic in the loop.
  unsigned ao_src[3], bo_src[4], ah_src = 0;
  bool ah_ok = false;
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int c = (wave + NWAVE * j) * 64 + lane, pr = 1 + (c >> 3), lc = (c & 7) ^ swz2(pr);      // physical row 1 + tile row
    ao_src[j] = (unsigned)(((long)(pr - 1) * p.lda + lc * 8) * 2);
  }
  // halo rows (rows of another sample = the conv's zero padding): wave 0, lanes 56-63 -> physical row 0 (the row above the tile);
  // wave 1, lanes 0-7 -> physical row 193 (the row below); the other lanes of both instructions fetch zeros into the padding
  if (wave == 0 && lane >= 56) { ah_ok = m0 % p.L != 0; ah_src = (unsigned)((((lane & 7) ^ swz2(0)) * 8) * 2); }
  if (wave == 1 && lane < 8) { ah_ok = (m0 + BM) % p.L != 0; ah_src = (unsigned)((((long)(BM + 1) * p.lda) + ((lane & 7) ^ swz2(BM + 1)) * 8) * 2); }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = (wave + NWAVE * j) * 64 + lane, nr = c >> 3, lc = (c & 7) ^ swz2(nr);
    bo_src[j] = KBLK ? (unsigned)((((long)(lc >> 2) * p.N + nr) * 32 + (lc & 3) * 8) * 2) : (unsigned)(((long)nr * p.ldb + lc * 8) * 2);
  }
  const long bstep = KBLK ? (long)2 * p.N * 32 : (long)BK;      // elements per K stage
  const bf16_t* const a_base = p.A + (long)m0 * p.lda;
  const bf16_t* const b_base = p.B + (KBLK ? (long)n0 * 32 : (long)n0 * p.ldb);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  constexpr int B0 = 2 * A_ALLOC;          // B pieces behind the two A tiles
  // DMA instruction k of piece (s, t): k < 4 the B piece; 4..6 the A tile of stage s and 7 its halo rows (pieces with t == 0 only)
  auto issue = [&](int s, int t, int k) __attribute__((always_inline)) {
    if (k < 4) {
      const int tw = FLIP ? 2 - t : t;
      dma16s(b_base + s * bstep + tw * p.sBt, bo_src[k], lds0 + B0 + t * B_ALLOC + (wave_u + NWAVE * k) * 1024);
    } else if (k < 7) {
      dma16s(a_base + (long)s * BK, ao_src[k - 4], lds0 + (s & 1) * A_ALLOC + A_PAD + ROWB + (wave_u + NWAVE * (k - 4)) * 1024);
    } else {
      if (wave_u < 2) {
        const bf16_t* hb = a_base - p.lda + (long)s * BK;         // base of the row above the tile
        const void* src = ah_ok ? (const void*)((const char*)hb + ah_src) : (const void*)zeros;
        dma16(src, lds0 + (s & 1) * A_ALLOC + (wave_u == 0 ? 0 : A_PAD + (BM + 1) * ROWB));
      }
    }
  };
  auto issue_all = [&](int pc) __attribute__((always_inline)) {
    const int s = pc / TAPS, t = pc % TAPS;
#pragma unroll
    for (int k = 0; k < 8; k++) if (k < 4 || t == 0) issue(s, t, k);
  };

  // ---- fragment addresses: one per-lane base per tap (A: physical row wm*96 + lm + t) and one for B; fragment i / j adds the
  // immediate 16 rows x 128 B = 2048 (the swizzle only looks at row bits 1-2), k-step 1 flips byte-offset bit 6
  unsigned aof[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; t++) { const int pr = wm * (FM * 16) + lm + t; aof[t] = A_PAD + pr * ROWB + ((q ^ swz2(pr)) << 4); }
  const unsigned bof = (wn * 64 + lm) * ROWB + ((q ^ swz2(lm)) << 4);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int j = 0; j < FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- schedule.  ONE barrier per phase, placed between its two k-steps:
  //   k-step 0 of phase p    24 MFMAs on the fragments read during phase p-1; the k-step-1 fragments of piece p stream in behind them
  //   wait + barrier B_p     each wave: its DMAs of piece p+1 have landed (counted vmcnt: piece p+2 may stay in flight) and its LDS
  //                          reads of piece p have returned; the barrier makes both true for the whole workgroup
  //   k-step 1 of phase p    issue the DMAs of piece p+3 into the buffer piece p just vacated; 24 MFMAs; the k-step-0 fragments of
  //                          piece p+1 stream in behind them
  // A piece is therefore issued TWO full phases before the barrier that needs it (B_{p+2} for piece p+3) with three B buffers --
  // putting the barrier at the top of the phase (round-4 first version) left it one phase, and the top-of-phase vmcnt(0) then waited on
  // DMAs issued ~0.6 us earlier: 2600 cycles per phase against 1536 of MFMA work.
  issue_all(0);
  issue_all(1);
  issue_all(2);

  // Fragment registers: ONE set of A fragments (af[i] is re-loaded for the next k-step as soon as row i's MFMAs of this k-step are
  // issued) and two sets of B fragments (all four are needed until the last row) -- 56 registers instead of 80 for full double buffering
  uint4 af[FM], bf0[FN], bf1[FN];
  // counted wait: at most `n` of this wave's DMA instructions still in flight.  Pieces with an A tile carry 7 instructions, 8 in
  // waves 0 and 1 (the halo rows); the others 4.
  auto wait_dma = [&](const int t_inflight, const bool any) __attribute__((always_inline)) {
    if (!any) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (t_inflight != 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (wave_u < 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  };
  // prologue: piece 0 landed (pieces 1 and 2 may still fly), first fragments
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // pieces 1 + 2 = 4 + 4 instructions may stay in flight
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if constexpr (XF != 0) {      // the first A tile: transformed here, behind one more barrier
#pragma unroll
    for (int k = 0; k < 4; k++) xform(0, 0, k);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(smem + B0 + bof + j * 2048);
#pragma unroll
  for (int i = 0; i < FM; i++) af[i] = *(const uint4*)(smem + aof[0] + i * 2048);
  // one phase; t (tap), has1 / has2 / has3 (piece p + 1 / 2 / 3 exists) are compile-time constants after unrolling, s is wave-uniform
  auto phase = [&](const int s, const int t, const bool has1, const bool has2, const bool has3) __attribute__((always_inline)) {
    const char* smA = smem + (s & 1) * A_ALLOC;
    const char* smB = smem + B0 + t * B_ALLOC;
    const int t1 = (t + 1) % TAPS, s1 = s + (t + 1) / TAPS;                       // piece p + 1
    const int t2 = (t + 2) % TAPS;                                                 // piece p + 2 (in flight across the barrier)
    // ---- k-step 0 (fragments already in registers); the k-step-1 fragments of this piece are read behind the MFMAs
#pragma unroll
    for (int i = 0; i < FM; i++) {
#pragma unroll
      for (int j = 0; j < FN; j++)
        if (!BIG_DBG(4)) acc[i][j] = big_mma<T>(bf0[j], af[i], acc[i][j]);
      if (i == 0) {
        __builtin_amdgcn_sched_barrier(0);     // (keeps the reads below behind row 0's MFMAs: the compiler's wait in front of the first MFMA after an LDS read is lgkmcnt(0))
#pragma unroll
        for (int j = 0; j < FN; j++) if (!BIG_DBG(2)) bf1[j] = *(const uint4*)(smB + (bof ^ 64) + j * 2048);
      }
      if (!BIG_DBG(2)) af[i] = *(const uint4*)(smA + (aof[t] ^ 64) + i * 2048);
      __builtin_amdgcn_sched_barrier(0);       // pin the order: hipcc otherwise sinks every fragment read to just before its first use
      if constexpr (XF != 0) {                 // second half of the transform of A(s + 1) (landed since the barrier of phase (s, 1))
        if (t == 2 && has1) { if (i == 1) xform((s + 1) & 1, s + 1, 2); if (i == 3) xform((s + 1) & 1, s + 1, 3); __builtin_amdgcn_sched_barrier(0); }
      }
    }
    // ---- B_p: piece p+1 landed everywhere, piece p read by everyone
    if (XF != 0 && t == 1 && has2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // XF: piece p+2 = the next A tile has to be in LDS now: it is transformed behind this barrier
    else if (has1) wait_dma(t2, has2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- k-step 1; piece p+3 = (s + 1, t) goes into the buffers piece p vacates; the first k-step of piece p+1 is read behind the MFMAs
    const char* smA1 = smem + (s1 & 1) * A_ALLOC;
    const char* smB1 = smem + B0 + t1 * B_ALLOC;
    if (has1) {
#pragma unroll
      for (int j = 0; j < FN; j++) if (!BIG_DBG(2)) bf0[j] = *(const uint4*)(smB1 + bof + j * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FM; i++) {
#pragma unroll
      for (int j = 0; j < FN; j++) {
        if (!BIG_DBG(4)) acc[i][j] = big_mma<T>(bf1[j], af[i], acc[i][j]);
        if (has3 && i <= 3 && !BIG_DBG(1)) {                  // DMA instructions 2i, 2i + 1 of piece p+3 go out behind MFMAs 1 and 3 of rows 0..3
          const int k = 2 * i + (j >> 1);
          if ((j & 1) && (k < 4 || t == 0)) issue(s + 1, t, k);
        }
      }
      if (has1 && !BIG_DBG(2)) af[i] = *(const uint4*)(smA1 + aof[t1] + i * 2048);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (XF != 0) {                 // first half of the transform of A(s + 1)
        if (t == 1 && has2) { if (i == 1) xform((s + 1) & 1, s + 1, 0); if (i == 3) xform((s + 1) & 1, s + 1, 1); __builtin_amdgcn_sched_barrier(0); }
      }
    }
  };
#pragma unroll 1
  for (int s = 0; s + 1 < S; s++) {
    phase(s, 0, true, true, true); phase(s, 1, true, true, true); phase(s, 2, true, true, true);
  }
  phase(S - 1, 0, true, true, false); phase(S - 1, 1, true, false, false); phase(S - 1, 2, false, false, false);
  __syncthreads();       // all waves done with the ring before the epilogue tile reuses it
  if (BIG_DBG(8)) { if (acc[0][0][0] == 123.456f) p.C[0] = 0; return; }
  big_epilogue<T>(p, acc, smem, m0, n0, tid, lm, q, wm, wn);
}

// ---- persistent variant's epilogue: no LDS.  The fragment layout gives a lane 4 consecutive columns (8 bytes as bf16) of row lm;
// v_permlane16_swap (gfx950) trades the packed halves of two neighbouring fragments between lane rows q and q ^ 1, after which a lane
// holds 8 consecutive columns (16 bytes) and a wave instruction writes 64 contiguous bytes of each of 16 rows.  EXACTLY NST store
// instructions per wave: the next tile's counted vmcnt waits step over them.
constexpr int NST = FM * (FN / 2);
template <typename T> __device__ __forceinline__ void big_store(const BigArgs& p, f32x4 (&acc)[FM][FN], int m0, int n0, int lm, int q, int wm, int wn) {
  bf16_t* cp = p.C + (long)(m0 + wm * (FM * 16) + lm) * p.ldc + n0 + wn * 64 + (q & 1) * 16 + (q >> 1) * 8;
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int jp = 0; jp < FN / 2; jp++) {
      const unsigned x0 = pack16x2<T>(acc[i][2 * jp][0], acc[i][2 * jp][1]), x1 = pack16x2<T>(acc[i][2 * jp][2], acc[i][2 * jp][3]);
      const unsigned y0 = pack16x2<T>(acc[i][2 * jp + 1][0], acc[i][2 * jp + 1][1]), y1 = pack16x2<T>(acc[i][2 * jp + 1][2], acc[i][2 * jp + 1][3]);
      // odd lane rows of x <-> even lane rows of y: q even keeps its fragment 2jp columns and receives the next four from q + 1;
      // q odd receives fragment 2jp + 1's previous four from q - 1 and keeps its own
      const auto s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
      uint4 o; o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
      if (!BIG_DBG(16) || o.x == 0x12345u) *(uint4*)(cp + (long)i * 16 * p.ldc + jp * 32) = o;      // (bit 16: ablation build without the output stores)
    }
}

// ---- persistent form of the 3-tap kernel: one workgroup per CU walks tiles blockIdx.x, + gridDim.x, ...
// What it is for: in the one-tile-per-workgroup form every CU reaches its epilogue at the same time, the 25 MB of a round drain at
// HBM rate while nothing computes (4.3 us per round measured: K sweep in tools/debug/gemm_big_check.py), and the next round's first
// pieces are requested only after that.  Here the next tile's first three pieces are requested BEFORE this tile's output leaves.
// gfx950 retires loads and stores through ONE in-order vmcnt, so (a) the first counted waits of the next tile step over the NST
// stores (`carry`), and (b) a wave that waits for a DMA issued behind its stores waits for the stores too -- therefore only waves 0-3
// (one per SIMD) issue DMAs and wait for them; waves 4-7 never wait on vmcnt in the loop, their half of the output drains under the
// next tile's K loop, and the loaders' half has until the third barrier of the next tile.
// Loader addressing: DMA instruction k of loader wave w fills LDS chunks ((w + 4k) * 64 + lane) = 8 tile rows; k advances the source
// by 32 rows -- a SCALAR step on the base -- so the per-lane part (row within the 8, swizzled 16-byte slot) is ONE VGPR per operand.
// K extension (XT): the S2 = K2 / 64 extra pieces X_e = (A2 tile of stage e, W2 piece of stage e) follow the 3 S main pieces.  Every X
// piece carries an A tile, and the ring has two A buffers, so X_e can only be requested once X_{e-2} (or, for X_1, the last main A tile)
// has been read: X pieces run TWO phases ahead instead of three (X_0 is requested behind the barrier of phase (S-1, 1), X_1 behind that
// of (S-1, 2), X_{e+2} behind that of X_e) and every barrier of that region waits for vmcnt(0) -- the schedule of gemm_big1p_kernel.
// A2 tiles land where the main tiles land (physical row = tile row + 1) and are read with the centre tap's fragment addresses; B pieces
// keep walking the three B buffers (X_e in buffer e % 3 = where piece 3 S + e would go).
template <bool KBLK, bool FLIP, int NLOAD, bool XT = false, typename T = bf16_t>      // NLOAD loader waves (4 or 2)
__global__ __launch_bounds__(NTHR, 2) void gemm_bigp_kernel(const BigArgs p) {
  constexpr int TAPS = 3;
  static_assert(!XT || (KBLK && !FLIP), "the K extension is a forward product on K-blocked weights");
  static_assert(NLOAD == 2 || NLOAD == 4, "halo rows belong to waves 0 and 1; vmcnt holds 63");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = p.tiles_m * p.tiles_n;
  // XCD-aware tile order as in gemm_big_kernel (gridDim.x is a multiple of 8, so a workgroup's tiles stay on its XCD)
  auto decode = [&](int w, int& tm0, int& tn0) __attribute__((always_inline)) {
    int tile_m, tile_n;
    if ((p.tiles_m & 7) == 0) { const int xcd = w & 7, slot = w >> 3; tile_n = slot % p.tiles_n; tile_m = (slot / p.tiles_n) * 8 + xcd; }
    else { tile_n = w % p.tiles_n; tile_m = w / p.tiles_n; }
    tm0 = tile_m * BM; tn0 = tile_n * BN;
  };
  const int S = p.K / BK;
  const int S2 = XT ? p.K2 / BK : 0;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  constexpr int B0 = 2 * A_ALLOC;          // B pieces behind the two A tiles

  // ---- fragment addresses (as in gemm_big_kernel)
  unsigned aof[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; t++) { const int pr = wm * (FM * 16) + lm + t; aof[t] = A_PAD + pr * ROWB + ((q ^ swz2(pr)) << 4); }
  const unsigned bof = (wn * 64 + lm) * ROWB + ((q ^ swz2(lm)) << 4);

  auto run = [&](auto loader_tag) __attribute__((always_inline)) {
    constexpr bool LOADER = decltype(loader_tag)::value;
    // ---- loader state
    const int r8 = lane >> 3;
    const int a_lc = (lane & 7) ^ swz2(1 + r8), b_lc = (lane & 7) ^ swz2(r8);           // (physical A row = 1 + tile row)
    const unsigned a_voff = (unsigned)(((long)r8 * p.lda + a_lc * 8) * 2);
    const unsigned b_voff = KBLK ? (unsigned)((((long)(b_lc >> 2) * p.N + r8) * 32 + (b_lc & 3) * 8) * 2) : (unsigned)(((long)r8 * p.ldb + b_lc * 8) * 2);
    const long a_kstep = 8 * NLOAD * p.lda, b_kstep = KBLK ? (long)8 * NLOAD * 32 : 8 * NLOAD * p.ldb;       // elements per DMA instruction step (8 NLOAD rows)
    const long bstep = KBLK ? (long)2 * p.N * 32 : (long)BK;                             // elements per K stage
    // halo rows (rows of another sample = the conv's zero padding): wave 0, lanes 56-63 -> physical row 0 (the row above the tile);
    // wave 1, lanes 0-7 -> physical row 193 (the row below); the other lanes of both instructions fetch zeros into the padding
    const bool ah_top = wave == 0 && lane >= 56, ah_bot = wave == 1 && lane < 8;
    unsigned ah_src = 0;
    if (ah_top) ah_src = (unsigned)((((lane & 7) ^ swz2(0)) * 8) * 2);
    if (ah_bot) ah_src = (unsigned)((((long)(BM + 1) * p.lda) + ((lane & 7) ^ swz2(BM + 1)) * 8) * 2);
    bool ah_ok = false;
    const bf16_t* a_base = nullptr; const bf16_t* b_base = nullptr; const bf16_t* h_base = nullptr;
    const bf16_t* a2_base = nullptr; const bf16_t* b2_base = nullptr;      // K extension
    auto set_tile = [&](int tm0, int tn0) __attribute__((always_inline)) {
      h_base = p.A + (long)tm0 * p.lda - p.lda;
      a_base = p.A + (long)tm0 * p.lda + (long)(8 * wave_u) * p.lda;
      b_base = p.B + (KBLK ? (long)tn0 * 32 : (long)tn0 * p.ldb) + (long)(8 * wave_u) * (KBLK ? 32 : p.ldb);
      if (XT) {
        a2_base = p.A2 + (long)tm0 * p.lda2 + (long)(8 * wave_u) * p.lda2;
        b2_base = p.B2 + (long)tn0 * 32 + (long)(8 * wave_u) * 32;
      }
      const bool top_ok = tm0 % p.L != 0, bot_ok = (tm0 + BM) % p.L != 0;       // wave-uniform
      ah_ok = (ah_top && top_ok) || (ah_bot && bot_ok);
    };
    // DMA instruction k of piece (s, t), per loader wave: k < 8 the B piece; 8..13 the A tile of stage s, 14 its halo rows (t == 0 only).
    // The source bases are RUNNING scalar pointers (begin_piece, then one step per instruction): written as base + k * step the
    // compiler hoists one 64-bit base per (tap, k) out of the K loop -- 60 SGPRs, spilled.
    constexpr int NB = 32 / NLOAD, NA = 24 / NLOAD;
    const bf16_t* bcur = nullptr; const bf16_t* acur = nullptr;
    auto begin_piece = [&](int s, int t) __attribute__((always_inline)) {
      const int tw = FLIP ? 2 - t : t;
      unsigned long long b = (unsigned long long)(b_base + s * bstep + tw * p.sBt), a = (unsigned long long)(a_base + (long)s * BK);
      asm volatile("" : "+s"(b), "+s"(a));
      bcur = (const bf16_t*)b; acur = (const bf16_t*)a;
    };
    auto issue = [&](int s, int t, int k) __attribute__((always_inline)) {
      if (k < NB) {
        dma16s(bcur, b_voff, lds0 + B0 + t * B_ALLOC + (wave_u + NLOAD * k) * 1024);
        bcur += b_kstep;
      } else if (k < NB + NA) {
        const int ka = k - NB;
        dma16s(acur, a_voff, lds0 + (s & 1) * A_ALLOC + A_PAD + ROWB + (wave_u + NLOAD * ka) * 1024);
        acur += a_kstep;
      } else if (wave_u < 2) {
        const bf16_t* hb = h_base + (long)s * BK;         // base of the row above the tile
        const void* src = ah_ok ? (const void*)((const char*)hb + ah_src) : p.zero_page;
        dma16(src, lds0 + (s & 1) * A_ALLOC + (wave_u == 0 ? 0 : A_PAD + (BM + 1) * ROWB));
      }
    };
    auto issue_all = [&](int pc) __attribute__((always_inline)) {
      const int s = pc / TAPS, t = pc % TAPS;
      begin_piece(s, t);
#pragma unroll
      for (int k = 0; k < NB + NA + 1; k++) if (k < NB || t == 0) issue(s, t, k);
    };
    // K extension: X piece e = NB weight instructions into B buffer `bb` + NA instructions of the A2 tile into A buffer `ab` (no halo rows:
    // one tap).  The per-lane A2 offset differs from a_voff only by the leading dimension.
    const unsigned a2_voff = XT ? (unsigned)(((long)r8 * p.lda2 + a_lc * 8) * 2) : 0u;
    const long a2_kstep = XT ? 8 * NLOAD * p.lda2 : 0;
    auto begin_xpiece = [&](int e) __attribute__((always_inline)) {
      unsigned long long b = (unsigned long long)(b2_base + e * bstep), a = (unsigned long long)(a2_base + (long)e * BK);
      asm volatile("" : "+s"(b), "+s"(a));
      bcur = (const bf16_t*)b; acur = (const bf16_t*)a;
    };
    auto issue_x = [&](unsigned ab_off, unsigned bb_off, int k) __attribute__((always_inline)) {
      if (k < NB) {
        dma16s(bcur, b_voff, lds0 + B0 + bb_off + (wave_u + NLOAD * k) * 1024);
        bcur += b_kstep;
      } else {
        dma16s(acur, a2_voff, lds0 + ab_off + A_PAD + ROWB + (wave_u + NLOAD * (k - NB)) * 1024);
        acur += a2_kstep;
      }
    };
    // counted wait of a loader: at most the instructions of piece p+2 (tap `t_inflight`) still in flight; pieces with an A tile carry
    // 14 instructions, 15 in waves 0 and 1; the others 8.  `c`: the previous tile's NST stores sit between pieces 0-2 and piece 3.
    auto wait_dma = [&](const int t_inflight, const bool any, const bool c) __attribute__((always_inline)) {
      if (!any) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (t_inflight != 0) {
        if (c) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB) : "memory");
      } else if (c) {
        if (wave_u < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + NA + 1 + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + NA + NST) : "memory");
      } else {
        if (wave_u < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + NA + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB + NA) : "memory");
      }
    };

    f32x4 acc[FM][FN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    // bias / embedding row / residual, added in the fragment layout (fp32) right behind the K loop and BEFORE the next tile's pieces
    // are requested (in-order vmcnt: a load issued behind the DMAs would wait for them)
    float4 add[FN];
    auto load_operands = [&](int tm0, int tn0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < FN; j++) {
        const int nj = tn0 + wn * 64 + j * 16 + q * 4;
        add[j] = p.bias ? *(const float4*)(p.bias + nj) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (XT && p.bias2) { const float4 e = *(const float4*)(p.bias2 + nj); add[j].x += e.x; add[j].y += e.y; add[j].z += e.z; add[j].w += e.w; }
        if (p.rowvec) {           // a tile lies inside one sample (rows_per_vec % 192 == 0)
          const float4 e = *(const float4*)(p.rowvec + (long)(tm0 / p.rows_per_vec) * p.ld_rowvec + nj);
          add[j].x += e.x; add[j].y += e.y; add[j].z += e.z; add[j].w += e.w;
        }
      }
      // the whole residual tile is requested at once (48 registers: the fragment registers are idle here) -- with a one-row lookahead
      // the six rows cost six HBM round trips: 15 us of a 76 us launch (tools/debug/gemm_big_ablate.sh, bits 32)
    };
    auto apply_operands = [&](int tm0, int tn0) __attribute__((always_inline)) {
      uint2 res[FM][FN];
      if (p.resid) {      // (one block for the loads and their use: split over two `if (p.resid)` the tile stays live across the K loop)
#pragma unroll
        for (int i = 0; i < FM; i++) {
          const bf16_t* rp = p.resid + (long)(tm0 + wm * (FM * 16) + i * 16 + lm) * p.ldr + tn0 + wn * 64 + q * 4;
#pragma unroll
          for (int j = 0; j < FN; j++) res[i][j] = *(const uint2*)(rp + j * 16);
        }
      }
      // (bias + embedding row + residual) first, then onto the accumulator: the summation order of big_epilogue / gemm.hip, bit for bit
      if (p.resid) {
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
          for (int j = 0; j < FN; j++) {
            const uint2 r = res[i][j];
            acc[i][j][0] += add[j].x + w16_lo<T>(r.x); acc[i][j][1] += add[j].y + w16_hi<T>(r.x);
            acc[i][j][2] += add[j].z + w16_lo<T>(r.y); acc[i][j][3] += add[j].w + w16_hi<T>(r.y);
          }
      } else {
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
          for (int j = 0; j < FN; j++) { acc[i][j][0] += add[j].x; acc[i][j][1] += add[j].y; acc[i][j][2] += add[j].z; acc[i][j][3] += add[j].w; }
      }
    };

    int tile_m0 = 0, tile_n0 = 0;      // the current tile, for the operand prefetch inside its last phase
    uint4 af[FM], bf0[FN], bf1[FN];
    // one phase (schedule of gemm_big_kernel: k-step 0 | wait piece p+1 + barrier | issue piece p+3, k-step 1)
    // xi >= 0 (K extension): X piece xi is requested in the second half of this phase (instead of main piece p + 3); nx: the NEXT piece is X_0
    auto phase = [&](const int s, const int t, const bool has1, const bool has2, const bool has3, const bool c, const bool pre = false, const int xi = -1, const bool nx = false) __attribute__((always_inline)) {
      const char* smA = smem + (s & 1) * A_ALLOC;
      const char* smB = smem + B0 + t * B_ALLOC;
      const int t1 = (t + 1) % TAPS, s1 = s + (t + 1) / TAPS;
      const int t2 = (t + 2) % TAPS;
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = big_mma<T>(bf0[j], af[i], acc[i][j]);
        if (i == 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FN; j++) bf1[j] = *(const uint4*)(smB + (bof ^ 64) + j * 2048);
        }
        af[i] = *(const uint4*)(smA + (aof[t] ^ 64) + i * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LOADER && has1) wait_dma(t2, has2, c);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* smA1 = smem + (s1 & 1) * A_ALLOC;
      const char* smB1 = smem + B0 + t1 * B_ALLOC;
      if (has1) {
#pragma unroll
        for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(smB1 + bof + j * 2048);
      }
      // last phase of a tile: nothing is in flight and nothing is read any more -- the epilogue operands are requested here and arrive
      // under the last 24 MFMAs
      if (pre && !BIG_DBG(32)) load_operands(tile_m0, tile_n0);
      __builtin_amdgcn_sched_barrier(0);
      const int n_dma = xi >= 0 ? NB + NA : (t == 0 ? NB + NA + 1 : NB);        // instructions of piece p+3 = (s + 1, t) / of X piece xi, spread over the 24 MFMA slots
      if (LOADER && has3) begin_piece(s + 1, t);
      if (XT && LOADER && xi >= 0) begin_xpiece(xi);
      const unsigned x_ab = (unsigned)(((S + xi) & 1) * A_ALLOC), x_bb = (unsigned)((xi % 3) * B_ALLOC);
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++) {
          acc[i][j] = big_mma<T>(bf1[j], af[i], acc[i][j]);
          if (LOADER && (has3 || (XT && xi >= 0))) {
#pragma unroll
            for (int k = 0; k < NB + NA + 1; k++) if (k < n_dma && (k * (FM * FN)) / n_dma == i * FN + j) { if (XT && xi >= 0) issue_x(x_ab, x_bb, k); else issue(s + 1, t, k); }
          }
        }
        if (has1) af[i] = *(const uint4*)(smA1 + aof[nx ? 1 : t1] + i * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    // K extension, phase of X piece e (A buffer (S + e) & 1 at byte offset `ab`, B buffer e % 3 at `bb`; centre-tap fragment addresses):
    // k-step 0 | X_{e+1} landed (vmcnt(0): it is the only piece in flight) + barrier | request X_{e+2} into the buffers X_e vacates, k-step 1
    auto xphase = [&](const int e, const unsigned ab, const unsigned bb, const unsigned bb1, const bool has1, const bool has2, const bool pre) __attribute__((always_inline)) {
      const char* smA = smem + ab;
      const char* smB = smem + B0 + bb;
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = big_mma<T>(bf0[j], af[i], acc[i][j]);
        if (i == 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FN; j++) bf1[j] = *(const uint4*)(smB + (bof ^ 64) + j * 2048);
        }
        af[i] = *(const uint4*)(smA + (aof[1] ^ 64) + i * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LOADER && has1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* smA1 = smem + (A_ALLOC - ab);        // the other A buffer
      const char* smB1 = smem + B0 + bb1;
      if (has1) {
#pragma unroll
        for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(smB1 + bof + j * 2048);
      }
      if (pre && !BIG_DBG(32)) load_operands(tile_m0, tile_n0);
      __builtin_amdgcn_sched_barrier(0);
      constexpr int n_dma = NB + NA;
      if (LOADER && has2) begin_xpiece(e + 2);
      const unsigned bb2 = bb == 0 ? 2u * B_ALLOC : bb - B_ALLOC;      // (e + 2) % 3 = (e - 1) % 3
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++) {
          acc[i][j] = big_mma<T>(bf1[j], af[i], acc[i][j]);
          if (LOADER && has2) {
#pragma unroll
            for (int k = 0; k < n_dma; k++) if ((k * (FM * FN)) / n_dma == i * FN + j) issue_x(ab, bb2, k);
          }
        }
        if (has1) af[i] = *(const uint4*)(smA1 + aof[1] + i * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    // statistics epilogue state: the wave's partial moments of the tile just finished; partner area in LDS (column group wn)
    int m0, n0;
    decode(blockIdx.x, m0, n0);
    if (LOADER) { set_tile(m0, n0); issue_all(0); issue_all(1); issue_all(2); }
    zero_acc();
    bool carry = false;
    int w = blockIdx.x;
#pragma unroll 1
    for (;;) {
      // piece 0 landed (pieces 1 and 2 and the previous tile's stores may still fly); first fragments
      if (LOADER) {
        if (carry) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NB) : "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(smem + B0 + bof + j * 2048);
#pragma unroll
      for (int i = 0; i < FM; i++) af[i] = *(const uint4*)(smem + aof[0] + i * 2048);
      bool c = carry;
#pragma unroll 1
      for (int s = 0; s + 1 < S; s++) {
        phase(s, 0, true, true, true, c); phase(s, 1, true, true, true, c); phase(s, 2, true, true, true, false);
        c = false;
      }
      tile_m0 = m0; tile_n0 = n0;
      if constexpr (!XT) {
        phase(S - 1, 0, true, true, false, c); phase(S - 1, 1, true, false, false, false); phase(S - 1, 2, false, false, false, false, true);
      } else {
        // X_0 is requested behind the barrier of (S-1, 1), X_1 behind that of (S-1, 2) (their A buffers are free only then); S2 >= 2
        phase(S - 1, 0, true, true, false, c); phase(S - 1, 1, true, false, false, false, false, 0); phase(S - 1, 2, true, false, false, false, false, 1, true);
        unsigned ab = (unsigned)((S & 1) * A_ALLOC), bb = 0;
#pragma unroll 1
        for (int e = 0; e + 2 < S2; e++) {
          const unsigned bb1 = bb == 2u * B_ALLOC ? 0u : bb + B_ALLOC;
          xphase(e, ab, bb, bb1, true, true, false);
          ab = A_ALLOC - ab; bb = bb1;
        }
        {
          const unsigned bb1 = bb == 2u * B_ALLOC ? 0u : bb + B_ALLOC;
          xphase(S2 - 2, ab, bb, bb1, true, false, false);
          ab = A_ALLOC - ab; bb = bb1;
          xphase(S2 - 1, ab, bb, 0u, false, false, true);
        }
      }
      // nobody reads the ring behind the last barrier
      const int w1 = w + (int)gridDim.x;
      const bool more = w1 < ntiles;
      int m1 = 0, n1 = 0;
      if (!BIG_DBG(32)) apply_operands(m0, n0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        decode(w1, m1, n1);
        if (LOADER) { set_tile(m1, n1); issue_all(0); issue_all(1); issue_all(2); }
        __builtin_amdgcn_sched_barrier(0);
      }
      big_store<T>(p, acc, m0, n0, lm, q, wm, wn);
      if (!more) return;
      __builtin_amdgcn_sched_barrier(0);
      zero_acc();
      m0 = m1; n0 = n1; w = w1; carry = true;
    }
  };
  if (wave_u < NLOAD) run(std::true_type{}); else run(std::false_type{});
}

// ---- one tap (1 x 1 convs / NT products): Y[r][n] = sum_k A[r][k] * W[n][k].  Every K stage carries its own A tile, so the ring is
// two (A, B) stage buffers (112 KB); same phase shape as the 3-tap kernel (barrier between the two k-steps, fragment reads one k-step
// ahead), with stage s + 2 issued behind B_s into the buffer stage s vacates: one phase of DMA lead.  Measured (tools/debug/
// gemm_big_check.py): equal to the 192 x 128 / 2-blocks-per-CU kernel of gemm.hip on the model's 1 x 1 shapes (their K loops are 8-16
// stages, so prologue + epilogue weigh as much as the loop and two co-resident blocks overlap them); kept for the data gradients,
// which it runs as NT products on the transposed weight copy (+3 % over the transposed-operand kernel).
constexpr int S1_BYTES = A_MAIN + B_ALLOC;          // 56 KB per stage
constexpr int LDS1_BYTES = 2 * S1_BYTES > EPI_BYTES ? 2 * S1_BYTES : EPI_BYTES;
template <bool KBLK, typename T = bf16_t>      // KBLK: B is [K / 32][N][32] (the data-gradient copy of a 1 x 1 conv weight), else plain [N][ldb]
__global__ __launch_bounds__(NTHR, 2) void gemm_big1_kernel(const BigArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  int tile_m, tile_n;
  {
    const int w = blockIdx.x;
    if ((p.tiles_m & 7) == 0) { const int xcd = w & 7, slot = w >> 3; tile_n = slot % p.tiles_n; tile_m = (slot / p.tiles_n) * 8 + xcd; }
    else { tile_n = w % p.tiles_n; tile_m = w / p.tiles_n; }
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int S = p.K / BK;
  unsigned ao_src[3], bo_src[4];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const int c = (wave + NWAVE * j) * 64 + lane, pr = c >> 3, lc = (c & 7) ^ swz2(pr);
    ao_src[j] = (unsigned)(((long)pr * p.lda + lc * 8) * 2);
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = (wave + NWAVE * j) * 64 + lane, nr = c >> 3, lc = (c & 7) ^ swz2(nr);
    bo_src[j] = KBLK ? (unsigned)((((long)(lc >> 2) * p.N + nr) * 32 + (lc & 3) * 8) * 2) : (unsigned)(((long)nr * p.ldb + lc * 8) * 2);
  }
  const long bstep = KBLK ? (long)2 * p.N * 32 : (long)BK;      // elements per K stage
  const bf16_t* const a_base = p.A + (long)m0 * p.lda;
  const bf16_t* const b_base = p.B + (KBLK ? (long)n0 * 32 : (long)n0 * p.ldb);
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto issue = [&](int s, int k) __attribute__((always_inline)) {       // DMA instruction k (0..3 B, 4..6 A) of stage s
    if (k < 4) dma16s(b_base + s * bstep, bo_src[k], lds0 + (s & 1) * S1_BYTES + A_MAIN + (wave_u + NWAVE * k) * 1024);
    else dma16s(a_base + (long)s * BK, ao_src[k - 4], lds0 + (s & 1) * S1_BYTES + (wave_u + NWAVE * (k - 4)) * 1024);
  };
  const int pr0 = wm * (FM * 16) + lm;
  const unsigned aof = pr0 * ROWB + ((q ^ swz2(pr0)) << 4);
  const unsigned bof = A_MAIN + (wn * 64 + lm) * ROWB + ((q ^ swz2(lm)) << 4);

  f32x4 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int j = 0; j < FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 7; k++) issue(0, k);
  if (S > 1) {
#pragma unroll
    for (int k = 0; k < 7; k++) issue(1, k);
  }
  uint4 af[FM], bf0[FN], bf1[FN];
  if (S > 1) asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stage 0 landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#pragma unroll
  for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(smem + bof + j * 2048);
#pragma unroll
  for (int i = 0; i < FM; i++) af[i] = *(const uint4*)(smem + aof + i * 2048);
  // phase s (same shape as the 3-tap kernel): k-step 0 | wait stage s+1 + barrier | issue stage s+2 into the buffer stage s vacates, k-step 1
  auto phase = [&](const int s, const bool has1, const bool has2) __attribute__((always_inline)) {
    const char* sm = smem + (s & 1) * S1_BYTES;
#pragma unroll
    for (int i = 0; i < FM; i++) {
#pragma unroll
      for (int j = 0; j < FN; j++)
        acc[i][j] = big_mma<T>(bf0[j], af[i], acc[i][j]);
      if (i == 0) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < FN; j++) bf1[j] = *(const uint4*)(sm + (bof ^ 64) + j * 2048);
      }
      af[i] = *(const uint4*)(sm + (aof ^ 64) + i * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // stage s+1 (the only DMA in flight) landed; reads of stage s returned
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* sm1 = smem + ((s + 1) & 1) * S1_BYTES;
    if (has1) {
#pragma unroll
      for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(sm1 + bof + j * 2048);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < FM; i++) {
#pragma unroll
      for (int j = 0; j < FN; j++) {
        acc[i][j] = big_mma<T>(bf1[j], af[i], acc[i][j]);
        if (has2 && i <= 3) { const int k = 2 * i + (j >> 1); if ((j & 1) && k < 7) issue(s + 2, k); }
      }
      if (has1) af[i] = *(const uint4*)(sm1 + aof + i * 2048);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
#pragma unroll 1
  for (int s = 0; s + 2 < S; s++) phase(s, true, true);
  if (S > 1) phase(S - 2, true, false);
  phase(S - 1, false, false);
  __syncthreads();
  big_epilogue<T>(p, acc, smem, m0, n0, tid, lm, q, wm, wn);
}

// ---- persistent form of the 1-tap kernel (structure and reasons: gemm_bigp_kernel).  Ring = two (A, B) stage buffers; behind the last
// barrier of a tile both are free, so stages 0 and 1 of the next tile are requested before the output leaves; stage 2 follows behind
// the first barrier of the next tile, and the loaders' stores have until its second barrier.
template <bool KBLK, int NLOAD, typename T = bf16_t>
__global__ __launch_bounds__(NTHR, 2) void gemm_big1p_kernel(const BigArgs p) {
  static_assert(NLOAD == 2 || NLOAD == 4, "vmcnt holds 63");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, q = lane >> 4;
  const int wm = wave >> 2, wn = wave & 3;
  const int ntiles = p.tiles_m * p.tiles_n;
  auto decode = [&](int w, int& tm0, int& tn0) __attribute__((always_inline)) {
    int tile_m, tile_n;
    if ((p.tiles_m & 7) == 0) { const int xcd = w & 7, slot = w >> 3; tile_n = slot % p.tiles_n; tile_m = (slot / p.tiles_n) * 8 + xcd; }
    else { tile_n = w % p.tiles_n; tile_m = w / p.tiles_n; }
    tm0 = tile_m * BM; tn0 = tile_n * BN;
  };
  const int S = p.K / BK;
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);
  const int pr0 = wm * (FM * 16) + lm;
  const unsigned aof = pr0 * ROWB + ((q ^ swz2(pr0)) << 4);
  const unsigned bof = A_MAIN + (wn * 64 + lm) * ROWB + ((q ^ swz2(lm)) << 4);

  auto run = [&](auto loader_tag) __attribute__((always_inline)) {
    constexpr bool LOADER = decltype(loader_tag)::value;
    const int r8 = lane >> 3;
    const int lc = (lane & 7) ^ swz2(r8);
    const unsigned a_voff = (unsigned)(((long)r8 * p.lda + lc * 8) * 2);
    const unsigned b_voff = KBLK ? (unsigned)((((long)(lc >> 2) * p.N + r8) * 32 + (lc & 3) * 8) * 2) : (unsigned)(((long)r8 * p.ldb + lc * 8) * 2);
    const long a_kstep = 8 * NLOAD * p.lda, b_kstep = KBLK ? (long)8 * NLOAD * 32 : 8 * NLOAD * p.ldb;
    const long bstep = KBLK ? (long)2 * p.N * 32 : (long)BK;
    const bf16_t* a_base = nullptr; const bf16_t* b_base = nullptr;
    auto set_tile = [&](int tm0, int tn0) __attribute__((always_inline)) {
      a_base = p.A + (long)tm0 * p.lda + (long)(8 * wave_u) * p.lda;
      b_base = p.B + (KBLK ? (long)tn0 * 32 : (long)tn0 * p.ldb) + (long)(8 * wave_u) * (KBLK ? 32 : p.ldb);
    };
    constexpr int NB = 32 / NLOAD, NA = 24 / NLOAD, NS = NB + NA;       // DMA instructions of one stage per loader wave
    const bf16_t* bcur = nullptr; const bf16_t* acur = nullptr;
    auto begin_stage = [&](int s) __attribute__((always_inline)) {
      unsigned long long b = (unsigned long long)(b_base + s * bstep), a = (unsigned long long)(a_base + (long)s * BK);
      asm volatile("" : "+s"(b), "+s"(a));
      bcur = (const bf16_t*)b; acur = (const bf16_t*)a;
    };
    auto issue = [&](int s, int k) __attribute__((always_inline)) {
      if (k < NB) {
        dma16s(bcur, b_voff, lds0 + (s & 1) * S1_BYTES + A_MAIN + (wave_u + NLOAD * k) * 1024);
        bcur += b_kstep;
      } else {
        dma16s(acur, a_voff, lds0 + (s & 1) * S1_BYTES + (wave_u + NLOAD * (k - NB)) * 1024);
        acur += a_kstep;
      }
    };
    auto issue_all = [&](int s) __attribute__((always_inline)) {
      begin_stage(s);
#pragma unroll
      for (int k = 0; k < NS; k++) issue(s, k);
    };

    f32x4 acc[FM][FN];
    auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    float4 add[FN];
    auto load_operands = [&](int tm0, int tn0) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < FN; j++) {
        const int nj = tn0 + wn * 64 + j * 16 + q * 4;
        add[j] = p.bias ? *(const float4*)(p.bias + nj) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.rowvec) {
          const float4 e = *(const float4*)(p.rowvec + (long)(tm0 / p.rows_per_vec) * p.ld_rowvec + nj);
          add[j].x += e.x; add[j].y += e.y; add[j].z += e.z; add[j].w += e.w;
        }
      }
      // the whole residual tile is requested at once (48 registers: the fragment registers are idle here) -- with a one-row lookahead
      // the six rows cost six HBM round trips: 15 us of a 76 us launch (tools/debug/gemm_big_ablate.sh, bits 32)
    };
    auto apply_operands = [&](int tm0, int tn0) __attribute__((always_inline)) {
      uint2 res[FM][FN];
      if (p.resid) {      // (one block for the loads and their use: split over two `if (p.resid)` the tile stays live across the K loop)
#pragma unroll
        for (int i = 0; i < FM; i++) {
          const bf16_t* rp = p.resid + (long)(tm0 + wm * (FM * 16) + i * 16 + lm) * p.ldr + tn0 + wn * 64 + q * 4;
#pragma unroll
          for (int j = 0; j < FN; j++) res[i][j] = *(const uint2*)(rp + j * 16);
        }
      }
      // (bias + embedding row + residual) first, then onto the accumulator: the summation order of big_epilogue / gemm.hip, bit for bit
      if (p.resid) {
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
          for (int j = 0; j < FN; j++) {
            const uint2 r = res[i][j];
            acc[i][j][0] += add[j].x + w16_lo<T>(r.x); acc[i][j][1] += add[j].y + w16_hi<T>(r.x);
            acc[i][j][2] += add[j].z + w16_lo<T>(r.y); acc[i][j][3] += add[j].w + w16_hi<T>(r.y);
          }
      } else {
#pragma unroll
        for (int i = 0; i < FM; i++)
#pragma unroll
          for (int j = 0; j < FN; j++) { acc[i][j][0] += add[j].x; acc[i][j][1] += add[j].y; acc[i][j][2] += add[j].z; acc[i][j][3] += add[j].w; }
      }
    };

    int tile_m0 = 0, tile_n0 = 0;
    uint4 af[FM], bf0[FN], bf1[FN];
    // phase s: k-step 0 | wait stage s+1 + barrier | issue stage s+2 into the buffer stage s vacates, k-step 1.  `c`: the previous tile's
    // stores are the only instructions younger than stage s+1 (phase 0 of a tile)
    auto phase = [&](const int s, const bool has1, const bool has2, const bool c, const bool pre = false) __attribute__((always_inline)) {
      const char* sm = smem + (s & 1) * S1_BYTES;
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++)
          acc[i][j] = big_mma<T>(bf0[j], af[i], acc[i][j]);
        if (i == 0) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < FN; j++) bf1[j] = *(const uint4*)(sm + (bof ^ 64) + j * 2048);
        }
        af[i] = *(const uint4*)(sm + (aof ^ 64) + i * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (LOADER && has1) {
        if (c) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char* sm1 = smem + ((s + 1) & 1) * S1_BYTES;
      if (has1) {
#pragma unroll
        for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(sm1 + bof + j * 2048);
      }
      if (pre) load_operands(tile_m0, tile_n0);      // (see gemm_bigp_kernel)
      __builtin_amdgcn_sched_barrier(0);
      if (LOADER && has2) begin_stage(s + 2);
#pragma unroll
      for (int i = 0; i < FM; i++) {
#pragma unroll
        for (int j = 0; j < FN; j++) {
          acc[i][j] = big_mma<T>(bf1[j], af[i], acc[i][j]);
          if (LOADER && has2) {
#pragma unroll
            for (int k = 0; k < NS; k++) if ((k * (FM * FN)) / NS == i * FN + j) issue(s + 2, k);
          }
        }
        if (has1) af[i] = *(const uint4*)(sm1 + aof + i * 2048);
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    int m0, n0;
    decode(blockIdx.x, m0, n0);
    if (LOADER) { set_tile(m0, n0); issue_all(0); if (S > 1) issue_all(1); }
    zero_acc();
    bool carry = false;
    int w = blockIdx.x;
#pragma unroll 1
    for (;;) {
      if (LOADER) {      // stage 0 landed (stage 1 and the previous tile's stores may still fly)
        if (S > 1) {
          if (carry) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS + NST) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NS) : "memory");
        } else {
          if (carry) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int j = 0; j < FN; j++) bf0[j] = *(const uint4*)(smem + bof + j * 2048);
#pragma unroll
      for (int i = 0; i < FM; i++) af[i] = *(const uint4*)(smem + aof + i * 2048);
      bool c = carry;
#pragma unroll 1
      for (int s = 0; s + 2 < S; s++) { phase(s, true, true, c); c = false; }
      if (S > 1) { phase(S - 2, true, false, c); c = false; }
      tile_m0 = m0; tile_n0 = n0;
      phase(S - 1, false, false, false, true);
      const int w1 = w + (int)gridDim.x;
      const bool more = w1 < ntiles;
      int m1 = 0, n1 = 0;
      apply_operands(m0, n0);
      __builtin_amdgcn_sched_barrier(0);
      if (more) {
        decode(w1, m1, n1);
        if (LOADER) { set_tile(m1, n1); issue_all(0); if (S > 1) issue_all(1); }
        __builtin_amdgcn_sched_barrier(0);
      }
      big_store<T>(p, acc, m0, n0, lm, q, wm, wn);
      if (!more) return;
      __builtin_amdgcn_sched_barrier(0);
      zero_acc();
      m0 = m1; n0 = n1; w = w1; carry = true;
    }
  };
  if (wave_u < NLOAD) run(std::true_type{}); else run(std::false_type{});
}

template <bool KBLK, int NLOAD, typename T = bf16_t>
int launch_big1p(eegldm_ctx* ctx, const BigArgs& a) {
  auto kern = gemm_big1p_kernel<KBLK, NLOAD, T>;
  static DevOnce attr_once;
  constexpr int LDSB = 2 * S1_BYTES;
  if (attr_once.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  int grid = a.tiles_m * a.tiles_n;
  const int cus = ctx->num_cu >= 8 ? ctx->num_cu & ~7 : ctx->num_cu;
  if (grid > cus) grid = cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), LDSB, ctx->stream, a);
  LAUNCH_CHECK();
  return 0;
}

template <bool KBLK, typename T = bf16_t>
int launch_big1(eegldm_ctx* ctx, const BigArgs& a) {
  EEG_ENV_VAR(bool, no_persist, getenv("EEGLDM_GEMM_BIG_NO_PERSIST") != nullptr || getenv("EEGLDM_GEMM_BIG1_NO_PERSIST") != nullptr);
  // (fp16 runs the persistent form only: the one-tile-per-workgroup kernels are a bf16 developer A/B)
  if (!no_persist || !Is16<T>::bf16) return launch_big1p<KBLK, 4, T>(ctx, a);
  if constexpr (Is16<T>::bf16) {
    auto kern = gemm_big1_kernel<KBLK, T>;
    static DevOnce attr_once;
    if (attr_once.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS1_BYTES));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(NTHR), LDS1_BYTES, ctx->stream, a);
    LAUNCH_CHECK();
  }
  return 0;
}

template <bool KBLK, bool FLIP, int NLOAD, typename T = bf16_t>
int launch_bigp(eegldm_ctx* ctx, const BigArgs& a) {
  auto kern = gemm_bigp_kernel<KBLK, FLIP, NLOAD, false, T>;
  static DevOnce attr_once;      // the dynamic-LDS attribute is per device
  constexpr int LDSB = RING;
  if (attr_once.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  int grid = a.tiles_m * a.tiles_n;
  const int cus = ctx->num_cu >= 8 ? ctx->num_cu & ~7 : ctx->num_cu;      // a multiple of 8 keeps a workgroup's tiles on its XCD
  if (grid > cus) grid = cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), LDSB, ctx->stream, a);
  LAUNCH_CHECK();
  return 0;
}

template <typename T = bf16_t>
int launch_bigx(eegldm_ctx* ctx, const BigArgs& a) {      // K extension: persistent form only
  auto kern = gemm_bigp_kernel<true, false, 4, true, T>;
  static DevOnce attr_once;
  constexpr int LDSB = RING;
  if (attr_once.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  int grid = a.tiles_m * a.tiles_n;
  const int cus = ctx->num_cu >= 8 ? ctx->num_cu & ~7 : ctx->num_cu;
  if (grid > cus) grid = cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NTHR), LDSB, ctx->stream, a);
  LAUNCH_CHECK();
  return 0;
}

template <int TAPS, bool KBLK, bool FLIP, typename T = bf16_t>
int launch_big(eegldm_ctx* ctx, const BigArgs& a) {
  EEG_ENV_VAR(bool, no_persist, getenv("EEGLDM_GEMM_BIG_NO_PERSIST") != nullptr);
  if (!no_persist || !Is16<T>::bf16) {      // (2 loader waves measured equal: tools/debug/gemm_big_check.py, round 4)
    return launch_bigp<KBLK, FLIP, 4, T>(ctx, a);
  }
  if constexpr (Is16<T>::bf16) {
    auto kern = gemm_big_kernel<TAPS, KBLK, FLIP, T>;
    static DevOnce attr_once;      // the dynamic-LDS attribute is per device
    if (attr_once.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(NTHR), LDS_BYTES, ctx->stream, a);
    LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace

// 1 = launched, 0 = not this kernel's shape (the caller takes gemm.hip), < 0 = error.
// GemmArgs of a forward-style product: A rows (GA_CONV with 3 taps, stride 1, pad 1; or GA_PLAIN with 1 tap), B = GB_NT weights (plain or
// K-blocked), bf16 in and out.
int gemm_big_try(eegldm_ctx* ctx, const GemmArgs& g) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_NO_GEMM_BIG") != nullptr);
  EEG_ENV_VAR(int, min_tiles, getenv("EEGLDM_GEMM_BIG_MIN_TILES") ? atoi(getenv("EEGLDM_GEMM_BIG_MIN_TILES")) : 128);
  if (off || (g.dtype != EEGLDM_BF16 && g.dtype != EEGLDM_F16) || g.bmode != GB_NT || g.batch != 1 || g.splitk > 1 || g.ztaps > 1 || g.out_f32 || g.atomic_out || g.colsum || g.ngroup) return 0;
  if (g.alpha != 1.0f || g.ups > 1) return 0;
  const bool conv3 = g.amode == GA_CONV && g.taps == 3 && g.stride == 1 && g.pad_l == 1 && g.Lin == g.Lout;
  const bool plain1 = g.amode == GA_PLAIN && g.taps == 1;
  EEG_ENV_VAR(bool, no1, getenv("EEGLDM_NO_GEMM_BIG1") != nullptr);
  if (!conv3 && !(plain1 && !no1)) return 0;
  const int L = conv3 ? g.Lout : BM;
  if (g.M % BM != 0 || L % BM != 0 || g.N % BN != 0 || g.K % BK != 0 || g.lda % 8 != 0 || g.ldc % 8 != 0) return 0;
  if (!g.b_kblk && g.ldb % 8 != 0) return 0;
  if (g.rowvec && (g.rows_per_vec % BM != 0 || g.ld_rowvec % 4 != 0)) return 0;
  if (g.resid && g.ldr % 4 != 0) return 0;
  if (((size_t)g.A | (size_t)g.B | (size_t)g.C) & 15) return 0;
  if (g.resid && ((size_t)g.resid & 7)) return 0;
  const int tm = g.M / BM, tn = g.N / BN;
  if ((long)tm * tn < min_tiles) return 0;         // small problems: the 128-row tiles fill the chip better
  BigArgs a;
  a.A = (const bf16_t*)g.A; a.lda = g.lda; a.B = (const bf16_t*)g.B; a.ldb = g.ldb; a.sBt = g.sBt;
  a.C = (bf16_t*)g.C; a.ldc = g.ldc; a.M = g.M; a.N = g.N; a.K = g.K; a.L = L;
  a.bias = g.bias; a.rowvec = g.rowvec; a.ld_rowvec = g.ld_rowvec; a.rows_per_vec = g.rows_per_vec > 0 ? g.rows_per_vec : 1;
  a.resid = (const bf16_t*)g.resid; a.ldr = g.ldr; a.zero_page = ctx->zero_page; a.tiles_m = tm; a.tiles_n = tn;
  a.A2 = nullptr; a.lda2 = 0; a.B2 = nullptr; a.K2 = 0; a.bias2 = nullptr;
  int rc;
  if (g.dtype == EEGLDM_F16) {      // round 5: the same kernels on IEEE-half operands (persistent forms)
    if (!conv3) rc = g.b_kblk ? launch_big1<true, f16_t>(ctx, a) : launch_big1<false, f16_t>(ctx, a);
    else if (g.tap_flip) rc = g.b_kblk ? launch_big<3, true, true, f16_t>(ctx, a) : launch_big<3, false, true, f16_t>(ctx, a);
    else rc = g.b_kblk ? launch_big<3, true, false, f16_t>(ctx, a) : launch_big<3, false, false, f16_t>(ctx, a);
  }
  else if (!conv3) rc = g.b_kblk ? launch_big1<true>(ctx, a) : launch_big1<false>(ctx, a);
  else if (g.tap_flip) rc = g.b_kblk ? launch_big<3, true, true>(ctx, a) : launch_big<3, false, true>(ctx, a);
  else rc = g.b_kblk ? launch_big<3, true, false>(ctx, a) : launch_big<3, false, false>(ctx, a);
  return rc < 0 ? rc : 1;
}

// Round-6 prototype: the 3-tap forward conv with act(x * scale + shift) applied to its operand tile IN LDS -- GroupNorm(+SiLU) /
// BatchNorm + LeakyReLU on the consumer's load.  g: as for gemm_big_try (GA_CONV, 3 taps, stride 1, pad 1, K-blocked weight), bf16.
// scale / shift: fp32 [M / L][ld] (ld = 0: one row for all samples).  act: 1 SiLU, 2 LeakyReLU(slope).  1 = launched, 0 = not eligible, < 0 = error.
int gemm_big_xf_try(eegldm_ctx* ctx, const GemmArgs& g, const float* scale, const float* shift, long ld, int act, float slope) {
  if (g.dtype != EEGLDM_BF16 || g.bmode != GB_NT || !g.b_kblk || g.batch != 1 || g.splitk > 1 || g.ztaps > 1 || g.out_f32 || g.atomic_out || g.colsum || g.ngroup || g.tap_flip) return 0;
  if (g.alpha != 1.0f || g.ups > 1 || !(g.amode == GA_CONV && g.taps == 3 && g.stride == 1 && g.pad_l == 1 && g.Lin == g.Lout)) return 0;
  if (g.M % BM != 0 || g.Lout % BM != 0 || g.N % BN != 0 || g.K % BK != 0 || g.lda % 8 != 0 || g.ldc % 8 != 0) return 0;
  if (g.rowvec && (g.rows_per_vec % BM != 0 || g.ld_rowvec % 4 != 0)) return 0;
  if (g.resid && (g.ldr % 4 != 0 || ((size_t)g.resid & 7))) return 0;
  if (((size_t)g.A | (size_t)g.B | (size_t)g.C) & 15) return 0;
  const int ldsb = (RING > EPI_BYTES ? RING : EPI_BYTES) + 2 * g.K * (int)sizeof(float);
  if (ldsb > 160 * 1024 || (act != 1 && act != 2) || !scale || !shift) return 0;
  BigArgs a;
  a.A = (const bf16_t*)g.A; a.lda = g.lda; a.B = (const bf16_t*)g.B; a.ldb = g.ldb; a.sBt = g.sBt;
  a.C = (bf16_t*)g.C; a.ldc = g.ldc; a.M = g.M; a.N = g.N; a.K = g.K; a.L = g.Lout;
  a.bias = g.bias; a.rowvec = g.rowvec; a.ld_rowvec = g.ld_rowvec; a.rows_per_vec = g.rows_per_vec > 0 ? g.rows_per_vec : 1;
  a.resid = (const bf16_t*)g.resid; a.ldr = g.ldr; a.zero_page = ctx->zero_page; a.tiles_m = g.M / BM; a.tiles_n = g.N / BN;
  a.A2 = nullptr; a.lda2 = 0; a.B2 = nullptr; a.K2 = 0; a.bias2 = nullptr;
  a.xf_scale = scale; a.xf_shift = shift; a.xf_ld = ld; a.xf_slope = slope;
  static DevOnce once1, once2;
  ProfRec rec; const bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = PROF_CONV_FWD; rec.flops = 2.0 * g.M * g.N * (double)g.K * 3; rec.M = g.M; rec.N = g.N; rec.K = g.K; rec.taps = 3; rec.splitk = 1;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b)); HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  if (act == 1) {
    auto kern = gemm_big_kernel<3, true, false, bf16_t, 1>;
    if (once1.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(NTHR), ldsb, ctx->stream, a);
  } else {
    auto kern = gemm_big_kernel<3, true, false, bf16_t, 2>;
    if (once2.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(NTHR), ldsb, ctx->stream, a);
  }
  LAUNCH_CHECK();
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return 1;
}

// The ResBlock tail  out = conv3(a2; W) + bias + skip_1x1(x2; W2) + bias2  as ONE launch (K extension of the persistent 3-tap kernel).
// g: the 3-tap forward conv (GA_CONV, stride 1, pad 1, K-blocked weight, no residual); w2_kblk: [K2 / 32][N][32] copy of the 1 x 1 weight.
// 1 = launched, 0 = not eligible (the caller launches the two convs separately), < 0 = error.
int gemm_big_skip_try(eegldm_ctx* ctx, const GemmArgs& g, const void* x2, long ldx2, const void* w2_kblk, int K2, const float* bias2) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_NO_GEMM_BIG") != nullptr || getenv("EEGLDM_NO_FUSED_SKIP") != nullptr || getenv("EEGLDM_GEMM_BIG_NO_PERSIST") != nullptr);
  EEG_ENV_VAR(int, min_tiles, getenv("EEGLDM_GEMM_BIG_MIN_TILES") ? atoi(getenv("EEGLDM_GEMM_BIG_MIN_TILES")) : 128);
  if (off || (g.dtype != EEGLDM_BF16 && g.dtype != EEGLDM_F16) || g.bmode != GB_NT || !g.b_kblk || g.batch > 1 || g.splitk > 1 || g.out_f32 || g.atomic_out || g.resid) return 0;
  if (!(g.amode == GA_CONV && g.taps == 3 && g.stride == 1 && g.pad_l == 1 && g.Lin == g.Lout) || g.ups > 1 || g.alpha != 1.0f) return 0;
  if (g.M % BM != 0 || g.Lout % BM != 0 || g.N % BN != 0 || g.K % BK != 0 || g.K < 2 * BK || g.lda % 8 != 0 || g.ldc % 8 != 0) return 0;
  if (!x2 || !w2_kblk || K2 % BK != 0 || K2 < 2 * BK || ldx2 % 8 != 0) return 0;
  if (g.rowvec && (g.rows_per_vec % BM != 0 || g.ld_rowvec % 4 != 0)) return 0;
  if (((size_t)g.A | (size_t)g.B | (size_t)g.C | (size_t)x2 | (size_t)w2_kblk) & 15) return 0;
  const int tm = g.M / BM, tn = g.N / BN;
  if ((long)tm * tn < min_tiles) return 0;
  BigArgs a;
  a.A = (const bf16_t*)g.A; a.lda = g.lda; a.B = (const bf16_t*)g.B; a.ldb = g.ldb; a.sBt = g.sBt;
  a.C = (bf16_t*)g.C; a.ldc = g.ldc; a.M = g.M; a.N = g.N; a.K = g.K; a.L = g.Lout;
  a.bias = g.bias; a.rowvec = g.rowvec; a.ld_rowvec = g.ld_rowvec; a.rows_per_vec = g.rows_per_vec > 0 ? g.rows_per_vec : 1;
  a.resid = nullptr; a.ldr = 0; a.zero_page = ctx->zero_page; a.tiles_m = tm; a.tiles_n = tn;
  a.A2 = (const bf16_t*)x2; a.lda2 = ldx2; a.B2 = (const bf16_t*)w2_kblk; a.K2 = K2; a.bias2 = bias2;
  ProfRec rec; const bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = PROF_CONV_FWD; rec.flops = 2.0 * g.M * g.N * ((double)g.K * 3 + K2);
    rec.M = g.M; rec.N = g.N; rec.K = g.K + K2 / 3; rec.taps = 3; rec.splitk = 1;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b));
    HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  const int rc = g.dtype == EEGLDM_F16 ? launch_bigx<f16_t>(ctx, a) : launch_bigx<bf16_t>(ctx, a);
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return rc < 0 ? rc : 1;
}
