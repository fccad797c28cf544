// Weight-stationary 3-tap convolution for the UNet's widest-and-shallowest layers (bf16, 128 reduction channels per tap).
//
// The level-0 convs of the config_ldm UNet (unet.py:263,291: Conv1d(128 -> 128, k3) over B x 768 rows) are 19.3 GFLOP against 2 x 50 MB
// of activations: HBM-bound (197 FLOP/B).  In the general implicit-GEMM kernel (gemm.hip) such a tile has only four K stages, so each
// 128 x 128 tile pays a 2.7 k-cycle prologue and a 5.7 k-cycle epilogue around a 10 k-cycle loop and re-stages the SAME 98 KB of
// weights it shares with every other tile: 45 us per launch = 2.2 TB/s, where a copy of the two tensors takes 15 us.
//
// Here the weights never move after the first instruction: every wave keeps its 32 output channels x (3 taps x 128) reduction
// elements as 24 MFMA operand fragments in 96 VGPRs for the whole launch, blocks are persistent over runs of 64-row tiles, and the only
// data that streams is the activation tile (66 rows x 256 B, LDS-DMA into a 2-deep ring, next tile in flight under this tile's MFMAs).
//   tile t:  wait DMA(t) | barrier | issue DMA(t+1) | 96 MFMA per wave (4 row fragments x 3 taps x 4 k-steps x 2 column fragments,
//            activation fragments by conflict-free ds_read_b128 from the XOR-swizzled tile, taps = row shifts of the same tile)
//            | + bias + time-embedding row vector + residual in the fragment layout, one rounding to bf16 | through LDS | whole
//            256-byte rows out.
// The data gradient of a 3-tap conv is the same kernel with the weight fragments gathered transposed and tap-flipped (once per block).
// Sample boundaries: L % 64 == 0, so a tile never straddles samples and only its two halo rows can belong to a neighbour: those
// rows are fetched from the zero page (= the conv's zero padding).
#include <stdlib.h>

#include "common.h"
#include "internal.h"

namespace {
constexpr int WS_ROWS = 64;                  // output rows per tile
constexpr int WS_BUF_ROWS = 68;              // 66 used (halo each side), rounded up to whole 4-row DMA instructions
constexpr int WS_ROW_BYTES = 256;            // 128 bf16
constexpr int WS_ABUF = WS_BUF_ROWS * WS_ROW_BYTES;      // 17 408 B
constexpr int WS_OUT = WS_ROWS * WS_ROW_BYTES;           // 16 384 B
constexpr int WS_RING = 3;                               // activation tiles in LDS: one being read, two in flight
constexpr int WS_LDS = WS_RING * WS_ABUF + WS_OUT;       // 68 608 B: two blocks per CU

struct WsArgs {
  const bf16_t* x; long ldx;
  const bf16_t* w; long sWt, sWn, sWk; int tflip;       // weight element (tap, out channel n, reduction index k) at w[tap * sWt + n * sWn + k * sWk]
  const float* bias; const float* rowvec; long ld_rowvec;
  const bf16_t* resid; long ldr;
  bf16_t* y; long ldy;
  int M, L, N, ntiles, tiles_per_block;
  const void* zero_page;
  // statistics epilogue (ST kernels, round 6): per-block partial (sum, sum of squares) of every output column over the block's tiles, fp32 of the
  // UNROUNDED outputs, row blockIdx.x of col_parts ([gridDim.x][2 N], interleaved) -- the batch statistics of the BatchNorm that reads this
  // tensor next (MONAI PatchDiscriminator layer = conv -> BatchNorm -> LeakyReLU), folded by losses.hip ls_bn_fold: no separate read of y
  float* col_parts;
};

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

// workgroup barrier WITHOUT the release fence of __syncthreads(): hipcc's fence would wait vmcnt(0) for the tile stores it knows about
// and thereby drain the LDS-DMA prefetch it does not know about.  LDS traffic is made visible by the explicit lgkmcnt(0).
__device__ __forceinline__ void ws_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

__device__ __forceinline__ void ws_dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(g) : "memory", "m0");
}
// 8-byte load hipcc does not see (cdna_hip_programming.md 5.7 form (ii)): its completion is counted by hand, so that hipcc's own
// wait for it cannot drain the LDS-DMA queue that was issued after it
__device__ __forceinline__ u32x2 ws_load8(const void* g) {
  u32x2 r;
  asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(r) : "v"(g) : "memory");
  return r;
}

// ST epilogue: the 16 lanes of a row group (lm) hold the same columns -> four DPP adds each, then lane lm == 0 writes its 8 columns' pairs
__device__ __forceinline__ float ws_row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
  return v;
}
__device__ __forceinline__ void ws_col_stats(float (&cs)[2][4], float (&cq)[2][4], float* __restrict__ row, int n0, int lm, int q) {
#pragma unroll
  for (int cf = 0; cf < 2; cf++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const float a = ws_row16_sum(cs[cf][r]), b = ws_row16_sum(cq[cf][r]);
      if (lm == 0) *(float2*)(row + 2 * (n0 + cf * 16 + q * 4 + r)) = make_float2(a, b);
    }
}

// VMEM bookkeeping (all waits below are by hand; loads retire in order):
//   iteration t issues, in this order: stores of tile t-1 (4), residual loads of tile t (8, asm), DMA of tile t+2 (4; wave 0: 5)
//   top of iteration t    : everything up to DMA(t) must have landed; younger loads = DMA(t+1) only          -> vmcnt(4) / vmcnt(0) at the tail
//   before the epilogue   : residual(t) must have landed; younger loads = DMA(t+2) only                      -> vmcnt(4) / vmcnt(0)
// (wave 0's fifth DMA instruction is waited for one step early; the stores are a tile old by the time a count could include them)
template <bool TR, typename T16 = bf16_t, bool ST = false>      // TR: weight fragments gathered with a stride (the data gradient reads the packed weights transposed); ST: column statistics (WsArgs::col_parts)
__global__ __launch_bounds__(256, 2) void conv3_ws_kernel(const WsArgs p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.y * 128 + wave * 32;             // this wave's 32 output channels
  const int t_begin = blockIdx.x * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block; if (t_end > p.ntiles) t_end = p.ntiles;
  if (t_begin >= t_end) return;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);      // M0 (the DMA's LDS base) must come from an SGPR

  // ---- activation tile DMA: buffer row j <-> global row 64 t - 1 + j; LDS chunk (j, c') holds source chunk c' ^ (j & 15)
  auto issue_dma = [&](int t) {
    const int buf = (t - t_begin) % WS_RING;
    const long row0 = (long)t * WS_ROWS;
    const bool has_left = (row0 % p.L) != 0, has_right = ((row0 + WS_ROWS) % p.L) != 0;
    for (int i = wave_u; i < WS_BUF_ROWS / 4; i += 4) {
      const int j = 4 * i + q, c = lm ^ (j & 15);
      const bool ok = (j >= 1 && j <= WS_ROWS) || (j == 0 && has_left) || (j == WS_ROWS + 1 && has_right);
      const char* src = ok ? (const char*)(p.x + (row0 - 1 + j) * p.ldx) + c * 16 : (const char*)p.zero_page;
      ws_dma16(src, __builtin_amdgcn_readfirstlane(lds_base + buf * WS_ABUF + i * 1024));
    }
  };
  issue_dma(t_begin);
  if (t_begin + 1 < t_end) issue_dma(t_begin + 1);

  // ---- stationary weight fragments: wf[tap][kk][cf] = W[tap][n0 + cf*16 + lm][kk*32 + q*8 .. +8]
  uint4 wf[3][4][2];
#pragma unroll
  for (int t = 0; t < 3; t++) {
    const bf16_t* wt = p.w + (long)(p.tflip ? 2 - t : t) * p.sWt;
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
#pragma unroll
      for (int cf = 0; cf < 2; cf++) {
        const bf16_t* wp = wt + (long)(n0 + cf * 16 + lm) * p.sWn + (long)(kk * 32 + q * 8) * p.sWk;
        if constexpr (!TR) wf[t][kk][cf] = *(const uint4*)wp;
        else {
          unsigned e[8];
#pragma unroll
          for (int i = 0; i < 8; i++) e[i] = wp[(long)i * p.sWk];
          wf[t][kk][cf] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
      }
  }
  float4 bv[2], ev[2];
#pragma unroll
  for (int cf = 0; cf < 2; cf++) bv[cf] = p.bias ? *(const float4*)(p.bias + n0 + cf * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  int ev_sample = -1;
  // the weight / bias loads are hipcc-visible: retire them HERE with a wait hipcc understands (vmcnt(0), other counters untouched).
  // Otherwise its scoreboard carries them into the loop and every iteration's first MFMAs sit behind vmcnt(23) ... vmcnt(0), which in
  // steady state drains the LDS-DMA prefetch instead
  __builtin_amdgcn_s_waitcnt(0x0F70);

  char* sOut = smem + WS_RING * WS_ABUF;
  auto store_tile = [&](int t) {        // whole 256-byte rows of tile t from the LDS output tile
    const long row0 = (long)t * WS_ROWS;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      const uint4 v = *(const uint4*)(sOut + row * WS_ROW_BYTES + ((ch ^ (row & 15)) * 16));
      *(uint4*)(p.y + (row0 + row) * p.ldy + blockIdx.y * 128 + ch * 8) = v;
    }
  };

  float cs[2][4], cq[2][4];      // ST: running column sums / sums of squares of this lane's rows (columns n0 + cf * 16 + q * 4 + r)
#pragma unroll
  for (int cf = 0; cf < 2; cf++)
#pragma unroll
    for (int r = 0; r < 4; r++) { cs[cf][r] = 0.f; cq[cf][r] = 0.f; }
  for (int t = t_begin; t < t_end; t++) {
    const int buf = (t - t_begin) % WS_RING;
    const long row0 = (long)t * WS_ROWS;
    if (t + 1 < t_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ws_barrier();
    if (t > t_begin) store_tile(t - 1);
    // per-sample addend (bias + time-embedding row): reloaded only when the tile enters another sample (a hipcc-visible load: its wait
    // drains the DMA queue, once per 12 tiles at L = 768)
    const int sample = (int)(row0 / p.L);
    if (sample != ev_sample) {
      ev_sample = sample;
#pragma unroll
      for (int cf = 0; cf < 2; cf++) {
        ev[cf] = bv[cf];
        if (p.rowvec) {
          const float4 e = *(const float4*)(p.rowvec + (long)sample * p.ld_rowvec + n0 + cf * 16 + q * 4);
          ev[cf].x += e.x; ev[cf].y += e.y; ev[cf].z += e.z; ev[cf].w += e.w;
        }
      }
    }
    // residual in the fragment layout: lane (lm, q) owns rows rf*16 + lm, channels n0 + cf*16 + q*4 .. +4
    u32x2 rr[4][2];
    if (p.resid) {
#pragma unroll
      for (int rf = 0; rf < 4; rf++)
#pragma unroll
        for (int cf = 0; cf < 2; cf++) rr[rf][cf] = ws_load8(p.resid + (row0 + rf * 16 + lm) * p.ldr + n0 + cf * 16 + q * 4);
    }
    const bool more = t + 2 < t_end;
    if (more) issue_dma(t + 2);

    const char* sA = smem + buf * WS_ABUF;
    f32x4 acc[4][2];
#pragma unroll
    for (int rf = 0; rf < 4; rf++)
#pragma unroll
      for (int cf = 0; cf < 2; cf++) acc[rf][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rf = 0; rf < 4; rf++)
#pragma unroll
      for (int tp = 0; tp < 3; tp++) {
        const int j = rf * 16 + lm + tp;
#pragma unroll
        for (int kk = 0; kk < 4; kk++) {
          const uint4 xf = *(const uint4*)(sA + j * WS_ROW_BYTES + (((kk * 4 + q) ^ (j & 15)) * 16));
#pragma unroll
          for (int cf = 0; cf < 2; cf++)
            if constexpr (Is16<T16>::f16) acc[rf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[tp][kk][cf]), __builtin_bit_cast(f16x8, xf), acc[rf][cf], 0, 0, 0);
            else acc[rf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[tp][kk][cf]), __builtin_bit_cast(bf16x8, xf), acc[rf][cf], 0, 0, 0);
        }
      }
    if (p.resid) {
      // residual(t) landed (DMA(t+2), issued after it, may stay in flight).  ONE statement names the registers ("+v" keeps every use
      // below it): two alternative statements in an if / else made hipcc reconcile their register assignments with v_mov copies
      // BEFORE the wait in one arm -- copies of registers whose data had not landed yet (wrong residuals in each block's last tiles)
      if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt vmcnt(4)" : "+v"(rr[0][0]), "+v"(rr[0][1]), "+v"(rr[1][0]), "+v"(rr[1][1]), "+v"(rr[2][0]), "+v"(rr[2][1]), "+v"(rr[3][0]), "+v"(rr[3][1]) :: "memory");
    }
    ws_barrier();             // every wave has read the previous output tile (store_tile above) before it is overwritten
    // ---- epilogue: one rounding to bf16 in the fragment layout, into the swizzled LDS output tile
#pragma unroll
    for (int rf = 0; rf < 4; rf++)
#pragma unroll
      for (int cf = 0; cf < 2; cf++) {
        float v0 = acc[rf][cf][0] + ev[cf].x, v1 = acc[rf][cf][1] + ev[cf].y, v2 = acc[rf][cf][2] + ev[cf].z, v3 = acc[rf][cf][3] + ev[cf].w;
        if (p.resid) {
          v0 += w16_lo<T16>(rr[rf][cf].x); v1 += w16_hi<T16>(rr[rf][cf].x);
          v2 += w16_lo<T16>(rr[rf][cf].y); v3 += w16_hi<T16>(rr[rf][cf].y);
        }
        if constexpr (ST) {
          cs[cf][0] += v0; cs[cf][1] += v1; cs[cf][2] += v2; cs[cf][3] += v3;
          cq[cf][0] = fmaf(v0, v0, cq[cf][0]); cq[cf][1] = fmaf(v1, v1, cq[cf][1]); cq[cf][2] = fmaf(v2, v2, cq[cf][2]); cq[cf][3] = fmaf(v3, v3, cq[cf][3]);
        }
        const int row = rf * 16 + lm, ch = wave * 4 + cf * 2 + (q >> 1);
        *(uint2*)(sOut + row * WS_ROW_BYTES + ((ch ^ (row & 15)) * 16) + (q & 1) * 8) = make_uint2(pack16x2<T16>(v0, v1), pack16x2<T16>(v2, v3));
      }
  }
  ws_barrier();
  store_tile(t_end - 1);
  if constexpr (ST) ws_col_stats(cs, cq, p.col_parts + (size_t)blockIdx.x * 2 * p.N, n0, lm, q);
}

// ================================================================ stride-2 Conv1d(128 -> 256, k 3, padding 1) over PAIRED rows (round 6)
// The PatchDiscriminator's third layer (config_aekl_eeg.yaml:30-40) is 9.7 GMAC against 2 x 49 MB at B = 256: HBM-bound, and the general
// implicit GEMM runs it at 2 TB/s (its 128 x 128 tiles have 12 K stages, prologue + epilogue as long as the loop).  With x'[m] = [x[2m] | x[2m + 1]]
// (two input rows of 128 channels = one row of 256) the stride disappears:
//   forward        y[m]                      = [0 | W0] x'[m - 1] + [W1 | W2] x'[m]                       (taps -1 and 0; 128 + 256 reduction elements)
//   data gradient  [dx[2m] | dx[2m + 1]]     = [W1^T | W2^T] dy[m] + [0 | W0^T] dy[m + 1]                 (taps 0 and +1; 256 + 256 for the odd half)
// Both are weight-stationary like conv3_ws_kernel: a wave keeps its 32 output columns x the NON-EMPTY (tap, 32-channel chunk) fragments in
// registers for the whole launch (12 / 8 / 16 k-steps x 2 column fragments = 96 / 64 / 128 VGPRs; read straight from the packed weight
// [3][256][128], the data gradient gathered transposed), only the activation tile streams: 32 rows x 512 B (+ 2 halo rows) by LDS-DMA into
// a 3-deep ring, 64 KB of LDS per block, two blocks per CU.
constexpr int W2_ROWS = 32, W2_BUF_ROWS = 36, W2_ROW_BYTES = 512;
constexpr int W2_ABUF = W2_BUF_ROWS * W2_ROW_BYTES;      // 18 432 B = 18 wave instructions of 2 rows
constexpr int W2_OUT = W2_ROWS * 256;                     // 32 rows x 128 output columns (one blockIdx.y half) x 2 B
constexpr int W2_LDS = WS_RING * W2_ABUF + W2_OUT;        // 63 488 B

struct Ws2Args {
  const bf16_t* x; long ldx;           // paired view: [M][256]
  const bf16_t* w;                     // packed conv weight [3][256][128]
  const float* bias;                   // forward only
  bf16_t* y; long ldy;                 // [M][256]
  int M, L, ntiles, tiles_per_block;   // rows of the paired view, rows per sample
  const void* zero_page;
  float* col_parts;                    // forward, ST kernels: as WsArgs::col_parts, rows of 2 x 256
};

// MODE 0: forward; 1 / 2: data gradient, even / odd half of the paired dx row (blockIdx.y).  SLOTS = non-empty (tap, k-step) pairs.
template <int MODE> struct W2Map;
template <> struct W2Map<0> { static constexpr int SLOTS = 12; static __device__ __forceinline__ void at(int s, int& tp, int& kk) { if (s < 4) { tp = 0; kk = 4 + s; } else { tp = 1; kk = s - 4; } } };
template <> struct W2Map<1> { static constexpr int SLOTS = 8;  static __device__ __forceinline__ void at(int s, int& tp, int& kk) { tp = 1; kk = s; } };
template <> struct W2Map<2> { static constexpr int SLOTS = 16; static __device__ __forceinline__ void at(int s, int& tp, int& kk) { tp = 1 + (s >> 3); kk = s & 7; } };

template <int MODE, typename T16, bool ST = false>
__device__ __forceinline__ void ws2_body(const Ws2Args& p, char* smem) {
  using Map = W2Map<MODE>;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lm = lane & 15, q = lane >> 4;
  const int nb0 = blockIdx.y * 128, n0 = nb0 + wave * 32;      // this block's / wave's output columns
  const int t_begin = blockIdx.x * p.tiles_per_block;
  int t_end = t_begin + p.tiles_per_block; if (t_end > p.ntiles) t_end = p.ntiles;
  if (t_begin >= t_end) return;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  // activation tile: buffer row j <-> global row 32 t - 1 + j; a DMA instruction moves 2 rows (32 lanes x 16 B each); chunk c of row j
  // lands at slot c ^ (j & 15) (32 slots per row: the XOR stays inside its 16-slot half)
  auto issue_dma = [&](int t) {
    const int buf = (t - t_begin) % WS_RING;
    const long row0 = (long)t * W2_ROWS;
    const bool has_left = (row0 % p.L) != 0, has_right = ((row0 + W2_ROWS) % p.L) != 0;
    for (int i = wave_u; i < W2_BUF_ROWS / 2; i += 4) {
      const int j = 2 * i + (lane >> 5), c = (lane & 31) ^ (j & 15);
      const bool ok = (j >= 1 && j <= W2_ROWS) || (j == 0 && has_left) || (j == W2_ROWS + 1 && has_right);
      const char* src = ok ? (const char*)(p.x + (row0 - 1 + j) * p.ldx) + c * 16 : (const char*)p.zero_page;
      ws_dma16(src, __builtin_amdgcn_readfirstlane(lds_base + buf * W2_ABUF + i * 1024));
    }
  };
  issue_dma(t_begin);
  if (t_begin + 1 < t_end) issue_dma(t_begin + 1);

  // ---- stationary weight fragments: wf[s][cf] = W'(tap, n0 + cf*16 + lm, kk*32 + q*8 .. +8) of slot s = (tap, kk)
  uint4 wf[Map::SLOTS][2];
#pragma unroll
  for (int s_ = 0; s_ < Map::SLOTS; s_++) {
    int tp, kk; Map::at(s_, tp, kk);
#pragma unroll
    for (int cf = 0; cf < 2; cf++) {
      const int n = n0 + cf * 16 + lm, k0 = kk * 32 + q * 8;
      if constexpr (MODE == 0) {
        // n = output channel; paired-row channel k0 .. +8: first half = x[2m + (tp == 1 ? 0 : *)], second half = x[2m - 1] (tp 0) / x[2m + 1] (tp 1)
        const int tap = k0 < 128 ? 1 : (tp == 0 ? 0 : 2), ci = k0 & 127;
        wf[s_][cf] = *(const uint4*)(p.w + ((long)tap * 256 + n) * 128 + ci);
      } else {
        // n = column of the paired dx row (this half: ci = n & 127), k = output channel co: W[tap][co .. co + 8][ci], gathered
        const int tap = MODE == 1 ? 1 : (tp == 1 ? 2 : 0), ci = n & 127;
        const bf16_t* wp = p.w + ((long)tap * 256 + k0) * 128 + ci;
        unsigned e[8];
#pragma unroll
        for (int i = 0; i < 8; i++) e[i] = wp[(long)i * 128];
        wf[s_][cf] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
      }
    }
  }
  float4 bv[2];
#pragma unroll
  for (int cf = 0; cf < 2; cf++) bv[cf] = (MODE == 0 && p.bias) ? *(const float4*)(p.bias + n0 + cf * 16 + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  __builtin_amdgcn_s_waitcnt(0x0F70);      // retire the compiler-visible loads here (see conv3_ws_kernel)

  char* sOut = smem + WS_RING * W2_ABUF;
  auto store_tile = [&](int t) {        // whole 256-byte row halves of tile t from the LDS output tile
    const long row0 = (long)t * W2_ROWS;
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
      const uint4 v = *(const uint4*)(sOut + row * 256 + ((ch ^ (row & 15)) * 16));
      *(uint4*)(p.y + (row0 + row) * p.ldy + nb0 + ch * 8) = v;
    }
  };
  float cs[2][4], cq[2][4];
#pragma unroll
  for (int cf = 0; cf < 2; cf++)
#pragma unroll
    for (int r = 0; r < 4; r++) { cs[cf][r] = 0.f; cq[cf][r] = 0.f; }
  for (int t = t_begin; t < t_end; t++) {
    const int buf = (t - t_begin) % WS_RING;
    // everything up to DMA(t) landed; younger: the 2 stores of tile t - 2 and DMA(t + 1) (4 instructions, 5 in waves 0 and 1)
    if (t + 1 < t_end) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ws_barrier();
    if (t > t_begin) store_tile(t - 1);
    if (t + 2 < t_end) issue_dma(t + 2);
    const char* sA = smem + buf * W2_ABUF;
    f32x4 acc[2][2];
#pragma unroll
    for (int rf = 0; rf < 2; rf++)
#pragma unroll
      for (int cf = 0; cf < 2; cf++) acc[rf][cf] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rf = 0; rf < 2; rf++)
#pragma unroll
      for (int s_ = 0; s_ < Map::SLOTS; s_++) {
        int tp, kk; Map::at(s_, tp, kk);
        const int j = rf * 16 + lm + tp;
        const uint4 xf = *(const uint4*)(sA + j * W2_ROW_BYTES + (((kk * 4 + q) ^ (j & 15)) * 16));
#pragma unroll
        for (int cf = 0; cf < 2; cf++)
          if constexpr (Is16<T16>::f16) acc[rf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, wf[s_][cf]), __builtin_bit_cast(f16x8, xf), acc[rf][cf], 0, 0, 0);
          else acc[rf][cf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[s_][cf]), __builtin_bit_cast(bf16x8, xf), acc[rf][cf], 0, 0, 0);
      }
    ws_barrier();             // every wave has read the previous output tile (store_tile above) before it is overwritten
#pragma unroll
    for (int rf = 0; rf < 2; rf++)
#pragma unroll
      for (int cf = 0; cf < 2; cf++) {
        const float v0 = acc[rf][cf][0] + bv[cf].x, v1 = acc[rf][cf][1] + bv[cf].y, v2 = acc[rf][cf][2] + bv[cf].z, v3 = acc[rf][cf][3] + bv[cf].w;
        if constexpr (ST) {
          cs[cf][0] += v0; cs[cf][1] += v1; cs[cf][2] += v2; cs[cf][3] += v3;
          cq[cf][0] = fmaf(v0, v0, cq[cf][0]); cq[cf][1] = fmaf(v1, v1, cq[cf][1]); cq[cf][2] = fmaf(v2, v2, cq[cf][2]); cq[cf][3] = fmaf(v3, v3, cq[cf][3]);
        }
        const int row = rf * 16 + lm, ch = wave * 4 + cf * 2 + (q >> 1);
        *(uint2*)(sOut + row * 256 + ((ch ^ (row & 15)) * 16) + (q & 1) * 8) = make_uint2(pack16x2<T16>(v0, v1), pack16x2<T16>(v2, v3));
      }
  }
  ws_barrier();
  store_tile(t_end - 1);
  if constexpr (ST) ws_col_stats(cs, cq, p.col_parts + (size_t)blockIdx.x * 2 * 256, n0, lm, q);
}

template <bool DGRAD, typename T16 = bf16_t, bool ST = false>
__global__ __launch_bounds__(256, 2) void conv3_ws2_kernel(const Ws2Args p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  if constexpr (!DGRAD) ws2_body<0, T16, ST>(p, smem);
  else { if (blockIdx.y == 0) ws2_body<1, T16>(p, smem); else ws2_body<2, T16>(p, smem); }
}
}  // namespace

// Y[r][n] = sum_t sum_k X[r + t - 1][k] * W(t, n, k) (+ bias[n] + rowvec[sample(r)][n] + resid[r][n]), rows flattened (sample, position), zero
// padding at sample boundaries.  Returns 1 when this kernel took the launch, 0 when the shape is not its (the caller falls back to the
// general kernel), < 0 on error.  transposed != 0: the data gradient -- X = dY, W(t, n, k) = w[2 - t][k][n] (w packed [tap][Cout][Cin]).
int conv_ws_try(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, int Cin, int Cout, int transposed, const float* bias,
                const float* rowvec, long ld_rowvec, const void* resid, long ldr, void* y, long ldy, int B, int L, float* col_parts, int* col_nparts) {
  if (col_nparts) *col_nparts = 0;
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_NO_CONV_WS") != nullptr);
  const int Kred = transposed ? Cout : Cin, N = transposed ? Cin : Cout;
  const long M = (long)B * L;
  if (off || (dtype != EEGLDM_BF16 && dtype != EEGLDM_F16) || Kred != 128 || N % 128 != 0 || L % WS_ROWS != 0 || M >= (1L << 31)) return 0;
  if (ldx % 8 != 0 || ldy % 8 != 0 || (resid && ldr % 4 != 0) || (rowvec && ld_rowvec % 4 != 0)) return 0;
  if (((size_t)x | (size_t)y | (size_t)w) % 16 != 0 || (resid && (size_t)resid % 8 != 0)) return 0;
  constexpr long min_rows = 16384;
  if (M < min_rows) return 0;                      // few tiles per block: the weight fetch is not amortised
  WsArgs a = {};
  a.x = (const bf16_t*)x; a.ldx = ldx; a.w = (const bf16_t*)w;
  if (!transposed) { a.sWt = (long)Cout * Cin; a.sWn = Cin; a.sWk = 1; a.tflip = 0; }
  else { a.sWt = (long)Cout * Cin; a.sWn = 1; a.sWk = Cin; a.tflip = 1; }
  a.bias = bias; a.rowvec = rowvec; a.ld_rowvec = ld_rowvec; a.resid = (const bf16_t*)resid; a.ldr = ldr;
  a.y = (bf16_t*)y; a.ldy = ldy; a.M = (int)M; a.L = L; a.N = N; a.ntiles = (int)(M / WS_ROWS); a.zero_page = ctx->zero_page;
  constexpr int bpc = 2;
  const int ny = N / 128;
  long nbx = (long)ctx->num_cu * bpc / ny; if (nbx < 1) nbx = 1; if (nbx > a.ntiles) nbx = a.ntiles;
  a.tiles_per_block = (int)((a.ntiles + nbx - 1) / nbx);
  nbx = (a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
  static DevOnce attr_once;      // (per device, not per process)
  if (attr_once.need(ctx->device)) {
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws_kernel<false, bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws_kernel<false, f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws_kernel<false, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws_kernel<true, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS));
  }
  ProfRec rec; const bool prof = ctx->prof_on;          // same per-class accounting as gemm_launch (bench.py roofline leg)
  if (prof) {
    rec.cls = transposed ? PROF_CONV_DGRAD : PROF_CONV_FWD; rec.flops = 2.0 * (double)M * N * Kred * 3.0;
    rec.M = (int)M; rec.N = N; rec.K = Kred; rec.taps = 3; rec.splitk = 1;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b));
    HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  // column statistics for a following BatchNorm: forward products whose partial rows (nbx x 2 N floats) fit the caller's area (16 MB)
  const bool st = col_parts && col_nparts && !transposed && (size_t)nbx * 2 * N * sizeof(float) <= (16u << 20);
  if (st) {
    a.col_parts = col_parts; *col_nparts = (int)nbx;
    if (dtype == EEGLDM_F16) hipLaunchKernelGGL((conv3_ws_kernel<false, f16_t, true>), dim3((unsigned)nbx, ny), dim3(256), WS_LDS, ctx->stream, a);
    else hipLaunchKernelGGL((conv3_ws_kernel<false, bf16_t, true>), dim3((unsigned)nbx, ny), dim3(256), WS_LDS, ctx->stream, a);
  } else if (dtype == EEGLDM_F16) {
    if (transposed) hipLaunchKernelGGL((conv3_ws_kernel<true, f16_t>), dim3((unsigned)nbx, ny), dim3(256), WS_LDS, ctx->stream, a);
    else hipLaunchKernelGGL((conv3_ws_kernel<false, f16_t>), dim3((unsigned)nbx, ny), dim3(256), WS_LDS, ctx->stream, a);
  } else if (transposed) hipLaunchKernelGGL(conv3_ws_kernel<true>, dim3((unsigned)nbx, ny), dim3(256), WS_LDS, ctx->stream, a);
  else hipLaunchKernelGGL(conv3_ws_kernel<false>, dim3((unsigned)nbx, ny), dim3(256), WS_LDS, ctx->stream, a);
  LAUNCH_CHECK();
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return 1;
}

// Stride-2 Conv1d(128 -> 256, k 3, padding 1) on contiguous NLC operands: forward (x: [B * 2 Lo][128] -> y: [B * Lo][256] + bias) or data gradient
// (x = dy: [B * Lo][256] -> y = dx: [B * 2 Lo][128]); w = the packed weight [3][256][128].  1 = launched, 0 = not this kernel's shape, < 0 = error.
int conv_ws2_try(eegldm_ctx* ctx, int dtype, int dgrad, const void* x, const void* w, const float* bias, void* y, int B, int Lo, float* col_parts, int* col_nparts) {
  if (col_nparts) *col_nparts = 0;
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_NO_CONV_WS") != nullptr);
  const long M = (long)B * Lo;
  if (off || (dtype != EEGLDM_BF16 && dtype != EEGLDM_F16) || Lo % W2_ROWS != 0 || M >= (1L << 31) || M < 8192) return 0;
  if (((size_t)x | (size_t)y | (size_t)w) % 16 != 0) return 0;
  Ws2Args a = {};
  a.x = (const bf16_t*)x; a.ldx = 256; a.w = (const bf16_t*)w; a.bias = dgrad ? nullptr : bias; a.y = (bf16_t*)y; a.ldy = 256;
  a.M = (int)M; a.L = Lo; a.ntiles = (int)(M / W2_ROWS); a.zero_page = ctx->zero_page;
  long nbx = (long)ctx->num_cu * 2 / 2; if (nbx < 1) nbx = 1; if (nbx > a.ntiles) nbx = a.ntiles;      // 2 blocks per CU, 2 column halves
  a.tiles_per_block = (int)((a.ntiles + nbx - 1) / nbx);
  nbx = (a.ntiles + a.tiles_per_block - 1) / a.tiles_per_block;
  static DevOnce attr_once;
  if (attr_once.need(ctx->device)) {
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws2_kernel<false, bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws2_kernel<false, f16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws2_kernel<false, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
    HIP_TRY(hipFuncSetAttribute((const void*)conv3_ws2_kernel<true, f16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS));
  }
  ProfRec rec; const bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = dgrad ? PROF_CONV_DGRAD : PROF_CONV_FWD; rec.flops = 2.0 * (double)M * 256 * 128 * 3.0;
    rec.M = (int)M; rec.N = 256; rec.K = 128; rec.taps = 3; rec.splitk = 1;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b)); HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  const dim3 grid((unsigned)nbx, 2);
  const bool st = col_parts && col_nparts && !dgrad && (size_t)nbx * 2 * 256 * sizeof(float) <= (16u << 20);
  if (st) {
    a.col_parts = col_parts; *col_nparts = (int)nbx;
    if (dtype == EEGLDM_F16) hipLaunchKernelGGL((conv3_ws2_kernel<false, f16_t, true>), grid, dim3(256), W2_LDS, ctx->stream, a);
    else hipLaunchKernelGGL((conv3_ws2_kernel<false, bf16_t, true>), grid, dim3(256), W2_LDS, ctx->stream, a);
  } else if (dtype == EEGLDM_F16) {
    if (dgrad) hipLaunchKernelGGL((conv3_ws2_kernel<true, f16_t>), grid, dim3(256), W2_LDS, ctx->stream, a);
    else hipLaunchKernelGGL((conv3_ws2_kernel<false, f16_t>), grid, dim3(256), W2_LDS, ctx->stream, a);
  } else if (dgrad) hipLaunchKernelGGL(conv3_ws2_kernel<true>, grid, dim3(256), W2_LDS, ctx->stream, a);
  else hipLaunchKernelGGL(conv3_ws2_kernel<false>, grid, dim3(256), W2_LDS, ctx->stream, a);
  LAUNCH_CHECK();
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return 1;
}
