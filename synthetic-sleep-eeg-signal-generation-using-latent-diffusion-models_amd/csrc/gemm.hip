// MFMA GEMM / implicit-GEMM 1-D convolution for gfx950 (wave64, 4 waves per block).
//
// One kernel template covers every dense contraction on the hot path, with all
// activations in the engine's internal "NLC" layout (rows = (sample, position),
// channels contiguous):
//   conv k3/k1 forward      Y[r][co]  = sum_t sum_ci X[row(r,t)][ci] * W[t][co][ci]      A=CONV  B=NT
//   conv dgrad              dX[r][ci] = sum_t sum_co dY[row'(r,t)][co] * W[T-1-t][co][ci] A=CONV  B=TR (tap_flip)
//   conv wgrad (per tap)    dW[t][co][ci] = sum_r dY[r][co] * X[row(r,t)][ci]            A=TR    B=TR (conv_map, split-K)
//   linear / attention      NT, NN and TN batched products
// (reference ops: /root/reference/src/models/unet.py:263,291,302,158,161,120,124,373-377)
//
// Per-wave MFMA: v_mfma_f32_16x16x32_bf16 (bf16 storage) or 4x v_mfma_f32_16x16x4_f32
// (fp32 storage: exact fp32, used for parity).  In both cases one operand fragment is
// 16 bytes per lane = one "super-chunk" of K (32 bf16 / 16 fp32).  For fp32 the k index
// inside a super-chunk is permuted identically for A and B (lane group q supplies
// k = 4q+j to MFMA step j), which is legal because K is a pure reduction index.
//
// Block tile 128 x BN (BN = 128/64/32), waves 2x2, wave tile 64 x BN/2.
// Operands are staged global -> registers -> LDS (register prefetch of the next stage
// overlaps the MFMA phase), because the conv halo, sample-boundary zero fill, tap flip
// and row maps need per-lane source addressing that LDS-DMA cannot express.
// K-contiguous operands are read with ds_read_b128; K-strided operands ("TR") with
// ds_read_b64_tr_b16 (bf16) or ds_read_b32 (fp32).  LDS images are XOR-swizzled (no padding) so
// that the b128 fragment reads, the transpose reads and the b128 staging writes are all
// bank-conflict-free for the 128-wide tiles (model + search: tools/lds_conflicts.py).
// The MFMA is issued with A and B swapped (D' = B.A^T), so every lane ends up holding FOUR
// CONSECUTIVE output channels of one row; the epilogue passes the tile through LDS once and
// writes full 16-byte (fp32) / 8-byte (bf16) vectors -- whole 256-byte rows per 32 lanes --
// with bias, timestep-embedding and residual adds done on those vectors.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {


template <typename T> struct Tr;
template <> struct Tr<float> { static constexpr int KC = 16; static constexpr int EPC = 4; };
template <> struct Tr<bf16_t> { static constexpr int KC = 32; static constexpr int EPC = 8; };
template <> struct Tr<f16_t> { static constexpr int KC = 32; static constexpr int EPC = 8; };

template <typename T> __device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x4& acc);
template <> __device__ __forceinline__ void mma<float>(const uint4& a, const uint4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<bf16_t>(const uint4& a, const uint4& b, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

template <> __device__ __forceinline__ void mma<f16_t>(const uint4& a, const uint4& b, f32x4& acc) {      // EEGLDM_F16: v_mfma_f32_16x16x32_f16
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

// ---- LDS swizzles -------------------------------------------------------------------
// NT tile: rows of KSUB*64 bytes, 16-byte slots; slot ^= f(row).
template <int KSUB> __device__ __forceinline__ int nt_swz(int row, int slot) {
  if constexpr (KSUB == 1) return slot ^ (((row >> 2) & 1) * 3);
  else return slot ^ (2 * ((row >> 1) & 3));
}
// TR bf16 tile: rows of BX*2 bytes; byte offset within the row ^= 32 * g(row) (mod row size)
template <int BX> __device__ __forceinline__ int tr_swz(int row, int colbyte) {
  // the XOR must stay inside the row: the whole row when its size is a power of two, else 128-byte windows (192-row tiles: 384 B)
  constexpr int WIN = ((BX * 2) & (BX * 2 - 1)) == 0 ? BX * 2 - 1 : 127;
  static_assert((BX * 2) % (WIN + 1) == 0, "TR rows must be whole swizzle windows");
  // ds_read_b64_tr_b16 is served in two 32-lane groups over 64 banks (MI355X_MICROARCH.md, LDS): lanes 0-15 read rows r .. r+3, lanes 16-31 rows
  // r+8 .. r+11.  Rows of 256 B or more: bit 3 of the row flips a 128-byte half and the two 16-lane groups land on different bank halves.
  // Rows of 128 B (64-wide tiles; 384-byte rows swizzle inside 128-byte windows too): the XOR is confined to (row & 3) and rows r and r+8 meet on
  // the same 32 B -- a 2-way conflict on every B-fragment read of the fused 3-tap weight gradient (SQ_LDS_BANK_CONFLICT = 1.2 cycles per LDS
  // instruction, 37 % of its LDS-array cycles, tools/r05/s18.sh).  A swizzle that lets bit 3 of the row flip the low chunk bit there
  // (rows r+8 .. r+11 take the chunks rows r .. r+3 leave free): conflicts 4.7 M -> 0, LDS-array cycles 12.6 M -> 7.9 M per launch -- and the
  // kernel is 2 % SLOWER (785-795 vs 809-815 TF/s over the UNet's shapes, LDM step +0.07 ms, tools/r05/s19.sh): the LDS array is 21-34 %
  // busy either way and is not what the loop waits for.  Measured in round 5, not adopted; the variant was removed in round 6 (HISTORY.md).
  return colbyte ^ ((((row & 3) | (((row >> 3) & 1) << 2)) * 32) & WIN);
}

// K-strided fragment read from a [k rows][x cols] tile.
template <typename T, int BX> struct TrPitch;
template <int BX> struct TrPitch<float, BX> { static constexpr int v = BX * 4 + 16; };
template <int BX> struct TrPitch<bf16_t, BX> { static constexpr int v = BX * 2; };
template <int BX> struct TrPitch<f16_t, BX> { static constexpr int v = BX * 2; };

template <int BX>
__device__ __forceinline__ uint4 read_tr_f32(const char* tile, int ks, int x0, int lm, int q, int rowoff) {
  constexpr int pitch = TrPitch<float, BX>::v;
  const char* p = tile + (ks * 16 + 4 * q + rowoff) * pitch + (x0 + lm) * 4;
  uint4 r;
  r.x = *(const unsigned*)(p);
  r.y = *(const unsigned*)(p + pitch);
  r.z = *(const unsigned*)(p + 2 * pitch);
  r.w = *(const unsigned*)(p + 3 * pitch);
  return r;
}
template <int BX>
__device__ __forceinline__ uint4 read_tr_bf16(const char* tile, int ks, int x0, int lm, int q, int rowoff) {
  // lane p of a 16-lane group supplies row (base + p/4), 4 columns at 4*(p%4); it receives
  // column p of the 4x16 block, rows base..base+3 (verified: tools/probes/layout_probe.hip).
  constexpr int pitch = TrPitch<bf16_t, BX>::v;
  const int row0 = ks * 32 + 8 * q + (lm >> 2) + rowoff, colb = (x0 + 4 * (lm & 3)) * 2;
  const char* p0 = tile + row0 * pitch + tr_swz<BX>(row0, colb);
  const char* p1 = tile + (row0 + 4) * pitch + tr_swz<BX>(row0 + 4, colb);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p1));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}
template <typename T, int BX>
__device__ __forceinline__ uint4 read_tr_t(const char* tile, int ks, int x0, int lm, int q, int rowoff = 0) {
  if constexpr (sizeof(T) == 4) return read_tr_f32<BX>(tile, ks, x0, lm, q, rowoff);
  else return read_tr_bf16<BX>(tile, ks, x0, lm, q, rowoff);
}
// byte offset of 16-byte chunk (krow, seg) of a TR tile
template <typename T, int BX>
__device__ __forceinline__ int tr_store_off(int krow, int seg) {
  if constexpr (sizeof(T) == 4) return krow * TrPitch<float, BX>::v + seg * 16;
  else return krow * TrPitch<bf16_t, BX>::v + tr_swz<BX>(krow, seg * 16);
}

template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int BN, int STRIDE, int WMT, bool DMA>
struct Cfg {
  // WMT encodes the M geometry: 2 / 4 = that many waves along M with four 16-row fragments each (128 / 256 rows);
  // 3 = two waves along M with SIX fragments each (192 rows): 25 % fewer L2->LDS bytes per MFMA than the 128-row tile at
  // the same 2 blocks per CU (the K loop is bound by L2->LDS queueing), and every UNet length (192/384/768) divides
  // 5 = FOUR waves along M with TWO fragments each (128 rows, 8 waves per block at 2 along N): the 128 x 64 x 3-tap weight-gradient tile cut into
  // 32 x 32 wave tiles -- 48 accumulator registers per lane instead of 96, so that two blocks per CU are four waves per SIMD instead of two
  // 6 = the geometry of 2 (two waves along M, four fragments each) with a THREE-deep LDS-DMA ring in the fused 3-tap weight gradient: the X
  // tile is allocated by the 9 DMA instructions it needs instead of whole rounds of 4 waves (9 KB, not 12), the three padding instructions of
  // the last round land in one shared 1 KB dump slot, and three 25 KB stages + the slot are 76 KB: two blocks per CU still fit the 160 KB LDS
  static constexpr int WM = (WMT == 3 || WMT == 6) ? 2 : (WMT == 5 ? 4 : WMT);                 // waves along M (2 along N)
  static constexpr int FM = (WMT == 3) ? 6 : (WMT == 5 ? 2 : 4);                   // 16-row fragments per wave along M
  static constexpr int BM = WM * FM * 16;                         // block rows
  // waves along N: 2, except the wide fused 3-tap weight-gradient tile (AMODE = BMODE = TR, TAPS = 3, BN = 128): FOUR, i.e. one 8-wave
  // block per CU computing 128 x 128 x 3 taps.  It is two of the 128 x 64 blocks that used to share a CU merged into one: the dY
  // tile is staged once instead of twice (33 KB instead of 49 KB per stage pair through the L2->LDS path that bounds the K loop) at
  // the same 2 waves per SIMD and 96 accumulator registers per lane.  (One 4-wave block per CU with 192 accumulators was 12 % SLOWER
  // than the 128 x 64 tile: with a single wave per SIMD nothing covers the 60-180 cycles each LDS-DMA instruction blocks its issuer.)
  static constexpr int WN = (AMODE == GA_TR && BMODE == GB_TR && TAPS == 3 && BN == 128) ? 4 : 2;
  static constexpr int BNW = BN / WN;                             // columns per wave
  static constexpr int NW = WN * WM;                              // waves per block
  static constexpr int NTHREADS = 64 * NW;
  static constexpr int KC = Tr<T>::KC;
  static constexpr int EPC = Tr<T>::EPC;
  static constexpr int KSTAGE = KSUB * KC;                       // K elements per stage
  static constexpr int SEGS = 4 * KSUB;                          // 16B chunks per NT row
  static constexpr int PITCH_NT = KSUB * 64;                     // bytes (XOR-swizzled, unpadded)
  static constexpr int A_ROWS_NT = (AMODE == GA_CONV) ? (BM * STRIDE + TAPS - 1) : BM;
  static constexpr int PITCH_A_TR = TrPitch<T, BM>::v;
  static constexpr int PITCH_B_TR = TrPitch<T, BN>::v;
  static constexpr int A_BYTES = (AMODE == GA_TR) ? KSTAGE * PITCH_A_TR : A_ROWS_NT * PITCH_NT;
  // WG3: fused 3-tap weight gradient -- dY tile staged once, X tile staged once with a 1-row halo each side,
  // the three taps are row-shifted transpose reads of the same X tile (3x less staging per MFMA than one tap per block)
  static constexpr bool WG3 = (AMODE == GA_TR && BMODE == GB_TR && TAPS == 3);
  static constexpr bool COLSUM = (AMODE == GA_TR && sizeof(T) == 2);   // fused column sums of A (GemmArgs::colsum)
  static constexpr int B_ROWS_TR = WG3 ? KSTAGE + 2 : KSTAGE;
  static constexpr int B_TILE_BYTES = (BMODE == GB_TR) ? B_ROWS_TR * PITCH_B_TR : BN * PITCH_NT;
  static constexpr int B_BYTES = (WG3 ? 1 : TAPS) * B_TILE_BYTES;
  static constexpr int NACC = WG3 ? 3 : 1;
  // LDS-DMA staging: every tile is a linear run of 16-byte chunks; one wave-instruction fills 64 of them (1 KiB)
  static constexpr int A_CHUNKS = A_BYTES / 16;
  static constexpr int B_CHUNKS = B_BYTES / 16;
  static constexpr int A_INSTR = (A_CHUNKS + 63) / 64;
  static constexpr int B_INSTR = (B_CHUNKS + 63) / 64;
  static constexpr int IA = (A_INSTR + NW - 1) / NW;              // DMA instructions per wave per stage (uniform over waves:
  static constexpr int IB = (B_INSTR + NW - 1) / NW;              //  the tail instructions fill padding from the zero page)
  static constexpr int A_ALLOC = IA * NW * 1024;
  static constexpr bool TIGHT3 = (AMODE == GA_TR && BMODE == GB_TR && TAPS == 3 && WMT == 6);
  static constexpr int B_ALLOC = TIGHT3 ? B_INSTR * 1024 : IB * NW * 1024;
  static constexpr int STAGE_BYTES = A_ALLOC + B_ALLOC;
  // Staging engine, chosen by measurement (profiles/r01_gemm_tile_sweep.txt): the 3-tap conv kernels are fastest with
  // LDS-DMA into a double-buffered ring (one barrier per stage); the 1-tap kernels (1x1 / Linear / attention / wgrad)
  // are fastest with register staging (global -> VGPR prefetch under the MFMA phase -> ds_write), single buffer.
  static constexpr bool USE_DMA = DMA;   // host picks DMA for the 3-tap conv kernels when K is a whole number of stages
  // LDS ring depth: 2 for the 3-tap conv tiles (33 KB stages, 2 blocks per CU), 3 for the 256-row variant, and 4 for the
  // short-stage (KSUB == 1) 1-tap tiles -- a 16 KB stage holds only 16 MFMAs per wave (~260 cycles), far less than the
  // ~2500-cycle DMA latency, so three stages are kept in flight
  // (the 128 x 128 fused 3-tap weight-gradient tile is one 8-wave block per CU: 3-deep ring of 36 KB stages)
  static constexpr bool WG3_WIDE = WG3 && BN == 128;
  static constexpr int NSTG = !USE_DMA ? 1 : (((WMT == 4 || WG3_WIDE || TIGHT3) && 3 * STAGE_BYTES <= 160 * 1024) ? 3 : (((TAPS == 1 || WG3) && KSUB == 1 && AMODE != GA_CONV) ? 4 : 2));
  static constexpr int CA = (A_CHUNKS + NTHREADS - 1) / NTHREADS;  // register-staged 16-byte chunks per thread
  static constexpr int CB = (B_CHUNKS + NTHREADS - 1) / NTHREADS;
  static constexpr int NSTG_BYTES_HINT = NSTG * STAGE_BYTES;
  static constexpr int FN = BNW / 16;                             // 16-wide fragments per wave along N
  static constexpr int EPI_PITCH = BN * 4 + 16;                   // fp32 epilogue tile [32][BN]
  // epilogue passes: EPI_I of the four 16-row fragment groups go through LDS per pass.  One pass of all four when the
  // staging ring already owns that much LDS (conv kernels), two otherwise -- fewer barriers and, more importantly,
  // all of a pass's bias / residual loads are independent and in flight together (the 4-pass version serialised
  // ~1 us of load latency per pass: 8-10 us per block, profiles/r01_gemm_fixed_cost.txt).
  static constexpr int EPI_I = (FM * 16 * WM * EPI_PITCH <= NSTG_BYTES_HINT) ? FM : 2;
  static constexpr int EPI_ROWS = EPI_I * 16 * WM;
  static constexpr int EPI_BYTES = EPI_ROWS * EPI_PITCH;
  static constexpr int EPI16_BYTES = (sizeof(T) == 2 && AMODE != GA_TR) ? BM * (BN * 2 + 16) : 0;   // packed bf16 output tile (one pass)
  static constexpr int DUMP_OFF = NSTG * STAGE_BYTES;             // TIGHT3: where the padding DMA instructions of the last B round land
  static constexpr int RING_BYTES = NSTG * STAGE_BYTES + (TIGHT3 ? 1024 : 0);
  static constexpr int LDS_BYTES0 = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static constexpr int LDS_BYTES = LDS_BYTES0 > EPI16_BYTES ? LDS_BYTES0 : EPI16_BYTES;
  // 16 zero bytes for conv rows of a neighbouring sample: the first padding chunk behind the A tile of buffer 0 when the
  // DMA pads that tile (it is then refilled from the zero page every stage), else a slot behind the tiles
  static constexpr bool ZPAD = DMA && AMODE == GA_CONV && (A_BYTES + 16 <= A_ALLOC);
  static constexpr int ZOFF = ZPAD ? A_BYTES : LDS_BYTES;
  static constexpr int LDS_TOTAL = LDS_BYTES + ((AMODE == GA_CONV && !ZPAD) ? 64 : 0);   // only conv kernels read the zero chunk
};

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* glb_void_ptr;

// One LDS-DMA wave-instruction: lane l's 16 source bytes -> LDS[lds_off + 16*l].  Issued as inline assembly on purpose:
// through the builtin, hipcc treats the instruction as an LDS store and puts s_waitcnt vmcnt(0) in front of EVERY later
// ds_read -- the "prefetch" of stage s+1 was waited for before the MFMAs of stage s could read their own (different)
// buffer, i.e. no DMA/MFMA overlap at all (tools/debug/stage_timing.py: 42 cycles per MFMA).  The kernel orders DMA
// against LDS reads itself: dma_wait_all() + barrier before a buffer is consumed or refilled.
__device__ __forceinline__ void dma16(const void* g, unsigned lds_off) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_off), "v"(g) : "memory", "m0");
}
__device__ __forceinline__ void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int BN, int STRIDE, int WMT, bool DMA>
__global__ __launch_bounds__((Cfg<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, DMA>::NTHREADS), (WMT == 5 ? 4 : 2))
void gemm_kernel(const GemmArgs p) {   // >= 2 waves per SIMD: <= 256 VGPR+AGPR
  using C = Cfg<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, DMA>;
  constexpr int FN = C::FN;
  constexpr int BM = C::BM, NTHREADS = C::NTHREADS, NW = C::NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, q = lane >> 4;
  const int wm = wave / C::WN, wn = wave % C::WN;
  // XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (linear id % 8), each with its own L2.  With the
  // plain (x = N tile, y = M tile) order the N tiles of one M tile -- which all read the same A rows -- land on different
  // XCDs and every L2 fetches those rows again.  When the M tile count is a multiple of 8, XCD x instead walks M tiles
  // x, x+8, ... and runs all N tiles of an M tile back to back, so A rows are fetched into one L2 once.
  int tile_n = blockIdx.x, tile_m = blockIdx.y;
  int z = blockIdx.z;
  if (p.xcd_swizzle == 1) {
    const int w = blockIdx.y * gridDim.x + blockIdx.x, xcd = w & 7, slot = w >> 3;
    tile_n = slot % (int)gridDim.x;
    tile_m = (slot / (int)gridDim.x) * 8 + xcd;
  } else if (p.xcd_swizzle == 2) {
    // split-K weight gradients: the tiles of one K split all read the same dY / X rows -> one split per XCD at a time
    const int tiles = gridDim.x * gridDim.y;
    const int w = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, xcd = w & 7, slot = w >> 3;
    const int tile = slot % tiles;
    z = (slot / tiles) * 8 + xcd;
    tile_n = tile % (int)gridDim.x; tile_m = tile / (int)gridDim.x;
  } else if (p.xcd_swizzle == 3) {
    // split-K with ANY split count (equal K chunks): the (split, tile) pairs in split-major order are cut into 8 contiguous runs, one
    // per XCD (bijective for every block count), so a split's tiles -- which all read the same dY / X rows -- sit on one XCD, two at
    // most.  Without it (split counts that are not multiples of 8: the qkv weight gradient runs 48 tiles x 10 splits) every XCD
    // fetched every row panel: the 1-tap TN class read 265 MB per launch where ~95 MB are needed (profiles/r02_pmc_hbm_traffic.json)
    const int tiles = gridDim.x * gridDim.y, nblk = tiles * (int)gridDim.z;
    const int w = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, xcd = w & 7, slot = w >> 3;
    const int q8 = nblk >> 3, r8 = nblk & 7;
    const int idx = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    // idx = (split, tap, tile) with the split slowest: the taps of a wgrad-by-tap launch (stride-2 convs) read the same dY rows and
    // overlapping X rows as well
    const int per_split = tiles * p.ztaps;
    const int spb = idx / per_split, rem = idx - spb * per_split;         // (problem, split) pair, the problem slowest: grouped launches keep
    const int tap = rem / tiles, tile = rem - tap * tiles;               // one problem's splits on neighbouring XCD runs
    const int pb = spb / p.splitk, sp = spb - pb * p.splitk;
    z = sp + p.splitk * (tap + p.ztaps * pb);
    tile_n = tile % (int)gridDim.x; tile_m = tile / (int)gridDim.x;
  }
  const int n0 = tile_n * BN;
  const int m0 = tile_m * BM;
  const int ksplit = z % p.splitk; z /= p.splitk;
  const int tz = z % p.ztaps;      z /= p.ztaps;
  const int bz = z;

  const T* __restrict__ Ag = p.ngroup ? (const T*)p.grp->A[bz] : (const T*)p.A + (long)bz * p.sAb;
  const T* __restrict__ Bg = p.ngroup ? (const T*)p.grp->B[bz] : (const T*)p.B + (long)bz * p.sBb;
  float* const colsum_g = p.ngroup ? p.grp->CS[bz] : p.colsum;
  const long lda_g = p.ngroup ? p.grp->lda[bz] : p.lda, ldb_g = p.ngroup ? p.grp->ldb[bz] : p.ldb;

  // K range of this block
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    if (p.k_skew > 0.f) {
      // Skewed split: with equal chunks every block of the single round finishes at the same moment and the chip then waits for
      // all their fp32 atomics to drain (24.6 k per block, ~370 G/s chip-wide: 34 of 105 us for 512x512x49152).  Chunk lengths
      // that grow linearly with the split index spread the finish times, so early blocks' atomics drain under the others' K loops.
      const int S = (p.K + C::KSTAGE - 1) / C::KSTAGE;
      const float sk = p.k_skew, inv = 1.0f / (float)p.splitk;
      auto bnd = [&](int i) { const float f = (float)i * inv; return i >= p.splitk ? S : (int)((float)S * ((1.0f - sk) * f + sk * f * f) + 0.5f); };
      kbeg = bnd(ksplit) * C::KSTAGE;
      kend = min(p.K, bnd(ksplit + 1) * C::KSTAGE);
    } else {
      int per = (p.K + p.splitk - 1) / p.splitk;
      per = (per + C::KSTAGE - 1) / C::KSTAGE * C::KSTAGE;
      kbeg = ksplit * per;
      kend = min(p.K, kbeg + per);
    }
    if (kbeg >= kend) return;
  }
  const int nstages = (kend - kbeg + C::KSTAGE - 1) / C::KSTAGE;

  const uint4 zero4 = make_uint4(0, 0, 0, 0);
  if constexpr (AMODE == GA_CONV && !C::ZPAD) { if (tid < 4) *(uint4*)(smem + C::LDS_BYTES + tid * 16) = zero4; }   // zero chunk behind the tiles (conv rows of other samples); visible after the first barrier
  const T* __restrict__ zeros = (const T*)p.zero_page;

  // ---- staging: global -> LDS by DMA (global_load_lds_dwordx4).  LDS chunk c of a tile receives the
  // 16 source bytes its swizzled position stands for; out-of-range chunks read the zero page.
  // (branch-free on purpose: a per-chunk branch makes hipcc wait for each register-staged load separately)
  auto a_dec = [&](int c, int k0, long& off) __attribute__((always_inline)) -> bool {
    bool ok = c < C::A_CHUNKS;
    off = 0;
    if constexpr (AMODE == GA_PLAIN) {
      const int row = c / C::SEGS, seg = nt_swz<KSUB>(row, c % C::SEGS);
      const int m = m0 + row, k = k0 + seg * C::EPC;
      ok = ok && m < p.M && k < kend;
      off = (long)m * lda_g + k;
    } else if constexpr (AMODE == GA_CONV) {
      const int row = c / C::SEGS, seg = nt_swz<KSUB>(row, c % C::SEGS);
      long fr = (long)m0 * STRIDE - p.pad_l + row;   // flattened virtual input row
      const int k = k0 + seg * C::EPC;
      ok = ok && fr >= 0 && fr < (long)p.M * STRIDE && k < kend;
      if (p.ups != 1) {                                // wave-uniform
        const long b = fr / p.Lin; const int vv = (int)(fr - b * p.Lin);
        ok = ok && (vv % p.ups) == 0 && (vv / p.ups) < p.Lsrc;
        fr = b * p.Lsrc + vv / p.ups;
      }
      off = fr * lda_g + k;
    } else {  // GA_TR: source [K][M], M contiguous
      constexpr int RCP = C::PITCH_A_TR / 16;
      const int krow = c / RCP, cs = c % RCP;
      int seg = cs;
      if constexpr (sizeof(T) == 4) ok = ok && cs < BM / 4;
      else seg = tr_swz<BM>(krow, cs * 16) >> 4;
      const int k = k0 + krow, m = m0 + seg * C::EPC;
      ok = ok && k < kend && m < p.M;
      off = (long)k * lda_g + m;
    }
    return ok;
  };
  auto b_dec = [&](int c, int k0, long& off) __attribute__((always_inline)) -> bool {
    bool ok = c < C::B_CHUNKS;
    off = 0;
    if constexpr (BMODE == GB_NT) {
      const int tap = c / (BN * C::SEGS), r = c % (BN * C::SEGS);
      const int row = r / C::SEGS, seg = nt_swz<KSUB>(row, r % C::SEGS);
      const int n = n0 + row, k = k0 + seg * C::EPC;
      const int tw = p.tap_flip ? (TAPS - 1 - tap) : tap;
      ok = ok && n < p.N && k < kend;
      if (p.b_kblk) off = (long)tw * p.sBt + ((long)(k / C::KC) * p.N + n) * C::KC + (k % C::KC);   // [tap][K / KC][N][KC]
      else off = (long)tw * p.sBt + (long)n * ldb_g + k;
    } else {  // GB_TR: source [K][N], N contiguous
      constexpr int RCP = C::PITCH_B_TR / 16;
      const int tap = c / (C::B_ROWS_TR * RCP), r = c % (C::B_ROWS_TR * RCP);
      const int krow = r / RCP, cs = r % RCP;
      int seg = cs;
      if constexpr (sizeof(T) == 4) ok = ok && cs < BN / 4;
      else seg = tr_swz<BN>(krow, cs * 16) >> 4;
      const int n = n0 + seg * C::EPC;
      if constexpr (C::WG3) {
        // tile row krow <-> input row k0 + krow - 1 (k=3, stride 1, pad 1); rows of another sample read as zero, which is
        // exactly the conv's zero padding because a K chunk never straddles a sample (Lout % KSTAGE == 0, checked on the host)
        const int xr = k0 + krow - 1;
        ok = ok && xr >= 0 && xr < p.K && (xr / p.Lout) == (k0 / p.Lout) && n < p.N;
        off = (long)xr * ldb_g + n;
        return ok;
      }
      const int k = k0 + krow;
      ok = ok && k < kend && n < p.N;
      if (p.conv_map) {   // wgrad (wave-uniform): K index = output row -> input row of tap tz
        const int bs = k / p.Lout, lo = k - bs * p.Lout;
        const int vv = lo * p.stride + tz - p.pad_l;
        ok = ok && vv >= 0 && vv < p.Lin;
        off = ((long)bs * p.Lin + vv) * ldb_g + n;
      } else {
        const int tw = p.tap_flip ? (TAPS - 1 - tap) : tap;
        off = (long)tw * p.sBt + (long)k * ldb_g + n;
      }
    }
    return ok;
  };
  // DMA kernels are only launched when K is a whole number of stages: the source of each chunk then moves by a constant
  // per stage, so the decode above runs ONCE per block and the K loop only adds a stride to a saved pointer
  // (decoding per stage cost ~7 VALU instructions per MFMA and a very branchy loop body).
  const T* apre[C::USE_DMA ? C::IA : 1]; const T* bpre[C::USE_DMA ? C::IB : 1];
  long astep = 0, bstep = 0;     // elements per stage
  unsigned btop = 0, bbot = 0;   // WG3: bit i set -> this lane's chunk of B instruction i lies in the top / bottom halo row
  if constexpr (C::USE_DMA) {
    astep = (AMODE == GA_TR) ? (long)C::KSTAGE * lda_g : (long)C::KSTAGE;   // conv / plain A is K-contiguous, TR A is K-strided
    bstep = (BMODE == GB_NT) ? (p.b_kblk ? (long)KSUB * p.N * C::KC : (long)C::KSTAGE) : (long)C::KSTAGE * ldb_g;
#pragma unroll
    for (int i = 0; i < C::IA; i++) { long off; const bool ok = a_dec((wave + NW * i) * 64 + lane, kbeg, off); apre[i] = ok ? Ag + off : nullptr; }
#pragma unroll
    for (int i = 0; i < C::IB; i++) {
      const int c = (wave + NW * i) * 64 + lane;
      if constexpr (C::WG3) {
        // halo rows are valid or zero depending on the stage (sample boundary), so only the static part is decoded here
        constexpr int RCP = C::PITCH_B_TR / 16;
        const int krow = c / RCP, cs = c % RCP;
        int seg = cs; bool ok = c < C::B_CHUNKS;
        if constexpr (sizeof(T) == 4) ok = ok && cs < BN / 4;
        else seg = tr_swz<BN>(krow, cs * 16) >> 4;
        const int n = n0 + seg * C::EPC;
        ok = ok && n < p.N;
        if (krow == 0) btop |= 1u << i;
        if (krow == C::KSTAGE + 1) bbot |= 1u << i;
        bpre[i] = ok ? Bg + ((long)(kbeg + krow - 1) * ldb_g + n) : nullptr;
      } else {
        long off; const bool ok = b_dec(c, kbeg, off); bpre[i] = ok ? Bg + off : nullptr;
      }
    }
  }
  const unsigned lds0 = (unsigned)(size_t)(lds_void_ptr)smem;           // LDS byte address of the tile area
  const unsigned wave_u = __builtin_amdgcn_readfirstlane(wave);   // scalar copy (M0 must come from an SGPR)
  auto b_dst = [&](int buf, int i) __attribute__((always_inline)) -> unsigned {      // LDS byte offset of this wave's B instruction i (wave-uniform)
    const unsigned idx = wave_u + NW * i;
    if constexpr (C::TIGHT3) { if (idx >= (unsigned)C::B_INSTR) return (unsigned)C::DUMP_OFF; }
    return (unsigned)(buf * C::STAGE_BYTES + C::A_ALLOC) + idx * 1024u;
  };
  auto issue_stage = [&](int s, int buf) __attribute__((always_inline)) {
    unsigned bdead = 0;
    if constexpr (C::WG3) {   // k=3, pad 1: the row above the first / below the last row of a sample is the conv's zero padding
      const int k0 = kbeg + s * C::KSTAGE;
      if (k0 % p.Lout == 0) bdead |= btop;
      if ((k0 + C::KSTAGE) % p.Lout == 0) bdead |= bbot;
    }
#pragma unroll
    for (int i = 0; i < C::IA; i++) {
      const T* src = apre[i] ? apre[i] + s * astep : zeros;
      dma16(src, lds0 + buf * C::STAGE_BYTES + (wave_u + NW * i) * 1024);
    }
#pragma unroll
    for (int i = 0; i < C::IB; i++) {
      const T* src = (bpre[i] && !((bdead >> i) & 1)) ? bpre[i] + s * bstep : zeros;
      dma16(src, lds0 + b_dst(buf, i));
    }
  };

  // one DMA instruction of a stage (piece k < IA: A, else B): the double-buffered kernels issue the next stage's pieces BETWEEN the
  // MFMAs of the current stage instead of as a burst after the barrier.  An LDS-DMA instruction costs the issuing wave 100-185
  // cycles when nothing else is in flight and ~60 among MFMAs (MI355X_MICROARCH.md): the burst of 9 pieces was most of the
  // ~1200-cycle "wait" half of every stage (stage stamps: wait 1212 + MFMA phase 1212 per stage), with the MFMA pipe idle.
  auto issue_piece = [&](int s, int buf, int k) __attribute__((always_inline)) {      // k is a constant after unrolling
    if (k < C::IA) {
      const T* src = apre[k] ? apre[k] + s * astep : zeros;
      dma16(src, lds0 + buf * C::STAGE_BYTES + (wave_u + NW * k) * 1024);
    } else {
      const int i = k - C::IA;
      bool dead = false;
      if constexpr (C::WG3) {
        const int k0 = kbeg + s * C::KSTAGE;
        dead = ((btop >> i) & 1) && (k0 % p.Lout == 0);
        dead = dead || (((bbot >> i) & 1) && ((k0 + C::KSTAGE) % p.Lout == 0));
      }
      const T* src = (bpre[i] && !dead) ? bpre[i] + s * bstep : zeros;
      dma16(src, lds0 + b_dst(buf, i));
    }
  };

  // ---- register staging (1-tap kernels): chunk c of the linear LDS image <- 16 bytes (predicated load, zeros otherwise)
  uint4 ra[C::USE_DMA ? 1 : C::CA], rb[C::USE_DMA ? 1 : C::CB];
  auto load_stage = [&](int s) __attribute__((always_inline)) {
    const int k0 = kbeg + s * C::KSTAGE;
#pragma unroll
    for (int i = 0; i < C::CA; i++) {
      long off; uint4 v = zero4;
      if (a_dec(tid + i * NTHREADS, k0, off)) v = *(const uint4*)(Ag + off);
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < C::CB; i++) {
      long off; uint4 v = zero4;
      if (b_dec(tid + i * NTHREADS, k0, off)) v = *(const uint4*)(Bg + off);
      rb[i] = v;
    }
  };
  auto store_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C::CA; i++) { const int c = tid + i * NTHREADS; if (c < C::A_CHUNKS) *(uint4*)(smem + c * 16) = ra[i]; }
#pragma unroll
    for (int i = 0; i < C::CB; i++) { const int c = tid + i * NTHREADS; if (c < C::B_CHUNKS) *(uint4*)(smem + C::A_ALLOC + c * 16) = rb[i]; }
  };

  // ---- per-lane conv validity masks -------------------------------------------------
  // bit (i*TAPS + t) set -> A fragment i, tap t reads a row of ANOTHER sample: use zeros.
  unsigned zmask = 0;
  if constexpr (AMODE == GA_CONV) {
#pragma unroll
    for (int i = 0; i < C::FM; i++) {
      const int m = m0 + wm * (C::FM * 16) + i * 16 + lm;
      const int lo = m % p.Lout;
#pragma unroll
      for (int t = 0; t < TAPS; t++) {
        const int vv = lo * STRIDE + t - p.pad_l;
        if (vv < 0 || vv >= p.Lin) zmask |= 1u << (i * TAPS + t);
      }
    }
  }

  f32x4 acc[C::NACC][C::FM][FN];
#pragma unroll
  for (int a = 0; a < C::NACC; a++)
#pragma unroll
    for (int i = 0; i < C::FM; i++)
#pragma unroll
      for (int j = 0; j < FN; j++) acc[a][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fused bias gradient (GemmArgs::colsum): the first N tile's wn == 0 waves (tap slice 0) also multiply their A fragments by ones
  f32x4 cs[C::COLSUM ? C::FM : 1];
  bool do_cs = false;
  if constexpr (C::COLSUM) {
    do_cs = colsum_g != nullptr && tile_n == 0 && wn == 0 && tz == 0 && (bz == 0 || p.ngroup);
#pragma unroll
    for (int i = 0; i < C::FM; i++) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  // bias chunk of this thread's epilogue columns: fetched now so its latency hides behind the whole K loop
  float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr bool EPI16 = sizeof(T) == 2 && AMODE != GA_TR;      // packed-bf16 epilogue available (see the epilogue)
  const bool use16 = EPI16 && !p.out_f32 && !p.atomic_out && p.N % 8 == 0 && p.ldc % 8 == 0 && (!p.rowvec || p.rows_per_vec >= BM);
  if (p.bias && !use16) { const int nb = n0 + (tid % (BN / 4)) * 4; bias4 = *(const float4*)(p.bias + (nb < p.N ? nb : 0)); }
  // addends of the packed-bf16 epilogue in the fragment layout: bias now, residual / embedding rows during the LAST K stage
  // (their ~1-2 us latency hides behind that stage's MFMAs instead of opening the epilogue)
  float4 pf_bias[EPI16 ? FN : 1];
  uint2 pf_res[EPI16 ? C::FM : 1][EPI16 ? FN : 1];
  if constexpr (EPI16) {
    if (use16) {
#pragma unroll
      for (int j = 0; j < FN; j++) {
        const int nj = n0 + wn * C::BNW + j * 16 + q * 4;
        pf_bias[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.bias) pf_bias[j] = *(const float4*)(p.bias + (nj < p.N ? nj : 0));
      }
    }
  }
  auto prefetch_epi = [&](const bool in_epilogue) __attribute__((always_inline)) {
    if constexpr (EPI16) {
      if (!use16) return;
      if (p.resid && (C::FM == 4 || in_epilogue)) {   // six-fragment tiles have no registers to spare during the K loop
#pragma unroll
        for (int i = 0; i < C::FM; i++) {
          const int mr = m0 + wm * (C::FM * 16) + i * 16 + lm;
          const T* rp = (const T*)p.resid + (long)(mr < p.M ? mr : m0) * p.ldr;
#pragma unroll
          for (int j = 0; j < FN; j++) { const int nj = n0 + wn * C::BNW + j * 16 + q * 4; pf_res[i][j] = *(const uint2*)(rp + (nj < p.N ? nj : 0)); }
        }
      }
    }
  };

  // ring of NSTG stage buffers, loads run NSTG-1 stages ahead; each wave issues exactly IA+IB DMA instructions per
  // stage, so "stage s has landed" == at most (NSTG-2)*(IA+IB) of this wave's DMAs still outstanding.
  constexpr int PER = C::IA + C::IB;
  constexpr bool INTERLEAVE = C::USE_DMA && (C::NSTG == 2 || C::NSTG == 3);
  constexpr int NMFMA_STAGE = TAPS * KSUB * C::FM * C::FN;                                  // MFMAs per wave and stage
  constexpr int DMA_FIRST = 1;
  constexpr int DMA_EVERY = (2 * NMFMA_STAGE / 3) / PER > 1 ? (2 * NMFMA_STAGE / 3) / PER : 1;      // the stage's DMA instructions spread over the first two thirds of its MFMAs
  static_assert(!INTERLEAVE || DMA_FIRST + (PER - 1) * DMA_EVERY < NMFMA_STAGE, "every DMA piece needs an MFMA slot");
  if constexpr (C::USE_DMA) {
    issue_stage(0, 0);
#pragma unroll
    for (int k = 1; k < C::NSTG - 1; k++) if (k < nstages) issue_stage(k, k);
  } else {
    load_stage(0);
  }
#ifdef EEG_STAGE_TIMING
  unsigned long long* tlog = (unsigned long long*)p.zero_page + 512 + ((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 64;   // debug only
  int tl = 0;
// slots 0..61: shader-cycle stamps (per-XCD counters); slot 62 / 63: first / latest 100 MHz wall clock (comparable across the chip)
#define TSTAMP() do { if (tid == 0 && tl < 62) { tlog[tl++] = __builtin_readcyclecounter(); const unsigned long long wc_ = wall_clock64(); if (tl == 1) tlog[62] = wc_; tlog[63] = wc_; } } while (0)
#else
#define TSTAMP() do {} while (0)
#endif
  TSTAMP();
  // one K stage; the last one is a separate copy of the body (peeled) so that the epilogue prefetch registers are not
  // live -- and not spilled -- across the whole loop
  auto run_stage = [&](const int s, const bool last) __attribute__((always_inline)) {
    if constexpr (!C::USE_DMA) {
      store_stage();
      __syncthreads();
      if (s + 1 < nstages) load_stage(s + 1);
    } else if constexpr (C::NSTG > 2) {
      // stage s has landed once at most min(NSTG-2, stages issued after s) * PER of this wave's DMAs are outstanding
      const int ahead = min(nstages - 1 - s, C::NSTG - 2);
      if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      else dma_wait_all();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if constexpr (!INTERLEAVE) { if (s + C::NSTG - 1 < nstages) issue_stage(s + C::NSTG - 1, (s + C::NSTG - 1) % C::NSTG); }
    } else {
      // vmcnt(0) + barrier: stage s has landed in buffer s&1, and every wave is done reading buffer (s+1)&1
      dma_wait_all();
#ifdef EEG_STAGE_SPLIT      // (developer build: DMA wait and barrier stamped separately, tools/debug/stage_split.py)
      TSTAMP();
#endif
      __syncthreads();
      if constexpr (!INTERLEAVE) { if (s + 1 < nstages) issue_stage(s + 1, (s + 1) & 1); }
    }
    TSTAMP();   // after wait+barrier(+issue of the next stage)
    if (last) prefetch_epi(false);
    const char* smA = smem + (s % C::NSTG) * C::STAGE_BYTES;
    const char* smB = smA + C::A_ALLOC;
    // Fragment loads run one (tap, k-sub) step ahead of the MFMAs that consume them: with one or two waves per SIMD the
    // ~200-cycle ds_read latency was fully exposed three times per stage (MFMA phase 2200 cycles for 770 cycles of MFMA work,
    // tools/debug/stage_timing.py).  Rows of another sample are read from a 16-byte zero chunk in LDS (address select)
    // instead of being cleared after the load.
    constexpr int NSTEP = TAPS * KSUB;
    // AREUSE (fused 3-tap weight gradient): the dY fragments do not depend on the tap -- the steps run k-sub-major and the A fragments
    // are read once per k-sub instead of once per (tap, k-sub): 40 instead of 72 LDS reads per wave and stage (+10 %, 750 -> 826 TF/s
    // over the UNet's shapes; the K loop is sensitive to LDS instructions)
    constexpr bool AREUSE = C::WG3;
    auto load_frags = [&](int st, uint4 (&af)[C::FM], uint4 (&bf)[FN]) __attribute__((always_inline)) {
      const int t = AREUSE ? st % TAPS : st / KSUB, ks = AREUSE ? st / TAPS : st % KSUB;
#pragma unroll
      for (int i = 0; i < C::FM; i++) {
        if (AREUSE && t != 0) break;
        if constexpr (AMODE == GA_TR) {
          af[i] = read_tr_t<T, BM>(smA, ks, wm * (C::FM * 16) + i * 16, lm, q);
        } else {
          int row = wm * (C::FM * 16) + i * 16 + lm;
          if constexpr (AMODE == GA_CONV) row = row * STRIDE + t;
          const char* ap = smA + row * C::PITCH_NT + nt_swz<KSUB>(row, ks * 4 + q) * 16;
          if constexpr (AMODE == GA_CONV) { if (zmask & (1u << (i * TAPS + t))) ap = smem + C::ZOFF; }
          af[i] = *(const uint4*)ap;
        }
      }
#pragma unroll
      for (int j = 0; j < FN; j++) {
        if constexpr (BMODE == GB_TR) {
          bf[j] = read_tr_t<T, BN>(smB + (C::WG3 ? 0 : t) * C::B_TILE_BYTES, ks, wn * C::BNW + j * 16, lm, q, C::WG3 ? t : 0);
        } else {
          const int row = wn * C::BNW + j * 16 + lm;
          bf[j] = *(const uint4*)(smB + t * C::B_TILE_BYTES + row * C::PITCH_NT + nt_swz<KSUB>(row, ks * 4 + q) * 16);
        }
      }
    };
    {
    uint4 af[2][C::FM], bf[2][FN];
    load_frags(0, af[0], bf[0]);
    __builtin_amdgcn_sched_barrier(0);
#ifdef EEG_STAGE_SPLIT
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); TSTAMP();      // the first fragments of the stage have arrived
    __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
    for (int st = 0; st < NSTEP; st++) {
      if (st + 1 < NSTEP) load_frags(st + 1, af[AREUSE ? ((st + 1) / TAPS) & 1 : (st + 1) & 1], bf[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);   // keep the prefetch above this step's MFMAs (hipcc otherwise sinks each read to just before its use)
      const int t = AREUSE ? st % TAPS : st / KSUB;
      const int ai = AREUSE ? (st / TAPS) & 1 : st & 1;
#pragma unroll
      for (int i = 0; i < C::FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++) {
#ifdef EEG_DBG_NO_MFMA
          asm volatile("" :: "v"(af[ai][i].x), "v"(bf[st & 1][j].x));
#else
          if constexpr (AMODE == GA_TR) mma<T>(af[ai][i], bf[st & 1][j], acc[C::WG3 ? t : 0][i][j]);   // TN products keep the natural fragment (atomic epilogue)
          else mma<T>(bf[st & 1][j], af[ai][i], acc[0][i][j]);                                        // swapped: acc = (B.A^T) fragment
#endif
          if constexpr (INTERLEAVE) {
            // piece number (m - DMA_FIRST) / DMA_EVERY goes out after MFMA m of the stage: all pieces are issued in the first part
            // of the MFMA sequence so that they have the rest of it (and the other block's phase) to land
            constexpr int MPS = C::FM * FN;                    // MFMAs per step
            const int m = st * MPS + i * FN + j;               // compile-time after unrolling
#ifndef EEG_DBG_NO_LOOP_DMA   // (developer builds: K loop without its loads / without its MFMAs, tools/debug/stage_timing.py)
            if (!last && m >= DMA_FIRST && (m - DMA_FIRST) % DMA_EVERY == 0 && (m - DMA_FIRST) / DMA_EVERY < PER) {
              __builtin_amdgcn_sched_barrier(0);
              if (C::NSTG == 2 || s + C::NSTG - 1 < nstages)      // deeper rings: the last NSTG-2 non-final stages have nothing left to fetch (wave-uniform)
                issue_piece(s + C::NSTG - 1, (s + C::NSTG - 1) % C::NSTG, (m - DMA_FIRST) / DMA_EVERY);
              __builtin_amdgcn_sched_barrier(0);
            }
#endif
          }
        }
    }
    }
    if constexpr (C::COLSUM) {
      // bias gradient for free: A (dY, transposed) times an all-ones B fragment = the row sums of this stage in every column.
      // A separate pass over the stage's A tile (one wave in 16 takes it) -- a branch inside the unrolled MFMA steps cost every
      // wave 8 % (512 x 512 x 49152: 1.03 -> 1.11 ms).
      if (do_cs) {
        constexpr unsigned ONE2 = Is16<T>::f16 ? 0x3C003C00u : 0x3F803F80u;      // 1.0 | 1.0 in the operand format
        const uint4 ones = make_uint4(ONE2, ONE2, ONE2, ONE2);
#pragma unroll
        for (int ks = 0; ks < KSUB; ks++)
#pragma unroll
          for (int i = 0; i < C::FM; i++) {
            const uint4 a = read_tr_t<T, BM>(smA, ks, wm * (C::FM * 16) + i * 16, lm, q);
            mma<T>(a, ones, cs[i]);
          }
      }
    }
    TSTAMP();   // after the MFMA phase of stage s
    if constexpr (!C::USE_DMA) __syncthreads();   // single buffer: reads of stage s done before stage s+1 is written
  };
  for (int s = 0; s + 1 < nstages; s++) run_stage(s, false);
  run_stage(nstages - 1, true);
  if constexpr (C::USE_DMA) __syncthreads();       // all waves done with the staging buffers before the epilogue tile reuses them

  TSTAMP();   // start of epilogue
  // ---- epilogue ---------------------------------------------------------------------
  // acc[i][j][r] = C[m = wm*64 + i*16 + lm][n = wn*(BN/2) + j*16 + q*4 + r] (operands were swapped).
  // 4/EPI_I passes through an fp32 LDS tile, then 4-wide vector read-modify-store.
  char* Cb = (char*)p.C;
  const long cbase = (long)bz * p.sCb + (long)tz * p.sCt + (long)ksplit * p.sCk;
  constexpr int CH = BN / 4;                      // 4-element chunks per row
  constexpr int NCH = C::EPI_ROWS * CH;           // chunks per pass
  if constexpr (C::COLSUM) {
    if (do_cs && lm == 0) {   // every column of cs holds the row sums: column 0's lanes add them (one atomic per row and K split)
#pragma unroll
      for (int i = 0; i < C::FM; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int m = m0 + wm * (C::FM * 16) + i * 16 + q * 4 + r;
          if (m < p.M) atomicAdd(colsum_g + m, cs[i][r]);
        }
    }
  }
  if constexpr (AMODE == GA_TR) {
    if (p.atomic_out) {
      // split-K weight gradients: natural fragment layout (rows q*4+r, col lm), fp32 atomics straight from registers.
      // All splits of a tile add into the same lines; each split starts a quarter of the way further round the tile so
      // that concurrently finishing blocks are spread over different lines / L2 channels instead of queueing on one.
      auto emit = [&](auto ROT) {
        constexpr int NAI = C::NACC * C::FM;
#pragma unroll
        for (int ai0 = 0; ai0 < NAI; ai0++) {
          constexpr int R = decltype(ROT)::value;
          const int ai = (ai0 + R * C::NACC) % NAI;
          const int a = ai / C::FM, i = ai % C::FM;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int m = m0 + wm * (C::FM * 16) + i * 16 + q * 4 + r;
            if (m >= p.M) continue;
#pragma unroll
            for (int j = 0; j < FN; j++) {
              const int n = n0 + wn * C::BNW + j * 16 + lm;
              if (n < p.N) atomicAdd((float*)Cb + cbase + (long)a * p.sCt + (long)m * p.ldc + n, acc[a][i][j][r] * p.alpha);
            }
          }
        }
      };
      switch (ksplit & 3) {
        case 0: emit(std::integral_constant<int, 0>{}); break;
        case 1: emit(std::integral_constant<int, 1>{}); break;
        case 2: emit(std::integral_constant<int, 2>{}); break;
        default: emit(std::integral_constant<int, 3>{}); break;
      }
      TSTAMP();   // atomics issued
#ifdef EEG_STAGE_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TSTAMP();   // atomics retired
#endif
      return;
    }
    if constexpr (C::WG3) {
      if (p.sCk != 0) {
        // split-K partial tile of the fused 3-tap weight gradient -> this split's slot of the workspace, PLAIN stores straight from
        // the accumulators (natural fragment layout: a wave instruction writes 4 rows x 64 contiguous bytes).  Measured with
        // tools/probes/atomic_probe.hip on the same access pattern: 512 blocks x 24576 values take 43 us as fp32 atomics
        // (290 G adds/s, the same at agent or workgroup scope, shared or per-XCD target) and 11.6 us as stores (4.3 TB/s); the fold
        // kernel then streams the partials once.
#pragma unroll
        for (int a = 0; a < C::NACC; a++)
#pragma unroll
          for (int i = 0; i < C::FM; i++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
              const int m = m0 + wm * (C::FM * 16) + i * 16 + q * 4 + r;
              if (m >= p.M) continue;
#pragma unroll
              for (int j = 0; j < FN; j++) {
                const int n = n0 + wn * C::BNW + j * 16 + lm;
                if (n < p.N) *((float*)Cb + cbase + (long)a * p.sCt + (long)m * p.ldc + n) = acc[a][i][j][r] * p.alpha;
              }
            }
        TSTAMP();
#ifdef EEG_STAGE_TIMING
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TSTAMP();
#endif
        return;
      }
    }
  }
  if constexpr (EPI16) {
    if (use16) {
      // bf16 output: bias / embedding / residual are added in the fragment layout (every lane owns 4 consecutive columns of
      // one row: acc[0][i][j][r] = C[wm*64 + i*16 + lm][wn*(BN/2) + j*16 + q*4 + r]), the result is packed to bf16 FIRST and
      // only then transposed through LDS: half the LDS bytes of the fp32 tile, one pass, and 16-byte global stores.
      constexpr int PITCH16 = BN * 2 + 16;
      static_assert(BM * PITCH16 <= C::LDS_BYTES, "bf16 epilogue tile must fit the staging area");
      if constexpr (C::FM != 4) prefetch_epi(true);
      float4 pf_rv[2][FN];
      int pf_bnd = 0;
      if (p.rowvec) {
        // a tile of BM rows touches at most two samples (rows_per_vec >= BM is part of use16)
        const int s0 = m0 / p.rows_per_vec;
        pf_bnd = (s0 + 1) * p.rows_per_vec;
        const int s1 = min(s0 + 1, (p.M - 1) / p.rows_per_vec);
#pragma unroll
        for (int j = 0; j < FN; j++) {
          const int nj = n0 + wn * C::BNW + j * 16 + q * 4, nc = nj < p.N ? nj : 0;
          pf_rv[0][j] = *(const float4*)(p.rowvec + (long)s0 * p.ld_rowvec + nc);
          pf_rv[1][j] = *(const float4*)(p.rowvec + (long)s1 * p.ld_rowvec + nc);
        }
      }
#pragma unroll
      for (int i = 0; i < C::FM; i++) {
        const bool hi = p.rowvec && (m0 + wm * (C::FM * 16) + i * 16 + lm) >= pf_bnd;
#pragma unroll
        for (int j = 0; j < FN; j++) {
          float4 a = pf_bias[j];
          if (p.rowvec) {
            a.x += hi ? pf_rv[1][j].x : pf_rv[0][j].x; a.y += hi ? pf_rv[1][j].y : pf_rv[0][j].y;
            a.z += hi ? pf_rv[1][j].z : pf_rv[0][j].z; a.w += hi ? pf_rv[1][j].w : pf_rv[0][j].w;
          }
          if (p.resid) {
            a.x += w16_lo<T>(pf_res[i][j].x); a.y += w16_hi<T>(pf_res[i][j].x);
            a.z += w16_lo<T>(pf_res[i][j].y); a.w += w16_hi<T>(pf_res[i][j].y);
          }
          uint2 o;
          o.x = pack16x2<T>(acc[0][i][j][0] * p.alpha + a.x, acc[0][i][j][1] * p.alpha + a.y);
          o.y = pack16x2<T>(acc[0][i][j][2] * p.alpha + a.z, acc[0][i][j][3] * p.alpha + a.w);
          *(uint2*)(smem + (wm * (C::FM * 16) + i * 16 + lm) * PITCH16 + (wn * C::BNW + j * 16 + q * 4) * 2) = o;
        }
      }
      TSTAMP();   // E1: LDS tile written
      __syncthreads();
      TSTAMP();   // E2
      constexpr int CH8 = BN / 8;                       // 16-byte chunks per row
      static_assert(NTHREADS % CH8 == 0, "");
      constexpr int RSTEP8 = NTHREADS / CH8;
      constexpr int NIT8 = (BM + RSTEP8 - 1) / RSTEP8;
      const int cs8 = tid % CH8, r08 = tid / CH8;
      const int n8 = n0 + cs8 * 8;
      uint4 v8[NIT8];
#pragma unroll
      for (int cc = 0; cc < NIT8; cc++) {
        const int row = r08 + cc * RSTEP8;
        v8[cc] = *(const uint4*)(smem + (row < BM ? row : 0) * PITCH16 + cs8 * 16);
      }
#pragma unroll
      for (int cc = 0; cc < NIT8; cc++) {
        const int row = r08 + cc * RSTEP8, mg = m0 + row;
        if (row < BM && mg < p.M && n8 < p.N) *(uint4*)((bf16_t*)Cb + cbase + (long)mg * p.ldc + n8) = v8[cc];
      }
      TSTAMP(); TSTAMP(); TSTAMP();   // keep the stamp count of the fp32 path
#ifdef EEG_STAGE_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TSTAMP();
#endif
      return;
    }
  }
  // tile row of (wm, i, r16) inside a pass: ((i % EPI_I) * WM + wm) * 16 + r16
  // (the fused 3-tap weight gradient has three accumulator sets: one tile per tap, sCt apart)
#pragma unroll
  for (int aset = 0; aset < C::NACC; aset++) {
  const long cbase_a = cbase + (long)aset * (C::NACC > 1 ? p.sCt : 0);
#pragma unroll
  for (int pass = 0; pass < C::FM / C::EPI_I; pass++) {
#pragma unroll
    for (int ii = 0; ii < C::EPI_I; ii++) {
      const int i = pass * C::EPI_I + ii;
      const int trow = (ii * C::WM + wm) * 16;
#pragma unroll
      for (int j = 0; j < FN; j++) {
        if constexpr (AMODE == GA_TR) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            *(float*)(smem + (trow + q * 4 + r) * C::EPI_PITCH + (wn * C::BNW + j * 16 + lm) * 4) = acc[aset][i][j][r] * p.alpha;
        } else {
          float4 v = make_float4(acc[0][i][j][0] * p.alpha, acc[0][i][j][1] * p.alpha, acc[0][i][j][2] * p.alpha, acc[0][i][j][3] * p.alpha);
          *(float4*)(smem + (trow + lm) * C::EPI_PITCH + (wn * C::BNW + j * 16 + q * 4) * 4) = v;
        }
      }
    }
    TSTAMP();   // E1: LDS tile written
    __syncthreads();
    TSTAMP();   // E2: barrier
    if (p.atomic_out) {
      // split-K / accumulate: one float per lane so each wave-instruction hits 256 contiguous bytes
      for (int c = tid; c < C::EPI_ROWS * BN; c += NTHREADS) {
        const int row = c / BN, col = c % BN;
        const int g = row >> 4, ii = g / C::WM, wmr = g % C::WM;
        const int m = m0 + wmr * (C::FM * 16) + (pass * C::EPI_I + ii) * 16 + (row & 15), n = n0 + col;
        if (m < p.M && n < p.N) atomicAdd((float*)Cb + cbase_a + (long)m * p.ldc + n, *(const float*)(smem + row * C::EPI_PITCH + col * 4));
      }
    } else {
      // Every thread owns one 4-column chunk (cs) of rows r0, r0+RSTEP, ...  All global reads of the pass are issued
      // back to back from clamped (always legal) addresses, then all LDS reads, then the stores: no per-row waits.
      static_assert(NTHREADS % CH == 0, "column chunk must be loop invariant");
      constexpr int RSTEP = NTHREADS / CH;
      constexpr int NIT = (C::EPI_ROWS + RSTEP - 1) / RSTEP;
      const int cs = tid % CH, r0 = tid / CH;
      const int n = n0 + cs * 4;
      const bool nok = n < p.N;
      const int nc = nok ? n : 0;
      int mm[NIT];
      float4 add[NIT];
#pragma unroll
      for (int cc = 0; cc < NIT; cc++) {
        const int row = r0 + cc * RSTEP;
        const int g = row >> 4, ii = g / C::WM, wmr = g % C::WM;
        const int m = m0 + wmr * (C::FM * 16) + (pass * C::EPI_I + ii) * 16 + (row & 15);
        mm[cc] = (row < C::EPI_ROWS && nok && m < p.M) ? m : -1;
        add[cc] = bias4;
      }
      if (p.rowvec) {
        // sample index of a row: the tile spans < 2 samples when rows_per_vec >= tile rows (one compare), else divide
        const int s0 = m0 / p.rows_per_vec;
        const int bnd = (s0 + 1) * p.rows_per_vec;
        const bool fast = p.rows_per_vec >= BM;
#pragma unroll
        for (int cc = 0; cc < NIT; cc++) {
          const int mc = mm[cc] < 0 ? m0 : mm[cc];
          const int sidx = fast ? s0 + (mc >= bnd ? 1 : 0) : mc / p.rows_per_vec;
          const float4 b = *(const float4*)(p.rowvec + (long)sidx * p.ld_rowvec + nc);
          add[cc].x += b.x; add[cc].y += b.y; add[cc].z += b.z; add[cc].w += b.w;
        }
      }
      if (p.resid) {
        if constexpr (sizeof(T) == 4) {
          float4 rv[NIT];
#pragma unroll
          for (int cc = 0; cc < NIT; cc++) rv[cc] = *(const float4*)((const T*)p.resid + (long)(mm[cc] < 0 ? m0 : mm[cc]) * p.ldr + nc);
#pragma unroll
          for (int cc = 0; cc < NIT; cc++) { add[cc].x += rv[cc].x; add[cc].y += rv[cc].y; add[cc].z += rv[cc].z; add[cc].w += rv[cc].w; }
        } else {
          uint2 rv[NIT];
#pragma unroll
          for (int cc = 0; cc < NIT; cc++) rv[cc] = *(const uint2*)((const T*)p.resid + (long)(mm[cc] < 0 ? m0 : mm[cc]) * p.ldr + nc);
#pragma unroll
          for (int cc = 0; cc < NIT; cc++) {
            add[cc].x += w16_lo<T>(rv[cc].x); add[cc].y += w16_hi<T>(rv[cc].x);
            add[cc].z += w16_lo<T>(rv[cc].y); add[cc].w += w16_hi<T>(rv[cc].y);
          }
        }
      }
      TSTAMP();   // E3: global reads issued
#ifdef EEG_STAGE_TIMING
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      TSTAMP();   // E4: global reads landed
#endif
      float4 v[NIT];
#pragma unroll
      for (int cc = 0; cc < NIT; cc++) {
        const int row = r0 + cc * RSTEP;
        v[cc] = *(const float4*)(smem + (row < C::EPI_ROWS ? row : 0) * C::EPI_PITCH + cs * 16);
        v[cc].x += add[cc].x; v[cc].y += add[cc].y; v[cc].z += add[cc].z; v[cc].w += add[cc].w;
      }
      if (p.out_f32 || sizeof(T) == 4) {
#pragma unroll
        for (int cc = 0; cc < NIT; cc++)
          if (mm[cc] >= 0) *(float4*)((float*)Cb + cbase_a + (long)mm[cc] * p.ldc + n) = v[cc];
      } else {
#pragma unroll
        for (int cc = 0; cc < NIT; cc++)
          if (mm[cc] >= 0) {
            uint2 o;
            o.x = pack16x2<T>(v[cc].x, v[cc].y);
            o.y = pack16x2<T>(v[cc].z, v[cc].w);
            *(uint2*)((bf16_t*)Cb + cbase_a + (long)mm[cc] * p.ldc + n) = o;
          }
      }
    }
    if (pass + 1 < C::FM / C::EPI_I || aset + 1 < C::NACC) __syncthreads();
  }
  }
  TSTAMP();   // E5: stores issued
#ifdef EEG_STAGE_TIMING
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TSTAMP();   // E6: stores retired
#endif
}

template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int BN, int STRIDE, int WMT, bool DMA>
int launch_k(eegldm_ctx* ctx, const GemmArgs& a_in) {
  using C = Cfg<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, DMA>;
  auto kern = gemm_kernel<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, DMA>;
  static DevOnce attr_once;
  static_assert(C::LDS_TOTAL <= 160 * 1024, "tile does not fit the 160 KiB LDS");
  if (C::LDS_TOTAL > 48 * 1024 && attr_once.need(ctx->device)) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_TOTAL));
  }
  GemmArgs a = a_in;
  dim3 grid((a.N + BN - 1) / BN, (a.M + C::BM - 1) / C::BM, a.batch * a.ztaps * a.splitk);
  a.xcd_swizzle = 0;
  if (a_in.xcd_swizzle) {
    if (AMODE == GA_TR && a.splitk > 1) {
      if ((a.batch == 1 || a.ngroup) && a.k_skew == 0.f && (long)grid.x * grid.y * a.ztaps > 1) a.xcd_swizzle = 3;   // (also wgrad-by-tap: ztaps = 3; grouped problems)
      else if (a.batch * a.ztaps == 1 && grid.x * grid.y > 1) {
        if (a.k_skew == 0.f) a.xcd_swizzle = 3;                    // equal chunks: contiguous runs, any split count
        else if (a.splitk % 8 == 0) a.xcd_swizzle = 2;             // skewed chunks: interleave the splits over the XCDs (mixes chunk lengths)
      }
    }
    else if (grid.x > 1 && grid.y % 8 == 0) a.xcd_swizzle = 1;
  }
  hipLaunchKernelGGL(kern, grid, dim3(C::NTHREADS), C::LDS_TOTAL, ctx->stream, a);
  LAUNCH_CHECK();
  return 0;
}
// staging engine: LDS-DMA ring for the 3-tap implicit-conv kernels when every block's K range is a whole number of
// stages (all production shapes), register staging otherwise (K tails, 1-tap kernels, fused wgrad)
template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int BN, int STRIDE, int WMT>
int launch_t(eegldm_ctx* ctx, const GemmArgs& a) {
  constexpr int KSTAGE = KSUB * Tr<T>::KC;
  if constexpr (AMODE == GA_CONV && TAPS == 3) {
    if (a.K % KSTAGE == 0 && a.splitk == 1) return launch_k<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, true>(ctx, a);
  }
  if constexpr (AMODE == GA_PLAIN && (WMT == 2 || WMT == 3)) {
    // 1x1 conv / Linear / attention products: same double-buffered DMA ring (K % KSTAGE == 0 on all production shapes)
    if (a.K % KSTAGE == 0 && a.splitk == 1) return launch_k<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, true>(ctx, a);
  }
  if constexpr (AMODE == GA_TR && BMODE == GB_TR && (WMT == 2 || WMT == 3)) {
    // weight gradients (fused 3-tap and 1-tap / Linear): every split is a whole number of stages when K is, and the source
    // of a chunk moves by a constant per stage unless the K index is remapped per tap (conv_map: unfused strided wgrad)
    if (a.K % KSTAGE == 0 && !a.conv_map) return launch_k<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, true>(ctx, a);
  }
  return launch_k<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE, WMT, false>(ctx, a);
}

// tile selection.  (Measured and removed from the dispatch in round 6, numbers in HISTORY.md: the 256-row / 8-wave variant WMT = 4 -- no faster than
// 128-row blocks at 2 blocks per CU, profiles/r01_gemm_tile_sweep.txt -- and 192-row tiles for the 3-tap kernels -- +5..8 % in isolation from K = 512
// up, nothing at step level, spills in the transposed-weight data gradient.)
template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int STRIDE>
int launch_bn(eegldm_ctx* ctx, const GemmArgs& a) {
  if (a.N > 64) return launch_t<T, AMODE, BMODE, TAPS, KSUB, 128, STRIDE, 2>(ctx, a);
  if (a.N > 32) return launch_t<T, AMODE, BMODE, TAPS, KSUB, 64, STRIDE, 2>(ctx, a);
  return launch_t<T, AMODE, BMODE, TAPS, KSUB, 32, STRIDE, 2>(ctx, a);
}

template <typename T>
int launch_modes(eegldm_ctx* ctx, const GemmArgs& a) {
  if (a.amode == GA_CONV) {
    EEG_CHECK(a.batch == 1, "conv mode expects flattened rows (batch=1)");
    EEG_CHECK(a.Lin == a.Lout * a.stride, "conv mode needs Lin == Lout*stride (got %d, %d, %d)", a.Lin, a.Lout, a.stride);
    if (a.bmode == GB_NT) {
      if (a.taps == 3 && a.stride == 1) return launch_bn<T, GA_CONV, GB_NT, 3, 1, 1>(ctx, a);
      if (a.taps == 3 && a.stride == 2) return launch_bn<T, GA_CONV, GB_NT, 3, 1, 2>(ctx, a);
    } else {
      EEG_CHECK(a.stride == 1, "dgrad is expressed as a stride-1 conv over a (virtually upsampled) gradient");
      if (a.taps == 3) return launch_bn<T, GA_CONV, GB_TR, 3, 1, 1>(ctx, a);
    }
    EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "conv gemm: taps=%d stride=%d bmode=%d", a.taps, a.stride, a.bmode);
  }
  if (a.amode == GA_TR && a.bmode == GB_TR && a.taps == 3) {   // fused 3-tap wgrad (see Cfg::WG3); BN <= 64 keeps 3 accumulator sets in registers
    // (The 128 x 128 x 3-tap tile as one 8-wave block per CU measured 5-8 % slower than two independent 128 x 64 blocks in round 2; removed in round 6.
    //  Round 5 measured two more forms of this tile, both equal to it within 1 % over the UNet's shapes and since removed from the dispatch:
    //  32-deep stages in a 4-deep LDS-DMA ring -- launch_t<T, GA_TR, GB_TR, 3, 1, 64, 1, 2> -- and eight waves of 32 x 32 x 3 taps, four per
    //  SIMD at 128 VGPRs -- WMT = 5; and a third, 15 % SLOWER: a three-deep ring at two blocks per CU -- WMT = 6, Cfg::TIGHT3.  DESIGN.md section 9.)
    if (a.N > 32) return launch_t<T, GA_TR, GB_TR, 3, 2, 64, 1, 2>(ctx, a);
    return launch_t<T, GA_TR, GB_TR, 3, 2, 32, 1, 2>(ctx, a);
  }
  EEG_CHECK(a.taps == 1, "taps>1 needs conv A mode");
  // 192-row tiles (two waves x six fragments) for the 1-tap kernels: a 16 KB stage of a 128-row tile carries only 32 MFMAs per
  // wave against the ~2500-cycle DMA round trip; 192 rows carry 48 at the same 2 blocks per CU (80 KB LDS each).  The attention
  // products at T = 192 additionally stop computing half-empty tiles (one sample = 1.5 tiles of 128 rows; 64-wide N tiles for
  // the 192-wide QK^T output)
  if constexpr (sizeof(T) == 2) {
    // (1-tap kernels: +2..8 % on every UNet shape; batched attention products also when 128 divides M -- T = 384 / 768, pixel-space model: +0.7 % of that step)
    const bool attn192 = a.batch > 1 && a.M % 192 == 0;
    const bool conv192 = a.batch == 1 && a.M % 192 == 0 && a.K >= 64;
    if (a.amode == GA_PLAIN && (attn192 || conv192) && a.splitk == 1 && a.K % 64 == 0) {
      if (a.bmode == GB_NT) return (a.N % 128 == 0) ? launch_t<T, GA_PLAIN, GB_NT, 1, 2, 128, 1, 3>(ctx, a) : launch_t<T, GA_PLAIN, GB_NT, 1, 2, 64, 1, 3>(ctx, a);
      return (a.N % 128 == 0) ? launch_t<T, GA_PLAIN, GB_TR, 1, 2, 128, 1, 3>(ctx, a) : launch_t<T, GA_PLAIN, GB_TR, 1, 2, 64, 1, 3>(ctx, a);
    }
  }
  if constexpr (sizeof(T) == 2) {
    // transposed-operand (TN) products whose M is a multiple of 192 but not of 128: the attention dK / dV gradients (192-row
    // samples, batched) stop computing a half-empty second 128-row tile; 1-tap weight gradients with 192 k output channels
    // (also when 128 divides M, and for the batched dK / dV products at T = 768: measured, no gain -- round 4)
    if (a.amode == GA_TR && a.bmode == GB_TR && a.taps == 1 && !a.conv_map && a.M % 192 == 0 && a.M % 128 != 0 && a.K % 64 == 0 && a.N % 64 == 0)
      return (a.N % 128 == 0) ? launch_t<T, GA_TR, GB_TR, 1, 2, 128, 1, 3>(ctx, a) : launch_t<T, GA_TR, GB_TR, 1, 2, 64, 1, 3>(ctx, a);
  }
  if (a.amode == GA_PLAIN && a.bmode == GB_NT) return launch_bn<T, GA_PLAIN, GB_NT, 1, 2, 1>(ctx, a);
  if (a.amode == GA_PLAIN && a.bmode == GB_TR) return launch_bn<T, GA_PLAIN, GB_TR, 1, 2, 1>(ctx, a);
  if (a.amode == GA_TR && a.bmode == GB_TR) return launch_bn<T, GA_TR, GB_TR, 1, 2, 1>(ctx, a);
  EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "gemm: amode=%d bmode=%d", a.amode, a.bmode);
}

}  // namespace

// dst[i] += sum_r ws[r][i].  A block owns 64 float4 outputs; its four waves each stream a quarter of the split-K partial tiles
// (independent 16-byte loads, 1 KB per wave instruction), the partial sums meet in LDS and wave 0 updates dst.
__global__ __launch_bounds__(256) void splitk_fold_kernel(const float* __restrict__ ws, int nsplit, long n, float* __restrict__ dst) {
  __shared__ float4 red[3][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + tx) * 4;
  const bool ok = i < n;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    int r = ty;
    for (; r + 12 < nsplit; r += 16) {
      const float4 v0 = *(const float4*)(ws + (long)r * n + i), v1 = *(const float4*)(ws + (long)(r + 4) * n + i);
      const float4 v2 = *(const float4*)(ws + (long)(r + 8) * n + i), v3 = *(const float4*)(ws + (long)(r + 12) * n + i);
      s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
      s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; r < nsplit; r += 4) {
      const float4 v = *(const float4*)(ws + (long)r * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (ty > 0) red[ty - 1][tx] = s;
  __syncthreads();
  if (ty == 0 && ok) {
    float4 d = *(const float4*)(dst + i);
#pragma unroll
    for (int k = 0; k < 3; k++) { const float4 v = red[k][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
    *(float4*)(dst + i) = d;
  }
}

// dst_b[i] += sum_r ws[(b * nsplit + r) * n + i] for every problem b of a grouped launch (blockIdx.y = b)
__global__ __launch_bounds__(256) void splitk_fold_grouped_kernel(const float* __restrict__ ws, int nsplit, long n, const GemmGroup* __restrict__ grp) {
  __shared__ float4 red[3][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long i = ((long)blockIdx.x * 64 + tx) * 4;
  const bool ok = i < n;
  const float* w = ws + (long)blockIdx.y * nsplit * n;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
    for (int r = ty; r < nsplit; r += 4) {
      const float4 v = *(const float4*)(w + (long)r * n + i);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  if (ty > 0) red[ty - 1][tx] = s;
  __syncthreads();
  if (ty == 0 && ok) {
    float* dp = grp->Dst[blockIdx.y] + i;
    float4 d = *(const float4*)dp;
#pragma unroll
    for (int k = 0; k < 3; k++) { const float4 v = red[k][tx]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
    *(float4*)dp = d;
  }
}

int gemm_launch_grouped(eegldm_ctx* ctx, const GemmArgs& a_in, const GemmGroup& g, int slot) {
  GemmArgs a = a_in;
  // device copy of the pointer table, cached per flush slot: the arena hands out the same addresses step after step, so after the first
  // step nothing is uploaded; a changed table is re-uploaded after draining both streams (an earlier launch may still read the old one)
  EEG_CHECK(slot >= 0 && slot < 4096, "group slot");
  if (slot >= ctx->grp_cap) {
    HIP_TRY(hipStreamSynchronize(ctx->stream)); if (ctx->side_on) HIP_TRY(hipStreamSynchronize(ctx->side));
    const int cap = slot + 64;
    GemmGroup* nd = nullptr; HIP_TRY(hipMalloc(&nd, sizeof(GemmGroup) * cap));
    if (ctx->grp_dev) { HIP_TRY(hipMemcpy(nd, ctx->grp_dev, sizeof(GemmGroup) * ctx->grp_cap, hipMemcpyDeviceToDevice)); HIP_TRY(hipFree(ctx->grp_dev)); }
    ctx->grp_dev = nd; ctx->grp_cap = cap; ctx->grp_host.resize(cap);
  }
  if ((int)ctx->grp_host.size() < ctx->grp_cap) ctx->grp_host.resize(ctx->grp_cap);
  if (memcmp(&ctx->grp_host[slot], &g, sizeof(GemmGroup)) != 0) {
    HIP_TRY(hipStreamSynchronize(ctx->stream)); if (ctx->side_on) HIP_TRY(hipStreamSynchronize(ctx->side));
    HIP_TRY(hipMemcpy(ctx->grp_dev + slot, &g, sizeof(GemmGroup), hipMemcpyHostToDevice));
    ctx->grp_host[slot] = g;
  }
  a.grp = ctx->grp_dev + slot;
  EEG_CHECK(a.ngroup >= 1 && a.ngroup <= GEMM_MAX_GROUP && a.amode == GA_TR && a.bmode == GB_TR && a.ztaps == 1 && (a.taps == 1 || a.taps == 3), "not a groupable weight gradient");
  const int epc = a.dtype == EEGLDM_F32 ? 4 : 8;
  EEG_CHECK(a.M % epc == 0 && a.N % epc == 0 && a.N % 4 == 0, "operand alignment");
  for (int i = 0; i < a.ngroup; i++) EEG_CHECK(g.lda[i] % epc == 0 && g.ldb[i] % epc == 0 && g.A[i] && g.B[i] && g.Dst[i], "group member %d: leading dimension / null pointer", i);
  if (a.splitk < 1) a.splitk = 1;
  a.batch = a.ngroup; a.out_f32 = 1; a.ups = 1; a.rows_per_vec = 1; a.ldc = a.N; a.sCt = (long)a.M * a.N; a.colsum = nullptr;
  const long fold_n = (long)a.taps * a.M * a.N;
  const int KST = 2 * (a.dtype == EEGLDM_F32 ? 16 : 32);
  int per = (a.K + a.splitk - 1) / a.splitk; per = (per + KST - 1) / KST * KST;
  a.splitk = (a.K + per - 1) / per;                     // every split writes its whole tile: no empty trailing splits
  const size_t need = (size_t)a.ngroup * a.splitk * fold_n * sizeof(float);
  if (ctx->splitk_ws_bytes < need) {
    if (ctx->splitk_ws) { HIP_TRY(hipStreamSynchronize(ctx->stream)); if (ctx->side_on) HIP_TRY(hipStreamSynchronize(ctx->side)); HIP_TRY(hipFree(ctx->splitk_ws)); ctx->splitk_ws = nullptr; ctx->splitk_ws_bytes = 0; }
    HIP_TRY(hipMalloc(&ctx->splitk_ws, need)); ctx->splitk_ws_bytes = need;
  }
  a.C = ctx->splitk_ws; a.sCk = fold_n; a.sCb = (long)a.splitk * fold_n; a.atomic_out = 0; a.k_skew = 0.f;
  a.zero_page = ctx->zero_page;
  a.xcd_swizzle = 1;
  ProfRec rec; const bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = a.taps == 3 ? PROF_CONV_WGRAD : PROF_GEMM_TN;
    rec.flops = 2.0 * a.M * a.N * (double)a.K * a.taps * a.ngroup;
    rec.M = a.M; rec.N = a.N; rec.K = a.K; rec.taps = a.taps; rec.splitk = a.splitk;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b));
    HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  int rc;
  if (a.dtype == EEGLDM_F32) rc = launch_modes<float>(ctx, a);
  else if (a.dtype == EEGLDM_BF16) rc = launch_modes<bf16_t>(ctx, a);
  else if (a.dtype == EEGLDM_F16) rc = launch_modes<f16_t>(ctx, a);
  else EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "dtype %d", a.dtype);
  if (rc == 0) {
    hipLaunchKernelGGL(splitk_fold_grouped_kernel, dim3((unsigned)((fold_n / 4 + 63) / 64), a.ngroup), dim3(256), 0, ctx->stream, (const float*)ctx->splitk_ws, a.splitk, fold_n, a.grp);
    LAUNCH_CHECK();
  }
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return rc;
}

int gemm_launch(eegldm_ctx* ctx, const GemmArgs& a_in) {
  GemmArgs a = a_in;
  if (a.splitk < 1) a.splitk = 1;
  if (a.ztaps < 1) a.ztaps = 1;
  if (a.ups < 1) a.ups = 1;
  if (a.batch < 1) a.batch = 1;
  if (a.rows_per_vec < 1) a.rows_per_vec = 1;
  const int epc = a.dtype == EEGLDM_F32 ? 4 : 8;
  EEG_CHECK(a.M > 0 && a.N > 0 && a.K > 0, "empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  // 16-byte vector loads: the contiguous dim of each operand must be a multiple of a chunk
  if (a.amode == GA_TR) EEG_CHECK(a.M % epc == 0 && a.lda % epc == 0, "A(TR): M, lda must be multiples of %d", epc);
  else EEG_CHECK(a.K % epc == 0 && a.lda % epc == 0, "A: K, lda must be multiples of %d (K=%d lda=%ld)", epc, a.K, a.lda);
  if (a.bmode == GB_TR) EEG_CHECK(a.N % epc == 0 && a.ldb % epc == 0, "B(TR): N, ldb must be multiples of %d", epc);
  else EEG_CHECK(a.K % epc == 0 && a.ldb % epc == 0, "B: K, ldb must be multiples of %d", epc);
  EEG_CHECK(!(a.atomic_out || a.splitk > 1) || a.out_f32, "atomic / split-K output must be f32");
  EEG_CHECK(a.N % 4 == 0 && a.ldc % 4 == 0, "N and ldc must be multiples of 4 (vector epilogue): N=%d ldc=%ld", a.N, a.ldc);
  EEG_CHECK(!a.resid || a.ldr % 4 == 0, "ldr must be a multiple of 4");
  EEG_CHECK(!a.rowvec || a.ld_rowvec % 4 == 0, "ld_rowvec must be a multiple of 4");
  EEG_CHECK(!a.colsum || (a.amode == GA_TR && a.dtype != EEGLDM_F32 && a.batch == 1), "fused column sums need a 16-bit transposed A operand and batch 1");
  // Split-K weight gradients of 1-tap TN products: partial tiles are WRITTEN to a workspace (coalesced stores through the
  // LDS epilogue) and folded into dW afterwards, instead of draining millions of fp32 atomics at ~370 G/s
  // (profiles/r01_gemm_stage_timing.txt); the fused 3-tap kernel keeps its register atomics (three accumulator sets).
  float* fold_dst = nullptr; long fold_n = 0;
  constexpr bool fused_ws = true;      // fused 3-tap kernel: workspace partials (plain stores from the accumulators) + fold (the register atomics of round 2 drained at ~370 G/s)
  // (deterministic mode: also the tap-by-tap weight gradients of strided convs, ztaps > 1 with contiguous taps -- their atomics are the only
  // other split-K route, and a single split is one block per tile walking all of B x L)
  const bool det_zt = eeg_deterministic() && a.ztaps > 1 && a.taps == 1 && a.sCt == (long)a.M * a.N;
  if (a.splitk > 1 && a.amode == GA_TR && a.bmode == GB_TR && (a.taps == 1 || (a.taps == 3 && a.sCt == (long)a.M * a.N && fused_ws)) && (a.ztaps == 1 || det_zt) && a.batch == 1 && a.atomic_out && a.ldc == a.N) {
    const size_t need = (size_t)a.splitk * a.taps * a.ztaps * a.M * a.N * sizeof(float);
    if (need <= (size_t)512 << 20) {
      if (ctx->splitk_ws_bytes < need) {
        if (ctx->splitk_ws) { HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(ctx->splitk_ws)); ctx->splitk_ws = nullptr; ctx->splitk_ws_bytes = 0; }
        HIP_TRY(hipMalloc(&ctx->splitk_ws, need)); ctx->splitk_ws_bytes = need;
      }
      // every split writes its whole tile (empty trailing splits are avoided by the even K partition below)
      const int KST = 2 * (a.dtype == EEGLDM_F32 ? 16 : 32);
      int per = (a.K + a.splitk - 1) / a.splitk; per = (per + KST - 1) / KST * KST;
      a.splitk = (a.K + per - 1) / per;
      fold_dst = (float*)a.C; fold_n = (long)a.taps * a.ztaps * a.M * a.N;
      a.C = ctx->splitk_ws; a.sCk = fold_n; a.atomic_out = 0;
      if (det_zt) HIP_TRY(hipMemsetAsync(ctx->splitk_ws, 0, (size_t)a.splitk * fold_n * sizeof(float), ctx->stream));      // (edge tiles of a strided map: nothing unwritten is ever folded)
    }
  }
  if (a.splitk > 1 && !fold_dst && eeg_deterministic()) a.splitk = 1;      // no workspace route for this product: one writer per output element instead of racing K splits
  if (a.splitk > 1 && !fold_dst) a.atomic_out = 1;
  a.zero_page = ctx->zero_page;
  a.xcd_swizzle = 1;
  ProfRec rec; bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = a.amode == GA_CONV ? (a.bmode == GB_NT ? PROF_CONV_FWD : PROF_CONV_DGRAD)
              : (a.amode == GA_TR ? ((a.conv_map || a.taps == 3) ? PROF_CONV_WGRAD : PROF_GEMM_TN) : (a.bmode == GB_NT ? PROF_GEMM_NT : PROF_GEMM_NN));
    // algorithmic work: the transposed (strided) dgrad multiplies a half-zero virtual signal; only the real taps count
    rec.flops = 2.0 * a.M * a.N * (double)a.K * a.taps * a.ztaps * a.batch / (a.ups > 1 ? a.ups : 1);
    rec.M = a.M; rec.N = a.N; rec.K = a.K; rec.taps = a.taps * a.ztaps; rec.splitk = a.splitk;
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b));
    HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  // skewed K chunks for the atomic split-K epilogue (see gemm_kernel); the workspace path writes plain stores and keeps equal chunks.
  // The atomics per block are constant (3 x 128 x 64), so the skew that pays shrinks with the stages per block (measured, B=256:
  // 12 or 24 stages per block -11 % at 0.5; 48 stages -3 % at 0.25; 73+ stages: any skew loses) -> (66 - stages) / 84, clamped.
  a.k_skew = 0.f;
  if (a.splitk > 1 && a.amode == GA_TR && a.atomic_out && !fold_dst) {
    const int kst = (a.dtype == EEGLDM_F32 ? 16 : 32) * 2;
    const long S = ((long)a.K + kst - 1) / kst;
    const float spb = (float)S / (float)a.splitk;
    float sk = (66.0f - spb) / 84.0f;
    sk = sk < 0.f ? 0.f : (sk > 0.5f ? 0.5f : sk);
    if (sk > 0.f && spb * (1.0f - sk) >= 2.0f) a.k_skew = sk;      // every chunk keeps at least two stages
  }
  int rc = 0;
  // 192 x 256 tile, one workgroup per CU (gemm_big.hip): 3-tap and 1 x 1 convs of the 256- / 512-channel levels, forward and (through the
  // transposed K-blocked weight copy B_alt) data gradient
  if ((a.amode == GA_CONV || (a.amode == GA_PLAIN && a.taps == 1 && a.batch == 1)) && (a.dtype == EEGLDM_BF16 || a.dtype == EEGLDM_F16) && !fold_dst) {
    if (a.bmode == GB_NT) rc = gemm_big_try(ctx, a);
    else if (a.B_alt) { GemmArgs b = a; b.B = a.B_alt; b.bmode = GB_NT; b.b_kblk = 1; rc = gemm_big_try(ctx, b); }
    if (rc < 0) return rc;
  }
  if (rc == 1) rc = 0;
  else if (a.dtype == EEGLDM_F32) rc = launch_modes<float>(ctx, a);
  else if (a.dtype == EEGLDM_BF16) rc = launch_modes<bf16_t>(ctx, a);
  else if (a.dtype == EEGLDM_F16) rc = launch_modes<f16_t>(ctx, a);
  else EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "dtype %d", a.dtype);
  if (rc == 0 && fold_dst) {
    hipLaunchKernelGGL(splitk_fold_kernel, dim3((unsigned)((fold_n / 4 + 63) / 64)), dim3(256), 0, ctx->stream, (const float*)ctx->splitk_ws, a.splitk, fold_n, fold_dst);
    LAUNCH_CHECK();
  }
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return rc;
}
