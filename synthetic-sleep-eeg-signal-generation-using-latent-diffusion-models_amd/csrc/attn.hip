// Fused single-head attention "chain" kernel for short sequences (T <= 256, the UNet's T = 192; unet.py:107-125):
//   forward :  S = alpha * Q K^T  ->  P = softmax_rows(S) (fp32, rounded to bf16)  ->  O = P V
//   backward:  dP = dO V^T        ->  dS = alpha * P o (dP - rowsum(dP o P))        ->  dQ = dS K
// One block = one sample x 64 query rows.  Both products run on MFMA with LDS-DMA staged operands; the T-wide score tile
// never leaves the CU: it lives in registers between the two products and is written once as bf16 (P for the backward pass,
// dS for the dK / dV products, which stay ordinary batched TN GEMMs).  Replaces QK^T (fp32 logits) + softmax + PV, i.e.
// three launches and ~95 MB of logits/probs traffic per attention block at B = 256.  bf16 only (the fp32 parity path keeps
// the unfused composition in ops.hip).
// Long sequences (T = 768, the pixel-space model's attention level, round 3): NCW = 8 column waves, each owning T/128 key fragments
// (96 score accumulators per lane instead of 192), one 8-wave block per CU: the product-1 ring (3 x 52 KB) takes the whole LDS.  The
// bf16 score tile (64 x 768 = 97 KB) is NOT kept in LDS for product 2: with it only 48 KB were left for the V / K tiles, two 16 KB
// tiles in flight against a ~2500-cycle DMA round trip and 8 MFMAs per wave and tile -- the pixel-space step did not move at all
// (22.16 vs 22.14 ms).  The scores stay in the owning wave's registers as packed bf16 and each 32-key slice (4 KB) is handed to the
// block through a double-buffered LDS slot just before the tiles that need it; the freed LDS holds an 8-deep ring of V / K tiles.
#include "common.h"
#include "internal.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_p;

__device__ __forceinline__ int swz1(int row, int slot) { return slot ^ (((row >> 2) & 1) * 3); }     // 64-byte rows (gemm.hip nt_swz<1>)
__device__ __forceinline__ int trswz256(int row, int colbyte) { return colbyte ^ ((((row & 3) | (((row >> 3) & 1) << 2)) * 32) & 511); }
__device__ __forceinline__ void dma16a(const void* g, unsigned lds_off) {
  const unsigned off_s = __builtin_amdgcn_readfirstlane(lds_off);      // wave-uniform by construction; M0 must come from an SGPR
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(off_s), "v"(g) : "memory", "m0");
}
template <typename T>
__device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x4& acc) {
  if constexpr (Is16<T>::f16) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
  else acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
// K-strided 16-column fragment of a [32 k rows][256 cols] bf16 tile (see gemm.hip read_tr_bf16)
__device__ __forceinline__ uint4 read_tr256(const char* tile, int x0, int lm, int q) {
  const int row0 = 8 * q + (lm >> 2), colb = (x0 + 4 * (lm & 3)) * 2;
  const char* p0 = tile + row0 * 512 + trswz256(row0, colb);
  const char* p1 = tile + (row0 + 4) * 512 + trswz256(row0 + 4, colb);
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p1));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}

// The same K-strided fragment from an UNSWIZZLED row-major tile of pitch `pitch` bytes (the score tile): lane (lm, q) gets rows
// k0 + 8q .. + 7 of column x0 + lm -- the MFMA operand of a product that reduces over the tile's ROWS (dK = dS^T Q, dV = P^T dO)
__device__ __forceinline__ uint4 read_tr_rows(const char* tile, int pitch, int k0, int x0, int lm, int q) {
  const int row0 = k0 + 8 * q + (lm >> 2), colb = (x0 + 4 * (lm & 3)) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(tile + row0 * pitch + colb));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(tile + (row0 + 4) * pitch + colb));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}

struct ChainArgs {
  const bf16_t* A1; long lda1, sA1;    // Q (fwd) / dO (bwd): [B][T][lda1], reduction (C) contiguous
  const bf16_t* B1; long ldb1, sB1;    // K (fwd) / V (bwd)
  const bf16_t* B2; long ldb2, sB2;    // V (fwd) / K (bwd): [T][C], used through transpose reads
  bf16_t* P; long sP;                  // probabilities [B][T][T]: written by fwd, read by bwd
  bf16_t* dS; long sdS;                // bwd: scaled score gradient [B][T][T]
  bf16_t* O; long ldo, sO;             // out (fwd) / dQ (bwd)
  int T, C; float alpha;
  int xcd;                             // XCD-aware block order (set by the launcher): all query tiles of a sample on one XCD
  unsigned long long* stamps;          // developer aid (EEGLDM_ATTN_STAMPS=1): shader-clock stamps of block 0's phases, else null
  // backward, whole-sample blocks only (fuse_kv): dK = dS^T Q as a second pass of product 2 over the score tile, read TRANSPOSED;
  // Q3 = the query rows of the sample; the result goes to dK (same leading dimension as O)
  // fuse_kv == 2 (round 5): a THIRD pass dV = P^T dO behind it -- the probabilities are fetched again (73 KB per sample, L2) at the end of
  // the dK pass and overwrite the dS tile, tiles = rows of dO (= A1); the result goes to dV
  int fuse_kv; const bf16_t* Q3; long ldq3, sQ3; bf16_t* dK; bf16_t* dV;
};
#define ATTN_STAMP(k) do { if (p.stamps && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) p.stamps[k] = __builtin_readcyclecounter(); } while (0)

constexpr int NTA = 256;

// NQ = 64-row query tiles per block: 1 (grid = T/64 x B) or NJ (one block = one whole sample: K / V tiles are staged once
// for all query rows and, at B = 256, the grid is exactly one block per CU -- no 1.5-round tail).
template <int NJT, int MODE, int NQ, int NCW, int NC2T, typename T16>
__global__ __launch_bounds__(64 * NCW * NQ) void attn_chain_kernel(const ChainArgs p) {
  constexpr int T = 64 * NJT;                     // keys
  constexpr int NJ = 4 * NJT / NCW;               // 16-column key fragments per column wave
  constexpr int QR = 64 * NQ;                     // query rows per block
  constexpr int NWV = NCW * NQ;                   // waves per block
  constexpr int A_BYTES = QR * 64, B_BYTES = T * 64, STG = A_BYTES + B_BYTES;
  // 3-deep DMA rings: a stage carries only 12-16 MFMAs per wave against a ~2500-cycle DMA round trip, so two stages are kept
  // in flight (with a 2-deep ring the kernel ran at one DMA latency per stage: 76 us per launch)
  constexpr int STAGE_AREA = (3 * STG > 49152) ? 3 * STG : 49152;
  constexpr int PP = T * 2 + 16;                  // padded row pitch of the bf16 score tile (conflict-free b128 fragment reads)
  // score tile behind the staging area -- or, when the product-1 ring needs (nearly) the whole LDS (NCW = 8), laid over that ring behind
  // the 48 KB of product-2 staging: product 1 is complete (block barrier) before the first score is written
  // NCW = 8 ("long" variant, see the file header): no score tile; an 8-deep ring of 16 KB product-2 tiles, two 4 KB score slices, `red`
  constexpr int D2 = 8, SL_OFF = D2 * 16384, RED_OFF8 = SL_OFF + 2 * 4096;
  // NCW = 4: product-2 tile ring inside the staging area, 4 deep where that area has 64 KB (whole-sample blocks at T = 192: every block
  // streams its own sample's V / K from HBM, ~3.5 us per round trip under load -- the unit time was latency / tiles in flight)
  constexpr int DO2 = (STAGE_AREA >= 65536) ? 4 : 3;
  extern __shared__ __attribute__((aligned(16))) char sm[];
  char* pt = sm + STAGE_AREA;                     // [QR][PP]   (NCW = 4 only)
  float* red = (NCW == 8) ? (float*)(sm + RED_OFF8) : (float*)(pt + QR * PP);            // [QR rows][NCW column waves]
  const int tid = threadIdx.x, lane = tid & 63, lm = lane & 15, q = lane >> 4;
  const unsigned wv = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave in block
  const unsigned w = wv % NCW, wq = wv / NCW;                    // column wave (owns NJ key fragments) / query tile of this wave
  // Block -> (sample, query tile).  Plain order deals the T / QR tiles of one sample round-robin to the 8 XCDs, so every XCD's L2 sees
  // the K / V of every sample in flight (T = 768, B = 64: 21 samples x 2.4 MB against 4 MB of L2 -- the kernel ran at the
  // Infinity-Cache rate, 205 us).  XCD-aware order (B % 8 == 0): ids = x (mod 8) run on XCD x; the j-th of them is tile j % nt of
  // sample (j / nt) * 8 + x, so a sample's tiles share one L2 and K / V come from HBM once.
  int b = blockIdx.y, m0 = blockIdx.x * QR;                      // first query row of the block
  if (p.xcd) {
    const int nt = gridDim.x, lin = blockIdx.y * nt + blockIdx.x, x = lin & 7, j = lin >> 3;
    b = (j / nt) * 8 + x; m0 = (j % nt) * QR;
  }
  const int mq = wq * 64;                                        // this wave's query rows inside the block tile
  const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)sm;
  const bf16_t* A1 = p.A1 + (long)b * p.sA1 + (long)m0 * p.lda1;
  const bf16_t* B1 = p.B1 + (long)b * p.sB1;
  const bf16_t* B2 = p.B2 + (long)b * p.sB2;

  // backward: this lane's probabilities (rows i*16+lm, columns (w*NJ+j)*16 + q*4 .. +3), fetched before anything else
  uint2 pr[4][NJ];
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++)
        pr[i][j] = *(const uint2*)(p.P + (long)b * p.sP + (long)(m0 + mq + i * 16 + lm) * T + (w * NJ + j) * 16 + q * 4);
  }

  // ---------------- product 2 staging (V / K tile [32 rows][256 cols], transpose-read layout) ----------------
  const int nks = T / 32, nnc = p.C / 256, nu = nks * nnc;
  // unit u of the product-2 sequence: pass pr = u / nu (0: the B2 operand -- V forward, K backward; 1 / 2: the fused dK / dV passes of
  // the backward, whose tiles are rows of Q / dO), 256-column group nc, 32-row slice ks
  // (every field is copied to a local first: selecting between members of the by-value argument struct at run time made hipcc keep
  //  a copy of the struct in scratch -- 320 bytes per lane)
  const bool fuse_kv = MODE == 1 && p.fuse_kv != 0;
  const bool fuse_v = MODE == 1 && p.fuse_kv == 2;
  const bf16_t* const src1 = fuse_kv ? p.Q3 + (long)b * p.sQ3 : B2;
  const bf16_t* const src2 = A1;                               // third pass: rows of dO (whole-sample blocks: m0 = 0)
  const long ld0 = p.ldb2, ld1 = fuse_kv ? p.ldq3 : p.ldb2, ld2 = p.lda1;
  bf16_t* const dst0 = p.O; bf16_t* const dst1 = fuse_kv ? p.dK : p.O; bf16_t* const dst2 = fuse_v ? p.dV : p.O;
  const long sO = p.sO, ldo = p.ldo;
  // (sources / destinations are switched at pass boundaries through two-way selects only: a three-way `pass == 0 ? a : pass == 1 ? b : c`
  //  became a lookup table on the stack -- 160 bytes of scratch per lane in every instantiation)
  auto issue_tile = [&](const bf16_t* src, const long ld, int uu, int buf) __attribute__((always_inline)) {
    const int nc = uu / nks, ks = uu - nc * nks;
    const unsigned base = lds0 + buf * 16384;
    // 16 wave-instructions per tile: instruction ii goes to wave ii % NWV (waves < 16 % NWV issue one more than the others)
#pragma unroll
    for (int i = 0; i < (16 + NWV - 1) / NWV; i++) {
      const int ii = wv + NWV * i;
      if (ii < 16) {
        const int c = ii * 64 + lane, krow = c >> 5, cs = c & 31;
        const int seg = trswz256(krow, cs * 16) >> 4;
        dma16a(src + (long)(ks * 32 + krow) * ld + nc * 256 + seg * 8, base + ii * 1024);
      }
    }
  };
  const int per2 = (int)((16 - (int)wv + NWV - 1) / NWV);       // this wave's DMA instructions per tile (wave-uniform)
  // ---------------- product 2: out[64][C] = tile[64][T] . B2[T][C], 256 output columns per pass ----------------
  // a wave's share of a pass: all 64 rows x 64 columns (4 column waves), or 32 rows x 64 columns (8 column waves: two row halves)
  constexpr int RF2 = 16 / NCW;                    // 16-row fragments per wave: 4 or 2
  const unsigned wc2 = w & 3;                      // 64-column quarter of the pass
  const int mq2 = mq + (int)(w >> 2) * 32;         // first row of this wave's share
  // ---------------- product 1: S^T fragments, reduction over C in 32-wide stages ----------------
  ATTN_STAMP(0);
  f32x4 acc[4][NJ];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = p.C / 32;
  // per-lane source pointers of this wave's DMA chunks (chunk c of a tile <-> row c/4, swizzled 16-byte slot)
  // A tile: QR*4 chunks = 4*NQ wave-instructions (waves 0 .. 4*NQ-1 take one each); B tile: T*4 chunks = 4*NJT instructions, NB1 per wave
  constexpr int NB1 = (4 * NJT + NWV - 1) / NWV;
  static_assert((4 * NJT) % NWV == 0, "B-tile DMA instructions must divide evenly over the waves (uniform vmcnt)");
  const bool a_mine = wv < 4 * NQ;                 // wave-uniform (NCW = 8: the upper half of the waves stages no A chunk)
  const bf16_t* asrc; const bf16_t* bsrc[NB1];
  { const int c = (a_mine ? wv : 0) * 64 + lane, row = c >> 2, slot = swz1(row, c & 3); asrc = A1 + (long)row * p.lda1 + slot * 8; }
#pragma unroll
  for (int i = 0; i < NB1; i++) { const int c = (wv + NWV * i) * 64 + lane, row = c >> 2, slot = swz1(row, c & 3); bsrc[i] = B1 + (long)row * p.ldb1 + slot * 8; }
  auto issue1 = [&](int s, int buf) __attribute__((always_inline)) {
    const unsigned base = lds0 + buf * STG;
    if (a_mine) dma16a(asrc + s * 32, base + wv * 1024);
#pragma unroll
    for (int i = 0; i < NB1; i++) dma16a(bsrc[i] + s * 32, base + A_BYTES + (wv + NWV * i) * 1024);
  };
  constexpr int PER1 = 1 + NB1;                   // DMA instructions per wave per stage (one less for waves without an A chunk)
  issue1(0, 0);
  if (nst > 1) issue1(1, 1);
  for (int s = 0; s < nst; s++) {
    if (s + 1 < nst) {                             // stage s landed, stage s+1 may be in flight
      if (a_mine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER1 - 1) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* sa = sm + (s % 3) * STG; const char* sb = sa + A_BYTES;
    uint4 af[4], bfr[NJ];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int row = mq + i * 16 + lm; af[i] = *(const uint4*)(sa + row * 64 + swz1(row, q) * 16); }
#pragma unroll
    for (int j = 0; j < NJ; j++) { const int row = (w * NJ + j) * 16 + lm; bfr[j] = *(const uint4*)(sb + row * 64 + swz1(row, q) * 16); }
    // The DMA instructions of stage s + 2 go out BETWEEN the MFMAs (one after every EVERY1-th): an LDS-DMA instruction holds its
    // issuer for 100-185 cycles with nothing else in flight and ~60 among MFMAs; as a burst behind the barrier the 7 of a T = 768 stage
    // were most of the stage (phase stamps, EEGLDM_ATTN_STAMPS=1: 2 300 cycles per stage for 384 cycles of MFMA work per wave)
    constexpr int EVERY1 = (4 * NJ) / PER1;
    static_assert(EVERY1 >= 1, "every DMA piece needs an MFMA slot");
    const bool more = s + 2 < nst;
    const unsigned base2 = lds0 + ((s + 2) % 3) * STG;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        mma16<T16>(bfr[j], af[i], acc[i][j]);     // swapped: acc[i][j][r] = S[i*16+lm][(w*NJ+j)*16 + q*4 + r]
        const int m = i * NJ + j;              // compile-time after unrolling
        if ((m + 1) % EVERY1 == 0 && (m + 1) / EVERY1 - 1 < PER1) {
          const int k = (m + 1) / EVERY1 - 1;
          if (more) {
            if (k == 0) { if (a_mine) dma16a(asrc + (s + 2) * 32, base2 + wv * 1024); }
            else dma16a(bsrc[k - 1] + (s + 2) * 32, base2 + A_BYTES + (wv + NWV * (k - 1)) * 1024);
          }
        }
      }
  }
  __syncthreads();                                 // staging area free
  ATTN_STAMP(1);

  // long variant: unit u = (32-key slice ks = u / nnc, 256-column pass nc = u % nnc) -- a score slice serves all passes back to back
  auto issue2l = [&](int u, int buf) __attribute__((always_inline)) {
    const int ks = u / nnc, nc = u - ks * nnc;
    const unsigned base = lds0 + buf * 16384;
#pragma unroll
    for (int i = 0; i < 2; i++) {                  // 16 wave-instructions per tile over 8 waves
      const int ii = wv + 8 * i;
      const int c = ii * 64 + lane, krow = c >> 5, cs = c & 31;
      const int seg = trswz256(krow, cs * 16) >> 4;
      dma16a(B2 + (long)(ks * 32 + krow) * p.ldb2 + nc * 256 + seg * 8, base + ii * 1024);
    }
  };
  if constexpr (NCW == 8) {
#pragma unroll
    for (int u = 0; u < D2 - NC2T; u++) if (u < nu) issue2l(u, u);    // D2 / NC2 - 1 key slices: land while the row operation runs
  } else {
    issue_tile(B2, ld0, 0, 0);                       // land while the row operation runs
    if (nu > 1) issue_tile(B2, ld0, 1, 1);
    if (DO2 == 4 && nu > 2) issue_tile(B2, ld0, 2, 2);
  }
  uint2 pk[NCW == 8 ? 4 : 1][NCW == 8 ? NJ : 1];    // long variant: this wave's scores / score gradients as packed bf16

  // ---------------- row operation on the score tile ----------------
  auto row_reduce = [&](float v[4], const bool is_max) __attribute__((always_inline)) {
    // v[i]: this lane's partial for row i*16+lm -> full row value over the 4 lane groups and the 4 waves
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float o1 = __shfl_xor(v[i], 16, 64); v[i] = is_max ? fmaxf(v[i], o1) : v[i] + o1;
      float o2 = __shfl_xor(v[i], 32, 64); v[i] = is_max ? fmaxf(v[i], o2) : v[i] + o2;
    }
    __syncthreads();                               // previous use of `red` is over
    if (q == 0) {
#pragma unroll
      for (int i = 0; i < 4; i++) red[(mq + i * 16 + lm) * NCW + w] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float4 t = *(const float4*)(red + (mq + i * 16 + lm) * NCW);
      v[i] = is_max ? fmaxf(fmaxf(t.x, t.y), fmaxf(t.z, t.w)) : (t.x + t.y) + (t.z + t.w);
      if constexpr (NCW == 8) {
        const float4 t2 = *(const float4*)(red + (mq + i * 16 + lm) * NCW + 4);
        const float v2 = is_max ? fmaxf(fmaxf(t2.x, t2.y), fmaxf(t2.z, t2.w)) : (t2.x + t2.y) + (t2.z + t2.w);
        v[i] = is_max ? fmaxf(v[i], v2) : v[i] + v2;
      }
    }
  };
  if (MODE == 0) {
    float mx[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float m = -3.0e38f;
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) { acc[i][j][r] *= p.alpha; m = fmaxf(m, acc[i][j][r]); }
      mx[i] = m;
    }
    row_reduce(mx, true);
    float sum[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) { const float e = __expf(acc[i][j][r] - mx[i]); acc[i][j][r] = e; s += e; }
      sum[i] = s;
    }
    row_reduce(sum, false);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const float inv = 1.0f / sum[i];
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        uint2 o; o.x = pack16x2<T16>(acc[i][j][0] * inv, acc[i][j][1] * inv); o.y = pack16x2<T16>(acc[i][j][2] * inv, acc[i][j][3] * inv);
        const int col = (w * NJ + j) * 16 + q * 4;
        if constexpr (NCW == 8) pk[i][j] = o; else *(uint2*)(pt + (mq + i * 16 + lm) * PP + col * 2) = o;
        *(uint2*)(p.P + (long)b * p.sP + (long)(m0 + mq + i * 16 + lm) * T + col) = o;
      }
    }
  } else {
    float dl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const float p0 = w16_lo<T16>(pr[i][j].x), p1 = w16_hi<T16>(pr[i][j].x);
        const float p2 = w16_lo<T16>(pr[i][j].y), p3 = w16_hi<T16>(pr[i][j].y);
        s += acc[i][j][0] * p0 + acc[i][j][1] * p1 + acc[i][j][2] * p2 + acc[i][j][3] * p3;
      }
      dl[i] = s;
    }
    row_reduce(dl, false);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NJ; j++) {
        const float p0 = w16_lo<T16>(pr[i][j].x), p1 = w16_hi<T16>(pr[i][j].x);
        const float p2 = w16_lo<T16>(pr[i][j].y), p3 = w16_hi<T16>(pr[i][j].y);
        uint2 o;
        o.x = pack16x2<T16>(p.alpha * p0 * (acc[i][j][0] - dl[i]), p.alpha * p1 * (acc[i][j][1] - dl[i]));
        o.y = pack16x2<T16>(p.alpha * p2 * (acc[i][j][2] - dl[i]), p.alpha * p3 * (acc[i][j][3] - dl[i]));
        const int col = (w * NJ + j) * 16 + q * 4;
        if constexpr (NCW == 8) pk[i][j] = o; else *(uint2*)(pt + (mq + i * 16 + lm) * PP + col * 2) = o;
        *(uint2*)(p.dS + (long)b * p.sdS + (long)(m0 + mq + i * 16 + lm) * T + col) = o;
      }
  }

  ATTN_STAMP(2);
  if constexpr (NCW == 8) {
    // ---------------- product 2, long variant: out[64][C] = scores[64][T] . B2[T][C], C = 256 * NC2 ----------------
    // Wave w: rows (w >> 2) * 32 .. + 32, columns (w & 3) * 64 .. + 64 of every 256-column pass.  Slice ks (keys 32 ks .. + 32) is the
    // fragment pair j = 2 (ks % 3), + 1 of column wave ks / 3, which stores it ([64 rows][32 keys], 64-byte rows: a fragment read is
    // one linear KB) into slot ks & 1 one slice ahead; the per-unit barrier orders the hand-over.
    static_assert(NJ == 6 && NQ == 1, "long variant: T = 768, one 64-row query tile per block");
    constexpr int NC2 = NC2T;
    const unsigned wc = w & 3; const int mr = (int)(w >> 2) * 32;
    f32x4 o2[NC2][2][4];
#pragma unroll
    for (int c = 0; c < NC2; c++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) o2[c][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto put_slice = [&](const int jn, const int slot) __attribute__((always_inline)) {     // jn: compile-time after unrolling
      char* sl = sm + SL_OFF + slot * 4096;
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int jj = 0; jj < 2; jj++) *(uint2*)(sl + (i * 16 + lm) * 64 + jj * 32 + q * 8) = pk[i][jn + jj];
    };
    if (w == 0) put_slice(0, 0);
    // One barrier per key slice: its NC2 tiles are waited for together, the score fragments are read once for all passes, and the
    // 2 * NC2 DMA instructions that refill the ring (slice ks + DS - 1 into the buffers slice ks - 1 was read from before this barrier)
    // go out between the MFMAs.  (One barrier per TILE, DMA burst first: 1 330 cycles per 16 KB tile and 8 MFMAs per wave.)
    constexpr int NKS = T / 32, DS = D2 / NC2;      // slices in the ring: one being read, DS - 1 in flight
    for (int wo = 0; wo < NCW; wo++) {
#pragma unroll
      for (int kk = 0; kk < 3; kk++) {
        const int ks = wo * 3 + kk;
        // slice ks landed: at most `ahead` newer slices (2 * NC2 DMA instructions per wave each) stay in flight; the very first wait also
        // retires the P / dS stores of the row operation (stores and loads share the counter but retire independently)
        const int ahead = min(NKS - 1 - ks, DS - 2), left = ahead * 2 * NC2;
        if (ks == 0 || left <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (left == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else if (left == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (left == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (left == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (left == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __syncthreads();
        if (ks + 1 < NKS) {                          // next slice, one ahead (slot (ks + 1) & 1 was last read before this barrier)
          const unsigned owner = (kk == 2) ? wo + 1 : wo;
          if (w == owner) put_slice(2 * ((kk + 1) % 3), (ks + 1) & 1);
        }
        const char* sl = sm + SL_OFF + (ks & 1) * 4096;
        uint4 af[2], bfr[NC2][4];
#pragma unroll
        for (int i = 0; i < 2; i++) af[i] = *(const uint4*)(sl + (mr + i * 16 + lm) * 64 + q * 16);
#pragma unroll
        for (int nc = 0; nc < NC2; nc++) {
          const char* sv = sm + ((ks * NC2 + nc) % D2) * 16384;
#pragma unroll
          for (int j = 0; j < 4; j++) bfr[nc][j] = read_tr256(sv, wc * 64 + j * 16, lm, q);
        }
        const int kn = ks + DS - 1;                  // slice whose tiles are issued now
        const bool more = kn < NKS;
#pragma unroll
        for (int nc = 0; nc < NC2; nc++)
#pragma unroll
          for (int i = 0; i < 2; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
              mma16<T16>(bfr[nc][j], af[i], o2[nc][i][j]);
              const int m = nc * 8 + i * 4 + j;       // compile-time after unrolling: DMA piece m / 4 after every fourth MFMA
              if (m % 4 == 3 && more) {
                const int pc = m / 4, nn = pc >> 1, ii = (int)wv + 8 * (pc & 1);
                const int c = ii * 64 + lane, krow = c >> 5, cs = c & 31;
                const int seg = trswz256(krow, cs * 16) >> 4;
                dma16a(B2 + (long)(kn * 32 + krow) * p.ldb2 + nn * 256 + seg * 8, lds0 + ((kn * NC2 + nn) % D2) * 16384 + ii * 1024);
              }
            }
      }
    }
#pragma unroll
    for (int c = 0; c < NC2; c++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint2 o; o.x = pack16x2<T16>(o2[c][i][j][0], o2[c][i][j][1]); o.y = pack16x2<T16>(o2[c][i][j][2], o2[c][i][j][3]);
          *(uint2*)(p.O + (long)b * p.sO + (long)(m0 + mr + i * 16 + lm) * p.ldo + c * 256 + wc * 64 + j * 16 + q * 4) = o;
        }
    ATTN_STAMP(3);
    return;
  }
  // ---------------- passes over the score tile as ONE unit sequence (the tile ring prefetches across the pass boundary) ----------------
  // pass 0: out (forward) / dQ = dS K (backward); pass 1 (backward, whole-sample blocks, fuse_kv): dK = dS^T Q -- the score tile read
  // TRANSPOSED, tiles = rows of Q; pass 2 (fuse_kv == 2, round 5): dV = P^T dO, tiles = rows of dO.  (Round 4 tried dV as a pass BEFORE
  // product 1, while P is still in registers: a second call site of this loop cost 30 registers -- 700 bytes of spills at the 168
  // registers a 12-wave block leaves each lane.  Here the loop stays one call site: the probabilities are fetched AGAIN at the end of
  // the dK pass -- 73 KB per sample from L2, behind the output stores of that pass, which the next unit waits for anyway -- and
  // overwrite the dS tile behind the barrier of the first dV unit.)
  const int pr_count = fuse_v ? 3 : (fuse_kv ? 2 : 1);
  bool stores_pending = true;                      // the P / dS stores of the row operation
  f32x4 acc2[RF2][4];
  const int nut = pr_count * nu;
  // issue side (runs DO2 - 1 units ahead) and consume side each carry their own (source | destination, transposed?) state
  const bf16_t* isrc = B2; long ild = ld0; int iuu = 0;
  auto issue_next = [&](int buf) __attribute__((always_inline)) {
    issue_tile(isrc, ild, iuu, buf);
    if (++iuu == nu) {                                           // next pass: dK (tiles = rows of Q), then dV (tiles = rows of dO)
      iuu = 0;
      const bool first = isrc == B2;
      isrc = first ? src1 : src2; ild = first ? ld1 : ld2;
    }
  };
  iuu = (DO2 - 1 < nu) ? DO2 - 1 : nu;             // the first tiles were issued before the row operation
  if (iuu == nu) { iuu = 0; isrc = src1; ild = ld1; }
  bf16_t* cdst = dst0; bool ctr = false;
  bool refill = false;                             // the score tile is to be overwritten with P before this unit (first unit of the dV pass)
  int uu = 0;
  for (int u = 0; u < nut; u++) {
    const int nc = uu / nks, ks = uu - nc * nks;
    if (ks == 0) {
#pragma unroll
      for (int i = 0; i < RF2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // tile u landed; up to DO2 - 2 newer tiles (per2 DMAs of this wave each) may stay in flight unless stores were issued since (they share the counter)
    const int left = stores_pending ? 0 : min(nut - 1 - u, DO2 - 2) * per2;
    if (left >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (left >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (left >= 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (left == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stores_pending = false;
    __syncthreads();                               // (for u = 0 also: the score tile is complete in LDS)
    if (MODE == 1 && refill) {                     // every wave is done with dS (barrier above); the P fragments arrived with the vmcnt(0) of this unit
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) *(uint2*)(pt + (mq + i * 16 + lm) * PP + ((w * NJ + j) * 16 + q * 4) * 2) = pr[i][j];
      refill = false;
      __syncthreads();
    }
    const char* sv = sm + (u % DO2) * 16384;
    uint4 af[RF2], bfr[4];
    if (MODE == 1 && ctr) {
      // transposed pass: output rows = KEY rows mq2 .. (whole-sample block: as many as query rows), reduction over the query rows of slice ks
#pragma unroll
      for (int i = 0; i < RF2; i++) af[i] = read_tr_rows(pt, PP, ks * 32, mq2 + i * 16, lm, q);
    } else {
#pragma unroll
      for (int i = 0; i < RF2; i++) af[i] = *(const uint4*)(pt + (mq2 + i * 16 + lm) * PP + (ks * 32 + q * 8) * 2);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) bfr[j] = read_tr256(sv, wc2 * 64 + j * 16, lm, q);
#pragma unroll
    for (int i = 0; i < RF2; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) mma16<T16>(bfr[j], af[i], acc2[i][j]);
    if (u + DO2 - 1 < nut) issue_next((u + DO2 - 1) % DO2);      // behind the MFMAs: the matrix pipe works while the DMA instructions issue
    if (ks == nks - 1) {
#pragma unroll
      for (int i = 0; i < RF2; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          uint2 o; o.x = pack16x2<T16>(acc2[i][j][0], acc2[i][j][1]); o.y = pack16x2<T16>(acc2[i][j][2], acc2[i][j][3]);
          *(uint2*)(cdst + (long)b * sO + (long)(m0 + mq2 + i * 16 + lm) * ldo + nc * 256 + wc2 * 64 + j * 16 + q * 4) = o;
        }
      stores_pending = true;
    }
    if (++uu == nu) {                              // next pass
      uu = 0;
      if (MODE == 1 && ctr && fuse_v) {            // dK done -> dV: fetch this lane's probabilities again (the next unit waits for vmcnt(0): stores_pending)
        refill = true;
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < NJ; j++)
            pr[i][j] = *(const uint2*)(p.P + (long)b * p.sP + (long)(m0 + mq + i * 16 + lm) * T + (w * NJ + j) * 16 + q * 4);
      }
      cdst = ctr ? dst2 : dst1; ctr = true;
    }
  }
  ATTN_STAMP(3);
}

template <int NJ, int MODE, int NQ, int NCW = 4, int NC2 = 1, typename T16 = bf16_t>
int launch_chain_q(eegldm_ctx* ctx, const ChainArgs& a, int B) {
  constexpr int T = 64 * NJ, QR = 64 * NQ;
  constexpr int STG = QR * 64 + T * 64;
  constexpr int STAGE_AREA = (3 * STG > 49152) ? 3 * STG : 49152;
  constexpr int TILE = QR * (T * 2 + 16) + QR * NCW * 4;          // score tile + row-reduction scratch
  constexpr int LONG2 = 8 * 16384 + 2 * 4096 + QR * NCW * 4;      // long variant after product 1: tile ring, two score slices, row-reduction scratch
  constexpr int LDS = (NCW == 8) ? ((STAGE_AREA > LONG2) ? STAGE_AREA : LONG2) : STAGE_AREA + TILE;
  static_assert(LDS <= 160 * 1024, "attention tile does not fit the LDS");
  auto kern = attn_chain_kernel<NJ, MODE, NQ, NCW, NC2, T16>;
  static DevOnce attr;
  if (LDS > 48 * 1024 && attr.need(ctx->device)) HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
  ChainArgs ax = a;
  ax.xcd = (T / QR > 1 && B % 8 == 0) ? 1 : 0;
  EEG_ENV_VAR(bool, stamps, getenv("EEGLDM_ATTN_STAMPS") != nullptr);
  static unsigned long long* sbuf = nullptr;
  if (stamps && !sbuf) HIP_TRY(hipMalloc(&sbuf, 64));
  ax.stamps = stamps ? sbuf : nullptr;
  hipLaunchKernelGGL(kern, dim3(T / QR, B), dim3(64 * NCW * NQ), LDS, ctx->stream, ax);
  LAUNCH_CHECK();
  if (stamps) {       // (synchronous on purpose: a developer run)
    unsigned long long h[4]; HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipMemcpy(h, sbuf, 32, hipMemcpyDeviceToHost));
    fprintf(stderr, "attn_chain<T=%d, mode %d, NQ %d, NCW %d> block 0: product 1 %llu, row op %llu, product 2 %llu shader cycles\n", T, MODE, NQ, NCW, h[1] - h[0], h[2] - h[1], h[3] - h[2]);
  }
  return 0;
}
template <int NJ, int MODE, typename T16>
int launch_chain(eegldm_ctx* ctx, const ChainArgs& a, int B) {
  // whole-sample blocks (12 waves at T = 192) when the batch alone fills the chip; 64-row blocks otherwise (more blocks)
  EEG_ENV_VAR(bool, no_whole, getenv("EEGLDM_ATTN_NO_WHOLE") != nullptr);
  if constexpr (NJ == 3) { if (!no_whole && B >= ctx->num_cu / 2) return launch_chain_q<NJ, MODE, NJ, 4, 1, T16>(ctx, a, B); }
  return launch_chain_q<NJ, MODE, 1, 4, 1, T16>(ctx, a, B);
}

}  // namespace

bool attn_chain_ok(int dtype, int T, int C, long ldq, long ldo) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_NO_FUSED_ATTENTION") != nullptr);
  EEG_ENV_VAR(bool, no_long, getenv("EEGLDM_ATTN_NO_LONG") != nullptr);      // T = 768 back to the GEMM + softmax composition
  return !off && (dtype == EEGLDM_BF16 || dtype == EEGLDM_F16) && (T == 64 || T == 128 || T == 192 || T == 256 || (T == 768 && !no_long && (C == 256 || C == 512))) && C % 256 == 0 && ldq % 8 == 0 && ldo % 8 == 0;
}

// forward: probs written, out = softmax(alpha q k^T) v
template <typename T16>
static int chain_fwd_t(eegldm_ctx* ctx, const void* qkv, long ldq, void* out, long ldo, void* probs, int B, int T, int C) {
  ChainArgs a = {};
  const bf16_t* q = (const bf16_t*)qkv;
  a.A1 = q; a.lda1 = ldq; a.sA1 = (long)T * ldq; a.B1 = q + C; a.ldb1 = ldq; a.sB1 = (long)T * ldq; a.B2 = q + 2 * C; a.ldb2 = ldq; a.sB2 = (long)T * ldq;
  a.P = (bf16_t*)probs; a.sP = (long)T * T; a.O = (bf16_t*)out; a.ldo = ldo; a.sO = (long)T * ldo; a.T = T; a.C = C; a.alpha = 1.0f / sqrtf((float)C);
  switch (T / 64) {
    case 1: return launch_chain<1, 0, T16>(ctx, a, B);
    case 2: return launch_chain<2, 0, T16>(ctx, a, B);
    case 3: return launch_chain<3, 0, T16>(ctx, a, B);
    case 12: return C == 256 ? launch_chain_q<12, 0, 1, 8, 1, T16>(ctx, a, B) : launch_chain_q<12, 0, 1, 8, 2, T16>(ctx, a, B);
    default: return launch_chain<4, 0, T16>(ctx, a, B);
  }
}
// backward part: dS (scaled) written, dq = dS k
// whole-sample blocks (T = 192, batch >= half the CUs): the backward kernel can also produce dK
bool attn_chain_bwd_fuses_kv(eegldm_ctx* ctx, int B, int T) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_ATTN_NO_FUSED_KV") != nullptr); EEG_ENV_VAR(bool, no_whole, getenv("EEGLDM_ATTN_NO_WHOLE") != nullptr);
  return !off && !no_whole && T == 192 && B >= ctx->num_cu / 2;
}
template <typename T16>
static int chain_bwd_t(eegldm_ctx* ctx, const void* qkv, long ldq, const void* probs, const void* dout, long lddo, void* dq, long lddq,
                       void* dS, int B, int T, int C, int fuse_kv) {
  ChainArgs a = {};
  const bf16_t* q = (const bf16_t*)qkv;
  a.A1 = (const bf16_t*)dout; a.lda1 = lddo; a.sA1 = (long)T * lddo; a.B1 = q + 2 * C; a.ldb1 = ldq; a.sB1 = (long)T * ldq;
  a.B2 = q + C; a.ldb2 = ldq; a.sB2 = (long)T * ldq; a.P = (bf16_t*)probs; a.sP = (long)T * T; a.dS = (bf16_t*)dS; a.sdS = (long)T * T;
  a.O = (bf16_t*)dq; a.ldo = lddq; a.sO = (long)T * lddq; a.T = T; a.C = C; a.alpha = 1.0f / sqrtf((float)C);
  if (fuse_kv) {
    EEG_CHECK(attn_chain_bwd_fuses_kv(ctx, B, T), "fused dK / dV needs whole-sample blocks");
    a.fuse_kv = fuse_kv == 2 ? 2 : 1; a.Q3 = q; a.ldq3 = ldq; a.sQ3 = (long)T * ldq; a.dK = (bf16_t*)dq + C; a.dV = (bf16_t*)dq + 2 * C;
  }
  switch (T / 64) {
    case 1: return launch_chain<1, 1, T16>(ctx, a, B);
    case 2: return launch_chain<2, 1, T16>(ctx, a, B);
    case 3: return launch_chain<3, 1, T16>(ctx, a, B);
    case 12: return C == 256 ? launch_chain_q<12, 1, 1, 8, 1, T16>(ctx, a, B) : launch_chain_q<12, 1, 1, 8, 2, T16>(ctx, a, B);
    default: return launch_chain<4, 1, T16>(ctx, a, B);
  }
}

int attn_chain_fwd(eegldm_ctx* ctx, const void* qkv, long ldq, void* out, long ldo, void* probs, int B, int T, int C, int dtype) {
  return dtype == EEGLDM_F16 ? chain_fwd_t<f16_t>(ctx, qkv, ldq, out, ldo, probs, B, T, C) : chain_fwd_t<bf16_t>(ctx, qkv, ldq, out, ldo, probs, B, T, C);
}
int attn_chain_bwd(eegldm_ctx* ctx, const void* qkv, long ldq, const void* probs, const void* dout, long lddo, void* dq, long lddq,
                   void* dS, int B, int T, int C, int fuse_kv, int dtype) {
  return dtype == EEGLDM_F16 ? chain_bwd_t<f16_t>(ctx, qkv, ldq, probs, dout, lddo, dq, lddq, dS, B, T, C, fuse_kv)
                             : chain_bwd_t<bf16_t>(ctx, qkv, ldq, probs, dout, lddo, dq, lddq, dS, B, T, C, fuse_kv);
}
