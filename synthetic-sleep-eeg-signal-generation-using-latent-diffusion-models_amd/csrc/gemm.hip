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
// ds_read_b64_tr_b16 (bf16) or ds_read_b32 (fp32).
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int NTHREADS = 256;

template <typename T> struct Tr;
template <> struct Tr<float> { static constexpr int KC = 16; static constexpr int EPC = 4; };
template <> struct Tr<bf16_t> { static constexpr int KC = 32; static constexpr int EPC = 8; };

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

typedef s16x4 __attribute__((address_space(3))) * lds_s16x4_ptr;

// K-strided fragment read from a [k rows][x cols] tile (row pitch in bytes).
template <typename T>
__device__ __forceinline__ uint4 read_tr(const char* tile, int pitch, int ks, int x0, int lm, int q);
template <> __device__ __forceinline__ uint4 read_tr<float>(const char* tile, int pitch, int ks, int x0, int lm, int q) {
  const char* p = tile + (ks * 16 + 4 * q) * pitch + (x0 + lm) * 4;
  uint4 r;
  r.x = *(const unsigned*)(p);
  r.y = *(const unsigned*)(p + pitch);
  r.z = *(const unsigned*)(p + 2 * pitch);
  r.w = *(const unsigned*)(p + 3 * pitch);
  return r;
}
template <> __device__ __forceinline__ uint4 read_tr<bf16_t>(const char* tile, int pitch, int ks, int x0, int lm, int q) {
  // lane p of a 16-lane group supplies row (base + p/4), 4 columns at 4*(p%4); it receives
  // column p of the 4x16 block, rows base..base+3 (verified: tools/probes/layout_probe.hip).
  const char* p0 = tile + (ks * 32 + 8 * q + (lm >> 2)) * pitch + (x0 + 4 * (lm & 3)) * 2;
  s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0));
  s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p0 + 4 * pitch));
  uint4 r;
  r.x = (unsigned)(unsigned short)lo[0] | ((unsigned)(unsigned short)lo[1] << 16);
  r.y = (unsigned)(unsigned short)lo[2] | ((unsigned)(unsigned short)lo[3] << 16);
  r.z = (unsigned)(unsigned short)hi[0] | ((unsigned)(unsigned short)hi[1] << 16);
  r.w = (unsigned)(unsigned short)hi[2] | ((unsigned)(unsigned short)hi[3] << 16);
  return r;
}

template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int BN, int STRIDE>
struct Cfg {
  static constexpr int KC = Tr<T>::KC;
  static constexpr int EPC = Tr<T>::EPC;
  static constexpr int KSTAGE = KSUB * KC;                       // K elements per stage
  static constexpr int SEGS = 4 * KSUB;                          // 16B chunks per NT row
  static constexpr int PITCH_NT = KSUB * 64 + 16;                // bytes
  static constexpr int A_ROWS_NT = (AMODE == GA_CONV) ? (BM * STRIDE + TAPS - 1) : BM;
  static constexpr int PITCH_A_TR = BM * (int)sizeof(T) + 16;
  static constexpr int PITCH_B_TR = BN * (int)sizeof(T) + 16;
  static constexpr int A_BYTES = (AMODE == GA_TR) ? KSTAGE * PITCH_A_TR : A_ROWS_NT * PITCH_NT;
  static constexpr int B_TILE_BYTES = (BMODE == GB_TR) ? KSTAGE * PITCH_B_TR : BN * PITCH_NT;
  static constexpr int B_BYTES = TAPS * B_TILE_BYTES;
  static constexpr int A_CHUNKS = (AMODE == GA_TR) ? KSTAGE * (BM / EPC) : A_ROWS_NT * SEGS;
  static constexpr int B_CHUNKS = TAPS * ((BMODE == GB_TR) ? KSTAGE * (BN / EPC) : BN * SEGS);
  static constexpr int CA = (A_CHUNKS + NTHREADS - 1) / NTHREADS;
  static constexpr int CB = (B_CHUNKS + NTHREADS - 1) / NTHREADS;
  static constexpr int FN = BN / 32;                              // 16-wide fragments per wave along N
  static constexpr int LDS_BYTES = A_BYTES + B_BYTES;
};

template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int BN, int STRIDE>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(const GemmArgs p) {
  using C = Cfg<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE>;
  constexpr int FN = C::FN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* smA = smem;
  char* smB = smem + C::A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  const int n0 = blockIdx.x * BN;
  const int m0 = blockIdx.y * BM;
  int z = blockIdx.z;
  const int ksplit = z % p.splitk; z /= p.splitk;
  const int tz = z % p.ztaps;      z /= p.ztaps;
  const int bz = z;

  const T* __restrict__ Ag = (const T*)p.A + (long)bz * p.sAb;
  const T* __restrict__ Bg = (const T*)p.B + (long)bz * p.sBb;

  // K range of this block
  int kbeg = 0, kend = p.K;
  if (p.splitk > 1) {
    int per = (p.K + p.splitk - 1) / p.splitk;
    per = (per + C::KSTAGE - 1) / C::KSTAGE * C::KSTAGE;
    kbeg = ksplit * per;
    kend = min(p.K, kbeg + per);
    if (kbeg >= kend) return;
  }
  const int nstages = (kend - kbeg + C::KSTAGE - 1) / C::KSTAGE;

  uint4 ra[C::CA], rb[C::CB];
  const uint4 zero4 = make_uint4(0, 0, 0, 0);

  // ---- staging: global -> registers ------------------------------------------------
  auto load_stage = [&](int s) {
    const int k0 = kbeg + s * C::KSTAGE;
#pragma unroll
    for (int i = 0; i < C::CA; i++) {
      const int c = tid + i * NTHREADS;
      uint4 v = zero4;
      if (c < C::A_CHUNKS) {
        if constexpr (AMODE == GA_PLAIN) {
          const int row = c / C::SEGS, seg = c % C::SEGS;
          const int m = m0 + row, k = k0 + seg * C::EPC;
          if (m < p.M && k < kend) v = *(const uint4*)(Ag + (long)m * p.lda + k);
        } else if constexpr (AMODE == GA_CONV) {
          const int row = c / C::SEGS, seg = c % C::SEGS;
          const long fr = (long)m0 * STRIDE - p.pad_l + row;   // flattened virtual input row
          const int k = k0 + seg * C::EPC;
          if (fr >= 0 && fr < (long)p.M * STRIDE && k < kend) {
            if (p.ups == 1) {
              v = *(const uint4*)(Ag + fr * p.lda + k);
            } else {
              const long b = fr / p.Lin; const int vv = (int)(fr - b * p.Lin);
              if ((vv % p.ups) == 0 && (vv / p.ups) < p.Lsrc)
                v = *(const uint4*)(Ag + (b * p.Lsrc + vv / p.ups) * p.lda + k);
            }
          }
        } else {  // GA_TR: source [K][M], M contiguous
          constexpr int RC = BM / C::EPC;
          const int krow = c / RC, seg = c % RC;
          const int k = k0 + krow, m = m0 + seg * C::EPC;
          if (k < kend && m < p.M) v = *(const uint4*)(Ag + (long)k * p.lda + m);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < C::CB; i++) {
      const int c = tid + i * NTHREADS;
      uint4 v = zero4;
      if (c < C::B_CHUNKS) {
        if constexpr (BMODE == GB_NT) {
          const int tap = c / (BN * C::SEGS), r = c % (BN * C::SEGS);
          const int n = n0 + r / C::SEGS, k = k0 + (r % C::SEGS) * C::EPC;
          const int tw = p.tap_flip ? (TAPS - 1 - tap) : tap;
          if (n < p.N && k < kend) v = *(const uint4*)(Bg + (long)tw * p.sBt + (long)n * p.ldb + k);
        } else {  // GB_TR: source [K][N], N contiguous
          constexpr int RC = BN / C::EPC;
          const int tap = c / (C::KSTAGE * RC), r = c % (C::KSTAGE * RC);
          const int krow = r / RC, seg = r % RC;
          const int k = k0 + krow, n = n0 + seg * C::EPC;
          const int tw = p.tap_flip ? (TAPS - 1 - tap) : tap;
          if (k < kend && n < p.N) {
            if (p.conv_map) {   // wgrad: K index = output row -> input row of tap tz
              const int bs = k / p.Lout, lo = k - bs * p.Lout;
              const int vv = lo * p.stride + tz - p.pad_l;
              if (vv >= 0 && vv < p.Lin) v = *(const uint4*)(Bg + ((long)bs * p.Lin + vv) * p.ldb + n);
            } else {
              v = *(const uint4*)(Bg + (long)tw * p.sBt + (long)k * p.ldb + n);
            }
          }
        }
      }
      rb[i] = v;
    }
  };
  // ---- staging: registers -> LDS ---------------------------------------------------
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < C::CA; i++) {
      const int c = tid + i * NTHREADS;
      if (c < C::A_CHUNKS) {
        if constexpr (AMODE == GA_TR) {
          constexpr int RC = BM / C::EPC;
          *(uint4*)(smA + (c / RC) * C::PITCH_A_TR + (c % RC) * 16) = ra[i];
        } else {
          *(uint4*)(smA + (c / C::SEGS) * C::PITCH_NT + (c % C::SEGS) * 16) = ra[i];
        }
      }
    }
#pragma unroll
    for (int i = 0; i < C::CB; i++) {
      const int c = tid + i * NTHREADS;
      if (c < C::B_CHUNKS) {
        if constexpr (BMODE == GB_TR) {
          constexpr int RC = BN / C::EPC;
          const int tap = c / (C::KSTAGE * RC), r = c % (C::KSTAGE * RC);
          *(uint4*)(smB + tap * C::B_TILE_BYTES + (r / RC) * C::PITCH_B_TR + (r % RC) * 16) = rb[i];
        } else {
          const int tap = c / (BN * C::SEGS), r = c % (BN * C::SEGS);
          *(uint4*)(smB + tap * C::B_TILE_BYTES + (r / C::SEGS) * C::PITCH_NT + (r % C::SEGS) * 16) = rb[i];
        }
      }
    }
  };

  // ---- per-lane conv validity masks -------------------------------------------------
  // bit (i*TAPS + t) set -> A fragment i, tap t reads a row of ANOTHER sample: use zeros.
  unsigned zmask = 0;
  if constexpr (AMODE == GA_CONV) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int m = m0 + wm * 64 + i * 16 + lm;
      const int lo = m % p.Lout;
#pragma unroll
      for (int t = 0; t < TAPS; t++) {
        const int vv = lo * STRIDE + t - p.pad_l;
        if (vv < 0 || vv >= p.Lin) zmask |= 1u << (i * TAPS + t);
      }
    }
  }

  f32x4 acc[4][FN];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  load_stage(0);
  for (int s = 0; s < nstages; s++) {
    store_stage();
    __syncthreads();
    if (s + 1 < nstages) load_stage(s + 1);
#pragma unroll
    for (int t = 0; t < TAPS; t++) {
#pragma unroll
      for (int ks = 0; ks < KSUB; ks++) {
        uint4 af[4], bf[FN];
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if constexpr (AMODE == GA_TR) {
            af[i] = read_tr<T>(smA, C::PITCH_A_TR, ks, wm * 64 + i * 16, lm, q);
          } else {
            int row = wm * 64 + i * 16 + lm;
            if constexpr (AMODE == GA_CONV) row = row * STRIDE + t;
            af[i] = *(const uint4*)(smA + row * C::PITCH_NT + ks * 64 + q * 16);
            if constexpr (AMODE == GA_CONV) {
              if (zmask & (1u << (i * TAPS + t))) af[i] = zero4;
            }
          }
        }
#pragma unroll
        for (int j = 0; j < FN; j++) {
          if constexpr (BMODE == GB_TR) {
            bf[j] = read_tr<T>(smB + t * C::B_TILE_BYTES, C::PITCH_B_TR, ks, wn * (BN / 2) + j * 16, lm, q);
          } else {
            const int row = wn * (BN / 2) + j * 16 + lm;
            bf[j] = *(const uint4*)(smB + t * C::B_TILE_BYTES + row * C::PITCH_NT + ks * 64 + q * 16);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < FN; j++) mma<T>(af[i], bf[j], acc[i][j]);
      }
    }
    __syncthreads();
  }

  // ---- epilogue ---------------------------------------------------------------------
  char* Cb = (char*)p.C;
  const long cbase = (long)bz * p.sCb + (long)tz * p.sCt;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int m = m0 + wm * 64 + i * 16 + q * 4 + r;
      if (m >= p.M) continue;
      const float* rv = p.rowvec ? p.rowvec + (long)(m / p.rows_per_vec) * p.ld_rowvec : nullptr;
#pragma unroll
      for (int j = 0; j < FN; j++) {
        const int n = n0 + wn * (BN / 2) + j * 16 + lm;
        if (n >= p.N) continue;
        float v = acc[i][j][r] * p.alpha;
        if (p.bias) v += p.bias[n];
        if (rv) v += rv[n];
        if (p.resid) v += ld_f32((const T*)p.resid + (long)m * p.ldr + n);
        const long off = cbase + (long)m * p.ldc + n;
        if (p.atomic_out) {
          atomicAdd((float*)Cb + off, v);
        } else if (p.out_f32) {
          ((float*)Cb)[off] = v;
        } else {
          st_f32((T*)Cb + off, v);
        }
      }
    }
  }
}

template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int BN, int STRIDE>
int launch_t(eegldm_ctx* ctx, const GemmArgs& a) {
  using C = Cfg<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE>;
  auto kern = gemm_kernel<T, AMODE, BMODE, TAPS, KSUB, BN, STRIDE>;
  static bool attr_set = false;
  if (!attr_set && C::LDS_BYTES > 64 * 1024) {
    HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_set = true;
  }
  dim3 grid((a.N + BN - 1) / BN, (a.M + BM - 1) / BM, a.batch * a.ztaps * a.splitk);
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), C::LDS_BYTES, ctx->stream, a);
  LAUNCH_CHECK();
  return 0;
}

template <typename T, int AMODE, int BMODE, int TAPS, int KSUB, int STRIDE>
int launch_bn(eegldm_ctx* ctx, const GemmArgs& a) {
  if (a.N > 64) return launch_t<T, AMODE, BMODE, TAPS, KSUB, 128, STRIDE>(ctx, a);
  if (a.N > 32) return launch_t<T, AMODE, BMODE, TAPS, KSUB, 64, STRIDE>(ctx, a);
  return launch_t<T, AMODE, BMODE, TAPS, KSUB, 32, STRIDE>(ctx, a);
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
  EEG_CHECK(a.taps == 1, "taps>1 needs conv A mode");
  if (a.amode == GA_PLAIN && a.bmode == GB_NT) return launch_bn<T, GA_PLAIN, GB_NT, 1, 2, 1>(ctx, a);
  if (a.amode == GA_PLAIN && a.bmode == GB_TR) return launch_bn<T, GA_PLAIN, GB_TR, 1, 2, 1>(ctx, a);
  if (a.amode == GA_TR && a.bmode == GB_TR) return launch_bn<T, GA_TR, GB_TR, 1, 2, 1>(ctx, a);
  EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "gemm: amode=%d bmode=%d", a.amode, a.bmode);
}

}  // namespace

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
  if (a.splitk > 1) a.atomic_out = 1;
  ProfRec rec; bool prof = ctx->prof_on;
  if (prof) {
    rec.cls = a.amode == GA_CONV ? (a.bmode == GB_NT ? PROF_CONV_FWD : PROF_CONV_DGRAD)
              : (a.amode == GA_TR ? (a.conv_map ? PROF_CONV_WGRAD : PROF_GEMM_TN) : (a.bmode == GB_NT ? PROF_GEMM_NT : PROF_GEMM_NN));
    // algorithmic work: the transposed (strided) dgrad multiplies a half-zero virtual signal; only the real taps count
    rec.flops = 2.0 * a.M * a.N * (double)a.K * a.taps * a.ztaps * a.batch / (a.ups > 1 ? a.ups : 1);
    HIP_TRY(hipEventCreate(&rec.a)); HIP_TRY(hipEventCreate(&rec.b));
    HIP_TRY(hipEventRecord(rec.a, ctx->stream));
  }
  int rc;
  if (a.dtype == EEGLDM_F32) rc = launch_modes<float>(ctx, a);
  else if (a.dtype == EEGLDM_BF16) rc = launch_modes<bf16_t>(ctx, a);
  else EEG_FAIL(EEGLDM_ERR_UNSUPPORTED, "dtype %d", a.dtype);
  if (prof) { HIP_TRY(hipEventRecord(rec.b, ctx->stream)); ctx->prof.push_back(rec); }
  return rc;
}
