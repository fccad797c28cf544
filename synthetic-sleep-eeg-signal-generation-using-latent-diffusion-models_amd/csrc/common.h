// Shared declarations for the eegldm HIP library (gfx950 only).
#pragma once
#include <stdlib.h>
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/eegldm.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16_t;  // storage type for bf16 tensors

// ---------------------------------------------------------------- errors
void eegldm_set_error(const std::string& msg);
#define EEG_FAIL(code, ...)                                  \
  do {                                                       \
    char _b[512];                                            \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
    eegldm_set_error(std::string(__func__) + ": " + _b);     \
    return (code);                                           \
  } while (0)
#define EEG_CHECK(cond, ...) \
  do { if (!(cond)) EEG_FAIL(EEGLDM_ERR_INVALID, __VA_ARGS__); } while (0)
#define HIP_TRY(expr)                                                          \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) EEG_FAIL(EEGLDM_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(_e)); \
  } while (0)
#define EEG_TRY(expr) do { int _r = (expr); if (_r != 0) return _r; } while (0)
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())

// ---------------------------------------------------------------- developer switches (environment variables)
// Every EEGLDM_* switch is cached in a function-local static and re-read only when eegldm_debug_reload_env() has bumped the epoch:
// production pays one integer compare per use, tests A/B a fast path against its predecessor inside ONE process
// (os.environ[...] = ...; lib.eegldm_debug_reload_env()) instead of spawning an interpreter per switch.
// Threading: the epoch and the live-context counter are atomics (one host thread per context / GPU may run concurrently); the cached value
// itself is re-written only with the value every thread would compute.  eegldm_debug_reload_env() / eegldm_set_deterministic() may only be
// called while no call of this library is in flight on any thread (a switch flipped in the middle of a backward would be read by some
// ops and not by others).
extern std::atomic<int> g_eeg_env_epoch;
#define EEG_ENV_VAR(T, name, ...)                                                                       \
  static T name;                                                                                        \
  do { static std::atomic<int> _ep_##name{-1};                                                          \
       const int _now_##name = g_eeg_env_epoch.load(std::memory_order_acquire);                         \
       if (_ep_##name.load(std::memory_order_acquire) != _now_##name) { name = (__VA_ARGS__); _ep_##name.store(_now_##name, std::memory_order_release); } } while (0)
// EEGLDM_DETERMINISTIC=1: bit-reproducible gradients and losses run to run.  Every reduction whose order depends on timing (fp32 global atomics
// of the bias / GroupNorm / thin-conv gradients and the loss sums; the fused column sums inside the weight-gradient GEMM; split-K without
// a workspace) takes a written-partials + fixed-order-fold route instead.  Slower by a few per cent; the default stays off.
static inline bool eeg_deterministic() {
  EEG_ENV_VAR(bool, det, getenv("EEGLDM_DETERMINISTIC") != nullptr && atoi(getenv("EEGLDM_DETERMINISTIC")) != 0);
  return det;
}
extern std::atomic<int> g_eeg_live_ctx;      // contexts alive in this process (eegldm_ctx_create / _destroy): kernels that are only safe alone on a CU ask

// ---------------------------------------------------------------- context
struct WgradRec;      // one deferred weight-gradient problem (defined below, after GemmArgs)
struct GemmGroup;
struct ProfRec { int cls; double flops; hipEvent_t a, b; int M, N, K, taps, splitk; };
enum { PROF_CONV_FWD = 0, PROF_CONV_DGRAD = 1, PROF_CONV_WGRAD = 2, PROF_GEMM_NT = 3, PROF_GEMM_NN = 4, PROF_GEMM_TN = 5, PROF_NCLASS = 6 };

struct eegldm_ctx {
  int device;
  hipStream_t stream;
  bool owns_stream;
  // scratch for small reductions / flags (device)
  void* scratch;
  size_t scratch_bytes;
  int num_cu;
  void* zero_page = nullptr;   // 4 KiB of zeros: source of out-of-range LDS-DMA chunks
  // second stream for the weight-gradient GEMMs of the backward pass (nothing downstream of a layer needs its dW before
  // the optimizer): they overlap the dgrad -> GroupNorm-backward chain instead of serialising with it
  void* splitk_ws = nullptr; size_t splitk_ws_bytes = 0;   // partial tiles of split-K weight gradients (grown on demand)
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_on = false;
  // optional per-launch HIP-event profiling of the GEMM family (bench.py roofline leg)
  bool prof_on = false;
  std::vector<ProfRec> prof;
  // K-blocked copies of 3-tap conv weights, keyed by the address of the plain [tap][Cout][Cin] bf16 weight (registered by NetBase)
  std::unordered_map<const void*, const void*> kblk;
  // data-gradient copies of 3-tap conv weights, [tap][Cout / 32][Cin][32] (the reduction index of the data gradient, Cout, K-blocked): keyed like kblk
  std::unordered_map<const void*, const void*> kblk_t;
  // stride-2 convs (Cin 64 -> Cout 128, k 3) as stride-1 weight-stationary convs over paired rows (elementwise.hip s2ws_pack): the repacked
  // [3][128][128] weights of the forward / the data gradient, keyed like kblk
  std::unordered_map<const void*, const void*> s2ws_f, s2ws_d;
  // fused train steps zero ALL their loss scalars with one memset and set this: the loss entry points then skip their own 4-byte memset
  // (every tiny launch costs ~5 us of dispatch: 22 memsets were 2.5 % of the AutoencoderKL / GAN step)
  bool loss_prezeroed = false;
  bool l1_overwrite = false;      // eegldm_l1_loss writes da instead of accumulating (the caller skipped zeroing it)
  int bn_flip = 0; int bn_dirty[2] = {0, 0};   // BatchNorm sum areas alternate; each call's fold kernel re-zeroes the other one (losses.hip)
  // running-statistics updates per training-mode BatchNorm forward (1; the fused AEKL / GAN step sets 2 around the discriminator forward whose
  // activations serve both the generator loss and the fake-sample loss: the reference runs that forward twice, train_autoencoderkl.py:213,225)
  int bn_running_repeats = 1;
  double prof_bracket_ms = 0.0;   // elapsed time of an EMPTY event pair on this stream (calibrated by eegldm_prof_enable): subtracted per launch
  // deferred weight gradients (ops.hip: op_conv_wgrad records instead of launching while defer_wgrad is set; op_wgrad_flush groups the
  // records by shape and launches every group as one grouped GEMM).  Set by the UNet backward only.
  bool defer_wgrad = false;
  std::vector<struct WgradRec> wgrad_pending;
  std::vector<struct GemmGroup> grp_host; struct GemmGroup* grp_dev = nullptr; int grp_cap = 0;   // cached group tables (gemm_launch_grouped)
  int grp_slot = 0;
  // batched dgamma / dbeta folds of the one-pass GroupNorm backward (norm.hip): in the deferred mode every launch gets its own 64-slot
  // region of gn_slot_arena and ONE kernel folds all of them when the weight gradients are flushed
  struct GnFoldRec { float* slots; float* dgamma; float* dbeta; int C; };
  std::vector<GnFoldRec> gn_fold_pending, gn_fold_host;
  float* gn_slot_arena = nullptr; GnFoldRec* gn_fold_dev = nullptr; int gn_fold_count = 0;
  // deterministic mode: one private buffer for written partial sums (grown on demand, zero where the consumer expects zeros)
  float* det_buf = nullptr; size_t det_buf_bytes = 0;
};

// per-DEVICE once flag (hipFuncSetAttribute and friends are per device; a process may drive several GPUs through several contexts)
struct DevOnce { unsigned long long done = 0; bool need(int dev) { const unsigned long long b = 1ull << (dev & 63); if (done & b) return false; done |= b; return true; } };

static inline size_t dtype_size(int dt) { return dt == EEGLDM_F32 ? 4 : 2; }
// side waits for everything enqueued on the main stream so far / main waits for everything enqueued on the side stream
int ctx_fork(eegldm_ctx* c);
int ctx_join(eegldm_ctx* c);
// RAII: launches inside the scope go to the side stream (pure GEMM work only: the context scratch belongs to the main stream)
struct SideScope {
  eegldm_ctx* c; hipStream_t saved;
  explicit SideScope(eegldm_ctx* ctx) : c(ctx), saved(ctx->stream) { if (c->side_on && !c->prof_on && !c->defer_wgrad && !eeg_deterministic()) c->stream = c->side; }   // serial while kernels are being timed / in the grouped-weight-gradient mode / in the deterministic mode (its partial-sum buffer belongs to one stream)
  ~SideScope() { c->stream = saved; }
};

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }
// gfx950 has a hardware fp32 -> packed bf16 conversion (round-to-nearest-even, NaN stays NaN): one VALU op for two values
// instead of the ~10-instruction exec-masked software sequence.
// Through the compiler's own vector conversion, NOT inline assembly: hipcc's hazard recognizer does not look inside an asm statement,
// so an asm `v_cvt_pk_bf16_f32` placed right behind the MFMA that produces its operand read the accumulator before the matrix pipe
// had written it (no s_nop inserted) -- the long attention kernel's C = 256 build stored garbage for exactly that reason (round 3),
// and every other epilogue was correct only because enough instructions happened to sit in between.
typedef float eeg_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 eeg_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 eeg_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const eeg_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, eeg_bf16x2));     // one v_cvt_pk_bf16_f32 on gfx950
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, f) & 0xffffu); }
// IEEE half storage (EEGLDM_F16, round 5): a DISTINCT type so that templates can tell the two 16-bit formats apart (bf16_t is a plain
// unsigned short).  Same size and alignment: every layout / tiling decision keyed on sizeof(T) == 2 holds for both.
struct f16_t { unsigned short v; };
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  const eeg_f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, eeg_f16x2));      // round-to-nearest-even
}
// 16-bit pairs packed in a 32-bit word (element 0 in the low half), by storage type
template <typename T> __device__ __forceinline__ float w16_lo(unsigned w);
template <typename T> __device__ __forceinline__ float w16_hi(unsigned w);
template <typename T> __device__ __forceinline__ unsigned pack16x2(float lo, float hi);
template <> __device__ __forceinline__ float w16_lo<bf16_t>(unsigned w) { return __uint_as_float(w << 16); }
template <> __device__ __forceinline__ float w16_hi<bf16_t>(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
template <> __device__ __forceinline__ unsigned pack16x2<bf16_t>(float lo, float hi) { return pack_bf16x2(lo, hi); }
template <> __device__ __forceinline__ float w16_lo<f16_t>(unsigned w) { return (float)__builtin_bit_cast(eeg_f16x2, w)[0]; }
template <> __device__ __forceinline__ float w16_hi<f16_t>(unsigned w) { return (float)__builtin_bit_cast(eeg_f16x2, w)[1]; }
template <> __device__ __forceinline__ unsigned pack16x2<f16_t>(float lo, float hi) { return pack_f16x2(lo, hi); }
// (so that code under `if constexpr (sizeof(T) == 2)` also parses for T = float)
template <> __device__ __forceinline__ float w16_lo<float>(unsigned w) { return __uint_as_float(w); }
template <> __device__ __forceinline__ float w16_hi<float>(unsigned w) { return __uint_as_float(w); }
template <> __device__ __forceinline__ unsigned pack16x2<float>(float lo, float) { return __float_as_uint(lo); }
template <typename T> struct Is16 { static constexpr bool bf16 = false, f16 = false; };
template <> struct Is16<bf16_t> { static constexpr bool bf16 = true, f16 = false; };
template <> struct Is16<f16_t> { static constexpr bool bf16 = false, f16 = true; };
template <typename T> __device__ __forceinline__ float ld_f32(const T* p);
template <> __device__ __forceinline__ float ld_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float ld_f32<f16_t>(const f16_t* p) { return (float)__builtin_bit_cast(_Float16, p->v); }
template <typename T> __device__ __forceinline__ void st_f32(T* p, float v);
template <> __device__ __forceinline__ void st_f32<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st_f32<bf16_t>(bf16_t* p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void st_f32<f16_t>(f16_t* p, float v) { p->v = __builtin_bit_cast(unsigned short, (_Float16)v); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE division: these run per element in HBM-bound kernels whose
// VALU budget is only a few dozen operations per element at 8 TB/s
__device__ __forceinline__ float silu_f(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-z));
  return s * (1.0f + z * (1.0f - s));
}

// ---------------------------------------------------------------- internal GEMM interface (gemm.hip)
enum { GA_PLAIN = 0, GA_CONV = 1, GA_TR = 2 };
enum { GB_NT = 0, GB_TR = 1 };

constexpr int GEMM_MAX_GROUP = 12;
struct GemmArgs {
  int dtype;             // EEGLDM_F32 / EEGLDM_BF16 (storage type of A, B, resid, and C unless out_f32)
  int amode, bmode;
  const void* A; long lda; long sAb;            // batch stride (elements)
  const void* B; long ldb; long sBb; long sBt;  // batch / tap strides (elements)
  void* C; long ldc; long sCb; long sCt;        // sCt: per-tap output stride (wgrad-by-tap)
  long sCk;                                     // per-K-split output stride (split-K partials written to a workspace), else 0
  int M, N, K;           // per batch (and per tap); K = reduction length
  int batch;
  int taps;              // taps folded in the K loop (conv fwd/dgrad): 1 or 3
  int ztaps;             // taps spread over grid.z (wgrad): 1 or 3
  int tap_flip;          // use W[taps-1-t] for tap t (dgrad)
  // conv geometry (GA_CONV rows, or GB_TR/GA_TR row map when conv_map)
  int Lout, Lin, stride, pad_l;
  int ups;               // GA_CONV: A is a virtual zero-upsampled signal (transposed conv): 1 or 2
  int Lsrc;              // rows per sample of the real source when ups > 1
  int conv_map;          // wgrad: map K index (output row) -> input row for B
  int splitk;            // >1: partial K ranges atomically added into f32 C
  float alpha;
  const float* bias;     // [N] or null
  const float* rowvec; long ld_rowvec; int rows_per_vec;  // per-sample vector add (time embedding)
  const void* resid; long ldr;                            // residual add (dtype)
  int out_f32;           // C is float regardless of dtype
  int atomic_out;        // C += result via float atomics (requires out_f32)
  float k_skew;          // split-K with atomic output: K chunk lengths grow linearly from (1-k_skew) to (1+k_skew) of the mean (0 = equal)
  float* colsum;         // GA_TR, 16-bit operands, batch 1: colsum[m] += sum_k A[k][m] (bias gradient of a conv whose dY is A), or null
  int b_kblk;            // GB_NT: B is K-BLOCKED, [tap][K / KC][N][KC] (KC = 32 16-bit elements = 64 bytes): a stage's B tile is one contiguous run
                         //  (op_conv_fwd with a packed weight copy, see kblk_pack); 0 = plain [tap][N][K]
  const void* B_alt;     // GA_CONV + GB_TR (data gradient): the [tap][K / 32][N][32] copy of the same weight, or null; lets gemm_big.hip run the product as an NT one
  int wide_n;            // fused 3-tap weight gradient: use the 128-wide N tile (one block per CU)
  int xcd_swizzle;       // set by the launcher: XCD-aware tile order (see gemm_kernel)
  const void* zero_page; // >= 16 zero bytes in device memory (set by gemm_launch)
  // GROUPED weight gradients (round 3): `batch` problems of identical shape and leading dimensions but unrelated addresses -- the same
  // conv shape in different layers of the network -- run as ONE launch.  ngroup > 0: problem b reads A = grpA[b], B = grpB[b] (instead
  // of A + b * sAb ...), adds its fused column sums into grpCS[b], and its folded result goes to grpDst[b] (gemm_launch_grouped).
  int ngroup;
  const struct GemmGroup* grp;       // DEVICE table of the group's pointers (a by-value array indexed by the block's problem number made
                                     // hipcc spill the whole argument struct to scratch: 672 bytes per lane in 20 instantiations)
};
struct GemmGroup {      // per problem: operands, their leading dimensions (dY is often a column view of a wider concat buffer), bias-gradient and dW targets
  const void* A[GEMM_MAX_GROUP]; const void* B[GEMM_MAX_GROUP]; float* CS[GEMM_MAX_GROUP]; float* Dst[GEMM_MAX_GROUP];
  long lda[GEMM_MAX_GROUP]; long ldb[GEMM_MAX_GROUP];
};
struct WgradRec { GemmArgs a; long tiles; int kstage; float* dst; float* dbias; };
int gemm_launch(eegldm_ctx* ctx, const GemmArgs& a);
int gemm_big_try(eegldm_ctx* ctx, const GemmArgs& a);      // gemm_big.hip: 1 = launched, 0 = not its shape, < 0 = error
// 3-tap conv + 1 x 1 skip conv over a second operand in ONE launch (K extension, gemm_big.hip): 1 = launched, 0 = not eligible, < 0 = error
int gemm_big_skip_try(eegldm_ctx* ctx, const GemmArgs& conv3, const void* x2, long ldx2, const void* w2_kblk, int K2, const float* bias2);
// 3-tap forward conv with act(x * scale + shift) applied to the operand tile in LDS (round-6 prototype, gemm_big.hip): 1 = launched, 0 = not eligible
int gemm_big_xf_try(eegldm_ctx* ctx, const GemmArgs& conv3, const float* scale, const float* shift, long ld, int act, float slope);
// grouped split-K weight gradient: a.ngroup problems whose pointers are in the HOST table `g` (a.batch / a.grp are set here), partial
// tiles to the context workspace, ONE batched fold into the Dst buffers.  `slot` = position of this launch in the step's flush order:
// the device copy of the table is cached per slot and re-uploaded only when its contents change (addresses repeat step after step).
int gemm_launch_grouped(eegldm_ctx* ctx, const GemmArgs& a, const GemmGroup& g, int slot);
