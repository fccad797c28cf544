/* eegldm.h -- C ABI of libeegldm.so: MI355X (gfx950) kernels and model executors for
 * the latent-diffusion hot path of the reference repository
 * (AutoencoderKL + PatchDiscriminator train step, UNet denoiser train step, DDIM sampling).
 *
 * The reference has no FFI layer: its boundary is the PyTorch nn.Module / callable
 * protocol of its entry scripts.  Each group below names the reference interface it
 * stands behind (file:line under /root/reference).  A Python binding mirroring those
 * classes lives in the package (`eegldm/`); INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer unless the name ends in _host.  The caller owns
 *    all tensors, parameters, gradients and optimizer state; the library owns only the
 *    context workspace and the opaque handles.  No hidden host copies.
 *  - All calls enqueue on the context's HIP stream and return; a context is not
 *    thread-safe; one context + one process per GPU.
 *  - Return value 0 = ok, negative = error code; eegldm_last_error() gives the
 *    thread-local message.  No C++ exception crosses this ABI.  Shapes / dtypes are
 *    validated before any launch.
 *  - "NCL" = the reference's contiguous (batch, channel, length) tensors.
 *    "NLC" = the engine's internal layout: rows = (sample, position), channels
 *    contiguous, explicit leading dimension `ld` in elements (so channel-concats are
 *    views).  Model-level entry points take/return NCL fp32 like the reference;
 *    primitive entry points work on NLC.
 *  - Convolution weights are held packed as [K][Cout][Cin] (tap-major); eegldm_pack_*
 *    converts from/to the reference (Cout, Cin, K) layout.
 *  - Random draws (noise, eps, timesteps) are inputs, so device and oracle runs see
 *    identical values; eegldm_randn / eegldm_randint fill them on-device for perf runs.
 */
#ifndef EEGLDM_H
#define EEGLDM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 4: + eegldm_ctx_stream, eegldm_linear_bwd, eegldm_disc_feature, eegldm_usleep_*, eegldm_feature_moments (additive).
 * 6 (round 5): + EEGLDM_F16, eegldm_conv1d_skip_fwd, eegldm_conv1d_fwd_qstats, eegldm_groupnorm_fwd_qstats, eegldm_batchnorm_lrelu_*,
 *    eegldm_kl_reparam_*, eegldm_conv1d_pack_kblocked_k, eegldm_avgpool2_*, eegldm_nearest2_* (additive).
 * 7 (round 6): + eegldm_resblock_create / eegldm_attnblock_create / eegldm_block_*, eegldm_timestep_embedding (primitive-granular UNet blocks),
 *    eegldm_conv1d_fwd_gn (GroupNorm + SiLU on the conv's operand load), eegldm_conv1d_pack_stride2; REMOVED: eegldm_conv1d_fwd_qstats and
 *    eegldm_groupnorm_fwd_qstats (round 5's GroupNorm-from-producer-moments path measured no gain and was taken out, HISTORY.md).
 * 8 (round 6): eegldm_unet_cfg grows by num_head_channels, num_heads_upsample, use_scale_shift_norm, resample_layers, resample_pool_only
 *    (zero = the config_ldm.yaml behaviour); eegldm_resblock_create gains use_scale_shift_norm, eegldm_attnblock_create gains num_heads;
 *    + eegldm_ddim_step_eta, eegldm_ddpm_step_var, eegldm_unet_set_dropout, eegldm_dropout. */
#define EEGLDM_ABI_VERSION 8

/* Storage / operand type of activations and compute-copy weights (accumulation, statistics, master weights and optimizer state are
 * always fp32).  EEGLDM_F16 = IEEE half: the type the reference trains in under `autocast` (src/training/training.py:423) with its
 * GradScaler (:334,441-443); every kernel family is instantiated for it (general, big-tile, fused attention, weight-stationary and
 * few-row convs, fused frozen encoder, pipelined GroupNorm backward) and it runs at bf16 speed. */
enum { EEGLDM_F32 = 0, EEGLDM_BF16 = 1, EEGLDM_F16 = 2 };
enum {
  EEGLDM_OK = 0,
  EEGLDM_ERR_INVALID = -1,
  EEGLDM_ERR_HIP = -2,
  EEGLDM_ERR_UNSUPPORTED = -3,
  EEGLDM_ERR_NOMEM = -4
};
enum { EEGLDM_PRED_EPSILON = 0, EEGLDM_PRED_V = 1, EEGLDM_PRED_SAMPLE = 2 };

typedef struct eegldm_ctx eegldm_ctx;
typedef struct eegldm_unet eegldm_unet;
typedef struct eegldm_aekl eegldm_aekl;
typedef struct eegldm_disc eegldm_disc;
typedef struct eegldm_block eegldm_block;

/* ------------------------------------------------------------------ library / context */
int eegldm_abi_version(void);
const char* eegldm_last_error(void);
/* hip_stream: the hipStream_t to enqueue on (e.g. torch's current stream; NULL = the
 * device's default stream, which is what torch uses unless told otherwise).
 * own_stream != 0: ignore hip_stream and create a private non-blocking stream. */
int eegldm_ctx_create(int hip_device, void* hip_stream, int own_stream, eegldm_ctx** out);
/* The stream the context enqueues on: `hip_stream` as given, or the library's own non-blocking stream when own_stream != 0. */
void* eegldm_ctx_stream(const eegldm_ctx*);
int eegldm_ctx_destroy(eegldm_ctx* ctx);
int eegldm_ctx_sync(eegldm_ctx* ctx);
/* HIP-event timing on the context's stream (bench.py measures kernels with these, since
 * torch.cuda.Event only sees torch's current stream). */
int eegldm_timer_start(eegldm_ctx* ctx);
int eegldm_timer_stop_ms(eegldm_ctx* ctx, float* ms_host);
/* Per-launch profiling of the MFMA GEMM / implicit-conv family: while enabled, every launch
 * is bracketed by HIP events on the context's stream.  eegldm_prof_summary returns, for one
 * kernel class (0 conv fwd, 1 conv dgrad, 2 conv wgrad, 3 gemm NT, 4 gemm NN, 5 gemm TN),
 * the summed algorithmic FLOPs (2*M*N*K*taps), summed kernel time and launch count since
 * eegldm_prof_enable(ctx, 1).  Host outputs.  Enabling also measures the elapsed time of an EMPTY
 * event pair on the stream (median of 33); the summary subtracts it once per launch so that the
 * durations are the kernels' own (they then agree with a rocprofv3 kernel trace of the same run);
 * eegldm_prof_bracket_overhead_ms reports that calibration value. */
int eegldm_prof_enable(eegldm_ctx* ctx, int on);
int eegldm_prof_summary(eegldm_ctx* ctx, int kernel_class, double* flops_host, double* ms_host, int* launches_host);
int eegldm_prof_bracket_overhead_ms(eegldm_ctx* ctx, double* ms_host);
/* developer aid: CSV of every profiled launch (class,M,N,K,taps,splitk,ms,gflop) */
int eegldm_prof_dump(eegldm_ctx* ctx, const char* path_host);
/* Developer switches (EEGLDM_* environment variables, README "Developer switches") are cached on first use;
 * eegldm_debug_reload_env() makes every one of them be read again on its next use (returns the new epoch).  Tests use it to
 * A/B an execution path against its predecessor inside one process.  Models / contexts created BEFORE the call keep whatever
 * they built from the old values (weight copies, streams).  No reference counterpart. */
int eegldm_debug_reload_env(void);
/* EEGLDM_DETERMINISTIC=1 (environment variable, read like the developer switches; eegldm.set_deterministic() in the Python mirror):
 * bit-reproducible losses, parameter gradients and optimiser steps run to run.  Every order-dependent reduction -- fp32 atomics of
 * the bias / GroupNorm / thin-conv gradients and of the loss sums, fused column sums inside the weight-gradient GEMM, split-K without
 * a workspace -- takes a written-partials + fixed-order-fold route; forward results are unchanged.  Reference counterpart:
 * torch.use_deterministic_algorithms(True) around src/train_ldm.py / src/train_autoencoderkl.py (the reference itself does not set it). */

/* ------------------------------------------------------------------ layout / packing */
int eegldm_ncl_to_nlc(eegldm_ctx*, const float* src_ncl, void* dst_nlc, long ld_dst, int B, int C, int L, int dst_dtype);
int eegldm_nlc_to_ncl(eegldm_ctx*, const void* src_nlc, long ld_src, float* dst_ncl, int B, int C, int L, int src_dtype);
/* (Cout,Cin,K) fp32  <->  [K][Cout][Cin] fp32 */
int eegldm_pack_conv_weight(eegldm_ctx*, const float* w_ref, float* w_packed, int Cout, int Cin, int K);
int eegldm_unpack_conv_weight(eegldm_ctx*, const float* w_packed, float* w_ref, int Cout, int Cin, int K);
/* fp32 -> compute dtype copy (n elements) */
int eegldm_cast(eegldm_ctx*, const float* src, void* dst, long n, int dst_dtype);

/* K-blocked copy of a 3-tap conv weight, [3][Cin/32][Cout][32] (16-bit dtypes, Cin % 32 == 0): writes it to w_kblocked
 * (same size as w) and registers it with the context, after which eegldm_conv1d_fwd(.., w, ..) reads its weight tiles from
 * the copy (one contiguous run per K stage; results are bit-identical).  The model executors do this for their own weights;
 * a caller of the primitives who updates w must pack again.  eegldm_conv1d_forget_kblocked removes the registration
 * (before freeing either buffer).  No reference counterpart: layout plumbing behind nn.Conv1d (unet.py:263). */
int eegldm_conv1d_pack_kblocked(eegldm_ctx*, const void* w, void* w_kblocked, int Cout, int Cin, int dtype);
/* the same for a weight of K taps (K = 1: [1][Cin/32][Cout][32], the copy eegldm_conv1d_skip_fwd reads its 1 x 1 weight from) */
int eegldm_conv1d_pack_kblocked_k(eegldm_ctx*, const void* w, void* w_kblocked, int Cout, int Cin, int K, int dtype);
int eegldm_conv1d_forget_kblocked(eegldm_ctx*, const void* w);
/* Stride-2 Conv1d(64 -> 128, k 3, padding 1) -- the PatchDiscriminator's second layer (config/config_aekl_eeg.yaml:30-40) -- as a stride-1
 * conv of the weight-stationary kernel over PAIRS of rows: fills w_fwd / w_dgrad ([3][128][128] elements each, caller-owned) from the packed
 * weight w ([3][128][64]) and registers them; eegldm_conv1d_fwd / _bwd_data then take that route for contiguous operands (ld = channels),
 * whole 64-row tiles per sample and >= 16 384 output rows.  eegldm_conv1d_forget_kblocked drops the registration. */
int eegldm_conv1d_pack_stride2(eegldm_ctx*, const void* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int dtype);
/* Data-gradient copy of a 3-tap conv weight, [3][Cout/32][Cin][32] (16-bit dtypes, Cout % 32 == 0): written to w_dgrad (same size as w)
 * and registered with the context, after which eegldm_conv1d_bwd_data(.., w, ..) may run the input gradient as a plain NT product on
 * the 192 x 256 tile (shapes with Cin % 256 == 0, Cout % 64 == 0, L % 192 == 0; other shapes are unaffected).  The model executors do
 * this for their own weights; eegldm_conv1d_forget_kblocked removes this registration too.  No reference counterpart: layout
 * plumbing behind the backward of nn.Conv1d (unet.py:263). */
int eegldm_conv1d_pack_dgrad(eegldm_ctx*, const void* w, void* w_dgrad, int Cout, int Cin, int dtype);
/* the same for a weight of K taps, [K][Cout][Cin] -> [K][Cout/32][Cin][32] (K = 1: 1 x 1 convs; K = 3 is eegldm_conv1d_pack_dgrad) */
int eegldm_conv1d_pack_dgrad_k(eegldm_ctx*, const void* w, void* w_dgrad, int Cout, int Cin, int K, int dtype);

/* ------------------------------------------------------------------ primitives (NLC)
 * nn.Conv1d as used at unet.py:263,291,302,385,504 and inside MONAI AutoencoderKL /
 * PatchDiscriminator.  w: packed [K][Cout][Cin] in `dtype`;
 * bias / rowvec / dw / dbias: fp32.  Output length Lout = (Lin + pad_l + pad_r - K)/stride + 1.
 * rowvec (optional): per-sample vector [B][ld_rowvec] added to every position (the
 * timestep-embedding add, unet.py:316-325).  resid (optional): tensor added in the
 * epilogue (residual / skip, unet.py:327). */
int eegldm_conv1d_fwd(eegldm_ctx*, const void* x, long ldx, const void* w, const float* bias,
                      void* y, long ldy, int B, int Lin, int Cin, int Cout, int K, int stride,
                      int pad_l, int pad_r, const float* rowvec, long ld_rowvec,
                      const void* resid, long ld_resid, int dtype);
/* The ResBlock tail `self.skip_connection(x) + h` with h = out_layers' conv (unet.py:302,327) as one operator:
 *   y = conv1d(x; w [3][Cout][Cin], pad 1) + bias + conv1d(x2; w2 [1][Cout][Cin2]) + bias2 (+ rowvec per sample)
 * With 16-bit operands, Cout % 256 == 0, B*L and L multiples of 192 and K-blocked copies of BOTH weights registered
 * (eegldm_conv1d_pack_kblocked / _k) it is ONE launch -- the 1 x 1 conv runs as further reduction stages of the 3-tap kernel: one
 * fp32 accumulator, one rounding, no intermediate tensor; otherwise the two convs are launched one after the other (y is then
 * rounded twice in 16-bit storage).  y must not alias x or x2. */
int eegldm_conv1d_skip_fwd(eegldm_ctx*, const void* x, long ldx, const void* w, const float* bias,
                           const void* x2, long ldx2, const void* w2, const float* bias2, void* y, long ldy,
                           int B, int L, int Cin, int Cin2, int Cout, const float* rowvec, long ld_rowvec, int dtype);
int eegldm_conv1d_bwd_data(eegldm_ctx*, const void* dy, long lddy, const void* w, void* dx, long lddx,
                           int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r,
                           const void* resid, long ld_resid, int dtype);
/* dw += ..., dbias += ... (fp32 accumulators; dbias may be NULL) */
int eegldm_conv1d_bwd_weight(eegldm_ctx*, const void* x, long ldx, const void* dy, long lddy,
                             float* dw, float* dbias, int B, int Lin, int Cin, int Cout, int K,
                             int stride, int pad_l, int pad_r, int dtype);
/* nn.Linear (unet.py:373-377, 277-285): y[M][N] = x[M][K] w[N][K]^T + bias; y is fp32 when out_f32.  M is free; N must be a
 * multiple of 4 (vector epilogue) and K a whole number of 16-byte chunks (8 bf16 / 4 fp32 elements): anything else is refused. */
int eegldm_linear_fwd(eegldm_ctx*, const void* x, long ldx, const void* w, const float* bias, void* y, long ldy,
                      int M, int N, int K, int dtype, int out_f32);
/* its backward (what autograd derives for nn.Linear): dx[M][K] = dy[M][N] w[N][K] (skipped when dx is NULL; fp32 when dx_f32),
 * dw[N][K] += dy^T x and dbias[N] += column sums of dy (fp32 accumulators; either may be NULL) */
int eegldm_linear_bwd(eegldm_ctx*, const void* x, long ldx, const void* w, const void* dy, long lddy, void* dx, long lddx,
                      float* dw, float* dbias, int M, int N, int K, int dtype, int dx_f32);

/* nn.GroupNorm(G, C, eps) [+ SiLU] (unet.py:71-74; MONAI norm_num_groups, eps 1e-6).
 * stats: [B][G][2] fp32 (mean, rstd), written by fwd and consumed by bwd.
 * resample: 0 none, 1 = AvgPool1d(2,2) after the activation, 2 = nearest x2 after the
 * activation (the up/down ResBlock, unet.py:308-313); then y has L/2 or 2L rows and
 * xr (optional) receives the equally resampled raw x. */
int eegldm_groupnorm_fwd(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta,
                         void* y, long ldy, float* stats, int B, int L, int C, int G, float eps,
                         int fuse_silu, int resample, void* xr, long ldxr, int dtype);
int eegldm_groupnorm_bwd(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta,
                         const float* stats, const void* dy, long lddy, void* dx, long lddx,
                         float* dgamma, float* dbeta, int B, int L, int C, int G, int fuse_silu,
                         int resample, const void* dxr, long lddxr, int dtype);

/* single-head QKV attention (unet.py:107-125): qkv [B*T][3C] (q|k|v), out [B*T][C].
 * probs: caller-provided [B][T][T] `dtype` buffer (kept for backward);
 * scratch_*: caller-provided work buffers, [B][T][T] fp32 (logits / dprobs) and
 * [B][T][T] `dtype` (dlogits). */
int eegldm_attention_fwd(eegldm_ctx*, const void* qkv, long ldqkv, void* out, long ldo, void* probs,
                         float* scratch_logits, int B, int T, int C, int dtype);
int eegldm_attention_bwd(eegldm_ctx*, const void* qkv, long ldqkv, const void* probs, const void* dout, long lddo,
                         void* dqkv, long lddqkv, float* scratch_dprobs, void* scratch_dlogits,
                         int B, int T, int C, int dtype);

/* ------------------------------------------------------------------ schedulers / losses / optimizer
 * DDPMScheduler.add_noise / get_velocity (training.py:429-436), DDIMScheduler.step
 * (sample_trials.py:163), F.mse_loss (training.py:437), torch.optim.Adam (train_ldm.py:208). */
int eegldm_add_noise(eegldm_ctx*, const float* x, const float* noise, const int64_t* t, const float* acp,
                     float* out, int B, long n_per_sample);
int eegldm_get_velocity(eegldm_ctx*, const float* x, const float* noise, const int64_t* t, const float* acp,
                        float* out, int B, long n_per_sample);
int eegldm_ddim_step(eegldm_ctx*, const float* model_out, const float* sample, float a_t, float a_prev,
                     int pred_type, int clip_sample, float* prev_sample, float* pred_x0, long n);
/* DDIMScheduler.step with eta >= 0 (the reference samples with eta = 0, sample_trials.py:163; eta > 0 is the scheduler's stochastic form):
 * sigma = eta sqrt((1 - a_prev) / (1 - a_t) (1 - a_t / a_prev)), prev = sqrt(a_prev) x0 + sqrt(1 - a_prev - sigma^2) eps + sigma noise.
 * eta == 0 is eegldm_ddim_step (noise may be NULL); eta == 1 over consecutive timesteps is the ancestral DDPM step (tests). (ABI 8) */
int eegldm_ddim_step_eta(eegldm_ctx*, const float* model_out, const float* sample, const float* noise, float a_t, float a_prev, float eta,
                         int pred_type, int clip_sample, float* prev_sample, float* pred_x0, long n);
/* DDPMScheduler.step (variance_type "fixed_small"; the ancestral sampler of util.py:241-243,261-285 and sample_trials_ddpm.py:99-102;
 * arithmetic pinned against DDPM.p_sample, /root/reference/src/models/ldm.py:311-357): x0 from the prediction type, optional clamp
 * to [-1,1], prev = c0*x0 + ct*sample + sqrt(max(var,1e-20))*noise with the posterior coefficients of (a_t, a_prev, beta_t);
 * a_prev == 1 (t == 0) adds no noise and `noise` may be NULL there.  pred_x0 nullable. */
int eegldm_ddpm_step(eegldm_ctx*, const float* model_out, const float* sample, const float* noise, float a_t, float a_prev,
                     float beta_t, int pred_type, int clip_sample, float* prev_sample, float* pred_x0, long n);
/* the same step with variance_type "fixed_large" when variance_large != 0: sigma^2 = beta_t instead of the posterior variance (ABI 8) */
int eegldm_ddpm_step_var(eegldm_ctx*, const float* model_out, const float* sample, const float* noise, float a_t, float a_prev,
                         float beta_t, int variance_large, int pred_type, int clip_sample, float* prev_sample, float* pred_x0, long n);
/* loss = mean((pred-target)^2); dpred = 2 (pred-target) / n * grad_scale (nullable) */
int eegldm_mse_loss(eegldm_ctx*, const float* pred, const float* target, float* loss, float* dpred, long n, float grad_scale);
int eegldm_adam_step(eegldm_ctx*, float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                     float beta2, float eps, int step, float grad_inv_scale);
/* found_inf[0] (device float) = 1 if any of g[0..n) is inf/nan, else 0 -- the check behind GradScaler.unscale_/step
 * (torch.cuda.amp.GradScaler at /root/reference/src/training/training.py:334,441-443). g must be 16-byte aligned. */
int eegldm_grad_check_finite(eegldm_ctx*, const float* g, long n, float* found_inf);
int eegldm_randn(eegldm_ctx*, float* out, long n, uint64_t seed, uint64_t offset);
int eegldm_randint(eegldm_ctx*, int64_t* out, long n, int64_t high, uint64_t seed, uint64_t offset);

/* ------------------------------------------------------------------ UNet denoiser
 * UNetModel(image_size, in_channels, model_channels, out_channels, num_res_blocks,
 * attention_resolutions, dropout=0, channel_mult, num_heads=1, use_scale_shift_norm=False,
 * resblock_updown=True) -- /root/reference/src/models/unet.py:330-563, config_ldm.yaml:30-43. */
typedef struct {
  int in_channels, out_channels, model_channels, num_res_blocks;
  int n_mult; int channel_mult[8];
  int n_attn; int attention_resolutions[8];
  int num_heads;           /* attention heads of the input / middle blocks (unet.py:344,146-153; 0 = 1).  QKVAttentionLegacy layout:
                            * the qkv projection's channels are [q_0 | k_0 | v_0 | q_1 | ...] (unet.py:107-125) */
  int dtype;               /* storage/compute dtype of activations: EEGLDM_F32, EEGLDM_BF16 or EEGLDM_F16 */
  /* ABI 8: the constructor branches every reference yaml leaves at the values of config_ldm.yaml; 0 = that default */
  int num_head_channels;   /* > 0: heads = channels / num_head_channels in every attention block (unet.py:149-153); <= 0: num_heads */
  int num_heads_upsample;  /* heads of the output blocks' attention (unet.py:354-355,481-487); <= 0: num_heads */
  int use_scale_shift_norm;/* != 0: h = GroupNorm(h) * (1 + scale) + shift, emb_layers 2 x out_channels wide (unet.py:279-284,318-322) */
  int resample_layers;     /* != 0: resblock_updown = False -- Downsample / Upsample layers instead of down / up ResBlocks (unet.py:462-470,493-498) */
  int resample_pool_only;  /* with resample_layers: != 0 = conv_resample False (AvgPool1d(2,2) / nearest x 2 only, unet.py:192-195,221);
                            * 0 = Conv1d(k 3, stride 2, padding 1) / nearest x 2 + Conv1d(k 3, padding 1) (unet.py:188-190,211-224) */
} eegldm_unet_cfg;

int eegldm_unet_create(eegldm_ctx*, const eegldm_unet_cfg* cfg, eegldm_unet** out);
int eegldm_unet_destroy(eegldm_unet*);
/* nn.Dropout(p) of ResBlock.out_layers (unet.py:289; every reference yaml sets 0.0): active in forwards with training != 0 only; masks come
 * from the on-device Philox stream (seed, running counter), are regenerated by the backward and never stored.  Calling it restarts the
 * counter: the same seed reproduces the same masks for the same sequence of forwards.  (ABI 8) */
int eegldm_unet_set_dropout(eegldm_unet*, float p, uint64_t seed);
/* The mask kernel itself, in place on a [rows][C] tensor (leading dimension ld, elements): x <- keep ? x / (1 - p) : 0 with keep drawn from
 * Philox(seed, offset + e / 4), e = row * C + column.  The same (seed, offset) gives the same mask: the backward applies it to the gradient. */
int eegldm_dropout(eegldm_ctx*, void* x, long ld, long rows, int C, float p, uint64_t seed, uint64_t offset, int dtype);
/* Parameter table: entry i <-> one reference state_dict key, stored at [offset, offset+numel)
 * of the flat fp32 parameter / gradient buffers.  Conv weights (ndim 3) are stored packed
 * [K][Cout][Cin]; `shape` reports the reference shape (Cout, Cin, K). */
int eegldm_unet_num_entries(const eegldm_unet*);
long eegldm_unet_num_params(const eegldm_unet*);
int eegldm_unet_entry(const eegldm_unet*, int i, char* name, int name_cap, long* offset, long* numel,
                      int* ndim, int shape[3]);
/* Bind caller-owned flat fp32 buffers (grads may be NULL for inference). */
int eegldm_unet_bind(eegldm_unet*, float* params, float* grads);
/* Optional host callback fired from eegldm_unet_backward / eegldm_ldm_train_step once the gradients in
 * [offset, offset + numel) of the flat gradient buffer (out, output_blocks, middle_block -- the tail of the layout) are
 * complete in stream order, while the input blocks' backward is still to be enqueued.  Data-parallel hosts start the
 * all-reduce of that slice there (replaces torch.nn.DataParallel's gather, train_ldm.py:190-192).  NULL disables. */
typedef void (*eegldm_grad_hook)(void* user, long offset, long numel);
int eegldm_unet_set_grad_hook(eegldm_unet*, eegldm_grad_hook fn, void* user);
/* Refresh the compute-dtype weight copies after `params` changed (no-op for fp32). */
int eegldm_unet_sync_weights(eegldm_unet*);
/* forward(x, timesteps): x, y are NCL fp32 (B, C, L); t int64 (B).  training != 0 keeps
 * activations for eegldm_unet_backward.  training == 0 (model.eval(): sampling) MAY skip them: for launches of a few hundred rows
 * (one window per call, sample_trials.py:149-163) the second GroupNorm of each ResBlock is applied inside the next conv and neither
 * its output nor its statistics are stored -- eegldm_unet_backward then fails with an error instead of using a partial tape.  The
 * reference has the same contract in another form: its sampling runs under torch.no_grad() (sample_trials.py:147). */
int eegldm_unet_forward(eegldm_unet*, const float* x, const int64_t* t, float* y, int B, int L, int training);
/* grads += d loss / d params; dx (nullable) = d loss / d x.  Gradients accumulate: zero the
 * flat gradient buffer (eegldm_fill) between steps, as optimizer.zero_grad does. */
int eegldm_unet_backward(eegldm_unet*, const float* dy, float* dx);
int eegldm_fill(eegldm_ctx*, float* p, long n, float value);

/* The body of train_epoch_ldm (training.py:419-443) after the frozen encoder: add_noise,
 * UNet forward, MSE vs noise / velocity, backward.  loss: device scalar. */
int eegldm_ldm_train_step(eegldm_unet*, const float* latents, const float* noise, const int64_t* t,
                          const float* acp, int pred_type, int B, int L, float grad_scale, float* loss);

/* Round-6 prototype: nn.Conv1d(k 3, padding 1) over SiLU(GroupNorm(x)) -- the in_layers / out_layers pair of
 * /root/reference/src/models/unet.py:261-263,287-291 -- with the normalisation applied to the conv's operand tile inside LDS, so the
 * normalised tensor is never written (no-grad forward only: nothing is kept for a backward).  gn_stats: fp32 [B][G][2] = (mean, rstd) of x, as
 * eegldm_groupnorm_fwd leaves them.  bf16, silu = 1, B * L and L multiples of 192, Cout a multiple of 256, Cin of 64 (<= 1024), weight
 * registered with eegldm_conv1d_pack_kblocked; anything else is an error (there is no fallback inside this entry). */
int eegldm_conv1d_fwd_gn(eegldm_ctx*, const void* x, long ldx, const void* w, const float* bias, const float* gn_gamma, const float* gn_beta,
                         const float* gn_stats, int G, int silu, void* y, long ldy, int B, int L, int Cin, int Cout,
                         const float* rowvec, long ld_rowvec, const void* resid, long ld_resid, int dtype);

/* ------------------------------------------------------------------ UNet building blocks at primitive granularity (ABI 7)
 * ONE ResBlock(channels, emb_channels, dropout=0, out_channels, up / down) -- /root/reference/src/models/unet.py:227-327 -- or ONE
 * AttentionBlock(channels, num_heads=1) -- unet.py:132-174 -- run through exactly the kernel sequences the UNet executor runs per block
 * (csrc/net.hip res_forward / res_backward / attn_forward / attn_backward), plus timestep_embedding(t, dim) -- unet.py:12-36.
 * They exist so that the reference's per-primitive goldens are checked against the kernels, not only whole models
 * (tests/test_gpu_primitive_goldens.py).  Entry names are the reference module's state_dict keys; parameters / gradients are flat fp32
 * buffers as for the models; conv weights packed [K][Cout][Cin].  updown: 0 none, 1 down (AvgPool1d(2)), 2 up (nearest x2).
 * forward: x (B, channels, L) and y (B, out_channels, L') fp32 NCL; emb (B, emb_channels) fp32 -- the ResBlock's `emb` argument
 * (emb_layers = SiLU -> Linear runs inside); NULL for an AttentionBlock.  backward: grads += d<dy, y>/dparams; dx and demb are
 * WRITTEN (both nullable). */
/* use_scale_shift_norm != 0: ResBlock(use_scale_shift_norm=True) -- emb_layers.1 is 2 x out_channels wide (unet.py:279-284,318-322).
 * num_heads: AttentionBlock(channels, num_heads) (unet.py:132-166; pass channels / num_head_channels for the other spelling). */
int eegldm_resblock_create(eegldm_ctx*, int channels, int out_channels, int emb_channels, int groups, int updown, int use_scale_shift_norm,
                           int dtype, eegldm_block** out);
int eegldm_attnblock_create(eegldm_ctx*, int channels, int num_heads, int dtype, eegldm_block** out);
int eegldm_block_destroy(eegldm_block*);
int eegldm_block_num_entries(const eegldm_block*);
long eegldm_block_num_params(const eegldm_block*);
int eegldm_block_entry(const eegldm_block*, int i, char* name, int name_cap, long* offset, long* numel, int* ndim, int shape[3]);
int eegldm_block_bind(eegldm_block*, float* params, float* grads);
int eegldm_block_forward(eegldm_block*, const float* x, const float* emb, float* y, int B, int L);
int eegldm_block_backward(eegldm_block*, const float* dy, float* dx, float* demb);
/* out (B, dim) fp32 = [cos(t f_0) .. cos(t f_{h-1}), sin(t f_0) .. sin(t f_{h-1})], f_i = exp(-ln(10000) i / h), h = dim / 2; t int64 on the device */
int eegldm_timestep_embedding(eegldm_ctx*, const int64_t* t, float* out, int B, int dim);

/* ------------------------------------------------------------------ AutoencoderKL / PatchDiscriminator primitives
 * MONAI PatchDiscriminator layer `Convolution(.., norm=BATCH, act=LEAKYRELU(0.2))` (config/config_aekl_eeg.yaml:30-40; twin
 * src/models/discriminator.py:47-66): y = LeakyReLU_slope(BatchNorm1d(x)) on NLC rows [rows][C].  training != 0: batch statistics
 * (biased variance, eps 1e-5) and, when running_mean is given, the running-statistics update (momentum 0.1, unbiased variance,
 * num_batches_tracked += 1 -- a float count); training == 0: running statistics.  stats [C][2] fp32 receives (mean, rstd) -- the
 * tape of the backward.  gamma == NULL: plain LeakyReLU (no statistics).
 * Backward: dx, and dgamma / dbeta ACCUMULATED (fp32 [C]); the statistics of the forward. */
int eegldm_batchnorm_lrelu_fwd(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta, float* stats,
                               float* running_mean, float* running_var, float* num_batches_tracked, void* y, long ldy,
                               long rows, int C, float slope, int training, int dtype);
int eegldm_batchnorm_lrelu_bwd(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta, const float* stats,
                               const void* dy, long lddy, void* dx, long lddx, float* dgamma, float* dbeta, long rows, int C,
                               float slope, int dtype);
/* AutoencoderKL.sampling + the KL term of train_autoencoderkl.py:210-211 in one pass over n = B * latent * L' elements:
 * sigma = exp(clamp(log_var, -30, 20) / 2) (fp32 out), z = mu + eps * sigma (eps NULL: z = mu), and
 * *kl += (1 / B) * sum 0.5 * (mu^2 + sigma^2 - log sigma^2 - 1) when kl != NULL (zero it first).
 * Backward of  <dz, z> + kl_weight * KL : dmu = dz + (kl_weight / B) mu;
 * dlog_var = [-30 < log_var < 20] * (dz eps + (kl_weight / B)(sigma - 1 / sigma)) * sigma / 2.  dz NULL: the KL part alone. */
int eegldm_kl_reparam_fwd(eegldm_ctx*, const void* mu, const void* log_var, const float* eps, void* z, float* sigma, float* kl,
                          long n, int B, int dtype);
int eegldm_kl_reparam_bwd(eegldm_ctx*, const void* mu, const void* log_var, const float* eps, const float* sigma, const void* dz,
                          void* dmu, void* dlog_var, long n, float kl_weight_over_B, int dtype);

/* Stand-alone resampling in the NLC layout (rows = B * L, leading dimensions in elements): Downsample / Upsample with use_conv = False
 * (src/models/unet.py:177-224: nn.AvgPool1d(2, 2) / F.interpolate(scale_factor = 2, mode = "nearest")), the ops a ResBlock applies to h
 * and x when up / down is set (unet.py:308-313).  The executors fuse them into the GroupNorm kernels (`resample` above); these four are
 * the same arithmetic at primitive granularity.  L is the length of the op's INPUT (x for _fwd, the op's input for _bwd as well:
 * avgpool2_bwd takes dy [B][L/2][C] -> dx [B][L][C]; nearest2_bwd takes dy [B][2L][C] -> dx [B][L][C]). */
int eegldm_avgpool2_fwd(eegldm_ctx*, const void* x, long ldx, void* y, long ldy, int B, int L, int C, int dtype);
int eegldm_avgpool2_bwd(eegldm_ctx*, const void* dy, long lddy, void* dx, long lddx, int B, int L, int C, int dtype);
int eegldm_nearest2_fwd(eegldm_ctx*, const void* x, long ldx, void* y, long ldy, int B, int L, int C, int dtype);
int eegldm_nearest2_bwd(eegldm_ctx*, const void* dy, long lddy, void* dx, long lddx, int B, int L, int C, int dtype);

/* ------------------------------------------------------------------ losses of the AEKL step (fp32 NCL tensors)
 * L1Loss (train_autoencoderkl.py:155,206): *loss = mean|a-b|; da_accum (nullable) += grad_weight * d/da.
 * PatchAdversarialLoss("least_squares") (:156,214,226,228): LeakyReLU(0.05) on the logits, MSE against 1 / 0;
 *   dlogits (nullable) = grad_weight * d/dlogits.
 * JukeboxLoss(spatial_dims=1, reduction="sum") (:158,208): sum over windows and bins of (|FFT_ortho(recon)| -
 *   |FFT_ortho(target)|)^2; d_recon_accum (nullable) += grad_weight * d/d recon.  C must be 1; L in {3072, 768, 256, 96}. */
int eegldm_l1_loss(eegldm_ctx*, const float* a, const float* b, float* loss, float* da_accum, long n, float grad_weight);
int eegldm_lsgan_loss(eegldm_ctx*, const float* logits, int target_is_real, float* loss, float* dlogits, long n, float grad_weight);
int eegldm_spectral_loss(eegldm_ctx*, const float* recon, const float* target, float* loss, float* d_recon_accum,
                         int B, int C, int L, float grad_weight);
int eegldm_axpy(eegldm_ctx*, float* y, const float* x, float a, long n);

/* ------------------------------------------------------------------ AutoencoderKL
 * AutoencoderKL(spatial_dims=1, in_channels, out_channels, num_channels, latent_channels, num_res_blocks,
 * norm_num_groups, attention_levels=all False, with_*_nonlocal_attn=False) -- monai-generative, as configured by
 * config/config_aekl_eeg.yaml:19-28 and train_autoencoderkl.py:129-133.  Parameter table / bind / sync as for the UNet;
 * keys follow monai-generative's state_dict naming (encoder.blocks.N..., quant_conv_mu.conv..., ...). */
typedef struct {
  int in_channels, out_channels;
  int n_levels; int num_channels[8];
  int latent_channels, num_res_blocks, norm_num_groups;
  int dtype;
} eegldm_aekl_cfg;
int eegldm_aekl_create(eegldm_ctx*, const eegldm_aekl_cfg* cfg, eegldm_aekl** out);
int eegldm_aekl_destroy(eegldm_aekl*);
int eegldm_aekl_num_entries(const eegldm_aekl*);
long eegldm_aekl_num_params(const eegldm_aekl*);
int eegldm_aekl_entry(const eegldm_aekl*, int i, char* name, int name_cap, long* offset, long* numel, int* ndim, int shape[3]);
int eegldm_aekl_bind(eegldm_aekl*, float* params, float* grads);
int eegldm_aekl_sync_weights(eegldm_aekl*);
/* encode + sampling (encode_stage_2_inputs, train_ldm.py:148; Stage1Wrapper, training.py:15-26):
 * x (B,in,L) -> z = mu + eps*sigma (B,lat,L/2^(levels-1)); eps NULL -> z = mu.  z / z_mu / z_sigma nullable. */
int eegldm_aekl_encode(eegldm_aekl*, const float* x, const float* eps, float* z, float* z_mu, float* z_sigma, int B, int L);
/* decode_stage_2_outputs (sample_trials.py:166): z (B,lat,Ll) -> (B,out,Ll*2^(levels-1)) */
int eegldm_aekl_decode(eegldm_aekl*, const float* z, float* recon, int B, int Ll);
/* forward(x) -> (reconstruction, z_mu, z_sigma) with the reparameterisation noise eps supplied by the caller;
 * kl (nullable device scalar) receives 0.5*sum(mu^2+sigma^2-log sigma^2-1)/B.  Keeps the tape for backward. */
int eegldm_aekl_forward(eegldm_aekl*, const float* x, const float* eps, float* recon, float* z_mu, float* z_sigma,
                        float* kl, int B, int L);
/* grads += d/dparams [ <d_recon, recon> + kl_weight * KL ]; dx nullable */
int eegldm_aekl_backward(eegldm_aekl*, const float* d_recon, float kl_weight, float* dx);
/* The same plus <d_mu, z_mu> + <d_sigma, z_sigma>: gradients a caller's own loss put on the z_mu / z_sigma tensors eegldm_aekl_forward
 * returned (fp32 (B, lat, L/2^(levels-1)), each nullable) -- what an autograd engine hands back when the KL term is written with tensor
 * ops on those outputs, as train_autoencoderkl.py:210-211 does (eegldm.autograd). */
int eegldm_aekl_backward_ex(eegldm_aekl*, const float* d_recon, const float* d_mu, const float* d_sigma, float kl_weight, float* dx);

/* ------------------------------------------------------------------ quality metrics (fp32 NCL tensors, results on the device)
 * 1-D multi-scale SSIM -- the reference's local adaptation of MONAI's MultiScaleSSIMMetric (compute_mmds.py:214-408; used with
 * spatial_dims=1, data_range=1.0, kernel_size=7 at :487): per scale a `ksize`-tap (gaussian) valid convolution gives the local
 * moments, cs / ssim maps are averaged over channels and positions, avg_pool1d(2) between scales, out[b] = prod_s relu(.)^w_s
 * with the last scale using ssim.  kernel_host [ksize], weights_host [n_scales] are HOST arrays.  L <= 4096. */
int eegldm_ms_ssim_1d(eegldm_ctx*, const float* a, const float* b, float* out, int B, int C, int L, const float* kernel_host,
                      int ksize, const float* weights_host, int n_scales, float data_range, float k1, float k2);
/* Multitaper PSD of single-channel windows x (B, L), one-sided bins 0..n_bins-1 (bin k = k*sfreq/L Hz) -- what
 * mne's Epochs.compute_psd(fmax=18) computes per epoch at sample_trials.py:172-181 (method "multitaper", normalization "length"):
 * tapers (device, [n_tapers][L]) and weights_host ([n_tapers], sqrt of the DPSS concentrations) come from the host. */
int eegldm_psd_multitaper(eegldm_ctx*, const float* x, const float* tapers, const float* weights_host, int n_tapers, float sfreq,
                          int n_bins, float* psd, int B, int L);

/* ------------------------------------------------------------------ sampler
 * The whole sampling loop of sample_trials.py:149-170 (DDIM, eta 0) or util.py:261-285 / sample_trials_ddpm.py:99-104 (ancestral
 * DDPM, `ancestral` != 0) as one call: x <- noise (B,C,L); for i < n_steps: out = UNet(x, timesteps_host[i]);
 * x = step(out, x; a_t_host[i], a_prev_host[i] [, beta_t_host[i]]); then latents_out (nullable) <- x and windows_out (nullable) <-
 * decode(x * inv_scale_factor) when `ae` is given, else x itself (pixel-space model).  The three schedule arrays are HOST arrays
 * of n_steps entries (a_prev = 1 for the final step).  Ancestral noise is drawn on-device (Philox, noise_seed).
 * use_graph != 0: the UNet forward is captured once per (B, L) into a hipGraph and replayed (launch-bound at small B -- the
 * reference samples one window per call); *graph_used_host (nullable) reports whether the replay path ran. */
int eegldm_sample(eegldm_unet*, eegldm_aekl* ae, const float* noise, const int64_t* timesteps_host, const float* a_t_host,
                  const float* a_prev_host, const float* beta_t_host, int n_steps, int ancestral, int pred_type, int clip_sample,
                  float inv_scale_factor, uint64_t noise_seed, float* latents_out, float* windows_out, int B, int L, int use_graph,
                  int* graph_used_host);

/* ------------------------------------------------------------------ data-parallel collectives (RCCL over xGMI)
 * One communicator per process / GPU.  Stands where the reference gathers gradients with single-process nn.DataParallel
 * (train_ldm.py:190-192, train_autoencoderkl.py:230-233): every rank means the model's flat fp32 gradient buffer across ranks, the
 * parameters of rank 0 are broadcast once at start (BASELINE configs[3]: 8 ranks, global batch 2048).  Collectives run on the
 * communicator's own stream, ordered after the work enqueued on the context's stream at the time of the call; eegldm_comm_wait makes
 * the context's stream wait for them.  RCCL is loaded with dlopen when the first communicator is created (no link-time dependency).
 * id128: the 128-byte ncclUniqueId made by ONE rank with eegldm_comm_unique_id and handed to the others by the launcher
 * (eegldm.distributed passes it through the torch.distributed store). */
typedef struct eegldm_comm eegldm_comm;
int eegldm_comm_unique_id(char* out128);
int eegldm_comm_create(eegldm_ctx*, const char* id128, int rank, int world, eegldm_comm** out);
int eegldm_comm_destroy(eegldm_comm*);
int eegldm_comm_rank(const eegldm_comm*);
int eegldm_comm_world(const eegldm_comm*);
/* buf[0..n) <- mean over ranks, in place, as ceil(n / bucket_elems) collectives in one group (bucket_elems <= 0: one) */
int eegldm_comm_allreduce_mean_f32(eegldm_comm*, float* buf, long n, long bucket_elems);
int eegldm_comm_broadcast_f32(eegldm_comm*, float* buf, long n, int root);
int eegldm_comm_wait(eegldm_comm*);

/* ------------------------------------------------------------------ PatchDiscriminator
 * PatchDiscriminator(spatial_dims=1, num_layers_d, num_channels, in_channels, out_channels, kernel_size=3,
 * norm="BATCH", bias, padding=1) -- monai-generative, config/config_aekl_eeg.yaml:30-40.  forward returns the
 * last feature map (the reference indexes [-1], train_autoencoderkl.py:213).  BatchNorm running statistics live
 * in a separate caller-owned flat fp32 `buffers` array (num_batches_tracked stored as a float count). */
typedef struct {
  int in_channels, out_channels, num_channels, num_layers_d, kernel_size, padding, bias;
  int dtype;
} eegldm_disc_cfg;
int eegldm_disc_create(eegldm_ctx*, const eegldm_disc_cfg* cfg, eegldm_disc** out);
int eegldm_disc_destroy(eegldm_disc*);
int eegldm_disc_num_entries(const eegldm_disc*);
long eegldm_disc_num_params(const eegldm_disc*);
int eegldm_disc_entry(const eegldm_disc*, int i, char* name, int name_cap, long* offset, long* numel, int* ndim, int shape[3]);
int eegldm_disc_num_buffer_entries(const eegldm_disc*);
long eegldm_disc_num_buffers(const eegldm_disc*);
int eegldm_disc_buffer_entry(const eegldm_disc*, int i, char* name, int name_cap, long* offset, long* numel, int* ndim, int shape[3]);
int eegldm_disc_bind(eegldm_disc*, float* params, float* grads, float* buffers);
int eegldm_disc_sync_weights(eegldm_disc*);
/* training == 1: batch statistics + running-stat update (momentum 0.1); training == 2: batch statistics, running statistics left
 * untouched (a re-forward of an earlier input for its backward); 0: running statistics */
int eegldm_disc_forward(eegldm_disc*, const float* x, float* logits, int B, int L, int training);
/* MONAI PatchDiscriminator.forward returns the list of per-block feature maps (callers index [-1] = the logits,
 * train_autoencoderkl.py:213).  Feature `index` (0 = initial conv + LeakyReLU, then one per conv + BatchNorm + LeakyReLU layer)
 * of the most recent forward, copied out as fp32 (B, C, L); out == NULL only reports *C / *L. */
int eegldm_disc_feature(eegldm_disc*, int index, float* out, int* C, int* L);
/* param_grads != 0: grads += d/dparams; dx (nullable) = d/dx */
int eegldm_disc_backward(eegldm_disc*, const float* dlogits, float* dx, int param_grads);

/* The AEKL/GAN step body (train_autoencoderkl.py:203-234) for one batch: generator forward, L1 + KL + adversarial
 * (+ spectral when use_spectral) backward into the autoencoder's gradient buffer, then the two discriminator
 * passes (fake -> 0, real -> 1, 0.5*adv_weight each) into the discriminator's gradient buffer; BatchNorm running
 * statistics are updated three times, as in the reference.  Both gradient buffers must be zeroed by the caller
 * (optimizer.zero_grad) and both Adam steps run after the call; the result equals the reference order because the
 * discriminator passes only consume `reconstruction.detach()`.
 * losses: device float[6] = recons L1, spectral, KL, generator adversarial, D fake, D real.  recon_out nullable. */
int eegldm_aekl_train_step(eegldm_aekl*, eegldm_disc*, const float* x, const float* eps, float adv_weight, float kl_weight,
                           float spectral_weight, int use_spectral, float* losses, float* recon_out, int B, int L);

/* ------------------------------------------------------------------ FID on U-Sleep features
 * The feature extractor of /root/reference/src/compute_fid.py:357-386: USleep(in_chans=2, sfreq=100, depth=12, ...) of
 * /root/reference/src/models/usleep.py:101-287, forward only, fp32, reference (B, C, T) layout.  kernel_size = round(time_conv_size_s *
 * sfreq) (7 at 100 Hz), input_size = ceil(input_size_s * sfreq) (the classifier's AvgPool1d window).  Parameters live in ONE flat
 * fp32 buffer (conv weights in the reference's (Cout, Cin, K) layout), BatchNorm running statistics + num_batches_tracked (as floats)
 * in a second one; eegldm_usleep_entry lists both in the reference's state_dict order (kind 0 = parameter, 1 = buffer). */
typedef struct eegldm_usleep eegldm_usleep;
typedef struct {
  int in_chans, depth, n_time_filters, n_classes, kernel_size, input_size, with_skip_connection;
  float complexity_factor;
} eegldm_usleep_cfg;
int eegldm_usleep_create(eegldm_ctx*, const eegldm_usleep_cfg*, eegldm_usleep** out);
int eegldm_usleep_destroy(eegldm_usleep*);
long eegldm_usleep_num_params(const eegldm_usleep*);
long eegldm_usleep_num_buffers(const eegldm_usleep*);
int eegldm_usleep_num_entries(const eegldm_usleep*);
int eegldm_usleep_channel(const eegldm_usleep*, int i);      /* c_i of usleep.py:165-172, i = 0 .. depth + 1; -1 out of range */
int eegldm_usleep_entry(const eegldm_usleep*, int i, char* name, int name_cap, int* kind, long* offset, long* numel, int* ndim, int shape[3]);
int eegldm_usleep_bind(eegldm_usleep*, float* params, float* buffers);
/* USleep.forward (usleep.py:249-287): x (B, in_chans, T) -> y_pred (B, n_classes, T / input_size), decoder output (B, c_1, T) and the
 * bottleneck (B, c_{depth+1}, Lb); each output is optional, and without y_pred and decoder_out the decoder is skipped (the FID feature
 * of compute_fid.py:380-381 is the bottleneck).  training != 0: BatchNorm on batch statistics + running-statistics update -- what the
 * reference script runs, since it never calls model.eval(). */
int eegldm_usleep_forward(eegldm_usleep*, const float* x, float* y_pred, float* decoder_out, float* bottom, int B, int T, int training);
/* First and second moments of a feature batch (N, D), accumulated in fp64 device buffers the caller zeroed: sum[D] += sum_n f[n],
 * outer[D][D] += sum_n f[n] f[n]^T.  mean = sum / N, covariance = (outer - N mean mean^T) / (N - 1) are the inputs of the Frechet
 * distance (compute_fid.py:412-414 = monai-generative FIDMetric: torch.mean, unbiased _cov, trace of the matrix square root). */
int eegldm_feature_moments(eegldm_ctx*, const float* feats, long N, int D, double* sum, double* outer);

#ifdef __cplusplus
}
#endif
#endif /* EEGLDM_H */
