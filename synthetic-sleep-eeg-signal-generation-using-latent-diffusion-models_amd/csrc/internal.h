// Internal (non-ABI) launchers shared between translation units.
#pragma once
#include "common.h"

// elementwise.hip
int ew_temb(eegldm_ctx*, const int64_t* t, void* out, int B, int dim, int dtype);
int ew_silu(eegldm_ctx*, const float* x, void* y, long n, int dtype);
int ew_silu_bwd(eegldm_ctx*, const float* dy, const float* x, void* dx, long n, int dtype);
// deterministic mode (common.h eeg_deterministic): a private buffer of at least `bytes` for written partial sums (contents undefined)
int eeg_det_buffer(eegldm_ctx*, size_t bytes, float** out);
// total[i] += sum_p parts[p * stride + off + i], i < n, in the order p = 0, 1, ... (one thread per element, fp64 accumulator)
int ew_fold_partials_det(eegldm_ctx*, const float* parts, int nparts, long stride, int off, int n, float* total);
int ew_colsum(eegldm_ctx*, const void* x, long ldx, float* out_ps, long ldo, float* total, int B, int L, int C, int dtype);
int ew_softmax(eegldm_ctx*, const float* S, void* P, long rows, int n, int dtype);
int ew_softmax_bwd(eegldm_ctx*, const float* dP, const void* P, void* dS, long rows, int n, float alpha, int dtype);
int ew_add_rows(eegldm_ctx*, void* dst, long ldd, const void* src, long lds, long rows, int C, int dtype);
// nn.Dropout(p) in place with a regenerated Philox mask (the backward calls it on the gradient with the same seed / offset)
int ew_dropout_rows(eegldm_ctx*, void* x, long ld, long rows, int C, float p, uint64_t seed, uint64_t offset, int dtype);
// use_scale_shift_norm (unet.py:318-322): a = silu(hn * (1 + scale[b]) + shift[b]); emb row b = [scale (C) | shift (C)] at emb + b * lde
int ew_film_silu_fwd(eegldm_ctx*, const void* hn, long ldh, const float* emb, long lde, void* a, long lda, int B, int L, int C, int dtype);
// dhn = da * silu'(u) * (1 + scale); demb row b (ASSIGNED) = [sum_l g * hn | sum_l g], g = da * silu'(u)
int ew_film_silu_bwd(eegldm_ctx*, const void* hn, long ldh, const float* emb, long lde, const void* da, long ldda, void* dhn, long lddh,
                     float* demb, long ldde, int B, int L, int C, int dtype);
int ew_copy_rows(eegldm_ctx*, void* dst, long ldd, const void* src, long lds, long rows, int C, int dtype);

// direct_conv.hip
bool conv_is_thin(int Cin, int Cout, int dtype);
// K-blocked weight copies (elementwise.hip): table entry = one [3][Cout][Cin] 16-bit weight at element offset `off` of both buffers
struct KbDesc { long off; long chunk0; int cout, cin; };
int kblk_pack(eegldm_ctx* ctx, const void* w_plain, void* w_packed, const KbDesc* d_table, int n, long total_chunks);
int kblk_pack_one(eegldm_ctx* ctx, const void* w_plain, void* w_packed, int Cout, int Cin, int taps = 3);
// data-gradient copies: [3][Cout][Cin] -> [3][Cout / 32][Cin][32] (Cout, the data gradient's reduction index, K-blocked)
int kblk_pack_t(eegldm_ctx* ctx, const void* w_plain, void* w_packed, const KbDesc* d_table, int n, long total_chunks);
int kblk_pack_t_one(eegldm_ctx* ctx, const void* w_plain, void* w_packed, int Cout, int Cin, int taps = 3);   // one weight, both pointers at its first element
int s2ws_pack(eegldm_ctx* ctx, const void* w, void* wf, void* wd, int Cout, int Cin);      // elementwise.hip
int dconv_run(eegldm_ctx*, int dtype, bool dgrad, const void* in, long ldin, const void* w, const float* bias,
              const void* resid, long ldr, void* out, long ldout, int B, int Lin, int Lout, int Cin, int Cout, int K,
              int stride, int pad_l, float act_slope = 0.f);
bool dconv_fuses_act(int dtype, int Cin, int Cout, int K, long ldout);
int dconv_wgrad(eegldm_ctx*, int dtype, const void* x, long ldx, const void* dy, long lddy, float* dw, int B, int Lin,
                int Lout, int Cin, int Cout, int K, int stride, int pad_l, float* dbias, int* bias_done);
bool dconv_wgrad_tinyv_ok(int Cin, int Cout, int K);

// conv_ws.hip: weight-stationary 3-tap conv for 128 reduction channels (1 = handled, 0 = not this kernel's shape, < 0 = error)
struct SkinnyGn { const float2* part; const float* gamma; const float* beta; int cpg; float eps; int silu;
                  const float2* part_b = nullptr; int nqa = 0; };   // part_b: second producer of a concatenated operand (quads >= nqa)
// K extension of the few-row conv: + conv1x1(x2; w2 [Cout][Cin2]) + bias2 in the same launch (the ResBlock's skip_connection); excludes a residual
struct SkinnyExt { const void* x2; long ldx2; const void* w2; int Cin2; const float* bias2; };
bool conv_skinny_takes(int dtype, int Cin, int Cout, int taps, int B, int L);
int conv_skinny_ex(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, int Cin, int Cout, int taps, const float* bias,
                   const float* rowvec, long ld_rowvec, const void* resid, long ldr, void* y, long ldy, int B, int L,
                   const SkinnyGn* gn, float2* part_out, const SkinnyExt* ext = nullptr);
int conv_skinny_try(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, int Cin, int Cout, int taps, const float* bias,
                    const float* rowvec, long ld_rowvec, const void* resid, long ldr, void* y, long ldy, int B, int L);
// col_parts / col_nparts (both optional): forward launches leave per-block (sum, sum of squares) of every output column in col_parts[nparts][2 N] for a
// following BatchNorm (*col_nparts = rows written, 0 = not produced)
int conv_ws2_try(eegldm_ctx*, int dtype, int dgrad, const void* x, const void* w, const float* bias, void* y, int B, int Lo,
                 float* col_parts = nullptr, int* col_nparts = nullptr);      // stride-2 128 -> 256 over paired rows
int conv_ws_try(eegldm_ctx*, int dtype, const void* x, long ldx, const void* w, int Cin, int Cout, int transposed, const float* bias,
                const float* rowvec, long ld_rowvec, const void* resid, long ldr, void* y, long ldy, int B, int L,
                float* col_parts = nullptr, int* col_nparts = nullptr);

// ops.hip
int op_conv_fwd(eegldm_ctx*, int dtype, const void* x, long ldx, const void* w, const float* bias, void* y, long ldy,
                int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r,
                const float* rowvec, long ld_rowvec, const void* resid, long ldr, float act_slope = 0.f,
                float* col_parts = nullptr, int* col_nparts = nullptr);      // column statistics for a following BatchNorm when the kernel that runs provides them (conv_ws.hip)
bool op_conv_fuses_act(int dtype, int Cin, int Cout, int K, long ldy);
// y = conv3(x; w, pad 1) + bias + conv1(x2; w2) + bias2 (+ rowvec): the ResBlock tail h = skip_connection(x) + out_layers(h) (unet.py:302,327).
// ONE launch when the big-tile kernel takes it (16-bit, Cout % 256 == 0, K-blocked copies of both weights registered), else the two convs.
int op_conv3_skip_fwd(eegldm_ctx*, int dtype, const void* x, long ldx, const void* w, const float* bias, const void* x2, long ldx2,
                      const void* w2, const float* bias2, void* y, long ldy, int B, int L, int Cin, int Cin2, int Cout,
                      const float* rowvec, long ld_rowvec);
int op_conv_dgrad(eegldm_ctx*, int dtype, const void* dy, long lddy, const void* w, void* dx, long lddx,
                  int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r, const void* resid, long ldr);
bool op_wgrad_fuses_bias(int dtype, int Cin, int Cout);
int op_gn_fold_flush(eegldm_ctx*);      // batched dgamma / dbeta folds recorded by the one-pass GroupNorm backward in the deferred mode (norm.hip)
int op_wgrad_flush(eegldm_ctx*);      // launch the weight gradients recorded while ctx->defer_wgrad was set (grouped by shape)
int op_conv_wgrad(eegldm_ctx*, int dtype, const void* x, long ldx, const void* dy, long lddy, float* dw, float* dbias,
                  int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r);
int op_linear(eegldm_ctx*, int dtype, const void* x, long ldx, const void* w, long ldw, const float* bias, void* y, long ldy,
              int M, int N, int K, int out_f32);
int op_linear_dgrad(eegldm_ctx*, int dtype, const void* dy, long lddy, const void* w, long ldw, void* dx, long lddx,
                    int M, int N, int K, int out_f32);
int op_linear_wgrad(eegldm_ctx*, int dtype, const void* x, long ldx, const void* dy, long lddy, float* dw, long lddw,
                    int M, int N, int K);
int op_attention_fwd(eegldm_ctx*, int dtype, const void* qkv, long ldq, void* out, long ldo, void* probs, float* logits,
                     int B, int T, int C);
int op_attention_bwd(eegldm_ctx*, int dtype, const void* qkv, long ldq, const void* probs, const void* dout, long lddo,
                     void* dqkv, long lddq, float* dprobs, void* dlogits, int B, int T, int C);
// GroupNorm backward with an optional fused per-sample column sum of dx (norm.hip); *colsum_done = 1 when produced
int op_groupnorm_bwd(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta, const float* stats,
                     const void* dy, long lddy, void* dx, long lddx, float* dgamma, float* dbeta, int B, int L, int C, int G,
                     int fuse_silu, int resample, const void* dxr, long lddxr, int dtype, float* colsum_ps, long ldps, int* colsum_done,
                     const void* dxr2 = nullptr, long lddxr2 = 0, int* dxr2_done = nullptr, int* slots_deferred = nullptr, int defer_region = 0);
int op_gn_slot_reduce_deferred(eegldm_ctx*, float* dgamma, float* dbeta, int C, int region = 0);   // when *slots_deferred came back 1 (stream-ordered after the backward)   // dxr2: second, un-resampled addend [B*L][C] (skip gradient); *dxr2_done = 1 when the kernel added it
int ew_fold_partials(eegldm_ctx*, const float* parts, int nparts, int n, float* total);
// frozen-encoder fusion (enc_fused.hip): GroupNorm(G = 1) + SiLU on the operand load of a 3-tap conv, next layer's statistics from its epilogue
bool pre_conv3_ok(int dtype, int Cin, int Cout, int L);
int pre_conv3_launch(eegldm_ctx*, const void* x, const double* in_stats, const float* gamma, const float* beta, const void* w,
                     const float* bias, const void* resid, void* y, double* out_stats, int B, int L, int Cin, int Cout, float eps, int dtype);
int sample_stats_launch(eegldm_ctx*, const void* x, long n_per_sample, int B, double* stats, int dtype);
// fused short-sequence attention (attn.hip)
bool attn_chain_ok(int dtype, int T, int C, long ldq, long ldo);
int attn_chain_fwd(eegldm_ctx*, const void* qkv, long ldq, void* out, long ldo, void* probs, int B, int T, int C, int dtype);
int attn_chain_bwd(eegldm_ctx*, const void* qkv, long ldq, const void* probs, const void* dout, long lddo, void* dq, long lddq,
                   void* dS, int B, int T, int C, int fuse_kv, int dtype);
bool attn_chain_bwd_fuses_kv(eegldm_ctx*, int B, int T);      // whole-sample blocks: the backward kernel also writes dK and dV (no TN GEMMs)
