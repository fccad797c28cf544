// Operator layer: maps nn.Conv1d / nn.Linear / QKV attention (forward + both backward
// passes) onto the MFMA GEMM family (gemm.hip) or the thin-channel direct kernels
// (direct_conv.hip).  Everything is NLC; see include/eegldm.h for the contracts.
#include <math.h>

#include "common.h"
#include "internal.h"
#include <string.h>

static inline int conv_lout(int Lin, int K, int stride, int pad_l, int pad_r) { return (Lin + pad_l + pad_r - K) / stride + 1; }

bool op_conv_fuses_act(int dtype, int Cin, int Cout, int K, long ldy) { return conv_is_thin(Cin, Cout, dtype) && dconv_fuses_act(dtype, Cin, Cout, K, ldy); }
int op_conv_fwd(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, const float* bias, void* y, long ldy,
                int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r,
                const float* rowvec, long ld_rowvec, const void* resid, long ldr, float act_slope, float* col_parts, int* col_nparts) {
  if (col_nparts) *col_nparts = 0;
  EEG_CHECK(B > 0 && Lin > 0 && Cin > 0 && Cout > 0 && (K == 1 || K == 3) && (stride == 1 || stride == 2),
            "unsupported conv B=%d Lin=%d Cin=%d Cout=%d K=%d stride=%d", B, Lin, Cin, Cout, K, stride);
  const int Lout = conv_lout(Lin, K, stride, pad_l, pad_r);
  EEG_CHECK(Lout > 0, "empty output");
  if (conv_is_thin(Cin, Cout, dtype)) {
    EEG_CHECK(rowvec == nullptr, "rowvec add is not available on the thin-channel path");
    return dconv_run(ctx, dtype, false, x, ldx, w, bias, resid, ldr, y, ldy, B, Lin, Lout, Cin, Cout, K, stride, pad_l, act_slope);
  }
  EEG_CHECK(act_slope <= 0.f, "fused activation is only available on the thin-input direct conv");
  // stride 2, 64 -> 128 channels (the discriminator's second layer, config_aekl_eeg.yaml:30-40): a stride-1 conv of the weight-stationary kernel
  // over pairs of input rows (elementwise.hip s2ws_pack) -- HBM-bound at 3+ TB/s where the general implicit GEMM's 6-k-step tiles reach 2
  if (K == 3 && stride == 2 && pad_l == 1 && pad_r == 1 && Cin == 64 && Cout == 128 && ldx == Cin && Lin == 2 * Lout && !rowvec && !resid &&
      dtype != EEGLDM_F32 && !ctx->s2ws_f.empty()) {
    auto it = ctx->s2ws_f.find(w);
    if (it != ctx->s2ws_f.end()) {
      const int rc = conv_ws_try(ctx, dtype, x, 2 * Cin, it->second, 128, 128, 0, bias, nullptr, 0, nullptr, 0, y, ldy, B, Lout, col_parts, col_nparts);
      if (rc != 0) return rc < 0 ? rc : 0;
    }
  }
  if (stride == 1 && ((K == 3 && pad_l == 1 && pad_r == 1) || (K == 1 && pad_l == 0 && pad_r == 0))) {   // a few hundred rows (one window per call): conv_skinny.hip
    const int rc = conv_skinny_try(ctx, dtype, x, ldx, w, Cin, Cout, K, bias, rowvec, ld_rowvec, resid, ldr, y, ldy, B, Lin);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  // stride 2, 128 -> 256 channels (the discriminator's third layer): the paired-row weight-stationary kernel of conv_ws.hip
  if (K == 3 && stride == 2 && pad_l == 1 && pad_r == 1 && Cin == 128 && Cout == 256 && ldx == Cin && ldy == Cout && Lin == 2 * Lout && !rowvec && !resid) {
    const int rc = conv_ws2_try(ctx, dtype, 0, x, w, bias, y, B, Lout, col_parts, col_nparts);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (K == 3 && stride == 1 && pad_l == 1 && pad_r == 1) {      // HBM-bound wide-and-shallow layers: weights stay in registers (conv_ws.hip)
    const int rc = conv_ws_try(ctx, dtype, x, ldx, w, Cin, Cout, 0, bias, rowvec, ld_rowvec, resid, ldr, y, ldy, B, Lin);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  GemmArgs a = {};
  a.dtype = dtype; a.A = x; a.lda = ldx; a.B = w; a.ldb = Cin; a.sBt = (long)Cout * Cin; a.C = y; a.ldc = ldy;
  a.M = B * Lout; a.N = Cout; a.K = Cin; a.batch = 1; a.taps = K; a.alpha = 1.0f; a.bias = bias;
  a.rowvec = rowvec; a.ld_rowvec = ld_rowvec; a.rows_per_vec = Lout; a.resid = resid; a.ldr = ldr;
  a.bmode = GB_NT;
  if (K == 1) {
    EEG_CHECK(stride == 1 && pad_l == 0 && pad_r == 0, "1x1 conv must be stride 1, no padding");
    a.amode = GA_PLAIN;
  } else {
    EEG_CHECK(Lin == Lout * stride, "k3 conv geometry Lin=%d Lout=%d stride=%d not supported by the implicit GEMM", Lin, Lout, stride);
    a.amode = GA_CONV; a.Lout = Lout; a.Lin = Lin; a.stride = stride; a.pad_l = pad_l;
    if (dtype != EEGLDM_F32 && !ctx->kblk.empty()) {   // a K-blocked copy of this weight (NetBase::bind): contiguous weight tiles per K stage
      auto it = ctx->kblk.find(w);
      if (it != ctx->kblk.end()) { a.B = it->second; a.b_kblk = 1; }
    }
  }
  return gemm_launch(ctx, a);
}

int op_conv3_skip_fwd(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, const float* bias, const void* x2, long ldx2,
                      const void* w2, const float* bias2, void* y, long ldy, int B, int L, int Cin, int Cin2, int Cout,
                      const float* rowvec, long ld_rowvec) {
  if (dtype != EEGLDM_F32 && !ctx->kblk.empty() && !conv_is_thin(Cin, Cout, dtype)) {
    auto it = ctx->kblk.find(w), it2 = ctx->kblk.find(w2);
    if (it != ctx->kblk.end() && it2 != ctx->kblk.end()) {
      GemmArgs a = {};
      a.dtype = dtype; a.A = x; a.lda = ldx; a.B = it->second; a.b_kblk = 1; a.ldb = Cin; a.sBt = (long)Cout * Cin; a.C = y; a.ldc = ldy;
      a.M = B * L; a.N = Cout; a.K = Cin; a.batch = 1; a.taps = 3; a.alpha = 1.0f; a.bias = bias; a.splitk = 1; a.ups = 1;
      a.rowvec = rowvec; a.ld_rowvec = ld_rowvec; a.rows_per_vec = L; a.bmode = GB_NT;
      a.amode = GA_CONV; a.Lout = L; a.Lin = L; a.stride = 1; a.pad_l = 1;
      const int rc = gemm_big_skip_try(ctx, a, x2, ldx2, it2->second, Cin2, bias2);
      if (rc != 0) return rc < 0 ? rc : 0;
    }
  }
  EEG_TRY(op_conv_fwd(ctx, dtype, x2, ldx2, w2, bias2, y, ldy, B, L, Cin2, Cout, 1, 1, 0, 0, nullptr, 0, nullptr, 0));
  return op_conv_fwd(ctx, dtype, x, ldx, w, bias, y, ldy, B, L, Cin, Cout, 3, 1, 1, 1, rowvec, ld_rowvec, y, ldy);
}

int op_conv_dgrad(eegldm_ctx* ctx, int dtype, const void* dy, long lddy, const void* w, void* dx, long lddx,
                  int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r, const void* resid, long ldr) {
  const int Lout = conv_lout(Lin, K, stride, pad_l, pad_r);
  if (conv_is_thin(Cin, Cout, dtype))
    return dconv_run(ctx, dtype, true, dy, lddy, w, nullptr, resid, ldr, dx, lddx, B, Lin, Lout, Cin, Cout, K, stride, pad_l);
  if (K == 3 && stride == 1 && pad_l == 1 && pad_r == 1) {
    const int rc = conv_ws_try(ctx, dtype, dy, lddy, w, Cin, Cout, 1, nullptr, nullptr, 0, resid, ldr, dx, lddx, B, Lin);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  if (K == 3 && stride == 2 && pad_l == 1 && pad_r == 1 && Cin == 128 && Cout == 256 && lddx == Cin && lddy == Cout && Lin == 2 * Lout && !resid) {
    const int rc = conv_ws2_try(ctx, dtype, 1, dy, w, nullptr, dx, B, Lout);
    if (rc != 0) return rc < 0 ? rc : 0;
  }
  // data gradient of the stride-2 64 -> 128 conv: a stride-1 weight-stationary conv over dy that writes PAIRS of dx rows (see op_conv_fwd)
  if (K == 3 && stride == 2 && pad_l == 1 && pad_r == 1 && Cin == 64 && Cout == 128 && lddx == Cin && Lin == 2 * Lout && !resid &&
      dtype != EEGLDM_F32 && !ctx->s2ws_d.empty()) {
    auto it = ctx->s2ws_d.find(w);
    if (it != ctx->s2ws_d.end()) {
      const int rc = conv_ws_try(ctx, dtype, dy, lddy, it->second, 128, 128, 0, nullptr, nullptr, 0, nullptr, 0, dx, 2 * Cin, B, Lout);
      if (rc != 0) return rc < 0 ? rc : 0;
    }
  }
  GemmArgs a = {};
  a.dtype = dtype; a.A = dy; a.lda = lddy; a.B = w; a.ldb = Cin; a.sBt = (long)Cout * Cin; a.C = dx; a.ldc = lddx;
  a.M = B * Lin; a.N = Cin; a.K = Cout; a.batch = 1; a.taps = K; a.alpha = 1.0f; a.resid = resid; a.ldr = ldr;
  a.bmode = GB_TR;
  if (K == 1) {
    a.amode = GA_PLAIN;
    if (dtype != EEGLDM_F32 && !ctx->kblk_t.empty()) {      // [Cout / 32][Cin][32] copy: the data gradient of a 1 x 1 conv as an NT product on the big tile
      auto it = ctx->kblk_t.find(w);
      if (it != ctx->kblk_t.end()) a.B_alt = it->second;
    }
  } else {
    EEG_CHECK(Lin == Lout * stride, "k3 dgrad geometry not supported");
    // transposed conv as a stride-1 conv over the (virtually zero-upsampled) gradient, taps flipped
    a.amode = GA_CONV; a.tap_flip = 1; a.Lout = Lin; a.Lin = Lin; a.stride = 1; a.pad_l = K - 1 - pad_l;
    a.ups = stride; a.Lsrc = Lout;
    if (dtype != EEGLDM_F32 && stride == 1 && !ctx->kblk_t.empty()) {      // [tap][Cout / 32][Cin][32] copy (NetBase::bind): the big-tile kernel runs the data gradient as an NT product
      auto it = ctx->kblk_t.find(w);
      if (it != ctx->kblk_t.end()) a.B_alt = it->second;
    }
  }
  return gemm_launch(ctx, a);
}

// true when op_conv_wgrad produces the bias gradient inside its weight-gradient kernel (no shared context scratch: safe on the
// side stream): the split-K GEMM with 16-bit operands, or the tiny direct kernel of the [2,2,4] autoencoder layers
bool op_wgrad_fuses_bias(int dtype, int Cin, int Cout) {
  EEG_ENV_VAR(bool, fuse_bias, getenv("EEGLDM_NO_FUSED_BIAS_GRAD") == nullptr);
  if (!fuse_bias) return false;
  if (conv_is_thin(Cin, Cout, dtype)) return dconv_wgrad_tinyv_ok(Cin, Cout, 3);
  if (eeg_deterministic()) return false;      // the fused column sums are one fp32 atomic per row and K split: bias gradients by the ordered column-sum kernels
  return dtype != EEGLDM_F32;
}

int op_conv_wgrad(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* dy, long lddy, float* dw, float* dbias,
                  int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r) {
  const int Lout = conv_lout(Lin, K, stride, pad_l, pad_r);
  // bias gradient = column sums of dY: the split-K GEMM produces them from the A fragments it already holds (16-bit operands),
  // the tiny direct kernel from the dy rows it reads; fp32 parity mode and the other thin shapes keep the separate column-sum kernel.
  const bool thin = conv_is_thin(Cin, Cout, dtype);
  if (thin) {
    EEG_ENV_VAR(bool, fuse_bias, getenv("EEGLDM_NO_FUSED_BIAS_GRAD") == nullptr);
    int done = 0;
    EEG_TRY(dconv_wgrad(ctx, dtype, x, ldx, dy, lddy, dw, B, Lin, Lout, Cin, Cout, K, stride, pad_l, fuse_bias ? dbias : nullptr, &done));
    if (dbias && !done) EEG_TRY(ew_colsum(ctx, dy, lddy, nullptr, 0, dbias, B, Lout, Cout, dtype));
    return 0;
  }
  const bool bias_in_gemm = dbias && op_wgrad_fuses_bias(dtype, Cin, Cout);
  if (dbias && !bias_in_gemm) EEG_TRY(ew_colsum(ctx, dy, lddy, nullptr, 0, dbias, B, Lout, Cout, dtype));
  GemmArgs a = {};
  a.colsum = bias_in_gemm ? dbias : nullptr;
  a.dtype = dtype; a.amode = GA_TR; a.bmode = GB_TR;
  a.A = dy; a.lda = lddy; a.B = x; a.ldb = ldx; a.C = dw; a.ldc = Cin; a.sCt = (long)Cout * Cin;
  a.M = Cout; a.N = Cin; a.K = B * Lout; a.batch = 1; a.taps = 1; a.ztaps = K; a.alpha = 1.0f;
  a.conv_map = !(K == 1 && stride == 1 && pad_l == 0);   // a 1x1 conv's K index is the input row itself (plain TN GEMM: LDS-DMA staging)
  a.Lout = Lout; a.Lin = Lin; a.stride = stride; a.pad_l = pad_l;
  a.out_f32 = 1; a.atomic_out = 1;
  const int kstage = 2 * (dtype == EEGLDM_F32 ? 16 : 32);
  const bool fused3 = (K == 3 && stride == 1 && pad_l == 1 && pad_r == 1 && Lout % kstage == 0);
  int bn = Cin > 64 ? 128 : (Cin > 32 ? 64 : 32);
  long tiles;
  int blocks_per_cu = 2;
  if (fused3) {   // one block = all three taps of a 128 x 64 (or 128 x 128) weight tile: dY and X staged once per K chunk
    a.taps = 3; a.ztaps = 1; a.conv_map = 0;
    bn = Cin > 32 ? 64 : 32;
    // (the 128 x 128 x 3 tile as ONE 8-wave block per CU measured 5-8 % slower than two independent 128 x 64 blocks, round 2; removed in round 6)
    tiles = (long)((Cout + 127) / 128) * ((Cin + bn - 1) / bn);
  } else {
    tiles = (long)((Cout + 127) / 128) * ((Cin + bn - 1) / bn) * K;
  }
  // fill the resident blocks of every CU in ONE round: rounding the split count up (11 x 48 tiles = 528 blocks on 512 slots) leaves a
  // second round of 16 blocks that costs as much as the first
  long want = ((long)ctx->num_cu * blocks_per_cu) / tiles;
  long maxs = ((long)a.K + 8 * kstage - 1) / (8 * kstage);     // at least 8 stages per split
  if (want > maxs) want = maxs;
  a.splitk = (int)(want < 1 ? 1 : want);
  // Deferred mode (the UNet backward): record the problem instead of launching it; op_wgrad_flush() later runs all layers of one
  // shape as ONE grouped launch with far fewer K splits.  Eligible = what gemm_launch would send through the split-K workspace:
  // the fused 3-tap kernel or a 1x1 conv, K a whole number of stages, contiguous dW.
  if (ctx->defer_wgrad && (fused3 || (K == 1 && stride == 1 && pad_l == 0 && pad_r == 0)) && a.K % kstage == 0 && !a.wide_n &&
      a.M % 16 == 0 && a.N % 4 == 0) {
    WgradRec r; r.a = a; r.tiles = tiles; r.kstage = kstage; r.dst = dw; r.dbias = bias_in_gemm ? dbias : nullptr;
    r.a.colsum = nullptr; r.a.C = nullptr;
    ctx->wgrad_pending.push_back(r);
    return 0;
  }
  return gemm_launch(ctx, a);
}

// Launch the deferred weight gradients (ctx->wgrad_pending): problems with identical (taps, M, N, K, leading dimensions, conv geometry)
// -- the same conv shape in different layers -- go out as ONE grouped launch of up to GEMM_MAX_GROUP problems.  The split count is
// chosen per group from a cost model in K-stage units: rounds x (stages per block + fixed prologue / epilogue cost); a single layer
// fills the chip only by cutting K into 16-256 splits (12-48 stages per block, 50 MB of partial tiles per launch whatever the
// layer), a group of n layers needs n times fewer.
int op_wgrad_flush(eegldm_ctx* ctx) {
  std::vector<WgradRec> recs; recs.swap(ctx->wgrad_pending);
  if (recs.empty()) return 0;
  // (ctx->grp_slot counts the grouped launches of the current backward: unet.hip resets it when the backward starts)
  std::vector<char> done(recs.size(), 0);
  const long slots = (long)ctx->num_cu * 2;
  for (size_t i = 0; i < recs.size(); i++) {
    if (done[i]) continue;
    const GemmArgs& k = recs[i].a;
    GemmArgs g = k; g.ngroup = 0;
    GemmGroup tab; memset(&tab, 0, sizeof(tab));
    for (size_t j = i; j < recs.size() && g.ngroup < GEMM_MAX_GROUP; j++) {
      const GemmArgs& o = recs[j].a;
      if (done[j] || o.dtype != k.dtype || o.taps != k.taps || o.ztaps != k.ztaps || o.M != k.M || o.N != k.N || o.K != k.K ||
          o.Lout != k.Lout || o.Lin != k.Lin || o.conv_map != k.conv_map || recs[j].tiles != recs[i].tiles) continue;
      tab.A[g.ngroup] = o.A; tab.B[g.ngroup] = o.B; tab.CS[g.ngroup] = recs[j].dbias; tab.Dst[g.ngroup] = recs[j].dst;
      tab.lda[g.ngroup] = o.lda; tab.ldb[g.ngroup] = o.ldb;
      g.ngroup++; done[j] = 1;
    }
    // split count: minimise rounds x (K stages per block + ~8 stages of fixed cost), a whisker of preference for fewer partial tiles
    const long tiles_total = recs[i].tiles * g.ngroup, S = ((long)g.K + recs[i].kstage - 1) / recs[i].kstage;
    long maxs = S / 8; if (maxs < 1) maxs = 1; if (maxs > 256) maxs = 256;
    double best = 1e30; int bs = 1;
    constexpr double c_fixed = 8.0, c_block = 0.05;      // (round 5 swept both 5-50x: the step moved by +-0.05 ms -- the partial writes hide behind the other blocks' K loops)
    for (long sp = 1; sp <= maxs; sp++) {
      const long blocks = tiles_total * sp, rounds = (blocks + slots - 1) / slots;
      const double cost = (double)rounds * ((double)S / (double)sp + c_fixed) + c_block * (double)blocks;
      if (cost < best) { best = cost; bs = (int)sp; }
    }
    g.splitk = bs; g.batch = g.ngroup;
    EEG_TRY(gemm_launch_grouped(ctx, g, tab, ctx->grp_slot++));
  }
  return 0;
}

int op_linear(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* w, long ldw, const float* bias, void* y, long ldy,
              int M, int N, int K, int out_f32) {
  GemmArgs a = {};
  a.dtype = dtype; a.amode = GA_PLAIN; a.bmode = GB_NT; a.A = x; a.lda = ldx; a.B = w; a.ldb = ldw; a.C = y; a.ldc = ldy;
  a.M = M; a.N = N; a.K = K; a.batch = 1; a.taps = 1; a.alpha = 1.0f; a.bias = bias; a.out_f32 = out_f32;
  return gemm_launch(ctx, a);
}
// dx[M][K] = dy[M][N] w[N][K]
int op_linear_dgrad(eegldm_ctx* ctx, int dtype, const void* dy, long lddy, const void* w, long ldw, void* dx, long lddx,
                    int M, int N, int K, int out_f32) {
  GemmArgs a = {};
  a.dtype = dtype; a.amode = GA_PLAIN; a.bmode = GB_TR; a.A = dy; a.lda = lddy; a.B = w; a.ldb = ldw; a.C = dx; a.ldc = lddx;
  a.M = M; a.N = K; a.K = N; a.batch = 1; a.taps = 1; a.alpha = 1.0f; a.out_f32 = out_f32;
  // few output tiles but a long reduction (the batched embedding projection, K = 6656): split K over the chip
  const long tiles = (long)((M + 127) / 128) * ((K + 127) / 128);
  if (out_f32 && tiles * 8 < ctx->num_cu && N >= 1024) {
    long sk = ctx->num_cu / tiles; const long maxs = N / 256; if (sk > maxs) sk = maxs;
    if (sk > 1) {
      HIP_TRY(hipMemset2DAsync(dx, (size_t)lddx * 4, 0, (size_t)K * 4, M, ctx->stream));
      a.splitk = (int)sk;
    }
  }
  return gemm_launch(ctx, a);
}
// dw[N][K] += dy[M][N]^T x[M][K]
int op_linear_wgrad(eegldm_ctx* ctx, int dtype, const void* x, long ldx, const void* dy, long lddy, float* dw, long lddw,
                    int M, int N, int K) {
  GemmArgs a = {};
  a.dtype = dtype; a.amode = GA_TR; a.bmode = GB_TR; a.A = dy; a.lda = lddy; a.B = x; a.ldb = ldx; a.C = dw; a.ldc = lddw;
  a.M = N; a.N = K; a.K = M; a.batch = 1; a.taps = 1; a.alpha = 1.0f; a.out_f32 = 1; a.atomic_out = 1;
  return gemm_launch(ctx, a);
}

// ------------------------------------------------------------------ attention (unet.py:107-125)
int op_attention_fwd(eegldm_ctx* ctx, int dtype, const void* qkv, long ldq, void* out, long ldo, void* probs, float* logits,
                     int B, int T, int C) {
  if (attn_chain_ok(dtype, T, C, ldq, ldo)) return attn_chain_fwd(ctx, qkv, ldq, out, ldo, probs, B, T, C, dtype);
  const size_t es = dtype_size(dtype);
  const char* q = (const char*)qkv; const char* k = q + (size_t)C * es; const char* v = q + (size_t)2 * C * es;
  GemmArgs a = {};
  a.dtype = dtype; a.amode = GA_PLAIN; a.bmode = GB_NT; a.A = q; a.lda = ldq; a.sAb = (long)T * ldq; a.B = k; a.ldb = ldq;
  a.sBb = (long)T * ldq; a.C = logits; a.ldc = T; a.sCb = (long)T * T; a.M = T; a.N = T; a.K = C; a.batch = B; a.taps = 1;
  a.alpha = 1.0f / sqrtf((float)C);   // (q*s)(k*s), s = C^-1/4
  a.out_f32 = 1;
  EEG_TRY(gemm_launch(ctx, a));
  EEG_TRY(ew_softmax(ctx, logits, probs, (long)B * T, T, dtype));
  GemmArgs b = {};
  b.dtype = dtype; b.amode = GA_PLAIN; b.bmode = GB_TR; b.A = probs; b.lda = T; b.sAb = (long)T * T; b.B = v; b.ldb = ldq;
  b.sBb = (long)T * ldq; b.C = out; b.ldc = ldo; b.sCb = (long)T * ldo; b.M = T; b.N = C; b.K = T; b.batch = B; b.taps = 1;
  b.alpha = 1.0f;
  return gemm_launch(ctx, b);
}

int op_attention_bwd(eegldm_ctx* ctx, int dtype, const void* qkv, long ldq, const void* probs, const void* dout, long lddo,
                     void* dqkv, long lddq, float* dprobs, void* dlogits, int B, int T, int C) {
  const size_t es = dtype_size(dtype);
  const char* q = (const char*)qkv; const char* k = q + (size_t)C * es; const char* v = q + (size_t)2 * C * es;
  char* dq = (char*)dqkv; char* dk = dq + (size_t)C * es; char* dv = dq + (size_t)2 * C * es;
  const float alpha = 1.0f / sqrtf((float)C);
  GemmArgs g = {};
  const bool fused = attn_chain_ok(dtype, T, C, ldq, lddo) && lddq % 8 == 0;
  // whole-sample blocks: the fused backward kernel also produces dK and (round 5) dV -- no batched TN GEMMs with K = T = 192 (three K
  // stages per tile: 220-260 TF/s, 43 us per launch x 6)
  EEG_ENV_VAR(bool, no_fused_v, getenv("EEGLDM_ATTN_NO_FUSED_DV") != nullptr);
  const int fk = (fused && attn_chain_bwd_fuses_kv(ctx, B, T)) ? (no_fused_v ? 1 : 2) : 0;
  if (fk != 2) {
  // dV[s][c] = sum_t P[t][s] dO[t][c]
  g.dtype = dtype; g.amode = GA_TR; g.bmode = GB_TR; g.A = probs; g.lda = T; g.sAb = (long)T * T; g.B = dout; g.ldb = lddo;
  g.sBb = (long)T * lddo; g.C = dv; g.ldc = lddq; g.sCb = (long)T * lddq; g.M = T; g.N = C; g.K = T; g.batch = B; g.taps = 1; g.alpha = 1.f;
  EEG_TRY(gemm_launch(ctx, g));
  }
  if (fused) {
    // dP = dO V^T, dS = alpha P o (dP - rowsum(dP o P)), dQ = dS K in one launch; whole-sample blocks also produce dK = dS^T Q (a second
    // pass over the score tile in LDS, read transposed: no batched TN GEMM with K = T = 192 -- three K stages per tile, 260 TF/s -- and
    // no re-read of dS); otherwise dK below from the written dS
    EEG_TRY(attn_chain_bwd(ctx, qkv, ldq, probs, dout, lddo, dq, lddq, dlogits, B, T, C, fk, dtype));
    if (fk) return 0;
  } else {
  // dP[t][s] = sum_c dO[t][c] V[s][c]
  g = GemmArgs{};
  g.dtype = dtype; g.amode = GA_PLAIN; g.bmode = GB_NT; g.A = dout; g.lda = lddo; g.sAb = (long)T * lddo; g.B = v; g.ldb = ldq;
  g.sBb = (long)T * ldq; g.C = dprobs; g.ldc = T; g.sCb = (long)T * T; g.M = T; g.N = T; g.K = C; g.batch = B; g.taps = 1; g.alpha = 1.f;
  g.out_f32 = 1;
  EEG_TRY(gemm_launch(ctx, g));
  EEG_TRY(ew_softmax_bwd(ctx, dprobs, probs, dlogits, (long)B * T, T, alpha, dtype));
  // dQ[t][c] = sum_s dS[t][s] K[s][c]
  g = GemmArgs{};
  g.dtype = dtype; g.amode = GA_PLAIN; g.bmode = GB_TR; g.A = dlogits; g.lda = T; g.sAb = (long)T * T; g.B = k; g.ldb = ldq;
  g.sBb = (long)T * ldq; g.C = dq; g.ldc = lddq; g.sCb = (long)T * lddq; g.M = T; g.N = C; g.K = T; g.batch = B; g.taps = 1; g.alpha = 1.f;
  EEG_TRY(gemm_launch(ctx, g));
  }
  // dK[s][c] = sum_t dS[t][s] Q[t][c]
  g = GemmArgs{};
  g.dtype = dtype; g.amode = GA_TR; g.bmode = GB_TR; g.A = dlogits; g.lda = T; g.sAb = (long)T * T; g.B = q; g.ldb = ldq;
  g.sBb = (long)T * ldq; g.C = dk; g.ldc = lddq; g.sCb = (long)T * lddq; g.M = T; g.N = C; g.K = T; g.batch = B; g.taps = 1; g.alpha = 1.f;
  return gemm_launch(ctx, g);
}

// ================================================================== C ABI
extern "C" int eegldm_conv1d_fwd(eegldm_ctx* ctx, const void* x, long ldx, const void* w, const float* bias, void* y, long ldy,
                                 int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r,
                                 const float* rowvec, long ld_rowvec, const void* resid, long ld_resid, int dtype) {
  EEG_CHECK(ctx && x && w && y, "null pointer");
  return op_conv_fwd(ctx, dtype, x, ldx, w, bias, y, ldy, B, Lin, Cin, Cout, K, stride, pad_l, pad_r, rowvec, ld_rowvec, resid, ld_resid);
}
extern "C" int eegldm_conv1d_pack_kblocked(eegldm_ctx* ctx, const void* w, void* w_kblocked, int Cout, int Cin, int dtype) {
  return eegldm_conv1d_pack_kblocked_k(ctx, w, w_kblocked, Cout, Cin, 3, dtype);
}
extern "C" int eegldm_conv1d_pack_kblocked_k(eegldm_ctx* ctx, const void* w, void* w_kblocked, int Cout, int Cin, int K, int dtype) {
  EEG_CHECK(K == 1 || K == 3, "kernel size 1 or 3");
  EEG_CHECK(ctx && w && w_kblocked && w != w_kblocked, "null or aliased pointer");
  EEG_CHECK(dtype != EEGLDM_F32 && Cin > 0 && Cout > 0 && Cin % 32 == 0, "K-blocked weights: 16-bit dtype and Cin %% 32 == 0 (got dtype %d, Cin %d)", dtype, Cin);
  EEG_CHECK(((size_t)w % 16 == 0) && ((size_t)w_kblocked % 16 == 0), "weights must be 16-byte aligned");
  EEG_TRY(kblk_pack_one(ctx, w, w_kblocked, Cout, Cin, K));
  ctx->kblk[w] = w_kblocked;
  return 0;
}
extern "C" int eegldm_conv1d_pack_dgrad(eegldm_ctx* ctx, const void* w, void* w_t, int Cout, int Cin, int dtype) {
  return eegldm_conv1d_pack_dgrad_k(ctx, w, w_t, Cout, Cin, 3, dtype);
}
extern "C" int eegldm_conv1d_pack_dgrad_k(eegldm_ctx* ctx, const void* w, void* w_t, int Cout, int Cin, int K, int dtype) {
  EEG_CHECK(K == 1 || K == 3, "kernel size 1 or 3");
  EEG_CHECK(ctx && w && w_t && w != w_t, "null or aliased pointer");
  EEG_CHECK(dtype != EEGLDM_F32 && Cin > 0 && Cout > 0 && Cout % 32 == 0, "data-gradient weight copy: 16-bit dtype and Cout %% 32 == 0 (got dtype %d, Cout %d)", dtype, Cout);
  EEG_CHECK(((size_t)w % 16 == 0) && ((size_t)w_t % 16 == 0), "weights must be 16-byte aligned");
  EEG_TRY(kblk_pack_t_one(ctx, w, w_t, Cout, Cin, K));
  ctx->kblk_t[w] = w_t;
  return 0;
}
extern "C" int eegldm_conv1d_pack_stride2(eegldm_ctx* ctx, const void* w, void* w_fwd, void* w_dgrad, int Cout, int Cin, int dtype) {
  EEG_CHECK(ctx && w && w_fwd && w_dgrad, "null pointer");
  EEG_CHECK(dtype != EEGLDM_F32 && Cin == 64 && Cout == 128, "stride-2 weight-stationary copies: 16-bit dtype, Cin 64, Cout 128");
  EEG_TRY(s2ws_pack(ctx, w, w_fwd, w_dgrad, Cout, Cin));
  ctx->s2ws_f[w] = w_fwd; ctx->s2ws_d[w] = w_dgrad;
  return 0;
}
extern "C" int eegldm_conv1d_forget_kblocked(eegldm_ctx* ctx, const void* w) {
  EEG_CHECK(ctx && w, "null pointer");
  ctx->kblk.erase(w); ctx->kblk_t.erase(w); ctx->s2ws_f.erase(w); ctx->s2ws_d.erase(w);
  return 0;
}
// ---- round-6 prototype: GroupNorm(+SiLU) on the consuming conv's operand load (gemm_big.hip XF kernels)
// scale[b][c] = gamma[c] rstd[b, g(c)], shift[b][c] = beta[c] - mean[b, g(c)] scale[b][c] from the (mean, rstd) pairs a GroupNorm statistics pass
// (or a producer's epilogue moments) left: B x C floats each, a few hundred KB
__global__ void gn_affine_table_kernel(const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                       float* __restrict__ scale, float* __restrict__ shift, int B, int C, int G) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * C) return;
  const int b = (int)(i / C), c = (int)(i - (long)b * C), g = c / (C / G);
  const float mean = stats[((long)b * G + g) * 2], rstd = stats[((long)b * G + g) * 2 + 1];
  const float sc = gamma[c] * rstd;
  scale[i] = sc; shift[i] = beta[c] - mean * sc;
}
// y = conv3(act(GroupNorm(x))) + bias (+ rowvec + resid): the normalised operand is never written.  gn_stats: [B][G][2] = (mean, rstd) of x.
extern "C" int eegldm_conv1d_fwd_gn(eegldm_ctx* ctx, const void* x, long ldx, const void* w, const float* bias, const float* gn_gamma,
                                    const float* gn_beta, const float* gn_stats, int G, int silu, void* y, long ldy, int B, int L, int Cin, int Cout,
                                    const float* rowvec, long ld_rowvec, const void* resid, long ld_resid, int dtype) {
  EEG_CHECK(ctx && x && w && y && gn_gamma && gn_beta && gn_stats, "null pointer");
  EEG_CHECK(dtype == EEGLDM_BF16 && silu == 1, "prototype: bf16, GroupNorm + SiLU");
  EEG_CHECK(G > 0 && Cin % G == 0 && (size_t)B * Cin * 2 * sizeof(float) <= (16u << 20), "bad GroupNorm shape");
  auto it = ctx->kblk.find(w);
  EEG_CHECK(it != ctx->kblk.end(), "the weight needs its K-blocked copy (eegldm_conv1d_pack_kblocked)");
  float* scale = (float*)((char*)ctx->scratch + (8u << 20)); float* shift = scale + (size_t)B * Cin;
  hipLaunchKernelGGL(gn_affine_table_kernel, dim3((unsigned)(((long)B * Cin + 255) / 256)), dim3(256), 0, ctx->stream, gn_stats, gn_gamma, gn_beta, scale, shift, B, Cin, G);
  LAUNCH_CHECK();
  GemmArgs a = {};
  a.dtype = dtype; a.A = x; a.lda = ldx; a.B = it->second; a.b_kblk = 1; a.ldb = Cin; a.sBt = (long)Cout * Cin; a.C = y; a.ldc = ldy;
  a.M = B * L; a.N = Cout; a.K = Cin; a.batch = 1; a.taps = 3; a.alpha = 1.0f; a.bias = bias; a.splitk = 1; a.ups = 1;
  a.rowvec = rowvec; a.ld_rowvec = ld_rowvec; a.rows_per_vec = L; a.resid = resid; a.ldr = ld_resid; a.bmode = GB_NT;
  a.amode = GA_CONV; a.Lout = L; a.Lin = L; a.stride = 1; a.pad_l = 1;
  const int rc = gemm_big_xf_try(ctx, a, scale, shift, Cin, 1, 0.f);
  if (rc < 0) return rc;
  EEG_CHECK(rc == 1, "not eligible: B * L %% 192, L %% 192, Cout %% 256, Cin %% 64, 16-byte aligned operands, Cin <= 1024");
  return 0;
}
extern "C" int eegldm_conv1d_skip_fwd(eegldm_ctx* ctx, const void* x, long ldx, const void* w, const float* bias, const void* x2, long ldx2,
                                      const void* w2, const float* bias2, void* y, long ldy, int B, int L, int Cin, int Cin2, int Cout,
                                      const float* rowvec, long ld_rowvec, int dtype) {
  EEG_CHECK(ctx && x && w && x2 && w2 && y, "null pointer");
  EEG_CHECK(B > 0 && L > 0 && Cin > 0 && Cin2 > 0 && Cout > 0, "bad sizes");
  return op_conv3_skip_fwd(ctx, dtype, x, ldx, w, bias, x2, ldx2, w2, bias2, y, ldy, B, L, Cin, Cin2, Cout, rowvec, ld_rowvec);
}
extern "C" int eegldm_conv1d_bwd_data(eegldm_ctx* ctx, const void* dy, long lddy, const void* w, void* dx, long lddx, int B, int Lin,
                                      int Cin, int Cout, int K, int stride, int pad_l, int pad_r, const void* resid, long ld_resid, int dtype) {
  EEG_CHECK(ctx && dy && w && dx, "null pointer");
  return op_conv_dgrad(ctx, dtype, dy, lddy, w, dx, lddx, B, Lin, Cin, Cout, K, stride, pad_l, pad_r, resid, ld_resid);
}
extern "C" int eegldm_conv1d_bwd_weight(eegldm_ctx* ctx, const void* x, long ldx, const void* dy, long lddy, float* dw, float* dbias,
                                        int B, int Lin, int Cin, int Cout, int K, int stride, int pad_l, int pad_r, int dtype) {
  EEG_CHECK(ctx && x && dy && dw, "null pointer");
  return op_conv_wgrad(ctx, dtype, x, ldx, dy, lddy, dw, dbias, B, Lin, Cin, Cout, K, stride, pad_l, pad_r);
}
extern "C" int eegldm_linear_fwd(eegldm_ctx* ctx, const void* x, long ldx, const void* w, const float* bias, void* y, long ldy,
                                 int M, int N, int K, int dtype, int out_f32) {
  EEG_CHECK(ctx && x && w && y, "null pointer");
  return op_linear(ctx, dtype, x, ldx, w, K, bias, y, ldy, M, N, K, out_f32);
}
// nn.Linear backward (autograd of unet.py:373-377,277-285): dx[M][K] = dy w (optional), dw[N][K] += dy^T x, dbias[N] += column sums of dy
extern "C" int eegldm_linear_bwd(eegldm_ctx* ctx, const void* x, long ldx, const void* w, const void* dy, long lddy, void* dx, long lddx,
                                 float* dw, float* dbias, int M, int N, int K, int dtype, int dx_f32) {
  EEG_CHECK(ctx && dy && (dx || dw || dbias), "null pointer");
  EEG_CHECK(!dx || w, "dx needs the weights"); EEG_CHECK(!dw || x, "dw needs the input");
  if (dx) EEG_TRY(op_linear_dgrad(ctx, dtype, dy, lddy, w, K, dx, lddx, M, N, K, dx_f32));
  if (dw) EEG_TRY(op_linear_wgrad(ctx, dtype, x, ldx, dy, lddy, dw, K, M, N, K));
  if (dbias) EEG_TRY(ew_colsum(ctx, dy, lddy, nullptr, 0, dbias, 1, M, N, dtype));
  return 0;
}
extern "C" int eegldm_attention_fwd(eegldm_ctx* ctx, const void* qkv, long ldqkv, void* out, long ldo, void* probs, float* scratch_logits,
                                    int B, int T, int C, int dtype) {
  EEG_CHECK(ctx && qkv && out && probs && scratch_logits, "null pointer");
  return op_attention_fwd(ctx, dtype, qkv, ldqkv, out, ldo, probs, scratch_logits, B, T, C);
}
extern "C" int eegldm_attention_bwd(eegldm_ctx* ctx, const void* qkv, long ldqkv, const void* probs, const void* dout, long lddo,
                                    void* dqkv, long lddqkv, float* scratch_dprobs, void* scratch_dlogits, int B, int T, int C, int dtype) {
  EEG_CHECK(ctx && qkv && probs && dout && dqkv && scratch_dprobs && scratch_dlogits, "null pointer");
  return op_attention_bwd(ctx, dtype, qkv, ldqkv, probs, dout, lddo, dqkv, lddqkv, scratch_dprobs, scratch_dlogits, B, T, C);
}
