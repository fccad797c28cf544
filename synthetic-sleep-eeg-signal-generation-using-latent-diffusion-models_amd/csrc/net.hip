// ResBlock / AttentionBlock kernel sequences shared by the model executors, plus the
// flat-parameter plumbing of NetBase.  Host code only.
//   ResBlock:       /root/reference/src/models/unet.py:307-327 (with timestep embedding) and the
//                   MONAI AutoencoderKL ResBlock (no embedding; twin /root/reference/src/models/ae_kl.py:67-80)
//   AttentionBlock: /root/reference/src/models/unet.py:168-174
#include <stdlib.h>
#include <string.h>

#include "net.h"

int NetBase::bind(float* p, float* g) {
  EEG_CHECK(p, "null parameter buffer");
  params = p; grads = g;
  if (dtype == EEGLDM_F32) { wT = p; owns_wT = false; }
  else if (!wT) { HIP_TRY(hipMalloc(&wT, (size_t)nparams * 2)); owns_wT = true; }
  EEG_ENV_VAR(bool, no_kblk, getenv("EEGLDM_NO_KBLOCKED_WEIGHTS") != nullptr);
  if (dtype != EEGLDM_F32 && !wK && !no_kblk) {
    std::vector<KbDesc> tab; long chunks = 0;
    for (const Entry& e : entries) {
      // [Cout][Cin][3] conv weights of the implicit-GEMM path; 16-byte chunks of both layouts must line up (offset % 8, Cin % 32)
      // (and the 1 x 1 skip_connection weights, [1][Cin / 32][Cout][32]: folded into the ResBlock's second conv as further K stages, gemm_big.hip)
      const bool skip1 = e.ndim == 3 && e.shape[2] == 1 && e.shape[1] % 64 == 0 && e.shape[0] % 256 == 0 && e.name.find("skip_connection") != std::string::npos;
      if (e.ndim != 3 || (e.shape[2] != 3 && !skip1) || e.shape[1] % 32 != 0 || e.offset % 8 != 0 || conv_is_thin(e.shape[1], e.shape[0], dtype)) continue;
      KbDesc d; d.off = e.offset; d.chunk0 = chunks; d.cout = e.shape[0]; d.cin = e.shape[1];
      chunks += e.numel / 8; tab.push_back(d);
    }
    if (!tab.empty()) {
      HIP_TRY(hipMalloc(&wK, (size_t)nparams * 2));
      HIP_TRY(hipMalloc(&d_kb, tab.size() * sizeof(KbDesc)));
      HIP_TRY(hipMemcpy(d_kb, tab.data(), tab.size() * sizeof(KbDesc), hipMemcpyHostToDevice));
      n_kb = (int)tab.size(); kb_chunks = chunks;
      for (const KbDesc& d : tab) ctx->kblk[W(d.off)] = (const char*)wK + (size_t)d.off * 2;
    }
  }
  if (dtype != EEGLDM_F32 && !wKT && !no_kblk) {
    std::vector<KbDesc> tab; long chunks = 0;
    for (const Entry& e : entries) {
      // data gradient as an NT product on the big tile: N = Cin a multiple of 256, K = Cout a multiple of 64
      if (e.ndim != 3 || (e.shape[2] != 3 && e.shape[2] != 1) || e.shape[1] % 256 != 0 || e.shape[0] % 64 != 0 || e.offset % 8 != 0) continue;      // (3-tap and 1 x 1 convs)
      KbDesc d; d.off = e.offset; d.chunk0 = chunks; d.cout = e.shape[0]; d.cin = e.shape[1];
      chunks += e.numel / 8; tab.push_back(d);
    }
    if (!tab.empty()) {
      HIP_TRY(hipMalloc(&wKT, (size_t)nparams * 2));
      HIP_TRY(hipMalloc(&d_kbt, tab.size() * sizeof(KbDesc)));
      HIP_TRY(hipMemcpy(d_kbt, tab.data(), tab.size() * sizeof(KbDesc), hipMemcpyHostToDevice));
      n_kbt = (int)tab.size(); kbt_chunks = chunks;
      for (const KbDesc& d : tab) ctx->kblk_t[W(d.off)] = (const char*)wKT + (size_t)d.off * 2;
    }
  }
  if (dtype != EEGLDM_F32 && !wS2 && !no_kblk) {
    for (const Entry& e : entries)
      if (e.ndim == 3 && e.shape[2] == 3 && e.shape[0] == 128 && e.shape[1] == 64 && e.offset % 8 == 0) s2_offs.push_back(e.offset);
    if (!s2_offs.empty()) {
      constexpr size_t ONE = (size_t)3 * 128 * 128 * 2;      // bytes of one repacked weight
      HIP_TRY(hipMalloc(&wS2, s2_offs.size() * 2 * ONE));
      for (size_t i = 0; i < s2_offs.size(); i++) {
        ctx->s2ws_f[W(s2_offs[i])] = (const char*)wS2 + (2 * i) * ONE;
        ctx->s2ws_d[W(s2_offs[i])] = (const char*)wS2 + (2 * i + 1) * ONE;
      }
    }
  }
  return sync_weights();
}
int NetBase::flush_gn_folds() {
  for (const GnFold& f : gn_pending) EEG_TRY(op_gn_slot_reduce_deferred(ctx, f.dgamma, f.dbeta, f.C, f.region));
  gn_pending.clear();
  return 0;
}
void NetBase::release_kblk() {      // the two copies are independent: a model may own either without the other
  if (wS2) {
    for (long o : s2_offs) { ctx->s2ws_f.erase(W(o)); ctx->s2ws_d.erase(W(o)); }
    (void)hipFree(wS2); wS2 = nullptr; s2_offs.clear();
  }
  if (wK) {
    for (auto it = ctx->kblk.begin(); it != ctx->kblk.end();) {
      const char* v = (const char*)it->second;
      if (v >= (const char*)wK && v < (const char*)wK + (size_t)nparams * 2) it = ctx->kblk.erase(it); else ++it;
    }
    (void)hipFree(wK); (void)hipFree(d_kb); wK = nullptr; d_kb = nullptr; n_kb = 0;
  }
  if (wKT) {
    for (auto it = ctx->kblk_t.begin(); it != ctx->kblk_t.end();) {
      const char* v = (const char*)it->second;
      if (v >= (const char*)wKT && v < (const char*)wKT + (size_t)nparams * 2) it = ctx->kblk_t.erase(it); else ++it;
    }
    (void)hipFree(wKT); (void)hipFree(d_kbt); wKT = nullptr; d_kbt = nullptr; n_kbt = 0;
  }
}
int NetBase::sync_weights() {
  EEG_CHECK(params, "bind parameters first");
  if (dtype == EEGLDM_F32) return 0;
  EEG_TRY(eegldm_cast(ctx, params, wT, nparams, dtype));
  EEG_TRY(kblk_pack(ctx, wT, wK, (const KbDesc*)d_kb, n_kb, kb_chunks));
  for (size_t i = 0; i < s2_offs.size(); i++) {
    constexpr size_t ONE = (size_t)3 * 128 * 128 * 2;
    EEG_TRY(s2ws_pack(ctx, W(s2_offs[i]), (char*)wS2 + (2 * i) * ONE, (char*)wS2 + (2 * i + 1) * ONE, 128, 64));
  }
  return kblk_pack_t(ctx, wT, wKT, (const KbDesc*)d_kbt, n_kbt, kbt_chunks);
}
int entry_query(const NetBase* u, int i, char* name, int cap, long* offset, long* numel, int* ndim, int shape[3]) {
  EEG_CHECK(u && i >= 0 && i < (int)u->entries.size(), "entry index %d out of range", i);
  const Entry& e = u->entries[i];
  if (name && cap > 0) { strncpy(name, e.name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (offset) *offset = e.offset;
  if (numel) *numel = e.numel;
  if (ndim) *ndim = e.ndim;
  if (shape) { shape[0] = e.shape[0]; shape[1] = e.shape[1]; shape[2] = e.shape[2]; }
  return 0;
}

// ------------------------------------------------------------------ ResBlock (unet.py:307-327)
int res_forward(NetBase* u, const ResDesc& r, const View& x, int B, int Lin, const View& out) {
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype;
  const int Lout = r.updown == 1 ? Lin / 2 : (r.updown == 2 ? Lin * 2 : Lin);
  ResTape t; t.x = x; t.B = B; t.Lin = Lin; t.Lout = Lout;
  ALLOC_OR_FAIL(t.st1, (float*)u->arena.alloc(sizeof(float) * 2 * B * r.groups));
  ALLOC_OR_FAIL(t.st2, (float*)u->arena.alloc(sizeof(float) * 2 * B * r.groups));
  const float* emb = (r.emb_col >= 0 && !r.ssn) ? u->emb_all + r.emb_col : nullptr;
  // ---- eval, few rows (NetBase::eval_fuse): conv1 leaves GroupNorm 2's statistics and conv2 normalises on load; conv2 leaves the
  // statistics of the block output; and when the producers of THIS block's input left theirs, GroupNorm 1 is folded into conv1 too
  const int cpg1 = r.cin / r.groups, cpg2 = r.cout / r.groups;
  const size_t need = (size_t)B * (Lout / 16) * (r.cout / 4);       // float2 slots of a (B * Lout) x cout tensor
  const bool fuse2 = u->eval_fuse && !r.ssn && Lout % 32 == 0 && cpg2 >= 4 && cpg2 % 4 == 0 && r.cout % r.groups == 0 &&
                     conv_skinny_takes(dt, r.cin, r.cout, 3, B, Lout) && conv_skinny_takes(dt, r.cout, r.cout, 3, B, Lout);
  const NetBase::PartReg *pa = nullptr, *pb = nullptr;
  bool fuse1 = false;
  if (fuse2 && !r.updown && cpg1 >= 4 && cpg1 % 4 == 0 && r.cin % r.groups == 0) {
    pa = u->find_part(x.p);
    if (pa && pa->nq * 4 < r.cin) pb = u->find_part((const char*)x.p + (size_t)pa->nq * 4 * dtype_size(dt));
    fuse1 = pa && (pa->nq * 4 == r.cin || (pb && (pa->nq + pb->nq) * 4 == r.cin));
  }
  auto norm1 = [&]() -> int {       // the stand-alone first GroupNorm (+ resampling)
    ALLOC_OR_FAIL(t.a1.p, u->alloc_act((long)B * Lout, r.cin)); t.a1.ld = r.cin; t.a1.C = r.cin;
    return eegldm_groupnorm_fwd(ctx, x.p, x.ld, u->P(r.gn1_w), u->P(r.gn1_b), t.a1.p, t.a1.ld, t.st1, B, Lin, r.cin, r.groups, GN_EPS, 1,
                                r.updown, r.updown ? t.xr.p : nullptr, t.xr.ld, dt);
  };
  if (r.updown) { ALLOC_OR_FAIL(t.xr.p, u->alloc_act((long)B * Lout, r.cin)); t.xr.ld = r.cin; t.xr.C = r.cin; } else t.xr = x;
  if (!fuse1) EEG_TRY(norm1());
  ALLOC_OR_FAIL(t.h1.p, u->alloc_act((long)B * Lout, r.cout)); t.h1.ld = r.cout; t.h1.C = r.cout;
  bool fused = false;
  float2* area = fuse2 ? u->fuse_alloc(need) : nullptr;
  if (!area && fuse1) { fuse1 = false; EEG_TRY(norm1()); }        // no room for the statistics: the plain path below needs the normalised tensor
  if (area) {
    SkinnyGn gn1 = {pa ? pa->slots : nullptr, u->P(r.gn1_w), u->P(r.gn1_b), cpg1, GN_EPS, 1};
    if (pb) { gn1.part_b = pb->slots; gn1.nqa = pa->nq; }
    int rc1 = conv_skinny_ex(ctx, dt, fuse1 ? x.p : t.a1.p, fuse1 ? x.ld : t.a1.ld, u->W(r.c1_w), r.cin, r.cout, 3, u->P(r.c1_b), emb, u->emb_ld,
                             nullptr, 0, t.h1.p, t.h1.ld, B, Lout, fuse1 ? &gn1 : nullptr, area);
    if (rc1 < 0) return rc1;
    if (rc1 == 1 && fuse1) u->fused_used = true;
    if (rc1 == 0 && fuse1) EEG_TRY(norm1());           // declined (alignment): the stand-alone norm after all
    if (rc1 == 1) {
      const SkinnyGn gn = {area, u->P(r.gn2_w), u->P(r.gn2_b), cpg2, GN_EPS, 1};
      float2* opart = u->fuse_alloc(need);              // statistics of the block output, for whoever normalises it next
      // skip_connection(x) + conv2(h): ONE launch with the 1 x 1 conv as a second reduction of the few-row kernel (round 5); the two-launch
      // form (the 1 x 1 conv writes `out`, conv2 adds it as its residual) when the extension does not apply
      EEG_ENV_VAR(bool, no_ext, getenv("EEGLDM_NO_FUSED_SKIP") != nullptr);
      int rc2 = 0; bool skip_done = false;
      if (r.sk_w >= 0 && !no_ext && r.cin % 32 == 0) {
        const SkinnyExt ext = {t.xr.p, t.xr.ld, u->W(r.sk_w), r.cin, u->P(r.sk_b)};
        rc2 = conv_skinny_ex(ctx, dt, t.h1.p, t.h1.ld, u->W(r.c2_w), r.cout, r.cout, 3, u->P(r.c2_b), nullptr, 0, nullptr, 0,
                             out.p, out.ld, B, Lout, &gn, opart, &ext);
        if (rc2 < 0) return rc2;
        skip_done = rc2 == 1;
      }
      if (rc2 != 1) {
        if (r.sk_w >= 0)
          EEG_TRY(op_conv_fwd(ctx, dt, t.xr.p, t.xr.ld, u->W(r.sk_w), u->P(r.sk_b), out.p, out.ld, B, Lout, r.cin, r.cout, 1, 1, 0, 0, nullptr, 0, nullptr, 0));
        const View res = r.sk_w >= 0 ? out : t.xr;
        rc2 = conv_skinny_ex(ctx, dt, t.h1.p, t.h1.ld, u->W(r.c2_w), r.cout, r.cout, 3, u->P(r.c2_b), nullptr, 0, res.p, res.ld,
                             out.p, out.ld, B, Lout, &gn, opart);
        if (rc2 < 0) return rc2;
        skip_done = r.sk_w >= 0;
      }
      (void)skip_done;
      if (rc2 == 1) { fused = true; u->fused_used = true; if (opart) u->part_reg.push_back({out.p, opart, r.cout / 4}); }
    } else {
      EEG_TRY(op_conv_fwd(ctx, dt, t.a1.p, t.a1.ld, u->W(r.c1_w), u->P(r.c1_b), t.h1.p, t.h1.ld, B, Lout, r.cin, r.cout, 3, 1, 1, 1, emb, u->emb_ld, nullptr, 0));
    }
    const int rc1_done = rc1;
    if (!fused) {      // conv2 declined (alignment): the stand-alone GroupNorm and the plain conv, as below (the skip conv already ran)
      ALLOC_OR_FAIL(t.a2.p, u->alloc_act((long)B * Lout, r.cout)); t.a2.ld = r.cout; t.a2.C = r.cout;
      EEG_TRY(eegldm_groupnorm_fwd(ctx, t.h1.p, t.h1.ld, u->P(r.gn2_w), u->P(r.gn2_b), t.a2.p, t.a2.ld, t.st2, B, Lout, r.cout, r.groups, GN_EPS, 1,
                                   0, nullptr, 0, dt));
      const bool skip_done = rc1_done == 1 && r.sk_w >= 0;
      if (r.sk_w >= 0 && !skip_done)
        EEG_TRY(op_conv_fwd(ctx, dt, t.xr.p, t.xr.ld, u->W(r.sk_w), u->P(r.sk_b), out.p, out.ld, B, Lout, r.cin, r.cout, 1, 1, 0, 0, nullptr, 0, nullptr, 0));
      const View res = r.sk_w >= 0 ? out : t.xr;
      EEG_TRY(op_conv_fwd(ctx, dt, t.a2.p, t.a2.ld, u->W(r.c2_w), u->P(r.c2_b), out.p, out.ld, B, Lout, r.cout, r.cout, 3, 1, 1, 1, nullptr, 0, res.p, res.ld));
    }
    u->rt.push_back(t);
    return 0;
  }
  EEG_TRY(op_conv_fwd(ctx, dt, t.a1.p, t.a1.ld, u->W(r.c1_w), u->P(r.c1_b), t.h1.p, t.h1.ld, B, Lout, r.cin, r.cout, 3, 1, 1, 1,
                      emb, u->emb_ld, nullptr, 0));
  ALLOC_OR_FAIL(t.a2.p, u->alloc_act((long)B * Lout, r.cout)); t.a2.ld = r.cout; t.a2.C = r.cout;
  if (r.ssn) {            // use_scale_shift_norm: hn = GroupNorm(h1) (no SiLU), a2 = SiLU(hn * (1 + scale) + shift); hn stays on the tape
    EEG_CHECK(r.emb_col >= 0, "use_scale_shift_norm needs the embedding projection");
    ALLOC_OR_FAIL(t.hn.p, u->alloc_act((long)B * Lout, r.cout)); t.hn.ld = r.cout; t.hn.C = r.cout;
    EEG_TRY(eegldm_groupnorm_fwd(ctx, t.h1.p, t.h1.ld, u->P(r.gn2_w), u->P(r.gn2_b), t.hn.p, t.hn.ld, t.st2, B, Lout, r.cout, r.groups, GN_EPS, 0,
                                 0, nullptr, 0, dt));
    EEG_TRY(ew_film_silu_fwd(ctx, t.hn.p, t.hn.ld, u->emb_all + r.emb_col, u->emb_ld, t.a2.p, t.a2.ld, B, Lout, r.cout, dt));
  } else
  EEG_TRY(eegldm_groupnorm_fwd(ctx, t.h1.p, t.h1.ld, u->P(r.gn2_w), u->P(r.gn2_b), t.a2.p, t.a2.ld, t.st2, B, Lout, r.cout, r.groups, GN_EPS, 1,
                               0, nullptr, 0, dt));
  if (u->dropout > 0.f && u->training) {      // out_layers[2] = nn.Dropout(p) between SiLU and conv2 (unet.py:286-294)
    t.dropped = true; t.drop_off = u->drop_ctr;
    EEG_TRY(ew_dropout_rows(ctx, t.a2.p, t.a2.ld, (long)B * Lout, r.cout, u->dropout, u->drop_seed, t.drop_off, dt));
    u->drop_ctr += ((unsigned long long)B * Lout * r.cout + 3) / 4;
  }
  if (r.sk_w >= 0) {      // skip_connection(x) + conv2(a2): one launch with the 1 x 1 conv as further K stages where the big-tile kernel takes it
    EEG_TRY(op_conv3_skip_fwd(ctx, dt, t.a2.p, t.a2.ld, u->W(r.c2_w), u->P(r.c2_b), t.xr.p, t.xr.ld, u->W(r.sk_w), u->P(r.sk_b), out.p, out.ld,
                              B, Lout, r.cout, r.cin, r.cout, nullptr, 0));
  } else {
    EEG_TRY(op_conv_fwd(ctx, dt, t.a2.p, t.a2.ld, u->W(r.c2_w), u->P(r.c2_b), out.p, out.ld, B, Lout, r.cout, r.cout, 3, 1, 1, 1, nullptr, 0, t.xr.p, t.xr.ld));
  }
  u->rt.push_back(t);
  return 0;
}

// dout: [B*Lout][cout]; writes dx: [B*Lin][cin]; accumulates parameter grads; demb_all gets per-sample sums
int res_backward(NetBase* u, const ResDesc& r, const ResTape& t, const View& dout, const View& dx, float* demb_all,
                 const View* extra, int* extra_done) {
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype; const int B = t.B, Lin = t.Lin, Lout = t.Lout;
  // deferred weight gradients (ctx->defer_wgrad): conv1's dY operand must outlive this block -> allocate it BELOW the release mark
  View dh1_keep;
  if (ctx->defer_wgrad && u->param_grads) { ALLOC_OR_FAIL(dh1_keep.p, u->alloc_act((long)B * t.Lout, r.cout)); dh1_keep.ld = r.cout; }
  Arena::Mark mk = u->arena.mark();
  const bool pg = u->param_grads;
  View dxr = dout;
  // weight gradients run on the side stream (pure GEMMs, no context scratch).  With 16-bit operands the GEMM also produces the
  // bias gradient (column sums of its dY operand); otherwise the bias sums are separate kernels on the main stream.
  const bool fb2 = pg && op_wgrad_fuses_bias(dt, r.cout, r.cout), fbs = pg && r.sk_w >= 0 && op_wgrad_fuses_bias(dt, r.cin, r.cout);
  const bool fb1 = pg && op_wgrad_fuses_bias(dt, r.cin, r.cout);
  if (pg) {
    EEG_TRY(ctx_fork(ctx));                       // dout (and everything before) is ready for the side stream
    SideScope side(ctx);
    EEG_TRY(u->flush_gn_folds());                 // the previous block's GN1 dgamma / dbeta fold, off the main chain
    if (r.sk_w >= 0) EEG_TRY(op_conv_wgrad(ctx, dt, t.xr.p, t.xr.ld, dout.p, dout.ld, u->G(r.sk_w), fbs ? u->G(r.sk_b) : nullptr, B, Lout, r.cin, r.cout, 1, 1, 0, 0));
    EEG_TRY(op_conv_wgrad(ctx, dt, t.a2.p, t.a2.ld, dout.p, dout.ld, u->G(r.c2_w), fb2 ? u->G(r.c2_b) : nullptr, B, Lout, r.cout, r.cout, 3, 1, 1, 1));
  }
  if (r.sk_w >= 0) {
    ALLOC_OR_FAIL(dxr.p, u->alloc_act((long)B * Lout, r.cin)); dxr.ld = r.cin; dxr.C = r.cin;
    EEG_TRY(op_conv_dgrad(ctx, dt, dout.p, dout.ld, u->W(r.sk_w), dxr.p, dxr.ld, B, Lout, r.cin, r.cout, 1, 1, 0, 0, nullptr, 0));
  }
  View da2; ALLOC_OR_FAIL(da2.p, u->alloc_act((long)B * Lout, r.cout)); da2.ld = r.cout;
  EEG_TRY(op_conv_dgrad(ctx, dt, dout.p, dout.ld, u->W(r.c2_w), da2.p, da2.ld, B, Lout, r.cout, r.cout, 3, 1, 1, 1, nullptr, 0));
  if (t.dropped) EEG_TRY(ew_dropout_rows(ctx, da2.p, da2.ld, (long)B * Lout, r.cout, u->dropout, u->drop_seed, t.drop_off, dt));      // the forward's mask, regenerated
  if (pg) {
    // conv2's bias gradient = column sums of dout; the skip conv's bias gradient is the same vector
    if (fb2 && (r.sk_w < 0 || fbs)) {
      // done inside the weight-gradient GEMMs
    } else if (r.sk_w >= 0) {
      float* tmp = (float*)((char*)ctx->scratch + (3u << 20));        // [cout] staging in the context scratch
      HIP_TRY(hipMemsetAsync(tmp, 0, sizeof(float) * r.cout, ctx->stream));
      EEG_TRY(ew_colsum(ctx, dout.p, dout.ld, nullptr, 0, tmp, B, Lout, r.cout, dt));
      // each bias takes the sum from exactly ONE producer: the side-stream weight-gradient GEMM when it fuses it, tmp otherwise
      if (!fb2) EEG_TRY(eegldm_axpy(ctx, u->G(r.c2_b), tmp, 1.0f, r.cout));
      if (!fbs) EEG_TRY(eegldm_axpy(ctx, u->G(r.sk_b), tmp, 1.0f, r.cout));
    } else {
      EEG_TRY(ew_colsum(ctx, dout.p, dout.ld, nullptr, 0, u->G(r.c2_b), B, Lout, r.cout, dt));
    }
  }
  View dh1 = dh1_keep;
  if (!dh1.p) { ALLOC_OR_FAIL(dh1.p, u->alloc_act((long)B * Lout, r.cout)); dh1.ld = r.cout; }
  // h1 = conv(a1) + b1 + emb_out[b]: the per-sample column sums of dh1 feed the embedding MLP, their total is db1.
  // The one-pass GroupNorm backward produces them while dh1 is still in registers; otherwise a separate column sum.
  float* ps = nullptr; long ldps = 0;
  const bool emb_add = r.emb_col >= 0 && !r.ssn;      // h1 = conv + b1 + emb: dh1's per-sample column sums are the embedding's gradient
  if (emb_add) { ps = demb_all + r.emb_col; ldps = u->etot; }
  else if (pg && !fb1) { ALLOC_OR_FAIL(ps, (float*)u->arena.alloc(sizeof(float) * (size_t)B * r.cout)); ldps = r.cout; }
  int cs_done = 0, gn2_deferred = 0;
  const void* gn2_dy = da2.p; long gn2_lddy = da2.ld;
  if (r.ssn) {            // a2 = SiLU(hn (1 + scale) + shift): d(hn) and the (scale, shift) gradient rows first, then GroupNorm without SiLU
    View dhn; ALLOC_OR_FAIL(dhn.p, u->alloc_act((long)B * Lout, r.cout)); dhn.ld = r.cout;
    EEG_TRY(ew_film_silu_bwd(ctx, t.hn.p, t.hn.ld, u->emb_all + r.emb_col, u->emb_ld, da2.p, da2.ld, dhn.p, dhn.ld, demb_all + r.emb_col, u->etot,
                             B, Lout, r.cout, dt));
    gn2_dy = dhn.p; gn2_lddy = dhn.ld;
  }
  EEG_TRY(op_groupnorm_bwd(ctx, t.h1.p, t.h1.ld, u->P(r.gn2_w), u->P(r.gn2_b), t.st2, gn2_dy, gn2_lddy, dh1.p, dh1.ld, pg ? u->G(r.gn2_w) : nullptr, pg ? u->G(r.gn2_b) : nullptr,
                           B, Lout, r.cout, r.groups, r.ssn ? 0 : 1, 0, nullptr, 0, dt, ps, ldps, &cs_done, nullptr, 0, nullptr, pg ? &gn2_deferred : nullptr));
  // few output channels = hundreds of K splits adding into the same bias entries (+17 us at 128 channels): when the one-pass
  // GroupNorm backward already produced per-sample sums, their total is cheaper
  const bool fb1e = fb1 && !(cs_done && ps && r.cout < 256);
  if (pg) {
    EEG_TRY(ctx_fork(ctx));                       // dh1 is ready
    SideScope side(ctx);
    if (gn2_deferred == 1) EEG_TRY(op_gn_slot_reduce_deferred(ctx, u->G(r.gn2_w), u->G(r.gn2_b), r.cout));   // dgamma / dbeta fold of GN2, off the main chain
    EEG_TRY(op_conv_wgrad(ctx, dt, t.a1.p, t.a1.ld, dh1.p, dh1.ld, u->G(r.c1_w), fb1e ? u->G(r.c1_b) : nullptr, B, Lout, r.cin, r.cout, 3, 1, 1, 1));
  }
  if (cs_done) {
    if (pg && !fb1e) EEG_TRY(ew_colsum(ctx, ps, ldps, nullptr, 0, u->G(r.c1_b), 1, B, r.cout, EEGLDM_F32));
  } else {
    if (emb_add) EEG_TRY(ew_colsum(ctx, dh1.p, dh1.ld, ps, ldps, pg && !fb1e ? u->G(r.c1_b) : nullptr, B, Lout, r.cout, dt));
    else if (pg && !fb1e) EEG_TRY(ew_colsum(ctx, dh1.p, dh1.ld, nullptr, 0, u->G(r.c1_b), B, Lout, r.cout, dt));
  }
  View da1; ALLOC_OR_FAIL(da1.p, u->alloc_act((long)B * Lout, r.cin)); da1.ld = r.cin;
  EEG_TRY(op_conv_dgrad(ctx, dt, dh1.p, dh1.ld, u->W(r.c1_w), da1.p, da1.ld, B, Lout, r.cin, r.cout, 3, 1, 1, 1, nullptr, 0));
  if (extra_done) *extra_done = 0;
  int gn1_deferred = 0;
  EEG_TRY(op_groupnorm_bwd(ctx, t.x.p, t.x.ld, u->P(r.gn1_w), u->P(r.gn1_b), t.st1, da1.p, da1.ld, dx.p, dx.ld, u->param_grads ? u->G(r.gn1_w) : nullptr, u->param_grads ? u->G(r.gn1_b) : nullptr,
                           B, Lin, r.cin, r.groups, 1, r.updown, dxr.p, dxr.ld, dt, nullptr, 0, nullptr,
                           extra ? extra->p : nullptr, extra ? extra->ld : 0, extra_done, pg ? &gn1_deferred : nullptr, 1 + u->gn_parity));
  if (gn1_deferred == 1) {   // folded inside the NEXT block's side-stream section (or by the executor's final flush); areas alternate, so the
    u->gn_pending.push_back({u->G(r.gn1_w), u->G(r.gn1_b), r.cin, 1 + u->gn_parity});   // block after that may write this one again
    u->gn_parity ^= 1;
  }
  if (pg) EEG_TRY(ctx_join(ctx));                 // the side stream's reads of dout / dh1 / tape are done before the arena is reused
  u->arena.release(mk);
  return 0;
}

// ------------------------------------------------------------------ AttentionBlock (unet.py:168-174)
int attn_forward(NetBase* u, const AttnDesc& a, const View& x, int B, int T, const View& out) {
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype; const int C = a.c; constexpr int AG = 32;
  AttnTape t; t.x = x; t.B = B; t.T = T;
  ALLOC_OR_FAIL(t.st, (float*)u->arena.alloc(sizeof(float) * 2 * B * AG));
  ALLOC_OR_FAIL(t.xn.p, u->alloc_act((long)B * T, C)); t.xn.ld = C;
  ALLOC_OR_FAIL(t.qkv.p, u->alloc_act((long)B * T, 3 * C)); t.qkv.ld = 3 * C;
  // eval, few rows: the preceding ResBlock's conv2 left this tensor's statistics; the qkv projection normalises on load (NetBase::eval_fuse)
  const NetBase::PartReg* in_part = u->eval_fuse ? u->find_part(x.p) : nullptr;
  int rcq = 0;
  if (in_part && in_part->nq * 4 == C) {
    const SkinnyGn gn = {in_part->slots, u->P(a.n_w), u->P(a.n_b), C / AG, GN_EPS, 0};
    rcq = conv_skinny_ex(ctx, dt, x.p, x.ld, u->W(a.qkv_w), C, 3 * C, 1, u->P(a.qkv_b), nullptr, 0, nullptr, 0, t.qkv.p, 3 * C, B, T, &gn, nullptr);
    if (rcq < 0) return rcq;
    if (rcq == 1) u->fused_used = true;
  }
  if (rcq != 1) {
    EEG_TRY(eegldm_groupnorm_fwd(ctx, x.p, x.ld, u->P(a.n_w), u->P(a.n_b), t.xn.p, C, t.st, B, T, C, AG, GN_EPS, 0, 0, nullptr, 0, dt));
    EEG_TRY(op_conv_fwd(ctx, dt, t.xn.p, C, u->W(a.qkv_w), u->P(a.qkv_b), t.qkv.p, 3 * C, B, T, C, 3 * C, 1, 1, 0, 0, nullptr, 0, nullptr, 0));
  }
  // heads (QKVAttentionLegacy, unet.py:107-125): head h owns columns [3 h ch, 3 (h + 1) ch) = [q | k | v] of the qkv rows and columns
  // [h ch, (h + 1) ch) of the output; one launch sequence per head on column views (the probabilities of all heads stay on the tape)
  const int H = a.heads, ch = C / H; const size_t es = dtype_size(dt);
  ALLOC_OR_FAIL(t.probs, u->alloc_act((long)H * B * T, T));
  ALLOC_OR_FAIL(t.o.p, u->alloc_act((long)B * T, C)); t.o.ld = C;
  Arena::Mark mk = u->arena.mark();
  float* logits; ALLOC_OR_FAIL(logits, (float*)u->arena.alloc(sizeof(float) * (size_t)B * T * T));
  for (int h = 0; h < H; h++)
    EEG_TRY(op_attention_fwd(ctx, dt, (const char*)t.qkv.p + (size_t)h * 3 * ch * es, 3 * C, (char*)t.o.p + (size_t)h * ch * es, C,
                             (char*)t.probs + (size_t)h * B * T * T * es, logits, B, T, ch));
  u->arena.release(mk);
  // eval, few rows: the projection leaves the statistics of the block output for the next ResBlock's first GroupNorm
  int rcp = 0;
  if (u->eval_fuse && T % 32 == 0 && conv_skinny_takes(dt, C, C, 1, B, T)) {
    float2* opart = u->fuse_alloc((size_t)B * (T / 16) * (C / 4));
    if (opart) {
      rcp = conv_skinny_ex(ctx, dt, t.o.p, C, u->W(a.pr_w), C, C, 1, u->P(a.pr_b), nullptr, 0, x.p, x.ld, out.p, out.ld, B, T, nullptr, opart);
      if (rcp < 0) return rcp;
      if (rcp == 1) u->part_reg.push_back({out.p, opart, C / 4});
    }
  }
  if (rcp != 1) {
    EEG_TRY(op_conv_fwd(ctx, dt, t.o.p, C, u->W(a.pr_w), u->P(a.pr_b), out.p, out.ld, B, T, C, C, 1, 1, 0, 0, nullptr, 0, x.p, x.ld));
  }
  u->at.push_back(t);
  return 0;
}

int attn_backward(NetBase* u, const AttnDesc& a, const AttnTape& t, const View& dout, const View& dx) {
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype; const int C = a.c, B = t.B, T = t.T; constexpr int AG = 32;
  void* dqkv_keep = nullptr;      // deferred weight gradients: the qkv conv's dY operand must outlive this block
  if (ctx->defer_wgrad && u->param_grads) ALLOC_OR_FAIL(dqkv_keep, u->alloc_act((long)B * T, 3 * C));
  Arena::Mark mk = u->arena.mark();
  EEG_TRY(op_conv_wgrad(ctx, dt, t.o.p, C, dout.p, dout.ld, u->G(a.pr_w), u->G(a.pr_b), B, T, C, C, 1, 1, 0, 0));
  void* d_o; ALLOC_OR_FAIL(d_o, u->alloc_act((long)B * T, C));
  EEG_TRY(op_conv_dgrad(ctx, dt, dout.p, dout.ld, u->W(a.pr_w), d_o, C, B, T, C, C, 1, 1, 0, 0, nullptr, 0));
  void* dqkv = dqkv_keep;
  if (!dqkv) ALLOC_OR_FAIL(dqkv, u->alloc_act((long)B * T, 3 * C));
  float* dprobs; ALLOC_OR_FAIL(dprobs, (float*)u->arena.alloc(sizeof(float) * (size_t)B * T * T));
  void* dlogits; ALLOC_OR_FAIL(dlogits, u->alloc_act((long)B * T, T));
  {
    const int H = a.heads, ch = C / H; const size_t es = dtype_size(dt);
    for (int h = 0; h < H; h++)
      EEG_TRY(op_attention_bwd(ctx, dt, (const char*)t.qkv.p + (size_t)h * 3 * ch * es, 3 * C, (const char*)t.probs + (size_t)h * B * T * T * es,
                               (const char*)d_o + (size_t)h * ch * es, C, (char*)dqkv + (size_t)h * 3 * ch * es, 3 * C, dprobs, dlogits, B, T, ch));
  }
  EEG_TRY(op_conv_wgrad(ctx, dt, t.xn.p, C, dqkv, 3 * C, u->G(a.qkv_w), u->G(a.qkv_b), B, T, C, 3 * C, 1, 1, 0, 0));
  void* dxn; ALLOC_OR_FAIL(dxn, u->alloc_act((long)B * T, C));
  EEG_TRY(op_conv_dgrad(ctx, dt, dqkv, 3 * C, u->W(a.qkv_w), dxn, C, B, T, C, 3 * C, 1, 1, 0, 0, nullptr, 0));
  // (dgamma / dbeta slot fold: batched with all the others in the grouped mode, else right here)
  int gn_deferred = 0;
  EEG_TRY(op_groupnorm_bwd(ctx, t.x.p, t.x.ld, u->P(a.n_w), u->P(a.n_b), t.st, dxn, C, dx.p, dx.ld, u->G(a.n_w), u->G(a.n_b),
                           B, T, C, AG, 0, 0, dout.p, dout.ld, dt, nullptr, 0, nullptr, nullptr, 0, nullptr, u->param_grads ? &gn_deferred : nullptr));
  if (gn_deferred == 1) EEG_TRY(op_gn_slot_reduce_deferred(ctx, u->G(a.n_w), u->G(a.n_b), C));
  u->arena.release(mk);
  return 0;
}


// ------------------------------------------------------------------ Downsample / Upsample layers (unet.py:177-224; resblock_updown = False)
int resample_forward(NetBase* u, const RsDesc& s, const View& x, int B, int Lin, const View& out, RsTape* tape) {
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype, C = s.c;
  RsTape t; t.x = x; t.B = B; t.Lin = Lin; t.Lout = s.up ? 2 * Lin : Lin / 2;
  if (!s.up) {
    EEG_CHECK(Lin % 2 == 0, "Downsample needs an even length (got %d)", Lin);
    if (s.conv) EEG_TRY(op_conv_fwd(ctx, dt, x.p, x.ld, u->W(s.w), u->P(s.b), out.p, out.ld, B, Lin, C, C, 3, 2, 1, 1, nullptr, 0, nullptr, 0));
    else EEG_TRY(eegldm_avgpool2_fwd(ctx, x.p, x.ld, out.p, out.ld, B, Lin, C, dt));
  } else if (s.conv) {
    ALLOC_OR_FAIL(t.xu.p, u->alloc_act((long)B * t.Lout, C)); t.xu.ld = C; t.xu.C = C;
    EEG_TRY(eegldm_nearest2_fwd(ctx, x.p, x.ld, t.xu.p, t.xu.ld, B, Lin, C, dt));
    EEG_TRY(op_conv_fwd(ctx, dt, t.xu.p, t.xu.ld, u->W(s.w), u->P(s.b), out.p, out.ld, B, t.Lout, C, C, 3, 1, 1, 1, nullptr, 0, nullptr, 0));
  } else {
    EEG_TRY(eegldm_nearest2_fwd(ctx, x.p, x.ld, out.p, out.ld, B, Lin, C, dt));
  }
  if (tape) *tape = t;
  return 0;
}
int resample_backward(NetBase* u, const RsDesc& s, const RsTape& t, const View& dout, const View& dx) {
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype, C = s.c, B = t.B;
  const bool pg = u->param_grads;
  if (!s.up) {
    if (s.conv) {
      if (pg) EEG_TRY(op_conv_wgrad(ctx, dt, t.x.p, t.x.ld, dout.p, dout.ld, u->G(s.w), u->G(s.b), B, t.Lin, C, C, 3, 2, 1, 1));
      EEG_TRY(op_conv_dgrad(ctx, dt, dout.p, dout.ld, u->W(s.w), dx.p, dx.ld, B, t.Lin, C, C, 3, 2, 1, 1, nullptr, 0));
    } else {
      EEG_TRY(eegldm_avgpool2_bwd(ctx, dout.p, dout.ld, dx.p, dx.ld, B, t.Lin, C, dt));
    }
  } else if (s.conv) {
    if (pg) EEG_TRY(op_conv_wgrad(ctx, dt, t.xu.p, t.xu.ld, dout.p, dout.ld, u->G(s.w), u->G(s.b), B, t.Lout, C, C, 3, 1, 1, 1));
    View dxu; ALLOC_OR_FAIL(dxu.p, u->alloc_act((long)B * t.Lout, C)); dxu.ld = C;       // (stays allocated: a deferred weight gradient may still read dout, not this)
    EEG_TRY(op_conv_dgrad(ctx, dt, dout.p, dout.ld, u->W(s.w), dxu.p, dxu.ld, B, t.Lout, C, C, 3, 1, 1, 1, nullptr, 0));
    EEG_TRY(eegldm_nearest2_bwd(ctx, dxu.p, dxu.ld, dx.p, dx.ld, B, t.Lin, C, dt));
  } else {
    EEG_TRY(eegldm_nearest2_bwd(ctx, dout.p, dout.ld, dx.p, dx.ld, B, t.Lin, C, dt));
  }
  return 0;
}
