// AutoencoderKL and PatchDiscriminator executors + the fused AEKL/GAN train-step body.
// Host code: sequences kernels of gemm.hip / direct_conv.hip / norm.hip / losses.hip / spectral.hip.
//
// Mirrors (structure per SURVEY.md Appendix B; MONAI-Generative source is absent, PARITY UNPINNED):
//   AutoencoderKL(spatial_dims=1, in/out_channels, num_channels, latent_channels, num_res_blocks,
//                 norm_num_groups, attention_levels all False)      config/config_aekl_eeg.yaml:19-28
//       encode / sampling / decode / forward                        twin /root/reference/src/models/ae_kl.py:259-291
//   PatchDiscriminator(spatial_dims=1, num_layers_d, num_channels, in/out_channels, kernel_size=3,
//                 norm="BATCH", bias=False, padding=1)              config/config_aekl_eeg.yaml:30-40
//   train step body                                                 /root/reference/src/train_autoencoderkl.py:200-234
#include <stdlib.h>

#include <string>
#include <vector>

#include "aekl_thin.h"
#include "net.h"

int ls_bn_lrelu_fwd(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta, float* stats, float* rmean, float* rvar,
                    float* nbt, void* y, long ldy, long rows, int C, float slope, int training, int dtype);
int ls_bn_lrelu_bwd(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta, const float* stats, const void* dy, long lddy,
                    void* dx, long lddx, float* dgamma, float* dbeta, long rows, int C, float slope, int dtype);
int ls_bn_stats(eegldm_ctx*, const void* x, long ldx, float* stats, float* rmean, float* rvar, float* nbt, long rows, int C, int training, int dtype);
int ls_bn_stats_from_parts(eegldm_ctx*, const float* parts, int nb, float* stats, float* rmean, float* rvar, float* nbt, long rows, int C);
int ls_bn_apply(eegldm_ctx*, const void* x, long ldx, const float* gamma, const float* beta, const float* stats, void* y, long ldy, long rows, int C, float slope, int dtype);
// fused tail of the discriminator (disc_tail.hip): BatchNorm + LeakyReLU of the last hidden layer + the one-channel final conv, forward and backward
bool disc_tail_ok(int dtype, int C, long ldy);
int disc_tail_fwd(eegldm_ctx*, int dtype, const void* y, long ldy, const float* gamma, const float* beta, const float* stats, const float* w3,
                  const float* bias, float slope, float* logits, int B, int L, int C);
int disc_tail_bwd(eegldm_ctx*, int dtype, const void* y, long ldy, const float* gamma, const float* beta, const float* stats, const float* w3,
                  float slope, const float* dlogits, void* dy, long lddy, float* dgamma, float* dbeta, float* dw3, float* dbias, int B, int L, int C);
// fused head of the discriminator (disc_tail.hip): LeakyReLU backward + the one-input-channel first conv's weight / bias / data gradients from da0 alone
bool disc_head_ok(int dtype, int C0, long ldda, int stride, int L, int Lo);
int disc_head_bwd(eegldm_ctx*, int dtype, const void* da, long ldda, const void* x, const void* w, const float* bias, float slope,
                  float* dw, float* db, float* dx, int B, int L, int Lo, int C0, int stride);
int ls_upsample2(eegldm_ctx*, const void* x, long ldx, void* y, long ldy, long rows_in, int C, int dtype);
int ls_upsample2_bwd(eegldm_ctx*, const void* dy, long lddy, void* dx, long lddx, long rows_in, int C, int dtype);
int ls_reparam(eegldm_ctx*, const void* mu, const void* lv, const float* eps, void* z, float* sigma, float* kl, long n, int B, int dtype);
int ls_reparam_bwd(eegldm_ctx*, const void* mu, const void* lv, const float* eps, const float* sigma, const void* dz, void* dmu, void* dlv, long n,
                   float klw_over_B, int dtype, const float* dmu_ext = nullptr, const float* dsg_ext = nullptr, int lat = 1, int Ll = 1);

namespace {

enum { OP_CONV = 0, OP_RES = 1, OP_UPS = 2, OP_GN = 3, OP_ACT = 4 };
struct Op {
  int kind = 0;
  int cin = 0, cout = 0, k = 3, stride = 1, pl = 1, pr = 1; long w = -1, b = -1;       // conv
  ResDesc r;                                                                           // res
  int groups = 1; long gw = -1, gb = -1;                                               // gn (no activation)
  long bn_w = -1, bn_b = -1, rm = -1, rv = -1, nbt = -1; float slope = 0.2f;           // act: bn_w < 0 -> plain LeakyReLU
};
struct OpTape { View x; float* st = nullptr; int Lin = 0, Lout = 0; };

struct SeqNet : NetBase {
  float* buffers = nullptr;     // BatchNorm running statistics (discriminator)
  int forward_seq(const std::vector<Op>& ops, View x, int B, int& L, View* out, std::vector<OpTape>& tape, int training);
  // keep_tape: the tape (and the activations it points to) stays valid for another backward over the same forward
  // first: stop in front of op `first` (its output gradient is returned in *dx; the caller handles ops [0, first) itself)
  int backward_seq(const std::vector<Op>& ops, std::vector<OpTape>& tape, int B, View dy, View* dx, bool need_dx, bool keep_tape = false, int first = 0);
};

int SeqNet::forward_seq(const std::vector<Op>& ops, View x, int B, int& L, View* out, std::vector<OpTape>& tape, int training) {
  const int dt = dtype;
  int bn_parts_n = 0;      // > 0: the conv that just ran left that many rows of column partials in the context scratch for the BatchNorm that follows
  for (size_t i = 0; i < ops.size(); i++) {
    const Op& o = ops[i];
    OpTape t; t.x = x; t.Lin = L;
    View y;
    if (o.kind == OP_CONV) {
      const int Lo = (L + o.pl + o.pr - o.k) / o.stride + 1;
      ALLOC_OR_FAIL(y.p, alloc_act((long)B * Lo, o.cout)); y.ld = o.cout; y.C = o.cout;
      // conv + plain LeakyReLU (the discriminator's first layer): the activation rides on the conv's store.  The tape then holds the
      // ACTIVATED tensor as the activation op's input: LeakyReLU keeps the sign, so its backward mask (x > 0) is unchanged.
      const bool fuse_act = i + 1 < ops.size() && ops[i + 1].kind == OP_ACT && ops[i + 1].bn_w < 0 && (long)B * Lo < (1L << 30) &&
                            op_conv_fuses_act(dt, o.cin, o.cout, o.k, y.ld);
      // a conv followed by a training-mode BatchNorm: kernels that can leave the column sums of their output do (conv_ws.hip), and the BatchNorm
      // below folds those instead of reading y again
      const bool bn_next = i + 1 < ops.size() && ops[i + 1].kind == OP_ACT && ops[i + 1].bn_w >= 0 && training != 0 && !eeg_deterministic();
      float* cparts = bn_next ? (float*)((char*)ctx->scratch + (8u << 20)) : nullptr;
      bn_parts_n = 0;
      EEG_TRY(op_conv_fwd(ctx, dt, x.p, x.ld, W(o.w), o.b >= 0 ? P(o.b) : nullptr, y.p, y.ld, B, L, o.cin, o.cout, o.k, o.stride, o.pl, o.pr,
                          nullptr, 0, nullptr, 0, fuse_act ? ops[i + 1].slope : 0.f, cparts, bn_next ? &bn_parts_n : nullptr));
      L = Lo;
      if (fuse_act) {
        t.Lout = L; tape.push_back(t);
        OpTape ta; ta.x = y; ta.Lin = L; ta.Lout = L; tape.push_back(ta);
        x = y; i++;
        continue;
      }
    } else if (o.kind == OP_RES) {
      ALLOC_OR_FAIL(y.p, alloc_act((long)B * L, o.r.cout)); y.ld = o.r.cout; y.C = o.r.cout;
      EEG_TRY(res_forward(this, o.r, x, B, L, y));
    } else if (o.kind == OP_UPS) {
      ALLOC_OR_FAIL(y.p, alloc_act((long)B * 2 * L, x.C)); y.ld = x.C; y.C = x.C;
      EEG_TRY(ls_upsample2(ctx, x.p, x.ld, y.p, y.ld, (long)B * L, x.C, dt));
      L *= 2;
    } else if (o.kind == OP_GN) {
      ALLOC_OR_FAIL(y.p, alloc_act((long)B * L, x.C)); y.ld = x.C; y.C = x.C;
      ALLOC_OR_FAIL(t.st, arena.alloc(sizeof(float) * 2 * B * o.groups));
      EEG_TRY(eegldm_groupnorm_fwd(ctx, x.p, x.ld, P(o.gw), P(o.gb), y.p, y.ld, t.st, B, L, x.C, o.groups, GN_EPS, 0, 0, nullptr, 0, dt));
    } else {  // OP_ACT
      ALLOC_OR_FAIL(y.p, alloc_act((long)B * L, x.C)); y.ld = x.C; y.C = x.C;
      if (o.bn_w >= 0) {
        ALLOC_OR_FAIL(t.st, arena.alloc(sizeof(float) * 2 * x.C));
        const bool upd = buffers && training != 2;      // training == 2: batch statistics, running statistics left alone (a re-forward for a second backward)
        if (bn_parts_n > 0 && training != 0) {      // the conv in front of this layer left the column sums of x (= its output)
          EEG_TRY(ls_bn_stats_from_parts(ctx, (const float*)((char*)ctx->scratch + (8u << 20)), bn_parts_n, t.st, upd ? buffers + o.rm : nullptr,
                                         upd ? buffers + o.rv : nullptr, upd ? buffers + o.nbt : nullptr, (long)B * L, x.C));
          EEG_TRY(ls_bn_apply(ctx, x.p, x.ld, P(o.bn_w), P(o.bn_b), t.st, y.p, y.ld, (long)B * L, x.C, o.slope, dt));
        } else
        EEG_TRY(ls_bn_lrelu_fwd(ctx, x.p, x.ld, P(o.bn_w), P(o.bn_b), t.st, upd ? buffers + o.rm : nullptr, upd ? buffers + o.rv : nullptr,
                                upd ? buffers + o.nbt : nullptr, y.p, y.ld, (long)B * L, x.C, o.slope, training, dt));
        bn_parts_n = 0;
      } else {
        EEG_TRY(ls_bn_lrelu_fwd(ctx, x.p, x.ld, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, y.p, y.ld, (long)B * L, x.C, o.slope, training, dt));
      }
    }
    t.Lout = L;
    tape.push_back(t);
    x = y;
  }
  *out = x;
  return 0;
}

int SeqNet::backward_seq(const std::vector<Op>& ops, std::vector<OpTape>& tape, int B, View dy, View* dx_out, bool need_dx, bool keep_tape, int first) {
  const int dt = dtype;
  for (int i = (int)ops.size() - 1; i >= first; i--) {
    const Op& o = ops[i];
    const OpTape t = keep_tape ? tape[tape.size() - (ops.size() - i)] : tape.back();
    if (!keep_tape) tape.pop_back();
    const bool want_dx = need_dx || i > first;
    View dx; dx.ld = t.x.C; dx.C = t.x.C;
    if (want_dx) ALLOC_OR_FAIL(dx.p, alloc_act((long)B * t.Lin, t.x.C));
    if (o.kind == OP_CONV) {
      if (param_grads)
        EEG_TRY(op_conv_wgrad(ctx, dt, t.x.p, t.x.ld, dy.p, dy.ld, G(o.w), o.b >= 0 ? G(o.b) : nullptr, B, t.Lin, o.cin, o.cout, o.k, o.stride, o.pl, o.pr));
      if (want_dx) EEG_TRY(op_conv_dgrad(ctx, dt, dy.p, dy.ld, W(o.w), dx.p, dx.ld, B, t.Lin, o.cin, o.cout, o.k, o.stride, o.pl, o.pr, nullptr, 0));
    } else if (o.kind == OP_RES) {
      // res_backward always produces dx (cheap relative to the block); allocate if the caller did not want it
      if (!want_dx) ALLOC_OR_FAIL(dx.p, alloc_act((long)B * t.Lin, t.x.C));
      const ResTape rtp = rt.back(); rt.pop_back();
      EEG_TRY(res_backward(this, o.r, rtp, dy, dx, nullptr));
    } else if (o.kind == OP_UPS) {
      if (want_dx) EEG_TRY(ls_upsample2_bwd(ctx, dy.p, dy.ld, dx.p, dx.ld, (long)B * t.Lin, t.x.C, dt));
    } else if (o.kind == OP_GN) {
      if (!want_dx) ALLOC_OR_FAIL(dx.p, alloc_act((long)B * t.Lin, t.x.C));
      EEG_TRY(eegldm_groupnorm_bwd(ctx, t.x.p, t.x.ld, P(o.gw), P(o.gb), t.st, dy.p, dy.ld, dx.p, dx.ld, param_grads ? G(o.gw) : nullptr,
                                   param_grads ? G(o.gb) : nullptr, B, t.Lin, t.x.C, o.groups, 0, 0, nullptr, 0, dt));
    } else {
      if (!want_dx) ALLOC_OR_FAIL(dx.p, alloc_act((long)B * t.Lin, t.x.C));
      if (o.bn_w >= 0)
        EEG_TRY(ls_bn_lrelu_bwd(ctx, t.x.p, t.x.ld, P(o.bn_w), P(o.bn_b), t.st, dy.p, dy.ld, dx.p, dx.ld, param_grads ? G(o.bn_w) : nullptr,
                                param_grads ? G(o.bn_b) : nullptr, (long)B * t.Lin, t.x.C, o.slope, dt));
      else
        EEG_TRY(ls_bn_lrelu_bwd(ctx, t.x.p, t.x.ld, nullptr, nullptr, nullptr, dy.p, dy.ld, dx.p, dx.ld, nullptr, nullptr, (long)B * t.Lin, t.x.C, o.slope, dt));
    }
    dy = dx;
  }
  EEG_TRY(flush_gn_folds());   // deferred dgamma / dbeta fold of the last ResBlock (net.hip)
  if (dx_out) *dx_out = dy;
  return 0;
}

struct Layout {
  long off = 0;
  long take(long n) { long o = off; off += (n + 7) / 8 * 8; return o; }
};

Op make_conv(NetBase* n, Layout& lay, const std::string& p, int cin, int cout, int k, int stride, int pl, int pr, bool bias) {
  Op o; o.kind = OP_CONV; o.cin = cin; o.cout = cout; o.k = k; o.stride = stride; o.pl = pl; o.pr = pr;
  o.w = lay.take((long)cout * cin * k); n->add_entry(p + ".weight", o.w, 3, cout, cin, k);
  if (bias) { o.b = lay.take(cout); n->add_entry(p + ".bias", o.b, 1, cout); }
  return o;
}
Op make_res(NetBase* n, Layout& lay, const std::string& p, int cin, int cout, int groups) {
  Op o; o.kind = OP_RES; ResDesc& r = o.r;
  r.cin = cin; r.cout = cout; r.updown = 0; r.groups = groups; r.emb_col = -1;
  r.gn1_w = lay.take(cin); r.gn1_b = lay.take(cin);
  n->add_entry(p + ".norm1.weight", r.gn1_w, 1, cin); n->add_entry(p + ".norm1.bias", r.gn1_b, 1, cin);
  r.c1_w = lay.take((long)cout * cin * 3); r.c1_b = lay.take(cout);
  n->add_entry(p + ".conv1.conv.weight", r.c1_w, 3, cout, cin, 3); n->add_entry(p + ".conv1.conv.bias", r.c1_b, 1, cout);
  r.gn2_w = lay.take(cout); r.gn2_b = lay.take(cout);
  n->add_entry(p + ".norm2.weight", r.gn2_w, 1, cout); n->add_entry(p + ".norm2.bias", r.gn2_b, 1, cout);
  r.c2_w = lay.take((long)cout * cout * 3); r.c2_b = lay.take(cout);
  n->add_entry(p + ".conv2.conv.weight", r.c2_w, 3, cout, cout, 3); n->add_entry(p + ".conv2.conv.bias", r.c2_b, 1, cout);
  r.sk_w = r.sk_b = -1;
  if (cin != cout) {
    r.sk_w = lay.take((long)cout * cin); r.sk_b = lay.take(cout);
    n->add_entry(p + ".nin_shortcut.conv.weight", r.sk_w, 3, cout, cin, 1); n->add_entry(p + ".nin_shortcut.conv.bias", r.sk_b, 1, cout);
  }
  return o;
}
Op make_gn(NetBase* n, Layout& lay, const std::string& p, int c, int groups) {
  Op o; o.kind = OP_GN; o.groups = groups; o.cin = o.cout = c;
  o.gw = lay.take(c); o.gb = lay.take(c);
  n->add_entry(p + ".weight", o.gw, 1, c); n->add_entry(p + ".bias", o.gb, 1, c);
  return o;
}
}  // namespace

// ================================================================== AutoencoderKL
struct eegldm_aekl;
struct eegldm_aekl : SeqNet {
  eegldm_aekl_cfg cfg;
  std::vector<Op> enc, dec; Op q_mu, q_lv;
  std::vector<OpTape> tape_enc, tape_dec;
  Arena stage;                      // fp32 NCL staging that must survive arena resets (train step)
  // tape of the latent head
  int B = 0, L = 0, Ll = 0; bool have_tape = false;
  View h_enc, mu, lv; float *eps_nlc = nullptr, *sigma = nullptr;
  // whole-network path for thin configurations (aekl_thin.hip): program compiled per window length
  ThinProgram thin; int thin_L = -1; bool thin_tape = false; float* thin_eps = nullptr;
  int thin_fail_L = -1;          // length for which the whole-network program did not fit (LDS footprint): use the general path
  ~eegldm_aekl() { thin_free(&thin); }
};

namespace {
// ---- thin whole-network path: compile enc / heads / dec into micro-ops over four LDS tensors (aekl_thin.h)
bool thin_eligible(const eegldm_aekl* a, int L) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_AEKL_NO_THIN") != nullptr);
  const eegldm_aekl_cfg& c = a->cfg;
  auto ok_c = [](int v) { return v == 1 || v == 2 || v == 4; };           // the kernels are specialised on channel counts of 1, 2 and 4
  if (off || c.norm_num_groups != 1 || !ok_c(c.in_channels) || !ok_c(c.out_channels) || !ok_c(c.latent_channels)) return false;
  int mx = 0;
  for (int i = 0; i < c.n_levels; i++) { if (!ok_c(c.num_channels[i])) return false; const int v = c.num_channels[i] * (L >> i); mx = v > mx ? v : mx; }
  if ((L >> (c.n_levels - 1)) % 4 != 0) return false;                      // float4 rows at every level
  return (long)mx <= THIN_MAX_FLOATS - 1024 && a->nparams <= 4096 && (long)c.in_channels * L <= THIN_MAX_FLOATS && (L % (1 << (c.n_levels - 1))) == 0;
}

struct ThinBuilder {
  ThinProgram& p; int nslot = 0;
  struct Seg { int kind; Op op; int L, Lout, slot, slot2, stat1, stat2, ups_slot, C; };   // kind: 0 conv, 1 res, 2 gn, 3 heads, 4 conv-after-upsample
  std::vector<Seg> segs;
  explicit ThinBuilder(ThinProgram& pp) : p(pp) {}
  static int pick(int a, int b = -1, int c = -1) { for (int i = 0; i < THIN_NBUF; i++) if (i != a && i != b && i != c) return i; return -1; }
  void track(int n) { n = (n + 3) & ~3; if (n > p.maxt) p.maxt = n; }
  int new_slot(int n) { p.tape_off.push_back(p.tape_stride); p.tape_stride += (n + 3) & ~3; return nslot++; }
  ThinOp blank(int kind) { ThinOp o = {}; o.kind = kind; o.src = o.dst = o.add = o.act = -1; o.b = o.b2 = -1; o.save = -1; o.k = 1; o.stride = 1; return o; }
  void f_save(int buf, int C, int L, int slot) { ThinOp o = blank(TF_SAVE); o.src = buf; o.cin = C; o.Lin = L; o.save = slot; p.fwd.push_back(o); }
  ThinOp conv_op(int kind, const Op& c, int Lin, int Lout) {
    ThinOp o = blank(kind); o.cin = c.cin; o.cout = c.cout; o.Lin = Lin; o.Lout = Lout; o.k = c.k; o.stride = c.stride; o.pad_l = c.pl; o.w = (int)c.w; o.b = (int)c.b; o.need_dx = 1;
    track(c.cin * Lin); track(c.cout * Lout);
    return o;
  }
  ThinOp gn_op(int kind, long gw, long gb, int C, int L, int silu, int stat) {
    ThinOp o = blank(kind); o.cin = o.cout = C; o.Lin = o.Lout = L; o.gw = (int)gw; o.gb = (int)gb; o.silu = silu; o.stat = stat; track(C * L);
    return o;
  }
  // one conv described by (w, b, cin, cout, k) of a ResDesc
  static Op res_conv(const ResDesc& r, int which) {
    Op c; c.kind = OP_CONV; c.stride = 1;
    if (which == 0) { c.cin = r.cin; c.cout = r.cout; c.k = 3; c.pl = c.pr = 1; c.w = r.c1_w; c.b = r.c1_b; }
    else if (which == 1) { c.cin = r.cout; c.cout = r.cout; c.k = 3; c.pl = c.pr = 1; c.w = r.c2_w; c.b = r.c2_b; }
    else { c.cin = r.cin; c.cout = r.cout; c.k = 1; c.pl = c.pr = 0; c.w = r.sk_w; c.b = r.sk_b; }
    return c;
  }
  // forward over an op list; cur = buffer holding the current tensor (C x L)
  void forward_ops(const std::vector<Op>& ops, int& cur, int& C, int& L) {
    bool after_ups = false; int ups_slot = -1;
    for (const Op& o : ops) {
      if (o.kind == OP_CONV) {
        const int Lo = (L + o.pl + o.pr - o.k) / o.stride + 1;
        Seg sg = {}; sg.kind = after_ups ? 4 : 0; sg.op = o; sg.L = L; sg.Lout = Lo; sg.ups_slot = ups_slot;
        if (!after_ups) { sg.slot = new_slot(C * L); f_save(cur, C, L, sg.slot); }
        ThinOp t = conv_op(TF_CONV, o, L, Lo); t.src = cur; t.dst = pick(cur); p.fwd.push_back(t);
        cur = t.dst; C = o.cout; L = Lo; after_ups = false; segs.push_back(sg);
      } else if (o.kind == OP_RES) {
        const ResDesc& r = o.r;
        Seg sg = {}; sg.kind = 1; sg.op = o; sg.L = L; sg.slot = new_slot(r.cin * L); sg.slot2 = new_slot(r.cout * L); sg.stat1 = p.nstat++; sg.stat2 = p.nstat++;
        f_save(cur, r.cin, L, sg.slot);
        const int t1 = pick(cur), t2 = pick(cur, t1), t3 = pick(cur, t1, t2);
        { ThinOp g = gn_op(TF_GN, r.gn1_w, r.gn1_b, r.cin, L, 1, sg.stat1); g.src = cur; g.dst = t1; p.fwd.push_back(g); }
        { ThinOp c = conv_op(TF_CONV, res_conv(r, 0), L, L); c.src = t1; c.dst = t2; p.fwd.push_back(c); }
        f_save(t2, r.cout, L, sg.slot2);
        { ThinOp g = gn_op(TF_GN, r.gn2_w, r.gn2_b, r.cout, L, 1, sg.stat2); g.src = t2; g.dst = t1; p.fwd.push_back(g); }
        if (r.sk_w >= 0) {
          { ThinOp c = conv_op(TF_CONV, res_conv(r, 2), L, L); c.src = cur; c.dst = t2; p.fwd.push_back(c); }
          { ThinOp c = conv_op(TF_CONV, res_conv(r, 1), L, L); c.src = t1; c.dst = t3; c.add = t2; p.fwd.push_back(c); }
          cur = t3;
        } else {
          { ThinOp c = conv_op(TF_CONV, res_conv(r, 1), L, L); c.src = t1; c.dst = t2; c.add = cur; p.fwd.push_back(c); }
          cur = t2;
        }
        C = r.cout; segs.push_back(sg);
      } else if (o.kind == OP_UPS) {
        ups_slot = new_slot(C * L); f_save(cur, C, L, ups_slot);
        ThinOp u = blank(TF_UPS); u.cin = u.cout = C; u.Lin = L; u.Lout = 2 * L; u.src = cur; u.dst = pick(cur); track(C * 2 * L); p.fwd.push_back(u);
        cur = u.dst; L *= 2; after_ups = true;
      } else if (o.kind == OP_GN) {
        Seg sg = {}; sg.kind = 2; sg.op = o; sg.L = L; sg.C = C; sg.slot = new_slot(C * L); sg.stat1 = p.nstat++;
        f_save(cur, C, L, sg.slot);
        ThinOp g = gn_op(TF_GN, o.gw, o.gb, C, L, 0, sg.stat1); g.src = cur; g.dst = pick(cur); p.fwd.push_back(g);
        cur = g.dst; segs.push_back(sg);
      }
    }
  }
  // backward over the recorded segments (reverse); g = buffer holding the gradient of the segment's output
  void backward_segs(int& g) {
    for (int i = (int)segs.size() - 1; i >= 0; i--) {
      const Seg& sg = segs[i];
      if (sg.kind == 0) {
        const int a = pick(g);
        { ThinOp l = blank(TB_LOADT); l.dst = a; l.cin = sg.op.cin; l.Lin = sg.L; l.save = sg.slot; p.bwd.push_back(l); }
        { ThinOp c = conv_op(TB_CONV, sg.op, sg.L, sg.Lout); c.src = g; c.act = a; p.bwd.push_back(c); }
        g = a;
      } else if (sg.kind == 4) {       // nearest x2 then conv: the saved tensor is the PRE-upsample input
        const int a = pick(g), t = pick(g, a), Lpre = sg.L / 2, Cc = sg.op.cin;
        { ThinOp l = blank(TB_LOADT); l.dst = a; l.cin = Cc; l.Lin = Lpre; l.save = sg.ups_slot; p.bwd.push_back(l); }
        { ThinOp u = blank(TB_UPS); u.cin = u.cout = Cc; u.Lin = Lpre; u.Lout = sg.L; u.src = a; u.dst = t; p.bwd.push_back(u); }
        { ThinOp c = conv_op(TB_CONV, sg.op, sg.L, sg.Lout); c.src = g; c.act = t; p.bwd.push_back(c); }
        { ThinOp u = blank(TB_UPSBWD); u.cin = u.cout = Cc; u.Lin = sg.L; u.Lout = Lpre; u.src = t; u.dst = a; p.bwd.push_back(u); }
        g = a;
      } else if (sg.kind == 2) {
        const int a = pick(g);
        { ThinOp l = blank(TB_LOADT); l.dst = a; l.cin = sg.C; l.Lin = sg.L; l.save = sg.slot; p.bwd.push_back(l); }
        { ThinOp n = gn_op(TB_GN, sg.op.gw, sg.op.gb, sg.C, sg.L, 0, sg.stat1); n.act = a; n.src = g; n.dst = g; p.bwd.push_back(n); }
      } else if (sg.kind == 1) {
        const ResDesc& r = sg.op.r; const int L = sg.L;
        int s = pick(g), a = pick(g, s), t = pick(g, s, a);
        if (r.sk_w >= 0) {
          { ThinOp l = blank(TB_LOADT); l.dst = s; l.cin = r.cin; l.Lin = L; l.save = sg.slot; p.bwd.push_back(l); }
          { ThinOp c = conv_op(TB_CONV, res_conv(r, 2), L, L); c.src = g; c.act = s; p.bwd.push_back(c); }      // s <- gradient through the shortcut
        } else {
          ThinOp c = blank(TB_COPY); c.src = g; c.dst = s; c.cin = r.cout; c.Lin = L; p.bwd.push_back(c);
        }
        { ThinOp l = blank(TB_LOADT); l.dst = a; l.cin = r.cout; l.Lin = L; l.save = sg.slot2; p.bwd.push_back(l); }
        { ThinOp n = gn_op(TB_RECOMP, r.gn2_w, r.gn2_b, r.cout, L, 1, sg.stat2); n.src = a; n.dst = t; p.bwd.push_back(n); }
        { ThinOp c = conv_op(TB_CONV, res_conv(r, 1), L, L); c.src = g; c.act = t; p.bwd.push_back(c); }
        { ThinOp n = gn_op(TB_GN, r.gn2_w, r.gn2_b, r.cout, L, 1, sg.stat2); n.act = a; n.src = t; n.dst = g; p.bwd.push_back(n); }
        { ThinOp l = blank(TB_LOADT); l.dst = a; l.cin = r.cin; l.Lin = L; l.save = sg.slot; p.bwd.push_back(l); }
        { ThinOp n = gn_op(TB_RECOMP, r.gn1_w, r.gn1_b, r.cin, L, 1, sg.stat1); n.src = a; n.dst = t; p.bwd.push_back(n); }
        { ThinOp c = conv_op(TB_CONV, res_conv(r, 0), L, L); c.src = g; c.act = t; p.bwd.push_back(c); }
        { ThinOp n = gn_op(TB_GN, r.gn1_w, r.gn1_b, r.cin, L, 1, sg.stat1); n.act = a; n.src = t; n.dst = g; n.add = s; p.bwd.push_back(n); }
      } else if (sg.kind == 3) {
        const int a = pick(g);
        { ThinOp l = blank(TB_LOADT); l.dst = a; l.cin = sg.C; l.Lin = sg.L; l.save = sg.slot; p.bwd.push_back(l); }
        ThinOp h = blank(TB_HEADS); h.cin = h.cout = sg.C; h.Lin = h.Lout = sg.L; h.src = g; h.dst = g; h.act = a; h.save = sg.slot2;
        h.w = (int)sg.op.w; h.b = (int)sg.op.b; h.w2 = sg.stat1; h.b2 = sg.stat2; p.bwd.push_back(h);
      }
    }
  }
};

int thin_build(eegldm_aekl* a, int L) {
  ThinProgram& p = a->thin;
  thin_free(&p);
  p = ThinProgram();
  const eegldm_aekl_cfg& c = a->cfg;
  ThinBuilder bld(p);
  p.nparams = (int)a->nparams;
  int cur = 0, C = c.in_channels, Lc = L;
  { ThinOp l = bld.blank(TF_LOAD); l.dst = 0; l.cin = C; l.Lin = L; bld.track(C * L); p.fwd.push_back(l); }
  bld.forward_ops(a->enc, cur, C, Lc);
  // heads: mu / log-variance 1x1 convs, clamp, sigma, z = mu + eps * sigma (+ KL)
  {
    ThinBuilder::Seg sg = {}; sg.kind = 3; sg.C = C; sg.L = Lc; sg.slot = bld.new_slot(C * Lc); sg.slot2 = bld.new_slot(C * Lc); bld.new_slot(C * Lc);   // h, mu, lv
    sg.op.w = a->q_mu.w; sg.op.b = a->q_mu.b; sg.stat1 = (int)a->q_lv.w; sg.stat2 = (int)a->q_lv.b;
    bld.f_save(cur, C, Lc, sg.slot);
    ThinOp h = bld.blank(TF_HEADS); h.cin = h.cout = C; h.Lin = h.Lout = Lc; h.src = cur; h.dst = ThinBuilder::pick(cur); h.save = sg.slot2;
    h.w = (int)a->q_mu.w; h.b = (int)a->q_mu.b; h.w2 = (int)a->q_lv.w; h.b2 = (int)a->q_lv.b; p.fwd.push_back(h);
    cur = h.dst; bld.segs.push_back(sg);
    p.lat = C; p.Ll = Lc;
  }
  bld.forward_ops(a->dec, cur, C, Lc);
  { ThinOp s = bld.blank(TF_STORE); s.src = cur; s.cin = C; s.Lin = Lc; p.fwd.push_back(s); }
  // backward program
  int g = 0;
  { ThinOp l = bld.blank(TB_LOADDY); l.dst = 0; l.cin = C; l.Lin = Lc; p.bwd.push_back(l); }
  bld.backward_segs(g);
  { ThinOp s = bld.blank(TB_STOREDX); s.src = g; s.cin = c.in_channels; s.Lin = L; p.bwd.push_back(s); }
  EEG_TRY(thin_upload(&p));
  a->thin_L = L;
  return 0;
}
// thin_eligible() bounds the tensors and the parameter count but not the real LDS footprint (four tensors + parameters + the op table,
// aekl_thin.hip::lds_bytes): a configuration can pass it and still not fit.  That is "not eligible", not an error -- remember the
// length and let the caller take the layer-by-layer path.
bool thin_ready(eegldm_aekl* a, int L) {
  if (!thin_eligible(a, L) || a->thin_fail_L == L) return false;
  if (a->thin_L == L) return true;
  if (thin_build(a, L) == 0) return true;
  a->thin_fail_L = L; a->thin_L = -1;
  return false;
}
}  // namespace

extern "C" int eegldm_aekl_create(eegldm_ctx* ctx, const eegldm_aekl_cfg* cfg, eegldm_aekl** out) {
  EEG_CHECK(ctx && cfg && out, "null argument");
  EEG_CHECK(cfg->n_levels >= 1 && cfg->n_levels <= 8 && cfg->num_res_blocks >= 1, "bad num_channels / num_res_blocks");
  EEG_CHECK(cfg->dtype == EEGLDM_F32 || cfg->dtype == EEGLDM_BF16 || cfg->dtype == EEGLDM_F16, "bad dtype");
  for (int i = 0; i < cfg->n_levels; i++)
    EEG_CHECK(cfg->num_channels[i] % cfg->norm_num_groups == 0, "num_channels[%d]=%d not divisible by norm_num_groups=%d", i, cfg->num_channels[i], cfg->norm_num_groups);
  eegldm_aekl* a = new eegldm_aekl();
  a->ctx = ctx; a->cfg = *cfg; a->dtype = cfg->dtype;
  a->arena.min_block = (size_t)256 << 20; a->stage.min_block = (size_t)32 << 20;
  const int nl = cfg->n_levels, G = cfg->norm_num_groups, lat = cfg->latent_channels;
  const int* nc = cfg->num_channels;
  Layout lay;
  // encoder (MONAI Encoder: conv, [res x nrb, downsample] per level, norm, conv)
  int bi = 0;
  auto ename = [&](const char* s) { return std::string("encoder.blocks.") + std::to_string(bi) + s; };
  a->enc.push_back(make_conv(a, lay, ename(".conv"), cfg->in_channels, nc[0], 3, 1, 1, 1, true)); bi++;
  int oc = nc[0];
  for (int i = 0; i < nl; i++) {
    int ic = oc; oc = nc[i];
    for (int r = 0; r < cfg->num_res_blocks; r++) { a->enc.push_back(make_res(a, lay, ename(""), ic, oc, G)); bi++; ic = oc; }
    if (i != nl - 1) { a->enc.push_back(make_conv(a, lay, ename(".conv.conv"), ic, ic, 3, 2, 0, 1, true)); bi++; }   // pad (0,1), stride 2
  }
  a->enc.push_back(make_gn(a, lay, ename(""), oc, G)); bi++;
  a->enc.push_back(make_conv(a, lay, ename(".conv"), oc, lat, 3, 1, 1, 1, true)); bi++;
  // decoder (conv, [res x nrb, upsample] per level reversed, norm, conv); post_quant_conv is prepended at run time
  bi = 0;
  auto dname = [&](const char* s) { return std::string("decoder.blocks.") + std::to_string(bi) + s; };
  std::vector<Op> dec;
  dec.push_back(make_conv(a, lay, dname(".conv"), lat, nc[nl - 1], 3, 1, 1, 1, true)); bi++;
  oc = nc[nl - 1];
  for (int i = nl - 1; i >= 0; i--) {
    int ic = oc; oc = nc[i];
    for (int r = 0; r < cfg->num_res_blocks; r++) { dec.push_back(make_res(a, lay, dname(""), ic, oc, G)); bi++; ic = oc; }
    if (i != 0) {
      Op u; u.kind = OP_UPS; dec.push_back(u);
      dec.push_back(make_conv(a, lay, dname(".conv.conv"), ic, ic, 3, 1, 1, 1, true)); bi++;
    }
  }
  dec.push_back(make_gn(a, lay, dname(""), oc, G)); bi++;
  dec.push_back(make_conv(a, lay, dname(".conv"), oc, cfg->out_channels, 3, 1, 1, 1, true)); bi++;
  a->q_mu = make_conv(a, lay, "quant_conv_mu.conv", lat, lat, 1, 1, 0, 0, true);
  a->q_lv = make_conv(a, lay, "quant_conv_log_sigma.conv", lat, lat, 1, 1, 0, 0, true);
  a->dec.push_back(make_conv(a, lay, "post_quant_conv.conv", lat, lat, 1, 1, 0, 0, true));
  for (auto& o : dec) a->dec.push_back(o);
  a->nparams = lay.off;
  *out = a;
  return 0;
}
eegldm_ctx* aekl_ctx(const eegldm_aekl* a) { return a->ctx; }
extern "C" int eegldm_aekl_destroy(eegldm_aekl* a) { delete a; return 0; }
extern "C" int eegldm_aekl_num_entries(const eegldm_aekl* a) { return (int)a->entries.size(); }
extern "C" long eegldm_aekl_num_params(const eegldm_aekl* a) { return a->nparams; }
extern "C" int eegldm_aekl_entry(const eegldm_aekl* a, int i, char* name, int cap, long* offset, long* numel, int* ndim, int shape[3]) {
  return entry_query(a, i, name, cap, offset, numel, ndim, shape);
}
extern "C" int eegldm_aekl_bind(eegldm_aekl* a, float* params, float* grads) { EEG_CHECK(a && params, "null argument"); return a->bind(params, grads); }
extern "C" int eegldm_aekl_sync_weights(eegldm_aekl* a) { EEG_CHECK(a, "null argument"); return a->sync_weights(); }

namespace {
int lat_len(const eegldm_aekl* a, int L) { return L >> (a->cfg.n_levels - 1); }

// encoder + heads + sampling.  Leaves mu / lv / sigma / z in the arena; returns z view.
bool enc_fused_eligible(const eegldm_aekl* a, int L);
int aekl_encode_fused_seq(eegldm_aekl* a, View x, int B, int& L, View* out);
int aekl_encode_impl(eegldm_aekl* a, const float* x, const float* eps, int B, int L, View* z_out, float* kl, bool frozen = false) {
  EEG_CHECK((L % (1 << (a->cfg.n_levels - 1))) == 0, "L=%d must be divisible by 2^(levels-1)", L);
  eegldm_ctx* ctx = a->ctx; const int dt = a->dtype, lat = a->cfg.latent_channels, cin = a->cfg.in_channels;
  a->arena.reset(); a->rt.clear(); a->tape_enc.clear(); a->tape_dec.clear();
  a->B = B; a->L = L; a->have_tape = false;
  View x0; ALLOC_OR_FAIL(x0.p, a->alloc_act((long)B * L, cin)); x0.ld = cin; x0.C = cin;
  EEG_TRY(eegldm_ncl_to_nlc(ctx, x, x0.p, cin, B, cin, L, dt));
  int Lc = L;
  if (frozen && enc_fused_eligible(a, L)) EEG_TRY(aekl_encode_fused_seq(a, x0, B, Lc, &a->h_enc));      // no tape: the caller never back-propagates
  else EEG_TRY(a->forward_seq(a->enc, x0, B, Lc, &a->h_enc, a->tape_enc, 1));
  a->Ll = Lc;
  const long n = (long)B * Lc * lat;
  ALLOC_OR_FAIL(a->mu.p, a->alloc_act((long)B * Lc, lat)); a->mu.ld = lat; a->mu.C = lat;
  ALLOC_OR_FAIL(a->lv.p, a->alloc_act((long)B * Lc, lat)); a->lv.ld = lat; a->lv.C = lat;
  EEG_TRY(op_conv_fwd(ctx, dt, a->h_enc.p, a->h_enc.ld, a->W(a->q_mu.w), a->P(a->q_mu.b), a->mu.p, lat, B, Lc, lat, lat, 1, 1, 0, 0, nullptr, 0, nullptr, 0));
  EEG_TRY(op_conv_fwd(ctx, dt, a->h_enc.p, a->h_enc.ld, a->W(a->q_lv.w), a->P(a->q_lv.b), a->lv.p, lat, B, Lc, lat, lat, 1, 1, 0, 0, nullptr, 0, nullptr, 0));
  a->eps_nlc = nullptr;
  if (eps) {
    ALLOC_OR_FAIL(a->eps_nlc, a->arena.alloc(sizeof(float) * n));
    EEG_TRY(eegldm_ncl_to_nlc(ctx, eps, a->eps_nlc, lat, B, lat, Lc, EEGLDM_F32));
  }
  ALLOC_OR_FAIL(a->sigma, a->arena.alloc(sizeof(float) * n));
  View z; ALLOC_OR_FAIL(z.p, a->alloc_act((long)B * Lc, lat)); z.ld = lat; z.C = lat;
  if (kl && !ctx->loss_prezeroed) HIP_TRY(hipMemsetAsync(kl, 0, sizeof(float), ctx->stream));
  EEG_TRY(ls_reparam(ctx, a->mu.p, a->lv.p, a->eps_nlc, z.p, a->sigma, kl, n, B, dt));
  *z_out = z;
  return 0;
}
// ---- frozen encode with fused pre-activation convs (enc_fused.hip).  No tape: only for callers that never back-propagate
// (eegldm_aekl_encode = Stage1Wrapper / encode_stage_2_inputs under no_grad, train_ldm.py:145-148, training.py:417-421).
bool enc_fused_eligible(const eegldm_aekl* a, int L) {
  if (getenv("EEGLDM_AEKL_NO_FUSED_ENC") != nullptr) return false;      // (read per call: tests toggle it inside one process)
  if ((a->dtype != EEGLDM_BF16 && a->dtype != EEGLDM_F16) || a->cfg.norm_num_groups != 1) return false;
  int Lc = L; bool any = false;
  for (const Op& o : a->enc) {
    if (o.kind == OP_CONV) Lc = (Lc + o.pl + o.pr - o.k) / o.stride + 1;
    else if (o.kind == OP_RES) { if (!pre_conv3_ok(a->dtype, o.r.cin, o.r.cout, Lc) || !pre_conv3_ok(a->dtype, o.r.cout, o.r.cout, Lc)) return false; any = true; }
    else if (o.kind != OP_GN) return false;
  }
  return any;
}
int aekl_encode_fused_seq(eegldm_aekl* a, View x, int B, int& L, View* out) {
  eegldm_ctx* ctx = a->ctx; const int dt = a->dtype;
  // one statistics slot (sum, sum of squares per sample) per tensor that a GroupNorm reads: zeroed together, filled by the producers
  size_t nslots = 0;
  for (const Op& o : a->enc) nslots += (o.kind == OP_RES) ? 2 : 0;
  double* slots; ALLOC_OR_FAIL(slots, (double*)a->arena.alloc(sizeof(double) * 2 * B * (nslots + 1)));
  HIP_TRY(hipMemsetAsync(slots, 0, sizeof(double) * 2 * B * (nslots + 1), ctx->stream));
  size_t used = 0;
  auto next_slot = [&]() { return slots + (used++) * 2 * (size_t)B; };
  double* cur_stats = nullptr;              // statistics of `x` (null: not computed yet)
  for (size_t i = 0; i < a->enc.size(); i++) {
    const Op& o = a->enc[i];
    const bool next_res = i + 1 < a->enc.size() && a->enc[i + 1].kind == OP_RES;
    if (o.kind == OP_CONV) {
      const int Lo = (L + o.pl + o.pr - o.k) / o.stride + 1;
      View y; ALLOC_OR_FAIL(y.p, a->alloc_act((long)B * Lo, o.cout)); y.ld = o.cout; y.C = o.cout;
      EEG_TRY(op_conv_fwd(ctx, dt, x.p, x.ld, a->W(o.w), o.b >= 0 ? a->P(o.b) : nullptr, y.p, y.ld, B, L, o.cin, o.cout, o.k, o.stride, o.pl, o.pr,
                          nullptr, 0, nullptr, 0));
      L = Lo; x = y; cur_stats = nullptr;
      if (next_res) { cur_stats = next_slot(); EEG_TRY(sample_stats_launch(ctx, x.p, (long)L * x.C, B, cur_stats, a->dtype)); }
    } else if (o.kind == OP_RES) {
      const ResDesc& r = o.r;
      EEG_CHECK(cur_stats, "fused encoder: no statistics for a ResBlock input");
      View h1; ALLOC_OR_FAIL(h1.p, a->alloc_act((long)B * L, r.cout)); h1.ld = r.cout; h1.C = r.cout;
      double* st_h1 = next_slot();
      EEG_TRY(pre_conv3_launch(ctx, x.p, cur_stats, a->P(r.gn1_w), a->P(r.gn1_b), a->W(r.c1_w), a->P(r.c1_b), nullptr, h1.p, st_h1, B, L, r.cin, r.cout, GN_EPS, a->dtype));
      View res = x;
      if (r.sk_w >= 0) {
        ALLOC_OR_FAIL(res.p, a->alloc_act((long)B * L, r.cout)); res.ld = r.cout; res.C = r.cout;
        EEG_TRY(op_conv_fwd(ctx, dt, x.p, x.ld, a->W(r.sk_w), a->P(r.sk_b), res.p, res.ld, B, L, r.cin, r.cout, 1, 1, 0, 0, nullptr, 0, nullptr, 0));
      }
      View y; ALLOC_OR_FAIL(y.p, a->alloc_act((long)B * L, r.cout)); y.ld = r.cout; y.C = r.cout;
      double* st_y = next_res ? next_slot() : nullptr;          // (a ResBlock followed by a conv / the final norm: nobody reads them)
      EEG_TRY(pre_conv3_launch(ctx, h1.p, st_h1, a->P(r.gn2_w), a->P(r.gn2_b), a->W(r.c2_w), a->P(r.c2_b), res.p, y.p, st_y, B, L, r.cout, r.cout, GN_EPS, a->dtype));
      x = y; cur_stats = st_y;
    } else {  // OP_GN (the final norm, no activation): the flat one-launch kernel
      View y; ALLOC_OR_FAIL(y.p, a->alloc_act((long)B * L, x.C)); y.ld = x.C; y.C = x.C;
      float* st; ALLOC_OR_FAIL(st, (float*)a->arena.alloc(sizeof(float) * 2 * B * o.groups));
      EEG_TRY(eegldm_groupnorm_fwd(ctx, x.p, x.ld, a->P(o.gw), a->P(o.gb), y.p, y.ld, st, B, L, x.C, o.groups, GN_EPS, 0, 0, nullptr, 0, dt));
      x = y; cur_stats = nullptr;
    }
  }
  *out = x;
  return 0;
}

int aekl_export_latents(eegldm_aekl* a, const View& z, float* z_ncl, float* mu_ncl, float* sigma_ncl) {
  const int lat = a->cfg.latent_channels;
  if (z_ncl) EEG_TRY(eegldm_nlc_to_ncl(a->ctx, z.p, lat, z_ncl, a->B, lat, a->Ll, a->dtype));
  if (mu_ncl) EEG_TRY(eegldm_nlc_to_ncl(a->ctx, a->mu.p, lat, mu_ncl, a->B, lat, a->Ll, a->dtype));
  if (sigma_ncl) EEG_TRY(eegldm_nlc_to_ncl(a->ctx, a->sigma, lat, sigma_ncl, a->B, lat, a->Ll, EEGLDM_F32));
  return 0;
}
int aekl_decode_impl(eegldm_aekl* a, const View& z, int B, int Ll, float* recon) {
  int Lc = Ll; View y;
  EEG_TRY(a->forward_seq(a->dec, z, B, Lc, &y, a->tape_dec, 1));
  return eegldm_nlc_to_ncl(a->ctx, y.p, y.ld, recon, B, a->cfg.out_channels, Lc, a->dtype);
}
}  // namespace

// encode_stage_2_inputs / Stage1Wrapper (training.py:15-26): z = mu + eps*sigma (eps NULL -> z = mu)
extern "C" int eegldm_aekl_encode(eegldm_aekl* a, const float* x, const float* eps, float* z, float* z_mu, float* z_sigma, int B, int L) {
  EEG_CHECK(a && x && a->params, "null argument / unbound parameters");
  View zv;
  EEG_TRY(aekl_encode_impl(a, x, eps, B, L, &zv, nullptr, true));
  return aekl_export_latents(a, zv, z, z_mu, z_sigma);
}
// decode_stage_2_outputs (sample_trials.py:166): z (B, lat, Ll) -> (B, out, Ll * 2^(levels-1))
extern "C" int eegldm_aekl_decode(eegldm_aekl* a, const float* z, float* recon, int B, int Ll) {
  EEG_CHECK(a && z && recon && a->params, "null argument / unbound parameters");
  const int lat = a->cfg.latent_channels;
  a->arena.reset(); a->rt.clear(); a->tape_enc.clear(); a->tape_dec.clear(); a->have_tape = false;
  View zv; ALLOC_OR_FAIL(zv.p, a->alloc_act((long)B * Ll, lat)); zv.ld = lat; zv.C = lat;
  EEG_TRY(eegldm_ncl_to_nlc(a->ctx, z, zv.p, lat, B, lat, Ll, a->dtype));
  return aekl_decode_impl(a, zv, B, Ll, recon);
}
// forward(x) -> (reconstruction, z_mu, z_sigma) with eps supplied; kl (nullable device scalar) = KL term
extern "C" int eegldm_aekl_forward(eegldm_aekl* a, const float* x, const float* eps, float* recon, float* z_mu, float* z_sigma, float* kl, int B, int L) {
  EEG_CHECK(a && x && recon && a->params, "null argument / unbound parameters");
  a->thin_tape = false;
  if (thin_ready(a, L)) {
    // whole-network path (aekl_thin.hip): one workgroup per window, two launches per forward + backward instead of ~300
    ThinProgram& p = a->thin;
    a->arena.reset(); a->rt.clear(); a->tape_enc.clear(); a->tape_dec.clear(); a->have_tape = false;
    a->B = B; a->L = L; a->Ll = p.Ll;
    ALLOC_OR_FAIL(p.tape, a->arena.alloc(sizeof(float) * (size_t)B * p.tape_stride));
    ALLOC_OR_FAIL(p.stats, a->arena.alloc(sizeof(float) * (size_t)B * p.nstat * 2));
    a->thin_eps = nullptr;
    if (eps) {      // kept for the backward pass (the caller's buffer may be gone by then)
      ALLOC_OR_FAIL(a->thin_eps, a->arena.alloc(sizeof(float) * (size_t)B * p.lat * p.Ll));
      HIP_TRY(hipMemcpyAsync(a->thin_eps, eps, sizeof(float) * (size_t)B * p.lat * p.Ll, hipMemcpyDeviceToDevice, a->ctx->stream));
    }
    if (kl && !a->ctx->loss_prezeroed) HIP_TRY(hipMemsetAsync(kl, 0, sizeof(float), a->ctx->stream));
    EEG_TRY(thin_forward(a->ctx, p, a->params, x, a->thin_eps, recon, z_mu, z_sigma, kl, B));
    a->have_tape = true; a->thin_tape = true;
    return 0;
  }
  View zv;
  EEG_TRY(aekl_encode_impl(a, x, eps, B, L, &zv, kl));
  EEG_TRY(aekl_export_latents(a, zv, nullptr, z_mu, z_sigma));
  EEG_TRY(aekl_decode_impl(a, zv, B, a->Ll, recon));
  a->have_tape = true;
  return 0;
}
// grads += d/dparams [ <d_recon, recon> + kl_weight * KL ]; dx nullable
extern "C" int eegldm_aekl_backward(eegldm_aekl* a, const float* d_recon, float kl_weight, float* dx) {
  return eegldm_aekl_backward_ex(a, d_recon, nullptr, nullptr, kl_weight, dx);
}
// ... + <d_mu, z_mu> + <d_sigma, z_sigma>: gradients a caller's own loss put on the z_mu / z_sigma tensors eegldm_aekl_forward returned
// (fp32 (B, lat, L / 2^(levels-1)), nullable) -- what torch.autograd hands back when the KL term is written with tensor ops on the
// outputs, as train_autoencoderkl.py:210-211 does (eegldm.autograd)
extern "C" int eegldm_aekl_backward_ex(eegldm_aekl* a, const float* d_recon, const float* d_mu, const float* d_sigma, float kl_weight, float* dx) {
  EEG_CHECK(a && d_recon, "null argument");
  EEG_CHECK(a->have_tape, "call eegldm_aekl_forward first");
  EEG_CHECK(a->grads, "no gradient buffer bound");
  a->have_tape = false;
  if (a->thin_tape) {
    a->thin_tape = false;
    return thin_backward(a->ctx, a->thin, a->params, a->grads, d_recon, a->thin_eps, kl_weight / (float)a->B, dx, a->B, d_mu, d_sigma);
  }
  eegldm_ctx* ctx = a->ctx; const int dt = a->dtype, lat = a->cfg.latent_channels, B = a->B, L = a->L, Ll = a->Ll, co = a->cfg.out_channels;
  View dy; ALLOC_OR_FAIL(dy.p, a->alloc_act((long)B * L, co)); dy.ld = co; dy.C = co;
  EEG_TRY(eegldm_ncl_to_nlc(ctx, d_recon, dy.p, co, B, co, L, dt));
  View dz;
  EEG_TRY(a->backward_seq(a->dec, a->tape_dec, B, dy, &dz, true));
  const long n = (long)B * Ll * lat;
  View dmu, dlv; ALLOC_OR_FAIL(dmu.p, a->alloc_act((long)B * Ll, lat)); ALLOC_OR_FAIL(dlv.p, a->alloc_act((long)B * Ll, lat));
  dmu.ld = dlv.ld = lat;
  EEG_TRY(ls_reparam_bwd(ctx, a->mu.p, a->lv.p, a->eps_nlc, a->sigma, dz.p, dmu.p, dlv.p, n, kl_weight / (float)B, dt, d_mu, d_sigma, lat, Ll));
  const int ce = a->h_enc.C;
  EEG_TRY(op_conv_wgrad(ctx, dt, a->h_enc.p, a->h_enc.ld, dmu.p, lat, a->G(a->q_mu.w), a->G(a->q_mu.b), B, Ll, lat, lat, 1, 1, 0, 0));
  EEG_TRY(op_conv_wgrad(ctx, dt, a->h_enc.p, a->h_enc.ld, dlv.p, lat, a->G(a->q_lv.w), a->G(a->q_lv.b), B, Ll, lat, lat, 1, 1, 0, 0));
  View dh; ALLOC_OR_FAIL(dh.p, a->alloc_act((long)B * Ll, ce)); dh.ld = ce; dh.C = ce;
  EEG_TRY(op_conv_dgrad(ctx, dt, dmu.p, lat, a->W(a->q_mu.w), dh.p, ce, B, Ll, lat, lat, 1, 1, 0, 0, nullptr, 0));
  EEG_TRY(op_conv_dgrad(ctx, dt, dlv.p, lat, a->W(a->q_lv.w), dh.p, ce, B, Ll, lat, lat, 1, 1, 0, 0, dh.p, ce));
  View dx0;
  EEG_TRY(a->backward_seq(a->enc, a->tape_enc, B, dh, &dx0, dx != nullptr));
  if (dx) EEG_TRY(eegldm_nlc_to_ncl(ctx, dx0.p, dx0.ld, dx, B, a->cfg.in_channels, L, dt));
  return 0;
}

// ================================================================== PatchDiscriminator
struct eegldm_disc : SeqNet {
  eegldm_disc_cfg cfg;
  std::vector<Op> ops;
  std::vector<OpTape> tape;
  std::vector<Entry> buf_entries; long nbuffers = 0;
  int B = 0, L = 0, Lo = 0; bool have_tape = false;
  // Fused tail (disc_tail.hip): the last hidden layer's BatchNorm + LeakyReLU and the one-channel final conv run as one pass over the last
  // hidden conv's output y (forward) / two passes (backward); `ops` minus those two = `head`, whose tape is `tape`; the tail keeps (y, statistics).
  std::vector<Op> head;
  bool tail_on = false; View tail_y; float* tail_st = nullptr;
  size_t tape_ops() const { return ops.size() - (tail_on ? 2 : 0); }
};
namespace {
bool disc_tail_eligible(const eegldm_disc* d) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_DISC_NO_FUSED_TAIL") != nullptr);
  const size_t n = d->ops.size();
  if (off || eeg_deterministic() || n < 4) return false;      // (deterministic mode: the tail's per-channel sums meet in fp64 LDS atomics, its parameter gradients in fp32 atomics)
  const Op& act = d->ops[n - 2]; const Op& fin = d->ops[n - 1]; const Op& hc = d->ops[n - 3];
  return hc.kind == OP_CONV && act.kind == OP_ACT && act.bn_w >= 0 && fin.kind == OP_CONV && fin.cout == 1 && fin.k == 3 && fin.stride == 1 &&
         fin.pl == 1 && fin.pr == 1 && fin.cin == hc.cout && disc_tail_ok(d->dtype, fin.cin, fin.cin);
}
// the first layer: conv (one input channel, k 3, padding 1) + plain LeakyReLU, both on the tape
bool disc_head_eligible(const eegldm_disc* d, const std::vector<Op>& ops) {
  EEG_ENV_VAR(bool, off, getenv("EEGLDM_DISC_NO_FUSED_HEAD") != nullptr);
  if (off || eeg_deterministic() || ops.size() < 3 || d->tape.size() < 2) return false;
  const Op& c0 = ops[0]; const Op& a0 = ops[1];
  if (c0.kind != OP_CONV || c0.cin != 1 || c0.k != 3 || c0.pl != 1 || c0.pr != 1 || a0.kind != OP_ACT || a0.bn_w >= 0) return false;
  const OpTape& t0 = d->tape[0];
  return t0.x.ld == 1 && disc_head_ok(d->dtype, c0.cout, c0.cout, c0.stride, t0.Lin, t0.Lout);
}
}  // namespace

extern "C" int eegldm_disc_create(eegldm_ctx* ctx, const eegldm_disc_cfg* cfg, eegldm_disc** out) {
  EEG_CHECK(ctx && cfg && out, "null argument");
  EEG_CHECK(cfg->kernel_size == 3 && cfg->padding == 1, "only kernel_size=3, padding=1 (config_aekl_eeg.yaml:36-40) is implemented");
  EEG_CHECK(cfg->num_layers_d >= 1 && cfg->num_layers_d <= 6, "bad num_layers_d");
  eegldm_disc* d = new eegldm_disc();
  d->ctx = ctx; d->cfg = *cfg; d->dtype = cfg->dtype; d->arena.min_block = (size_t)256 << 20;
  Layout lay, blay;
  d->ops.push_back(make_conv(d, lay, "initial_conv.conv", cfg->in_channels, cfg->num_channels, 3, 2, 1, 1, true));
  { Op a; a.kind = OP_ACT; a.slope = 0.2f; d->ops.push_back(a); }
  int ic = cfg->num_channels, oc = ic * 2;
  for (int l = 0; l < cfg->num_layers_d; l++) {
    const int stride = (l == cfg->num_layers_d - 1) ? 1 : 2;
    const std::string p = std::to_string(l);
    d->ops.push_back(make_conv(d, lay, p + ".conv", ic, oc, 3, stride, 1, 1, cfg->bias != 0));
    Op a; a.kind = OP_ACT; a.slope = 0.2f;
    a.bn_w = lay.take(oc); a.bn_b = lay.take(oc);
    d->add_entry(p + ".adn.N.weight", a.bn_w, 1, oc); d->add_entry(p + ".adn.N.bias", a.bn_b, 1, oc);
    a.rm = blay.take(oc); a.rv = blay.take(oc); a.nbt = blay.take(1);
    Entry e; e.ndim = 1; e.shape[0] = oc; e.shape[1] = e.shape[2] = 0; e.numel = oc;
    e.name = p + ".adn.N.running_mean"; e.offset = a.rm; d->buf_entries.push_back(e);
    e.name = p + ".adn.N.running_var"; e.offset = a.rv; d->buf_entries.push_back(e);
    e.name = p + ".adn.N.num_batches_tracked"; e.offset = a.nbt; e.ndim = 0; e.numel = 1; e.shape[0] = 0; d->buf_entries.push_back(e);
    d->ops.push_back(a);
    ic = oc; oc *= 2;
  }
  d->ops.push_back(make_conv(d, lay, "final_conv.conv", ic, cfg->out_channels, 3, 1, 1, 1, true));
  d->nparams = lay.off; d->nbuffers = blay.off;
  d->head.assign(d->ops.begin(), d->ops.end() - 2);
  *out = d;
  return 0;
}
extern "C" int eegldm_disc_destroy(eegldm_disc* d) { delete d; return 0; }
extern "C" int eegldm_disc_num_entries(const eegldm_disc* d) { return (int)d->entries.size(); }
extern "C" long eegldm_disc_num_params(const eegldm_disc* d) { return d->nparams; }
extern "C" int eegldm_disc_entry(const eegldm_disc* d, int i, char* name, int cap, long* offset, long* numel, int* ndim, int shape[3]) {
  return entry_query(d, i, name, cap, offset, numel, ndim, shape);
}
extern "C" int eegldm_disc_num_buffer_entries(const eegldm_disc* d) { return (int)d->buf_entries.size(); }
extern "C" long eegldm_disc_num_buffers(const eegldm_disc* d) { return d->nbuffers; }
extern "C" int eegldm_disc_buffer_entry(const eegldm_disc* d, int i, char* name, int cap, long* offset, long* numel, int* ndim, int shape[3]) {
  EEG_CHECK(d && i >= 0 && i < (int)d->buf_entries.size(), "buffer entry %d out of range", i);
  NetBase tmp; tmp.entries.push_back(d->buf_entries[i]);
  return entry_query(&tmp, 0, name, cap, offset, numel, ndim, shape);
}
// buffers: flat fp32 [num_buffers] holding running_mean / running_var / num_batches_tracked (as a float count)
extern "C" int eegldm_disc_bind(eegldm_disc* d, float* params, float* grads, float* buffers) {
  EEG_CHECK(d && params && buffers, "null argument");
  d->buffers = buffers;
  return d->bind(params, grads);
}
extern "C" int eegldm_disc_sync_weights(eegldm_disc* d) { EEG_CHECK(d, "null argument"); return d->sync_weights(); }

// forward(x)[-1]: x (B, in, L) -> logits (B, out, L/2^(num_layers_d)); training: batch statistics + running-stat update
extern "C" int eegldm_disc_forward(eegldm_disc* d, const float* x, float* logits, int B, int L, int training) {
  EEG_CHECK(d && x && logits && d->params, "null argument / unbound parameters");
  const int cin = d->cfg.in_channels;
  d->arena.reset(); d->tape.clear(); d->rt.clear();
  View x0; ALLOC_OR_FAIL(x0.p, d->alloc_act((long)B * L, cin)); x0.ld = cin; x0.C = cin;
  EEG_TRY(eegldm_ncl_to_nlc(d->ctx, x, x0.p, cin, B, cin, L, d->dtype));
  int Lc = L; View y;
  d->tail_on = disc_tail_eligible(d);
  if (d->tail_on) {
    const Op& act = d->ops[d->ops.size() - 2]; const Op& fin = d->ops.back();
    EEG_TRY(d->forward_seq(d->head, x0, B, Lc, &y, d->tape, training));
    const int C = y.C;
    ALLOC_OR_FAIL(d->tail_st, (float*)d->arena.alloc(sizeof(float) * 2 * C));
    const bool upd = d->buffers && training != 2;      // as SeqNet::forward_seq: training == 2 leaves the running statistics alone
    EEG_TRY(ls_bn_stats(d->ctx, y.p, y.ld, d->tail_st, upd ? d->buffers + act.rm : nullptr, upd ? d->buffers + act.rv : nullptr,
                        upd ? d->buffers + act.nbt : nullptr, (long)B * Lc, C, training, d->dtype));
    d->tail_y = y;
    d->B = B; d->L = L; d->Lo = Lc; d->have_tape = true;      // (the final conv keeps the length: k = 3, stride 1, padding 1)
    return disc_tail_fwd(d->ctx, d->dtype, y.p, y.ld, d->P(act.bn_w), d->P(act.bn_b), d->tail_st, d->P(fin.w), fin.b >= 0 ? d->P(fin.b) : nullptr,
                         act.slope, logits, B, Lc, C);
  }
  EEG_TRY(d->forward_seq(d->ops, x0, B, Lc, &y, d->tape, training));
  d->B = B; d->L = L; d->Lo = Lc; d->have_tape = true;
  return eegldm_nlc_to_ncl(d->ctx, y.p, y.ld, logits, B, d->cfg.out_channels, Lc, d->dtype);
}
// The reference's PatchDiscriminator.forward returns the LIST of feature maps, one per block (initial conv + LeakyReLU, every
// conv + BatchNorm + LeakyReLU layer, final conv); the trainer indexes [-1] (train_autoencoderkl.py:213).  The maps before the last
// are the activation ops' outputs, which the tape of the most recent forward still holds: feature `index` (0 .. num_layers_d) is
// copied out as fp32 (B, C, L); out == NULL only reports the shape.  The last entry of the list is the logits eegldm_disc_forward wrote.
extern "C" int eegldm_disc_feature(eegldm_disc* d, int index, float* out, int* C, int* L) {
  EEG_CHECK(d && d->have_tape && d->tape.size() == d->tape_ops(), "call eegldm_disc_forward first");
  int seen = -1;
  for (size_t i = 0; i + 1 < d->ops.size(); i++) {
    if (d->ops[i].kind != OP_ACT) continue;
    if (++seen != index) continue;
    if (d->tail_on && i == d->ops.size() - 2) {      // the fused tail never stores this activation: apply the layer's BatchNorm + LeakyReLU to its input now
      const Op& act = d->ops[i]; const int Cc = d->tail_y.C;
      if (C) *C = Cc;
      if (L) *L = d->Lo;
      if (out) {
        void* a; ALLOC_OR_FAIL(a, d->alloc_act((long)d->B * d->Lo, Cc));
        EEG_TRY(ls_bn_apply(d->ctx, d->tail_y.p, d->tail_y.ld, d->P(act.bn_w), d->P(act.bn_b), d->tail_st, a, Cc, (long)d->B * d->Lo, Cc, act.slope, d->dtype));
        EEG_TRY(eegldm_nlc_to_ncl(d->ctx, a, Cc, out, d->B, Cc, d->Lo, d->dtype));
      }
      return 0;
    }
    const OpTape& nx = d->tape[i + 1];                    // the next op's input IS this activation's output
    if (C) *C = nx.x.C;
    if (L) *L = nx.Lin;
    if (out) EEG_TRY(eegldm_nlc_to_ncl(d->ctx, nx.x.p, nx.x.ld, out, d->B, nx.x.C, nx.Lin, d->dtype));
    return 0;
  }
  EEG_FAIL(EEGLDM_ERR_INVALID, "feature index out of range");
}
// param_grads != 0: grads += d/dparams; dx (nullable) = d/dx
static int disc_backward_impl(eegldm_disc* d, const float* dlogits, float* dx, int param_grads, bool keep_tape);
extern "C" int eegldm_disc_backward(eegldm_disc* d, const float* dlogits, float* dx, int param_grads) { return disc_backward_impl(d, dlogits, dx, param_grads, false); }
static int disc_backward_impl(eegldm_disc* d, const float* dlogits, float* dx, int param_grads, bool keep_tape) {
  EEG_CHECK(d && dlogits, "null argument");
  EEG_CHECK(d->have_tape, "call eegldm_disc_forward first");
  EEG_CHECK(!param_grads || d->grads, "no gradient buffer bound");
  d->have_tape = keep_tape;
  const int co = d->cfg.out_channels, B = d->B;
  EEG_CHECK(!keep_tape || d->rt.empty(), "tape reuse is for ResBlock-free stacks");
  View dx0; int rc, first = 0;
  if (d->tail_on) {
    // gradient of the last hidden conv's output straight from the fp32 logit gradients (the final conv's data gradient is recomputed inside)
    const Op& act = d->ops[d->ops.size() - 2]; const Op& fin = d->ops.back(); const int C = d->tail_y.C; const bool pg = param_grads != 0;
    View dy; ALLOC_OR_FAIL(dy.p, d->alloc_act((long)B * d->Lo, C)); dy.ld = C; dy.C = C;
    EEG_TRY(disc_tail_bwd(d->ctx, d->dtype, d->tail_y.p, d->tail_y.ld, d->P(act.bn_w), d->P(act.bn_b), d->tail_st, d->P(fin.w), act.slope, dlogits,
                          dy.p, C, pg ? d->G(act.bn_w) : nullptr, pg ? d->G(act.bn_b) : nullptr, pg ? d->G(fin.w) : nullptr,
                          pg && fin.b >= 0 ? d->G(fin.b) : nullptr, B, d->Lo, C));
    d->param_grads = pg;
    first = disc_head_eligible(d, d->head) ? 2 : 0;
    rc = d->backward_seq(d->head, d->tape, B, dy, &dx0, dx != nullptr || first > 0, keep_tape, first);
  } else {
    View dy; ALLOC_OR_FAIL(dy.p, d->alloc_act((long)B * d->Lo, co)); dy.ld = co; dy.C = co;
    EEG_TRY(eegldm_ncl_to_nlc(d->ctx, dlogits, dy.p, co, B, co, d->Lo, d->dtype));
    d->param_grads = param_grads != 0;
    first = disc_head_eligible(d, d->ops) ? 2 : 0;
    rc = d->backward_seq(d->ops, d->tape, B, dy, &dx0, dx != nullptr || first > 0, keep_tape, first);
  }
  d->param_grads = true;
  EEG_TRY(rc);
  if (first > 0) {
    // first layer fused (disc_tail.hip): dx0 is the gradient of the ACTIVATED first-layer output; mask, weight / bias gradients and the input
    // gradient come from it and the window itself in one pass each; dx leaves as fp32 NCL directly
    const Op& c0 = d->ops[0]; const Op& a0 = d->ops[1]; const OpTape t0 = d->tape[0]; const bool pg = param_grads != 0;
    if (!keep_tape) d->tape.clear();
    return disc_head_bwd(d->ctx, d->dtype, dx0.p, dx0.ld, t0.x.p, d->W(c0.w), c0.b >= 0 ? d->P(c0.b) : nullptr, a0.slope, pg ? d->G(c0.w) : nullptr,
                         pg && c0.b >= 0 ? d->G(c0.b) : nullptr, dx, B, t0.Lin, t0.Lout, c0.cout, c0.stride);
  }
  if (dx) EEG_TRY(eegldm_nlc_to_ncl(d->ctx, dx0.p, dx0.ld, dx, B, d->cfg.in_channels, d->L, d->dtype));
  return 0;
}

// ================================================================== fused train step (train_autoencoderkl.py:200-234)
extern "C" int eegldm_l1_loss(eegldm_ctx*, const float*, const float*, float*, float*, long, float);
extern "C" int eegldm_lsgan_loss(eegldm_ctx*, const float*, int, float*, float*, long, float);
extern "C" int eegldm_axpy(eegldm_ctx*, float*, const float*, float, long);

extern "C" int eegldm_aekl_train_step(eegldm_aekl* a, eegldm_disc* d, const float* x, const float* eps, float adv_weight, float kl_weight,
                                      float spectral_weight, int use_spectral, float* losses, float* recon_out, int B, int L) {
  EEG_CHECK(a && d && x && eps && losses, "null argument");
  EEG_CHECK(a->cfg.in_channels == a->cfg.out_channels && d->cfg.in_channels == a->cfg.out_channels, "channel mismatch between autoencoder and discriminator");
  eegldm_ctx* ctx = a->ctx;
  const int C = a->cfg.in_channels;
  const long n = (long)B * C * L;
  int Ld = L; for (int i = 0; i < d->cfg.num_layers_d; i++) Ld = (Ld + 2 - 3) / 2 + 1;   // initial + (num_layers_d-1) stride-2 convs
  const long nl = (long)B * d->cfg.out_channels * Ld;
  a->stage.reset();
  float* recon; ALLOC_OR_FAIL(recon, a->stage.alloc(sizeof(float) * n));
  float* drecon; ALLOC_OR_FAIL(drecon, a->stage.alloc(sizeof(float) * n));
  float* dxd; ALLOC_OR_FAIL(dxd, a->stage.alloc(sizeof(float) * n));
  float* logits; ALLOC_OR_FAIL(logits, a->stage.alloc(sizeof(float) * nl));
  float* dlogits; ALLOC_OR_FAIL(dlogits, a->stage.alloc(sizeof(float) * nl));
  // losses[0..5] = recons (L1), spectral, kl, generator adv, D fake, D real
  // ---- generator
  // one memset for the six loss scalars (the loss entry points skip theirs while loss_prezeroed is set); the L1 term WRITES drecon
  struct Prezero { eegldm_ctx* c; explicit Prezero(eegldm_ctx* cc) : c(cc) { c->loss_prezeroed = true; } ~Prezero() { c->loss_prezeroed = false; c->l1_overwrite = false; } } prezero(ctx);
  HIP_TRY(hipMemsetAsync(losses, 0, sizeof(float) * 6, ctx->stream));
  EEG_TRY(eegldm_aekl_forward(a, x, eps, recon, nullptr, nullptr, losses + 2, B, L));
  ctx->l1_overwrite = true;
  EEG_TRY(eegldm_l1_loss(ctx, recon, x, losses + 0, drecon, n, 1.0f));
  ctx->l1_overwrite = false;
  // (Round 6 tried this one latency-bound launch -- one workgroup per window, three 3072-point DFTs in LDS, ~100 us at B = 256 -- on the auxiliary
  // stream beside the discriminator's forward / backward on the reconstruction: 3.20 against 3.13 ms serial on one box, like round 5's attempt
  // with the thin autoencoder's backward.  Two streams are two hardware queues, and alternating between them costs more than the overlap
  // returns.  Removed; HISTORY.md keeps the numbers.)
  EEG_TRY(eegldm_spectral_loss(ctx, recon, x, losses + 1, use_spectral ? drecon : nullptr, B, C, L, spectral_weight));
  // (this forward serves the generator loss AND the fake-sample loss below: its BatchNorm layers make both running-statistics updates at once)
  EEG_ENV_VAR(bool, refwd, getenv("EEGLDM_AEKL_REFORWARD_D") != nullptr);       // developer switch: the literal three-forward sequence
  struct Repeats { eegldm_ctx* c; Repeats(eegldm_ctx* cc, int n) : c(cc) { c->bn_running_repeats = n; } ~Repeats() { c->bn_running_repeats = 1; } };
  { Repeats two(ctx, refwd ? 1 : 2); EEG_TRY(eegldm_disc_forward(d, recon, logits, B, L, 1)); }
  EEG_TRY(eegldm_lsgan_loss(ctx, logits, 1, losses + 3, dlogits, nl, adv_weight));
  // The reference forwards the discriminator on the reconstruction twice: for the generator loss (:213) and, detached, for the
  // fake-sample loss (:225).  Between the two only the GENERATOR's parameters change, so activations and logits are identical:
  // the second forward is replaced by a second backward over the kept tape (other dlogits, this time into D's gradients); the
  // BatchNorm running-statistics update that a second forward would have made (same batch statistics) was made by the first (above).
  EEG_TRY(disc_backward_impl(d, dlogits, dxd, 0, !refwd));
  EEG_TRY(eegldm_axpy(ctx, drecon, dxd, 1.0f, n));
  // (Round 5 also tried the thin autoencoder's one-launch backward on the auxiliary stream beside the discriminator's launches: 3.86 against
  // 3.77 ms -- its 256 workgroups hold ~100 KB of LDS on every CU and the GEMMs cannot become resident beside them.  Removed in round 6;
  // HISTORY.md keeps the numbers.)
  EEG_TRY(eegldm_aekl_backward(a, drecon, kl_weight, nullptr));
  // ---- discriminator: 0.5 * adv_weight * (fake->0 + real->1)
  if (refwd) EEG_TRY(eegldm_disc_forward(d, recon, logits, B, L, 1));
  EEG_TRY(eegldm_lsgan_loss(ctx, logits, 0, losses + 4, dlogits, nl, 0.5f * adv_weight));
  EEG_TRY(eegldm_disc_backward(d, dlogits, nullptr, 1));
  EEG_TRY(eegldm_disc_forward(d, x, logits, B, L, 1));
  EEG_TRY(eegldm_lsgan_loss(ctx, logits, 1, losses + 5, dlogits, nl, 0.5f * adv_weight));
  EEG_TRY(eegldm_disc_backward(d, dlogits, nullptr, 1));
  if (recon_out) HIP_TRY(hipMemcpyAsync(recon_out, recon, sizeof(float) * n, hipMemcpyDeviceToDevice, ctx->stream));
  return 0;
}
