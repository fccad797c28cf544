// UNet denoiser executor: forward, hand-written backward and the LDM train-step body.
// Mirrors UNetModel of /root/reference/src/models/unet.py:330-563 (constructor loops
// :382-499, forward :512-563), dropout=0.  The configuration of config/config_ldm.yaml:30-43 (resblock_updown=True,
// use_scale_shift_norm=False, num_heads=1) is the one the fused fast paths are built for; the other constructor branches -- several
// attention heads (num_heads / num_head_channels / num_heads_upsample), use_scale_shift_norm=True, resblock_updown=False with conv or
// pool / nearest resampling layers -- run through the same kernels at layer granularity (round 6; goldens tests/golden/unet_opt_*.npz).
//
// Host side only: this file sequences kernels from gemm.hip / norm.hip / elementwise.hip
// / direct_conv.hip on the context's stream.  All activations are NLC in the model dtype;
// channel concatenations of the up path (`th.cat([h, h_pop], 1)`, unet.py:553) are never
// materialised: the producer of each half writes straight into its column range of a
// pre-sized buffer (every kernel takes a leading dimension).
//
// Parameters live in ONE flat fp32 buffer owned by the caller (so Adam is one launch and
// DDP is one all-reduce over a contiguous gradient buffer); conv weights are stored packed
// [K][Cout][Cin]; the 21 timestep-embedding Linear layers of the ResBlocks are stored
// contiguously so `emb_layers` of every block is ONE GEMM per step.
#include <string.h>

#include <string>
#include <vector>

#include "net.h"

namespace {
constexpr int GN_G = 32;
struct Layer { int kind; ResDesc r; AttnDesc a; RsDesc s; };   // kind 0 = res, 1 = attn, 2 = Downsample / Upsample layer (s.up)
inline int layer_cin(const Layer& l) { return l.kind == 0 ? l.r.cin : (l.kind == 1 ? l.a.c : l.s.c); }
inline int layer_cout(const Layer& l) { return l.kind == 0 ? l.r.cout : (l.kind == 1 ? l.a.c : l.s.c); }
inline int layer_len(const Layer& l, int L) {
  if (l.kind == 0) return l.r.updown == 1 ? L / 2 : (l.r.updown == 2 ? L * 2 : L);
  if (l.kind == 2) return l.s.up ? L * 2 : L / 2;
  return L;
}
struct Block { std::vector<Layer> layers; int cin, cout; };
}  // namespace

struct eegldm_unet : NetBase {
  eegldm_unet_cfg cfg;
  int mc, te;
  long off_emb_w = 0, off_emb_b = 0, off_te0_w, off_te0_b, off_te2_w, off_te2_b, off_cin_w, off_cin_b, off_out_gw, off_out_gb, off_out_w, off_out_b;
  std::vector<Block> in_blocks, out_blocks; Block mid;
  long off_mid_begin = 0;        // flat offset of the first middle-block parameter: [off_mid_begin, nparams) is final once the middle block's backward is enqueued
  eegldm_grad_hook grad_hook = nullptr; void* grad_hook_user = nullptr;
  std::vector<int> skip_c1;      // per output block: channels of h entering the concat
  // ---- forward tape
  int B = 0, L = 0; bool have_tape = false;
  View x0, h_last, a_out; float* st_out = nullptr;
  float *e0 = nullptr, *a1e = nullptr, *semb = nullptr, *h1e = nullptr, *emb = nullptr;   // embedding MLP runs in fp32
  std::vector<View> in_out;      // outputs of the input blocks (views into concat buffers)
  std::vector<View> cat;         // concat buffers per output block
  std::vector<RsTape> st;        // tapes of the Downsample / Upsample layers
};

namespace {

// Builds the block plan exactly like the reference constructor and lays out the flat buffer.
int build_plan(eegldm_unet* u) {
  const eegldm_unet_cfg& c = u->cfg;
  const int mc = c.model_channels, te = 4 * mc;
  u->mc = mc; u->te = te;
  auto has_attn = [&](int ds) { for (int i = 0; i < c.n_attn; i++) if (c.attention_resolutions[i] == ds) return true; return false; };

  // ---- pass 1: structure
  struct Tmp { int kind, cin, cout, updown; int heads = 1; };      // kind 2: updown = 1 down / 2 up
  const bool rs_layers = c.resample_layers != 0, rs_conv = c.resample_pool_only == 0, ssn = c.use_scale_shift_norm != 0;
  const int nh_in = c.num_heads > 0 ? c.num_heads : 1, nh_up = c.num_heads_upsample > 0 ? c.num_heads_upsample : nh_in;      // unet.py:354-355
  auto heads_of = [&](int chn, int n) { return c.num_head_channels > 0 ? chn / c.num_head_channels : n; };                // unet.py:146-153
  std::vector<std::vector<Tmp>> inp, outp; std::vector<Tmp> mid;
  std::vector<int> chans{mc};
  inp.push_back({});   // block 0 = conv_in (handled separately)
  int ch = mc, ds = 1;
  for (int level = 0; level < c.n_mult; level++) {
    const int mult = c.channel_mult[level];
    for (int r = 0; r < c.num_res_blocks; r++) {
      std::vector<Tmp> l{{0, ch, mult * mc, 0}};
      ch = mult * mc;
      if (has_attn(ds)) l.push_back({1, ch, ch, 0, heads_of(ch, nh_in)});
      inp.push_back(l); chans.push_back(ch);
    }
    if (level != c.n_mult - 1) { inp.push_back({{rs_layers ? 2 : 0, ch, ch, 1}}); chans.push_back(ch); ds *= 2; }
  }
  mid = {{0, ch, ch, 0}, {1, ch, ch, 0, heads_of(ch, nh_in)}, {0, ch, ch, 0}};
  std::vector<int> chans_pop = chans;
  for (int level = c.n_mult - 1; level >= 0; level--) {
    const int mult = c.channel_mult[level];
    for (int i = 0; i <= c.num_res_blocks; i++) {
      const int ich = chans_pop.back(); chans_pop.pop_back();
      u->skip_c1.push_back(ch);
      std::vector<Tmp> l{{0, ch + ich, mc * mult, 0}};
      ch = mc * mult;
      if (has_attn(ds)) l.push_back({1, ch, ch, 0, heads_of(ch, nh_up)});
      if (level && i == c.num_res_blocks) { l.push_back({rs_layers ? 2 : 0, ch, ch, 2}); ds /= 2; }
      outp.push_back(l);
    }
  }
  EEG_CHECK(ch == mc, "plan error: final channels %d != model_channels %d", ch, mc);

  // ---- pass 2: flat layout.  Region A: all ResBlock emb Linear weights [etot][te] then biases [etot].
  int etot = 0;
  const int ew = ssn ? 2 : 1;       // use_scale_shift_norm: every ResBlock's emb_layers yields (scale, shift) = 2 cout values (unet.py:279-284)
  auto count_emb = [&](const std::vector<Tmp>& l) { for (auto& t : l) if (t.kind == 0) etot += ew * t.cout; };
  for (auto& l : inp) count_emb(l);
  count_emb(mid);
  for (auto& l : outp) count_emb(l);
  u->etot = etot;
  long off = 0;
  u->off_emb_w = off; off += (long)etot * te;
  u->off_emb_b = off; off += etot;
  auto take = [&](long n) { long o = off; off += (n + 7) / 8 * 8; return o; };   // keep every tensor 32-byte aligned

  u->off_te0_w = take((long)te * mc); u->off_te0_b = take(te);
  u->off_te2_w = take((long)te * te); u->off_te2_b = take(te);
  u->add_entry("time_embed.0.weight", u->off_te0_w, 2, te, mc); u->add_entry("time_embed.0.bias", u->off_te0_b, 1, te);
  u->add_entry("time_embed.2.weight", u->off_te2_w, 2, te, te); u->add_entry("time_embed.2.bias", u->off_te2_b, 1, te);
  u->off_cin_w = take((long)mc * c.in_channels * 3); u->off_cin_b = take(mc);
  u->add_entry("input_blocks.0.0.weight", u->off_cin_w, 3, mc, c.in_channels, 3); u->add_entry("input_blocks.0.0.bias", u->off_cin_b, 1, mc);

  int emb_col = 0;
  auto lay = [&](const std::string& prefix, const std::vector<Tmp>& l, Block& blk) {
    for (size_t j = 0; j < l.size(); j++) {
      const std::string p = prefix + std::to_string(j) + ".";
      Layer L; L.kind = l[j].kind;
      if (l[j].kind == 0) {
        ResDesc& r = L.r; r.cin = l[j].cin; r.cout = l[j].cout; r.updown = l[j].updown; r.groups = GN_G;
        r.gn1_w = take(r.cin); r.gn1_b = take(r.cin);
        u->add_entry(p + "in_layers.0.weight", r.gn1_w, 1, r.cin); u->add_entry(p + "in_layers.0.bias", r.gn1_b, 1, r.cin);
        r.c1_w = take((long)r.cout * r.cin * 3); r.c1_b = take(r.cout);
        u->add_entry(p + "in_layers.2.weight", r.c1_w, 3, r.cout, r.cin, 3); u->add_entry(p + "in_layers.2.bias", r.c1_b, 1, r.cout);
        r.emb_col = emb_col; emb_col += ew * r.cout; r.ssn = ssn ? 1 : 0;
        u->add_entry(p + "emb_layers.1.weight", u->off_emb_w + (long)r.emb_col * te, 2, ew * r.cout, te);
        u->add_entry(p + "emb_layers.1.bias", u->off_emb_b + r.emb_col, 1, ew * r.cout);
        r.gn2_w = take(r.cout); r.gn2_b = take(r.cout);
        u->add_entry(p + "out_layers.0.weight", r.gn2_w, 1, r.cout); u->add_entry(p + "out_layers.0.bias", r.gn2_b, 1, r.cout);
        r.c2_w = take((long)r.cout * r.cout * 3); r.c2_b = take(r.cout);
        u->add_entry(p + "out_layers.3.weight", r.c2_w, 3, r.cout, r.cout, 3); u->add_entry(p + "out_layers.3.bias", r.c2_b, 1, r.cout);
        r.sk_w = r.sk_b = -1;
        if (r.cin != r.cout) {
          r.sk_w = take((long)r.cout * r.cin); r.sk_b = take(r.cout);
          u->add_entry(p + "skip_connection.weight", r.sk_w, 3, r.cout, r.cin, 1); u->add_entry(p + "skip_connection.bias", r.sk_b, 1, r.cout);
        }
      } else if (l[j].kind == 2) {
        RsDesc& d = L.s; d.c = l[j].cin; d.up = l[j].updown == 2; d.conv = rs_conv ? 1 : 0;
        if (d.conv) {      // Downsample.op / Upsample.conv (unet.py:188-190,211-213)
          const std::string nm = d.up ? "conv" : "op";
          d.w = take((long)d.c * d.c * 3); d.b = take(d.c);
          u->add_entry(p + nm + ".weight", d.w, 3, d.c, d.c, 3); u->add_entry(p + nm + ".bias", d.b, 1, d.c);
        }
      } else {
        AttnDesc& a = L.a; a.c = l[j].cin; a.heads = l[j].heads;
        a.n_w = take(a.c); a.n_b = take(a.c);
        u->add_entry(p + "norm.weight", a.n_w, 1, a.c); u->add_entry(p + "norm.bias", a.n_b, 1, a.c);
        a.qkv_w = take((long)3 * a.c * a.c); a.qkv_b = take(3 * a.c);
        u->add_entry(p + "qkv.weight", a.qkv_w, 3, 3 * a.c, a.c, 1); u->add_entry(p + "qkv.bias", a.qkv_b, 1, 3 * a.c);
        a.pr_w = take((long)a.c * a.c); a.pr_b = take(a.c);
        u->add_entry(p + "proj_out.weight", a.pr_w, 3, a.c, a.c, 1); u->add_entry(p + "proj_out.bias", a.pr_b, 1, a.c);
      }
      blk.layers.push_back(L);
    }
    blk.cin = l.front().cin; blk.cout = l.back().cout;
  };
  u->in_blocks.resize(inp.size());
  for (size_t i = 1; i < inp.size(); i++) lay("input_blocks." + std::to_string(i) + ".", inp[i], u->in_blocks[i]);
  u->off_mid_begin = off;
  lay("middle_block.", mid, u->mid);
  u->out_blocks.resize(outp.size());
  for (size_t i = 0; i < outp.size(); i++) lay("output_blocks." + std::to_string(i) + ".", outp[i], u->out_blocks[i]);
  u->off_out_gw = take(mc); u->off_out_gb = take(mc);
  u->add_entry("out.0.weight", u->off_out_gw, 1, mc); u->add_entry("out.0.bias", u->off_out_gb, 1, mc);
  u->off_out_w = take((long)c.out_channels * mc * 3); u->off_out_b = take(c.out_channels);
  u->add_entry("out.2.weight", u->off_out_w, 3, c.out_channels, mc, 3); u->add_entry("out.2.bias", u->off_out_b, 1, c.out_channels);
  u->nparams = off;
  return 0;
}

int block_out_len(const Block& b, int Lin) {
  int L = Lin;
  for (auto& l : b.layers) L = layer_len(l, L);
  return L;
}

// runs the layers of one block; the LAST layer writes into `out`
int block_forward(eegldm_unet* u, const Block& b, View x, int B, int& L, const View& out) {
  for (size_t j = 0; j < b.layers.size(); j++) {
    const Layer& l = b.layers[j];
    const bool last = j + 1 == b.layers.size();
    const int cout = layer_cout(l);
    const int Lo = layer_len(l, L);
    View y = out;
    if (!last) { ALLOC_OR_FAIL(y.p, u->alloc_act((long)B * Lo, cout)); y.ld = cout; y.C = cout; }
    if (l.kind == 0) EEG_TRY(res_forward(u, l.r, x, B, L, y));
    else if (l.kind == 1) EEG_TRY(attn_forward(u, l.a, x, B, L, y));
    else { RsTape t; EEG_TRY(resample_forward(u, l.s, x, B, L, y, &t)); u->st.push_back(t); }
    x = y; L = Lo;
  }
  return 0;
}

// backward through a block: dout = gradient of the block output; dx_dest = where the gradient of
// the block input goes.  Tapes are consumed from the back of u->rt / u->at.
// extra (optional): added into dx_dest (skip gradient of the input path); fused into the block's last GroupNorm kernel when it can, else add_rows
int block_backward(eegldm_unet* u, const Block& b, View dout, const View& dx_dest, int B, size_t& ri, size_t& ai, float* demb_all,
                   const View* extra = nullptr, long extra_rows = 0) {
  for (int j = (int)b.layers.size() - 1; j >= 0; j--) {
    const Layer& l = b.layers[j];
    View dx = dx_dest;
    if (j > 0) {
      const int cin = layer_cin(l);
      const long rows = l.kind == 0 ? (long)u->rt[ri - 1].B * u->rt[ri - 1].Lin
                                    : (l.kind == 1 ? (long)u->at[ai - 1].B * u->at[ai - 1].T : (long)u->st.back().B * u->st.back().Lin);
      ALLOC_OR_FAIL(dx.p, u->alloc_act(rows, cin)); dx.ld = cin; dx.C = cin;
    }
    int fused = 0;
    if (l.kind == 0) { EEG_TRY(res_backward(u, l.r, u->rt[--ri], dout, dx, demb_all, j == 0 ? extra : nullptr, &fused)); }
    else if (l.kind == 1) { EEG_TRY(attn_backward(u, l.a, u->at[--ai], dout, dx)); }
    else { const RsTape t = u->st.back(); u->st.pop_back(); EEG_TRY(resample_backward(u, l.s, t, dout, dx)); }      // (consumed from the back, like rt / at)
    if (j == 0 && extra && !fused) EEG_TRY(ew_add_rows(u->ctx, dx.p, dx.ld, extra->p, extra->ld, extra_rows, extra->C, u->dtype));
    dout = dx;
  }
  return 0;
}

}  // namespace

// ================================================================== C ABI
extern "C" int eegldm_unet_create(eegldm_ctx* ctx, const eegldm_unet_cfg* cfg, eegldm_unet** out) {
  EEG_CHECK(ctx && cfg && out, "null argument");
  EEG_CHECK(cfg->num_heads >= 0 && cfg->num_heads <= 64 && cfg->num_heads_upsample <= 64, "bad num_heads");
  {      // every attention block: channels divisible by its head count, and head width a multiple of 8 (16-byte column views of the qkv rows)
    const int nh = cfg->num_heads > 0 ? cfg->num_heads : 1, nu = cfg->num_heads_upsample > 0 ? cfg->num_heads_upsample : nh;
    for (int i = 0; i < cfg->n_mult; i++) {
      const int chn = cfg->channel_mult[i] * cfg->model_channels;
      if (cfg->num_head_channels > 0) EEG_CHECK(chn % cfg->num_head_channels == 0 && cfg->num_head_channels % 8 == 0,
                                                "num_head_channels=%d must divide %d channels and be a multiple of 8", cfg->num_head_channels, chn);
      else EEG_CHECK(chn % nh == 0 && chn % nu == 0 && (chn / nh) % 8 == 0 && (chn / nu) % 8 == 0,
                     "%d channels / %d (%d) heads: the head width must be a whole multiple of 8", chn, nh, nu);
    }
  }
  EEG_CHECK(cfg->model_channels % 32 == 0, "model_channels must be a multiple of 32 (GroupNorm(32))");
  EEG_CHECK(cfg->n_mult >= 1 && cfg->n_mult <= 8 && cfg->n_attn >= 0 && cfg->n_attn <= 8, "bad channel_mult / attention_resolutions");
  EEG_CHECK(cfg->dtype == EEGLDM_F32 || cfg->dtype == EEGLDM_BF16 || cfg->dtype == EEGLDM_F16, "bad dtype");
  eegldm_unet* u = new eegldm_unet();
  u->ctx = ctx; u->cfg = *cfg; u->dtype = cfg->dtype;
  int rc = build_plan(u);
  if (rc) { delete u; return rc; }
  *out = u;
  return 0;
}
eegldm_ctx* unet_ctx(const eegldm_unet* u) { return u->ctx; }
int unet_in_channels(const eegldm_unet* u) { return u->cfg.in_channels; }
int unet_out_channels(const eegldm_unet* u) { return u->cfg.out_channels; }
extern "C" int eegldm_unet_destroy(eegldm_unet* u) {
  if (!u) return 0;
  sampler_release(u);
  if (u->fuse_stats) (void)hipFree(u->fuse_stats);
  delete u;
  return 0;
}
// ResBlock dropout (unet.py:289; config_ldm.yaml:38 sets 0.0): probability for training-mode forwards and the seed of the mask stream
// (the counter restarts, so the same seed reproduces the same masks for the same sequence of forwards)
extern "C" int eegldm_unet_set_dropout(eegldm_unet* u, float p, uint64_t seed) {
  EEG_CHECK(u, "null unet");
  EEG_CHECK(p >= 0.f && p < 1.f, "dropout probability %g outside [0, 1)", (double)p);
  u->dropout = p; u->drop_seed = seed; u->drop_ctr = 0;
  return 0;
}
extern "C" int eegldm_unet_num_entries(const eegldm_unet* u) { return (int)u->entries.size(); }
extern "C" long eegldm_unet_num_params(const eegldm_unet* u) { return u->nparams; }
extern "C" int eegldm_unet_set_grad_hook(eegldm_unet* u, eegldm_grad_hook fn, void* user) {
  EEG_CHECK(u, "null unet");
  u->grad_hook = fn; u->grad_hook_user = user;
  return 0;
}
extern "C" int eegldm_unet_entry(const eegldm_unet* u, int i, char* name, int cap, long* offset, long* numel, int* ndim, int shape[3]) {
  return entry_query(u, i, name, cap, offset, numel, ndim, shape);
}
extern "C" int eegldm_unet_bind(eegldm_unet* u, float* params, float* grads) {
  EEG_CHECK(u && params, "null argument");
  // a captured sampling graph (sampler.hip) bakes in the addresses of the bound parameter buffer and of the weight copies derived
  // from it: re-binding must drop those graphs, or the next eegldm_sample would replay against the old (possibly freed) memory
  sampler_release(u);
  return u->bind(params, grads);
}
extern "C" int eegldm_unet_sync_weights(eegldm_unet* u) {
  EEG_CHECK(u, "null argument");
  return u->sync_weights();
}

// temb -> Linear -> SiLU -> Linear -> SiLU -> every ResBlock's embedding projection in one Linear (unet.py:526-529, 316); n rows
static int embed_chain(eegldm_unet* u, const int64_t* tsteps, int n, float* e0, float* h1e, float* a1e, float* emb, float* semb, float* emb_all) {
  eegldm_ctx* ctx = u->ctx; const int mc = u->mc, te = u->te, F = EEGLDM_F32;
  EEG_TRY(ew_temb(ctx, tsteps, e0, n, mc, F));
  EEG_TRY(op_linear(ctx, F, e0, mc, u->P(u->off_te0_w), mc, u->P(u->off_te0_b), h1e, te, n, te, mc, 1));
  EEG_TRY(ew_silu(ctx, h1e, a1e, (long)n * te, F));
  EEG_TRY(op_linear(ctx, F, a1e, te, u->P(u->off_te2_w), te, u->P(u->off_te2_b), emb, te, n, te, te, 1));
  EEG_TRY(ew_silu(ctx, emb, semb, (long)n * te, F));
  return op_linear(ctx, F, semb, te, u->P(u->off_emb_w), te, u->P(u->off_emb_b), emb_all, u->etot, n, u->etot, te, 1);
}
// The sampler's timesteps are known before the loop starts and shared by all samples: their embedding rows as ONE batch
// (table[i] = the etot projections of tsteps[i]; work: n * unet_embed_work_floats() floats), instead of six tiny launches per step.
int unet_emb_width(const eegldm_unet* u) { return u->etot; }
long unet_embed_work_floats(const eegldm_unet* u) { return (long)u->mc + 4L * u->te; }
int unet_embed_table(eegldm_unet* u, const int64_t* tsteps_dev, int n, float* table, float* work) {
  EEG_CHECK(u && u->params && tsteps_dev && table && work && n > 0, "bad argument");
  const long te = u->te;
  float* e0 = work; float* h1e = e0 + (long)n * u->mc; float* a1e = h1e + n * te; float* emb = a1e + n * te; float* semb = emb + n * te;
  return embed_chain(u, tsteps_dev, n, e0, h1e, a1e, emb, semb, table);
}
void unet_set_shared_emb(eegldm_unet* u, const float* row) { u->emb_shared = row; }

extern "C" int eegldm_unet_forward(eegldm_unet* u, const float* x, const int64_t* tsteps, float* y, int B, int L, int training) {
  EEG_CHECK(u && x && tsteps && y, "null argument");
  EEG_CHECK(u->params, "bind parameters first");
  EEG_CHECK(B > 0 && L > 0 && (L % (1 << (u->cfg.n_mult - 1))) == 0, "L=%d must be divisible by 2^(levels-1)", L);
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype; const int mc = u->mc, te = u->te;
  u->arena.reset(); u->rt.clear(); u->at.clear(); u->st.clear(); u->in_out.clear(); u->cat.clear();
  u->B = B; u->L = L; u->have_tape = false; u->training = training != 0;

  // ---- eval-mode GroupNorm fusion for few-row launches (NetBase::eval_fuse): one statistics area per ResBlock
  {
    EEG_ENV_VAR(bool, no_fuse, getenv("EEGLDM_NO_EVAL_GN_FUSE") != nullptr);
    u->eval_fuse = !training && !no_fuse && (dt == EEGLDM_BF16 || dt == EEGLDM_F16); u->fused_used = false; u->fuse_used = 0; u->part_reg.clear();
    if (u->eval_fuse) {
      // slots: B * L_out / 16 * cout / 4 per ResBlock; an upper bound from the widest / longest block keeps this simple
      size_t nres = 0; int cmax = 0;
      auto count = [&](const Block& b) { for (auto& l : b.layers) { nres++; if (l.kind == 0 && l.r.cout > cmax) cmax = l.r.cout; } };   // one area per ResBlock, one more per AttentionBlock
      for (auto& b : u->in_blocks) count(b);
      count(u->mid);
      for (auto& b : u->out_blocks) count(b);
      const size_t need = 2 * nres * (size_t)B * (L / 16 + 1) * (cmax / 4);      // a ResBlock uses two areas (conv1's output, its own output)
      if (need > (size_t)4 << 20) u->eval_fuse = false;               // few-row launches only (32 MB of slots at most)
      else if (u->fuse_cap < need) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (u->fuse_stats) (void)hipFree(u->fuse_stats);
        u->fuse_stats = nullptr; u->fuse_cap = 0;
        HIP_TRY(hipMalloc(&u->fuse_stats, sizeof(float2) * need)); u->fuse_cap = need;
      }
    }
  }

  // ---- timestep embedding MLP + all ResBlock embedding projections (unet.py:526-529, 316).
  // Tiny (B x 4mc): always fp32 on the fp32 master weights, whatever the activation dtype.
  if (u->emb_shared && !training) {
    // the sampler computed this timestep's row once for the whole run (unet_embed_table): every sample shares it (row stride 0)
    u->emb_all = const_cast<float*>(u->emb_shared); u->emb_ld = 0;
  } else {
    auto fbuf = [&](long n) { return (float*)u->arena.alloc(sizeof(float) * (size_t)n); };
    ALLOC_OR_FAIL(u->e0, fbuf((long)B * mc));
    ALLOC_OR_FAIL(u->h1e, fbuf((long)B * te));
    ALLOC_OR_FAIL(u->a1e, fbuf((long)B * te));
    ALLOC_OR_FAIL(u->emb, fbuf((long)B * te));
    ALLOC_OR_FAIL(u->semb, fbuf((long)B * te));
    ALLOC_OR_FAIL(u->emb_all, fbuf((long)B * u->etot));
    u->emb_ld = u->etot;
    EEG_TRY(embed_chain(u, tsteps, B, u->e0, u->h1e, u->a1e, u->emb, u->semb, u->emb_all));
  }

  // ---- concat buffers: output block j consumes [h (c1) | skip (ich)] where skip = input block n_in-1-j
  const int n_in = (int)u->in_blocks.size(), n_out = (int)u->out_blocks.size();
  std::vector<int> in_len(n_in), in_ch(n_in);
  { int Lc = L; in_len[0] = L; in_ch[0] = mc;
    for (int i = 1; i < n_in; i++) { Lc = block_out_len(u->in_blocks[i], Lc); in_len[i] = Lc; in_ch[i] = u->in_blocks[i].cout; } }
  u->cat.resize(n_out);
  for (int j = 0; j < n_out; j++) {
    const int i = n_in - 1 - j, c1 = u->skip_c1[j], C = c1 + in_ch[i];
    ALLOC_OR_FAIL(u->cat[j].p, u->alloc_act((long)B * in_len[i], C)); u->cat[j].ld = C; u->cat[j].C = C;
  }
  u->in_out.resize(n_in);
  for (int i = 0; i < n_in; i++) { const int j = n_in - 1 - i; u->in_out[i] = col_view(u->cat[j], u->skip_c1[j], in_ch[i], dt); }

  // ---- input path
  ALLOC_OR_FAIL(u->x0.p, u->alloc_act((long)B * L, u->cfg.in_channels)); u->x0.ld = u->cfg.in_channels; u->x0.C = u->cfg.in_channels;
  EEG_TRY(eegldm_ncl_to_nlc(ctx, x, u->x0.p, u->x0.ld, B, u->cfg.in_channels, L, dt));
  EEG_TRY(op_conv_fwd(ctx, dt, u->x0.p, u->x0.ld, u->W(u->off_cin_w), u->P(u->off_cin_b), u->in_out[0].p, u->in_out[0].ld, B, L,
                      u->cfg.in_channels, mc, 3, 1, 1, 1, nullptr, 0, nullptr, 0));
  View h = u->in_out[0]; int Lc = L;
  for (int i = 1; i < n_in; i++) { EEG_TRY(block_forward(u, u->in_blocks[i], h, B, Lc, u->in_out[i])); h = u->in_out[i]; }
  // ---- middle: writes into columns [0, c1) of concat buffer 0
  { View dst = col_view(u->cat[0], 0, u->skip_c1[0], dt); EEG_TRY(block_forward(u, u->mid, h, B, Lc, dst)); }
  // ---- output path
  for (int j = 0; j < n_out; j++) {
    View dst;
    if (j + 1 < n_out) dst = col_view(u->cat[j + 1], 0, u->skip_c1[j + 1], dt);
    else { ALLOC_OR_FAIL(dst.p, u->alloc_act((long)B * L, mc)); dst.ld = mc; dst.C = mc; }
    int Lj = in_len[n_in - 1 - j];
    EEG_TRY(block_forward(u, u->out_blocks[j], u->cat[j], B, Lj, dst));
    h = dst;
  }
  // ---- out: GN + SiLU + conv3 (unet.py:501-505)
  u->h_last = h;
  ALLOC_OR_FAIL(u->st_out, (float*)u->arena.alloc(sizeof(float) * 2 * B * GN_G));
  ALLOC_OR_FAIL(u->a_out.p, u->alloc_act((long)B * L, mc)); u->a_out.ld = mc; u->a_out.C = mc;
  EEG_TRY(eegldm_groupnorm_fwd(ctx, h.p, h.ld, u->P(u->off_out_gw), u->P(u->off_out_gb), u->a_out.p, mc, u->st_out, B, L, mc, GN_G, GN_EPS, 1, 0, nullptr, 0, dt));
  Arena::Mark mk = u->arena.mark();
  void* yo; ALLOC_OR_FAIL(yo, u->alloc_act((long)B * L, u->cfg.out_channels));
  EEG_TRY(op_conv_fwd(ctx, dt, u->a_out.p, mc, u->W(u->off_out_w), u->P(u->off_out_b), yo, u->cfg.out_channels, B, L, mc, u->cfg.out_channels,
                      3, 1, 1, 1, nullptr, 0, nullptr, 0));
  EEG_TRY(eegldm_nlc_to_ncl(ctx, yo, u->cfg.out_channels, y, B, u->cfg.out_channels, L, dt));
  u->arena.release(mk);
  u->have_tape = !u->fused_used;      // a fused eval forward did not keep what the backward needs
  return 0;
}

extern "C" int eegldm_unet_backward(eegldm_unet* u, const float* dy, float* dx_out) {
  EEG_CHECK(u && dy, "null argument");
  EEG_CHECK(u->have_tape, "call eegldm_unet_forward first (with training != 0: an eval-mode forward does not keep the activations)");
  EEG_CHECK(u->grads, "no gradient buffer bound");
  eegldm_ctx* ctx = u->ctx; const int dt = u->dtype; const int mc = u->mc, te = u->te, B = u->B, L = u->L;
  const int cin = u->cfg.in_channels, cout = u->cfg.out_channels;
  const int n_in = (int)u->in_blocks.size(), n_out = (int)u->out_blocks.size();
  u->have_tape = false;   // the tape is consumed
  // Weight gradients of the GEMM-path convs are RECORDED during the backward chain and launched grouped by shape (ops.hip:
  // op_wgrad_flush) when their slice of the gradient buffer is due: at the all-reduce hook (out / output_blocks / middle_block) and
  // at the end.  EEGLDM_NO_GROUPED_WGRAD=1 restores one launch per layer on the side stream.
  EEG_ENV_VAR(bool, no_grouped, getenv("EEGLDM_NO_GROUPED_WGRAD") != nullptr);
  struct DeferGuard {
    eegldm_ctx* c; bool on;
    DeferGuard(eegldm_ctx* ctx_, bool on_) : c(ctx_), on(on_) { if (on) { c->defer_wgrad = true; c->grp_slot = 0; c->gn_fold_count = 0; c->gn_fold_pending.clear(); } }
    ~DeferGuard() { if (on) { c->defer_wgrad = false; c->wgrad_pending.clear(); c->gn_fold_pending.clear(); } }
  } defer_guard(ctx, !no_grouped && u->param_grads);

  float* demb_all; ALLOC_OR_FAIL(demb_all, (float*)u->arena.alloc(sizeof(float) * (size_t)B * u->etot));
  // ---- out conv + GN
  View dyv; ALLOC_OR_FAIL(dyv.p, u->alloc_act((long)B * L, cout)); dyv.ld = cout;
  EEG_TRY(eegldm_ncl_to_nlc(ctx, dy, dyv.p, cout, B, cout, L, dt));
  EEG_TRY(op_conv_wgrad(ctx, dt, u->a_out.p, mc, dyv.p, cout, u->G(u->off_out_w), u->G(u->off_out_b), B, L, mc, cout, 3, 1, 1, 1));
  View da; ALLOC_OR_FAIL(da.p, u->alloc_act((long)B * L, mc)); da.ld = mc;
  EEG_TRY(op_conv_dgrad(ctx, dt, dyv.p, cout, u->W(u->off_out_w), da.p, mc, B, L, mc, cout, 3, 1, 1, 1, nullptr, 0));
  View dh; ALLOC_OR_FAIL(dh.p, u->alloc_act((long)B * L, mc)); dh.ld = mc; dh.C = mc;
  {
    int gn_deferred = 0;       // (slot fold batched with the others in the grouped mode)
    EEG_TRY(op_groupnorm_bwd(ctx, u->h_last.p, u->h_last.ld, u->P(u->off_out_gw), u->P(u->off_out_gb), u->st_out, da.p, mc, dh.p, mc,
                             u->G(u->off_out_gw), u->G(u->off_out_gb), B, L, mc, GN_G, 1, 0, nullptr, 0, dt, nullptr, 0, nullptr, nullptr, 0, nullptr,
                             u->param_grads ? &gn_deferred : nullptr));
    if (gn_deferred == 1) EEG_TRY(op_gn_slot_reduce_deferred(ctx, u->G(u->off_out_gw), u->G(u->off_out_gb), mc));
  }
  // ---- output blocks, reversed.  Each yields d(cat_j); its skip half is kept for the input path.
  size_t ri = u->rt.size(), ai = u->at.size();
  std::vector<View> dskip(n_in);
  std::vector<int> in_len(n_in), in_ch(n_in);
  { int Lc = L; in_len[0] = L; in_ch[0] = mc;
    for (int i = 1; i < n_in; i++) { Lc = block_out_len(u->in_blocks[i], Lc); in_len[i] = Lc; in_ch[i] = u->in_blocks[i].cout; } }
  View dout = dh;
  for (int j = n_out - 1; j >= 0; j--) {
    const int i = n_in - 1 - j, C = u->cat[j].C;
    View dcat; ALLOC_OR_FAIL(dcat.p, u->alloc_act((long)B * in_len[i], C)); dcat.ld = C; dcat.C = C;
    EEG_TRY(block_backward(u, u->out_blocks[j], dout, dcat, B, ri, ai, demb_all));
    dout = col_view(dcat, 0, u->skip_c1[j], dt);
    dskip[i] = col_view(dcat, u->skip_c1[j], in_ch[i], dt);
  }
  // ---- middle
  { View g; ALLOC_OR_FAIL(g.p, u->alloc_act((long)B * in_len[n_in - 1], in_ch[n_in - 1])); g.ld = in_ch[n_in - 1]; g.C = g.ld;
    // (the skip gradient of the deepest input block joins here; each input block's backward then adds the next one)
    EEG_TRY(block_backward(u, u->mid, dout, g, B, ri, ai, demb_all, n_in > 1 ? &dskip[n_in - 1] : nullptr, (long)B * in_len[n_in - 1]));
    dout = g; }
  // gradients of out / output_blocks / middle_block are complete (in stream order): the host may start reducing them
  // across ranks while the input blocks' backward runs
  EEG_TRY(u->flush_gn_folds());   // the last middle ResBlock's deferred dgamma / dbeta fold
  if (u->grad_hook) { EEG_TRY(op_wgrad_flush(ctx)); EEG_TRY(op_gn_fold_flush(ctx)); }     // the hook promises final gradients for this slice: launch what is pending for it
  if (u->grad_hook) u->grad_hook(u->grad_hook_user, u->off_mid_begin, u->nparams - u->off_mid_begin);
  // ---- input blocks, reversed: gradient of block i's output = consumer's dx + skip gradient
  for (int i = n_in - 1; i >= 1; i--) {
    View g; ALLOC_OR_FAIL(g.p, u->alloc_act((long)B * in_len[i - 1], in_ch[i - 1])); g.ld = in_ch[i - 1]; g.C = g.ld;
    EEG_TRY(block_backward(u, u->in_blocks[i], dout, g, B, ri, ai, demb_all, &dskip[i - 1], (long)B * in_len[i - 1]));
    dout = g;
  }
  if (n_in == 1) EEG_TRY(ew_add_rows(ctx, dout.p, dout.ld, dskip[0].p, dskip[0].ld, (long)B * L, mc, dt));
  EEG_TRY(u->flush_gn_folds());
  // ---- conv_in
  EEG_TRY(op_conv_wgrad(ctx, dt, u->x0.p, u->x0.ld, dout.p, dout.ld, u->G(u->off_cin_w), u->G(u->off_cin_b), B, L, cin, mc, 3, 1, 1, 1));
  if (dx_out) {
    void* dx0; ALLOC_OR_FAIL(dx0, u->alloc_act((long)B * L, cin));
    EEG_TRY(op_conv_dgrad(ctx, dt, dout.p, dout.ld, u->W(u->off_cin_w), dx0, cin, B, L, cin, mc, 3, 1, 1, 1, nullptr, 0));
    EEG_TRY(eegldm_nlc_to_ncl(ctx, dx0, cin, dx_out, B, cin, L, dt));
  }
  // ---- embedding MLP backward (fp32)
  const int E = u->etot, F = EEGLDM_F32;
  auto fbuf = [&](long n) { return (float*)u->arena.alloc(sizeof(float) * (size_t)n); };
  EEG_TRY(ew_colsum(ctx, demb_all, E, nullptr, 0, u->G(u->off_emb_b), 1, B, E, F));
  EEG_TRY(op_linear_wgrad(ctx, F, u->semb, te, demb_all, E, u->G(u->off_emb_w), te, B, E, te));
  float* dsemb; ALLOC_OR_FAIL(dsemb, fbuf((long)B * te));
  EEG_TRY(op_linear_dgrad(ctx, F, demb_all, E, u->P(u->off_emb_w), te, dsemb, te, B, E, te, 1));
  float* demb; ALLOC_OR_FAIL(demb, fbuf((long)B * te));
  EEG_TRY(ew_silu_bwd(ctx, dsemb, u->emb, demb, (long)B * te, F));
  EEG_TRY(ew_colsum(ctx, demb, te, nullptr, 0, u->G(u->off_te2_b), 1, B, te, F));
  EEG_TRY(op_linear_wgrad(ctx, F, u->a1e, te, demb, te, u->G(u->off_te2_w), te, B, te, te));
  float* da1; ALLOC_OR_FAIL(da1, fbuf((long)B * te));
  EEG_TRY(op_linear_dgrad(ctx, F, demb, te, u->P(u->off_te2_w), te, da1, te, B, te, te, 1));
  float* dh1; ALLOC_OR_FAIL(dh1, fbuf((long)B * te));
  EEG_TRY(ew_silu_bwd(ctx, da1, u->h1e, dh1, (long)B * te, F));
  EEG_TRY(ew_colsum(ctx, dh1, te, nullptr, 0, u->G(u->off_te0_b), 1, B, te, F));
  EEG_TRY(op_linear_wgrad(ctx, F, u->e0, mc, dh1, te, u->G(u->off_te0_w), mc, B, te, mc));
  EEG_TRY(op_wgrad_flush(ctx));      // the remaining (input-block) weight gradients, grouped by shape
  EEG_TRY(op_gn_fold_flush(ctx));    // and the remaining GroupNorm dgamma / dbeta folds
  return 0;
}

extern "C" int eegldm_ldm_train_step(eegldm_unet* u, const float* latents, const float* noise, const int64_t* t, const float* acp,
                                     int pred_type, int B, int L, float grad_scale, float* loss) {
  EEG_CHECK(u && latents && noise && t && acp && loss, "null argument");
  EEG_CHECK(pred_type == EEGLDM_PRED_EPSILON || pred_type == EEGLDM_PRED_V, "prediction type must be epsilon or v_prediction");
  eegldm_ctx* ctx = u->ctx;
  const int C = u->cfg.in_channels;
  EEG_CHECK(u->cfg.out_channels == C, "LDM training needs in_channels == out_channels");
  const long n = (long)B * C * L;
  // small fp32 NCL staging buffers; they must outlive the arena reset inside forward, so they live in a side arena block
  static thread_local float* stage = nullptr; static thread_local size_t stage_cap = 0;
  if (stage_cap < (size_t)n * 4) {
    if (stage) hipFree(stage);
    HIP_TRY(hipMalloc(&stage, sizeof(float) * n * 4)); stage_cap = (size_t)n * 4;
  }
  float *noisy = stage, *pred = stage + n, *target = stage + 2 * n, *dpred = stage + 3 * n;
  EEG_TRY(eegldm_add_noise(ctx, latents, noise, t, acp, noisy, B, (long)C * L));
  EEG_TRY(eegldm_unet_forward(u, noisy, t, pred, B, L, 1));
  const float* tgt = noise;
  if (pred_type == EEGLDM_PRED_V) { EEG_TRY(eegldm_get_velocity(ctx, latents, noise, t, acp, target, B, (long)C * L)); tgt = target; }
  EEG_TRY(eegldm_mse_loss(ctx, pred, tgt, loss, dpred, n, grad_scale));
  return eegldm_unet_backward(u, dpred, nullptr);
}
