// Shared host-side pieces of the model executors (UNet, AutoencoderKL, PatchDiscriminator):
// bump arena for activations, NLC views, the flat-parameter base struct, and the
// ResBlock / AttentionBlock forward+backward sequences both model families use.
#pragma once
#include <string>
#include <vector>

#include "common.h"
#include "internal.h"

struct Arena {
  struct Block { char* p; size_t cap; };
  std::vector<Block> blocks;
  size_t cur_block = 0, cur_off = 0;
  size_t min_block = (size_t)512 << 20;
  ~Arena() { for (auto& b : blocks) (void)hipFree(b.p); }
  void reset() { cur_block = 0; cur_off = 0; }
  struct Mark { size_t b, o; };
  Mark mark() const { return {cur_block, cur_off}; }
  void release(Mark m) { cur_block = m.b; cur_off = m.o; }
  void* alloc(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    while (true) {
      if (cur_block < blocks.size()) {
        if (cur_off + bytes <= blocks[cur_block].cap) { void* r = blocks[cur_block].p + cur_off; cur_off += bytes; return r; }
        cur_block++; cur_off = 0;
        continue;
      }
      size_t cap = bytes > min_block ? bytes : min_block;
      char* p = nullptr;
      if (hipMalloc(&p, cap) != hipSuccess) return nullptr;
      blocks.push_back({p, cap});
    }
  }
};

struct View { void* p = nullptr; long ld = 0; int C = 0; };
static inline View col_view(const View& v, int col, int C, int dtype) {
  View r; r.p = (char*)v.p + (size_t)col * dtype_size(dtype); r.ld = v.ld; r.C = C; return r;
}

struct Entry { std::string name; long offset, numel; int ndim; int shape[3]; };

struct ResDesc {
  int cin, cout, updown;   // updown: 0 none, 1 down (AvgPool2), 2 up (nearest x2) -- UNet only
  int groups;              // GroupNorm groups (32 in the UNet, norm_num_groups in the AutoencoderKL)
  long gn1_w, gn1_b, c1_w, c1_b, gn2_w, gn2_b, c2_w, c2_b, sk_w, sk_b;
  int emb_col;             // column in the batched embedding projection, or -1 (no timestep embedding)
  // use_scale_shift_norm (unet.py:279-284,318-322): the projection is 2 * cout wide, columns [emb_col, emb_col + cout) = scale and
  // [emb_col + cout, emb_col + 2 cout) = shift, and h = GroupNorm(conv1(..)) * (1 + scale) + shift ahead of SiLU + conv2 (no h + emb)
  int ssn = 0;
};
// heads: QKVAttentionLegacy (unet.py:97-125) -- the qkv projection's output channels are [q_0 | k_0 | v_0 | q_1 | k_1 | v_1 | ...], c / heads each
struct AttnDesc { int c; long n_w, n_b, qkv_w, qkv_b, pr_w, pr_b; int heads = 1; };
// Downsample / Upsample layers of resblock_updown = False (unet.py:177-224): conv != 0: Conv1d(c, c, 3, stride 2, padding 1) /
// nearest x 2 + Conv1d(c, c, 3, padding 1); conv == 0: AvgPool1d(2, 2) / nearest x 2 only (w = b = -1)
struct RsDesc { int c = 0, up = 0, conv = 0; long w = -1, b = -1; };
struct RsTape { View x, xu; int B, Lin, Lout; };
struct ResTape { View x, a1, xr, h1, a2, hn; float *st1, *st2; int B, Lin, Lout; unsigned long long drop_off = 0; bool dropped = false; };
struct AttnTape { View x, xn, qkv, o; void* probs; float* st; int B, T; };

struct NetBase {
  eegldm_ctx* ctx = nullptr;
  int dtype = EEGLDM_F32;
  std::vector<Entry> entries;
  long nparams = 0;
  float* params = nullptr; float* grads = nullptr;
  void* wT = nullptr;            // compute-dtype copy of params (bf16) or == params (fp32)
  bool owns_wT = false;
  // K-blocked second copy of the 3-tap conv weights that run on the implicit-GEMM path (16-bit dtypes): [tap][Cin / 32][Cout][32].
  // The forward conv's weight tile of one K stage is then ONE contiguous run instead of 64-byte row segments -- the L2 -> LDS path
  // moves 64-byte segments at half the rate of >= 128-byte ones (tools/probes/fill_pattern_probe.hip: 29 vs 48-58 B/clk/CU).
  // Same offsets as wT; refreshed by sync_weights(); registered in ctx->kblk so that op_conv_fwd finds it by the plain weight's address.
  void* wK = nullptr; void* d_kb = nullptr; int n_kb = 0; long kb_chunks = 0;
  // data-gradient copies [tap][Cout / 32][Cin][32] of the 3-tap weights whose data gradient fits the 192 x 256 tile (gemm_big.hip): same offsets, ctx->kblk_t
  void* wKT = nullptr; void* d_kbt = nullptr; int n_kbt = 0; long kbt_chunks = 0;
  // stride-2 64 -> 128 convs (the PatchDiscriminator's second layer) on the weight-stationary kernel: per eligible weight a forward and a
  // data-gradient copy of [3][128][128] elements (elementwise.hip s2ws_pack), refreshed by sync_weights(), registered in ctx->s2ws_f / _d
  void* wS2 = nullptr; std::vector<long> s2_offs;
  // dgamma / dbeta folds of a ResBlock's first GroupNorm, launched one block later on the side stream (see res_backward)
  struct GnFold { float* dgamma; float* dbeta; int C, region; };
  std::vector<GnFold> gn_pending; int gn_parity = 0;
  int flush_gn_folds();
  bool param_grads = true;       // false: backward propagates to the input only (G step through D)
  // nn.Dropout(p) of ResBlock.out_layers (unet.py:289; every reference yaml: 0).  Active in training-mode forwards only; the mask of each
  // block is Philox(drop_seed, counter) and is regenerated by the backward (ResTape::drop_off), never stored.
  float dropout = 0.f; bool training = false; unsigned long long drop_seed = 0x0D50ull, drop_ctr = 0;
  Arena arena;
  float* emb_all = nullptr; int etot = 0;   // batched timestep-embedding projections (UNet)
  long emb_ld = 0;                           // row stride of emb_all: etot, or 0 when all samples share one precomputed row (sampler)
  const float* emb_shared = nullptr;         // set by the sampler for the duration of an eval forward (unet_set_shared_emb)
  // Eval-mode forward of a few-row launch (sampling one window per call): the second GroupNorm of a ResBlock is not launched -- conv1
  // (conv_skinny.hip) leaves the (sum, sum of squares) of every 16-row x 4-channel piece of its output in `fuse_stats` and conv2 folds
  // them per group and normalises its operand on load.  The normalised tensor and the (mean, rstd) pairs are then NOT on the tape: such a forward cannot be back-propagated.
  bool eval_fuse = false, fused_used = false;
  // statistics slots by tensor: every few-row conv that writes a block output registers (output pointer -> slots); the GroupNorm that
  // reads that tensor later -- the next ResBlock's first norm, an AttentionBlock's norm, or, for the output path's concatenations
  // [h | skip], two producers at x and x + c1 -- finds them by address.  Cleared at the start of every forward.
  struct PartReg { const void* p; const float2* slots; int nq; };
  std::vector<PartReg> part_reg;
  const PartReg* find_part(const void* p) const { for (auto it = part_reg.rbegin(); it != part_reg.rend(); ++it) if (it->p == p) return &*it; return nullptr; }
  float2* fuse_alloc(size_t slots) { if (fuse_used + slots > fuse_cap) return nullptr; float2* a = fuse_stats + fuse_used; fuse_used += slots; return a; }
  float2* fuse_stats = nullptr; size_t fuse_cap = 0, fuse_used = 0;     // float2 slots; every slot of a used area is written by its producer
  std::vector<ResTape> rt; std::vector<AttnTape> at;

  const void* W(long off) const { return (const char*)wT + (size_t)off * dtype_size(dtype); }
  const float* P(long off) const { return params + off; }
  float* G(long off) const { return grads + off; }
  void* alloc_act(long rows, long cols) { return arena.alloc((size_t)rows * cols * dtype_size(dtype)); }
  void add_entry(const std::string& name, long off, int ndim, int s0, int s1 = 0, int s2 = 0) {
    Entry e; e.name = name; e.offset = off; e.ndim = ndim; e.shape[0] = s0; e.shape[1] = s1; e.shape[2] = s2;
    e.numel = (long)s0 * (ndim > 1 ? s1 : 1) * (ndim > 2 ? s2 : 1);
    entries.push_back(e);
  }
  int bind(float* p, float* g);
  int sync_weights();
  void release_kblk();
  ~NetBase() { release_kblk(); if (owns_wT && wT) (void)hipFree(wT); }      // (release_kblk also drops the s2ws copies)
};

#define ALLOC_OR_FAIL(var, expr)                                                     \
  do { (var) = (decltype(var))(expr); if (!(var)) EEG_FAIL(EEGLDM_ERR_NOMEM, "workspace allocation failed"); } while (0)

constexpr float GN_EPS = 1e-6f;

int res_forward(NetBase* u, const ResDesc& r, const View& x, int B, int Lin, const View& out);
// extra (optional): a tensor of dx's shape to be added into dx (the UNet skip gradient); *extra_done = 1 when the last GroupNorm kernel took it
int res_backward(NetBase* u, const ResDesc& r, const ResTape& t, const View& dout, const View& dx, float* demb_all,
                 const View* extra = nullptr, int* extra_done = nullptr);
int attn_forward(NetBase* u, const AttnDesc& a, const View& x, int B, int T, const View& out);
int attn_backward(NetBase* u, const AttnDesc& a, const AttnTape& t, const View& dout, const View& dx);
int resample_forward(NetBase* u, const RsDesc& s, const View& x, int B, int Lin, const View& out, RsTape* tape);
int resample_backward(NetBase* u, const RsDesc& s, const RsTape& t, const View& dout, const View& dx);
// accessors for the sampler (sampler.hip); the struct itself is private to unet.hip
eegldm_ctx* unet_ctx(const eegldm_unet* u);
int unet_in_channels(const eegldm_unet* u);
int unet_emb_width(const eegldm_unet* u);
long unet_embed_work_floats(const eegldm_unet* u);
int unet_embed_table(eegldm_unet* u, const int64_t* tsteps_dev, int n, float* table, float* work);
void unet_set_shared_emb(eegldm_unet* u, const float* row);
int unet_out_channels(const eegldm_unet* u);
void sampler_release(const eegldm_unet* u);
eegldm_ctx* aekl_ctx(const eegldm_aekl* a);
int entry_query(const NetBase* u, int i, char* name, int cap, long* offset, long* numel, int* ndim, int shape[3]);
