// Whole-network kernels for THIN AutoencoderKL configurations (every layer <= 4 channels, e.g. num_channels [2,2,4] of
// config/config_aekl_eeg_2_2_4_spec.yaml, BASELINE configs[1]): see aekl_thin.hip.
#pragma once
#include <vector>

#include "common.h"

// forward micro-ops, executed in order by ONE workgroup per window with every tensor resident in LDS
enum { TF_LOAD = 0,     // global NCL input (cin x Lin) -> buffer dst
       TF_CONV,         // dst = conv(src) [+ buffer add]          (k = 1 / 3, stride 1 / 2, left pad pad_l, zero fill)
       TF_GN,           // dst = GroupNorm(G = 1)(src) [SiLU]; statistics -> stat slot
       TF_UPS,          // dst = nearest x2 (src)
       TF_SAVE,         // buffer src -> tape slot `save` (kept for the backward pass)
       TF_HEADS,        // src = h (lat x L): mu / log-variance heads, clamp, sigma, z = mu + eps * sigma -> dst; KL partial
       TF_STORE };      // buffer src -> global NCL output
// backward micro-ops
enum { TB_LOADDY = 0,   // global NCL gradient -> buffer dst
       TB_LOADT,        // tape slot `save` -> buffer dst
       TB_RECOMP,       // dst = GroupNorm apply [SiLU] of src with the SAVED statistics (recomputes a conv's input activation)
       TB_UPS,          // dst = nearest x2 (src)                   (recomputes the input of an upsample conv)
       TB_CONV,         // dY in src, conv input in act: dW / db accumulate; then act <- dX (overwrites the activation)
       TB_GN,           // x in act, dY in src: dgamma / dbeta accumulate; dst = dX [+ buffer add]
       TB_UPSBWD,       // dst[c][l] = src[c][2l] + src[c][2l+1]
       TB_COPY,         // dst = src
       TB_HEADS,        // dz in src: gradients of the two heads and of the KL term; dst = dh
       TB_STOREDX };    // buffer src -> global NCL gradient of the input

struct ThinOp {
  int kind;
  int cin, cout, Lin, Lout, k, stride, pad_l;
  int w, b;                 // conv parameter offsets (packed [K][Cout][Cin] weights), b < 0: no bias
  int src, dst, add, act;   // LDS buffer ids (-1 = none)
  int gw, gb, silu, stat;   // GroupNorm parameter offsets, activation flag, statistics slot
  int save;                 // tape slot
  int w2, b2;               // TF_HEADS / TB_HEADS: the log-variance head (w, b = the mu head)
  int need_dx;              // TB_CONV: 0 = the input gradient is not needed (first layer)
};

struct ThinProgram {
  std::vector<ThinOp> fwd, bwd;
  std::vector<int> tape_off;      // float offset of every tape slot inside a window's tape
  int tape_stride = 0;            // floats per window
  int nstat = 0;                  // statistics slots per window
  int maxt = 0;                   // largest tensor (floats): LDS buffer size
  int lat = 0, Ll = 0;
  int nparams = 0;                // flat parameter count (copied to LDS by the kernels)
  ThinOp* d_fwd = nullptr; ThinOp* d_bwd = nullptr; int* d_tape_off = nullptr;   // device copies (owned)
  float* tape = nullptr; float* stats = nullptr;                                 // per-call workspace (arena)
};

constexpr int THIN_MAXC = 4;           // channels per layer the kernels are written for
constexpr int THIN_NBUF = 4;           // LDS tensors
constexpr int THIN_MAX_FLOATS = 9216;  // per LDS tensor: 4 x 36 KB = 144 KB

int thin_upload(ThinProgram* p);
void thin_free(ThinProgram* p);
int thin_forward(eegldm_ctx* ctx, const ThinProgram& p, const float* params, const float* x, const float* eps, float* recon, float* z_mu,
                 float* z_sigma, float* kl, int B);
int thin_backward(eegldm_ctx* ctx, const ThinProgram& p, const float* params, float* grads, const float* d_recon, const float* eps,
                  float klw_over_B, float* dx, int B, const float* dmu_ext = nullptr, const float* dsg_ext = nullptr);
