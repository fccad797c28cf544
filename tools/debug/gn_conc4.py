"""Developer tool: which neighbour kernel disturbs which narrow-block GroupNorm kernel?  victim x noise matrix (see gn_conc2.py).
argv[1] = victim: bwd | fwd;  EEGLDM_GN_BWD_NTH / EEGLDM_GN_FWD_NTH pick the block size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check, Context
victim = sys.argv[1] if len(sys.argv) > 1 else "bwd"
opts = sys.argv[2:]          # nosilu | nodg | noadd
ctx = eegldm.default_context(0)
ctx2 = Context(0, use_torch_stream=False)
torch.manual_seed(0)
B = 256
Lw, Cw = 192, 512
xw = torch.randn(B * Lw, Cw, device="cuda").bfloat16(); dyw = torch.randn(B * Lw, Cw, device="cuda").bfloat16(); yw = torch.empty_like(xw)
ww = (torch.randn(3 * Cw * Cw, device="cuda") * 0.02).bfloat16(); w1 = (torch.randn(Cw * Cw, device="cuda") * 0.02).bfloat16()
dw = torch.zeros(3 * Cw * Cw, device="cuda"); dbw = torch.zeros(Cw, device="cuda")
def noise_wgrad(): check(lib.eegldm_conv1d_bwd_weight(ctx2.h, ptr(xw), Cw, ptr(dyw), Cw, ptr(dw), ptr(dbw), B, Lw, Cw, Cw, 3, 1, 1, 1, 1))
def noise_fwd(): check(lib.eegldm_conv1d_fwd(ctx2.h, ptr(xw), Cw, ptr(ww), None, ptr(yw), Cw, B, Lw, Cw, Cw, 3, 1, 1, 1, None, 0, None, 0, 1))
def noise_dgrad(): check(lib.eegldm_conv1d_bwd_data(ctx2.h, ptr(dyw), Cw, ptr(ww), ptr(yw), Cw, B, Lw, Cw, Cw, 3, 1, 1, 1, None, 0, 1))
def noise_lin(): check(lib.eegldm_linear_fwd(ctx2.h, ptr(xw), Cw, ptr(w1), None, ptr(yw), Cw, B * Lw, Cw, Cw, 1, 0))
L, C = 384, 512
R = B * L
x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); ad = torch.randn(R, C, device="cuda").bfloat16()
ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); y = torch.empty_like(x)
check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1))
torch.cuda.synchronize()
def run(noise):
    out = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); st2 = torch.empty_like(st)
    torch.cuda.synchronize(); ctx2.sync()
    if noise:
        for _ in range(10): noise()
    for _ in range(6):
        if victim == "bwd":
            check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(out), C, None if "nodg" in opts else ptr(dg), None if "nodg" in opts else ptr(db),
                                           B, L, C, 32, 0 if "nosilu" in opts else 1, 0, None if "noadd" in opts else ptr(ad), C, 1))
        else:
            check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(out), C, ptr(st2), B, L, C, 32, 1e-6, 1, 0, None, 0, 1))
    torch.cuda.synchronize(); ctx2.sync()
    return out
quiet = run(None)
for name, fn in [("wgrad3", noise_wgrad), ("conv3_fwd", noise_fwd), ("conv3_dgrad", noise_dgrad), ("linear_1tap", noise_lin)]:
    nd = [int((quiet != run(fn)).sum()) for _ in range(3)]
    print(f"[{' '.join(opts)}] victim gn_{victim} (BWD_NTH={os.environ.get('EEGLDM_GN_BWD_NTH', '1024')} FWD_NTH={os.environ.get('EEGLDM_GN_FWD_NTH', '1024')}) beside {name}: differing elements {nd}", flush=True)
