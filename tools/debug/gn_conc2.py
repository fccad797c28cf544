"""Developer tool: one-pass GroupNorm backward (main context) beside THIS library's weight-gradient GEMM running on a second context
with its own stream.  Compares every noisy run bitwise with a quiet run (EEGLDM_GN_BWD_NTH = block size under test,
EEGLDM_WGRAD_NO_DMA=1 = register-staged weight gradient)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check, Context
ctx = eegldm.default_context(0)
ctx2 = Context(0, use_torch_stream=False)
torch.manual_seed(0)
B = 256
Lw, Cw = 192, 512
xw = torch.randn(B * Lw, Cw, device="cuda").bfloat16(); dyw = torch.randn(B * Lw, Cw, device="cuda").bfloat16()
dw = torch.zeros(3 * Cw * Cw, device="cuda"); dbw = torch.zeros(Cw, device="cuda")
for (L, C) in [(192, 512), (384, 256), (384, 512), (768, 128)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); ad = torch.randn(R, C, device="cuda").bfloat16()
    ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); y = torch.empty_like(x)
    check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1))
    def run(noise):
        dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        torch.cuda.synchronize(); ctx2.sync()
        if noise:
            for _ in range(12):
                check(lib.eegldm_conv1d_bwd_weight(ctx2.h, ptr(xw), Cw, ptr(dyw), Cw, ptr(dw), ptr(dbw), B, Lw, Cw, Cw, 3, 1, 1, 1, 1))
        for _ in range(8):
            check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, ptr(ad), C, 1))
        torch.cuda.synchronize(); ctx2.sync()
        return dx, dg
    quiet, qg = run(False)
    res = [run(True) for _ in range(4)]
    nd = [int((quiet != r[0]).sum()) for r in res]
    print(f"NTH={os.environ.get('EEGLDM_GN_BWD_NTH', 'default')} NO_DMA={os.environ.get('EEGLDM_WGRAD_NO_DMA', '0')} L={L} C={C}: dx elements differing from the quiet run {nd}; "
          f"dgamma rel {[float((r[1] - qg).norm() / qg.norm()) for r in res][:2]}", flush=True)
