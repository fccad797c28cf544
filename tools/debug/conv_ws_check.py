"""conv_ws.hip: correctness against a torch fp32 reference on the device and timing, for forward (bias + rowvec + residual) and the
data gradient.  Run twice (EEGLDM_NO_CONV_WS=1 for the general kernel) to compare."""
import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
def t(fn, n=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(n): fn()
    return ctx.timer_stop_ms() / n * 1e3
tag = "general" if os.environ.get("EEGLDM_NO_CONV_WS") else "ws"
torch.manual_seed(0)
for (B, L, Cin, Cout) in [(256, 768, 128, 128), (256, 768, 128, 256), (256, 768, 256, 128), (256, 768, 384, 128), (256, 384, 128, 256), (3, 64, 128, 128)]:
    M = B * L
    x = torch.randn(M, Cin, device="cuda").bfloat16()
    w = (torch.randn(3, Cout, Cin, device="cuda") / (3 * Cin) ** 0.5).bfloat16()          # packed [tap][Cout][Cin]
    bias = torch.randn(Cout, device="cuda"); emb = torch.randn(B, Cout, device="cuda"); res = torch.randn(M, Cout, device="cuda").bfloat16()
    y = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16)
    f = lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), Cin, ptr(w), ptr(bias), ptr(y), Cout, B, L, Cin, Cout, 3, 1, 1, 1, ptr(emb), Cout, ptr(res), Cout, 1))
    f(); torch.cuda.synchronize()
    xr = x.float().reshape(B, L, Cin).permute(0, 2, 1); wr = w.float().permute(1, 2, 0)
    ref = F.conv1d(xr, wr, bias, padding=1) + emb[:, :, None] + res.float().reshape(B, L, Cout).permute(0, 2, 1)
    got = y.float().reshape(B, L, Cout).permute(0, 2, 1)
    ef = float((got - ref).abs().max()); rf = float((got - ref).norm() / ref.norm())
    tf = t(f)
    # data gradient: dX = conv_transpose
    dy = torch.randn(M, Cout, device="cuda").bfloat16(); dx = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16); rs2 = torch.randn(M, Cin, device="cuda").bfloat16()
    g = lambda: check(lib.eegldm_conv1d_bwd_data(ctx.h, ptr(dy), Cout, ptr(w), ptr(dx), Cin, B, L, Cin, Cout, 3, 1, 1, 1, ptr(rs2), Cin, 1))
    g(); torch.cuda.synchronize()
    dyr = dy.float().reshape(B, L, Cout).permute(0, 2, 1)
    refd = F.conv_transpose1d(dyr, wr, padding=1) + rs2.float().reshape(B, L, Cin).permute(0, 2, 1)
    gotd = dx.float().reshape(B, L, Cin).permute(0, 2, 1)
    ed = float((gotd - refd).norm() / refd.norm())
    td = t(g)
    print(f"[{tag}] B={B} L={L} {Cin}->{Cout}: fwd {tf:6.1f} us rel {rf:.2e} max {ef:.3f} | dgrad {td:6.1f} us rel {ed:.2e}", flush=True)
