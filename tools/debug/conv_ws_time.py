import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
def t(fn, n=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(n): fn()
    return ctx.timer_stop_ms() / n * 1e3
out = []
for B in (256, 1024):
    L, Cin, Cout = 768, 128, 128
    M = B * L
    x = torch.randn(M, Cin, device="cuda").bfloat16(); w = (torch.randn(3, Cout, Cin, device="cuda") / 20).bfloat16()
    y = torch.empty(M, Cout, device="cuda", dtype=torch.bfloat16); res = torch.randn(M, Cout, device="cuda").bfloat16()
    f0 = t(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), Cin, ptr(w), None, ptr(y), Cout, B, L, Cin, Cout, 3, 1, 1, 1, None, 0, None, 0, 1)))
    f1 = t(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), Cin, ptr(w), None, ptr(y), Cout, B, L, Cin, Cout, 3, 1, 1, 1, None, 0, ptr(res), Cout, 1)))
    out.append(f"B={B}: plain {f0:.1f} resid {f1:.1f}")
print("dbg", os.environ.get("EEGLDM_CONV_WS_DBG", "0"), "bpc", os.environ.get("EEGLDM_CONV_WS_BLOCKS_PER_CU", "2"), "|", " | ".join(out))
