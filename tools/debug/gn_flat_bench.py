import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B = 256
for (L, C) in [(3072, 32), (1536, 32), (768, 64), (3072, 2), (768, 4)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); y = torch.empty_like(x); ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * 2, device="cuda")
    def t(fn, n=20):
        for _ in range(3): fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(n): fn()
        return ctx.timer_stop_ms() / n
    f = t(lambda: check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 1, 1e-6, 1, 0, None, 0, 1)))
    print(f"L={L} C={C} G=1: {R*C*2/1e6:.0f} MB  fwd {f*1e3:.1f} us")
