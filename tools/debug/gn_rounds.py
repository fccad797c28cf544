"""GroupNorm one-pass kernels: true-traffic bandwidth against batch size (1 round of blocks vs several) and against a plain copy."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
def t(fn, n=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(n): fn()
    return ctx.timer_stop_ms() / n
for (L, C) in [(768, 128), (384, 256), (192, 512)]:
    for B in (64, 128, 256, 512, 1024):
        R = B * L
        x = torch.randn(R, C, device="cuda").bfloat16(); y = torch.empty_like(x); dy = torch.randn(R, C, device="cuda").bfloat16(); dx = torch.empty_like(x)
        ad = torch.randn(R, C, device="cuda").bfloat16()
        ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        nb = R * C * 2
        cp = t(lambda: y.copy_(x))
        f = t(lambda: check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1)))
        b = t(lambda: check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, None, 0, 1)))
        b2 = t(lambda: check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, ptr(ad), C, 1)))
        print(f"L={L} C={C} B={B}: {nb/1e6:.0f} MB | copy {cp*1e3:.1f} us {2*nb/cp/1e9:.2f} TB/s | fwd {f*1e3:.1f} us {2*nb/f/1e9:.2f} TB/s | bwd {b*1e3:.1f} us {3*nb/b/1e9:.2f} TB/s | bwd+addend {b2*1e3:.1f} us {4*nb/b2/1e9:.2f} TB/s", flush=True)
