import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B = 256
for (L, C) in [(768, 128), (384, 256), (192, 512), (768, 256), (768, 384), (384, 512), (384, 768), (192, 1024), (192, 1536)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); y = torch.empty_like(x); dy = torch.randn(R, C, device="cuda").bfloat16(); dx = torch.empty_like(x)
    ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    tot = torch.zeros(C, device="cuda"); ps = torch.zeros(B, C, device="cuda")
    def t(fn, n=10):
        for _ in range(2): fn()
        ctx.sync(); ctx.timer_start()
        for _ in range(n): fn()
        return ctx.timer_stop_ms() / n
    nb = R * C * 2
    f = t(lambda: check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1)))
    b = t(lambda: check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, None, 0, 1)))
    print(f"L={L} C={C}: tensor {nb/1e6:.0f} MB  fwd {f*1e3:.1f} us ({3*nb/f/1e9:.2f} TB/s for 2R+1W)  bwd {b*1e3:.1f} us ({5*nb/b/1e9:.2f} TB/s for 4R+1W)")
