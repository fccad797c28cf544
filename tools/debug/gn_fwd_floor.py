"""Developer microbench: what bounds the one-pass GroupNorm forward on a 50 MB tensor from HBM?  SiLU on / off against a device copy of the same
bytes, rotating over buffer sets larger than the 256 MB Infinity Cache."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B, NSET = 256, 8
for (L, C) in [(768, 128), (384, 256), (192, 512)]:
    R = B * L
    xs = [torch.randn(R, C, device="cuda").bfloat16() for _ in range(NSET)]; ys = [torch.empty_like(xs[0]) for _ in range(NSET)]
    ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda")
    def t(fn, n=24):
        for i in range(NSET): fn(i)
        ctx.sync(); ctx.timer_start()
        for i in range(n): fn(i % NSET)
        return ctx.timer_stop_ms() / n * 1e3
    nb = R * C * 2
    f1 = t(lambda i: check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(xs[i]), C, ptr(ga), ptr(be), ptr(ys[i]), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1)))
    f0 = t(lambda i: check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(xs[i]), C, ptr(ga), ptr(be), ptr(ys[i]), C, ptr(st), B, L, C, 32, 1e-6, 0, 0, None, 0, 1)))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for i in range(NSET): ys[i].copy_(xs[i])
        s.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record(s)
        for i in range(24): ys[i % NSET].copy_(xs[i % NSET])
        e1.record(s); s.synchronize()
    cp = e0.elapsed_time(e1) / 24 * 1e3
    print(f"L={L} C={C}: {nb/1e6:.0f} MB in + out | GroupNorm+SiLU {f1:.1f} us ({2*nb/f1/1e6:.2f} TB/s)  GroupNorm only {f0:.1f} us ({2*nb/f0/1e6:.2f} TB/s)  torch copy {cp:.1f} us ({2*nb/cp/1e6:.2f} TB/s)")
