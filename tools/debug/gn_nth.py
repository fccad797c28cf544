"""One-pass GroupNorm kernels at B = 256 for the env-selected block size (EEGLDM_GN_FWD_NTH / EEGLDM_GN_BWD_NTH / EEGLDM_GN_NO_XCD)."""
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
B = 256
out = []
for (L, C) in [(768, 128), (384, 256), (192, 512), (768, 256), (384, 512), (192, 1024)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); y = torch.empty_like(x); dy = torch.randn(R, C, device="cuda").bfloat16(); dx = torch.empty_like(x)
    ad = torch.randn(R, C, device="cuda").bfloat16()
    ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    f = t(lambda: check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1)))
    b2 = t(lambda: check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, ptr(ad), C, 1)))
    out.append(f"{L}x{C}: fwd {f*1e3:5.1f} bwd+add {b2*1e3:5.1f}")
    # cheap checksum so that variants can be compared for equality of results
    out[-1] += f" [y {float(y.float().abs().sum()):.6e} dx {float(dx.float().abs().sum()):.6e}]"
print(os.environ.get("EEGLDM_GN_FWD_NTH", "1024"), os.environ.get("EEGLDM_GN_BWD_NTH", "1024"), "noxcd" if os.environ.get("EEGLDM_GN_NO_XCD") else "xcd", "|", " | ".join(out))
