import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
L, co, ci = 768, 128, 32
def timeit(fn, reps=20):
    for _ in range(3): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop_ms() / reps
for B in (1, 8, 32, 64, 85, 86, 128, 171, 256, 512):
    R = B * L
    x = torch.randn(R, ci, device="cuda").bfloat16(); w = (torch.randn(3, co, ci, device="cuda") * 0.05).bfloat16(); b = torch.zeros(co, device="cuda")
    y = torch.empty(R, co, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w), ptr(b), ptr(y), co, B, L, ci, co, 3, 1, 1, 1, None, 0, None, 0, 1)))
    print(f"B={B:4d} blocks={R//128:5d} ({R/128/512:.2f} rounds of 512): {t*1e3:7.1f} us   out {R*co*2/1e6:.1f} MB -> {R*co*2/t/1e9:.2f} TB/s written")
