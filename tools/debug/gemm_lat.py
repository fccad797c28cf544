import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
def timeit(fn, reps=50):
    for _ in range(5): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop_ms() / reps
z = torch.zeros(1024, device="cuda")
t = timeit(lambda: check(lib.eegldm_fill(ctx.h, ptr(z), 1024, 1.0)))
print(f"fill kernel (1 block): {t*1e3:.2f} us per launch (launch floor)")
B, L, co = 1, 128, 128
for ci in (32, 128, 512, 1024):
    R = B * L
    x = torch.randn(R, ci, device="cuda").bfloat16(); w = (torch.randn(3, co, ci, device="cuda") * 0.05).bfloat16(); b = torch.zeros(co, device="cuda")
    y = torch.empty(R, co, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w), ptr(b), ptr(y), co, B, L, ci, co, 3, 1, 1, 1, None, 0, None, 0, 1)))
    t1 = timeit(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w), ptr(b), ptr(y), co, B, L, ci, co, 1, 1, 0, 0, None, 0, None, 0, 1)))
    print(f"single block conv3 {ci}->{co}: {t*1e3:.2f} us ({ci//32} stages) ; 1x1 (reg-staged, {ci//64} stages): {t1*1e3:.2f} us")
