import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B, L, co = 256, 768, 128
def timeit(fn, reps=10):
    for _ in range(2): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop_ms() / reps
for ci in (32, 64, 128, 256, 512):
    R = B * L
    x = torch.randn(R, ci, device="cuda").bfloat16(); w = (torch.randn(3, co, ci, device="cuda") * 0.05).bfloat16(); b = torch.zeros(co, device="cuda")
    y = torch.empty(R, co, device="cuda", dtype=torch.bfloat16)
    t = timeit(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w), ptr(b), ptr(y), co, B, L, ci, co, 3, 1, 1, 1, None, 0, None, 0, 1)))
    print(f"conv3 {ci}->{co} @L=768 B=256: {t*1e3:.1f} us, stages {ci//32}, blocks {R//128}, {2.0*R*ci*co*3/t/1e9:.0f} TF, in+out {(R*ci+R*co)*2/1e6:.0f} MB -> {(R*ci+R*co)*2/t/1e9:.2f} TB/s")
