"""Developer microbench: conv3 fwd / dgrad / wgrad and 1x1 at the UNet's shapes (B=256), TF/s per shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
dt = 1 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else 0
which = sys.argv[2] if len(sys.argv) > 2 else "all"
B = 256
tdt = torch.bfloat16 if dt == 1 else torch.float32
shapes = [(128, 128, 768), (256, 256, 384), (512, 512, 192), (1024, 512, 192), (768, 256, 384), (384, 128, 768), (256, 128, 768)]
if which == "one":
    shapes = [(512, 512, 192)]
reps = 10
def timeit(fn):
    for _ in range(2): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop_ms() / reps
tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0], "1x1": [0, 0]}
for (ci, co, L) in shapes:
    R = B * L
    x = torch.randn(R, ci, device="cuda").to(tdt); w = (torch.randn(3, co, ci, device="cuda") * 0.05).to(tdt); b = torch.zeros(co, device="cuda")
    y = torch.empty(R, co, device="cuda", dtype=tdt); dx = torch.empty(R, ci, device="cuda", dtype=tdt); dw = torch.zeros(3, co, ci, device="cuda")
    fl = 2.0 * R * ci * co * 3
    t1 = timeit(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w), ptr(b), ptr(y), co, B, L, ci, co, 3, 1, 1, 1, None, 0, None, 0, dt)))
    t2 = timeit(lambda: check(lib.eegldm_conv1d_bwd_data(ctx.h, ptr(y), co, ptr(w), ptr(dx), ci, B, L, ci, co, 3, 1, 1, 1, None, 0, dt)))
    t3 = timeit(lambda: check(lib.eegldm_conv1d_bwd_weight(ctx.h, ptr(x), ci, ptr(y), co, ptr(dw), None, B, L, ci, co, 3, 1, 1, 1, dt)))
    w1 = w[0].contiguous()
    t4 = timeit(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w1), ptr(b), ptr(y), co, B, L, ci, co, 1, 1, 0, 0, None, 0, None, 0, dt)))
    for k, t, f in (("fwd", t1, fl), ("dgrad", t2, fl), ("wgrad", t3, fl), ("1x1", t4, fl / 3)):
        tot[k][0] += f; tot[k][1] += t
    print(f"{ci:5d}->{co:4d} L={L:4d}: fwd {fl/t1/1e9:7.1f} TF ({t1*1e3:7.1f} us)  dgrad {fl/t2/1e9:7.1f} TF  wgrad {fl/t3/1e9:7.1f} TF  1x1 {fl/3/t4/1e9:7.1f} TF ({t4*1e3:6.1f} us)")
print("TOTAL  " + "  ".join(f"{k} {v[0]/v[1]/1e9:7.1f} TF" for k, v in tot.items()))
