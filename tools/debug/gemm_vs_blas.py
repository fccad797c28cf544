"""Developer microbench: what does the vendor library (torch.mm -> hipBLASLt / rocBLAS) reach on the GEMM shapes of the UNet step?
Same shapes as gemm_bench.py; for the 3-tap convs the library gets the EQUIVALENT plain GEMM ([R, 3ci] x [3ci, co], i.e. an im2col
matrix it would have to be handed: 3x the A bytes of the implicit GEMM).  Yardstick only -- nothing here is on the product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B, reps = 256, 10
shapes = [(128, 128, 768), (256, 256, 384), (512, 512, 192), (1024, 512, 192), (768, 256, 384), (384, 128, 768), (256, 128, 768)]
def t_torch(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
def t_mine(fn):
    for _ in range(2): fn()
    ctx.sync(); ctx.timer_start()
    for _ in range(reps): fn()
    return ctx.timer_stop_ms() / reps
for (ci, co, L) in shapes:
    R = B * L
    x = torch.randn(R, ci, device="cuda").bfloat16(); x3 = torch.randn(R, 3 * ci, device="cuda").bfloat16()
    w = (torch.randn(3, co, ci, device="cuda") * 0.05).bfloat16(); w1 = w[0].contiguous(); w3 = w.permute(1, 0, 2).reshape(co, 3 * ci).contiguous()
    b = torch.zeros(co, device="cuda"); y = torch.empty(R, co, device="cuda", dtype=torch.bfloat16); dw = torch.zeros(3, co, ci, device="cuda")
    dy = torch.randn(R, co, device="cuda").bfloat16()
    f1 = 2.0 * R * ci * co
    m1 = t_mine(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w1), ptr(b), ptr(y), co, B, L, ci, co, 1, 1, 0, 0, None, 0, None, 0, 1)))
    m3 = t_mine(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w), ptr(b), ptr(y), co, B, L, ci, co, 3, 1, 1, 1, None, 0, None, 0, 1)))
    mw = t_mine(lambda: check(lib.eegldm_conv1d_bwd_weight(ctx.h, ptr(x), ci, ptr(dy), co, ptr(dw), None, B, L, ci, co, 3, 1, 1, 1, 1)))
    l1 = t_torch(lambda: torch.mm(x, w1.t(), out=y))
    l3 = t_torch(lambda: torch.mm(x3, w3.t(), out=y))
    dwl = torch.empty(co, 3 * ci, device="cuda", dtype=torch.bfloat16)
    lw = t_torch(lambda: torch.mm(dy.t(), x3, out=dwl))
    print(f"{ci:5d}->{co:4d} L={L:4d} | 1x1 mine {f1/m1/1e9:6.0f} TF ({m1*1e3:6.1f} us) lib {f1/l1/1e9:6.0f} TF ({l1*1e3:6.1f} us) | "
          f"k3 fwd mine {3*f1/m3/1e9:6.0f} TF ({m3*1e3:6.1f} us) lib(im2col'd) {3*f1/l3/1e9:6.0f} TF ({l3*1e3:6.1f} us) | "
          f"k3 wgrad mine {3*f1/mw/1e9:6.0f} TF ({mw*1e3:6.1f} us) lib {3*f1/lw/1e9:6.0f} TF ({lw*1e3:6.1f} us)", flush=True)
# a large square product: what the library reaches when nothing about the shape is awkward
for n in (4096, 8192):
    a = torch.randn(n, n, device="cuda").bfloat16(); bb = torch.randn(n, n, device="cuda").bfloat16(); c = torch.empty(n, n, device="cuda", dtype=torch.bfloat16)
    t = t_torch(lambda: torch.mm(a, bb.t(), out=c))
    print(f"square {n}: lib {2.0*n**3/t/1e9:6.0f} TF")
