"""Developer probe: fp32 OUTPUTS (dgamma, dbeta, dW, dbias) of bf16 primitives against torch fp32 on the SAME bf16-rounded inputs.
With fp32 accumulation inside the kernels the relative error must be ~1e-6; anything larger is internal low-precision arithmetic."""
import sys, os, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import torch.nn.functional as F
import gpu_util as G
from param_gen import normal

c = G.ctx(); dt = G.BF16
def r(t): return t.bfloat16().float()
for (B, L, C, Gr, silu) in [(2, 256, 2, 1, 1), (2, 256, 4, 1, 1), (2, 256, 8, 1, 1), (2, 64, 16, 1, 1), (2, 3072, 4, 1, 1), (2, 256, 32, 1, 1), (2, 768, 128, 32, 1)]:
    x = r(torch.from_numpy(normal((B, C, L), seed=1)) * 1.5 + 0.7).requires_grad_(True)
    ga = (1 + 0.1 * torch.from_numpy(normal((C,), seed=2))).requires_grad_(True); be = (0.1 * torch.from_numpy(normal((C,), seed=3))).requires_grad_(True)
    h = F.group_norm(x, Gr, ga, be, eps=1e-6); h = F.silu(h) if silu else h
    dy = r(torch.from_numpy(normal(tuple(h.shape), seed=4)))
    (h * dy).sum().backward()
    xd = G.nlc(x.detach(), dt); gad, bed = ga.detach().to(G.DEV), be.detach().to(G.DEV)
    yd = torch.empty(B * L, C, device=G.DEV, dtype=torch.bfloat16); st = torch.empty(B * Gr * 2, device=G.DEV)
    G.check(G.lib.eegldm_groupnorm_fwd(c.h, G.ptr(xd), C, G.ptr(gad), G.ptr(bed), G.ptr(yd), C, G.ptr(st), B, L, C, Gr, 1e-6, silu, 0, None, 0, dt))
    dyd = G.nlc(dy, dt); dxd = torch.empty(B * L, C, device=G.DEV, dtype=torch.bfloat16); dga = torch.zeros(C, device=G.DEV); dbe = torch.zeros(C, device=G.DEV)
    G.check(G.lib.eegldm_groupnorm_bwd(c.h, G.ptr(xd), C, G.ptr(gad), G.ptr(bed), G.ptr(st), G.ptr(dyd), C, G.ptr(dxd), C, G.ptr(dga), G.ptr(dbe), B, L, C, Gr, silu, 0, None, 0, dt))
    print(f"GN B{B} L{L} C{C} G{Gr}: y {G.rel_l2(G.ncl(yd, B, L), h):.1e} dx {G.rel_l2(G.ncl(dxd, B, L), x.grad):.1e} dgamma {G.rel_l2(dga, ga.grad):.1e} dbeta {G.rel_l2(dbe, be.grad):.1e}")
for (B, L, Ci, Co, K, s, pl, pr) in [(2, 256, 2, 2, 3, 1, 1, 1), (2, 256, 4, 4, 3, 1, 1, 1), (2, 256, 8, 8, 3, 1, 1, 1), (2, 256, 1, 4, 3, 1, 1, 1), (2, 256, 4, 4, 3, 2, 0, 1),
                                  (2, 128, 4, 16, 3, 1, 1, 1), (2, 128, 4, 16, 1, 1, 0, 0), (2, 64, 16, 32, 3, 1, 1, 1), (2, 256, 32, 32, 3, 1, 1, 1), (8, 3072, 4, 4, 3, 1, 1, 1)]:
    x = r(torch.from_numpy(normal((B, Ci, L), seed=1))).requires_grad_(True)
    w = r(torch.from_numpy(normal((Co, Ci, K), seed=2)) / math.sqrt(Ci * K)).requires_grad_(True); b = torch.zeros(Co, requires_grad=True)
    y = F.conv1d(F.pad(x, (pl, pr)), w, b, stride=s); Lo = y.shape[-1]
    dy = r(torch.from_numpy(normal(tuple(y.shape), seed=4))); y.backward(dy)
    xd, wd = G.nlc(x.detach(), dt), G.pack_w(w.detach(), dt); dyd = G.nlc(dy, dt)
    dwd = torch.zeros(K, Co, Ci, device=G.DEV); dbd = torch.zeros(Co, device=G.DEV)
    G.check(G.lib.eegldm_conv1d_bwd_weight(c.h, G.ptr(xd), Ci, G.ptr(dyd), Co, G.ptr(dwd), G.ptr(dbd), B, L, Ci, Co, K, s, pl, pr, dt))
    dxd = torch.empty(B * L, Ci, device=G.DEV, dtype=torch.bfloat16)
    G.check(G.lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Co, G.ptr(wd), G.ptr(dxd), Ci, B, L, Ci, Co, K, s, pl, pr, None, 0, dt))
    print(f"conv B{B} L{L} {Ci}->{Co} k{K} s{s}: dw {G.rel_l2(G.unpack_w(dwd), w.grad):.1e} db {G.rel_l2(dbd, b.grad):.1e} dx {G.rel_l2(G.ncl(dxd, B, L), x.grad):.1e}")
