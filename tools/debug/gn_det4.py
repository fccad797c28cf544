import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
torch.manual_seed(0)
B = 256
for (L, C) in [(384, 384), (192, 768), (192, 1536), (384, 768), (384, 512)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); ad = torch.randn(R, C, device="cuda").bfloat16()
    ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); y = torch.empty_like(x)
    check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1))
    outs = []
    for _ in range(4):
        dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, ptr(ad), C, 1))
        torch.cuda.synchronize(); outs.append(dx.clone())
    nd = [int((outs[0] != o).sum()) for o in outs[1:]]
    # reference on a few samples
    xs = x[:4 * L].float().reshape(4, L, C).permute(0, 2, 1).requires_grad_(True)
    yr = F.silu(F.group_norm(xs, 32, ga, be, eps=1e-6))
    yr.backward(dy[:4 * L].float().reshape(4, L, C).permute(0, 2, 1))
    ref = xs.grad.permute(0, 2, 1).reshape(4 * L, C) + ad[:4 * L].float()
    err = float((outs[0][:4 * L].float() - ref).norm() / ref.norm())
    print(f"NTH={os.environ.get('EEGLDM_GN_BWD_NTH','default')} L={L} C={C}: differing elements {nd} rel err vs torch {err:.2e}")
