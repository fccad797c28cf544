import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
torch.manual_seed(0)
B = 256
for (L, C) in [(384, 256), (192, 512), (192, 1024), (384, 128)]:
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
    md = [float((outs[0].float() - o.float()).abs().max()) for o in outs[1:]]
    print(f"NTH={os.environ.get('EEGLDM_GN_BWD_NTH','default')} L={L} C={C}: differing elements {nd} max abs diff {md}")
