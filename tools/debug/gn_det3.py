import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
torch.manual_seed(0)
B = 256
for (L, C, rs, silu) in [(384, 256, 1, 1), (192, 512, 2, 1), (384, 256, 2, 1), (192, 512, 0, 0), (384, 256, 0, 0), (192, 512, 1, 1)]:
    R = B * L
    Ld = L // 2 if rs == 1 else (2 * L if rs == 2 else L)
    x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(B * Ld, C, device="cuda").bfloat16(); ad = torch.randn(B * Ld, C, device="cuda").bfloat16()
    ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); y = torch.empty(B * Ld, C, device="cuda", dtype=torch.bfloat16)
    check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, silu, rs, None, 0, 1))
    outs = []
    for _ in range(4):
        dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, silu, rs, ptr(ad), C, 1))
        torch.cuda.synchronize(); outs.append(dx.clone())
    nd = [int((outs[0] != o).sum()) for o in outs[1:]]
    print(f"NTH={os.environ.get('EEGLDM_GN_BWD_NTH','default')} L={L} C={C} resample={rs} silu={silu}: differing elements {nd} nan {int(torch.isnan(outs[0].float()).sum())}")
