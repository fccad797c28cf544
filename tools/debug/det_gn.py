import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
def run(B, L, C, G, silu, use_dxr, n=4):
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); dx = torch.empty_like(x); dxr = torch.randn(R, C, device="cuda").bfloat16()
    ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * G * 2, device="cuda"); y = torch.empty_like(x)
    check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, G, 1e-6, silu, 0, None, 0, 1))
    outs = []
    for _ in range(n):
        dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda"); dx.zero_()
        check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, G, silu, 0, ptr(dxr) if use_dxr else None, C, 1))
        ctx.sync(); outs.append((dx.clone(), dg.clone(), db.clone()))
    bad = sum(int((o[0].view(torch.int16) != outs[0][0].view(torch.int16)).sum()) for o in outs[1:])
    rows = set()
    for o in outs[1:]:
        idx = (o[0].view(torch.int16) != outs[0][0].view(torch.int16)).nonzero()
        for r, c in idx[:2000].tolist(): rows.add((r // L, c // (C // G)))
    dgd = max(float((o[1] - outs[0][1]).abs().max() / (outs[0][1].abs().max() + 1e-30)) for o in outs[1:])
    print(f"B={B} L={L} C={C} G={G} silu={silu} dxr={use_dxr}: dx mismatches {bad} in {len(rows)} (sample,group) pairs; dgamma rel diff {dgd:.1e}")
for a in [(256, 192, 512, 32, 1, 1), (256, 192, 512, 32, 0, 1), (256, 192, 512, 32, 1, 0), (8, 192, 512, 32, 1, 1), (256, 384, 256, 32, 1, 1)]:
    run(*a)
