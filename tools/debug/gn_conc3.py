"""Developer tool: anatomy of the wrong one-pass GroupNorm backward output beside the LDS-DMA weight-gradient GEMM (see gn_conc2.py)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check, Context
ctx = eegldm.default_context(0)
ctx2 = Context(0, use_torch_stream=False)
torch.manual_seed(0)
B = 256
Lw, Cw = 192, 512
xw = torch.randn(B * Lw, Cw, device="cuda").bfloat16(); dyw = torch.randn(B * Lw, Cw, device="cuda").bfloat16()
dw = torch.zeros(3 * Cw * Cw, device="cuda"); dbw = torch.zeros(Cw, device="cuda")
L, C = 384, 512
R = B * L
x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); ad = torch.randn(R, C, device="cuda").bfloat16()
ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); y = torch.empty_like(x)
check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1))
def run(noise, reps=1):
    dx = torch.full_like(x, 777.0); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    torch.cuda.synchronize(); ctx2.sync()
    if noise:
        for _ in range(6):
            check(lib.eegldm_conv1d_bwd_weight(ctx2.h, ptr(xw), Cw, ptr(dyw), Cw, ptr(dw), ptr(dbw), B, Lw, Cw, Cw, 3, 1, 1, 1, 1))
    for _ in range(reps):
        check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, ptr(ad), C, 1))
    torch.cuda.synchronize(); ctx2.sync()
    return dx
quiet = run(False).float().reshape(B, L, C)
for t in range(3):
    d = run(True).float().reshape(B, L, C) - quiet
    bad = d != 0
    print(f"run {t}: {int(bad.sum())} elements differ; never-written (777) elements {int((d + quiet == 777.0).sum())}")
    if not bad.any(): continue
    per_bc = bad.sum(1)                                   # [B][C] rows differing per (sample, channel)
    bs = torch.unique(bad.nonzero()[:, 0]).tolist()
    print(f"  bad samples ({len(bs)}): {bs[:24]}")
    b0 = bs[0]
    ch = (per_bc[b0] > 0).nonzero()[:, 0].tolist()
    print(f"  sample {b0}: bad channels {ch[:40]}{'...' if len(ch) > 40 else ''} ({len(ch)}), rows differing per bad channel min {int(per_bc[b0][ch].min())} max {int(per_bc[b0][ch].max())} of {L}")
    c0 = ch[0]
    rows = bad[b0, :, c0].nonzero()[:, 0].tolist()
    print(f"  sample {b0} channel {c0}: differing rows {rows[:16]}... diffs {[round(float(v), 4) for v in d[b0, rows[:8], c0]]} quiet {[round(float(v), 4) for v in quiet[b0, rows[:8], c0]]}")
    # is the difference of the form a + b * xhat per group (wrong group sums)?  fit per channel over rows
    xs = x.float().reshape(B, L, C)[b0, :, c0]; dd = d[b0, :, c0]
    A = torch.stack([torch.ones_like(xs), xs], 1); sol = torch.linalg.lstsq(A, dd.unsqueeze(1)).solution.squeeze()
    resid = dd - A @ sol
    print(f"  fit diff = a + b*x on that channel: a {float(sol[0]):.4e} b {float(sol[1]):.4e}, residual rms {float(resid.pow(2).mean().sqrt()):.3e} vs diff rms {float(dd.pow(2).mean().sqrt()):.3e}")
