"""Developer tool: randomised GroupNorm(+SiLU, + resampling) forward/backward against torch autograd."""
import os, random, sys, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gpu_util as G
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = G.ctx(); bad = 0
for it in range(n_cases):
    dtype = random.choice([0, 1, 1])
    Gn = random.choice([32, 32, 1])
    Cc = random.choice([32, 64, 96, 128, 256, 384, 512, 768]) if Gn == 32 else random.choice([4, 8, 32, 64])
    B = random.choice([1, 2, 3, 8, 64, 300])
    L = random.choice([32, 64, 96, 192, 384, 768])
    if B * L * Cc > 40 * 1024 * 1024: continue
    silu = random.choice([0, 1]); rs = random.choice([0, 0, 1, 2]); with_dxr = random.random() < 0.5
    g = torch.Generator().manual_seed(1000 + it)
    x = torch.randn(B, Cc, L, generator=g) * 1.5 + 0.3; ga = torch.randn(Cc, generator=g); be = torch.randn(Cc, generator=g)
    if dtype == 1: x = x.bfloat16().float()
    x.requires_grad_(True); ga.requires_grad_(True); be.requires_grad_(True)
    y = F.group_norm(x, Gn, ga, be, eps=1e-6)
    if silu: y = F.silu(y)
    xr = x
    if rs == 1: y = F.avg_pool1d(y, 2, 2); xr = F.avg_pool1d(x, 2, 2)
    elif rs == 2: y = F.interpolate(y, scale_factor=2, mode="nearest"); xr = F.interpolate(x, scale_factor=2, mode="nearest")
    Lo = y.shape[-1]
    dy = torch.randn(y.shape, generator=g); dxr_t = torch.randn(y.shape, generator=g)
    if dtype == 1: dy = dy.bfloat16().float(); dxr_t = dxr_t.bfloat16().float()
    loss = (y * dy).sum() + ((xr * dxr_t).sum() if with_dxr else 0.0)
    loss.backward()
    xd = G.nlc(x.detach(), dtype); yd = torch.empty(B * Lo, Cc, device=G.DEV, dtype=G.TDT[dtype]); xrd = torch.empty_like(yd)
    st = torch.empty(B * Gn * 2, device=G.DEV); gad, bed = ga.detach().to(G.DEV), be.detach().to(G.DEV)
    G.check(G.lib.eegldm_groupnorm_fwd(c.h, G.ptr(xd), Cc, G.ptr(gad), G.ptr(bed), G.ptr(yd), Cc, G.ptr(st), B, L, Cc, Gn, 1e-6, silu, rs,
                                       G.ptr(xrd) if rs else None, Cc, dtype))
    dyd = G.nlc(dy, dtype); dxrd = G.nlc(dxr_t, dtype); dxd = torch.empty(B * L, Cc, device=G.DEV, dtype=G.TDT[dtype])
    dg = torch.zeros(Cc, device=G.DEV); db = torch.zeros(Cc, device=G.DEV)
    G.check(G.lib.eegldm_groupnorm_bwd(c.h, G.ptr(xd), Cc, G.ptr(gad), G.ptr(bed), G.ptr(st), G.ptr(dyd), Cc, G.ptr(dxd), Cc, G.ptr(dg), G.ptr(db),
                                       B, L, Cc, Gn, silu, rs, G.ptr(dxrd) if with_dxr else None, Cc, dtype))
    def rel(a, bref): return float((a - bref).abs().max() / (bref.abs().max() + 1e-9))
    errs = (rel(G.ncl(yd, B, Lo).float().cpu(), y.detach()), rel(G.ncl(dxd, B, L).float().cpu(), x.grad), rel(dg.cpu(), ga.grad), rel(db.cpu(), be.grad))
    tol = 3e-2 if dtype == 1 else 5e-4
    ok = all(v < tol for v in errs)
    bad += 0 if ok else 1
    print(f"{'ok ' if ok else 'BAD'} dt={dtype} B={B} L={L} C={Cc} G={Gn} silu={silu} rs={rs} dxr={int(with_dxr)} errs y {errs[0]:.1e} dx {errs[1]:.1e} dg {errs[2]:.1e} db {errs[3]:.1e}")
print("BAD CASES:", bad)
