"""Developer tool: randomised shapes for conv fwd / dgrad / wgrad against torch (bf16-rounded operands, fp32 reference)."""
import math, os, random, sys, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import gpu_util as G
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = G.ctx()
bad = 0
for it in range(n_cases):
    dtype = random.choice([0, 1, 1, 1])
    K = random.choice([1, 3, 3])
    B = random.choice([1, 2, 3, 8, 16, 32])
    L = random.choice([64, 96, 128, 192, 256, 384, 512])
    Cin = random.choice([16, 32, 64, 96, 128, 192, 256, 384, 512])
    Cout = random.choice([16, 32, 64, 96, 128, 192, 256, 384, 512])
    if B * L * max(Cin, Cout) > 64 * 1024 * 1024: continue
    pl = pr = 1 if K == 3 else 0
    g = torch.Generator().manual_seed(it)
    x = torch.randn(B, Cin, L, generator=g); w = torch.randn(Cout, Cin, K, generator=g) / math.sqrt(Cin * K); b = torch.randn(Cout, generator=g)
    r = torch.randn(B, Cout, L, generator=g); e = torch.randn(B, Cout, generator=g)
    if dtype == 1: x, w, r = x.bfloat16().float(), w.bfloat16().float(), r.bfloat16().float()
    x.requires_grad_(True); w.requires_grad_(True)
    use_add = random.random() < 0.5
    y_ref = F.conv1d(F.pad(x, (pl, pr)), w, b)
    if use_add: y_ref = y_ref + e[:, :, None] + r
    dy = torch.randn(y_ref.shape, generator=g)
    if dtype == 1: dy = dy.bfloat16().float()
    y_ref.backward(dy)
    xd, wd, bd = G.nlc(x.detach(), dtype), G.pack_w(w.detach(), dtype), b.to(G.DEV)
    yd = torch.empty(B * L, Cout, device=G.DEV, dtype=G.TDT[dtype])
    ed, rd = e.to(G.DEV), G.nlc(r, dtype)
    G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, K, 1, pl, pr,
                                    G.ptr(ed) if use_add else None, Cout if use_add else 0, G.ptr(rd) if use_add else None, Cout if use_add else 0, dtype))
    dyd = G.nlc(dy, dtype); dxd = torch.empty(B * L, Cin, device=G.DEV, dtype=G.TDT[dtype])
    G.check(G.lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dxd), Cin, B, L, Cin, Cout, K, 1, pl, pr, None, 0, dtype))
    dwd = torch.zeros(K, Cout, Cin, device=G.DEV); dbd = torch.zeros(Cout, device=G.DEV)
    G.check(G.lib.eegldm_conv1d_bwd_weight(c.h, G.ptr(xd), Cin, G.ptr(dyd), Cout, G.ptr(dwd), G.ptr(dbd), B, L, Cin, Cout, K, 1, pl, pr, dtype))
    def rel(a, bref): return float((a - bref).abs().max() / (bref.abs().max() + 1e-9))
    errs = (rel(G.ncl(yd, B, L).float().cpu(), y_ref.detach()), rel(G.ncl(dxd, B, L).float().cpu(), x.grad), rel(G.unpack_w(dwd).cpu(), w.grad), rel(dbd.cpu(), dy.sum((0, 2))))
    tol = 2e-2 if dtype == 1 else 2e-4
    ok = all(v < tol for v in errs)
    if not ok: bad += 1
    print(f"{'ok ' if ok else 'BAD'} dt={dtype} K={K} B={B} L={L} {Cin}->{Cout} add={int(use_add)} errs y {errs[0]:.1e} dx {errs[1]:.1e} dw {errs[2]:.1e} db {errs[3]:.1e}")
print("BAD CASES:", bad)
