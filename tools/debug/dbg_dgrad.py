import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import gpu_util as G
torch.set_printoptions(linewidth=200, precision=2, sci_mode=False)
c = G.ctx()
B, L, Cin, Cout, K = 2, 64, 32, 32, 3
for dtype in (0, 1):
    w = torch.zeros(Cout, Cin, K)
    for co in range(Cout):
        for ci in range(Cin):
            w[co, ci, 1] = 1.0 if co == ci else 0.0
    w[5, 7, 1] = 2.0   # dx[ci=7] += 2*dy[co=5]
    dy = torch.zeros(B, Cout, L)
    for co in range(Cout):
        dy[:, co, :] = co + 1
    dy[1] *= 10
    wd = G.pack_w(w, dtype); dyd = G.nlc(dy, dtype)
    dxd = torch.full((B * L, Cin), -7.0, device=G.DEV, dtype=G.TDT[dtype])
    G.check(G.lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dxd), Cin, B, L, Cin, Cout, K, 1, 1, 1, None, 0, dtype))
    torch.cuda.synchronize()
    out = dxd.float().cpu()
    print("dtype", dtype, "rows 0,1,63,64,65,127:")
    for r in (0, 1, 63, 64, 65, 127):
        print(r, out[r].tolist())
