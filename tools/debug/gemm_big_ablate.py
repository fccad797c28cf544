"""Ablation of the big-tile conv kernel: time per launch with parts of the K loop compiled out (tools/debug/gemm_big_ablate.sh builds one
library per variant, -DEEG_BIG_DBG bits: 1 no DMA, 2 no fragment reads, 4 no MFMAs, 8 no epilogue).  Results are garbage, timings are not.
Run one variant: EEGLDM_LIB=tools/debug/libeegldm_big<bits>.so python tools/debug/gemm_big_ablate.py"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch
import gpu_util as G
lib = G.lib; c = G.ctx(); dt = G.BF16
for (B, L, Cin, Cout) in [(256, 192, 512, 512), (256, 192, 1024, 512), (256, 384, 256, 256)]:
    xd = torch.randn(B * L, Cin, device=G.DEV).bfloat16(); wd = (torch.randn(3, Cout, Cin, device=G.DEV) / math.sqrt(3 * Cin)).bfloat16()
    bd = torch.randn(Cout, device=G.DEV); yd = torch.empty(B * L, Cout, device=G.DEV, dtype=torch.bfloat16)
    wk = torch.empty_like(wd); G.check(lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(wd), G.ptr(wk), Cout, Cin, dt))
    line = []
    full = os.environ.get("BIG_TIME_FULL") is not None      # with the embedding row and the residual
    ed = torch.randn(B, Cout, device=G.DEV); rd = torch.randn(B * L, Cout, device=G.DEV).bfloat16()
    for bits in (os.environ.get("EEGLDM_LIB", "production"),):
        call = lambda: G.check(lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, 3, 1, 1, 1,
                                                     G.ptr(ed) if full else None, Cout if full else 0, G.ptr(rd) if full else None, Cout if full else 0, dt))
        for _ in range(3): call()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): call()
        torch.cuda.synchronize(); line.append(f"{os.path.basename(bits)}: {(time.perf_counter() - t0) / 20 * 1e6:.1f}")
    print(f"B={B} L={L} Cin={Cin} Cout={Cout}{' full' if full else ''} (phases {3 * Cin // 64}): us per launch by ablation bits  " + "  ".join(line), flush=True)
    G.check(lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd)))
