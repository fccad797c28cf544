"""Correctness + timing of the 192 x 256 big-tile conv kernel (gemm_big.hip) against the 128 x 128 kernel of gemm.hip and torch's fp32 conv.
   python tools/debug/gemm_big_check.py [check|time]"""
import math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import torch, torch.nn.functional as F
import gpu_util as G
from param_gen import normal
lib = G.lib
mode = sys.argv[1] if len(sys.argv) > 1 else "check"
c = G.ctx(); dt = G.BF16


def setenv(**kv):
    for k, v in kv.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = str(v)
    lib.eegldm_debug_reload_env()


def run_fwd(B, L, Cin, Cout, rv, rs, kblk, x, w, b, e, r, K=3):
    xd, wd, bd = G.nlc(x, dt), G.pack_w(w, dt), b.to(G.DEV)
    ed = e.to(G.DEV) if rv else None; rd = G.nlc(r, dt) if rs else None
    yd = torch.full((B * L, Cout), float("nan"), device=G.DEV, dtype=torch.bfloat16)
    wk = None
    if kblk and K == 3:
        wk = torch.empty_like(wd); G.check(lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(wd), G.ptr(wk), Cout, Cin, dt))
    G.check(lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, K, 1, K // 2, K // 2,
                                  G.ptr(ed) if rv else None, Cout if rv else 0, G.ptr(rd) if rs else None, Cout if rs else 0, dt))
    torch.cuda.synchronize()
    if kblk and K == 3: G.check(lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd)))
    return G.ncl(yd, B, L).float().cpu()


def run_dgrad(B, L, Cin, Cout, rs, packed, dy, w, r, K=3):
    dyd, wd = G.nlc(dy, dt), G.pack_w(w, dt)
    rd = G.nlc(r, dt) if rs else None
    dxd = torch.full((B * L, Cin), float("nan"), device=G.DEV, dtype=torch.bfloat16)
    wt = None
    if packed:
        wt = torch.empty_like(wd); G.check(lib.eegldm_conv1d_pack_dgrad_k(c.h, G.ptr(wd), G.ptr(wt), Cout, Cin, K, dt))
    G.check(lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dxd), Cin, B, L, Cin, Cout, K, 1, K // 2, K // 2, G.ptr(rd) if rs else None, Cin if rs else 0, dt))
    torch.cuda.synchronize()
    if packed: G.check(lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd)))
    return G.ncl(dxd, B, L).float().cpu()


if mode == "check":
    setenv(EEGLDM_GEMM_BIG_MIN_TILES=1)
    CASES = [(4, 192, 512, 512, 1, 0), (4, 192, 1024, 512, 0, 1), (2, 384, 768, 256, 1, 1), (3, 192, 64, 256, 0, 0), (2, 768, 256, 256, 1, 0), (1, 192, 128, 512, 0, 1)]
    for ci, (B, L, Cin, Cout, rv, rs) in enumerate(CASES):
        x = torch.from_numpy(normal((B, Cin, L), seed=10 + ci)).bfloat16().float()
        w = (torch.from_numpy(normal((Cout, Cin, 3), seed=40 + ci)) / math.sqrt(Cin * 3)).bfloat16().float()
        b = torch.from_numpy(normal((Cout,), seed=70 + ci))
        e = torch.from_numpy(normal((B, Cout), seed=100 + ci)) if rv else None
        r = torch.from_numpy(normal((B, Cout, L), seed=130 + ci)).bfloat16().float() if rs else None
        ref = F.conv1d(x, w, b, padding=1)
        if rv: ref = ref + e[:, :, None]
        if rs: ref = ref + r
        for kblk in (0, 1):
            setenv(EEGLDM_NO_GEMM_BIG=None)
            y = run_fwd(B, L, Cin, Cout, rv, rs, kblk, x, w, b, e, r)
            setenv(EEGLDM_NO_GEMM_BIG=1)
            y0 = run_fwd(B, L, Cin, Cout, rv, rs, kblk, x, w, b, e, r)
            err, err0 = float((y - ref).abs().max()), float((y0 - ref).abs().max())
            d = float((y - y0).norm() / y0.norm())
            print(f"fwd case {ci} {CASES[ci]} kblk={kblk}: big max|err| {err:.3e}  old {err0:.3e}  big-vs-old rel {d:.2e}  finite {bool(torch.isfinite(y).all())}", flush=True)
            G.assert_close(y, ref, **G.TOL[dt], name=f"fwd {ci}")
        # dgrad: dx = conv_transpose(dy, w)
        dy = torch.from_numpy(normal((B, Cout, L), seed=160 + ci)).bfloat16().float()
        rr = torch.from_numpy(normal((B, Cin, L), seed=190 + ci)).bfloat16().float() if rs else None
        refd = F.conv_transpose1d(dy, w, padding=1)
        if rs: refd = refd + rr
        for packed in (0, 1):
            setenv(EEGLDM_NO_GEMM_BIG=None)
            dx = run_dgrad(B, L, Cin, Cout, rs, packed, dy, w, rr)
            err = float((dx - refd).abs().max())
            print(f"dgrad case {ci} packed={packed}: max|err| {err:.3e} (ref scale {float(refd.abs().max()):.2f}) finite {bool(torch.isfinite(dx).all())}", flush=True)
            G.assert_close(dx, refd, **G.GTOL[dt], name=f"dgrad {ci}")
    for ci, (B, L, Cin, Cout, rs) in enumerate([(4, 192, 512, 1536, 0), (4, 192, 512, 512, 1), (2, 384, 768, 256, 0), (3, 192, 1024, 512, 1), (1, 192, 64, 256, 0)]):
        x = torch.from_numpy(normal((B, Cin, L), seed=310 + ci)).bfloat16().float()
        w = (torch.from_numpy(normal((Cout, Cin, 1), seed=340 + ci)) / math.sqrt(Cin)).bfloat16().float()
        b = torch.from_numpy(normal((Cout,), seed=370 + ci))
        r = torch.from_numpy(normal((B, Cout, L), seed=430 + ci)).bfloat16().float() if rs else None
        ref = F.conv1d(x, w, b)
        if rs: ref = ref + r
        setenv(EEGLDM_NO_GEMM_BIG=None); y = run_fwd(B, L, Cin, Cout, 0, rs, 0, x, w, b, None, r, K=1)
        setenv(EEGLDM_NO_GEMM_BIG=1); y0 = run_fwd(B, L, Cin, Cout, 0, rs, 0, x, w, b, None, r, K=1)
        print(f"1x1 fwd case {ci}: big max|err| {float((y - ref).abs().max()):.3e} old {float((y0 - ref).abs().max()):.3e} big-vs-old rel {float((y - y0).norm() / y0.norm()):.2e}", flush=True)
        G.assert_close(y, ref, **G.TOL[dt], name=f"1x1 fwd {ci}")
        dy = torch.from_numpy(normal((B, Cout, L), seed=460 + ci)).bfloat16().float()
        refd = F.conv_transpose1d(dy, w)
        for packed in (0, 1):
            setenv(EEGLDM_NO_GEMM_BIG=None)
            dx = run_dgrad(B, L, Cin, Cout, 0, packed, dy, w, None, K=1)
            print(f"1x1 dgrad case {ci} packed={packed}: max|err| {float((dx - refd).abs().max()):.3e} (ref scale {float(refd.abs().max()):.2f})", flush=True)
            G.assert_close(dx, refd, **G.GTOL[dt], name=f"1x1 dgrad {ci}")
    print("check ok")
else:
    if os.environ.get("BIG_SHAPES"):      # "B,L,Cin,Cout;..." : only these 3-tap shapes
        SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["BIG_SHAPES"].split(";")]
    else: SHAPES = [(256, 192, 512, 512), (256, 192, 1024, 512), (256, 192, 768, 512), (256, 384, 256, 256), (256, 384, 512, 256), (256, 384, 768, 256), (256, 384, 128, 256), (256, 192, 256, 512)]
    for (B, L, Cin, Cout) in SHAPES:
        xd = torch.randn(B * L, Cin, device=G.DEV).bfloat16(); wd = (torch.randn(3, Cout, Cin, device=G.DEV) / math.sqrt(3 * Cin)).bfloat16()
        bd = torch.randn(Cout, device=G.DEV); yd = torch.empty(B * L, Cout, device=G.DEV, dtype=torch.bfloat16)
        wk = torch.empty_like(wd); G.check(lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(wd), G.ptr(wk), Cout, Cin, dt))
        dyd = torch.randn(B * L, Cout, device=G.DEV).bfloat16(); dxd = torch.empty(B * L, Cin, device=G.DEV, dtype=torch.bfloat16)
        wt = torch.empty_like(wd); G.check(lib.eegldm_conv1d_pack_dgrad(c.h, G.ptr(wd), G.ptr(wt), Cout, Cin, dt))
        res = {}
        ed = torch.randn(B, Cout, device=G.DEV); rd = torch.randn(B * L, Cout, device=G.DEV).bfloat16(); rdx = torch.randn(B * L, Cin, device=G.DEV).bfloat16()
        full = os.environ.get("BIG_TIME_FULL") is not None      # with the embedding row and the residual (the ResBlock's second conv)
        for name, env, envp in (("bigp", None, None), ("big", None, 1), ("old", 1, None)):
            setenv(EEGLDM_NO_GEMM_BIG=env, EEGLDM_GEMM_BIG_NO_PERSIST=envp)
            for kind in ("fwd", "dgrad"):
                def call():
                    if kind == "fwd":
                        G.check(lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, 3, 1, 1, 1,
                                                      G.ptr(ed) if full else None, Cout if full else 0, G.ptr(rd) if full else None, Cout if full else 0, dt))
                    else:
                        G.check(lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dxd), Cin, B, L, Cin, Cout, 3, 1, 1, 1, G.ptr(rdx) if full else None, Cin if full else 0, dt))
                for _ in range(3): call()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): call()
                torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 20 * 1e6
                res[(name, kind)] = us
        setenv(EEGLDM_NO_GEMM_BIG=None, EEGLDM_GEMM_BIG_NO_PERSIST=None)
        fl = 2.0 * B * L * Cin * Cout * 3
        print(f"B={B} L={L} Cin={Cin} Cout={Cout}{' full' if full else ''}: " + ";  ".join(
            f"{kind} " + " ".join(f"{n} {res[(n, kind)]:.1f} us ({fl / res[(n, kind)] / 1e6:.0f})" for n in ("bigp", "big", "old")) for kind in ("fwd", "dgrad")), flush=True)
        G.check(lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd)))
    for (B, L, Cin, Cout) in ([] if os.environ.get("BIG_SHAPES") else [(256, 192, 512, 1536), (256, 192, 512, 512), (256, 192, 1024, 512), (256, 384, 768, 256), (256, 384, 512, 256), (256, 768, 256, 256)]):
        xd = torch.randn(B * L, Cin, device=G.DEV).bfloat16(); wd = (torch.randn(1, Cout, Cin, device=G.DEV) / math.sqrt(Cin)).bfloat16()
        bd = torch.randn(Cout, device=G.DEV); yd = torch.empty(B * L, Cout, device=G.DEV, dtype=torch.bfloat16)
        dyd = torch.randn(B * L, Cout, device=G.DEV).bfloat16(); dxd = torch.empty(B * L, Cin, device=G.DEV, dtype=torch.bfloat16)
        wt = torch.empty_like(wd); G.check(lib.eegldm_conv1d_pack_dgrad_k(c.h, G.ptr(wd), G.ptr(wt), Cout, Cin, 1, dt))
        res = {}
        V = (("bigp", None, None), ("big", None, 1), ("old", 1, None))
        for name, env, envp in V:
            setenv(EEGLDM_NO_GEMM_BIG=env, EEGLDM_GEMM_BIG_NO_PERSIST=envp)
            for kind in ("fwd", "dgrad"):
                def call():
                    if kind == "fwd":
                        G.check(lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, 1, 1, 0, 0, None, 0, None, 0, dt))
                    else:
                        G.check(lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dxd), Cin, B, L, Cin, Cout, 1, 1, 0, 0, None, 0, dt))
                for _ in range(3): call()
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(20): call()
                torch.cuda.synchronize(); res[(name, kind)] = (time.perf_counter() - t0) / 20 * 1e6
        setenv(EEGLDM_NO_GEMM_BIG=None, EEGLDM_GEMM_BIG_NO_PERSIST=None)
        fl = 2.0 * B * L * Cin * Cout
        print(f"1x1 B={B} L={L} Cin={Cin} Cout={Cout}: " + ";  ".join(
            f"{kind} " + " ".join(f"{n} {res[(n, kind)]:.1f} us ({fl / res[(n, kind)] / 1e6:.0f})" for n, *_ in V) for kind in ("fwd", "dgrad")), flush=True)
        G.check(lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd)))
