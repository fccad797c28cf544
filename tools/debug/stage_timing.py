"""Developer tool: per-stage timeline of the conv kernel from an instrumented build (-DEEG_STAGE_TIMING)."""
import ctypes as C, os, sys, numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["EEGLDM_LIB"] = os.environ.get("EEGLDM_DBG_LIB") or os.path.join(HERE, "libeegldm_dbg.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B, L = 256, 192
for (ci, co, k) in ([] if os.environ.get('ST_WGRAD_ONLY') else [(512, 512, 3), (128, 128, 3), (512, 512, 1)]):
    if ci == 128: L = 768
    else: L = 192
    R = B * L
    x = torch.randn(R, ci, device="cuda").bfloat16(); w = (torch.randn(k, co, ci, device="cuda") * 0.05).bfloat16(); b = torch.zeros(co, device="cuda")
    y = torch.empty(R, co, device="cuda", dtype=torch.bfloat16)
    nblk = (R // 128) * (co // 128)
    buf = np.zeros(nblk * 64, dtype=np.uint64)
    pad = 1 if k == 3 else 0
    for _ in range(3):
        check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), ci, ptr(w), ptr(b), ptr(y), co, B, L, ci, co, k, 1, pad, pad, None, 0, None, 0, 1))
        lib.eegldm_debug_read_tlog(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_long(buf.size))
    t = buf.reshape(nblk, 64).astype(np.int64)
    t = t[t[:, 0] != 0]          # fewer blocks than assumed when a larger tile variant is selected
    nst = ci // (32 if k == 3 else 64)
    # stamps: [0]=before loop, then per stage: after barrier/issue, after mfma ; then epilogue start, end
    d = np.diff(t[:, :2 * nst + 8], axis=1)
    wait = d[:, 0:2 * nst:2]      # before-loop/prev-mfma -> after barrier+issue
    mfma = d[:, 1:2 * nst:2]
    pro = d[:, 0]
    e = d[:, 2 * nst:]            # [sync->epi start, E1 lds written, E2 barrier, E3 reads issued, E4 reads landed, E5 stores issued, E6 stores retired]
    med = lambda a: float(np.median(a))
    print(f"conv k{k} {ci}->{co} L={L}: blocks {nblk}, stages {nst}; values in shader cycles")
    print(f"  first wait {med(pro):.0f}; per-stage wait {med(wait[:,1:]):.0f} (p90 {np.percentile(wait[:,1:],90):.0f}); per-stage MFMA phase {med(mfma):.0f} (p90 {np.percentile(mfma,90):.0f})")
    names = ["final sync", "LDS tile written", "barrier", "global reads issued", "global reads landed", "LDS read + stores issued", "stores retired"]
    for n, c in zip(names, e.T): print(f"  epilogue {n:26s} {med(c):8.0f}  (p90 {np.percentile(c,90):.0f})")
    last = np.array([row[:62][row[:62] != 0][-1] for row in t]); first = t[:, 0]
    w0 = t[:, 62]; w1 = t[:, 63]                      # 100 MHz wall clock of the first / last stamp of each block
    span_us = (w1.max() - w0.min()) / 100.0
    rel = np.sort(w0 - w0.min()) / 100.0
    endrel = np.sort(w1 - w0.min()) / 100.0
    print(f"  block duration median {med(last - first):.0f} cycles = {med(w1 - w0)/100.0:.1f} us; kernel span {span_us:.1f} us; block starts p10/p50/p90/max {rel[len(rel)//10]:.1f}/{rel[len(rel)//2]:.1f}/{rel[9*len(rel)//10]:.1f}/{rel[-1]:.1f} us; first block end {endrel[0]:.1f} us")

# weight-gradient kernels (fused 3-tap TN, split-K, direct register atomics)
for (ci, co, L) in [(128, 128, 768), (256, 256, 384), (512, 512, 192)]:
    R = B * L
    x = torch.randn(R, ci, device="cuda").bfloat16(); dy = torch.randn(R, co, device="cuda").bfloat16()
    dw = torch.zeros(3, co, ci, device="cuda")
    tiles = ((co + 127) // 128) * ((ci + 63) // 64)
    splitk = min((512 + tiles - 1) // tiles, (R + 511) // 512)
    nblk = tiles * splitk
    buf = np.zeros(nblk * 64, dtype=np.uint64)
    for _ in range(3):
        check(lib.eegldm_conv1d_bwd_weight(ctx.h, ptr(x), ci, ptr(dy), co, ptr(dw), None, B, L, ci, co, 3, 1, 1, 1, 1))
        lib.eegldm_debug_read_tlog(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_long(buf.size))
    t = buf.reshape(nblk, 64).astype(np.int64)
    cnt = (t[:, :62] != 0).sum(axis=1)
    n = int(np.median(cnt))
    t = t[cnt == n]
    d = np.diff(t[:, :n], axis=1)
    nst = (n - 4) // 2
    med = lambda a: float(np.median(a))
    print(f"wgrad k3 {ci}->{co} L={L}: blocks {nblk} (tiles {tiles} x splitk {splitk}), stamps {n}, stages {nst}")
    print(f"  first wait {med(d[:,0]):.0f}; per-stage wait {med(d[:,2:2*nst:2]):.0f}; per-stage MFMA {med(d[:,1:2*nst:2]):.0f}")
    print(f"  tail: {[int(med(c)) for c in d[:, 2*nst:].T]}  (final sync, atomics issued, atomics retired); total {med(t[:,n-1]-t[:,0]):.0f}")
    w0 = t[:, 62]; w1 = t[:, 63]; cyc = t[:, n - 1] - t[:, 0]
    print(f"  stamped span: {med(cyc):.0f} cycles in {med(w1 - w0) / 100.0:.1f} us -> {med(cyc / np.maximum(w1 - w0, 1)) * 100 / 1e3:.2f} GHz shader clock; kernel span {(w1.max() - w0.min()) / 100.0:.1f} us")
