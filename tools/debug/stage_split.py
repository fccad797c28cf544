"""Developer tool: the stage of the fused 3-tap weight-gradient kernel split into DMA wait | barrier | first fragments | MFMA phase
(library built with -DEEG_STAGE_TIMING -DEEG_STAGE_SPLIT: tools/debug/libeegldm_split.so)."""
import ctypes as C, os, sys, numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["EEGLDM_LIB"] = os.path.join(HERE, "libeegldm_split.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B = 256
for (ci, co, L) in [(128, 128, 768), (256, 256, 384), (512, 512, 192)]:
    R = B * L
    x = torch.randn(R, ci, device="cuda").bfloat16(); dy = torch.randn(R, co, device="cuda").bfloat16(); dw = torch.zeros(3, co, ci, device="cuda")
    nblk = 4096
    buf = np.zeros(nblk * 64, dtype=np.uint64)
    for _ in range(3):
        check(lib.eegldm_conv1d_bwd_weight(ctx.h, ptr(x), ci, ptr(dy), co, ptr(dw), None, B, L, ci, co, 3, 1, 1, 1, 1))
        lib.eegldm_debug_read_tlog(ctx.h, buf.ctypes.data_as(C.c_void_p), C.c_long(buf.size))
    t = buf.reshape(nblk, 64).astype(np.int64); t = t[t[:, 0] != 0]
    cnt = (t[:, :62] != 0).sum(axis=1); n = int(np.median(cnt)); t = t[cnt == n]
    d = np.diff(t[:, :n], axis=1)                    # stamp 0 = before the loop; then per stage: dma wait | barrier (+issue) | first fragments | MFMA phase
    nst = (n - 1) // 4
    med = lambda a: float(np.median(a))
    seg = [d[:, 4 + k: 4 * nst: 4] for k in range(4)]      # skip the first stage (prologue)
    print(f"wgrad k3 {ci}->{co} L={L}: {len(t)} blocks, {nst} stamped stages | DMA wait {med(seg[0]):.0f}  barrier {med(seg[1]):.0f}  first fragments {med(seg[2]):.0f}  MFMA phase {med(seg[3]):.0f}  (p90: {np.percentile(seg[0],90):.0f} {np.percentile(seg[1],90):.0f} {np.percentile(seg[2],90):.0f} {np.percentile(seg[3],90):.0f}) cycles")
