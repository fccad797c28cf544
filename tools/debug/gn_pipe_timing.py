"""Developer tool (make dbg build): per-slab phase timeline of the pipelined GroupNorm backward kernel, thread 0 of the first 512 workgroups."""
import ctypes as C, os, sys, numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["EEGLDM_LIB"] = os.path.join(HERE, "libeegldm_dbg.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B = 256
names = ["barrier A (slab landed)", "LDS -> registers + barrier B", "issue x / dy / stats DMA", "pass 1 + LDS atomics", "barrier C, group sums, addend wait, barrier D",
         "pass 2 + stores issued", "barrier E, column sums, addend DMA, counted wait"]
for with_e in (0, 1):
  for (L, Cc) in [(768, 128), (384, 256), (192, 512), (192, 1024)]:
    R = B * L
    sets = []
    for s in range(3):
        sets.append((torch.randn(R, Cc, device="cuda").bfloat16(), torch.randn(R, Cc, device="cuda").bfloat16(), torch.randn(R, Cc, device="cuda").bfloat16(), torch.empty(R, Cc, device="cuda", dtype=torch.bfloat16)))
    ga = torch.ones(Cc, device="cuda"); be = torch.zeros(Cc, device="cuda"); st = torch.zeros(B * 32 * 2, device="cuda"); st[1::2] = 1.0
    dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    for i in range(4):
        x, dy, e, dx = sets[i % 3]
        check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), Cc, ptr(ga), ptr(be), ptr(st), ptr(dy), Cc, ptr(dx), Cc, ptr(dg), ptr(db), B, L, Cc, 32, 1, 0, ptr(e) if with_e else None, Cc, 1))
    ctx.sync()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.eegldm_debug_read_gn_tlog.argtypes = [C.c_void_p, C.c_long]
    lib.eegldm_debug_read_gn_tlog(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.size))
    t = buf.reshape(512, 8, 8).astype(np.int64)[:256]
    nk = int((t[0, :, 0] != 0).sum())
    print(f"addend={with_e} L={L} C={Cc}: {nk} slabs per workgroup; kernel span {int(t[:, nk-1, 7].max() - t[:, 0, 0].min())} cycles (100 MHz counter? see ratio); first-slab start spread {int(t[:,0,0].max()-t[:,0,0].min())}")
    for k in range(nk):
        d = np.diff(t[:, k, :], axis=1)
        print(f"   slab {k}: total {int(np.median(t[:,k,7]-t[:,k,0])):6d} | " + " ".join(f"{int(np.median(c)):6d}" for c in d.T))
    print("   phases: " + " | ".join(names))
