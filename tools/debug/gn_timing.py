"""Developer tool (make dbg build): phase timeline of the one-pass GroupNorm backward kernel, thread 0 of every block."""
import ctypes as C, os, sys, numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["EEGLDM_LIB"] = os.path.join(HERE, "libeegldm_dbg.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B = 256
for (L, Cc) in [(768, 128), (384, 256), (192, 512)]:
    R = B * L
    x = torch.randn(R, Cc, device="cuda").bfloat16(); dy = torch.randn(R, Cc, device="cuda").bfloat16(); dx = torch.empty_like(x)
    ga = torch.ones(Cc, device="cuda"); be = torch.zeros(Cc, device="cuda"); st = torch.zeros(B * 32 * 2, device="cuda"); st[1::2] = 1.0
    dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
    for _ in range(3):
        check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), Cc, ptr(ga), ptr(be), ptr(st), ptr(dy), Cc, ptr(dx), Cc, ptr(dg), ptr(db), B, L, Cc, 32, 1, 0, None, 0, 1))
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.eegldm_debug_read_gn_tlog.argtypes = [C.c_void_p, C.c_long]
    lib.eegldm_debug_read_gn_tlog(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.size))
    t = buf.reshape(4096, 8).astype(np.int64); t = t[t[:, 0] != 0]
    d = np.diff(t[:, :6], axis=1)
    names = ["loads (x batched, dy) landed", "barrier", "pass 1 + LDS atomics", "barrier", "pass 2 + stores retired"]
    print(f"L={L} C={Cc}: {len(t)} blocks; start spread {int(t[:,0].max()-t[:,0].min())} cycles; block total median {int(np.median(t[:,5]-t[:,0]))}")
    for n, c in zip(names, d.T): print(f"   {n:32s} {int(np.median(c)):8d}  (p90 {int(np.percentile(c,90))})")
    s0 = np.sort(t[:, 0] - t[:, 0].min())
    print("   block start offsets (cycles): p25 %d p50 %d p75 %d" % (s0[len(s0)//4], s0[len(s0)//2], s0[3*len(s0)//4]))

print("---- forward one-pass kernel")
for (L, Cc) in [(768, 128), (192, 512)]:
    R = B * L
    x = torch.randn(R, Cc, device="cuda").bfloat16(); y = torch.empty_like(x)
    ga = torch.ones(Cc, device="cuda"); be = torch.zeros(Cc, device="cuda"); st = torch.zeros(B * 32 * 2, device="cuda")
    for _ in range(3):
        check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), Cc, ptr(ga), ptr(be), ptr(y), Cc, ptr(st), B, L, Cc, 32, 1e-6, 1, 0, None, 0, 1))
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.eegldm_debug_read_gn_tlog(buf.ctypes.data_as(C.c_void_p), C.c_long(buf.size))
    t = buf.reshape(4096, 8).astype(np.int64); t = t[t[:, 0] != 0][:512 if Cc == 128 else 512]
    d = np.diff(t[:, :6], axis=1)
    names = ["x loads landed", "barrier", "sum + barrier (mean)", "centered squares + barrier (var)", "normalise + SiLU + stores retired"]
    print(f"L={L} C={Cc}: {len(t)} blocks; block total median {int(np.median(t[:,5]-t[:,0]))}")
    for n, c in zip(names, d.T): print(f"   {n:36s} {int(np.median(c)):8d}  (p90 {int(np.percentile(c,90))})")
