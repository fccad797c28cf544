"""Round-6 measurement: Conv1d(k 3) over SiLU(GroupNorm(x)) as ONE launch (normalisation applied to the operand tile in LDS,
eegldm_conv1d_fwd_gn -> gemm_big_kernel<XF = 1>) against the two launches of the product path (gn_fwd_resident writes the normalised tensor, the
persistent big-tile conv reads it) and against the non-persistent conv kernel the prototype is built on (EEGLDM_GEMM_BIG_NO_PERSIST=1).
Buffers rotate through more than the 256 MB Infinity Cache.  Usage: python tools/r06/gn_onload_bench.py [B L Cin Cout]"""
import math, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, check, ptr, BF16
B, L, Cin, Cout = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (256, 192, 512, 512)
G, NBUF, REP = 32, 8, 40
ctx = eegldm.default_context(0)
dev = "cuda"
xs = [torch.randn(B * L, Cin, device=dev).bfloat16() for _ in range(NBUF)]
as_ = [torch.empty(B * L, Cin, device=dev, dtype=torch.bfloat16) for _ in range(NBUF)]
ys = [torch.empty(B * L, Cout, device=dev, dtype=torch.bfloat16) for _ in range(NBUF)]
w = (torch.randn(3, Cout, Cin, device=dev) / math.sqrt(3 * Cin)).bfloat16().contiguous(); wk = torch.empty_like(w)
bias = torch.randn(Cout, device=dev); gamma = torch.ones(Cin, device=dev); beta = torch.zeros(Cin, device=dev)
st = torch.empty(B * G * 2, device=dev)
check(lib.eegldm_conv1d_pack_kblocked(ctx.h, ptr(w), ptr(wk), Cout, Cin, BF16))
def gn(i): check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(xs[i]), Cin, ptr(gamma), ptr(beta), ptr(as_[i]), Cin, ptr(st), B, L, Cin, G, 1e-6, 1, 0, None, 0, BF16))
def conv(i): check(lib.eegldm_conv1d_fwd(ctx.h, ptr(as_[i]), Cin, ptr(w), ptr(bias), ptr(ys[i]), Cout, B, L, Cin, Cout, 3, 1, 1, 1, None, 0, None, 0, BF16))
def fused(i): check(lib.eegldm_conv1d_fwd_gn(ctx.h, ptr(xs[i]), Cin, ptr(w), ptr(bias), ptr(gamma), ptr(beta), ptr(st), G, 1, ptr(ys[i]), Cout, B, L, Cin, Cout, None, 0, None, 0, BF16))
def timeit(f):
    for i in range(NBUF): f(i)
    torch.cuda.synchronize(); t0 = time.time()
    for r in range(REP): f(r % NBUF)
    torch.cuda.synchronize(); return (time.time() - t0) / REP * 1e6
def setenv(k, v):
    if v is None: os.environ.pop(k, None)
    else: os.environ[k] = v
    lib.eegldm_debug_reload_env()
gn(0)
res = {}
for rep in range(3):
    res.setdefault("gn_fwd (read x, write a)", []).append(timeit(gn))
    res.setdefault("conv, persistent big tile (product path)", []).append(timeit(conv))
    setenv("EEGLDM_GEMM_BIG_NO_PERSIST", "1")
    res.setdefault("conv, one tile per workgroup (the prototype's base kernel)", []).append(timeit(conv))
    setenv("EEGLDM_GEMM_BIG_NO_PERSIST", None)
    res.setdefault("gn + conv, two launches back to back", []).append(timeit(lambda i: (gn(i), conv(i))))
    res.setdefault("fused: table kernel + conv with the transform in LDS", []).append(timeit(fused))
flops = 2.0 * B * L * Cout * Cin * 3
print(f"B {B} L {L} {Cin} -> {Cout}  ({flops/1e9:.1f} GFLOP)")
for k, v in res.items():
    m = min(v); print(f"  {k:62s} {m:8.1f} us   ({flops/m/1e6:7.1f} TF/s-equivalent)   all {[round(t, 1) for t in v]}")
