"""Developer tool: one-pass GroupNorm backward beside an UNRELATED kernel stream (torch matmuls on another stream).
Compares every run bitwise with a quiet run of the 1024-thread kernel (EEGLDM_GN_BWD_NTH picks the block size under test)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
torch.manual_seed(0)
B = 256
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda").bfloat16(); b = torch.randn(4096, 4096, device="cuda").bfloat16()
for (L, C) in [(192, 512), (384, 256), (384, 512), (192, 1024), (768, 128)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); ad = torch.randn(R, C, device="cuda").bfloat16()
    ga = torch.rand(C, device="cuda") + 0.5; be = torch.randn(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); y = torch.empty_like(x)
    check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1))
    def run(noise):
        dx = torch.empty_like(x); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
        torch.cuda.synchronize()
        if noise:
            with torch.cuda.stream(side):
                for _ in range(40): c = a @ b
        for _ in range(6):
            check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, ptr(ad), C, 1))
        torch.cuda.synchronize()
        return dx
    quiet = run(False)
    nd = [int((quiet != run(True)).sum()) for _ in range(4)]
    print(f"NTH={os.environ.get('EEGLDM_GN_BWD_NTH', 'default')} L={L} C={C}: elements differing from the quiet run {nd}; checksum {float(quiet.float().abs().sum()):.6e}", flush=True)
