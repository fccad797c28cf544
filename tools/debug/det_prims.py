"""Developer tool: bitwise run-to-run reproducibility of the forward primitives at the UNet's shapes (bf16, B=256)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B = 256
def same(fn, out, n=3):
    fn(); ctx.sync(); ref = out.clone()
    bad = 0
    for _ in range(n):
        out.zero_(); fn(); ctx.sync(); bad += int((out.view(torch.int16) != ref.view(torch.int16)).sum())
    return bad
for (L, C, G) in [(768, 128, 32), (384, 256, 32), (192, 512, 32), (192, 1024, 32), (384, 768, 32), (768, 384, 32)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); y = torch.empty_like(x); ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * G * 2, device="cuda")
    print(f"GN fwd L={L} C={C}: mismatching elements", same(lambda: check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, G, 1e-6, 1, 0, None, 0, 1)), y))
for (L, Ci, Co, K) in [(768, 128, 128, 3), (384, 256, 256, 3), (192, 512, 512, 3), (192, 1024, 512, 3), (192, 512, 1536, 1), (192, 512, 512, 1), (768, 1, 128, 3), (768, 128, 1, 3)]:
    R = B * L
    x = torch.randn(R, Ci, device="cuda").bfloat16(); w = (torch.randn(K, Co, Ci, device="cuda") * 0.05).bfloat16(); bias = torch.randn(Co, device="cuda"); y = torch.empty(R, Co, device="cuda", dtype=torch.bfloat16)
    rv = torch.randn(B, Co, device="cuda"); res = torch.randn(R, Co, device="cuda").bfloat16()
    p = (K - 1) // 2
    print(f"conv fwd L={L} {Ci}->{Co} k{K}: mismatching", same(lambda: check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), Ci, ptr(w), ptr(bias), ptr(y), Co, B, L, Ci, Co, K, 1, p, p, (ptr(rv) if min(Ci, Co) >= 16 else None), Co, (ptr(res) if min(Ci, Co) >= 16 else None), Co, 1)), y))
T, C = 192, 512
qkv = torch.randn(B * T, 3 * C, device="cuda").bfloat16(); out = torch.empty(B * T, C, device="cuda", dtype=torch.bfloat16); probs = torch.empty(B, T, T, device="cuda", dtype=torch.bfloat16); lg = torch.empty(B, T, T, device="cuda")
print("attention fwd: mismatching", same(lambda: check(lib.eegldm_attention_fwd(ctx.h, ptr(qkv), 3 * C, ptr(out), C, ptr(probs), ptr(lg), B, T, C, 1)), out))
print("--- backward (dx outputs)")
for (L, C, G) in [(768, 128, 32), (384, 256, 32), (192, 512, 32), (192, 1024, 32), (384, 768, 32)]:
    R = B * L
    x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); dx = torch.empty_like(x); dxr = torch.randn(R, C, device="cuda").bfloat16()
    ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * G * 2, device="cuda"); y = torch.empty_like(x)
    check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, G, 1e-6, 1, 0, None, 0, 1))
    dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    print(f"GN bwd L={L} C={C}: mismatching", same(lambda: check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, G, 1, 0, ptr(dxr), C, 1)), dx))
for (L, Ci, Co, K) in [(768, 128, 128, 3), (192, 512, 512, 3), (192, 512, 1024, 3), (192, 512, 1536, 1), (768, 1, 128, 3), (768, 128, 1, 3)]:
    R = B * L
    dy = torch.randn(R, Co, device="cuda").bfloat16(); w = (torch.randn(K, Co, Ci, device="cuda") * 0.05).bfloat16(); dx = torch.empty(R, Ci, device="cuda", dtype=torch.bfloat16)
    p = (K - 1) // 2
    print(f"conv dgrad L={L} {Ci}<-{Co} k{K}: mismatching", same(lambda: check(lib.eegldm_conv1d_bwd_data(ctx.h, ptr(dy), Co, ptr(w), ptr(dx), Ci, B, L, Ci, Co, K, 1, p, p, None, 0, 1)), dx))
do = torch.randn(B * T, C, device="cuda").bfloat16(); dqkv = torch.empty(B * T, 3 * C, device="cuda", dtype=torch.bfloat16); dl = torch.empty(B, T, T, device="cuda", dtype=torch.bfloat16)
check(lib.eegldm_attention_fwd(ctx.h, ptr(qkv), 3 * C, ptr(out), C, ptr(probs), ptr(lg), B, T, C, 1))
print("attention bwd: mismatching", same(lambda: check(lib.eegldm_attention_bwd(ctx.h, ptr(qkv), 3 * C, ptr(probs), ptr(do), C, ptr(dqkv), 3 * C, ptr(lg), ptr(dl), B, T, C, 1)), dqkv))
