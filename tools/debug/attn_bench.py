"""Developer tool: time eegldm_attention_fwd / _bwd on one shape (default: the pixel-space model's B = 64, T = 768, C = 512, bf16)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
B, T, C = [int(v) for v in sys.argv[1:4]] if len(sys.argv) > 3 else (64, 768, 512)
ctx = eegldm.default_context(0)
torch.manual_seed(0)
qkv = torch.randn(B * T, 3 * C, device="cuda").bfloat16(); dout = torch.randn(B * T, C, device="cuda").bfloat16()
out = torch.empty(B * T, C, device="cuda", dtype=torch.bfloat16); pr = torch.empty(B * T * T, device="cuda", dtype=torch.bfloat16)
s1 = torch.empty(B * T * T, device="cuda"); s2 = torch.empty(B * T * T, device="cuda", dtype=torch.bfloat16); dq = torch.empty(B * T, 3 * C, device="cuda", dtype=torch.bfloat16)
def fwd(): check(lib.eegldm_attention_fwd(ctx.h, ptr(qkv), 3 * C, ptr(out), C, ptr(pr), ptr(s1), B, T, C, 1))
def bwd(): check(lib.eegldm_attention_bwd(ctx.h, ptr(qkv), 3 * C, ptr(pr), ptr(dout), C, ptr(dq), 3 * C, ptr(s1), ptr(s2), B, T, C, 1))
for name, fn, gf in [("fwd", fwd, 4.0 * B * T * T * C), ("bwd", bwd, 8.0 * B * T * T * C)]:
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): fn()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    print(f"attention {name} B={B} T={T} C={C} [{'composition' if os.environ.get('EEGLDM_ATTN_NO_LONG') else 'fused'}]: {dt*1e6:.0f} us  {gf/dt*1e-12:.0f} TF/s")
