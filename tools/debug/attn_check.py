"""Developer tool: eegldm_attention_fwd / _bwd (bf16) against a torch fp32 reference on the GPU; reports which (sample, 64-row tile) blocks are off."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
for (B, T, C) in ([(5, 768, 256)] if os.environ.get("ATTN_CHECK_SHORT") else [(5, 768, 256), (8, 768, 256), (2, 768, 512), (16, 768, 512), (64, 768, 512)]):
    torch.manual_seed(B + C)
    qkv = torch.randn(B * T, 3 * C, device="cuda").bfloat16(); dout = torch.randn(B * T, C, device="cuda").bfloat16()
    q, k, v = [t.float().reshape(B, T, C) for t in qkv.float().split(C, dim=1)]
    q.requires_grad_(True); k.requires_grad_(True); v.requires_grad_(True)
    w = torch.softmax(torch.einsum("btc,bsc->bts", q, k) / C ** 0.5, dim=-1)
    ref = torch.einsum("bts,bsc->btc", w, v)
    ref.backward(dout.float().reshape(B, T, C))
    dref = torch.cat([q.grad, k.grad, v.grad], dim=2)
    for rep in range(3):
        out = torch.empty(B * T, C, device="cuda", dtype=torch.bfloat16); pr = torch.empty(B * T * T, device="cuda", dtype=torch.bfloat16)
        s1 = torch.empty(B * T * T, device="cuda"); s2 = torch.empty(B * T * T, device="cuda", dtype=torch.bfloat16); dq = torch.empty(B * T, 3 * C, device="cuda", dtype=torch.bfloat16)
        check(lib.eegldm_attention_fwd(ctx.h, ptr(qkv), 3 * C, ptr(out), C, ptr(pr), ptr(s1), B, T, C, 1))
        check(lib.eegldm_attention_bwd(ctx.h, ptr(qkv), 3 * C, ptr(pr), ptr(dout), C, ptr(dq), 3 * C, ptr(s1), ptr(s2), B, T, C, 1))
        torch.cuda.synchronize()
        e = (out.float().reshape(B, T, C) - ref.detach()); e = torch.nan_to_num(e, nan=1e9).abs().reshape(B, T // 64, 64 * C).amax(-1)
        bad = (e > 0.05).nonzero().tolist()
        g = (dq.float().reshape(B, T, 3 * C) - dref); g = torch.nan_to_num(g, nan=1e9).abs()
        gq = g[:, :, :C].reshape(B, T // 64, 64 * C).amax(-1); badq = (gq > 0.05 * float(dref.abs().max())).nonzero().tolist()
        gkv = g[:, :, C:].amax((1, 2))
        print(f"B={B} T={T} C={C} rep {rep}: out max err {float(e.max()):.3e}, bad (sample, tile) blocks {bad[:12]} ({len(bad)}); dq bad blocks {badq[:12]} ({len(badq)}); dk/dv max err per sample {[round(float(x), 3) for x in gkv[:8]]} (scale {float(dref.abs().max()):.2f})", flush=True)
