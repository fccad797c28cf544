import os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
torch.manual_seed(0)
B, L, Cin, Cout = 256, 768, 128, 128
M = B * L
x = torch.randn(M, Cin, device="cuda").bfloat16()
w = (torch.randn(3, Cout, Cin, device="cuda") / (3 * Cin) ** 0.5).bfloat16()
for mode in ("plain", "bias", "emb", "res"):
    bias = torch.randn(Cout, device="cuda") if mode == "bias" else None
    emb = torch.randn(B, Cout, device="cuda") if mode == "emb" else None
    res = torch.randn(M, Cout, device="cuda").bfloat16() if mode == "res" else None
    y = torch.zeros(M, Cout, device="cuda", dtype=torch.bfloat16)
    check(lib.eegldm_conv1d_fwd(ctx.h, ptr(x), Cin, ptr(w), ptr(bias), ptr(y), Cout, B, L, Cin, Cout, 3, 1, 1, 1, ptr(emb), Cout if emb is not None else 0, ptr(res), Cout if res is not None else 0, 1))
    torch.cuda.synchronize()
    xr = x.float().reshape(B, L, Cin).permute(0, 2, 1); wr = w.float().permute(1, 2, 0)
    ref = F.conv1d(xr, wr, bias, padding=1)
    if emb is not None: ref = ref + emb[:, :, None]
    if res is not None: ref = ref + res.float().reshape(B, L, Cout).permute(0, 2, 1)
    got = y.float().reshape(B, L, Cout).permute(0, 2, 1)
    err = (got - ref).abs().amax(dim=1).reshape(-1)          # per flattened row
    bad = (err > 0.1).nonzero().reshape(-1)
    tiles = torch.unique(bad // 64)
    print(mode, "bad rows", int(bad.numel()), "bad tiles", int(tiles.numel()), "tile%6:", torch.bincount(tiles % 6, minlength=6).tolist(), "first bad rows", bad[:8].tolist(), "row%64 hist", torch.bincount(bad % 64, minlength=64).tolist()[:8])
