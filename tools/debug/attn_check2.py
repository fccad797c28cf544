import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B, T, C = 1, 768, 256
torch.manual_seed(1)
qkv = torch.randn(B * T, 3 * C, device="cuda").bfloat16()
q, k, v = [t.float().reshape(B, T, C) for t in qkv.float().split(C, dim=1)]
w = torch.softmax(torch.einsum("btc,bsc->bts", q, k) / C ** 0.5, dim=-1)
out = torch.empty(B * T, C, device="cuda", dtype=torch.bfloat16); pr = torch.empty(B * T * T, device="cuda", dtype=torch.bfloat16); s1 = torch.empty(B * T * T, device="cuda")
check(lib.eegldm_attention_fwd(ctx.h, ptr(qkv), 3 * C, ptr(out), C, ptr(pr), ptr(s1), B, T, C, 1)); torch.cuda.synchronize()
P = pr.float().reshape(B, T, T)
print("P (probabilities written by the kernel) max err vs torch:", float((P - w).abs().max()))
o = torch.nan_to_num(out.float().reshape(T, C), nan=1e9, posinf=1e9, neginf=-1e9)
# which 32-key slices are consistent?  out_ref(ks_set) -- find per (row-half of tile 0, col quarter) the error when slice ks is dropped / doubled
ref = torch.einsum("ts,sc->tc", P[0], v[0])
e = (o - ref).abs()
print("tile 0: bad fraction per (32-row half, 64-col quarter):", [[round(float((e[h*32:(h+1)*32, c*64:(c+1)*64] > 0.05).float().mean()), 3) for c in range(4)] for h in range(2)])
print("rows of tile 0 with any bad element:", (e[:64] > 0.05).any(1).nonzero()[:, 0].tolist())
print("cols with any bad element (tile 0):", (e[:64] > 0.05).any(0).nonzero()[:, 0].tolist()[:40])
print("sample values o[0,:8]", o[0, :8].tolist(), "ref", ref[0, :8].tolist())
# least squares: o[0:64] ~ sum_ks a_ks * (P[:, ks slice] @ v[ks slice]) : which slices are used with which weight
parts = torch.stack([P[0, :64, ks*32:(ks+1)*32] @ v[0, ks*32:(ks+1)*32] for ks in range(24)], 0)      # [24][64][C]
good = (e[:64] < 1e3)
A = parts.reshape(24, -1).T; bvec = torch.clamp(o[:64], -1e3, 1e3).reshape(-1, 1)
sol = torch.linalg.lstsq(A.cpu(), bvec.cpu()).solution.squeeze()
print("least-squares weight of each key slice in tile 0 (1 = used once):", [round(float(x), 2) for x in sol])
