"""-m gpu: round-6 prototype -- Conv1d(k 3, padding 1) over SiLU(GroupNorm(x)) with the normalisation applied to the conv's operand tile in LDS
(csrc/gemm_big.hip, gemm_big_kernel<.., XF = 1>; eegldm_conv1d_fwd_gn) against the two-launch form of the same library (eegldm_groupnorm_fwd
writes the normalised tensor, eegldm_conv1d_fwd reads it) and against torch fp32.  Reference ops: in_layers / out_layers of
/root/reference/src/models/unet.py:261-263,287-291."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import normal  # noqa: E402


@pytest.mark.parametrize("B,L,Cin,Cout,G,extras", [(4, 192, 512, 512, 32, False), (2, 384, 256, 256, 32, True), (3, 192, 1024, 512, 32, True),
                                                    (1, 768, 128, 256, 32, False)])
def test_conv3_with_groupnorm_silu_on_operand_load(B, L, Cin, Cout, G, extras):
    import gpu_util as G_
    dt = G_.BF16
    x = torch.from_numpy(normal((B, Cin, L), seed=1)) * 1.7 + 0.3
    w = torch.from_numpy(normal((Cout, Cin, 3), seed=2)) / math.sqrt(3 * Cin)
    b = torch.from_numpy(normal((Cout,), seed=3)); gamma = 1 + 0.2 * torch.from_numpy(normal((Cin,), seed=4)); beta = 0.3 * torch.from_numpy(normal((Cin,), seed=5))
    emb = torch.from_numpy(normal((B, Cout), seed=6)) if extras else None
    res = torch.from_numpy(normal((B, Cout, L), seed=7)) if extras else None
    c = G_.ctx()
    xd, wd, bd = G_.nlc(x, dt), G_.pack_w(w, dt), b.to(G_.DEV)
    gd, bed = gamma.to(G_.DEV), beta.to(G_.DEV)
    wk = torch.empty_like(wd)
    G_.check(G_.lib.eegldm_conv1d_pack_kblocked(c.h, G_.ptr(wd), G_.ptr(wk), Cout, Cin, dt))
    try:
        # two launches: GroupNorm + SiLU writes `a`, the conv reads it
        a = torch.empty_like(xd); st = torch.empty(B * G * 2, device=G_.DEV)
        G_.check(G_.lib.eegldm_groupnorm_fwd(c.h, G_.ptr(xd), Cin, G_.ptr(gd), G_.ptr(bed), G_.ptr(a), Cin, G_.ptr(st), B, L, Cin, G, 1e-6, 1, 0, None, 0, dt))
        embd = emb.to(G_.DEV).contiguous() if extras else None
        resd = G_.nlc(res, dt) if extras else None
        y2 = torch.empty(B * L, Cout, device=G_.DEV, dtype=torch.bfloat16)
        G_.check(G_.lib.eegldm_conv1d_fwd(c.h, G_.ptr(a), Cin, G_.ptr(wd), G_.ptr(bd), G_.ptr(y2), Cout, B, L, Cin, Cout, 3, 1, 1, 1,
                                          G_.ptr(embd), Cout if extras else 0, G_.ptr(resd), Cout if extras else 0, dt))
        # one launch: the conv normalises its operand tile in LDS (statistics from the pass above)
        y1 = torch.empty_like(y2)
        G_.check(G_.lib.eegldm_conv1d_fwd_gn(c.h, G_.ptr(xd), Cin, G_.ptr(wd), G_.ptr(bd), G_.ptr(gd), G_.ptr(bed), G_.ptr(st), G, 1, G_.ptr(y1), Cout,
                                             B, L, Cin, Cout, G_.ptr(embd), Cout if extras else 0, G_.ptr(resd), Cout if extras else 0, dt))
        torch.cuda.synchronize()
    finally:
        G_.lib.eegldm_conv1d_forget_kblocked(c.h, G_.ptr(wd))
    # same arithmetic: both round the normalised operand to bf16 once, then the same MFMA product (different kernels: summation order differs)
    assert G_.rel_l2(y1, y2) < 3e-3, G_.rel_l2(y1, y2)
    # torch fp32 on the bf16-rounded inputs
    xq, wq = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv1d(F.silu(F.group_norm(xq, G, gamma, beta, eps=1e-6)), wq, b, padding=1)
    if extras:
        ref = ref + emb[:, :, None] + res.bfloat16().float()
    assert G_.rel_l2(G_.ncl(y1, B, L), ref) < 1.2e-2, G_.rel_l2(G_.ncl(y1, B, L), ref)
