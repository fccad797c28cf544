"""-m gpu: stand-alone AvgPool1d(2, 2) / nearest x 2 behind the C ABI (eegldm_avgpool2_ / nearest2_{fwd,bwd}) against torch's
nn.AvgPool1d(2, 2) and F.interpolate(scale_factor=2, mode="nearest") and their autograd -- the ops of Downsample / Upsample with
use_conv = False (/root/reference/src/models/unet.py:177-224) that a ResBlock applies to h and x when up / down is set (unet.py:308-313).
The executors fuse them into the GroupNorm kernels; these entries are the same arithmetic at primitive granularity, in the three storage
types, on dense tensors and on column views of wider buffers (leading dimension > C)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import normal  # noqa: E402


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("shape,pad", [((3, 128, 768), 0), ((2, 256, 384), 0), ((2, 6, 50), 0), ((2, 40, 96), 24)])
def test_avgpool2_and_nearest2_forward_backward(shape, pad, dtype_name):
    import gpu_util as G
    c = G.ctx(); dt = {"float32": G.F32, "bfloat16": G.BF16, "float16": G.F16}[dtype_name]
    td = G.TDT[dt]
    B, C, L = shape
    rnd = lambda t: t.to(td).float()
    x = rnd(torch.from_numpy(normal((B, C, L), seed=1))).requires_grad_(True)
    ld = C + pad
    # ---- AvgPool1d(2, 2)
    y = F.avg_pool1d(x, 2, 2)
    dy = rnd(torch.from_numpy(normal(tuple(y.shape), seed=2)))
    y.backward(dy)
    xd = G.nlc(x.detach(), dt, ld=ld); yd = torch.full((B * (L // 2), ld), float("nan"), device=G.DEV, dtype=td)
    G.check(G.lib.eegldm_avgpool2_fwd(c.h, G.ptr(xd), ld, G.ptr(yd), ld, B, L, C, dt))
    got = G.ncl(yd[:, :C].float().contiguous(), B, L // 2)
    assert torch.equal(got.cpu(), rnd(y.detach())), "avgpool2 forward: one rounding of the exact fp32 mean"
    dyd = G.nlc(dy, dt, ld=ld); dxd = torch.full((B * L, ld), float("nan"), device=G.DEV, dtype=td)
    G.check(G.lib.eegldm_avgpool2_bwd(c.h, G.ptr(dyd), ld, G.ptr(dxd), ld, B, L, C, dt))
    assert torch.equal(G.ncl(dxd[:, :C].float().contiguous(), B, L).cpu(), rnd(x.grad)), "avgpool2 backward"
    if pad: assert torch.isnan(yd[:, C:].float()).all() and torch.isnan(dxd[:, C:].float()).all(), "columns beyond C are not touched"
    # ---- nearest x 2
    x2 = rnd(torch.from_numpy(normal((B, C, L), seed=3))).requires_grad_(True)
    u = F.interpolate(x2, scale_factor=2, mode="nearest")
    du = rnd(torch.from_numpy(normal(tuple(u.shape), seed=4)))
    u.backward(du)
    x2d = G.nlc(x2.detach(), dt, ld=ld); ud = torch.full((B * 2 * L, ld), float("nan"), device=G.DEV, dtype=td)
    G.check(G.lib.eegldm_nearest2_fwd(c.h, G.ptr(x2d), ld, G.ptr(ud), ld, B, L, C, dt))
    assert torch.equal(G.ncl(ud[:, :C].float().contiguous(), B, 2 * L).cpu(), u.detach()), "nearest x 2 forward is a copy"
    dud = G.nlc(du, dt, ld=ld); dx2d = torch.full((B * L, ld), float("nan"), device=G.DEV, dtype=td)
    G.check(G.lib.eegldm_nearest2_bwd(c.h, G.ptr(dud), ld, G.ptr(dx2d), ld, B, L, C, dt))
    assert torch.equal(G.ncl(dx2d[:, :C].float().contiguous(), B, L).cpu(), rnd(x2.grad)), "nearest x 2 backward: one rounding of the exact pair sum"


def test_odd_length_is_refused():
    import gpu_util as G
    c = G.ctx()
    x = torch.zeros(2 * 7, 8, device=G.DEV); y = torch.zeros(2 * 3, 8, device=G.DEV)
    assert G.lib.eegldm_avgpool2_fwd(c.h, G.ptr(x), 8, G.ptr(y), 8, 2, 7, 8, G.F32) != 0
