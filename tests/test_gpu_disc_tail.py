"""-m gpu: the fused discriminator tail (csrc/disc_tail.hip: last BatchNorm + LeakyReLU + one-channel final conv in one forward pass and two
backward passes over the last hidden conv's output) and the fused head (first layer's LeakyReLU backward + weight / bias / input gradients of the
one-input-channel conv in one pass over the gradient of its activated output; EEGLDM_DISC_NO_FUSED_HEAD=1) against the layer-by-layer path of the same library (EEGLDM_DISC_NO_FUSED_TAIL=1) and
against the oracle.  The oracle / reference-twin comparisons of tests/test_gpu_aekl.py run on the fused path by default; this file pins the two
paths to each other on shapes that exercise every lane mapping (C = 512: one wave per row; C = 128: 16 / 32 lanes per row), rows-per-block
boundaries that fall inside samples, a length that is not a multiple of the block's row count, and the no-parameter-gradient backward the
generator step uses (MONAI PatchDiscriminator, config/config_aekl_eeg.yaml:30-40; train_autoencoderkl.py:213-228)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from param_gen import gen_param, normal  # noqa: E402
from gpu_util import rel_l2  # noqa: E402


def _run(cfg, dtype, x, dy, sd, param_grads=True):
    from eegldm.models import PatchDiscriminator
    net = PatchDiscriminator(**cfg, dtype=dtype)
    net.load_state_dict(sd)
    feats = net(x)
    net.zero_grad()
    dx = net.backward(dy, need_dx=True, in_shape=tuple(x.shape), param_grads=param_grads)
    return [f.clone() for f in feats], dx.clone(), {k: v.clone() for k, v in net.grad_dict().items()}, {k: v.clone() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("dtype", ["float32", "bfloat16", "float16"])
@pytest.mark.parametrize("nch,B,L", [(64, 3, 200), (16, 5, 136), (64, 2, 3000)])
def test_fused_tail_matches_layerwise(env_switches, dtype, nch, B, L):
    from eegldm.models import PatchDiscriminator
    cfg = dict(spatial_dims=1, num_layers_d=3, num_channels=nch, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
    probe = PatchDiscriminator(**cfg, dtype=dtype)
    sd = {}
    for k, (_o, _n, shape) in list(probe.entries.items()) + list(probe.buf_entries.items()):
        v = torch.from_numpy(gen_param(31, k, shape))
        sd[k] = v * 2.0 if k.endswith("conv.weight") else v
    del probe
    x = torch.from_numpy(normal((B, 1, L), seed=5))
    Lo = L
    for _ in range(3):
        Lo = (Lo + 2 - 3) // 2 + 1
    dy = torch.from_numpy(normal((B, 1, Lo), seed=6))
    f1, dx1, g1, s1 = _run(cfg, dtype, x, dy, sd)
    env_switches(EEGLDM_DISC_NO_FUSED_TAIL="1", EEGLDM_DISC_NO_FUSED_HEAD="1")
    f0, dx0, g0, s0 = _run(cfg, dtype, x, dy, sd)
    f32 = dtype == "float32"
    # fp32: the same arithmetic up to summation order.  16-bit: the layer-wise path rounds the activation, the logits and the final conv's data
    # gradient to the storage type, the fused path keeps them in fp32 -- they differ by that rounding (one part in 2^8 / 2^11 per tensor)
    tol = 2e-5 if f32 else (3e-2 if dtype == "bfloat16" else 4e-3)
    assert len(f1) == len(f0) == 5
    for i, (a, b) in enumerate(zip(f1, f0)):
        assert a.shape == b.shape and rel_l2(a, b) < tol, (i, rel_l2(a, b))
    assert rel_l2(dx1, dx0) < (1e-4 if f32 else 3 * tol), rel_l2(dx1, dx0)
    gscale = max(float(v.double().norm()) for v in g0.values())
    for k in g0:
        err = float((g1[k].double() - g0[k].double()).norm()) / (float(g0[k].double().norm()) + 1e-3 * gscale)
        assert err < (1e-4 if f32 else 3 * tol), (k, err)
    for k in s0:
        if "running" in k or "num_batches" in k:
            assert rel_l2(s1[k].float(), s0[k].float()) < (1e-5 if f32 else 2e-2), k


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fused_tail_vs_oracle_and_generator_backward(dtype):
    """Fused path against the CPU oracle (logits, input gradient, the tail's own parameter gradients), and the backward without parameter
    gradients (the generator's pass through D, train_autoencoderkl.py:213-215) must give the same input gradient and leave the gradient buffer alone."""
    from eegldm.models import PatchDiscriminator
    from oracle import aekl as A
    cfg = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
    B, L = 3, 264
    sd = {}
    for k, s in A.disc_param_shapes(cfg).items():
        v = torch.from_numpy(gen_param(21, k, s))
        sd[k] = (v * 2.0 if k.endswith("conv.weight") else v)
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    x = torch.from_numpy(normal((B, 1, L), seed=8)).requires_grad_(True)
    logits = A.disc_forward(sdr, cfg, x, True, {})[-1]
    dy = torch.from_numpy(normal(tuple(logits.shape), seed=9))
    (logits * dy).sum().backward()
    net = PatchDiscriminator(**cfg, dtype=dtype); net.load_state_dict(sd)
    out = net(x.detach())[-1]
    f32 = dtype == "float32"
    assert rel_l2(out, logits.detach()) < (2e-5 if f32 else 5e-2)
    net.zero_grad()
    dx = net.backward(dy, need_dx=True, in_shape=tuple(x.shape))
    assert rel_l2(dx, x.grad) < (2e-4 if f32 else 0.15), rel_l2(dx, x.grad)
    g = net.grad_dict()
    for k in ("final_conv.conv.weight", "final_conv.conv.bias", "2.adn.N.weight", "2.adn.N.bias", "initial_conv.conv.weight", "initial_conv.conv.bias"):
        # bf16: y (the last hidden conv's output) is stored rounded, so a share of the LeakyReLU masks near z = 0 flips against the fp32 oracle --
        # the BatchNorm shift gradient (a plain sum of masked terms) is the most exposed: 0.08 measured, the layer-wise path the same
        assert rel_l2(g[k], sdr[k].grad) < (2e-4 if f32 else 0.12), (k, rel_l2(g[k], sdr[k].grad))
    # generator pass: same forward, backward without parameter gradients
    net.zero_grad(); net(x.detach())
    dx2 = net.backward(dy, need_dx=True, in_shape=tuple(x.shape), param_grads=False)
    assert rel_l2(dx2, dx) < 1e-6 if f32 else rel_l2(dx2, dx) < 1e-3
    assert float(net.flat_grad.abs().max()) == 0.0


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("B,L,bias", [(32, 1536, False), (24, 1408, True)])
def test_stride2_conv_on_the_weight_stationary_kernel(dtype, B, L, bias):
    """Conv1d(64 -> 128, k 3, stride 2, padding 1) -- the discriminator's second layer -- forward and data gradient as stride-1 convs of
    conv3_ws_kernel over pairs of rows (eegldm_conv1d_pack_stride2) against torch fp32 on the rounded operands, and bit-for-bit layout checks
    against the general kernel (same entry points without the registration)."""
    import math
    import torch.nn.functional as F
    import gpu_util as G
    dt = G.BF16 if dtype == "bfloat16" else G.F16
    tdt = G.TDT[dt]
    Cin, Cout = 64, 128
    x = torch.from_numpy(normal((B, Cin, L), seed=1)).to(tdt).float()
    w = (torch.from_numpy(normal((Cout, Cin, 3), seed=2)) / math.sqrt(3 * Cin)).to(tdt).float()
    b = torch.from_numpy(normal((Cout,), seed=3)) if bias else None
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv1d(F.pad(xr, (1, 1)), w, b, stride=2)
    Lo = y_ref.shape[-1]
    dy = torch.from_numpy(normal((B, Cout, Lo), seed=4)).to(tdt).float()
    y_ref.backward(dy)
    c = G.ctx()
    xd, wd, dyd = G.nlc(x, dt), G.pack_w(w, dt), G.nlc(dy, dt)
    bd = b.to(G.DEV) if bias else None
    wf = torch.empty(3 * 128 * 128, device=G.DEV, dtype=tdt); wdg = torch.empty_like(wf)

    def run():
        y = torch.empty(B * Lo, Cout, device=G.DEV, dtype=tdt); dx = torch.empty(B * L, Cin, device=G.DEV, dtype=tdt)
        G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(y), Cout, B, L, Cin, Cout, 3, 2, 1, 1, None, 0, None, 0, dt))
        G.check(G.lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dx), Cin, B, L, Cin, Cout, 3, 2, 1, 1, None, 0, dt))
        torch.cuda.synchronize()
        return y, dx
    y0, dx0 = run()                                   # general implicit GEMM
    G.check(G.lib.eegldm_conv1d_pack_stride2(c.h, G.ptr(wd), G.ptr(wf), G.ptr(wdg), Cout, Cin, dt))
    try:
        y1, dx1 = run()                               # weight-stationary route (L / 2 a multiple of 64, >= 16 384 output rows)
    finally:
        G.lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd))
    tol = G.TOL[dt]; gtol = G.GTOL[dt]
    G.assert_close(G.ncl(y1, B, Lo), y_ref, **tol, name="y (weight-stationary route)")
    G.assert_close(G.ncl(dx1, B, L), xr.grad, **gtol, name="dx (weight-stationary route)")
    assert rel_l2(y1, y0) < 1e-2 and rel_l2(dx1, dx0) < 1e-2      # same product, other summation order
    assert not torch.equal(y1, y0) or B * Lo < 16384              # (the two routes are different kernels: identical bits would mean the registration was not used)


@pytest.mark.parametrize("dtype", ["bfloat16", "float16"])
@pytest.mark.parametrize("B,L,bias", [(32, 768, False), (12, 1472, True), (3, 128, True)])
def test_stride2_conv_128_256_on_the_paired_row_kernel(dtype, B, L, bias, env_switches):
    """Conv1d(128 -> 256, k 3, stride 2, padding 1) -- the discriminator's third layer -- forward and data gradient on conv3_ws2_kernel (paired rows,
    weight-stationary, 32-row tiles) against torch fp32 on the rounded operands and against the general kernel (EEGLDM_NO_CONV_WS=1).  The last
    shape is below the kernel's row threshold and stays on the general path in both runs."""
    import math
    import torch.nn.functional as F
    import gpu_util as G
    dt = G.BF16 if dtype == "bfloat16" else G.F16
    tdt = G.TDT[dt]
    Cin, Cout = 128, 256
    x = torch.from_numpy(normal((B, Cin, L), seed=1)).to(tdt).float()
    w = (torch.from_numpy(normal((Cout, Cin, 3), seed=2)) / math.sqrt(3 * Cin)).to(tdt).float()
    b = torch.from_numpy(normal((Cout,), seed=3)) if bias else None
    xr = x.clone().requires_grad_(True)
    y_ref = F.conv1d(F.pad(xr, (1, 1)), w, b, stride=2)
    Lo = y_ref.shape[-1]
    dy = torch.from_numpy(normal((B, Cout, Lo), seed=4)).to(tdt).float()
    y_ref.backward(dy)
    c = G.ctx()
    xd, wd, dyd = G.nlc(x, dt), G.pack_w(w, dt), G.nlc(dy, dt)
    bd = b.to(G.DEV) if bias else None

    def run():
        y = torch.empty(B * Lo, Cout, device=G.DEV, dtype=tdt); dx = torch.empty(B * L, Cin, device=G.DEV, dtype=tdt)
        G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(y), Cout, B, L, Cin, Cout, 3, 2, 1, 1, None, 0, None, 0, dt))
        G.check(G.lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dx), Cin, B, L, Cin, Cout, 3, 2, 1, 1, None, 0, dt))
        torch.cuda.synchronize()
        return y, dx
    y1, dx1 = run()
    env_switches(EEGLDM_NO_CONV_WS="1")
    y0, dx0 = run()
    G.assert_close(G.ncl(y1, B, Lo), y_ref, **G.TOL[dt], name="y (paired-row kernel)")
    G.assert_close(G.ncl(dx1, B, L), xr.grad, **G.GTOL[dt], name="dx (paired-row kernel)")
    assert rel_l2(y1, y0) < 1e-2 and rel_l2(dx1, dx0) < 1e-2
