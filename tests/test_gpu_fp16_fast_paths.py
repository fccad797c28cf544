"""-m gpu: the specialised kernels instantiated for IEEE-half operands (EEGLDM_F16, the reference's autocast dtype, training.py:423):
weight-stationary level-0 convs (conv_ws.hip), the few-row conv of one-window sampling with its GroupNorm-on-load chain (conv_skinny.hip),
the fused frozen encoder (enc_fused.hip) and the pipelined GroupNorm backward (norm.hip gn_bwd_pipe_kernel).  Each is held against
(a) torch's fp32 op on fp16-rounded operands under the fp16 tolerances of tests/gpu_util.py and (b) the GENERAL fp16 kernel it replaces
(switched in process: conftest.env_switches -> eegldm_debug_reload_env), which rounds at the same points, so the two may differ only by
accumulation order.  tests/test_gpu_fp16.py covers the general, big-tile and fused-attention kernels."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import normal  # noqa: E402


def h(t):
    return t.half().float()


def _ulp_close(a, b, name, frac=2e-3):
    """same rounding points, different summation order: isolated last-place flips of the fp16 result, nothing larger"""
    a, b = a.float(), b.float()
    tol = 2.0 ** -9 * torch.maximum(a.abs(), b.abs()) + 1e-3
    bad = (a - b).abs() > tol
    assert int(bad.sum()) == 0, f"{name}: {int(bad.sum())} elements beyond two fp16 ulps, worst {float((a - b).abs().max()):.3e}"
    assert float((a != b).float().mean()) <= max(frac, 0.05), f"{name}: too many flips"


@pytest.mark.parametrize("case", [(32, 768, 128, 128), (24, 768, 128, 256), (32, 768, 256, 128)])
def test_weight_stationary_conv_fp16(case, env_switches):
    """conv3_ws_kernel<*, f16_t>: forward (bias + embedding row + residual) for 128 -> N, data gradient (transposed weights) for N -> 128"""
    import gpu_util as G
    c = G.ctx(); dt = G.F16
    B, L, Cin, Cout = case
    x = h(torch.from_numpy(normal((B, Cin, L), seed=1))); w = h(torch.from_numpy(normal((Cout, Cin, 3), seed=2)) / math.sqrt(Cin * 3))
    b = torch.from_numpy(normal((Cout,), seed=3)); e = torch.from_numpy(normal((B, Cout), seed=5)); r = h(torch.from_numpy(normal((B, Cout, L), seed=6)))
    xd, wd, bd, ed, rd = G.nlc(x, dt), G.pack_w(w, dt), b.to(G.DEV), e.to(G.DEV), G.nlc(r, dt)
    dy = h(torch.from_numpy(normal((B, Cout, L), seed=7))); dyd = G.nlc(dy, dt)
    out = {}
    for mode in ("ws", "general"):
        env_switches(EEGLDM_NO_CONV_WS=None if mode == "ws" else "1")
        yd = torch.full((B * L, Cout), float("nan"), device=G.DEV, dtype=torch.float16)
        G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, 3, 1, 1, 1, G.ptr(ed), Cout, G.ptr(rd), Cout, dt))
        dxd = torch.full((B * L, Cin), float("nan"), device=G.DEV, dtype=torch.float16)
        G.check(G.lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dxd), Cin, B, L, Cin, Cout, 3, 1, 1, 1, None, 0, dt))
        torch.cuda.synchronize(); out[mode] = (yd, dxd)
    if Cin == 128: G.assert_close(G.ncl(out["ws"][0], B, L), F.conv1d(x, w, b, padding=1) + e[:, :, None] + r, **G.TOL[dt], name="y")
    if Cout == 128: G.assert_close(G.ncl(out["ws"][1], B, L), F.conv_transpose1d(dy, w, padding=1), **G.GTOL[dt], name="dx")
    _ulp_close(out["ws"][0], out["general"][0], "y vs general kernel"); _ulp_close(out["ws"][1], out["general"][1], "dx vs general kernel")


SK_CASES = [  # B, L, Cin, Cout, K, rowvec, resid
    (1, 768, 128, 128, 3, 1, 0), (1, 384, 256, 256, 3, 1, 0), (1, 192, 512, 512, 3, 0, 1), (1, 192, 1024, 512, 3, 1, 0),
    (1, 192, 512, 1536, 1, 0, 0), (1, 192, 512, 512, 1, 0, 1), (2, 72, 96, 48, 3, 1, 1), (3, 40, 160, 48, 1, 0, 1), (5, 192, 512, 512, 3, 1, 1),
]


def test_few_row_conv_fp16(env_switches):
    import gpu_util as G
    c = G.ctx(); dt = G.F16
    res = {}
    for mode in ("few_row", "general"):
        env_switches(EEGLDM_NO_CONV_SKINNY=None if mode == "few_row" else "1")
        outs = []
        for ci, (B, L, Cin, Cout, K, rv, rs) in enumerate(SK_CASES):
            x = h(torch.from_numpy(normal((B, Cin, L), seed=10 + ci))); w = h(torch.from_numpy(normal((Cout, Cin, K), seed=40 + ci)) / math.sqrt(Cin * K))
            b = torch.from_numpy(normal((Cout,), seed=70 + ci))
            e = torch.from_numpy(normal((B, Cout), seed=100 + ci)) if rv else None
            r = h(torch.from_numpy(normal((B, Cout, L), seed=130 + ci))) if rs else None
            ref = F.conv1d(x, w, b, padding=K // 2)
            if rv: ref = ref + e[:, :, None]
            if rs: ref = ref + r
            xd, wd, bd = G.nlc(x, dt), G.pack_w(w, dt), b.to(G.DEV)
            ed = e.to(G.DEV) if rv else None; rd = G.nlc(r, dt) if rs else None
            yd = torch.full((B * L, Cout), float("nan"), device=G.DEV, dtype=torch.float16)
            G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, K, 1, K // 2, K // 2,
                                            G.ptr(ed) if rv else None, Cout if rv else 0, G.ptr(rd) if rs else None, Cout if rs else 0, dt))
            G.assert_close(G.ncl(yd, B, L), ref, **G.TOL[dt], name="%s case %d %s" % (mode, ci, SK_CASES[ci]))
            outs.append(yd)
        res[mode] = outs
    for ci in range(len(SK_CASES)):
        _ulp_close(res["few_row"][ci], res["general"][ci], "case %d" % ci)


UCFG = dict(in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)


def _seeded_unet(L, dtype, seed=0):
    from eegldm.models import UNetModel
    torch.manual_seed(seed)                 # the module's default init draws from torch's global generator
    net = UNetModel(image_size=L, **UCFG, dtype=dtype)
    g = torch.Generator().manual_seed(seed); sd = net.state_dict()
    net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()})
    return net, {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}


def test_one_window_eval_forward_fp16_fused_chain_against_general_kernels_and_fp32(env_switches):
    """B = 1 eval forward in fp16: GroupNorm applied on the few-row conv's operand load, skip convs as a K extension -- against the same
    engine on the general kernels and against the fp32 engine (fp16 storage error only)."""
    L = 768
    net16, w = _seeded_unet(L, "float16")
    from eegldm.models import UNetModel
    net32 = UNetModel(image_size=L, dtype="float32", **UCFG); net32.load_state_dict(w)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 1, L, generator=g).cuda(); t = torch.tensor([417], device="cuda")
    net16.eval(); net32.eval()
    with torch.no_grad():
        y32 = net32(x, timesteps=t).float().clone()
        env_switches(EEGLDM_NO_CONV_SKINNY=None, EEGLDM_NO_EVAL_GN_FUSE=None)
        y_f = net16(x, timesteps=t).float().clone()
        env_switches(EEGLDM_NO_CONV_SKINNY="1", EEGLDM_NO_EVAL_GN_FUSE="1")
        y_g = net16(x, timesteps=t).float().clone()
    sc = float(y32.abs().max())
    assert torch.isfinite(y_f).all()
    e_f, e_g = float((y_f - y32).abs().max()) / sc, float((y_g - y32).abs().max()) / sc
    assert e_g < 2e-2 and e_f < max(2e-2, 2.0 * e_g), (e_f, e_g)                 # the fused chain is as close to fp32 as the layer-by-layer fp16 forward
    assert float((y_f - y_g).abs().max()) / sc < 2e-2


def _mk_ae(channels, dtype, seed=0):
    from eegldm.models import AutoencoderKL
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=channels, latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False] * len(channels), dtype=dtype)
    g = torch.Generator().manual_seed(seed); new = {}
    for k, v in ae.state_dict().items():
        if v.dim() == 1 and k.endswith(".weight"): new[k] = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        elif v.dim() == 1: new[k] = 0.1 * torch.randn(v.shape, generator=g)
        else: new[k] = torch.randn(v.shape, generator=g) / (v.shape[1] * v.shape[2]) ** 0.5
    ae.load_state_dict(new)
    return ae, new


@pytest.mark.parametrize("channels,B,L", [([32, 32, 64], 3, 1024), ([64, 64, 64], 2, 3072)])
def test_fused_frozen_encoder_fp16(channels, B, L):
    ae, w = _mk_ae(channels, "float16")
    ae32, _ = _mk_ae(channels, "float32"); ae32.load_state_dict(w)
    x = torch.randn(B, 1, L, generator=torch.Generator().manual_seed(5)).cuda()
    os.environ["EEGLDM_AEKL_NO_FUSED_ENC"] = "1"
    try: mu0, sg0 = ae.encode(x)
    finally: del os.environ["EEGLDM_AEKL_NO_FUSED_ENC"]
    mu1, sg1 = ae.encode(x); mu32, sg32 = ae32.encode(x)
    torch.cuda.synchronize()
    sc = float(mu32.abs().max())
    assert torch.isfinite(mu1).all() and torch.isfinite(sg1).all() and sc > 0
    e1, e0 = float((mu1.float() - mu32).abs().max()) / sc, float((mu0.float() - mu32).abs().max()) / sc
    assert e0 < 1e-2 and e1 < max(1e-2, 2.0 * e0), (e1, e0)                      # fp16 storage: 3 more mantissa bits than the bf16 bound of test_gpu_fused_encoder.py
    assert float((mu1.float() - mu0.float()).abs().max()) <= 5e-3 * sc
    assert float((sg1.float() - sg0.float()).abs().max()) <= 5e-3 * float(sg32.abs().max())


PIPE_CASES = [(24, 384, 256, 1, 0), (24, 192, 512, 1, 0), (16, 768, 128, 1, 0), (16, 192, 1024, 0, 0), (32, 384, 512, 1, 1)]


@pytest.mark.parametrize("case", PIPE_CASES)
def test_pipelined_groupnorm_backward_fp16(case, env_switches):
    import gpu_util as G
    c = G.ctx(); dt = G.F16
    B, L, Cc, silu, has_r = case
    x = h(torch.from_numpy(normal((B, Cc, L), seed=11)) * 1.5 + 0.7).requires_grad_(True)
    ga = (1 + 0.1 * torch.from_numpy(normal((Cc,), seed=12))).requires_grad_(True); be = (0.1 * torch.from_numpy(normal((Cc,), seed=13))).requires_grad_(True)
    y = F.group_norm(x, 32, ga, be, eps=1e-6)
    if silu: y = F.silu(y)
    dy = h(torch.from_numpy(normal((B, Cc, L), seed=14))); dxr = h(torch.from_numpy(normal((B, Cc, L), seed=15))) if has_r else None
    y.backward(dy)
    want_dx = x.grad + (dxr if has_r else 0)
    xd = G.nlc(x.detach(), dt); gad, bed = ga.detach().to(G.DEV), be.detach().to(G.DEV)
    yd = torch.empty_like(xd); st = torch.empty(B * 32 * 2, device=G.DEV)
    G.check(G.lib.eegldm_groupnorm_fwd(c.h, G.ptr(xd), Cc, G.ptr(gad), G.ptr(bed), G.ptr(yd), Cc, G.ptr(st), B, L, Cc, 32, 1e-6, silu, 0, None, 0, dt))
    dyd = G.nlc(dy, dt); dxrd = G.nlc(dxr, dt) if has_r else None
    res = {}
    for mode in ("pipe", "resident"):
        env_switches(EEGLDM_GN_NO_PIPE=None if mode == "pipe" else "1", EEGLDM_GN_PIPE_ADDEND="1" if (mode == "pipe" and has_r) else None,
                     EEGLDM_GN_PIPE_MIN_SLABS="1" if mode == "pipe" else None)
        dxd = torch.empty_like(xd); dga = torch.zeros(Cc, device=G.DEV); dbe = torch.zeros(Cc, device=G.DEV)
        G.check(G.lib.eegldm_groupnorm_bwd(c.h, G.ptr(xd), Cc, G.ptr(gad), G.ptr(bed), G.ptr(st), G.ptr(dyd), Cc, G.ptr(dxd), Cc, G.ptr(dga), G.ptr(dbe),
                                           B, L, Cc, 32, silu, 0, G.ptr(dxrd) if has_r else None, Cc, dt))
        torch.cuda.synchronize(); res[mode] = (dxd, dga, dbe)
    G.assert_close(G.ncl(res["pipe"][0], B, L), want_dx, **G.GTOL[dt], name="dx")
    for got, want, name in ((res["pipe"][1], ga.grad, "dgamma"), (res["pipe"][2], be.grad, "dbeta")):
        G.assert_close(got.cpu(), want, rtol=G.GTOL[dt]["rtol"], atol=G.GTOL[dt]["atol"] * max(1.0, float(want.abs().max())), name=name)
    _ulp_close(res["pipe"][0], res["resident"][0], "dx vs resident kernel")
    assert float((res["pipe"][1] - res["resident"][1]).abs().max()) <= 2e-5 * max(1.0, float(res["resident"][1].abs().max()))
