"""-m gpu: GroupNorm forward as ONE streaming pass from the moments the producing conv's epilogue left (round 5: gemm_big.hip big_qstats ->
norm.hip gn_apply_q_kernel; reference ops: `normalization(channels)` + SiLU behind a conv, /root/reference/src/models/unet.py:71-74,
261-263, 287-291, and behind the concatenation of unet.py:553).
* primitive level: eegldm_conv1d_fwd_qstats against fp64 sums of the conv output (torch fp32 conv on the bf16-rounded operands), with bias,
  embedding row and residual, 3-tap and 1 x 1, the fused skip-connection tail; eegldm_groupnorm_fwd_qstats against eegldm_groupnorm_fwd on
  the same tensor (y and the (mean, rstd) tape), single producer and a concatenated input with two producers whose group straddles the seam
  (C = 768 = 512 + 256: 24-channel groups), group widths 8 / 16 / 24 / 32;
* model level: UNet forward / backward with the path (EEGLDM_GN_QSTATS=1: opt-in, it measured no faster than the resident kernel) and without agree within bf16 noise, and the path is really
  taken (fewer resident GroupNorm launches is not observable from here, so the test reads the statistics tape through the backward: equal
  gradients within the bound mean the (mean, rstd) pairs the streaming kernel wrote are the ones the backward needs)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import normal  # noqa: E402


def _conv_q(G, c, x, w, b, e, r, K, x2=None, w2=None, b2=None):
    import ctypes as C
    dt = G.BF16
    B, Cin, L = x.shape; Cout = w.shape[0]
    xd, wd, bd = G.nlc(x, dt), G.pack_w(w, dt), b.to(G.DEV)
    ed = e.to(G.DEV) if e is not None else None
    rd = G.nlc(r, dt) if r is not None else None
    yd = torch.full((B * L, Cout), float("nan"), device=G.DEV, dtype=torch.bfloat16)
    qs = torch.zeros(B, Cout // 4, 2, device=G.DEV, dtype=torch.float64)
    filled = C.c_int(0)
    wk = torch.empty_like(wd)
    if K == 3:
        G.check(G.lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(wd), G.ptr(wk), Cout, Cin, dt))
    try:
        G.check(G.lib.eegldm_conv1d_fwd_qstats(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, K, 1, K // 2, K // 2,
                                               G.ptr(ed) if e is not None else None, Cout if e is not None else 0,
                                               G.ptr(rd) if r is not None else None, Cout if r is not None else 0, dt, G.ptr(qs), C.byref(filled)))
        torch.cuda.synchronize()
    finally:
        if K == 3: G.check(G.lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd)))
    return yd, qs, filled.value


@pytest.mark.parametrize("case", [(4, 192, 512, 512, 3, 1, 0), (2, 384, 768, 256, 3, 1, 1), (3, 192, 512, 512, 1, 0, 1), (2, 768, 256, 256, 3, 0, 1)])
def test_conv_epilogue_moments(case, env_switches):
    import gpu_util as G
    c = G.ctx()
    B, L, Cin, Cout, K, rv, rs = case
    x = torch.from_numpy(normal((B, Cin, L), seed=10)).bfloat16().float()
    w = (torch.from_numpy(normal((Cout, Cin, K), seed=40)) / math.sqrt(Cin * K)).bfloat16().float()
    b = torch.from_numpy(normal((Cout,), seed=70)) + 0.5                    # a mean well away from zero
    e = torch.from_numpy(normal((B, Cout), seed=100)) if rv else None
    r = torch.from_numpy(normal((B, Cout, L), seed=130)).bfloat16().float() if rs else None
    ref = F.conv1d(x, w, b, padding=K // 2)
    if rv: ref = ref + e[:, :, None]
    if rs: ref = ref + r
    env_switches(EEGLDM_GEMM_BIG_MIN_TILES="1", EEGLDM_NO_CONV_SKINNY="1")       # small problems: force the big tile (the few-row kernel would take them)
    y, qs, filled = _conv_q(G, c, x, w, b, e, r, K)
    assert filled == 1
    G.assert_close(G.ncl(y, B, L), ref, **G.TOL[G.BF16], name="y")
    rq = ref.double().reshape(B, Cout // 4, 4, L)
    s1 = rq.sum(dim=(2, 3)); s2 = (rq * rq).sum(dim=(2, 3))
    got = qs.cpu()
    n = 4 * L
    # fp32 partial sums of <= 96 x 4 values inside a wave, fp64 across waves / tiles: relative error ~1e-6 of sum |v|
    assert float((got[..., 0] - s1).abs().max()) < 2e-5 * n, float((got[..., 0] - s1).abs().max())
    assert float(((got[..., 1] - s2).abs() / s2).max()) < 2e-5
    # a second call ADDS
    y2, qs2, _ = _conv_q(G, c, x, w, b, e, r, K)
    env_switches(EEGLDM_NO_GEMM_BIG="1")
    _, qs0, filled0 = _conv_q(G, c, x, w, b, e, r, K)
    assert filled0 == 0 and float(qs0.abs().max()) == 0.0                 # any other kernel: untouched, and says so


@pytest.mark.parametrize("case", [(3, 192, 512, 0, 32, 1), (2, 384, 256, 0, 32, 1), (2, 192, 512, 256, 32, 1), (2, 192, 512, 512, 32, 0), (2, 384, 256, 128, 32, 1), (2, 768, 256, 0, 8, 1)])
def test_groupnorm_forward_from_producer_moments(case):
    """y and the (mean, rstd) tape against the resident one-pass kernel on the same stored tensor; the moments are computed here in fp64 from
    the STORED bf16 tensor (what the epilogue sees is the value before rounding: covered at model level)."""
    import gpu_util as G
    c = G.ctx(); dt = G.BF16
    B, L, Ca, Cb, Gn, silu = case
    C = Ca + Cb
    x = (torch.from_numpy(normal((B, C, L), seed=5)) * 1.7 + 0.8).bfloat16().float()
    gamma = 1.0 + 0.1 * torch.from_numpy(normal((C,), seed=6)); beta = 0.1 * torch.from_numpy(normal((C,), seed=7))
    xd = G.nlc(x, dt); gd, bd = gamma.to(G.DEV), beta.to(G.DEV)
    y0 = torch.empty_like(xd); st0 = torch.empty(B, Gn, 2, device=G.DEV)
    G.check(G.lib.eegldm_groupnorm_fwd(c.h, G.ptr(xd), C, G.ptr(gd), G.ptr(bd), G.ptr(y0), C, G.ptr(st0), B, L, C, Gn, 1e-6, silu, 0, None, 0, dt))
    xq = x.double().reshape(B, C // 4, 4, L)
    q = torch.stack([xq.sum(dim=(2, 3)), (xq * xq).sum(dim=(2, 3))], dim=-1)           # [B][C/4][2]
    qa = q[:, :Ca // 4].contiguous().to(G.DEV); qb = q[:, Ca // 4:].contiguous().to(G.DEV) if Cb else None
    y1 = torch.full_like(xd, float("nan")); st1 = torch.full_like(st0, float("nan"))
    G.check(G.lib.eegldm_groupnorm_fwd_qstats(c.h, G.ptr(xd), C, G.ptr(gd), G.ptr(bd), G.ptr(y1), C, G.ptr(st1), B, L, C, Gn, 1e-6, silu,
                                              G.ptr(qa), Ca // 4, G.ptr(qb) if Cb else None, Cb // 4, dt))
    torch.cuda.synchronize()
    assert torch.allclose(st1[..., 0], st0[..., 0], rtol=0, atol=2e-6 * float(st0[..., 0].abs().max() + 1)), float((st1[..., 0] - st0[..., 0]).abs().max())
    assert torch.allclose(st1[..., 1], st0[..., 1], rtol=2e-6, atol=0), float((st1[..., 1] / st0[..., 1] - 1).abs().max())
    nd = int((y1.view(torch.int16) != y0.view(torch.int16)).sum())
    worst = float((y1.float() - y0.float()).abs().max())
    print(f"{nd} of {y0.numel()} outputs differ from the resident kernel (worst {worst:.3e})")
    assert nd <= 2e-3 * y0.numel() and worst <= 2.0 ** -6 * float(y0.float().abs().max())


def test_unet_forward_backward_with_streaming_groupnorm_matches_the_resident_path(env_switches):
    import gpu_util as G
    from eegldm.models import UNetModel
    from param_gen import gen_param, timesteps
    cfg = dict(in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)
    net = UNetModel(image_size=768, **cfg, dtype="bfloat16")
    net.load_state_dict({k: torch.from_numpy(gen_param(5, k, tuple(v.shape))) for k, v in net.state_dict().items()})
    x = torch.from_numpy(normal((8, 1, 768), seed=3)).cuda(); t = torch.from_numpy(timesteps(8, seed=4)).cuda()
    dy = torch.from_numpy(normal((8, 1, 768), seed=6)).cuda()
    outs = {}
    for name, sw in (("stream", "1"), ("resident", None), ("stream2", "1")):
        env_switches(EEGLDM_GN_QSTATS=sw, EEGLDM_GEMM_BIG_MIN_TILES="1")
        net.train(); net.zero_grad()
        y = net._forward_native(x, t)
        net.backward(dy)
        outs[name] = (y.clone(), net.flat_grad.clone())
    ya, yb = outs["stream"][0], outs["resident"][0]
    assert not torch.equal(ya, yb), "the streaming path was not taken (outputs bit-identical to the resident path)"
    assert float((ya - yb).abs().max()) <= 3e-2 * float(yb.abs().max()), float((ya - yb).abs().max()) / float(yb.abs().max())
    ga, gb = outs["stream"][1], outs["resident"][1]
    assert float((ga - gb).norm() / gb.norm()) < 3e-2, float((ga - gb).norm() / gb.norm())
    # the forward is reproducible run to run (fp64 atomics: order-independent up to 1e-16, far below the fp32 (mean, rstd))
    assert float((outs["stream2"][0] - ya).abs().max()) <= 1e-3 * float(ya.abs().max())
