"""-m gpu: primitive-level parity of the HIP kernels (through the C ABI) against torch
fp32 CPU functional ops -- the oracle's building blocks -- on identical seeded inputs."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import normal  # noqa: E402


def _imports():
    import gpu_util as G
    return G


CONV_CASES = [
    # B, L, Cin, Cout, K, stride, pad_l, pad_r
    (2, 64, 32, 32, 3, 1, 1, 1),
    (3, 192, 128, 256, 3, 1, 1, 1),      # tiles span samples (L not a multiple of 128)
    (2, 96, 96, 64, 3, 1, 1, 1),
    (2, 128, 64, 128, 1, 1, 0, 0),
    (2, 40, 160, 32, 1, 1, 0, 0),        # K tail (160 = 2.5 stages in fp32... ), BN=32
    (2, 128, 32, 32, 3, 2, 0, 1),        # AEKL downsample: right pad only
    (2, 128, 64, 128, 3, 2, 1, 1),       # discriminator stride 2
    (2, 64, 1, 32, 3, 1, 1, 1),          # thin: conv_in
    (2, 64, 3, 32, 3, 1, 1, 1),
    (2, 64, 32, 1, 3, 1, 1, 1),          # thin: conv_out
    (2, 64, 2, 4, 3, 1, 1, 1),           # [2,2,4] autoencoder
    (2, 64, 4, 4, 3, 2, 0, 1),
    (2, 64, 1, 64, 3, 2, 1, 1),          # discriminator first conv
    (2, 48, 512, 1, 3, 1, 1, 1),         # discriminator last conv
    (8, 128, 64, 256, 3, 1, 1, 1),       # 8 M tiles x 2 N tiles: XCD-aware tile order (all N tiles of an M tile on one XCD)
    (8, 128, 64, 256, 1, 1, 0, 0),       # same for the 1-tap kernels
    (32, 512, 128, 256, 3, 1, 1, 1),     # split-K weight gradient with splitk % 8 == 0: one K split per XCD
    # production reduction lengths of the config_ldm UNet's deepest level (T = 192): K up to 3 x 1024 per output, split-K with
    # skewed chunks, 192-row tiles, fused bias gradient (colsum) at Cin in {512, 768, 1024}
    (32, 192, 512, 512, 3, 1, 1, 1),
    (32, 192, 768, 512, 3, 1, 1, 1),
    (32, 192, 1024, 512, 3, 1, 1, 1),
    (32, 192, 1024, 512, 1, 1, 0, 0),    # skip_connection of the first output block
    (32, 192, 512, 1536, 1, 1, 0, 0),    # qkv
    (16, 384, 768, 256, 3, 1, 1, 1),     # output_blocks.3: 512 + 256 -> 256 at T = 384
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv1d_fwd_bwd(case, dtype, conv_kernel_path):
    G = _imports()
    B, L, Cin, Cout, K, stride, pl, pr = case
    x = torch.from_numpy(normal((B, Cin, L), seed=1)).requires_grad_(True)
    w = (torch.from_numpy(normal((Cout, Cin, K), seed=2)) / math.sqrt(Cin * K)).requires_grad_(True)
    b = torch.from_numpy(normal((Cout,), seed=3)).requires_grad_(True)
    if dtype == G.BF16:   # compare against the oracle evaluated on the bf16-rounded operands
        x = x.detach().bfloat16().float().requires_grad_(True); w = w.detach().bfloat16().float().requires_grad_(True)
    y_ref = F.conv1d(F.pad(x, (pl, pr)), w, b, stride=stride)
    Lout = y_ref.shape[-1]
    dy = torch.from_numpy(normal(tuple(y_ref.shape), seed=4))
    if dtype == G.BF16:
        dy = dy.bfloat16().float()
    y_ref.backward(dy)

    c = G.ctx()
    xd, wd, bd = G.nlc(x.detach(), dtype), G.pack_w(w.detach(), dtype), b.detach().to(G.DEV)
    yd = torch.empty(B * Lout, Cout, device=G.DEV, dtype=G.TDT[dtype])
    G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, K, stride, pl, pr,
                                    None, 0, None, 0, dtype))
    G.assert_close(G.ncl(yd, B, Lout), y_ref, **G.TOL[dtype], name="y")

    dyd = G.nlc(dy, dtype)
    dxd = torch.empty(B * L, Cin, device=G.DEV, dtype=G.TDT[dtype])
    G.check(G.lib.eegldm_conv1d_bwd_data(c.h, G.ptr(dyd), Cout, G.ptr(wd), G.ptr(dxd), Cin, B, L, Cin, Cout, K, stride, pl, pr, None, 0, dtype))
    G.assert_close(G.ncl(dxd, B, L), x.grad, **G.GTOL[dtype], name="dx")

    dwd = torch.zeros(K, Cout, Cin, device=G.DEV); dbd = torch.zeros(Cout, device=G.DEV)
    G.check(G.lib.eegldm_conv1d_bwd_weight(c.h, G.ptr(xd), Cin, G.ptr(dyd), Cout, G.ptr(dwd), G.ptr(dbd), B, L, Cin, Cout, K, stride, pl, pr, dtype))
    scale = float(w.grad.abs().max())
    G.assert_close(G.unpack_w(dwd), w.grad, rtol=G.GTOL[dtype]["rtol"], atol=G.GTOL[dtype]["atol"] * max(1.0, scale), name="dw")
    G.assert_close(dbd, b.grad, rtol=G.GTOL[dtype]["rtol"], atol=G.GTOL[dtype]["atol"] * max(1.0, float(b.grad.abs().max())), name="db")


KBLK_CASES = [   # (B, L, Cin, Cout, stride): K-blocked weight copy vs the plain layout, same kernel, results must be bit-identical
    (8, 192, 512, 512, 1), (4, 192, 1024, 512, 1), (4, 384, 256, 192, 1),   # Cout = 192: a partial 128-column tile
    (8, 768, 128, 128, 1), (4, 384, 64, 128, 2), (2, 192, 32, 64, 1), (4, 96, 768, 256, 1),
]


@pytest.mark.parametrize("case", KBLK_CASES)
def test_conv1d_kblocked_weights_bit_identical(case):
    """eegldm_conv1d_pack_kblocked: [3][Cout][Cin] -> [3][Cin/32][Cout][32]; the forward conv then reads contiguous weight tiles.
    Only the addressing of the weight operand changes, so the outputs are compared for equality, and against the fp32 reference."""
    G = _imports()
    B, L, Cin, Cout, stride = case
    dtype = G.BF16
    x = torch.from_numpy(normal((B, Cin, L), seed=11)).bfloat16().float()
    w = (torch.from_numpy(normal((Cout, Cin, 3), seed=12)) / math.sqrt(3 * Cin)).bfloat16().float()
    b = torch.from_numpy(normal((Cout,), seed=13))
    pl, pr = (1, 1) if stride == 1 else (0, 1)
    y_ref = F.conv1d(F.pad(x, (pl, pr)), w, b, stride=stride)
    Lout = y_ref.shape[-1]
    c = G.ctx()
    xd, wd, bd = G.nlc(x, dtype), G.pack_w(w, dtype), b.to(G.DEV)
    run = lambda out: G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(out), Cout, B, L, Cin, Cout, 3, stride, pl, pr,
                                                      None, 0, None, 0, dtype))
    y_plain = torch.empty(B * Lout, Cout, device=G.DEV, dtype=G.TDT[dtype]); run(y_plain)
    wk = torch.full_like(wd, float("nan"))
    G.check(G.lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(wd), G.ptr(wk), Cout, Cin, dtype))
    try:
        # the copy is a permutation of the weight: [t][ci/32][co][ci%32]
        exp = wd.view(3, Cout, Cin // 32, 32).permute(0, 2, 1, 3).contiguous().view(-1)
        assert torch.equal(wk.view(-1).view(torch.int16), exp.view(torch.int16)), "packed layout"
        y_kb = torch.empty_like(y_plain); run(y_kb)
    finally:
        G.check(G.lib.eegldm_conv1d_forget_kblocked(c.h, G.ptr(wd)))
    assert torch.equal(y_kb.view(torch.int16), y_plain.view(torch.int16)), "K-blocked weights changed the result"
    G.assert_close(G.ncl(y_kb, B, Lout), y_ref, **G.TOL[dtype], name="y (K-blocked)")
    y_again = torch.empty_like(y_plain); run(y_again)                  # registration removed: plain path again
    assert torch.equal(y_again.view(torch.int16), y_plain.view(torch.int16))


def test_conv1d_pack_kblocked_rejects_bad_arguments():
    G = _imports()
    c = G.ctx()
    w = torch.zeros(3 * 64 * 48, device=G.DEV, dtype=torch.bfloat16); wk = torch.empty_like(w)
    assert G.lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(w), G.ptr(wk), 64, 48, G.BF16) != 0      # Cin % 32 != 0
    wf = torch.zeros(3 * 64 * 64, device=G.DEV); wkf = torch.empty_like(wf)
    assert G.lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(wf), G.ptr(wkf), 64, 64, G.F32) != 0    # fp32 has no K-blocked path
    assert G.lib.eegldm_conv1d_pack_kblocked(c.h, G.ptr(w), G.ptr(w), 64, 64, G.BF16) != 0        # aliased


@pytest.mark.parametrize("dtype", [0, 1])
def test_conv1d_epilogue_rowvec_resid(dtype, conv_kernel_path):
    """bias + per-sample embedding row + residual in the GEMM epilogue; L = 192 puts a sample boundary inside a 128-row tile
    (fp32: LDS fp32 tile path; bf16: fragment-layout addends, packed-bf16 LDS transpose, prefetch under the last K stage)."""
    G = _imports()
    B, L, Cin, Cout = 3, 192, 64, 128
    x = torch.from_numpy(normal((B, Cin, L), seed=1)); w = torch.from_numpy(normal((Cout, Cin, 3), seed=2)) / 14.0
    b = torch.from_numpy(normal((Cout,), seed=3)); e = torch.from_numpy(normal((B, Cout), seed=4)); r = torch.from_numpy(normal((B, Cout, L), seed=5))
    if dtype == G.BF16:
        x, w, r = x.bfloat16().float(), w.bfloat16().float(), r.bfloat16().float()
    y_ref = F.conv1d(x, w, b, padding=1) + e[:, :, None] + r
    c = G.ctx()
    xd, wd, rd = G.nlc(x, dtype), G.pack_w(w, dtype), G.nlc(r, dtype)
    yd = torch.empty(B * L, Cout, device=G.DEV, dtype=G.TDT[dtype]); bd, ed = b.to(G.DEV), e.to(G.DEV)
    G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, 3, 1, 1, 1,
                                    G.ptr(ed), Cout, G.ptr(rd), Cout, dtype))
    G.assert_close(G.ncl(yd, B, L), y_ref, **G.TOL[dtype], name="y")


GN_CASES = [  # B, L, C, G, silu, resample
    (2, 64, 32, 32, 1, 0), (3, 96, 128, 32, 1, 0), (2, 64, 768, 32, 1, 0), (2, 64, 64, 32, 0, 0),
    (2, 64, 32, 32, 1, 1), (2, 32, 64, 32, 1, 2), (2, 128, 32, 1, 1, 0), (2, 128, 2, 1, 1, 0), (2, 64, 4, 1, 0, 0),
    (2, 3072, 32, 1, 1, 0),
    # thin AutoencoderKL layers (G = 1, C <= 8): one-launch flat kernels, small (3 chunks/thread) and large (12) variants
    (2, 1536, 32, 1, 1, 0), (3, 768, 64, 1, 0, 0), (2, 100, 16, 1, 1, 0),      # [32,32,64] AutoencoderKL: wide flat forward (1024 threads)
    (4, 3072, 2, 1, 1, 0), (3, 768, 4, 1, 1, 0), (2, 96, 8, 1, 1, 0), (2, 40, 1, 1, 1, 0), (2, 3072, 8, 1, 0, 0), (2, 1000, 4, 1, 1, 0),
]


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", GN_CASES)
def test_groupnorm_fwd_bwd(case, dtype):
    G = _imports()
    B, L, Cc, Gr, silu, rs = case
    x = (torch.from_numpy(normal((B, Cc, L), seed=1)) * 1.5 + 0.7)
    if dtype == G.BF16:
        x = x.bfloat16().float()
    x.requires_grad_(True)
    ga = (1 + 0.1 * torch.from_numpy(normal((Cc,), seed=2))).requires_grad_(True)
    be = (0.1 * torch.from_numpy(normal((Cc,), seed=3))).requires_grad_(True)
    h = F.group_norm(x, Gr, ga, be, eps=1e-6)
    if silu:
        h = F.silu(h)
    xr_ref = x
    if rs == 1:
        h = F.avg_pool1d(h, 2, 2); xr_ref = F.avg_pool1d(x, 2, 2)
    elif rs == 2:
        h = F.interpolate(h, scale_factor=2, mode="nearest"); xr_ref = F.interpolate(x, scale_factor=2, mode="nearest")
    Lo = h.shape[-1]
    dy = torch.from_numpy(normal(tuple(h.shape), seed=4)); dxr = torch.from_numpy(normal(tuple(h.shape), seed=5))
    if dtype == G.BF16:
        dy, dxr = dy.bfloat16().float(), dxr.bfloat16().float()
    ((h * dy).sum() + (xr_ref * dxr).sum()).backward()

    c = G.ctx()
    xd = G.nlc(x.detach(), dtype); gad, bed = ga.detach().to(G.DEV), be.detach().to(G.DEV)
    yd = torch.empty(B * Lo, Cc, device=G.DEV, dtype=G.TDT[dtype]); xrd = torch.empty_like(yd)
    st = torch.empty(B * Gr * 2, device=G.DEV)
    G.check(G.lib.eegldm_groupnorm_fwd(c.h, G.ptr(xd), Cc, G.ptr(gad), G.ptr(bed), G.ptr(yd), Cc, G.ptr(st), B, L, Cc, Gr, 1e-6, silu, rs,
                                       G.ptr(xrd), Cc, dtype))
    G.assert_close(G.ncl(yd, B, Lo), h, **G.TOL[dtype], name="y")
    if rs:
        G.assert_close(G.ncl(xrd, B, Lo), xr_ref, **G.TOL[dtype], name="xr")
    dyd, dxrd = G.nlc(dy, dtype), G.nlc(dxr, dtype)
    dxd = torch.empty(B * L, Cc, device=G.DEV, dtype=G.TDT[dtype]); dga = torch.zeros(Cc, device=G.DEV); dbe = torch.zeros(Cc, device=G.DEV)
    G.check(G.lib.eegldm_groupnorm_bwd(c.h, G.ptr(xd), Cc, G.ptr(gad), G.ptr(bed), G.ptr(st), G.ptr(dyd), Cc, G.ptr(dxd), Cc, G.ptr(dga), G.ptr(dbe),
                                       B, L, Cc, Gr, silu, rs, G.ptr(dxrd), Cc, dtype))
    G.assert_close(G.ncl(dxd, B, L), x.grad, **G.GTOL[dtype], name="dx")
    sc = max(1.0, float(ga.grad.abs().max()))
    G.assert_close(dga, ga.grad, rtol=G.GTOL[dtype]["rtol"], atol=G.GTOL[dtype]["atol"] * sc, name="dgamma")
    G.assert_close(dbe, be.grad, rtol=G.GTOL[dtype]["rtol"], atol=G.GTOL[dtype]["atol"] * sc, name="dbeta")


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("B,T,Cc", [(2, 24, 32), (2, 16, 128), (3, 192, 512), (2, 72, 64), (2, 64, 256), (3, 128, 256), (2, 256, 256),
                                    (130, 192, 256), (2, 384, 128), (1, 768, 64), (2, 768, 512), (5, 768, 256)])
# (130, 192, 256): fused chain kernel with whole-sample (12-wave) blocks; (*, 768, 512 / 256): the 8-column-wave chain kernel of the
# pixel-space model's attention level (bf16; fp32 and C = 64 take the GEMM + softmax composition)
def test_attention_fwd_bwd(B, T, Cc, dtype):
    G = _imports()
    from oracle.unet import qkv_attention
    qkv = torch.from_numpy(normal((B, 3 * Cc, T), seed=1))
    if dtype == G.BF16:
        qkv = qkv.bfloat16().float()
    qkv.requires_grad_(True)
    a = qkv_attention(qkv)
    dy = torch.from_numpy(normal(tuple(a.shape), seed=2))
    if dtype == G.BF16:
        dy = dy.bfloat16().float()
    a.backward(dy)
    c = G.ctx()
    qd = G.nlc(qkv.detach(), dtype)
    od = torch.empty(B * T, Cc, device=G.DEV, dtype=G.TDT[dtype]); pr = torch.empty(B * T * T, device=G.DEV, dtype=G.TDT[dtype])
    s1 = torch.empty(B * T * T, device=G.DEV); s2 = torch.empty(B * T * T, device=G.DEV, dtype=G.TDT[dtype])
    G.check(G.lib.eegldm_attention_fwd(c.h, G.ptr(qd), 3 * Cc, G.ptr(od), Cc, G.ptr(pr), G.ptr(s1), B, T, Cc, dtype))
    G.assert_close(G.ncl(od, B, T), a, **G.TOL[dtype], name="attn out")
    dod = G.nlc(dy, dtype); dq = torch.empty(B * T, 3 * Cc, device=G.DEV, dtype=G.TDT[dtype])
    G.check(G.lib.eegldm_attention_bwd(c.h, G.ptr(qd), 3 * Cc, G.ptr(pr), G.ptr(dod), Cc, G.ptr(dq), 3 * Cc, G.ptr(s1), G.ptr(s2), B, T, Cc, dtype))
    G.assert_close(G.ncl(dq, B, T), qkv.grad, **G.GTOL[dtype], name="dqkv")


def test_layout_roundtrip_and_pack():
    G = _imports()
    c = G.ctx()
    for (B, Cc, L) in [(2, 1, 70), (3, 130, 65), (2, 3, 768)]:
        x = torch.from_numpy(normal((B, Cc, L), seed=9)).to(G.DEV)
        t = torch.empty(B * L, Cc, device=G.DEV); back = torch.empty_like(x)
        G.check(G.lib.eegldm_ncl_to_nlc(c.h, G.ptr(x), G.ptr(t), Cc, B, Cc, L, 0))
        G.assert_close(t, x.permute(0, 2, 1).reshape(B * L, Cc), rtol=0, atol=0, name="ncl->nlc")
        G.check(G.lib.eegldm_nlc_to_ncl(c.h, G.ptr(t), Cc, G.ptr(back), B, Cc, L, 0))
        G.assert_close(back, x, rtol=0, atol=0, name="roundtrip")
    w = torch.from_numpy(normal((5, 7, 3), seed=1)).to(G.DEV); p = torch.empty(3, 5, 7, device=G.DEV); u = torch.empty_like(w)
    G.check(G.lib.eegldm_pack_conv_weight(c.h, G.ptr(w), G.ptr(p), 5, 7, 3))
    G.assert_close(p, w.permute(2, 0, 1), rtol=0, atol=0, name="pack")
    G.check(G.lib.eegldm_unpack_conv_weight(c.h, G.ptr(p), G.ptr(u), 5, 7, 3))
    G.assert_close(u, w, rtol=0, atol=0, name="unpack")


def test_scheduler_mse_adam_rng():
    G = _imports()
    from oracle import losses as Ls
    from oracle.steps import adam_update
    c = G.ctx()
    B, n = 5, 768
    x, nz = torch.from_numpy(normal((B, 1, n), seed=1)), torch.from_numpy(normal((B, 1, n), seed=2))
    t = torch.tensor([0, 3, 499, 998, 999]); acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0195)
    out = torch.empty(B, 1, n, device=G.DEV)
    xd, nzd, td, acpd = x.to(G.DEV), nz.to(G.DEV), t.to(G.DEV), acp.to(G.DEV)     # keep alive: the library only sees raw pointers
    G.check(G.lib.eegldm_add_noise(c.h, G.ptr(xd), G.ptr(nzd), G.ptr(td), G.ptr(acpd), G.ptr(out), B, n))
    G.assert_close(out, Ls.add_noise(acp, x, nz, t), rtol=1e-5, atol=1e-6, name="add_noise")
    G.check(G.lib.eegldm_get_velocity(c.h, G.ptr(xd), G.ptr(nzd), G.ptr(td), G.ptr(acpd), G.ptr(out), B, n))
    G.assert_close(out, Ls.get_velocity(acp, x, nz, t), rtol=1e-5, atol=1e-6, name="velocity")
    for pred, name in [(0, "epsilon"), (1, "v_prediction"), (2, "sample")]:
        for tt in (980, 20, 0):
            prev, x0 = Ls.ddim_step(acp, nz, tt, x, 1000, 50, name, clip_sample=(pred == 0))
            a_prev = float(acp[tt - 20]) if tt - 20 >= 0 else 1.0
            p_d, x0_d = torch.empty(B * n, device=G.DEV), torch.empty(B * n, device=G.DEV)
            G.check(G.lib.eegldm_ddim_step(c.h, G.ptr(nzd), G.ptr(xd), float(acp[tt]), a_prev, pred, int(pred == 0), G.ptr(p_d), G.ptr(x0_d), B * n))
            G.assert_close(p_d.reshape(B, 1, n), prev, rtol=2e-5, atol=2e-5, name=f"ddim prev {name} t={tt}")
            G.assert_close(x0_d.reshape(B, 1, n), x0, rtol=2e-5, atol=2e-5, name="ddim x0")
    loss = torch.zeros(1, device=G.DEV); dp = torch.empty(B * n, device=G.DEV)
    G.check(G.lib.eegldm_mse_loss(c.h, G.ptr(xd), G.ptr(nzd), G.ptr(loss), G.ptr(dp), B * n, 1.0))
    xr = x.clone().requires_grad_(True); l = F.mse_loss(xr, nz); l.backward()
    G.assert_close(loss, l.reshape(1), rtol=1e-5, atol=1e-7, name="mse"); G.assert_close(dp.reshape(B, 1, n), xr.grad, rtol=1e-5, atol=1e-9, name="dmse")
    p = torch.from_numpy(normal((1000,), seed=5)); st = {}
    pd = p.clone().to(G.DEV); m = torch.zeros(1000, device=G.DEV); v = torch.zeros(1000, device=G.DEV)
    params = {"p": p.clone()}
    for step in range(1, 4):
        g = torch.from_numpy(normal((1000,), seed=10 + step)); gd = g.to(G.DEV)
        params = adam_update(params, {"p": g}, st, 1e-2, step)
        G.check(G.lib.eegldm_adam_step(c.h, G.ptr(pd), G.ptr(gd), G.ptr(m), G.ptr(v), 1000, 1e-2, 0.9, 0.999, 1e-8, step, 1.0))
        torch.cuda.synchronize()
    G.assert_close(pd, params["p"], rtol=1e-5, atol=1e-6, name="adam")
    r = torch.empty(1 << 20, device=G.DEV)
    G.check(G.lib.eegldm_randn(c.h, G.ptr(r), r.numel(), 1234, 0))
    assert abs(float(r.mean())) < 5e-3 and abs(float(r.std()) - 1) < 5e-3 and abs(float((r ** 4).mean()) - 3) < 0.05
    ti = torch.empty(1 << 16, device=G.DEV, dtype=torch.int64)
    G.check(G.lib.eegldm_randint(c.h, G.ptr(ti), ti.numel(), 1000, 7, 0))
    assert int(ti.min()) >= 0 and int(ti.max()) <= 999 and abs(float(ti.float().mean()) - 499.5) < 5


LINEAR_CASES = [(8, 512, 128), (256, 512, 512), (256, 7168, 512), (5, 96, 40), (3, 8, 32)]     # M, N, K: time_embed.0 / .2, the 21 stacked emb_layers, ragged M, one N vector group


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("case", LINEAR_CASES)
def test_linear_fwd_bwd(case, dtype, conv_kernel_path):
    """nn.Linear (unet.py:373-377, 277-285) forward and its autograd backward through eegldm_linear_fwd / eegldm_linear_bwd."""
    G = _imports()
    M, N, K = case
    x = torch.from_numpy(normal((M, K), seed=11)).requires_grad_(True)
    w = (torch.from_numpy(normal((N, K), seed=12)) / math.sqrt(K)).requires_grad_(True)
    b = torch.from_numpy(normal((N,), seed=13)).requires_grad_(True)
    dy = torch.from_numpy(normal((M, N), seed=14))
    if dtype == G.BF16:
        x = x.detach().bfloat16().float().requires_grad_(True); w = w.detach().bfloat16().float().requires_grad_(True); dy = dy.bfloat16().float()
    y_ref = F.linear(x, w, b)
    y_ref.backward(dy)
    c = G.ctx()
    td = G.TDT[dtype]
    xd, wd, bd, dyd = x.detach().to(G.DEV).to(td), w.detach().to(G.DEV).to(td), b.detach().to(G.DEV), dy.to(G.DEV).to(td)
    yd = torch.empty(M, N, device=G.DEV)
    G.check(G.lib.eegldm_linear_fwd(c.h, G.ptr(xd), K, G.ptr(wd), G.ptr(bd), G.ptr(yd), N, M, N, K, dtype, 1))
    G.assert_close(yd, y_ref, **G.TOL[dtype], name="y")
    dxd = torch.empty(M, K, device=G.DEV); dwd = torch.zeros(N, K, device=G.DEV); dbd = torch.zeros(N, device=G.DEV)
    G.check(G.lib.eegldm_linear_bwd(c.h, G.ptr(xd), K, G.ptr(wd), G.ptr(dyd), N, G.ptr(dxd), K, G.ptr(dwd), G.ptr(dbd), M, N, K, dtype, 1))
    G.assert_close(dxd, x.grad, **G.GTOL[dtype], name="dx")
    G.assert_close(dwd, w.grad, **G.GTOL[dtype], name="dw")
    G.assert_close(dbd, b.grad, **G.GTOL[dtype], name="dbias")
    # accumulation semantics (+=) and the optional outputs
    G.check(G.lib.eegldm_linear_bwd(c.h, G.ptr(xd), K, None, G.ptr(dyd), N, None, 0, G.ptr(dwd), None, M, N, K, dtype, 1))
    G.assert_close(dwd, 2 * w.grad, **G.GTOL[dtype], name="dw accumulated")
    # the reduction length must be a whole number of 16-byte chunks (MFMA operand rows): a clean error, not a wrong result
    with pytest.raises(RuntimeError, match="multiples of"):
        G.check(G.lib.eegldm_linear_fwd(c.h, G.ptr(xd), 33, G.ptr(wd), G.ptr(bd), G.ptr(yd), N, 1, 1, 33, dtype, 1))
