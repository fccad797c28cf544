"""-m gpu: AutoencoderKL / PatchDiscriminator / losses / fused GAN train step on the HIP engine
against the CPU oracle (oracle/aekl.py, oracle/losses.py, oracle/steps.py) on identical seeded inputs.
The oracle itself is "parity unpinned" at the MONAI boundary (see oracle/aekl.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import gen_param, normal, eeg_windows  # noqa: E402

AE_CASES = {
    "c32_32_64_lat1": dict(num_channels=[32, 32, 64], latent_channels=1),
    "c32_32_64_lat3": dict(num_channels=[32, 32, 64], latent_channels=3),
    "c2_2_4_lat1": dict(num_channels=[2, 2, 4], latent_channels=1),
    "c16_32_lat2": dict(num_channels=[16, 32], latent_channels=2),
    # reference configs config_aekl_eeg_4_16_32 / _4_4_16 / _8_8_16: ResBlocks with a 1x1 shortcut whose bias gradient shares its
    # column sum with conv2's (round-1 advisor finding: double-counted when only one of the two weight-gradient GEMMs fuses it)
    "c4_16_32_lat1": dict(num_channels=[4, 16, 32], latent_channels=1),
    "c4_4_16_lat1": dict(num_channels=[4, 4, 16], latent_channels=1),
    "c8_8_16_lat1": dict(num_channels=[8, 8, 16], latent_channels=1),
}
D_CFG = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1); b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


def check_grads(got, want, tol, floor_frac, label):
    gscale = max(float(v.norm()) for v in want.values())
    worst = ("", 0.0)
    for k, w in want.items():
        e = float((got[k].cpu().double() - w.double()).norm()) / (float(w.norm()) + floor_frac * gscale)
        if e > worst[1]:
            worst = (k, e)
        assert e < tol, f"{label} grad {k}: rel err {e:.3e}"
    return worst


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("name", list(AE_CASES))
def test_autoencoderkl_fwd_bwd(name, dtype):
    from eegldm.models import AutoencoderKL
    from oracle import aekl as A, losses as Ls
    cfg = dict(AE_CASES[name], in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    B, L = 2, 256
    shapes = A.aekl_param_shapes(cfg)
    sd = {k: torch.from_numpy(gen_param(11, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(eeg_windows(B, seed=5, length=L, pad=8)).requires_grad_(True)
    Ll = L >> (len(cfg["num_channels"]) - 1)
    eps = torch.from_numpy(normal((B, cfg["latent_channels"], Ll), seed=6))
    recon, mu, sg = A.forward(sd, cfg, x, eps)
    kl = Ls.kl_loss(mu, sg)
    dy = torch.from_numpy(normal(tuple(recon.shape), seed=7))
    klw = 0.3
    ((recon * dy).sum() + klw * kl).backward()

    net = AutoencoderKL(spatial_dims=1, attention_levels=[False] * len(cfg["num_channels"]), dtype=dtype, **cfg)
    assert list(net.entries.keys()) == list(shapes.keys())
    assert sum(int(np.prod(s)) for s in shapes.values()) == sum(n for (_o, n, _s) in net.entries.values())
    net.load_state_dict({k: v.detach() for k, v in sd.items()})
    klo = torch.zeros(1, device=net.device)
    r_d, mu_d, sg_d = net(x.detach(), eps=eps, kl_out=klo)
    f32 = dtype == "float32"
    t = 2e-5 if f32 else 5e-2
    assert rel_l2(r_d, recon) < t and rel_l2(mu_d, mu) < t and rel_l2(sg_d, sg) < t, (rel_l2(r_d, recon), rel_l2(mu_d, mu), rel_l2(sg_d, sg))
    assert abs(float(klo) - float(kl)) < (1e-4 if f32 else 5e-2) * abs(float(kl)) + 1e-5
    net.zero_grad()
    dx = net.backward(dy, kl_weight=klw, need_dx=True)
    if f32:
        assert rel_l2(dx, x.grad) < 1e-4, rel_l2(dx, x.grad)
        worst = check_grads(net.grad_dict(), {k: v.grad for k, v in sd.items()}, 2e-3, 1e-3, name)
    else:
        # bounds derived from the bf16-storage oracle (gpu_util.assert_bf16_grads) instead of the round-1 0.12 / 0.15 constants
        import gpu_util as G
        from oracle import quant as Q
        sdq = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}
        xq = x.detach().clone().requires_grad_(True)
        with Q.bf16_storage(True):
            rq, muq, sgq = A.forward(sdq, cfg, xq, eps)
            ((rq * dy).sum() + klw * Ls.kl_loss(muq, sgq)).backward()
        assert rel_l2(r_d, recon) < G.bf16_gap_bound(rel_l2(rq, recon)) and rel_l2(dx, x.grad) < G.bf16_gap_bound(rel_l2(xq.grad, x.grad)), \
            (rel_l2(r_d, recon), rel_l2(rq, recon), rel_l2(dx, x.grad), rel_l2(xq.grad, x.grad))
        msg = G.assert_bf16_grads(net.grad_dict(), {k: v.grad for k, v in sd.items()}, {k: v.grad for k, v in sdq.items()}, name, floor_frac=3e-2, factor=2.5)
        worst = (msg, 0.0)
    # encode / decode entry points agree with forward
    z = net.encode_stage_2_inputs(x.detach(), eps=eps)
    assert rel_l2(z, (mu + eps * sg)) < t
    assert rel_l2(net.decode(z), recon) < (t if f32 else 8e-2)
    print(f"{name} {dtype}: recon {rel_l2(r_d, recon):.2e} dx {rel_l2(dx, x.grad):.2e} worst grad {worst[0]} {worst[1]:.2e}")


def test_autoencoderkl_param_counts():
    """SURVEY.md Appendix B structural recount: 174 184 ([32,32,64], lat 1), 174 984 (lat 3), 934 ([2,2,4])."""
    from eegldm.models import AutoencoderKL
    for nc, lat, want in [([32, 32, 64], 1, 174184), ([32, 32, 64], 3, 174984), ([2, 2, 4], 1, 934)]:
        net = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=nc, latent_channels=lat, num_res_blocks=2,
                            norm_num_groups=1, attention_levels=[False, False, False])
        assert sum(n for (_o, n, _s) in net.entries.values()) == want
        r, mu, sg = net(torch.zeros(2, 1, 3072) + 0.5)
        assert r.shape == (2, 1, 3072) and mu.shape == (2, lat, 768) and sg.shape == (2, lat, 768)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_patch_discriminator_fwd_bwd(dtype):
    from eegldm.models import PatchDiscriminator
    from oracle import aekl as A
    B, L = 3, 256
    shapes = A.disc_param_shapes(D_CFG)
    sd = {}
    for k, s in shapes.items():
        v = torch.from_numpy(gen_param(21, k, s))
        if k.endswith("conv.weight"):
            v = v * 2.0
        sd[k] = v.requires_grad_(True) if v.is_floating_point() and "running" not in k else v
    x = torch.from_numpy(normal((B, 1, L), seed=8)).requires_grad_(True)
    running = {}
    feats_ref = A.disc_forward(sd, D_CFG, x, True, running)
    logits = feats_ref[-1]
    dy = torch.from_numpy(normal(tuple(logits.shape), seed=9))
    (logits * dy).sum().backward()
    net = PatchDiscriminator(**D_CFG, dtype=dtype)
    assert sum(n for (_o, n, _s) in net.entries.values()) == 519681
    net.load_state_dict({k: v.detach() for k, v in sd.items()})
    f32 = dtype == "float32"
    feats = net(x.detach())
    # the reference's return value is the LIST of per-block feature maps (initial, one per layer, logits): all of them, not only [-1]
    assert len(feats) == len(feats_ref) == D_CFG["num_layers_d"] + 2
    for i, (got_f, want_f) in enumerate(zip(feats, feats_ref)):
        assert tuple(got_f.shape) == tuple(want_f.shape), (i, got_f.shape, want_f.shape)
        assert rel_l2(got_f, want_f.detach()) < (2e-5 if f32 else 5e-2), (i, rel_l2(got_f, want_f.detach()))
    out = feats[-1]
    assert out.shape == logits.shape
    assert rel_l2(out, logits) < (2e-5 if f32 else 5e-2), rel_l2(out, logits)
    net.zero_grad()
    dx = net.backward(dy, need_dx=True, in_shape=tuple(x.shape))
    want = {k: v.grad for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point() and v.grad is not None}
    got = net.grad_dict()
    if f32:
        assert rel_l2(dx, x.grad) < 2e-4, rel_l2(dx, x.grad)
        worst = check_grads({k: got[k] for k in want}, want, 2e-3, 1e-3, "disc")
    else:
        import gpu_util as G
        from oracle import quant as Q
        sdq = {k: (v.detach().clone().requires_grad_(True) if (torch.is_tensor(v) and v.requires_grad) else v) for k, v in sd.items()}
        xq = x.detach().clone().requires_grad_(True)
        with Q.bf16_storage(True):
            lq = A.disc_forward(sdq, D_CFG, xq, True, {})[-1]
            (lq * dy).sum().backward()
        assert rel_l2(dx, x.grad) < G.bf16_gap_bound(rel_l2(xq.grad, x.grad)), (rel_l2(dx, x.grad), rel_l2(xq.grad, x.grad))
        msg = G.assert_bf16_grads({k: got[k] for k in want}, want, {k: sdq[k].grad for k in want}, "disc", floor_frac=3e-2, factor=2.5)
        worst = (msg, 0.0)
    new = net.state_dict()
    for k, v in running.items():
        assert rel_l2(new[k], v) < (1e-5 if f32 else 2e-2), k
    assert int(new["0.adn.N.num_batches_tracked"]) == 1
    # eval mode uses the running statistics
    net.eval()
    sd_eval = {k: (new[k].cpu() if k in new else v) for k, v in sd.items()}
    ref_eval = A.disc_forward({k: v.detach() if torch.is_tensor(v) else v for k, v in sd_eval.items()}, D_CFG, x.detach(), False)[-1]
    assert rel_l2(net(x.detach())[-1], ref_eval) < (2e-5 if f32 else 5e-2)
    print(f"disc {dtype}: logits {rel_l2(out, logits):.2e} dx {rel_l2(dx, x.grad):.2e} worst grad {worst[0]} {worst[1]:.2e}")


@pytest.mark.parametrize("L", [3072, 768, 256, 96])
def test_spectral_l1_lsgan_losses(L):
    import gpu_util as G
    from oracle import losses as Ls
    c = G.ctx(); B = 3
    a = torch.from_numpy(eeg_windows(B, seed=1, length=L, pad=4)) + 0.05 * torch.from_numpy(normal((B, 1, L), seed=2))
    b = torch.from_numpy(eeg_windows(B, seed=3, length=L, pad=4))
    ar = a.clone().requires_grad_(True)
    spec = Ls.jukebox_loss(ar, b, "sum"); spec.backward()
    ad, bd = a.to(G.DEV), b.to(G.DEV)
    loss = torch.zeros(1, device=G.DEV); d = torch.zeros(B, 1, L, device=G.DEV)
    G.check(G.lib.eegldm_spectral_loss(c.h, G.ptr(ad), G.ptr(bd), G.ptr(loss), G.ptr(d), B, 1, L, 2.5))
    assert abs(float(loss) - float(spec)) < 2e-5 * abs(float(spec)) + 1e-6, (float(loss), float(spec))
    assert rel_l2(d, 2.5 * ar.grad) < 5e-5, rel_l2(d, 2.5 * ar.grad)
    # closed forms: loss(x, x) == 0 ; Parseval
    G.check(G.lib.eegldm_spectral_loss(c.h, G.ptr(ad), G.ptr(ad), G.ptr(loss), None, B, 1, L, 0.0))
    assert float(loss) < 1e-8
    z = torch.zeros_like(ad)
    G.check(G.lib.eegldm_spectral_loss(c.h, G.ptr(ad), G.ptr(z), G.ptr(loss), None, B, 1, L, 0.0))
    assert abs(float(loss) - float((a ** 2).sum())) < 1e-4 * float((a ** 2).sum())
    # L1
    ar = a.clone().requires_grad_(True); l1 = F.l1_loss(ar, b); l1.backward()
    d.zero_()
    G.check(G.lib.eegldm_l1_loss(c.h, G.ptr(ad), G.ptr(bd), G.ptr(loss), G.ptr(d), B * L, 1.0))
    assert abs(float(loss) - float(l1)) < 1e-5 * float(l1) and rel_l2(d, ar.grad) < 1e-6
    # LSGAN
    lg = torch.from_numpy(normal((B, 1, L), seed=4))
    for real, for_d in [(True, False), (False, True), (True, True)]:
        lr = lg.clone().requires_grad_(True); ref = Ls.patch_adv_loss(lr, real, for_d); ref.backward()
        lgd = lg.to(G.DEV); dl = torch.empty(B, 1, L, device=G.DEV)
        G.check(G.lib.eegldm_lsgan_loss(c.h, G.ptr(lgd), int(real), G.ptr(loss), G.ptr(dl), B * L, 1.0))
        assert abs(float(loss) - float(ref)) < 1e-5 * float(ref) and rel_l2(dl, lr.grad) < 1e-6


@pytest.mark.parametrize("use_spectral", [False, True])
def test_fused_aekl_gan_train_step_matches_oracle(use_spectral):
    """Two full G+D steps (forward, all losses, backward, Adam x2, BatchNorm running stats) vs oracle/steps.py."""
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from eegldm.training import Adam, aekl_train_step
    from oracle import aekl as A, steps as S
    cfg = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    B, L = 4, 256
    ae_sd = {k: torch.from_numpy(gen_param(31, k, s)) for k, s in A.aekl_param_shapes(cfg).items()}
    d_sd = {k: torch.from_numpy(gen_param(32, k, s)) for k, s in A.disc_param_shapes(D_CFG).items()}
    adv_w, kl_w, spec_w = 0.01, 1e-3, 1e-2
    ae = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, **cfg); ae.load_state_dict(ae_sd)
    disc = PatchDiscriminator(**D_CFG); disc.load_state_dict(d_sd)
    og, od = Adam(ae, lr=5e-3), Adam(disc, lr=5e-4)
    sg, sdd = {}, {}
    for step in (1, 2):
        x = torch.from_numpy(eeg_windows(B, seed=40 + step, length=L, pad=8))
        eps = torch.from_numpy(normal((B, 1, L // 4), seed=50 + step))
        losses, ae_sd, d_sd, recon, _gg, _dg = S.aekl_train_step(ae_sd, cfg, d_sd, D_CFG, x, eps, adv_w, kl_w, spec_w, use_spectral,
                                                                 5e-3, 5e-4, step, sg, sdd)
        ae.zero_grad(); disc.zero_grad()
        rec_d = torch.empty(B, 1, L, device=ae.device)
        out = aekl_train_step(ae, disc, x.to(ae.device), eps.to(ae.device), adv_w, kl_w, spec_w, use_spectral, recon_out=rec_d)
        og.step(); od.step()
        o = out.cpu()
        want = [losses["recons"], losses["spectral"], losses["kl"], losses["gen"]]
        for i, w in enumerate(want):
            assert abs(float(o[i]) - float(w)) < 2e-4 * abs(float(w)) + 1e-6, (step, i, float(o[i]), float(w))
        assert abs(0.5 * float(o[4] + o[5]) - float(losses["disc"])) < 2e-4 * float(losses["disc"]) + 1e-6
        assert rel_l2(rec_d, recon) < 2e-5
    # Adam normalises every gradient to ~+-lr, so an element whose true gradient is at rounding-noise level can move by
    # up to 2*lr in either direction; parameters are therefore compared in units of lr: every element within two Adam
    # steps' worth, and the mean deviation a small fraction of one step.
    got = ae.state_dict()
    for k, v in ae_sd.items():
        d = (got[k].cpu() - v).abs()
        assert float(d.max()) < 2.5 * 5e-3 and float(d.mean()) < 0.05 * 5e-3, f"ae {k}: max {float(d.max()):.3e} mean {float(d.mean()):.3e}"
    gotd = disc.state_dict()
    for k, v in d_sd.items():
        if "num_batches" in k:
            assert int(gotd[k]) == 6
        elif "running" in k:
            assert float((gotd[k].cpu().float() - v.float()).abs().max()) < 1e-4 + 1e-5 * float(v.abs().max()), f"disc {k}"
        else:
            d = (gotd[k].cpu().float() - v.float()).abs()
            assert float(d.max()) < 2.5 * 5e-4 and float(d.mean()) < 0.05 * 5e-4, f"disc {k}: max {float(d.max()):.3e} mean {float(d.mean()):.3e}"


def test_autoencoderkl_edge_inputs():
    """Single window, empty batch, and inputs the strided encoder cannot take (loud ValueError, no launch)."""
    from eegldm.models import AutoencoderKL
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype="bfloat16", device=0)
    r, mu, sg = ae(torch.randn(1, 1, 3072))
    assert r.shape == (1, 1, 3072) and mu.shape == sg.shape == (1, 1, 768) and torch.isfinite(r).all()
    r0, mu0, _ = ae(torch.randn(0, 1, 3072))
    assert r0.shape == (0, 1, 3072) and mu0.shape == (0, 1, 768)
    assert ae.decode(torch.randn(0, 1, 768)).shape == (0, 1, 3072)
    assert ae.decode(torch.randn(2, 1, 50)).shape == (2, 1, 200)          # any latent length decodes
    with pytest.raises(ValueError, match="multiple of 4"):
        ae.encode(torch.randn(2, 1, 3001))
    with pytest.raises(ValueError, match=r"\(B, 1, L\)"):
        ae.encode(torch.randn(2, 3, 3072))


def test_spectral_loss_zero_amplitude_gradient_is_zero_not_nan():
    """Documented deviation: where |FFT(recon)| is exactly 0 the derivative of sqrt(re^2 + im^2) is 0/0; torch /
    MONAI's JukeboxLoss propagates NaN from there (the instability the reference README.md:17 mentions), the engine defines the
    sub-gradient as 0.  An all-zero reconstruction makes every bin's amplitude exactly 0: the loss value still matches the oracle,
    the oracle's gradient is NaN, the engine's is exactly 0 for that window and matches the oracle for the other windows."""
    import gpu_util as G
    from oracle import losses as Ls
    c = G.ctx(); B, L = 3, 3072
    a = torch.from_numpy(eeg_windows(B, seed=1, length=L)); a[1] = 0.0
    b = torch.from_numpy(eeg_windows(B, seed=3, length=L))
    ar = a.clone().requires_grad_(True)
    spec = Ls.jukebox_loss(ar, b, "sum"); spec.backward()
    assert torch.isnan(ar.grad[1]).any() and torch.isfinite(ar.grad[0]).all() and torch.isfinite(ar.grad[2]).all()
    ad, bd = a.to(G.DEV), b.to(G.DEV)
    loss = torch.zeros(1, device=G.DEV); d = torch.zeros(B, 1, L, device=G.DEV)
    G.check(G.lib.eegldm_spectral_loss(c.h, G.ptr(ad), G.ptr(bd), G.ptr(loss), G.ptr(d), B, 1, L, 1.0))
    assert abs(float(loss) - float(spec)) < 2e-5 * float(spec)
    assert torch.isfinite(d).all() and float(d[1].abs().max()) == 0.0
    assert rel_l2(d[0], ar.grad[0]) < 5e-5 and rel_l2(d[2], ar.grad[2]) < 5e-5


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_autoencoderkl_vs_reference_local_autoencoder_golden(golden_dir, dtype):
    """The engine's AutoencoderKL at the production channels [32,32,64] against golden vectors produced by the REFERENCE's own local
    autoencoder (src/models/ae_kl.py:123-291, mid-attention removed = the MONAI structure, GroupNorm(32)): forward (recon, mu, sigma,
    KL), input gradient and all 126 parameter gradients of recon.dy + 0.3 KL.  tests/golden/make_golden_r2.py::case_aekl_twin."""
    import os
    import gpu_util as G
    from eegldm.models import AutoencoderKL
    g = np.load(os.path.join(golden_dir, "aekl_twin_32_32_64.npz"))
    cfg = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=32)
    sw, sx, se, sdy = [int(v) for v in g["seeds"]]
    net = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype=dtype, **cfg)
    assert list(net.entries.keys()) == [str(k) for k in g["keys"]]
    net.load_state_dict({k: torch.from_numpy(gen_param(sw, k, shape)) for k, (_o, _n, shape) in net.entries.items()})
    x = torch.from_numpy(eeg_windows(2, seed=sx, length=256, pad=8)); eps = torch.from_numpy(normal((2, 1, 64), seed=se))
    dy = torch.from_numpy(normal((2, 1, 256), seed=sdy))
    klo = torch.zeros(1, device=net.device)
    recon, mu, sg = net(x, eps=eps, kl_out=klo)
    net.zero_grad()
    dx = net.backward(dy, kl_weight=0.3, need_dx=True)
    grads = net.grad_dict()
    f32 = dtype == "float32"
    if f32:
        G.assert_close(recon, g["recon"], rtol=2e-4, atol=5e-5, name="recon"); G.assert_close(mu, g["z_mu"], rtol=2e-4, atol=5e-5, name="mu")
        G.assert_close(sg, g["z_sigma"], rtol=2e-4, atol=5e-5, name="sigma"); G.assert_close(dx, g["dx"], rtol=2e-3, atol=5e-4, name="dx")
        assert abs(float(klo) - float(g["kl"])) < 1e-4 * abs(float(g["kl"]))
    else:
        assert rel_l2(recon, g["recon"]) < 4e-2 and rel_l2(mu, g["z_mu"]) < 4e-2 and rel_l2(dx, g["dx"]) < 8e-2
    gscale = max(float(g["g_l2:" + k]) for k in net.entries)
    worst = 0.0
    for k in net.entries:
        gr = grads[k].double().reshape(-1).cpu(); l2 = float(g["g_l2:" + k]); floor = (1e-3 if f32 else 3e-2) * gscale
        rel = abs(float(gr.norm()) - l2) / (l2 + floor)
        head = g["g_head:" + k].astype(np.float64)
        he = float(np.linalg.norm(gr[:16].numpy() - head)) / (float(np.linalg.norm(head)) + floor)
        worst = max(worst, rel, he)
        assert rel < (2e-3 if f32 else 8e-2) and he < (3e-3 if f32 else 0.15), f"{k}: norm {rel:.2e} head {he:.2e}"
    print(f"aekl twin {dtype}: recon {rel_l2(recon, g['recon']):.2e} dx {rel_l2(dx, g['dx']):.2e} worst grad digest {worst:.2e}")


@pytest.mark.parametrize("fixture", ["aekl_twin_32_32_64_g1.npz", "aekl_twin_2_2_4_g1.npz"])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_autoencoderkl_one_group_vs_reference_local_autoencoder_golden(golden_dir, dtype, fixture, env_switches):
    """Round 5: the same twin with the reference's `Normalize` (src/models/ae_kl.py:15-16) patched to ONE group =
    `norm_num_groups: 1` of every AutoencoderKL config -- so the G = 1 kernels (gn_flat_* / gn_flat_fwd_wide, the whole-network LDS kernels of
    aekl_thin.hip for [2,2,4], and, at [32,32,64], the layer-by-layer path) are pinned to REFERENCE code, not only to the oracle.
    tests/golden/make_golden_r5.py.  [2,2,4] runs twice: whole-network kernels and EEGLDM_AEKL_NO_THIN=1."""
    import os
    import gpu_util as G
    from eegldm.models import AutoencoderKL
    g = np.load(os.path.join(golden_dir, fixture))
    nc = [int(v) for v in g["num_channels"]]; B, L = int(g["B"]), int(g["L"])
    cfg = dict(num_channels=nc, latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    sw, sx, se, sdy = [int(v) for v in g["seeds"]]
    x = torch.from_numpy(eeg_windows(B, seed=sx, length=L, pad=8)); eps = torch.from_numpy(normal((B, 1, L // 4), seed=se))
    dy = torch.from_numpy(normal((B, 1, L), seed=sdy))
    f32 = dtype == "float32"
    for no_thin in ([None, "1"] if nc[0] == 2 else [None]):
        env_switches(EEGLDM_AEKL_NO_THIN=no_thin)
        net = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype=dtype, **cfg)
        assert list(net.entries.keys()) == [str(k) for k in g["keys"]]
        net.load_state_dict({k: torch.from_numpy(gen_param(sw, k, shape)) for k, (_o, _n, shape) in net.entries.items()})
        klo = torch.zeros(1, device=net.device)
        recon, mu, sg = net(x, eps=eps, kl_out=klo)
        net.zero_grad()
        dx = net.backward(dy, kl_weight=0.3, need_dx=True)
        grads = net.grad_dict()
        thin_f32 = nc[0] == 2 and no_thin is None        # the whole-network kernels compute in fp32 whatever the engine dtype
        if f32 or thin_f32:
            G.assert_close(recon, g["recon"], rtol=2e-4, atol=5e-5, name="recon"); G.assert_close(mu, g["z_mu"], rtol=2e-4, atol=5e-5, name="mu")
            G.assert_close(sg, g["z_sigma"], rtol=2e-4, atol=5e-5, name="sigma"); G.assert_close(dx, g["dx"], rtol=2e-3, atol=5e-4, name="dx")
            assert abs(float(klo) - float(g["kl"])) < 1e-4 * abs(float(g["kl"]))
        else:
            assert rel_l2(recon, g["recon"]) < 4e-2 and rel_l2(mu, g["z_mu"]) < 4e-2 and rel_l2(dx, g["dx"]) < 8e-2
        exact = f32 or thin_f32
        gscale = max(float(g["g_l2:" + k]) for k in net.entries)
        worst = 0.0
        for k in net.entries:
            gr = grads[k].double().reshape(-1).cpu(); l2 = float(g["g_l2:" + k]); floor = (1e-3 if exact else 3e-2) * gscale
            rel = abs(float(gr.norm()) - l2) / (l2 + floor)
            head = g["g_head:" + k].astype(np.float64)
            he = float(np.linalg.norm(gr[:16].numpy() - head)) / (float(np.linalg.norm(head)) + floor)
            worst = max(worst, rel, he)
            # bf16 storage on the layer-by-layer path of the [2,2,4] model: 2-4 channel activations, single-element 1x1 weights -- one rounding
            # of a latent-head input moves such a gradient by 10 % (the fp32 run of the same path holds the tight bound)
            lim = (2e-3, 3e-3) if exact else ((0.2, 0.3) if nc[0] == 2 else (8e-2, 0.15))
            assert rel < lim[0] and he < lim[1], f"{k}: norm {rel:.2e} head {he:.2e}"
        print(f"aekl G=1 twin {fixture} {dtype} no_thin={no_thin}: recon {rel_l2(recon, g['recon']):.2e} dx {rel_l2(dx, g['dx']):.2e} worst grad digest {worst:.2e}")


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_patch_discriminator_vs_reference_local_discriminator_golden(golden_dir, dtype):
    """The engine's PatchDiscriminator against golden vectors of the REFERENCE's own local Discriminator (src/models/discriminator.py:15-84,
    kernel-4 convs swapped for the kernel-3 ones the config asks for): logits, input gradient, all parameter gradients, BatchNorm
    running statistics after one training forward.  tests/golden/make_golden_r2.py::case_disc_twin."""
    import os
    import gpu_util as G
    from eegldm.models import PatchDiscriminator
    g = np.load(os.path.join(golden_dir, "disc_twin_k3.npz"))
    sw, sx, sdy = [int(v) for v in g["seeds"]]
    net = PatchDiscriminator(**D_CFG, dtype=dtype)
    sd = {}
    for k in [str(k) for k in g["keys"]]:
        shape = net.entries[k][2] if k in net.entries else net.buf_entries[k][2]
        v = torch.from_numpy(gen_param(sw, k, shape))
        sd[k] = v * 2.0 if k.endswith("conv.weight") else v
    net.load_state_dict(sd)
    x = torch.from_numpy(normal((3, 1, 256), seed=sx))
    logits = net(x)[-1]
    dy = torch.from_numpy(normal(tuple(logits.shape), seed=sdy))
    net.zero_grad()
    dx = net.backward(dy, need_dx=True, in_shape=tuple(x.shape))
    f32 = dtype == "float32"
    if f32:
        G.assert_close(logits, g["logits"], rtol=2e-4, atol=5e-5, name="logits"); G.assert_close(dx, g["dx"], rtol=2e-3, atol=5e-4, name="dx")
    else:
        assert rel_l2(logits, g["logits"]) < 4e-2 and rel_l2(dx, g["dx"]) < 0.12
    grads, new = net.grad_dict(), net.state_dict()
    gscale = max(float(g[f]) for f in g.files if f.startswith("g_l2:"))
    for k in sd:
        if "running" in k:
            assert rel_l2(new[k], g["buf:" + k]) < (1e-5 if f32 else 2e-2), k
        elif "num_batches" in k:
            assert int(new[k]) == int(g["buf:" + k]) == 1
        else:
            gr = grads[k].double().reshape(-1).cpu(); l2 = float(g["g_l2:" + k]); floor = (1e-3 if f32 else 3e-2) * gscale
            assert abs(float(gr.norm()) - l2) / (l2 + floor) < (2e-3 if f32 else 8e-2), k
            head = g["g_head:" + k].astype(np.float64)
            assert float(np.linalg.norm(gr[:16].numpy() - head)) / (float(np.linalg.norm(head)) + floor) < (3e-3 if f32 else 0.15), k


@pytest.mark.parametrize("dtype,tag", [("bfloat16", "bf16"), ("float16", "f16")])
def test_16bit_autoencoder_is_as_close_to_fp32_as_the_reference_autocast_run(golden_dir, dtype, tag):
    """The production stage-1 autoencoder [32,32,64] (frozen encoder of the LDM step, decoder of sampling): the reference's local twin under
    torch.autocast(bfloat16) is its own reduced-precision run (tests/golden/make_golden_autocast.py -> aekl_autocast_bf16.npz holds that
    run's distance from the fp32 run); the bf16 engine is held to the same distance (factor 1.5: not the same rounding points)."""
    import os
    import gpu_util as G
    from eegldm.models import AutoencoderKL
    from oracle import aekl as A
    a = np.load(os.path.join(golden_dir, f"aekl_autocast_{tag}.npz"))
    g = np.load(os.path.join(golden_dir, "aekl_twin_32_32_64_g1.npz"))
    nc = [int(v) for v in g["num_channels"]]; B, L = int(g["B"]), int(g["L"])
    cfg = dict(num_channels=nc, latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    sw, sx, se, sdy = [int(v) for v in g["seeds"]]
    x = torch.from_numpy(eeg_windows(B, seed=sx, length=L, pad=8)); eps = torch.from_numpy(normal((B, 1, L // 4), seed=se))
    dy = torch.from_numpy(normal((B, 1, L), seed=sdy))
    net = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype=dtype, **cfg)
    sd = {k: torch.from_numpy(gen_param(sw, k, shape)) for k, (_o, _n, shape) in net.entries.items()}
    net.load_state_dict(sd)
    klo = torch.zeros(1, device=net.device)
    recon, mu, sg = net(x, eps=eps, kl_out=klo)
    net.zero_grad()
    dx = net.backward(dy, kl_weight=0.3, need_dx=True)
    grads = {k: v.float().cpu() for k, v in net.grad_dict().items()}
    # full fp32 parameter gradients from the oracle (pinned to the same twin golden)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ro, mo, so = A.forward(p, cfg, xr, eps)
    kl = 0.5 * torch.sum(mo.pow(2) + so.pow(2) - torch.log(so.pow(2)) - 1, dim=[1]); kl = torch.sum(kl) / kl.shape[0]
    ((ro * dy).sum() + 0.3 * kl).backward()
    np.testing.assert_allclose(ro.detach().numpy(), g["recon"], rtol=2e-4, atol=5e-5)
    rel = lambda u, v: float((torch.as_tensor(u).double().cpu() - torch.as_tensor(v).double()).norm() / (torch.as_tensor(v).double().norm() + 1e-30))
    e = dict(recon=rel(recon, g["recon"]), mu=rel(mu, g["z_mu"]), sigma=rel(sg, g["z_sigma"]), dx=rel(dx, g["dx"]))
    num = sum(float((grads[k].double() - p[k].grad.double()).norm() ** 2) for k in p); den = sum(float(p[k].grad.double().norm() ** 2) for k in p)
    e["grads"] = (num / den) ** 0.5
    assert [str(k) for k in a["keys"]] == list(p.keys())
    ge, gl = a["g_err"].astype(np.float64), a["g_l2"].astype(np.float64)
    r = dict(recon=float(a["recon_err"]), mu=float(a["mu_err"]), sigma=float(a["sigma_err"]), dx=float(a["dx_err"]),
             grads=float(np.sqrt(((ge * gl) ** 2).sum() / (gl ** 2).sum())))
    print(f"engine {tag} vs fp32:", {k: f"{v:.3e}" for k, v in e.items()}, f"| reference autocast-{tag} vs fp32:", {k: f"{v:.3e}" for k, v in r.items()})
    for k in e:
        assert e[k] <= 1.5 * r[k], (k, e[k], r[k])


@pytest.mark.parametrize("dtype,tag", [("bfloat16", "bf16"), ("float16", "f16")])
def test_16bit_discriminator_is_as_close_to_fp32_as_the_reference_autocast_run(golden_dir, dtype, tag):
    """PatchDiscriminator (training-mode BatchNorm): the reference's local twin under torch.autocast(bfloat16) against its fp32 run
    (disc_autocast_bf16.npz) beside the bf16 engine against the same fp32 numbers; factor 1.5 as for the UNet and the autoencoder."""
    import os
    import gpu_util as G
    from eegldm.models import PatchDiscriminator
    from oracle import aekl as A
    a = np.load(os.path.join(golden_dir, f"disc_autocast_{tag}.npz"))
    g = np.load(os.path.join(golden_dir, "disc_twin_k3.npz"))
    sw, sx, sdy = [int(v) for v in g["seeds"]]
    net = PatchDiscriminator(**D_CFG, dtype=dtype)
    sd = {}
    for k in [str(k) for k in g["keys"]]:
        shape = net.entries[k][2] if k in net.entries else net.buf_entries[k][2]
        v = torch.from_numpy(gen_param(sw, k, shape))
        sd[k] = v * 2.0 if k.endswith("conv.weight") else v
    net.load_state_dict(sd)
    x = torch.from_numpy(normal((3, 1, 256), seed=sx))
    logits = net(x)[-1]
    dy = torch.from_numpy(normal(tuple(logits.shape), seed=sdy))
    net.zero_grad()
    dx = net.backward(dy, need_dx=True, in_shape=tuple(x.shape))
    grads = {k: v.float().cpu() for k, v in net.grad_dict().items()}
    pkeys = [str(k) for k in a["keys"]]
    p = {k: (v.clone().requires_grad_(True) if k in pkeys else v.clone()) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    lo = A.disc_forward(p, D_CFG, xr, training=True)
    lo = lo[-1] if isinstance(lo, (list, tuple)) else lo
    (lo * dy).sum().backward()
    np.testing.assert_allclose(lo.detach().numpy(), g["logits"], rtol=2e-4, atol=5e-5)
    rel = lambda u, v: float((torch.as_tensor(u).double().cpu() - torch.as_tensor(v).double()).norm() / (torch.as_tensor(v).double().norm() + 1e-30))
    e = dict(logits=rel(logits, g["logits"]), dx=rel(dx, g["dx"]))
    num = sum(float((grads[k].double() - p[k].grad.double()).norm() ** 2) for k in pkeys); den = sum(float(p[k].grad.double().norm() ** 2) for k in pkeys)
    e["grads"] = (num / den) ** 0.5
    ge, gl = a["g_err"].astype(np.float64), a["g_l2"].astype(np.float64)
    r = dict(logits=float(a["logits_err"]), dx=float(a["dx_err"]), grads=float(np.sqrt(((ge * gl) ** 2).sum() / (gl ** 2).sum())))
    print(f"engine {tag} vs fp32:", {k: f"{v:.3e}" for k, v in e.items()}, f"| reference autocast-{tag} vs fp32:", {k: f"{v:.3e}" for k, v in r.items()})
    for k in e:
        assert e[k] <= 1.5 * r[k], (k, e[k], r[k])
