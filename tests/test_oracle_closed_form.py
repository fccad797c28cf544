"""CPU: known-answer tests for the oracle pieces whose reference implementation (monai-generative)
is absent -- SURVEY.md 8c list -- plus structural checks of the AutoencoderKL / PatchDiscriminator
restatement against the reference's local structural twin where their configurations coincide."""
import math
import sys

import numpy as np
import torch

from oracle import aekl as A
from oracle import losses as Ls
from param_gen import gen_param, normal, eeg_windows


def test_jukebox_closed_forms():
    x = torch.from_numpy(eeg_windows(2, seed=3))
    assert float(Ls.jukebox_loss(x, x)) == 0.0
    # Parseval with ortho norm: sum |FFT|^2 == sum x^2, so loss(x, 0) == sum x^2
    np.testing.assert_allclose(float(Ls.jukebox_loss(x, torch.zeros_like(x))), float((x ** 2).sum()), rtol=1e-5)
    # pure cosine of amplitude a at bin k: two bins of amplitude a*sqrt(N)/2
    N, k, a = 3072, 40, 0.3
    c = (a * torch.cos(2 * math.pi * k * torch.arange(N) / N)).reshape(1, 1, N)
    f = torch.fft.fftn(c, dim=(1, 2), norm="ortho").abs().reshape(-1)
    np.testing.assert_allclose(f[k], a * math.sqrt(N) / 2, rtol=1e-4); np.testing.assert_allclose(f[N - k], a * math.sqrt(N) / 2, rtol=1e-4)
    assert float(f.sum() - f[k] - f[N - k]) < 1e-2


def test_kl_and_lsgan_closed_forms():
    assert float(Ls.kl_loss(torch.zeros(2, 1, 8), torch.ones(2, 1, 8))) == 0.0
    mu = torch.full((2, 1, 4), 0.5); sg = torch.full((2, 1, 4), 2.0)
    want = 0.5 * (0.25 + 4 - math.log(4) - 1) * 4           # sum over the 4 positions, mean over batch
    np.testing.assert_allclose(float(Ls.kl_loss(mu, sg)), want, rtol=1e-6)
    assert float(Ls.patch_adv_loss(torch.ones(2, 1, 5), True, False)) == 0.0
    np.testing.assert_allclose(float(Ls.patch_adv_loss(-torch.ones(2, 1, 5), True, False)), (-0.05 - 1) ** 2, rtol=1e-6)
    np.testing.assert_allclose(float(Ls.patch_adv_loss(-torch.ones(2, 1, 5), False, True)), 0.05 ** 2, rtol=1e-5)
    # the generator always targets "real"
    assert float(Ls.patch_adv_loss(torch.ones(2, 1, 5), False, False)) == 0.0


def test_scheduler_closed_forms():
    acp_lin = Ls.alphas_cumprod("linear_beta", 1000, 0.0015, 0.0195).double().numpy()
    np.testing.assert_allclose(acp_lin[[0, 499, 999]], [0.99850, 0.0493666, 2.56920e-5], rtol=3e-4)
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0205)
    x0 = torch.from_numpy(normal((2, 1, 16), seed=1)); e = torch.from_numpy(normal((2, 1, 16), seed=2))
    t = torch.tensor([980, 980])
    xt = Ls.add_noise(acp, x0, e, t)
    # DDIM (eta 0) with the true epsilon reproduces x0 exactly, for both parameterisations
    _prev, px0 = Ls.ddim_step(acp, e, 980, xt, 1000, 50, "epsilon", clip_sample=False)
    np.testing.assert_allclose(px0.numpy(), x0.numpy(), atol=2e-4)
    v = Ls.get_velocity(acp, x0, e, t)
    prev_v, px0v = Ls.ddim_step(acp, v, 980, xt, 1000, 50, "v_prediction", clip_sample=False)
    np.testing.assert_allclose(px0v.numpy(), x0.numpy(), atol=2e-4)
    prev_e, _ = Ls.ddim_step(acp, e, 980, xt, 1000, 50, "epsilon", clip_sample=False)
    np.testing.assert_allclose(prev_v.numpy(), prev_e.numpy(), atol=2e-4)
    # last step lands on x0 (alpha_prev = 1)
    last, _ = Ls.ddim_step(acp, e, 0, Ls.add_noise(acp, x0, e, torch.tensor([0, 0])), 1000, 50, "epsilon", clip_sample=False)
    np.testing.assert_allclose(last.numpy(), x0.numpy(), atol=1e-5)
    np.testing.assert_array_equal(Ls.ddim_timesteps(1000, 50)[[0, 1, -1]], [980, 960, 0])


def test_aekl_shapes_and_param_counts():
    for nc, lat, want in [([32, 32, 64], 1, 174184), ([32, 32, 64], 3, 174984), ([2, 2, 4], 1, 934)]:
        cfg = dict(num_channels=nc, latent_channels=lat, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
        shapes = A.aekl_param_shapes(cfg)
        assert sum(int(np.prod(s)) for s in shapes.values()) == want       # SURVEY Appendix B recount
        sd = {k: torch.from_numpy(gen_param(1, k, s)) for k, s in shapes.items()}
        x = torch.from_numpy(eeg_windows(2, seed=1))
        recon, mu, sg = A.forward(sd, cfg, x, torch.zeros(2, lat, 768))
        assert recon.shape == (2, 1, 3072) and mu.shape == sg.shape == (2, lat, 768)
        assert float(sg.min()) > 0
    assert sum(int(np.prod(s)) for k, s in A.disc_param_shapes(dict(num_channels=64, num_layers_d=3, kernel_size=3, in_channels=1, out_channels=1)).items()
               if "running" not in k and "num_batches" not in k) == 519681


def test_aekl_blocks_match_reference_local_twin():
    """Where the reference's own (legacy) autoencoder coincides with the MONAI configuration -- ResBlock,
    right-pad stride-2 Downsample, nearest+conv Upsample (/root/reference/src/models/ae_kl.py:20-80) -- the
    restatement is checked against it directly (GroupNorm(32) there, so 32+ channel blocks with groups=32)."""
    import os
    if not os.path.isdir("/root/reference/src"):
        import pytest
        pytest.skip("reference tree not present (GPU box)")
    sys.path.insert(0, "/root/reference/src")
    from models import ae_kl as R
    import torch.nn.functional as F
    torch.manual_seed(0)
    x = torch.randn(2, 32, 40)
    rb = R.ResBlock(32, 64)
    sd = {"b.norm1.weight": rb.norm1.weight, "b.norm1.bias": rb.norm1.bias, "b.conv1.conv.weight": rb.conv1.weight, "b.conv1.conv.bias": rb.conv1.bias,
          "b.norm2.weight": rb.norm2.weight, "b.norm2.bias": rb.norm2.bias, "b.conv2.conv.weight": rb.conv2.weight, "b.conv2.conv.bias": rb.conv2.bias,
          "b.nin_shortcut.conv.weight": rb.nin_shortcut.weight, "b.nin_shortcut.conv.bias": rb.nin_shortcut.bias}
    np.testing.assert_allclose(A._resblock(sd, "b", x, 32).detach().numpy(), rb(x).detach().numpy(), rtol=1e-4, atol=1e-5)
    dn = R.Downsample(32)
    got = F.conv1d(F.pad(x, (0, 1)), dn.conv.weight, dn.conv.bias, stride=2)
    np.testing.assert_allclose(got.detach().numpy(), dn(x).detach().numpy(), rtol=1e-5, atol=1e-6)
    up = R.Upsample(32)
    got = F.conv1d(F.interpolate(x, scale_factor=2.0, mode="nearest"), up.conv.weight, up.conv.bias, padding=1)
    np.testing.assert_allclose(got.detach().numpy(), up(x).detach().numpy(), rtol=1e-5, atol=1e-6)


def test_grad_scaler_update_rule_matches_torch():
    """The oracle's GradScaler.update restatement and the product's host-side class against torch.amp.GradScaler itself
    (CPU device): same scale after every step of a finite / inf gradient sequence, and skipped steps leave the parameter alone."""
    from oracle import steps as S
    import eegldm
    from eegldm.training import GradScaler
    seq = [False, False, True, False, False, False, True, True, False, False, False, False]
    ref = torch.amp.GradScaler("cpu", init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    p = torch.nn.Parameter(torch.ones(4)); opt = torch.optim.SGD([p], lr=0.1)
    mine = GradScaler(init_scale=1024.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=3)
    scale, tracker = 1024.0, 0
    for bad in seq:
        opt.zero_grad()
        loss = (p * (float("inf") if bad else 1.0)).sum()
        before = p.detach().clone()
        ref.scale(loss).backward(); ref.step(opt); ref.update()
        assert torch.equal(before, p.detach()) == bad                # skipped exactly when the gradients overflowed
        scale, tracker = S.grad_scaler_update(scale, tracker, bad, 2.0, 0.5, 3)
        mine._found_inf = bad; mine.update()
        assert float(ref.get_scale()) == scale == mine.get_scale()
    assert mine.state_dict()["_growth_tracker"] == tracker
    off = GradScaler(enabled=False)
    assert off.get_scale() == 1.0 and off.state_dict() == {}
