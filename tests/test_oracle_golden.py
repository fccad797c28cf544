"""Pins the oracle (oracle/unet.py, schedule values) against golden vectors
produced by importing the reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from param_gen import gen_param, normal
from make_golden_cases import UNET_CASES, UNET_OPTION_CASES
from oracle import losses as Ls
from oracle import unet as U

FWD = dict(rtol=1e-4, atol=1e-5)
GRAD = dict(rtol=1e-3, atol=1e-5)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_timestep_embedding(golden_dir):
    g = _load(golden_dir, "timestep_embedding.npz")
    e = U.timestep_embedding(torch.from_numpy(g["t"]), 128)
    np.testing.assert_allclose(e.numpy(), g["emb"], rtol=1e-6, atol=1e-6)


def test_qkv_attention(golden_dir):
    g = _load(golden_dir, "qkv_attention.npz")
    qkv = torch.from_numpy(normal(tuple(g["shape"]), seed=int(g["seed_qkv"]))).requires_grad_(True)
    a = U.qkv_attention(qkv)
    a.backward(torch.from_numpy(normal(tuple(a.shape), seed=int(g["seed_dy"]))))
    np.testing.assert_allclose(a.detach().numpy(), g["out"], **FWD)
    np.testing.assert_allclose(qkv.grad.numpy(), g["dqkv"], **GRAD)


RES = {"plain": (32, 32, ""), "skip": (32, 64, ""), "wide_in": (96, 32, ""), "down": (32, 32, "down"), "up": (64, 64, "up")}


@pytest.mark.parametrize("name", list(RES))
def test_resblock(golden_dir, name):
    g = _load(golden_dir, "resblocks.npz")
    ci, co, flag = RES[name]
    sw, sx, se, sdy = [int(v) for v in g[name + ":seeds"]]
    shapes = {"in_layers.0.weight": (ci,), "in_layers.0.bias": (ci,), "in_layers.2.weight": (co, ci, 3),
              "in_layers.2.bias": (co,), "emb_layers.1.weight": (co, 128), "emb_layers.1.bias": (co,),
              "out_layers.0.weight": (co,), "out_layers.0.bias": (co,), "out_layers.3.weight": (co, co, 3),
              "out_layers.3.bias": (co,)}
    if ci != co:
        shapes["skip_connection.weight"] = (co, ci, 1); shapes["skip_connection.bias"] = (co,)
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(normal((2, ci, 32), seed=sx)).requires_grad_(True)
    emb = torch.from_numpy(normal((2, 128), seed=se)).requires_grad_(True)
    y = U.resblock(sd, "", x, emb, up=flag == "up", down=flag == "down")
    y.backward(torch.from_numpy(normal(tuple(y.shape), seed=sdy)))
    np.testing.assert_allclose(y.detach().numpy(), g[name + ":y"], **FWD)
    np.testing.assert_allclose(x.grad.numpy(), g[name + ":dx"], **GRAD)
    np.testing.assert_allclose(emb.grad.numpy(), g[name + ":demb"], **GRAD)
    for k in shapes:
        np.testing.assert_allclose(sd[k].grad.numpy(), g[name + ":g:" + k], rtol=1e-3, atol=1e-4)


def test_attention_block(golden_dir):
    g = _load(golden_dir, "attention_block.npz")
    sw, sx, sdy = [int(v) for v in g["seeds"]]
    shapes = {"norm.weight": (64,), "norm.bias": (64,), "qkv.weight": (192, 64, 1), "qkv.bias": (192,),
              "proj_out.weight": (64, 64, 1), "proj_out.bias": (64,)}
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(normal((2, 64, 24), seed=sx)).requires_grad_(True)
    y = U.attention_block(sd, "", x)
    y.backward(torch.from_numpy(normal(tuple(y.shape), seed=sdy)))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], **FWD)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], **GRAD)
    for k in shapes:
        np.testing.assert_allclose(sd[k].grad.numpy(), g["g:" + k], rtol=1e-3, atol=1e-4)


RES_R6 = {"ssn_plain": (32, 32, ""), "ssn_skip": (32, 64, ""), "ssn_down": (32, 32, "down"), "ssn_up": (64, 64, "up")}


@pytest.mark.parametrize("name", list(RES_R6))
def test_resblock_scale_shift_norm(golden_dir, name):
    """ResBlock(use_scale_shift_norm=True), unet.py:318-322 (tests/golden/make_golden_r6.py)"""
    g = _load(golden_dir, "blocks_r6.npz")
    ci, co, flag = RES_R6[name]
    sw, sx, se, sdy = [int(v) for v in g[name + ":seeds"]]
    shapes = {"in_layers.0.weight": (ci,), "in_layers.0.bias": (ci,), "in_layers.2.weight": (co, ci, 3),
              "in_layers.2.bias": (co,), "emb_layers.1.weight": (2 * co, 128), "emb_layers.1.bias": (2 * co,),
              "out_layers.0.weight": (co,), "out_layers.0.bias": (co,), "out_layers.3.weight": (co, co, 3),
              "out_layers.3.bias": (co,)}
    if ci != co:
        shapes["skip_connection.weight"] = (co, ci, 1); shapes["skip_connection.bias"] = (co,)
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(normal((2, ci, 32), seed=sx)).requires_grad_(True)
    emb = torch.from_numpy(normal((2, 128), seed=se)).requires_grad_(True)
    y = U.resblock(sd, "", x, emb, up=flag == "up", down=flag == "down", scale_shift=True)
    y.backward(torch.from_numpy(normal(tuple(y.shape), seed=sdy)))
    np.testing.assert_allclose(y.detach().numpy(), g[name + ":y"], **FWD)
    np.testing.assert_allclose(x.grad.numpy(), g[name + ":dx"], **GRAD)
    np.testing.assert_allclose(emb.grad.numpy(), g[name + ":demb"], **GRAD)
    for k in shapes:
        np.testing.assert_allclose(sd[k].grad.numpy(), g[name + ":g:" + k], rtol=1e-3, atol=1e-4)


ATT_R6 = {"attn_heads4": (64, 4), "attn_headch16": (64, 4), "attn_heads2_c96": (96, 2)}


@pytest.mark.parametrize("name", list(ATT_R6))
def test_attention_block_heads(golden_dir, name):
    """AttentionBlock with several heads (num_heads / num_head_channels, unet.py:132-166; QKVAttentionLegacy's per-head [q | k | v] layout)"""
    g = _load(golden_dir, "blocks_r6.npz")
    c, heads = ATT_R6[name]
    sw, sx, sdy = [int(v) for v in g[name + ":seeds"]]
    shapes = {"norm.weight": (c,), "norm.bias": (c,), "qkv.weight": (3 * c, c, 1), "qkv.bias": (3 * c,),
              "proj_out.weight": (c, c, 1), "proj_out.bias": (c,)}
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(normal((2, c, 24), seed=sx)).requires_grad_(True)
    y = U.attention_block(sd, "", x, heads)
    y.backward(torch.from_numpy(normal(tuple(y.shape), seed=sdy)))
    np.testing.assert_allclose(y.detach().numpy(), g[name + ":y"], **FWD)
    np.testing.assert_allclose(x.grad.numpy(), g[name + ":dx"], **GRAD)
    for k in shapes:
        np.testing.assert_allclose(sd[k].grad.numpy(), g[name + ":g:" + k], rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize("name", list(UNET_CASES) + list(UNET_OPTION_CASES))
def test_unet_vs_reference(golden_dir, name):
    g = _load(golden_dir, f"unet_{name}.npz")
    cfg, B, L = (UNET_CASES.get(name) or UNET_OPTION_CASES[name])
    sw, sx, _st, sdy = [int(v) for v in g["seeds"]]
    shapes = U.unet_param_shapes(cfg)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]          # state-dict key order/naming pinned
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"])
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(normal((B, cfg["in_channels"], L), seed=sx)).requires_grad_(True)
    y = U.unet_forward(sd, cfg, x, torch.from_numpy(g["t"]))
    y.backward(torch.from_numpy(normal(tuple(y.shape), seed=sdy)))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], **FWD)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], **GRAD)
    for k in shapes:
        gr = sd[k].grad.double().reshape(-1)
        np.testing.assert_allclose(gr[:32].float().numpy(), g["g_head:" + k], rtol=2e-3, atol=2e-4)
        assert abs(float(gr.norm()) - float(g["g_l2:" + k])) <= 1e-3 * float(g["g_l2:" + k]) + 1e-5


def test_full_size_key_list(golden_dir):
    g = _load(golden_dir, "unet_full_keys.npz")
    cfg = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2,
               attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)
    shapes = U.unet_param_shapes(cfg)
    assert [str(k) for k in g["keys"]] == list(shapes.keys())
    assert [",".join(str(d) for d in s) for s in shapes.values()] == [str(s) for s in g["shapes"]]
    assert int(g["n_params"]) == 30533121 == sum(int(np.prod(s)) for s in shapes.values())


@pytest.mark.parametrize("name,b0,b1", [("train_0.0015_0.0195", 0.0015, 0.0195), ("sample_0.0015_0.0205", 0.0015, 0.0205)])
def test_scaled_linear_schedule(golden_dir, name, b0, b1):
    g = _load(golden_dir, "schedules.npz")
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, b0, b1).double().numpy()
    np.testing.assert_allclose(acp, g[name + ":alphas_cumprod"], rtol=2e-5)
    # SURVEY §8c known answers
    ref = {"train_0.0015_0.0195": (0.99850, 0.115844, 1.42304e-4), "sample_0.0015_0.0205": (0.99850, 0.108576, 9.69109e-5)}[name]
    np.testing.assert_allclose(acp[[0, 499, 999]], ref, rtol=2e-4)


# ------------------------------------------------------------------ round 2: DDPM.q_sample / p_sample and the full-size UNet
def test_add_noise_vs_reference_q_sample(golden_dir):
    """oracle add_noise == DDPM.q_sample (/root/reference/src/models/ldm.py:392-408) on the reference's own sqrt tables."""
    g = _load(golden_dir, "ddpm_steps.npz")
    sx, sn = [int(v) for v in g["q_sample:seeds"]]
    x0, nz = torch.from_numpy(normal((3, 1, 64), seed=sx)), torch.from_numpy(normal((3, 1, 64), seed=sn))
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0195)      # local "linear" == sqrt-space schedule
    np.testing.assert_allclose(acp.numpy(), g["alphas_cumprod"], rtol=2e-5)
    out = Ls.add_noise(acp, x0, nz, torch.from_numpy(g["q_sample:t"]))
    np.testing.assert_allclose(out.numpy(), g["q_sample:out"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("par,pred", [("eps", "epsilon"), ("x0", "sample")])
@pytest.mark.parametrize("tval", [0, 1, 500, 999])
@pytest.mark.parametrize("clip", [False, True])
def test_ddpm_step_vs_reference_p_sample(golden_dir, par, pred, tval, clip):
    """oracle ddpm_step (the MONAI DDPMScheduler.step formula, fixed_small variance) == DDPM.p_sample
    (/root/reference/src/models/ldm.py:311-357): posterior mean coefficients, clipped log-variance, no noise at t = 0."""
    g = _load(golden_dir, "ddpm_steps.npz")
    so, sx, sz = [int(v) for v in g[f"p_sample:{par}:seeds"]]
    mo = torch.from_numpy(normal((3, 1, 64), seed=so)) * (1.0 if par == "eps" else 0.6)
    xt, zn = torch.from_numpy(normal((3, 1, 64), seed=sx)), torch.from_numpy(normal((3, 1, 64), seed=sz))
    betas = Ls.make_betas("scaled_linear_beta", 1000, 0.0015, 0.0195)
    np.testing.assert_allclose(betas.numpy(), g["betas"], rtol=2e-5)
    acp = torch.cumprod(1 - betas, 0)
    prev, x0 = Ls.ddpm_step(acp, betas, mo, tval, xt, zn, pred, clip)
    # 1/sqrt(acp[999]) ~ 84 amplifies fp32 rounding of the schedule tables in the unclipped x0
    tol = dict(rtol=2e-4, atol=2e-4 if tval < 999 else 5e-3)
    np.testing.assert_allclose(x0.numpy(), g[f"p_sample:{par}:t{tval}:clip{int(clip)}:x0"], **tol)
    np.testing.assert_allclose(prev.numpy(), g[f"p_sample:{par}:t{tval}:clip{int(clip)}:prev"], **tol)


def test_ddpm_step_with_unet_vs_reference(golden_dir):
    """One step of the 1000-step ancestral sampler with the real (tiny) reference UNet inside p_sample."""
    g = _load(golden_dir, "ddpm_steps.npz")
    cfg, _B, _L = UNET_CASES["tiny_l64"]
    sd = {k: torch.from_numpy(gen_param(42, k, s)) for k, s in U.unet_param_shapes(cfg).items()}
    _so, sx, sz = [int(v) for v in g["p_sample:eps:seeds"]]
    xt, zn = torch.from_numpy(normal((3, 1, 64), seed=sx)), torch.from_numpy(normal((3, 1, 64), seed=sz))
    t = int(g["p_sample_unet:t"])
    betas = Ls.make_betas("scaled_linear_beta", 1000, 0.0015, 0.0195); acp = torch.cumprod(1 - betas, 0)
    with torch.no_grad():
        out = U.unet_forward(sd, cfg, xt, torch.full((3,), t, dtype=torch.int64))
    prev, x0 = Ls.ddpm_step(acp, betas, out, t, xt, zn, "epsilon", False)
    np.testing.assert_allclose(x0.numpy(), g["p_sample_unet:x0"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(prev.numpy(), g["p_sample_unet:prev"], rtol=1e-3, atol=1e-3)


def test_unet_full_size_vs_reference(golden_dir):
    """The BASELINE UNet itself (config_ldm.yaml: model_channels 128, 30 533 121 parameters, Cin up to 1024), B=2, L=768:
    oracle forward, input gradient and every parameter gradient against the imported reference."""
    from make_golden_cases import UNET_FULL
    g = _load(golden_dir, "unet_full_l768.npz")
    cfg, B, L = UNET_FULL
    sw, sx, _st, sdy = [int(v) for v in g["seeds"]]
    shapes = U.unet_param_shapes(cfg)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]] and int(g["n_params"]) == 30533121
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(normal((B, 1, L), seed=sx)).requires_grad_(True)
    y = U.unet_forward(sd, cfg, x, torch.from_numpy(g["t"]))
    y.backward(torch.from_numpy(normal(tuple(y.shape), seed=sdy)))
    np.testing.assert_allclose(y.detach().numpy(), g["y"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=2e-3, atol=2e-5)
    for k in shapes:
        gr = sd[k].grad.double().reshape(-1)
        l2 = float(g["g_l2:" + k])
        np.testing.assert_allclose(gr[:16].float().numpy(), g["g_head:" + k], rtol=2e-3, atol=2e-4 * max(1.0, l2 / np.sqrt(gr.numel())))
        assert abs(float(gr.norm()) - l2) <= 1e-3 * l2 + 1e-5, k


def test_aekl_oracle_vs_reference_local_autoencoder(golden_dir):
    """oracle/aekl.py (encode -> heads -> clamp -> sigma -> reparameterise -> decode, and its autograd backward incl. the KL term)
    against the reference's OWN local AutoencoderKL (/root/reference/src/models/ae_kl.py:123-291) with its mid-attention blocks
    removed -- exactly the block structure of the MONAI model the configs instantiate, at num_channels [32,32,64], latent 1,
    GroupNorm(32) (the configs use norm_num_groups = 1: a parameter here).  Pins block order, right-pad stride-2 downsample,
    nearest x2 + conv upsample, clamp(-30, 20), sigma = exp(log_var / 2), post_quant_conv placement."""
    from oracle import aekl as A
    from param_gen import eeg_windows
    g = _load(golden_dir, "aekl_twin_32_32_64.npz")
    cfg = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=32)
    sw, sx, se, sdy = [int(v) for v in g["seeds"]]
    shapes = A.aekl_param_shapes(cfg)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(eeg_windows(2, seed=sx, length=256, pad=8)).requires_grad_(True)
    eps = torch.from_numpy(normal((2, 1, 64), seed=se)); dy = torch.from_numpy(normal((2, 1, 256), seed=sdy))
    recon, mu, sg = A.forward(sd, cfg, x, eps)
    kl = Ls.kl_loss(mu, sg)
    ((recon * dy).sum() + 0.3 * kl).backward()
    np.testing.assert_allclose(recon.detach().numpy(), g["recon"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(mu.detach().numpy(), g["z_mu"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(sg.detach().numpy(), g["z_sigma"], rtol=2e-4, atol=2e-5)
    assert abs(float(kl) - float(g["kl"])) < 1e-5 * abs(float(g["kl"]))
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=2e-3, atol=2e-4)
    gscale = max(float(g["g_l2:" + k]) for k in shapes)
    for k in shapes:
        gr = sd[k].grad.double().reshape(-1); l2 = float(g["g_l2:" + k])
        assert abs(float(gr.norm()) - l2) <= 1e-3 * l2 + 1e-5 * gscale, k
        np.testing.assert_allclose(gr[:16].float().numpy(), g["g_head:" + k], rtol=2e-3, atol=2e-4 * max(1.0, gscale / 100))


@pytest.mark.parametrize("fixture", ["aekl_twin_32_32_64_g1.npz", "aekl_twin_2_2_4_g1.npz"])
def test_aekl_oracle_one_group_vs_reference_local_autoencoder(golden_dir, fixture):
    """The same twin with the reference's Normalize (ae_kl.py:15-16) patched to ONE group (norm_num_groups: 1, what every AutoencoderKL
    config asks for), at the production width and at [2,2,4] (config_aekl_eeg_2_2_4_spec.yaml): tests/golden/make_golden_r5.py."""
    from oracle import aekl as A
    from param_gen import eeg_windows
    g = _load(golden_dir, fixture)
    nc = [int(v) for v in g["num_channels"]]; B, L = int(g["B"]), int(g["L"])
    cfg = dict(num_channels=nc, latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    sw, sx, se, sdy = [int(v) for v in g["seeds"]]
    shapes = A.aekl_param_shapes(cfg)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]
    sd = {k: torch.from_numpy(gen_param(sw, k, s)).requires_grad_(True) for k, s in shapes.items()}
    x = torch.from_numpy(eeg_windows(B, seed=sx, length=L, pad=8)).requires_grad_(True)
    eps = torch.from_numpy(normal((B, 1, L // 4), seed=se)); dy = torch.from_numpy(normal((B, 1, L), seed=sdy))
    recon, mu, sg = A.forward(sd, cfg, x, eps)
    kl = Ls.kl_loss(mu, sg)
    ((recon * dy).sum() + 0.3 * kl).backward()
    np.testing.assert_allclose(recon.detach().numpy(), g["recon"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(mu.detach().numpy(), g["z_mu"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(sg.detach().numpy(), g["z_sigma"], rtol=2e-4, atol=2e-5)
    assert abs(float(kl) - float(g["kl"])) < 1e-5 * abs(float(g["kl"]))
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=2e-3, atol=2e-4)
    gscale = max(float(g["g_l2:" + k]) for k in shapes)
    for k in shapes:
        gr = sd[k].grad.double().reshape(-1); l2 = float(g["g_l2:" + k])
        assert abs(float(gr.norm()) - l2) <= 1e-3 * l2 + 1e-5 * gscale, k
        np.testing.assert_allclose(gr[:16].float().numpy(), g["g_head:" + k], rtol=2e-3, atol=2e-4 * max(1.0, gscale / 100))


def _disc_twin_inputs(g):
    from oracle import aekl as A
    dcfg = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
    sw, sx, sdy = [int(v) for v in g["seeds"]]
    shapes = A.disc_param_shapes(dcfg)
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]
    sd = {}
    for k, s in shapes.items():
        v = torch.from_numpy(gen_param(sw, k, s))
        sd[k] = v * 2.0 if k.endswith("conv.weight") else v
    return dcfg, shapes, sd, torch.from_numpy(normal((3, 1, 256), seed=sx)), sdy


def test_discriminator_oracle_vs_reference_local_discriminator(golden_dir):
    """oracle disc_forward (+ autograd) against the reference's own local PatchGAN Discriminator (/root/reference/src/models/
    discriminator.py:15-84) with its kernel-4 convs swapped for kernel-3 ones (what config_aekl_eeg.yaml:30-40 asks MONAI for):
    layer sequence, strides, bias placement, LeakyReLU(0.2), train-mode BatchNorm incl. the running-statistics update."""
    from oracle import aekl as A
    g = _load(golden_dir, "disc_twin_k3.npz")
    dcfg, shapes, sd, x, sdy = _disc_twin_inputs(g)
    sd = {k: (v.requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v) for k, v in sd.items()}
    x.requires_grad_(True)
    running = {}
    logits = A.disc_forward(sd, dcfg, x, True, running)[-1]
    (logits * torch.from_numpy(normal(tuple(logits.shape), seed=sdy))).sum().backward()
    np.testing.assert_allclose(logits.detach().numpy(), g["logits"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(x.grad.numpy(), g["dx"], rtol=2e-3, atol=2e-4)
    for k in shapes:
        if "running" in k:
            np.testing.assert_allclose(running[k].numpy(), g["buf:" + k], rtol=1e-5, atol=1e-6)
        elif "num_batches" not in k:
            gr = sd[k].grad.double().reshape(-1); l2 = float(g["g_l2:" + k])
            assert abs(float(gr.norm()) - l2) <= 1e-3 * l2 + 1e-5, k
            np.testing.assert_allclose(gr[:16].float().numpy(), g["g_head:" + k], rtol=2e-3, atol=1e-4 * max(1.0, l2))
