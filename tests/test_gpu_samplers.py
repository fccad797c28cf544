"""-m gpu: scheduler steps, DiffusionInferer and the native sampler (eegldm_sample).

* DDPMScheduler.step vs the golden vectors of the reference's own DDPM.p_sample (tests/golden/ddpm_steps.npz) and
  add_noise vs DDPM.q_sample (/root/reference/src/models/ldm.py:311-357,392-408).
* DiffusionInferer.__call__ / .sample (training_diffusion.py:146, sample_trials_ddpm.py:99-102) vs the oracle.
* eegldm_sample (one native call, hipGraph replay of the UNet forward) vs the host-driven loop and vs oracle.steps.ddim_sample,
  at the reference's batch of ONE window per call and at a larger batch; DDIM and ancestral DDPM."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from make_golden_cases import UNET_CASES  # noqa: E402
from param_gen import gen_param, normal, timesteps  # noqa: E402

ACFG = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)


def test_add_noise_vs_reference_q_sample(golden_dir):
    import gpu_util as G
    from eegldm.schedulers import DDPMScheduler
    g = np.load(os.path.join(golden_dir, "ddpm_steps.npz"))
    sx, sn = [int(v) for v in g["q_sample:seeds"]]
    x0, nz = torch.from_numpy(normal((3, 1, 64), seed=sx)), torch.from_numpy(normal((3, 1, 64), seed=sn))
    s = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
    out = s.add_noise(original_samples=x0, noise=nz, timesteps=torch.from_numpy(g["q_sample:t"]))
    G.assert_close(out, g["q_sample:out"], rtol=2e-5, atol=2e-6, name="q_sample")


@pytest.mark.parametrize("par,pred", [("eps", "epsilon"), ("x0", "sample")])
@pytest.mark.parametrize("tval", [0, 1, 500, 999])
@pytest.mark.parametrize("clip", [False, True])
def test_ddpm_step_vs_reference_p_sample(golden_dir, par, pred, tval, clip):
    import gpu_util as G
    from eegldm.schedulers import DDPMScheduler
    g = np.load(os.path.join(golden_dir, "ddpm_steps.npz"))
    so, sx, sz = [int(v) for v in g[f"p_sample:{par}:seeds"]]
    mo = torch.from_numpy(normal((3, 1, 64), seed=so)) * (1.0 if par == "eps" else 0.6)
    xt, zn = torch.from_numpy(normal((3, 1, 64), seed=sx)), torch.from_numpy(normal((3, 1, 64), seed=sz))
    s = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195, prediction_type=pred,
                      clip_sample=clip)
    prev, x0 = s.step(mo, tval, xt, noise=zn)
    tol = dict(rtol=2e-4, atol=2e-4 if tval < 999 else 5e-3)      # 1/sqrt(acp[999]) ~ 84 amplifies table rounding in the unclipped x0
    G.assert_close(x0, g[f"p_sample:{par}:t{tval}:clip{int(clip)}:x0"], name="x0", **tol)
    G.assert_close(prev, g[f"p_sample:{par}:t{tval}:clip{int(clip)}:prev"], name="prev", **tol)


def test_ddpm_step_with_unet_vs_reference(golden_dir):
    """UNet forward + ancestral step == one p_sample of the reference with its own (tiny) UNet inside."""
    import gpu_util as G
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    g = np.load(os.path.join(golden_dir, "ddpm_steps.npz"))
    cfg, _B, _L = UNET_CASES["tiny_l64"]
    net = UNetModel(**cfg)
    net.load_state_dict({k: torch.from_numpy(gen_param(42, k, shape)) for k, (_o, _n, shape) in net.entries.items()})
    _so, sx, sz = [int(v) for v in g["p_sample:eps:seeds"]]
    xt, zn = torch.from_numpy(normal((3, 1, 64), seed=sx)), torch.from_numpy(normal((3, 1, 64), seed=sz))
    t = int(g["p_sample_unet:t"])
    s = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195, clip_sample=False)
    net.eval()
    prev, x0 = s.step(net(xt, timesteps=torch.full((3,), t)), t, xt, noise=zn)
    G.assert_close(x0, g["p_sample_unet:x0"], rtol=1e-3, atol=1e-3, name="x0")
    G.assert_close(prev, g["p_sample_unet:prev"], rtol=1e-3, atol=1e-3, name="prev")


def _tiny(seed):
    from eegldm.models import UNetModel
    from oracle import unet as U
    cfg, _B, _L = UNET_CASES["tiny_l64"]
    sd = {k: torch.from_numpy(gen_param(seed, k, s)) for k, s in U.unet_param_shapes(cfg).items()}
    net = UNetModel(**cfg); net.load_state_dict(sd)
    return cfg, sd, net


def test_diffusion_inferer_call_and_sample_vs_oracle():
    """DiffusionInferer.__call__ = add_noise + model (training_diffusion.py:146); .sample = the scheduler loop
    (sample_trials_ddpm.py:99-102) -- with a DDIM scheduler (deterministic) and with the ancestral DDPM scheduler (noise passed
    through a seeded torch generator on both sides)."""
    import gpu_util as G
    from eegldm.schedulers import DDIMScheduler, DDPMScheduler, DiffusionInferer
    from oracle import losses as Ls, unet as U
    cfg, sd, net = _tiny(171)
    B, L = 3, 64
    x = torch.from_numpy(normal((B, 1, L), seed=172)); nz = torch.from_numpy(normal((B, 1, L), seed=173)); t = torch.from_numpy(timesteps(B, seed=174))
    betas = Ls.make_betas("linear_beta", 1000, 0.0015, 0.0195); acp = torch.cumprod(1 - betas, 0)
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195, clip_sample=False)
    inf = DiffusionInferer(sched)
    net.eval()
    got = inf(inputs=x, diffusion_model=net, noise=nz, timesteps=t)
    with torch.no_grad():
        want = U.unet_forward(sd, cfg, Ls.add_noise(acp, x, nz, t), t)
    assert G.rel_l2(got, want) < 2e-5
    # .sample with DDIM, 8 steps
    ddim = DDIMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195, clip_sample=False)
    ddim.set_timesteps(8)
    got = inf.sample(input_noise=nz, diffusion_model=net, scheduler=ddim)
    xr = nz
    with torch.no_grad():
        for tt in Ls.ddim_timesteps(1000, 8):
            xr, _ = Ls.ddim_step(acp, U.unet_forward(sd, cfg, xr, torch.full((B,), int(tt))), int(tt), xr, 1000, 8, "epsilon", False)
    assert G.rel_l2(got, xr) < 1e-4, G.rel_l2(got, xr)
    # .sample with the ancestral scheduler on the last 6 timesteps (5..0), identical noise via per-step tensors
    sched.timesteps = torch.arange(5, -1, -1)
    noises = {tt: torch.from_numpy(normal((B, 1, L), seed=900 + tt)) for tt in range(6)}
    real_step = sched.step
    sched.step = lambda out, tt, img, **k: real_step(out, tt, img, noise=noises[int(tt)])
    got, inter = inf.sample(input_noise=nz, diffusion_model=net, scheduler=sched, save_intermediates=True, intermediate_steps=2)
    assert len(inter) == 3
    xr = nz
    with torch.no_grad():
        for tt in range(5, -1, -1):
            xr, _ = Ls.ddpm_step(acp, betas, U.unet_forward(sd, cfg, xr, torch.full((B,), tt)), tt, xr, noises[tt], "epsilon", False)
    assert G.rel_l2(got, xr) < 1e-4, G.rel_l2(got, xr)


@pytest.mark.parametrize("B", [1, 5])
@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
def test_native_sampler_matches_hostloop_and_oracle(B, pred):
    """eegldm_sample (graph replay on) == the Python-driven loop == oracle.steps.ddim_sample, incl. z / scale_factor and decode."""
    import gpu_util as G
    from eegldm.models import AutoencoderKL
    from eegldm.sampling import ddim_sample, ddim_sample_hostloop, make_sampling_scheduler
    from oracle import aekl as A, losses as Ls, steps as S
    cfg, usd, unet = _tiny(61)
    asd = {k: torch.from_numpy(gen_param(62, k, s)) for k, s in A.aekl_param_shapes(ACFG).items()}
    ae = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, **ACFG); ae.load_state_dict(asd)
    Ll, steps = 64, 10
    noise = torch.from_numpy(normal((B, 1, Ll), seed=63))
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0205)
    want, zl = S.ddim_sample(usd, cfg, asd, ACFG, noise, steps, acp, scale_factor=0.7, prediction_type=pred, crop=8)
    sched = make_sampling_scheduler(steps, prediction_type=pred)
    info = {}
    got, z = ddim_sample(unet, ae, sched, noise, scale_factor=0.7, crop=8, use_graph=True, info=info)
    assert info["graph"], "hipGraph capture of the UNet forward failed (the sampler fell back to eager launches)"
    assert got.shape == want.shape == (B, 1, 4 * Ll - 16)
    assert G.rel_l2(z, zl) < 5e-5 and G.rel_l2(got, want) < 5e-5, (G.rel_l2(z, zl), G.rel_l2(got, want))
    got2, z2 = ddim_sample(unet, ae, sched, noise, scale_factor=0.7, crop=8, use_graph=True)       # replay of the cached graph
    assert torch.equal(z2, z) and torch.equal(got2, got)
    got3, z3 = ddim_sample(unet, ae, sched, noise, scale_factor=0.7, crop=8, use_graph=False)
    got4, z4 = ddim_sample_hostloop(unet, ae, sched, noise, scale_factor=0.7, crop=8)
    # (the host loop forms 1/scale_factor - 1 in double, the native call in float: the decoded windows agree to rounding, the latents exactly)
    assert torch.equal(z3, z) and torch.equal(z4, z) and torch.equal(got3, got) and G.rel_l2(got4, got) < 1e-6
    # new weights must be picked up by the cached graph (it holds pointers, not values)
    unet.load_state_dict({k: v * 1.01 for k, v in usd.items()})
    z5 = ddim_sample(unet, ae, sched, noise, scale_factor=0.7, crop=8, use_graph=True)[1]
    z6 = ddim_sample_hostloop(unet, ae, sched, noise, scale_factor=0.7, crop=8)[1]
    assert torch.equal(z5, z6) and not torch.equal(z5, z)


def test_native_ancestral_sampler_pixel_space():
    """Pixel-space model, ancestral DDPM steps inside eegldm_sample (sample_trials_ddpm.py:99-104): the on-device Philox noise is
    reproduced on the host side of the test through eegldm_randn with the same (seed, offset) and fed to the oracle loop."""
    import gpu_util as G
    from eegldm.schedulers import DDPMScheduler
    from eegldm.sampling import ddim_sample
    from eegldm.training import randn
    from oracle import losses as Ls, unet as U
    cfg, sd, net = _tiny(181)
    B, L, seed = 2, 64, 77
    nz0 = torch.from_numpy(normal((B, 1, L), seed=182))
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195, clip_sample=True)
    sched.timesteps = torch.tensor([400, 300, 2, 1, 0])
    win, lat = ddim_sample(net, None, sched, nz0, crop=4, seed=seed)
    assert win.shape == (B, 1, L - 8) and torch.equal(win, lat[:, :, 4:-4])
    betas = Ls.make_betas("linear_beta", 1000, 0.0015, 0.0195); acp = torch.cumprod(1 - betas, 0)
    n = B * L
    xr = nz0
    with torch.no_grad():
        for i, tt in enumerate([400, 300, 2, 1, 0]):
            eps = randn(net.ctx, (B, 1, L), seed=seed, offset=i * ((n + 3) // 4)).cpu()
            xr, _ = Ls.ddpm_step(acp, betas, U.unet_forward(sd, cfg, xr, torch.full((B,), tt)), tt, xr, eps, "epsilon", True)
    assert G.rel_l2(lat, xr) < 1e-4, G.rel_l2(lat, xr)


@pytest.mark.parametrize("pred", ["epsilon", "sample", "v_prediction"])
@pytest.mark.parametrize("clip", [False, True])
@pytest.mark.parametrize("eta", [0.0, 0.3, 1.0])
def test_ddim_step_eta_vs_oracle(pred, clip, eta):
    """DDIMScheduler.step(eta) against the oracle's restatement of eq. 12 / 16; eta = 0 through the new entry equals the deterministic
    kernel bit for bit."""
    import gpu_util as G
    from eegldm.schedulers import DDIMScheduler, PRED
    from oracle import losses as Ls
    s = DDIMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205, prediction_type=pred,
                      clip_sample=clip)
    s.set_timesteps(50)
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0205)
    mo, x, nz = (torch.from_numpy(normal((3, 1, 96), seed=sd)) for sd in (31, 32, 33))
    for t in (980, 500, 20, 0):
        prev, x0 = s.step(mo, t, x, eta=eta, noise=nz)
        rp, r0 = Ls.ddim_step(acp, mo, t, x, 1000, 50, prediction_type=pred, clip_sample=clip, eta=eta, noise=nz)
        G.assert_close(prev, rp.numpy(), rtol=2e-5, atol=2e-5, name=f"prev t={t}")
        G.assert_close(x0, r0.numpy(), rtol=2e-5, atol=2e-5, name=f"x0 t={t}")
        if eta == 0.0:
            md, xd, nd = (a.to(G.DEV).contiguous() for a in (mo, x, nz))
            o1, o2 = torch.empty_like(md), torch.empty_like(md)
            a_t = float(s.alphas_cumprod[t]); a_prev = float(s.alphas_cumprod[t - 20]) if t >= 20 else 1.0
            G.check(G.lib.eegldm_ddim_step_eta(G.ctx().h, G.ptr(md), G.ptr(xd), None, a_t, a_prev, 0.0, PRED[pred],
                                               int(clip), G.ptr(o1), None, md.numel()))
            assert torch.equal(o1, prev.to(G.DEV))


def test_ddim_eta_one_is_the_ancestral_ddpm_step():
    """Size-independent property: with consecutive timesteps (50 -> 1000 inference steps: prev_t = t - 1) DDIM at eta = 1 IS the DDPM
    ancestral step with the posterior variance (sigma_t(1)^2 = (1 - a_prev) / (1 - a_t) beta_t), for any model output and the same noise."""
    import gpu_util as G
    from eegldm.schedulers import DDIMScheduler, DDPMScheduler
    kw = dict(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205, clip_sample=False)
    di, dp = DDIMScheduler(**kw), DDPMScheduler(**kw)
    di.set_timesteps(1000); dp.set_timesteps(1000)
    mo, x, nz = (torch.from_numpy(normal((256, 1, 768), seed=sd)) for sd in (41, 42, 43))
    for t in (999, 640, 77, 1):
        a, _ = di.step(mo, t, x, eta=1.0, noise=nz)
        b, _ = dp.step(mo, t, x, noise=nz)
        G.assert_close(a, b.cpu().numpy(), rtol=3e-5, atol=3e-5, name=f"t={t}")


@pytest.mark.parametrize("tval", [0, 1, 500, 999])
def test_ddpm_step_fixed_large_vs_oracle(tval):
    import gpu_util as G
    from eegldm.schedulers import DDPMScheduler
    from oracle import losses as Ls
    s = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195, variance_type="fixed_large")
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0195); betas = Ls.make_betas("scaled_linear_beta", 1000, 0.0015, 0.0195)
    mo, x, nz = (torch.from_numpy(normal((2, 1, 64), seed=sd)) for sd in (51, 52, 53))
    prev, x0 = s.step(mo, tval, x, noise=nz)
    rp, r0 = Ls.ddpm_step(acp, betas, mo, tval, x, nz, variance_type="fixed_large")
    G.assert_close(prev, rp.numpy(), rtol=2e-5, atol=2e-5, name="prev")
    G.assert_close(x0, r0.numpy(), rtol=2e-5, atol=2e-5, name="x0")
    with pytest.raises(NotImplementedError):
        DDPMScheduler(num_train_timesteps=1000, variance_type="learned")


@pytest.mark.parametrize("name", ["opt_all", "opt_heads2_up4", "opt_pool_resample"])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_native_sampler_with_constructor_options(name, dtype):
    """The native DDIM loop (pixel-space call: no autoencoder) over a UNet built with the optional constructor branches -- eager, graph
    replay and the Python-driven loop agree with each other, and the fp32 engine with the oracle's loop."""
    import gpu_util as G
    from make_golden_cases import UNET_OPTION_CASES
    from eegldm.models import UNetModel
    from eegldm.sampling import ddim_sample, ddim_sample_hostloop, make_sampling_scheduler
    from oracle import losses as Ls, unet as U
    cfg, _B, L = UNET_OPTION_CASES[name]
    net = UNetModel(**cfg, dtype=dtype)
    sd = {k: torch.from_numpy(gen_param(71, k, shape)) for k, (_o, _n, shape) in net.entries.items()}
    net.load_state_dict(sd)
    steps, B, C = 6, 2, cfg["in_channels"]
    noise = torch.from_numpy(normal((B, C, L), seed=72))
    sched = make_sampling_scheduler(steps)
    info = {}
    _w, z = ddim_sample(net, None, sched, noise, crop=0, use_graph=True, info=info)
    _w, z2 = ddim_sample(net, None, sched, noise, crop=0, use_graph=False)
    _w, z3 = ddim_sample_hostloop(net, None, sched, noise, crop=0)
    assert torch.isfinite(z).all() and torch.equal(z, z2) and torch.equal(z, z3)
    if dtype == "float32":
        acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0205)
        x = noise.clone()
        with torch.no_grad():
            for t in Ls.ddim_timesteps(1000, steps):
                out = U.unet_forward(sd, cfg, x, torch.full((B,), int(t), dtype=torch.int64))
                x, _ = Ls.ddim_step(acp, out, int(t), x, 1000, steps, clip_sample=False)      # sample_trials.py:136-145 builds the scheduler with clip_sample=False
        assert G.rel_l2(z, x) < 1e-4, G.rel_l2(z, x)
