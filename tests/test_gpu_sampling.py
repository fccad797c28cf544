"""-m gpu: DDIM sampling + decode vs oracle/steps.py::ddim_sample on identical weights and noise;
LDM train step with the frozen AutoencoderKL encoder in front (training.py:419-443)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from make_golden_cases import UNET_CASES  # noqa: E402
from param_gen import gen_param, normal, eeg_windows, timesteps  # noqa: E402


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1); b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.mark.parametrize("pred", ["epsilon", "v_prediction"])
def test_ddim_sampling_matches_oracle(pred):
    from eegldm.models import AutoencoderKL, UNetModel
    from eegldm.sampling import ddim_sample, make_sampling_scheduler
    from oracle import aekl as A, losses as Ls, steps as S, unet as U
    ucfg, _B, _L = UNET_CASES["tiny_l64"]
    acfg = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    usd = {k: torch.from_numpy(gen_param(61, k, s)) for k, s in U.unet_param_shapes(ucfg).items()}
    asd = {k: torch.from_numpy(gen_param(62, k, s)) for k, s in A.aekl_param_shapes(acfg).items()}
    B, Ll, steps = 3, 64, 10
    noise = torch.from_numpy(normal((B, 1, Ll), seed=63))
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0205)
    want, zl = S.ddim_sample(usd, ucfg, asd, acfg, noise, steps, acp, scale_factor=0.7, prediction_type=pred, crop=8)
    unet = UNetModel(**ucfg); unet.load_state_dict(usd)
    ae = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, **acfg); ae.load_state_dict(asd)
    sched = make_sampling_scheduler(steps, prediction_type=pred)
    got, z = ddim_sample(unet, ae, sched, noise, scale_factor=0.7, crop=8)
    assert got.shape == want.shape == (B, 1, 4 * Ll - 16)
    assert rel_l2(z, zl) < 5e-5 and rel_l2(got, want) < 5e-5, (rel_l2(z, zl), rel_l2(got, want))


def test_ldm_step_with_frozen_encoder_matches_oracle():
    from eegldm.models import AutoencoderKL, UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import ldm_train_step
    from oracle import aekl as A, losses as Ls, steps as S, unet as U
    ucfg, _B, _L = UNET_CASES["tiny_l64"]
    acfg = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    usd = {k: torch.from_numpy(gen_param(71, k, s)) for k, s in U.unet_param_shapes(ucfg).items()}
    asd = {k: torch.from_numpy(gen_param(72, k, s)) for k, s in A.aekl_param_shapes(acfg).items()}
    B, L = 3, 256
    x = torch.from_numpy(eeg_windows(B, seed=73, length=L, pad=8)); eps = torch.from_numpy(normal((B, 1, L // 4), seed=74))
    noise = torch.from_numpy(normal((B, 1, L // 4), seed=75)); t = torch.from_numpy(timesteps(B, seed=76)); scale = 1.7
    with torch.no_grad():
        mu, sg = A.encode(asd, acfg, x)
        lat_ref = (mu + eps * sg) * scale
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0195)
    loss_ref, grads_ref, _ = S.ldm_train_step(usd, ucfg, acp, lat_ref, noise, t, "v_prediction")
    unet = UNetModel(**ucfg); unet.load_state_dict(usd)
    ae = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, **acfg); ae.load_state_dict(asd)
    sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195, prediction_type="v_prediction")
    lat = ae.encode_stage_2_inputs(x, eps=eps, scale_factor=scale)
    assert rel_l2(lat, lat_ref) < 1e-5
    unet.zero_grad()
    loss = ldm_train_step(unet, sched, lat, noise.to(unet.device), t.to(unet.device))
    assert abs(float(loss) - float(loss_ref)) < 1e-4 * float(loss_ref)
    g = unet.grad_dict()
    num = sum(float((g[k].cpu() - grads_ref[k]).double().pow(2).sum()) for k in grads_ref)
    den = sum(float(grads_ref[k].double().pow(2).sum()) for k in grads_ref)
    assert (num / den) ** 0.5 < 1e-4


@pytest.mark.parametrize("spectral", [False, True])
def test_pixel_dm_step_matches_oracle(spectral):
    """Pixel-space diffusion step (training_diffusion.py:141-151, config_dm.yaml): mse(noise_pred, noise) [+ w * JukeboxLoss(sum)]
    on raw windows; L = 256 is one of the FFT lengths the spectral kernel implements."""
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import dm_train_step
    from oracle import losses as Ls, steps as S, unet as U
    ucfg, _B, _L = UNET_CASES["tiny_l64"]
    usd = {k: torch.from_numpy(gen_param(81, k, s)) for k, s in U.unet_param_shapes(ucfg).items()}
    B, L, w = 3, 256, 0.05
    x = torch.from_numpy(eeg_windows(B, seed=83, length=L, pad=8))
    noise = torch.from_numpy(normal((B, 1, L), seed=85)); t = torch.from_numpy(timesteps(B, seed=86))
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0195)
    loss_ref, grads_ref, _ = S.dm_train_step(usd, ucfg, acp, x, noise, t, spectral_weight=w, spectral_loss=spectral)
    unet = UNetModel(**ucfg); unet.load_state_dict(usd)
    sched = DDPMScheduler(1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195)
    unet.zero_grad()
    loss = dm_train_step(unet, sched, x, noise, t.to(unet.device), spectral_weight=w, spectral_loss=spectral)
    assert abs(float(loss) - float(loss_ref)) < 2e-4 * abs(float(loss_ref)), (float(loss), float(loss_ref))
    g = unet.grad_dict()
    num = sum(float((g[k].cpu() - grads_ref[k]).double().pow(2).sum()) for k in grads_ref)
    den = sum(float(grads_ref[k].double().pow(2).sum()) for k in grads_ref)
    assert (num / den) ** 0.5 < 2e-4


@pytest.mark.gpu
def test_grad_scaler_skips_overflow_and_unscales():
    """GradScaler (training.py:334,441-443): a step whose gradients hold an inf is skipped and halves the scale; a clean step
    equals a plain Adam step on the un-scaled gradients; the scale grows after growth_interval clean steps."""
    import eegldm
    from eegldm.models import UNetModel
    from eegldm.training import Adam, GradScaler
    from oracle import steps as S
    net = UNetModel(image_size=64, in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, attention_resolutions=[4],
                    channel_mult=[1, 2], resblock_updown=True, dtype="float32", device=0)
    opt = Adam(net, lr=1e-3)
    sc = GradScaler(init_scale=256.0, growth_interval=2)
    g = torch.Generator().manual_seed(3)
    grad = torch.randn(net.flat.numel(), generator=g).to(net.flat.device)
    p0 = net.flat.clone()
    # overflow: skipped
    net.flat_grad.copy_(grad * sc.get_scale()); net.flat_grad[12345 % net.flat.numel()] = float("inf")
    sc.step(opt); sc.update()
    assert torch.equal(net.flat, p0) and opt.step_count == 0 and sc.get_scale() == 128.0
    net.flat_grad.copy_(grad * sc.get_scale()); net.flat_grad[-1] = float("nan")
    sc.step(opt); sc.update()
    assert torch.equal(net.flat, p0) and sc.get_scale() == 64.0
    # clean steps: Adam on grad (oracle), scale doubles after two of them
    ref = {"p": p0.cpu().clone()}; st = {}
    for i in range(2):
        net.flat_grad.copy_(grad * sc.get_scale())
        sc.step(opt); sc.update()
        ref = S.adam_update(ref, {"p": grad.cpu()}, st, 1e-3, i + 1)
    assert opt.step_count == 2 and sc.get_scale() == 128.0
    torch.testing.assert_close(net.flat.cpu(), ref["p"], rtol=1e-5, atol=1e-7)
