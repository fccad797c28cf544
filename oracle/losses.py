"""ORACLE (test infrastructure only -- never imported by the product path).

Losses and schedulers on the hot path.

* KL / L1 / MSE: reference call sites /root/reference/src/train_autoencoderkl.py:206-211,
  /root/reference/src/training/training.py:437 -- pinned by closed forms.
* JukeboxLoss, PatchAdversarialLoss, DDPMScheduler, DDIMScheduler live in
  monai-generative (un-pinned, /root/reference/requirements.txt:12; source absent):
  PARITY UNPINNED at that boundary.  The schedule values themselves ARE pinned:
  MONAI `scaled_linear_beta` == the reference's local `make_beta_schedule("linear")`
  (/root/reference/src/models/ldm.py:40-49), checked against
  tests/golden/schedules.npz; add_noise == DDPM.q_sample (ldm.py:392-408).
"""
import numpy as np
import torch
import torch.nn.functional as F


def kl_loss(z_mu, z_sigma):
    """train_autoencoderkl.py:210-211: sum over dim 1, then sum over all / batch."""
    kl = 0.5 * torch.sum(z_mu.pow(2) + z_sigma.pow(2) - torch.log(z_sigma.pow(2)) - 1, dim=[1])
    return torch.sum(kl) / kl.shape[0]


def jukebox_loss(inp, target, reduction="sum"):
    """JukeboxLoss(spatial_dims=1): full complex FFT over dims (1, 2), ortho norm,
    amplitude sqrt(re^2 + im^2), squared difference."""
    def amp(x):
        f = torch.fft.fftn(x, dim=(1, 2), norm="ortho")
        return torch.sqrt(torch.real(f) ** 2 + torch.imag(f) ** 2)
    loss = F.mse_loss(amp(inp), amp(target), reduction="none")
    return loss.sum() if reduction == "sum" else loss.mean()


def patch_adv_loss(logits, target_is_real, for_discriminator):
    """PatchAdversarialLoss(criterion="least_squares"): LeakyReLU(0.05) on the
    logits, then MSE against constant 1 (real) / 0 (fake); the generator always
    targets real."""
    if not for_discriminator:
        target_is_real = True
    a = F.leaky_relu(logits, 0.05)
    tgt = torch.full_like(a, 1.0 if target_is_real else 0.0)
    return F.mse_loss(a, tgt)


# ------------------------------------------------------------------ schedulers
def make_betas(schedule, num_train_timesteps=1000, beta_start=1e-4, beta_end=2e-2):
    if schedule in ("linear_beta", "linear"):
        return torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
    if schedule in ("scaled_linear_beta", "scaled_linear"):
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
    raise ValueError(schedule)


def alphas_cumprod(schedule, num_train_timesteps=1000, beta_start=1e-4, beta_end=2e-2):
    return torch.cumprod(1.0 - make_betas(schedule, num_train_timesteps, beta_start, beta_end), dim=0)


def add_noise(acp, x, noise, t):
    sa = acp[t] ** 0.5
    sb = (1 - acp[t]) ** 0.5
    shape = (-1,) + (1,) * (x.dim() - 1)
    return sa.reshape(shape) * x + sb.reshape(shape) * noise


def get_velocity(acp, x, noise, t):
    sa = acp[t] ** 0.5
    sb = (1 - acp[t]) ** 0.5
    shape = (-1,) + (1,) * (x.dim() - 1)
    return sa.reshape(shape) * noise - sb.reshape(shape) * x


def ddim_timesteps(num_train_timesteps, num_inference_steps):
    step_ratio = num_train_timesteps // num_inference_steps
    return (np.arange(0, num_inference_steps) * step_ratio).round()[::-1].copy().astype(np.int64)


def ddim_step(acp, model_output, t, sample, num_train_timesteps, num_inference_steps,
              prediction_type="epsilon", clip_sample=True, eta=0.0, noise=None):
    """DDIM step (Song et al. 2021, eq. 12 with sigma_t(eta) of eq. 16; the reference samples with eta = 0); returns (prev_sample, pred_original)."""
    prev_t = t - num_train_timesteps // num_inference_steps
    a_t = acp[t]
    a_prev = acp[prev_t] if prev_t >= 0 else torch.tensor(1.0)
    b_t = 1 - a_t
    if prediction_type == "epsilon":
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        e = model_output
    elif prediction_type == "sample":
        x0 = model_output
        e = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
    elif prediction_type == "v_prediction":
        x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
        e = a_t ** 0.5 * model_output + b_t ** 0.5 * sample
    else:
        raise ValueError(prediction_type)
    if clip_sample:
        x0 = torch.clamp(x0, -1, 1)
    if eta == 0.0:
        return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * e, x0
    var = (1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev)
    sigma = eta * var ** 0.5
    prev = a_prev ** 0.5 * x0 + (1 - a_prev - sigma ** 2) ** 0.5 * e + sigma * noise
    return prev, x0


def ddpm_step(acp, betas, model_output, t, sample, noise, prediction_type="epsilon", clip_sample=True, variance_type="fixed_small"):
    """DDPM ancestral step (variance_type fixed_small), used only by the
    reference's logging sampler (/root/reference/src/util.py:241-243)."""
    a_t = acp[t]
    a_prev = acp[t - 1] if t > 0 else torch.tensor(1.0)
    b_t, b_prev = 1 - a_t, 1 - a_prev
    if prediction_type == "epsilon":
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
    elif prediction_type == "sample":
        x0 = model_output
    else:
        x0 = a_t ** 0.5 * sample - b_t ** 0.5 * model_output
    if clip_sample:
        x0 = torch.clamp(x0, -1, 1)
    alpha_t = 1 - betas[t]
    c0 = (a_prev ** 0.5 * betas[t]) / b_t
    ct = alpha_t ** 0.5 * b_prev / b_t
    mean = c0 * x0 + ct * sample
    if t > 0:
        var = betas[t] if variance_type == "fixed_large" else torch.clamp(b_prev / b_t * betas[t], min=1e-20)
        mean = mean + var ** 0.5 * noise
    return mean, x0
