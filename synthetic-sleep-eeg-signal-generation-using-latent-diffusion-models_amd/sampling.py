"""DDIM sampling of 30-s EEG windows: the loop of /root/reference/src/sample_trials.py:149-170
(noise -> UNet/DDIM steps -> decode(z / scale_factor) -> crop [36:-36]), batched over seeds instead of
one window at a time (the reference runs batch 1, which is launch-bound), with the step count a real
parameter (the reference hard-codes 200, sample_trials.py:144)."""
import torch

from ._lib import lib, check, ptr
from .schedulers import DDIMScheduler
from .training import randn


@torch.no_grad()
def ddim_sample(unet, autoencoder, scheduler, noise, scale_factor=1.0, crop=36):
    """noise (B, lat, Ll) on the device -> (windows (B, out, 3072 - 2*crop), final latents)."""
    unet.eval()
    x = noise.to(unet.device, torch.float32).contiguous()
    B = x.shape[0]
    tt = torch.empty(B, device=unet.device, dtype=torch.int64)
    for t in scheduler.timesteps:
        tt.fill_(int(t))
        out = unet(x, timesteps=tt)
        x, _ = scheduler.step(out, int(t), x)
    if autoencoder is None:      # pixel-space model (sample_trials_ddpm.py:99-104): the UNet output IS the window
        return (x[:, :, crop:-crop] if crop else x), x
    z = x
    if float(scale_factor) != 1.0:
        z = x.clone()
        check(lib.eegldm_axpy(unet.ctx.h, ptr(z), ptr(z), 1.0 / float(scale_factor) - 1.0, z.numel()))
    sample = autoencoder.decode_stage_2_outputs(z)
    return (sample[:, :, crop:-crop] if crop else sample), x


def make_sampling_scheduler(num_inference_steps=50, prediction_type="epsilon", beta_start=0.0015, beta_end=0.0205, device=0):
    """DDIMScheduler as built at sample_trials.py:136-145 (scaled-linear betas, clip_sample=False).  The reference
    script passes prediction_type="v_prediction" while training with epsilon: it is a parameter here."""
    s = DDIMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=beta_start, beta_end=beta_end,
                      prediction_type=prediction_type, clip_sample=False, device=device)
    s.set_timesteps(num_inference_steps)
    return s


def sample_seeds(unet, autoencoder, scheduler, seeds, latent_len=768, scale_factor=1.0, crop=36):
    """One window per seed (sample_trials.py:149-151 draws a fresh N(0,1) latent per seed), batched.
    autoencoder=None samples a pixel-space model: latent_len is then the window length (3072)."""
    lat = unet.in_channels
    noise = torch.empty(len(seeds), lat, latent_len, device=unet.device)
    for i, sd in enumerate(seeds):
        noise[i] = randn(unet.ctx, (lat, latent_len), seed=int(sd))
    return ddim_sample(unet, autoencoder, scheduler, noise, scale_factor, crop)
