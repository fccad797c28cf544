"""DDIM sampling of 30-s EEG windows: the loop of /root/reference/src/sample_trials.py:149-170
(noise -> UNet/DDIM steps -> decode(z / scale_factor) -> crop [36:-36]), batched over seeds instead of
one window at a time (the reference runs batch 1, which is launch-bound), with the step count a real
parameter (the reference hard-codes 200, sample_trials.py:144)."""
import ctypes as C
import os

import torch

from ._lib import lib, check, ptr, PRED
from .schedulers import DDIMScheduler
from .training import randn


def _step_tables(scheduler):
    """Host-side schedule arrays for eegldm_sample: (timesteps, a_t, a_prev, beta_t, ancestral)."""
    from .schedulers import DDPMScheduler
    ts = [int(t) for t in scheduler.timesteps]
    acp = scheduler.alphas_cumprod
    ancestral = isinstance(scheduler, DDPMScheduler)
    if ancestral:
        prev = [t - 1 for t in ts]
        final = 1.0
    else:
        ratio = scheduler.num_train_timesteps // scheduler.num_inference_steps
        prev = [t - ratio for t in ts]
        final = scheduler.final_alpha_cumprod
    a_t = [float(acp[t]) for t in ts]
    a_prev = [float(acp[p]) if p >= 0 else float(final) for p in prev]
    beta = [float(scheduler.betas[t]) for t in ts]
    return ts, a_t, a_prev, beta, ancestral


@torch.no_grad()
def ddim_sample(unet, autoencoder, scheduler, noise, scale_factor=1.0, crop=36, use_graph=None, seed=0, info=None):
    """noise (B, lat, Ll) on the device -> (windows (B, out, 3072 - 2*crop), final latents).  ONE native call
    (eegldm_sample): the scheduler loop, z / scale_factor and the decode run inside the library.  The UNet forward CAN be
    replayed from a hipGraph (use_graph=True or EEGLDM_SAMPLE_GRAPH=1) but that is no longer the default: measured on
    MI355X (rounds 2 and 3) the replay is SLOWER than the eager launches at the reference's batch of one window per call
    (sample_trials.py:149-163: 92.6 vs 87 ms per 50-step window -- ROCm's graph launch does not shorten the ~5 us per
    dependent kernel) and indistinguishable at batch 256, where launch overhead does not matter.
    `scheduler` may be a DDIMScheduler (eta 0) or a DDPMScheduler (ancestral steps; noise from the device Philox
    stream `seed`).  info (optional dict) receives {"graph": bool}."""
    unet.eval()
    x = noise.to(unet.device, torch.float32).contiguous()
    B, Cc, L = x.shape
    if Cc != unet.in_channels:
        raise ValueError(f"noise has {Cc} channels, the UNet takes {unet.in_channels}")
    ts, a_t, a_prev, beta, ancestral = _step_tables(scheduler)
    n = len(ts)
    if use_graph is None:
        use_graph = os.environ.get("EEGLDM_SAMPLE_GRAPH", "0") == "1" and os.environ.get("EEGLDM_NO_GRAPH") is None
    down = autoencoder.down if autoencoder is not None else 1
    out_c = autoencoder.out_channels if autoencoder is not None else Cc
    lat = torch.empty_like(x)
    win = torch.empty(B, out_c, L * down, device=unet.device, dtype=torch.float32)
    if B == 0:
        return (win[:, :, crop:-crop] if crop else win), lat
    used = C.c_int(0)
    check(lib.eegldm_sample(unet.h, autoencoder.h if autoencoder is not None else None, ptr(x), (C.c_int64 * n)(*ts),
                            (C.c_float * n)(*a_t), (C.c_float * n)(*a_prev), (C.c_float * n)(*beta), n, 1 if ancestral else 0,
                            PRED[scheduler.prediction_type], int(scheduler.clip_sample), 1.0 / float(scale_factor), int(seed),
                            ptr(lat), ptr(win), B, L, 1 if use_graph else 0, C.byref(used)))
    unet._bump_tape()
    if autoencoder is not None:
        autoencoder._bump_tape()
    if info is not None:
        info["graph"] = bool(used.value)
    return (win[:, :, crop:-crop] if crop else win), lat


@torch.no_grad()
def ddim_sample_hostloop(unet, autoencoder, scheduler, noise, scale_factor=1.0, crop=36):
    """The same loop driven from Python, one scheduler.step call per timestep (what round 1 shipped; kept as the
    reference composition the native sampler is tested against, and for schedulers the native loop does not know)."""
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
