"""DDPMScheduler / DDIMScheduler / DiffusionInferer with the monai-generative interface the
reference scripts use (/root/reference/src/train_ldm.py:199-200, src/training/training.py:420-436,
src/sample_trials.py:136-163, src/train_pure_ldm.py:124,134, src/sample_trials_ddpm.py:83-102).
Schedule tables are tiny host-side constants; the per-element arithmetic runs in
libeegldm (eegldm_add_noise / eegldm_get_velocity / eegldm_ddim_step)."""
import numpy as np
import torch

from ._lib import lib, check, ptr, default_context, PRED

_SCHEDULE_ALIASES = {"linear": "linear_beta", "linear_beta": "linear_beta", "scaled_linear": "scaled_linear_beta",
                     "scaled_linear_beta": "scaled_linear_beta"}


def _betas(schedule, n, beta_start, beta_end):
    schedule = _SCHEDULE_ALIASES[schedule]
    if schedule == "linear_beta":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2


class _Scheduler:
    def __init__(self, num_train_timesteps=1000, schedule="linear_beta", beta_start=1e-4, beta_end=2e-2,
                 prediction_type="epsilon", clip_sample=True, beta_schedule=None, device=0, ctx=None):
        if beta_schedule is not None:       # old monai-generative kwarg, still used by train_ldm.py:199
            schedule = beta_schedule
        if prediction_type not in PRED:
            raise ValueError(f"prediction_type must be one of {list(PRED)}")
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        self.clip_sample = clip_sample
        self.betas = _betas(schedule, num_train_timesteps, beta_start, beta_end)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.ctx = ctx or default_context(device)
        self.device = torch.device("cuda", self.ctx.device)
        self._acp_dev = self.alphas_cumprod.to(self.device)
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))
        self.num_inference_steps = num_train_timesteps

    def to(self, *a, **k):
        return self

    def add_noise(self, original_samples, noise, timesteps):
        x = original_samples.to(self.device, torch.float32).contiguous()
        nz = noise.to(self.device, torch.float32).contiguous()
        t = timesteps.to(self.device, torch.int64).contiguous()
        out = torch.empty_like(x)
        check(lib.eegldm_add_noise(self.ctx.h, ptr(x), ptr(nz), ptr(t), ptr(self._acp_dev), ptr(out), x.shape[0], x[0].numel()))
        return out

    def get_velocity(self, sample, noise, timesteps):
        x = sample.to(self.device, torch.float32).contiguous()
        nz = noise.to(self.device, torch.float32).contiguous()
        t = timesteps.to(self.device, torch.int64).contiguous()
        out = torch.empty_like(x)
        check(lib.eegldm_get_velocity(self.ctx.h, ptr(x), ptr(nz), ptr(t), ptr(self._acp_dev), ptr(out), x.shape[0], x[0].numel()))
        return out


class DDPMScheduler(_Scheduler):
    """Training-side scheduler (add_noise / get_velocity) and the ancestral `step` of the reference's 1000-step samplers
    (util.py:241-243,261-285; sample_trials_ddpm.py:99-102), variance_type "fixed_small".  The step arithmetic is pinned against
    the reference's own DDPM.p_sample (/root/reference/src/models/ldm.py:311-357; tests/golden/ddpm_steps.npz)."""

    def __init__(self, *a, variance_type="fixed_small", **k):
        if variance_type not in ("fixed_small", "fixed_large"):      # "learned" / "learned_range" need a model with 2 x out_channels outputs
            raise NotImplementedError("variance_type 'fixed_small' (the reference's, DDPMScheduler default) or 'fixed_large'")
        k.setdefault("clip_sample", True)
        super().__init__(*a, **k)
        self.variance_type = variance_type
        self._step_calls = 0

    def set_timesteps(self, num_inference_steps):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64))

    def step(self, model_output, timestep, sample, generator=None, noise=None):
        """-> (pred_prev_sample, pred_original_sample).  `noise` (optional, the parity tests pass it) replaces the N(0,1) draw;
        otherwise the draw comes from `generator` (a torch CPU/GPU generator) or the device Philox stream."""
        t = int(timestep)
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[t - 1]) if t > 0 else 1.0
        mo = model_output.to(self.device, torch.float32).contiguous()
        x = sample.to(self.device, torch.float32).contiguous()
        nz = None
        if t > 0:
            if noise is not None:
                nz = noise.to(self.device, torch.float32).contiguous()
            elif generator is not None:
                nz = torch.randn(x.shape, generator=generator, device=generator.device).to(self.device)
            else:
                nz = torch.empty_like(x)
                check(lib.eegldm_randn(self.ctx.h, ptr(nz), nz.numel(), 0x5EED + t, self._step_calls * ((nz.numel() + 3) // 4)))
                self._step_calls += 1
        prev, x0 = torch.empty_like(x), torch.empty_like(x)
        check(lib.eegldm_ddpm_step_var(self.ctx.h, ptr(mo), ptr(x), ptr(nz), a_t, a_prev, float(self.betas[t]), int(self.variance_type == "fixed_large"),
                                       PRED[self.prediction_type], int(self.clip_sample), ptr(prev), ptr(x0), x.numel()))
        return prev, x0


class DDIMScheduler(_Scheduler):
    def __init__(self, *a, set_alpha_to_one=True, steps_offset=0, **k):
        super().__init__(*a, **k)
        self.final_alpha_cumprod = 1.0 if set_alpha_to_one else float(self.alphas_cumprod[0])
        self.steps_offset = steps_offset

    def set_timesteps(self, num_inference_steps):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("num_inference_steps cannot exceed num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = torch.from_numpy((np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)) + self.steps_offset

    def step(self, model_output, timestep, sample, eta=0.0, generator=None, noise=None):
        """-> (pred_prev_sample, pred_original_sample).  eta = 0: the deterministic step the reference samples with (sample_trials.py:163);
        eta > 0: sigma_t(eta) noise on top (`noise` given by the caller, else drawn from `generator` or the device Philox stream)."""
        t = int(timestep)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev_t]) if prev_t >= 0 else self.final_alpha_cumprod
        mo = model_output.to(self.device, torch.float32).contiguous()
        x = sample.to(self.device, torch.float32).contiguous()
        prev, x0 = torch.empty_like(x), torch.empty_like(x)
        if eta < 0:
            raise ValueError("eta must be >= 0")
        if eta == 0.0:
            check(lib.eegldm_ddim_step(self.ctx.h, ptr(mo), ptr(x), a_t, a_prev, PRED[self.prediction_type], int(self.clip_sample),
                                       ptr(prev), ptr(x0), x.numel()))
            return prev, x0
        if noise is not None:
            nz = noise.to(self.device, torch.float32).contiguous()
        elif generator is not None:
            nz = torch.randn(x.shape, generator=generator, device=generator.device).to(self.device)
        else:
            nz = torch.empty_like(x)
            self._step_calls = getattr(self, "_step_calls", 0)
            check(lib.eegldm_randn(self.ctx.h, ptr(nz), nz.numel(), 0xD1D1 + t, self._step_calls * ((nz.numel() + 3) // 4)))
            self._step_calls += 1
        check(lib.eegldm_ddim_step_eta(self.ctx.h, ptr(mo), ptr(x), ptr(nz), a_t, a_prev, float(eta), PRED[self.prediction_type],
                                       int(self.clip_sample), ptr(prev), ptr(x0), x.numel()))
        return prev, x0


class DiffusionInferer:
    """training_diffusion.py:146 / sample_trials_ddpm.py:99-102."""

    def __init__(self, scheduler):
        self.scheduler = scheduler

    def __call__(self, inputs, diffusion_model, noise, timesteps, condition=None):
        noisy = self.scheduler.add_noise(original_samples=inputs, noise=noise, timesteps=timesteps)
        return diffusion_model(x=noisy, timesteps=timesteps)

    @torch.no_grad()
    def sample(self, input_noise, diffusion_model, scheduler=None, save_intermediates=False, intermediate_steps=100,
               conditioning=None, verbose=False):
        """The loop of DiffusionInferer.sample (sample_trials_ddpm.py:99-102, util.py:261-285): model call + scheduler.step per
        entry of scheduler.timesteps; with save_intermediates returns (image, [every intermediate_steps-th image]) like MONAI."""
        scheduler = scheduler or self.scheduler
        image = input_noise
        intermediates = []
        for t in scheduler.timesteps:
            tt = torch.full((image.shape[0],), int(t), dtype=torch.int64)
            out = diffusion_model(image, timesteps=tt)
            image, _ = scheduler.step(out, int(t), image)
            if save_intermediates and int(t) % intermediate_steps == 0:
                intermediates.append(image)
        return (image, intermediates) if save_intermediates else image
