"""Train-step bodies of the reference loops, each one native call per phase:
  ldm_train_step  <-> /root/reference/src/training/training.py:419-443
  Adam            <-> torch.optim.Adam as used at /root/reference/src/train_ldm.py:208
Gradients live in each model's flat fp32 buffer, so data-parallel training is ONE
all-reduce over a contiguous tensor (see eegldm.distributed)."""
import ctypes as C

import torch

from ._lib import lib, check, ptr, PRED


class Adam:
    """torch.optim.Adam defaults (betas 0.9/0.999, eps 1e-8, no weight decay) as one fused HIP
    kernel over the model's flat parameter buffer."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        self.model, self.lr, self.betas, self.eps = model, lr, betas, eps
        self.m = torch.zeros_like(model.flat)
        self.v = torch.zeros_like(model.flat)
        self.step_count = 0
        self.param_groups = [{"lr": lr}]

    def zero_grad(self, set_to_none=True):
        self.model.zero_grad()

    def step(self, grad_inv_scale=1.0):
        self.step_count += 1
        md = self.model
        check(lib.eegldm_adam_step(md.ctx.h, ptr(md.flat), ptr(md.flat_grad), ptr(self.m), ptr(self.v), md.flat.numel(),
                                   self.param_groups[0]["lr"], self.betas[0], self.betas[1], self.eps, self.step_count, grad_inv_scale))
        md.sync_weights()

    # ---- checkpoint wire format: torch.optim.Adam's own state_dict layout, so that the optimizer entry of a reference
    # checkpoint.pth (train_autoencoderkl.py:320-329, training/training.py:381-388 store `optimizer.state_dict()`) resumes here
    # and a checkpoint written here resumes under torch.optim.Adam.  Parameter index i = the i-th entry of the model's
    # parameter table (= the order of nn.Module.parameters() in the reference; BatchNorm buffers are not parameters).
    def state_dict(self):
        return flat_to_torch_adam_state(self.model.entries, self.m, self.v, self.step_count, self.param_groups[0]["lr"], self.betas, self.eps)

    def load_state_dict(self, sd):
        if "state" not in sd:             # round-1 private format {step, exp_avg, exp_avg_sq, lr} (flat tensors)
            self.step_count = int(sd["step"]); self.m.copy_(sd["exp_avg"]); self.v.copy_(sd["exp_avg_sq"])
            self.param_groups[0]["lr"] = sd.get("lr", self.lr)
            return
        step, hyper = torch_adam_state_to_flat(self.model.entries, sd, self.m, self.v)
        self.step_count = step
        self.param_groups[0]["lr"] = hyper.get("lr", self.lr)
        self.betas = tuple(hyper.get("betas", self.betas)); self.eps = hyper.get("eps", self.eps)


def _ref_layout(t, shape):
    """flat packed slice -> tensor in the reference shape (conv weights are stored [K][Cout][Cin])"""
    if len(shape) == 3:
        t = t.reshape(shape[2], shape[0], shape[1]).permute(1, 2, 0)
    return t.reshape(shape).contiguous()


def flat_to_torch_adam_state(entries, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8):
    """{'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]} exactly as torch.optim.Adam.state_dict() lays it out."""
    state = {}
    if step > 0:                       # torch creates per-parameter state lazily at the first step
        for i, (_k, (o, n, shape)) in enumerate(entries.items()):
            state[i] = {"step": torch.tensor(float(step)), "exp_avg": _ref_layout(m[o:o + n], shape).cpu().clone(),
                        "exp_avg_sq": _ref_layout(v[o:o + n], shape).cpu().clone()}
    group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
             "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
             "params": list(range(len(entries)))}
    return {"state": state, "param_groups": [group]}


def torch_adam_state_to_flat(entries, sd, m_out, v_out):
    """Inverse of flat_to_torch_adam_state for a state_dict written by torch.optim.Adam (any torch version: `step` may be an int
    or a tensor) or by this class.  Fills the flat moment buffers; returns (step, hyper-parameters of the single param group)."""
    groups = sd["param_groups"]
    if len(groups) != 1:
        raise ValueError("the reference optimizers have a single param group")
    g = groups[0]
    if g.get("amsgrad") or g.get("weight_decay", 0) not in (0, 0.0) or g.get("maximize"):
        raise NotImplementedError("Adam with amsgrad / weight_decay / maximize is not used by the reference (train_ldm.py:208)")
    if len(g["params"]) != len(entries):
        raise ValueError(f"optimizer state covers {len(g['params'])} parameters, the model has {len(entries)}")
    state = sd["state"]
    m_out.zero_(); v_out.zero_()
    steps = set()
    for pos, (k, (o, n, shape)) in enumerate(entries.items()):
        st = state.get(g["params"][pos], state.get(str(g["params"][pos])))
        if st is None:
            continue                   # parameter that never received a gradient
        for name, dst in (("exp_avg", m_out), ("exp_avg_sq", v_out)):
            t = torch.as_tensor(st[name]).detach().to(torch.float32)
            if tuple(t.shape) != tuple(shape):
                raise ValueError(f"{k}: optimizer state shape {tuple(t.shape)} != parameter shape {tuple(shape)}")
            if len(shape) == 3:
                t = t.permute(2, 0, 1)
            dst[o:o + n].copy_(t.reshape(-1).to(dst.device))
        steps.add(int(float(st["step"])))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}); the fused Adam keeps one step count")
    return (steps.pop() if steps else 0), {k2: g[k2] for k2 in ("lr", "betas", "eps") if k2 in g}


class GradScaler:
    """Dynamic loss scaling with torch.cuda.amp.GradScaler's interface and update rule (the reference's LDM / DM loops:
    /root/reference/src/training/training.py:334,441-443 -- ``scaler.scale(loss).backward(); scaler.step(opt); scaler.update()``).

    The native train steps take the scale as their ``grad_scale`` argument (``get_scale()``), ``step`` checks the flat
    gradient buffer for inf/nan on the device (one host read of the flag, as torch's scaler does), skips the optimizer
    step when one is found and otherwise un-scales inside the Adam kernel; ``update`` backs off / grows the scale.
    bf16 and fp32 share fp32's exponent range, so the entry scripts leave it disabled unless asked (--grad-scaler)."""

    def __init__(self, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000, enabled=True):
        self._scale, self._growth_factor, self._backoff_factor = float(init_scale), float(growth_factor), float(backoff_factor)
        self._growth_interval, self._enabled = int(growth_interval), bool(enabled)
        self._growth_tracker, self._found_inf, self._flag = 0, None, None

    def is_enabled(self):
        return self._enabled

    def get_scale(self):
        return self._scale if self._enabled else 1.0

    def scale(self, outputs):
        return outputs * self.get_scale() if self._enabled else outputs

    def unscale_(self, optimizer):
        """Records whether optimizer.model's gradients hold an inf/nan (the division itself happens inside the Adam kernel)."""
        if not self._enabled:
            return
        md = optimizer.model
        if self._flag is None or self._flag.device != md.flat_grad.device:
            self._flag = torch.zeros(1, device=md.flat_grad.device)
        check(lib.eegldm_grad_check_finite(md.ctx.h, ptr(md.flat_grad), md.flat_grad.numel(), ptr(self._flag)))
        md.ctx.sync()
        self._found_inf = bool(float(self._flag) != 0.0)

    def step(self, optimizer):
        if not self._enabled:
            return optimizer.step()
        if self._found_inf is None:
            self.unscale_(optimizer)
        if self._found_inf:
            return None                     # skipped: parameters, moments and the optimizer's step count stay as they are
        return optimizer.step(grad_inv_scale=1.0 / self._scale)

    def update(self, new_scale=None):
        if not self._enabled:
            return
        if new_scale is not None:
            self._scale, self._found_inf = float(new_scale), None
            return
        found = bool(self._found_inf)
        if found:
            self._scale *= self._backoff_factor
            self._growth_tracker = 0
        else:
            self._growth_tracker += 1
            if self._growth_tracker == self._growth_interval:
                self._scale *= self._growth_factor
                self._growth_tracker = 0
        self._found_inf = None

    def state_dict(self):
        return {"scale": self._scale, "growth_factor": self._growth_factor, "backoff_factor": self._backoff_factor,
                "growth_interval": self._growth_interval, "_growth_tracker": self._growth_tracker} if self._enabled else {}

    def load_state_dict(self, sd):
        if not self._enabled or not sd:
            return
        self._scale, self._growth_factor, self._backoff_factor = float(sd["scale"]), float(sd["growth_factor"]), float(sd["backoff_factor"])
        self._growth_interval, self._growth_tracker = int(sd["growth_interval"]), int(sd["_growth_tracker"])


GRAD_HOOK = C.CFUNCTYPE(None, C.c_void_p, C.c_long, C.c_long)


def set_grad_hook(unet, fn):
    """fn(offset, numel) is called from inside the native backward as soon as unet.flat_grad[offset:offset+numel]
    (out / output_blocks / middle_block) is final in stream order; None removes the hook.  The ctypes thunk is kept
    alive on the model."""
    if fn is None:
        unet._grad_hook_thunk = None
        check(lib.eegldm_unet_set_grad_hook(unet.h, None, None))
        return
    thunk = GRAD_HOOK(lambda _user, off, n: fn(int(off), int(n)))
    unet._grad_hook_thunk = thunk
    check(lib.eegldm_unet_set_grad_hook(unet.h, C.cast(thunk, C.c_void_p), None))


def ldm_train_step(unet, scheduler, latents, noise, timesteps, loss_out=None, grad_scale=1.0, grad_sync=None):
    """add_noise -> UNet forward -> MSE against noise (epsilon) or velocity (v_prediction) -> backward.
    Accumulates into unet.flat_grad; returns the device scalar loss tensor.  grad_sync: an
    eegldm.distributed.OverlappedGradSync -- the tail of the gradient buffer is all-reduced while the input blocks'
    backward still runs, the rest right after the call; the caller then only has to `grad_sync.wait()`."""
    if loss_out is None:
        loss_out = torch.zeros(1, device=unet.device)
    B, _C, L = latents.shape
    if grad_sync is not None:
        grad_sync.begin()
        set_grad_hook(unet, grad_sync.on_ready)
    try:
        check(lib.eegldm_ldm_train_step(unet.h, ptr(latents), ptr(noise), ptr(timesteps), ptr(scheduler._acp_dev),
                                        PRED[scheduler.prediction_type], B, L, grad_scale, ptr(loss_out)))
    finally:
        unet._bump_tape()                    # forward + backward ran inside the call: an older autograd graph's tape is gone
        if grad_sync is not None:
            set_grad_hook(unet, None)
    if grad_sync is not None:
        grad_sync.finish()
    return loss_out


def dm_train_step(unet, scheduler, images, noise, timesteps, spectral_weight=0.0, spectral_loss=False, loss_out=None, grad_sync=None, grad_scale=1.0):
    """Pixel-space diffusion step of /root/reference/src/training/training_diffusion.py:141-151 (config_dm.yaml, BASELINE C5):
    epsilon prediction directly on the (B,1,3072) windows, loss = mse(noise_pred, noise) [+ spectral_weight *
    JukeboxLoss(sum)(noise_pred, noise)].  Composed from the same native calls as the latent step: add_noise, UNet forward,
    MSE (writes d pred), spectral loss (accumulates its gradient into d pred), hand-written backward.  Returns the loss tensor.
    grad_scale: the GradScaler's loss scale (training_diffusion.py:37,149-151 -- `scaler.scale(loss).backward()`): it multiplies d pred, i.e.
    every gradient of the backward; the reported loss stays unscaled and the optimizer step divides the scale out again (GradScaler.step)."""
    dev = unet.device
    if loss_out is None:
        loss_out = torch.zeros(1, device=dev)
    unet.train()
    x = images.to(dev, torch.float32).contiguous(); nz = noise.to(dev, torch.float32).contiguous()
    B, Cc, L = x.shape
    noisy = scheduler.add_noise(original_samples=x, noise=nz, timesteps=timesteps)
    pred = unet(noisy, timesteps=timesteps)
    dpred = torch.empty_like(pred)
    check(lib.eegldm_mse_loss(unet.ctx.h, ptr(pred), ptr(nz), ptr(loss_out), ptr(dpred), pred.numel(), float(grad_scale)))
    if spectral_loss:
        spec = torch.zeros(1, device=dev)
        check(lib.eegldm_spectral_loss(unet.ctx.h, ptr(pred), ptr(nz), ptr(spec), ptr(dpred), B, Cc, L, float(spectral_weight) * float(grad_scale)))
        loss_out.add_(spec, alpha=float(spectral_weight))
    if grad_sync is not None:
        grad_sync.begin()
        set_grad_hook(unet, grad_sync.on_ready)
    try:
        unet.backward(dpred)
    finally:
        if grad_sync is not None:
            set_grad_hook(unet, None)
    if grad_sync is not None:
        grad_sync.finish()
    return loss_out


def randn(ctx, shape, seed, offset=0, device=None):
    out = torch.empty(shape, device=device or torch.device("cuda", ctx.device), dtype=torch.float32)
    check(lib.eegldm_randn(ctx.h, ptr(out), out.numel(), seed, offset))
    return out


def randint(ctx, n, high, seed, offset=0, device=None):
    out = torch.empty(n, device=device or torch.device("cuda", ctx.device), dtype=torch.int64)
    check(lib.eegldm_randint(ctx.h, ptr(out), n, high, seed, offset))
    return out


def aekl_train_step(autoencoder, discriminator, x, eps, adv_weight, kl_weight, spectral_weight, use_spectral,
                    losses_out=None, recon_out=None):
    """The step body of /root/reference/src/train_autoencoderkl.py:203-234 as ONE native call: fills
    autoencoder.flat_grad and discriminator.flat_grad (both must be zeroed first) and returns the device
    tensor [recons L1, spectral, KL, generator adversarial, D fake, D real]."""
    dev = autoencoder.device
    if losses_out is None:
        losses_out = torch.zeros(6, device=dev)
    B, _c, L = x.shape
    check(lib.eegldm_aekl_train_step(autoencoder.h, discriminator.h, ptr(x), ptr(eps), float(adv_weight), float(kl_weight),
                                     float(spectral_weight), 1 if use_spectral else 0, ptr(losses_out), ptr(recon_out), B, L))
    autoencoder._bump_tape(); discriminator._bump_tape()
    return losses_out
