"""torch.autograd bridge over the native forward / backward entry points, so that the reference's OWN loop bodies run after the import swap:

    /root/reference/src/training/training.py:419-443        noise_pred = model(x=noisy_e, timesteps=timesteps); loss = F.mse_loss(...)
                                                            scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()
    /root/reference/src/train_autoencoderkl.py:203-234      reconstruction, z_mu, z_sigma = model(images); logits_fake = discriminator(...)[-1]
                                                            loss_g.backward(); optimizer_g.step(); ...; loss_d.backward(); optimizer_d.step()
    /root/reference/src/train_ldm.py:208                    torch.optim.Adam(diffusion.parameters(), lr=...)

How it maps onto the engine (which keeps ONE flat fp32 parameter buffer and ONE flat gradient buffer per model, and a hand-written backward):

* `model.parameters()` yields ONE torch.nn.Parameter that aliases the flat buffer; its `.grad` IS the flat gradient buffer the native backward
  accumulates into.  torch.optim.Adam / GradScaler work on it element-wise exactly as they would on the 278 separate tensors.
  `optimizer.zero_grad(set_to_none=True)` is honoured: the next backward zeroes the buffer and re-attaches it.
* after an in-place update by a torch optimizer the bf16 / K-blocked weight copies are stale; the tensor version counter tells, and the next
  forward refreshes them (`FlatModule._sync_if_stale`).
* every model call under grad mode goes through a torch.autograd.Function whose backward calls the native backward (`eegldm_unet_backward`,
  `eegldm_aekl_backward_ex`, `eegldm_disc_backward`).  A native executor holds the tape of its LATEST forward only; when a backward arrives for
  an older forward (the discriminator is called on the fake and on the real batch before `loss_d.backward()`), the bridge re-runs that forward
  first -- for the discriminator with BatchNorm in "batch statistics, running statistics untouched" mode (eegldm_disc_forward training = 2), so
  the running statistics see exactly the three updates per step the reference's three training-mode forwards make.
* L1Loss / JukeboxLoss / PatchAdversarialLoss / mse_loss are Functions over the native loss kernels (value and gradient from one launch).

The fused entry points (eegldm.training.ldm_train_step / aekl_train_step) remain the fast path: one native call per step, no Python between
the phases.  tests/test_gpu_autograd.py replays the oracle's training trajectories through BOTH and compares them.
"""
import torch

from ._lib import lib, check, ptr


class FlatModule:
    """Mixin of the model shims: torch-parameter view of the flat buffers + staleness tracking of the compute-dtype weight copies."""

    def _flat_param(self):
        p = getattr(self, "_param", None)
        if p is None or p.data_ptr() != self.flat.data_ptr():
            p = torch.nn.Parameter(self.flat, requires_grad=True)     # aliases self.flat (same storage)
            p.grad = self.flat_grad
            self._param = p
        return p

    def parameters(self):
        return [self._flat_param()]

    def named_parameters(self):
        return [("flat", self._flat_param())]

    def requires_grad_(self, flag=True):
        self._flat_param().requires_grad_(flag)
        return self

    def _versions(self):
        p = getattr(self, "_param", None)
        return (self.flat._version, p._version if p is not None else -1)

    def _mark_synced(self):
        self._synced = self._versions()

    def _sync_if_stale(self):
        if getattr(self, "_synced", None) != self._versions():
            self.sync_weights()
            self._mark_synced()

    def _adopt_grad(self):
        """Called at the top of a bridge backward: make `.grad` of the flat parameter the native gradient buffer."""
        p = self._flat_param()
        if p.grad is None:                       # optimizer.zero_grad(set_to_none=True)
            self.flat_grad.zero_()
            p.grad = self.flat_grad
        elif p.grad.data_ptr() != self.flat_grad.data_ptr():
            raise RuntimeError("the flat parameter's .grad was replaced by another tensor; the native backward accumulates into model.flat_grad")

    def _bump_tape(self):
        """Every native entry that rewrites or consumes the executor's single tape calls this (forward in any mode, encode / decode,
        the fused train steps, the sampler, backward): a bridge backward whose forward is no longer the latest one then sees a different
        id and re-runs its forward first -- also when the intervening call never went through a torch.autograd.Function (a validation
        forward under no_grad, an eval-mode call, a fused step)."""
        self._tape_id = getattr(self, "_tape_id", 0) + 1
        return self._tape_id

    def _wants_graph(self, *tensors):
        if not torch.is_grad_enabled():
            return False
        p = getattr(self, "_param", None)
        return (p is not None and p.requires_grad) or any(torch.is_tensor(t) and t.requires_grad for t in tensors)


# ---------------------------------------------------------------------------------------------------------------- UNet
class _UNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, t, _p):
        out = net._forward_native(x, t)                 # (bumps net._tape_id)
        ctx.net, ctx.tape = net, net._tape_id
        ctx.save_for_backward(x, t)
        return out

    @staticmethod
    def backward(ctx, dy):
        net = ctx.net
        x, t = ctx.saved_tensors
        if net._tape_id != ctx.tape:             # another native call rewrote the tape since: rebuild this call's tape
            net._forward_native(x, t)
        net._adopt_grad()
        dx = net.backward(dy.contiguous(), need_dx=ctx.needs_input_grad[1])      # (bumps: the tape is consumed)
        return None, dx, None, None


# ---------------------------------------------------------------------------------------------------------------- AutoencoderKL
class _AeklFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, eps, _p):
        recon, mu, sg = net._forward_native(x, eps)
        ctx.net, ctx.tape = net, net._tape_id
        ctx.save_for_backward(x, eps)
        return recon, mu, sg

    @staticmethod
    def backward(ctx, d_recon, d_mu, d_sigma):
        net = ctx.net
        x, eps = ctx.saved_tensors
        if net._tape_id != ctx.tape:
            net._forward_native(x, eps)
        net._adopt_grad()
        dev = net.device
        f = lambda g: None if g is None else g.to(dev, torch.float32).contiguous()
        d_recon = f(d_recon) if d_recon is not None else torch.zeros(x.shape[0], net.out_channels, x.shape[2], device=dev)
        d_mu, d_sigma = f(d_mu), f(d_sigma)
        dx = torch.empty_like(x) if ctx.needs_input_grad[1] else None
        check(lib.eegldm_aekl_backward_ex(net.h, ptr(d_recon), ptr(d_mu), ptr(d_sigma), 0.0, ptr(dx)))
        net._bump_tape()
        return None, dx, None, None


# ---------------------------------------------------------------------------------------------------------------- PatchDiscriminator
class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, x, _p):
        logits = net._forward_native(x, 1 if net.training else 0)
        ctx.net, ctx.tape, ctx.was_training = net, net._tape_id, net.training
        ctx.save_for_backward(x)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        net = ctx.net
        (x,) = ctx.saved_tensors
        if net._tape_id != ctx.tape:             # e.g. D(fake) then D(real) before loss_d.backward(): re-forward, running statistics untouched
            net._forward_native(x, 2 if ctx.was_training else 0)
        net._adopt_grad()
        p = net._flat_param()
        dx = net.backward(dlogits.contiguous(), need_dx=ctx.needs_input_grad[1], param_grads=p.requires_grad, in_shape=tuple(x.shape))
        return None, dx, None


# ---------------------------------------------------------------------------------------------------------------- losses
class _GradLoss(torch.autograd.Function):
    """loss value + d loss / d input from ONE native launch; backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, fn, inp):
        need = ctx.needs_input_grad[1]
        g = torch.zeros_like(inp) if need else None
        out = fn(inp, g)
        if need:
            ctx.save_for_backward(g)
        ctx.has_grad = need
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.has_grad:
            return None, None
        (g,) = ctx.saved_tensors
        return None, g * dout


def grad_loss(fn, inp):
    """fn(input, grad_buffer_or_None) -> 0-d loss tensor; differentiable w.r.t. `inp` when it requires grad."""
    if torch.is_tensor(inp) and inp.requires_grad and torch.is_grad_enabled():
        return _GradLoss.apply(fn, inp)
    return fn(inp, None)
