"""Loss callables with the interfaces the reference trainer uses
(/root/reference/src/train_autoencoderkl.py:155-158): L1Loss, PatchAdversarialLoss("least_squares"),
JukeboxLoss(spatial_dims=1, reduction="sum"), plus mse_loss.  Each call is one native launch that produces the value and -- when the
input carries a graph (eegldm.autograd) -- its gradient; the fused native train step (eegldm.training.aekl_train_step) remains the
fast path."""
import torch

from ._lib import lib, check, ptr, default_context
from .autograd import grad_loss


class _Loss:
    def __init__(self, device=0, ctx=None):
        self.ctx = ctx or default_context(device)
        self.device = torch.device("cuda", self.ctx.device)

    def _prep(self, t):
        return t.to(self.device, torch.float32).contiguous()


class L1Loss(_Loss):
    def __call__(self, inp, target):
        a, b = self._prep(inp), self._prep(target).detach()

        def run(x, g):
            out = torch.zeros((), device=self.device)
            check(lib.eegldm_l1_loss(self.ctx.h, ptr(x), ptr(b), ptr(out), ptr(g), x.numel(), 1.0 if g is not None else 0.0))
            return out
        return grad_loss(run, a)


class JukeboxLoss(_Loss):
    def __init__(self, spatial_dims=1, fft_signal_size=None, fft_norm="ortho", reduction="mean", **k):
        super().__init__(**k)
        if spatial_dims != 1 or fft_signal_size is not None or fft_norm != "ortho":
            raise NotImplementedError("reference uses JukeboxLoss(spatial_dims=1, reduction='sum')")
        self.reduction = reduction

    def __call__(self, inp, target):
        a, b = self._prep(inp), self._prep(target).detach()
        B, Cc, L = a.shape

        def run(x, g):
            out = torch.zeros((), device=self.device)
            check(lib.eegldm_spectral_loss(self.ctx.h, ptr(x), ptr(b), ptr(out), ptr(g), B, Cc, L, 1.0 if g is not None else 0.0))
            return out
        out = grad_loss(run, a)
        return out if self.reduction == "sum" else out / a.numel()


class PatchAdversarialLoss(_Loss):
    def __init__(self, reduction="mean", criterion="least_squares", no_activation_leastsq=False, **k):
        super().__init__(**k)
        if criterion != "least_squares" or no_activation_leastsq or reduction != "mean":
            raise NotImplementedError("reference uses PatchAdversarialLoss(criterion='least_squares')")

    def __call__(self, logits, target_is_real, for_discriminator):
        if isinstance(logits, (list, tuple)):
            logits = logits[-1]
        if not for_discriminator:
            target_is_real = True
        a = self._prep(logits)

        def run(x, g):
            out = torch.zeros((), device=self.device)
            check(lib.eegldm_lsgan_loss(self.ctx.h, ptr(x), 1 if target_is_real else 0, ptr(out), ptr(g), x.numel(), 1.0 if g is not None else 0.0))
            return out
        return grad_loss(run, a)


def mse_loss(inp, target, ctx=None):
    """F.mse_loss(input, target) (training.py:437) on the native kernel: value and gradient from one launch."""
    c = ctx or default_context(inp.device.index or 0)
    a = inp.to(torch.float32).contiguous(); b = target.to(a.device, torch.float32).contiguous().detach()

    def run(x, g):
        out = torch.zeros((), device=a.device)
        check(lib.eegldm_mse_loss(c.h, ptr(x), ptr(b), ptr(out), ptr(g), x.numel(), 1.0))
        return out
    return grad_loss(run, a)
