"""Loss callables with the interfaces the reference trainer uses
(/root/reference/src/train_autoencoderkl.py:155-158): L1Loss, PatchAdversarialLoss("least_squares"),
JukeboxLoss(spatial_dims=1, reduction="sum").  Forward values only (device scalars); gradients
are produced by the fused native train step (eegldm.training.aekl_train_step)."""
import torch

from ._lib import lib, check, ptr, default_context


class _Loss:
    def __init__(self, device=0, ctx=None):
        self.ctx = ctx or default_context(device)
        self.device = torch.device("cuda", self.ctx.device)

    def _prep(self, t):
        return t.to(self.device, torch.float32).contiguous()


class L1Loss(_Loss):
    def __call__(self, inp, target):
        a, b = self._prep(inp), self._prep(target)
        out = torch.zeros((), device=self.device)
        check(lib.eegldm_l1_loss(self.ctx.h, ptr(a), ptr(b), ptr(out), None, a.numel(), 0.0))
        return out


class JukeboxLoss(_Loss):
    def __init__(self, spatial_dims=1, fft_signal_size=None, fft_norm="ortho", reduction="mean", **k):
        super().__init__(**k)
        if spatial_dims != 1 or fft_signal_size is not None or fft_norm != "ortho":
            raise NotImplementedError("reference uses JukeboxLoss(spatial_dims=1, reduction='sum')")
        self.reduction = reduction

    def __call__(self, inp, target):
        a, b = self._prep(inp), self._prep(target)
        B, Cc, L = a.shape
        out = torch.zeros((), device=self.device)
        check(lib.eegldm_spectral_loss(self.ctx.h, ptr(a), ptr(b), ptr(out), None, B, Cc, L, 0.0))
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
        out = torch.zeros((), device=self.device)
        check(lib.eegldm_lsgan_loss(self.ctx.h, ptr(a), 1 if target_is_real else 0, ptr(out), None, a.numel(), 0.0))
        return out
