"""ORACLE (test infrastructure only -- never imported by the product path).

The three units of work of the hot path, composed exactly as the reference
training / sampling loops compose them, in fp32 on the CPU with torch autograd
and torch.optim.Adam (what the reference itself calls):

* ldm_train_step   /root/reference/src/training/training.py:419-443
* dm_train_step    /root/reference/src/training/training_diffusion.py:133-158 (pixel-space DM, optional spectral term)
* aekl_train_step  /root/reference/src/train_autoencoderkl.py:200-234
* ddim_sample      /root/reference/src/sample_trials.py:149-170

Random draws (timesteps, noise, eps) are INPUTS so that device and oracle see
identical values.
"""
import torch
import torch.nn.functional as F

from . import aekl as A
from . import losses as Ls
from . import unet as U


def _leafify(sd):
    """Trainable leaves for parameters; BatchNorm running statistics stay plain buffers."""
    return {k: (v.detach().clone().requires_grad_(True) if (v.is_floating_point() and "running" not in k) else v.detach().clone())
            for k, v in sd.items()}


def ldm_train_step(unet_sd, unet_cfg, acp, latents, noise, t, prediction_type="epsilon"):
    """One forward/backward of the denoiser on given latents (already encoded
    and scaled).  Returns (loss, grads dict, pred)."""
    sd = _leafify(unet_sd)
    noisy = Ls.add_noise(acp, latents, noise, t)
    pred = U.unet_forward(sd, unet_cfg, noisy, t)
    target = noise if prediction_type == "epsilon" else Ls.get_velocity(acp, latents, noise, t)
    loss = F.mse_loss(pred.float(), target.float())
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
    return loss.detach(), grads, pred.detach()


def dm_train_step(unet_sd, unet_cfg, acp, images, noise, t, spectral_weight=0.0, spectral_loss=False):
    """Pixel-space diffusion step (training_diffusion.py:141-151): epsilon prediction on the raw windows,
    loss = mse(noise_pred, noise) [+ spectral_weight * JukeboxLoss(sum)(noise_pred, noise)]."""
    sd = _leafify(unet_sd)
    noisy = Ls.add_noise(acp, images, noise, t)
    pred = U.unet_forward(sd, unet_cfg, noisy, t)
    loss = F.mse_loss(pred.float(), noise.float())
    if spectral_loss:
        loss = loss + spectral_weight * Ls.jukebox_loss(pred.float(), noise.float(), reduction="sum")
    loss.backward()
    grads = {k: v.grad for k, v in sd.items() if v.is_floating_point() and v.grad is not None}
    return loss.detach(), grads, pred.detach()


def grad_scaler_update(scale, growth_tracker, found_inf, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
    """torch.cuda.amp.GradScaler.update (used at /root/reference/src/training/training.py:334,443): a step with inf/nan gradients
    multiplies the scale by backoff_factor and resets the streak; growth_interval consecutive clean steps multiply it by
    growth_factor.  Returns (new_scale, new_growth_tracker).  Pinned against torch.amp.GradScaler in tests/test_oracle_closed_form.py."""
    if found_inf:
        return scale * backoff_factor, 0
    t = growth_tracker + 1
    if t == growth_interval:
        return scale * growth_factor, 0
    return scale, t


def adam_update(params, grads, state, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam semantics (no weight decay, no amsgrad), in place."""
    for k, g in grads.items():
        m, v = state.setdefault(k, (torch.zeros_like(g), torch.zeros_like(g)))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        params[k] = params[k] - (lr / bc1) * (m / denom)
    return params


def aekl_train_step(ae_sd, ae_cfg, d_sd, d_cfg, x, eps, adv_weight, kl_weight, spectral_weight, use_spectral,
                    lr_g, lr_d, step, opt_state_g, opt_state_d):
    """Generator update then discriminator update on the same batch.
    Returns dict of losses, updated (ae_sd, d_sd) and the reconstruction."""
    g = _leafify(ae_sd)
    d = _leafify(d_sd)
    running = {}
    recon, z_mu, z_sigma = A.forward(g, ae_cfg, x, eps)
    rec_loss = F.l1_loss(recon.float(), x.float())
    spec = Ls.jukebox_loss(recon.float(), x.float(), "sum")
    kl = Ls.kl_loss(z_mu, z_sigma)
    logits_fake = A.disc_forward(d, d_cfg, recon.contiguous().float(), True, running)[-1]
    for k, v in running.items():      # BN running stats update #1
        d[k] = v
    gen_loss = Ls.patch_adv_loss(logits_fake, True, False)
    loss_g = rec_loss + kl_weight * kl + adv_weight * gen_loss
    if use_spectral:
        loss_g = loss_g + spec * spectral_weight
    loss_g.backward()
    g_grads = {k: v.grad for k, v in g.items() if v.grad is not None}
    new_g = adam_update({k: v.detach() for k, v in g.items()}, g_grads, opt_state_g, lr_g, step)

    d2 = _leafify({k: v.detach() for k, v in d.items()})
    running = {}
    lf = A.disc_forward(d2, d_cfg, recon.detach().contiguous(), True, running)[-1]
    for k, v in running.items():      # update #2
        d2[k] = v
    loss_d_fake = Ls.patch_adv_loss(lf, False, True)
    running = {}
    lr_ = A.disc_forward(d2, d_cfg, x.contiguous(), True, running)[-1]
    loss_d_real = Ls.patch_adv_loss(lr_, True, True)
    disc_loss = (loss_d_fake + loss_d_real) * 0.5
    (adv_weight * disc_loss).backward()
    d_grads = {k: v.grad for k, v in d2.items() if v.grad is not None}
    new_d = adam_update({k: v.detach() for k, v in d2.items()}, d_grads, opt_state_d, lr_d, step)
    for k, v in running.items():      # update #3
        new_d[k] = v
    losses = dict(recons=rec_loss.detach(), spectral=spec.detach(), kl=kl.detach(), gen=gen_loss.detach(),
                  disc=disc_loss.detach())
    return losses, new_g, new_d, recon.detach(), g_grads, d_grads


@torch.no_grad()
def ddim_sample(unet_sd, unet_cfg, ae_sd, ae_cfg, noise, num_inference_steps, acp, scale_factor=1.0,
                prediction_type="epsilon", clip_sample=False, num_train_timesteps=1000, crop=36):
    x = noise
    for t in Ls.ddim_timesteps(num_train_timesteps, num_inference_steps):
        tt = torch.full((x.shape[0],), int(t), dtype=torch.int64)
        out = U.unet_forward(unet_sd, unet_cfg, x, tt)
        x, _ = Ls.ddim_step(acp, out, int(t), x, num_train_timesteps, num_inference_steps, prediction_type, clip_sample)
    sample = A.decode(ae_sd, ae_cfg, x / scale_factor)
    return sample[:, :, crop:-crop] if crop else sample, x
