"""-m gpu: the torch.autograd bridge (eegldm/autograd.py) -- the reference's OWN loop bodies, written with torch.optim.Adam, GradScaler,
F.mse_loss and tensor-op KL exactly as /root/reference/src/training/training.py:419-443 and /root/reference/src/train_autoencoderkl.py:203-234
write them, replay the oracle's training trajectories (tests/golden/ldm_traj_c2.json, aekl_traj_thin.json, aekl_traj_c1.json) on the engine,
next to the fused native steps (eegldm.training.ldm_train_step / aekl_train_step) on the same seeds.

"Bit for bit" is not the claim: the parameter gradients of both paths come out of the same kernels, whose split-K / slot sums use fp32
atomics (order-dependent at ~2e-7), and torch.optim.Adam rounds its bias corrections differently from the native Adam kernel.  The bounds:
the bridge follows the ORACLE within the same tolerance as the fused path, and the two engine paths agree with each other at 1e-5 over the
first steps (before the GAN amplifies anything)."""
import json
import os
import sys

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))

from param_gen import gen_param, eeg_windows, normal, timesteps  # noqa: E402

UCFG = dict(in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)


def _golden(name):
    with open(os.path.join(HERE, "golden", name)) as fh:
        return json.load(fh)


@pytest.mark.parametrize("dtype,tol", [("float32", 2e-3), ("bfloat16", 6e-2)])
def test_reference_ldm_loop_body_runs_on_the_engine_and_follows_the_oracle(dtype, tol):
    """training.py:419-443 verbatim (minus the frozen encoder: the fixture starts from latents): torch.optim.Adam over model.parameters()
    (train_ldm.py:208), GradScaler (training.py:334), model(x=..., timesteps=...), F.mse_loss, scaler.scale(loss).backward(), scaler.step, scaler.update."""
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, ldm_train_step
    g = _golden("ldm_traj_c2.json")
    sd = None
    nets = []
    for _ in range(2):
        net = UNetModel(image_size=768, **UCFG, dtype=dtype)
        if sd is None:
            sd = {k: torch.from_numpy(gen_param(g["param_seed"], k, tuple(v.shape))) for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        nets.append(net)
    model, fused = nets
    scheduler = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
    optimizer = torch.optim.Adam(model.parameters(), lr=g["lr"])
    scaler = torch.cuda.amp.GradScaler()
    opt_f = Adam(fused, lr=g["lr"])
    B, POOL = g["batch"], g["pool"]
    pool = torch.from_numpy(eeg_windows(POOL, seed=g["latent_seed"], length=768)).cuda()
    lf = torch.zeros(1, device="cuda")
    worst = worst_f = 0.0
    nsteps = 12
    for i in range(1, nsteps + 1):
        s = ((i - 1) * B) % POOL
        e = pool[s:s + B]
        noise = torch.from_numpy(normal((B, 1, 768), seed=g["noise_seed_base"] + i)).cuda()
        t = torch.from_numpy(timesteps(B, seed=g["t_seed_base"] + i)).cuda()
        # ---- the reference's loop body
        optimizer.zero_grad(set_to_none=True)
        with torch.autocast("cuda", enabled=True):
            noisy_e = scheduler.add_noise(original_samples=e, noise=noise, timesteps=t)
            noise_pred = model(x=noisy_e, timesteps=t)
            loss = F.mse_loss(noise_pred.float(), noise.float())
        scaler.scale(loss).backward()
        scaler.step(optimizer)
        scaler.update()
        # ---- the fused native step on a twin model
        fused.zero_grad()
        ldm_train_step(fused, scheduler, e, noise, t, loss_out=lf)
        opt_f.step()
        want, got, gf = g["loss"][i - 1], float(loss), float(lf)
        worst = max(worst, abs(got - want) / want); worst_f = max(worst_f, abs(got - gf) / abs(gf))
        assert abs(got - want) <= tol * want + 1e-6, (dtype, i, got, want)
        assert abs(got - gf) <= (2e-4 if dtype == "float32" else 3e-2) * abs(gf), (dtype, i, got, gf)
    print(f"reference LDM loop body [{dtype}]: worst gap to the oracle {worst:.2e}, to the fused native step {worst_f:.2e} over {nsteps} steps")


@pytest.mark.parametrize("fixture", ["aekl_traj_thin.json", "aekl_traj_c1.json"])
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_reference_aekl_gan_loop_body_runs_on_the_engine_and_follows_the_oracle(dtype, fixture):
    """train_autoencoderkl.py:203-234 verbatim: model(images) -> discriminator(reconstruction)[-1] -> L1 / tensor-op KL / adversarial /
    spectral -> loss_g.backward(); optimizer_g.step(); discriminator on the detached reconstruction and on the images; loss_d.backward();
    optimizer_d.step().  (`eps=` on the model call injects the fixture's reparameterisation noise; the reference draws it inside.)"""
    from eegldm.losses import L1Loss, PatchAdversarialLoss, JukeboxLoss
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from eegldm.training import Adam, aekl_train_step
    from test_gpu_zz_convergence import aekl_bound
    g = _golden(fixture)
    mk_ae = lambda: AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=g["num_channels"], latent_channels=1, num_res_blocks=2,
                                  norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=0)
    mk_d = lambda: PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3,
                                      norm="BATCH", bias=False, padding=1, dtype=dtype, device=0)
    model, discriminator, ae_f, d_f = mk_ae(), mk_d(), mk_ae(), mk_d()
    sd_a = {k: torch.from_numpy(gen_param(g["param_seeds"][0], k, tuple(v.shape))) for k, v in model.state_dict().items()}
    sd_d = {k: (torch.from_numpy(gen_param(g["param_seeds"][1], k, tuple(v.shape))) if v.dtype.is_floating_point and "running" not in k and "num_batches" not in k else v)
            for k, v in discriminator.state_dict().items()}
    for m in (model, ae_f): m.load_state_dict(sd_a)
    for m in (discriminator, d_f): m.load_state_dict(sd_d)
    optimizer_g = torch.optim.Adam(model.parameters(), lr=g["lr"][0])
    optimizer_d = torch.optim.Adam(discriminator.parameters(), lr=g["lr"][1])
    og, od = Adam(ae_f, lr=g["lr"][0]), Adam(d_f, lr=g["lr"][1])
    l1_loss, adv_loss, jukebox_loss = L1Loss(), PatchAdversarialLoss(criterion="least_squares"), JukeboxLoss(spatial_dims=1, reduction="sum")
    w = g["weights"]; adv_weight, kl_weight, spectral_weight = w["adv"], w["kl"], w["spectral"]
    B, POOL = g["batch"], g["pool"]
    xs = torch.from_numpy(eeg_windows(POOL, seed=g["window_seed"])).cuda()
    lo = torch.zeros(6, device="cuda")
    nsteps = 16
    worst = {}
    for i in range(1, nsteps + 1):
        s = ((i - 1) * B) % POOL
        images = xs[s:s + B]
        ew = torch.from_numpy(normal((B, 1, 768), seed=g["eps_seed_base"] + i)).cuda()
        # ---- the reference's loop body (train_autoencoderkl.py:203-234)
        optimizer_g.zero_grad(set_to_none=True)
        reconstruction, z_mu, z_sigma = model(images, eps=ew)
        logits_fake = discriminator(reconstruction.contiguous().float())[-1]
        recons_loss = l1_loss(reconstruction.float(), images.float())
        spectral_loss = jukebox_loss(reconstruction.float(), images.float())
        kl_loss = 0.5 * torch.sum(z_mu.pow(2) + z_sigma.pow(2) - torch.log(z_sigma.pow(2)) - 1, dim=[1, 2])
        kl_loss = torch.sum(kl_loss) / kl_loss.shape[0]
        generator_loss = adv_loss(logits_fake, target_is_real=True, for_discriminator=False)
        loss_g = recons_loss + kl_weight * kl_loss + adv_weight * generator_loss + spectral_weight * spectral_loss
        loss_g.backward()
        optimizer_g.step()
        optimizer_d.zero_grad(set_to_none=True)
        logits_fake = discriminator(reconstruction.contiguous().detach())[-1]
        loss_d_fake = adv_loss(logits_fake, target_is_real=False, for_discriminator=True)
        logits_real = discriminator(images.contiguous().detach())[-1]
        loss_d_real = adv_loss(logits_real, target_is_real=True, for_discriminator=True)
        discriminator_loss = (loss_d_fake + loss_d_real) * 0.5
        loss_d = adv_weight * discriminator_loss
        loss_d.backward()
        optimizer_d.step()
        # ---- the fused native step on twin models
        ae_f.zero_grad(); d_f.zero_grad()
        aekl_train_step(ae_f, d_f, images, ew, adv_weight, kl_weight, spectral_weight, True, losses_out=lo)
        og.step(); od.step()
        v = [float(x) for x in lo.cpu()]
        fus = {"recons": v[0], "spectral": v[1], "kl": v[2], "gen": v[3], "disc": 0.5 * (v[4] + v[5])}
        got = {"recons": float(recons_loss), "spectral": float(spectral_loss), "kl": float(kl_loss), "gen": float(generator_loss), "disc": float(discriminator_loss)}
        want = g["losses"][i - 1]
        for k in got:
            assert abs(got[k] - want[k]) <= aekl_bound(dtype, k, i, want[k]), (dtype, fixture, i, k, got[k], want[k])
            assert abs(got[k] - fus[k]) <= 2.0 * aekl_bound(dtype, k, i, want[k]), (dtype, fixture, i, k, got[k], fus[k])     # two engine runs, each within the bound of the oracle
            worst[k] = max(worst.get(k, 0.0), abs(got[k] - fus[k]) / (abs(fus[k]) + 1e-12))
            if i <= 3 and dtype == "float32":          # before the GAN dynamics amplify the ~2e-7 differences of the fp32 atomics
                assert abs(got[k] - fus[k]) <= 2e-5 * abs(fus[k]) + 1e-7, (fixture, i, k, got[k], fus[k])
    # BatchNorm running statistics saw the same three updates per step in both paths
    sa, sb = discriminator.state_dict(), d_f.state_dict()
    for k in sa:
        if "num_batches_tracked" in k:
            assert int(sa[k]) == int(sb[k]) == 3 * nsteps, (k, int(sa[k]), int(sb[k]))
    print(f"reference AEKL/GAN loop body {fixture} [{dtype}]: worst gap to the fused native step over {nsteps} steps " + ", ".join(f"{k} {x:.1e}" for k, x in worst.items()))


def test_loss_functions_are_differentiable_and_match_torch():
    from eegldm.losses import L1Loss, PatchAdversarialLoss, JukeboxLoss, mse_loss
    from oracle import losses as Ls
    gen = torch.Generator().manual_seed(0)
    a = torch.randn(3, 1, 768, generator=gen); b = torch.randn(3, 1, 768, generator=gen)
    for name, fn, ref in [
        ("l1", lambda x: L1Loss()(x, b.cuda()), lambda x: (x - b).abs().mean()),
        ("mse", lambda x: mse_loss(x, b.cuda()), lambda x: F.mse_loss(x, b)),
        ("jukebox", lambda x: JukeboxLoss(spatial_dims=1, reduction="sum")(x, b.cuda()), lambda x: Ls.jukebox_loss(x, b, "sum")),
        ("lsgan_real", lambda x: PatchAdversarialLoss(criterion="least_squares")(x, target_is_real=True, for_discriminator=True), lambda x: Ls.patch_adv_loss(x, True, True)),
        ("lsgan_fake", lambda x: PatchAdversarialLoss(criterion="least_squares")(x, target_is_real=False, for_discriminator=True), lambda x: Ls.patch_adv_loss(x, False, True)),
    ]:
        xd = a.clone().cuda().requires_grad_(True); xc = a.clone().requires_grad_(True)
        ld = fn(xd); lc = ref(xc)
        (3.0 * ld).backward(); (3.0 * lc).backward()
        assert abs(float(ld) - float(lc)) <= 1e-4 * abs(float(lc)) + 1e-6, (name, float(ld), float(lc))
        err = float((xd.grad.cpu() - xc.grad).norm() / (xc.grad.norm() + 1e-12))
        assert err < 2e-4, (name, err)
        with torch.no_grad():      # no graph, no gradient buffer
            assert not fn(a.cuda()).requires_grad


def test_discriminator_backward_of_an_older_forward_reforwards_without_touching_running_statistics():
    """D(a) then D(b) then backward through BOTH (loss_d.backward() in train_autoencoderkl.py:225-233): the native executor only holds b's
    tape, so a's backward re-forwards a first -- with batch statistics but WITHOUT a running-statistics update."""
    from eegldm.models import PatchDiscriminator
    torch.manual_seed(0)
    mk = lambda: PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1, dtype="float32")
    d1, d2 = mk(), mk()
    d2.load_state_dict(d1.state_dict())
    gen = torch.Generator().manual_seed(1)
    a, b = torch.randn(4, 1, 768, generator=gen).cuda(), torch.randn(4, 1, 768, generator=gen).cuda()
    ga, gb = torch.randn(4, 1, 96, generator=gen).cuda(), torch.randn(4, 1, 96, generator=gen).cuda()
    d1.parameters(); d2.parameters()
    # both live, one backward
    la, lb = d1(a)[-1], d1(b)[-1]
    ((la * ga).sum() + (lb * gb).sum()).backward()
    # one at a time (each backward right after its own forward); same order of the running-statistics updates
    (d2(a)[-1] * ga).sum().backward()
    (d2(b)[-1] * gb).sum().backward()
    g1, g2 = d1.flat_grad, d2.flat_grad
    assert float((g1 - g2).norm() / g2.norm()) < 1e-5
    s1, s2 = d1.state_dict(), d2.state_dict()
    for k in s1:
        if "running" in k or "num_batches" in k:
            assert torch.allclose(s1[k].float(), s2[k].float(), rtol=1e-6, atol=1e-7), k


def test_intervening_native_calls_do_not_corrupt_an_older_graphs_backward():
    """ADVICE r4 (autograd.py): a forward that does not go through the autograd Function (torch.no_grad() validation call, eval mode,
    the fused train step) overwrites the executor's single tape; the bridge must notice and rebuild the tape of the graph being
    back-propagated.  Gradients after such an intervening call == gradients of an undisturbed forward / backward."""
    from eegldm.models import UNetModel, AutoencoderKL
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import ldm_train_step
    import eegldm
    eegldm.set_deterministic(True)
    try:
        small = dict(in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, attention_resolutions=[2], channel_mult=[1, 2], resblock_updown=True)
        net = UNetModel(image_size=64, **small, dtype="float32")
        sd = {k: torch.from_numpy(gen_param(7, k, tuple(v.shape))) for k, v in net.state_dict().items()}
        net.load_state_dict(sd)
        (p,) = net.parameters()
        x = torch.from_numpy(normal((4, 1, 64), seed=11)).cuda(); t = torch.from_numpy(timesteps(4, seed=12)).cuda()
        x2 = torch.from_numpy(normal((4, 1, 64), seed=13)).cuda() * 3.0; t2 = torch.from_numpy(timesteps(4, seed=14)).cuda()
        tgt = torch.from_numpy(normal((4, 1, 64), seed=15)).cuda()
        sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)

        def grads(disturb):
            net.train(); net.flat_grad.zero_(); p.grad = net.flat_grad
            loss = F.mse_loss(net(x, timesteps=t), tgt)
            if disturb == "no_grad":
                with torch.no_grad():
                    net(x2, timesteps=t2)
            elif disturb == "eval":
                net.eval(); net(x2, timesteps=t2); net.train()
            elif disturb == "fused":
                keep = net.flat_grad.clone()
                ldm_train_step(net, sched, x2, tgt, t2)
                net.flat_grad.copy_(keep)
            loss.backward()
            return net.flat_grad.clone()

        ref = grads(None)
        assert float(ref.abs().max()) > 0
        for d in ("no_grad", "eval", "fused"):
            g = grads(d)
            assert torch.equal(g, ref), f"{d}: max diff {float((g - ref).abs().max()):.3e}"

        # the autoencoder: encode() / decode() between forward and backward
        ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                           norm_num_groups=1, attention_levels=[False, False, False], dtype="float32")
        sd = {k: torch.from_numpy(gen_param(8, k, tuple(v.shape))) for k, v in ae.state_dict().items()}
        ae.load_state_dict(sd)
        (pa,) = ae.parameters()
        w = torch.from_numpy(eeg_windows(2, seed=5, length=256)).cuda(); eps = torch.from_numpy(normal((2, 1, 64), seed=6)).cuda()
        w2 = torch.from_numpy(eeg_windows(2, seed=9, length=256)).cuda() * 2.0

        def ae_grads(disturb):
            ae.train(); ae.flat_grad.zero_(); pa.grad = ae.flat_grad
            recon, mu, sg = ae(w, eps=eps)
            loss = F.l1_loss(recon, w) + 1e-3 * (mu.pow(2) + sg.pow(2)).sum()
            if disturb:
                with torch.no_grad():
                    m2, _ = ae.encode(w2); ae.decode(m2)
            loss.backward()
            return ae.flat_grad.clone()

        ref = ae_grads(False)
        assert float(ref.abs().max()) > 0
        g = ae_grads(True)
        assert torch.equal(g, ref), f"aekl: max diff {float((g - ref).abs().max()):.3e}"
    finally:
        eegldm.set_deterministic(False)
