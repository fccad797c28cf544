"""-m gpu, collected LAST (zz): the long / chaotic tests.  A GAN trajectory amplifies one-ulp differences, so nothing here may sit in front
of the deterministic parity tests under `pytest -x` (round 3 lost 219 tests to one assertion of this file).

The production train step TRAINS -- 40 optimiser steps of the config_ldm.yaml UNet over frozen AutoencoderKL latents
(train_ldm.py:199-204 schedule and scale factor; training.py:399-452 loop) on a fixed pool of synthetic windows:

* the epsilon-MSE falls from ~1.0 (zero-initialised output conv) to well below it,
* the bf16 engine follows the fp32 engine's loss curve on the same seeds (Philox streams are engine-independent),
* torch's allocator does not grow over the run (every step reuses the context workspace).
A single-step parity test cannot see a gradient that is slightly wrong in a way that stalls training; this one does.
The long form (500 steps, batch 256) is tools/soak_ldm.py -> profiles/r03_soak_ldm.log.txt."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_ldm_training_converges_and_bf16_follows_fp32():
    import soak_ldm
    B, steps = 64, 40
    b = soak_ldm.run("bfloat16", steps, B, 768, 512, 10, 1e-4, True)
    f = soak_ldm.run("float32", steps, B, 768, 512, 10, 1e-4, False)
    cb, cf = dict(b["curve"]), dict(f["curve"])
    assert abs(cb[0] - 1.0) < 0.05 and abs(cf[0] - 1.0) < 0.05, (cb[0], cf[0])       # zero-initialised head: loss = E[noise^2]
    assert cb[steps - 1] < 0.6 * cb[0] and cf[steps - 1] < 0.6 * cf[0], (cb, cf)
    for i in cf:
        assert abs(cb[i] - cf[i]) / cf[i] < 0.03, (i, cb[i], cf[i])                  # measured: 0.05 % at step 25, 0.8 % at step 50 (B=256)
    assert b["alloc_first_last"][1] <= b["alloc_first_last"][0] and f["alloc_first_last"][1] <= f["alloc_first_last"][0]
    assert b["sample"]["finite"], b["sample"]       # 40 steps do not make a usable epsilon model (latent std ~40; 1.4 after 500 steps): only finiteness


def _golden(name):
    import json
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name)) as fh:
        return json.load(fh)


sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
UCFG = dict(in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)


def replay_ldm(dtype, tol=None):
    """30 optimiser steps of ldm_traj_c2.json on the engine; returns the worst relative loss gap (asserts per step when tol is given)."""
    import torch
    from param_gen import gen_param, eeg_windows, normal, timesteps
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, ldm_train_step
    g = _golden("ldm_traj_c2.json")
    net = UNetModel(image_size=768, **UCFG, dtype=dtype)
    net.load_state_dict({k: torch.from_numpy(gen_param(g["param_seed"], k, tuple(v.shape))) for k, v in net.state_dict().items()})
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
    opt = Adam(net, lr=g["lr"])
    B, POOL = g["batch"], g["pool"]
    pool = torch.from_numpy(eeg_windows(POOL, seed=g["latent_seed"], length=768)).cuda()
    loss = torch.zeros(1, device="cuda")
    worst = 0.0
    for i in range(1, g["steps"] + 1):
        s = ((i - 1) * B) % POOL
        nz = torch.from_numpy(normal((B, 1, 768), seed=g["noise_seed_base"] + i)).cuda()
        t = torch.from_numpy(timesteps(B, seed=g["t_seed_base"] + i)).cuda()
        net.zero_grad()
        ldm_train_step(net, sched, pool[s:s + B], nz, t, loss_out=loss)
        opt.step()
        want = g["loss"][i - 1]; got = float(loss)
        worst = max(worst, abs(got - want) / want)
        if tol is not None:
            assert abs(got - want) <= tol * want + 1e-6, (dtype, i, got, want)
    return worst


@pytest.mark.parametrize("dtype,tol", [("float32", 2e-3), ("bfloat16", 6e-2)])
def test_ldm_training_trajectory_matches_the_oracle(dtype, tol):
    """30 optimiser steps of the config_ldm.yaml UNet (add_noise -> UNet -> MSE -> Adam, training.py:419-443) on seeded weights, latents,
    noise and timesteps: the loss of EVERY step against the CPU oracle's trajectory (tests/golden/make_ldm_traj.py -> ldm_traj_c2.json).
    A gradient or optimiser defect that a one-step parity test tolerates compounds here: the fp32 engine must stay within 0.2 %
    of the oracle over the whole run (measured: 1.7e-5 while the loss falls 1.75 -> 0.016; 100x margin), the bf16 engine within 6 % (measured: 1.0 %)."""
    worst = replay_ldm(dtype, tol)
    print(f"LDM trajectory [{dtype}]: worst relative loss gap over 30 steps {worst:.2e}")


def test_16bit_ldm_trajectories_beside_the_reference_loops_autocast_runs():
    """ldm_traj_c2_reference.json: the reference's own loop body on its own UNetModel, fp32 (= the oracle's fixture to 3.5e-7), under bf16
    autocast, and under fp16 autocast + GradScaler (its production setting).  The bf16 engine's per-step loss stays as close to fp32 as the
    reference's bf16 run does (factor 1.5 on the worst step), and so does the fp16 engine beside the reference's fp16 run."""
    import torch
    from param_gen import gen_param, eeg_windows, normal, timesteps
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, GradScaler, ldm_train_step
    g, r = _golden("ldm_traj_c2.json"), _golden("ldm_traj_c2_reference.json")
    ref32 = r["loss_fp32"]
    worst_ref_bf16 = max(abs(a - b) / b for a, b in zip(r["loss_bf16"], ref32))
    worst_bf16 = replay_ldm("bfloat16")
    print(f"bf16: engine worst gap to fp32 {worst_bf16:.2e}; reference bf16-autocast worst gap {worst_ref_bf16:.2e}")
    assert worst_bf16 <= 1.5 * worst_ref_bf16, (worst_bf16, worst_ref_bf16)
    # fp16 + GradScaler.  At torch's default initial scale (65536, as the reference constructs it: training.py:334) torch's CPU fp16 kernels
    # overflow twice and that curve runs two optimiser steps behind; the engine accumulates in fp32, rounds to fp16 only when it stores, and does
    # not overflow there.  The step-by-step comparison is made at init_scale = 1024, where neither side backs off.
    def fp16_run(init_scale):
        net = UNetModel(image_size=768, **UCFG, dtype="float16")
        net.load_state_dict({k: torch.from_numpy(gen_param(g["param_seed"], k, tuple(v.shape))) for k, v in net.state_dict().items()})
        sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
        opt, scaler = Adam(net, lr=g["lr"]), GradScaler(init_scale=init_scale)
        B, POOL = g["batch"], g["pool"]
        pool = torch.from_numpy(eeg_windows(POOL, seed=g["latent_seed"], length=768)).cuda()
        loss = torch.zeros(1, device="cuda")
        got, backoffs = [], []
        for i in range(1, g["steps"] + 1):
            s = ((i - 1) * B) % POOL
            nz = torch.from_numpy(normal((B, 1, 768), seed=g["noise_seed_base"] + i)).cuda()
            t = torch.from_numpy(timesteps(B, seed=g["t_seed_base"] + i)).cuda()
            net.zero_grad()
            before = scaler.get_scale()
            ldm_train_step(net, sched, pool[s:s + B], nz, t, loss_out=loss, grad_scale=before)
            scaler.step(opt); scaler.update()
            if scaler.get_scale() < before:
                backoffs.append(i)
            got.append(float(loss))
        return got, backoffs
    got, backoffs = fp16_run(1024.0)
    assert backoffs == [] and r["f16_scale1024_backoffs"] == 0
    ref16 = r["loss_f16_scale1024"]
    worst_ref_f16 = max(abs(a - b) / b for a, b in zip(ref16, ref32))
    worst_f16 = max(abs(a - b) / b for a, b in zip(got, ref32))
    print(f"fp16 (init_scale 1024): engine worst gap to fp32 {worst_f16:.2e}; reference fp16-autocast worst gap {worst_ref_f16:.2e}")
    assert worst_f16 <= 1.5 * worst_ref_f16 + 1e-3, (worst_f16, worst_ref_f16)
    got_d, backoffs_d = fp16_run(65536.0)
    print("fp16 at the default scale: engine back-offs", backoffs_d, "| reference back-offs", r["f16_scaler_backoffs"])
    assert r["f16_scaler_backoffs"] == 2 and len(backoffs_d) <= 2
    if not backoffs_d:
        assert max(abs(a - b) / b for a, b in zip(got_d, ref32)) <= 1.5 * worst_ref_f16 + 1e-3


def replay_aekl(fixture, dtype):
    """The 40 GAN steps of an aekl_traj_*.json fixture on the engine.  Returns {"rel": {term: [per-step |got-want|/|want|]}, "got": ..., "want": ...}."""
    import torch
    from param_gen import gen_param, eeg_windows, normal
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from eegldm.training import Adam, aekl_train_step
    g = _golden(fixture)
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=g["num_channels"], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=0)
    disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3,
                              norm="BATCH", bias=False, padding=1, dtype=dtype, device=0)
    ae.load_state_dict({k: torch.from_numpy(gen_param(g["param_seeds"][0], k, tuple(v.shape))) for k, v in ae.state_dict().items()})
    disc.load_state_dict({k: (torch.from_numpy(gen_param(g["param_seeds"][1], k, tuple(v.shape)))
                              if v.dtype.is_floating_point and "running" not in k and "num_batches" not in k else v) for k, v in disc.state_dict().items()})
    og, od = Adam(ae, lr=g["lr"][0]), Adam(disc, lr=g["lr"][1])
    B, POOL, w = g["batch"], g["pool"], g["weights"]
    xs = torch.from_numpy(eeg_windows(POOL, seed=g["window_seed"])).cuda()
    lo = torch.zeros(6, device="cuda")
    terms = ("recons", "spectral", "kl", "gen", "disc")
    out = {"rel": {k: [] for k in terms}, "got": {k: [] for k in terms}, "want": {k: [] for k in terms}}
    for i in range(1, g["steps"] + 1):
        s = ((i - 1) * B) % POOL
        ew = torch.from_numpy(normal((B, 1, 768), seed=g["eps_seed_base"] + i)).cuda()
        ae.zero_grad(); disc.zero_grad()
        aekl_train_step(ae, disc, xs[s:s + B], ew, w["adv"], w["kl"], w["spectral"], True, losses_out=lo)
        og.step(); od.step()
        v = [float(x) for x in lo.cpu()]
        got = {"recons": v[0], "spectral": v[1], "kl": v[2], "gen": v[3], "disc": 0.5 * (v[4] + v[5])}
        want = g["losses"][i - 1]
        for k in terms:
            out["got"][k].append(got[k]); out["want"][k].append(want[k])
            out["rel"][k].append(abs(got[k] - want[k]) / (abs(want[k]) + 1e-12))
    return out


# Bounds of the AEKL / GAN trajectories (profiles/r04_traj_spread_*.txt: 5 replays on each of 2 boxes).
#  * reconstruction L1, spectral, KL: smooth functionals of the generator -- tight relative bound at EVERY step.
#  * generator / discriminator adversarial terms: D and G chase each other, a one-ulp difference in a weight gradient (the engine's fp32
#    sums run in another order than torch's) is amplified step after step.  Round 3 bounded them by ONE measured run x 1.6 and went red on
#    the next box.  Now: the first ADV_TIGHT_STEPS steps, before the amplification sets in, keep the tight bound (a wrong adversarial
#    gradient shows there at O(1)); later steps get an envelope that grows geometrically from the tight bound, capped at ADV_CAP.
ADV_TIGHT_STEPS = 10
ADV_GROWTH = 1.12
ADV_CAP = {"float32": 3e-2, "bfloat16": 0.25}
TIGHT = {"float32": (2e-3, 2e-4), "bfloat16": (8e-2, 2e-2)}        # (relative, absolute floor)


def aekl_bound(dtype, term, step, want):
    rel, floor = TIGHT[dtype]
    if term == "spectral" and dtype == "bfloat16":
        rel = 2 * rel          # a sum of squared amplitude differences (x 1e4 in the loss): the most rounding-sensitive term, the bf16 run sits 3-9 % below
    if term in ("gen", "disc") and step > ADV_TIGHT_STEPS:
        rel = min(rel * ADV_GROWTH ** (step - ADV_TIGHT_STEPS), ADV_CAP[dtype])
        floor = min(floor * ADV_GROWTH ** (step - ADV_TIGHT_STEPS), 10 * floor)
    return rel * abs(want) + floor


@pytest.mark.parametrize("fixture", ["aekl_traj_c1.json", "aekl_traj_thin.json"])      # [32,32,64] layer by layer; [2,2,4] = the whole-network aekl_thin kernels
@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_aekl_gan_training_trajectory_matches_the_oracle(dtype, fixture):
    """40 optimiser steps of the AutoencoderKL + PatchDiscriminator GAN training (train_autoencoderkl.py:203-234, reference
    loss weights incl. the 1e4 x spectral term, both Adam updates, BatchNorm running statistics): reconstruction L1, spectral, KL, generator
    and discriminator losses of EVERY step against the CPU oracle's trajectory (tests/golden/make_aekl_traj.py -> aekl_traj_*.json).
    Bounds: aekl_bound above; measured spread: profiles/r04_traj_spread_*.txt."""
    r = replay_aekl(fixture, dtype)
    for k in r["rel"]:
        for i, (got, want) in enumerate(zip(r["got"][k], r["want"][k]), start=1):
            assert abs(got - want) <= aekl_bound(dtype, k, i, want), (dtype, fixture, i, k, got, want, aekl_bound(dtype, k, i, want))
    print(f"AEKL/GAN trajectory {fixture} [{dtype}]: worst relative gaps over 40 steps " + ", ".join(f"{k} {max(x):.1e}" for k, x in r["rel"].items()))


def replay_dm(dtype, tol=None):
    import torch
    from param_gen import gen_param, eeg_windows, normal, timesteps
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, dm_train_step
    g = _golden("dm_traj_c5.json")
    net = UNetModel(image_size=3072, **UCFG, dtype=dtype)
    net.load_state_dict({k: torch.from_numpy(gen_param(g["param_seed"], k, tuple(v.shape))) for k, v in net.state_dict().items()})
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
    opt = Adam(net, lr=g["lr"])
    B, POOL = g["batch"], g["pool"]
    pool = torch.from_numpy(eeg_windows(POOL, seed=g["window_seed"])).cuda()
    loss = torch.zeros(1, device="cuda")
    worst = 0.0
    for i in range(1, g["steps"] + 1):
        s = ((i - 1) * B) % POOL
        nz = torch.from_numpy(normal((B, 1, 3072), seed=g["noise_seed_base"] + i)).cuda()
        t = torch.from_numpy(timesteps(B, seed=g["t_seed_base"] + i)).cuda()
        net.zero_grad()
        dm_train_step(net, sched, pool[s:s + B], nz, t, spectral_weight=g["spectral_weight"], spectral_loss=True, loss_out=loss)
        opt.step()
        want = g["loss"][i - 1]; got = float(loss)
        worst = max(worst, abs(got - want) / want)
        if tol is not None:
            assert abs(got - want) <= tol * want + 1e-6, (dtype, i, got, want)
    return worst


@pytest.mark.parametrize("dtype,tol", [("float32", 2e-3), ("bfloat16", 8e-2)])
def test_pixel_dm_training_trajectory_matches_the_oracle(dtype, tol):
    """12 optimiser steps of the pixel-space diffusion model (training_diffusion.py:141-151: the config_dm.yaml UNet on raw (B,1,3072) windows,
    T = 768 attention, epsilon MSE + 1e-6 x JukeboxLoss(sum), Adam 1e-4) against the CPU oracle's trajectory
    (tests/golden/make_dm_traj.py -> dm_traj_c5.json), every step.  Measured: fp32 7.6e-6 (260x margin), bf16 1.2 %."""
    worst = replay_dm(dtype, tol)
    print(f"pixel-space DM trajectory [{dtype}]: worst relative loss gap over 12 steps {worst:.2e}")


# ---------------------------------------------------------------------------------------------------------------- round 5
def test_trajectories_in_deterministic_mode_are_reproducible_and_tight():
    """EEGLDM_DETERMINISTIC=1 removes the engine's own run-to-run spread (the 2e-7 of the fp32 gradient atomics that the GAN amplifies), so what
    is left between engine and oracle is the distance between two fp32 implementations.  (i) two replays of each fp32 trajectory are
    IDENTICAL, loss for loss; (ii) the smooth terms hold bounds an order tighter than the default-mode envelopes: LDM 2e-4 (default bound
    2e-3), pixel-space DM 2e-4 (2e-3), AEKL recons / KL 2e-4 and spectral 4e-4 (2e-3); (iii) the adversarial terms -- where the oracle-side
    fp32 rounding is amplified just the same -- hold the TIGHT default bound (2e-3) for the first 10 steps and HALF the default envelope after."""
    import eegldm
    eegldm.set_deterministic(True)
    try:
        w1 = replay_ldm("float32", 2e-4); w2 = replay_ldm("float32", 2e-4)
        assert w1 == w2, (w1, w2)
        d1 = replay_dm("float32", 2e-4)
        a = replay_aekl("aekl_traj_thin.json", "float32"); b = replay_aekl("aekl_traj_thin.json", "float32")
        assert a["got"] == b["got"], "the deterministic mode did not reproduce the AEKL / GAN trajectory"
        for k in ("recons", "kl", "spectral"):
            lim = 1e-4 if k == "spectral" else 5e-5          # measured 5.8e-6 / 2.7e-6 / 5.6e-6
            for i, (got, want) in enumerate(zip(a["got"][k], a["want"][k]), start=1):
                assert abs(got - want) <= lim * abs(want) + 2e-5, (k, i, got, want)
        for k in ("gen", "disc"):
            for i, (got, want) in enumerate(zip(a["got"][k], a["want"][k]), start=1):
                bound = aekl_bound("float32", k, i, want)
                assert abs(got - want) <= (bound if i <= ADV_TIGHT_STEPS else 0.5 * bound), (k, i, got, want, bound)
        print(f"deterministic mode: LDM {w1:.2e}, DM {d1:.2e}, AEKL " + ", ".join(f"{k} {max(x):.1e}" for k, x in a["rel"].items()))
    finally:
        eegldm.set_deterministic(False)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_single_step_gradient_parity_after_warm_up_from_the_oracles_own_weights(dtype):
    """the trajectory envelopes widen after step 10, so a gradient defect of the generator / discriminator path that only shows once
    the BatchNorm statistics, the clamp of log_var or the LeakyReLU masks have moved away from their initial regime could hide inside them.
    Independent of the chaotic trajectory: the CPU oracle runs the [2,2,4] GAN training (train_autoencoderkl.py:203-234) for 40 steps; at
    steps 20 and 40 the engine is loaded with the ORACLE's weights and BatchNorm buffers of that moment and takes ONE step on the same
    batch -- losses and every parameter gradient of both networks against the oracle's, at single-step tolerance."""
    import torch
    from param_gen import gen_param, eeg_windows, normal
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from eegldm.training import aekl_train_step
    import oracle.aekl as A
    import oracle.steps as S
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    ACFG = dict(num_channels=[2, 2, 4], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    DCFG = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
    B, POOL, L = 4, 32, 3072
    st = {"ae": {k: torch.from_numpy(gen_param(42, k, s)) for k, s in A.aekl_param_shapes(ACFG).items()},
          "d": {k: torch.from_numpy(gen_param(43, k, s)) for k, s in A.disc_param_shapes(DCFG).items()}, "og": {}, "od": {}}
    xs = torch.from_numpy(eeg_windows(POOL, seed=778))
    ae = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype=dtype, **ACFG)
    disc = PatchDiscriminator(**DCFG, dtype=dtype)
    f32 = dtype == "float32"
    checked = 0
    for i in range(1, 41):
        s = ((i - 1) * B) % POOL
        x = xs[s:s + B]; ew = torch.from_numpy(normal((B, 1, L // 4), seed=300 + i))
        before = ({k: v.clone() for k, v in st["ae"].items()}, {k: v.clone() for k, v in st["d"].items()})
        l, st["ae"], st["d"], _r, gg, dg = S.aekl_train_step(st["ae"], ACFG, st["d"], DCFG, x, ew, 0.01, 1e-6, 1.0, True, 5e-3, 5e-4, i, st["og"], st["od"])
        if i not in (20, 40):
            continue
        ae.load_state_dict(before[0]); disc.load_state_dict(before[1])
        ae.zero_grad(); disc.zero_grad()
        o = aekl_train_step(ae, disc, x.cuda(), ew.cuda(), 0.01, 1e-6, 1.0, True).cpu()
        tol = 3e-4 if f32 else 6e-2
        want = [l["recons"], l["spectral"], l["kl"], l["gen"]]
        for j, w in enumerate(want):
            assert abs(float(o[j]) - float(w)) < tol * abs(float(w)) + 1e-6, (i, j, float(o[j]), float(w))
        assert abs(0.5 * float(o[4] + o[5]) - float(l["disc"])) < tol * float(l["disc"]) + 1e-6
        for name, got, ref in (("generator", ae.grad_dict(), gg), ("discriminator", disc.grad_dict(), dg)):
            gscale = max(float(v.norm()) for v in ref.values())
            worst = ("", 0.0)
            for k, w in ref.items():
                e = float((got[k].cpu().double() - w.double()).norm()) / (float(w.norm()) + (1e-3 if f32 else 3e-2) * gscale)
                if e > worst[1]:
                    worst = (k, e)
            assert worst[1] < (2e-3 if f32 else 0.12), f"step {i} {name}: {worst[0]} rel err {worst[1]:.3e}"
            print(f"step {i} [{dtype}] {name}: worst gradient {worst[0]} {worst[1]:.2e}")
        checked += 1
    assert checked == 2
