"""-m gpu: oracle / reference parity at the BASELINE sizes the round-1 suite only property-tested.

* config_ldm.yaml UNet (model_channels 128, 278 keys, Cin up to 1024 -> split-K / skewed chunks / 192-row tiles / fused bias
  gradients at production K), B=2, L=768: forward, input gradient and every parameter gradient against the golden vectors
  produced by the imported reference itself (tests/golden/unet_full_l768.npz), fp32 and bf16.
* the LDM train step (L=768) and the pixel-space DM step (L=3072, T=768 attention) of the same UNet against
  oracle.steps.{ldm,dm}_train_step evaluated on the host, fp32 and bf16.
bf16 bounds come from the bf16-storage oracle (gpu_util.bf16_gap_bound)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from make_golden_cases import UNET_FULL  # noqa: E402
from param_gen import gen_param, normal, timesteps, eeg_windows  # noqa: E402


def _threads():
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_unet_full_size_vs_reference_golden(golden_dir, dtype):
    import gpu_util as G
    from eegldm.models import UNetModel
    from oracle import quant as Q, unet as U
    g = np.load(os.path.join(golden_dir, "unet_full_l768.npz"))
    cfg, B, L = UNET_FULL
    sw, sx, _st, sdy = [int(v) for v in g["seeds"]]
    net = UNetModel(**cfg, dtype=dtype)
    assert list(net.entries.keys()) == [str(k) for k in g["keys"]] and net.n_flat >= 30533121
    sd = {k: torch.from_numpy(gen_param(sw, k, shape)) for k, (_o, _n, shape) in net.entries.items()}
    net.load_state_dict(sd)
    x = torch.from_numpy(normal((B, 1, L), seed=sx)); t = torch.from_numpy(g["t"]); dy = torch.from_numpy(normal((B, 1, L), seed=sdy))
    y = net(x, timesteps=t)
    net.zero_grad()
    dx = net.backward(dy, need_dx=True)
    grads = net.grad_dict()
    f32 = dtype == "float32"
    gscale = max(float(g["g_l2:" + k]) for k in net.entries)
    if f32:
        G.assert_close(y, g["y"], rtol=2e-4, atol=5e-5, name="y")
        G.assert_close(dx, g["dx"], rtol=2e-3, atol=5e-5, name="dx")
        assert G.rel_l2(y, g["y"]) < 2e-5 and G.rel_l2(dx, g["dx"]) < 5e-5
        worst = 0.0
        for k in net.entries:
            gr = grads[k].double().reshape(-1).cpu(); l2 = float(g["g_l2:" + k]); floor = 1e-3 * gscale
            rel = abs(float(gr.norm()) - l2) / (l2 + floor)
            head = g["g_head:" + k].astype(np.float64)
            he = float(np.linalg.norm(gr[:16].numpy() - head)) / (float(np.linalg.norm(head)) + floor / np.sqrt(max(1, gr.numel() / 16)))
            se = abs(float(gr.sum()) - float(g["g_sum:" + k])) / (abs(float(g["g_sum:" + k])) + l2 + floor)
            worst = max(worst, rel, he, se)
            assert rel < 1e-3 and he < 2e-3 and se < 2e-3, f"{k}: norm {rel:.2e} head {he:.2e} sum {se:.2e}"
        print(f"full-size fp32 vs reference: y {G.rel_l2(y, g['y']):.2e} dx {G.rel_l2(dx, g['dx']):.2e} worst grad digest err {worst:.2e}")
        return
    # bf16: measure what the storage format alone does to this network (oracle with bf16 storage vs fp32 oracle), bound the engine by it
    _threads()
    def run(emul):
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        with Q.bf16_storage(emul):
            yo = U.unet_forward(p, cfg, xr, t)
            yo.backward(dy)
        return yo.detach(), xr.grad, {k: v.grad for k, v in p.items()}
    y32, dx32, g32 = run(False)
    yq, dxq, gq = run(True)
    np.testing.assert_allclose(y32.numpy(), g["y"], rtol=2e-4, atol=5e-5)       # the fp32 oracle IS the reference here
    gap_y, gap_dx = G.rel_l2(yq, y32), G.rel_l2(dxq, dx32)
    ey, edx = G.rel_l2(y, y32), G.rel_l2(dx, dx32)
    assert ey < G.bf16_gap_bound(gap_y) and edx < G.bf16_gap_bound(gap_dx), (ey, gap_y, edx, gap_dx)
    msg = G.assert_bf16_grads(grads, g32, gq, "UNet")
    print(f"full-size bf16: y {ey:.2e} (gap {gap_y:.2e}) dx {edx:.2e} (gap {gap_dx:.2e}); {msg}")


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
@pytest.mark.parametrize("workload", ["ldm_l768", "dm_l3072_spectral"])
def test_full_size_train_step_vs_oracle(workload, dtype):
    """training.py:419-443 (latents, L=768) and training_diffusion.py:141-151 (raw windows, L=3072, T=768 attention,
    spectral term on) with the config_ldm.yaml / config_dm.yaml UNet, B=2: loss and all 278 parameter gradients."""
    import gpu_util as G
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import ldm_train_step, dm_train_step
    from oracle import losses as Ls, quant as Q, steps as S, unet as U
    _threads()
    cfg, B, _L = UNET_FULL
    L = 768 if workload == "ldm_l768" else 3072
    cfg = dict(cfg, image_size=L)
    sd = {k: torch.from_numpy(gen_param(91, k, s)) for k, s in U.unet_param_shapes(cfg).items()}
    noise = torch.from_numpy(normal((B, 1, L), seed=92)); t = torch.from_numpy(timesteps(B, seed=93))
    acp = Ls.alphas_cumprod("linear_beta", 1000, 0.0015, 0.0195)
    if workload == "ldm_l768":
        x = torch.from_numpy(normal((B, 1, L), seed=94))
        ref = lambda: S.ldm_train_step(sd, cfg, acp, x, noise, t)
    else:
        x = torch.from_numpy(eeg_windows(B, seed=94, length=L))
        ref = lambda: S.dm_train_step(sd, cfg, acp, x, noise, t, spectral_weight=1e-3, spectral_loss=True)
    loss32, g32, pred32 = ref()
    net = UNetModel(**cfg, dtype=dtype); net.load_state_dict(sd)
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
    net.zero_grad()
    dev = net.device
    if workload == "ldm_l768":
        loss = ldm_train_step(net, sched, x.to(dev), noise.to(dev), t.to(dev))
    else:
        loss = dm_train_step(net, sched, x, noise, t.to(dev), spectral_weight=1e-3, spectral_loss=True)
    grads = net.grad_dict()
    if dtype == "float32":
        assert abs(float(loss) - float(loss32)) < 2e-4 * abs(float(loss32)), (float(loss), float(loss32))
        err = G.grads_rel_errors(grads, g32, 1e-3)
        k, e = max(err.items(), key=lambda kv: kv[1])
        assert e < 2e-3, f"{k}: {e:.3e}"
        print(f"{workload} fp32: loss {float(loss):.6f} vs {float(loss32):.6f}; worst grad {k} {e:.2e}")
        return
    with Q.bf16_storage(True):
        lossq, gq, _ = ref()
    gap_l = abs(float(lossq) - float(loss32)) / abs(float(loss32))
    el = abs(float(loss) - float(loss32)) / abs(float(loss32))
    assert el < G.bf16_gap_bound(gap_l, 2e-2), (float(loss), float(loss32), float(lossq))
    msg = G.assert_bf16_grads(grads, g32, gq, workload)
    print(f"{workload} bf16: loss err {el:.2e} (gap {gap_l:.2e}); {msg}")


AEKL_C2 = dict(num_channels=[2, 2, 4], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
D_CFG = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_fused_aekl_gan_step_config_2_2_4_spec(dtype):
    """BASELINE configs[1] (config_aekl_eeg_2_2_4_spec.yaml): the fused eegldm_aekl_train_step with channels [2,2,4], spectral loss
    ON, L=3072, B=8, two full G+D steps (losses, both gradient sets of step 1, BatchNorm running statistics) vs
    oracle.steps.aekl_train_step (train_autoencoderkl.py:203-234)."""
    import gpu_util as G
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from eegldm.training import Adam, aekl_train_step
    from oracle import aekl as A, quant as Q, steps as S
    _threads()
    B, L = 8, 3072
    adv_w, kl_w, spec_w = 0.01, 1e-6, 1e-2
    ae_sd0 = {k: torch.from_numpy(gen_param(131, k, s)) for k, s in A.aekl_param_shapes(AEKL_C2).items()}
    d_sd0 = {k: torch.from_numpy(gen_param(132, k, s)) for k, s in A.disc_param_shapes(D_CFG).items()}
    xs = [torch.from_numpy(eeg_windows(B, seed=140 + s, length=L)) for s in (1, 2)]
    es = [torch.from_numpy(normal((B, 1, L // 4), seed=150 + s)) for s in (1, 2)]

    def oracle_run(emul):
        ae_sd, d_sd, sg, sdd, rec = dict(ae_sd0), dict(d_sd0), {}, {}, []
        with Q.bf16_storage(emul):
            for step in (1, 2):
                losses, ae_sd, d_sd, recon, gg, dg = S.aekl_train_step(ae_sd, AEKL_C2, d_sd, D_CFG, xs[step - 1], es[step - 1], adv_w, kl_w, spec_w,
                                                                        True, 5e-3, 5e-4, step, sg, sdd)
                rec.append((losses, recon, gg, dg))
        return rec, ae_sd, d_sd
    rec32, ae32, d32 = oracle_run(False)
    f32 = dtype == "float32"
    recq = oracle_run(True)[0] if not f32 else None

    ae = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype=dtype, **AEKL_C2); ae.load_state_dict(ae_sd0)
    disc = PatchDiscriminator(**D_CFG, dtype=dtype); disc.load_state_dict(d_sd0)
    og, od = Adam(ae, lr=5e-3), Adam(disc, lr=5e-4)
    names = ["recons", "spectral", "kl", "gen"]
    for step in (1, 2):
        losses32, recon32, gg32, dg32 = rec32[step - 1]
        ae.zero_grad(); disc.zero_grad()
        rec_d = torch.empty(B, 1, L, device=ae.device)
        o = aekl_train_step(ae, disc, xs[step - 1].to(ae.device), es[step - 1].to(ae.device), adv_w, kl_w, spec_w, True, recon_out=rec_d).cpu()
        got_l = {n: float(o[i]) for i, n in enumerate(names)}; got_l["disc"] = 0.5 * float(o[4] + o[5])
        if step == 1:     # identical parameters on both sides: compare everything
            if f32:
                for n_, v in got_l.items():
                    assert abs(v - float(losses32[n_])) < 3e-4 * abs(float(losses32[n_])) + 1e-6, (n_, v, float(losses32[n_]))
                assert G.rel_l2(rec_d, recon32) < 5e-5
                eg = G.grads_rel_errors(ae.grad_dict(), gg32, 1e-3); ed = G.grads_rel_errors(disc.grad_dict(), dg32, 1e-3)
                kg, vg = max(eg.items(), key=lambda kv: kv[1]); kd, vd = max(ed.items(), key=lambda kv: kv[1])
                assert vg < 3e-3 and vd < 3e-3, (kg, vg, kd, vd)
                print(f"C2 fused step fp32: worst G grad {kg} {vg:.2e}, worst D grad {kd} {vd:.2e}")
            else:
                lq, reconq, ggq, dgq = recq[0]
                for n_, v in got_l.items():
                    ref = float(losses32[n_]); gap = abs(float(lq[n_]) - ref) / (abs(ref) + 1e-12)
                    assert abs(v - ref) / (abs(ref) + 1e-12) < G.bf16_gap_bound(gap, 3e-2), (n_, v, ref, float(lq[n_]))
                assert G.rel_l2(rec_d, recon32) < G.bf16_gap_bound(G.rel_l2(reconq, recon32))
                for nm, got, want, wq in (("G", ae.grad_dict(), gg32, ggq), ("D", disc.grad_dict(), dg32, dgq)):
                    print("C2 fused step bf16", G.assert_bf16_grads(got, want, wq, nm, floor_frac=3e-2, factor=2.5))
        og.step(); od.step()
    # after two Adam steps: parameters in units of lr (see test_gpu_aekl.py).  Adam normalises every gradient to ~+-lr, so an element
    # whose true gradient is rounding noise may move by up to 2 lr against the oracle; the mean deviation is therefore taken over the
    # whole model (934 / 519 681 elements), not per tensor (a [2,2,4] GroupNorm scale has 2 elements).
    got = ae.state_dict()
    devs = []
    for k, v in ae32.items():
        d = (got[k].cpu() - v).abs()
        assert float(d.max()) < (2.5 if f32 else 4.1) * 5e-3, f"ae {k}: max {float(d.max()):.3e}"     # 4 lr = two steps in opposite directions
        devs.append(d.reshape(-1))
    mean_dev = float(torch.cat(devs).mean())
    assert mean_dev < (0.05 if f32 else 0.25) * 5e-3, f"ae mean |dp| {mean_dev:.3e}"
    gotd = disc.state_dict()
    devs = []
    for k, v in d32.items():
        if "num_batches" in k:
            assert int(gotd[k]) == 6
        elif "running" in k:
            assert G.rel_l2(gotd[k], v) < (1e-4 if f32 else 3e-2), k
        else:
            d = (gotd[k].cpu().float() - v.float()).abs()
            assert float(d.max()) < (2.5 if f32 else 4.1) * 5e-4, f"disc {k}: max {float(d.max()):.3e}"
            devs.append(d.reshape(-1))
    mean_dev = float(torch.cat(devs).mean())
    assert mean_dev < (0.05 if f32 else 0.25) * 5e-4, f"disc mean |dp| {mean_dev:.3e}"
