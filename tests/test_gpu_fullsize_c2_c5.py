"""-m gpu: the two BASELINE configurations that only bench.py used to run at full size, checked through size-independent
properties (the B = 8 / B = 2 oracle comparisons are tests/test_gpu_fullsize_parity.py):

* configs[1] config_aekl_eeg_2_2_4_spec.yaml -- fused AutoencoderKL[2,2,4] + PatchDiscriminator GAN step with the spectral loss,
  **B = 256, L = 3072, bf16** (train_autoencoderkl.py:203-234).  At this batch the kernel selection differs from B = 8: the
  whole-network autoencoder kernels run 256 workgroups, the discriminator's BatchNorm reductions fold 100+ partial-sum blocks through
  the alternating self-cleaning sum areas, the GEMM tiles use the XCD-aware order and split-K weight gradients.
    - the autoencoder is per-sample (GroupNorm with one group, no BatchNorm): reconstruction i of the 256-batch == the same window
      run in a batch of 3 and in the oracle-tested batch of 8; per-sample L1 terms agree; the step's L1 / spectral / KL losses equal
      the reductions recomputed on the host from the returned reconstruction (and from forward()'s mu / sigma);
    - the discriminator's BatchNorm couples the batch, so its results are compared with the fp32 ENGINE on the same 256 windows
      (that engine is oracle-tested to 3e-3 at B = 8): running statistics after the step's three updates, the three adversarial
      losses, and both flat gradient buffers;
* configs[4] config_dm.yaml -- pixel-space UNet step on raw windows, **B = 64, L = 3072** (training_diffusion.py:133-158): batch
  independence (B = 64 vs 3: split vs one-pass GroupNorm, T = 768 attention blocks) and bf16 results bounded by the gap the
  oracle's bf16-storage emulation opens against the fp32 oracle on the same windows, fp32 engine == fp32 oracle, exact linearity of the
  backward in dy, checksum of the last conv's bias gradient, and one dm_train_step (mse + 1e-6 x spectral) bf16 vs fp32 engine."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from param_gen import gen_param, normal, eeg_windows  # noqa: E402

AEKL_C2 = dict(num_channels=[2, 2, 4], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
D_CFG = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
ADV_W, KL_W, SPEC_W = 0.01, 1e-6, 1e-2
B2, L2 = 256, 3072


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1); b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


def _gan_nets(dtype):
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from oracle import aekl as A
    ae_sd = {k: torch.from_numpy(gen_param(131, k, s)) for k, s in A.aekl_param_shapes(AEKL_C2).items()}
    d_sd = {k: torch.from_numpy(gen_param(132, k, s)) for k, s in A.disc_param_shapes(D_CFG).items()}
    ae = AutoencoderKL(spatial_dims=1, attention_levels=[False] * 3, dtype=dtype, **AEKL_C2); ae.load_state_dict(ae_sd)
    disc = PatchDiscriminator(**D_CFG, dtype=dtype); disc.load_state_dict(d_sd)
    return ae, disc


def _gan_step(dtype, x, eps):
    from eegldm.training import aekl_train_step
    ae, disc = _gan_nets(dtype)
    ae.zero_grad(); disc.zero_grad()
    rec = torch.empty(x.shape[0], 1, x.shape[2], device=ae.device)
    losses = aekl_train_step(ae, disc, x.to(ae.device), eps.to(ae.device), ADV_W, KL_W, SPEC_W, True, recon_out=rec).cpu()
    torch.cuda.synchronize()
    return dict(losses=losses, recon=rec.cpu(), g=ae.flat_grad.clone().cpu(), d=disc.flat_grad.clone().cpu(), buffers=disc.state_dict(), ae=ae, disc=disc)


@pytest.fixture(scope="module")
def gan_runs():
    x = torch.from_numpy(eeg_windows(B2, seed=171, length=L2)); eps = torch.from_numpy(normal((B2, 1, L2 // 4), seed=172))
    return x, eps, _gan_step("bfloat16", x, eps), _gan_step("float32", x, eps)


def test_c2_gan_step_b256_autoencoder_is_per_sample(gan_runs):
    x, eps, rb, _rf = gan_runs
    assert torch.isfinite(rb["losses"]).all() and torch.isfinite(rb["g"]).all() and torch.isfinite(rb["d"]).all()
    assert float(rb["g"].abs().max()) > 0 and float(rb["d"].abs().max()) > 0
    ae = rb["ae"]
    for idx in ([0, 1, 2], [126, 127, 128], [253, 254, 255]):
        r3, _mu, _sg = ae(x[idx], eps=eps[idx])
        # same whole-network kernel, other grid: bit-identical per window (nothing in the autoencoder couples windows)
        assert torch.equal(r3.cpu(), rb["recon"][idx]), (idx, rel_l2(r3, rb["recon"][idx]))
    # the oracle-tested batch of 8 (tests/test_gpu_fullsize_parity.py::test_fused_aekl_gan_step_config_2_2_4_spec): same reconstructions
    # and per-sample L1 terms for the windows the two batches share
    r8 = _gan_step("bfloat16", x[:8], eps[:8])
    assert torch.equal(r8["recon"], rb["recon"][:8])
    l1_256 = (rb["recon"][:8] - x[:8]).abs().mean(dim=(1, 2)); l1_8 = (r8["recon"] - x[:8]).abs().mean(dim=(1, 2))
    assert torch.equal(l1_256, l1_8)
    assert abs(float(r8["losses"][0]) - float(l1_8.double().mean())) < 1e-5 * float(l1_8.mean())


def test_c2_gan_step_b256_losses_match_host_reductions(gan_runs):
    """L1 (mean over every element, pads included), spectral (sum over B x 3072 bins of (|FFT_ortho(recon)| - |FFT_ortho(x)|)^2,
    two-sided: SURVEY Appendix C.9) and KL (0.5 sum(mu^2 + sigma^2 - ln sigma^2 - 1) / B, C.8) recomputed on the host in fp64 from the
    tensors the engine returned: the 256-workgroup loss reductions, independent of the kernels that produced the tensors."""
    x, eps, rb, _rf = gan_runs
    rec = rb["recon"].double(); xd = x.double()
    l1 = float((rec - xd).abs().mean())
    spec = float(((torch.fft.fft(rec, dim=-1, norm="ortho").abs() - torch.fft.fft(xd, dim=-1, norm="ortho").abs()) ** 2).sum())
    _r, mu, sg = rb["ae"](x, eps=eps)
    mu, sg = mu.double().cpu(), sg.double().cpu()
    kl = float(0.5 * (mu ** 2 + sg ** 2 - torch.log(sg ** 2) - 1).sum() / B2)
    got = [float(v) for v in rb["losses"][:3]]
    assert abs(got[0] - l1) < 2e-5 * l1, (got[0], l1)
    assert abs(got[1] - spec) < 2e-4 * spec, (got[1], spec)          # fp32 LDS FFT vs fp64 host FFT
    assert abs(got[2] - kl) < 2e-3 * abs(kl) + 1e-6, (got[2], kl)    # mu / sigma come back through bf16-free fp32 buffers; the sum is 196 608 terms


def test_c2_gan_step_b256_bf16_tracks_fp32_engine(gan_runs):
    _x, _eps, rb, rf = gan_runs
    # BatchNorm running statistics after the step's three updates (generator pass, fake, real): momentum 0.1, unbiased variance
    nb = 0
    for k, v in rf["buffers"].items():
        if "running" not in k:
            continue
        nb += 1
        e = rel_l2(rb["buffers"][k], v)
        assert e < 2e-2, (k, e)
    assert nb == 6
    for k in ("0.adn.N.num_batches_tracked", "2.adn.N.num_batches_tracked"):
        assert int(rb["buffers"][k]) == int(rf["buffers"][k]) == 3        # three forwards per step
    names = ["recons", "spectral", "kl", "gen", "d_fake", "d_real"]
    tol = [2e-2, 5e-2, 2e-2, 5e-2, 5e-2, 5e-2]
    for i, n in enumerate(names):
        a, b = float(rb["losses"][i]), float(rf["losses"][i])
        assert abs(a - b) <= tol[i] * abs(b) + 1e-7, (n, a, b)
    # gradients of both networks: whole-buffer relative L2 (per-tensor gap-derived bounds are the B = 8 oracle test's job).  The
    # discriminator's are sums over 256 x 384..1536 positions of bf16-rounded products; the [2,2,4] autoencoder's 934 gradients pass
    # through the discriminator's input gradient and the spectral term
    ed, eg = rel_l2(rb["d"], rf["d"]), rel_l2(rb["g"], rf["g"])
    assert ed < 6e-2 and eg < 1.5e-1, (ed, eg)
    print(f"C2 B=256 bf16 vs fp32 engine: D grads {ed:.2e}, G grads {eg:.2e}")


# ------------------------------------------------------------------------------------------------ configs[4]: pixel-space model
DM_CFG = dict(image_size=3072, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
              channel_mult=[1, 2, 4], resblock_updown=True)       # config_dm.yaml (train_pure_ldm.py:113-115 forces in/out channels to 1)
B5, L5 = 64, 3072


def _weights(net, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = net.state_dict()
    return {k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()}


@pytest.fixture(scope="module")
def dm_net():
    from eegldm.models import UNetModel
    nb = UNetModel(**DM_CFG, dtype="bfloat16"); w = _weights(nb, 3); nb.load_state_dict(w)
    x = torch.from_numpy(eeg_windows(B5, seed=181, length=L5))
    t = torch.randint(0, 1000, (B5,), generator=torch.Generator().manual_seed(4))
    return nb, w, x, t


def _bf16_gap_l3072(w, x, t):
    """What bf16 STORAGE alone does to this network at L = 3072 (oracle with bf16 storage emulated at every activation / weight read
    vs the fp32 oracle, oracle/quant.py) -- the yardstick the engine's bf16 results are bounded by (gpu_util.bf16_gap_bound)."""
    import os
    from oracle import quant as Q, unet as U
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 2) // 2)))
    sd = {k: v.float() for k, v in w.items()}
    with torch.no_grad():
        y32 = U.unet_forward(sd, DM_CFG, x, t)
        with Q.bf16_storage(True):
            yq = U.unet_forward(sd, DM_CFG, x, t)
    return y32, rel_l2(yq, y32)


def test_c5_batch_independence_and_bf16_bound_b64_l3072(dm_net):
    """B = 64 vs B = 3 select different kernels at L = 3072 (one-pass vs split GroupNorm, attention tiling, split-K counts).  Two bf16
    evaluations whose statistics are summed in a different order differ by INDEPENDENT bf16 roundings compounded through ~50 layers
    -- at this length about half of what each differs from fp32 (measured: mutual 3.2e-2, each 6.7e-2 from the fp32 engine;
    1.5e-2 / 3e-2 at L = 768).  Yardsticks: the fp32 ORACLE on the same three windows and the gap its bf16-storage emulation opens.
    Both engine results must lie within the gap bound of the oracle, the fp32 engine must reproduce the oracle, and the two bf16
    results must be closer to each other than to fp32; a mis-addressed tile / halo / sample boundary would be an O(1) error."""
    import gpu_util as G
    from eegldm.models import UNetModel
    nb, w, x, t = dm_net
    nb.eval()
    y = nb(x, timesteps=t).float().cpu()
    assert y.shape == (B5, 1, L5) and torch.isfinite(y).all()
    nf = UNetModel(**DM_CFG, dtype="float32"); nf.load_state_dict(w); nf.eval()
    for idx in ([0, 1, 2], [61, 62, 63]):
        y32, gap = _bf16_gap_l3072(w, x[idx], t[idx])
        ys = nb(x[idx], timesteps=t[idx]).float().cpu()
        yf = nf(x[idx], timesteps=t[idx]).float().cpu()
        assert rel_l2(yf, y32) < 5e-5, rel_l2(yf, y32)                                # fp32 engine == fp32 oracle at L = 3072
        e_big, e_small, e_mutual = rel_l2(y[idx], y32), rel_l2(ys, y32), rel_l2(ys, y[idx])
        bound = G.bf16_gap_bound(gap)
        assert e_big < bound and e_small < bound, (idx, e_big, e_small, gap)
        assert e_mutual < max(e_big, e_small), (idx, e_mutual, e_big, e_small)
        print(f"C5 {idx}: bf16 B=64 {e_big:.2e}, bf16 B=3 {e_small:.2e}, mutual {e_mutual:.2e}, storage gap {gap:.2e}")
    ys = nb(x[[31, 32, 33]], timesteps=t[[31, 32, 33]]).float().cpu()
    assert rel_l2(ys, y[[31, 32, 33]]) < 5e-2
    del nf


def test_c5_backward_linear_in_dy_and_bias_checksum_b64(dm_net):
    nb, _w, x, t = dm_net
    nb.train()
    dy = torch.randn(B5, 1, L5, generator=torch.Generator().manual_seed(6))
    nb(x, timesteps=t); nb.zero_grad(); dx1 = nb.backward(dy, need_dx=True).float().cpu(); g1 = nb.flat_grad.clone()
    nb(x, timesteps=t); nb.zero_grad(); dx2 = nb.backward(2.0 * dy, need_dx=True).float().cpu(); g2 = nb.flat_grad.clone()
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert rel_l2(dx2, 2.0 * dx1) < 1e-6, rel_l2(dx2, 2.0 * dx1)
    assert rel_l2(g2, 2.0 * g1) < 2e-5, rel_l2(g2, 2.0 * g1)
    off, n, _shape = nb.entries["out.2.bias"]
    want = float(dy.bfloat16().double().sum()); got = float(g1[off:off + n].double().sum())
    assert abs(got - want) <= 2e-3 * (float((dy.double() ** 2).sum()) ** 0.5), (got, want)


def test_c5_dm_train_step_b64_bf16_vs_fp32_engine(dm_net):
    """training_diffusion.py:141-151 at the per-GPU batch of BASELINE configs[4]: epsilon MSE + 1e-6 x JukeboxLoss(sum), one step."""
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import dm_train_step
    nb, w, x, t = dm_net
    noise = torch.from_numpy(normal((B5, 1, L5), seed=7))
    out = {}
    for name, net in (("bf16", nb), ("f32", None)):
        if net is None:
            net = UNetModel(**DM_CFG, dtype="float32"); net.load_state_dict(w)
        sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
        net.zero_grad()
        loss = dm_train_step(net, sched, x, noise, t.to(net.device), spectral_weight=1e-6, spectral_loss=True)
        torch.cuda.synchronize()
        out[name] = (float(loss), net.flat_grad.clone().cpu())
    (lb, gb), (lf, gf) = out["bf16"], out["f32"]
    assert lb == lb and lf == lf and abs(lb - lf) < 2e-2 * abs(lf), (lb, lf)
    e = rel_l2(gb, gf)
    assert e < 1.2e-1, e              # whole 30.5 M-element gradient vector, bf16 storage through ~50 layers at L = 3072
    print(f"C5 B=64 dm step: loss bf16 {lb:.5f} fp32 {lf:.5f}, grad rel-L2 {e:.2e}")
