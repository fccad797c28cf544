"""-m gpu: BASELINE.json's full sizes (config_ldm.yaml UNet, per-GPU batch 256, latent length 768, bf16) checked through
properties that do not need an oracle run of that size:

* batch independence -- every op of the path is per-sample (GroupNorm, single-head attention) or per-row (convs, linears), so
  sample i of a 256-batch must equal the same sample run in a batch of 3 (tile edges, XCD tile order, 192-row tiles, halo masks
  and whole-sample attention blocks all change between the two launches);
* bf16 engine vs fp32 engine on the same weights and inputs;
* the backward is linear in dy (scaling by 2 is exact in floating point: only the order of the split-K atomics may differ);
* checksum of the bias gradient of the last conv: d(out.2.bias) = sum of dy;
* DDIM sampling of a sharded seed range equals the un-sharded run (sample_trials.py:149-151; distributed.shard_range).
The small-shape parity against the reference's own golden vectors is tests/test_gpu_unet.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
           channel_mult=[1, 2, 4], resblock_updown=True)      # config_ldm.yaml:30-43
B, L = 256, 768


def _weights(net, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = net.state_dict()
    # module default init, then N(0, 0.02) into the zero-initialised layers so every path carries signal
    return {k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()}


def rel_l2(a, b):
    a = a.detach().double().cpu().reshape(-1); b = b.detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


@pytest.fixture(scope="module")
def nets():
    from eegldm.models import UNetModel
    nb = UNetModel(**CFG, dtype="bfloat16"); w = _weights(nb); nb.load_state_dict(w)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 1, L, generator=g); t = torch.randint(0, 1000, (B,), generator=g)
    return nb, w, x, t


def test_batch_independence_full_size(nets):
    nb, _w, x, t = nets
    nb.eval()
    y = nb(x, timesteps=t).float().cpu()
    assert y.shape == (B, 1, L) and torch.isfinite(y).all()
    for idx in ([0, 1, 2], [127, 128, 129], [253, 254, 255]):
        ys = nb(x[idx], timesteps=t[idx]).float().cpu()
        # the two batch sizes select different kernels (whole-sample vs 64-row attention blocks, one-pass vs split GroupNorm, tile
        # shapes): results agree to bf16 rounding compounded over ~50 layers, a mis-addressed tile would be O(1)
        assert rel_l2(ys, y[idx]) < 1.5e-2, (idx, rel_l2(ys, y[idx]))


def test_forward_and_input_gradient_are_bit_reproducible_full_size(nets):
    """Two identical runs: outputs and dx bitwise equal (seeded sampling is reproducible), parameter gradients to fp32 atomics order."""
    nb, _w, x, t = nets
    nb.train()
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(B, 1, L, generator=g)
    outs = []
    for _ in range(2):
        y = nb(x, timesteps=t).clone(); nb.zero_grad(); dx = nb.backward(dy, need_dx=True).clone(); outs.append((y, dx, nb.flat_grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert rel_l2(outs[1][2], outs[0][2]) < 1e-5


def test_bf16_engine_tracks_fp32_engine_full_size(nets):
    from eegldm.models import UNetModel
    nb, w, x, t = nets
    nf = UNetModel(**CFG, dtype="float32"); nf.load_state_dict(w); nf.eval(); nb.eval()
    sub = slice(0, 32)                                        # fp32 parity engine: 32 windows are enough to see every tile class
    yf = nf(x[sub], timesteps=t[sub]); yb = nb(x[sub], timesteps=t[sub])
    assert rel_l2(yb, yf) < 4e-2, rel_l2(yb, yf)
    del nf


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_backward_linear_in_dy_and_bias_checksum_full_size(nets, dtype):
    """Scaling dy by 2 is exact in floating point and the activation path (forward, dx) is bit-reproducible -- GroupNorm
    statistics are accumulated in fp64 so that the arrival order of waves cannot show -- so dx must scale exactly; parameter
    gradients may differ by the summation order of the split-K / slot fp32 atomics (~2e-7, tools/debug/det_check.py)."""
    from eegldm.models import UNetModel
    nb, w, x, t = nets
    f32 = dtype == "float32"
    net = nb
    if f32:
        net = UNetModel(**CFG, dtype="float32"); net.load_state_dict(w)
    net.train()
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(B, 1, L, generator=g)
    net(x, timesteps=t); net.zero_grad(); dx1 = net.backward(dy, need_dx=True).float().cpu(); g1 = net.flat_grad.clone()
    net(x, timesteps=t); net.zero_grad(); dx2 = net.backward(2.0 * dy, need_dx=True).float().cpu(); g2 = net.flat_grad.clone()
    assert torch.isfinite(g1).all() and float(g1.abs().max()) > 0
    assert rel_l2(dx2, 2.0 * dx1) < 1e-6, rel_l2(dx2, 2.0 * dx1)
    assert rel_l2(g2, 2.0 * g1) < 2e-5, rel_l2(g2, 2.0 * g1)
    off, n, _shape = net.entries["out.2.bias"]
    # the last conv's bias gradient is the plain sum of dy (as the engine sees it: rounded to its storage type on entry)
    want = float((dy if f32 else dy.bfloat16()).double().sum())
    got = float(g1[off:off + n].double().sum())
    assert abs(got - want) <= (1e-4 if f32 else 2e-3) * (float((dy.double() ** 2).sum()) ** 0.5), (got, want)


def test_ddim_sharded_seeds_equal_unsharded_full_size(nets):
    from eegldm.models import AutoencoderKL
    from eegldm.schedulers import DDIMScheduler
    from eegldm.sampling import sample_seeds
    from eegldm.distributed import shard_range
    nb, _w, _x, _t = nets
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype="bfloat16", device=0)
    sched = DDIMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0205, device=0)
    sched.set_timesteps(5)                                     # the property does not depend on the step count
    seeds = list(range(40, 40 + 64))
    full, _ = sample_seeds(nb, ae, sched, seeds, latent_len=L)
    assert full.shape == (64, 1, 3000) and torch.isfinite(full).all()
    parts = []
    for r in range(4):
        lo, hi = shard_range(len(seeds), r, 4)
        w, _ = sample_seeds(nb, ae, sched, seeds[lo:hi], latent_len=L)
        parts.append(w)
    assert rel_l2(torch.cat(parts), full) < 8e-2           # different kernels per batch size, bf16 roundings through 5 UNet calls + decoder (3e-2 measured)
