"""-m gpu: the few-row forward conv / linear kernel (csrc/conv_skinny.hip) that serves the UNet when ONE window is sampled per call
(sample_trials.py:149-163).  Every k3 / 1x1 layer shape of the config_ldm UNet at B = 1 (and ragged cases whose row count is
not a multiple of the 16/32-row tiles), bias + time-embedding row + residual, against torch's fp32 conv on the bf16-rounded
operands; the three register tilings (EEGLDM_CONV_SKINNY_TILE) and the general kernel (EEGLDM_NO_CONV_SKINNY=1) must agree with
each other to accumulation order.  The switches are toggled IN PROCESS (conftest.env_switches -> eegldm_debug_reload_env).

The chain built on that kernel -- GroupNorm applied on the consuming conv's operand load (conv_skinny<GN>, NetBase::eval_fuse), the per-run
embedding table, DDIM-50 of a couple of windows -- is held against the CPU ORACLE (oracle.unet.unet_forward / oracle.steps.ddim_sample, which
restate unet.py:512-563 and sample_trials.py:149-170) under the bf16-storage bound of tests/gpu_util.py, next to the engine-vs-engine A/Bs."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import normal  # noqa: E402

CASES = [  # B, L, Cin, Cout, K, rowvec, resid
    (1, 768, 128, 128, 3, 1, 0), (1, 768, 128, 128, 3, 0, 1), (1, 384, 256, 256, 3, 1, 0), (1, 192, 512, 512, 3, 0, 1),
    (1, 192, 1024, 512, 3, 1, 0), (1, 192, 768, 512, 3, 1, 0), (1, 384, 768, 256, 3, 1, 0), (1, 384, 512, 256, 3, 1, 0),
    (1, 768, 384, 128, 3, 1, 0), (1, 768, 256, 128, 3, 1, 0), (1, 384, 128, 256, 3, 1, 0), (1, 192, 256, 512, 3, 1, 0),
    (1, 192, 512, 1536, 1, 0, 0), (1, 192, 512, 512, 1, 0, 1), (1, 192, 1024, 512, 1, 0, 0), (1, 768, 384, 128, 1, 0, 0),
    (2, 72, 96, 48, 3, 1, 1), (3, 40, 160, 48, 1, 0, 1), (2, 24, 32, 16, 3, 0, 0), (5, 192, 512, 512, 3, 1, 1),
]
UCFG = dict(in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
            resblock_updown=True)
ACFG = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)


def _run_cases():
    """every case through eegldm_conv1d_fwd (bf16), each checked against torch's fp32 conv; returns the outputs"""
    import gpu_util as G
    c = G.ctx(); dt = G.BF16; out = []
    for ci, (B, L, Cin, Cout, K, rv, rs) in enumerate(CASES):
        x = torch.from_numpy(normal((B, Cin, L), seed=10 + ci)).bfloat16().float()
        w = (torch.from_numpy(normal((Cout, Cin, K), seed=40 + ci)) / math.sqrt(Cin * K)).bfloat16().float()
        b = torch.from_numpy(normal((Cout,), seed=70 + ci))
        e = torch.from_numpy(normal((B, Cout), seed=100 + ci)) if rv else None
        r = torch.from_numpy(normal((B, Cout, L), seed=130 + ci)).bfloat16().float() if rs else None
        ref = F.conv1d(x, w, b, padding=K // 2)
        if rv: ref = ref + e[:, :, None]
        if rs: ref = ref + r
        xd, wd, bd = G.nlc(x, dt), G.pack_w(w, dt), b.to(G.DEV)
        ed = e.to(G.DEV) if rv else None; rd = G.nlc(r, dt) if rs else None
        yd = torch.full((B * L, Cout), float("nan"), device=G.DEV, dtype=torch.bfloat16)
        G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yd), Cout, B, L, Cin, Cout, K, 1, K // 2, K // 2,
                                        G.ptr(ed) if rv else None, Cout if rv else 0, G.ptr(rd) if rs else None, Cout if rs else 0, dt))
        y = G.ncl(yd, B, L).float().cpu()
        assert torch.isfinite(y).all(), CASES[ci]
        G.assert_close(y, ref, **G.TOL[dt], name="case %d %s" % (ci, CASES[ci]))
        out.append(y.numpy())
        if rs:   # in place: the residual IS the output buffer (net.hip: out = conv2(h) + out)
            yi = rd.clone()
            G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yi), Cout, B, L, Cin, Cout, K, 1, K // 2, K // 2,
                                            G.ptr(ed) if rv else None, Cout if rv else 0, G.ptr(yi), Cout, dt))
            assert torch.equal(yi.view(torch.int16), yd.view(torch.int16)), ("in-place residual", CASES[ci])
    return out


def test_skinny_conv_matches_fp32_reference_and_general_kernel(env_switches):
    ref = _run_cases()
    for name, env in [("tile11", {"EEGLDM_CONV_SKINNY_TILE": "11"}), ("tile21", {"EEGLDM_CONV_SKINNY_TILE": "21"}),
                      ("tile22", {"EEGLDM_CONV_SKINNY_TILE": "22"}), ("general_kernel", {"EEGLDM_CONV_SKINNY_TILE": None, "EEGLDM_NO_CONV_SKINNY": "1"})]:
        env_switches(**env)
        v = _run_cases()                       # each run already checked itself against the fp32 reference
        for k in range(len(ref)):
            d = float(np.linalg.norm(v[k] - ref[k])) / (float(np.linalg.norm(ref[k])) + 1e-12)
            assert d < 4e-3, (name, CASES[k], d)      # same products, other summation order: bf16 rounding flips only


def _seeded_unet(L, dtype, seed=0):
    from eegldm.models import UNetModel
    torch.manual_seed(seed)                 # the module's default init draws from torch's global generator
    net = UNetModel(image_size=L, **UCFG, dtype=dtype)
    g = torch.Generator().manual_seed(seed); sd = net.state_dict()
    net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()})
    return net, {k: v.detach().float().cpu().clone() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("L,batches", [(768, (1, 2, 3)), (3072, (1,))])      # latent model; pixel-space model (192 statistics slots per group: the loop path)
def test_eval_forward_with_fused_groupnorm_matches_the_oracle_and_the_train_mode_forward(L, batches):
    """Eval-mode forward of a few windows (NetBase::eval_fuse, net.hip): conv1 of every ResBlock leaves the statistics of its output and
    conv2 applies GroupNorm + SiLU on its operand load (conv_skinny<..., GN = true>), so 22 GroupNorm launches per forward disappear.
    ORACLE leg: the result against oracle.unet.unet_forward (unet.py:512-563 restated) on the same weights / inputs, bounded by the
    error bf16 storage alone causes in that oracle (oracle/quant.py, gpu_util.bf16_gap_bound) -- the same bound the unfused bf16 forward meets.
    Extras: the train-mode forward of the same network (stand-alone GroupNorm kernels) and the fp32 engine; the eval forward refuses a backward."""
    import gpu_util as G
    from eegldm.models import UNetModel
    from oracle import quant as Q
    from oracle import unet as U
    net, w = _seeded_unet(L, "bfloat16")
    nf = UNetModel(image_size=L, **UCFG, dtype="float32"); nf.load_state_dict(w); nf.eval()
    g = torch.Generator().manual_seed(1)
    rel = lambda a, b: float((a - b).norm() / b.norm())
    for B in batches:
        x = torch.randn(B, 1, L, generator=g); t = torch.randint(0, 1000, (B,), generator=g)
        with torch.no_grad():
            y32 = U.unet_forward(w, UCFG, x, t)                              # the fp32 oracle
            with Q.bf16_storage(True):
                yq = U.unet_forward(w, UCFG, x, t)                           # ... and with bf16 storage emulated: the yardstick
        gap = rel(yq, y32)
        yf = nf(x, timesteps=t).float().cpu().clone()                        # fp32 engine
        net.train(); yt = net(x, timesteps=t).float().cpu().clone()
        net.eval(); ye = net(x, timesteps=t).float().cpu().clone()
        assert torch.isfinite(ye).all()
        print(f"L={L} B={B}: oracle bf16-storage gap {gap:.2e}; eval(fused) vs oracle {rel(ye, y32):.2e}, train(unfused) vs oracle {rel(yt, y32):.2e}, "
              f"fp32 engine vs oracle {rel(yf, y32):.2e}")
        assert rel(yf, y32) < 2e-4, (B, rel(yf, y32))                        # fp32 engine == oracle (sums reordered)
        assert rel(ye, y32) < G.bf16_gap_bound(gap), (B, rel(ye, y32), gap)  # fused eval forward within the storage bound of the ORACLE
        assert rel(yt, y32) < G.bf16_gap_bound(gap), (B, rel(yt, y32), gap)
        # engine vs engine: the two bf16 forwards differ by rounding flips compounded over ~50 layers
        assert rel(ye, yt) < 1.5e-2, (B, rel(ye, yt))
        ye2 = net(x, timesteps=t).float().cpu()
        assert torch.equal(ye, ye2), "eval forward not reproducible"     # fp64 atomics of ~12 partials per group: order-independent to fp32
    with pytest.raises(RuntimeError):
        net.backward(torch.zeros(x.shape[0], 1, L))
    net.train(); net(x, timesteps=t); net.zero_grad(); net.backward(torch.zeros(x.shape[0], 1, L))       # a train-mode forward restores the tape


def _ddim_run(dtype, w_unet, w_ae, noise):
    from eegldm.models import UNetModel, AutoencoderKL
    from eegldm.sampling import ddim_sample, make_sampling_scheduler
    u = UNetModel(image_size=768, **UCFG, dtype=dtype); u.load_state_dict(w_unet)
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2, norm_num_groups=1,
                       attention_levels=[False] * 3, dtype=dtype)
    ae.load_state_dict(w_ae)
    x, z = ddim_sample(u, ae, make_sampling_scheduler(50), noise)
    x, z = x.float().cpu().numpy(), z.float().cpu().numpy()
    assert np.isfinite(x).all() and np.isfinite(z).all(), dtype
    return x, z


def test_ddim50_of_two_windows_few_row_chain_against_the_oracle_general_kernels_and_fp32(env_switches):
    """The whole one-window chain (few-row convs, GroupNorm folded into them, per-run embedding table, caller-stream launches) over a full
    DDIM-50 run + decode (sample_trials.py:149-170): 50 UNet forwards feed each other, so this is where a small systematic error of the
    fused forward would grow.  ORACLE leg: oracle.steps.ddim_sample on the same weights and noise -- final latents and decoded windows of the
    fp32 engine within 2e-3, of the bf16 few-row chain within the bf16-storage bound measured on the oracle itself (the same 50-step run with
    storage emulated).  Extras: few-row chain vs every few-row switch off (measured 5e-4), both vs the fp32 engine (8e-4)."""
    import eegldm
    import gpu_util as G
    from eegldm.models import AutoencoderKL
    from eegldm.training import randn
    from oracle import losses as Ls
    from oracle import quant as Q
    from oracle import steps as S
    _net, w_unet = _seeded_unet(768, "float32", seed=42)
    del _net
    torch.manual_seed(3)
    ae0 = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2, norm_num_groups=1,
                        attention_levels=[False] * 3, dtype="float32")
    w_ae = {k: v.detach().float().cpu().clone() for k, v in ae0.state_dict().items()}
    del ae0
    noise = randn(eegldm.default_context(0), (2, 1, 768), seed=4242)
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0205)
    xo, zo = S.ddim_sample(w_unet, UCFG, w_ae, ACFG, noise.cpu(), 50, acp)
    with Q.bf16_storage(True):
        xq, zq = S.ddim_sample(w_unet, UCFG, w_ae, ACFG, noise.cpu(), 50, acp)
    xo, zo, xq, zq = xo.numpy(), zo.numpy(), xq.numpy(), zq.numpy()
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))
    gap_z, gap_x = rel(zq, zo), rel(xq, xo)
    x, z = {}, {}
    x["fp32"], z["fp32"] = _ddim_run("float32", w_unet, w_ae, noise)
    x["few_row"], z["few_row"] = _ddim_run("bfloat16", w_unet, w_ae, noise)
    env_switches(EEGLDM_NO_CONV_SKINNY="1", EEGLDM_NO_EVAL_GN_FUSE="1", EEGLDM_SAMPLE_NO_EMB_TABLE="1", EEGLDM_GN_NO_FEW_SLAB_NARROW="1",
                 EEGLDM_SAMPLE_OWN_STREAM="1")
    x["general"], z["general"] = _ddim_run("bfloat16", w_unet, w_ae, noise)
    print(f"DDIM-50 vs ORACLE: fp32 engine latents {rel(z['fp32'], zo):.2e} windows {rel(x['fp32'], xo):.2e}; bf16 few-row latents {rel(z['few_row'], zo):.2e} "
          f"windows {rel(x['few_row'], xo):.2e}; bf16 general latents {rel(z['general'], zo):.2e}; oracle bf16-storage gap latents {gap_z:.2e} windows {gap_x:.2e}")
    print(f"DDIM-50 engine A/B: few-row vs general {rel(z['few_row'], z['general']):.2e}, few-row vs fp32 {rel(z['few_row'], z['fp32']):.2e}, "
          f"general vs fp32 {rel(z['general'], z['fp32']):.2e}; decoded windows few-row vs general {rel(x['few_row'], x['general']):.2e}")
    assert rel(z["fp32"], zo) < 2e-3 and rel(x["fp32"], xo) < 2e-3
    for name in ("few_row", "general"):
        assert rel(z[name], zo) < G.bf16_gap_bound(gap_z), (name, rel(z[name], zo), gap_z)
        assert rel(x[name], xo) < G.bf16_gap_bound(gap_x), (name, rel(x[name], xo), gap_x)
    assert rel(z["few_row"], z["general"]) < 5e-3 and rel(z["few_row"], z["fp32"]) < 5e-3 and rel(z["general"], z["fp32"]) < 5e-3
    assert rel(x["few_row"], x["general"]) < 2e-2


def test_skip_connection_inside_the_second_convs_few_row_launch(env_switches):
    """Round 5: at one window per call the ResBlock tail skip_connection(x) + conv2(h) (unet.py:302,327) is ONE few-row launch -- the 1 x 1 conv
    runs as a second reduction of conv2's kernel (conv_skinny.hip K extension) -- instead of two: 11 launches fewer per forward, the same
    output up to the one bf16 rounding of the intermediate that the fusion removes."""
    import csv, os, tempfile
    from eegldm._lib import lib, check
    net, _w = _seeded_unet(768, "bfloat16", seed=42)
    net.eval()
    x = torch.from_numpy(normal((1, 1, 768), seed=31)).cuda(); t = torch.full((1,), 431, dtype=torch.int64).cuda()
    outs = {}
    for name, sw in (("fused", None), ("two", "1")):
        env_switches(EEGLDM_NO_FUSED_SKIP=sw)
        net(x, timesteps=t)
        net.ctx.prof_enable(True)
        y = net(x, timesteps=t)
        path = os.path.join(tempfile.gettempdir(), "eegldm_sk_rows.csv")
        check(lib.eegldm_prof_dump(net.ctx.h, path.encode()))
        net.ctx.prof_enable(False)
        with open(path) as fh:
            n = len(list(csv.DictReader(fh)))
        outs[name] = (y.clone(), n)
    assert outs["two"][1] - outs["fused"][1] == 11, (outs["fused"][1], outs["two"][1])       # one per ResBlock with a skip_connection
    ya, yb = outs["fused"][0], outs["two"][0]
    assert float((ya - yb).abs().max()) <= 2e-2 * float(yb.abs().max()), float((ya - yb).abs().max()) / float(yb.abs().max())
