"""-m gpu: the few-row forward conv / linear kernel (csrc/conv_skinny.hip) that serves the UNet when ONE window is sampled per call
(sample_trials.py:149-163).  Every k3 / 1x1 layer shape of the config_ldm UNet at B = 1 (and a ragged B = 2 case whose row count is
not a multiple of the 16/32-row tiles), bias + time-embedding row + residual, against torch's fp32 conv on the bf16-rounded
operands; the three register tilings (EEGLDM_CONV_SKINNY_TILE) and the general kernel (EEGLDM_NO_CONV_SKINNY=1) must agree with
each other to accumulation order (each in a subprocess: the switches are read once per process)."""
import math
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, math, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, %r); sys.path.insert(0, %r)
import gpu_util as G
from param_gen import normal
CASES = [  # B, L, Cin, Cout, K, rowvec, resid
    (1, 768, 128, 128, 3, 1, 0), (1, 768, 128, 128, 3, 0, 1), (1, 384, 256, 256, 3, 1, 0), (1, 192, 512, 512, 3, 0, 1),
    (1, 192, 1024, 512, 3, 1, 0), (1, 192, 768, 512, 3, 1, 0), (1, 384, 768, 256, 3, 1, 0), (1, 384, 512, 256, 3, 1, 0),
    (1, 768, 384, 128, 3, 1, 0), (1, 768, 256, 128, 3, 1, 0), (1, 384, 128, 256, 3, 1, 0), (1, 192, 256, 512, 3, 1, 0),
    (1, 192, 512, 1536, 1, 0, 0), (1, 192, 512, 512, 1, 0, 1), (1, 192, 1024, 512, 1, 0, 0), (1, 768, 384, 128, 1, 0, 0),
    (2, 72, 96, 48, 3, 1, 1), (3, 40, 160, 48, 1, 0, 1), (2, 24, 32, 16, 3, 0, 0), (5, 192, 512, 512, 3, 1, 1),
]
c = G.ctx(); dt = G.BF16; out = {}
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
    G.assert_close(y, ref, **G.TOL[dt], name="case %%d %%s" %% (ci, CASES[ci]))
    out["y%%d" %% ci] = y.numpy()
    if rs:   # in place: the residual IS the output buffer (net.hip: out = conv2(h) + out)
        yi = rd.clone()
        G.check(G.lib.eegldm_conv1d_fwd(c.h, G.ptr(xd), Cin, G.ptr(wd), G.ptr(bd), G.ptr(yi), Cout, B, L, Cin, Cout, K, 1, K // 2, K // 2,
                                        G.ptr(ed) if rv else None, Cout if rv else 0, G.ptr(yi), Cout, dt))
        assert torch.equal(yi.view(torch.int16), yd.view(torch.int16)), ("in-place residual", CASES[ci])
np.savez(sys.argv[1], **out)
print("ok")
''' % (ROOT, os.path.join(ROOT, "tests"))


def _run(tmp_path, name, env_extra):
    out = tmp_path / (name + ".npz")
    env = dict(os.environ); env.update(env_extra)
    env["PYTHONPATH"] = os.path.join(ROOT, "tests", "golden") + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", SCRIPT, str(out)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (name, r.stdout[-2000:], r.stderr[-4000:])
    return np.load(out)


def test_skinny_conv_matches_fp32_reference_and_general_kernel(tmp_path):
    ref = _run(tmp_path, "default", {})
    for name, env in [("tile11", {"EEGLDM_CONV_SKINNY_TILE": "11"}), ("tile21", {"EEGLDM_CONV_SKINNY_TILE": "21"}),
                      ("tile22", {"EEGLDM_CONV_SKINNY_TILE": "22"}), ("general_kernel", {"EEGLDM_NO_CONV_SKINNY": "1"})]:
        v = _run(tmp_path, name, env)          # each run already checked itself against the fp32 reference
        for k in ref.files:
            d = float(np.linalg.norm(v[k] - ref[k])) / (float(np.linalg.norm(ref[k])) + 1e-12)
            assert d < 4e-3, (name, k, d)      # same products, other summation order: bf16 rounding flips only


def test_small_shape_primitives_still_pass_on_the_general_kernel():
    """The small conv cases of test_gpu_primitives.py (tile edges inside samples, K tails, partial column tiles) take the few-row
    kernel by default; run them once more with it switched off so the general kernel's edge handling stays covered."""
    env = dict(os.environ); env["EEGLDM_NO_CONV_SKINNY"] = "1"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_primitives.py"), "-x", "-q", "-m", "gpu",
                        "-k", "conv1d or linear", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize("L,batches", [(768, (1, 2, 3)), (3072, (1,))])      # latent model; pixel-space model (192 statistics slots per group: the loop path)
def test_eval_forward_with_fused_groupnorm_matches_train_mode_forward(L, batches):
    """Eval-mode forward of a few windows (NetBase::eval_fuse, net.hip): conv1 of every ResBlock leaves the statistics of its output and
    conv2 applies GroupNorm + SiLU on its operand load, so 22 GroupNorm launches per forward disappear.  The train-mode forward of the
    same network runs the stand-alone GroupNorm kernels: both must agree to bf16 rounding, and the eval forward must refuse a backward."""
    import torch
    from eegldm.models import UNetModel
    cfg = dict(image_size=L, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
               channel_mult=[1, 2, 4], resblock_updown=True)
    net = UNetModel(**cfg, dtype="bfloat16")
    g = torch.Generator().manual_seed(0); sd = net.state_dict()
    net.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sd.items()})
    w = {k: v.cpu() for k, v in net.state_dict().items()}
    nf = UNetModel(**cfg, dtype="float32"); nf.load_state_dict(w); nf.eval()
    for B in batches:
        x = torch.randn(B, 1, L, generator=g); t = torch.randint(0, 1000, (B,), generator=g)
        yf = nf(x, timesteps=t).float().cpu().clone()                     # fp32 engine: the common yardstick
        net.train(); yt = net(x, timesteps=t).float().cpu().clone()
        net.eval(); ye = net(x, timesteps=t).float().cpu().clone()
        assert torch.isfinite(ye).all()
        rel = lambda a, b: float((a - b).norm() / b.norm())
        # the two bf16 forwards differ by rounding flips compounded over ~50 layers (the same class as batch 256 vs batch 3 in
        # test_gpu_fullsize.py: < 1.5e-2); against the fp32 engine the fused forward must be no worse than the unfused one
        assert rel(ye, yt) < 1.5e-2, (B, rel(ye, yt))
        assert rel(ye, yf) < 4e-2 and rel(ye, yf) < 1.25 * rel(yt, yf) + 2e-3, (B, rel(ye, yf), rel(yt, yf))
        ye2 = net(x, timesteps=t).float().cpu()
        assert torch.equal(ye, ye2), "eval forward not reproducible"     # fp64 atomics of ~12 partials per group: order-independent to fp32
    with pytest.raises(RuntimeError):
        net.backward(torch.zeros(x.shape[0], 1, L))
    net.train(); net(x, timesteps=t); net.zero_grad(); net.backward(torch.zeros(x.shape[0], 1, L))       # a train-mode forward restores the tape


DDIM_SCRIPT = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
import eegldm
from eegldm.models import UNetModel, AutoencoderKL
from eegldm.training import randn
from eegldm.sampling import ddim_sample, make_sampling_scheduler
dt = sys.argv[2]
ctx = eegldm.default_context(0)
torch.manual_seed(0)      # the module's default init draws from torch's global generator: same weights in every process
u = UNetModel(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4], channel_mult=[1, 2, 4],
              resblock_updown=True, dtype=dt)
g = torch.Generator().manual_seed(42); sd = u.state_dict()
u.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2, norm_num_groups=1,
                   attention_levels=[False] * 3, dtype=dt)
x, z = ddim_sample(u, ae, make_sampling_scheduler(50), randn(ctx, (2, 1, 768), seed=4242))
np.savez(sys.argv[1], x=x.float().cpu().numpy(), z=z.float().cpu().numpy())
''' % ROOT


def test_ddim50_of_two_windows_few_row_chain_against_general_kernels_and_fp32(tmp_path):
    """The whole one-window chain (few-row convs, GroupNorm folded into them, per-run embedding table, caller-stream launches) over a full
    DDIM-50 run + decode: 50 UNet forwards feed each other, so this is where a small systematic error of the fused forward would grow.
    Same seeded weights and noise in three processes: fp32 engine, bf16 with the few-row chain (default), bf16 with every few-row switch off.
    Measured (tools/debug/ddim_ab.py): final latents few-row vs general 5.0e-4, both 8e-4 from the fp32 engine."""
    off = {"EEGLDM_NO_CONV_SKINNY": "1", "EEGLDM_NO_EVAL_GN_FUSE": "1", "EEGLDM_SAMPLE_NO_EMB_TABLE": "1", "EEGLDM_GN_NO_FEW_SLAB_NARROW": "1",
           "EEGLDM_SAMPLE_OWN_STREAM": "1"}
    z, x = {}, {}
    for name, env_extra, dt in [("fp32", {}, "float32"), ("few_row", {}, "bfloat16"), ("general", off, "bfloat16")]:
        out = tmp_path / (name + ".npz")
        env = dict(os.environ); env.update(env_extra)
        r = subprocess.run([sys.executable, "-c", DDIM_SCRIPT, str(out), dt], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-3000:])
        d = np.load(out)
        assert np.isfinite(d["x"]).all() and np.isfinite(d["z"]).all(), name
        z[name] = d["z"]; x[name] = d["x"]
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))
    print(f"DDIM-50 final latents: few-row vs general {rel(z['few_row'], z['general']):.2e}, few-row vs fp32 {rel(z['few_row'], z['fp32']):.2e}, "
          f"general vs fp32 {rel(z['general'], z['fp32']):.2e}; decoded windows few-row vs general {rel(x['few_row'], x['general']):.2e}")
    assert rel(z["few_row"], z["general"]) < 5e-3 and rel(z["few_row"], z["fp32"]) < 5e-3 and rel(z["general"], z["fp32"]) < 5e-3
    assert rel(x["few_row"], x["general"]) < 2e-2
