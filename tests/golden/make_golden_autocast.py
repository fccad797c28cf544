"""Round-6 golden vectors: the REFERENCE's own low-precision run (build container only: needs /root/reference).

    python tests/golden/make_golden_autocast.py

The reference trains under `torch.autocast` (src/training/training.py:423; fp16 on its CUDA devices).  The same model code under
`torch.autocast("cpu", dtype=torch.bfloat16)` -- and dtype=torch.float16, the reference's own type -- is the closest thing to "a reference
reduced-precision run" that exists: torch's CPU autocast policy sends
conv1d / linear / matmul through bf16 and keeps GroupNorm, softmax and the loss in fp32.  For every UNet golden case this script records, next
to the fp32 vectors of make_golden.py, how far THAT run is from the reference's fp32 run: relative L2 error of y, dx and of every parameter
gradient (unet_autocast_bf16.npz).  tests/test_gpu_unet.py holds the bf16 engine to those numbers: an engine that stores activations in bf16
has to land at least as close to the fp32 result as the reference's own reduced-precision execution does.
aekl_autocast_bf16.npz / disc_autocast_bf16.npz: the same for the production stage-1 autoencoder and the PatchGAN discriminator (the reference's
local twins: ae_kl.py with one-group Normalize, discriminator.py with kernel-3 convs).
Inputs and weights are regenerated from seeds (param_gen.py); the files hold error figures and the autocast outputs only.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")

from param_gen import gen_param, normal  # noqa: E402
from models import unet as R  # noqa: E402
from make_golden_cases import UNET_CASES, UNET_FULL  # noqa: E402

torch.set_num_threads(8)


def rel(a, b):
    a = a.double().reshape(-1); b = b.double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


DT, TAG = torch.bfloat16, "bf16"      # set by __main__: the script runs once per reduced-precision type (bf16, then fp16 -- the reference's own, training.py:423)


def run(net, x0, t, dy, autocast):
    for p in net.parameters():
        p.grad = None
    x = x0.clone().requires_grad_(True)
    if autocast:
        with torch.autocast("cpu", dtype=DT):
            y = net(x, timesteps=t)
    else:
        y = net(x, timesteps=t)
    y.float().backward(dy)
    return y.detach().float(), x.grad.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}


def main():
    out = {}
    cases = dict(UNET_CASES)
    cases["full_l768"] = UNET_FULL
    for name, (kw, B, L) in cases.items():
        g = np.load(os.path.join(HERE, f"unet_{name}.npz"))
        sw, sx, _st, sdy = [int(v) for v in g["seeds"]]
        net = R.UNetModel(**kw)
        net.load_state_dict({k: torch.from_numpy(gen_param(sw, k, v.shape)) for k, v in net.state_dict().items()})
        x = torch.from_numpy(normal((B, kw["in_channels"], L), seed=sx)); t = torch.from_numpy(g["t"])
        dy = torch.from_numpy(normal((B, kw["out_channels"], L), seed=sdy))
        y32, dx32, g32 = run(net, x, t, dy, False)
        np.testing.assert_allclose(y32.numpy(), g["y"], rtol=1e-4, atol=2e-5)          # the same fp32 run the round-1 golden holds
        ya, dxa, ga = run(net, x, t, dy, True)
        out[name + ":y_err"] = np.float64(rel(ya, y32)); out[name + ":dx_err"] = np.float64(rel(dxa, dx32))
        out[name + ":y"] = ya.numpy().astype(np.float32) if name != "full_l768" else ya.numpy()[:, :, :64].astype(np.float32)
        keys = list(g32.keys())
        out[name + ":keys"] = np.array(keys)
        out[name + ":g_err"] = np.array([rel(ga[k], g32[k]) for k in keys])
        out[name + ":g_l2"] = np.array([float(g32[k].double().norm()) for k in keys])
        ge = out[name + ":g_err"]
        print(f"{name}: autocast-{TAG} vs fp32: y {out[name + ':y_err']:.3e} dx {out[name + ':dx_err']:.3e} param grads median {np.median(ge):.3e} max {ge.max():.3e}")
    np.savez_compressed(os.path.join(HERE, f"unet_autocast_{TAG}.npz"), **out)


def aekl():
    """The production stage-1 autoencoder ([32,32,64], norm_num_groups 1: the frozen encoder of the LDM step and the decoder of sampling) as the
    reference's local twin (make_golden_r5.build_twin_g1), fp32 against bf16 autocast: recon, z_mu, z_sigma, dx, every parameter gradient of
    <recon, dy> + 0.3 KL (the quantities of aekl_twin_32_32_64_g1.npz)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from make_golden_r5 import build_twin_g1
    from param_gen import eeg_windows
    g = np.load(os.path.join(HERE, "aekl_twin_32_32_64_g1.npz"))
    sw, sx, se, sdy = [int(v) for v in g["seeds"]]; B, L = int(g["B"]), int(g["L"])
    net, ref_param, shapes, _nc = build_twin_g1(32, L, sw)
    x0 = torch.from_numpy(eeg_windows(B, seed=sx, length=L, pad=8)); eps = torch.from_numpy(normal((B, 1, L // 4), seed=se))
    dy = torch.from_numpy(normal((B, 1, L), seed=sdy))

    def run(autocast):
        for k in shapes:
            ref_param(k).grad = None
        x = x0.clone().requires_grad_(True)
        ctx = torch.autocast("cpu", dtype=DT) if autocast else torch.autocast("cpu", enabled=False)
        with ctx:
            z_mu, z_sigma = net.encode(x)
            recon = net.decode(z_mu + eps * z_sigma)
        z_mu, z_sigma, recon = z_mu.float(), z_sigma.float(), recon.float()
        kl = 0.5 * torch.sum(z_mu.pow(2) + z_sigma.pow(2) - torch.log(z_sigma.pow(2)) - 1, dim=[1])
        kl = torch.sum(kl) / kl.shape[0]
        ((recon * dy).sum() + 0.3 * kl).backward()
        return recon.detach(), z_mu.detach(), z_sigma.detach(), x.grad.clone(), {k: ref_param(k).grad.clone() for k in shapes}

    r32 = run(False); ra = run(True)
    np.testing.assert_allclose(r32[0].numpy(), g["recon"], rtol=2e-4, atol=5e-5)
    out = {"recon_err": np.float64(rel(ra[0], r32[0])), "mu_err": np.float64(rel(ra[1], r32[1])), "sigma_err": np.float64(rel(ra[2], r32[2])),
           "dx_err": np.float64(rel(ra[3], r32[3])), "keys": np.array(list(shapes.keys())),
           "g_err": np.array([rel(ra[4][k], r32[4][k]) for k in shapes]), "g_l2": np.array([float(r32[4][k].double().norm()) for k in shapes])}
    print(f"aekl [32,32,64] g1: autocast-{TAG} vs fp32:" " recon %.3e mu %.3e sigma %.3e dx %.3e grads median %.3e" %
          (out["recon_err"], out["mu_err"], out["sigma_err"], out["dx_err"], float(np.median(out["g_err"]))))
    np.savez_compressed(os.path.join(HERE, f"aekl_autocast_{TAG}.npz"), **out)


def disc():
    """The reference's local PatchGAN Discriminator with the kernel-3 convs of the config (the twin of make_golden_r2.case_disc_twin, same seeds),
    training mode, fp32 against bf16 autocast: logits, dx, every parameter gradient of <logits, dy> (disc_twin_k3.npz's quantities)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from models import discriminator as RD
    from oracle import aekl as A
    g = np.load(os.path.join(HERE, "disc_twin_k3.npz"))
    sw, sx, sdy = [int(v) for v in g["seeds"]]
    dcfg = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
    shapes = A.disc_param_shapes(dcfg)
    pkeys = [k for k in shapes if "running" not in k and "num_batches" not in k]

    def build():
        net = RD.Discriminator(input_nc=1, ndf=64, n_layers=3)
        main_ = net.main
        for i in (0, 2, 5, 8, 11):
            c = main_[i]
            main_[i] = torch.nn.Conv1d(c.in_channels, c.out_channels, kernel_size=3, stride=c.stride, padding=1, bias=c.bias is not None)
        def ref_tensor(key):
            parts = key.split(".")
            if parts[0] == "initial_conv":
                return getattr(main_[0], parts[-1])
            if parts[0] == "final_conv":
                return getattr(main_[11], parts[-1])
            l = int(parts[0])
            return getattr(main_[2 + 3 * l], parts[-1]) if parts[1] == "conv" else getattr(main_[3 + 3 * l], parts[-1])
        with torch.no_grad():
            for k, shp in shapes.items():
                v = torch.from_numpy(gen_param(sw, k, shp))
                ref_tensor(k).copy_(v * 2.0 if k.endswith("conv.weight") else v)
        net.train()
        return net, ref_tensor

    x0 = torch.from_numpy(normal((3, 1, 256), seed=sx))

    def run(autocast):
        net, ref_tensor = build()                       # fresh running statistics for each run
        x = x0.clone().requires_grad_(True)
        with (torch.autocast("cpu", dtype=DT) if autocast else torch.autocast("cpu", enabled=False)):
            logits = net(x)
        logits = logits.float()
        dy = torch.from_numpy(normal(tuple(logits.shape), seed=sdy))
        (logits * dy).sum().backward()
        return logits.detach(), x.grad.clone(), {k: ref_tensor(k).grad.clone() for k in pkeys}

    r32 = run(False); ra = run(True)
    np.testing.assert_allclose(r32[0].numpy(), g["logits"], rtol=2e-4, atol=5e-5)
    out = {"logits_err": np.float64(rel(ra[0], r32[0])), "dx_err": np.float64(rel(ra[1], r32[1])), "keys": np.array(pkeys),
           "g_err": np.array([rel(ra[2][k], r32[2][k]) for k in pkeys]), "g_l2": np.array([float(r32[2][k].double().norm()) for k in pkeys])}
    print(f"discriminator twin: autocast-{TAG} vs fp32:" " logits %.3e dx %.3e grads median %.3e" % (out["logits_err"], out["dx_err"], float(np.median(out["g_err"]))))
    np.savez_compressed(os.path.join(HERE, f"disc_autocast_{TAG}.npz"), **out)


if __name__ == "__main__":
    for DT, TAG in ((torch.bfloat16, "bf16"), (torch.float16, "f16")):
        if "--f16-only" in sys.argv and TAG != "f16":
            continue
        main()
        aekl()
        disc()
