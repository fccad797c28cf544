"""Round-6 golden vectors: the REFERENCE's own low-precision run (build container only: needs /root/reference).

    python tests/golden/make_golden_autocast.py

The reference trains under `torch.autocast` (src/training/training.py:423; fp16 on its CUDA devices).  The same model code under
`torch.autocast("cpu", dtype=torch.bfloat16)` is the closest thing to "a reference bf16 run" that exists: torch's CPU autocast policy sends
conv1d / linear / matmul through bf16 and keeps GroupNorm, softmax and the loss in fp32.  For every UNet golden case this script records, next
to the fp32 vectors of make_golden.py, how far THAT run is from the reference's fp32 run: relative L2 error of y, dx and of every parameter
gradient (unet_autocast_bf16.npz).  tests/test_gpu_unet.py holds the bf16 engine to those numbers: an engine that stores activations in bf16
has to land at least as close to the fp32 result as the reference's own reduced-precision execution does.
Inputs and weights are regenerated from seeds (param_gen.py); the file holds error figures and the autocast outputs only.
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


def run(net, x0, t, dy, autocast):
    for p in net.parameters():
        p.grad = None
    x = x0.clone().requires_grad_(True)
    if autocast:
        with torch.autocast("cpu", dtype=torch.bfloat16):
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
        print(f"{name}: autocast-bf16 vs fp32: y {out[name + ':y_err']:.3e} dx {out[name + ':dx_err']:.3e} param grads median {np.median(ge):.3e} max {ge.max():.3e}")
    np.savez_compressed(os.path.join(HERE, "unet_autocast_bf16.npz"), **out)


if __name__ == "__main__":
    main()
