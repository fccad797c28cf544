"""Generates tests/golden/aekl_traj_c1.json: the loss trajectory of 40 optimiser steps of the AutoencoderKL [32,32,64] +
PatchDiscriminator GAN training (train_autoencoderkl.py:203-234; reference loss weights adv 0.01, kl 1e-9, spectral 1e4; Adam 1e-3 /
5e-4) computed by the CPU oracle (oracle/steps.py::aekl_train_step, fp32 torch autograd) on seeded parameters, windows and
posterior noise.  tests/test_gpu_zz_convergence.py replays the same 40 steps through the HIP engines.  ~1 minute on 16 cores.

    python tests/golden/make_aekl_traj.py          # [32,32,64] -> aekl_traj_c1.json
    python tests/golden/make_aekl_traj.py thin     # [2,2,4] (the whole-network aekl_thin kernels, Adam 5e-3 / 5e-4) -> aekl_traj_thin.json
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)
from param_gen import gen_param, eeg_windows, normal      # noqa: E402
import oracle.aekl as A                                   # noqa: E402
import oracle.steps as S                                  # noqa: E402

ACFG = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
DCFG = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
STEPS, B, POOL = 40, 8, 64


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    thin = len(sys.argv) > 1 and sys.argv[1] == "thin"
    ACFG = dict(globals()["ACFG"], num_channels=[2, 2, 4]) if thin else globals()["ACFG"]
    lr_g = 5e-3 if thin else 1e-3
    st = {"ae": {k: torch.from_numpy(gen_param(42, k, s)) for k, s in A.aekl_param_shapes(ACFG).items()},
          "d": {k: torch.from_numpy(gen_param(43, k, s)) for k, s in A.disc_param_shapes(DCFG).items()}, "og": {}, "od": {}}
    xs = torch.from_numpy(eeg_windows(POOL, seed=777))
    traj = []
    for i in range(1, STEPS + 1):
        s = ((i - 1) * B) % POOL
        ew = torch.from_numpy(normal((B, 1, 768), seed=100 + i))
        l, st["ae"], st["d"], _r, _g, _d = S.aekl_train_step(st["ae"], ACFG, st["d"], DCFG, xs[s:s + B], ew, 0.01, 1e-9, 1e4, True, lr_g, 5e-4, i, st["og"], st["od"])
        traj.append({k: float(v) for k, v in l.items()})
        print(i, traj[-1], flush=True)
    with open(os.path.join(HERE, "aekl_traj_thin.json" if thin else "aekl_traj_c1.json"), "w") as fh:
        json.dump({"steps": STEPS, "batch": B, "pool": POOL, "window_seed": 777, "eps_seed_base": 100, "param_seeds": [42, 43],
                   "num_channels": ACFG["num_channels"], "weights": {"adv": 0.01, "kl": 1e-9, "spectral": 1e4}, "lr": [lr_g, 5e-4], "losses": traj}, fh, indent=0)


if __name__ == "__main__":
    main()
