"""Generates tests/golden/dm_traj_c5.json: the loss trajectory of 12 optimiser steps of the pixel-space diffusion model
(training_diffusion.py:141-151: config_dm.yaml UNet directly on (B,1,3072) windows, T = 768 attention, epsilon MSE + 1e-6 x
JukeboxLoss(sum), Adam 1e-4) computed by the CPU oracle (oracle/steps.py::dm_train_step + adam_update).  ~30 s on 16 cores.

    python tests/golden/make_dm_traj.py
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, HERE)
from param_gen import gen_param, eeg_windows, normal, timesteps      # noqa: E402
import oracle.losses as Ls                                            # noqa: E402
import oracle.steps as S                                              # noqa: E402
import oracle.unet as U                                               # noqa: E402

CFG = dict(image_size=3072, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
           channel_mult=[1, 2, 4], resblock_updown=True)
STEPS, B, POOL = 12, 2, 16


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(gen_param(42, k, s)) for k, s in U.unet_param_shapes(CFG).items()}
    acp = Ls.alphas_cumprod("linear_beta", 1000, 0.0015, 0.0195)
    pool = torch.from_numpy(eeg_windows(POOL, seed=888))
    opt, traj = {}, []
    for i in range(1, STEPS + 1):
        s = ((i - 1) * B) % POOL
        nz = torch.from_numpy(normal((B, 1, 3072), seed=400 + i)); t = torch.from_numpy(timesteps(B, seed=500 + i))
        l, grads, _ = S.dm_train_step(sd, CFG, acp, pool[s:s + B], nz, t, spectral_weight=1e-6, spectral_loss=True)
        sd = S.adam_update(sd, grads, opt, 1e-4, i)
        traj.append(float(l))
        print(i, traj[-1], flush=True)
    with open(os.path.join(HERE, "dm_traj_c5.json"), "w") as fh:
        json.dump({"steps": STEPS, "batch": B, "pool": POOL, "window_seed": 888, "noise_seed_base": 400, "t_seed_base": 500, "param_seed": 42,
                   "lr": 1e-4, "spectral_weight": 1e-6, "schedule": ["linear_beta", 1000, 0.0015, 0.0195], "loss": traj}, fh, indent=0)


if __name__ == "__main__":
    main()
