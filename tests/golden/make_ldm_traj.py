"""Generates tests/golden/ldm_traj_c2.json: the epsilon-MSE trajectory of 30 optimiser steps of the config_ldm.yaml UNet
(training.py:419-443: add_noise -> UNet -> MSE -> Adam 1e-4; linear betas 0.0015-0.0195 as train_ldm.py:199-200) computed by the
CPU oracle (oracle/steps.py::ldm_train_step + adam_update, fp32 torch autograd) on seeded parameters, latents, noise and timesteps.
tests/test_gpu_zz_convergence.py replays the same steps through the HIP engines.  ~20 s on 16 cores.

    python tests/golden/make_ldm_traj.py
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

CFG = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
           channel_mult=[1, 2, 4], resblock_updown=True)               # config_ldm.yaml:30-43
STEPS, B, POOL = 30, 8, 64


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    sd = {k: torch.from_numpy(gen_param(42, k, s)) for k, s in U.unet_param_shapes(CFG).items()}
    acp = Ls.alphas_cumprod("linear_beta", 1000, 0.0015, 0.0195)
    pool = torch.from_numpy(eeg_windows(POOL, seed=555, length=768))          # stand-in latents with temporal structure
    opt, traj = {}, []
    for i in range(1, STEPS + 1):
        s = ((i - 1) * B) % POOL
        nz = torch.from_numpy(normal((B, 1, 768), seed=200 + i)); t = torch.from_numpy(timesteps(B, seed=300 + i))
        l, grads, _ = S.ldm_train_step(sd, CFG, acp, pool[s:s + B], nz, t)
        sd = S.adam_update(sd, grads, opt, 1e-4, i)
        traj.append(float(l))
        print(i, traj[-1], flush=True)
    with open(os.path.join(HERE, "ldm_traj_c2.json"), "w") as fh:
        json.dump({"steps": STEPS, "batch": B, "pool": POOL, "latent_seed": 555, "noise_seed_base": 200, "t_seed_base": 300, "param_seed": 42,
                   "lr": 1e-4, "schedule": ["linear_beta", 1000, 0.0015, 0.0195], "loss": traj}, fh, indent=0)


if __name__ == "__main__":
    main()
