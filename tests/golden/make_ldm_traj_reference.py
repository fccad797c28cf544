"""Generates tests/golden/ldm_traj_c2_reference.json by running the REFERENCE's UNetModel through the reference's train-step body
(build container only: needs /root/reference).

    python tests/golden/make_ldm_traj_reference.py

The 30 steps of ldm_traj_c2.json (same seeded parameters, latents, noise and timesteps) with the loop body of
/root/reference/src/training/training.py:419-443 -- `optimizer.zero_grad(set_to_none=True)`; under autocast: add_noise, `model(x=noisy_e,
timesteps=timesteps)`, `F.mse_loss(noise_pred.float(), target.float())`; `scaler.scale(loss).backward(); scaler.step(optimizer);
scaler.update()` -- on `models.unet.UNetModel` (imported) with `torch.optim.Adam(lr=1e-4)` (train_ldm.py:208) and the q_sample arithmetic of
models/ldm.py:392-408 for add_noise (MONAI's DDPMScheduler is not installed; tests/golden/ddpm_steps.npz pins that formula to the reference).
Three runs:
  loss_fp32      autocast off: what the oracle's trajectory (make_ldm_traj.py) restates -- tests/test_oracle_trajectory.py compares the two;
  loss_bf16      torch.autocast("cpu", dtype=torch.bfloat16);
  loss_f16       torch.autocast("cpu", dtype=torch.float16) + torch.amp.GradScaler: the reference's own configuration (training.py:334,423);
  loss_f16_scale1024   the same with init_scale = 1024 (no back-off on either side: the run the fp16 engine is compared with step by step).
tests/test_gpu_zz_convergence.py holds the 16-bit engines' trajectories to the distance the reference's own reduced-precision runs keep from fp32.
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/src")
from param_gen import gen_param, eeg_windows, normal, timesteps      # noqa: E402
from models import unet as R                                          # noqa: E402  (the reference module)

G = json.load(open(os.path.join(HERE, "ldm_traj_c2.json")))
CFG = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
           channel_mult=[1, 2, 4], dropout=0.0, conv_resample=True, num_heads=1, use_scale_shift_norm=False, resblock_updown=True)   # config_ldm.yaml:30-43


def run(dtype, init_scale=65536.0):
    torch.manual_seed(0)
    net = R.UNetModel(**CFG)
    net.load_state_dict({k: torch.from_numpy(gen_param(G["param_seed"], k, v.shape)) for k, v in net.state_dict().items()})
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=G["lr"])
    use_scaler = dtype == torch.float16
    scaler = torch.amp.GradScaler("cpu", init_scale=init_scale, enabled=use_scaler)
    _, n, b0, b1 = G["schedule"]
    acp = torch.cumprod(1.0 - torch.linspace(b0, b1, n, dtype=torch.float32), dim=0)          # "linear_beta" (train_ldm.py:199-200)
    pool = torch.from_numpy(eeg_windows(G["pool"], seed=G["latent_seed"], length=768))
    B, traj, skipped = G["batch"], [], 0
    for i in range(1, G["steps"] + 1):
        s = ((i - 1) * B) % G["pool"]
        e = pool[s:s + B]
        noise = torch.from_numpy(normal((B, 1, 768), seed=G["noise_seed_base"] + i)); t = torch.from_numpy(timesteps(B, seed=G["t_seed_base"] + i))
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cpu", dtype=dtype if dtype is not None else torch.bfloat16, enabled=dtype is not None):
            sa = acp[t].sqrt().reshape(-1, 1, 1); sb = (1 - acp[t]).sqrt().reshape(-1, 1, 1)
            noisy_e = sa * e + sb * noise
            noise_pred = net(x=noisy_e, timesteps=t)
            loss = F.mse_loss(noise_pred.float(), noise.float())
        scale_before = scaler.get_scale() if use_scaler else 1.0
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        if use_scaler and scaler.get_scale() < scale_before:
            skipped += 1
        traj.append(float(loss))
        print(str(dtype), i, traj[-1], flush=True)
    return traj, skipped


def main():
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    path = os.path.join(HERE, "ldm_traj_c2_reference.json")
    out = {"fixture": "ldm_traj_c2.json", "body": "training.py:419-443 on models.unet.UNetModel, torch.optim.Adam(lr 1e-4), q_sample add_noise"}
    if "--add-missing" in sys.argv and os.path.exists(path):      # ~10 minutes per run on 8 cores: keep the runs a previous invocation recorded
        out = json.load(open(path))
    if "loss_fp32" not in out:
        out["loss_fp32"], _ = run(None)
    if "loss_bf16" not in out:
        out["loss_bf16"], _ = run(torch.bfloat16)
    if "loss_f16" not in out:
        out["loss_f16"], out["f16_scaler_backoffs"] = run(torch.float16)
    # torch's CPU fp16 kernels overflow at the default initial scale (two back-offs: the curve above runs two optimiser steps behind); an engine
    # that accumulates in fp32 does not, so the step-by-step comparison of fp16 NUMERICS is made at an initial scale neither side backs off from
    if "loss_f16_scale1024" not in out:
        out["loss_f16_scale1024"], out["f16_scale1024_backoffs"] = run(torch.float16, init_scale=1024.0)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0)
    import numpy as np
    ref = np.array(out["loss_fp32"]); orc = np.array(G["loss"])
    print("reference fp32 vs oracle fixture: worst relative gap", float(np.max(np.abs(ref - orc) / orc)))
    print("fp16 back-offs: default scale", out["f16_scaler_backoffs"], "| init_scale 1024:", out["f16_scale1024_backoffs"])
    for k in ("loss_bf16", "loss_f16", "loss_f16_scale1024"):
        print(k, "vs reference fp32: worst relative gap", float(np.max(np.abs(np.array(out[k]) - ref) / ref)))


if __name__ == "__main__":
    main()
