"""Soak / convergence run of the production latent-diffusion train step (config_ldm.yaml UNet over frozen AutoencoderKL latents):
N optimiser steps over a fixed pool of synthetic windows, the same seeds through the bf16 engine and (optionally) the fp32 engine.
Prints the loss every `--every` steps, torch's allocated/reserved bytes (a leak shows as growth) and, at the end, DDIM-50 samples'
statistics.  Used to check that all the fused / grouped / side-stream paths together still TRAIN (loss falls, bf16 follows fp32),
which no single-step parity test can show.

    python tools/soak_ldm.py --steps 400 --batch 256 --fp32_steps 100
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def run(dtype, steps, B, L, pool, every, lr, sample):
    import torch
    import eegldm
    from eegldm.models import UNetModel, AutoencoderKL
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, GradScaler, ldm_train_step, randint, randn
    from eegldm.sampling import ddim_sample, make_sampling_scheduler
    from param_gen import eeg_windows
    from bench import UNET_CFG

    ctx = eegldm.default_context(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    unet = UNetModel(**UNET_CFG, dtype=dtype, device=0)                 # module default init (zero-initialised out conv etc.)
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=0)
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195, device=0)
    opt = Adam(unet, lr=lr)
    scaler = GradScaler(enabled=(dtype == "float16"))       # the reference's AMP recipe (training.py:334,441-443): fp16 storage needs it
    skipped = 0
    loss = torch.zeros(1, device=dev)
    windows = torch.from_numpy(eeg_windows(pool, seed=4321, length=4 * L)).to(dev)
    sf = 1.0 / float(ae.encode_stage_2_inputs(windows[:B], eps=randn(ctx, (B, 1, L), seed=99)).std())
    curve, mem = [], []
    t0 = time.time()
    for i in range(steps):
        lo = (i * B) % pool
        xb = windows[lo:lo + B] if lo + B <= pool else torch.cat([windows[lo:], windows[:lo + B - pool]])
        t = randint(ctx, B, 1000, seed=11, offset=i * B)
        noise = randn(ctx, (B, 1, L), seed=12, offset=i * B * L)
        eps = randn(ctx, (B, 1, L), seed=13, offset=i * B * L)
        lat = ae.encode_stage_2_inputs(xb, eps=eps, scale_factor=sf)
        unet.zero_grad()
        ldm_train_step(unet, sched, lat, noise, t, loss_out=loss, grad_scale=scaler.get_scale())
        scaler.step(opt)
        if scaler.is_enabled() and scaler._found_inf: skipped += 1
        scaler.update()
        if i % every == 0 or i == steps - 1:
            curve.append((i, float(loss))); mem.append((torch.cuda.memory_allocated(), torch.cuda.memory_reserved()))
            print(f"[{dtype}] step {i:5d} loss {curve[-1][1]:.5f} alloc {mem[-1][0] / 2**20:.0f} MiB reserved {mem[-1][1] / 2**20:.0f} MiB "
                  f"{time.time() - t0:.1f}s", flush=True)
    if scaler.is_enabled(): print(f"[{dtype}] GradScaler: final scale {scaler.get_scale():.0f}, {skipped} skipped steps", flush=True)
    out = {"dtype": dtype, "curve": curve, "grad_scaler": {"enabled": scaler.is_enabled(), "final_scale": scaler.get_scale(), "skipped_steps": skipped}, "alloc_first_last": [mem[0][0], mem[-1][0]], "reserved_first_last": [mem[0][1], mem[-1][1]]}
    if sample:
        unet.eval()
        ss = make_sampling_scheduler(50, device=0)
        x, z = ddim_sample(unet, ae, ss, randn(ctx, (64, 1, L), seed=77), scale_factor=sf)
        x = x.float()
        out["sample"] = {"latent_mean": float(z.float().mean()), "latent_std": float(z.float().std()), "finite": bool(torch.isfinite(x).all()),
                         "window_std": float(x.std())}
        print(f"[{dtype}] DDIM-50 of 64 windows: latent mean {out['sample']['latent_mean']:.4f} std {out['sample']['latent_std']:.4f}, "
              f"decoded std {out['sample']['window_std']:.4f}, finite {out['sample']['finite']}", flush=True)
    del unet, ae, opt
    torch.cuda.empty_cache()
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=400); p.add_argument("--fp32_steps", type=int, default=0)
    p.add_argument("--batch", type=int, default=256); p.add_argument("--length", type=int, default=768)
    p.add_argument("--pool", type=int, default=2048); p.add_argument("--every", type=int, default=25)
    p.add_argument("--lr", type=float, default=1e-4); p.add_argument("--no_sample", action="store_true")
    p.add_argument("--out", default=None); p.add_argument("--dtype", default="bfloat16", choices=["bfloat16", "float16"])
    a = p.parse_args()
    res = [run(a.dtype, a.steps, a.batch, a.length, a.pool, a.every, a.lr, not a.no_sample)]
    if a.fp32_steps:
        res.append(run("float32", a.fp32_steps, a.batch, a.length, a.pool, a.every, a.lr, False))
        f = dict(res[1]["curve"]); b = dict(res[0]["curve"])
        gaps = [(i, b[i], f[i], abs(b[i] - f[i]) / f[i]) for i in sorted(set(f) & set(b))]
        print(f"{a.dtype} vs fp32 loss on common steps (step, {a.dtype}, fp32, rel gap):")
        for g in gaps:
            print("  %5d %.5f %.5f %.4f" % g)
        res.append({"max_rel_gap": max(g[3] for g in gaps)})
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
