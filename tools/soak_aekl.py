"""Soak / convergence run of the AutoencoderKL + PatchDiscriminator GAN step (train_autoencoderkl.py:203-234 as one native call):
N optimiser steps over a fixed pool of synthetic 30-s windows for the two autoencoder widths the reference uses ([2,2,4]: the
whole-network `aekl_thin` kernels; [32,32,64]: the layer-by-layer path), bf16 engine and -- for the first steps -- the fp32 engine
on the same seeds.  Prints the six loss terms, torch's allocator state, and at the end the reconstruction error of held-out windows.

    python tools/soak_aekl.py --steps 300 --batch 64 --fp32_steps 40
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
NAMES = ["l1", "spectral", "kl", "g_adv", "d_fake", "d_real"]


def run(channels, dtype, steps, B, L, pool, every):
    import torch
    import eegldm
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from eegldm.training import Adam, aekl_train_step, randn
    from param_gen import eeg_windows

    ctx = eegldm.default_context(0)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=channels, latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=0)
    disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3,
                              norm="BATCH", bias=False, padding=1, dtype=dtype, device=0)
    og, od = Adam(ae, lr=5e-3 if channels[0] <= 4 else 1e-3), Adam(disc, lr=5e-4)        # config_aekl_eeg.yaml learning rates (thin config)
    xs = torch.from_numpy(eeg_windows(pool + B, seed=777, length=L)).to(dev)
    lo = torch.zeros(6, device=dev)
    curve, mem = [], []
    t0 = time.time()
    for i in range(steps):
        s = (i * B) % pool
        xb = xs[s:s + B] if s + B <= pool else torch.cat([xs[s:pool], xs[:s + B - pool]])
        eps = randn(ctx, (B, 1, L // 4), seed=21, offset=i * B * (L // 4))
        ae.zero_grad(); disc.zero_grad()
        aekl_train_step(ae, disc, xb, eps, 0.01, 1e-9, 1e4, True, losses_out=lo)             # config_aekl_eeg.yaml:13-17 weights
        og.step(); od.step()
        if i % every == 0 or i == steps - 1:
            v = [float(t) for t in lo.cpu()]
            curve.append((i, v)); mem.append(torch.cuda.memory_allocated())
            print(f"[{channels} {dtype}] step {i:4d} " + " ".join(f"{n} {x:.5g}" for n, x in zip(NAMES, v)) +
                  f"  alloc {mem[-1] / 2**20:.0f} MiB {time.time() - t0:.1f}s", flush=True)
    ae.eval()
    held = xs[pool:pool + B]
    rec = ae.reconstruct(held) if hasattr(ae, "reconstruct") else ae(held)[0]
    err = float((rec.float() - held).abs().mean()); ref = float(held.abs().mean())
    print(f"[{channels} {dtype}] held-out windows: mean |recon - x| {err:.5f} against mean |x| {ref:.5f}", flush=True)
    out = {"channels": channels, "dtype": dtype, "curve": curve, "alloc_first_last": [mem[0], mem[-1]], "heldout_l1": err, "heldout_abs": ref}
    del ae, disc, og, od
    torch.cuda.empty_cache()
    return out


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=300); p.add_argument("--fp32_steps", type=int, default=0)
    p.add_argument("--batch", type=int, default=64); p.add_argument("--length", type=int, default=3072)
    p.add_argument("--pool", type=int, default=512); p.add_argument("--every", type=int, default=20)
    p.add_argument("--out", default=None)
    a = p.parse_args()
    res = []
    for ch in ([2, 2, 4], [32, 32, 64]):
        rb = run(ch, "bfloat16", a.steps, a.batch, a.length, a.pool, a.every); res.append(rb)
        if a.fp32_steps:
            rf = run(ch, "float32", a.fp32_steps, a.batch, a.length, a.pool, a.every); res.append(rf)
            fb, ff = dict(rb["curve"]), dict(rf["curve"])
            for i in sorted(set(fb) & set(ff)):
                print(f"  {ch} step {i:4d}: l1 bf16 {fb[i][0]:.5f} fp32 {ff[i][0]:.5f}   spectral bf16 {fb[i][1]:.5g} fp32 {ff[i][1]:.5g}")
    if a.out:
        with open(a.out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
