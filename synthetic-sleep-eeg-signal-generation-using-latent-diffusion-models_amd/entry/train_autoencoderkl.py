"""AutoencoderKL + PatchDiscriminator training -- counterpart of /root/reference/src/train_autoencoderkl.py
(same flags and yaml schema; checkpoint files/keys as :316-338).  `python -m eegldm.entry.train_autoencoderkl --config_file ...`"""
import argparse
import os
import time

import torch

from ..models import AutoencoderKL, PatchDiscriminator
from ..training import Adam, aekl_train_step, randn
from .common import ParseListAction, WindowLoader, load_config, rng_seed, setup_run_dir


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", required=True)
    p.add_argument("--path_train_ids", default=None); p.add_argument("--path_valid_ids", default=None)
    p.add_argument("--path_cached_data", default=None); p.add_argument("--path_pre_processed", default=None)
    p.add_argument("--num_channels", action=ParseListAction, default=None)
    p.add_argument("--spe", default="no-spectral", choices=["spectral", "no-spectral"])
    p.add_argument("--latent_channels", type=int, default=None)
    p.add_argument("--type_dataset", default="edfx"); p.add_argument("--dataset", default="edfx", choices=["edfx", "shhs", "shhsh"])
    # additions (not in the reference): synthetic data, engine dtype, short runs, output override
    p.add_argument("--synthetic_windows", type=int, default=0); p.add_argument("--dtype", default="float32")
    p.add_argument("--max_steps", type=int, default=0); p.add_argument("--output_dir", default=None)
    return p.parse_args(argv)


def main(args):
    config = load_config(args.config_file)
    torch.manual_seed(config.train.seed)
    run_dir, resume = setup_run_dir(config, args)
    ae_args = dict(config.autoencoderkl.params)
    if args.num_channels is not None:            # CLI overrides yaml only when given
        ae_args["num_channels"] = args.num_channels
    if args.latent_channels is not None:
        ae_args["latent_channels"] = args.latent_channels
    model = AutoencoderKL(**ae_args, dtype=args.dtype)
    disc = PatchDiscriminator(**dict(config.patchdiscriminator.params), dtype=args.dtype)
    opt_g, opt_d = Adam(model, lr=config.models.optimizer_g_lr), Adam(disc, lr=config.models.optimizer_d_lr)
    adv_w, kl_w = config.models.adv_weight, config.models.kl_weight
    spec_w = config.models.get("spectral_weight", 0.0)
    train = WindowLoader(args.path_pre_processed, config.train.batch_size, args.synthetic_windows, seed=config.train.seed, drop_last=config.train.drop_last,
                         path_ids=args.path_train_ids, dataset=args.type_dataset)
    # validation reads the VALID split (dataset.py:83-99); without id CSVs (synthetic / bare directory runs) it is a held-out synthetic set
    # or, as a last resort, the same directory
    val = WindowLoader(args.path_pre_processed, config.train.batch_size, max(args.synthetic_windows // 4, config.train.batch_size) if args.synthetic_windows else 0,
                       seed=config.train.seed + 1, shuffle=False, path_ids=args.path_valid_ids, dataset=args.type_dataset)
    start_epoch, best, steps = 0, float("inf"), 0
    if resume:
        ck = torch.load(os.path.join(run_dir, "checkpoint.pth"), map_location="cpu")
        model.load_state_dict(ck["state_dict"]); disc.load_state_dict(ck["discriminator"])
        opt_g.load_state_dict(ck["optimizer_g"]); opt_d.load_state_dict(ck["optimizer_d"])
        start_epoch, best = ck["epoch"], ck["best_loss"]
        steps = int(ck.get("steps", 0))       # global step = the RNG offset of the reparameterisation noise: a resumed run must not replay it
    dev, ctx = model.device, model.ctx
    losses = torch.zeros(6, device=dev)
    t0, seen, steps_run = time.time(), 0, 0
    s_eps = rng_seed(config.train.seed, 4)
    for epoch in range(start_epoch, config.train.n_epochs):
        model.train(); disc.train()
        acc = torch.zeros(6)
        for batch in train:
            x = batch["eeg"].to(dev)
            eps = randn(ctx, (x.shape[0], model.latent_channels, x.shape[2] // model.down), seed=s_eps, offset=steps * x.shape[0] * (x.shape[2] // model.down) * model.latent_channels)
            opt_g.zero_grad(); opt_d.zero_grad()
            aekl_train_step(model, disc, x, eps, adv_w, kl_w, spec_w, args.spe == "spectral", losses_out=losses)
            opt_g.step(); opt_d.step()
            acc += losses.cpu(); steps += 1; steps_run += 1; seen += x.shape[0]
            if args.max_steps and steps_run >= args.max_steps:
                break
        n = max(1, len(train))
        print(f"epoch {epoch}: recons {acc[0]/n:.5f} spectral {acc[1]/n:.3f} kl {acc[2]/n:.3f} gen {acc[3]/n:.5f} disc {(acc[4]+acc[5])/(2*n):.5f} "
              f"| {seen/(time.time()-t0):.1f} windows/s", flush=True)
        if (epoch + 1) % config.train.val_interval == 0 or (args.max_steps and steps_run >= args.max_steps):
            model.eval()
            vl = sum(float((model.reconstruct(b["eeg"].to(dev)) - b["eeg"].to(dev)).abs().mean()) for b in val) / max(1, len(val))
            if vl <= best:
                best = vl
                torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(run_dir, "best_model.pth"))
            torch.save({"epoch": epoch + 1, "state_dict": {k: v.cpu() for k, v in model.state_dict().items()},
                        "discriminator": {k: v.cpu() for k, v in disc.state_dict().items()}, "optimizer_g": opt_g.state_dict(),
                        "optimizer_d": opt_d.state_dict(), "best_loss": best, "steps": steps}, os.path.join(run_dir, "checkpoint.pth"))
        if args.max_steps and steps_run >= args.max_steps:
            break
    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(run_dir, "final_model.pth"))
    return run_dir


if __name__ == "__main__":
    main(parse_args())
