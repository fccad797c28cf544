"""AutoencoderKL + PatchDiscriminator training -- counterpart of /root/reference/src/train_autoencoderkl.py
(same flags and yaml schema; checkpoint files/keys as :316-338).  `python -m eegldm.entry.train_autoencoderkl --config_file ...`

Data parallelism: the reference wraps BOTH networks in nn.DataParallel (train_autoencoderkl.py:141-144).  Here it is one process per
GPU under torch.distributed.run: parameters and the discriminator's BatchNorm buffers are broadcast from rank 0 once, the loader is
rank-sharded (per-GPU batch = batch_size / world), and after the fused GAN step the two flat gradient buffers (autoencoder 0.70 MB,
discriminator 2.08 MB at [32,32,64]) are all-reduced to their mean before the two Adam steps.  BatchNorm batch statistics stay per
rank and rank 0's running statistics are the ones checkpointed -- exactly what DataParallel does (replica 0 shares the module's
buffers)."""
import argparse
import os
import time

import torch

from .. import distributed as D
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
    p.add_argument("--deterministic", action="store_true", help="bit-reproducible steps (eegldm.set_deterministic(): ordered reductions instead of fp32 atomics; "
                   "what torch.use_deterministic_algorithms(True) would be for the reference's loop)")
    return p.parse_args(argv)


LAST_RUN = {}      # what the most recent main() ended with (rank-local): read by the multi-rank tests


def main(args):
    if getattr(args, "deterministic", False):
        from .._lib import set_deterministic
        set_deterministic(True)
    rank, local, world = D.init_from_env()
    torch.cuda.set_device(local)
    config = load_config(args.config_file)
    torch.manual_seed(config.train.seed)
    run_dir, resume = setup_run_dir(config, args)
    ae_args = dict(config.autoencoderkl.params)
    if args.num_channels is not None:            # CLI overrides yaml only when given
        ae_args["num_channels"] = args.num_channels
    if args.latent_channels is not None:
        ae_args["latent_channels"] = args.latent_channels
    if str(args.dtype) in ("float16", "fp16", "half"):
        # The reference runs this loop in fp32 without autocast (train_autoencoderkl.py:200-234; the GradScaler variant in
        # training/training.py:52-207 is commented out) and the fused native step has no loss-scale argument: fp16 activation
        # gradients would under- / overflow unguarded.  bfloat16 (fp32's exponent range) is the 16-bit mode of this entry point.
        raise ValueError("--dtype float16 is not supported by train_autoencoderkl (no loss scaling in the AEKL / GAN step): use bfloat16 or float32")
    model = AutoencoderKL(**ae_args, dtype=args.dtype, device=local)
    disc = PatchDiscriminator(**dict(config.patchdiscriminator.params), dtype=args.dtype, device=local)
    opt_g, opt_d = Adam(model, lr=config.models.optimizer_g_lr), Adam(disc, lr=config.models.optimizer_d_lr)
    adv_w, kl_w = config.models.adv_weight, config.models.kl_weight
    spec_w = config.models.get("spectral_weight", 0.0)
    bs = max(1, config.train.batch_size // world)
    train = WindowLoader(args.path_pre_processed, bs, args.synthetic_windows, seed=rng_seed(config.train.seed, 8, rank, world), drop_last=config.train.drop_last,
                         path_ids=args.path_train_ids, dataset=args.type_dataset, shard=(rank, world))
    # validation reads the VALID split (dataset.py:83-99); without id CSVs (synthetic / bare directory runs) it is a held-out synthetic set
    # or, as a last resort, the same directory.  Every rank scores its shard; (sum, count) are added over ranks
    val = WindowLoader(args.path_pre_processed, bs, max(args.synthetic_windows // 4, config.train.batch_size) // world if args.synthetic_windows else 0,
                       seed=rng_seed(config.train.seed, 9, rank, world), shuffle=False, path_ids=args.path_valid_ids, dataset=args.type_dataset, shard=(rank, world))
    start_epoch, best, steps = 0, float("inf"), 0
    init_batch = None
    if resume:
        ck = torch.load(os.path.join(run_dir, "checkpoint.pth"), map_location="cpu")
        model.load_state_dict(ck["state_dict"]); disc.load_state_dict(ck["discriminator"])
        opt_g.load_state_dict(ck["optimizer_g"]); opt_d.load_state_dict(ck["optimizer_d"])
        start_epoch, best = ck["epoch"], ck["best_loss"]
        steps = int(ck.get("steps", 0))       # global step = the RNG offset of the reparameterisation noise: a resumed run must not replay it
        init_batch = ck.get("init_batch")     # written by the reference (train_autoencoderkl.py:327) and read unconditionally on its resume (:182)
    # identical replicas: parameters of both networks and the discriminator's BatchNorm running statistics come from rank 0
    D.broadcast_flat(model.flat); model.sync_weights()
    D.broadcast_flat(disc.flat); D.broadcast_flat(disc.buffers); disc.sync_weights()
    dev, ctx = model.device, model.ctx
    losses = torch.zeros(6, device=dev)
    if init_batch is None:                    # first(train_loader)['eeg'][:, :, 36:-36] (train_autoencoderkl.py:188): the batch the reference logs reconstructions of
        for b0 in train:
            init_batch = b0["eeg"][:, :, 36:-36].clone().cpu()
            break
    t0, seen, steps_run = time.time(), 0, 0
    s_eps = rng_seed(config.train.seed, 4, rank, world)
    for epoch in range(start_epoch, config.train.n_epochs):
        model.train(); disc.train()
        acc = torch.zeros(6)
        for batch in train:
            x = batch["eeg"].to(dev)
            eps = randn(ctx, (x.shape[0], model.latent_channels, x.shape[2] // model.down), seed=s_eps, offset=steps * x.shape[0] * (x.shape[2] // model.down) * model.latent_channels)
            opt_g.zero_grad(); opt_d.zero_grad()
            aekl_train_step(model, disc, x, eps, adv_w, kl_w, spec_w, args.spe == "spectral", losses_out=losses)
            # the ONE exchange of the step: mean over ranks of both flat gradient buffers, then identical Adam steps everywhere
            D.allreduce_mean_flat(model.flat_grad); D.allreduce_mean_flat(disc.flat_grad)
            opt_g.step(); opt_d.step()
            acc += losses.cpu(); steps += 1; steps_run += 1; seen += x.shape[0] * world
            if args.max_steps and steps_run >= args.max_steps:
                break
        n = max(1, len(train))
        acc = torch.tensor(D.allreduce_sum_scalars(acc.tolist(), like=losses)) / world       # logged losses: mean over the replicas' batches
        if rank == 0:
            print(f"epoch {epoch}: recons {acc[0]/n:.5f} spectral {acc[1]/n:.3f} kl {acc[2]/n:.3f} gen {acc[3]/n:.5f} disc {(acc[4]+acc[5])/(2*n):.5f} "
                  f"| {seen/(time.time()-t0):.1f} windows/s", flush=True)
        if (epoch + 1) % config.train.val_interval == 0 or (args.max_steps and steps_run >= args.max_steps):
            model.eval()
            v_sum, v_n = 0.0, 0
            for b in val:
                xv = b["eeg"].to(dev)
                v_sum += float((model.reconstruct(xv) - xv).abs().mean()) * xv.shape[0]; v_n += xv.shape[0]
            v_sum, v_n = D.allreduce_sum_scalars([v_sum, v_n], like=losses)
            vl = v_sum / max(1.0, v_n)
            if rank == 0:
                if vl <= best:
                    best = vl
                    torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(run_dir, "best_model.pth"))
                torch.save({"epoch": epoch + 1, "state_dict": {k: v.cpu() for k, v in model.state_dict().items()},
                            "discriminator": {k: v.cpu() for k, v in disc.state_dict().items()}, "optimizer_g": opt_g.state_dict(),
                            "optimizer_d": opt_d.state_dict(), "best_loss": best, "init_batch": init_batch, "steps": steps}, os.path.join(run_dir, "checkpoint.pth"))
            best = D.broadcast_scalar(best, src=0, like=losses)
        if args.max_steps and steps_run >= args.max_steps:
            break
    if rank == 0:
        torch.save({k: v.cpu() for k, v in model.state_dict().items()}, os.path.join(run_dir, "final_model.pth"))
    LAST_RUN.clear(); LAST_RUN.update(rank=rank, world=world, steps=steps_run, ae_sum=float(model.flat.double().sum()), d_sum=float(disc.flat.double().sum()))
    return run_dir


if __name__ == "__main__":
    main(parse_args())
