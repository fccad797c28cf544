"""Pixel-space diffusion training directly on the (B,1,3072) windows -- counterpart of
/root/reference/src/train_pure_ldm.py + src/training/training_diffusion.py::train_epoch_diffusion (config_dm.yaml; BASELINE C5):
UNet with in/out channels forced to 1 (train_pure_ldm.py:113-115), DDPMScheduler("linear_beta", 0.0015, 0.0195) (:123-124),
Adam 1e-4 (:136), loss = mse(noise_pred, noise) [+ 1e-6 * JukeboxLoss(sum) with --spe spectral (:128-132, :157-158)].
One process per GPU under torch.distributed.run for data parallelism."""
import argparse
import os
import time

import torch

from .. import distributed as D
from ..models import UNetModel
from ..schedulers import DDPMScheduler
from ..training import Adam, GradScaler, dm_train_step, randint, randn
from .common import WindowLoader, load_config, rng_seed, setup_run_dir


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", required=True)
    p.add_argument("--path_train_ids", default=None); p.add_argument("--path_valid_ids", default=None)
    p.add_argument("--path_cached_data", default=None); p.add_argument("--path_pre_processed", default=None)
    p.add_argument("--spe", default="no-spectral"); p.add_argument("--type_dataset", default="edfx"); p.add_argument("--dataset", default="edfx")
    p.add_argument("--synthetic_windows", type=int, default=0); p.add_argument("--dtype", default="float32")
    p.add_argument("--max_steps", type=int, default=0); p.add_argument("--output_dir", default=None)
    p.add_argument("--grad_scaler", action="store_true", help="dynamic loss scaling as training_diffusion.py:37 does (always on with --dtype float16)")
    p.add_argument("--deterministic", action="store_true", help="bit-reproducible steps (eegldm.set_deterministic(): ordered reductions instead of fp32 atomics; "
                   "what torch.use_deterministic_algorithms(True) would be for the reference's loop)")
    return p.parse_args(argv)


def main(args):
    if getattr(args, "deterministic", False):
        from .._lib import set_deterministic
        set_deterministic(True)
    rank, local, world = D.init_from_env()
    torch.cuda.set_device(local)
    config = load_config(args.config_file)
    torch.manual_seed(config.train.seed)
    run_dir, resume = setup_run_dir(config, args)
    up = dict(config.model.params.unet_config.params)
    up["in_channels"] = up["out_channels"] = 1                                  # train_pure_ldm.py:113-115
    unet = UNetModel(**up, dtype=args.dtype, device=local)
    D.broadcast_flat(unet.flat); unet.sync_weights()
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195, device=local)
    opt = Adam(unet, lr=1e-4)
    # training_diffusion.py:37,149-151 pairs its fp16 autocast with a GradScaler: fp16 activation gradients under- / overflow without the loss scale
    scaler = GradScaler(enabled=args.grad_scaler or str(args.dtype) in ("float16", "fp16", "half"))
    spectral = args.spe == "spectral"
    bs = max(1, config.train.batch_size // world)
    train = WindowLoader(args.path_pre_processed, bs, args.synthetic_windows, seed=rng_seed(config.train.seed, 8, rank, world), drop_last=config.train.drop_last,
                         path_ids=args.path_train_ids, dataset=args.type_dataset, shard=(rank, world))
    s_t, s_noise = rng_seed(config.train.seed, 1, rank, world), rng_seed(config.train.seed, 3, rank, world)
    dev, ctx = unet.device, unet.ctx
    loss = torch.zeros(1, device=dev)
    gsync = D.OverlappedGradSync(unet.flat_grad, ctx=unet.ctx, comm=D.make_comm(unet.ctx))   # no-op with one process; EEGLDM_NATIVE_COLLECTIVES=1: RCCL through the C ABI
    steps, t0, seen, best, start_epoch, gstep = 0, time.time(), 0, float("inf"), 0, 0      # gstep: steps over all invocations (RNG offsets)
    if resume:      # continue from {run_dir}/checkpoint.pth
        ck = torch.load(os.path.join(run_dir, "checkpoint.pth"), map_location="cpu")
        unet.load_state_dict(ck["diffusion"]); opt.load_state_dict(ck["optimizer"])
        if "scaler" in ck:
            scaler.load_state_dict(ck["scaler"])
        start_epoch, best, gstep = int(ck["epoch"]), float(ck["best_loss"]), int(ck.get("steps", 0))
        if rank == 0:
            print(f"Resuming from epoch {start_epoch} (best loss {best:.5f})")
    for epoch in range(start_epoch, config.train.n_epochs):
        unet.train()
        for batch in train:
            x = batch["eeg"].to(dev)
            B = x.shape[0]
            t = randint(ctx, B, sched.num_train_timesteps, seed=s_t, offset=gstep * B)
            noise = randn(ctx, tuple(x.shape), seed=s_noise, offset=gstep * x.numel())
            opt.zero_grad()
            dm_train_step(unet, sched, x, noise, t, spectral_weight=1e-6, spectral_loss=spectral, loss_out=loss, grad_sync=gsync,
                          grad_scale=scaler.get_scale())
            gsync.wait()
            scaler.step(opt); scaler.update()
            steps += 1; gstep += 1; seen += B * world
            if args.max_steps and steps >= args.max_steps:
                break
        if rank == 0:
            print(f"epoch {epoch}: loss {float(loss):.5f} | {seen/(time.time()-t0):.1f} windows/s", flush=True)
            cur = float(loss)
            if cur <= best:
                best = cur
                torch.save({k: v.cpu() for k, v in unet.state_dict().items()}, os.path.join(run_dir, "best_model.pth"))
            torch.save({"epoch": epoch + 1, "diffusion": {k: v.cpu() for k, v in unet.state_dict().items()}, "optimizer": opt.state_dict(),
                        "best_loss": best, "steps": gstep, "scaler": scaler.state_dict()}, os.path.join(run_dir, "checkpoint.pth"))
        if args.max_steps and steps >= args.max_steps:
            break
    if rank == 0:
        torch.save({k: v.cpu() for k, v in unet.state_dict().items()}, os.path.join(run_dir, "final_model.pth"))
    return run_dir


if __name__ == "__main__":
    main(parse_args())
