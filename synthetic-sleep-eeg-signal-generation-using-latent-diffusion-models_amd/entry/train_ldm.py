"""Latent-diffusion UNet training over frozen AutoencoderKL latents -- counterpart of
/root/reference/src/train_ldm.py + src/training/training.py::train_ldm (flags, yaml schema, checkpoint keys
training.py:381-397).  One process per GPU under torch.distributed.run for data parallelism."""
import argparse
import os
import time

import torch

from .. import distributed as D
from ..models import AutoencoderKL, UNetModel
from ..schedulers import DDPMScheduler
from ..training import Adam, GradScaler, ldm_train_step, randint, randn
from .common import ParseListAction, WindowLoader, load_config, rng_seed, setup_run_dir


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--config_file", required=True)
    p.add_argument("--path_train_ids", default=None); p.add_argument("--path_valid_ids", default=None)
    p.add_argument("--path_cached_data", default=None); p.add_argument("--path_pre_processed", default=None)
    p.add_argument("--num_channels", action=ParseListAction, default=None)
    p.add_argument("--autoencoderkl_config_file_path", required=True); p.add_argument("--best_model_path", default=None)
    p.add_argument("--spe", default="no-spectral"); p.add_argument("--latent_channels", type=int, default=1)
    p.add_argument("--type_dataset", default="edfx"); p.add_argument("--dataset", default="edfx")
    p.add_argument("--synthetic_windows", type=int, default=0); p.add_argument("--dtype", default="float32")
    p.add_argument("--max_steps", type=int, default=0); p.add_argument("--output_dir", default=None)
    p.add_argument("--prediction_type", default="epsilon")
    p.add_argument("--schedule", default="linear_beta", choices=["linear_beta", "scaled_linear_beta"],
                   help="training noise schedule; the reference builds DDPMScheduler(beta_schedule='linear') = plain linspace (train_ldm.py:199-200)")
    p.add_argument("--grad_scaler", action="store_true", help="dynamic loss scaling as in the reference loop (training.py:334,441-443); always on with --dtype float16 (the reference's autocast dtype), bf16/fp32 do not need it")
    p.add_argument("--deterministic", action="store_true", help="bit-reproducible steps (eegldm.set_deterministic(): ordered reductions instead of fp32 atomics; "
                   "what torch.use_deterministic_algorithms(True) would be for the reference's loop)")
    return p.parse_args(argv)


@torch.no_grad()
def validate(unet, stage1, sched, loader, scale_factor, seeds, latent_channels):
    """eval_ldm (training.py:455-497): mean epsilon-MSE over the validation windows, fixed noise stream.  `seeds` = the three Philox
    keys (timesteps, posterior eps, diffusion noise), each from `rng_seed` with its own role.  Returns (sum of per-window losses,
    number of windows) so that data-parallel ranks can add their shards up."""
    from .._lib import lib, check, ptr
    unet.eval()
    tot, n, dev, ctx = 0.0, 0, unet.device, unet.ctx
    s_t, s_eps, s_noise = seeds
    out = torch.zeros(1, device=dev)
    seen = 0
    for batch in loader:
        x = batch["eeg"].to(dev); B = x.shape[0]
        Ll = x.shape[2] // stage1.down
        per = latent_channels * Ll                      # random numbers per window: offsets never overlap, whatever latent_channels is
        t = randint(ctx, B, sched.num_train_timesteps, seed=s_t, offset=seen)
        eps = randn(ctx, (B, latent_channels, Ll), seed=s_eps, offset=seen * per)
        noise = randn(ctx, eps.shape, seed=s_noise, offset=seen * per)
        e = stage1.encode_stage_2_inputs(x, eps=eps, scale_factor=scale_factor)
        pred = unet(sched.add_noise(original_samples=e, noise=noise, timesteps=t), timesteps=t)
        check(lib.eegldm_mse_loss(ctx.h, ptr(pred), ptr(noise), ptr(out), None, pred.numel(), 1.0))
        tot += float(out) * B; n += B; seen += B
    unet.train()
    return tot, n


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
    ae_cfg = dict(load_config(args.autoencoderkl_config_file_path).autoencoderkl.params)
    if args.num_channels is not None:
        ae_cfg["num_channels"] = args.num_channels
    ae_cfg["latent_channels"] = args.latent_channels
    stage1 = AutoencoderKL(**ae_cfg, dtype=args.dtype, device=local)
    if args.best_model_path:
        stage1.load_state_dict(torch.load(os.path.join(args.best_model_path, "best_model.pth"), map_location="cpu"))
    stage1.eval()
    up = dict(config.model.params.unet_config.params)
    up["in_channels"] = up["out_channels"] = args.latent_channels            # train_ldm.py:184-187
    unet = UNetModel(**up, dtype=args.dtype, device=local)
    D.broadcast_flat(unet.flat); unet.sync_weights(); D.broadcast_flat(stage1.flat); stage1.sync_weights()
    # train_ldm.py:199-200: monai-generative DDPMScheduler(beta_schedule="linear", 0.0015, 0.0195) = plain linspace of the betas
    # (alpha_bar[999] = 2.57e-5).  The sqrt-space "linear" of the reference's local models/ldm.py is a different table and is not
    # what train_ldm uses; --schedule scaled_linear_beta selects it deliberately.
    sched = DDPMScheduler(num_train_timesteps=1000, schedule=args.schedule, beta_start=0.0015, beta_end=0.0195,
                          prediction_type=args.prediction_type, device=local)
    opt = Adam(unet, lr=config.train.get("base_lr", 1e-4))
    scaler = GradScaler(enabled=args.grad_scaler or str(args.dtype) in ("float16", "fp16", "half"))      # fp16 activations: the loss scale is what keeps their gradients out of the subnormal range
    bs = max(1, config.train.batch_size // world)
    train = WindowLoader(args.path_pre_processed, bs, args.synthetic_windows, seed=rng_seed(config.train.seed, 8, rank, world), drop_last=config.train.drop_last,
                         path_ids=args.path_train_ids, dataset=args.type_dataset, shard=(rank, world))
    # validation: every rank scores its own shard (equal lengths, wrap-around: a recording may be scored twice when N % world != 0)
    # and the (sum, count) pairs are added over ranks -- model selection sees the WHOLE validation split, as the reference's does
    valid = WindowLoader(args.path_pre_processed, bs, 0, seed=rng_seed(config.train.seed, 9, 0, world), shuffle=False, path_ids=args.path_valid_ids,
                         dataset=args.type_dataset, shard=(rank, world)) if args.path_valid_ids else None
    v_seeds = tuple(rng_seed(config.train.seed, role, rank, world) for role in (5, 6, 7))
    s_t, s_eps, s_noise = (rng_seed(config.train.seed, role, rank, world) for role in (1, 2, 3))
    dev, ctx = unet.device, unet.ctx
    first = next(iter(train))["eeg"].to(dev)
    z = stage1.encode_stage_2_inputs(first)
    # train_ldm.py:203-204 (unbiased std of one batch).  The reference is ONE process: one value for all replicas.  Here the loader is
    # rank-sharded, so every rank sees a different first batch -- rank 0's value is broadcast (and is the one checkpointed below)
    scale_factor = D.broadcast_scalar(1.0 / float(z.std()), src=0, like=z)
    if rank == 0:
        print(f"Scaling factor set to {scale_factor}")
    loss = torch.zeros(1, device=dev)
    gsync = D.OverlappedGradSync(unet.flat_grad, ctx=unet.ctx, comm=D.make_comm(unet.ctx))   # no-op with one process; EEGLDM_NATIVE_COLLECTIVES=1: RCCL through the C ABI
    steps, t0, seen, best, start_epoch, gstep = 0, time.time(), 0, float("inf"), 0, 0      # gstep: steps over all invocations (RNG offsets)
    if resume:
        # continue from {run_dir}/checkpoint.pth (keys as written below = training.py:381-387).  The reference computes `resume`
        # but always restarts at epoch 0 (train_ldm.py:113,210-211); here the run really continues, with the saved scale_factor
        ck = torch.load(os.path.join(run_dir, "checkpoint.pth"), map_location="cpu")
        unet.load_state_dict(ck["diffusion"]); opt.load_state_dict(ck["optimizer"])
        if "scaler" in ck:
            scaler.load_state_dict(ck["scaler"])
        start_epoch, best, scale_factor = int(ck["epoch"]), float(ck["best_loss"]), float(ck["scale_factor"])
        gstep = int(ck.get("steps", 0))
        if rank == 0:
            print(f"Resuming from epoch {start_epoch} (best loss {best:.5f}, scale factor {scale_factor})")
    for epoch in range(start_epoch, config.train.n_epochs):
        unet.train()
        for batch in train:
            x = batch["eeg"].to(dev)
            B = x.shape[0]
            t = randint(ctx, B, sched.num_train_timesteps, seed=s_t, offset=gstep * B)
            eps = randn(ctx, (B, args.latent_channels, x.shape[2] // stage1.down), seed=s_eps, offset=gstep * z[0].numel() * B)
            noise = randn(ctx, eps.shape, seed=s_noise, offset=gstep * z[0].numel() * B)
            e = stage1.encode_stage_2_inputs(x, eps=eps, scale_factor=scale_factor)
            opt.zero_grad()
            ldm_train_step(unet, sched, e, noise, t, loss_out=loss, grad_scale=scaler.get_scale(), grad_sync=gsync)
            gsync.wait()
            scaler.step(opt); scaler.update()
            steps += 1; gstep += 1; seen += B * world
            if args.max_steps and steps >= args.max_steps:
                break
        do_eval = (epoch + 1) % config.train.get("eval_freq", 1) == 0 or bool(args.max_steps and steps >= args.max_steps)
        cur = float(loss)
        if do_eval and valid is not None:      # model selection on the validation split (training.py:356-380), epsilon MSE over its windows
            v_sum, v_n = D.allreduce_sum_scalars(validate(unet, stage1, sched, valid, scale_factor, v_seeds, args.latent_channels), like=loss)
            cur = v_sum / max(1.0, v_n)
        if rank == 0:
            print(f"epoch {epoch}: loss {float(loss):.5f} | {seen/(time.time()-t0):.1f} windows/s", flush=True)
            if do_eval:
                if cur <= best:
                    best = cur
                    torch.save({k: v.cpu() for k, v in unet.state_dict().items()}, os.path.join(run_dir, "best_model.pth"))
                torch.save({"epoch": epoch + 1, "diffusion": {k: v.cpu() for k, v in unet.state_dict().items()}, "optimizer": opt.state_dict(),
                            "best_loss": best, "scale_factor": torch.tensor(scale_factor), "scaler": scaler.state_dict(), "steps": gstep},
                           os.path.join(run_dir, "checkpoint.pth"))
        if args.max_steps and steps >= args.max_steps:
            break
    if rank == 0:
        torch.save({k: v.cpu() for k, v in unet.state_dict().items()}, os.path.join(run_dir, "final_model.pth"))
    LAST_RUN.clear(); LAST_RUN.update(rank=rank, world=world, scale_factor=scale_factor, steps=steps, param_sum=float(unet.flat.double().sum()))
    return run_dir


if __name__ == "__main__":
    main(parse_args())
