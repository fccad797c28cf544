"""Headline benchmark: 30-s EEG windows/sec on the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

A "step" is one LDM train step of the reference loop (/root/reference/src/training/training.py:419-443)
on one synthetic batch of raw (B,1,3072) windows per GPU: draw timesteps + noise + eps on device, frozen
AutoencoderKL [32,32,64] encode + sampling x scale_factor, add_noise, UNet(config_ldm.yaml) forward, MSE,
hand-written backward, gradient all-reduce (N>1, RCCL), fused Adam, bf16 weight refresh.
Secondary measurements (same JSON line, `parts`): the AEKL/GAN train step of BASELINE configs[1]
(config_aekl_eeg_2_2_4_spec.yaml, spectral loss on, batch 256) and DDIM-50 sampling + decode.  Inputs are resident in HBM when the
timed region starts.  Prints ONE JSON line on rank 0; `roofline` is measured live with HIP events
around every launch of the dominant kernel class on the library's stream; `cpu_baseline` times the
oracle (torch CPU fp32, same math) on this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

UNET_CFG = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2,
                attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)   # config_ldm.yaml:30-43, latent_channels=1
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}     # dense, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="windows per GPU per step (C4: 2048 / 8 GPUs)")
    ap.add_argument("--length", type=int, default=768, help="latent length (3072 / 4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parts", action="store_true", help="skip the secondary AEKL-GAN / DDIM-50 measurements")
    return ap.parse_args()


def cpu_baseline(batch=8, length=768, steps=2):
    """Oracle LDM train step (fp32, torch CPU ops + autograd + torch.optim-style Adam) on a bounded sample."""
    import torch
    from oracle import losses as Ls
    from oracle import steps as S
    from oracle import unet as U
    from param_gen import gen_param, normal, timesteps
    cores = max(1, (os.cpu_count() or 2) // 2)
    torch.set_num_threads(cores)
    cfg = dict(UNET_CFG)
    sd = {k: torch.from_numpy(gen_param(42, k, s)) for k, s in U.unet_param_shapes(cfg).items()}
    acp = Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0195)
    lat, nz = torch.from_numpy(normal((batch, 1, length), seed=1)), torch.from_numpy(normal((batch, 1, length), seed=2))
    t = torch.from_numpy(timesteps(batch, seed=3))
    state = {}
    times = []
    for i in range(steps + 1):
        t0 = time.time()
        _loss, grads, _ = S.ldm_train_step(sd, cfg, acp, lat, nz, t)
        sd = S.adam_update(sd, grads, state, 1e-4, i + 1)
        times.append(time.time() - t0)
    dt = sorted(times[1:])[len(times[1:]) // 2]
    return {"value": batch / dt, "unit": "windows/s", "cores": cores, "kind": "port",
            "sample": f"oracle LDM train step (UNet config_ldm fwd+bwd+Adam, fp32), batch {batch} x (1,{length}), median of {steps} steps after 1 warm-up"}


def main():
    args = parse()
    import torch
    import eegldm
    from eegldm import distributed as D
    from eegldm.models import UNetModel, AutoencoderKL, PatchDiscriminator
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, ldm_train_step, aekl_train_step, randint, randn
    from eegldm.sampling import ddim_sample, make_sampling_scheduler
    from param_gen import eeg_windows

    rank, local, world = D.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    torch.cuda.set_device(local)
    ctx = eegldm.default_context(local)
    dev = torch.device("cuda", local)
    B, L = args.batch, args.length
    dtype = {"bf16": "bfloat16", "f32": "float32"}[args.dtype]

    unet = UNetModel(**UNET_CFG, dtype=dtype, device=local)
    g = torch.Generator().manual_seed(42)
    sd = unet.state_dict()
    # random-init weights of the named architecture; zero-initialised layers get N(0, 0.02) so no work is trivially zero
    unet.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
    D.broadcast_flat(unet.flat); unet.sync_weights()
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="scaled_linear_beta", beta_start=0.0015, beta_end=0.0195, device=local)
    opt = Adam(unet, lr=1e-4)
    loss = torch.zeros(1, device=dev)
    # frozen stage-1 autoencoder (production channels [32,32,64], latent 1: clusters/run_aekl_shhs_1.sh:8-10)
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=local)
    D.broadcast_flat(ae.flat); ae.sync_weights()
    # synthetic 30-s windows (SURVEY 8d recipe), resident in HBM before the timed region; per-rank stream
    windows = torch.from_numpy(eeg_windows(B, seed=1234 + rank, length=4 * L)).to(dev)
    scale_factor = 1.0 / float(ae.encode_stage_2_inputs(windows, eps=randn(ctx, (B, 1, L), seed=99)).std())   # train_ldm.py:203-204

    gsync = D.OverlappedGradSync(unet.flat_grad) if world > 1 else None

    def step(i, sync=True):
        t = randint(ctx, B, 1000, seed=1235 + rank, offset=i * B)
        noise = randn(ctx, (B, 1, L), seed=1236 + rank, offset=i * B * L)
        eps = randn(ctx, (B, 1, L), seed=1237 + rank, offset=i * B * L)
        latents = ae.encode_stage_2_inputs(windows, eps=eps, scale_factor=scale_factor)
        unet.zero_grad()
        # N > 1: the all-reduce of out / output_blocks / middle_block gradients starts inside the native backward (grad hook)
        # and overlaps the input blocks' backward; the rest follows the call
        gs = gsync if sync else None          # the rank-0-only profiling leg below must not enter a collective
        ldm_train_step(unet, sched, latents, noise, t, loss_out=loss, grad_sync=gs)
        if gs is not None:
            gs.wait()
        opt.step()

    for i in range(args.warmup):
        step(i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.time()
    for i in range(args.steps):
        step(args.warmup + i)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = torch.tensor([time.time() - t0], device=dev)
    if world > 1:
        torch.distributed.all_reduce(dt, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(dt)
    final_loss = float(loss)

    roofline = None
    if rank == 0 and not args.no_roofline:
        ctx.prof_enable(True)
        for i in range(2):
            step(args.warmup + args.steps + i, sync=False)
        torch.cuda.synchronize()
        summ = ctx.prof_summary()
        ctx.prof_enable(False)
        dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        k, v = dom
        ach = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch of this kernel class from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
        # separate runs of the same step, tools/pmc_traffic.py; FETCH_SIZE doubled per the gfx950 note in the guide)
        traffic, traffic_src = None, None
        try:
            pj = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic.json")
            if args.dtype == "bf16" and args.batch == 256:
                traffic = round(json.load(open(pj))["classes"][k]["hbm_bytes_per_launch"])
                traffic_src = "profiles/r01_pmc_hbm_traffic.json (bytes per launch, B=256 bf16)"
        except Exception:
            pass
        # MFMA-busy share of the same class from the committed SQ counter pass (profiles/r01_pmc_mfma_busy.json)
        mfma_util = None
        try:
            pm = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_mfma_busy.json")
            if args.dtype == "bf16" and args.batch == 256:
                mfma_util = json.load(open(pm))["classes"][k]["MfmaUtil_pct"]
        except Exception:
            pass
        roofline = {"bound": "mfma", "kernel": k, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "traffic_source": traffic_src, "mfma_busy_pct_pmc": mfma_util,
                    "event_bracket_overhead_us_subtracted": round(ctx.prof_bracket_overhead_us(), 2), "launches_per_step": v["launches"] // 2, "avg_launch_us": round(1e3 * v["ms"] / max(1, v["launches"]), 2),
                    "gflop_per_launch": round(v["flops"] / max(1, v["launches"]) / 1e9, 3),
                    "all_gemm_classes": {kk: {"tflops": round(vv["flops"] / (vv["ms"] * 1e-3) / 1e12, 1) if vv["ms"] > 0 else 0.0,
                                              "ms_per_step": round(vv["ms"] / 2, 3), "launches_per_step": vv["launches"] // 2}
                                         for kk, vv in summ.items() if vv["launches"]}}

    # ---- secondary workloads (rank 0 only; not part of `value`)
    parts = None
    if rank == 0 and world == 1 and not args.no_parts:      # secondary workloads only on the single-GPU run
        parts = {}
        Ba = args.batch
        ae2 = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[2, 2, 4], latent_channels=1, num_res_blocks=2,
                            norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=local)
        disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3,
                                  norm="BATCH", bias=False, padding=1, dtype=dtype, device=local)
        og, od = Adam(ae2, lr=5e-3), Adam(disc, lr=5e-4)
        xw = torch.from_numpy(eeg_windows(Ba, seed=77, length=4 * L)).to(dev)
        lo = torch.zeros(6, device=dev)

        def gan_step(i):
            eps = randn(ctx, (Ba, 1, L), seed=555, offset=i * Ba * L)
            ae2.zero_grad(); disc.zero_grad()
            aekl_train_step(ae2, disc, xw, eps, 0.01, 1e-9, 1e4, True, losses_out=lo)   # config_aekl_eeg.yaml:13-17 weights
            og.step(); od.step()

        for i in range(2):
            gan_step(i)
        torch.cuda.synchronize(); t1 = time.time()
        n_gan = max(3, args.steps // 2)
        for i in range(n_gan):
            gan_step(2 + i)
        torch.cuda.synchronize(); dtg = (time.time() - t1) / n_gan
        parts["aekl_gan_train_step"] = {"windows_per_s": round(Ba / dtg, 1), "ms_per_step": round(1e3 * dtg, 3), "batch": Ba,
                                        "config": "config_aekl_eeg_2_2_4_spec.yaml: AutoencoderKL [2,2,4] + PatchDiscriminator(64, 3 layers) + "
                                                  "L1 + KL + LSGAN + spectral(1e4), Adam 5e-3 / 5e-4",
                                        "losses": [round(float(v), 5) for v in lo.cpu()]}
        Bs = min(args.batch, 256)
        sch = make_sampling_scheduler(50, device=local)
        nz = randn(ctx, (Bs, 1, L), seed=4242)
        ddim_sample(unet, ae, sch, nz[:8], scale_factor=scale_factor)      # warm-up
        torch.cuda.synchronize(); t1 = time.time()
        out, _ = ddim_sample(unet, ae, sch, nz, scale_factor=scale_factor)
        torch.cuda.synchronize(); dts = time.time() - t1
        parts["ddim50_sample_decode"] = {"windows_per_s": round(Bs / dts, 1), "seconds": round(dts, 3), "batch": Bs, "steps": 50,
                                         "config": "config_ldm.yaml UNet, DDIM-50 (scaled-linear 0.0015-0.0205, eta 0), decode [32,32,64], crop to 3000",
                                         "out_shape": list(out.shape), "gflop_per_window": 695.3}

        # configs[4]: config_dm.yaml pixel-space model on raw (B,1,3072) windows (global 512 over 8 GPUs = 64 per GPU): long-sequence
        # convs and T = 768 attention (training_diffusion.py:133-158, spectral term on as in train_pure_ldm --spe spectral)
        from eegldm.training import dm_train_step
        from eegldm.schedulers import DDPMScheduler
        Bd, Ld = 64, 4 * L
        unet_dm = UNetModel(image_size=Ld, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
                            channel_mult=[1, 2, 4], resblock_updown=True, dtype=dtype, device=local)
        gz = torch.Generator().manual_seed(9)
        sdm = unet_dm.state_dict()
        unet_dm.load_state_dict({k: (torch.randn(v.shape, generator=gz) * 0.02 if float(v.abs().sum()) == 0 else v.cpu()) for k, v in sdm.items()})
        sch_dm = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195, device=local)
        opt_dm = Adam(unet_dm, lr=1e-4)
        xdm = xw[:Bd].contiguous()
        ldm = torch.zeros(1, device=dev)

        def dm_step(i):
            tdm = randint(ctx, Bd, 1000, seed=31, offset=i * Bd)
            ndm = randn(ctx, (Bd, 1, Ld), seed=32, offset=i * Bd * Ld)
            opt_dm.zero_grad()
            dm_train_step(unet_dm, sch_dm, xdm, ndm, tdm, spectral_weight=1e-6, spectral_loss=True, loss_out=ldm)
            opt_dm.step()

        for i in range(2):
            dm_step(i)
        torch.cuda.synchronize(); t1 = time.time()
        n_dm = max(3, args.steps // 2)
        for i in range(n_dm):
            dm_step(2 + i)
        torch.cuda.synchronize(); dtd = (time.time() - t1) / n_dm
        parts["pixel_dm_train_step"] = {"windows_per_s": round(Bd / dtd, 1), "ms_per_step": round(1e3 * dtd, 3), "batch": Bd,
                                        "config": "config_dm.yaml UNet on raw (B,1,3072) windows (T=768 attention), epsilon MSE + 1e-6 x spectral, "
                                                  "Adam 1e-4; per-GPU batch 64 = global 512 / 8 [BASELINE configs[4]]",
                                        "final_loss": round(float(ldm), 5), "gflop_per_window": 183.0}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        out = {
            "metric": "EEG windows/sec (LDM train step)", "value": round(world * B * args.steps / elapsed, 2), "unit": "windows/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "LDM train step on raw (B,1,3072) windows: frozen AutoencoderKL[32,32,64] encode+sample x scale, add_noise, "
                                   "config_ldm.yaml UNet (30.5M params, latent_channels=1) fwd+bwd, epsilon MSE, Adam lr 1e-4 "
                                   "[BASELINE configs[2]/[3]: per-GPU batch 256 = global 2048 / 8]",
                       "per_gpu_batch": B, "global_batch": world * B, "latent_len": L, "parallelism": f"dp{world}",
                       "final_loss": round(final_loss, 5), "gflop_per_window": 41.9},
            "roofline": roofline, "cpu_baseline": cpu, "parts": parts,
        }
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
