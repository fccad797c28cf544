"""Headline benchmark: 30-s EEG windows/sec on the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N = 1 runs in this process; N > 1 without a torch.distributed.run environment re-executes itself as
        `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
        (one rank per GPU over RCCL); launched under torch.distributed.run directly it reads RANK / LOCAL_RANK / WORLD_SIZE.

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
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

UNET_CFG = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2,
                attention_resolutions=[8, 4], channel_mult=[1, 2, 4], resblock_updown=True)   # config_ldm.yaml:30-43, latent_channels=1
ROOFLINE_KERNEL_CLASS = "conv_wgrad_splitk_gemm"      # see main(): the class of the fused 3-tap weight-gradient kernel, the largest single kernel of the step
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}     # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                                  # HBM3E spec peak (same guide; ~6.3 TB/s is what a streaming kernel reaches)
PKG = os.path.join(ROOT, "synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd")


def kernel_source_hash():
    """sha256 over csrc/*.hip|*.h: the PMC summaries under profiles/ carry the hash of the build they were collected on, so a
    stale counter file is flagged instead of silently reported."""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def load_pmc(name):
    """profiles/<name> -> (dict, stale flag) or (None, None)."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
        return j, (j.get("kernel_source_sha16") != kernel_source_hash())
    except Exception:
        return None, None


# ---------------------------------------------------------------- algorithmic byte model
def aekl_gan_step_bytes(num_channels, L, s, n_res=2, d_ch=64, d_layers=3, fused=False):
    """Algorithmic HBM bytes of ONE window through the AEKL/GAN train step, by the survey's layer-granular formula
    bytes_fwd = sum_conv (|in| + |out|) s + sum_norm/act 2 |in| s (+ parameters, negligible per window);
    bytes_step = 3 x bytes_fwd of everything that runs forward AND backward: the autoencoder once, the discriminator three
    times (generator pass, fake, real).  s = storage bytes per activation element.
    fused=True: the FUSED FLOOR -- every normalisation / activation rides on a convolution's operand load or store (GroupNorm / BatchNorm
    statistics from the producing conv's epilogue, apply + SiLU / LeakyReLU on the consumer's load), so only the convolutions' own
    |in| + |out| remain.  The layer-granular figure bills every BatchNorm / LeakyReLU pass as necessary and flatters the fraction."""
    na = 0 if fused else 1          # norm / activation passes counted?
    def res(c_in, c_out, l):
        e = na * 2 * c_in * l + (c_in + c_out) * l + na * 2 * c_out * l + 2 * c_out * l      # norm1, conv1, norm2, conv2
        return e + ((c_in + c_out) * l if c_in != c_out else 0)
    nc = list(num_channels)
    e, l, c = (1 + nc[0]) * L, L, nc[0]                                              # encoder conv_in
    for i, co in enumerate(nc):
        for _ in range(n_res):
            e += res(c, co, l); c = co
        if i != len(nc) - 1:
            e += c * l + c * l // 2; l //= 2
    e += na * 2 * c * l + (c + 1) * l + 2 * 2 * l                                    # norm, conv -> latent, mu / log-sigma heads
    e += 2 * l + (1 + c) * l                                                         # decoder: post_quant, conv_in
    for i, co in enumerate(reversed(nc)):
        for _ in range(n_res):
            e += res(c, co, l); c = co
        if i != len(nc) - 1:
            e += na * (c * l + c * 2 * l) + (c * 2 * l + c * 2 * l if fused else 2 * c * 2 * l); l *= 2   # nearest x2 (+ its own pass when unfused) + conv
    e += na * 2 * c * l + (c + 1) * l
    ae = e
    d, l, c = L + d_ch * L // 2 + na * 2 * d_ch * L // 2, L // 2, d_ch              # conv s2 + LeakyReLU
    for j in range(d_layers):
        st = 1 if j == d_layers - 1 else 2
        d += c * l + 2 * c * (l // st) + na * 2 * 2 * c * (l // st); l //= st; c *= 2     # conv + BatchNorm/LeakyReLU
    d += (c + 1) * l
    return 3 * s * (ae + 3 * d)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="windows per GPU per step (C4: 2048 / 8 GPUs)")
    ap.add_argument("--length", type=int, default=768, help="latent length (3072 / 4)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"],
                    help="storage / operand type; f16 = the reference's autocast dtype, run with its GradScaler (training.py:334,441-443) inside the timed step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parts", action="store_true", help="skip the secondary AEKL-GAN / DDIM-50 measurements")
    return ap.parse_args()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or (os.cpu_count() or 2) // 2
    except Exception:
        return max(1, (os.cpu_count() or 2) // 2)


def _cpu_quota_cores():
    """CPUs' worth of time the container may use per period (cgroup v2 cpu.max / v1 cfs quota); None = unlimited or unknown.  The MI355X
    leases of this pool show all 256 logical CPUs of a 2 x 64-core EPYC 9575F but carry `cpu.max = 1600000 100000`: 16 CPUs -- which is why
    every sweep point above 16 threads gets SLOWER there (throttling), pinned or not (tools/r06/cpu_sweep.py, gpurun_out/r06_cpu)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def _first_physical_cpus(n):
    """logical ids of the first n physical cores in (package, core) order, one hardware thread each"""
    seen, order = set(), []
    base = "/sys/devices/system/cpu"
    try:
        cpus = sorted(int(d[3:]) for d in os.listdir(base) if d.startswith("cpu") and d[3:].isdigit())
    except OSError:
        return []
    for c in cpus:
        try:
            key = (int(open(f"{base}/cpu{c}/topology/physical_package_id").read()), int(open(f"{base}/cpu{c}/topology/core_id").read()))
        except (OSError, ValueError):
            continue
        if key not in seen:
            seen.add(key); order.append(key + (c,))
    order.sort()
    allowed = os.sched_getaffinity(0)
    return [c for _, _, c in order if c in allowed][:n]


def cpu_pinned_child(batch, nthreads, seconds):
    """`python bench.py --cpu-child B NT S` (started by cpu_baseline with OMP_NUM_THREADS / OMP_PROC_BIND=close / OMP_PLACES=cores): the oracle's
    LDM train step at batch B with torch's intra-op pool bound to the first NT physical cores -- the pool has to be bound BEFORE torch creates
    it, hence a process of its own.  Prints one JSON line."""
    ids = _first_physical_cpus(nthreads)
    if ids:
        os.sched_setaffinity(0, ids)
    import torch
    torch.set_num_threads(nthreads)
    from oracle import losses as Ls
    from oracle import steps as S
    from oracle import unet as U
    from param_gen import gen_param, normal, timesteps
    cfg = dict(UNET_CFG)
    sd = {k: torch.from_numpy(gen_param(42, k, s)) for k, s in U.unet_param_shapes(cfg).items()}
    acp = Ls.alphas_cumprod("linear_beta", 1000, 0.0015, 0.0195)
    lat, nz = torch.from_numpy(normal((batch, 1, 768), seed=1)), torch.from_numpy(normal((batch, 1, 768), seed=2))
    t = torch.from_numpy(timesteps(batch, seed=3))
    st = {"sd": dict(sd), "opt": {}, "i": 0}
    def f():
        _l, grads, _ = S.ldm_train_step(st["sd"], cfg, acp, lat, nz, t)
        st["i"] += 1
        st["sd"] = S.adam_update(st["sd"], grads, st["opt"], 1e-4, st["i"])
    f(); f()
    ts, t_end = [], time.time() + seconds
    while len(ts) < 10 and (time.time() < t_end or len(ts) < 2):
        t0 = time.time(); f(); ts.append(time.time() - t0)
    print(json.dumps({"windows_per_s": round(batch / sorted(ts)[len(ts) // 2], 3), "batch": batch, "threads": nthreads, "steps_timed": len(ts),
                      "pinned_cpus": len(ids)}))


def cpu_baseline(budget_s=60.0):
    """The oracle (torch CPU fp32: same math as the engine, pinned to the reference goldens) on this host's cores, protocol of
    BASELINE.md 3 / SURVEY 8d: 3 warm-up + 10 timed steps, median.  Thread count: a short sweep over {8, 16, 32} (never more than 32
    threads, never more than the physical cores; it stops as soon as a point is 1.3x slower than the best so far) -- on the driver's
    128-core EPYC 9575F the round-2 sweep also visited 64 and 128 threads, where torch's intra-op pool runs this model 10-20x slower
    (1.8 windows/s), ate the whole budget and left the three secondary legs skipped.  Every leg has its OWN time slice (budget_s is
    split: sweep 20 %, headline 25 %, the three legs the rest, each bounded by what is left), a leg that would overrun reports the
    steps it managed instead of "skipped", and the secondary legs run right after the headline with the chosen thread count.
    Headline leg = the LDM train step (B = 8); legs: AEKL/GAN step C1 (B = 32), pixel-space DM step (B = 2, L = 3072), DDIM-50 + decode."""
    import torch
    from oracle import aekl as A
    from oracle import losses as Ls
    from oracle import steps as S
    from oracle import unet as U
    from param_gen import eeg_windows, gen_param, normal, timesteps
    t_start = time.time()
    phys = _physical_cores()
    cfg = dict(UNET_CFG)
    sd = {k: torch.from_numpy(gen_param(42, k, s)) for k, s in U.unet_param_shapes(cfg).items()}
    acp = Ls.alphas_cumprod("linear_beta", 1000, 0.0015, 0.0195)

    def ldm_step_fn(batch, length):
        lat, nz = torch.from_numpy(normal((batch, 1, length), seed=1)), torch.from_numpy(normal((batch, 1, length), seed=2))
        t = torch.from_numpy(timesteps(batch, seed=3))
        c = dict(cfg, image_size=length)
        st = {"sd": dict(sd), "opt": {}, "i": 0}
        def f():
            _l, grads, _ = S.ldm_train_step(st["sd"], c, acp, lat, nz, t)
            st["i"] += 1
            st["sd"] = S.adam_update(st["sd"], grads, st["opt"], 1e-4, st["i"])
        return f

    def timed(f, warm, n, deadline=None):
        """median of up to n timed calls after `warm` untimed ones; stops early (keeping what it has, at least one) at `deadline`"""
        for _ in range(warm):
            f()
            if deadline is not None and time.time() > deadline:
                break
        ts = []
        for _ in range(n):
            t0 = time.time(); f(); ts.append(time.time() - t0)
            if deadline is not None and time.time() > deadline:
                break
        return sorted(ts)[len(ts) // 2], ts

    f8 = ldm_step_fn(8, 768)
    torch.set_num_threads(min(8, max(1, phys)))
    f8()                       # first-touch / allocator warm-up outside the sweep
    setup_s = time.time() - t_start
    t_sweep = time.time()
    sweep, best_rate = {}, 0.0
    quota = _cpu_quota_cores()
    for nt in sorted({n for n in (8, 16, 32) if n <= max(phys, 8) and (quota is None or n <= max(quota, 8))}):
        torch.set_num_threads(nt)
        rate = 8 / timed(f8, 1, 2, deadline=t_sweep + 0.2 * budget_s)[0]
        sweep[nt] = round(rate, 2)
        if rate < best_rate / 1.3 or time.time() - t_sweep > 0.2 * budget_s:
            break              # past the knee (or out of sweep time): more threads only get slower on this host
        best_rate = max(best_rate, rate)
    best = max(sweep, key=sweep.get)
    torch.set_num_threads(best)
    med, ts_head = timed(f8, 3, 10, deadline=time.time() + 0.25 * budget_s)
    out = {"value": round(8 / med, 3), "unit": "windows/s", "cores": best, "kind": "port", "cpu_model": _cpu_model(), "physical_cores": phys,
           "thread_sweep_windows_per_s": sweep, "steps_timed": len(ts_head),
           "sample": "oracle LDM train step (config_ldm UNet fwd+bwd+Adam, fp32), batch 8 x (1,768), 3 warm-up + 10 timed steps, median",
           "legs": {}}
    out["cpu_quota_cores"] = quota
    # The same step at B = 32 in a process whose OpenMP pool is BOUND to the first `best` physical cores (the bound,
    # larger-batch point is the fastest this host offers: 52 against 38-42 windows/s on the 16-CPU leases, tools/r06/cpu_sweep.py).
    # `value` is the better of the two; both stay in the record.
    out["in_process_b8_windows_per_s"] = out["value"]
    try:
        env = dict(os.environ, OMP_NUM_THREADS=str(best), MKL_NUM_THREADS=str(best), OMP_PROC_BIND="close", OMP_PLACES="cores")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-child", "32", str(best), str(round(0.2 * budget_s, 1))], env=env,
                           capture_output=True, text=True, timeout=max(60.0, budget_s))
        pl = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if pl:
            pinned = json.loads(pl[-1])
            out["pinned_b32"] = pinned
            if pinned["windows_per_s"] > out["value"]:
                out["value"] = pinned["windows_per_s"]; out["steps_timed"] = pinned["steps_timed"]
                out["sample"] = (f"oracle LDM train step (config_ldm UNet fwd+bwd+Adam, fp32), batch 32 x (1,768), OpenMP pool of {best} threads bound to "
                                 f"{pinned['pinned_cpus']} physical cores (own process), 2 warm-up + {pinned['steps_timed']} timed steps, median")
    except Exception as e:      # the in-process number stands
        out["pinned_b32"] = {"error": repr(e)[:200]}
    legs = out["legs"]
    t_legs = time.time()
    leg_budget = max(15.0, budget_s - (t_legs - t_start) + setup_s)      # parameter generation is not charged to the legs

    def leg(name, fn, batch, warm, n, note, share):
        m, ts = timed(fn, warm, n, deadline=time.time() + share * leg_budget)
        legs[name] = {"windows_per_s": round(batch / m, 3), "batch": batch, "steps_timed": len(ts), "sample": note}
        out[name + "_windows_per_s"] = legs[name]["windows_per_s"]      # flat copy: the driver's record keeps scalar keys only

    # C1: AEKL GAN step, channels [32,32,64], B = 32 (config_aekl_eeg.yaml)
    acfg = dict(num_channels=[32, 32, 64], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    dcfg = dict(spatial_dims=1, num_layers_d=3, num_channels=64, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH", bias=False, padding=1)
    st = {"ae": {k: torch.from_numpy(gen_param(42, k, s)) for k, s in A.aekl_param_shapes(acfg).items()},
          "d": {k: torch.from_numpy(gen_param(43, k, s)) for k, s in A.disc_param_shapes(dcfg).items()}, "og": {}, "od": {}, "i": 0}
    xw = torch.from_numpy(eeg_windows(32, seed=1234)); ew = torch.from_numpy(normal((32, 1, 768), seed=1236))
    def aekl_fn():
        st["i"] += 1
        _l, st["ae"], st["d"], _r, _g, _d = S.aekl_train_step(st["ae"], acfg, st["d"], dcfg, xw, ew, 0.01, 1e-6, 1e4, True, 5e-3, 5e-4, st["i"], st["og"], st["od"])
    leg("aekl_gan_train_step_c1", aekl_fn, 32, 2, 5, "oracle AEKL [32,32,64] + PatchDiscriminator GAN step incl. spectral loss and both Adam updates, B = 32", 0.25)
    # C5: pixel-space DM step, B = 2, L = 3072
    leg("pixel_dm_train_step", ldm_step_fn(2, 3072), 2, 1, 5, "oracle UNet train step on (2,1,3072), epsilon MSE + Adam", 0.3)
    # C3: DDIM-50 + decode, B = 8 (one run = 50 UNet forwards; B = 2 if the headline says that B = 8 would not fit what is left)
    left = leg_budget - (time.time() - t_legs)
    bd = 8 if 50 * 0.4 * med < max(left, 10.0) else 2
    def ddim_fn():
        S.ddim_sample(sd, cfg, st["ae"], acfg, torch.from_numpy(normal((bd, 1, 768), seed=5)), 50, Ls.alphas_cumprod("scaled_linear_beta", 1000, 0.0015, 0.0205))
    leg("ddim50_sample_decode", ddim_fn, bd, 0, 1, f"oracle DDIM-50 (50 UNet forwards) + decode [32,32,64], B = {bd}, one run", 1.0)
    out["seconds_spent"] = round(time.time() - t_start, 1)
    return out


def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no rendezvous environment: start N ranks through torch.distributed.run."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0)); port = so.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn_under_torchrun(args))
    import torch
    import eegldm
    from eegldm import distributed as D
    from eegldm.models import UNetModel, AutoencoderKL, PatchDiscriminator
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, GradScaler, ldm_train_step, aekl_train_step, randint, randn
    from eegldm.sampling import ddim_sample, make_sampling_scheduler
    from param_gen import eeg_windows

    rank, local, world = D.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local)
    ctx = eegldm.default_context(local)
    dev = torch.device("cuda", local)
    B, L = args.batch, args.length
    dtype = {"bf16": "bfloat16", "f16": "float16", "f32": "float32"}[args.dtype]

    unet = UNetModel(**UNET_CFG, dtype=dtype, device=local)
    g = torch.Generator().manual_seed(42)
    sd = unet.state_dict()
    # random-init weights of the named architecture; zero-initialised layers get N(0, 0.02) so no work is trivially zero
    unet.load_state_dict({k: (torch.randn(v.shape, generator=g) * 0.02 if float(v.abs().sum()) == 0.0 else v) for k, v in sd.items()})
    D.broadcast_flat(unet.flat); unet.sync_weights()
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195, device=local)   # train_ldm.py:199-200
    opt = Adam(unet, lr=1e-4)
    scaler = GradScaler(enabled=(args.dtype == "f16"))      # fp16 storage: loss scaling, finite check and skip logic are part of the step
    loss = torch.zeros(1, device=dev)
    # frozen stage-1 autoencoder (production channels [32,32,64], latent 1: clusters/run_aekl_shhs_1.sh:8-10)
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[32, 32, 64], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, attention_levels=[False, False, False], dtype=dtype, device=local)
    D.broadcast_flat(ae.flat); ae.sync_weights()
    # synthetic 30-s windows (SURVEY 8d recipe), resident in HBM before the timed region; per-rank stream
    windows = torch.from_numpy(eeg_windows(B, seed=1234 + rank, length=4 * L)).to(dev)
    # train_ldm.py:203-204: ONE scale factor for all replicas -- rank 0's first batch, broadcast (each rank has its own window stream)
    scale_factor = D.broadcast_scalar(1.0 / float(ae.encode_stage_2_inputs(windows, eps=randn(ctx, (B, 1, L), seed=99)).std()), src=0, like=windows)

    gsync = D.OverlappedGradSync(unet.flat_grad, ctx=ctx, comm=D.make_comm(ctx)) if world > 1 else None   # EEGLDM_NATIVE_COLLECTIVES=1: eegldm_comm_* (RCCL via the C ABI)

    def step(i, sync=True):
        t = randint(ctx, B, 1000, seed=1235 + rank, offset=i * B)
        noise = randn(ctx, (B, 1, L), seed=1236 + rank, offset=i * B * L)
        eps = randn(ctx, (B, 1, L), seed=1237 + rank, offset=i * B * L)
        latents = ae.encode_stage_2_inputs(windows, eps=eps, scale_factor=scale_factor)
        unet.zero_grad()
        # N > 1: the all-reduce of out / output_blocks / middle_block gradients starts inside the native backward (grad hook)
        # and overlaps the input blocks' backward; the rest follows the call
        gs = gsync if sync else None          # the rank-0-only profiling leg below must not enter a collective
        ldm_train_step(unet, sched, latents, noise, t, loss_out=loss, grad_scale=scaler.get_scale(), grad_sync=gs)
        if gs is not None:
            gs.wait()
        scaler.step(opt); scaler.update()      # disabled scaler: plain opt.step()

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

    # ---- N > 1: make the record self-checking -- how many ranks the collective really spans, and what the exchange costs
    comm_info = None
    if world > 1:
        import torch.distributed as dist
        n_grad = int(unet.flat_grad.numel())
        probe = torch.ones(1, device=dev)
        dist.all_reduce(probe)                                  # = number of ranks that took part in a real collective
        # (a) the bare exchange: bucketed all-reduce-mean of the 122 MB flat gradient buffer, nothing else running
        gbuf = torch.zeros_like(unet.flat_grad)
        def bare():
            if gsync is not None and gsync.comm is not None:
                gsync.comm.allreduce_mean(gbuf); gsync.comm.wait()
            else:
                D.allreduce_mean_flat(gbuf)
        bare(); dist.barrier(); torch.cuda.synchronize(); t1 = time.time()
        for _ in range(5):
            bare()
        torch.cuda.synchronize(); ar = torch.tensor([(time.time() - t1) / 5], device=dev)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        # (b) the same step without the exchange (rank-local): timed - this = what of the exchange is NOT hidden behind the backward
        n_loc = max(3, min(10, args.steps))
        torch.cuda.synchronize(); t1 = time.time()
        for i in range(n_loc):
            step(args.warmup + args.steps + i, sync=False)
        torch.cuda.synchronize(); loc = torch.tensor([(time.time() - t1) / n_loc], device=dev)
        dist.all_reduce(loc, op=dist.ReduceOp.MAX)
        D.broadcast_flat(unet.flat); unet.sync_weights()        # the un-synchronised steps let the replicas drift: realign before anything else
        dist.barrier()
        ms_step = 1e3 * elapsed / args.steps
        # self-diagnosis of the first real multi-GPU run: with one visible GPU per rank the collectives MUST be RCCL ("nccl" on ROCm).
        # gloo is legal only under the single-GPU test hook (EEGLDM_LOCAL_DEVICE: several ranks share one device).
        backend = dist.get_backend()
        if torch.cuda.device_count() >= world and "EEGLDM_LOCAL_DEVICE" not in os.environ:
            assert backend == "nccl", f"{world} ranks on {torch.cuda.device_count()} visible GPUs must use RCCL, got backend {backend!r}"
        assert int(round(float(probe))) == world, f"the probe all-reduce spanned {float(probe)} ranks, expected {world}"
        comm_info = {"backend": backend, "collective_ranks": int(round(float(probe))), "collective_is_rccl": backend == "nccl",
                     "world_size": dist.get_world_size(),
                     "native_collectives": bool(gsync is not None and gsync.comm is not None),
                     "native_comm_world": (int(eegldm._lib.lib.eegldm_comm_world(gsync.comm.h)) if gsync is not None and gsync.comm is not None else None),
                     "grad_bytes_per_rank": 4 * n_grad, "bucket_bytes": 4 * D.BUCKET_ELEMS,
                     "allreduce_ms_bare": round(1e3 * float(ar), 3),
                     "allreduce_busbw_GBs": round(2 * (world - 1) / world * 4 * n_grad / float(ar) / 1e9, 1),
                     "compute_only_ms_per_step": round(1e3 * float(loc), 3),
                     "exposed_comm_ms_per_step": round(ms_step - 1e3 * float(loc), 3)}

    roofline = None
    if rank == 0 and not args.no_roofline:
        ctx.prof_enable(True)
        ROOF_STEPS = 6      # 252 launches of the dominant class: two steps were too few to average out a transient (one run read 650 TF/s where the
        for i in range(ROOF_STEPS):      # rocprof trace of the same box gave 760)
            step(args.warmup + args.steps + i, sync=False)
        torch.cuda.synchronize()
        summ = ctx.prof_summary()
        ctx.prof_enable(False)
        # `roofline.kernel` is FIXED, not the class that happens to win this run: the single kernel with the largest total time in the committed
        # rocprof table of this command (profiles/r06_ldm_step_bf16_B256_kernel_stats_v1.txt, r05_..._v5.txt: the fused 3-tap weight gradient
        # gemm_kernel<u16, 2, 1, 3, 2, 64, 1, 2, true>, 14-15 % of the step) = the class conv_wgrad_splitk_gemm.  The forward big-tile class is within
        # 1 % of it in total time and runs at a higher rate; picking "whichever is larger today" made the headline fraction flip between rounds.
        k = ROOFLINE_KERNEL_CLASS if summ.get(ROOFLINE_KERNEL_CLASS, {}).get("launches") else max(summ.items(), key=lambda kv: kv[1]["ms"])[0]
        v = summ[k]
        runner_up = max(((kk, vv) for kk, vv in summ.items() if kk != k), key=lambda kv: kv[1]["ms"])[0]
        ach = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0.0
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # HBM bytes per launch of this kernel class and its MFMA-busy share come from separate rocprofv3 --pmc passes over this same
        # step (tools/pmc_traffic.py / tools/pmc_collect.sh -> profiles/r02_pmc_*.json; FETCH_SIZE doubled per the gfx950 note in the
        # guide).  Counters cannot be read from inside the timed process, so the files carry the hash of the kernel sources they
        # were collected on and `traffic_stale` says whether that is still the build being timed.
        traffic, traffic_src, traffic_stale, mfma_util, step_bytes = None, None, None, None, None
        if args.dtype == "bf16" and args.batch == 256:
            for fname in ("r06_pmc_hbm_traffic.json", "r05_pmc_hbm_traffic.json", "r04_pmc_hbm_traffic.json", "r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
                pj, stale = load_pmc(fname)
                if pj and k in pj.get("classes", {}):
                    traffic = round(pj["classes"][k]["hbm_bytes_per_launch"]); traffic_stale = bool(stale)
                    traffic_src = f"profiles/{fname} (bytes per launch, B=256 bf16)"
                    step_bytes = pj.get("hbm_bytes_per_step")
                    if step_bytes is None:      # files written before round 6: sum of (bytes per launch x launches sampled) / steps, steps = Adam launches
                        fams = dict(pj.get("classes", {})); fams.update(pj.get("hbm_bound_families", {}))
                        nst = max(1, fams.get("adam", {}).get("launches_sampled", 1))
                        step_bytes = sum(f_["hbm_bytes_per_launch"] * f_["launches_sampled"] for kk, f_ in fams.items()
                                         if kk not in ("__amd_rocclr_copyBuffer", "vectorized_elementwise", "__amd_rocclr_fillBufferAligned")) / nst
                    break
            for fname in ("r06_pmc_mfma_busy.json", "r05_pmc_mfma_busy.json", "r04_pmc_mfma_busy.json", "r03_pmc_mfma_busy.json", "r02_pmc_mfma_busy.json", "r01_pmc_mfma_busy.json"):
                pm, _st = load_pmc(fname)
                if pm and k in pm.get("classes", {}):
                    mfma_util = pm["classes"][k]["MfmaUtil_pct"]
                    break
        roofline = {"bound": "mfma", "kernel": k, "kernel_choice": "fixed: class of the single kernel with the largest total time in the committed rocprof table",
                    "runner_up_class": runner_up, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                    "traffic": traffic, "_step_bytes": step_bytes, "traffic_source": traffic_src, "traffic_stale": traffic_stale, "mfma_busy_pct_pmc": mfma_util,
                    "event_bracket_overhead_us_subtracted": round(ctx.prof_bracket_overhead_us(), 2), "launches_per_step": v["launches"] // ROOF_STEPS, "steps_measured": ROOF_STEPS, "avg_launch_us": round(1e3 * v["ms"] / max(1, v["launches"]), 2),
                    "gflop_per_launch": round(v["flops"] / max(1, v["launches"]) / 1e9, 3),
                    "all_gemm_classes": {kk: {"tflops": round(vv["flops"] / (vv["ms"] * 1e-3) / 1e12, 1) if vv["ms"] > 0 else 0.0,
                                              "ms_per_step": round(vv["ms"] / ROOF_STEPS, 3), "launches_per_step": vv["launches"] // ROOF_STEPS}
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
        esz = 2 if args.dtype in ("bf16", "f16") else 4
        abytes = aekl_gan_step_bytes([2, 2, 4], 4 * L, esz) * Ba + 16 * (int(ae2.flat.numel()) + int(disc.flat.numel()))
        hbm_ach = abytes / dtg / 1e9
        fbytes = aekl_gan_step_bytes([2, 2, 4], 4 * L, esz, fused=True) * Ba + 16 * (int(ae2.flat.numel()) + int(disc.flat.numel()))
        fused_ach = fbytes / dtg / 1e9
        pj, stale = None, None
        for aekl_pmc_name in ("r06_pmc_aekl_step.json", "r05_pmc_aekl_step.json", "r04_pmc_aekl_step.json", "r03_pmc_aekl_step.json", "r02_pmc_aekl_step.json"):
            pj, stale = load_pmc(aekl_pmc_name)
            if pj is not None:
                break
        parts["aekl_gan_train_step"] = {"windows_per_s": round(Ba / dtg, 1), "ms_per_step": round(1e3 * dtg, 3), "batch": Ba,
                                        "roofline": {"bound": "hbm", "achieved": round(hbm_ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                     "frac": round(hbm_ach / HBM_PEAK_GBS, 4),
                                                     "algorithmic_bytes_per_step": int(abytes), "algorithmic_MB_per_window": round(abytes / Ba / 1e6, 2),
                                                     "fused_floor": {"algorithmic_bytes_per_step": int(fbytes), "algorithmic_MB_per_window": round(fbytes / Ba / 1e6, 2),
                                                                     "achieved": round(fused_ach, 1), "frac": round(fused_ach / HBM_PEAK_GBS, 4),
                                                                     "model": "only the convolutions' |in| + |out| (norm / activation passes fused away), x3 for fwd + bwd, D three times"},
                                                     "traffic": (round(pj["hbm_bytes_per_step"]) if pj else None),
                                                     "traffic_source": (f"profiles/{aekl_pmc_name} (FETCH_SIZE x2 + WRITE_SIZE summed over the step's kernels)" if pj else None),
                                                     "traffic_stale": (bool(stale) if pj else None),
                                                     "flops_view": {"gflop_per_window": 3.76, "achieved_tflops": round(3.76e9 * Ba / dtg / 1e12, 1),
                                                                    "frac_of_mfma_peak": round(3.76e9 * Ba / dtg / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 4)}},
                                        "config": "config_aekl_eeg_2_2_4_spec.yaml: AutoencoderKL [2,2,4] + PatchDiscriminator(64, 3 layers) + "
                                                  "L1 + KL + LSGAN + spectral(1e4), Adam 5e-3 / 5e-4",
                                        "losses": [round(float(v), 5) for v in lo.cpu()]}
        Bs = min(args.batch, 256)
        sch = make_sampling_scheduler(50, device=local)
        nz = randn(ctx, (Bs, 1, L), seed=4242)
        info = {}
        ddim_sample(unet, ae, sch, nz, scale_factor=scale_factor, info=info)      # warm-up (captures the hipGraph of the UNet forward for this batch)
        torch.cuda.synchronize(); t1 = time.time()
        out, _ = ddim_sample(unet, ae, sch, nz, scale_factor=scale_factor)
        torch.cuda.synchronize(); dts = time.time() - t1
        # the reference samples ONE window per call (sample_trials.py:149-163): latency of that mode, graph replay vs eager launches
        lat = {}
        for mode, ug in (("graph", True), ("eager", False)):
            ddim_sample(unet, ae, sch, nz[:1], scale_factor=scale_factor, use_graph=ug)
            torch.cuda.synchronize(); t1 = time.time()
            for _ in range(3):
                ddim_sample(unet, ae, sch, nz[:1], scale_factor=scale_factor, use_graph=ug)
            torch.cuda.synchronize(); lat[mode] = (time.time() - t1) / 3
        ddim_tf = 695.3e9 * Bs / dts / 1e12
        parts["ddim50_sample_decode"] = {"windows_per_s": round(Bs / dts, 1), "seconds": round(dts, 3), "batch": Bs, "steps": 50,
                                         "native_sampler": True, "hipgraph": bool(info.get("graph")),
                                         "roofline": {"bound": "mfma", "achieved": round(ddim_tf, 1), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                                      "frac": round(ddim_tf / MFMA_PEAK_TFLOPS[args.dtype], 4)},
                                         "batch1_latency_ms": {k2: round(1e3 * v2, 2) for k2, v2 in lat.items()},
                                         "batch1_windows_per_s": round(1.0 / min(lat.values()), 2), "batch1_default_mode": "eager",
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
        dm_tf = 183.0e9 * Bd / dtd / 1e12
        parts["pixel_dm_train_step"] = {"windows_per_s": round(Bd / dtd, 1), "ms_per_step": round(1e3 * dtd, 3), "batch": Bd,
                                        "roofline": {"bound": "mfma", "achieved": round(dm_tf, 1), "peak": MFMA_PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                                                     "frac": round(dm_tf / MFMA_PEAK_TFLOPS[args.dtype], 4)},
                                        "config": "config_dm.yaml UNet on raw (B,1,3072) windows (T=768 attention), epsilon MSE + 1e-6 x spectral, "
                                                  "Adam 1e-4; per-GPU batch 64 = global 512 / 8 [BASELINE configs[4]]",
                                        "final_loss": round(float(ldm), 5), "gflop_per_window": 183.0}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0 and roofline is not None:
        # flat copies of what the nested dicts hold (the driver's record keeps scalar keys of `roofline` / `cpu_baseline` only)
        step_tf = 41.9e9 * B * args.steps / elapsed / 1e12      # whole LDM step, per GPU
        roofline["ldm_step_tflops"] = round(step_tf, 1); roofline["ldm_step_frac_of_mfma_peak"] = round(step_tf / MFMA_PEAK_TFLOPS[args.dtype], 4)
        # the other side of the joint limit: HBM bytes of the WHOLE step (PMC FETCH x 2 + WRITE over every kernel of a step, tools/pmc_traffic.py)
        # against the timed step -> fraction of the 8 TB/s peak
        sb = roofline.pop("_step_bytes", None)
        roofline["hbm_bytes_per_step"] = (round(sb) if sb else None)
        roofline["hbm_frac"] = (round(sb / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 4) if sb else None)
        for kk, vv in roofline.get("all_gemm_classes", {}).items():
            roofline[f"class_{kk}_tflops"] = vv["tflops"]
        if parts:
            for pk, short in (("aekl_gan_train_step", "aekl_gan_step"), ("ddim50_sample_decode", "ddim50"), ("pixel_dm_train_step", "pixel_dm_step")):
                if pk in parts:
                    roofline[f"{short}_{parts[pk]['roofline']['bound']}_frac"] = parts[pk]["roofline"]["frac"]
                    roofline[f"{short}_windows_per_s"] = parts[pk]["windows_per_s"]
            if "aekl_gan_train_step" in parts and "fused_floor" in parts["aekl_gan_train_step"]["roofline"]:
                roofline["aekl_gan_step_hbm_frac_fused_floor"] = parts["aekl_gan_train_step"]["roofline"]["fused_floor"]["frac"]
            if "ddim50_sample_decode" in parts:
                roofline["ddim50_batch1_latency_ms"] = min(parts["ddim50_sample_decode"]["batch1_latency_ms"].values())
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
            "roofline": roofline, "cpu_baseline": cpu, "parts": parts, "kernel_source_sha16": kernel_source_hash(),
        }
        if comm_info is not None:
            out["comm"] = comm_info
            out["collective_ranks"] = comm_info["collective_ranks"]; out["collective_backend"] = comm_info["backend"]
            out["allreduce_ms_per_step"] = comm_info["allreduce_ms_bare"]
            out["exposed_comm_ms_per_step"] = comm_info["exposed_comm_ms_per_step"]
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-child":
        cpu_pinned_child(int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
        sys.exit(0)
    main()
