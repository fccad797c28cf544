"""-m gpu: the multi-process data-parallel path with the REAL native backward, two ranks sharing the one GPU of the test box
(gloo on device tensors; RCCL needs one GPU per rank, which the driver's multi-GPU node provides):
 * OverlappedGradSync driven by the native grad hook == plain post-backward bucketed all-reduce == the mean of the two ranks'
   single-process gradients (train_ldm.py:190-192 replaces nn.DataParallel with this);
 * `python bench.py --gpus 2` spawns its own ranks through torch.distributed.run and prints one valid JSON line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      EEGLDM_DIST_BACKEND="gloo", EEGLDM_LOCAL_DEVICE="0")
    import torch.distributed as dist
    from eegldm import distributed as D
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import ldm_train_step
    r, local, w = D.init_from_env()
    assert (r, local, w) == (rank, 0, world)
    torch.manual_seed(0)
    net = UNetModel(image_size=64, in_channels=1, out_channels=1, model_channels=32, num_res_blocks=1, attention_resolutions=[2],
                    channel_mult=[1, 2], resblock_updown=True, dtype="float32")
    sd = net.state_dict()
    torch.manual_seed(1 + rank)                               # different initial weights per rank: the broadcast must fix that
    net.load_state_dict({k: torch.randn(v.shape) * 0.05 for k, v in sd.items()})
    D.broadcast_flat(net.flat); net.sync_weights()
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
    dev = net.device

    def batch(rk):
        g = torch.Generator().manual_seed(100 + rk)
        return (torch.randn(4, 1, 64, generator=g).to(dev), torch.randn(4, 1, 64, generator=g).to(dev), torch.randint(0, 1000, (4,), generator=g).to(dev))

    lat, noise, t = batch(rank)
    # reference: this rank computes BOTH ranks' gradients alone and averages them
    want = torch.zeros_like(net.flat_grad)
    for rk in range(world):
        net.zero_grad(); ldm_train_step(net, sched, *batch(rk)); want += net.flat_grad / world
    net.zero_grad(); ldm_train_step(net, sched, lat, noise, t); D.allreduce_mean_flat(net.flat_grad, bucket_elems=50000); plain = net.flat_grad.clone()
    gs = D.OverlappedGradSync(net.flat_grad, bucket_elems=50000, ctx=net.ctx)
    net.zero_grad(); ldm_train_step(net, sched, lat, noise, t, grad_sync=gs); gs.wait()
    torch.cuda.synchronize()
    scale = float(want.abs().max())
    q.put((rank, float((plain - want).abs().max()) / scale, float((net.flat_grad - want).abs().max()) / scale, [list(x) for x in gs.done],
           float(net.flat.double().sum())))
    dist.destroy_process_group()


def test_overlapped_grad_sync_with_native_backward_two_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda v: v[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][4] == res[1][4]                              # parameters identical after the broadcast
    for rank, e_plain, e_overlap, done, _s in res:
        assert e_plain < 1e-5 and e_overlap < 1e-5, (rank, e_plain, e_overlap)
        assert len(done) == 1 and done[0][0] > 0              # the hook reported one tail slice (middle + output blocks)


def test_bench_self_spawns_two_ranks():
    """`python bench.py --gpus 2` (no rendezvous environment) -> torch.distributed.run with 2 ranks; both share GPU 0 over gloo here."""
    env = dict(os.environ, EEGLDM_DIST_BACKEND="gloo", EEGLDM_LOCAL_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "16",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 32 and j["config"]["parallelism"] == "dp2" and j["scaling"] == "weak"
    assert j["value"] > 0 and j["steps"] == 2 and j["unit"] == "windows/s" and j["roofline"]["bound"] == "mfma"


def test_bench_self_spawns_eight_ranks_on_one_gpu():
    """The driver's 8-GPU line at its REAL world size: `python bench.py --gpus 8` respawns itself under torch.distributed.run
    with 8 ranks -- all on GPU 0 over gloo here (EEGLDM_LOCAL_DEVICE), tiny per-rank batch -- so the respawn, the rank-sharded seeds, the walk over
    the gradient buckets in 8 ranks' hooks, the max-over-ranks timing and every `comm` key of the record run with world = 8 before a node does."""
    env = dict(os.environ, EEGLDM_DIST_BACKEND="gloo", EEGLDM_LOCAL_DEVICE="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "2",
                          "--no-cpu-baseline", "--no-roofline"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 16 and j["config"]["parallelism"] == "dp8" and j["scaling"] == "weak"
    assert j["value"] > 0 and j["steps"] == 2 and j["unit"] == "windows/s"
    c = j["comm"]
    assert c["world_size"] == 8 and c["collective_ranks"] == 8 and c["backend"] == "gloo" and c["collective_is_rccl"] is False
    assert c["grad_bytes_per_rank"] == 4 * 30533121 or c["grad_bytes_per_rank"] > 4 * 30_000_000      # the flat gradient buffer of the config_ldm UNet (padded layout)
    for key in ("allreduce_ms_bare", "allreduce_busbw_GBs", "compute_only_ms_per_step", "exposed_comm_ms_per_step", "bucket_bytes", "native_collectives"):
        assert key in c, key
    assert j["collective_ranks"] == 8 and j["collective_backend"] == "gloo"


def test_native_rccl_communicator_one_rank():
    """eegldm_comm_* (csrc/comm.hip): RCCL resolved with dlopen, communicator of ONE rank on this GPU.  With one rank the mean / broadcast
    are identities, so this checks the plumbing only -- id exchange, stream ordering against the Context, bucketing, ncclAvg, wait.
    (No multi-GPU node: the N > 1 behaviour of this path is untested; the torch.distributed path has the two-process tests above.)"""
    import torch
    import eegldm
    from eegldm import distributed as D
    ctx = eegldm.default_context(0)
    comm = D.NativeComm(ctx, 0, 1, D.NativeComm.unique_id())
    try:
        from eegldm._lib import lib
        assert lib.eegldm_comm_rank(comm.h) == 0 and lib.eegldm_comm_world(comm.h) == 1
        g = torch.randn(1_000_003, device="cuda"); ref = g.clone()
        comm.allreduce_mean(g, bucket_elems=70_001); comm.wait(); torch.cuda.synchronize()
        assert torch.equal(g, ref)
        comm.broadcast(g, 0); comm.wait(); torch.cuda.synchronize()
        assert torch.equal(g, ref)
        # the overlapped gradient sync driven as the native backward hook drives it: tail first, the rest at the end
        gs = D.OverlappedGradSync(g, bucket_elems=70_001, ctx=ctx, comm=comm)
        gs.begin(); gs.on_ready(600_000, 400_003); g[:10].add_(0.0); gs.finish(); gs.wait(); torch.cuda.synchronize()
        assert torch.equal(g, ref) and sorted(gs.done) == [(600_000, 1_000_003)]
        with pytest.raises(ValueError):
            comm.allreduce_mean(g.double())
    finally:
        comm.close()


# ---------------------------------------------------------------- AutoencoderKL + PatchDiscriminator GAN step, data-parallel
def _aekl_worker(rank, world, port, q):
    """train_autoencoderkl.py:141-144 wraps both networks in nn.DataParallel; here: per-rank fused GAN step (per-rank BatchNorm batch
    statistics, as DataParallel's replicas have), then the mean of both flat gradient buffers over ranks."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      EEGLDM_DIST_BACKEND="gloo", EEGLDM_LOCAL_DEVICE="0")
    import torch.distributed as dist
    from eegldm import distributed as D
    from eegldm.models import AutoencoderKL, PatchDiscriminator
    from eegldm.training import aekl_train_step
    D.init_from_env()
    torch.manual_seed(10 + rank)                              # different initial weights and BatchNorm buffers per rank: the broadcast must fix that
    ae = AutoencoderKL(spatial_dims=1, in_channels=1, out_channels=1, num_channels=[8, 8, 16], latent_channels=1, num_res_blocks=2,
                       norm_num_groups=1, dtype="float32")
    disc = PatchDiscriminator(spatial_dims=1, num_layers_d=3, num_channels=16, in_channels=1, out_channels=1, kernel_size=3, norm="BATCH",
                              bias=False, padding=1, dtype="float32")
    ae.load_state_dict({k: torch.randn(v.shape) * 0.1 + (1.0 if "norm" in k and k.endswith("weight") else 0.0) for k, v in ae.state_dict().items()})
    disc.buffers.add_(0.25 * rank)
    D.broadcast_flat(ae.flat); ae.sync_weights()
    D.broadcast_flat(disc.flat); D.broadcast_flat(disc.buffers); disc.sync_weights()
    dev = ae.device
    buf0 = disc.buffers.clone()

    def batch(rk):
        g = torch.Generator().manual_seed(200 + rk)
        x = torch.rand(4, 1, 256, generator=g); x[:, :, :8] = 0; x[:, :, -8:] = 0
        return x.to(dev), torch.randn(4, 1, 64, generator=g).to(dev)

    def step(rk):
        ae.zero_grad(); disc.zero_grad(); disc.buffers.copy_(buf0)
        aekl_train_step(ae, disc, *batch(rk), 0.01, 1e-6, 0.1, True)
        return ae.flat_grad.clone(), disc.flat_grad.clone()

    want_g, want_d = torch.zeros_like(ae.flat_grad), torch.zeros_like(disc.flat_grad)
    for rk in range(world):                                   # this rank computes every rank's gradients alone and averages them
        g, d = step(rk); want_g += g / world; want_d += d / world
    step(rank)
    D.allreduce_mean_flat(ae.flat_grad); D.allreduce_mean_flat(disc.flat_grad)
    torch.cuda.synchronize()
    eg = float((ae.flat_grad - want_g).abs().max() / want_g.abs().max()); ed = float((disc.flat_grad - want_d).abs().max() / want_d.abs().max())
    q.put((rank, eg, ed, float(ae.flat.double().sum()), float(disc.flat.double().sum()), float(buf0.double().sum()),
           float(want_g.abs().max()), float(want_d.abs().max())))
    dist.destroy_process_group()


def _spawn2(target, extra=()):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 150
    procs = [ctx.Process(target=target, args=(r, 2, port, q) + tuple(extra)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda v: v[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


def test_aekl_gan_step_two_ranks_grad_mean():
    res = _spawn2(_aekl_worker)
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4] and res[0][5] == res[1][5]     # params of both nets + BatchNorm buffers identical
    for rank, eg, ed, _a, _d, _b, mg, md in res:
        assert mg > 0 and md > 0
        assert eg < 1e-5 and ed < 1e-5, (rank, eg, ed)      # all-reduced == mean of the per-rank gradients


def _entry_worker(rank, world, port, q, out):
    """The two training entry points under a 2-rank rendezvous: replicas must end with identical parameters, and train_ldm's
    scale_factor (1/std of the FIRST batch, which differs per rank because the loader is rank-sharded) must be rank 0's everywhere."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      EEGLDM_DIST_BACKEND="gloo", EEGLDM_LOCAL_DEVICE="0")
    import torch.distributed as dist
    from eegldm.entry import train_autoencoderkl as TA, train_ldm as TL
    a_yaml, l_yaml = os.path.join(out, "aekl.yaml"), os.path.join(out, "ldm.yaml")
    run_a = TA.main(TA.parse_args(["--config_file", a_yaml, "--spe", "spectral", "--synthetic_windows", "16", "--latent_channels", "1", "--max_steps", "3"]))
    a = dict(TA.LAST_RUN)
    dist.barrier()                                            # rank 0 has written best_model.pth
    TL.main(TL.parse_args(["--config_file", l_yaml, "--autoencoderkl_config_file_path", a_yaml, "--best_model_path", run_a,
                           "--synthetic_windows", "16", "--latent_channels", "1", "--max_steps", "2"]))
    l = dict(TL.LAST_RUN)
    # what this rank would have computed on its own first batch (the round-2 behaviour): must differ between ranks for the test to mean anything
    q.put((rank, a["ae_sum"], a["d_sum"], a["steps"], l["scale_factor"], l["param_sum"], l["steps"]))
    dist.destroy_process_group()


def test_entry_points_two_ranks_identical_replicas(tmp_path):
    import yaml
    from test_gpu_entry import AEKL_YAML, LDM_YAML
    out = str(tmp_path)
    a = dict(AEKL_YAML); a["train"] = dict(a["train"], output_dir=out)
    l = dict(LDM_YAML); l["train"] = dict(l["train"], output_dir=out)
    yaml.safe_dump(a, open(os.path.join(out, "aekl.yaml"), "w")); yaml.safe_dump(l, open(os.path.join(out, "ldm.yaml"), "w"))
    res = _spawn2(_entry_worker, extra=(out,))
    r0, r1 = res
    assert r0[1] == r1[1] and r0[2] == r1[2] and r0[3] == r1[3] == 3           # autoencoder / discriminator replicas identical after 3 steps
    assert r0[4] == r1[4] and r0[4] > 0                                          # ONE scale factor (rank 0's), bit-identical
    assert r0[5] == r1[5] and r0[6] == r1[6] == 2                              # UNet replicas identical
    ck = torch.load(os.path.join(out, "ldm_eeg_no-spectral_edfx", "checkpoint.pth"))
    assert float(ck["scale_factor"]) == pytest.approx(r0[4], rel=1e-6)


# ---------------------------------------------------------------- data-parallel TRAINING TRAJECTORY against the single-process oracle
def _ldm_traj_worker(rank, world, port, q, nsteps):
    """Each rank takes its quarter/half of every global batch of tests/golden/ldm_traj_c2.json (the CPU oracle's single-process run at
    B = 8), back-propagates with the overlapped gradient sync, steps Adam: the mean of the ranks' losses must follow the oracle's
    loss step by step -- gradient averaging, its overlap with the backward, and replica consistency over several optimiser steps."""
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      EEGLDM_DIST_BACKEND="gloo", EEGLDM_LOCAL_DEVICE="0")
    import torch.distributed as dist
    from param_gen import gen_param, eeg_windows, normal, timesteps
    from eegldm import distributed as D
    from eegldm.models import UNetModel
    from eegldm.schedulers import DDPMScheduler
    from eegldm.training import Adam, ldm_train_step
    with open(os.path.join(ROOT, "tests", "golden", "ldm_traj_c2.json")) as fh:
        g = json.load(fh)
    D.init_from_env()
    cfg = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
               channel_mult=[1, 2, 4], resblock_updown=True)
    net = UNetModel(**cfg, dtype="float32")
    if rank == 0:                                             # only rank 0 holds the fixture's weights: the broadcast hands them out
        net.load_state_dict({k: torch.from_numpy(gen_param(g["param_seed"], k, tuple(v.shape))) for k, v in net.state_dict().items()})
    D.broadcast_flat(net.flat); net.sync_weights()
    sched = DDPMScheduler(num_train_timesteps=1000, schedule="linear_beta", beta_start=0.0015, beta_end=0.0195)
    opt = Adam(net, lr=g["lr"])
    gs = D.OverlappedGradSync(net.flat_grad, ctx=net.ctx)
    B, POOL = g["batch"], g["pool"]; per = B // world
    pool = torch.from_numpy(eeg_windows(POOL, seed=g["latent_seed"], length=768)).cuda()
    loss = torch.zeros(1, device="cuda"); gaps = []
    for i in range(1, nsteps + 1):
        s = ((i - 1) * B) % POOL + rank * per
        nz = torch.from_numpy(normal((B, 1, 768), seed=g["noise_seed_base"] + i))[rank * per:(rank + 1) * per].cuda()
        t = torch.from_numpy(timesteps(B, seed=g["t_seed_base"] + i))[rank * per:(rank + 1) * per].cuda()
        net.zero_grad()
        ldm_train_step(net, sched, pool[s:s + per], nz, t, loss_out=loss, grad_sync=gs); gs.wait()
        opt.step()
        tot = loss.detach().clone().cpu(); dist.all_reduce(tot)
        gaps.append(abs(float(tot) / world - g["loss"][i - 1]) / g["loss"][i - 1])
    torch.cuda.synchronize()
    q.put((rank, gaps, float(net.flat.double().sum())))
    dist.destroy_process_group()


def test_data_parallel_training_follows_the_single_process_oracle_trajectory():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + os.getpid() % 90
    nsteps = 8
    procs = [ctx.Process(target=_ldm_traj_worker, args=(r, 2, port, q, nsteps)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda v: v[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert res[0][2] == res[1][2], "replicas drifted apart"    # identical parameters on both ranks after 8 synchronised steps
    for rank, gaps, _s in res:
        assert max(gaps) < 2e-3, (rank, gaps)                  # the single-GPU fp32 engine follows this fixture to 2e-5 (test_gpu_zz_convergence.py)


# ---------------------------------------------------------------- real RCCL, one GPU per rank (auto-skips on a single-GPU box)
def _rccl_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ.pop("EEGLDM_DIST_BACKEND", None); os.environ.pop("EEGLDM_LOCAL_DEVICE", None)
    import torch.distributed as dist
    import eegldm
    from eegldm import distributed as D
    r, local, w = D.init_from_env()
    assert dist.get_backend() == "nccl" and local == rank
    ctx = eegldm.default_context(local)
    dev = torch.device("cuda", local)
    g = torch.full((3_000_001,), float(rank + 1), device=dev)
    D.allreduce_mean_flat(g, bucket_elems=1_000_000)                               # torch.distributed (RCCL) path
    ok_torch = bool(torch.allclose(g, torch.full_like(g, (world + 1) / 2.0)))
    comm = D.NativeComm.from_process_group(ctx)                                    # unique id from rank 0 through the initialised group
    h = torch.full((2_000_003,), float(rank + 1), device=dev)
    comm.allreduce_mean(h, bucket_elems=700_001); comm.wait(); torch.cuda.synchronize()
    ok_native = bool(torch.allclose(h, torch.full_like(h, (world + 1) / 2.0)))
    comm.close()
    q.put((rank, ok_torch, ok_native))
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="native RCCL with more than one rank needs one GPU per rank (the driver's multi-GPU node)")
def test_rccl_two_ranks_one_gpu_each():
    """The first place RCCL runs with world > 1: backend must resolve to nccl, the bucketed gradient mean must be exact for both the
    torch.distributed path and the C-ABI communicator (eegldm_comm_*)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 90
    procs = [ctx.Process(target=_rccl_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda v: v[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok_torch, ok_native in res:
        assert ok_torch and ok_native, (rank, ok_torch, ok_native)
