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
