"""CPU, world_size 2, gloo: the data-parallel glue (flat-gradient all-reduce mean, parameter broadcast,
seed sharding) gives the same result as one rank on the concatenated batch."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from eegldm import distributed as D
    r, _l, w = D.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    flat = torch.randn(1000) if rank == 0 else torch.zeros(1000)
    D.broadcast_flat(flat)                              # parameters: rank 0 -> all
    grads = torch.arange(3000, dtype=torch.float32) * (rank + 1)
    D.allreduce_mean_flat(grads, bucket_elems=700)      # several uneven buckets
    lo, hi = D.shard_range(11, rank, world)
    # overlapped gradient sync: the "native backward" reports a finished tail slice first, the head follows
    g2 = torch.arange(3000, dtype=torch.float32) * (rank + 1)
    gs = D.OverlappedGradSync(g2, bucket_elems=700)
    gs.begin(); gs.on_ready(1800, 1200); gs.finish(); gs.wait()
    assert torch.allclose(g2, grads), "overlapped sync must equal the plain bucketed mean"
    q.put((rank, flat.sum().item(), grads.tolist(), (lo, hi)))     # plain python: no fd passing after the child exits
    dist.destroy_process_group()


def test_gloo_world2_allreduce_and_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 400
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1]                                    # broadcast made parameters identical
    want = torch.arange(3000, dtype=torch.float32) * 1.5             # mean of 1x and 2x
    assert torch.allclose(torch.tensor(res[0][2]), want) and torch.allclose(torch.tensor(res[1][2]), want)
    assert res[0][3] == (0, 6) and res[1][3] == (6, 11)             # 11 seeds over 2 ranks, contiguous, complete


def test_single_process_is_noop():
    sys.path.insert(0, ROOT)
    from eegldm import distributed as D
    t = torch.ones(10)
    D.allreduce_mean_flat(t); D.broadcast_flat(t)
    assert torch.equal(t, torch.ones(10)) and D.shard_range(5, 0, 1) == (0, 5)
    gs = D.OverlappedGradSync(t)
    gs.begin(); gs.on_ready(4, 6); gs.finish(); gs.wait()
    assert torch.equal(t, torch.ones(10))


def _scalar_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from eegldm import distributed as D
    D.init_from_env(backend="gloo")
    sf = D.broadcast_scalar(1.0 / (3.0 + rank), src=0)          # per-rank value differs (rank-sharded first batch); rank 0's must win
    tot = D.allreduce_sum_scalars([1.5 * (rank + 1), 4 + rank])
    q.put((rank, sf, tot))
    dist.destroy_process_group()


def test_gloo_world2_broadcast_scalar_and_sum():
    """train_ldm's scale_factor: computed on rank 0, identical everywhere; validation (sum, count) added over ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + os.getpid() % 300
    procs = [ctx.Process(target=_scalar_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 1.0 / 3.0
    assert res[0][2] == res[1][2] == [4.5, 9.0]


def test_shard_files_equal_length_and_cover():
    """Every rank gets ceil(N/world) files (ranks with unequal batch counts would hang in the per-step all-reduce); the union covers
    the list; a run that names real data and finds none raises instead of training on synthetic windows."""
    sys.path.insert(0, ROOT)
    import pytest
    from eegldm.entry.common import WindowLoader, shard_files
    files = [f"f{i}" for i in range(31)]
    for world in (1, 2, 3, 4, 8):
        shards = [shard_files(files, r, world) for r in range(world)]
        assert len({len(s) for s in shards}) == 1 and len(shards[0]) == -(-31 // world)
        assert set().union(*map(set, shards)) == set(files)
    assert shard_files(["a"], 3, 4) == ["a"]                    # fewer files than ranks: wrap around, never an empty shard
    assert D_single().broadcast_scalar(2.5) == 2.5 and D_single().allreduce_sum_scalars([1, 2]) == [1.0, 2.0]
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        with pytest.raises(FileNotFoundError):
            WindowLoader(d, 4)                                   # an empty data directory is an error, not a synthetic run
    assert len(WindowLoader(None, 4, n_synthetic=8)) == 2       # synthetic runs still work when asked for


def D_single():
    from eegldm import distributed as D
    return D
