"""-m gpu: the C-ABI communicator (csrc/comm.hip, eegldm_comm_*) driven for world sizes 2 / 4 / 8 on a ONE-GPU box through a call-recording
stand-in for librccl (tests/fake_rccl/fake_rccl.hip, selected with EEGLDM_RCCL_LIB).  What this pins before the first real multi-GPU run
(VERDICT r4 item 7): the bucket arithmetic of eegldm_comm_allreduce_mean_f32 (every element in exactly one bucket, for buffer lengths
that are not multiples of the bucket; bucket <= 0 = one collective), the group bracket, ncclAvg on fp32, the ordering of the collective
stream against the context's stream in both directions, the unique id arriving intact, create / destroy balance, and
OverlappedGradSync's slice bookkeeping (tail slice from the backward hook + the head afterwards) on the native communicator.
The stand-in's "all-reduce" adds 1.0 to every element it is handed, on the stream it is handed.
Reference being replaced: nn.DataParallel's gradient reduction, /root/reference/src/train_ldm.py:190-192."""
import ctypes as C
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FAKE = os.path.join(HERE, "fake_rccl", "libfake_rccl.so")


@pytest.fixture()
def fake():
    assert os.path.exists(FAKE), "build tests/fake_rccl/libfake_rccl.so first (make / __graft_entry__.build())"
    old = os.environ.get("EEGLDM_RCCL_LIB")
    os.environ["EEGLDM_RCCL_LIB"] = FAKE
    f = C.CDLL(FAKE)
    f.fake_rccl_log.argtypes = [C.c_int, C.POINTER(C.c_long)]
    f.fake_rccl_reset()
    yield f
    if old is None:
        os.environ.pop("EEGLDM_RCCL_LIB", None)
    else:
        os.environ["EEGLDM_RCCL_LIB"] = old


def _log(f):
    out = []
    buf = (C.c_long * 8)()
    for i in range(f.fake_rccl_log_size()):
        assert f.fake_rccl_log(i, buf) == 0
        out.append(dict(kind=buf[0], off=buf[1], count=buf[2], dtype=buf[3], op=buf[4], in_group=buf[5], world=buf[6], rank=buf[7]))
    return out


@pytest.mark.parametrize("world", [2, 4, 8])
def test_bucket_arithmetic_and_stream_ordering(fake, world):
    import eegldm
    from eegldm.distributed import NativeComm
    ctx = eegldm.default_context(0)
    uid = NativeComm.unique_id()
    assert uid == b"\x5a" * 128
    comm = NativeComm(ctx, world - 1, world, uid)            # the last rank: rank and world must reach ncclCommInitRank as given
    try:
        assert fake.fake_rccl_live() == 1
        n = 30_533_121                                      # the config_ldm UNet's parameter count: not a multiple of any bucket
        for bucket in (8 * 1024 * 1024, 1_000_003, 0, n, n + 5):
            fake.fake_rccl_reset()
            # the collective must be ordered AFTER work already enqueued on the context's stream (a long fill), and work enqueued after
            # comm.wait() must see its result
            g = torch.empty(n, device="cuda")
            for _ in range(4):
                g.fill_(3.0)                                 # torch's current stream == the context's stream
            comm.allreduce_mean(g, bucket)
            comm.wait()
            h = g * 2.0
            torch.cuda.synchronize()
            assert float(h.min()) == 8.0 and float(h.max()) == 8.0, (bucket, float(h.min()), float(h.max()))
            log = _log(fake)
            step = bucket if bucket > 0 else n
            want = [(s, min(step, n - s)) for s in range(0, n, step)]
            assert [(r["off"], r["count"]) for r in log] == want
            assert all(r["kind"] == 0 and r["in_group"] == 1 and r["world"] == world and r["rank"] == world - 1 for r in log)
            assert all(r["dtype"] == fake.fake_rccl_f32() and r["op"] == fake.fake_rccl_avg_op() for r in log)
        # empty buffer: no call at all
        fake.fake_rccl_reset()
        comm.allreduce_mean(torch.empty(0, device="cuda"), 1024)
        assert fake.fake_rccl_log_size() == 0
        # broadcast: one call, root passed through
        p = torch.ones(1000, device="cuda")
        comm.broadcast(p, root=0); comm.wait()
        assert [(r["kind"], r["count"], r["op"]) for r in _log(fake)] == [(1, 1000, 0)]
    finally:
        comm.close()
    assert fake.fake_rccl_live() == 0


def test_overlapped_sync_covers_every_element_once_on_the_native_communicator(fake):
    """The native backward reports ONE finished tail slice through the hook; finish() reduces the head.  Every element exactly once."""
    import eegldm
    from eegldm.distributed import NativeComm, OverlappedGradSync
    ctx = eegldm.default_context(0)
    comm = NativeComm(ctx, 0, 8, NativeComm.unique_id())
    try:
        n = 5_000_001
        g = torch.zeros(n, device="cuda")
        sync = OverlappedGradSync(g, bucket_elems=1 << 20, ctx=ctx, comm=comm)
        for tail_start in (0, 1, 1_234_567, n - 1, n):
            g.zero_(); fake.fake_rccl_reset()
            sync.begin()
            sync.on_ready(tail_start, n - tail_start)
            sync.finish()
            sync.wait()
            torch.cuda.synchronize()
            assert float(g.min()) == 1.0 and float(g.max()) == 1.0, tail_start
            log = _log(fake)
            base = min(r["off"] for r in log)            # the stand-in logs offsets relative to the FIRST call's pointer (the tail slice)
            covered = sorted((r["off"] - base, r["off"] - base + r["count"]) for r in log)
            pos = 0
            for a, b in covered:
                assert a == pos; pos = b
            assert pos == n
    finally:
        comm.close()
