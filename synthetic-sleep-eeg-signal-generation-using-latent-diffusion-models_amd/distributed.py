"""Data-parallel glue: one process per GPU (torch.distributed, backend "nccl" == RCCL over
xGMI on ROCm; "gloo" for the CPU tests).  Replaces the reference's single-process
nn.DataParallel (/root/reference/src/train_ldm.py:190-192): parameters are broadcast once
at start, and each step all-reduces the model's FLAT fp32 gradient buffer (122 MB for the
config_ldm UNet) in a few large buckets -- sized for xGMI's per-link bandwidth, not for
NVSwitch -- then scales by 1/world.  Sampling shards seeds and needs no collective."""
import os

import torch
import torch.distributed as dist

BUCKET_ELEMS = 8 * 1024 * 1024      # 32 MB fp32 buckets: 4 in-flight collectives for the UNet


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: run several ranks on ONE GPU over gloo (EEGLDM_DIST_BACKEND=gloo EEGLDM_LOCAL_DEVICE=0) to exercise the
    # multi-process control flow where only a single-GPU box is available
    if "EEGLDM_LOCAL_DEVICE" in os.environ:
        local = int(os.environ["EEGLDM_LOCAL_DEVICE"])
    backend = backend or os.environ.get("EEGLDM_DIST_BACKEND")
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def broadcast_flat(t, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)


def _comm_device(like=None):
    """Device a small control tensor must live on for the initialised backend (nccl = RCCL: the rank's GPU; gloo: CPU)."""
    if dist.get_backend() == "nccl":
        return like.device if (like is not None and like.is_cuda) else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def broadcast_scalar(value, src=0, like=None):
    """A Python float that is the SAME on every rank afterwards (rank `src`'s value).  Used for quantities the reference computes
    once in its single process and that data-parallel replicas must agree on -- the latent `scale_factor = 1 / std(z)` of the first
    batch (/root/reference/src/train_ldm.py:145-148,203-204): computed per rank from a rank-sharded loader it would differ per
    replica while only rank 0's value reaches checkpoint.pth."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_comm_device(like))
    dist.broadcast(t, src=src)
    return float(t.item())


def allreduce_sum_scalars(values, like=None):
    """Element-wise sum over ranks of a short list of Python numbers (validation loss sums / counts)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=_comm_device(like))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t.tolist()]


_AVG_OK = None      # ReduceOp.AVG usable on this backend build (probed once, collectively: every rank takes the same branch)


def probe_mean_op(like=None):
    """Decide ONCE, at set-up time and collectively, whether the gradient mean runs as ReduceOp.AVG (RCCL's ncclAvg: no extra read + write
    of the 122 MB gradient buffer per step) or as SUM + one 1/world pass (gloo; an RCCL build without AVG; EEGLDM_NO_NCCL_AVG set).

    Called from OverlappedGradSync.__init__ / the first allreduce_mean_flat -- i.e. OUTSIDE the native backward's gradient hook, where a
    blocking all-reduce plus a host read would sit in the middle of the backward.  Every rank contributes its own verdict (its
    environment override included) to a MIN all-reduce, so a switch set on only some ranks can no longer mix AVG and SUM on one buffer."""
    global _AVG_OK
    if _AVG_OK is not None or not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(_AVG_OK)
    dev = _comm_device(like)      # nccl: the rank's GPU; gloo: the CPU
    mine = 0
    if dist.get_backend() == "nccl" and dev.type == "cuda":
        mine = 0 if os.environ.get("EEGLDM_NO_NCCL_AVG") is not None else 1
        try:      # every nccl rank runs the probe collective, whatever its override says: the call sequence is the same on all ranks
            probe = torch.full((1,), 3.0, device=dev)
            dist.all_reduce(probe, op=dist.ReduceOp.AVG)
            if abs(float(probe) - 3.0) >= 1e-6:
                mine = 0
        except Exception as e:      # noqa: BLE001
            print(f"[eegldm] ReduceOp.AVG unavailable ({e}); gradient mean as SUM + scale", flush=True)
            mine = 0
    flag = torch.tensor([mine], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    _AVG_OK = bool(int(flag.item()))
    return _AVG_OK


def _mean_op(t):
    """(reduce op, needs a separate 1/world pass) -- the verdict of probe_mean_op (made at set-up; probed here only if nobody did)."""
    if _AVG_OK is None:
        probe_mean_op(t)
    if _AVG_OK and t.is_cuda and dist.get_backend() == "nccl":
        return dist.ReduceOp.AVG, False
    return dist.ReduceOp.SUM, True


def allreduce_mean_flat(t, bucket_elems=BUCKET_ELEMS):
    """In-place mean over ranks of a flat tensor, bucketed; async ops are all issued before the first wait."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    world = dist.get_world_size()
    op, post = _mean_op(t)
    works = []
    for s in range(0, t.numel(), bucket_elems):
        works.append(dist.all_reduce(t[s:s + bucket_elems], op=op, async_op=True))
    for w in works:
        w.wait()
    if post:
        t.mul_(1.0 / world)


class NativeComm:
    """RCCL communicator behind the C ABI (csrc/comm.hip, `eegldm_comm_*`): the gradient mean runs as ncclAvg collectives on the
    communicator's own stream, ordered after the Context's stream by an event, without going through torch.distributed.
    Opt-in (`EEGLDM_NATIVE_COLLECTIVES=1`, see `make_comm`): with real RCCL it has only ever run with ONE rank -- no multi-GPU node was
    available to the builder; its bucket arithmetic / stream ordering run for world 2 / 4 / 8 against a call-recording stand-in
    (tests/test_gpu_comm_fake.py) -- whereas the torch.distributed path is exercised with two processes (gloo) by the tests."""

    def __init__(self, ctx, rank, world, unique_id):
        import ctypes as C
        from ._lib import lib, check
        assert len(unique_id) == 128
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        h = C.c_void_p()
        check(lib.eegldm_comm_create(ctx.h, C.c_char_p(bytes(unique_id)), self.rank, self.world, C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id():
        import ctypes as C
        from ._lib import lib, check
        buf = C.create_string_buffer(128)
        check(lib.eegldm_comm_unique_id(buf))
        return buf.raw

    @classmethod
    def from_process_group(cls, ctx):
        """One id made by rank 0 and handed round through the initialised torch.distributed group (any backend)."""
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        # rank 0 may fail to make the id (librccl / EEGLDM_RCCL_LIB not loadable): it still takes part in the broadcast, with the error as the
        # payload, so that every rank leaves this function the same way (an exception that make_comm turns into a collective fallback)
        # instead of rank 0 moving on to the next collective while the others wait in this one
        box = [None]
        if rank == 0:
            try:
                box = [cls.unique_id()]
            except Exception as e:      # noqa: BLE001
                box = [("error", str(e))]
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if not isinstance(box[0], (bytes, bytearray)):
            raise RuntimeError(f"rank 0 could not create the RCCL unique id: {box[0][1] if box[0] else 'no id'}")
        return cls(ctx, rank, world, box[0])

    def _chk(self, t):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("NativeComm works on contiguous fp32 device tensors (the flat parameter / gradient buffers)")

    def allreduce_mean(self, t, bucket_elems=BUCKET_ELEMS):
        from ._lib import lib, check, ptr
        self._chk(t)
        check(lib.eegldm_comm_allreduce_mean_f32(self.h, ptr(t), t.numel(), int(bucket_elems)))

    def broadcast(self, t, root=0):
        from ._lib import lib, check, ptr
        self._chk(t)
        check(lib.eegldm_comm_broadcast_f32(self.h, ptr(t), t.numel(), int(root)))

    def wait(self):
        from ._lib import lib, check
        check(lib.eegldm_comm_wait(self.h))

    def close(self):
        from ._lib import lib
        if self.h:
            lib.eegldm_comm_destroy(self.h); self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_comm(ctx):
    """The native communicator when asked for (EEGLDM_NATIVE_COLLECTIVES=1) and more than one GPU rank is running, else None
    (callers then use the torch.distributed collectives -- ReduceOp.AVG on RCCL, so neither path makes a separate 1/world pass).

    Why it is not the default: RCCL with more than one rank has never executed for this builder (no multi-GPU node), and the first run that
    does is the driver's scaling bench.  torch.distributed's "nccl" backend IS RCCL over xGMI and is the path thousands of jobs exercise;
    the C-ABI communicator binds to the same library but its multi-rank bring-up (ncclCommInitRank inside a process that already holds
    torch's communicator) is only covered by a call-recording stand-in (tests/test_gpu_comm_fake.py).  `tools/first_multigpu.sh` runs both
    on the first node that has two GPUs.  Creation is collective: every rank reports success and ALL fall back together if any failed."""
    if os.environ.get("EEGLDM_NATIVE_COLLECTIVES", "0") != "1" or not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    if not torch.cuda.is_available():
        return None
    comm, ok = None, 1
    try:
        comm = NativeComm.from_process_group(ctx)
    except Exception as e:      # noqa: BLE001  (any failure on any rank: everybody falls back)
        print(f"[eegldm] rank {dist.get_rank()}: native RCCL communicator failed ({e}); falling back to torch.distributed", flush=True)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=_comm_device())
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()) == 0:
        if comm is not None:
            comm.close()
        return None
    return comm


class OverlappedGradSync:
    """Data-parallel gradient mean with the communication started before backward has finished.

    `on_ready(offset, numel)` (called from the native backward through eegldm.training.set_grad_hook) launches the
    bucketed async all-reduce of a finished slice of the flat gradient buffer; `finish()` reduces whatever has not been
    covered yet; `wait()` blocks the current stream on all of it and applies the 1/world scaling.  With one process it
    does nothing.  Slices must not overlap (the native hook reports one tail range)."""

    def __init__(self, flat_grad, bucket_elems=BUCKET_ELEMS, ctx=None, comm=None):
        self.g = flat_grad
        self.ctx = ctx          # eegldm.Context whose stream the native backward enqueues on (checked in on_ready)
        self.comm = comm        # NativeComm: collectives through the C ABI instead of torch.distributed
        self.bucket = int(bucket_elems)
        self.works = []
        self.done = []          # (start, end) ranges already launched
        self._post_scale = True
        if comm is None:        # AVG-or-SUM is agreed on here, once and by all ranks -- not inside the backward's gradient hook
            probe_mean_op(flat_grad)

    @property
    def active(self):
        if self.comm is not None:
            return True
        return dist.is_initialized() and dist.get_world_size() > 1

    def begin(self):
        self.works, self.done = [], []

    def _launch(self, a, b):
        if self.comm is not None:     # ordered after the Context's stream inside the library; one group of bucketed ncclAvg collectives
            self.comm.allreduce_mean(self.g[a:b], self.bucket)
            return
        op, post = _mean_op(self.g)
        self._post_scale = post
        for s in range(a, b, self.bucket):
            e = min(b, s + self.bucket)
            self.works.append(dist.all_reduce(self.g[s:e], op=op, async_op=True))

    def on_ready(self, offset, numel):
        if not self.active or numel <= 0:
            return
        # dist.all_reduce orders itself after torch's CURRENT stream; the backward that produced this slice was enqueued on the
        # context's stream (torch's current stream when the Context was created).  They must be the same stream.  `stream_handle`
        # is the stream the library REALLY uses (eegldm_ctx_stream): a Context(use_torch_stream=False) owns a private stream that
        # can never equal torch's, so it is refused here instead of passing as "stream 0 == torch's default stream".
        if self.comm is None and self.ctx is not None and self.g.is_cuda and torch.cuda.current_stream(self.g.device).cuda_stream != self.ctx.stream_handle:
            raise RuntimeError("OverlappedGradSync: torch's current stream differs from the stream the eegldm Context enqueues on; "
                               "the all-reduce would not be ordered after the backward (create the Context under the stream you train on)")
        self._launch(offset, offset + numel)
        self.done.append((offset, offset + numel))

    def finish(self):
        if not self.active:
            return
        pos = 0
        for a, b in sorted(self.done):
            if a > pos:
                self._launch(pos, a)
            pos = max(pos, b)
        if pos < self.g.numel():
            self._launch(pos, self.g.numel())

    def wait(self):
        if not self.active:
            return
        if self.comm is not None:
            self.comm.wait()          # the mean is part of the collective (ncclAvg)
            return
        for w in self.works:
            w.wait()
        self.works = []
        if self._post_scale:          # gloo: sum, then one pass; RCCL: the mean was the collective (ReduceOp.AVG)
            self.g.mul_(1.0 / dist.get_world_size())


def shard_range(n, rank, world):
    """Contiguous split of n independent items (sampling seeds) over ranks."""
    per, rem = divmod(n, world)
    start = rank * per + min(rank, rem)
    return start, start + per + (1 if rank < rem else 0)
