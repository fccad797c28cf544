"""ORACLE (test infrastructure only -- never imported by the product path).

bf16 storage emulation for the oracle.  The engine's bf16 mode stores every activation (and every activation
gradient) as bf16 while accumulating in fp32; `q()` reproduces that in the fp32 CPU oracle: the forward value is
rounded to bf16 (round-to-nearest-even, what v_cvt_pk_bf16_f32 does) and so is the gradient flowing back through
the same point.  With the switch off (default) `q` is the identity, so the pinned fp32 oracle is untouched.

Used by the -m gpu tests to DERIVE the bf16 tolerances instead of guessing them: gap = ||oracle_bf16 - oracle_fp32||
is the error the storage format alone introduces with a different summation order; the engine's bf16 output must
lie within a small multiple of that gap of the fp32 oracle (tests/gpu_util.py::bf16_gap_bound).
"""
import contextlib

import torch

_ON = False
_FMT = torch.bfloat16          # storage format being emulated (round 5: torch.float16 for the engine's EEGLDM_F16 mode)


class _RoundBF16(torch.autograd.Function):      # (name kept: rounds to the CURRENT storage format)
    @staticmethod
    def forward(ctx, x):
        return x.to(_FMT).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g.to(_FMT).to(torch.float32)


def q(x):
    """Activation storage point."""
    return _RoundBF16.apply(x) if _ON and x.dtype == torch.float32 else x


def qw(w):
    """Weight read point: the engine's GEMMs read a bf16 copy of the fp32 master weights (the gradient stays fp32)."""
    if not _ON or w.dtype != torch.float32:
        return w
    return w + (w.to(_FMT).to(torch.float32) - w).detach()


@contextlib.contextmanager
def bf16_storage(on=True, fmt=torch.bfloat16):
    global _ON, _FMT
    prev, prev_fmt = _ON, _FMT
    _ON, _FMT = bool(on), fmt
    try:
        yield
    finally:
        _ON, _FMT = prev, prev_fmt


def f16_storage(on=True):
    """The same emulation with IEEE half as the storage format (the reference's autocast dtype, src/training/training.py:423)."""
    return bf16_storage(on, torch.float16)
