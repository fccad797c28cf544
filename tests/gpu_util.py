"""Helpers for the -m gpu parity tests: NCL<->NLC plumbing and tolerance checks.
torch here is plumbing (device memory); the compute under test is libeegldm."""
import numpy as np
import torch

import eegldm
from eegldm._lib import lib, ptr, check, F32, BF16

DEV = "cuda:0"
TDT = {F32: torch.float32, BF16: torch.bfloat16}


def ctx():
    return eegldm.default_context(0)


def nlc(x_ncl, dtype=F32, ld=None):
    """(B,C,L) cpu/gpu tensor -> device NLC buffer [(B*L), ld] of the engine dtype (torch permute: plumbing)."""
    B, Cc, L = x_ncl.shape
    t = x_ncl.permute(0, 2, 1).contiguous().reshape(B * L, Cc).to(DEV)
    if ld is not None and ld != Cc:
        buf = torch.zeros(B * L, ld, device=DEV, dtype=TDT[dtype])
        buf[:, :Cc] = t.to(TDT[dtype])
        return buf
    return t.to(TDT[dtype]).contiguous()


def ncl(x_nlc, B, L):
    """device NLC [(B*L), C] -> cpu fp32 (B,C,L)"""
    Cc = x_nlc.shape[1]
    return x_nlc.float().reshape(B, L, Cc).permute(0, 2, 1).contiguous().cpu()


def pack_w(w_ref, dtype=F32):
    """(Cout,Cin,K) -> [K][Cout][Cin] on device in the engine dtype"""
    return w_ref.permute(2, 0, 1).contiguous().to(DEV).to(TDT[dtype]).contiguous()


def unpack_w(w_packed):
    return w_packed.float().permute(1, 2, 0).contiguous().cpu()


def assert_close(got, want, rtol, atol, name=""):
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    assert got.shape == want.shape, f"{name}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    if not (err <= tol).all():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{name}: max violation at {i}: got {got[i]} want {want[i]} (|err| {err[i]:.3e}, tol {tol[i]:.3e}); "
                             f"max|err| {err.max():.3e}, mean|err| {err.mean():.3e}, bad {int((err > tol).sum())}/{err.size}")


# tolerances: fp32 path (exact-fp32 MFMA, fp32 stats) vs the fp32 CPU oracle; bf16 path = storage rounding
TOL = {F32: dict(rtol=1e-4, atol=1e-5), BF16: dict(rtol=3e-2, atol=3e-2)}
GTOL = {F32: dict(rtol=1e-3, atol=1e-4), BF16: dict(rtol=5e-2, atol=5e-2)}
