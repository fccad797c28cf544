"""Helpers for the -m gpu parity tests: NCL<->NLC plumbing and tolerance checks.
torch here is plumbing (device memory); the compute under test is libeegldm."""
import numpy as np
import torch

import eegldm
from eegldm._lib import lib, ptr, check, F32, BF16, F16

DEV = "cuda:0"
TDT = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}


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
TOL = {F32: dict(rtol=1e-4, atol=1e-5), BF16: dict(rtol=3e-2, atol=3e-2), F16: dict(rtol=4e-3, atol=4e-3)}       # fp16: 11 significant bits against bf16's 8
GTOL = {F32: dict(rtol=1e-3, atol=1e-4), BF16: dict(rtol=5e-2, atol=5e-2), F16: dict(rtol=8e-3, atol=8e-3)}


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().double().cpu().reshape(-1); b = torch.as_tensor(b).detach().double().cpu().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-12))


# bf16 bounds are DERIVED, not guessed: `gap` = rel-L2 distance between the fp32 oracle and the same oracle with bf16 storage
# emulated at every activation / activation-gradient / weight-read point (oracle/quant.py).  That is the error the storage
# format alone causes when the summation order differs; an engine result must lie within BF16_GAP_FACTOR x gap of the fp32
# oracle (plus a floor of a few bf16 ulps for tensors whose gap happens to be tiny).  Measured on MI355X (round 2): the
# config_ldm UNet sits at 1.0-1.2 x gap on every one of its 278 gradients, the AutoencoderKL / discriminator at 1.0-1.8 x.
BF16_GAP_FACTOR = 2.0
BF16_FLOOR = 2.0 ** -7
# EEGLDM_F16 (round 5): the same derivation with IEEE half as the emulated storage format (oracle.quant.f16_storage).  The floor is two fp16
# ulps: where the measured gap of a tensor is below one ulp the bound is the floor, and the engine's error there is one draw of rounding noise
# that depends on which kernel (summation order) served the layer -- 0.7-1.2 ulp over the kernels of a 32-64-channel autoencoder.
F16_FLOOR = 2.0 ** -9
SMALL = 64          # tensors with fewer elements are judged pooled (see assert_bf16_grads)


def bf16_gap_bound(gap, floor=BF16_FLOOR, factor=BF16_GAP_FACTOR):
    return max(factor * gap, floor)


def grads_rel_errors(got, want, floor_frac):
    """per-key relative L2 error of parameter gradients; `floor_frac` of the model-wide gradient scale is added to the
    denominator so tensors whose true gradient is rounding noise do not dominate"""
    gscale = max(float(torch.as_tensor(v).double().norm()) for v in want.values())
    out = {}
    for k, w in want.items():
        w = torch.as_tensor(w).double().cpu()
        out[k] = float((got[k].double().cpu() - w).norm()) / (float(w.norm()) + floor_frac * gscale)
    return out


def assert_bf16_grads(got, want32, wantq, label, floor_frac=2e-2, factor=BF16_GAP_FACTOR, floor=BF16_FLOOR):
    """Engine bf16 parameter gradients `got` against the fp32 oracle `want32`, bounded by the storage gap measured with the
    bf16-storage oracle `wantq`.  Tensors of >= SMALL elements are bounded one by one.  For a tensor of a handful of elements
    (a 2-channel GroupNorm scale) the gap is ONE draw of a very noisy quantity (the ratio of two such draws exceeds 3 in 10 % of
    cases at 2 elements), so the small tensors are bounded POOLED (root-mean-square over all of them, ~100 elements in total)
    and each of them only by a gross-error cap (a wrong tap or a double-counted bias is a 100 % error).  Returns a summary string."""
    gap = grads_rel_errors(wantq, want32, floor_frac); err = grads_rel_errors(got, want32, floor_frac)
    big = [k for k in err if want32[k].numel() >= SMALL]; small = [k for k in err if want32[k].numel() < SMALL]
    worst = ("", 0.0, 0.0)
    for k in big:
        b = bf16_gap_bound(gap[k], floor=floor, factor=factor)
        if err[k] / b > worst[1]:
            worst = (k, err[k] / b, gap[k])
        assert err[k] < b, f"{label} {k}: engine {err[k]:.3e} vs storage gap {gap[k]:.3e} (bound {b:.3e})"
    msg = f"{label}: worst {worst[0]} at {worst[1]:.2f} of its bound (gap {worst[2]:.2e})"
    if small:
        pe = (sum(err[k] ** 2 for k in small) / len(small)) ** 0.5; pg = (sum(gap[k] ** 2 for k in small) / len(small)) ** 0.5
        if sum(want32[k].numel() for k in small) >= SMALL:       # a pool of a few elements is as noisy as its members: gross cap only
            assert pe < bf16_gap_bound(pg, floor=floor, factor=factor), f"{label}: pooled small-tensor error {pe:.3e} vs pooled gap {pg:.3e}"
        for k in small:
            assert err[k] < max(8.0 * gap[k], 0.15), f"{label} {k}: engine {err[k]:.3e} vs storage gap {gap[k]:.3e} (gross-error cap)"
        msg += f"; {len(small)} small tensors pooled {pe:.2e} (gap {pg:.2e})"
    return msg
