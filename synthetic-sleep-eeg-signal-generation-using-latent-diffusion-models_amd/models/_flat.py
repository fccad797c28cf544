"""Shared state handling of the model shims: one flat fp32 parameter tensor + one flat gradient
tensor per model (bound to the native executor), viewed as a reference-compatible state_dict."""
import ctypes as C
from collections import OrderedDict

import torch

from .._lib import F32, BF16, F16

DT = {"float32": F32, "fp32": F32, torch.float32: F32, "bfloat16": BF16, "bf16": BF16, torch.bfloat16: BF16, "float16": F16, "fp16": F16, "half": F16, torch.float16: F16, F32: F32, BF16: BF16, F16: F16}


def read_entries(h, n_fn, entry_fn):
    entries = OrderedDict()
    name = C.create_string_buffer(256)
    off, numel, ndim, shape = C.c_long(), C.c_long(), C.c_int(), (C.c_int * 3)()
    for i in range(n_fn(h)):
        rc = entry_fn(h, i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), shape)
        assert rc == 0
        entries[name.value.decode()] = (off.value, numel.value, tuple(shape[k] for k in range(ndim.value)))
    return entries


def unpack(flat, entries):
    out = OrderedDict()
    for k, (o, n, shape) in entries.items():
        t = flat[o:o + n]
        if len(shape) == 3:       # packed [K][Cout][Cin] -> reference (Cout, Cin, K)
            t = t.reshape(shape[2], shape[0], shape[1]).permute(1, 2, 0)
        out[k] = t.reshape(shape).clone()
    return out


def pack_into(flat, entries, sd, strict=True, extra_ok=()):
    sd = {(k[7:] if k.startswith("module.") else k): v for k, v in sd.items()}   # DataParallel prefix (compute_mmds.py:475-478)
    missing = [k for k in entries if k not in sd]
    extra = [k for k in sd if k not in entries and k not in extra_ok]
    if strict and (missing or extra):
        raise KeyError(f"state_dict mismatch: missing {missing[:4]}, unexpected {extra[:4]}")
    for k, (o, n, shape) in entries.items():
        if k not in sd:
            continue
        v = torch.as_tensor(sd[k]).detach().to(torch.float32)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"{k}: shape {tuple(v.shape)} != {tuple(shape)}")
        if len(shape) == 3:
            v = v.permute(2, 0, 1)
        flat[o:o + n].copy_(v.reshape(-1).to(flat.device))
