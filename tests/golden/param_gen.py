"""Deterministic parameter / input generators shared by the golden-vector
generator (tests/golden/make_golden.py) and the tests.

Weights are never stored in fixtures: they are regenerated from a numpy
Generator keyed on (seed, state-dict key), so a fixture only has to hold the
inputs' seeds and the expected outputs.  The zero-initialised layers of the
reference UNet (`zero_module`, /root/reference/src/models/unet.py:39-45, used
at :161, :290-292, :504) are overwritten like every other tensor, otherwise
the network output is identically zero and any comparison is vacuous
(SURVEY.md Appendix C item 12).
"""
import zlib

import numpy as np


def _rng(seed, key):
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def gen_param(seed, key, shape):
    """One tensor of a state dict.  Norm scales ~ 1 + 0.1 N, biases ~ 0.1 N,
    conv / linear weights ~ N(0, 1/fan_in) * 1.0 (keeps activations O(1))."""
    r = _rng(seed, key)
    shape = tuple(int(s) for s in shape)
    leaf = key.split(".")[-1]
    if len(shape) == 1:
        if leaf == "weight":          # GroupNorm / BatchNorm scale
            return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
        if leaf == "running_var":
            return (1.0 + 0.1 * r.random(shape)).astype(np.float32)
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 0:
        return np.zeros((), np.int64)
    fan_in = int(np.prod(shape[1:]))
    return (r.standard_normal(shape) / np.sqrt(fan_in)).astype(np.float32)


def gen_state_dict(seed, shapes):
    """shapes: mapping key -> shape (ordered).  Returns key -> np.ndarray."""
    return {k: gen_param(seed, k, s) for k, s in shapes.items()}


def eeg_windows(batch, seed=1234, length=3072, pad=36):
    """Synthetic 30-s windows, SURVEY.md §8(d): interior 3000 samples =
    clip(0.5 + 0.1*sum_f a_f sin(2 pi f n/100 + phi_f) + 0.05 N(0,1), 0, 1),
    f in {1.5, 6, 10, 13} Hz, exact zeros in the first / last 36 samples
    (mimics /root/reference/src/dataset/dataset.py:12-19)."""
    r = np.random.default_rng(seed)
    n = np.arange(length - 2 * pad, dtype=np.float64)
    x = np.zeros((batch, 1, length), np.float32)
    for b in range(batch):
        sig = np.zeros_like(n)
        for f in (1.5, 6.0, 10.0, 13.0):
            a = r.uniform(0.2, 1.0)
            ph = r.uniform(0.0, 2 * np.pi)
            sig += a * np.sin(2 * np.pi * f * n / 100.0 + ph)
        w = 0.5 + 0.1 * sig + 0.05 * r.standard_normal(n.shape)
        x[b, 0, pad:length - pad] = np.clip(w, 0.0, 1.0).astype(np.float32)
    return x


def timesteps(batch, seed=1235, num_train=1000):
    return np.random.default_rng(seed).integers(0, num_train, size=(batch,)).astype(np.int64)


def normal(shape, seed=1236):
    return np.random.default_rng(seed).standard_normal(shape).astype(np.float32)
