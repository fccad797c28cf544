"""Shared plumbing of the entry scripts: yaml config (PyYAML; attribute access like OmegaConf),
the reference's `--num_channels "[32,32,64]"` list argument (src/util.py:23-26), run-dir / resume convention
(src/util.py:29-43) and a numpy window loader honouring the loader's output contract
(src/dataset/dataset.py:10-30: dict batch, key 'eeg', float32 (B,1,3072), zero-padded 36 samples each side)."""
import argparse
import ast
import glob
import os

import numpy as np
import torch
import yaml


class Cfg(dict):
    """dict with attribute access, recursively (OmegaConf-like, read side only)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return Cfg(v) if isinstance(v, dict) else v


def load_config(path):
    with open(path) as f:
        return Cfg(yaml.safe_load(f))


class ParseListAction(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, ast.literal_eval(values))


def setup_run_dir(config, args, base_path=None):
    """{output_dir}/{run_dir}_{spe}_{dataset}; resume iff checkpoint.pth exists (util.py:29-43)."""
    name = f"{config.train.run_dir}_{getattr(args, 'spe', 'no-spectral')}_{getattr(args, 'dataset', getattr(args, 'type_dataset', 'edfx'))}"
    run_dir = os.path.join(getattr(args, "output_dir", None) or config.train.output_dir, name)
    os.makedirs(run_dir, exist_ok=True)
    return run_dir, os.path.exists(os.path.join(run_dir, "checkpoint.pth"))


def synthetic_windows(n, seed, length=3072, pad=36):
    """SURVEY 8d synthetic 30-s windows: sinusoid mix + noise in [0,1], exact zeros in the pads."""
    r = np.random.default_rng(seed)
    t = np.arange(length - 2 * pad, dtype=np.float64)
    x = np.zeros((n, 1, length), np.float32)
    for b in range(n):
        sig = sum(r.uniform(0.2, 1.0) * np.sin(2 * np.pi * f * t / 100.0 + r.uniform(0, 2 * np.pi)) for f in (1.5, 6.0, 10.0, 13.0))
        x[b, 0, pad:length - pad] = np.clip(0.5 + 0.1 * sig + 0.05 * r.standard_normal(t.shape), 0, 1)
    return x


class WindowLoader:
    """Iterable of {'eeg': float32 (B,1,3072)} batches.  Source: a directory of pre-processed per-recording .npy files
    (one channel x samples, as written by the reference's preprocessing) or, when absent / --synthetic_windows is given,
    synthetic windows.  Each item = min-max normalise the recording, random 3000-sample crop, 36-sample zero pad
    (dataset.py:12-19); recordings are cached in memory after the first read."""

    def __init__(self, path_pre_processed, batch_size, n_synthetic=0, seed=0, drop_last=False, shuffle=True):
        self.batch_size, self.drop_last, self.shuffle = batch_size, drop_last, shuffle
        self.rng = np.random.default_rng(seed)
        files = sorted(glob.glob(os.path.join(path_pre_processed or "", "**", "*.npy"), recursive=True)) if path_pre_processed else []
        if files and not n_synthetic:
            self.recordings = []
            for f in files:
                a = np.load(f).astype(np.float32).reshape(-1)
                a = a * (1 + 1e6)
                lo, hi = float(a.min()), float(a.max())
                self.recordings.append((a - lo) / (hi - lo + 1e-12))
            self.windows = None
            self.n = len(self.recordings)
        else:
            self.windows = synthetic_windows(n_synthetic or 4 * batch_size, seed)
            self.n = len(self.windows)

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def _item(self, i):
        if self.windows is not None:
            return self.windows[i]
        a = self.recordings[i]
        s = int(self.rng.integers(0, max(1, a.shape[0] - 3000)))
        w = np.zeros((1, 3072), np.float32)
        w[0, 36:3036] = a[s:s + 3000]
        return w

    def __iter__(self):
        order = self.rng.permutation(self.n) if self.shuffle else np.arange(self.n)
        for k in range(len(self)):
            idx = order[k * self.batch_size:(k + 1) * self.batch_size]
            yield {"eeg": torch.from_numpy(np.stack([self._item(int(i)) for i in idx]))}
