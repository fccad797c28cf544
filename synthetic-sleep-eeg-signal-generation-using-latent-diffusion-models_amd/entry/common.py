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


WINDOW = 3000      # 30 s at 100 Hz (dataset.py:7-8)
BORDER = 36        # BorderPadD(spatial_border=[36]) (dataset.py:18)


def normalise_recording(a):
    """The deterministic head of get_trans (dataset.py:12-16): ScaleIntensityD(factor=1e6) = x * (1 + 1e6), then
    ScaleIntensityD(minv=0, maxv=1) = min-max over the WHOLE recording (a constant recording maps to zeros, as
    monai.transforms.utils.rescale_array does).  float32 in, float32 out, flattened to one channel."""
    a = np.asarray(a, dtype=np.float32).reshape(-1) * np.float32(1.0 + 1e6)
    lo, hi = a.min(), a.max()
    if hi == lo:
        return np.zeros_like(a)
    return (a - lo) / (hi - lo)


def crop_and_pad(rec, start):
    """RandSpatialCropD(roi_size=[3000], random_size=False) at offset `start`, then 36 zeros on each side -> (1, 3072)."""
    w = np.zeros((1, WINDOW + 2 * BORDER), np.float32)
    w[0, BORDER:BORDER + WINDOW] = rec[start:start + WINDOW]
    return w


def read_ids(path_ids, path_pre_processed, dataset="edfx"):
    """File list from an id CSV (column FILE_NAME_EEG), as get_datalist builds it (dataset.py:33-60): '.npy' suffix for edfx."""
    import csv
    with open(path_ids, newline="") as f:
        rows = list(csv.DictReader(f))
    if rows and "FILE_NAME_EEG" not in rows[0]:
        raise ValueError(f"{path_ids}: no FILE_NAME_EEG column (dataset.py:49)")
    final = ".npy" if dataset == "edfx" else ""
    return [os.path.join(path_pre_processed or "", r["FILE_NAME_EEG"] + final) for r in rows]


def shard_files(files, rank, world):
    """Rank `rank`'s slice of the file list, the SAME length on every rank: ceil(N / world) entries, taken round-robin and wrapped
    around the end of the list (what torch's DistributedSampler does with drop_last=False).  Equal lengths matter: every rank runs
    one gradient all-reduce per batch, so a rank with one more batch than the others would wait in a collective nobody else enters."""
    if world <= 1 or not files:
        return list(files)
    per = -(-len(files) // world)
    return [files[(rank + i * world) % len(files)] for i in range(per)]


class WindowLoader:
    """Iterable of {'eeg': float32 (B,1,3072)} batches -- the loader output contract of dataset.py:10-30,62-69.

    Source: the recordings named by an id CSV (`path_ids`, column FILE_NAME_EEG: the reference's train / valid / test split), or
    every *.npy under `path_pre_processed` when no CSV is given, or synthetic windows (`n_synthetic`).  Each item = normalise the
    whole recording (x * (1 + 1e6), min-max), random 3000-sample crop, 36-sample zero pad.  The reference re-reads and re-normalises
    a whole night (~23 MB) for every 12 KB window (PersistentDataset(cache_dir=None)); here a recording is read and normalised ONCE
    and kept in memory, and `windows_per_recording` crops are drawn from it per epoch (1 = the reference's epoch definition).
    `shard=(rank, world)` gives every data-parallel rank its own slice of the file list (an epoch is the data once, not world
    times; `shard_files`: equal length on every rank, so all ranks run the same number of steps per epoch).  A run that names real
    data (`path_ids` / `path_pre_processed`) and finds none raises instead of silently training on synthetic windows.  Crop offsets come from a numpy Generator seeded per loader (`crop_starts` can be injected for parity tests)."""

    def __init__(self, path_pre_processed, batch_size, n_synthetic=0, seed=0, drop_last=False, shuffle=True, path_ids=None,
                 dataset="edfx", shard=(0, 1), windows_per_recording=1, crop_starts=None):
        self.batch_size, self.drop_last, self.shuffle = batch_size, drop_last, shuffle
        self.rng = np.random.default_rng(seed)
        self.wpr = max(1, int(windows_per_recording))
        self.crop_starts = crop_starts
        self.recordings, self.windows, self.files = None, None, []
        if n_synthetic:
            files = []
        elif path_ids:
            files = read_ids(path_ids, path_pre_processed, dataset)
            missing = [f for f in files if not os.path.exists(f)]
            if missing:
                raise FileNotFoundError(f"{len(missing)} recordings listed in {path_ids} are missing, e.g. {missing[0]}")
        else:
            files = sorted(glob.glob(os.path.join(path_pre_processed or "", "**", "*.npy"), recursive=True)) if path_pre_processed else []
        rank, world = shard
        if (path_ids or path_pre_processed) and not n_synthetic and not files:
            raise FileNotFoundError(f"no recordings found (path_ids={path_ids!r}, path_pre_processed={path_pre_processed!r}); "
                                    "refusing to fall back to synthetic windows for a run that named real data")
        files = shard_files(files, rank, world)
        if files:
            self.files = files
            self.recordings = [None] * len(files)          # read + normalised on first use, then cached
            self.n = len(files) * self.wpr
        else:
            self.windows = synthetic_windows(n_synthetic or 4 * batch_size, seed)
            self.n = len(self.windows)

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def _recording(self, r):
        if self.recordings[r] is None:
            rec = normalise_recording(np.load(self.files[r]))
            if rec.shape[0] < WINDOW:
                raise ValueError(f"{self.files[r]}: {rec.shape[0]} samples, need at least {WINDOW}")
            self.recordings[r] = rec
        return self.recordings[r]

    def _item(self, i):
        if self.windows is not None:
            return self.windows[i]
        rec = self._recording(i // self.wpr)
        if self.crop_starts is not None:
            s = int(self.crop_starts[i])
        else:
            s = int(self.rng.integers(0, rec.shape[0] - WINDOW + 1))       # RandSpatialCrop: every valid start, last one included
        return crop_and_pad(rec, s)

    def __iter__(self):
        order = self.rng.permutation(self.n) if self.shuffle else np.arange(self.n)
        for k in range(len(self)):
            idx = order[k * self.batch_size:(k + 1) * self.batch_size]
            yield {"eeg": torch.from_numpy(np.stack([self._item(int(i)) for i in idx]))}


def rng_seed(base_seed, role, rank=0, world=1):
    """Distinct Philox key per (role, rank): base * 2^20 + role * 4096 + rank.  Roles: 1 timesteps, 2 posterior eps, 3 diffusion
    noise, 4 autoencoder eps, 5-7 validation draws, 8 training loader (crop / shuffle / synthetic windows), 9 validation loader.  (seed + role + rank collides across ranks: rank r's noise stream == rank r+1's eps stream.)"""
    assert 0 <= rank < 4096 and 0 <= role < 256
    return (int(base_seed) << 20) + (int(role) << 12) + int(rank)
