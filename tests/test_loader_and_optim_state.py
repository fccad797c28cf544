"""CPU: (a) WindowLoader against the numpy restatement of the reference transform chain (oracle/dataset.py <-> dataset.py:10-30)
on a synthetic "night"; id-CSV split handling and rank sharding.  (b) the Adam checkpoint wire format against torch.optim.Adam's own
state_dict (train_autoencoderkl.py:320-329, training/training.py:381-388)."""
import os
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle.dataset import get_trans_edfx


def _night(seed, n):
    r = np.random.default_rng(seed)
    t = np.arange(n) / 100.0
    return (40e-6 * np.sin(2 * np.pi * 1.2 * t) + 15e-6 * r.standard_normal(n) + 5e-6).astype(np.float64)[None]   # volts, (1, n) as preprocessing writes


@pytest.fixture()
def nights(tmp_path):
    names = [f"SC40{i}1E0-PSG-Fpz-Cz" for i in range(5)]
    for i, nm in enumerate(names):
        np.save(tmp_path / f"{nm}.npy", _night(i, 20000 + 1000 * i))
    for split, rows in (("train", names[:3]), ("valid", names[3:])):
        with open(tmp_path / f"ids_{split}.csv", "w") as f:
            f.write(",subject,night,age,gender,LightsOff,FILE_NAME,FILE_NAME_EEG\n")
            for j, nm in enumerate(rows):
                f.write(f"{j},{j},1,33,1,0:38,{nm[:8]},{nm}\n")
    return tmp_path, names


def test_window_loader_matches_reference_transform_chain(nights):
    from eegldm.entry.common import WindowLoader
    d, names = nights
    starts = [0, 17000, 19000 - 3000 + 1000]          # incl. the last valid start of recording 0 (20000 - 3000 = 17000)
    ld = WindowLoader(str(d), batch_size=3, path_ids=str(d / "ids_train.csv"), shuffle=False, crop_starts=starts)
    assert len(ld) == 1
    batch = next(iter(ld))["eeg"]
    assert batch.dtype == torch.float32 and tuple(batch.shape) == (3, 1, 3072)
    for i in range(3):
        want = get_trans_edfx(np.load(d / f"{names[i]}.npy"), starts[i])
        np.testing.assert_array_equal(batch[i].numpy(), want.astype(np.float32))
        assert (batch[i, 0, :36] == 0).all() and (batch[i, 0, -36:] == 0).all()
        assert 0.0 <= float(batch[i].min()) and float(batch[i].max()) <= 1.0
    # the whole-night min-max means SOME window of the night touches 0 and 1, not every window
    rec = ld._recording(0)
    assert rec.min() == 0.0 and rec.max() == 1.0 and rec.dtype == np.float32


def test_window_loader_random_crops_cover_every_valid_start(nights):
    from eegldm.entry.common import WindowLoader
    d, _names = nights
    np.save(d / "short.npy", _night(9, 3002))
    with open(d / "ids_short.csv", "w") as f:
        f.write("FILE_NAME_EEG\nshort\n")
    ld = WindowLoader(str(d), batch_size=1, path_ids=str(d / "ids_short.csv"), seed=3, windows_per_recording=64)
    rec = ld._recording(0)
    seen = set()
    for b in ld:
        w = b["eeg"][0, 0, 36:3036].numpy()
        seen.add(next(s for s in range(3) if np.array_equal(w, rec[s:s + 3000])))
    assert seen == {0, 1, 2}                           # RandSpatialCrop draws from [0, n - roi] inclusive
    assert len(ld) == 64


def test_window_loader_splits_and_shards(nights):
    from eegldm.entry.common import WindowLoader, read_ids
    d, names = nights
    assert [os.path.basename(f) for f in read_ids(str(d / "ids_valid.csv"), str(d))] == [n + ".npy" for n in names[3:]]
    tr = WindowLoader(str(d), 8, path_ids=str(d / "ids_train.csv"))
    va = WindowLoader(str(d), 8, path_ids=str(d / "ids_valid.csv"))
    assert set(tr.files).isdisjoint(va.files) and len(tr.files) == 3 and len(va.files) == 2
    shards = [WindowLoader(str(d), 8, path_ids=str(d / "ids_train.csv"), shard=(r, 2)).files for r in range(2)]
    # 3 recordings over 2 ranks: equal shard lengths (every rank runs the same number of all-reduces per epoch), the union is the whole
    # split, and the one repeat is the wrap-around (torch DistributedSampler's drop_last=False convention)
    assert len(shards[0]) == len(shards[1]) == 2 and set(shards[0]) | set(shards[1]) == set(tr.files)
    assert shards[0] == [tr.files[0], tr.files[2]] and shards[1] == [tr.files[1], tr.files[0]]
    loaders = [WindowLoader(str(d), 1, path_ids=str(d / "ids_train.csv"), shard=(r, 2), drop_last=True) for r in range(2)]
    assert len(loaders[0]) == len(loaders[1])
    even = [WindowLoader(str(d), 8, path_ids=str(d / "ids_valid.csv"), shard=(r, 2)).files for r in range(2)]
    assert set(even[0]).isdisjoint(even[1]) and sorted(even[0] + even[1]) == sorted(va.files)       # N % world == 0: a plain partition
    with pytest.raises(FileNotFoundError):
        with open(d / "ids_bad.csv", "w") as f:
            f.write("FILE_NAME_EEG\nnot_there\n")
        WindowLoader(str(d), 8, path_ids=str(d / "ids_bad.csv"))
    with pytest.raises(ValueError, match="FILE_NAME_EEG"):
        WindowLoader(str(d), 8, path_ids="/root/repo/tests/golden/make_golden_cases.py")
    # no ids: every recording in the directory; constant recording normalises to zeros (monai rescale_array)
    assert len(WindowLoader(str(d), 8).files) == 5
    from eegldm.entry.common import normalise_recording
    assert not normalise_recording(np.full((1, 10), 3e-5)).any()


def test_rng_seeds_do_not_collide_across_ranks_and_roles():
    from eegldm.entry.common import rng_seed
    seeds = {rng_seed(42, role, rank, 8) for role in (1, 2, 3, 4, 5) for rank in range(8)}
    assert len(seeds) == 40
    assert rng_seed(42, 3, 0) != rng_seed(42, 2, 1)        # the round-1 collision: seed+13+r == seed+12+(r+1)


# ------------------------------------------------------------------ Adam wire format
def _entries(shapes):
    e, off = OrderedDict(), 0
    for k, s in shapes.items():
        n = int(np.prod(s))
        e[k] = (off, n, tuple(s)); off += (n + 7) // 8 * 8
    return e, off


def _pack(entries, tensors, total):
    flat = torch.zeros(total)
    for k, (o, n, shape) in entries.items():
        t = tensors[k]
        flat[o:o + n] = (t.permute(2, 0, 1) if len(shape) == 3 else t).reshape(-1)
    return flat


def test_adam_state_roundtrip_with_torch_optim_adam():
    from eegldm.training import flat_to_torch_adam_state, torch_adam_state_to_flat
    from oracle import aekl as A
    from param_gen import gen_param
    cfg = dict(num_channels=[4, 4, 16], latent_channels=1, in_channels=1, out_channels=1, num_res_blocks=2, norm_num_groups=1)
    shapes = A.aekl_param_shapes(cfg)
    entries, total = _entries(shapes)
    params = [torch.nn.Parameter(torch.from_numpy(gen_param(5, k, s))) for k, s in shapes.items()]
    opt = torch.optim.Adam(params, lr=5e-3)
    g = torch.Generator().manual_seed(1)
    for _ in range(3):
        for p in params:
            p.grad = torch.randn(p.shape, generator=g)
        opt.step()
    sd = opt.state_dict()
    m, v = torch.zeros(total), torch.zeros(total)
    step, hyper = torch_adam_state_to_flat(entries, sd, m, v)
    assert step == 3 and hyper["lr"] == 5e-3 and tuple(hyper["betas"]) == (0.9, 0.999)
    want_m = _pack(entries, {k: sd["state"][i]["exp_avg"] for i, k in enumerate(shapes)}, total)
    assert torch.equal(m, want_m)
    back = flat_to_torch_adam_state(entries, m, v, step, hyper["lr"])
    assert set(back["param_groups"][0]) >= set(sd["param_groups"][0])          # every key torch writes is there
    for i in sd["state"]:
        for name in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(back["state"][i][name], sd["state"][i][name])
        assert float(back["state"][i]["step"]) == float(sd["state"][i]["step"])
    # and torch itself accepts it and continues identically
    params2 = [torch.nn.Parameter(p.detach().clone()) for p in params]
    opt2 = torch.optim.Adam(params2, lr=1.0)
    opt2.load_state_dict(back)
    for p, p2 in zip(params, params2):
        gr = torch.randn(p.shape, generator=g); p.grad = gr.clone(); p2.grad = gr.clone()
    opt.step(); opt2.step()
    for p, p2 in zip(params, params2):
        assert torch.equal(p, p2)
    # old torch versions store `step` as a python int
    for st in sd["state"].values():
        st["step"] = 3
    assert torch_adam_state_to_flat(entries, sd, m, v)[0] == 3
    with pytest.raises(ValueError, match="covers"):
        torch_adam_state_to_flat(OrderedDict(list(entries.items())[:-1]), sd, m, v)
