"""CPU: the U-Sleep / FID oracle (oracle/usleep.py, oracle/fid.py) against the golden vectors generated from the imported reference
class (tests/golden/make_golden_r3.py -> usleep_d12.npz) and against closed forms of the Frechet distance (monai-generative's
FIDMetric source is absent: 'parity unpinned' for that formula, see oracle/fid.py)."""
import os

import numpy as np
import pytest
import torch

from param_gen import eeg_windows, gen_param, normal


@pytest.fixture(scope="module")
def usleep_golden(golden_dir):
    return np.load(os.path.join(golden_dir, "usleep_d12.npz"))


def _sd(g):
    from oracle import usleep as OU
    shapes = OU.usleep_param_shapes()
    assert list(shapes.keys()) == [str(k) for k in g["keys"]]                      # 265 state-dict keys, reference order
    assert [",".join(str(d) for d in s) for s in shapes.values()] == [str(s) for s in g["shapes"]]
    return {k: torch.from_numpy(gen_param(int(g["seeds"][0]), k, s)) for k, s in shapes.items()}


def test_usleep_structure_matches_reference():
    from oracle import usleep as OU
    assert OU.usleep_channels() == [2, 6, 9, 11, 15, 20, 28, 40, 55, 77, 108, 152, 214, 302]      # usleep.py:165-172
    assert OU.usleep_kernel_size(100) == 7 and OU.usleep_kernel_size(128) == 9                       # round(9/128 * sfreq)
    shapes = OU.usleep_param_shapes()
    n = sum(int(np.prod(s)) for k, s in shapes.items() if "running" not in k and "num_batches" not in k)
    assert n == 2482011
    with pytest.raises(ValueError):
        OU.usleep_kernel_size(114)                                                                   # even kernel (usleep.py:157-163)


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_usleep_forward_vs_reference_golden(usleep_golden, mode):
    from oracle import usleep as OU
    g = usleep_golden
    sd = _sd(g)
    x = torch.from_numpy(normal((3, 2, 3000), seed=int(g["seeds"][1])))
    run = {}
    y, dec, bottom = OU.usleep_forward(sd, x, training=(mode == "train"), running=run)
    for name, got in (("y_pred", y), ("decoder", dec), ("bottom", bottom)):
        want = torch.from_numpy(g[f"{mode}:{name}"])
        assert got.shape == want.shape
        err = float((got - want).abs().max()); scale = float(want.abs().max())
        assert err < 2e-5 * max(1.0, scale), (name, err, scale)
    if mode == "train":
        for k in ("encoder.0.block_prepool.2", "encoder.11.block_prepool.2", "bottom.2", "decoder.0.block_preskip.3", "decoder.11.block_postskip.2"):
            for leaf in ("running_mean", "running_var"):
                want = torch.from_numpy(g[f"train:{k}.{leaf}"])
                assert float((run[f"{k}.{leaf}"] - want).abs().max()) < 1e-5 * max(1.0, float(want.abs().max())), (k, leaf)


def test_fid_features_vs_reference_golden(usleep_golden):
    from oracle import usleep as OU
    g = usleep_golden
    w = torch.from_numpy(eeg_windows(6, seed=int(g["seeds"][2])))
    f = OU.fid_features(_sd(g), w)
    want = torch.from_numpy(g["fid_features:eval"])
    assert f.shape == want.shape == (6, 302)
    assert float((f - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))
    # already-cropped (B,1,3000) windows (the sampler's sample_{i}.npy files, compute_fid.py:396-404) take the same path
    assert torch.equal(OU.fid_features(_sd(g), w[:, :, 36:-36]), f)


def test_frechet_distance_closed_forms():
    from oracle import fid as OF
    r = np.random.default_rng(3)
    a = torch.from_numpy(r.standard_normal((400, 12)))
    assert abs(float(OF.fid(a, a))) < 1e-8                                              # identical sets
    shift = torch.from_numpy(r.standard_normal(12))
    assert abs(float(OF.fid(a + shift, a)) - float(shift.dot(shift))) < 1e-8            # same covariance: |mu1 - mu2|^2
    # commuting covariances: FID = |dmu|^2 + sum_i (sqrt(l1_i) - sqrt(l2_i))^2
    mu = torch.zeros(5, dtype=torch.double); l1 = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0], dtype=torch.double); l2 = torch.tensor([2.0, 2.0, 0.5, 9.0, 5.0], dtype=torch.double)
    want = float(((l1.sqrt() - l2.sqrt()) ** 2).sum())
    assert abs(float(OF.frechet_distance(mu, torch.diag(l1), mu, torch.diag(l2))) - want) < 1e-10
    q, _ = np.linalg.qr(r.standard_normal((5, 5))); q = torch.from_numpy(q)
    assert abs(float(OF.frechet_distance(mu, q @ torch.diag(l1) @ q.T, mu, q @ torch.diag(l2) @ q.T)) - want) < 1e-9
    # symmetric in its arguments, unbiased covariance
    b = torch.from_numpy(r.standard_normal((300, 12)) * 1.5 + 0.2)
    assert abs(float(OF.fid(a, b)) - float(OF.fid(b, a))) < 1e-7 * float(OF.fid(a, b))
    assert torch.allclose(OF._cov(a), torch.from_numpy(np.cov(a.numpy(), rowvar=False)))
    with pytest.raises(ValueError):
        OF.fid(torch.zeros(2, 3, 4), torch.zeros(2, 3, 4))


def test_product_frechet_distance_matches_oracle_on_host():
    """eegldm.metrics.frechet_distance is host linear algebra (two symmetric eigen-decompositions in fp64): checked here without a GPU
    against the oracle's scipy.linalg.sqrtm restatement of monai-generative's formula, on full-rank covariances."""
    from eegldm.metrics import frechet_distance
    from oracle import fid as OF
    r = np.random.default_rng(5)
    for d, n in ((8, 100), (40, 500), (302, 1200)):
        a = torch.from_numpy(r.standard_normal((n, d)) * (1 + r.random(d))); b = torch.from_numpy(r.standard_normal((n + 7, d)) * 0.8 + 0.1)
        want = float(OF.fid(a, b))
        got = frechet_distance(a.mean(0).numpy(), OF._cov(a).numpy(), b.mean(0).numpy(), OF._cov(b).numpy())
        assert abs(got - want) < 1e-8 * abs(want), (d, got, want)
    l1 = np.array([1.0, 2.0, 3.0]); l2 = np.array([4.0, 0.0, 3.0])                      # a singular covariance: exact closed form
    assert abs(frechet_distance(np.zeros(3), np.diag(l1), np.ones(3), np.diag(l2)) - (3.0 + ((np.sqrt(l1) - np.sqrt(l2)) ** 2).sum())) < 1e-12
