"""CPU: the committed training-trajectory fixtures (tests/golden/ldm_traj_c2.json, aekl_traj_c1.json) are what the oracle produces --
their first steps are recomputed here from the seeds stored in the files (the full runs are tests/golden/make_ldm_traj.py /
make_aekl_traj.py; the HIP engines replay all steps in tests/test_gpu_zz_convergence.py)."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden")); sys.path.insert(0, os.path.dirname(HERE))


def _load(name):
    with open(os.path.join(HERE, "golden", name)) as fh:
        return json.load(fh)


def test_ldm_trajectory_fixture_is_the_oracles():
    import make_ldm_traj as M
    from param_gen import gen_param, eeg_windows, normal, timesteps
    import oracle.losses as Ls, oracle.steps as S, oracle.unet as U
    g = _load("ldm_traj_c2.json")
    assert g["steps"] == len(g["loss"]) == M.STEPS and g["batch"] == M.B
    assert g["loss"][-1] < 0.05 * g["loss"][0]                      # the run it records does train
    sd = {k: torch.from_numpy(gen_param(g["param_seed"], k, s)) for k, s in U.unet_param_shapes(M.CFG).items()}
    acp = Ls.alphas_cumprod(*g["schedule"])
    pool = torch.from_numpy(eeg_windows(g["pool"], seed=g["latent_seed"], length=768))
    opt = {}
    for i in (1, 2):
        s = ((i - 1) * g["batch"]) % g["pool"]
        nz = torch.from_numpy(normal((g["batch"], 1, 768), seed=g["noise_seed_base"] + i)); t = torch.from_numpy(timesteps(g["batch"], seed=g["t_seed_base"] + i))
        l, grads, _ = S.ldm_train_step(sd, M.CFG, acp, pool[s:s + g["batch"]], nz, t)
        sd = S.adam_update(sd, grads, opt, g["lr"], i)
        assert abs(float(l) - g["loss"][i - 1]) <= 1e-4 * g["loss"][i - 1], (i, float(l), g["loss"][i - 1])     # thread-count dependent summation order only


def test_ldm_trajectory_fixture_equals_the_reference_loops_own_trajectory():
    """tests/golden/make_ldm_traj_reference.py ran the REFERENCE's UNetModel through the reference's train-step body (training.py:419-443,
    torch.optim.Adam) on the fixture's seeds: its fp32 losses are the oracle's fixture to 1e-5 over all 30 steps (measured 3.5e-7) -- the
    trajectory the engines are held to is the reference loop's own, not only the oracle's."""
    g, r = _load("ldm_traj_c2.json"), _load("ldm_traj_c2_reference.json")
    assert len(r["loss_fp32"]) == len(g["loss"]) == 30
    for i, (a, b) in enumerate(zip(r["loss_fp32"], g["loss"])):
        assert abs(a - b) <= 1e-5 * b, (i + 1, a, b)
    # the reference's own reduced-precision runs: bf16 autocast stays within 2 % of fp32 at every step; fp16 autocast + GradScaler skips its
    # first two steps (scale 65536 -> 16384), so its curve runs two optimiser steps behind
    assert max(abs(a - b) / b for a, b in zip(r["loss_bf16"], r["loss_fp32"])) < 2e-2
    assert r["f16_scaler_backoffs"] == 2 and abs(r["loss_f16"][0] - r["loss_fp32"][0]) < 2e-3 * r["loss_fp32"][0]
    assert r["f16_scale1024_backoffs"] == 0 and max(abs(a - b) / b for a, b in zip(r["loss_f16_scale1024"], r["loss_fp32"])) < 2e-2


import pytest


@pytest.mark.parametrize("fixture", ["aekl_traj_c1.json", "aekl_traj_thin.json"])
def test_aekl_trajectory_fixture_is_the_oracles(fixture):
    import make_aekl_traj as M
    from param_gen import gen_param, eeg_windows, normal
    import oracle.aekl as A, oracle.steps as S
    g = _load(fixture)
    acfg = dict(M.ACFG, num_channels=g["num_channels"])
    assert g["steps"] == len(g["losses"]) == M.STEPS
    assert g["losses"][-1]["recons"] < 0.2 * g["losses"][0]["recons"]
    st = {"ae": {k: torch.from_numpy(gen_param(g["param_seeds"][0], k, s)) for k, s in A.aekl_param_shapes(acfg).items()},
          "d": {k: torch.from_numpy(gen_param(g["param_seeds"][1], k, s)) for k, s in A.disc_param_shapes(M.DCFG).items()}}
    xs = torch.from_numpy(eeg_windows(g["pool"], seed=g["window_seed"]))
    ew = torch.from_numpy(normal((g["batch"], 1, 768), seed=g["eps_seed_base"] + 1))
    w = g["weights"]
    l, *_ = S.aekl_train_step(st["ae"], acfg, st["d"], M.DCFG, xs[:g["batch"]], ew, w["adv"], w["kl"], w["spectral"], True, g["lr"][0], g["lr"][1], 1, {}, {})
    for k, v in g["losses"][0].items():
        assert abs(float(l[k]) - v) <= 1e-4 * abs(v) + 1e-6, (k, float(l[k]), v)


def test_dm_trajectory_fixture_is_the_oracles():
    import make_dm_traj as M
    from param_gen import gen_param, eeg_windows, normal, timesteps
    import oracle.losses as Ls, oracle.steps as S, oracle.unet as U
    g = _load("dm_traj_c5.json")
    assert g["steps"] == len(g["loss"]) == M.STEPS and g["loss"][-1] < 0.05 * g["loss"][0]
    sd = {k: torch.from_numpy(gen_param(g["param_seed"], k, s)) for k, s in U.unet_param_shapes(M.CFG).items()}
    acp = Ls.alphas_cumprod(*g["schedule"])
    pool = torch.from_numpy(eeg_windows(g["pool"], seed=g["window_seed"]))
    nz = torch.from_numpy(normal((g["batch"], 1, 3072), seed=g["noise_seed_base"] + 1)); t = torch.from_numpy(timesteps(g["batch"], seed=g["t_seed_base"] + 1))
    l, _grads, _ = S.dm_train_step(sd, M.CFG, acp, pool[:g["batch"]], nz, t, spectral_weight=g["spectral_weight"], spectral_loss=True)
    assert abs(float(l) - g["loss"][0]) <= 1e-4 * g["loss"][0], (float(l), g["loss"][0])
