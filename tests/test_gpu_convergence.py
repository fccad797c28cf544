"""-m gpu: the production train step TRAINS -- 40 optimiser steps of the config_ldm.yaml UNet over frozen AutoencoderKL latents
(train_ldm.py:199-204 schedule and scale factor; training.py:399-452 loop) on a fixed pool of synthetic windows:

* the epsilon-MSE falls from ~1.0 (zero-initialised output conv) to well below it,
* the bf16 engine follows the fp32 engine's loss curve on the same seeds (Philox streams are engine-independent),
* torch's allocator does not grow over the run (every step reuses the context workspace).
A single-step parity test cannot see a gradient that is slightly wrong in a way that stalls training; this one does.
The long form (500 steps, batch 256) is tools/soak_ldm.py -> profiles/r03_soak_ldm.log.txt."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))


def test_ldm_training_converges_and_bf16_follows_fp32():
    import soak_ldm
    B, steps = 64, 40
    b = soak_ldm.run("bfloat16", steps, B, 768, 512, 10, 1e-4, True)
    f = soak_ldm.run("float32", steps, B, 768, 512, 10, 1e-4, False)
    cb, cf = dict(b["curve"]), dict(f["curve"])
    assert abs(cb[0] - 1.0) < 0.05 and abs(cf[0] - 1.0) < 0.05, (cb[0], cf[0])       # zero-initialised head: loss = E[noise^2]
    assert cb[steps - 1] < 0.6 * cb[0] and cf[steps - 1] < 0.6 * cf[0], (cb, cf)
    for i in cf:
        assert abs(cb[i] - cf[i]) / cf[i] < 0.03, (i, cb[i], cf[i])                  # measured: 0.05 % at step 25, 0.8 % at step 50 (B=256)
    assert b["alloc_first_last"][1] <= b["alloc_first_last"][0] and f["alloc_first_last"][1] <= f["alloc_first_last"][0]
    assert b["sample"]["finite"], b["sample"]       # 40 steps do not make a usable epsilon model (latent std ~40; 1.4 after 500 steps): only finiteness
