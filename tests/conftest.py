import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def env_switches():
    """In-process A/B of the library's developer switches: `env_switches(EEGLDM_NO_CONV_SKINNY="1")` sets the variables and makes
    libeegldm re-read its cached switches (eegldm_debug_reload_env); everything is restored (and re-read again) at teardown.
    Contexts / models built before the call keep what they derived from the old values, so build them after."""
    import os
    from eegldm._lib import lib
    saved = {}

    def apply(**kv):
        for k, v in kv.items():
            saved.setdefault(k, os.environ.get(k))
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        lib.eegldm_debug_reload_env()

    yield apply
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    lib.eegldm_debug_reload_env()


@pytest.fixture(params=["default", "general_kernel"])
def conv_kernel_path(request, env_switches):
    """Small conv / linear shapes take the few-row kernel (conv_skinny.hip) by default; the second value runs the same cases on the
    general tiled GEMM (EEGLDM_NO_CONV_SKINNY=1) so its edge handling (tile edges inside samples, K tails, partial column tiles) stays covered."""
    if request.param == "general_kernel":
        env_switches(EEGLDM_NO_CONV_SKINNY="1")
    return request.param
