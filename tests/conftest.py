import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running")


@pytest.fixture(scope="session")
def native():
    """Build (if stale) and load the product libraries + the CPU checker."""
    from mitsuba2_amd import build
    if not os.environ.get("MIW_TEST_NO_BUILD"):                   # (tools/sanitize_cpu.sh runs the tests under LD_PRELOAD=libasan: no compiler runs in there)
        build.build_all(oracle=True)
    from mitsuba2_amd import api
    api.host_lib()
    return api


@pytest.fixture(scope="session")
def oracle(native):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    return oracle_py.load()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


SRGB_COEFF = os.path.join(ROOT, "mitsuba2_amd", "data", "srgb.coeff")


@pytest.fixture()
def spectral(native):
    """Switches the host layer to the scalar_spectral variant for one test (mitsuba.set_variant)."""
    if not os.path.exists(SRGB_COEFF):
        pytest.skip("mitsuba2_amd/data/srgb.coeff missing (generated with the reference's ext/rgb2spec by mitsuba2_amd.build)")
    native.set_variant("scalar_spectral")
    native.set_srgb_model(SRGB_COEFF)
    yield native
    native.set_variant("scalar_rgb")


@pytest.fixture(scope="session")
def oracle_spectral(native):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    return oracle_py.load("scalar_spectral")
