import os
import sys

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.abspath(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a device kernel that never returns must fail ONE test, not hang the whole run (pytest-timeout, when installed)
    if config.pluginmanager.hasplugin("timeout") and not getattr(config.option, "timeout", None):
        config.option.timeout = 300


@pytest.fixture(scope="session")
def qo():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    oracle.set_threads(min(8, oracle.max_threads()))
    return oracle


@pytest.fixture(scope="session")
def hip():
    """A handle on libquatro_hip.so; GPU tests call the product ONLY through this C ABI binding."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from quatro_amd import lib as ql
    h = ql.Handle(0)
    yield h
    h.close()


@pytest.fixture(scope="session")
def small_pair():
    """A reduced synthetic scan pair (fewer beams' worth of points) that the oracle handles in ~1 s."""
    from quatro_amd import synth
    s, t, T = synth.kitti64_pair(2)
    return s, t, T


def bits32(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
