import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "small_frames.npz"))


@pytest.fixture(scope="session")
def scene_c2_small():
    """A reduced C2-style scene (320x240, ~150 superpixels) the oracle finishes in seconds."""
    from mcmc_amd.scene import make_scene
    return make_scene(width=320, height=240, n_planes=5, n_superpixels=130, seed=7)


@pytest.fixture(scope="session")
def gpu_ctx_factory():
    from mcmc_amd import host
    made = []

    def make(**kw):
        c = host.Context(**kw)
        made.append(c)
        return c
    yield make
    for c in made:
        c.close()
