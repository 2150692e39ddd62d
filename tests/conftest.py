import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def have_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker).  Built on demand from oracle/vr_oracle.c."""
    import oracle
    oracle.lib()
    return oracle


@pytest.fixture(scope="session")
def vr():
    import volume_renderer_amd
    return volume_renderer_amd


# (kernel variant, exact empty-space skipping): 0 = global gathers (baseline), 1 = per-wave LDS staging (default)
KERNEL_VARIANTS = [(0, 0), (1, 0), (1, 1)]


@pytest.fixture(params=KERNEL_VARIANTS, ids=[f"k{k}{'s' if s else ''}" for k, s in KERNEL_VARIANTS])
def rm(vr, request):
    """A fresh RayMarcher on GPU 0, once per kernel configuration.  GPU tests fail loudly if the HIP library or
    the device is missing."""
    if not have_gpu():
        pytest.skip("no /dev/kfd: GPU tests only run on the GPU box (-m gpu)")
    r = vr.RayMarcher(0, 256, 256)
    r.set_option(vr.OPT_KERNEL, request.param[0])
    r.set_option(vr.OPT_SKIP_EMPTY, request.param[1])
    yield r
    r.close()


def sphere_cam(orc, ix, iy, xdiv=8, ydiv=8, radius=2.5, aspect=1.0):
    return orc.camera_uniform(orc.sphere_position(ix, iy, xdiv, ydiv, radius), aspect=aspect)


def image_metrics(a, b):
    """similarity.py:59-62 restated: MSE over the uint8 arrays (float64), rmse = sqrt(mse); plus max |delta|."""
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return {"max": int(np.abs(d).max()), "rmse_lsb": mse ** 0.5, "rmse_norm": mse ** 0.5 / 255.0}


@pytest.fixture
def rm1(vr):
    """One RayMarcher with the default kernel configuration (tests that do not depend on the march variant)."""
    if not have_gpu():
        pytest.skip("no /dev/kfd: GPU tests only run on the GPU box (-m gpu)")
    r = vr.RayMarcher(0, 256, 256)
    yield r
    r.close()
