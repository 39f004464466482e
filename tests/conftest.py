import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_lib():
    """Build (if stale) and load libmvin_hip.so."""
    from mvin_amd import _lib, build
    build.build()
    return _lib.load()


# The whole-pass single-launch kernel (mvin_score_small_fwd) is the product's default for batches of at most 1 024
# pairs (MVIN.small_max_batch; the entry point itself takes any batch and tests drive it to 16 384).  Most GPU test modules exist to pin ONE of the other kernels (packed / split / wave-per-parent fused kernels,
# the key-addressing family, the tail, the native multi-launch schedule) through small MVIN forwards: those keep their
# kernels (MVIN_SMALL=0, read when a model is built).  The modules below run the product default.
SMALL_KERNEL_MODULES = {"test_gpu_small", "test_gpu_ref_pins", "test_gpu_api", "test_gpu_properties", "test_gpu_dist"}


@pytest.fixture(autouse=True)
def _pin_kernel_under_test(request, monkeypatch):
    mod = request.module.__name__.rsplit(".", 1)[-1]
    if mod.startswith("test_gpu") and mod not in SMALL_KERNEL_MODULES:
        monkeypatch.setenv("MVIN_SMALL", "0")
        # ... and the form of the two deepest levels they pin: the projected-tables form (taken by batch size: B K >= 16 n_entity)
        # has its own module; an explicit MVIN_PRJ in the environment (a forced run of the whole suite) is respected
        # -- and so does test_gpu_bench_scale, which must run what bench.py times: the AUTOMATIC rule
        if mod not in ("test_gpu_prj", "test_gpu_bench_scale") and "MVIN_PRJ" not in os.environ:
            monkeypatch.setenv("MVIN_PRJ", "0")
    yield
