import os
import sys

import pytest

# torch bundles its own HIP runtime (ROCm 7.0); importing it before libicp_mi355x.so is
# dlopen'ed makes the whole process use that one runtime instead of loading a second copy
# from /opt/rocm next to it.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session", params=["exact_f64", "mfma_bf16", "mfma_pruned"])
def gpu_ctx(request):
    """One context per search engine for the whole GPU session; every parity test runs
    against both (they must return identical indices).  Fails loudly (no skip, no
    fallback) when the HIP library or the device is missing."""
    from lidar_slam_from_scratch_amd import build, capi
    build.build_library()
    search = {"exact_f64": capi.SEARCH_EXACT_F64, "mfma_bf16": capi.SEARCH_MFMA_BF16,
              "mfma_pruned": capi.SEARCH_MFMA_PRUNED}[request.param]
    ctx = capi.Context(device=0, search=search, profile=2)
    ctx.engine = request.param
    yield ctx
    ctx.close()
