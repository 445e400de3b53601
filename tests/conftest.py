import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build libmld_hip.so / the oracle if they are missing (hipcc cross-compiles without a GPU)."""
    from mono_lidar_depth_amd import capi
    from oracle import oracle
    stale = False
    if capi.LIB_PATH.exists():
        try:  # through capi.load(): torch's HIP runtime has to be in the process before this library
            stale = capi.load().mld_abi_version() != capi.MLD_ABI_VERSION
        except (OSError, AttributeError, RuntimeError):
            stale = True
    if stale or not capi.LIB_PATH.exists() or not oracle.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
        capi._lib = None  # load the rebuilt libraries
        capi._lib_ab = None
    yield


@pytest.fixture(params=["default", "fused", "wave-only", "dense"])
def feature_kernel_path(request, monkeypatch):
    """Every test runs four ways:
    default     the shipped library (libmld_hip.so): batches on k_classify + k_feature_fused (+ k_feature_wave for long
                lists), single frames of up to 16 384 features on the wave-cooperative kernel
    fused       the lane-per-feature kernel for single frames as well: the test build of the same sources
                (libmld_hip_ab.so, -DMLD_AB_SWITCHES) with MLD_FORCE_THREAD_PATH=1, read by its mld_create
    wave-only   every feature, batches included, through the wave-cooperative kernel (MLD_FORCE_WAVE_PATH=1)
    dense       as "fused" with the list capacities of a dense cloud (MLD_K1MAX=48): the DENSE instantiation of the
                lane-per-feature kernel (two wavefronts per SIMD, in-register corner search up to 24 points)"""
    if request.param != "default":
        monkeypatch.setenv("MLD_FORCE_WAVE_PATH" if request.param == "wave-only" else "MLD_FORCE_THREAD_PATH", "1")
        if request.param == "dense":
            monkeypatch.setenv("MLD_K1MAX", "48")
        from mono_lidar_depth_amd import capi
        monkeypatch.setattr(capi, "_lib", capi.load_ab())
    yield request.param
