"""The restatement against the reference's OWN object code, for the two translation units of the hot path that build
without Eigen / PCL / OpenCV: Histogram.cpp and TresholdDepthGlobal.cpp (compiled from /root/reference by
`make -C oracle ref` into oracle/_ref/, see oracle/ref_shim.cpp).  Nothing here reads /root/reference at run time; the
tests are skipped where the library has not been built."""
import ctypes as C

import numpy as np
import pytest

from mono_lidar_depth_amd import capi
from oracle import oracle

ref = oracle.load_reference_parts()
pytestmark = pytest.mark.skipif(ref is None, reason="oracle/_ref/libmld_ref.so not built (needs /root/reference)")


def _ref_counts(values, bin_width, bin_count):
    v = np.ascontiguousarray(values, dtype=np.float64)
    out = np.zeros(bin_count, dtype=np.int32)
    ref.ref_histogram_counts(v.ctypes.data, v.size, float(bin_width), int(bin_count), out.ctypes.data)
    return out


@pytest.mark.parametrize("bin_width,bin_count", [(0.3, 40), (0.1, 500), (1.0, 7), (2.5, 2), (0.3, 1)])
def test_histogram_binning_equals_the_reference_class(bin_width, bin_count):
    rng = np.random.default_rng(int(bin_width * 100) + bin_count)
    hi = bin_width * bin_count
    values = np.concatenate([
        rng.uniform(0, 1.5 * hi, 4000),                                   # inside and beyond the last bin
        np.arange(0, bin_count + 3) * bin_width,                          # exact bin borders
        np.nextafter(np.arange(1, bin_count + 1) * bin_width, 0.0),       # one ulp below a border
        np.array([0.0, -0.0, -0.4 * bin_width, -3.7 * bin_width, 1e10, 1e12, 1e300, 999.0]),  # signs, clamp at 1e10
    ])
    assert np.array_equal(oracle.histogram_counts(values, bin_width, bin_count), _ref_counts(values, bin_width, bin_count))


def test_histogram_kat_through_the_reference_class():
    """The depth list of the reference's own test Histogram.FilterPointsMinDistBlob (test_monolidar_fusion.cpp:306-374),
    binned by the reference's class and by the restatement."""
    depths = [1.1, 1.2, 1.3, 4.1, 4.2, 4.3, 4.4, 8.0, 8.1]
    assert np.array_equal(oracle.histogram_counts(depths, 1.0, 10), _ref_counts(depths, 1.0, 10))


@pytest.mark.parametrize("mode", [0, 1])
def test_global_threshold_equals_the_reference_class(mode):
    rng = np.random.default_rng(mode)
    P = capi.params_c0().replace(treshold_depth_mode=mode, treshold_depth_min=2, treshold_depth_max=60)
    depths = np.concatenate([rng.uniform(-10, 120, 3000), [2.0, 60.0, np.nextafter(2.0, 0), np.nextafter(60.0, 100), -1.0, 0.0]])
    for d in depths:
        r0, d0 = oracle.threshold_global(P, float(d))
        dd = C.c_double(float(d))
        r1 = ref.ref_threshold_global(mode, 2.0, 60.0, C.byref(dd))
        assert r0 == r1 and d0 == dd.value, d
