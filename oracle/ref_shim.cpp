// C entry points around the two translation units of the reference's hot path that compile without Eigen / PCL /
// OpenCV: Histogram (monolidar_fusion/src/Histogram.cpp) and TresholdDepthGlobal (src/TresholdDepthGlobal.cpp).
// Built by `make -C oracle ref` FROM THE SOURCES UNDER /root/reference (never copied into this repository) into
// oracle/_ref/libmld_ref.so, and used by tests/test_reference_parts.py to check the restatement against the
// reference's own object code.  Test infrastructure only.
#include "Histogram.h"
#include "TresholdDepthGlobal.h"

extern "C" {

// Histogram(binWitdh, binCount); AddElement(value) for every value; counts[b] = binElemCount(b)
int ref_histogram_counts(const double* values, int n, double bin_width, int bin_count, int* counts_out) {
    Mono_Lidar::Histogram h(bin_width, bin_count);
    for (int i = 0; i < n; i++) h.AddElement(values[i]);
    for (int b = 0; b < bin_count; b++) counts_out[b] = h.binElemCount(b);
    return 0;
}

// TresholdDepthGlobal(mode, min, max).CheckInDepth(depth): returns eTresholdResult, depth updated in place
int ref_threshold_global(int mode, double min_value, double max_value, double* depth) {
    Mono_Lidar::TresholdDepthGlobal t(static_cast<Mono_Lidar::eTresholdDepthMode>(mode), min_value, max_value);
    return static_cast<int>(t.CheckInDepth(*depth));
}

}  // extern "C"
