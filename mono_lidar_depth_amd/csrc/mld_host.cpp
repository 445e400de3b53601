// Host-side staging helpers of the C-ABI (no GPU call in this file).
//
// mld_pack_points_host: the reference's caller holds its clouds as pcl::PointXYZI records - 32 bytes per point
// (x, y, z, pad | intensity, pad, pad, pad; DepthEstimator.h:62-63, Transform_Cloud_LidarToCamera reads rows 0-2 of the
// 8-float map, DepthEstimator.cpp:169).  A driver that streams frames through pinned host buffers copies every cloud
// once on the host anyway; doing that copy as a 32 -> 16 byte repack halves what then crosses PCIe, the bound of the
// streamed path.
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/mld.h"

namespace {

void pack_range(float* dst, const unsigned char* src, int64_t i0, int64_t i1, int stride) {
    const bool stream = (reinterpret_cast<uintptr_t>(dst) & 15u) == 0;  // (16-byte records: every one is aligned then)
    if (stride == 16) {
        std::memcpy(dst + 4 * i0, src + 16 * i0, (size_t)(i1 - i0) * 16);
        return;
    }
#if defined(__SSE2__)
    for (int64_t i = i0; i < i1; i++) {
        const float* s = reinterpret_cast<const float*>(src + (size_t)i * (size_t)stride);
        const __m128 xyzp = _mm_loadu_ps(s);                                   // x y z pad
        const __m128 iv = _mm_load_ss(s + 4);                                  // intensity 0 0 0
        const __m128 zi = _mm_shuffle_ps(xyzp, iv, _MM_SHUFFLE(0, 0, 2, 2));   // z z i i
        const __m128 out = _mm_shuffle_ps(xyzp, zi, _MM_SHUFFLE(2, 0, 1, 0));  // x y z i
        if (stream)
            _mm_stream_ps(dst + 4 * i, out);  // (the destination is read next by the DMA engine, not by this core)
        else
            _mm_storeu_ps(dst + 4 * i, out);
    }
    if (stream) _mm_sfence();
#else  // hosts without SSE2: the same repack, scalar
    (void)stream;
    for (int64_t i = i0; i < i1; i++) {
        const float* s = reinterpret_cast<const float*>(src + (size_t)i * (size_t)stride);
        float* d = dst + 4 * i;
        d[0] = s[0];
        d[1] = s[1];
        d[2] = s[2];
        d[3] = s[4];
    }
#endif
}

}  // namespace

extern "C" int mld_pack_points_host(void* dst16, const void* src, int64_t n, int src_stride_bytes, int n_threads) {
    if (n < 0 || (n > 0 && (!dst16 || !src))) return MLD_ERR_INVALID_ARG;
    if (src_stride_bytes != 16 && src_stride_bytes != 32) return MLD_ERR_INVALID_ARG;
    if (n == 0) return MLD_OK;
    float* dst = static_cast<float*>(dst16);
    const unsigned char* s = static_cast<const unsigned char*>(src);
    // a thread is worth starting for ~64 K points (1-2 MB of source)
    const int64_t want = (n + 65535) / 65536;
    const int T = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(n_threads, 64), want));
    if (T == 1) {
        pack_range(dst, s, 0, n, src_stride_bytes);
        return MLD_OK;
    }
    // No C++ exception crosses the C ABI: when a helper thread cannot be started (std::system_error under a thread limit,
    // bad_alloc), the ranges that have no thread yet are packed by the calling thread.
    std::vector<std::thread> th;
    const int64_t per = ((n + T - 1) / T + 3) & ~(int64_t)3;
    int64_t next = std::min<int64_t>(n, per);  // first point no helper thread has taken
    try {
        th.reserve((size_t)T - 1);
        for (int t = 1; t < T; t++) {
            const int64_t a = std::min<int64_t>(n, per * t), b = std::min<int64_t>(n, per * (t + 1));
            if (a < b) th.emplace_back(pack_range, dst, s, a, b, src_stride_bytes);
            next = b;
        }
    } catch (...) {
    }
    pack_range(dst, s, 0, std::min<int64_t>(n, per), src_stride_bytes);
    if (next < n) pack_range(dst, s, next, n, src_stride_bytes);  // (only after a failed thread start)
    for (std::thread& x : th) x.join();
    return MLD_OK;
}
