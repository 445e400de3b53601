// Shared host/device declarations of the MI355X DepthEstimator path (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mld {

constexpr int kWave = 64;
// pixel-map key: [tag:7 | (0x7FFFFF - origIdx):23 | ground-plane state:2]; atomicMax keeps the smallest original index
// of the newest tag (indices are unique, so the state bits never decide).  The state is valid when the plane was known
// at projection time (SlotDesc::mask_in_key): what CalculateDepthSegmentationPlane (DepthEstimator.cpp:782-900) asks
// about the point - 0 near the plane, not an inlier; 1 near, inlier; 2 farther than the distance threshold (:810-815);
// 3 too close to the threshold for the projection's single-precision test: the feature kernel decides exactly.
constexpr uint32_t kIdxBits = 24;
constexpr uint32_t kIdxMask = (1u << kIdxBits) - 1u;  // list entries keep the point index in their low 24 bits
constexpr uint32_t kTagShift = 25;                    // map keys: the tag sits above index and flags
constexpr uint32_t kKeyIdxMax = (1u << 23) - 1u;
constexpr int64_t kMaxPoints = (int64_t)kKeyIdxMax;  // points per cloud representable in a key
enum : uint32_t { kPtNear = 0u, kPtInlier = 1u, kPtFar = 2u, kPtUnsure = 3u };
__host__ __device__ inline uint32_t make_key(uint32_t tag, uint32_t idx, uint32_t flags) {
    return (tag << kTagShift) | ((kKeyIdxMax - idx) << 2) | (flags & 3u);
}
__host__ __device__ inline uint32_t key_index(uint32_t key) { return kKeyIdxMax - ((key >> 2) & kKeyIdxMax); }
constexpr uint32_t kMaxTag = 127;
// list entries of the fused kernel: [narrow window:1 | ground-plane state:2 | ... | point index:24]
constexpr uint32_t kEntStateShift = 29, kEntNarrow = 1u << 31;
constexpr int kMapPadCells = 16;  // the thread path reads rows with 16-byte loads that may overrun the last cell

// Per-context constants, passed to every kernel by value (kernarg segment -> SGPRs / scalar loads).
struct Calib {
    double T[12];     // lidar -> camera, row-major 3x4
    double Tinv[12];  // camera -> lidar
    double Kinv[9];   // inverse intrinsics, row-major
    double f, cu, cv;
    float Tf[12];           // single-precision copies for the conservative pre-cull of k_project_scatter
    float Tfmax[3];  // max |Tf[r][0..2]| per row, rounded up (bounds of the f32 pre-cull)
    float ff, cuf, cvf;
    float far_elin;  // |T^-1 (T p + t) + t' - p| <= far_elin * (|x|+|y|+|z|) + far_econst in the f64 round trip of
                     // DepthEstimator.cpp:810 (host-computed residual + rounding bound; k_project_scatter's far test)
    // margins of the pre-cull as linear functions of m1 = |x|+|y|+|z| (rounded up): z test pcm[0] m1 + pcm[1], u tests
    // pcm[2] m1 + pcm[3], v tests pcm[4] m1 + pcm[5]
    float pcm[6];
    float far_econst;
    float roadDistThrF;  // (float)roadDistThr
    float padf2_;
    double halfX1, halfY1;  // main search window half sizes  (scale 1.0, 1.0)
    double halfX2, halfY2;  // road search window half sizes  (scale 2.0, 1.5)
    double binW;
    double thrG_min, thrG_max;
    double thrL_val;
    double planarThr, orthThr;
    double roadDistThr, zxMinRel;
    double pcaAbsMin, pcaRelMax, pcaRelMin;
    int W, H;
    int cap;  // capacity (entries) of the per-wave neighbour list in LDS
    int minCount;
    unsigned countMin;
    int useHist;
    int thrG_en, thrG_mode;
    int thrL_en, thrL_mode, thrL_type;
    int useTriMax, checkPlanar, cutBehind;
    int useRoad;   // do_use_ransac_plane
    int roadMode;  // 0 = M-estimator, 1 = max spanning triangle
    int usePCA;
    int bmStride;    // words per 32-pixel column of the occupancy bitmap (word = (x >> 5) * bmStride + y): H + slack
    int k1max;       // lane-per-feature kernels: longest scanned-window (road-window) list a lane takes
    int kMain;       // ... longest narrow-window list
    int kTotal;      // ... entries of LDS per lane, shared by the two lists (the narrow lists start where the longest wide
                     //     list of the wavefront ends): a feature stays on the lane path when k2 <= k1max, k1 <= kMain and
                     //     (longest k2 of its wavefront) + k1 <= kTotal
    int sortClasses; // k_classify: 4 = live queue ordered by neighbour-count class, then image row; 1 = by row alone
    int xcdAware;    // 1: blocks of one slot are congruent mod 8 (same XCD under round-robin dispatch)
    int threadPath;  // 1: thread-per-feature fast path with wave-cooperative overflow; 0: wave path only
};

// A ground plane estimated on the device (batched RANSAC): what set_plane_coeffs() computes on the host, left in device
// memory so that no host round trip separates the estimation from the kernels that use it.
struct PlaneDev {
    double prior_n[3];
    double prior_off;
    float coeffs[4];
    int has_plane;   // 0: the estimation failed (GroundPlane::ExceptionPclInvalid): the road fallback is off for the frame
    int status;      // 0 ok, 1 too few points / no model
    int n_inliers, iterations, best_draw, best_count, S, pad_;
    float far_mg0, far_mg1;  // margins of the projection's single-precision far test for this plane (far_margins)
};

// Margin of k_project_scatter's single-precision "far from the ground plane" test as a linear function of
// m1 = |x|+|y|+|z|: mg0 * m1 + mg1.  The f64 round trip of DepthEstimator.cpp:810 returns the raw float coordinates up
// to e = far_elin * m1 + far_econst, so the distance evaluated on the RAW floats differs from the reference's by at
// most cs (2^-24 m1 + 2 e) + 8 * 2^-24 (cs m1 + |d|), cs = |a|+|b|+|c|; the margin is several times that (rounded up),
// plus the rounding of the threshold.  Computed once per plane (host, or k_rs_batch), not per wavefront.
__host__ __device__ inline void far_margins(const float co[4], float far_elin, float far_econst, float thr, float& mg0,
                                            float& mg1) {
    const float cs = (fabsf(co[0]) + fabsf(co[1]) + fabsf(co[2])) * 1.001f;
    mg0 = (2e-6f * cs + 4.f * cs * far_elin) * 1.001f;
    mg1 = (2e-6f * fabsf(co[3]) + 4.f * cs * far_econst + 2e-7f * fabsf(thr)) * 1.001f + 1e-30f;
}

// Per-frame-slot descriptor (device-resident array, or passed by value for single-slot calls).
struct SlotDesc {
    const unsigned char* cloud;   // float32 records, `stride` bytes apart (x,y,z first)
    uint32_t* map;                // W*H keys
    uint32_t* bitmap;             // occupancy bits of the map, column-of-words layout (cleared per cloud)
    const double* uv;             // 2 x F column-major
    double* depth;                // F
    int32_t* type;                // F or nullptr
    const uint32_t* inlier_mask;  // bit i = original point i is a ground-plane inlier; nullptr = no plane
    int32_t* ovf_queue;           // (feature index, code) pairs queued for k_feature_wave (long lists)
    int32_t* ovf_count;
    int32_t* live_queue;          // indices of the features k_classify found live, sorted by image row
    int32_t* live_count;          // their number
    double* corners;              // debug mode only: 9 x F triangle corners (NaN = none), written by k_feature_wave
    const long long* F_dev;       // optional device-side feature count (<= F); used when the count is produced on the GPU
    const PlaneDev* plane_dev;    // optional: the slot's plane lives in device memory (overrides the plane fields below)
    long long n;                  // points
    long long F;                  // features
    double prior_n[3];            // M-estimator prior (normalised lidar-frame normal, DepthEstimator.cpp:286-292)
    double prior_off;
    float coeffs[4];              // plane a,b,c,d (lidar frame)
    int stride;
    uint32_t tag;  // current map tag, 1..kMaxTag (127)
    int has_plane;
    int mask_in_key;  // 1: the inlier / far flags of every map key are valid (the plane was known when the cloud was projected)
    float far_mg0, far_mg1;  // margins of the projection's single-precision far test for `coeffs` (far_margins)
    long long F_dev_off;     // this descriptor's first feature among those *F_dev counts (feature groups of a slot; else 0)
};

// One sequence of the batched tracklet layer (mld_tracklets_depths_device): the arrays of
// TrackletDepthModule::process (tracklet_depth_module.cpp:261-396) for that sequence's current frame.
struct TrkSeq {
    const float *u_new, *v_new, *u_old, *v_old;  // newest / previous feature of every track
    const uint8_t* is_new;
    long long n;                    // tracks
    double *uv_cur, *uv_last;       // marshalled features (2 x n column-major; uv_last compacted to the new tracks)
    int32_t* rank;                  // rank among the new tracks, or -1
    long long* n_new;               // number of new tracks (device)
    const double *depth_cur, *depth_last;
    const int32_t *type_cur, *type_last;
    float *d_cur_out, *d_last_out;  // FeaturePoint.d of the newest / previous features
    int32_t *type_cur_out, *type_last_out;
    int have_last, pad_;
};

}  // namespace mld
