/*
 * mld.h — C-ABI of the MI355X-native DepthEstimator hot path.
 *
 * This is the drop-in boundary for monolidar_fusion's `Mono_Lidar::DepthEstimator`
 * (reference: monolidar_fusion/include/monolidar_fusion/DepthEstimator.h:39-359).
 * The reference has no FFI; its boundary is a C++ class.  Every entry point below names
 * the reference member it replaces.  The C++ shim class that keeps the reference's
 * method names on top of this ABI lives in
 * mono_lidar_depth_amd/host/monolidar_fusion/DepthEstimator.h; INTEGRATION.md shows how
 * a maintainer links it into tracklets_depth.
 *
 * Conventions
 *   - plain pointers and sizes only; caller-owned buffers; no callee resize.
 *   - return value: MLD_OK (0) or a negative mld_status; mld_last_error() gives the text
 *     (the reference throws `const char*` / std::string / std::runtime_error instead:
 *     DepthEstimator.cpp:38,57,94,227,270,439,608,797).
 *   - per-feature failures are never errors: (resultType, depth = -1), as in the reference
 *     (eDepthResultType.h:8-30).
 *   - one context per (GPU, stream).  A context owns `max_frames` independent frame slots so
 *     that many frames can be processed by ONE kernel launch set (frames of a sequence, or a
 *     micro-batch of sequences).  slot 0 alone gives the reference's one-frame-at-a-time use.
 *   - `*_device` entry points take device pointers (zero-copy: the cloud is read in place),
 *     are asynchronous on the context's stream, and never touch the host buffers.
 *   - threading: like a reference instance (mutable `_points`, DepthEstimator.h:300-334), a context serves one call
 *     at a time; different contexts are independent and may be driven from different host threads or processes
 *     (one per GPU).  mld_last_error / mld_create_error return thread-unsafe static text.
 *   - ordering with the caller's own GPU work: record/wait events on mld_get_stream() (see bench.py streaming_leg).
 */
#ifndef MLD_H_
#define MLD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLD_ABI_VERSION 8

typedef enum mld_status {
    MLD_OK = 0,
    MLD_ERR_INVALID_ARG = -1,        /* null pointer, bad slot, bad stride ...                         */
    MLD_ERR_NOT_INITIALIZED = -2,    /* "call of 'CalculateDepth' without 'SetInputCloud'" (:439)      */
    MLD_ERR_UNSUPPORTED_MODE = -3,   /* neighbor_search_mode != 0 (:57), region growing (:608), ...    */
    MLD_ERR_NO_ROAD_ESTIMATOR = -4,  /* "No road depth estimator selected." (:94)                      */
    MLD_ERR_NO_GROUND_PLANE = -5,    /* do_use_ransac_plane set but no plane supplied for the slot     */
    MLD_ERR_CLOUD_TOO_SMALL = -6,    /* GroundPlane::ExceptionPclInvalid (RansacPlane.cpp:44-50)       */
    MLD_ERR_HIP = -7,                /* a HIP runtime call failed; text in mld_last_error              */
    MLD_ERR_CAPACITY = -8            /* n / F / window larger than the context was created for (clouds: 8 388 607 points) */
} mld_status;

/* DepthResultType — monolidar_fusion/include/monolidar_fusion/eDepthResultType.h:8-30 (same values). */
typedef enum mld_result_type {
    MLD_Unspecified = 0,
    MLD_Success = 1,
    MLD_RadiusSearchInsufficientPoints = 2,
    MLD_HistogramNoLocalMax = 3,
    MLD_TresholdDepthGlobalGreaterMax = 4,
    MLD_TresholdDepthGlobalSmallerMin = 5,
    MLD_TresholdDepthLocalGreaterMax = 6,
    MLD_TresholdDepthLocalSmallerMin = 7,
    MLD_TriangleNotPlanar = 8,
    MLD_TriangleNotPlanarInsufficientPoints = 9,
    MLD_CornerBehindCamera = 10,
    MLD_PlaneViewrayNotOrthogonal = 11,
    MLD_PcaIsPoint = 12,
    MLD_PcaIsLine = 13,
    MLD_PcaIsCubic = 14,
    MLD_InsufficientRoadPoints = 15,
    MLD_SuccessRoad = 16,
    MLD_RegionGrowingNearestSeedNotAvailable = 17,
    MLD_RegionGrowingSeedsOutOfRange = 18,
    MLD_RegionGrowingInsufficientPoints = 19,
    MLD_SuccessRegionGrowing = 20,
    MLD_RESULT_TYPE_COUNT = 21
} mld_result_type;

/* CameraPinhole(width, height, f, cu, cv) — camera_pinhole.h:29-34. */
typedef struct mld_camera {
    double focal_length;
    double principal_point_x;
    double principal_point_y;
    int32_t width;
    int32_t height;
} mld_camera;

/*
 * The fields of DepthEstimatorParameters (DepthEstimatorParameters.h:7-173) that steer the hot
 * path and the per-frame ground-plane estimator, same names (including the reference's spelling).
 * Fields the path never reads (kd-tree, region growing, debug knobs) are not mirrored.
 * Layout: all doubles first, then int32s — no implicit padding except the tail.
 */
typedef struct mld_params {
    double histogram_segmentation_bin_witdh;     /* :43  */
    double treshold_depth_local_value;           /* :92  */
    double pca_treshold_3_abs_min;               /* :102 */
    double pca_treshold_3_2_rel_max;             /* :103 */
    double pca_treshold_2_1_rel_min;             /* :104 */
    double ransac_plane_point_distance_treshold; /* :122 */
    double ransac_plane_distance_treshold;       /* :114  RANSAC inlier distance (ground-plane estimation) */
    double ransac_plane_min_z;                   /* :115 */
    double ransac_plane_max_z;                   /* :116 */
    double ransac_plane_refinement_treshold;     /* :119 */
    double ransac_plane_probability;             /* :124 */
    double plane_estimator_z_x_min_relation;     /* :141 */
    double triangleplanar_crossnorm_treshold;    /* :154 */
    double viewray_plane_orthoganality_treshold; /* :155 */
    double ransac_plane_treshold_camx;           /* :121  |x_cam| bound of the ground-plane debug cloud */
    int32_t neighbor_search_mode;                /* :20  must be 0 */
    int32_t pixelarea_search_witdh;              /* :21  */
    int32_t pixelarea_search_height;             /* :22  */
    int32_t radiusSearch_count_min;              /* :35  */
    int32_t do_use_histogram_segmentation;       /* :42  */
    int32_t histogram_segmentation_min_pointcount; /* :44 */
    int32_t do_use_depth_segmentation;           /* :48  must be 0 (reference throws, DepthEstimator.cpp:608) */
    int32_t treshold_depth_enabled;              /* :78  */
    int32_t treshold_depth_mode;                 /* :79  0 Dispose, 1 Adjust */
    int32_t treshold_depth_max;                  /* :80  (int in the reference) */
    int32_t treshold_depth_min;                  /* :81  */
    int32_t treshold_depth_local_enabled;        /* :89  */
    int32_t treshold_depth_local_mode;           /* :90  */
    int32_t treshold_depth_local_valuetype;      /* :91  0 absolute, 1 relative */
    int32_t do_use_PCA;                          /* :100 */
    int32_t do_use_ransac_plane;                 /* :113 */
    int32_t plane_estimator_use_triangle_maximation; /* :140 */
    int32_t plane_estimator_use_leastsquares;    /* :142 (Ceres variant: unsupported, UB in the reference) */
    int32_t plane_estimator_use_mestimator;      /* :143 */
    int32_t do_use_cut_behind_camera;            /* :151 */
    int32_t do_use_triangle_size_maximation;     /* :152 */
    int32_t do_check_triangleplanar_condition;   /* :153 */
    int32_t set_all_depths_to_zero;              /* :156 */
    int32_t ransac_plane_max_iterations;         /* :117 */
    int32_t ransac_plane_use_refinement;         /* :118 */
    int32_t ransac_plane_use_camx_treshold;      /* :120 */
} mld_params;

/* Header defaults of DepthEstimatorParameters.h (note viewray_plane_orthoganality_treshold{01} == 1.0). */
void mld_params_default(mld_params* p);
/* "C0": monolidar_fusion/parameters.yaml with do_use_depth_segmentation: 0. */
void mld_params_c0(mld_params* p);
/* DepthEstimatorParameters::fromFile (DepthEstimatorParameters.cpp:16-114): `key: value` YAML subset, cv::FileStorage
 * semantics: an absent key reads as 0, `(int)node` rounds to nearest (saturating here).  On failure `err` holds the
 * reason; on success it holds a note listing the mirrored keys the file lacks ("absent (read as 0): a, b", or ""). */
int mld_params_from_file(mld_params* p, const char* path, char* err, int err_len);

typedef struct mld_ctx mld_ctx;

int mld_abi_version(void);

/*
 * InitConfig + Initialize (DepthEstimator.cpp:35-154).
 *   T_cam_lidar: row-major 3x4 [R|t] of the lidar->camera Affine3d.
 *   max_frames:  number of frame slots (>=1).  max_points / max_features: per-slot capacities for
 *   the host-pointer entry points (device-pointer entry points borrow the caller's buffers).
 * Returns NULL on failure; *status_out (optional) receives the reason, and
 * mld_create_error() the text.
 */
mld_ctx* mld_create(const mld_params* params, const mld_camera* camera, const double T_cam_lidar[12],
                    int device, int max_frames, int64_t max_points, int64_t max_features, int* status_out);
const char* mld_create_error(void);
void mld_destroy(mld_ctx* ctx);
const char* mld_last_error(const mld_ctx* ctx);

/* The hipStream_t the context launches on (as void*), for callers that order their own work. */
void* mld_get_stream(mld_ctx* ctx);

/*
 * Do the streams of two contexts execute side by side?  1 = yes; 0 = they share a hardware queue of the HIP runtime
 * (their kernels run strictly in turn: mld_order_after[_classify] / mld_set_shared_gpu then buy nothing) or are the same
 * context; < 0 = error.  The runtime multiplexes a process's streams on a few hardware queues (four per priority unless
 * GPU_MAX_HW_QUEUES says otherwise), so mld_create probes a new context's stream against every live context of its device
 * and replaces it until it has a queue of its own (up to six tries); this call repeats the probe on demand (both streams
 * are synchronised, two one-wavefront kernels run: ~50 us; 3 ms when the answer is 0).  No reference counterpart.
 */
int mld_contexts_concurrent(mld_ctx* a, mld_ctx* b);
int mld_synchronize(mld_ctx* ctx);

/*
 * Two (or more) contexts on one GPU, used alternately ("double buffering"; DESIGN.md §4 "Batches").
 * The projection kernel streams the clouds from HBM; the feature kernels wait for scattered fetches and move a tenth of
 * the bytes.  Run beside each other they finish sooner than one after the other (HBM busy 0.8 of the step instead of
 * 0.6) -- the reference has no counterpart: its stage A is serial and its feature loop is the only
 * parallel part (DepthEstimator.cpp:156-217 vs :455).  Schedule per batch, contexts taken round-robin:
 *     mld_set_clouds_*_device(ctx_k, ...);      projection of batch i on context k
 *     mld_order_after(ctx_next, ctx_k);          the NEXT context's projection starts when this one is done ...
 *       (or mld_order_after_classify: ... when this context's classification kernel is done)
 *     mld_calculate_depths_device(ctx_k, ...);   ... i.e. beside these feature kernels
 *
 * mld_order_after: everything submitted to `ctx` from now on starts after everything submitted to `other` so far has
 *   finished (one event; no host synchronisation).  Both contexts must live on the same device.
 * mld_set_shared_gpu(ctx, 1): the lane-per-feature kernel of `ctx` keeps to ten wavefronts per CU instead of sixteen (it
 *   requests more LDS per block), which leaves registers for the other context's projection wavefronts on every CU.  A
 *   context that has the GPU to itself is ~20 % slower in this mode; the alternating pair is ~20 % faster (bench.py
 *   default: 0.68-0.72 instead of 0.89 ms per 1024 frames of config 2).  `shared`: bit 0 = on; bits 8..15 = wavefronts of
 *   the lane-per-feature kernel per CU in that mode (0 = the default, 10): fewer leave more of every CU to the projection.
 */
int mld_order_after(mld_ctx* ctx, mld_ctx* other);
/*
 * mld_order_after_classify: the same hand-over, but it is `ctx`'s next PROJECTION KERNEL (and what follows it) that waits -
 *   the bitmap fill and descriptor upload queued ahead of that kernel do not -, and it is released behind the
 *   classification kernel of `other`'s NEXT mld_calculate_depth(s)_device call instead of at once: the classification
 *   (one 1024-thread block and 63 KB of LDS per frame, 40 us per 1024 frames) then has the GPU to itself instead of competing with 131 072 projection blocks for wave
 *   slots, and the projection of `ctx` still runs beside the long feature kernels (measured: k_classify 70-90 -> 44 us,
 *   the step 0.5-1.5 % shorter and steadier; LAB.md 4.17).  The release itself is a one-wavefront kernel (k_gate) queued
 *   in front of the projection that polls a counter of the classification blocks that have been placed on a CU - once the
 *   last one runs, nothing of that kernel still waits for wave slots or LDS (LAB.md 5.29) - 2-3 us from there to the
 *   projection's start instead of the ~15 us of a cross-stream event (LAB.md 4.28); its polling is bounded (it gives up
 *   after 100 000 polls, a fraction of a second), because nothing depends on it: purely a scheduling hint, the contexts share no data.  Nothing waits if
 *   `other` never issues that call; a pending hand-over ends with either context.
 */
int mld_order_after_classify(mld_ctx* ctx, mld_ctx* other);
/*
 * mld_pair_contexts(a, b): the batched setInputCloud entry points (mld_set_clouds_*_device) of BOTH contexts run their
 *   projection on one stream created here (owned by `a`), back to back in call order, while each context's feature
 *   kernels stay on its own stream (two events per batch join them).  With the schedule above this replaces
 *   mld_order_after: the projection of batch i+1 follows that of batch i without a cross-stream hand-over (~50 us per
 *   1024-frame batch) and still runs beside the feature kernels of batch i.  A projection waits for everything queued on
 *   its own context before (the kernels still reading the slots' maps).  Destroy `b` before `a`.
 */
int mld_pair_contexts(mld_ctx* a, mld_ctx* b);
int mld_set_shared_gpu(mld_ctx* ctx, int shared);
/*
 * Capacities (entries per feature) of the lane-per-feature kernel's two neighbour lists: the scanned window (the road
 * window, 2.0 x 1.5 the search window, when the fallback is on) and the narrow search window.  Features whose lists are
 * longer are handled by the wave-cooperative kernel (same results, ~10x the cost per feature).  Default 32 / 24: right
 * for 64-beam clouds (config 2: 11 / 2.3 neighbours on average).  Dense clouds (128 beams x 4096: 16 / 6 on average, 48
 * at most in the road window) want 48 / 24: BASELINE config 5 at batch size runs 1.8x faster with it.  LDS per wavefront
 * = (wide + narrow) * 256 bytes unless mld_set_list_budget lowers it.  8 <= narrow <= wide <= 64.  Capacities beyond the default also select the kernel's
 * DENSE instantiations: the corner search over the segmented points and the histogram's depths in registers - up to 24 of
 * each at two wavefronts per SIMD when the context has the GPU to itself, up to 16 inside 168 registers when
 * mld_set_shared_gpu is on, so that another context's projection finds room beside it (config 5 at 256 sequences per
 * step: 1.12 G associations/s in round 4 -> 1.85 G with one context, 1.98 G with two in turn).
 */
int mld_set_list_capacity(mld_ctx* ctx, int wide_entries, int narrow_entries);

/*
 * LDS entries per feature that the two lists SHARE (the narrow lists of a wavefront's 64 features are stored behind the
 * longest of their wide lists): a feature stays on the lane-per-feature kernel when wide <= wide capacity, narrow <= narrow
 * capacity and (longest wide list of its wavefront) + narrow <= total.
 * LDS per wavefront = total * 256 bytes, which sets how many wavefronts of the kernel a CU holds (160 KB / that, at most
 * four per SIMD).  A new context: capacities 32 / 24, budget 40 (10 KB, 16 wavefronts per CU; on 64-beam and 16-beam
 * clouds no list pair comes near it: wide + narrow <= 27 on BASELINE configs 2 and 3).  mld_set_list_capacity resets the
 * budget to wide + narrow (both lists at their longest: the dense 128-beam clouds of config 5 reach 66 of 48 + 24).
 * wide capacity <= total_entries <= wide + narrow capacity; 0 = wide + narrow.
 */
int mld_set_list_budget(mld_ctx* ctx, int total_entries);

/*
 * setInputCloud (DepthEstimator.cpp:220-312): projection + pixel->point map for one slot.
 *   pts: x,y,z[,..] float32 records, stride_bytes 16 (packed xyzi) or 32 (pcl::PointXYZI layout,
 *   DepthEstimator.cpp:169 reads rows 0-2 of the 8-float map).
 */
int mld_set_cloud(mld_ctx* ctx, int slot, const void* pts_host, int64_t n, int stride_bytes);
int mld_set_cloud_device(mld_ctx* ctx, int slot, const void* pts_dev, int64_t n, int stride_bytes);
/* All slots [0, n_slots) in ONE launch; pts_dev[i] / n[i] are host arrays of device pointers / counts. */
int mld_set_clouds_device(mld_ctx* ctx, int n_slots, const void* const* pts_dev, const int64_t* n,
                          int stride_bytes);
/*
 * Host-side staging for callers that stream frames through (pinned) upload buffers: copies n records of
 * src_stride_bytes (32: pcl::PointXYZI as the reference's caller holds them, DepthEstimator.h:62-63; or 16) into packed
 * 16-byte records {x, y, z, intensity} at dst16, with up to n_threads host threads (one per ~64 K points).  The copy a
 * driver makes anyway then halves what crosses PCIe, the bound of the streamed path.  No GPU call; no context.
 */
int mld_pack_points_host(void* dst16, const void* src, int64_t n, int src_stride_bytes, int n_threads);

/*
 * setInputCloud(cloud, groundPlane) with an already segmented plane (DepthEstimator.cpp:220-312, :281: nothing to
 * estimate) for slots [0, n_slots) in ONE launch: coeffs is n_slots x 4 (host), mask_dev a host array of device
 * bitmask pointers (bit i of word i/32 = point i is an inlier, the device form of `_pointIsInPlane`,
 * RansacPlane.h:116-122).  Because the plane is known when the cloud is projected, every pixel-map entry carries
 * its point's inlier flag and the road fallback needs no mask lookups.  Equivalent to mld_set_clouds_device +
 * mld_set_ground_planes_mask_device, which remain for planes that arrive after the cloud.
 */
int mld_set_clouds_planes_device(mld_ctx* ctx, int n_slots, const void* const* pts_dev, const int64_t* n,
                                 int stride_bytes, const float* coeffs, const uint32_t* const* mask_dev);

/*
 * setInputCloud(cloud, groundPlane) with a plane that is NOT segmented yet — the reference's default call — for slots
 * [0, n_slots) in one asynchronous launch set: RansacPlane::CalculateInliersPlane (RansacPlane.cpp:41-140) runs for
 * every slot on the GPU (one block per slot), the planes stay in device memory where the projection (inlier flags in
 * the map keys) and the feature kernels read them; nothing returns to the host.  seeds: one per slot (the reference's
 * pcl::RandomSample is time-seeded).  A slot whose estimation fails (GroundPlane::ExceptionPclInvalid, which the
 * caller of the reference catches, tracklet_depth_module.cpp:321,338) runs without the road fallback.
 * mld_get_estimated_planes returns coefficients / inlier counts / status (0 ok, 1 failed) and synchronises; the inlier
 * sets are available through mld_get_ground_plane_inliers.
 */
int mld_set_clouds_estimate_planes_device(mld_ctx* ctx, int n_slots, const void* const* pts_dev, const int64_t* n,
                                          int stride_bytes, const uint32_t* seeds);
int mld_get_estimated_planes(mld_ctx* ctx, int n_slots, float* coeffs_out, int64_t* n_inliers_out, int32_t* status_out);

/*
 * The GroundPlane object handed to setInputCloud/CalculateDepth (RansacPlane.h:38-123):
 * coefficients a,b,c,d in the LIDAR frame + the inlier index set keyed by ORIGINAL cloud index
 * (`_pointIsInPlane`, RansacPlane.h:116-122).  coeffs == NULL: "ransacPlane == nullptr"
 * (road fallback skipped, DepthEstimator.cpp:580).  Must be called after the slot's cloud is set.
 */
int mld_set_ground_plane(mld_ctx* ctx, int slot, const float coeffs[4], const int32_t* inlier_idx_host,
                         int64_t n_inliers);
int mld_set_ground_plane_device(mld_ctx* ctx, int slot, const float coeffs[4], const int32_t* inlier_idx_dev,
                                int64_t n_inliers);
/*
 * RansacPlane::CalculateInliersPlane on the GPU (monolidar_fusion/src/RansacPlane.cpp:41-140) — what setInputCloud
 * runs when the GroundPlane handed in is not segmented yet (DepthEstimator.cpp:275-283).  Estimates the plane of
 * the slot's cloud (parameters ransac_plane_*), installs it as the slot's ground plane and returns the
 * coefficients and the inlier count.  `seed` makes the random draws reproducible (the reference is time-seeded).
 * Synchronises.  MLD_ERR_CLOUD_TOO_SMALL: fewer than 3 usable points / no model (ExceptionPclInvalid).
 */
int mld_estimate_ground_plane(mld_ctx* ctx, int slot, uint32_t seed, float coeffs_out[4], int64_t* n_inliers_out);
/*
 * SemanticPlane::CalculateInliersPlane on the GPU (monolidar_fusion/src/RansacPlane.cpp:195-274) — the ground plane
 * tracklets_depth uses when a semantic label image accompanies the frame (tracklet_depth_module.cpp:270-284):
 * points whose projection falls on a pixel with a ground label (:198-221), least-squares plane through them
 * (:238-246), all cloud points within `inlier_threshold` of it (:252), plane refitted to those (:253); installs the
 * result as the slot's ground plane (coefficients of the refit, inliers = the selected points, :259-268).
 * SemanticPlane::Camera is the context's own calibration (f, cu, cv, T_cam_lidar), as the caller builds it.
 * label_image: rows x cols uint8, row_stride_bytes apart.  Fewer than 3 labelled points: MLD_ERR_CLOUD_TOO_SMALL
 * (ExceptionPclInvalid, :224-227).  Pixels x == cols / y == rows, which the reference reads out of bounds, carry no
 * label here.  PCL's sequential float sums are replaced by 256 interleaved partial sums (parity unpinned, as for
 * mld_estimate_ground_plane).
 */
int mld_estimate_semantic_plane(mld_ctx* ctx, int slot, const uint8_t* label_image_host, int rows, int cols,
                                int row_stride_bytes, const int32_t* ground_labels, int n_labels, double inlier_threshold,
                                float coeffs_out[4], int64_t* n_inliers_out);
int mld_estimate_semantic_plane_device(mld_ctx* ctx, int slot, const uint8_t* label_image_dev, int rows, int cols,
                                       int row_stride_bytes, const int32_t* ground_labels, int n_labels,
                                       double inlier_threshold, float coeffs_out[4], int64_t* n_inliers_out);
/* The slot's current inlier set as ascending original indices (GroundPlane::getInlinersIndex). */
int mld_get_ground_plane_inliers(mld_ctx* ctx, int slot, int32_t* index_out, int64_t capacity, int64_t* n_out);
/* Same, with the inlier set already as a device bitmask (bit i of word i/32 = point i is an inlier). */
int mld_set_ground_plane_mask_device(mld_ctx* ctx, int slot, const float coeffs[4], const uint32_t* mask_dev);
/* Slots [0, n_slots): coeffs is n_slots x 4 (host), mask_dev a host array of device bitmask pointers. */
int mld_set_ground_planes_mask_device(mld_ctx* ctx, int n_slots, const float* coeffs, const uint32_t* const* mask_dev);

/*
 * CalculateDepth(Matrix2Xd, VectorXd&, VectorXi&, GroundPlane::Ptr) (DepthEstimator.cpp:429-488).
 *   uv: 2 x F column-major (u0,v0,u1,v1,...) float64, as Eigen::Matrix2Xd stores it.
 *   depth_out: F float64 (metres, camera-frame z of the intersection, -1 on failure);
 *   type_out: F int32 mld_result_type (may be NULL: the overload without resultType, :422-427).
 * The host-pointer form copies in, runs, copies out and synchronises.
 */
int mld_calculate_depth(mld_ctx* ctx, int slot, const double* uv_host, int64_t F, double* depth_out_host,
                        int32_t* type_out_host);
/* flags of mld_calculate_depth_opts */
#define MLD_CALC_SKIP_ROAD 1u /* this call only: "ransacPlane == nullptr" (DepthEstimator.cpp:580), the slot's plane stays */
int mld_calculate_depth_opts(mld_ctx* ctx, int slot, const double* uv_host, int64_t F, double* depth_out_host,
                             int32_t* type_out_host, uint32_t flags);
int mld_calculate_depth_device(mld_ctx* ctx, int slot, const double* uv_dev, int64_t F, double* depth_out_dev,
                               int32_t* type_out_dev);
/*
 * CalculateDepth(cloud, uv, depths, resultType, groundPlane) (DepthEstimator.cpp:404-420: setInputCloud followed by
 * the per-feature loop) for ONE frame held in host memory, in a single call: mld_set_cloud + mld_set_ground_plane +
 * mld_calculate_depth with the small transfers combined and the plane installed before the projection.  The one-frame
 * latency path of a ROS callback (tracklet_depth_module.cpp:63-117).  coeffs == NULL: no plane (road fallback off).
 * Synchronises.  type_out may be NULL.
 */
int mld_calculate_depth_frame(mld_ctx* ctx, int slot, const void* pts_host, int64_t n, int stride_bytes,
                              const float coeffs[4], const int32_t* inlier_idx_host, int64_t n_inliers,
                              const double* uv_host, int64_t F, double* depth_out_host, int32_t* type_out_host);
/*
 * The same single call for the reference's PRODUCTION use: the GroundPlane handed to CalculateDepth(cloud, uv, depths,
 * types, groundPlane) is NOT segmented yet - TrackletDepthModule::process builds a fresh one for every frame
 * (tracklet_depth_module.cpp:269-284) and setInputCloud estimates it (DepthEstimator.cpp:275-283) before the feature loop
 * runs.  One asynchronous chain on the context's stream, no host round trip before the result:
 *     H2D (cloud, features, label image) -> plane estimation on the slot (RansacPlane::CalculateInliersPlane,
 *     RansacPlane.cpp:41-140, one block; or SemanticPlane::CalculateInliersPlane, :195-274) -> projection WITH that plane
 *     (the points' ground-plane state rides in the pixel-map keys) -> feature kernel -> one D2H (depths, types, plane)
 * The estimated plane stays installed on the slot (later mld_calculate_depth calls use it); its inlier index list is
 * not produced here - mld_get_ground_plane_inliers fetches it on demand (GroundPlane::getInlinersIndex).
 *   plane->kind MLD_PLANE_RANSAC:   `seed` fixes the draws (the reference's pcl::RandomSample is time-seeded)
 *   plane->kind MLD_PLANE_SEMANTIC: label_image (HOST memory, rows x cols uint8, row_stride_bytes apart), ground_labels,
 *                                   inlier_threshold as mld_estimate_semantic_plane
 *   plane_out (optional): coefficients, inlier count, status 0 / 1, RANSAC iterations.
 * Without do_use_ransac_plane the plane request is ignored (as setInputCloud does, :274).  A failed estimation
 * (GroundPlane::ExceptionPclInvalid: fewer than 3 usable points / no model) returns MLD_ERR_CLOUD_TOO_SMALL and leaves
 * depth_out / type_out untouched - the reference throws out of setInputCloud before any depth is computed
 * (tracklet_depth_module.cpp:321,338 catch it).  Synchronises.  type_out may be NULL.
 */
#define MLD_PLANE_RANSAC 0
#define MLD_PLANE_SEMANTIC 1
typedef struct mld_plane_request {
    int32_t kind;
    uint32_t seed;
    const uint8_t* label_image;
    int32_t rows, cols, row_stride_bytes;
    int32_t n_labels;
    const int32_t* ground_labels;
    double inlier_threshold;
} mld_plane_request;
typedef struct mld_plane_result {
    float coeffs[4];
    int64_t n_inliers;
    int32_t status;      /* 0 ok, 1 = ExceptionPclInvalid */
    int32_t iterations;  /* RANSAC iterations PCL's stopping rule counted (0 for the semantic plane) */
} mld_plane_result;
int mld_calculate_depth_frame_estimate(mld_ctx* ctx, int slot, const void* pts_host, int64_t n, int stride_bytes,
                                       const mld_plane_request* plane, const double* uv_host, int64_t F,
                                       double* depth_out_host, int32_t* type_out_host, mld_plane_result* plane_out);
/*
 * Where the time of the LAST one-frame call (mld_calculate_depth_frame / _frame_estimate) went.  The host-clock entries
 * ([4]-[6], [8], [9]) are always filled; the GPU phases ([0]-[3], [7]) only when timing is enabled (mld_timing_enable: the
 * hipEvents that bracket the phases cost the call tens of microseconds, so the un-instrumented call is faster than their
 * sum) and are 0 otherwise.  out_us[10]:
 *   [0] h2d      cloud copy on the context's stream (the small inputs travel beside it on a side stream)
 *   [1] plane    plane estimation kernels (0 for a supplied plane)
 *   [2] kernels  projection + feature kernel(s)
 *   [3] d2h      result copy
 *   [4] api      host time to enqueue everything (entry -> last enqueue returned)
 *   [5] wait     host time blocked in the final synchronise
 *   [6] total    host wall time of the call
 *   [7] gpu      first event -> last event on the stream ([0]+[1]+[2]+[3] plus the gaps between them)
 *   [8] pre      host time before the cloud copy is submitted (validation, staging of the small inputs, side-stream work)
 *   [9] copycall host time inside the cloud's hipMemcpyAsync (a pageable source blocks until it is staged)
 */
int mld_frame_timing(mld_ctx* ctx, double out_us[10]);
/* All slots [0, n_slots) in ONE launch set (host arrays of device pointers / counts). */
int mld_calculate_depths_device(mld_ctx* ctx, int n_slots, const double* const* uv_dev, const int64_t* F,
                                double* const* depth_out_dev, int32_t* const* type_out_dev);

/*
 * Tracklet gather / scatter either side of the path — TrackletDepthModule::process
 * (tracklets_depth/src/tracklet_depth_module.cpp:23-61 ExractNewTrackletFrames, :63-117 CalculateFeatureDepths*,
 * :119-169 SaveFeatureDepths).  For n_tracks tracks:
 *   u_new/v_new : newest feature of every track            (feature_points.at(0), float32 as in the message)
 *   u_old/v_old : previous feature (feature_points.at(1)); read only where is_new[i] != 0
 *   is_new      : the track id is not yet in the tracklet map (a new tracklet contributes TWO features)
 * Features are truncated to integer pixels as the reference's std::pair<int,int> does.  The current frame's
 * features are evaluated on `slot_cur`, the previous features of NEW tracks on `slot_last` — the previous frame's
 * slot, whose projection and pixel map are still resident (no re-projection); slot_last = -1: no previous cloud,
 * those depths are -1 (:93-96).  Both slots must have their cloud (and ground plane) set.
 *   d_cur_out  : n_tracks float32 depths of the newest features            (FeaturePoint.d)
 *   d_last_out : n_tracks float32; written only where is_new[i] != 0
 *   type_*_out : optional int32 result types, same indexing
 *   n_new_host : optional; receives the number of new tracks (synchronises)
 * The `_device` form takes device pointers and is asynchronous on the context's stream.
 */
int mld_tracklets_depth_device(mld_ctx* ctx, int slot_cur, int slot_last, const float* u_new, const float* v_new,
                               const float* u_old, const float* v_old, const uint8_t* is_new, int64_t n_tracks,
                               float* d_cur_out, float* d_last_out, int32_t* type_cur_out, int32_t* type_last_out,
                               int64_t* n_new_host);
int mld_tracklets_depth(mld_ctx* ctx, int slot_cur, int slot_last, const float* u_new, const float* v_new,
                        const float* u_old, const float* v_old, const uint8_t* is_new, int64_t n_tracks,
                        float* d_cur_out, float* d_last_out, int32_t* type_cur_out, int32_t* type_last_out,
                        int64_t* n_new_host);
/*
 * TrackletDepthModule::process (tracklet_depth_module.cpp:261-396) for ONE frame held in host memory as a single call -
 * the ROS callback's whole GPU side: setInputCloud of the new cloud on `slot_cur` with its ground plane (`plane`: not
 * segmented yet, estimated on the GPU inside the call as in mld_calculate_depth_frame_estimate - what the reference's
 * process() does every frame; or plane == NULL and coeffs / inliers: a segmented plane; both NULL: no plane), the
 * feature marshalling, both CalculateDepth calls (the previous frame from its resident `slot_last`, -1: none) and the
 * float32 scatter of mld_tracklets_depth.  One asynchronous chain, one synchronisation; track arrays and results travel in
 * one pinned block each way.  Arrays and outputs as mld_tracklets_depth.  A failed plane estimation
 * (GroundPlane::ExceptionPclInvalid) returns MLD_ERR_CLOUD_TOO_SMALL with d_cur_out = -1 and d_last_out answered from
 * the previous frame, as the reference's two try blocks do (:318-347).
 */
int mld_tracklets_frame(mld_ctx* ctx, int slot_cur, int slot_last, const void* pts_host, int64_t n, int stride_bytes,
                        const mld_plane_request* plane, const float coeffs[4], const int32_t* inlier_idx_host,
                        int64_t n_inliers, const float* u_new, const float* v_new, const float* u_old, const float* v_old,
                        const uint8_t* is_new, int64_t n_tracks, float* d_cur_out, float* d_last_out, int32_t* type_cur_out,
                        int32_t* type_last_out, int64_t* n_new_host, mld_plane_result* plane_out);
/*
 * The tracklet layer at batch size: the current frames of n_seq independent SEQUENCES in one launch set
 * (TrackletDepthModule::process, tracklet_depth_module.cpp:261-396, once per sequence).  The context's frame slots are
 * used as two banks of n_seq slots (create the context with max_frames >= 2 * n_seq): sequence s keeps its current
 * frame in slot bank_cur * n_seq + s and its previous frame, still resident, in slot (1 - bank_cur) * n_seq + s; the
 * caller flips bank_cur every frame.
 *   mld_set_clouds_planes_range_device  setInputCloud(cloud, plane) for the slots [first_slot, first_slot + n_slots)
 *                                       (coeffs / mask_dev: both or neither), i.e. for one bank
 *   mld_tracklets_depths_device         gather -> depth of every track's newest feature on the current frame and of the
 *                                       NEW tracks' previous features on the previous frame (have_last == 0: no previous
 *                                       clouds yet, those depths are -1) -> float32 scatter; arrays of n_seq device
 *                                       pointers / counts, indexing and meaning per sequence as mld_tracklets_depth_device
 * Asynchronous on the context's stream; per-sequence results equal n_seq single-sequence calls.
 */
int mld_set_clouds_planes_range_device(mld_ctx* ctx, int first_slot, int n_slots, const void* const* pts_dev,
                                       const int64_t* n, int stride_bytes, const float* coeffs,
                                       const uint32_t* const* mask_dev);
int mld_tracklets_depths_device(mld_ctx* ctx, int n_seq, int bank_cur, int have_last, const float* const* u_new,
                                const float* const* v_new, const float* const* u_old, const float* const* v_old,
                                const uint8_t* const* is_new, const int64_t* n_tracks, float* const* d_cur_out,
                                float* const* d_last_out, int32_t* const* type_cur_out, int32_t* const* type_last_out);

/*
 * Debug / parity getters (host buffers; each synchronises).
 *   mld_get_visible_count           -> _points_cs_image_visible.cols()         (DepthEstimator.cpp:192)
 *   mld_get_visible_image_points    -> getPointsCloudImageCs, 2 x Nvis col-major (:392-394)
 *   mld_get_point_index             -> PointcloudData::_pointIndex, Nvis int32   (:203)
 *   mld_get_cloud_camera_cs         -> getCloudCameraCs, 3 x N col-major float64 (:314-334)
 *   mld_get_pixel_map               -> NeighborFinderPixel::_img_points_lidar, W*H int32, index x + y*W,
 *                                      value = VISIBLE index or -1               (NeighborFinderPixel.cpp:38-55)
 *   mld_get_point_depth_cam_visible -> getPointDepthCamVisible(index)            (DepthEstimator.h:111-113)
 */
int mld_get_visible_count(mld_ctx* ctx, int slot, int64_t* n_visible);
/*
 * Which kernel answered the features of the slot's last BATCHED CalculateDepth call (mld_calculate_depths_device and the
 * tracklet entry points; no reference counterpart): `lane_path` = features the classification queued for the
 * lane-per-feature kernel, `handed_over` = features the wave-cooperative kernel worked on (lists beyond the capacities /
 * the budget of mld_set_list_capacity / mld_set_list_budget, windows wider than 32 cells, road fits whose error estimate
 * asks for the QR route; a feature handed over by the lane kernel counts in both).  The feedback for choosing capacities:
 * a wave-kernel feature costs ~10x a lane-path one.  Synchronises the context's stream.
 */
int mld_get_path_counts(mld_ctx* ctx, int slot, int64_t* lane_path, int64_t* handed_over);
int mld_get_visible_image_points(mld_ctx* ctx, int slot, double* uv_out, int64_t capacity);
int mld_get_point_index(mld_ctx* ctx, int slot, int32_t* index_out, int64_t capacity);
int mld_get_cloud_camera_cs(mld_ctx* ctx, int slot, double* xyz_out, int64_t capacity);
int mld_get_pixel_map(mld_ctx* ctx, int slot, int32_t* map_out, int64_t capacity);
int mld_get_point_depth_cam_visible(mld_ctx* ctx, int slot, int64_t visible_index, double* depth_out);

/*
 * Debug mode (ActivateDebugMode, DepthEstimator.h:85-87) — the debug vectors behind getCloudTriangleCorners /
 * getCloudRansacPlane (DepthEstimator.cpp:347-349, 396-398).
 *   mld_calculate_depth_debug  -> CalculateDepth plus `_points_triangle_corners` as filled by
 *                                 CalculatePlaneCorners (PlaneEstimationCalcMaxSpanningTriangle.cpp:20-35,
 *                                 called from DepthEstimator.cpp:916): corners_out is 9 x F column-major
 *                                 (corner1 xyz, corner2 xyz, corner3 xyz; camera frame) for every feature whose
 *                                 spanning triangle was found, NaN otherwise.  Feature order instead of the
 *                                 reference's unordered OpenMP push order.  Same depths/types as
 *                                 mld_calculate_depth; slower (every feature takes the wave-cooperative kernel).
 *   mld_get_ground_plane_cloud -> `_points_groundplane` (DepthEstimator.cpp:294-308): camera-frame coordinates of
 *                                 the slot's ground-plane inliers, ascending original index, filtered by
 *                                 ransac_plane_use_camx_treshold / ransac_plane_treshold_camx.  3 x n
 *                                 column-major; xyz_out == NULL only counts.
 */
int mld_calculate_depth_debug(mld_ctx* ctx, int slot, const double* uv_host, int64_t F, double* depth_out_host,
                              int32_t* type_out_host, double* corners_out_host);
int mld_get_ground_plane_cloud(mld_ctx* ctx, int slot, double* xyz_out, int64_t capacity, int64_t* n_out);

/* DepthCalculationStatistics counterpart: histogram of a resultType array (counts[MLD_RESULT_TYPE_COUNT]). */
int mld_result_histogram(const int32_t* types, int64_t F, int64_t counts[MLD_RESULT_TYPE_COUNT]);

/*
 * Measurement hooks for bench.py (HIP events on the context's stream).
 *   mld_kernel_time_ms: average duration in ms of the `which` kernel (0 = k_project_scatter, 1 = k_feature_fused,
 *   3 = k_feature_wave, 5 = k_classify) over the launches since mld_timing_reset, measured with hipEvents recorded
 *   around each launch when timing is enabled.
 */
int mld_timing_enable(mld_ctx* ctx, int enable);
int mld_timing_reset(mld_ctx* ctx);
int mld_kernel_time_ms(mld_ctx* ctx, int which, double* avg_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* MLD_H_ */
