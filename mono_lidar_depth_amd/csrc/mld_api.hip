// C-ABI (include/mld.h) of the MI355X DepthEstimator path: context, frame slots, launches.
//
// Host counterpart of Mono_Lidar::DepthEstimator (monolidar_fusion/src/DepthEstimator.cpp): Initialize /
// InitConfig -> mld_create, setInputCloud -> mld_set_cloud*, CalculateDepth -> mld_calculate_depth*.
// There is no CPU fallback: every entry point that computes goes through the HIP kernels.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <ctime>
#include <functional>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mld.h"
#include "mld_device.h"
#include "mld_kernels.hip"
#include "mld_ransac.hip"

using namespace mld;

namespace {

thread_local std::string g_create_error;

struct Slot {
    SlotDesc d{};
    bool cloud_set = false;
    bool plane_decided = false;
    // buffers owned by the context (host-pointer entry points)
    unsigned char* cloud_buf = nullptr;
    size_t cloud_cap = 0;
    double* uv_buf = nullptr;
    double* depth_buf = nullptr;
    int32_t* type_buf = nullptr;
    size_t feat_cap = 0;
    int32_t* inl_buf = nullptr;
    size_t inl_cap = 0;
    uint32_t* mask_buf = nullptr;
    size_t mask_words = 0;
    size_t road_cap = 0;  // capacity (features) of d.live_queue / d.ovf_queue
    bool queues_in_slab = false;  // they are carved out of mld_ctx::queue_slab
    // lazy PointcloudData for the debug getters
    bool full_valid = false;
    size_t dbg_cap = 0;
    double* cam = nullptr;
    double* img = nullptr;
    int32_t* vis = nullptr;
    int32_t* rank = nullptr;
    int32_t* pidx = nullptr;
    double* img_vis = nullptr;
    int32_t* block_sums = nullptr;
    int32_t* d_total = nullptr;
    int64_t nvis = 0;
};

struct TimedLaunch {
    hipEvent_t e0, e1;
    int which;
};

}  // namespace

// Helper thread of the one-frame calls (mld_calculate_depth_frame*): while the calling thread sits in the cloud's
// hipMemcpyAsync - a pageable source blocks it for the whole 2.1 MB - the helper stages the small inputs (features,
// inlier list, label image) and queues their DMA, the mask build and the bitmap clear on the side stream.  Done by the
// caller itself that work (25-30 us of host time) used to delay the START of the cloud copy.  One job at a time; the
// thread sleeps on a condition variable between frames and is created by the first one-frame call of a context.
struct FrameHelper {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int(std::string&)> job;
    bool pending = false, quit = false;
    int rc = 0;
    std::string err;
    void run() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return pending || quit; });
            if (quit) return;
            std::function<int(std::string&)> j = std::move(job);
            lk.unlock();
            std::string e;
            const int r = j(e);
            lk.lock();
            rc = r;
            err = std::move(e);
            pending = false;
            cv.notify_all();
        }
    }
    void submit(std::function<int(std::string&)> j) {
        {
            std::lock_guard<std::mutex> lk(m);
            job = std::move(j);
            pending = true;
        }
        cv.notify_all();
    }
    int wait(std::string& e) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return !pending; });
        e = err;
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(m);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

// The projection stream two paired contexts share (mld_pair_contexts).  Either context may be destroyed first.
struct ProjShare {
    hipStream_t stream = nullptr;
    int refs = 0;
};

struct mld_ctx {
    mld_params P{};
    mld_camera cam{};
    Calib calib{};
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = true;
    std::vector<Slot> slots;
    SlotDesc* d_slots = nullptr;
    unsigned char* dummy = nullptr;  // 256 zero bytes: the cloud of a slot without points (list loops pad with point 0)
    int32_t* queue_counts = nullptr;  // per-slot queue lengths (live features, then long-list overflow), contiguous,
                                     // placed in front of the bitmaps so that one fill clears both
    // One allocation per kind for all slots instead of one per slot: a 1024-slot context would otherwise hold 4096
    // small mappings, and the gathers of the feature kernels would walk as many page-table fragments.
    uint32_t* map_slab = nullptr;    // pixel maps of all slots
    size_t map_stride = 0;           // words per slot in map_slab
    int32_t* queue_slab = nullptr;   // per slot: overflow queue (2F), live queue (F)
    uint32_t* bitmaps = nullptr;  // occupancy bitmaps of all slots, contiguous
    size_t bitmap_words = 0;      // per slot
    std::vector<SlotDesc> h_descs;      // what the device copy of the slot descriptors holds (upload_descs)
    std::vector<SlotDesc> desc_stage;   // descriptors about to be uploaded; committed to h_descs once the upload is queued
    size_t lds_bytes = 0;     // k_feature_wave: xyz list of one window
    size_t lds_fused = 0;     // k_feature_fused: wide list + narrow list per wave
    size_t lds_classify = 0;  // k_classify: bucket counters + the slot's bitmap
    bool classify_staged = true;  // the bitmap fits the LDS budget of k_classify
    int bm_ncol = 0, bm_ncolp = 0;  // bitmap word columns (incl. the slack column) / padded LDS row length
    bool force_thread_path = false;  // test build only: single-slot calls use the lane-per-feature kernel too
    size_t proj_lds = 0;             // test build only (MLD_PROJ_LDS): unused dynamic LDS per projection block, caps its occupancy
    std::string err;
    // ground-plane estimation scratch (device)
    int32_t* rs_flags = nullptr;
    int32_t* rs_cand = nullptr;
    int32_t* rs_block = nullptr;
    size_t rs_cap = 0;
    int32_t* rs_M = nullptr;
    int32_t* rs_S = nullptr;
    int32_t* rs_sample = nullptr;
    float* rs_sp = nullptr;
    int32_t* rs_counts = nullptr;
    size_t rs_counts_cap = 0;
    int32_t* rs_inl = nullptr;
    ransac::Result* rs_res = nullptr;
    // semantic ground plane scratch (device)
    unsigned char* sem_img = nullptr;
    size_t sem_img_cap = 0;
    float* sem_coeffs = nullptr;  // dummy prior, first fit, second fit
    unsigned char* sem_groups = nullptr;  // per 64 points: the moment sums of the group's members (ransac::GroupSums)
    size_t sem_groups_cap = 0;
    ransac::SemResult* sem_res = nullptr;
    // tracklet gather/scatter scratch (device)
    double* trk_uv_cur = nullptr;
    double* trk_uv_last = nullptr;
    double* trk_depth_cur = nullptr;
    double* trk_depth_last = nullptr;
    int32_t* trk_type_cur = nullptr;
    int32_t* trk_type_last = nullptr;
    int32_t* trk_rank = nullptr;
    long long* trk_n_new = nullptr;
    size_t trk_cap = 0;
    // batched tracklet layer (mld_tracklets_depths_device): per-sequence scratch, one allocation per kind
    unsigned char* trkb = nullptr;   // [uv_cur | uv_last | depth_cur | depth_last | type_cur | type_last | rank | n_new]
    size_t trkb_cap = 0;             // tracks per sequence
    int trkb_seqs = 0;
    TrkSeq* trkb_desc = nullptr;     // device copy of the per-sequence descriptors
    std::vector<TrkSeq> trkb_host;
    // feature groups (mld_tracklets_depths_device with few sequences): G descriptor copies per slot, each over a slice of
    // the slot's features with its own queue slices and counters - a slot's classification is then G blocks instead of one
    SlotDesc* d_sub = nullptr;
    int32_t* sub_counts = nullptr;  // [overflow, live] per group descriptor
    size_t sub_cap = 0;             // group descriptors allocated
    std::vector<SlotDesc> h_sub;
    // staging for the host-pointer tracklet entry point
    unsigned char* trk_stage = nullptr;
    size_t trk_stage_cap = 0;
    // one-frame-per-call host path (mld_calculate_depth_frame): pinned host staging + its device image
    unsigned char* fr_host = nullptr;  // pinned: [inlier indices | uv] in, [depth | type] out
    unsigned char* fr_dev = nullptr;
    size_t fr_cap = 0;
    unsigned char* fr_host_dev = nullptr;  // the pinned block as the kernels address it (results are written there directly)
    bool fr_zero_copy = true;              // false (test build: MLD_FRAME_COPY=1): results through device memory + a D2H copy
    bool fr_helper_on = true;              // false (test build: MLD_FRAME_HELPER=0): the calling thread queues the side-stream work
    hipEvent_t fr_ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // phase marks of a timed one-frame call
    double fr_us[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                    // mld_frame_timing
    // batched ground-plane estimation (mld_set_clouds_estimate_planes_device)
    uint32_t* rsb_masks = nullptr;   // inlier bitmasks of all slots, contiguous
    size_t rsb_mask_words = 0;       // per slot
    PlaneDev* rsb_planes = nullptr;  // one per slot
    uint32_t* rsb_seeds = nullptr;
    hipStream_t side = nullptr;  // the plane's inlier mask is built here while the cloud is still in flight
    hipEvent_t side_done = nullptr, side_start = nullptr;
    FrameHelper* helper = nullptr;  // stages and queues the side-stream work of a one-frame call beside the cloud copy
    hipEvent_t order_ev = nullptr;  // mld_order_after
    mld_ctx* release_waiter = nullptr;  // mld_order_after_classify: released behind this context's next k_classify
    mld_ctx* waiting_on = nullptr;      //   (back pointer: either context may be destroyed first)
    bool order_wait_pending = false;
    uint32_t* cls_done = nullptr;       //   gate hand-over: k_classify blocks of this context that have been placed (device)
    uint32_t cls_target = 0;            //   ... and how many there will be once everything queued so far has run
    const uint32_t* gate_counter = nullptr;  // the counter / value this context's next projection waits for (k_gate)
    uint32_t gate_target = 0;
    bool gate_mode = true;              //   hand over through k_gate (a polling wavefront) instead of a cross-stream event
    Calib* d_calib = nullptr;      // device copies of `calib` for the feature kernels (fields fetched where they are used):
    Calib calib_uploaded[2]{};      //   [0] the context's, [1] a call's override; what the copies hold
    mld_ctx* gate_src = nullptr;        //   the context whose cls_done this context's pending gate reads (its gate_waiter is this one)
    mld_ctx* gate_waiter = nullptr;     //   the context whose pending gate reads cls_done (either may be destroyed first)    //   order_ev is recorded; the next projection launch of this context waits for it
    // mld_pair_contexts: the batched projections of two contexts share ONE stream (back to back, no cross-stream
    // hand-over between them); each context's feature kernels stay on its own stream, joined by two events per batch
    hipStream_t proj_stream = nullptr;  // nullptr: projections run on `stream`
    struct ProjShare* proj_share = nullptr;  // the pair's stream, reference counted: the last context to go destroys it
    hipEvent_t proj_fork = nullptr, proj_join = nullptr;
    size_t lds_fused_pad = 0;       // mld_set_shared_gpu
    int queue_probe_tries = 0;      // separate_stream_from_live_contexts: probes run when the context was created
    int queue_shared_with_live = 0; // ... live contexts whose hardware queue its stream still shares (0 = none)
    int shared_arg = 0;             // its last argument (re-applied when the list capacities change)
    int fused_blocks_per_cu = 8;    // mld_set_shared_gpu: wavefronts of k_feature_fused per CU in the shared mode
    size_t lds_per_cu = 0;          // device property
    bool timing = false;
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t> event_pool;
    // Small host -> device uploads of a step (slot and sequence descriptors, constants) leave from a ring of pinned
    // generations owned by the context and are moved by a kernel (k_upload), not by hipMemcpyAsync: with several such
    // copies in flight the runtime spreads them over further DMA engines, and the first copy on an engine sets its queue up
    // - ~6 ms inside that hipMemcpyAsync on the submitting thread, once or twice somewhere in a run of short steps
    // (LAB.md 5.16: the "bimodal" legs of rounds 4 and 5).
    static constexpr int kUpGens = 16;
    unsigned char* up_base = nullptr;
    size_t up_gen_bytes = 0;
    int up_next = 0;
    hipEvent_t up_ev[kUpGens] = {};
    bool up_busy[kUpGens] = {};
};

namespace {

#define HIP_TRY(ctx, call)                                                                                  \
    do {                                                                                                    \
        hipError_t e_ = (call);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                 \
            return MLD_ERR_HIP;                                                                             \
        }                                                                                                   \
    } while (0)

int fail(mld_ctx* ctx, int code, const std::string& msg) {
    if (ctx) ctx->err = msg;
    return code;
}

// 3x3 inverse by cofactors (Eigen compute_inverse<Matrix3d>): result(i,j) = cofactor<j,i> / det.
double cof(const double m[9], int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
void inverse3(const double m[9], double out[9]) {
    double c0 = cof(m, 0, 0), c1 = cof(m, 1, 0), c2 = cof(m, 2, 0);
    double det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
    double invdet = 1.0 / det;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[i * 3 + j] = cof(m, j, i) * invdet;
}

// Upper bound on the cells one search window can cover: floor(2*half) + 2 columns / rows.
int window_cells(double halfX, double halfY) {
    int nx = (int)std::floor(2.0 * halfX) + 2, ny = (int)std::floor(2.0 * halfY) + 2;
    if (nx < 1) nx = 1;
    if (ny < 1) ny = 1;
    return nx * ny;
}

int validate_params(const mld_params& P, std::string& why) {
    if (P.neighbor_search_mode != 0) {
        why = "neighbor_search_mode has the invalid value: " + std::to_string(P.neighbor_search_mode);  // :57
        return MLD_ERR_UNSUPPORTED_MODE;
    }
    if (P.do_use_depth_segmentation) {
        why = "DepthEstimator: Region growing not supported!";  // :608
        return MLD_ERR_UNSUPPORTED_MODE;
    }
    if (P.do_use_ransac_plane) {
        if (P.plane_estimator_use_triangle_maximation) {
        } else if (P.plane_estimator_use_leastsquares) {
            why = "plane_estimator_use_leastsquares (Ceres variant) is not supported: out-of-bounds read in the "
                  "reference (PlaneEstimationLeastSquares.cpp:19,41)";
            return MLD_ERR_UNSUPPORTED_MODE;
        } else if (P.plane_estimator_use_mestimator) {
        } else {
            why = "No road depth estimator selected.";  // :94
            return MLD_ERR_NO_ROAD_ESTIMATOR;
        }
    }
    if (P.pixelarea_search_witdh < 0 || P.pixelarea_search_height < 0) {
        why = "negative search window";
        return MLD_ERR_INVALID_ARG;
    }
    return MLD_OK;
}

void build_calib(mld_ctx* ctx, const double T[12]) {
    Calib& c = ctx->calib;
    const mld_params& P = ctx->P;
    std::memcpy(c.T, T, sizeof(double) * 12);
    // _transform_cam_to_lidar = transform_lidar_to_cam.inverse()   (DepthEstimator.cpp:44)
    double L[9] = {T[0], T[1], T[2], T[4], T[5], T[6], T[8], T[9], T[10]}, Li[9];
    inverse3(L, Li);
    double t[3] = {T[3], T[7], T[11]};
    for (int i = 0; i < 3; i++) {
        c.Tinv[i * 4 + 0] = Li[i * 3 + 0];
        c.Tinv[i * 4 + 1] = Li[i * 3 + 1];
        c.Tinv[i * 4 + 2] = Li[i * 3 + 2];
        c.Tinv[i * 4 + 3] = (-Li[i * 3 + 0]) * t[0] + ((-Li[i * 3 + 1]) * t[1] + (-Li[i * 3 + 2]) * t[2]);
    }
    double K[9] = {ctx->cam.focal_length, 0, ctx->cam.principal_point_x, 0, ctx->cam.focal_length,
                   ctx->cam.principal_point_y, 0, 0, 1};
    inverse3(K, c.Kinv);
    c.f = ctx->cam.focal_length;
    c.cu = ctx->cam.principal_point_x;
    c.cv = ctx->cam.principal_point_y;
    for (int t = 0; t < 12; t++) c.Tf[t] = (float)T[t];
    for (int r = 0; r < 3; r++) {
        float m = 0.f;
        for (int k = 0; k < 3; k++) m = std::max(m, std::fabs(c.Tf[4 * r + k]));
        c.Tfmax[r] = m * 1.000001f;
    }
    c.ff = (float)c.f;
    c.cuf = (float)c.cu;
    c.cvf = (float)c.cv;
    {
        // 1e-5 x (sum of the absolute values of the terms), the sums bounded by rowmax * m1 + |t|; evaluated in double
        // and rounded up, so the margins are never below the 1e-5 * S the kernel's comment promises
        const double e = 1e-5, up = 1.000001;
        const double Wf = (double)(float)ctx->cam.width, Hf = (double)(float)ctx->cam.height;
        const double tx = std::fabs((double)c.Tf[3]), ty = std::fabs((double)c.Tf[7]), tz = std::fabs((double)c.Tf[11]);
        const double fa = std::fabs((double)c.ff), ua = std::fabs((double)c.cuf) + Wf, va = std::fabs((double)c.cvf) + Hf;
        c.pcm[0] = (float)(e * (double)c.Tfmax[2] * up);
        c.pcm[1] = (float)(e * tz * up);
        c.pcm[2] = (float)(e * (fa * (double)c.Tfmax[0] + ua * (double)c.Tfmax[2]) * up);
        c.pcm[3] = (float)(e * (fa * tx + ua * tz) * up);
        c.pcm[4] = (float)(e * (fa * (double)c.Tfmax[1] + va * (double)c.Tfmax[2]) * up);
        c.pcm[5] = (float)(e * (fa * ty + va * tz) * up);
    }
    {
        // How far T^-1 (T p + t) + t' can land from p when evaluated in f64 (DepthEstimator.cpp:810 after :169-173): the
        // residual of the computed inverse (in long double) plus a generous bound on the roundings of the two products.
        // k_project_scatter's single-precision far test treats a point as undecided within a margin built on this.
        long double lin = 0.0L, cst = 0.0L, mlin = 0.0L, mcst = 0.0L;
        for (int i = 0; i < 3; i++) {
            long double rl = 0.0L, ml = 0.0L;
            for (int j = 0; j < 3; j++) {
                long double rij = (i == j) ? -1.0L : 0.0L, mij = 0.0L;
                for (int k = 0; k < 3; k++) {
                    rij += (long double)c.Tinv[4 * i + k] * (long double)T[4 * k + j];
                    mij += fabsl((long double)c.Tinv[4 * i + k]) * fabsl((long double)T[4 * k + j]);
                }
                rl += fabsl(rij);
                ml += mij;
            }
            long double ri = (long double)c.Tinv[4 * i + 3], mi = fabsl((long double)c.Tinv[4 * i + 3]);
            for (int k = 0; k < 3; k++) {
                ri += (long double)c.Tinv[4 * i + k] * (long double)T[4 * k + 3];
                mi += fabsl((long double)c.Tinv[4 * i + k]) * fabsl((long double)T[4 * k + 3]);
            }
            lin = std::max(lin, rl);
            mlin = std::max(mlin, ml);
            cst = std::max(cst, fabsl(ri));
            mcst = std::max(mcst, mi);
        }
        const long double u64 = 1.1102230246251565e-16L;
        const long double el = lin + 64.0L * u64 * mlin, ec = cst + 64.0L * u64 * mcst;
        const float inf = std::numeric_limits<float>::infinity();
        c.far_elin = (std::isfinite((double)el) && el < 1e30L) ? (float)((double)el * 1.000001) : inf;
        c.far_econst = (std::isfinite((double)ec) && ec < 1e30L) ? (float)((double)ec * 1.000001) : inf;
        if (!(c.far_elin > 0.f)) c.far_elin = 1e-37f;  // (never below what a float can hold)
    }
    c.W = ctx->cam.width;
    c.H = ctx->cam.height;
    c.bmStride = (ctx->cam.height + 16 + 3) / 4 * 4;  // words per 32-pixel column: H rows + 16 rows of read slack
    // NeighborFinderPixel.cpp:67-68, scales (1,1) and (2.0f,1.5f) (DepthEstimator.cpp:509,585)
    c.halfX1 = (double)P.pixelarea_search_witdh * 0.5 * (double)1.0f;
    c.halfY1 = (double)P.pixelarea_search_height * 0.5 * (double)1.0f;
    c.halfX2 = (double)P.pixelarea_search_witdh * 0.5 * (double)2.0f;
    c.halfY2 = (double)P.pixelarea_search_height * 0.5 * (double)1.5f;
    int cap = window_cells(c.halfX1, c.halfY1);
    if (P.do_use_ransac_plane) cap = std::max(cap, window_cells(c.halfX2, c.halfY2));
    cap = (cap + 1) & ~1;
    c.cap = cap;
    c.binW = P.histogram_segmentation_bin_witdh;
    c.minCount = P.histogram_segmentation_min_pointcount;
    c.countMin = (unsigned)P.radiusSearch_count_min;
    c.useHist = P.do_use_histogram_segmentation;
    c.thrG_en = P.treshold_depth_enabled;
    c.thrG_mode = P.treshold_depth_mode;
    c.thrG_min = (double)P.treshold_depth_min;
    c.thrG_max = (double)P.treshold_depth_max;
    c.thrL_en = P.treshold_depth_local_enabled;
    c.thrL_mode = P.treshold_depth_local_mode;
    c.thrL_type = P.treshold_depth_local_valuetype;
    c.thrL_val = P.treshold_depth_local_value;
    c.useTriMax = P.do_use_triangle_size_maximation;
    c.checkPlanar = P.do_check_triangleplanar_condition;
    c.planarThr = P.triangleplanar_crossnorm_treshold;
    c.orthThr = P.viewray_plane_orthoganality_treshold;
    c.cutBehind = P.do_use_cut_behind_camera;
    c.useRoad = P.do_use_ransac_plane;
    c.roadMode = P.plane_estimator_use_triangle_maximation ? 1 : 0;
    c.roadDistThr = P.ransac_plane_point_distance_treshold;
    c.roadDistThrF = (float)c.roadDistThr;
    c.padf2_ = 0.f;
    c.zxMinRel = P.plane_estimator_z_x_min_relation;
    c.usePCA = P.do_use_PCA;
    c.pcaAbsMin = P.pca_treshold_3_abs_min;
    c.pcaRelMax = P.pca_treshold_3_2_rel_max;
    c.pcaRelMin = P.pca_treshold_2_1_rel_min;
    ctx->lds_bytes = (size_t)cap * (3 * sizeof(double) + 2 * sizeof(int));
    c.threadPath = 1;
    c.xcdAware = 1;
    c.sortClasses = 4;
    // list capacities of the fused kernel: up to 32 entries for the scanned (road) window, up to 24 for the narrow one,
    // 40 entries of LDS per lane for the two together (the narrow list sits behind the lane's wide list) - 10 KB per
    // wavefront: 16 wavefronts per CU, four per SIMD; longer lists overflow to the wave-cooperative kernel
    c.k1max = 32;
    c.kMain = 24;
    c.kTotal = 40;
#ifdef MLD_AB_SWITCHES
    // Test / measurement build only (libmld_hip_ab.so): the shipped library has one code path and reads no environment.
    //   MLD_FORCE_WAVE_PATH=1  every feature through the wave-cooperative kernel (the parity suite runs both paths)
    //   MLD_FORCE_THREAD_PATH=1  single-frame calls through the lane-per-feature kernel as well (they default to the
    //                            wave-cooperative one: launch_features)
    //   MLD_NO_XCD=1           plain block -> slot mapping        MLD_K1MAX / MLD_KMAIN   list capacities
    if (const char* e = std::getenv("MLD_FORCE_WAVE_PATH")) c.threadPath = (e[0] == '1') ? 0 : 1;
    if (const char* e = std::getenv("MLD_NO_XCD")) c.xcdAware = (e[0] == '1') ? 0 : 1;
    //   MLD_SORT_CLASSES=1     live queue in row order alone (A/B of the count-class order)
    if (const char* e = std::getenv("MLD_SORT_CLASSES")) c.sortClasses = (e[0] == '4') ? 4 : 1;
    if (const char* e = std::getenv("MLD_FORCE_THREAD_PATH")) ctx->force_thread_path = e[0] == '1';
    //   MLD_PROJ_LDS=bytes     occupancy experiments: the batched projection asks for that much (unused) LDS per block
    if (const char* e = std::getenv("MLD_PROJ_LDS")) ctx->proj_lds = (size_t)std::atoll(e);
    //   MLD_GATE=0             mld_order_after_classify hands over through a cross-stream event instead of k_gate (A/B)
    if (const char* e = std::getenv("MLD_GATE")) ctx->gate_mode = e[0] != '0';
    //   MLD_FRAME_COPY=1       one-frame calls return their results through device memory and a D2H copy (A/B of the
    //                          direct stores into the pinned block)
    if (const char* e = std::getenv("MLD_FRAME_COPY")) ctx->fr_zero_copy = e[0] != '1';
    //   MLD_FRAME_HELPER=0     one-frame calls without the helper thread (A/B)
    if (const char* e = std::getenv("MLD_FRAME_HELPER")) ctx->fr_helper_on = e[0] != '0';
    if (const char* e = std::getenv("MLD_K1MAX")) c.k1max = std::min(std::max(std::atoi(e), 8), kK1MaxLimit);
    if (const char* e = std::getenv("MLD_KMAIN")) c.kMain = std::min(std::max(std::atoi(e), 8), c.k1max);
    if (std::getenv("MLD_K1MAX") || std::getenv("MLD_KMAIN")) c.kTotal = c.k1max + c.kMain;
    //   MLD_KTOTAL=n           LDS entries per lane shared by the two lists (k1max <= n <= k1max + kMain)
    if (const char* e = std::getenv("MLD_KTOTAL")) c.kTotal = std::min(std::max(std::atoi(e), c.k1max), c.k1max + c.kMain);
#endif
    ctx->lds_fused = (size_t)c.kTotal * kWave * sizeof(uint32_t);
    ctx->bm_ncol = (ctx->cam.width + 31) / 32 + 1;
    ctx->bm_ncolp = ctx->bm_ncol | 1;
    const size_t cls_fixed = (size_t)(kClsBuckets + kClsThreads / kWave + 4) * sizeof(int);
    const size_t cls_bitmap = (size_t)c.bmStride * (size_t)ctx->bm_ncolp * sizeof(uint32_t);
    ctx->classify_staged = cls_fixed + cls_bitmap <= 128 * 1024;
#ifdef MLD_AB_SWITCHES
    //   MLD_CLASSIFY_STAGED=0  k_classify reads the occupancy bitmap in place (4 KB of LDS per block instead of 68)
    if (const char* e = std::getenv("MLD_CLASSIFY_STAGED")) ctx->classify_staged = ctx->classify_staged && e[0] != '0';
#endif
    ctx->lds_classify = cls_fixed + (ctx->classify_staged ? cls_bitmap : 0);
}

// Batched projections of a paired context (mld_pair_contexts) run on the pair's projection stream: it first waits for
// everything queued on the context's own stream so far (the feature kernels that still read the slots' maps), and the
// context's stream then waits for the projection.  Unpaired: both are the context's own stream, nothing to do.
hipStream_t projection_fork(mld_ctx* ctx, int& rc) {
    rc = MLD_OK;
    if (!ctx->proj_stream) return ctx->stream;
    hipError_t e = hipEventRecord(ctx->proj_fork, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->proj_stream, ctx->proj_fork, 0);
    if (e != hipSuccess) {
        ctx->err = std::string("projection stream (fork): ") + hipGetErrorString(e);
        rc = MLD_ERR_HIP;
    }
    return ctx->proj_stream;
}
int projection_join(mld_ctx* ctx) {
    if (!ctx->proj_stream) return MLD_OK;
    HIP_TRY(ctx, hipEventRecord(ctx->proj_join, ctx->proj_stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->proj_join, 0));
    return MLD_OK;
}
// After a successful fork every exit joins: whatever was queued on the projection stream before an error is ordered
// before the context's own stream again (mld_synchronize relies on it).  finish() is the normal exit and reports the
// join's own status; the destructor covers the early returns without touching the error text of the failed call.
struct ProjectionScope {
    mld_ctx* ctx;
    bool open = true;
    explicit ProjectionScope(mld_ctx* c) : ctx(c) {}
    int finish() {
        open = false;
        return projection_join(ctx);
    }
    ~ProjectionScope() {
        if (open && ctx->proj_stream) {
            const std::string keep = ctx->err;
            (void)projection_join(ctx);
            ctx->err = keep;
        }
    }
};

// The dynamic-LDS limit of k_rs_batch is an attribute of the FUNCTION on a device, not of a context: contexts with
// clouds of different sizes share it, so the size enabled so far is kept per device (it only ever grows).
int enable_rs_batch_lds(mld_ctx* ctx, size_t lds) {
    static size_t enabled[64] = {};
    static std::mutex mtx;  // (contexts of one device may be driven from different host threads)
    std::lock_guard<std::mutex> lk(mtx);
    size_t& cur = enabled[ctx->device & 63];
    if (lds <= cur) return MLD_OK;
    HIP_TRY(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(ransac::k_rs_batch),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    cur = lds;
    return MLD_OK;
}

int check_slot(mld_ctx* ctx, int slot) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (slot < 0 || slot >= (int)ctx->slots.size()) return fail(ctx, MLD_ERR_INVALID_ARG, "slot out of range");
    return MLD_OK;
}

int bind_device(mld_ctx* ctx) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return MLD_OK;
}

template <typename T>
int grow(mld_ctx* ctx, T*& ptr, size_t& cap, size_t need) {
    if (need <= cap && ptr) return MLD_OK;
    if (ptr) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ptr));
        ptr = nullptr;
    }
    size_t n = need < 1 ? 1 : need;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)));
    cap = n;
    return MLD_OK;
}

hipEvent_t get_event(mld_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct ScopedTimer {
    mld_ctx* ctx;
    TimedLaunch t{};
    bool on;
    hipStream_t st;
    ScopedTimer(mld_ctx* c, int which, hipStream_t stream = nullptr) : ctx(c), on(c->timing), st(stream ? stream : c->stream) {
        if (on) {
            t.e0 = get_event(ctx);
            t.e1 = get_event(ctx);
            t.which = which;
            (void)hipEventRecord(t.e0, st);
        }
    }
    ~ScopedTimer() {
        if (on) {
            (void)hipEventRecord(t.e1, st);
            ctx->timed.push_back(t);
        }
    }
};

int check_cloud_args(mld_ctx* ctx, const void* dev_ptr, int64_t n, int stride) {
    if (n < 0 || n > kMaxPoints) return fail(ctx, MLD_ERR_CAPACITY, "cloud larger than 8 388 607 points");
    if (stride != 16 && stride != 32) return fail(ctx, MLD_ERR_INVALID_ARG, "stride_bytes must be 16 or 32");
    if (!dev_ptr && n > 0) return fail(ctx, MLD_ERR_INVALID_ARG, "null cloud pointer");
    if (((size_t)dev_ptr & 3) != 0) return fail(ctx, MLD_ERR_INVALID_ARG, "cloud pointer must be 4-byte aligned");
    return MLD_OK;
}

// New cloud for a slot: bump the map tag (zero-fill on wrap), forget the previous plane / debug data.
int begin_cloud(mld_ctx* ctx, Slot& s, const void* dev_ptr, int64_t n, int stride, bool clear_bitmap = true,
                hipStream_t st = nullptr, bool map_cleared = false) {
    if (!st) st = ctx->stream;
    if (int rc = check_cloud_args(ctx, dev_ptr, n, stride)) return rc;
    if (clear_bitmap) HIP_TRY(ctx, hipMemsetAsync(s.d.bitmap, 0, ctx->bitmap_words * sizeof(uint32_t), st));
    if (s.d.tag >= kMaxTag) {
        if (!map_cleared)
            HIP_TRY(ctx, hipMemsetAsync(s.d.map, 0,
                                        ((size_t)ctx->cam.width * ctx->cam.height + kMapPadCells) * sizeof(uint32_t), st));
        s.d.tag = 1;
    } else {
        s.d.tag += 1;
    }
    // an empty cloud may come with a null pointer (e.g. the data pointer of an empty tensor): the kernels' padding
    // loads of "point 0" still need an address
    s.d.cloud = (n > 0) ? static_cast<const unsigned char*>(dev_ptr) : ctx->dummy;
    s.d.n = n;
    s.d.stride = stride;
    s.d.has_plane = 0;
    s.d.inlier_mask = nullptr;
    s.d.mask_in_key = 0;
    s.d.plane_dev = nullptr;
    s.cloud_set = true;
    s.plane_decided = false;
    s.full_valid = false;
    return MLD_OK;
}

// Tag wrap of a whole batch (every slot of the range is at the last tag: the steady state of a batch that is always
// projected together): ONE fill over the slots' contiguous maps instead of one per slot (1024 fills of 1.86 MB cost
// 6 ms every 127 batches - 6 % of a step; one fill of 1.9 GB 0.5 ms).  Returns true when it cleared them.
bool clear_maps_on_common_wrap(mld_ctx* ctx, int first, int n_slots, hipStream_t st, int& rc) {
    rc = MLD_OK;
    for (int i = 0; i < n_slots; i++)
        if (ctx->slots[first + i].d.tag < kMaxTag) return false;
    hipError_t e = hipMemsetAsync(ctx->map_slab + (size_t)first * ctx->map_stride, 0,
                                  ctx->map_stride * (size_t)n_slots * sizeof(uint32_t), st);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMemsetAsync(maps): ") + hipGetErrorString(e);
        rc = MLD_ERR_HIP;
        return false;
    }
    return true;
}

// The map tag shared by slots [0, n_slots), or 0 when they differ (then the per-slot tags of the uploaded
// descriptors are used).
uint32_t common_tag(mld_ctx* ctx, int n_slots, int first = 0) {
    const uint32_t t = ctx->slots[first].d.tag;
    for (int i = 1; i < n_slots; i++)
        if (ctx->slots[first + i].d.tag != t) return 0u;
    return t;
}

int launch_project(mld_ctx* ctx, int n_slots, int64_t max_n, bool single, int slot, hipStream_t st = nullptr) {
    if (!st) st = ctx->stream;
    if (max_n <= 0) return MLD_OK;
    const int per_block = kProjThreads * kProjPerThread;
    int per_slot = (int)((max_n + per_block - 1) / per_block);
#ifdef MLD_AB_SWITCHES
    //   MLD_DIAG_PROJ_FRACTION=p   diagnostic (WRONG results): batches project only the first p percent of every cloud -
    //                              how the two-context step would answer to a faster projection
    if (!single)
        if (const char* e = std::getenv("MLD_DIAG_PROJ_FRACTION")) per_slot = std::max(1, per_slot * std::atoi(e) / 100);
#endif
    // (clouds on 16-byte boundaries - every allocator's - take the instantiation without the scalar-load alternative)
    bool aligned = true;
    for (int i = slot; i < slot + (single ? 1 : n_slots); i++)
        aligned = aligned && (((uintptr_t)ctx->slots[i].d.cloud) & 15) == 0;
    // magic = ceil(2^32 / per_slot): the kernel divides block indices by per_slot with a multiplication (div_by_magic)
    const uint32_t ps_magic = per_slot > 1 ? (uint32_t)(((1ull << 32) + (uint64_t)per_slot - 1) / (uint64_t)per_slot) : 0u;
    if (ctx->order_wait_pending) {  // mld_order_after_classify
        ctx->order_wait_pending = false;
        if (ctx->gate_counter) {
            hipLaunchKernelGGL(mld::k_gate, dim3(1), dim3(kWave), 0, st, ctx->gate_counter, ctx->gate_target, 100000);
            ctx->gate_counter = nullptr;
            // (the gate's source, not `waiting_on`: the context may have been re-armed behind another one since)
            if (ctx->gate_src && ctx->gate_src->gate_waiter == ctx) ctx->gate_src->gate_waiter = nullptr;
            ctx->gate_src = nullptr;
        } else {
            HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->order_ev, 0));
        }
    }
    ScopedTimer tm(ctx, 0, st);
    if (single) {
        auto kp = aligned ? mld::k_project_scatter<true, true> : mld::k_project_scatter<false, true>;
        hipLaunchKernelGGL(kp, dim3(per_slot), dim3(kProjThreads), 0, st, ctx->d_slots, ctx->slots[slot].d, ctx->calib, 1,
                           per_slot, ps_magic, 0u);
    } else {
        // (batch: the slots [slot, slot + n_slots))
        auto kp = aligned ? mld::k_project_scatter<true, false> : mld::k_project_scatter<false, false>;
        hipLaunchKernelGGL(kp, dim3((unsigned)per_slot * n_slots), dim3(kProjThreads), ctx->proj_lds, st, ctx->d_slots + slot,
                           SlotDesc{}, ctx->calib, n_slots, per_slot, ps_magic, common_tag(ctx, n_slots, slot));
    }
    HIP_TRY(ctx, hipGetLastError());
    return MLD_OK;
}

// The work queues of a slot hold one entry per feature: the live queue (feature indices in row order, written by
// k_classify) and the overflow queue ((feature, code) pairs for k_feature_wave).
int ensure_queues(mld_ctx* ctx, Slot& s, int64_t F) {
    if ((size_t)F <= s.road_cap) return MLD_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!s.queues_in_slab) {
        if (s.d.ovf_queue) HIP_TRY(ctx, hipFree(s.d.ovf_queue));
        if (s.d.live_queue) HIP_TRY(ctx, hipFree(s.d.live_queue));
    }
    s.queues_in_slab = false;
    s.d.live_queue = nullptr;
    s.d.ovf_queue = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&s.d.ovf_queue, (size_t)F * 2 * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&s.d.live_queue, (size_t)F * sizeof(int32_t)));
    s.road_cap = (size_t)F;
    return MLD_OK;
}

int upload_descs(mld_ctx* ctx, int n_slots, hipStream_t st = nullptr, int first = 0, bool tags_in_descs = false);

// `bytes` of host memory to `dst` on `st` through the context's pinned ring (see mld_ctx::up_base): `src` may be rewritten
// as soon as this returns; the call waits only when the host runs kUpGens uploads ahead of the device.
int upload_small(mld_ctx* ctx, void* dst, const void* src, size_t bytes, hipStream_t st) {
    static_assert(sizeof(SlotDesc) % 4 == 0 && sizeof(TrkSeq) % 4 == 0 && sizeof(Calib) % 4 == 0, "k_upload moves 32-bit words");
    if (!bytes) return MLD_OK;
    if (bytes % 4) return fail(ctx, MLD_ERR_INVALID_ARG, "upload_small: size not a multiple of 4");
    if (bytes > ctx->up_gen_bytes) return fail(ctx, MLD_ERR_CAPACITY, "upload_small: table larger than the context's upload ring");
    const int g = ctx->up_next;
    if (ctx->up_busy[g]) HIP_TRY(ctx, hipEventSynchronize(ctx->up_ev[g]));
    if (!ctx->up_ev[g]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->up_ev[g], hipEventDisableTiming));
    unsigned char* stage = ctx->up_base + (size_t)g * ctx->up_gen_bytes;
    std::memcpy(stage, src, bytes);
    const int words = (int)(bytes / 4);  // (every table is made of 4- and 8-byte fields)
    hipLaunchKernelGGL(mld::k_upload, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, st, reinterpret_cast<uint32_t*>(dst),
                       reinterpret_cast<const uint32_t*>(stage), words);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->up_ev[g], st));
    ctx->up_busy[g] = true;
    ctx->up_next = (g + 1) % mld_ctx::kUpGens;
    return MLD_OK;
}

// k_classify (per slot) -> k_feature_fused over the live queues -> k_feature_wave over the overflow queues.
// k_classify sets both queue lengths, so no counter needs clearing.
int launch_features(mld_ctx* ctx, int n_slots, int64_t max_F, bool single, int slot, const Calib* override_calib = nullptr,
                    const SlotDesc* group_descs = nullptr) {
    // group_descs: n_slots descriptors in device memory that are not the context's slots (feature groups: tags inside)
    if (max_F <= 0) return MLD_OK;
    const SlotDesc* const descs = group_descs ? group_descs : ctx->d_slots;
    Calib calib = override_calib ? *override_calib : ctx->calib;
    // ONE frame per call (the ROS usage, the tracklet path): too few wavefronts for the lane-per-feature kernel to be
    // anything but one wavefront's lifetime (41 us for 2000 features); the wave-cooperative kernel, one feature per
    // wavefront, finishes the frame in the time of its slowest feature, and settles the dead features itself.
    const bool few = single && max_F <= 16384 && !ctx->force_thread_path;
    if (few) calib.threadPath = 0;
    const int per_slot = (int)((max_F + kWave - 1) / kWave);
    const uint32_t tag_all = (single || group_descs) ? 0u : common_tag(ctx, n_slots);
    const SlotDesc one = single ? ctx->slots[slot].d : SlotDesc{};
    const int use_single = single ? 1 : 0, ns = single ? 1 : n_slots;
    if (few) {
        // (the wave kernel settles the features without neighbours itself: k_feature_wave, direct mode)
    } else {
        ScopedTimer ts(ctx, 5);
        auto kc = ctx->classify_staged ? mld::k_classify<true, kClsThreads, kClsKeep>
                                       : mld::k_classify<false, kClsThreads, kClsKeep>;
        const bool gate = ctx->gate_mode && ctx->release_waiter && !single;
        if (gate && !ctx->cls_done) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->cls_done, sizeof(uint32_t)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->cls_done, 0, sizeof(uint32_t), ctx->stream));
        }
        hipLaunchKernelGGL(kc, dim3((unsigned)ns), dim3(kClsThreads), ctx->lds_classify, ctx->stream, descs, one,
                           use_single, calib, ctx->bm_ncol, ctx->bm_ncolp, gate ? ctx->cls_done : (uint32_t*)nullptr);
        if (gate) {
            mld_ctx* w = ctx->release_waiter;
            ctx->release_waiter = nullptr;
            w->waiting_on = nullptr;
            ctx->cls_target += (uint32_t)ns;
            // an earlier gate that was never consumed: of `w` on another context, or of another context on this one
            if (w->gate_src && w->gate_src != ctx && w->gate_src->gate_waiter == w) w->gate_src->gate_waiter = nullptr;
            if (ctx->gate_waiter && ctx->gate_waiter != w) {
                ctx->gate_waiter->gate_counter = nullptr;
                ctx->gate_waiter->order_wait_pending = false;
                ctx->gate_waiter->gate_src = nullptr;
            }
            w->gate_counter = ctx->cls_done;
            w->gate_target = ctx->cls_target;
            w->order_wait_pending = true;
            w->gate_src = ctx;
            ctx->gate_waiter = w;
        }
    }
    // mld_order_after_classify: the waiting context is released here - behind the classification (which wants every
    // CU's wave slots and LDS for 40 us), ahead of the long feature kernels
    if (ctx->release_waiter) {
        mld_ctx* w = ctx->release_waiter;
        ctx->release_waiter = nullptr;
        w->waiting_on = nullptr;
        // The waiting context inserts the wait right in front of its next projection KERNEL (launch_project): what it
        // queues ahead of that kernel - the batch's bitmap fill, descriptor upload, a map clear on tag wrap - does not
        // wait and is out of the way when the projection is released.
        HIP_TRY(ctx, hipEventRecord(w->order_ev, ctx->stream));
        // (a stale, unconsumed gate must not be polled instead of this event)
        if (w->gate_src && w->gate_src->gate_waiter == w) w->gate_src->gate_waiter = nullptr;
        w->gate_src = nullptr;
        w->gate_counter = nullptr;
        w->order_wait_pending = true;
    }
    int rc_up = MLD_OK;
    // The two feature kernels read the per-context constants in device memory: entry 0 = the context's own (uploaded when
    // they change: creation, mld_set_list_capacity), entry 1 = the variant of a call that overrides them (the per-call
    // "no road fallback" of mld_calculate_depth_opts).  `threadPath` is left out of the comparison: only k_classify -
    // which takes its copy by value - looks at it, and the one-frame / debug routes differ from the context in nothing else.
    const Calib* d_calib = ctx->d_calib;
    {
        Calib want = calib;
        want.threadPath = ctx->calib.threadPath;
        const int which = std::memcmp(&want, &ctx->calib, sizeof(Calib)) == 0 ? 0 : 1;
        if (std::memcmp(&want, &ctx->calib_uploaded[which], sizeof(Calib)) != 0) {
            // (pageable source: staged by the runtime before the call returns)
            int rc_c = upload_small(ctx, ctx->d_calib + which, &want, sizeof(Calib), ctx->stream);
            if (rc_c) return rc_c;
            ctx->calib_uploaded[which] = want;
        }
        d_calib = ctx->d_calib + which;
    }
    if (calib.threadPath) {
        ScopedTimer tm(ctx, 1);
        // long lists (mld_set_list_capacity beyond the default 32 / 24: dense clouds) take a DENSE instantiation: 2 = two
        // wavefronts per SIMD, which is what their LDS allows anyway, and the registers of the third for the in-register
        // corner search (up to 24 points); 1 = in the shared-GPU mode, whose point is to leave registers to the other
        // context's projection: the same within 168 registers (corner search up to 16 points)
        int dense = (calib.k1max > 32 || calib.kMain > 24) ? ((ctx->shared_arg & 1) ? 1 : 2) : 0;
#ifdef MLD_AB_SWITCHES
        //   MLD_FORCE_DENSE=0|1|2  (measurement) the instantiation of the lane-per-feature kernel, whatever the capacities
        if (const char* e = std::getenv("MLD_FORCE_DENSE")) dense = std::min(std::max(std::atoi(e), 0), 2);
#endif
        auto kf = dense == 2 ? (calib.roadMode ? mld::k_feature_fused<1, 2> : mld::k_feature_fused<0, 2>)
                  : dense == 1 ? (calib.roadMode ? mld::k_feature_fused<1, 1> : mld::k_feature_fused<0, 1>)
                               : (calib.roadMode ? mld::k_feature_fused<1, 0> : mld::k_feature_fused<0, 0>);
        // the kernel reads its slot descriptors in device memory (a single-slot call - the lane-per-feature kernel runs
        // for one frame in the test routes only - uploads that slot's first; batches have theirs in place, tags included:
        // the kernel never needs the map tag)
        if (single && (rc_up = upload_descs(ctx, 1, nullptr, slot, true))) return rc_up;
        hipLaunchKernelGGL(kf, dim3((unsigned)per_slot * (unsigned)ns), dim3(kWave), ctx->lds_fused + ctx->lds_fused_pad, ctx->stream,
                           descs + (single ? slot : 0), d_calib, ns, per_slot);
    }
    {
        ScopedTimer tm(ctx, 3);
        // queue entries per block and iteration: a wavefront works on ONE feature at a time.  Batches: eight, about
        // 4096 blocks in all (the queues are empty for KITTI-like clouds and hold every feature for dense ones); a
        // single frame: one or two entries per block and up to 16384 blocks (config 5, 10 000 tracks: 65 -> 52 us)
        const int chunk = few ? (int)std::min<int64_t>(8, std::max<int64_t>(1, (max_F + 8191) / 8192)) : 8;
        const int want = (int)((max_F + chunk - 1) / chunk);
        const int pw = std::max(1, std::min(want, std::max(4, (few ? 16384 : 4096) / ns)));
        auto kw = few ? mld::k_feature_wave<true> : mld::k_feature_wave<false>;
        hipLaunchKernelGGL(kw, dim3((unsigned)pw * (unsigned)ns), dim3(kWave), ctx->lds_bytes, ctx->stream, descs, one,
                           use_single, d_calib, ns, pw, tag_all, chunk);
    }
    HIP_TRY(ctx, hipGetLastError());
    return MLD_OK;
}

int upload_descs(mld_ctx* ctx, int n_slots, hipStream_t st, int first, bool tags_in_descs) {
    if (!st) st = ctx->stream;
    // Steady-state batches (same buffers every step) change nothing but the map tags, and a tag common to the
    // batch travels as a kernel argument: skip the upload when the device copy is still right.
    const bool tags_by_arg = !tags_in_descs && common_tag(ctx, n_slots, first) != 0u;
    // The host cache (h_descs = what the device holds) is committed only AFTER the upload was queued: a failed upload
    // (ring capacity, a HIP error) must leave the range dirty, or the next call would skip it and the kernels - which read
    // the device copy only - would run on stale descriptors.
    bool dirty = false;
    ctx->desc_stage.resize((size_t)n_slots);
    for (int i = first; i < first + n_slots; i++) {
        SlotDesc& d = ctx->desc_stage[(size_t)(i - first)];
        d = ctx->slots[i].d;
        if (tags_by_arg) d.tag = 0;
        dirty = dirty || std::memcmp(&d, &ctx->h_descs[i], sizeof(SlotDesc)) != 0;
    }
    if (!dirty) return MLD_OK;
    const int rc = upload_small(ctx, ctx->d_slots + first, ctx->desc_stage.data(), sizeof(SlotDesc) * n_slots, st);
    if (rc != MLD_OK) return rc;
    std::memcpy(ctx->h_descs.data() + first, ctx->desc_stage.data(), sizeof(SlotDesc) * n_slots);
    return MLD_OK;
}

int precheck_calc(mld_ctx* ctx, Slot& s, int64_t F) {
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "call of 'CalculateDepth' without 'SetInputCloud'");
    if (F < 0) return fail(ctx, MLD_ERR_INVALID_ARG, "negative feature count");
    if (F > 0x7FFFFFFFLL) return fail(ctx, MLD_ERR_CAPACITY, "more than 2^31-1 features in one call");  // queues hold int32 indices
    if (ctx->P.do_use_ransac_plane && !s.plane_decided && !ctx->P.set_all_depths_to_zero)
        return fail(ctx, MLD_ERR_NO_GROUND_PLANE,
                    "do_use_ransac_plane is set but no ground plane was supplied for this cloud "
                    "(mld_set_ground_plane; pass NULL coefficients for 'no plane')");
    return MLD_OK;
}

// Commits the plane coefficients of a slot: coeffs + the M-estimator prior (DepthEstimator.cpp:286-292).  Callers
// validate their arguments and build the inlier mask BEFORE this, so that a failed call leaves the slot as it was.
void set_plane_coeffs(mld_ctx* ctx, Slot& s, const float coeffs[4]) {
    std::memcpy(s.d.coeffs, coeffs, sizeof(float) * 4);
    far_margins(coeffs, ctx->calib.far_elin, ctx->calib.far_econst, ctx->calib.roadDistThrF, s.d.far_mg0, s.d.far_mg1);
    // DepthEstimator.cpp:289-291: prior = Hyperplane(Vector3d(a,b,c).normalized(), d)
    double a = (double)coeffs[0], b = (double)coeffs[1], cc = (double)coeffs[2];
    double z = a * a + (b * b + cc * cc);
    if (z > 0.0) {
        double nrm = std::sqrt(z);
        a /= nrm;
        b /= nrm;
        cc /= nrm;
    }
    s.d.prior_n[0] = a;
    s.d.prior_n[1] = b;
    s.d.prior_n[2] = cc;
    s.d.prior_off = (double)coeffs[3];
    s.d.has_plane = 1;
    s.d.plane_dev = nullptr;
    s.plane_decided = true;
}

// "ransacPlane == nullptr" (DepthEstimator.cpp:580): the road fallback is skipped for this cloud.
void clear_plane(Slot& s) {
    s.d.has_plane = 0;
    s.d.inlier_mask = nullptr;
    s.d.mask_in_key = 0;
    s.d.plane_dev = nullptr;
    s.plane_decided = true;
}

int build_mask_from_indices(mld_ctx* ctx, Slot& s, const int32_t* idx_dev, int64_t n_inl) {
    size_t words = (size_t)((s.d.n + 31) / 32);
    int rc = grow(ctx, s.mask_buf, s.mask_words, words);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(s.mask_buf, 0, (words < 1 ? 1 : words) * sizeof(uint32_t), ctx->stream));
    if (n_inl > 0) {
        int blocks = (int)((n_inl + 255) / 256);
        hipLaunchKernelGGL(k_build_mask, dim3(blocks), dim3(256), 0, ctx->stream, idx_dev, (long long)n_inl,
                           (long long)s.d.n, s.mask_buf);
        HIP_TRY(ctx, hipGetLastError());
    }
    return MLD_OK;
}

int ensure_full(mld_ctx* ctx, Slot& s) {
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "no cloud set for this slot");
    if (s.full_valid) return MLD_OK;
    size_t n = (size_t)s.d.n;
    if (n > s.dbg_cap || !s.cam) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        void* olds[] = {s.cam, s.img, s.vis, s.rank, s.pidx, s.img_vis, s.block_sums, s.d_total};
        for (void* p : olds)
            if (p) HIP_TRY(ctx, hipFree(p));
        size_t m = n < 1 ? 1 : n;
        size_t nb = (m + kScanBlock - 1) / kScanBlock;
        HIP_TRY(ctx, hipMalloc((void**)&s.cam, m * 3 * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void**)&s.img, m * 2 * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void**)&s.vis, m * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&s.rank, m * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&s.pidx, m * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&s.img_vis, m * 2 * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void**)&s.block_sums, nb * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&s.d_total, sizeof(int32_t)));
        s.dbg_cap = m;
    }
    int32_t total = 0;
    if (n > 0) {
        int nb = (int)((n + kScanBlock - 1) / kScanBlock);
        hipLaunchKernelGGL(k_project_full, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, s.d,
                           ctx->calib, s.cam, s.img, s.vis);
        hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(kScanBlock), 0, ctx->stream, s.vis, (long long)n,
                           s.block_sums);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, ctx->stream, s.block_sums, nb, s.d_total);
        hipLaunchKernelGGL(k_scan_final, dim3(nb), dim3(kScanBlock), 0, ctx->stream, s.vis, (long long)n,
                           s.block_sums, s.img, s.rank, s.pidx, s.img_vis);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipMemcpyAsync(&total, s.d_total, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    s.nvis = total;
    s.full_valid = true;
    return MLD_OK;
}

}  // namespace

extern "C" {

const char* mld_create_error(void) { return g_create_error.c_str(); }

// ---------------------------------------------------------------------------- hardware queues of the contexts' streams
// Contexts of one device are meant to run side by side (mld_order_after[_classify], mld_set_shared_gpu), but the HIP
// runtime hands a new stream one of a few hardware queues - four per priority by default (GPU_MAX_HW_QUEUES), shared with
// every other stream of the process: the caller's, torch's stream pool ... - and streams on one queue execute strictly in
// turn.  Two contexts that landed on one queue lose their whole overlap, silently and for the life of the process (seen in
// round 6: the secondary bench legs ran 1.37x slower in a process whose earlier legs had left streams behind).  So a new
// context PROBES its stream against the stream of every live context of its device (k_probe_wait / k_probe_set) and, while
// it shares a queue with one of them, takes another stream (the rejected ones are held until the search ends, so that the
// runtime's least-used-queue choice moves on).
namespace {
std::mutex g_live_mu;
std::vector<mld_ctx*> g_live;  // contexts between mld_create and mld_destroy

// 1 = the streams run side by side, 0 = they share a hardware queue (or the GPU was too busy to tell), < 0 = HIP error
int probe_streams_concurrent(hipStream_t waiter, hipStream_t setter) {
    uint32_t* host = nullptr;  // [0] flag, [16] result: pinned, device-visible
    if (hipHostMalloc((void**)&host, 128, hipHostMallocDefault) != hipSuccess) return -1;
    host[0] = 0u;
    host[16] = 0u;
    int out = -1;
    // both streams idle first: the setter must not sit behind queued work of its own stream
    if (hipStreamSynchronize(setter) == hipSuccess && hipStreamSynchronize(waiter) == hipSuccess) {
        hipLaunchKernelGGL(mld::k_probe_wait, dim3(1), dim3(kWave), 0, waiter, host, host + 16, 3000);  // gives up after ~3 ms
        hipLaunchKernelGGL(mld::k_probe_set, dim3(1), dim3(kWave), 0, setter, host);
        if (hipGetLastError() == hipSuccess && hipStreamSynchronize(waiter) == hipSuccess &&
            hipStreamSynchronize(setter) == hipSuccess)
            out = host[16] == 1u ? 1 : 0;
    }
    (void)hipHostFree(host);
    return out;
}

// The new context's stream: distinct hardware queue from every live context of the device, if a handful of tries finds one.
void separate_stream_from_live_contexts(mld_ctx* ctx) {
    std::lock_guard<std::mutex> lk(g_live_mu);
    std::vector<hipStream_t> rejected;
    ctx->queue_probe_tries = 0;
    ctx->queue_shared_with_live = 0;
    for (int attempt = 0; attempt < 6; attempt++) {
        int shared = 0;
        for (mld_ctx* o : g_live) {
            if (o == ctx || o->device != ctx->device || !o->stream) continue;
            const int r = probe_streams_concurrent(ctx->stream, o->stream);
            ctx->queue_probe_tries++;
            if (r == 0) shared++;
            if (r < 0) {  // (a HIP error here is not this function's to report: the context works either way)
                (void)hipGetLastError();
                shared = 0;
                attempt = 99;
                break;
            }
        }
        ctx->queue_shared_with_live = shared;
        if (!shared || attempt >= 5) break;
        hipStream_t fresh = nullptr;
        if (hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking) != hipSuccess) break;
        rejected.push_back(ctx->stream);
        ctx->stream = fresh;
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    g_live.push_back(ctx);
}
}  // namespace

mld_ctx* mld_create(const mld_params* params, const mld_camera* camera, const double T_cam_lidar[12], int device,
                    int max_frames, int64_t max_points, int64_t max_features, int* status_out) {
    auto bail = [&](int code, const std::string& msg) -> mld_ctx* {
        g_create_error = msg;
        if (status_out) *status_out = code;
        return nullptr;
    };
    if (status_out) *status_out = MLD_OK;
    if (!params || !camera || !T_cam_lidar) return bail(MLD_ERR_INVALID_ARG, "null argument");
    if (max_frames < 1) return bail(MLD_ERR_INVALID_ARG, "max_frames must be >= 1");
    if (camera->width < 1 || camera->height < 1) return bail(MLD_ERR_INVALID_ARG, "bad image size");
    // pixel-map cells are addressed with 32-bit offsets and 24-bit multiplies (k_project_scatter)
    if (camera->width > 65535 || camera->height > 65535 || (int64_t)camera->width * camera->height > (int64_t)1 << 30)
        return bail(MLD_ERR_CAPACITY, "image larger than 65535 pixels on a side or 2^30 pixels in all");
    // a singular or non-finite intrinsic matrix has no inverse (camera_pinhole.h:65 would produce NaN rays)
    if (!std::isfinite(camera->focal_length) || camera->focal_length == 0.0 || !std::isfinite(camera->principal_point_x) ||
        !std::isfinite(camera->principal_point_y))
        return bail(MLD_ERR_INVALID_ARG, "bad camera intrinsics");
    for (int t = 0; t < 12; t++)
        if (!std::isfinite(T_cam_lidar[t])) return bail(MLD_ERR_INVALID_ARG, "non-finite lidar->camera transform");
    if (max_points > kMaxPoints) return bail(MLD_ERR_CAPACITY, "max_points exceeds 8 388 607");
    std::string why;
    int v = validate_params(*params, why);
    if (v != MLD_OK) return bail(v, why);

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return bail(MLD_ERR_HIP, std::string("no HIP device available (") + hipGetErrorString(e) +
                                     "): the DepthEstimator path has no CPU fallback");
    if (device < 0 || device >= ndev) return bail(MLD_ERR_INVALID_ARG, "device index out of range");

    mld_ctx* ctx = new mld_ctx();
    ctx->P = *params;
    ctx->cam = *camera;
    ctx->device = device;
    build_calib(ctx, T_cam_lidar);
    auto hip_bail = [&](hipError_t err, const char* what) -> mld_ctx* {
        std::string m = std::string(what) + ": " + hipGetErrorString(err);
        mld_destroy(ctx);
        return bail(MLD_ERR_HIP, m);
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return hip_bail(e, "hipSetDevice");
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return hip_bail(e, "hipGetDeviceProperties");
    ctx->lds_per_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;  // 160 KB on gfx950
    if (ctx->lds_bytes > (size_t)prop.sharedMemPerBlock) {
        mld_destroy(ctx);
        return bail(MLD_ERR_CAPACITY, "search window too large for the LDS-staged neighbour list");
    }
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
        return hip_bail(e, "hipStreamCreate");
    separate_stream_from_live_contexts(ctx);  // (a hardware queue of its own among the device's contexts)
    if (ctx->lds_bytes > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(mld::k_feature_wave<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)ctx->lds_bytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(mld::k_feature_wave<false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_bytes);
        if (e != hipSuccess) return hip_bail(e, "hipFuncSetAttribute(k_feature_wave)");
    }
    if (ctx->lds_classify > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(mld::k_classify<true, kClsThreads, kClsKeep>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_classify);

        if (e != hipSuccess) return hip_bail(e, "hipFuncSetAttribute(k_classify)");
    }
    if (ctx->lds_fused > 48 * 1024) {
        const void* fused[] = {reinterpret_cast<const void*>(mld::k_feature_fused<0, 0>),
                               reinterpret_cast<const void*>(mld::k_feature_fused<1, 0>),
                               reinterpret_cast<const void*>(mld::k_feature_fused<0, 1>),
                               reinterpret_cast<const void*>(mld::k_feature_fused<1, 1>),
                               reinterpret_cast<const void*>(mld::k_feature_fused<0, 2>),
                               reinterpret_cast<const void*>(mld::k_feature_fused<1, 2>)};
        for (const void* f : fused) {
            e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ctx->lds_fused);
            if (e != hipSuccess) return hip_bail(e, "hipFuncSetAttribute(k_feature_fused)");
        }
    }
    ctx->slots.resize(max_frames);
    ctx->h_descs.resize(max_frames);
    if ((e = hipMalloc((void**)&ctx->d_slots, sizeof(SlotDesc) * max_frames)) != hipSuccess)
        return hip_bail(e, "hipMalloc(slots)");
    if ((e = hipMalloc((void**)&ctx->dummy, 256)) != hipSuccess) return hip_bail(e, "hipMalloc(dummy)");
    if ((e = hipMalloc((void**)&ctx->d_calib, 2 * sizeof(Calib))) != hipSuccess) return hip_bail(e, "hipMalloc(calib)");
    std::memset(ctx->calib_uploaded, 0xFF, sizeof(ctx->calib_uploaded));  // (nothing uploaded yet)
    // the upload ring at its working size now (pinning costs milliseconds): a step's largest upload is its descriptors
    // (contexts with several slots may deal their slots' features to group descriptors - 256 of them, room for 512:
    // mld_tracklets_depths_device)
    constexpr size_t kMaxGroupDescs = 256;  // mld_tracklets_depths_device: G <= 256 / n_desc groups of n_desc descriptors
    static_assert(kMaxGroupDescs <= 512, "the group descriptors of a launch set must fit one generation of the upload ring");
    // (a launch set of n_seq sequences uses 2 n_seq <= max_frames slots: its TrkSeq table fits where its descriptors do)
    static_assert(sizeof(TrkSeq) <= 2 * sizeof(SlotDesc), "a sequence's TrkSeq must fit the ring share of its two slots");
    const size_t up_descs = max_frames >= 2 ? std::max<size_t>((size_t)max_frames, 512) : 1;
    ctx->up_gen_bytes = (std::max(sizeof(SlotDesc) * up_descs, sizeof(Calib)) + 4095) / 4096 * 4096;
    if ((e = hipHostMalloc((void**)&ctx->up_base, ctx->up_gen_bytes * mld_ctx::kUpGens, hipHostMallocDefault)) != hipSuccess)
        return hip_bail(e, "hipHostMalloc(upload ring)");
    if ((e = hipMemsetAsync(ctx->dummy, 0, 256, ctx->stream)) != hipSuccess) return hip_bail(e, "hipMemset(dummy)");
    size_t cells = (size_t)camera->width * camera->height + kMapPadCells;
    ctx->bitmap_words = (size_t)ctx->calib.bmStride * (size_t)((camera->width + 31) / 32 + 1) + 4;  // + a slack column
    // one allocation: [queue lengths: max_frames road fallback + max_frames long-list overflow][bitmaps of the slots]
    const size_t cnt_words = 2 * (size_t)max_frames;
    if ((e = hipMalloc((void**)&ctx->queue_counts, (cnt_words + ctx->bitmap_words * (size_t)max_frames) * sizeof(uint32_t))) != hipSuccess)
        return hip_bail(e, "hipMalloc(bitmaps)");
    ctx->bitmaps = reinterpret_cast<uint32_t*>(ctx->queue_counts) + cnt_words;
    if ((e = hipMemsetAsync(ctx->queue_counts, 0, cnt_words * sizeof(uint32_t), ctx->stream)) != hipSuccess)
        return hip_bail(e, "hipMemset(counters)");
    for (size_t si = 0; si < ctx->slots.size(); si++) ctx->slots[si].d.bitmap = ctx->bitmaps + si * ctx->bitmap_words;
    for (size_t si = 0; si < ctx->slots.size(); si++) {
        ctx->slots[si].d.live_count = ctx->queue_counts + si;
        ctx->slots[si].d.ovf_count = ctx->queue_counts + (size_t)max_frames + si;
    }
    const size_t map_stride = (cells + 63) & ~(size_t)63;  // words; slots start on 256-byte boundaries
    ctx->map_stride = map_stride;
    if ((e = hipMalloc((void**)&ctx->map_slab, map_stride * (size_t)max_frames * sizeof(uint32_t))) != hipSuccess)
        return hip_bail(e, "hipMalloc(maps)");
    if ((e = hipMemsetAsync(ctx->map_slab, 0, map_stride * (size_t)max_frames * sizeof(uint32_t), ctx->stream)) != hipSuccess)
        return hip_bail(e, "hipMemset(maps)");
    const size_t qF = max_features > 0 ? (((size_t)max_features + 63) & ~(size_t)63) : 0;  // queue entries per slot
    const size_t q_stride = qF * 3;  // int32 words per slot: overflow queue (2F) + live queue (F)
    if (qF && (e = hipMalloc((void**)&ctx->queue_slab, q_stride * (size_t)max_frames * sizeof(int32_t))) != hipSuccess)
        return hip_bail(e, "hipMalloc(queues)");
    for (size_t si = 0; si < ctx->slots.size(); si++) {
        Slot& s = ctx->slots[si];
        s.d.map = ctx->map_slab + si * map_stride;
        s.d.tag = 0;
        if (qF) {
            // work queues of the feature kernels: no allocation in the first CalculateDepth
            int32_t* q = ctx->queue_slab + si * q_stride;
            s.d.ovf_queue = q;
            s.d.live_queue = q + 2 * qF;
            s.road_cap = qF;
            s.queues_in_slab = true;
        }
        if (max_points > 0) {
            if ((e = hipMalloc((void**)&s.cloud_buf, (size_t)max_points * 32)) != hipSuccess)
                return hip_bail(e, "hipMalloc(cloud)");
            s.cloud_cap = (size_t)max_points * 32;
        }
        if (max_features > 0) {
            size_t F = (size_t)max_features;
            if ((e = hipMalloc((void**)&s.uv_buf, F * 2 * sizeof(double))) != hipSuccess) return hip_bail(e, "hipMalloc(uv)");
            if ((e = hipMalloc((void**)&s.depth_buf, F * sizeof(double))) != hipSuccess) return hip_bail(e, "hipMalloc(depth)");
            if ((e = hipMalloc((void**)&s.type_buf, F * sizeof(int32_t))) != hipSuccess) return hip_bail(e, "hipMalloc(type)");
            s.feat_cap = F;
        }
    }
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return hip_bail(e, "hipStreamSynchronize");
    return ctx;
}

void mld_destroy(mld_ctx* ctx) {
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.erase(std::remove(g_live.begin(), g_live.end(), ctx), g_live.end());
    }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (Slot& s : ctx->slots) {
        if (!s.queues_in_slab) {
            if (s.d.ovf_queue) (void)hipFree(s.d.ovf_queue);
            if (s.d.live_queue) (void)hipFree(s.d.live_queue);
        }
        void* ptrs[] = {s.cloud_buf, s.uv_buf, s.depth_buf, s.type_buf, s.inl_buf,    s.mask_buf, s.cam,
                        s.img,       s.vis,    s.rank,      s.pidx,     s.img_vis,    s.block_sums, s.d_total};
        for (void* p : ptrs)
            if (p) (void)hipFree(p);
    }
    if (ctx->map_slab) (void)hipFree(ctx->map_slab);
    if (ctx->queue_slab) (void)hipFree(ctx->queue_slab);
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    if (ctx->dummy) (void)hipFree(ctx->dummy);
    if (ctx->d_calib) (void)hipFree(ctx->d_calib);
    if (ctx->queue_counts) (void)hipFree(ctx->queue_counts);  // also holds the bitmaps
    void* rsp[] = {ctx->rs_flags, ctx->rs_cand, ctx->rs_block, ctx->rs_M, ctx->rs_S, ctx->rs_sample, ctx->rs_sp,
                   ctx->rs_counts, ctx->rs_inl, ctx->rs_res, ctx->sem_img, ctx->sem_coeffs, ctx->sem_res, ctx->sem_groups};
    for (void* p : rsp)
        if (p) (void)hipFree(p);
    if (ctx->trkb) (void)hipFree(ctx->trkb);
    if (ctx->trkb_desc) (void)hipFree(ctx->trkb_desc);
    if (ctx->d_sub) (void)hipFree(ctx->d_sub);
    if (ctx->sub_counts) (void)hipFree(ctx->sub_counts);
    void* trk[] = {ctx->trk_uv_cur, ctx->trk_uv_last, ctx->trk_depth_cur, ctx->trk_depth_last, ctx->trk_type_cur,
                   ctx->trk_type_last, ctx->trk_rank, ctx->trk_n_new, ctx->trk_stage};
    for (void* p : trk)
        if (p) (void)hipFree(p);
    if (ctx->rsb_masks) (void)hipFree(ctx->rsb_masks);
    if (ctx->rsb_planes) (void)hipFree(ctx->rsb_planes);
    if (ctx->rsb_seeds) (void)hipFree(ctx->rsb_seeds);
    if (ctx->fr_host) (void)hipHostFree(ctx->fr_host);
    if (ctx->fr_dev) (void)hipFree(ctx->fr_dev);
    if (ctx->helper) {
        ctx->helper->stop();
        delete ctx->helper;
    }
    // a pending mld_order_after_classify hand-over dies with either of its contexts
    if (ctx->waiting_on && ctx->waiting_on->release_waiter == ctx) ctx->waiting_on->release_waiter = nullptr;
    if (ctx->gate_src && ctx->gate_src->gate_waiter == ctx) ctx->gate_src->gate_waiter = nullptr;
    if (ctx->release_waiter) ctx->release_waiter->waiting_on = nullptr;
    if (ctx->gate_waiter) {  // (its gate would poll a counter that is about to be freed)
        ctx->gate_waiter->gate_counter = nullptr;
        ctx->gate_waiter->order_wait_pending = false;
        ctx->gate_waiter->gate_src = nullptr;
    }
    if (ctx->cls_done) (void)hipFree(ctx->cls_done);
    for (hipEvent_t e : ctx->fr_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->side_start) (void)hipEventDestroy(ctx->side_start);
    if (ctx->side) (void)hipStreamDestroy(ctx->side);
    if (ctx->side_done) (void)hipEventDestroy(ctx->side_done);
    if (ctx->order_ev) (void)hipEventDestroy(ctx->order_ev);
    if (ctx->proj_stream) (void)hipStreamSynchronize(ctx->proj_stream);
    if (ctx->proj_fork) (void)hipEventDestroy(ctx->proj_fork);
    if (ctx->proj_join) (void)hipEventDestroy(ctx->proj_join);
    if (ctx->proj_share && --ctx->proj_share->refs == 0) {  // the partner is gone already (or never came to be)
        (void)hipStreamDestroy(ctx->proj_share->stream);
        delete ctx->proj_share;
    }
    for (TimedLaunch& t : ctx->timed) {
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    for (hipEvent_t e : ctx->event_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->up_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->up_base) (void)hipHostFree(ctx->up_base);
    if (ctx->stream && ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char* mld_last_error(const mld_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void* mld_get_stream(mld_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int mld_contexts_concurrent(mld_ctx* a, mld_ctx* b) {
    if (!a || !b) return MLD_ERR_INVALID_ARG;
    if (a == b) return 0;
    if (a->device != b->device) return 1;
    int rc = bind_device(a);
    if (rc) return rc;
    const int r = probe_streams_concurrent(a->stream, b->stream);
    if (r < 0) {
        (void)hipGetLastError();
        return fail(a, MLD_ERR_HIP, "mld_contexts_concurrent: the probe kernels could not be run");
    }
    return r;
}

int mld_synchronize(mld_ctx* ctx) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    // (work on a pair's projection stream is always joined into the context's own stream before the call returns)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MLD_OK;
}

int mld_pair_contexts(mld_ctx* a, mld_ctx* b) {
    if (!a || !b || a == b) return MLD_ERR_INVALID_ARG;
    if (a->device != b->device) return fail(a, MLD_ERR_INVALID_ARG, "contexts live on different devices");
    if (a->proj_stream || b->proj_stream) return fail(a, MLD_ERR_INVALID_ARG, "context already paired");
    int rc = bind_device(a);
    if (rc) return rc;
    ProjShare* sh = new ProjShare();
    hipError_t e = hipSuccess;
#ifdef MLD_AB_SWITCHES
    // Measurement build only.  MLD_CU_SPLIT="N[,s]": the pair's projection stream is confined to N of the 256 CUs and both
    // contexts' own streams (classification, feature kernels) to the other 256 - N, so that the two kinds of kernels never
    // share a register file (LAB.md "CU split").  Mask bits: the low N bits, or with ",s" every bit whose index mod 256/g
    // falls in the first N/g (g = 8 groups), i.e. the same share of every group of 32 consecutive bits.
    bool split = false;
    if (const char* cs = std::getenv("MLD_CU_SPLIT")) {
        const int n_proj = std::atoi(cs);
        const bool strided = std::strchr(cs, 's') != nullptr;
        if (n_proj > 0 && n_proj < 256) {
            uint32_t mp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, mf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b = 0; b < 256; b++) {
                const bool proj = strided ? (b % 32) < n_proj / 8 : b < n_proj;
                (proj ? mp : mf)[b >> 5] |= 1u << (b & 31);
            }
            e = hipExtStreamCreateWithCUMask(&sh->stream, 8, mp);
            for (mld_ctx* c : {a, b}) {
                if (e != hipSuccess) break;
                (void)hipStreamSynchronize(c->stream);
                if (c->own_stream) (void)hipStreamDestroy(c->stream);
                c->stream = nullptr;
                e = hipExtStreamCreateWithCUMask(&c->stream, 8, mf);
                c->own_stream = true;
            }
            split = true;
        }
    }
    if (!split)
#endif
    e = hipStreamCreateWithFlags(&sh->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete sh;
        return fail(a, MLD_ERR_HIP, std::string("hipStreamCreate(projection stream): ") + hipGetErrorString(e));
    }
    for (mld_ctx* c : {a, b}) {  // (the stream is reference counted: the contexts may be destroyed in any order)
        c->proj_share = sh;
        c->proj_stream = sh->stream;
        sh->refs++;
    }
    for (mld_ctx* c : {a, b}) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->proj_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->proj_join, hipEventDisableTiming));
    }
    return MLD_OK;
}

int mld_order_after(mld_ctx* ctx, mld_ctx* other) {
    if (!ctx || !other) return MLD_ERR_INVALID_ARG;
    if (ctx == other) return MLD_OK;
    if (ctx->device != other->device) return fail(ctx, MLD_ERR_INVALID_ARG, "contexts live on different devices");
    int rc = bind_device(ctx);
    if (rc) return rc;
    if (!ctx->order_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->order_ev, other->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->order_ev, 0));
    return MLD_OK;
}

int mld_order_after_classify(mld_ctx* ctx, mld_ctx* other) {
    if (!ctx || !other) return MLD_ERR_INVALID_ARG;
    if (ctx == other) return MLD_OK;
    if (ctx->device != other->device) return fail(ctx, MLD_ERR_INVALID_ARG, "contexts live on different devices");
    int rc = bind_device(ctx);
    if (rc) return rc;
    if (!ctx->order_ev) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->order_ev, hipEventDisableTiming));
    if (ctx->waiting_on && ctx->waiting_on->release_waiter == ctx) ctx->waiting_on->release_waiter = nullptr;
    if (other->release_waiter) other->release_waiter->waiting_on = nullptr;
    other->release_waiter = ctx;  // (the event is recorded, and waited for, when `other` queues its next feature kernels)
    ctx->waiting_on = other;
    return MLD_OK;
}

int mld_set_shared_gpu(mld_ctx* ctx, int shared) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    // k_feature_fused is capped by its LDS request: 10 KB of lists per wavefront (the default budget) -> 16 per CU, four per
    // SIMD, when the context has the GPU to itself; beside another context's projection it is padded to 16 KB -> 10 per
    // CU, so that the register file keeps room for the projection's wavefronts (52 VGPRs each).  Round 6, one box, three
    // alternations: 8 / 10 / 12 per CU = 0.692-0.707 / 0.680-0.683 / 0.686-0.706 ms per step (LAB.md 6.5; rounds 2-5 ran 8).
    int blocks = (shared >> 8) & 0xFF;  // bits 8..15: wavefronts per CU (0 = the default, 10)
    if (blocks <= 0) blocks = 10;
    // 4..16 wavefronts per CU: fewer would ask for more dynamic LDS per one-wave block (53 KB and up) than the kernel
    // is enabled for, more than 16 cannot be reached at its register count anyway
    if (blocks < 4 || blocks > 16 || (shared & ~0xFF01) != 0)
        return fail(ctx, MLD_ERR_INVALID_ARG, "mld_set_shared_gpu: bit 0 = on, bits 8..15 = wavefronts per CU (0 or 4..16)");
    ctx->shared_arg = shared;
    ctx->fused_blocks_per_cu = blocks;
    const size_t per_cu = ctx->lds_per_cu ? ctx->lds_per_cu : 160 * 1024, want = (per_cu / (size_t)blocks) & ~(size_t)255;
    ctx->lds_fused_pad = ((shared & 1) && ctx->lds_fused < want) ? want - ctx->lds_fused : 0;
    return MLD_OK;
}

int mld_set_list_capacity(mld_ctx* ctx, int wide_entries, int narrow_entries) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (narrow_entries < 8 || narrow_entries > wide_entries || wide_entries > kK1MaxLimit)
        return fail(ctx, MLD_ERR_INVALID_ARG, "list capacities: 8 <= narrow <= wide <= 64");
    ctx->calib.k1max = wide_entries;
    ctx->calib.kMain = narrow_entries;
    ctx->calib.kTotal = wide_entries + narrow_entries;  // room for both lists at their longest; mld_set_list_budget lowers it
#ifdef MLD_AB_SWITCHES
    // (measurement: capacities / budget from the environment also where a caller sets them)
    if (const char* e = std::getenv("MLD_CAP_WIDE")) ctx->calib.k1max = wide_entries = std::min(std::max(std::atoi(e), 8), kK1MaxLimit);
    if (const char* e = std::getenv("MLD_CAP_NARROW")) ctx->calib.kMain = narrow_entries = std::min(std::max(std::atoi(e), 8), wide_entries);
    ctx->calib.kTotal = wide_entries + narrow_entries;
    if (const char* e = std::getenv("MLD_KTOTAL"))
        ctx->calib.kTotal = std::min(std::max(std::atoi(e), wide_entries), wide_entries + narrow_entries);
#endif
    ctx->lds_fused = (size_t)ctx->calib.kTotal * kWave * sizeof(uint32_t);
    return mld_set_shared_gpu(ctx, ctx->shared_arg);
}

int mld_set_list_budget(mld_ctx* ctx, int total_entries) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    const int wide = ctx->calib.k1max, both = ctx->calib.k1max + ctx->calib.kMain;
    if (total_entries == 0) total_entries = both;
    if (total_entries < wide || total_entries > both)
        return fail(ctx, MLD_ERR_INVALID_ARG, "list budget: wide capacity <= total <= wide + narrow capacity (0 = wide + narrow)");
    ctx->calib.kTotal = total_entries;
    ctx->lds_fused = (size_t)total_entries * kWave * sizeof(uint32_t);
    return mld_set_shared_gpu(ctx, ctx->shared_arg);
}

// ---------------------------------------------------------------------------- setInputCloud
int mld_set_cloud_device(mld_ctx* ctx, int slot, const void* pts_dev, int64_t n, int stride_bytes) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if ((rc = begin_cloud(ctx, s, pts_dev, n, stride_bytes))) return rc;
    return launch_project(ctx, 1, n, true, slot);
}

int mld_set_cloud(mld_ctx* ctx, int slot, const void* pts_host, int64_t n, int stride_bytes) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    if (n < 0 || (!pts_host && n > 0)) return fail(ctx, MLD_ERR_INVALID_ARG, "bad cloud");
    if (stride_bytes != 16 && stride_bytes != 32) return fail(ctx, MLD_ERR_INVALID_ARG, "stride_bytes must be 16 or 32");
    Slot& s = ctx->slots[slot];
    size_t bytes = (size_t)n * (size_t)stride_bytes;
    if ((rc = grow(ctx, s.cloud_buf, s.cloud_cap, bytes))) return rc;
    if (bytes) HIP_TRY(ctx, hipMemcpyAsync(s.cloud_buf, pts_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    if ((rc = begin_cloud(ctx, s, s.cloud_buf, n, stride_bytes))) return rc;
    return launch_project(ctx, 1, n, true, slot);
}

// setInputCloud for slots [0, n_slots) in one launch; coeffs / mask_dev (both or neither): the ground planes, known
// before the projection, whose inlier flags then travel in the map keys.
static int set_clouds_common(mld_ctx* ctx, int n_slots, const void* const* pts_dev, const int64_t* n, int stride_bytes,
                             const float* coeffs, const uint32_t* const* mask_dev, int first = 0) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (n_slots < 1 || first < 0 || first + n_slots > (int)ctx->slots.size() || !pts_dev || !n)
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad slot count / null arrays");
    int rc = bind_device(ctx);
    if (rc) return rc;
    if (mask_dev)
        for (int i = 0; i < n_slots; i++)
            if (!mask_dev[i]) return fail(ctx, MLD_ERR_INVALID_ARG, "null mask");
    // every slot's arguments are checked before anything is queued or any slot state changes
    for (int i = 0; i < n_slots; i++)
        if ((rc = check_cloud_args(ctx, pts_dev[i], n[i], stride_bytes))) return rc;
    int64_t max_n = 0;
    hipStream_t st = projection_fork(ctx, rc);
    if (rc) return rc;
    ProjectionScope joined(ctx);
    // the slots' occupancy bitmaps are contiguous: one fill for the whole batch
    HIP_TRY(ctx, hipMemsetAsync(ctx->bitmaps + (size_t)first * ctx->bitmap_words, 0,
                                ctx->bitmap_words * (size_t)n_slots * sizeof(uint32_t), st));
    const bool maps_cleared = clear_maps_on_common_wrap(ctx, first, n_slots, st, rc);
    if (rc) return rc;
    for (int i = 0; i < n_slots; i++) {
        Slot& s = ctx->slots[first + i];
        if ((rc = begin_cloud(ctx, s, pts_dev[i], n[i], stride_bytes, false, st, maps_cleared))) return rc;
        if (coeffs) {
            set_plane_coeffs(ctx, s, coeffs + 4 * i);
            s.d.inlier_mask = mask_dev[i];
            s.d.mask_in_key = 1;
        }
        max_n = std::max(max_n, n[i]);
    }
    if ((rc = upload_descs(ctx, n_slots, st, first))) return rc;
    if ((rc = launch_project(ctx, n_slots, max_n, false, first, st))) return rc;
    return joined.finish();
}

int mld_set_clouds_device(mld_ctx* ctx, int n_slots, const void* const* pts_dev, const int64_t* n, int stride_bytes) {
    return set_clouds_common(ctx, n_slots, pts_dev, n, stride_bytes, nullptr, nullptr);
}

int mld_set_clouds_planes_device(mld_ctx* ctx, int n_slots, const void* const* pts_dev, const int64_t* n, int stride_bytes,
                                 const float* coeffs, const uint32_t* const* mask_dev) {
    if (ctx && (!coeffs || !mask_dev)) return fail(ctx, MLD_ERR_INVALID_ARG, "null plane arrays");
    return set_clouds_common(ctx, n_slots, pts_dev, n, stride_bytes, coeffs, mask_dev);
}

// The same for the slots [first_slot, first_slot + n_slots): the batched tracklet layer keeps the current frames of its
// sequences in one bank of slots and the previous frames in the other.
int mld_set_clouds_planes_range_device(mld_ctx* ctx, int first_slot, int n_slots, const void* const* pts_dev,
                                       const int64_t* n, int stride_bytes, const float* coeffs,
                                       const uint32_t* const* mask_dev) {
    if (ctx && ((coeffs == nullptr) != (mask_dev == nullptr))) return fail(ctx, MLD_ERR_INVALID_ARG, "coeffs and masks: both or neither");
    return set_clouds_common(ctx, n_slots, pts_dev, n, stride_bytes, coeffs, mask_dev, first_slot);
}

// setInputCloud for slots [0, n_slots) with the ground plane ESTIMATED for every slot (the reference's default: the
// GroundPlane handed in is not segmented yet, DepthEstimator.cpp:275-283 -> RansacPlane::CalculateInliersPlane): one
// launch of k_rs_batch (a block per slot) ahead of the projection, nothing returns to the host.
int mld_set_clouds_estimate_planes_device(mld_ctx* ctx, int n_slots, const void* const* pts_dev, const int64_t* n,
                                          int stride_bytes, const uint32_t* seeds) {
    using namespace ransac;
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (n_slots < 1 || n_slots > (int)ctx->slots.size() || !pts_dev || !n || !seeds)
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad slot count / null arrays");
    int rc = bind_device(ctx);
    if (rc) return rc;
    const mld_params& P = ctx->P;
    if (!P.do_use_ransac_plane) return set_clouds_common(ctx, n_slots, pts_dev, n, stride_bytes, nullptr, nullptr);
    const int n_draws = P.ransac_plane_max_iterations + 1;
    if (n_draws < 1) return fail(ctx, MLD_ERR_INVALID_ARG, "ransac_plane_max_iterations must be >= 0");
    // every slot's arguments are checked before anything is queued or any slot state changes
    for (int i = 0; i < n_slots; i++)
        if ((rc = check_cloud_args(ctx, pts_dev[i], n[i], stride_bytes))) return rc;
    int64_t max_n = 0;
    for (int i = 0; i < n_slots; i++) max_n = std::max(max_n, n[i]);
    // z pass-through (RansacPlane.cpp:57-64) inside the batched kernel: one byte of LDS per 64 points + one int per 1024
    const bool pass = P.ransac_plane_min_z > -1001.;
    const size_t lds_fixed = (size_t)kSample * 5 * sizeof(float) + (size_t)kPartials * 9 * sizeof(float) +
                             (2 * kRsRound + kRsMisc + kRsEpochInts) * sizeof(int);
    const size_t n_chunks = (size_t)((max_n + 1023) / 1024);
    const size_t lds = lds_fixed + (pass ? (n_chunks + 1) * sizeof(int) + 16 * n_chunks + 16 : 0);
    if (lds > 158 * 1024) {
        // clouds beyond ~0.4 M points with the pass-through: the per-slot estimator (ordered compaction in device
        // memory), which synchronises once per slot
        if ((rc = set_clouds_common(ctx, n_slots, pts_dev, n, stride_bytes, nullptr, nullptr))) return rc;
        for (int i = 0; i < n_slots; i++) {
            rc = mld_estimate_ground_plane(ctx, i, seeds[i], nullptr, nullptr);
            if (rc == MLD_ERR_CLOUD_TOO_SMALL) {  // the frame goes on without a plane (the caller's catch, :321,338)
                clear_plane(ctx->slots[i]);
                rc = MLD_OK;
            }
            if (rc) return rc;
        }
        return MLD_OK;
    }
    const size_t words = (((size_t)((max_n + 31) / 32) + 2) & ~(size_t)1);  // even: the words double as 64-bit group masks
    if (!ctx->rsb_planes || words > ctx->rsb_mask_words) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->rsb_masks) HIP_TRY(ctx, hipFree(ctx->rsb_masks));
        ctx->rsb_masks = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rsb_masks, words * ctx->slots.size() * sizeof(uint32_t)));
        ctx->rsb_mask_words = words;
        if (!ctx->rsb_planes) {
            HIP_TRY(ctx, hipMalloc((void**)&ctx->rsb_planes, ctx->slots.size() * sizeof(PlaneDev)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->rsb_seeds, ctx->slots.size() * sizeof(uint32_t)));
        }
    }
    if ((rc = enable_rs_batch_lds(ctx, lds))) return rc;
    hipStream_t st = projection_fork(ctx, rc);
    if (rc) return rc;
    ProjectionScope joined(ctx);
    // occupancy bitmaps and inlier masks of the batch: one fill each
    HIP_TRY(ctx, hipMemsetAsync(ctx->bitmaps, 0, ctx->bitmap_words * (size_t)n_slots * sizeof(uint32_t), st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->rsb_masks, 0, ctx->rsb_mask_words * (size_t)n_slots * sizeof(uint32_t), st));
    if ((rc = upload_small(ctx, ctx->rsb_seeds, seeds, (size_t)n_slots * sizeof(uint32_t), st))) return rc;
    const bool maps_cleared = clear_maps_on_common_wrap(ctx, 0, n_slots, st, rc);
    if (rc) return rc;
    for (int i = 0; i < n_slots; i++) {
        Slot& s = ctx->slots[i];
        if ((rc = begin_cloud(ctx, s, pts_dev[i], n[i], stride_bytes, false, st, maps_cleared))) return rc;
        s.d.inlier_mask = ctx->rsb_masks + (size_t)i * ctx->rsb_mask_words;
        s.d.mask_in_key = 1;
        s.d.plane_dev = ctx->rsb_planes + i;
        s.d.has_plane = 1;  // the device copy decides (PlaneDev::has_plane)
        s.plane_decided = true;
    }
    if ((rc = upload_descs(ctx, n_slots, st))) return rc;
    RS_STAMP_LAUNCH(st, 0);
    {
        ScopedTimer tm(ctx, 4, st);
        hipLaunchKernelGGL(k_rs_batch, dim3((unsigned)n_slots), dim3(kRsThreads), lds, st, ctx->d_slots, ctx->rsb_seeds,
                           n_draws, P.ransac_plane_max_iterations, P.ransac_plane_probability,
                           P.ransac_plane_distance_treshold, P.ransac_plane_refinement_treshold,
                           P.ransac_plane_use_refinement, ctx->rsb_planes, pass ? 1 : 0, (float)P.ransac_plane_min_z,
                           (float)P.ransac_plane_max_z, ctx->calib.far_elin, ctx->calib.far_econst,
                           ctx->calib.roadDistThrF, -1, SlotDesc{}, 0u, (PlaneDev*)nullptr);
        HIP_TRY(ctx, hipGetLastError());
    }
    RS_STAMP_LAUNCH(st, 1);
    if ((rc = launch_project(ctx, n_slots, max_n, false, 0, st))) return rc;
    return joined.finish();
}

// The planes of the last mld_set_clouds_estimate_planes_device: coefficients (n_slots x 4), inlier counts and status
// (0 ok, 1 = GroundPlane::ExceptionPclInvalid: that frame runs without the road fallback).  Synchronises.
int mld_get_estimated_planes(mld_ctx* ctx, int n_slots, float* coeffs_out, int64_t* n_inliers_out, int32_t* status_out) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (n_slots < 1 || n_slots > (int)ctx->slots.size()) return fail(ctx, MLD_ERR_INVALID_ARG, "bad slot count");
    int rc = bind_device(ctx);
    if (rc) return rc;
    std::vector<PlaneDev> h((size_t)n_slots);
    if (ctx->rsb_planes)
        HIP_TRY(ctx, hipMemcpyAsync(h.data(), ctx->rsb_planes, h.size() * sizeof(PlaneDev), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < n_slots; i++) {
        const Slot& s = ctx->slots[i];
        const bool on_dev = s.d.plane_dev != nullptr && ctx->rsb_planes;
        if (coeffs_out)
            for (int t = 0; t < 4; t++) coeffs_out[4 * i + t] = on_dev ? h[i].coeffs[t] : (s.d.has_plane ? s.d.coeffs[t] : 0.f);
        if (status_out) status_out[i] = on_dev ? h[i].status : (s.d.has_plane ? 0 : 1);
        if (n_inliers_out) {
            if (on_dev) {
                n_inliers_out[i] = h[i].n_inliers;
            } else if (s.d.has_plane && s.d.inlier_mask) {
                // a slot whose plane was estimated by the per-slot path (or handed in): count its mask
                int64_t k = 0;
                int rc2 = mld_get_ground_plane_inliers(ctx, i, nullptr, 0, &k);
                if (rc2) return rc2;
                n_inliers_out[i] = k;
            } else {
                n_inliers_out[i] = 0;
            }
        }
    }
    return MLD_OK;
}

// ---------------------------------------------------------------------------- ground plane
int mld_set_ground_plane_device(mld_ctx* ctx, int slot, const float coeffs[4], const int32_t* inlier_idx_dev,
                                int64_t n_inliers) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "ground plane set before the slot's cloud");
    if (!coeffs) {
        clear_plane(s);
        return MLD_OK;
    }
    if (n_inliers < 0 || (!inlier_idx_dev && n_inliers > 0)) return fail(ctx, MLD_ERR_INVALID_ARG, "bad inlier list");
    // mask first, commit afterwards: a failure leaves the slot's previous plane state untouched
    if ((rc = build_mask_from_indices(ctx, s, inlier_idx_dev, n_inliers))) return rc;
    set_plane_coeffs(ctx, s, coeffs);
    s.d.inlier_mask = s.mask_buf;
    s.d.mask_in_key = 0;
    return MLD_OK;
}

int mld_set_ground_plane(mld_ctx* ctx, int slot, const float coeffs[4], const int32_t* inlier_idx_host,
                         int64_t n_inliers) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "ground plane set before the slot's cloud");
    if (!coeffs) {
        clear_plane(s);
        return MLD_OK;
    }
    if (n_inliers < 0 || (!inlier_idx_host && n_inliers > 0)) return fail(ctx, MLD_ERR_INVALID_ARG, "bad inlier list");
    if ((rc = grow(ctx, s.inl_buf, s.inl_cap, (size_t)n_inliers))) return rc;
    if (n_inliers)
        HIP_TRY(ctx, hipMemcpyAsync(s.inl_buf, inlier_idx_host, (size_t)n_inliers * sizeof(int32_t),
                                    hipMemcpyHostToDevice, ctx->stream));
    if ((rc = build_mask_from_indices(ctx, s, s.inl_buf, n_inliers))) return rc;
    set_plane_coeffs(ctx, s, coeffs);
    s.d.inlier_mask = s.mask_buf;
    s.d.mask_in_key = 0;
    return MLD_OK;
}

// per-point flags, their order-preserving compaction and the block sums in between
static int ensure_scan_buffers(mld_ctx* ctx, long long n) {
    if ((size_t)n <= ctx->rs_cap) return MLD_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    void* olds[] = {ctx->rs_flags, ctx->rs_cand, ctx->rs_block};
    for (void* p : olds)
        if (p) HIP_TRY(ctx, hipFree(p));
    ctx->rs_flags = nullptr;
    ctx->rs_cand = nullptr;
    ctx->rs_block = nullptr;
    ctx->rs_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_flags, (size_t)n * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_cand, (size_t)n * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_block, ((size_t)n / kScanBlock + 1) * sizeof(int32_t)));
    ctx->rs_cap = (size_t)n;
    return MLD_OK;
}

int mld_estimate_ground_plane(mld_ctx* ctx, int slot, uint32_t seed, float coeffs_out[4], int64_t* n_inliers_out) {
    using namespace ransac;
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "ground plane estimation before the slot's cloud");
    const mld_params& P = ctx->P;
    const long long n = s.d.n;
    if (n < 3) return fail(ctx, MLD_ERR_CLOUD_TOO_SMALL, "In GroundPlane: Input pointcloud is invalid");  // :44-50
    const int n_draws = P.ransac_plane_max_iterations + 1;
    if (n_draws < 1) return fail(ctx, MLD_ERR_INVALID_ARG, "ransac_plane_max_iterations must be >= 0");
    // scratch
    if (!ctx->rs_res) {
        if (!ctx->rs_M) HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_M, sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_S, sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_sample, kSample * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_sp, kSample * 3 * sizeof(float)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_inl, kSample * sizeof(int32_t)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_res, sizeof(Result)));
    }
    if ((size_t)n_draws > ctx->rs_counts_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->rs_counts) HIP_TRY(ctx, hipFree(ctx->rs_counts));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rs_counts, (size_t)n_draws * sizeof(int32_t)));
        ctx->rs_counts_cap = (size_t)n_draws;
    }
    const bool pass = P.ransac_plane_min_z > -1001.;  // RansacPlane.cpp:57
    if (pass) {
        if ((rc = ensure_scan_buffers(ctx, n))) return rc;
        const int nb = (int)((n + kScanBlock - 1) / kScanBlock);
        hipLaunchKernelGGL(k_rs_flags, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, s.d.cloud, n,
                           s.d.stride, (float)P.ransac_plane_min_z, (float)P.ransac_plane_max_z, ctx->rs_flags);
        hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(kScanBlock), 0, ctx->stream, ctx->rs_flags, n, ctx->rs_block);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(kScanBlock), 0, ctx->stream, ctx->rs_block, nb, ctx->rs_M);
        hipLaunchKernelGGL(k_rs_compact, dim3(nb), dim3(kScanBlock), 0, ctx->stream, ctx->rs_flags, n, ctx->rs_block,
                           ctx->rs_cand);
    }
    hipLaunchKernelGGL(k_rs_sample, dim3((kSample + 255) / 256), dim3(256), 0, ctx->stream, s.d.cloud, s.d.stride,
                       pass ? ctx->rs_cand : (int32_t*)nullptr, ctx->rs_M, n, seed, ctx->rs_sample, ctx->rs_sp, ctx->rs_S);
    hipLaunchKernelGGL(k_rs_hypotheses, dim3((unsigned)n_draws), dim3(kWave), 0, ctx->stream, ctx->rs_sp, ctx->rs_S, seed,
                       n_draws, P.ransac_plane_distance_treshold, ctx->rs_counts);
    hipLaunchKernelGGL(k_rs_select, dim3(1), dim3(64), 0, ctx->stream, ctx->rs_counts, ctx->rs_sp, ctx->rs_S, seed,
                       n_draws, P.ransac_plane_max_iterations, P.ransac_plane_probability, ctx->rs_res);
    const size_t words = (size_t)((n + 31) / 32);
    if ((rc = grow(ctx, s.mask_buf, s.mask_words, words))) return rc;
    HIP_TRY(ctx, hipMemsetAsync(s.mask_buf, 0, words * sizeof(uint32_t), ctx->stream));
    hipLaunchKernelGGL(k_rs_refine, dim3(1), dim3(kPartials), 0, ctx->stream, ctx->rs_sp, ctx->rs_sample,
                       P.ransac_plane_distance_treshold, P.ransac_plane_refinement_treshold,
                       P.ransac_plane_use_refinement, ctx->rs_inl, s.mask_buf, ctx->rs_res);
    HIP_TRY(ctx, hipGetLastError());
    Result res;
    HIP_TRY(ctx, hipMemcpyAsync(&res, ctx->rs_res, sizeof(Result), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (res.status != 0) {
        clear_plane(s);  // the slot's mask buffer has been overwritten
        s.plane_decided = false;
        return fail(ctx, MLD_ERR_CLOUD_TOO_SMALL, "In GroundPlane: Input pointcloud is invalid");
    }
    set_plane_coeffs(ctx, s, res.coeffs);
    s.d.inlier_mask = s.mask_buf;
    s.d.mask_in_key = 0;
    if (coeffs_out)
        for (int t = 0; t < 4; t++) coeffs_out[t] = res.coeffs[t];
    if (n_inliers_out) *n_inliers_out = res.n_inliers;
    return MLD_OK;
}

// SemanticPlane::CalculateInliersPlane (RansacPlane.cpp:195-274) for the slot's cloud, asynchronous part: four launches
// on stream `st` that leave coefficients / counts / status in ctx->sem_res, the inlier bitmask in s.mask_buf and - with
// pd - the plane in device memory.
static int semantic_plane_launch(mld_ctx* ctx, Slot& s, const unsigned char* img_dev, int rows, int cols, int row_stride,
                                 const int32_t* labels, int n_labels, double inlier_threshold, hipStream_t st,
                                 PlaneDev* pd = nullptr, PlaneDev* pd_copy = nullptr) {
    using namespace ransac;
    const long long n = s.d.n;
    LabelSet ls{};
    for (int i = 0; i < n_labels; i++)
        if (labels[i] >= 0 && labels[i] < 256) ls.w[labels[i] >> 5] |= 1u << (labels[i] & 31);
    SemCalib sc{};
    for (int t = 0; t < 12; t++) sc.T[t] = ctx->calib.T[t];
    sc.f = ctx->cam.focal_length;
    sc.cu = ctx->cam.principal_point_x;
    sc.cv = ctx->cam.principal_point_y;
    const dim3 gp((unsigned)((n + 255) / 256)), bp(256);
    const int G = (int)((n + kWave - 1) / kWave);
    GroupSums* gs = reinterpret_cast<GroupSums*>(ctx->sem_groups);
    unsigned long long* inl = reinterpret_cast<unsigned long long*>(s.mask_buf);
    // candidates by label, first fit
    hipLaunchKernelGGL(k_sem_candidates, gp, bp, 0, st, s.d.cloud, n, s.d.stride, sc, img_dev, rows, cols, row_stride, ls, gs);
    hipLaunchKernelGGL(k_sem_fit, dim3(1), dim3(kPartials), 0, st, gs, G, ctx->sem_coeffs, ctx->sem_coeffs + 4, 3, 0,
                       ctx->sem_res, (PlaneDev*)nullptr, (PlaneDev*)nullptr, 0.f, 0.f, 0.f);
    // re-selection over the whole cloud (its ballots are the inlier mask), second fit
    hipLaunchKernelGGL(k_sem_select, gp, bp, 0, st, s.d.cloud, n, s.d.stride, ctx->sem_coeffs + 4, inlier_threshold, inl, gs);
    hipLaunchKernelGGL(k_sem_fit, dim3(1), dim3(kPartials), 0, st, gs, G, ctx->sem_coeffs + 4, ctx->sem_coeffs + 8, 0, 1,
                       ctx->sem_res, pd, pd_copy, ctx->calib.far_elin, ctx->calib.far_econst, ctx->calib.roadDistThrF);
    HIP_TRY(ctx, hipGetLastError());
    return MLD_OK;
}

// scratch of the semantic estimator (allocated on first use; may synchronise then)
static int semantic_plane_scratch(mld_ctx* ctx, Slot& s, long long n) {
    using namespace ransac;
    int rc = MLD_OK;
    if (!ctx->sem_res) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->sem_coeffs, 12 * sizeof(float)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->sem_res, sizeof(SemResult)));
        const float init[12] = {0.f, 0.f, 1.f, 0.f, 0, 0, 0, 0, 0, 0, 0, 0};  // dummy prior (:241-242)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->sem_coeffs, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    // per 64 points: the group's moment sums (40 bytes) and one 64-bit word of the inlier mask
    const size_t groups = (size_t)((n + 63) / 64);
    if ((rc = grow(ctx, ctx->sem_groups, ctx->sem_groups_cap, groups * sizeof(GroupSums)))) return rc;
    return grow(ctx, s.mask_buf, s.mask_words, groups * 2);
}

static int semantic_plane_core(mld_ctx* ctx, Slot& s, const unsigned char* img_dev, int rows, int cols, int row_stride,
                               const int32_t* labels, int n_labels, double inlier_threshold, float coeffs_out[4],
                               int64_t* n_inliers_out) {
    using namespace ransac;
    int rc = MLD_OK;
    if (s.d.n < 3) return fail(ctx, MLD_ERR_CLOUD_TOO_SMALL, "In GroundPlane: Input pointcloud is invalid");  // (:224-227)
    if ((rc = semantic_plane_scratch(ctx, s, s.d.n))) return rc;
    if ((rc = semantic_plane_launch(ctx, s, img_dev, rows, cols, row_stride, labels, n_labels, inlier_threshold, ctx->stream)))
        return rc;
    SemResult res;
    HIP_TRY(ctx, hipMemcpyAsync(&res, ctx->sem_res, sizeof(SemResult), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (res.status != 0) {  // :224-227
        clear_plane(s);  // the slot's mask buffer has been overwritten
        s.plane_decided = false;
        return fail(ctx, MLD_ERR_CLOUD_TOO_SMALL, "In GroundPlane: Input pointcloud is invalid");
    }
    set_plane_coeffs(ctx, s, res.coeffs);
    s.d.inlier_mask = s.mask_buf;
    s.d.mask_in_key = 0;
    if (coeffs_out)
        for (int t = 0; t < 4; t++) coeffs_out[t] = res.coeffs[t];
    if (n_inliers_out) *n_inliers_out = res.n_inliers;
    return MLD_OK;
}

static int semantic_plane_args(mld_ctx* ctx, int slot, const void* img, int rows, int cols, int row_stride,
                               const int32_t* labels, int n_labels) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "ground plane estimation before the slot's cloud");
    if (!img || rows <= 0 || cols <= 0 || row_stride < cols || n_labels < 0 || (n_labels > 0 && !labels))
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad label image / label set");
    return MLD_OK;
}

int mld_estimate_semantic_plane_device(mld_ctx* ctx, int slot, const uint8_t* label_image_dev, int rows, int cols,
                                       int row_stride_bytes, const int32_t* ground_labels, int n_labels,
                                       double inlier_threshold, float coeffs_out[4], int64_t* n_inliers_out) {
    int rc = semantic_plane_args(ctx, slot, label_image_dev, rows, cols, row_stride_bytes, ground_labels, n_labels);
    if (rc) return rc;
    return semantic_plane_core(ctx, ctx->slots[slot], label_image_dev, rows, cols, row_stride_bytes, ground_labels,
                               n_labels, inlier_threshold, coeffs_out, n_inliers_out);
}

int mld_estimate_semantic_plane(mld_ctx* ctx, int slot, const uint8_t* label_image_host, int rows, int cols,
                                int row_stride_bytes, const int32_t* ground_labels, int n_labels, double inlier_threshold,
                                float coeffs_out[4], int64_t* n_inliers_out) {
    int rc = semantic_plane_args(ctx, slot, label_image_host, rows, cols, row_stride_bytes, ground_labels, n_labels);
    if (rc) return rc;
    const size_t bytes = (size_t)rows * (size_t)row_stride_bytes;
    if ((rc = grow(ctx, ctx->sem_img, ctx->sem_img_cap, bytes))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->sem_img, label_image_host, bytes, hipMemcpyHostToDevice, ctx->stream));
    return semantic_plane_core(ctx, ctx->slots[slot], ctx->sem_img, rows, cols, row_stride_bytes, ground_labels, n_labels,
                               inlier_threshold, coeffs_out, n_inliers_out);
}

int mld_get_ground_plane_inliers(mld_ctx* ctx, int slot, int32_t* index_out, int64_t capacity, int64_t* n_out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if (!s.cloud_set || !s.d.has_plane || !s.d.inlier_mask) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "no ground plane set");
    const size_t words = (size_t)((s.d.n + 31) / 32);
    std::vector<uint32_t> m(words);
    HIP_TRY(ctx, hipMemcpyAsync(m.data(), s.d.inlier_mask, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int64_t k = 0;
    for (long long i = 0; i < s.d.n; i++)
        if ((m[(size_t)i >> 5] >> (i & 31)) & 1u) {
            if (index_out && k < capacity) index_out[k] = (int32_t)i;
            k++;
        }
    if (n_out) *n_out = k;
    if (index_out && k > capacity) return fail(ctx, MLD_ERR_CAPACITY, "output buffer too small");
    return MLD_OK;
}

int mld_set_ground_plane_mask_device(mld_ctx* ctx, int slot, const float coeffs[4], const uint32_t* mask_dev) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    Slot& s = ctx->slots[slot];
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "ground plane set before the slot's cloud");
    if (!coeffs) {
        clear_plane(s);
        return MLD_OK;
    }
    if (!mask_dev) return fail(ctx, MLD_ERR_INVALID_ARG, "null mask");
    set_plane_coeffs(ctx, s, coeffs);
    s.d.inlier_mask = mask_dev;
    s.d.mask_in_key = 0;
    return MLD_OK;
}

int mld_set_ground_planes_mask_device(mld_ctx* ctx, int n_slots, const float* coeffs, const uint32_t* const* mask_dev) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (n_slots < 1 || n_slots > (int)ctx->slots.size() || !coeffs || !mask_dev)
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad slot count / null arrays");
    for (int i = 0; i < n_slots; i++) {
        int rc = mld_set_ground_plane_mask_device(ctx, i, coeffs + 4 * i, mask_dev[i]);
        if (rc) return rc;
    }
    return MLD_OK;
}

// ---------------------------------------------------------------------------- CalculateDepth
static int calc_one(mld_ctx* ctx, int slot, const double* uv_dev, int64_t F, double* depth_dev, int32_t* type_dev,
                    double* corners_dev = nullptr, bool skip_road = false) {
    Slot& s = ctx->slots[slot];
    s.d.uv = uv_dev;
    s.d.F = F;
    s.d.depth = depth_dev;
    s.d.type = type_dev;
    if (F == 0) return MLD_OK;
    {
        int rcq = ensure_queues(ctx, s, F);
        if (rcq) return rcq;
    }
    if (ctx->P.set_all_depths_to_zero) {
        hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, ctx->stream, depth_dev,
                           type_dev, (long long)F);
        HIP_TRY(ctx, hipGetLastError());
        return MLD_OK;
    }
    if (corners_dev) {
        // debug mode: every feature takes the wave-cooperative kernel, which also stores the triangle corners
        Calib dbg = ctx->calib;
        dbg.threadPath = 0;
        s.d.corners = corners_dev;
        int rc = launch_features(ctx, 1, F, true, slot, &dbg);
        s.d.corners = nullptr;
        return rc;
    }
    if (skip_road) {  // "ransacPlane == nullptr" for this call only (DepthEstimator.cpp:580)
        Calib nr = ctx->calib;
        nr.useRoad = 0;
        return launch_features(ctx, 1, F, true, slot, &nr);
    }
    return launch_features(ctx, 1, F, true, slot);
}

int mld_calculate_depth_device(mld_ctx* ctx, int slot, const double* uv_dev, int64_t F, double* depth_out_dev,
                               int32_t* type_out_dev) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    if ((rc = precheck_calc(ctx, ctx->slots[slot], F))) return rc;
    if (F > 0 && (!uv_dev || !depth_out_dev)) return fail(ctx, MLD_ERR_INVALID_ARG, "null feature/output pointer");
    return calc_one(ctx, slot, uv_dev, F, depth_out_dev, type_out_dev);
}

int mld_calculate_depth(mld_ctx* ctx, int slot, const double* uv_host, int64_t F, double* depth_out_host,
                        int32_t* type_out_host) {
    return mld_calculate_depth_opts(ctx, slot, uv_host, F, depth_out_host, type_out_host, 0u);
}

int mld_calculate_depth_opts(mld_ctx* ctx, int slot, const double* uv_host, int64_t F, double* depth_out_host,
                             int32_t* type_out_host, uint32_t flags) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    const bool skip_road = (flags & MLD_CALC_SKIP_ROAD) != 0;
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "call of 'CalculateDepth' without 'SetInputCloud'");
    if (!skip_road && (rc = precheck_calc(ctx, s, F))) return rc;
    if (F < 0 || F > 0x7FFFFFFFLL) return fail(ctx, MLD_ERR_INVALID_ARG, "bad feature count");
    if (F == 0) return MLD_OK;
    if (!uv_host || !depth_out_host) return fail(ctx, MLD_ERR_INVALID_ARG, "null feature/output pointer");
    if ((size_t)F > s.feat_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        void* olds[] = {s.uv_buf, s.depth_buf, s.type_buf};
        for (void* p : olds)
            if (p) HIP_TRY(ctx, hipFree(p));
        s.uv_buf = nullptr;
        s.depth_buf = nullptr;
        s.type_buf = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&s.uv_buf, (size_t)F * 2 * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void**)&s.depth_buf, (size_t)F * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void**)&s.type_buf, (size_t)F * sizeof(int32_t)));
        s.feat_cap = (size_t)F;
    }
    HIP_TRY(ctx, hipMemcpyAsync(s.uv_buf, uv_host, (size_t)F * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if ((rc = calc_one(ctx, slot, s.uv_buf, F, s.depth_buf, s.type_buf, nullptr, skip_road))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(depth_out_host, s.depth_buf, (size_t)F * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    if (type_out_host)
        HIP_TRY(ctx, hipMemcpyAsync(type_out_host, s.type_buf, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToHost,
                                    ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MLD_OK;
}

// setInputCloud + CalculateDepth of ONE frame from host memory in a single call (the reference's
// CalculateDepth(cloud, uv, depths, types, groundPlane), DepthEstimator.cpp:404-420): the small inputs travel in one
// pinned block and one DMA on a side stream, the results come back in one, and the plane - supplied, or ESTIMATED on the
// slot when the GroundPlane handed in is not segmented yet (:275-283) - is in place BEFORE the projection, so that the
// points' ground-plane state rides in the map keys.  One asynchronous chain, one synchronisation at the end.
namespace {
struct FramePlane {
    enum Kind { NONE, SUPPLIED, RANSAC, SEMANTIC } kind = NONE;
    const float* coeffs = nullptr;  // SUPPLIED
    const int32_t* inliers = nullptr;
    int64_t n_inliers = 0;
    const mld_plane_request* req = nullptr;  // RANSAC / SEMANTIC
};
// The feature side of a one-frame call given as tracklets (TrackletDepthModule::process): gather -> depths of the newest
// features on the frame being uploaded, of the NEW tracks' previous features on the resident previous frame -> scatter.
struct FrameTracks {
    int slot_last = -1;
    const float *u_new = nullptr, *v_new = nullptr, *u_old = nullptr, *v_old = nullptr;
    const uint8_t* is_new = nullptr;
    int64_t n = 0;
    float *d_cur_out = nullptr, *d_last_out = nullptr;
    int32_t *type_cur_out = nullptr, *type_last_out = nullptr;
    int64_t* n_new_out = nullptr;
};
double now_us() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}
}  // namespace

static int ensure_tracklet_scratch(mld_ctx* ctx, size_t n);

static int frame_call(mld_ctx* ctx, int slot, const void* pts_host, int64_t n, int stride_bytes, const FramePlane& fp,
                      const double* uv_host, int64_t F, double* depth_out_host, int32_t* type_out_host,
                      mld_plane_result* plane_out, const FrameTracks* trk = nullptr) {
    using namespace ransac;
    const double t_entry = now_us();
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    const int64_t nt = trk ? trk->n : 0;
    if (trk) {
        if (trk->slot_last >= 0 && (rc = check_slot(ctx, trk->slot_last))) return rc;
        if (trk->slot_last == slot) return fail(ctx, MLD_ERR_INVALID_ARG, "slot_last must differ from slot_cur");
        if (nt < 0 || nt > 0x7FFFFFFFLL) return fail(ctx, MLD_ERR_INVALID_ARG, "bad track count");
        if (nt > 0 && (!trk->u_new || !trk->v_new || !trk->u_old || !trk->v_old || !trk->is_new || !trk->d_cur_out || !trk->d_last_out))
            return fail(ctx, MLD_ERR_INVALID_ARG, "null tracklet array");
        if (trk->slot_last >= 0 && (rc = precheck_calc(ctx, ctx->slots[trk->slot_last], nt))) return rc;
        if (ctx->P.set_all_depths_to_zero) return fail(ctx, MLD_ERR_UNSUPPORTED_MODE, "set_all_depths_to_zero: use mld_tracklets_depth");
    }
    if (n < 0 || (!pts_host && n > 0)) return fail(ctx, MLD_ERR_INVALID_ARG, "bad cloud");
    if (stride_bytes != 16 && stride_bytes != 32) return fail(ctx, MLD_ERR_INVALID_ARG, "stride_bytes must be 16 or 32");
    if (n > kMaxPoints) return fail(ctx, MLD_ERR_CAPACITY, "cloud larger than 8 388 607 points");
    if (F < 0 || F > 0x7FFFFFFFLL) return fail(ctx, MLD_ERR_INVALID_ARG, "bad feature count");
    if (F > 0 && (!uv_host || !depth_out_host)) return fail(ctx, MLD_ERR_INVALID_ARG, "null feature/output pointer");
    const mld_params& P = ctx->P;
    FramePlane::Kind kind = fp.kind;
    if (!P.do_use_ransac_plane && kind != FramePlane::NONE) kind = FramePlane::NONE;  // the plane is ignored (:274)
    if (kind == FramePlane::SUPPLIED && (fp.n_inliers < 0 || (!fp.inliers && fp.n_inliers > 0)))
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad inlier list");
    // RansacPlane.cpp:44-50 / :224-227: fewer than three points cannot carry a plane - the reference throws out of
    // setInputCloud before anything else happens.  With tracklets the previous frame's features are still answered
    // (tracklet_depth_module.cpp:318-347): the frame runs without a plane and the call reports the failure at the end.
    const bool too_small = (kind == FramePlane::RANSAC || kind == FramePlane::SEMANTIC) && n < 3;
    if (too_small && !trk) return fail(ctx, MLD_ERR_CLOUD_TOO_SMALL, "In GroundPlane: Input pointcloud is invalid");
    if (too_small) kind = FramePlane::NONE;
    const bool estimate = kind == FramePlane::RANSAC || kind == FramePlane::SEMANTIC;
    const mld_plane_request* rq = fp.req;
    if (kind == FramePlane::SEMANTIC &&
        (!rq->label_image || rq->rows <= 0 || rq->cols <= 0 || rq->row_stride_bytes < rq->cols || rq->n_labels < 0 ||
         (rq->n_labels > 0 && !rq->ground_labels)))
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad label image / label set");
    const int n_draws = P.ransac_plane_max_iterations + 1;
    if (kind == FramePlane::RANSAC && n_draws < 1) return fail(ctx, MLD_ERR_INVALID_ARG, "ransac_plane_max_iterations must be >= 0");
    Slot& s = ctx->slots[slot];
    // RANSAC with the z pass-through on a cloud too large for the one-block kernel's LDS: the per-slot estimator
    // (it synchronises once before the feature kernels)
    const bool pass = P.ransac_plane_min_z > -1001.;
    size_t rs_lds = 0;
    if (kind == FramePlane::RANSAC) {
        const size_t lds_fixed = (size_t)kSample * 5 * sizeof(float) + (size_t)kPartials * 9 * sizeof(float) +
                                 (2 * kRsRound + kRsMisc + kRsEpochInts) * sizeof(int);
        const size_t n_chunks = (size_t)((n + 1023) / 1024);
        rs_lds = lds_fixed + (pass ? (n_chunks + 1) * sizeof(int) + 16 * n_chunks + 16 : 0);
        if (rs_lds > 158 * 1024) {
            float co[4] = {0.f, 0.f, 0.f, 0.f};
            int64_t ni = 0;
            if ((rc = mld_set_cloud(ctx, slot, pts_host, n, stride_bytes))) return rc;
            const int rc_est = mld_estimate_ground_plane(ctx, slot, rq->seed, co, &ni);
            if (rc_est && (rc_est != MLD_ERR_CLOUD_TOO_SMALL || !trk)) return rc_est;
            if (plane_out) {
                for (int t = 0; t < 4; t++) plane_out->coeffs[t] = rc_est ? 0.f : co[t];
                plane_out->n_inliers = rc_est ? 0 : ni;
                plane_out->status = rc_est ? 1 : 0;
                plane_out->iterations = 0;
            }
            if (trk) {
                // the tracklet side of the call on the same route: both CalculateDepth calls + scatter.  A failed
                // estimation (GroundPlane::ExceptionPclInvalid; the estimator has cleared the slot's plane) still answers
                // the previous frame's features and invalidates the current ones (tracklet_depth_module.cpp:318-347),
                // as the one-chain path below does.
                const std::string why = ctx->err;
                if (rc_est) clear_plane(s);  // this frame runs without a plane ("decided": the feature calls accept it)
                struct Undecide {  // ... and the slot forgets it afterwards, as the reference forgets cloud and plane
                    Slot& s;
                    bool on;
                    ~Undecide() {
                        if (on) s.plane_decided = false;
                    }
                } undecide{s, rc_est != 0};
                if ((rc = mld_tracklets_depth(ctx, slot, trk->slot_last, trk->u_new, trk->v_new, trk->u_old, trk->v_old,
                                              trk->is_new, nt, trk->d_cur_out, trk->d_last_out, trk->type_cur_out,
                                              trk->type_last_out, trk->n_new_out)))
                    return rc;
                if (rc_est) {
                    for (int64_t i = 0; i < nt; i++) {
                        trk->d_cur_out[i] = -1.0f;
                        if (trk->type_cur_out) trk->type_cur_out[i] = (int32_t)MLD_Unspecified;
                    }
                    return fail(ctx, rc_est, why);
                }
                return MLD_OK;
            }
            return F > 0 ? mld_calculate_depth(ctx, slot, uv_host, F, depth_out_host, type_out_host) : MLD_OK;
        }
    }
    const size_t n_inl = kind == FramePlane::SUPPLIED ? (size_t)fp.n_inliers : 0;
    // staging block: [inlier list | inputs] travel to the device in one DMA, [outputs | plane] come back
    //   features: inputs uv (2 F doubles), outputs depth (F doubles) + type (F ints)
    //   tracklets: inputs u_new, v_new, u_old, v_old (floats) + is_new (bytes), outputs d_cur, d_last (floats) + type_cur,
    //              type_last (ints) + the number of new tracks
    const size_t off_uv = (n_inl * sizeof(int32_t) + 15) & ~(size_t)15;
    const size_t in_bytes = trk ? (((size_t)nt * 17 + 15) & ~(size_t)15) : (size_t)F * 2 * sizeof(double);
    const size_t off_depth = off_uv + in_bytes;
    const size_t off_type = off_depth + (size_t)F * sizeof(double);  // (features only)
    const size_t out_bytes = trk ? (size_t)nt * 16 + 16 : (size_t)F * (sizeof(double) + sizeof(int32_t));
    const size_t off_plane = (off_depth + out_bytes + 15) & ~(size_t)15;
    const size_t total = off_plane + sizeof(PlaneDev) + 16;
    if (total > ctx->fr_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->fr_host) HIP_TRY(ctx, hipHostFree(ctx->fr_host));
        if (ctx->fr_dev) HIP_TRY(ctx, hipFree(ctx->fr_dev));
        ctx->fr_host = nullptr;
        ctx->fr_dev = nullptr;
        ctx->fr_cap = 0;
        const size_t cap = total + total / 2;
        HIP_TRY(ctx, hipHostMalloc((void**)&ctx->fr_host, cap, hipHostMallocDefault));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->fr_dev, cap));
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&ctx->fr_host_dev, ctx->fr_host, 0));
        ctx->fr_cap = cap;
    }
    // Results (depths, types, the estimated plane: 24 KB for 2000 features) are written by the kernels straight into the
    // pinned host block: no copy node between the last kernel and the synchronisation (a 24 KB DMA is 15 us of latency)
    unsigned char* const out_base = ctx->fr_zero_copy ? ctx->fr_host_dev : ctx->fr_dev;
    const size_t bytes = (size_t)n * (size_t)stride_bytes;
    if ((rc = grow(ctx, s.cloud_buf, s.cloud_cap, bytes))) return rc;
    // inlier bitmask of the slot; the one-block RANSAC also keeps its 64-bit pass-through group masks there (even word count)
    const size_t words = kind == FramePlane::RANSAC ? (((size_t)((n + 31) / 32) + 2) & ~(size_t)1) : (size_t)((n + 31) / 32);
    if (kind != FramePlane::NONE && (rc = grow(ctx, s.mask_buf, s.mask_words, words))) return rc;
    if ((rc = ensure_queues(ctx, s, trk ? nt : F))) return rc;
    if (trk) {
        if ((rc = ensure_tracklet_scratch(ctx, (size_t)nt))) return rc;
        if (trk->slot_last >= 0 && (rc = ensure_queues(ctx, ctx->slots[trk->slot_last], nt))) return rc;
    }
    if (!ctx->side) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->side, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_done, hipEventDisableTiming));
    }
    if (!ctx->side_start) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->side_start, hipEventDisableTiming));
    if (!ctx->helper && ctx->fr_helper_on) {
        ctx->helper = new FrameHelper();
        ctx->helper->th = std::thread([h = ctx->helper] { h->run(); });
    }
    if (estimate && !ctx->rsb_planes) {
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rsb_planes, ctx->slots.size() * sizeof(PlaneDev)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->rsb_seeds, ctx->slots.size() * sizeof(uint32_t)));
    }
    size_t img_bytes = 0;
    if (kind == FramePlane::SEMANTIC) {
        if ((rc = semantic_plane_scratch(ctx, s, n))) return rc;  // (sized by the cloud that is about to arrive)
        img_bytes = (size_t)rq->rows * (size_t)rq->row_stride_bytes;
        if ((rc = grow(ctx, ctx->sem_img, ctx->sem_img_cap, img_bytes))) return rc;
    }
    if (kind == FramePlane::RANSAC && (rc = enable_rs_batch_lds(ctx, rs_lds))) return rc;
    const bool timed = ctx->timing;
    if (timed && !ctx->fr_ev[0])
        for (int i = 0; i < 5; i++) HIP_TRY(ctx, hipEventCreate(&ctx->fr_ev[i]));
    // The small inputs travel on the side stream: their DMA, the mask build and the bitmap clear run while the cloud is
    // in flight.  (The side stream starts after whatever is already queued on the context's stream: an earlier
    // asynchronous call on this slot may still read the mask buffer and the staging block that are rewritten here.)
    // Staging and queueing them is the helper thread's job, so that THIS thread can submit the cloud copy at once.
    HIP_TRY(ctx, hipEventRecord(ctx->side_start, ctx->stream));
    const int device = ctx->device;
    unsigned char* const fr_host = ctx->fr_host;
    unsigned char* const fr_dev = ctx->fr_dev;
    unsigned char* const sem_img = ctx->sem_img;
    uint32_t* const mask_buf = s.mask_buf;
    uint32_t* const bitmap = s.d.bitmap;
    const size_t bitmap_bytes = ctx->bitmap_words * sizeof(uint32_t);
    hipStream_t side = ctx->side;
    hipEvent_t ev_start = ctx->side_start, ev_done = ctx->side_done;
    const int32_t* const inl_src = fp.inliers;
    const uint8_t* const img_src = kind == FramePlane::SEMANTIC ? rq->label_image : nullptr;
    const bool clear_mask = kind == FramePlane::SUPPLIED || kind == FramePlane::RANSAC;
    const FrameTracks tk = trk ? *trk : FrameTracks{};
    double* const trk_uv_cur = ctx->trk_uv_cur;
    double* const trk_uv_last = ctx->trk_uv_last;
    int32_t* const trk_rank = ctx->trk_rank;
    long long* const trk_n_new = ctx->trk_n_new;
    auto side_work = [=](std::string& err) -> int {
        auto bad = [&](hipError_t e, const char* what) {
            err = std::string(what) + ": " + hipGetErrorString(e);
            return (int)MLD_ERR_HIP;
        };
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return bad(e, "hipSetDevice");
        if ((e = hipStreamWaitEvent(side, ev_start, 0)) != hipSuccess) return bad(e, "hipStreamWaitEvent(side)");
        if (n_inl) std::memcpy(fr_host, inl_src, n_inl * sizeof(int32_t));
        if (F) std::memcpy(fr_host + off_uv, uv_host, (size_t)F * 2 * sizeof(double));
        if (nt) {
            unsigned char* q = fr_host + off_uv;
            const float* src[4] = {tk.u_new, tk.v_new, tk.u_old, tk.v_old};
            for (int a = 0; a < 4; a++) std::memcpy(q + (size_t)a * (size_t)nt * 4, src[a], (size_t)nt * 4);
            std::memcpy(q + (size_t)nt * 16, tk.is_new, (size_t)nt);
        }
        if (off_depth && (e = hipMemcpyAsync(fr_dev, fr_host, off_depth, hipMemcpyHostToDevice, side)) != hipSuccess)
            return bad(e, "hipMemcpyAsync(features)");
        if (nt) {  // ExractNewTrackletFrames + the features' truncation to integer pixels, off the critical path
            const float* in = reinterpret_cast<const float*>(fr_dev + off_uv);
            hipLaunchKernelGGL(k_tracklet_gather, dim3(1), dim3(kTrkBlock), 0, side, in, in + nt, in + 2 * nt, in + 3 * nt,
                               reinterpret_cast<const uint8_t*>(in + 4 * nt), (long long)nt, trk_uv_cur, trk_uv_last, trk_rank,
                               trk_n_new);
            if ((e = hipGetLastError()) != hipSuccess) return bad(e, "k_tracklet_gather");
        }
        if (img_src && (e = hipMemcpyAsync(sem_img, img_src, img_bytes, hipMemcpyHostToDevice, side)) != hipSuccess)
            return bad(e, "hipMemcpyAsync(label image)");
        if (clear_mask) {
            if ((e = hipMemsetAsync(mask_buf, 0, (words < 1 ? 1 : words) * sizeof(uint32_t), side)) != hipSuccess)
                return bad(e, "hipMemsetAsync(mask)");
            if (n_inl) {
                hipLaunchKernelGGL(k_build_mask, dim3((unsigned)((n_inl + 255) / 256)), dim3(256), 0, side,
                                   reinterpret_cast<const int32_t*>(fr_dev), (long long)n_inl, (long long)n, mask_buf);
                if ((e = hipGetLastError()) != hipSuccess) return bad(e, "k_build_mask");
            }
        }
        // (the slot's occupancy bitmap is cleared there as well: off the cloud copy's critical path)
        if ((e = hipMemsetAsync(bitmap, 0, bitmap_bytes, side)) != hipSuccess) return bad(e, "hipMemsetAsync(bitmap)");
        if ((e = hipEventRecord(ev_done, side)) != hipSuccess) return bad(e, "hipEventRecord(side)");
        return (int)MLD_OK;
    };
    // (never return while the helper still reads the caller's arrays)
    struct HelperGuard {
        FrameHelper* h;
        ~HelperGuard() {
            std::string ignored;
            if (h) (void)h->wait(ignored);
        }
    } helper_guard{ctx->helper};
    int side_rc = MLD_OK;
    std::string side_err;
    if (ctx->helper)
        ctx->helper->submit(side_work);
    else
        side_rc = side_work(side_err);  // (no helper thread: the caller queues the side-stream work itself, ahead of the copy)
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->fr_ev[0], ctx->stream));
    const double t_copy0 = now_us();
    // The cloud, straight from the caller's memory (measured: the runtime's own staging of a pageable source moves
    // 2.1 MB in 51 us, as fast as from pinned memory; copying through a pinned buffer of ours in pieces was slower).
    hipError_t e_copy = hipSuccess;
    if (bytes) e_copy = hipMemcpyAsync(s.cloud_buf, pts_host, bytes, hipMemcpyHostToDevice, ctx->stream);
    const double t_copy1 = now_us();
    if (ctx->helper) side_rc = ctx->helper->wait(side_err);
    HIP_TRY(ctx, e_copy);
    if (side_rc) return fail(ctx, side_rc, side_err);
    if ((rc = begin_cloud(ctx, s, s.cloud_buf, n, stride_bytes, false))) return rc;  // (bitmap cleared on the side stream)
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->fr_ev[1], ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done, 0));
    PlaneDev* pd_copy = reinterpret_cast<PlaneDev*>(out_base + off_plane);
    if (kind == FramePlane::SUPPLIED) {
        set_plane_coeffs(ctx, s, fp.coeffs);
        s.d.inlier_mask = s.mask_buf;
        s.d.mask_in_key = 1;
    } else if (kind == FramePlane::NONE) {
        clear_plane(s);
    } else {
        // the plane is estimated on the slot and stays in device memory (PlaneDev): no host round trip before the
        // projection, which reads it there for the points' ground-plane state
        PlaneDev* pd = ctx->rsb_planes + slot;
        s.d.inlier_mask = s.mask_buf;
        if (kind == FramePlane::RANSAC) {
            hipLaunchKernelGGL(k_rs_batch, dim3(1), dim3(kRsThreads), rs_lds, ctx->stream, ctx->d_slots, ctx->rsb_seeds, n_draws,
                               P.ransac_plane_max_iterations, P.ransac_plane_probability, P.ransac_plane_distance_treshold,
                               P.ransac_plane_refinement_treshold, P.ransac_plane_use_refinement, ctx->rsb_planes, pass ? 1 : 0,
                               (float)P.ransac_plane_min_z, (float)P.ransac_plane_max_z, ctx->calib.far_elin,
                               ctx->calib.far_econst, ctx->calib.roadDistThrF, slot, s.d, rq->seed, pd_copy);
            HIP_TRY(ctx, hipGetLastError());
        } else {
            if ((rc = semantic_plane_launch(ctx, s, ctx->sem_img, rq->rows, rq->cols, rq->row_stride_bytes, rq->ground_labels,
                                            rq->n_labels, rq->inlier_threshold, ctx->stream, pd, pd_copy)))
                return rc;
        }
        s.d.mask_in_key = 1;
        s.d.plane_dev = pd;
        s.d.has_plane = 1;  // the device copy decides (PlaneDev::has_plane)
        s.plane_decided = true;
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(ctx->fr_ev[2], ctx->stream));
    if ((rc = launch_project(ctx, 1, n, true, slot))) return rc;
    double* d_depth = reinterpret_cast<double*>(out_base + off_depth);
    int32_t* d_type = reinterpret_cast<int32_t*>(out_base + off_type);
    if (F > 0) rc = calc_one(ctx, slot, reinterpret_cast<const double*>(ctx->fr_dev + off_uv), F, d_depth, d_type, nullptr, false);
    if (trk && nt > 0) {
        // TrackletDepthModule::process (tracklet_depth_module.cpp:23-169) on the frame just projected: marshal the
        // features, both CalculateDepth calls (the previous frame from its resident slot), float32 scatter
        float* o_cur = reinterpret_cast<float*>(out_base + off_depth);
        float* o_last = o_cur + nt;
        int32_t* o_tc = reinterpret_cast<int32_t*>(o_last + nt);
        int32_t* o_tl = o_tc + nt;
        long long* o_nn = reinterpret_cast<long long*>(o_tl + nt);
        const int slot_last = trk->slot_last;
        // (the feature marshalling - k_tracklet_gather - ran on the side stream, beside the cloud copy)
        rc = calc_one(ctx, slot, ctx->trk_uv_cur, nt, ctx->trk_depth_cur, ctx->trk_type_cur, nullptr, false);
        if (rc == MLD_OK && slot_last >= 0) {
            Slot& sl = ctx->slots[slot_last];
            sl.d.F_dev = ctx->trk_n_new;  // the number of new tracks is only known on the device
            rc = calc_one(ctx, slot_last, ctx->trk_uv_last, nt, ctx->trk_depth_last, ctx->trk_type_last, nullptr, false);
            sl.d.F_dev = nullptr;
        }
        if (rc == MLD_OK) {
            hipLaunchKernelGGL(k_tracklet_scatter, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, ctx->stream,
                               ctx->trk_depth_cur, ctx->trk_type_cur, ctx->trk_depth_last, ctx->trk_type_last, ctx->trk_rank,
                               (long long)nt, slot_last >= 0 ? 1 : 0, o_cur, o_last, o_tc, o_tl,
                               (const long long*)ctx->trk_n_new, o_nn);
            if (hipGetLastError() != hipSuccess) rc = fail(ctx, MLD_ERR_HIP, "k_tracklet_scatter");
        }
    }
    if (rc == MLD_OK) {
        hipError_t e = hipSuccess;
        if (timed) e = hipEventRecord(ctx->fr_ev[3], ctx->stream);
        const size_t back_bytes = (estimate ? off_plane + sizeof(PlaneDev) : off_depth + out_bytes) - off_depth;
        if (e == hipSuccess && back_bytes && !ctx->fr_zero_copy)
            e = hipMemcpyAsync(ctx->fr_host + off_depth, d_depth, back_bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && timed) e = hipEventRecord(ctx->fr_ev[4], ctx->stream);
        const double t_enq = now_us();
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        const double t_done = now_us();
        if (e != hipSuccess) {
            ctx->err = std::string("mld_calculate_depth_frame: ") + hipGetErrorString(e);
            rc = MLD_ERR_HIP;
        } else {
            // host clock: always (a few clock reads); GPU phases only when the events were recorded
            ctx->fr_us[4] = t_enq - t_entry;
            ctx->fr_us[5] = t_done - t_enq;
            ctx->fr_us[6] = t_done - t_entry;  // (up to here: the copies into the caller's arrays follow)
            ctx->fr_us[8] = t_copy0 - t_entry;
            ctx->fr_us[9] = t_copy1 - t_copy0;
            for (int i = 0; i < 4; i++) ctx->fr_us[i] = 0.0;
            ctx->fr_us[7] = 0.0;
            if (timed) {
                float ms[4] = {0, 0, 0, 0}, all = 0;
                for (int i = 0; i < 4; i++) (void)hipEventElapsedTime(&ms[i], ctx->fr_ev[i], ctx->fr_ev[i + 1]);
                (void)hipEventElapsedTime(&all, ctx->fr_ev[0], ctx->fr_ev[4]);
                for (int i = 0; i < 4; i++) ctx->fr_us[i] = 1e3 * (double)ms[i];
                ctx->fr_us[7] = 1e3 * (double)all;
            }
        }
    }
    // the slot must not keep pointers into the staging block (it is reused by the next call)
    s.d.uv = nullptr;
    s.d.depth = nullptr;
    s.d.type = nullptr;
    s.d.F = 0;
    if (rc) return rc;
    // tracklets: float32 depths back to the tracks (d_last / type_last are defined for the new tracks only)
    auto deliver_tracks = [&](bool cur_valid) {
        if (!trk || nt == 0) {
            if (trk && trk->n_new_out) *trk->n_new_out = 0;
            return;
        }
        const float* o_cur = reinterpret_cast<const float*>(ctx->fr_host + off_depth);
        const float* o_last = o_cur + nt;
        const int32_t* o_tc = reinterpret_cast<const int32_t*>(o_last + nt);
        const int32_t* o_tl = o_tc + nt;
        int64_t nn = 0;
        for (int64_t i = 0; i < nt; i++) {
            trk->d_cur_out[i] = cur_valid ? o_cur[i] : -1.0f;
            if (trk->type_cur_out) trk->type_cur_out[i] = cur_valid ? o_tc[i] : (int32_t)MLD_Unspecified;
            if (trk->is_new[i]) {
                trk->d_last_out[i] = o_last[i];
                if (trk->type_last_out) trk->type_last_out[i] = o_tl[i];
                nn++;
            }
        }
        if (trk->n_new_out) *trk->n_new_out = nn;
    };
    if (estimate) {
        PlaneDev h;
        std::memcpy(&h, ctx->fr_host + off_plane, sizeof(PlaneDev));
        if (plane_out) {
            for (int t = 0; t < 4; t++) plane_out->coeffs[t] = h.status == 0 ? h.coeffs[t] : 0.f;
            plane_out->n_inliers = h.status == 0 ? h.n_inliers : 0;
            plane_out->status = h.status;
            plane_out->iterations = h.iterations;
        }
        if (h.status != 0) {  // GroundPlane::ExceptionPclInvalid: the reference never gets to the feature loop
            clear_plane(s);
            s.plane_decided = false;
            // (tracklets: the previous frame's features are still answered, the current ones are invalid -
            // tracklet_depth_module.cpp:320-347)
            deliver_tracks(false);
            return fail(ctx, MLD_ERR_CLOUD_TOO_SMALL, "In GroundPlane: Input pointcloud is invalid");
        }
        // host-side copy of what the kernels read from the slot's PlaneDev (for the getters)
        std::memcpy(s.d.coeffs, h.coeffs, sizeof(float) * 4);
        for (int t = 0; t < 3; t++) s.d.prior_n[t] = h.prior_n[t];
        s.d.prior_off = h.prior_off;
        s.d.far_mg0 = h.far_mg0;
        s.d.far_mg1 = h.far_mg1;
    }
    if (F > 0) {
        std::memcpy(depth_out_host, ctx->fr_host + off_depth, (size_t)F * sizeof(double));
        if (type_out_host) std::memcpy(type_out_host, ctx->fr_host + off_type, (size_t)F * sizeof(int32_t));
    }
    if (too_small) {  // (tracklets only: see above)
        if (plane_out) *plane_out = mld_plane_result{{0.f, 0.f, 0.f, 0.f}, 0, 1, 0};
        s.plane_decided = false;
        deliver_tracks(false);
        return fail(ctx, MLD_ERR_CLOUD_TOO_SMALL, "In GroundPlane: Input pointcloud is invalid");
    }
    deliver_tracks(true);
    return MLD_OK;
}

int mld_tracklets_frame(mld_ctx* ctx, int slot_cur, int slot_last, const void* pts_host, int64_t n, int stride_bytes,
                        const mld_plane_request* plane, const float coeffs[4], const int32_t* inlier_idx_host,
                        int64_t n_inliers, const float* u_new, const float* v_new, const float* u_old, const float* v_old,
                        const uint8_t* is_new, int64_t n_tracks, float* d_cur_out, float* d_last_out, int32_t* type_cur_out,
                        int32_t* type_last_out, int64_t* n_new_host, mld_plane_result* plane_out) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    FramePlane fp;
    if (plane) {
        if (plane->kind != MLD_PLANE_RANSAC && plane->kind != MLD_PLANE_SEMANTIC)
            return fail(ctx, MLD_ERR_INVALID_ARG, "mld_plane_request: kind must be MLD_PLANE_RANSAC or MLD_PLANE_SEMANTIC");
        fp.kind = plane->kind == MLD_PLANE_RANSAC ? FramePlane::RANSAC : FramePlane::SEMANTIC;
        fp.req = plane;
    } else if (coeffs) {
        fp.kind = FramePlane::SUPPLIED;
        fp.coeffs = coeffs;
        fp.inliers = inlier_idx_host;
        fp.n_inliers = n_inliers;
    }
    FrameTracks tk;
    tk.slot_last = slot_last;
    tk.u_new = u_new;
    tk.v_new = v_new;
    tk.u_old = u_old;
    tk.v_old = v_old;
    tk.is_new = is_new;
    tk.n = n_tracks;
    tk.d_cur_out = d_cur_out;
    tk.d_last_out = d_last_out;
    tk.type_cur_out = type_cur_out;
    tk.type_last_out = type_last_out;
    tk.n_new_out = n_new_host;
    return frame_call(ctx, slot_cur, pts_host, n, stride_bytes, fp, nullptr, 0, nullptr, nullptr, plane_out, &tk);
}

int mld_calculate_depth_frame(mld_ctx* ctx, int slot, const void* pts_host, int64_t n, int stride_bytes,
                              const float coeffs[4], const int32_t* inlier_idx_host, int64_t n_inliers,
                              const double* uv_host, int64_t F, double* depth_out_host, int32_t* type_out_host) {
    FramePlane fp;
    if (coeffs) {
        fp.kind = FramePlane::SUPPLIED;
        fp.coeffs = coeffs;
        fp.inliers = inlier_idx_host;
        fp.n_inliers = n_inliers;
    }
    return frame_call(ctx, slot, pts_host, n, stride_bytes, fp, uv_host, F, depth_out_host, type_out_host, nullptr);
}

int mld_calculate_depth_frame_estimate(mld_ctx* ctx, int slot, const void* pts_host, int64_t n, int stride_bytes,
                                       const mld_plane_request* plane, const double* uv_host, int64_t F,
                                       double* depth_out_host, int32_t* type_out_host, mld_plane_result* plane_out) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (!plane || (plane->kind != MLD_PLANE_RANSAC && plane->kind != MLD_PLANE_SEMANTIC))
        return fail(ctx, MLD_ERR_INVALID_ARG, "mld_plane_request: kind must be MLD_PLANE_RANSAC or MLD_PLANE_SEMANTIC");
    FramePlane fp;
    fp.kind = plane->kind == MLD_PLANE_RANSAC ? FramePlane::RANSAC : FramePlane::SEMANTIC;
    fp.req = plane;
    return frame_call(ctx, slot, pts_host, n, stride_bytes, fp, uv_host, F, depth_out_host, type_out_host, plane_out);
}

int mld_frame_timing(mld_ctx* ctx, double out_us[10]) {
    if (!ctx || !out_us) return MLD_ERR_INVALID_ARG;
    for (int i = 0; i < 10; i++) out_us[i] = ctx->fr_us[i];
    return MLD_OK;
}

int mld_calculate_depth_debug(mld_ctx* ctx, int slot, const double* uv_host, int64_t F, double* depth_out_host,
                              int32_t* type_out_host, double* corners_out_host) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if ((rc = precheck_calc(ctx, s, F))) return rc;
    if (F == 0) return MLD_OK;
    if (!uv_host || !depth_out_host || !corners_out_host) return fail(ctx, MLD_ERR_INVALID_ARG, "null feature/output pointer");
    double *uv = nullptr, *depth = nullptr, *corners = nullptr;
    int32_t* type = nullptr;
    hipError_t e = hipMalloc((void**)&uv, (size_t)F * 2 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&depth, (size_t)F * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&type, (size_t)F * sizeof(int32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&corners, (size_t)F * 9 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(uv, uv_host, (size_t)F * 2 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(corners, 0xFF, (size_t)F * 9 * sizeof(double), ctx->stream);  // all-ones = NaN
    if (e == hipSuccess) {
        // set_all_depths_to_zero never reaches the plane code: no corners
        rc = calc_one(ctx, slot, uv, F, depth, type, ctx->P.set_all_depths_to_zero ? nullptr : corners);
        if (rc == MLD_OK) {
            e = hipMemcpyAsync(depth_out_host, depth, (size_t)F * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess && type_out_host)
                e = hipMemcpyAsync(type_out_host, type, (size_t)F * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(corners_out_host, corners, (size_t)F * 9 * sizeof(double), hipMemcpyDeviceToHost,
                                   ctx->stream);
        }
    }
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    // the slot must not keep pointers into the temporaries
    s.d.uv = nullptr;
    s.d.depth = nullptr;
    s.d.type = nullptr;
    s.d.F = 0;
    for (void* q : {(void*)uv, (void*)depth, (void*)type, (void*)corners})
        if (q) (void)hipFree(q);
    if (rc) return rc;
    HIP_TRY(ctx, e);
    HIP_TRY(ctx, e2);
    return MLD_OK;
}

int mld_get_ground_plane_cloud(mld_ctx* ctx, int slot, double* xyz_out, int64_t capacity, int64_t* n_out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if (!n_out) return fail(ctx, MLD_ERR_INVALID_ARG, "null count pointer");
    *n_out = 0;
    if (!s.cloud_set) return fail(ctx, MLD_ERR_NOT_INITIALIZED, "no cloud set for this slot");
    if (!s.d.has_plane || !s.d.inlier_mask || s.d.n == 0) return MLD_OK;
    if ((rc = ensure_full(ctx, s))) return rc;
    const size_t words = ((size_t)s.d.n + 31) / 32;
    std::vector<uint32_t> mask(words);
    std::vector<double> cam((size_t)s.d.n * 3);
    HIP_TRY(ctx, hipMemcpyAsync(mask.data(), s.d.inlier_mask, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(cam.data(), s.cam, cam.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    int64_t k = 0;
    for (int64_t i = 0; i < s.d.n; i++) {
        if (!((mask[(size_t)i >> 5] >> (i & 31)) & 1u)) continue;
        const double* q = &cam[(size_t)i * 3];
        if (ctx->P.ransac_plane_use_camx_treshold && !(std::fabs(q[0]) <= ctx->P.ransac_plane_treshold_camx))
            continue;  // DepthEstimator.cpp:300-307
        if (xyz_out && k < capacity) {
            xyz_out[3 * k] = q[0];
            xyz_out[3 * k + 1] = q[1];
            xyz_out[3 * k + 2] = q[2];
        }
        k++;
    }
    *n_out = k;
    if (xyz_out && k > capacity) return fail(ctx, MLD_ERR_CAPACITY, "output buffer too small");
    return MLD_OK;
}

int mld_calculate_depths_device(mld_ctx* ctx, int n_slots, const double* const* uv_dev, const int64_t* F,
                                double* const* depth_out_dev, int32_t* const* type_out_dev) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (n_slots < 1 || n_slots > (int)ctx->slots.size() || !uv_dev || !F || !depth_out_dev)
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad slot count / null arrays");
    int rc = bind_device(ctx);
    if (rc) return rc;
    int64_t max_F = 0;
    for (int i = 0; i < n_slots; i++) {
        Slot& s = ctx->slots[i];
        if ((rc = precheck_calc(ctx, s, F[i]))) return rc;
        if (F[i] > 0 && (!uv_dev[i] || !depth_out_dev[i])) return fail(ctx, MLD_ERR_INVALID_ARG, "null feature/output pointer");
        s.d.uv = uv_dev[i];
        s.d.F = F[i];
        s.d.depth = depth_out_dev[i];
        s.d.type = type_out_dev ? type_out_dev[i] : nullptr;
        if ((rc = ensure_queues(ctx, s, F[i]))) return rc;
        max_F = std::max(max_F, F[i]);
    }
    if (ctx->P.set_all_depths_to_zero) {
        for (int i = 0; i < n_slots; i++)
            if (F[i] > 0)
                hipLaunchKernelGGL(k_fill_zero, dim3((unsigned)((F[i] + 255) / 256)), dim3(256), 0, ctx->stream,
                                   depth_out_dev[i], type_out_dev ? type_out_dev[i] : nullptr, (long long)F[i]);
        HIP_TRY(ctx, hipGetLastError());
        return MLD_OK;
    }
    if ((rc = upload_descs(ctx, n_slots))) return rc;
    return launch_features(ctx, n_slots, max_F, false, 0);
}

// ---------------------------------------------------------------------------- tracklets
static int ensure_tracklet_scratch(mld_ctx* ctx, size_t n) {
    if (n <= ctx->trk_cap && ctx->trk_uv_cur) return MLD_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    void* olds[] = {ctx->trk_uv_cur, ctx->trk_uv_last, ctx->trk_depth_cur, ctx->trk_depth_last, ctx->trk_type_cur,
                    ctx->trk_type_last, ctx->trk_rank};
    for (void* p : olds)
        if (p) HIP_TRY(ctx, hipFree(p));
    size_t m = n < 1 ? 1 : n;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_uv_cur, m * 2 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_uv_last, m * 2 * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_depth_cur, m * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_depth_last, m * sizeof(double)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_type_cur, m * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_type_last, m * sizeof(int32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_rank, m * sizeof(int32_t)));
    if (!ctx->trk_n_new) HIP_TRY(ctx, hipMalloc((void**)&ctx->trk_n_new, sizeof(long long)));
    ctx->trk_cap = m;
    return MLD_OK;
}

int mld_tracklets_depth_device(mld_ctx* ctx, int slot_cur, int slot_last, const float* u_new, const float* v_new,
                               const float* u_old, const float* v_old, const uint8_t* is_new, int64_t n_tracks,
                               float* d_cur_out, float* d_last_out, int32_t* type_cur_out, int32_t* type_last_out,
                               int64_t* n_new_host) {
    int rc = check_slot(ctx, slot_cur);
    if (rc) return rc;
    if (slot_last >= 0 && (rc = check_slot(ctx, slot_last))) return rc;
    if (slot_last == slot_cur) return fail(ctx, MLD_ERR_INVALID_ARG, "slot_last must differ from slot_cur");
    if ((rc = bind_device(ctx))) return rc;
    if (n_tracks < 0) return fail(ctx, MLD_ERR_INVALID_ARG, "negative track count");
    if (n_tracks > 0 && (!u_new || !v_new || !u_old || !v_old || !is_new || !d_cur_out || !d_last_out))
        return fail(ctx, MLD_ERR_INVALID_ARG, "null tracklet array");
    if ((rc = precheck_calc(ctx, ctx->slots[slot_cur], n_tracks))) return rc;
    if (slot_last >= 0 && (rc = precheck_calc(ctx, ctx->slots[slot_last], n_tracks))) return rc;
    if ((rc = ensure_tracklet_scratch(ctx, (size_t)n_tracks))) return rc;
    if (n_tracks == 0) {
        if (n_new_host) *n_new_host = 0;
        return MLD_OK;
    }
    hipLaunchKernelGGL(k_tracklet_gather, dim3(1), dim3(kTrkBlock), 0, ctx->stream, u_new, v_new, u_old, v_old, is_new,
                       (long long)n_tracks, ctx->trk_uv_cur, ctx->trk_uv_last, ctx->trk_rank, ctx->trk_n_new);
    HIP_TRY(ctx, hipGetLastError());
    if ((rc = calc_one(ctx, slot_cur, ctx->trk_uv_cur, n_tracks, ctx->trk_depth_cur, ctx->trk_type_cur))) return rc;
    if (slot_last >= 0) {
        Slot& sl = ctx->slots[slot_last];
        sl.d.F_dev = ctx->trk_n_new;  // the number of new tracks is only known on the device
        rc = calc_one(ctx, slot_last, ctx->trk_uv_last, n_tracks, ctx->trk_depth_last, ctx->trk_type_last);
        sl.d.F_dev = nullptr;
        if (rc) return rc;
    }
    hipLaunchKernelGGL(k_tracklet_scatter, dim3((unsigned)((n_tracks + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->trk_depth_cur, ctx->trk_type_cur, ctx->trk_depth_last, ctx->trk_type_last, ctx->trk_rank,
                       (long long)n_tracks, slot_last >= 0 ? 1 : 0, d_cur_out, d_last_out, type_cur_out, type_last_out);
    HIP_TRY(ctx, hipGetLastError());
    if (n_new_host) {
        long long nn = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&nn, ctx->trk_n_new, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        *n_new_host = (int64_t)nn;
    }
    return MLD_OK;
}

// TrackletDepthModule::process (tracklet_depth_module.cpp:261-396) for the current frames of n_seq SEQUENCES in one
// launch set.  The context's frame slots form two banks of n_seq slots: sequence s has its current frame in slot
// bank_cur * n_seq + s (cloud and plane set with mld_set_clouds_planes_range_device) and its previous frame, still
// resident, in slot (1 - bank_cur) * n_seq + s.  One gather launch, k_classify -> k_feature_fused -> k_feature_wave over
// all 2 * n_seq slots (the previous-frame slots evaluate the previous features of the NEW tracks only), one scatter
// launch.  Asynchronous; per-sequence results are identical to n_seq calls of mld_tracklets_depth_device.
int mld_tracklets_depths_device(mld_ctx* ctx, int n_seq, int bank_cur, int have_last, const float* const* u_new,
                                const float* const* v_new, const float* const* u_old, const float* const* v_old,
                                const uint8_t* const* is_new, const int64_t* n_tracks, float* const* d_cur_out,
                                float* const* d_last_out, int32_t* const* type_cur_out, int32_t* const* type_last_out) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    if (n_seq < 1 || 2 * n_seq > (int)ctx->slots.size() || (bank_cur != 0 && bank_cur != 1))
        return fail(ctx, MLD_ERR_INVALID_ARG, "bad sequence count / bank (the context needs 2 * n_seq frame slots)");
    if (!u_new || !v_new || !u_old || !v_old || !is_new || !n_tracks || !d_cur_out || !d_last_out)
        return fail(ctx, MLD_ERR_INVALID_ARG, "null tracklet array");
    int rc = bind_device(ctx);
    if (rc) return rc;
    int64_t max_n = 0;
    const int cur0 = bank_cur * n_seq, last0 = (1 - bank_cur) * n_seq;
    for (int q = 0; q < n_seq; q++) {
        if (n_tracks[q] < 0) return fail(ctx, MLD_ERR_INVALID_ARG, "negative track count");
        if (n_tracks[q] > 0 && (!u_new[q] || !v_new[q] || !u_old[q] || !v_old[q] || !is_new[q] || !d_cur_out[q] || !d_last_out[q]))
            return fail(ctx, MLD_ERR_INVALID_ARG, "null tracklet array");
        if ((rc = precheck_calc(ctx, ctx->slots[cur0 + q], n_tracks[q]))) return rc;
        if (have_last && (rc = precheck_calc(ctx, ctx->slots[last0 + q], n_tracks[q]))) return rc;
        max_n = std::max(max_n, n_tracks[q]);
    }
    if (max_n == 0) return MLD_OK;
    if (ctx->P.set_all_depths_to_zero) return fail(ctx, MLD_ERR_UNSUPPORTED_MODE, "set_all_depths_to_zero: use the per-sequence call");
    // scratch: per sequence cap tracks of [uv_cur 16 | uv_last 16 | depth_cur 8 | depth_last 8 | type_cur 4 | type_last 4 |
    // rank 4] bytes, then the new-track counters
    if ((size_t)max_n > ctx->trkb_cap || n_seq > ctx->trkb_seqs || !ctx->trkb) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->trkb) HIP_TRY(ctx, hipFree(ctx->trkb));
        if (ctx->trkb_desc) HIP_TRY(ctx, hipFree(ctx->trkb_desc));
        ctx->trkb = nullptr;
        ctx->trkb_desc = nullptr;
        const size_t cap = ((size_t)std::max<int64_t>(max_n, (int64_t)ctx->trkb_cap) + 63) & ~(size_t)63;
        const int seqs = std::max(n_seq, ctx->trkb_seqs);
        HIP_TRY(ctx, hipMalloc((void**)&ctx->trkb, (size_t)seqs * (cap * 60 + 64)));
        HIP_TRY(ctx, hipMalloc((void**)&ctx->trkb_desc, (size_t)seqs * sizeof(TrkSeq)));
        ctx->trkb_cap = cap;
        ctx->trkb_seqs = seqs;
    }
    const size_t cap = ctx->trkb_cap, per_seq = cap * 60 + 64;
    ctx->trkb_host.resize((size_t)n_seq);
    for (int q = 0; q < n_seq; q++) {
        unsigned char* base = ctx->trkb + (size_t)q * per_seq;
        TrkSeq& t = ctx->trkb_host[(size_t)q];
        t.u_new = u_new[q];
        t.v_new = v_new[q];
        t.u_old = u_old[q];
        t.v_old = v_old[q];
        t.is_new = is_new[q];
        t.n = n_tracks[q];
        t.uv_cur = reinterpret_cast<double*>(base);
        t.uv_last = reinterpret_cast<double*>(base + cap * 16);
        double* depth_cur = reinterpret_cast<double*>(base + cap * 32);
        double* depth_last = reinterpret_cast<double*>(base + cap * 40);
        int32_t* type_cur = reinterpret_cast<int32_t*>(base + cap * 48);
        int32_t* type_last = reinterpret_cast<int32_t*>(base + cap * 52);
        t.rank = reinterpret_cast<int32_t*>(base + cap * 56);
        t.n_new = reinterpret_cast<long long*>(base + cap * 60);
        t.depth_cur = depth_cur;
        t.depth_last = depth_last;
        t.type_cur = type_cur;
        t.type_last = type_last;
        t.d_cur_out = d_cur_out[q];
        t.d_last_out = d_last_out[q];
        t.type_cur_out = type_cur_out ? type_cur_out[q] : nullptr;
        t.type_last_out = type_last_out ? type_last_out[q] : nullptr;
        t.have_last = have_last ? 1 : 0;
        t.pad_ = 0;
        // the two slots of the sequence: every track on the current frame, the new tracks' previous features on the
        // previous frame (their number is only known on the device)
        Slot& sc = ctx->slots[cur0 + q];
        sc.d.uv = t.uv_cur;
        sc.d.F = t.n;
        sc.d.F_dev = nullptr;
        sc.d.depth = depth_cur;
        sc.d.type = type_cur;
        if ((rc = ensure_queues(ctx, sc, t.n))) return rc;
        Slot& sl = ctx->slots[last0 + q];
        sl.d.uv = t.uv_last;
        sl.d.F = have_last ? t.n : 0;
        sl.d.F_dev = have_last ? t.n_new : nullptr;
        sl.d.depth = depth_last;
        sl.d.type = type_last;
        if (have_last && (rc = ensure_queues(ctx, sl, t.n))) return rc;
    }
    if ((rc = upload_small(ctx, ctx->trkb_desc, ctx->trkb_host.data(), (size_t)n_seq * sizeof(TrkSeq), ctx->stream))) return rc;
    const unsigned chunks = (unsigned)((max_n + kTrkBlock - 1) / kTrkBlock);
    hipLaunchKernelGGL(k_tracklets_gather, dim3(chunks, (unsigned)n_seq), dim3(kTrkBlock), 0, ctx->stream, ctx->trkb_desc);
    HIP_TRY(ctx, hipGetLastError());
    // Few sequences: one classification block per slot leaves most CUs idle (16 sequences of 10 000 tracks: 32 blocks of
    // 60 us on 256 CUs).  The features of every slot are then dealt to G group descriptors - copies of the slot's with
    // the feature arrays, the queues and the counters sliced - so that the launch set has ~256 classification blocks (more
    // groups cost the feature kernel more partly filled wavefronts than the classification gains: LAB.md 5.24).
    // Group g of slot i sits at g * n_desc + i: the groups of a slot stay on the slot's XCD (decode_block).
    const int n_desc = 2 * n_seq;
    const int G = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(8, 256 / n_desc), max_n / 1024));
    if (G > 1) {
        const int64_t Fg = (((max_n + G - 1) / G) + 63) & ~(int64_t)63;
        const size_t n_sub = (size_t)n_desc * (size_t)G;
        if (n_sub > ctx->sub_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->d_sub) HIP_TRY(ctx, hipFree(ctx->d_sub));
            if (ctx->sub_counts) HIP_TRY(ctx, hipFree(ctx->sub_counts));
            ctx->d_sub = nullptr;
            ctx->sub_counts = nullptr;
            ctx->sub_cap = 0;
            HIP_TRY(ctx, hipMalloc((void**)&ctx->d_sub, n_sub * sizeof(SlotDesc)));
            HIP_TRY(ctx, hipMalloc((void**)&ctx->sub_counts, n_sub * 2 * sizeof(int32_t)));
            ctx->sub_cap = n_sub;
        }
        ctx->h_sub.resize(n_sub);
        for (int g = 0; g < G; g++) {
            const int64_t off = (int64_t)g * Fg;
            for (int i = 0; i < n_desc; i++) {
                const size_t idx = (size_t)g * (size_t)n_desc + (size_t)i;
                SlotDesc d = ctx->slots[i].d;
                const int64_t left = d.F - off;
                d.F = left < 0 ? 0 : (left < Fg ? left : Fg);
                if (d.F > 0) {  // (a group without features keeps the slot's pointers: nothing is read through them)
                    d.uv += 2 * off;
                    d.depth += off;
                    if (d.type) d.type += off;
                    if (d.corners) d.corners += 9 * off;
                    if (d.ovf_queue) d.ovf_queue += 2 * off;
                    if (d.live_queue) d.live_queue += off;
                }
                d.F_dev_off = off;
                d.ovf_count = ctx->sub_counts + 2 * idx;
                d.live_count = ctx->sub_counts + 2 * idx + 1;
                ctx->h_sub[idx] = d;
            }
        }
        if ((rc = upload_small(ctx, ctx->d_sub, ctx->h_sub.data(), n_sub * sizeof(SlotDesc), ctx->stream))) return rc;
        rc = launch_features(ctx, (int)n_sub, Fg, false, 0, nullptr, ctx->d_sub);
    } else {
        // one launch set over both banks; the banks carry different map tags, so the descriptors hold them
        if ((rc = upload_descs(ctx, n_desc, nullptr, 0, true))) return rc;
        rc = launch_features(ctx, n_desc, max_n, false, 0);
    }
    for (int q = 0; q < n_seq; q++) ctx->slots[last0 + q].d.F_dev = nullptr;  // (the uploaded copy keeps it)
    if (rc) return rc;
    hipLaunchKernelGGL(k_tracklets_scatter, dim3((unsigned)((max_n + 255) / 256), (unsigned)n_seq), dim3(256), 0, ctx->stream,
                       ctx->trkb_desc);
    HIP_TRY(ctx, hipGetLastError());
    return MLD_OK;
}

int mld_tracklets_depth(mld_ctx* ctx, int slot_cur, int slot_last, const float* u_new, const float* v_new,
                        const float* u_old, const float* v_old, const uint8_t* is_new, int64_t n_tracks,
                        float* d_cur_out, float* d_last_out, int32_t* type_cur_out, int32_t* type_last_out,
                        int64_t* n_new_host) {
    int rc = check_slot(ctx, slot_cur);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    if (n_tracks < 0) return fail(ctx, MLD_ERR_INVALID_ARG, "negative track count");
    if (n_tracks > 0 && (!u_new || !v_new || !u_old || !v_old || !is_new || !d_cur_out || !d_last_out))
        return fail(ctx, MLD_ERR_INVALID_ARG, "null tracklet array");
    const size_t n = (size_t)n_tracks;
    // staging layout: 4 float inputs | 2 float outputs | 2 int32 outputs | is_new bytes
    const size_t bytes = n * (4 * 4 + 2 * 4 + 2 * 4 + 1) + 64;
    if ((rc = grow(ctx, ctx->trk_stage, ctx->trk_stage_cap, bytes))) return rc;
    float* du0 = reinterpret_cast<float*>(ctx->trk_stage);
    float *dv0 = du0 + n, *du1 = dv0 + n, *dv1 = du1 + n, *dd0 = dv1 + n, *dd1 = dd0 + n;
    int32_t* dt0 = reinterpret_cast<int32_t*>(dd1 + n);
    int32_t* dt1 = dt0 + n;
    uint8_t* dnew = reinterpret_cast<uint8_t*>(dt1 + n);
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(du0, u_new, n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dv0, v_new, n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(du1, u_old, n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dv1, v_old, n * 4, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(dnew, is_new, n, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(dd1, 0, n * 4, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(dt1, 0, n * 4, ctx->stream));
    }
    rc = mld_tracklets_depth_device(ctx, slot_cur, slot_last, du0, dv0, du1, dv1, dnew, n_tracks, dd0, dd1, dt0, dt1,
                                    nullptr);
    if (rc) return rc;
    if (n) {
        HIP_TRY(ctx, hipMemcpyAsync(d_cur_out, dd0, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (type_cur_out) HIP_TRY(ctx, hipMemcpyAsync(type_cur_out, dt0, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        // d_last / type_last are defined for new tracks only: copy through a host temporary and merge
        std::vector<float> tmp(n);
        std::vector<int32_t> tmpt(n);
        HIP_TRY(ctx, hipMemcpyAsync(tmp.data(), dd1, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(tmpt.data(), dt1, n * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        int64_t nn = 0;
        for (size_t i = 0; i < n; i++)
            if (is_new[i]) {
                d_last_out[i] = tmp[i];
                if (type_last_out) type_last_out[i] = tmpt[i];
                nn++;
            }
        if (n_new_host) *n_new_host = nn;
    } else if (n_new_host) {
        *n_new_host = 0;
    }
    return MLD_OK;
}

// ---------------------------------------------------------------------------- getters
int mld_get_path_counts(mld_ctx* ctx, int slot, int64_t* lane_path, int64_t* handed_over) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!lane_path || !handed_over) return fail(ctx, MLD_ERR_INVALID_ARG, "null output");
    if ((rc = bind_device(ctx))) return rc;
    const Slot& s = ctx->slots[slot];
    int32_t live = 0, ovf = 0;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (s.d.live_count) HIP_TRY(ctx, hipMemcpy(&live, s.d.live_count, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (s.d.ovf_count) HIP_TRY(ctx, hipMemcpy(&ovf, s.d.ovf_count, sizeof(int32_t), hipMemcpyDeviceToHost));
    *lane_path = live;
    *handed_over = ovf;
    return MLD_OK;
}

int mld_get_visible_count(mld_ctx* ctx, int slot, int64_t* n_visible) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if (!n_visible) return fail(ctx, MLD_ERR_INVALID_ARG, "null output");
    if ((rc = bind_device(ctx))) return rc;
    if ((rc = ensure_full(ctx, ctx->slots[slot]))) return rc;
    *n_visible = ctx->slots[slot].nvis;
    return MLD_OK;
}

int mld_get_visible_image_points(mld_ctx* ctx, int slot, double* uv_out, int64_t capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if ((rc = ensure_full(ctx, s))) return rc;
    if (capacity < s.nvis) return fail(ctx, MLD_ERR_CAPACITY, "output buffer too small");
    if (s.nvis) {
        HIP_TRY(ctx, hipMemcpyAsync(uv_out, s.img_vis, (size_t)s.nvis * 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MLD_OK;
}

int mld_get_point_index(mld_ctx* ctx, int slot, int32_t* index_out, int64_t capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if ((rc = ensure_full(ctx, s))) return rc;
    if (capacity < s.nvis) return fail(ctx, MLD_ERR_CAPACITY, "output buffer too small");
    if (s.nvis) {
        HIP_TRY(ctx, hipMemcpyAsync(index_out, s.pidx, (size_t)s.nvis * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MLD_OK;
}

int mld_get_cloud_camera_cs(mld_ctx* ctx, int slot, double* xyz_out, int64_t capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if ((rc = ensure_full(ctx, s))) return rc;
    if (capacity < s.d.n) return fail(ctx, MLD_ERR_CAPACITY, "output buffer too small");
    if (s.d.n) {
        HIP_TRY(ctx, hipMemcpyAsync(xyz_out, s.cam, (size_t)s.d.n * 3 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MLD_OK;
}

int mld_get_pixel_map(mld_ctx* ctx, int slot, int32_t* map_out, int64_t capacity) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if ((rc = ensure_full(ctx, s))) return rc;
    long long cells = (long long)ctx->cam.width * ctx->cam.height;
    if (capacity < cells) return fail(ctx, MLD_ERR_CAPACITY, "output buffer too small");
    int32_t* tmp = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&tmp, (size_t)cells * sizeof(int32_t)));
    hipLaunchKernelGGL(k_export_map, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, ctx->stream, s.d.map, s.d.tag,
                       s.rank, cells, tmp);
    hipError_t e = hipMemcpyAsync(map_out, tmp, (size_t)cells * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    HIP_TRY(ctx, e);
    return MLD_OK;
}

int mld_get_point_depth_cam_visible(mld_ctx* ctx, int slot, int64_t visible_index, double* depth_out) {
    int rc = check_slot(ctx, slot);
    if (rc) return rc;
    if ((rc = bind_device(ctx))) return rc;
    Slot& s = ctx->slots[slot];
    if ((rc = ensure_full(ctx, s))) return rc;
    if (!depth_out || visible_index < 0 || visible_index >= s.nvis) return fail(ctx, MLD_ERR_INVALID_ARG, "index out of range");
    int32_t raw = 0;
    HIP_TRY(ctx, hipMemcpy(&raw, s.pidx + visible_index, sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipMemcpy(depth_out, s.cam + 3 * (size_t)raw + 2, sizeof(double), hipMemcpyDeviceToHost));
    return MLD_OK;
}

// ---------------------------------------------------------------------------- timing
int mld_timing_enable(mld_ctx* ctx, int enable) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    ctx->timing = enable != 0;
    return MLD_OK;
}

int mld_timing_reset(mld_ctx* ctx) {
    if (!ctx) return MLD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (TimedLaunch& t : ctx->timed) {
        ctx->event_pool.push_back(t.e0);
        ctx->event_pool.push_back(t.e1);
    }
    ctx->timed.clear();
    return MLD_OK;
}

int mld_kernel_time_ms(mld_ctx* ctx, int which, double* avg_ms, int64_t* launches) {
    if (!ctx || !avg_ms) return MLD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0;
    int64_t cnt = 0;
    for (TimedLaunch& t : ctx->timed) {
        if (t.which != which) continue;
        float ms = 0;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, t.e0, t.e1));
        total += ms;
        cnt++;
    }
    *avg_ms = cnt ? total / (double)cnt : 0.0;
    if (launches) *launches = cnt;
    return MLD_OK;
}

}  // extern "C"

// read-back entry points of the diagnostic builds' instrumentation (nothing in the product build)
#define MLD_DIAG_HOST_PART
#include "mld_diag.h"
