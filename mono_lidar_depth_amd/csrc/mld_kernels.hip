// HIP kernels of the DepthEstimator hot path for gfx950 (MI355X, CDNA4, wave64).
//
// Compile with -ffp-contract=off: the pixel a LiDAR point falls into depends on int(u), int(v)
// (NeighborFinderPixel.cpp:41-42), so the projection must round exactly like the CPU path.
//
// Kernels
//   k_project_scatter : Transform_Cloud_LidarToCamera + getImagePoints + InitializeLidarProjection
//                       (DepthEstimator.cpp:156-217, camera_pinhole.h:84-97, NeighborFinderPixel.cpp:29-58)
//                       fused: one coalesced pass over the cloud, no intermediate arrays.
//   k_classify        : one block per frame slot; the slot's occupancy bitmap staged in LDS; features without enough
//                       neighbours get their result at once, the others go, in image-row order, to the live queue.
//   k_feature_fused   : the per-feature loop (DepthEstimator.cpp:455-597), one lane per live feature: ONE scan of the
//                       road window serves the main path (its narrow sub-list) and the road fallback.
//   k_feature_wave    : wavefront-per-feature variant for lists beyond the thread kernel's capacities (and debug mode).
//   k_project_full / k_scan_* / k_export_* : lazy debug getters (visible list, _pointIndex, pixel map).
//
// Pixel map encoding.  The reference map holds the VISIBLE index of the first point (in cloud order,
// with z_cam > 0) that falls into the pixel.  Visible indices preserve cloud order, so "first visible
// index" == "smallest ORIGINAL index".  The device map stores
//     key = tag << 25 | (0x7FFFFF - origIdx) << 2 | state   (atomicMax: smallest origIdx of the newest tag wins;
//                                      state = the point's relation to the ground plane, when known at projection)
// where tag is bumped per setInputCloud, which makes clearing the 1.86 MB map unnecessary (a stale key has
// a smaller tag and loses / is ignored).  The map is zero-filled when the tag wraps (every 127 frames).
// Neighbours are re-derived from the raw float point at gather time (bit-identical arithmetic), so no
// camera-frame copy of the cloud is ever written.
#include "mld_device.h"
#include "../../include/mld.h"
#include "mld_diag.h"  // instrumentation of the diagnostic builds; every macro is empty in the product build

namespace mld {

// Pointers that arrive inside SlotDesc are generic to the compiler, which would make every access a flat_*
// instruction.  They are always device-global memory: cast to address space 1 so global_load/store/atomic is emitted.
#define GPTR(T, p) ((const T __attribute__((address_space(1)))*)(p))
#define GPTRW(T, p) ((T __attribute__((address_space(1)))*)(p))
// ... and in the constant address space: wave-uniform addresses of memory no kernel writes while this one runs - scalar loads
#define CPTR(T, p) ((const T __attribute__((address_space(4)))*)(p))
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4u __attribute__((ext_vector_type(4), aligned(4)));  // 16-byte load from a 4-byte aligned address

// ------------------------------------------------------------------------------------------------
// wave64 helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int prefix_count(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// "any lane" straight from the comparison mask (__any() turns the predicate into an integer and compares it again)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// Wavefront max / min as DPP reductions: four row shifts, two row broadcasts, one v_readlane - six dependent VALU steps
// and no LDS (the __shfl_xor butterfly is six dependent ds_bpermute round trips, ~100 cycles each, through the same LDS
// pipe the lists of the feature kernels live in).  A lane without a valid DPP source keeps its own value (old = v), so
// no identity element is needed.  For wave-uniform control flow with all 64 lanes active (every caller: one-wave blocks);
// the result is uniform (lane 63's).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_self(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_self_f64(double v) {
    const int lo = dpp_self<CTRL, ROW_MASK>(__double2loint(v)), hi = dpp_self<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "row_bcast:15 / row_bcast:31 (DPP controls 0x142 / 0x143) exist on GFX9 / CDNA only: this library is written for gfx950"
#endif
// The reductions read lane 63 after row-wise prefix steps: a partial wave (or a call from divergent control flow) would
// return a stale register.  The test build (-DMLD_AB_SWITCHES, which the whole GPU parity suite also runs) traps on it.
#ifdef MLD_AB_SWITCHES
#define MLD_ASSERT_FULL_WAVE()                                            \
    do {                                                                  \
        if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();      \
    } while (0)
#else
#define MLD_ASSERT_FULL_WAVE() do {} while (0)
#endif
#define MLD_DPP_REDUCE(OP, MOV)                       \
    MLD_ASSERT_FULL_WAVE();                           \
    v = OP(v, MOV<0x111, 0xf>(v)); /* row_shr:1 */    \
    v = OP(v, MOV<0x112, 0xf>(v)); /* row_shr:2 */    \
    v = OP(v, MOV<0x114, 0xf>(v)); /* row_shr:4 */    \
    v = OP(v, MOV<0x118, 0xf>(v)); /* row_shr:8 */    \
    v = OP(v, MOV<0x142, 0xa>(v)); /* row_bcast:15 */ \
    v = OP(v, MOV<0x143, 0xc>(v)); /* row_bcast:31 */
__device__ __forceinline__ double wave_max_f64(double v) {
    MLD_DPP_REDUCE(fmax, dpp_self_f64)
    return readlane_f64(v, 63);
}
__device__ __forceinline__ double wave_min_f64(double v) {
    MLD_DPP_REDUCE(fmin, dpp_self_f64)
    return readlane_f64(v, 63);
}
// Sum with the same steps (a lane without a source adds +0.0).  Its association - prefix sums along the rows, then the
// rows' totals - is neither the reference's sequential one nor a butterfly's; the callers are the road estimator and
// the PCA of the wave-cooperative kernel, which answer to the 1e-4 m tolerance, not to bit equality.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_zero_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_add(double a, double b) { return a + b; }
__device__ __forceinline__ double wave_sum_f64(double v) {
    MLD_DPP_REDUCE(dpp_add, dpp_zero_f64)
    return readlane_f64(v, 63);
}
__device__ __forceinline__ int wave_max_i32(int v) {
    MLD_DPP_REDUCE(max, dpp_self)
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v) {
    MLD_DPP_REDUCE(min, dpp_self)
    return __builtin_amdgcn_readlane(v, 63);
}
#undef MLD_DPP_REDUCE

// blockIdx -> (slot, block-within-slot).  Slots are taken in groups of 8 whose blocks are interleaved, so that all
// blocks of one slot are congruent mod 8, i.e. land on the same XCD under round-robin dispatch: the slot's map and
// cloud stay in one L2.  The last n_slots % 8 slots (and batches of fewer than 8) use the plain slot-major order.
// This is a speed-only mapping; nothing depends on placement.
// (a negative n_slots selects the plain mapping throughout: MLD_NO_XCD=1, for A/B measurements)
// The same with the division by per_slot replaced by a multiplication with magic = ceil(2^32 / per_slot) (host side:
// div_magic) and one correction step: the quotient estimate umulhi(q, magic) is never too small and at most one too large.
// (The compiler's generic 32-bit division is ~35 scalar instructions in front of everything else a wavefront does.)
__device__ __forceinline__ int div_by_magic(int q, int d, uint32_t magic) {
    int e = (int)__umulhi((uint32_t)q, magic);
    e -= (q - e * d < 0) ? 1 : 0;
    return e;
}
__device__ __forceinline__ void decode_block_magic(int b, int n_slots, int per_slot, uint32_t magic, int& slot, int& j) {
    const int n8 = n_slots > 0 ? (n_slots & ~7) : 0;
    const int b8 = n8 * per_slot;
    if (b < b8) {
        const int x = b & 7, q = b >> 3;
        const int sq = div_by_magic(q, per_slot, magic);
        j = q - sq * per_slot;
        slot = sq * 8 + x;
    } else {
        const int r = b - b8;
        const int sr = div_by_magic(r, per_slot, magic);
        slot = n8 + sr;
        j = r - sr * per_slot;
    }
}
__device__ __forceinline__ void decode_block(int b, int n_slots, int per_slot, int& slot, int& j) {
    const int n8 = n_slots > 0 ? (n_slots & ~7) : 0;
    const int b8 = n8 * per_slot;
    if (b < b8) {
        int x = b & 7, q = b >> 3;
        int sq = q / per_slot;
        j = q - sq * per_slot;
        slot = sq * 8 + x;
    } else {
        const int r = b - b8;
        const int sr = r / per_slot;
        slot = n8 + sr;
        j = r - sr * per_slot;
    }
}

// ------------------------------------------------------------------------------------------------
// 3-vector arithmetic with the reference's (Eigen fixed-size) evaluation order
// ------------------------------------------------------------------------------------------------
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 vscale(V3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ V3 vdivs(V3 a, double s) { return {a.x / s, a.y / s, a.z / s}; }
__device__ __forceinline__ double vdot(V3 a, V3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
__device__ __forceinline__ double vsqnorm(V3 a) { return a.x * a.x + (a.y * a.y + a.z * a.z); }
__device__ __forceinline__ double vnorm(V3 a) { return sqrt(vsqnorm(a)); }
__device__ __forceinline__ V3 vnormalized(V3 a) {
    double z = vsqnorm(a);
    if (z > 0.0) return vdivs(a, sqrt(z));
    return a;
}
__device__ __forceinline__ V3 vcross(V3 a, V3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// A plane estimated on the device overrides the descriptor's host-side plane fields (wave-uniform scalar loads).
__device__ __forceinline__ void apply_plane_dev(SlotDesc& s) {
    if (s.plane_dev) {
        const auto* p = GPTR(PlaneDev, s.plane_dev);
        s.has_plane = p->has_plane;
        s.prior_n[0] = p->prior_n[0];
        s.prior_n[1] = p->prior_n[1];
        s.prior_n[2] = p->prior_n[2];
        s.prior_off = p->prior_off;
        s.coeffs[0] = p->coeffs[0];
        s.coeffs[1] = p->coeffs[1];
        s.coeffs[2] = p->coeffs[2];
        s.coeffs[3] = p->coeffs[3];
    }
}

// Raw float point -> camera frame (DepthEstimator.cpp:169,173).
__device__ __forceinline__ void load_point(const SlotDesc& s, long long i, double& x, double& y, double& z) {
    const unsigned char* p = s.cloud + (size_t)i * (size_t)s.stride;
    // (one 12-byte load whatever the cloud's alignment: see load_raw)
    x = (double)GPTR(float, p)[0];
    y = (double)GPTR(float, p)[1];
    z = (double)GPTR(float, p)[2];
}
__device__ __forceinline__ V3 lidar_to_cam(const Calib& c, double x, double y, double z) {
    V3 r;
    r.x = c.T[3] + ((c.T[0] * x + c.T[1] * y) + c.T[2] * z);
    r.y = c.T[7] + ((c.T[4] * x + c.T[5] * y) + c.T[6] * z);
    r.z = c.T[11] + ((c.T[8] * x + c.T[9] * y) + c.T[10] * z);
    return r;
}
// camera_pinhole.h:88-90: q = K p with all nine products, hnormalized.
__device__ __forceinline__ void project(const Calib& c, V3 p, double& u, double& v) {
    double q0 = (c.f * p.x + 0.0 * p.y) + c.cu * p.z;
    double q1 = (0.0 * p.x + c.f * p.y) + c.cv * p.z;
    double q2 = (0.0 * p.x + 0.0 * p.y) + 1.0 * p.z;
    u = q0 / q2;
    v = q1 / q2;
}

// ------------------------------------------------------------------------------------------------
// K1: projection + pixel-map scatter
// ------------------------------------------------------------------------------------------------
#ifndef MLD_PROJ_THREADS
#define MLD_PROJ_THREADS 256
#endif
constexpr int kProjThreads = MLD_PROJ_THREADS;
#ifndef MLD_PROJ_PER_THREAD
#define MLD_PROJ_PER_THREAD 4
#endif
constexpr int kProjPerThread = MLD_PROJ_PER_THREAD;

// tag_all != 0: every slot of the batch carries this map tag (the per-slot tag of the descriptor array is then not
// kept up to date, which lets steady-state batches run without re-uploading the descriptors).
#ifdef MLD_PROJ_VGPRS  // measurement builds: cap the register allocation (how many projection wavefronts fit beside the feature kernel)
#define MLD_PROJ_ATTR __attribute__((amdgpu_num_vgpr(MLD_PROJ_VGPRS)))
#else
#define MLD_PROJ_ATTR
#endif
// ALIGNED16: every cloud of the launch starts on a 16-byte boundary (the host checks): the kernel then holds ONE load
// sequence, and a point's load is waited for where the point is first used.  (With both sequences in one kernel their
// results meet in register copies behind the loads, i.e. every load of the thread is waited for before the first
// point is touched.)
// SINGLE: the one-frame call - its slot descriptor travels by value (no upload in front of the kernel); batches read
// theirs in device memory.  (As one kernel with a run-time flag every descriptor field was a select between the two behind
// a uniform branch: fourteen of them in front of the first point load.)
template <bool ALIGNED16, bool SINGLE>
__global__ __launch_bounds__(kProjThreads) MLD_PROJ_ATTR void k_project_scatter(const SlotDesc* __restrict__ slots, SlotDesc single,
                                                                  Calib c, int n_slots, int per_slot, uint32_t ps_magic,
                                                                  uint32_t tag_all) {
    // Raised issue priority: beside another context's feature kernels (long f64 sequences, always ready to issue)
    // the few instructions a projection wave needs between its loads and its atomics would otherwise wait their
    // turn; measured side by side 0.843 -> 0.815 ms per 1024 frames, alone no difference.
#ifndef MLD_PROJ_PRIO
#define MLD_PROJ_PRIO 3
#endif
    __builtin_amdgcn_s_setprio(MLD_PROJ_PRIO);
    int slot = 0, j = (int)blockIdx.x;
    if (!SINGLE) decode_block_magic((int)blockIdx.x, c.xcdAware ? n_slots : -n_slots, per_slot, ps_magic, slot, j);
    // ---- FIRST what the point loads need - cloud, point count, stride - and the loads themselves; the rest of the
    // descriptor and the plane's parameters are fetched while they are in flight.  (Round 6: the prologue used to resolve
    // the whole descriptor and - through six VECTOR loads and read-first-lanes, a full memory round trip - the plane's
    // coefficients before the first point load was issued: four dependent round trips per wavefront instead of two.)
    const SlotDesc& sg = slots[SINGLE ? 0 : slot];  // (never dereferenced when SINGLE)
    const unsigned char* cl = SINGLE ? single.cloud : sg.cloud;
    // 32-bit index and offset arithmetic throughout: a cloud has at most 2^23 - 1 points of 16 or 32 bytes
    const int n = (int)(SINGLE ? single.n : sg.n);
    const int stride = SINGLE ? single.stride : sg.stride;
    const int blk0 = j * (kProjThreads * kProjPerThread);
    // (cloud pointer and stride are fetched WITH the count - one scalar round trip - not behind the test below)
    asm volatile("" ::"s"(stride), "s"((uint32_t)(uintptr_t)cl), "s"((uint32_t)((uintptr_t)cl >> 32)));
    if (blk0 >= n) return;  // block-uniform (the clouds of a batch may differ in size)
    const int base = blk0 + (int)threadIdx.x;
    const uint32_t sh = (stride == 32) ? 5u : 4u;
    float fx[kProjPerThread], fy[kProjPerThread], fz[kProjPerThread];
    // every lane loads a valid point (index clamped to the cloud); lanes beyond the end are masked out of the result
    if (ALIGNED16 || (((size_t)cl) & 15) == 0) {
#pragma unroll
        for (int r = 0; r < kProjPerThread; r++) {
            const int i = min(base + r * kProjThreads, n - 1);
            // streamed once: do not pollute the caches
            f32x4 q = __builtin_nontemporal_load(GPTR(f32x4, cl + (size_t)((uint32_t)i << sh)));
            fx[r] = q.x;
            fy[r] = q.y;
            fz[r] = q.z;
        }
    } else {
#pragma unroll
        for (int r = 0; r < kProjPerThread; r++) {
            const int i = min(base + r * kProjThreads, n - 1);
            const auto* q = GPTR(float, cl + (size_t)((uint32_t)i << sh));
            fx[r] = q[0];
            fy[r] = q[1];
            fz[r] = q[2];
        }
    }
    __builtin_amdgcn_sched_barrier(0);  // (nothing below moves in front of the loads, no load behind what follows)
    // ---- the rest of the slot: tag, map, bitmap, inlier mask; ground-plane state of the visible points (rides in the map
    // keys when the plane is already known).  "Far" is the distance test of CalculateDepthSegmentationPlane
    // (DepthEstimator.cpp:810-815): camera -> lidar in f64, float, un-normalised float plane distance > threshold - a
    // property of the point alone.  The f64 round trip returns the raw float coordinates up to e = far_elin * m1 +
    // far_econst (host bound), so the distance evaluated on the RAW floats differs from the reference's by at most
    // cs (2^-24 m1 + 2 e) + 8 * 2^-24 (cs m1 + |d|), cs = |a|+|b|+|c|: a point is marked far / near only beyond a margin of
    // several times that, otherwise "unsure" (exact test in the feature kernel; a handful of points per frame at most).
    struct {
        uint32_t* map;
        uint32_t* bitmap;
        const uint32_t* inlier_mask;
        uint32_t tag;
    } s;
    s.map = SINGLE ? single.map : sg.map;
    s.bitmap = SINGLE ? single.bitmap : sg.bitmap;
    s.inlier_mask = SINGLE ? single.inlier_mask : sg.inlier_mask;
    s.tag = tag_all ? tag_all : (SINGLE ? single.tag : sg.tag);
    const bool flags_on = (SINGLE ? single.mask_in_key : sg.mask_in_key) != 0;
    float pa, pb, pcz, pd, fmg0, fmg1;  // (far_margins: prepared once per plane, not per wavefront)
    if (SINGLE) {
        pa = single.coeffs[0];
        pb = single.coeffs[1];
        pcz = single.coeffs[2];
        pd = single.coeffs[3];
        fmg0 = single.far_mg0;
        fmg1 = single.far_mg1;
        if (flags_on && single.plane_dev) {  // the plane of an estimation on the device lives in device memory
            const auto* q = CPTR(PlaneDev, single.plane_dev);
            pa = q->coeffs[0];
            pb = q->coeffs[1];
            pcz = q->coeffs[2];
            pd = q->coeffs[3];
            fmg0 = q->far_mg0;
            fmg1 = q->far_mg1;
        }
    } else {
        // block-uniform addresses in memory nothing writes while this kernel runs: scalar loads (constant address space),
        // into scalar registers (the register file decides how many projection wavefronts fit beside the other context's
        // feature kernel)
        const PlaneDev* pdv = sg.plane_dev;
        const bool dev = flags_on && pdv;
        const auto* cf = CPTR(float, dev ? pdv->coeffs : sg.coeffs);
        const auto* mg = CPTR(float, dev ? &pdv->far_mg0 : &sg.far_mg0);
        pa = cf[0];
        pb = cf[1];
        pcz = cf[2];
        pd = cf[3];
        fmg0 = mg[0];
        fmg1 = mg[1];
    }
    const float thrf = c.roadDistThrF;
    const double Wd = (double)c.W, Hd = (double)c.H;
    const float Wf = (float)c.W, Hf = (float)c.H;
    // translation terms of the single-precision transform: used by every point, kept in vector registers
    float t3 = c.Tf[3], t7 = c.Tf[7], t11 = c.Tf[11];
    asm volatile("" : "+v"(t3), "+v"(t7), "+v"(t11));
#pragma unroll
    for (int r = 0; r < kProjPerThread; r++) {
#ifdef MLD_PROJ_SCHED_BARRIER  // measurement builds: nothing moves across the points of a thread (LAB.md 4.19)
        __builtin_amdgcn_sched_barrier(0);
#endif
        int w = -1 - (int)threadIdx.x;  // occupancy-bitmap word of the point (or a unique negative value)
        uint32_t bits = 0u;             // its bit
        const int i = base + r * kProjThreads;
        const float x = fx[r], y = fy[r], z = fz[r];
        // Conservative single-precision pre-cull.  86 % of a 360-degree scan is behind the camera or outside its
        // field of view; those points need none of the f64 work below.  Every f32 quantity here is within 1e-6 * S
        // of its exact value, S being the sum of the absolute values of its terms; a point is dropped only if it
        // misses the image by more than 1e-5 * S, so no point the exact test would keep is lost.  The sums S are
        // bounded from above by rowmax * (|x|+|y|+|z|) + |t|, so each margin is linear in m1 with coefficients the host
        // prepared (Calib::pcm, rounded up).  A NaN intermediate compares false: the point falls through to the exact path.  The
        // stages are skipped per 64-point group (wave-uniform branches, no lane masking): consecutive points of a
        // scan ring leave the field of view together.
        const float* T = c.Tf;
        const float m1 = fabsf(x) + fabsf(y) + fabsf(z);
        const float zc = fmaf(T[8], x, fmaf(T[9], y, fmaf(T[10], z, t11)));
        // (m1 is NaN exactly when a coordinate is: such a point - a "no return" of an organised cloud - has a NaN camera
        // depth on the exact path and is never visible; without this test a single one keeps its whole group on the
        // exact path.)
        // (the lanes that are still candidates are kept as a 64-bit wave mask: every comparison lands in a scalar
        // register pair and the combinations, the "any lane left?" tests included, run on the scalar unit)
#define MLD_BALLOT(cond) __builtin_amdgcn_ballot_w64(cond)
        unsigned long long pm = MLD_BALLOT(i < n) & MLD_BALLOT(m1 == m1);
        // A coordinate beyond 1e18 (or infinite) anywhere in the group: no tests for this group.  Single-precision
        // products of such values overflow, and an overflowed +inf in "q - W z" would cull a point that the f64
        // arithmetic projects into the image.  (Never taken for a real scan.)
        if (!(MLD_BALLOT(!(m1 < 1e18f)) & pm)) {
            pm &= MLD_BALLOT(!(zc < -fmaf(c.pcm[0], m1, c.pcm[1])));
            if (!pm) continue;
            const float xc = fmaf(T[0], x, fmaf(T[1], y, fmaf(T[2], z, t3)));
            const float yc = fmaf(T[4], x, fmaf(T[5], y, fmaf(T[6], z, t7)));
            const float qa = fmaf(c.ff, xc, c.cuf * zc), qb = fmaf(c.ff, yc, c.cvf * zc);  // ~ u*z, v*z
            const float ma = fmaf(c.pcm[2], m1, c.pcm[3]), mb = fmaf(c.pcm[4], m1, c.pcm[5]);
            pm &= MLD_BALLOT(!(qa < -ma)) & MLD_BALLOT(!(fmaf(-Wf, zc, qa) > ma));
            pm &= MLD_BALLOT(!(qb < -mb)) & MLD_BALLOT(!(fmaf(-Hf, zc, qb) > mb));
        }
#undef MLD_BALLOT
        if (!pm) continue;
        const bool pass = (pm >> threadIdx.x % kWave) & 1ull;
        // exact path (identical to the CPU arithmetic)
        const V3 pc = lidar_to_cam(c, (double)x, (double)y, (double)z);
        // camera_pinhole.h:88-90 without the six products by zero of K: for finite points q0 = (f x + 0 y) + cu z,
        // q1 = (0 x + f y) + cv z and q2 = (0 x + 0 y) + 1 z are bit-identical to f x + cu z, f y + cv z and z (the
        // sign of a zero sum cannot turn 0 < u into true), and a non-finite coordinate fails the bounds below either way
        const double u = (c.f * pc.x + c.cu * pc.z) / pc.z;
        const double v = (c.f * pc.y + c.cv * pc.z) / pc.z;
        // NeighborFinderPixel.cpp:51: only z > 0 enters the map; DepthEstimator.cpp:186-187 strict bounds (imply the
        // inclusive test of camera_pinhole.h:93-95)
        if (pass && (pc.z > 0.0) && (u > 0.0) && (u < Wd) && (v > 0.0) && (v < Hd)) {
            const int xi = (int)u, yi = (int)v;  // truncation, NeighborFinderPixel.cpp:41-42
            uint32_t fl = 0u;
            if (flags_on) {
                const uint32_t inl = (GPTR(uint32_t, s.inlier_mask)[(uint32_t)i >> 5] >> ((uint32_t)i & 31u)) & 1u;
                const float dr = fabsf(pa * x + pb * y + pcz * z + pd);
                fl = (dr > thrf) ? (uint32_t)kPtFar : inl;
                if (!(fabsf(dr - thrf) > fmaf(fmg0, m1, fmg1))) fl = kPtUnsure;  // (a NaN distance is "unsure" too)
            }
            const uint32_t key = make_key(s.tag, (uint32_t)i, fl);
            // (24-bit multiplies: image sides and the bitmap stride are far below 2^24)
            __hip_atomic_fetch_max(GPTRW(uint32_t, s.map) + (uint32_t)__mul24(yi, c.W) + (uint32_t)xi, key,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w = __mul24(xi >> 5, c.bmStride) + yi;
            bits = 1u << (xi & 31);
        }
        // Occupancy bits (lets the feature kernel skip the mostly empty key rows).  Consecutive lanes hold consecutive
        // points of a scan ring, i.e. neighbouring pixels of one image row: bits of a run of lanes with the same
        // bitmap word are OR-ed along the run and only the run's last lane issues the atomic (~10x fewer atomics).
        // Runs need not be exact: every lane's bit reaches the last lane of its contiguous run (within its 16-lane
        // row), which always writes.  (The four point groups of a wavefront are a quarter of a scan ring apart:
        // usually only one or two of them reach this point.)
        if (!wave_any(bits != 0u)) continue;
        // DPP row shifts (no LDS traffic): runs are merged inside 16-lane rows; a row's last lane always writes
#define MLD_ROW_STEP(D)                                                                                     \
    {                                                                                                       \
        const int wn = __builtin_amdgcn_update_dpp((int)-0x40000000, w, 0x110 + (D), 0xf, 0xf, false);      \
        const uint32_t bn = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bits, 0x110 + (D), 0xf, 0xf, false); \
        if (wn == w) bits |= bn;                                                                            \
    }
        MLD_ROW_STEP(1)
        MLD_ROW_STEP(2)
        MLD_ROW_STEP(4)
        MLD_ROW_STEP(8)
#undef MLD_ROW_STEP
        const int wnext = __builtin_amdgcn_update_dpp((int)-0x40000000, w, 0x101 /* row_shl:1 */, 0xf, 0xf, false);
        if ((wnext != w) && bits)
            __hip_atomic_fetch_or(GPTRW(uint32_t, s.bitmap) + (uint32_t)w, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative per-feature depth (k_feature_wave: long lists, single frames, debug mode).  A wavefront takes the
// queue entries of its block one feature at a time:
//   phase 1 (lanes = window cells / neighbours, one feature at a time, wave-uniform control flow):
//           window scan -> ordered neighbour list in LDS -> histogram segmentation -> max-spanning triangle
//   phase 2 (lanes = features): planarity, viewing ray, ray/plane intersection, thresholds
//   phase 3 (lanes = neighbours, only features that fell through): wide window, ground-plane inliers, fit sums
//   phase 4 (lanes = features): road plane, intersection, thresholds
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool window_bounds(const Calib& c, double u, double v, double halfX, double halfY, int& x0,
                                              int& y0, int& nx, int& ny);

struct Lists {
    double* x;
    double* y;
    double* z;
    int* idx;  // original cloud index
    int* bin;
};

constexpr int kRecFields = 11;

// Window scan (NeighborFinderPixel.cpp:60-95) + 3-D gather (NeighborFinderBase.cpp:15-27).  Returns the
// neighbour count (wave-uniform); entries are in the reference's row-major scan order.
// inliers_only: only the points whose key carries the state "plane inlier" (valid when SlotDesc::mask_in_key).
__device__ int gather_window(const Calib& c, const SlotDesc& s, double u, double v, double halfX, double halfY,
                             const Lists& L, int lane, bool inliers_only = false) {
    if (!(isfinite(u) && isfinite(v))) return 0;  // reference: undefined behaviour (int cast of NaN)
    double a;
    a = u - halfX;
    double left = (a < 0.) ? 0. : a;  // std::max(a, 0.)
    a = u + halfX;
    double right = ((double)(c.W - 1) < a) ? (double)(c.W - 1) : a;  // std::min(a, W-1)
    a = v - halfY;
    double top = (a < 0.) ? 0. : a;
    a = v + halfY;
    double bottom = ((double)(c.H - 1) < a) ? (double)(c.H - 1) : a;
    int x0 = (int)left, x1 = (int)right, y0 = (int)top, y1 = (int)bottom;
    x0 = uniform(x0);
    x1 = uniform(x1);
    y0 = uniform(y0);
    y1 = uniform(y1);
    int nx = x1 - x0 + 1, ny = y1 - y0 + 1;
    if (nx <= 0 || ny <= 0) return 0;
    if (x0 < 0 || y0 < 0 || x1 >= c.W || y1 >= c.H) return 0;  // unreachable for finite features
    const int ncell = nx * ny;
    const float rnx = 1.0f / (float)nx;
    int k = 0;
    // Three 64-cell groups per pass - the road window of the reference's parameters (13 x 14 cells) in one -: their keys
    // are fetched together, then their points together (two memory round trips per pass instead of two per group;
    // unconditional loads from selected addresses: a predicated load is a branch and a full wait per group, LAB.md 6.14).
    constexpr int KG = 3;
    for (int base = 0; base < ncell; base += KG * kWave) {
        uint32_t key[KG];
        bool val[KG];
#pragma unroll
        for (int g = 0; g < KG; g++) {
            const int cidx = base + g * kWave + lane;
            val[g] = cidx < ncell;
            const int row = (int)(((float)cidx + 0.5f) * rnx);
            const int col = cidx - row * nx;
            const size_t off = val[g] ? ((size_t)(x0 + col) + (size_t)(y0 + row) * (size_t)c.W) : (size_t)0;
            key[g] = GPTR(uint32_t, s.map)[off];
        }
        int orig[KG], rank[KG];
        bool take[KG];
#pragma unroll
        for (int g = 0; g < KG; g++) {
            const bool has = val[g] && (key[g] >> kTagShift) == s.tag && (!inliers_only || (key[g] & 3u) == kPtInlier);
            orig[g] = (int)key_index(key[g]);
            const unsigned long long m = __ballot(has);
            rank[g] = k + prefix_count(m);
            take[g] = has && rank[g] < c.cap;
            k += __popcll(m);
        }
        double px[KG], py[KG], pz[KG];
#pragma unroll
        for (int g = 0; g < KG; g++) load_point(s, take[g] ? orig[g] : 0, px[g], py[g], pz[g]);
#pragma unroll
        for (int g = 0; g < KG; g++)
            if (take[g]) {
                const V3 pc = lidar_to_cam(c, px[g], py[g], pz[g]);
                L.x[rank[g]] = pc.x;
                L.y[rank[g]] = pc.y;
                L.z[rank[g]] = pc.z;
                L.idx[rank[g]] = orig[g];
            }
    }
    k = uniform(k);
    return k < c.cap ? k : c.cap;
}

// The same window through the keys alone (valid when SlotDesc::mask_in_key): the number of neighbours, of plane inliers
// among them, and whether any neighbour is far from the plane / undecided (k_project_scatter's ground-plane state).
__device__ void window_states(const Calib& c, const SlotDesc& s, double u, double v, double halfX, double halfY, int lane,
                              int& k_all, int& k_inl, bool& any_far, bool& any_unsure) {
    k_all = 0;
    k_inl = 0;
    any_far = false;
    any_unsure = false;
    int x0, y0, nx, ny;
    if (!window_bounds(c, u, v, halfX, halfY, x0, y0, nx, ny)) return;
    x0 = uniform(x0);
    y0 = uniform(y0);
    nx = uniform(nx);
    ny = uniform(ny);
    const int ncell = nx * ny;
    const float rnx = 1.0f / (float)nx;
    constexpr int KG = 3;  // (as gather_window: the keys of three 64-cell groups in one round trip)
    for (int base = 0; base < ncell; base += KG * kWave) {
        uint32_t key[KG];
        bool val[KG];
#pragma unroll
        for (int g = 0; g < KG; g++) {
            const int cidx = base + g * kWave + lane;
            val[g] = cidx < ncell;
            const int row = (int)(((float)cidx + 0.5f) * rnx);
            const int col = cidx - row * nx;
            const size_t off = val[g] ? ((size_t)(x0 + col) + (size_t)(y0 + row) * (size_t)c.W) : (size_t)0;
            key[g] = GPTR(uint32_t, s.map)[off];
        }
#pragma unroll
        for (int g = 0; g < KG; g++) {
            const uint32_t st = (val[g] && (key[g] >> kTagShift) == s.tag) ? (key[g] & 3u) : 0xFFu;
            k_all += (int)__popcll(__ballot(st != 0xFFu));
            k_inl += (int)__popcll(__ballot(st == kPtInlier));
            any_far = any_far || (__ballot(st == kPtFar) != 0ull);
            any_unsure = any_unsure || (__ballot(st == kPtUnsure) != 0ull);
        }
    }
}

// PointHistogram::FilterPointsMinDistBlob (HistogramPointDepth.cpp:15-123, Histogram.cpp:14-49).
// Compacts the list in place to the points of the first local-maximum bin.  Returns new count or -1.
__device__ int hist_segment(const Calib& c, int k, const Lists& L, int lane) {
    int md = 0;
    for (int b = 0; b < k; b += kWave) {
        int i = b + lane;
        if (i < k) {
            double d = L.z[i];
            d = (999. < d) ? 999. : d;  // DepthEstimator.cpp:743
            int ce = (int)ceil(d);
            md = max(md, ce);
        }
    }
    md = uniform(wave_max_i32(md));
    const int binCount = (int)((double)md / c.binW + 1.0);  // HistogramPointDepth.cpp:43
    if (binCount <= 1) return -1;
    int bmin = 0x7fffffff;
    for (int b = 0; b < k; b += kWave) {
        int i = b + lane;
        if (i < k) {
            double d = L.z[i];
            d = (999. < d) ? 999. : d;
            double value = (1e10 < d) ? 1e10 : d;  // Histogram.cpp:29
            double q = fabs(value / c.binW);
            double lim = (double)binCount - 1.;
            int bi = (int)((lim < q) ? lim : q);  // Histogram.cpp:30
            L.bin[i] = bi;
            bmin = min(bmin, bi);
        }
    }
    bmin = uniform(wave_min_i32(bmin));
    // HistogramPointDepth.cpp:70-85, started at the first non-empty bin (leading empty bins are no-ops)
    int binMaxId = -1, binMaxVal = -1, binValue = 0;
    for (int i = bmin; i < binCount; i++) {
        int last = binValue;
        int cnt = 0;
        for (int b = 0; b < k; b += kWave) {
            int e = b + lane;
            cnt += __popcll(__ballot(e < k && L.bin[e] == i));
        }
        binValue = uniform(cnt);
        if ((binValue > binMaxVal) && (binValue >= c.minCount)) {
            binMaxVal = binValue;
            binMaxId = i;
        } else if (binValue < binMaxVal)
            break;
        if ((last > 0) && (binValue == 0)) return -1;
    }
    if (binMaxId < 0) return -1;
    const double lower = (double)binMaxId * c.binW - 0.0 * c.binW;   // :99
    const double higher = (double)binMaxId * c.binW + 1.0 * c.binW;  // :100
    int kk = 0;
    for (int b = 0; b < k; b += kWave) {
        int i = b + lane;
        bool keep = false;
        double x = 0, y = 0, z = 0;
        int id = 0;
        if (i < k) {
            z = L.z[i];
            double d = (999. < z) ? 999. : z;
            keep = (d >= lower) && (d < higher);
            x = L.x[i];
            y = L.y[i];
            id = L.idx[i];
        }
        unsigned long long m = __ballot(keep);
        int rank = kk + prefix_count(m);
        if (keep) {
            L.x[rank] = x;
            L.y[rank] = y;
            L.z[rank] = z;
            L.idx[rank] = id;
        }
        kk += __popcll(m);
    }
    return uniform(kk);
}

// PlaneEstimationCalcMaxSpanningTriangle::CalculatePlaneCorners (PlaneEstimationCalcMaxSpanningTriangle.cpp:37-100)
// with _distTreshold = 0.  Serial tie-breaking is reproduced: strict '>' keeps the FIRST maximal pair in
// (i,j)-lexicographic order and the first maximal k.
__device__ bool max_spanning_triangle(int n, const Lists& L, int lane, int& ci, int& cj, int& ck) {
    if (n < 3) return false;
    // Farthest pair.  Every lane keeps the maximum of its own pairs and the smallest pair id (i * n + j) that reaches
    // it; one reduction at the end.  The serial loop's strict '>' keeps the first maximal pair in (i,j) order, i.e. the
    // smallest id among the pairs at the global maximum - which is what the final min over the lanes at the maximum
    // returns.  Rows i and n-2-i hold n-1-i and i+1 pairs, n together: they share an iteration (n/2 iterations with
    // n lanes busy instead of n*n/64 iterations of a half-empty square with a wave-wide reduction in each).
    double bd = -2.0;
    int bid = 0x7fffffff;
    const int half = n >> 1;  // rows 0..n-2 in pairs (i, n-2-i); the middle row of an odd number of rows alone
    for (int i = 0; i < half; i++) {
        const int i2 = n - 2 - i;
        const int na = n - 1 - i;                  // pairs of row i
        const int nb = (i2 > i) ? n - 1 - i2 : 0;  // pairs of row i2
        for (int l0 = 0; l0 < na + nb; l0 += kWave) {
            const int l = l0 + lane;
            int r = -1, j = 0;
            if (l < na) {
                r = i;
                j = i + 1 + l;
            } else if (l < na + nb) {
                r = i2;
                j = i2 + 1 + (l - na);
            }
            if (r >= 0) {
                const V3 pr = {L.x[r], L.y[r], L.z[r]}, pj = {L.x[j], L.y[j], L.z[j]};
                const double d = vsqnorm(vsub(pr, pj));
                const int id = r * n + j;
                if (d > bd || (d == bd && id < bid)) {
                    bd = d;
                    bid = id;
                }
            }
        }
    }
    const double best = wave_max_f64(bd);
    int cand_id = (bd == best) ? bid : 0x7fffffff;
    cand_id = uniform(wave_min_i32(cand_id));
    int bi = cand_id / n, bj = cand_id - (cand_id / n) * n;
    bi = uniform(bi);
    bj = uniform(bj);
    if (best <= 0.0) return false;  // :65
    V3 pa = {L.x[bi], L.y[bi], L.z[bi]}, pb = {L.x[bj], L.y[bj], L.z[bj]};
    double best2 = -1.0;
    int bk = -1;
    for (int b = 0; b < n - 1; b += kWave) {  // :71 last point never considered
        int kx = b + lane;
        bool valid = (kx < n - 1) && (kx != bi) && (kx != bj);
        double d = -2.0;
        if (valid) {
            V3 pk = {L.x[kx], L.y[kx], L.z[kx]};
            double d1 = vsqnorm(vsub(pk, pa));
            double d2 = vsqnorm(vsub(pk, pb));
            valid = !(d1 <= 0.0) && !(d2 <= 0.0);
            if (valid) d = d1 + d2;
        }
        double m = wave_max_f64(d);
        if (m > best2) {
            unsigned long long who = __ballot(valid && d == m);
            if (who) {
                int first = __ffsll((long long)who) - 1;
                best2 = m;
                bk = b + first;
            }
        }
    }
    bk = uniform(bk);
    if (bi == -1 || bj == -1 || bk == -1) return false;
    ci = bi;
    cj = bj;
    ck = bk;
    return true;
}

__device__ __forceinline__ void store_rec(double* rec, int f, int field, double v) { rec[field * kWave + f] = v; }
__device__ __forceinline__ double load_rec(const double* rec, int f, int field) { return rec[field * kWave + f]; }

// min / max of z over the first n list entries (TresholdDepthLocal.cpp:23-29)
__device__ void list_minmax_z(int n, const Lists& L, int lane, double& mn, double& mx) {
    double a = 1.7976931348623157e308, b = -1.7976931348623157e308;
    for (int base = 0; base < n; base += kWave) {
        int i = base + lane;
        if (i < n) {
            double z = L.z[i];
            if (z < a) a = z;
            if (z > b) b = z;
        }
    }
    mn = wave_min_f64(a);
    mx = wave_max_f64(b);
}

// Cyclic Jacobi eigen-decomposition of a symmetric 3x3 (a[0..5] = xx,xy,xz,yy,yz,zz).  Returns the
// eigenvalues ascending in ev[] and the eigenvector of the smallest in n0.
__device__ void jacobi_eig3(const double s[6], double ev[3], V3& n0) {
    double a[3][3] = {{s[0], s[1], s[2]}, {s[1], s[3], s[4]}, {s[2], s[4], s[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (!(off > 1e-300) || off <= 1e-32 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    double d0 = a[0][0], d1 = a[1][1], d2 = a[2][2];
    // stable ascending order of (d0,d1,d2) by index (matches std::sort on distinct values)
    int i0 = 0, i1 = 1, i2 = 2;
    if (d1 < d0) {
        double t = d0;
        d0 = d1;
        d1 = t;
        int ti = i0;
        i0 = i1;
        i1 = ti;
    }
    if (d2 < d1) {
        double t = d1;
        d1 = d2;
        d2 = t;
        int ti = i1;
        i1 = i2;
        i2 = ti;
        if (d1 < d0) {
            t = d0;
            d0 = d1;
            d1 = t;
            ti = i0;
            i0 = i1;
            i1 = ti;
        }
    }
    ev[0] = d0;
    ev[1] = d1;
    ev[2] = d2;
    n0.x = (i0 == 0) ? v[0][0] : ((i0 == 1) ? v[0][1] : v[0][2]);
    n0.y = (i0 == 0) ? v[1][0] : ((i0 == 1) ? v[1][1] : v[1][2]);
    n0.z = (i0 == 0) ? v[2][0] : ((i0 == 1) ? v[2][1] : v[2][2]);
}

// ---- reduced-cost arithmetic for the ROAD path only ------------------------------------------------------------
// The road / PCA results are held to the 1e-4 m tolerance, not to bit-exactness (their sums are reordered anyway), so
// the M-estimator replaces IEEE divisions and square roots (30-40 instructions each in f64) by the hardware
// reciprocal / reciprocal-square-root estimates refined with two Newton steps (relative error ~1e-16).
__device__ __forceinline__ double fast_rcp(double d) {
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    e = fma(-d, x, 1.0);
    x = fma(x, e, x);
    return x;
}
__device__ __forceinline__ double fast_rsq(double d) {  // d in (0, inf)
    double y = __builtin_amdgcn_rsq(d);
    double h = 0.5 * d;
    double e = fma(-h * y, y, 0.5);  // 0.5 - 0.5 d y^2
    y = fma(y, e, y);
    e = fma(-h * y, y, 0.5);
    y = fma(y, e, y);
    return y;
}

// Cyclic Jacobi as jacobi_eig3 below, with the rotation angles from fast_rcp / fast_rsq: the rotations stay
// orthogonal to rounding, the sweeps converge as before.
__device__ void jacobi_eig3_fast(const double s[6], double ev[3], V3& n0) {
    double a[3][3] = {{s[0], s[1], s[2]}, {s[1], s[3], s[4]}, {s[2], s[4], s[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (!(off > 1e-300) || off <= 1e-30 * diag) break;
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
            for (int q = p + 1; q < 3; q++) {
                const double apq = a[p][q];
                const bool rot = apq != 0.0;
                const double theta = (a[q][q] - a[p][p]) * 0.5 * fast_rcp(rot ? apq : 1.0);
                const double r = fabs(theta);
                // sqrt(theta^2 + 1) = |theta| to double precision beyond 1e8 (and theta^2 may overflow)
                const double x = fma(r < 1e100 ? r : 1.0, r < 1e100 ? r : 1.0, 1.0);
                const double root = r < 1e100 ? x * fast_rsq(x) : r;
                double t = fast_rcp(r + root);
                t = theta >= 0 ? t : -t;
                double cs = fast_rsq(fma(t, t, 1.0)), sn = t * cs;
                cs = rot ? cs : 1.0;
                sn = rot ? sn : 0.0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    double d0 = a[0][0], d1 = a[1][1], d2 = a[2][2];
    int i0 = 0;
    if (d1 < d0) {
        d0 = d1;
        i0 = 1;
    }
    if (d2 < d0) {
        d0 = d2;
        i0 = 2;
    }
    // (smallest first; the other two as the middle and the largest)
    const double e1 = (i0 == 0) ? a[1][1] : a[0][0], e2 = (i0 == 2) ? a[1][1] : a[2][2];
    ev[0] = d0;
    ev[1] = e1 < e2 ? e1 : e2;
    ev[2] = e1 < e2 ? e2 : e1;
    n0.x = (i0 == 0) ? v[0][0] : ((i0 == 1) ? v[0][1] : v[0][2]);
    n0.y = (i0 == 0) ? v[1][0] : ((i0 == 1) ? v[1][1] : v[1][2]);
    n0.z = (i0 == 0) ? v[2][0] : ((i0 == 1) ? v[2][1] : v[2][2]);
}

// Eigenvector of the SMALLEST eigenvalue of a symmetric positive semi-definite 3x3 (s = xx,xy,xz,yy,yz,zz) without
// sweeps: the smallest root of the characteristic cubic by Newton from 0 (the cubic is increasing and concave below
// its first root, so the iterates rise monotonically to it), a first vector from the cross products of the rows of
// A - lambda I, then two Rayleigh-quotient steps through the adjugate (cubic convergence; they repair the cancellation
// in the cubic's coefficients).  ~300 instructions against ~2000 for the Jacobi sweeps of a wavefront.  Returns false
// when the two smallest eigenvalues are closer than 1e-6 of the largest (or the input is not finite): the caller then
// takes the Jacobi solver.  gap2 = ((lambda2 - lambda1)(lambda3 - lambda1) / trace^2)^2.  Measured against LAPACK on 40 000 road-like weighted
// scatters: eigenvector error <= 1e-10 wherever it returns true.
__device__ __forceinline__ bool smallest_eigvec_sym3(const double s[6], V3& n0, double& gap2) {
    const double tr = s[0] + s[3] + s[5];
    const double sc = fast_rcp(tr);
    const double a00 = s[0] * sc, a01 = s[1] * sc, a02 = s[2] * sc, a11 = s[3] * sc, a12 = s[4] * sc, a22 = s[5] * sc;
    const double c2 = a00 + a11 + a22;
    const double c1 = (a00 * a11 - a01 * a01) + (a00 * a22 - a02 * a02) + (a11 * a22 - a12 * a12);
    const double c0 = a00 * (a11 * a22 - a12 * a12) - a01 * (a01 * a22 - a12 * a02) + a02 * (a01 * a12 - a11 * a02);
    double l = 0.0;
#pragma unroll
    for (int it = 0; it < 8; it++) {
        const double q = fma(fma(l - c2, l, c1), l, -c0);
        const double dq = fma(fma(3.0, l, -2.0 * c2), l, c1);
        l = fma(-q, fast_rcp(dq), l);
    }
    // rows of B = A - l I and their cross products; the largest one spans the null direction best
    const V3 r0 = {a00 - l, a01, a02}, r1 = {a01, a11 - l, a12}, r2 = {a02, a12, a22 - l};
    const V3 x01 = vcross(r0, r1), x02 = vcross(r0, r2), x12 = vcross(r1, r2);
    const double m01 = vsqnorm(x01), m02 = vsqnorm(x02), m12 = vsqnorm(x12);
    V3 v = x01;
    double m = m01;
    if (m02 > m) {
        v = x02;
        m = m02;
    }
    if (m12 > m) {
        v = x12;
        m = m12;
    }
    v = vscale(v, fast_rsq(m > 0.0 ? m : 1.0));
    double g2 = 0.0;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const V3 av = {a00 * v.x + (a01 * v.y + a02 * v.z), a01 * v.x + (a11 * v.y + a12 * v.z),
                       a02 * v.x + (a12 * v.y + a22 * v.z)};
        const double lam = vdot(v, av);
        const double b00 = a00 - lam, b11 = a11 - lam, b22 = a22 - lam;
        // adjugate of the symmetric B = A - lam I
        const double d00 = b11 * b22 - a12 * a12, d01 = a02 * a12 - a01 * b22, d02 = a01 * a12 - a02 * b11;
        const double d11 = b00 * b22 - a02 * a02, d12 = a01 * a02 - b00 * a12, d22 = b00 * b11 - a01 * a01;
        const V3 w = {d00 * v.x + (d01 * v.y + d02 * v.z), d01 * v.x + (d11 * v.y + d12 * v.z),
                      d02 * v.x + (d12 * v.y + d22 * v.z)};
        g2 = vsqnorm(w);
        v = vscale(w, fast_rsq(g2 > 0.0 ? g2 : 1.0));
    }
    n0 = v;
    // |adj(B) v| ~ (lambda2 - lambda1)(lambda3 - lambda1) in units of the trace
    gap2 = g2;
    return (tr > 0.0) && (g2 > 1e-12) && isfinite(g2);
}

struct Plane {
    V3 n;
    double offset;
};

// Eigen::Hyperplane<double,3>::Through (call site LinePlaneIntersectionBase.cpp:41).
__device__ Plane plane_through(V3 p0, V3 p1, V3 p2) {
    V3 v0 = vsub(p2, p0), v1 = vsub(p1, p0);
    V3 n = vcross(v0, v1);
    double nn = vnorm(n);
    Plane r;
    if (nn <= vnorm(v0) * vnorm(v1) * 2.220446049250313e-16) {
        // degenerate triangle: smallest eigenvector of m^T m, m = [v0^T; v1^T]
        double s[6] = {v0.x * v0.x + v1.x * v1.x, v0.x * v0.y + v1.x * v1.y, v0.x * v0.z + v1.x * v1.z,
                       v0.y * v0.y + v1.y * v1.y, v0.y * v0.z + v1.y * v1.z, v0.z * v0.z + v1.z * v1.z};
        double ev[3];
        jacobi_eig3(s, ev, r.n);
    } else {
        r.n = vdivs(n, nn);
    }
    r.offset = -vdot(p0, r.n);
    return r;
}

// CameraPinhole::getViewingRays (camera_pinhole.h:52-69) + flip (DepthEstimator.cpp:938-939)
__device__ V3 viewing_ray(const Calib& c, double u, double v) {
    V3 d = {(c.Kinv[0] * u + c.Kinv[1] * v) + c.Kinv[2], (c.Kinv[3] * u + c.Kinv[4] * v) + c.Kinv[5],
            (c.Kinv[6] * u + c.Kinv[7] * v) + c.Kinv[8]};
    d = vnormalized(d);
    if (d.z < 0) d = vscale(d, -1.0);
    return d;
}

// LinePlaneIntersection{OrthogonalTreshold,Normal}::GetIntersection
__device__ bool intersect(const Calib& c, bool orth, Plane pl, V3 n0, V3 n1, double& depth) {
    if (orth) {
        V3 ln = vnormalized(n1), pn = vnormalized(pl.n);
        if (!(fabs(vdot(pn, ln)) >= c.orthThr)) return false;
    }
    V3 dir = vnormalized(vsub(n1, n0));  // ParametrizedLine::Through(n0, n1)
    double t = -(pl.offset + vdot(pl.n, n0)) / vdot(pl.n, dir);
    V3 p = vadd(n0, vscale(dir, t));
    depth = p.z;
    return true;
}

// TresholdDepthGlobal::CheckInDepth / TresholdDepthLocal::CheckInBounds, then the caller's mapping to
// result types.  Returns 0 when in bounds (depth possibly adjusted) or the failing result type.
__device__ int apply_thresholds(const Calib& c, double minZ, double maxZ, double& depth) {
    if (c.thrG_en) {
        if (depth < c.thrG_min) {
            if (c.thrG_mode == 0) return MLD_TresholdDepthGlobalSmallerMin;
            depth = c.thrG_min;
        } else if (depth > c.thrG_max) {
            if (c.thrG_mode == 0) return MLD_TresholdDepthGlobalGreaterMax;
            depth = c.thrG_max;
        }
    }
    if (c.thrL_en) {
        double interval = maxZ - minZ, lo, hi;
        if (c.thrL_type == 1) {
            double r = interval * c.thrL_val;
            lo = minZ - r;
            hi = maxZ + r;
        } else {
            lo = minZ - c.thrL_val;
            hi = maxZ + c.thrL_val;
        }
        if (depth < lo) {
            if (c.thrL_mode == 0) return MLD_TresholdDepthLocalSmallerMin;
            depth = lo;
        } else if (depth > hi) {
            if (c.thrL_mode == 0) return MLD_TresholdDepthLocalGreaterMax;
            depth = hi;
        }
    }
    return 0;
}

// Weighted / unweighted first and second moments over the first n list entries whose flag is set
// (flag == nullptr: all).  Used by the M-estimator (weights 1/|prior distance|) and PCA (weights 1).
// Tree reductions across lanes: the summation ORDER differs from the serial CPU loops (results agree to
// rounding; parity tolerance for these paths is 1e-4 m, not bit-exact).
__device__ void moments(int n, const Lists& L, int lane, bool weighted, const SlotDesc& s, double center[3],
                        double cov[6], bool divide_by_n) {
    double sw = 0, sx = 0, sy = 0, sz = 0;
    for (int b = 0; b < n; b += kWave) {
        int i = b + lane;
        if (i < n) {
            V3 p = {L.x[i], L.y[i], L.z[i]};
            double w = 1.0;
            if (weighted) {
                V3 pn = {s.prior_n[0], s.prior_n[1], s.prior_n[2]};
                w = 1 / fabs(vdot(pn, p) + s.prior_off);  // PlaneEstimationMEstimator.cpp:32
            }
            sw += w;
            sx += w * p.x;
            sy += w * p.y;
            sz += w * p.z;
        }
    }
    sw = wave_sum_f64(sw);
    sx = wave_sum_f64(sx);
    sy = wave_sum_f64(sy);
    sz = wave_sum_f64(sz);
    double den = divide_by_n ? (double)n : sw;
    double cx = sx / den, cy = sy / den, cz = sz / den;
    double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    for (int b = 0; b < n; b += kWave) {
        int i = b + lane;
        if (i < n) {
            V3 p = {L.x[i], L.y[i], L.z[i]};
            double w = 1.0;
            if (weighted) {
                V3 pn = {s.prior_n[0], s.prior_n[1], s.prior_n[2]};
                w = 1 / fabs(vdot(pn, p) + s.prior_off);
            }
            double dx = p.x - cx, dy = p.y - cy, dz = p.z - cz;
            c0 += w * dx * dx;
            c1 += w * dx * dy;
            c2 += w * dx * dz;
            c3 += w * dy * dy;
            c4 += w * dy * dz;
            c5 += w * dz * dz;
        }
    }
    center[0] = cx;
    center[1] = cy;
    center[2] = cz;
    cov[0] = wave_sum_f64(c0);
    cov[1] = wave_sum_f64(c1);
    cov[2] = wave_sum_f64(c2);
    cov[3] = wave_sum_f64(c3);
    cov[4] = wave_sum_f64(c4);
    cov[5] = wave_sum_f64(c5);
}

// PlaneEstimationMEstimator::EstimatePlane (:18-55) up to the decomposition: the weighted centre and the matrix
// M = [sqrt(w_i) (p_i - c)] whose left singular vector of the smallest singular value is the plane's normal (JacobiSVD, :49).
// Road points seen through a narrow window are nearly COLLINEAR (returns of neighbouring rings at one azimuth): the two
// small singular values then differ by the coordinates' rounding noise, and the scatter matrix M M^T - which squares the
// singular values - no longer holds the normal at all (estimates 0.3 ... 4 mm off, LAB.md 5.33).  So the wavefront takes
// the thin QR of M^T (k x 3, rows in LDS over the list itself; modified Gram-Schmidt, whose R is that of a matrix within
// rounding of M^T) and hands on R: M M^T = R^T R, and a one-sided Jacobi on the three rows of R^T (finish_road) is as
// accurate as the reference's SVD of M.  out: center[3], R = r11 r12 r13 r22 r23 r33.
__device__ void road_qr(int n, const Lists& L, int lane, const SlotDesc& s, double center[3], double R[6]) {
    const V3 pn = {s.prior_n[0], s.prior_n[1], s.prior_n[2]};
    double sw = 0, sx = 0, sy = 0, sz = 0;
    for (int b = 0; b < n; b += kWave) {
        const int i = b + lane;
        if (i < n) {
            const V3 p = {L.x[i], L.y[i], L.z[i]};
            const double w = 1 / fabs(vdot(pn, p) + s.prior_off);  // PlaneEstimationMEstimator.cpp:32
            sw += w;
            sx += w * p.x;
            sy += w * p.y;
            sz += w * p.z;
        }
    }
    sw = wave_sum_f64(sw);
    sx = wave_sum_f64(sx);
    sy = wave_sum_f64(sy);
    sz = wave_sum_f64(sz);
    const double cx = sx / sw, cy = sy / sw, cz = sz / sw;
    center[0] = cx;
    center[1] = cy;
    center[2] = cz;
    // rows a_i = sqrt(w_i) (p_i - c), written over the list; first column's norm on the way
    double t0 = 0;
    for (int b = 0; b < n; b += kWave) {
        const int i = b + lane;
        if (i < n) {
            const V3 p = {L.x[i], L.y[i], L.z[i]};
            const double ws = sqrt(1 / fabs(vdot(pn, p) + s.prior_off));
            const double ax = ws * (p.x - cx), ay = ws * (p.y - cy), az = ws * (p.z - cz);
            L.x[i] = ax;
            L.y[i] = ay;
            L.z[i] = az;
            t0 += ax * ax;
        }
    }
    const double r11 = sqrt(wave_sum_f64(t0));
    const double i11 = r11 > 0.0 ? 1 / r11 : 0.0;  // (all points in one place: a zero column stays zero)
    double t1 = 0, t2 = 0;
    for (int b = 0; b < n; b += kWave) {
        const int i = b + lane;
        if (i < n) {
            const double q = L.x[i] * i11;
            L.x[i] = q;
            t1 += q * L.y[i];
            t2 += q * L.z[i];
        }
    }
    const double r12 = wave_sum_f64(t1), r13 = wave_sum_f64(t2);
    t0 = 0;
    for (int b = 0; b < n; b += kWave) {
        const int i = b + lane;
        if (i < n) {
            const double q = L.x[i];
            const double ay = L.y[i] - r12 * q;
            L.y[i] = ay;
            L.z[i] = L.z[i] - r13 * q;
            t0 += ay * ay;
        }
    }
    const double r22 = sqrt(wave_sum_f64(t0));
    const double i22 = r22 > 0.0 ? 1 / r22 : 0.0;
    t1 = 0;
    for (int b = 0; b < n; b += kWave) {
        const int i = b + lane;
        if (i < n) {
            const double q = L.y[i] * i22;
            L.y[i] = q;
            t1 += q * L.z[i];
        }
    }
    const double r23 = wave_sum_f64(t1);
    t0 = 0;
    for (int b = 0; b < n; b += kWave) {
        const int i = b + lane;
        if (i < n) {
            const double az = L.z[i] - r23 * L.y[i];
            t0 += az * az;
        }
    }
    const double r33 = sqrt(wave_sum_f64(t0));
    R[0] = r11;
    R[1] = r12;
    R[2] = r13;
    R[3] = r22;
    R[4] = r23;
    R[5] = r33;
}

// Left singular vector of the smallest singular value of M from the R of road_qr: one-sided (Hestenes) Jacobi on the
// three rows of R^T - M = R^T Q^T, so the rows' inner products are M's - as the CPU restatement of the reference runs
// it on the rows of M itself.
__device__ __forceinline__ V3 normal_from_R(const double R[6]) {
    double B[3][3] = {{R[0], 0.0, 0.0}, {R[1], R[3], 0.0}, {R[2], R[4], R[5]}};
    double U[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 60; sweep++) {
        bool rotated = false;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                const double app = B[p][0] * B[p][0] + B[p][1] * B[p][1] + B[p][2] * B[p][2];
                const double aqq = B[q][0] * B[q][0] + B[q][1] * B[q][1] + B[q][2] * B[q][2];
                const double apq = B[p][0] * B[q][0] + B[p][1] * B[q][1] + B[p][2] * B[q][2];
                if (!(fabs(apq) > 1e-15 * sqrt(app * aqq)) || apq == 0.0) continue;
                rotated = true;
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const double a = B[p][k], b = B[q][k];
                    B[p][k] = cs * a - sn * b;
                    B[q][k] = sn * a + cs * b;
                    const double ua = U[k][p], ub = U[k][q];
                    U[k][p] = cs * ua - sn * ub;
                    U[k][q] = sn * ua + cs * ub;
                }
            }
        if (!rotated) break;
    }
    double nrm[3];
#pragma unroll
    for (int r = 0; r < 3; r++) nrm[r] = B[r][0] * B[r][0] + B[r][1] * B[r][1] + B[r][2] * B[r][2];
    int m = 0;
    if (nrm[1] < nrm[m]) m = 1;
    if (nrm[2] < nrm[m]) m = 2;
    V3 n0;
    n0.x = (m == 0) ? U[0][0] : ((m == 1) ? U[0][1] : U[0][2]);
    n0.y = (m == 0) ? U[1][0] : ((m == 1) ? U[1][1] : U[1][2]);
    n0.z = (m == 0) ? U[2][0] : ((m == 1) ? U[2][1] : U[2][2]);
    return n0;
}

enum : int {
    ST_FINAL = 0,       // mytype/mydepth are final
    ST_TRIANGLE = 1,    // record = 3 corners + minZ,maxZ           -> phase 2
    ST_PCA = 2,         // record = mean, cov, minZ,maxZ            -> phase 2
    ST_ROAD_MEST = 3,   // record = centre, R (road_qr), minZ,maxZ  -> phase 4
    ST_ROAD_TRI = 4     // record = 3 corners + minZ,maxZ           -> phase 4
};

// Tail of CalculateDepthSegmented (DepthEstimator.cpp:928-1036), one feature per lane.
//   r[0..8]: three corners (triangle) or mean + scatter xx,xy,xz,yy,yz,zz (PCA); r[9], r[10]: min / max z of the
//   segmented points (TresholdDepthLocal).
__device__ __forceinline__ void finish_main(const Calib& c, bool pca, double u, double v, const double r[kRecFields], int& out_type,
                            double& out_depth) {
    const bool orth = c.orthThr > 0;  // DepthEstimator.cpp:77-81
    int type = MLD_Success;
    double depth = -1.0;
    V3 dir = viewing_ray(c, u, v);
    V3 support = {0, 0, 0};
    if (!pca) {
        V3 c1 = {r[0], r[1], r[2]}, c2 = {r[3], r[4], r[5]}, c3 = {r[6], r[7], r[8]};
        if (c.checkPlanar) {  // PlaneEstimationCheckPlanar.cpp:18-44
            V3 e1 = vnormalized(vsub(c2, c1)), e2 = vnormalized(vsub(c3, c1)), e3 = vnormalized(vsub(c3, c2));
            double l12 = vnorm(vcross(e1, e2)), l13 = vnorm(vcross(e1, e3)), l23 = vnorm(vcross(e2, e3));
            if (!((l12 >= c.planarThr) && (l13 >= c.planarThr) && (l23 >= c.planarThr))) type = MLD_TriangleNotPlanar;
        }
        if (type == MLD_Success) {
            Plane pl = plane_through(c1, c2, c3);
            if (!intersect(c, orth, pl, support, dir, depth)) type = MLD_PlaneViewrayNotOrthogonal;
        }
    } else {
        // Mono_LidarPipeline::PCA (PCA.cpp:11-62)
        double ev[3];
        V3 n0;
        jacobi_eig3(&r[3], ev, n0);
        n0 = vdivs(n0, vnorm(n0));
        float planarity = (float)((ev[1] - ev[0]) / ev[2]);
        float linearity = (float)((ev[2] - ev[1]) / ev[2]);
        if ((double)planarity < c.pcaRelMin)
            type = MLD_PcaIsCubic;
        else if ((double)linearity > c.pcaRelMax)
            type = MLD_PcaIsLine;
        else if (ev[2] < c.pcaAbsMin)
            type = MLD_PcaIsPoint;
        if (type == MLD_Success) {
            V3 mean = {r[0], r[1], r[2]};
            Plane pl = {n0, -vdot(n0, mean)};
            if (!intersect(c, orth, pl, support, dir, depth)) type = MLD_PlaneViewrayNotOrthogonal;
        }
    }
    if (type == MLD_Success) {
        int t = apply_thresholds(c, r[9], r[10], depth);
        if (t) type = t;
    }
    if (type == MLD_Success && depth < 0 && c.cutBehind) type = MLD_CornerBehindCamera;
    out_type = type;
    out_depth = (type == MLD_Success) ? depth : -1.0;
}

// Tail of the road estimators (RoadDepthEstimatorMEstimator.cpp:28-74, RoadDepthEstimatorMaxSpanningTriangle.cpp:
// 24-75), one feature per lane.  r[0..8]: weighted centre + weighted scatter (M-estimator) or three corners.
__device__ __forceinline__ void finish_road(const Calib& c, bool triangle, double u, double v, const double r[kRecFields], int& out_type,
                            double& out_depth) {
    V3 dir = viewing_ray(c, u, v);
    V3 support = {0, 0, 0};
    Plane pl;
    if (!triangle) {
        // PlaneEstimationMEstimator::EstimatePlane (:18-55): direction of least weighted variance
        // (r[3..8] = the R of road_qr, not the scatter: see there)
        V3 center = {r[0], r[1], r[2]};
        V3 n0 = normal_from_R(&r[3]);
        if (isnan(center.x) || isnan(center.y) || isnan(center.z)) {
            double qn = __builtin_nan("");
            n0 = {qn, qn, qn};
        }
        n0 = vnormalized(n0);
        pl.n = n0;
        pl.offset = -vdot(n0, center);
    } else {
        pl = plane_through({r[0], r[1], r[2]}, {r[3], r[4], r[5]}, {r[6], r[7], r[8]});
    }
    double depth = -1.0;
    intersect(c, false, pl, dir, support, depth);  // n0 = direction, n1 = support (swapped, as the reference)
    int type = MLD_SuccessRoad;
    int t = apply_thresholds(c, r[9], r[10], depth);
    if (t) type = t;
    out_type = type;
    out_depth = (type == MLD_SuccessRoad) ? depth : -1.0;
}

// The M-estimator tail of finish_road with the reduced-cost arithmetic (thread path; results within the road
// tolerance of the exact form).  r[0..2]: weighted centre, r[3..8]: weighted scatter, r[9], r[10]: min / max z.
// Returns false where the scatter cannot carry the normal to the road tolerance: the eigenvector's error is the sums'
// rounding (a few 1e-15 of the trace) over the gap between the two small eigenvalues - nearly collinear points: returns of
// one ring, of one azimuth - and the depth multiplies it by depth / cos(ray, normal).  Where that estimate could change
// the OUTCOME - a depth handed out with more than 2e-5 m of it, a threshold decision within ten times of it - or the input
// is not finite, the caller hands the feature to the wave kernel (road_qr: SVD accuracy).  LAB.md 5.33, 5.34.
__device__ __forceinline__ bool finish_road_fast(const Calib& c, double u, double v, const double r[kRecFields],
                                                 const double sw, const double cdev, int& out_type, double& out_depth) {
    const V3 dir = viewing_ray(c, u, v);
    const V3 support = {0, 0, 0};
    const V3 center = {r[0], r[1], r[2]};
    V3 n0;
    double gap2 = 0.0;
    const bool direct = smallest_eigvec_sym3(&r[3], n0, gap2);
    double gap = sqrt(gap2);
    if (__any(!direct)) {  // (nearly) equal small eigenvalues somewhere in the wavefront: the sweeps for those lanes
        double ev[3];
        V3 nj;
        jacobi_eig3_fast(&r[3], ev, nj);
        if (!direct) {
            n0 = nj;
            gap = (ev[1] - ev[0]) / ev[2];  // (NaN / negative noise: the estimate below fails, as it should)
        }
    }
    bool well = true;
    if (isnan(center.x) || isnan(center.y) || isnan(center.z)) {
        const double qn = __builtin_nan("");
        n0 = {qn, qn, qn};
    }
    Plane pl;
    pl.n = vnormalized(n0);
    pl.offset = -vdot(pl.n, center);
    // the intersection itself keeps the IEEE operations of the exact form: a plane through the camera centre gives
    // the depth 0.0 exactly there, which sits ON the global threshold (treshold_depth_min = 0)
    double depth = -1.0;
    intersect(c, false, pl, dir, support, depth);  // n0 = direction, n1 = support (swapped, as the reference)
    // (error estimate: see the head comment; a ray nearly parallel to the plane or an intersection kilometres away -
    // depth thresholds off - fail it as well as a vanishing gap)
    const double nd = fabs(vdot(pl.n, dir));
#ifndef MLD_ROAD_ERR_C
#define MLD_ROAD_ERR_C 1e-14
#endif
    const double raw = depth;
    const double den = gap * nd;
    // The sums' rounding relative to the trace: a few 1e-15 from the reciprocal estimates of the update, and - West's
    // update subtracts a ROUNDED running mean from every point - 2^-53 |mean| per deviation, i.e. up to
    // ~4 * 2^-53 |mean| sum(w |dev|) <= 4 * 2^-53 |mean| sqrt(sw * trace) in the scatter.  With the sums taken about the
    // camera's origin, returns 75 m away that are collinear to 0.3 mm (three returns of one azimuth) carried 7e-14 of the
    // trace, not 1e-14 (profiles/tools/random_sweep.py, seeds 112686 and 110803 on the lane-per-feature route: 1.1e-4 m
    // and 3.5e-5 m with estimates of 1.5e-5 and 8e-6; LAB.md 6.24).  The sums are therefore taken about the list's first
    // point (road_after_scan) - |mean| is then the set's own extent, cdev - and twice that bound enters the estimate
    // (sw = sum of the weights); it matters where the origin is not an inlier (the general loop, a far object first).
    const double errc = fmax(MLD_ROAD_ERR_C, 8.8817841970012523e-16 * cdev * sqrt(sw / (r[3] + r[6] + r[8])));
    // (est = 0 only for a clean estimate; a vanishing or NaN denominator gives +inf / NaN, which fails every test below)
    const double est = fabs(raw) * errc / den;
    int type = MLD_SuccessRoad;
    const int th = apply_thresholds(c, r[9], r[10], depth);
    if (th) type = th;
    out_type = type;
    out_depth = (type == MLD_SuccessRoad) ? depth : -1.0;
    // The depth itself is only handed out where no threshold touched it: there it must be good to 2e-5 m.  Where a
    // threshold rejected or clamped it, the outcome (a result type, or the bound) is exact as long as the estimate stays
    // clear of every bound by ten times its possible error - most ill-conditioned fits end kilometres away and are
    // rejected by thresholds they miss by far; those need no second opinion.
    if (type == MLD_SuccessRoad && depth == raw) {
        well = est <= 2e-5;
    } else {
        const double m = 10.0 * est;
        bool clear = m == m && m < 1.7976931348623157e308;  // (finite)
        if (c.thrG_en) clear = clear && fabs(raw - c.thrG_min) > m && fabs(raw - c.thrG_max) > m;
        if (c.thrL_en) {
            const double tol = (c.thrL_type == 1) ? (r[10] - r[9]) * c.thrL_val : c.thrL_val;
            // (after a global clamp the local test sees the bound, which is exact; before one, the raw depth)
            clear = clear && fabs(raw - (r[9] - tol)) > m && fabs(raw - (r[10] + tol)) > m;
        }
        well = clear;
    }
    return well;
}

// Wave-cooperative path: processes the features whose bit is set in `mask` (wave-uniform), one at a time, with
// all 64 lanes working on that feature's window / neighbour list.  Handles lists of any length (up to the
// window size); used for the features the thread-per-feature path cannot hold (long lists).
#define SREC(t, v)                   \
    do {                             \
        double v_ = (v);             \
        if (lane == fi) myr[t] = v_; \
    } while (0)

__device__ __forceinline__ void wave_path(const Calib& c, const SlotDesc& s, unsigned char* smem, const int lane,
                                       const unsigned long long mask, const double myu, const double myv, int& mytype,
                                       double& mydepth, const unsigned long long main_mask, double (&corners)[9],
                                       bool& has_corners ST_ARG) {
    Lists L;
    L.x = reinterpret_cast<double*>(smem);
    L.y = L.x + c.cap;
    L.z = L.y + c.cap;
    L.idx = reinterpret_cast<int*>(L.z + c.cap);
    L.bin = L.idx + c.cap;
    double myr[kRecFields];  // record of the feature this lane owns (values below are wave-uniform)
#pragma unroll
    for (int t = 0; t < kRecFields; t++) myr[t] = 0.0;
    const bool inmask = (mask >> lane) & 1ull;
    int mystate = ST_FINAL;

    // ---------------- phase 1 ----------------
    // (lanes of `mask` outside `main_mask` already went through the main path; mytype holds its result)
    for (unsigned long long m1 = main_mask; m1; m1 &= m1 - 1) {
        const int fi = __ffsll((long long)m1) - 1;
        const double u = readlane_f64(myu, fi), v = readlane_f64(myv, fi);
        int state = ST_FINAL, type = MLD_Unspecified;
        int k = gather_window(c, s, u, v, c.halfX1, c.halfY1, L, lane);
        ST_USE_U32(k);
        ST_MARK(1);
        if ((unsigned)k < c.countMin) {  // DepthEstimator.cpp:680
            type = MLD_RadiusSearchInsufficientPoints;
        } else {
            int ks = k;
            if (c.useHist) ks = hist_segment(c, k, L, lane);  // DepthEstimator.cpp:726-780
            ST_USE_U32(ks);
            ST_MARK(2);
            if (ks < 0) {
                type = MLD_HistogramNoLocalMax;
            } else {
                // CalculateDepthSegmented, corner selection (DepthEstimator.cpp:915-926)
                int ci = 0, cj = 1, ck = 2;
                bool ok = true;
                if (!c.usePCA && c.useTriMax) {
                    ok = max_spanning_triangle(ks, L, lane, ci, cj, ck);
                    ST_USE_U32(ci);
                    ST_MARK(3);
                    if (!ok) type = MLD_TriangleNotPlanarInsufficientPoints;
                } else if (ks < 3) {
                    ok = false;
                    type = MLD_HistogramNoLocalMax;
                }
                if (ok) {
                    double mn, mx;
                    list_minmax_z(ks, L, lane, mn, mx);
                    if (c.usePCA) {
                        double ctr[3], cov[6];
                        moments(ks, L, lane, false, s, ctr, cov, true);
                        {
                            for (int t = 0; t < 3; t++) SREC(t, ctr[t]);
                            for (int t = 0; t < 6; t++) SREC(3 + t, cov[t]);
                        }
                        state = ST_PCA;
                    } else {
                        {
                            SREC(0, L.x[ci]);
                            SREC(1, L.y[ci]);
                            SREC(2, L.z[ci]);
                            SREC(3, L.x[cj]);
                            SREC(4, L.y[cj]);
                            SREC(5, L.z[cj]);
                            SREC(6, L.x[ck]);
                            SREC(7, L.y[ck]);
                            SREC(8, L.z[ck]);
                        }
                        state = ST_TRIANGLE;
                    }
                    {
                        SREC(9, mn);
                        SREC(10, mx);
                    }
                }
            }
        }
        if (lane == fi) {
            mystate = state;
            mytype = type;
            mydepth = -1.0;
        }
    }

    ST_MARK(4);
    // ---------------- phase 2 (lane = feature) ----------------
    if (mystate == ST_TRIANGLE || mystate == ST_PCA) {
        // debug vector of CalculatePlaneCorners (PlaneEstimationCalcMaxSpanningTriangle.cpp:20-35)
        has_corners = (mystate == ST_TRIANGLE) && !c.usePCA && c.useTriMax;
#pragma unroll
        for (int t = 0; t < 9; t++) corners[t] = myr[t];
        finish_main(c, mystate == ST_PCA, myu, myv, myr, mytype, mydepth);
        mystate = ST_FINAL;
    }

    ST_USE_F64(mydepth);
    ST_MARK(5);
    // ---------------- phase 3: road fallback (DepthEstimator.cpp:578-597) ----------------
    // Candidates: everything that is not Success and did not already return at :509-510.
    const bool road_on = c.useRoad && s.has_plane && s.inlier_mask;
    bool cand = road_on && inmask && (mytype != MLD_Success) && (mytype != MLD_RadiusSearchInsufficientPoints);
    unsigned long long cm = __ballot(cand);
    while (cm) {
        const int fi = __ffsll((long long)cm) - 1;
        cm &= cm - 1;
        const double u = readlane_f64(myu, fi), v = readlane_f64(myv, fi);
        const int resultOld = __builtin_amdgcn_readlane(mytype, fi);
        int state = ST_FINAL, type = resultOld;
        // With the points' ground-plane state in the keys (the plane was known at projection time) the window is first
        // read through the keys alone: a far neighbour or fewer than three inliers settle the feature without a single
        // point fetch (:591), otherwise only the inliers are gathered - in the same order as the general loop below
        // compacts them, so the estimator sees the same list.
        bool by_state = false, anyFar = false;
        int k = 0, kk = 0;
        if (s.mask_in_key) {
            bool unsure = false;
            int k_inl = 0;
            window_states(c, s, u, v, c.halfX2, c.halfY2, lane, k, k_inl, anyFar, unsure);
            by_state = !unsure;
            if (by_state && (unsigned)k >= c.countMin && !anyFar && k_inl >= 3)
                kk = gather_window(c, s, u, v, c.halfX2, c.halfY2, L, lane, true);
        }
        if (!by_state) k = gather_window(c, s, u, v, c.halfX2, c.halfY2, L, lane);  // :585 scale 2.0, 1.5
        ST_USE_U32(k);
        ST_MARK(6);
        if ((unsigned)k < c.countMin) {
            type = MLD_RadiusSearchInsufficientPoints;
        } else {
            // CalculateDepthSegmentationPlane (DepthEstimator.cpp:782-900)
            for (int b = 0; !by_state && b < k; b += kWave) {
                int i = b + lane;
                bool far = false, inl = false;
                double x = 0, y = 0, z = 0;
                int id = 0;
                if (i < k) {
                    x = L.x[i];
                    y = L.y[i];
                    z = L.z[i];
                    id = L.idx[i];
                    // :810 T_lidar_cam * p (fixed-size product order), :811 float, :812 float plane distance
                    double xl = c.Tinv[3] + (c.Tinv[0] * x + (c.Tinv[1] * y + c.Tinv[2] * z));
                    double yl = c.Tinv[7] + (c.Tinv[4] * x + (c.Tinv[5] * y + c.Tinv[6] * z));
                    double zl = c.Tinv[11] + (c.Tinv[8] * x + (c.Tinv[9] * y + c.Tinv[10] * z));
                    float xf = (float)xl, yf = (float)yl, zf = (float)zl;
                    float d = fabsf(s.coeffs[0] * xf + s.coeffs[1] * yf + s.coeffs[2] * zf + s.coeffs[3]);
                    far = (double)d > c.roadDistThr;
                    inl = (GPTR(uint32_t, s.inlier_mask)[id >> 5] >> (id & 31)) & 1u;
                }
                anyFar = anyFar || (__ballot(far) != 0ull);
                unsigned long long m = __ballot(inl);
                int rank = kk + prefix_count(m);
                if (inl) {
                    L.x[rank] = x;
                    L.y[rank] = y;
                    L.z[rank] = z;
                    L.idx[rank] = id;
                }
                kk += __popcll(m);
            }
            kk = uniform(kk);
            ST_USE_U32(kk);
            ST_MARK(7);
            if (anyFar || kk < 3) {
                type = resultOld;  // :591
            } else {
                double mn, mx;
                list_minmax_z(kk, L, lane, mn, mx);
                if (c.roadMode == 0) {
                    double ctr[3], R[6];
                    road_qr(kk, L, lane, s, ctr, R);  // (rewrites the list: nothing reads it afterwards)
                    {
                        for (int t = 0; t < 3; t++) SREC(t, ctr[t]);
                        for (int t = 0; t < 6; t++) SREC(3 + t, R[t]);
                    }
                    state = ST_ROAD_MEST;
                } else {
                    // RoadDepthEstimatorMaxSpanningTriangle::CalculateDepth (:24-75)
                    int ci, cj, ck;
                    if (!max_spanning_triangle(kk, L, lane, ci, cj, ck)) {
                        type = MLD_RadiusSearchInsufficientPoints;
                    } else {
                        // LinePlaneIntersectionCeckXZTreshold::Check
                        double ax = 1.7976931348623157e308, bx = -1.7976931348623157e308;
                        for (int b = 0; b < kk; b += kWave) {
                            int i = b + lane;
                            if (i < kk) {
                                double x = L.x[i];
                                if (x < ax) ax = x;
                                if (x > bx) bx = x;
                            }
                        }
                        ax = wave_min_f64(ax);
                        bx = wave_max_f64(bx);
                        double relation = (mx - mn) / (bx - ax);
                        if (!(relation >= c.zxMinRel)) {
                            type = MLD_InsufficientRoadPoints;
                        } else {
                            {
                                SREC(0, L.x[ci]);
                                SREC(1, L.y[ci]);
                                SREC(2, L.z[ci]);
                                SREC(3, L.x[cj]);
                                SREC(4, L.y[cj]);
                                SREC(5, L.z[cj]);
                                SREC(6, L.x[ck]);
                                SREC(7, L.y[ck]);
                                SREC(8, L.z[ck]);
                            }
                            state = ST_ROAD_TRI;
                        }
                    }
                }
                if (state != ST_FINAL) {
                    SREC(9, mn);
                    SREC(10, mx);
                }
            }
        }
        if (lane == fi) {
            mystate = state;
            mytype = type;
            mydepth = -1.0;
        }
    }

    ST_MARK(8);
    // ---------------- phase 4 (lane = feature) ----------------
    if (mystate == ST_ROAD_MEST) {
        finish_road(c, false, myu, myv, myr, mytype, mydepth);  // (normal from the R of road_qr)
    } else if (mystate == ST_ROAD_TRI) {
        finish_road(c, true, myu, myv, myr, mytype, mydepth);
    }
    ST_USE_F64(mydepth);
    ST_MARK(9);
}

// ------------------------------------------------------------------------------------------------
// Lane-per-feature building blocks (k_feature_fused).  With the ~2-10 neighbours a 64-beam cloud gives a 7x10 window,
// a wave-per-feature design leaves most lanes idle; here every lane walks its own short list.  The only per-lane
// storage is the neighbour INDEX list in LDS (transposed, bank-conflict free); points are re-read from the cloud
// (L1/L2 hits) and re-transformed where needed.  All loops over list entries run in the reference's serial order.
// Features whose lists exceed the capacities are queued for k_feature_wave.
// ------------------------------------------------------------------------------------------------
constexpr int kK1MaxLimit = 64;  // upper bound of Calib::k1max, the per-feature neighbour list capacity
                                 // (LDS: k1max entries x 64 lanes x 4 B per wave; relative bins need k1max < 254)
constexpr int kK2Max = 16;  // longest list of the TRIANGLE ROAD ESTIMATOR's in-lane corner search (longer ones: wave kernel)
#ifndef MLD_TRI_SMALL
#define MLD_TRI_SMALL 8
#endif
constexpr int kTriSmall = MLD_TRI_SMALL;
constexpr int kTriTiny = 4;  // short tier of the default instantiation (segmented lists of at most 4 points in the wavefront)
#ifndef MLD_TRI_MID
#define MLD_TRI_MID 12
#endif
#ifndef MLD_TRI_LARGE
#define MLD_TRI_LARGE 16
#endif
#ifndef MLD_TRI_HUGE
#define MLD_TRI_HUGE 24
#endif
constexpr int kTriMid = MLD_TRI_MID, kTriLarge = MLD_TRI_LARGE, kTriHuge = MLD_TRI_HUGE;  // tiers of the in-register
                                                                                           // triangle search (DENSE kernel)
#ifndef MLD_KZC
#define MLD_KZC 12
#endif
constexpr int kZcDefault = MLD_KZC;  // list entries whose depth stays in registers over the histogram passes ...
#ifndef MLD_KZC_MID
#define MLD_KZC_MID 16
#endif
#ifndef MLD_KZC_DENSE
#define MLD_KZC_DENSE 24
#endif
constexpr int kZcMid = MLD_KZC_MID, kZcDense = MLD_KZC_DENSE;  // ... and in the further tiers of the DENSE instantiation
constexpr int kZcTiny = 4, kZcShort = 8;  // ... and in the short tiers of the default one
#ifndef MLD_ROAD_BATCH
#define MLD_ROAD_BATCH 4
#endif
constexpr int kRoadBatch = MLD_ROAD_BATCH;  // wide-window neighbours fetched per round trip by the road fallback
#ifndef MLD_LIST_BATCH
#define MLD_LIST_BATCH 4
#endif
constexpr int kBatch = MLD_LIST_BATCH;    // list entries fetched ahead of use in the per-lane list loops  // lists up to this length use the fully unrolled in-register triangle search

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

#define LST(e) lst[(e) * kWave + lane]

// The lane-per-feature kernel's view of a frame slot: the descriptor stays in device memory and a field is fetched - a
// scalar load, the address is wave-uniform - where it is used, instead of all 50 scalar registers' worth of it being
// loaded up front and carried (or spilled to vector lanes) across the whole kernel.
struct SlotRef {
    const SlotDesc& g;
    const double* plane;  // prior_n[3], prior_off, coeffs[4] (floats): the descriptor's, or the device-estimated plane's
                          // (PlaneDev starts with the same layout)
    int has_plane;
};
static_assert(offsetof(SlotDesc, prior_off) == offsetof(SlotDesc, prior_n) + 24 && offsetof(SlotDesc, coeffs) == offsetof(SlotDesc, prior_n) + 32 &&
                  offsetof(PlaneDev, prior_n) == 0 && offsetof(PlaneDev, prior_off) == 24 && offsetof(PlaneDev, coeffs) == 32,
              "SlotRef::plane reads the prior and the coefficients through one pointer");

__device__ __forceinline__ V3 cam_point(const Calib& c, const SlotRef& s, uint32_t orig) {
    double x, y, z;
    load_point(s.g, (long long)orig, x, y, z);
    return lidar_to_cam(c, x, y, z);
}

// Raw float point of the cloud and its camera-frame image (same arithmetic as load_point + lidar_to_cam).  The list
// loops below fetch several raw points before consuming any of them, so that the L2 latency is paid once per
// group of entries instead of once per entry.
struct RawP {
    float x, y, z;
};
__device__ __forceinline__ RawP load_raw(const SlotRef& s, uint32_t i) {
    MLD_DIAG_FAKE_POINT(i);
    const unsigned char* cl = s.g.cloud;
    const unsigned char* p = cl + (size_t)i * (size_t)s.g.stride;
    // one 12-byte load whatever the cloud's alignment (a 16-byte load for aligned clouds was a uniform branch per point:
    // control flow between the loads of a batch, and with it the compiler's `s_waitcnt vmcnt(0)` at every join)
    RawP r;
    r.x = GPTR(float, p)[0];
    r.y = GPTR(float, p)[1];
    r.z = GPTR(float, p)[2];
    return r;
}
__device__ __forceinline__ double raw_z(const Calib& c, RawP r) {
    return c.T[11] + ((c.T[8] * (double)r.x + c.T[9] * (double)r.y) + c.T[10] * (double)r.z);
}
__device__ __forceinline__ V3 raw_point(const Calib& c, RawP r) { return lidar_to_cam(c, (double)r.x, (double)r.y, (double)r.z); }
// list entry e of this lane, or 0 (a valid point index whenever any list is non-empty) beyond the list end
// (lcap: capacity of the list `lst` addresses - the wide list's c.k1max or the narrow list's c.kMain)
// (The LDS read is unconditional and the result masked: written as a select, the compiler predicates the read - a branch
//  and a full `s_waitcnt vmcnt(0)` per entry, which serialises the batch of point loads the entries are fetched for.)
#define LST_ID(e, n) (LST(min((e), lcap - 1)) & kIdxMask & (0u - (uint32_t)((e) < (n))))
#define LST_IF(cond, e, cap) (LST(min((e), (cap) - 1)) & (0u - (uint32_t)(cond)))

// Max-spanning triangle for lists of at most M entries, fully unrolled: all M points are fetched in one batch and
// kept in registers, pairs are visited in the reference's (i,j) order with strict '>' (first maximal pair wins),
// then the third corner over k < n-1.  One memory round trip instead of one per pair.
template <int M>
__device__ __forceinline__ bool triangle_small(const Calib& c, const SlotRef& s, int n, bool want, const uint32_t* lst,
                                               int lane, const int lcap, V3& c1, V3& c2, V3& c3) {
    const int nn = want ? n : 0;
    RawP rp[M];
#pragma unroll
    for (int q = 0; q < M; q++) rp[q] = load_raw(s, LST_ID(q, nn));
    V3 P[M];
#pragma unroll
    for (int q = 0; q < M; q++) P[q] = raw_point(c, rp[q]);
    double best = -1.0;
    int bi = -1, bj = -1;
#pragma unroll
    for (int i = 0; i < M - 1; i++) {
#pragma unroll
        for (int j = i + 1; j < M; j++) {
            const double d = vsqnorm(vsub(P[i], P[j]));
            const bool upd = (j < nn) && (d > best);
            best = upd ? d : best;
            bi = upd ? i : bi;
            bj = upd ? j : bj;
        }
    }
    bool ok = (nn >= 3) && !(best <= 0.0) && (bi >= 0);
    V3 pa = P[0], pb = P[0];
#pragma unroll
    for (int q = 1; q < M; q++) {
        pa.x = (bi == q) ? P[q].x : pa.x;
        pa.y = (bi == q) ? P[q].y : pa.y;
        pa.z = (bi == q) ? P[q].z : pa.z;
        pb.x = (bj == q) ? P[q].x : pb.x;
        pb.y = (bj == q) ? P[q].y : pb.y;
        pb.z = (bj == q) ? P[q].z : pb.z;
    }
    double best2 = -1.0;
    int bk = -1;
    V3 pk = P[0];
#pragma unroll
    for (int kx = 0; kx < M - 1; kx++) {  // last point never considered (:71)
        const double d1 = vsqnorm(vsub(P[kx], pa));
        const double d2 = vsqnorm(vsub(P[kx], pb));
        const double d = d1 + d2;
        const bool upd = ok && (kx < nn - 1) && (kx != bi) && (kx != bj) && !(d1 <= 0.0) && !(d2 <= 0.0) && (d > best2);
        best2 = upd ? d : best2;
        bk = upd ? kx : bk;
        pk.x = upd ? P[kx].x : pk.x;
        pk.y = upd ? P[kx].y : pk.y;
        pk.z = upd ? P[kx].z : pk.z;
    }
    ok = ok && (bk >= 0);
    c1 = pa;
    c2 = pb;
    c3 = pk;
    return ok;
}

// Max-spanning triangle over the thread's list entries [0, n) (PlaneEstimationCalcMaxSpanningTriangle.cpp:37-100),
// serial O(n^2) loops over memory (no register tier).  Callers: the main path for any segmented list up to the capacity
// kMain (the in-register tiers take the short ones first), the triangle road estimator for n <= kK2Max.  The wave iterates
// to the longest list; shorter lanes idle.
__device__ __forceinline__ bool triangle_thread(const Calib& c, const SlotRef& s, int n, bool want, const uint32_t* lst, int lane, V3& c1,
                                V3& c2, V3& c3) {
    bool act = want && n >= 3;
    int i = 0, j = 1, bi = -1, bj = -1;
    double best = -1.0;
    V3 pi = {0, 0, 0};
    bool fresh = true;
    while (__any(act)) {
        if (act) {
            if (fresh) pi = cam_point(c, s, LST(i));
            V3 pj = cam_point(c, s, LST(j));
            double d = vsqnorm(vsub(pi, pj));
            if (d > best) {
                best = d;
                bi = i;
                bj = j;
            }
            j++;
            fresh = false;
            if (j >= n) {
                i++;
                j = i + 1;
                fresh = true;
                if (i >= n - 1) act = false;
            }
        }
    }
    bool ok = want && n >= 3 && !(best <= 0.0) && bi >= 0;
    V3 pa = {0, 0, 0}, pb = {0, 0, 0};
    if (ok) {
        pa = cam_point(c, s, LST(bi));
        pb = cam_point(c, s, LST(bj));
    }
    int bk = -1;
    double best2 = -1.0;
    V3 pkbest = {0, 0, 0};
    const int lim = ok ? n - 1 : 0;  // last point never considered (:71)
    const int limmax = uniform(wave_max_i32(lim));
    for (int kx = 0; kx < limmax; kx++) {
        if (kx < lim && kx != bi && kx != bj) {
            V3 pk = cam_point(c, s, LST(kx));
            double d1 = vsqnorm(vsub(pk, pa));
            double d2 = vsqnorm(vsub(pk, pb));
            if (!(d1 <= 0.0) && !(d2 <= 0.0)) {
                double d = d1 + d2;
                if (d > best2) {
                    best2 = d;
                    bk = kx;
                    pkbest = pk;
                }
            }
        }
    }
    ok = ok && bk >= 0;
    c1 = pa;
    c2 = pb;
    c3 = pkbest;
    return ok;
}

// Appends (feature index, code) for the lanes with `want` set to a per-slot queue; one atomic per wavefront.
__device__ __forceinline__ void enqueue_features(int32_t* queue, int32_t* count, bool want, int lane, long long feature,
                                                 int code) {
    const unsigned long long m = __ballot(want);
    if (!m) return;
    int qbase = 0;
    if (lane == 0)
        qbase = __hip_atomic_fetch_add(GPTRW(int32_t, count), (int)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    qbase = uniform(qbase);
    if (want) {
        const int pos = qbase + prefix_count(m);
        GPTRW(int32_t, queue)[2 * (size_t)pos] = (int32_t)feature;
        GPTRW(int32_t, queue)[2 * (size_t)pos + 1] = code;
    }
}

// Road fallback of the thread path (DepthEstimator.cpp:578-597) for the lanes with `cand` set; mytype holds the main
// path's result (resultOld) on entry.  Lanes whose wide-window list exceeds the capacities set `overflow`.
// ROAD_MODE: 0 = M-estimator, 1 = max-spanning triangle, -1 = decided at run time (c.roadMode).
// The road fallback once the wide-window list (k2 entries, original point indices in the low 24 bits) is in `lst`.
// RB: wide-window neighbours fetched per round trip (kRoadBatch; twice that in the DENSE 2 instantiation, whose lists are
// long and whose two wavefronts per SIMD leave the registers: config 5 at 256 sequences -1.3 %, LAB.md 6.7)
template <int ROAD_MODE, int RB>
__device__ __forceinline__ void road_after_scan(const Calib& c, const SlotRef& s, uint32_t* lst, const int lane, bool cand,
                                                const int k2, const uint32_t list_states, const double myu,
                                                const double myv, int& mytype,
                                                double& mydepth, bool& overflow ST_ARG) {
    const int roadMode = ROAD_MODE >= 0 ? ROAD_MODE : c.roadMode;
    const int resultOld = mytype;
    const int lcap = c.k1max;  // `lst` is the wide list
    if (cand && k2 > c.k1max) {
        overflow = true;
        cand = false;
    }
    if (cand && (unsigned)k2 < c.countMin) {
        mytype = MLD_RadiusSearchInsufficientPoints;
        mydepth = -1.0;
        cand = false;
    }
    // CalculateDepthSegmentationPlane (DepthEstimator.cpp:782-900) + first M-estimator pass, serial order
    const int n2 = cand ? k2 : 0;
    bool far = false;
    int kk = 0;
    double zmn = 1.7976931348623157e308, zmx = -1.7976931348623157e308;
    double xmn = zmn, xmx = zmx;
    // weighted mean / scatter of the inliers, updated point by point in list order (West's one-pass update: no second
    // sweep over the list, no cancellation, and every increment of the scatter is a positive semi-definite term, so
    // degenerate (collinear) inlier sets keep their exactly-zero eigenvalues); the two divisions per point are
    // reciprocal estimates + Newton steps (road tolerance)
    double sw = 0, mx = 0, my = 0, mz = 0;
    double q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
    // origin of the sums: the first point of the list (for neighbouring returns the differences are exact, and the running
    // mean - whose rounding the scatter inherits - is then as small as the set is wide, not as far as it is away)
    V3 org = {0.0, 0.0, 0.0};
    const auto* pl = GPTR(double, s.plane);
    const V3 pn = {pl[0], pl[1], pl[2]};
    const double prior_off = pl[3];
    // one inlier of the plane: depth / lateral extent of the set and West's update of the weighted mean and scatter
    auto add_inlier = [&](const V3 p) {
        zmn = fmin(zmn, p.z);  // (a NaN coordinate leaves the extremes alone, as `if (p.z < zmn)` does)
        zmx = fmax(zmx, p.z);
        if (roadMode != 0) {   // the lateral extent is the triangle estimator's (LinePlaneIntersectionCeckXZTreshold)
            xmn = fmin(xmn, p.x);
            xmx = fmax(xmx, p.x);
        }
        if (roadMode == 0) {
            // PlaneEstimationMEstimator.cpp:32.  The distance is a cancellation (a road point lies on the prior plane to
            // within centimetres or less) and its reciprocal is the weight: the point and the dot product are therefore
            // evaluated in the reference's own operation order, rounding for rounding - a fused chain changes a weight by
            // 1e-11 relative where the distance is 1e-4 m, and an estimate kilometres away (thresholds off) by 2e-4 m
            // (profiles/tools/random_sweep.py, seed 1990; LAB.md 5.32)
            const double w = fast_rcp(fabs(vdot(pn, p) + prior_off));
            const double swn = sw + w;
            const double r = w * fast_rcp(swn);
            const double sx = p.x - org.x, sy = p.y - org.y, sz = p.z - org.z;
            const double dx = sx - mx, dy = sy - my, dz = sz - mz;
            mx = fma(dx, r, mx);
            my = fma(dy, r, my);
            mz = fma(dz, r, mz);
            const double ex = sx - mx, ey = sy - my, ez = sz - mz;
            const double wdx = w * dx, wdy = w * dy, wdz = w * dz;
            q0 = fma(wdx, ex, q0);
            q1 = fma(wdx, ey, q1);
            q2 = fma(wdx, ez, q2);
            q3 = fma(wdy, ey, q3);
            q4 = fma(wdy, ez, q4);
            q5 = fma(wdz, ez, q5);
            sw = swn;
        }
    };
    // The plane was known when the cloud was projected: every entry carries its point's state (k_project_scatter).
    // Whether the estimator runs at all (:591: no far point, at least three inliers) is then decided from the list
    // alone, and only the inliers' points are fetched - in list order, as the serial loop meets them.  A wavefront
    // that holds an "unsure" entry (|distance - threshold| within the projection's single-precision margin) takes the
    // general loop below.
    bool by_state = false;
    if (s.g.mask_in_key) {
        // (what the scan saw in the lane's keys: scan_window_flagged)
        far = n2 > 0 && (list_states & 1u);
        by_state = !wave_any(n2 > 0 && (list_states & 2u));
        if (!by_state) far = false;
    }
    if (by_state) {
        const int n2f = far ? 0 : n2;  // :591 a far point in the window: the feature keeps the main path's result
        const int n2fmax = uniform(wave_max_i32(n2f));
        // four independent LDS reads per step; kk <= e: entries not yet read are never overwritten
        for (int e = 0; e < n2fmax; e += 4) {
            uint32_t ent[4];
#pragma unroll
            for (int q = 0; q < 4; q++) ent[q] = LST_IF(e + q < n2f, e + q, c.k1max);
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (((ent[q] >> kEntStateShift) & 3u) == kPtInlier) {
                    LST(kk) = ent[q] & kIdxMask;
                    kk++;
                }
        }
        const int ni = (cand && !far && kk >= 3) ? kk : 0;
        const int nimax = uniform(wave_max_i32(ni));
        for (int e0 = 0; e0 < nimax; e0 += RB) {
            RawP rp[RB];
#pragma unroll
            for (int q = 0; q < RB; q++) rp[q] = load_raw(s, LST_ID(e0 + q, ni));
            if (e0 == 0 && roadMode == 0) org = raw_point(c, rp[0]);  // (wave-uniform branch; a lane without inliers adds nothing)
#pragma unroll
            for (int q = 0; q < RB; q++)
                if (e0 + q < ni) add_inlier(raw_point(c, rp[q]));
        }
    } else {
    const int n2max = uniform(wave_max_i32(n2));
    for (int e0 = 0; e0 < n2max; e0 += RB) {
        RawP rp[RB];
        uint32_t ids[RB], mw[RB];
#pragma unroll
        for (int q = 0; q < RB; q++) {
            const uint32_t ent = LST_IF(e0 + q < n2, e0 + q, c.k1max);
            ids[q] = ent & kIdxMask;
            rp[q] = load_raw(s, ids[q]);
            mw[q] = GPTR(uint32_t, s.g.inlier_mask)[ids[q] >> 5];
        }
        if (e0 == 0 && roadMode == 0) org = raw_point(c, rp[0]);  // (the window's first return: near the inliers or not, see finish_road_fast)
#pragma unroll
        for (int q = 0; q < RB; q++) {
          if (e0 + q < n2) {
            const uint32_t id = ids[q];
            V3 p = raw_point(c, rp[q]);
            double xl = c.Tinv[3] + (c.Tinv[0] * p.x + (c.Tinv[1] * p.y + c.Tinv[2] * p.z));
            double yl = c.Tinv[7] + (c.Tinv[4] * p.x + (c.Tinv[5] * p.y + c.Tinv[6] * p.z));
            double zl = c.Tinv[11] + (c.Tinv[8] * p.x + (c.Tinv[9] * p.y + c.Tinv[10] * p.z));
            float xf = (float)xl, yf = (float)yl, zf = (float)zl;
            const auto* co = GPTR(float, s.plane + 4);
            float d = fabsf(co[0] * xf + co[1] * yf + co[2] * zf + co[3]);
            far = far || ((double)d > c.roadDistThr);
            bool inl = (mw[q] >> (id & 31)) & 1u;
            if (inl) {
                LST(kk) = id;
                kk++;
                add_inlier(p);
            }
          }
        }
    }
    }
    ST_USE_F64(sw);
    ST_MARK(13);
    if (cand && (far || kk < 3)) {
        mytype = resultOld;  // :591
        mydepth = -1.0;
        cand = false;
    }
    double rr[kRecFields];
#pragma unroll
    for (int t = 0; t < kRecFields; t++) rr[t] = 0.0;
    rr[9] = zmn;
    rr[10] = zmx;
    if (roadMode == 0) {
        // weighted centre (PlaneEstimationMEstimator.cpp:31-37) and scatter about it (:39-46)
        rr[0] = org.x + mx; rr[1] = org.y + my; rr[2] = org.z + mz;
        rr[3] = q0; rr[4] = q1; rr[5] = q2; rr[6] = q3; rr[7] = q4; rr[8] = q5;
        const double cdev = fmax(fabs(mx), fmax(fabs(my), fabs(mz)));
        if (cand && !finish_road_fast(c, myu, myv, rr, sw, cdev, mytype, mydepth)) overflow = true;  // (redone by the wave kernel)
        ST_USE_F64(mydepth);
        ST_MARK(14);
    } else {
        // RoadDepthEstimatorMaxSpanningTriangle::CalculateDepth (:24-75)
        if (cand && kk > kK2Max) {
            overflow = true;
            cand = false;
        }
        V3 c1, c2, c3;
        bool ok = triangle_small<kTriSmall>(c, s, kk, cand && kk <= kTriSmall, lst, lane, lcap, c1, c2, c3);
        if (__any(cand && kk > kTriSmall)) {
            V3 d1, d2, d3;
            bool ok2 = triangle_thread(c, s, kk, cand && kk > kTriSmall, lst, lane, d1, d2, d3);
            if (kk > kTriSmall) {
                ok = ok2;
                c1 = d1;
                c2 = d2;
                c3 = d3;
            }
        }
        if (cand && !ok) {
            mytype = MLD_RadiusSearchInsufficientPoints;
            mydepth = -1.0;
            cand = false;
        }
        if (cand) {
            double relation = (zmx - zmn) / (xmx - xmn);
            if (!(relation >= c.zxMinRel)) {
                mytype = MLD_InsufficientRoadPoints;
                mydepth = -1.0;
                cand = false;
            }
        }
        rr[0] = c1.x; rr[1] = c1.y; rr[2] = c1.z;
        rr[3] = c2.x; rr[4] = c2.y; rr[5] = c2.z;
        rr[6] = c3.x; rr[7] = c3.y; rr[8] = c3.z;
        if (cand) finish_road(c, true, myu, myv, rr, mytype, mydepth);
    }
}

// Everything between the window scan and the road fallback for one feature per lane (DepthEstimator.cpp:564-576):
// histogram segmentation, corner selection, tail.  `lst` / `lane` address the lane's index list (they need not be
// the executing wave's own region: k_feature_main re-deals the live features of a block to dense wavefronts).
// First half: depth segmentation of the lane's neighbour list (DepthEstimator.cpp:726-780).  On return the list holds
// the segmented points (ks entries, reference order) and minZ / maxZ their depth range; a failed histogram sets
// mytype = HistogramNoLocalMax and clears `live`.
template <int kZc>  // list entries whose depth stays in registers over the histogram passes
__device__ __forceinline__ void main_hist(const Calib& c, const SlotRef& s, uint32_t* lst, const int lane, const int lcap,
                                          const int k, bool& live, int& mytype, int& ks, double& minZ, double& maxZ ST_ARG) {
    ks = live ? k : 0;
    minZ = 1.7976931348623157e308;
    maxZ = -1.7976931348623157e308;
    if (c.useHist) {
        // PointHistogram::FilterPointsMinDistBlob (HistogramPointDepth.cpp:15-123), per thread
        const int kmax = uniform(wave_max_i32(ks));
        int md = 0;
        double dmin = 1.7976931348623157e308;
        // z of the first kZc entries stays in registers over the three passes; longer lists reload the rest
        double zc[kZc];
        {
            RawP rp[kZc];
#pragma unroll
            for (int q = 0; q < kZc; q++) rp[q] = load_raw(s, LST_ID(q, ks));
#pragma unroll
            for (int q = 0; q < kZc; q++) {
                zc[q] = raw_z(c, rp[q]);
                double d = (999. < zc[q]) ? 999. : zc[q];
                int ce = (int)ceil(d);
                const bool ok = q < ks;
                md = ok ? max(md, ce) : md;
                dmin = (ok && d < dmin) ? d : dmin;
            }
        }
        for (int e0 = kZc; e0 < kmax; e0 += kBatch) {
            RawP rp[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; q++) rp[q] = load_raw(s, LST_ID(e0 + q, ks));
#pragma unroll
            for (int q = 0; q < kBatch; q++) {
                double d = raw_z(c, rp[q]);
                d = (999. < d) ? 999. : d;
                int ce = (int)ceil(d);
                const bool ok = e0 + q < ks;
                md = ok ? max(md, ce) : md;
                dmin = (ok && d < dmin) ? d : dmin;
            }
        }
        ST_USE_U32(md);
        ST_MARK(8);
        const int binCount = (int)((double)md / c.binW + 1.0);
        bool hfail = binCount <= 1;
        const double lim = (double)binCount - 1.;
        int bmin = 0;
        {
            double value = (1e10 < dmin) ? 1e10 : dmin;
            double q = fabs(value / c.binW);
            bmin = (int)((lim < q) ? lim : q);  // bin index is monotone in d: the smallest d gives the first bin
        }
        // bins (relative to the first one, one byte each) of the first kZc entries: packed in registers, 0xFF beyond
        // the list's end (never a bin the scan below asks for); the entries behind them carry theirs in the list
        constexpr int kRelWords = (kZc + 3) / 4;
        uint32_t relw[kRelWords];
#pragma unroll
        for (int t = 0; t < kRelWords; t++) relw[t] = 0xFFFFFFFFu;
#pragma unroll
        for (int q = 0; q < kZc; q++) {
            double d = (999. < zc[q]) ? 999. : zc[q];
            double value = (1e10 < d) ? 1e10 : d;
            double qq = fabs(value / c.binW);
            int bi = (int)((lim < qq) ? lim : qq);
            int rel = bi - bmin;
            rel = rel > 255 ? 255 : rel;  // (the scan below never asks for bin 255: 0xFF doubles as "no entry")
            const uint32_t byte = (q < ks) ? (uint32_t)rel : 0xFFu;
            relw[q >> 2] = (relw[q >> 2] & ~(0xFFu << (8 * (q & 3)))) | (byte << (8 * (q & 3)));
        }
        for (int e0 = kZc; e0 < kmax; e0 += kBatch) {
            RawP rp[kBatch];
            uint32_t ids[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; q++) {
                ids[q] = LST_ID(e0 + q, ks);
                rp[q] = load_raw(s, ids[q]);
            }
#pragma unroll
            for (int q = 0; q < kBatch; q++) {
                double d = raw_z(c, rp[q]);
                d = (999. < d) ? 999. : d;
                double value = (1e10 < d) ? 1e10 : d;
                double qq = fabs(value / c.binW);
                int bi = (int)((lim < qq) ? lim : qq);
                int rel = bi - bmin;
                rel = rel > 255 ? 255 : rel;
                if (e0 + q < ks) LST(e0 + q) = ids[q] | ((uint32_t)rel << kIdxBits);
            }
        }
        // scan of the bins (HistogramPointDepth.cpp:70-85) from the first non-empty one
        int binMaxRel = -1, binMaxVal = -1, binValue = 0, irel = 0;
        bool run = live && !hfail && ks > 0;
        while (__any(run)) {
            if (run) {
                if (bmin + irel >= binCount) {
                    run = false;
                } else {
                    int last = binValue;
                    int cnt = 0;
                    // the first kZc entries: bytes equal to irel, counted in the packed registers (a zero byte of
                    // x ^ irel-in-every-byte; exact zero-byte mask, one population count per word)
                    {
                        const uint32_t rep = (uint32_t)irel * 0x01010101u;
#pragma unroll
                        for (int t = 0; t < kRelWords; t++) {
                            const uint32_t y = relw[t] ^ rep;
                            const uint32_t nz = ((y & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | y | 0x7F7F7F7Fu;
                            cnt += __popc(~nz);
                        }
                    }
                    // the rest, if any lane's list is longer: four independent LDS reads per step (wave-uniform
                    // bound) instead of one dependent read per entry
                    for (int e = kZc; e < kmax; e += 4) {
                        uint32_t vv[4];
#pragma unroll
                        for (int q = 0; q < 4; q++) vv[q] = LST(min(e + q, lcap - 1));
#pragma unroll
                        for (int q = 0; q < 4; q++)
                            cnt += ((e + q < ks) && ((vv[q] >> kIdxBits) == (uint32_t)irel)) ? 1 : 0;
                    }
                    binValue = cnt;
                    if ((binValue > binMaxVal) && (binValue >= c.minCount)) {
                        binMaxVal = binValue;
                        binMaxRel = irel;
                    } else if (binValue < binMaxVal) {
                        run = false;
                    }
                    if (run && (last > 0) && (binValue == 0)) {
                        hfail = true;
                        run = false;
                    }
                    irel++;
                    if (irel >= 255) run = false;  // cannot happen: at most ks+1 <= 65 bins are visited
                }
            }
        }
        if (binMaxRel < 0) hfail = true;
        const double lower = (double)(bmin + binMaxRel) * c.binW - 0.0 * c.binW;
        const double higher = (double)(bmin + binMaxRel) * c.binW + 1.0 * c.binW;
        int kk = 0;
#pragma unroll
        for (int q = 0; q < kZc; q++) {
            const uint32_t packed = LST_IF(q < ks, q, lcap);
            const int rel = (int)((relw[q >> 2] >> (8 * (q & 3))) & 0xFFu);
            const double z = zc[q];
            const double d = (999. < z) ? 999. : z;
            const bool keep = (q < ks) && !hfail && (rel >= binMaxRel - 1) && (rel <= binMaxRel + 1) &&
                              (d >= lower) && (d < higher);
            if (keep) {
                LST(kk) = packed & kIdxMask;
                kk++;
                if (z < minZ) minZ = z;
                if (z > maxZ) maxZ = z;
            }
        }
        for (int e0 = kZc; e0 < kmax; e0 += kBatch) {
            RawP rp[kBatch];
            uint32_t packed[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; q++) {
                packed[q] = LST_IF(e0 + q < ks, e0 + q, lcap);
                rp[q] = load_raw(s, packed[q] & kIdxMask);
            }
#pragma unroll
            for (int q = 0; q < kBatch; q++) {
                const int rel = (int)(packed[q] >> kIdxBits);
                const uint32_t id = packed[q] & kIdxMask;
                const double z = raw_z(c, rp[q]);
                const double d = (999. < z) ? 999. : z;
                // membership is by [lower, higher), not by bin; only the neighbouring bins can qualify
                const bool keep = (e0 + q < ks) && !hfail && (rel >= binMaxRel - 1) && (rel <= binMaxRel + 1) &&
                                  (d >= lower) && (d < higher);
                if (keep) {
                    LST(kk) = id;  // kk <= e0 + q: entries not yet read are never overwritten
                    kk++;
                    if (z < minZ) minZ = z;
                    if (z > maxZ) maxZ = z;
                }
            }
        }
        if (live && hfail) {
            mytype = MLD_HistogramNoLocalMax;
            live = false;
        }
        ks = live ? kk : 0;
    } else {
        const int kmax = uniform(wave_max_i32(ks));
        for (int e0 = 0; e0 < kmax; e0 += kBatch) {
            RawP rp[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; q++) rp[q] = load_raw(s, LST_ID(e0 + q, ks));
#pragma unroll
            for (int q = 0; q < kBatch; q++) {
                const double z = raw_z(c, rp[q]);
                if (e0 + q < ks) {
                    if (z < minZ) minZ = z;
                    if (z > maxZ) maxZ = z;
                }
            }
        }
    }

    ST_USE_U32(ks);
    ST_MARK(9);
}

// Second half: CalculateDepthSegmented (DepthEstimator.cpp:903-1037) on the segmented list: corner selection (max
// spanning triangle / first three points / PCA moments), planarity, ray-plane intersection, thresholds.
template <int DENSE>
__device__ __forceinline__ void main_tail(const Calib& c, const SlotRef& s, uint32_t* lst, const int lane, const int lcap,
                                          const int ks_in,
                                          bool live, const double minZ, const double maxZ, const double myu,
                                          const double myv, int& mytype, double& mydepth, bool& overflow ST_ARG) {
    MLD_DIAG_SKIP_TAIL();
    int ks = live ? ks_in : 0;
    double r[kRecFields];
#pragma unroll
    for (int t = 0; t < kRecFields; t++) r[t] = 0.0;
    bool pca = false;
    if (!c.usePCA && c.useTriMax) {
        if (live && ks < 3) {
            mytype = MLD_TriangleNotPlanarInsufficientPoints;
            live = false;
        }
        // Corner search with the segmented points in REGISTERS, fully unrolled, sized by the longest list of the
        // wavefront (k_classify orders the queue by neighbour count, so the lists of a wavefront are alike): one memory
        // round trip for all points and then no load inside the O(n^2) pair loop.  Beyond kTriSmall entries that takes
        // more registers than three wavefronts per SIMD leave (168): the DENSE instantiation of the kernel (two
        // wavefronts per SIMD, which is all its LDS allows with long lists anyway) has tiers up to kTriHuge entries; the
        // other one falls back to the serial loops (triangle_thread: a dependent L1 / L2 round trip per pair, ~500 cycles
        // each - a dense 128-beam cloud spent 39 % of the kernel there).
        V3 c1, c2, c3;
        bool ok;
        const int ksmax = uniform(wave_max_i32(live ? ks : 0));
        // (DENSE 2: tiers up to 24 points; DENSE 1 - the dense kernel beside another context's projection, 168 registers -
        // up to 16; the default instantiation 8)
        constexpr int kTop = DENSE == 2 ? kTriHuge : (DENSE == 1 ? kTriLarge : kTriSmall);
        if (DENSE == 0 && ksmax <= kTriTiny) {
            ok = triangle_small<DENSE == 0 ? kTriTiny : 1>(c, s, ks, live, lst, lane, lcap, c1, c2, c3);  // 6 pairs instead of 28
        } else if (ksmax <= kTriSmall || ksmax > kTop) {
            ok = triangle_small<kTriSmall>(c, s, ks, live && ks <= kTriSmall, lst, lane, lcap, c1, c2, c3);
            if (ksmax > kTriSmall) {  // longer segmented lists: the generic serial loops
                V3 d1, d2, d3;
                bool ok2 = triangle_thread(c, s, ks, live && ks > kTriSmall, lst, lane, d1, d2, d3);
                if (ks > kTriSmall) {
                    ok = ok2;
                    c1 = d1;
                    c2 = d2;
                    c3 = d3;
                }
            }
        } else if (ksmax <= kTriMid) {
            ok = triangle_small<DENSE ? kTriMid : 1>(c, s, ks, live, lst, lane, lcap, c1, c2, c3);
        } else if (ksmax <= kTriLarge) {
            ok = triangle_small<DENSE ? kTriLarge : 1>(c, s, ks, live, lst, lane, lcap, c1, c2, c3);
        } else {
            ok = triangle_small<DENSE == 2 ? kTriHuge : 1>(c, s, ks, live, lst, lane, lcap, c1, c2, c3);
        }
        if (live && !ok) {
            mytype = MLD_TriangleNotPlanarInsufficientPoints;
            live = false;
        }
        r[0] = c1.x; r[1] = c1.y; r[2] = c1.z;
        r[3] = c2.x; r[4] = c2.y; r[5] = c2.z;
        r[6] = c3.x; r[7] = c3.y; r[8] = c3.z;
    } else {
        if (live && ks < 3) {
            mytype = MLD_HistogramNoLocalMax;  // :920-921
            live = false;
        }
        if (c.usePCA) {
            // Mono_LidarPipeline::PCA::CalculatePCA (PCA.cpp:45-62): mean, scatter, in index order
            pca = true;
            const int n = live ? ks : 0;
            const int nmax = uniform(wave_max_i32(n));
            double sx = 0, sy = 0, sz = 0;
            for (int e = 0; e < nmax; e++)
                if (e < n) {
                    V3 p = cam_point(c, s, LST(e));
                    sx += p.x;
                    sy += p.y;
                    sz += p.z;
                }
            const double mx = sx / (double)n, my = sy / (double)n, mz = sz / (double)n;
            double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
            for (int e = 0; e < nmax; e++)
                if (e < n) {
                    V3 p = cam_point(c, s, LST(e));
                    double dx = p.x - mx, dy = p.y - my, dz = p.z - mz;
                    c0 += dx * dx;
                    c1 += dx * dy;
                    c2 += dx * dz;
                    c3 += dy * dy;
                    c4 += dy * dz;
                    c5 += dz * dz;
                }
            r[0] = mx; r[1] = my; r[2] = mz;
            r[3] = c0; r[4] = c1; r[5] = c2; r[6] = c3; r[7] = c4; r[8] = c5;
        } else if (live) {
            V3 p0 = cam_point(c, s, LST(0)), p1 = cam_point(c, s, LST(1)), p2 = cam_point(c, s, LST(2));
            r[0] = p0.x; r[1] = p0.y; r[2] = p0.z;
            r[3] = p1.x; r[4] = p1.y; r[5] = p1.z;
            r[6] = p2.x; r[7] = p2.y; r[8] = p2.z;
        }
    }
    r[9] = minZ;
    r[10] = maxZ;
    ST_USE_F64(r[0]);
    ST_USE_F64(r[8]);
    ST_MARK(10);
    if (live) finish_main(c, pca, myu, myv, r, mytype, mydepth);
    ST_USE_F64(mydepth);
    ST_MARK(11);
}

template <int DENSE>
__device__ __forceinline__ void main_after_scan(const Calib& c, const SlotRef& s, uint32_t* lst, const int lane,
                                                const int lcap, const int k, bool live, const double myu, const double myv,
                                                int& mytype, double& mydepth, bool& overflow ST_ARG) {
    int ks;
    double minZ, maxZ;
    // (DENSE: a wavefront whose longest list exceeds the first tier keeps 16 / 24 depths in registers - one memory round
    // trip for the depths instead of three passes of four-entry round trips over the tail)
    const int kmax0 = uniform(wave_max_i32(live ? k : 0));
    if (DENSE && kmax0 > kZcDefault && (kmax0 <= kZcMid || DENSE == 1))
        main_hist<DENSE ? kZcMid : kZcDefault>(c, s, lst, lane, lcap, k, live, mytype, ks, minZ, maxZ ST_PASS);
    else if (DENSE == 2 && kmax0 > kZcMid)
        main_hist<DENSE == 2 ? kZcDense : kZcDefault>(c, s, lst, lane, lcap, k, live, mytype, ks, minZ, maxZ ST_PASS);
    // (default instantiation: wavefronts whose longest narrow list has at most 4 / 8 entries - k_classify orders the queue
    //  by list length; random features on a 64-beam cloud: most of them - fetch and bin 4 / 8 depths instead of 12)
    // (the DENSE instantiations keep 12 as their shortest tier: with the short ones as well config 5 at S = 256 ran 2.5 %
    //  SLOWER - 52 instead of 46 spilled registers, 32 k instructions -, LAB.md 6.20)
    else if (DENSE == 0 && kmax0 <= kZcTiny)
        main_hist<DENSE == 0 ? kZcTiny : kZcDefault>(c, s, lst, lane, lcap, k, live, mytype, ks, minZ, maxZ ST_PASS);
    else if (DENSE == 0 && kmax0 <= kZcShort)
        main_hist<DENSE == 0 ? kZcShort : kZcDefault>(c, s, lst, lane, lcap, k, live, mytype, ks, minZ, maxZ ST_PASS);
    else
        main_hist<kZcDefault>(c, s, lst, lane, lcap, k, live, mytype, ks, minZ, maxZ ST_PASS);
    main_tail<DENSE>(c, s, lst, lane, lcap, ks, live, minZ, maxZ, myu, myv, mytype, mydepth, overflow ST_PASS);
}

// Wave-cooperative kernel for the features the thread kernels could not hold (lists longer than their capacities).
// Queue entries: (feature, -1) = main path + road fallback, (feature, t >= 0) = road fallback only, t being the
// main path's result.
// `chunk` (<= 64) queue entries per block and iteration: a wavefront works on ONE feature at a time, so small chunks
// spread a long queue (dense clouds: every feature overflows) over many more wavefronts.
// (171 VGPRs, two wavefronts per SIMD; 168 / 128 registers for three / four were measured: same time - the kernel is
// bound by instruction issue, ~5000 VALU instructions per feature, not by occupancy)
// direct != 0 (ONE frame per call): no queue - entry e IS feature e, and the block settles the features without
// enough neighbours itself (what k_classify does for batches: window count in the occupancy bitmap, which the projection
// has just left in L2), so a single-frame call is projection + this kernel.
// (DIRECT as a template parameter: the one-frame instantiation is straight-line code - one pass per block, nothing for
// the compiler to hoist in front of a loop and carry in spilled scalar registers)
template <bool DIRECT>
__global__ __launch_bounds__(kWave) void k_feature_wave(const SlotDesc* __restrict__ slots, SlotDesc single, int use_single,
                                                        const Calib* __restrict__ calib, int n_slots, int per_slot,
                                                        uint32_t tag_all, int chunk) {
    constexpr int direct = DIRECT ? 1 : 0;
    const Calib& c = *calib;  // (in device memory, as for k_feature_fused: fields are fetched where they are used)
    extern __shared__ __align__(16) unsigned char smem[];
    int slot, j;
    decode_block((int)blockIdx.x, c.xcdAware ? n_slots : -n_slots, per_slot, slot, j);
    SlotDesc s = use_single ? single : slots[slot];
    apply_plane_dev(s);
    if (tag_all) s.tag = tag_all;
    int count;
    if (direct) {
        long long Fn = s.F;
        if (s.F_dev) {
            long long fd = *GPTR(long long, s.F_dev) - s.F_dev_off;
            fd = fd < 0 ? 0 : fd;
            Fn = fd < Fn ? fd : Fn;
        }
        count = (int)Fn;
    } else {
        if (!s.ovf_count) return;
        count = *GPTR(int32_t, s.ovf_count);
    }
    const int lane = threadIdx.x;
    auto pass = [&](const int e0) {
        bool active = lane < chunk && e0 + lane < count;
        long long f = 0;
        int code = 0;
        double myu = 0, myv = 0;
        if (active && direct) {
            f = e0 + lane;
            code = -1;
            const auto* q = GPTR(double, s.uv) + 2 * f;
            myu = q[0];
            myv = q[1];
            // DepthEstimator.cpp:680: fewer than radiusSearch_count_min neighbours in the search window
            int x0, y0, nx, ny;
            if (window_bounds(c, myu, myv, c.halfX1, c.halfY1, x0, y0, nx, ny) && nx <= 32) {
                const auto* bm = GPTR(uint32_t, s.bitmap) + (size_t)(x0 >> 5) * (size_t)c.bmStride;
                const uint32_t sh = (uint32_t)(x0 & 31);
                const uint32_t colmask = (nx >= 32) ? ~0u : ((1u << nx) - 1u);
                int k1 = 0;
                for (int r = 0; r < ny; r++) {
                    const uint32_t lo = bm[y0 + r], hi = bm[c.bmStride + y0 + r];
                    k1 += __popc(__builtin_amdgcn_alignbit(hi, lo, sh) & colmask);
                }
                if ((unsigned)k1 < c.countMin) {
                    GPTRW(double, s.depth)[f] = -1.0;
                    if (s.type) GPTRW(int32_t, s.type)[f] = MLD_RadiusSearchInsufficientPoints;
                    active = false;
                }
            } else if (nx <= 32) {  // no window at all (non-finite feature): zero neighbours
                if (0u < c.countMin) {
                    GPTRW(double, s.depth)[f] = -1.0;
                    if (s.type) GPTRW(int32_t, s.type)[f] = MLD_RadiusSearchInsufficientPoints;
                    active = false;
                }
            }
        } else if (active) {
            f = (long long)GPTR(int32_t, s.ovf_queue)[2 * (size_t)(e0 + lane)];
            code = GPTR(int32_t, s.ovf_queue)[2 * (size_t)(e0 + lane) + 1];
            const auto* q = GPTR(double, s.uv) + 2 * f;
            myu = q[0];
            myv = q[1];
        }
        int mytype = code < 0 ? (int)MLD_Unspecified : code;
        double mydepth = -1.0;
        const unsigned long long all = __ballot(active);
        const unsigned long long full = __ballot(active && code < 0);
        double corners[9];
        bool has_corners = false;
        ST_BEGIN(1);
        wave_path(c, s, smem, lane, all, myu, myv, mytype, mydepth, full, corners, has_corners ST_PASS);
        if (active) {
            GPTRW(double, s.depth)[f] = mydepth;
            if (s.type) GPTRW(int32_t, s.type)[f] = mytype;
            if (s.corners && has_corners) {
#pragma unroll
                for (int t = 0; t < 9; t++) GPTRW(double, s.corners)[9 * f + t] = corners[t];
            }
        }
        ST_MARK(12);
        ST_END();
    };
    if (DIRECT) {  // (the host sizes the one-frame grid so that one pass per block covers every feature)
        if (j * chunk < count) pass(j * chunk);
    } else {
        for (int e0 = j * chunk; e0 < count; e0 += per_slot * chunk) pass(e0);
    }
}

// ------------------------------------------------------------------------------------------------
// Classification + dense fused feature kernel (the default path).
//
// What bounds the per-feature work on MI355X is the rate of DIVERGENT gathers (one lane = one cache line): about
// 0.45 lanes/clk/CU from L2 and 0.08 from HBM (profiles/tools/randgather.hip), whatever the occupancy.  So the path is
// organised to issue as few of them as possible:
//   k_classify      one block per frame slot.  The slot's occupancy bitmap (63 KB) is staged in LDS with coalesced
//                   loads and every feature's narrow window is counted there: features without enough neighbours
//                   (RadiusSearchInsufficientPoints, 43 % of a KITTI-like frame) get their result at once and never
//                   touch global memory again; the others are written, sorted by image row, to the slot's LIVE queue.
//   k_feature_fused one lane per live feature (dense wavefronts, no dealing, no barrier).  ONE scan of the wide
//                   (road) window serves both the main path and the road fallback: its entries carry a flag "inside
//                   the narrow window"; the flagged sub-list (same row-major order) feeds histogram / triangle / tail,
//                   the whole list the road estimator.  Keys and points of a window are fetched once.
// ------------------------------------------------------------------------------------------------

// Integer window bounds exactly as NeighborFinderPixel::getNeighbors (NeighborFinderPixel.cpp:67-76): clamped doubles,
// truncation.  false: no window (non-finite feature — undefined behaviour in the reference — or empty).
__device__ __forceinline__ bool window_bounds(const Calib& c, double u, double v, double halfX, double halfY, int& x0,
                                              int& y0, int& nx, int& ny) {
    x0 = 0;
    y0 = 0;
    nx = 0;
    ny = 0;
    if (!(isfinite(u) && isfinite(v))) return false;
    double a;
    a = u - halfX;
    const double left = (a < 0.) ? 0. : a;
    a = u + halfX;
    const double right = ((double)(c.W - 1) < a) ? (double)(c.W - 1) : a;
    a = v - halfY;
    const double top = (a < 0.) ? 0. : a;
    a = v + halfY;
    const double bottom = ((double)(c.H - 1) < a) ? (double)(c.H - 1) : a;
    const int xa = (int)left, ya = (int)top, xb = (int)right, yb = (int)bottom;
    const int wx = xb - xa + 1, wy = yb - ya + 1;
    if (wx <= 0 || wy <= 0 || xa < 0 || ya < 0 || xb >= c.W || yb >= c.H) return false;
    x0 = xa;
    y0 = ya;
    nx = wx;
    ny = wy;
    return true;
}

constexpr int kClsThreads = 1024;  // the LDS image of the bitmap allows two blocks per CU: 16 waves each fill it
constexpr int kClsBuckets = 1024;
constexpr int kClsKeep = 2;  // features per thread whose class stays in registers between the two passes
enum : int { CLS_DEAD = -1, CLS_OVF = -2, CLS_NONE = -3 };

// STAGED: the slot's bitmap is copied to LDS (row-major there: word [y * ncolp + column], ncolp odd) and read from
// it; otherwise (bitmap larger than the LDS budget: very large images) it is read in place.
template <bool STAGED, int kClsThreads, int kClsKeep>  // (block shape as template parameters: 256 / 512-thread variants
                                                       //  were measured for the shared-GPU schedule and lost)
__global__ __launch_bounds__(kClsThreads) void k_classify(const SlotDesc* __restrict__ slots, SlotDesc single,
                                                          int use_single, Calib c, int ncol, int ncolp,
                                                          uint32_t* __restrict__ done) {
    extern __shared__ __align__(16) unsigned char smem[];
    int* hist = reinterpret_cast<int*>(smem);           // [kClsBuckets]
    int* wsum = hist + kClsBuckets;                     // [kClsThreads / kWave]
    int* ctr = wsum + kClsThreads / kWave;              // [0] overflow-queue cursor
    uint32_t* lbm = reinterpret_cast<uint32_t*>(ctr + 4);  // staged bitmap
    SlotDesc s = use_single ? single : slots[blockIdx.x];
    apply_plane_dev(s);
    long long Fn = s.F;
    if (s.F_dev) {
        long long fd = *GPTR(long long, s.F_dev) - s.F_dev_off;
        fd = fd < 0 ? 0 : fd;
        Fn = fd < Fn ? fd : Fn;
    }
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    // (hand-over by gate, mld_order_after_classify: blocks that have been PLACED, counted for k_gate of the other context.
    // Once the last block runs, nothing of this kernel still waits for wave slots or LDS, which is all the other context's
    // projection blocks could take from it; counting at the end released them a block's lifetime - 12 us - later.)
    if (done && tid == 0) __hip_atomic_fetch_add(GPTRW(uint32_t, done), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int b = tid; b < kClsBuckets; b += kClsThreads) hist[b] = 0;
    if (tid == 0) ctr[0] = 0;
    const auto* bm = GPTR(uint32_t, s.bitmap);
    if (STAGED) {
        // global layout: word = column * bmStride + y (bmStride a multiple of 4): 16-byte pieces of four rows
        const int q4 = c.bmStride >> 2, n4 = ncol * q4;
        const float rq = 1.0f / (float)q4;
        constexpr int kStage = 4;  // 16-byte loads in flight per thread
        for (int g0 = tid; g0 < n4; g0 += kClsThreads * kStage) {
            u32x4u w[kStage];
            int col[kStage], y[kStage];
#pragma unroll
            for (int t = 0; t < kStage; t++) {
                const int g = g0 + t * kClsThreads;
                col[t] = (int)(((float)g + 0.5f) * rq);
                y[t] = (g - col[t] * q4) << 2;
                w[t] = u32x4u{0u, 0u, 0u, 0u};
                if (g < n4) w[t] = *GPTR(u32x4u, bm + (size_t)col[t] * c.bmStride + y[t]);
            }
#pragma unroll
            for (int t = 0; t < kStage; t++) {
                if (g0 + t * kClsThreads < n4) {
                    lbm[(y[t] + 0) * ncolp + col[t]] = w[t][0];
                    lbm[(y[t] + 1) * ncolp + col[t]] = w[t][1];
                    lbm[(y[t] + 2) * ncolp + col[t]] = w[t][2];
                    lbm[(y[t] + 3) * ncolp + col[t]] = w[t][3];
                }
            }
        }
    }
    __syncthreads();
    const bool road_on = c.useRoad && s.has_plane && s.inlier_mask;
    const auto* uv = GPTR(double, s.uv);
    const int rowBuckets = c.sortClasses > 1 ? kClsBuckets / 4 : kClsBuckets;
    const double scale = (double)rowBuckets / (double)c.H;
    // occupied cells of a window, counted in the (staged) bitmap
    auto count_window = [&](int x0, int y0, int nx, int ny) -> int {
        const int cx = x0 >> 5, sh = x0 & 31;
        const uint32_t colmask = (nx >= 32) ? ~0u : ((1u << nx) - 1u);
        int k = 0;
        if (STAGED) {
            for (int r = 0; r < ny; r++) {
                const int y = y0 + r;
                const uint32_t lo = lbm[y * ncolp + cx], hi = lbm[y * ncolp + cx + 1];
                k += __popc(__builtin_amdgcn_alignbit(hi, lo, (uint32_t)sh) & colmask);
            }
        } else {
            // in place: the rows of a word column are contiguous - four rows per 16-byte load, as the feature kernel's
            // scan reads them (the slack behind a column covers the rows read past the window)
            const auto* c0 = bm + (size_t)cx * c.bmStride + y0;
            const auto* c1 = c0 + c.bmStride;
            for (int r0 = 0; r0 < ny; r0 += 4) {
                const u32x4u a = *GPTR(u32x4u, c0 + r0), b = *GPTR(u32x4u, c1 + r0);
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (r0 + q < ny) k += __popc(__builtin_amdgcn_alignbit(b[q], a[q], (uint32_t)sh) & colmask);
            }
        }
        return k;
    };
    // class of feature i: CLS_DEAD (fewer than radiusSearch_count_min neighbours, DepthEstimator.cpp:680), CLS_OVF
    // (a window wider than 32 cells, or the wave-only mode: handled by k_feature_wave), or its row bucket (live).
    // (Sorting the live features by list length as well, so that the lanes of a wavefront iterate equally long, was
    // measured: the fused kernel gained what the extra counting cost here.)
    auto classify = [&](double u, double v) -> int {
        int x0, y0, nx, ny;
        int k1 = 0;
        if (window_bounds(c, u, v, c.halfX1, c.halfY1, x0, y0, nx, ny)) {
            if (nx > 32) return CLS_OVF;
            k1 = count_window(x0, y0, nx, ny);
            if (!c.threadPath)  // wave-only processing: dead features are still settled here
                return ((unsigned)k1 < c.countMin) ? (int)CLS_DEAD : (int)CLS_OVF;
            if (road_on) {
                int a0, a1, anx, any_;
                if (window_bounds(c, u, v, c.halfX2, c.halfY2, a0, a1, anx, any_) && anx > 32) return CLS_OVF;
            }
            // Lists beyond the fused kernel's capacities go straight to the wave kernel (dense clouds: every feature):
            // the narrow count is known; the scanned (road) window holds ~2.6x as many cells, so it is counted only
            // when the narrow one is already long.
            if (k1 > c.kMain || k1 > c.kTotal) return CLS_OVF;
            if (road_on && (3 * k1 > c.k1max || 4 * k1 > c.kTotal)) {
                int a0, a1, anx, any_;
                if (window_bounds(c, u, v, c.halfX2, c.halfY2, a0, a1, anx, any_)) {
                    const int k2 = count_window(a0, a1, anx, any_);
                    if (k2 > c.k1max || k1 + k2 > c.kTotal) return CLS_OVF;   // (both lists share kTotal entries of LDS)
                }
            }
        }
        if ((unsigned)k1 < c.countMin) return CLS_DEAD;
        if (!c.threadPath) return CLS_OVF;
        // Queue order: by neighbour-count class first (longest lists first), by image row inside a class.  The lanes of a
        // wavefront of the fused kernel iterate to the longest list among them - scan rounds, key / point round trips,
        // histogram passes, the O(n^2) corner search - so wavefronts of like features waste the fewest lane-iterations; on
        // a dense cloud (128 beams: 2 ... 21 neighbours, mean 8) that is worth more than the row order alone.  A 64-beam
        // cloud keeps (nearly) every feature in the first class: row order as before.
        int b = 0;
        if (v > 0.0) b = (v < (double)c.H) ? (int)(v * scale) : rowBuckets - 1;
        b = b < rowBuckets ? b : rowBuckets - 1;
        const int cls = (k1 <= 8) ? 0 : ((k1 <= 12) ? 1 : ((k1 <= 16) ? 2 : 3));
        return (c.sortClasses > 1 ? (3 - cls) * rowBuckets : 0) + b;
    };
    auto settle = [&](long long i, int cls) {  // results / overflow entries of the features that are not live
        if (cls == CLS_DEAD) {
            GPTRW(double, s.depth)[i] = -1.0;
            if (s.type) GPTRW(int32_t, s.type)[i] = MLD_RadiusSearchInsufficientPoints;
        } else if (cls == CLS_OVF) {
            const int pos = atomicAdd(&ctr[0], 1);
            GPTRW(int32_t, s.ovf_queue)[2 * (size_t)pos] = (int32_t)i;
            GPTRW(int32_t, s.ovf_queue)[2 * (size_t)pos + 1] = -1;
        }
    };
    int kept[kClsKeep];
    {
        // the features of the first pass are fetched before the barrier-free classification loop starts
        double fu[kClsKeep], fv[kClsKeep];
#pragma unroll
        for (int q = 0; q < kClsKeep; q++) {
            const long long i = tid + (long long)q * kClsThreads;
            fu[q] = 0.0;
            fv[q] = 0.0;
            if (i < Fn) {
                fu[q] = uv[2 * i];
                fv[q] = uv[2 * i + 1];
            }
        }
#pragma unroll
        for (int q = 0; q < kClsKeep; q++) {
            const long long i = tid + (long long)q * kClsThreads;
            kept[q] = (i < Fn) ? classify(fu[q], fv[q]) : (int)CLS_NONE;
            if (kept[q] >= 0) atomicAdd(&hist[kept[q]], 1);
            if (i < Fn) settle(i, kept[q]);
        }
    }
    for (long long i = tid + (long long)kClsKeep * kClsThreads; i < Fn; i += kClsThreads) {
        const int cls = classify(uv[2 * i], uv[2 * i + 1]);
        if (cls >= 0) atomicAdd(&hist[cls], 1);
        settle(i, cls);
    }
    __syncthreads();
    // exclusive scan of the bucket counts: kPer buckets per thread, wave scan, block offsets
    constexpr int kPer = kClsBuckets / kClsThreads;
    static_assert(kPer >= 1 && kPer * kClsThreads == kClsBuckets, "bucket count must be a multiple of the block size");
    int loc[kPer], sum = 0;
#pragma unroll
    for (int q = 0; q < kPer; q++) {
        loc[q] = hist[tid * kPer + q];
        sum += loc[q];
    }
    int incl = sum;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == kWave - 1) wsum[wave] = incl;
    __syncthreads();
    int base = incl - sum;
    for (int q = 0; q < wave; q++) base += wsum[q];
    if (tid == kClsThreads - 1) {
        *GPTRW(int32_t, s.live_count) = base + sum;  // number of live features
        *GPTRW(int32_t, s.ovf_count) = ctr[0];       // the later kernels append with atomics
    }
#pragma unroll
    for (int q = 0; q < kPer; q++) {
        hist[tid * kPer + q] = base;
        base += loc[q];
    }
    __syncthreads();
    auto* live = GPTRW(int32_t, s.live_queue);
#pragma unroll
    for (int q = 0; q < kClsKeep; q++)
        if (kept[q] >= 0) live[atomicAdd(&hist[kept[q]], 1)] = (int32_t)(tid + q * kClsThreads);
    for (long long i = tid + (long long)kClsKeep * kClsThreads; i < Fn; i += kClsThreads) {
        const int cls = classify(uv[2 * i], uv[2 * i + 1]);
        if (cls >= 0) live[atomicAdd(&hist[cls], 1)] = (int32_t)i;
    }
}

// One wavefront that ends when `*counter` has reached `target` (wrap-safe), or after a bounded number of polls: the
// kernels queued behind it on its stream start ~2 us after the last counted block instead of the ~15 us a cross-stream
// event takes.  Scheduling hint only - nothing depends on it for correctness, hence the bound instead of a guarantee.
__global__ __launch_bounds__(kWave) void k_gate(const uint32_t* __restrict__ counter, uint32_t target, int max_polls) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < max_polls; i++) {
        const uint32_t v = __hip_atomic_load(GPTR(uint32_t, counter), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int32_t)(v - target) >= 0) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

// Do two streams run side by side?  (The HIP runtime multiplexes a process's streams on a few hardware queues - four per
// priority by default - and two streams that share a queue execute strictly one kernel after the other: two contexts meant
// to overlap would silently serialise.)  k_probe_wait, on one stream, polls a word that k_probe_set, launched AFTER it on
// the other stream, sets: on distinct queues the setter runs while the waiter polls (result 1 within microseconds); on a
// shared queue the setter cannot start before the waiter has given up (result 2 after max_polls x ~1 us).
__global__ __launch_bounds__(kWave) void k_probe_wait(const uint32_t* flag, uint32_t* result, int max_polls) {
    if (threadIdx.x != 0) return;
    uint32_t seen = 2u;
    for (int i = 0; i < max_polls; i++) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) {
            seen = 1u;
            break;
        }
        __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(result, seen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_probe_set(uint32_t* flag) {
    if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// A step's small tables (slot / sequence descriptors, constants) from the context's pinned host block to device memory,
// in stream order, by load / store: no DMA engine takes part (see upload_small in mld_api.hip).
__global__ __launch_bounds__(256) void k_upload(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src_host, int n_words) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_words) dst[i] = __builtin_nontemporal_load(src_host + i);
}

#ifndef MLD_KEY_BATCH_F
#define MLD_KEY_BATCH_F 8
#endif
constexpr int kKeyBatchF = MLD_KEY_BATCH_F;  // map keys fetched per round trip by the fused kernel
#ifndef MLD_DENSE_BATCH_MUL
#define MLD_DENSE_BATCH_MUL 2
#endif
constexpr int kDenseBatchMul = MLD_DENSE_BATCH_MUL;  // the DENSE 2 instantiation fetches that many times the keys / road points

// One scan of the window (x0, y0, nx, ny) through the occupancy bitmap; entries in the reference's row-major order,
// bit 31 set for the cells that also lie inside the narrow window (xn0, yn0, nxn, nyn).  The lane's list ends up
// holding ORIGINAL POINT INDICES (low 24 bits) | flag.  Returns the entry count (may exceed c.k1max: overflow), the
// number of flagged entries in kflag.
// While the keys are in registers the scan also writes the narrow list (the flagged entries' point indices, same order)
// BEHIND the wide lists of the wavefront: it starts at row `nbase` (wave-uniform, so that every list access keeps its
// scalar address arithmetic; normally the length of the longest wide list among the lanes) of the c.kTotal rows of LDS and
// may hold `nroom` = min(c.kMain, c.kTotal - nbase) entries - the two lists share one budget: long road windows leave less
// room for the narrow lists and the other way round.  It also gathers what the road fallback wants to know about the lane's list
// before it touches a point: states = bit 0 "a far point", bit 1 "an unsure point" (k_project_scatter's plane states).
template <int KB>  // map keys fetched per round trip (KB; twice that in the DENSE 2 instantiation)
__device__ __forceinline__ int scan_window_flagged(const Calib& c, const SlotRef& s, int x0, int y0, int nx, int ny,
                                                   int xn0, int yn0, int nxn, int nyn, uint32_t* lst,
                                                   int lane, int& kflag, uint32_t& states, int& nbase, int& nroom ST_ARG) {
    const int nymax = uniform(wave_max_i32(ny));
    const auto* bm = GPTR(uint32_t, s.g.bitmap);
    // windows are at most 32 cells wide here (k_classify routes wider ones to the wave kernel): 32-bit row masks
    const uint32_t colmask = (nx >= 32) ? ~0u : ((1u << nx) - 1u);
    // columns of the narrow window relative to x0
    const uint32_t nmask = (nxn > 0) ? (((nxn >= 32) ? ~0u : ((1u << nxn) - 1u)) << (xn0 - x0)) : 0u;
    int k = 0, kf = 0;
    const auto* col0 = bm + (size_t)(x0 >> 5) * (size_t)c.bmStride + (size_t)y0;
    const auto* col1 = col0 + c.bmStride;
    const int sh = x0 & 31;
    const int nxmax = uniform(wave_max_i32(nx));
    const int kclamp = c.kTotal - 1;  // last row of a lane's column
    const bool packed = nxmax <= 16;  // (wave-uniform) which of the two walks below wrote the codes
    for (int r0 = 0; r0 < nymax; r0 += 16) {
        // sixteen rows (four 16-byte pieces from each of the two word columns) in ONE round trip
        u32x4u a[4], b2[4];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            a[g] = u32x4u{0u, 0u, 0u, 0u};
            b2[g] = u32x4u{0u, 0u, 0u, 0u};
            if (r0 + 4 * g < nymax) {  // wave-uniform
                if (r0 + 4 * g < ny) {
                    a[g] = *GPTR(u32x4u, col0 + r0 + 4 * g);
                    b2[g] = *GPTR(u32x4u, col1 + r0 + 4 * g);
                }
            }
        }
        ST_USE_U32(a[0][0]);
        ST_USE_U32(b2[0][0]);
        ST_USE_U32(a[3][0]);
        ST_USE_U32(b2[3][0]);
        ST_MARK(2);
        if (packed) {
            // Windows up to 16 cells wide (the C0 parameter set: 7 and 13): two rows are packed into one 32-bit word
            // (16 bits each) and the set bits of a word are walked in one loop.  A wavefront iterates to the largest
            // bit count among its lanes, and lanes hit the LiDAR rings in different rows: per pair of rows that maximum
            // is about the points of one ring, per single row it is the same, so this halves the iterations.
#pragma unroll
            for (int h = 0; h < 8; h++) {
                if (r0 + 2 * h < nymax) {  // wave-uniform
                    const int qa = 2 * h, qb = 2 * h + 1;
                    const uint32_t va = __builtin_amdgcn_alignbit(b2[qa >> 2][qa & 3], a[qa >> 2][qa & 3], (uint32_t)sh);
                    const uint32_t vb = __builtin_amdgcn_alignbit(b2[qb >> 2][qb & 3], a[qb >> 2][qb & 3], (uint32_t)sh);
                    const uint32_t ba = ((r0 + qa) < ny) ? (va & colmask) : 0u;
                    const uint32_t bb = ((r0 + qb) < ny) ? (vb & colmask) : 0u;
                    uint32_t b = ba | (bb << 16);
                    const int rowa = y0 + r0 + qa;
                    const uint32_t fa = (rowa >= yn0 && rowa < yn0 + nyn) ? nmask : 0u;
                    const uint32_t fb = (rowa + 1 >= yn0 && rowa + 1 < yn0 + nyn) ? nmask : 0u;
                    const uint32_t flags = fa | (fb << 16);
                    // (the walk stores a CODE per set bit - row pair and bit position; the key loop below turns it into
                    //  the cell index, eight entries at a time, instead of this serial loop doing it bit by bit.  A lane
                    //  past its capacity keeps counting and writes to the last row of its column: it is handed over)
                    const uint32_t pairbase = (uint32_t)((r0 >> 1) + h) << 5;  // wave-uniform
                    kf += __popc(b & flags);   // (counted per word, not per bit)
                    while (b) {
                        const uint32_t p = (uint32_t)__ffs((int)b) - 1u;
                        b &= b - 1u;
                        LST(min(k, kclamp)) = pairbase | p | ((flags >> p) << 31);
                        k++;
                    }
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (r0 + q < nymax) {
                // (hi:lo) >> sh, low 32 bits: the window's cells of this row
                const uint32_t v = __builtin_amdgcn_alignbit(b2[q >> 2][q & 3], a[q >> 2][q & 3], (uint32_t)sh);
                uint32_t b = ((r0 + q) < ny) ? (v & colmask) : 0u;
                const int row = y0 + r0 + q;
                const uint32_t rowcode = (uint32_t)(r0 + q) << 5;  // wave-uniform
                const uint32_t rowflags = (row >= yn0 && row < yn0 + nyn) ? nmask : 0u;
                kf += __popc(b & rowflags);
                while (b) {
                    const uint32_t col = (uint32_t)__ffs((int)b) - 1u;
                    b &= b - 1u;
                    LST(min(k, kclamp)) = rowcode | col | ((rowflags >> col) << 31);
                    k++;
                }
            }
        }
        }
    }
    ST_MARK(3);
    // cell index -> original point index (every set bit has a key of the current tag)
    const int kk = k <= c.k1max ? k : 0;
    const int kmax = uniform(wave_max_i32(kk));
    const auto* mp = GPTR(uint32_t, s.g.map);
    int kn = 0;
    uint32_t st_any = 0u;
    // Where the narrow lists start: every lane has its own column of the c.kTotal rows, so a lane fits iff its wide list
    // ends at or before the (wave-uniform) row nb and its narrow list within the c.kTotal - nb rows behind it; a lane that
    // does not fit is handed to the wave kernel ALONE.  Two candidates - behind the longest wide list, or as far up as the
    // longest narrow list allows - and the one that hands over fewer lanes (on 64-beam clouds the first, with nobody
    // handed over; a lane beside a close object - 30 returns in its road window - would otherwise take its neighbours along).
    int nb = kmax;
    {
        const int kfm = min(uniform(wave_max_i32(k <= c.k1max ? kf : 0)), c.kMain);
        const int nb2 = max(c.kTotal - kfm, 0);
        if (nb2 < nb) {
            const int lost1 = (int)__popcll(__ballot(k <= c.k1max && kf <= c.kMain && kf > c.kTotal - nb));
            const int lost2 = (int)__popcll(__ballot(k <= c.k1max && kf <= c.kMain && k > nb2));
            if (lost2 < lost1) nb = nb2;
        }
    }
    uint32_t* const nl = lst + nb * kWave;
    nbase = nb;
    nroom = min(c.kMain, c.kTotal - nb);
    const bool fits = kk <= nb;  // (a lane whose wide list reaches beyond nb is handed over: it writes no narrow entry)
    for (int e0 = 0; e0 < kmax; e0 += KB) {
        uint32_t cell[KB], key[KB];
#pragma unroll
        for (int q = 0; q < KB; q++) {
            // code of the bit walk -> cell index (x + y W) | narrow flag
            const uint32_t code = LST_IF(e0 + q < kk, e0 + q, c.k1max);
            const uint32_t v = code & 0x7FFFFFFFu;
            const uint32_t row = packed ? (((v >> 5) << 1) + ((v >> 4) & 1u)) : (v >> 5);
            const uint32_t col = packed ? (v & 15u) : (v & 31u);
            cell[q] = ((uint32_t)__mul24(y0 + (int)row, c.W) + (uint32_t)(x0 + (int)col)) | (code & kEntNarrow);
        }
#pragma unroll
        // (an unconditional load from a selected address - cell 0 beyond the lane's list - instead of a predicated one: a
        //  select per entry where a predicated load is a branch per entry)
        for (int q = 0; q < KB; q++) key[q] = MLD_DIAG_KEY(mp[(e0 + q < kk) ? (cell[q] & 0x7FFFFFFFu) : 0u], s.g.tag, cell[q]);
#pragma unroll
        for (int q = 0; q < KB; q++)
            if (e0 + q < kk) {
                const uint32_t idx = key_index(key[q]), st = key[q] & 3u;
                LST(e0 + q) = idx | (st << kEntStateShift) | (cell[q] & kEntNarrow);
                st_any |= (st == kPtFar ? 1u : 0u) | (st == kPtUnsure ? 2u : 0u);
                if ((cell[q] & kEntNarrow) && kn < nroom && fits) {
                    nl[kn * kWave + lane] = idx;
                    kn++;
                }
            }
    }
    ST_MARK(4);
    kflag = kf;
    states = st_any;
    return k;
}

// One lane per LIVE feature (queue written by k_classify).  LDS per wave: c.kTotal entries per lane, transposed
// [entry][lane]: the lanes' wide lists (k2 <= c.k1max entries) and, behind the longest of them, their narrow lists
// (k1 <= c.kMain entries).
// (Rounds 1-5 kept the two lists in separate regions of k1max + kMain entries: 14 KB per wavefront at the default
// capacities = 11 wavefronts per CU; sharing one region the same lists fit 10 KB = 16 per CU, four per SIMD.)
#ifndef MLD_FUSED_WAVES
#define MLD_FUSED_WAVES 3
#endif
#ifdef MLD_FUSED_VGPRS
#define MLD_FUSED_ATTR __attribute__((amdgpu_num_vgpr(MLD_FUSED_VGPRS)))
#else
#define MLD_FUSED_ATTR
#endif
// DENSE = 0: the default instantiation: four wavefronts per SIMD (<= 128 registers; 10 KB of LDS at the default list budget).
// DENSE = 2: the instantiation for long lists (dense clouds; list capacities beyond the default): two wavefronts per SIMD -
// all the LDS of such lists allows - and therefore up to 256 registers, which the in-register corner search of main_tail uses.
// DENSE = 1: the same kernel for long lists within 168 registers (three wavefronts per SIMD's worth: tiers up to 16 points /
// 16 depths), for the shared-GPU mode - it leaves a third of the register file to the other context's projection.
template <int ROAD_MODE, int DENSE>
__global__ __launch_bounds__(kWave, DENSE == 2 ? 2 : (DENSE == 1 ? MLD_FUSED_WAVES : 4)) MLD_FUSED_ATTR void k_feature_fused(const SlotDesc* __restrict__ slots,
                                                         const Calib* __restrict__ calib, int n_slots, int per_slot) {
    extern __shared__ __align__(16) unsigned char smem[];
    // The per-context constants stay in device memory as well (mld_ctx::d_calib): a field is fetched by a scalar load
    // where a stage uses it.  Passed by value, the compiler fetched all 600 bytes in the prologue and carried - spilled -
    // them through the kernel.
    const Calib& c = *calib;
    // One wavefront of features per block, n_slots * per_slot blocks: straight-line code.  (A grid-stride loop around
    // this body - there for a persistent-grid experiment that lost - made the compiler hoist every wave-uniform quantity
    // of the body in front of the loop: ~190 scalars written to vector lanes in the prologue and 850 v_readlane in the
    // body to get them back, in a kernel that is bound by VALU issue.)
    {
        const int w = (int)blockIdx.x;
        if (w >= n_slots * per_slot) return;
        int slot, j;
        decode_block(w, c.xcdAware ? n_slots : -n_slots, per_slot, slot, j);
        // (nothing of the descriptor but what a stage needs is ever held: see SlotRef.  The map tag is not among it - every
        // bit of the occupancy bitmap has a key of the current tag.)
        const SlotDesc& sg = slots[slot];
        const auto* pdev = GPTR(PlaneDev, sg.plane_dev);  // a plane estimated on the device overrides the descriptor's
        const SlotRef s = {sg, sg.plane_dev ? reinterpret_cast<const double*>(sg.plane_dev) : sg.prior_n,
                           sg.plane_dev ? pdev->has_plane : sg.has_plane};
        const int count = *GPTR(int32_t, sg.live_count);
        const int e0 = j * kWave;
        if (e0 >= count) return;
        ST_BEGIN(0);
        const int lane = threadIdx.x;
        const bool active = e0 + lane < count;
        uint32_t* lst = reinterpret_cast<uint32_t*>(smem);  // wide list; the narrow list follows it (per lane)
        long long f = 0;
        double myu = 0, myv = 0;
        if (active) {
            f = (long long)GPTR(int32_t, sg.live_queue)[e0 + lane];
            const auto* q = GPTR(double, sg.uv) + 2 * f;
            myu = q[0];
            myv = q[1];
        }
        ST_USE_F64(myu);
        ST_USE_F64(myv);
        ST_MARK(1);
        const bool road_on = c.useRoad && s.has_plane && sg.inlier_mask;
        // narrow window (DepthEstimator.cpp:509, scale 1 x 1) and the window that is scanned: the road window
        // (:585, scale 2.0 x 1.5), a superset, when the fallback is on
        int xn0 = 0, yn0 = 0, nxn, nyn, x0 = 0, y0 = 0, nx, ny;
        const bool hasn = active && window_bounds(c, myu, myv, c.halfX1, c.halfY1, xn0, yn0, nxn, nyn);
        if (!hasn) {
            nxn = 0;
            nyn = 0;
        }
        if (road_on) {
            if (!(active && window_bounds(c, myu, myv, c.halfX2, c.halfY2, x0, y0, nx, ny))) {
                nx = 0;
                ny = 0;
            }
        } else {
            x0 = xn0;
            y0 = yn0;
            nx = nxn;
            ny = nyn;
        }
        int k1 = 0;
        uint32_t list_states = 0u;
        int nbase = 0, nroom = 0;
        const int k2 = scan_window_flagged<DENSE == 2 ? kDenseBatchMul * kKeyBatchF : kKeyBatchF>(c, s, x0, y0, nx, ny, xn0, yn0, nxn, nyn, lst,
                                                                                    lane, k1, list_states, nbase, nroom ST_PASS);
        // (k_classify keeps windows wider than 32 cells out of the live queue)
        bool overflow = active && (k2 > c.k1max || k2 > nbase || k1 > nroom);
        // the narrow lists: behind the longest wide list of the wavefront.  (No room for a single entry: every lane with
        // a narrow entry has overflowed; the reads below are clamped to the last entry of the region.)
        uint32_t* const nl = lst + min(nbase, c.kTotal - 1) * kWave;
        const int ncap = max(nroom, 1);  // entries the narrow list may address
        int ovf_code = -1;
        // (the narrow list - the flagged entries, same order - was written by the scan)
        ST_MARK(5);
        int mytype = MLD_Unspecified;
        double mydepth = -1.0;
        bool live = active && !overflow;
        if (live && (unsigned)k1 < c.countMin) {  // DepthEstimator.cpp:680 (k_classify has filtered these out already)
            mytype = MLD_RadiusSearchInsufficientPoints;
            live = false;
        }
        // fewer neighbours than the histogram's minimum bin count: no bin can become a maximum, FilterPointsMinDistBlob
        // returns false on every path (HistogramPointDepth.cpp:53,84,95)
        if (live && c.useHist && c.minCount >= 1 && k1 < c.minCount) {
            mytype = MLD_HistogramNoLocalMax;
            live = false;
        }
        {
            bool ovf1 = false;
            main_after_scan<DENSE>(c, s, nl, lane, ncap, k1, live, myu, myv, mytype, mydepth, ovf1 ST_PASS);
            if (live && ovf1) overflow = true;  // (a list the per-thread corner search cannot take: wave kernel)
        }
        // ---------------- road fallback (DepthEstimator.cpp:578-597) on the list already scanned ----------------
        const bool cand = road_on && active && !overflow && (mytype != MLD_Success) &&
                          (mytype != MLD_RadiusSearchInsufficientPoints);
        if (__any(cand)) {
            const int resultOld = mytype;
            bool ovf2 = false;
            road_after_scan<ROAD_MODE, DENSE == 2 ? kDenseBatchMul * kRoadBatch : kRoadBatch>(c, s, lst, lane, cand, k2, list_states, myu, myv,
                                                                                mytype, mydepth, ovf2 ST_PASS);
            if (ovf2) {  // only the road part is redone by the wave kernel
                overflow = true;
                ovf_code = resultOld;
            }
        }
        enqueue_features(sg.ovf_queue, sg.ovf_count, overflow && active, lane, f, ovf_code);
        if (active) {
            GPTRW(double, sg.depth)[f] = mydepth;
            if (sg.type) GPTRW(int32_t, sg.type)[f] = mytype;
        }
        ST_MARK(12);
        ST_END();
    }
}

// set_all_depths_to_zero short-circuit (DepthEstimator.cpp:448-453)
__global__ void k_fill_zero(double* depth, int32_t* type, long long F) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < F) {
        depth[i] = -1.0;
        if (type) type[i] = 1;
    }
}

// Inlier index list -> bitmask keyed by original cloud index (replaces std::map<int,bool>, RansacPlane.h:116-122)
__global__ void k_build_mask(const int32_t* __restrict__ idx, long long n_inl, long long n_points, uint32_t* mask) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_inl) {
        int id = idx[i];
        if (id >= 0 && (long long)id < n_points) atomicOr(&mask[id >> 5], 1u << (id & 31));
    }
}

// ------------------------------------------------------------------------------------------------
// Tracklet gather / scatter either side of the path (tracklets_depth/src/tracklet_depth_module.cpp:23-169)
// ------------------------------------------------------------------------------------------------
constexpr int kTrkBlock = 1024;

// ExractNewTrackletFrames + CalculateFeatureDepths{Cur,Last}Frame feature marshalling (:23-117):
//   uv_cur[i]  = (int)newest feature of track i                       (every track, track order)
//   uv_last[r] = (int)previous feature of the r-th NEW track          (new tracks only, track order)
// The float -> std::pair<int,int> conversion of the reference truncates toward zero (TempTrackletFrame.h:7-8).
// One block; the order-preserving ranks of the new tracks come from a chunked ballot scan.
__global__ __launch_bounds__(kTrkBlock) void k_tracklet_gather(const float* __restrict__ u_new,
                                                             const float* __restrict__ v_new,
                                                             const float* __restrict__ u_old,
                                                             const float* __restrict__ v_old,
                                                             const uint8_t* __restrict__ is_new, long long n,
                                                             double* uv_cur, double* uv_last, int32_t* rank,
                                                             long long* n_new_out) {
    __shared__ int wsum[kTrkBlock / kWave];
    __shared__ int base_s;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    const int w = threadIdx.x / kWave;
    for (long long c0 = 0; c0 < n; c0 += kTrkBlock) {
        const long long i = c0 + threadIdx.x;
        const bool in = i < n;
        const bool nw = in && is_new[i] != 0;
        const unsigned long long m = __ballot(nw);
        if ((threadIdx.x & (kWave - 1)) == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int q = 0; q < w; q++) off += wsum[q];
        const int r = off + prefix_count(m);
        if (in) {
            uv_cur[2 * i] = (double)(int)u_new[i];
            uv_cur[2 * i + 1] = (double)(int)v_new[i];
            rank[i] = nw ? r : -1;
            if (nw) {
                uv_last[2 * (long long)r] = (double)(int)u_old[i];
                uv_last[2 * (long long)r + 1] = (double)(int)v_old[i];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int q = 0; q < kTrkBlock / kWave; q++) t += wsum[q];
            base_s += t;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *n_new_out = base_s;
}

// SaveFeatureDepths / convert_tracklets_to_matches_msg (:119-169, :209-259): depths back to the tracks as float32
// FeaturePoint.d (matches_msg_depth_ros/msg/FeaturePoint.msg:3-5).  d_last is written for new tracks only; without
// a previous cloud their previous-frame depth is -1 (:93-96).
__global__ void k_tracklet_scatter(const double* __restrict__ depth_cur, const int32_t* __restrict__ type_cur,
                                   const double* __restrict__ depth_last, const int32_t* __restrict__ type_last,
                                   const int32_t* __restrict__ rank, long long n, int have_last, float* d_cur_out,
                                   float* d_last_out, int32_t* type_cur_out, int32_t* type_last_out,
                                   const long long* n_new_dev = nullptr, long long* n_new_out = nullptr) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 && n_new_out) *n_new_out = *n_new_dev;  // (one-frame call: the count travels with the depths)
    if (i >= n) return;
    d_cur_out[i] = (float)depth_cur[i];
    if (type_cur_out) type_cur_out[i] = type_cur[i];
    const int r = rank[i];
    if (r >= 0) {
        d_last_out[i] = have_last ? (float)depth_last[r] : -1.0f;
        if (type_last_out) type_last_out[i] = have_last ? type_last[r] : (int32_t)MLD_Unspecified;
    }
}

// The same marshalling for MANY sequences at once (mld_tracklets_depths_device): block (c, s) handles tracks
// [c * 1024, (c + 1) * 1024) of sequence s.  The order-preserving rank of a new track = new tracks in the chunks before
// (recounted by every block from the flag bytes: at most a few KB) + the ballot scan inside the chunk.
__global__ __launch_bounds__(kTrkBlock) void k_tracklets_gather(const TrkSeq* __restrict__ seqs) {
    __shared__ int wsum[kTrkBlock / kWave];
    __shared__ int before_s;
    const TrkSeq q = seqs[blockIdx.y];
    const long long c0 = (long long)blockIdx.x * kTrkBlock;
    if (c0 >= q.n) return;
    const int w = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    if (threadIdx.x == 0) before_s = 0;
    __syncthreads();
    int mine = 0;
    for (long long i = threadIdx.x; i < c0; i += kTrkBlock) mine += q.is_new[i] != 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0 && mine) atomicAdd(&before_s, mine);
    const long long i = c0 + threadIdx.x;
    const bool in = i < q.n;
    const bool nw = in && q.is_new[i] != 0;
    const unsigned long long m = __ballot(nw);
    if (lane == 0) wsum[w] = __popcll(m);
    __syncthreads();
    int off = before_s;
    for (int k = 0; k < w; k++) off += wsum[k];
    const int r = off + prefix_count(m);
    if (in) {
        q.uv_cur[2 * i] = (double)(int)q.u_new[i];
        q.uv_cur[2 * i + 1] = (double)(int)q.v_new[i];
        q.rank[i] = nw ? r : -1;
        if (nw) {
            q.uv_last[2 * (long long)r] = (double)(int)q.u_old[i];
            q.uv_last[2 * (long long)r + 1] = (double)(int)q.v_old[i];
        }
    }
    if (c0 + kTrkBlock >= q.n && threadIdx.x == 0) {  // the sequence's last chunk knows the total
        int t = before_s;
        for (int k = 0; k < kTrkBlock / kWave; k++) t += wsum[k];
        *q.n_new = t;
    }
}

__global__ void k_tracklets_scatter(const TrkSeq* __restrict__ seqs) {
    const TrkSeq q = seqs[blockIdx.y];
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= q.n) return;
    q.d_cur_out[i] = (float)q.depth_cur[i];
    if (q.type_cur_out) q.type_cur_out[i] = q.type_cur[i];
    const int r = q.rank[i];
    if (r >= 0) {
        q.d_last_out[i] = q.have_last ? (float)q.depth_last[r] : -1.0f;
        if (q.type_last_out) q.type_last_out[i] = q.have_last ? q.type_last[r] : (int32_t)MLD_Unspecified;
    }
}

// ------------------------------------------------------------------------------------------------
// Lazy debug getters: full PointcloudData (PointcloudData.h:13-68)
// ------------------------------------------------------------------------------------------------
// cam: 3 x N col-major, img: 2 x N, vis: N flags (in range && strict bounds)
__global__ void k_project_full(SlotDesc s, Calib c, double* cam, double* img, int32_t* vis) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= s.n) return;
    double x, y, z;
    load_point(s, i, x, y, z);
    V3 pc = lidar_to_cam(c, x, y, z);
    double u, v;
    project(c, pc, u, v);
    cam[3 * i] = pc.x;
    cam[3 * i + 1] = pc.y;
    cam[3 * i + 2] = pc.z;
    img[2 * i] = u;
    img[2 * i + 1] = v;
    const double Wd = (double)c.W, Hd = (double)c.H;
    bool inr = (u >= 0.) && (u <= Wd) && (v >= 0.) && (v <= Hd);
    bool strict = (u > 0.) && (u < Wd) && (v > 0.) && (v < Hd);
    vis[i] = (inr && strict) ? 1 : 0;
}

constexpr int kScanBlock = 1024;

// Exclusive scan of vis[] in three steps (block sums, scan of block sums by one block, final ranks).
__global__ __launch_bounds__(kScanBlock) void k_scan_block_sums(const int32_t* __restrict__ vis, long long n,
                                                               int32_t* block_sums) {
    __shared__ int wsum[kScanBlock / kWave];
    long long i = (long long)blockIdx.x * kScanBlock + threadIdx.x;
    int v = (i < n) ? vis[i] : 0;
    int cnt = __popcll(__ballot(v != 0));
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < kScanBlock / kWave; w++) t += wsum[w];
        block_sums[blockIdx.x] = t;
    }
}
// Exclusive scan of the block sums in place + their total, by ONE block of kScanBlock threads (n_blocks <= 16384 for
// 16.7 M points): a thread owns a run of consecutive entries, the runs' totals are scanned across the block.  (A single
// thread walking the array - a dependent global load and store per entry - took 60 us for the 128 blocks of a 131 072-point
// cloud, twice per semantic plane.)
__global__ __launch_bounds__(kScanBlock) void k_scan_sums(int32_t* block_sums, int n_blocks, int32_t* total) {
    __shared__ int wsum[kScanBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid / kWave;
    const int per = (n_blocks + kScanBlock - 1) / kScanBlock;
    const int b0 = tid * per;
    int loc = 0;
    for (int q = 0; q < per; q++)
        if (b0 + q < n_blocks) loc += block_sums[b0 + q];
    int incl = loc;
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
    }
    if (lane == kWave - 1) wsum[w] = incl;
    __syncthreads();
    int acc = incl - loc;
    for (int q = 0; q < w; q++) acc += wsum[q];
    for (int q = 0; q < per; q++)
        if (b0 + q < n_blocks) {
            const int t = block_sums[b0 + q];
            block_sums[b0 + q] = acc;
            acc += t;
        }
    if (tid == kScanBlock - 1) *total = acc;
}
// rank[i] = visible index of point i (valid where vis[i]); also emits the compacted lists.
__global__ __launch_bounds__(kScanBlock) void k_scan_final(const int32_t* __restrict__ vis, long long n,
                                                          const int32_t* __restrict__ block_off,
                                                          const double* __restrict__ img, int32_t* rank,
                                                          int32_t* point_index, double* img_vis) {
    __shared__ int wsum[kScanBlock / kWave];
    long long i = (long long)blockIdx.x * kScanBlock + threadIdx.x;
    int v = (i < n) ? vis[i] : 0;
    unsigned long long m = __ballot(v != 0);
    int within = prefix_count(m);
    int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[w] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    int r = block_off[blockIdx.x] + woff + within;
    if (i < n) {
        rank[i] = v ? r : -1;
        if (v) {
            point_index[r] = (int32_t)i;
            img_vis[2 * (long long)r] = img[2 * i];
            img_vis[2 * (long long)r + 1] = img[2 * i + 1];
        }
    }
}
// Reference-format pixel map: visible index of the winning point, or -1.
__global__ void k_export_map(const uint32_t* __restrict__ map, uint32_t tag, const int32_t* __restrict__ rank,
                             long long cells, int32_t* out) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cells) return;
    uint32_t key = map[i];
    int32_t r = -1;
    if ((key >> kTagShift) == tag) r = rank[key_index(key)];
    out[i] = r;
}

}  // namespace mld
