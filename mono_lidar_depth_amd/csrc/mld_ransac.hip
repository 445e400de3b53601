// Per-frame ground plane on the GPU: RansacPlane::CalculateInliersPlane (monolidar_fusion/src/RansacPlane.cpp:41-140).
//
// SURVEY.md §8(f)-1, a "next" row: the plane used to be an input only.  The pipeline of the reference is kept —
// optional z pass-through (:57-64), sub-sample to 6000 points (:66-74), perpendicular-plane RANSAC around the z axis
// with PCL's adaptive iteration bound (:94-108), least-squares refinement and re-selection within
// ransac_plane_refinement_treshold of the UNREFINED model (:117-126), inlier lookup keyed by original index
// (:128-133, here the bitmask the feature kernel reads).  PARITY UNPINNED against PCL: the reference's RandomSample
// is time-seeded, i.e. not reproducible; draws here come from a counter-based hash (bit-identical to the CPU
// restatement used by the tests), so all hypotheses are independent and evaluated in parallel, and PCL's
// sequential stopping rule is applied afterwards to the per-draw inlier counts.
#include "mld_device.h"

namespace mld {
namespace ransac {

constexpr int kSample = 6000;  // RansacPlane.cpp:32
constexpr int kPartials = 256;

__host__ __device__ inline uint32_t mix(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u;
    h ^= b + 0x85EBCA6Bu + (h << 6) + (h >> 2);
    h ^= c * 0xC2B2AE35u + (h << 6) + (h >> 2);
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

struct Model {
    float c[4];
    int degenerate;
    int valid;
};

// sac_model_plane computeModelCoefficients + sac_model_perpendicular_plane isModelValid, float arithmetic
__device__ inline Model plane_from(const float* p0, const float* p1, const float* p2) {
    Model m;
    float a0 = p1[0] - p0[0], a1 = p1[1] - p0[1], a2 = p1[2] - p0[2];
    float b0 = p2[0] - p0[0], b1 = p2[1] - p0[1], b2 = p2[2] - p0[2];
    float r0 = a0 / b0, r1 = a1 / b1, r2 = a2 / b2;
    m.degenerate = ((r0 == r1) && (r2 == r1)) ? 1 : 0;
    float n0 = a1 * b2 - a2 * b1, n1 = a2 * b0 - a0 * b2, n2 = a0 * b1 - a1 * b0;
    float nn = sqrtf(n0 * n0 + n1 * n1 + n2 * n2);
    n0 /= nn;
    n1 /= nn;
    n2 /= nn;
    m.c[0] = n0;
    m.c[1] = n1;
    m.c[2] = n2;
    m.c[3] = -1.0f * (n0 * p0[0] + n1 * p0[1] + n2 * p0[2]);
    if (!(nn > 0.0f) || !isfinite(nn)) m.degenerate = 1;
    m.valid = (!m.degenerate && (fabs((double)n2) >= 0.984807753012208)) ? 1 : 0;  // cos(pi/18): 10 degrees
    return m;
}
__device__ inline float plane_dist(const float c[4], const float* p) {
    return fabsf(c[0] * p[0] + c[1] * p[1] + c[2] * p[2] + c[3]);
}
__device__ inline Model draw_model(const float* sp, int S, uint32_t seed, int d) {
    uint32_t a = mix(seed, (uint32_t)d, 1u) % (uint32_t)S;
    uint32_t b = mix(seed, (uint32_t)d, 2u) % (uint32_t)S;
    uint32_t c = mix(seed, (uint32_t)d, 3u) % (uint32_t)S;
    if (b == a) b = (b + 1) % (uint32_t)S;
    while (c == a || c == b) c = (c + 1) % (uint32_t)S;
    return plane_from(sp + 3 * a, sp + 3 * b, sp + 3 * c);
}

// pcl::PassThrough on z with float limits (:57-64): flag = finite && lo <= z <= hi
__global__ void k_rs_flags(const unsigned char* cloud, long long n, int stride, float lo, float hi, int32_t* flags) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = reinterpret_cast<const float*>(cloud + (size_t)i * stride);
    float x = p[0], y = p[1], z = p[2];
    flags[i] = (isfinite(x) && isfinite(y) && isfinite(z) && !(z < lo) && !(z > hi)) ? 1 : 0;
}
// order-preserving compaction of the flagged indices (block offsets from k_scan_block_sums / k_scan_sums)
__global__ __launch_bounds__(1024) void k_rs_compact(const int32_t* __restrict__ flags, long long n,
                                                     const int32_t* __restrict__ block_off, int32_t* cand) {
    __shared__ int wsum[1024 / kWave];
    long long i = (long long)blockIdx.x * 1024 + threadIdx.x;
    int v = (i < n) ? flags[i] : 0;
    unsigned long long m = __ballot(v != 0);
    int within = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) wsum[w] = __popcll(m);
    __syncthreads();
    int woff = 0;
    for (int q = 0; q < w; q++) woff += wsum[q];
    if (v) cand[block_off[blockIdx.x] + woff + within] = (int32_t)i;
}

// Stratified sub-sample (restatement of pcl::RandomSample, :66-74): sample[j] = cand[floor((j + u_j) M / 6000)].
// M is read from device memory when the candidates come from the pass-through.  Also gathers the xyz of the
// sample into a compact array and publishes S.
__global__ void k_rs_sample(const unsigned char* cloud, int stride, const int32_t* cand, const int32_t* M_dev,
                            long long M_host, uint32_t seed, int32_t* sample_idx, float* sp, int32_t* S_out) {
    const long long M = cand ? (long long)*M_dev : M_host;
    const int S = M > kSample ? kSample : (int)M;
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) *S_out = S;
    if (j >= S) return;
    long long pos = j;
    if (M > kSample) {
        double u = (double)mix(seed, (uint32_t)j, 0x5A17u) * (1.0 / 4294967296.0);
        pos = (long long)(((double)j + u) * (double)M / (double)kSample);
        if (pos > M - 1) pos = M - 1;
    }
    int32_t idx = cand ? cand[pos] : (int32_t)pos;
    sample_idx[j] = idx;
    const float* p = reinterpret_cast<const float*>(cloud + (size_t)idx * stride);
    sp[3 * j] = p[0];
    sp[3 * j + 1] = p[1];
    sp[3 * j + 2] = p[2];
}

// One wavefront per draw: inlier count of the hypothesis (countWithinDistance), -1 for a skipped (degenerate) draw.
__global__ __launch_bounds__(kWave) void k_rs_hypotheses(const float* __restrict__ sp, const int32_t* S_dev, uint32_t seed,
                                                        int n_draws, double thr, int32_t* counts) {
    const int d = blockIdx.x;
    if (d >= n_draws) return;
    const int S = *S_dev;
    if (S < 3) {
        if (threadIdx.x == 0) counts[d] = -1;
        return;
    }
    const Model m = draw_model(sp, S, seed, d);
    int cnt = 0;
    if (m.valid) {
        for (int j = threadIdx.x; j < S; j += kWave)
            if ((double)plane_dist(m.c, sp + 3 * j) < thr) cnt++;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (threadIdx.x == 0) counts[d] = m.degenerate ? -1 : cnt;
}

struct Result {
    float coeffs[4];     // final (refined) coefficients
    float ransac[4];     // RANSAC model (unrefined)
    int32_t status;      // 0 ok, 1 no model / too few points
    int32_t best_draw;
    int32_t best_count;
    int32_t n_inliers;   // final inlier count (bits set in the mask)
    int32_t iterations;
    int32_t S;
};

// PCL's sequential RANSAC loop (ransac.hpp computeModel) replayed over the per-draw counts: adaptive bound
// k = log(1-p)/log(1-w^3), skipped draws do not count as iterations, stop at iterations >= k or > max_iterations.
__global__ void k_rs_select(const int32_t* __restrict__ counts, const float* __restrict__ sp, const int32_t* S_dev,
                            uint32_t seed, int n_draws, int max_it, double probability, Result* res) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const int S = *S_dev;
    res->S = S;
    res->status = 1;
    res->best_draw = -1;
    res->n_inliers = 0;
    if (S < 3) return;
    int iterations = 0, best = -2147483647, best_draw = -1;
    double k = 1.0;
    const double log_probability = log(1.0 - probability);
    const double one_over = 1.0 / (double)S;
    for (int d = 0; d < n_draws && (double)iterations < k; d++) {
        int cnt = counts[d];
        if (cnt < 0) continue;  // skipped
        if (cnt > best) {
            best = cnt;
            best_draw = d;
            double w = (double)best * one_over;
            double p_no = 1.0 - w * w * w;
            p_no = fmax(2.220446049250313e-16, p_no);
            p_no = fmin(1.0 - 2.220446049250313e-16, p_no);
            k = log_probability / log(p_no);
        }
        ++iterations;
        if (iterations > max_it) break;
    }
    res->iterations = iterations;
    res->best_draw = best_draw;
    res->best_count = best;
    if (best_draw < 0) return;
    Model m = draw_model(sp, S, seed, best_draw);
    for (int t = 0; t < 4; t++) {
        res->ransac[t] = m.c[t];
        res->coeffs[t] = m.c[t];
    }
    res->status = 0;
}

// Symmetric 3x3 Jacobi (double), smallest eigenvector; a = xx,xy,xz,yy,yz,zz
__device__ inline void rs_smallest_eigvec(const double s[6], double n0[3]) {
    double a[3][3] = {{s[0], s[1], s[2]}, {s[1], s[3], s[4]}, {s[2], s[4], s[5]}};
    double v[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double diag = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (!(off > 1e-300) || off <= 1e-32 * diag) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                double apq = a[p][q];
                if (apq == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; k++) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; k++) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; k++) {
                    double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = cs * vkp - sn * vkq;
                    v[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    // index of the smallest diagonal entry, ties resolved like a stable ascending sort of (d0,d1,d2)
    int i0 = 0;
    double d0 = a[0][0];
    if (a[1][1] < d0) {
        d0 = a[1][1];
        i0 = 1;
    }
    if (a[2][2] < d0) i0 = 2;
    n0[0] = v[0][i0];
    n0[1] = v[1][i0];
    n0[2] = v[2][i0];
}

// Refinement (:117-126) and the final inlier bitmask (:128-133).  One block of 256 threads.
//   optimizeModelCoefficients: float sums over the RANSAC inliers; thread p owns inliers p, p+256, ... (in list
//   order), partials combined in index order — the same association as the CPU restatement.
__global__ __launch_bounds__(kPartials) void k_rs_refine(const float* __restrict__ sp, const int32_t* __restrict__ sample_idx,
                                                        double thr, double refine_thr, int use_refinement,
                                                        int32_t* inl_pos /* scratch, kSample */, uint32_t* mask,
                                                        Result* res) {
    __shared__ float acc[kPartials][9];
    __shared__ int wsum[kPartials / kWave];
    __shared__ int base_s, total_s;
    if (res->status != 0) return;
    const int S = res->S;
    float rm[4] = {res->ransac[0], res->ransac[1], res->ransac[2], res->ransac[3]};
    const bool valid = fabs((double)rm[2]) >= 0.984807753012208;
    const int tid = threadIdx.x, w = tid / kWave;
    if (tid == 0) {
        base_s = 0;
        total_s = 0;
    }
    __syncthreads();
    // ordered list of the RANSAC inliers (positions in the sample)
    for (int c0 = 0; c0 < S; c0 += kPartials) {
        const int j = c0 + tid;
        const bool in = (j < S) && valid && ((double)plane_dist(rm, sp + 3 * j) < thr);
        const unsigned long long m = __ballot(in);
        if ((tid & (kWave - 1)) == 0) wsum[w] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int q = 0; q < w; q++) off += wsum[q];
        if (in) inl_pos[off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = j;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int q = 0; q < kPartials / kWave; q++) t += wsum[q];
            base_s += t;
        }
        __syncthreads();
    }
    const int ni = base_s;
    if (use_refinement && ni > 3) {
        float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int q = tid; q < ni; q += kPartials) {
            const float* v = sp + 3 * inl_pos[q];
            a[0] += v[0] * v[0];
            a[1] += v[0] * v[1];
            a[2] += v[0] * v[2];
            a[3] += v[1] * v[1];
            a[4] += v[1] * v[2];
            a[5] += v[2] * v[2];
            a[6] += v[0];
            a[7] += v[1];
            a[8] += v[2];
        }
        for (int t = 0; t < 9; t++) acc[tid][t] = a[t];
        __syncthreads();
        if (tid == 0) {
            float s9[9];
            for (int t = 0; t < 9; t++) {
                float sum = 0.0f;
                for (int p = 0; p < kPartials; p++) sum += acc[p][t];
                s9[t] = sum / (float)ni;
            }
            float cov[6] = {s9[0] - s9[6] * s9[6], s9[1] - s9[6] * s9[7], s9[2] - s9[6] * s9[8],
                            s9[3] - s9[7] * s9[7], s9[4] - s9[7] * s9[8], s9[5] - s9[8] * s9[8]};
            double sd[6] = {cov[0], cov[1], cov[2], cov[3], cov[4], cov[5]}, n0[3];
            rs_smallest_eigvec(sd, n0);
            float e0 = (float)n0[0], e1 = (float)n0[1], e2 = (float)n0[2];
            res->coeffs[0] = e0;
            res->coeffs[1] = e1;
            res->coeffs[2] = e2;
            res->coeffs[3] = -1.0f * (e0 * s9[6] + e1 * s9[7] + e2 * s9[8]);
        }
    }
    // final inlier set: RANSAC inliers, or (with refinement) the sample points within refine_thr of the UNREFINED model
    const double sel_thr = use_refinement ? refine_thr : thr;
    int cnt = 0;
    for (int j = tid; j < S; j += kPartials) {
        if (valid && ((double)plane_dist(rm, sp + 3 * j) < sel_thr)) {
            const int32_t id = sample_idx[j];
            const uint32_t bit = 1u << (id & 31);
            cnt += (atomicOr(&mask[id >> 5], bit) & bit) ? 0 : 1;  // a stratified sample may name a point twice
        }
    }
    atomicAdd(&total_s, cnt);
    __syncthreads();
    if (tid == 0) res->n_inliers = total_s;
}

// ------------------------------------------------------------------------------------------------
// Batched, asynchronous form: ONE block per frame slot runs the whole estimator — stratified sample gathered into LDS,
// hypotheses in rounds of 16 (one wavefront each) with PCL's sequential stopping rule replayed after every round,
// refinement, inlier bitmask — and leaves the plane in device memory (PlaneDev) where the feature kernels read it.
// No host round trip.  Arithmetic and association orders are those of the single-slot kernels above (bit-identical
// results).
// The z pass-through (:57-64) - an order-preserving selection over the whole cloud - needs no compacted index list here:
// a first pass leaves one 64-bit candidate mask per 64 points in the slot's (still unused) inlier-mask words and the
// candidate count of every group in LDS; the pos-th candidate of the stratified sample is then found by a binary search
// over per-1024-point prefix sums, a walk over the sixteen group counts of that chunk and a bit select in its mask.
// ------------------------------------------------------------------------------------------------
#ifndef MLD_RS_THREADS
#define MLD_RS_THREADS 1024
#endif
constexpr int kRsThreads = MLD_RS_THREADS;
constexpr int kRsRound = kRsThreads / kWave;  // hypotheses evaluated per round
constexpr int kRsPerThread = (kSample + kRsThreads - 1) / kRsThreads;  // sample points per thread
constexpr int kRsEpoch = kRsThreads;  // draws whose models are set up together (after a slot's first 64)
constexpr int kRsLater = 4;       // draws per wavefront and round after a slot's first round
constexpr int kRsEpochInts = kRsEpoch * 6;  // LDS of an epoch: model (4 floats), count, list entry

__device__ inline long long sample_pos(long long M, int j, uint32_t seed) {
    long long pos = j;
    if (M > kSample) {
        double u = (double)mix(seed, (uint32_t)j, 0x5A17u) * (1.0 / 4294967296.0);
        pos = (long long)(((double)j + u) * (double)M / (double)kSample);
        if (pos > M - 1) pos = M - 1;
    }
    return pos;
}

// k-th (0-based) set bit of m
__device__ inline int select_bit(unsigned long long m, int k) {
    for (int t = 0; t < k; t++) m &= m - 1ull;
    return __ffsll((long long)m) - 1;
}

// Inliers of NM planes among the S sample points (three coordinate arrays in LDS), by one wavefront: a lane reads four
// consecutive points with three 16-byte LDS loads - once for all NM planes (the loop is bound by LDS bandwidth with one
// plane per pass).  Two points per instruction (v_pk_mul_f32 / v_pk_add_f32: the same IEEE operations as plane_dist,
// no contraction), the inliers counted from the comparison masks on the scalar unit.  out[m] is wavefront-uniform.
template <int NM>
__device__ inline void rs_count_inliers(const float* sx, const float* sy, const float* sz, int S, int lane, float thr_f,
                                        const float (&c)[NM][4], int (&out)[NM]) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    int cnt[NM], cnt_wave[NM];  // per lane (the ragged end of the sample) / per wavefront
#pragma unroll
    for (int m = 0; m < NM; m++) cnt[m] = cnt_wave[m] = 0;
    auto four = [&](int j0) {  // four consecutive points per lane
        const f4 x = *reinterpret_cast<const f4*>(sx + j0), y = *reinterpret_cast<const f4*>(sy + j0),
                 z = *reinterpret_cast<const f4*>(sz + j0);
#pragma unroll
        for (int m = 0; m < NM; m++) {
            const f2 C0 = {c[m][0], c[m][0]}, C1 = {c[m][1], c[m][1]}, C2 = {c[m][2], c[m][2]}, C3 = {c[m][3], c[m][3]};
            const f2 da = ((C0 * x.xy + C1 * y.xy) + C2 * z.xy) + C3;
            const f2 db = ((C0 * x.zw + C1 * y.zw) + C2 * z.zw) + C3;
            cnt_wave[m] += __popcll(__ballot(fabsf(da.x) < thr_f)) + __popcll(__ballot(fabsf(da.y) < thr_f)) +
                           __popcll(__ballot(fabsf(db.x) < thr_f)) + __popcll(__ballot(fabsf(db.y) < thr_f));
        }
    };
    const int n_full = (S >> 2) / kWave;  // iterations in which every lane holds four points
    int it = 0;
    if (NM == 1)
        for (; it + 2 <= n_full; it += 2) {  // (unrolled by hand: the ballots keep the compiler from doing it)
            four(4 * lane + it * (4 * kWave));
            four(4 * lane + (it + 1) * (4 * kWave));
        }
    for (; it < n_full; it++) four(4 * lane + it * (4 * kWave));
    const int j0 = 4 * lane + n_full * (4 * kWave);
    if (j0 + 3 < S) four(j0);
    else
        for (int j = j0; j < S; j++)  // the ragged end of the sample
#pragma unroll
            for (int m = 0; m < NM; m++)
                cnt[m] += (fabsf(c[m][0] * sx[j] + c[m][1] * sy[j] + c[m][2] * sz[j] + c[m][3]) < thr_f) ? 1 : 0;
#pragma unroll
    for (int m = 0; m < NM; m++) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt[m] += __shfl_xor(cnt[m], o);
        out[m] = cnt[m] + cnt_wave[m];
    }
}

// One block per frame slot.  pass != 0: pcl::PassThrough on z with the float limits [lo, hi] ahead of the sub-sampling
// (LDS beyond the fixed part: one byte per 64 points + one int per 1024 points).  Where a block's time goes:
// LAB.md 3.17-3.21 (about half of it is the wait for the 6000 scattered cloud reads of the sample).
// single_slot >= 0 (ONE frame per call, mld_calculate_depth_frame_estimate): a grid of one block works on the descriptor
// passed by value, takes `single_seed`, writes that slot's PlaneDev and leaves a second copy of the plane in `copy_out` - the staging block that travels
// back to the host with the depths - beside the slot's resident PlaneDev.
__global__ __launch_bounds__(kRsThreads) void k_rs_batch(const SlotDesc* __restrict__ slots, const uint32_t* __restrict__ seeds,
                                                        int n_draws, int max_it, double probability, double thr,
                                                        double refine_thr, int use_refinement, PlaneDev* out, int pass,
                                                        float lo, float hi,
                                                        float far_elin, float far_econst, float far_thr,
                                                        int single_slot, SlotDesc single, uint32_t single_seed,
                                                        PlaneDev* copy_out) {
    extern __shared__ __align__(16) unsigned char rs_smem[];
    // the sample as three coordinate arrays (structure of arrays): a lane reads four consecutive points with three
    // 16-byte LDS loads in the hypothesis loop
    float* sx = reinterpret_cast<float*>(rs_smem);                 // [kSample]
    float* sy = sx + kSample;                                      // [kSample]
    float* sz = sy + kSample;                                      // [kSample]
    int* inl_pos = reinterpret_cast<int*>(sz + kSample);           // [kSample]
    uint32_t* sidx = reinterpret_cast<uint32_t*>(inl_pos + kSample);  // [kSample] original index of sample point j
    float* acc = reinterpret_cast<float*>(sidx + kSample);         // [kPartials * 9]
    // [0] inliers of the best draw, [1] inlier total, [2] pass-through candidates
    int* misc = reinterpret_cast<int*>(acc + kPartials * 9) + 2 * kRsRound;
    RS_BEGIN();
    // kernel arguments are otherwise fetched where they are first used, in the middle of a slot's serial phases
    asm volatile("" ::"s"(n_draws), "s"(max_it), "s"(probability), "s"(thr), "s"(refine_thr), "s"(use_refinement), "s"(pass));
    asm volatile("" ::"s"(lo), "s"(hi), "s"(far_elin), "s"(far_econst), "s"(far_thr));
    RS_PHASE(15);  // (diagnostic) kernel arguments
    // "(double)distance < thr" for a float distance == "distance < thr_f" with thr_f the smallest float >= thr
    auto up = [](double t) {
        float f = (float)t;
        if ((double)f < t) f = nextafterf(f, __builtin_huge_valf());
        return f;
    };
    const float thr_f = up(thr), sel_thr = up(use_refinement ? refine_thr : thr);
    const double log_probability = log(1.0 - probability);
    asm volatile("" ::"v"(log_probability));  // here, beside the first cloud reads, not where it is first used
    const int tid = threadIdx.x, lane = tid & (kWave - 1), w = tid >> 6;
    const int slot_i = single_slot >= 0 ? single_slot : (int)blockIdx.x;
    const SlotDesc s = single_slot >= 0 ? single : slots[slot_i];
    PlaneDev* pd = out + slot_i;
    const uint32_t seed = single_slot >= 0 ? single_seed : seeds[slot_i];
    asm volatile("" ::"s"(s.cloud), "s"(s.inlier_mask), "s"(s.n), "s"(s.stride), "s"(seed), "s"(pd));
    long long M = s.n;
    int S = M > kSample ? kSample : (int)M;
    auto fail = [&]() {
        if (tid == 0) {
            pd->has_plane = 0;
            pd->status = 1;
            pd->n_inliers = 0;
            pd->S = S;
            if (copy_out) *copy_out = *pd;
        }
    };
    if (M < 3) {  // RansacPlane.cpp:44-50
        fail();
        return;
    }
    // original indices of this thread's sample points (j = tid, tid + 1024, ...); they stay in LDS for the inlier mask
    uint32_t ids[kRsPerThread];
#pragma unroll
    for (int q = 0; q < kRsPerThread; q++) ids[q] = 0u;
    // A cloud read: 16-byte records on a 16-byte boundary take ONE global, non-temporal 16-byte load per point (the
    // sample's points are ~300 B apart, every one its own line that nothing else wants: three scalar loads through the
    // flat path cost 1.17 memory requests per point and 300 us per 1024 frames, this 0.98 and 234).
    const bool vec = (s.stride & 15) == 0 && (((size_t)s.cloud) & 15) == 0;
    auto load_point = [](const unsigned char* rec, bool vec16, float& x, float& y, float& z) {
        if (vec16) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 v = __builtin_nontemporal_load((const f4 __attribute__((address_space(1)))*)rec);
            x = v.x;
            y = v.y;
            z = v.z;
        } else {
            const float* p = reinterpret_cast<const float*>(rec);
            x = p[0];
            y = p[1];
            z = p[2];
        }
    };
    // the sample itself: all of a thread's loads in flight at once
    auto gather_sample = [&]() {
        float px[kRsPerThread], py[kRsPerThread], pz[kRsPerThread];
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++) {
            px[q] = py[q] = pz[q] = 0.0f;
            if (tid + q * kRsThreads < S) load_point(s.cloud + (size_t)ids[q] * (size_t)s.stride, vec, px[q], py[q], pz[q]);
        }
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++) {
            const int j = tid + q * kRsThreads;
            if (j < S) {
                sx[j] = px[q];
                sy[j] = py[q];
                sz[j] = pz[q];
                sidx[j] = ids[q];
            }
        }
    };
    if (pass) {
        const long long n = s.n;
        const int G = (int)((n + kWave - 1) / kWave), NC = (G + 15) / 16;  // 64-point groups, 1024-point chunks
        int* cpre = misc + kRsMisc + kRsEpochInts;                                              // [NC + 1] candidates before chunk c
        unsigned char* gcnt = reinterpret_cast<unsigned char*>(cpre + NC + 1);  // [16 * NC] candidates per group
        // the slot's inlier-mask words (n / 8 bytes, written only at the very end) hold the group masks meanwhile
        unsigned long long* gm = reinterpret_cast<unsigned long long*>(const_cast<uint32_t*>(s.inlier_mask));
        for (int g = w; g < 16 * NC; g += kRsRound) {  // a wavefront per group, no barrier in between
            const long long i = (long long)g * kWave + lane;
            bool f = false;
            if (i < n) {
                float x, y, z;
                load_point(s.cloud + (size_t)i * (size_t)s.stride, vec, x, y, z);
                f = isfinite(x) && isfinite(y) && isfinite(z) && !(z < lo) && !(z > hi);  // k_rs_flags
            }
            const unsigned long long m = __ballot(f);
            if (lane == 0) {
                if (g < G) gm[g] = m;
                gcnt[g] = (unsigned char)__popcll(m);
            }
        }
        __syncthreads();
        // exclusive prefix over the chunks (thread c sums its sixteen groups; one wavefront scans)
        for (int c = tid; c < NC; c += kRsThreads) {
            int t = 0;
            for (int q = 0; q < 16; q++) t += gcnt[16 * c + q];
            cpre[c + 1] = t;
        }
        if (tid == 0) cpre[0] = 0;
        __syncthreads();
        if (w == 0) {
            int carry = 0;
            for (int c0 = 0; c0 < NC; c0 += kWave) {
                const int c = c0 + lane;
                int v = (c < NC) ? cpre[c + 1] : 0, incl = v;
#pragma unroll
                for (int d = 1; d < kWave; d <<= 1) {
                    const int t = __shfl_up(incl, d);
                    if (lane >= d) incl += t;
                }
                if (c < NC) cpre[c + 1] = carry + incl;
                carry += __shfl(incl, kWave - 1);
            }
            if (lane == 0) misc[2] = carry;
        }
        __syncthreads();
        M = misc[2];
        S = M > kSample ? kSample : (int)M;
        if (M < 3) {  // fewer than three candidates: no model (k_rs_select: status 1)
            __syncthreads();
            for (int g = tid; g < G; g += kRsThreads) gm[g] = 0ull;  // hand the mask words back cleared
            fail();
            return;
        }
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++) {
            const int j = tid + q * kRsThreads;
            if (j >= S) continue;
            const int pos = (int)sample_pos(M, j, seed);
            int a = 0, b = NC;  // cpre[a] <= pos < cpre[b]
            while (b - a > 1) {
                const int mid = (a + b) >> 1;
                if (cpre[mid] <= pos) a = mid; else b = mid;
            }
            int r = pos - cpre[a], g = 16 * a;
            while (r >= (int)gcnt[g]) {
                r -= (int)gcnt[g];
                g++;
            }
            ids[q] = (uint32_t)(g * kWave + select_bit(gm[g], r));
        }
        gather_sample();
        __syncthreads();
        for (int g = tid; g < G; g += kRsThreads) gm[g] = 0ull;  // the words become the inlier mask again
    } else {
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++) {
            const int j = tid + q * kRsThreads;
            ids[q] = j < S ? (uint32_t)sample_pos(M, j, seed) : 0u;
        }
        gather_sample();
    }
    // point j of the sample / its distance to a model (plane_dist's arithmetic)
    auto dist = [&](const float c[4], int j) { return fabsf(c[0] * sx[j] + c[1] * sy[j] + c[2] * sz[j] + c[3]); };
    auto model_of = [&](int d) {  // draw_model on the LDS sample
        uint32_t a = mix(seed, (uint32_t)d, 1u) % (uint32_t)S;
        uint32_t b = mix(seed, (uint32_t)d, 2u) % (uint32_t)S;
        uint32_t c = mix(seed, (uint32_t)d, 3u) % (uint32_t)S;
        if (b == a) b = (b + 1) % (uint32_t)S;
        while (c == a || c == b) c = (c + 1) % (uint32_t)S;
        const float p0[3] = {sx[a], sy[a], sz[a]}, p1[3] = {sx[b], sy[b], sz[b]}, p2[3] = {sx[c], sy[c], sz[c]};
        return plane_from(p0, p1, p2);
    };
    if (tid == 0) {
        misc[0] = 0;
        misc[1] = 0;
    }
    __syncthreads();
    RS_PHASE(0);  // sample gathered
    RS_PHASE(14);  // (diagnostic) cost of a marker
    // ---- hypotheses.  The models of an epoch of draws are set up lane-parallel; only the draws whose model is valid
    // (about half) need their inliers counted, so a round hands the next VALID draws to the wavefronts: most slots are
    // decided in one round.  The stopping rule is replayed over the draws in order (skipped and invalid ones included)
    // by every thread on the same data. ----
    int iterations = 0, best = -2147483647, best_draw = -1;
    double k = 1.0;
    const double one_over = 1.0 / (double)S;
    bool done = false;
    RS_PIN(log_probability);
    RS_PIN(one_over);
    RS_PHASE(12);  // (diagnostic) constants of the stopping rule
    float* mc = reinterpret_cast<float*>(misc + kRsMisc);  // [kRsEpoch][4] model of the draw
    int* mcount = reinterpret_cast<int*>(mc + 4 * kRsEpoch);  // [kRsEpoch] its inliers; 0: model not valid; -1: skipped draw
    int* vlist = mcount + kRsEpoch;                           // [kRsEpoch] the epoch's valid draws, in order
    int* wcnt = misc - 2 * kRsRound;  // [kRsRound] valid draws a wavefront found in its share of the epoch
    for (int e0 = 0, cap = 0; e0 < n_draws && !done; e0 += cap) {
        // The first epoch is one wavefront's worth of draws (most slots stop within it).  A slot that goes on has a small
        // inlier share or a cloud whose draws are mostly skipped (NaN points): every wavefront sets up 64 models then.
        cap = e0 == 0 ? kWave : kRsEpoch;
        const int ne = n_draws - e0 < cap ? n_draws - e0 : cap;
        const int dl = w * kWave + lane;  // this thread's draw of the epoch
        bool v = false;
        if (dl < ne) {
            const Model m = model_of(e0 + dl);
            v = m.valid != 0;
#pragma unroll
            for (int t = 0; t < 4; t++) mc[4 * dl + t] = m.c[t];
            mcount[dl] = m.degenerate ? -1 : 0;
            RS_PIN(m.c[3]);
        }
        RS_PHASE(13);  // (diagnostic) model_of
        const unsigned long long vm = __ballot(v);
        if (lane == 0) wcnt[w] = __popcll(vm);
        __syncthreads();
        int nv = 0, off = 0;
#pragma unroll
        for (int q = 0; q < kRsRound; q++) {
            const int c = wcnt[q];
            off += q < w ? c : 0;
            nv += c;
        }
        if (v) vlist[off + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(vm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)vm, 0u))] = dl;
        __syncthreads();
        RS_PHASE(9);  // (diagnostic) models of the epoch
        if (threadIdx.x == 0) RS_COUNT(15, 1);
        int next = 0;  // draws of the epoch replayed so far
        for (int v0 = 0, round_n = 0; !done && next < ne; v0 += round_n) {
            // the first round decides most slots with one draw per wavefront; a slot that goes on needs many more
            // (a small inlier share): kRsLater draws per wavefront from then on, the sample read once for all of them
            const int per = v0 == 0 ? 1 : kRsLater, first = v0 + w * per;
            round_n = kRsRound * per;
            if (first < nv) {
                typedef float f4 __attribute__((ext_vector_type(4)));
                if (per == 1) {
                    const int q = vlist[first];
                    const f4 mq = *reinterpret_cast<const f4*>(mc + 4 * q);
                    const float c1[1][4] = {{mq.x, mq.y, mq.z, mq.w}};
                    int n1[1];
                    rs_count_inliers<1>(sx, sy, sz, S, lane, thr_f, c1, n1);
                    if (lane == 0) mcount[q] = n1[0];
                } else {
                    const int mine = nv - first < kRsLater ? nv - first : kRsLater;
                    int qs[kRsLater], nl[kRsLater];
                    float cl[kRsLater][4];
#pragma unroll
                    for (int t = 0; t < kRsLater; t++) {  // (the last draw again where the list ends)
                        qs[t] = vlist[first + (t < mine ? t : mine - 1)];
                        const f4 mq = *reinterpret_cast<const f4*>(mc + 4 * qs[t]);
                        cl[t][0] = mq.x;
                        cl[t][1] = mq.y;
                        cl[t][2] = mq.z;
                        cl[t][3] = mq.w;
                    }
                    rs_count_inliers<kRsLater>(sx, sy, sz, S, lane, thr_f, cl, nl);
#pragma unroll
                    for (int t = 0; t < kRsLater; t++)
                        if (lane == 0 && t < mine) mcount[qs[t]] = nl[t];
                }
            }
            RS_PHASE(10);  // (diagnostic) distances of the sample
            __syncthreads();
            RS_PHASE(11);  // (diagnostic) wait for the round's other wavefronts
            // every draw ahead of the next valid one that has not been counted yet can be replayed now
            const int upto = v0 + round_n < nv ? vlist[v0 + round_n] : ne;
            // ransac.hpp computeModel, as k_rs_select.  One draw after the other is an LDS round trip per draw (60 ns; a
            // slot whose draws are mostly skipped replays hundreds): the counts of 64 draws are fetched at once and the
            // runs of draws that neither improve on the best model nor reach a stopping rule are booked in one step.
            auto one_draw = [&](int c, int draw) {  // the reference's loop body; false: stop
                if (!((double)iterations < k)) return false;
                if (c < 0) return true;  // skipped draw
                if (c > best) {
                    best = c;
                    best_draw = draw;
                    const double wr = (double)best * one_over;
                    double p_no = 1.0 - wr * wr * wr;
                    p_no = fmax(2.220446049250313e-16, p_no);
                    p_no = fmin(1.0 - 2.220446049250313e-16, p_no);
                    k = log_probability / log(p_no);
                }
                ++iterations;
                return !(iterations > max_it);
            };
            while (next < upto && !done) {
                const int nb = upto - next < kWave ? upto - next : kWave;
                const int c = lane < nb ? mcount[next + lane] : -1;
                const unsigned long long counted = __ballot(c >= 0);
                int pos = 0;
                while (pos < nb && !done) {
                    const unsigned long long rest = ~0ull << pos;
                    const unsigned long long better = __ballot(c > best) & counted & rest;
                    const int ipos = better ? __ffsll((long long)better) - 1 : nb;  // the next draw that improves
                    const unsigned long long run = counted & rest & (ipos < kWave ? ~(~0ull << ipos) : ~0ull);
                    const int m = __popcll(run);  // counted draws ahead of it
                    // first iteration counts at which the rules stop the loop: !(i < k) before a draw, i > max_it after one
                    const int stop_k = k >= 2147483647.0 ? 2147483647 : (int)ceil(k);
                    if (iterations + m < stop_k && iterations + m <= max_it) {
                        iterations += m;  // none of them stops the loop
                        pos = ipos;
                        if (ipos < nb) {
                            if (!one_draw(__shfl(c, ipos), e0 + next + ipos)) done = true;
                            pos = ipos + 1;
                        }
                    } else {  // a stopping rule fires within the run (or right after it): draw by draw
                        const int end = ipos < nb ? ipos + 1 : nb;
                        for (; pos < end && !done; pos++)
                            if (!one_draw(__shfl(c, pos), e0 + next + pos)) done = true;
                    }
                }
                next += nb;
            }
        }
        __syncthreads();  // the next epoch rewrites the models and counts
    }
    RS_PHASE(1);  // hypothesis rounds
    if (threadIdx.x == 0) RS_COUNT(8, iterations);
    if (best_draw < 0) {
        fail();
        return;
    }
    const Model bm = model_of(best_draw);
    const float rm[4] = {bm.c[0], bm.c[1], bm.c[2], bm.c[3]};
    float coeffs[4] = {rm[0], rm[1], rm[2], rm[3]};
    const bool valid = fabs((double)rm[2]) >= 0.984807753012208;
    // ---- ordered list of the RANSAC inliers (positions in the sample), as k_rs_refine: the flags of a thread's (up to
    // six) points at once, one scan over the 6 x 16 (chunk, wavefront) counts ----
    {
        int* wcnt = reinterpret_cast<int*>(acc);       // [kRsPerThread * kRsRound] (the partial sums are not in use yet)
        int* wpre = wcnt + kRsPerThread * kRsRound;    // exclusive prefix of wcnt
        int rank[kRsPerThread];
        unsigned in_bits = 0u;
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++) {
            const int j = tid + q * kRsThreads;
            const bool in = (j < S) && valid && (dist(rm, j) < thr_f);
            const unsigned long long m = __ballot(in);
            rank[q] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            in_bits |= in ? (1u << q) : 0u;
            if (lane == 0) wcnt[q * kRsRound + w] = __popcll(m);
        }
        __syncthreads();
        if (w == 0) {
            static_assert(kRsPerThread * kRsRound <= 2 * kWave, "two counts per lane");
            const int n_cnt = kRsPerThread * kRsRound;
            const int v0 = lane < n_cnt ? wcnt[lane] : 0, v1 = kWave + lane < n_cnt ? wcnt[kWave + lane] : 0;
            int i0 = v0, i1 = v1;
#pragma unroll
            for (int d = 1; d < kWave; d <<= 1) {
                const int t0 = __shfl_up(i0, d), t1 = __shfl_up(i1, d);
                if (lane >= d) {
                    i0 += t0;
                    i1 += t1;
                }
            }
            const int total0 = __shfl(i0, kWave - 1);
            if (lane < n_cnt) wpre[lane] = i0 - v0;
            if (kWave + lane < n_cnt) wpre[kWave + lane] = total0 + i1 - v1;
            if (lane == kWave - 1) misc[0] = total0 + i1;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++)
            if (in_bits & (1u << q)) inl_pos[wpre[q * kRsRound + w] + rank[q]] = tid + q * kRsThreads;
        __syncthreads();
    }
    RS_PHASE(2);  // inlier list
    const int ni = misc[0];
    if (use_refinement && ni > 3) {
        if (tid < kPartials) {  // thread p owns inliers p, p + 256, ...: the association of the single-slot kernel
            float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (int q = tid; q < ni; q += kPartials) {
                const int jq = inl_pos[q];
                const float v[3] = {sx[jq], sy[jq], sz[jq]};
                a[0] += v[0] * v[0];
                a[1] += v[0] * v[1];
                a[2] += v[0] * v[2];
                a[3] += v[1] * v[1];
                a[4] += v[1] * v[2];
                a[5] += v[2] * v[2];
                a[6] += v[0];
                a[7] += v[1];
                a[8] += v[2];
            }
            for (int t = 0; t < 9; t++) acc[tid * 9 + t] = a[t];
        }
        __syncthreads();
        RS_PHASE(3);  // partial sums
        // the 256 partials of each of the nine sums are combined in index order (the restatement's association) - by
        // nine lanes side by side, each with its 256 LDS reads in flight at once, instead of 2304 dependent steps of
        // one thread (which used to be most of this kernel's time)
        float mine = 0.0f;
        if (tid < 9) {
            float sum = 0.0f;
#pragma unroll 16
            for (int p = 0; p < kPartials; p++) sum += acc[p * 9 + tid];
            mine = sum / (float)ni;
        }
        __syncthreads();
        if (tid < 9) acc[tid] = mine;
        __syncthreads();
        RS_PHASE(4);  // combination
        if (tid == 0) {
            float s9[9];
            for (int t = 0; t < 9; t++) s9[t] = acc[t];
            float cov[6] = {s9[0] - s9[6] * s9[6], s9[1] - s9[6] * s9[7], s9[2] - s9[6] * s9[8],
                            s9[3] - s9[7] * s9[7], s9[4] - s9[7] * s9[8], s9[5] - s9[8] * s9[8]};
            double sd[6] = {cov[0], cov[1], cov[2], cov[3], cov[4], cov[5]}, n0[3];
            rs_smallest_eigvec(sd, n0);
            const float e0 = (float)n0[0], e1 = (float)n0[1], e2 = (float)n0[2];
            coeffs[0] = e0;
            coeffs[1] = e1;
            coeffs[2] = e2;
            coeffs[3] = -1.0f * (e0 * s9[6] + e1 * s9[7] + e2 * s9[8]);
        }
    }
    RS_PHASE(5);  // eigenvector
    // ---- final inlier set (bitmask keyed by original index; the host cleared it before the launch) ----
    int cnt = 0;
    // Clouds up to 32 x kSample points: the slot's whole mask is put together where the inlier list was (LDS atomics)
    // and leaves as plain coalesced stores - 4 M global atomics with return per 1024 frames were a fifth of this kernel.
    const int mask_words = (int)((s.n + 31) >> 5);
    const bool lds_mask = mask_words <= kSample;
    uint32_t* lm = reinterpret_cast<uint32_t*>(inl_pos);
    uint32_t* gmask = const_cast<uint32_t*>(s.inlier_mask);
    if (lds_mask) {
        for (int i = tid; i < mask_words; i += kRsThreads) lm[i] = 0u;
        __syncthreads();
    }
    {
        // (global path: the thread's atomics travel together.)  The return values tell duplicates apart: they count once
        uint32_t prev[kRsPerThread], bits[kRsPerThread];
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++) {
            const int j = tid + q * kRsThreads;
            const bool in = j < S && valid && (dist(rm, j) < sel_thr);
            const uint32_t id = in ? sidx[j] : 0u;
            bits[q] = 1u << (id & 31);
            prev[q] = !in ? bits[q] : lds_mask ? atomicOr(lm + (id >> 5), bits[q]) : atomicOr(gmask + (id >> 5), bits[q]);
        }
#pragma unroll
        for (int q = 0; q < kRsPerThread; q++) cnt += (prev[q] & bits[q]) ? 0 : 1;
    }
    if (lds_mask) {
        __syncthreads();
        for (int i = tid; i < mask_words; i += kRsThreads) gmask[i] = lm[i];
    }
    RS_PHASE(6);  // mask
    atomicAdd(&misc[1], cnt);
    __syncthreads();
    if (tid == 0) {
        // the plane as the feature kernels consume it: coefficients + M-estimator prior (DepthEstimator.cpp:286-292)
        for (int t = 0; t < 4; t++) pd->coeffs[t] = coeffs[t];
        double a = (double)coeffs[0], b = (double)coeffs[1], cc = (double)coeffs[2];
        const double z = a * a + (b * b + cc * cc);
        if (z > 0.0) {
            const double nrm = sqrt(z);
            a /= nrm;
            b /= nrm;
            cc /= nrm;
        }
        pd->prior_n[0] = a;
        pd->prior_n[1] = b;
        pd->prior_n[2] = cc;
        pd->prior_off = (double)coeffs[3];
        far_margins(coeffs, far_elin, far_econst, far_thr, pd->far_mg0, pd->far_mg1);
        pd->n_inliers = misc[1];
        pd->iterations = iterations;
        pd->best_draw = best_draw;
        pd->best_count = best;
        pd->S = S;
        pd->status = 0;
        pd->has_plane = 1;
        if (copy_out) *copy_out = *pd;
    }
    RS_PHASE(7);  // plane written
    RS_FLUSH();
}

// ------------------------------------------------------------------------------------------------
// SemanticPlane::CalculateInliersPlane (monolidar_fusion/src/RansacPlane.cpp:195-274): ground candidates from a
// label image, least-squares plane, re-selection over the whole cloud, second fit.  Flag kernels feed the same
// order-preserving compaction as the pass-through above; the fits use the 256-partial float association of
// k_rs_refine.
struct LabelSet {
    uint32_t w[8];  // bit l set: label l (0..255) is ground
};
struct SemCalib {
    double T[12];  // lidar -> camera, row-major 3x4
    double f, cu, cv;
};
struct SemResult {
    float coeffs[4];
    int32_t n_candidates;
    int32_t n_inliers;
    int32_t status;  // 0 ok, 1 fewer than 3 candidates
    int32_t pad_;
};

// Four launches, no index lists and no host round trip: (1) candidates by label, (2) first fit, (3) selection over the
// whole cloud - its ballots ARE the slot's inlier bitmask -, (4) second fit + the plane as the other kernels consume it.
// The least-squares fits (optimizeModelCoefficients = computeMeanAndCovarianceMatrix, float accumulators) never see an
// index list: the kernels that decide membership also reduce the nine moment terms of their members per GROUP of 64
// consecutive cloud points (non-members contribute +0.0f, which leaves a float sum unchanged; a fixed binary tree over the
// 64 positions - strides 32, 16, ..., 1), and the fit kernel adds the group sums, in group order, into 256 interleaved
// partials (group g -> partial g % 256) that are combined in index order.  PCL adds the members one after the other;
// neither order is the other's (parity unpinned, as stated in mld.h) - this one is the same in the CPU restatement the tests
// compare with and on the GPU, bit for bit, and needs no ordered compaction.
struct GroupSums {
    float s[9];  // xx, xy, xz, yy, yz, zz, x, y, z of the group's members
    int count;   // members
};

// the nine terms of a member (zeros for a non-member) reduced over the wavefront's 64 points; lane 0 stores them
__device__ inline void sem_group_reduce(bool member, float x, float y, float z, GroupSums* __restrict__ out, long long g,
                                        bool store) {
    float a[9];
    a[0] = member ? x * x : 0.0f;
    a[1] = member ? x * y : 0.0f;
    a[2] = member ? x * z : 0.0f;
    a[3] = member ? y * y : 0.0f;
    a[4] = member ? y * z : 0.0f;
    a[5] = member ? z * z : 0.0f;
    a[6] = member ? x : 0.0f;
    a[7] = member ? y : 0.0f;
    a[8] = member ? z : 0.0f;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int t = 0; t < 9; t++) a[t] = a[t] + __shfl_xor(a[t], o);  // (commutative: every lane holds the tree's value)
    const int cnt = (int)__popcll(__ballot(member));
    if (store && (threadIdx.x & (kWave - 1)) == 0) {
        GroupSums r;
#pragma unroll
        for (int t = 0; t < 9; t++) r.s[t] = a[t];
        r.count = cnt;
        out[g] = r;
    }
}

// :198-221  pcl::transformPointCloud (double arithmetic, float result), project() (K * p as Eigen evaluates a
// 3x3 * 3x1 product: x0 + (x1 + x2); p /= p[2]; cv::Point truncation), image bounds, label lookup.  The pixels
// x == cols / y == rows the reference reads out of bounds count as unlabeled; non-finite projections as invalid.
__global__ __launch_bounds__(256) void k_sem_candidates(const unsigned char* cloud, long long n, int stride, SemCalib sc,
                                                        const unsigned char* img, int rows, int cols, int row_stride,
                                                        LabelSet ls, GroupSums* __restrict__ gs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool flag = false;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (i < n) {
        const float* q = reinterpret_cast<const float*>(cloud + (size_t)i * stride);
        fx = q[0];
        fy = q[1];
        fz = q[2];
        const double x = fx, y = fy, z = fz;
        const float xc = (float)(((sc.T[0] * x + sc.T[1] * y) + sc.T[2] * z) + sc.T[3]);
        const float yc = (float)(((sc.T[4] * x + sc.T[5] * y) + sc.T[6] * z) + sc.T[7]);
        const float zc = (float)(((sc.T[8] * x + sc.T[9] * y) + sc.T[10] * z) + sc.T[11]);
        const double px = (double)xc, py = (double)yc, pz = (double)zc;
        const double p0 = sc.f * px + (0.0 * py + sc.cu * pz);
        const double p1 = 0.0 * px + (sc.f * py + sc.cv * pz);
        const double p2 = 0.0 * px + (0.0 * py + 1.0 * pz);
        const double u = p0 / p2, v = p1 / p2;
        if (isfinite(u) && isfinite(v) && fabs(u) < 2147483648.0 && fabs(v) < 2147483648.0) {
            const int ix = (int)u, iy = (int)v;
            if (ix >= 0 && ix < cols && iy >= 0 && iy < rows) {
                const unsigned l = img[(size_t)iy * row_stride + ix];
                flag = (ls.w[l >> 5] >> (l & 31)) & 1u;
            }
        }
    }
    const long long i0 = i - (long long)(threadIdx.x & (kWave - 1));  // the wavefront's first point
    sem_group_reduce(flag, fx, fy, fz, gs, i0 >> 6, i0 < n);
}

// SampleConsensusModelPlane::selectWithinDistance over the whole cloud (:252), float distance < double threshold.  The
// ballots ARE the slot's inlier bitmask (bit i of 32-bit word i / 32).
__global__ __launch_bounds__(256) void k_sem_select(const unsigned char* cloud, long long n, int stride,
                                                    const float* __restrict__ coeffs, double thr,
                                                    unsigned long long* __restrict__ gm, GroupSums* __restrict__ gs) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    bool flag = false;
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (i < n) {
        const float c[4] = {coeffs[0], coeffs[1], coeffs[2], coeffs[3]};
        const float* q = reinterpret_cast<const float*>(cloud + (size_t)i * stride);
        fx = q[0];
        fy = q[1];
        fz = q[2];
        flag = (double)plane_dist(c, q) < thr;
    }
    const unsigned long long m = __ballot(flag);
    const long long i0 = i - (long long)(threadIdx.x & (kWave - 1));
    if ((threadIdx.x & (kWave - 1)) == 0 && i0 < n) gm[i0 >> 6] = m;
    sem_group_reduce(flag, fx, fy, fz, gs, i0 >> 6, i0 < n);
}

// optimizeModelCoefficients from the group sums (see above): fewer than 4 members return `fallback`.  ONE block of 256
// threads.  stage 0: the candidates' fit (records their number; fewer than min_count = ExceptionPclInvalid, :224-227).
// stage 1: the refit on the selected points (records the inlier count) and, with pd, the plane as the projection and the
// feature kernels consume it (what set_plane_coeffs prepares on the host for a plane that was read back): coefficients,
// M-estimator prior (DepthEstimator.cpp:286-292), margins of the projection's far test; copy_out as in k_rs_batch.
__global__ __launch_bounds__(kPartials) void k_sem_fit(const GroupSums* __restrict__ gs, int G,
                                                       const float* __restrict__ fallback, float* __restrict__ out,
                                                       int min_count, int stage, SemResult* res, PlaneDev* pd,
                                                       PlaneDev* copy_out, float far_elin, float far_econst, float far_thr) {
    __shared__ float acc[kPartials][9];
    __shared__ float s9s[9];
    __shared__ int wcnt[kPartials / kWave];
    const int tid = threadIdx.x;
    float a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cnt = 0;
    for (int g = tid; g < G; g += kPartials) {
        const GroupSums r = gs[g];
#pragma unroll
        for (int t = 0; t < 9; t++) a[t] += r.s[t];
        cnt += r.count;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if ((tid & (kWave - 1)) == 0) wcnt[tid / kWave] = cnt;
    for (int t = 0; t < 9; t++) acc[tid][t] = a[t];
    __syncthreads();
    int m = 0;
    for (int q = 0; q < kPartials / kWave; q++) m += wcnt[q];
    if (tid == 0) {
        if (stage == 0) {
            res->n_candidates = m;
            res->status = (m < min_count) ? 1 : 0;
        } else {
            res->n_inliers = m;
        }
    }
    // the 256 partials of each sum combined in index order, by nine lanes side by side
    if (tid < 9) {
        float sum = 0.0f;
#pragma unroll 16
        for (int p = 0; p < kPartials; p++) sum += acc[p][tid];
        s9s[tid] = sum / (float)m;
    }
    __syncthreads();
    if (tid != 0) return;
    float c4[4] = {fallback[0], fallback[1], fallback[2], fallback[3]};
    if (m >= 4) {
        float s9[9];
        for (int t = 0; t < 9; t++) s9[t] = s9s[t];
        const float cov[6] = {s9[0] - s9[6] * s9[6], s9[1] - s9[6] * s9[7], s9[2] - s9[6] * s9[8],
                              s9[3] - s9[7] * s9[7], s9[4] - s9[7] * s9[8], s9[5] - s9[8] * s9[8]};
        double sd[6] = {cov[0], cov[1], cov[2], cov[3], cov[4], cov[5]}, n0[3];
        rs_smallest_eigvec(sd, n0);
        const float e0 = (float)n0[0], e1 = (float)n0[1], e2 = (float)n0[2];
        c4[0] = e0;
        c4[1] = e1;
        c4[2] = e2;
        c4[3] = -1.0f * (e0 * s9[6] + e1 * s9[7] + e2 * s9[8]);
    }
    for (int t = 0; t < 4; t++) {
        out[t] = c4[t];
        if (stage == 1) res->coeffs[t] = c4[t];
    }
    if (stage == 1 && pd) {
        const int ok = res->status == 0;
        for (int t = 0; t < 4; t++) pd->coeffs[t] = c4[t];
        double na = (double)c4[0], nb = (double)c4[1], nc = (double)c4[2];
        const double z = na * na + (nb * nb + nc * nc);
        if (z > 0.0) {
            const double nrm = sqrt(z);
            na /= nrm;
            nb /= nrm;
            nc /= nrm;
        }
        pd->prior_n[0] = na;
        pd->prior_n[1] = nb;
        pd->prior_n[2] = nc;
        pd->prior_off = (double)c4[3];
        far_margins(c4, far_elin, far_econst, far_thr, pd->far_mg0, pd->far_mg1);
        pd->n_inliers = ok ? m : 0;
        pd->iterations = 0;
        pd->best_draw = -1;
        pd->best_count = res->n_candidates;
        pd->S = res->n_candidates;
        pd->status = ok ? 0 : 1;
        pd->has_plane = ok;
        if (copy_out) *copy_out = *pd;
    }
}

}  // namespace ransac
}  // namespace mld

