// Instrumentation of the DIAGNOSTIC builds of libmld_hip (profiles/tools/*.sh build them with -DMLD_STAMPS,
// -DMLD_DIAG_NO_KEYS / -DMLD_DIAG_NO_POINTS, -DMLD_DIAG_RS_PHASES).  The product build defines none of these: every
// macro below is then empty and the kernels contain no trace of it.
//   ST_*      s_memtime stamps around the phases of the feature kernels (profiles/tools/stamps.py)
//   MLD_DIAG_FAKE_POINT / MLD_DIAG_KEY   the feature kernel without its point / key gathers (same instruction stream,
//             wrong results: what the random memory traffic costs the kernel beside it)
//   RS_*      phase clocks of k_rs_batch (profiles/tools/rs_phases.py)
// Included twice: by mld_kernels.hip (device part) and, with MLD_DIAG_HOST_PART defined, at the end of mld_api.hip (the
// extern "C" read-back entry points of the diagnostic builds).
#ifndef MLD_DIAG_HOST_PART
#ifndef MLD_DIAG_H_
#define MLD_DIAG_H_
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mld {
#ifdef MLD_STAMPS
constexpr int kStampWaves = 32768;  // per-wave slots (no atomics: contended adds would distort what they measure)
__device__ unsigned g_stamps[2][kStampWaves][16];
struct Stamps {
    unsigned long long last;
    unsigned acc[16];
    int kern;
    __device__ __forceinline__ void begin(int k) {
        kern = k;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0;
        last = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int i) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        acc[i] += (unsigned)(t - last);
        last = t;
        __builtin_amdgcn_sched_barrier(0);
    }
    __device__ __forceinline__ void flush() {
        const unsigned w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        if ((threadIdx.x & 63) == 0 && kern < 2 && w < (unsigned)kStampWaves) {
#pragma unroll
            for (int i = 0; i < 15; i++) g_stamps[kern][w][i] += acc[i];
            g_stamps[kern][w][15] += 1u;
        }
    }
};
#define ST_ARG , Stamps& st
#define ST_PASS , st
#define ST_MARK(i) st.mark(i)
#define ST_USE_F64(x) asm volatile("" ::"v"(x))
#define ST_USE_U32(x) asm volatile("" ::"v"(x))
#define ST_BEGIN(k) \
    Stamps st;      \
    st.begin(k)
#define ST_END() st.flush()
#else
#define ST_ARG
#define ST_PASS
#define ST_MARK(i)
#define ST_USE_F64(x)
#define ST_USE_U32(x)
#define ST_BEGIN(k)
#define ST_END()
#endif

#ifdef MLD_DIAG_NO_POINTS
#define MLD_DIAG_FAKE_POINT(i) \
    return RawP{5.0f + (float)((i) & 1023u) * 0.01f, (float)((i) & 63u) * 0.05f - 1.6f, -1.7f + (float)((i) & 7u) * 0.01f}
#else
#define MLD_DIAG_FAKE_POINT(i)
#endif
#ifdef MLD_DIAG_NO_KEYS
#define MLD_DIAG_KEY(load, tag, cell) make_key((tag), ((cell) * 2654435761u) >> 15, 1u)
#else
#define MLD_DIAG_KEY(load, tag, cell) (load)
#endif

// -DMLD_DIAG_NO_TAIL: the lane-per-feature kernel without corner selection + tail (wrong results): what those stages
// cost in TIME, i.e. the most a deferred, dense execution of them could return
#ifdef MLD_DIAG_NO_TAIL
#define MLD_DIAG_SKIP_TAIL() \
    do {                     \
        if (live) {          \
            mytype = MLD_TriangleNotPlanar; \
            mydepth = -1.0;  \
        }                    \
        return;              \
    } while (0)
#else
#define MLD_DIAG_SKIP_TAIL()
#endif

namespace ransac {
#ifdef MLD_DIAG_RS_PHASES
// diagnostic build only: time per phase of k_rs_batch (100 MHz ticks of thread 0, collected in LDS, summed over the
// blocks at the end)
__device__ unsigned long long g_rs_phase[4096 * 16];  // per block (no atomics: they would sit between a block's end and the next block's start)
__device__ unsigned long long g_rs_stamp[4];  // wall clock just before / after the launch (k_rs_stamp), in stream order
__global__ void k_rs_stamp(int i) { g_rs_stamp[i] = wall_clock64(); }
__device__ unsigned int g_rs_slot[4096 * 4];  // per slot: ticks in all, ticks of the rounds, iterations, epochs
constexpr int kRsMisc = 8 + 32;
#define RS_PHASE(i)                                                                     \
    do {                                                                                \
        if (threadIdx.x == 0) {                                                         \
            const unsigned long long t_now = wall_clock64();                            \
            reinterpret_cast<unsigned long long*>(misc + 8)[i] += t_now - t_prev;       \
            t_prev = t_now;                                                             \
        }                                                                               \
    } while (0)
#define RS_COUNT(i, v) reinterpret_cast<unsigned long long*>(misc + 8)[i] += (unsigned long long)(v)
#define RS_PIN(x) asm volatile("" ::"v"(x))  /* the value is computed before the next marker */
#define RS_FLUSH()                                                                                              \
    do {                                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                            \
            unsigned long long* a_ = reinterpret_cast<unsigned long long*>(misc + 8);                           \
            unsigned long long all_ = 0;                                                                        \
            for (int i_ = 0; i_ < 16; i_++) all_ += i_ == 8 ? 0 : a_[i_];                                       \
            g_rs_slot[4 * blockIdx.x + 0] = (unsigned int)all_;                                                 \
            g_rs_slot[4 * blockIdx.x + 1] = (unsigned int)wall_clock64(); /* end, absolute */                    \
            g_rs_slot[4 * blockIdx.x + 2] = (unsigned int)a_[8];                                                \
            g_rs_slot[4 * blockIdx.x + 3] = (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xFFFFF) | /* HW_ID */ \
                                            ((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) << 20); /* XCC_ID */ \
        }                                                                                                       \
        if (threadIdx.x == 0)                                                                                   \
            for (int i_ = 0; i_ < 16; i_++) g_rs_phase[16 * (blockIdx.x & 4095) + i_] = reinterpret_cast<unsigned long long*>(misc + 8)[i_]; \
    } while (0)
#define RS_BEGIN()                                                                                       \
    unsigned long long t_prev = wall_clock64();                                                        \
    if (threadIdx.x == 0)                                                                              \
        for (int i_ = 0; i_ < 16; i_++) reinterpret_cast<unsigned long long*>(misc + 8)[i_] = 0ull
#define RS_STAMP_LAUNCH(st, i) hipLaunchKernelGGL(mld::ransac::k_rs_stamp, dim3(1), dim3(1), 0, (st), (i))
#else
constexpr int kRsMisc = 8;
#define RS_PHASE(i) do {} while (0)
#define RS_COUNT(i, v) do {} while (0)
#define RS_PIN(x) do {} while (0)
#define RS_FLUSH() do {} while (0)
#define RS_BEGIN() do {} while (0)
#define RS_STAMP_LAUNCH(st, i) do {} while (0)
#endif
}  // namespace ransac
}  // namespace mld
#endif  // MLD_DIAG_H_

#else  // MLD_DIAG_HOST_PART: read-back entry points (diagnostic builds only)
#ifdef MLD_STAMPS
// Diagnostic build only: read and clear the per-wave phase stamps of the feature kernels (2 x 32768 x 16 uint32).
extern "C" int mld_debug_read_stamps(unsigned* out) {
    const size_t bytes = sizeof(unsigned) * 2 * mld::kStampWaves * 16;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(out, HIP_SYMBOL(mld::g_stamps), bytes);
    void* sym = nullptr;
    if (e == hipSuccess) e = hipGetSymbolAddress(&sym, HIP_SYMBOL(mld::g_stamps));
    if (e == hipSuccess) e = hipMemset(sym, 0, bytes);
    return e == hipSuccess ? 0 : -7;
}
#endif
#ifdef MLD_DIAG_RS_PHASES
// diagnostic build only (profiles/tools/rs_phases.sh): reads and clears the phase clocks of k_rs_batch
extern "C" int mld_debug_rs_phases(unsigned long long* out16) {  // sums over the blocks; clears
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    static unsigned long long h[4096 * 16];
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(mld::ransac::g_rs_phase), sizeof(h)) != hipSuccess) return -1;
    for (int i = 0; i < 16; i++) out16[i] = 0;
    for (int b = 0; b < 4096; b++)
        for (int i = 0; i < 16; i++) out16[i] += h[16 * b + i];
    for (auto& v : h) v = 0;
    return hipMemcpyToSymbol(HIP_SYMBOL(mld::ransac::g_rs_phase), h, sizeof(h)) == hipSuccess ? 0 : -1;
}
extern "C" int mld_debug_rs_stamps(unsigned long long* out4) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out4, HIP_SYMBOL(mld::ransac::g_rs_stamp), 4 * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
extern "C" int mld_debug_rs_slots(unsigned int* out) {  // 4096 x (ticks, ticks of the rounds, iterations, epochs)
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(mld::ransac::g_rs_slot), 4096 * 4 * sizeof(unsigned int)) == hipSuccess ? 0 : -1;
}
#endif
#endif
