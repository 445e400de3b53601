// Microbenchmark: what a launch of MANY LARGE BLOCKS costs beyond the blocks' own work.  1024 blocks of T threads with L
// bytes of dynamic LDS, every block idling for `us` microseconds (s_sleep on a 100 MHz clock); with one block per CU
// at a time (L > half the LDS) the ideal is 4 x us on 256 CUs.  Prints launch time, the span from the first block's
// start to the last block's end, how long it takes until 256 blocks are running, and the gap between a block's end
// and the next block's start on the same CU.  (k_rs_batch: 1024 threads, 150 KB, 47 us -> 295 instead of 188 us.)
//   usage: dispatch_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>
struct Rec {
    unsigned long long t0, t1;
    unsigned hw;
};
// gather != 0: every thread first reads `gather` random 16-byte pieces of buf (the sample gather of k_rs_batch: all
// blocks of a round flood the memory system at the same time)
__global__ void k_idle(int us, Rec* rec, const uint4* __restrict__ buf, unsigned n16, int gather) {
    extern __shared__ unsigned char smem[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) smem[0] = 1;  // the allocation is used
    if (gather) {
        unsigned acc = 0, x = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u + 12345u;
        for (int q = 0; q < gather; q++) {
            x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
            acc += buf[x % n16].x;
        }
        if (acc == 0x12345678u) smem[1] = 2;
    }
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    if (threadIdx.x == 0) {
        rec[blockIdx.x].t0 = t0;
        rec[blockIdx.x].t1 = wall_clock64();
        rec[blockIdx.x].hw = (__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) & 0xFFFFF) |
                             ((__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xF) << 20);
    }
}
int main() {
    const int blocks = 1024, us = 45;
    Rec* d;
    (void)hipMalloc((void**)&d, blocks * sizeof(Rec));
    const unsigned n16 = 1u << 27;  // 2 GiB
    uint4* buf;
    (void)hipMalloc((void**)&buf, (size_t)n16 * sizeof(uint4));
    (void)hipMemset(buf, 1, (size_t)n16 * sizeof(uint4));
    std::vector<Rec> h(blocks);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    std::printf("%8s %8s | %9s %9s %12s %22s %8s\n", "threads", "lds KB", "launch us", "span us", "256 running", "gap us mean/p50/p90", "blk/CU");
    for (int gather : {0, 6})
    for (int lds_kb : {0, 82, 150})
        for (int T : {256, 512, 1024}) {
            if (gather && (lds_kb != 150 || T != 1024)) continue;
            const size_t lds = (size_t)lds_kb * 1024;
            if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_idle), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(k_idle, dim3(blocks), dim3(T), lds, 0, us, d, buf, n16, gather);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h.data(), d, blocks * sizeof(Rec), hipMemcpyDeviceToHost);
            unsigned long long tmin = ~0ull, tmax = 0;
            std::vector<unsigned long long> starts;
            std::map<unsigned, std::vector<Rec>> per_cu;
            for (const Rec& r : h) {
                tmin = std::min(tmin, r.t0);
                tmax = std::max(tmax, r.t1);
                starts.push_back(r.t0);
                const unsigned cu = ((r.hw >> 8) & 0xF) | (((r.hw >> 12) & 0x3) << 4) | (((r.hw >> 13) & 0x7) << 6) | ((r.hw >> 20) << 10);
                per_cu[cu].push_back(r);
            }
            std::sort(starts.begin(), starts.end());
            std::vector<double> gaps;
            size_t most = 0;
            for (auto& kv : per_cu) {
                auto& v = kv.second;
                most = std::max(most, v.size());
                std::sort(v.begin(), v.end(), [](const Rec& a, const Rec& b) { return a.t0 < b.t0; });
                // with several blocks per CU at a time the "next block" is the next start after this block's end
                for (size_t i = 0; i < v.size(); i++)
                    for (size_t j = i + 1; j < v.size(); j++)
                        if (v[j].t0 >= v[i].t1) {
                            gaps.push_back((double)(v[j].t0 - v[i].t1) / 100.0);
                            break;
                        }
            }
            std::sort(gaps.begin(), gaps.end());
            double mean = 0;
            for (double g : gaps) mean += g;
            if (!gaps.empty()) mean /= (double)gaps.size();
            if (gather) std::printf("with %d random 16-byte reads per thread at the start of every block:\n", gather);
            std::printf("%8d %8d | %9.1f %9.1f %12.1f %8.1f /%6.1f /%6.1f %8zu\n", T, lds_kb, ms * 1e3, (double)(tmax - tmin) / 100.0,
                        (double)(starts[std::min<size_t>(255, starts.size() - 1)] - tmin) / 100.0, mean,
                        gaps.empty() ? 0.0 : gaps[gaps.size() / 2], gaps.empty() ? 0.0 : gaps[gaps.size() * 9 / 10], most);
        }
    return 0;
}
