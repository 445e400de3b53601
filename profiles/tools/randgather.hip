// Microbenchmark: the rate of RANDOM line gathers MI355X sustains, by where the lines live (L1 / L2 / Infinity Cache /
// HBM) — the ceilings for the gather phases of the feature kernels.  Each lane loads 4 or 16 bytes from a pseudo-random
// 64-byte line of a buffer of the given size; PER independent loads in flight per lane.
//   usage: randgather   (prints a table: buffer size x access width)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
template <int PER, int W>
__global__ __launch_bounds__(256) void k_gather(const unsigned char* __restrict__ buf, unsigned lines, unsigned seed,
                                                 int rounds, unsigned* out) {
    unsigned gid = blockIdx.x * 256u + threadIdx.x;
    unsigned acc = 0;
    for (int r = 0; r < rounds; r++) {
        unsigned v[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            unsigned line = hash32(gid * 977u + (unsigned)(r * PER + q) * 0x9E3779B9u + seed) % lines;
            const unsigned char* p = buf + (size_t)line * 64ull;
            if (W == 16) { uint4 t = *reinterpret_cast<const uint4*>(p); v[q] = t.x ^ t.w; }
            else v[q] = *reinterpret_cast<const unsigned*>(p);
        }
#pragma unroll
        for (int q = 0; q < PER; q++) acc += v[q];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int PER, int W>
double run(const unsigned char* buf, unsigned lines, int blocks, int rounds, unsigned* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_gather<PER, W><<<blocks, 256>>>(buf, lines, 1u, rounds, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) k_gather<PER, W><<<blocks, 256>>>(buf, lines, 7u + i, rounds, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double loads = 5.0 * blocks * 256.0 * rounds * PER;
    return loads / (ms * 1e-3);
}
int main() {
    unsigned char* buf; unsigned* out;
    (void)hipMalloc(&buf, (size_t)4096 << 20); (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 1, (size_t)4096 << 20);
    const double ghz = 2.4, cus = 256;
    for (size_t kb : {16ul, 256ul, 2048ul, 65536ul, 4194304ul}) {
        unsigned lines = (unsigned)((kb << 10) / 64);
        double a = run<4, 4>(buf, lines, 8192, 16, out), b = run<8, 4>(buf, lines, 8192, 8, out),
               c = run<4, 16>(buf, lines, 8192, 16, out);
        printf("buffer %8zu KiB: 4B x4 %.1f G/s (%.3f lanes/clk/CU) | 4B x8 %.1f G/s | 16B x4 %.1f G/s (%.3f lanes/clk/CU)\n", kb,
               a / 1e9, a / 1e9 / ghz / cus, b / 1e9, c / 1e9, c / 1e9 / ghz / cus);
    }
    return 0;
}
