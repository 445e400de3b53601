// Microbenchmark: the rate at which MI355X STREAMS a buffer that is read once (the projection's access pattern: 2.1 GB of
// 16-byte point records per 1024-frame launch, nothing re-used) - the achievable ceiling beside the 8 TB/s of the data
// sheet, for kernels shaped like k_project_scatter: blocks of 256 threads, PER 16-byte loads per thread in flight,
// one block per PER * 4 KB (the "grid" shape) or a resident grid striding over the buffer (the "persistent" shape),
// non-temporal or plain loads.  Also with a 4-byte store per KEEP-th record scattered into a second buffer (the map keys'
// share of the projection's traffic is written line by line).
//   usage: streamread [GiB]   (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PER, bool NT>
__global__ __launch_bounds__(256) void k_grid(const f4* __restrict__ buf, size_t n, unsigned* out) {
    const size_t base = (size_t)blockIdx.x * (256u * PER) + threadIdx.x;
    f4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const size_t i = base + (size_t)q * 256u;
        const f4* p = buf + (i < n ? i : 0);
        v[q] = NT ? __builtin_nontemporal_load((const f4 __attribute__((address_space(1)))*)p) : *p;
    }
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < PER; q++) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    if (acc == 1.2345678f) out[0] = 1u;
}
template <int PER, bool NT>
__global__ __launch_bounds__(256) void k_persistent(const f4* __restrict__ buf, size_t n, unsigned* out) {
    float acc = 0.f;
    const size_t step = (size_t)gridDim.x * (256u * PER);
    for (size_t base = (size_t)blockIdx.x * (256u * PER) + threadIdx.x; base < n; base += step) {
        f4 v[PER];
#pragma unroll
        for (int q = 0; q < PER; q++) {
            const size_t i = base + (size_t)q * 256u;
            const f4* p = buf + (i < n ? i : 0);
            v[q] = NT ? __builtin_nontemporal_load((const f4 __attribute__((address_space(1)))*)p) : *p;
        }
#pragma unroll
        for (int q = 0; q < PER; q++) acc += v[q].x + v[q].y + v[q].z + v[q].w;
    }
    if (acc == 1.2345678f) out[0] = 1u;
}
// the projection's write side: one record in four leaves a 4-byte key at a pseudo-random cell of a 1.9 MB map per
// 131072 records (a frame), by atomicMax
template <int PER>
__global__ __launch_bounds__(256) void k_grid_scatter(const f4* __restrict__ buf, size_t n, unsigned* map, size_t map_words,
                                                      unsigned* out) {
    const size_t base = (size_t)blockIdx.x * (256u * PER) + threadIdx.x;
    f4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const size_t i = base + (size_t)q * 256u;
        v[q] = __builtin_nontemporal_load((const f4 __attribute__((address_space(1)))*)(buf + (i < n ? i : 0)));
    }
#pragma unroll
    for (int q = 0; q < PER; q++) {
        const size_t i = base + (size_t)q * 256u;
        if ((i & 3) == 0) {
            const size_t frame = i >> 17;
            unsigned h = (unsigned)i * 0x9E3779B9u;
            h ^= h >> 15;
            // (neighbouring records land in neighbouring cells of one row, as a lidar ring does)
            const size_t cell = frame * 465750ull + ((h % 375u) * 1242ull + ((i >> 2) % 1242ull));
            atomicMax(map + (cell < map_words ? cell : 0), (unsigned)__float_as_uint(v[q].x) | 1u);
        }
    }
    if (v[0].w == 1.2345678f) out[0] = 1u;
}
template <typename L>
double timed(L launch, double bytes) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    launch();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int i = 0; i < 5; i++) launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return 5.0 * bytes / (ms * 1e-3) / 1e12;
}
int main(int argc, char** argv) {
    const double gib = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n = (size_t)(gib * 1024.0 * 1024.0 * 1024.0 / 16.0);
    f4* buf;
    unsigned *out, *map;
    const size_t map_words = (n >> 17) * 465750ull + 465750ull;
    if (hipMalloc((void**)&buf, n * 16) != hipSuccess || hipMalloc((void**)&out, 64) != hipSuccess ||
        hipMalloc((void**)&map, map_words * 4) != hipSuccess) {
        printf("allocation failed\n");
        return 1;
    }
    (void)hipMemset(buf, 1, n * 16);
    (void)hipMemset(map, 0, map_words * 4);
    const double bytes = (double)n * 16.0;
    printf("streaming %.2f GiB of 16-byte records once per launch (TB/s of records read; 5 launches)\n", gib);
#define GRID(PER, NT) printf("  grid, %d loads per thread, %s: %.2f\n", PER, NT ? "non-temporal" : "plain", \
    timed([&] { k_grid<PER, NT><<<(unsigned)((n + 256 * PER - 1) / (256 * PER)), 256>>>(buf, n, out); }, bytes))
    GRID(1, true); GRID(2, true); GRID(4, true); GRID(8, true); GRID(4, false); GRID(8, false);
#define PERS(PER, NT, BPC) printf("  persistent, %d blocks per CU, %d loads per thread, %s: %.2f\n", BPC, PER, NT ? "non-temporal" : "plain", \
    timed([&] { k_persistent<PER, NT><<<256 * BPC, 256>>>(buf, n, out); }, bytes))
    PERS(4, true, 4); PERS(4, true, 8); PERS(8, true, 4); PERS(8, true, 8); PERS(4, false, 8);
    printf("  grid, 4 loads per thread, non-temporal, + a 4-byte atomicMax per 4th record into a %.2f GB map: %.2f\n",
           map_words * 4 / 1e9,
           timed([&] { k_grid_scatter<4><<<(unsigned)((n + 1023) / 1024), 256>>>(buf, n, map, map_words, out); }, bytes));
    (void)hipFree(buf);
    (void)hipFree(map);
    (void)hipFree(out);
    return 0;
}
