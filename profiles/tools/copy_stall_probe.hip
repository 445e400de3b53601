// Probe (LAB.md 5.16): does a small hipMemcpyAsync block the submitting thread when a host runs far ahead of the device?
// Two streams take turns; a "step" = one 25 KB host -> device copy (pinned or pageable source) + a kernel idling `us`
// microseconds.  The host queues `steps` steps without waiting, `reps` times, and the probe prints the longest single
// hipMemcpyAsync call of every repetition (normal: a few microseconds).
//   usage: copy_stall_probe [us=400] [steps=12] [reps=8] [pinned=1]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_idle(int us, int* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);  // 100 MHz clock
    if (sink && us < 0) *sink = 1;
}
int main(int argc, char** argv) {
    const int us = argc > 1 ? atoi(argv[1]) : 400, steps = argc > 2 ? atoi(argv[2]) : 12, reps = argc > 3 ? atoi(argv[3]) : 8;
    const bool pinned = argc > 4 ? atoi(argv[4]) != 0 : true;
    const size_t bytes = 25600;
    hipStream_t st[2];
    char *dev[2], *host;
    for (int i = 0; i < 2; i++) {
        (void)hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        (void)hipMalloc((void**)&dev[i], bytes);
    }
    if (pinned) (void)hipHostMalloc((void**)&host, bytes * 64, hipHostMallocDefault);
    else host = (char*)malloc(bytes * 64);
    using clk = std::chrono::steady_clock;
    for (int r = 0; r < reps; r++) {
        double worst = 0, total = 0;
        int worst_at = -1;
        const auto t0 = clk::now();
        for (int it = 0; it < steps; it++) {
            const int s = it & 1;
            const auto a = clk::now();
            (void)hipMemcpyAsync(dev[s], host + (size_t)(it % 64) * bytes, bytes, hipMemcpyHostToDevice, st[s]);
            const double d = std::chrono::duration<double, std::milli>(clk::now() - a).count();
            if (d > worst) worst = d, worst_at = it;
            hipLaunchKernelGGL(k_idle, dim3(256), dim3(64), 0, st[s], us, (int*)nullptr);
        }
        total = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        (void)hipStreamSynchronize(st[0]);
        (void)hipStreamSynchronize(st[1]);
        const double wall = std::chrono::duration<double, std::milli>(clk::now() - t0).count();
        printf("rep %d: submit %.3f ms, wall %.3f ms, longest hipMemcpyAsync %.3f ms (step %d)\n", r, total, wall, worst, worst_at);
    }
    return 0;
}
