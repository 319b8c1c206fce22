// hipMalloc / hipFree time by block size (fresh each time): where does "fresh HBM costs ~50 ms per GB to map" start and stop?
// build: hipcc --offload-arch=gfx950 -O2 -o malloc_sizes malloc_sizes.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t MB = 1ull << 20;
    const size_t sizes[] = {256 * MB, 1024 * MB, 2048 * MB, 4096 * MB, 8192 * MB, 12288 * MB, 16384 * MB, 20480 * MB, 24576 * MB, 32768 * MB, 40960 * MB, 49152 * MB, 65536 * MB, 16384 * MB, 16000 * MB, 8192 * MB};
    for (size_t sz : sizes) {
        void* p = nullptr; const double t0 = now();
        if (hipMalloc(&p, sz) != hipSuccess) { printf("%zu MB: failed\n", sz / MB); continue; }
        const double t1 = now(); hipMemset(p, 0, 4096); hipDeviceSynchronize(); const double t2 = now(); hipFree(p); const double t3 = now();
        printf("%6zu MB: hipMalloc %8.1f ms, hipFree %6.1f ms\n", sz / MB, t1 - t0, t3 - t2);
    }
    // the same sizes a second time round (anything cached by the runtime?)
    for (size_t sz : {16384 * MB, 16384 * MB, 20480 * MB}) {
        void* p = nullptr; const double t0 = now(); if (hipMalloc(&p, sz) != hipSuccess) continue; const double t1 = now(); hipFree(p);
        printf("again %6zu MB: hipMalloc %8.1f ms\n", sz / MB, t1 - t0);
    }
    return 0;
}
