// How long does device memory take to come back in another shape?  hipMalloc / hipFree against hipMallocAsync / hipFreeAsync on a pool that keeps what it is given
// (release threshold = max): two 40 GB blocks, freed, then four 15 GB + one 20 GB blocks, freed, and round again -- the pattern of the memory-diet mode's phases.
// build: hipcc --offload-arch=gfx950 -O2 -o mempool mempool.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void touch(char* p, size_t n) { size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4096; if (i < n) p[i] = 1; }
int main() {
    const size_t GB = 1ull << 30;
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int mode = 0; mode < 2; mode++) {
        hipMemPool_t pool = nullptr;
        if (mode == 1) { CK(hipDeviceGetDefaultMemPool(&pool, 0)); uint64_t thr = ~0ull; CK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &thr)); }
        for (int round = 0; round < 3; round++) {
            for (int shape = 0; shape < 2; shape++) {
                std::vector<size_t> sz = shape == 0 ? std::vector<size_t>{40 * GB, 40 * GB} : std::vector<size_t>{15 * GB, 15 * GB, 15 * GB, 15 * GB, 20 * GB};
                std::vector<void*> p(sz.size());
                const double t0 = now();
                for (size_t i = 0; i < sz.size(); i++) { if (mode == 0) CK(hipMalloc(&p[i], sz[i])); else CK(hipMallocAsync(&p[i], sz[i], s)); }
                CK(hipStreamSynchronize(s));
                const double t1 = now();
                for (size_t i = 0; i < sz.size(); i++) hipLaunchKernelGGL(touch, dim3((unsigned)((sz[i] / 4096 + 255) / 256)), dim3(256), 0, s, (char*)p[i], sz[i]);
                CK(hipStreamSynchronize(s));
                const double t2 = now();
                for (size_t i = 0; i < sz.size(); i++) { if (mode == 0) CK(hipFree(p[i])); else CK(hipFreeAsync(p[i], s)); }
                CK(hipStreamSynchronize(s));
                const double t3 = now();
                printf("%s round %d shape %d: alloc %.1f ms, first touch %.1f ms, free %.1f ms\n", mode ? "pool (async)" : "hipMalloc", round, shape, t1 - t0, t2 - t1, t3 - t2);
            }
        }
    }
    return 0;
}
