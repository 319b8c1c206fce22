// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for a few integer ops,
// as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32; typedef unsigned long long u64;
template <int OP>
__global__ __launch_bounds__(256) void k(u32* out, u32 seed, int iters, u64* cyc) {
    u32 a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = seed * (i + 1) + threadIdx.x;
    u32 s = seed | 1u;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 8; rep++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) a[i] = a[i] ^ (a[(i + 1) & 7] + s);                       // v_add + v_xor  (2 ops)  -> may fuse to v_xad? count 2
                if (OP == 1) a[i] = __builtin_amdgcn_alignbit(a[i], a[(i + 3) & 7], s & 31u);  // v_alignbit
                if (OP == 2) a[i] = a[i] * 0x9E3779B1u + 1u;                           // v_mul_lo_u32 (+add -> v_mad_u32_u24? no: mul_lo + add)
                if (OP == 3) { u64 v = ((u64)a[i] << 32 | a[(i + 1) & 7]) >> (s & 31u); a[i] = (u32)v; }   // v_lshrrev_b64
                if (OP == 4) a[i] = a[i] > s ? a[(i + 1) & 7] : a[i];                  // v_cmp + v_cndmask (2 ops)
                if (OP == 5) a[i] = min(a[i], a[(i + 1) & 7] ^ s);                     // v_xor + v_min (2 ops)
            }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    u32 r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, int opsPerStep) {
    u32* out; u64* cyc; hipMalloc(&out, 256 * 1024 * 64 * sizeof(u32)); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {                       // waves per SIMD = blocks of 256 threads (4 waves, one per SIMD) per CU
        const int blocks = 256 * wps;
        k<OP><<<blocks, 256>>>(out, 12345u, iters, cyc); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<OP><<<blocks, 256>>>(out, 12345u, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); u64 c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double inst = (double)iters * 64 * opsPerStep;                 // per wave
        printf("%-22s waves/SIMD %d: %.2f cycles per wave-instruction (one wave's view), %.2f cycles/instr/SIMD (aggregate), %.3f ms\n",
               name, wps, (double)c / inst, (double)c / (inst * wps), ms);
    }
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("add+xor", 2); run<1>("alignbit", 1); run<2>("mul_lo+add", 2); run<3>("lshrrev_b64", 1); run<4>("cmp+cndmask", 2); run<5>("xor+min", 2);
    return 0;
}
