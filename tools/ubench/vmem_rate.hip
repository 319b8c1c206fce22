// L1 (TA/TCP) issue-rate microbenchmark for gfx950: cycles per wave64 global-load instruction per CU when everything hits the L1,
// for the access shapes of the probe kernel: dword loads (tag scan), dwordx2, dwordx4 (read slots), with LINES distinct 128-byte
// lines per instruction.  Build: hipcc --offload-arch=gfx950 -O3 vmem_rate.hip -o vmem_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32; typedef unsigned long long u64;
template <int W, int LINES>
__global__ __launch_bounds__(256) void k(const u32* __restrict__ tab, u32* out, int iters, u64* cyc) {
    const u32 lane = threadIdx.x & 63;
    // lanes share LINES distinct 128-byte lines; inside a line the lane's 16 loads walk 8-byte records like the scan does
    const u32* p = tab + (size_t)(lane % LINES) * 32 + (size_t)(blockIdx.x & 7) * 4096;
    u32 acc = 0;
    const u64 t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        const u32* q = p + ((it & 3) * 1024);
        if (W == 1) {
#pragma unroll
            for (int u = 0; u < 16; u++) acc += q[2 * u + 1] ^ (u32)u;       // 16 dword loads (high halves of 16 records)
        } else if (W == 2) {
#pragma unroll
            for (int u = 0; u < 16; u++) { const uint2 v = *(const uint2*)(q + 2 * u); asm volatile("" :: "v"(v.x)); acc += v.y ^ (u32)u; }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint4 v = *(const uint4*)(q + 4 * u); asm volatile("" :: "v"(v.x), "v"(v.z)); acc += (v.y ^ (u32)u) + v.w; }
        }
    }
    const u64 t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int W, int LINES> void run(const char* name, int instrPerIter) {
    u32* tab; u32* out; u64* cyc; hipMalloc(&tab, 1 << 20); hipMemset(tab, 1, 1 << 20); hipMalloc(&out, 256 * 8 * 256 * sizeof(u32)); hipMalloc(&cyc, 8);
    const int iters = 4000;
    for (int wps : {1, 2, 4}) {
        const int blocks = 256 * wps;                   // one block of 4 waves per CU and per wps
        k<W, LINES><<<blocks, 256>>>(tab, out, iters, cyc); hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); k<W, LINES><<<blocks, 256>>>(tab, out, iters, cyc); hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); u64 c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        const double inst = (double)iters * instrPerIter;                   // per wave
        // s_memtime ticks at 100 MHz on this part: report wall time per CU-instruction instead (ms * 2.4e6 cycles / (instr * waves per CU))
        printf("%-26s waves/SIMD %d: %.1f ns per wave-instruction per CU = %.1f cycles @2.4GHz (%.3f ms)\n", name, wps,
               ms * 1e6 / (inst * 4 * wps), ms * 1e6 / (inst * 4 * wps) * 2.4, ms);
    }
    hipFree(tab); hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 1>("dword, 1 line", 16); run<1, 8>("dword, 8 lines", 16); run<1, 64>("dword, 64 lines", 16);
    run<2, 8>("dwordx2, 8 lines", 16); run<4, 1>("dwordx4, 1 line", 8); run<4, 8>("dwordx4, 8 lines", 8); run<4, 64>("dwordx4, 64 lines", 8);
    return 0;
}
