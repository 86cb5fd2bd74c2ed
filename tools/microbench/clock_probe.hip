// clock_probe.hip -- shader clock seen by a wave (s_memtime / s_memrealtime x 100 MHz) when
// the chip is nearly idle (one wave) vs busy (every SIMD running dependent FMAs).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long* out, int iters, float seed) {
    float a = seed + threadIdx.x;
    unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 32; ++k) a = __builtin_fmaf(a, 1.0000001f, 1e-7f);
    }
    unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = r1 - r0; }
    if (a == 123.0f) out[0] = 0;
}
int main() {
    unsigned long long* d; hipMalloc(&d, 4096 * 16);
    struct { int blocks, threads, iters; const char* name; } cfgs[] = {
        {1, 64, 20000, "1 wave on the chip, ~3 ms"}, {1, 64, 2000, "1 wave, ~0.3 ms"},
        {256, 64, 20000, "1 wave per CU"}, {256, 256, 20000, "1 wave per SIMD"},
        {2048, 256, 4000, "8 waves per SIMD (busy)"}};
    for (auto& c : cfgs) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(spin, dim3(c.blocks), dim3(c.threads), 0, 0, d, c.iters, 1.0f);
            hipDeviceSynchronize();
        }
        unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-28s memtime %llu realtime %llu -> %.0f MHz\n", c.name, h[0], h[1], 100.0 * double(h[0]) / double(h[1]));
    }
    // back-to-back short launches, as in the benchmark loop
    for (int rep = 0; rep < 2000; ++rep) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, 0, d, 1500, 1.0f);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-28s memtime %llu realtime %llu -> %.0f MHz\n", "after 2000 back-to-back", h[0], h[1], 100.0 * double(h[0]) / double(h[1]));
    return 0;
}
