// does the shader clock (and s_memtime) depend on how many CUs are busy?  A: one 1024-thread workgroup with a fixed integer-VALU loop;
// B: many workgroups spinning on integer VALU until a flag is set.  Compare A's wall time and tick count alone vs next to B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
__global__ void kA(int *out, unsigned long long *ticks, int iters) {
    int v = threadIdx.x, w = v * 3 + 1;
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) { v = v * 1664525 + w; w = max(w ^ v, v + i); v = min(v, w) + (w >> 3); }
    const long long t1 = clock64();
    if (threadIdx.x == 0) { *ticks = (unsigned long long)(t1 - t0); }
    out[threadIdx.x] = v + w;
}
__global__ void kB(int *out, int nouter, int mem) {
    int v = threadIdx.x + blockIdx.x, w = v * 3 + 1; int n = 0;
    while (n < nouter) {
        for (int i = 0; i < 4096; ++i) { v = v * 1664525 + w; w = max(w ^ v, v + i); v = min(v, w) + (w >> 3); }
        if (mem) out[(blockIdx.x * 1024 + threadIdx.x + n * 65536) & ((1 << 26) - 1)] = v;
        n += 1;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = v + w;
}
int main() {
    int *out, *outb; unsigned long long *ticks;
    hipMalloc(&out, 4096); hipMalloc(&outb, (1 << 26) * 4 + (1 << 22)); hipMalloc(&ticks, 8);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    const int iters = 4000000; const int nouter = 2500; // B lasts several times longer than A
    for (int mode = 0; mode < 4; ++mode) { // 0: alone, 1: with 255 WGs x1024 VALU, 2: with 2040 WGs x 64 (narrow), 3: narrow + stores
        if (mode == 1) hipLaunchKernelGGL(kB, dim3(255), dim3(1024), 0, s2, outb, nouter, 0);
        if (mode == 2) hipLaunchKernelGGL(kB, dim3(4000), dim3(64), 0, s2, outb, nouter, 0);
        if (mode == 3) hipLaunchKernelGGL(kB, dim3(4000), dim3(64), 0, s2, outb, nouter, 1);
        auto t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(kA, dim3(1), dim3(1024), 0, s1, out, ticks, iters);
        hipStreamSynchronize(s1);
        auto t1 = std::chrono::steady_clock::now();
        auto t2 = std::chrono::steady_clock::now(); hipStreamSynchronize(s2); auto t3 = std::chrono::steady_clock::now();
        unsigned long long tk; hipMemcpy(&tk, ticks, 8, hipMemcpyDeviceToHost);
        const double sec = std::chrono::duration<double>(t1 - t0).count();
        printf("mode %d: A wall %.3f s, ticks %.3e, ticks/s %.3e  (B ran %.3f s longer)\n", mode, sec, (double)tk, tk / sec, std::chrono::duration<double>(t3 - t2).count()); fflush(stdout);
    }
    return 0;
}
