// micro-benchmark: dependent-chain latency of the ops the ordered sums are made of (one wave, empty GPU)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float *out, long long *cyc, float a0, double h, int n, int mode, unsigned mask) {
    float a = a0;
    long long t0 = clock64();
    if (mode == 0) {
#pragma unroll 16
        for (int i = 0; i < n; i++) a += a0; }
    else if (mode == 1) {
#pragma unroll 16
        for (int i = 0; i < n; i++) a = (float)((double)a + h); }
    else if (mode == 2) {
#pragma unroll 16
        for (int i = 0; i < n; i++) { float c = a + a0; float t = (float)((double)a + h); a = ((mask >> (i & 31)) & 1u) ? t : c; } }
    else if (mode == 3) { double d = a0;
#pragma unroll 16
        for (int i = 0; i < n; i++) d += h; a = (float)d; }
    else if (mode == 4) { double d = a0;
#pragma unroll 16
        for (int i = 0; i < n; i++) d += (double)(a0 * (float)i); a = (float)d; }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1 << 16);
    const char *names[] = {"f32 add chain", "(float)((double)a+h) chain", "select(core,tail) chain", "f64 add chain", "f64 add chain + mul/cvt feed"};
    for (int waves = 1; waves <= 8; waves *= 2)
        for (int mode = 0; mode < 5; mode++) {
            long long c[1];
            const int n = 4096;
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, 1.5f, 0.4, n, mode, 0x5a5a5a5au);
            hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, 1.5f, 0.4, n, mode, 0x5a5a5a5au);
            hipMemcpy(c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d waves/SIMD  %-32s %.1f cycles per element\n", waves, names[mode], (double)c[0] / n);
        }
    return 0;
}
