// micro-benchmark: issue cost of the instruction mix of the Gauss-Newton sums (one wave per SIMD, empty GPU):
// cycles per element for  acc += (double)(x*y)  with 1, 2 and 4 independent accumulators, and of its parts
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *out, long long *cyc, const float *xs, int n, int mode) {
    const float x0 = xs[threadIdx.x & 63], y0 = xs[64 + (threadIdx.x & 63)];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    float f0 = x0, f1 = y0, f2 = x0 + 1, f3 = y0 + 1;
    long long t0 = clock64();
    if (mode == 0) { // one chain: mul, cvt, add
#pragma unroll 8
        for (int i = 0; i < n; i++) { a0 += (double)(f0 * f1); f0 += 1.0f; }
    } else if (mode == 1) { // two chains
#pragma unroll 8
        for (int i = 0; i < n; i++) { a0 += (double)(f0 * f1); a1 += (double)(f2 * f3); f0 += 1.0f; f2 += 1.0f; }
    } else if (mode == 2) { // four chains
#pragma unroll 8
        for (int i = 0; i < n; i++) { a0 += (double)(f0 * f1); a1 += (double)(f2 * f3); a2 += (double)(f0 * f3); a3 += (double)(f2 * f1); f0 += 1.0f; f2 += 1.0f; }
    } else if (mode == 3) { // cvt only, independent
#pragma unroll 8
        for (int i = 0; i < n; i++) { a0 = (double)f0; a1 = (double)f1; a2 = (double)f2; a3 = (double)f3; f0 += 1.0f; f1 += 1.0f; f2 += 1.0f; f3 += 1.0f; asm volatile("" :: "v"(a0), "v"(a1), "v"(a2), "v"(a3)); }
    } else if (mode == 4) { // f64 add only, four independent chains
        const double h = x0;
#pragma unroll 8
        for (int i = 0; i < n; i++) { a0 += h; a1 += h; a2 += h; a3 += h; }
    } else if (mode == 5) { // f64 add, one chain
        const double h = x0;
#pragma unroll 8
        for (int i = 0; i < n; i++) { a0 += h; }
    } else if (mode == 6) { // f32 mul only, four independent
#pragma unroll 8
        for (int i = 0; i < n; i++) { f0 *= 1.0001f; f1 *= 1.0001f; f2 *= 1.0001f; f3 *= 1.0001f; }
    } else if (mode == 7) { // f64 fma, four independent chains
        const double h = x0;
#pragma unroll 8
        for (int i = 0; i < n; i++) { a0 = __builtin_fma(a0, 1.0000001, h); a1 = __builtin_fma(a1, 1.0000001, h); a2 = __builtin_fma(a2, 1.0000001, h); a3 = __builtin_fma(a3, 1.0000001, h); }
    }
    long long t1 = clock64();
    out[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    double *out; long long *cyc; float *xs;
    hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1 << 16); hipMalloc(&xs, 1024);
    float h[128]; for (int i = 0; i < 128; i++) h[i] = 1.0f + i * 0.01f;
    hipMemcpy(xs, h, sizeof h, hipMemcpyHostToDevice);
    const char *names[] = {"acc += (double)(x*y), 1 chain", "same, 2 chains", "same, 4 chains", "4 independent v_cvt_f64_f32", "4 independent f64 add chains", "1 f64 add chain",
                           "4 independent f32 mul chains", "4 independent f64 fma chains"};
    const int per[] = {1, 2, 4, 4, 4, 1, 4, 4};
    for (int waves = 1; waves <= 2; waves++)
        for (int mode = 0; mode < 8; mode++) {
            long long c[1];
            const int n = 4096;
            for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64 * waves * 4), 0, 0, out, cyc, xs, n, mode);
            hipMemcpy(c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%d wave(s)/SIMD  %-36s %.1f cycles per loop pass = %.1f per accumulator update\n", waves, names[mode], (double)c[0] / n, (double)c[0] / n / per[mode]);
        }
    return 0;
}
