// Does a transfer from page-locked host memory (by a kernel, or by hipMemcpyAsync) overlap GPU-filling kernels of ANOTHER stream?
// Three streams in rotation, each: [copy 19 MB host -> device] -> [a "stage" of 16 dependent GPU-filling kernels, ~0.4 ms]; a fourth
// stream runs a short "map" stage per group after the group's stage (event), and a group's copy waits for the map stage of the
// group three before (event) -- the dependency pattern of dsm_replay_enqueue_host.  Prints groups/s for: no copy, SDMA copy, kernel copy.
//     hipcc --offload-arch=gfx950 -O2 tools/micro/copy_overlap.hip -o /tmp/copy_overlap && /tmp/copy_overlap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void busy(float *p, int iters) {
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; i++) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ void copyk(v4 *dst, const v4 *src, unsigned n) {
    const unsigned stride = gridDim.x * 256u;
    for (unsigned i0 = blockIdx.x * 256u + threadIdx.x; i0 < n; i0 += stride * 8) {
        v4 v[8];
        for (int q = 0; q < 8; q++) if (i0 + q * stride < n) v[q] = __builtin_nontemporal_load(src + i0 + q * stride);
        for (int q = 0; q < 8; q++) if (i0 + q * stride < n) dst[i0 + q * stride] = v[q];
    }
}
int main() {
    const size_t bytes = 19u << 20;
    const int L = 3, groups = 1200;
    hipStream_t lead[L], map;
    for (int i = 0; i < L; i++) CK(hipStreamCreateWithFlags(&lead[i], hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&map, hipStreamNonBlocking));
    void *host; CK(hipHostMalloc(&host, bytes * L, hipHostMallocDefault));
    void *dev; CK(hipMalloc(&dev, bytes * L));
    float *work; CK(hipMalloc((void **)&work, sizeof(float) * 256 * 8192 * (L + 1)));
    hipEvent_t ev_sp[L], ev_map[L];
    for (int i = 0; i < L; i++) { CK(hipEventCreateWithFlags(&ev_sp[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ev_map[i], hipEventDisableTiming)); }
    // the stage as a graph per lead (captured on a stream of its own), as the library launches it
    hipGraphExec_t gexec[L];
    for (int l = 0; l < L; l++) {
        hipStream_t cs; CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
        CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
        for (int k = 0; k < 16; k++) hipLaunchKernelGGL(busy, dim3(8192), dim3(256), 0, cs, work + (size_t)l * 256 * 8192, 95);
        hipGraph_t gr; CK(hipStreamEndCapture(cs, &gr));
        CK(hipGraphInstantiate(&gexec[l], gr, nullptr, nullptr, 0));
        CK(hipGraphDestroy(gr)); CK(hipStreamDestroy(cs));
    }
    for (int mode = 0; mode < 5; mode++) {
        const bool use_graph = mode >= 3;
        const int cmode = mode >= 3 ? mode - 3 : mode; // 3: graph, no copy; 4: graph + hipMemcpyAsync
        CK(hipDeviceSynchronize());
        const auto t0 = std::chrono::steady_clock::now();
        for (int g = 0; g < groups; g++) {
            const int l = g % L;
            if (g >= L) CK(hipStreamWaitEvent(lead[l], ev_map[l], 0));
            if (cmode == 1) CK(hipMemcpyAsync((char *)dev + l * bytes, (char *)host + l * bytes, bytes, hipMemcpyHostToDevice, lead[l]));
            if (cmode == 2) hipLaunchKernelGGL(copyk, dim3(512), dim3(256), 0, lead[l], (v4 *)((char *)dev + l * bytes), (const v4 *)((char *)host + l * bytes), (unsigned)(bytes / 16));
            if (use_graph) CK(hipGraphLaunch(gexec[l], lead[l]));
            else for (int k = 0; k < 16; k++) hipLaunchKernelGGL(busy, dim3(8192), dim3(256), 0, lead[l], work + (size_t)l * 256 * 8192, 95);
            CK(hipEventRecord(ev_sp[l], lead[l]));
            CK(hipStreamWaitEvent(map, ev_sp[l], 0));
            for (int k = 0; k < 16; k++) hipLaunchKernelGGL(busy, dim3(64), dim3(256), 0, map, work + (size_t)L * 256 * 8192, 2500);
            CK(hipEventRecord(ev_map[l], map));
        }
        const double te = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        CK(hipDeviceSynchronize());
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("mode %d (%s): %.1f groups/s, %.3f ms per group, host enqueue %.3f s of %.3f s%s\n", mode, mode == 0 ? "no copy" : mode == 1 ? "hipMemcpyAsync" : mode == 2 ? "copy kernel" : mode == 3 ? "graph, no copy" : "graph + hipMemcpyAsync",
               groups / dt, dt / groups * 1e3, te, dt, mode ? "" : "");
        if (cmode) printf("   copy rate if serial: %.1f GB/s per group-time\n", bytes / (dt / groups) / 1e9);
    }
    return 0;
}
