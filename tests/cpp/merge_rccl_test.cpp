// include/dsm_merge.h on one GPU: a communicator of ONE (ncclCommInitRank), three clouds -- a few surfels, none, 2 GB --
// merged through RCCL itself; the merged cloud must be the cloud.  Prints one JSON line.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dsm_merge.h"

#define CHECK(x) do { if (!(x)) { fprintf(stderr, "FAILED %s (line %d): %s\n", #x, __LINE__, dsm_merge_last_error()); return 1; } } while (0)

int main() {
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) { fprintf(stderr, "no device\n"); return 77; }
    CHECK(hipSetDevice(0) == hipSuccess);
    ncclUniqueId id;
    CHECK(ncclGetUniqueId(&id) == ncclSuccess);
    ncclComm_t comm;
    CHECK(ncclCommInitRank(&comm, 1, id, 0) == ncclSuccess);
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
    const int64_t sizes[3] = {1000, 0, 48000000};
    double ms[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
        const int64_t n = sizes[k];
        dsm_surfel *d_in = nullptr, *d_out = nullptr;
        std::vector<unsigned char> host((size_t)n * 44);
        for (size_t i = 0; i < host.size(); i++) host[i] = (unsigned char)((i * 2654435761u + k) >> 13);
        if (n) {
            CHECK(hipMalloc((void **)&d_in, host.size()) == hipSuccess);
            CHECK(hipMalloc((void **)&d_out, host.size()) == hipSuccess);
            CHECK(hipMemcpy(d_in, host.data(), host.size(), hipMemcpyHostToDevice) == hipSuccess);
            CHECK(hipMemset(d_out, 0, host.size()) == hipSuccess);
        }
        int64_t counts[1] = {-1};
        // too small a destination is reported with the counts filled in
        if (n) {
            CHECK(dsm_merge_clouds_rccl(comm, 1, 0, d_in, n, d_out, n - 1, counts, st) == DSM_E_CAPACITY);
            CHECK(counts[0] == n);
        }
        CHECK(dsm_merge_clouds_rccl(comm, 1, 0, d_in, n, d_out, n, counts, st) == DSM_OK); // (first call of a size: buffers, connections)
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(dsm_merge_clouds_rccl(comm, 1, 0, d_in, n, d_out, n, counts, st) == DSM_OK);
        ms[k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        CHECK(counts[0] == n);
        if (n) {
            std::vector<unsigned char> back(host.size());
            CHECK(hipMemcpy(back.data(), d_out, back.size(), hipMemcpyDeviceToHost) == hipSuccess);
            CHECK(memcmp(back.data(), host.data(), host.size()) == 0);
            CHECK(hipFree(d_in) == hipSuccess);
            CHECK(hipFree(d_out) == hipSuccess);
        }
    }
    CHECK(dsm_merge_clouds_rccl(comm, 2, 0, nullptr, 0, nullptr, 0, nullptr, st) == DSM_E_INVALID);
    int64_t c2[2];
    CHECK(dsm_merge_clouds_rccl(comm, 2, 0, nullptr, 0, nullptr, 0, c2, st) == DSM_E_INVALID); // the communicator has one rank
    CHECK(ncclCommDestroy(comm) == ncclSuccess);
    printf("{\"small_ms\": %.3f, \"empty_ms\": %.3f, \"surfels_2GB\": %lld, \"ms_2GB\": %.3f}\n", ms[0], ms[1], (long long)sizes[2], ms[2]);
    return 0;
}
