// dsm_merge_rccl.cpp -- include/dsm_merge.h: the final clouds of a sharded replay merged by two RCCL all-gathers.
// One process per GPU; xGMI is point to point and fully connected, so an all-gather of W equal blocks is W - 1 direct
// transfers per rank -- the clouds are padded to the largest count for that, and compacted on the device afterwards.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/dsm_merge.h"

namespace {
thread_local std::string g_err;
int fail(int code, const char *what, const char *detail) {
    g_err = std::string(what) + ": " + detail;
    return code;
}
struct DeviceBuf {
    void *p = nullptr;
    ~DeviceBuf() { if (p) (void)hipFree(p); }
};
} // namespace

#define M_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(DSM_E_HIP, #x, hipGetErrorString(e_)); } while (0)
#define M_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(DSM_E_HIP, #x, ncclGetErrorString(r_)); } while (0)

extern "C" const char *dsm_merge_last_error(void) { return g_err.c_str(); }

extern "C" int dsm_merge_clouds_rccl(void *nccl_comm, int world, int rank, const dsm_surfel *d_cloud, int64_t n, dsm_surfel *d_merged,
                                     int64_t cap, int64_t *counts, void *hip_stream) {
    static_assert(sizeof(dsm_surfel) == 44, "SurfelElement is 44 bytes (elements.h:22-31)");
    if (!nccl_comm || world < 1 || rank < 0 || rank >= world || n < 0 || cap < 0 || !counts || (n > 0 && !d_cloud))
        return fail(DSM_E_INVALID, "dsm_merge_clouds_rccl", "null / negative / out-of-range argument");
    ncclComm_t comm = (ncclComm_t)nccl_comm;
    hipStream_t st = (hipStream_t)hip_stream;
    int seen = 0;
    M_NCCL(ncclCommCount(comm, &seen));
    if (seen != world) return fail(DSM_E_INVALID, "dsm_merge_clouds_rccl", "the communicator's size is not `world`");
    // 1. the counts
    DeviceBuf d_counts;
    M_HIP(hipMalloc(&d_counts.p, sizeof(int64_t) * (size_t)(world + 1)));
    int64_t *dc = (int64_t *)d_counts.p;
    M_HIP(hipMemcpyAsync(dc + world, &n, sizeof n, hipMemcpyHostToDevice, st));
    M_NCCL(ncclAllGather(dc + world, dc, 1, ncclInt64, comm, st));
    M_HIP(hipMemcpyAsync(counts, dc, sizeof(int64_t) * (size_t)world, hipMemcpyDeviceToHost, st));
    M_HIP(hipStreamSynchronize(st));
    int64_t total = 0, largest = 0;
    for (int r = 0; r < world; r++) {
        if (counts[r] < 0) return fail(DSM_E_HIP, "dsm_merge_clouds_rccl", "a rank reported a negative count");
        total += counts[r];
        largest = counts[r] > largest ? counts[r] : largest;
    }
    if (total > cap) return fail(DSM_E_CAPACITY, "dsm_merge_clouds_rccl", "the merged cloud does not fit `cap` (counts[] is valid)");
    if (total == 0) return DSM_OK; // (every rank empty: nothing to gather -- and every rank knows it)
    if (!d_merged) return fail(DSM_E_INVALID, "dsm_merge_clouds_rccl", "d_merged is null");
    // 2. the clouds, padded to the largest count (equal blocks: one direct transfer per peer)
    const size_t block = (size_t)largest * sizeof(dsm_surfel);
    if (world == 1) { // a group of one still goes through the library: the gather of one block onto itself
        M_NCCL(ncclAllGather(d_cloud, d_merged, block, ncclUint8, comm, st));
        M_HIP(hipStreamSynchronize(st));
        return DSM_OK;
    }
    DeviceBuf mine, all;
    M_HIP(hipMalloc(&mine.p, block));
    M_HIP(hipMalloc(&all.p, block * (size_t)world));
    if (n > 0) M_HIP(hipMemcpyAsync(mine.p, d_cloud, (size_t)n * sizeof(dsm_surfel), hipMemcpyDeviceToDevice, st));
    M_NCCL(ncclAllGather(mine.p, all.p, block, ncclUint8, comm, st));
    // 3. rank after rank, without the padding
    size_t at = 0;
    for (int r = 0; r < world; r++) {
        const size_t bytes = (size_t)counts[r] * sizeof(dsm_surfel);
        if (bytes) M_HIP(hipMemcpyAsync((char *)d_merged + at, (const char *)all.p + block * (size_t)r, bytes, hipMemcpyDeviceToDevice, st));
        at += bytes;
    }
    M_HIP(hipStreamSynchronize(st));
    return DSM_OK;
}
