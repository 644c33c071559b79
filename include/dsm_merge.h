/* dsm_merge.h -- BASELINE configs[2]'s last step from host C++: the ranks of a sharded replay merge
 * their final surfel clouds with two RCCL all-gathers over xGMI -- the counts, then the clouds padded to the largest count --
 * and every rank ends up with all clouds, rank after rank.  The reference has no counterpart (one process, one map:
 * surfel_map.cpp:1060-1113 ends a frame, nothing merges maps); the Python driver's version of this is
 * densesurfelmapping_amd/replay.py, merge_clouds (torch.distributed).
 *
 * A separate small library (libdsm_merge_rccl.so, links librccl) so that libdsm_hip.so itself has no RCCL dependency.  The
 * communicator is the caller's (ncclCommInitRank with an id it distributed by its own means -- MPI, a file, a socket); it is
 * passed as void* so that this header needs no RCCL header. */
#ifndef DSM_MERGE_H
#define DSM_MERGE_H
#include <stdint.h>
#include "dsm.h"
#ifdef __cplusplus
extern "C" {
#endif

/* d_cloud: this rank's n surfels in device memory (dsm_map_copy_to_device gives them without a host trip; n may be 0, d_cloud
 * then may be NULL).  d_merged: device memory for cap surfels; on return the clouds of ranks 0 .. world-1 back to back.
 * counts[world] (host): every rank's count.  Returns DSM_OK, DSM_E_CAPACITY when cap is smaller than the sum (counts is valid:
 * the caller can allocate and call again), DSM_E_HIP for a runtime or RCCL error (dsm_merge_last_error).  Collective: every rank
 * of the communicator calls it; synchronous (the stream is drained before it returns). */
int dsm_merge_clouds_rccl(void *nccl_comm, int world, int rank, const dsm_surfel *d_cloud, int64_t n, dsm_surfel *d_merged,
                          int64_t cap, int64_t *counts, void *hip_stream);
const char *dsm_merge_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
