// dsm_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the per-frame surfel-fusion hot path.
//
// Reference functions covered ("FF.cpp" = surfel_fusion/src/fusion_functions.cpp, "SM.cpp" =
// surfel_fusion/src/surfel_map.cpp of the reference):
//   k_init_seeds    initialize_seeds_kernel          FF.cpp:577-629
//   k_assign        update_pixels_kernel             FF.cpp:389-453 (+ calculate_cost 364-387)
//   k_resolve       the sequential `stable` skip rule of FF.cpp:400,445,450 as a fixed point
//   k_update_seeds  update_seeds_kernel              FF.cpp:468-562 (+ the new label image of the sweep)
//   k_commit_seeds  the early `return` of FF.cpp:516-517 (per worker chunk)
//   k_pixel_normals calculate_pixels_norms_kernel (the normals that are read)   FF.cpp:664-712
//   k_seed_stats    calculate_spaces / calculate_sp_depth_norms up to the fit's starting point
//                                                    FF.cpp:644-662, 792-871, 104-120
//   k_seed_fit      get_huber_norm's Gauss-Newton steps, the seed's plane / position / view angle
//                                                    FF.cpp:128-188, 872-914
//                   and the per-seed part of initialize_surfels, FF.cpp:315-361
//   k_fuse_surfels  fuse_surfels_kernel              FF.cpp:190-313
//   k_frame_tail    initialize_surfels (the `fused` test and the ordered list), FF.cpp:315-361;
//                   SurfelMap::fuse_map refill + swap-with-last, SM.cpp:1077-1109
//   k_warp          warp_{active,inactive}_surfels_cpu_kernel          SM.cpp:681-789
//   k_mark_key / k_scan_marks / k_extract_marked   move_add_surfels removal, SM.cpp:1476-1497
//
// One translation unit; the kernels by stage: dsm_k_superpixel.h (k_init_seeds .. k_commit_seeds), dsm_k_planes.h (k_seed_points,
// k_pixel_normals, k_seed_stats, k_seed_fit, k_seed_finish), dsm_k_map.h (k_fuse_surfels, k_frame_tail, k_warp, the active-set
// and upload kernels), over dsm_k_common.h; this file holds the launcher of a frame's sixteen stages.
//
// Build with -ffp-contract=off: results are required to match the CPU reference bit for bit.
// The work is stencil / gather / ordered reduction: no MFMA (nothing is a dense contraction).  The superpixel kernels
// are bound by VALU instruction issue (mixed fp32 / fp64 scalar-style arithmetic in the reference's order), the
// map-sized ones (k_fuse_surfels, k_warp) by HBM.  Every frame kernel exists twice: for one handle (context in the
// kernel arguments) and, BATCH, for several handles advancing in lockstep (context array, handle = low bits of the
// dispatch index: one handle per XCD) -- see DESIGN.md section 4.
#include "dsm_device.h"

#include "dsm_k_common.h"
#include "dsm_k_superpixel.h"
#include "dsm_k_planes.h"
#include "dsm_k_map.h"

namespace dsm {

// ------------------------------------------------------------------------------ launcher
const char *const kStageNames[kNumStages] = {
    "init_seeds", "assign_0",  "update_seeds_0", "commit_seeds_0", "assign_1",    "resolve_1",    "update_seeds_1", "commit_seeds_1",
    "assign_2",   "resolve_2", "update_seeds_2", "commit_seeds_2", "seed_points", "seed_fit", "fuse_surfels", "frame_tail",
};

// d_batch != nullptr: the kernels take their context from d_batch[blockIdx.z], z < n_batch (handles of equal geometry
// advancing in lockstep: one launch per kernel for all of them); hc is then any one of them (grid sizes).
hipError_t launch_frame(const DeviceCtx &hc, int map_upper_bound, int tail_map_bound, bool with_compaction,
                        hipStream_t st, hipEvent_t *ev, int stage_lo, int stage_hi, const DeviceCtx *d_batch, int n_batch, int lanes_from) {
    int stage = 0;
    hipError_t err = hipSuccess;
    const bool batched = d_batch != nullptr;
    const unsigned nz = batched ? (unsigned)n_batch : 1u;
// launches only the stages whose index (position in kStageNames) lies in [stage_lo, stage_hi]
#define hipLaunchStage(kernel_single, kernel_batch, grid, block, ...)                                        \
    do {                                                                                                     \
        if (stage - 1 >= stage_lo && stage - 1 <= stage_hi) {                                                \
            const dim3 g_((grid).x, (grid).y, nz);                                                           \
            if (batched) hipLaunchKernelGGL(kernel_batch, g_, block, 0, st, hc, d_batch, ##__VA_ARGS__);   \
            else hipLaunchKernelGGL(kernel_single, g_, block, 0, st, hc, d_batch, ##__VA_ARGS__);          \
        }                                                                                                    \
    } while (0)
#define DSM_MARK()                                                                      \
    do {                                                                                \
        if (ev) {                                                                       \
            err = hipEventRecord(ev[stage], st);                                        \
            if (err != hipSuccess) return err;                                          \
        }                                                                               \
        stage++;                                                                        \
    } while (0)
    const int S = hc.n_seed;
    const dim3 g_seed_thr((S + 255) / 256);
    const dim3 g_seed_wave((S + 3) / 4);
    const dim3 g_seed_lane((S + 63) / 64); // one lane per seed
    const dim3 g_seed_rest((S + 63) / 64 + kRestOverBlocks); // packed queue entries, then the seeds with oversized lists
    // Two forms of the per-seed stages (same results): a wave per seed where the launch's latency counts -- one handle, or
    // the few of a frame group -- and a lane per seed (and four pixels per thread in k_assign) where the instructions
    // issued count: launches batched over kLaneBatch handles or more.
    const bool lanes = batched && n_batch >= (lanes_from > 0 ? lanes_from : kLaneBatch);
    const dim3 g_tile1((hc.w + kTileW - 1) / kTileW, (hc.h + AssignTile<1>::kH - 1) / AssignTile<1>::kH);
    const dim3 g_tile4((hc.w + kTileW - 1) / kTileW, (hc.h + AssignTile<4>::kH - 1) / AssignTile<4>::kH);
    const dim3 g_row8((hc.pitch / 8 + 63) / 64, (hc.h + 3) / 4); // thread per eight pixels of a row, 512 x 4 per block
    if (ev) hipLaunchKernelGGL(k_delay, dim3(1), dim3(64), 0, st, 40000LL); // 400 us
    DSM_MARK();
    if (lanes) hipLaunchStage(k_init_seeds_lanes<true>, k_init_seeds_lanes<true>, g_seed_thr, dim3(256));
    else hipLaunchStage(k_init_seeds<false>, k_init_seeds<true>, dim3((S + kInitSeedsPerBlock - 1) / kInitSeedsPerBlock), dim3(256));
    DSM_MARK();
    for (int sweep = 0; sweep < kSweeps; sweep++) {
        if (sweep == 0) {
            if (lanes) hipLaunchStage((k_assign<true, true, 4>), (k_assign<true, true, 4>), g_tile4, dim3(256), sweep);
            else hipLaunchStage((k_assign<true, false, 1>), (k_assign<true, true, 1>), g_tile1, dim3(256), sweep);
            DSM_MARK();
        } else {
            if (lanes) hipLaunchStage((k_assign<false, true, 4>), (k_assign<false, true, 4>), g_tile4, dim3(256), sweep);
            else hipLaunchStage((k_assign<false, false, 1>), (k_assign<false, true, 1>), g_tile1, dim3(256), sweep);
            DSM_MARK();
            hipLaunchStage(k_resolve<false>, k_resolve<true>, dim3(1), dim3(256), sweep);
            hipLaunchStage(k_apply_labels<false>, k_apply_labels<true>, dim3(g_row8.x, (hc.h + 4 * kApplyRows - 1) / (4 * kApplyRows)), dim3(256), sweep); // (part of the resolve stage: the sweep's label image)
            DSM_MARK();
        }
        if (lanes) {
            hipLaunchStage(k_update_seeds<true>, k_update_seeds<true>, g_seed_lane, dim3(64), sweep);
            hipLaunchStage(k_update_seeds_rest<true>, k_update_seeds_rest<true>, g_seed_rest, dim3(64), sweep);
        } else {
            hipLaunchStage(k_update_seeds_wave<false>, k_update_seeds_wave<true>, g_seed_wave, dim3(256), sweep);
        }
        DSM_MARK();
        hipLaunchStage(k_commit_seeds<false>, k_commit_seeds<true>, g_seed_thr, dim3(256), sweep);
        DSM_MARK();
    }
    if (lanes) {
        hipLaunchStage(k_pixel_normals<true>, k_pixel_normals<true>, dim3((hc.w + 63) / 64, (hc.h + 4 * kNormalRows - 1) / (4 * kNormalRows)), dim3(256));
        hipLaunchStage(k_seed_stats<true>, k_seed_stats<true>, g_seed_lane, dim3(64));
    } else {
        hipLaunchStage(k_seed_points<false>, k_seed_points<true>, g_seed_wave, dim3(256));
    }
    DSM_MARK();
    hipLaunchStage((k_seed_fit<false, kFitAll>), (k_seed_fit<true, kFitSmall>), dim3((S + kFitSeeds - 1) / kFitSeeds), dim3(64));
    if (batched) hipLaunchStage((k_seed_fit<false, kFitAll>), (k_seed_fit<true, kFitLarge>), dim3(kFitLargeBlocks), dim3(64));
    hipLaunchStage(k_seed_finish<false>, k_seed_finish<true>, g_seed_thr, dim3(256));
    DSM_MARK();
    // grid-stride over the map with no more workgroups than the device holds at once (the ones that start late would run
    // all their trips after the others have finished theirs), and fewer than that: see resident_blocks
    int fuse_blocks = (map_upper_bound + 255) / 256;
    if (fuse_blocks < 1) fuse_blocks = 1;
    const int fuse_cap = batched ? resident_blocks(k_fuse_surfels<true>, 256, kFuseBlocksPerCu) : resident_blocks(k_fuse_surfels<false>, 256, kFuseBlocksPerCu);
    if (fuse_blocks > fuse_cap) fuse_blocks = fuse_cap;
    // ... for the launch as a whole: the handles of a batch share the cap (round 6).  Capped per handle, a launch over 32 handles
    // was 24 000 workgroups of one or two trips each, every one of them setting up its context, constants and the two
    // matrices first (80 scalar instructions per wave); +2.6 % on the headline over five alternating runs
    if (batched && nz > 1 && fuse_blocks > (fuse_cap / nz > 1 ? fuse_cap / nz : 1)) fuse_blocks = fuse_cap / nz > 1 ? fuse_cap / nz : 1;
    hipLaunchStage(k_fuse_surfels<false>, k_fuse_surfels<true>, dim3(fuse_blocks), dim3(256));
    DSM_MARK();
    // (a map that may be beyond the tail's one-workgroup path gets a workgroup per chunk of its hole bitmap on top; they
    // leave at once while the map is small, but starting them is not free -- 8 us per launch for eight handles -- so
    // the callers pass 0 until the map can be that large, dsm_api.hip: tail_bound)
    int tail_blocks = 1;
    if (with_compaction && tail_map_bound > kTailFastWords * 64) {
        tail_blocks = 1 + (tail_map_bound / 64 + kTailChunkWords) / kTailChunkWords;
        if (tail_blocks > kTailMaxBlocks) tail_blocks = kTailMaxBlocks;
    }
    hipLaunchStage(k_frame_tail<false>, k_frame_tail<true>, dim3(tail_blocks), dim3(1024), with_compaction ? 1 : 0);
    DSM_MARK();
    if (ev) { // empty interval: what a pair of event records costs by itself
        err = hipEventRecord(ev[stage], st);
        if (err != hipSuccess) return err;
    }
#undef DSM_MARK
#undef hipLaunchStage
    return hipGetLastError();
}

} // namespace dsm
