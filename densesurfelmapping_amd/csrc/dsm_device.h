// dsm_device.h -- device-resident state of one handle, shared by dsm_kernels.hip and dsm_api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dsm.h"
#include "dsm_math.h"

namespace dsm {

// per-wave phase stamps of the per-seed kernels (dsm_debug_wave_stamps): compiled in only by -DDSM_WAVE_STAMPS=1
#ifndef DSM_WAVE_STAMPS
#define DSM_WAVE_STAMPS 0
#endif
constexpr bool kWaveStamps = DSM_WAVE_STAMPS != 0;

constexpr int kIntMax = 0x7fffffff;
typedef uint16_t label_t;             // a pixel's superpixel index in the label planes
constexpr int kNoLabel = 0xffff;      // the reference's label -1 (pixels no cell reaches) in a label plane
constexpr int kMaxSeeds = 0xffff;     // superpixels a frame can have (dsm_create checks): every index fits a label_t below kNoLabel

// Per-frame inputs that change from frame to frame.  A ring of these lives in HBM; kernels pick
// entry (cursor % ring) so a captured hipGraph can be replayed without touching its arguments.
struct FrameParams {
    float pose[16]; // cam -> world, column-major
    float inv[16];  // world -> cam (host-computed with inverse4<float>, FF.cpp:59)
    int32_t ref_idx;
    int32_t slot;
    int32_t pad[2];
};

// The frame being processed: written to device memory by k_init_seeds (first kernel of a frame) from
// params[cursor % n_params], read by every later kernel of the frame.
struct FrameCur {
    FrameParams p;
    const uint8_t *img;
    const float *dep;
};

// Hand-off from k_seed_stats to k_seed_fit: the start of get_huber_norm (FF.cpp:104-120) for one seed; the fit gathers the
// centred inlier points itself.  A superpixel's pixels lie within 8 of its centre in both axes (FF.cpp:413-422): at most
// 15 x 15 = 225 members.
struct GnHeader {
    int32_t m_in;      // inliers handed to the fit; 0 = no fit, the seed keeps its defaults (FF.cpp:841, 862)
    float nx, ny, nz;  // normalised sum of the inliers' pixel normals (FF.cpp:852-871)
    float mx, my, mz;  // centroid (FF.cpp:111-120)
    float far2;        // squared superpixel radius (FF.cpp:820-824)
};
constexpr int kGnCap = 232;
constexpr int kRestListCap = 127; // = kLaneCap of k_update_seeds: the longest list a lane keeps in LDS
constexpr int kFitSmallCap = 120; // longest list the short-column tier of k_seed_fit takes (batched launches)

// Everything a kernel needs.  Passed to every kernel BY VALUE (kernel-argument segment): the pointers and
// sizes never change after dsm_create, so a captured graph stays valid, and a kernel reaches its data
// without first loading a context from memory (one dependent round trip less per kernel).
struct DeviceCtx {
    // geometry
    int32_t w, h, pitch; // pitch = row stride in elements of every image-shaped plane (multiple of 64)
    int32_t gw, gh, n_seed;
    uint32_t gw_magic; // floor(2^32 / gw) + 1: s / gw == mulhi(s, gw_magic) for every seed index s < 65 536 (seed_cell)
    Intrinsics k;
    float far_d, near_d;
    double huber, baseline, disp_err, min_tol;
    const float *ray_x, *ray_y; // [w + 1], [h + 1]: (x - cx) / fx, (y - cy) / fy of back_project (dsm_math.h, ray_coeff)
    // frame slots (HBM-resident inputs)
    const uint8_t *img_base;
    const float *depth_base;
    int64_t slot_elems; // pitch * h
    int32_t n_slots;
    // superpixel state
    // 16 bits per pixel (label_t: a frame has at most kMaxSeeds = 65 535 superpixels, kNoLabel = the reference's -1): the label
    // image is read by every stage of a frame, by the window walks several times over -- a third of a frame's memory-side
    // traffic when it was 4 bytes per pixel
    label_t *label; // [h][pitch] superpixel index of every pixel: every sweep's image in turn (k_apply_labels works in place), then the final one
    label_t *cand;  // [h][pitch] seed picked by this sweep before the stable-skip rule is applied
    float4 *core;   // [S] x, y, mean_intensity, mean_depth  (live seed state during the sweeps)
    double *inv_depth; // [S] 1.0 / mean_depth, FF.cpp:380
    float4 *core_stage; // [S] update_seeds output before the chunk-commit rule
    int32_t *stable_stage;
    // tmin[s]: -1 = seed unstable when the sweep started; INT_MAX = stable and never picked;
    // otherwise the first pixel key (row-major) at which an evaluated pixel picked the seed.
    int32_t *tmin;
    int32_t *first_empty; // [kSweeps][kWorkers] first unstable seed without pixels, per worker chunk
    int32_t *worklist;    // pixel keys whose old and new seeds were both stable at sweep start
    int32_t *work_count;
    int32_t fit_small_cap;  // kFitSmallCap, or less (dsm_debug_set_fit_small_cap: lets a test push ordinary groups through the other tier)
    int32_t *fit_big_count; // groups of seeds queued in `worklist` for the full-length tier of k_seed_fit (batched launches)
    int32_t *rest_count;    // [kSweeps][2] entries k_update_seeds queued in `worklist` for k_update_seeds_rest (batched launches):
                            // seeds that need more Huber passes | seeds whose depth list outgrew its LDS row
    float *rest_list;       // [ceil(S / 64)][kRestListCap][64] depth lists of the queued seeds, entry-major within a group of 64
    GnHeader *gn_hdr; // [S]
    float4 *plane;    // [S] the plane k_seed_fit fitted (normal, offset: before plane_finish), for k_seed_finish
    float *normals;   // [h][pitch][3] forward-difference normal of every depth inlier of its own superpixel (k_pixel_normals); other entries stale, never read
    dsm_seed *seeds; // [S] final seed table, reference layout
    // what initialize_surfels (FF.cpp:315-361) would create from each seed, prepared by k_seed_planes (every
    // input but the `fused` flag is known there): the surfel, whether the seed qualifies, the flag itself
    dsm_surfel *spawn_rec; // [S]
    uint8_t *spawn_ok;     // [S]
    uint8_t *fused_flag;   // [S] set by k_fuse_surfels (besides the byte in `seeds`)
    float *seed_weight;    // [S] depth_weight(mean_depth) of the final seed, for k_fuse_surfels
    int32_t *spawn_idx;    // [S] seeds that do create a surfel, ascending
    // surfel map
    dsm_surfel *local;
    int32_t cap;
    dsm_surfel *fresh; // [S] surfels created by the current frame, seed order
    int32_t *n_local;
    int32_t *n_local_next;
    int32_t *n_new;
    // [cap/64 + 1] group g = records 64g .. 64g+63: a record of it was written since the flags were last cleared (k_fuse_surfels:
    // a wave stores its 64 records back only if one changed; k_frame_tail: refills, moves, appended surfels).  The drop-in call
    // downloads the flagged groups only (k_delta_pack clears them); flags left by other frames are cleared before it looks
    uint8_t *grp_dirty;
    uint64_t *hole_mask;  // [cap/64] bit i of word v: surfel 64v+i has update_times == 0
    int32_t *wave_prefix; // [cap/64] exclusive prefix of popcounts
    int32_t *holes;       // [cap] ascending indices of deleted slots
    int32_t *n_holes;
    // large maps (more than kTailFastWords * 64 surfels): holes per chunk of kTailChunkWords bitmap words, counted by
    // k_fuse_surfels, for the workgroups of k_frame_tail that list them; [n_hole_chunk] = their ticket counter,
    // [n_hole_chunk + 1] = the map size the frame started with.  All zero between frames (the tail's last workgroup resets them).
    int32_t *hole_chunk;
    int32_t n_hole_chunk;
    // sequencing
    const FrameParams *params;
    int32_t n_params;
    int32_t *cursor;      // frames this pipeline has finished; its next frame is params[cursor * cursor_mul + cursor_add]
    int32_t cursor_mul, cursor_add;
    int32_t *status; // sticky device-side error bits
    FrameCur *cur; // device memory, see FrameCur
    // optional per-wave phase stamps (shader clock) of the per-seed kernels; null unless DSM_FLAG_WAVE_STAMPS
    long long *stamps; // [5 kernels][n_seed][8]
};

// k_frame_tail: a thread owns kScanWords words of the hole bitmap per round, a workgroup kTailChunkWords; maps of up to
// kTailFastWords * 64 surfels take its one-workgroup fast path
constexpr int kScanWords = 8, kTailChunkWords = 1024 * kScanWords, kTailFastWords = 4096;
constexpr int kTailMaxBlocks = 33; // workgroups of k_frame_tail for a large map: one for the new surfels, the rest list holes

constexpr int kStatusCapacity = 1;
constexpr int kStatusBadPick = 2;
constexpr int kStatusBadLabels = 4; // a superpixel with more members than its 15 x 15 reach (injected label image)

// launch all kernels of one frame on `stream`.  with_compaction: SurfelMap::fuse_map semantics,
// otherwise FusionFunctions::fuse_initialize_map.  If ev != nullptr, an event is recorded before
// the first kernel and after every kernel (ev[0..n_stages]).
constexpr int kNumStages = 16;
constexpr int kLastSuperpixelStage = 13; // init_seeds .. seed_fit need the frame only; fuse_surfels + frame_tail need the map
extern const char *const kStageNames[kNumStages];
// map_upper_bound sizes the grid-stride pass of k_fuse_surfels (any bound does, the loop covers the map); tail_map_bound
// must be an upper bound of the map size whenever that exceeds kTailFastWords * 64 (it decides whether k_frame_tail gets
// the workgroups that list a large map's holes), and may be 0 while the map is known to be smaller.
// lanes_from: frames per launch from which the per-seed stages take their lane-per-seed forms (0 = the default for independent
// handles, kLaneBatch; the frame groups of one sequence pass 4: with other groups' kernels sharing the GPU the lane forms win
// from four frames on -- profiles/r06_wave_vs_lane.md)
hipError_t launch_frame(const DeviceCtx &ctx, int map_upper_bound, int tail_map_bound, bool with_compaction,
                        hipStream_t stream, hipEvent_t *ev, int stage_lo = 0, int stage_hi = kNumStages - 1,
                        const DeviceCtx *d_batch = nullptr, int n_batch = 1, int lanes_from = 0);

struct WarpMat {
    float m[16]; // column-major 4x4
};
// d_mats: per-group matrices in device memory, or nullptr: `single16` (host pointer, copied into the kernel arguments)
hipError_t launch_warp(dsm_surfel *surfels, const int32_t *n_ptr, int n_fixed, const float *d_mats, const float *single16,
                       const int32_t *d_offsets, int n_groups, int n_upper, hipStream_t st,
                       const uint8_t *d_group_on = nullptr, float4 *d_cloud = nullptr);
hipError_t launch_mark(const DeviceCtx &ctx, int key, int n_upper, hipStream_t st);
hipError_t launch_repack_frames(uint8_t *d_img, float *d_depth, int pitch, int64_t slot_elems, const uint8_t *s_img, const float *s_depth, int w, int h,
                                int frames, hipStream_t st);
hipError_t launch_repack(uint8_t *d_img, float *d_depth, int pitch, const uint8_t *s_img, const float *s_depth, int w, int h,
                         hipStream_t st);
hipError_t launch_extract_marked(const DeviceCtx &ctx, dsm_surfel *out, int cap, float4 *cloud_out, hipStream_t st);
hipError_t launch_extract(const DeviceCtx &ctx, int key, dsm_surfel *out, int cap, int n_upper, hipStream_t st);
hipError_t launch_append_count(const DeviceCtx &ctx, int n, hipStream_t st);
// the flagged 64-record groups of the map (DeviceCtx::grp_dirty) packed into `buf` ([cap_groups][64] records) with their
// group numbers in `idx`; count[0] = how many there are (may exceed cap_groups: those are not packed), flags cleared
hipError_t launch_delta_pack(const DeviceCtx &ctx, dsm_surfel *buf, int32_t *idx, int32_t *count, int cap_groups, int n_upper, hipStream_t st);

} // namespace dsm
