/* dsm.h -- C ABI of the MI355X-native surfel-fusion hot path.
 *
 * This is the boundary a maintainer of HKUST-Aerial-Robotics/DenseSurfelMapping binds
 * instead of the in-process CPU engine.  Reference interfaces replaced (paths relative
 * to the reference checkout):
 *
 *   dsm_create / dsm_destroy      <- FusionFunctions::initialize
 *                                    surfel_fusion/src/fusion_functions.h:84-87, .cpp:7-28
 *   dsm_fuse_initialize_map       <- FusionFunctions::fuse_initialize_map
 *                                    surfel_fusion/src/fusion_functions.h:88-94, .cpp:30-83
 *   dsm_fuse_map                  <- SurfelMap::fuse_map (engine call + hole refill /
 *                                    swap-with-last compaction)
 *                                    surfel_fusion/src/surfel_map.cpp:1060-1113
 *   dsm_surfel / dsm_seed         <- SurfelElement / Superpixel_seed, bit-for-bit
 *                                    surfel_fusion/src/elements.h:5-31
 *
 * Everything else (resident map, frame slots, replay queue, taps) is what an HBM-resident
 * engine needs and the CPU reference has no counterpart for.
 *
 * Conventions: plain pointers and sizes, no exceptions; every call returns DSM_OK (0) or a
 * negative dsm_status; dsm_last_error() gives the message.  A handle owns its device buffers and, with
 * pipeline_depth < 4, all of its streams (from 4 on it also launches on the device's shared per-queue streams,
 * see dsm_config.pipeline_depth) and is single-threaded-use (like a FusionFunctions instance,
 * whose scratch buffers are members: fusion_functions.h:34-37); use one handle per
 * concurrent subsequence.  There is no CPU fallback: dsm_create fails with
 * DSM_E_NO_DEVICE when no gfx950 device is visible.
 */
#ifndef DSM_H
#define DSM_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSM_ABI_VERSION 4 /* 3: the *_inv entry points (the caller's own pose inverse).  4: a batch no longer touches its handles'
                             streams at every call -- a handle's stream comes behind its batch at the handle's NEXT dsm_* call
                             (dsm_stream's note): a caller that cached the hipStream_t must fetch it again after batch calls;
                             dsm_create refuses DSM_FLAG_WAVE_STAMPS in a library built without stamp code; dsm_debug_tier_counts */

typedef enum {
    DSM_OK = 0,
    DSM_E_INVALID = -1,   /* bad argument (null pointer, dims, steps, slot index) */
    DSM_E_NO_DEVICE = -2, /* no HIP device / not gfx950 */
    DSM_E_HIP = -3,       /* a HIP runtime call failed; see dsm_last_error */
    DSM_E_CAPACITY = -4,  /* surfel capacity exceeded */
    DSM_E_STATE = -5      /* call made in the wrong state (e.g. nothing uploaded) */
} dsm_status;

/* elements.h:22-31 -- 44 bytes.  update_times == 0 marks a deleted slot. */
typedef struct dsm_surfel {
    float px, py, pz;
    float nx, ny, nz;
    float size;
    float color;
    float weight;
    int32_t update_times;
    int32_t last_update;
} dsm_surfel;

/* elements.h:5-20 -- 60 bytes (bool fused @48, bool stable @49, 2 pad bytes). */
typedef struct dsm_seed {
    float x, y;
    float size;
    float norm_x, norm_y, norm_z;
    float posi_x, posi_y, posi_z;
    float view_cos;
    float mean_depth;
    float mean_intensity;
    uint8_t fused;
    uint8_t stable;
    uint8_t pad_[2];
    float min_eigen_value, max_eigen_value;
} dsm_seed;

/* Constructor arguments of FusionFunctions::initialize plus the compile-time constant set of
 * fusion_functions.h:7-21 made run-time (the RGB-D set of lines 17-21 is needed for
 * 640x480 RGB-D input). */
typedef struct dsm_config {
    int32_t width, height;
    float fx, fy, cx, cy;
    float far_dist, near_dist;
    double huber_range;       /* HUBER_RANGE        0.4  (rgbd 0.05) */
    double baseline;          /* BASELINE           0.5  (rgbd 0.08) */
    double disparity_error;   /* DISPARITY_ERROR    4.0  (rgbd 1.0)  */
    double min_tolerate_diff; /* MIN_TOLERATE_DIFF  0.1  (rgbd 0.05) */
    int32_t device;           /* HIP device ordinal */
    int32_t surfel_capacity;  /* max resident surfels; 0 = default (4 Mi) */
    int32_t frame_slots;      /* resident frame slots in HBM; 0 = default (2) */
    uint32_t flags;           /* DSM_FLAG_* */
    int32_t pipeline_depth;   /* frames of one sequence whose superpixel stages may be in flight at once
                                 (1, 2, 4, 8, 12, 16, 24 or 32); 0 = default (4).  Results do not depend on it.  From 4 on,
                                 dsm_replay_enqueue launches the superpixel stages of G consecutive frames as one batch
                                 (every kernel once for all of them) while fuse + compaction of the previous group run
                                 in frame order: G = depth / 2 (two groups in turn) for 4 and 8, depth / 4 (four groups)
                                 for 16 and 32, depth / 3 (three groups) for 12 and 24.  From depth 4 on the streams the
                                 groups are launched on -- and with 12 / 24 the map stream too -- are the device's four
                                 process-wide per-queue streams, shared with batches and other handles: they carry
                                 launches only (graphs are captured on private streams), the handle's buffers stay its
                                 own.  24 is the fastest single-sequence setting on MI355X. */
} dsm_config;

#define DSM_FLAG_NO_GRAPH 1u /* launch kernels eagerly instead of replaying a hipGraph */
#define DSM_FLAG_UPLOAD_STREAM 2u /* dsm_frame_upload (the blocking call) runs on a stream of its own, overlapping the
                                     frames in flight on other slots (live feeds: the node).  Off by default: HIP spreads
                                     streams over 4 hardware queues in creation order, and one more stream per handle
                                     spreads the map streams of many handles unevenly (8-subsequence replay: -22 %).
                                     Handles of a batch cannot have it; replays that stream their frames -- batched or
                                     not -- use dsm_frames_upload_async, which needs no flag and no stream per handle. */

#define DSM_FLAG_WAVE_STAMPS 4u /* debug: allocate the per-wave phase-stamp buffer read by dsm_debug_wave_stamps -- in a library
                                   built with -DDSM_WAVE_STAMPS=1 (tools/wave_stamps.py); the shipped build has no stamp code in
                                   its kernels and dsm_create REFUSES the flag (DSM_E_INVALID): a profiling run on the wrong
                                   library fails at once instead of reading an empty buffer later */

typedef struct dsm_handle dsm_handle;

/* Fill cfg with the driving constant set (fusion_functions.h:13-16) or, if rgbd != 0, the
 * RGB-D set (fusion_functions.h:17-21). */
int dsm_config_init(dsm_config *cfg, int width, int height, float fx, float fy, float cx, float cy,
                    float far_dist, float near_dist, int rgbd);

/* (width / 8) * (height / 8) <= 65 535 superpixels per frame (DSM_E_INVALID beyond: 4K frames are out of range, 1920x1080
 * has 32 400): superpixel indices are 16 bits in the device's label planes; the taps below speak int32, -1 = no superpixel. */
int dsm_create(const dsm_config *cfg, dsm_handle **out);
void dsm_destroy(dsm_handle *h);
const char *dsm_last_error(const dsm_handle *h); /* h may be NULL: last dsm_create error */
int dsm_abi_version(void);

/* ---- drop-in calls (host buffers in, host buffers out; synchronous) -------------------- */

/* FusionFunctions::fuse_initialize_map: image is 8-bit grey (img_step bytes per row), depth is
 * float metres with 0 = invalid (depth_step bytes per row), pose16 is the cam->world matrix,
 * column-major (Eigen::Matrix4f storage).  local[0..n_local) is updated in place; surfels
 * created by this frame are written to new_out[0..*n_new) in seed order. */
int dsm_fuse_initialize_map(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step,
                            const float *depth, size_t depth_step, const float *pose16, dsm_surfel *local,
                            int32_t n_local, dsm_surfel *new_out, int32_t new_cap, int32_t *n_new);

/* SurfelMap::fuse_map: the call above followed by the reference's order-exact refill of deleted
 * slots and swap-with-last compaction.  *n_local is updated; cap is the capacity of local[]. */
int dsm_fuse_map(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step,
                 const float *depth, size_t depth_step, const float *pose16, dsm_surfel *local,
                 int32_t *n_local, int32_t cap, int32_t *n_new);

/* The same two calls with the world->cam matrix handed in.  FusionFunctions::fuse_initialize_map inverts the pose with the
 * caller's matrix library (`pose.inverse()`, Eigen::Matrix4f, fusion_functions.cpp:59); Eigen is not part of this library,
 * which uses the adjugate / determinant closed form instead -- equal to Eigen's result up to the last place, and a last
 * place of this matrix can flip a create / delete decision downstream.  A caller that has the
 * reference's matrix type computes inv_pose16 = pose.inverse() itself (include/dsm_fusion_functions.hpp does) and gets the
 * arithmetic of ITS Eigen build bit for bit; NULL = the closed form. */
int dsm_fuse_initialize_map_inv(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step,
                                const float *depth, size_t depth_step, const float *pose16, const float *inv_pose16,
                                dsm_surfel *local, int32_t n_local, dsm_surfel *new_out, int32_t new_cap, int32_t *n_new);
int dsm_fuse_map_inv(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step,
                     const float *depth, size_t depth_step, const float *pose16, const float *inv_pose16,
                     dsm_surfel *local, int32_t *n_local, int32_t cap, int32_t *n_new);

/* Page-locked host memory for frames handed to dsm_frame_upload / dsm_fuse_*: uploads from it run at PCIe
 * rate without a staging copy.  Plain malloc'ed buffers work too, slower. */
int dsm_host_alloc(void **out, size_t bytes);
void dsm_host_free(void *p);
/* n of the caller's frames (frame i: images[i] / depths[i] with their row steps in bytes -- n cv::Mat pairs as
 * SurfelMap::image_input / depth_input receive them, surfel_map.cpp:83-101) copied into page-locked memory in the layout the
 * asynchronous uploads read (dsm_frames_upload_async, dsm_replay_enqueue_host: rows dst_*_step bytes apart, frames
 * dst_*_frame_step apart) by the library's host threads, the caller's among them; pad bytes are left as they are.  Synchronous;
 * calls from several threads take turns. */
int dsm_host_pack_frames(int32_t n, int32_t width, int32_t height, const uint8_t *const *images, const size_t *image_steps,
                         const float *const *depths, const size_t *depth_steps, uint8_t *dst_image, size_t dst_img_step,
                         size_t dst_img_frame_step, float *dst_depth, size_t dst_depth_step, size_t dst_depth_frame_step);

/* ---- resident path: map and frames stay in HBM, calls are asynchronous on the handle's stream */

int dsm_map_upload(dsm_handle *h, const dsm_surfel *surfels, int32_t n);
int dsm_map_size(dsm_handle *h, int32_t *n);                       /* synchronises */
int dsm_map_capacity(const dsm_handle *h, int32_t *cap);           /* slots of the resident map (surfel_capacity rounded up) */
int dsm_map_download(dsm_handle *h, dsm_surfel *out, int32_t cap, int32_t *n); /* synchronises */
/* device-to-device copy of the resident map into caller-owned device memory (e.g. a torch
 * tensor's data_ptr) for the multi-GPU merge; synchronises. */
int dsm_map_copy_to_device(dsm_handle *h, void *dst_device, int32_t cap, int32_t *n);

/* ---- map maintenance between frames (the data-parallel parts of surfel_map.cpp:681-824, 1456-1595) ---- */

/* SurfelMap::warp_active_surfels_cpu_kernel (surfel_map.cpp:750-789): p' = M*(p,1), n' = M[0:3,0:3]*n for
 * every resident surfel; warp16 = (loop_pose * cam_pose^-1).cast<float>(), column-major (surfel_map.cpp:808-813).
 * Asynchronous on the handle's stream. */
int dsm_map_warp(dsm_handle *h, const float *warp16);
/* SurfelMap::warp_inactive_surfels_cpu_kernel (surfel_map.cpp:681-748) on caller-owned DEVICE memory: the
 * attached surfels of n_groups keyframes stored back to back, group g = [offsets[g], offsets[g+1]) warped by
 * mats16[16 g ..].  offsets (n_groups+1) and mats16 are host arrays.  Synchronises. */
int dsm_warp_grouped_device(dsm_handle *h, void *surfels_device, int32_t n_groups, const int32_t *offsets,
                            const float *mats16);
/* move_add_surfels, removal half (surfel_map.cpp:1476-1497): copy the live resident surfels whose
 * last_update == key to out[] in index order and mark their slots deleted.  Synchronises. */
int dsm_map_extract(dsm_handle *h, int32_t key, dsm_surfel *out, int32_t cap, int32_t *n);
/* move_add_surfels, insertion half (surfel_map.cpp:1583-1590): append to the resident map. */
int dsm_map_append(dsm_handle *h, const dsm_surfel *surfels, int32_t n);

/* ---- inactive store: the surfels of keyframes that left the local window stay in HBM, back to back in
 * deactivation order, each with the XYZI point the reference keeps in `inactive_pointcloud`
 * (surfel_map.cpp:1476-1497 fills poses_database[i].attached_surfels and inactive_pointcloud; here both are
 * one device arena indexed like inactive_pointcloud).  The segment table (which keyframe owns [begin,
 * begin+n)) is host state of the caller -- include/dsm_surfel_map.h keeps it as the reference does
 * (points_begin_index / points_pose_index / pointcloud_pose_index). ---- */

/* surfel_map.cpp:1476-1497: move the live resident surfels with last_update == key, in index order, to the
 * end of the store (slots marked deleted in the map).  Returns the segment.  Synchronises. */
int dsm_store_deactivate(dsm_handle *h, int32_t key, int32_t *begin, int32_t *n);
/* surfel_map.cpp:1583-1590: append store[begin, begin+n) to the resident map (the store is not changed). */
int dsm_store_activate(dsm_handle *h, int32_t begin, int32_t n);
/* surfel_map.cpp:1551-1553 (`inactive_pointcloud->erase`): remove [begin, begin+n); the tail moves down. */
int dsm_store_erase(dsm_handle *h, int32_t begin, int32_t n);
/* surfel_map.cpp:681-748 for every keyframe at once: offsets[n_groups+1] tile the store, group g is warped
 * by mats16[16 g ..] (column-major float) when changed[g] != 0 and left alone otherwise; the XYZI shadow is
 * refreshed for all points of a warped group except its last (surfel_map.cpp:742).  Synchronises. */
int dsm_store_warp(dsm_handle *h, int32_t n_groups, const int32_t *offsets, const float *mats16,
                   const uint8_t *changed);
int dsm_store_size(dsm_handle *h, int32_t *n);
/* either output may be NULL; xyzi_out receives 4 floats per point (x, y, z, intensity).  Synchronises. */
int dsm_store_download(dsm_handle *h, int32_t begin, int32_t n, dsm_surfel *surfels_out, float *xyzi_out);

/* Copy a frame into frame slot `slot` (0 .. frame_slots-1 of the config) and return when it is there (the host
 * buffers may be reused).  By default the copy is ordered behind everything enqueued so far.  With
 * DSM_FLAG_UPLOAD_STREAM it runs on the handle's upload stream instead: it waits only for the enqueued frames that
 * still read this slot and overlaps frames in flight on OTHER slots -- alternate two slots to upload frame t+1
 * while frame t is fused. */
int dsm_frame_upload(dsm_handle *h, int slot, const uint8_t *image, size_t img_step, const float *depth,
                     size_t depth_step);
/* same, sources already in device memory (the caller makes sure they have been written) */
int dsm_frame_upload_device(dsm_handle *h, int slot, const void *image_dev, size_t img_step,
                            const void *depth_dev, size_t depth_step);

/* ---- streamed input: frames of a replay arrive from host memory while earlier frames are being fused (the reference
 * receives every frame through image_input / depth_input, surfel_map.cpp:83-101).  dsm_frame_upload_async returns at once:
 * the copy runs on the device's upload stream, ordered behind the frames enqueued so far that may still read the slots it
 * writes -- the newest dsm_replay_enqueue call that reads one of them (the last four calls are kept apart by the slots they
 * read); behind EVERYTHING enqueued so far for the handle if frames were enqueued any other way (frame by frame, through a
 * batch) since its last upload -- and a frame enqueued AFTER the call, alone or through a batch, waits for it if it reads
 * one of the slots it writes (the last four uploads of a handle are kept apart by slot range; older ones count as one).
 * So a replay buffers in chunks: upload chunk k+1, THEN enqueue chunk k -- chunk k waits for upload k only, and upload k+1
 * runs beside its kernels.  With two groups of slots upload k+1 waits for chunk k-1, whose slots it overwrites; with three
 * in turn it waits for chunk k-2, which finished long ago, and never holds up the hardware queue it shares with the handle's
 * own streams (densesurfelmapping_amd/replay.py: 13-14 k instead of 11 k frames/s for one sequence at 1226x370).  The source must be page-locked (dsm_host_alloc) and stay untouched until
 * dsm_frame_uploads_wait (or dsm_synchronize after a frame that reads the slot).  Rows laid out with the slot pitch
 * (dsm_frame_pitch elements per row: img_step = pitch, depth_step = 4 * pitch; the pad columns are never read) go up
 * as one transfer per plane, any other step row by row. ---- */
int dsm_frame_pitch(const dsm_handle *h, int32_t *pitch);
int dsm_frame_upload_async(dsm_handle *h, int slot, const uint8_t *image, size_t img_step, const float *depth,
                           size_t depth_step);
/* n frames into slots slot0 .. slot0+n-1: frame i at image + i * img_frame_step / depth + i * depth_frame_step (bytes).
 * Frames laid out back to back exactly like the slots (row steps = the pitch, frame steps = pitch * height elements) go
 * up as ONE transfer per plane for all n -- a replay that keeps its frames in that layout pays two transfers per chunk
 * instead of two per frame.  TIGHT rows (row steps = width elements, frames back to back) are taken too: one transfer per
 * plane into a staging buffer of the handle (allocated at the first such call: that call waits for the upload stream) and
 * a kernel that sets the rows to the slots' pitch -- 4.4 % fewer bytes over the link at 1226 pixels, yet measured SLOWER than
 * the pitched layout (the kernel sits between two transfers of its stream); any other layout goes row by row. */
int dsm_frames_upload_async(dsm_handle *h, int slot0, int n, const uint8_t *image, size_t img_step, size_t img_frame_step,
                            const float *depth, size_t depth_step, size_t depth_frame_step);
int dsm_frame_uploads_wait(dsm_handle *h); /* blocks the host until this handle's asynchronous uploads have landed */

/* enqueue SurfelMap::fuse_map for the frame in `slot` against the resident map */
int dsm_fuse_frame_resident(dsm_handle *h, int slot, int reference_frame_index, const float *pose16);
/* enqueue n frames: frame i uses slots[i], ref_idx[i], poses16[16*i .. 16*i+16) */
int dsm_replay_enqueue(dsm_handle *h, int32_t n, const int32_t *slots, const int32_t *ref_idx,
                       const float *poses16);
/* the same with the caller's own pose inverses (see dsm_fuse_map_inv); inv_pose(s)16 may be NULL */
int dsm_fuse_frame_resident_inv(dsm_handle *h, int slot, int reference_frame_index, const float *pose16,
                                const float *inv_pose16);
int dsm_replay_enqueue_inv(dsm_handle *h, int32_t n, const int32_t *slots, const int32_t *ref_idx,
                           const float *poses16, const float *inv_poses16);
/* The same with the frames COMING WITH THE CALL (round 6): n frames in page-locked host memory (dsm_host_alloc), frame i at
 * image + i * img_frame_step / depth + i * depth_frame_step, rows img_step / depth_step bytes apart (rows at the frame slots' own
 * pitch -- dsm_frame_pitch elements -- and frames one slot apart go up as one transfer per plane and group of frames).  What a
 * replay of a log does (the reference receives every frame through image_input / depth_input, surfel_map.cpp:83-101).  Each
 * group of frames is uploaded ON THE STREAM THAT RUNS ITS SUPERPIXEL STAGES, right in front of them, into the frame slots of the
 * pipelines that take it (frame f -> slot f mod pipeline_depth: the handle needs frame_slots >= pipeline_depth, and whatever
 * those slots held is overwritten): no upload stream and no event between a transfer and its consumer, the transfer of one
 * group runs beside the kernels of the groups before it, and nothing is waited for on the host.  The host memory of a call
 * may be rewritten once dsm_replay_wait says the call's frames are done. */
int dsm_replay_enqueue_host(dsm_handle *h, int32_t n, const uint8_t *image, size_t img_step, size_t img_frame_step, const float *depth,
                            size_t depth_step, size_t depth_frame_step, const int32_t *ref_idx, const float *poses16,
                            const float *inv_poses16 /* may be NULL */);
/* host wait until the frames of the dsm_replay_enqueue_host call `calls_back` calls ago (0 = the latest; < 8) have been fused */
int dsm_replay_wait(dsm_handle *h, int32_t calls_back);
int dsm_synchronize(dsm_handle *h);
/* number of new surfels created by the last completed frame; synchronises */
int dsm_last_new_count(dsm_handle *h, int32_t *n_new);
/* the handle's hipStream_t, for event timing by the caller.  (A handle that advances with a batch: its stream is put behind
 * the batch's work at the handle's next dsm_* call -- this one included -- not after every batch call; fetch the stream after
 * the batch calls whose results the caller's own work on it should come behind.) */
int dsm_stream(dsm_handle *h, void **hip_stream);

/* ---- batches: handles of equal image size on one device, each with its own map and frames (independent subsequences,
 * surfel_map.cpp has no counterpart: it fuses one frame at a time), advancing in LOCKSTEP: every kernel of a frame is
 * launched once for the whole batch (grid z = handle).  Launch overheads, cold caches and the slowest waves of a kernel
 * are shared by all subsequences instead of paid by each: this is what bench.py's headline replays.  Handles must have
 * pipeline_depth 1 and a resident map; they stay usable on their own between batch calls (a handle's stream is ordered
 * behind the batch when it is next used, by every per-handle call and by dsm_batch_synchronize; a batch call itself touches
 * no stream but the batch's own -- the handles' streams share hardware queues with the other batches). ---- */
typedef struct dsm_batch dsm_batch;
int dsm_batch_create(dsm_handle *const *handles, int32_t n, dsm_batch **out);
void dsm_batch_destroy(dsm_batch *b);
const char *dsm_batch_last_error(const dsm_batch *b);
/* enqueue n_frames frames for every handle: frame i of handle j uses slots[j * n_frames + i], ref_idx[...], and
 * poses16[(j * n_frames + i) * 16 ..] */
int dsm_batch_replay_enqueue(dsm_batch *b, int32_t n_frames, const int32_t *slots, const int32_t *ref_idx, const float *poses16);
int dsm_batch_replay_enqueue_inv(dsm_batch *b, int32_t n_frames, const int32_t *slots, const int32_t *ref_idx,
                                 const float *poses16, const float *inv_poses16); /* inv_poses16 laid out like poses16; may be NULL */
/* wait for everything enqueued and report the first handle's error, if any */
int dsm_batch_synchronize(dsm_batch *b);

/* ---- parity taps (state after the last completed frame; synchronise) ------------------- */
int dsm_get_labels(dsm_handle *h, int32_t *out /* height*width */);
int dsm_get_seeds(dsm_handle *h, dsm_seed *out /* (width/8)*(height/8) */);
int dsm_seed_count(const dsm_handle *h);

/* ---- state-level test taps (poke superpixel state, run single stages) ------------------
 * Stage indices are positions in the frame's kernel sequence: 0 init_seeds, 1 assign_0, 2 update_seeds_0,
 * 3 commit_seeds_0, 4 assign_1, 5 resolve_1, 6 update_seeds_1, 7 commit_seeds_1, 8 assign_2, 9 resolve_2,
 * 10 update_seeds_2, 11 commit_seeds_2, 12 seed_points, 13 seed_fit, 14 fuse_surfels, 15 frame_tail.
 * ABI 3 notes for users of these taps:
 *  - there is ONE label image, updated in place from sweep to sweep; `which` = 0 and 1 both name it (the sweeps used to take
 *    turns between two buffers, and old callers pass `sweep & 1`);
 *  - a sweep >= 1 is the PAIR assign_k + resolve_k (stages 4-5 and 8-9): assign leaves its picks in a side plane and
 *    resolve rewrites the label image from them.  Re-running assign_k alone after its resolve_k reads the already updated
 *    image and does not reproduce the sweep: inject the pre-sweep image (dsm_debug_set_label_buffer) and run the pair;
 *  - dsm_debug_set_label_buffer checks the image: every entry a superpixel index of this grid, and -1 exactly at the pixels
 *    beyond every cell's reach (image sizes with (size mod 8) > 4), where the assignment stage itself writes -1 --
 *    DSM_E_INVALID otherwise (the kernels index per-seed arrays with the labels they read). */
int dsm_debug_run_stages(dsm_handle *h, int slot, int reference_frame_index, const float *pose16, int first_stage,
                         int last_stage);
int dsm_debug_get_label_buffer(dsm_handle *h, int which, int32_t *out);
int dsm_debug_set_label_buffer(dsm_handle *h, int which, const int32_t *in);
/* live seed state during the sweeps: core = (x, y, mean_intensity, mean_depth) per seed, stable = 0/1 */
int dsm_debug_get_seed_state(dsm_handle *h, float *core4, int32_t *stable);
int dsm_debug_set_seed_state(dsm_handle *h, const float *core4, const int32_t *stable);

/* debug tap: shader-clock stamps of the phases of the per-seed kernels, [5][n_seed][8] (kernel 0..2 =
 * update_seeds of sweep 0..2, 3 = seed_points, 4 = seed_fit at the first seed of each group of four); needs DSM_FLAG_WAVE_STAMPS in dsm_config.flags */
int dsm_debug_wave_stamps(dsm_handle *h, int64_t *out);

/* test knob: longest point list (<= 120) the short-column tier of the batched plane fit takes; fresh handles only (before the
 * first frame, before dsm_batch_create) */
int dsm_debug_set_fit_small_cap(dsm_handle *h, int32_t cap);

/* debug tap: occupancy of the second tiers of the lane-per-seed kernels (launches batched over >= 8 frames) in the LATEST
 * frame of this handle: out[2 s + 0] = superpixels whose robust mean depth (FF.cpp:530-556) needed more than one Huber pass in
 * sweep s, out[2 s + 1] = superpixels whose depth list outgrew its 127-entry LDS row in sweep s, out[6] = groups of four
 * superpixels the plane fit (FF.cpp:128-180) took in its full-length tier, out[7] = 0 (reserved).  All zero after a frame that
 * ran the wave-per-seed kernels.  Synchronises. */
int dsm_debug_tier_counts(dsm_handle *h, int32_t *out /* 8 */);

/* debug tap: the drop-in calls (dsm_fuse_map / dsm_fuse_initialize_map) bring back only the 64-record groups of the map a frame
 * changed: out[0] = drop-in calls so far, out[1] = of them, calls that took the delta path (the others changed most of the map
 * and took one full download), out[2] = groups those brought back in all, out[3] = groups of the latest call; out[4..7] = host
 * microseconds spent so far in dsm_fuse_map calls on: staging the frame and launching its superpixel stages | comparing the
 * caller's array with the shadow (and uploading it if it differs) | waiting for the GPU | fetching and patching what changed. */
int dsm_debug_dropin_stats(dsm_handle *h, int64_t *out /* 8 */);

/* ---- per-kernel timing (hip events on the handle's stream) ----------------------------- */
#define DSM_MAX_STAGES 32
typedef struct dsm_stage_times {
    int32_t n_stages;
    const char *name[DSM_MAX_STAGES];
    double ms[DSM_MAX_STAGES];      /* accumulated kernel time per stage */
    int64_t launches[DSM_MAX_STAGES];
    int64_t frames;
    double event_overhead_ms;       /* accumulated length of one empty event-to-event interval per frame */
    int64_t sum_new;                /* surfels created, summed over the frames (K of SURVEY.md's B_alg) */
    int64_t sum_local;              /* live map size after each frame, summed over the frames (M) */
} dsm_stage_times;
/* run the resident fuse for n frames eagerly with an event pair around every kernel and
 * accumulate per-stage durations */
int dsm_replay_timed(dsm_handle *h, int32_t n, const int32_t *slots, const int32_t *ref_idx,
                     const float *poses16, dsm_stage_times *out);
/* the same for a batch (arguments as dsm_batch_replay_enqueue): durations are those of the batched launches; `frames`,
 * sum_new and sum_local count handle-frames */
int dsm_batch_replay_timed(dsm_batch *b, int32_t n_frames, const int32_t *slots, const int32_t *ref_idx,
                           const float *poses16, dsm_stage_times *out);

#ifdef __cplusplus
}
#endif
#endif /* DSM_H */
