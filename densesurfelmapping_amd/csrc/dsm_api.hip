// dsm_api.hip -- the C ABI of include/dsm.h: handle, HBM buffers, stream, hipGraph replay.
//
// Host-side arithmetic is limited to the 4x4 pose inverse (FF.cpp:59, fp32), done with the same
// inverse4<float> the oracle uses; everything else runs in dsm_kernels.hip.  There is no CPU path.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <new>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <chrono>
#include <unordered_map>
#include <vector>

#include "dsm_device.h"

using namespace dsm;

namespace {

thread_local std::string g_create_error;

constexpr int kParamRing = 4096;
constexpr int kMaxPipes = 32;
constexpr uint64_t kSerialBit = 1ull << 63;
constexpr int kDefaultCapacity = 4 * 1024 * 1024;

constexpr int kBatchStreams = 4;
hipError_t batch_streams_reserve(int device); // see BatchStreamPool
hipStream_t batch_stream_at(int device, int i);
constexpr int kUploadStreams = 2;
hipStream_t device_upload_stream(int device, int *which, bool high_priority); // asynchronous frame uploads of the handles on the device (dsm_frame_upload_async)
constexpr uint64_t kBatchBit = 1ull << 62;    // UpEntry::pending: a batch stream has not waited for this upload of the handle yet

} // namespace

// the handles alive in this process: dsm_batch_destroy gives a handle back its freedom only if it still exists
static std::mutex g_live_mu;
static std::unordered_map<dsm_handle *, uint64_t> g_live; // live handles -> their generation (an address can be handed out again)
static uint64_t g_next_generation = 1;

// A few host threads for the drop-in calls' bulk copies (frame rows into page-locked staging, the caller's surfel
// array against / into / out of its page-locked shadow): one core moves ~15 GB/s, the copies of a 100 k-surfel map
// would otherwise be most of a drop-in frame.
class HostPool {
  public:
    explicit HostPool(int n_workers) {
        for (int i = 0; i < n_workers; i++) workers_.emplace_back([this] { loop(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread &t : workers_) t.join();
    }
    // fn(i) for i in [0, n): the caller takes part.  A job lives on the caller's stack; workers pin it (users) under
    // the lock before touching it and the caller leaves only when every task is done and nobody holds the job any more,
    // so a worker that wakes late finds either the live job or none.
    void run(int n, const std::function<void(int)> &fn) {
        if (n <= 1 || workers_.empty()) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        Job job;
        job.fn = &fn;
        job.n = n;
        {
            std::lock_guard<std::mutex> lk(mu_);
            job_ = &job;
            gen_++;
        }
        cv_.notify_all();
        help(job);
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return job.done.load() == n && job.users == 0; });
        job_ = nullptr;
    }

  private:
    struct Job {
        const std::function<void(int)> *fn = nullptr;
        int n = 0;
        std::atomic<int> next{0}, done{0};
        int users = 0; // guarded by mu_
    };
    void help(Job &job) {
        for (;;) {
            const int i = job.next.fetch_add(1);
            if (i >= job.n) return;
            (*job.fn)(i);
            if (job.done.fetch_add(1) + 1 == job.n) {
                std::lock_guard<std::mutex> lk(mu_);
                done_cv_.notify_all();
            }
        }
    }
    void loop() {
        unsigned seen = 0;
        for (;;) {
            Job *job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                job = job_;
                if (job) job->users++;
            }
            if (!job) continue;
            help(*job);
            std::lock_guard<std::mutex> lk(mu_);
            if (--job->users == 0) done_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    Job *job_ = nullptr;
    unsigned gen_ = 0;
    bool stop_ = false;
};

struct dsm_handle {
    uint64_t generation = 0; // unique per dsm_create: a batch remembers it, so that a new handle at a destroyed handle's address is not taken for the old one
    dsm_config cfg;
    int device = 0;
    hipStream_t stream = nullptr; // the map stream: fuse + tail of every frame, in frame order; uploads; params
    bool own_stream = true;
    DeviceCtx hc;              // context of the pipeline that handled the latest frame (taps, shared pointers)
    // Superpixel stages (init_seeds .. seed_planes) depend on the frame only, so frame f runs them on pipeline
    // f % n_pipe -- its own stream and its own superpixel buffers -- while the map stream still fuses
    // earlier frames.  fuse_surfels + frame_tail of frame f wait for them and run in frame order.
    struct Pipe {
        hipStream_t stream = nullptr;
        bool own_stream = true; // false: one of the device's reserved per-queue streams (frame-group leads)
        DeviceCtx ctx;
        hipGraphExec_t g_sp = nullptr, g_map[2] = {nullptr, nullptr}, g_all[2] = {nullptr, nullptr};
        hipGraphExec_t g_sp_main = nullptr; // superpixel stages on the map stream (drop-in calls)
        hipEvent_t ev_sp = nullptr, ev_map = nullptr;
        hipEvent_t ev_free = nullptr; // what says this pipeline's buffers are free again: its own ev_map, or the one recorded for its whole frame group
    } pipe[kMaxPipes];
    int n_pipe = 1;
    // frame groups (pipeline_depth >= 4): the superpixel stages of n_pipe / 2 consecutive frames as ONE batched launch
    // per kernel over that half of the pipelines (submit_group)
    DeviceCtx *d_pipe_ctxs = nullptr;                 // [n_pipe] the pipelines' contexts, for the batched kernels
    hipGraphExec_t g_group[4] = {nullptr, nullptr, nullptr, nullptr}; // superpixel stages of pipelines [kG, (k+1)G)
    hipGraphExec_t g_group_map[4] = {nullptr, nullptr, nullptr, nullptr}; // their fuse + tail stages, frame after frame
    // ... and the same with k_frame_tail's workgroups for a large map, captured beside the first (while the replay warms up) if
    // the handle's capacity allows the map to get there: map_grows then only swaps them in (a capture in the middle of a
    // replay is 50 ms of host time during which the GPU runs dry)
    hipGraphExec_t g_group_map_large[4] = {nullptr, nullptr, nullptr, nullptr};
    uint64_t params_pending = 0; // bit p: pipeline p has not yet waited for the latest params upload; kSerialBit: the map stream (serial / drop-in calls)
    hipStream_t copy_stream = nullptr; // per-frame params go up here, so that they never queue behind the map stream
    hipEvent_t ev_params = nullptr;
    std::vector<void *> allocs; // every hipMalloc of this handle
    std::vector<hipGraphExec_t> retired; // graphs replaced while possibly in flight (map_grows): destroyed at the next synchronisation point
    FrameParams *h_params = nullptr; // pinned staging ring
    int32_t *h_scalars = nullptr;    // pinned: [0] n_local, [1] n_new, [2] status, [3] scratch, [4] delta groups; [64..127] the device's scalar block as it came
    int32_t *d_scalars = nullptr;    // the device's scalar block (64 ints: n_local @8, n_local_next @16, n_new @24, n_holes @32, status @48, delta count @56)
    FrameParams *d_params = nullptr;
    uint8_t *d_stage_img = nullptr; // one tightly packed frame on its way into a pitched slot
    float *d_stage_depth = nullptr;
    // the asynchronous uploads' staging for TIGHT rows (dsm_frames_upload_async): frames of the newest call, reused by the next
    // call on the same upload stream (in order behind this one's repack); grown on demand
    uint8_t *d_stage_frames_img = nullptr;
    float *d_stage_frames_depth = nullptr;
    int stage_frames_cap = 0;
    // DSM_FLAG_UPLOAD_STREAM: frames go up on a stream of their own, so that the upload of the next frame overlaps the
    // kernels of the current one; ev_slot[s] = the last frame submitted by dsm_fuse_frame_resident that reads slot s
    // has finished (recorded on the map stream).  Otherwise up_stream == stream.
    hipStream_t up_stream = nullptr;
    bool own_up_stream = false;
    std::vector<hipEvent_t> ev_slot;
    std::vector<char> slot_used;
    // batched replays do not pay an event per frame (a marker packet between kernels costs the 8-stream replay a fifth
    // of its throughput): uploads that follow one wait for ev_fence, recorded once behind the batch
    hipEvent_t ev_fence = nullptr;
    bool fence_pending = false;
    // dsm_frame(s)_upload_async: the last kUpRing asynchronous uploads of this handle, oldest first.  ev = the upload has
    // landed (recorded on the device's upload stream the handle was dealt: ONE stream, so an entry's event covers every
    // entry before it); [lo, hi) = the frame slots it wrote; pending = which consumers have not been ordered behind it
    // yet (bit p: pipeline p, kSerialBit: the map stream, kBatchBit: a batch stream).  A consumer waits for the NEWEST
    // upload that wrote a slot it reads -- not for the latest upload whatever it wrote: in the double-buffered pattern
    // (send chunk k+1, then enqueue chunk k) chunk k waits for upload k, and upload k+1 -- ordered behind chunk k-1, whose
    // slots it overwrites -- runs beside chunk k's kernels.  (Until round 5 there was one event, re-recorded by every
    // upload: chunk k waited for upload k+1, which waited for chunk k-1 -- a handle streaming alone had no overlap at all.)
    struct UpEntry {
        hipEvent_t ev = nullptr;
        int lo = 0, hi = 0;
        uint64_t pending = 0;
    };
    static constexpr int kUpRing = 4;
    UpEntry up_ring[kUpRing];
    int up_n = 0;
    // the frame slots the frames being submitted read (set by the enqueue calls that know them; everything otherwise)
    int read_lo = 0, read_hi = INT32_MAX;
    // ... and the other direction: which enqueue calls still READ which slots.  rd_ring = the last kRdRing calls of
    // dsm_replay_enqueue, oldest first: ev = everything the call enqueued has finished (recorded on the map stream behind it:
    // it covers the calls before it), [lo, hi) = the slots its frames read.  An asynchronous upload waits for the newest
    // entry that reads a slot it overwrites -- not for everything enqueued so far: with three groups of slots in turn
    // (replay.py) the upload of chunk k + 1 waits for chunk k - 2, which finished long ago, so the wait never stalls the
    // hardware queue the upload stream shares with the handle's pipeline streams.  reads_untracked: frames were enqueued
    // some other way (frame by frame, through a batch, timed / debug / drop-in calls): the next upload waits for the whole
    // map stream once (ev_fence), which covers them -- and, the upload stream being in order, every later upload too.
    struct RdEntry {
        hipEvent_t ev = nullptr;
        int lo = 0, hi = 0;
    };
    static constexpr int kRdRing = 4;
    RdEntry rd_ring[kRdRing];
    int rd_n = 0;
    bool reads_untracked = true;
    int up_which = -1; // which of the device's upload streams this handle's uploads take (dealt out at its first upload)
    // dsm_replay_enqueue_host: an event behind each of the last kHostRing calls (dsm_replay_wait)
    static constexpr int kHostRing = 8;
    hipEvent_t host_ring[kHostRing] = {};
    int64_t host_calls = 0;
    bool touched = true; // a per-handle call may have put work on the handle's stream since a batch last ordered itself behind it
    // the other way round: the handle's stream has to come behind the batch it last advanced with (its frames read and
    // write the handle's buffers) -- the wait is put onto the stream when the stream is next used (bind_device), not after
    // every enqueue call of the batch: a wait is a barrier packet on a hardware queue that other batches' graphs share
    hipEvent_t batch_order_ev = nullptr;
    int64_t frames_submitted = 0, frames_done = 0;
    int batches_joined = 0; // dsm_batch_create copied this handle's context: it must not change any more
    int map_upper = 0; // host-side upper bound of the resident map size
    bool tail_large = false; // the map may exceed k_frame_tail's one-workgroup path: graphs hold the tail's extra workgroups
    bool map_valid = false;
    hipEvent_t ev[kNumStages + 2];
    bool have_events = false;
    // inactive store: surfels of keyframes outside the local window, back to back in HBM, with the XYZI
    // shadow the reference publishes and saves (`inactive_pointcloud`)
    dsm_surfel *d_store = nullptr;
    float4 *d_cloud = nullptr;
    void *d_store_tmp = nullptr;
    size_t store_tmp_bytes = 0;
    int store_cap = 0, store_n = 0;
    // drop-in calls (dsm_fuse_map / dsm_fuse_initialize_map): page-locked staging owned by the handle
    uint8_t *pin_frame = nullptr; // one frame, image then depth, rows at the frame slots' pitch
    dsm_surfel *pin_map = nullptr; // shadow of the caller's array: what the last drop-in call returned == the device map
    size_t pin_map_cap = 0;
    int shadow_n = -1;             // -1: the device map is not known to equal the shadow (resident calls in between)
    // delta download: the 64-record groups a frame changed, packed by k_delta_pack (device), and their page-locked landing place
    dsm_surfel *d_delta = nullptr, *pin_delta = nullptr;
    int32_t *d_delta_idx = nullptr, *pin_delta_idx = nullptr;
    int delta_cap_groups = 0;
    bool dirty_flags_clean = false; // every DeviceCtx::grp_dirty flag is 0 (true after a drop-in call; any other frame may set some)
    int64_t dropin_calls = 0, dropin_delta_calls = 0, dropin_delta_groups = 0; // (statistics: dsm_debug_dropin_stats)
    int dropin_last_groups = 0; // groups the previous drop-in call brought back: sizes the next call's first transfer
    double dropin_us[4] = {0, 0, 0, 0}; // host time of the drop-in calls so far: frame staging | map compare / upload | wait for the GPU | patching the host copies
    HostPool *pool = nullptr;
    std::string err;
};

namespace {

int fail(dsm_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(h, expr)                                                                           \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(h, DSM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T> hipError_t dev_alloc(dsm_handle *h, T **out, size_t count) {
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, count * sizeof(T) + 256);
    if (e != hipSuccess) return e;
    h->allocs.push_back(p);
    *out = (T *)p;
    return hipMemsetAsync(p, 0, count * sizeof(T) + 256, h->stream);
}

int order_behind_batch(dsm_handle *h) {
    if (h->batch_order_ev) {
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->batch_order_ev, 0));
        h->batch_order_ev = nullptr;
    }
    return DSM_OK;
}
int bind_device(dsm_handle *h) {
    HIP_TRY(h, hipSetDevice(h->device));
    h->touched = true; // (every per-handle entry point comes through here; the batch calls do not: see batch_stage)
    return order_behind_batch(h);
}

// make room in the parameter rings for n more frames
int reserve_params(dsm_handle *h, int n) {
    if (h->frames_submitted + n - h->frames_done > kParamRing) {
        if (int rc = order_behind_batch(h)) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        h->frames_done = h->frames_submitted;
    }
    return DSM_OK;
}

// inv16: the caller's own world -> cam matrix (pose.inverse() of ITS matrix library, FF.cpp:59), or nullptr: the closed
// form of dsm_math.h
int stage_params(dsm_handle *h, int slot, int ref_idx, const float *pose16, const float *inv16 = nullptr) {
    if (slot < 0 || slot >= h->hc.n_slots) return fail(h, DSM_E_INVALID, "frame slot %d out of range [0,%d)", slot, h->hc.n_slots);
    int rc = reserve_params(h, 1);
    if (rc) return rc;
    const int ring = (int)(h->frames_submitted % kParamRing);
    FrameParams &fp = h->h_params[ring];
    memcpy(fp.pose, pose16, sizeof fp.pose);
    if (inv16) memcpy(fp.inv, inv16, sizeof fp.inv);
    else inverse4<float>(fp.pose, fp.inv); // FF.cpp:59
    fp.ref_idx = ref_idx;
    fp.slot = slot;
    fp.pad[0] = fp.pad[1] = 0;
    if (h->n_pipe == 1) { // everything is on one stream: stream order is enough
        HIP_TRY(h, hipMemcpyAsync(h->d_params + ring, &fp, sizeof fp, hipMemcpyHostToDevice, h->stream));
        return DSM_OK;
    }
    HIP_TRY(h, hipMemcpyAsync(h->d_params + ring, &fp, sizeof fp, hipMemcpyHostToDevice, h->copy_stream));
    HIP_TRY(h, hipEventRecord(h->ev_params, h->copy_stream));
    h->params_pending = ~0ull;
    return DSM_OK;
}

// stage the params of up to n consecutive frames with one host-to-device copy; returns how many
// were staged (limited by the ring's wrap-around point)
// defer_copy: the entries are only written to the page-locked ring; whoever submits the frames copies them to the device on the
// stream that runs them (dsm_replay_enqueue_host: no copy stream, no event in front of a frame group)
int stage_params_batch(dsm_handle *h, int n, const int32_t *slots, const int32_t *ref_idx, const float *poses16, const float *inv_poses16, int *staged,
                       hipStream_t batch_stream = nullptr, bool defer_copy = false) {
    const int ring = (int)(h->frames_submitted % kParamRing);
    int m = n < kParamRing - ring ? n : kParamRing - ring;
    if (m > kParamRing / 2) m = kParamRing / 2;
    int rc = reserve_params(h, m);
    if (rc) return rc;
    for (int i = 0; i < m; i++) {
        if (slots[i] < 0 || slots[i] >= h->hc.n_slots) return fail(h, DSM_E_INVALID, "frame slot %d out of range [0,%d)", slots[i], h->hc.n_slots);
        FrameParams &fp = h->h_params[ring + i];
        memcpy(fp.pose, poses16 + 16 * (size_t)i, sizeof fp.pose);
        if (inv_poses16) memcpy(fp.inv, inv_poses16 + 16 * (size_t)i, sizeof fp.inv);
        else inverse4<float>(fp.pose, fp.inv); // FF.cpp:59
        fp.ref_idx = ref_idx[i];
        fp.slot = slots[i];
        fp.pad[0] = fp.pad[1] = 0;
    }
    if (defer_copy) {
    } else if (h->n_pipe == 1) {
        // (a batch copies them on ITS stream, in order with the kernels that read them: see batch_stage)
        HIP_TRY(h, hipMemcpyAsync(h->d_params + ring, h->h_params + ring, sizeof(FrameParams) * (size_t)m, hipMemcpyHostToDevice, batch_stream ? batch_stream : h->stream));
    } else {
        HIP_TRY(h, hipMemcpyAsync(h->d_params + ring, h->h_params + ring, sizeof(FrameParams) * (size_t)m, hipMemcpyHostToDevice, h->copy_stream));
        HIP_TRY(h, hipEventRecord(h->ev_params, h->copy_stream));
        h->params_pending = ~0ull;
    }
    *staged = m;
    return DSM_OK;
}

// Order `st` behind the asynchronous frame uploads that wrote a slot in [h->read_lo, h->read_hi), once per consumer
// (`bits` = the consumer's bits of UpEntry::pending, see dsm_handle): the newest such upload's event covers the older ones.
int wait_uploads(dsm_handle *h, hipStream_t st, uint64_t bits) {
    for (int i = h->up_n - 1; i >= 0; i--) {
        dsm_handle::UpEntry &e = h->up_ring[i];
        if (!(e.pending & bits) || e.lo >= h->read_hi || h->read_lo >= e.hi) continue;
        HIP_TRY(h, hipStreamWaitEvent(st, e.ev, 0));
        for (int k = 0; k <= i; k++) h->up_ring[k].pending &= ~bits;
        break;
    }
    return DSM_OK;
}
// [lo, hi) of n slot indices
void slot_range(const int32_t *slots, int n, int *lo, int *hi) {
    int a = INT32_MAX, b = 0;
    for (int i = 0; i < n; i++) {
        a = slots[i] < a ? slots[i] : a;
        b = slots[i] + 1 > b ? slots[i] + 1 : b;
    }
    *lo = n > 0 ? a : 0;
    *hi = n > 0 ? b : 0;
}
struct ReadSlots { // the slots an enqueue call reads, for the lifetime of the call
    dsm_handle *h;
    ReadSlots(dsm_handle *h_, int lo, int hi) : h(h_) { h->read_lo = lo; h->read_hi = hi; }
    ~ReadSlots() { h->read_lo = 0; h->read_hi = INT32_MAX; }
};

int fuse_grid_bound(const dsm_handle *h) { return h->hc.cap; }

// What launch_frame is told about the map size for k_frame_tail (dsm_device.h): graphs are captured once, so they carry
// the capacity -- but only from the moment the map can be large (map_grows); eager launches pass the running bound.
int tail_bound(const dsm_handle *h) { return h->tail_large ? h->hc.cap : 0; }

// Before `frames` more frames are enqueued: once the map can pass the size k_frame_tail's one-workgroup path takes, the
// captured graphs that hold the tail are dropped (once per handle) and come back with its extra workgroups.
int map_grows(dsm_handle *h, int frames) {
    if (h->tail_large) return DSM_OK;
    if ((int64_t)h->map_upper + (int64_t)(frames - 1) * h->hc.n_seed <= (int64_t)kTailFastWords * 64) return DSM_OK;
    h->tail_large = true;
    // The graphs may be in flight: they are set aside, not destroyed -- and nothing is waited for (until round 6 the map stream
    // was drained here: 50 ms in the middle of a streamed replay, a quarter of a 3 000-frame run) -- and go at the handle's next
    // synchronisation point (retire_graphs).
    auto retire = [&](hipGraphExec_t &g) { if (g) { h->retired.push_back(g); g = nullptr; } };
    for (int i = 0; i < 4; i++) {
        retire(h->g_group_map[i]);
        h->g_group_map[i] = h->g_group_map_large[i]; // (captured beside the small-map form, if at all)
        h->g_group_map_large[i] = nullptr;
    }
    for (int p = 0; p < kMaxPipes; p++)
        for (int i = 0; i < 2; i++) {
            retire(h->pipe[p].g_map[i]);
            retire(h->pipe[p].g_all[i]);
        }
    return DSM_OK;
}
// after the handle's streams have been waited for: the graphs map_grows set aside
void retire_graphs(dsm_handle *h) {
    for (hipGraphExec_t g : h->retired) (void)hipGraphExecDestroy(g);
    h->retired.clear();
}

// grow-only device scratch of the handle (tail copy of dsm_store_erase, argument blocks of the warps)
int scratch_reserve(dsm_handle *h, size_t need) {
    if (need <= h->store_tmp_bytes) return DSM_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->d_store_tmp) (void)hipFree(h->d_store_tmp);
    h->d_store_tmp = nullptr;
    h->store_tmp_bytes = 0;
    HIP_TRY(h, hipMalloc(&h->d_store_tmp, need * 2));
    h->store_tmp_bytes = need * 2;
    return DSM_OK;
}

// Graphs are captured on a stream of their own that lives for the capture only -- never on a stream that executes work.
// The streams that launch graphs may be shared (the device's reserved per-queue streams serve every batch and every
// frame-group lead of the process): a capture begun on one of them would swallow whatever another thread enqueues there
// meanwhile (its hipGraphLaunch / hipStreamWaitEvent would be recorded into this graph, or fail with a capture error).  A
// graph is not tied to the stream it was captured on; thread-local capture mode keeps other threads' API calls legal.
// Returns "" or what failed.
std::string capture_graph(const std::function<hipError_t(hipStream_t)> &launch, hipGraphExec_t *out) {
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return std::string("hipStreamCreateWithFlags (capture stream): ") + hipGetErrorString(e);
    e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(st);
        return std::string("hipStreamBeginCapture: ") + hipGetErrorString(e);
    }
    hipGraph_t g = nullptr;
    const hipError_t le = launch(st);
    const hipError_t ce = hipStreamEndCapture(st, &g);
    (void)hipStreamDestroy(st);
    if (le != hipSuccess || ce != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        return le != hipSuccess ? std::string("kernel launch during capture: ") + hipGetErrorString(le)
                                : std::string("hipStreamEndCapture: ") + hipGetErrorString(ce);
    }
    const hipError_t ie = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ie != hipSuccess) return std::string("hipGraphInstantiate: ") + hipGetErrorString(ie);
    return std::string();
}

int capture(dsm_handle *h, const DeviceCtx &ctx, bool with_compaction, int lo, int hi, hipGraphExec_t *out) {
    if (*out) return DSM_OK;
    const std::string err = capture_graph([&](hipStream_t st) { return launch_frame(ctx, fuse_grid_bound(h), tail_bound(h), with_compaction, st, nullptr, lo, hi); }, out);
    if (!err.empty()) return fail(h, DSM_E_HIP, "%s", err.c_str());
    return DSM_OK;
}

// Frames that come with the enqueue call (dsm_replay_enqueue_host): `n` frames in page-locked host memory, frame i at
// image + i * img_frame_step / depth + i * depth_frame_step.  They go up ON THE STREAM THAT RUNS THEIR SUPERPIXEL STAGES, right
// in front of them, into the frame slots of the pipelines that take them (slot = pipeline): no upload stream, no event between
// an upload and its consumer, and a transfer holds up only the hardware queue of the work that waits for it anyway.
struct HostFrames {
    const uint8_t *image = nullptr;
    const float *depth = nullptr;
    size_t img_step = 0, img_frame_step = 0, depth_step = 0, depth_frame_step = 0;
};
// frames [first, first + n) of `f0` into the frame slots slot_base + first ..., on `st`
int upload_host_frames(dsm_handle *h, const HostFrames &f0, int slot_base, int first, int n, hipStream_t st) {
    constexpr int which = 3;
    if (n <= 0) return DSM_OK;
    HostFrames f = f0;
    f.image = f0.image + (size_t)first * f0.img_frame_step;
    f.depth = (const float *)((const char *)f0.depth + (size_t)first * f0.depth_frame_step);
    const int slot0 = slot_base + first;
    const int w = h->hc.w, hh = h->hc.h, pitch = h->hc.pitch;
    uint8_t *di = (uint8_t *)h->hc.img_base + (int64_t)slot0 * h->hc.slot_elems;
    float *dd = (float *)h->hc.depth_base + (int64_t)slot0 * h->hc.slot_elems;
    const size_t plane = (size_t)pitch * (size_t)hh;
    const bool img_flat = f.img_step == (size_t)pitch, dep_flat = f.depth_step == (size_t)pitch * 4;
    if (!(which & 1)) {
    } else if (img_flat && (n == 1 || f.img_frame_step == plane)) {
        HIP_TRY(h, hipMemcpyAsync(di, f.image, plane * (size_t)(n - 1) + (size_t)pitch * (size_t)(hh - 1) + (size_t)w, hipMemcpyHostToDevice, st));
    } else {
        for (int i = 0; i < n; i++) {
            const uint8_t *src = f.image + (size_t)i * f.img_frame_step;
            if (img_flat) HIP_TRY(h, hipMemcpyAsync(di + (size_t)i * plane, src, (size_t)pitch * (size_t)(hh - 1) + (size_t)w, hipMemcpyHostToDevice, st));
            else HIP_TRY(h, hipMemcpy2DAsync(di + (size_t)i * plane, (size_t)pitch, src, f.img_step, (size_t)w, (size_t)hh, hipMemcpyHostToDevice, st));
        }
    }
    if (!(which & 2)) {
    } else if (dep_flat && (n == 1 || f.depth_frame_step == plane * 4)) {
        HIP_TRY(h, hipMemcpyAsync(dd, f.depth, (plane * (size_t)(n - 1) + (size_t)pitch * (size_t)(hh - 1) + (size_t)w) * 4, hipMemcpyHostToDevice, st));
    } else {
        for (int i = 0; i < n; i++) {
            const float *src = (const float *)((const char *)f.depth + (size_t)i * f.depth_frame_step);
            if (dep_flat) HIP_TRY(h, hipMemcpyAsync(dd + (size_t)i * plane, src, ((size_t)pitch * (size_t)(hh - 1) + (size_t)w) * 4, hipMemcpyHostToDevice, st));
            else HIP_TRY(h, hipMemcpy2DAsync(dd + (size_t)i * plane, (size_t)pitch * 4, src, f.depth_step, (size_t)w * 4, (size_t)hh, hipMemcpyHostToDevice, st));
        }
    }
    return DSM_OK;
}

// enqueue the kernels of one frame whose params were staged by stage_params: superpixel stages on the
// frame's pipeline stream, fuse + tail on the map stream.  host: the frame itself, to go up in front of them (slot = pipeline)
int submit_frame(dsm_handle *h, bool with_compaction, const HostFrames *host = nullptr) {
    h->shadow_n = -1;
    h->dirty_flags_clean = false;
    h->reads_untracked = true; // (dsm_replay_enqueue, which lists what it reads, puts the flag back)
    if (int rc = map_grows(h, 1)) return rc;
    const int p = (int)(h->frames_submitted % h->n_pipe);
    dsm_handle::Pipe &pp = h->pipe[p];
    const bool eager = (h->cfg.flags & DSM_FLAG_NO_GRAPH) != 0;
    const int wc = with_compaction ? 1 : 0;
    if (h->n_pipe == 1) { // everything on the map stream, one graph
        if (int rc = wait_uploads(h, h->stream, kSerialBit)) return rc;
        if (host) {
            const int ring = (int)(h->frames_submitted % kParamRing);
            HIP_TRY(h, hipMemcpyAsync(h->d_params + ring, h->h_params + ring, sizeof(FrameParams), hipMemcpyHostToDevice, h->stream));
            if (int rc = upload_host_frames(h, *host, 0, 0, 1, h->stream)) return rc;
        }
        if (eager) {
            hipError_t e = launch_frame(pp.ctx, h->map_upper, h->map_upper, with_compaction, h->stream, nullptr);
            if (e != hipSuccess) return fail(h, DSM_E_HIP, "kernel launch: %s", hipGetErrorString(e));
        } else {
            int rc = capture(h, pp.ctx, with_compaction, 0, kNumStages - 1, &pp.g_all[wc]);
            if (rc) return rc;
            HIP_TRY(h, hipGraphLaunch(pp.g_all[wc], h->stream));
        }
    } else {
        // the pipeline's buffers are free once the map stream has finished the frame that used them last,
        // and the frame's params must have landed
        HIP_TRY(h, hipStreamWaitEvent(pp.stream, pp.ev_free ? pp.ev_free : pp.ev_map, 0));
        if (h->params_pending & (1ull << p)) {
            HIP_TRY(h, hipStreamWaitEvent(pp.stream, h->ev_params, 0));
            h->params_pending &= ~(1ull << p);
        }
        if (int rc = wait_uploads(h, pp.stream, 1ull << p)) return rc;
        if (host) { // (behind ev_free: the frame that last read slot p has been fused)
            const int ring = (int)(h->frames_submitted % kParamRing);
            HIP_TRY(h, hipMemcpyAsync(h->d_params + ring, h->h_params + ring, sizeof(FrameParams), hipMemcpyHostToDevice, pp.stream));
            if (int rc = upload_host_frames(h, *host, p, 0, 1, pp.stream)) return rc;
        }
        if (eager) {
            hipError_t e = launch_frame(pp.ctx, h->map_upper, h->map_upper, with_compaction, pp.stream, nullptr, 0, kLastSuperpixelStage);
            if (e != hipSuccess) return fail(h, DSM_E_HIP, "kernel launch: %s", hipGetErrorString(e));
        } else {
            int rc = capture(h, pp.ctx, with_compaction, 0, kLastSuperpixelStage, &pp.g_sp);
            if (rc) return rc;
            HIP_TRY(h, hipGraphLaunch(pp.g_sp, pp.stream));
        }
        HIP_TRY(h, hipEventRecord(pp.ev_sp, pp.stream));
        HIP_TRY(h, hipStreamWaitEvent(h->stream, pp.ev_sp, 0));
        if (eager) {
            hipError_t e = launch_frame(pp.ctx, h->map_upper, h->map_upper, with_compaction, h->stream, nullptr, kLastSuperpixelStage + 1, kNumStages - 1);
            if (e != hipSuccess) return fail(h, DSM_E_HIP, "kernel launch: %s", hipGetErrorString(e));
        } else {
            int rc = capture(h, pp.ctx, with_compaction, kLastSuperpixelStage + 1, kNumStages - 1, &pp.g_map[wc]);
            if (rc) return rc;
            HIP_TRY(h, hipGraphLaunch(pp.g_map[wc], h->stream));
        }
    }
    if (h->n_pipe > 1) {
        HIP_TRY(h, hipEventRecord(pp.ev_map, h->stream));
        pp.ev_free = pp.ev_map;
    }
    h->hc = pp.ctx; // taps read the state of the latest frame
    h->frames_submitted++;
    if (with_compaction) {
        h->map_upper += h->hc.n_seed;
        if (h->map_upper > h->hc.cap) h->map_upper = h->hc.cap;
    }
    return DSM_OK;
}

// Frame groups.  The superpixel stages of a frame need nothing but the frame, so those of G consecutive frames -- which
// live on G different pipelines -- are launched as ONE batch (every kernel once for the G frames, grid z = pipeline: the
// batched instantiations of dsm_kernels.hip), on the stream of the group's first pipeline; fuse + tail of the G frames
// then follow on the map stream in frame order.  The pipelines form two groups (four from depth 16 on) used in turn,
// so the batches of the next group(s) run while the map stream works the previous one off.  Same results as frame by
// frame; the params of the G frames must have been staged, and frames_submitted be a multiple of G.
int group_size(const dsm_handle *h) { return (h->n_pipe == 12 || h->n_pipe == 24) ? h->n_pipe / 3 : h->n_pipe >= 16 ? h->n_pipe / 4 : h->n_pipe / 2; }
bool group_path(const dsm_handle *h) {
    return h->n_pipe >= 4 && h->d_pipe_ctxs && !(h->cfg.flags & DSM_FLAG_NO_GRAPH);
}
int submit_group(dsm_handle *h, const HostFrames *host = nullptr) {
    h->shadow_n = -1;
    h->dirty_flags_clean = false;
    h->reads_untracked = true;
    const int G = group_size(h);
    if (int rc = map_grows(h, G)) return rc;
    const int p0 = (int)(h->frames_submitted % h->n_pipe); // a multiple of G
    const int half = p0 / G;
    // group k runs on the stream of pipeline k: HIP spreads streams over its four hardware queues in creation order, and
    // the groups' first pipelines (0, G, 2G ...) would all sit on one queue -- their batches would run one after another
    dsm_handle::Pipe &lead = h->pipe[half];
    // the group's buffers are free once the map stream has finished the frames that used them last (it is in order:
    // the last pipeline's event covers the others), and the frames' params must have landed
    {
        const dsm_handle::Pipe &last = h->pipe[p0 + G - 1];
        HIP_TRY(h, hipStreamWaitEvent(lead.stream, last.ev_free ? last.ev_free : last.ev_map, 0));
    }
    // Only the lead stream waits for the params upload, yet the bits of all G pipelines are cleared: the group's
    // superpixel launch on `lead` is the only work that reads the params of these G frames before the map stream does,
    // and the map stream waits for lead.ev_sp below.  A later submit_frame on one of these pipelines (ragged end of a
    // replay) stages its own params first, which sets its bit again (stage_params: params_pending = ~0u) -- so a cleared
    // bit never stands for an upload its stream has not been ordered behind.
    const uint64_t mask = ((1ull << G) - 1ull) << p0;
    if (h->params_pending & mask) {
        HIP_TRY(h, hipStreamWaitEvent(lead.stream, h->ev_params, 0));
        h->params_pending &= ~mask;
    }
    // (only `lead` reads these frames' slots before the map stream does -- which waits for lead.ev_sp below -- and only the
    // lead stream's bit is cleared: a later frame-by-frame submit on another of these pipelines waits for itself)
    if (int rc = wait_uploads(h, lead.stream, 1ull << half)) return rc;
    // the G frames themselves, if they came with the call: slots p0 .. p0 + G - 1, behind the wait for the frames that read them last
    if (host) {
        // ... and their parameters in front of them, on the same stream (the ring entries of G consecutive frames are adjacent: the
        // ring's length is a multiple of every group size)
        const int ring = (int)(h->frames_submitted % kParamRing);
        HIP_TRY(h, hipMemcpyAsync(h->d_params + ring, h->h_params + ring, sizeof(FrameParams) * (size_t)G, hipMemcpyHostToDevice, lead.stream));
        if (int rc = upload_host_frames(h, *host, p0, 0, G, lead.stream)) return rc;
    }
    if (!h->g_group[half]) {
        const std::string err = capture_graph([&](hipStream_t st) {
            return launch_frame(lead.ctx, fuse_grid_bound(h), tail_bound(h), true, st, nullptr, 0, kLastSuperpixelStage, h->d_pipe_ctxs + p0, G, 4);
        }, &h->g_group[half]);
        if (!err.empty()) return fail(h, DSM_E_HIP, "%s", err.c_str());
    }
    HIP_TRY(h, hipGraphLaunch(h->g_group[half], lead.stream));
    HIP_TRY(h, hipEventRecord(lead.ev_sp, lead.stream));
    HIP_TRY(h, hipStreamWaitEvent(h->stream, lead.ev_sp, 0));
    // fuse + tail of the G frames, in frame order, as one graph on the map stream (a graph launch per frame costs the
    // map stream more than the two kernels do)
    if (!h->g_group_map[half]) {
        const std::string err = capture_graph([&](hipStream_t st) {
            hipError_t le = hipSuccess;
            for (int j = 0; j < G && le == hipSuccess; j++)
                le = launch_frame(h->pipe[p0 + j].ctx, fuse_grid_bound(h), tail_bound(h), true, st, nullptr, kLastSuperpixelStage + 1, kNumStages - 1);
            return le;
        }, &h->g_group_map[half]);
        if (!err.empty()) return fail(h, DSM_E_HIP, "%s", err.c_str());
        if (!h->tail_large && h->hc.cap > kTailFastWords * 64 && !h->g_group_map_large[half]) { // the large-map form, for later
            const std::string err2 = capture_graph([&](hipStream_t st) {
                hipError_t le = hipSuccess;
                for (int j = 0; j < G && le == hipSuccess; j++)
                    le = launch_frame(h->pipe[p0 + j].ctx, fuse_grid_bound(h), h->hc.cap, true, st, nullptr, kLastSuperpixelStage + 1, kNumStages - 1);
                return le;
            }, &h->g_group_map_large[half]);
            if (!err2.empty()) return fail(h, DSM_E_HIP, "%s", err2.c_str());
        }
    }
    HIP_TRY(h, hipGraphLaunch(h->g_group_map[half], h->stream));
    // ONE event for the group (every record is a marker packet on the map stream, the serial spine of a sequence)
    HIP_TRY(h, hipEventRecord(h->pipe[p0 + G - 1].ev_map, h->stream));
    for (int j = 0; j < G; j++) h->pipe[p0 + j].ev_free = h->pipe[p0 + G - 1].ev_map;
    h->hc = h->pipe[p0 + G - 1].ctx; // taps read the state of the latest frame
    h->frames_submitted += G;
    const int64_t up = (int64_t)h->map_upper + (int64_t)G * h->hc.n_seed;
    h->map_upper = up > h->hc.cap ? h->hc.cap : (int)up;
    return DSM_OK;
}

// run some or all stages of the next frame serially on the map stream (timed replays, state-level taps)
int submit_serial(dsm_handle *h, bool with_compaction, hipEvent_t *ev, int lo, int hi) {
    h->shadow_n = -1;
    h->dirty_flags_clean = false;
    h->reads_untracked = true;
    const int p = (int)(h->frames_submitted % h->n_pipe);
    dsm_handle::Pipe &pp = h->pipe[p];
    if (h->n_pipe > 1 && (h->params_pending & kSerialBit)) {
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_params, 0));
        h->params_pending &= ~kSerialBit;
    }
    if (int rc = wait_uploads(h, h->stream, kSerialBit)) return rc;
    hipError_t e = launch_frame(pp.ctx, h->map_upper, h->map_upper, with_compaction, h->stream, ev, lo, hi);
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "kernel launch: %s", hipGetErrorString(e));
    if (h->n_pipe > 1) {
        HIP_TRY(h, hipEventRecord(pp.ev_map, h->stream));
        pp.ev_free = pp.ev_map;
    }
    h->hc = pp.ctx;
    h->fence_pending = true; // (timed / debug replays: uploads wait for the whole stream)
    if (hi == kNumStages - 1) { // the tail advanced the pipeline's cursor
        h->frames_submitted++;
        if (with_compaction) {
            h->map_upper += h->hc.n_seed;
            if (h->map_upper > h->hc.cap) h->map_upper = h->hc.cap;
        }
    }
    return DSM_OK;
}

// ---- drop-in calls: everything on the map stream, in two parts, so that the host can look at the caller's surfel
// array while the superpixel stages (which need the frame only) already run
int submit_part(dsm_handle *h, bool with_compaction, bool map_part) {
    h->reads_untracked = true;
    if (int rc = map_grows(h, 1)) return rc;
    const int p = (int)(h->frames_submitted % h->n_pipe);
    dsm_handle::Pipe &pp = h->pipe[p];
    if (h->n_pipe > 1 && (h->params_pending & kSerialBit)) {
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_params, 0));
        h->params_pending &= ~kSerialBit;
    }
    if (int rc = wait_uploads(h, h->stream, kSerialBit)) return rc;
    const int lo = map_part ? kLastSuperpixelStage + 1 : 0, hi = map_part ? kNumStages - 1 : kLastSuperpixelStage;
    if (h->cfg.flags & DSM_FLAG_NO_GRAPH) {
        hipError_t e = launch_frame(pp.ctx, h->map_upper, h->map_upper, with_compaction, h->stream, nullptr, lo, hi);
        if (e != hipSuccess) return fail(h, DSM_E_HIP, "kernel launch: %s", hipGetErrorString(e));
    } else {
        hipGraphExec_t *g = map_part ? &pp.g_map[with_compaction ? 1 : 0] : &pp.g_sp_main;
        int rc = capture(h, pp.ctx, with_compaction, lo, hi, g);
        if (rc) return rc;
        HIP_TRY(h, hipGraphLaunch(*g, h->stream));
    }
    if (map_part) {
        if (h->n_pipe > 1) {
            HIP_TRY(h, hipEventRecord(pp.ev_map, h->stream));
            pp.ev_free = pp.ev_map;
        }
        h->hc = pp.ctx;
        h->frames_submitted++;
        if (with_compaction) {
            h->map_upper += h->hc.n_seed;
            if (h->map_upper > h->hc.cap) h->map_upper = h->hc.cap;
        }
    }
    return DSM_OK;
}

constexpr size_t kParChunk = 1 << 20; // bytes per task of the parallel copies

void par_copy(dsm_handle *h, void *dst, const void *src, size_t bytes) {
    const int n = (int)((bytes + kParChunk - 1) / kParChunk);
    h->pool->run(n, [&](int i) {
        const size_t o = (size_t)i * kParChunk, len = bytes - o < kParChunk ? bytes - o : kParChunk;
        memcpy((char *)dst + o, (const char *)src + o, len);
    });
}

bool par_equal(dsm_handle *h, const void *a, const void *b, size_t bytes) {
    const int n = (int)((bytes + kParChunk - 1) / kParChunk);
    std::atomic<int> differ{0};
    h->pool->run(n, [&](int i) {
        const size_t o = (size_t)i * kParChunk, len = bytes - o < kParChunk ? bytes - o : kParChunk;
        if (!differ.load(std::memory_order_relaxed) && memcmp((const char *)a + o, (const char *)b + o, len) != 0) differ.store(1);
    });
    return differ.load() == 0;
}

// page-locked staging of the drop-in calls, grown on demand
int dropin_reserve(dsm_handle *h, size_t map_records) {
    if (!h->pool) { // the caller + up to seven helpers for the page-sized host copies and compares (a quarter of the machine at most)
        const unsigned hw = std::thread::hardware_concurrency();
        h->pool = new HostPool(hw >= 32 ? 7 : hw >= 8 ? 3 : 1);
    }
    if (!h->pin_frame) {
        HIP_TRY(h, hipHostMalloc((void **)&h->pin_frame, (size_t)h->hc.pitch * h->hc.h * 5, hipHostMallocDefault));
        memset(h->pin_frame, 0, (size_t)h->hc.pitch * h->hc.h * 5); // (the pad columns travel with the rows; no kernel reads them)
    }
    if (map_records > h->pin_map_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        size_t cap = h->pin_map_cap ? h->pin_map_cap : (size_t)1 << 16;
        while (cap < map_records) cap *= 2;
        dsm_surfel *n = nullptr;
        HIP_TRY(h, hipHostMalloc((void **)&n, cap * sizeof(dsm_surfel), hipHostMallocDefault));
        if (h->pin_map) HIP_TRY(h, hipHostFree(h->pin_map));
        h->pin_map = n;
        h->pin_map_cap = cap;
        h->shadow_n = -1;
        // the delta download's buffers, sized for the same extent: every 64-record group of it may change in one frame
        const size_t groups = cap / 64 + 1;
        if (h->d_delta) HIP_TRY(h, hipFree(h->d_delta));
        if (h->d_delta_idx) HIP_TRY(h, hipFree(h->d_delta_idx));
        if (h->pin_delta) HIP_TRY(h, hipHostFree(h->pin_delta));
        if (h->pin_delta_idx) HIP_TRY(h, hipHostFree(h->pin_delta_idx));
        h->d_delta = nullptr; h->d_delta_idx = nullptr; h->pin_delta = nullptr; h->pin_delta_idx = nullptr;
        h->delta_cap_groups = 0;
        HIP_TRY(h, hipMalloc((void **)&h->d_delta, groups * 64 * sizeof(dsm_surfel)));
        HIP_TRY(h, hipMalloc((void **)&h->d_delta_idx, groups * sizeof(int32_t)));
        HIP_TRY(h, hipHostMalloc((void **)&h->pin_delta, groups * 64 * sizeof(dsm_surfel), hipHostMallocDefault));
        HIP_TRY(h, hipHostMalloc((void **)&h->pin_delta_idx, groups * sizeof(int32_t), hipHostMallocDefault));
        h->delta_cap_groups = (int)groups;
    }
    return DSM_OK;
}

int sync_and_fetch_counts(dsm_handle *h);
void retire_graphs(dsm_handle *h);

// Drop-in calls, the way back: what the frame changed of the map, into the shadow and into the caller's array.  Before the
// frame the three were equal over the caller's n records (dropin_map_in); the kernels flagged every 64-record group they wrote
// (DeviceCtx::grp_dirty).  dropin_delta_begin goes between the upload of the map and the frame's map stages, dropin_delta_end
// behind them: it packs the flagged groups on the device, brings them over in one transfer -- sized by the previous call's
// count, so that one host wait serves the sizes and the records alike; a second transfer follows only if more changed --
// and patches both host copies.  A frame that changed most of the map takes the plain full download.
int dropin_delta_begin(dsm_handle *h) {
    if (!h->dirty_flags_clean) { // flags left by frames of the resident / batched kind
        HIP_TRY(h, hipMemsetAsync(h->hc.grp_dirty, 0, (size_t)h->hc.cap / 64 + 1, h->stream));
        h->dirty_flags_clean = true;
    }
    HIP_TRY(h, hipMemsetAsync(h->d_scalars + 56, 0, 4, h->stream));
    return DSM_OK;
}
// m_bound: the largest size the map can have now.  Synchronises; on return h_scalars[0..2] hold the frame's counts.
int dropin_delta_end(dsm_handle *h, dsm_surfel *local, int32_t cap_local, int m_bound, int *m_out) {
    const hipError_t e = launch_delta_pack(h->hc, h->d_delta, h->d_delta_idx, h->d_scalars + 56, h->delta_cap_groups, m_bound, h->stream);
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "delta pack: %s", hipGetErrorString(e));
    int guess = h->dropin_last_groups + h->dropin_last_groups / 4 + 8;
    if (guess > h->delta_cap_groups) guess = h->delta_cap_groups;
    HIP_TRY(h, hipMemcpyAsync(h->pin_delta_idx, h->d_delta_idx, (size_t)guess * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->pin_delta, h->d_delta, (size_t)guess * 64 * sizeof(dsm_surfel), hipMemcpyDeviceToHost, h->stream));
    h->shadow_n = -1;
    const auto t_w0 = std::chrono::steady_clock::now();
    if (int rc = sync_and_fetch_counts(h)) return rc;
    const auto t_w1 = std::chrono::steady_clock::now();
    h->dropin_us[2] += std::chrono::duration<double, std::micro>(t_w1 - t_w0).count();
    struct PatchTimer { // (everything from here on is bringing the rest over and patching)
        dsm_handle *h;
        std::chrono::steady_clock::time_point t;
        ~PatchTimer() { h->dropin_us[3] += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t).count(); }
    } patch_timer{h, t_w1};
    const int m = h->h_scalars[0], n_grp = h->h_scalars[4];
    *m_out = m;
    h->dropin_calls++;
    h->dropin_last_groups = n_grp;
    if (m > cap_local) return fail(h, DSM_E_CAPACITY, "%d surfels exceed the caller's capacity %d", m, cap_local);
    if ((size_t)m > h->pin_map_cap) return fail(h, DSM_E_STATE, "drop-in shadow smaller than the map");
    if ((int64_t)n_grp * 64 * 10 > (int64_t)m * 6 || n_grp > h->delta_cap_groups) { // most of it changed: everything, in one piece
        if (m) {
            HIP_TRY(h, hipMemcpy(h->pin_map, h->hc.local, (size_t)m * sizeof(dsm_surfel), hipMemcpyDeviceToHost));
            par_copy(h, local, h->pin_map, (size_t)m * sizeof(dsm_surfel));
        }
        return DSM_OK;
    }
    if (n_grp > guess) { // more changed than the last call suggested: the rest
        HIP_TRY(h, hipMemcpyAsync(h->pin_delta_idx + guess, h->d_delta_idx + guess, (size_t)(n_grp - guess) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->pin_delta + (size_t)guess * 64, h->d_delta + (size_t)guess * 64, (size_t)(n_grp - guess) * 64 * sizeof(dsm_surfel),
                                  hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
    }
    h->dropin_delta_calls++;
    h->dropin_delta_groups += n_grp;
    constexpr int kPatch = 128; // groups per task
    h->pool->run((n_grp + kPatch - 1) / kPatch, [&](int t) {
        for (int i = t * kPatch; i < n_grp && i < (t + 1) * kPatch; i++) {
            const int g = h->pin_delta_idx[i];
            const int recs = m - g * 64 < 64 ? m - g * 64 : 64;
            if (recs <= 0) continue;
            const dsm_surfel *src = h->pin_delta + (size_t)i * 64;
            memcpy(h->pin_map + (size_t)g * 64, src, (size_t)recs * sizeof(dsm_surfel));
            memcpy(local + (size_t)g * 64, src, (size_t)recs * sizeof(dsm_surfel));
        }
    });
    return DSM_OK;
}

// frame into slot 0 without a host wait: the caller's rows go into page-locked staging laid out like the slot itself (the
// slot's pitch; the host threads copy row by row anyway), then one transfer per plane straight into the slot
int dropin_frame(dsm_handle *h, const uint8_t *image, size_t img_step, const float *depth, size_t depth_step) {
    if (!image || !depth) return fail(h, DSM_E_INVALID, "null image/depth");
    const int w = h->hc.w, hh = h->hc.h, pitch = h->hc.pitch;
    if (img_step < (size_t)w || depth_step < (size_t)w * 4) return fail(h, DSM_E_INVALID, "row step smaller than a row");
    const size_t plane = (size_t)pitch * hh;
    uint8_t *pi = h->pin_frame;
    uint8_t *pd = h->pin_frame + plane;
    constexpr int kRows = 16; // rows per task
    h->pool->run((hh + kRows - 1) / kRows, [&](int t) {
        for (int y = t * kRows; y < hh && y < (t + 1) * kRows; y++) {
            memcpy(pi + (size_t)y * pitch, image + (size_t)y * img_step, (size_t)w);
            memcpy(pd + (size_t)y * pitch * 4, (const uint8_t *)depth + (size_t)y * depth_step, (size_t)w * 4);
        }
    });
    HIP_TRY(h, hipMemcpyAsync((uint8_t *)h->hc.img_base, pi, plane, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipMemcpyAsync((float *)h->hc.depth_base, pd, plane * 4, hipMemcpyHostToDevice, h->stream));
    return DSM_OK;
}

// the caller's surfel array becomes the device map -- unless it still is what the previous drop-in call handed back
// (SurfelMap::fuse_map passes the same vector every frame and edits it only when keyframes enter or leave the window)
int dropin_map_in(dsm_handle *h, const dsm_surfel *local, int n) {
    if (n < 0 || (n > 0 && !local)) return fail(h, DSM_E_INVALID, "bad surfel array");
    if (n > h->hc.cap) return fail(h, DSM_E_CAPACITY, "%d surfels exceed the handle's capacity %d", n, h->hc.cap);
    const size_t bytes = (size_t)n * sizeof(dsm_surfel);
    if (h->shadow_n == n && h->map_valid && par_equal(h, local, h->pin_map, bytes)) return DSM_OK;
    h->shadow_n = -1;
    if (n) {
        par_copy(h, h->pin_map, local, bytes);
        HIP_TRY(h, hipMemcpyAsync(h->hc.local, h->pin_map, bytes, hipMemcpyHostToDevice, h->stream));
    }
    h->h_scalars[3] = n;
    HIP_TRY(h, hipMemcpyAsync(h->hc.n_local, &h->h_scalars[3], 4, hipMemcpyHostToDevice, h->stream));
    h->map_upper = n;
    h->map_valid = true;
    return DSM_OK;
}

int check_status(dsm_handle *h) {
    // caller has synchronised and copied status into h_scalars[2]
    const int st = h->h_scalars[2];
    if (st) (void)hipMemsetAsync(h->hc.status, 0, 4, h->stream); // report once, then start clean
    if (st & kStatusCapacity) return fail(h, DSM_E_CAPACITY, "resident surfel capacity %d exceeded", h->hc.cap);
    if (st & kStatusBadLabels) return fail(h, DSM_E_INVALID, "a superpixel had more member pixels than its 15x15 reach allows: the label image was not produced by the assignment stage (dsm_debug_set_label_buffer?)");
    if (st & kStatusBadPick) return fail(h, DSM_E_INVALID, "a pixel had no candidate superpixel below the reference's 1e6 cost sentinel (depth outside the sensor range?); the reference indexes seeds[-1] here");
    return DSM_OK;
}

int sync_and_fetch_counts(dsm_handle *h) {
    // ONE transfer for the whole scalar block (a call into the runtime costs more than the 256 bytes)
    HIP_TRY(h, hipMemcpyAsync(&h->h_scalars[64], h->d_scalars, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    retire_graphs(h); // (they ran on this stream)
    h->h_scalars[0] = h->h_scalars[64 + 8];
    h->h_scalars[1] = h->h_scalars[64 + 24];
    h->h_scalars[2] = h->h_scalars[64 + 48];
    h->h_scalars[4] = h->h_scalars[64 + 56];
    h->frames_done = h->frames_submitted;
    h->map_upper = h->h_scalars[0];
    h->fence_pending = false; // nothing is in flight any more
    std::fill(h->slot_used.begin(), h->slot_used.end(), 0);
    return check_status(h);
}

int upload_frame(dsm_handle *h, int slot, const void *image, size_t img_step, const void *depth, size_t depth_step,
                 hipMemcpyKind kind) {
    if (!image || !depth) return fail(h, DSM_E_INVALID, "null image/depth");
    if (slot < 0 || slot >= h->hc.n_slots) return fail(h, DSM_E_INVALID, "frame slot %d out of range [0,%d)", slot, h->hc.n_slots);
    const int w = h->hc.w, hh = h->hc.h, pitch = h->hc.pitch;
    if (img_step < (size_t)w || depth_step < (size_t)w * 4) return fail(h, DSM_E_INVALID, "row step smaller than a row");
    uint8_t *di = (uint8_t *)h->hc.img_base + (int64_t)slot * h->hc.slot_elems;
    float *dd = (float *)h->hc.depth_base + (int64_t)slot * h->hc.slot_elems;
    hipStream_t up = h->up_stream;
    if (h->own_up_stream) {
        if (h->fence_pending) { // frames enqueued by a batched / timed replay may read any slot
            HIP_TRY(h, hipEventRecord(h->ev_fence, h->stream));
            HIP_TRY(h, hipStreamWaitEvent(up, h->ev_fence, 0));
            h->fence_pending = false;
        }
        if (h->slot_used[(size_t)slot]) HIP_TRY(h, hipStreamWaitEvent(up, h->ev_slot[(size_t)slot], 0)); // frames still reading this slot
    }
    // tightly packed rows (the usual case): one 1-D copy each, then a repack into the pitched slot on the device
    const bool img_tight = img_step == (size_t)w, dep_tight = depth_step == (size_t)w * 4;
    const size_t n = (size_t)w * (size_t)hh;
    const uint8_t *s_img = nullptr;
    const float *s_dep = nullptr;
    if (img_tight) {
        if (kind == hipMemcpyDeviceToDevice) s_img = (const uint8_t *)image;
        else {
            HIP_TRY(h, hipMemcpyAsync(h->d_stage_img, image, n, kind, up));
            s_img = h->d_stage_img;
        }
    } else
        HIP_TRY(h, hipMemcpy2DAsync(di, (size_t)pitch, image, img_step, (size_t)w, (size_t)hh, kind, up));
    if (dep_tight) {
        if (kind == hipMemcpyDeviceToDevice) s_dep = (const float *)depth;
        else {
            HIP_TRY(h, hipMemcpyAsync(h->d_stage_depth, depth, n * 4, kind, up));
            s_dep = h->d_stage_depth;
        }
    } else
        HIP_TRY(h, hipMemcpy2DAsync(dd, (size_t)pitch * 4, depth, depth_step, (size_t)w * 4, (size_t)hh, kind, up));
    if (s_img || s_dep) {
        const hipError_t e = launch_repack(di, dd, pitch, s_img, s_dep, w, hh, up);
        if (e != hipSuccess) return fail(h, DSM_E_HIP, "frame repack: %s", hipGetErrorString(e));
    }
    HIP_TRY(h, hipStreamSynchronize(up)); // the frame is in its slot, the caller may reuse its buffers
    return DSM_OK;
}

int set_map(dsm_handle *h, const dsm_surfel *src, int n) {
    h->shadow_n = -1;
    if (n < 0 || (n > 0 && !src)) return fail(h, DSM_E_INVALID, "bad surfel array");
    if (n > h->hc.cap) return fail(h, DSM_E_CAPACITY, "%d surfels exceed the handle's capacity %d", n, h->hc.cap);
    if (n) HIP_TRY(h, hipMemcpyAsync(h->hc.local, src, (size_t)n * sizeof(dsm_surfel), hipMemcpyHostToDevice, h->stream));
    // the 4-byte count goes through a pinned scratch word; wait so that it can be reused at once
    h->h_scalars[3] = n;
    HIP_TRY(h, hipMemcpyAsync(h->hc.n_local, &h->h_scalars[3], 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->frames_done = h->frames_submitted;
    h->map_upper = n;
    h->map_valid = true;
    return DSM_OK;
}

} // namespace

extern "C" {

int dsm_abi_version(void) { return DSM_ABI_VERSION; }

int dsm_config_init(dsm_config *cfg, int width, int height, float fx, float fy, float cx, float cy, float far_dist,
                    float near_dist, int rgbd) {
    if (!cfg) return DSM_E_INVALID;
    memset(cfg, 0, sizeof *cfg);
    cfg->width = width; cfg->height = height;
    cfg->fx = fx; cfg->fy = fy; cfg->cx = cx; cfg->cy = cy;
    cfg->far_dist = far_dist; cfg->near_dist = near_dist;
    if (rgbd) { // fusion_functions.h:17-21
        cfg->huber_range = 0.05; cfg->baseline = 0.08; cfg->disparity_error = 1.0; cfg->min_tolerate_diff = 0.05;
    } else { // fusion_functions.h:13-16
        cfg->huber_range = 0.4; cfg->baseline = 0.5; cfg->disparity_error = 4.0; cfg->min_tolerate_diff = 0.1;
    }
    return DSM_OK;
}

const char *dsm_last_error(const dsm_handle *h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dsm_create(const dsm_config *cfg, dsm_handle **out) {
    if (!cfg || !out) return fail(nullptr, DSM_E_INVALID, "null argument");
    *out = nullptr;
    const int w = cfg->width, hh = cfg->height;
    if (w < 3 * kCell || hh < 3 * kCell || w > 32767 || hh > 32767)
        return fail(nullptr, DSM_E_INVALID, "image size %dx%d out of range", w, hh);
    // (size mod 8) > 4 (KITTI's 1242x375, 1238x374) leaves border pixels with no candidate seed: they are labelled
    // -1 and belong to no superpixel (dsm_math.h, has_candidate_cell) -- the reference runs these sizes the same
    // way, through reads and writes of superpixel_seeds[-1] that happen to be harmless (FF.cpp:442-451).
    if ((w / kCell) * (hh / kCell) > kMaxSeeds) return fail(nullptr, DSM_E_INVALID, "more than 65535 superpixels");
    if (!(cfg->fx != 0) || !(cfg->fy != 0)) return fail(nullptr, DSM_E_INVALID, "zero focal length");
    // the kernels compare floats against these double constants in fp32 (dsm_math.h, flt_above / flt_below): the
    // neighbouring-float construction holds for positive normal thresholds only
    if (!(cfg->huber_range >= 1e-30 && cfg->huber_range <= 1e30))
        return fail(nullptr, DSM_E_INVALID, "huber_range %g must be a positive, finite threshold", cfg->huber_range);
    if (!(cfg->min_tolerate_diff >= 1e-30 && cfg->min_tolerate_diff <= 1e30))
        return fail(nullptr, DSM_E_INVALID, "min_tolerate_diff %g must be a positive, finite threshold", cfg->min_tolerate_diff);
    if (!(cfg->baseline > 0) || !(cfg->disparity_error > 0)) return fail(nullptr, DSM_E_INVALID, "baseline and disparity_error must be positive");

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(nullptr, DSM_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= n_dev) return fail(nullptr, DSM_E_NO_DEVICE, "device %d not present", cfg->device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, DSM_E_NO_DEVICE, "cannot query device");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, DSM_E_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", cfg->device, prop.gcnArchName);

    dsm_handle *h = new (std::nothrow) dsm_handle();
    if (!h) return fail(nullptr, DSM_E_HIP, "out of host memory");
    h->cfg = *cfg;
    h->device = cfg->device;
    int rc = DSM_OK;
    auto bail = [&](int code) {
        g_create_error = h->err;
        dsm_destroy(h);
        return code;
    };
#define CREATE_TRY(expr)                                                                  \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            fail(h, DSM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));                   \
            return bail(DSM_E_HIP);                                                       \
        }                                                                                 \
    } while (0)
    CREATE_TRY(hipSetDevice(h->device));
    CREATE_TRY(batch_streams_reserve(h->device)); // (before any handle stream of this process: see BatchStreamPool)
    // Depths 12 and 24 run THREE frame groups: their lead streams and the map stream are the device's four reserved
    // streams, one hardware queue each.  (With four groups the map stream shares a queue with one of them, and a stream
    // waiting for an event holds up whatever else is queued behind it on the same hardware queue.)
    const int np_req = cfg->pipeline_depth > 0 ? cfg->pipeline_depth : 4;
    if (np_req == 12 || np_req == 24) h->stream = batch_stream_at(h->device, kBatchStreams - 1);
    if (h->stream) h->own_stream = false;
    else CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->up_stream = h->stream;
    if (cfg->flags & DSM_FLAG_UPLOAD_STREAM) {
        CREATE_TRY(hipStreamCreateWithFlags(&h->up_stream, hipStreamNonBlocking));
        h->own_up_stream = true;
    }

    DeviceCtx &c = h->hc;
    memset(&c, 0, sizeof c);
    c.w = w; c.h = hh;
    c.pitch = (w + 63) / 64 * 64;
    c.gw = w / kCell; c.gh = hh / kCell; // FF.cpp:14-15
    c.gw_magic = c.gw > 1 ? (uint32_t)(0x100000000ull / (uint64_t)c.gw) + 1u : 0u; // (gw == 1: seed_cell special-cases it)
    c.n_seed = c.gw * c.gh;
    c.k.fx = cfg->fx; c.k.fy = cfg->fy; c.k.cx = cfg->cx; c.k.cy = cfg->cy;
    c.far_d = cfg->far_dist; c.near_d = cfg->near_dist;
    c.huber = cfg->huber_range; c.baseline = cfg->baseline;
    c.disp_err = cfg->disparity_error; c.min_tol = cfg->min_tolerate_diff;
    c.slot_elems = (int64_t)c.pitch * c.h;
    c.n_slots = cfg->frame_slots > 0 ? cfg->frame_slots : 2;
    c.cap = cfg->surfel_capacity > 0 ? cfg->surfel_capacity : kDefaultCapacity;
    c.cap = (c.cap + 63) / 64 * 64;
    c.n_params = kParamRing;

    uint8_t *img = nullptr; float *dep = nullptr;
    CREATE_TRY(dev_alloc(h, &img, (size_t)c.slot_elems * c.n_slots));
    CREATE_TRY(dev_alloc(h, &dep, (size_t)c.slot_elems * c.n_slots));
    c.img_base = img; c.depth_base = dep;
    {   // ray coefficients of every pixel column and row (host and device divide alike: correctly rounded fp32)
        std::vector<float> rays((size_t)w + 1 + (size_t)hh + 1);
        for (int x = 0; x <= w; x++) rays[(size_t)x] = ray_coeff(x, cfg->cx, cfg->fx);
        for (int y = 0; y <= hh; y++) rays[(size_t)w + 1 + (size_t)y] = ray_coeff(y, cfg->cy, cfg->fy);
        float *d_rays = nullptr;
        CREATE_TRY(dev_alloc(h, &d_rays, rays.size()));
        CREATE_TRY(hipMemcpyAsync(d_rays, rays.data(), rays.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
        CREATE_TRY(hipStreamSynchronize(h->stream)); // `rays` is on the stack frame's heap block
        c.ray_x = d_rays;
        c.ray_y = d_rays + w + 1;
    }
    CREATE_TRY(dev_alloc(h, &c.local, (size_t)c.cap));
    CREATE_TRY(dev_alloc(h, &c.fresh, (size_t)c.n_seed));
    CREATE_TRY(dev_alloc(h, &c.grp_dirty, (size_t)c.cap / 64 + 1));
    CREATE_TRY(dev_alloc(h, &c.hole_mask, (size_t)c.cap / 64 + 1));
    CREATE_TRY(dev_alloc(h, &c.wave_prefix, (size_t)c.cap / 64 + 1));
    CREATE_TRY(dev_alloc(h, &c.holes, (size_t)c.cap));
    c.n_hole_chunk = c.cap / (64 * kTailChunkWords) + 1;
    CREATE_TRY(dev_alloc(h, &c.hole_chunk, (size_t)c.n_hole_chunk + 2));
    int32_t *scalars = nullptr; // shared: n_local, n_local_next, n_new, n_holes, status
    CREATE_TRY(dev_alloc(h, &scalars, 64));
    h->d_scalars = scalars;
    c.n_local = scalars + 8; c.n_local_next = scalars + 16; c.n_new = scalars + 24;
    c.n_holes = scalars + 32; c.status = scalars + 48;
    CREATE_TRY(dev_alloc(h, &h->d_params, (size_t)kParamRing));
    c.params = h->d_params;
    h->ev_slot.assign((size_t)c.n_slots, nullptr);
    h->slot_used.assign((size_t)c.n_slots, 0);
    CREATE_TRY(hipEventCreateWithFlags(&h->ev_fence, hipEventDisableTiming));
    CREATE_TRY(dev_alloc(h, &h->d_stage_img, (size_t)w * hh));
    CREATE_TRY(dev_alloc(h, &h->d_stage_depth, (size_t)w * hh));
    // per-pipeline superpixel state
    int np = cfg->pipeline_depth > 0 ? cfg->pipeline_depth : 4;
    if (np != 1 && np != 2 && np != 4 && np != 8 && np != 12 && np != 16 && np != 24 && np != 32) { fail(h, DSM_E_INVALID, "pipeline_depth must be 1, 2, 4, 8, 12, 16, 24 or 32"); return bail(DSM_E_INVALID); }
    if ((cfg->flags & DSM_FLAG_WAVE_STAMPS) && !kWaveStamps) {
        fail(h, DSM_E_INVALID, "DSM_FLAG_WAVE_STAMPS: this library was built without phase stamps (-DDSM_WAVE_STAMPS=1: tools/wave_stamps.py builds such a copy)");
        return bail(DSM_E_INVALID);
    }
    h->n_pipe = np;
    if (np > 1) {
        CREATE_TRY(hipEventCreateWithFlags(&h->ev_params, hipEventDisableTiming));
        CREATE_TRY(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    }
    for (int p = 0; p < np; p++) {
        dsm_handle::Pipe &pp = h->pipe[p];
        pp.ctx = c;
        DeviceCtx &q = pp.ctx;
        q.cursor_mul = np; q.cursor_add = p;
        if (np > 1) {
            // frame groups launch on the streams of pipelines 0..3 (submit_group): those are the device's four reserved
            // streams, one per hardware queue (BatchStreamPool), so that the groups' batches never share a queue
            if (np >= 4 && p < kBatchStreams - ((np == 12 || np == 24) ? 1 : 0)) pp.stream = batch_stream_at(h->device, p);
            if (pp.stream) pp.own_stream = false;
            else CREATE_TRY(hipStreamCreateWithFlags(&pp.stream, hipStreamNonBlocking));
            CREATE_TRY(hipEventCreateWithFlags(&pp.ev_sp, hipEventDisableTiming));
            CREATE_TRY(hipEventCreateWithFlags(&pp.ev_map, hipEventDisableTiming));
        }
        CREATE_TRY(dev_alloc(h, &q.label, (size_t)c.slot_elems));
        CREATE_TRY(dev_alloc(h, &q.cand, (size_t)c.slot_elems));
        CREATE_TRY(dev_alloc(h, &q.worklist, (size_t)c.slot_elems));
        CREATE_TRY(dev_alloc(h, &q.core, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.inv_depth, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.core_stage, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.stable_stage, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.tmin, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.first_empty, (size_t)kSweeps * kWorkers));
        CREATE_TRY(dev_alloc(h, &q.gn_hdr, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.normals, (size_t)c.slot_elems * 3));
        CREATE_TRY(dev_alloc(h, &q.plane, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.seeds, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.spawn_rec, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.spawn_ok, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.fused_flag, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.seed_weight, (size_t)c.n_seed));
        CREATE_TRY(dev_alloc(h, &q.spawn_idx, (size_t)c.n_seed));
        int32_t *ps = nullptr; // work_count, cursor, fit_big_count, rest_count[3][2]
        CREATE_TRY(dev_alloc(h, &ps, 64));
        q.work_count = ps + 0; q.cursor = ps + 8; q.fit_big_count = ps + 24; q.rest_count = ps + 32;
        q.fit_small_cap = kFitSmallCap;
        CREATE_TRY(dev_alloc(h, &q.cur, 1));
        CREATE_TRY(dev_alloc(h, &q.rest_list, (size_t)((c.n_seed + 63) / 64) * kRestListCap * 64));
        if ((cfg->flags & DSM_FLAG_WAVE_STAMPS) && kWaveStamps) CREATE_TRY(dev_alloc(h, &q.stamps, (size_t)5 * c.n_seed * 8));
        q.params = h->d_params;
        if (np > 1) CREATE_TRY(hipEventRecord(pp.ev_map, h->stream)); // "buffers free"
    }
    h->hc = h->pipe[0].ctx;
    if (np >= 4) {
        CREATE_TRY(dev_alloc(h, &h->d_pipe_ctxs, (size_t)np));
        DeviceCtx tmp[kMaxPipes];
        for (int p = 0; p < np; p++) tmp[p] = h->pipe[p].ctx;
        CREATE_TRY(hipMemcpyAsync(h->d_pipe_ctxs, tmp, sizeof(DeviceCtx) * (size_t)np, hipMemcpyHostToDevice, h->stream));
        CREATE_TRY(hipStreamSynchronize(h->stream)); // tmp is on the stack
    }
    CREATE_TRY(hipHostMalloc((void **)&h->h_params, sizeof(FrameParams) * kParamRing, hipHostMallocDefault));
    CREATE_TRY(hipHostMalloc((void **)&h->h_scalars, 512, hipHostMallocDefault));
    memset(h->h_scalars, 0, 512);
    for (int i = 0; i <= kNumStages + 1; i++) CREATE_TRY(hipEventCreate(&h->ev[i]));
    h->have_events = true;
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    (void)rc;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        h->generation = g_next_generation++;
        g_live[h] = h->generation;
    }
    *out = h;
    return DSM_OK;
}

void dsm_destroy(dsm_handle *h) {
    if (!h) return;
    {
        std::lock_guard<std::mutex> lk(g_live_mu);
        g_live.erase(h);
    }
    (void)hipSetDevice(h->device);
    if (h->stream && h->batch_order_ev) (void)hipStreamWaitEvent(h->stream, h->batch_order_ev, 0); // (a batch may still be working on this handle's buffers)
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    retire_graphs(h);
    for (int i = 0; i < 4; i++) {
        if (h->g_group[i]) (void)hipGraphExecDestroy(h->g_group[i]);
        if (h->g_group_map[i]) (void)hipGraphExecDestroy(h->g_group_map[i]);
        if (h->g_group_map_large[i]) (void)hipGraphExecDestroy(h->g_group_map_large[i]);
    }
    for (int p = 0; p < kMaxPipes; p++) {
        dsm_handle::Pipe &pp = h->pipe[p];
        if (pp.stream) (void)hipStreamSynchronize(pp.stream);
        if (pp.g_sp) (void)hipGraphExecDestroy(pp.g_sp);
        if (pp.g_sp_main) (void)hipGraphExecDestroy(pp.g_sp_main);
        for (int i = 0; i < 2; i++) {
            if (pp.g_map[i]) (void)hipGraphExecDestroy(pp.g_map[i]);
            if (pp.g_all[i]) (void)hipGraphExecDestroy(pp.g_all[i]);
        }
        if (pp.ev_sp) (void)hipEventDestroy(pp.ev_sp);
        if (pp.ev_map) (void)hipEventDestroy(pp.ev_map);
        if (pp.stream && pp.own_stream) (void)hipStreamDestroy(pp.stream);
    }
    if (h->copy_stream) { (void)hipStreamSynchronize(h->copy_stream); (void)hipStreamDestroy(h->copy_stream); }
    if (h->own_up_stream && h->up_stream) { (void)hipStreamSynchronize(h->up_stream); (void)hipStreamDestroy(h->up_stream); }
    for (hipEvent_t e : h->ev_slot)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_fence) (void)hipEventDestroy(h->ev_fence);
    for (int i = h->up_n - 1; i >= 0; i--) {
        if (i == h->up_n - 1) (void)hipEventSynchronize(h->up_ring[i].ev);
        (void)hipEventDestroy(h->up_ring[i].ev);
    }
    for (int i = 0; i < dsm_handle::kRdRing; i++)
        if (h->rd_ring[i].ev) (void)hipEventDestroy(h->rd_ring[i].ev);
    for (int i = 0; i < dsm_handle::kHostRing; i++)
        if (h->host_ring[i]) (void)hipEventDestroy(h->host_ring[i]);
    if (h->ev_params) (void)hipEventDestroy(h->ev_params);
    if (h->have_events)
        for (int i = 0; i <= kNumStages + 1; i++) (void)hipEventDestroy(h->ev[i]);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->d_stage_frames_img) (void)hipFree(h->d_stage_frames_img);
    if (h->d_stage_frames_depth) (void)hipFree(h->d_stage_frames_depth);
    if (h->d_store) (void)hipFree(h->d_store);
    if (h->d_cloud) (void)hipFree(h->d_cloud);
    if (h->d_store_tmp) (void)hipFree(h->d_store_tmp);
    if (h->h_params) (void)hipHostFree(h->h_params);
    if (h->h_scalars) (void)hipHostFree(h->h_scalars);
    if (h->pin_frame) (void)hipHostFree(h->pin_frame);
    if (h->pin_map) (void)hipHostFree(h->pin_map);
    if (h->pin_delta) (void)hipHostFree(h->pin_delta);
    if (h->pin_delta_idx) (void)hipHostFree(h->pin_delta_idx);
    if (h->d_delta) (void)hipFree(h->d_delta);
    if (h->d_delta_idx) (void)hipFree(h->d_delta_idx);
    delete h->pool;
    if (h->stream && h->own_stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int dsm_host_alloc(void **out, size_t bytes) {
    if (!out) return DSM_E_INVALID;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return fail(nullptr, DSM_E_HIP, "hipHostMalloc of %zu bytes failed", bytes);
    return DSM_OK;
}
void dsm_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// The caller's frames (one pointer + row step each: n cv::Mat pairs) into page-locked memory laid out like frame slots, by a
// few host threads of the library (process-wide, made at the first call; the caller's thread takes part): one core copies
// ~10 GB/s of 1226-pixel rows, a replay at fifteen thousand frames a second needs 35.
int dsm_host_pack_frames(int32_t n, int32_t width, int32_t height, const uint8_t *const *images, const size_t *image_steps,
                         const float *const *depths, const size_t *depth_steps, uint8_t *dst_image, size_t dst_img_step,
                         size_t dst_img_frame_step, float *dst_depth, size_t dst_depth_step, size_t dst_depth_frame_step) {
    if (n < 0 || width <= 0 || height <= 0) return fail(nullptr, DSM_E_INVALID, "dsm_host_pack_frames: negative count or empty image");
    if (n == 0) return DSM_OK;
    if (!images || !image_steps || !depths || !depth_steps || !dst_image || !dst_depth) return fail(nullptr, DSM_E_INVALID, "dsm_host_pack_frames: null argument");
    const size_t row_i = (size_t)width, row_d = (size_t)width * 4;
    if (dst_img_step < row_i || dst_depth_step < row_d) return fail(nullptr, DSM_E_INVALID, "dsm_host_pack_frames: destination row step smaller than a row");
    if (n > 1 && (dst_img_frame_step < dst_img_step * (size_t)height || dst_depth_frame_step < dst_depth_step * (size_t)height))
        return fail(nullptr, DSM_E_INVALID, "dsm_host_pack_frames: destination frame step smaller than a frame");
    for (int i = 0; i < n; i++)
        if (!images[i] || !depths[i] || image_steps[i] < row_i || depth_steps[i] < row_d)
            return fail(nullptr, DSM_E_INVALID, "dsm_host_pack_frames: frame %d: null plane or row step smaller than a row", i);
    static std::mutex mu; // one packing call at a time: the pool runs one job
    static HostPool *pool = nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!pool) {
        const unsigned hw = std::thread::hardware_concurrency();
        pool = new HostPool(hw >= 32 ? 7 : hw >= 8 ? 3 : 1);
    }
    // a task = the image plane of a frame, or a quarter of its depth plane's rows (equal bytes)
    constexpr int kParts = 5;
    pool->run(n * kParts, [&](int t) {
        const int i = t / kParts, part = t % kParts;
        if (part == 0) {
            const uint8_t *src = images[i];
            uint8_t *dst = dst_image + (size_t)i * dst_img_frame_step;
            for (int y = 0; y < height; y++) memcpy(dst + (size_t)y * dst_img_step, src + (size_t)y * image_steps[i], row_i);
        } else {
            const int y0 = (int)((int64_t)height * (part - 1) / 4), y1 = (int)((int64_t)height * part / 4);
            const char *src = (const char *)depths[i];
            char *dst = (char *)dst_depth + (size_t)i * dst_depth_frame_step;
            for (int y = y0; y < y1; y++) memcpy(dst + (size_t)y * dst_depth_step, src + (size_t)y * depth_steps[i], row_d);
        }
    });
    return DSM_OK;
}

int dsm_seed_count(const dsm_handle *h) { return h ? h->hc.n_seed : DSM_E_INVALID; }

int dsm_stream(dsm_handle *h, void **hip_stream) {
    if (!h || !hip_stream) return DSM_E_INVALID;
    if (int rc = bind_device(h)) return rc; // (the caller is about to use the stream: it comes behind the handle's batch first)
    *hip_stream = (void *)h->stream;
    return DSM_OK;
}

// ------------------------------------------------------------------ drop-in calls

int dsm_fuse_initialize_map(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step,
                            const float *depth, size_t depth_step, const float *pose16, dsm_surfel *local,
                            int32_t n_local, dsm_surfel *new_out, int32_t new_cap, int32_t *n_new) {
    return dsm_fuse_initialize_map_inv(h, reference_frame_index, image, img_step, depth, depth_step, pose16, nullptr, local, n_local,
                                       new_out, new_cap, n_new);
}

int dsm_fuse_initialize_map_inv(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step,
                                const float *depth, size_t depth_step, const float *pose16, const float *inv_pose16,
                                dsm_surfel *local, int32_t n_local, dsm_surfel *new_out, int32_t new_cap, int32_t *n_new) {
    if (!h) return DSM_E_INVALID;
    if (!pose16 || !n_new || (new_cap > 0 && !new_out) || new_cap < 0) return fail(h, DSM_E_INVALID, "null/negative argument");
    int rc = bind_device(h);
    if (rc) return rc;
    const int S = h->hc.n_seed;
    if ((rc = dropin_reserve(h, (size_t)(n_local > 0 ? n_local : 0) + (size_t)S))) return rc;
    if ((rc = dropin_frame(h, image, img_step, depth, depth_step))) return rc;
    if ((rc = stage_params(h, 0, reference_frame_index, pose16, inv_pose16))) return rc;
    if ((rc = submit_part(h, false, false))) return rc;            // superpixels run while the host compares / copies the map
    if ((rc = dropin_map_in(h, local, n_local))) return rc;
    if ((rc = dropin_delta_begin(h))) return rc;
    if ((rc = submit_part(h, false, true))) return rc;
    // results: the new surfels behind the shadow's map part, and of the map what the frame changed (no compaction: the
    // fused and the deleted surfels)
    HIP_TRY(h, hipMemcpyAsync(h->pin_map + n_local, h->hc.fresh, (size_t)S * sizeof(dsm_surfel), hipMemcpyDeviceToHost, h->stream));
    int m = 0;
    if ((rc = dropin_delta_end(h, local, n_local, n_local, &m))) { h->shadow_n = -1; return rc; }
    const int k = h->h_scalars[1];
    *n_new = k;
    h->shadow_n = n_local; // no compaction: the device map keeps its size
    if (k > new_cap) return fail(h, DSM_E_CAPACITY, "%d new surfels exceed new_cap %d", k, new_cap);
    if (k) memcpy(new_out, h->pin_map + n_local, (size_t)k * sizeof(dsm_surfel));
    return DSM_OK;
}

int dsm_fuse_map(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step, const float *depth,
                 size_t depth_step, const float *pose16, dsm_surfel *local, int32_t *n_local, int32_t cap,
                 int32_t *n_new) {
    return dsm_fuse_map_inv(h, reference_frame_index, image, img_step, depth, depth_step, pose16, nullptr, local, n_local, cap, n_new);
}

int dsm_fuse_map_inv(dsm_handle *h, int reference_frame_index, const uint8_t *image, size_t img_step, const float *depth,
                     size_t depth_step, const float *pose16, const float *inv_pose16, dsm_surfel *local, int32_t *n_local,
                     int32_t cap, int32_t *n_new) {
    if (!h) return DSM_E_INVALID;
    if (!pose16 || !n_local || !n_new || cap < 0 || *n_local < 0 || *n_local > cap)
        return fail(h, DSM_E_INVALID, "null/negative argument");
    int rc = bind_device(h);
    if (rc) return rc;
    const int n_in = *n_local, S = h->hc.n_seed;
    // after the frame the map holds at most n_in + S surfels (every seed creates at most one)
    size_t n_back = (size_t)n_in + (size_t)S;
    if (n_back > (size_t)h->hc.cap) n_back = (size_t)h->hc.cap;
    if ((rc = dropin_reserve(h, n_back > (size_t)n_in ? n_back : (size_t)n_in))) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    if ((rc = dropin_frame(h, image, img_step, depth, depth_step))) return rc;
    if ((rc = stage_params(h, 0, reference_frame_index, pose16, inv_pose16))) return rc;
    if ((rc = submit_part(h, true, false))) return rc;             // superpixels run while the host compares / copies the map
    const auto t1 = std::chrono::steady_clock::now();
    if ((rc = dropin_map_in(h, local, n_in))) return rc;
    const auto t2 = std::chrono::steady_clock::now();
    h->dropin_us[0] += std::chrono::duration<double, std::micro>(t1 - t0).count();
    h->dropin_us[1] += std::chrono::duration<double, std::micro>(t2 - t1).count();
    if ((rc = dropin_delta_begin(h))) return rc;
    if ((rc = submit_part(h, true, true))) return rc;
    // of the map, what the frame changed comes back: the surfels it fused or deleted, the slots it refilled or moved, what it
    // appended (dropin_delta_end)
    int m = 0;
    if ((rc = dropin_delta_end(h, local, cap, (int)n_back, &m))) return rc;
    *n_new = h->h_scalars[1];
    *n_local = m;
    h->shadow_n = m;
    return DSM_OK;
}

// ------------------------------------------------------------------ resident path

int dsm_map_upload(dsm_handle *h, const dsm_surfel *surfels, int32_t n) {
    if (!h) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    return set_map(h, surfels, n);
}

int dsm_map_size(dsm_handle *h, int32_t *n) {
    if (!h || !n) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    if ((rc = sync_and_fetch_counts(h))) return rc;
    *n = h->h_scalars[0];
    return DSM_OK;
}

int dsm_map_capacity(const dsm_handle *h, int32_t *cap) {
    if (!h || !cap) return DSM_E_INVALID;
    *cap = h->hc.cap;
    return DSM_OK;
}

int dsm_map_download(dsm_handle *h, dsm_surfel *out, int32_t cap, int32_t *n) {
    if (!h || !n || cap < 0) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    if ((rc = sync_and_fetch_counts(h))) return rc;
    const int m = h->h_scalars[0];
    *n = m;
    if (m > cap) return fail(h, DSM_E_CAPACITY, "%d surfels exceed the caller's capacity %d", m, cap);
    if (m && !out) return fail(h, DSM_E_INVALID, "null output");
    if (m) HIP_TRY(h, hipMemcpy(out, h->hc.local, (size_t)m * sizeof(dsm_surfel), hipMemcpyDeviceToHost));
    return DSM_OK;
}

int dsm_map_copy_to_device(dsm_handle *h, void *dst_device, int32_t cap, int32_t *n) {
    if (!h || !n || cap < 0) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    if ((rc = sync_and_fetch_counts(h))) return rc;
    const int m = h->h_scalars[0];
    *n = m;
    if (m > cap) return fail(h, DSM_E_CAPACITY, "%d surfels exceed the caller's capacity %d", m, cap);
    if (m && !dst_device) return fail(h, DSM_E_INVALID, "null output");
    if (m) HIP_TRY(h, hipMemcpy(dst_device, h->hc.local, (size_t)m * sizeof(dsm_surfel), hipMemcpyDeviceToDevice));
    return DSM_OK;
}

// ------------------------------------------------------------------ map maintenance

int dsm_map_warp(dsm_handle *h, const float *warp16) {
    if (!h) return DSM_E_INVALID;
    h->shadow_n = -1;
    if (!warp16) return fail(h, DSM_E_INVALID, "null matrix");
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map");
    int rc = bind_device(h);
    if (rc) return rc;
    // the matrix travels in the kernel arguments: nothing to stage, nothing to wait for
    hipError_t e = launch_warp(h->hc.local, h->hc.n_local, 0, nullptr, warp16, nullptr, 0, h->map_upper, h->stream);
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "warp launch: %s", hipGetErrorString(e));
    return DSM_OK;
}

int dsm_warp_grouped_device(dsm_handle *h, void *surfels_device, int32_t n_groups, const int32_t *offsets,
                            const float *mats16) {
    if (!h) return DSM_E_INVALID;
    if (n_groups < 0 || (n_groups > 0 && (!surfels_device || !offsets || !mats16))) return fail(h, DSM_E_INVALID, "null/negative argument");
    if (n_groups == 0) return DSM_OK;
    for (int g = 0; g < n_groups; g++)
        if (offsets[g] > offsets[g + 1] || offsets[0] != 0) return fail(h, DSM_E_INVALID, "offsets must start at 0 and ascend");
    int rc = bind_device(h);
    if (rc) return rc;
    // argument block (matrices, offsets) in the handle's grow-only scratch, like dsm_store_warp
    const size_t b_m = sizeof(float) * 16 * (size_t)n_groups, b_o = sizeof(int32_t) * ((size_t)n_groups + 1);
    if ((rc = scratch_reserve(h, b_m + b_o))) return rc;
    char *d = (char *)h->d_store_tmp;
    hipError_t e = hipMemcpyAsync(d, mats16, b_m, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + b_m, offsets, b_o, hipMemcpyHostToDevice, h->stream);
    const int n = offsets[n_groups];
    if (e == hipSuccess) e = launch_warp((dsm_surfel *)surfels_device, nullptr, n, (const float *)d, nullptr, (const int32_t *)(d + b_m), n_groups, n, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // the host arrays may be reused
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "grouped warp: %s", hipGetErrorString(e));
    return DSM_OK;
}

int dsm_map_extract(dsm_handle *h, int32_t key, dsm_surfel *out, int32_t cap, int32_t *n) {
    if (!h || !n || cap < 0 || (cap > 0 && !out)) return DSM_E_INVALID;
    h->shadow_n = -1;
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map");
    int rc = bind_device(h);
    if (rc) return rc;
    // staged in `fresh`'s neighbour: the holes array doubles as the index list, the copy goes to a scratch buffer
    if ((rc = sync_and_fetch_counts(h))) return rc;
    const int m = h->h_scalars[0];
    dsm_surfel *d_out = nullptr;
    HIP_TRY(h, hipMalloc((void **)&d_out, sizeof(dsm_surfel) * (size_t)(m > 0 ? m : 1)));
    hipError_t e = launch_extract(h->hc, key, d_out, m, m, h->stream);
    int32_t k = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&h->h_scalars[3], h->hc.n_holes, 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e == hipSuccess) {
        k = h->h_scalars[3];
        if (k > cap) { (void)hipFree(d_out); *n = k; return fail(h, DSM_E_CAPACITY, "%d surfels of keyframe %d exceed cap %d (map left with them deleted)", k, key, cap); }
        if (k) e = hipMemcpy(out, d_out, sizeof(dsm_surfel) * (size_t)k, hipMemcpyDeviceToHost);
    }
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "extract: %s", hipGetErrorString(e));
    *n = k;
    return DSM_OK;
}

int dsm_map_append(dsm_handle *h, const dsm_surfel *surfels, int32_t n) {
    if (!h || n < 0 || (n > 0 && !surfels)) return DSM_E_INVALID;
    h->shadow_n = -1;
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map");
    int rc = bind_device(h);
    if (rc) return rc;
    if ((rc = sync_and_fetch_counts(h))) return rc;
    const int m = h->h_scalars[0];
    if (m + n > h->hc.cap) return fail(h, DSM_E_CAPACITY, "%d + %d surfels exceed the handle's capacity %d", m, n, h->hc.cap);
    if (n) HIP_TRY(h, hipMemcpy(h->hc.local + m, surfels, sizeof(dsm_surfel) * (size_t)n, hipMemcpyHostToDevice));
    hipError_t e = launch_append_count(h->hc, n, h->stream);
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "append: %s", hipGetErrorString(e));
    h->map_upper = m + n;
    return DSM_OK;
}

// ------------------------------------------------------------------ inactive store
namespace {
int store_reserve(dsm_handle *h, int need) {
    if (need <= h->store_cap) return DSM_OK;
    int cap = h->store_cap ? h->store_cap : (1 << 18);
    while (cap < need) {
        if (cap > (1 << 30)) return fail(h, DSM_E_CAPACITY, "inactive store of %d surfels", need);
        cap *= 2;
    }
    dsm_surfel *ns = nullptr;
    float4 *nc = nullptr;
    HIP_TRY(h, hipMalloc((void **)&ns, sizeof(dsm_surfel) * (size_t)cap + 256));
    hipError_t e = hipMalloc((void **)&nc, sizeof(float4) * (size_t)cap);
    if (e != hipSuccess) { (void)hipFree(ns); return fail(h, DSM_E_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    if (h->store_n) {
        e = hipMemcpyAsync(ns, h->d_store, sizeof(dsm_surfel) * (size_t)h->store_n, hipMemcpyDeviceToDevice, h->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(nc, h->d_cloud, sizeof(float4) * (size_t)h->store_n, hipMemcpyDeviceToDevice, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { (void)hipFree(ns); (void)hipFree(nc); return fail(h, DSM_E_HIP, "store growth: %s", hipGetErrorString(e)); }
    if (h->d_store) (void)hipFree(h->d_store);
    if (h->d_cloud) (void)hipFree(h->d_cloud);
    h->d_store = ns;
    h->d_cloud = nc;
    h->store_cap = cap;
    return DSM_OK;
}
} // namespace

int dsm_store_size(dsm_handle *h, int32_t *n) {
    if (!h || !n) return DSM_E_INVALID;
    *n = h->store_n;
    return DSM_OK;
}

int dsm_store_deactivate(dsm_handle *h, int32_t key, int32_t *begin, int32_t *n) {
    if (!h || !begin || !n) return DSM_E_INVALID;
    h->shadow_n = -1;
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map");
    int rc = bind_device(h);
    if (rc) return rc;
    if ((rc = sync_and_fetch_counts(h))) return rc;
    const int m = h->h_scalars[0];
    hipError_t e = launch_mark(h->hc, key, m, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(&h->h_scalars[3], h->hc.n_holes, 4, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "deactivate: %s", hipGetErrorString(e));
    const int k = h->h_scalars[3];
    if ((rc = store_reserve(h, h->store_n + k))) return rc;
    if (k) {
        e = launch_extract_marked(h->hc, h->d_store + h->store_n, k, h->d_cloud + h->store_n, h->stream);
        if (e != hipSuccess) return fail(h, DSM_E_HIP, "deactivate: %s", hipGetErrorString(e));
    }
    *begin = h->store_n;
    *n = k;
    h->store_n += k;
    return DSM_OK;
}

int dsm_store_activate(dsm_handle *h, int32_t begin, int32_t n) {
    if (!h) return DSM_E_INVALID;
    h->shadow_n = -1;
    if (begin < 0 || n < 0 || begin + n > h->store_n) return fail(h, DSM_E_INVALID, "store range [%d,+%d) outside [0,%d)", begin, n, h->store_n);
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map");
    int rc = bind_device(h);
    if (rc) return rc;
    if ((rc = sync_and_fetch_counts(h))) return rc;
    const int m = h->h_scalars[0];
    if (m + n > h->hc.cap) return fail(h, DSM_E_CAPACITY, "%d + %d surfels exceed the handle's capacity %d", m, n, h->hc.cap);
    if (n) HIP_TRY(h, hipMemcpyAsync(h->hc.local + m, h->d_store + begin, sizeof(dsm_surfel) * (size_t)n, hipMemcpyDeviceToDevice, h->stream));
    hipError_t e = launch_append_count(h->hc, n, h->stream);
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "activate: %s", hipGetErrorString(e));
    h->map_upper = m + n;
    return DSM_OK;
}

int dsm_store_erase(dsm_handle *h, int32_t begin, int32_t n) {
    if (!h) return DSM_E_INVALID;
    if (begin < 0 || n < 0 || begin + n > h->store_n) return fail(h, DSM_E_INVALID, "store range [%d,+%d) outside [0,%d)", begin, n, h->store_n);
    int rc = bind_device(h);
    if (rc) return rc;
    const int tail = h->store_n - (begin + n);
    if (n && tail) { // the tail moves down through a scratch copy (the ranges overlap)
        const size_t need = sizeof(dsm_surfel) * (size_t)tail;
        if ((rc = scratch_reserve(h, need))) return rc;
        HIP_TRY(h, hipMemcpyAsync(h->d_store_tmp, h->d_store + begin + n, need, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_store + begin, h->d_store_tmp, need, hipMemcpyDeviceToDevice, h->stream));
        const size_t need_c = sizeof(float4) * (size_t)tail;
        HIP_TRY(h, hipMemcpyAsync(h->d_store_tmp, h->d_cloud + begin + n, need_c, hipMemcpyDeviceToDevice, h->stream));
        HIP_TRY(h, hipMemcpyAsync(h->d_cloud + begin, h->d_store_tmp, need_c, hipMemcpyDeviceToDevice, h->stream));
    }
    h->store_n -= n;
    return DSM_OK;
}

int dsm_store_warp(dsm_handle *h, int32_t n_groups, const int32_t *offsets, const float *mats16, const uint8_t *changed) {
    if (!h) return DSM_E_INVALID;
    if (n_groups < 0 || (n_groups > 0 && (!offsets || !mats16 || !changed))) return fail(h, DSM_E_INVALID, "null/negative argument");
    if (n_groups == 0) return DSM_OK;
    if (offsets[0] != 0 || offsets[n_groups] != h->store_n) return fail(h, DSM_E_INVALID, "offsets must tile the store [0,%d)", h->store_n);
    for (int g = 0; g < n_groups; g++)
        if (offsets[g] > offsets[g + 1]) return fail(h, DSM_E_INVALID, "offsets must ascend");
    if (h->store_n == 0) return DSM_OK;
    int rc = bind_device(h);
    if (rc) return rc;
    const size_t b_m = sizeof(float) * 16 * (size_t)n_groups, b_o = sizeof(int32_t) * ((size_t)n_groups + 1);
    const size_t need = b_m + b_o + (size_t)n_groups;
    if ((rc = scratch_reserve(h, need))) return rc; // the scratch of dsm_store_erase doubles as the argument block
    char *d = (char *)h->d_store_tmp;
    hipError_t e = hipMemcpyAsync(d, mats16, b_m, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + b_m, offsets, b_o, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d + b_m + b_o, changed, (size_t)n_groups, hipMemcpyHostToDevice, h->stream);
    if (e == hipSuccess)
        e = launch_warp(h->d_store, nullptr, h->store_n, (const float *)d, nullptr, (const int32_t *)(d + b_m), n_groups, h->store_n, h->stream,
                        (const uint8_t *)(d + b_m + b_o), h->d_cloud);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream); // the host arrays may be reused
    if (e != hipSuccess) return fail(h, DSM_E_HIP, "store warp: %s", hipGetErrorString(e));
    return DSM_OK;
}

int dsm_store_download(dsm_handle *h, int32_t begin, int32_t n, dsm_surfel *surfels_out, float *xyzi_out) {
    if (!h) return DSM_E_INVALID;
    if (begin < 0 || n < 0 || begin + n > h->store_n) return fail(h, DSM_E_INVALID, "store range [%d,+%d) outside [0,%d)", begin, n, h->store_n);
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (n && surfels_out) HIP_TRY(h, hipMemcpy(surfels_out, h->d_store + begin, sizeof(dsm_surfel) * (size_t)n, hipMemcpyDeviceToHost));
    if (n && xyzi_out) HIP_TRY(h, hipMemcpy(xyzi_out, h->d_cloud + begin, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost));
    return DSM_OK;
}

int dsm_frame_upload(dsm_handle *h, int slot, const uint8_t *image, size_t img_step, const float *depth,
                     size_t depth_step) {
    if (!h) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    return upload_frame(h, slot, image, img_step, depth, depth_step, hipMemcpyHostToDevice);
}

int dsm_frame_upload_device(dsm_handle *h, int slot, const void *image_dev, size_t img_step, const void *depth_dev,
                            size_t depth_step) {
    if (!h) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    return upload_frame(h, slot, image_dev, img_step, depth_dev, depth_step, hipMemcpyDeviceToDevice);
}

int dsm_frame_pitch(const dsm_handle *h, int32_t *pitch) {
    if (!h || !pitch) return DSM_E_INVALID;
    *pitch = h->hc.pitch;
    return DSM_OK;
}

int dsm_frame_upload_async(dsm_handle *h, int slot, const uint8_t *image, size_t img_step, const float *depth, size_t depth_step) {
    return dsm_frames_upload_async(h, slot, 1, image, img_step, 0, depth, depth_step, 0);
}

int dsm_frames_upload_async(dsm_handle *h, int slot0, int n, const uint8_t *image, size_t img_step, size_t img_frame_step,
                            const float *depth, size_t depth_step, size_t depth_frame_step) {
    if (!h) return DSM_E_INVALID;
    if (!image || !depth) return fail(h, DSM_E_INVALID, "null image/depth");
    if (n < 1 || slot0 < 0 || slot0 + n > h->hc.n_slots) return fail(h, DSM_E_INVALID, "frame slots [%d,%d) out of range [0,%d)", slot0, slot0 + n, h->hc.n_slots);
    const int w = h->hc.w, hh = h->hc.h, pitch = h->hc.pitch;
    if (img_step < (size_t)w || depth_step < (size_t)w * 4) return fail(h, DSM_E_INVALID, "row step smaller than a row");
    if (n > 1 && (img_frame_step < img_step * (size_t)hh || depth_frame_step < depth_step * (size_t)hh)) return fail(h, DSM_E_INVALID, "frame step smaller than a frame");
    // A handle that advances with a batch and whose own stream has carried nothing since the batch last ordered itself
    // behind it (`touched` false: only batch calls since): whatever may still read these slots is covered by the batch's
    // latest marker -- the upload stream waits for THAT, directly.  The handle's own stream is not involved: no wait and no
    // marker on it now, and none for the batch's next call to come behind (those are barrier packets on hardware queues
    // that other batches' graphs share: with 128 streamed subsequences, 256 of them per chunk).
    const bool via_batch = h->batch_order_ev != nullptr && !h->touched;
    if (via_batch) HIP_TRY(h, hipSetDevice(h->device));
    else if (int rc = bind_device(h)) return rc;
    hipStream_t up = device_upload_stream(h->device, &h->up_which, h->batches_joined > 0); // (decided at the handle's first upload)
    if (!up) return fail(h, DSM_E_HIP, "no upload stream on device %d", h->device);
    // behind the frames that may still read these slots (dsm_handle::rd_ring): the newest dsm_replay_enqueue call that
    // reads one of them -- or, if frames were enqueued some other way since the last upload, behind everything enqueued so
    // far for this handle (its map stream runs fuse + tail of every frame after the superpixel stages that read the slots,
    // and waits for the batches the handle takes part in)
    if (via_batch) {
        if (h->reads_untracked) {
            HIP_TRY(h, hipStreamWaitEvent(up, h->batch_order_ev, 0));
            h->reads_untracked = false;
            h->rd_n = 0;
        }
    } else if (h->reads_untracked) {
        HIP_TRY(h, hipEventRecord(h->ev_fence, h->stream));
        HIP_TRY(h, hipStreamWaitEvent(up, h->ev_fence, 0));
        h->reads_untracked = false;
        h->rd_n = 0; // (covered by the fence; the events stay in their places for reuse)
    } else {
        for (int i = h->rd_n - 1; i >= 0; i--) {
            const dsm_handle::RdEntry &e = h->rd_ring[i];
            if (e.lo >= slot0 + n || slot0 >= e.hi) continue;
            HIP_TRY(h, hipStreamWaitEvent(up, e.ev, 0));
            break;
        }
    }
    uint8_t *di = (uint8_t *)h->hc.img_base + (int64_t)slot0 * h->hc.slot_elems;
    float *dd = (float *)h->hc.depth_base + (int64_t)slot0 * h->hc.slot_elems;
    const size_t plane = (size_t)pitch * (size_t)hh; // elements of one slot
    // Rows laid out with the slot's own pitch go up as ONE transfer per plane -- and n frames laid out back to back like
    // the slots themselves as one transfer per plane for all of them (a transfer costs ~10 us before its first byte:
    // two per frame hold a 2.4 MB frame to a third of the link's rate).  Any other row step goes row by row (a 2-D copy
    // is hundreds of small DMA transfers: correct, and several times slower).
    // TIGHT rows (w elements apart, frames back to back) where the slots are pitched: the link carries w of every pitch elements
    // (4.4 % fewer bytes at 1226 pixels) -- one transfer per plane into a staging buffer, one kernel that sets the rows w apart
    // to their pitch, both on the upload stream.  (Accepted, and far better than the row-by-row copy below -- but NOT faster than
    // rows at the slots' pitch: the kernel sits between two transfers of its stream, and 128 streamed subsequences reach 19.4-21.2 k
    // frames/s like this against 21.7 k with pitched rows, VERDICT r05's tight-row upload measured.  dsm_host_pack_frames lays
    // frames out at the pitch.)
    const bool img_flat = img_step == (size_t)pitch, dep_flat = depth_step == (size_t)pitch * 4;
    const size_t tight = (size_t)w * (size_t)hh;
    const bool img_tight = !img_flat && img_step == (size_t)w && (n == 1 || img_frame_step == tight);
    const bool dep_tight = !dep_flat && depth_step == (size_t)w * 4 && (n == 1 || depth_frame_step == tight * 4);
    if ((img_tight || dep_tight) && n > h->stage_frames_cap) {
        HIP_TRY(h, hipStreamSynchronize(up)); // (an earlier call's repack may still read the old buffers)
        if (h->d_stage_frames_img) HIP_TRY(h, hipFree(h->d_stage_frames_img));
        if (h->d_stage_frames_depth) HIP_TRY(h, hipFree(h->d_stage_frames_depth));
        h->d_stage_frames_img = nullptr; h->d_stage_frames_depth = nullptr; h->stage_frames_cap = 0;
        HIP_TRY(h, hipMalloc((void **)&h->d_stage_frames_img, tight * (size_t)n));
        HIP_TRY(h, hipMalloc((void **)&h->d_stage_frames_depth, tight * (size_t)n * 4));
        h->stage_frames_cap = n;
    }
    if (img_tight) HIP_TRY(h, hipMemcpyAsync(h->d_stage_frames_img, image, tight * (size_t)n, hipMemcpyHostToDevice, up));
    if (dep_tight) HIP_TRY(h, hipMemcpyAsync(h->d_stage_frames_depth, depth, tight * (size_t)n * 4, hipMemcpyHostToDevice, up));
    if (img_tight || dep_tight) {
        const hipError_t e = launch_repack_frames(di, dd, pitch, (int64_t)plane, img_tight ? h->d_stage_frames_img : nullptr,
                                                  dep_tight ? h->d_stage_frames_depth : nullptr, w, hh, n, up);
        if (e != hipSuccess) return fail(h, DSM_E_HIP, "frame repack: %s", hipGetErrorString(e));
    }
    if (img_tight) { // (already on its way)
    } else if (img_flat && (n == 1 || img_frame_step == plane)) {
        HIP_TRY(h, hipMemcpyAsync(di, image, plane * (size_t)(n - 1) + (size_t)pitch * (size_t)(hh - 1) + (size_t)w, hipMemcpyHostToDevice, up));
    } else {
        for (int i = 0; i < n; i++) {
            const uint8_t *src = image + (size_t)i * img_frame_step;
            if (img_flat) HIP_TRY(h, hipMemcpyAsync(di + (size_t)i * plane, src, (size_t)pitch * (size_t)(hh - 1) + (size_t)w, hipMemcpyHostToDevice, up));
            else HIP_TRY(h, hipMemcpy2DAsync(di + (size_t)i * plane, (size_t)pitch, src, img_step, (size_t)w, (size_t)hh, hipMemcpyHostToDevice, up));
        }
    }
    if (dep_tight) { // (already on its way)
    } else if (dep_flat && (n == 1 || depth_frame_step == plane * 4)) {
        HIP_TRY(h, hipMemcpyAsync(dd, depth, (plane * (size_t)(n - 1) + (size_t)pitch * (size_t)(hh - 1) + (size_t)w) * 4, hipMemcpyHostToDevice, up));
    } else {
        for (int i = 0; i < n; i++) {
            const float *src = (const float *)((const char *)depth + (size_t)i * depth_frame_step);
            if (dep_flat) HIP_TRY(h, hipMemcpyAsync(dd + (size_t)i * plane, src, ((size_t)pitch * (size_t)(hh - 1) + (size_t)w) * 4, hipMemcpyHostToDevice, up));
            else HIP_TRY(h, hipMemcpy2DAsync(dd + (size_t)i * plane, (size_t)pitch * 4, src, depth_step, (size_t)w * 4, (size_t)hh, hipMemcpyHostToDevice, up));
        }
    }
    // a ring entry for this upload; when the ring is full the oldest entry is folded into the one after it (whose event
    // is later on the same stream: waiting for it instead is safe) and its event is reused
    hipEvent_t ev = nullptr;
    if (h->up_n == dsm_handle::kUpRing) {
        dsm_handle::UpEntry old = h->up_ring[0];
        for (int i = 1; i < h->up_n; i++) h->up_ring[i - 1] = h->up_ring[i];
        h->up_n--;
        dsm_handle::UpEntry &nx = h->up_ring[0];
        if (old.pending) {
            nx.lo = old.lo < nx.lo ? old.lo : nx.lo;
            nx.hi = old.hi > nx.hi ? old.hi : nx.hi;
            nx.pending |= old.pending;
        }
        ev = old.ev;
    }
    if (!ev) HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    dsm_handle::UpEntry &e = h->up_ring[h->up_n++];
    e.ev = ev;
    e.lo = slot0;
    e.hi = slot0 + n;
    e.pending = ~0ull;
    HIP_TRY(h, hipEventRecord(e.ev, up));
    return DSM_OK;
}

int dsm_frame_uploads_wait(dsm_handle *h) {
    if (!h) return DSM_E_INVALID;
    if (!h->up_n) return DSM_OK;
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipEventSynchronize(h->up_ring[h->up_n - 1].ev));
    return DSM_OK;
}

int dsm_fuse_frame_resident(dsm_handle *h, int slot, int reference_frame_index, const float *pose16) {
    return dsm_fuse_frame_resident_inv(h, slot, reference_frame_index, pose16, nullptr);
}

int dsm_fuse_frame_resident_inv(dsm_handle *h, int slot, int reference_frame_index, const float *pose16, const float *inv_pose16) {
    if (!h) return DSM_E_INVALID;
    if (!pose16) return fail(h, DSM_E_INVALID, "null pose");
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map: call dsm_map_upload first (n may be 0)");
    int rc = bind_device(h);
    if (rc) return rc;
    const ReadSlots reads(h, slot, slot + 1);
    if ((rc = stage_params(h, slot, reference_frame_index, pose16, inv_pose16))) return rc;
    if ((rc = submit_frame(h, true))) return rc;
    if (!h->own_up_stream) return DSM_OK;
    // live path: the next upload into this slot waits for exactly this frame
    if (!h->ev_slot[(size_t)slot]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_slot[(size_t)slot], hipEventDisableTiming));
    HIP_TRY(h, hipEventRecord(h->ev_slot[(size_t)slot], h->stream));
    h->slot_used[(size_t)slot] = 1;
    return DSM_OK;
}

int dsm_replay_enqueue(dsm_handle *h, int32_t n, const int32_t *slots, const int32_t *ref_idx, const float *poses16) {
    return dsm_replay_enqueue_inv(h, n, slots, ref_idx, poses16, nullptr);
}

int dsm_replay_enqueue_host(dsm_handle *h, int32_t n, const uint8_t *image, size_t img_step, size_t img_frame_step, const float *depth,
                            size_t depth_step, size_t depth_frame_step, const int32_t *ref_idx, const float *poses16, const float *inv_poses16) {
    if (!h) return DSM_E_INVALID;
    if (n < 0 || (n > 0 && (!image || !depth || !ref_idx || !poses16))) return fail(h, DSM_E_INVALID, "null/negative argument");
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map: call dsm_map_upload first (n may be 0)");
    if (h->hc.n_slots < h->n_pipe) return fail(h, DSM_E_INVALID, "dsm_replay_enqueue_host keeps a frame in the slot of its pipeline: %d frame slots for pipeline_depth %d", h->hc.n_slots, h->n_pipe);
    if (img_step < (size_t)h->hc.w || depth_step < (size_t)h->hc.w * 4) return fail(h, DSM_E_INVALID, "row step smaller than a row");
    if (n > 1 && (img_frame_step < img_step * (size_t)h->hc.h || depth_frame_step < depth_step * (size_t)h->hc.h)) return fail(h, DSM_E_INVALID, "frame step smaller than a frame");
    int rc = bind_device(h);
    if (rc) return rc;
    // the frames read -- and overwrite -- the slots of the pipelines they run on: uploads of the asynchronous kind that
    // follow wait for the whole stream
    std::vector<int32_t> slots((size_t)(n > 0 ? n : 0));
    for (int i = 0; i < n; i++) slots[(size_t)i] = (int32_t)((h->frames_submitted + i) % h->n_pipe);
    HostFrames hf;
    hf.img_step = img_step; hf.img_frame_step = img_frame_step; hf.depth_step = depth_step; hf.depth_frame_step = depth_frame_step;
    for (int i = 0; i < n;) {
        int m = 0;
        if ((rc = stage_params_batch(h, n - i, slots.data() + i, ref_idx + i, poses16 + 16 * (size_t)i,
                                     inv_poses16 ? inv_poses16 + 16 * (size_t)i : nullptr, &m, nullptr, true))) return rc;
        for (int j = 0; j < m;) {
            const int G = group_size(h);
            hf.image = image + (size_t)(i + j) * img_frame_step;
            hf.depth = (const float *)((const char *)depth + (size_t)(i + j) * depth_frame_step);
            if (group_path(h) && m - j >= G && h->frames_submitted % G == 0) {
                if ((rc = submit_group(h, &hf))) return rc;
                j += G;
            } else {
                if ((rc = submit_frame(h, true, &hf))) return rc;
                j++;
            }
        }
        i += m;
    }
    if (n) {
        h->fence_pending = true;
        h->reads_untracked = true;
        // an event behind this call's frames, for dsm_replay_wait (the page-locked frames may be rewritten once it has fired)
        hipEvent_t &ev = h->host_ring[h->host_calls % dsm_handle::kHostRing];
        if (!ev) HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(ev, h->stream));
        h->host_calls++;
    }
    return DSM_OK;
}

int dsm_replay_wait(dsm_handle *h, int32_t calls_back) {
    if (!h) return DSM_E_INVALID;
    if (calls_back < 0 || calls_back >= dsm_handle::kHostRing) return fail(h, DSM_E_INVALID, "calls_back %d out of range [0, %d)", calls_back, dsm_handle::kHostRing);
    if (h->host_calls <= calls_back) return DSM_OK; // no such call yet: nothing to wait for
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipEventSynchronize(h->host_ring[(h->host_calls - 1 - calls_back) % dsm_handle::kHostRing]));
    return DSM_OK;
}

int dsm_replay_enqueue_inv(dsm_handle *h, int32_t n, const int32_t *slots, const int32_t *ref_idx, const float *poses16,
                           const float *inv_poses16) {
    if (!h) return DSM_E_INVALID;
    if (n < 0 || (n > 0 && (!slots || !ref_idx || !poses16))) return fail(h, DSM_E_INVALID, "null/negative argument");
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map: call dsm_map_upload first (n may be 0)");
    int rc = bind_device(h);
    if (rc) return rc;
    int r_lo, r_hi;
    slot_range(slots, n, &r_lo, &r_hi);
    const ReadSlots reads(h, r_lo, r_hi);
    const bool untracked_before = h->reads_untracked;
    for (int i = 0; i < n;) {
        int m = 0;
        if ((rc = stage_params_batch(h, n - i, slots + i, ref_idx + i, poses16 + 16 * (size_t)i,
                                     inv_poses16 ? inv_poses16 + 16 * (size_t)i : nullptr, &m))) return rc;
        for (int j = 0; j < m;) {
            const int G = group_size(h);
            if (group_path(h) && m - j >= G && h->frames_submitted % G == 0) {
                if ((rc = submit_group(h))) return rc;
                j += G;
            } else {
                if ((rc = submit_frame(h, true))) return rc;
                j++;
            }
        }
        i += m;
    }
    if (n) {
        h->fence_pending = true;
        // what this call reads, for the uploads that follow (dsm_handle::rd_ring); the oldest entry folds into the one after it
        hipEvent_t ev = nullptr;
        if (h->rd_n == dsm_handle::kRdRing) {
            const dsm_handle::RdEntry old = h->rd_ring[0];
            for (int i = 1; i < h->rd_n; i++) h->rd_ring[i - 1] = h->rd_ring[i];
            h->rd_n--;
            dsm_handle::RdEntry &nx = h->rd_ring[0];
            nx.lo = old.lo < nx.lo ? old.lo : nx.lo;
            nx.hi = old.hi > nx.hi ? old.hi : nx.hi;
            ev = old.ev;
        }
        if (!ev) ev = h->rd_ring[h->rd_n].ev; // (left behind by an earlier reset)
        if (!ev) HIP_TRY(h, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        dsm_handle::RdEntry &e = h->rd_ring[h->rd_n++];
        e.ev = ev;
        e.lo = r_lo;
        e.hi = r_hi;
        HIP_TRY(h, hipEventRecord(e.ev, h->stream));
        h->reads_untracked = untracked_before;
    }
    return DSM_OK;
}

int dsm_synchronize(dsm_handle *h) {
    if (!h) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    return sync_and_fetch_counts(h);
}

int dsm_last_new_count(dsm_handle *h, int32_t *n_new) {
    if (!h || !n_new) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    if ((rc = sync_and_fetch_counts(h))) return rc;
    *n_new = h->h_scalars[1];
    return DSM_OK;
}

// ------------------------------------------------------------------ taps

// The label plane is 16 bits per pixel on the device (label_t); the taps speak the reference's int.
static int labels_to_host(dsm_handle *h, int32_t *out) {
    const size_t w = (size_t)h->hc.w, n = w * (size_t)h->hc.h;
    std::vector<label_t> tmp(n);
    HIP_TRY(h, hipMemcpy2D(tmp.data(), w * sizeof(label_t), h->hc.label, (size_t)h->hc.pitch * sizeof(label_t), w * sizeof(label_t),
                           (size_t)h->hc.h, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++) out[i] = tmp[i] == kNoLabel ? -1 : (int32_t)tmp[i];
    return DSM_OK;
}

int dsm_get_labels(dsm_handle *h, int32_t *out) {
    if (!h || !out) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return labels_to_host(h, out);
}

int dsm_get_seeds(dsm_handle *h, dsm_seed *out) {
    if (!h || !out) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->hc.seeds, (size_t)h->hc.n_seed * sizeof(dsm_seed), hipMemcpyDeviceToHost));
    return DSM_OK;
}

// ------------------------------------------------------------------ state-level test taps

int dsm_debug_run_stages(dsm_handle *h, int slot, int reference_frame_index, const float *pose16, int first_stage,
                         int last_stage) {
    if (!h) return DSM_E_INVALID;
    if (!pose16 || first_stage < 0 || last_stage >= kNumStages || first_stage > last_stage) return fail(h, DSM_E_INVALID, "bad stage range");
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map: call dsm_map_upload first (n may be 0)");
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->frames_done = h->frames_submitted;
    if ((rc = stage_params(h, slot, reference_frame_index, pose16))) return rc; // ring slot of the current cursor
    if ((rc = submit_serial(h, true, nullptr, first_stage, last_stage))) return rc;
    return sync_and_fetch_counts(h);
}

int dsm_debug_get_label_buffer(dsm_handle *h, int which, int32_t *out) {
    if (!h || !out || which < 0 || which > 1) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    return labels_to_host(h, out);
}

int dsm_debug_set_label_buffer(dsm_handle *h, int which, const int32_t *in) {
    if (!h || !in || which < 0 || which > 1) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    // the kernels index per-seed arrays with the labels they read (tmin[label], core[label]): an injected image must hold
    // seed indices, and -1 exactly where the assignment stage itself writes it (pixels beyond every cell's reach)
    for (int y = 0; y < h->hc.h; y++)
        for (int x = 0; x < h->hc.w; x++) {
            const int32_t l = in[(size_t)y * h->hc.w + x];
            const bool reach = has_candidate_cell(x, y, h->hc.gw, h->hc.gh);
            if (l >= h->hc.n_seed || l < -1 || (l == -1) == reach)
                return fail(h, DSM_E_INVALID, "label %d at (%d, %d) is not a superpixel index of this %d-seed grid", l, x, y, h->hc.n_seed);
        }
    const size_t w = (size_t)h->hc.w, n = w * (size_t)h->hc.h;
    std::vector<label_t> tmp(n);
    for (size_t i = 0; i < n; i++) tmp[i] = in[i] < 0 ? (label_t)kNoLabel : (label_t)in[i];
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy2D(h->hc.label, (size_t)h->hc.pitch * sizeof(label_t), tmp.data(), w * sizeof(label_t), w * sizeof(label_t),
                           (size_t)h->hc.h, hipMemcpyHostToDevice));
    return DSM_OK;
}

int dsm_debug_get_seed_state(dsm_handle *h, float *core4, int32_t *stable) {
    if (!h || !core4 || !stable) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t S = (size_t)h->hc.n_seed;
    HIP_TRY(h, hipMemcpy(core4, h->hc.core, S * 16, hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(stable, h->hc.tmin, S * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < S; i++) stable[i] = stable[i] == kIntMax ? 1 : 0;
    return DSM_OK;
}

int dsm_debug_set_seed_state(dsm_handle *h, const float *core4, const int32_t *stable) {
    if (!h || !core4 || !stable) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t S = (size_t)h->hc.n_seed;
    std::vector<double> inv(S);
    std::vector<int32_t> t(S);
    for (size_t i = 0; i < S; i++) {
        inv[i] = 1.0 / (double)core4[4 * i + 3];
        t[i] = stable[i] ? kIntMax : -1;
    }
    HIP_TRY(h, hipMemcpy(h->hc.core, core4, S * 16, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->hc.inv_depth, inv.data(), S * 8, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->hc.tmin, t.data(), S * 4, hipMemcpyHostToDevice));
    return DSM_OK;
}

// test knob: the longest list the short-column tier of k_seed_fit takes in batched launches (lets a test push ordinary
// groups of seeds through the queue tier).  The contexts are baked into captured graphs and copied by dsm_batch_create,
// so it must be called on a fresh handle: before the first frame and before the handle joins a batch.
int dsm_debug_set_fit_small_cap(dsm_handle *h, int32_t cap) {
    if (!h) return DSM_E_INVALID;
    if (cap < 0 || cap > kFitSmallCap) return fail(h, DSM_E_INVALID, "fit_small_cap %d out of range [0, %d]", cap, kFitSmallCap);
    if (h->frames_submitted != 0) return fail(h, DSM_E_STATE, "dsm_debug_set_fit_small_cap: the handle has already fused frames");
    if (h->batches_joined != 0) return fail(h, DSM_E_STATE, "dsm_debug_set_fit_small_cap: the handle has already joined a batch (which holds a copy of its context)");
    int rc = bind_device(h);
    if (rc) return rc;
    for (int p = 0; p < h->n_pipe; p++) h->pipe[p].ctx.fit_small_cap = cap;
    h->hc.fit_small_cap = cap;
    if (h->d_pipe_ctxs) {
        DeviceCtx tmp[kMaxPipes];
        for (int p = 0; p < h->n_pipe; p++) tmp[p] = h->pipe[p].ctx;
        HIP_TRY(h, hipMemcpy(h->d_pipe_ctxs, tmp, sizeof(DeviceCtx) * (size_t)h->n_pipe, hipMemcpyHostToDevice));
    }
    return DSM_OK;
}

// debug tap: the drop-in calls' delta downloads so far
int dsm_debug_dropin_stats(dsm_handle *h, int64_t *out /* 8 */) {
    if (!h || !out) return DSM_E_INVALID;
    out[0] = h->dropin_calls; out[1] = h->dropin_delta_calls; out[2] = h->dropin_delta_groups; out[3] = h->dropin_last_groups;
    for (int i = 0; i < 4; i++) out[4 + i] = (int64_t)h->dropin_us[i];
    return DSM_OK;
}

// debug tap: how many seeds the latest frame's lane-per-seed kernels handed on to their second tiers
int dsm_debug_tier_counts(dsm_handle *h, int32_t *out /* 8 */) {
    if (!h || !out) return DSM_E_INVALID;
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->hc.rest_count, 6 * sizeof(int32_t), hipMemcpyDeviceToHost));
    HIP_TRY(h, hipMemcpy(out + 6, h->hc.fit_big_count, sizeof(int32_t), hipMemcpyDeviceToHost));
    out[7] = 0;
    return DSM_OK;
}

// debug tap: per-wave phase stamps of the per-seed kernels (only with DSM_FLAG_WAVE_STAMPS)
int dsm_debug_wave_stamps(dsm_handle *h, int64_t *out /* 5 * n_seed * 8 */) {
    if (!h || !out) return DSM_E_INVALID;
    if (!kWaveStamps) return fail(h, DSM_E_STATE, "this library was built without phase stamps (-DDSM_WAVE_STAMPS=1: tools/wave_stamps.py builds such a copy)");
    if (!h->hc.stamps) return fail(h, DSM_E_STATE, "handle was created without DSM_FLAG_WAVE_STAMPS");
    int rc = bind_device(h);
    if (rc) return rc;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(out, h->hc.stamps, (size_t)5 * h->hc.n_seed * 8 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return DSM_OK;
}

} // extern "C"

// ------------------------------------------------------------------ batches
// Streams for batches, one per hardware queue.  HIP multiplexes its streams over four hardware queues (GPU_MAX_HW_QUEUES)
// and gives a new stream the queue with the fewest streams on it at that moment: a batch stream created after a few
// hundred handle streams lands wherever the count happens to be lowest, and two batches that share a queue run one
// after the other (measured: streams 129..132 of a 128-handle bench on queues 4, 2, 1, 4 -- the two batches on queue 4
// busy 45 % and 54 % of the time, the other two 78 % and 73 %).  The first four streams of a process, created together
// before anything else, get a queue each; batches take them in turn.  Created by the first dsm_create on the device.
namespace {
struct BatchStreamPool {
    std::mutex mu;
    hipStream_t st[64][kBatchStreams] = {};
    bool made[64] = {};
    int next[64] = {};
    hipStream_t up[64][2 * kUploadStreams] = {}; // asynchronous frame uploads, created at the first dsm_frame_upload_async on the device: [0, kUploadStreams) at normal priority, the rest at the highest
    int up_next[64][2] = {};
} g_batch_streams;

// (The reserved streams live as long as the process, like the HIP context they belong to: a static destructor would run
// after the runtime's own teardown.)
hipError_t batch_streams_reserve(int device) {
    if (device < 0 || device >= 64) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_batch_streams.mu);
    if (g_batch_streams.made[device]) return hipSuccess;
    // "A queue each" holds only while the queues carry equally many streams when the four are created.  A host that has used
    // the device before the first dsm_create -- torch.cuda.set_device, torch.distributed's eager RCCL communicator, any stream
    // made and destroyed -- leaves them uneven, and the fewest-streams rule then puts two of the four on one queue (rocprofv3
    // queue ids of bench.py's four batches: queues 2, 3, 4, 4, queue 1 left to the host's own stream).  Measured on one box,
    // alternating: the headline 41.2 k frames/s like that against 45.9 k with a queue each; one streamed sequence with an RCCL
    // group open 12.7-13.5 k against 15.3-17.9 k (round 6, profiles/r06_queue_levelling.md).  So the pool is levelled first:
    // ballast streams fill the emptier queues up (the same rule, working for us), the four are created on the level pool, the
    // ballast goes again -- the four stay where they are.
    constexpr int kBallast = 24;
    hipStream_t ballast[kBallast] = {};
    for (int i = 0; i < kBallast; i++)
        if (hipStreamCreateWithFlags(&ballast[i], hipStreamNonBlocking) != hipSuccess) { ballast[i] = nullptr; break; }
    struct Drop {
        hipStream_t *b;
        ~Drop() { for (int i = 0; i < kBallast; i++) if (b[i]) (void)hipStreamDestroy(b[i]); }
    } drop{ballast};
    for (int i = 0; i < kBatchStreams; i++) {
        const hipError_t e = hipStreamCreateWithFlags(&g_batch_streams.st[device][i], hipStreamNonBlocking);
        if (e != hipSuccess) { // all or nothing: the next dsm_create tries again
            for (int j = 0; j < i; j++) { (void)hipStreamDestroy(g_batch_streams.st[device][j]); g_batch_streams.st[device][j] = nullptr; }
            g_batch_streams.st[device][i] = nullptr;
            return e;
        }
    }
    g_batch_streams.made[device] = true;
    return hipSuccess;
}
hipStream_t batch_stream_at(int device, int i) {
    if (device < 0 || device >= 64 || i < 0 || i >= kBatchStreams) return nullptr;
    std::lock_guard<std::mutex> lk(g_batch_streams.mu);
    return g_batch_streams.made[device] ? g_batch_streams.st[device][i] : nullptr;
}
// TWO upload streams per device and priority, shared by its handles (a stream per handle would spread the handles' own
// streams unevenly over the hardware queues, see DSM_FLAG_UPLOAD_STREAM in dsm.h): the copies are DMA transfers ordered by
// events, they need no queue of their own per subsequence -- but one stream's transfers run one after the other on one
// DMA engine (36.7 GB/s of 1226x370 frames on one box); two keep two engines busy.  *which < 0: deal the next one out.
// high_priority (the handles of batches): streams at the highest stream priority.  The runtime keeps a pool of hardware
// queues per priority, so these do not share a hardware queue with any kernel-launching stream of the process -- a copy that
// waits for its slots to be free is a barrier packet, and a barrier packet holds up whatever else is multiplexed onto its
// queue.  Four batches streaming their frames: 21.5-21.7 k frames/s against 15.0-15.4 k with normal-priority upload streams
// on the same box (51 of the 47-57 GB/s the link delivers alone; three alternating rounds, profiles/r05_streaming.md).  One
// handle streaming alone is the other way round (11.5 k against 12.9 k): it keeps the normal-priority streams.
hipStream_t device_upload_stream(int device, int *which, bool high_priority) {
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_batch_streams.mu);
    if (*which < 0) *which = (high_priority ? kUploadStreams : 0) + g_batch_streams.up_next[device][high_priority ? 1 : 0]++ % kUploadStreams;
    hipStream_t &st = g_batch_streams.up[device][*which];
    if (!st) {
        int least = 0, greatest = 0;
        hipError_t e = hipSuccess;
        if (*which >= kUploadStreams && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, greatest);
        else
            e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e != hipSuccess) st = nullptr;
    }
    return st;
}
hipStream_t batch_stream_take(int device) {
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(g_batch_streams.mu);
    if (!g_batch_streams.made[device]) return nullptr;
    const int i = g_batch_streams.next[device]++ % kBatchStreams;
    return g_batch_streams.st[device][i];
}
} // namespace

struct dsm_batch {
    std::vector<dsm_handle *> hs;
    std::vector<uint64_t> gens; // the handles' generations at dsm_batch_create
    std::vector<hipGraphExec_t> retired; // graphs replaced while possibly in flight (batch_map_grows)
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false; // false: one of the device's reserved batch streams
    DeviceCtx *d_ctxs = nullptr;
    hipGraphExec_t graph = nullptr;
    hipEvent_t ev_out = nullptr;
    hipEvent_t ev[kNumStages + 2];
    bool have_events = false;
    int cap_max = 0;
    bool tail_large = false; // a handle's map may exceed k_frame_tail's one-workgroup path (see map_grows)
    std::string err;
};

namespace {
thread_local std::string g_batch_create_error;

int bfail(dsm_batch *b, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (b) b->err = buf; else g_batch_create_error = buf;
    return code;
}
#define BHIP_TRY(b, expr)                                                                           \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return bfail(b, DSM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));  \
    } while (0)

// the handles of a batch advance their parameter rings together; one that was used alone for a while is out of step
int batch_rings_in_step(dsm_batch *b) {
    const int64_t r0 = b->hs[0]->frames_submitted % kParamRing;
    for (size_t j = 1; j < b->hs.size(); j++)
        if (b->hs[j]->frames_submitted % kParamRing != r0)
            return bfail(b, DSM_E_STATE, "handle %zu has fused %lld frames, handle 0 %lld: the parameter rings of a batch must be in step "
                         "(a handle was used alone for a different number of frames); nothing was enqueued", j,
                         (long long)b->hs[j]->frames_submitted, (long long)b->hs[0]->frames_submitted);
    return DSM_OK;
}

// params of frames [i0, i0+m) of every handle go up on the BATCH stream, in order with the kernels that read them.  (Until
// round 5 they went up on the handles' own streams, with an event per handle for the batch stream to wait for: 32 copies and
// 32 markers per enqueue call on streams that HIP multiplexes onto the same four hardware queues the OTHER batches' graphs
// are queued on -- a batch's next chunk could not start before whatever another batch had queued in front of those markers
// had run.)  The batch stream is ordered behind a handle's own stream only when a per-handle call may have put work there
// since the last time (dsm_handle::touched: uploads, a replay of its own, a map download ...).
int batch_stage(dsm_batch *b, int n_frames, int i0, int m, const int32_t *slots, const int32_t *ref_idx, const float *poses16,
                const float *inv_poses16 = nullptr) {
    for (size_t j = 0; j < b->hs.size(); j++) {
        dsm_handle *h = b->hs[j];
        if (!h->map_valid) return bfail(b, DSM_E_STATE, "handle %zu has no resident map: call dsm_map_upload first (n may be 0)", j);
        const size_t o = j * (size_t)n_frames + (size_t)i0;
        if (h->touched) {
            BHIP_TRY(b, hipEventRecord(h->ev_fence, h->stream));
            BHIP_TRY(b, hipStreamWaitEvent(b->stream, h->ev_fence, 0));
            h->touched = false;
        }
        // (a handle that last advanced with ANOTHER batch, and whose own stream has not been used since: behind that batch)
        if (h->batch_order_ev && h->batch_order_ev != b->ev_out) BHIP_TRY(b, hipStreamWaitEvent(b->stream, h->batch_order_ev, 0));
        int staged = 0;
        const int rc = stage_params_batch(h, m, slots + o, ref_idx + o, poses16 + 16 * o, inv_poses16 ? inv_poses16 + 16 * o : nullptr, &staged,
                                          b->stream);
        if (rc) return bfail(b, rc, "handle %zu: %s", j, h->err.c_str());
        if (staged != m) return bfail(b, DSM_E_STATE, "handle %zu: parameter rings of the batch are out of step", j);
        if (h->up_n) { // frames this handle was sent with dsm_frame(s)_upload_async: the uploads that wrote the slots these frames read
            int r_lo, r_hi;
            slot_range(slots + o, m, &r_lo, &r_hi);
            const ReadSlots reads(h, r_lo, r_hi);
            if (wait_uploads(h, b->stream, kBatchBit)) return bfail(b, DSM_E_HIP, "handle %zu: %s", j, h->err.c_str());
        }
    }
    return DSM_OK;
}
// the handles' bookkeeping after m frames were enqueued on the batch stream; their streams wait for the batch
// map_grows for a batch: its one graph holds the tails of all its handles
int batch_map_grows(dsm_batch *b, int m) {
    if (b->tail_large) return DSM_OK;
    bool large = false;
    for (const dsm_handle *h : b->hs) large = large || (int64_t)h->map_upper + (int64_t)(m - 1) * h->hc.n_seed > (int64_t)kTailFastWords * 64;
    if (!large) return DSM_OK;
    b->tail_large = true;
    if (b->graph) { // (possibly in flight: set aside without a wait, destroyed at the batch's next synchronisation point)
        b->retired.push_back(b->graph);
        b->graph = nullptr;
    }
    return DSM_OK;
}
int batch_advance(dsm_batch *b, int m) {
    BHIP_TRY(b, hipEventRecord(b->ev_out, b->stream));
    for (dsm_handle *h : b->hs) {
        h->batch_order_ev = b->ev_out; // (waited for when the handle's stream is next used)
        h->shadow_n = -1;
        h->dirty_flags_clean = false;
        h->reads_untracked = true;
        h->frames_submitted += m;
        const int64_t up = (int64_t)h->map_upper + (int64_t)m * h->hc.n_seed;
        h->map_upper = up > h->hc.cap ? h->hc.cap : (int)up;
        h->fence_pending = true;
    }
    return DSM_OK;
}
} // namespace

extern "C" {

const char *dsm_batch_last_error(const dsm_batch *b) { return b ? b->err.c_str() : g_batch_create_error.c_str(); }

int dsm_batch_create(dsm_handle *const *handles, int32_t n, dsm_batch **out) {
    if (!handles || !out || n < 1 || n > 1024) return bfail(nullptr, DSM_E_INVALID, "bad batch arguments");
    *out = nullptr;
    for (int j = 0; j < n; j++) {
        const dsm_handle *h = handles[j];
        if (!h) return bfail(nullptr, DSM_E_INVALID, "null handle");
        if (h->n_pipe != 1) return bfail(nullptr, DSM_E_INVALID, "handles of a batch need pipeline_depth 1");
        if (h->own_up_stream) return bfail(nullptr, DSM_E_INVALID, "handles of a batch must not use DSM_FLAG_UPLOAD_STREAM (a batch does not wait on a handle's upload stream)");
        if (h->device != handles[0]->device || h->hc.w != handles[0]->hc.w || h->hc.h != handles[0]->hc.h)
            return bfail(nullptr, DSM_E_INVALID, "handles of a batch must share device and image size");
        if (h->frames_submitted % kParamRing != handles[0]->frames_submitted % kParamRing)
            return bfail(nullptr, DSM_E_STATE, "handles of a batch must have fused the same number of frames (their parameter rings advance together)");
    }
    dsm_batch *b = new (std::nothrow) dsm_batch();
    if (!b) return bfail(nullptr, DSM_E_HIP, "out of host memory");
    b->hs.assign(handles, handles + n);
    for (int j = 0; j < n; j++) b->gens.push_back(handles[j]->generation);
    for (dsm_handle *h : b->hs) h->batches_joined++;
    b->device = handles[0]->device;
    auto bail = [&](int code) {
        g_batch_create_error = b->err;
        dsm_batch_destroy(b);
        return code;
    };
#define BCREATE_TRY(expr)                                                       \
    do {                                                                        \
        hipError_t e_ = (expr);                                                 \
        if (e_ != hipSuccess) {                                                 \
            bfail(b, DSM_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));        \
            return bail(DSM_E_HIP);                                             \
        }                                                                       \
    } while (0)
    BCREATE_TRY(hipSetDevice(b->device));
    b->stream = batch_stream_take(b->device);
    if (!b->stream) {
        BCREATE_TRY(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
        b->own_stream = true;
    }
    BCREATE_TRY(hipEventCreateWithFlags(&b->ev_out, hipEventDisableTiming));
    for (int i = 0; i <= kNumStages + 1; i++) BCREATE_TRY(hipEventCreate(&b->ev[i]));
    b->have_events = true;
    std::vector<DeviceCtx> ctxs((size_t)n);
    for (int j = 0; j < n; j++) {
        ctxs[(size_t)j] = handles[j]->pipe[0].ctx;
        b->cap_max = std::max(b->cap_max, handles[j]->hc.cap);
    }
    BCREATE_TRY(hipMalloc((void **)&b->d_ctxs, sizeof(DeviceCtx) * (size_t)n));
    BCREATE_TRY(hipMemcpy(b->d_ctxs, ctxs.data(), sizeof(DeviceCtx) * (size_t)n, hipMemcpyHostToDevice));
#undef BCREATE_TRY
    *out = b;
    return DSM_OK;
}

void dsm_batch_destroy(dsm_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    { // the batch's copy of the handles' contexts goes with it (a handle destroyed before its batch is simply skipped)
        std::lock_guard<std::mutex> lk(g_live_mu);
        for (size_t j = 0; j < b->hs.size(); j++) {
            dsm_handle *h = b->hs[j];
            const auto it = g_live.find(h);
            if (it == g_live.end() || j >= b->gens.size() || it->second != b->gens[j]) continue; // destroyed (its address may belong to a newer handle by now)
            if (h->batches_joined > 0) h->batches_joined--;
            if (h->batch_order_ev == b->ev_out) h->batch_order_ev = nullptr; // (the batch has finished: nothing left to wait for)
        }
    }
    if (b->graph) (void)hipGraphExecDestroy(b->graph);
    for (hipGraphExec_t g : b->retired) (void)hipGraphExecDestroy(g); // (the batch's stream was waited for above)
    if (b->ev_out) (void)hipEventDestroy(b->ev_out);
    if (b->have_events)
        for (int i = 0; i <= kNumStages + 1; i++) (void)hipEventDestroy(b->ev[i]);
    if (b->d_ctxs) (void)hipFree(b->d_ctxs);
    if (b->stream && b->own_stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int dsm_batch_replay_enqueue(dsm_batch *b, int32_t n_frames, const int32_t *slots, const int32_t *ref_idx, const float *poses16) {
    return dsm_batch_replay_enqueue_inv(b, n_frames, slots, ref_idx, poses16, nullptr);
}

int dsm_batch_replay_enqueue_inv(dsm_batch *b, int32_t n_frames, const int32_t *slots, const int32_t *ref_idx, const float *poses16,
                                 const float *inv_poses16) {
    if (!b) return DSM_E_INVALID;
    if (n_frames < 0 || (n_frames > 0 && (!slots || !ref_idx || !poses16))) return bfail(b, DSM_E_INVALID, "null/negative argument");
    BHIP_TRY(b, hipSetDevice(b->device));
    if (int rc = batch_rings_in_step(b)) return rc;
    dsm_handle *h0 = b->hs[0];
    for (int i = 0; i < n_frames;) {
        int m = n_frames - i;
        const int ring = (int)(h0->frames_submitted % kParamRing);
        if (m > kParamRing - ring) m = kParamRing - ring;
        if (m > kParamRing / 2) m = kParamRing / 2;
        int rc = batch_stage(b, n_frames, i, m, slots, ref_idx, poses16, inv_poses16);
        if (rc) return rc;
        if ((rc = batch_map_grows(b, m))) return rc;
        if (!b->graph) {
            const std::string err = capture_graph([&](hipStream_t st) {
                return launch_frame(h0->pipe[0].ctx, b->cap_max, b->tail_large ? b->cap_max : 0, true, st, nullptr, 0, kNumStages - 1, b->d_ctxs, (int)b->hs.size());
            }, &b->graph);
            if (!err.empty()) return bfail(b, DSM_E_HIP, "%s", err.c_str());
        }
        for (int f = 0; f < m; f++) BHIP_TRY(b, hipGraphLaunch(b->graph, b->stream));
        if ((rc = batch_advance(b, m))) return rc;
        i += m;
    }
    return DSM_OK;
}

int dsm_batch_synchronize(dsm_batch *b) {
    if (!b) return DSM_E_INVALID;
    BHIP_TRY(b, hipSetDevice(b->device));
    for (size_t j = 0; j < b->hs.size(); j++) {
        int rc = order_behind_batch(b->hs[j]); // (the handles' streams come behind the batch when they are next used: now)
        if (!rc) rc = sync_and_fetch_counts(b->hs[j]);
        if (rc) return bfail(b, rc, "handle %zu: %s", j, b->hs[j]->err.c_str());
    }
    // (the handles' streams came behind the batch's and have been waited for: a graph set aside by batch_map_grows is done)
    for (hipGraphExec_t g : b->retired) (void)hipGraphExecDestroy(g);
    b->retired.clear();
    return DSM_OK;
}

int dsm_batch_replay_timed(dsm_batch *b, int32_t n_frames, const int32_t *slots, const int32_t *ref_idx, const float *poses16,
                           dsm_stage_times *out) {
    if (!b || !out) return DSM_E_INVALID;
    if (n_frames < 0 || (n_frames > 0 && (!slots || !ref_idx || !poses16))) return bfail(b, DSM_E_INVALID, "null/negative argument");
    BHIP_TRY(b, hipSetDevice(b->device));
    if (int rc = batch_rings_in_step(b)) return rc;
    out->n_stages = kNumStages;
    for (int s = 0; s < kNumStages; s++) out->name[s] = kStageNames[s];
    dsm_handle *h0 = b->hs[0];
    const int n = (int)b->hs.size();
    for (int i = 0; i < n_frames; i++) {
        int rc = batch_stage(b, n_frames, i, 1, slots, ref_idx, poses16);
        if (rc) return rc;
        if ((rc = batch_map_grows(b, 1))) return rc;
        const hipError_t e = launch_frame(h0->pipe[0].ctx, b->cap_max, b->tail_large ? b->cap_max : 0, true, b->stream, b->ev, 0, kNumStages - 1, b->d_ctxs, n);
        if (e != hipSuccess) return bfail(b, DSM_E_HIP, "kernel launch: %s", hipGetErrorString(e));
        if ((rc = batch_advance(b, 1))) return rc;
        if ((rc = dsm_batch_synchronize(b))) return rc;
        for (int s = 0; s < kNumStages; s++) {
            float ms = 0.0f;
            BHIP_TRY(b, hipEventElapsedTime(&ms, b->ev[s], b->ev[s + 1]));
            out->ms[s] += (double)ms;
            out->launches[s] += 1;
        }
        float cal = 0.0f;
        BHIP_TRY(b, hipEventElapsedTime(&cal, b->ev[kNumStages], b->ev[kNumStages + 1]));
        out->event_overhead_ms += (double)cal * n; // (per handle-frame, like `frames`)
        out->frames += n;
        for (dsm_handle *h : b->hs) {
            out->sum_new += h->h_scalars[1];
            out->sum_local += h->h_scalars[0];
        }
    }
    return DSM_OK;
}

} // extern "C"

// ------------------------------------------------------------------ per-kernel timing
extern "C" {

int dsm_replay_timed(dsm_handle *h, int32_t n, const int32_t *slots, const int32_t *ref_idx, const float *poses16,
                     dsm_stage_times *out) {
    if (!h || !out) return DSM_E_INVALID;
    if (n < 0 || (n > 0 && (!slots || !ref_idx || !poses16))) return fail(h, DSM_E_INVALID, "null/negative argument");
    if (!h->map_valid) return fail(h, DSM_E_STATE, "no resident map: call dsm_map_upload first (n may be 0)");
    int rc = bind_device(h);
    if (rc) return rc;
    out->n_stages = kNumStages;
    for (int s = 0; s < kNumStages; s++) {
        out->name[s] = kStageNames[s];
    }
    for (int i = 0; i < n; i++) {
        if ((rc = stage_params(h, slots[i], ref_idx[i], poses16 + 16 * (size_t)i))) return rc;
        if ((rc = submit_serial(h, true, h->ev, 0, kNumStages - 1))) return rc;
        if ((rc = sync_and_fetch_counts(h))) return rc;
        for (int s = 0; s < kNumStages; s++) {
            float ms = 0.0f;
            HIP_TRY(h, hipEventElapsedTime(&ms, h->ev[s], h->ev[s + 1]));
            out->ms[s] += (double)ms;
            out->launches[s] += 1;
        }
        float cal = 0.0f;
        HIP_TRY(h, hipEventElapsedTime(&cal, h->ev[kNumStages], h->ev[kNumStages + 1]));
        out->event_overhead_ms += (double)cal;
        out->frames += 1;
        out->sum_new += h->h_scalars[1];
        out->sum_local += h->h_scalars[0];
    }
    return DSM_OK;
}

} // extern "C"
