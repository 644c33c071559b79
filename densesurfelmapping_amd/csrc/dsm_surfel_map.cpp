// dsm_surfel_map.cpp -- host logic of the `surfel_fusion` node above the engine's C ABI
// (include/dsm_surfel_map.h).  Follows class SurfelMap of the reference (surfel_fusion/src/surfel_map.cpp;
// line numbers below refer to it) statement by statement where state is concerned; surfel data stay on
// the device: local_surfels = the handle's resident map, attached_surfels + inactive_pointcloud = the
// handle's inactive store.
//
// Pose arithmetic is fp64 on the host as in the reference.  Eigen3 (un-vendored, un-pinned there) is
// absent from this image; the few operations the node takes from it are written out from their published
// definitions: 4x4 product with left-to-right accumulation, 4x4 inverse by the adjugate closed form,
// Quaterniond <-> rotation matrix (Eigen/src/Geometry/Quaternion.h), Vector3f normalize / cross.
#include "../../include/dsm_surfel_map.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <list>
#include <locale>
#include <set>
#include <sstream>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

namespace {

// ------------------------------------------------------------------ fp64 rigid-transform helpers
struct Mat4 {
    double d[16]; // column-major, d[j*4+i] = (i,j)
    double &operator()(int i, int j) { return d[j * 4 + i]; }
    double operator()(int i, int j) const { return d[j * 4 + i]; }
};

Mat4 identity4() {
    Mat4 m;
    for (int k = 0; k < 16; k++) m.d[k] = 0.0;
    m(0, 0) = m(1, 1) = m(2, 2) = m(3, 3) = 1.0;
    return m;
}

Mat4 mul(const Mat4 &a, const Mat4 &b) {
    Mat4 c;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++) c(i, j) = ((a(i, 0) * b(0, j) + a(i, 1) * b(1, j)) + a(i, 2) * b(2, j)) + a(i, 3) * b(3, j);
    return c;
}

// adjugate / determinant from the twelve 2x2 minors of the top and bottom row pairs
Mat4 inverse(const Mat4 &a) {
    const double s0 = a(0, 0) * a(1, 1) - a(1, 0) * a(0, 1), s1 = a(0, 0) * a(1, 2) - a(1, 0) * a(0, 2);
    const double s2 = a(0, 0) * a(1, 3) - a(1, 0) * a(0, 3), s3 = a(0, 1) * a(1, 2) - a(1, 1) * a(0, 2);
    const double s4 = a(0, 1) * a(1, 3) - a(1, 1) * a(0, 3), s5 = a(0, 2) * a(1, 3) - a(1, 2) * a(0, 3);
    const double c5 = a(2, 2) * a(3, 3) - a(3, 2) * a(2, 3), c4 = a(2, 1) * a(3, 3) - a(3, 1) * a(2, 3);
    const double c3 = a(2, 1) * a(3, 2) - a(3, 1) * a(2, 2), c2 = a(2, 0) * a(3, 3) - a(3, 0) * a(2, 3);
    const double c1 = a(2, 0) * a(3, 2) - a(3, 0) * a(2, 2), c0 = a(2, 0) * a(3, 1) - a(3, 0) * a(2, 1);
    const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
    const double r = 1.0 / det;
    Mat4 b;
    b(0, 0) = (a(1, 1) * c5 - a(1, 2) * c4 + a(1, 3) * c3) * r;
    b(0, 1) = (-a(0, 1) * c5 + a(0, 2) * c4 - a(0, 3) * c3) * r;
    b(0, 2) = (a(3, 1) * s5 - a(3, 2) * s4 + a(3, 3) * s3) * r;
    b(0, 3) = (-a(2, 1) * s5 + a(2, 2) * s4 - a(2, 3) * s3) * r;
    b(1, 0) = (-a(1, 0) * c5 + a(1, 2) * c2 - a(1, 3) * c1) * r;
    b(1, 1) = (a(0, 0) * c5 - a(0, 2) * c2 + a(0, 3) * c1) * r;
    b(1, 2) = (-a(3, 0) * s5 + a(3, 2) * s2 - a(3, 3) * s1) * r;
    b(1, 3) = (a(2, 0) * s5 - a(2, 2) * s2 + a(2, 3) * s1) * r;
    b(2, 0) = (a(1, 0) * c4 - a(1, 1) * c2 + a(1, 3) * c0) * r;
    b(2, 1) = (-a(0, 0) * c4 + a(0, 1) * c2 - a(0, 3) * c0) * r;
    b(2, 2) = (a(3, 0) * s4 - a(3, 1) * s2 + a(3, 3) * s0) * r;
    b(2, 3) = (-a(2, 0) * s4 + a(2, 1) * s2 - a(2, 3) * s0) * r;
    b(3, 0) = (-a(1, 0) * c3 + a(1, 1) * c1 - a(1, 2) * c0) * r;
    b(3, 1) = (a(0, 0) * c3 - a(0, 1) * c1 + a(0, 2) * c0) * r;
    b(3, 2) = (-a(3, 0) * s3 + a(3, 1) * s1 - a(3, 2) * s0) * r;
    b(3, 3) = (a(2, 0) * s3 - a(2, 1) * s1 + a(2, 2) * s0) * r;
    return b;
}

// SurfelMap::pose_ros2eigen (:367-379): Quaterniond::toRotationMatrix + translation
Mat4 pose_to_matrix(const dsm_pose_msg &p) {
    Mat4 t = identity4();
    const double x = p.qx, y = p.qy, z = p.qz, w = p.qw;
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    t(0, 0) = 1.0 - (tyy + tzz);
    t(0, 1) = txy - twz;
    t(0, 2) = txz + twy;
    t(1, 0) = txy + twz;
    t(1, 1) = 1.0 - (txx + tzz);
    t(1, 2) = tyz - twx;
    t(2, 0) = txz - twy;
    t(2, 1) = tyz + twx;
    t(2, 2) = 1.0 - (txx + tyy);
    t(0, 3) = p.px;
    t(1, 3) = p.py;
    t(2, 3) = p.pz;
    return t;
}

// SurfelMap::pose_eigen2ros (:381-391): Quaterniond(Matrix3d) -- trace branch, else largest diagonal
dsm_pose_msg matrix_to_pose(const Mat4 &m) {
    double q[4]; // x y z w
    double t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[3] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (m(2, 1) - m(1, 2)) * t;
        q[1] = (m(0, 2) - m(2, 0)) * t;
        q[2] = (m(1, 0) - m(0, 1)) * t;
    } else {
        int i = 0;
        if (m(1, 1) > m(0, 0)) i = 1;
        if (m(2, 2) > m(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0);
        q[i] = 0.5 * t;
        t = 0.5 / t;
        q[3] = (m(k, j) - m(j, k)) * t;
        q[j] = (m(j, i) + m(i, j)) * t;
        q[k] = (m(k, i) + m(i, k)) * t;
    }
    dsm_pose_msg p;
    p.qx = q[0]; p.qy = q[1]; p.qz = q[2]; p.qw = q[3];
    p.px = m(0, 3); p.py = m(1, 3); p.pz = m(2, 3);
    return p;
}

void to_float16(const Mat4 &m, float *out) { // .cast<float>(), column-major like Eigen::Matrix4f::data()
    for (int k = 0; k < 16; k++) out[k] = (float)m.d[k];
}

bool same_position(const dsm_pose_msg &a, const dsm_pose_msg &b) { return a.px == b.px && a.py == b.py && a.pz == b.pz; }

double to_sec(dsm_stamp s) { return (double)s.sec + 1e-9 * (double)s.nsec; }

// ------------------------------------------------------------------ node state
struct PoseElement { // surfel_map.h:36-46; attached_surfels live in the handle's store
    dsm_pose_msg cam_pose, loop_pose;
    std::vector<int> linked_pose_index;
    int segment = -1; // index into dsm_surfel_map::segments while the keyframe is inactive, else -1
    dsm_stamp cam_stamp = {0, 0};
};

// The inactive set as a segment table.  The handle's store holds the surfels of the inactive keyframes back to
// back in deactivation order; entry i of the table says which keyframe owns the i-th run and how long it is, and a
// run starts where the runs before it end.  This one table is what the reference spreads over three members:
// PoseElement::attached_surfels.size() (count), PoseElement::points_begin_index (start) and
// pointcloud_pose_index / PoseElement::points_pose_index (the table order and its inverse), surfel_map.h:36-46,134.
struct Segment {
    int keyframe;
    int begin; // sum of the counts before this entry (kept, not recomputed: the taps read it)
    int count;
};

struct Frame {
    dsm_stamp stamp;
    uint8_t *bytes; // tightly packed rows: a page-locked block of the node's pool, or (overflow) pageable memory
    bool pinned;
};

// Frames wait for their pose in page-locked memory so that the upload of a frame is one DMA -- but only the first
// kPinnedFrames of each kind: the reference's subscriber queues are 5000 deep (ros_node.cpp:24-25) in PAGEABLE memory, and
// a stalled pose source must not pin 5000 x 2.3 MB of host RAM.  The overflow lives in pageable blocks (their upload is a
// staged copy: slower, still correct) that are freed as soon as they leave the queue; free page-locked blocks beyond
// kPooledFrames go back to the system as well.
constexpr size_t kPinnedFrames = 256, kPooledFrames = 64;

struct FramePool {
    std::vector<uint8_t *> free_blocks; // page-locked, ready for reuse
    size_t pinned_live = 0;             // page-locked blocks handed out and not yet released
    uint8_t *take(size_t bytes, bool *pinned) {
        if (!free_blocks.empty()) {
            uint8_t *p = free_blocks.back();
            free_blocks.pop_back();
            pinned_live++;
            *pinned = true;
            return p;
        }
        if (pinned_live < kPinnedFrames) {
            void *p = nullptr;
            if (dsm_host_alloc(&p, bytes) == DSM_OK) {
                pinned_live++;
                *pinned = true;
                return (uint8_t *)p;
            }
        }
        *pinned = false;
        return (uint8_t *)malloc(bytes ? bytes : 1);
    }
    void release(const Frame &f) {
        if (!f.pinned) { free(f.bytes); return; }
        pinned_live--;
        if (free_blocks.size() < kPooledFrames) free_blocks.push_back(f.bytes);
        else dsm_host_free(f.bytes);
    }
    void drain() {
        for (uint8_t *p : free_blocks) dsm_host_free(p);
        free_blocks.clear();
    }
};

} // namespace

struct dsm_surfel_map {
    dsm_surfel_map_config cfg;
    dsm_handle *engine = nullptr;
    std::list<Frame> image_buffer, depth_buffer;                                 // surfel_map.h:96-97
    FramePool image_pool, depth_pool;                                            // where the buffered frames' bytes live
    std::list<std::tuple<dsm_stamp, dsm_pose_msg, int>> pose_reference_buffer; // :98
    std::vector<PoseElement> poses_database;                                     // :120
    std::set<int> local_surfels_indexs;                                          // :122
    std::vector<Segment> segments;                                               // inactive set, store order (:134)
    int64_t poses_dropped = 0, frames_dropped = 0;
    bool failed = false; // an engine call failed half-way through a state change: refuse further input
    Mat4 transform_kitti = identity4();                                          // function-static at surfel_map.cpp:215
    int64_t frames_fused = 0;
    std::string err;
};

namespace {

int fail(dsm_surfel_map *m, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m) m->err = buf;
    return code;
}

int engine_fail(dsm_surfel_map *m, int rc, const char *what) { return fail(m, rc, "%s: %s", what, dsm_last_error(m->engine)); }

#define ENGINE_TRY(m, expr)                                  \
    do {                                                     \
        const int rc_ = (expr);                              \
        if (rc_ != DSM_OK) return engine_fail(m, rc_, #expr); \
    } while (0)

// SurfelMap::get_driftfree_poses (:1643-1673): breadth-first over linked_pose_index, root first,
// driftfree_range - 1 levels, each pose once in discovery order
void get_driftfree_poses(const dsm_surfel_map *m, int root_index, std::vector<int> &driftfree_poses, int driftfree_range) {
    if ((int)m->poses_database.size() < root_index + 1) return;
    std::vector<int> this_level, next_level;
    this_level.push_back(root_index);
    driftfree_poses.push_back(root_index);
    for (int i = 1; i < driftfree_range; i++) {
        for (int p : this_level)
            for (int linked : m->poses_database[p].linked_pose_index)
                if (std::find(driftfree_poses.begin(), driftfree_poses.end(), linked) == driftfree_poses.end()) {
                    next_level.push_back(linked);
                    driftfree_poses.push_back(linked);
                }
        this_level.swap(next_level);
        next_level.clear();
    }
}

// SurfelMap::get_add_remove_poses (:1597-1641)
void get_add_remove_poses(const dsm_surfel_map *m, int root_index, std::vector<int> &pose_to_add, std::vector<int> &pose_to_remove) {
    std::vector<int> driftfree_poses;
    get_driftfree_poses(m, root_index, driftfree_poses, m->cfg.drift_free_poses);
    pose_to_add.clear();
    pose_to_remove.clear();
    for (int p : driftfree_poses)
        if (m->local_surfels_indexs.find(p) == m->local_surfels_indexs.end()) pose_to_add.push_back(p);
    for (int p : m->local_surfels_indexs)
        if (std::find(driftfree_poses.begin(), driftfree_poses.end(), p) == driftfree_poses.end()) pose_to_remove.push_back(p);
}

// SurfelMap::move_add_surfels (:1456-1595): keyframes that left the drift-free window hand their surfels to the
// inactive set, keyframes that entered it get theirs back.
int move_add_surfels(dsm_surfel_map *m, int reference_index) {
    std::vector<int> poses_to_add, poses_to_remove;
    get_add_remove_poses(m, reference_index, poses_to_add, poses_to_remove);

    // Check before anything is changed: the returning surfels must fit (deactivation only frees slots, so the
    // bound holds whatever it removes).  After this point an engine failure leaves map, store and table out of
    // step, and the node refuses further input (`failed`).
    if (!poses_to_add.empty()) {
        int64_t returning = 0;
        for (int k : poses_to_add) {
            const int sg = m->poses_database[(size_t)k].segment;
            if (sg < 0) return fail(m, DSM_E_STATE, "keyframe %d is neither local nor in the inactive set", k);
            returning += m->segments[(size_t)sg].count;
        }
        int32_t live = 0, cap = 0;
        ENGINE_TRY(m, dsm_map_size(m->engine, &live));
        ENGINE_TRY(m, dsm_map_capacity(m->engine, &cap));
        if ((int64_t)live + returning > (int64_t)cap)
            return fail(m, DSM_E_CAPACITY, "%lld returning surfels do not fit: %d of %d slots in use", (long long)returning, live, cap);
    }
#define CHANGE_TRY(expr)                                          \
    do {                                                          \
        const int rc_ = (expr);                                   \
        if (rc_ != DSM_OK) {                                      \
            m->failed = true;                                     \
            return engine_fail(m, rc_, #expr);                    \
        }                                                         \
    } while (0)

    // leaving keyframes, ascending (:1467-1503): each becomes the last segment
    for (int k : poses_to_remove) {
        Segment sg;
        sg.keyframe = k;
        CHANGE_TRY(dsm_store_deactivate(m->engine, k, &sg.begin, &sg.count)); // begin == store size before the move
        m->poses_database[(size_t)k].segment = (int)m->segments.size();
        m->segments.push_back(sg);
        m->local_surfels_indexs.erase(k);
    }

    if (!poses_to_add.empty()) {
        m->local_surfels_indexs.insert(poses_to_add.begin(), poses_to_add.end());
        // their surfels go to the end of the active map in the order of poses_to_add (:1583-1590) ...
        for (int k : poses_to_add) {
            const Segment &sg = m->segments[(size_t)m->poses_database[(size_t)k].segment];
            if (sg.count) CHANGE_TRY(dsm_store_activate(m->engine, sg.begin, sg.count));
        }
        // ... and their segments leave the table (:1511-1579).  Adjacent leaving segments are one erase; runs are
        // taken from the back of the store so that the offsets of the runs still to go stay valid.
        std::vector<char> leaving(m->segments.size(), 0);
        for (int k : poses_to_add) leaving[(size_t)m->poses_database[(size_t)k].segment] = 1;
        for (size_t hi = m->segments.size(); hi > 0;) {
            if (!leaving[hi - 1]) { hi--; continue; }
            size_t lo = hi - 1;
            while (lo > 0 && leaving[lo - 1]) lo--;
            const int begin = m->segments[lo].begin, end = m->segments[hi - 1].begin + m->segments[hi - 1].count;
            if (end > begin) CHANGE_TRY(dsm_store_erase(m->engine, begin, end - begin));
            hi = lo;
        }
        std::vector<Segment> kept;
        int at = 0;
        for (size_t i = 0; i < m->segments.size(); i++) {
            Segment sg = m->segments[i];
            if (leaving[i]) { m->poses_database[(size_t)sg.keyframe].segment = -1; continue; }
            sg.begin = at;
            at += sg.count;
            m->poses_database[(size_t)sg.keyframe].segment = (int)kept.size();
            kept.push_back(sg);
        }
        m->segments.swap(kept);
    }
#undef CHANGE_TRY
    return DSM_OK;
}

// SurfelMap::warp_surfels (:791-824) with its two kernels (:681-789).  The reference starts ten threads
// over the keyframes, computes the active map's matrix from the first local keyframe's cam_pose on the
// main thread while they run, and only then joins; a worker that reaches that keyframe first overwrites
// cam_pose with loop_pose (:698-702).  Here the active matrix is taken before any cam_pose is
// overwritten -- the schedule in which the main thread wins that race.
int warp_surfels(dsm_surfel_map *m) {
    if (m->local_surfels_indexs.empty()) return fail(m, DSM_E_STATE, "no local keyframe to take the active warp from");
    const int local_index = *m->local_surfels_indexs.begin();
    float warp_pose[16];
    to_float16(mul(pose_to_matrix(m->poses_database[local_index].loop_pose), inverse(pose_to_matrix(m->poses_database[local_index].cam_pose))),
               warp_pose);

    // inactive keyframes: one grouped pass over the store, groups in pointcloud_pose_index order
    const int n_groups = (int)m->segments.size();
    std::vector<int32_t> offsets((size_t)n_groups + 1, 0);
    std::vector<float> mats((size_t)n_groups * 16, 0.f);
    std::vector<uint8_t> changed((size_t)n_groups, 0);
    bool any = false;
    for (size_t i = 0; i < m->poses_database.size(); i++) {
        PoseElement &pe = m->poses_database[i];
        if (same_position(pe.cam_pose, pe.loop_pose)) continue; // :691-695
        if (pe.segment >= 0 && m->segments[(size_t)pe.segment].count > 0) {
            const int g = pe.segment;
            to_float16(mul(pose_to_matrix(pe.loop_pose), inverse(pose_to_matrix(pe.cam_pose))), &mats[(size_t)g * 16]); // :706-710
            changed[(size_t)g] = 1;
            any = true;
        }
        pe.cam_pose = pe.loop_pose; // :698-702, :741
    }
    if (any) {
        int32_t total = 0;
        ENGINE_TRY(m, dsm_store_size(m->engine, &total));
        for (int g = 0; g < n_groups; g++) offsets[(size_t)g] = m->segments[(size_t)g].begin;
        offsets[(size_t)n_groups] = total;
        ENGINE_TRY(m, dsm_store_warp(m->engine, n_groups, offsets.data(), mats.data(), changed.data()));
    }
    ENGINE_TRY(m, dsm_map_warp(m->engine, warp_pose)); // :750-789, :815-819
    return DSM_OK;
}

// SurfelMap::synchronize_msgs (:103-203)
int synchronize_msgs(dsm_surfel_map *m) {
    bool find_image = false, find_depth = false;
    // :114-139.  Buffered frames older than the front pose are dropped, an equal stamp is the match.  A front frame
    // NEWER than the pose stamp means the pose's own frame was lost (stamps only grow): the reference's loop then
    // spins forever (neither branch pops or breaks).  Here that pose is dropped and counted
    // (dsm_surfel_map_dropped_poses) and the next one is tried, so one lost message cannot stall the node.
    for (;;) {
        if (m->pose_reference_buffer.empty()) return DSM_OK;
        const double pose_reference_time = to_sec(std::get<0>(m->pose_reference_buffer.front()));
        bool lost = false;
        find_image = find_depth = false;
        while (!m->image_buffer.empty()) {
            const double t = to_sec(m->image_buffer.front().stamp);
            if (t < pose_reference_time) { m->image_pool.release(m->image_buffer.front()); m->image_buffer.pop_front(); }
            else { find_image = t == pose_reference_time; lost |= !find_image; break; }
        }
        while (!m->depth_buffer.empty()) {
            const double t = to_sec(m->depth_buffer.front().stamp);
            if (t < pose_reference_time) { m->depth_pool.release(m->depth_buffer.front()); m->depth_buffer.pop_front(); }
            else { find_depth = t == pose_reference_time; lost |= !find_depth; break; }
        }
        if (!lost) break;
        fprintf(stderr, "dsm_surfel_map: the frame of the pose stamped %.6f was never received or already dropped; pose skipped\n", pose_reference_time);
        m->pose_reference_buffer.pop_front();
        m->poses_dropped++;
    }
    if (!find_image || !find_depth) return DSM_OK; // an empty buffer: the pose waits for its frame

    const dsm_pose_msg relative_pose_ros = std::get<1>(m->pose_reference_buffer.front());
    const int relative_index = std::get<2>(m->pose_reference_buffer.front());
    const Mat4 fuse_pose = mul(pose_to_matrix(m->poses_database[(size_t)relative_index].cam_pose), pose_to_matrix(relative_pose_ros)); // :147-150

    int rc = move_add_surfels(m, relative_index); // :154
    if (rc) return rc;

    // SurfelMap::fuse_map (:1060-1113): engine + order-exact refill / compaction, all on the device
    float pose16[16];
    to_float16(fuse_pose, pose16);
    const int w = m->cfg.cam_width;
    // two frame slots in turn: this frame goes up (on the engine's upload stream) while the previous one is still being fused
    const int slot = (int)(m->frames_fused & 1);
    ENGINE_TRY(m, dsm_frame_upload(m->engine, slot, m->image_buffer.front().bytes, (size_t)w,
                                   (const float *)m->depth_buffer.front().bytes, (size_t)w * 4));
    ENGINE_TRY(m, dsm_fuse_frame_resident(m->engine, slot, relative_index, pose16));
    m->pose_reference_buffer.pop_front(); // :163
    m->frames_fused++;
    return DSM_OK;
}

int copy_frame(dsm_surfel_map *m, std::list<Frame> &buffer, FramePool &pool, dsm_stamp stamp, int32_t width,
               int32_t height, size_t step, const void *data, size_t elem) {
    if (!data) return fail(m, DSM_E_INVALID, "null image data");
    if (width != m->cfg.cam_width || height != m->cfg.cam_height)
        return fail(m, DSM_E_INVALID, "image is %dx%d, the node was configured for %dx%d", width, height, m->cfg.cam_width, m->cfg.cam_height);
    if (step < (size_t)width * elem) return fail(m, DSM_E_INVALID, "row step smaller than a row");
    Frame f;
    f.stamp = stamp;
    f.bytes = pool.take((size_t)width * (size_t)height * elem, &f.pinned);
    if (!f.bytes) return fail(m, DSM_E_HIP, "no host memory for a frame");
    for (int y = 0; y < height; y++) memcpy(f.bytes + (size_t)y * width * elem, (const uint8_t *)data + (size_t)y * step, (size_t)width * elem);
    buffer.push_back(f);
    // frames nobody claims (no pose ever arrives for them) must not pile up: the oldest go (default 5000 = the
    // reference's subscriber queue depth, ros_node.cpp:24-25; a drop is reported, never silent).  Only the first
    // kPinnedFrames of them are page-locked (FramePool).
    const int lim = m->cfg.max_buffered_frames;
    const size_t keep = lim > 0 ? (size_t)lim : lim == 0 ? (size_t)5000 : (size_t)-1;
    while (buffer.size() > keep) {
        fprintf(stderr, "dsm_surfel_map: more than %zu %s frames wait for a pose; dropping the one stamped %.6f (its pose will be skipped)\n",
                keep, elem == 1 ? "image" : "depth", to_sec(buffer.front().stamp));
        m->frames_dropped++;
        pool.release(buffer.front());
        buffer.pop_front();
    }
    return DSM_OK;
}

// SurfelMap::push_a_surfel (:1176-1216): six corners of a hexagon of circumradius `size` in the surfel's plane
void push_a_surfel(std::vector<float> &vertexs, const dsm_surfel &s) {
    const int surfel_color = (int)s.color;
    const float pos[3] = {s.px, s.py, s.pz}, nrm[3] = {s.nx, s.ny, s.nz};
    float x_dir[3] = {-1 * s.ny, s.nx, 0};
    // Vector3f::normalize(): divide by sqrt(squaredNorm) when that is positive (Eigen 3.3); the three squares
    // of a fixed-size 3-vector are summed as a0 + (a1 + a2)
    const float z = x_dir[0] * x_dir[0] + (x_dir[1] * x_dir[1] + x_dir[2] * x_dir[2]);
    if (z > 0.f) {
        const float nn = std::sqrt(z);
        for (int i = 0; i < 3; i++) x_dir[i] = x_dir[i] / nn;
    }
    const float y_dir[3] = {nrm[1] * x_dir[2] - nrm[2] * x_dir[1], nrm[2] * x_dir[0] - nrm[0] * x_dir[2], nrm[0] * x_dir[1] - nrm[1] * x_dir[0]};
    const float radius = s.size;
    const float h_r = (float)(radius * 0.5);
    const float t_r = (float)(radius * 0.86603);
    float pt[6][3];
    for (int i = 0; i < 3; i++) {
        pt[0][i] = (pos[i] - x_dir[i] * h_r) - y_dir[i] * t_r;
        pt[1][i] = (pos[i] + x_dir[i] * h_r) - y_dir[i] * t_r;
        pt[2][i] = pos[i] - x_dir[i] * radius;
        pt[3][i] = pos[i] + x_dir[i] * radius;
        pt[4][i] = (pos[i] - x_dir[i] * h_r) + y_dir[i] * t_r;
        pt[5][i] = (pos[i] + x_dir[i] * h_r) + y_dir[i] * t_r;
    }
    for (int k = 0; k < 6; k++) {
        for (int i = 0; i < 3; i++) vertexs.push_back(pt[k][i]);
        for (int i = 0; i < 3; i++) vertexs.push_back((float)surfel_color);
    }
}

int download_active(dsm_surfel_map *m, std::vector<dsm_surfel> &out) {
    int32_t n = 0;
    ENGINE_TRY(m, dsm_map_size(m->engine, &n));
    out.resize((size_t)n);
    ENGINE_TRY(m, dsm_map_download(m->engine, out.data(), n, &n));
    out.resize((size_t)n);
    return DSM_OK;
}

} // namespace

extern "C" {

int dsm_surfel_map_create(const dsm_surfel_map_config *cfg, dsm_surfel_map **out) {
    if (!cfg || !out) return DSM_E_INVALID;
    *out = nullptr;
    if (cfg->struct_size != sizeof(dsm_surfel_map_config)) return DSM_E_INVALID; // built against another header
    if (cfg->drift_free_poses < 1) return DSM_E_INVALID;
    dsm_surfel_map *m = new dsm_surfel_map();
    m->cfg = *cfg;
    dsm_config ec;
    dsm_config_init(&ec, cfg->cam_width, cfg->cam_height, cfg->cam_fx, cfg->cam_fy, cfg->cam_cx, cfg->cam_cy, cfg->fuse_far_distence,
                    cfg->fuse_near_distence, cfg->rgbd ? 1 : 0);
    ec.device = cfg->device;
    ec.surfel_capacity = cfg->surfel_capacity;
    ec.pipeline_depth = 1; // live callbacks: one frame at a time,
    ec.flags |= DSM_FLAG_UPLOAD_STREAM; // the next frame goes up while this one is fused
    int rc = dsm_create(&ec, &m->engine); // SurfelMap::SurfelMap -> fusion_functions.initialize (:53)
    if (rc == DSM_OK) rc = dsm_map_upload(m->engine, nullptr, 0);
    if (rc != DSM_OK) {
        dsm_destroy(m->engine);
        delete m;
        return rc;
    }
    *out = m;
    return DSM_OK;
}

void dsm_surfel_map_destroy(dsm_surfel_map *m) {
    if (!m) return;
    dsm_destroy(m->engine);
    for (const Frame &f : m->image_buffer) m->image_pool.release(f);
    for (const Frame &f : m->depth_buffer) m->depth_pool.release(f);
    m->image_pool.drain();
    m->depth_pool.drain();
    delete m;
}

const char *dsm_surfel_map_last_error(const dsm_surfel_map *m) { return m ? m->err.c_str() : "null surfel map"; }

int dsm_surfel_map_image_input(dsm_surfel_map *m, dsm_stamp stamp, int32_t width, int32_t height, size_t step, const char *encoding,
                               const uint8_t *data) {
    if (!m) return DSM_E_INVALID;
    if (m->failed) return fail(m, DSM_E_STATE, "the node failed half-way through a state change earlier and takes no more input");
    if (!encoding || (strcmp(encoding, "mono8") != 0 && strcmp(encoding, "8UC1") != 0))
        return fail(m, DSM_E_INVALID, "image encoding '%s': only mono8 is taken (cv_bridge is not part of this library)", encoding ? encoding : "(null)");
    const int rc = copy_frame(m, m->image_buffer, m->image_pool, stamp, width, height, step, data, 1);
    return rc ? rc : synchronize_msgs(m);
}

int dsm_surfel_map_depth_input(dsm_surfel_map *m, dsm_stamp stamp, int32_t width, int32_t height, size_t step, const char *encoding,
                               const void *data) {
    if (!m) return DSM_E_INVALID;
    if (m->failed) return fail(m, DSM_E_STATE, "the node failed half-way through a state change earlier and takes no more input");
    if (!encoding || strcmp(encoding, "32FC1") != 0)
        return fail(m, DSM_E_INVALID, "depth encoding '%s': only 32FC1 is taken", encoding ? encoding : "(null)");
    const int rc = copy_frame(m, m->depth_buffer, m->depth_pool, stamp, width, height, step, data, 4);
    return rc ? rc : synchronize_msgs(m);
}

int dsm_surfel_map_orb_results_input(dsm_surfel_map *m, dsm_stamp loop_stamp, const float *loop_values, int32_t n_loop_values,
                                     const dsm_pose_msg *loop_path, int32_t n_loop_path, dsm_stamp this_stamp,
                                     const dsm_pose_msg *this_pose, const double *covariance36) {
    if (!m) return DSM_E_INVALID;
    if (m->failed) return fail(m, DSM_E_STATE, "the node failed half-way through a state change earlier and takes no more input");
    if (!this_pose || !covariance36 || n_loop_values < 0 || n_loop_path < 0 || (n_loop_values && !loop_values) || (n_loop_path && !loop_path))
        return fail(m, DSM_E_INVALID, "null/negative argument");
    std::vector<PoseElement> &db = m->poses_database;
    // refuse what the reference would index out of range with
    if (!db.empty() && n_loop_path == 0 ) return fail(m, DSM_E_INVALID, "empty loop path with %zu keyframes (surfel_map.cpp:258-262 reads poses[-1])", db.size());
    const int relative_index = (int)covariance36[1];
    const bool is_new_keyframe = covariance36[0] > 0 || db.empty();
    const size_t n_after = db.size() + (is_new_keyframe ? 1 : 0);
    if (relative_index < 0 || (size_t)relative_index >= n_after || (is_new_keyframe && !db.empty() && (size_t)relative_index >= db.size()))
        return fail(m, DSM_E_INVALID, "reference keyframe %d of %zu", relative_index, n_after);

    // :213-232 the SLAM frame is turned so that the first camera looks along +y with z up
    dsm_pose_msg input_pose = *this_pose;
    {
        const Mat4 received = pose_to_matrix(input_pose);
        if (db.empty()) {
            Mat4 idea;
            for (int k = 0; k < 16; k++) idea.d[k] = 0.0;
            idea(0, 0) = 1.0;
            idea(1, 2) = 1.0;
            idea(2, 1) = -1.0;
            idea(3, 3) = 1.0;
            m->transform_kitti = mul(idea, inverse(received));
        }
        input_pose = matrix_to_pose(mul(m->transform_kitti, received));
    }

    // :235-252 take over the loop-corrected keyframe poses
    bool loop_changed = false;
    for (size_t i = 0; i < db.size() && i < (size_t)n_loop_path; i++) {
        db[i].loop_pose = matrix_to_pose(mul(m->transform_kitti, pose_to_matrix(loop_path[i])));
        if (!same_position(db[i].loop_pose, db[i].cam_pose)) loop_changed = true;
    }
    // :254-270 keyframes the path does not cover yet follow the last covered one rigidly
    if (db.size() > (size_t)n_loop_path) {
        const size_t last_update_index = (size_t)n_loop_path - 1;
        const Mat4 warp_pose = mul(pose_to_matrix(db[last_update_index].loop_pose), inverse(pose_to_matrix(db[last_update_index].cam_pose)));
        for (size_t i = (size_t)n_loop_path; i < db.size(); i++) db[i].loop_pose = matrix_to_pose(mul(warp_pose, pose_to_matrix(db[i].cam_pose)));
    }

    if (loop_changed) { // :277-280
        const int rc = warp_surfels(m);
        if (rc) return rc;
    }

    // :287-314 loop edges
    const int loop_num = n_loop_values / 2;
    for (int i = 0; i < loop_num; i++) {
        const int loop_first = (int)loop_values[i * 2], loop_second = (int)loop_values[i * 2 + 1];
        // the reference compares the ints against size_t: negative indices count as "not found"
        if (loop_first >= 0 && loop_second >= 0 && (size_t)loop_first < db.size() && (size_t)loop_second < db.size()) {
            std::vector<int> &a = db[(size_t)loop_first].linked_pose_index;
            if (std::find(a.begin(), a.end(), loop_second) == a.end()) {
                a.push_back(loop_second);
                std::vector<int> &b = db[(size_t)loop_second].linked_pose_index;
                if (std::find(b.begin(), b.end(), loop_first) == b.end()) b.push_back(loop_first);
            }
        }
    }

    if (is_new_keyframe) { // :316-348
        PoseElement pe;
        const int this_pose_index = (int)db.size();
        pe.cam_pose = input_pose;
        pe.loop_pose = input_pose;
        pe.cam_stamp = this_stamp;
        if (!db.empty()) {
            pe.linked_pose_index.push_back(relative_index);
            db[(size_t)relative_index].linked_pose_index.push_back(this_pose_index);
        }
        db.push_back(pe);
        m->local_surfels_indexs.insert(this_pose_index);
    }

    // :351-358 queue the frame for fusion, relative to its reference keyframe
    const Mat4 relative_pose = mul(inverse(pose_to_matrix(db[(size_t)relative_index].cam_pose)), pose_to_matrix(input_pose));
    m->pose_reference_buffer.push_back(std::make_tuple(loop_stamp, matrix_to_pose(relative_pose), relative_index));
    return synchronize_msgs(m);
}

// SurfelMap::save_cloud (:1153-1174).  pcl::io::savePCDFile(name, cloud) writes ASCII PCD v0.7 (PCL is an
// un-vendored dependency of the reference; this is its published file layout: eight significant digits,
// "nan" for NaNs, one trimmed line per point).
int dsm_surfel_map_save_cloud(dsm_surfel_map *m, const char *path) {
    if (!m || !path) return DSM_E_INVALID;
    std::vector<dsm_surfel> active;
    int rc = download_active(m, active);
    if (rc) return rc;
    std::vector<float> pts;
    for (const dsm_surfel &s : active) {
        if (s.update_times < 5) continue;
        pts.push_back(s.px); pts.push_back(s.py); pts.push_back(s.pz); pts.push_back(s.color);
    }
    int32_t n_in = 0;
    ENGINE_TRY(m, dsm_store_size(m->engine, &n_in));
    const size_t n_act = pts.size() / 4;
    pts.resize((n_act + (size_t)n_in) * 4);
    if (n_in) ENGINE_TRY(m, dsm_store_download(m->engine, 0, n_in, nullptr, &pts[n_act * 4]));
    const size_t n = pts.size() / 4;
    if (n == 0) return fail(m, DSM_E_STATE, "save_cloud: no points (pcl::PCDWriter throws \"Input point cloud has no data!\")");
    std::ofstream fs(path);
    if (!fs) return fail(m, DSM_E_INVALID, "cannot open %s", path);
    fs.precision(8);
    fs.imbue(std::locale::classic());
    fs << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
       << "WIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA ascii\n";
    std::ostringstream line;
    line.precision(8);
    line.imbue(std::locale::classic());
    for (size_t i = 0; i < n; i++) {
        line.str("");
        for (int d = 0; d < 4; d++) {
            const float v = pts[i * 4 + d];
            if (std::isnan(v)) line << "nan"; else line << v;
            if (d < 3) line << " ";
        }
        fs << line.str() << "\n";
    }
    fs.close();
    return fs ? DSM_OK : fail(m, DSM_E_INVALID, "write to %s failed", path);
}

// SurfelMap::save_mesh (:1219-1281)
int dsm_surfel_map_save_mesh(dsm_surfel_map *m, const char *path) {
    if (!m || !path) return DSM_E_INVALID;
    std::ofstream stream(path);
    if (!stream) return DSM_OK; // :1221-1223: the reference returns silently
    std::vector<float> vertexs;
    int32_t n_in = 0;
    ENGINE_TRY(m, dsm_store_size(m->engine, &n_in));
    std::vector<dsm_surfel> inactive((size_t)n_in);
    if (n_in) ENGINE_TRY(m, dsm_store_download(m->engine, 0, n_in, inactive.data(), nullptr));
    for (const PoseElement &pe : m->poses_database) { // keyframe order, not store order
        if (pe.segment < 0) continue;
        const Segment &sg = m->segments[(size_t)pe.segment];
        for (int j = 0; j < sg.count; j++) push_a_surfel(vertexs, inactive[(size_t)sg.begin + (size_t)j]);
    }
    std::vector<dsm_surfel> active;
    const int rc = download_active(m, active);
    if (rc) return rc;
    for (const dsm_surfel &s : active)
        if (s.update_times >= 5) push_a_surfel(vertexs, s);

    const size_t numPoints = vertexs.size() / 6, numSurfels = numPoints / 6;
    stream << "ply\nformat ascii 1.0\nelement vertex " << numPoints << "\nproperty float x\nproperty float y\nproperty float z\n"
           << "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face " << numSurfels * 4
           << "\nproperty list uchar int vertex_index\nend_header\n";
    for (size_t i = 0; i < numPoints; i++) {
        for (int j = 0; j < 6; j++) stream << vertexs[i * 6 + (size_t)j] << " ";
        stream << "\n";
    }
    for (size_t i = 0; i < numSurfels; i++) {
        const size_t p1 = i * 6, p2 = i * 6 + 1, p3 = i * 6 + 2, p4 = i * 6 + 3, p5 = i * 6 + 4, p6 = i * 6 + 5;
        stream << "3 " << p1 << " " << p2 << " " << p3 << "\n";
        stream << "3 " << p2 << " " << p4 << " " << p3 << "\n";
        stream << "3 " << p3 << " " << p4 << " " << p5 << "\n";
        stream << "3 " << p5 << " " << p4 << " " << p6 << "\n";
    }
    stream.close();
    return DSM_OK;
}

int dsm_surfel_map_save_map(dsm_surfel_map *m, const char *path) { return dsm_surfel_map_save_mesh(m, path); }

dsm_handle *dsm_surfel_map_engine(dsm_surfel_map *m) { return m ? m->engine : nullptr; }
int64_t dsm_surfel_map_frames_fused(const dsm_surfel_map *m) { return m ? m->frames_fused : -1; }
int64_t dsm_surfel_map_dropped_poses(const dsm_surfel_map *m) { return m ? m->poses_dropped : -1; }
int32_t dsm_surfel_map_pose_count(const dsm_surfel_map *m) { return m ? (int32_t)m->poses_database.size() : DSM_E_INVALID; }

int dsm_surfel_map_get_pose(const dsm_surfel_map *m, int32_t i, dsm_pose_msg *cam_pose, dsm_pose_msg *loop_pose, int32_t *n_attached,
                            int32_t *points_begin_index, int32_t *is_local) {
    if (!m || i < 0 || (size_t)i >= m->poses_database.size()) return DSM_E_INVALID;
    const PoseElement &pe = m->poses_database[(size_t)i];
    if (cam_pose) *cam_pose = pe.cam_pose;
    if (loop_pose) *loop_pose = pe.loop_pose;
    if (n_attached) *n_attached = pe.segment >= 0 ? m->segments[(size_t)pe.segment].count : 0;
    if (points_begin_index) *points_begin_index = pe.segment >= 0 ? m->segments[(size_t)pe.segment].begin : -1;
    if (is_local) *is_local = m->local_surfels_indexs.count(i) ? 1 : 0;
    return DSM_OK;
}

int32_t dsm_surfel_map_get_links(const dsm_surfel_map *m, int32_t i, int32_t *out, int32_t cap) {
    if (!m || i < 0 || (size_t)i >= m->poses_database.size() || cap < 0 || (cap && !out)) return DSM_E_INVALID;
    const std::vector<int> &l = m->poses_database[(size_t)i].linked_pose_index;
    for (size_t k = 0; k < l.size() && k < (size_t)cap; k++) out[k] = l[k];
    return (int32_t)l.size();
}

int dsm_surfel_map_get_attached(dsm_surfel_map *m, int32_t i, dsm_surfel *out, int32_t cap, int32_t *n) {
    if (!m || !n || i < 0 || (size_t)i >= m->poses_database.size() || cap < 0 || (cap && !out)) return DSM_E_INVALID;
    const PoseElement &pe = m->poses_database[(size_t)i];
    const Segment none = {i, 0, 0};
    const Segment &sg = pe.segment >= 0 ? m->segments[(size_t)pe.segment] : none;
    *n = sg.count;
    if (sg.count > cap) return fail(m, DSM_E_CAPACITY, "%d attached surfels exceed cap %d", sg.count, cap);
    if (sg.count) ENGINE_TRY(m, dsm_store_download(m->engine, sg.begin, sg.count, out, nullptr));
    return DSM_OK;
}

int dsm_surfel_map_get_inactive_cloud(dsm_surfel_map *m, float *xyzi_out, int32_t cap, int32_t *n) {
    if (!m || !n || cap < 0 || (cap && !xyzi_out)) return DSM_E_INVALID;
    int32_t total = 0;
    ENGINE_TRY(m, dsm_store_size(m->engine, &total));
    *n = total;
    if (total > cap) return fail(m, DSM_E_CAPACITY, "%d inactive points exceed cap %d", total, cap);
    if (total) ENGINE_TRY(m, dsm_store_download(m->engine, 0, total, nullptr, xyzi_out));
    return DSM_OK;
}

} // extern "C"
