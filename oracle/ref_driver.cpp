// ORACLE / TEST INFRASTRUCTURE -- never linked into or called by the product path.
//
// C-callable driver around the *reference's own* translation unit
// /root/reference/surfel_fusion/src/fusion_functions.cpp, which is compiled in
// place (never copied) by `#include`-ing it below with oracle/shims/ on the
// include path in front of the reference's src/ directory.
// Built by oracle/Makefile into oracle/_ref/ only.
//
// What is reference code here: everything reached through FusionFunctions.
// What is restated here: SurfelMap::fuse_map's hole-refill + swap-with-last
// compaction (surfel_map.cpp:1077-1109), because surfel_map.cpp needs ROS/PCL
// and cannot be compiled in this image.
#include "fusion_functions.h"
#ifdef DSM_REF_RGBD
// The reference keeps its RGB-D constant set as comments (fusion_functions.h:17-21).
#undef HUBER_RANGE
#undef BASELINE
#undef DISPARITY_ERROR
#undef MIN_TOLERATE_DIFF
#define HUBER_RANGE 0.05
#define BASELINE 0.08
#define DISPARITY_ERROR 1.0
#define MIN_TOLERATE_DIFF 0.05
#endif
#include "fusion_functions.cpp"  // resolved via -I/root/reference/surfel_fusion/src

#include <cstdint>
#include <cstring>
#include <vector>

namespace {
struct RefHandle {
    FusionFunctions ff;
    int w, h;
    std::vector<SurfelElement> local, fresh;
};

Eigen::Matrix4f pose_from(const float *p) {
    Eigen::Matrix4f m;
    for (int i = 0; i < 16; i++) m.d[i] = p[i];  // both column-major
    return m;
}
}  // namespace

#ifdef DSM_ORACLE_EIGEN_PERTURB
int dsm_eigen_ulps_f[16] = {0}, dsm_eigen_ulps_d[16] = {0};
#endif

extern "C" {

#ifdef DSM_ORACLE_EIGEN_PERTURB
// exposure study only (tools/eigen_exposure.py): ulps added to every element of Matrix4f::inverse() (FF.cpp:59) and
// Matrix4d::inverse() (FF.cpp:176)
void dsmref_set_eigen_perturb(const int *f16, const int *d16) {
    for (int i = 0; i < 16; i++) { dsm_eigen_ulps_f[i] = f16 ? f16[i] : 0; dsm_eigen_ulps_d[i] = d16 ? d16[i] : 0; }
}
#endif

// the TU's own Matrix4f::inverse() (FF.cpp:59) of a column-major pose -- what a caller with the reference's matrix type
// hands to the product's *_inv entry points (tests/golden/make_golden_inv.py records it)
void dsmref_inverse4f(const float *pose16, float *inv16) {
    const Eigen::Matrix4f inv = pose_from(pose16).inverse();
    for (int i = 0; i < 16; i++) inv16[i] = inv.d[i];
}

void *dsmref_create(int w, int h, float fx, float fy, float cx, float cy, float far_d, float near_d) {
    RefHandle *r = new RefHandle();
    r->w = w;
    r->h = h;
    r->ff.initialize(w, h, fx, fy, cx, cy, far_d, near_d);
    return r;
}

void dsmref_destroy(void *hv) { delete (RefHandle *)hv; }

int dsmref_sizeof_seed(void) { return (int)sizeof(Superpixel_seed); }
int dsmref_sizeof_surfel(void) { return (int)sizeof(SurfelElement); }

// FusionFunctions::fuse_initialize_map (FF.cpp:30-83): local updated in place,
// new surfels written to new_out (capacity new_cap), count to *n_new.
int dsmref_fuse_initialize_map(void *hv, int ref_idx, const uint8_t *img, size_t img_step, const float *depth,
                               size_t depth_step, const float *pose16, SurfelElement *local, int n_local,
                               SurfelElement *new_out, int new_cap, int *n_new) {
    RefHandle *r = (RefHandle *)hv;
    cv::Mat image(r->h, r->w, img_step, (void *)img);
    cv::Mat dep(r->h, r->w, depth_step, (void *)depth);
    Eigen::Matrix4f pose = pose_from(pose16);
    r->local.assign(local, local + n_local);
    r->ff.fuse_initialize_map(ref_idx, image, dep, pose, r->local, r->fresh);
    if (n_local) memcpy(local, r->local.data(), sizeof(SurfelElement) * (size_t)n_local);
    *n_new = (int)r->fresh.size();
    if ((int)r->fresh.size() > new_cap) return -1;
    if (!r->fresh.empty()) memcpy(new_out, r->fresh.data(), sizeof(SurfelElement) * r->fresh.size());
    return 0;
}

// SurfelMap::fuse_map (surfel_map.cpp:1060-1113): the call above, then the
// serial refill/compaction loop restated from surfel_map.cpp:1077-1109.
int dsmref_fuse_map(void *hv, int ref_idx, const uint8_t *img, size_t img_step, const float *depth, size_t depth_step,
                    const float *pose16, SurfelElement *local, int *n_local, int cap, int *n_new) {
    RefHandle *r = (RefHandle *)hv;
    cv::Mat image(r->h, r->w, img_step, (void *)img);
    cv::Mat dep(r->h, r->w, depth_step, (void *)depth);
    Eigen::Matrix4f pose = pose_from(pose16);
    std::vector<SurfelElement> &ls = r->local;
    ls.assign(local, local + *n_local);
    r->ff.fuse_initialize_map(ref_idx, image, dep, pose, ls, r->fresh);
    std::vector<int> holes;
    for (int i = 0; i < (int)ls.size(); i++)
        if (ls[i].update_times == 0) holes.push_back(i);
    int added = 0;
    for (size_t j = 0; j < r->fresh.size(); j++) {
        if (r->fresh[j].update_times == 0) continue;
        if (!holes.empty()) {
            ls[holes.back()] = r->fresh[j];
            holes.pop_back();
        } else {
            ls.push_back(r->fresh[j]);
        }
        added++;
    }
    while (!holes.empty()) {
        ls[holes.back()] = ls.back();
        holes.pop_back();
        ls.pop_back();
    }
    *n_new = added;
    if ((int)ls.size() > cap) return -1;
    *n_local = (int)ls.size();
    if (!ls.empty()) memcpy(local, ls.data(), sizeof(SurfelElement) * ls.size());
    return 0;
}

// ---- parity taps (private members reached with -fno-access-control) ----
void dsmref_get_labels(void *hv, int32_t *out) {
    RefHandle *r = (RefHandle *)hv;
    memcpy(out, r->ff.superpixel_index.data(), sizeof(int) * r->ff.superpixel_index.size());
}
void dsmref_set_labels(void *hv, const int32_t *in) {
    RefHandle *r = (RefHandle *)hv;
    memcpy(r->ff.superpixel_index.data(), in, sizeof(int) * r->ff.superpixel_index.size());
}
void dsmref_get_seeds(void *hv, void *out) {
    RefHandle *r = (RefHandle *)hv;
    memcpy(out, r->ff.superpixel_seeds.data(), sizeof(Superpixel_seed) * r->ff.superpixel_seeds.size());
}
void dsmref_set_seeds(void *hv, const void *in) {
    RefHandle *r = (RefHandle *)hv;
    memcpy(r->ff.superpixel_seeds.data(), in, sizeof(Superpixel_seed) * r->ff.superpixel_seeds.size());
}
void dsmref_get_norm_map(void *hv, float *out) {
    RefHandle *r = (RefHandle *)hv;
    memcpy(out, r->ff.norm_map.data(), sizeof(float) * r->ff.norm_map.size());
}
void dsmref_get_space_map(void *hv, double *out) {
    RefHandle *r = (RefHandle *)hv;
    memcpy(out, r->ff.space_map.data(), sizeof(double) * r->ff.space_map.size());
}

// ---- stage-level entry points for state-level unit tests ----
void dsmref_set_frame(void *hv, const uint8_t *img, size_t img_step, const float *depth, size_t depth_step) {
    RefHandle *r = (RefHandle *)hv;
    r->ff.image = cv::Mat(r->h, r->w, img_step, (void *)img);
    r->ff.depth = cv::Mat(r->h, r->w, depth_step, (void *)depth);
}
void dsmref_generate_super_pixels(void *hv) { ((RefHandle *)hv)->ff.generate_super_pixels(); }
void dsmref_initialize_seeds(void *hv) { ((RefHandle *)hv)->ff.initialize_seeds(); }
void dsmref_update_pixels(void *hv) { ((RefHandle *)hv)->ff.update_pixels(); }
void dsmref_update_seeds(void *hv) { ((RefHandle *)hv)->ff.update_seeds(); }
void dsmref_calculate_norms(void *hv) { ((RefHandle *)hv)->ff.calculate_norms(); }

}  // extern "C"
