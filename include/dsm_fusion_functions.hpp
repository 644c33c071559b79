// dsm_fusion_functions.hpp -- C++ host side above the C ABI (include/dsm.h), with the reference's
// own class and method names so that surfel_fusion/src/surfel_map.cpp compiles against it unchanged:
//
//   reference                                             this header
//   ----------------------------------------------------  ---------------------------------------
//   FusionFunctions::initialize(w,h,fx,fy,cx,cy,far,near)  dsm::FusionFunctions::initialize(...)
//     fusion_functions.h:84-87, .cpp:7-28
//   FusionFunctions::fuse_initialize_map(ref, image,       dsm::FusionFunctions::fuse_initialize_map(...)
//     depth, pose, local_surfels, new_surfels)
//     fusion_functions.h:88-94, .cpp:30-83
//   SurfelMap::fuse_map body (engine call + refill +       dsm::FusionFunctions::fuse_map(...)
//     swap-with-last) surfel_map.cpp:1060-1113
//
// The image / matrix types are template parameters with the members the reference uses
// (cv::Mat: rows, cols, step, data; Eigen::Matrix4f: data(), column-major), so the header needs
// neither OpenCV nor Eigen.  SurfelElement must have the 44-byte layout of elements.h:22-31.
// Errors: the reference returns void and prints; these methods throw std::runtime_error with the
// library's message (define DSM_NO_EXCEPTIONS to get the int status instead).
#ifndef DSM_FUSION_FUNCTIONS_HPP
#define DSM_FUSION_FUNCTIONS_HPP

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dsm.h"

namespace dsm {

namespace detail {
// pose.inverse() with the CALLER's matrix library when the pose type has one (Eigen::Matrix4f in the reference,
// fusion_functions.cpp:59): the engine then uses exactly the world->cam matrix the reference's own statement would have
// produced in this build, whatever Eigen version / instruction set it was compiled with.  A pose type without
// inverse() (a plain array wrapper) leaves it to the library's closed form (dsm.h, dsm_fuse_map_inv).
template <typename Pose> struct PoseInverse {
    template <typename P> static auto get(const P &p, Pose &out, int) -> decltype(out = p.inverse(), static_cast<const float *>(out.data())) {
        out = p.inverse();
        return out.data();
    }
    template <typename P> static const float *get(const P &, Pose &, long) { return nullptr; }
};
} // namespace detail

class FusionFunctions {
  public:
    FusionFunctions() = default;
    FusionFunctions(const FusionFunctions &) = delete;
    FusionFunctions &operator=(const FusionFunctions &) = delete;
    ~FusionFunctions() { dsm_destroy(h_); }

    // fusion_functions.h:84-87.  rgbd selects the constant set of fusion_functions.h:17-21.
    int initialize(int width, int height, float fx, float fy, float cx, float cy, float far_dist, float near_dist,
                   bool rgbd = false, int device = 0, int surfel_capacity = 0) {
        dsm_destroy(h_);
        h_ = nullptr;
        dsm_config cfg;
        dsm_config_init(&cfg, width, height, fx, fy, cx, cy, far_dist, near_dist, rgbd ? 1 : 0);
        cfg.device = device;
        cfg.surfel_capacity = surfel_capacity;
        n_seed_ = (width / 8) * (height / 8);
        return check(dsm_create(&cfg, &h_), nullptr);
    }

    // fusion_functions.h:88-94: local_surfels is updated in place, new_surfels is cleared and filled.
    template <typename Mat, typename Pose, typename Surfel>
    int fuse_initialize_map(int reference_frame_index, const Mat &image, const Mat &depth, const Pose &pose,
                            std::vector<Surfel> &local_surfels, std::vector<Surfel> &new_surfels) {
        static_assert(sizeof(Surfel) == sizeof(dsm_surfel), "SurfelElement must keep the layout of elements.h:22-31");
        new_surfels.resize((size_t)n_seed_);
        int32_t n_new = 0;
        Pose inv_store(pose);
        const float *inv = detail::PoseInverse<Pose>::get(pose, inv_store, 0); // FF.cpp:59, in the caller's arithmetic
        const int rc = dsm_fuse_initialize_map_inv(h_, reference_frame_index, (const uint8_t *)image.data, (size_t)image.step,
                                                   (const float *)depth.data, (size_t)depth.step, pose.data(), inv,
                                                   reinterpret_cast<dsm_surfel *>(local_surfels.data()),
                                                   (int32_t)local_surfels.size(), reinterpret_cast<dsm_surfel *>(new_surfels.data()),
                                                   (int32_t)new_surfels.size(), &n_new);
        new_surfels.resize(rc == DSM_OK ? (size_t)n_new : 0);
        return check(rc, h_);
    }

    // surfel_map.cpp:1060-1113 in one call: engine + refill of deleted slots + swap-with-last.
    template <typename Mat, typename Pose, typename Surfel>
    int fuse_map(int reference_index, const Mat &image, const Mat &depth, const Pose &pose, std::vector<Surfel> &local_surfels,
                 int *n_new_out = nullptr) {
        static_assert(sizeof(Surfel) == sizeof(dsm_surfel), "SurfelElement must keep the layout of elements.h:22-31");
        int32_t n_local = (int32_t)local_surfels.size(), n_new = 0;
        local_surfels.resize((size_t)n_local + (size_t)n_seed_);
        Pose inv_store(pose);
        const float *inv = detail::PoseInverse<Pose>::get(pose, inv_store, 0); // FF.cpp:59, in the caller's arithmetic
        const int rc = dsm_fuse_map_inv(h_, reference_index, (const uint8_t *)image.data, (size_t)image.step, (const float *)depth.data,
                                        (size_t)depth.step, pose.data(), inv, reinterpret_cast<dsm_surfel *>(local_surfels.data()),
                                        &n_local, (int32_t)local_surfels.size(), &n_new);
        local_surfels.resize((size_t)n_local);
        if (n_new_out) *n_new_out = n_new;
        return check(rc, h_);
    }

    dsm_handle *handle() const { return h_; }

  private:
    static int check(int rc, dsm_handle *h) {
#ifndef DSM_NO_EXCEPTIONS
        if (rc != DSM_OK) throw std::runtime_error(std::string("dsm: ") + dsm_last_error(h));
#else
        (void)h;
#endif
        return rc;
    }
    dsm_handle *h_ = nullptr;
    int n_seed_ = 0;
};

} // namespace dsm
#endif
