// TEST: the C++ facade (include/dsm_fusion_functions.hpp) used the way surfel_map.cpp uses the
// reference's FusionFunctions, checked against the CPU oracle (oracle/dsm_oracle.h).
//   exit 0  = parity (bit-exact, NaN == NaN) over a short synthetic sequence
//   exit 77 = no gfx950 device: the facade threw, nothing was computed on the CPU
//   exit 1  = mismatch / error
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/dsm_fusion_functions.hpp"
#include "../../oracle/dsm_oracle.h"

struct Mat { // the members of cv::Mat the reference touches
    int rows, cols;
    size_t step;
    unsigned char *data;
};
struct Matrix4f { // Eigen::Matrix4f: 16 floats, column-major
    float m[16];
    const float *data() const { return m; }
};
struct SurfelElement { // elements.h:22-31
    float px, py, pz, nx, ny, nz, size, color, weight;
    int update_times, last_update;
};

static bool same_bits(const SurfelElement &a, const dsmo_surfel &b) {
    const float *x = &a.px, *y = &b.px;
    for (int i = 0; i < 9; i++)
        if (memcmp(&x[i], &y[i], 4) != 0 && !(std::isnan(x[i]) && std::isnan(y[i]))) return false;
    return a.update_times == b.update_times && a.last_update == b.last_update;
}

int main() {
    const int W = 320, H = 200;
    const float fx = 260, fy = 260, cx = 159.5f, cy = 99.5f;
    std::vector<unsigned char> img((size_t)W * H);
    std::vector<float> dep((size_t)W * H);
    dsm::FusionFunctions ff;
    try {
        ff.initialize(W, H, fx, fy, cx, cy, 30.0f, 0.5f);
    } catch (const std::exception &e) {
        printf("no device: %s\n", e.what());
        return 77;
    }
    dsmo_ctx *orc = dsmo_create(W, H, fx, fy, cx, cy, 30.0f, 0.5f);
    std::vector<SurfelElement> local;
    std::vector<dsmo_surfel> olocal;
    for (int t = 0; t < 6; t++) {
        unsigned rng = 1234u + 77u * (unsigned)t;
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                rng = rng * 1664525u + 1013904223u;
                // ground plane below the camera plus a slanted wall, 3 % holes
                const float ray_y = ((float)y - cy) / fy, ray_x = ((float)x - cx) / fx;
                float d = ray_y > 0.05f ? 1.6f / ray_y : 0.0f;
                const float wall = (12.0f - 0.3f * (float)t) / (1.0f + 0.4f * ray_x);
                if (d == 0.0f || wall < d) d = wall;
                if (d > 40.0f || (rng >> 8) % 100 < 3) d = 0.0f;
                dep[(size_t)y * W + x] = d * (1.0f + 0.002f * ((float)((rng >> 12) & 255) / 255.0f - 0.5f));
                img[(size_t)y * W + x] = (unsigned char)(((x / 16 + y / 16) & 1 ? 150 : 90) + (rng >> 20) % 20);
            }
        Mat image{H, W, (size_t)W, img.data()}, depth{H, W, (size_t)W * 4, (unsigned char *)dep.data()};
        Matrix4f pose;
        memset(pose.m, 0, sizeof pose.m);
        pose.m[0] = pose.m[5] = pose.m[10] = pose.m[15] = 1.0f;
        pose.m[14] = 0.3f * (float)t; // forward motion
        int n_new = 0;
        ff.fuse_map(t / 2, image, depth, pose, local, &n_new);
        int on = (int)olocal.size(), o_new = 0;
        olocal.resize((size_t)on + (W / 8) * (H / 8));
        if (dsmo_fuse_map(orc, t / 2, img.data(), W, dep.data(), (size_t)W * 4, pose.m, olocal.data(), &on, (int)olocal.size(), &o_new)) return 1;
        olocal.resize((size_t)on);
        if (n_new != o_new || local.size() != olocal.size()) {
            printf("frame %d: counts differ: %d/%zu vs %d/%zu\n", t, n_new, local.size(), o_new, olocal.size());
            return 1;
        }
        for (size_t i = 0; i < local.size(); i++)
            if (!same_bits(local[i], olocal[i])) {
                printf("frame %d: surfel %zu differs\n", t, i);
                return 1;
            }
        printf("frame %d: %zu surfels, %d new: identical\n", t, local.size(), n_new);
    }
    dsmo_destroy(orc);
    return 0;
}
