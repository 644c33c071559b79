// TEST: include/dsm_surfel_map.hpp (the reference's SurfelMap interface) fed with a recorded message stream.
//   node_replay_test <events.bin> <out.PCD> <out.PLY>
// events.bin (written by tests: little-endian): header int32 {W, H, drift_free_poses}, float32 {fx, fy, cx, cy,
// far, near}; then records: int32 kind (0 image, 1 depth, 2 orb, -1 end), uint32 sec, uint32 nsec, payload:
//   image: W*H bytes; depth: W*H float32; orb: int32 n_values, float32 values[], int32 n_path, float64 path[n][7],
//   float64 pose[7], float64 cov[36].
//   exit 0 = replayed and saved; 77 = no gfx950 device (the constructor threw, nothing was computed); 1 = error
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/dsm_surfel_map.hpp"

template <typename T> static bool rd(FILE *f, T *p, size_t n = 1) { return fread(p, sizeof(T), n, f) == n; }

static dsm::msg::Pose pose_from(const double *p) {
    dsm::msg::Pose o;
    o.position.x = p[0]; o.position.y = p[1]; o.position.z = p[2];
    o.orientation.x = p[3]; o.orientation.y = p[4]; o.orientation.z = p[5]; o.orientation.w = p[6];
    return o;
}

int main(int argc, char **argv) {
    if (argc != 4) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 1;
    int32_t hdr[3];
    float cam[6];
    if (!rd(f, hdr, 3) || !rd(f, cam, 6)) return 1;
    dsm::SurfelMap::Params p;
    p.cam_width = hdr[0]; p.cam_height = hdr[1]; p.drift_free_poses = hdr[2];
    p.cam_fx = cam[0]; p.cam_fy = cam[1]; p.cam_cx = cam[2]; p.cam_cy = cam[3];
    p.fuse_far_distence = cam[4]; p.fuse_near_distence = cam[5];
    try {
        dsm::SurfelMap surfel_map(p);
        const size_t n_px = (size_t)hdr[0] * (size_t)hdr[1];
        for (;;) {
            int32_t kind;
            uint32_t st[2];
            if (!rd(f, &kind) || kind < 0) break;
            if (!rd(f, st, 2)) return 1;
            if (kind == 0 || kind == 1) {
                std::shared_ptr<dsm::msg::Image> m(new dsm::msg::Image);
                m->header.stamp.sec = st[0]; m->header.stamp.nsec = st[1];
                m->width = (uint32_t)hdr[0]; m->height = (uint32_t)hdr[1];
                const size_t elem = kind == 0 ? 1 : 4;
                m->step = (uint32_t)(hdr[0] * elem);
                m->encoding = kind == 0 ? "mono8" : "32FC1";
                m->data.resize(n_px * elem);
                if (!rd(f, m->data.data(), m->data.size())) return 1;
                if (kind == 0) surfel_map.image_input(dsm::msg::ImageConstPtr(m));
                else surfel_map.depth_input(dsm::msg::ImageConstPtr(m));
            } else {
                std::shared_ptr<dsm::msg::PointCloud> ls(new dsm::msg::PointCloud);
                std::shared_ptr<dsm::msg::Path> lp(new dsm::msg::Path);
                std::shared_ptr<dsm::msg::Odometry> od(new dsm::msg::Odometry);
                ls->header.stamp.sec = st[0]; ls->header.stamp.nsec = st[1];
                od->header.stamp = ls->header.stamp;
                int32_t nv, np;
                if (!rd(f, &nv)) return 1;
                ls->channels.resize(1);
                ls->channels[0].values.resize((size_t)nv);
                if (nv && !rd(f, ls->channels[0].values.data(), (size_t)nv)) return 1;
                if (!rd(f, &np)) return 1;
                lp->poses.resize((size_t)np);
                for (int i = 0; i < np; i++) {
                    double q[7];
                    if (!rd(f, q, 7)) return 1;
                    lp->poses[(size_t)i].pose = pose_from(q);
                }
                double q[7];
                if (!rd(f, q, 7) || !rd(f, od->pose.covariance, 36)) return 1;
                od->pose.pose = pose_from(q);
                surfel_map.orb_results_input(dsm::msg::PointCloudConstPtr(ls), dsm::msg::PathConstPtr(lp), dsm::msg::OdometryConstPtr(od));
            }
        }
        surfel_map.save_cloud(argv[2]);
        std::shared_ptr<dsm::msg::String> name(new dsm::msg::String);
        name->data = argv[3];
        surfel_map.save_map(dsm::msg::StringConstPtr(name)); // = save_mesh (surfel_map.cpp:75-81)
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return strstr(e.what(), "device") || strstr(e.what(), "GPU") ? 77 : 1;
    }
    fclose(f);
    return 0;
}
