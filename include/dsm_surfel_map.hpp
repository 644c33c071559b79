// dsm_surfel_map.hpp -- C++ host side above include/dsm_surfel_map.h with the reference's own class and
// method names, so that surfel_fusion/src/ros_node.cpp wires its subscribers to it unchanged:
//
//   reference (surfel_map.h:48-62)                                   this header
//   ---------------------------------------------------------------  ------------------------------------
//   SurfelMap(ros::NodeHandle &)   params of surfel_map.cpp:13-28     dsm::SurfelMap(const Params &)
//   image_input(const sensor_msgs::ImageConstPtr &)                   image_input(const ImagePtr &)
//   depth_input(const sensor_msgs::ImageConstPtr &)                   depth_input(const ImagePtr &)
//   orb_results_input(const sensor_msgs::PointCloudConstPtr &,        orb_results_input(const PointCloudPtr &,
//       const nav_msgs::PathConstPtr &, const nav_msgs::OdometryConstPtr &)   const PathPtr &, const OdometryPtr &)
//   save_map(const std_msgs::StringConstPtr &)                        save_map(const StringPtr &)
//   save_cloud(string) / save_mesh(string)                            save_cloud / save_mesh
//
// The callbacks are templates over the message pointer type: anything with the members the reference reads
// works -- the real ROS messages (header.stamp.sec/.nsec, width, height, step, encoding, data; channels[0].values;
// poses[i].pose; pose.pose, pose.covariance) when roscpp is present, or the plain structs of namespace
// dsm::msg below when it is not (this image has no ROS).  Images must already be mono8 / 32FC1: the reference
// converts with cv_bridge::toCvCopy (surfel_map.cpp:86,96), which is not part of this library.
// Errors: the reference returns void and prints; these throw std::runtime_error (-DDSM_NO_EXCEPTIONS: status).
#ifndef DSM_SURFEL_MAP_HPP
#define DSM_SURFEL_MAP_HPP

#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "dsm_surfel_map.h"

namespace dsm {

namespace msg { // ROS-free mirrors of the message fields the node reads
struct Time {
    uint32_t sec = 0, nsec = 0;
};
struct Header {
    uint32_t seq = 0;
    Time stamp;
    std::string frame_id;
};
struct Point {
    double x = 0, y = 0, z = 0;
};
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 1;
};
struct Pose {
    Point position;
    Quaternion orientation;
};
struct PoseStamped {
    Header header;
    Pose pose;
};
struct PoseWithCovariance {
    Pose pose;
    double covariance[36] = {0};
};
struct Image { // sensor_msgs/Image
    Header header;
    uint32_t height = 0, width = 0;
    std::string encoding;
    uint8_t is_bigendian = 0;
    uint32_t step = 0;
    std::vector<uint8_t> data;
};
struct ChannelFloat32 {
    std::string name;
    std::vector<float> values;
};
struct PointCloud { // sensor_msgs/PointCloud: channels[0].values = flat pairs of keyframe indices
    Header header;
    std::vector<ChannelFloat32> channels;
};
struct Path { // nav_msgs/Path
    Header header;
    std::vector<PoseStamped> poses;
};
struct Odometry { // nav_msgs/Odometry: pose.covariance[0] > 0 = new keyframe, [1] = reference keyframe
    Header header;
    PoseWithCovariance pose;
};
struct String {
    std::string data;
};
typedef std::shared_ptr<const Image> ImageConstPtr;
typedef std::shared_ptr<const PointCloud> PointCloudConstPtr;
typedef std::shared_ptr<const Path> PathConstPtr;
typedef std::shared_ptr<const Odometry> OdometryConstPtr;
typedef std::shared_ptr<const String> StringConstPtr;
} // namespace msg

class SurfelMap {
  public:
    struct Params { // nh.getParam(...) of surfel_map.cpp:13-28
        int cam_width = 0, cam_height = 0;
        float cam_fx = 0, cam_fy = 0, cam_cx = 0, cam_cy = 0;
        float fuse_far_distence = 30.f, fuse_near_distence = 0.5f; // kitti_orb.launch:15-16
        int drift_free_poses = 10;
        bool rgbd = false;
        int device = 0;
        int surfel_capacity = 0;
        int max_buffered_frames = 0; // frames kept waiting for a pose: 0 = 5000 (ros_node.cpp:24-25), < 0 = unbounded (dsm_surfel_map.h)
    };

    explicit SurfelMap(const Params &p) {
        dsm_surfel_map_config c = dsm_surfel_map_config();
        c.struct_size = sizeof c;
        c.cam_width = p.cam_width;
        c.cam_height = p.cam_height;
        c.cam_fx = p.cam_fx;
        c.cam_fy = p.cam_fy;
        c.cam_cx = p.cam_cx;
        c.cam_cy = p.cam_cy;
        c.fuse_far_distence = p.fuse_far_distence;
        c.fuse_near_distence = p.fuse_near_distence;
        c.drift_free_poses = p.drift_free_poses;
        c.rgbd = p.rgbd ? 1 : 0;
        c.device = p.device;
        c.surfel_capacity = p.surfel_capacity;
        c.max_buffered_frames = p.max_buffered_frames;
        const int rc = dsm_surfel_map_create(&c, &m_);
        if (rc != DSM_OK) {
            m_ = nullptr;
#ifndef DSM_NO_EXCEPTIONS
            throw std::runtime_error(std::string("dsm::SurfelMap: ") + dsm_last_error(nullptr));
#endif
        }
    }
    SurfelMap(const SurfelMap &) = delete;
    SurfelMap &operator=(const SurfelMap &) = delete;
    ~SurfelMap() { dsm_surfel_map_destroy(m_); }

    template <typename ImagePtr> int image_input(const ImagePtr &image_input) {
        return check(dsm_surfel_map_image_input(m_, stamp_of(image_input->header.stamp), (int32_t)image_input->width,
                                                (int32_t)image_input->height, (size_t)image_input->step,
                                                image_input->encoding.c_str(), image_input->data.data()));
    }
    template <typename ImagePtr> int depth_input(const ImagePtr &depth_input) {
        return check(dsm_surfel_map_depth_input(m_, stamp_of(depth_input->header.stamp), (int32_t)depth_input->width,
                                                (int32_t)depth_input->height, (size_t)depth_input->step,
                                                depth_input->encoding.c_str(), depth_input->data.data()));
    }
    template <typename PointCloudPtr, typename PathPtr, typename OdometryPtr>
    int orb_results_input(const PointCloudPtr &loop_stamp_input, const PathPtr &loop_path_input, const OdometryPtr &this_pose_input) {
        std::vector<dsm_pose_msg> path(loop_path_input->poses.size());
        for (size_t i = 0; i < path.size(); i++) path[i] = pose_of(loop_path_input->poses[i].pose);
        const dsm_pose_msg this_pose = pose_of(this_pose_input->pose.pose);
        double cov[36];
        for (int i = 0; i < 36; i++) cov[i] = this_pose_input->pose.covariance[i];
        static const float none = 0.f;
        const float *values = &none;
        int32_t n_values = 0;
        if (!loop_stamp_input->channels.empty() && !loop_stamp_input->channels[0].values.empty()) {
            values = loop_stamp_input->channels[0].values.data();
            n_values = (int32_t)loop_stamp_input->channels[0].values.size();
        }
        return check(dsm_surfel_map_orb_results_input(m_, stamp_of(loop_stamp_input->header.stamp), values, n_values, path.data(),
                                                      (int32_t)path.size(), stamp_of(this_pose_input->header.stamp), &this_pose, cov));
    }
    template <typename StringPtr> int save_map(const StringPtr &save_map_input) { return save_mesh(save_map_input->data); }
    int save_cloud(const std::string &save_path_name) { return check(dsm_surfel_map_save_cloud(m_, save_path_name.c_str())); }
    int save_mesh(const std::string &save_path_name) { return check(dsm_surfel_map_save_mesh(m_, save_path_name.c_str())); }

    dsm_surfel_map *handle() const { return m_; }
    dsm_handle *engine() const { return dsm_surfel_map_engine(m_); }

  private:
    template <typename T> static dsm_stamp stamp_of(const T &t) {
        dsm_stamp s;
        s.sec = (uint32_t)t.sec;
        s.nsec = (uint32_t)t.nsec;
        return s;
    }
    template <typename P> static dsm_pose_msg pose_of(const P &p) {
        dsm_pose_msg o;
        o.px = p.position.x; o.py = p.position.y; o.pz = p.position.z;
        o.qx = p.orientation.x; o.qy = p.orientation.y; o.qz = p.orientation.z; o.qw = p.orientation.w;
        return o;
    }
    int check(int rc) {
#if !defined(DSM_NO_EXCEPTIONS) && !defined(DSM_WITH_ROS)
        if (rc != DSM_OK) throw std::runtime_error(std::string("dsm::SurfelMap: ") + dsm_surfel_map_last_error(m_));
#endif
        return rc; // (with DSM_WITH_ROS the callbacks report and carry on, as the reference's void callbacks do)
    }
    dsm_surfel_map *m_ = nullptr;
};

} // namespace dsm

#ifdef DSM_WITH_ROS
// ---- the reference's class, signature for signature (surfel_map.h:48-62), for roscpp builds ------------------------
// `SurfelMap surfel_map(nh);` and the subscriber wiring of surfel_fusion/src/ros_node.cpp:22-32
//     nh.subscribe("image", 5000, &SurfelMap::image_input, &surfel_map);
//     sync.registerCallback(boost::bind(&SurfelMap::orb_results_input, &surfel_map, _1, _2, _3));
// bind to these members unchanged: they are plain (non-template, non-overloaded) member functions returning void.
// include/ros_compat/surfel_map.h puts this class behind the reference's own header name, so that ros_node.cpp
// compiles without an edit when that directory precedes the reference's src/ on the include path.
// Errors: the reference returns void and prints (surfel_map.cpp:31-32); so does this class (stderr), except that a
// constructor that cannot reach a gfx950 device throws std::runtime_error -- there is no CPU path to fall back to.
#include <cstdio>

#include <nav_msgs/Odometry.h>
#include <nav_msgs/Path.h>
#include <ros/ros.h>
#include <sensor_msgs/Image.h>
#include <sensor_msgs/PointCloud.h>
#include <std_msgs/String.h>

class SurfelMap {
  public:
    SurfelMap(ros::NodeHandle &_nh) : nh(_nh), impl_(params_from(_nh)) {}
    ~SurfelMap() {}

    void image_input(const sensor_msgs::ImageConstPtr &image_input) { report(impl_.image_input(image_input), "image_input"); }
    void depth_input(const sensor_msgs::ImageConstPtr &image_input) { report(impl_.depth_input(image_input), "depth_input"); }
    void orb_results_input(const sensor_msgs::PointCloudConstPtr &loop_stamp_input, const nav_msgs::PathConstPtr &loop_path_input,
                           const nav_msgs::OdometryConstPtr &this_pose_input) {
        report(impl_.orb_results_input(loop_stamp_input, loop_path_input, this_pose_input), "orb_results_input");
    }
    void save_cloud(std::string save_path_name) { report(impl_.save_cloud(save_path_name), "save_cloud"); }
    void save_mesh(std::string save_path_name) { report(impl_.save_mesh(save_path_name), "save_mesh"); }
    void save_map(const std_msgs::StringConstPtr &save_map_input) { // surfel_map.cpp:75-81
        std::string save_name = save_map_input->data;
        printf("save mesh modelt to %s.\n", save_name.c_str());
        save_mesh(save_name);
        printf("save done!\n");
    }

    dsm::SurfelMap &engine_node() { return impl_; }

  private:
    // the nine parameters of surfel_map.cpp:14-29
    static dsm::SurfelMap::Params params_from(ros::NodeHandle &nh) {
        dsm::SurfelMap::Params p;
        bool get_all = true;
        get_all &= nh.getParam("cam_width", p.cam_width);
        get_all &= nh.getParam("cam_height", p.cam_height);
        get_all &= nh.getParam("cam_fx", p.cam_fx);
        get_all &= nh.getParam("cam_cx", p.cam_cx);
        get_all &= nh.getParam("cam_fy", p.cam_fy);
        get_all &= nh.getParam("cam_cy", p.cam_cy);
        get_all &= nh.getParam("fuse_far_distence", p.fuse_far_distence);
        get_all &= nh.getParam("fuse_near_distence", p.fuse_near_distence);
        get_all &= nh.getParam("drift_free_poses", p.drift_free_poses);
        if (!get_all) printf("ERROR! Do not have enough parameters!");
        else printf("fuse the distence between %4f m and %4f m.\n", p.fuse_near_distence, p.fuse_far_distence);
        return p;
    }
    void report(int rc, const char *what) {
        if (rc != DSM_OK) fprintf(stderr, "SurfelMap::%s: %s\n", what, dsm_surfel_map_last_error(impl_.handle()));
    }
    ros::NodeHandle &nh;
    dsm::SurfelMap impl_;
};
#endif // DSM_WITH_ROS
#endif
