// ORACLE / TEST INFRASTRUCTURE -- never linked into or called by the product path.
//
// C-callable driver around the reference's own node class: /root/reference/surfel_fusion/src/surfel_map.cpp
// is compiled in place (never copied) by #include-ing it below, with oracle/shims/ in front of the
// reference's src/ directory on the include path (ROS messages, cv_bridge, PCL containers, Eigen, boost:
// storage and the few published algorithms listed in the shim headers).  fusion_functions.cpp is compiled
// the same way by ref_map_ff.cpp.  Built by oracle/Makefile into oracle/_ref/ only.
//
// Schedule: DSM_ORACLE_DEFERRED_THREADS (shims/opencv2/opencv.hpp) -- workers run in index order when joined.
// What is reference code here: every SurfelMap method.  What is stubbed: CameraPoseVisualization (RViz
// markers for publish_camera_position, SM.cpp:906-922; its .cpp is not compiled) and message transport.
#include "surfel_map.cpp"  // resolved via -I/root/reference/surfel_fusion/src

#include <cstdint>
#include <cstring>

CameraPoseVisualization::CameraPoseVisualization(float, float, float, float) : m_scale(0), m_line_width(0) {}
void CameraPoseVisualization::setScale(double s) { m_scale = s; }
void CameraPoseVisualization::setLineWidth(double w) { m_line_width = w; }
void CameraPoseVisualization::add_pose(const Eigen::Vector3d &, const Eigen::Quaterniond &) {}
void CameraPoseVisualization::publish_by(ros::Publisher &, ros::Time &) {}

namespace {
struct RefMap {
    ros::NodeHandle nh;
    SurfelMap *map;
};

geometry_msgs::Pose pose_from(const double *p) {  // px py pz qx qy qz qw
    geometry_msgs::Pose o;
    o.position.x = p[0]; o.position.y = p[1]; o.position.z = p[2];
    o.orientation.x = p[3]; o.orientation.y = p[4]; o.orientation.z = p[5]; o.orientation.w = p[6];
    return o;
}
void pose_to(const geometry_msgs::Pose &o, double *p) {
    p[0] = o.position.x; p[1] = o.position.y; p[2] = o.position.z;
    p[3] = o.orientation.x; p[4] = o.orientation.y; p[5] = o.orientation.z; p[6] = o.orientation.w;
}
sensor_msgs::ImageConstPtr image_msg(uint32_t sec, uint32_t nsec, int w, int h, size_t step, const char *enc, const void *data) {
    sensor_msgs::Image *m = new sensor_msgs::Image;
    m->header.stamp = ros::Time(sec, nsec);
    m->width = (uint32_t)w;
    m->height = (uint32_t)h;
    m->step = (uint32_t)step;
    m->encoding = enc;
    m->data.assign((const uint8_t *)data, (const uint8_t *)data + step * (size_t)h);
    return sensor_msgs::ImageConstPtr(m);
}
}  // namespace

extern "C" {

void *refmap_create(int w, int h, float fx, float fy, float cx, float cy, float far_d, float near_d, int drift_free_poses) {
    RefMap *r = new RefMap();
    r->nh.params["cam_width"] = w;
    r->nh.params["cam_height"] = h;
    r->nh.params["cam_fx"] = fx;
    r->nh.params["cam_fy"] = fy;
    r->nh.params["cam_cx"] = cx;
    r->nh.params["cam_cy"] = cy;
    r->nh.params["fuse_far_distence"] = far_d;
    r->nh.params["fuse_near_distence"] = near_d;
    r->nh.params["drift_free_poses"] = drift_free_poses;
    r->map = new SurfelMap(r->nh);
    return r;
}
void refmap_destroy(void *hv) {
    RefMap *r = (RefMap *)hv;
    delete r->map;
    delete r;
}

void refmap_image_input(void *hv, uint32_t sec, uint32_t nsec, int w, int h, size_t step, const uint8_t *data) {
    ((RefMap *)hv)->map->image_input(image_msg(sec, nsec, w, h, step, "mono8", data));
}
void refmap_depth_input(void *hv, uint32_t sec, uint32_t nsec, int w, int h, size_t step, const float *data) {
    ((RefMap *)hv)->map->depth_input(image_msg(sec, nsec, w, h, step, "32FC1", data));
}
void refmap_orb_results_input(void *hv, uint32_t loop_sec, uint32_t loop_nsec, const float *values, int n_values, const double *path7,
                              int n_path, uint32_t this_sec, uint32_t this_nsec, const double *pose7, const double *cov36) {
    sensor_msgs::PointCloud *ls = new sensor_msgs::PointCloud;
    ls->header.stamp = ros::Time(loop_sec, loop_nsec);
    ls->channels.resize(1);
    ls->channels[0].values.assign(values, values + n_values);
    nav_msgs::Path *lp = new nav_msgs::Path;
    lp->header.stamp = ls->header.stamp;
    lp->poses.resize((size_t)n_path);
    for (int i = 0; i < n_path; i++) lp->poses[(size_t)i].pose = pose_from(path7 + 7 * i);
    nav_msgs::Odometry *od = new nav_msgs::Odometry;
    od->header.stamp = ros::Time(this_sec, this_nsec);
    od->pose.pose = pose_from(pose7);
    for (int i = 0; i < 36; i++) od->pose.covariance[i] = cov36[i];
    ((RefMap *)hv)->map->orb_results_input(sensor_msgs::PointCloudConstPtr(ls), nav_msgs::PathConstPtr(lp), nav_msgs::OdometryConstPtr(od));
}

int refmap_pending_poses(void *hv) { return (int)((RefMap *)hv)->map->pose_reference_buffer.size(); }
int refmap_local_count(void *hv) { return (int)((RefMap *)hv)->map->local_surfels.size(); }
void refmap_get_local(void *hv, SurfelElement *out) {
    SurfelMap *m = ((RefMap *)hv)->map;
    if (!m->local_surfels.empty()) memcpy(out, m->local_surfels.data(), sizeof(SurfelElement) * m->local_surfels.size());
}
int refmap_pose_count(void *hv) { return (int)((RefMap *)hv)->map->poses_database.size(); }
void refmap_get_pose(void *hv, int i, double *cam7, double *loop7, int *n_attached, int *points_begin_index, int *is_local) {
    SurfelMap *m = ((RefMap *)hv)->map;
    PoseElement &pe = m->poses_database[(size_t)i];
    pose_to(pe.cam_pose, cam7);
    pose_to(pe.loop_pose, loop7);
    *n_attached = (int)pe.attached_surfels.size();
    *points_begin_index = pe.points_begin_index;
    *is_local = m->local_surfels_indexs.count(i) ? 1 : 0;
}
int refmap_get_links(void *hv, int i, int *out, int cap) {
    std::vector<int> &l = ((RefMap *)hv)->map->poses_database[(size_t)i].linked_pose_index;
    for (size_t k = 0; k < l.size() && k < (size_t)cap; k++) out[k] = l[k];
    return (int)l.size();
}
void refmap_get_attached(void *hv, int i, SurfelElement *out) {
    std::vector<SurfelElement> &a = ((RefMap *)hv)->map->poses_database[(size_t)i].attached_surfels;
    if (!a.empty()) memcpy(out, a.data(), sizeof(SurfelElement) * a.size());
}
int refmap_cloud_count(void *hv) { return (int)((RefMap *)hv)->map->inactive_pointcloud->size(); }
void refmap_get_cloud(void *hv, float *xyzi) {
    PointCloud &c = *((RefMap *)hv)->map->inactive_pointcloud;
    for (size_t i = 0; i < c.size(); i++) {
        xyzi[4 * i + 0] = c.points[i].x;
        xyzi[4 * i + 1] = c.points[i].y;
        xyzi[4 * i + 2] = c.points[i].z;
        xyzi[4 * i + 3] = c.points[i].intensity;
    }
}
int refmap_save_cloud(void *hv, const char *path) {
    try {
        ((RefMap *)hv)->map->save_cloud(path);
    } catch (const std::exception &) {
        return -1;
    }
    return 0;
}
void refmap_save_mesh(void *hv, const char *path) { ((RefMap *)hv)->map->save_mesh(path); }

}  // extern "C"
