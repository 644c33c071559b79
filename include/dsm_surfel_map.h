/* dsm_surfel_map.h -- the node-level interface of DenseSurfelMapping's `surfel_fusion` without ROS.
 *
 * Mirrors class SurfelMap (reference surfel_fusion/src/surfel_map.h:48-147): the three callbacks the node
 * wires to its subscribers (ros_node.cpp:24-32) plus save_cloud / save_mesh / save_map, with the ROS
 * message types replaced by plain C structs that carry the same fields the callbacks read.  Everything
 * the callbacks do between the messages and the per-frame engine -- exact-stamp matching
 * (synchronize_msgs, surfel_map.cpp:103-203), the KITTI axis transform and pose-graph bookkeeping
 * (orb_results_input, :205-365), the drift-free window (get_driftfree_poses / get_add_remove_poses,
 * :1597-1673), moving keyframes' surfels between the active map and the inactive set
 * (move_add_surfels, :1456-1595) and the loop-closure deformation (warp_surfels, :681-824) -- is host
 * logic in this library; the surfels themselves never leave HBM: the active map is the resident map of
 * a dsm_handle (include/dsm.h) and the inactive set is its device-side store (dsm_store_*).
 *
 * Not mirrored: the publish_* methods (RViz markers / point-cloud topics, surfel_map.cpp:906-1058,
 * 1115-1151, 1283-1454) -- read the same data through the taps at the end of this header.
 *
 * Errors: the reference returns void and prints; these return a dsm_status (include/dsm.h) and keep a
 * message for dsm_surfel_map_last_error.  Inputs on which the reference indexes out of range (undefined
 * behaviour) are refused with DSM_E_INVALID instead. */
#ifndef DSM_SURFEL_MAP_H
#define DSM_SURFEL_MAP_H

#include <stddef.h>
#include <stdint.h>

#include "dsm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dsm_surfel_map dsm_surfel_map;

/* ros::Time: toSec() = sec + 1e-9 nsec; stamps are matched by exact equality of that double
 * (surfel_map.cpp:116-139). */
typedef struct dsm_stamp {
    uint32_t sec, nsec;
} dsm_stamp;

/* geometry_msgs::Pose */
typedef struct dsm_pose_msg {
    double px, py, pz;
    double qx, qy, qz, qw;
} dsm_pose_msg;

/* the node's parameters (surfel_map.cpp:13-28) */
typedef struct dsm_surfel_map_config {
    uint32_t struct_size;    /* sizeof(dsm_surfel_map_config) of the caller's header: a caller built against another
                                layout is refused (DSM_E_INVALID) instead of having trailing fields read from garbage */
    int32_t cam_width, cam_height;
    float cam_fx, cam_fy, cam_cx, cam_cy;
    float fuse_far_distence, fuse_near_distence; /* spelling of the reference's parameter names */
    int32_t drift_free_poses;
    int32_t rgbd;            /* constant set of fusion_functions.h:17-21 instead of :7-16 */
    int32_t device;          /* HIP device ordinal */
    int32_t surfel_capacity; /* active-map capacity, 0 = default of dsm_create */
    int32_t max_buffered_frames; /* images / depths kept waiting for a pose, oldest dropped (and reported on stderr)
                                    beyond; 0 = 5000, the depth of the reference's subscriber queues (ros_node.cpp:24-25;
                                    its own lists behind them are unbounded, surfel_map.h:96-97); < 0 = unbounded.  Only the
                                    first 256 waiting frames of each kind sit in page-locked memory, the rest is pageable */
} dsm_surfel_map_config;

int dsm_surfel_map_create(const dsm_surfel_map_config *cfg, dsm_surfel_map **out); /* SurfelMap::SurfelMap */
void dsm_surfel_map_destroy(dsm_surfel_map *m);
const char *dsm_surfel_map_last_error(const dsm_surfel_map *m);

/* SurfelMap::image_input (surfel_map.cpp:83-91): sensor_msgs/Image already in MONO8 (the reference
 * converts with cv_bridge; other encodings are refused here).  The pixels are copied. */
int dsm_surfel_map_image_input(dsm_surfel_map *m, dsm_stamp stamp, int32_t width, int32_t height, size_t step,
                               const char *encoding, const uint8_t *data);
/* SurfelMap::depth_input (:93-101): TYPE_32FC1, metres, 0 = invalid. */
int dsm_surfel_map_depth_input(dsm_surfel_map *m, dsm_stamp stamp, int32_t width, int32_t height, size_t step,
                               const char *encoding, const void *data);
/* SurfelMap::orb_results_input (:205-365).
 *   loop_stamp       header.stamp of the sensor_msgs/PointCloud (it becomes the fuse stamp, :363)
 *   loop_values      channels[0].values: flat pairs of keyframe indices, as float32
 *   loop_path        nav_msgs/Path poses (loop-corrected keyframe poses, SLAM frame)
 *   this_stamp       header.stamp of the nav_msgs/Odometry
 *   this_pose        pose.pose
 *   covariance       pose.covariance: [0] > 0 marks a new keyframe, [1] = reference keyframe index */
int dsm_surfel_map_orb_results_input(dsm_surfel_map *m, dsm_stamp loop_stamp, const float *loop_values,
                                     int32_t n_loop_values, const dsm_pose_msg *loop_path, int32_t n_loop_path,
                                     dsm_stamp this_stamp, const dsm_pose_msg *this_pose, const double *covariance36);

int dsm_surfel_map_save_cloud(dsm_surfel_map *m, const char *path); /* :1153-1174, ASCII PCD of XYZI points */
int dsm_surfel_map_save_mesh(dsm_surfel_map *m, const char *path);  /* :1176-1281, ASCII PLY, one hexagon per surfel */
int dsm_surfel_map_save_map(dsm_surfel_map *m, const char *path);   /* :75-81 = save_mesh */

/* ---- taps (what the publish_* methods read) ---- */
dsm_handle *dsm_surfel_map_engine(dsm_surfel_map *m); /* active map: dsm_map_size / dsm_map_download */
int64_t dsm_surfel_map_frames_fused(const dsm_surfel_map *m);
/* poses whose image or depth never arrived (a newer frame was already waiting): the reference spins forever on
 * these (surfel_map.cpp:114-139); here they are dropped so that later poses proceed */
int64_t dsm_surfel_map_dropped_poses(const dsm_surfel_map *m);
int32_t dsm_surfel_map_pose_count(const dsm_surfel_map *m);
/* poses_database[i]: cam_pose, loop_pose, number of attached (inactive) surfels, points_begin_index,
 * whether i is in local_surfels_indexs; any output may be NULL */
int dsm_surfel_map_get_pose(const dsm_surfel_map *m, int32_t i, dsm_pose_msg *cam_pose, dsm_pose_msg *loop_pose,
                            int32_t *n_attached, int32_t *points_begin_index, int32_t *is_local);
/* poses_database[i].linked_pose_index, in insertion order; returns the count (or a negative status) */
int32_t dsm_surfel_map_get_links(const dsm_surfel_map *m, int32_t i, int32_t *out, int32_t cap);
/* poses_database[i].attached_surfels */
int dsm_surfel_map_get_attached(dsm_surfel_map *m, int32_t i, dsm_surfel *out, int32_t cap, int32_t *n);
/* inactive_pointcloud: 4 floats per point (x, y, z, intensity) */
int dsm_surfel_map_get_inactive_cloud(dsm_surfel_map *m, float *xyzi_out, int32_t cap, int32_t *n);

#ifdef __cplusplus
}
#endif
#endif
