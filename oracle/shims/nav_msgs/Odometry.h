// Oracle shim (test infrastructure): pose.covariance[0] > 0 = new keyframe, [1] = reference keyframe (SM.cpp:320,337,356)
#pragma once
#include "geometry_msgs/Pose.h"
namespace nav_msgs {
struct Odometry {
    std_msgs::Header header;
    std::string child_frame_id;
    geometry_msgs::PoseWithCovariance pose;
    geometry_msgs::TwistWithCovariance twist;
};
typedef boost::shared_ptr<const Odometry> OdometryConstPtr;
}
