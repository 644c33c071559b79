// Oracle shim (test infrastructure): sensor_msgs/PointCloud (loop_stamps: channels[0].values = keyframe index pairs)
#pragma once
#include "ros/ros.h"
namespace sensor_msgs {
struct ChannelFloat32 { std::string name; std::vector<float> values; };
struct Point32 { float x, y, z; };
struct PointCloud {
    std_msgs::Header header;
    std::vector<Point32> points;
    std::vector<ChannelFloat32> channels;
};
typedef boost::shared_ptr<const PointCloud> PointCloudConstPtr;
}
