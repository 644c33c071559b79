// Oracle shim (test infrastructure): sensor_msgs/Image
#pragma once
#include "ros/ros.h"
namespace sensor_msgs {
struct Image {
    std_msgs::Header header;
    uint32_t height, width;
    std::string encoding;
    uint8_t is_bigendian;
    uint32_t step;
    std::vector<uint8_t> data;
    Image() : height(0), width(0), is_bigendian(0), step(0) {}
};
typedef boost::shared_ptr<const Image> ImageConstPtr;
}
