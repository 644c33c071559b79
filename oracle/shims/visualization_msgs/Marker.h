// Oracle shim (test infrastructure): markers are built by the publish_* methods and dropped by Publisher.
#pragma once
#include "geometry_msgs/Pose.h"
#include "std_msgs/ColorRGBA.h"
namespace visualization_msgs {
struct Marker {
    enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, CUBE_LIST = 6, SPHERE_LIST = 7, POINTS = 8 };
    enum { ADD = 0, MODIFY = 0, DELETE = 2 };
    std_msgs::Header header;
    std::string ns;
    int32_t id, type, action;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    std_msgs::ColorRGBA color;
    std::vector<geometry_msgs::Point> points;
    std::vector<std_msgs::ColorRGBA> colors;
    Marker() : id(0), type(0), action(0) {}
};
}
