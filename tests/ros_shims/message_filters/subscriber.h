// TEST INFRASTRUCTURE: the slice of message_filters that surfel_fusion/src/ros_node.cpp:27-31 uses.
#pragma once
#include <functional>
#include <string>
#include <tuple>

#include "ros/ros.h"

namespace message_filters {
template <class M> struct Subscriber {
    std::string topic;
    Subscriber(ros::NodeHandle &, const std::string &t, uint32_t) : topic(t) {}
};
}  // namespace message_filters

// boost::bind with the global placeholders _1 .. _3, as roscpp's headers bring them in
namespace boost {
using std::bind;
}
using namespace std::placeholders;
