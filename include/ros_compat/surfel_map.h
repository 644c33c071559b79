// ros_compat/surfel_map.h -- `#include <surfel_map.h>` of surfel_fusion/src/ros_node.cpp:15 resolved to the
// MI355X-native node class.  Put this directory BEFORE the reference's src/ on the include path (and link
// libdsm_hip.so): ros_node.cpp then compiles unchanged -- `SurfelMap surfel_map(nh)`, the three nh.subscribe /
// message_filters bindings and the save calls of ros_node.cpp:22-51 all bind to the class below.
#ifndef DSM_ROS_COMPAT_SURFEL_MAP_H
#define DSM_ROS_COMPAT_SURFEL_MAP_H
#ifndef DSM_WITH_ROS
#define DSM_WITH_ROS 1
#endif
#include <cstdlib>
#include <string>

#include "../dsm_surfel_map.hpp"

using std::string; // ros_node.cpp uses the unqualified name (the reference's surfel_map.h:32 says `using namespace std`)
#endif
