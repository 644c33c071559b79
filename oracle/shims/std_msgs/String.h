// Oracle shim (test infrastructure)
#pragma once
#include "ros/ros.h"
namespace std_msgs {
struct String { std::string data; };
typedef boost::shared_ptr<const String> StringConstPtr;
}
