// Oracle shim (test infrastructure)
#pragma once
#include <string>
namespace sensor_msgs { namespace image_encodings {
const std::string MONO8 = "mono8";
const std::string TYPE_32FC1 = "32FC1";
} }
