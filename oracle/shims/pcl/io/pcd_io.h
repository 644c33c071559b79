#pragma once
#include "pcl_ros/point_cloud.h"
