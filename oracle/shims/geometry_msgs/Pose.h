// Oracle shim (test infrastructure): stands in for <geometry_msgs/Pose.h>
// (surfel_fusion/src/elements.h:3); nothing from it is used on the hot path.
#pragma once
