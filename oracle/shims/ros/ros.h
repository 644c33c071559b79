// Oracle shim (test infrastructure): stands in for <ros/ros.h>, which
// surfel_fusion/src/elements.h:2 includes but whose symbols the fusion hot
// path never uses.  The real header transitively provides <math.h>, which is
// what makes fabs(float) resolve to the float overload.
#pragma once
#include <math.h>
