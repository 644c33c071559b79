// ORACLE / TEST INFRASTRUCTURE: the reference's fusion_functions.cpp compiled in place as the second
// translation unit of oracle/_ref/libdsm_ref_map.so (see ref_map_driver.cpp).
#include "fusion_functions.h"
#ifdef DSM_REF_RGBD
// The reference keeps its RGB-D constant set as comments (fusion_functions.h:17-21).
#undef HUBER_RANGE
#undef BASELINE
#undef DISPARITY_ERROR
#undef MIN_TOLERATE_DIFF
#define HUBER_RANGE 0.05
#define BASELINE 0.08
#define DISPARITY_ERROR 1.0
#define MIN_TOLERATE_DIFF 0.05
#endif
#include "fusion_functions.cpp"  // resolved via -I/root/reference/surfel_fusion/src
