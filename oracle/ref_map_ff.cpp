// ORACLE / TEST INFRASTRUCTURE: the reference's fusion_functions.cpp compiled in place as the second
// translation unit of oracle/_ref/libdsm_ref_map.so (see ref_map_driver.cpp).
#include "fusion_functions.cpp"  // resolved via -I/root/reference/surfel_fusion/src
