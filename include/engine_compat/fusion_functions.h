// engine_compat/fusion_functions.h -- `#include <fusion_functions.h>` of the reference's surfel_map.h:30 resolved to
// the MI355X-native engine facade: with this directory BEFORE the reference's src/ on the include path (and
// libdsm_hip.so linked), surfel_fusion/src/surfel_map.{h,cpp} compile unchanged and `FusionFunctions
// fusion_functions` (surfel_map.h:118) is the HIP engine.  Provides what the reference's header provides to its
// includer: Eigen, OpenCV, elements.h (fusion_functions.h:1-5) and the class name.
#ifndef DSM_ENGINE_COMPAT_FUSION_FUNCTIONS_H
#define DSM_ENGINE_COMPAT_FUSION_FUNCTIONS_H
#include <Eigen/Eigen>
#include <opencv2/opencv.hpp>

#include <elements.h>

#include "../dsm_fusion_functions.hpp"

typedef dsm::FusionFunctions FusionFunctions;
#endif
