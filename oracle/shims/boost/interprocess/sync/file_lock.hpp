// Oracle shim: surfel_map.h:11 includes this header and uses nothing from it.
#pragma once
