// Oracle shim: surfel_map.h:10 includes this header and uses nothing from it.
#pragma once
