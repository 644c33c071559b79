// Oracle shim (test infrastructure): boost::shared_ptr as the node uses it (message ConstPtrs, PointCloud::Ptr).
#pragma once
#include <memory>
namespace boost {
template <typename T> using shared_ptr = std::shared_ptr<T>;
}
