// TEST INFRASTRUCTURE: sync policy tag (see synchronizer.h)
#pragma once
#include <cstdint>
namespace message_filters {
namespace sync_policies {
template <class A, class B, class C> struct ApproximateTime {
    typedef A M0;
    typedef B M1;
    typedef C M2;
    explicit ApproximateTime(uint32_t) {}
};
}  // namespace sync_policies
}  // namespace message_filters
