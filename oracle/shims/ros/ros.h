// Oracle shim (test infrastructure): the slice of roscpp that surfel_fusion/src/surfel_map.{h,cpp} touch.
// fusion_functions.cpp only needs the header to exist (elements.h:2).  No transport: NodeHandle serves
// parameters from a map filled by the driver, Publisher::publish drops the message.
//
// The real header transitively provides <math.h>, which is what makes fabs(float) resolve to the float
// overload in fusion_functions.cpp -- keep it.
#pragma once
#include <math.h>
#include <cstdint>
#include <map>
#include <string>
#include <vector>
#include "boost/shared_ptr.hpp"

namespace ros {

struct Time {  // ros::TimeBase: toSec() = sec + 1e-9 * nsec
    uint32_t sec, nsec;
    Time() : sec(0), nsec(0) {}
    Time(uint32_t s, uint32_t n) : sec(s), nsec(n) {}
    double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
};

struct Publisher {
    template <typename M> void publish(const M &) const {}
};

struct NodeHandle {
    std::map<std::string, double> params;
    template <typename T> bool getParam(const std::string &key, T &out) const {
        std::map<std::string, double>::const_iterator it = params.find(key);
        if (it == params.end()) return false;
        out = (T)it->second;
        return true;
    }
    template <typename M> Publisher advertise(const std::string &, int) { return Publisher(); }
};

}  // namespace ros

namespace std_msgs {
struct Header {
    uint32_t seq;
    ros::Time stamp;
    std::string frame_id;
    Header() : seq(0) {}
};
}  // namespace std_msgs
