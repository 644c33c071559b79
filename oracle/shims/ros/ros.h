// Oracle shim (test infrastructure): the slice of roscpp that surfel_fusion/src/surfel_map.{h,cpp} touch.
// fusion_functions.cpp only needs the header to exist (elements.h:2).  No transport: NodeHandle serves
// parameters from a map filled by the driver, Publisher::publish drops the message.
//
// The real header transitively provides <math.h>, which is what makes fabs(float) resolve to the float
// overload in fusion_functions.cpp -- keep it.
#pragma once
#include <math.h>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
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

// Message pump for builds of the reference's ros_node.cpp (tests/ros_shims/ros_shim_bus.cpp fills it from a recorded
// message log): topic -> type-erased handler, as registered through NodeHandle::subscribe / message_filters.
namespace shim {
struct Bus {
    std::map<std::string, std::function<void(const std::shared_ptr<const void> &)> > handlers;
    std::function<bool()> pump; // delivers the next recorded message; false at the end of the log
    bool alive;
    Bus() : alive(true) {}
    static Bus &get() {
        static Bus b;
        return b;
    }
};
}  // namespace shim

struct Subscriber {};

struct NodeHandle {
    std::map<std::string, double> params;
    std::map<std::string, std::string> string_params;
    NodeHandle() {}
    explicit NodeHandle(const std::string &ns); // defined by the message-log pump (tests/ros_shims/ros_shim_bus.cpp)
    template <typename T> bool getParam(const std::string &key, T &out) const {
        std::map<std::string, double>::const_iterator it = params.find(key);
        if (it == params.end()) return false;
        out = (T)it->second;
        return true;
    }
    bool getParam(const std::string &key, std::string &out) const {
        std::map<std::string, std::string>::const_iterator it = string_params.find(key);
        if (it == string_params.end()) return false;
        out = it->second;
        return true;
    }
    template <typename M> Publisher advertise(const std::string &, int) { return Publisher(); }
    // roscpp's member-function overload: M is deduced from the callback's parameter
    template <class M, class T>
    Subscriber subscribe(const std::string &topic, uint32_t, void (T::*fp)(const boost::shared_ptr<M const> &), T *obj) {
        shim::Bus::get().handlers[topic] = [fp, obj](const std::shared_ptr<const void> &p) { (obj->*fp)(std::static_pointer_cast<const M>(p)); };
        return Subscriber();
    }
};

inline void init(int &, char **, const std::string &) {}
inline bool ok() { return shim::Bus::get().alive; }
inline void spinOnce() {
    shim::Bus &b = shim::Bus::get();
    if (!b.pump || !b.pump()) b.alive = false;
}

}  // namespace ros

namespace std_msgs {
struct Header {
    uint32_t seq;
    ros::Time stamp;
    std::string frame_id;
    Header() : seq(0) {}
};
}  // namespace std_msgs
