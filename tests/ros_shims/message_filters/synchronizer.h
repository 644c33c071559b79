// TEST INFRASTRUCTURE: message_filters::Synchronizer over three topics.  The recorded message logs hold the three
// ORB-SLAM messages of one stamp as one record, so "synchronisation" is delivery of that triple.
#pragma once
#include "message_filters/subscriber.h"

namespace message_filters {
template <class Policy> struct Synchronizer {
    typedef boost::shared_ptr<const typename Policy::M0> P0;
    typedef boost::shared_ptr<const typename Policy::M1> P1;
    typedef boost::shared_ptr<const typename Policy::M2> P2;
    typedef std::tuple<P0, P1, P2> Triple;
    std::string key;
    template <class S0, class S1, class S2> Synchronizer(const Policy &, S0 &a, S1 &b, S2 &c) : key("sync:" + a.topic + "|" + b.topic + "|" + c.topic) {}
    template <class F> void registerCallback(const F &f) {
        std::function<void(const P0 &, const P1 &, const P2 &)> fn = f;
        ros::shim::Bus::get().handlers[key] = [fn](const std::shared_ptr<const void> &p) {
            const Triple &t = *std::static_pointer_cast<const Triple>(p);
            fn(std::get<0>(t), std::get<1>(t), std::get<2>(t));
        };
    }
};
}  // namespace message_filters
