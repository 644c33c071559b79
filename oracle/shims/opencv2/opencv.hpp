// Oracle shim (test infrastructure): the slice of OpenCV that
// surfel_fusion/src/fusion_functions.cpp touches -- storage and accessors only,
// no arithmetic.  cv::Mat here is a non-owning view over a
// caller buffer; copies are shallow exactly as FF.cpp:45-46 relies on.
#pragma once
#include <math.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>
#include <cstddef>
#include <memory>

#define CV_8UC1 0
#define CV_8UC3 16
#define CV_32FC1 5

namespace cv {

typedef unsigned char uchar;

struct Vec3b {
    unsigned char v[3];
    Vec3b() { v[0] = v[1] = v[2] = 0; }
    Vec3b(unsigned char a, unsigned char b, unsigned char c) { v[0] = a; v[1] = b; v[2] = c; }
    unsigned char &operator[](int i) { return v[i]; }
};

class Mat {
public:
    int rows, cols;
    size_t step;          // bytes per row
    unsigned char *data;  // not owned unless scratch_ is used
    Mat() : rows(0), cols(0), step(0), data(nullptr) {}
    Mat(int r, int c, int type) : rows(r), cols(c) {
        size_t esz = (type == CV_8UC3) ? 3 : (type == CV_32FC1 ? 4 : 1);
        step = esz * (size_t)c;
        scratch_.reset(new std::vector<unsigned char>(step * (size_t)r));
        data = scratch_->data();
    }
    Mat(int r, int c, size_t step_bytes, void *ext) : rows(r), cols(c), step(step_bytes), data((unsigned char *)ext) {}
    template <typename T> T &at(int r, int c) { return *(T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
    template <typename T> const T &at(int r, int c) const { return *(const T *)(data + (size_t)r * step + (size_t)c * sizeof(T)); }
private:
    std::shared_ptr<std::vector<unsigned char>> scratch_;
};

inline void imshow(const char *, const Mat &) {}
inline int waitKey(int) { return 0; }

}  // namespace cv

// uchar is used unqualified inside cv::Mat::at<uchar> call sites (FF.cpp:402).
using cv::uchar;

#ifdef DSM_ORACLE_DEFERRED_THREADS
// Deterministic schedule for the node (surfel_map.cpp): a std::thread stand-in that runs its callable
// when it is joined.  SurfelMap::warp_surfels starts ten workers over the keyframes, then reads
// poses_database[local].cam_pose on the main thread, then starts ten more and joins all twenty
// (SM.cpp:791-824); the first ten overwrite cam_pose (SM.cpp:698-702,741), so running them inside the
// constructor would change what the main thread reads.  Run-at-join is the schedule in which the main
// thread gets there first; workers then execute in index order.  For fusion_functions.cpp (create all,
// then join all in order) it is the same index-order schedule as DSM_ORACLE_SERIAL_THREADS.
#include <thread>
#include <functional>
namespace std {
class dsm_serial_thread {
public:
    dsm_serial_thread() : pending_(false) {}
    template <class F, class... A> explicit dsm_serial_thread(F &&f, A &&...a)
        : f_(std::bind(std::forward<F>(f), std::forward<A>(a)...)), pending_(true) {}
    dsm_serial_thread(dsm_serial_thread &&o) : f_(std::move(o.f_)), pending_(o.pending_) { o.pending_ = false; }
    dsm_serial_thread(const dsm_serial_thread &) = delete;
    bool joinable() const { return pending_; }
    void join() {
        pending_ = false;
        f_();
    }
private:
    std::function<void()> f_;
    bool pending_;
};
}  // namespace std
#define thread dsm_serial_thread
#endif

#ifdef DSM_ORACLE_SERIAL_THREADS
// Deterministic schedule: a std::thread stand-in whose constructor runs the
// callable immediately on the calling thread.  Workers therefore execute in
// index order 0..THREAD_NUM-1, which is one legal schedule of the reference
// (SURVEY.md §8(c) "Is the real-threaded reference deterministic?").
#include <thread>
#include <functional>
namespace std {
class dsm_serial_thread {
public:
    dsm_serial_thread() {}
    template <class F, class... A> explicit dsm_serial_thread(F &&f, A &&...a) {
        std::bind(std::forward<F>(f), std::forward<A>(a)...)();
    }
    dsm_serial_thread(dsm_serial_thread &&) {}
    dsm_serial_thread(const dsm_serial_thread &) = delete;
    bool joinable() const { return false; }
    void join() {}
};
}  // namespace std
#define thread dsm_serial_thread
#endif

#ifdef DSM_ORACLE_QUIET
// The reference prints three timing lines per frame (FF.cpp:55,75,82).
static inline int dsm_oracle_noprintf(const char *, ...) { return 0; }
#define printf dsm_oracle_noprintf
#endif
