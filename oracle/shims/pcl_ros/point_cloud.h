// Oracle shim (test infrastructure): pcl::PointXYZI / pcl::PointCloud as a std::vector with the members
// surfel_map.cpp uses, pcl_conversions::toPCL, and pcl::io::savePCDFile's ASCII layout.  PCL is an
// un-vendored, un-pinned dependency of the reference (surfel_fusion/CMakeLists.txt); the writer below
// restates the published PCD v0.7 ASCII format of pcl::PCDWriter::writeASCII (precision 8, classic locale,
// "nan" for NaNs, one trimmed line per point) -- "parity unpinned" on the bytes of that file.
#pragma once
#include <cmath>
#include <fstream>
#include <locale>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include "ros/ros.h"

namespace pcl {
struct PCLHeader {
    uint32_t seq;
    uint64_t stamp;
    std::string frame_id;
    PCLHeader() : seq(0), stamp(0) {}
};
struct PointXYZI {
    float x, y, z, w_;
    float intensity;
    PointXYZI() : x(0), y(0), z(0), w_(1.f), intensity(0) {}
};
template <typename PointT> class PointCloud {
public:
    typedef boost::shared_ptr<PointCloud<PointT> > Ptr;
    typedef typename std::vector<PointT>::iterator iterator;
    PCLHeader header;
    std::vector<PointT> points;
    uint32_t width, height;
    PointCloud() : width(0), height(0) {}
    size_t size() const { return points.size(); }
    void reserve(size_t n) { points.reserve(n); }
    void push_back(const PointT &p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    iterator begin() { return points.begin(); }
    iterator end() { return points.end(); }
    iterator erase(iterator a, iterator b) { iterator r = points.erase(a, b); width = (uint32_t)points.size(); height = 1; return r; }
    template <typename It> void insert(iterator pos, It a, It b) { points.insert(pos, a, b); width = (uint32_t)points.size(); height = 1; }
    PointT &at(size_t i) { return points.at(i); }
    PointT &front() { return points.front(); }
    PointT &back() { return points.back(); }
    PointCloud &operator+=(const PointCloud &o) {
        points.insert(points.end(), o.points.begin(), o.points.end());
        width = (uint32_t)points.size();
        height = 1;
        return *this;
    }
};
namespace io {
template <typename PointT> int savePCDFile(const std::string &file_name, const PointCloud<PointT> &cloud) {
    if (cloud.points.empty()) throw std::runtime_error("[pcl::PCDWriter::writeASCII] Input point cloud has no data!");
    std::ofstream fs(file_name.c_str());
    if (!fs.is_open()) throw std::runtime_error("[pcl::PCDWriter::writeASCII] Could not open file for writing!");
    fs.precision(8);
    fs.imbue(std::locale::classic());
    const size_t n = cloud.points.size();
    fs << "# .PCD v0.7 - Point Cloud Data file format\n"
       << "VERSION 0.7\n"
       << "FIELDS x y z intensity\n"
       << "SIZE 4 4 4 4\n"
       << "TYPE F F F F\n"
       << "COUNT 1 1 1 1\n"
       << "WIDTH " << n << "\n"
       << "HEIGHT 1\n"
       << "VIEWPOINT 0 0 0 1 0 0 0\n"
       << "POINTS " << n << "\n"
       << "DATA ascii\n";
    std::ostringstream stream;
    stream.precision(8);
    stream.imbue(std::locale::classic());
    for (size_t i = 0; i < n; i++) {
        const float f[4] = {cloud.points[i].x, cloud.points[i].y, cloud.points[i].z, cloud.points[i].intensity};
        for (int d = 0; d < 4; d++) {
            if (std::isnan(f[d])) stream << "nan";
            else stream << f[d];
            if (d < 3) stream << " ";
        }
        fs << stream.str() << "\n";
        stream.str("");
    }
    fs.close();
    return 0;
}
}  // namespace io
}  // namespace pcl

namespace pcl_conversions {
inline void toPCL(const ros::Time &t, uint64_t &pcl_stamp) { pcl_stamp = (uint64_t)t.sec * 1000000ull + t.nsec / 1000ull; }
}
