// Oracle shim (test infrastructure): cv_bridge::toCvCopy for messages that already have the requested
// encoding (SM.cpp:86,96 ask for MONO8 / TYPE_32FC1; the driver only sends those) -- a deep copy.
#pragma once
#include <cstring>
#include "opencv2/opencv.hpp"
#include "sensor_msgs/Image.h"
namespace cv_bridge {
struct CvImage {
    std_msgs::Header header;
    std::string encoding;
    cv::Mat image;
};
typedef boost::shared_ptr<CvImage> CvImagePtr;
inline CvImagePtr toCvCopy(const sensor_msgs::ImageConstPtr &src, const std::string &encoding) {
    if (src->encoding != encoding) abort();
    CvImagePtr out(new CvImage);
    out->header = src->header;
    out->encoding = encoding;
    const bool f32 = encoding == "32FC1";
    out->image = cv::Mat((int)src->height, (int)src->width, f32 ? CV_32FC1 : CV_8UC1);
    const size_t row = (size_t)src->width * (f32 ? 4 : 1);
    for (uint32_t y = 0; y < src->height; y++) memcpy(out->image.data + (size_t)y * out->image.step, &src->data[(size_t)y * src->step], row);
    return out;
}
}
