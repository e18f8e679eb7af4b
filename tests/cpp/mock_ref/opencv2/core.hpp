// mock of the part of cv::Mat / cv::Point3d the adapter uses (see glog/logging.h in this directory)
#pragma once
#include <cstdint>
#define CV_64FC1 6
#define CV_32SC1 4
namespace cv {
struct Point3d { double x, y, z; };
class Mat {
public:
    int rows = 0, cols = 0;
    bool isContinuous() const { return true; }
    int type() const { return type_; }
    template <class T> T* ptr() { return reinterpret_cast<T*>(data_); }
    template <class T> const T* ptr() const { return reinterpret_cast<const T*>(data_); }
private:
    int type_ = CV_64FC1; unsigned char* data_ = nullptr;
};
}  // namespace cv
