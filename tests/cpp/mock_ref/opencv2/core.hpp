// mock of the part of cv::Mat / cv::Point3d the adapter and tests/cpp/adapter_test.cpp use (see glog/logging.h in this directory): a
// reference-counted, continuous, row-major plane with the shallow-copy semantics of cv::Mat (Gx = Gx_new shares the buffer, solver.cpp:304).
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>
#define CV_64FC1 6
#define CV_32SC1 4
namespace cv {
struct Point3d { double x, y, z; };
class Mat {
public:
    int rows = 0, cols = 0;
    Mat() {}
    static Mat zeros(int r, int c, int type)
    {
        Mat m; m.rows = r; m.cols = c; m.type_ = type;
        m.buf_ = std::make_shared<std::vector<unsigned char>>((size_t)r * c * (type == CV_64FC1 ? 8 : 4), (unsigned char)0);
        return m;
    }
    Mat clone() const { Mat m = *this; if (buf_) m.buf_ = std::make_shared<std::vector<unsigned char>>(*buf_); return m; }
    void copyTo(Mat& dst) const     // like cv::Mat::copyTo: an existing destination of the same size and type keeps its buffer
    {
        if (dst.buf_ && buf_ && dst.buf_->size() == buf_->size() && dst.type_ == type_) { std::memcpy(dst.buf_->data(), buf_->data(), buf_->size()); dst.rows = rows; dst.cols = cols; }
        else dst = clone();
    }
    void setTo(int v) { if (buf_) std::memset(buf_->data(), v, buf_->size()); }     // (cv::Mat::setTo(Scalar): the tests only use 0)
    bool isContinuous() const { return true; }
    int type() const { return type_; }
    template <class T> T* ptr() { return buf_ ? reinterpret_cast<T*>(buf_->data()) : nullptr; }
    template <class T> const T* ptr() const { return buf_ ? reinterpret_cast<const T*>(buf_->data()) : nullptr; }
private:
    int type_ = CV_64FC1;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};
}  // namespace cv
