// oracle/ref_shim/opencv2/highgui/highgui.hpp -- TEST INFRASTRUCTURE.  OpenCV is not installed here.  The reference code compiled into
// oracle/_ref/libref.so names cv::Mat in declarations (FullSystem.h, IOWrapper/ImageRW.h) and, in FullSystem.cpp's lidar-mask helper
// (setMask / makeNewTraces, not run by the pinned checks), uses an 8-bit single-channel image through zeros / at<uchar> / release.
#pragma once
#include <memory>
#include <vector>
typedef unsigned char uchar;
#define CV_8UC1 0
namespace cv {
class Mat {
    std::shared_ptr<std::vector<unsigned char> > d_;
public:
    int rows = 0, cols = 0;
    Mat() {}
    Mat(int r, int c, int) : d_(new std::vector<unsigned char>((size_t)r * c, 0)), rows(r), cols(c) {}
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
    template <typename T> T& at(int r, int c) { return reinterpret_cast<T*>(d_->data())[(size_t)r * cols + c]; }
    void release() { d_.reset(); rows = cols = 0; }
    bool empty() const { return !d_; }
    Mat clone() const { Mat m; m.rows = rows; m.cols = cols; if (d_) m.d_.reset(new std::vector<unsigned char>(*d_)); return m; }
};
}
