// BOUNDARY-TEST STUB, not OpenCV and not an oracle: the few members of cv::Mat that
// include/find_chessboard_corners_amd.hh touches, so that the shim can be COMPILED in an image without OpenCV
// (tests/test_cvmat_shim.py).  It pins no arithmetic: nothing of OpenCV's is computed through it.
#pragma once
#include <cstddef>
#define CV_8U 0
#define CV_16S 3
namespace cv {
struct Mat {
    int rows, cols;
    unsigned char* data;
    size_t step;
    int type_;
    Mat(int r, int c, int t, void* d, size_t s) : rows(r), cols(c), data((unsigned char*)d), step(s), type_(t) {}
    int type() const { return type_; }
};
}  // namespace cv
