// Image decoding for the file-based entry points and the command-line tool (the reference uses
// cv::imread; OpenCV is not available to this build): binary PGM (P5, 8 or 16 bit) and
// non-interlaced PNG (8 or 16 bit; grey, grey+alpha, RGB, RGBA, 8-bit palette) via zlib.  Colour is
// reduced to grey with the fixed-point BT.601 weights (4899 R + 9617 G + 1868 B + 8192) >> 14;
// byte-identity with cv::imread on colour files is not claimed (its conversion depends on the codec
// build).
#pragma once
#include <stdint.h>

#include <functional>
#include <vector>

namespace mrg {

struct Image {
    int w = 0, h = 0, depth = 0;  // depth 8 or 16
    std::vector<uint8_t> px8;
    std::vector<uint16_t> px16;
    std::vector<uint8_t> file;  // the undecoded file; an Image that is reused keeps its buffers (and their warm pages)
    // called by read_image right before px8 / px16 has to GROW beyond its capacity (arguments: the element counts about to
    // be needed): an owner that has page-locked the old storage (hipHostRegister) must let go of it before it is freed
    std::function<void(size_t n8, size_t n16)> before_grow;
};

bool read_image(const char* path, Image& im);
// 16 -> 8 bit the way the reference CLI does it: convertTo(CV_8U, 255./65535.) (mrgingham-from-image.cc:91)
void to_8bit(const Image& im, std::vector<uint8_t>& out);
// 16 -> 8 bit the way cv::imread(IMREAD_GRAYSCALE) without IMREAD_ANYDEPTH does it (the reference's file
// entry points, find_chessboard_corners.cc:637-639, mrgingham.cc:158-160): the high byte
void to_8bit_imread(const Image& im, std::vector<uint8_t>& out);

// 8-bit grey PNG (what the reference's --debug dumps are written with cv::imwrite)
bool write_png_gray8(const char* path, const uint8_t* px, int w, int h);

}  // namespace mrg
