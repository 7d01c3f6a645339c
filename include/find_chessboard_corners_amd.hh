// find_chessboard_corners_amd.hh -- the reference's C++ entry points of the corner-candidate path as inline forwards
// to libmrgingham_amd.so: include this INSTEAD of compiling find_chessboard_corners.cc (it defines what
// find_chessboard_corners.hh:12-30, :32-44 and :51-72 declare; same names, arguments, defaults and return values).
// cv::Mat cannot cross a C-ABI, so this header is compiled where OpenCV's headers are -- the reference's own build --
// and hands (rows, cols, step, data) to the C symbols of include/mrgingham_amd.h.  With it mrgingham.cc:50 and
// :89-94 (the level loop and the refine loop around the host grid finder) build and run unmodified.
#pragma once
#include <opencv2/core/core.hpp>

#include <vector>

#include "mrgingham_amd.h"
#include "point.hh"

namespace mrgingham {

// find_chessboard_corners.hh:12-30, find_chessboard_corners.cc:568-587: appends the candidates (x, y) * FIND_GRID_SCALE to
// *points_scaled_out; true when the vector is not empty afterwards (:354, :583-586)
inline bool find_chessboard_corners_from_image_array(std::vector<mrgingham::PointInt>* points_scaled_out,
                                                     const cv::Mat& image_input, int image_pyramid_level, bool debug = false,
                                                     const char* debug_image_filename = NULL) {
    (void)debug_image_filename;  // (names the background image of the reference's debug plot only)
    if (image_input.type() == CV_8U) {  // anything else: "I can only handle CV_8U", no points (find_chessboard_corners.cc:468-473)
        bool (*add)(int*, int, double, void*) = [](int* xy, int N, double, void* cookie) -> bool {
            std::vector<mrgingham::PointInt>* v = static_cast<std::vector<mrgingham::PointInt>*>(cookie);
            for (int i = 0; i < N; i++) v->push_back(mrgingham::PointInt(xy[2 * i], xy[2 * i + 1]));
            return true;
        };
        find_chessboard_corners_from_image_array_C(image_input.rows, image_input.cols, (int)image_input.step,
                                                   (char*)image_input.data, image_pyramid_level, false, debug, add,
                                                   points_scaled_out);
    }
    return points_scaled_out->size() > 0;
}

// find_chessboard_corners.hh:32-44, find_chessboard_corners.cc:623-648 (the library decodes binary PGM and PNG itself)
inline bool find_chessboard_corners_from_image_file(std::vector<mrgingham::PointInt>* points, const char* filename,
                                                    int image_pyramid_level, bool debug = false) {
    bool (*add)(int*, int, double, void*) = [](int* xy, int N, double, void* cookie) -> bool {
        std::vector<mrgingham::PointInt>* v = static_cast<std::vector<mrgingham::PointInt>*>(cookie);
        for (int i = 0; i < N; i++) v->push_back(mrgingham::PointInt(xy[2 * i], xy[2 * i + 1]));
        return true;
    };
    find_chessboard_corners_from_image_file_C(filename, image_pyramid_level, debug, add, points);
    return points->size() > 0;
}

// find_chessboard_corners.hh:51-72, find_chessboard_corners.cc:591-619: refines in place the points whose level is
// image_pyramid_level + 1; returns how many were refined
inline int refine_chessboard_corners_from_image_array(std::vector<mrgingham::PointDouble>* points, signed char* level,
                                                      const cv::Mat& image_input, int image_pyramid_level,
                                                      bool debug = false, const char* debug_image_filename = NULL) {
    (void)debug_image_filename;
    static_assert(sizeof(mrgingham::PointDouble) == 2 * sizeof(double), "PointDouble must be two doubles (bridge.cc:133-134)");
    if (image_input.type() != CV_8U || points->empty()) return 0;
    return refine_chessboard_corners_from_image_array_C(image_input.rows, image_input.cols, (int)image_input.step,
                                                        (char*)image_input.data, &(*points)[0].x, level, (int)points->size(),
                                                        image_pyramid_level, debug);
}

}  // namespace mrgingham
