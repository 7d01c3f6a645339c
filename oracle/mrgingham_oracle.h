/* TEST INFRASTRUCTURE ONLY -- see the header comment of mrgingham_oracle.c.
 * CPU restatement of the reference hot path; never linked into the product. */
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ChESS.c:56-106.  Writes the interior [7,w-7)x[7,h-7) only. */
void oracle_chess_response_5(int16_t* response, const uint8_t* image, int w, int h, int stride);

/* Output size of cv::resize(..., 1/2^level) (find_chessboard_corners.cc:449-450).
 * Returns -1 for level outside [0,10] (:433-441). */
int oracle_level_dims(int W, int H, int level, int* w, int* h);

/* Level image: dense w x h bytes (level 0 = a dense copy). */
int oracle_decimate(uint8_t* out, const uint8_t* in, int W, int H, int stride, int level);

/* cv::blur with a (2r+1)^2 box, BORDER_REFLECT_101 (mrgingham-from-image.cc:106-111). */
void oracle_box_blur(uint8_t* out, const uint8_t* in, int w, int h, int stride, int radius);

/* The CLI's contrast preprocessing (mrgingham-from-image.cc:38-45, :71-111); OpenCV arithmetic,
 * parity unpinned (see the .c file). */
void oracle_normalize_lut(uint8_t lut[256], int vmin, int vmax);
void oracle_normalize_minmax(uint8_t* out, const uint8_t* in, int w, int h, int stride);
int oracle_clahe(uint8_t* out, const uint8_t* in, int w, int h, int stride, double clip_limit);
int oracle_preprocess(uint8_t* out, const uint8_t* in, int w, int h, int stride, int do_clahe, int blur_radius);

/* the 16-bit branch (mrgingham-from-image.cc:85-92); stride in elements */
void oracle_normalize16(uint16_t* out, const uint16_t* in, int w, int h, int stride);
int oracle_clahe16(uint16_t* out, const uint16_t* in, int w, int h, int stride, double clip_limit);
int oracle_preprocess16(uint8_t* out, const uint16_t* in, int w, int h, int stride, int do_clahe, int blur_radius);

/* Clamped (negatives -> 0, border zero) response and the level image the
 * connected-component stage sees (find_chessboard_corners.cc:495-529). */
int oracle_clamped_response(int16_t* resp_out, uint8_t* level_image_out, const uint8_t* image, int H, int W,
                            int stride, int level);

/* find_chessboard_corners_from_image_array (find_chessboard_corners.cc:568-587).
 * xy_out receives min(N,cap) interleaved (x,y)*1000 ints; returns N, or -1 on
 * the errors the reference reports with a message and "no points". */
int oracle_find_corners(int32_t* xy_out, int cap, const uint8_t* image, int H, int W, int stride, int level);

/* refine_chessboard_corners_from_image_array (find_chessboard_corners.cc:591-619).
 * xy: npoints interleaved doubles, updated in place; returns the number refined. */
int oracle_refine_corners(double* xy, signed char* level_of_point, int npoints, const uint8_t* image, int H,
                          int W, int stride, int level);

/* The connected-component stage alone, on a caller-built clamped response
 * (mutated) and level image. */
int oracle_cc_detect_on_response(int32_t* xy_out, int cap, int16_t* d, const uint8_t* level_image, int w,
                                 int h, int level);
int oracle_cc_refine_on_response(double* xy, signed char* level_of_point, int npoints, int16_t* d,
                                 const uint8_t* level_image, int w, int h, int level);

/* find_blobs_from_image_array (find_blobs.cc:14-46): cv::SimpleBlobDetector keypoints as (x, y) * 1000 ints
 * (blobs_oracle.c; OpenCV arithmetic, parity unpinned).  Returns the number of blobs; the first
 * min(N, cap) are stored. */
int oracle_find_blobs(int32_t* xy_out, int cap, const uint8_t* image, int w, int h, int stride);

#ifdef __cplusplus
}
#endif
