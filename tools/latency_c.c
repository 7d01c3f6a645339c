/* Single-image latency of the reference's symbols as a C caller sees it (no Python in the loop):
 *   gcc -O2 -std=c99 -Iinclude tools/latency_c.c -o /tmp/latency_c -Lmrgingham_amd -lmrgingham_amd -Wl,-rpath,$PWD/mrgingham_amd -Wl,-rpath,/opt/rocm/lib
 *   /tmp/latency_c image.pgm [iterations]
 * The image sits in page-locked memory (mrgingham_amd_host_alloc); results land in ordinary heap memory. */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mrgingham_amd.h"

static int g_n;
static bool count_ints(int* xy, int N, double scale, void* cookie) { (void)xy; (void)scale; (void)cookie; g_n = N; return true; }
static bool count_doubles(double* xy, int N, void* cookie) { (void)xy; (void)cookie; g_n = N; return true; }
static double now_ms(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

int main(int argc, char** argv) {
    int w = 0, h = 0, depth = 0, it, iters = argc > 2 ? atoi(argv[2]) : 200;
    uint8_t* img;
    int16_t* resp;
    double t0, t_resp, t_pts0, t_pts2, t_board;
    if (argc < 2 || mrgingham_amd_read_image(argv[1], 0, NULL, 0, &w, &h, &depth) != 0) return 2;
    img = (uint8_t*)mrgingham_amd_host_alloc((size_t)w * h);
    resp = (int16_t*)malloc((size_t)w * h * 2);
    if (!img || !resp || mrgingham_amd_read_image(argv[1], 0, img, (size_t)w * h, &w, &h, &depth) != 0) return 3;
    memset(resp, 0, (size_t)w * h * 2);
    for (it = 0; it < 10; ++it) {  /* warm-up: context, scratch, clocks */
        mrgingham_ChESS_response_5(resp, img, w, h, w);
        find_chessboard_corners_from_image_array_C(h, w, w, (char*)img, 0, false, false, count_ints, NULL);
        find_chessboard_from_image_array_C(h, w, w, (char*)img, 10, -1, false, false, -1, -1, count_doubles, NULL);
    }
    t0 = now_ms();
    for (it = 0; it < iters; ++it) mrgingham_ChESS_response_5(resp, img, w, h, w);
    t_resp = (now_ms() - t0) / iters;
    t0 = now_ms();
    for (it = 0; it < iters; ++it) find_chessboard_corners_from_image_array_C(h, w, w, (char*)img, 0, false, false, count_ints, NULL);
    t_pts0 = (now_ms() - t0) / iters;
    t0 = now_ms();
    for (it = 0; it < iters; ++it) find_chessboard_corners_from_image_array_C(h, w, w, (char*)img, 2, false, false, count_ints, NULL);
    t_pts2 = (now_ms() - t0) / iters;
    t0 = now_ms();
    for (it = 0; it < iters; ++it) find_chessboard_from_image_array_C(h, w, w, (char*)img, 10, -1, false, false, -1, -1, count_doubles, NULL);
    t_board = (now_ms() - t0) / iters;
    printf("%dx%d, C caller, image page-locked: mrgingham_ChESS_response_5 %.3f ms, find_chessboard_corners (level 0) %.3f ms, (level 2) %.3f ms, "
           "find_chessboard (level search + refinement) %.3f ms (%d corners)\n", w, h, t_resp, t_pts0, t_pts2, t_board, g_n);
    mrgingham_amd_host_free(img);
    free(resp);
    return 0;
}
