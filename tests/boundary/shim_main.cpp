// Boundary test program (tests/test_cvmat_shim.py): include/find_chessboard_corners_amd.hh compiled against the
// structural cv::Mat stub, its signatures checked against what find_chessboard_corners.hh:12-30, :32-44, :51-72
// declare, and -- with a file name -- one detection + refinement through it, printed for the test to compare.
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "find_chessboard_corners_amd.hh"

using namespace mrgingham;
static_assert(std::is_same<decltype(&find_chessboard_corners_from_image_array),
                           bool (*)(std::vector<PointInt>*, const cv::Mat&, int, bool, const char*)>::value,
              "find_chessboard_corners.hh:12-30");
static_assert(std::is_same<decltype(&find_chessboard_corners_from_image_file),
                           bool (*)(std::vector<PointInt>*, const char*, int, bool)>::value,
              "find_chessboard_corners.hh:32-44");
static_assert(std::is_same<decltype(&refine_chessboard_corners_from_image_array),
                           int (*)(std::vector<PointDouble>*, signed char*, const cv::Mat&, int, bool, const char*)>::value,
              "find_chessboard_corners.hh:51-72");

int main(int argc, char** argv) {
    if (argc < 5) return 0;  // compiled and linked: that is the CPU half of the test
    // usage: shim_main raw.bin W H level  (raw.bin: W*H bytes)
    const int W = atoi(argv[2]), H = atoi(argv[3]), level = atoi(argv[4]);
    std::vector<unsigned char> px((size_t)W * H);
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(px.data(), 1, px.size(), f) != px.size()) return 2;
    fclose(f);
    cv::Mat m(H, W, CV_8U, px.data(), (size_t)W);
    std::vector<PointInt> pts;
    const bool found = find_chessboard_corners_from_image_array(&pts, m, level);        // defaults: debug = false, NULL
    printf("found %d n %zu\n", (int)found, pts.size());
    for (const PointInt& p : pts) printf("p %d %d\n", p.x, p.y);
    std::vector<PointDouble> dp;
    for (const PointInt& p : pts) dp.push_back(PointDouble((double)p.x / 1000., (double)p.y / 1000.));
    std::vector<signed char> lv(dp.size(), (signed char)level);
    int nref = 0;
    if (level > 0) nref = refine_chessboard_corners_from_image_array(&dp, lv.data(), m, level - 1);
    printf("refined %d\n", nref);
    for (size_t i = 0; i < dp.size(); ++i) printf("r %.17g %.17g %d\n", dp[i].x, dp[i].y, (int)lv[i]);
    cv::Mat bad(H, W, CV_16S, px.data(), (size_t)W * 2);
    std::vector<PointInt> none;
    printf("bad_type %d\n", (int)find_chessboard_corners_from_image_array(&none, bad, 0));
    return 0;
}
