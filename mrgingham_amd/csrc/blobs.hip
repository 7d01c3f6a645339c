// Blob path, rows a-19 / (f)-4: mrgingham's find_blobs_from_image_array (find_blobs.cc:14-46) = a
// cv::SimpleBlobDetector (minArea 20, maxArea 80000, minDistBetweenBlobs 5, dark blobs, everything else at
// its default) whose keypoints become (int)(x * 1000 + 0.5) candidates for the grid finder.
//
// OpenCV arithmetic throughout (parity unpinned, like the other OpenCV steps): restated from its published
// algorithm -- threshold sweep 50..210 step 10, cv::findContours(RETR_LIST, CHAIN_APPROX_NONE) per
// threshold (Suzuki-Abe border following), per contour the moment / inertia / convexity / colour filters,
// median radius, grouping of the centres across thresholds (blobdetector.cpp, contours.cpp, moments.cpp).
//
// Split: the device does what touches every pixel -- the 17 binarised bit planes in one pass over the
// frame, the border start candidates of every plane, and the border following itself, one lane per
// candidate.  A border is followed once by design of the sequential algorithm (it marks what it
// followed); here every candidate start follows its border independently and gives up as soon as it
// meets a start of the same border with a smaller raster position, so exactly one lane -- the one
// the raster scan would have started from -- completes it, accumulating the integer sums of
// contourMoments on the way (Green's theorem: exact, order-independent).  Borders that pass the area
// filter (an integer comparison) are followed once more to write their points.  The host finishes the
// few survivors in double precision (inertia, convex hull, colour, median radius, grouping): microseconds,
// like the grid finder that consumes the result.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace mrg {

namespace {

constexpr int kNumThresh = 17;            // 50, 60, .. 210 (minThreshold 50, maxThreshold 220, step 10)
constexpr int kThresh0 = 50, kThreshStep = 10;
constexpr int kMaxBorder = 1 << 16;       // lanes give up on longer borders ...
constexpr int kMaxExtent = 2048;          // ... and on borders that stray this far from their start (see blob_follow)

struct BitPlanes {
    const uint32_t* bits;  // [kNumThresh][h][wpr]
    int w, h, wpr;
    __device__ __forceinline__ int at(int t, int x, int y) const {  // 0 outside the image: OpenCV pads with zeros
        if ((unsigned)x >= (unsigned)w || (unsigned)y >= (unsigned)h) return 0;
        return (bits[((long long)t * h + y) * wpr + (x >> 5)] >> (x & 31)) & 1u;
    }
};

// direction codes of the border follower: 0 = +x, then counter-clockwise on the screen (y down)
__device__ __constant__ int kDX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__device__ __constant__ int kDY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// bits (x-1, x, x+1) of row y of plane t, 0 outside the image
__device__ __forceinline__ uint32_t row3(const BitPlanes& bp, int t, int x, int y) {
    if ((unsigned)y >= (unsigned)bp.h) return 0u;
    const uint32_t* row = bp.bits + ((long long)t * bp.h + y) * bp.wpr;
    const int wx = x >> 5, b = x & 31;
    const uint32_t cur = row[wx];
    if (b == 0) return ((cur << 1) | (wx > 0 ? row[wx - 1] >> 31 : 0u)) & 7u;
    if (b == 31) return ((cur >> 30) | (wx + 1 < bp.wpr ? (row[wx + 1] & 1u) << 2 : 0u)) & 7u;
    return (cur >> (b - 1)) & 7u;
}

// The 8 neighbours of (x, y) as a mask, bit s = the pixel in direction s: three independent loads, after
// which every search of the follower is register work.
__device__ __forceinline__ uint32_t neighbours(const BitPlanes& bp, int t, int x, int y) {
    const uint32_t up = row3(bp, t, x, y - 1), mid = row3(bp, t, x, y), dn = row3(bp, t, x, y + 1);
    return ((mid >> 2) & 1u) | (((up >> 2) & 1u) << 1) | (((up >> 1) & 1u) << 2) | ((up & 1u) << 3) |
           ((mid & 1u) << 4) | ((dn & 1u) << 5) | (((dn >> 1) & 1u) << 6) | (((dn >> 2) & 1u) << 7);
}

// first set direction going clockwise (decreasing code) from `from` (exclusive, wrapping back to it); -1 if none
__device__ __forceinline__ int first_cw(uint32_t m, int from) {
    if (!m) return -1;
    // rotate so that direction from-1 becomes bit 7, from-2 bit 6, ...: bit (7 - k) <-> direction from-1-k
    const uint32_t r = ((m | (m << 8)) >> (from & 7)) & 0xffu;  // bit j = direction from + j; from-1-k = from + (7-k)
    return (from + (31 - __clz((int)r))) & 7;
}
// first set direction going counter-clockwise (increasing code) from s + 1
__device__ __forceinline__ int first_ccw(uint32_t m, int s) {
    const uint32_t r = ((m | (m << 8)) >> ((s + 1) & 7)) & 0xffu;  // bit j = direction s + 1 + j
    return (s + 1 + (__ffs((int)r) - 1)) & 7;
}

// one thread = 32 pixels of a row -> one word of each of the 17 planes
__global__ __launch_bounds__(256) void blob_bitplanes_kernel(const uint8_t* img, int stride, int w, int h, int wpr,
                                                             uint32_t* bits) {
    const int wx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (wx >= wpr) return;
    uint32_t word[kNumThresh];
#pragma unroll
    for (int t = 0; t < kNumThresh; ++t) word[t] = 0;
    const uint8_t* row = img + (long long)y * stride;
    for (int i = 0; i < 32; ++i) {
        const int x = wx * 32 + i;
        if (x >= w) break;
        const int v = row[x];
#pragma unroll
        for (int t = 0; t < kNumThresh; ++t) word[t] |= (uint32_t)(v > kThresh0 + kThreshStep * t) << i;  // THRESH_BINARY
    }
#pragma unroll
    for (int t = 0; t < kNumThresh; ++t) bits[((long long)t * h + y) * wpr + wx] = word[t];
}

// Border start candidates of plane t, the two conditions of the raster scan (contours.cpp): a white pixel
// whose left neighbour is 0 (outer-type start), a white pixel whose right neighbour is 0 and inside the
// image (hole-type start: the scan never looks at the zero pad).  key = (y << 16 | x << 1 | type): raster
// order, outer-type first at the same pixel.
__global__ __launch_bounds__(256) void blob_candidates_kernel(BitPlanes bp, uint32_t* cand_all, int cand_cap,
                                                              int* cand_cnt_all) {
    const int wx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = blockIdx.z;
    uint32_t* cand = cand_all + (long long)t * cand_cap;
    int* cand_cnt = cand_cnt_all + t;
    if (wx >= bp.wpr) return;
    const uint32_t* row = bp.bits + ((long long)t * bp.h + y) * bp.wpr;
    const uint32_t cur = row[wx];
    if (cur == 0) return;
    const uint32_t left_in = wx > 0 ? row[wx - 1] >> 31 : 0u;
    const uint32_t right_in = wx + 1 < bp.wpr ? row[wx + 1] & 1u : 0u;
    uint32_t outer = cur & ~((cur << 1) | left_in);
    uint32_t hole = cur & ~((cur >> 1) | (right_in << 31));
    while (outer) {
        const int i = __ffs(outer) - 1;
        outer &= outer - 1;
        const int k = atomicAdd(cand_cnt, 1);
        if (k < cand_cap) cand[k] = ((uint32_t)y << 16) | ((uint32_t)(wx * 32 + i) << 1);
    }
    while (hole) {
        const int i = __ffs(hole) - 1;
        hole &= hole - 1;
        const int x = wx * 32 + i;
        if (x >= bp.w - 1) continue;  // the pixel to the right is the pad
        const int k = atomicAdd(cand_cnt, 1);
        if (k < cand_cap) cand[k] = ((uint32_t)y << 16) | ((uint32_t)x << 1) | 1u;
    }
}

struct BlobContour {      // one followed border that passed the area filter
    uint32_t key;         // its start (raster position and type): the order of discovery
    int32_t t;            // threshold index
    int32_t n;            // points
    uint32_t points_off;  // first point in the arena
    long long a00, a10, a01, a20, a11, a02;  // contourMoments sums
};

// Follows the border that starts at candidate `key` exactly like icvFetchContour (CHAIN_APPROX_NONE): from the
// start, the predecessor is the first white neighbour clockwise from west (outer start) or east (hole start);
// then, at every point, the next one is the first white neighbour counter-clockwise from the direction we
// came from, until the start is reached again from that predecessor.
// WRITE = false: returns false as soon as another start of the same border with a smaller key is met (that
// lane owns the border) or the border is longer than kMaxBorder; otherwise fills the sums and the length.
// WRITE = true: stores the points ((y << 16) | x) to `pts`.
// kMaxBorder / kMaxExtent: a border that passes the filters encloses less than 80000 px^2, fills >= 95 % of its
// convex hull (hull area H < 84211) and has an inertia ratio >= 0.1.  A convex region of area H that is D
// pixels long is at most 2H/D wide, so its inertia ratio is of the order (2H/D^2)^2: at D = 2048 that is
// below 0.002 -- such a border cannot pass.  And what a passing border can spend on detours is bounded by
// the 5 % of hull area it may waste (each step of a detour wastes about half a pixel): about 10^4 steps on
// top of a perimeter of at most a few thousand, far below 2^16.  Longer or wider borders (the image
// frame, background ridges in noise) can only be rejected, and following them to the end in one lane
// would take most of the call.
template <bool WRITE>
__device__ __forceinline__ bool blob_follow(const BitPlanes& bp, int t, uint32_t key, BlobContour& c, uint32_t* pts) {
    const int x0 = (int)((key >> 1) & 0x7fffu), y0 = (int)(key >> 16), is_hole = (int)(key & 1u);
    uint32_t m = neighbours(bp, t, x0, y0);
    int s = first_cw(m, is_hole ? 0 : 4);
    if (s < 0) return false;  // a single pixel: area 0, never a blob
    const int x1 = x0 + kDX[s], y1 = y0 + kDY[s];  // the predecessor of the start on the border
    int x3 = x0, y3 = y0, n = 0;
    unsigned long long a00 = 0, a10 = 0, a01 = 0, a20 = 0, a11 = 0, a02 = 0;  // wrap-around integers: exact results
    for (;;) {
        s = first_ccw(m, s);  // the next border point
        const int x4 = x3 + kDX[s], y4 = y3 + kDY[s];
        if (WRITE) {
            pts[n] = ((uint32_t)y3 << 16) | (uint32_t)x3;
        } else {
            // contourMoments term of the pair (this point -> next point)
            const long long xa = x3, ya = y3, xb = x4, yb = y4;
            const long long dxy = xa * yb - xb * ya, xii = xa + xb, yii = ya + yb;
            a00 += (unsigned long long)dxy;
            a10 += (unsigned long long)(dxy * xii);
            a01 += (unsigned long long)(dxy * yii);
            a20 += (unsigned long long)(dxy * (xa * xii + xb * xb));
            a11 += (unsigned long long)(dxy * (xa * (yii + ya) + xb * (yii + yb)));
            a02 += (unsigned long long)(dxy * (ya * yii + yb * yb));
        }
        ++n;
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;  // closed
        if (n >= kMaxBorder || abs(x4 - x0) > kMaxExtent || abs(y4 - y0) > kMaxExtent) return false;
        // now at (x4, y4), having come from direction s + 4
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
        m = neighbours(bp, t, x3, y3);
        if (!WRITE) {
            // is this pixel a start of the same border that the raster scan meets earlier?
            const uint32_t here = ((uint32_t)y3 << 16) | ((uint32_t)x3 << 1), mine = key & ~1u;
            if ((here < mine || (here == mine && is_hole)) && !(m & 0x10u) && first_cw(m, 4) == s) return false;
            if (here < mine && x3 < bp.w - 1 && !(m & 1u) && first_cw(m, 0) == s) return false;
        }
    }
    c.n = n;
    c.a00 = (long long)a00; c.a10 = (long long)a10; c.a01 = (long long)a01;
    c.a20 = (long long)a20; c.a11 = (long long)a11; c.a02 = (long long)a02;
    return true;
}

__global__ __launch_bounds__(256) void blob_trace_kernel(BitPlanes bp, const uint32_t* cand_all, const int* cand_cnt_all,
                                                         int cand_cap, BlobContour* recs, int rec_cap, int* counters) {
    const int t = blockIdx.y;
    const uint32_t* cand = cand_all + (long long)t * cand_cap;
    const int ncand = min(cand_cnt_all[t], cand_cap);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ncand; i += gridDim.x * 256) {
        BlobContour c;
        c.key = cand[i];
        c.t = t;
        if (!blob_follow<false>(bp, t, c.key, c, nullptr)) continue;
        // filterByArea on m00 = |a00| / 2: minArea 20 <= m00 < maxArea 80000, exact in integers
        const long long a = c.a00 < 0 ? -c.a00 : c.a00;
        if (a < 2 * 20 || a >= 2 * 80000) continue;
        const int r = atomicAdd(counters + 0, 1);
        const unsigned off = (unsigned)atomicAdd(counters + 1, c.n);
        c.points_off = off;
        if (r < rec_cap) recs[r] = c;
    }
}

__global__ __launch_bounds__(256) void blob_points_kernel(BitPlanes bp, const BlobContour* recs, int rec_cap,
                                                          const int* counters, uint32_t* pts, int pts_cap) {
    const int nrec = min(counters[0], rec_cap);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nrec; i += gridDim.x * 256) {
        BlobContour c = recs[i];
        if ((long long)c.points_off + c.n > pts_cap) continue;
        blob_follow<true>(bp, c.t, c.key, c, pts + c.points_off);
    }
}

// ---------------------------------------------------------------------------------------------
// host: the double-precision part of SimpleBlobDetector::findBlobs and the grouping of detect()
// ---------------------------------------------------------------------------------------------
struct Center { double x, y, radius, confidence; };

struct IPt { int x, y; };

double polygon_area(const IPt* p, int n) {  // cv::contourArea, not oriented
    if (n == 0) return 0.;
    double a00 = 0, xp = p[n - 1].x, yp = p[n - 1].y;
    for (int i = 0; i < n; ++i) {
        const double x = p[i].x, y = p[i].y;
        a00 += xp * y - x * yp;
        xp = x;
        yp = y;
    }
    return std::fabs(a00 * 0.5);
}

double hull_area(std::vector<IPt> s) {  // area of the convex hull of the point set (monotone chain)
    const int n = (int)s.size();
    std::sort(s.begin(), s.end(), [](const IPt& a, const IPt& b) { return a.x != b.x ? a.x < b.x : a.y < b.y; });
    auto cross = [](const IPt& o, const IPt& a, const IPt& b) {
        return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
    };
    std::vector<IPt> hull((size_t)2 * n + 2);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        while (k >= 2 && cross(hull[k - 2], hull[k - 1], s[i]) <= 0) --k;
        hull[k++] = s[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
        while (k >= t && cross(hull[k - 2], hull[k - 1], s[i]) <= 0) --k;
        hull[k++] = s[i];
    }
    return polygon_area(hull.data(), k > 1 ? k - 1 : k);
}

// blobdetector.cpp findBlobs for one contour that passed the area filter
bool contour_to_center(const BlobContour& c, const uint32_t* pts, const uint8_t* img, int w, int h, int stride,
                       Center* out) {
    // contourMoments: scale the sums, then completeMomentState
    const double a00 = (double)c.a00, a10 = (double)c.a10, a01 = (double)c.a01;
    const double a20 = (double)c.a20, a11 = (double)c.a11, a02 = (double)c.a02;
    if (!(std::fabs(a00) > FLT_EPSILON)) return false;
    const double sg = a00 > 0 ? 1.0 : -1.0;
    const double m00 = a00 * (sg * 0.5), m10 = a10 * (sg * 0.16666666666666666666666666666667);
    const double m01 = a01 * (sg * 0.16666666666666666666666666666667);
    const double m20 = a20 * (sg * 0.083333333333333333333333333333333);
    const double m11 = a11 * (sg * 0.041666666666666666666666666666667);
    const double m02 = a02 * (sg * 0.083333333333333333333333333333333);
    double cx = 0, cy = 0;
    if (std::fabs(m00) > DBL_EPSILON) {
        const double inv = 1. / m00;
        cx = m10 * inv;
        cy = m01 * inv;
    }
    const double mu20 = m20 - m10 * cx, mu11 = m11 - m10 * cy, mu02 = m02 - m01 * cy;
    if (m00 < 20. || m00 >= 80000.) return false;  // filterByArea (the device already applied it)
    double ratio;                                   // filterByInertia, minInertiaRatio 0.1
    const double denominator = std::sqrt((2 * mu11) * (2 * mu11) + (mu20 - mu02) * (mu20 - mu02));
    if (denominator > 1e-2) {
        const double cosmin = (mu20 - mu02) / denominator, sinmin = 2 * mu11 / denominator;
        const double cosmax = -cosmin, sinmax = -sinmin;
        const double imin = 0.5 * (mu20 + mu02) - 0.5 * (mu20 - mu02) * cosmin - mu11 * sinmin;
        const double imax = 0.5 * (mu20 + mu02) - 0.5 * (mu20 - mu02) * cosmax - mu11 * sinmax;
        ratio = imin / imax;
    } else {
        ratio = 1;
    }
    if (ratio < (double)0.1f || ratio >= FLT_MAX) return false;  // Params::minInertiaRatio is a float: 0.100000001490...
    std::vector<IPt> p((size_t)c.n);
    for (int i = 0; i < c.n; ++i) p[i] = IPt{(int)(pts[i] & 0xffffu), (int)(pts[i] >> 16)};
    {  // filterByConvexity, minConvexity 0.95
        const double carea = polygon_area(p.data(), c.n), harea = hull_area(p);
        if (std::fabs(harea) < DBL_EPSILON) return false;
        const double conv = carea / harea;
        if (conv < (double)0.95f || conv >= FLT_MAX) return false;  // Params::minConvexity is a float: 0.949999988079...
    }
    if (m00 == 0.0) return false;
    out->x = m10 / m00;
    out->y = m01 / m00;
    out->confidence = ratio * ratio;
    {  // filterByColor, blobColor 0: the binarised pixel at the rounded centre must be dark
        const int ix = (int)std::nearbyint(out->x), iy = (int)std::nearbyint(out->y);  // cvRound
        if (ix < 0 || ix >= w || iy < 0 || iy >= h) return false;
        if (img[(size_t)iy * stride + ix] > kThresh0 + kThreshStep * c.t) return false;
    }
    std::vector<double> d((size_t)c.n);
    for (int i = 0; i < c.n; ++i) {
        const double dx = out->x - p[i].x, dy = out->y - p[i].y;
        d[i] = std::sqrt(dx * dx + dy * dy);
    }
    std::sort(d.begin(), d.end());
    out->radius = (d[(size_t)(c.n - 1) / 2] + d[(size_t)c.n / 2]) / 2.;
    return true;
}

}  // namespace

size_t blob_scratch_bytes(int w, int h, BlobScratchLayout* lay) {
    BlobScratchLayout L;
    L.wpr = (w + 31) / 32;
    const long long px = (long long)w * h;
    L.cand_cap = (int)std::max<long long>(1 << 16, px / 4);
    L.rec_cap = (int)std::max<long long>(4096, px / 64);
    L.pts_cap = (int)std::max<long long>(1 << 18, px);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.o_counters = take(64 * sizeof(int));
    L.o_bits = take((size_t)kNumThresh * h * L.wpr * 4);
    L.o_cand = take((size_t)kNumThresh * L.cand_cap * 4);
    L.o_recs = take((size_t)L.rec_cap * sizeof(BlobContour));
    L.o_pts = take((size_t)L.pts_cap * 4);
    if (lay) *lay = L;
    return off;
}

// d_img: the frame on the device; h_img: the same pixels on the host (colour filter).  Appends the keypoints
// as (x, y) * 1000 ints in SimpleBlobDetector's output order.  false on a device error or when the frame
// has more borders than the scratch holds (err says which).
bool blob_detect(const uint8_t* d_img, int d_stride, const uint8_t* h_img, int h_stride, int w, int h, void* scratch,
                 hipStream_t s, std::vector<int32_t>& xy_out, std::string& err) {
    if (w <= 0 || h <= 0) return true;
    BlobScratchLayout L;
    blob_scratch_bytes(w, h, &L);
    char* base = (char*)scratch;
    int* counters = (int*)(base + L.o_counters);  // [0] records, [1] points, [2 + t] candidates of plane t
    uint32_t* bits = (uint32_t*)(base + L.o_bits);
    uint32_t* cand = (uint32_t*)(base + L.o_cand);
    BlobContour* recs = (BlobContour*)(base + L.o_recs);
    uint32_t* pts = (uint32_t*)(base + L.o_pts);
    const BitPlanes bp{bits, w, h, L.wpr};
    hipMemsetAsync(counters, 0, 64 * sizeof(int), s);
    const dim3 grid_rows((L.wpr + 255) / 256, h);
    hipLaunchKernelGGL(blob_bitplanes_kernel, grid_rows, dim3(256), 0, s, d_img, d_stride, w, h, L.wpr, bits);
    // every plane at once: the planes are independent, and a launch lasts as long as its longest border
    hipLaunchKernelGGL(blob_candidates_kernel, dim3(grid_rows.x, grid_rows.y, kNumThresh), dim3(256), 0, s, bp, cand,
                       L.cand_cap, counters + 2);
    hipLaunchKernelGGL(blob_trace_kernel, dim3(256, kNumThresh), dim3(256), 0, s, bp, (const uint32_t*)cand,
                       (const int*)(counters + 2), L.cand_cap, recs, L.rec_cap, counters);
    hipLaunchKernelGGL(blob_points_kernel, dim3(256), dim3(256), 0, s, bp, (const BlobContour*)recs, L.rec_cap,
                       (const int*)counters, pts, L.pts_cap);
    int hc[64];
    if (hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        err = "blob detector: device error";
        return false;
    }
    for (int t = 0; t < kNumThresh; ++t)
        if (hc[2 + t] > L.cand_cap) { err = "blob detector: more border starts than the scratch holds (pure noise?)"; return false; }
    if (hc[0] > L.rec_cap || hc[1] > L.pts_cap) { err = "blob detector: more contours than the scratch holds"; return false; }
    std::vector<BlobContour> hrec((size_t)hc[0]);
    std::vector<uint32_t> hpts((size_t)hc[1]);
    if ((hc[0] && hipMemcpy(hrec.data(), recs, hrec.size() * sizeof(BlobContour), hipMemcpyDeviceToHost) != hipSuccess) ||
        (hc[1] && hipMemcpy(hpts.data(), pts, hpts.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)) {
        err = "blob detector: download failed";
        return false;
    }
    // contours of a threshold in cv::findContours' RETR_LIST order: discovery order (raster order of the
    // starts), reversed
    std::sort(hrec.begin(), hrec.end(), [](const BlobContour& a, const BlobContour& b) {
        return a.t != b.t ? a.t < b.t : a.key > b.key;
    });
    std::vector<std::vector<Center>> groups;  // blobdetector.cpp detect(): centres of one blob across thresholds
    size_t i = 0;
    for (int t = 0; t < kNumThresh; ++t) {
        std::vector<Center> cur;
        for (; i < hrec.size() && hrec[i].t == t; ++i) {
            Center c;
            if (contour_to_center(hrec[i], hpts.data() + hrec[i].points_off, h_img, w, h, h_stride, &c)) cur.push_back(c);
        }
        std::vector<std::vector<Center>> fresh;
        for (const Center& c : cur) {
            bool is_new = true;
            for (auto& g : groups) {
                const Center& mid = g[g.size() / 2];
                const double dx = mid.x - c.x, dy = mid.y - c.y;
                const double dist = std::sqrt(dx * dx + dy * dy);
                is_new = dist >= 5. && dist >= mid.radius && dist >= c.radius;  // minDistBetweenBlobs 5
                if (!is_new) {
                    g.push_back(c);
                    size_t k = g.size() - 1;
                    while (k > 0 && c.radius < g[k - 1].radius) { g[k] = g[k - 1]; --k; }
                    g[k] = c;
                    break;
                }
            }
            if (is_new) fresh.push_back(std::vector<Center>(1, c));
        }
        for (auto& g : fresh) groups.push_back(std::move(g));
    }
    for (const auto& g : groups) {
        if (g.size() < 2) continue;  // minRepeatability 2
        double sx = 0, sy = 0, norm = 0;
        for (const Center& c : g) {
            sx += c.confidence * c.x;
            sy += c.confidence * c.y;
            norm += c.confidence;
        }
        const double inv = 1. / norm;
        sx *= inv;
        sy *= inv;
        const float fx = (float)sx, fy = (float)sy;        // KeyPoint::pt is a Point2f
        const float tx = fx * 1000.0f, ty = fy * 1000.0f;  // it->pt.x * FIND_GRID_SCALE, find_blobs.cc:40-41
        xy_out.push_back((int)((double)tx + 0.5));
        xy_out.push_back((int)((double)ty + 0.5));
    }
    return true;
}

}  // namespace mrg
