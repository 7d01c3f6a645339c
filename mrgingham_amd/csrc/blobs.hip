// Blob path, rows a-19 / (f)-4: mrgingham's find_blobs_from_image_array (find_blobs.cc:14-46) = a
// cv::SimpleBlobDetector (minArea 20, maxArea 80000, minDistBetweenBlobs 5, dark blobs, everything else at
// its default) whose keypoints become (int)(x * 1000 + 0.5) candidates for the grid finder.
//
// OpenCV arithmetic throughout (parity unpinned, like the other OpenCV steps): restated from its published
// algorithm -- threshold sweep 50..210 step 10, cv::findContours(RETR_LIST, CHAIN_APPROX_NONE) per
// threshold (Suzuki-Abe border following), per contour the moment / inertia / convexity / colour filters,
// median radius, grouping of the centres across thresholds (blobdetector.cpp, contours.cpp, moments.cpp).
//
// Split: the device does what touches every pixel -- the 17 binarised bit planes in one pass over the frame, the
// border start candidates of every plane and the border following itself (in parallel: arcs between candidates,
// then pointer jumping over the cycles of arcs, see below), with the contour-area sum of contourMoments (Green's
// theorem: exact in integers, order-independent) for the area filter.  Borders that pass it get their points
// written, every arc its own stretch.  The host finishes the few survivors in double precision (the other
// moments, inertia, convex hull, colour, median radius, grouping): microseconds, like the grid finder that
// consumes the result.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include "kernels.h"

namespace mrg {

namespace {

constexpr int kNumThresh = 17;            // 50, 60, .. 210 (minThreshold 50, maxThreshold 220, step 10)
constexpr int kThresh0 = 50, kThreshStep = 10;

struct BitPlanes {
    const uint32_t* bits;  // [kNumThresh][h][wpr]
    int w, h, wpr;
};

// direction codes of the border follower: 0 = +x, then counter-clockwise on the screen (y down)
__device__ __constant__ int kDX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__device__ __constant__ int kDY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

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

// ---------------------------------------------------------------------------------------------
// Border following, in parallel.  The sequential algorithm (icvFetchContour, CHAIN_APPROX_NONE) walks a border
// pixel by pixel; the state of the walk is (pixel, direction of the pixel it came from) and the next state is a
// function of the state and the pixel's 8 neighbours.  The raster scan starts a walk at a white pixel whose left
// neighbour is 0 (outer-type start) or whose right neighbour is 0 (hole-type start), in the state the start rule
// gives it -- and EVERY left-edge / right-edge pixel of a border is such a start candidate whose start state lies
// on that border's cycle of states.  So the candidates cut every border into ARCS: a lane walks from its
// candidate only to the next candidate state it meets (whatever its key), a handful of steps except along
// horizontal runs, and leaves (next candidate, steps, the arc's share of the contour area sum).  Then the cycles
// of `next` are reduced with pointer jumping: the smallest candidate of a cycle is the start the raster scan
// would have used (the owner; its key orders the contours), a list ranking from it gives every arc its offset in
// the contour's point list and the owner the totals.  Round 2 walked every border end to end in one lane from
// every candidate (20-65 ms per 12 MP frame, a launch lasting as long as its longest border, and give-up rules
// for borders of more than 65 536 steps); this is O(log) rounds over short arcs and needs no give-up rule.
//
// Candidates are numbered in raster order per plane (row offsets + per-word prefix counts + popcounts, no atomics,
// no sort), planes one after the other: node = candidate = arc.  Two candidates at one pixel with the same start
// state (a one-pixel-wide stroke: left AND right neighbour 0) are the same state; the outer-type one is the node,
// the hole-type one is dead (contours.cpp: the outer check comes first and marks the pixel).
// ---------------------------------------------------------------------------------------------
constexpr uint32_t kNoNode = 0xffffffffu;
constexpr int kMaxArc = 1 << 20;  // steps; an arc is bounded by the longest horizontal run of a border (the frame's width)

// candidate masks of one word of a row: bit i = pixel 32 * wx + i starts an outer-type / a hole-type border walk
__device__ __forceinline__ void word_candidates(uint32_t cur, uint32_t left_in, uint32_t right_in, int wx, int w,
                                                uint32_t& outer, uint32_t& hole) {
    outer = cur & ~((cur << 1) | left_in);
    hole = cur & ~((cur >> 1) | (right_in << 31));
    (void)wx;
    (void)w;
}
// The raster scan never looks at the zero pad, so a white pixel of the LAST COLUMN is no hole-type start.  It is kept
// as a node all the same -- it cuts the walk down the frame's right edge into arcs like every other candidate (without
// it that is one arc of `height` vertical steps, a row load each: 2.5 ms at 3072 rows, the whole launch) -- but it can
// never own a border: it does not take part in the election of the smallest node.
__device__ __forceinline__ bool pseudo_node(uint32_t key, int w) { return (key & 1u) && (int)((key >> 1) & 0x7fffu) == w - 1; }

// rows y-1, y, y+1 around word wx of plane t, each as 34 bits: bit 0 = the previous word's bit 31, bits 1..32 = the
// word, bit 33 = the next word's bit 0.  A walk reloads a row only when it leaves the word or changes row.
struct BlobWin {
    int wx, y;
    unsigned long long r[3];
};
__device__ __forceinline__ unsigned long long row34(const BitPlanes& bp, int t, int wx, int y) {
    if ((unsigned)y >= (unsigned)bp.h) return 0ull;
    const uint32_t* row = bp.bits + ((long long)t * bp.h + y) * bp.wpr;
    const uint32_t cur = row[wx];
    const uint32_t prev = wx > 0 ? row[wx - 1] : 0u, next = wx + 1 < bp.wpr ? row[wx + 1] : 0u;
    return (unsigned long long)(prev >> 31) | ((unsigned long long)cur << 1) | ((unsigned long long)(next & 1u) << 33);
}
__device__ __forceinline__ void win_goto(BlobWin& wn, const BitPlanes& bp, int t, int x, int y) {
    const int wx = x >> 5;
    if (wx == wn.wx && y == wn.y) return;
    if (wx == wn.wx && y == wn.y + 1) {
        wn.r[0] = wn.r[1]; wn.r[1] = wn.r[2]; wn.r[2] = row34(bp, t, wx, y + 1);
    } else if (wx == wn.wx && y == wn.y - 1) {
        wn.r[2] = wn.r[1]; wn.r[1] = wn.r[0]; wn.r[0] = row34(bp, t, wx, y - 1);
    } else {
        wn.r[0] = row34(bp, t, wx, y - 1); wn.r[1] = row34(bp, t, wx, y); wn.r[2] = row34(bp, t, wx, y + 1);
    }
    wn.wx = wx;
    wn.y = y;
}
// the 8 neighbours of pixel x of the window's centre row as a mask, bit s = the pixel in direction s
__device__ __forceinline__ uint32_t win_neighbours(const BlobWin& wn, int x) {
    const int b = x & 31;
    const uint32_t up = (uint32_t)(wn.r[0] >> b) & 7u, mid = (uint32_t)(wn.r[1] >> b) & 7u, dn = (uint32_t)(wn.r[2] >> b) & 7u;
    return ((mid >> 2) & 1u) | (((up >> 2) & 1u) << 1) | (((up >> 1) & 1u) << 2) | ((up & 1u) << 3) |
           ((mid & 1u) << 4) | ((dn & 1u) << 5) | (((dn >> 1) & 1u) << 6) | (((dn >> 2) & 1u) << 7);
}

struct BlobNodes {         // per node (= candidate = arc), N of them; planes one after the other, raster order within a plane
    uint32_t* key;         // (y << 16) | (x << 1) | type
    uint32_t* next;        // the node whose start state the arc ends in (itself: a border of one arc; a dead node: itself)
    unsigned long long* a00;  // the arc's share of the contourMoments sum a00 (wrap-around)
    uint32_t* n;           // steps = points of the arc; 0 = dead node
    uint32_t* jmp;         // pointer jumping (leader rounds)
    uint32_t* leader;      // the smallest node of the cycle
    uint32_t* ptr[2];      // list ranking (double-buffered): successor 2^r arcs on, kNoNode at the end of the list
    unsigned long long* sa[2];  // ... a00 summed from this arc to the end of the list
    uint32_t* sn[2];       // ... points from this arc to the end of the list
    int32_t* off;          // per node: first point of its contour in the point arena (at the owner; -1: contour rejected)
};
constexpr size_t kBlobNodeBytes = 4 + 4 + 8 + 4 + 4 + 4 + 8 + 16 + 8 + 4;  // 64

// pass 1: candidates per word (exclusive prefix within the row) and per row.  One workgroup per (row, plane).
__global__ __launch_bounds__(256) void blob_count_kernel(BitPlanes bp, uint32_t* wordpre, uint32_t* rowcnt) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    const int y = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    const uint32_t* row = bp.bits + ((long long)t * bp.h + y) * bp.wpr;
    uint32_t* pre = wordpre + ((long long)t * bp.h + y) * bp.wpr;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int w0 = 0; w0 < bp.wpr; w0 += 256) {
        const int wx = w0 + tid;
        uint32_t c = 0;
        if (wx < bp.wpr) {
            const uint32_t cur = row[wx];
            if (cur) {
                uint32_t outer, hole;
                word_candidates(cur, wx > 0 ? row[wx - 1] >> 31 : 0u, wx + 1 < bp.wpr ? row[wx + 1] & 1u : 0u, wx, bp.w, outer, hole);
                c = __popc(outer) + __popc(hole);
            }
        }
        part[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {  // inclusive scan
            const uint32_t v = tid >= d ? part[tid - d] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (wx < bp.wpr) pre[wx] = carry + part[tid] - c;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) rowcnt[t * bp.h + y] = carry;
}

// pass 2: first node of every row WITHIN its plane and the nodes per plane (one workgroup per plane), then the first
// node of every plane and the total (blob_bases_kernel).  counters: [3] = N, [4 + t] = nodes of plane t,
// [4 + kNumThresh + t] = first node of plane t.
__global__ __launch_bounds__(256) void blob_rowscan_kernel(int h, const uint32_t* rowcnt, uint32_t* rowoff, int* counters) {
    __shared__ uint32_t part[256];
    __shared__ uint32_t carry;
    const int tid = threadIdx.x, t = blockIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int y0 = 0; y0 < h; y0 += 256) {
        const int y = y0 + tid;
        const uint32_t c = y < h ? rowcnt[t * h + y] : 0u;
        part[tid] = c;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {
            const uint32_t v = tid >= d ? part[tid - d] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (y < h) rowoff[t * h + y] = carry + part[tid] - c;
        __syncthreads();
        if (tid == 255) carry += part[255];
        __syncthreads();
    }
    if (tid == 0) counters[4 + t] = (int)carry;
}
__global__ void blob_bases_kernel(int* counters) {
    if (threadIdx.x != 0) return;
    uint32_t base = 0;
    for (int t = 0; t < kNumThresh; ++t) {
        counters[4 + kNumThresh + t] = (int)base;
        base += (uint32_t)counters[4 + t];
    }
    counters[3] = (int)base;
}

// node of candidate (x, type) of row y of plane t; `r1` = the row as 34 bits around word x >> 5 (row34)
__device__ __forceinline__ uint32_t node_of(const BitPlanes& bp, int t, int x, int y, int type, unsigned long long r1,
                                            const uint32_t* wordpre, const uint32_t* rowoff, uint32_t plane_base) {
    const int wx = x >> 5, b = x & 31;
    uint32_t outer, hole;
    word_candidates((uint32_t)(r1 >> 1), (uint32_t)(r1 & 1ull), (uint32_t)(r1 >> 33) & 1u, wx, bp.w, outer, hole);
    const uint32_t below = (1u << b) - 1u;
    return plane_base + rowoff[t * bp.h + y] + wordpre[((long long)t * bp.h + y) * bp.wpr + wx] + __popc(outer & below) +
           __popc(hole & below) + (type ? (outer >> b) & 1u : 0u);
}

// pass 3: the keys of the nodes, in place (thread per word)
__global__ __launch_bounds__(256) void blob_keys_kernel(BitPlanes bp, const uint32_t* wordpre, const uint32_t* rowoff, uint32_t* key,
                                                        const int* counters) {
    const int wx = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, t = blockIdx.z;
    if (wx >= bp.wpr) return;
    const uint32_t* row = bp.bits + ((long long)t * bp.h + y) * bp.wpr;
    const uint32_t cur = row[wx];
    if (cur == 0) return;
    uint32_t outer, hole;
    word_candidates(cur, wx > 0 ? row[wx - 1] >> 31 : 0u, wx + 1 < bp.wpr ? row[wx + 1] & 1u : 0u, wx, bp.w, outer, hole);
    uint32_t k = (uint32_t)counters[4 + kNumThresh + t] + rowoff[t * bp.h + y] + wordpre[((long long)t * bp.h + y) * bp.wpr + wx];
    uint32_t both = outer | hole;
    while (both) {
        const int i = __ffs(both) - 1;
        both &= both - 1;
        const uint32_t xy = ((uint32_t)y << 16) | ((uint32_t)(wx * 32 + i) << 1);
        if ((outer >> i) & 1u) key[k++] = xy;
        if ((hole >> i) & 1u) key[k++] = xy | 1u;
    }
}

// The arc that starts at candidate `key` of plane t.  WRITE = false: -> (next node, steps, a00 share); a dead node
// (a single pixel, or the hole-type twin of an outer-type candidate in the same state) has 0 steps and is its own
// successor.  WRITE = true: stores the arc's points ((y << 16) | x) to pts[0 .. steps).
template <bool WRITE>
__device__ __forceinline__ void blob_arc(const BitPlanes& bp, int t, uint32_t key, uint32_t self, const uint32_t* wordpre,
                                         const uint32_t* rowoff, uint32_t plane_base, uint32_t& next, uint32_t& steps,
                                         unsigned long long& a00, uint32_t* pts, int* err) {
    const int x0 = (int)((key >> 1) & 0x7fffu), y0 = (int)(key >> 16), is_hole = (int)(key & 1u);
    next = self;
    steps = 0;
    a00 = 0;
    BlobWin wn;
    wn.wx = -2;
    wn.y = -2;
    win_goto(wn, bp, t, x0, y0);
    uint32_t m = win_neighbours(wn, x0);
    int s = first_cw(m, is_hole ? 0 : 4);  // the predecessor of the start on the border
    if (s < 0) return;                     // a single pixel: area 0, never a blob
    if (is_hole && !(m & 0x10u) && first_cw(m, 4) == s) return;  // the outer-type candidate of this pixel is this state
    int x3 = x0, y3 = y0;
    uint32_t n = 0;
    constexpr unsigned long long kMask34 = (1ull << 34) - 1ull;
    for (;;) {
        const int from = s;
        s = first_ccw(m, s);  // the next border point
        // A straight horizontal run in one go.  Having come from the west and going east means that the three pixels
        // below were looked at first and are black, and the walk goes on east for as long as that stays so and the pixel
        // ahead is white (likewise westwards under a black row above): no candidate state on the way -- the pixel behind
        // is white (no outer-type start) and so is the one ahead (no hole-type start).  The run is read off the window's
        // rows with bit operations, up to the window's edge (32 steps); without this an arc along the frame's edge is
        // thousands of dependent iterations of ~150 instructions in one lane, and the launch lasts as long as that lane.
        int k = 1;
        const int b = x3 & 31;
        if (s == 0 && from == 4) {
            const unsigned long long dn = ~wn.r[2] & kMask34;
            const unsigned long long ok = dn & (dn >> 1) & (dn >> 2) & (wn.r[1] >> 2);  // bit b + j: step j of the run is possible
            k = __ffsll((long long)~(ok >> b)) - 1;
        } else if (s == 4 && from == 0) {
            const unsigned long long up = ~wn.r[0] & kMask34;
            const unsigned long long ok = up & (up >> 1) & (up >> 2) & wn.r[1];         // bit b - j
            k = __clzll((long long)~(ok << (63 - b)));
        }
        const int x4 = x3 + k * kDX[s], y4 = y3 + kDY[s];
        if (WRITE) {
            for (int j = 0; j < k; ++j) pts[n + j] = ((uint32_t)y3 << 16) | (uint32_t)(x3 + j * kDX[s]);
        } else {
            // contourMoments, the pairs (point -> next point) of the k steps: x * y' - x' * y each; in a horizontal run - y or + y
            a00 += k == 1 ? (unsigned long long)((long long)x3 * y4 - (long long)x4 * y3)
                          : (unsigned long long)((long long)k * (s == 0 ? -(long long)y3 : (long long)y3));
        }
        n += (uint32_t)k;
        // now at (x4, y4), having come from direction s + 4
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
        win_goto(wn, bp, t, x3, y3);
        m = win_neighbours(wn, x3);
        const bool outer_c = !(m & 0x10u) && first_cw(m, 4) == s;
        const bool hole_c = !(m & 1u) && first_cw(m, 0) == s;  // (the last column too: pseudo_node)
        if (outer_c || hole_c) {  // a candidate's start state (possibly this arc's own: a border of one arc)
            if (!WRITE) next = node_of(bp, t, x3, y3, outer_c ? 0 : 1, wn.r[1], wordpre, rowoff, plane_base);
            break;
        }
        if (n >= (uint32_t)kMaxArc) {
            if (!WRITE) atomicOr(err, 1);
            break;
        }
    }
    steps = n;
}

__device__ __forceinline__ int plane_of(uint32_t node, const int* counters) {
    int t = 0;
#pragma unroll
    for (int u = 1; u < kNumThresh; ++u) t += node >= (uint32_t)counters[4 + kNumThresh + u];
    return t;
}

// pass 4: every arc
__global__ __launch_bounds__(256) void blob_arcs_kernel(BitPlanes bp, const uint32_t* wordpre, const uint32_t* rowoff, BlobNodes nd,
                                                        int* counters) {
    const uint32_t N = (uint32_t)counters[3];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    uint32_t next, steps;
    unsigned long long a00;
    const int t = plane_of(i, counters);
    blob_arc<false>(bp, t, nd.key[i], i, wordpre, rowoff, (uint32_t)counters[4 + kNumThresh + t], next, steps, a00, nullptr, counters + 2);
    nd.next[i] = next;
    nd.n[i] = steps;
    nd.a00[i] = a00;
    nd.jmp[i] = next;
    nd.leader[i] = pseudo_node(nd.key[i], bp.w) ? kNoNode : i;
}

// pass 5, R rounds: the smallest node of every cycle.  Double-buffered ((jmp, leader) <-> the two list-ranking
// pointer arrays, which are free until pass 6): a round must see a node's minimum and its pointer from the SAME round,
// or the minimum it takes over may not cover the stretch the pointer skips.
__global__ __launch_bounds__(256) void blob_leader_round_kernel(const uint32_t* jmp, const uint32_t* lead, uint32_t* jmp_out,
                                                                uint32_t* lead_out, const int* counters) {
    const uint32_t N = (uint32_t)counters[3];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    const uint32_t j = jmp[i];
    lead_out[i] = min(lead[i], lead[j]);
    jmp_out[i] = jmp[j];
}

// pass 6: the cycle cut open at its owner, then R rounds of list ranking (suffix sums towards the end of the list)
__global__ __launch_bounds__(256) void blob_rank_init_kernel(BlobNodes nd, const int* counters) {
    const uint32_t N = (uint32_t)counters[3];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    const uint32_t nx = nd.next[i];
    nd.ptr[0][i] = nx == nd.leader[i] ? kNoNode : nx;
    nd.sa[0][i] = nd.a00[i];
    nd.sn[0][i] = nd.n[i];
    nd.off[i] = -1;
}
__global__ __launch_bounds__(256) void blob_rank_round_kernel(BlobNodes nd, int from, const int* counters) {
    const uint32_t N = (uint32_t)counters[3];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N) return;
    const int to = from ^ 1;
    const uint32_t p = nd.ptr[from][i];
    unsigned long long a = nd.sa[from][i];
    uint32_t n = nd.sn[from][i], q = kNoNode;
    if (p != kNoNode) {
        a += nd.sa[from][p];
        n += nd.sn[from][p];
        q = nd.ptr[from][p];
    }
    nd.ptr[to][i] = q;
    nd.sa[to][i] = a;
    nd.sn[to][i] = n;
}

struct BlobContour {      // one border that passed the area filter
    uint32_t key;         // its start (raster position and type): the order of discovery
    int32_t t;            // threshold index
    int32_t n;            // points
    uint32_t points_off;  // first point in the arena
    long long a00, a10, a01, a20, a11, a02;  // contourMoments sums (a00 from the device, the rest from the points on the host)
};

// pass 7: the owners: filterByArea on m00 = |a00| / 2 (minArea 20 <= m00 < maxArea 80000, exact in integers).
// WRITE = false: how many contours pass and how many points they have (counters [0], [1]) -- the host sizes the
// record and point arrays from that; WRITE = true: a record and a place in the point arena for each of them.
template <bool WRITE>
__global__ __launch_bounds__(256) void blob_records_kernel(BlobNodes nd, int rk, BlobContour* recs, int* counters) {
    const uint32_t N = (uint32_t)counters[3];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N || nd.leader[i] != i || nd.n[i] == 0) return;
    const long long a00 = (long long)nd.sa[rk][i];
    const long long a = a00 < 0 ? -a00 : a00;
    if (a < 2 * 20 || a >= 2 * 80000) return;
    const int n = (int)nd.sn[rk][i];
    if (!WRITE) {
        atomicAdd(counters + 0, 1);
        atomicAdd(counters + 1, n);
        return;
    }
    BlobContour c;
    c.key = nd.key[i];
    c.t = plane_of(i, counters);
    c.n = n;
    c.a00 = a00;
    c.a10 = c.a01 = c.a20 = c.a11 = c.a02 = 0;
    const int r = atomicAdd(counters + 5 + 2 * kNumThresh, 1);
    c.points_off = (unsigned)atomicAdd(counters + 6 + 2 * kNumThresh, n);
    nd.off[i] = (int)c.points_off;
    recs[r] = c;
}

// pass 8: the points of the contours that passed, every arc its own stretch: the arc's offset in its contour is
// what precedes it in the list from the owner = total - (points from this arc to the end)
__global__ __launch_bounds__(256) void blob_points_kernel(BitPlanes bp, const uint32_t* wordpre, const uint32_t* rowoff, BlobNodes nd,
                                                          int rk, uint32_t* pts, int* counters) {
    const uint32_t N = (uint32_t)counters[3];
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= N || nd.n[i] == 0) return;
    const uint32_t L = nd.leader[i];
    const int off = nd.off[L];
    if (off < 0) return;
    const uint32_t total = nd.sn[rk][L], at = total - nd.sn[rk][i];
    uint32_t next, steps;
    unsigned long long a00;
    blob_arc<true>(bp, plane_of(i, counters), nd.key[i], i, wordpre, rowoff, 0u, next, steps, a00, pts + off + at, nullptr);
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
    if (m00 == 0.0) return false;
    out->x = m10 / m00;
    out->y = m01 / m00;
    out->confidence = ratio * ratio;
    {  // filterByColor, blobColor 0: the binarised pixel at the rounded centre must be dark.  (blobdetector.cpp checks it
       // after the convexity; the filters are independent conditions, and this one spares the contours of the WHITE
       // regions -- half of all -- their convex hull)
        const int ix = (int)std::nearbyint(out->x), iy = (int)std::nearbyint(out->y);  // cvRound
        if (ix < 0 || ix >= w || iy < 0 || iy >= h) return false;
        if (img[(size_t)iy * stride + ix] > kThresh0 + kThreshStep * c.t) return false;
    }
    std::vector<IPt> p((size_t)c.n);
    for (int i = 0; i < c.n; ++i) p[i] = IPt{(int)(pts[i] & 0xffffu), (int)(pts[i] >> 16)};
    {  // filterByConvexity, minConvexity 0.95
        const double carea = polygon_area(p.data(), c.n), harea = hull_area(p);
        if (std::fabs(harea) < DBL_EPSILON) return false;
        const double conv = carea / harea;
        if (conv < (double)0.95f || conv >= FLT_MAX) return false;  // Params::minConvexity is a float: 0.949999988079...
    }
    std::vector<double> d((size_t)c.n);
    for (int i = 0; i < c.n; ++i) {
        const double dx = out->x - p[i].x, dy = out->y - p[i].y;
        d[i] = std::sqrt(dx * dx + dy * dy);
    }
    // the two middle order statistics (blobdetector.cpp sorts the whole list; the values are the same)
    const size_t lo = (size_t)(c.n - 1) / 2, hi = (size_t)c.n / 2;
    std::nth_element(d.begin(), d.begin() + lo, d.end());
    const double dlo = d[lo];
    const double dhi = hi == lo ? dlo : *std::min_element(d.begin() + lo + 1, d.end());
    out->radius = (dlo + dhi) / 2.;
    return true;
}

// contourMoments' integer sums of a closed contour from its points, wrap-around like the arithmetic of the device
// (the exact results fit; moments.cpp accumulates the same terms in double)
void contour_sums(BlobContour& c, const uint32_t* pts) {
    unsigned long long a00 = 0, a10 = 0, a01 = 0, a20 = 0, a11 = 0, a02 = 0;
    for (int i = 0; i < c.n; ++i) {
        const uint32_t p = pts[i], q = pts[i + 1 < c.n ? i + 1 : 0];
        const long long xa = (long long)(p & 0xffffu), ya = (long long)(p >> 16), xb = (long long)(q & 0xffffu), yb = (long long)(q >> 16);
        const long long dxy = xa * yb - xb * ya, xii = xa + xb, yii = ya + yb;
        a00 += (unsigned long long)dxy;
        a10 += (unsigned long long)(dxy * xii);
        a01 += (unsigned long long)(dxy * yii);
        a20 += (unsigned long long)(dxy * (xa * xii + xb * xb));
        a11 += (unsigned long long)(dxy * (xa * (yii + ya) + xb * (yii + yb)));
        a02 += (unsigned long long)(dxy * (ya * yii + yb * yb));
    }
    c.a00 = (long long)a00; c.a10 = (long long)a10; c.a01 = (long long)a01;
    c.a20 = (long long)a20; c.a11 = (long long)a11; c.a02 = (long long)a02;
}

}  // namespace

size_t blob_scratch_bytes(int w, int h, BlobScratchLayout* lay) {
    BlobScratchLayout L;
    L.wpr = (w + 31) / 32;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) / 256 * 256; return o; };
    L.o_counters = take(64 * sizeof(int));
    L.o_bits = take((size_t)kNumThresh * h * L.wpr * 4);
    L.o_wordpre = take((size_t)kNumThresh * h * L.wpr * 4);
    L.o_rowcnt = take((size_t)kNumThresh * h * 4);
    L.o_rowoff = take((size_t)kNumThresh * h * 4);
    if (lay) *lay = L;
    return off;
}

// d_img: the frame on the device; h_img: the same pixels on the host (colour filter).  `node_scratch(bytes)` and
// `out_scratch(bytes)` return device memory of at least that size for the per-candidate arrays and for the records and
// points of the contours that pass the area filter (how many there are is only known after they have been counted:
// no capacity to run out of).  Appends the keypoints as (x, y) * 1000 ints in SimpleBlobDetector's output order.
// false on a device error or when the frame has more borders than the scratch holds (err says which).
bool blob_detect(const uint8_t* d_img, int d_stride, const uint8_t* h_img, int h_stride, int w, int h, void* scratch,
                 const std::function<void*(size_t)>& node_scratch, const std::function<void*(size_t)>& out_scratch, hipStream_t s,
                 std::vector<int32_t>& xy_out, std::string& err) {
    if (w <= 0 || h <= 0) return true;
    if (w > 32767 || h > 65535) { err = "blob detector: frames up to 32767 x 65535"; return false; }
    BlobScratchLayout L;
    blob_scratch_bytes(w, h, &L);
    char* base = (char*)scratch;
    int* counters = (int*)(base + L.o_counters);  // [0] records, [1] points, [2] error, [3] nodes, [4 + t] of plane t, [21 + t] first of plane t, [39], [40] write cursors
    uint32_t* bits = (uint32_t*)(base + L.o_bits);
    uint32_t* wordpre = (uint32_t*)(base + L.o_wordpre);
    uint32_t* rowcnt = (uint32_t*)(base + L.o_rowcnt);
    uint32_t* rowoff = (uint32_t*)(base + L.o_rowoff);
    BlobContour* recs = nullptr;
    uint32_t* pts = nullptr;
    const BitPlanes bp{bits, w, h, L.wpr};
    hipMemsetAsync(counters, 0, 64 * sizeof(int), s);
    const dim3 grid_rows((L.wpr + 255) / 256, h);
    hipLaunchKernelGGL(blob_bitplanes_kernel, grid_rows, dim3(256), 0, s, d_img, d_stride, w, h, L.wpr, bits);
    hipLaunchKernelGGL(blob_count_kernel, dim3(h, kNumThresh), dim3(256), 0, s, bp, wordpre, rowcnt);
    hipLaunchKernelGGL(blob_rowscan_kernel, dim3(kNumThresh), dim3(256), 0, s, h, (const uint32_t*)rowcnt, rowoff, counters);
    hipLaunchKernelGGL(blob_bases_kernel, dim3(1), dim3(64), 0, s, counters);
    int hc[64];
    if (hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        err = "blob detector: device error";
        return false;
    }
    const long long N = (uint32_t)hc[3];
    if (N > (1ll << 27)) { err = "blob detector: more border starts than it takes (pure noise?)"; return false; }
    if (N > 0) {
        char* nb = (char*)node_scratch((size_t)N * kBlobNodeBytes + 16 * 256);
        if (!nb) { err = "blob detector: out of device memory"; return false; }
        BlobNodes nd;
        size_t off = 0;
        auto take = [&](size_t bytes) { char* p = nb + off; off += (bytes * (size_t)N + 255) / 256 * 256; return p; };
        nd.a00 = (unsigned long long*)take(8);
        nd.sa[0] = (unsigned long long*)take(8);
        nd.sa[1] = (unsigned long long*)take(8);
        nd.key = (uint32_t*)take(4);
        nd.next = (uint32_t*)take(4);
        nd.n = (uint32_t*)take(4);
        nd.jmp = (uint32_t*)take(4);
        nd.leader = (uint32_t*)take(4);
        nd.ptr[0] = (uint32_t*)take(4);
        nd.ptr[1] = (uint32_t*)take(4);
        nd.sn[0] = (uint32_t*)take(4);
        nd.sn[1] = (uint32_t*)take(4);
        nd.off = (int32_t*)take(4);
        int maxn = 1;
        for (int t = 0; t < kNumThresh; ++t) maxn = std::max(maxn, hc[4 + t]);
        int rounds = 1;  // 2^rounds >= the longest cycle (in arcs) there can be
        while ((1ll << rounds) < maxn) ++rounds;
        const dim3 gn((unsigned)((N + 255) / 256));
        hipLaunchKernelGGL(blob_keys_kernel, dim3(grid_rows.x, grid_rows.y, kNumThresh), dim3(256), 0, s, bp, (const uint32_t*)wordpre,
                           (const uint32_t*)rowoff, nd.key, (const int*)counters);
        hipLaunchKernelGGL(blob_arcs_kernel, gn, dim3(256), 0, s, bp, (const uint32_t*)wordpre, (const uint32_t*)rowoff, nd, counters);
        for (int r = 0; r < (rounds + 1) / 2 * 2; ++r)  // (an even number: the result is back in (jmp, leader))
            if (r & 1)
                hipLaunchKernelGGL(blob_leader_round_kernel, gn, dim3(256), 0, s, (const uint32_t*)nd.ptr[0], (const uint32_t*)nd.ptr[1],
                                   nd.jmp, nd.leader, (const int*)counters);
            else
                hipLaunchKernelGGL(blob_leader_round_kernel, gn, dim3(256), 0, s, (const uint32_t*)nd.jmp, (const uint32_t*)nd.leader,
                                   nd.ptr[0], nd.ptr[1], (const int*)counters);
        hipLaunchKernelGGL(blob_rank_init_kernel, gn, dim3(256), 0, s, nd, (const int*)counters);
        int rk = 0;
        for (int r = 0; r < rounds; ++r, rk ^= 1)
            hipLaunchKernelGGL(blob_rank_round_kernel, gn, dim3(256), 0, s, nd, rk, (const int*)counters);
        // how many contours pass the area filter, and how many points they have: the arrays for them are sized from that
        hipLaunchKernelGGL(blob_records_kernel<false>, gn, dim3(256), 0, s, nd, rk, recs, counters);
        if (hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) {
            err = "blob detector: device error";
            return false;
        }
        if (hc[2]) { err = "blob detector: a border arc longer than 2^20 steps"; return false; }
        if (hc[0] > 0) {
            const size_t rec_bytes = ((size_t)hc[0] * sizeof(BlobContour) + 255) / 256 * 256;
            char* ob = (char*)out_scratch(rec_bytes + (size_t)hc[1] * 4);
            if (!ob) { err = "blob detector: out of device memory"; return false; }
            recs = (BlobContour*)ob;
            pts = (uint32_t*)(ob + rec_bytes);
            hipLaunchKernelGGL(blob_records_kernel<true>, gn, dim3(256), 0, s, nd, rk, recs, counters);
            hipLaunchKernelGGL(blob_points_kernel, gn, dim3(256), 0, s, bp, (const uint32_t*)wordpre, (const uint32_t*)rowoff, nd, rk,
                               pts, counters);
        }
    } else {
        hc[0] = hc[1] = 0;
    }
    std::vector<BlobContour> hrec((size_t)hc[0]);
    std::vector<uint32_t> hpts((size_t)hc[1]);
    if ((hc[0] && hipMemcpyAsync(hrec.data(), recs, hrec.size() * sizeof(BlobContour), hipMemcpyDeviceToHost, s) != hipSuccess) ||
        (hc[1] && hipMemcpyAsync(hpts.data(), pts, hpts.size() * 4, hipMemcpyDeviceToHost, s) != hipSuccess) ||
        hipStreamSynchronize(s) != hipSuccess) {
        err = "blob detector: download failed";
        return false;
    }
    // contours of a threshold in cv::findContours' RETR_LIST order: discovery order (raster order of the
    // starts), reversed
    std::sort(hrec.begin(), hrec.end(), [](const BlobContour& a, const BlobContour& b) {
        return a.t != b.t ? a.t < b.t : a.key > b.key;
    });
    // the per-contour filters (moments, hull, radius: independent, ~40 ns per contour point) on a few host threads;
    // the grouping below walks the results in order
    std::vector<Center> centers(hrec.size());
    std::vector<char> keep(hrec.size(), 0);
    {
        auto work = [&](size_t lo, size_t hi) {
            for (size_t k = lo; k < hi; ++k) {
                contour_sums(hrec[k], hpts.data() + hrec[k].points_off);
                keep[k] = contour_to_center(hrec[k], hpts.data() + hrec[k].points_off, h_img, w, h, h_stride, &centers[k]);
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nthr = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 16), hpts.size() / 16384 + 1);
        if (nthr <= 1) {
            work(0, hrec.size());
        } else {  // contiguous ranges of about equal point counts
            std::vector<std::thread> pool;
            size_t lo = 0, acc = 0, part = 1;
            for (size_t k = 0; k < hrec.size(); ++k) {
                acc += (size_t)hrec[k].n;
                if (acc * nthr >= hpts.size() * part && part < nthr) {
                    pool.emplace_back(work, lo, k + 1);
                    lo = k + 1;
                    ++part;
                }
            }
            work(lo, hrec.size());
            for (std::thread& th : pool) th.join();
        }
    }
    std::vector<std::vector<Center>> groups;  // blobdetector.cpp detect(): centres of one blob across thresholds
    size_t i = 0;
    for (int t = 0; t < kNumThresh; ++t) {
        std::vector<Center> cur;
        for (; i < hrec.size() && hrec[i].t == t; ++i)
            if (keep[i]) cur.push_back(centers[i]);
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
