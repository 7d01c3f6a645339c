// Shared definitions of the HIP implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct mrgingham_amd_ctx;

namespace mrg {

int fail_hip(mrgingham_amd_ctx* ctx, hipError_t e, const char* what, const char* file, int line);

// Detector constants: compile-time in the reference as well
// (find_chessboard_corners.cc:18-44, :559-564; mrgingham-internal.h:3).
constexpr int kRespMin = 15;         // RESPONSE_MIN_THRESHOLD
constexpr int kPeakMin = 120;        // RESPONSE_MIN_PEAK_THRESHOLD
constexpr int kBlobMinPixels = 2;    // CONNECTED_COMPONENT_MIN_SIZE
constexpr int kVarWindowR = 10;      // CONSTANCY_WINDOW_R
constexpr int kVarMin = 20 * 20;     // STDEV_THRESHOLD^2
constexpr int kMargin = 7;           // ChESS ring radius 5 + blur border 2 (ChESS.c:61)
constexpr double kGridScale = 1000.; // FIND_GRID_SCALE

// One pyramid level of a batch, as every kernel of the level sees it.
struct LevelBatch {
    int nframes;
    int w, h;                 // level image size
    const uint8_t* img;       // level images (the caller's frames at level 0)
    long long img_pitch;      // bytes between frames
    int img_stride;           // bytes between rows
    int16_t* resp;            // dense w*h int16 per frame
    long long resp_pitch;     // elements between frames
    // engine-clock probe of the response kernels (mrgingham_amd_sclk_mhz): when set, workgroup 0 of the launch adds the
    // shader cycles (s_memtime) and the constant-rate ticks (s_memrealtime) of its own run to clk[0] / clk[1]
    unsigned long long* clk = nullptr;
};

// the probe's two halves (all threads of the workgroup call them; `on` is workgroup-uniform)
struct ClockProbe {
    unsigned long long c0, r0;
};
__device__ __forceinline__ ClockProbe clock_probe_begin(bool on) {
    ClockProbe p{0, 0};
    if (on) {
        p.c0 = __builtin_readcyclecounter();
        p.r0 = __builtin_amdgcn_s_memrealtime();
    }
    return p;
}
__device__ __forceinline__ void clock_probe_end(bool on, const ClockProbe& p, unsigned long long* clk) {
    if (on && threadIdx.x == 0) {
        atomicAdd(clk, (unsigned long long)__builtin_readcyclecounter() - p.c0);
        atomicAdd(clk + 1, (unsigned long long)__builtin_amdgcn_s_memrealtime() - p.r0);
    }
}

// The response kernels cut a frame into `nsegs` row segments of whole granules (G rows: one loop iteration of the
// kernel), as equal as whole granules allow: the first (granules % nsegs) segments have one granule more, the frame's
// ragged last granule is in the last segment.  Equal segments matter: workgroups go to the CUs round robin, and a
// periodic mix of tall and short ones (256 + 256 + 256 + 256 + 56 rows at 1080) puts the tall ones on the same CUs
// -- measured 117 us against 107.5 us for 4 x 272 on 64 frames of 1920x1080.
__host__ __device__ __forceinline__ void segment_rows(int h, int nsegs, int i, int G, int& ys, int& ye) {
    const int granules = (h + G - 1) / G, q = granules / nsegs, r = granules - q * nsegs;
    ys = (i * q + (i < r ? i : r)) * G;
    const int y1 = ys + (q + (i < r ? 1 : 0)) * G;
    ye = y1 < h ? y1 : h;
}
// how many segments a requested segment height (rows) makes of a frame: at least 1, at most one per granule
inline int segments_for_rows(int h, int rows, int G) {
    const int granules = (h + G - 1) / G;
    int n = rows > 0 ? (h + rows - 1) / rows : 1;
    return n < 1 ? 1 : (n > granules ? granules : n);
}

// How many segments: the cost model both response kernels share, fitted to tools/seg_rounds_sweep.py (64 frames of 640x480
// .. 4096x3072, every k that leaves segments of 32 rows or more).  Workgroups are dealt to the 256
// CUs round robin (to the XCDs first), and the kernels are VALU-bound with a CU's slots full, so a launch costs what the
// BUSIEST CU computes -- its workgroups x (rows of a segment + `per_wg` rows' worth of prologue and epilogue that the
// other resident workgroups do not hide) -- plus a drain while the last workgroups run with the CU half empty, a fraction
// `drain` of one workgroup's own run time (its rows + `latency` rows: the staged prologue in full).  Picks within 1-2 %
// of the measured best height at every size of the sweep; the curve is flat around the optimum.
struct SegModel {
    double per_wg;    // rows' worth of un-hidden time per workgroup
    double drain;     // fraction of a workgroup's run time the launch drains over
    double latency;   // rows' worth of time before a workgroup's first response row
    int min_rows;     // shortest segment considered
    int max_rows;     // tallest segment considered
};
inline int pick_balanced_segments(int strips, int h, int nframes, int G, const SegModel& m) {
    const int granules = (h + G - 1) / G;
    int best = 1;
    double best_cost = 0;
    for (int k = 1; k <= granules; ++k) {
        const int rows = ((granules + k - 1) / k) * G;  // the tallest segment
        if (rows > m.max_rows && k < granules) continue;
        if (rows < m.min_rows && best_cost != 0) break;
        const long long nwg = (long long)strips * k * nframes;
        const long long per_cu = ((nwg + 7) / 8 + 31) / 32;
        const double cost = (double)per_cu * ((double)granules * G / k + m.per_wg) + m.drain * (rows + m.latency);
        if (best_cost == 0 || cost < 0.999 * best_cost) {  // (ties go to the taller segments)
            best = k;
            best_cost = cost;
        }
    }
    return best;
}

// Per-frame tables of the component search, all of capacity `cap` entries per
// frame unless noted.  "hot" pixels are those with response > kRespMin: the
// only pixels that can seed, join or extend a component.
struct CompTables {
    int cap;                  // hot-pixel capacity per frame
    int32_t* hot_cnt;         // [nframes]     number of hot-list entries made (may exceed cap)
    uint32_t* hot_xy;         // [nframes*cap] the hot list: pixel as (y << 16) | x; kHotDead = unused slot
    int32_t* parent;          // [nframes*cap] union-find forest over hot-list indices
    int32_t* comp_cnt;        // [nframes*cap] hot pixels per super-component (at its root)
    int4* comp_box;           // [nframes*cap] (xmin, ymin, xmax, ymax) at the root
    int32_t* roots;           // [nframes*cap] compacted root list / claim table (refine)
    int32_t* comp_first;      // [nframes*cap] smallest (y << 16) | x of the super-component (at its root)
    // pixel -> hot-list index, one entry per aligned group of 8 pixels of a row: .x = list index of the
    // group's first hot pixel, .y = 8-bit mask of its hot pixels.  The hot pixels of a group are
    // consecutive list entries in ascending x, so index(x, y) = .x + popcount(.y & ((1 << (x & 7)) - 1)).
    // Written only for groups that have a hot pixel, read only at pixels known to be hot: never
    // initialised.  1 byte per pixel (a dense int32-per-pixel index took 4).
    uint2* gidx;              // [nframes*gidx_pitch]
    long long gidx_pitch;     // entries between frames = gw * h
    int gw;                   // groups per row = ceil(w / 8)
    uint32_t* arena;          // [nframes*arena_cap] DFS stacks
    long long arena_cap;      // entries per frame
    int cand_cap;             // candidate capacity per frame
    struct Cand* cand;        // [nframes*cand_cap]
    unsigned long long* sortkeys; // [nframes*cand_cap_pow2]
    int sort_cap;             // power of two >= cand_cap
    int32_t* status;          // [nframes] bit 0: hot table overflow, bit 1: candidate overflow
    // Component search out of LDS (cc.hip, "LDS path"): path[f] = 1 when frame f was handled there, 0 when
    // it is left to the global-memory kernels (more hot pixels / components than the LDS tables hold), 2 when
    // a banded refinement was begun there and the global-memory kernel finishes it.
    int32_t* path;            // [nframes]
    int lds_path;             // 0: the LDS kernels are not launched and path[] is not consulted
    // Dense repeat of the frames a sparse refinement could not take (api.hip, queue_sparse_levels): when set, the
    // response kernels work on the frames LISTED here -- only[0] = how many, only[1 ..] = which -- with a grid that is
    // laid out for kOnlySlots frames (a workgroup takes frames slot, slot + kOnlySlots, ... of the list): the list is
    // nearly always empty, and a workgroup of the response kernel that finds nothing to do has still waited for 40 KB
    // of LDS and 500 registers on a chip the pixel stream keeps full.  NULL everywhere else.
    const int32_t* only;      // [1 + nframes]
};

// A component that passed the size / peak / margin tests and waits for the
// variance test and the ordering by seed (find_chessboard_corners.cc:193-209).
struct Cand {
    unsigned long long sum_rx, sum_ry, sum_r;
    int32_t seed;             // (y << 16) | x of the seed pixel: orders like its raster index (output order key)
    uint16_t x_peak, y_peak;
    int32_t ok;               // set by the variance stage
    int32_t pad;
};

enum : int { kStatusHotOverflow = 1, kStatusCandOverflow = 2, kStatusSparse = 4 };  // (bits 8.. of a hot overflow: the demand)
// sparse refinement: words per frame of the cell-list header -- count (-1: the frame was given up), log2 cell size,
// then the span of the cell bitmap: first cell x, y, cells per row, rows
constexpr int kCellHdr = 8;
constexpr int kOnlySlots = 2;  // frames the grid of a response launch restricted to a frame list (CompTables::only) is laid out for
constexpr uint32_t kHotDead = 0xffffffffu;       // hot list slot that holds no pixel
constexpr uint32_t kHotSingleton = 0x80000000u;  // flag in hot_xy[]: the pixel has no hot 4-neighbour

#define MRG_HIP_CHECK(expr)                                                                          \
    do {                                                                                             \
        hipError_t _e = (expr);                                                                      \
        if (_e != hipSuccess) return ::mrg::fail_hip(ctx, _e, #expr, __FILE__, __LINE__);            \
    } while (0)

}  // namespace mrg
