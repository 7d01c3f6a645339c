// Exact, order-preserving connected-component search on the GPU (gfx950).
//
// The reference (find_chessboard_corners.cc:159-267, :284-397) walks the clamped
// response in raster order and flood-fills with a LIFO whose running-maximum
// threshold makes the result depend on visiting order.  It is reproduced
// bit-exactly, in parallel, from two facts:
//
//  (1) only "hot" pixels (response > 15) are ever accumulated or expanded: a
//      pixel in (0,15] that gets pushed is popped, found invalid and zeroed with
//      no other effect (:243-247), so it can simply not be pushed;
//  (2) a fill never leaves the 4-connected region of hot pixels that contains
//      its seed (a "super-component"), responses only ever decrease to 0, and
//      the margin flag depends only on coordinates (:216-221).  Super-components
//      are therefore independent of each other; only WITHIN one must the
//      reference's sequence (raster order of seeds, push order +x,-x,+y,-y,
//      first-maximum-wins) be replayed, and that is done by a single lane.
//
// Two implementations of that, tried in this order per frame (CompTables::path says which one took it):
//   * out of LDS (second half of this file): the hot list, the values of the listed pixels, a hash map and
//     the LIFOs of a frame -- or of a band of it -- in 40 KB; the common case;
//   * in global memory (first half): one 512-thread workgroup per frame, every hand-off a workgroup barrier
//     (no cross-XCD traffic, no grid sync):
//       P0-P2 union-find over the hot list (left/up neighbours), flatten, per-root count / box / first pixel
//       P3    one lane per root: scan its box in raster order, replay the fills (detect)  |  group the points
//             that share super-components, one lane per group replays them in index order (refine)
//       P4    21x21 variance test of every surviving component (:50-88), same lane
//       P5    order by seed raster index (detect: bitonic sort) and emit coordinates.
//
// Floating point: centroid, level rescaling and the *1000 rounding are the
// reference's exact double expressions (:262-263, :278-279, :350-351); the file
// is compiled with -ffp-contract=off so no FMA changes a truncation.
#include "common.h"
#include "hotlist.h"
#include "kernels.h"

namespace mrg {

// Timing ablations and phase clocks (they change results or write debug data) exist only in builds made with
// -DMRG_EXPERIMENT (tools/build_variant.sh); the shipped library ignores those bits of CompTables::lds_path.
#ifdef MRG_EXPERIMENT
#define MRG_EXP(bits) ((bits) != 0)
#else
#define MRG_EXP(bits) false
#endif

constexpr int CC_THREADS = 256;

// Every table of a frame is only ever touched by ONE workgroup per kernel (the detect / refine kernels
// run one workgroup per frame), so the atomics on them are
// WORKGROUP scope: they execute in the XCD's L2.  Agent-scope atomics on this multi-XCD part go to
// the memory side instead; a few hundred thousand of them per level were slowing the HBM-streaming
// pixel kernels they run underneath by ~6 % (measured by replacing the labelling kernels with empty
// ones).  Kernel boundaries make the results visible to the next kernel.
#define MRG_WG __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP
__device__ __forceinline__ int aload(const int32_t* p) { return __hip_atomic_load(p, MRG_WG); }
__device__ __forceinline__ int wg_min(int32_t* p, int v) { return __hip_atomic_fetch_min(p, v, MRG_WG); }
__device__ __forceinline__ int wg_max(int32_t* p, int v) { return __hip_atomic_fetch_max(p, v, MRG_WG); }
__device__ __forceinline__ int wg_add(int32_t* p, int v) { return __hip_atomic_fetch_add(p, v, MRG_WG); }
__device__ __forceinline__ int wg_or(int32_t* p, int v) { return __hip_atomic_fetch_or(p, v, MRG_WG); }
// status word of a hot-list overflow: the flag + the number of hot pixels the frame has, in units of 64, above bit 8
// (the host grows the tables to that, api.hip mrgingham_amd_sync)
__device__ __forceinline__ int hot_overflow_status(int hot_cnt) {
    const uint32_t units = ((uint32_t)hot_cnt + 63u) >> 6;
    return (int)((uint32_t)kStatusHotOverflow | ((units > 0x7fffffu ? 0x7fffffu : units) << 8));
}
// Reports it in the frame's status word: the flag bits are OR-ed, the demand field keeps the MAXIMUM -- status words
// accumulate until the host looks at them, and pipelined calls that reuse a scratch set must not OR two demands into a
// number neither frame asked for (the tables would over-grow by up to 2x).  One thread per frame calls this.
__device__ __forceinline__ void report_hot_overflow(int32_t* word, int hot_cnt) {
    const uint32_t want = (uint32_t)hot_overflow_status(hot_cnt);
    uint32_t old = (uint32_t)aload(word);
    while (true) {
        const uint32_t demand = (old >> 8) > (want >> 8) ? (old >> 8) : (want >> 8);
        const uint32_t merged = ((old | want) & 0xffu) | (demand << 8);
        if (merged == old) return;
        int32_t expected = (int32_t)old;
        if (__hip_atomic_compare_exchange_strong(word, &expected, (int32_t)merged, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_WORKGROUP))
            return;
        old = (uint32_t)expected;  // somebody else's flag arrived in between: merge again
    }
}

__device__ __forceinline__ int uf_root(const int32_t* parent, int i) {
    int p = aload(parent + i);
    while (p != i) {
        i = p;
        p = aload(parent + i);
    }
    return i;
}

// Lock-free union by minimum index.  A failed atomicMin (the target stopped
// being a root meanwhile) still leaves the forest connected: continue with the
// displaced parent.
__device__ __forceinline__ void uf_unite(int32_t* parent, int a, int b) {
    while (true) {
        a = uf_root(parent, a);
        b = uf_root(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = wg_min(parent + a, b);
        if (old == a) return;
        a = old;
    }
}

struct FrameView {
    int w, h, n;  // level size, number of hot pixels
    const uint8_t* img;
    int img_stride;
    int16_t* d;
    uint32_t* hot_xy;
    int32_t *parent, *comp_cnt, *roots, *comp_first;
    const uint2* gidx;
    int gw;
    int4* comp_box;
    uint32_t* arena;
    long long arena_cap;
    Cand* cand;
    int cand_cap;
    unsigned long long* sortkeys;
    int sort_cap;
    int32_t* status;
};

__device__ __forceinline__ FrameView make_view(const LevelBatch& lb, const CompTables& t, int frame) {
    FrameView v;
    v.w = lb.w;
    v.h = lb.h;
    v.img = lb.img + (long long)frame * lb.img_pitch;
    v.img_stride = lb.img_stride;
    v.d = lb.resp + (long long)frame * lb.resp_pitch;
    const long long e = (long long)frame * t.cap;
    v.hot_xy = t.hot_xy + e;
    v.parent = t.parent + e;
    v.comp_cnt = t.comp_cnt + e;
    v.roots = t.roots + e;
    v.comp_first = t.comp_first + e;
    v.comp_box = t.comp_box + e;
    v.gidx = t.gidx + (long long)frame * t.gidx_pitch;
    v.gw = t.gw;
    v.arena = t.arena + (long long)frame * t.arena_cap;
    v.arena_cap = t.arena_cap;
    v.cand = t.cand + (long long)frame * t.cand_cap;
    v.cand_cap = t.cand_cap;
    v.sortkeys = t.sortkeys + (long long)frame * t.sort_cap;
    v.sort_cap = t.sort_cap;
    v.status = t.status + frame;
    const int cnt = t.hot_cnt[frame];
    v.n = cnt < t.cap ? cnt : t.cap;
    return v;
}

// P0, P1 and P2 open the global-memory kernels below, which run one 512-thread workgroup per frame (so that
// workgroup-scope atomics suffice, see above): at full resolution a noisy frame has tens of thousands
// of hot pixels, nearly all of them isolated, which P1 flags so that P2 skips them.  (They were a kernel of
// their own until the LDS path made these kernels the rare case: one launch per level less on the component
// stream, whose kernels mostly find their frames done and leave.)
// 512 threads at no more than 64 VGPRs: the two waves per SIMD of such a workgroup fit into the registers ONE
// retiring wave of the pixel kernels frees (128).  With 1024 threads the kernels -- which mostly only look at
// the path word and leave -- waited for two: 0.98 -> 1.06 ms per step.
constexpr int CCG_THREADS = 512;
// P0: reset the per-entry tables (the pixel kernel only writes the hot list and the pixel -> index
// map); P1: union-find over the hot list (left / up neighbours); P2: flatten (parent[i] = root of i),
// per-root pixel count, bounding box and smallest raster position.  Workgroup barriers in between; all
// threads of the workgroup call it.
__device__ __forceinline__ void label_frame(const FrameView& v) {
    const int w = v.w;
    for (int i = threadIdx.x; i < v.n; i += CCG_THREADS) {
        v.parent[i] = i;
        v.comp_cnt[i] = 0;
        v.comp_box[i] = make_int4(0x7fffffff, 0x7fffffff, -1, -1);
        v.roots[i] = 0x7fffffff;
        v.comp_first[i] = 0x7fffffff;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < v.n; i += CCG_THREADS) {
        const uint32_t e = v.hot_xy[i];
        if (e == kHotDead) continue;  // unused slot: flagged already (bit 31), P2 skips it
        const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
        const int p = y * w + x;
        // all four neighbours lie inside the image (p is in [7,w-7) x [7,h-7)); the frame is zero
        const bool l = v.d[p - 1] > kRespMin, u = v.d[p - w] > kRespMin;
        const bool r = v.d[p + 1] > kRespMin, dn = v.d[p + w] > kRespMin;
        // the hot pixels of an aligned group of 8 are consecutive list entries in ascending x
        if (l) uf_unite(v.parent, i, (x & 7) ? i - 1 : hot_index_of(v.gidx, v.gw, x - 1, y));
        if (u) uf_unite(v.parent, i, hot_index_of(v.gidx, v.gw, x, y - 1));
        // An isolated hot pixel is a finished super-component of size one: flag it so that P2 does
        // not spend five atomics on it (at full resolution most hot pixels are isolated noise).
        if (!(l || u || r || dn)) v.hot_xy[i] = e | kHotSingleton;
    }
    __syncthreads();  // every union of the frame is done (the tables are only touched by this workgroup)
    for (int i = threadIdx.x; i < v.n; i += CCG_THREADS) {
        const uint32_t e = v.hot_xy[i];
        if (e & kHotSingleton) continue;  // its own root, count 0: never a blob, never shared
        const int r = uf_root(v.parent, i);
        // other threads may still walk through i: r is an ancestor, so their walks stay valid
        __hip_atomic_store(v.parent + i, r, MRG_WG);
        const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
        int* box = reinterpret_cast<int*>(v.comp_box + r);
        wg_min(box + 0, x);
        wg_min(box + 1, y);
        wg_max(box + 2, x);
        wg_max(box + 3, y);
        wg_add(v.comp_cnt + r, 1);
        wg_min(v.comp_first + r, (int)e);  // (y << 16) | x orders like the raster index
    }
    __syncthreads();
}

void launch_cc_detect_lds(const LevelBatch& lb, const CompTables& t, int level, const DetectOut& out, int frame0,
                          int nframes, hipStream_t s);
void launch_cc_refine_lds(const LevelBatch& lb, const CompTables& t, int level, const RefineIO& io, int frame0,
                          int nframes, hipStream_t s);

// Hot list of a CALLER-SUPPLIED response (mrgingham_amd_cc_on_response_batch, the entry point the
// rule tests drive): copies it into the level scratch the way the component search expects it --
// negatives clamped to 0 (find_chessboard_corners.cc:527-529), the 7-pixel frame zero (:506) -- and
// appends the hot pixels exactly as the ChESS epilogue does (same list / map format).
// One wave = 64 aligned groups of 8 pixels of one row.
__global__ __launch_bounds__(256) void hot_from_response_kernel(const int16_t* src, LevelBatch lb, CompTables t,
                                                                int frame0) {
    const int frame = frame0 + blockIdx.z;
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (y >= lb.h) return;  // wave-uniform
    const int w = lb.w, h = lb.h;
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 8;
    const int16_t* in = src + (long long)frame * w * h + (long long)y * w;
    int16_t* out = lb.resp + (long long)frame * lb.resp_pitch + (long long)y * w;
    const bool row_in = y >= kMargin && y < h - kMargin;
    uint32_t bits = 0;
    for (int i = 0; i < 8; ++i) {
        const int x = x0 + i;
        if (x >= w) break;
        int v = in[x];
        if (v < 0 || !row_in || x < kMargin || x >= w - kMargin) v = 0;
        out[x] = (int16_t)v;
        bits |= (uint32_t)(v > kRespMin) << i;
    }
    append_groups_wave(t, frame, bits, ((uint32_t)y << 16) | (uint32_t)x0);
}

void launch_hot_from_response(const int16_t* src, const LevelBatch& lb, const CompTables& t, int frame0, int nframes,
                              hipStream_t s) {
    if (nframes <= 0 || lb.w <= 0 || lb.h <= 0) return;
    const dim3 grid((t.gw + 63) / 64, (lb.h + 3) / 4, nframes);
    hipLaunchKernelGGL(hot_from_response_kernel, grid, dim3(256), 0, s, src, lb, t, frame0);
}

struct Blob {
    unsigned long long srx, sry, sr;
    int npix, rmax, xpk, ypk;
    bool touched;
};

// Drains the LIFO exactly like follow_connected_component (:236-256) and returns how many hot
// pixels it consumed (zeroed).  Latency is what matters here (one lane, dependent global
// accesses, usually underneath a bandwidth-saturating pixel kernel), so per pop there is ONE
// round trip for the pixel and its four neighbours together (their addresses do not depend on
// the pixel's value) and the top of the stack lives in a register (the entry pushed last is the
// one popped next).
__device__ __forceinline__ int drain_lifo(int16_t* d, int w, int h, uint32_t* stk, int sp, Blob& b) {
    b.srx = b.sry = b.sr = 0;
    b.npix = 0;
    b.rmax = 0;
    b.xpk = b.ypk = 0;
    b.touched = false;
    int consumed = 0;
    uint32_t top = 0;
    bool has_top = false;
    auto push = [&](uint32_t e) {
        if (has_top) stk[sp++] = top;
        top = e;
        has_top = true;
    };
    while (true) {
        uint32_t e;
        if (has_top) { e = top; has_top = false; }
        else if (sp > 0) e = stk[--sp];
        else break;
        const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
        const int q = y * w + x;
        // q is inside the fill region [7,w-7) x [7,h-7), so all four neighbours are inside the image
        const int v = d[q], vxp = d[q + 1], vxm = d[q - 1], vyp = d[q + w], vym = d[q - w];
        if (v <= 0) continue;                                  // visited already: d[q] = 0 is a no-op
        d[q] = 0;                                              // :245 / :250
        consumed += v > kRespMin;
        if (!(v > kRespMin && v > (b.rmax >> 4))) continue;    // :159-171 with :27
        if (v > b.rmax) { b.rmax = v; b.xpk = x; b.ypk = y; }  // :176-181, first maximum wins
        b.srx += (unsigned long long)(v * x);
        b.sry += (unsigned long long)(v * y);
        b.sr += (unsigned long long)v;
        b.npix++;
        // :252-255 then :216-226 (only hot pixels are worth pushing, see (1) above)
        if (x + 1 >= w - kMargin) b.touched = true;
        else if (vxp > kRespMin) push(e + 1u);
        if (x - 1 < kMargin) b.touched = true;
        else if (vxm > kRespMin) push(e - 1u);
        if (y + 1 >= h - kMargin) b.touched = true;
        else if (vyp > kRespMin) push(e + 0x10000u);
        if (y - 1 < kMargin) b.touched = true;
        else if (vym > kRespMin) push(e - 0x10000u);
    }
    return consumed;
}

__device__ __forceinline__ bool blob_passes_cheap_tests(const Blob& b) {
    return !b.touched && b.npix >= kBlobMinPixels && b.rmax > kPeakMin;  // :259, :205-206
}

// The 21x21 window test of high_variance (:50-88), by the lane that owns the blob, in one
// pass: with S1 = sum(v), S2 = sum(v^2) and the reference's truncated mean m = S1/441,
// sum((v-m)^2) = S2 - 2*m*S1 + 441*m^2 exactly (all integers), so var = that / 441 with the
// same truncations.  The 84 loads of a window (8+8+4+1 bytes per row, never past the window)
// are independent of each other: one round trip instead of a wave-wide phase and a barrier.
__device__ __forceinline__ bool window_variance_high(const uint8_t* img, int stride, int w, int h, int x, int y) {
    constexpr int R = kVarWindowR, D = 2 * R + 1, NPIX = D * D;  // 441
    if (x - R < 0 || x + R >= w || y - R < 0 || y + R >= h) return false;  // :52-57
    const uint8_t* p = img + (long long)(y - R) * stride + (x - R);
    uint32_t s1 = 0, s2 = 0;
#pragma unroll 3
    for (int r = 0; r < D; ++r) {
        uint32_t q[5];
        __builtin_memcpy(q, p, 20);
        const uint32_t last = p[20];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            s1 = __builtin_amdgcn_udot4(q[k], 0x01010101u, s1, false);
            s2 = __builtin_amdgcn_udot4(q[k], q[k], s2, false);
        }
        s1 += last;
        s2 += last * last;
        p += stride;
    }
    const long long mean = s1 / NPIX;                                              // :69-70
    const long long ssd = (long long)s2 - 2 * mean * (long long)s1 + NPIX * mean * mean;
    return ssd / NPIX > kVarMin;                                                   // :80-87
}

// (p + 0.5) * scale - 0.5, find_chessboard_corners.cc:278-279
__device__ __forceinline__ double rescale_coord(double p, double scale) { return (p + 0.5) * scale - 0.5; }

// Block-wide bitonic sort of n_pad (power of two) 64-bit keys in global memory.
__device__ void bitonic_sort(unsigned long long* keys, int n_pad) {
    for (int k = 2; k <= n_pad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_pad; i += (int)blockDim.x) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = keys[i], b = keys[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
                }
            }
            __syncthreads();
        }
}

// Candidates in output order -> coordinates (the reference's exact double expressions), and the chain's
// hand-over to the refinement.  keys[k] & 0xffffffff indexes v.cand; all threads of the workgroup call it.
__device__ __forceinline__ void emit_detect_outputs(const FrameView& v, const unsigned long long* keys, int nvalid,
                                                    int level, const DetectOut& out, int frame) {
    const double scale = (double)(uint16_t)(1u << level);  // :319
    int32_t* oxy = out.xy + (long long)frame * out.capacity * 2;
    const int nout = nvalid < out.capacity ? nvalid : out.capacity;
    // the chain's hand-over to refinement, fused: every candidate becomes a corner at this level
    // ((double)x / 1000, find_grid.cc:353-354; level tags, mrgingham.cc:81-85)
    const int npt = out.points ? (nout < out.points_pitch ? nout : out.points_pitch) : 0;
    double* opt = out.points ? out.points + (long long)frame * out.points_pitch * 2 : nullptr;
    signed char* olv = out.points ? out.levels + (long long)frame * out.points_pitch : nullptr;
    for (int k = threadIdx.x; k < nout; k += (int)blockDim.x) {
        const Cand& cd = v.cand[(uint32_t)(keys[k] & 0xffffffffu)];
        const double cx = (double)cd.sum_rx / (double)cd.sum_r;  // :262-263
        const double cy = (double)cd.sum_ry / (double)cd.sum_r;
        const double px = rescale_coord(cx, scale), py = rescale_coord(cy, scale);  // :346
        const int ix = (int)(0.5 + px * kGridScale), iy = (int)(0.5 + py * kGridScale);  // :350-351
        oxy[2 * k + 0] = ix;
        oxy[2 * k + 1] = iy;
        if (k < npt) {
            opt[2 * k + 0] = (double)ix / kGridScale;
            opt[2 * k + 1] = (double)iy / kGridScale;
            olv[k] = (signed char)level;
        }
    }
    if (threadIdx.x == 0) {
        out.counts[frame] = nvalid;
        if (out.points) out.npoints[frame] = npt;
    }
}

// ---------------------------------------------------------------------------
// Detect: process_connected_components, points_scaled_out branch (:330-355)
// ---------------------------------------------------------------------------
// (the body: cc_detect_kernel runs it for one level, cc_detect_levels_kernel for several levels in one grid)
__device__ __forceinline__ void cc_detect_frame(const LevelBatch& lb, const CompTables& t, int level, const DetectOut& out,
                                                int frame) {
    __shared__ int s_nroots, s_ncand, s_arena_full;
    __shared__ unsigned long long s_arena_top;
    // latency-bound and tiny next to the pixel kernels it shares CUs with: take issue priority
    __builtin_amdgcn_s_setprio(3);
    if (t.lds_path && t.path[frame] == 1) return;  // done out of LDS
    if (t.hot_cnt[frame] > t.cap) {  // table overflow: report (with what the frame asked for), produce nothing
        if (threadIdx.x == 0) {
            report_hot_overflow(t.status + frame, t.hot_cnt[frame]);
            out.counts[frame] = -1;
        }
        return;
    }
    const FrameView v = make_view(lb, t, frame);
    if (threadIdx.x == 0) { s_nroots = 0; s_ncand = 0; s_arena_top = 0; s_arena_full = 0; }
    label_frame(v);

    // P3a: compact the roots.  A super-component of a single hot pixel can only ever give a
    // one-pixel blob, which the size test rejects (:205), and nothing else can reach it: skipped.
    for (int i = threadIdx.x; i < v.n; i += CCG_THREADS)
        if (v.parent[i] == i && v.comp_cnt[i] >= kBlobMinPixels) v.roots[atomicAdd(&s_nroots, 1)] = i;
    __syncthreads();
    const int nroots = s_nroots;

    // P3b: one lane per super-component replays the reference's sequence
    const int w = v.w, h = v.h;
    for (int k = threadIdx.x; k < nroots; k += CCG_THREADS) {
        const int r = v.roots[k];
        const int4 box = v.comp_box[r];
        const int cnt = v.comp_cnt[r];
        const unsigned long long off = atomicAdd(&s_arena_top, (unsigned long long)(4 * cnt + 1));
        if (off + (unsigned long long)(4 * cnt + 1) > (unsigned long long)v.arena_cap) { s_arena_full = 1; continue; }
        uint32_t* stk = v.arena + off;
        // seeds live in [8, w-8) x [8, h-8) (:332-333)
        const int ylo = max(box.y, kMargin + 1), yhi = min(box.w, h - kMargin - 2);
        const int xlo = max(box.x, kMargin + 1), xhi = min(box.z, w - kMargin - 2);
        int left = cnt;  // hot pixels of this super-component not consumed yet
        auto fill_from = [&](int x, int y) {
            stk[0] = (uint32_t)x | ((uint32_t)y << 16);  // :338
            Blob b;
            left -= drain_lifo(v.d, w, h, stk, 1, b);
            if (!blob_passes_cheap_tests(b)) return;
            if (!window_variance_high(v.img, v.img_stride, w, h, b.xpk, b.ypk)) return;  // :207
            const int c = atomicAdd(&s_ncand, 1);
            if (c < v.cand_cap) {
                Cand cd;
                cd.sum_rx = b.srx; cd.sum_ry = b.sry; cd.sum_r = b.sr;
                cd.seed = (y << 16) | x;  // orders like the raster index of the seed
                cd.x_peak = (uint16_t)b.xpk; cd.y_peak = (uint16_t)b.ypk;
                cd.ok = 1; cd.pad = 0;
                v.cand[c] = cd;
            }
        };
        // The raster scan meets this super-component first at its smallest raster index.  If that
        // pixel may seed, fill from it straight away; in the common case the fill consumes every
        // hot pixel of the super-component and no scan of the bounding box is needed at all.
        {
            const int e = v.comp_first[r];
            const int y = e >> 16, x = e & 0xffff;
            if (x >= xlo && x <= xhi && y >= ylo && y <= yhi) fill_from(x, y);
        }
        for (int y = ylo; y <= yhi && left > 0; ++y)
            for (int x = xlo; x <= xhi && left > 0; ++x) {
                if (!(v.d[y * w + x] > kRespMin)) continue;                  // is_valid(.., NULL), :335
                if (v.parent[hot_index_of(v.gidx, v.gw, x, y)] != r) continue;  // someone else's super-component
                fill_from(x, y);
            }
    }
    __syncthreads();
    if (s_ncand > v.cand_cap || s_arena_full) {
        if (threadIdx.x == 0) { wg_or(v.status, kStatusCandOverflow); out.counts[frame] = -1; }
        return;
    }
    const int nvalid = s_ncand;

    // P5: order by seed raster index = the reference's output order (:332-353)
    for (int c = threadIdx.x; c < nvalid; c += CCG_THREADS)
        v.sortkeys[c] = ((unsigned long long)(uint32_t)v.cand[c].seed << 32) | (uint32_t)c;
    int n_pad = 1;
    while (n_pad < nvalid) n_pad <<= 1;
    for (int i = nvalid + threadIdx.x; i < n_pad; i += CCG_THREADS) v.sortkeys[i] = ~0ull;
    __syncthreads();
    bitonic_sort(v.sortkeys, n_pad);
    emit_detect_outputs(v, v.sortkeys, nvalid, level, out, frame);
}
__global__ __launch_bounds__(CCG_THREADS, 4) void cc_detect_kernel(LevelBatch lb, CompTables t, int level,
                                                               DetectOut out, int frame0) {
    cc_detect_frame(lb, t, level, out, frame0 + blockIdx.x);
}
// Several levels of the same frames in ONE grid (blockIdx.y = the level's slot): the first pass of the full detector
// searches levels 3, 2 and 1 at once, and three launches of 64 workgroups one behind the other were three times the
// latency of one of 192 (a single frame through find_chessboard: 165 -> 65 us).
__global__ __launch_bounds__(CCG_THREADS, 4) void cc_detect_levels_kernel(DetectLevels a) {
    const int k = blockIdx.y;
    cc_detect_frame(a.lb[k], a.t[k], a.level[k], a.out[k], blockIdx.x);
}

void launch_cc_detect(const LevelBatch& lb, const CompTables& t, int level, const DetectOut& out, int frame0,
                      int nframes, hipStream_t s) {
    if (nframes <= 0) return;
    launch_cc_detect_lds(lb, t, level, out, frame0, nframes, s);
    hipLaunchKernelGGL(cc_detect_kernel, dim3(nframes), dim3(CCG_THREADS), 0, s, lb, t, level, out, frame0);
}

// ---------------------------------------------------------------------------
// Refine: process_connected_components, points_refinement branch (:356-397)
// ---------------------------------------------------------------------------
// (the body: cc_refine_kernel runs it for one level, cc_refine_flagged_levels_kernel level after level)
__device__ __forceinline__ void cc_refine_frame(const LevelBatch& lb, const CompTables& t, int level, const RefineIO& io,
                                                int frame) {
    __shared__ int s_changed, s_nref, s_arena_full;
    __shared__ unsigned long long s_arena_top;
    if (t.lds_path && t.path[frame] == 1) return;  // done out of LDS
    if (t.hot_cnt[frame] > t.cap) {
        if (threadIdx.x == 0) {
            report_hot_overflow(t.status + frame, t.hot_cnt[frame]);
            if (io.nrefined) io.nrefined[frame] = -1;
        }
        return;
    }
    const FrameView v = make_view(lb, t, frame);
    if (threadIdx.x == 0) { s_changed = 0; s_nref = 0; s_arena_top = 0; s_arena_full = 0; }
    label_frame(v);  // (presets v.roots[] to INT_MAX: it is the claim table here)

    const int w = v.w, h = v.h;
    const int npts = min(io.npoints[frame], io.pitch);
    const long long pb = (long long)frame * io.pitch;
    double* pts = io.points + 2 * pb;
    signed char* lv = io.levels + pb;
    int32_t* leader = io.leader + pb;
    int32_t* need = io.need + pb;
    int32_t* nseeds = io.nseeds + pb;
    uint32_t* seeds = io.seeds + 9 * pb;
    int32_t* sroot = io.sroot + 9 * pb;
    const uint16_t coord_scale = (uint16_t)(1u << level);

    // R1: seeds of every refinable point (:362-382), in the reference's push order, and the
    // super-component (root) each seed belongs to
    for (int i = threadIdx.x; i < npts; i += CCG_THREADS) {
        int ns = -1;  // -1: not refinable at this level
        if (lv[i] == level + 1) {
            ns = 0;
            const double lx = rescale_coord(pts[2 * i + 0], 1.0 / coord_scale);  // :369
            const double ly = rescale_coord(pts[2 * i + 1], 1.0 / coord_scale);
            const int x = (int)(lx + 0.5), y = (int)(ly + 0.5);  // :371-372
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy) {
                    const int sx = (int16_t)(x + dx), sy = (int16_t)(y + dy);  // is_valid takes int16_t
                    if (sx < 0 || sx >= w || sy < 0 || sy >= h) continue;
                    const int p = sy * w + sx;
                    if (!(v.d[p] > kRespMin)) continue;
                    seeds[9 * i + ns] = (uint32_t)sx | ((uint32_t)sy << 16);
                    sroot[9 * i + ns] = v.parent[hot_index_of(v.gidx, v.gw, sx, sy)];
                    ++ns;
                }
        }
        nseeds[i] = ns;
        leader[i] = i;
        need[i] = 0;
    }
    __syncthreads();

    // R2: points whose seeds share a super-component must be replayed in index
    // order by one lane; label-propagate the minimum point index over the
    // bipartite graph points <-> super-components until nothing changes.
    while (true) {
        for (int i = threadIdx.x; i < npts; i += CCG_THREADS) {
            const int ns = nseeds[i];
            if (ns <= 0) continue;
            int m = leader[i];
            for (int k = 0; k < ns; ++k) m = min(m, aload(v.roots + sroot[9 * i + k]));
            bool changed = m < leader[i];
            for (int k = 0; k < ns; ++k)
                if (wg_min(v.roots + sroot[9 * i + k], m) > m) changed = true;
            leader[i] = m;
            if (changed) s_changed = 1;
        }
        __syncthreads();
        const int changed = s_changed;
        __syncthreads();
        if (!changed) break;
        if (threadIdx.x == 0) s_changed = 0;
        __syncthreads();
    }

    // R3: stack demand of each group = 4 * (hot pixels of its super-components), counted once
    for (int i = threadIdx.x; i < npts; i += CCG_THREADS) {
        const int ns = nseeds[i];
        for (int k = 0; k < ns; ++k) {
            const int old = wg_or(v.comp_cnt + sroot[9 * i + k], (int)0x80000000);
            if (old >= 0) wg_add(need + leader[i], 4 * old);
        }
    }
    __syncthreads();

    // R4: one lane per group, members in index order (:358); accepted points are written in place
    for (int i = threadIdx.x; i < npts; i += CCG_THREADS) {
        if (nseeds[i] < 0 || leader[i] != i) continue;
        const unsigned long long off = atomicAdd(&s_arena_top, (unsigned long long)(need[i] + 10));
        if (off + (unsigned long long)(need[i] + 10) > (unsigned long long)v.arena_cap) { s_arena_full = 1; continue; }
        uint32_t* stk = v.arena + off;
        for (int j = i; j < npts; ++j) {
            if (nseeds[j] < 0 || leader[j] != i) continue;
            const int ns = nseeds[j];
            for (int k = 0; k < ns; ++k) stk[k] = seeds[9 * j + k];
            Blob b;
            drain_lifo(v.d, w, h, stk, ns, b);
            if (!blob_passes_cheap_tests(b)) continue;
            if (!window_variance_high(v.img, v.img_stride, w, h, b.xpk, b.ypk)) continue;  // :207
            const double cx = (double)b.srx / (double)b.sr;  // :262-263
            const double cy = (double)b.sry / (double)b.sr;
            pts[2 * j + 0] = rescale_coord(cx, (double)coord_scale);  // :390
            pts[2 * j + 1] = rescale_coord(cy, (double)coord_scale);
            lv[j] = (signed char)level;  // :393
            atomicAdd(&s_nref, 1);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (s_arena_full) wg_or(v.status, kStatusCandOverflow);  // points refined so far stay refined; the call fails
        // path 2: the LDS kernel refined the points of the bands it finished before it gave up (their count
        // is in nrefined already); what is refined here is the rest
        const int before = (t.lds_path && t.path[frame] == 2 && io.nrefined) ? io.nrefined[frame] : 0;
        if (io.nrefined) io.nrefined[frame] = s_arena_full ? -1 : before + s_nref;
    }
}

__global__ __launch_bounds__(CCG_THREADS, 4) void cc_refine_kernel(LevelBatch lb, CompTables t, int level,
                                                               RefineIO io, int frame0) {
    __builtin_amdgcn_s_setprio(3);
    cc_refine_frame(lb, t, level, io, frame0 + blockIdx.x);
}

// The dense repeat of the frames a sparse chain reported, in one launch (kernels.h, launch_cc_refine_flagged_levels).
struct RefineLevels {
    LevelBatch lb[kRefineLevelsMax];  // indexed by level
    CompTables t[kRefineLevelsMax];
    int n;
    RefineIO io;
    SparseRestore restore;
    const int32_t* list;  // the frames (launch_sparse_flag_list)
    int32_t* status0;
    int level_stride;
    int32_t* counter;
};
constexpr int kFlaggedSlots = 8;  // workgroups of cc_refine_flagged_levels_kernel: workgroup b takes frames b, b + 8, ... of the list
__global__ __launch_bounds__(CCG_THREADS, 8) void cc_refine_flagged_levels_kernel(RefineLevels a) {  // (<= 64 VGPRs: see CCG_THREADS)
    __builtin_amdgcn_s_setprio(3);
    const int nlisted = a.list[0];
    for (int li = blockIdx.x; li < nlisted; li += kFlaggedSlots) {
        const int frame = a.list[1 + li];
        {   // the points as they were before the first sparse level
            const int n = min(a.io.npoints[frame], a.io.pitch);
            const long long pb = (long long)frame * a.io.pitch;
            for (int i = threadIdx.x; i < n; i += CCG_THREADS) {
                if (a.restore.xy) {  // emit_detect_outputs' hand-over, again
                    const int32_t* xy = a.restore.xy + ((long long)frame * a.restore.xy_pitch + i) * 2;
                    a.io.points[2 * (pb + i) + 0] = (double)xy[0] / kGridScale;
                    a.io.points[2 * (pb + i) + 1] = (double)xy[1] / kGridScale;
                    a.io.levels[pb + i] = (signed char)a.restore.level;
                } else {
                    a.io.points[2 * (pb + i) + 0] = a.restore.pts0[2 * (pb + i) + 0];
                    a.io.points[2 * (pb + i) + 1] = a.restore.pts0[2 * (pb + i) + 1];
                    a.io.levels[pb + i] = a.restore.lv0[pb + i];
                }
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int L = a.n - 1; L >= 0; --L) {
            cc_refine_frame(a.lb[L], a.t[L], L, a.io, frame);
            __syncthreads();  // the level below reads the points and level tags this one wrote
        }
        if (threadIdx.x == 0) {
            for (int L = 0; L < a.n; ++L) atomicAnd(a.status0 + (long long)L * a.level_stride + frame, ~(int)kStatusSparse);
            if (a.counter) atomicAdd(a.counter, 1);
        }
    }
}
void launch_cc_refine_flagged_levels(const LevelBatch* lbs, const CompTables* ts, int nlevels, const RefineIO& io,
                                     const SparseRestore& restore, const int32_t* list, int32_t* status_level0,
                                     int level_stride, int32_t* counter, hipStream_t s) {
    if (nlevels <= 0 || nlevels > kRefineLevelsMax) return;
    RefineLevels a;
    for (int L = 0; L < kRefineLevelsMax; ++L) {
        a.lb[L] = lbs[L < nlevels ? L : 0];
        a.t[L] = ts[L < nlevels ? L : 0];
        a.t[L].lds_path = 0;  // every listed frame is this kernel's
        a.t[L].only = nullptr;
    }
    a.n = nlevels;
    a.io = io;
    a.restore = restore;
    a.list = list;
    a.status0 = status_level0;
    a.level_stride = level_stride;
    a.counter = counter;
    hipLaunchKernelGGL(cc_refine_flagged_levels_kernel, dim3(kFlaggedSlots), dim3(CCG_THREADS), 0, s, a);
}
// The frames a sparse chain reported, as a list: list[0] = how many, list[1 ..] = which (any order).  One workgroup.
__global__ __launch_bounds__(256) void sparse_flag_list_kernel(const int32_t* status0, int nframes, int32_t* list) {
    __shared__ int n;
    if (threadIdx.x == 0) n = 0;
    __syncthreads();
    for (int f = threadIdx.x; f < nframes; f += 256)
        if (status0[f] & kStatusSparse) list[1 + atomicAdd(&n, 1)] = f;
    __syncthreads();
    if (threadIdx.x == 0) list[0] = n;
}
void launch_sparse_flag_list(const int32_t* status_level0, int nframes, int32_t* list, hipStream_t s) {
    hipLaunchKernelGGL(sparse_flag_list_kernel, dim3(1), dim3(256), 0, s, status_level0, nframes, list);
}

void launch_cc_refine(const LevelBatch& lb, const CompTables& t, int level, const RefineIO& io, int frame0,
                      int nframes, hipStream_t s) {
    if (nframes <= 0) return;
    launch_cc_refine_lds(lb, t, level, io, frame0, nframes, s);
    // (sparse refinement: nothing for the global-memory kernel to work on -- a frame the LDS kernel cannot take is reported)
    if (!(t.lds_path & kLdsPathSparse))
        hipLaunchKernelGGL(cc_refine_kernel, dim3(nframes), dim3(CCG_THREADS), 0, s, lb, t, level, io, frame0);
}


// ===========================================================================
// LDS path.  A calibration frame has ~10^3 hot pixels per pyramid level (a dozen per corner), and the
// kernels above spend their time in chains of dependent global accesses (2-3 us each underneath a
// bandwidth-saturating pixel kernel): 50-100 us for the labelling, 100-260 us for the fills.  When a
// frame's hot list fits -- at most LN entries -- the whole search runs out of LDS instead: the list, the
// response VALUES of the listed pixels (the only responses the search ever uses, see (1) at the top), a
// hash map pixel -> entry for the neighbour lookups, labels, and the LIFOs.  Same sequence of
// operations as above, hence the same results bit for bit; the dense response is only read (once per
// hot pixel), never written.  Frames that do not fit (hot pixels, components or LIFO demand) are left
// to the global-memory kernels through CompTables::path.
//
// LDS per workgroup: 40 KB, the slot one ChESS workgroup leaves when it retires (39 952 B there, 40 960 B in
// allocation granules: four per CU).
//
// Frames with MORE hot pixels than the tables hold (a 14x14 board has ~2600 at level 0) are cut into
// horizontal BANDS of at most LN hot pixels each, separated by three consecutive rows without a hot pixel,
// and the same workgroup runs the search band after band on the same tables:
//   * no 4-connected component crosses a row without hot pixels, so every component -- and with it every
//     fill, its running maximum and its order of operations -- lies inside one band;
//   * the 3x3 seed window of a refined point spans three rows, so it cannot hold hot pixels of two bands
//     (they are at least four rows apart): a point is refined in the band its seeds are in, and points
//     that share a component share the band;
//   * the output order of detect is by seed position and is restored by the final sort.
// A frame whose rows do not offer such separators (or with more than kMaxBands * LN hot pixels, or more
// than kBandRows rows) goes to the global-memory kernels like before.  (Round 2 first had a second kernel
// with 4096-entry tables = 80 KB = two ChESS workgroup slots: it waited 100-900 us for two ADJACENT slots to
// fall free underneath the level-0 launch, and BASELINE config 3 as stated was bound by that wait.)
// ===========================================================================
constexpr int kMaxBands = 8;
template <int N>
struct LdsCCT {
    static constexpr int LN = N;              // hot-list entries
    static constexpr int LHASH = 2 * N;       // 16-bit hash slots (load factor <= 0.5)
    static constexpr int LSTK = 5 * N / 2;    // 16-bit LIFO words shared by the fills of a frame
    static constexpr int LROOTS = N / 4;      // super-components with >= 2 pixels (detect)
    static constexpr int LEPT = N / CC_THREADS;  // list entries per thread
    uint32_t xy[LN];              // (y << 16) | x, kHotDead for an unused slot
    int16_t val[LN];              // clamped response of the pixel; 0 once consumed by a fill
    int16_t lab[LN];              // smallest list index of the pixel's super-component
    uint32_t hashw[LHASH / 2];    // two 16-bit slots per word: list index, 0xffff = empty
    union {
        int16_t stk[LSTK];        // LIFOs (list indices)
        int32_t acc[LN];          // per-root accumulators / claim table, before the fills
        unsigned long long keys[LSTK / 4];  // sort keys, after the fills
    } u;
    union {
        // detect, per super-component with >= 2 pixels: list index of its root (later: offset of its member
        // list), pixel count, LIFO demand, list index of its pixel with the smallest raster position
        struct { int16_t root[LROOTS], cnt[LROOTS], soff[LROOTS], fidx[LROOTS]; } r;
        int16_t need16[LN + 512]; // refine: LIFO demand of the super-component, at its root; behind them lead16[LPTS]
    } w;
    int nroots, ncand, top, total, changed, nref, mtop, nload, leak;
    int nbands, best, band_y[kMaxBands + 1], shear;
    uint32_t edge[4];
};
constexpr int LPTS = 512;                    // points per frame the LDS refine kernel takes (LdsCCT::w.need16 has room for it)
constexpr int LPPT = LPTS / CC_THREADS;     // points per thread
// One workgroup slot of the pixel kernels, in LDS allocation granules (1280 B on this part: 39 952 B of a ChESS
// workgroup occupy 40 960, four of them the whole 160 KB): anything above 40 960 B would need two.
static_assert(sizeof(LdsCCT<2048>) <= 40960, "must fit into the LDS slot of one ChESS workgroup");
static_assert(offsetof(LdsCCT<2048>, nroots) >= (8192 + CC_THREADS / 64) * 4, "the band planner's key arrays overlay the tables");

// Fibonacci hashing with an independent multiplier per coordinate: the hot pixels of a calibration board sit on a
// lattice, and ONE multiplier on the packed (y << 16 | x) lets only the low 16 bits of the constant act on y --
// at level 1 of a 14x14 board at 4096x3072 that put the lattice in resonance with the table (11.6 probes per
// miss, 71 at worst; the fills ran 4x longer).  Measured on 16 board / level combinations: 1.03-1.3 probes per
// hit, 1.1-2.1 per miss (tools/hash_probe.py).
// The map is bucketed: a 32-bit word is a bucket of two 16-bit list indices (0xffff = empty), probing goes bucket
// by bucket, and an element sits in the first bucket of its probe sequence that had an empty half when it came.
// A wave pays for the LONGEST probe sequence among its 64 lanes; with two candidates per probe that maximum is
// ~40 % shorter than with one (bench frames, level 0: 2.95 -> 1.73 probes, 14x14 level 1: 6.2 -> 3.4), at the
// same two dependent LDS round trips per probe (the word, then both positions).
template <class LdsCC>
__device__ __forceinline__ uint32_t lds_hash(uint32_t e) {
    static_assert(LdsCC::LHASH == 4096, "the shift below takes the top 11 bits: LHASH / 2 buckets");
    return ((e & 0xffffu) * 0x9E3779B1u + (e >> 16) * 0x85EBCA77u) >> 21;
}
template <class LdsCC>
__device__ __forceinline__ uint32_t lds_next_bucket(uint32_t b) { return (b + 1u) & (uint32_t)(LdsCC::LHASH / 2 - 1); }

template <class LdsCC>
__device__ __forceinline__ void lds_insert(LdsCC& L, uint32_t e, int i) {
    uint32_t b = lds_hash<LdsCC>(e);
    while (true) {
        uint32_t* wp = &L.hashw[b];
        const uint32_t old = __hip_atomic_load(wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        uint32_t nw;
        if ((old & 0xffffu) == 0xffffu) nw = (old & 0xffff0000u) | (uint32_t)i;
        else if ((old >> 16) == 0xffffu) nw = (old & 0xffffu) | ((uint32_t)i << 16);
        else { b = lds_next_bucket<LdsCC>(b); continue; }
        if (atomicCAS(wp, old, nw) == old) return;  // else: the word changed under us, look again
    }
}

// One probe: the bucket's two candidates against pixel q.  Returns true when the lookup is settled (j = list
// index, or -1: a bucket with an empty half ends every probe sequence that reaches it).
template <class LdsCC>
__device__ __forceinline__ bool lds_probe(uint32_t wv, uint32_t xlo, uint32_t xhi, uint32_t q, int& j) {
    const uint32_t lo = wv & 0xffffu, hi = wv >> 16;
    if (lo != 0xffffu && xlo == q) { j = (int)lo; return true; }
    if (hi != 0xffffu && xhi == q) { j = (int)hi; return true; }
    if (lo == 0xffffu || hi == 0xffffu) { j = -1; return true; }
    return false;
}

// list index of pixel e, or -1 when it is not hot
template <class LdsCC>
__device__ __forceinline__ int lds_find(const LdsCC& L, uint32_t e) {
    uint32_t b = lds_hash<LdsCC>(e);
    while (true) {
        const uint32_t wv = L.hashw[b];
        const uint32_t xlo = L.xy[wv & (uint32_t)(LdsCC::LN - 1)], xhi = L.xy[(wv >> 16) & (uint32_t)(LdsCC::LN - 1)];
        int j;
        if (lds_probe<LdsCC>(wv, xlo, xhi, e, j)) return j;
        b = lds_next_bucket<LdsCC>(b);
    }
}

// The four neighbours of pixel e at once: the first probes of the four lookups are independent, so their bucket
// reads and then their position reads go out together (two dependent LDS round trips for all four in the
// common case); whatever is not settled by then continues on its own.
template <class LdsCC>
__device__ __forceinline__ void lds_find4(const LdsCC& L, uint32_t e, int (&j)[4]) {
    const uint32_t q[4] = {e + 1u, e - 1u, e + 0x10000u, e - 0x10000u};
    uint32_t b[4], wv[4], xlo[4], xhi[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) b[k] = lds_hash<LdsCC>(q[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k) wv[k] = L.hashw[b[k]];
#pragma unroll
    for (int k = 0; k < 4; ++k) {  // (any slot: compared in lds_probe)
        xlo[k] = L.xy[wv[k] & (uint32_t)(LdsCC::LN - 1)];
        xhi[k] = L.xy[(wv[k] >> 16) & (uint32_t)(LdsCC::LN - 1)];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (lds_probe<LdsCC>(wv[k], xlo[k], xhi[k], q[k], j[k])) continue;
        uint32_t bk = lds_next_bucket<LdsCC>(b[k]);
        while (true) {
            const uint32_t w2 = L.hashw[bk];
            const uint32_t y0 = L.xy[w2 & (uint32_t)(LdsCC::LN - 1)], y1 = L.xy[(w2 >> 16) & (uint32_t)(LdsCC::LN - 1)];
            if (lds_probe<LdsCC>(w2, y0, y1, q[k], j[k])) break;
            bk = lds_next_bucket<LdsCC>(bk);
        }
    }
}

// Band key of a pixel for shear k (in 1/32 pixels of y per pixel of x, |k| <= 32): k = 0 is the row.  A board
// that is rotated in the image has its corner rows on slanted lines, and no image row between them is free of
// hot pixels -- but a sheared "row" that follows the slant is.
__device__ __forceinline__ int band_key(uint32_t e, int k, int w) {
    const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
    const int ak = k < 0 ? -k : k;
    return y + (((k >= 0 ? x : w - 1 - x) * ak) >> 5);
}
// Two pixels at most 2 apart in x and in y (4-neighbours; two seeds of one 3x3 window) differ in key by at most
// this much: a band boundary with that many empty keys keeps them in one band.
__device__ __forceinline__ int band_gap(int k) {
    const int ak = k < 0 ? -k : k;
    return 2 + (ak ? (2 * ak) / 32 + 1 : 0);
}
// One pass of the workgroup over a frame's hot list (entries that hold a pixel): eight independent loads per thread
// in flight at a time.  As `for (i = tid; i < n; i += CC_THREADS) f(hot[i])` the pass is one global round trip per
// iteration -- 2-3 us each underneath the pixel kernels, i.e. 0.5 ms for the 66 000 entries of a textured frame.
template <class F>
__device__ __forceinline__ void scan_hot_list(const uint32_t* hot, int n, F&& f) {
    constexpr int U = 8;
    for (int i0 = threadIdx.x; i0 < n; i0 += CC_THREADS * U) {
        uint32_t e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * CC_THREADS;
            e[u] = i < n ? hot[i] : kHotDead;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (e[u] != kHotDead) f(e[u]);
    }
}

constexpr int kBandKeys = 8192;  // keys 0 .. h - 1 + (w - 1) * |k| / 32 must stay below this

// One attempt at cutting the frame into bands of at most LN hot pixels along shear k.  Leaves L.nbands,
// L.band_y[0 .. nbands] (key bounds) and L.shear; returns the number of bands, 0 (uniformly) when this shear
// offers no separators.  Uses the table storage as scratch.  All threads call it.  A thread owns 32 consecutive
// keys and keeps their counts, prefix sums and "a band may end here" bits in registers, so that a greedy step
// costs one LDS read, one LDS atomic and two barriers (~10 us per attempt; with every test read from LDS in
// dependent order it was 35-50).
template <class LdsCC>
__device__ __noinline__ int lds_try_bands(LdsCC& L, const FrameView& v, int nraw, int k) {
    constexpr int LN = LdsCC::LN;
    const int tid = threadIdx.x, w = v.w;
    const int nkeys = v.h + (((w - 1) * (k < 0 ? -k : k)) >> 5);
    if (nkeys > kBandKeys) return 0;
    uint32_t* rcw = reinterpret_cast<uint32_t*>(&L);  // hot pixels per key, two 16-bit counters per word
    uint32_t* cumw = rcw + kBandKeys / 2;             // hot pixels below the key, likewise
    uint32_t* part = cumw + kBandKeys / 2;            // per-wave totals
    constexpr int WPT = kBandKeys / 2 / CC_THREADS;   // words per thread = 16 (keys 32 * tid ..)
    static_assert(WPT == 16, "the register arrays below assume 32 keys per thread");
    {
        uint4* z = reinterpret_cast<uint4*>(rcw + WPT * tid);
        z[0] = z[1] = z[2] = z[3] = make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    scan_hot_list(v.hot_xy, nraw, [&](uint32_t e) {
        const int b = band_key(e, k, w);
        if (b < kBandKeys) atomicAdd(&rcw[b >> 1], 1u << ((b & 1) * 16));  // (n <= 16384: a counter cannot carry)
    });
    __syncthreads();
    uint32_t wv[WPT + 2];
    {
        const uint4* src = reinterpret_cast<const uint4*>(rcw + WPT * tid);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint4 x = src[q];
            wv[4 * q] = x.x; wv[4 * q + 1] = x.y; wv[4 * q + 2] = x.z; wv[4 * q + 3] = x.w;
        }
        wv[WPT] = tid + 1 < CC_THREADS ? rcw[WPT * (tid + 1)] : 0u;  // the four keys after mine (the gap test)
        wv[WPT + 1] = tid + 1 < CC_THREADS ? rcw[WPT * (tid + 1) + 1] : 0u;
    }
    uint32_t mine = 0;
    unsigned long long emptym = 0;  // bit q: key 32 * tid + q holds no pixel
#pragma unroll
    for (int q = 0; q < WPT + 2; ++q) {
        const uint32_t lo = wv[q] & 0xffffu, hi = wv[q] >> 16;
        if (q < WPT) mine += lo + hi;
        emptym |= (unsigned long long)(lo == 0) << (2 * q) | (unsigned long long)(hi == 0) << (2 * q + 1);
    }
    // exclusive prefix of `mine` over the workgroup
    uint32_t incl = mine;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if ((tid & 63) >= d) incl += o;
    }
    if ((tid & 63) == 63) part[tid >> 6] = incl;
    __syncthreads();
    uint32_t run = incl - mine, total = 0;
    for (int q = 0; q < CC_THREADS / 64; ++q) {
        if (q < (tid >> 6)) run += part[q];
        total += part[q];
    }
    const uint32_t start = run;  // pixels below key 32 * tid
#pragma unroll
    for (int q = 0; q < WPT; ++q) {
        const uint32_t lo = run, hi = run + (wv[q] & 0xffffu);
        cumw[WPT * tid + q] = lo | (hi << 16);  // (<= 16384: fits)
        run = hi + (wv[q] >> 16);
    }
    // a band may end at key r when keys r .. r + gap - 1 hold no pixel (keys past the frame hold none)
    const int gap = band_gap(k);
    unsigned long long sepm = emptym;
    for (int g = 1; g < gap; ++g) sepm &= emptym >> g;
    const uint32_t sep = (uint32_t)sepm;
    __syncthreads();
    int y0 = 0, nb = 0;
    while (true) {
        if (tid == 0) L.best = -1;
        __syncthreads();
        const uint32_t base = (cumw[y0 >> 1] >> ((y0 & 1) * 16)) & 0xffffu;
        int end = nkeys;
        if (total - base > (uint32_t)LN) {
            // the last key r > y0 the band [y0, r) may end at with at most LN hot pixels in it
            uint32_t ok = 0, below = start;  // pixels below key r
#pragma unroll
            for (int q = 0; q < 2 * WPT; ++q) {
                const int r = 2 * WPT * tid + q;
                ok |= (uint32_t)(r > y0 && r < nkeys && below - base <= (uint32_t)LN) << q;
                below += (q & 1) ? wv[q >> 1] >> 16 : wv[q >> 1] & 0xffffu;
            }
            ok &= sep;
            if (ok) atomicMax(&L.best, 2 * WPT * tid + 31 - __builtin_clz(ok));
            __syncthreads();
            end = L.best;
            if (end < 0) return 0;
        }
        if (tid == 0) L.band_y[nb] = y0;
        ++nb;
        y0 = end;
        if (end >= nkeys) break;
        if (nb == kMaxBands) return 0;
        __syncthreads();  // everybody has read L.best
    }
    if (tid == 0) { L.band_y[nb] = nkeys; L.nbands = nb; L.shear = k; }
    __syncthreads();
    return nb;
}

// Cut the frame into bands of at most LN hot pixels (see the top of this section): rows first, then sheared
// rows along the slopes of the upper and the lower edge of the hot pixels (a rotated board) and between them.
// Returns the number of bands, 0 (uniformly) when the frame cannot be banded.  All threads call it.
template <class LdsCC>
__device__ __noinline__ int lds_plan_bands(LdsCC& L, const FrameView& v, int nraw) {
    constexpr int LN = LdsCC::LN;
    const int tid = threadIdx.x, w = v.w, h = v.h;
    if (nraw <= LN) {
        if (tid == 0) { L.nbands = 1; L.band_y[0] = 0; L.band_y[1] = h; L.shear = 0; }
        __syncthreads();
        return 1;
    }
    if (nraw > LN * kMaxBands || h > kBandKeys) return 0;
    int nb = lds_try_bands(L, v, nraw, 0);
    if (nb) return nb;
    // upper / lower edge of the hot pixels in the left and in the right third of the frame
    if (tid < 4) L.edge[tid] = (tid & 1) ? 0u : 0xffffffffu;  // [0] min left, [1] max left, [2] min right, [3] max right
    __syncthreads();
    {
        uint32_t mn[2] = {0xffffffffu, 0xffffffffu}, mx[2] = {0u, 0u};
        scan_hot_list(v.hot_xy, nraw, [&](uint32_t e) {
            const int x = (int)(e & 0xffffu);
            const int side = 3 * x < w ? 0 : (3 * x >= 2 * w ? 1 : -1);
            if (side >= 0) {
                mn[side] = min(mn[side], e);
                mx[side] = max(mx[side], e);
            }
        });
        for (int sd = 0; sd < 2; ++sd) {
            if (mn[sd] != 0xffffffffu) atomicMin(&L.edge[2 * sd], mn[sd]);
            if (mx[sd] != 0u) atomicMax(&L.edge[2 * sd + 1], mx[sd]);
        }
    }
    __syncthreads();
    const uint32_t e0 = L.edge[0], e1 = L.edge[1], e2 = L.edge[2], e3 = L.edge[3];
    __syncthreads();
    if (e0 == 0xffffffffu || e2 == 0xffffffffu) return 0;  // nothing in one of the thirds: not a board that spans the frame
    auto slope32 = [](uint32_t a, uint32_t b) {  // shear that takes pixel a (left) and pixel b (right) to the same key
        const int dx = (int)(b & 0xffffu) - (int)(a & 0xffffu), dy = (int)(b >> 16) - (int)(a >> 16);
        int k = dx > 0 ? (-dy * 32 + (dy < 0 ? dx / 2 : -dx / 2)) / dx : 0;
        return k < -32 ? -32 : (k > 32 ? 32 : k);
    };
    const int kt = slope32(e0, e2), kb = slope32(e1, e3), km = (kt + kb) / 2;
    const int cand[9] = {km, kt, kb, km + 1, km - 1, kt + 1, kt - 1, kb + 1, kb - 1};
    for (int c = 0; c < 9; ++c) {
        const int k = cand[c];
        if (k == 0 || k < -32 || k > 32) continue;
        bool seen = false;
        for (int p = 0; p < c; ++p) seen = seen || cand[p] == k;
        if (seen) continue;
        nb = lds_try_bands(L, v, nraw, k);
        if (nb) return nb;
    }
    return 0;
}

// follow_connected_component (:236-256) on the LDS tables; the LIFO holds list indices.  The four neighbours of
// every entry have been looked up beforehand (lds_build_neighbours): a pop is two dependent LDS round trips
// (entry: value, position, neighbours; then the neighbours' values) instead of eleven through the hash map; the
// refine kernel's fills went from 58 to 36 us per launch with it (level 0 of the bench frames).
constexpr uint32_t kNoNb = 0xfffu;  // 12 bits per neighbour: a list index (< 2048) or this
template <class LdsCC>
__device__ __forceinline__ int drain_nb(LdsCC& L, const uint32_t* nlo, const uint16_t* nhi, int w, int h, int16_t* stk, int sp,
                                        Blob& b) {
    b.srx = b.sry = b.sr = 0;
    b.npix = 0;
    b.rmax = 0;
    b.xpk = b.ypk = 0;
    b.touched = false;
    int consumed = 0;
    while (sp > 0) {
        const int i = stk[--sp];
        const int v = L.val[i];
        const uint32_t e = L.xy[i];
        const uint32_t lo = nlo[i], hi = nhi[i];
        if (v <= 0) continue;  // visited already
        const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
        const uint32_t jxp = lo & 0xfffu, jxm = (lo >> 12) & 0xfffu, jyp = (lo >> 24) | ((hi & 0xfu) << 8), jym = hi >> 4;
        // the neighbours' values do not depend on v: read together
        const int vxp = jxp != kNoNb ? (int)L.val[jxp] : 0, vxm = jxm != kNoNb ? (int)L.val[jxm] : 0;
        const int vyp = jyp != kNoNb ? (int)L.val[jyp] : 0, vym = jym != kNoNb ? (int)L.val[jym] : 0;
        L.val[i] = 0;  // :245 / :250
        ++consumed;    // every listed pixel is hot
        if (!(v > (b.rmax >> 4))) continue;                    // :159-171 with :27 (v > 15 holds)
        if (v > b.rmax) { b.rmax = v; b.xpk = x; b.ypk = y; }  // :176-181, first maximum wins
        b.srx += (unsigned long long)(v * x);
        b.sry += (unsigned long long)(v * y);
        b.sr += (unsigned long long)v;
        b.npix++;
        // :252-255 then :216-226; a neighbour is worth pushing only while it is hot and unvisited
        if (x + 1 >= w - kMargin) b.touched = true;
        else if (vxp > 0) stk[sp++] = (int16_t)jxp;
        if (x - 1 < kMargin) b.touched = true;
        else if (vxm > 0) stk[sp++] = (int16_t)jxm;
        if (y + 1 >= h - kMargin) b.touched = true;
        else if (vyp > 0) stk[sp++] = (int16_t)jyp;
        if (y - 1 < kMargin) b.touched = true;
        else if (vym > 0) stk[sp++] = (int16_t)jym;
    }
    return consumed;
}

// The neighbour table of drain_nb: 48 bits per entry (+x, -x, +y, -y at 12 bits each), the low 32 over the hash
// map (which must be dead, and a barrier behind its last reader), the high 16 wherever the caller has 2 bytes per
// entry to spare (refine: the labels; detect: the front of the LIFO space).  The loader has looked the neighbours up
// for the labelling already and parked them in global scratch; every thread fetches its own entries back (one
// coalesced round trip: 2-3 us where looking them up a second time took 9-25).  All threads call it.
template <class LdsCC>
__device__ __forceinline__ void lds_build_neighbours(LdsCC& L, const FrameView& v, int n, uint16_t* nhi) {
    constexpr int LEPT = LdsCC::LEPT;
    static_assert(sizeof(L.hashw) >= (size_t)LdsCC::LN * 4, "32 bits per entry over the hash map");
    static_assert(LdsCC::LN <= (int)kNoNb, "12-bit list indices");
    const int tid = threadIdx.x;
    const uint2* parked = reinterpret_cast<const uint2*>(v.arena);
    uint32_t* nlo = L.hashw;
    static_assert(LEPT % 4 == 0, "four entries at a time");
#pragma unroll
    for (int k0 = 0; k0 < LEPT; k0 += 4) {
        uint2 p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + CC_THREADS * (k0 + k);
            p[k] = i < n ? parked[i] : make_uint2(0, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = tid + CC_THREADS * (k0 + k);
            if (i < n) {
                nlo[i] = p[k].x;
                nhi[i] = (uint16_t)p[k].y;
            }
        }
    }
    __syncthreads();
}

// Load the hot pixels with band keys in [y0, y1) (`banded`; otherwise the whole list as it stands) into LDS,
// label the super-components (lab = smallest list index) and leave in L.u.acc, at every root, (pixels of the
// super-component) | (sum of hot-neighbour counts << 13): the latter bounds the pushes of any fill of it.
// n = entries loaded.  Returns false (uniformly) when they do not fit.  All threads call it.
// Window mode of the loader (refinement of frames with far more hot pixels than the tables hold -- a textured
// scene): only the hot pixels in the CELLS around the points to refine are loaded.  `WinSel` is a bitmap over cells
// of 2^cs x 2^cs pixels (cw cells per row); the cell of every refinable point and its eight neighbours are marked, so
// a point's 3x3 seeds are at least 2^cs pixels away from the edge of what is loaded.  A super-component that
// reaches that edge -- a member with a hot 4-neighbour in an unmarked cell -- is flagged "open" at its root
// (`openbits`); the caller declines the frame if a seed falls into an open one (its fill could leave the loaded
// set).  Everything else about the search only ever looks at the super-components of the seeds, so leaving the
// rest of the frame's hot pixels out changes nothing.
struct WinSel {
    const uint32_t* bits;  // LDS
    uint32_t* openbits;    // LDS, LN bits
    int cs;                // cells of 2^cs pixels, on the grid that starts at pixel (0, 0); -1: no selection
    int ox, oy, cw, chh;   // the bitmap covers cells ox .. ox + cw - 1, oy .. oy + chh - 1 (nothing outside is marked)
    // BOXED = false: the bitmap spans the frame from cell (0, 0) (the dense schedule's selection): no bounds to check,
    // and the kernel that only ever asks this way does not keep the span in registers.  BOXED = true is a level of a
    // sparse chain: the dense response only holds the marked cells, and a neighbour in an unmarked cell is taken to be
    // hot.  (A template parameter, not a flag in here: as a run-time flag in the two scans over a textured frame's 6e4
    // hot pixels it cost the dense schedule 95 us per refinement launch.)
    template <bool BOXED>
    __device__ __forceinline__ bool marked(int x, int y) const {
        int c;
        if (!BOXED) {
            c = (y >> cs) * cw + (x >> cs);
        } else {
            const int cx = (x >> cs) - ox, cy = (y >> cs) - oy;
            if ((unsigned)cx >= (unsigned)cw || (unsigned)cy >= (unsigned)chh) return false;
            c = cy * cw + cx;
        }
        return (bits[c >> 5] >> (c & 31)) & 1u;
    }
};

template <bool BOXED = false, class LdsCC>
__device__ __forceinline__ bool lds_load_and_label(LdsCC& L, const FrameView& v, int nraw, int cap, bool banded, int y0,
                                                   int y1, int& n, const WinSel* win = nullptr, bool preloaded = false) {
    constexpr int LN = LdsCC::LN, LHASH = LdsCC::LHASH, LEPT = LdsCC::LEPT;
    const int tid = threadIdx.x;
    n = 0;
    if (win) banded = true;  // a selection out of the frame's list, like a band
    if (nraw > cap || (!banded && nraw > LN)) return false;
    const int w = v.w;
    for (int k = tid; k < LHASH / 2; k += CC_THREADS) L.hashw[k] = 0xffffffffu;
    const int shear = L.shear;
    if (tid == 0) { L.nroots = 0; L.top = 0; L.total = 0; L.changed = 0; L.mtop = 0; L.nload = 0; L.leak = 0; }
    __syncthreads();
    if (preloaded) {  // (sparse refinement: the whole frame's pixels, all in marked cells, are in L.xy already)
        n = nraw;
    } else if (banded) {
        scan_hot_list(v.hot_xy, nraw, [&](uint32_t e) {
            bool take = true;
            if (win) take = win->template marked<BOXED>((int)(e & 0xffffu), (int)(e >> 16));
            // (BOXED: cells AND a band for a frame of a sparse level whose cells hold more than the tables)
            if (!win || (BOXED && y1 > y0)) {
                const int y = band_key(e, shear, w);
                take = take && y >= y0 && y < y1;
            }
            if (take) {
                const int slot = atomicAdd(&L.nload, 1);
                if (slot < LN) L.xy[slot] = e;
            }
        });
        __syncthreads();
        n = L.nload;
        if (n > LN) return false;  // bands: the planner counted the same pixels, cannot happen; windows: too many
    } else {
        n = nraw;
    }
    uint32_t own[LEPT];
#pragma unroll
    for (int k = 0; k < LEPT; ++k) {
        const int i = tid + CC_THREADS * k;
        own[k] = kHotDead;
        if (i < n) {
            const uint32_t e = banded ? L.xy[i] : v.hot_xy[i];
            own[k] = e;
            L.xy[i] = e;
            L.lab[i] = (int16_t)i;
            L.u.acc[i] = 0;
            if (e != kHotDead) {
                L.val[i] = v.d[(int)(e >> 16) * w + (int)(e & 0xffffu)];
                lds_insert(L, e, i);
            } else {
                L.val[i] = 0;
            }
        }
    }
    __syncthreads();
    // the four neighbours of every entry, 12 bits each (kNoNb = none), packed like the table of drain_nb
    uint32_t nlo[LEPT];
    uint16_t nhi[LEPT];
#pragma unroll
    for (int k = 0; k < LEPT; ++k) {
        const uint32_t e = own[k];
        int f[4] = {-1, -1, -1, -1};
        if (e != kHotDead) lds_find4(L, e, f);
        uint32_t j[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) j[q] = f[q] < 0 ? kNoNb : (uint32_t)f[q];
        nlo[k] = j[0] | (j[1] << 12) | (j[2] << 24);
        nhi[k] = (uint16_t)((j[2] >> 8) | (j[3] << 4));
        // parked for lds_build_neighbours (same thread, same entries) in the LIFO arena of the global-memory
        // kernels, which nothing uses while a frame is searched out of LDS: 2 words per entry of >= 16384
        const int i = tid + CC_THREADS * k;
        if (i < n) reinterpret_cast<uint2*>(v.arena)[i] = make_uint2(nlo[k], nhi[k]);
    }
    auto nb_of = [&](int k, int q) -> uint32_t {  // q static after unrolling
        return q == 0 ? nlo[k] & 0xfffu : q == 1 ? (nlo[k] >> 12) & 0xfffu
             : q == 2 ? (nlo[k] >> 24) | (((uint32_t)nhi[k] & 0xfu) << 8) : (uint32_t)nhi[k] >> 4;
    };
    // min-label propagation with shortcutting; labels only ever decrease and always name a member of
    // the same super-component, so unsynchronised reads within a round are harmless
    while (true) {
        bool ch = false;
#pragma unroll
        for (int k = 0; k < LEPT; ++k) {
            if (own[k] == kHotDead) continue;
            const int i = tid + CC_THREADS * k;
            const int cur = L.lab[i];
            int m = cur;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t jq = nb_of(k, q);
                if (jq != kNoNb) m = min(m, (int)L.lab[jq]);
            }
            m = min(m, (int)L.lab[m]);
            if (m < cur) { L.lab[i] = (int16_t)m; ch = true; }
        }
        if (ch) L.changed = 1;
        __syncthreads();
        const int c = L.changed;
        __syncthreads();
        if (!c) break;
        if (tid == 0) L.changed = 0;
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < LEPT; ++k) {
        if (own[k] == kHotDead) continue;
        const int i = tid + CC_THREADS * k;
        int deg = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) deg += nb_of(k, q) != kNoNb;
        atomicAdd(&L.u.acc[L.lab[i]], 1 + (deg << 13));
    }
    __syncthreads();
    if (win) {
        // open super-components (a pass of its own, rolled: the unrolled loops above hold eight entries' state in
        // registers).  A neighbour that is not in the list is either not hot (its cell is loaded) or was not loaded;
        // the neighbours come back from where the loop above parked them.
        const uint2* parked = reinterpret_cast<const uint2*>(v.arena);
#pragma unroll 1
        for (int i = tid; i < n; i += CC_THREADS) {
            const uint32_t e = L.xy[i];
            if (e == kHotDead) continue;
            const uint2 pk = parked[i];
            const uint32_t nb[4] = {pk.x & 0xfffu, (pk.x >> 12) & 0xfffu, (pk.x >> 24) | ((pk.y & 0xfu) << 8), (pk.y >> 4) & 0xfffu};
            const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
            bool open = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nx = x + (q == 0) - (q == 1), ny = y + (q == 2) - (q == 3);
                if (nb[q] == kNoNb && nx >= 0 && nx < w && ny >= 0 && ny < v.h && !win->template marked<BOXED>(nx, ny) &&
                    (BOXED || v.d[ny * w + nx] > kRespMin))  // (BOXED: a sparse level, nothing computed there -- taken to be hot)
                    open = true;
            }
            if (open) {
                const int r = L.lab[i];
                atomicOr(&win->openbits[r >> 5], 1u << (r & 31));
            }
        }
        __syncthreads();
    }
    return true;
}

// Marks the cells around the refinable points of a frame (see WinSel).  All threads call these.
// TIGHT (sparse refinement): the cells that overlap the square of half a cell around every seed instead of the
// seed's cell and its eight neighbours -- at most 2 x 2 per seed, a seed is then >= 2^(cs-1) pixels from the
// edge of what is marked -- and the bitmap only spans the box around the points, which buys cells of 16 pixels
// instead of 32 for a board that fills a quarter of a 12 MP frame (the response is computed in every marked cell).
struct NoCellSink { __device__ __forceinline__ void operator()(int, int, int) const {} };
// A listed cell: (cell y << 16) | (subset << 12) | cell x -- cells are >= 16 pixels, a side is < 32768, so x and y are < 2048;
// the subset (0 .. kSubsets - 1) is the workgroup of the refinement kernel that owns the cell (below: "Several workgroups").
__device__ __forceinline__ int cell_x(uint32_t c) { return (int)(c & 0xfffu); }
__device__ __forceinline__ int cell_y(uint32_t c) { return (int)(c >> 16); }
__device__ __forceinline__ int cell_sub(uint32_t c) { return (int)((c >> 12) & 0xfu); }
constexpr int kWinWords = 1280;  // cell bitmap: 40 960 cells (sizeof(LdsCCT<2048>::w) / 4)

// the nine seed positions of point i exactly as the seeding loop forms them (int16 conversions of is_valid included)
template <class F>
__device__ __forceinline__ void for_each_seed(int w, int h, const double* pts, int i, int level, F f) {
    const uint16_t coord_scale = (uint16_t)(1u << level);
    const double lx = rescale_coord(pts[2 * i + 0], 1.0 / coord_scale);
    const double ly = rescale_coord(pts[2 * i + 1], 1.0 / coord_scale);
    const int x = (int)(lx + 0.5), y = (int)(ly + 0.5);
    for (int sdx = -1; sdx <= 1; ++sdx)
        for (int sdy = -1; sdy <= 1; ++sdy) {
            const int sx = (int16_t)(x + sdx), sy = (int16_t)(y + sdy);
            if (sx >= 0 && sx < w && sy >= 0 && sy < h) f(sx, sy);
        }
}

// Cell size and span of the bitmap for the pixel box [x0, x1] x [y0, y1] that the seeds can reach.  cs = -1: the bitmap
// cannot hold it.
__device__ __forceinline__ void win_cells_of_box(WinSel& ws, int x0, int y0, int x1, int y1, int max_words, bool TIGHT) {
    ws.cs = TIGHT ? 4 : 5;
    while (true) {
        const int half = TIGHT ? 1 << (ws.cs - 1) : 0;
        ws.ox = max(x0 - half, 0) >> ws.cs;
        ws.oy = max(y0 - half, 0) >> ws.cs;
        ws.cw = ((x1 + half) >> ws.cs) - ws.ox + 1;
        ws.chh = ((y1 + half) >> ws.cs) - ws.oy + 1;
        if ((ws.cw * ws.chh + 31) / 32 <= max_words) break;
        if (++ws.cs > 15) { ws.cs = -1; break; }
    }
}

// Geometry: cell size and the span of the bitmap.  `box` = 4 words of LDS.  cs = -1: the bitmap cannot hold the frame.
__device__ __forceinline__ WinSel win_geometry(int w, int h, const double* pts, const signed char* lv, int npts, int level,
                                               int max_words, bool TIGHT, uint32_t* box) {
    WinSel ws;
    ws.bits = nullptr;
    ws.openbits = nullptr;
    int x0 = 0, y0 = 0, x1 = w - 1, y1 = h - 1;  // pixels the marked cells can reach
    if (TIGHT) {
        if (threadIdx.x < 4) box[threadIdx.x] = (threadIdx.x & 1) ? 0u : 0xffffffffu;  // min x, max x, min y, max y
        __syncthreads();
        uint32_t mnx = 0xffffffffu, mxx = 0, mny = 0xffffffffu, mxy = 0;
        for (int i = threadIdx.x; i < npts; i += CC_THREADS) {
            if (lv[i] != level + 1) continue;
            for_each_seed(w, h, pts, i, level, [&](int sx, int sy) {
                mnx = min(mnx, (uint32_t)sx); mxx = max(mxx, (uint32_t)sx);
                mny = min(mny, (uint32_t)sy); mxy = max(mxy, (uint32_t)sy);
            });
        }
        if (mnx != 0xffffffffu) {
            atomicMin(&box[0], mnx); atomicMax(&box[1], mxx);
            atomicMin(&box[2], mny); atomicMax(&box[3], mxy);
        }
        __syncthreads();
        const uint32_t b0 = box[0], b1 = box[1], b2 = box[2], b3 = box[3];
        __syncthreads();
        if (b0 == 0xffffffffu) { x0 = y0 = 0; x1 = y1 = 0; }  // nothing to refine: one cell
        else { x0 = (int)b0; x1 = (int)b1; y0 = (int)b2; y1 = (int)b3; }
    }
    win_cells_of_box(ws, x0, y0, x1, y1, max_words, TIGHT);
    return ws;
}

// Marking, with the geometry given: `bits` must hold (cw * chh + 31) / 32 words.  `sink(cell x, cell y)` (cells on
// the frame's grid) is called by the thread that sets a cell's bit first.
// `psub` != NULL: only the points of subset `sub` (psub[i] == sub) mark.  `outside` (one word of LDS, or NULL): bit 4 is set
// when a cell a seed reaches lies outside the span.
template <int LNBITS, class Sink = NoCellSink>
__device__ __forceinline__ void win_mark(WinSel& ws, int w, int h, const double* pts, const signed char* lv, int npts, int level,
                                         uint32_t* bits, uint32_t* openbits, bool TIGHT, Sink sink = Sink(),
                                         const int32_t* psub = nullptr, int sub = 0, int* outside = nullptr) {
    ws.bits = bits;
    ws.openbits = openbits;
    const int nw = (ws.cw * ws.chh + 31) / 32;
    for (int k = threadIdx.x; k < nw; k += CC_THREADS) bits[k] = 0;
    if (openbits)
        for (int k = threadIdx.x; k < LNBITS / 32; k += CC_THREADS) openbits[k] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < npts; i += CC_THREADS) {
        if (lv[i] != level + 1) continue;
        if (psub && psub[i] != sub) continue;
        for_each_seed(w, h, pts, i, level, [&](int sx, int sy) {
            const int half = 1 << (ws.cs - 1);
            const int ax0 = TIGHT ? max(sx - half, 0) >> ws.cs : (sx >> ws.cs) - 1;
            const int ax1 = TIGHT ? (sx + half) >> ws.cs : (sx >> ws.cs) + 1;
            const int ay0 = TIGHT ? max(sy - half, 0) >> ws.cs : (sy >> ws.cs) - 1;
            const int ay1 = TIGHT ? (sy + half) >> ws.cs : (sy >> ws.cs) + 1;
            for (int ay = ay0; ay <= ay1; ++ay)
                for (int ax = ax0; ax <= ax1; ++ax) {
                    const int cx = TIGHT ? ax - ws.ox : ax, cy = TIGHT ? ay - ws.oy : ay;  // (not TIGHT: the span starts at cell (0, 0))
                    if ((unsigned)cx >= (unsigned)ws.cw || (unsigned)cy >= (unsigned)ws.chh) {
                        // a span that was not made from these points (a split level's, list_cells_split): the cell cannot
                        // be listed, the seeds in it would look "not hot" -- the caller gives the frame up instead
                        if (outside) atomicOr(outside, 4);
                        continue;
                    }
                    const int c = cy * ws.cw + cx;
                    const uint32_t bit = 1u << (c & 31);
                    if (!(bits[c >> 5] & bit) && !(atomicOr(&bits[c >> 5], bit) & bit)) sink(ax, ay, i);
                }
        });
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------------
// Several workgroups per frame (round 5).  The refinement of a sparse level is a latency chain whose length grows with the
// frame's hot pixels and points (tools/sparse_phases.py: a 5x5 board's frame takes 53 us, a 10x10's 95, a 14x14's 245), and
// one workgroup per frame leaves the chip 94 % empty.  Points whose marked cells do not touch cannot share a
// super-component (a fill never leaves the 4-connected hot region of its seeds, find_chessboard_corners.cc:356-397, and a
// region that reaches an unmarked cell is "open": the frame is given up), so a frame's points are cut into up to kSubsets
// SUBSETS that are far enough apart, every listed cell carries the subset of the point that marked it, and workgroup
// (frame, s) of the refinement kernel loads the cells, seeds the points and fills the components of subset s alone --
// outputs go to the points' own slots, so the order of the list is untouched.
//   * The cut is made by a workgroup that owns the whole frame (sparse_cells_kernel for the first sparse level, the
//     refinement kernel of the level above while it is not split): a point with no other point within kLinkDist pixels (at
//     the level's coordinates, either axis) can go to any subset, all the others stay together in subset 0.  At 16-pixel
//     cells the cells of a point lie within 24 pixels of it: points >= 64 apart have cells that do not even touch.
//   * A split level hands ITS subsets on to the next level (nobody sees the whole frame any more) after checking that
//     they stay apart: a point's refined position lies inside its marked cells, i.e. within 24 pixels of where it was;
//     each workgroup compares its own refined points with every point of the other subsets (read while those workgroups
//     may still be writing: old or new position) and asks for ONE workgroup at the next level (kFlagSingle) when any pair
//     is closer than kKeepDist -- 48.5 for cells that stay disjoint whatever the other point does, a cell more to be sure.
//   * Cells of more than 16 pixels (a box of more than 40 960 cells), fewer than kSplitMinPoints points, a single
//     cluster: one workgroup, as before.  A subset whose cells hold more hot pixels than the tables gives the frame up
//     (the dense repeat takes it), like every other case the LDS kernel cannot take.
// Header of a level's list, kCellHdr words per frame: [0] cells (-1: given up), [1] log2 cell size, [2..5] span of the
// bitmap, [6] subsets (<= 1: one workgroup), [7] flags.  sparse_cells_kernel zeroes [0], [6], [7] of every level below
// the first; a split level ADDS its cells to [0].
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kSubsets = 4;
constexpr int kHdrSub = 6, kHdrFlags = 7;
constexpr int kFlagGivenUp = 1, kFlagSingle = 2;
constexpr int kSplitMinPoints = 128;  // (a 10x10 board: one workgroup is faster -- more workgroups in flight slow the pixel stream by more than the shorter chain gains; 14x14: -15 %)
constexpr float kLinkDist = 64.f, kKeepDist = 64.f;

struct PartScratch {  // LDS of partition_points
    float x[LPTS], y[LPTS];
    int16_t lab[LPTS];
    int nroots, nlive;
};

// Subsets of the points to refine at `level` (lv[i] == level + 1), from their positions now: psub[i] for every point of the
// frame (0 for the others), returns how many subsets (uniform; 1 = no cut).  All threads of the workgroup.
__device__ __forceinline__ int partition_points(const double* pts, const signed char* lv, int npts, int level, int32_t* psub,
                                                PartScratch& S, int max_sub) {
    const int tid = threadIdx.x;
    if (npts > LPTS || max_sub < 2) {
        for (int i = tid; i < npts; i += CC_THREADS) psub[i] = 0;
        return 1;
    }
    const float inv = 1.0f / (float)(1 << level);
    if (tid == 0) { S.nroots = 0; S.nlive = 0; }
    __syncthreads();
    for (int i = tid; i < npts; i += CC_THREADS) {
        const bool live = lv[i] == level + 1;
        S.x[i] = live ? ((float)pts[2 * i] + 0.5f) * inv : 1e30f;
        S.y[i] = live ? ((float)pts[2 * i + 1] + 0.5f) * inv : 1e30f;
        S.lab[i] = (int16_t)i;
        if (live) atomicAdd(&S.nlive, 1);
    }
    __syncthreads();
    const int nlive = S.nlive;
    if (nlive < kSplitMinPoints) {  // (uniform)
        for (int i = tid; i < npts; i += CC_THREADS) psub[i] = 0;
        return 1;
    }
    // No clustering proper: a point with no other point within kLinkDist is a subset candidate of its own, ALL the others
    // go together (several clusters in one subset are as good as one; label propagation over a board whose points are all
    // linked -- a coarse level -- took 80 us of a 100-us kernel).  One pass over the pairs.
    for (int i = tid; i < npts; i += CC_THREADS) {
        const float xi = S.x[i], yi = S.y[i];
        if (xi > 1e29f) { S.lab[i] = -1; continue; }
        bool linked = false;
        for (int j = 0; j < npts; ++j)
            linked |= j != i && fabsf(S.x[j] - xi) < kLinkDist && fabsf(S.y[j] - yi) < kLinkDist;
        S.lab[i] = linked ? 1 : 0;   // -1 not to be refined, 0 on its own, 1 with the rest
        if (linked) atomicAdd(&S.nroots, 1);
    }
    __syncthreads();
    const int nrest = S.nroots, niso = nlive - nrest;
    int nsub = min(min(kSubsets, max_sub), niso + (nrest > 0 ? 1 : 0));
    if (nsub < 2) nsub = 1;
    // the rest is subset 0; the points on their own are dealt out so that the subsets come out even
    const bool rest_full = nrest * nsub >= nlive;
    for (int i = tid; i < npts; i += CC_THREADS) {
        int sb = 0;
        if (S.lab[i] == 0 && nsub > 1) {
            int r = 0;
            for (int j = 0; j < i; ++j) r += S.lab[j] == 0;
            sb = rest_full ? 1 + r % (nsub - 1) : (r + nrest) % nsub;
        }
        psub[i] = sb;
    }
    __threadfence_block();
    __syncthreads();
    return nsub > 1 ? nsub : 1;
}

// Sparse refinement, step 1: the cells around the points of a frame to refine at `level`, as a list for the kernel
// that computes the response there (chess_cells_kernel): cnt[0] = how many (-1: more than the list or the mask area
// holds, the refinement kernel reports the frame), cnt[1] = their size (log2), cnt[2..5] = the span of the bitmap,
// list = (cell y << 16) | (subset << 12) | cell x.  The refinement kernel marks exactly the listed cells for itself.  All
// threads of the workgroup, which owns the WHOLE frame; `bits` = kWinWords words, `box` = 4 words, `n` = one word of LDS.
// `psub` / `nsub`: the cut of partition_points (nsub <= 1: none); cells of more than 16 pixels are not cut.
__device__ __forceinline__ void list_cells(int w, int h, const double* pts, const signed char* lv, int npts, int level,
                                           uint32_t* bits, uint32_t* box, int* n, uint32_t* list, int list_pitch,
                                           long long max_items, int32_t* cnt, const int32_t* psub = nullptr, int nsub = 1) {
    if (threadIdx.x == 0) *n = 0;
    WinSel ws = win_geometry(w, h, pts, lv, npts, level, kWinWords, true, box);  // (a barrier first: *n is 0 below)
    const bool cut = nsub > 1 && ws.cs == 4;
    auto sink = [&](int ax, int ay, int i) {
        const int k = atomicAdd(n, 1);
        if (k < list_pitch) list[k] = ((uint32_t)ay << 16) | (cut ? (uint32_t)psub[i] << 12 : 0u) | (uint32_t)ax;
    };
    if (ws.cs >= 0) win_mark<2048>(ws, w, h, pts, lv, npts, level, bits, nullptr, true, sink);
    __syncthreads();
    if (threadIdx.x == 0) {
        const int k = *n;
        const bool ok = ws.cs >= 0 && k <= list_pitch && ((long long)k << (2 * (ws.cs - 4))) <= max_items;
        cnt[0] = ok ? k : -1;
        cnt[1] = ws.cs;
        cnt[2] = ws.ox; cnt[3] = ws.oy; cnt[4] = ws.cw; cnt[5] = ws.chh;
        cnt[kHdrSub] = cut ? nsub : 1;
        cnt[kHdrFlags] = ok ? 0 : kFlagGivenUp;
    }
}

// The same by workgroup `sub` of a SPLIT level, for the next level down: the cells of its own points, ADDED to the list the
// workgroups of the frame share (cnt[0], zero before the kernel); the geometry every workgroup must agree on comes from this
// level's span (`hdr`: a refined point lies inside the cells that were marked for it, so twice the span holds every seed of
// the next level); the subsets are kept if they stay apart (see above), else the next level runs as one workgroup.
// `tmp`: LDS, `tmp_cap` words.
__device__ __forceinline__ void list_cells_split(int w, int h, const double* pts, const signed char* lv, int npts, int level,
                                                 uint32_t* bits, int* n, uint32_t* tmp, int tmp_cap, const int32_t* hdr,
                                                 uint32_t* list, int list_pitch, long long max_items, int32_t* cnt,
                                                 const int32_t* psub, int sub, int nsub) {
    const int tid = threadIdx.x;
    if (tid == 0) { n[0] = 0; n[1] = 0; }
    WinSel ws;
    ws.bits = nullptr;
    ws.openbits = nullptr;
    {
        const int pcs = hdr[1], X0 = hdr[2] << pcs, Y0 = hdr[3] << pcs, X1 = ((hdr[2] + hdr[4]) << pcs) - 1, Y1 = ((hdr[3] + hdr[5]) << pcs) - 1;
        win_cells_of_box(ws, max(2 * X0 - 2, 0), max(2 * Y0 - 2, 0), min(2 * X1 + 3, w - 1), min(2 * Y1 + 3, h - 1), kWinWords, true);
    }
    __syncthreads();
    auto sink = [&](int ax, int ay, int) {
        const int k = atomicAdd(&n[0], 1);
        if (k < tmp_cap) tmp[k] = ((uint32_t)ay << 16) | ((uint32_t)sub << 12) | (uint32_t)ax;
    };
    if (ws.cs >= 0) win_mark<2048>(ws, w, h, pts, lv, npts, level, bits, nullptr, true, sink, psub, sub, &n[1]);
    // do the subsets stay apart?  own points as they are now against every point of the others (level + 1 coordinates: the
    // level this kernel has just refined).  n[1] bit 1: a pair closer than kKeepDist -> one workgroup at the next level;
    // bit 2: a pair so close that both can mark the SAME cell (a cell of 2^cs pixels is marked by seeds up to half a cell
    // beyond either edge: seeds less than 2 * 2^cs apart at this level, i.e. points less than 2^cs (+ rounding and the
    // seed ring) apart at level + 1) -- every workgroup keeps a bitmap of its own, so the shared list would then hold the
    // cell twice and the one workgroup of the next level would expand its hot pixels twice: the frame is given up (the
    // dense repeat takes it); bit 4: a seed's cell outside the span the workgroups agreed on (win_mark).
    {
        const float inv = 1.0f / (float)(2 << level);
        const float dup = ws.cs >= 0 ? (float)(1 << min(ws.cs, 15)) + 2.f : 0.f;
        bool close = false, twice = false;
        for (int i = tid; i < npts; i += CC_THREADS) {
            if (psub[i] != sub || lv[i] != level + 1) continue;
            const float xi = ((float)pts[2 * i] + 0.5f) * inv, yi = ((float)pts[2 * i + 1] + 0.5f) * inv;
            for (int j = 0; j < npts; ++j)
                if (psub[j] != sub && lv[j] <= level + 2) {  // (lv > level + 2: never refined again)
                    const float dx = fabsf(((float)pts[2 * j] + 0.5f) * inv - xi), dy = fabsf(((float)pts[2 * j + 1] + 0.5f) * inv - yi);
                    close |= dx < kKeepDist && dy < kKeepDist;
                    twice |= dx < dup && dy < dup;
                }
        }
        if (close || twice) atomicOr(&n[1], (close ? 1 : 0) | (twice ? 2 : 0));
    }
    __syncthreads();
    const int k = n[0];
    __syncthreads();
    if (tid == 0) {
        int flags = (ws.cs != 4 || (n[1] & 1)) ? kFlagSingle : 0;
        int base = 0;
        if (ws.cs < 0 || k > tmp_cap || (n[1] & 6)) {
            flags |= kFlagGivenUp;
        } else {
            base = atomicAdd(&cnt[0], k);
            if (base + k > list_pitch || ((long long)(base + k) << (2 * (ws.cs - 4))) > max_items) flags |= kFlagGivenUp;
        }
        if (flags) atomicOr(&cnt[kHdrFlags], flags);
        cnt[1] = ws.cs;  // (the same values from every workgroup of the frame)
        cnt[2] = ws.ox; cnt[3] = ws.oy; cnt[4] = ws.cw; cnt[5] = ws.chh;
        cnt[kHdrSub] = nsub;
        n[0] = (flags & kFlagGivenUp) ? -1 : base;
    }
    __syncthreads();
    const int base = n[0];
    if (base >= 0)
        for (int q = tid; q < k; q += CC_THREADS) list[base + q] = tmp[q];
}

// One workgroup per frame: the cells of the first level below the start level (its points come from the detection;
// below that the refinement kernel of a level lists the cells of the next one itself), the cut of its points into
// subsets, and the headers of the levels below it zeroed (cnt_all: [level][frame][kCellHdr]).
__global__ __launch_bounds__(CC_THREADS) void sparse_cells_kernel(int w, int h, int level, RefineIO io, uint32_t* cell_list,
                                                               int32_t* cell_cnt, int list_pitch, long long max_items,
                                                               int frame0, int32_t* cnt_all, int nframes_all) {
    __shared__ uint32_t bits[kWinWords];
    __shared__ uint32_t box[4];
    __shared__ int n;
    __shared__ PartScratch part;
    const int frame = frame0 + blockIdx.x;
    const long long pb = (long long)frame * io.pitch;
    if (cnt_all && threadIdx.x < level) {
        int32_t* hd = cnt_all + ((size_t)threadIdx.x * nframes_all + frame) * kCellHdr;
        hd[0] = 0; hd[kHdrSub] = 0; hd[kHdrFlags] = 0;
    }
    const int npts = min(io.npoints[frame], io.pitch);
    const int nsub = partition_points(io.points + 2 * pb, io.levels + pb, npts, level, io.leader + pb, part, io.subsets);
    list_cells(w, h, io.points + 2 * pb, io.levels + pb, npts, level, bits, box, &n,
               cell_list + (long long)frame * list_pitch, list_pitch, max_items, cell_cnt + kCellHdr * frame, io.leader + pb, nsub);
}

// Sparse refinement, step 3a: the hot pixels of a frame out of the masks chess_cells_kernel left (32 bytes per 16 x 16
// micro-tile, byte 2 * row + half = the 8 pixels x .. x + 7).  One workgroup per frame, no counter shared with
// anybody.  The entries go straight into the LDS list (`lds_xy`, `lds_cap` entries: the frame is then loaded, see
// lds_load_and_label's `preloaded`); only if there are more -- a frame that needs bands -- a second pass writes the
// global list the band planner and the loader read, like a dense level's.  Returns the number of hot pixels
// (uniform), -1 when the frame was given up by whoever listed the cells.  `cnt` = one word of LDS.
// `only` >= 0: the cells of that subset alone (a split level).
template <class Put>
__device__ __forceinline__ void expand_masks(const uint32_t* masks, const uint32_t* list, int nwords, int cs, int* cnt, Put put,
                                             int only = -1) {
    const int sub = cs - 4;
    constexpr int U = 8;  // (a clean 10x10 board at 16-pixel cells: ~3800 words, two rounds of 256 x 8)
    for (int k0 = threadIdx.x; k0 < nwords; k0 += CC_THREADS * U) {
        uint32_t m[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + u * CC_THREADS;
            m[u] = (k < nwords && (only < 0 || cell_sub(list[(k >> 3) >> (2 * sub)]) == only)) ? masks[k] : 0u;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!m[u]) continue;
            const int k = k0 + u * CC_THREADS, it = k >> 3, j = k & 7;
            const uint32_t c = list[it >> (2 * sub)];
            const int si = it & ((1 << (2 * sub)) - 1);
            const int xt = (cell_x(c) << cs) + 16 * (si & ((1 << sub) - 1));
            const int yt = (cell_y(c) << cs) + 16 * (si >> sub);
            int slot = atomicAdd(cnt, __popc(m[u]));
            uint32_t mm = m[u];
            while (mm) {
                const int b = __ffs(mm) - 1;  // byte q = b >> 3: row 2j + (q >> 1), half q & 1; pixel b & 7 of its group
                mm &= mm - 1;
                const int q = b >> 3;
                put(slot++, ((uint32_t)(yt + 2 * j + (q >> 1)) << 16) | (uint32_t)(xt + 8 * (q & 1) + (b & 7)));
            }
        }
    }
}
// the cell bitmap of a frame straight from its list: what is marked IS what was computed
__device__ __forceinline__ void mark_listed_cells(const WinSel& ws, const uint32_t* list, int ncell, uint32_t* bits,
                                                  uint32_t* openbits, int nopen_words, int only = -1) {
    const int nw = (ws.cw * ws.chh + 31) / 32;
    for (int k = threadIdx.x; k < nw; k += CC_THREADS) bits[k] = 0;
    for (int k = threadIdx.x; k < nopen_words; k += CC_THREADS) openbits[k] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < ncell; k += CC_THREADS) {
        const uint32_t c = list[k];
        if (only >= 0 && cell_sub(c) != only) continue;  // (another subset's cell: unmarked here, i.e. "not computed")
        const int cx = cell_x(c) - ws.ox, cy = cell_y(c) - ws.oy;
        if ((unsigned)cx < (unsigned)ws.cw && (unsigned)cy < (unsigned)ws.chh) {
            const int b = cy * ws.cw + cx;
            atomicOr(&bits[b >> 5], 1u << (b & 31));
        }
    }
    __syncthreads();
}
// -> number of hot pixels (uniform; -1: the frame was given up), `ws` = the selection (cells marked in `bits`)
// `only` >= 0 (a split level): the cells of that subset; `hot_xy` / `cap` are then the subset's part of the frame's global list.
__device__ __forceinline__ int hot_list_from_masks(const RefineIO& io, const CompTables& t, int frame, uint32_t* hot_xy, int cap, int* cnt,
                                                   uint32_t* lds_xy, int lds_cap, WinSel& ws, uint32_t* bits, uint32_t* openbits,
                                                   int nopen_words, int only = -1) {
    const int32_t* hdr = io.cell_cnt + kCellHdr * frame;
    const int ncell = (hdr[kHdrFlags] & kFlagGivenUp) ? -1 : hdr[0];
    ws.cs = hdr[1]; ws.ox = hdr[2]; ws.oy = hdr[3]; ws.cw = hdr[4]; ws.chh = hdr[5];
    ws.bits = bits;
    ws.openbits = openbits;
    if (threadIdx.x == 0) *cnt = 0;
    __syncthreads();
    if (ncell < 0 || ws.cs < 4 || (ws.cw * ws.chh + 31) / 32 > kWinWords) return -1;
    const int nwords = (ncell << (2 * (ws.cs - 4))) * 8;
    const uint32_t* list = io.cell_list + (long long)frame * io.list_pitch;
    const uint32_t* masks = reinterpret_cast<const uint32_t*>(t.gidx + (long long)frame * t.gidx_pitch);
    mark_listed_cells(ws, list, ncell, bits, openbits, nopen_words, only);
    expand_masks(masks, list, nwords, ws.cs, cnt, [&](int slot, uint32_t e) { if (slot < lds_cap) lds_xy[slot] = e; }, only);
    __syncthreads();
    const int n = *cnt;
    __syncthreads();
    if (n <= lds_cap) return n;
    if (threadIdx.x == 0) *cnt = 0;
    __syncthreads();
    expand_masks(masks, list, nwords, ws.cs, cnt, [&](int slot, uint32_t e) { if (slot < cap) hot_xy[slot] = e; }, only);
    __threadfence();  // the list is read back by other waves of this workgroup
    __syncthreads();
    return n;
}

// What a declining kernel leaves behind: the frame stays with the global-memory kernels (path 0).  A kernel
// that declines after its first band has already appended candidates (detect: scratch the fallback
// overwrites) or refined the points of the bands it finished (refine: their level is updated, so the
// fallback skips them, and their components are disjoint from what is left -- same result).
// Sparse refinement (lds_path bit 1024): there is no global-memory kernel to leave the frame to (the dense response
// only holds the cells around the points): the frame is reported instead (kStatusSparse -> the caller repeats the
// call without the option).
constexpr int kLdsSparse = kLdsPathSparse;
// (`next_hdr`: the header of the next level's cell list, or NULL -- nobody lists that level's cells now)
__device__ __forceinline__ void lds_decline(const CompTables& t, int frame, int32_t* next_hdr = nullptr) {
    if (threadIdx.x != 0) return;
    if (t.lds_path & kLdsSparse) {
        t.path[frame] = 1;
        atomicOr(t.status + frame, kStatusSparse);  // (device scope: a frame may have several workgroups)
        if (next_hdr) atomicOr(next_hdr + kHdrFlags, kFlagGivenUp);
    } else {
        t.path[frame] = 0;
    }
}
// refine after the first band: path 2 = "the global-memory kernel finishes the frame and ADDS to nrefined"
__device__ __forceinline__ void lds_decline_refine(const CompTables& t, int frame, int band, const RefineIO& io, int nref) {
    if (threadIdx.x != 0) return;
    if (t.lds_path & kLdsSparse) {
        t.path[frame] = 1;
        atomicOr(t.status + frame, kStatusSparse);
        if (io.next_cnt) atomicOr(io.next_cnt + kCellHdr * frame + kHdrFlags, kFlagGivenUp);
        return;
    }
    t.path[frame] = band > 0 ? 2 : 0;
    if (band > 0 && io.nrefined) io.nrefined[frame] = nref;
}

// (launch bounds: at most 128 VGPRs, so that a wave of these kernels fits into what ONE retiring wave of the pixel
// kernels frees on a SIMD -- at 129 VGPRs the refine kernel waited for two, 80 -> 270 us per launch)
template <int N>
__device__ __forceinline__ void cc_detect_lds_frame(const LevelBatch& lb, const CompTables& t, int level, const DetectOut& out,
                                                    int frame) {
    using LdsCC = LdsCCT<N>;
    constexpr int LROOTS = LdsCC::LROOTS, LEPT = LdsCC::LEPT;
    constexpr int LSTKD = LdsCC::LSTK - LdsCC::LN;  // LIFO words of the fills: the neighbour table takes the first LN
    extern __shared__ __attribute__((aligned(16))) char lds_cc_raw[];
    LdsCC& L = *reinterpret_cast<LdsCC*>(lds_cc_raw);
    if (!MRG_EXP(t.lds_path & 16)) __builtin_amdgcn_s_setprio(3);
    const int tid = threadIdx.x;
    const int nraw = t.hot_cnt[frame];
    FrameView v = make_view(lb, t, frame);
    const int w = v.w, h = v.h;
    int nbands = 0;  // cc_lds bit 256: no banding (test hook)
    if (nraw <= t.cap && !(nraw > LdsCC::LN && (t.lds_path & 256))) nbands = lds_plan_bands(L, v, nraw);
    if (nbands == 0) {
        lds_decline(t, frame);
        return;
    }
    if (tid == 0) L.ncand = 0;
    // seeds live in [8, w-8) x [8, h-8) (:332-333)
    auto seedable = [&](uint32_t e) {
        const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
        return x > kMargin && x < w - kMargin - 1 && y > kMargin && y < h - kMargin - 1;
    };
    for (int band = 0; band < nbands; ++band) {
        int n;
        if (!lds_load_and_label(L, v, nraw, t.cap, nbands > 1, L.band_y[band], L.band_y[band + 1], n)) {
            lds_decline(t, frame);
            return;
        }
        if (MRG_EXP(t.lds_path & 8)) { if (tid == 0) { t.path[frame] = 1; out.counts[frame] = 0; } return; }  // ablation (timing only)
        // roots with >= 2 pixels (a single hot pixel can only give a one-pixel blob, :205); each gets a LIFO of
        // (sum of hot-neighbour counts + 1) words, which bounds the pushes of all fills of it together
        for (int i = tid; i < n; i += CC_THREADS) {
            if (L.xy[i] == kHotDead || L.lab[i] != i) continue;
            const int a = L.u.acc[i], cnt = a & 0x1fff, need = (a >> 13) + 1;
            if (cnt < kBlobMinPixels) continue;
            const int r = atomicAdd(&L.nroots, 1);
            if (need > LSTKD) L.total = 1;  // one super-component alone wants more LIFO than there is
            if (r < LROOTS) { L.w.r.root[r] = (int16_t)i; L.w.r.cnt[r] = (int16_t)cnt; L.w.r.soff[r] = (int16_t)min(need, LSTKD); }
        }
        __syncthreads();
        if (L.nroots > LROOTS || L.total) {  // does not fit
            lds_decline(t, frame);
            return;
        }
        const int nroots = L.nroots;
        // smallest raster position of every super-component: the first seed the raster scan meets
        for (int i = tid; i < n; i += CC_THREADS) L.u.acc[i] = 0x7fffffff;
        __syncthreads();
        for (int i = tid; i < n; i += CC_THREADS)
            if (L.xy[i] != kHotDead) atomicMin(&L.u.acc[L.lab[i]], (int)L.xy[i]);
        __syncthreads();
        // ... as a list index: the fills below have no hash map any more
        for (int r = tid; r < nroots; r += CC_THREADS)
            L.w.r.fidx[r] = (int16_t)lds_find(L, (uint32_t)L.u.acc[L.w.r.root[r]]);
        __syncthreads();
        // member lists (list indices of the pixels of a super-component, unordered): what the "raster scan goes
        // on" step below walks instead of the whole hot list.  They take over the storage of the labels.
        int mylab[LEPT];
#pragma unroll
        for (int k = 0; k < LEPT; ++k) {
            const int i = tid + CC_THREADS * k;
            mylab[k] = (i < n && L.xy[i] != kHotDead) ? (int)L.lab[i] : -1;
        }
        for (int i = tid; i < n; i += CC_THREADS) L.u.acc[i] = -1;
        __syncthreads();
        for (int r = tid; r < nroots; r += CC_THREADS) {
            const int mo = atomicAdd(&L.mtop, (int)L.w.r.cnt[r]);
            L.u.acc[L.w.r.root[r]] = mo;      // running write position of this list
            L.w.r.root[r] = (int16_t)mo;      // the root's list index is not needed any more
        }
        __syncthreads();
        int16_t* members = L.lab;
#pragma unroll
        for (int k = 0; k < LEPT; ++k) {
            if (mylab[k] < 0 || L.u.acc[mylab[k]] < 0) continue;  // (acc only grows: a list's slot stays >= 0)
            members[atomicAdd(&L.u.acc[mylab[k]], 1)] = (int16_t)(tid + CC_THREADS * k);
        }
        __syncthreads();  // the accumulators are dead: their storage becomes the LIFOs
        // neighbour table of the fills (drain_nb): low words over the hash map, high halves in the first LN words of
        // the LIFO space, the LIFOs behind them
        uint16_t* nhi = reinterpret_cast<uint16_t*>(L.u.stk);
        lds_build_neighbours(L, v, n, nhi);
        int16_t* lifo = L.u.stk + LdsCC::LN;

        // The fills of a band share LSTKD LIFO words.  When the super-components together want more (a 14x14 board:
        // ~150 of them per band at ~50 words each), they run in rounds: every pending root asks for its words, the
        // ones that still fit run, the others wait for the next round (the first to ask always fits).
        static_assert(LROOTS <= 2 * CC_THREADS, "a thread owns at most two roots");
        bool pending[2] = {tid < nroots, tid + CC_THREADS < nroots};
        while (true) {
        if (tid == 0) { L.top = 0; L.changed = 0; }
        __syncthreads();
        for (int rr = 0; rr < 2; ++rr) {
            if (!pending[rr]) continue;
            const int r = tid + CC_THREADS * rr;
            const int need = L.w.r.soff[r];
            const int so = atomicAdd(&L.top, need);
            if (so + need > LSTKD) { L.changed = 1; continue; }
            pending[rr] = false;
            const int cnt = L.w.r.cnt[r], mo = L.w.r.root[r];
            int left = cnt;
            int16_t* stk = lifo + so;
            int si = L.w.r.fidx[r];
            uint32_t seed = L.xy[si];
            bool have = seedable(seed);
            while (true) {
                if (!have) {
                    // the raster scan goes on: the smallest seedable position among what is left of this
                    // super-component (pixels below the running-maximum threshold are consumed but not
                    // expanded, so the fringe of a blob is often left over)
                    uint32_t best = kHotDead;
                    for (int q = 0; q < cnt; ++q) {
                        const int i = members[mo + q];
                        const uint32_t e = L.xy[i];
                        if (L.val[i] > 0 && seedable(e) && e < best) { best = e; si = i; }
                    }
                    if (best == kHotDead) break;
                    seed = best;
                }
                have = false;
                stk[0] = (int16_t)si;  // :338
                Blob b;
                if (MRG_EXP(t.lds_path & 4)) { b.touched = true; left = 0; }  // ablation (timing only)
                else left -= drain_nb(L, L.hashw, nhi, w, h, stk, 1, b);
                if (blob_passes_cheap_tests(b) &&
                    (MRG_EXP(t.lds_path & 2) || window_variance_high(v.img, v.img_stride, w, h, b.xpk, b.ypk))) {  // :207
                    const int c = atomicAdd(&L.ncand, 1);
                    if (c < v.cand_cap) {
                        Cand cd;
                        cd.sum_rx = b.srx; cd.sum_ry = b.sry; cd.sum_r = b.sr;
                        cd.seed = (int32_t)seed;
                        cd.x_peak = (uint16_t)b.xpk; cd.y_peak = (uint16_t)b.ypk;
                        cd.ok = 1; cd.pad = 0;
                        v.cand[c] = cd;
                    }
                }
                if (left <= 0) break;
            }
        }
        __syncthreads();
        if (!L.changed) break;
        __syncthreads();  // everybody has read the flag before it is reset
        }
    }
    // order by seed position = the reference's output order (:332-353), sorted in LDS.  The tables are dead:
    // the keys take the whole allocation (one band: at most LN / 2 candidates; several: whatever they gave)
    const int nvalid = L.ncand;
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds_cc_raw);
    constexpr int kMaxKeys = 4096;
    static_assert(kMaxKeys * 8 <= (int)offsetof(LdsCC, nroots), "the sort keys must not reach the counters");
    if (nvalid > kMaxKeys || nvalid > v.cand_cap) {
        lds_decline(t, frame);
        return;
    }
    __syncthreads();  // every thread has read L.ncand before the keys overwrite the tables
    if (tid == 0) t.path[frame] = 1;
    int n_pad = 1;
    while (n_pad < nvalid) n_pad <<= 1;
    for (int c = tid; c < n_pad; c += CC_THREADS)
        keys[c] = c < nvalid ? (((unsigned long long)(uint32_t)v.cand[c].seed << 32) | (uint32_t)c) : ~0ull;
    __syncthreads();
    bitonic_sort(keys, n_pad);
    emit_detect_outputs(v, keys, nvalid, level, out, frame);
}
template <int N>
__global__ __launch_bounds__(CC_THREADS, 4) void cc_detect_lds_kernel(LevelBatch lb, CompTables t, int level,
                                                                   DetectOut out, int frame0) {
    cc_detect_lds_frame<N>(lb, t, level, out, frame0 + blockIdx.x);
}
template <int N>  // several levels in one grid: see cc_detect_levels_kernel
__global__ __launch_bounds__(CC_THREADS, 4) void cc_detect_lds_levels_kernel(DetectLevels a) {
    const int k = blockIdx.y;
    cc_detect_lds_frame<N>(a.lb[k], a.t[k], a.level[k], a.out[k], blockIdx.x);
}

// SPARSE: the instantiation behind a level of a sparse chain (kLdsPathSparse).  Two kernels, so that what the sparse
// schedule adds (mask expansion, the next level's cell list) costs the dense one neither registers nor spills.
template <int N, bool SPARSE>
__global__ __launch_bounds__(CC_THREADS, 4) void cc_refine_lds_kernel(LevelBatch lb, CompTables t, int level,
                                                                   RefineIO io, int frame0) {
    using LdsCC = LdsCCT<N>;
    constexpr int LN = LdsCC::LN, LSTK = LdsCC::LSTK;
    extern __shared__ __attribute__((aligned(16))) char lds_cc_raw[];
    LdsCC& L = *reinterpret_cast<LdsCC*>(lds_cc_raw);
    if (!MRG_EXP(t.lds_path & 16)) __builtin_amdgcn_s_setprio(3);
    const int frame = frame0 + blockIdx.x, tid = threadIdx.x;
    // phase clock (cc_lds bit 512, mrgingham_amd_debug_refine_clock, tools/cc_phases.py): thread 0 of the first
    // frame leaves 100 MHz ticks of the phase boundaries of its first band in the scratch of the global-memory kernel
    const bool clk = MRG_EXP(t.lds_path & 512) && blockIdx.x == 0 && tid == 0;
    long long* tk = reinterpret_cast<long long*>(io.sroot);  // (scratch of the global-memory kernel, unused here)
    auto tick = [&](int k) { if (clk) tk[k] = wall_clock64(); };
    tick(0);
    constexpr bool sparse = SPARSE;  // the response exists in the cells around the points only
    const int npts = min(io.npoints[frame], io.pitch);
    FrameView v = make_view(lb, t, frame);
    // several workgroups per frame ("Several workgroups" above): this one is subset `sub` of `nsub`
    int nsub = 1;
    const int sub = sparse ? (int)blockIdx.y : 0;
    if (sparse) {
        const int32_t* hdr = io.cell_cnt + kCellHdr * frame;
        nsub = hdr[kHdrSub];
        if (nsub < 1 || (hdr[kHdrFlags] & kFlagSingle)) nsub = 1;
        if (sub >= nsub) return;  // (nothing of the frame is this workgroup's: no header, no status is touched)
    }
    const bool split = nsub > 1;
    const int32_t* psub = io.leader + (long long)frame * io.pitch;  // (scratch of the global-memory kernel: here the points' subsets)
    int32_t* const next_hdr = (sparse && io.next_cnt) ? io.next_cnt + kCellHdr * frame : nullptr;
    // its own part of the frame's LIFO arena (lds_load_and_label parks there) and of its global hot list (a subset with more
    // hot pixels than the LDS list holds goes band by band over it, like a whole frame does)
    const int cap = split ? t.cap / kSubsets : t.cap;
    if (split) {
        v.arena += (long long)sub * (2 * LN);
        v.hot_xy += (long long)sub * cap;
    }
    // The cell bitmap lives in L.w (dead until the LIFO demands are written), the open flags behind the
    // accumulators in L.u (dead until the fills).
    WinSel ws;
    ws.cs = -1;
    uint32_t* const wbits = reinterpret_cast<uint32_t*>(&L.w);
    uint32_t* const obits = reinterpret_cast<uint32_t*>(L.u.stk) + LN;
    static_assert(sizeof(L.u) >= (size_t)LN * 4 + (size_t)LN / 8, "open flags behind the accumulators");
    static_assert(sizeof(L.w) / 4 == kWinWords, "list_cells sizes the bitmap for kWinWords");
    const int nraw = sparse ? hot_list_from_masks(io, t, frame, v.hot_xy, cap, &L.nload, L.xy, LN, ws, wbits, obits, LN / 32, split ? sub : -1)
                            : t.hot_cnt[frame];
    const bool preloaded = sparse && nraw <= LN;
    if (npts > LPTS || nraw < 0) {  // the LDS kernel does not take that many points (sparse: nor that many cells)
        lds_decline(t, frame, next_hdr);
        return;
    }
    const int w = v.w, h = v.h;
    const long long pb = (long long)frame * io.pitch;
    double* pts = io.points + 2 * pb;
    signed char* lv = io.levels + pb;
    int nbands = 0;  // cc_lds bit 256: no banding, no windows (test hook)
    const bool may_select = nraw <= cap && !(nraw > LN && (t.lds_path & 256));
    // More hot pixels than the tables hold: first try to load only the cells around the points (one pass over the
    // list; a textured scene has 10^4 - 10^5 hot pixels of which the refinement needs ~10^3), then bands.
    // (sparse refinement: the selection is what was computed, marked above, and every listed pixel is in it)
    bool windowed = sparse && nraw <= LN;
    // a list only a little longer than the tables is a large board on a flat background (14x14: 2600): every hot
    // pixel is near a point, the cells would hold them all -- bands first there, cells only if no band cut exists
    const bool bands_first = !sparse && nraw <= LN + LN / 2;
    auto try_windows = [&]() {
        ws = win_geometry(w, h, pts, lv, npts, level, kWinWords, false, L.edge);
        if (ws.cs < 0) return;
        win_mark<LN>(ws, w, h, pts, lv, npts, level, wbits, obits, false);
        // do the marked cells hold few enough hot pixels?  (one more pass over the list)
        if (tid == 0) L.nload = 0;
        __syncthreads();
        int cnt = 0;
        scan_hot_list(v.hot_xy, nraw, [&](uint32_t e) { cnt += ws.template marked<false>((int)(e & 0xffffu), (int)(e >> 16)); });
        if (cnt) atomicAdd(&L.nload, cnt);
        __syncthreads();
        windowed = L.nload <= LN;
        __syncthreads();
    };
    if (may_select && !sparse && nraw > LN && !bands_first) try_windows();
    if (!windowed && may_select && !sparse) nbands = lds_plan_bands(L, v, nraw);
    if (!windowed && nbands == 0 && may_select && nraw > LN && bands_first) try_windows();
    // sparse refinement, the cells hold more hot pixels than the tables (a 14x14 board at level 1: 3000): bands of the
    // list -- everything in it is in a marked cell --, the cells marked again before every band (the bitmap shares its
    // LDS with the LIFO demands of the band before).  A band boundary may cross hot pixels that are not in the list;
    // those are in unmarked cells, which is exactly what the open check looks for.
    bool win_bands = false;
    if (sparse && !windowed && may_select && ws.cs >= 0) {
        nbands = lds_plan_bands(L, v, nraw);
        win_bands = windowed = nbands > 0;
    } else if (windowed) {
        if (tid == 0) { L.band_y[0] = 0; L.band_y[1] = 0; L.shear = 0; }
        nbands = 1;
        __syncthreads();
    }
    if (nbands == 0) {
        lds_decline(t, frame, next_hdr);
        return;
    }
    if (tid == 0) L.nref = 0;
    tick(1);
    uint32_t* seeds = io.seeds + 9 * pb;  // here: list indices, read back by the group's leader lane
    int32_t* nseeds = io.nseeds + pb;
    int32_t* gneed = io.need + pb;  // per leader
    const uint16_t coord_scale = (uint16_t)(1u << level);
    // the group leader of every point, -1 for a point that is not refinable at this level: behind
    // need16[] in the same union (LN * 2 bytes used of 2.5 LN), npts <= LPTS entries
    int16_t* lead16 = L.w.need16 + LN;
    static_assert(sizeof(L.w) >= (size_t)LN * 2 + (size_t)LPTS * 2, "lead16 must fit behind need16");

    for (int band = 0; band < nbands; ++band) {
        int n;
        if (win_bands) {
            __syncthreads();
            mark_listed_cells(ws, io.cell_list + (long long)frame * io.list_pitch, io.cell_cnt[kCellHdr * frame], wbits, obits, LN / 32,
                              split ? sub : -1);
        }
        if (!lds_load_and_label<SPARSE>(L, v, nraw, cap, nbands > 1, L.band_y[band], L.band_y[band + 1], n,
                                        windowed ? &ws : nullptr, preloaded)) {
            // (window mode: the cells around the points hold more hot pixels than the tables do -- band 0, plain decline)
            lds_decline_refine(t, frame, band, io, L.nref);
            return;
        }
        if (MRG_EXP(t.lds_path & 8)) { if (tid == 0) t.path[frame] = 1; return; }  // ablation (timing only)
        if (band == 0) tick(2);
        // LIFO demand of every super-component at its root, then the accumulators become the claim table
        for (int i = tid; i < n; i += CC_THREADS) L.w.need16[i] = (int16_t)((L.u.acc[i] >> 13) + 1);
        __syncthreads();
        int32_t* claim = L.u.acc;
        for (int i = tid; i < n; i += CC_THREADS) claim[i] = 0x7fffffff;

        // R1: seeds of every refinable point (:362-382), in the reference's push order.  A thread owns points
        // tid and tid + 256 and keeps their seed roots, seed counts and leaders in registers.  (With bands: a
        // point finds its seeds in exactly one band -- the hash only holds this band's pixels -- and is not
        // refinable any more once a band has refined it.)
        int ns_[LPPT], lead_[LPPT];
        short sroot_[LPPT][9];
#pragma unroll
        for (int q = 0; q < LPPT; ++q) {
            const int i = tid + CC_THREADS * q;
            int ns = -1;  // -1: not refinable at this level (or no such point)
            if (i < npts && lv[i] == level + 1 && (!split || psub[i] == sub)) {
                ns = 0;
                const double lx = rescale_coord(pts[2 * i + 0], 1.0 / coord_scale);  // :369
                const double ly = rescale_coord(pts[2 * i + 1], 1.0 / coord_scale);
                const int x = (int)(lx + 0.5), y = (int)(ly + 0.5);  // :371-372
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
                    for (int dy = -1; dy <= 1; ++dy) {
                        const int sx = (int16_t)(x + dx), sy = (int16_t)(y + dy);  // is_valid takes int16_t
                        int j = -1;
                        if (sx >= 0 && sx < w && sy >= 0 && sy < h)
                            j = lds_find(L, ((uint32_t)sy << 16) | (uint32_t)sx);  // hot <=> listed (nothing consumed yet)
                        if (j >= 0) {
                            seeds[9 * i + ns] = (uint32_t)j;
                            const int r = L.lab[j];
#pragma unroll
                            for (int k = 0; k < 9; ++k)  // static register index
                                if (k == ns) sroot_[q][k] = (short)r;
                            ++ns;
                            // window mode: a seed in a super-component that reaches the edge of what was loaded
                            if (windowed && ((ws.openbits[r >> 5] >> (r & 31)) & 1u)) L.leak = 1;
                        }
                    }
                nseeds[i] = ns;
            }
            ns_[q] = ns;
            lead_[q] = i;
        }
        __syncthreads();
        if (windowed && L.leak) {  // (uniform) nothing has been refined yet: the global-memory kernel takes the frame
            lds_decline(t, frame, next_hdr);
            return;
        }
        if (band == 0) tick(3);

        // R2: points whose seeds share a super-component are replayed in index order by one lane:
        // propagate the minimum point index over the bipartite graph points <-> super-components
        while (true) {
            bool changed = false;
#pragma unroll
            for (int q = 0; q < LPPT; ++q) {
                if (ns_[q] <= 0) continue;
                int m = lead_[q];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (k < ns_[q])
                        m = min(m, __hip_atomic_load(&claim[sroot_[q][k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                changed |= m < lead_[q];
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    if (k < ns_[q] && atomicMin(&claim[sroot_[q][k]], m) > m) changed = true;
                lead_[q] = m;
            }
            if (changed) L.changed = 1;
            __syncthreads();
            const int c = L.changed;
            __syncthreads();
            if (!c) break;
            if (tid == 0) L.changed = 0;
            __syncthreads();
        }
#pragma unroll
        for (int q = 0; q < LPPT; ++q) {
            const int i = tid + CC_THREADS * q;
            if (i < npts) lead16[i] = (int16_t)(ns_[q] < 0 ? -1 : lead_[q]);
        }

        if (band == 0) tick(4);
        // R3: LIFO demand of each group = sum over its super-components, each counted once; groups take
        // their LIFOs in the order of a running counter
#pragma unroll
        for (int q = 0; q < LPPT; ++q) {
            const int i = tid + CC_THREADS * q;
            if (i < npts && (!split || psub[i] == sub)) gneed[i] = 0;  // (a group's leader is one of the subset's own points)
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LPPT; ++q) {
            if (ns_[q] <= 0) continue;  // nothing hot around the point (in this band): :383-384, no fill
            const int i = tid + CC_THREADS * q, ld = lead_[q];
            int add = ld == i ? 10 : 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (k >= ns_[q]) continue;
                const int root = sroot_[q][k];
                if (atomicCAS(&claim[root], ld, ld | 0x40000000) == ld) add += (int)L.w.need16[root];
            }
            // bits 20..: members of the group (so that its lane knows when it has seen the last one)
            wg_add(gneed + ld, add + (1 << 20));
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LPPT; ++q) {
            const int i = tid + CC_THREADS * q;
            if (ns_[q] > 0 && lead_[q] == i) atomicMax(&L.total, aload(gneed + i) & 0xfffff);
        }
        __syncthreads();
        if (L.total > LSTK) {  // one group alone wants more LIFO than there is
            lds_decline_refine(t, frame, band, io, L.nref);
            return;
        }
        __syncthreads();  // the claim table is dead: its storage becomes the LIFOs
        // (here rather than behind R1, where labels and hash map die: the seed roots of R1-R3 are out of the
        // registers by now)
        uint16_t* nhi = reinterpret_cast<uint16_t*>(L.lab);  // (the labels are dead as well)
        lds_build_neighbours(L, v, n, nhi);

        if (band == 0) tick(5);
        // R4: one lane per group, members in index order (:358); accepted points are written in place.  The
        // groups share LSTK LIFO words and run in rounds when together they want more (see the detect kernel).
        bool pending[LPPT];
#pragma unroll
        for (int q = 0; q < LPPT; ++q) pending[q] = ns_[q] > 0 && lead_[q] == tid + CC_THREADS * q;
        while (true) {
        if (tid == 0) { L.top = 0; L.changed = 0; }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < LPPT; ++q) {
            const int i = tid + CC_THREADS * q;
            if (!pending[q]) continue;
            const int gn = aload(gneed + i), need = gn & 0xfffff;
            const int so = atomicAdd(&L.top, need);
            if (so + need > LSTK) { L.changed = 1; continue; }
            pending[q] = false;
            int16_t* stk = L.u.stk + so;
            int left = gn >> 20;  // members not met yet: most groups are one point, and the walk ends at once
            for (int j = i; left > 0; ++j) {
                if (lead16[j] != i) continue;
                --left;
                const int ns = j == i ? ns_[q] : aload(nseeds + j);
                for (int k = 0; k < ns; ++k) stk[k] = (int16_t)__hip_atomic_load(&seeds[9 * j + k], MRG_WG);
                Blob b;
                if (MRG_EXP(t.lds_path & 4)) continue;  // ablation (timing only)
                drain_nb(L, L.hashw, nhi, w, h, stk, ns, b);
                if (!blob_passes_cheap_tests(b)) continue;
                if (!MRG_EXP(t.lds_path & 2) && !window_variance_high(v.img, v.img_stride, w, h, b.xpk, b.ypk)) continue;  // :207
                const double cx = (double)b.srx / (double)b.sr;  // :262-263
                const double cy = (double)b.sry / (double)b.sr;
                pts[2 * j + 0] = rescale_coord(cx, (double)coord_scale);  // :390
                pts[2 * j + 1] = rescale_coord(cy, (double)coord_scale);
                lv[j] = (signed char)level;  // :393
                atomicAdd(&L.nref, 1);
            }
        }
        __syncthreads();
        if (!L.changed) break;
        __syncthreads();  // everybody has read the flag before it is reset
        }
        if (band == 0) tick(6);
        __threadfence_block();
        __syncthreads();  // the next band reads the levels this one wrote
        // Several bands: what this band's fills consumed goes back into the dense response, like the
        // global-memory kernel leaves it.  If a later band has to give the frame up, that kernel finishes
        // it, and a point of THIS band that was rejected because an earlier point had consumed its
        // component must find it consumed again.
        if (nbands > 1) {
            for (int i = tid; i < n; i += CC_THREADS)
                if (L.val[i] == 0) v.d[(int)(L.xy[i] >> 16) * w + (int)(L.xy[i] & 0xffffu)] = 0;
            __syncthreads();
        }
    }
    if (tid == 0) {
        t.path[frame] = 1;
        if (io.nrefined) {
            if (split) atomicAdd(io.nrefined + frame, L.nref);  // (zeroed by the launcher)
            else io.nrefined[frame] = L.nref;
        }
    }
    if (sparse && io.next_cnt) {
        // the cells of the next level down, around the points as they are now (the barrier at the end of the last band
        // has made them visible): saves a launch -- and its dependent round trips under a saturated HBM -- per level.
        // They go into the OTHER list buffer: a workgroup of a split level may get here while the others still read
        // this level's.  The tables are dead: the cut of the next level's points and the split list build in them.
        uint32_t* const nlist = io.next_list + (long long)frame * io.list_pitch;
        if (!split) {
            static_assert(sizeof(L.u) >= sizeof(PartScratch), "partition_points works in the LIFO storage");
            const int nnext = partition_points(pts, lv, npts, level - 1, io.leader + (long long)frame * io.pitch,
                                               *reinterpret_cast<PartScratch*>(&L.u), io.subsets);
            list_cells(io.next_w, io.next_h, pts, lv, npts, level - 1, reinterpret_cast<uint32_t*>(&L.w), L.edge, &L.nload,
                       nlist, io.list_pitch, io.next_max_items, next_hdr, psub, nnext);
        } else {
            list_cells_split(io.next_w, io.next_h, pts, lv, npts, level - 1, reinterpret_cast<uint32_t*>(&L.w), &L.nload, L.xy, LN,
                             io.cell_cnt + kCellHdr * frame, nlist, io.list_pitch, io.next_max_items, next_hdr, psub, sub, nsub);
        }
    }
    if (clk) {
        tick(7);
        tk[8] = nraw; tk[9] = npts; tk[10] = nbands; tk[11] = level;
    }
}

template <int N, class K, class... A>
static void launch_lds_grid(K kernel, dim3 grid, hipStream_t s, A... args);
template <int N, class K, class... A>
static void launch_lds(K kernel, int nframes, hipStream_t s, A... args) { launch_lds_grid<N>(kernel, dim3(nframes), s, args...); }
template <int N, class K, class... A>
static void launch_lds_grid(K kernel, dim3 grid, hipStream_t s, A... args) {
    // MRGINGHAM_AMD_CC_LDS_PAD: extra bytes of dynamic LDS per workgroup (experiment: where does the allocation
    // stop fitting into one slot of the pixel kernels?)
#ifdef MRG_EXPERIMENT
    static const int pad = [] { const char* e = getenv("MRGINGHAM_AMD_CC_LDS_PAD"); return e ? atoi(e) : 0; }();
#else
    constexpr int pad = 0;
#endif
    static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LdsCCT<N>) + pad), true);
    (void)once;
    hipLaunchKernelGGL(kernel, grid, dim3(CC_THREADS), sizeof(LdsCCT<N>) + pad, s, args...);
}

void launch_cc_detect_lds(const LevelBatch& lb, const CompTables& t, int level, const DetectOut& out, int frame0,
                          int nframes, hipStream_t s) {
    if (!t.lds_path || nframes <= 0) return;
    launch_lds<2048>(cc_detect_lds_kernel<2048>, nframes, s, lb, t, level, out, frame0);
}

void launch_cc_detect_levels(const LevelBatch* lbs, const CompTables* ts, const int* levels, const DetectOut* outs, int nlevels,
                             int nframes, hipStream_t s) {
    if (nframes <= 0 || nlevels <= 0) return;
    if (nlevels == 1 || nlevels > kDetectLevelsMax) {
        for (int k = 0; k < nlevels; ++k) launch_cc_detect(lbs[k], ts[k], levels[k], outs[k], 0, nframes, s);
        return;
    }
    DetectLevels a;
    bool lds = true;
    for (int k = 0; k < kDetectLevelsMax; ++k) {
        const int q = k < nlevels ? k : 0;
        a.lb[k] = lbs[q];
        a.t[k] = ts[q];
        a.level[k] = levels[q];
        a.out[k] = outs[q];
        lds = lds && ts[q].lds_path;
    }
    if (lds) {
        static bool once = (hipFuncSetAttribute(reinterpret_cast<const void*>(cc_detect_lds_levels_kernel<2048>),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(LdsCCT<2048>)), true);
        (void)once;
        hipLaunchKernelGGL(cc_detect_lds_levels_kernel<2048>, dim3(nframes, nlevels), dim3(CC_THREADS), sizeof(LdsCCT<2048>), s, a);
    }
    hipLaunchKernelGGL(cc_detect_levels_kernel, dim3(nframes, nlevels), dim3(CCG_THREADS), 0, s, a);
}

void launch_sparse_cells(const LevelBatch& lb, const CompTables& t, int level, const RefineIO& io, uint32_t* cell_list,
                         int32_t* cell_cnt, int list_pitch, int frame0, int nframes, hipStream_t s, int32_t* cnt_all, int nframes_all) {
    if (nframes <= 0) return;
    // the masks of chess_cells_kernel (32 B per micro-tile) go where the pixel -> index map of a dense level is
    hipLaunchKernelGGL(sparse_cells_kernel, dim3(nframes), dim3(CC_THREADS), 0, s, lb.w, lb.h, level, io, cell_list, cell_cnt,
                       list_pitch, t.gidx_pitch / 4, frame0, cnt_all, nframes_all);
}

void launch_cc_refine_lds(const LevelBatch& lb, const CompTables& t, int level, const RefineIO& io, int frame0,
                          int nframes, hipStream_t s) {
    if (!t.lds_path || nframes <= 0) return;
    if (t.lds_path & kLdsPathSparse) {
        // up to kSubsets workgroups per frame (those that find no subset of theirs leave at once); they ADD to nrefined
        if (io.nrefined) (void)hipMemsetAsync(io.nrefined + frame0, 0, (size_t)nframes * 4, s);
        const int ky = io.subsets < 1 ? 1 : (io.subsets > kSubsets ? kSubsets : io.subsets);
        launch_lds_grid<2048>(cc_refine_lds_kernel<2048, true>, dim3(nframes, ky), s, lb, t, level, io, frame0);
    } else {
        launch_lds<2048>(cc_refine_lds_kernel<2048, false>, nframes, s, lb, t, level, io, frame0);
    }
}

}  // namespace mrg
