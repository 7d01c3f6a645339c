// ChESS radius-5 response kernels for gfx950 (wave64).
//
// What is computed: exactly mrgingham_ChESS_response_5 (ChESS.c:56-106) -- 16
// ring samples at radius 5 and a 3-pixel horizontal local mean per interior
// pixel, pure integer -- optionally fused with what the reference does right
// after it (find_chessboard_corners.cc:506-529): the zeroed 7-pixel frame, the
// clamp of negative responses to 0, and (new here) the compaction of "hot"
// pixels (response > 15, the only pixels the component search can ever seed
// from or extend through) into a per-frame list, so that the component search
// never has to scan the dense response.
//
// Algebra used by every kernel below (exact in integers):
//   with t1 = a+c, t2 = b+d per quadruple (a,b,c,d) = (s[i],s[i+4],s[i+8],s[i+12])
//     |a-c|     = 2*max(a,c) - t1
//     |t1 - t2| = 2*max(t1,t2) - (t1+t2)
//   so  sum_response - diff_response = 2*(Y - X),
//     Y = sum_i max(t1_i, t2_i),  X = sum_i max(a_i,c_i) + max(b_i,d_i)
//   and response = 2*(Y-X) - |M - local_mean|,  M = sum of the 16 samples.
#include "common.h"
#include "kernels.h"

namespace mrg {

// ---------------------------------------------------------------------------
// Shared epilogue: append the hot pixels of a wave to the frame's hot list.
// Must be called by all 64 lanes of a wave (uniform control flow).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void append_hot(bool hot, int p, const CompTables& t, int frame) {
    const unsigned long long m = __ballot(hot);
    if (m == 0) return;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(t.hot_cnt + frame, __popcll(m));
    base = __shfl(base, leader);
    if (hot) {
        const int idx = base + __popcll(m & ((1ull << lane) - 1ull));
        if (idx < t.cap) {
            const long long e = (long long)frame * t.cap + idx;
            t.hot_pix[e] = p;
            t.parent[e] = idx;
            t.comp_cnt[e] = 0;
            t.comp_box[e] = make_int4(0x7fffffff, 0x7fffffff, -1, -1);
            t.roots[e] = 0x7fffffff;
            t.lidx[(long long)frame * t.lidx_pitch + p] = idx;
        }
    }
}

// ---------------------------------------------------------------------------
// v0: reference-shaped kernel.  One thread per pixel column of a 64x16 tile,
// bytes staged in LDS with a 5-pixel halo.  Kept as the on-device cross-check
// of the tuned kernel (tests compare the two) -- not the production path.
// ---------------------------------------------------------------------------
constexpr int V0_TW = 64, V0_TH = 16, V0_HALO = 5;
constexpr int V0_LW = V0_TW + 2 * V0_HALO;  // 74
constexpr int V0_LH = V0_TH + 2 * V0_HALO;  // 26
constexpr int V0_LS = 76;                   // LDS row stride

template <bool CLAMP, bool HOT>
__global__ __launch_bounds__(256) void chess_v0_kernel(LevelBatch lb, CompTables t, int frame0) {
    __shared__ uint8_t tile[V0_LH * V0_LS];
    const int frame = frame0 + blockIdx.z;
    const int w = lb.w, h = lb.h;
    const uint8_t* img = lb.img + (long long)frame * lb.img_pitch;
    int16_t* resp = lb.resp + (long long)frame * lb.resp_pitch;
    const int x0 = blockIdx.x * V0_TW, y0 = blockIdx.y * V0_TH;

    for (int i = threadIdx.x; i < V0_LH * V0_LW; i += 256) {
        const int ly = i / V0_LW, lx = i - ly * V0_LW;
        int gx = x0 + lx - V0_HALO, gy = y0 + ly - V0_HALO;
        gx = min(max(gx, 0), w - 1);
        gy = min(max(gy, 0), h - 1);
        tile[ly * V0_LS + lx] = img[(long long)gy * lb.img_stride + gx];
    }
    __syncthreads();

    const int tx = threadIdx.x & 63, ty0 = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int ty = ty0 + 4 * k;
        const int x = x0 + tx, y = y0 + ty;
        const uint8_t* c = tile + (ty + V0_HALO) * V0_LS + tx + V0_HALO;
        auto S = [&](int dx, int dy) -> int { return c[dy * V0_LS + dx]; };
        // ring, ChESS.c:68-83
        const int s0 = S(2, -5), s1 = S(0, -5), s2 = S(-2, -5), s3 = S(-4, -4);
        const int s4 = S(-5, -2), s5 = S(-5, 0), s6 = S(-5, 2), s7 = S(-4, 4);
        const int s8 = S(-2, 5), s9 = S(0, 5), s10 = S(2, 5), s11 = S(4, 4);
        const int s12 = S(5, 2), s13 = S(5, 0), s14 = S(5, -2), s15 = S(4, -4);
        const int local_mean = (S(-1, 0) + S(0, 0) + S(1, 0)) * 16 / 3;  // ChESS.c:86
        int sum = 0, diff = 0, mean = 0;
#define MRG_QUAD(a, b, c_, d)                  \
    sum += abs((a) - (b) + (c_) - (d));        \
    diff += abs((a) - (c_)) + abs((b) - (d));  \
    mean += (a) + (b) + (c_) + (d);
        MRG_QUAD(s0, s4, s8, s12) MRG_QUAD(s1, s5, s9, s13) MRG_QUAD(s2, s6, s10, s14) MRG_QUAD(s3, s7, s11, s15)
#undef MRG_QUAD
        int r = sum - diff - abs(mean - local_mean);  // ChESS.c:104
        const bool inimg = x < w && y < h;
        const bool interior = x >= kMargin && x < w - kMargin && y >= kMargin && y < h - kMargin;
        if (!interior) r = 0;
        if (CLAMP) r = max(r, 0);
        if (inimg) resp[(long long)y * w + x] = (int16_t)r;
        if (HOT) append_hot(interior && r > kRespMin, y * w + x, t, frame);
    }
}

void launch_chess_v0(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, bool clamp, bool hot,
                     hipStream_t s) {
    dim3 grid((lb.w + V0_TW - 1) / V0_TW, (lb.h + V0_TH - 1) / V0_TH, nframes);
    if (hot)
        hipLaunchKernelGGL((chess_v0_kernel<true, true>), grid, dim3(256), 0, s, lb, t, frame0);
    else if (clamp)
        hipLaunchKernelGGL((chess_v0_kernel<true, false>), grid, dim3(256), 0, s, lb, t, frame0);
    else
        hipLaunchKernelGGL((chess_v0_kernel<false, false>), grid, dim3(256), 0, s, lb, t, frame0);
}

// Production entry point.  (The tuned kernel replaces this forwarding.)
void launch_chess(const LevelBatch& lb, const CompTables& t, int frame0, int nframes, bool clamp, bool hot,
                  hipStream_t s) {
    launch_chess_v0(lb, t, frame0, nframes, clamp, hot, s);
}

}  // namespace mrg
