// Pyramid level images and the caller-side box blur, for gfx950.
//
// decimate: what find_chessboard_corners.cc:445-452 asks of
//   cv::resize(src, dst, Size(), 1/2^L, 1/2^L, INTER_LINEAR) on CV_8UC1:
//   every level is cut from the FULL-RESOLUTION frame (not cascaded).
//   L = 1 -> OpenCV's 2x2 area-fast path, (a+b+c+d+2)>>2, with the partial-cell
//            average (round half to even) at a ragged right/bottom edge;
//   L >= 2 -> 11-bit fixed-point bilinear whose sample point is the cell centre:
//            the four pixels at (s/2-1, s/2) of each s x s cell, weights 1/4,
//            right column / bottom row replicated when they fall off the frame.
//   (OpenCV arithmetic, not vendored by the reference: parity unpinned.)
// box blur: cv::blur((2r+1)^2), BORDER_REFLECT_101, (sum + area/2)/area
//   (mrgingham-from-image.cc:106-111).
#include "common.h"
#include "kernels.h"

namespace mrg {

__device__ __forceinline__ int clipi(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

// One output pixel of pyramid level `level` (>= 1).
__device__ __forceinline__ int decimate_pixel(const uint8_t* src, int W, int H, int st, int level, int dx, int dy) {
    if (level == 1) {
        const int sx0 = 2 * dx, sy0 = 2 * dy;
        if (sy0 >= H) return 0;
        if (sy0 + 2 <= H && dx < W / 2) {
            const uint8_t* r0 = src + (long long)sy0 * st + sx0;
            return (r0[0] + r0[1] + r0[st] + r0[st + 1] + 2) >> 2;
        }
        int sum = 0, count = 0;
        for (int sy = 0; sy < 2 && sy0 + sy < H; ++sy)
            for (int sx = 0; sx < 2 && sx0 + sx < W; ++sx) {
                sum += src[(long long)(sy0 + sy) * st + sx0 + sx];
                ++count;
            }
        if (count == 0) return 0;
        int q = sum / count;  // cvRound((float)sum/count): count is 1, 2 or 4, ties to even
        const int rem = sum - q * count;
        if (2 * rem > count || (2 * rem == count && (q & 1))) ++q;
        return q;
    }
    const int s = 1 << level;
    int sx = s * dx + s / 2 - 1;
    int a0 = 1024, a1 = 1024;
    if (sx >= W - 1) { sx = W - 1; a0 = 2048; a1 = 0; }
    const int sx1 = sx + 1 < W ? sx + 1 : sx;
    const int sy = s * dy + s / 2 - 1;
    const uint8_t* r0 = src + (long long)clipi(sy, H) * st;
    const uint8_t* r1 = src + (long long)clipi(sy + 1, H) * st;
    const int S0 = r0[sx] * a0 + r0[sx1] * a1;
    const int S1 = r1[sx] * a0 + r1[sx1] * a1;
    return (((1024 * (S0 >> 4)) >> 16) + ((1024 * (S1 >> 4)) >> 16) + 2) >> 2;
}

__global__ __launch_bounds__(256) void decimate_kernel(FrameBatch in, int level, uint8_t* out, long long out_pitch,
                                                       int ow, int oh, int frame0) {
    const int frame = frame0 + blockIdx.z;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= ow || dy >= oh) return;
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    const int v = decimate_pixel(src, in.width, in.height, in.stride, level, dx, dy);
    out[(long long)frame * out_pitch + (long long)dy * ow + dx] = (uint8_t)v;
}

void launch_decimate(const FrameBatch& in, int level, uint8_t* out, long long out_pitch, int ow, int oh, int frame0,
                     int nframes, hipStream_t s) {
    if (ow <= 0 || oh <= 0 || nframes <= 0) return;
    dim3 grid((ow + 63) / 64, (oh + 3) / 4, nframes);
    hipLaunchKernelGGL(decimate_kernel, grid, dim3(256), 0, s, in, level, out, out_pitch, ow, oh, frame0);
}

// Levels 1..3 in ONE pass over the frame: the thread grid is the level-1 image;
// the thread of every 2nd / 4th level-1 pixel also produces the level-2 / level-3
// pixel of its cell, whose four source pixels its neighbours have just pulled
// through the vector cache.  HBM traffic: the frame once + the three level images.
__global__ __launch_bounds__(256) void pyramid_kernel(FrameBatch in, PyramidOut po, int top) {
    const int frame = blockIdx.z;
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    const int W = in.width, H = in.height, st = in.stride;
    if (po.out[0] && dx < po.w[0] && dy < po.h[0])
        po.out[0][((long long)frame * po.h[0] + dy) * po.w[0] + dx] = (uint8_t)decimate_pixel(src, W, H, st, 1, dx, dy);
    if (top >= 2 && po.out[1] && !(dx & 1) && !(dy & 1)) {
        const int X = dx >> 1, Y = dy >> 1;
        if (X < po.w[1] && Y < po.h[1])
            po.out[1][((long long)frame * po.h[1] + Y) * po.w[1] + X] = (uint8_t)decimate_pixel(src, W, H, st, 2, X, Y);
    }
    if (top >= 3 && po.out[2] && !(dx & 3) && !(dy & 3)) {
        const int X = dx >> 2, Y = dy >> 2;
        if (X < po.w[2] && Y < po.h[2])
            po.out[2][((long long)frame * po.h[2] + Y) * po.w[2] + X] = (uint8_t)decimate_pixel(src, W, H, st, 3, X, Y);
    }
}

// Fast path of the pyramid for frames whose width is a multiple of 16 and height
// a multiple of 8 (every BASELINE size): one thread owns a 16 x 8 source block,
// pulls it in with eight 16-byte loads and emits 8x4 level-1, 4x2 level-2 and
// 2x1 level-3 pixels with 8- / 4- / 2-byte stores.  Same arithmetic as
// decimate_pixel (whole cells only, so every level is (a+b+c+d+2)>>2).
__global__ __launch_bounds__(256) void pyramid_fast_kernel(FrameBatch in, PyramidOut po, int top) {
    const int frame = blockIdx.z;
    const int bx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int by = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = in.width, H = in.height, st = in.stride;
    if (bx * 16 >= W || by * 8 >= H) return;
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch + (long long)(by * 8) * st + bx * 16;
    uint32_t r[8][4];
    // non-temporal loads: the frame is streamed through once here (-13 % launch time measured)
    using u32x4v = uint32_t __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32x4v q = __builtin_nontemporal_load(reinterpret_cast<const u32x4v*>(src + (long long)i * st));
        r[i][0] = q.x; r[i][1] = q.y; r[i][2] = q.z; r[i][3] = q.w;
    }
    auto px = [&](int row, int col) -> uint32_t { return (r[row][col >> 2] >> (8 * (col & 3))) & 0xffu; };

    if (po.out[0]) {
        uint8_t* o = po.out[0] + ((long long)frame * po.h[0] + by * 4) * po.w[0] + bx * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t packed[2];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                // byte pairs of both rows summed as two u16 lanes: (b0+b1, b2+b3)
                const uint32_t a = r[2 * j][d], b = r[2 * j + 1][d];
                uint32_t v = (a & 0x00ff00ffu) + ((a >> 8) & 0x00ff00ffu) + (b & 0x00ff00ffu) + ((b >> 8) & 0x00ff00ffu);
                v = ((v + 0x00020002u) >> 2) & 0x00ff00ffu;
                const uint32_t two = (v | (v >> 8)) & 0xffffu;  // two level-1 pixels
                if (d & 1) packed[d >> 1] |= two << 16; else packed[d >> 1] = two;
            }
            __builtin_memcpy(o + (long long)j * po.w[0], packed, 8);
        }
    }
    if (top >= 2 && po.out[1]) {
        uint8_t* o = po.out[1] + ((long long)frame * po.h[1] + by * 2) * po.w[1] + bx * 4;
#pragma unroll
        for (int Y = 0; Y < 2; ++Y) {
            uint32_t v = 0;
#pragma unroll
            for (int X = 0; X < 4; ++X) {
                const int c = 4 * X + 1, rr = 4 * Y + 1;
                v |= ((px(rr, c) + px(rr, c + 1) + px(rr + 1, c) + px(rr + 1, c + 1) + 2) >> 2) << (8 * X);
            }
            __builtin_memcpy(o + (long long)Y * po.w[1], &v, 4);
        }
    }
    if (top >= 3 && po.out[2]) {
        uint8_t* o = po.out[2] + ((long long)frame * po.h[2] + by) * po.w[2] + bx * 2;
        uint16_t v = 0;
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int c = 8 * X + 3;
            v |= (uint16_t)(((px(3, c) + px(3, c + 1) + px(4, c) + px(4, c + 1) + 2) >> 2) << (8 * X));
        }
        __builtin_memcpy(o, &v, 2);
    }
}

int pyramid_lds_pad = -1;  // tuning hook "pyramid_lds_pad" (experiment builds): -1 = the rule below, else bytes

// `gentle`: the pass runs beside latency-bound kernels (the component chains of sparse steps).  At full occupancy it
// keeps ~6 TB/s of requests in flight and every global round trip of those kernels takes 2-3x as long (refinement 110 ->
// 250 us per launch, cell responses 30 -> 125 us: rocprofv3 timeline of `bench.py --sparse-refine`); with two
// workgroups per CU -- 80 KB of dynamic LDS nobody uses -- it is ~20 % slower itself and the step is 13 % shorter
// (0.380 -> 0.330 ms per 64 x 4096x3072; 3 workgroups 0.350, 1 workgroup 0.383).
void launch_pyramid(const FrameBatch& in, const PyramidOut& po, int top, int nframes, hipStream_t s, bool gentle) {
    if (nframes <= 0 || top < 1) return;
    const int pad = pyramid_lds_pad >= 0 ? pyramid_lds_pad : (gentle ? 80000 : 0);
    const bool aligned16 = in.stride % 16 == 0 && in.frame_pitch % 16 == 0 && ((uintptr_t)in.frames & 15) == 0;
    if (in.width % 16 == 0 && in.height % 8 == 0 && in.width > 0 && in.height > 0 && aligned16) {
        dim3 grid((in.width / 16 + 63) / 64, (in.height / 8 + 3) / 4, nframes);
        hipLaunchKernelGGL(pyramid_fast_kernel, grid, dim3(256), pad, s, in, po, top);
        return;
    }
    int gw = 0, gh = 0;  // thread grid in level-1 pixels, large enough for every requested level
    for (int L = 1; L <= top && L <= 3; ++L)
        if (po.out[L - 1]) {
            gw = max(gw, po.w[L - 1] << (L - 1));
            gh = max(gh, po.h[L - 1] << (L - 1));
        }
    if (gw <= 0 || gh <= 0) return;
    dim3 grid((gw + 63) / 64, (gh + 3) / 4, nframes);
    hipLaunchKernelGGL(pyramid_kernel, grid, dim3(256), 0, s, in, po, top);
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

__global__ __launch_bounds__(256) void box_blur_kernel(FrameBatch in, int radius, uint8_t* out, int frame0) {
    const int frame = frame0 + blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int W = in.width, H = in.height;
    if (x >= W || y >= H) return;
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    const int area = (2 * radius + 1) * (2 * radius + 1);
    int sum = 0;
    for (int dy = -radius; dy <= radius; ++dy) {
        const uint8_t* row = src + (long long)reflect101(y + dy, H) * in.stride;
        for (int dx = -radius; dx <= radius; ++dx) sum += row[reflect101(x + dx, W)];
    }
    out[((long long)frame * H + y) * W + x] = (uint8_t)((sum + area / 2) / area);
}

// 3x3 fast path (the CLI's default radius 1): a thread owns a 16-pixel column chunk and rolls down
// `rows` rows, so every input row is loaded once (one 16-byte load + the two bytes beside the chunk)
// and every output row leaves as one 16-byte store: ~1.06 B/px read + 1 B/px written.
// Column sums of three rows are kept as packed 16-bit pairs; (sum + 4) / 9 = ((sum + 4) * 7282) >> 16
// exactly for sum <= 2295.  Needs width % 16 == 0 and 16-byte aligned rows.
struct Cols18 {
    uint32_t p[9];  // 18 columns (x0-1 .. x0+16) as 9 packed u16 pairs
};
__device__ __forceinline__ Cols18 blur3_load_row(const uint8_t* row, int x0, int W) {
    const uint4 g = *reinterpret_cast<const uint4*>(row + x0);
    // BORDER_REFLECT_101: column -1 is column 1, column W is column W-2
    const uint32_t left = row[x0 > 0 ? x0 - 1 : 1], right = row[x0 + 16 < W ? x0 + 16 : W - 2];
    const uint32_t q[4] = {g.x, g.y, g.z, g.w};
    Cols18 c;
    // pair k holds columns (2k-1, 2k) relative to x0: pair 0 = (left, b0), pair 1 = (b1, b2), ...
    c.p[0] = left | ((q[0] & 0xffu) << 16);
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const int b = 2 * k - 1;  // byte index of the low half
        const uint32_t lo = (q[b >> 2] >> (8 * (b & 3))) & 0xffu;
        const uint32_t hi = (q[(b + 1) >> 2] >> (8 * ((b + 1) & 3))) & 0xffu;
        c.p[k] = lo | (hi << 16);
    }
    c.p[8] = (q[3] >> 24) | (right << 16);
    return c;
}
__global__ __launch_bounds__(256) void box_blur3_kernel(FrameBatch in, uint8_t* out, int frame0, int rows) {
    const int frame = frame0 + blockIdx.z;
    const int W = in.width, H = in.height;
    const int x0 = (blockIdx.x * 64 + (threadIdx.x & 63)) * 16;  // a wave spans 1024 pixels of one row slab
    const int y0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * rows, y1 = min(y0 + rows, H);
    if (x0 >= W || y0 >= H) return;
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    uint8_t* dst = out + (long long)frame * W * H;
    auto rowp = [&](int y) { return src + (long long)reflect101(y, H) * in.stride; };
    Cols18 a = blur3_load_row(rowp(y0 - 1), x0, W), b = blur3_load_row(rowp(y0), x0, W);
    for (int y = y0; y < y1; ++y) {
        const Cols18 c = blur3_load_row(rowp(y + 1), x0, W);
        uint32_t v[9];  // vertical sums, packed pairs, each half <= 765
#pragma unroll
        for (int k = 0; k < 9; ++k) v[k] = a.p[k] + b.p[k] + c.p[k];
        // output pixel j (0..15) = columns j-1, j, j+1 relative to x0 = pair-halves (j), (j+1), (j+2)
        // in the flat sequence v[0].lo, v[0].hi, v[1].lo, ...
        uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            auto col = [&](int i) { return (v[i >> 1] >> (16 * (i & 1))) & 0xffffu; };
            const uint32_t sum = col(j) + col(j + 1) + col(j + 2);
            o[j >> 2] |= (((sum + 4u) * 7282u) >> 16) << (8 * (j & 3));
        }
        *reinterpret_cast<uint4*>(dst + (long long)y * W + x0) = make_uint4(o[0], o[1], o[2], o[3]);
        a = b;
        b = c;
    }
}

void launch_box_blur(const FrameBatch& in, int radius, uint8_t* out, int frame0, int nframes, hipStream_t s) {
    if (in.width <= 0 || in.height <= 0 || nframes <= 0) return;
    const bool aligned = in.width % 16 == 0 && in.stride % 16 == 0 && in.frame_pitch % 16 == 0 &&
                         ((uintptr_t)in.frames & 15) == 0 && ((uintptr_t)out & 15) == 0;
    if (radius == 1 && aligned && in.width >= 16 && in.height >= 2) {
        const int rows = 32;
        dim3 grid((in.width / 16 + 63) / 64, (in.height + 4 * rows - 1) / (4 * rows), nframes);
        hipLaunchKernelGGL(box_blur3_kernel, grid, dim3(256), 0, s, in, out, frame0, rows);
        return;
    }
    dim3 grid((in.width + 63) / 64, (in.height + 3) / 4, nframes);
    hipLaunchKernelGGL(box_blur_kernel, grid, dim3(256), 0, s, in, radius, out, frame0);
}

}  // namespace mrg
