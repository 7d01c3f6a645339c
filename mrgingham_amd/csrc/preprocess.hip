// Contrast preprocessing the reference CLI applies to an 8-bit frame before detection
// (mrgingham-from-image.cc:38-45, :71-79):
//     cv::normalize(image, image, 0, 255, NORM_MINMAX);  clahe->apply(image, image1);
// with cv::createCLAHE() defaults (8x8 tiles) and setClipLimit(8).  Row (f)-2 of the scope table:
// beside the hot path, so that raw camera frames can go straight to HBM.
//
// The arithmetic is OpenCV's (un-vendored upstream, version unpinned) -> PARITY UNPINNED; this
// follows OpenCV's published algorithm (core: minMaxIdx + convertTo(float scale, shift);
// imgproc clahe.cpp: tile histograms on the REFLECT_101-extended frame, clip + redistribute,
// cumulative LUT, bilinear blend of the four surrounding tile LUTs in single precision) and is
// compared bit-exactly with the test suite's CPU restatement of the same.  HBM-bound byte work:
//   histograms     1 B/px read            (raw values, 16 interleaved LDS copies per workgroup;
//                                          the frame extrema and the normalisation are applied to
//                                          the 64 x 256 bins afterwards, not to the pixels)
//   tile LUTs      64 x 256 bins per frame (negligible)
//   apply          1 B/px read + 1 B/px written
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mrg {

constexpr int kTiles = 8, kBins = 256;
using u32x4n = uint32_t __attribute__((ext_vector_type(4)));  // for non-temporal 16-byte accesses

static __device__ __forceinline__ int reflect101_pp(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

static __device__ __forceinline__ uint8_t sat_u8_rint(float v) {
    const float r = __builtin_rintf(v);  // cvRound: half to even
    return (uint8_t)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
}

// normalised value of v for a frame whose extrema are (vmin, vmax): cv::normalize's double
// scale / shift, then convertTo's float multiply and add (two roundings, no fma)
static __device__ __forceinline__ uint8_t normalize_value(int v, int vmin, int vmax) {
    const double smin = vmin, smax = vmax;
    const double scale = 255. * (smax - smin > 2.220446049250313e-16 ? 1. / (smax - smin) : 0.);
    const double shift = 0. - smin * scale;
    const float a = (float)scale, b = (float)shift;
    const float prod = __fmul_rn((float)v, a);
    return sat_u8_rint(__fadd_rn(prod, b));
}

// Frame extrema from the raw tile histograms (no extra pass over the pixels): grid (nframes), 256
// threads = 256 raw values.
__global__ __launch_bounds__(256) void minmax_from_hist_kernel(const int* hist, int* mm) {
    __shared__ int rmin[4], rmax[4];
    const int frame = blockIdx.x, i = threadIdx.x;
    const int* h = hist + (long long)frame * 64 * 256;
    int any = 0;
    for (int t = 0; t < 64; ++t) any |= h[t * 256 + i];
    int smin = any ? i : 255, smax = any ? i : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        smin = min(smin, __shfl_xor(smin, o));
        smax = max(smax, __shfl_xor(smax, o));
    }
    if ((i & 63) == 0) {
        rmin[i >> 6] = smin;
        rmax[i >> 6] = smax;
    }
    __syncthreads();
    if (i == 0) {
        mm[2 * frame] = min(min(rmin[0], rmin[1]), min(rmin[2], rmin[3]));
        mm[2 * frame + 1] = max(max(rmax[0], rmax[1]), max(rmax[2], rmax[3]));
    }
}

struct ClaheGeom {
    int ew, eh;  // extended frame
    int tw, th;  // tile size
};

// grid (slabs, 64 tiles, nframes).  The workgroup histograms RAW pixel values into 16 interleaved
// LDS copies (copy = lane % 16, layout [bin][copy]: lanes that hit the same bin land in different
// banks and only 4 lanes of a wave share a counter) and adds them to the frame's raw tile histogram.
// cv::normalize is a per-frame value map, so it is applied to the BINS afterwards (clahe_lut_kernel)
// and the frame extrema it needs are read off these histograms: one pass over the pixels.
constexpr int kHistCopies = 16;
__global__ __launch_bounds__(256) void clahe_hist_kernel(FrameBatch in, ClaheGeom g, int* hist, int rows_per_block) {
    __shared__ int lh[kBins * kHistCopies];
    const int frame = blockIdx.z, tile = blockIdx.y, ty = tile / kTiles, tx = tile % kTiles;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, copy = tid & (kHistCopies - 1);
    for (int i = tid; i < kBins * kHistCopies; i += 256) lh[i] = 0;
    __syncthreads();
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    const int y0 = ty * g.th + blockIdx.x * rows_per_block, y1 = min(y0 + rows_per_block, (ty + 1) * g.th);
    const int x0 = tx * g.tw;
    const bool vec = g.ew == in.width && g.eh == in.height && g.tw % 16 == 0 && in.stride % 16 == 0 &&
                     in.frame_pitch % 16 == 0 && ((uintptr_t)in.frames & 15) == 0;
    if (vec) {
        const int chunks = g.tw / 16;
        for (int y = y0 + wv; y < y1; y += 4) {
            const uint4* row = reinterpret_cast<const uint4*>(src + (long long)y * in.stride + x0);
            for (int c = lane; c < chunks; c += 64) {
                const u32x4n vv = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(row + c));
                const uint4 v = make_uint4(vv.x, vv.y, vv.z, vv.w);
                const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        atomicAdd(&lh[((q[k] >> (8 * b)) & 0xffu) * kHistCopies + copy], 1);
                }
            }
        }
    } else {
        for (int y = y0 + wv; y < y1; y += 4) {
            const uint8_t* row = src + (long long)reflect101_pp(y, in.height) * in.stride;
            for (int x = x0 + lane; x < x0 + g.tw; x += 64)
                atomicAdd(&lh[row[reflect101_pp(x, in.width)] * kHistCopies + copy], 1);
        }
    }
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int c = 0; c < kHistCopies; ++c) total += lh[tid * kHistCopies + ((c + tid) & (kHistCopies - 1))];
    if (total) atomicAdd(hist + ((long long)frame * kTiles * kTiles + tile) * kBins + tid, total);
}

// grid (64 tiles, nframes), 256 threads = 256 bins: clip, redistribute, cumulative LUT
// The table written is indexed by the RAW pixel value: lut[tile][v] = LUT_tile[normalised(v)].
__global__ __launch_bounds__(256) void clahe_lut_kernel(const int* hist, const int* mm, ClaheGeom g, int clip,
                                                        float lut_scale, uint8_t* lut, int do_normalize) {
    __shared__ int red[4];
    __shared__ int scan[kBins];
    __shared__ uint8_t tl[kBins];
    const int frame = blockIdx.y, tile = blockIdx.x, i = threadIdx.x;
    const long long base = ((long long)frame * kTiles * kTiles + tile) * kBins;
    const int nv = do_normalize ? normalize_value(i, mm[2 * frame], mm[2 * frame + 1]) : i;
    // histogram of the normalised frame = raw bins moved to their normalised value
    scan[i] = 0;
    __syncthreads();
    atomicAdd(&scan[nv], hist[base + i]);
    __syncthreads();
    int h = scan[i];
    __syncthreads();
    if (clip > 0) {
        int over = 0;
        if (h > clip) {
            over = h - clip;
            h = clip;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) over += __shfl_xor(over, o);
        if ((i & 63) == 0) red[i >> 6] = over;
        __syncthreads();
        const int clipped = red[0] + red[1] + red[2] + red[3];
        const int batch = clipped / kBins;
        const int residual = clipped - batch * kBins;
        h += batch;
        if (residual != 0) {
            const int step = max(kBins / residual, 1);
            // for (i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++
            if (i % step == 0 && i / step < residual) ++h;
        }
    }
    // inclusive scan over the 256 bins
    scan[i] = h;
    __syncthreads();
    for (int o = 1; o < kBins; o <<= 1) {
        const int add = i >= o ? scan[i - o] : 0;
        __syncthreads();
        scan[i] += add;
        __syncthreads();
    }
    tl[i] = sat_u8_rint(__fmul_rn((float)scan[i], lut_scale));
    __syncthreads();
    lut[base + i] = tl[nv];
}

// Generic blend (any frame size): grid (ceil(w/256), ceil(h/rows), nframes), all 64 tile LUTs of the
// frame in LDS, one pixel per thread per row.
__global__ __launch_bounds__(256) void clahe_apply_kernel(FrameBatch in, ClaheGeom g, const uint8_t* lut,
                                                          uint8_t* out, int rows_per_block) {
    __shared__ __attribute__((aligned(16))) uint8_t sl[kTiles * kTiles * kBins];
    const int frame = blockIdx.z, tid = threadIdx.x;
    {
        const uint4* src = reinterpret_cast<const uint4*>(lut + (long long)frame * kTiles * kTiles * kBins);
        uint4* dst = reinterpret_cast<uint4*>(sl);
        for (int i = tid; i < kTiles * kTiles * kBins / 16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int x = blockIdx.x * 256 + tid;
    if (x >= in.width) return;
    const float inv_tw = __fdiv_rn(1.0f, (float)g.tw), inv_th = __fdiv_rn(1.0f, (float)g.th);
    const float txf = __fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f);
    int tx1 = (int)__builtin_floorf(txf), tx2 = tx1 + 1;
    const float xa = __fsub_rn(txf, (float)tx1), xa1 = __fsub_rn(1.0f, xa);
    tx1 = max(tx1, 0);
    tx2 = min(tx2, kTiles - 1);
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    uint8_t* dst = out + (long long)frame * in.width * in.height;
    const int y0 = blockIdx.y * rows_per_block, y1 = min(y0 + rows_per_block, in.height);
    for (int y = y0; y < y1; ++y) {
        const float tyf = __fsub_rn(__fmul_rn((float)y, inv_th), 0.5f);
        int ty1 = (int)__builtin_floorf(tyf), ty2 = ty1 + 1;
        const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
        ty1 = max(ty1, 0);
        ty2 = min(ty2, kTiles - 1);
        const int v = src[(long long)y * in.stride + x];
        const float l11 = sl[(ty1 * kTiles + tx1) * kBins + v], l12 = sl[(ty1 * kTiles + tx2) * kBins + v];
        const float l21 = sl[(ty2 * kTiles + tx1) * kBins + v], l22 = sl[(ty2 * kTiles + tx2) * kBins + v];
        const float top = __fmul_rn(__fadd_rn(__fmul_rn(l11, xa1), __fmul_rn(l12, xa)), ya1);
        const float bot = __fmul_rn(__fadd_rn(__fmul_rn(l21, xa1), __fmul_rn(l22, xa)), ya);
        dst[(long long)y * in.width + x] = sat_u8_rint(__fadd_rn(top, bot));
    }
}

// Fast blend for frames whose tiles are at least 256 x (16*rows) pixels and 16-byte aligned rows:
// a workgroup covers 256 columns x 16*rows rows, i.e. at most 2 x 2 interpolation cells (a cell is
// the rectangle between four neighbouring tile centres).  For each of them LDS holds, per raw pixel
// value, the four LUT bytes the blend needs as ONE dword, so a pixel costs one ds_read_b32.
// Thread = 16 adjacent pixels (one 16-byte load / store) x `rows` rows.
__global__ __launch_bounds__(256) void clahe_apply_fast_kernel(FrameBatch in, ClaheGeom g, const uint8_t* lut,
                                                               uint8_t* out, int rows) {
    __shared__ uint32_t quad[4][kBins];
    const int frame = blockIdx.z, tid = threadIdx.x;
    const int bx0 = blockIdx.x * 256, by0 = blockIdx.y * 16 * rows;
    const float inv_tw = __fdiv_rn(1.0f, (float)g.tw), inv_th = __fdiv_rn(1.0f, (float)g.th);
    auto cell_x = [&](int x) { return (int)__builtin_floorf(__fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f)); };
    auto cell_y = [&](int y) { return (int)__builtin_floorf(__fsub_rn(__fmul_rn((float)y, inv_th), 0.5f)); };
    const int cx0 = cell_x(bx0), cy0 = cell_y(by0);  // first cell (tx1 / ty1 before clamping, -1..7)
    {
        const uint8_t* fl = lut + (long long)frame * kTiles * kTiles * kBins;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int tx1 = cx0 + (c & 1), ty1 = cy0 + (c >> 1);
            const int txa = max(tx1, 0), txb = min(tx1 + 1, kTiles - 1);
            const int tya = min(max(ty1, 0), kTiles - 1), tyb = min(ty1 + 1, kTiles - 1);
            const int txa_c = min(txa, kTiles - 1);
            quad[c][tid] = (uint32_t)fl[(tya * kTiles + txa_c) * kBins + tid] |
                           ((uint32_t)fl[(tya * kTiles + txb) * kBins + tid] << 8) |
                           ((uint32_t)fl[(tyb * kTiles + txa_c) * kBins + tid] << 16) |
                           ((uint32_t)fl[(tyb * kTiles + txb) * kBins + tid] << 24);
        }
    }
    __syncthreads();
    const int x0 = bx0 + (tid & 15) * 16, ys = by0 + (tid >> 4) * rows;
    if (x0 >= in.width || ys >= in.height) return;
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    uint8_t* dst = out + (long long)frame * in.width * in.height;
    // per-column cell and weights of the thread's 16 pixels
    float xa[16];
    uint32_t cxbit = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float txf = __fsub_rn(__fmul_rn((float)(x0 + j), inv_tw), 0.5f);
        const int tx1 = (int)__builtin_floorf(txf);
        xa[j] = __fsub_rn(txf, (float)tx1);
        cxbit |= (uint32_t)(tx1 - cx0) << j;
    }
    const int ye = min(ys + rows, in.height);
    for (int y = ys; y < ye; ++y) {
        const float tyf = __fsub_rn(__fmul_rn((float)y, inv_th), 0.5f);
        const int ty1 = (int)__builtin_floorf(tyf);
        const float ya = __fsub_rn(tyf, (float)ty1), ya1 = __fsub_rn(1.0f, ya);
        const uint32_t* qrow = &quad[(ty1 - cy0) * 2][0];
        const u32x4n gvv = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(src + (long long)y * in.stride + x0));
        const uint4 gv = make_uint4(gvv.x, gvv.y, gvv.z, gvv.w);
        const uint32_t gq[4] = {gv.x, gv.y, gv.z, gv.w};
        uint32_t o[4] = {0, 0, 0, 0};
        // two pixels per packed-f32 instruction (v_pk_mul_f32 / v_pk_add_f32: the same IEEE single
        // operations as the scalar expression, nothing fused), and v_cvt_pk_u8_f32 for the
        // round-half-even + saturate + byte insert of saturate_cast<uchar>
        using f32x2 = float __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int j = 0; j < 16; j += 2) {
            const uint32_t v0 = (gq[j >> 2] >> (8 * (j & 3))) & 0xffu, v1 = (gq[j >> 2] >> (8 * ((j + 1) & 3))) & 0xffu;
            const uint32_t q0 = qrow[((cxbit >> j) & 1u) * kBins + v0], q1 = qrow[((cxbit >> (j + 1)) & 1u) * kBins + v1];
            const f32x2 l11 = {(float)(q0 & 0xffu), (float)(q1 & 0xffu)};
            const f32x2 l12 = {(float)((q0 >> 8) & 0xffu), (float)((q1 >> 8) & 0xffu)};
            const f32x2 l21 = {(float)((q0 >> 16) & 0xffu), (float)((q1 >> 16) & 0xffu)};
            const f32x2 l22 = {(float)(q0 >> 24), (float)(q1 >> 24)};
            const f32x2 a = {xa[j], xa[j + 1]};
            const f32x2 a1 = (f32x2){1.0f, 1.0f} - a;
            const f32x2 top = (l11 * a1 + l12 * a) * (f32x2){ya1, ya1};
            const f32x2 bot = (l21 * a1 + l22 * a) * (f32x2){ya, ya};
            const f32x2 r = top + bot;
            o[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(r.x, j & 3, o[j >> 2]);
            o[j >> 2] = __builtin_amdgcn_cvt_pk_u8_f32(r.y, (j + 1) & 3, o[j >> 2]);
        }
        *reinterpret_cast<uint4*>(dst + (long long)y * in.width + x0) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

size_t clahe_scratch_bytes(int nframes) {
    // extrema (2 ints) + 64 histograms + 64 LUTs per frame
    return (size_t)nframes * (2 * sizeof(int) + (size_t)kTiles * kTiles * kBins * (sizeof(int) + 1)) + 256;
}

// normalize + CLAHE(clip_limit) of every frame; `out` receives dense w x h bytes per frame.
// `scratch` holds clahe_scratch_bytes(nframes).  Returns false when the frame is too small to tile.
bool launch_clahe(const FrameBatch& in, int nframes, double clip_limit, bool do_normalize, uint8_t* out,
                  void* scratch, hipStream_t s) {
    if (nframes <= 0 || in.width <= 0 || in.height <= 0) return true;
    ClaheGeom g;
    g.ew = in.width;
    g.eh = in.height;
    if (in.width % kTiles != 0 || in.height % kTiles != 0) {
        g.ew = in.width + (kTiles - in.width % kTiles);
        g.eh = in.height + (kTiles - in.height % kTiles);
    }
    g.tw = g.ew / kTiles;
    g.th = g.eh / kTiles;
    if (g.tw <= 0 || g.th <= 0) return false;
    const int area = g.tw * g.th;
    const float lut_scale = (float)(kBins - 1) / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * area / kBins);
        if (clip < 1) clip = 1;
    }
    // scratch layout (16-byte aligned pieces): extrema | histograms | LUTs
    int* mm = (int*)scratch;
    int* hist = mm + (2 * (size_t)nframes + 3) / 4 * 4;
    uint8_t* lut = (uint8_t*)(hist + (size_t)nframes * kTiles * kTiles * kBins);
    hipMemsetAsync(hist, 0, (size_t)nframes * kTiles * kTiles * kBins * sizeof(int), s);
    {
        const int rpb = 64;
        hipLaunchKernelGGL(clahe_hist_kernel, dim3((g.th + rpb - 1) / rpb, kTiles * kTiles, nframes), dim3(256), 0, s,
                           in, g, hist, rpb);
    }
    if (do_normalize) hipLaunchKernelGGL(minmax_from_hist_kernel, dim3(nframes), dim3(256), 0, s, hist, mm);
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(kTiles * kTiles, nframes), dim3(256), 0, s, hist, mm, g, clip, lut_scale,
                       lut, do_normalize ? 1 : 0);
    const int rows = 8;
    const bool fast = g.tw >= 256 && g.th >= 16 * rows && in.width % 16 == 0 && in.stride % 16 == 0 &&
                      in.frame_pitch % 16 == 0 && ((uintptr_t)in.frames & 15) == 0 && ((uintptr_t)out & 15) == 0;
    if (fast) {
        hipLaunchKernelGGL(clahe_apply_fast_kernel,
                           dim3((in.width + 255) / 256, (in.height + 16 * rows - 1) / (16 * rows), nframes), dim3(256),
                           0, s, in, g, lut, out, rows);
    } else {
        const int rpb = 32;
        hipLaunchKernelGGL(clahe_apply_kernel, dim3((in.width + 255) / 256, (in.height + rpb - 1) / rpb, nframes),
                           dim3(256), 0, s, in, g, lut, out, rpb);
    }
    return true;
}

}  // namespace mrg
