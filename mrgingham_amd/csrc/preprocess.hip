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

#include "common.h"
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
int clahe_hist_copies = 16;  // tuning hook "clahe_hist_copies" (experiment builds: 8 / 16 / 32)
template <int kHistCopies>
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
        // the slab's 16-byte chunks, rows x chunks per row, dealt to the 256 threads in one flat sequence (a tile row of 512
        // pixels is 32 chunks: a wave per row, as before round 6, left half the lanes idle), two loads in flight per thread
        const int chunks = g.tw / 16, nchunks = (y1 - y0) * chunks;
        const uint8_t* base = src + (long long)y0 * in.stride + x0;
        auto load = [&](int i) {
            const int r = i / chunks, c = i - r * chunks;
            return __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(base + (long long)r * in.stride + 16 * c));
        };
        auto count = [&](const u32x4n& vv) {
            const uint32_t q[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    atomicAdd(&lh[((q[k] >> (8 * b)) & 0xffu) * kHistCopies + copy], 1);
            }
        };
        for (int i = tid; i < nchunks; i += 512) {
            const u32x4n a = load(i);
            if (i + 256 < nchunks) {
                const u32x4n b = load(i + 256);
                count(a);
                count(b);
            } else {
                count(a);
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

// ---------------------------------------------------------------------------------------------------------------
// CLAHE blend + 3x3 box blur in ONE pass (round 6): the reference tool's default chain ends clahe->apply(..);
// cv::blur(.., Size(3, 3)) (mrgingham-from-image.cc:71-111), and as two kernels that was a read + a write of the
// frame each -- 4 B/px where 2 suffice.  Here a thread owns a 16-pixel column chunk and rolls down a band of ROWS
// output rows: per input row it loads its 16 raw pixels and the pixel on either side, blends all 18 (the neighbours'
// two are recomputed: the same single-precision expression at the neighbour's own coordinates gives the same byte),
// keeps the horizontal 3-sums of the last two rows as packed u16 pairs in registers and stores one blurred row.
// The image border is BORDER_REFLECT_101 OF THE BLENDED IMAGE: column -1 is column 1 blended with column 1's weights,
// row -1 is row 1.  A wave spans 1024 pixels of a row; the four waves of a workgroup take four consecutive bands.
// The interpolation cells (rectangles between four neighbouring tile centres) such a workgroup meets -- at most
// 9 across, 2 down when 4 * ROWS + 2 <= tile height -- come from a per-frame table clahe_quad_kernel makes once:
// quad[cy][cx][v] = the four LUT values a pixel of value v needs in cell (cx - 1, cy - 1) as four HALF floats (8 bytes: one
// ds_read_b64; integers up to 255 are exact in f16): v_fma_mix_f32 takes an f16 operand as it is, so l * weight is ONE
// instruction per product -- round(l * w + 0) = the IEEE single product OpenCV computes -- where the byte table needed
// a v_cvt_f32_ubyte per value and then the multiply (291 -> 246 VALU instructions per 16 pixels).
// LDS layout [cx][cy][256] entries, so that the cell row is a constant offset.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCells = kTiles + 1;  // interpolation cells per axis: tx1 = -1 .. 7

// grid (81 cells, nframes), 256 threads = 256 pixel values
using half2v = _Float16 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void clahe_quad_kernel(const uint8_t* lut, uint2* quad) {
    const int frame = blockIdx.y, cell = blockIdx.x, cy = cell / kCells - 1, cx = cell % kCells - 1, v = threadIdx.x;
    const uint8_t* fl = lut + (long long)frame * kTiles * kTiles * kBins;
    const int txa = max(cx, 0), txb = min(cx + 1, kTiles - 1), tya = max(cy, 0), tyb = min(cy + 1, kTiles - 1);
    const half2v top = {(_Float16)(int)fl[(tya * kTiles + txa) * kBins + v], (_Float16)(int)fl[(tya * kTiles + txb) * kBins + v]};
    const half2v bot = {(_Float16)(int)fl[(tyb * kTiles + txa) * kBins + v], (_Float16)(int)fl[(tyb * kTiles + txb) * kBins + v]};
    quad[((long long)frame * kCells * kCells + cell) * kBins + v] = make_uint2(__builtin_bit_cast(uint32_t, top), __builtin_bit_cast(uint32_t, bot));
}

namespace fused {
using f32x2 = float __attribute__((ext_vector_type(2)));
using u32x2 = uint32_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mad_u16lo(uint32_t a, uint32_t b, uint32_t c) {  // a.lo16 * b.lo16 + c
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t mad_u16hi(uint32_t a, uint32_t b, uint32_t c) {  // a.hi16 * b.lo16 + c
    uint32_t r;
    asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

struct Raw18 {      // one input row of a thread: 16 pixels and the one on either side
    uint4 g;
    uint32_t left, right;
};

// f16 half of `h` (lo / hi) times the f32 `w`, as an f32: round(h * w + 0), one instruction
__device__ __forceinline__ float mul_f16lo(uint32_t h, float w) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(w));
    return r;
}
__device__ __forceinline__ float mul_f16hi(uint32_t h, float w) {
    float r;
    asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(w));
    return r;
}

// the horizontal 3-sums of the blended row, 16 outputs as 8 packed u16 pairs.  CY: the row's cell row inside the LDS table.
template <int CY>
__device__ __forceinline__ void blend_hsum(const char* q, const uint32_t (&cb)[18], const float (&xa)[18], const float (&xa1)[18],
                                           float ya, const Raw18& r, uint32_t (&hs)[8]) {
    const float ya1 = __fsub_rn(1.0f, ya);
    const uint32_t gq[4] = {r.g.x, r.g.y, r.g.z, r.g.w};
    uint32_t v[18];
    v[0] = r.left;
#pragma unroll
    for (int j = 0; j < 16; ++j) v[1 + j] = (gq[j >> 2] >> (8 * (j & 3))) & 0xffu;
    v[17] = r.right;
    uint32_t p[9];  // blended columns -1 .. 16 as packed u16 pairs: p[k] = (column 2k - 1, column 2k)
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int i0 = 2 * k, i1 = 2 * k + 1;
        const uint2 q0 = *reinterpret_cast<const uint2*>(q + CY * (kBins * 8) + cb[i0] + 8 * v[i0]);
        const uint2 q1 = *reinterpret_cast<const uint2*>(q + CY * (kBins * 8) + cb[i1] + 8 * v[i1]);
        const f32x2 p11 = {mul_f16lo(q0.x, xa1[i0]), mul_f16lo(q1.x, xa1[i1])};
        const f32x2 p12 = {mul_f16hi(q0.x, xa[i0]), mul_f16hi(q1.x, xa[i1])};
        const f32x2 p21 = {mul_f16lo(q0.y, xa1[i0]), mul_f16lo(q1.y, xa1[i1])};
        const f32x2 p22 = {mul_f16hi(q0.y, xa[i0]), mul_f16hi(q1.y, xa[i1])};
        const f32x2 top = (p11 + p12) * (f32x2){ya1, ya1};
        const f32x2 bot = (p21 + p22) * (f32x2){ya, ya};
        // saturate_cast<uchar>(cvRound(res)): the sum is in [0, 255.001], so adding 2^23 leaves rint(res) -- round half to
        // even, the FPU's default mode -- in the low bits of the mantissa
        const f32x2 rr = (top + bot) + (f32x2){8388608.0f, 8388608.0f};
        // (the bits of the WHOLE vector: hipcc 7.2 compiles __builtin_bit_cast(uint32_t, rr.y) of a vector ELEMENT as the bits of rr.x)
        const u32x2 rb = __builtin_bit_cast(u32x2, rr);
        p[k] = __builtin_amdgcn_perm(rb.y, rb.x, 0x05040100u);
    }
    // output pixel pair (2k, 2k + 1) = columns (2k-1, 2k) + (2k, 2k+1) + (2k+1, 2k+2)
#pragma unroll
    for (int k = 0; k < 8; ++k) hs[k] = p[k] + __builtin_amdgcn_alignbit(p[k + 1], p[k], 16) + p[k + 1];
}
}  // namespace fused

template <int ROWS>
__global__ __launch_bounds__(256) void clahe_blur3_kernel(FrameBatch in, ClaheGeom g, const uint2* quad, uint8_t* out,
                                                          int ncx_max, unsigned long long* clk) {
    using namespace fused;
    const bool probe = clk != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0;  // (mrgingham_amd_sclk_mhz)
    const ClockProbe clkp = clock_probe_begin(probe);
    extern __shared__ __attribute__((aligned(16))) char qlds[];  // [ncx][2][256] entries of 8 bytes
    const int frame = blockIdx.z, tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // (a scalar: the row arithmetic below stays on the scalar unit)
    const int W = in.width, H = in.height;
    const int bx0 = blockIdx.x * 1024, by0 = blockIdx.y * 4 * ROWS;
    const float inv_tw = __fdiv_rn(1.0f, (float)g.tw), inv_th = __fdiv_rn(1.0f, (float)g.th);
    auto cell_x = [&](int x) { return (int)__builtin_floorf(__fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f)); };
    auto cell_y = [&](int y) { return (int)__builtin_floorf(__fsub_rn(__fmul_rn((float)y, inv_th), 0.5f)); };
    // cells of the workgroup's samples: columns bx0 - 1 .. bx0 + 1024, rows by0 - 1 .. by0 + 4 ROWS (reflected ones are inside)
    const int cxa = cell_x(max(bx0 - 1, 0)), ncx = min(cell_x(min(bx0 + 1024, W - 1)) - cxa + 1, ncx_max);
    const int cya = cell_y(max(by0 - 1, 0));
    {
        const uint4* src = reinterpret_cast<const uint4*>(quad + (long long)frame * kCells * kCells * kBins);
        uint4* dst = reinterpret_cast<uint4*>(qlds);
        constexpr int kPer = kBins / 2;  // 16-byte pieces of a cell's table
        for (int i = tid; i < ncx * 2 * kPer; i += 256) {
            const int c = i / (2 * kPer), rest = i - c * (2 * kPer), r = rest / kPer, e = rest % kPer;
            const int cy = min(cya + r, kTiles - 1);
            dst[i] = src[(((cy + 1) * kCells) + (cxa + c + 1)) * kPer + e];
        }
    }
    __syncthreads();
    const int x0 = bx0 + lane * 16;
    const int y0 = by0 + wv * ROWS, y1 = min(y0 + ROWS, H);
    if (x0 >= W || y0 >= H) return;  // (never thread 0 of workgroup 0: the probe's second half is reached)
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    uint8_t* dst = out + (long long)frame * W * H;
    // per-column constants of the 18 samples: sample i is column x0 + i - 1, reflected at the frame's edge
    uint32_t cb[18];
    float xa[18], xa1[18];
    const int xl = x0 > 0 ? x0 - 1 : 1, xr = x0 + 16 < W ? x0 + 16 : W - 2;
#pragma unroll
    for (int i = 0; i < 18; ++i) {
        const int x = i == 0 ? xl : (i == 17 ? xr : x0 + i - 1);
        const float txf = __fsub_rn(__fmul_rn((float)x, inv_tw), 0.5f);
        const int tx1 = (int)__builtin_floorf(txf);
        xa[i] = __fsub_rn(txf, (float)tx1);
        xa1[i] = __fsub_rn(1.0f, xa[i]);
        cb[i] = (uint32_t)(tx1 - cxa) * (2 * kBins * 8);
    }
    auto load_row = [&](int y) {  // y in -1 .. H: the blended image's REFLECT_101 border
        const int ys = y < 0 ? -y : (y >= H ? 2 * (H - 1) - y : y);
        const uint8_t* row = src + (long long)ys * in.stride;
        Raw18 r;
        const u32x4n gv = __builtin_nontemporal_load(reinterpret_cast<const u32x4n*>(row + x0));
        r.g = make_uint4(gv.x, gv.y, gv.z, gv.w);
        r.left = row[xl];
        r.right = row[xr];
        return r;
    };
    auto hsum_row = [&](int y, const Raw18& r, uint32_t (&hs)[8]) {
        const int ys = y < 0 ? -y : (y >= H ? 2 * (H - 1) - y : y);
        const float tyf = __fsub_rn(__fmul_rn((float)ys, inv_th), 0.5f);
        const int ty1 = (int)__builtin_floorf(tyf);
        const float ya = __fsub_rn(tyf, (float)ty1);
        if (ty1 == cya) blend_hsum<0>(qlds, cb, xa, xa1, ya, r, hs);  // (wave-uniform)
        else blend_hsum<1>(qlds, cb, xa, xa1, ya, r, hs);
    };
    auto emit = [&](int y, const uint32_t (&a)[8], const uint32_t (&b)[8], const uint32_t (&c)[8]) {
        uint32_t o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            // (sum + 4) / 9 = ((sum + 4) * 7282) >> 16 exactly for sum <= 2295 (box_blur3_kernel, decimate.hip)
            const uint32_t s0 = a[2 * m] + b[2 * m] + c[2 * m], s1 = a[2 * m + 1] + b[2 * m + 1] + c[2 * m + 1];
            const uint32_t t0 = mad_u16lo(s0, 7282u, 29128u), t1 = mad_u16hi(s0, 7282u, 29128u);
            const uint32_t t2 = mad_u16lo(s1, 7282u, 29128u), t3 = mad_u16hi(s1, 7282u, 29128u);
            o[m] = __builtin_amdgcn_perm(t1, t0, 0x0c0c0602u) | __builtin_amdgcn_perm(t3, t2, 0x06020c0cu);
        }
        *reinterpret_cast<uint4*>(dst + (long long)y * W + x0) = make_uint4(o[0], o[1], o[2], o[3]);
    };
    // three row sums rotate through ha / hb / hc (the loop body is three rows, so nothing is copied); the raw row after
    // the one being blended is always in flight
    uint32_t ha[8], hb[8], hc[8];
    Raw18 nx;
    {
        const Raw18 r0 = load_row(y0 - 1), r1 = load_row(y0);
        nx = load_row(y0 + 1);
        hsum_row(y0 - 1, r0, ha);
        hsum_row(y0, r1, hb);
    }
#define MRG_ROW(Y, A, B, C)                          \
    {                                                \
        const Raw18 cur = nx;                        \
        if ((Y) + 2 <= y1) nx = load_row((Y) + 2);   \
        hsum_row((Y) + 1, cur, C);                   \
        emit((Y), A, B, C);                          \
    }
    for (int y = y0; y < y1; y += 3) {
        MRG_ROW(y, ha, hb, hc)
        if (y + 1 >= y1) break;
        MRG_ROW(y + 1, hb, hc, ha)
        if (y + 2 >= y1) break;
        MRG_ROW(y + 2, hc, ha, hb)
    }
#undef MRG_ROW
    clock_probe_end(probe, clkp, clk);
}

// rows per wave of the fused kernel for a geometry: the workgroup's 4 * rows + 2 sample rows must fit one tile height
// (two cell rows); 0 = the frame is too small (or not 16-byte aligned): the two-kernel path takes it
static int fused_rows(const FrameBatch& in, const ClaheGeom& g, const uint8_t* out) {
    const bool aligned = in.width % 16 == 0 && in.width >= 32 && in.stride % 16 == 0 && in.frame_pitch % 16 == 0 &&
                         ((uintptr_t)in.frames & 15) == 0 && ((uintptr_t)out & 15) == 0 && in.height >= 2;
    if (!aligned) return 0;
    for (int rows : {32, 16, 8})
        if (4 * rows + 2 <= g.th) return rows;
    return 0;
}

size_t clahe_scratch_bytes(int nframes) {
    // extrema (2 ints) + 64 histograms + 64 LUTs + 81 cell tables per frame
    return (size_t)nframes * (2 * sizeof(int) + (size_t)kTiles * kTiles * kBins * (sizeof(int) + 1) +
                              (size_t)kCells * kCells * kBins * sizeof(uint2)) + 512;
}

static ClaheGeom clahe_geom(const FrameBatch& in) {
    ClaheGeom g;
    g.ew = in.width;
    g.eh = in.height;
    if (in.width % kTiles != 0 || in.height % kTiles != 0) {
        g.ew = in.width + (kTiles - in.width % kTiles);
        g.eh = in.height + (kTiles - in.height % kTiles);
    }
    g.tw = g.ew / kTiles;
    g.th = g.eh / kTiles;
    return g;
}

// whether launch_clahe(.., blur3 = true) blends and blurs in one pass (then it needs no intermediate image)
bool clahe_blur3_fused(const FrameBatch& in, const uint8_t* out) {
    if (in.width <= 0 || in.height <= 0) return false;
    const ClaheGeom g = clahe_geom(in);
    return g.tw > 0 && g.th > 0 && fused_rows(in, g, out) > 0;
}

// (Measured and dropped in round 6: the frame extrema folded into clahe_hist_kernel's epilogue -- two device-scope atomics per
// workgroup on 128 words cost 181 -> 226 us, more than the 10 us kernel they replace --, and a two-stream pipeline over
// chunks of frames, the table kernels of chunk k + 1 under the blend of chunk k: 0.686 against 0.653 ms per 64 x 4096x3072 and
// +0.1 ms on small batches, the cross-stream waits cost more than the overlap gives.)
// normalize + CLAHE(clip_limit) of every frame; `out` receives dense w x h bytes per frame.
// `scratch` holds clahe_scratch_bytes(nframes).  Returns false when the frame is too small to tile.
// `blur3`: followed by cv::blur(3x3) -- in the same pass when clahe_blur3_fused(in, out), else through `tmp` (dense
// w x h bytes per frame) and launch_box_blur.
bool launch_clahe(const FrameBatch& in, int nframes, double clip_limit, bool do_normalize, uint8_t* out,
                  void* scratch, hipStream_t s, bool blur3, uint8_t* tmp, unsigned long long* clk) {
    if (nframes <= 0 || in.width <= 0 || in.height <= 0) return true;
    const ClaheGeom g = clahe_geom(in);
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
        const dim3 hg((g.th + rpb - 1) / rpb, kTiles * kTiles, nframes);
        // (copies: 8 -> 262 us, 16 -> 181, 32 -> 190, 64 -> 358 per 64 x 4096x3072: it is the LDS atomic unit's rate, 8 lanes per
        // clock, not bank conflicts, that bounds this kernel -- SQ_ACTIVE_INST_LDS covers the whole launch)
#ifdef MRG_EXPERIMENT
        if (clahe_hist_copies == 8) hipLaunchKernelGGL(clahe_hist_kernel<8>, hg, dim3(256), 0, s, in, g, hist, rpb);
        else if (clahe_hist_copies == 32) hipLaunchKernelGGL(clahe_hist_kernel<32>, hg, dim3(256), 0, s, in, g, hist, rpb);
        else
#endif
        hipLaunchKernelGGL(clahe_hist_kernel<16>, hg, dim3(256), 0, s, in, g, hist, rpb);
    }
    if (do_normalize) hipLaunchKernelGGL(minmax_from_hist_kernel, dim3(nframes), dim3(256), 0, s, hist, mm);
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(kTiles * kTiles, nframes), dim3(256), 0, s, hist, mm, g, clip, lut_scale,
                       lut, do_normalize ? 1 : 0);
    if (blur3) {
        const int frows = fused_rows(in, g, out);
        if (frows > 0) {
            uint2* quad = (uint2*)(((uintptr_t)(lut + (size_t)nframes * kTiles * kTiles * kBins) + 255) & ~(uintptr_t)255);
            hipLaunchKernelGGL(clahe_quad_kernel, dim3(kCells * kCells, nframes), dim3(256), 0, s, lut, quad);
            int ncx_max = 1025 / g.tw + 2;  // cells a run of 1026 columns can meet
            if (ncx_max > kCells) ncx_max = kCells;
            const dim3 grid((in.width + 1023) / 1024, (in.height + 4 * frows - 1) / (4 * frows), nframes);
            const size_t lds = (size_t)ncx_max * 2 * kBins * sizeof(uint2);
            if (frows == 32) hipLaunchKernelGGL(clahe_blur3_kernel<32>, grid, dim3(256), lds, s, in, g, quad, out, ncx_max, clk);
            else if (frows == 16) hipLaunchKernelGGL(clahe_blur3_kernel<16>, grid, dim3(256), lds, s, in, g, quad, out, ncx_max, clk);
            else hipLaunchKernelGGL(clahe_blur3_kernel<8>, grid, dim3(256), lds, s, in, g, quad, out, ncx_max, clk);
            return true;
        }
    }
    uint8_t* const final_out = out;
    if (blur3) out = tmp;
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
    if (blur3) {
        const FrameBatch tb{out, (long long)in.width * in.height, in.width, in.height, in.width};
        launch_box_blur(tb, 1, final_out, 0, nframes, s);
    }
    return true;
}

}  // namespace mrg
