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
//   min/max        1 B/px read
//   histograms     1 B/px read            (LDS histograms per workgroup, one global add per bin)
//   tile LUTs      64 x 256 bins per frame (negligible)
//   apply          1 B/px read + 1 B/px written
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace mrg {

namespace {

constexpr int kTiles = 8, kBins = 256;

__device__ __forceinline__ int reflect101_pp(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

__device__ __forceinline__ uint8_t sat_u8_rint(float v) {
    const float r = __builtin_rintf(v);  // cvRound: half to even
    return (uint8_t)(r < 0.f ? 0.f : r > 255.f ? 255.f : r);
}

// normalised value of v for a frame whose extrema are (vmin, vmax): cv::normalize's double
// scale / shift, then convertTo's float multiply and add (two roundings, no fma)
__device__ __forceinline__ uint8_t normalize_value(int v, int vmin, int vmax) {
    const double smin = vmin, smax = vmax;
    const double scale = 255. * (smax - smin > 2.220446049250313e-16 ? 1. / (smax - smin) : 0.);
    const double shift = 0. - smin * scale;
    const float a = (float)scale, b = (float)shift;
    const float prod = __fmul_rn((float)v, a);
    return sat_u8_rint(__fadd_rn(prod, b));
}

__global__ void minmax_init_kernel(int* mm, int nframes) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nframes) {
        mm[2 * i] = 255;
        mm[2 * i + 1] = 0;
    }
}

// grid (slabs, 1, nframes): every workgroup scans a slab of rows
__global__ __launch_bounds__(256) void minmax_kernel(FrameBatch in, int* mm, int rows_per_block) {
    const int frame = blockIdx.z;
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    const int y0 = blockIdx.x * rows_per_block, y1 = min(y0 + rows_per_block, in.height);
    const bool vec = (in.width % 16 == 0) && (in.stride % 16 == 0) && (((uintptr_t)src & 15) == 0);
    int smin = 255, smax = 0;
    if (vec) {
        const int chunks = in.width / 16;
        for (int y = y0; y < y1; ++y) {
            const uint4* row = reinterpret_cast<const uint4*>(src + (long long)y * in.stride);
            for (int c = threadIdx.x; c < chunks; c += 256) {
                const uint4 v = row[c];
                const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
#pragma unroll
                    for (int b = 0; b < 4; ++b) {
                        const int p = (q[k] >> (8 * b)) & 0xff;
                        smin = min(smin, p);
                        smax = max(smax, p);
                    }
                }
            }
        }
    } else {
        for (int y = y0; y < y1; ++y) {
            const uint8_t* row = src + (long long)y * in.stride;
            for (int x = threadIdx.x; x < in.width; x += 256) {
                const int p = row[x];
                smin = min(smin, p);
                smax = max(smax, p);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        smin = min(smin, __shfl_xor(smin, o));
        smax = max(smax, __shfl_xor(smax, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm + 2 * frame, smin);
        atomicMax(mm + 2 * frame + 1, smax);
    }
}

struct ClaheGeom {
    int ew, eh;  // extended frame
    int tw, th;  // tile size
};

// grid (slabs, 64 tiles, nframes); LDS: one histogram per wave
__global__ __launch_bounds__(256) void clahe_hist_kernel(FrameBatch in, const int* mm, ClaheGeom g, int* hist,
                                                         int rows_per_block, int do_normalize) {
    __shared__ int lh[4][kBins];
    __shared__ uint8_t norm[kBins];
    const int frame = blockIdx.z, tile = blockIdx.y, ty = tile / kTiles, tx = tile % kTiles;
    const int tid = threadIdx.x, wv = tid >> 6;
    for (int i = tid; i < 4 * kBins; i += 256) (&lh[0][0])[i] = 0;
    norm[tid] = do_normalize ? normalize_value(tid, mm[2 * frame], mm[2 * frame + 1]) : (uint8_t)tid;
    __syncthreads();
    const uint8_t* src = in.frames + (long long)frame * in.frame_pitch;
    const int y0 = ty * g.th + blockIdx.x * rows_per_block, y1 = min(y0 + rows_per_block, (ty + 1) * g.th);
    const int x0 = tx * g.tw;
    for (int y = y0 + wv; y < y1; y += 4) {
        const uint8_t* row = src + (long long)reflect101_pp(y, in.height) * in.stride;
        for (int x = x0 + (tid & 63); x < x0 + g.tw; x += 64) {
            const int v = norm[row[reflect101_pp(x, in.width)]];
            atomicAdd(&lh[wv][v], 1);
        }
    }
    __syncthreads();
    const int total = lh[0][tid] + lh[1][tid] + lh[2][tid] + lh[3][tid];
    if (total) atomicAdd(hist + ((long long)frame * kTiles * kTiles + tile) * kBins + tid, total);
}

// grid (64 tiles, nframes), 256 threads = 256 bins: clip, redistribute, cumulative LUT
__global__ __launch_bounds__(256) void clahe_lut_kernel(const int* hist, ClaheGeom g, int clip, float lut_scale,
                                                        uint8_t* lut) {
    __shared__ int red[4];
    __shared__ int scan[kBins];
    const int frame = blockIdx.y, tile = blockIdx.x, i = threadIdx.x;
    const long long base = ((long long)frame * kTiles * kTiles + tile) * kBins;
    int h = hist[base + i];
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
    lut[base + i] = sat_u8_rint(__fmul_rn((float)scan[i], lut_scale));
}

// grid (ceil(w/256), ceil(h/rows), nframes): all 64 tile LUTs of the frame live in LDS
__global__ __launch_bounds__(256) void clahe_apply_kernel(FrameBatch in, const int* mm, ClaheGeom g,
                                                          const uint8_t* lut, uint8_t* out, int rows_per_block,
                                                          int do_normalize) {
    __shared__ __attribute__((aligned(16))) uint8_t sl[kTiles * kTiles * kBins];
    __shared__ uint8_t norm[kBins];
    const int frame = blockIdx.z, tid = threadIdx.x;
    {
        const uint4* src = reinterpret_cast<const uint4*>(lut + (long long)frame * kTiles * kTiles * kBins);
        uint4* dst = reinterpret_cast<uint4*>(sl);
        for (int i = tid; i < kTiles * kTiles * kBins / 16; i += 256) dst[i] = src[i];
    }
    norm[tid] = do_normalize ? normalize_value(tid, mm[2 * frame], mm[2 * frame + 1]) : (uint8_t)tid;
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
        const int v = norm[src[(long long)y * in.stride + x]];
        const float l11 = sl[(ty1 * kTiles + tx1) * kBins + v], l12 = sl[(ty1 * kTiles + tx2) * kBins + v];
        const float l21 = sl[(ty2 * kTiles + tx1) * kBins + v], l22 = sl[(ty2 * kTiles + tx2) * kBins + v];
        const float top = __fmul_rn(__fadd_rn(__fmul_rn(l11, xa1), __fmul_rn(l12, xa)), ya1);
        const float bot = __fmul_rn(__fadd_rn(__fmul_rn(l21, xa1), __fmul_rn(l22, xa)), ya);
        dst[(long long)y * in.width + x] = sat_u8_rint(__fadd_rn(top, bot));
    }
}

}  // namespace

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
    if (do_normalize) {
        hipLaunchKernelGGL(minmax_init_kernel, dim3((nframes + 255) / 256), dim3(256), 0, s, mm, nframes);
        const int rpb = 32;
        hipLaunchKernelGGL(minmax_kernel, dim3((in.height + rpb - 1) / rpb, 1, nframes), dim3(256), 0, s, in, mm, rpb);
    }
    hipMemsetAsync(hist, 0, (size_t)nframes * kTiles * kTiles * kBins * sizeof(int), s);
    {
        const int rpb = 64;
        hipLaunchKernelGGL(clahe_hist_kernel, dim3((g.th + rpb - 1) / rpb, kTiles * kTiles, nframes), dim3(256), 0, s,
                           in, mm, g, hist, rpb, do_normalize ? 1 : 0);
    }
    hipLaunchKernelGGL(clahe_lut_kernel, dim3(kTiles * kTiles, nframes), dim3(256), 0, s, hist, g, clip, lut_scale, lut);
    {
        const int rpb = 32;
        hipLaunchKernelGGL(clahe_apply_kernel, dim3((in.width + 255) / 256, (in.height + rpb - 1) / rpb, nframes),
                           dim3(256), 0, s, in, mm, g, lut, out, rpb, do_normalize ? 1 : 0);
    }
    return true;
}

}  // namespace mrg
