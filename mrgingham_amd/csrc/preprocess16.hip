// 16-bit input of the command-line tool (mrgingham-from-image.cc:85-92): cv::normalize(0, 65535, NORM_MINMAX)
// -> CLAHE(8) on 16 bits (65 536 bins per tile) -> convertTo(CV_8U, 255/65535).  OpenCV arithmetic, restated
// from its published algorithm like the 8-bit path in preprocess.hip (parity unpinned); the HIP kernels
// here equal oracle_preprocess16 bit for bit.  A 65 536-bin histogram does not fit LDS: the tile
// histograms live in global memory (64 tiles x 256 KB per frame, L2-resident atomics), the rest is the
// same three steps -- histogram, clipped cumulative LUT, bilinear blend of the four tile LUTs.
// Not a throughput path (16-bit calibration images arrive one at a time); kept simple.
#include "common.h"
#include "kernels.h"

namespace mrg {

namespace {

constexpr int kTiles16 = 8, kBins16 = 65536;

__device__ __forceinline__ int reflect101_16(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ unsigned sat_rint(float v, float hi) {  // saturate_cast<>(cvRound(v)): round half to even
    const float r = __builtin_rintf(v);
    return (unsigned)(r < 0.f ? 0.f : (r > hi ? hi : r));
}

struct Geom16 {
    int w, h, ew, eh, tw, th;
};

__global__ __launch_bounds__(256) void minmax16_kernel(const uint16_t* in, long long pitch, int w, int h, int stride,
                                                       unsigned* mm) {
    const int f = blockIdx.y;
    const uint16_t* img = in + (long long)f * pitch;
    unsigned lo = 65535u, hi = 0u;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)w * h; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        const unsigned v = img[(long long)y * stride + x];
        lo = min(lo, v);
        hi = max(hi, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (unsigned)__shfl_xor((int)lo, o));
        hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(mm + 2 * f, lo);
        atomicMax(mm + 2 * f + 1, hi);
    }
}

// cv::normalize(.., 0, 65535, NORM_MINMAX) = convertTo(CV_16U, scale, shift) with single-precision
// multiply and add (no contraction) and cvRound; dense output
__global__ __launch_bounds__(256) void normalize16_kernel(const uint16_t* in, long long pitch, int w, int h, int stride,
                                                          const unsigned* mm, uint16_t* out) {
    const int f = blockIdx.y;
    const double smin = mm[2 * f], smax = mm[2 * f + 1];
    const double scale = 65535.0 * (smax - smin > 2.220446049250313e-16 ? 1.0 / (smax - smin) : 0.0);
    const double shift = 0.0 - smin * scale;
    const float a = (float)scale, b = (float)shift;
    const uint16_t* img = in + (long long)f * pitch;
    uint16_t* o = out + (long long)f * w * h;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)w * h; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        const float prod = (float)img[(long long)y * stride + x] * a;
        o[i] = (uint16_t)sat_rint(prod + b, 65535.f);
    }
}

__global__ __launch_bounds__(256) void hist16_kernel(const uint16_t* img, Geom16 g, unsigned* hist) {
    const int f = blockIdx.z, tile = blockIdx.y, ty = tile / kTiles16, tx = tile % kTiles16;
    const uint16_t* im = img + (long long)f * g.w * g.h;
    unsigned* hh = hist + ((long long)f * kTiles16 * kTiles16 + tile) * kBins16;
    const long long area = (long long)g.tw * g.th;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < area; i += (long long)gridDim.x * 256) {
        const int yy = (int)(i / g.tw), xx = (int)(i - (long long)yy * g.tw);
        const int y = reflect101_16(ty * g.th + yy, g.h), x = reflect101_16(tx * g.tw + xx, g.w);
        __hip_atomic_fetch_add(hh + im[(long long)y * g.w + x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// clip, redistribute, cumulative LUT of one tile; one workgroup, thread t owns bins [256 t, 256 t + 256)
__global__ __launch_bounds__(256) void lut16_kernel(const unsigned* hist, int clip, float lut_scale, uint16_t* lut) {
    __shared__ long long part[256];
    const long long tile = (long long)blockIdx.y * kTiles16 * kTiles16 + blockIdx.x;
    const unsigned* hh = hist + tile * kBins16;
    uint16_t* tl = lut + tile * kBins16;
    const int t = threadIdx.x, b0 = t * 256;
    long long clipped = 0;
    if (clip > 0)
        for (int i = 0; i < 256; ++i) {
            const long long v = hh[b0 + i];
            if (v > clip) clipped += v - clip;
        }
    part[t] = clipped;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (t < o) part[t] += part[t + o];
        __syncthreads();
    }
    const long long total_clipped = part[0];
    __syncthreads();
    const long long batch = total_clipped / kBins16;
    const int residual = (int)(total_clipped - batch * kBins16);
    const int step = residual != 0 ? max(kBins16 / residual, 1) : 1;
    auto adj = [&](int i) -> long long {
        long long v = hh[i];
        if (clip > 0) {
            if (v > clip) v = clip;
            v += batch;
            if (residual != 0 && i % step == 0 && i / step < residual) v += 1;
        }
        return v;
    };
    long long local = 0;
    for (int i = 0; i < 256; ++i) local += adj(b0 + i);
    part[t] = local;
    __syncthreads();
    if (t == 0) {
        long long run = 0;
        for (int k = 0; k < 256; ++k) { const long long v = part[k]; part[k] = run; run += v; }
    }
    __syncthreads();
    long long sum = part[t];
    for (int i = 0; i < 256; ++i) {
        sum += adj(b0 + i);
        tl[b0 + i] = (uint16_t)sat_rint((float)sum * lut_scale, 65535.f);
    }
}

// bilinear blend of the four tile LUTs (OpenCV's exact single-precision expression), then
// convertTo(CV_8U, 255/65535)
__global__ __launch_bounds__(256) void apply16_kernel(const uint16_t* img, Geom16 g, const uint16_t* lut, uint8_t* out) {
    const int f = blockIdx.z;
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= g.w) return;
    const float inv_tw = 1.0f / (float)g.tw, inv_th = 1.0f / (float)g.th;
    const float tyf = (float)y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0);
    ty2 = min(ty2, kTiles16 - 1);
    const float txf = (float)x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0);
    tx2 = min(tx2, kTiles16 - 1);
    const int v = img[((long long)f * g.h + y) * g.w + x];
    const uint16_t* lf = lut + (long long)f * kTiles16 * kTiles16 * kBins16;
    const float l11 = lf[(long long)(ty1 * kTiles16 + tx1) * kBins16 + v], l12 = lf[(long long)(ty1 * kTiles16 + tx2) * kBins16 + v];
    const float l21 = lf[(long long)(ty2 * kTiles16 + tx1) * kBins16 + v], l22 = lf[(long long)(ty2 * kTiles16 + tx2) * kBins16 + v];
    const float p11 = l11 * xa1, p12 = l12 * xa, p21 = l21 * xa1, p22 = l22 * xa;
    const float top = (p11 + p12) * ya1, bot = (p21 + p22) * ya;
    const unsigned r16 = sat_rint(top + bot, 65535.f);
    out[((long long)f * g.h + y) * g.w + x] = (uint8_t)sat_rint((float)r16 * (float)(255. / 65535.), 255.f);
}

__global__ __launch_bounds__(256) void convert16to8_kernel(const uint16_t* in, long long pitch, int w, int h, int stride,
                                                           uint8_t* out) {
    const int f = blockIdx.y;
    const uint16_t* img = in + (long long)f * pitch;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < (long long)w * h; i += (long long)gridDim.x * 256) {
        const int y = (int)(i / w), x = (int)(i - (long long)y * w);
        out[(long long)f * w * h + i] = (uint8_t)sat_rint((float)img[(long long)y * stride + x] * (float)(255. / 65535.), 255.f);
    }
}

}  // namespace

size_t preprocess16_scratch_bytes(int nframes, int w, int h) {
    const size_t nf = (size_t)nframes;
    return 256 + nf * 8 + nf * (size_t)w * h * 2 + nf * kTiles16 * kTiles16 * (size_t)kBins16 * (4 + 2) + 64;
}

// frames: device, uint16, `stride` and `pitch` in ELEMENTS.  out8: dense width x height bytes per frame.
bool launch_preprocess16(const uint16_t* frames, long long pitch, int nframes, int w, int h, int stride, bool do_clahe,
                         double clip_limit, uint8_t* out8, void* scratch, hipStream_t s) {
    if (nframes <= 0 || w <= 0 || h <= 0) return true;
    const long long nb = ((long long)w * h + 255) / 256;
    const int blocks = (int)(nb < 2048 ? nb : 2048);
    if (!do_clahe) {
        hipLaunchKernelGGL(convert16to8_kernel, dim3(blocks, nframes), dim3(256), 0, s, frames, pitch, w, h, stride, out8);
        return true;
    }
    Geom16 g{w, h, w, h, 0, 0};
    if (w % kTiles16 != 0 || h % kTiles16 != 0) {
        g.ew = w + (kTiles16 - w % kTiles16);
        g.eh = h + (kTiles16 - h % kTiles16);
    }
    g.tw = g.ew / kTiles16;
    g.th = g.eh / kTiles16;
    if (g.tw <= 0 || g.th <= 0) return false;
    const long long area = (long long)g.tw * g.th;
    const float lut_scale = (float)(kBins16 - 1) / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * (double)area / kBins16);
        if (clip < 1) clip = 1;
    }
    char* p = (char*)scratch;
    unsigned* mm = (unsigned*)p;                      p += (((size_t)nframes * 8 + 255) / 256) * 256;
    uint16_t* norm = (uint16_t*)p;                    p += (((size_t)nframes * w * h * 2 + 255) / 256) * 256;
    unsigned* hist = (unsigned*)p;                    p += (size_t)nframes * kTiles16 * kTiles16 * kBins16 * 4;
    uint16_t* lut = (uint16_t*)p;
    // extrema start at (65535, 0): 0x0000ffff then 0 per frame
    hipMemsetAsync(mm, 0, (size_t)nframes * 8, s);
    hipMemset2DAsync(mm, 8, 0xff, 2, nframes, s);   // low two bytes of every minimum word
    hipMemsetAsync(hist, 0, (size_t)nframes * kTiles16 * kTiles16 * kBins16 * 4, s);
    hipLaunchKernelGGL(minmax16_kernel, dim3(blocks, nframes), dim3(256), 0, s, frames, pitch, w, h, stride, mm);
    hipLaunchKernelGGL(normalize16_kernel, dim3(blocks, nframes), dim3(256), 0, s, frames, pitch, w, h, stride, mm, norm);
    const int hb = (int)((area + 255) / 256 < 256 ? (area + 255) / 256 : 256);
    hipLaunchKernelGGL(hist16_kernel, dim3(hb, kTiles16 * kTiles16, nframes), dim3(256), 0, s, norm, g, hist);
    hipLaunchKernelGGL(lut16_kernel, dim3(kTiles16 * kTiles16, nframes), dim3(256), 0, s, hist, clip, lut_scale, lut);
    hipLaunchKernelGGL(apply16_kernel, dim3((w + 255) / 256, h, nframes), dim3(256), 0, s, norm, g, lut, out8);
    return true;
}

}  // namespace mrg
