/*
 * mrgingham_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the reference's chessboard-
 * corner candidate path.  It exists so that tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg can check / time the HIP product against
 * the reference's arithmetic.  Nothing under mrgingham_amd/ may include, link
 * or call this file: the product path is HIP only and fails loudly when the
 * HIP library is missing.
 *
 * Every function cites the reference file:line (paths relative to the
 * upstream dkogan/mrgingham tree) whose behaviour it restates.  This is a
 * restatement written from the algorithm, not a copy of the sources.
 *
 * PARITY PIN STATUS
 *   - oracle_chess_response_5: PINNED.  oracle/Makefile compiles the upstream
 *     ChESS.c where it lies (never copied) into oracle/_ref/libchess_ref.so,
 *     tests/test_oracle.py checks this restatement against it bit-for-bit, and
 *     tests/golden/chess_kat.npz holds known-answer vectors produced by that
 *     real reference build (tests/golden/make_golden.py).
 *   - connected components / detect / refine (find_chessboard_corners.cc):
 *     PARITY UNPINNED.  The upstream file needs OpenCV headers, which this
 *     image lacks, and the upstream project holds no test, golden vector or
 *     fixture for this path, so the restatement cannot be
 *     executed against the real thing here.  It follows the upstream logic
 *     statement by statement (citations below).
 *   - level decimation (cv::resize INTER_LINEAR) and box blur (cv::blur):
 *     PARITY UNPINNED.  The arithmetic lives in OpenCV (un-vendored, version
 *     unpinned by upstream: "opencv >= 3.2", packaging/mrgingham.spec:12).  The
 *     functions below restate OpenCV's published 8-bit algorithms
 *     (modules/imgproc/src/resize.cpp: area-fast 2x2 path and the fixed-point
 *     bilinear path with INTER_RESIZE_COEF_BITS = 11;
 *     modules/imgproc/src/box_filter: normalised box, BORDER_REFLECT_101).
 */
#include "mrgingham_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* ChESS radius-5 response.  Restates ChESS.c:56-106.                        */
/* ------------------------------------------------------------------------- */

/* Ring sample k sits at (RING_DX[k], RING_DY[k]) from the centre pixel
 * (ChESS.c:68-83).  Samples k, k+4, k+8, k+12 form one quadruple. */
static const int RING_DX[16] = {+2, 0, -2, -4, -5, -5, -5, -4, -2, 0, +2, +4, +5, +5, +5, +4};
static const int RING_DY[16] = {-5, -5, -5, -4, -2, 0, +2, +4, +5, +5, +5, +4, +2, 0, -2, -4};

void oracle_chess_response_5(int16_t* response, const uint8_t* image, int w, int h, int stride)
{
    /* interior only: ChESS.c:62-63.  Pixels outside are NOT written. */
    for (int y = 7; y < h - 7; y++)
        for (int x = 7; x < w - 7; x++) {
            const uint8_t* p = image + (ptrdiff_t)y * stride + x;
            int s[16];
            for (int k = 0; k < 16; k++)
                s[k] = p[(ptrdiff_t)RING_DY[k] * stride + RING_DX[k]];

            /* ChESS.c:86 -- truncating integer division, result fits uint16 */
            const int local_mean = ((int)p[-1] + (int)p[0] + (int)p[1]) * 16 / 3;

            int sum_response = 0, diff_response = 0, mean = 0;
            for (int i = 0; i < 4; i++) { /* ChESS.c:93-102 */
                const int a = s[i], b = s[i + 4], c = s[i + 8], d = s[i + 12];
                sum_response += abs(a - b + c - d);
                diff_response += abs(a - c) + abs(b - d);
                mean += a + b + c + d;
            }
            /* ChESS.c:104 */
            response[(ptrdiff_t)y * w + x] =
                (int16_t)(sum_response - diff_response - abs(mean - local_mean));
        }
}

/* ------------------------------------------------------------------------- */
/* Level decimation.  Restates what find_chessboard_corners.cc:445-452 asks  */
/* of cv::resize(src, dst, Size(), 1/2^L, 1/2^L, INTER_LINEAR) for CV_8UC1.  */
/* ------------------------------------------------------------------------- */

/* cvRound(): round half to even (lrint in the default rounding mode). */
static int round_half_even(double v) { return (int)lrint(v); }

int oracle_level_dims(int W, int H, int level, int* w, int* h)
{
    if (level < 0 || level > 10) return -1; /* find_chessboard_corners.cc:433-441 */
    const double inv = 1.0 / (double)(1 << level);
    /* resize(): dsize = Size(saturate_cast<int>(ssize.width*inv_scale_x), ...) */
    *w = round_half_even((double)W * inv);
    *h = round_half_even((double)H * inv);
    return 0;
}

/* clip(x, 0, n): OpenCV's row clamp in resizeGeneric_ */
static int clip_idx(int x, int n) { return x < 0 ? 0 : (x >= n ? n - 1 : x); }

int oracle_decimate(uint8_t* out, const uint8_t* in, int W, int H, int stride, int level)
{
    int ow, oh;
    if (oracle_level_dims(W, H, level, &ow, &oh) != 0) return -1;
    if (level == 0) {
        for (int y = 0; y < H; y++) memcpy(out + (size_t)y * W, in + (size_t)y * stride, (size_t)W);
        return 0;
    }
    const int s = 1 << level;
    if (s == 2) {
        /* resize(): INTER_LINEAR with an exact 2x2 integer shrink is rerouted to
         * INTER_AREA; ResizeAreaFastVec computes (a+b+c+d+2)>>2 for whole cells,
         * resizeAreaFast_Invoker averages the partial cells at a ragged edge
         * with saturate_cast<uchar>((float)sum/count). */
        const int full_w = W / 2;
        for (int dy = 0; dy < oh; dy++) {
            uint8_t* D = out + (size_t)dy * ow;
            const int sy0 = dy * 2;
            if (sy0 >= H) { memset(D, 0, (size_t)ow); continue; }
            const int wfull = (sy0 + 2 <= H) ? full_w : 0;
            const uint8_t* S0 = in + (size_t)sy0 * stride;
            const uint8_t* S1 = S0 + stride;
            int dx = 0;
            for (; dx < wfull && dx < ow; dx++)
                D[dx] = (uint8_t)((S0[2 * dx] + S0[2 * dx + 1] + S1[2 * dx] + S1[2 * dx + 1] + 2) >> 2);
            for (; dx < ow; dx++) {
                int sum = 0, count = 0;
                const int sx0 = 2 * dx;
                for (int sy = 0; sy < 2 && sy0 + sy < H; sy++)
                    for (int sx = 0; sx < 2 && sx0 + sx < W; sx++) {
                        sum += in[(size_t)(sy0 + sy) * stride + sx0 + sx];
                        count++;
                    }
                D[dx] = count ? (uint8_t)lrintf((float)sum / (float)count) : 0;
            }
        }
        return 0;
    }

    /* True bilinear, 8-bit fixed point (INTER_RESIZE_COEF_BITS = 11).
     * Horizontal pass keeps ints scaled by 2048, vertical pass is
     * VResizeLinear<uchar,int,short,...>:
     *   dst = (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2            */
    const double scale = (double)s;
    int* xofs = (int*)malloc(sizeof(int) * (size_t)ow);
    short* xa = (short*)malloc(sizeof(short) * 2 * (size_t)ow);
    int* row0 = (int*)malloc(sizeof(int) * (size_t)ow);
    int* row1 = (int*)malloc(sizeof(int) * (size_t)ow);
    if (!xofs || !xa || !row0 || !row1) { free(xofs); free(xa); free(row0); free(row1); return -1; }
    for (int dx = 0; dx < ow; dx++) {
        float fx = (float)((dx + 0.5) * scale - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= W - 1) { fx = 0.f; sx = W - 1; }
        xofs[dx] = sx;
        xa[2 * dx + 0] = (short)lrintf((1.f - fx) * 2048.f);
        xa[2 * dx + 1] = (short)lrintf(fx * 2048.f);
    }
    for (int dy = 0; dy < oh; dy++) {
        float fy = (float)((dy + 0.5) * scale - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        const int b0 = (short)lrintf((1.f - fy) * 2048.f);
        const int b1 = (short)lrintf(fy * 2048.f);
        const uint8_t* R0 = in + (size_t)clip_idx(sy, H) * stride;
        const uint8_t* R1 = in + (size_t)clip_idx(sy + 1, H) * stride;
        for (int dx = 0; dx < ow; dx++) {
            const int sx = xofs[dx];
            const int sx1 = sx + 1 < W ? sx + 1 : sx; /* weight is 0 there */
            row0[dx] = R0[sx] * xa[2 * dx] + R0[sx1] * xa[2 * dx + 1];
            row1[dx] = R1[sx] * xa[2 * dx] + R1[sx1] * xa[2 * dx + 1];
        }
        uint8_t* D = out + (size_t)dy * ow;
        for (int dx = 0; dx < ow; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(xa); free(row0); free(row1);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Box blur the CLI applies before the detector.                              */
/* Restates mrgingham-from-image.cc:106-111 -> cv::blur(img, img,            */
/* Size(1+2r,1+2r)), default border BORDER_REFLECT_101, 8-bit rounding.      */
/* ------------------------------------------------------------------------- */
static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * (n - 1) - i;
    }
    return i;
}

void oracle_box_blur(uint8_t* out, const uint8_t* in, int w, int h, int stride, int radius)
{
    const int k = 2 * radius + 1, area = k * k;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int sum = 0;
            for (int dy = -radius; dy <= radius; dy++) {
                const uint8_t* row = in + (size_t)reflect101(y + dy, h) * stride;
                for (int dx = -radius; dx <= radius; dx++) sum += row[reflect101(x + dx, w)];
            }
            /* round to nearest; area is odd so ties cannot occur */
            out[(size_t)y * w + x] = (uint8_t)((sum + area / 2) / area);
        }
}

/* ------------------------------------------------------------------------- */
/* The CLI's contrast preprocessing (mrgingham-from-image.cc:38-45, :71-79):   */
/*   cv::normalize(image, image, 0, 255, NORM_MINMAX); clahe->apply(image)     */
/* with cv::createCLAHE() defaults (8x8 tiles) and setClipLimit(8).            */
/* PARITY UNPINNED: the arithmetic is OpenCV's (un-vendored, version unpinned  */
/* upstream, and its convertTo has FMA-dispatching SIMD paths).  Restated from  */
/* OpenCV's published algorithm (modules/core norm.cpp / convert_scale,         */
/* modules/imgproc clahe.cpp), single-precision, unfused multiply-add,          */
/* round-half-even.                                                             */
/* ------------------------------------------------------------------------- */
#define CLAHE_TILES 8
#define CLAHE_BINS 256

static uint8_t sat_u8_rint(float v)
{
    const long r = lrintf(v); /* cvRound: round half to even (default rounding mode) */
    return (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
}

/* cv::normalize(src, dst, 0, 255, NORM_MINMAX) for CV_8U: scale/shift in double, then
 * convertTo's float multiply-add per pixel.  lut[v] = normalised value of v. */
void oracle_normalize_lut(uint8_t lut[256], int vmin, int vmax)
{
    const double smin = vmin, smax = vmax, dmin = 0., dmax = 255.;
    const double scale = (dmax - dmin) * (smax - smin > 2.220446049250313e-16 ? 1. / (smax - smin) : 0.);
    const double shift = dmin - smin * scale;
    const float a = (float)scale, b = (float)shift;
    for (int v = 0; v < 256; v++) {
        const float prod = (float)v * a; /* kept as its own rounding step: no fused multiply-add */
        lut[v] = sat_u8_rint(prod + b);
    }
}

void oracle_normalize_minmax(uint8_t* out, const uint8_t* in, int w, int h, int stride)
{
    int vmin = 255, vmax = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int v = in[(size_t)y * stride + x];
            if (v < vmin) vmin = v;
            if (v > vmax) vmax = v;
        }
    uint8_t lut[256];
    oracle_normalize_lut(lut, vmin, vmax);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) out[(size_t)y * w + x] = lut[in[(size_t)y * stride + x]];
}

/* CLAHE, 8-bit, tiles 8x8.  clip_limit is the user-facing value (8 in the CLI). */
int oracle_clahe(uint8_t* out, const uint8_t* in, int w, int h, int stride, double clip_limit)
{
    /* frames that do not divide into the tile grid are extended to the right and below with
     * BORDER_REFLECT_101 by (tiles - size % tiles) -- on BOTH axes as soon as either needs it */
    int ew = w, eh = h;
    if (w % CLAHE_TILES != 0 || h % CLAHE_TILES != 0) {
        ew = w + (CLAHE_TILES - w % CLAHE_TILES);
        eh = h + (CLAHE_TILES - h % CLAHE_TILES);
    }
    const int tw = ew / CLAHE_TILES, th = eh / CLAHE_TILES, area = tw * th;
    if (tw <= 0 || th <= 0) return -1;
    const float lut_scale = (float)(CLAHE_BINS - 1) / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * area / CLAHE_BINS);
        if (clip < 1) clip = 1;
    }
    uint8_t* lut = (uint8_t*)malloc((size_t)CLAHE_TILES * CLAHE_TILES * CLAHE_BINS);
    if (!lut) return -1;
    for (int ty = 0; ty < CLAHE_TILES; ty++)
        for (int tx = 0; tx < CLAHE_TILES; tx++) {
            int hist[CLAHE_BINS] = {0};
            for (int y = ty * th; y < (ty + 1) * th; y++) {
                const uint8_t* row = in + (size_t)reflect101(y, h) * stride;
                for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[row[reflect101(x, w)]]++;
            }
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < CLAHE_BINS; i++)
                    if (hist[i] > clip) {
                        clipped += hist[i] - clip;
                        hist[i] = clip;
                    }
                const int batch = clipped / CLAHE_BINS;
                int residual = clipped - batch * CLAHE_BINS;
                for (int i = 0; i < CLAHE_BINS; i++) hist[i] += batch;
                if (residual != 0) {
                    const int step = CLAHE_BINS / residual > 1 ? CLAHE_BINS / residual : 1;
                    for (int i = 0; i < CLAHE_BINS && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            uint8_t* tl = lut + (size_t)(ty * CLAHE_TILES + tx) * CLAHE_BINS;
            int sum = 0;
            for (int i = 0; i < CLAHE_BINS; i++) {
                sum += hist[i];
                tl[i] = sat_u8_rint((float)sum * lut_scale);
            }
        }
    /* bilinear interpolation between the four surrounding tile LUTs (tile centres) */
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < h; y++) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > CLAHE_TILES - 1) ty2 = CLAHE_TILES - 1;
        for (int x = 0; x < w; x++) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > CLAHE_TILES - 1) tx2 = CLAHE_TILES - 1;
            const int v = in[(size_t)y * stride + x];
            const float l11 = lut[(size_t)(ty1 * CLAHE_TILES + tx1) * CLAHE_BINS + v];
            const float l12 = lut[(size_t)(ty1 * CLAHE_TILES + tx2) * CLAHE_BINS + v];
            const float l21 = lut[(size_t)(ty2 * CLAHE_TILES + tx1) * CLAHE_BINS + v];
            const float l22 = lut[(size_t)(ty2 * CLAHE_TILES + tx2) * CLAHE_BINS + v];
            const float p11 = l11 * xa1, p12 = l12 * xa, p21 = l21 * xa1, p22 = l22 * xa;
            const float top = (p11 + p12) * ya1, bot = (p21 + p22) * ya;
            out[(size_t)y * w + x] = sat_u8_rint(top + bot);
        }
    }
    free(lut);
    return 0;
}

/* ---- the CLI's 16-bit branch (mrgingham-from-image.cc:85-92): normalize to 0..65535, CLAHE on 16 bits
 * (OpenCV's CLAHE_CalcLut_Body<ushort, 65536, 0> / CLAHE_Interpolation_Body<ushort, 0>: the 8-bit
 * algorithm with 65536 bins), convertTo(CV_8U, 255/65535).  stride in ELEMENTS.  Parity unpinned. ---- */
static uint16_t sat_u16_rint(float v)
{
    const float r = rintf(v);
    return (uint16_t)(r < 0.f ? 0.f : (r > 65535.f ? 65535.f : r));
}

void oracle_normalize16(uint16_t* out, const uint16_t* in, int w, int h, int stride)
{
    int vmin = 65535, vmax = 0;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const int v = in[(size_t)y * stride + x];
            if (v < vmin) vmin = v;
            if (v > vmax) vmax = v;
        }
    const double smin = vmin, smax = vmax;
    const double scale = 65535.0 * (smax - smin > 2.220446049250313e-16 ? 1. / (smax - smin) : 0.);
    const double shift = 0.0 - smin * scale;
    const float a = (float)scale, b = (float)shift;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            const float prod = (float)in[(size_t)y * stride + x] * a; /* its own rounding step: no fused multiply-add */
            out[(size_t)y * w + x] = sat_u16_rint(prod + b);
        }
}

int oracle_clahe16(uint16_t* out, const uint16_t* in, int w, int h, int stride, double clip_limit)
{
    enum { BINS = 65536 };
    int ew = w, eh = h;
    if (w % CLAHE_TILES != 0 || h % CLAHE_TILES != 0) {
        ew = w + (CLAHE_TILES - w % CLAHE_TILES);
        eh = h + (CLAHE_TILES - h % CLAHE_TILES);
    }
    const int tw = ew / CLAHE_TILES, th = eh / CLAHE_TILES;
    if (tw <= 0 || th <= 0) return -1;
    const long long area = (long long)tw * th;
    const float lut_scale = (float)(BINS - 1) / (float)area;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * (double)area / BINS);
        if (clip < 1) clip = 1;
    }
    uint16_t* lut = (uint16_t*)malloc((size_t)CLAHE_TILES * CLAHE_TILES * BINS * sizeof(uint16_t));
    int* hist = (int*)malloc((size_t)BINS * sizeof(int));
    if (!lut || !hist) { free(lut); free(hist); return -1; }
    for (int ty = 0; ty < CLAHE_TILES; ty++)
        for (int tx = 0; tx < CLAHE_TILES; tx++) {
            memset(hist, 0, (size_t)BINS * sizeof(int));
            for (int y = ty * th; y < (ty + 1) * th; y++) {
                const uint16_t* row = in + (size_t)reflect101(y, h) * stride;
                for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[row[reflect101(x, w)]]++;
            }
            if (clip > 0) {
                long long clipped = 0;
                for (int i = 0; i < BINS; i++)
                    if (hist[i] > clip) {
                        clipped += hist[i] - clip;
                        hist[i] = clip;
                    }
                const int batch = (int)(clipped / BINS);
                int residual = (int)(clipped - (long long)batch * BINS);
                for (int i = 0; i < BINS; i++) hist[i] += batch;
                if (residual != 0) {
                    const int step = BINS / residual > 1 ? BINS / residual : 1;
                    for (int i = 0; i < BINS && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            uint16_t* tl = lut + (size_t)(ty * CLAHE_TILES + tx) * BINS;
            long long sum = 0;
            for (int i = 0; i < BINS; i++) {
                sum += hist[i];
                tl[i] = sat_u16_rint((float)sum * lut_scale);
            }
        }
    const float inv_tw = 1.0f / (float)tw, inv_th = 1.0f / (float)th;
    for (int y = 0; y < h; y++) {
        const float tyf = (float)y * inv_th - 0.5f;
        int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
        const float ya = tyf - (float)ty1, ya1 = 1.0f - ya;
        if (ty1 < 0) ty1 = 0;
        if (ty2 > CLAHE_TILES - 1) ty2 = CLAHE_TILES - 1;
        for (int x = 0; x < w; x++) {
            const float txf = (float)x * inv_tw - 0.5f;
            int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
            const float xa = txf - (float)tx1, xa1 = 1.0f - xa;
            if (tx1 < 0) tx1 = 0;
            if (tx2 > CLAHE_TILES - 1) tx2 = CLAHE_TILES - 1;
            const int v = in[(size_t)y * stride + x];
            const float l11 = lut[(size_t)(ty1 * CLAHE_TILES + tx1) * BINS + v];
            const float l12 = lut[(size_t)(ty1 * CLAHE_TILES + tx2) * BINS + v];
            const float l21 = lut[(size_t)(ty2 * CLAHE_TILES + tx1) * BINS + v];
            const float l22 = lut[(size_t)(ty2 * CLAHE_TILES + tx2) * BINS + v];
            const float p11 = l11 * xa1, p12 = l12 * xa, p21 = l21 * xa1, p22 = l22 * xa;
            const float top = (p11 + p12) * ya1, bot = (p21 + p22) * ya;
            out[(size_t)y * w + x] = sat_u16_rint(top + bot);
        }
    }
    free(lut);
    free(hist);
    return 0;
}

/* mrgingham-from-image.cc:85-111 for a 16-bit frame -> the 8-bit image the detector sees */
int oracle_preprocess16(uint8_t* out, const uint16_t* in, int w, int h, int stride, int do_clahe, int blur_radius)
{
    const size_t n = (size_t)w * h;
    uint16_t* a = (uint16_t*)malloc(n * 2 + 2);
    uint16_t* b = (uint16_t*)malloc(n * 2 + 2);
    uint8_t* c = (uint8_t*)malloc(n + 1);
    if (!a || !b || !c) { free(a); free(b); free(c); return -1; }
    const uint16_t* src = in;
    int sstride = stride;
    if (do_clahe) {
        oracle_normalize16(a, in, w, h, stride);
        if (oracle_clahe16(b, a, w, h, w, 8.0) != 0) { free(a); free(b); free(c); return -1; }
        src = b;
        sstride = w;
    }
    const float k = (float)(255. / 65535.);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) c[(size_t)y * w + x] = sat_u8_rint((float)src[(size_t)y * sstride + x] * k);
    if (blur_radius > 0) oracle_box_blur(out, c, w, h, w, blur_radius);
    else memcpy(out, c, n);
    free(a); free(b); free(c);
    return 0;
}

/* mrgingham-from-image.cc:71-111 for an 8-bit frame: [normalize + CLAHE(8)] then box blur. */
int oracle_preprocess(uint8_t* out, const uint8_t* in, int w, int h, int stride, int do_clahe, int blur_radius)
{
    uint8_t* a = (uint8_t*)malloc((size_t)w * h);
    uint8_t* b = (uint8_t*)malloc((size_t)w * h);
    if (!a || !b) { free(a); free(b); return -1; }
    const uint8_t* cur = in;
    int cur_stride = stride;
    if (do_clahe) {
        oracle_normalize_minmax(a, in, w, h, stride);
        if (oracle_clahe(b, a, w, h, w, 8.0) != 0) { free(a); free(b); return -1; }
        cur = b;
        cur_stride = w;
    }
    if (blur_radius > 0) oracle_box_blur(out, cur, w, h, cur_stride, blur_radius);
    else
        for (int y = 0; y < h; y++) memcpy(out + (size_t)y * w, cur + (size_t)y * cur_stride, (size_t)w);
    free(a);
    free(b);
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Connected components over the clamped response.                            */
/* Restates find_chessboard_corners.cc:18-44 (thresholds), :50-88             */
/* (high_variance), :91-141 (LIFO), :143-267 (fill), :269-280 (scaling),      */
/* :284-411 (detect / refine loops), :481-565 (glue).                         */
/* ------------------------------------------------------------------------- */
#define PEAK_MIN 120      /* RESPONSE_MIN_PEAK_THRESHOLD  :18 */
#define RESP_MIN 15       /* RESPONSE_MIN_THRESHOLD       :22 */
#define BLOB_MIN_PIXELS 2 /* CONNECTED_COMPONENT_MIN_SIZE :29 */
#define VAR_WINDOW_R 10   /* CONSTANCY_WINDOW_R           :38 */
#define VAR_MIN (20 * 20) /* STDEV_THRESHOLD^2            :39,:44 */
#define CHESS_MARGIN 7    /* :559-564 */
#define GRID_SCALE 1000   /* FIND_GRID_SCALE, mrgingham-internal.h:3 */

typedef struct { int16_t x, y; } pix_t;
typedef struct { pix_t* v; int n, cap; } lifo_t;

static void lifo_push(lifo_t* l, int16_t x, int16_t y)
{
    if (l->n == l->cap) {
        l->cap = l->cap ? 2 * l->cap : 128;
        l->v = (pix_t*)realloc(l->v, sizeof(pix_t) * (size_t)l->cap);
    }
    l->v[l->n].x = x;
    l->v[l->n].y = y;
    l->n++;
}

typedef struct {
    uint64_t sum_rx, sum_ry, sum_r; /* :145 */
    int npix;                       /* :146 */
    uint16_t x_peak, y_peak;        /* :155 */
    int16_t r_max;                  /* :156 */
} blob_t;

/* :159-171.  blob == NULL means "seed test": only the absolute threshold. */
static int pixel_ok(int16_t x, int16_t y, int16_t w, int16_t h, const int16_t* d, const blob_t* blob)
{
    if (x < 0 || x >= w || y < 0 || y >= h) return 0;
    const int16_t r = d[x + y * w];
    if (!(r > RESP_MIN)) return 0;
    if (blob == NULL) return 1;
    return r > (int16_t)(((uint16_t)blob->r_max) >> 4); /* :27 */
}

/* :50-88 */
static int window_variance_high(int16_t x, int16_t y, int16_t w, int16_t h, const uint8_t* image)
{
    if (x - VAR_WINDOW_R < 0 || x + VAR_WINDOW_R >= w || y - VAR_WINDOW_R < 0 || y + VAR_WINDOW_R >= h)
        return 0;
    const int n = (2 * VAR_WINDOW_R + 1) * (2 * VAR_WINDOW_R + 1);
    int32_t sum = 0;
    for (int dy = -VAR_WINDOW_R; dy <= VAR_WINDOW_R; dy++)
        for (int dx = -VAR_WINDOW_R; dx <= VAR_WINDOW_R; dx++) sum += image[x + dx + (y + dy) * w];
    const int32_t mean = sum / n;
    int32_t ssd = 0;
    for (int dy = -VAR_WINDOW_R; dy <= VAR_WINDOW_R; dy++)
        for (int dx = -VAR_WINDOW_R; dx <= VAR_WINDOW_R; dx++) {
            const int32_t dev = (int32_t)image[x + dx + (y + dy) * w] - mean;
            ssd += dev * dev;
        }
    return ssd / n > VAR_MIN;
}

/* :210-227 */
static void visit_neighbour(lifo_t* l, int* touched_margin, int16_t x, int16_t y, int16_t w, int16_t h,
                            const int16_t* d)
{
    if (!(x >= CHESS_MARGIN && x < w - CHESS_MARGIN && y >= CHESS_MARGIN && y < h - CHESS_MARGIN)) {
        *touched_margin = 1;
        return;
    }
    if (d[x + y * w] <= 0) return;
    lifo_push(l, x, y);
}

/* :228-267.  Drains the LIFO; returns 1 and the weighted centroid on accept. */
static int flood_blob(double* cx, double* cy, lifo_t* l, int16_t w, int16_t h, int16_t* d,
                      const uint8_t* image)
{
    blob_t b;
    memset(&b, 0, sizeof(b));
    int touched_margin = 0;

    while (l->n > 0) {
        l->n--;
        const int16_t x = l->v[l->n].x, y = l->v[l->n].y;
        if (!pixel_ok(x, y, w, h, d, &b)) {
            d[x + y * w] = 0; /* :245 */
            continue;
        }
        /* :172-186 */
        const int16_t r = d[x + y * w];
        if (r > b.r_max) {
            b.r_max = r;
            b.x_peak = (uint16_t)x;
            b.y_peak = (uint16_t)y;
        }
        b.sum_rx += (uint64_t)((int)r * (int)x);
        b.sum_ry += (uint64_t)((int)r * (int)y);
        b.sum_r += (uint64_t)r;
        b.npix++;
        d[x + y * w] = 0; /* :250 */

        visit_neighbour(l, &touched_margin, (int16_t)(x + 1), y, w, h, d); /* :252-255 */
        visit_neighbour(l, &touched_margin, (int16_t)(x - 1), y, w, h, d);
        visit_neighbour(l, &touched_margin, x, (int16_t)(y + 1), w, h, d);
        visit_neighbour(l, &touched_margin, x, (int16_t)(y - 1), w, h, d);
    }

    if (touched_margin) return 0;                 /* :259 */
    if (!(b.npix >= BLOB_MIN_PIXELS)) return 0;   /* :205 */
    if (!(b.r_max > PEAK_MIN)) return 0;          /* :206 */
    if (!window_variance_high((int16_t)b.x_peak, (int16_t)b.y_peak, w, h, image)) return 0; /* :207 */
    *cx = (double)b.sum_rx / (double)b.sum_r; /* :262-263 */
    *cy = (double)b.sum_ry / (double)b.sum_r;
    return 1;
}

/* :269-280: (-0.5,-0.5) is the fixed point of the level scaling */
static double rescale_coord(double p, double scale) { return (p + 0.5) * scale - 0.5; }

/* Steps shared by detect and refine (find_chessboard_corners.cc:495-529):
 * decimate, zeroed response, ChESS with stride = w, clamp negatives. */
static int prepare_level(uint8_t** img_out, int16_t** resp_out, int* w_out, int* h_out,
                         const uint8_t* image, int H, int W, int stride, int level)
{
    int w, h;
    if (oracle_level_dims(W, H, level, &w, &h) != 0) {
        fprintf(stderr, "oracle: unreasonable image_pyramid_level = %d\n", level);
        return -1;
    }
    /* :461-466: level 0 takes the caller's buffer as is and insists on stride == width
     * (cv::Mat::isContinuous(); a single-row Mat is always continuous) */
    if (level == 0 && stride != W && H != 1) {
        fprintf(stderr, "oracle: only continuous arrays (stride == width) are handled at level 0\n");
        return -1;
    }
    if (w > 32767 || h > 32767) return -1; /* int16 coordinates throughout */
    const size_t npx = (size_t)w * (size_t)h;
    uint8_t* img = (uint8_t*)malloc(npx ? npx : 1);
    int16_t* resp = (int16_t*)calloc(npx ? npx : 1, sizeof(int16_t)); /* :506 */
    if (!img || !resp) { free(img); free(resp); return -1; }
    if (oracle_decimate(img, image, W, H, stride, level) != 0) { free(img); free(resp); return -1; }
    oracle_chess_response_5(resp, img, w, h, w); /* :511 */
    for (size_t i = 0; i < (size_t)w * h; i++)   /* :527-529 */
        if (resp[i] < 0) resp[i] = 0;
    *img_out = img; *resp_out = resp; *w_out = w; *h_out = h;
    return 0;
}

int oracle_clamped_response(int16_t* resp_out, uint8_t* level_image_out, const uint8_t* image, int H, int W,
                            int stride, int level)
{
    uint8_t* img; int16_t* resp; int w, h;
    if (prepare_level(&img, &resp, &w, &h, image, H, W, stride, level) != 0) return -1;
    if (resp_out) memcpy(resp_out, resp, sizeof(int16_t) * (size_t)w * h);
    if (level_image_out) memcpy(level_image_out, img, (size_t)w * h);
    free(img); free(resp);
    return 0;
}

/* Detect.  Restates find_chessboard_corners_from_image_array
 * (find_chessboard_corners.cc:568-587) -> :481-565 -> :330-355. */
int oracle_find_corners(int32_t* xy_out, int cap, const uint8_t* image, int H, int W, int stride, int level)
{
    uint8_t* img; int16_t* d; int w, h;
    if (prepare_level(&img, &d, &w, &h, image, H, W, stride, level) != 0) return -1;

    const double scale = (double)(uint16_t)(1U << level); /* :319 */
    lifo_t l = {0};
    int n = 0;
    for (int16_t y = CHESS_MARGIN + 1; y < h - CHESS_MARGIN - 1; y++)     /* :332 */
        for (int16_t x = CHESS_MARGIN + 1; x < w - CHESS_MARGIN - 1; x++) /* :333 */
        {
            if (!pixel_ok(x, y, (int16_t)w, (int16_t)h, d, NULL)) continue;
            l.n = 0;
            lifo_push(&l, x, y); /* :338 */
            double cx, cy;
            if (!flood_blob(&cx, &cy, &l, (int16_t)w, (int16_t)h, d, img)) continue;
            const double px = rescale_coord(cx, scale), py = rescale_coord(cy, scale); /* :346 */
            if (n < cap) {
                xy_out[2 * n + 0] = (int32_t)(0.5 + px * GRID_SCALE); /* :350-351 */
                xy_out[2 * n + 1] = (int32_t)(0.5 + py * GRID_SCALE);
            }
            n++;
        }
    free(l.v); free(img); free(d);
    return n;
}

/* Refine.  Restates refine_chessboard_corners_from_image_array
 * (find_chessboard_corners.cc:591-619) -> :481-565 -> :356-397. */
int oracle_refine_corners(double* xy, signed char* level_of_point, int npoints, const uint8_t* image, int H,
                          int W, int stride, int level)
{
    uint8_t* img; int16_t* d; int w, h;
    if (prepare_level(&img, &d, &w, &h, image, H, W, stride, level) != 0) return 0; /* :498 */

    const uint16_t coord_scale = (uint16_t)(1U << level);
    lifo_t l = {0};
    int nrefined = 0;
    for (int i = 0; i < npoints; i++) {
        if (level_of_point[i] != level + 1) continue; /* :362 */
        const double lx = rescale_coord(xy[2 * i + 0], 1.0 / coord_scale); /* :369 */
        const double ly = rescale_coord(xy[2 * i + 1], 1.0 / coord_scale);
        const int x = (int)(lx + 0.5), y = (int)(ly + 0.5); /* :371-372 */
        l.n = 0;
        for (int dx = -1; dx <= 1; dx++) /* :379-382: dx outer, dy inner */
            for (int dy = -1; dy <= 1; dy++)
                if (pixel_ok((int16_t)(x + dx), (int16_t)(y + dy), (int16_t)w, (int16_t)h, d, NULL))
                    lifo_push(&l, (int16_t)(x + dx), (int16_t)(y + dy));
        double cx, cy;
        if (flood_blob(&cx, &cy, &l, (int16_t)w, (int16_t)h, d, img)) {
            xy[2 * i + 0] = rescale_coord(cx, (double)coord_scale); /* :390 */
            xy[2 * i + 1] = rescale_coord(cy, (double)coord_scale);
            level_of_point[i] = (signed char)level; /* :393 */
            nrefined++;
        }
    }
    free(l.v); free(img); free(d);
    return nrefined;
}

/* Same fill logic, but over a caller-supplied clamped response + level image
 * (no ChESS, no decimation).  Lets tests drive the connected-component rules
 * with hand-built adversarial responses. */
int oracle_cc_detect_on_response(int32_t* xy_out, int cap, int16_t* d, const uint8_t* level_image, int w,
                                 int h, int level)
{
    const double scale = (double)(uint16_t)(1U << level);
    lifo_t l = {0};
    int n = 0;
    for (int16_t y = CHESS_MARGIN + 1; y < h - CHESS_MARGIN - 1; y++)
        for (int16_t x = CHESS_MARGIN + 1; x < w - CHESS_MARGIN - 1; x++) {
            if (!pixel_ok(x, y, (int16_t)w, (int16_t)h, d, NULL)) continue;
            l.n = 0;
            lifo_push(&l, x, y);
            double cx, cy;
            if (!flood_blob(&cx, &cy, &l, (int16_t)w, (int16_t)h, d, level_image)) continue;
            const double px = rescale_coord(cx, scale), py = rescale_coord(cy, scale);
            if (n < cap) {
                xy_out[2 * n + 0] = (int32_t)(0.5 + px * GRID_SCALE);
                xy_out[2 * n + 1] = (int32_t)(0.5 + py * GRID_SCALE);
            }
            n++;
        }
    free(l.v);
    return n;
}

int oracle_cc_refine_on_response(double* xy, signed char* level_of_point, int npoints, int16_t* d,
                                 const uint8_t* level_image, int w, int h, int level)
{
    const uint16_t coord_scale = (uint16_t)(1U << level);
    lifo_t l = {0};
    int nrefined = 0;
    for (int i = 0; i < npoints; i++) {
        if (level_of_point[i] != level + 1) continue;
        const double lx = rescale_coord(xy[2 * i + 0], 1.0 / coord_scale);
        const double ly = rescale_coord(xy[2 * i + 1], 1.0 / coord_scale);
        const int x = (int)(lx + 0.5), y = (int)(ly + 0.5);
        l.n = 0;
        for (int dx = -1; dx <= 1; dx++)
            for (int dy = -1; dy <= 1; dy++)
                if (pixel_ok((int16_t)(x + dx), (int16_t)(y + dy), (int16_t)w, (int16_t)h, d, NULL))
                    lifo_push(&l, (int16_t)(x + dx), (int16_t)(y + dy));
        double cx, cy;
        if (flood_blob(&cx, &cy, &l, (int16_t)w, (int16_t)h, d, level_image)) {
            xy[2 * i + 0] = rescale_coord(cx, (double)coord_scale);
            xy[2 * i + 1] = rescale_coord(cy, (double)coord_scale);
            level_of_point[i] = (signed char)level;
            nrefined++;
        }
    }
    free(l.v);
    return nrefined;
}
