/*
 * blobs_oracle.c -- TEST INFRASTRUCTURE ONLY (see mrgingham_oracle.c).
 *
 * CPU restatement of mrgingham's blob path, find_blobs_from_image_array (find_blobs.cc:14-46): a
 * cv::SimpleBlobDetector with minArea 20, maxArea 80000, minDistBetweenBlobs 5, blobColor 0 and every
 * other parameter at its default, keypoints emitted as (int)(x * 1000 + 0.5) (find_blobs.cc:40-41).
 *
 * PARITY UNPINNED.  All arithmetic lives in OpenCV (un-vendored, version unpinned by upstream), which
 * this image does not have; the upstream project ships no test or fixture for this path.  What follows
 * restates OpenCV's published algorithms:
 *   - SimpleBlobDetector::detect / findBlobs (modules/features2d/src/blobdetector.cpp): threshold sweep
 *     50, 60, .. 210 (THRESH_BINARY: pixel > t), findContours(RETR_LIST, CHAIN_APPROX_NONE) per
 *     threshold, per contour: area filter on the contour moments, inertia-ratio filter (>= 0.1),
 *     convexity filter (contour area / hull area >= 0.95), centre = m10/m00, m01/m00, colour filter
 *     (binarised pixel at the rounded centre == 0), radius = median distance centre -> contour points;
 *     centres of successive thresholds are grouped (distance < 5, or < either radius), groups seen at
 *     >= 2 thresholds give a keypoint at the confidence-weighted mean (confidence = inertia ratio^2);
 *   - findContours: Suzuki & Abe border following as OpenCV implements it (modules/imgproc/src/
 *     contours.cpp: the image is zero-padded by one pixel, outer borders start where 0 -> 1, hole
 *     borders where >= 1 -> 0, traced borders are marked so that each is followed once; every border
 *     point is kept);
 *   - moments of an integer contour (modules/imgproc/src/moments.cpp, contourMoments: Green's theorem,
 *     exact integer sums), contourArea, convex hull area.
 * Order of the contours of one threshold: order of discovery by the raster scan, reversed (what
 * cv::findContours returns for RETR_LIST).
 */
#include "mrgingham_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int x, y; } pt_t;
typedef struct { pt_t* p; int n, cap; } ptvec_t;
typedef struct { double x, y, radius, confidence; } center_t;
typedef struct { center_t* c; int n, cap; } cvec_t;

static void pt_push(ptvec_t* v, int x, int y)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 64;
        v->p = (pt_t*)realloc(v->p, (size_t)v->cap * sizeof(pt_t));
    }
    v->p[v->n].x = x;
    v->p[v->n].y = y;
    v->n++;
}

static void c_push(cvec_t* v, center_t c)
{
    if (v->n == v->cap) {
        v->cap = v->cap ? 2 * v->cap : 16;
        v->c = (center_t*)realloc(v->c, (size_t)v->cap * sizeof(center_t));
    }
    v->c[v->n++] = c;
}

/* direction codes of the border follower: 0 = +x, then counter-clockwise on the screen (y down) */
static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

/* One border, starting at padded position (x0, y0); img: padded (w+2) x (h+2) signed bytes, step bytes per
 * row; points are written in unpadded coordinates.  Marks followed pixels like icvFetchContour. */
static void follow_border(signed char* img, int step, int x0, int y0, int is_hole, ptvec_t* out)
{
    const int nbd = 2;
    signed char* i0 = img + (size_t)y0 * step + x0;
    int s_end = is_hole ? 0 : 4, s = s_end;
    signed char* i1;
    do {
        s = (s - 1) & 7;
        i1 = i0 + DY[s] * step + DX[s];
    } while (*i1 == 0 && s != s_end);
    int px = x0 - 1, py = y0 - 1;
    if (s == s_end) { /* single pixel */
        *i0 = (signed char)(nbd | -128);
        pt_push(out, px, py);
        return;
    }
    signed char* i3 = i0;
    for (;;) {
        s_end = s;
        signed char* i4;
        for (;;) {
            ++s;
            i4 = i3 + DY[s & 7] * step + DX[s & 7];
            if (*i4 != 0) break;
        }
        s &= 7;
        /* the pixel on the right was examined as 0: negative mark (no hole border will start here) */
        if ((unsigned)(s - 1) < (unsigned)s_end) *i3 = (signed char)(nbd | -128);
        else if (*i3 == 1) *i3 = (signed char)nbd;
        pt_push(out, px, py);
        px += DX[s];
        py += DY[s];
        if (i4 == i0 && i3 == i1) break;
        i3 = i4;
        s = (s + 4) & 7;
    }
}

/* moments of an integer contour, contourMoments (moments.cpp): m00 m10 m01 m20 m11 m02 mu20 mu11 mu02 */
static void contour_moments(const ptvec_t* c, double m[9])
{
    memset(m, 0, 9 * sizeof(double));
    if (c->n == 0) return;
    double a00 = 0, a10 = 0, a01 = 0, a20 = 0, a11 = 0, a02 = 0;
    double xi_1 = c->p[c->n - 1].x, yi_1 = c->p[c->n - 1].y;
    for (int i = 0; i < c->n; i++) {
        const double xi = c->p[i].x, yi = c->p[i].y;
        const double xi2 = xi * xi, yi2 = yi * yi;
        const double dxy = xi_1 * yi - xi * yi_1;
        const double xii_1 = xi_1 + xi, yii_1 = yi_1 + yi;
        a00 += dxy;
        a10 += dxy * xii_1;
        a01 += dxy * yii_1;
        a20 += dxy * (xi_1 * xii_1 + xi2);
        a11 += dxy * (xi_1 * (yii_1 + yi_1) + xi * (yii_1 + yi));
        a02 += dxy * (yi_1 * yii_1 + yi2);
        xi_1 = xi;
        yi_1 = yi;
    }
    if (fabs(a00) > FLT_EPSILON) {
        double db1_2, db1_6, db1_12, db1_24;
        if (a00 > 0) { db1_2 = 0.5; db1_6 = 0.16666666666666666666666666666667; db1_12 = 0.083333333333333333333333333333333; db1_24 = 0.041666666666666666666666666666667; }
        else { db1_2 = -0.5; db1_6 = -0.16666666666666666666666666666667; db1_12 = -0.083333333333333333333333333333333; db1_24 = -0.041666666666666666666666666666667; }
        m[0] = a00 * db1_2;
        m[1] = a10 * db1_6;
        m[2] = a01 * db1_6;
        m[3] = a20 * db1_12;
        m[4] = a11 * db1_24;
        m[5] = a02 * db1_12;
        /* completeMomentState */
        double cx = 0, cy = 0;
        if (fabs(m[0]) > DBL_EPSILON) {
            const double inv = 1. / m[0];
            cx = m[1] * inv;
            cy = m[2] * inv;
        }
        m[6] = m[3] - m[1] * cx;
        m[7] = m[4] - m[1] * cy;
        m[8] = m[5] - m[2] * cy;
    }
}

static double polygon_area(const pt_t* p, int n) /* contourArea, not oriented */
{
    if (n == 0) return 0.;
    double a00 = 0;
    double xp = p[n - 1].x, yp = p[n - 1].y;
    for (int i = 0; i < n; i++) {
        const double x = p[i].x, y = p[i].y;
        a00 += xp * y - x * yp;
        xp = x;
        yp = y;
    }
    return fabs(a00 * 0.5);
}

static int cmp_pt(const void* a, const void* b)
{
    const pt_t *p = (const pt_t*)a, *q = (const pt_t*)b;
    if (p->x != q->x) return p->x < q->x ? -1 : 1;
    return p->y < q->y ? -1 : (p->y > q->y ? 1 : 0);
}

static long long cross(pt_t o, pt_t a, pt_t b)
{
    return (long long)(a.x - o.x) * (b.y - o.y) - (long long)(a.y - o.y) * (b.x - o.x);
}

/* area of the convex hull of a point set (any correct hull gives the same area) */
static double hull_area(const ptvec_t* c)
{
    const int n = c->n;
    pt_t* s = (pt_t*)malloc((size_t)n * sizeof(pt_t));
    pt_t* hull = (pt_t*)malloc((size_t)(2 * n + 2) * sizeof(pt_t));
    memcpy(s, c->p, (size_t)n * sizeof(pt_t));
    qsort(s, (size_t)n, sizeof(pt_t), cmp_pt);
    int k = 0;
    for (int i = 0; i < n; i++) {
        while (k >= 2 && cross(hull[k - 2], hull[k - 1], s[i]) <= 0) k--;
        hull[k++] = s[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; i--) {
        while (k >= t && cross(hull[k - 2], hull[k - 1], s[i]) <= 0) k--;
        hull[k++] = s[i];
    }
    const double a = polygon_area(hull, k > 1 ? k - 1 : k);
    free(s);
    free(hull);
    return a;
}

static int cmp_dbl(const void* a, const void* b)
{
    const double x = *(const double*)a, y = *(const double*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static double cv_round(double v) { return nearbyint(v); } /* cvRound: to nearest, ties to even */

/* SimpleBlobDetector::findBlobs for one contour; returns 1 and fills *out when the contour is a blob */
static int contour_to_center(const ptvec_t* c, const uint8_t* image, int w, int h, int stride, int thresh, center_t* out)
{
    double m[9];
    contour_moments(c, m);
    const double area = m[0];
    if (area < 20. || area >= 80000.) return 0;           /* filterByArea: minArea 20, maxArea 80000 */
    double ratio;                                          /* filterByInertia: minInertiaRatio 0.1 */
    {
        const double mu20 = m[6], mu11 = m[7], mu02 = m[8];
        const double denominator = sqrt((2 * mu11) * (2 * mu11) + (mu20 - mu02) * (mu20 - mu02));
        if (denominator > 1e-2) {
            const double cosmin = (mu20 - mu02) / denominator, sinmin = 2 * mu11 / denominator;
            const double cosmax = -cosmin, sinmax = -sinmin;
            const double imin = 0.5 * (mu20 + mu02) - 0.5 * (mu20 - mu02) * cosmin - mu11 * sinmin;
            const double imax = 0.5 * (mu20 + mu02) - 0.5 * (mu20 - mu02) * cosmax - mu11 * sinmax;
            ratio = imin / imax;
        } else {
            ratio = 1;
        }
        if (ratio < (double)0.1f || ratio >= FLT_MAX) return 0; /* float Params promoted to double; maxInertiaRatio = numeric_limits<float>::max() */
    }
    {                                                      /* filterByConvexity: minConvexity 0.95 */
        const double carea = polygon_area(c->p, c->n), harea = hull_area(c);
        if (fabs(harea) < DBL_EPSILON) return 0;
        const double conv = carea / harea;
        if (conv < (double)0.95f || conv >= FLT_MAX) return 0; /* (float)0.95 = 0.949999988... */
    }
    if (m[0] == 0.0) return 0;
    out->x = m[1] / m[0];
    out->y = m[2] / m[0];
    out->confidence = ratio * ratio;
    {                                                      /* filterByColor: blobColor 0 */
        const int cx = (int)cv_round(out->x), cy = (int)cv_round(out->y);
        if (cx < 0 || cx >= w || cy < 0 || cy >= h) return 0;   /* (cannot happen for a closed contour) */
        if (image[(size_t)cy * stride + cx] > thresh) return 0;
    }
    {
        double* d = (double*)malloc((size_t)c->n * sizeof(double));
        for (int i = 0; i < c->n; i++) {
            const double dx = out->x - c->p[i].x, dy = out->y - c->p[i].y;
            d[i] = sqrt(dx * dx + dy * dy);
        }
        qsort(d, (size_t)c->n, sizeof(double), cmp_dbl);
        out->radius = (d[(c->n - 1) / 2] + d[c->n / 2]) / 2.;
        free(d);
    }
    return 1;
}

/* Every contour of the binarised image (pixel > thresh), in cv::findContours' RETR_LIST order. */
static void find_contours(const uint8_t* image, int w, int h, int stride, int thresh, ptvec_t** contours_out, int* n_out)
{
    const int step = w + 2;
    signed char* img = (signed char*)calloc((size_t)step * (h + 2), 1);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) img[(size_t)(y + 1) * step + x + 1] = image[(size_t)y * stride + x] > thresh;
    ptvec_t* list = NULL;
    int n = 0, cap = 0;
    for (int y = 1; y <= h; y++) {
        signed char* row = img + (size_t)y * step;
        int prev = row[0];
        for (int x = 1; x <= w; x++) { /* image columns only: the zero pad is never a hole */
            int p = row[x];
            if (p == prev) continue;
            int is_hole = 0, start = 1;
            if (!(prev == 0 && p == 1)) {
                if (p != 0 || prev < 1) start = 0;
                else is_hole = 1;
            }
            if (start) {
                if (n == cap) {
                    cap = cap ? 2 * cap : 64;
                    list = (ptvec_t*)realloc(list, (size_t)cap * sizeof(ptvec_t));
                }
                memset(&list[n], 0, sizeof(ptvec_t));
                follow_border(img, step, x - is_hole, y, is_hole, &list[n]);
                n++;
                p = row[x]; /* the follower may have marked this very pixel */
            }
            prev = p;
        }
    }
    free(img);
    for (int i = 0; i < n / 2; i++) { /* last found first */
        const ptvec_t t = list[i];
        list[i] = list[n - 1 - i];
        list[n - 1 - i] = t;
    }
    *contours_out = list;
    *n_out = n;
}

int oracle_find_blobs(int32_t* xy_out, int cap_out, const uint8_t* image, int w, int h, int stride)
{
    cvec_t* groups = NULL; /* centers of one blob across thresholds, kept sorted by radius */
    int ngroups = 0, gcap = 0;
    for (int thresh = 50; thresh < 220; thresh += 10) {
        ptvec_t* contours;
        int nc;
        find_contours(image, w, h, stride, thresh, &contours, &nc);
        cvec_t cur = {0, 0, 0};
        for (int i = 0; i < nc; i++) {
            center_t c;
            if (contour_to_center(&contours[i], image, w, h, stride, thresh, &c)) c_push(&cur, c);
            free(contours[i].p);
        }
        free(contours);
        const int ngroups_before = ngroups;
        for (int i = 0; i < cur.n; i++) {
            int is_new = 1;
            for (int j = 0; j < ngroups_before; j++) {
                const center_t* mid = &groups[j].c[groups[j].n / 2];
                const double dx = mid->x - cur.c[i].x, dy = mid->y - cur.c[i].y;
                const double dist = sqrt(dx * dx + dy * dy);
                is_new = dist >= 5. && dist >= mid->radius && dist >= cur.c[i].radius;
                if (!is_new) {
                    c_push(&groups[j], cur.c[i]);
                    int k = groups[j].n - 1;
                    while (k > 0 && cur.c[i].radius < groups[j].c[k - 1].radius) {
                        groups[j].c[k] = groups[j].c[k - 1];
                        k--;
                    }
                    groups[j].c[k] = cur.c[i];
                    break;
                }
            }
            if (is_new) { /* joins the list only after this threshold's centres are all placed */
                if (ngroups == gcap) {
                    gcap = gcap ? 2 * gcap : 64;
                    groups = (cvec_t*)realloc(groups, (size_t)gcap * sizeof(cvec_t));
                }
                memset(&groups[ngroups], 0, sizeof(cvec_t));
                c_push(&groups[ngroups], cur.c[i]);
                ngroups++;
            }
        }
        free(cur.c);
    }
    int nout = 0;
    for (int i = 0; i < ngroups; i++) {
        if (groups[i].n >= 2) { /* minRepeatability */
            double sx = 0, sy = 0, norm = 0;
            for (int j = 0; j < groups[i].n; j++) {
                sx += groups[i].c[j].confidence * groups[i].c[j].x;
                sy += groups[i].c[j].confidence * groups[i].c[j].y;
                norm += groups[i].c[j].confidence;
            }
            const double inv = 1. / norm;
            sx *= inv;
            sy *= inv;
            const float fx = (float)sx, fy = (float)sy;       /* KeyPoint::pt is a Point2f */
            const float tx = fx * 1000.0f, ty = fy * 1000.0f; /* it->pt.x * FIND_GRID_SCALE, find_blobs.cc:40-41 */
            if (nout < cap_out) {
                xy_out[2 * nout + 0] = (int)((double)tx + 0.5);
                xy_out[2 * nout + 1] = (int)((double)ty + 0.5);
            }
            nout++;
        }
        free(groups[i].c);
    }
    free(groups);
    return nout;
}
