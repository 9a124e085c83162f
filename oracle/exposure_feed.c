/*
 * exposure_feed.c -- CPU ORACLE (test infrastructure): exposure-gain ESTIMATION (calibration side).
 *
 * Follows, line by line:
 *   R64/src/exposure_compensate.cpp:73-147    GainCompensator::feed  (overlap statistics N, I; normal equations)
 *   R64/src/exposure_compensate.cpp:165-222   BlocksGainCompensator::feed (blocks as images, two smoothing passes)
 *   (= the driver's BlockFeed / BlockFeedRoot, DRV:251-306)
 * OpenCV 2.4.11 primitives restated from the published algorithm ("spec to confirm"):
 *   cv::solve(A, b, x) with DECOMP_LU on CV_64F (core/src/lapack.cpp): closed forms for n <= 3, else LUImpl --
 *   partial pivoting by strict > on |.|, rows scaled by -1/pivot, back substitution with the stored reciprocals;
 *   cv::sepFilter2D(CV_32F, [1/4 1/2 1/4] twice) (imgproc/src/filter.cpp, SymmRowSmallFilter / SymmColumnSmallFilter
 *   for a symmetric 3-tap float kernel): center * k0 + (left + right) * k1 per pass, BORDER_REFLECT_101, float rows.
 */
#include "mst_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* cv::solve, DECOMP_LU, CV_64F, one right-hand side.  Returns 0 when singular (x untouched = zeros in OpenCV). */
int mo_solve_lu_64f(const double *A_in, const double *b_in, int n, double *x)
{
    if (n == 1) {
        double d = A_in[0];
        if (d == 0.) { x[0] = 0; return 0; }
        x[0] = b_in[0] * (1. / d);                 /* lapack.cpp: n == 1: d = 1./d; dst = src2 * d */
        return 1;
    }
    if (n == 2) {                                   /* Cramer, lapack.cpp:1211-1230 */
#define a(y, x_) A_in[(y) * 2 + (x_)]
        double d = a(0,0) * a(1,1) - a(0,1) * a(1,0);
        if (d == 0.) { x[0] = x[1] = 0; return 0; }
        d = 1. / d;
        double t = (b_in[0] * a(1,1) - b_in[1] * a(0,1)) * d;
        x[1] = (b_in[1] * a(0,0) - b_in[0] * a(1,0)) * d;
        x[0] = t;
#undef a
        return 1;
    }
    if (n == 3) {                                   /* lapack.cpp:1232-1270 */
#define a(y, x_) A_in[(y) * 3 + (x_)]
        double d = a(0,0) * (a(1,1) * a(2,2) - a(1,2) * a(2,1)) - a(0,1) * (a(1,0) * a(2,2) - a(1,2) * a(2,0)) +
                   a(0,2) * (a(1,0) * a(2,1) - a(1,1) * a(2,0));
        if (d == 0.) { x[0] = x[1] = x[2] = 0; return 0; }
        d = 1. / d;
        double t[3];
        t[0] = d * (b_in[0] * (a(1,1) * a(2,2) - a(1,2) * a(2,1)) - a(0,1) * (b_in[1] * a(2,2) - a(1,2) * b_in[2]) +
                    a(0,2) * (b_in[1] * a(2,1) - a(1,1) * b_in[2]));
        t[1] = d * (a(0,0) * (b_in[1] * a(2,2) - a(1,2) * b_in[2]) - b_in[0] * (a(1,0) * a(2,2) - a(1,2) * a(2,0)) +
                    a(0,2) * (a(1,0) * b_in[2] - b_in[1] * a(2,0)));
        t[2] = d * (a(0,0) * (a(1,1) * b_in[2] - b_in[1] * a(2,1)) - a(0,1) * (a(1,0) * b_in[2] - b_in[1] * a(2,0)) +
                    b_in[0] * (a(1,0) * a(2,1) - a(1,1) * a(2,0)));
        x[0] = t[0]; x[1] = t[1]; x[2] = t[2];
#undef a
        return 1;
    }
    double *A = (double *)malloc(sizeof(double) * (size_t)n * n);
    double *b = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(A, A_in, sizeof(double) * (size_t)n * n);
    memcpy(b, b_in, sizeof(double) * (size_t)n);
    int ok = 1;
    for (int i = 0; i < n; ++i) {                   /* LUImpl<double> */
        int k = i;
        for (int j = i + 1; j < n; ++j)
            if (fabs(A[j * n + i]) > fabs(A[k * n + i])) k = j;
        if (fabs(A[k * n + i]) < DBL_EPSILON) { ok = 0; break; }
        if (k != i) {
            for (int j = i; j < n; ++j) { double t = A[i * n + j]; A[i * n + j] = A[k * n + j]; A[k * n + j] = t; }
            double t = b[i]; b[i] = b[k]; b[k] = t;
        }
        double d = -1 / A[i * n + i];
        for (int j = i + 1; j < n; ++j) {
            double alpha = A[j * n + i] * d;
            for (k = i + 1; k < n; ++k) A[j * n + k] += alpha * A[i * n + k];
            b[j] += alpha * b[i];
        }
        A[i * n + i] = -d;
    }
    if (ok) {
        for (int i = n - 1; i >= 0; --i) {
            double s = b[i];
            for (int k = i + 1; k < n; ++k) s -= A[i * n + k] * b[k];
            b[i] = s * A[i * n + i];
        }
        memcpy(x, b, sizeof(double) * (size_t)n);
    } else {
        memset(x, 0, sizeof(double) * (size_t)n);
    }
    free(A);
    free(b);
    return ok;
}

static int reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

/* sepFilter2D(m, m, CV_32F, [.25 .5 .25], [.25 .5 .25]) in place */
void mo_smooth_gain_map(float *m, int w, int h)
{
    const float k0 = 0.5f, k1 = 0.25f;
    float *tmp = (float *)malloc(sizeof(float) * (size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float *r = m + (size_t)y * w;
            tmp[(size_t)y * w + x] = r[x] * k0 + (r[reflect101(x - 1, w)] + r[reflect101(x + 1, w)]) * k1;
        }
    for (int y = 0; y < h; ++y) {
        const float *r0 = tmp + (size_t)reflect101(y - 1, h) * w, *r1 = tmp + (size_t)y * w,
                    *r2 = tmp + (size_t)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; ++x)
            m[(size_t)y * w + x] = (r0[x] + r2[x]) * k1 + r1[x] * k0;
    }
    free(tmp);
}

/* GainCompensator::feed on `n` images given as sub-rectangles of parent buffers (BlocksGainCompensator hands it
 * blocks of the images); level mask value 255 (ExposureCompensator::feed, :63-70). */
typedef struct { const uint8_t *img; size_t istride; const uint8_t *mask; size_t mstride; int x, y, w, h; } sub_image;

static void gain_feed(const sub_image *im, int n, double *gains)
{
    int *N = (int *)calloc((size_t)n * n, sizeof(int));
    double *I = (double *)calloc((size_t)n * n, sizeof(double));
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            mo_rect roi;
            if (!mo_overlap_roi(im[i].x, im[i].y, im[j].x, im[j].y, im[i].w, im[i].h, im[j].w, im[j].h, &roi)) continue;
            int count = 0;
            double Isum1 = 0, Isum2 = 0;
            for (int y = 0; y < roi.height; ++y) {
                const uint8_t *r1 = im[i].img + (size_t)(roi.y - im[i].y + y) * im[i].istride + (size_t)(roi.x - im[i].x) * 3;
                const uint8_t *r2 = im[j].img + (size_t)(roi.y - im[j].y + y) * im[j].istride + (size_t)(roi.x - im[j].x) * 3;
                const uint8_t *m1 = im[i].mask + (size_t)(roi.y - im[i].y + y) * im[i].mstride + (roi.x - im[i].x);
                const uint8_t *m2 = im[j].mask + (size_t)(roi.y - im[j].y + y) * im[j].mstride + (roi.x - im[j].x);
                for (int x = 0; x < roi.width; ++x) {
                    if (m1[x] == 255 && m2[x] == 255) {
                        ++count;
                        int b1 = r1[3 * x], g1 = r1[3 * x + 1], c1 = r1[3 * x + 2];
                        int b2 = r2[3 * x], g2 = r2[3 * x + 1], c2 = r2[3 * x + 2];
                        Isum1 += sqrt((double)(b1 * b1 + g1 * g1 + c1 * c1));
                        Isum2 += sqrt((double)(b2 * b2 + g2 * g2 + c2 * c2));
                    }
                }
            }
            int nn = count > 1 ? count : 1;                     /* max(1, countNonZero(intersect)) */
            N[i * n + j] = N[j * n + i] = nn;
            I[i * n + j] = Isum1 / nn;
            I[j * n + i] = Isum2 / nn;
        }
    const double alpha = 0.01, beta = 100;
    double *A = (double *)calloc((size_t)n * n, sizeof(double));
    double *b = (double *)calloc((size_t)n, sizeof(double));
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            b[i] += beta * N[i * n + j];
            A[i * n + i] += beta * N[i * n + j];
            if (j == i) continue;
            A[i * n + i] += 2 * alpha * I[i * n + j] * I[i * n + j] * N[i * n + j];
            A[i * n + j] -= 2 * alpha * I[i * n + j] * I[j * n + i] * N[i * n + j];
        }
    mo_solve_lu_64f(A, b, n, gains);
    free(N); free(I); free(A); free(b);
}

void mo_gain_feed(int n, const int *corners_xy, const uint8_t *const *images, const size_t *image_strides,
                  const int *sizes_wh, const uint8_t *const *masks, const size_t *mask_strides, double *gains)
{
    sub_image *im = (sub_image *)malloc(sizeof(sub_image) * (size_t)n);
    for (int i = 0; i < n; ++i) {
        sub_image s = { images[i], image_strides[i], masks[i], mask_strides[i], corners_xy[2 * i], corners_xy[2 * i + 1],
                        sizes_wh[2 * i], sizes_wh[2 * i + 1] };
        im[i] = s;
    }
    gain_feed(im, n, gains);
    free(im);
}

void mo_gain_map_size(int w, int h, int bl_width, int bl_height, int *gw, int *gh)
{
    *gw = (w + bl_width - 1) / bl_width;
    *gh = (h + bl_height - 1) / bl_height;
}

/* gain_maps[i]: gh_i x gw_i floats (mo_gain_map_size) */
void mo_blocks_gain_feed(int n, const int *corners_xy, const uint8_t *const *images, const size_t *image_strides,
                         const int *sizes_wh, const uint8_t *const *masks, const size_t *mask_strides,
                         int bl_width_, int bl_height_, float *const *gain_maps)
{
    int total = 0;
    for (int i = 0; i < n; ++i) {
        int gw, gh;
        mo_gain_map_size(sizes_wh[2 * i], sizes_wh[2 * i + 1], bl_width_, bl_height_, &gw, &gh);
        total += gw * gh;
    }
    sub_image *bl = (sub_image *)calloc((size_t)total + 1, sizeof(sub_image));
    int k = 0;
    for (int i = 0; i < n; ++i) {
        const int cols = sizes_wh[2 * i], rows = sizes_wh[2 * i + 1];
        int gw, gh;
        mo_gain_map_size(cols, rows, bl_width_, bl_height_, &gw, &gh);
        const int bl_width = (cols + gw - 1) / gw, bl_height = (rows + gh - 1) / gh;
        for (int by = 0; by < gh; ++by)
            for (int bx = 0; bx < gw; ++bx) {
                const int x0 = bx * bl_width, y0 = by * bl_height;
                const int x1 = x0 + bl_width < cols ? x0 + bl_width : cols, y1 = y0 + bl_height < rows ? y0 + bl_height : rows;
                sub_image s = { images[i] + (size_t)y0 * image_strides[i] + (size_t)x0 * 3, image_strides[i],
                                masks[i] + (size_t)y0 * mask_strides[i] + x0, mask_strides[i],
                                corners_xy[2 * i] + x0, corners_xy[2 * i + 1] + y0, x1 - x0, y1 - y0 };
                bl[k++] = s;
            }
    }
    double *gains = (double *)malloc(sizeof(double) * (size_t)total);
    gain_feed(bl, total, gains);
    k = 0;
    for (int i = 0; i < n; ++i) {
        int gw, gh;
        mo_gain_map_size(sizes_wh[2 * i], sizes_wh[2 * i + 1], bl_width_, bl_height_, &gw, &gh);
        for (int j = 0; j < gw * gh; ++j) gain_maps[i][j] = (float)gains[k++];
        mo_smooth_gain_map(gain_maps[i], gw, gh);
        mo_smooth_gain_map(gain_maps[i], gw, gh);
    }
    free(gains);
    free(bl);
}
