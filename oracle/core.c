/*
 * core.c -- CPU ORACLE (test infrastructure): scalar conversions and the
 * OpenCV 2.4.11 `core` primitives the blenders call.  Restated from the
 * published 2.4.11 algorithm (source not in /root/reference; "spec to
 * confirm", see mst_oracle.h).
 *
 * Call sites in the reference (R64/src):
 *   convertTo(CV_16S)           stitcher.cpp:285, DRV:755
 *   convertTo(CV_8U)            stitcher.cpp:313, DRV:762
 *   copyMakeBorder              blenders.cpp:273 (REFLECT), :295 (CONSTANT)
 *   add / subtract (CV_16S)     blenders.cpp:486, :528
 */
#include "mst_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int g_mo_threads = 1;
void mo_set_num_threads(int n) { g_mo_threads = n < 1 ? 1 : n; }
int mo_get_num_threads(void) { return g_mo_threads; }

/* cvRound on x64 = _mm_cvtsd_si32: round-half-to-even under the default MXCSR,
 * "integer indefinite" 0x80000000 when the result is not representable. */
int mo_cv_round(double v)
{
    double r = nearbyint(v);
    if (!(r >= -2147483648.0 && r <= 2147483647.0))
        return INT_MIN;
    return (int)r;
}

int mo_cv_floor(double v)
{
    double r = floor(v);
    if (!(r >= -2147483648.0 && r <= 2147483647.0))
        return INT_MIN;
    return (int)r;
}

int16_t mo_sat_s16(int v)
{
    return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v));
}

uint8_t mo_sat_u8(int v)
{
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* static_cast<short>(float) in the reference-owned loops (blenders.cpp:321-323,
 * :400-402).  MSVC x64 emits cvttss2si (truncate toward zero, 0x80000000 when
 * out of int32 range or NaN) and keeps the low 16 bits. */
int16_t mo_trunc_s16(float v)
{
    int32_t t;
    if (!(v > -2147483904.0f && v < 2147483648.0f))
        t = INT_MIN;
    else
        t = (int32_t)v;
    return (int16_t)(uint16_t)((uint32_t)t & 0xFFFFu);
}

/* cv::borderInterpolate (imgproc/filter.cpp in 2.4.11) */
int mo_border_interpolate(int p, int len, int border_type)
{
    if ((unsigned)p < (unsigned)len)
        return p;
    if (border_type == MO_BORDER_REPLICATE)
        return p < 0 ? 0 : len - 1;
    if (border_type == MO_BORDER_REFLECT || border_type == MO_BORDER_REFLECT_101) {
        int delta = border_type == MO_BORDER_REFLECT_101;
        if (len == 1)
            return 0;
        do {
            if (p < 0)
                p = -p - 1 + delta;
            else
                p = len - 1 - (p - len) - delta;
        } while ((unsigned)p >= (unsigned)len);
        return p;
    }
    if (border_type == MO_BORDER_WRAP) {
        if (p < 0)
            p -= ((p - len + 1) / len) * len;
        if (p >= len)
            p %= len;
        return p;
    }
    return -1; /* BORDER_CONSTANT */
}

void mo_convert_8u_16s(const uint8_t *src, size_t sstride, int16_t *dst, size_t dstride_elems,
                       int w_elems, int h)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w_elems; ++x)
            dst[(size_t)y * dstride_elems + x] = (int16_t)src[(size_t)y * sstride + x];
}

void mo_convert_16s_8u(const int16_t *src, size_t sstride_elems, uint8_t *dst, size_t dstride,
                       int w_elems, int h)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w_elems; ++x)
            dst[(size_t)y * dstride + x] = mo_sat_u8(src[(size_t)y * sstride_elems + x]);
}

/* cv::copyMakeBorder: dst is (w+left+right) x (h+top+bottom), tightly packed */
void mo_copy_make_border_16s(const int16_t *src, int w, int h, int cn, int16_t *dst,
                             int top, int bottom, int left, int right, int border_type)
{
    int dw = w + left + right, dh = h + top + bottom;
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < dh; ++y) {
        int sy = mo_border_interpolate(y - top, h, border_type);
        for (int x = 0; x < dw; ++x) {
            int sx = mo_border_interpolate(x - left, w, border_type);
            for (int c = 0; c < cn; ++c)
                dst[((size_t)y * dw + x) * cn + c] =
                    (sx < 0 || sy < 0) ? 0 : src[((size_t)sy * w + sx) * cn + c];
        }
    }
}

void mo_copy_make_border_32f(const float *src, int w, int h, float *dst,
                             int top, int bottom, int left, int right, int border_type)
{
    int dw = w + left + right, dh = h + top + bottom;
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < dh; ++y) {
        int sy = mo_border_interpolate(y - top, h, border_type);
        for (int x = 0; x < dw; ++x) {
            int sx = mo_border_interpolate(x - left, w, border_type);
            dst[(size_t)y * dw + x] = (sx < 0 || sy < 0) ? 0.f : src[(size_t)sy * w + sx];
        }
    }
}

/* cv::add / cv::subtract on CV_16S saturate (core/arithm.cpp OpAdd<short>) */
void mo_add_16s(const int16_t *a, const int16_t *b, int16_t *dst, size_t n)
{
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (size_t i = 0; i < n; ++i)
        dst[i] = mo_sat_s16((int)a[i] + (int)b[i]);
}

void mo_subtract_16s(const int16_t *a, const int16_t *b, int16_t *dst, size_t n)
{
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (size_t i = 0; i < n; ++i)
        dst[i] = mo_sat_s16((int)a[i] - (int)b[i]);
}

void mo_free(void *p)
{
    free(p);
}

/* util.cpp:127-140 resultRoi(corners, sizes) */
mo_rect mo_result_roi(const int *corners_xy, const int *sizes_wh, int n)
{
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (int i = 0; i < n; ++i) {
        int cx = corners_xy[2 * i], cy = corners_xy[2 * i + 1];
        int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        if (cx < tlx) tlx = cx;
        if (cy < tly) tly = cy;
        if (cx + w > brx) brx = cx + w;
        if (cy + h > bry) bry = cy + h;
    }
    mo_rect r = { tlx, tly, brx - tlx, bry - tly };
    return r;
}

/* util.cpp:103-115 overlapRoi */
int mo_overlap_roi(int tl1x, int tl1y, int tl2x, int tl2y, int w1, int h1, int w2, int h2, mo_rect *roi)
{
    int x_tl = tl1x > tl2x ? tl1x : tl2x;
    int y_tl = tl1y > tl2y ? tl1y : tl2y;
    int x_br = (tl1x + w1 < tl2x + w2) ? tl1x + w1 : tl2x + w2;
    int y_br = (tl1y + h1 < tl2y + h2) ? tl1y + h1 : tl2y + h2;
    if (x_tl < x_br && y_tl < y_br) {
        roi->x = x_tl; roi->y = y_tl; roi->width = x_br - x_tl; roi->height = y_br - y_tl;
        return 1;
    }
    return 0;
}

/* cv::distanceTransform(mask, dist, CV_DIST_L1, 3): 3x3 chamfer with a=1, b=2 in
 * 16.16 fixed point over a one-pixel border initialised to INT_MAX>>2
 * (imgproc/distransform.cpp icvDistanceTransform_3x3_C1R).  Zero pixels are the
 * sources; the image border is not a source. */
void mo_distance_transform_l1(const uint8_t *mask, int w, int h, float *dist)
{
    const int SHIFT = 16;
    const unsigned INIT0 = (unsigned)(INT_MAX >> 2);
    const unsigned a = 1u << SHIFT, b = 2u << SHIFT;
    const float scale = 1.f / (float)(1 << SHIFT);
    int step = w + 2;
    unsigned *tmp = (unsigned *)malloc(sizeof(unsigned) * (size_t)step * (h + 2));
    for (size_t i = 0; i < (size_t)step * (h + 2); ++i)
        tmp[i] = INIT0;
    /* forward pass */
    for (int y = 0; y < h; ++y) {
        unsigned *t = tmp + (size_t)(y + 1) * step + 1;
        const uint8_t *s = mask + (size_t)y * w;
        for (int x = 0; x < w; ++x) {
            if (!s[x]) {
                t[x] = 0;
            } else {
                unsigned t0 = t[x - step - 1] + b;
                unsigned v = t[x - step] + a;
                if (t0 > v) t0 = v;
                v = t[x - step + 1] + b;
                if (t0 > v) t0 = v;
                v = t[x - 1] + a;
                if (t0 > v) t0 = v;
                t[x] = t0;
            }
        }
    }
    /* backward pass */
    for (int y = h - 1; y >= 0; --y) {
        unsigned *t = tmp + (size_t)(y + 1) * step + 1;
        float *d = dist + (size_t)y * w;
        for (int x = w - 1; x >= 0; --x) {
            unsigned t0 = t[x];
            if (t0 > a) {
                unsigned v = t[x + step + 1] + b;
                if (t0 > v) t0 = v;
                v = t[x + step] + a;
                if (t0 > v) t0 = v;
                v = t[x + step - 1] + b;
                if (t0 > v) t0 = v;
                v = t[x + 1] + a;
                if (t0 > v) t0 = v;
                t[x] = t0;
            }
            d[x] = (float)((float)t0 * scale);
        }
    }
    free(tmp);
}
