/*
 * pyramid.c -- CPU ORACLE (test infrastructure): cv::pyrDown / cv::pyrUp as
 * OpenCV 2.4.11 imgproc/pyramids.cpp computes them (source not in
 * /root/reference: "spec to confirm").
 *
 * Call sites in the reference (R64/src/blenders.cpp):
 *   pyrDown CV_16SC3  :481 (createLaplacePyr, 16S branch)
 *   pyrDown CV_32FC1  :298 (weight pyramid)
 *   pyrUp   CV_16SC3  :485 (createLaplacePyr), :527 (restoreImageFromLaplacePyr)
 *
 *   pyrDown_<FixPtCast<short,8>, PyrDownVec_32s16s>  -- integer, (v+128)>>8
 *   pyrDown_<FltCast<float,8>,  PyrDownVec_32f>      -- float; the SSE vertical pass
 *        (x64 build) sums  (r0+r4) + 2*r2 + 4*((r1+r3)+r2)  for x < width&~7 and the
 *        scalar tail sums  r2*6 + (r1+r3)*4 + r0 + r4 ; both then multiply by 1/256.
 *   pyrUp_<FixPtCast<short,6>, NoVec>                -- integer, (v+32)>>6
 * Borders: BORDER_REFLECT_101 (BORDER_DEFAULT).
 */
#include "mst_oracle.h"

#include <stdlib.h>

/* Separable evaluation (horizontal pass into int rows, then vertical), exactly as
 * pyrDown_ does; integer arithmetic, so the result does not depend on the order. */
void mo_pyr_down_16s(const int16_t *src, int w, int h, int cn, int16_t *dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int width = dw * cn;
    int *hbuf = (int *)malloc(sizeof(int) * (size_t)h * width);
    int *tab = (int *)malloc(sizeof(int) * 5 * (size_t)dw);
    for (int x = 0; x < dw; ++x)
        for (int k = 0; k < 5; ++k)
            tab[5 * x + k] = mo_border_interpolate(2 * x + k - 2, w, MO_BORDER_REFLECT_101) * cn;
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < h; ++y) {
        const int16_t *s = src + (size_t)y * w * cn;
        int *row = hbuf + (size_t)y * width;
        for (int x = 0; x < dw; ++x) {
            const int *t = tab + 5 * x;
            for (int c = 0; c < cn; ++c)
                row[x * cn + c] = s[t[2] + c] * 6 + (s[t[1] + c] + s[t[3] + c]) * 4 + s[t[0] + c] + s[t[4] + c];
        }
    }
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < dh; ++y) {
        const int *r[5];
        for (int k = 0; k < 5; ++k)
            r[k] = hbuf + (size_t)mo_border_interpolate(2 * y + k - 2, h, MO_BORDER_REFLECT_101) * width;
        int16_t *d = dst + (size_t)y * width;
        for (int x = 0; x < width; ++x)
            d[x] = mo_sat_s16((r[2][x] * 6 + (r[1][x] + r[3][x]) * 4 + r[0][x] + r[4][x] + 128) >> 8);
    }
    free(tab);
    free(hbuf);
}

void mo_pyr_down_32f(const float *src, int w, int h, int cn, float *dst)
{
    int dw = (w + 1) / 2, dh = (h + 1) / 2;
    int width = dw * cn;
    int vec_end = width - (width % 8);        /* PyrDownVec_32f handles x <= width-8 in steps of 8 */
    float *hbuf = (float *)malloc(sizeof(float) * (size_t)h * width);
    int *tab = (int *)malloc(sizeof(int) * 5 * (size_t)dw);
    for (int x = 0; x < dw; ++x)
        for (int k = 0; k < 5; ++k)
            tab[5 * x + k] = mo_border_interpolate(2 * x + k - 2, w, MO_BORDER_REFLECT_101) * cn;
    /* horizontal pass (scalar in 2.4.11): src[2x]*6 + (src[2x-1]+src[2x+1])*4 + src[2x-2] + src[2x+2] */
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < h; ++y) {
        const float *s = src + (size_t)y * w * cn;
        float *row = hbuf + (size_t)y * width;
        for (int x = 0; x < dw; ++x) {
            const int *t = tab + 5 * x;
            for (int c = 0; c < cn; ++c)
                row[x * cn + c] = s[t[2] + c] * 6 + (s[t[1] + c] + s[t[3] + c]) * 4 + s[t[0] + c] + s[t[4] + c];
        }
    }
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < dh; ++y) {
        const float *r[5];
        for (int k = 0; k < 5; ++k)
            r[k] = hbuf + (size_t)mo_border_interpolate(2 * y + k - 2, h, MO_BORDER_REFLECT_101) * width;
        const float *row0 = r[0], *row1 = r[1], *row2 = r[2], *row3 = r[3], *row4 = r[4];
        float *d = dst + (size_t)y * width;
        int x = 0;
        for (; x < vec_end; ++x) {
            float r0 = row0[x] + row4[x];
            float r1 = (row1[x] + row3[x]) + row2[x];
            r0 = r0 + (row2[x] + row2[x]);
            float t0 = r0 + r1 * 4.f;
            d[x] = t0 * (1.f / 256);
        }
        for (; x < width; ++x)
            d[x] = (row2[x] * 6 + (row1[x] + row3[x]) * 4 + row0[x] + row4[x]) * (1.f / 256);
    }
    free(tab);
    free(hbuf);
}

/* pyrUp to an explicit even dsize == 2*ssize (the only case the multi-band path produces:
 * every bordered image and the canvas are multiples of 2^num_bands, blenders.cpp:212-213,253-258) */
void mo_pyr_up_16s(const int16_t *src, int sw, int sh, int cn, int16_t *dst, int dw, int dh)
{
    if (dw != 2 * sw || dh != 2 * sh)
        abort();
    int width = dw * cn;
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < sh; ++y) {
        int *buf = (int *)malloc(sizeof(int) * 3 * (size_t)width);
        /* horizontal pass of source rows y-1, y, y+1; row index via reflect101(2*sy, dh)/2 */
        for (int k = 0; k < 3; ++k) {
            int sy = mo_border_interpolate((y - 1 + k) * 2, dh, MO_BORDER_REFLECT_101) / 2;
            const int16_t *s = src + (size_t)sy * sw * cn;
            int *row = buf + (size_t)k * width;
            if (sw == 1) {
                for (int c = 0; c < cn; ++c)
                    row[c] = row[c + cn] = s[c] * 8;
                continue;
            }
            for (int c = 0; c < cn; ++c) {
                row[c] = s[c] * 6 + s[c + cn] * 2;
                row[c + cn] = (s[c] + s[c + cn]) * 4;
                int sx = (sw - 1) * cn + c;
                row[2 * sx - c] = s[sx - cn] + s[sx] * 7;          /* dx = 2*(sw-1)*cn + c */
                row[2 * sx - c + cn] = s[sx] * 8;
            }
            for (int x = 1; x < sw - 1; ++x)
                for (int c = 0; c < cn; ++c) {
                    int e = x * cn + c, dx = 2 * x * cn + c;
                    row[dx] = s[e - cn] + s[e] * 6 + s[e + cn];
                    row[dx + cn] = (s[e] + s[e + cn]) * 4;
                }
        }
        const int *row0 = buf, *row1 = buf + width, *row2 = buf + 2 * (size_t)width;
        int16_t *d0 = dst + (size_t)(2 * y) * width, *d1 = d0 + width;
        for (int x = 0; x < width; ++x) {
            d1[x] = mo_sat_s16(((row1[x] + row2[x]) * 4 + 32) >> 6);
            d0[x] = mo_sat_s16((row0[x] + row1[x] * 6 + row2[x] + 32) >> 6);
        }
        free(buf);
    }
}
