/*
 * exposure.c -- CPU ORACLE (test infrastructure): exposure-gain apply.
 *
 * Follows R64/src/exposure_compensate.cpp:225-246 (BlocksGainCompensator::apply,
 * identical to the driver's BlockApply, DRV:310-331) and :150-153
 * (GainCompensator::apply = `image *= gain`).
 * cv::resize(CV_32FC1, INTER_LINEAR) restated from OpenCV 2.4.11
 * imgproc/imgwarp.cpp (resizeGeneric_<HResizeLinear<float,float,float,1,HResizeNoVec>,
 * VResizeLinear<float,float,float,Cast<float,float>,VResizeLinearVec_32f>>):
 * "spec to confirm".
 */
#include "mst_oracle.h"

#include <math.h>
#include <stdlib.h>

void mo_resize_linear_32f(const float *src, int sw, int sh, float *dst, int dw, int dh)
{
    double inv_scale_x = (double)dw / sw, inv_scale_y = (double)dh / sh;
    double scale_x = 1. / inv_scale_x, scale_y = 1. / inv_scale_y;
    int *xofs = (int *)malloc(sizeof(int) * dw);
    float *alpha = (float *)malloc(sizeof(float) * 2 * dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = mo_cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) {
            if (dx < xmax) xmax = dx;
            if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        }
        xofs[dx] = sx;
        alpha[2 * dx] = 1.f - fx;
        alpha[2 * dx + 1] = fx;
    }
    float *row0 = (float *)malloc(sizeof(float) * dw);
    float *row1 = (float *)malloc(sizeof(float) * dw);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = mo_cv_floor(fy);
        fy -= sy;
        float b0 = 1.f - fy, b1 = fy;
        /* rows sy, sy+1 clipped into the image; weights unchanged */
        int y0 = sy < 0 ? 0 : (sy > sh - 1 ? sh - 1 : sy);
        int y1 = sy + 1 < 0 ? 0 : (sy + 1 > sh - 1 ? sh - 1 : sy + 1);
        const float *S0 = src + (size_t)y0 * sw, *S1 = src + (size_t)y1 * sw;
        for (int dx = 0; dx < dw; ++dx) {               /* HResizeLinear */
            int sx = xofs[dx];
            if (dx < xmax) {
                row0[dx] = S0[sx] * alpha[2 * dx] + S0[sx + 1] * alpha[2 * dx + 1];
                row1[dx] = S1[sx] * alpha[2 * dx] + S1[sx + 1] * alpha[2 * dx + 1];
            } else {
                row0[dx] = S0[sx] * 1.f;
                row1[dx] = S1[sx] * 1.f;
            }
        }
        float *D = dst + (size_t)dy * dw;
        for (int dx = 0; dx < dw; ++dx)                 /* VResizeLinear */
            D[dx] = row0[dx] * b0 + row1[dx] * b1;
    }
    free(row0); free(row1); free(xofs); free(alpha);
}

/* exposure_compensate.cpp:225-246 */
void mo_blocks_gain_apply(uint8_t *img, int w, int h, size_t stride,
                          const float *gain_map, int gw, int gh)
{
    float *full = NULL;
    const float *g = gain_map;
    if (gw != w || gh != h) {
        full = (float *)malloc(sizeof(float) * (size_t)w * h);
        mo_resize_linear_32f(gain_map, gw, gh, full, w, h);
        g = full;
    }
    for (int y = 0; y < h; ++y) {
        const float *gain_row = g + (size_t)y * w;
        uint8_t *row = img + (size_t)y * stride;
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) {
                /* saturate_cast<uchar>(uchar * float) = clamp(cvRound(float)) */
                float v = row[3 * x + c] * gain_row[x];
                row[3 * x + c] = mo_sat_u8(mo_cv_round((double)v));
            }
    }
    free(full);
}

/* exposure_compensate.cpp:150-153: `image *= gain` -> convertTo(image, CV_8U, gain):
 * cvtScale_<uchar,uchar,float>: saturate_cast<uchar>(src * (float)gain + 0.f) */
void mo_gain_apply(uint8_t *img, int w, int h, size_t stride, int cn, double gain)
{
    float a = (float)gain, b = 0.f;
    for (int y = 0; y < h; ++y) {
        uint8_t *row = img + (size_t)y * stride;
        for (int x = 0; x < w * cn; ++x) {
            float v = row[x] * a + b;
            row[x] = mo_sat_u8(mo_cv_round((double)v));
        }
    }
}
