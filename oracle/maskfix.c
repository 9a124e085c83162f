/*
 * maskfix.c -- CPU ORACLE (test infrastructure): the seam-mask fix-up of the compose loop,
 *
 *   dilate(masks_warped[img_idx], dilated_mask, Mat());                    R64/src/stitcher.cpp:291
 *   resize(dilated_mask, seam_mask, mask_warped.size());                   R64/src/stitcher.cpp:292
 *   mask_warped = seam_mask & mask_warped;                                 R64/src/stitcher.cpp:293-294
 *   (= DRV:676-678, CLI:721-723)
 *
 * and the OpenCV 2.4.11 primitives under it, restated from the published algorithm ("spec to confirm"):
 *   cv::dilate with the default 3x3 rectangle, anchor (1,1), one iteration, border = identity of max
 *   (imgproc/src/morph.cpp, morphologyDefaultBorderValue);
 *   cv::resize(INTER_LINEAR) of CV_8UC1 to an explicit size (imgproc/src/imgwarp.cpp, resizeGeneric_ +
 *   HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,int,short,FixedPtCast<int,uchar,22>>):
 *   weights saturate_cast<short>(w * 2048); rows (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
 */
#include "mst_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void mo_dilate3x3_8u(const uint8_t *src, int w, int h, uint8_t *dst)
{
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int m = 0;
            for (int dy = -1; dy <= 1; ++dy) {
                int yy = y + dy;
                if (yy < 0 || yy >= h) continue;
                for (int dx = -1; dx <= 1; ++dx) {
                    int xx = x + dx;
                    if (xx < 0 || xx >= w) continue;
                    int v = src[(size_t)yy * w + xx];
                    if (v > m) m = v;
                }
            }
            dst[(size_t)y * w + x] = (uint8_t)m;
        }
}

static short to_coef(float w)
{
    /* saturate_cast<short>(float): cvRound, clamp */
    int r = mo_cv_round((double)(w * 2048.f));
    return (short)(r < -32768 ? -32768 : (r > 32767 ? 32767 : r));
}

/* cv::resize(INTER_LINEAR) on CV_8UC(cn) with the two scale factors given (imgwarp.cpp of 2.4.11: `scale = 1 / inv_scale`, where
 * inv_scale is dsize / ssize when the caller passed a size -- stitcher.cpp:292 -- and the caller's fx / fy when it passed Size() and
 * factors -- the sample's `resize(full_img, img, Size(), compose_scale, compose_scale)`, CLI:699-700) */
void mo_resize_linear_8u_cn(const uint8_t *src, int sw, int sh, int cn, uint8_t *dst, int dw, int dh, double scale_x, double scale_y)
{
    int *xofs = (int *)malloc(sizeof(int) * (size_t)dw);
    short *alpha = (short *)malloc(sizeof(short) * 2 * (size_t)dw);
    int xmax = dw;
    for (int dx = 0; dx < dw; ++dx) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = mo_cv_floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= sw) { if (xmax > dx) xmax = dx; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = to_coef(1.f - fx);
        alpha[2 * dx + 1] = to_coef(fx);
    }
    int *rows[2];
    rows[0] = (int *)malloc(sizeof(int) * (size_t)dw * cn);
    rows[1] = (int *)malloc(sizeof(int) * (size_t)dw * cn);
    for (int dy = 0; dy < dh; ++dy) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = mo_cv_floor(fy);
        fy -= sy;
        short b0 = to_coef(1.f - fy), b1 = to_coef(fy);
        for (int k = 0; k < 2; ++k) {
            int yy = sy + k;
            yy = yy < 0 ? 0 : (yy > sh - 1 ? sh - 1 : yy);        /* rows are clipped, the weights stay */
            const uint8_t *S = src + (size_t)yy * sw * cn;
            int *D = rows[k];
            for (int dx = 0; dx < dw; ++dx) {
                int sx = xofs[dx];
                for (int c = 0; c < cn; ++c)
                    D[dx * cn + c] = dx < xmax ? S[sx * cn + c] * alpha[2 * dx] + S[(sx + 1) * cn + c] * alpha[2 * dx + 1] : S[sx * cn + c] * 2048;
            }
        }
        uint8_t *out = dst + (size_t)dy * dw * cn;
        for (int i = 0; i < dw * cn; ++i) {
            int v = (((b0 * (rows[0][i] >> 4)) >> 16) + ((b1 * (rows[1][i] >> 4)) >> 16) + 2) >> 2;
            out[i] = mo_sat_u8(v);
        }
    }
    free(rows[0]); free(rows[1]); free(xofs); free(alpha);
}

void mo_resize_linear_8u(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh)
{
    double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    mo_resize_linear_8u_cn(src, sw, sh, 1, dst, dw, dh, 1. / inv_x, 1. / inv_y);
}

/* stitcher.cpp:291-294: seam-scale mask -> final full-resolution mask */
void mo_mask_fixup(const uint8_t *seam_mask, int sw, int sh, const uint8_t *mask_warped, int w, int h, uint8_t *out)
{
    uint8_t *dil = (uint8_t *)malloc((size_t)sw * sh);
    uint8_t *big = (uint8_t *)malloc((size_t)w * h);
    mo_dilate3x3_8u(seam_mask, sw, sh, dil);
    mo_resize_linear_8u(dil, sw, sh, big, w, h);
    for (size_t i = 0; i < (size_t)w * h; ++i)
        out[i] = big[i] & mask_warped[i];
    free(dil);
    free(big);
}
