/*
 * remap.c -- CPU ORACLE (test infrastructure): cv::remap for CV_8U images with
 * CV_32FC1 x/y maps, restated from OpenCV 2.4.11 imgproc/imgwarp.cpp
 * (remap -> remapNearest / remapBilinear<FixPtCast<uchar,15>, RemapVec_8u, short>,
 * initInterTab2D).  Source not in /root/reference: "spec to confirm".
 *
 * Call sites in the reference: warpers_inl.hpp:96 (RotationWarperBase::warp),
 * warpers.cpp:122 (PlaneWarper::warp), DRV:195 (warpsize), DRV:752 (per-frame).
 *   image: INTER_LINEAR  + BORDER_REFLECT   (stitcher.cpp:275)
 *   mask : INTER_NEAREST + BORDER_CONSTANT  (stitcher.cpp:280)
 */
#include "mst_oracle.h"

#include <limits.h>
#include <string.h>

#define INTER_BITS 5
#define INTER_TAB_SIZE (1 << INTER_BITS)
#define INTER_TAB_SIZE2 (INTER_TAB_SIZE * INTER_TAB_SIZE)
#define INTER_REMAP_COEF_BITS 15
#define INTER_REMAP_COEF_SCALE (1 << INTER_REMAP_COEF_BITS)

/* 4 extra entries: initInterTab2D's sum fix-up scans itab[k1*ksize+k2] for k1,k2 in
 * [ksize/2, ksize/2+2), which for ksize==2 reads past the 2x2 block. */
static int16_t g_bilinear_tab_i[INTER_TAB_SIZE2 * 4 + 8];
static int g_tab_ready = 0;

static int16_t sat_short_from_float(float v)
{
    int iv = mo_cv_round((double)v);
    return mo_sat_s16(iv);
}

static void init_inter_tab_2d(void)
{
    float tab1d[INTER_TAB_SIZE * 2];
    const float scale = 1.f / INTER_TAB_SIZE;
    const int ksize = 2;
    memset(g_bilinear_tab_i, 0, sizeof(g_bilinear_tab_i));
    for (int i = 0; i < INTER_TAB_SIZE; ++i) {       /* interpolateLinear */
        float x = i * scale;
        tab1d[i * 2 + 0] = 1.f - x;
        tab1d[i * 2 + 1] = x;
    }
    int16_t *itab = g_bilinear_tab_i;
    for (int i = 0; i < INTER_TAB_SIZE; ++i)
        for (int j = 0; j < INTER_TAB_SIZE; ++j, itab += ksize * ksize) {
            int isum = 0;
            for (int k1 = 0; k1 < ksize; ++k1) {
                float vy = tab1d[i * ksize + k1];
                for (int k2 = 0; k2 < ksize; ++k2) {
                    float v = vy * tab1d[j * ksize + k2];
                    isum += itab[k1 * ksize + k2] = sat_short_from_float(v * INTER_REMAP_COEF_SCALE);
                }
            }
            if (isum != INTER_REMAP_COEF_SCALE) {
                int diff = isum - INTER_REMAP_COEF_SCALE;
                int ksize2 = ksize / 2, Mk1 = ksize2, Mk2 = ksize2, mk1 = ksize2, mk2 = ksize2;
                for (int k1 = ksize2; k1 < ksize2 + 2; ++k1)
                    for (int k2 = ksize2; k2 < ksize2 + 2; ++k2) {
                        if (itab[k1 * ksize + k2] < itab[mk1 * ksize + mk2])
                            mk1 = k1, mk2 = k2;
                        else if (itab[k1 * ksize + k2] > itab[Mk1 * ksize + Mk2])
                            Mk1 = k1, Mk2 = k2;
                    }
                if (diff < 0)
                    itab[Mk1 * ksize + Mk2] = (int16_t)(itab[Mk1 * ksize + Mk2] - diff);
                else
                    itab[mk1 * ksize + mk2] = (int16_t)(itab[mk1 * ksize + mk2] + diff);
            }
        }
    g_tab_ready = 1;
}

const int16_t *mo_bilinear_tab_i(void)
{
    if (!g_tab_ready)
        init_inter_tab_2d();
    return g_bilinear_tab_i;
}

void mo_remap_8u(const uint8_t *src, int sw, int sh, size_t sstride, int cn,
                 uint8_t *dst, int dw, int dh, size_t dstride,
                 const float *xmap, const float *ymap, size_t map_stride_elems,
                 int interp, int border_type)
{
    const int16_t *wtab = mo_bilinear_tab_i();
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int dy = 0; dy < dh; ++dy) {
        const float *sX = xmap + (size_t)dy * map_stride_elems;
        const float *sY = ymap + (size_t)dy * map_stride_elems;
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx, D += cn) {
            if (interp == MO_INTER_NEAREST) {
                /* XY = saturate_cast<short>(map) ; remapNearest */
                int sx = mo_sat_s16(mo_cv_round((double)sX[dx]));
                int sy = mo_sat_s16(mo_cv_round((double)sY[dx]));
                if ((unsigned)sx < (unsigned)sw && (unsigned)sy < (unsigned)sh) {
                    for (int k = 0; k < cn; ++k) D[k] = src[(size_t)sy * sstride + sx * cn + k];
                } else if (border_type == MO_BORDER_CONSTANT) {
                    for (int k = 0; k < cn; ++k) D[k] = 0;
                } else {
                    int bx = mo_border_interpolate(sx, sw, border_type);
                    int by = mo_border_interpolate(sy, sh, border_type);
                    for (int k = 0; k < cn; ++k) D[k] = src[(size_t)by * sstride + bx * cn + k];
                }
                continue;
            }
            /* fixed-point coordinates: sx = cvRound(map*32) (float product), integer part
             * saturated to short, 5+5 fraction bits index the weight table */
            int fsx = mo_cv_round((double)(sX[dx] * (float)INTER_TAB_SIZE));
            int fsy = mo_cv_round((double)(sY[dx] * (float)INTER_TAB_SIZE));
            int a = (fsy & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (fsx & (INTER_TAB_SIZE - 1));
            int sx = mo_sat_s16(fsx >> INTER_BITS);
            int sy = mo_sat_s16(fsy >> INTER_BITS);
            const int16_t *w = wtab + a * 4;
            if (border_type == MO_BORDER_CONSTANT) {
                for (int k = 0; k < cn; ++k) {
                    int v[4];
                    for (int t = 0; t < 4; ++t) {
                        int tx = sx + (t & 1), ty = sy + (t >> 1);
                        v[t] = ((unsigned)tx < (unsigned)sw && (unsigned)ty < (unsigned)sh)
                                   ? src[(size_t)ty * sstride + tx * cn + k] : 0;
                    }
                    int s = v[0] * w[0] + v[1] * w[1] + v[2] * w[2] + v[3] * w[3];
                    D[k] = mo_sat_u8((s + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS);
                }
            } else {
                int sx0 = mo_border_interpolate(sx, sw, border_type);
                int sx1 = mo_border_interpolate(sx + 1, sw, border_type);
                int sy0 = mo_border_interpolate(sy, sh, border_type);
                int sy1 = mo_border_interpolate(sy + 1, sh, border_type);
                for (int k = 0; k < cn; ++k) {
                    int v0 = src[(size_t)sy0 * sstride + sx0 * cn + k];
                    int v1 = src[(size_t)sy0 * sstride + sx1 * cn + k];
                    int v2 = src[(size_t)sy1 * sstride + sx0 * cn + k];
                    int v3 = src[(size_t)sy1 * sstride + sx1 * cn + k];
                    int s = v0 * w[0] + v1 * w[1] + v2 * w[2] + v3 * w[3];
                    D[k] = mo_sat_u8((s + (1 << (INTER_REMAP_COEF_BITS - 1))) >> INTER_REMAP_COEF_BITS);
                }
            }
        }
    }
}
