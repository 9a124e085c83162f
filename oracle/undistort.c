/*
 * undistort.c -- CPU ORACLE (test infrastructure): the lens-undistort remap of the shipping frame loop
 * (SURVEY.md 8 row a22).
 *
 * Reference call sites: DRV:201-238 RunTestEyeFishMap -> initUndistortRectifyMap(K, D(1x4), Mat(), K,
 * size, CV_16SC2, mapEye1, mapEye2), and DRV:741 remap(iimg, img3[i], mapEye1[i], mapEye2[i],
 * INTER_LINEAR) (border mode defaulted: BORDER_CONSTANT, value 0).
 *
 * Both callees live in OpenCV 2.4.11 (imgproc/undistort.cpp, imgproc/imgwarp.cpp), which is not in
 * /root/reference: restated from the published algorithm -- PARITY UNPINNED ("spec to confirm").
 *   initUndistortRectifyMap: iR = (newK * R)^-1 (closed-form 3x3 inverse in double); per row the ray is
 *   started at (i*ir1+ir2, i*ir4+ir5, i*ir7+ir8) and ADVANCED INCREMENTALLY by (ir0, ir3, ir6) per column;
 *   radial/tangential model in double; CV_16SC2 output: iu = cvRound(u*32), map1 = iu >> 5 (as short),
 *   map2 = (iv & 31) * 32 + (iu & 31).
 *   remap with (CV_16SC2, CV_16UC1) maps: integer tap from map1, weight-table row from map2 & 1023, the same
 *   remapBilinear core as the float-map path (remap.c).
 */
#include "mst_oracle.h"

#include <string.h>

#define INTER_BITS 5
#define INTER_TAB_SIZE (1 << INTER_BITS)
#define INTER_REMAP_COEF_BITS 15

/* cv::invert(DECOMP_LU) on a 3x3 CV_64F matrix: closed form, d = 1/det first (matrix.cpp) */
static int invert3x3_64f(const double s[9], double t[9])
{
    double d = s[0] * (s[4] * s[8] - s[5] * s[7]) - s[1] * (s[3] * s[8] - s[5] * s[6]) +
               s[2] * (s[3] * s[7] - s[4] * s[6]);
    if (d == 0.)
        return 0;
    d = 1. / d;
    t[0] = (s[4] * s[8] - s[5] * s[7]) * d;
    t[1] = (s[2] * s[7] - s[1] * s[8]) * d;
    t[2] = (s[1] * s[5] - s[2] * s[4]) * d;
    t[3] = (s[5] * s[6] - s[3] * s[8]) * d;
    t[4] = (s[0] * s[8] - s[2] * s[6]) * d;
    t[5] = (s[2] * s[3] - s[0] * s[5]) * d;
    t[6] = (s[3] * s[7] - s[4] * s[6]) * d;
    t[7] = (s[1] * s[6] - s[0] * s[7]) * d;
    t[8] = (s[0] * s[4] - s[1] * s[3]) * d;
    return 1;
}

int mo_init_undistort_rectify_map(const double K[9], const double *dist, int ndist, const double *R,
                                  const double newK[9], int width, int height, int16_t *map1, uint16_t *map2)
{
    double ArR[9], ir[9];
    if (R) {                                  /* Ar * R, row times column in index order (gemm) */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0;
                for (int k = 0; k < 3; ++k) s += newK[3 * i + k] * R[3 * k + j];
                ArR[3 * i + j] = s;
            }
    } else {
        memcpy(ArR, newK, sizeof(ArR));       /* R = eye: the product is Ar itself */
    }
    if (!invert3x3_64f(ArR, ir)) memset(ir, 0, sizeof(ir));      /* Mat::inv of a singular matrix: zeros */
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    const double k1 = ndist > 0 ? dist[0] : 0., k2 = ndist > 1 ? dist[1] : 0.;
    const double p1 = ndist > 2 ? dist[2] : 0., p2 = ndist > 3 ? dist[3] : 0.;
    const double k3 = ndist >= 5 ? dist[4] : 0., k4 = ndist >= 8 ? dist[5] : 0.;
    const double k5 = ndist >= 8 ? dist[6] : 0., k6 = ndist >= 8 ? dist[7] : 0.;
    if (!(ndist == 4 || ndist == 5 || ndist == 8 || ndist == 0))
        return -1;                            /* CV_Assert on the coefficient vector shape */
    for (int i = 0; i < height; ++i) {
        int16_t *m1 = map1 + (size_t)i * width * 2;
        uint16_t *m2 = map2 + (size_t)i * width;
        double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
        for (int j = 0; j < width; ++j, _x += ir[0], _y += ir[3], _w += ir[6]) {
            double w = 1. / _w, x = _x * w, y = _y * w;
            double x2 = x * x, y2 = y * y;
            double r2 = x2 + y2, _2xy = 2 * x * y;
            double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            double u = fx * (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2)) + u0;
            double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy) + v0;
            int iu = mo_cv_round(u * INTER_TAB_SIZE);          /* saturate_cast<int>(double) */
            int iv = mo_cv_round(v * INTER_TAB_SIZE);
            m1[j * 2] = (int16_t)(iu >> INTER_BITS);
            m1[j * 2 + 1] = (int16_t)(iv >> INTER_BITS);
            m2[j] = (uint16_t)((iv & (INTER_TAB_SIZE - 1)) * INTER_TAB_SIZE + (iu & (INTER_TAB_SIZE - 1)));
        }
    }
    return 0;
}

/* cv::remap(src, dst, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR, BORDER_CONSTANT, Scalar()) on CV_8U */
void mo_remap_8u_fixed(const uint8_t *src, int sw, int sh, size_t sstride, int cn,
                       uint8_t *dst, int dw, int dh, size_t dstride,
                       const int16_t *map1, const uint16_t *map2)
{
    const int16_t *wtab = mo_bilinear_tab_i();
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int dy = 0; dy < dh; ++dy) {
        const int16_t *XY = map1 + (size_t)dy * dw * 2;
        const uint16_t *FXY = map2 + (size_t)dy * dw;
        uint8_t *D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; ++dx, D += cn) {
            const int sx = XY[dx * 2], sy = XY[dx * 2 + 1];
            const int16_t *w = wtab + (FXY[dx] & (INTER_TAB_SIZE * INTER_TAB_SIZE - 1)) * 4;
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
        }
    }
}


/* ---- the fused (single-resampling) variant: see mst_oracle.h ------------------------------------------------------- */
static void lens_at(const int16_t *map1, const uint16_t *map2, int w, int x, int y, float *mx, float *my)
{
    size_t i = (size_t)y * w + x;
    *mx = (float)map1[2 * i] + (float)(map2[i] & 31) * (1.0f / 32.0f);
    *my = (float)map1[2 * i + 1] + (float)((map2[i] >> 5) & 31) * (1.0f / 32.0f);
}

void mo_compose_lens_maps(float *xmap, float *ymap, size_t n, const int16_t *map1, const uint16_t *map2, int w, int h)
{
    for (size_t k = 0; k < n; ++k) {
        float u = xmap[k], v = ymap[k];
        /* NaN compares false: it ends up at 0 like the clamp of a negative coordinate */
        if (!(u > 0.0f)) u = 0.0f;
        if (!(v > 0.0f)) v = 0.0f;
        if (u > (float)(w - 1)) u = (float)(w - 1);
        if (v > (float)(h - 1)) v = (float)(h - 1);
        int x0 = (int)u, y0 = (int)v;
        if (x0 > w - 2) x0 = w - 2;
        if (y0 > h - 2) y0 = h - 2;
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        int x1 = x0 + 1 < w ? x0 + 1 : x0, y1 = y0 + 1 < h ? y0 + 1 : y0;
        float ax = u - (float)x0, ay = v - (float)y0;
        float ax0 = 1.0f - ax, ay0 = 1.0f - ay;
        float m00x, m00y, m10x, m10y, m01x, m01y, m11x, m11y;
        lens_at(map1, map2, w, x0, y0, &m00x, &m00y);
        lens_at(map1, map2, w, x1, y0, &m10x, &m10y);
        lens_at(map1, map2, w, x0, y1, &m01x, &m01y);
        lens_at(map1, map2, w, x1, y1, &m11x, &m11y);
        float r0x = m00x * ax0 + m10x * ax, r1x = m01x * ax0 + m11x * ax;
        float r0y = m00y * ax0 + m10y * ax, r1y = m01y * ax0 + m11y * ax;
        xmap[k] = r0x * ay0 + r1x * ay;
        ymap[k] = r0y * ay0 + r1y * ay;
    }
}


/* ---- NV12 -> BGR (mst_oracle.h) ------------------------------------------------------------------------------------ */
static uint8_t sat_u8_i(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

void mo_cvt_nv12_bgr(const uint8_t *yuv, int w, int h, size_t stride, uint8_t *bgr, size_t bgr_stride)
{
    /* color.cpp: ITUR_BT_601_CY / CUB / CUG / CVG / CVR, ITUR_BT_601_SHIFT = 20 */
    const int CY = 1220542, CUB = 2116026, CUG = -409993, CVG = -852492, CVR = 1673527, SHIFT = 20;
    const uint8_t *uvp = yuv + stride * (size_t)h;
    for (int j = 0; j < h; j += 2) {
        const uint8_t *y1 = yuv + stride * (size_t)j, *y2 = y1 + stride, *uv = uvp + stride * (size_t)(j / 2);
        uint8_t *row1 = bgr + bgr_stride * (size_t)j, *row2 = row1 + bgr_stride;
        for (int i = 0; i < w; i += 2) {
            int u = (int)uv[i] - 128, v = (int)uv[i + 1] - 128;
            int ruv = (1 << (SHIFT - 1)) + CVR * v;
            int guv = (1 << (SHIFT - 1)) + CVG * v + CUG * u;
            int buv = (1 << (SHIFT - 1)) + CUB * u;
            const uint8_t *ys[4] = {y1 + i, y1 + i + 1, y2 + i, y2 + i + 1};
            uint8_t *os[4] = {row1 + 3 * i, row1 + 3 * i + 3, row2 + 3 * i, row2 + 3 * i + 3};
            for (int k = 0; k < 4; ++k) {
                int yy = (int)*ys[k] - 16;
                if (yy < 0) yy = 0;
                yy *= CY;
                os[k][2] = sat_u8_i((yy + ruv) >> SHIFT);      /* R */
                os[k][1] = sat_u8_i((yy + guv) >> SHIFT);      /* G */
                os[k][0] = sat_u8_i((yy + buv) >> SHIFT);      /* B */
            }
        }
    }
}
