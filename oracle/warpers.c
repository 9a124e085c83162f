/*
 * warpers.c -- CPU ORACLE (test infrastructure): rotation-model warpers.
 *
 * Follows, line by line:
 *   R64/src/camera.cpp:66-72                         CameraParams::K
 *   R64/src/warpers.cpp:50-78                        ProjectorBase::setCameraParams
 *   R64/src/warpers.cpp:90-112, 139-168              PlaneWarper::buildMaps / detectResultRoi
 *   R64/src/warpers.cpp:171-212                      SphericalWarper::detectResultRoi
 *   R64/include/opencv2/stitching/detail/warpers_inl.hpp:62-85    buildMaps
 *   ...warpers_inl.hpp:129-140                       warpRoi
 *   ...warpers_inl.hpp:143-203                       detectResultRoi / detectResultRoiByBorder
 *   ...warpers_inl.hpp:206-300                       Plane/Spherical/Cylindrical mapForward/mapBackward
 *   R64/include/opencv2/stitching/detail/warpers.hpp:118-147      which detectResultRoi each warper uses
 * cv::Mat::inv() / operator* on 3x3 CV_32F restated from OpenCV 2.4.11
 * core/lapack.cpp (invert, n==3 closed form) and core/matmul.cpp (gemm, len==3
 * fast path): "spec to confirm".
 */
#include "mst_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MO_PI 3.1415926535897932384626433832795 /* CV_PI */

void mo_camera_K(double focal, double aspect, double ppx, double ppy, float K[9])
{
    /* camera.cpp:66-72 builds K in double; callers convertTo(CV_32F) (stitcher.cpp:271-272) */
    double k[9] = { focal, 0, ppx, 0, focal * aspect, ppy, 0, 0, 1 };
    for (int i = 0; i < 9; ++i)
        K[i] = (float)k[i];
}

/* cv::invert, DECOMP_LU, 3x3 CV_32F: determinant in float arithmetic (the det3
 * macro multiplies float operands), cofactors in double, scaled by 1/det. */
void mo_invert3x3_32f(const float m[9], float out[9], int *ok)
{
#define Sf(y, x) m[(y) * 3 + (x)]
    double d = Sf(0,0) * (Sf(1,1) * Sf(2,2) - Sf(1,2) * Sf(2,1)) -
               Sf(0,1) * (Sf(1,0) * Sf(2,2) - Sf(1,2) * Sf(2,0)) +
               Sf(0,2) * (Sf(1,0) * Sf(2,1) - Sf(1,1) * Sf(2,0));
    if (d != 0.) {
        float t[9];
        d = 1. / d;
        t[0] = (float)(((double)Sf(1,1) * Sf(2,2) - (double)Sf(1,2) * Sf(2,1)) * d);
        t[1] = (float)(((double)Sf(0,2) * Sf(2,1) - (double)Sf(0,1) * Sf(2,2)) * d);
        t[2] = (float)(((double)Sf(0,1) * Sf(1,2) - (double)Sf(0,2) * Sf(1,1)) * d);
        t[3] = (float)(((double)Sf(1,2) * Sf(2,0) - (double)Sf(1,0) * Sf(2,2)) * d);
        t[4] = (float)(((double)Sf(0,0) * Sf(2,2) - (double)Sf(0,2) * Sf(2,0)) * d);
        t[5] = (float)(((double)Sf(0,2) * Sf(1,0) - (double)Sf(0,0) * Sf(1,2)) * d);
        t[6] = (float)(((double)Sf(1,0) * Sf(2,1) - (double)Sf(1,1) * Sf(2,0)) * d);
        t[7] = (float)(((double)Sf(0,1) * Sf(2,0) - (double)Sf(0,0) * Sf(2,1)) * d);
        t[8] = (float)(((double)Sf(0,0) * Sf(1,1) - (double)Sf(0,1) * Sf(1,0)) * d);
        memcpy(out, t, sizeof(t));
        if (ok) *ok = 1;
    } else {
        memset(out, 0, 9 * sizeof(float));
        if (ok) *ok = 0;
    }
#undef Sf
}

/* cv::gemm fast path for len==3, CV_32F, no C: float products summed left to right */
void mo_matmul3x3_32f(const float a[9], const float b[9], float out[9])
{
    float t[9];
    for (int i = 0; i < 3; ++i) {
        const float *ar = a + 3 * i;
        for (int j = 0; j < 3; ++j) {
            float s = ar[0] * b[j] + ar[1] * b[3 + j] + ar[2] * b[6 + j];
            t[3 * i + j] = s;
        }
    }
    memcpy(out, t, sizeof(t));
}

/* warpers.cpp:50-78 */
void mo_set_camera_params(mo_projector *p, const float K[9], const float R[9], const float T[3])
{
    float rinv[9], kinv[9];
    memcpy(p->k, K, 9 * sizeof(float));
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            rinv[3 * i + j] = R[3 * j + i];          /* Rinv = R.t() */
    memcpy(p->rinv, rinv, sizeof(rinv));
    mo_invert3x3_32f(K, kinv, NULL);
    mo_matmul3x3_32f(R, kinv, p->r_kinv);            /* R * K.inv() */
    mo_matmul3x3_32f(K, rinv, p->k_rinv);            /* K * Rinv */
    if (T) { p->t[0] = T[0]; p->t[1] = T[1]; p->t[2] = T[2]; }
    else   { p->t[0] = p->t[1] = p->t[2] = 0.f; }    /* RotationWarperBase: T = zeros */
}

/* warpers_inl.hpp:206-300 */
void mo_map_forward(const mo_projector *p, int kind, float x, float y, float *u, float *v)
{
    const float *r_kinv = p->r_kinv, *t = p->t;
    float scale = p->scale;
    float x_ = r_kinv[0] * x + r_kinv[1] * y + r_kinv[2];
    float y_ = r_kinv[3] * x + r_kinv[4] * y + r_kinv[5];
    float z_ = r_kinv[6] * x + r_kinv[7] * y + r_kinv[8];
    if (kind == MO_WARP_PLANE) {
        x_ = t[0] + x_ / z_ * (1 - t[2]);
        y_ = t[1] + y_ / z_ * (1 - t[2]);
        *u = scale * x_;
        *v = scale * y_;
    } else if (kind == MO_WARP_SPHERICAL) {
        *u = scale * atan2f(x_, z_);
        float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        *v = scale * ((float)MO_PI - acosf(w == w ? w : 0));
    } else {
        *u = scale * atan2f(x_, z_);
        *v = scale * y_ / sqrtf(x_ * x_ + z_ * z_);
    }
}

void mo_map_backward(const mo_projector *p, int kind, float u, float v, float *x, float *y)
{
    const float *k_rinv = p->k_rinv, *t = p->t;
    float scale = p->scale;
    float z;
    if (kind == MO_WARP_PLANE) {
        u = u / scale - t[0];
        v = v / scale - t[1];
        *x = k_rinv[0] * u + k_rinv[1] * v + k_rinv[2] * (1 - t[2]);
        *y = k_rinv[3] * u + k_rinv[4] * v + k_rinv[5] * (1 - t[2]);
        z  = k_rinv[6] * u + k_rinv[7] * v + k_rinv[8] * (1 - t[2]);
        *x /= z;
        *y /= z;
        return;
    }
    float x_, y_, z_;
    u /= scale;
    v /= scale;
    if (kind == MO_WARP_SPHERICAL) {
        float sinv = sinf((float)MO_PI - v);
        x_ = sinv * sinf(u);
        y_ = cosf((float)MO_PI - v);
        z_ = sinv * cosf(u);
    } else {
        x_ = sinf(u);
        y_ = v;
        z_ = cosf(u);
    }
    *x = k_rinv[0] * x_ + k_rinv[1] * y_ + k_rinv[2] * z_;
    *y = k_rinv[3] * x_ + k_rinv[4] * y_ + k_rinv[5] * z_;
    z  = k_rinv[6] * x_ + k_rinv[7] * y_ + k_rinv[8] * z_;
    if (z > 0) { *x /= z; *y /= z; }
    else *x = *y = -1;
}

typedef struct { float tl_u, tl_v, br_u, br_v; } bounds_t;

static void bounds_init(bounds_t *b)
{
    b->tl_u = FLT_MAX; b->tl_v = FLT_MAX; b->br_u = -FLT_MAX; b->br_v = -FLT_MAX;
}

static void bounds_add(bounds_t *b, float u, float v)
{
    /* std::min(a, b) = (b < a) ? b : a -- NaN in u/v leaves the bound untouched */
    b->tl_u = (u < b->tl_u) ? u : b->tl_u;  b->tl_v = (v < b->tl_v) ? v : b->tl_v;
    b->br_u = (b->br_u < u) ? u : b->br_u;  b->br_v = (b->br_v < v) ? v : b->br_v;
}

/* warpers_inl.hpp:169-203 */
static void detect_by_border(const mo_projector *p, int kind, int w, int h, bounds_t *b)
{
    float u, v;
    bounds_init(b);
    for (float x = 0; x < w; ++x) {
        mo_map_forward(p, kind, x, 0, &u, &v);              bounds_add(b, u, v);
        mo_map_forward(p, kind, x, (float)(h - 1), &u, &v); bounds_add(b, u, v);
    }
    for (int y = 0; y < h; ++y) {
        mo_map_forward(p, kind, 0, (float)y, &u, &v);              bounds_add(b, u, v);
        mo_map_forward(p, kind, (float)(w - 1), (float)y, &u, &v); bounds_add(b, u, v);
    }
}

void mo_detect_result_roi(const mo_projector *p, int kind, int src_w, int src_h,
                          int *tl_x, int *tl_y, int *br_x, int *br_y)
{
    bounds_t b;
    float u, v;
    if (kind == MO_WARP_PLANE) {
        /* warpers.cpp:139-168: four corners only */
        bounds_init(&b);
        mo_map_forward(p, kind, 0, 0, &u, &v);                                   bounds_add(&b, u, v);
        mo_map_forward(p, kind, 0, (float)(src_h - 1), &u, &v);                  bounds_add(&b, u, v);
        mo_map_forward(p, kind, (float)(src_w - 1), 0, &u, &v);                  bounds_add(&b, u, v);
        mo_map_forward(p, kind, (float)(src_w - 1), (float)(src_h - 1), &u, &v); bounds_add(&b, u, v);
    } else if (kind == MO_WARP_SPHERICAL) {
        /* warpers.cpp:171-212 */
        int tx, ty, bx, by;
        detect_by_border(p, kind, src_w, src_h, &b);
        tx = (int)b.tl_u; ty = (int)b.tl_v; bx = (int)b.br_u; by = (int)b.br_v;
        float tl_uf = (float)tx, tl_vf = (float)ty, br_uf = (float)bx, br_vf = (float)by;
        float x = p->rinv[1], y = p->rinv[4], z = p->rinv[7];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                float pole = (float)(MO_PI * p->scale);
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, pole);
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, pole);
            }
        }
        x = p->rinv[1]; y = -p->rinv[4]; z = p->rinv[7];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, 0.f);
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, 0.f);
            }
        }
        b.tl_u = tl_uf; b.tl_v = tl_vf; b.br_u = br_uf; b.br_v = br_vf;
    } else {
        /* CylindricalWarper::detectResultRoi -> detectResultRoiByBorder (detail/warpers.hpp:143-146) */
        detect_by_border(p, kind, src_w, src_h, &b);
    }
    *tl_x = (int)b.tl_u; *tl_y = (int)b.tl_v;
    *br_x = (int)b.br_u; *br_y = (int)b.br_v;
}

static void make_projector(mo_projector *p, float scale, const float K[9], const float R[9],
                           const float T[3])
{
    memset(p, 0, sizeof(*p));
    p->scale = scale;
    mo_set_camera_params(p, K, R, T);
}

/* warpers_inl.hpp:129-140 / warpers.cpp:128-136 */
mo_rect mo_warp_roi(int kind, float scale, const float K[9], const float R[9], const float T[3],
                    int src_w, int src_h)
{
    mo_projector p;
    int tlx, tly, brx, bry;
    make_projector(&p, scale, K, R, T);
    mo_detect_result_roi(&p, kind, src_w, src_h, &tlx, &tly, &brx, &bry);
    mo_rect r = { tlx, tly, brx + 1 - tlx, bry + 1 - tly };
    return r;
}

/* warpers_inl.hpp:62-85 / warpers.cpp:90-112.  Returns the warpRoi()-sized rect (the
 * reference returns Rect(tl, br), one less in each dimension, and every caller adds 1). */
mo_rect mo_build_maps(int kind, float scale, const float K[9], const float R[9], const float T[3],
                      int src_w, int src_h, float *xmap, float *ymap)
{
    mo_projector p;
    int tlx, tly, brx, bry;
    make_projector(&p, scale, K, R, T);
    mo_detect_result_roi(&p, kind, src_w, src_h, &tlx, &tly, &brx, &bry);
    int w = brx - tlx + 1;
    for (int v = tly; v <= bry; ++v)
        for (int u = tlx; u <= brx; ++u) {
            float x, y;
            mo_map_backward(&p, kind, (float)u, (float)v, &x, &y);
            xmap[(size_t)(v - tly) * w + (u - tlx)] = x;
            ymap[(size_t)(v - tly) * w + (u - tlx)] = y;
        }
    mo_rect r = { tlx, tly, w, bry - tly + 1 };
    return r;
}

/* warpers_inl.hpp:88-99: warp = buildMaps + remap */
mo_rect mo_warp_8u(int kind, float scale, const float K[9], const float R[9], const float T[3],
                   const uint8_t *src, int sw, int sh, size_t sstride, int cn,
                   int interp, int border_type, uint8_t *dst)
{
    mo_rect roi = mo_warp_roi(kind, scale, K, R, T, sw, sh);
    size_t n = (size_t)roi.width * roi.height;
    float *xmap = (float *)malloc(n * sizeof(float));
    float *ymap = (float *)malloc(n * sizeof(float));
    mo_build_maps(kind, scale, K, R, T, sw, sh, xmap, ymap);
    mo_remap_8u(src, sw, sh, sstride, cn, dst, roi.width, roi.height, (size_t)roi.width * cn,
                xmap, ymap, (size_t)roi.width, interp, border_type);
    free(xmap);
    free(ymap);
    return roi;
}
