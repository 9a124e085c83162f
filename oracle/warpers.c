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
 *   ...warpers_inl.hpp:302-759                       the other eleven projectors (fisheye ... plane portrait)
 *   R64/include/opencv2/stitching/detail/warpers.hpp:190-503, R64/src/warpers.cpp:389-430
 *                                                    their warpers and which detectResultRoi each uses
 * Those eleven call libm per pixel (atan2f, sqrtf, sinf, tanf, asinf, sinhf, logf ...); here that is glibc's,
 * the reference ran MSVC 2010's CRT: the last bit of a map value is not pinned by anything in this image.
 * Where the reference writes cos(x) / sin(x) / fabs(x) without the f suffix the call is taken as the C one
 * (double), as the 2.4.11 headers give no float overload a better claim.
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

/* cv::invert, DECOMP_LU, 3x3 CV_32F: determinant AND cofactors in double, scaled by 1/det.
 * Round 5, read off the shipped opencv_core2411.dll (0x1800af63f-0x1800af728: every operand through cvtps2pd, mulsd / subsd / addsd,
 * then divsd): det3 is ((m11 m22 - m12 m21) m00 - (m10 m22 - m12 m20) m01) + (m10 m21 - m11 m20) m02 in DOUBLE -- 2.4.11's macro casts
 * its inner products to double.  The first statement of this function (rounds 1-4) had the determinant in float: corrected. */
void mo_invert3x3_32f(const float m[9], float out[9], int *ok)
{
#define Sf(y, x) m[(y) * 3 + (x)]
#define Sd(y, x) ((double)m[(y) * 3 + (x)])
    double d = Sd(0,0) * (Sd(1,1) * Sd(2,2) - Sd(1,2) * Sd(2,1)) -
               Sd(0,1) * (Sd(1,0) * Sd(2,2) - Sd(1,2) * Sd(2,0)) +
               Sd(0,2) * (Sd(1,0) * Sd(2,1) - Sd(1,1) * Sd(2,0));
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
#undef Sd
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

static float custom_a = 1.f, custom_b = 1.f;
void mo_set_custom_ab(float a, float b) { custom_a = a; custom_b = b; }

static void kind_ab(int kind, float *a, float *b)
{
    switch ((kind >> 8) & 3) {
    case 1:  *a = 2.f;   *b = 1.f; break;      /* CLI:561,563,565,567 */
    case 2:  *a = 1.5f;  *b = 1.f; break;      /* CLI:562,564,566,568 */
    case 3:  *a = custom_a; *b = custom_b; break;
    default: *a = 1.f;   *b = 1.f; break;      /* float A = 1, float B = 1 (detail/warpers.hpp:232,254,275,297) */
    }
}

/* warpers_inl.hpp:302-759, forward direction */
static void map_forward_other(const mo_projector *p, int kind, float x, float y, float *u, float *v)
{
    const float *r_kinv = p->r_kinv, *t = p->t;
    const float scale = p->scale, a = p->a, b = p->b;
    float x_, y_, z_;
    if (kind == MO_WARP_COMPRESSED_PLANE_PORTRAIT || kind == MO_WARP_PANINI_PORTRAIT) {
        y_ = r_kinv[0] * x + r_kinv[1] * y + r_kinv[2];        /* :408-410, :505-507: x_ and y_ trade places */
        x_ = r_kinv[3] * x + r_kinv[4] * y + r_kinv[5];
        z_ = r_kinv[6] * x + r_kinv[7] * y + r_kinv[8];
    } else {
        x_ = r_kinv[0] * x + r_kinv[1] * y + r_kinv[2];
        y_ = r_kinv[3] * x + r_kinv[4] * y + r_kinv[5];
        z_ = r_kinv[6] * x + r_kinv[7] * y + r_kinv[8];
    }
    switch (kind) {
    case MO_WARP_FISHEYE: {                                   /* :303-315 */
        float u_ = atan2f(x_, z_);
        float v_ = (float)MO_PI - acosf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        *u = scale * v_ * cosf(u_);
        *v = scale * v_ * sinf(u_);
        break;
    }
    case MO_WARP_STEREOGRAPHIC: {                             /* :343-357 */
        float u_ = atan2f(x_, z_);
        float v_ = (float)MO_PI - acosf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float r = sinf(v_) / (1 - cosf(v_));
        *u = (float)(scale * r * cos(u_));                    /* cos / sin, not cosf / sinf, in the reference */
        *v = (float)(scale * r * sin(u_));
        break;
    }
    case MO_WARP_COMPRESSED_PLANE: {                          /* :386-397 */
        float u_ = atan2f(x_, z_);
        float v_ = asinf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        *u = scale * a * tanf(u_ / a);
        *v = scale * b * tanf(v_) / cosf(u_);
        break;
    }
    case MO_WARP_COMPRESSED_PLANE_PORTRAIT: {                 /* :427-438 */
        float u_ = atan2f(x_, z_);
        float v_ = asinf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        *u = - scale * a * tanf(u_ / a);
        *v = scale * b * tanf(v_) / cosf(u_);
        break;
    }
    case MO_WARP_PANINI:                                      /* :468-486 */
    case MO_WARP_PANINI_PORTRAIT: {                           /* :515-533 */
        float u_ = atan2f(x_, z_);
        float v_ = asinf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float tg = a * tanf(u_ / a);
        *u = kind == MO_WARP_PANINI ? scale * tg : - scale * tg;
        float sinu = sinf(u_);
        if (fabs(sinu) < 1E-7)
            *v = scale * b * tanf(v_);
        else
            *v = scale * b * tg * tanf(v_) / sinu;
        break;
    }
    case MO_WARP_MERCATOR: {                                  /* :555-567 */
        float u_ = atan2f(x_, z_);
        float v_ = asinf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        *u = scale * u_;
        *v = scale * logf(tanf((float)(MO_PI / 4) + v_ / 2));
        break;
    }
    case MO_WARP_TRANSVERSE_MERCATOR: {                       /* :594-608 */
        float u_ = atan2f(x_, z_);
        float v_ = asinf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_));
        float B = cosf(v_) * sinf(u_);
        *u = scale / 2 * logf((1 + B) / (1 - B));
        *v = scale * atan2f(tanf(v_), cosf(u_));
        break;
    }
    case MO_WARP_SPHERICAL_PORTRAIT: {                        /* :635-651 */
        float x0_ = x_, y0_ = y_;
        x_ = y0_; y_ = x0_;
        float uu = scale * atan2f(x_, z_);
        float vv = scale * ((float)MO_PI - acosf(y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_)));
        *u = -uu;
        *v = vv;
        break;
    }
    case MO_WARP_CYLINDRICAL_PORTRAIT: {                      /* :685-701 */
        float x0_ = x_, y0_ = y_;
        x_ = y0_; y_ = x0_;
        float uu = scale * atan2f(x_, z_);
        float vv = scale * y_ / sqrtf(x_ * x_ + z_ * z_);
        *u = -uu;
        *v = vv;
        break;
    }
    default: {                                                /* plane portrait :733-752 */
        float x0_ = x_, y0_ = y_;
        x_ = y0_; y_ = x0_;
        x_ = t[0] + x_ / z_ * (1 - t[2]);
        y_ = t[1] + y_ / z_ * (1 - t[2]);
        float uu = scale * x_, vv = scale * y_;
        *u = -uu;
        *v = vv;
        break;
    }
    }
}

/* warpers_inl.hpp:302-759, backward direction */
static void map_backward_other(const mo_projector *p, int kind, float u, float v, float *x, float *y)
{
    const float *k_rinv = p->k_rinv, *t = p->t;
    const float scale = p->scale, a = p->a, b = p->b;
    float x_, y_, z_, z;
    if (kind == MO_WARP_PLANE_PORTRAIT) {                     /* :755-773: no z test, v and u swapped */
        float uu = -u, vv = v;
        uu = uu / scale - t[0];
        vv = vv / scale - t[1];
        *x = k_rinv[0] * vv + k_rinv[1] * uu + k_rinv[2] * (1 - t[2]);
        *y = k_rinv[3] * vv + k_rinv[4] * uu + k_rinv[5] * (1 - t[2]);
        z  = k_rinv[6] * vv + k_rinv[7] * uu + k_rinv[8] * (1 - t[2]);
        *x /= z;
        *y /= z;
        return;
    }
    switch (kind) {
    case MO_WARP_FISHEYE: {                                   /* :318-340 */
        u /= scale; v /= scale;
        float u_ = atan2f(v, u);
        float v_ = sqrtf(u * u + v * v);
        float sinv = sinf((float)MO_PI - v_);
        x_ = sinv * sinf(u_);
        y_ = cosf((float)MO_PI - v_);
        z_ = sinv * cosf(u_);
        break;
    }
    case MO_WARP_STEREOGRAPHIC: {                             /* :360-383 */
        u /= scale; v /= scale;
        float u_ = atan2f(v, u);
        float r = sqrtf(u * u + v * v);
        float v_ = 2 * atanf(1.f / r);
        float sinv = sinf((float)MO_PI - v_);
        x_ = sinv * sinf(u_);
        y_ = cosf((float)MO_PI - v_);
        z_ = sinv * cosf(u_);
        break;
    }
    case MO_WARP_COMPRESSED_PLANE:                            /* :400-424 */
    case MO_WARP_COMPRESSED_PLANE_PORTRAIT: {                 /* :441-465 */
        if (kind == MO_WARP_COMPRESSED_PLANE) u /= scale; else u /= - scale;
        v /= scale;
        float aatg = a * atanf(u / a);
        float u_ = aatg;
        float v_ = atanf(v * cosf(aatg) / b);
        float cosv = cosf(v_);
        if (kind == MO_WARP_COMPRESSED_PLANE) { x_ = cosv * sinf(u_); y_ = sinf(v_); }
        else                                  { y_ = cosv * sinf(u_); x_ = sinf(v_); }
        z_ = cosv * cosf(u_);
        break;
    }
    case MO_WARP_PANINI:                                      /* :489-512 */
    case MO_WARP_PANINI_PORTRAIT: {                           /* :536-552 */
        if (kind == MO_WARP_PANINI) u /= scale; else u /= - scale;
        v /= scale;
        float lamda = a * atanf(u / a);
        float u_ = lamda;
        float v_;
        if (fabs(lamda) > 1E-7)
            v_ = atanf(v * sinf(lamda) / (b * a * tanf(lamda / a)));
        else
            v_ = atanf(v / b);
        float cosv = cosf(v_);
        if (kind == MO_WARP_PANINI) { x_ = cosv * sinf(u_); y_ = sinf(v_); }
        else                        { y_ = cosv * sinf(u_); x_ = sinf(v_); }
        z_ = cosv * cosf(u_);
        break;
    }
    case MO_WARP_MERCATOR: {                                  /* :570-591 */
        u /= scale; v /= scale;
        float v_ = atanf(sinhf(v));
        float u_ = u;
        float cosv = cosf(v_);
        x_ = cosv * sinf(u_);
        y_ = sinf(v_);
        z_ = cosv * cosf(u_);
        break;
    }
    case MO_WARP_TRANSVERSE_MERCATOR: {                       /* :611-632 */
        u /= scale; v /= scale;
        float v_ = asinf(sinf(v) / coshf(u));
        float u_ = atan2f(sinhf(u), (float)cos(v));          /* cos(v): the double one, narrowed for atan2f */
        float cosv = cosf(v_);
        x_ = cosv * sinf(u_);
        y_ = sinf(v_);
        z_ = cosv * cosf(u_);
        break;
    }
    case MO_WARP_SPHERICAL_PORTRAIT: {                        /* :655-682 */
        float uu = -u, vv = v;
        uu /= scale; vv /= scale;
        float sinv = sinf((float)MO_PI - vv);
        float x0_ = sinv * sinf(uu);
        float y0_ = cosf((float)MO_PI - vv);
        z_ = sinv * cosf(uu);
        x_ = y0_; y_ = x0_;
        break;
    }
    default: {                                                /* cylindrical portrait :705-730 */
        float uu = -u, vv = v;
        uu /= scale; vv /= scale;
        float x0_ = sinf(uu);
        float y0_ = vv;
        z_ = cosf(uu);
        x_ = y0_; y_ = x0_;
        break;
    }
    }
    *x = k_rinv[0] * x_ + k_rinv[1] * y_ + k_rinv[2] * z_;
    *y = k_rinv[3] * x_ + k_rinv[4] * y_ + k_rinv[5] * z_;
    z  = k_rinv[6] * x_ + k_rinv[7] * y_ + k_rinv[8] * z_;
    if (z > 0) { *x /= z; *y /= z; }
    else *x = *y = -1;
}

/* warpers_inl.hpp:206-300 */
void mo_map_forward(const mo_projector *p, int kind, float x, float y, float *u, float *v)
{
    const float *r_kinv = p->r_kinv, *t = p->t;
    float scale = p->scale;
    kind &= 0xFF;
    if (kind >= MO_WARP_FISHEYE) { map_forward_other(p, kind, x, y, u, v); return; }
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
    kind &= 0xFF;
    if (kind >= MO_WARP_FISHEYE) { map_backward_other(p, kind, u, v, x, y); return; }
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
    kind &= 0xFF;
    if (kind == MO_WARP_SPHERICAL_PORTRAIT) {
        /* warpers.cpp:389-430: border walk, then the two poles -- which lie on the x axis here */
        int tx, ty, bx, by;
        detect_by_border(p, kind, src_w, src_h, &b);
        tx = (int)b.tl_u; ty = (int)b.tl_v; bx = (int)b.br_u; by = (int)b.br_v;
        float tl_uf = (float)tx, tl_vf = (float)ty, br_uf = (float)bx, br_vf = (float)by;
        float x = p->rinv[0], y = p->rinv[3], z = p->rinv[6];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                float pole = (float)(MO_PI * p->scale);
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, pole);
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, pole);
            }
        }
        x = p->rinv[0]; y = -p->rinv[3]; z = p->rinv[6];
        if (y > 0.f) {
            float x_ = (p->k[0] * x + p->k[1] * y) / z + p->k[2];
            float y_ = p->k[4] * y / z + p->k[5];
            if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                tl_uf = fminf(tl_uf, 0.f); tl_vf = fminf(tl_vf, 0.f);
                br_uf = fmaxf(br_uf, 0.f); br_vf = fmaxf(br_vf, 0.f);
            }
        }
        b.tl_u = tl_uf; b.tl_v = tl_vf; b.br_u = br_uf; b.br_v = br_vf;
    } else if (kind == MO_WARP_CYLINDRICAL_PORTRAIT || kind == MO_WARP_PLANE_PORTRAIT) {
        /* detail/warpers.hpp:479-483,499-502: detectResultRoiByBorder */
        detect_by_border(p, kind, src_w, src_h, &b);
    } else if (kind >= MO_WARP_FISHEYE) {
        /* RotationWarperBase::detectResultRoi (warpers_inl.hpp:143-166): every source pixel */
        bounds_init(&b);
        for (int yy = 0; yy < src_h; ++yy)
            for (int xx = 0; xx < src_w; ++xx) {
                mo_map_forward(p, kind, (float)xx, (float)yy, &u, &v);
                bounds_add(&b, u, v);
            }
    } else if (kind == MO_WARP_PLANE) {
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

static void make_projector(mo_projector *p, int kind, float scale, const float K[9], const float R[9],
                           const float T[3])
{
    memset(p, 0, sizeof(*p));
    p->scale = scale;
    kind_ab(kind, &p->a, &p->b);
    mo_set_camera_params(p, K, R, T);
}

/* warpers_inl.hpp:129-140 / warpers.cpp:128-136 */
mo_rect mo_warp_roi(int kind, float scale, const float K[9], const float R[9], const float T[3],
                    int src_w, int src_h)
{
    mo_projector p;
    int tlx, tly, brx, bry;
    make_projector(&p, kind, scale, K, R, T);
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
    make_projector(&p, kind, scale, K, R, T);
    mo_detect_result_roi(&p, kind, src_w, src_h, &tlx, &tly, &brx, &bry);
    int w = brx - tlx + 1;
#pragma omp parallel for schedule(static)
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
