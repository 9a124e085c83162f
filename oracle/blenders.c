/*
 * blenders.c -- CPU ORACLE (test infrastructure): Blender / FeatherBlender /
 * MultiBandBlender and their helpers.
 *
 * Follows R64/src/blenders.cpp line by line:
 *   Blender::prepare/feed/blend                 :65-112
 *   FeatherBlender::prepare/feed/blend          :115-155
 *   MultiBandBlender ctor/prepare/feed/blend    :189-377
 *   normalizeUsingWeightMap                     :383-424
 *   createWeightMap                             :427-432
 *   createLaplacePyr (16S branch)               :477-488
 *   restoreImageFromLaplacePyr                  :520-530
 * WEIGHT_EPS = 1e-5f (:50).
 */
#include "mst_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static const float WEIGHT_EPS = 1e-5f;

#define MAX_LEVELS 32

struct mo_blender {
    int kind;
    /* Blender */
    int16_t *dst;            /* CV_16SC3, dst_roi sized */
    uint8_t *dst_mask;       /* CV_8U */
    mo_rect dst_roi;
    /* FeatherBlender */
    float sharpness;
    float *dst_weight_map;
    /* MultiBandBlender */
    int actual_num_bands, num_bands, weight_type;
    mo_rect dst_roi_final;
    int16_t *dst_pyr_laplace[MAX_LEVELS];  /* [0] aliases dst */
    void    *dst_band_weights[MAX_LEVELS]; /* float or int16 */
    int lvl_w[MAX_LEVELS], lvl_h[MAX_LEVELS];
};

/* ---- helpers --------------------------------------------------------------- */

/* :383-424 */
void mo_normalize_using_weight_map_32f(const float *weight, int16_t *src, int w, int h)
{
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
    for (int y = 0; y < h; ++y) {
        int16_t *row = src + (size_t)y * w * 3;
        const float *weight_row = weight + (size_t)y * w;
        for (int x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c)
                row[3 * x + c] = mo_trunc_s16(row[3 * x + c] / (weight_row[x] + WEIGHT_EPS));
    }
}

void mo_normalize_using_weight_map_16s(const int16_t *weight, int16_t *src, int w, int h)
{
    for (int y = 0; y < h; ++y) {
        const int16_t *weight_row = weight + (size_t)y * w;
        int16_t *row = src + (size_t)y * w * 3;
        for (int x = 0; x < w; ++x) {
            int wv = weight_row[x] + 1;
            for (int c = 0; c < 3; ++c)
                row[3 * x + c] = (int16_t)(uint16_t)(((row[3 * x + c] << 8) / wv) & 0xFFFF);
        }
    }
}

/* :427-432  distanceTransform(L1,3) then threshold(weight*sharpness, 1, TRUNC) */
void mo_create_weight_map(const uint8_t *mask, int w, int h, float sharpness, float *weight)
{
    mo_distance_transform_l1(mask, w, h, weight);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        float v = weight[i] * sharpness;        /* MatExpr scale -> convertTo: src*alpha + 0 */
        v = v + 0.f;
        weight[i] = v > 1.f ? 1.f : v;          /* THRESH_TRUNC */
    }
}

static void level_sizes(int w, int h, int n, int *lw, int *lh)
{
    lw[0] = w; lh[0] = h;
    for (int i = 1; i <= n; ++i) { lw[i] = (lw[i - 1] + 1) / 2; lh[i] = (lh[i - 1] + 1) / 2; }
}

/* :477-488 (img.depth() == CV_16S): pyr[0] = img; pyrDown chain; then pyrUp + subtract */
void mo_create_laplace_pyr_16s(int16_t **pyr, int w, int h, int cn, int num_levels)
{
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    level_sizes(w, h, num_levels, lw, lh);
    for (int i = 0; i < num_levels; ++i)
        mo_pyr_down_16s(pyr[i], lw[i], lh[i], cn, pyr[i + 1]);
    int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h * cn);
    for (int i = 0; i < num_levels; ++i) {
        mo_pyr_up_16s(pyr[i + 1], lw[i + 1], lh[i + 1], cn, tmp, lw[i], lh[i]);
        mo_subtract_16s(pyr[i], tmp, pyr[i], (size_t)lw[i] * lh[i] * cn);
    }
    free(tmp);
}

/* :520-530 */
void mo_restore_image_from_laplace_pyr(int16_t **pyr, int w, int h, int cn, int num_levels)
{
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    level_sizes(w, h, num_levels, lw, lh);
    int16_t *tmp = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h * cn);
    for (int i = num_levels; i > 0; --i) {
        mo_pyr_up_16s(pyr[i], lw[i], lh[i], cn, tmp, lw[i - 1], lh[i - 1]);
        mo_add_16s(tmp, pyr[i - 1], pyr[i - 1], (size_t)lw[i - 1] * lh[i - 1] * cn);
    }
    free(tmp);
}

/* ---- lifetime -------------------------------------------------------------- */

mo_blender *mo_blender_create(int kind, int num_bands, float sharpness, int weight_type)
{
    mo_blender *b = (mo_blender *)calloc(1, sizeof(*b));
    b->kind = kind;
    b->actual_num_bands = num_bands;     /* MultiBandBlender(try_gpu, num_bands = 5, CV_32F) hpp:101 */
    b->sharpness = sharpness;            /* FeatherBlender(sharpness = 0.02f) hpp:75 */
    b->weight_type = weight_type;
    return b;
}

static void release_internals(mo_blender *b)
{
    for (int i = 1; i < MAX_LEVELS; ++i) { free(b->dst_pyr_laplace[i]); b->dst_pyr_laplace[i] = NULL; }
    for (int i = 0; i < MAX_LEVELS; ++i) { free(b->dst_band_weights[i]); b->dst_band_weights[i] = NULL; }
    free(b->dst); b->dst = NULL; b->dst_pyr_laplace[0] = NULL;
    free(b->dst_mask); b->dst_mask = NULL;
    free(b->dst_weight_map); b->dst_weight_map = NULL;
}

void mo_blender_destroy(mo_blender *b)
{
    if (!b) return;
    release_internals(b);
    free(b);
}

/* ---- prepare --------------------------------------------------------------- */

/* :71-78 */
static void base_prepare(mo_blender *b, mo_rect dst_roi)
{
    release_internals(b);
    size_t n = (size_t)dst_roi.width * dst_roi.height;
    b->dst = (int16_t *)calloc(n * 3, sizeof(int16_t));
    b->dst_mask = (uint8_t *)calloc(n, 1);
    b->dst_roi = dst_roi;
}

void mo_blender_prepare_rect(mo_blender *b, mo_rect dst_roi)
{
    if (b->kind == MO_BLEND_FEATHER) {
        /* :115-120 */
        base_prepare(b, dst_roi);
        b->dst_weight_map = (float *)calloc((size_t)dst_roi.width * dst_roi.height, sizeof(float));
    } else if (b->kind == MO_BLEND_MULTI_BAND) {
        /* :203-233 */
        b->dst_roi_final = dst_roi;
        double max_len = (double)(dst_roi.width > dst_roi.height ? dst_roi.width : dst_roi.height);
        int crop = (int)ceil(log(max_len) / log(2.0));
        b->num_bands = b->actual_num_bands < crop ? b->actual_num_bands : crop;
        int m = 1 << b->num_bands;
        dst_roi.width += (m - dst_roi.width % m) % m;
        dst_roi.height += (m - dst_roi.height % m) % m;
        base_prepare(b, dst_roi);
        level_sizes(dst_roi.width, dst_roi.height, b->num_bands, b->lvl_w, b->lvl_h);
        size_t wsz = b->weight_type == MO_WEIGHT_32F ? sizeof(float) : sizeof(int16_t);
        b->dst_pyr_laplace[0] = b->dst;
        b->dst_band_weights[0] = calloc((size_t)dst_roi.width * dst_roi.height, wsz);
        for (int i = 1; i <= b->num_bands; ++i) {
            size_t n = (size_t)b->lvl_w[i] * b->lvl_h[i];
            b->dst_pyr_laplace[i] = (int16_t *)calloc(n * 3, sizeof(int16_t));
            b->dst_band_weights[i] = calloc(n, wsz);
        }
    } else {
        base_prepare(b, dst_roi);
    }
}

/* :65-68 */
void mo_blender_prepare(mo_blender *b, const int *corners_xy, const int *sizes_wh, int n)
{
    mo_blender_prepare_rect(b, mo_result_roi(corners_xy, sizes_wh, n));
}

/* ---- feed ------------------------------------------------------------------ */

/* :241-269, integer ROI alignment of MultiBandBlender::feed */
void mo_multiband_feed_geometry(mo_rect dst_roi, int num_bands, int tl_x, int tl_y, int w, int h,
                                int out[8])
{
    int gap = 3 * (1 << num_bands);
    int roi_brx = dst_roi.x + dst_roi.width, roi_bry = dst_roi.y + dst_roi.height;
    int tlnx = dst_roi.x > tl_x - gap ? dst_roi.x : tl_x - gap;
    int tlny = dst_roi.y > tl_y - gap ? dst_roi.y : tl_y - gap;
    int brnx = roi_brx < tl_x + w + gap ? roi_brx : tl_x + w + gap;
    int brny = roi_bry < tl_y + h + gap ? roi_bry : tl_y + h + gap;
    int m = 1 << num_bands;
    tlnx = dst_roi.x + (((tlnx - dst_roi.x) >> num_bands) << num_bands);
    tlny = dst_roi.y + (((tlny - dst_roi.y) >> num_bands) << num_bands);
    int width = brnx - tlnx, height = brny - tlny;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    brnx = tlnx + width;
    brny = tlny + height;
    int dy = brny - roi_bry > 0 ? brny - roi_bry : 0;
    int dx = brnx - roi_brx > 0 ? brnx - roi_brx : 0;
    tlnx -= dx; brnx -= dx;
    tlny -= dy; brny -= dy;
    out[0] = tlnx; out[1] = tlny; out[2] = brnx; out[3] = brny;
    out[4] = tl_y - tlny;            /* top */
    out[5] = tl_x - tlnx;            /* left */
    out[6] = brny - tl_y - h;        /* bottom */
    out[7] = brnx - tl_x - w;        /* right */
}

static void multiband_feed(mo_blender *b, const int16_t *img, const uint8_t *mask, int w, int h,
                           int tl_x, int tl_y)
{
    int g[8];
    int nb = b->num_bands;
    mo_multiband_feed_geometry(b->dst_roi, nb, tl_x, tl_y, w, h, g);
    int top = g[4], left = g[5], bottom = g[6], right = g[7];
    int bw = w + left + right, bh = h + top + bottom;
    int lw[MAX_LEVELS], lh[MAX_LEVELS];
    level_sizes(bw, bh, nb, lw, lh);

    /* :271-279 source Laplacian pyramid */
    int16_t *src_pyr[MAX_LEVELS];
    for (int i = 0; i <= nb; ++i)
        src_pyr[i] = (int16_t *)malloc(sizeof(int16_t) * (size_t)lw[i] * lh[i] * 3);
    mo_copy_make_border_16s(img, w, h, 3, src_pyr[0], top, bottom, left, right, MO_BORDER_REFLECT);
    mo_create_laplace_pyr_16s(src_pyr, bw, bh, 3, nb);

    /* :281-298 weight Gaussian pyramid */
    void *wpyr[MAX_LEVELS];
    if (b->weight_type == MO_WEIGHT_32F) {
        float *wm = (float *)malloc(sizeof(float) * (size_t)w * h);
        const float alpha = (float)(1. / 255.);      /* convertTo(CV_32F, 1./255.): src*alpha + 0 */
        for (size_t i = 0; i < (size_t)w * h; ++i)
            wm[i] = mask[i] * alpha + 0.f;
        for (int i = 0; i <= nb; ++i)
            wpyr[i] = malloc(sizeof(float) * (size_t)lw[i] * lh[i]);
        mo_copy_make_border_32f(wm, w, h, (float *)wpyr[0], top, bottom, left, right, MO_BORDER_CONSTANT);
        free(wm);
        for (int i = 0; i < nb; ++i)
            mo_pyr_down_32f((const float *)wpyr[i], lw[i], lh[i], 1, (float *)wpyr[i + 1]);
    } else {
        int16_t *wm = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h);
        for (size_t i = 0; i < (size_t)w * h; ++i)
            wm[i] = (int16_t)(mask[i] + (mask[i] != 0));   /* convertTo(16S); add 1 where mask != 0 */
        for (int i = 0; i <= nb; ++i)
            wpyr[i] = malloc(sizeof(int16_t) * (size_t)lw[i] * lh[i]);
        mo_copy_make_border_16s(wm, w, h, 1, (int16_t *)wpyr[0], top, bottom, left, right, MO_BORDER_CONSTANT);
        free(wm);
        for (int i = 0; i < nb; ++i)
            mo_pyr_down_16s((const int16_t *)wpyr[i], lw[i], lh[i], 1, (int16_t *)wpyr[i + 1]);
    }

    int y_tl = g[1] - b->dst_roi.y, y_br = g[3] - b->dst_roi.y;
    int x_tl = g[0] - b->dst_roi.x, x_br = g[2] - b->dst_roi.x;

    /* :304-356 weighted accumulate */
    for (int i = 0; i <= nb; ++i) {
        int dw = b->lvl_w[i];
#pragma omp parallel for if (mo_get_num_threads() > 1) num_threads(mo_get_num_threads())
        for (int y = y_tl; y < y_br; ++y) {
            int y_ = y - y_tl;
            const int16_t *src_row = src_pyr[i] + (size_t)y_ * lw[i] * 3;
            int16_t *dst_row = b->dst_pyr_laplace[i] + (size_t)y * dw * 3;
            if (b->weight_type == MO_WEIGHT_32F) {
                const float *weight_row = (const float *)wpyr[i] + (size_t)y_ * lw[i];
                float *dst_weight_row = (float *)b->dst_band_weights[i] + (size_t)y * dw;
                for (int x = x_tl; x < x_br; ++x) {
                    int x_ = x - x_tl;
                    for (int c = 0; c < 3; ++c) {
                        int16_t add = mo_trunc_s16(src_row[3 * x_ + c] * weight_row[x_]);
                        dst_row[3 * x + c] = (int16_t)(uint16_t)((dst_row[3 * x + c] + add) & 0xFFFF);
                    }
                    dst_weight_row[x] += weight_row[x_];
                }
            } else {
                const int16_t *weight_row = (const int16_t *)wpyr[i] + (size_t)y_ * lw[i];
                int16_t *dst_weight_row = (int16_t *)b->dst_band_weights[i] + (size_t)y * dw;
                for (int x = x_tl; x < x_br; ++x) {
                    int x_ = x - x_tl;
                    for (int c = 0; c < 3; ++c) {
                        int16_t add = (int16_t)(uint16_t)(((src_row[3 * x_ + c] * weight_row[x_]) >> 8) & 0xFFFF);
                        dst_row[3 * x + c] = (int16_t)(uint16_t)((dst_row[3 * x + c] + add) & 0xFFFF);
                    }
                    dst_weight_row[x] = (int16_t)(uint16_t)((dst_weight_row[x] + weight_row[x_]) & 0xFFFF);
                }
            }
        }
        x_tl /= 2; y_tl /= 2;
        x_br /= 2; y_br /= 2;
    }
    for (int i = 0; i <= nb; ++i) { free(src_pyr[i]); free(wpyr[i]); }
}

void mo_blender_feed(mo_blender *b, const int16_t *img, const uint8_t *mask, int w, int h,
                     int tl_x, int tl_y)
{
    if (b->kind == MO_BLEND_MULTI_BAND) {
        multiband_feed(b, img, mask, w, h, tl_x, tl_y);
        return;
    }
    int dx = tl_x - b->dst_roi.x, dy = tl_y - b->dst_roi.y;
    int dw = b->dst_roi.width;
    if (b->kind == MO_BLEND_FEATHER) {
        /* :123-146 */
        float *weight_map = (float *)malloc(sizeof(float) * (size_t)w * h);
        mo_create_weight_map(mask, w, h, b->sharpness, weight_map);
        for (int y = 0; y < h; ++y) {
            const int16_t *src_row = img + (size_t)y * w * 3;
            int16_t *dst_row = b->dst + (size_t)(dy + y) * dw * 3;
            const float *weight_row = weight_map + (size_t)y * w;
            float *dst_weight_row = b->dst_weight_map + (size_t)(dy + y) * dw;
            for (int x = 0; x < w; ++x) {
                for (int c = 0; c < 3; ++c) {
                    int16_t add = mo_trunc_s16(src_row[3 * x + c] * weight_row[x]);
                    dst_row[3 * (dx + x) + c] = (int16_t)(uint16_t)((dst_row[3 * (dx + x) + c] + add) & 0xFFFF);
                }
                dst_weight_row[dx + x] += weight_row[x];
            }
        }
        free(weight_map);
        return;
    }
    /* :81-102 */
    for (int y = 0; y < h; ++y) {
        const int16_t *src_row = img + (size_t)y * w * 3;
        int16_t *dst_row = b->dst + (size_t)(dy + y) * dw * 3;
        const uint8_t *mask_row = mask + (size_t)y * w;
        uint8_t *dst_mask_row = b->dst_mask + (size_t)(dy + y) * dw;
        for (int x = 0; x < w; ++x) {
            if (mask_row[x])
                memcpy(dst_row + 3 * (dx + x), src_row + 3 * x, 3 * sizeof(int16_t));
            dst_mask_row[dx + x] |= mask_row[x];
        }
    }
}

/* ---- blend ----------------------------------------------------------------- */

/* :105-112 -- hands out dst_/dst_mask_ and releases internals */
static mo_rect base_blend(mo_blender *b, int16_t **dst, uint8_t **dst_mask, int w, int h, int stride_px)
{
    /* (w,h) is the cropped view into buffers with row pitch stride_px */
    int16_t *out = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h * 3);
    uint8_t *outm = (uint8_t *)malloc((size_t)w * h);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            uint8_t m = b->dst_mask[(size_t)y * stride_px + x];
            outm[(size_t)y * w + x] = m;
            for (int c = 0; c < 3; ++c)
                out[((size_t)y * w + x) * 3 + c] = m ? b->dst[((size_t)y * stride_px + x) * 3 + c] : 0;
        }
    *dst = out;
    *dst_mask = outm;
    mo_rect r = { b->dst_roi.x, b->dst_roi.y, w, h };
    release_internals(b);
    return r;
}

mo_rect mo_blender_blend(mo_blender *b, int16_t **dst, uint8_t **dst_mask)
{
    int W = b->dst_roi.width, H = b->dst_roi.height;
    if (b->kind == MO_BLEND_FEATHER) {
        /* :149-154 */
        mo_normalize_using_weight_map_32f(b->dst_weight_map, b->dst, W, H);
        for (size_t i = 0; i < (size_t)W * H; ++i)
            b->dst_mask[i] = b->dst_weight_map[i] > WEIGHT_EPS ? 255 : 0;
        return base_blend(b, dst, dst_mask, W, H, W);
    }
    if (b->kind == MO_BLEND_MULTI_BAND) {
        /* :359-377 */
        for (int i = 0; i <= b->num_bands; ++i) {
            if (b->weight_type == MO_WEIGHT_32F)
                mo_normalize_using_weight_map_32f((const float *)b->dst_band_weights[i],
                                                  b->dst_pyr_laplace[i], b->lvl_w[i], b->lvl_h[i]);
            else
                mo_normalize_using_weight_map_16s((const int16_t *)b->dst_band_weights[i],
                                                  b->dst_pyr_laplace[i], b->lvl_w[i], b->lvl_h[i]);
        }
        mo_restore_image_from_laplace_pyr(b->dst_pyr_laplace, W, H, 3, b->num_bands);
        for (size_t i = 0; i < (size_t)W * H; ++i) {
            /* dst_mask_ = dst_band_weights_[0] > WEIGHT_EPS (compare converts eps to the weight type) */
            if (b->weight_type == MO_WEIGHT_32F)
                b->dst_mask[i] = ((const float *)b->dst_band_weights[0])[i] > WEIGHT_EPS ? 255 : 0;
            else
                b->dst_mask[i] = (double)((const int16_t *)b->dst_band_weights[0])[i] > (double)WEIGHT_EPS ? 255 : 0;
        }
        return base_blend(b, dst, dst_mask, b->dst_roi_final.width, b->dst_roi_final.height, W);
    }
    return base_blend(b, dst, dst_mask, W, H, W);
}

/* ---- introspection ----------------------------------------------------------- */

int mo_blender_num_bands(const mo_blender *b) { return b->num_bands; }
mo_rect mo_blender_dst_roi(const mo_blender *b) { return b->dst_roi; }

const int16_t *mo_blender_band(const mo_blender *b, int level, int *w, int *h)
{
    if (b->kind != MO_BLEND_MULTI_BAND) {
        *w = b->dst_roi.width; *h = b->dst_roi.height;
        return b->dst;
    }
    *w = b->lvl_w[level]; *h = b->lvl_h[level];
    return b->dst_pyr_laplace[level];
}

const float *mo_blender_band_weight(const mo_blender *b, int level, int *w, int *h)
{
    if (b->kind == MO_BLEND_FEATHER) {
        *w = b->dst_roi.width; *h = b->dst_roi.height;
        return b->dst_weight_map;
    }
    *w = b->lvl_w[level]; *h = b->lvl_h[level];
    return (const float *)b->dst_band_weights[level];
}
