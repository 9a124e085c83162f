/*
 * compose.c -- CPU ORACLE (test infrastructure): the per-frame compose loop.
 *
 * mo_compose_panorama follows R64/src/stitcher.cpp:264-313 (== CLI:705-760): per image
 * warp (INTER_LINEAR, BORDER_REFLECT) -> exposure apply -> convertTo(CV_16S) ->
 * blender feed; then blend and convertTo(CV_8U).  The per-camera warp/gain/convert
 * stage runs under `#pragma omp parallel for` like the driver's frame loop (DRV:748-756).
 * The final mask (stitcher.cpp:291-294) is an input (static per calibration).
 *
 * mo_lut_* follow the driver's own fast path: prepareSize/feedSize (DRV:94-148),
 * UpdateMat (DRV:696-722) and feedSizeRemap (DRV:150-177) + convertTo(CV_8U) (DRV:762).
 */
#include "mst_oracle.h"

#include <stdlib.h>
#include <string.h>

mo_rect mo_compose_panorama(const mo_compose_config *cfg, const mo_compose_camera *cams, int n,
                            uint8_t **pano, uint8_t **pano_mask)
{
    int *corners = (int *)malloc(sizeof(int) * 2 * n);
    int *sizes = (int *)malloc(sizeof(int) * 2 * n);
    int16_t **warped_s = (int16_t **)calloc(n, sizeof(int16_t *));
    int nthreads = cfg->num_threads > 0 ? cfg->num_threads : 1;

    /* stitcher.cpp:227-263: corners/sizes from warpRoi */
    for (int i = 0; i < n; ++i) {
        mo_rect roi = mo_warp_roi(cfg->warp_kind, cfg->scale, cams[i].K, cams[i].R, cams[i].T,
                                  cams[i].src_w, cams[i].src_h);
        corners[2 * i] = roi.x; corners[2 * i + 1] = roi.y;
        sizes[2 * i] = roi.width; sizes[2 * i + 1] = roi.height;
    }

#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) if (nthreads > 1)
    for (int i = 0; i < n; ++i) {
        int w = sizes[2 * i], h = sizes[2 * i + 1];
        uint8_t *img_warped = (uint8_t *)malloc((size_t)w * h * 3);
        /* :275 w->warp(img, K, R, INTER_LINEAR, BORDER_REFLECT, img_warped) */
        if (cams[i].lens_map1 && cams[i].lens_map2) {
            /* the fused lens variant (mst_oracle.h: mo_compose_lens_maps): buildMaps, compose, one remap of the raw frame */
            float *xm = (float *)malloc(sizeof(float) * (size_t)w * h), *ym = (float *)malloc(sizeof(float) * (size_t)w * h);
            mo_build_maps(cfg->warp_kind, cfg->scale, cams[i].K, cams[i].R, cams[i].T, cams[i].src_w, cams[i].src_h, xm, ym);
            mo_compose_lens_maps(xm, ym, (size_t)w * h, cams[i].lens_map1, cams[i].lens_map2, cams[i].src_w, cams[i].src_h);
            mo_remap_8u(cams[i].frame, cams[i].src_w, cams[i].src_h, cams[i].src_stride, 3, img_warped, w, h, (size_t)w * 3,
                        xm, ym, (size_t)w, MO_INTER_LINEAR, MO_BORDER_REFLECT);
            free(xm); free(ym);
        } else
        mo_warp_8u(cfg->warp_kind, cfg->scale, cams[i].K, cams[i].R, cams[i].T,
                   cams[i].frame, cams[i].src_w, cams[i].src_h, cams[i].src_stride, 3,
                   MO_INTER_LINEAR, MO_BORDER_REFLECT, img_warped);
        /* :283 exposure_comp_->apply */
        if (cfg->expos_kind == MO_EXPOS_GAIN_BLOCKS)
            mo_blocks_gain_apply(img_warped, w, h, (size_t)w * 3, cams[i].gain_map,
                                 cams[i].gain_w, cams[i].gain_h);
        else if (cfg->expos_kind == MO_EXPOS_GAIN)
            mo_gain_apply(img_warped, w, h, (size_t)w * 3, 3, cams[i].gain);
        /* :285 convertTo(CV_16S) */
        warped_s[i] = (int16_t *)malloc(sizeof(int16_t) * (size_t)w * h * 3);
        mo_convert_8u_16s(img_warped, (size_t)w * 3, warped_s[i], (size_t)w * 3, w * 3, h);
        free(img_warped);
    }

    int prev_threads = mo_get_num_threads();
    mo_set_num_threads(nthreads);
    mo_blender *b = mo_blender_create(cfg->blend_kind, cfg->num_bands, cfg->sharpness, cfg->weight_type);
    mo_blender_prepare(b, corners, sizes, n);                       /* :296-300 */
    for (int i = 0; i < n; ++i) {
        mo_blender_feed(b, warped_s[i], cams[i].mask, sizes[2 * i], sizes[2 * i + 1],
                        corners[2 * i], corners[2 * i + 1]);        /* :303 */
        free(warped_s[i]);
    }
    int16_t *result; uint8_t *result_mask;
    mo_rect r = mo_blender_blend(b, &result, &result_mask);         /* :307 */
    mo_blender_destroy(b);
    mo_set_num_threads(prev_threads);

    /* :313 result.convertTo(pano_, CV_8U) */
    uint8_t *out = (uint8_t *)malloc((size_t)r.width * r.height * 3);
    mo_convert_16s_8u(result, (size_t)r.width * 3, out, (size_t)r.width * 3, r.width * 3, r.height);
    mo_free(result);
    *pano = out;
    if (pano_mask) *pano_mask = result_mask; else mo_free(result_mask);
    free(corners); free(sizes); free(warped_s);
    return r;
}

/* ---- driver LUT path --------------------------------------------------------- */

struct mo_lut {
    mo_rect roi;
    int n;
    float *xmapdet, *ymapdet, *indxmap;   /* CV_32F planes, DRV:107-112 */
    uint8_t *mask;
    int *sizes;                           /* per-camera (w, h) */
};

/* prepareSize (DRV:94-115) + feedSize for every camera in order (DRV:117-148, :682-684) */
mo_lut *mo_lut_build(const int *corners_xy, const int *sizes_wh, const uint8_t *const *masks, int n)
{
    mo_lut *l = (mo_lut *)calloc(1, sizeof(*l));
    l->roi = mo_result_roi(corners_xy, sizes_wh, n);
    l->n = n;
    size_t npx = (size_t)l->roi.width * l->roi.height;
    l->xmapdet = (float *)calloc(npx, sizeof(float));
    l->ymapdet = (float *)calloc(npx, sizeof(float));
    l->indxmap = (float *)calloc(npx, sizeof(float));
    l->mask = (uint8_t *)calloc(npx, 1);
    l->sizes = (int *)malloc(sizeof(int) * 2 * n);
    memcpy(l->sizes, sizes_wh, sizeof(int) * 2 * n);
    for (int i = 0; i < n; ++i) {
        int dx = corners_xy[2 * i] - l->roi.x, dy = corners_xy[2 * i + 1] - l->roi.y;
        int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        for (int y = 0; y < h; ++y) {
            size_t drow = (size_t)(dy + y) * l->roi.width;
            const uint8_t *mask_row = masks[i] + (size_t)y * w;
            for (int x = 0; x < w; ++x) {
                if (mask_row[x]) {
                    l->xmapdet[drow + dx + x] = (float)x;
                    l->ymapdet[drow + dx + x] = (float)y;
                    l->indxmap[drow + dx + x] = (float)i;
                }
                l->mask[drow + dx + x] |= mask_row[x];
            }
        }
    }
    return l;
}

mo_rect mo_lut_canvas(const mo_lut *l) { return l->roi; }

/* UpdateMat sizes ResultStitch (DRV:702); feedSizeRemap gathers (DRV:150-177); DRV:762 to 8U */
mo_rect mo_lut_compose(const mo_lut *l, const int16_t *const *imgs, float upblack, float downblack,
                       float leftblack, float rightblack, uint8_t **pano)
{
    /* Size(dst_roi_1.width - leftblack - rightblack, dst_roi_1.height * (1 - upblack - downblack)):
     * int - float - float -> float, int * float -> float, both truncated by Size_<int>(float) */
    int out_w = (int)((float)l->roi.width - leftblack - rightblack);
    int out_h = (int)((float)l->roi.height * (1 - upblack - downblack));
    /* int yy = dst_.rows/(1 - upblack - downblack)*upblack; */
    int yy = (int)((float)out_h / (1 - upblack - downblack) * upblack);
    uint8_t *out = (uint8_t *)malloc((size_t)out_w * out_h * 3);
    for (int y = 0; y < out_h; ++y) {
        size_t lrow = (size_t)(y + yy) * l->roi.width;
        for (int x = 0; x < out_w; ++x) {
            int xx = (int)((float)x + leftblack);
            int cam = (int)l->indxmap[lrow + xx];
            int sy = (int)l->ymapdet[lrow + xx], sx = (int)l->xmapdet[lrow + xx];
            const int16_t *px = imgs[cam] + ((size_t)sy * l->sizes[2 * cam] + sx) * 3;
            for (int c = 0; c < 3; ++c)
                out[((size_t)y * out_w + x) * 3 + c] = mo_sat_u8(px[c]);
        }
    }
    *pano = out;
    mo_rect r = { l->roi.x + (int)leftblack, l->roi.y + yy, out_w, out_h };
    return r;
}

void mo_lut_destroy(mo_lut *l)
{
    if (!l) return;
    free(l->xmapdet); free(l->ymapdet); free(l->indxmap); free(l->mask); free(l->sizes);
    free(l);
}
