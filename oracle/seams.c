/*
 * seams.c -- CPU ORACLE (test infrastructure): the Voronoi seam finder.
 *
 * Follows R64/src/seam_finders.cpp:74-85 (PairwiseSeamFinder::run: every overlapping pair i < j, in order, masks
 * updated in place) and :108-160 (VoronoiSeamFinder::findInPair: sub-masks with a 10-pixel gap, collision removed,
 * two L1 distance transforms, `dist1 < dist2` decides who keeps each overlap pixel).
 */
#include "mst_oracle.h"

#include <stdlib.h>
#include <string.h>

static void find_in_pair(uint8_t *mask1, int w1, int h1, int tl1x, int tl1y, uint8_t *mask2, int w2, int h2, int tl2x,
                         int tl2y, mo_rect roi)
{
    const int gap = 10;
    const int sw = roi.width + 2 * gap, sh = roi.height + 2 * gap;
    uint8_t *sub1 = (uint8_t *)malloc((size_t)sw * sh), *sub2 = (uint8_t *)malloc((size_t)sw * sh);
    uint8_t *inv1 = (uint8_t *)malloc((size_t)sw * sh), *inv2 = (uint8_t *)malloc((size_t)sw * sh);
    float *dist1 = (float *)malloc(sizeof(float) * (size_t)sw * sh), *dist2 = (float *)malloc(sizeof(float) * (size_t)sw * sh);
    for (int y = -gap; y < roi.height + gap; ++y)
        for (int x = -gap; x < roi.width + gap; ++x) {
            int y1 = roi.y - tl1y + y, x1 = roi.x - tl1x + x;
            sub1[(size_t)(y + gap) * sw + (x + gap)] = (y1 >= 0 && x1 >= 0 && y1 < h1 && x1 < w1) ? mask1[(size_t)y1 * w1 + x1] : 0;
            int y2 = roi.y - tl2y + y, x2 = roi.x - tl2x + x;
            sub2[(size_t)(y + gap) * sw + (x + gap)] = (y2 >= 0 && x2 >= 0 && y2 < h2 && x2 < w2) ? mask2[(size_t)y2 * w2 + x2] : 0;
        }
    for (size_t i = 0; i < (size_t)sw * sh; ++i) {
        int collision = sub1[i] != 0 && sub2[i] != 0;
        uint8_t u1 = collision ? 0 : sub1[i], u2 = collision ? 0 : sub2[i];
        inv1[i] = u1 == 0 ? 255 : 0;                 /* unique1 == 0 */
        inv2[i] = u2 == 0 ? 255 : 0;
    }
    mo_distance_transform_l1(inv1, sw, sh, dist1);
    mo_distance_transform_l1(inv2, sw, sh, dist2);
    for (int y = 0; y < roi.height; ++y)
        for (int x = 0; x < roi.width; ++x) {
            size_t i = (size_t)(y + gap) * sw + (x + gap);
            if (dist1[i] < dist2[i]) mask2[(size_t)(roi.y - tl2y + y) * w2 + (roi.x - tl2x + x)] = 0;
            else                     mask1[(size_t)(roi.y - tl1y + y) * w1 + (roi.x - tl1x + x)] = 0;
        }
    free(sub1); free(sub2); free(inv1); free(inv2); free(dist1); free(dist2);
}

/* masks[i]: sizes_wh[2i] x sizes_wh[2i+1], tightly packed, updated in place */
void mo_voronoi_seams(int n, const int *corners_xy, const int *sizes_wh, uint8_t *const *masks)
{
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            mo_rect roi;
            if (mo_overlap_roi(corners_xy[2 * i], corners_xy[2 * i + 1], corners_xy[2 * j], corners_xy[2 * j + 1],
                               sizes_wh[2 * i], sizes_wh[2 * i + 1], sizes_wh[2 * j], sizes_wh[2 * j + 1], &roi))
                find_in_pair(masks[i], sizes_wh[2 * i], sizes_wh[2 * i + 1], corners_xy[2 * i], corners_xy[2 * i + 1],
                             masks[j], sizes_wh[2 * j], sizes_wh[2 * j + 1], corners_xy[2 * j], corners_xy[2 * j + 1], roi);
        }
}
