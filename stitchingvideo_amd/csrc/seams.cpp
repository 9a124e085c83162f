// seams.cpp -- the Voronoi seam finder on the device: PairwiseSeamFinder::run +
// VoronoiSeamFinder::findInPair, R64/src/seam_finders.cpp:74-85, 108-160.  Pairs are visited in the reference's order
// (i < j) and update the masks in place, so they are chained on one stream; each pair is three small launches around
// the separable exact L1 distance transform the feather weight maps use (kernels.hip, k_dt_columns / k_dt_rows).
#include <algorithm>
#include <vector>

#include "engine.hpp"
#include "kernels.hpp"

namespace mst {

void voronoi_seams(int device, int n, const int32_t* corners_xy, const int32_t* sizes_wh, uint8_t* const* masks,
                   const size_t* mask_strides)
{
    select_device(device);
    DevicePool pool;
    hipStream_t s = nullptr;
    check_hip(hipStreamCreate(&s), "hipStreamCreate");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } guard{s};
    std::vector<uint8_t*> d_mask(n);
    size_t scratch = 0;
    for (int i = 0; i < n; ++i) {
        const int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        if (w <= 0 || h <= 0 || !masks[i] || mask_strides[i] < static_cast<size_t>(w)) throw Error{MST_ERR_INVALID_ARG, "bad mask"};
        d_mask[i] = pool.alloc<uint8_t>(static_cast<size_t>(w) * h);
        check_hip(hipMemcpy2DAsync(d_mask[i], w, masks[i], mask_strides[i], w, h, hipMemcpyHostToDevice, s), "hipMemcpy2DAsync(mask)");
    }
    struct Pair { int i, j, x, y, w, h; };
    std::vector<Pair> pairs;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {       // overlapRoi, util.cpp:103-115
            const int x_tl = std::max(corners_xy[2 * i], corners_xy[2 * j]), y_tl = std::max(corners_xy[2 * i + 1], corners_xy[2 * j + 1]);
            const int x_br = std::min(corners_xy[2 * i] + sizes_wh[2 * i], corners_xy[2 * j] + sizes_wh[2 * j]);
            const int y_br = std::min(corners_xy[2 * i + 1] + sizes_wh[2 * i + 1], corners_xy[2 * j + 1] + sizes_wh[2 * j + 1]);
            if (x_tl < x_br && y_tl < y_br) {
                pairs.push_back(Pair{i, j, x_tl, y_tl, x_br - x_tl, y_br - y_tl});
                scratch = std::max(scratch, static_cast<size_t>(x_br - x_tl + 20) * (y_br - y_tl + 20));
            }
        }
    if (!pairs.empty()) {
        uint8_t* inv1 = pool.alloc<uint8_t>(scratch);
        uint8_t* inv2 = pool.alloc<uint8_t>(scratch);
        int32_t* tmp = pool.alloc<int32_t>(scratch);
        float* dist1 = pool.alloc<float>(scratch);
        float* dist2 = pool.alloc<float>(scratch);
        for (const Pair& p : pairs)
            launch_voronoi_pair(d_mask[p.i], sizes_wh[2 * p.i], sizes_wh[2 * p.i + 1], p.x - corners_xy[2 * p.i], p.y - corners_xy[2 * p.i + 1],
                                d_mask[p.j], sizes_wh[2 * p.j], sizes_wh[2 * p.j + 1], p.x - corners_xy[2 * p.j], p.y - corners_xy[2 * p.j + 1],
                                p.w, p.h, inv1, inv2, tmp, dist1, dist2, s);
        check_hip(hipGetLastError(), "voronoi kernels");
    }
    for (int i = 0; i < n; ++i) {
        const int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        check_hip(hipMemcpy2DAsync(masks[i], mask_strides[i], d_mask[i], w, w, h, hipMemcpyDeviceToHost, s), "hipMemcpy2DAsync(mask)");
    }
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(seams)");
}

}  // namespace mst
