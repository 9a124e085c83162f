// mb_plan.hpp -- per-calibration work lists of the fused multi-band kernels (host side).
//
// The blend weight of a camera is non-zero only near its seam mask (the mask dilated by the
// [1 4 6 4 1] support of each pyramid level).  Everything outside cannot reach the panorama:
// short(laplacian * 0.f) == 0 (R64/src/blenders.cpp:321-323).  The plan walks that dependency
// backwards once per calibration and keeps only the tiles that matter:
//
//   nz[l]    where W_c[l] != 0                                (read back from the device)
//   need[l]  Gaussian pixels of level l that some kept output reads:
//            need[L] = nz[L] | pyrUp-parents(nz[L-1])
//            need[l] = nz[l] | pyrUp-parents(nz[l-1]) | pyrDown-footprint(need[l+1])
//
// A camera that straddles the +-pi seam of a 360-degree canvas has a roi as wide as the canvas
// (detectResultRoiByBorder, detail/warpers_inl.hpp:169-203); with these lists it costs no more
// than any other camera.
#pragma once

#include <cstdint>
#include <vector>

#include "device_types.hpp"

namespace mst {

struct MbCameraNeed {
    // [level] -> lh*lw bytes
    std::vector<std::vector<uint8_t>> nz, need;
};

struct MbPlanHost {
    std::vector<uint32_t> warp_tiles;                         // kernel A, grouped by camera: cam << 24 | ty << 12 | tx, bit 31 =
                                                              // level 0 of the tile is not stored (no level-0 weight under it)
    std::vector<int> warp_cam_start;                          // n+1 offsets into warp_tiles
    // A step that starts on a large level (base <= 2) is run by the streaming kernel instead, one launch per level:
    // units[j] = the mb_stream_unit_cols() x mb_stream_unit_rows() output units of level base + 1 + j that hold a needed pixel
    // (cam << 24 | uy << 12 | ux, grouped by camera: unit_cam_start[j]); empty when the step stays with k_pyr_chain.
    // `tiles` always describes what the step computes (the camera-sharded payload is cut from it).
    struct ChainStep {
        int base, k;
        std::vector<uint32_t> tiles; std::vector<int> cam_start;
        std::vector<uint32_t> units[2]; std::vector<int> unit_cam_start[2];
    };
    std::vector<ChainStep> chain;                             // kernel B launches
    std::vector<uint32_t> block_cams;                         // k_blend_fine, one entry per panorama tile (BlendArgs)
    std::vector<uint32_t> block_order;                        // k_blend_fine, tile index launched as block i
    int blend_tiles_x = 0, blend_tiles_y = 0;
    // k_blend_coarse tiles for each tile height the launcher chooses between (index 0: 16 level-1 rows, for one or two
    // frame sets per launch; 1: 32 rows, for batches): per tile the camera bits over levels 1 .. top (bit 31: every
    // level owned by the one camera), the levels owned by one camera | owner << 16, the launch order, the LDS layout
    struct CoarseTiles {
        int th = 0, tiles_x = 0, tiles_y = 0;
        std::vector<uint32_t> cams, pure, order;
        // strip engines with three and more levels: k_blend_coarse's own launch list -- `order` plus the tiles whose
        // level-2 interiors the tiles of `order` read as pyrUp halo (k_blend_mid); empty: the same list
        std::vector<uint32_t> order2;
        int lds_off[kMaxLevels] = {};
        size_t lds_bytes = 0;
    };
    CoarseTiles coarse[2];
    // camera-sharded hand-off: per camera, the interiors of its launched tiles at every level (what the canvas kernel
    // can read of that camera and nothing else), packed back to back
    std::vector<PayloadRect> payload_rects;                   // grouped by camera
    std::vector<int> payload_cam_start;                       // n+1 offsets into payload_rects
    std::vector<uint32_t> payload_px;                         // pixels (dwords) per camera
    std::vector<MbCameraNeed> cams;
    int clip_x0 = 0, clip_x1 = 0;                             // strip engines: BlendArgs::clip_x0 / clip_x1 (0, 0: whole canvas)
};

// weight_class[c][l]: one byte per pixel of camera c's weight pyramid level l -- bit 0: the weight is non-zero,
// bit 1: it is exactly the weight of a fully owned pixel (1.0f; 256 with CV_16S weights) -- classified on the
// device (k_weight_class), so that the plan never needs the float pyramids on the host.  `blend` must have
// nlev/cw/ch/final_w/final_h filled in.  keep_level0 (MST_FLAG_DEBUG): store the warped level 0 of every launched tile,
// also of those no panorama pixel reads it from (mst_debug_read then shows all of it).
// strip_x0 / strip_x1: the columns [x0, x1) of the (padded) canvas this engine composes (canvas split over GPUs); every
// list is restricted to what those panorama columns can depend on.  0 / cw[0]: everything.
void build_mb_plan(const std::vector<CamDesc>& desc, const std::vector<std::vector<std::vector<uint8_t>>>& weight_class,
                   const BlendArgs& blend, bool keep_level0, MbPlanHost& plan, int strip_x0 = 0, int strip_x1 = 1 << 30);

}  // namespace mst
