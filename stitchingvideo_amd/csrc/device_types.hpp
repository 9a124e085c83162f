// device_types.hpp -- plain structs shared by the host engine and the HIP kernels.
#pragma once

#include <cstddef>
#include <cstdint>

namespace mst {

constexpr int kMaxCams = 16;
constexpr int kMaxLevels = 11;   // MST_MAX_BANDS + 1

// Level 0 of a camera's pyramid in memory: packed BGR8 (3 bytes a pixel, round 4) or a dword a pixel like the levels above
// (B | G << 8 | R << 16, rounds 1-3).  A compile-time switch so that the two can be timed in one run on one box
// (tools/variants.sh "dword:-DMST_L0_PACKED=0"; profiles/r05_experiments.md); both are parity-tested.  Rows are whole groups
// of four pixels either way (12 or 16 bytes), so every group the warp kernel stores is aligned and inside its row.
#ifndef MST_L0_PACKED
#define MST_L0_PACKED 1
#endif
constexpr int kL0Bpp = MST_L0_PACKED ? 3 : 4;      // bytes per level-0 pixel

// Per roi column / row constants of the backward map and of the gain field, laid out so that a
// thread fetches everything it needs for one axis with 16-byte loads.
//
// mapBackward forms  X_i = k_rinv[3i]*x_ + k_rinv[3i+1]*y_ + k_rinv[3i+2]*z_  evaluated left to right,
// i.e. (A_i + B_i) + C_i.  B_i depends only on the row for all three projections; A_i and C_i depend
// only on the column for the plane and cylindrical projections (for the spherical one x_ = ra*ca and
// z_ = ra*cb need one multiply per pixel first).  The products are the same IEEE float products the
// reference forms, just hoisted.
struct ColEntry {          // one per roi column (48 bytes)
    float ca, cb;          // backward-map column terms (geometry.hpp MapTables)
    float a[3];            // k_rinv[0,3,6] * x_      (plane / cylindrical)
    float c[3];            // k_rinv[2,5,8] * z_      (plane: * (1 - t2); cylindrical: * cb)
    int32_t gx0, gx1;      // gain map columns (cv::resize xofs)
    float gxw0, gxw1;      // their weights
};
struct RowEntry {          // one per roi row (32 bytes)
    float ra;              // spherical: sinf(pi - v/scale)
    float b[3];            // k_rinv[1,4,7] * y_
    int32_t gy0, gy1;      // gain map rows (cv::resize yofs, clipped into the map)
    float gyw0, gyw1;      // their weights
};

// One camera of one calibration state.  Lives in device memory as an array; kernels index it
// with a block-uniform camera id so every field is fetched with scalar loads.
struct CamDesc {
    // ---- backward map (a3): per-axis tables + k_rinv, evaluated in registers
    int32_t kind;
    int32_t roi_w, roi_h;          // warpRoi size (sizes[i])
    int32_t src_w, src_h;
    float   kr[9];                 // k_rinv
    float   one_minus_t2;          // (1 - t[2]) of the plane projector
    const float* col_a; const float* col_b;
    const float* row_a; const float* row_b;
    // non-separable projectors (kind >= 3): the backward map itself, roi_h x roi_w floats each, evaluated on the
    // host per calibration (RotationWarperBase::buildMaps); nullptr for plane / spherical / cylindrical
    const float* xmap; const float* ymap;
    const ColEntry* cols;          // same data, packed per column (fused kernels)
    const RowEntry* rows;
    // ---- exposure (a7)
    int32_t expos;                 // MST_EXPOS_*
    float   gain;                  // GAIN
    const float* gain_map;         // GAIN_BLOCKS: gh x gw
    int32_t gw, gh;
    const int32_t* gx0; const int32_t* gx1; const float* gxw0; const float* gxw1;   // per roi column
    const int32_t* gy0; const int32_t* gy1; const float* gyw0; const float* gyw1;   // per roi row
    // ---- static per-calibration rasters
    const uint8_t* mask;           // final mask_warped, roi_h x roi_w, tight
    const uint8_t* mask_b;         // the same mask inside the bordered level-0 window (lh[0] x lw[0], zeros around
                                   // the roi = copyMakeBorder(BORDER_CONSTANT)); multi-band only
    const float*   feather_w;      // createWeightMap(mask), roi_h x roi_w (feather only)
    int32_t corner_x, corner_y;    // corners[i] - canvas.tl  (feather / paste)
    // ---- multi-band (a15-a18): bordered image geometry and pyramids
    int32_t top, left;             // copyMakeBorder offsets
    int32_t nlev;                  // num_bands + 1
    int32_t lw[kMaxLevels], lh[kMaxLevels];   // bordered pyramid level sizes
    int32_t ox[kMaxLevels], oy[kMaxLevels];   // window origin inside the canvas level
    // Gaussian pyramid.  Every level of a CV_8U-sourced pyramid stays in [0,255] ((sum+128)>>8 of
    // bytes), so a pixel of levels 1 .. is stored as one dword B | G<<8 | R<<16 instead of three shorts.  Level 0 -- written
    // once by the warp kernel and read once by k_blend_fine, the two largest streams of the path -- is packed BGR8 (round 4):
    // three bytes a pixel, rows l0_pitch bytes apart (a multiple of 12: whole groups of four pixels, dword-aligned rows).
    uint32_t* G[kMaxLevels];       // G[l >= 1]: lh x lw dwords; G[0]: lh[0] rows of l0_pitch bytes = warped + gained + bordered level 0
    int32_t l0_pitch;
    const float* W[kMaxLevels];    // weight pyramid, lh x lw floats (W[0] = mask/255 with border)
};

struct FrameArgs {                 // per-frame source pointers, passed by value
    const uint8_t* ptr[kMaxCams];
    uint32_t stride[kMaxCams];
};

// Frame sets of one batched launch (mst_compose_device_batch): set f is handled by the blocks with
// blockIdx.y == f, against the f-th copy of the camera descriptors (own Gaussian pyramids).
constexpr int kMaxBatch = 8;
struct FrameBatch {
    FrameArgs f[kMaxBatch];
};

// ---- single-pass composites (Blender::NO, the driver's LUT, FeatherBlender): kernels_compose.hip ------------------
// The output (the canvas, or its cropped part on the LUT path) is cut into kCompTW x kCompTH tiles.  A tile is fed by
// one SEGMENT per camera that contributes to it; a segment carries, per tile pixel, the backward map in cv::remap's
// fixed-point form relative to the segment's window origin (one dword), the exposure gain (one float) and, for the
// feather blend, the camera's blend weight (one float).  All of it is frame-invariant: built by mst_update.
constexpr int kCompTW = 128, kCompTH = 32, kCompPx = kCompTW * kCompTH;
constexpr int kCompModePaste = 0, kCompModeLut = 1, kCompModeFeather = 2;
struct ComposeSeg {
    int32_t cam;
    int32_t bx, by;              // smallest window coordinate over the segment's pixels (source pixels)
    uint32_t flags;              // kCompSegUnit, kCompSegFill
    // the source window of the segment as it is staged in LDS (kernels_compose.hip): columns [wx0, wx0 + 4 * groups)
    // with wx0 = bx rounded down to a multiple of four pixels, rows [by, by + rows)
    int32_t wx0, groups, rows, pad;
};
struct ComposeTile {
    uint32_t seg_first;          // first segment (and, * kCompPx, first table entry) of the tile
    uint32_t seg_count;
    uint32_t flags;              // bit 0: some pixel needs the general per-pixel path (border taps, huge gains, far windows)
    uint32_t tx_ty;              // tx | ty << 16
};
constexpr uint32_t kCompTileGeneral = 1u;      // some pixel needs the general per-pixel path
constexpr uint32_t kCompTileDirect = 2u;       // some segment's source window does not fit the LDS budget: gather from memory
constexpr uint32_t kCompTileNoGather = 4u;     // some window touches the frame's last pixels: 12-byte gathers would overrun, LDS only
constexpr uint32_t kCompSegUnit = 1u;
constexpr uint32_t kCompSegFill = 2u;          // LUT: the segment of the pixels nobody owns (camera 0 at roi (0, 0))
constexpr int kCompLdsPx = 10240;              // staged window budget per block, in 4-byte pixels (40 KB)
struct ComposeArgs {
    const CamDesc* cams;
    int32_t n_cams, mode;
    const ComposeTile* tiles;    // the launch's tile list, in launch order (most segments first)
    const ComposeSeg* segs;
    const uint32_t* table;       // per segment kCompPx entries: fx | fy << 6 | dx << 12 | dy << 22 | valid << 31
    const float* gain;           // per segment kCompPx floats
    const float* weight;         // feather: per segment kCompPx floats
    const uint8_t* mask_plane;   // out_h x out_w: the blender's dst_mask (static per calibration)
    int32_t n_tiles, n_sets;
    int32_t out_w, out_h, crop_x, crop_y;      // output rectangle inside the canvas
    int32_t canvas_w, canvas_h, tiles_x;
    const float* feather_ws;     // canvas-sized weight sum (general path)
    // lens-undistort pre-pass through the same kernel (remap with BORDER_CONSTANT): the staged window may reach outside
    // the pad_w x pad_h frame, where it is filled with zeros; 0 = frames are only read inside (the composites)
    int32_t pad_w, pad_h;
    uint8_t* pano[kMaxBatch]; uint32_t pano_stride;
    uint8_t* pano_mask[kMaxBatch]; uint32_t mask_stride;
};

struct CanvasDesc {
    int32_t n_cams;
    int32_t nlev;
    int32_t cw[kMaxLevels], ch[kMaxLevels];   // padded canvas pyramid level sizes
    int16_t* C[kMaxLevels];        // debug copy of the collapsed canvas levels, planar 3 x ch x cw (optional)
    const float* WS[kMaxLevels];   // dst_band_weights_ (kept for introspection; kernels re-sum the weights)
    int32_t final_w, final_h;      // dst_roi_final_ (un-padded)
};

// Lens-undistort pre-pass: per camera the fixed-point maps of initUndistortRectifyMap(CV_16SC2).
// map1 == nullptr: the camera's frames are used as they come.
struct UndistortCam {
    const int16_t* map1;           // src_h x src_w x 2: integer source coordinate (x, y)
    const uint16_t* map2;          // src_h x src_w: (fy << 5) | fx
    int32_t w, h;                  // frame size (the maps are frame-sized: the reference undistorts in place size)
};

// Arguments of the canvas kernels (k_blend_coarse: levels 1 .. top -> collapsed level-1 canvas; k_blend_fine: level 0
// -> panorama), passed by value.
struct BlendArgs {
    int32_t nlev, n_cams;
    int32_t cw[kMaxLevels], ch[kMaxLevels];   // padded canvas pyramid level sizes
    int32_t final_w, final_h;                 // dst_roi_final_
    uint8_t* pano[kMaxBatch]; uint32_t pano_stride;          // one panorama (and optional mask) per frame set
    uint8_t* pano_mask[kMaxBatch]; uint32_t mask_stride;
    uint64_t* c1[kMaxBatch];                   // collapsed level-1 canvas of each frame set: ch[1] x cw[1], 8 bytes a pixel
    // Round 4: with three and more levels k_blend_coarse stops at level 2 and hands its tile's interior of the collapsed
    // level-2 canvas (ch[2] x cw[2], 8 bytes a pixel) to k_blend_mid, which forms level 1 the way k_blend_fine forms level 0
    uint64_t* c2[kMaxBatch];
    int32_t split;                             // 1: k_blend_coarse -> c2, k_blend_mid -> c1; 0: k_blend_coarse -> c1 (two levels)
    const uint32_t* coarse_order2;             // k_blend_coarse's own launch order when split (a strip engine also needs the
    int32_t n_coarse2;                         // tiles whose level-2 interiors its level-1 tiles read as pyrUp halo)
    // Canvas strips: the level-1 columns [clip_x0, clip_x1) a strip engine needs (both even; clip_x1 == 0: no clip).  The
    // level-1 regions of k_blend_coarse's tiles are cut to them, so the tile beside the strip costs its two halo columns
    // and their dependency cone, not a whole tile.
    int32_t clip_x0, clip_x1;
    // k_blend_fine, per panorama tile: bit i = camera i has a non-zero level-0 weight under the tile; bit 31 = the
    // tile's level 0 is owned by ONE camera with weight exactly 1 (integer band path) -- then only that camera's bit;
    // bit 30 = one or two cameras with 0/1 weights, never both non-zero on a pixel (integer path with masks)
    const uint32_t* block_cams;
    const uint32_t* block_order;              // tile index (ty * tiles_x + tx) handled by block i
    int32_t tiles_x;
    // k_blend_coarse, per level-1 tile: camera bits over the levels 1 .. top (bit 31: every level owned by the one
    // camera); coarse_pure: bit l set = level l of the tile is owned by ONE camera, that camera's index << 16
    const uint32_t* coarse_cams;
    const uint32_t* coarse_pure;
    const uint32_t* coarse_order;
    int32_t coarse_tiles_x;
    int32_t cc_th;                            // height of a k_blend_coarse tile in level-1 rows (16 or 32) and
    int32_t fine_threads;                     // k_blend_fine block size (256 or 128): chosen per launch by the engine
    int32_t n_tiles, n_coarse_tiles, n_sets;  // tiles and frame sets of this launch (filled in by launch_blend)
    int32_t lds_off[kMaxLevels];              // k_blend_coarse: LDS offset (in pixels) of each level's region buffer
    int16_t* dbg_C[kMaxLevels];               // optional dump of the collapsed levels (MST_FLAG_DEBUG)
    float alpha;                              // (float)(1./255.)
    int32_t weights_16s;                      // MultiBandBlender weight_type CV_16S: integer weights, >> 8 accumulate
};

// One rectangle of a camera's hand-off payload (camera-sharded multi-GPU mode): the interior of a launched tile at
// one pyramid level, `off` dwords into the camera's packed payload (levels above 0: a dword a pixel; level 0: packed BGR8,
// 3 w h bytes rounded up to a dword).  Static per calibration.
struct PayloadRect {
    int32_t level, x0, y0, w, h;
    uint32_t off;
};

// Exposure-gain estimation (GainCompensator::feed): images and masks on the device, and one overlapping pair of
// (block) images: its overlap rectangle's origin inside each parent image, and its size.
struct FeedImage { const uint8_t* img; const uint8_t* mask; uint32_t istride, mstride; };
struct FeedPair { int32_t a, b, ax, ay, bx, by, w, h, i, j; };
struct FeedStat { double sum_a, sum_b; int32_t count; int32_t pad; };

// Work lists (static per calibration): only tiles whose output can reach the panorama are launched.
struct TileList {
    const uint32_t* tiles;         // cam << 24 | ty << 12 | tx
    int32_t count;
    int32_t first;                 // index of tiles[0] in the calibration's full list (per-tile tables)
    // optional: entry li of the launch is tile index[li] of the FULL list (then `tiles` is the full list and `first` is 0)
    const uint32_t* index;
};

// The source window of one warp-kernel tile as it is staged in LDS (k_warp_pyr_lds): columns [wx0, wx0 + 4 * groups),
// rows [by, by + rows) of the camera's frame; the tile's table entries are relative to (bx, by).
struct WarpWin {
    int32_t wx0, by, groups, rows;
    int32_t pitch;                 // LDS row pitch in groups (>= groups; padded so that the taps of a slanted canvas row stay on distinct banks)
    uint32_t flags;                // bit 0: the tile cannot take the staged form (k_warp_pyr handles it)
    // filled in for the launch-ordered copies the staged kernel reads (one scalar load gives the block everything):
    uint32_t tile;                 // cam << 24 | ty << 12 | tx (| bit 31), the tile-list word
    int32_t index;                 // the tile's index in the calibration's full list (table / gain rows)
};

}  // namespace mst
