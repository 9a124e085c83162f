// kernels.hpp -- launchers of the hand-written HIP kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include "device_types.hpp"

namespace mst {

// ---- stage-level (RotationWarper / ExposureCompensator mirrors) ----
void launch_build_maps(const CamDesc* d_cam, float* d_xmap, float* d_ymap, int w, int h, hipStream_t s);
void launch_warp_u8(const CamDesc* d_cam, const uint8_t* d_src, size_t src_stride, int channels,
                    int interp, int border, uint8_t* d_dst, size_t dst_stride, int w, int h, hipStream_t s);
void launch_exposure_apply(const CamDesc* d_cam, uint8_t* d_img, size_t stride, int w, int h, hipStream_t s);
// lens-undistort pre-pass: dst.f[set].ptr[cam] <- remap(src.f[set].ptr[cam]) for every camera that has maps
// NV12 frames ((h * 3 / 2) rows of stride bytes) -> BGR frames, every camera of every frame set in one launch; sizes[cam] = (w, h)
void launch_nv12_to_bgr(const FrameBatch& src, const FrameBatch& dst, const int2* d_sizes, int n_cams, int max_w, int max_h, int batch, hipStream_t s);
void launch_remap_fixed(const UndistortCam* d_ucams, int n_cams, int max_w, int max_h, const FrameBatch& src,
                        const FrameBatch& dst, int batch, hipStream_t s);

// ---- calibration-time static products ----
// weights_16s: MultiBandBlender weight_type CV_16S -- integer weights (0, 2..256) and integer pyrDown, stored as floats
void launch_weight_level0(const CamDesc* d_cams, int cam, float* d_w0, int bw, int bh, bool weights_16s, hipStream_t s);
void launch_border_mask(const CamDesc* d_cams, int cam, uint8_t* d_out, int bw, int bh, hipStream_t s);
void launch_pyrdown_f32(const float* d_src, int w, int h, float* d_dst, bool weights_16s, hipStream_t s);
// mask fix-up of the compose loop (stitcher.cpp:291-294) for one camera; axis tables from build_resize8u_axis
void launch_mask_fixup(const CamDesc* d_cam, const uint8_t* d_seam, int sw, int sh, const int32_t* xo0, const int32_t* xo1,
                       const int16_t* xa0, const int16_t* xa1, const int32_t* yo0, const int32_t* yo1, const int16_t* yb0,
                       const int16_t* yb1, uint8_t* d_out, int w, int h, hipStream_t s);
// FeatherBlender's createWeightMap(mask, sharpness) (blenders.cpp:427-432); d_tmp: w * h ints of scratch
void launch_feather_weight_map(const uint8_t* d_mask, int w, int h, float sharpness, int32_t* d_tmp, float* d_weight,
                               hipStream_t s);
// GainCompensator::feed overlap statistics, one thread per overlapping pair (exposure_feed.cpp)
void launch_overlap_stats(const FeedImage* d_imgs, const FeedPair* d_pairs, int n_pairs, FeedStat* d_out, hipStream_t s);
// camera-sharded hand-off: pack (pyramid -> buffer) or unpack (buffer -> pyramid) one camera's payload rectangles
void launch_payload_copy(const CamDesc* d_cam, const PayloadRect* d_rects, int n_rects, uint32_t* d_buf, bool pack, hipStream_t s);
// VoronoiSeamFinder::findInPair for one overlapping pair (masks tight on the device, updated in place); scratch:
// (rw + 20) * (rh + 20) bytes / ints / floats in d_inv1, d_inv2 / d_tmp / d_dist1, d_dist2
void launch_voronoi_pair(uint8_t* d_m1, int w1, int h1, int ox1, int oy1, uint8_t* d_m2, int w2, int h2, int ox2, int oy2,
                         int rw, int rh, uint8_t* d_inv1, uint8_t* d_inv2, int32_t* d_tmp, float* d_dist1, float* d_dist2,
                         hipStream_t s);
void launch_weight_class(const float* d_w, uint8_t* d_out, size_t n, float unit, hipStream_t s);   // see k_weight_class
void launch_band_weight_sum(const CamDesc* d_cams, int n_cams, int level, float* d_ws, int cw, int ch,
                            hipStream_t s);
void launch_feather_weight_sum(const CamDesc* d_cams, int n_cams, float* d_ws, int cw, int ch, hipStream_t s);

// ---- per-frame hot path: multi-band (kernels_mb.hip) ----
void mb_tile_dims(int* a_tw0, int* a_th0, int* b_t, int* c_tw, int* c_th);
int  mb_read_stamps(int which, unsigned long long* out, size_t n);                // diagnostic builds (-DMST_STAMPS) only
uint32_t mb_build_ablations();      // kernels_mb.hip / kernels_compose.hip: bit mask of the -DMST_ABLATE_* / -DMST_STAMPS switches compiled in
uint32_t compose_build_ablations();
int  mb_read_stamps(int which, unsigned long long* out, size_t n);                // diagnostic builds (-DMST_STAMPS) only
int  mb_chain_depth();                                                           // levels per k_pyr_chain launch
int mb_coarse_tile_width();                                                        // k_blend_coarse tiles, in level-1 pixels
// Launch variants of the canvas kernels: 0 for one or two frame sets per launch sequence, 1 for batches
int mb_launch_variant(int frame_sets);
int mb_coarse_tile_height(int variant);                                            // (BlendArgs::cc_th: 16 or 32)
int mb_fine_threads(int variant);                                                  // (BlendArgs::fine_threads: 256 or 128)
// region of tile (tx, ty) -- a panorama tile (base 0) or a level-1 tile (base 1) -- at `level` >= base: x0, y0, x1, y1
void mb_blend_region(const BlendArgs& a, int base, int tx, int ty, int level, int out[4]);
int  mb_blend_region_lds_pixels(const BlendArgs& a, int base, int tx, int ty, int level);   // LDS footprint of that region
size_t mb_warp_table_entries(int num_tiles);                                         // 32-bit entries for `num_tiles` tiles
size_t mb_warp_blob_entries(int num_tiles);                                          // 32-bit words of per-tile constants
// per calibration: the backward-map table and the per-tile constants of every launched tile of kernel A
void launch_warp_table(const CamDesc* d_cams, int kind, const TileList& list, uint32_t* d_table, uint32_t* d_blobs,
                       hipStream_t s);
// `batch` frame sets per launch: set f uses frames.f[f] and the descriptors d_cams[f * n_cams ...]
// (the per-frame launchers return the name of the kernel they launched -- as rocprofv3 prints it -- or nullptr when the
// list was empty: what the engine's stage timer reports)
const char* launch_warp_pyr(const CamDesc* d_cams, int n_cams, int kind, const TileList& list, const FrameBatch& frames,
                            int batch, const uint32_t* d_table, const uint32_t* d_blobs, hipStream_t s);
// the staged form of the warp kernel (source window through LDS): per-calibration map + gain tables in its lane order and
// the tiles' windows (flags bit 0: the tile stays with k_warp_pyr), then the per-frame launch over a tile list
size_t mb_warp_tile_px();
void launch_warp_table_lds(const CamDesc* d_cams, int kind, const TileList& list, uint32_t* d_table, float* d_gain, WarpWin* d_wins,
                           int lds_budget_px, hipStream_t s);
const char* launch_warp_pyr_lds(const CamDesc* d_cams, int n_cams, int n_tiles, const FrameBatch& frames, int batch,
                                const uint32_t* d_table, const float* d_gain, const WarpWin* d_wins, size_t lds_bytes, hipStream_t s);
const char* launch_pyr_chain(const CamDesc* d_cams, int n_cams, const TileList& list, int base, int k, int batch, hipStream_t s);
// the streaming form of one pyrDown level (level -> level + 1): units of mb_stream_unit_cols() output columns x mb_stream_unit_rows() output
// rows (cam << 24 | uy << 12 | ux), one wave each; source levels of even size only
int  mb_stream_unit_rows();
int  mb_stream_unit_cols();
const char* launch_pyrdown_stream(const CamDesc* d_cams, int n_cams, const uint32_t* d_units, int n_units, int level, int batch, hipStream_t s);
// which: bit 0 = k_blend_coarse (levels 1 .. top -> collapsed level-1 canvas; with a.c2 set: levels 2 .. top -> collapsed
// level-2 canvas), bit 2 = k_blend_mid (only with a.c2 set: level 1), bit 1 = k_blend_fine (level 0 -> panorama).  Returns the
// name of the last kernel launched.
const char* launch_blend(const CamDesc* d_cams, const BlendArgs& a, int num_blocks, int num_coarse, int batch, size_t lds_bytes,
                         hipStream_t s, int which = 7);

// ---- per-frame hot path: no-blend / driver LUT / feather composites (kernels_compose.hip) ----
// calibration: bit i of d_out[canvas tile] = camera i contributes to the tile
void launch_compose_classify(const ComposeArgs& a, int n_canvas_tiles, uint32_t* d_out, hipStream_t s);
void launch_compose_mask(const ComposeArgs& a, uint8_t* d_plane, hipStream_t s);            // the blender's dst_mask
// calibration: per segment the map / gain / weight tables, window origin and flags; sets the tiles' general-path flag
void launch_compose_table(const ComposeArgs& a, int kind, int n_segs, ComposeSeg* d_segs, ComposeTile* d_tiles,
                          const uint32_t* d_seg_tile, uint32_t* d_table, float* d_gain, float* d_weight, hipStream_t s);
// calibration, after the tile classes are known: the table / gain / weight rows of the given segments (those of the staged
// tiles) into the staged kernel's lane order -- entry of tile column c at slot (c & 31) * 4 + (c >> 5)
void launch_compose_permute(const uint32_t* d_seg_ids, int n_segs, uint32_t* d_table, float* d_gain, float* d_weight, hipStream_t s);
// lens-undistort pre-pass in the composites' table form: one tile + segment + 4096 entries per 128 x 32 tile of the frame
// (tile flag kCompTileGeneral: the camera keeps the gathering k_remap_fixed)
void launch_undistort_table(const UndistortCam& u, int tiles_x, int tiles_y, ComposeSeg* d_segs, ComposeTile* d_tiles, uint32_t* d_table,
                            int lds_budget_px, hipStream_t s);
// per frame: a.n_sets frame sets in one launch (a.pano[f], frames.f[f]) over the tile list a.tiles.  general: the
// per-pixel path; else lds_bytes != 0: the LDS-staged form (every segment of the list must fit), 0: gathers from memory
const char* launch_compose(const ComposeArgs& a, const FrameBatch& frames, size_t lds_bytes, bool general, hipStream_t s);

}  // namespace mst
