// device_types.hpp -- plain structs shared by the host engine and the HIP kernels.
#pragma once

#include <cstddef>
#include <cstdint>

namespace mst {

constexpr int kMaxCams = 16;
constexpr int kMaxLevels = 11;   // MST_MAX_BANDS + 1

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
    // ---- exposure (a7)
    int32_t expos;                 // MST_EXPOS_*
    float   gain;                  // GAIN
    const float* gain_map;         // GAIN_BLOCKS: gh x gw
    int32_t gw, gh;
    const int32_t* gx0; const int32_t* gx1; const float* gxw0; const float* gxw1;   // per roi column
    const int32_t* gy0; const int32_t* gy1; const float* gyw0; const float* gyw1;   // per roi row
    // ---- static per-calibration rasters
    const uint8_t* mask;           // final mask_warped, roi_h x roi_w, tight
    const float*   feather_w;      // createWeightMap(mask), roi_h x roi_w (feather only)
    int32_t corner_x, corner_y;    // corners[i] - canvas.tl  (feather / paste)
    // ---- multi-band (a15-a18): bordered image geometry and pyramids
    int32_t top, left;             // copyMakeBorder offsets
    int32_t nlev;                  // num_bands + 1
    int32_t lw[kMaxLevels], lh[kMaxLevels];   // bordered pyramid level sizes
    int32_t ox[kMaxLevels], oy[kMaxLevels];   // window origin inside the canvas level
    int16_t* G[kMaxLevels];        // Gaussian pyramid, planar 3 x lh x lw
    const float* W[kMaxLevels];    // weight pyramid, lh x lw
};

struct FrameArgs {                 // per-frame source pointers, passed by value
    const uint8_t* ptr[kMaxCams];
    size_t stride[kMaxCams];
};

struct CanvasDesc {
    int32_t n_cams;
    int32_t nlev;
    int32_t cw[kMaxLevels], ch[kMaxLevels];   // padded canvas pyramid level sizes
    int16_t* C[kMaxLevels];        // collapsed canvas level, planar 3 x ch x cw
    const float* WS[kMaxLevels];   // dst_band_weights_
    int32_t final_w, final_h;      // dst_roi_final_ (un-padded)
};

struct SegTable {                  // block index -> (camera, local block) for batched launches
    int32_t start[kMaxCams + 1];
    int32_t tiles_x[kMaxCams];
};

}  // namespace mst
