// kernels_compose.hip -- the single-pass composites as tiled, table-driven HIP kernels for gfx950:
//
//   Blender::NO   prepare / feed / blend                       R64/src/blenders.cpp:65-112
//   driver LUT    prepareSize / feedSize / feedSizeRemap       R64/stitching_detailed.cpp:94-177 (crop: DRV:153,169,702)
//   FeatherBlender prepare / feed / blend                      R64/src/blenders.cpp:115-155
//
// each fused with the per-frame stages in front of it (RotationWarper::warp = cv::remap INTER_LINEAR +
// BORDER_REFLECT, detail/warpers_inl.hpp:88-99; ExposureCompensator::apply, exposure_compensate.cpp:150-153,
// 225-246; convertTo(16S), stitcher.cpp:285) and the final convertTo(8U) (stitcher.cpp:313): source frames in,
// panorama out, nothing in between touches memory.
//
// What the reference recomputes per frame and what is frame-invariant are separated the way its own driver does it
// (the LUT of DRV:94-148 is exactly a per-calibration "which camera, which source pixel" table): mst_update cuts the
// output into 128 x 32 tiles and gives every tile one SEGMENT per camera that reaches it.  A segment holds, per
// pixel, the backward map in cv::remap's fixed-point form (window offset from the segment's window origin and the
// two 5-bit fractions: one dword), the exposure gain (one float; BlocksGainCompensator's resize() evaluated once
// instead of once per frame) and for the feather blend the camera's weight.  Per frame a lane then handles four
// adjacent panorama pixels of a row: one 16-byte table load and one 16-byte gain load, eight aligned 12-byte
// gathers, the separable v_dot4 bilinear form, and one 12-byte panorama store; the frame sets of a batch that share a
// tile run back to back on one XCD, so the tables come out of that XCD's L2 for all but the first.
//
// Measured on MI355X (tools/ubench/sampler.hip): this loop is bound by the memory pipeline (gather + streaming
// loads + stores through one in-order queue per wave), not by instruction issue -- halving the VALU count changed
// nothing, four pixels per lane (a quarter of the streaming instructions) and keeping stores out from between the
// gathers of a row did.  Unaligned 8-byte gathers (no funnel shift) are slower than aligned 12-byte ones.
//
// Tiles whose pixels cannot all take that form -- a window on the frame's last pixels that the 12-byte reads would
// overrun, a window further than 1023 x 511 pixels from the segment origin, a gain of 8e6 and more -- run the
// general per-pixel path (the round-1 kernels' arithmetic, kept below as device functions) in a launch of their own.
#include <algorithm>
#include <climits>
#include <cstdlib>

#include "kernels.hpp"
#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace mst {

using namespace dev;

namespace {

typedef uint32_t __attribute__((ext_vector_type(3))) u32x3;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
typedef float __attribute__((ext_vector_type(4))) f32x4;
typedef uint32_t __attribute__((aligned(1))) u32_any;      // a dword at any byte address (gfx950 serves unaligned accesses)
struct __attribute__((packed, aligned(1))) Bytes12 { uint32_t x, y, z; };

__device__ __forceinline__ int xcd_run_index(int b, int n)      // see kernels_mb.hip
{
    const int q = n >> 3, r = n & 7, k = b & 7, j = b >> 3;
    return k * q + min(k, r) + j;
}

// ------------------------------------------------------------------------------------------------------
// per-pixel semantics (shared by the table builders and the general path)
// ------------------------------------------------------------------------------------------------------

// Blender::feed: later cameras overwrite earlier ones where their mask is set; dst_mask |= mask
__device__ __forceinline__ int paste_owner(const CamDesc* __restrict__ cams, int n_cams, int x, int y, int& olx, int& oly,
                                           int& mask_or)
{
    int owner = -1;
    mask_or = 0;
    olx = oly = 0;
    for (int i = 0; i < n_cams; ++i) {
        const CamDesc& c = cams[i];
        const int lx = x - c.corner_x, ly = y - c.corner_y;
        if (static_cast<unsigned>(lx) >= static_cast<unsigned>(c.roi_w) ||
            static_cast<unsigned>(ly) >= static_cast<unsigned>(c.roi_h))
            continue;
        const int mv = gp(c.mask)[static_cast<size_t>(ly) * c.roi_w + lx];
        mask_or |= mv;
        if (mv) { owner = i; olx = lx; oly = ly; }
    }
    return owner;
}

// one panorama pixel of the no-blend / LUT composite, the general way (canvas coordinates)
__device__ __forceinline__ void paste_pixel(const ComposeArgs& a, const FrameArgs& frames, int x, int y, int v[3], int& mask_or)
{
    int olx, oly;
    int owner = paste_owner(a.cams, a.n_cams, x, y, olx, oly, mask_or);
    v[0] = v[1] = v[2] = 0;
    if (owner < 0 && a.mode == kCompModeLut) { owner = 0; olx = 0; oly = 0; }   // zero-initialised LUT entry (DRV:107-112)
    if (owner >= 0) warped_pixel(a.cams[owner], frames.ptr[owner], frames.stride[owner], olx, oly, v);
}

// one panorama pixel of the feather blend, the general way
__device__ __forceinline__ void feather_pixel(const ComposeArgs& a, const FrameArgs& frames, int x, int y, int v[3], bool& m)
{
    int acc[3] = {0, 0, 0};
    for (int i = 0; i < a.n_cams; ++i) {
        const CamDesc& c = a.cams[i];
        const int lx = x - c.corner_x, ly = y - c.corner_y;
        if (static_cast<unsigned>(lx) >= static_cast<unsigned>(c.roi_w) ||
            static_cast<unsigned>(ly) >= static_cast<unsigned>(c.roi_h))
            continue;
        const float w = gp(c.feather_w)[static_cast<size_t>(ly) * c.roi_w + lx];
        if (w == 0.f) continue;
        int s[3];
        warped_pixel(c, frames.ptr[i], frames.stride[i], lx, ly, s);
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[p] = wrap_s16(acc[p] + trunc_s16(static_cast<float>(s[p]) * w));
    }
    const float wsum = gp(a.feather_ws)[static_cast<size_t>(y) * a.canvas_w + x];
    m = wsum > kWeightEps;
    const float ws = wsum + kWeightEps;
#pragma unroll
    for (int p = 0; p < 3; ++p) v[p] = m ? sat_u8(trunc_s16(static_cast<float>(acc[p]) / ws)) : 0;
}

// ------------------------------------------------------------------------------------------------------
// calibration: which cameras reach a tile, the segment tables, the static mask plane
// ------------------------------------------------------------------------------------------------------

// Does camera `cam` contribute to canvas pixel (x, y), and at which roi position?  Paste / LUT: it owns the pixel
// (LUT: pixels nobody owns read camera 0 at roi (0, 0)); feather: the pixel lies in its roi with a non-zero weight.
__device__ __forceinline__ bool contributes(const ComposeArgs& a, int cam, int x, int y, int& lx, int& ly, bool fill = false)
{
    if (a.mode == kCompModeFeather) {
        const CamDesc& c = a.cams[cam];
        lx = x - c.corner_x; ly = y - c.corner_y;
        if (static_cast<unsigned>(lx) >= static_cast<unsigned>(c.roi_w) ||
            static_cast<unsigned>(ly) >= static_cast<unsigned>(c.roi_h))
            return false;
        return gp(c.feather_w)[static_cast<size_t>(ly) * c.roi_w + lx] != 0.f;
    }
    int mask_or;
    const int owner = paste_owner(a.cams, a.n_cams, x, y, lx, ly, mask_or);
    // LUT: the pixels nobody owns read camera 0 at roi (0, 0) (the zero-initialised LUT entry, DRV:107-112).  They get a
    // segment of their own (`fill`): one source window, usually far from the windows of camera 0's own pixels.
    if (fill) { lx = 0; ly = 0; return owner < 0; }
    return owner == cam;
}

// bit i of out[tile] = camera i contributes to some pixel of the tile (tiles in canvas order); bit 16: LUT fill pixels
__global__ __launch_bounds__(256) void k_compose_classify(ComposeArgs a, uint32_t* __restrict__ out)
{
    __shared__ uint32_t s_bits;
    if (threadIdx.x == 0) s_bits = 0;
    __syncthreads();
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    uint32_t bits = 0;
    for (int i = threadIdx.x; i < kCompPx; i += 256) {
        const int ox = tx * kCompTW + (i % kCompTW), oy = ty * kCompTH + (i / kCompTW);
        if (ox >= a.out_w || oy >= a.out_h) continue;
        const int x = ox + a.crop_x, y = oy + a.crop_y;
        if (a.mode == kCompModeFeather) {
            for (int c = 0; c < a.n_cams; ++c) {
                int lx, ly;
                if (contributes(a, c, x, y, lx, ly)) bits |= 1u << c;
            }
        } else {
            int lx, ly, mask_or;
            const int owner = paste_owner(a.cams, a.n_cams, x, y, lx, ly, mask_or);
            if (owner >= 0) bits |= 1u << owner;
            else if (a.mode == kCompModeLut) bits |= 1u << 16;          // the fill segment (camera 0 at roi (0, 0))
        }
    }
    if (bits) atomicOr(&s_bits, bits);
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s_bits;
}

// the blender's dst_mask, static per calibration: Blender::feed ORs the camera masks (blenders.cpp:98-100); the
// feather blend's is `dst_weight_map_ > WEIGHT_EPS` (blenders.cpp:153)
__global__ __launch_bounds__(256) void k_compose_mask(ComposeArgs a, uint8_t* __restrict__ plane)
{
    const int ox = blockIdx.x * 64 + (threadIdx.x & 63), oy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ox >= a.out_w || oy >= a.out_h) return;
    const int x = ox + a.crop_x, y = oy + a.crop_y;
    int m;
    if (a.mode == kCompModeFeather) {
        m = gp(a.feather_ws)[static_cast<size_t>(y) * a.canvas_w + x] > kWeightEps ? 255 : 0;
    } else {
        int lx, ly;
        (void)paste_owner(a.cams, a.n_cams, x, y, lx, ly, m);
    }
    plane[static_cast<size_t>(oy) * a.out_w + ox] = static_cast<uint8_t>(m);
}

// One block per segment: the table, gain and weight entries of its 128 x 32 pixels, the segment's window origin and
// flags, and the owning tile's "general path" flag.  seg_tile[s]: index of the segment's tile in `tiles`.
constexpr uint32_t kEntryValid = 1u << 31;
template <int KIND>
__global__ __launch_bounds__(256) void k_compose_table(ComposeArgs a, ComposeSeg* __restrict__ segs, ComposeTile* __restrict__ tiles,
                                                       const uint32_t* __restrict__ seg_tile, uint32_t* __restrict__ table,
                                                       float* __restrict__ gain, float* __restrict__ weight)
{
    __shared__ int s_min[2], s_max[2];
    __shared__ uint32_t s_flags[2];      // [0]: bit 0 general pixels seen, bit 1 windows only LDS can read; [1]: a contributing weight != 1
    const int s = blockIdx.x;
    const ComposeTile t = tiles[seg_tile[s]];
    const int cam = segs[s].cam;
    const bool fill = (segs[s].flags & kCompSegFill) != 0;
    const CamDesc& c = a.cams[cam];
    const int tx = t.tx_ty & 0xFFFF, ty = t.tx_ty >> 16;
    if (threadIdx.x == 0) { s_min[0] = INT_MAX; s_min[1] = INT_MAX; s_max[0] = INT_MIN; s_max[1] = INT_MIN; s_flags[0] = 0; s_flags[1] = 0; }
    __syncthreads();
    float kr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) kr[i] = c.kr[i];
    constexpr int kPer = kCompPx / 256;
    // pass 1: the maps, their smallest window coordinate
    int x_lo = INT_MAX, y_lo = INT_MAX, x_hi = INT_MIN, y_hi = INT_MIN;
    uint32_t valid_bits = 0;
    for (int j = 0; j < kPer; ++j) {
        const int i = threadIdx.x + 256 * j;
        const int ox = tx * kCompTW + (i % kCompTW), oy = ty * kCompTH + (i / kCompTW);
        if (ox >= a.out_w || oy >= a.out_h) continue;
        int lx, ly;
        if (!contributes(a, cam, ox + a.crop_x, oy + a.crop_y, lx, ly, fill)) continue;
        valid_bits |= 1u << j;
        const MapPx m = map_px<KIND>(c, kr, c.src_w, c.src_h, load_global(c.cols + lx), load_global(c.rows + ly), lx, ly);
        if (m.inside) { x_lo = min(x_lo, m.ix); y_lo = min(y_lo, m.iy); x_hi = max(x_hi, m.ix + 1); y_hi = max(y_hi, m.iy + 1); }
    }
    if (x_lo != INT_MAX) { atomicMin(&s_min[0], x_lo); atomicMin(&s_min[1], y_lo); atomicMax(&s_max[0], x_hi); atomicMax(&s_max[1], y_hi); }
    __syncthreads();
    const int bx = s_min[0] == INT_MAX ? 0 : s_min[0], by = s_min[1] == INT_MAX ? 0 : s_min[1];
    // pass 2: pack
    bool general = false, not_unit = false, no_gather = false;
    for (int j = 0; j < kPer; ++j) {
        const int i = threadIdx.x + 256 * j;
        uint32_t e = 0;
        float g = 1.f, w = 0.f;
        if (valid_bits >> j & 1u) {
            const int ox = tx * kCompTW + (i % kCompTW), oy = ty * kCompTH + (i / kCompTW);
            int lx, ly;
            (void)contributes(a, cam, ox + a.crop_x, oy + a.crop_y, lx, ly, fill);
            const MapPx m = map_px<KIND>(c, kr, c.src_w, c.src_h, load_global(c.cols + lx), load_global(c.rows + ly), lx, ly);
            // (a window that folds differently for the two forms cannot happen: `fast` implies the same folded window)
            const uint32_t dx = static_cast<uint32_t>(m.ix - bx), dy = static_cast<uint32_t>(m.iy - by);
            g = c.expos == 0 ? 1.f : gain_at(c, lx, ly);
            if (a.mode == kCompModeFeather) {
                w = gp(c.feather_w)[static_cast<size_t>(ly) * c.roi_w + lx];
                not_unit |= w != 1.0f;
            }
            if (m.inside && dx < 1024u && dy < 512u && !(g >= 8.0e6f)) {
                e = static_cast<uint32_t>(m.ifx) | static_cast<uint32_t>(m.ify) << 6 | dx << 12 | dy << 22 | kEntryValid;
                no_gather |= !m.fast;
            } else {
                general = true;
            }
        }
        const size_t o = static_cast<size_t>(s) * kCompPx + i;
        table[o] = e;
        if (gain) gain[o] = g;
        if (weight) weight[o] = w;
    }
    if (general) atomicOr(&s_flags[0], 1u);
    if (no_gather) atomicOr(&s_flags[0], 2u);
    if (not_unit) atomicOr(&s_flags[1], 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
        segs[s].bx = bx;
        segs[s].by = by;
        segs[s].flags = ((a.mode == kCompModeFeather && !s_flags[1]) ? kCompSegUnit : 0u) | (fill ? kCompSegFill : 0u);
        // the window as staged in LDS: whole groups of four pixels from a multiple of four (their 12 source bytes
        // start on a dword of an aligned frame).  Too large for the budget, or a last group that would read past
        // the end of the frame's last row: the tile gathers from memory instead.
        const int wx0 = bx & ~3;
        const int groups = s_max[0] == INT_MIN ? 0 : (s_max[0] - wx0 + 4) / 4, rows = s_max[1] == INT_MIN ? 0 : s_max[1] - by + 1;
        segs[s].wx0 = wx0; segs[s].groups = groups; segs[s].rows = rows; segs[s].pad = 0;
        const bool overrun = wx0 + 4 * groups > c.src_w && by + rows >= c.src_h;
        uint32_t tf = ((s_flags[0] & 1u) ? kCompTileGeneral : 0u) | ((s_flags[0] & 2u) ? kCompTileNoGather : 0u);
        if (groups * 4 * rows > kCompLdsPx || groups > 256 || overrun || groups == 0) tf |= kCompTileDirect;
        if (tf) atomicOr(&tiles[seg_tile[s]].flags, tf);
    }
}

// The staged kernel's lanes own the tile columns l, l + 32, l + 64, l + 96 (round 4: the 32 lanes of a tap read then touch
// neighbouring source pixels -- one LDS bank each -- where four adjacent pixels per lane put them at a stride of four
// dwords, eight banks four lanes deep; kernels_mb.hip explains).  A lane still fetches its four entries of a row with one
// 16-byte load: the rows of the segments of every STAGED tile are re-ordered in place, once per calibration, after the
// tile classes are known: entry of column c -> slot (c & 31) * 4 + (c >> 5).  Tiles of the other classes keep the natural
// order (k_compose reads four adjacent pixels per lane).
__global__ __launch_bounds__(256) void k_compose_permute(const uint32_t* __restrict__ seg_ids, uint32_t* __restrict__ table,
                                                         float* __restrict__ gain, float* __restrict__ weight)
{
    const size_t base = static_cast<size_t>(seg_ids[blockIdx.x]) * kCompPx;
    constexpr int kPer = kCompPx / 256;
    auto permute = [&](uint32_t* __restrict__ t) {
        uint32_t v[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) v[j] = t[base + threadIdx.x + 256 * j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int i = threadIdx.x + 256 * j, row = i / kCompTW, c = i % kCompTW;
            t[base + row * kCompTW + (c & 31) * 4 + (c >> 5)] = v[j];
        }
        __syncthreads();
    };
    permute(table);
    if (gain) permute(reinterpret_cast<uint32_t*>(gain));
    if (weight) permute(reinterpret_cast<uint32_t*>(weight));
}

// ---- lens-undistort pre-pass (DRV:232,741) through the staged kernel ------------------------------------------------
// cv::remap(frame, undistorted, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR, BORDER_CONSTANT 0): the integer tap and the
// 5-bit fractions come ready-made from initUndistortRectifyMap.  Per calibration they are recast into the composites'
// table form (one segment per 128 x 32 tile of the frame, the frame being its own "panorama"); taps outside the frame
// are zeros, which the staged window provides by itself (stage_window<PAD>), so a window only has to start at -1.
// A pixel all of whose taps lie outside is simply 0 (entry invalid).
__global__ __launch_bounds__(256) void k_undistort_table(UndistortCam u, int tiles_x, ComposeSeg* __restrict__ segs, ComposeTile* __restrict__ tiles,
                                                         uint32_t* __restrict__ table, int lds_budget_px)
{
    __shared__ int s_min[2], s_max[2];
    const int s = blockIdx.x, tx = s % tiles_x, ty = s / tiles_x;
    if (threadIdx.x == 0) { s_min[0] = INT_MAX; s_min[1] = INT_MAX; s_max[0] = INT_MIN; s_max[1] = INT_MIN; }
    __syncthreads();
    constexpr int kPer = kCompPx / 256;
    int x_lo = INT_MAX, y_lo = INT_MAX, x_hi = INT_MIN, y_hi = INT_MIN;
    uint32_t valid_bits = 0;
    for (int j = 0; j < kPer; ++j) {
        const int i = threadIdx.x + 256 * j;
        const int ox = tx * kCompTW + (i % kCompTW), oy = ty * kCompTH + (i / kCompTW);
        if (ox >= u.w || oy >= u.h) continue;
        const size_t mi = static_cast<size_t>(oy) * u.w + ox;
        const int sx = gp(u.map1)[2 * mi], sy = gp(u.map1)[2 * mi + 1];
        if (sx < -1 || sx >= u.w || sy < -1 || sy >= u.h) continue;       // every tap outside: 0
        valid_bits |= 1u << j;
        x_lo = min(x_lo, sx); y_lo = min(y_lo, sy); x_hi = max(x_hi, sx + 1); y_hi = max(y_hi, sy + 1);
    }
    if (x_lo != INT_MAX) { atomicMin(&s_min[0], x_lo); atomicMin(&s_min[1], y_lo); atomicMax(&s_max[0], x_hi); atomicMax(&s_max[1], y_hi); }
    __syncthreads();
    const int bx = s_min[0] == INT_MAX ? 0 : s_min[0], by = s_min[1] == INT_MAX ? 0 : s_min[1];
    bool far = false;
    for (int j = 0; j < kPer; ++j) {
        const int i = threadIdx.x + 256 * j;
        uint32_t e = 0;
        if (valid_bits >> j & 1u) {
            const int ox = tx * kCompTW + (i % kCompTW), oy = ty * kCompTH + (i / kCompTW);
            const size_t mi = static_cast<size_t>(oy) * u.w + ox;
            const int sx = gp(u.map1)[2 * mi], sy = gp(u.map1)[2 * mi + 1];
            const uint32_t fr = gp(u.map2)[mi];
            const uint32_t dx = static_cast<uint32_t>(sx - bx), dy = static_cast<uint32_t>(sy - by);
            if (dx < 1024u && dy < 512u) e = (fr & 31u) | ((fr >> 5) & 31u) << 6 | dx << 12 | dy << 22 | kEntryValid;
            else far = true;
        }
        table[static_cast<size_t>(s) * kCompPx + i] = e;
    }
    const int any_far = __syncthreads_or(far);
    if (threadIdx.x == 0) {
        ComposeSeg sg{};
        sg.cam = 0; sg.bx = bx; sg.by = by; sg.flags = 0;
        // floor to a multiple of four that may be negative (column -1 lives in the all-zero group [-4, -1])
        sg.wx0 = bx >= 0 ? (bx & ~3) : -4;
        sg.groups = s_max[0] == INT_MIN ? 0 : (s_max[0] - sg.wx0 + 4) / 4;
        sg.rows = s_max[1] == INT_MIN ? 0 : s_max[1] - by + 1;
        segs[s] = sg;
        ComposeTile t{};
        t.seg_first = static_cast<uint32_t>(s);
        t.seg_count = sg.groups ? 1u : 0u;
        t.flags = (any_far || sg.groups * 4 * sg.rows > lds_budget_px || sg.groups > 256) ? kCompTileGeneral : 0u;
        t.tx_ty = static_cast<uint32_t>(tx) | static_cast<uint32_t>(ty) << 16;
        tiles[s] = t;
    }
}

// ------------------------------------------------------------------------------------------------------
// per frame
// ------------------------------------------------------------------------------------------------------

// cv::remap's fixed-point bilinear sum for one BGR pixel from the aligned dwords covering the two 6-byte pixel pairs,
// then the exposure gain: (sum_y wy * (sum_x wx * v) + 512) >> 10 == (sum v * wx * wy * 32 + 2^14) >> 15 in exact
// integers (bilinear_window_c3, kernels_common.hpp).  The vertical weights come in multiplied by 64, so that the
// rounded byte lands in bits 16..23 and v_cvt_f32_ubyte2 picks it up without a shift; saturate_cast<uchar>(v * gain)
// for gains below 8e6 is what v_cvt_pk_u8_f32 computes (kernels_mb.hip, blend_pixel).
__device__ __forceinline__ uint32_t sample_window(u32x3 p0, u32x3 p1, uint32_t o0, uint32_t o1, uint32_t wx,
                                                  uint32_t wy0s, uint32_t wy1s, float gain)
{
    const uint32_t lo0 = __builtin_amdgcn_alignbyte(p0.y, p0.x, o0), hi0 = __builtin_amdgcn_alignbyte(p0.z, p0.y, o0);
    const uint32_t lo1 = __builtin_amdgcn_alignbyte(p1.y, p1.x, o1), hi1 = __builtin_amdgcn_alignbyte(p1.z, p1.y, o1);
    const uint32_t g0 = __builtin_amdgcn_alignbyte(hi0, lo0, 1), r0 = __builtin_amdgcn_alignbyte(hi0, lo0, 2);
    const uint32_t g1 = __builtin_amdgcn_alignbyte(hi1, lo1, 1), r1 = __builtin_amdgcn_alignbyte(hi1, lo1, 2);
    const uint32_t hb0 = __builtin_amdgcn_udot4(lo0, wx, 0u, false), hb1 = __builtin_amdgcn_udot4(lo1, wx, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(g0, wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1, wx, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(r0, wx, 0u, false), hr1 = __builtin_amdgcn_udot4(r1, wx, 0u, false);
    uint32_t b = __umul24(hb1, wy1s) + (__umul24(hb0, wy0s) + 32768u);
    uint32_t g = __umul24(hg1, wy1s) + (__umul24(hg0, wy0s) + 32768u);
    uint32_t r = __umul24(hr1, wy1s) + (__umul24(hr0, wy0s) + 32768u);
    // (all 32 bits "used": when only bits 16..23 are demanded the compiler rewrites the 24-bit multiplies as plain
    // 32-bit ones and selects the quarter-rate v_mul_lo_u32 for them)
    asm("" : "+v"(b)); asm("" : "+v"(g)); asm("" : "+v"(r));
    uint32_t px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((b >> 16) & 0xFFu) * gain, 0u, 0u);
    px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((g >> 16) & 0xFFu) * gain, 1u, px);
    px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((r >> 16) & 0xFFu) * gain, 2u, px);
    return px;
}

// per byte: b - (b != 0)   (the normalised value of a pixel one camera owns with weight exactly 1:
// short(v / (1.0f + 1e-5f)) == v - 1 for v in 1..255, see band_quad in kernels_mb.hip)
__device__ __forceinline__ uint32_t bytes_toward_zero(uint32_t v)
{
    const uint32_t nz = (((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v) & 0x80808080u;
    return v - (nz >> 7);
}

// Tiles on the general per-pixel path: 16 blocks per (tile, frame set), one pixel per thread -- these are few (frame
// borders, degenerate sources), and a whole tile per block would leave the launch waiting for one long block.
constexpr int kGeneralSplit = kCompPx / 256;
template <int MODE>
__global__ __launch_bounds__(256) void k_compose_general(ComposeArgs a, FrameBatch fb)
{
    const int sub = blockIdx.x % kGeneralSplit, pair = blockIdx.x / kGeneralSplit;
    const int set = pair % a.n_sets;
    const ComposeTile t = a.tiles[pair / a.n_sets];
    const int tx = t.tx_ty & 0xFFFF, ty = t.tx_ty >> 16;
    uint8_t* __restrict__ pano = a.pano[set];
    uint8_t* __restrict__ pano_mask = a.pano_mask[set];
    const int i = sub * 256 + threadIdx.x;
    const int ox = tx * kCompTW + (i % kCompTW), oy = ty * kCompTH + (i / kCompTW);
    if (ox >= a.out_w || oy >= a.out_h) return;
    int v[3], mv;
    if (MODE == kCompModeFeather) {
        bool m;
        feather_pixel(a, fb.f[set], ox + a.crop_x, oy + a.crop_y, v, m);
        mv = m ? 255 : 0;
    } else {
        paste_pixel(a, fb.f[set], ox + a.crop_x, oy + a.crop_y, v, mv);
    }
    uint8_t* o = pano + __umul24(oy, a.pano_stride) + ox * 3;
    o[0] = static_cast<uint8_t>(v[0]); o[1] = static_cast<uint8_t>(v[1]); o[2] = static_cast<uint8_t>(v[2]);
    if (pano_mask) pano_mask[__umul24(oy, a.mask_stride) + ox] = static_cast<uint8_t>(mv);
}


// FeatherBlender's normalisation divides three sums of one pixel by the same weight sum.  IEEE division is a
// reciprocal estimate, one Newton step on it, and two fused corrections of the quotient; the first half depends on
// the divisor only, so it is done once per pixel.  Without the range scaling of the general sequence this is the
// correctly rounded quotient for every divisor in [1e-5, 17] and every integer numerator in [0, 4095] -- the only
// operands the sum of at most 16 cameras' short(pixel * weight) over (weight sum + 1e-5) can present --
// checked exhaustively on gfx950 against the compiler's division (tools/ubench/div_exact.hip: 174 M divisors x 4096).
struct SharedDivisor { float d, r; };
__device__ __forceinline__ SharedDivisor shared_divisor(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    return SharedDivisor{d, __builtin_fmaf(e, r0, r0)};
}
__device__ __forceinline__ float divide_by(float n, const SharedDivisor& s)
{
    const float q0 = n * s.r;
    const float e1 = __builtin_fmaf(-s.d, q0, n);
    const float q1 = __builtin_fmaf(e1, s.r, q0);
    const float e2 = __builtin_fmaf(-s.d, q1, n);
    return __builtin_fmaf(e2, s.r, q1);
}
// the normalised, saturated byte of one channel: quotients lie in [0, 255) (sum of v_i * w_i <= 255 * sum of w_i), so
// short() and the 8-bit saturation of the general form never act
__device__ __forceinline__ uint32_t feather_norm(int acc, const SharedDivisor& s)
{
    return static_cast<uint32_t>(static_cast<int>(divide_by(static_cast<float>(acc), s)));
}

constexpr int kRowsPerLane = 4;          // 32 lanes x 4 pixels = one tile row; 8 row groups x 4 rows = the tile
static_assert(kCompTW == 128 && kCompTH == 8 * kRowsPerLane, "thread layout of k_compose");

// ROWS: rows per lane.  4 = one block per tile; 1 = four blocks per tile (short lists: a quarter of the dependent
// table -> gather -> store round trips per block)
template <int MODE, int ROWS>
__global__ __launch_bounds__(256) void k_compose(ComposeArgs a, FrameBatch fb)
{
    constexpr int kSplit = kRowsPerLane / ROWS;
    const int sub = blockIdx.x % kSplit;
    const int pair = xcd_run_index(blockIdx.x / kSplit, a.n_tiles * a.n_sets);
    const int set = pair % a.n_sets;
    const ComposeTile t = a.tiles[pair / a.n_sets];
    const int tx = t.tx_ty & 0xFFFF, ty = t.tx_ty >> 16;
    uint8_t* __restrict__ pano = a.pano[set];
    uint8_t* __restrict__ pano_mask = a.pano_mask[set];
    const FrameArgs& frames = fb.f[set];
    const int lane = threadIdx.x & 31, rgrp = threadIdx.x >> 5;
    const int ox = tx * kCompTW + lane * 4;
    const bool aligned_rows = ((static_cast<uint32_t>(reinterpret_cast<uintptr_t>(pano)) | a.pano_stride) & 3u) == 0;
    const bool pure = MODE == kCompModeFeather && t.seg_count == 1 && (a.segs[t.seg_first].flags & kCompSegUnit);
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        const int row = (sub * 8 + rgrp) * ROWS + it, oy = ty * kCompTH + row;
        uint32_t px[4] = {0u, 0u, 0u, 0u};
        int acc[4][3];
        f32x4 ws = {0.f, 0.f, 0.f, 0.f};       // dst_weight_map_: the cameras' weights summed in feed order (entries of other cameras hold 0)
        if (MODE == kCompModeFeather) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[k][0] = acc[k][1] = acc[k][2] = 0;
        }
        for (uint32_t s = 0; s < t.seg_count; ++s) {
            const ComposeSeg seg = a.segs[t.seg_first + s];
            const uint32_t stride = frames.stride[seg.cam];
            const uint8_t* p = frames.ptr[seg.cam] + (static_cast<size_t>(seg.by) * stride + static_cast<size_t>(seg.bx) * 3u);
            const uint32_t adj = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p)) & 3u;
            const uint8_t MST_GLOBAL* base = gp(p - adj);
            const size_t idx = static_cast<size_t>(t.seg_first + s) * kCompPx + row * kCompTW + lane * 4;
            const u32x4 e = *(const u32x4 MST_GLOBAL*)(a.table + idx);
            f32x4 g = {1.f, 1.f, 1.f, 1.f};
            if (a.gain) g = *(const f32x4 MST_GLOBAL*)(a.gain + idx);
            f32x4 w = {0.f, 0.f, 0.f, 0.f};
            if (MODE == kCompModeFeather && !pure) { w = *(const f32x4 MST_GLOBAL*)(a.weight + idx); ws += w; }
            u32x3 p0[4], p1[4];
            uint32_t o0[4], o1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // entries that are not this camera's read the segment origin's window and drop the result
                const uint32_t dx = (e[k] >> 12) & 1023u, dy = (e[k] >> 22) & 511u;
                o0[k] = __umul24(dy, stride) + dx * 3u + adj;
                o1[k] = o0[k] + stride;
                p0[k] = *(const u32x3 MST_GLOBAL*)(base + (o0[k] & ~3u));
                p1[k] = *(const u32x3 MST_GLOBAL*)(base + (o1[k] & ~3u));
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t fx = e[k] & 63u;
                const uint32_t wx = __umul24(fx, 0xFFFFFFu) + 32u;            // (32 - fx) | fx << 24
                const uint32_t wy1s = e[k] & 0xFC0u, wy0s = 2048u - wy1s;     // fy * 64, (32 - fy) * 64
                const uint32_t v = sample_window(p0[k], p1[k], o0[k], o1[k], wx, wy0s, wy1s, g[k]);
                const bool valid = (e[k] & kEntryValid) != 0;
                if (MODE != kCompModeFeather || pure) {
                    px[k] = valid ? v : px[k];
                } else if (valid) {
                    // dst += static_cast<short>(src * w): products within [0, 255], sums of at most 16 of them
                    acc[k][0] += static_cast<int>(static_cast<float>(v & 0xFFu) * w[k]);
                    acc[k][1] += static_cast<int>(static_cast<float>((v >> 8) & 0xFFu) * w[k]);
                    acc[k][2] += static_cast<int>(static_cast<float>((v >> 16) & 0xFFu) * w[k]);
                }
            }
        }
        if (MODE == kCompModeFeather) {
            if (pure) {
#pragma unroll
                for (int k = 0; k < 4; ++k) px[k] = bytes_toward_zero(px[k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool m = ws[k] > kWeightEps;
                    const SharedDivisor sd = shared_divisor(ws[k] + kWeightEps);
                    uint32_t o = 0;
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) o |= feather_norm(acc[k][ch], sd) << (8 * ch);
                    px[k] = m ? o : 0u;
                }
            }
        }
        if (oy >= a.out_h || ox >= a.out_w) continue;
        uint8_t* o = pano + __umul24(oy, a.pano_stride) + ox * 3;
        if (ox + 3 < a.out_w) {
            const uint32_t w0 = px[0] | px[1] << 24, w1 = px[1] >> 8 | px[2] << 16, w2 = px[2] >> 16 | px[3] << 8;
            if (aligned_rows) *reinterpret_cast<u32x3 MST_GLOBAL*>(gp(o)) = u32x3{w0, w1, w2};
            else *reinterpret_cast<Bytes12*>(o) = Bytes12{w0, w1, w2};
            if (pano_mask)
                *reinterpret_cast<u32_any*>(pano_mask + __umul24(oy, a.mask_stride) + ox) =
                    *reinterpret_cast<const u32_any*>(a.mask_plane + static_cast<size_t>(oy) * a.out_w + ox);
        } else {
            for (int k = 0; k < 4 && ox + k < a.out_w; ++k) {
                o[3 * k] = static_cast<uint8_t>(px[k]); o[3 * k + 1] = static_cast<uint8_t>(px[k] >> 8);
                o[3 * k + 2] = static_cast<uint8_t>(px[k] >> 16);
                if (pano_mask) pano_mask[__umul24(oy, a.mask_stride) + ox + k] = a.mask_plane[static_cast<size_t>(oy) * a.out_w + ox + k];
            }
        }
    }
}


// ---- the LDS-staged form ---------------------------------------------------------------------------------------
// Every global access of the loop above that is not 16 bytes per lane costs the texture path one slot per 32-byte
// sector it touches; gathers touch a sector or two per LANE, and rows r and r + 1 of a tile ask for the same source
// lines back to back (hit-on-miss stalls: half of the L1's busy cycles, rocprofv3 TCP_PENDING_STALL_CYCLES).  So the
// segment's source window goes through the L1 exactly once, as coalesced 12-byte-per-lane loads of four pixels,
// into LDS as 4-byte pixels (B G R x), and the bilinear taps come out of LDS with ds_read2_b32: no funnel shift, no
// cache.  tools/ubench/sampler.hip: 366 -> 555 Gpixel/s for the same bytes in and out.
//
// With 4-byte pixels the weights go the other way round: vertical pass first, in the 16-bit lanes of a register
// (8 bits x 6 bits; B and R of one pixel, or G of both), horizontal pass as v_dot2_u32_u16 over (left, right) pairs
// with the weights times 64, which again leaves the rounded byte in bits 16..23.  Same integers:
//   (sum_x wx * (sum_y wy * v) + 512) >> 10.
// ROWS rows per lane: a block has 32 lanes x (32 / ROWS) row groups = 1024 / ROWS threads over one staged window
#ifndef MST_COMPOSE_WAVES_PASTE
#define MST_COMPOSE_WAVES_PASTE 8
#endif
#ifndef MST_COMPOSE_WAVES_FEATHER
#define MST_COMPOSE_WAVES_FEATHER 6
#endif
template <int MODE, int ROWS>
__global__ __launch_bounds__(1024 / ROWS) __attribute__((amdgpu_waves_per_eu(ROWS == 2 ? (MODE == kCompModeFeather ? MST_COMPOSE_WAVES_FEATHER : MST_COMPOSE_WAVES_PASTE) : 2)))
void k_compose_lds(ComposeArgs a, FrameBatch fb)
{
    constexpr int NT = 1024 / ROWS;
    extern __shared__ __attribute__((aligned(16))) uint32_t s_win[];
    const int pair = xcd_run_index(blockIdx.x, a.n_tiles * a.n_sets);
    const int set = pair % a.n_sets;
    const ComposeTile t = a.tiles[pair / a.n_sets];
    const int tx = t.tx_ty & 0xFFFF, ty = t.tx_ty >> 16;
    uint8_t* __restrict__ pano = a.pano[set];
    uint8_t* __restrict__ pano_mask = a.pano_mask[set];
    const FrameArgs& frames = fb.f[set];
    const int lane = threadIdx.x & 31, rgrp = threadIdx.x >> 5;
    const int ox = tx * kCompTW + lane * 4;
    const bool aligned_rows = ((static_cast<uint32_t>(reinterpret_cast<uintptr_t>(pano)) | a.pano_stride) & 3u) == 0;
    const bool pure = MODE == kCompModeFeather && t.seg_count == 1 && (a.segs[t.seg_first].flags & kCompSegUnit);
    uint32_t px[ROWS][4];
    uint32_t acc_bg[ROWS][4], acc_r[ROWS][4];     // feather across cameras: B | G << 16, R (sums below 2^13)
#pragma unroll
    for (int it = 0; it < ROWS; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) { px[it][k] = 0u; acc_bg[it][k] = 0u; acc_r[it][k] = 0u; }
    f32x4 ws[ROWS];                               // dst_weight_map_, summed in feed order as the segments go by
#pragma unroll
    for (int it = 0; it < ROWS; ++it) ws[it] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (uint32_t s = 0; s < t.seg_count; ++s) {
        const ComposeSeg seg = a.segs[t.seg_first + s];
        const uint32_t stride = frames.stride[seg.cam];
        const size_t tbl = static_cast<size_t>(t.seg_first + s) * kCompPx + (rgrp * ROWS) * kCompTW + lane * 4;
        // the lane's table, gain and weight rows do not depend on the staged window: their loads go out first, so
        // that one round trip to memory covers them and the staging loads
        u32x4 e_r[ROWS];
        f32x4 g_r[ROWS], w_r[ROWS];
#pragma unroll
        for (int it = 0; it < ROWS; ++it) {
            e_r[it] = *(const u32x4 MST_GLOBAL*)(a.table + tbl + it * kCompTW);
            g_r[it] = f32x4{1.f, 1.f, 1.f, 1.f};
            if (a.gain) g_r[it] = *(const f32x4 MST_GLOBAL*)(a.gain + tbl + it * kCompTW);
            w_r[it] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (MODE == kCompModeFeather && !pure) w_r[it] = *(const f32x4 MST_GLOBAL*)(a.weight + tbl + it * kCompTW);
        }
        if (s) __syncthreads();                          // the previous segment's window has been read
        {
            const uint8_t MST_GLOBAL* wbase = gp(frames.ptr[seg.cam]) + (static_cast<ptrdiff_t>(seg.by) * static_cast<ptrdiff_t>(stride) + static_cast<ptrdiff_t>(seg.wx0) * 3);
            if (MODE == kCompModePaste && a.pad_w) stage_window<NT, true, MST_STAGE_U_PASTE>(s_win, wbase, stride, seg.groups, seg.rows, seg.wx0, seg.by, a.pad_w, a.pad_h);
            else stage_window<NT, false, (MODE == kCompModeFeather ? MST_STAGE_U_FEATHER : MST_STAGE_U_PASTE)>(s_win, wbase, stride, seg.groups, seg.rows);
        }
        __syncthreads();
        const uint32_t pitch_b = static_cast<uint32_t>(seg.groups) * 16u;
        const char* win = reinterpret_cast<const char*>(s_win) + (seg.bx - seg.wx0) * 4;
#pragma unroll
        for (int it = 0; it < ROWS; ++it) {
            const u32x4 e = e_r[it];
            const f32x4 g = g_r[it], w = w_r[it];
            if (MODE == kCompModeFeather) ws[it] += w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t dx = (e[k] >> 12) & 1023u, dy = (e[k] >> 22) & 511u;
                const uint32_t addr = __umul24(dy, pitch_b) + dx * 4u;
                const uint32_t v = sample_lds(reinterpret_cast<const uint32_t*>(win + addr),
                                              reinterpret_cast<const uint32_t*>(win + addr + pitch_b), e[k], g[k]);
                const bool valid = (e[k] & kEntryValid) != 0;
                if (MODE != kCompModeFeather || pure) {
                    px[it][k] = valid ? v : px[it][k];
                } else if (valid) {
                    const uint32_t tb = static_cast<uint32_t>(static_cast<int>(static_cast<float>(v & 0xFFu) * w[k]));
                    const uint32_t tg = static_cast<uint32_t>(static_cast<int>(static_cast<float>((v >> 8) & 0xFFu) * w[k]));
                    const uint32_t tr = static_cast<uint32_t>(static_cast<int>(static_cast<float>((v >> 16) & 0xFFu) * w[k]));
                    acc_bg[it][k] += tb | tg << 16;
                    acc_r[it][k] += tr;
                }
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        if (MODE == kCompModeFeather) {
            if (pure) {
#pragma unroll
                for (int k = 0; k < 4; ++k) px[it][k] = bytes_toward_zero(px[it][k]);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool m = ws[it][k] > kWeightEps;
                    const SharedDivisor sd = shared_divisor(ws[it][k] + kWeightEps);
                    const int cb = static_cast<int>(acc_bg[it][k] & 0xFFFFu), cg = static_cast<int>(acc_bg[it][k] >> 16), cr = static_cast<int>(acc_r[it][k]);
                    const uint32_t o = feather_norm(cb, sd) | feather_norm(cg, sd) << 8 | feather_norm(cr, sd) << 16;
                    px[it][k] = m ? o : 0u;
                }
            }
        }
    }
    // A lane holds the columns l + 32 k of its rows; the panorama wants runs of four adjacent pixels (12 bytes) per lane:
    // the tile goes through LDS once (over the last window: 16 KB, conflict-free both ways)
    __syncthreads();                                     // every wave is done with the last segment's window
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        uint32_t* trow = s_win + (rgrp * ROWS + it) * kCompTW + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) trow[32 * k] = px[it][k];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        const int row = rgrp * ROWS + it, oy = ty * kCompTH + row;
        if (oy >= a.out_h || ox >= a.out_w) continue;
        const u32x4 v = *reinterpret_cast<const u32x4*>(s_win + row * kCompTW + 4 * lane);
        uint8_t* o = pano + __umul24(oy, a.pano_stride) + ox * 3;
        if (ox + 3 < a.out_w) {
            const uint32_t w0 = v.x | v.y << 24, w1 = v.y >> 8 | v.z << 16, w2 = v.z >> 16 | v.w << 8;
            if (aligned_rows) *reinterpret_cast<u32x3 MST_GLOBAL*>(gp(o)) = u32x3{w0, w1, w2};
            else *reinterpret_cast<Bytes12*>(o) = Bytes12{w0, w1, w2};
            if (pano_mask)
                *reinterpret_cast<u32_any*>(pano_mask + __umul24(oy, a.mask_stride) + ox) =
                    *reinterpret_cast<const u32_any*>(a.mask_plane + static_cast<size_t>(oy) * a.out_w + ox);
        } else {
            for (int k = 0; k < 4 && ox + k < a.out_w; ++k) {
                o[3 * k] = static_cast<uint8_t>(v[k]); o[3 * k + 1] = static_cast<uint8_t>(v[k] >> 8);
                o[3 * k + 2] = static_cast<uint8_t>(v[k] >> 16);
                if (pano_mask) pano_mask[__umul24(oy, a.mask_stride) + ox + k] = a.mask_plane[static_cast<size_t>(oy) * a.out_w + ox + k];
            }
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------

void launch_compose_classify(const ComposeArgs& a, int n_canvas_tiles, uint32_t* d_out, hipStream_t s)
{
    if (n_canvas_tiles <= 0) return;
    hipLaunchKernelGGL(k_compose_classify, dim3(n_canvas_tiles), dim3(256), 0, s, a, d_out);
}

void launch_compose_mask(const ComposeArgs& a, uint8_t* d_plane, hipStream_t s)
{
    hipLaunchKernelGGL(k_compose_mask, dim3((a.out_w + 63) / 64, (a.out_h + 3) / 4), dim3(256), 0, s, a, d_plane);
}

void launch_compose_permute(const uint32_t* d_seg_ids, int n_segs, uint32_t* d_table, float* d_gain, float* d_weight, hipStream_t s)
{
    if (n_segs <= 0) return;
    hipLaunchKernelGGL(k_compose_permute, dim3(n_segs), dim3(256), 0, s, d_seg_ids, d_table, d_gain, d_weight);
}

void launch_compose_table(const ComposeArgs& a, int kind, int n_segs, ComposeSeg* d_segs, ComposeTile* d_tiles,
                          const uint32_t* d_seg_tile, uint32_t* d_table, float* d_gain, float* d_weight, hipStream_t s)
{
    if (n_segs <= 0) return;
#define MST_LAUNCH_CT(K) hipLaunchKernelGGL(k_compose_table<K>, dim3(n_segs), dim3(256), 0, s, a, d_segs, d_tiles, d_seg_tile, d_table, d_gain, d_weight)
    if (kind == 0) MST_LAUNCH_CT(0); else if (kind == 1) MST_LAUNCH_CT(1); else if (kind == 2) MST_LAUNCH_CT(2); else MST_LAUNCH_CT(3);
#undef MST_LAUNCH_CT
}

void launch_undistort_table(const UndistortCam& u, int tiles_x, int tiles_y, ComposeSeg* d_segs, ComposeTile* d_tiles, uint32_t* d_table,
                            int lds_budget_px, hipStream_t s)
{
    hipLaunchKernelGGL(k_undistort_table, dim3(tiles_x * tiles_y), dim3(256), 0, s, u, tiles_x, d_segs, d_tiles, d_table, lds_budget_px);
}

uint32_t compose_build_ablations()
{
#if defined(MST_ABLATE_STAGE) || defined(MST_ABLATE_NV12STAGE)
    return 8u;
#else
    return 0u;
#endif
}

const char* launch_compose(const ComposeArgs& a, const FrameBatch& frames, size_t lds_bytes, bool general, hipStream_t s)
{
    if (a.n_tiles <= 0 || a.n_sets <= 0) return nullptr;
    // (kernel names as rocprofv3 prints them: MODE 0 = paste / LUT, 2 = feather)
    static_assert(kCompModePaste == 0 && kCompModeFeather == 2, "kernel names below");
    if (general) {
        const dim3 gg(static_cast<unsigned>(a.n_tiles * a.n_sets * kGeneralSplit));
        if (a.mode == kCompModeFeather) { hipLaunchKernelGGL(k_compose_general<kCompModeFeather>, gg, dim3(256), 0, s, a, frames); return "k_compose_general<2>"; }
        hipLaunchKernelGGL(k_compose_general<kCompModePaste>, gg, dim3(256), 0, s, a, frames);
        return "k_compose_general<0>";
    }
    const dim3 grid(static_cast<unsigned>(a.n_tiles * a.n_sets));
    if (lds_bytes) {       // every tile of the list fits the staged form
        static const int rows_env = [] { const char* v = std::getenv("MST_COMPOSE_ROWS"); return v ? std::atoi(v) : 0; }();
        const int rows = rows_env ? rows_env : 2;     // 512 threads over one window: measured best (C3 / C5 LUT +15 % over 256)
#define MST_LAUNCH_CL(M, MN, R) [&] { hipLaunchKernelGGL((k_compose_lds<M, R>), grid, dim3(1024 / R), lds_bytes, s, a, frames); return "k_compose_lds<" #MN ", " #R ">"; }()
        if (a.mode == kCompModeFeather) return rows == 1 ? MST_LAUNCH_CL(kCompModeFeather, 2, 1) : rows == 2 ? MST_LAUNCH_CL(kCompModeFeather, 2, 2) : MST_LAUNCH_CL(kCompModeFeather, 2, 4);
        return rows == 1 ? MST_LAUNCH_CL(kCompModePaste, 0, 1) : rows == 2 ? MST_LAUNCH_CL(kCompModePaste, 0, 2) : MST_LAUNCH_CL(kCompModePaste, 0, 4);
#undef MST_LAUNCH_CL
    }
    if (a.n_tiles * a.n_sets < 4096) {      // a short list: four blocks per tile
        const dim3 g4(grid.x * 4);
        if (a.mode == kCompModeFeather) { hipLaunchKernelGGL((k_compose<kCompModeFeather, 1>), g4, dim3(256), 0, s, a, frames); return "k_compose<2, 1>"; }
        hipLaunchKernelGGL((k_compose<kCompModePaste, 1>), g4, dim3(256), 0, s, a, frames);
        return "k_compose<0, 1>";
    }
    if (a.mode == kCompModeFeather) { hipLaunchKernelGGL((k_compose<kCompModeFeather, 4>), grid, dim3(256), 0, s, a, frames); return "k_compose<2, 4>"; }
    hipLaunchKernelGGL((k_compose<kCompModePaste, 4>), grid, dim3(256), 0, s, a, frames);
    return "k_compose<0, 4>";
}

}  // namespace mst
