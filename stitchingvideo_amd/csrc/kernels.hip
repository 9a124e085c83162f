// kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4, wave64): stage-level entry points,
// and calibration-time static products.
// The multi-band frame path lives in kernels_mb.hip.
//
// All of these are HBM-bound integer/byte image operators: no MFMA.  Arithmetic follows the
// reference bit for bit (R64 = /root/reference/stitching/OpenCV2.4.11-Stitching-64bit/
// OpenCV2.4.11-Stitching):
//   backward_map          {Plane,Spherical,Cylindrical}Projector::mapBackward
//                                                           detail/warpers_inl.hpp:221-300
//   sample_*              cv::remap INTER_LINEAR (5-bit fractions, 15-bit weights) / INTER_NEAREST,
//                         called at detail/warpers_inl.hpp:96, DRV:752
//   gain_at / apply_gain  BlocksGainCompensator::apply      R64/src/exposure_compensate.cpp:225-246
//                         GainCompensator::apply            R64/src/exposure_compensate.cpp:150-153
//   k_weight_level0       mask.convertTo(32F,1/255) + copyMakeBorder(CONSTANT)   R64/src/blenders.cpp:287,295
//   k_pyrdown_f32         cv::pyrDown CV_32F (weights)      R64/src/blenders.cpp:298
//   k_band_weight_sum     dst_band_weights_ accumulation    R64/src/blenders.cpp:324
// (the single-pass feather / no-blend / driver-LUT composites live in kernels_compose.hip)
#include "kernels.hpp"
#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace mst {

using namespace dev;

namespace {

constexpr int kTileW = 64, kTileH = 4;   // 256-thread blocks: one wave per row of 64 pixels

// ------------------------------------------------------------------------------------------
// stage-level kernels
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_build_maps(const CamDesc* __restrict__ cam, float* __restrict__ xmap,
                                                    float* __restrict__ ymap, int w, int h)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= w || y >= h) return;
    float mx, my;
    backward_map(*cam, x, y, mx, my);
    xmap[static_cast<size_t>(y) * w + x] = mx;
    ymap[static_cast<size_t>(y) * w + x] = my;
}

template <int CN, bool LINEAR, bool REFLECT_BORDER>
__global__ __launch_bounds__(256) void k_warp_u8(const CamDesc* __restrict__ cam, const uint8_t* __restrict__ src,
                                                 size_t src_stride, uint8_t* __restrict__ dst, size_t dst_stride,
                                                 int w, int h)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= w || y >= h) return;
    const CamDesc& c = *cam;
    float mx, my;
    backward_map(c, x, y, mx, my);
    uint8_t* d = dst + static_cast<size_t>(y) * dst_stride + x * CN;
    if (LINEAR) {
        const Taps t = linear_taps<REFLECT_BORDER>(mx, my, c.src_w, c.src_h);
#pragma unroll
        for (int k = 0; k < CN; ++k)
            d[k] = static_cast<uint8_t>(blend_taps(t, tap_u8(src, src_stride, CN, t.x0, t.y0, k),
                                                   tap_u8(src, src_stride, CN, t.x1, t.y0, k),
                                                   tap_u8(src, src_stride, CN, t.x0, t.y1, k),
                                                   tap_u8(src, src_stride, CN, t.x1, t.y1, k)));
    } else {
        // remapNearest: saturate_cast<short>(cvRound(map))
        int sx = sat_s16(cv_round(mx)), sy = sat_s16(cv_round(my));
        if (REFLECT_BORDER) { sx = reflect<0>(sx, c.src_w); sy = reflect<0>(sy, c.src_h); }
        const bool inside = static_cast<unsigned>(sx) < static_cast<unsigned>(c.src_w) &&
                            static_cast<unsigned>(sy) < static_cast<unsigned>(c.src_h);
#pragma unroll
        for (int k = 0; k < CN; ++k)
            d[k] = inside ? src[static_cast<size_t>(sy) * src_stride + sx * CN + k] : 0;
    }
}

// cv::remap(src, dst, map1 CV_16SC2, map2 CV_16UC1, INTER_LINEAR) with the default constant border (DRV:741):
// the integer tap comes from map1, the 5-bit fractions from map2, the weights and rounding are those of the
// float-map path.  One launch covers every camera of every frame set of a batch: blockIdx.z = set * n_cams + cam.
// A thread produces four consecutive pixels (one 16-byte and one 8-byte map load, 12 output bytes stored as three
// dwords when the row allows it); a pixel whose 2x2 window lies inside the frame reads it as two aligned 12-byte
// loads, the others resolve the constant border tap by tap.
constexpr int kRemapPx = 4;
__global__ __launch_bounds__(256) void k_remap_fixed(const UndistortCam* __restrict__ ucams, int n_cams, FrameBatch src,
                                                     FrameBatch dst, int bx, int by, int n_sets)
{
    // 1-D grid over (camera, block row, block column, frame set), frame set fastest, walked in XCD runs (blocks are
    // dealt to the 8 XCDs round-robin: XCD k = blockIdx.x % 8 takes one contiguous run): the frame sets of a block
    // are dispatched together onto one XCD, so its 6 bytes per pixel of maps -- frame-invariant, and as large a
    // stream as the frames themselves -- come from HBM once per launch instead of once per frame set.
    const int total = bx * by * n_cams * n_sets;
    const int q8 = total >> 3, r8 = total & 7, xk = blockIdx.x & 7, xj = blockIdx.x >> 3;
    int idx1 = xk * q8 + min(xk, r8) + xj;
    const int set = idx1 % n_sets; idx1 /= n_sets;
    const int bxi = idx1 % bx; idx1 /= bx;
    const int byi = idx1 % by, cam = idx1 / by;
    const UndistortCam u = ucams[cam];
    const uint8_t* __restrict__ s = src.f[set].ptr[cam];
    if (!u.map1 || !s) return;                      // no maps, or a camera that is not part of this call
    const int x0 = (bxi * kTileW + threadIdx.x) * kRemapPx, y = byi * kTileH + threadIdx.y;
    if (x0 >= u.w || y >= u.h) return;
    const uint32_t sstride = src.f[set].stride[cam], dstride = dst.f[set].stride[cam];
    uint8_t* __restrict__ drow = const_cast<uint8_t*>(dst.f[set].ptr[cam]) + static_cast<size_t>(y) * dstride;
    const uint32_t lo2 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(s)) & 3u;
    const uint8_t* __restrict__ s_al = s - lo2;
    const uint32_t idx = static_cast<uint32_t>(y) * u.w + x0;
    const int npx = min(kRemapPx, u.w - x0);
    uint32_t xy[kRemapPx];
    uint32_t fr[kRemapPx];
    const int16_t MST_GLOBAL* m1 = gp(u.map1) + 2 * static_cast<size_t>(idx);
    const uint16_t MST_GLOBAL* m2 = gp(u.map2) + idx;
    if (npx == kRemapPx && (idx & 3u) == 0) {         // 16-byte / 8-byte aligned: the maps are tight and w * y + x0 is 0 mod 4
        const uint4 v = load_global(reinterpret_cast<const uint4*>(u.map1 + 2 * static_cast<size_t>(idx)));
        const uint2 f = load_global(reinterpret_cast<const uint2*>(u.map2 + idx));
        xy[0] = v.x; xy[1] = v.y; xy[2] = v.z; xy[3] = v.w;
        fr[0] = f.x & 0xFFFFu; fr[1] = f.x >> 16; fr[2] = f.y & 0xFFFFu; fr[3] = f.y >> 16;
    } else {
#pragma unroll
        for (int k = 0; k < kRemapPx; ++k) {
            const int kk = min(k, npx - 1);
            xy[k] = static_cast<uint32_t>(static_cast<uint16_t>(m1[2 * kk])) | static_cast<uint32_t>(static_cast<uint16_t>(m1[2 * kk + 1])) << 16;
            fr[k] = m2[kk];
        }
    }
    // issue the window loads of all four pixels before the first use
    uint32_t a[kRemapPx][3], c[kRemapPx][3], sh0[kRemapPx], sh1[kRemapPx];
    bool fast[kRemapPx];
#pragma unroll
    for (int k = 0; k < kRemapPx; ++k) {
        const int sx = static_cast<int>(static_cast<short>(xy[k] & 0xFFFFu)), sy = static_cast<int>(xy[k]) >> 16;
        fast[k] = sx >= 0 && sx < u.w - 1 && sy >= 0 && sy < u.h - 1 && (sx < u.w - 3 || sy < u.h - 2);
        const uint32_t ox = fast[k] ? static_cast<uint32_t>(sx) : 0u, oy = fast[k] ? static_cast<uint32_t>(sy) : 0u;
        const uint32_t o0 = __umul24(oy, sstride) + ox * 3u + lo2, o1 = o0 + sstride;
        sh0[k] = o0 & 3u; sh1[k] = o1 & 3u;
        const uint32_t* q0 = reinterpret_cast<const uint32_t*>(s_al + (o0 & ~3u));
        const uint32_t* q1 = reinterpret_cast<const uint32_t*>(s_al + (o1 & ~3u));
        a[k][0] = q0[0]; a[k][1] = q0[1]; a[k][2] = q0[2];
        c[k][0] = q1[0]; c[k][1] = q1[1]; c[k][2] = q1[2];
    }
    uint32_t px[kRemapPx];
#pragma unroll
    for (int k = 0; k < kRemapPx; ++k) {
        const int fx = fr[k] & 31, fy = (fr[k] >> 5) & 31;
        int o[3];
        if (fast[k]) {
            bilinear_window_c3(a[k][0], a[k][1], a[k][2], c[k][0], c[k][1], c[k][2], sh0[k], sh1[k], fx, fy, o);
        } else {
            const int sx = static_cast<int>(static_cast<short>(xy[k] & 0xFFFFu)), sy = static_cast<int>(xy[k]) >> 16;
            Taps t;
            t.w00 = (32 - fx) * (32 - fy) * 32; t.w01 = fx * (32 - fy) * 32;
            t.w10 = (32 - fx) * fy * 32;        t.w11 = fx * fy * 32;
            t.x0 = static_cast<unsigned>(sx) < static_cast<unsigned>(u.w) ? sx : -1;
            t.x1 = static_cast<unsigned>(sx + 1) < static_cast<unsigned>(u.w) ? sx + 1 : -1;
            t.y0 = static_cast<unsigned>(sy) < static_cast<unsigned>(u.h) ? sy : -1;
            t.y1 = static_cast<unsigned>(sy + 1) < static_cast<unsigned>(u.h) ? sy + 1 : -1;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)
                o[ch] = blend_taps(t, tap_u8(s, sstride, 3, t.x0, t.y0, ch), tap_u8(s, sstride, 3, t.x1, t.y0, ch),
                                   tap_u8(s, sstride, 3, t.x0, t.y1, ch), tap_u8(s, sstride, 3, t.x1, t.y1, ch));
        }
        px[k] = static_cast<uint32_t>(o[0]) | static_cast<uint32_t>(o[1]) << 8 | static_cast<uint32_t>(o[2]) << 16;
    }
    uint8_t* d = drow + x0 * 3;
    if (npx == kRemapPx && ((reinterpret_cast<uintptr_t>(d)) & 3u) == 0) {
        uint32_t MST_GLOBAL* d32 = (uint32_t MST_GLOBAL*)d;               // B0 G0 R0 B1 | G1 R1 B2 G2 | R2 B3 G3 R3
        d32[0] = px[0] | px[1] << 24;
        d32[1] = px[1] >> 8 | px[2] << 16;
        d32[2] = px[2] >> 16 | px[3] << 8;
    } else {
        for (int k = 0; k < npx; ++k) {
            d[3 * k] = static_cast<uint8_t>(px[k]); d[3 * k + 1] = static_cast<uint8_t>(px[k] >> 8);
            d[3 * k + 2] = static_cast<uint8_t>(px[k] >> 16);
        }
    }
}

__global__ __launch_bounds__(256) void k_exposure_apply(const CamDesc* __restrict__ cam, uint8_t* __restrict__ img,
                                                        size_t stride, int w, int h)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= w || y >= h) return;
    const float g = gain_at(*cam, x, y);
    uint8_t* p = img + static_cast<size_t>(y) * stride + x * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k) p[k] = static_cast<uint8_t>(apply_gain(p[k], g));
}

// ------------------------------------------------------------------------------------------
// calibration-time static products
// ------------------------------------------------------------------------------------------

// mask.convertTo(CV_32F, 1./255.) + copyMakeBorder(BORDER_CONSTANT)   blenders.cpp:287,295
// weight_type CV_16S (alpha == 0): mask.convertTo(CV_16S); add(weight, 1, weight, mask != 0)   blenders.cpp:289-292;
// those weights (0, 2..256) and every level pyrDown'ed from them are small integers, kept exactly in the float arrays
__global__ __launch_bounds__(256) void k_weight_level0(const CamDesc* __restrict__ cams, int cam,
                                                       float* __restrict__ w0, int bw, int bh, float alpha)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= bw || y >= bh) return;
    const CamDesc& c = cams[cam];
    const int rx = x - c.left, ry = y - c.top;
    float w = 0.f;
    if (static_cast<unsigned>(rx) < static_cast<unsigned>(c.roi_w) &&
        static_cast<unsigned>(ry) < static_cast<unsigned>(c.roi_h))
    {
        const int m = gp(c.mask)[static_cast<size_t>(ry) * c.roi_w + rx];
        w = alpha != 0.f ? static_cast<float>(m) * alpha + 0.f : static_cast<float>(m + (m != 0));
    }
    w0[static_cast<size_t>(y) * bw + x] = w;
}

// the final mask inside the bordered level-0 window: copyMakeBorder(mask, BORDER_CONSTANT 0)   blenders.cpp:295
__global__ __launch_bounds__(256) void k_border_mask(const CamDesc* __restrict__ cams, int cam,
                                                     uint8_t* __restrict__ out, int bw, int bh)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= bw || y >= bh) return;
    const CamDesc& c = cams[cam];
    const int rx = x - c.left, ry = y - c.top;
    uint8_t m = 0;
    if (static_cast<unsigned>(rx) < static_cast<unsigned>(c.roi_w) &&
        static_cast<unsigned>(ry) < static_cast<unsigned>(c.roi_h))
        m = gp(c.mask)[static_cast<size_t>(ry) * c.roi_w + rx];
    out[static_cast<size_t>(y) * bw + x] = m;
}

// cv::pyrDown CV_32FC1: scalar horizontal pass, then the vertical pass in the order of the x64
// build: PyrDownVec_32f (SSE) for x < width - width%8, scalar tail after it.
__global__ __launch_bounds__(256) void k_pyrdown_f32(const float* __restrict__ src, int w, int h,
                                                     float* __restrict__ dst, int dw, int dh)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= dw || y >= dh) return;
    const int xm2 = reflect<1>(2 * x - 2, w), xm1 = reflect<1>(2 * x - 1, w), x0 = reflect<1>(2 * x, w),
              xp1 = reflect<1>(2 * x + 1, w), xp2 = reflect<1>(2 * x + 2, w);
    float row[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float* s = src + static_cast<size_t>(reflect<1>(2 * y + k - 2, h)) * w;
        row[k] = s[x0] * 6 + (s[xm1] + s[xp1]) * 4 + s[xm2] + s[xp2];
    }
    const int vec_end = dw - (dw % 8);
    float out;
    if (x < vec_end) {
        float r0 = row[0] + row[4];
        float r1 = (row[1] + row[3]) + row[2];
        r0 = r0 + (row[2] + row[2]);
        out = (r0 + r1 * 4.f) * (1.f / 256);
    } else {
        out = (row[2] * 6 + (row[1] + row[3]) * 4 + row[0] + row[4]) * (1.f / 256);
    }
    dst[static_cast<size_t>(y) * dw + x] = out;
}

// cv::pyrDown on a CV_16SC1 weight level (weight_type CV_16S, blenders.cpp:298): integer [1 4 6 4 1]^2,
// (v + 128) >> 8, BORDER_REFLECT_101; values travel as exact small integers in float arrays
__global__ __launch_bounds__(256) void k_pyrdown_s16(const float* __restrict__ src, int w, int h,
                                                     float* __restrict__ dst, int dw, int dh)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= dw || y >= dh) return;
    int xi[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) xi[k] = reflect<1>(2 * x + k - 2, w);
    int acc = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const float* s = src + static_cast<size_t>(reflect<1>(2 * y + k - 2, h)) * w;
        const int r = static_cast<int>(s[xi[2]]) * 6 + (static_cast<int>(s[xi[1]]) + static_cast<int>(s[xi[3]])) * 4 +
                      static_cast<int>(s[xi[0]]) + static_cast<int>(s[xi[4]]);
        acc += r * ((k == 2) ? 6 : ((k == 1 || k == 3) ? 4 : 1));
    }
    dst[static_cast<size_t>(y) * dw + x] = static_cast<float>((acc + 128) >> 8);
}

// dst_band_weights_[level]: `dst_weight_row[x] += weight_row[x_]` in feed order   blenders.cpp:324
__global__ __launch_bounds__(256) void k_band_weight_sum(const CamDesc* __restrict__ cams, int n_cams, int level,
                                                         float* __restrict__ ws, int cw, int ch)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= cw || y >= ch) return;
    float acc = 0.f;
    for (int i = 0; i < n_cams; ++i) {
        const CamDesc& c = cams[i];
        const int lx = x - c.ox[level], ly = y - c.oy[level];
        if (static_cast<unsigned>(lx) < static_cast<unsigned>(c.lw[level]) &&
            static_cast<unsigned>(ly) < static_cast<unsigned>(c.lh[level]))
            acc += gp(c.W[level])[static_cast<size_t>(ly) * c.lw[level] + lx];
    }
    ws[static_cast<size_t>(y) * cw + x] = acc;
}

// The mask fix-up of the compose loop (stitcher.cpp:291-294 = DRV:676-678), one thread per pixel of the final mask:
//   dilate(seam mask, 3x3)  ->  resize(INTER_LINEAR) to the warped size  ->  & warp(all-255, NEAREST, CONSTANT)
// cv::resize on CV_8U works in 11-bit fixed point: rows of D = S[x0] * a0 + S[x1] * a1 (ints), then
// (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2; the dilated values are formed on the fly.
__global__ __launch_bounds__(256) void k_mask_fixup(const CamDesc* __restrict__ cam, const uint8_t* __restrict__ seam,
                                                    int sw, int sh, const int32_t* __restrict__ xo0,
                                                    const int32_t* __restrict__ xo1, const int16_t* __restrict__ xa0,
                                                    const int16_t* __restrict__ xa1, const int32_t* __restrict__ yo0,
                                                    const int32_t* __restrict__ yo1, const int16_t* __restrict__ yb0,
                                                    const int16_t* __restrict__ yb1, uint8_t* __restrict__ out, int w, int h)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= w || y >= h) return;
    const CamDesc& c = *cam;
    auto dil = [&](int px, int py) {              // cv::dilate, 3x3 rectangle, pixels outside the image ignored
        int m = 0;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = py + dy;
            if (yy < 0 || yy >= sh) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = px + dx;
                if (xx < 0 || xx >= sw) continue;
                m = max(m, static_cast<int>(seam[static_cast<size_t>(yy) * sw + xx]));
            }
        }
        return m;
    };
    const int x0 = xo0[x], x1 = xo1[x], a0 = xa0[x], a1 = xa1[x];
    const int y0 = yo0[y], y1 = yo1[y], b0 = yb0[y], b1 = yb1[y];
    const int d0 = dil(x0, y0) * a0 + dil(x1, y0) * a1;
    const int d1 = dil(x0, y1) * a0 + dil(x1, y1) * a1;
    const int v = sat_u8((((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2);
    // warp(Mat(size, CV_8U, 255), INTER_NEAREST, BORDER_CONSTANT): 255 where the rounded map lands in the source
    float mx, my;
    backward_map(c, x, y, mx, my);
    const int sx = sat_s16(cv_round(mx)), sy = sat_s16(cv_round(my));
    const bool inside = static_cast<unsigned>(sx) < static_cast<unsigned>(c.src_w) &&
                        static_cast<unsigned>(sy) < static_cast<unsigned>(c.src_h);
    out[static_cast<size_t>(y) * w + x] = static_cast<uint8_t>(inside ? v : 0);
}

// createWeightMap (blenders.cpp:427-432) on the device: distanceTransform(mask, CV_DIST_L1, 3) is a 3x3 chamfer with
// costs 1 (edge) and 2 (diagonal) in 16.16 fixed point over a frame of INT_MAX >> 2 -- i.e. the exact city-block
// distance to the nearest zero pixel, or "frame value + distance to the frame" where that is smaller -- which
// separates into a column pass and a row pass of running minima.  Pass 1: per column, distance along the column.
constexpr int kDtInf = INT_MAX >> 2, kDtStep = 1 << 16;
__global__ __launch_bounds__(256) void k_dt_columns(const uint8_t* __restrict__ mask, int w, int h, int32_t* __restrict__ g)
{
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= w) return;
    int run = kDtInf;                              // the frame row above the image
    for (int y = 0; y < h; ++y) {
        run = mask[static_cast<size_t>(y) * w + x] ? run + kDtStep : 0;
        g[static_cast<size_t>(y) * w + x] = run;
    }
    run = kDtInf;                                  // the frame row below
    for (int y = h - 1; y >= 0; --y) {
        const size_t i = static_cast<size_t>(y) * w + x;
        run = mask[i] ? run + kDtStep : 0;
        g[i] = min(g[i], run);
    }
}
// Pass 2: per row, min over columns of g + |dx| (and the frame columns left and right), then the float scaling of
// distanceTransform (* 2^-16), `weight * sharpness` and threshold(..., 1, THRESH_TRUNC).  One wave per row: a
// running minimum is a scan, done here as two sequential sweeps by lane 0's neighbours in chunks.
// RAW: the distance itself (distanceTransform's output: value * 2^-16), for the Voronoi seam finder
template <bool RAW>
__global__ __launch_bounds__(64) void k_dt_rows(const int32_t* __restrict__ g, int w, int h, float sharpness,
                                                float* __restrict__ weight)
{
    const int y = blockIdx.x * 64 + threadIdx.x;
    if (y >= h) return;
    const int32_t* __restrict__ row = g + static_cast<size_t>(y) * w;
    float* __restrict__ out = weight + static_cast<size_t>(y) * w;
    // backward sweep first, kept in the output row as bit patterns; forward sweep finishes the value
    int run = kDtInf;                              // frame column right of the image
    for (int x = w - 1; x >= 0; --x) {
        run = min(row[x], run + kDtStep);
        out[x] = __int_as_float(run);
    }
    run = kDtInf;                                  // frame column left
    for (int x = 0; x < w; ++x) {
        run = min(row[x], run + kDtStep);
        const int v = min(run, __float_as_int(out[x]));
        float d = static_cast<float>(v) * (1.f / 65536.f);
        if (!RAW) {
            d = d * sharpness + 0.f;
            d = d > 1.f ? 1.f : d;
        }
        out[x] = d;
    }
}

// VoronoiSeamFinder::findInPair (seam_finders.cpp:108-160), step 1: cut the two sub-masks around the overlap (with the
// reference's 10-pixel gap), drop the collision, and write the distance transforms' inputs `unique == 0`.
struct VoronoiPair {
    uint8_t* m1; uint8_t* m2;       // the two masks (tight), updated in place by step 3
    int32_t w1, h1, w2, h2;
    int32_t ox1, oy1, ox2, oy2;     // roi.tl - tl1, roi.tl - tl2
    int32_t rw, rh, gap;
};
__global__ __launch_bounds__(256) void k_voronoi_cut(VoronoiPair p, uint8_t* __restrict__ inv1, uint8_t* __restrict__ inv2)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    const int sw = p.rw + 2 * p.gap, sh = p.rh + 2 * p.gap;
    if (x >= sw || y >= sh) return;
    const int x1 = p.ox1 + x - p.gap, y1 = p.oy1 + y - p.gap, x2 = p.ox2 + x - p.gap, y2 = p.oy2 + y - p.gap;
    const int s1 = (x1 >= 0 && y1 >= 0 && x1 < p.w1 && y1 < p.h1) ? p.m1[static_cast<size_t>(y1) * p.w1 + x1] : 0;
    const int s2 = (x2 >= 0 && y2 >= 0 && x2 < p.w2 && y2 < p.h2) ? p.m2[static_cast<size_t>(y2) * p.w2 + x2] : 0;
    const bool collision = s1 != 0 && s2 != 0;
    const size_t i = static_cast<size_t>(y) * sw + x;
    inv1[i] = (collision || s1 == 0) ? 255 : 0;          // unique1 == 0
    inv2[i] = (collision || s2 == 0) ? 255 : 0;
}
// step 3: `seam = dist1 < dist2` over the overlap: the nearer image keeps the pixel
__global__ __launch_bounds__(256) void k_voronoi_apply(VoronoiPair p, const float* __restrict__ d1, const float* __restrict__ d2)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= p.rw || y >= p.rh) return;
    const int sw = p.rw + 2 * p.gap;
    const size_t i = static_cast<size_t>(y + p.gap) * sw + (x + p.gap);
    if (d1[i] < d2[i]) p.m2[static_cast<size_t>(p.oy2 + y) * p.w2 + (p.ox2 + x)] = 0;
    else p.m1[static_cast<size_t>(p.oy1 + y) * p.w1 + (p.ox1 + x)] = 0;
}

// GainCompensator::feed's overlap statistics (exposure_compensate.cpp:92-120): one thread per overlapping pair,
// walking the overlap row by row exactly as the reference does, so that Isum1 / Isum2 -- sums of correctly rounded
// double square roots -- are accumulated in the reference's order.  Pairs are small (32 x 32 blocks) and many.
__global__ __launch_bounds__(64) void k_overlap_stats(const FeedImage* __restrict__ imgs, const FeedPair* __restrict__ pairs,
                                                      int n_pairs, FeedStat* __restrict__ out)
{
    const int k = blockIdx.x * 64 + threadIdx.x;
    if (k >= n_pairs) return;
    const FeedPair p = pairs[k];
    const FeedImage ia = imgs[p.a], ib = imgs[p.b];
    int count = 0;
    double s1 = 0., s2 = 0.;
    for (int y = 0; y < p.h; ++y) {
        const uint8_t* r1 = ia.img + static_cast<size_t>(p.ay + y) * ia.istride + static_cast<size_t>(p.ax) * 3;
        const uint8_t* r2 = ib.img + static_cast<size_t>(p.by + y) * ib.istride + static_cast<size_t>(p.bx) * 3;
        const uint8_t* m1 = ia.mask + static_cast<size_t>(p.ay + y) * ia.mstride + p.ax;
        const uint8_t* m2 = ib.mask + static_cast<size_t>(p.by + y) * ib.mstride + p.bx;
        for (int x = 0; x < p.w; ++x) {
            if (m1[x] != 255 || m2[x] != 255) continue;
            ++count;
            const int b1 = r1[3 * x], g1 = r1[3 * x + 1], c1 = r1[3 * x + 2];
            const int b2 = r2[3 * x], g2 = r2[3 * x + 1], c2 = r2[3 * x + 2];
            s1 += sqrt(static_cast<double>(b1 * b1 + g1 * g1 + c1 * c1));
            s2 += sqrt(static_cast<double>(b2 * b2 + g2 * g2 + c2 * c2));
        }
    }
    out[k].sum_a = s1;
    out[k].sum_b = s2;
    out[k].count = count;
    out[k].pad = 0;
}

// Camera-sharded hand-off: copy the rectangles of one camera's payload between its pyramid levels and a
// packed buffer.  PACK: pyramid -> buffer (the sending rank), else buffer -> pyramid (the blending rank).
template <bool PACK>
__global__ __launch_bounds__(256) void k_payload_copy(const CamDesc* __restrict__ cam, const PayloadRect* __restrict__ rects,
                                                      uint32_t* __restrict__ buf)
{
    const PayloadRect r = rects[blockIdx.x];
    const CamDesc& c = *cam;
    uint32_t MST_GLOBAL* __restrict__ b = gp(buf) + r.off;
    const int n = r.w * r.h;
    if (r.level == 0) {
        // level 0 is packed BGR8 in the pyramid (device_types.hpp) and in the payload: 3 w h bytes, rows back to back.  Rectangles
        // that start on a four-pixel group and are whole groups wide (every tile interior but the last of a row) move as
        // 12-byte groups, the others byte by byte.
        uint8_t MST_GLOBAL* __restrict__ g8 = (uint8_t MST_GLOBAL*)c.G[0];
        uint8_t MST_GLOBAL* __restrict__ b8 = (uint8_t MST_GLOBAL*)b;
        if (MST_L0_PACKED && ((r.x0 | r.w) & 3) == 0) {
            const int gw = r.w >> 2;
            for (int i = threadIdx.x; i < gw * r.h; i += 256) {
                const int y = i / gw, g = i - y * gw;
                u32x3_t MST_GLOBAL* p = (u32x3_t MST_GLOBAL*)(g8 + static_cast<size_t>(r.y0 + y) * c.l0_pitch + 3 * static_cast<size_t>(r.x0 + 4 * g));
                u32x3_t MST_GLOBAL* q = (u32x3_t MST_GLOBAL*)(b8 + 12 * static_cast<size_t>(i));
                if (PACK) *q = *p; else *p = *q;
            }
        } else {
            for (int i = threadIdx.x; i < 3 * n; i += 256) {
                const int y = i / (3 * r.w), xb = i - y * (3 * r.w);
                // (the payload is packed BGR8 whatever the pyramid's level 0 is: byte xb of the row is channel xb % 3 of pixel xb / 3)
                const size_t in_row = MST_L0_PACKED ? static_cast<size_t>(xb) : static_cast<size_t>(xb / 3) * 4 + static_cast<size_t>(xb % 3);
                uint8_t MST_GLOBAL* p = g8 + static_cast<size_t>(r.y0 + y) * c.l0_pitch + kL0Bpp * static_cast<size_t>(r.x0) + in_row;
                if (PACK) b8[i] = *p; else *p = b8[i];
            }
        }
        return;
    }
    uint32_t MST_GLOBAL* __restrict__ g = gp(c.G[r.level]);
    const int lw = c.lw[r.level];
    for (int i = threadIdx.x; i < n; i += 256) {
        const int y = i / r.w, x = i - y * r.w;
        const size_t gi = static_cast<size_t>(r.y0 + y) * lw + (r.x0 + x);
        if (PACK) b[i] = g[gi];
        else g[gi] = b[i];
    }
}

// What the work-list planner needs to know about a weight pyramid level, one byte per pixel: bit 0 = non-zero
// (short(laplacian * 0.f) == 0: such a pixel cannot reach the panorama), bit 1 = exactly the weight of a fully owned
// pixel (the integer band path applies).  A quarter of the float pyramid's bytes crosses to the host.
__global__ __launch_bounds__(256) void k_weight_class(const float* __restrict__ w, uint8_t* __restrict__ out, size_t n, float unit)
{
    const size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = w[i];
    out[i] = static_cast<uint8_t>((v != 0.f ? 1 : 0) | (v == unit ? 2 : 0));
}

// FeatherBlender: dst_weight_map_ accumulation in feed order   blenders.cpp:143
__global__ __launch_bounds__(256) void k_feather_weight_sum(const CamDesc* __restrict__ cams, int n_cams,
                                                            float* __restrict__ ws, int cw, int ch)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= cw || y >= ch) return;
    float acc = 0.f;
    for (int i = 0; i < n_cams; ++i) {
        const CamDesc& c = cams[i];
        const int lx = x - c.corner_x, ly = y - c.corner_y;
        if (static_cast<unsigned>(lx) < static_cast<unsigned>(c.roi_w) &&
            static_cast<unsigned>(ly) < static_cast<unsigned>(c.roi_h))
            acc += gp(c.feather_w)[static_cast<size_t>(ly) * c.roi_w + lx];
    }
    ws[static_cast<size_t>(y) * cw + x] = acc;
}

inline dim3 grid2d(int w, int h) { return dim3((w + kTileW - 1) / kTileW, (h + kTileH - 1) / kTileH); }
inline dim3 block2d() { return dim3(kTileW, kTileH); }

}  // namespace

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------

void launch_build_maps(const CamDesc* d_cam, float* d_xmap, float* d_ymap, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_build_maps, grid2d(w, h), block2d(), 0, s, d_cam, d_xmap, d_ymap, w, h);
}

void launch_warp_u8(const CamDesc* d_cam, const uint8_t* d_src, size_t src_stride, int channels, int interp,
                    int border, uint8_t* d_dst, size_t dst_stride, int w, int h, hipStream_t s)
{
    const bool lin = interp == 1, refl = border == 2;
#define MST_LAUNCH_WARP(CN, L, R)                                                                         \
    hipLaunchKernelGGL((k_warp_u8<CN, L, R>), grid2d(w, h), block2d(), 0, s, d_cam, d_src, src_stride, d_dst, \
                       dst_stride, w, h)
    if (channels == 3) {
        if (lin && refl) MST_LAUNCH_WARP(3, true, true);
        else if (lin) MST_LAUNCH_WARP(3, true, false);
        else if (refl) MST_LAUNCH_WARP(3, false, true);
        else MST_LAUNCH_WARP(3, false, false);
    } else {
        if (lin && refl) MST_LAUNCH_WARP(1, true, true);
        else if (lin) MST_LAUNCH_WARP(1, true, false);
        else if (refl) MST_LAUNCH_WARP(1, false, true);
        else MST_LAUNCH_WARP(1, false, false);
    }
#undef MST_LAUNCH_WARP
}

// ---- NV12 ingest (mi355_stitch.h: MST_FLAG_FRAMES_NV12) ---------------------------------------------------------
// cv::cvtColor(COLOR_YUV2BGR_NV12) of OpenCV 2.4.11 (color.cpp, YUV420sp2RGB888Invoker<0, 0>): BT.601 limited range,
// 20-bit fixed point.  A thread converts 4 x 2 pixels (two chroma pairs): one dword of each luma row, one dword of
// chroma in, 12 bytes per row out.  HBM-bound (1.5 bytes in, 3 out per pixel).
namespace {
__device__ __forceinline__ uint32_t nv12_px(int y, int ruv, int guv, int buv)
{
    const int yy = max(0, y - 16) * 1220542;
    const int r = min(max((yy + ruv) >> 20, 0), 255), g = min(max((yy + guv) >> 20, 0), 255), b = min(max((yy + buv) >> 20, 0), 255);
    return static_cast<uint32_t>(b) | static_cast<uint32_t>(g) << 8 | static_cast<uint32_t>(r) << 16;
}
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
struct __attribute__((packed, aligned(1))) Px4 { uint32_t x, y, z; };

__global__ __launch_bounds__(256) void k_nv12_to_bgr(FrameBatch src, FrameBatch dst, const int2* __restrict__ sizes, int n_cams, int gx, int gy)
{
    const int per = gx * gy;
    const int which = blockIdx.x / per, rem = blockIdx.x - which * per;
    const int set = which / n_cams, cam = which - set * n_cams;
    const int w = sizes[cam].x, h = sizes[cam].y;
    const uint8_t* __restrict__ yuv = src.f[set].ptr[cam];
    uint8_t* __restrict__ out = const_cast<uint8_t*>(dst.f[set].ptr[cam]);
    if (!yuv || !out) return;
    const uint32_t ss = src.f[set].stride[cam], ds = dst.f[set].stride[cam];
    const int x = ((rem % gx) * 64 + (threadIdx.x & 63)) * 4, y = ((rem / gx) * 4 + (threadIdx.x >> 6)) * 2;
    if (x >= w || y >= h) return;
    const uint8_t* y1 = yuv + static_cast<size_t>(y) * ss + x;
    const uint8_t* uv = yuv + static_cast<size_t>(h) * ss + static_cast<size_t>(y >> 1) * ss + x;
    uint32_t l1, l2, c;
    if (x + 4 <= w) {
        l1 = *reinterpret_cast<const u32_unaligned*>(y1);
        l2 = *reinterpret_cast<const u32_unaligned*>(y1 + ss);
        c = *reinterpret_cast<const u32_unaligned*>(uv);
    } else {                      // (w == 2 mod 4: the last thread of a row has one chroma pair)
        l1 = y1[0] | y1[1] << 8; l2 = y1[ss] | y1[ss + 1] << 8; c = uv[0] | uv[1] << 8;
    }
    uint32_t p[2][4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = static_cast<int>((c >> (16 * k)) & 0xFFu) - 128, v = static_cast<int>((c >> (16 * k + 8)) & 0xFFu) - 128;
        const int ruv = (1 << 19) + 1673527 * v, guv = (1 << 19) - 852492 * v - 409993 * u, buv = (1 << 19) + 2116026 * u;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            p[0][2 * k + q] = nv12_px(static_cast<int>((l1 >> (16 * k + 8 * q)) & 0xFFu), ruv, guv, buv);
            p[1][2 * k + q] = nv12_px(static_cast<int>((l2 >> (16 * k + 8 * q)) & 0xFFu), ruv, guv, buv);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        uint8_t* o = out + static_cast<size_t>(y + r) * ds + static_cast<size_t>(x) * 3;
        if (x + 4 <= w) {
            *reinterpret_cast<Px4*>(o) = Px4{p[r][0] | p[r][1] << 24, p[r][1] >> 8 | p[r][2] << 16, p[r][2] >> 16 | p[r][3] << 8};
        } else {
            for (int k = 0; k < 2; ++k) { o[3 * k] = static_cast<uint8_t>(p[r][k]); o[3 * k + 1] = static_cast<uint8_t>(p[r][k] >> 8); o[3 * k + 2] = static_cast<uint8_t>(p[r][k] >> 16); }
        }
    }
}
}  // namespace

void launch_nv12_to_bgr(const FrameBatch& src, const FrameBatch& dst, const int2* d_sizes, int n_cams, int max_w, int max_h, int batch, hipStream_t s)
{
    if (n_cams <= 0 || batch <= 0 || max_w <= 0 || max_h <= 0) return;
    const int gx = (max_w + 255) / 256, gy = (max_h + 7) / 8;         // a block: 64 x 4 threads of 4 x 2 pixels
    hipLaunchKernelGGL(k_nv12_to_bgr, dim3(static_cast<unsigned>(gx * gy * n_cams * batch)), dim3(256), 0, s, src, dst, d_sizes, n_cams, gx, gy);
}

void launch_remap_fixed(const UndistortCam* d_ucams, int n_cams, int max_w, int max_h, const FrameBatch& src,
                        const FrameBatch& dst, int batch, hipStream_t s)
{
    if (n_cams <= 0 || batch <= 0 || max_w <= 0 || max_h <= 0) return;
    const dim3 g2 = grid2d((max_w + kRemapPx - 1) / kRemapPx, max_h);
    const int bx = static_cast<int>(g2.x), by = static_cast<int>(g2.y);
    hipLaunchKernelGGL(k_remap_fixed, dim3(static_cast<unsigned>(bx * by * n_cams * batch)), block2d(), 0, s, d_ucams, n_cams, src,
                       dst, bx, by, batch);
}

void launch_exposure_apply(const CamDesc* d_cam, uint8_t* d_img, size_t stride, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_exposure_apply, grid2d(w, h), block2d(), 0, s, d_cam, d_img, stride, w, h);
}

void launch_weight_level0(const CamDesc* d_cams, int cam, float* d_w0, int bw, int bh, bool weights_16s, hipStream_t s)
{
    const float alpha = weights_16s ? 0.f : static_cast<float>(1. / 255.);
    hipLaunchKernelGGL(k_weight_level0, grid2d(bw, bh), block2d(), 0, s, d_cams, cam, d_w0, bw, bh, alpha);
}

void launch_border_mask(const CamDesc* d_cams, int cam, uint8_t* d_out, int bw, int bh, hipStream_t s)
{
    hipLaunchKernelGGL(k_border_mask, grid2d(bw, bh), block2d(), 0, s, d_cams, cam, d_out, bw, bh);
}

void launch_pyrdown_f32(const float* d_src, int w, int h, float* d_dst, bool weights_16s, hipStream_t s)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    if (weights_16s) hipLaunchKernelGGL(k_pyrdown_s16, grid2d(dw, dh), block2d(), 0, s, d_src, w, h, d_dst, dw, dh);
    else hipLaunchKernelGGL(k_pyrdown_f32, grid2d(dw, dh), block2d(), 0, s, d_src, w, h, d_dst, dw, dh);
}

void launch_mask_fixup(const CamDesc* d_cam, const uint8_t* d_seam, int sw, int sh, const int32_t* xo0, const int32_t* xo1,
                       const int16_t* xa0, const int16_t* xa1, const int32_t* yo0, const int32_t* yo1, const int16_t* yb0,
                       const int16_t* yb1, uint8_t* d_out, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_mask_fixup, grid2d(w, h), block2d(), 0, s, d_cam, d_seam, sw, sh, xo0, xo1, xa0, xa1, yo0, yo1,
                       yb0, yb1, d_out, w, h);
}

void launch_feather_weight_map(const uint8_t* d_mask, int w, int h, float sharpness, int32_t* d_tmp, float* d_weight,
                               hipStream_t s)
{
    hipLaunchKernelGGL(k_dt_columns, dim3((w + 255) / 256), dim3(256), 0, s, d_mask, w, h, d_tmp);
    hipLaunchKernelGGL(k_dt_rows<false>, dim3((h + 63) / 64), dim3(64), 0, s, d_tmp, w, h, sharpness, d_weight);
}

void launch_voronoi_pair(uint8_t* d_m1, int w1, int h1, int ox1, int oy1, uint8_t* d_m2, int w2, int h2, int ox2, int oy2,
                         int rw, int rh, uint8_t* d_inv1, uint8_t* d_inv2, int32_t* d_tmp, float* d_dist1, float* d_dist2,
                         hipStream_t s)
{
    VoronoiPair p{d_m1, d_m2, w1, h1, w2, h2, ox1, oy1, ox2, oy2, rw, rh, 10};
    const int sw = rw + 2 * p.gap, sh = rh + 2 * p.gap;
    hipLaunchKernelGGL(k_voronoi_cut, grid2d(sw, sh), block2d(), 0, s, p, d_inv1, d_inv2);
    hipLaunchKernelGGL(k_dt_columns, dim3((sw + 255) / 256), dim3(256), 0, s, d_inv1, sw, sh, d_tmp);
    hipLaunchKernelGGL(k_dt_rows<true>, dim3((sh + 63) / 64), dim3(64), 0, s, d_tmp, sw, sh, 0.f, d_dist1);
    hipLaunchKernelGGL(k_dt_columns, dim3((sw + 255) / 256), dim3(256), 0, s, d_inv2, sw, sh, d_tmp);
    hipLaunchKernelGGL(k_dt_rows<true>, dim3((sh + 63) / 64), dim3(64), 0, s, d_tmp, sw, sh, 0.f, d_dist2);
    hipLaunchKernelGGL(k_voronoi_apply, grid2d(rw, rh), block2d(), 0, s, p, d_dist1, d_dist2);
}

void launch_overlap_stats(const FeedImage* d_imgs, const FeedPair* d_pairs, int n_pairs, FeedStat* d_out, hipStream_t s)
{
    if (n_pairs <= 0) return;
    hipLaunchKernelGGL(k_overlap_stats, dim3((n_pairs + 63) / 64), dim3(64), 0, s, d_imgs, d_pairs, n_pairs, d_out);
}

void launch_payload_copy(const CamDesc* d_cam, const PayloadRect* d_rects, int n_rects, uint32_t* d_buf, bool pack, hipStream_t s)
{
    if (n_rects <= 0) return;
    if (pack) hipLaunchKernelGGL(k_payload_copy<true>, dim3(n_rects), dim3(256), 0, s, d_cam, d_rects, d_buf);
    else hipLaunchKernelGGL(k_payload_copy<false>, dim3(n_rects), dim3(256), 0, s, d_cam, d_rects, d_buf);
}

void launch_weight_class(const float* d_w, uint8_t* d_out, size_t n, float unit, hipStream_t s)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_weight_class, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, s, d_w, d_out, n, unit);
}

void launch_band_weight_sum(const CamDesc* d_cams, int n_cams, int level, float* d_ws, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_band_weight_sum, grid2d(cw, ch), block2d(), 0, s, d_cams, n_cams, level, d_ws, cw, ch);
}

void launch_feather_weight_sum(const CamDesc* d_cams, int n_cams, float* d_ws, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_feather_weight_sum, grid2d(cw, ch), block2d(), 0, s, d_cams, n_cams, d_ws, cw, ch);
}

}  // namespace mst
