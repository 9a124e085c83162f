// kernels.hip -- hand-written HIP kernels for gfx950 (CDNA4, wave64).
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
//   k_warp_border_s16     warp + apply + convertTo(16S) + copyMakeBorder(REFLECT)
//                                                           R64/src/stitcher.cpp:275-285, blenders.cpp:272-274
//   k_pyrdown_s16         cv::pyrDown CV_16S                R64/src/blenders.cpp:481
//   k_pyrdown_f32         cv::pyrDown CV_32F (weights)      R64/src/blenders.cpp:298
//   k_band_blend/_final   createLaplacePyr pyrUp+subtract   R64/src/blenders.cpp:482-487
//                         weighted accumulate               R64/src/blenders.cpp:306-330
//                         normalizeUsingWeightMap           R64/src/blenders.cpp:383-405
//                         restoreImageFromLaplacePyr        R64/src/blenders.cpp:520-530
//                         crop, mask, Blender::blend, to 8U R64/src/blenders.cpp:369-376,105-112; stitcher.cpp:313
//   k_feather_compose     FeatherBlender::feed/blend        R64/src/blenders.cpp:123-155
//   k_paste_compose       Blender::feed/blend               R64/src/blenders.cpp:81-112
//                         feedSize/feedSizeRemap (LUT)      DRV:117-177
//
// Floating point: compiled with -ffp-contract=off and IEEE-correct division; expressions keep
// the reference's left-to-right order.  The remaining float work (map, gain, weights) is
// therefore bit-identical to an SSE2 scalar evaluation.
#include "kernels.hpp"

#include <climits>

#pragma clang fp contract(off)

namespace mst {

namespace {

constexpr float kWeightEps = 1e-5f;   // WEIGHT_EPS, blenders.cpp:50

// ------------------------------------------------------------------------------------------
// scalar helpers
// ------------------------------------------------------------------------------------------

// cvRound on x64 (cvtsd2si / cvtps2dq): round-half-even, 0x80000000 when unrepresentable
__device__ __forceinline__ int cv_round(float v)
{
    float r = rintf(v);
    return (r >= -2147483648.f && r < 2147483648.f) ? static_cast<int>(r) : INT_MIN;
}

__device__ __forceinline__ int sat_s16(int v) { return min(max(v, -32768), 32767); }
__device__ __forceinline__ int sat_u8(int v) { return min(max(v, 0), 255); }

// static_cast<short>(float) as the x64 build performs it: cvttss2si, keep the low 16 bits
__device__ __forceinline__ int trunc_s16(float v)
{
    int t = (v > -2147483904.f && v < 2147483648.f) ? static_cast<int>(v) : INT_MIN;
    return static_cast<int>(static_cast<short>(t));
}

__device__ __forceinline__ int wrap_s16(int v) { return static_cast<int>(static_cast<short>(v)); }

template <int DELTA>   // DELTA 0 = BORDER_REFLECT, 1 = BORDER_REFLECT_101 (cv::borderInterpolate)
__device__ __forceinline__ int reflect(int p, int len)
{
    if (static_cast<unsigned>(p) < static_cast<unsigned>(len)) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p - 1 + DELTA;
        else p = len - 1 - (p - len) - DELTA;
    } while (static_cast<unsigned>(p) >= static_cast<unsigned>(len));
    return p;
}

// ------------------------------------------------------------------------------------------
// backward map, sampler, gain
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ void backward_map(const CamDesc& c, int x, int y, float& mx, float& my)
{
    const float ca = c.col_a[x];
    const float ra = c.row_a[y];
    float X, Y, Z;
    if (c.kind == 0) {
        const float omt = c.one_minus_t2;
        X = c.kr[0] * ca + c.kr[1] * ra + c.kr[2] * omt;
        Y = c.kr[3] * ca + c.kr[4] * ra + c.kr[5] * omt;
        Z = c.kr[6] * ca + c.kr[7] * ra + c.kr[8] * omt;
        mx = X / Z;
        my = Y / Z;
        return;
    }
    float x_, y_, z_;
    const float cb = c.col_b[x];
    if (c.kind == 1) {
        x_ = ra * ca;
        y_ = c.row_b[y];
        z_ = ra * cb;
    } else {
        x_ = ca;
        y_ = ra;
        z_ = cb;
    }
    X = c.kr[0] * x_ + c.kr[1] * y_ + c.kr[2] * z_;
    Y = c.kr[3] * x_ + c.kr[4] * y_ + c.kr[5] * z_;
    Z = c.kr[6] * x_ + c.kr[7] * y_ + c.kr[8] * z_;
    if (Z > 0) { mx = X / Z; my = Y / Z; }
    else { mx = -1.f; my = -1.f; }
}

struct Taps {
    int x0, x1, y0, y1;     // tap coordinates (already border-resolved, or -1 for constant border)
    int w00, w01, w10, w11; // 15-bit weights, sum 32768
};

// fixed-point coordinates of cv::remap: cvRound(map*32), integer part saturated to short
template <bool REFLECT_BORDER>
__device__ __forceinline__ Taps linear_taps(float mx, float my, int sw, int sh)
{
    const int fsx = cv_round(mx * 32.f), fsy = cv_round(my * 32.f);
    const int fx = fsx & 31, fy = fsy & 31;
    const int sx = sat_s16(fsx >> 5), sy = sat_s16(fsy >> 5);
    Taps t;
    // BilinearTab_i: (1-fy)(1-fx), (1-fy)fx, fy(1-fx), fy*fx scaled by 2^15.  The table's entry
    // for fx=fy=0 is {32767,0,0,1} after OpenCV's sum fix-up; on 8-bit data that entry and the
    // exact {32768,0,0,0} produce the same byte for every input, so the closed
    // form is used.
    t.w00 = (32 - fx) * (32 - fy) * 32;
    t.w01 = fx * (32 - fy) * 32;
    t.w10 = (32 - fx) * fy * 32;
    t.w11 = fx * fy * 32;
    if (REFLECT_BORDER) {
        t.x0 = reflect<0>(sx, sw);     t.x1 = reflect<0>(sx + 1, sw);
        t.y0 = reflect<0>(sy, sh);     t.y1 = reflect<0>(sy + 1, sh);
    } else {
        t.x0 = static_cast<unsigned>(sx) < static_cast<unsigned>(sw) ? sx : -1;
        t.x1 = static_cast<unsigned>(sx + 1) < static_cast<unsigned>(sw) ? sx + 1 : -1;
        t.y0 = static_cast<unsigned>(sy) < static_cast<unsigned>(sh) ? sy : -1;
        t.y1 = static_cast<unsigned>(sy + 1) < static_cast<unsigned>(sh) ? sy + 1 : -1;
    }
    return t;
}

__device__ __forceinline__ int tap_u8(const uint8_t* __restrict__ src, size_t stride, int cn, int x, int y, int k)
{
    return (x < 0 || y < 0) ? 0 : static_cast<int>(src[static_cast<size_t>(y) * stride + x * cn + k]);
}

__device__ __forceinline__ int blend_taps(const Taps& t, int v00, int v01, int v10, int v11)
{
    return sat_u8((v00 * t.w00 + v01 * t.w01 + v10 * t.w10 + v11 * t.w11 + (1 << 14)) >> 15);
}

// INTER_LINEAR + BORDER_REFLECT on CV_8UC3: the image path of RotationWarper::warp
__device__ __forceinline__ void sample_linear_reflect_c3(const uint8_t* __restrict__ src, size_t stride,
                                                         int sw, int sh, float mx, float my, int out[3])
{
    const Taps t = linear_taps<true>(mx, my, sw, sh);
    const uint8_t* r0 = src + static_cast<size_t>(t.y0) * stride;
    const uint8_t* r1 = src + static_cast<size_t>(t.y1) * stride;
    const uint8_t* p00 = r0 + t.x0 * 3; const uint8_t* p01 = r0 + t.x1 * 3;
    const uint8_t* p10 = r1 + t.x0 * 3; const uint8_t* p11 = r1 + t.x1 * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        out[k] = blend_taps(t, p00[k], p01[k], p10[k], p11[k]);
}

__device__ __forceinline__ float gain_at(const CamDesc& c, int x, int y)
{
    if (c.expos == 2) {
        const int x0 = c.gx0[x], x1 = c.gx1[x];
        const float a0 = c.gxw0[x], a1 = c.gxw1[x];
        const float* g0 = c.gain_map + c.gy0[y] * c.gw;
        const float* g1 = c.gain_map + c.gy1[y] * c.gw;
        const float b0 = c.gyw0[y], b1 = c.gyw1[y];
        const float r0 = g0[x0] * a0 + g0[x1] * a1;    // HResizeLinear
        const float r1 = g1[x0] * a0 + g1[x1] * a1;
        return r0 * b0 + r1 * b1;                      // VResizeLinear
    }
    return c.gain;
}

// saturate_cast<uchar>(uchar * float)
__device__ __forceinline__ int apply_gain(int v, float g) { return sat_u8(cv_round(static_cast<float>(v) * g)); }

// warped + exposure-compensated pixel of camera c at roi position (x, y), as CV_16S values
__device__ __forceinline__ void warped_pixel(const CamDesc& c, const uint8_t* __restrict__ src, size_t stride,
                                             int x, int y, int out[3])
{
    float mx, my;
    backward_map(c, x, y, mx, my);
    sample_linear_reflect_c3(src, stride, c.src_w, c.src_h, mx, my, out);
    if (c.expos != 0) {
        const float g = gain_at(c, x, y);
#pragma unroll
        for (int k = 0; k < 3; ++k) out[k] = apply_gain(out[k], g);
    }
}

// one output sample of cv::pyrUp (CV_16S, dst = 2*src) at (x, y) of the fine grid
__device__ __forceinline__ int pyrup_at(const int16_t* __restrict__ s, int sw, int sh, int x, int y)
{
    const int j = x >> 1, i = y >> 1;
    const int jm = j == 0 ? min(1, sw - 1) : j - 1, jp = min(j + 1, sw - 1);
    const int im = i == 0 ? min(1, sh - 1) : i - 1, ip = min(i + 1, sh - 1);
    const int16_t* r0 = s + static_cast<size_t>(im) * sw;
    const int16_t* r1 = s + static_cast<size_t>(i) * sw;
    const int16_t* r2 = s + static_cast<size_t>(ip) * sw;
    int h0, h1, h2;
    if (x & 1) {
        h0 = (r0[j] + r0[jp]) * 4;
        h1 = (r1[j] + r1[jp]) * 4;
        h2 = (r2[j] + r2[jp]) * 4;
    } else {
        h0 = r0[jm] + r0[j] * 6 + r0[jp];
        h1 = r1[jm] + r1[j] * 6 + r1[jp];
        h2 = r2[jm] + r2[j] * 6 + r2[jp];
    }
    const int v = (y & 1) ? (h1 + h2) * 4 : h0 + h1 * 6 + h2;
    return sat_s16((v + 32) >> 6);
}

__device__ __forceinline__ int find_segment(const SegTable& seg, int n, int b)
{
    int c = 0;
    while (c + 1 < n && b >= seg.start[c + 1]) ++c;
    return c;
}

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
        w = static_cast<float>(c.mask[static_cast<size_t>(ry) * c.roi_w + rx]) * alpha + 0.f;
    w0[static_cast<size_t>(y) * bw + x] = w;
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
            acc += c.W[level][static_cast<size_t>(ly) * c.lw[level] + lx];
    }
    ws[static_cast<size_t>(y) * cw + x] = acc;
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
            acc += c.feather_w[static_cast<size_t>(ly) * c.roi_w + lx];
    }
    ws[static_cast<size_t>(y) * cw + x] = acc;
}

// ------------------------------------------------------------------------------------------
// per-frame hot path: multi-band
// ------------------------------------------------------------------------------------------

// Level 0 of every camera's pyramid: warp + gain + widen to 16S, written straight into the
// BORDER_REFLECT-padded image (the border re-evaluates the mirrored roi pixel).
__global__ __launch_bounds__(256) void k_warp_border_s16(const CamDesc* __restrict__ cams, SegTable seg, int n_cams,
                                                         FrameArgs frames)
{
    const int cam = find_segment(seg, n_cams, blockIdx.x);
    const CamDesc& c = cams[cam];
    const int lb = blockIdx.x - seg.start[cam];
    const int bx = (lb % seg.tiles_x[cam]) * kTileW + threadIdx.x;
    const int by = (lb / seg.tiles_x[cam]) * kTileH + threadIdx.y;
    const int bw = c.lw[0], bh = c.lh[0];
    if (bx >= bw || by >= bh) return;
    const int rx = reflect<0>(bx - c.left, c.roi_w), ry = reflect<0>(by - c.top, c.roi_h);
    int v[3];
    warped_pixel(c, frames.ptr[cam], frames.stride[cam], rx, ry, v);
    const size_t plane = static_cast<size_t>(bw) * bh;
    int16_t* g = c.G[0] + static_cast<size_t>(by) * bw + bx;
    g[0] = static_cast<int16_t>(v[0]);
    g[plane] = static_cast<int16_t>(v[1]);
    g[2 * plane] = static_cast<int16_t>(v[2]);
}

// cv::pyrDown CV_16S, all three planes of every camera, level -> level+1
__global__ __launch_bounds__(256) void k_pyrdown_s16(const CamDesc* __restrict__ cams, SegTable seg, int n_cams,
                                                     int level)
{
    const int cam = find_segment(seg, n_cams, blockIdx.x);
    const CamDesc& c = cams[cam];
    const int lb = blockIdx.x - seg.start[cam];
    const int x = (lb % seg.tiles_x[cam]) * kTileW + threadIdx.x;
    const int y = (lb / seg.tiles_x[cam]) * kTileH + threadIdx.y;
    const int w = c.lw[level], h = c.lh[level], dw = c.lw[level + 1], dh = c.lh[level + 1];
    if (x >= dw || y >= dh) return;
    int xi[5], yi[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        xi[k] = reflect<1>(2 * x + k - 2, w);
        yi[k] = reflect<1>(2 * y + k - 2, h);
    }
    const size_t splane = static_cast<size_t>(w) * h, dplane = static_cast<size_t>(dw) * dh;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int16_t* s = c.G[level] + p * splane;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int16_t* r = s + static_cast<size_t>(yi[k]) * w;
            const int hsum = r[xi[2]] * 6 + (r[xi[1]] + r[xi[3]]) * 4 + r[xi[0]] + r[xi[4]];
            acc += hsum * (k == 2 ? 6 : ((k == 1 || k == 3) ? 4 : 1));
        }
        c.G[level + 1][p * dplane + static_cast<size_t>(y) * dw + x] = static_cast<int16_t>(sat_s16((acc + 128) >> 8));
    }
}

// One canvas pixel of one band: sum over cameras of short(laplacian * weight) (wrapping),
// normalise by the static weight sum, add the up-sampled coarser collapsed level.
__device__ __forceinline__ void band_pixel(const CamDesc* __restrict__ cams, const CanvasDesc& cv, int level,
                                           int x, int y, int out[3])
{
    int acc[3] = {0, 0, 0};
    const bool top_level = level == cv.nlev - 1;
    for (int i = 0; i < cv.n_cams; ++i) {
        const CamDesc& c = cams[i];
        const int lw = c.lw[level], lh = c.lh[level];
        const int lx = x - c.ox[level], ly = y - c.oy[level];
        if (static_cast<unsigned>(lx) >= static_cast<unsigned>(lw) ||
            static_cast<unsigned>(ly) >= static_cast<unsigned>(lh))
            continue;
        const float w = c.W[level][static_cast<size_t>(ly) * lw + lx];
        if (w == 0.f) continue;     // short(v * 0.f) == 0: contributes nothing
        const size_t plane = static_cast<size_t>(lw) * lh;
        const int16_t* g = c.G[level] + static_cast<size_t>(ly) * lw + lx;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            int lap = g[p * plane];
            if (!top_level) {
                const int sw = c.lw[level + 1], sh = c.lh[level + 1];
                const int up = pyrup_at(c.G[level + 1] + p * static_cast<size_t>(sw) * sh, sw, sh, lx, ly);
                lap = sat_s16(lap - up);                              // cv::subtract saturates
            }
            acc[p] = wrap_s16(acc[p] + trunc_s16(static_cast<float>(lap) * w));
        }
    }
    const int cw = cv.cw[level], ch = cv.ch[level];
    const float ws = cv.WS[level][static_cast<size_t>(y) * cw + x] + kWeightEps;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        int v = trunc_s16(static_cast<float>(acc[p]) / ws);
        if (!top_level) {
            const int sw = cv.cw[level + 1], sh = cv.ch[level + 1];
            const int up = pyrup_at(cv.C[level + 1] + p * static_cast<size_t>(sw) * sh, sw, sh, x, y);
            v = sat_s16(up + v);                                      // cv::add saturates
        }
        out[p] = v;
    }
}

__global__ __launch_bounds__(256) void k_band_blend(const CamDesc* __restrict__ cams, CanvasDesc cv, int level)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    const int cw = cv.cw[level], ch = cv.ch[level];
    if (x >= cw || y >= ch) return;
    int v[3];
    band_pixel(cams, cv, level, x, y, v);
    const size_t plane = static_cast<size_t>(cw) * ch;
    int16_t* o = cv.C[level] + static_cast<size_t>(y) * cw + x;
    o[0] = static_cast<int16_t>(v[0]);
    o[plane] = static_cast<int16_t>(v[1]);
    o[2 * plane] = static_cast<int16_t>(v[2]);
}

// Level 0 + crop to dst_roi_final_ + mask (w0 > eps) + zero outside + convertTo(8U)
__global__ __launch_bounds__(256) void k_band_final(const CamDesc* __restrict__ cams, CanvasDesc cv,
                                                    uint8_t* __restrict__ pano, size_t pano_stride,
                                                    uint8_t* __restrict__ pano_mask, size_t mask_stride)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= cv.final_w || y >= cv.final_h) return;
    int v[3];
    band_pixel(cams, cv, 0, x, y, v);
    const bool m = cv.WS[0][static_cast<size_t>(y) * cv.cw[0] + x] > kWeightEps;
    uint8_t* o = pano + static_cast<size_t>(y) * pano_stride + x * 3;
    o[0] = m ? static_cast<uint8_t>(sat_u8(v[0])) : 0;
    o[1] = m ? static_cast<uint8_t>(sat_u8(v[1])) : 0;
    o[2] = m ? static_cast<uint8_t>(sat_u8(v[2])) : 0;
    if (pano_mask) pano_mask[static_cast<size_t>(y) * mask_stride + x] = m ? 255 : 0;
}

// ------------------------------------------------------------------------------------------
// per-frame hot path: feather and paste -- one fused kernel each (source pixels in, panorama out)
// ------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_feather_compose(const CamDesc* __restrict__ cams, int n_cams,
                                                         FrameArgs frames, const float* __restrict__ ws_map,
                                                         int cw, int ch, uint8_t* __restrict__ pano,
                                                         size_t pano_stride, uint8_t* __restrict__ pano_mask,
                                                         size_t mask_stride)
{
    const int x = blockIdx.x * kTileW + threadIdx.x, y = blockIdx.y * kTileH + threadIdx.y;
    if (x >= cw || y >= ch) return;
    int acc[3] = {0, 0, 0};
    for (int i = 0; i < n_cams; ++i) {
        const CamDesc& c = cams[i];
        const int lx = x - c.corner_x, ly = y - c.corner_y;
        if (static_cast<unsigned>(lx) >= static_cast<unsigned>(c.roi_w) ||
            static_cast<unsigned>(ly) >= static_cast<unsigned>(c.roi_h))
            continue;
        const float w = c.feather_w[static_cast<size_t>(ly) * c.roi_w + lx];
        if (w == 0.f) continue;
        int v[3];
        warped_pixel(c, frames.ptr[i], frames.stride[i], lx, ly, v);
#pragma unroll
        for (int p = 0; p < 3; ++p) acc[p] = wrap_s16(acc[p] + trunc_s16(static_cast<float>(v[p]) * w));
    }
    const float wsum = ws_map[static_cast<size_t>(y) * cw + x];
    const bool m = wsum > kWeightEps;
    const float ws = wsum + kWeightEps;
    uint8_t* o = pano + static_cast<size_t>(y) * pano_stride + x * 3;
#pragma unroll
    for (int p = 0; p < 3; ++p) o[p] = m ? static_cast<uint8_t>(sat_u8(trunc_s16(static_cast<float>(acc[p]) / ws))) : 0;
    if (pano_mask) pano_mask[static_cast<size_t>(y) * mask_stride + x] = m ? 255 : 0;
}

__global__ __launch_bounds__(256) void k_paste_compose(const CamDesc* __restrict__ cams, int n_cams, FrameArgs frames,
                                                       int cw, int ch, int crop_x, int crop_y, int out_w, int out_h,
                                                       int driver_lut, uint8_t* __restrict__ pano, size_t pano_stride,
                                                       uint8_t* __restrict__ pano_mask, size_t mask_stride)
{
    const int ox = blockIdx.x * kTileW + threadIdx.x, oy = blockIdx.y * kTileH + threadIdx.y;
    if (ox >= out_w || oy >= out_h) return;
    const int x = ox + crop_x, y = oy + crop_y;
    int owner = -1, olx = 0, oly = 0, mask_or = 0;
    for (int i = 0; i < n_cams; ++i) {          // later cameras overwrite earlier ones (feed order)
        const CamDesc& c = cams[i];
        const int lx = x - c.corner_x, ly = y - c.corner_y;
        if (static_cast<unsigned>(lx) >= static_cast<unsigned>(c.roi_w) ||
            static_cast<unsigned>(ly) >= static_cast<unsigned>(c.roi_h))
            continue;
        const int mv = c.mask[static_cast<size_t>(ly) * c.roi_w + lx];
        mask_or |= mv;
        if (mv) { owner = i; olx = lx; oly = ly; }
    }
    int v[3] = {0, 0, 0};
    if (owner < 0 && driver_lut) { owner = 0; olx = 0; oly = 0; }   // zero-initialised LUT entry (DRV:107-112)
    if (owner >= 0) warped_pixel(cams[owner], frames.ptr[owner], frames.stride[owner], olx, oly, v);
    uint8_t* o = pano + static_cast<size_t>(oy) * pano_stride + ox * 3;
    o[0] = static_cast<uint8_t>(v[0]);
    o[1] = static_cast<uint8_t>(v[1]);
    o[2] = static_cast<uint8_t>(v[2]);
    if (pano_mask) pano_mask[static_cast<size_t>(oy) * mask_stride + ox] = static_cast<uint8_t>(mask_or);
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

void launch_exposure_apply(const CamDesc* d_cam, uint8_t* d_img, size_t stride, int w, int h, hipStream_t s)
{
    hipLaunchKernelGGL(k_exposure_apply, grid2d(w, h), block2d(), 0, s, d_cam, d_img, stride, w, h);
}

void launch_weight_level0(const CamDesc* d_cams, int cam, float* d_w0, int bw, int bh, hipStream_t s)
{
    const float alpha = static_cast<float>(1. / 255.);
    hipLaunchKernelGGL(k_weight_level0, grid2d(bw, bh), block2d(), 0, s, d_cams, cam, d_w0, bw, bh, alpha);
}

void launch_pyrdown_f32(const float* d_src, int w, int h, float* d_dst, hipStream_t s)
{
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    hipLaunchKernelGGL(k_pyrdown_f32, grid2d(dw, dh), block2d(), 0, s, d_src, w, h, d_dst, dw, dh);
}

void launch_band_weight_sum(const CamDesc* d_cams, int n_cams, int level, float* d_ws, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_band_weight_sum, grid2d(cw, ch), block2d(), 0, s, d_cams, n_cams, level, d_ws, cw, ch);
}

void launch_feather_weight_sum(const CamDesc* d_cams, int n_cams, float* d_ws, int cw, int ch, hipStream_t s)
{
    hipLaunchKernelGGL(k_feather_weight_sum, grid2d(cw, ch), block2d(), 0, s, d_cams, n_cams, d_ws, cw, ch);
}

void launch_warp_border_s16(const CamDesc* d_cams, const SegTable& seg, const FrameArgs& frames, hipStream_t s)
{
    int n = 0;
    while (n < kMaxCams && seg.start[n + 1] > seg.start[n]) ++n;
    hipLaunchKernelGGL(k_warp_border_s16, dim3(seg.start[n]), block2d(), 0, s, d_cams, seg, n, frames);
}

void launch_pyrdown_s16(const CamDesc* d_cams, const SegTable& seg, int level, hipStream_t s)
{
    int n = 0;
    while (n < kMaxCams && seg.start[n + 1] > seg.start[n]) ++n;
    hipLaunchKernelGGL(k_pyrdown_s16, dim3(seg.start[n]), block2d(), 0, s, d_cams, seg, n, level);
}

void launch_band_blend(const CamDesc* d_cams, const CanvasDesc& cv, int level, hipStream_t s)
{
    hipLaunchKernelGGL(k_band_blend, grid2d(cv.cw[level], cv.ch[level]), block2d(), 0, s, d_cams, cv, level);
}

void launch_band_final(const CamDesc* d_cams, const CanvasDesc& cv, uint8_t* d_pano, size_t pano_stride,
                       uint8_t* d_pano_mask, size_t mask_stride, hipStream_t s)
{
    hipLaunchKernelGGL(k_band_final, grid2d(cv.final_w, cv.final_h), block2d(), 0, s, d_cams, cv, d_pano,
                       pano_stride, d_pano_mask, mask_stride);
}

void launch_feather_compose(const CamDesc* d_cams, int n_cams, const FrameArgs& frames, const float* d_ws, int cw,
                            int ch, uint8_t* d_pano, size_t pano_stride, uint8_t* d_pano_mask, size_t mask_stride,
                            hipStream_t s)
{
    hipLaunchKernelGGL(k_feather_compose, grid2d(cw, ch), block2d(), 0, s, d_cams, n_cams, frames, d_ws, cw, ch,
                       d_pano, pano_stride, d_pano_mask, mask_stride);
}

void launch_paste_compose(const CamDesc* d_cams, int n_cams, const FrameArgs& frames, int cw, int ch, int crop_x,
                          int crop_y, int out_w, int out_h, int driver_lut, uint8_t* d_pano, size_t pano_stride,
                          uint8_t* d_pano_mask, size_t mask_stride, hipStream_t s)
{
    hipLaunchKernelGGL(k_paste_compose, grid2d(out_w, out_h), block2d(), 0, s, d_cams, n_cams, frames, cw, ch,
                       crop_x, crop_y, out_w, out_h, driver_lut, d_pano, pano_stride, d_pano_mask, mask_stride);
}

}  // namespace mst
