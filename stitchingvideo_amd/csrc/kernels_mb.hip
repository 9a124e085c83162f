// kernels_mb.hip -- the multi-band frame path as three fused HIP kernels for gfx950.
//
//   A  k_warp_pyr    per camera tile: backward map + fixed-point bilinear sample + exposure gain
//                    -> level 0 of the BORDER_REFLECT-padded image (kept in LDS, also stored packed)
//                    -> cv::pyrDown to level 1, straight from LDS.
//                    (RotationWarper::warp, ExposureCompensator::apply, convertTo(16S),
//                     copyMakeBorder, first pyrDown: stitcher.cpp:275-285, blenders.cpp:272-274,481)
//   B  k_pyr_chain   per camera tile: up to four further pyrDown levels inside LDS (blenders.cpp:481)
//   C  k_blend       per canvas tile, all bands coarse-to-fine in one block: Laplacian = G_l - pyrUp(G_l+1)
//                    (blenders.cpp:482-487), short(lap * w) accumulate over cameras (:306-330),
//                    normalise by the re-summed weights (:383-405), add pyrUp of the coarser collapsed
//                    level held in LDS (:520-530), crop + mask + convertTo(8U) (:369-376, stitcher.cpp:313).
//
// Data layout: every Gaussian level of a CV_8U-sourced pyramid stays within [0,255], so a pixel is
// one dword B | G<<8 | R<<16.  The [1 4 6 4 1] sums are evaluated two channels at a time in the 16-bit
// halves of a register (SWAR): 256*255 + 128 < 65536, so no carry crosses a half.  All of this is exact
// integer arithmetic, identical to the reference's int accumulators.
//
// Work lists: only tiles whose output can reach the panorama (non-zero blend weight somewhere down
// the chain) are launched; the lists are built once per calibration on the host.
#include "kernels.hpp"
#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace mst {

using namespace dev;

namespace {

// ---- packed-pixel helpers --------------------------------------------------------------------------
__device__ __forceinline__ uint32_t lanes_br(uint32_t p) { return p & 0x00FF00FFu; }
__device__ __forceinline__ uint32_t lane_g(uint32_t p) { return (p >> 8) & 0xFFu; }

// ======================================================================================================
// A: warp + gain -> level 0 (LDS + global) -> pyrDown -> level 1
// ======================================================================================================
constexpr int A_TW0 = 124, A_TH0 = 32;         // level-0 interior of a tile
constexpr int A_LW = A_TW0 + 4, A_LH = A_TH0 + 4;   // with the 2-pixel pyrDown halo: 128 x 36
constexpr int A_TW1 = A_TW0 / 2, A_TH1 = A_TH0 / 2; // level-1 outputs: 62 x 16
constexpr int A_ROWS_PER_THREAD = A_LH / 2;    // 256 threads = 128 columns x 2 row halves

// one warped, gain-compensated pixel, packed B | G<<8 | R<<16
template <int KIND>
__device__ __forceinline__ uint32_t warp_gain_packed(const CamDesc& c, const ColEntry& ce, const RowEntry& re,
                                                     const uint8_t* __restrict__ src, uint32_t stride)
{
    float X, Y, Z, mx, my;
    if (KIND == 1) {
        const float x_ = re.ra * ce.ca, z_ = re.ra * ce.cb;
        X = (c.kr[0] * x_ + re.b[0]) + c.kr[2] * z_;
        Y = (c.kr[3] * x_ + re.b[1]) + c.kr[5] * z_;
        Z = (c.kr[6] * x_ + re.b[2]) + c.kr[8] * z_;
    } else {
        X = (ce.a[0] + re.b[0]) + ce.c[0];
        Y = (ce.a[1] + re.b[1]) + ce.c[1];
        Z = (ce.a[2] + re.b[2]) + ce.c[2];
    }
    if (KIND == 0 || Z > 0) { mx = X / Z; my = Y / Z; }
    else { mx = -1.f; my = -1.f; }

    const int fsx = cv_round(mx * 32.f), fsy = cv_round(my * 32.f);
    const int fx = fsx & 31, fy = fsy & 31;
    const int sx = sat_s16(fsx >> 5), sy = sat_s16(fsy >> 5);
    const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32;
    const int w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    int b, g, r;
    if (static_cast<unsigned>(sx) < static_cast<unsigned>(c.src_w - 3) &&
        static_cast<unsigned>(sy) < static_cast<unsigned>(c.src_h - 1)) {
        // all four taps inside and 12 readable bytes from the aligned address below: fetch the two
        // 6-byte pixel pairs with aligned dwordx3 loads and a byte funnel shift
        const uint8_t* p0 = src + static_cast<uint32_t>(sy) * stride + static_cast<uint32_t>(sx) * 3u;
        const uint32_t sh = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p0)) & 3u;
        const uint32_t* q0 = reinterpret_cast<const uint32_t*>(p0 - sh);
        const uint8_t* p1 = p0 + stride;
        const uint32_t sh1 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(p1)) & 3u;
        const uint32_t* q1 = reinterpret_cast<const uint32_t*>(p1 - sh1);
        const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2];
        const uint32_t c0 = q1[0], c1 = q1[1], c2 = q1[2];
        const uint32_t lo0 = __builtin_amdgcn_alignbyte(a1, a0, sh), hi0 = __builtin_amdgcn_alignbyte(a2, a1, sh);
        const uint32_t lo1 = __builtin_amdgcn_alignbyte(c1, c0, sh1), hi1 = __builtin_amdgcn_alignbyte(c2, c1, sh1);
        b = ((lo0 & 0xFF) * w00 + (lo0 >> 24) * w01 + (lo1 & 0xFF) * w10 + (lo1 >> 24) * w11 + (1 << 14)) >> 15;
        g = (((lo0 >> 8) & 0xFF) * w00 + (hi0 & 0xFF) * w01 + ((lo1 >> 8) & 0xFF) * w10 + (hi1 & 0xFF) * w11 + (1 << 14)) >> 15;
        r = (((lo0 >> 16) & 0xFF) * w00 + ((hi0 >> 8) & 0xFF) * w01 + ((lo1 >> 16) & 0xFF) * w10 +
             ((hi1 >> 8) & 0xFF) * w11 + (1 << 14)) >> 15;
    } else {
        const int x0 = reflect<0>(sx, c.src_w), x1 = reflect<0>(sx + 1, c.src_w);
        const int y0 = reflect<0>(sy, c.src_h), y1 = reflect<0>(sy + 1, c.src_h);
        const uint8_t* r0 = src + static_cast<uint32_t>(y0) * stride;
        const uint8_t* r1 = src + static_cast<uint32_t>(y1) * stride;
        const uint8_t* p00 = r0 + x0 * 3; const uint8_t* p01 = r0 + x1 * 3;
        const uint8_t* p10 = r1 + x0 * 3; const uint8_t* p11 = r1 + x1 * 3;
        b = (p00[0] * w00 + p01[0] * w01 + p10[0] * w10 + p11[0] * w11 + (1 << 14)) >> 15;
        g = (p00[1] * w00 + p01[1] * w01 + p10[1] * w10 + p11[1] * w11 + (1 << 14)) >> 15;
        r = (p00[2] * w00 + p01[2] * w01 + p10[2] * w10 + p11[2] * w11 + (1 << 14)) >> 15;
    }
    if (c.expos != 0) {
        float gain = c.gain;
        if (c.expos == 2) {
            const float* g0 = c.gain_map + re.gy0;
            const float* g1 = c.gain_map + re.gy1;
            const float r0 = g0[ce.gx0] * ce.gxw0 + g0[ce.gx1] * ce.gxw1;
            const float r1 = g1[ce.gx0] * ce.gxw0 + g1[ce.gx1] * ce.gxw1;
            gain = r0 * re.gyw0 + r1 * re.gyw1;
        }
        b = apply_gain(b, gain);
        g = apply_gain(g, gain);
        r = apply_gain(r, gain);
    }
    return static_cast<uint32_t>(b) | (static_cast<uint32_t>(g) << 8) | (static_cast<uint32_t>(r) << 16);
}

template <int KIND>
__global__ __launch_bounds__(256) void k_warp_pyr(const CamDesc* __restrict__ cams, TileList list, FrameArgs frames)
{
    __shared__ uint32_t s_l0[A_LH][A_LW];       // level-0 tile with halo, packed pixels (18 KB)
    __shared__ uint2 s_h[A_LH][A_TW1];          // horizontal [1 4 6 4 1] sums: (B,R lanes | G)  (17.4 KB)
    __shared__ RowEntry s_rows[A_LH];

    const uint32_t t = list.tiles[blockIdx.x];
    const int cam = t >> 24, ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
    const CamDesc& c = cams[cam];
    const int bw = c.lw[0], bh = c.lh[0];
    const int lx0 = tx * A_TW0 - 2, ly0 = ty * A_TH0 - 2;    // level-0 origin of the halo'd tile
    const int tid = threadIdx.x;

    // per tile row: pyrDown's REFLECT_101 into the bordered image, then copyMakeBorder's REFLECT into the roi
    if (tid < A_LH) {
        const int by = reflect<1>(ly0 + tid, bh);
        const int ry = reflect<0>(by - c.top, c.roi_h);
        s_rows[tid] = c.rows[ry];
    }
    const int col = tid & (A_LW - 1);
    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);
    const int bx = reflect<1>(lx0 + col, bw);
    const int rx = reflect<0>(bx - c.left, c.roi_w);
    const ColEntry ce = c.cols[rx];
    const uint8_t* __restrict__ src = frames.ptr[cam];
    const uint32_t stride = frames.stride[cam];
    __syncthreads();

    const int gx = lx0 + col;
    const bool col_interior = col >= 2 && col < A_LW - 2 && gx < bw;
    for (int it = 0; it < A_ROWS_PER_THREAD; ++it) {
        const int row = half * A_ROWS_PER_THREAD + it;
        const RowEntry re = s_rows[row];
        const uint32_t px = warp_gain_packed<KIND>(c, ce, re, src, stride);
        s_l0[row][col] = px;
        const int gy = ly0 + row;
        if (col_interior && row >= 2 && row < A_LH - 2 && gy < bh)
            c.G[0][static_cast<uint32_t>(gy) * bw + gx] = px;
    }
    if (c.nlev < 2) return;
    __syncthreads();

    // horizontal pass
    for (int idx = tid; idx < A_LH * A_TW1; idx += 256) {
        const int row = idx / A_TW1, j = idx - row * A_TW1;
        const uint32_t* p = &s_l0[row][2 * j];
        const uint32_t p0 = p[0], p1 = p[1], p2 = p[2], p3 = p[3], p4 = p[4];
        const uint32_t hbr = lanes_br(p0) + lanes_br(p4) + 4u * (lanes_br(p1) + lanes_br(p3)) + 6u * lanes_br(p2);
        const uint32_t hg = lane_g(p0) + lane_g(p4) + 4u * (lane_g(p1) + lane_g(p3)) + 6u * lane_g(p2);
        s_h[row][j] = make_uint2(hbr, hg);
    }
    __syncthreads();
    // vertical pass + (v + 128) >> 8
    const int lw1 = c.lw[1], lh1 = c.lh[1];
    const int x1 = tx * A_TW1, y1 = ty * A_TH1;
    for (int idx = tid; idx < A_TH1 * A_TW1; idx += 256) {
        const int i = idx / A_TW1, j = idx - i * A_TW1;
        if (x1 + j >= lw1 || y1 + i >= lh1) continue;
        const uint2 h0 = s_h[2 * i][j], h1 = s_h[2 * i + 1][j], h2 = s_h[2 * i + 2][j], h3 = s_h[2 * i + 3][j],
                    h4 = s_h[2 * i + 4][j];
        const uint32_t vbr = h0.x + h4.x + 4u * (h1.x + h3.x) + 6u * h2.x;
        const uint32_t vg = h0.y + h4.y + 4u * (h1.y + h3.y) + 6u * h2.y;
        const uint32_t obr = ((vbr + 0x00800080u) >> 8) & 0x00FF00FFu;
        const uint32_t og = (vg + 128u) >> 8;
        c.G[1][static_cast<uint32_t>(y1 + i) * lw1 + x1 + j] = obr | (og << 8);
    }
}

// ======================================================================================================
// B: pyrDown chain level b -> b+1 .. b+k (k <= 4) inside LDS
// ======================================================================================================
constexpr int B_T = 64;                         // base-level interior of a tile
constexpr int B_E0 = 109, B_E1 = 53, B_E2 = 25, B_E3 = 11;   // extents for a full k = 4 chain (e4 = 4)

__device__ __forceinline__ uint32_t pyrdown_tap25(const uint32_t* __restrict__ s, int pitch, const int xi[5],
                                                  const int yi[5])
{
    uint32_t vbr = 0, vg = 0;
#pragma unroll
    for (int ky = 0; ky < 5; ++ky) {
        const uint32_t* r = s + yi[ky] * pitch;
        const uint32_t p0 = r[xi[0]], p1 = r[xi[1]], p2 = r[xi[2]], p3 = r[xi[3]], p4 = r[xi[4]];
        const uint32_t hbr = lanes_br(p0) + lanes_br(p4) + 4u * (lanes_br(p1) + lanes_br(p3)) + 6u * lanes_br(p2);
        const uint32_t hg = lane_g(p0) + lane_g(p4) + 4u * (lane_g(p1) + lane_g(p3)) + 6u * lane_g(p2);
        const uint32_t wy = (ky == 2) ? 6u : ((ky == 1 || ky == 3) ? 4u : 1u);
        vbr += wy * hbr;
        vg += wy * hg;
    }
    return (((vbr + 0x00800080u) >> 8) & 0x00FF00FFu) | (((vg + 128u) >> 8) << 8);
}

__global__ __launch_bounds__(256) void k_pyr_chain(const CamDesc* __restrict__ cams, TileList list, int base, int k)
{
    // regions of levels base .. base+3, back to back (47.5 + 11.2 + 2.5 + 0.5 KB)
    constexpr int kOff1 = B_E0 * B_E0, kOff2 = kOff1 + B_E1 * B_E1, kOff3 = kOff2 + B_E2 * B_E2;
    __shared__ uint32_t s_all[kOff3 + B_E3 * B_E3];
    uint32_t* const s_e0 = s_all;

    const uint32_t t = list.tiles[blockIdx.x];
    const int cam = t >> 24, ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
    const CamDesc& c = cams[cam];
    const int tid = threadIdx.x;

    // extents e[j] and origins lo[j] (in level base+j coordinates) of the regions each level must cover
    int e[5], lox[5], loy[5];
    e[k] = B_T >> k;
    lox[k] = (tx * B_T) >> k;
    loy[k] = (ty * B_T) >> k;
    for (int j = k - 1; j >= 0; --j) {
        e[j] = 2 * e[j + 1] + 3;
        lox[j] = 2 * lox[j + 1] - 2;
        loy[j] = 2 * loy[j + 1] - 2;
    }
    const int offs[4] = {0, kOff1, kOff2, kOff3};

    // load the base region (coordinates outside the image are never referenced: taps are reflected first)
    {
        const int w = c.lw[base], h = c.lh[base];
        const uint32_t* __restrict__ g = c.G[base];
        for (int idx = tid; idx < e[0] * e[0]; idx += 256) {
            const int yy = idx / e[0], xx = idx - yy * e[0];
            const int gx = lox[0] + xx, gy = loy[0] + yy;
            uint32_t v = 0;
            if (static_cast<unsigned>(gx) < static_cast<unsigned>(w) && static_cast<unsigned>(gy) < static_cast<unsigned>(h))
                v = g[static_cast<uint32_t>(gy) * w + gx];
            s_e0[idx] = v;
        }
    }
    __syncthreads();

    for (int j = 0; j < k; ++j) {
        const int sw = c.lw[base + j], sh = c.lh[base + j];          // source level size
        const int dw = c.lw[base + j + 1], dh = c.lh[base + j + 1];  // destination level size
        const uint32_t* __restrict__ s = s_all + offs[j];
        uint32_t* __restrict__ d = (j + 1 < 4) ? s_all + offs[j + 1] : nullptr;
        uint32_t* __restrict__ gout = c.G[base + j + 1];
        const int en = e[j + 1], es = e[j];
        // interior of this tile at the destination level
        const int ix0 = (tx * B_T) >> (j + 1), iy0 = (ty * B_T) >> (j + 1), isz = B_T >> (j + 1);
        for (int idx = tid; idx < en * en; idx += 256) {
            const int yy = idx / en, xx = idx - yy * en;
            const int ox = lox[j + 1] + xx, oy = loy[j + 1] + yy;    // destination coordinates
            if (static_cast<unsigned>(ox) >= static_cast<unsigned>(dw) ||
                static_cast<unsigned>(oy) >= static_cast<unsigned>(dh))
                continue;
            int xi[5], yi[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                xi[q] = reflect<1>(2 * ox + q - 2, sw) - lox[j];
                yi[q] = reflect<1>(2 * oy + q - 2, sh) - loy[j];
            }
            const uint32_t v = pyrdown_tap25(s, es, xi, yi);
            if (d && j + 1 < k) d[idx] = v;
            if (ox >= ix0 && ox < ix0 + isz && oy >= iy0 && oy < iy0 + isz)
                gout[static_cast<uint32_t>(oy) * dw + ox] = v;
        }
        __syncthreads();
    }
}

// ======================================================================================================
// C: all bands of one canvas tile, coarse to fine, one block
// ======================================================================================================
constexpr int C_TW = 128, C_TH = 64;

struct Region { int x0, y0, x1, y1; };

__host__ __device__ inline Region blend_region(const BlendArgs& a, int tx, int ty, int level)
{
    Region r;
    auto lo = [](int v, int m) { return v < m ? v : m; };
    auto hi = [](int v, int m) { return v > m ? v : m; };
    r.x0 = tx * C_TW; r.y0 = ty * C_TH;
    r.x1 = lo(r.x0 + C_TW, a.cw[0]); r.y1 = lo(r.y0 + C_TH, a.ch[0]);
    for (int l = 1; l <= level; ++l) {
        r.x0 = hi(0, ((r.x0 >> 1) - 1) & ~1);
        r.y0 = hi(0, ((r.y0 >> 1) - 1) & ~1);
        r.x1 = lo(a.cw[l], (((r.x1 - 1) >> 1) + 3) & ~1);
        r.y1 = lo(a.ch[l], (((r.y1 - 1) >> 1) + 3) & ~1);
    }
    return r;
}

// clamped 3-tap source indices of cv::pyrUp around coarse index j: s[-1] := s[1], s[n] := s[n-1]
__device__ __forceinline__ void up_idx(int j, int n, int& jm, int& jp)
{
    jm = j == 0 ? min(1, n - 1) : j - 1;
    jp = min(j + 1, n - 1);
}

// pyrUp of the packed (unsigned) Gaussian level around coarse pixel (j, i): four fine outputs as
// (B,R lanes) and G words, before the (v + 32) >> 6 rounding
__device__ __forceinline__ void pyrup_quad_packed(const uint32_t* __restrict__ g, int sw, int sh, int j, int i,
                                                  uint32_t obr[4], uint32_t og[4])
{
    int jm, jp, im, ip;
    up_idx(j, sw, jm, jp);
    up_idx(i, sh, im, ip);
    uint32_t hebr[3], hobr[3], heg[3], hog[3];
    const int rows[3] = {im, i, ip};
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t* p = g + static_cast<uint32_t>(rows[r]) * sw;
        const uint32_t a = p[jm], b = p[j], c = p[jp];
        hebr[r] = lanes_br(a) + 6u * lanes_br(b) + lanes_br(c);
        hobr[r] = 4u * (lanes_br(b) + lanes_br(c));
        heg[r] = lane_g(a) + 6u * lane_g(b) + lane_g(c);
        hog[r] = 4u * (lane_g(b) + lane_g(c));
    }
    // quad order: (x even,y even), (x odd,y even), (x even,y odd), (x odd,y odd)
    obr[0] = hebr[0] + 6u * hebr[1] + hebr[2];  og[0] = heg[0] + 6u * heg[1] + heg[2];
    obr[1] = hobr[0] + 6u * hobr[1] + hobr[2];  og[1] = hog[0] + 6u * hog[1] + hog[2];
    obr[2] = 4u * (hebr[1] + hebr[2]);          og[2] = 4u * (heg[1] + heg[2]);
    obr[3] = 4u * (hobr[1] + hobr[2]);          og[3] = 4u * (hog[1] + hog[2]);
}

__device__ __forceinline__ int s16_lo(uint32_t v) { return static_cast<int>(static_cast<short>(v & 0xFFFFu)); }
__device__ __forceinline__ int s16_hi(uint32_t v) { return static_cast<int>(v) >> 16; }
__device__ __forceinline__ uint2 pack3(int c0, int c1, int c2)
{
    return make_uint2((static_cast<uint32_t>(c0) & 0xFFFFu) | (static_cast<uint32_t>(c1) << 16),
                      static_cast<uint32_t>(c2) & 0xFFFFu);
}

// pyrUp of the collapsed coarser level held in LDS (signed shorts) around coarse pixel (j, i) given in
// level coordinates; rx0/ry0/pitch describe the LDS region.  out[q][ch], already (v + 32) >> 6 and saturated.
__device__ __forceinline__ void pyrup_quad_lds(const uint2* __restrict__ s, int rx0, int ry0, int pitch, int sw, int sh,
                                               int j, int i, int out[4][3])
{
    int jm, jp, im, ip;
    up_idx(j, sw, jm, jp);
    up_idx(i, sh, im, ip);
    const int rows[3] = {im - ry0, i - ry0, ip - ry0};
    const int c0 = jm - rx0, c1 = j - rx0, c2 = jp - rx0;
    int he[3][3], ho[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint2* p = s + rows[r] * pitch;
        const uint2 a = p[c0], b = p[c1], c = p[c2];
        const int av[3] = {s16_lo(a.x), s16_hi(a.x), s16_lo(a.y)};
        const int bv[3] = {s16_lo(b.x), s16_hi(b.x), s16_lo(b.y)};
        const int cv[3] = {s16_lo(c.x), s16_hi(c.x), s16_lo(c.y)};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            he[r][ch] = av[ch] + 6 * bv[ch] + cv[ch];
            ho[r][ch] = 4 * (bv[ch] + cv[ch]);
        }
    }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        out[0][ch] = sat_s16((he[0][ch] + 6 * he[1][ch] + he[2][ch] + 32) >> 6);
        out[1][ch] = sat_s16((ho[0][ch] + 6 * ho[1][ch] + ho[2][ch] + 32) >> 6);
        out[2][ch] = sat_s16((4 * (he[1][ch] + he[2][ch]) + 32) >> 6);
        out[3][ch] = sat_s16((4 * (ho[1][ch] + ho[2][ch]) + 32) >> 6);
    }
}

// One 2x2 quad of canvas level `l` at even coordinates (x, y).  TOP: coarsest level (no pyrUp anywhere);
// FINAL: level 0 (weights come from the masks, result goes to the panorama).
template <bool TOP, bool FINAL>
__device__ __forceinline__ void blend_quad(const CamDesc* __restrict__ cams, const BlendArgs& a, int l, int x, int y,
                                           const bool valid[4], uint32_t cam_bits, const uint2* __restrict__ s_up,
                                           const Region& up_r, int out[4][3], float ws[4])
{
    int acc[4][3];
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc[q][0] = acc[q][1] = acc[q][2] = 0; ws[q] = 0.f; }

    while (cam_bits) {
        const int ci = __builtin_ctz(cam_bits);
        cam_bits &= cam_bits - 1;
        const CamDesc& c = cams[ci];
        const int lw = c.lw[l], lh = c.lh[l];
        const int lx = x - c.ox[l], ly = y - c.oy[l];
        float w[4];
        bool in[4];
        if (TOP) {
            // windows may start on odd coordinates at the coarsest level: test pixel by pixel
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = lx + (q & 1), py = ly + (q >> 1);
                in[q] = valid[q] && static_cast<unsigned>(px) < static_cast<unsigned>(lw) &&
                        static_cast<unsigned>(py) < static_cast<unsigned>(lh);
                w[q] = in[q] ? (FINAL ? 0.f : c.W[l][static_cast<uint32_t>(py) * lw + px]) : 0.f;
            }
        } else {
            // below the top level every window origin and size is even: the quad is inside or outside as a whole
            const bool inside = static_cast<unsigned>(lx) < static_cast<unsigned>(lw) &&
                                static_cast<unsigned>(ly) < static_cast<unsigned>(lh);
            if (!inside) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) in[q] = valid[q];
            if (!FINAL) {
                const float2 w0 = *reinterpret_cast<const float2*>(c.W[l] + static_cast<uint32_t>(ly) * lw + lx);
                const float2 w1 = *reinterpret_cast<const float2*>(c.W[l] + static_cast<uint32_t>(ly + 1) * lw + lx);
                w[0] = w0.x; w[1] = w0.y; w[2] = w1.x; w[3] = w1.y;
            }
        }
        if (FINAL) {
            // level-0 weight: mask.convertTo(32F, 1/255) inside the roi, BORDER_CONSTANT 0 around it
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rx = lx + (q & 1) - c.left, ry = ly + (q >> 1) - c.top;
                const bool inr = in[q] && static_cast<unsigned>(rx) < static_cast<unsigned>(c.roi_w) &&
                                 static_cast<unsigned>(ry) < static_cast<unsigned>(c.roi_h);
                w[q] = inr ? static_cast<float>(c.mask[static_cast<uint32_t>(ry) * c.roi_w + rx]) * a.alpha + 0.f : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ws[q] += w[q];              // dst_weight_row[x] += weight_row[x_]
        if (w[0] == 0.f && w[1] == 0.f && w[2] == 0.f && w[3] == 0.f) continue;   // short(v * 0.f) == 0

        uint32_t g[4];
        if (TOP) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                g[q] = in[q] ? c.G[l][static_cast<uint32_t>(ly + (q >> 1)) * lw + lx + (q & 1)] : 0u;
        } else {
            const uint2 g0 = *reinterpret_cast<const uint2*>(c.G[l] + static_cast<uint32_t>(ly) * lw + lx);
            const uint2 g1 = *reinterpret_cast<const uint2*>(c.G[l] + static_cast<uint32_t>(ly + 1) * lw + lx);
            g[0] = g0.x; g[1] = g0.y; g[2] = g1.x; g[3] = g1.y;
        }
        uint32_t ubr[4] = {0, 0, 0, 0}, ug[4] = {0, 0, 0, 0};
        if (!TOP) {
            pyrup_quad_packed(c.G[l + 1], c.lw[l + 1], c.lh[l + 1], lx >> 1, ly >> 1, ubr, ug);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ubr[q] = ((ubr[q] + 0x00200020u) >> 6) & 0x03FF03FFu;
                ug[q] = (ug[q] + 32u) >> 6;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // laplacian = cv::subtract(G_l, pyrUp(G_l+1)) (|.| <= 255, the saturation cannot trigger);
            // dst += static_cast<short>(src * w): |src * w| < 256, so plain truncation is the x64 conversion
            const int lb = static_cast<int>(g[q] & 0xFF) - static_cast<int>(ubr[q] & 0xFFFF);
            const int lg = static_cast<int>((g[q] >> 8) & 0xFF) - static_cast<int>(ug[q]);
            const int lr = static_cast<int>((g[q] >> 16) & 0xFF) - static_cast<int>(ubr[q] >> 16);
            acc[q][0] = wrap_s16(acc[q][0] + static_cast<int>(static_cast<float>(lb) * w[q]));
            acc[q][1] = wrap_s16(acc[q][1] + static_cast<int>(static_cast<float>(lg) * w[q]));
            acc[q][2] = wrap_s16(acc[q][2] + static_cast<int>(static_cast<float>(lr) * w[q]));
        }
    }

    int up[4][3];
    if (!TOP) pyrup_quad_lds(s_up, up_r.x0, up_r.y0, up_r.x1 - up_r.x0, a.cw[l + 1], a.ch[l + 1], x >> 1, y >> 1, up);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float d = ws[q] + kWeightEps;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            // normalizeUsingWeightMap: static_cast<short>(v / (w + eps)); |v / d| < 2^31 always (d >= 1e-5)
            int v = wrap_s16(static_cast<int>(static_cast<float>(acc[q][ch]) / d));
            if (!TOP) v = sat_s16(up[q][ch] + v);                   // restoreImageFromLaplacePyr: cv::add saturates
            out[q][ch] = v;
        }
    }
}

template <bool TOP, bool FINAL>
__device__ __forceinline__ void blend_level(const CamDesc* __restrict__ cams, const BlendArgs& a, int l, int tx, int ty,
                                            uint32_t cam_bits, uint2* __restrict__ lds)
{
    const Region r = blend_region(a, tx, ty, l);
    Region ur = r;
    const uint2* s_up = nullptr;
    if (!TOP) {
        ur = blend_region(a, tx, ty, l + 1);
        s_up = lds + a.lds_off[l + 1];
    }
    uint2* s_out = FINAL ? nullptr : lds + a.lds_off[l];
    const int rw = r.x1 - r.x0, rh = r.y1 - r.y0;
    const int qw = (rw + 1) >> 1, qh = (rh + 1) >> 1;
    const int cw = a.cw[l], ch = a.ch[l];
    for (int q = threadIdx.x; q < qw * qh; q += 256) {
        const int qy = q / qw, qx = q - qy * qw;
        const int x = r.x0 + 2 * qx, y = r.y0 + 2 * qy;
        bool valid[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) valid[k] = (x + (k & 1)) < r.x1 && (y + (k >> 1)) < r.y1;
        int out[4][3];
        float ws[4];
        blend_quad<TOP, FINAL>(cams, a, l, x, y, valid, cam_bits, s_up, ur, out, ws);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (!valid[k]) continue;
            const int px = x + (k & 1), py = y + (k >> 1);
            if (FINAL) {
                if (px >= a.final_w || py >= a.final_h) continue;       // crop to dst_roi_final_
                const bool m = ws[k] > kWeightEps;                      // dst_mask_ = dst_band_weights_[0] > eps
                uint8_t* o = a.pano + static_cast<uint32_t>(py) * a.pano_stride + px * 3;
                o[0] = m ? static_cast<uint8_t>(sat_u8(out[k][0])) : 0;
                o[1] = m ? static_cast<uint8_t>(sat_u8(out[k][1])) : 0;
                o[2] = m ? static_cast<uint8_t>(sat_u8(out[k][2])) : 0;
                if (a.pano_mask) a.pano_mask[static_cast<uint32_t>(py) * a.mask_stride + px] = m ? 255 : 0;
            } else {
                s_out[(py - r.y0) * rw + (px - r.x0)] = pack3(out[k][0], out[k][1], out[k][2]);
                if (a.dbg_C[l]) {
                    const uint32_t plane = static_cast<uint32_t>(cw) * ch;
                    int16_t* dd = a.dbg_C[l] + static_cast<uint32_t>(py) * cw + px;
                    dd[0] = static_cast<int16_t>(out[k][0]);
                    dd[plane] = static_cast<int16_t>(out[k][1]);
                    dd[2 * plane] = static_cast<int16_t>(out[k][2]);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_blend(const CamDesc* __restrict__ cams, BlendArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2* lds = reinterpret_cast<uint2*>(smem);
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    const uint32_t cam_bits = a.block_cams[blockIdx.x];
    const int top = a.nlev - 1;
    if (top == 0) {
        blend_level<true, true>(cams, a, 0, tx, ty, cam_bits, lds);
        return;
    }
    blend_level<true, false>(cams, a, top, tx, ty, cam_bits, lds);
    __syncthreads();
    for (int l = top - 1; l >= 1; --l) {
        blend_level<false, false>(cams, a, l, tx, ty, cam_bits, lds);
        __syncthreads();
    }
    blend_level<false, true>(cams, a, 0, tx, ty, cam_bits, lds);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// launchers + host-side geometry helpers
// ------------------------------------------------------------------------------------------------------

void mb_tile_dims(int* a_tw0, int* a_th0, int* b_t, int* c_tw, int* c_th)
{
    *a_tw0 = A_TW0; *a_th0 = A_TH0; *b_t = B_T; *c_tw = C_TW; *c_th = C_TH;
}

void mb_blend_region(const BlendArgs& a, int tx, int ty, int level, int out[4])
{
    const Region r = blend_region(a, tx, ty, level);
    out[0] = r.x0; out[1] = r.y0; out[2] = r.x1; out[3] = r.y1;
}

void launch_warp_pyr(const CamDesc* d_cams, int kind, const TileList& list, const FrameArgs& frames, hipStream_t s)
{
    if (list.count <= 0) return;
    if (kind == 0) hipLaunchKernelGGL(k_warp_pyr<0>, dim3(list.count), dim3(256), 0, s, d_cams, list, frames);
    else if (kind == 1) hipLaunchKernelGGL(k_warp_pyr<1>, dim3(list.count), dim3(256), 0, s, d_cams, list, frames);
    else hipLaunchKernelGGL(k_warp_pyr<2>, dim3(list.count), dim3(256), 0, s, d_cams, list, frames);
}

void launch_pyr_chain(const CamDesc* d_cams, const TileList& list, int base, int k, hipStream_t s)
{
    if (list.count <= 0 || k <= 0) return;
    hipLaunchKernelGGL(k_pyr_chain, dim3(list.count), dim3(256), 0, s, d_cams, list, base, k);
}

void launch_blend(const CamDesc* d_cams, const BlendArgs& a, int num_blocks, size_t lds_bytes, hipStream_t s)
{
    hipLaunchKernelGGL(k_blend, dim3(num_blocks), dim3(256), lds_bytes, s, d_cams, a);
}

}  // namespace mst
