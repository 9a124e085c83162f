// kernels_common.hpp -- device helpers shared by kernels.hip and kernels_mb.hip.
//
// Scalar semantics follow the reference's x64/SSE2 build bit for bit (see kernels.hip header
// for the reference lines).  Everything is compiled with -ffp-contract=off and IEEE-correct
// f32 division.
#pragma once

#include <hip/hip_runtime.h>

#include <climits>

#include "device_types.hpp"

#pragma clang fp contract(off)

namespace mst {
namespace dev {

constexpr float kWeightEps = 1e-5f;   // WEIGHT_EPS, blenders.cpp:50

// Pointers read out of a descriptor are "generic" to the compiler, which then emits FLAT loads and stores:
// those tick both memory counters, complete out of order with respect to LDS traffic, and force every later
// wait to vmcnt(0) lgkmcnt(0) -- which serialises a software-pipelined loop.  Everything a descriptor points
// at is device global memory, so the kernels say so.
#define MST_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ __forceinline__ T MST_GLOBAL* gp(T* p) { return (T MST_GLOBAL*)p; }

// a trivially copyable object (size a multiple of 4) out of global memory
template <typename T> __device__ __forceinline__ T load_global(const T* p)
{
    static_assert(sizeof(T) % 4 == 0, "dword-sized objects only");
    T out;
    const uint32_t MST_GLOBAL* s = (const uint32_t MST_GLOBAL*)p;
    uint32_t* d = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) d[i] = s[i];
    return out;
}
template <typename T> __device__ __forceinline__ void store_global(T* p, const T& v)
{
    static_assert(sizeof(T) % 4 == 0, "dword-sized objects only");
    uint32_t MST_GLOBAL* d = (uint32_t MST_GLOBAL*)p;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&v);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 4; ++i) d[i] = s[i];
}

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

// ---- backward map (generic form, table per axis) -------------------------------------------------
__device__ __forceinline__ void backward_map(const CamDesc& c, int x, int y, float& mx, float& my)
{
    if (c.kind >= 3) {          // non-separable projector: the host-built map (device_types.hpp)
        const uint32_t i = static_cast<uint32_t>(y) * static_cast<uint32_t>(c.roi_w) + static_cast<uint32_t>(x);
        mx = gp(c.xmap)[i];
        my = gp(c.ymap)[i];
        return;
    }
    const float ca = gp(c.col_a)[x];
    const float ra = gp(c.row_a)[y];
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
    const float cb = gp(c.col_b)[x];
    if (c.kind == 1) {
        x_ = ra * ca;
        y_ = gp(c.row_b)[y];
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
    // exact {32768,0,0,0} produce the same byte for every input (tests/test_oracle_kat.py), so the
    // closed form is used.
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

// cv::remap's fixed-point bilinear sum for one BGR pixel whose 2x2 window lies inside the image, from the aligned
// dwords covering the two 6-byte pixel pairs (a0..a2 top row, c0..c2 bottom row; sh0/sh1 = byte offset of the
// pair inside its dwords).  With wx in {32-fx, fx} and wy in {32-fy, fy},
//   (sum v * wx * wy * 32 + 2^14) >> 15  ==  (sum_y wy * (sum_x wx * v) + 512) >> 10   in exact integers;
// after the funnel shift the pair sits at fixed bytes (B0 G0 R0 B1 | G1 R1 . .), so one more constant shift per
// channel lines every channel up with the same byte-0 / byte-3 weight register for v_dot4_u32_u8.
__device__ __forceinline__ void bilinear_window_c3(uint32_t a0, uint32_t a1, uint32_t a2, uint32_t c0, uint32_t c1,
                                                   uint32_t c2, uint32_t sh0, uint32_t sh1, int fx, int fy, int out[3])
{
    const uint32_t wx = static_cast<uint32_t>(32 - fx) | static_cast<uint32_t>(fx) << 24;
    const uint32_t wy0 = static_cast<uint32_t>(32 - fy), wy1 = static_cast<uint32_t>(fy);
    const uint32_t lo0 = __builtin_amdgcn_alignbyte(a1, a0, sh0), hi0 = __builtin_amdgcn_alignbyte(a2, a1, sh0);
    const uint32_t lo1 = __builtin_amdgcn_alignbyte(c1, c0, sh1), hi1 = __builtin_amdgcn_alignbyte(c2, c1, sh1);
    const uint32_t g0 = __builtin_amdgcn_alignbyte(hi0, lo0, 1), r0 = __builtin_amdgcn_alignbyte(hi0, lo0, 2);
    const uint32_t g1 = __builtin_amdgcn_alignbyte(hi1, lo1, 1), r1 = __builtin_amdgcn_alignbyte(hi1, lo1, 2);
    const uint32_t hb0 = __builtin_amdgcn_udot4(lo0, wx, 0u, false), hb1 = __builtin_amdgcn_udot4(lo1, wx, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(g0, wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1, wx, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(r0, wx, 0u, false), hr1 = __builtin_amdgcn_udot4(r1, wx, 0u, false);
    out[0] = static_cast<int>((__umul24(hb0, wy0) + __umul24(hb1, wy1) + 512u) >> 10);
    out[1] = static_cast<int>((__umul24(hg0, wy0) + __umul24(hg1, wy1) + 512u) >> 10);
    out[2] = static_cast<int>((__umul24(hr0, wy0) + __umul24(hr1, wy1) + 512u) >> 10);
}

// INTER_LINEAR + BORDER_REFLECT on CV_8UC3: the image path of RotationWarper::warp
__device__ __forceinline__ void sample_linear_reflect_c3(const uint8_t* __restrict__ src, size_t stride,
                                                         int sw, int sh, float mx, float my, int out[3])
{
    // common case: the 2x2 window lies inside the source and its two 12-byte reads stay inside the frame (they
    // run at most 8 bytes past the pixel pair: the next row, except on the last one) -> two aligned dwordx3 loads
    // and the separable v_dot4 form (bilinear_window_c3); frames up to 4 GiB with rows below 16 MiB only
    {
        const int fsx = cv_round(mx * 32.f), fsy = cv_round(my * 32.f);
        const int sx = sat_s16(fsx >> 5), sy = sat_s16(fsy >> 5);
        if (sx >= 0 && sx < sw - 1 && sy >= 0 && sy < sh - 1 && (sx < sw - 3 || sy < sh - 2) && stride < (size_t{1} << 24)) {
            const uint32_t lo2 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(src)) & 3u;
            const uint8_t* s_al = src - lo2;
            const uint32_t o0 = __umul24(sy, static_cast<uint32_t>(stride)) + static_cast<uint32_t>(sx) * 3u + lo2;
            const uint32_t o1 = o0 + static_cast<uint32_t>(stride);
            const uint32_t* q0 = reinterpret_cast<const uint32_t*>(s_al + (o0 & ~3u));
            const uint32_t* q1 = reinterpret_cast<const uint32_t*>(s_al + (o1 & ~3u));
            bilinear_window_c3(q0[0], q0[1], q0[2], q1[0], q1[1], q1[2], o0 & 3u, o1 & 3u, fsx & 31, fsy & 31, out);
            return;
        }
    }
    const Taps t = linear_taps<true>(mx, my, sw, sh);
    const uint8_t* r0 = src + static_cast<size_t>(t.y0) * stride;
    const uint8_t* r1 = src + static_cast<size_t>(t.y1) * stride;
    const uint8_t* p00 = r0 + t.x0 * 3; const uint8_t* p01 = r0 + t.x1 * 3;
    const uint8_t* p10 = r1 + t.x0 * 3; const uint8_t* p11 = r1 + t.x1 * 3;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        out[k] = blend_taps(t, p00[k], p01[k], p10[k], p11[k]);
}

// ---- backward map in cv::remap's fixed-point form (shared by the multi-band and the single-pass kernels) ----------
// The backward map of one destination pixel in the fixed-point form cv::remap uses: integer source
// coordinate of the 2x2 window, 5-bit fractions (0..32 after border folding), and whether the window can be
// read with the two 12-byte loads.
struct MapPx {
    int sx, sy, fx, fy;
    bool fast;
    // the same sample as a 2x2 window INSIDE the source (BORDER_REFLECT folded in) whether or not its 12-byte reads
    // would stay inside the frame: what a sampler that never reads past the window needs (LDS-staged windows)
    int ix, iy, ifx, ify;
    bool inside;
};

// KIND 3 stands for every non-separable projector: its map comes from the host-built xmap / ymap (rx, ry index it)
template <int KIND>
__device__ __forceinline__ MapPx map_px(const CamDesc& c, const float kr[9], int src_w, int src_h, const ColEntry& ce,
                                        const RowEntry& re, int rx, int ry)
{
    float X, Y, Z, mx, my;
    if (KIND == 3) {
        backward_map(c, rx, ry, mx, my);
    } else {
        if (KIND == 1) {
            const float x_ = re.ra * ce.ca, z_ = re.ra * ce.cb;
            X = (kr[0] * x_ + re.b[0]) + kr[2] * z_;
            Y = (kr[3] * x_ + re.b[1]) + kr[5] * z_;
            Z = (kr[6] * x_ + re.b[2]) + kr[8] * z_;
        } else {
            X = (ce.a[0] + re.b[0]) + ce.c[0];
            Y = (ce.a[1] + re.b[1]) + ce.c[1];
            Z = (ce.a[2] + re.b[2]) + ce.c[2];
        }
        if (KIND == 0 || Z > 0) { mx = X / Z; my = Y / Z; }
        else { mx = -1.f; my = -1.f; }
    }
    const int fsx = cv_round(mx * 32.f), fsy = cv_round(my * 32.f);
    int sx = sat_s16(fsx >> 5), sy = sat_s16(fsy >> 5);
    int fx = fsx & 31, fy = fsy & 31;
    // fast = the 2x2 window lies inside the source and its two 12-byte reads stay inside the frame: they may
    // run up to 8 bytes past the pixel pair, which is still this frame (next row) except on the last row
    auto window_ok = [&](int x, int y) {
        return x >= 0 && x < src_w - 1 && y >= 0 && y < src_h - 1 && (x < src_w - 3 || y < src_h - 2);
    };
    bool fast = window_ok(sx, sy);
    MapPx m;
    m.ix = sx; m.iy = sy; m.ifx = fx; m.ify = fy; m.inside = fast;
    if (!fast) {
        // BORDER_REFLECT folds both taps of an axis onto neighbouring (or the same) source pixels, so the
        // sample is again a 2x2 window with the two weights of the axis swapped or merged: the integer sum
        // sum(v * wx * wy) is the same one remap forms.  Only windows that cannot be read with the 12-byte
        // loads (the last pixels of the frame, one-pixel sources) stay on the per-tap path.
        const int x0 = reflect<0>(sx, src_w), x1 = reflect<0>(sx + 1, src_w);
        const int y0 = reflect<0>(sy, src_h), y1 = reflect<0>(sy + 1, src_h);
        int xs = min(x0, x1), ys = min(y0, y1);
        if (xs == src_w - 1) --xs;                  // both taps on the last pixel: window [len-2, len-1], all weight high
        if (ys == src_h - 1) --ys;
        const int wxl = (x0 == xs ? 32 - fx : 0) + (x1 == xs ? fx : 0);
        const int wyl = (y0 == ys ? 32 - fy : 0) + (y1 == ys ? fy : 0);
        m.ix = xs; m.iy = ys; m.ifx = 32 - wxl; m.ify = 32 - wyl;
        m.inside = xs >= 0 && xs < src_w - 1 && ys >= 0 && ys < src_h - 1;
        if (window_ok(xs, ys)) {
            sx = xs; sy = ys; fx = 32 - wxl; fy = 32 - wyl; fast = true;
        }
    }
    m.sx = sx; m.sy = sy; m.fx = fx; m.fy = fy; m.fast = fast;
    return m;
}

__device__ __forceinline__ float gain_at(const CamDesc& c, int x, int y)
{
    if (c.expos == 2) {
        const int x0 = gp(c.gx0)[x], x1 = gp(c.gx1)[x];
        const float a0 = gp(c.gxw0)[x], a1 = gp(c.gxw1)[x];
        const float MST_GLOBAL* g0 = gp(c.gain_map) + gp(c.gy0)[y] * c.gw;
        const float MST_GLOBAL* g1 = gp(c.gain_map) + gp(c.gy1)[y] * c.gw;
        const float b0 = gp(c.gyw0)[y], b1 = gp(c.gyw1)[y];
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


// ---- LDS-staged source windows (kernels_compose.hip explains the why; kernels_mb.hip's warp kernel uses them too) ----
typedef uint32_t __attribute__((ext_vector_type(3))) u32x3_t;
typedef uint32_t __attribute__((ext_vector_type(4))) u32x4_t;
// `groups` x `rows` groups of four pixels (12 source bytes on a dword boundary -> 16 LDS bytes, B G R x), by NT threads
// PAD: the window may reach outside the fw x fh frame (origin (wx0, wy0) in frame pixels, wx0 and fw multiples of four, so
// a group lies inside or outside as a whole); groups outside are zeros -- cv::remap's BORDER_CONSTANT
#ifndef MST_STAGE_U_PASTE
#define MST_STAGE_U_PASTE 3         // 512 threads: 6 K pixels per pass, 63 VGPRs (four blocks per CU)
#endif
#ifndef MST_STAGE_U_FEATHER
#define MST_STAGE_U_FEATHER 1       // batching costs the feather kernel its third block per CU (72 -> 84 VGPRs): -3 .. -7 %
#endif
#ifndef MST_STAGE_U_WARP
#define MST_STAGE_U_WARP 4          // 384 threads: 6 K pixels per pass
#endif
// U: loads a thread keeps in flight.  One load per trip of a plain loop means one memory round trip per trip (the LDS
// write needs the data): a 6 K pixel window over 512 threads is three round trips in a row -- with the staging loads
// ablated the LUT composite ran 42 % faster.  The trips are therefore taken U at a time: U loads issued back to back,
// then U LDS writes.  Same box, same run: kernel A 31.1 -> 29.1 us per frame set, C3 / C5 LUT 0.459 / 0.529 -> 0.474 / 0.551.
// pitch: the LDS row pitch in groups (0: = groups, rows back to back).  Kernel A pads it (round 5) so that a row step inside a
// half-wave's taps does not change the bank of a column.
template <int NT, bool PAD = false, int U = 4>
__device__ __forceinline__ void stage_window(uint32_t* __restrict__ s_win, const uint8_t MST_GLOBAL* __restrict__ base, uint32_t stride,
                                             int groups, int rows, int wx0 = 0, int wy0 = 0, int fw = 0, int fh = 0, int pitch = 0)
{
    const int n = groups * rows;
    if (pitch == 0) pitch = groups;
    if constexpr (U == 1) {      // one load per trip, scheduled by the compiler (fewest registers: the feather blend's choice)
        for (int i = threadIdx.x; i < n; i += NT) {
            const int r = i / groups, g = i - r * groups;
            u32x3_t d = {0u, 0u, 0u};
            if (!PAD || (static_cast<unsigned>(wy0 + r) < static_cast<unsigned>(fh) && static_cast<unsigned>(wx0 + 4 * g) < static_cast<unsigned>(fw)))
                d = *(const u32x3_t MST_GLOBAL*)(base + static_cast<ptrdiff_t>(r) * static_cast<ptrdiff_t>(stride) + static_cast<ptrdiff_t>(g) * 12);
            u32x4_t p;
            p.x = d.x; p.y = __builtin_amdgcn_alignbyte(d.y, d.x, 3); p.z = __builtin_amdgcn_alignbyte(d.z, d.y, 2); p.w = d.z >> 8;
            *reinterpret_cast<u32x4_t*>(&s_win[(r * pitch + g) * 4]) = p;
        }
        return;
    } else {
    // the window's base is block-uniform: as a scalar pair for the loads' saddr operand
    const uint64_t vb = reinterpret_cast<uint64_t>(base);
    const uint64_t sbase = static_cast<uint64_t>(static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(vb >> 32)))) << 32 |
                           static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<uint32_t>(vb)));      // (the builtin returns int)
    // (row, group) of trip k of this thread: i = threadIdx.x + k * NT, walked without a division per trip
    const uint32_t inv = ((1u << 20) + groups - 1) / groups;      // i / groups == (i * inv) >> 20 for i < 4096, groups <= 256
    const int step_r = static_cast<int>((static_cast<uint32_t>(NT) * inv) >> 20), step_g = NT - step_r * groups;
    int r = static_cast<int>((static_cast<uint32_t>(threadIdx.x) * inv) >> 20), g = static_cast<int>(threadIdx.x) - r * groups;
    for (int i0 = threadIdx.x; i0 < n; i0 += NT * U) {
        u32x3_t d[U];
        bool ok[U];
        int at[U];               // the group's place in LDS: row * pitch + group
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // trips past the end re-read the window's last row, groups outside the frame (PAD) the nearest one inside:
            // a load that is always legal needs no branch around it.  The loads are written as asm so that all U are
            // issued back to back: left to the compiler, the last ones sink below the first LDS writes (register
            // shuffling for ds_write_b128) and every trip waits for its own round trip again.
            int rr = min(r, rows - 1), gg = g;
            ok[u] = true;
            at[u] = r * pitch + g;
            if (PAD) {
                ok[u] = static_cast<unsigned>(wy0 + rr) < static_cast<unsigned>(fh) && static_cast<unsigned>(wx0 + 4 * gg) < static_cast<unsigned>(fw);
                rr = min(max(wy0 + rr, 0), fh - 1) - wy0;
                gg = (min(max(wx0 + 4 * gg, 0), fw - 4) - wx0) >> 2;
            }
            // (32-bit offset from the block-uniform base: frames are below 4 GiB, mst_compose* checks)
            const uint32_t off = static_cast<uint32_t>(rr) * stride + static_cast<uint32_t>(gg) * 12u;
#ifdef MST_ABLATE_STAGE          // (timing experiment, tools/variants.sh: what the source stream costs; the pixels are wrong)
            d[u] = u32x3_t{off, off, off};
#else
            asm volatile("global_load_dwordx3 %0, %1, %2" : "=v"(d[u]) : "v"(off), "s"(sbase) : "memory");
#endif
            r += step_r; g += step_g;
            if (g >= groups) { g -= groups; ++r; }
        }
        // every d[u] is an operand of the wait: nothing may read (or copy) one before it
        if constexpr (U == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]) : : "memory");
        else if constexpr (U == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]) : : "memory");
        else if constexpr (U == 5) asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]) : : "memory");
        else if constexpr (U == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]) : : "memory");
        else static_assert(U >= 3 && U <= 6, "stage_window: 1 or 3 to 6 loads in flight");
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * NT;
            if (i < n) {
                if (PAD && !ok[u]) d[u] = u32x3_t{0u, 0u, 0u};
#ifdef MST_ABLATE_NV12STAGE      // (timing experiment, round 5: what converting NV12 inside the staging loop would cost kernel A.  The group's first dword
                                 // stands in for four luma bytes, the second for two U V pairs; the arithmetic is cvtColor(COLOR_YUV2BGR_NV12)'s, per pixel,
                                 // as k_nv12_to_bgr does it.  The pixels are wrong -- mst_build_info reports the build)
                {
                    const uint32_t yy = d[u].x, uv = d[u].y;
                    uint32_t px4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int cu = static_cast<int>((uv >> (16 * (q >> 1))) & 0xFFu) - 128, cv = static_cast<int>((uv >> (16 * (q >> 1) + 8)) & 0xFFu) - 128;
                        const int ruv = (1 << 19) + 1673527 * cv, guv = (1 << 19) - 852492 * cv - 409993 * cu, buv = (1 << 19) + 2116026 * cu;
                        const int y0 = max(0, static_cast<int>((yy >> (8 * q)) & 0xFFu) - 16) * 1220542;
                        const int b = min(max((y0 + buv) >> 20, 0), 255), g = min(max((y0 + guv) >> 20, 0), 255), r = min(max((y0 + ruv) >> 20, 0), 255);
                        px4[q] = static_cast<uint32_t>(b) | static_cast<uint32_t>(g) << 8 | static_cast<uint32_t>(r) << 16;
                    }
                    *reinterpret_cast<u32x4_t*>(&s_win[at[u] * 4]) = u32x4_t{px4[0], px4[1], px4[2], px4[3]};
                    continue;
                }
#endif
                u32x4_t p;
                p.x = d[u].x;                                            // B0 G0 R0 (B1)
                p.y = __builtin_amdgcn_alignbyte(d[u].y, d[u].x, 3);     // B1 G1 R1 (B2)
                p.z = __builtin_amdgcn_alignbyte(d[u].z, d[u].y, 2);     // B2 G2 R2 (B3)
                p.w = d[u].z >> 8;                                       // B3 G3 R3 0
                *reinterpret_cast<u32x4_t*>(&s_win[at[u] * 4]) = p;      // (row r, group g) sits at r * pitch + g
            }
        }
    }
    }
}

typedef unsigned short __attribute__((ext_vector_type(2))) us2;
__device__ __forceinline__ uint32_t dot2_u16(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}
// The staged sampler on the four taps' dwords (top row lo0 hi0, bottom row lo1 hi1: B G R x each) with the fractions
// already unpacked.
__device__ __forceinline__ uint32_t sample_taps(uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1, uint32_t fx, uint32_t fy, float gain)
{
    const uint32_t wxp = __umul24(fx, 0x3FFFC0u) + 2048u;         // (32 - fx) * 64 | fx * 64 << 16
    const uint32_t wy1 = fy, wy0 = 32u - fy;
    // one byte of both pixels of a row per register (X0 | X1 << 16): three perms per row, vertical lerp on both pixels
    // at once (24-bit multiplies, no carry between the halves: 255 * 32 < 2^16), then the horizontal pass is one v_dot2
    // per channel with no shuffle in front of it
    const uint32_t bt = __builtin_amdgcn_perm(hi0, lo0, 0x0C040C00u), bb = __builtin_amdgcn_perm(hi1, lo1, 0x0C040C00u);
    const uint32_t gt = __builtin_amdgcn_perm(hi0, lo0, 0x0C050C01u), gb = __builtin_amdgcn_perm(hi1, lo1, 0x0C050C01u);
    const uint32_t rt = __builtin_amdgcn_perm(hi0, lo0, 0x0C060C02u), rb = __builtin_amdgcn_perm(hi1, lo1, 0x0C060C02u);
    uint32_t b = dot2_u16(__umul24(bt, wy0) + __umul24(bb, wy1), wxp, 32768u);
    uint32_t g = dot2_u16(__umul24(gt, wy0) + __umul24(gb, wy1), wxp, 32768u);
    uint32_t r = dot2_u16(__umul24(rt, wy0) + __umul24(rb, wy1), wxp, 32768u);
    asm("" : "+v"(b)); asm("" : "+v"(g)); asm("" : "+v"(r));      // (see sample_window)
    uint32_t px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((b >> 16) & 0xFFu) * gain, 0u, 0u);
    px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((g >> 16) & 0xFFu) * gain, 1u, px);
    px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((r >> 16) & 0xFFu) * gain, 2u, px);
    return px;
}
__device__ __forceinline__ uint32_t sample_lds(const uint32_t* __restrict__ p0, const uint32_t* __restrict__ p1, uint32_t e, float gain)
{
    return sample_taps(p0[0], p0[1], p1[0], p1[1], e & 63u, (e >> 6) & 63u, gain);
}


}  // namespace dev
}  // namespace mst
