// kernels_mb.hip -- the multi-band frame path as fused HIP kernels for gfx950.
//
//   A  k_warp_pyr    per camera tile: backward map + fixed-point bilinear sample + exposure gain
//                    -> level 0 of the BORDER_REFLECT-padded image (kept in LDS, also stored packed)
//                    -> cv::pyrDown to level 1, straight from LDS.
//                    (RotationWarper::warp, ExposureCompensator::apply, convertTo(16S),
//                     copyMakeBorder, first pyrDown: stitcher.cpp:275-285, blenders.cpp:272-274,481)
//   B  k_pyr_chain   per camera tile: two further pyrDown levels inside LDS per launch (blenders.cpp:481)
//   C1 k_blend_coarse  per level-1 canvas tile, every band above level 0 in one block: Laplacian = G_l - pyrUp(G_l+1)
//                    (blenders.cpp:482-487), short(lap * w) accumulate over cameras (:306-330), normalise by the
//                    re-summed weights (:383-405), collapse inside LDS (:520-530) -> collapsed level-1 canvas.
//   C2 k_blend_fine  per panorama tile: level-0 band + pyrUp of the collapsed level 1, crop + mask +
//                    convertTo(8U) (:369-376, stitcher.cpp:313).  No LDS, no barrier.
//
// Data layout: every Gaussian level of a CV_8U-sourced pyramid stays within [0,255], so a pixel is
// one dword B | G<<8 | R<<16.  The [1 4 6 4 1] sums are evaluated two channels at a time in the 16-bit
// halves of a register (SWAR): 256*255 + 128 < 65536, so no carry crosses a half.  All of this is exact
// integer arithmetic, identical to the reference's int accumulators.
//
// Work lists: only tiles whose output can reach the panorama (non-zero blend weight somewhere down
// the chain) are launched; the lists are built once per calibration on the host (mb_plan.cpp).
//
// Tuning notes (MI355X, rocprofv3, profiles/): per frame set the path moves ~240 MB at the
// fabric (3.9x the compulsory bytes) and issues ~33 M VALU wave-instructions.  Kernel A and the coarse canvas kernel
// are bound by instruction issue (50-65 % VALU-busy with the other pipes partly loaded), the level-0 canvas kernel by
// memory (5-6 TB/s), the pyrDown chain by dependent round trips per block.  What paid off, in order: never launching
// tiles that cannot reach the panorama, packed pixels + SWAR pyramids, the per-calibration map table with
// BORDER_REFLECT folded into it, integer paths wherever the weights are exactly 0/1, register-rolling quad-row
// strips with DPP neighbour exchange for those paths (6 loads per quad instead of 20), keeping the software
// pipelines intact through the compiler (global address space, no branches around in-flight loads, full
// unrolling), batching frame sets, and telling the caches what streams (non-temporal level-0 traffic) and which
// blocks share lines (launch order dealt to the XCDs in groups of neighbours).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "kernels.hpp"
#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace mst {

using namespace dev;

#if defined(MST_STAMPS)
// Diagnostic build only (-DMST_STAMPS, tools/stamps.sh): per block, shader-clock stamps at the phase boundaries of
// kernels A and C, written to a buffer nothing else reads and copied out by mst_debug_stamps().  Never timed.
__device__ unsigned long long g_stamps[3][1 << 17];       // k_warp_pyr, k_blend_coarse, k_blend_fine
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MST_STAMP(which, slot) do { if (threadIdx.x == 0 && blockIdx.x < (1u << 13)) g_stamps[which][blockIdx.x * 16 + (slot)] = stamp_now(); } while (0)
#define MST_STAMP_VAL(which, slot, v) do { if (threadIdx.x == 0 && blockIdx.x < (1u << 13)) g_stamps[which][blockIdx.x * 16 + (slot)] = (v); } while (0)
#else
#define MST_STAMP(which, slot) do {} while (0)
#define MST_STAMP_VAL(which, slot, v) do {} while (0)
#endif

namespace {

// ---- packed-pixel helpers --------------------------------------------------------------------------
// two dwords in global memory: a pixel pair of a Gaussian level, or a pixel of the collapsed level-1 canvas (HIP's uint2
// class cannot be used through address-space pointers)
typedef uint32_t __attribute__((ext_vector_type(2))) GPix;
// ---- level 0 of a camera's pyramid: packed BGR8 (device_types.hpp) ----
// eight bytes at a 2-byte-aligned address (gfx950 under HSA serves unaligned global accesses): a pixel pair's six bytes
// B0 G0 R0 B1 | G1 R1 . .  (the buffers carry 16 bytes of slack behind the last row)
typedef GPix __attribute__((aligned(2))) GPixU;
typedef uint32_t __attribute__((aligned(1))) U32U;
__device__ __forceinline__ const uint8_t MST_GLOBAL* l0_at(const CamDesc& c, int lx, int ly)
{
    return (const uint8_t MST_GLOBAL*)c.G[0] + (__umul24(static_cast<uint32_t>(ly), static_cast<uint32_t>(c.l0_pitch)) + static_cast<uint32_t>(kL0Bpp) * static_cast<uint32_t>(lx));
}
// pixels (lx, ly) and (lx + 1, ly) as the dwords the other levels store (B | G << 8 | R << 16; byte 3 of .x is B1: ignored by
// every reader, which extract bytes 0 .. 2)
__device__ __forceinline__ uint2 l0_pair(const CamDesc& c, int lx, int ly)
{
#if MST_L0_PACKED
    const GPix v = *(const GPixU MST_GLOBAL*)l0_at(c, lx, ly);
    return make_uint2(v.x, __builtin_amdgcn_alignbyte(v.y, v.x, 3));
#else            // (dword level 0: lx is even wherever pairs are read, rows are 16-byte aligned)
    const GPix v = *(const GPix MST_GLOBAL*)l0_at(c, lx, ly);
    return make_uint2(v.x, v.y);
#endif
}
__device__ __forceinline__ uint32_t l0_px(const CamDesc& c, int lx, int ly)
{
#if MST_L0_PACKED
    return *(const U32U MST_GLOBAL*)l0_at(c, lx, ly);
#else
    return *(const uint32_t MST_GLOBAL*)l0_at(c, lx, ly);
#endif
}
// four level-0 pixels (dwords) -> their twelve packed bytes
__device__ __forceinline__ u32x3_t l0_pack4(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3)
{
    return u32x3_t{__builtin_amdgcn_perm(p1, p0, 0x04020100u), __builtin_amdgcn_perm(p2, p1, 0x05040201u), __builtin_amdgcn_perm(p3, p2, 0x06050402u)};
}
// a group of four level-0 pixels (column gx, a multiple of four; row gy) leaves a warp tile: one non-temporal 12- or 16-byte store
// (non-temporal: level 0 is next read a whole batch later, by k_blend_fine; keeping it out of the caches leaves them to the
// frames and the map table: k_warp_pyr 287.5 -> 274.0 us per 8 frame sets)
__device__ __forceinline__ void l0_store4(uint8_t MST_GLOBAL* g0, uint32_t pitch0, int gx, int gy, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3)
{
    uint8_t MST_GLOBAL* o = g0 + (__umul24(static_cast<uint32_t>(gy), pitch0) + static_cast<uint32_t>(kL0Bpp) * static_cast<uint32_t>(gx));
#if MST_L0_PACKED
#if !defined(MST_NO_NT_STORE)
    __builtin_nontemporal_store(l0_pack4(p0, p1, p2, p3), reinterpret_cast<u32x3_t MST_GLOBAL*>(o));
#else
    *reinterpret_cast<u32x3_t MST_GLOBAL*>(o) = l0_pack4(p0, p1, p2, p3);
#endif
#else
#if !defined(MST_NO_NT_STORE)
    __builtin_nontemporal_store(u32x4_t{p0, p1, p2, p3}, reinterpret_cast<u32x4_t MST_GLOBAL*>(o));
#else
    *reinterpret_cast<u32x4_t MST_GLOBAL*>(o) = u32x4_t{p0, p1, p2, p3};
#endif
#endif
}
__device__ __forceinline__ uint32_t lanes_br(uint32_t p) { return p & 0x00FF00FFu; }
__device__ __forceinline__ uint32_t lane_g(uint32_t p) { return (p >> 8) & 0xFFu; }

// vertical/horizontal [1 4 6 4 1] on five packed pixels: (B,R lanes) and G
__device__ __forceinline__ void tap5(uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3, uint32_t p4, uint32_t& hbr,
                                     uint32_t& hg)
{
    hbr = lanes_br(p0) + lanes_br(p4) + 4u * (lanes_br(p1) + lanes_br(p3)) + 6u * lanes_br(p2);
    hg = lane_g(p0) + lane_g(p4) + 4u * (lane_g(p1) + lane_g(p3)) + 6u * lane_g(p2);
}

__device__ __forceinline__ uint32_t pyrdown_finish(uint32_t vbr, uint32_t vg)
{
    return (((vbr + 0x00800080u) >> 8) & 0x00FF00FFu) | (((vg + 128u) >> 8) << 8);
}

// values of the lanes beside this one within a DPP row of 16 lanes (row_shr:1 / row_shl:1; the row's first / last
// lane keeps its own value).  Callers fetch the outer neighbour of every row's end lanes themselves (one masked
// load whether there are two such lanes or eight).  A DPP read of a lane that a branch has switched off returns the
// reader's own `old` value, so the shifts must be executed by all lanes BEFORE any select: `edge ? loaded : shift(v)`
// evaluates shift() under the branch and silently hands the end lanes' neighbours stale data
// (tests/test_gpu_edge_cases.py caught exactly that).
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v)      // lane i <- lane i - 1
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x111, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t from_lane_above(uint32_t v)      // lane i <- lane i + 1
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x101, 0xF, 0xF, false));
}

// Workgroups are dealt to the 8 XCDs round-robin (each XCD has its own L2), so block b works on list entry
// xcd_run_index(b, n): XCD k = b % 8 walks one contiguous run of the list, and tiles that are neighbours in the
// image (shared source rows, shared pyramid halos) meet in one L2.  A bijection on [0, n) for every n.
__device__ __forceinline__ int xcd_run_index(int b, int n)
{
#if defined(MST_NO_XCD_SWIZZLE)
    return b;
#else
    const int q = n >> 3, r = n & 7, k = b & 7, j = b >> 3;
    return k * q + min(k, r) + j;
#endif
}

// ======================================================================================================
// A: warp + gain -> level 0 (LDS + global) -> pyrDown -> level 1
// ======================================================================================================
#ifndef MST_A_TH0
#define MST_A_TH0 32
#endif
#ifndef MST_ABLATE_A
#define MST_ABLATE_A 0     // timing experiments only: 1 no gather, 2 no gain, 4 no pyrDown, 8 no map, 16 no blend math,
                           // 32 every window on the fast path, 128 no level-0 store pass, 256 store level 0 of every tile
#endif
constexpr int A_TW0 = 124, A_TH0 = MST_A_TH0;  // level-0 interior of a tile
constexpr int A_LW = A_TW0 + 4, A_LH = A_TH0 + 4;   // with the 2-pixel pyrDown halo: 128 x 36
constexpr int A_TW1 = A_TW0 / 2, A_TH1 = A_TH0 / 2; // level-1 outputs: 62 x 16
constexpr int A_ROWS_PER_THREAD = A_LH / 2;    // 256 threads = 128 columns x 2 row halves
constexpr int A_GAIN_ROWS = 4;                 // gain-map rows a tile may span and still be served from LDS

// Everything about a tile that depends on the calibration only, as one block of A_BLOB dwords per launched tile,
// built by k_warp_table next to the map table: a thread block copies it into LDS with one coalesced load per
// thread, where it used to chase list -> descriptor -> row table -> gain map through four dependent round trips
// and three barriers (a quarter of the block's lifetime, by in-kernel stamps).
//   [0, 512)     BlocksGainCompensator: HResizeLinear of the A_GAIN_ROWS gain-map rows the tile spans, per column
//   [512, 548)   per tile row: offsets (in floats) of its two map rows in that table, low | high << 16
//   [548, 620)   per tile row: the two VResizeLinear weights
//   [620, 624)   gain mode (see fetch_pixel), "some gain >= 8e6" flag, tile_base.x, tile_base.y
constexpr int A_BLOB = 640, A_BLOB_GY = A_GAIN_ROWS * A_LW, A_BLOB_W = A_BLOB_GY + A_LH, A_BLOB_META = A_BLOB_W + 2 * A_LH;
static_assert(A_BLOB_META + 4 <= A_BLOB && A_BLOB <= 3 * 256, "tile block layout");

// saturate_cast<uchar>(uchar * float) = clamp(cvRound(product), 0, 255).  For gain < 8e6 the product is far inside
// int range and clamping in float first is the same function (NaN -> 0 both ways): that bounded form is what
// v_cvt_pk_u8_f32 computes (blend_pixel); tiles with a larger gain use the general apply_gain.
struct PixelFetch {           // one pixel between "addresses issued" and "blended"
    uint32_t a0, a1, a2, c0, c1, c2;   // raw dwords covering the two 6-byte pixel pairs (fast path)
    uint32_t sh0, sh1;                 // byte offsets of the pairs inside those dwords
    uint32_t wx, wy0, wy1;             // bilinear weights: (32 - fx) | fx << 24, 32 - fy, fy
    int sx, sy;                        // integer source coordinate (slow path)
    float gain;
    bool fast;
};

struct WarpConsts {           // per-camera scalars, copied out of the descriptor once
    int src_w, src_h, expos, gw, gh;
    float gain;
    const float* gain_map;
    const uint8_t* src;                // frame base
    const uint8_t* src_al;             // frame base rounded down to a dword boundary, src_lo2 bytes below src
    uint32_t src_lo2;
    uint32_t stride;                   // < 2^24 (checked by the engine): row offsets use the 24-bit multiplier
};

// The map is frame-invariant (the reference's driver caches xmap/ymap the same way, DRV:658-660,712-713), so it
// is tabulated once per calibration: that trades two IEEE divides, the rotation products and the cvRound logic
// per pixel per frame (~55 VALU instructions) for one coalesced 4-byte load.  Table entry, relative to the
// tile's smallest window coordinate (bx, by):  dx | dy << 10 | fx << 19 | fy << 25 | fast << 31; a window
// further than 1023 x 511 from that corner (extreme stretch) is simply left to the per-tap path, which
// re-evaluates the map.
__device__ __forceinline__ uint32_t pack_map(const MapPx& m, int bx, int by)
{
    const uint32_t dx = static_cast<uint32_t>(m.sx - bx), dy = static_cast<uint32_t>(m.sy - by);
    const bool ok = m.fast && dx < 1024u && dy < 512u;
    return ok ? (dx | dy << 10 | static_cast<uint32_t>(m.fx) << 19 | static_cast<uint32_t>(m.fy) << 25 | 1u << 31) : 0u;
}
__device__ __forceinline__ MapPx unpack_map(uint32_t e, int bx, int by)
{
    MapPx m;
    m.sx = bx + static_cast<int>(e & 1023u);
    m.sy = by + static_cast<int>((e >> 10) & 511u);
    m.fx = static_cast<int>((e >> 19) & 63u);
    m.fy = static_cast<int>((e >> 25) & 63u);
    m.fast = (e >> 31) != 0;
    return m;
}

// GAIN: how the exposure gain of the pixel is formed -- 0: the camera's scalar gain (1 for "no compensation":
// uchar * 1.0f rounds to itself), 1: block gain map, horizontally interpolated rows of this tile in LDS,
// 2: block gain map read from global memory (tile spans more map rows than the LDS table holds; rare, slow).
// The mode is a per-tile constant, so the loop body carries no uniform branch.
struct GainSrc {
    const float* col;          // LDS: this column of the tile's gain rows
    const uint32_t* gy;        // LDS: per tile row, offsets of its two map rows
    const float* w;            // LDS: per tile row, the two weights (w[row], w[A_LH + row])
    const CamDesc* cam;        // mode 2
    int rx, ly0;               // mode 2: roi column of this thread, level-0 row of tile row 0
};

template <int GAIN>
__device__ __forceinline__ void fetch_pixel(const WarpConsts& k, const MapPx& e, const GainSrc& gs, int row, PixelFetch& f)
{
    const int fx = e.fx, fy = e.fy;
    f.sx = e.sx;
    f.sy = e.sy;
    f.wx = static_cast<uint32_t>(32 - fx) | static_cast<uint32_t>(fx) << 24;
    f.wy0 = static_cast<uint32_t>(32 - fy); f.wy1 = static_cast<uint32_t>(fy);
    f.fast = (MST_ABLATE_A & 32) ? true : e.fast;
    {
        // Fast path = a 2x2 window inside the source whose two 6-byte pixel pairs can be read as aligned dwords;
        // the byte funnel shift happens at blend time.  The loads are issued unconditionally (pixels off the
        // fast path read the frame's first bytes and ignore them): a branch here would put the register copies
        // of its join -- and with them a wait for the data -- right behind the loads and serialise the loop.
        const uint32_t ox = f.fast ? static_cast<uint32_t>(f.sx) : 0u, oy = f.fast ? static_cast<uint32_t>(f.sy) : 0u;
        // 32-bit byte offsets from the (uniform) dword-aligned frame base: scalar base + vector offset addressing
        const uint32_t o0 = __umul24(oy, k.stride) + ox * 3u + k.src_lo2;
        const uint32_t o1 = o0 + k.stride;
        f.sh0 = o0 & 3u;
        f.sh1 = o1 & 3u;
        const uint32_t* q0 = reinterpret_cast<const uint32_t*>(k.src_al + (o0 & ~3u));
        const uint32_t* q1 = reinterpret_cast<const uint32_t*>(k.src_al + (o1 & ~3u));
        if (MST_ABLATE_A & 1) {
            f.a0 = f.sx; f.a1 = f.sy; f.a2 = f.sx ^ f.sy; f.c0 = f.sh0; f.c1 = f.sh1; f.c2 = f.sx + f.sy;
        } else {
            f.a0 = q0[0]; f.a1 = q0[1]; f.a2 = q0[2];
            f.c0 = q1[0]; f.c1 = q1[1]; f.c2 = q1[2];
        }
    }
    // exposure gain: k.gain is the camera's scalar gain (1 for "no compensation": uchar * 1.0f rounds to itself);
    // with a block gain map the two horizontally interpolated map rows of this tile row come from LDS.
    f.gain = k.gain;
    if (!(MST_ABLATE_A & 2)) {
        if (GAIN == 1) {
            const uint32_t o = gs.gy[row];
            const float r0 = gs.col[o & 0xFFFFu], r1 = gs.col[o >> 16];
            f.gain = r0 * gs.w[row] + r1 * gs.w[A_LH + row];                 // VResizeLinear
        } else if (GAIN == 2) {
            const CamDesc& c = *gs.cam;
            const int by = reflect<1>(gs.ly0 + row, c.lh[0]);
            f.gain = gain_at(c, gs.rx, reflect<0>(by - c.top, c.roi_h));
        }
    }
}

// SLOW = the border path (per-tap BORDER_REFLECT, byte loads); the main loop only runs the fast form
// and patches the few border pixels afterwards, so that no divergent load sits inside the pipeline.
// BIG = some gain of the tile is >= 8e6: use the general saturate_cast form (int overflow -> 0x80000000 -> 0).
template <bool SLOW, bool BIG>
__device__ __forceinline__ uint32_t blend_pixel(const WarpConsts& k, const PixelFetch& f)
{
    int b, g, r;
    if ((MST_ABLATE_A & 16) && !SLOW) {
        b = (f.a0 ^ f.c0) & 0xFF; g = (f.a1 ^ f.c1) & 0xFF; r = (f.a2 ^ f.c2 ^ f.wx ^ f.wy0 ^ f.wy1) & 0xFF;
    } else if (!SLOW) {
        // cv::remap's fixed-point bilinear sum, separably: with wx in {32-fx, fx} and wy in {32-fy, fy},
        // sum(v * wx * wy * 32) + 2^14 >> 15  ==  (sum_y wy * (sum_x wx * v)) + 512 >> 10  in exact integers.
        // After the funnel shifts the pairs sit at fixed bytes (B0 G0 R0 B1 | G1 R1 . .), so one more constant
        // shift per channel lines every channel up with the same byte-0 / byte-3 weight register for v_dot4.
        const uint32_t lo0 = __builtin_amdgcn_alignbyte(f.a1, f.a0, f.sh0), hi0 = __builtin_amdgcn_alignbyte(f.a2, f.a1, f.sh0);
        const uint32_t lo1 = __builtin_amdgcn_alignbyte(f.c1, f.c0, f.sh1), hi1 = __builtin_amdgcn_alignbyte(f.c2, f.c1, f.sh1);
        const uint32_t g0w = __builtin_amdgcn_alignbyte(hi0, lo0, 1), r0w = __builtin_amdgcn_alignbyte(hi0, lo0, 2);
        const uint32_t g1w = __builtin_amdgcn_alignbyte(hi1, lo1, 1), r1w = __builtin_amdgcn_alignbyte(hi1, lo1, 2);
        const uint32_t hb0 = __builtin_amdgcn_udot4(lo0, f.wx, 0u, false), hb1 = __builtin_amdgcn_udot4(lo1, f.wx, 0u, false);
        const uint32_t hg0 = __builtin_amdgcn_udot4(g0w, f.wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1w, f.wx, 0u, false);
        const uint32_t hr0 = __builtin_amdgcn_udot4(r0w, f.wx, 0u, false), hr1 = __builtin_amdgcn_udot4(r1w, f.wx, 0u, false);
        b = static_cast<int>((__umul24(hb0, f.wy0) + __umul24(hb1, f.wy1) + 512u) >> 10);
        g = static_cast<int>((__umul24(hg0, f.wy0) + __umul24(hg1, f.wy1) + 512u) >> 10);
        r = static_cast<int>((__umul24(hr0, f.wy0) + __umul24(hr1, f.wy1) + 512u) >> 10);
    } else {
        const int wx0 = static_cast<int>(f.wx & 0xFFu), wx1 = static_cast<int>(f.wx >> 24);
        const int w00 = wx0 * static_cast<int>(f.wy0) * 32, w01 = wx1 * static_cast<int>(f.wy0) * 32;
        const int w10 = wx0 * static_cast<int>(f.wy1) * 32, w11 = wx1 * static_cast<int>(f.wy1) * 32;
        const int x0 = reflect<0>(f.sx, k.src_w), x1 = reflect<0>(f.sx + 1, k.src_w);
        const int y0 = reflect<0>(f.sy, k.src_h), y1 = reflect<0>(f.sy + 1, k.src_h);
        const uint8_t* r0 = k.src + static_cast<uint32_t>(y0) * k.stride;
        const uint8_t* r1 = k.src + static_cast<uint32_t>(y1) * k.stride;
        const uint8_t* p00 = r0 + x0 * 3; const uint8_t* p01 = r0 + x1 * 3;
        const uint8_t* p10 = r1 + x0 * 3; const uint8_t* p11 = r1 + x1 * 3;
        b = (p00[0] * w00 + p01[0] * w01 + p10[0] * w10 + p11[0] * w11 + (1 << 14)) >> 15;
        g = (p00[1] * w00 + p01[1] * w01 + p10[1] * w10 + p11[1] * w11 + (1 << 14)) >> 15;
        r = (p00[2] * w00 + p01[2] * w01 + p10[2] * w10 + p11[2] * w11 + (1 << 14)) >> 15;
    }
    if (!(MST_ABLATE_A & 2)) {
        if (BIG) {
            b = apply_gain(b, f.gain); g = apply_gain(g, f.gain); r = apply_gain(r, f.gain);
        } else {
            // saturate_cast<uchar>(uchar * gain) for gains below 8e6 is clamp(rint(product), 0, 255) with NaN -> 0 (see
            // apply_gain_bounded), which is what v_cvt_pk_u8_f32 computes in the default rounding mode -- checked on
            // gfx950 over 2^28 sampled bit patterns and every tie (tools/ubench/cvt_pk_u8.hip) -- and it drops the byte
            // into the packed pixel as well: three instructions per channel instead of seven.
            uint32_t px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>(b) * f.gain, 0u, 0u);
            px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>(g) * f.gain, 1u, px);
            px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>(r) * f.gain, 2u, px);
            return px;
        }
    }
    return static_cast<uint32_t>(b) | (static_cast<uint32_t>(g) << 8) | (static_cast<uint32_t>(r) << 16);
}

// Per-calibration: the backward map of every pixel of every launched tile, in the tile/row/column order in
// which k_warp_pyr's threads consume it, relative to the tile's smallest window coordinate, and the tile's block of
// constants (layout above).
template <int KIND>
__global__ __launch_bounds__(256) void k_warp_table(const CamDesc* __restrict__ cams, TileList list,
                                                    uint32_t* __restrict__ table, uint32_t* __restrict__ blobs)
{
    __shared__ RowEntry s_rows[A_LH];
    __shared__ int s_ry[A_LH];
    __shared__ int s_min[2];
    const uint32_t t = list.tiles[blockIdx.x];
    const int cam = (t >> 24) & 0x7F, ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
    const CamDesc& c = cams[cam];
    const int bw = c.lw[0], bh = c.lh[0];
    const int lx0 = tx * A_TW0 - 2, ly0 = ty * A_TH0 - 2;
    const int tid = threadIdx.x;
    float kr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) kr[i] = c.kr[i];
    // per tile row: pyrDown's REFLECT_101 into the bordered image, then copyMakeBorder's REFLECT into the roi
    if (tid < A_LH) {
        const int by = reflect<1>(ly0 + tid, bh);
        const int ry = reflect<0>(by - c.top, c.roi_h);
        s_rows[tid] = load_global(c.rows + ry);
        s_ry[tid] = ry;
    }
    if (tid == 0) { s_min[0] = INT_MAX; s_min[1] = INT_MAX; }
    const int col = tid & (A_LW - 1);
    const int half = tid >> 7;
    const int bx = reflect<1>(lx0 + col, bw);
    const int rx = reflect<0>(bx - c.left, c.roi_w);
    const ColEntry ce = load_global(c.cols + rx);
    __syncthreads();
    int x_lo = INT_MAX, y_lo = INT_MAX;
    for (int it = 0; it < A_ROWS_PER_THREAD; ++it) {
        const int row = half * A_ROWS_PER_THREAD + it;
        const MapPx m = map_px<KIND>(c, kr, c.src_w, c.src_h, ce, s_rows[row], rx, s_ry[row]);
        if (m.fast) { x_lo = min(x_lo, m.sx); y_lo = min(y_lo, m.sy); }
    }
    if (x_lo != INT_MAX) { atomicMin(&s_min[0], x_lo); atomicMin(&s_min[1], y_lo); }
    __syncthreads();
    const int tbx = s_min[0] == INT_MAX ? 0 : s_min[0], tby = s_min[1] == INT_MAX ? 0 : s_min[1];
    uint32_t* __restrict__ out = table + static_cast<size_t>(list.first + blockIdx.x) * (A_LH * A_LW) + col;
    for (int it = 0; it < A_ROWS_PER_THREAD; ++it) {
        const int row = half * A_ROWS_PER_THREAD + it;
        out[row * A_LW] = pack_map(map_px<KIND>(c, kr, c.src_w, c.src_h, ce, s_rows[row], rx, s_ry[row]), tbx, tby);
    }

    // ---- the tile's block of constants
    uint32_t* __restrict__ blob = blobs + static_cast<size_t>(list.first + blockIdx.x) * A_BLOB;
    // BlocksGainCompensator's resize(gain_map): which map rows the tile touches
    int gmin = INT_MAX, gmax = INT_MIN;
    for (int r = 0; r < A_LH; ++r) {
        gmin = min(gmin, s_rows[r].gy0);
        gmax = max(gmax, s_rows[r].gy1);
    }
    // how this tile forms its gains (see fetch_pixel) and whether any of them can overflow the bounded rounding
    const float scalar_gain = c.expos == 0 ? 1.f : c.gain;
    int gain_mode = 0;
    int big = scalar_gain >= 8.0e6f;
    if (c.expos == 2) {
        if (gmax - gmin < A_GAIN_ROWS) {
            gain_mode = 1;
            bool col_big = false;
            if (half == 0) {
#pragma unroll
                for (int i = 0; i < A_GAIN_ROWS; ++i) {
                    // this column's horizontal interpolation of map row gmin + i (same float products and sums as the
                    // per-pixel form)
                    const float MST_GLOBAL* g = gp(c.gain_map) + min(gmin + i, c.gh - 1) * c.gw;
                    const float v = g[ce.gx0] * ce.gxw0 + g[ce.gx1] * ce.gxw1;
                    blob[i * A_LW + col] = __float_as_uint(v);
                    col_big |= v >= 4.0e6f;     // conservative: the vertical interpolation stays below 2x the rows
                }
            }
            big = __syncthreads_or(col_big) != 0;
        } else {
            gain_mode = 2;
            big = 1;
        }
    }
    if (gain_mode != 1 && half == 0)
        for (int i = 0; i < A_GAIN_ROWS; ++i) blob[i * A_LW + col] = 0u;
    if (tid < A_LH) {
        const RowEntry re = s_rows[tid];
        const bool lds_rows = gain_mode == 1;
        blob[A_BLOB_GY + tid] = lds_rows ? static_cast<uint32_t>((re.gy0 - gmin) * A_LW) | static_cast<uint32_t>((re.gy1 - gmin) * A_LW) << 16 : 0u;
        blob[A_BLOB_W + tid] = __float_as_uint(re.gyw0);
        blob[A_BLOB_W + A_LH + tid] = __float_as_uint(re.gyw1);
    }
    if (tid == 0) {
        blob[A_BLOB_META + 0] = static_cast<uint32_t>(gain_mode);
        blob[A_BLOB_META + 1] = static_cast<uint32_t>(big);
        blob[A_BLOB_META + 2] = static_cast<uint32_t>(tbx);
        blob[A_BLOB_META + 3] = static_cast<uint32_t>(tby);
    }
    if (tid >= A_BLOB_META + 4 - 512 + 0 && tid < A_BLOB - 512) blob[512 + tid] = 0u;       // padding: deterministic
}

#ifndef MST_A_WAVES
#define MST_A_WAVES 7      // 7 blocks of 256 threads per CU is what the LDS tile allows: keep the registers within that
#endif
template <int KIND>
__global__ __launch_bounds__(256, MST_A_WAVES) void k_warp_pyr(const CamDesc* __restrict__ cams_all, TileList list, FrameBatch batch,
                                                  const uint32_t* __restrict__ table, const uint32_t* __restrict__ blobs,
                                                  int n_cams, int n_sets)
{
    // One block per (tile, frame set) of a batched launch, on a 1-D grid ordered so that the frame sets of a tile are
    // consecutive entries of one XCD's run (xcd_run_index): they are dispatched together onto that XCD, and the
    // tile's slice of the backward-map table -- frame-invariant, 18 KB, and at 46.6 MB per frame set the largest
    // stream of this kernel when every frame set fetched it from HBM for itself -- is fetched once into that XCD's
    // L2 and served from there to the other frame sets.  (With the frame set on grid.y the same tile's blocks were a
    // whole grid row apart: different XCDs, and long evicted.)  A frame set has its own frames and its own copy of
    // the descriptors (pyramids).
    MST_STAMP(0, 0);
    __shared__ __attribute__((aligned(16))) uint32_t s_l0[A_LH][A_LW];   // level-0 tile with halo, packed (18 KB)
    __shared__ __attribute__((aligned(16))) uint32_t s_blob[A_BLOB];     // the tile's calibration constants
    const int tid = threadIdx.x;
    const int pair = xcd_run_index(blockIdx.x, list.count * n_sets);
    const int set = pair % n_sets;
    const int li = pair / n_sets;                               // position in the tile list
    const int gi = list.index ? static_cast<int>(list.index[li]) : list.first + li;     // position in the calibration's full list
    // the tile's block of constants: its address needs nothing but the block index, so the loads go out first
    {
        const uint32_t MST_GLOBAL* __restrict__ gb = gp(blobs) + static_cast<size_t>(gi) * A_BLOB;
        const uint32_t v0 = gb[tid], v1 = gb[tid + 256];
        const uint32_t v2 = tid < A_BLOB - 512 ? gb[tid + 512] : 0u;
        s_blob[tid] = v0;
        s_blob[tid + 256] = v1;
        if (tid < A_BLOB - 512) s_blob[tid + 512] = v2;
    }
    const CamDesc* __restrict__ cams = cams_all + set * n_cams;
    const FrameArgs& frames = batch.f[set];
    const uint32_t t = list.tiles[list.index ? gi : li];
    const int cam = (t >> 24) & 0x7F, ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
    const CamDesc& c = cams[cam];
    const int bw = c.lw[0], bh = c.lh[0], nlev = c.nlev;
    const int lx0 = tx * A_TW0 - 2, ly0 = ty * A_TH0 - 2;    // level-0 origin of the halo'd tile
    uint32_t MST_GLOBAL* __restrict__ g1out = gp(c.G[1]);
    const int lw1 = c.lw[1], lh1 = c.lh[1];

    WarpConsts k;
    k.src_w = c.src_w; k.src_h = c.src_h; k.expos = c.expos; k.gw = c.gw; k.gh = c.gh;
    k.gain = c.expos == 0 ? 1.f : c.gain; k.gain_map = c.gain_map;
    k.src = frames.ptr[cam]; k.stride = frames.stride[cam];
    k.src_lo2 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(k.src)) & 3u;
    k.src_al = k.src - k.src_lo2;
    // this tile's slice of the per-calibration map table: [row][col], one coalesced 4-byte load per pixel
    const uint32_t* __restrict__ tab = table + static_cast<size_t>(gi) * (A_LH * A_LW);
    const int col = tid & (A_LW - 1);
    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);
    __syncthreads();                                             // the tile's constants are in LDS
    const int gain_mode = __builtin_amdgcn_readfirstlane(static_cast<int>(s_blob[A_BLOB_META + 0]));
    const bool big = __builtin_amdgcn_readfirstlane(static_cast<int>(s_blob[A_BLOB_META + 1])) != 0;
    const int tbx = __builtin_amdgcn_readfirstlane(static_cast<int>(s_blob[A_BLOB_META + 2]));
    const int tby = __builtin_amdgcn_readfirstlane(static_cast<int>(s_blob[A_BLOB_META + 3]));
    GainSrc gs;
    gs.col = reinterpret_cast<const float*>(s_blob) + col;
    gs.gy = s_blob + A_BLOB_GY;
    gs.w = reinterpret_cast<const float*>(s_blob) + A_BLOB_W;
    gs.cam = &c;
    gs.ly0 = ly0;
    // roi column of this thread: pyrDown's REFLECT_101 into the bordered image, then copyMakeBorder's REFLECT into
    // the roi (only the rare paths below evaluate anything per column: the map table and the tile's constants
    // already carry the rest)
    gs.rx = reflect<0>(reflect<1>(lx0 + col, bw) - c.left, c.roi_w);
    MST_STAMP(0, 1);

    const int row0 = half * A_ROWS_PER_THREAD;
    // software pipeline over two explicit register sets: the gather of the next pixel is in flight while
    // the current one is blended (A_ROWS_PER_THREAD is even; the last prefetch re-reads the last row)
    static_assert(A_ROWS_PER_THREAD % 2 == 0, "two-stage pipeline needs an even row count");
    // Level 0 goes to LDS only inside the loop and to global memory in one pass after it.  A store issued between
    // the gathers is not free even though nothing waits for it explicitly: the memory counter retires in issue
    // order, so every gather issued after a store could only be consumed once that store had been acknowledged by
    // L2 -- which tied the loop to the write latency (11.6 of this kernel's 37.5 us per frame set, by ablation).
    // (Nor a store behind a uniform condition: the branch alone cuts the loop body into basic blocks and pins the
    // loads behind it.  Levels whose rows are not 16-byte aligned -- one band -- take a dword-wide pass instead.)
    auto emit = [&](int row, uint32_t px) { s_l0[row][col] = px; };
    const int row_last = row0 + A_ROWS_PER_THREAD - 1;
    const uint32_t* __restrict__ tcol = tab + col;
    auto entry = [&](int row) -> MapPx {
        if (MST_ABLATE_A & 8) {
            MapPx m;
            m.sx = col + 200; m.sy = row + 200; m.fx = (col * 7) & 31; m.fy = (row * 3) & 31; m.fast = true;
            return m;
        }
        return unpack_map(tcol[row * A_LW], tbx, tby);
    };
    // Table entries run two rows ahead of the gather, the gather one row ahead of the blend.  The loop body is
    // one basic block (no uniform branch on the gain mode, no divergent branch for border pixels: those are
    // redone after the loop), so the scheduler is free to hoist the loads of both pixels.
    auto run = [&](auto gain_tag, auto big_tag) {
        constexpr int GM = decltype(gain_tag)::value;
        constexpr bool BG = decltype(big_tag)::value;
        static_assert(A_ROWS_PER_THREAD <= 32, "slow_rows is a 32-bit mask");
        PixelFetch fa, fb;
        uint32_t slow_rows = 0;           // rows of this column that need the per-tap border path
        MapPx e0 = entry(row0), e1 = entry(row0 + 1);
        fetch_pixel<GM>(k, e0, gs, row0, fa);
        // fully unrolled: a back edge would carry the in-flight gather registers through copies, and a copy of
        // a register that is still being loaded is a wait for the load
#pragma unroll
        for (int it = 0; it < A_ROWS_PER_THREAD; it += 2) {
            const int row = row0 + it;
            e0 = entry(min(row + 2, row_last));
            fetch_pixel<GM>(k, e1, gs, row + 1, fb);
            e1 = entry(min(row + 3, row_last));
            slow_rows |= static_cast<uint32_t>(!fa.fast) << it;
            emit(row, blend_pixel<false, BG>(k, fa));
            fetch_pixel<GM>(k, e0, gs, min(row + 2, row_last), fa);
            slow_rows |= static_cast<uint32_t>(!fb.fast) << (it + 1);
            emit(row + 1, blend_pixel<false, BG>(k, fb));
        }
        while (slow_rows) {               // rare: windows at the very end of the frame, one-pixel sources
            const int it = __builtin_ctz(slow_rows);
            slow_rows &= slow_rows - 1;
            // (the table does not keep coordinates of windows it cannot address: evaluate the map from the camera's
            // per-column / per-row constants)
            float kr[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) kr[i] = c.kr[i];
            const int by = reflect<1>(ly0 + row0 + it, bh);
            const int ry = reflect<0>(by - c.top, c.roi_h);
            const ColEntry ce = load_global(c.cols + gs.rx);
            const RowEntry re = load_global(c.rows + ry);
            fetch_pixel<GM>(k, map_px<KIND>(c, kr, k.src_w, k.src_h, ce, re, gs.rx, ry), gs, row0 + it, fa);
            emit(row0 + it, blend_pixel<true, BG>(k, fa));
        }
    };
    using Yes = std::true_type;
    using No = std::false_type;
    if (gain_mode == 1) {
        if (big) run(std::integral_constant<int, 1>{}, Yes{}); else run(std::integral_constant<int, 1>{}, No{});
    } else if (gain_mode == 2) {
        run(std::integral_constant<int, 2>{}, Yes{});
    } else {
        if (big) run(std::integral_constant<int, 0>{}, Yes{}); else run(std::integral_constant<int, 0>{}, No{});
    }
    MST_STAMP(0, 2);
    __syncthreads();
    const bool store0 = !(t >> 31) || (MST_ABLATE_A & 256);    // (bit 31: nothing reads this tile's level 0 from memory, see mb_plan.cpp)
    if (store0 && !(MST_ABLATE_A & 128)) {
        // interior of the tile as packed BGR8, four pixels (twelve bytes) per lane: a tile row is 31 such groups; the tile
        // origin is a multiple of 124 and the row pitch covers whole groups, so a group that starts inside the level is
        // stored whole, dword-aligned (8-byte aligned in LDS: the interior starts at column 2)
        constexpr int kGroups = A_TW0 / 4;
        uint8_t MST_GLOBAL* __restrict__ g0b = (uint8_t MST_GLOBAL*)c.G[0];
        const uint32_t pitch0 = static_cast<uint32_t>(c.l0_pitch);
        for (int i = tid; i < A_TH0 * kGroups; i += 256) {
            const int r = i / kGroups, g = i - r * kGroups;
            const int gy = ly0 + 2 + r, gx4 = lx0 + 2 + 4 * g;
            if (gy >= bh || gx4 >= bw) continue;
            const uint2 lo = *reinterpret_cast<const uint2*>(&s_l0[2 + r][2 + 4 * g]);
            const uint2 hi = *reinterpret_cast<const uint2*>(&s_l0[2 + r][4 + 4 * g]);
            l0_store4(g0b, pitch0, gx4, gy, lo.x, lo.y, hi.x, hi.y);
        }
    }
    if (nlev < 2 || (MST_ABLATE_A & 4)) return;
    MST_STAMP(0, 3);

    // pyrDown straight from the LDS tile: output (j, i) reads rows 2i..2i+4, columns 2j..2j+4.  A thread owns one
    // output column and a run of A_TH1 / 4 consecutive output rows, so the horizontal [1 4 6 4 1] of a tile row is
    // formed once for every output row that uses it (2 R + 3 row sums for R outputs instead of 5 R).
    const int x1 = tx * A_TW1, y1 = ty * A_TH1;
    if (A_TH1 % 4 == 0 && !(MST_ABLATE_A & 64)) {
        constexpr int R = A_TH1 / 4;
        if (tid < 4 * A_TW1) {
            const int g = tid / A_TW1, j = tid - g * A_TW1;
            uint32_t hbr[2 * R + 3], hg[2 * R + 3];
#pragma unroll
            for (int r = 0; r < 2 * R + 3; ++r) {
                const uint32_t* p = &s_l0[2 * R * g + r][2 * j];
                const uint2 p01 = *reinterpret_cast<const uint2*>(p);      // 8-byte aligned: 2j is even
                const uint2 p23 = *reinterpret_cast<const uint2*>(p + 2);
                tap5(p01.x, p01.y, p23.x, p23.y, p[4], hbr[r], hg[r]);
            }
#pragma unroll
            for (int o = 0; o < R; ++o) {
                const int i = R * g + o;
                if (x1 + j >= lw1 || y1 + i >= lh1) continue;
                const uint32_t vbr = hbr[2 * o] + hbr[2 * o + 4] + 4u * (hbr[2 * o + 1] + hbr[2 * o + 3]) + 6u * hbr[2 * o + 2];
                const uint32_t vg = hg[2 * o] + hg[2 * o + 4] + 4u * (hg[2 * o + 1] + hg[2 * o + 3]) + 6u * hg[2 * o + 2];
                g1out[__umul24(y1 + i, lw1) + x1 + j] = pyrdown_finish(vbr, vg);
            }
        }
        MST_STAMP(0, 4);
        return;
    }
    for (int idx = tid; idx < A_TH1 * A_TW1; idx += 256) {
        const int i = idx / A_TW1, j = idx - i * A_TW1;
        if (x1 + j >= lw1 || y1 + i >= lh1) continue;
        uint32_t vbr = 0, vg = 0;
#pragma unroll
        for (int ky = 0; ky < 5; ++ky) {
            const uint32_t* r = &s_l0[2 * i + ky][2 * j];
            const uint2 p01 = *reinterpret_cast<const uint2*>(r);          // 8-byte aligned: 2j is even
            const uint2 p23 = *reinterpret_cast<const uint2*>(r + 2);
            const uint32_t p4 = r[4];
            uint32_t hbr, hg;
            tap5(p01.x, p01.y, p23.x, p23.y, p4, hbr, hg);
            const uint32_t wy = (ky == 2) ? 6u : ((ky == 1 || ky == 3) ? 4u : 1u);
            vbr += wy * hbr;
            vg += wy * hg;
        }
        g1out[__umul24(y1 + i, lw1) + x1 + j] = pyrdown_finish(vbr, vg);
    }
}

// ======================================================================================================
// A, staged form: the tile's source window through LDS (see kernels_compose.hip / tools/ubench/sampler.hip)
// ======================================================================================================
// The gathering kernel above asks the texture path for two 12-byte windows per pixel and writes level 0 twice (LDS,
// then a store pass); measured against the micro-benchmark's forms it sits at the rate of its memory pipeline, not of
// its instructions.  Here the 128 x 36 tile's source window is staged once (coalesced 12-byte loads of four pixels ->
// 4-byte pixels in LDS) and a lane samples four pixels of A2_ROWS rows out of LDS (one 16-byte table load and one
// 16-byte gain load per row: the exposure gain of every tile pixel is tabulated per calibration like the map, so the
// gain rows / weights of the per-tile constant block are gone).
//
// Lane layout of a tile row (round 4): lane l of a 32-lane row group owns the tile columns l, l + 32, l + 64, l + 96.
// The taps of one sampling step are then 32 neighbouring source pixels -- consecutive LDS dwords, one per bank --
// where round 3's "four adjacent pixels per lane" put the 32 lanes of a ds_read at a stride of four dwords: eight
// banks, four lanes each (SQ_LDS_BANK_CONFLICT was 51 % of the kernel's LDS-active cycles).  The tables are stored in
// lane order (slot = lane * 4 + k).  Level 0 no longer leaves from the registers -- a lane's four pixels are 32 columns
// apart -- but from the LDS tile the pyrDown needs anyway: the tile is dropped over the dead window (one ds_write_b32
// per pixel, conflict-free), and after the barrier a pass of 16-byte LDS reads / 16-byte non-temporal stores writes
// the interior while the other waves start on the pyrDown.
// Table entry: byte offset of the 2 x 2 window's first pixel inside the staged window | fx << 17 | fy << 23 | 1 << 31
// (the window's row pitch is a per-tile constant, so the offset is final at calibration time: one v_and per pixel
// where round 3 unpacked dx, dy and multiplied).
#ifndef MST_A2_ROWS
#define MST_A2_ROWS 3
#endif
#ifndef MST_ABLATE_A2
#define MST_ABLATE_A2 0    // timing experiments only (wrong pixels): 1 no sampling arithmetic, 2 no level-0 stores, 4 no pyrDown, 8 no table / gain loads, 16 no gain loads, 32 no table loads
#endif
constexpr int A2_ROWS = MST_A2_ROWS, A2_RG = A_LH / A2_ROWS, A2_NT = 32 * A2_RG;      // 3 rows per lane: 12 row groups, 384 threads (2 rows / 576 threads: 230 vs 200 us per 8 frame sets)
static_assert(A_LW == 128 && A_LH % A2_ROWS == 0 && A_TW0 == 124, "lane layout of the staged warp kernel");
constexpr uint32_t kWinGeneral = 1u;
// LDS image of the level-0 tile: tile column c of row r at dword r * A2_TPITCH + 2 + c, so that the interior's groups of
// four (columns 2 + 4 g ..) are 16-byte aligned for the store pass and the pyrDown's pixel pairs (columns 2 j, 2 j + 1)
// 8-byte aligned
constexpr int A2_TPITCH = 132, A2_TOFF = 2;
constexpr size_t A2_TILE_LDS = static_cast<size_t>(A_LH) * A2_TPITCH * sizeof(uint32_t);
constexpr uint32_t kA2MaxWindowBytes = 1u << 17;     // the entry's offset field
#ifndef MST_A2_PITCH_ALIGN
#define MST_A2_PITCH_ALIGN 1      // groups of four pixels; 1 = rows back to back.  16 (64 dwords: a column keeps its bank on every row) was
#endif                            // measured in round 5: conflicts 0.38 -> 0.33, no time -- the replays are the 37-column span of a tap read, not its row steps
constexpr int A2_PITCH_ALIGN = MST_A2_PITCH_ALIGN;

__device__ __forceinline__ int a2_slot_of_col(int col)        // position of tile column `col` in a table row
{
    return (col & 31) * 4 + (col >> 5);
}

// Per calibration: map, gain and window of every launched tile in the staged kernel's order.  One block per tile.
template <int KIND>
__global__ __launch_bounds__(256) void k_warp_table_lds(const CamDesc* __restrict__ cams, TileList list, uint32_t* __restrict__ table,
                                                        float* __restrict__ gain, WarpWin* __restrict__ wins, int lds_budget_px)
{
    __shared__ int s_min[2], s_max[2];
    __shared__ uint32_t s_general;
    const uint32_t t = list.tiles[blockIdx.x];
    const int cam = (t >> 24) & 0x7F, ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
    const CamDesc& c = cams[cam];
    const int bw = c.lw[0], bh = c.lh[0];
    const int lx0 = tx * A_TW0 - 2, ly0 = ty * A_TH0 - 2;
    const int tid = threadIdx.x;
    if (tid == 0) { s_min[0] = INT_MAX; s_min[1] = INT_MAX; s_max[0] = INT_MIN; s_max[1] = INT_MIN; s_general = 0; }
    __syncthreads();
    float kr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) kr[i] = c.kr[i];
    const int col = tid & (A_LW - 1), half = tid >> 7;
    // pyrDown's REFLECT_101 into the bordered image, then copyMakeBorder's REFLECT into the roi
    const int rx = reflect<0>(reflect<1>(lx0 + col, bw) - c.left, c.roi_w);
    const ColEntry ce = load_global(c.cols + rx);
    int x_lo = INT_MAX, y_lo = INT_MAX, x_hi = INT_MIN, y_hi = INT_MIN;
    bool general = false;
    for (int it = 0; it < A_ROWS_PER_THREAD; ++it) {
        const int row = half * A_ROWS_PER_THREAD + it;
        const int ry = reflect<0>(reflect<1>(ly0 + row, bh) - c.top, c.roi_h);
        const MapPx m = map_px<KIND>(c, kr, c.src_w, c.src_h, ce, load_global(c.rows + ry), rx, ry);
        if (m.inside) { x_lo = min(x_lo, m.ix); y_lo = min(y_lo, m.iy); x_hi = max(x_hi, m.ix + 1); y_hi = max(y_hi, m.iy + 1); }
        else general = true;
    }
    if (x_lo != INT_MAX) { atomicMin(&s_min[0], x_lo); atomicMin(&s_min[1], y_lo); atomicMax(&s_max[0], x_hi); atomicMax(&s_max[1], y_hi); }
    __syncthreads();
    const int bx = s_min[0] == INT_MAX ? 0 : s_min[0], by = s_min[1] == INT_MAX ? 0 : s_min[1];
    const int wx0 = bx & ~3;
    const int groups = s_max[0] == INT_MIN ? 0 : (s_max[0] - wx0 + 4) / 4;
    const int rows = s_max[1] == INT_MIN ? 0 : s_max[1] - by + 1;
    // LDS row pitch: a multiple of A2_PITCH_ALIGN groups (1: rows back to back).  Round 5 tried 16 -- a column then keeps its bank
    // on every window row, so the row steps of a slanted canvas row cannot put two lanes of a tap read on one bank -- and measured
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.38 -> 0.33 and no time at all: what replays is the 37-column span of 32 lanes
    // at 1.17 source pixels per canvas pixel, as round 4 read it (profiles/r05_experiments.md).
    const int pitch_g = (groups + A2_PITCH_ALIGN - 1) / A2_PITCH_ALIGN * A2_PITCH_ALIGN;
    const uint32_t pitch_b = static_cast<uint32_t>(pitch_g) * 16u;
    const bool addressable = static_cast<uint64_t>(pitch_b) * static_cast<uint32_t>(rows) <= kA2MaxWindowBytes;
    const size_t base = static_cast<size_t>(list.first + blockIdx.x) * (A_LH * A_LW);
    for (int it = 0; it < A_ROWS_PER_THREAD; ++it) {
        const int row = half * A_ROWS_PER_THREAD + it;
        const int ry = reflect<0>(reflect<1>(ly0 + row, bh) - c.top, c.roi_h);
        const MapPx m = map_px<KIND>(c, kr, c.src_w, c.src_h, ce, load_global(c.rows + ry), rx, ry);
        const float g = c.expos == 0 ? 1.f : gain_at(c, rx, ry);
        uint32_t e = 0;
        if (m.inside && addressable && !(g >= 8.0e6f)) {
            const uint32_t off = static_cast<uint32_t>(m.iy - by) * pitch_b + static_cast<uint32_t>(m.ix - wx0) * 4u;
            e = off | static_cast<uint32_t>(m.ifx) << 17 | static_cast<uint32_t>(m.ify) << 23 | 1u << 31;
        } else {
            general = true;
        }
        const size_t o = base + row * A_LW + a2_slot_of_col(col);
        table[o] = e;
        gain[o] = g;
    }
    if (general) atomicOr(&s_general, 1u);
    __syncthreads();
    if (tid == 0) {
        WarpWin w;
        w.wx0 = wx0; w.by = by;
        w.groups = groups;
        w.rows = rows;
        w.pitch = pitch_g;
        const bool overrun = w.wx0 + 4 * w.groups > c.src_w && by + w.rows >= c.src_h;
        w.flags = (s_general || pitch_g * 4 * w.rows > lds_budget_px || w.groups > 256 || w.groups == 0 || overrun) ? kWinGeneral : 0u;
        w.tile = 0; w.index = 0;
        wins[list.first + blockIdx.x] = w;
    }
}

__global__ __launch_bounds__(A2_NT) void k_warp_pyr_lds(const CamDesc* __restrict__ cams_all, int n_tiles, FrameBatch batch,
                                                        const uint32_t* __restrict__ table, const float* __restrict__ gain,
                                                        const WarpWin* __restrict__ wins, int n_cams, int n_sets)
{
    MST_STAMP(0, 0);
    extern __shared__ __attribute__((aligned(16))) uint32_t s_mem[];     // the staged window, later the level-0 tile [A_LH][A2_TPITCH]
    const int tid = threadIdx.x, lane = tid & 31;
    const int rgrp = __builtin_amdgcn_readfirstlane(tid >> 6) * 2 + ((tid >> 5) & 1);
    const int pair = xcd_run_index(blockIdx.x, n_tiles * n_sets);
    const int set = pair % n_sets, li = pair / n_sets;
    const WarpWin w = wins[li];                     // (launch order: window, tile word and table index in one load)
    const int gi = w.index;
    const uint32_t t = w.tile;
    const int cam = (t >> 24) & 0x7F, ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
    const CamDesc* __restrict__ cams = cams_all + set * n_cams;
    const CamDesc& c = cams[cam];
    const FrameArgs& frames = batch.f[set];
    const uint32_t stride = frames.stride[cam];
    const int bw = c.lw[0], bh = c.lh[0], nlev = c.nlev;
    // table and gain rows of this lane: issued before the staging loads, one round trip covers both
    const size_t tbl = static_cast<size_t>(gi) * (A_LH * A_LW) + (rgrp * A2_ROWS) * A_LW + lane * 4;
    u32x4_t e_r[A2_ROWS];
    typedef float __attribute__((ext_vector_type(4))) f32x4_t;
    f32x4_t g_r[A2_ROWS];
#pragma unroll
    for (int it = 0; it < A2_ROWS; ++it) {
        if (MST_ABLATE_A2 & 8) {         // (timing experiments: no table / gain streams)
            e_r[it] = u32x4_t{static_cast<uint32_t>(lane * 4 + 16 * it) | 5u << 17 | 9u << 23, static_cast<uint32_t>(lane * 4 + 4) | 7u << 17, static_cast<uint32_t>(lane * 4 + 8) | 3u << 23, static_cast<uint32_t>(lane * 4 + 12)};
            g_r[it] = f32x4_t{1.01f, 0.99f, 1.02f, 1.f};
            continue;
        }
        if (MST_ABLATE_A2 & 32) e_r[it] = u32x4_t{static_cast<uint32_t>(lane * 4 + 16 * it) | 5u << 17 | 9u << 23, static_cast<uint32_t>(lane * 4 + 4) | 7u << 17, static_cast<uint32_t>(lane * 4 + 8) | 3u << 23, static_cast<uint32_t>(lane * 4 + 12)};
        else e_r[it] = *(const u32x4_t MST_GLOBAL*)(table + tbl + it * A_LW);
        if (MST_ABLATE_A2 & 16) g_r[it] = f32x4_t{1.01f, 0.99f, 1.02f, 1.f};
        else g_r[it] = *(const f32x4_t MST_GLOBAL*)(gain + tbl + it * A_LW);
    }
    stage_window<A2_NT, false, MST_STAGE_U_WARP>(s_mem, gp(frames.ptr[cam]) + (static_cast<size_t>(w.by) * stride + static_cast<size_t>(w.wx0) * 3u), stride,
                        w.groups, w.rows, 0, 0, 0, 0, w.pitch);
    MST_STAMP(0, 1);
    __syncthreads();
    MST_STAMP(0, 2);
    const uint32_t pitch_b = static_cast<uint32_t>(w.pitch) * 16u;
    const char* win = reinterpret_cast<const char*>(s_mem);
    uint32_t px[A2_ROWS][4];
#pragma unroll
    for (int it = 0; it < A2_ROWS; ++it)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = e_r[it][k];
            const uint32_t addr = e & (kA2MaxWindowBytes - 1u);
            const uint32_t* p0 = reinterpret_cast<const uint32_t*>(win + addr);
            const uint32_t* p1 = reinterpret_cast<const uint32_t*>(win + addr + pitch_b);
            if (MST_ABLATE_A2 & 1) px[it][k] = p0[0] ^ p1[1] ^ e;          // (timing experiments: taps without the arithmetic)
            else px[it][k] = sample_taps(p0[0], p0[1], p1[0], p1[1], (e >> 17) & 63u, (e >> 23) & 63u, g_r[it][k]);
        }
    MST_STAMP(0, 3);
    __syncthreads();                                // every wave is done with the window: the tile takes its place
#pragma unroll
    for (int it = 0; it < A2_ROWS; ++it) {
        uint32_t* trow = s_mem + (rgrp * A2_ROWS + it) * A2_TPITCH + A2_TOFF + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) trow[32 * k] = px[it][k];
    }
    __syncthreads();
    MST_STAMP(0, 4);
    // level 0, interior of the tile: 31 groups of four pixels per row, 16 bytes each way (16-byte aligned in global memory
    // when the level's width is a multiple of four: 124 * 4 bytes is)
    const int lx0 = tx * A_TW0 - 2, ly0 = ty * A_TH0 - 2;
    const bool store0 = !(t >> 31) && !(MST_ABLATE_A2 & 2);     // (bit 31: nothing reads this tile's level 0 from memory, see mb_plan.cpp)
    if (store0) {
        // packed BGR8, twelve bytes per group of four pixels (gx is a multiple of four and the row pitch a multiple of twelve
        // that covers the last group whole: no partial store)
        uint8_t MST_GLOBAL* __restrict__ g0out = (uint8_t MST_GLOBAL*)c.G[0];
        const uint32_t pitch0 = static_cast<uint32_t>(c.l0_pitch);
        constexpr int kGroups = A_TW0 / 4;
#pragma unroll
        for (int i0 = 0; i0 < A_TH0 * kGroups; i0 += A2_NT) {
            const int i = i0 + tid;
            const int r = static_cast<int>((static_cast<uint32_t>(i) * 2115u) >> 16);       // i / 31 for i < 1984
            const int g = i - r * kGroups;
            const int gy = ly0 + 2 + r, gx = lx0 + 2 + 4 * g;
            if (i >= A_TH0 * kGroups || gy >= bh || gx >= bw) continue;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(s_mem + (2 + r) * A2_TPITCH + A2_TOFF + 2 + 4 * g);
            l0_store4(g0out, pitch0, gx, gy, v.x, v.y, v.z, v.w);
        }
    }
    if (nlev < 2 || (MST_ABLATE_A2 & 4)) return;
    MST_STAMP(0, 5);
    // pyrDown straight from the LDS tile (as in k_warp_pyr): a thread owns one output column and four output rows
    uint32_t MST_GLOBAL* __restrict__ g1out = gp(c.G[1]);
    const int lw1 = c.lw[1], lh1 = c.lh[1];
    const int x1 = tx * A_TW1, y1 = ty * A_TH1;
    static_assert(A_TH1 % 4 == 0, "pyrDown rows per thread");
    constexpr int R = A_TH1 / 4;
    if (tid < 4 * A_TW1) {
        const int g = tid / A_TW1, j = tid - g * A_TW1;
        uint32_t hbr[2 * R + 3], hg[2 * R + 3];
#pragma unroll
        for (int r = 0; r < 2 * R + 3; ++r) {
            const uint32_t* p = s_mem + (2 * R * g + r) * A2_TPITCH + A2_TOFF + 2 * j;
            const uint2 p01 = *reinterpret_cast<const uint2*>(p);
            const uint2 p23 = *reinterpret_cast<const uint2*>(p + 2);
            tap5(p01.x, p01.y, p23.x, p23.y, p[4], hbr[r], hg[r]);
        }
#pragma unroll
        for (int o = 0; o < R; ++o) {
            const int i = R * g + o;
            if (x1 + j >= lw1 || y1 + i >= lh1) continue;
            const uint32_t vbr = hbr[2 * o] + hbr[2 * o + 4] + 4u * (hbr[2 * o + 1] + hbr[2 * o + 3]) + 6u * hbr[2 * o + 2];
            const uint32_t vg = hg[2 * o] + hg[2 * o + 4] + 4u * (hg[2 * o + 1] + hg[2 * o + 3]) + 6u * hg[2 * o + 2];
            g1out[__umul24(y1 + i, lw1) + x1 + j] = pyrdown_finish(vbr, vg);
        }
    }
    MST_STAMP(0, 6);
    MST_STAMP_VAL(0, 8, static_cast<unsigned long long>(w.pitch) * 16u * w.rows);
}

// ======================================================================================================
// B: pyrDown chain level b -> b+1 .. b+k (k <= B_KMAX) inside LDS
// ======================================================================================================
// Measured on MI355X (C3): a 4-deep chain per block needs 62 KB of LDS and serialises four barrier-separated
// phases in too few resident blocks (36-67 us); 2-deep launches it is.  These launches are latency-bound (load ->
// barrier -> reduce -> barrier -> reduce), so what helps is many small blocks in different phases on a CU: 32 x 32
// tiles with 256 threads (8 blocks per CU) run 8.6 us per batched frame set against 9.4 for 64 x 64 / 1024
// threads and 14.9 for 128 x 128, although the smaller tile re-reads more halo (41^2 loads for 32^2 pixels).
#ifndef MST_B_T
#define MST_B_T 32
#endif
constexpr int B_T = MST_B_T;                    // base-level interior of a tile (a power of two >= 16)
#ifndef MST_B_THREADS
#define MST_B_THREADS 256
#endif
constexpr int B_THREADS = MST_B_THREADS;
constexpr int B_KMAX = 2;
// region extents of a full k = 2 chain: e2 = B_T / 4 outputs, each level below needs 2 e + 3 inputs
constexpr int B_E1 = 2 * (B_T / 4) + 3, B_E0 = 2 * B_E1 + 3;

// LDS row pitch of a region whose rows are read by consecutive lanes at a stride of two pixels (8-byte reads),
// `en` lanes per output row (en even): with pitch == en (mod 32) the lanes that continue on the next output row
// start exactly where the previous row's lanes stopped (mod the 64 banks), so a wave's reads tile the banks without
// a conflict; any other pitch puts two lanes on a bank for part of every row pair.  The pitch stays even, which keeps
// every row's pixel pairs 8-byte aligned.
constexpr int chain_pitch(int es, int en)
{
    int p = (es / 32) * 32 + (en % 32);
    return p >= es ? p : p + 32;
}
constexpr int even_up(int v) { return (v + 1) & ~1; }

template <int K>
__global__ __launch_bounds__(B_THREADS) void k_pyr_chain(const CamDesc* __restrict__ cams_all, TileList list, int base,
                                                         int n_cams)
{
    const CamDesc* __restrict__ cams = cams_all + blockIdx.y * n_cams;
    constexpr int k = K;
    // extents e[j] (compile-time) of the regions each level covers, and their LDS pitches
    constexpr int e[B_KMAX + 1] = {K == 2 ? B_E0 : 2 * (B_T / 2) + 3, K == 2 ? B_E1 : B_T / 2, B_T / 4};
    constexpr int pitch[B_KMAX] = {chain_pitch(e[0], even_up(e[1])), chain_pitch(e[1], even_up(e[2]))};
    static_assert(pitch[0] % 2 == 0 && pitch[1] % 2 == 0, "8-byte aligned pixel pairs need an even pitch");
    // regions of levels base and base+1, back to back
    constexpr int kOff1 = (pitch[0] * e[0] + 3) & ~3;
    __shared__ __attribute__((aligned(16))) uint32_t s_all[kOff1 + pitch[1] * e[1]];
    uint32_t* const s_e0 = s_all;

    const uint32_t t = list.tiles[xcd_run_index(blockIdx.x, list.count)];
    const int cam = (t >> 24) & 0x7F, ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
    const CamDesc& c = cams[cam];
    const int tid = threadIdx.x;

    // origins lo[j] (in level base+j coordinates) of the regions
    int lox[B_KMAX + 1], loy[B_KMAX + 1];
    lox[k] = (tx * B_T) >> k;
    loy[k] = (ty * B_T) >> k;
#pragma unroll
    for (int j = k - 1; j >= 0; --j) {
        lox[j] = 2 * lox[j + 1] - 2;
        loy[j] = 2 * loy[j + 1] - 2;
    }
    constexpr int offs[B_KMAX] = {0, kOff1};

    // load the base region (coordinates outside the image are never referenced: taps are reflected first)
    {
        const int w = c.lw[base], h = c.lh[base];
        const uint32_t MST_GLOBAL* __restrict__ g = gp(c.G[base]);
        constexpr int n0 = e[0] * e[0];
        constexpr int kBatch = (n0 + B_THREADS - 1) / B_THREADS;       // 6 independent loads per thread
        uint32_t v[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int idx = tid + u * B_THREADS;
            const int yy = idx / e[0], xx = idx - yy * e[0];
            const int gx = lox[0] + xx, gy = loy[0] + yy;
            v[u] = 0;
            if (idx < n0 && static_cast<unsigned>(gx) < static_cast<unsigned>(w) &&
                static_cast<unsigned>(gy) < static_cast<unsigned>(h))
                v[u] = g[__umul24(gy, w) + gx];
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int idx = tid + u * B_THREADS;
            const int yy = idx / e[0], xx = idx - yy * e[0];
            if (idx < n0) s_e0[yy * pitch[0] + xx] = v[u];
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < k; ++j) {
        const int sw = c.lw[base + j], sh = c.lh[base + j];          // source level size
        const int dw = c.lw[base + j + 1], dh = c.lh[base + j + 1];  // destination level size
        const uint32_t* __restrict__ s = s_all + offs[j];
        uint32_t* __restrict__ d = (j + 1 < B_KMAX) ? s_all + offs[j + 1] : nullptr;
        uint32_t MST_GLOBAL* __restrict__ gout = gp(c.G[base + j + 1]);
        const int en = e[j + 1], es = pitch[j];
        const int enp = even_up(en);                                 // lanes per output row (see chain_pitch)
        // interior of this tile at the destination level
        const int ix0 = (tx * B_T) >> (j + 1), iy0 = (ty * B_T) >> (j + 1), isz = B_T >> (j + 1);
        for (int idx = tid; idx < enp * en; idx += B_THREADS) {
            const int yy = idx / enp, xx = idx - yy * enp;
            const int ox = lox[j + 1] + xx, oy = loy[j + 1] + yy;    // destination coordinates
            if (xx >= en || static_cast<unsigned>(ox) >= static_cast<unsigned>(dw) ||
                static_cast<unsigned>(oy) >= static_cast<unsigned>(dh))
                continue;
            uint32_t vbr = 0, vg = 0;
            // lox[j] == 2*lox[j+1] - 2, so away from the image border tap q of output xx is column 2*xx + q
            const bool interior = 2 * ox - 2 >= 0 && 2 * ox + 2 < sw && 2 * oy - 2 >= 0 && 2 * oy + 2 < sh;
            if (interior) {
                const uint32_t* r = s + (2 * yy) * es + 2 * xx;          // even dword offset: 8-byte aligned pairs
#pragma unroll
                for (int ky = 0; ky < 5; ++ky, r += es) {
                    uint32_t hbr, hg;
                    const uint2 p01 = *reinterpret_cast<const uint2*>(r);
                    const uint2 p23 = *reinterpret_cast<const uint2*>(r + 2);
                    tap5(p01.x, p01.y, p23.x, p23.y, r[4], hbr, hg);
                    const uint32_t wy = (ky == 2) ? 6u : ((ky == 1 || ky == 3) ? 4u : 1u);
                    vbr += wy * hbr;
                    vg += wy * hg;
                }
            } else {
                int xi[5], yi[5];
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    xi[q] = reflect<1>(2 * ox + q - 2, sw) - lox[j];
                    yi[q] = reflect<1>(2 * oy + q - 2, sh) - loy[j];
                }
#pragma unroll
                for (int ky = 0; ky < 5; ++ky) {
                    const uint32_t* r = s + __mul24(yi[ky], es);
                    uint32_t hbr, hg;
                    tap5(r[xi[0]], r[xi[1]], r[xi[2]], r[xi[3]], r[xi[4]], hbr, hg);
                    const uint32_t wy = (ky == 2) ? 6u : ((ky == 1 || ky == 3) ? 4u : 1u);
                    vbr += wy * hbr;
                    vg += wy * hg;
                }
            }
            const uint32_t v = pyrdown_finish(vbr, vg);
            if (d && j + 1 < k) d[yy * pitch[(j + 1 < B_KMAX) ? j + 1 : 0] + xx] = v;
            if (ox >= ix0 && ox < ix0 + isz && oy >= iy0 && oy < iy0 + isz)
                gout[__umul24(oy, dw) + ox] = v;
        }
        __syncthreads();
    }
}

// ======================================================================================================
// B, streaming form: one cv::pyrDown level l -> l + 1 without LDS (round 4)
// ======================================================================================================
// k_pyr_chain spends 264 VALU lane-instructions per output pixel (index arithmetic of its LDS regions, the halo every
// chained level recomputes, per-element divisions) and three barrier-separated phases per block; on the two large
// levels (1 -> 2 -> 3: 58 of the 67 us the chain launches take per 8 frame sets on C3) that is the whole cost.  Here a
// wave owns PS_W = 62 output columns x PS_R output rows: lane i holds the pixel pair (columns 2 ox, 2 ox + 1: 8 bytes,
// the wave 512 contiguous bytes per source row) of output column ox = 62 ux + i - 1, so lanes 1..62 produce outputs and
// lanes 0 and 63 only carry the pairs beside them; the pair on a lane's left and the pixel on its right come over
// wave-wide DPP shifts (wave_shr:1 / wave_shl:1).  All 2 R + 3 source rows of the unit are requested before the first is
// used (one basic block, no branch: the compiler counts vmcnt down row by row), every row is filtered horizontally
// once and the last five filtered rows are kept in registers.  Source levels have even sizes (the bordered level 0 is a
// multiple of 2^bands), so REFLECT_101 only ever folds column / row -2, -1 and `size` back inside.
constexpr int PS_R = 8, PS_W = 62;
__device__ __forceinline__ uint32_t wave_from_below(uint32_t v)      // lane i <- lane i - 1 (lane 0 keeps its own)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x138, 0xF, 0xF, false));
}
__device__ __forceinline__ uint32_t wave_from_above(uint32_t v)      // lane i <- lane i + 1 (lane 63 keeps its own)
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(static_cast<int>(v), static_cast<int>(v), 0x130, 0xF, 0xF, false));
}
template <int R>
__global__ __launch_bounds__(256) void k_pyrdown_stream(const CamDesc* __restrict__ cams_all, const uint32_t* __restrict__ units,
                                                        int n_units, int l, int n_cams)
{
    const CamDesc* __restrict__ cams = cams_all + blockIdx.y * n_cams;
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int ui = xcd_run_index(blockIdx.x, (n_units + 3) >> 2) * 4 + wave;
    if (ui >= n_units) return;                                   // (wave-uniform)
    const uint32_t u = units[ui];
    const int cam = (u >> 24) & 0x7F, uy = (u >> 12) & 0xFFF, ux = u & 0xFFF;
    const CamDesc& c = cams[cam];
    const int sw = c.lw[l], sh = c.lh[l], dw = c.lw[l + 1], dh = c.lh[l + 1];
    uint32_t MST_GLOBAL* __restrict__ dst = gp(c.G[l + 1]);
    const int ox = ux * PS_W + lane - 1, oy0 = uy * R;
    const int pc = min(max(ox, 0), dw - 1);                      // this lane's pixel pair (lanes outside the level: the nearest one)
    const bool left_border = ox <= 0, right_border = ox >= dw - 1;
    const uint32_t MST_GLOBAL* __restrict__ col = gp(c.G[l]) + 2 * pc;
    struct Row { uint32_t hbr, hg; };
    auto row_of = [&](int iy) {                                  // REFLECT_101, -2 <= iy <= sh (scalar)
        const int a = iy < 0 ? -iy : iy;
        return a >= sh ? 2 * sh - 2 - a : a;
    };
    // (the DPP shifts first, on every lane: see from_lane_below)
    auto filter = [&](const uint2& p, Row& h) {
        const uint32_t bx = wave_from_below(p.x), by = wave_from_below(p.y), ax = wave_from_above(p.x);
        const uint32_t lx = left_border ? ax : bx, ly = left_border ? p.y : by;      // columns -2, -1 -> 2, 1
        const uint32_t rx = right_border ? p.x : ax;                                 // column sw -> sw - 2
        tap5(lx, ly, p.x, p.y, rx, h.hbr, h.hg);
    };
    constexpr int kRows = 2 * R + 3;
    uint2 raw[kRows];
#pragma unroll
    for (int t = 0; t < kRows; ++t) {
        const GPix v = *(const GPix MST_GLOBAL*)(col + __umul24(static_cast<uint32_t>(row_of(min(2 * oy0 + t - 2, sh))), static_cast<uint32_t>(sw)));
        raw[t] = make_uint2(v.x, v.y);
    }
    Row h[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) filter(raw[t], h[t]);
    const bool col_ok = lane >= 1 && lane <= PS_W && ox < dw;
    uint32_t MST_GLOBAL* o = dst + __umul24(oy0, dw) + max(ox, 0);
#pragma unroll
    for (int t = 0; t < R; ++t) {
        const uint32_t vbr = h[0].hbr + h[4].hbr + 4u * (h[1].hbr + h[3].hbr) + 6u * h[2].hbr;
        const uint32_t vg = h[0].hg + h[4].hg + 4u * (h[1].hg + h[3].hg) + 6u * h[2].hg;
        if (col_ok && oy0 + t < dh) *o = pyrdown_finish(vbr, vg);
        o += dw;
        if (t + 1 < R) {
            h[0] = h[2]; h[1] = h[3]; h[2] = h[4];
            filter(raw[2 * t + 5], h[3]); filter(raw[2 * t + 6], h[4]);
        }
    }
}

// ======================================================================================================
// C: the canvas -- k_blend_coarse (levels 1 .. top of a level-1 tile in one block) and k_blend_fine (level 0)
// ======================================================================================================
// Tile heights, C3 on MI355X, canvas stage per frame set (8 sets per launch) / one-set-at-a-time panoramas per s:
//   fine 64, coarse 64: 33.1 us / 7.5 k      fine 64, coarse 32: 32.5 us / 8.5 k      fine 32, coarse 32: 30.3 us / 8.8 k
//   fine 32, coarse 16: 32.4 us / 9.5 k      fine 16, coarse 16: 35.9 us / 9.4 k
#ifndef MST_C_TH
#define MST_C_TH 32        // k_blend_fine: 128 x 32 panorama tiles
#endif
constexpr int C_TW = 128, C_TH = MST_C_TH;
#ifndef MST_CC_TW
#define MST_CC_TW 128      // k_blend_coarse: tile size in level-1 pixels
#endif
constexpr int CC_TW = MST_CC_TW;
// k_blend_coarse tile height (level-1 rows) is chosen per launch (BlendArgs::cc_th; both tile tables are planned at
// calibration): 32 for batched launches, 16 for one or two frame sets (four times as many, shorter blocks: 9.5 k instead
// of 8.8 k panoramas/s one at a time, 3 % slower in batches of 8)
constexpr int CC_TH_MAX = 32;
// k_blend_fine block size, chosen per launch like the coarse tile height (BlendArgs::fine_threads): 128 for batched
// launches -- two waves of eight quad rows, a wave re-reads 2 source rows in 10 instead of 4 in 12, which shows once the
// level-0 loads stream past the caches (157.6 -> 147.7 us per 8 frame sets) -- and 256 (four waves of four rows: half
// the dependent steps per block) when frame sets are composed one at a time
#ifndef MST_CC_THREADS
#define MST_CC_THREADS 256
#endif
constexpr int CC_THREADS = MST_CC_THREADS;

struct Region { int x0, y0, x1, y1; };

// The canvas is cut into C_TW x C_TH tiles at level `base` (0: the panorama tiles of k_blend_fine, 1: the level-1
// tiles of k_blend_coarse); a tile's region at a coarser level is what pyrUp needs of that level to rebuild the tile.
__host__ __device__ inline Region blend_region(const BlendArgs& a, int base, int tx, int ty, int level)
{
    Region r;
    auto lo = [](int v, int m) { return v < m ? v : m; };
    auto hi = [](int v, int m) { return v > m ? v : m; };
    const int tw = base ? CC_TW : C_TW, th = base ? a.cc_th : C_TH;
    r.x0 = tx * tw; r.y0 = ty * th;
    r.x1 = lo(r.x0 + tw, a.cw[base]); r.y1 = lo(r.y0 + th, a.ch[base]);
    if (base == 1 && a.clip_x1 > 0) {          // a strip engine: only the level-1 columns its panorama tiles read
        r.x0 = hi(r.x0, a.clip_x0);
        r.x1 = hi(r.x0, lo(r.x1, a.clip_x1));
    }
    for (int l = base + 1; l <= level; ++l) {
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

// pyrUp of a packed (unsigned) Gaussian level from its 3x3 source neighbourhood nb[row][col]: four fine
// outputs per channel pair, already (v + 32) >> 6.  Quad order: (x even,y even), (x odd,y even),
// (x even,y odd), (x odd,y odd).
__device__ __forceinline__ void pyrup_quad_packed(const uint32_t nb[3][3], uint32_t obr[4], uint32_t og[4])
{
    uint32_t hebr[3], hobr[3], heg[3], hog[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint32_t a = nb[r][0], b = nb[r][1], c = nb[r][2];
        hebr[r] = lanes_br(a) + 6u * lanes_br(b) + lanes_br(c);
        hobr[r] = 4u * (lanes_br(b) + lanes_br(c));
        heg[r] = lane_g(a) + 6u * lane_g(b) + lane_g(c);
        hog[r] = 4u * (lane_g(b) + lane_g(c));
    }
    obr[0] = hebr[0] + 6u * hebr[1] + hebr[2];  og[0] = heg[0] + 6u * heg[1] + heg[2];
    obr[1] = hobr[0] + 6u * hobr[1] + hobr[2];  og[1] = hog[0] + 6u * hog[1] + hog[2];
    obr[2] = 4u * (hebr[1] + hebr[2]);          og[2] = 4u * (heg[1] + heg[2]);
    obr[3] = 4u * (hobr[1] + hobr[2]);          og[3] = 4u * (hog[1] + hog[2]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        obr[q] = ((obr[q] + 0x00200020u) >> 6) & 0x03FF03FFu;
        og[q] = (og[q] + 32u) >> 6;
    }
}

__device__ __forceinline__ int s16_lo(uint32_t v) { return static_cast<int>(static_cast<short>(v & 0xFFFFu)); }
__device__ __forceinline__ int s16_hi(uint32_t v) { return static_cast<int>(v) >> 16; }
__device__ __forceinline__ uint2 pack3(int c0, int c1, int c2)
{
    return make_uint2((static_cast<uint32_t>(c0) & 0xFFFFu) | (static_cast<uint32_t>(c1) << 16),
                      static_cast<uint32_t>(c2) & 0xFFFFu);
}

// Two horizontally adjacent pixels of an LDS level region as one 16-byte access.  Region origins, widths and quad
// coordinates are all even and the dynamic LDS base is 16-byte aligned, so (row * width + x) * 8 is a multiple of
// 16 for every quad: one ds_write_b128 / ds_read_b128 per quad row instead of two 8-byte accesses at a 16-byte
// lane stride (which land two lanes on every bank).
__device__ __forceinline__ void lds_store_pair(uint2* s, int idx, const int a[3], const int b[3])
{
    const uint2 pa = pack3(a[0], a[1], a[2]), pb = pack3(b[0], b[1], b[2]);
    *reinterpret_cast<uint4*>(s + idx) = make_uint4(pa.x, pa.y, pb.x, pb.y);
}

// pyrUp of the collapsed coarser level held in LDS (signed shorts) around coarse pixel (j, i) given in
// level coordinates; rx0/ry0/pitch describe the LDS region.  out[q][ch], already (v + 32) >> 6 and saturated.
// (the arithmetic, on a 3x3 neighbourhood already in registers)
__device__ __forceinline__ void pyrup_quad_s16(const uint2 v[3][3], int out[4][3])
{
    int he[3][3], ho[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const uint2 a = v[r][0], b = v[r][1], c = v[r][2];
        const int av[3] = {s16_lo(a.x), s16_hi(a.x), s16_lo(a.y)};
        const int bv[3] = {s16_lo(b.x), s16_hi(b.x), s16_lo(b.y)};
        const int cv[3] = {s16_lo(c.x), s16_hi(c.x), s16_lo(c.y)};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            he[r][ch] = av[ch] + 6 * bv[ch] + cv[ch];
            ho[r][ch] = 4 * (bv[ch] + cv[ch]);
        }
    }
    // (cv::pyrUp's saturate_cast<short> cannot trigger: every output is (sum of w_i * v_i + 32) >> 6 with non-negative
    // weights summing to 64 over CV_16S inputs, i.e. it lies between the smallest and the largest input)
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        out[0][ch] = (he[0][ch] + 6 * he[1][ch] + he[2][ch] + 32) >> 6;
        out[1][ch] = (ho[0][ch] + 6 * ho[1][ch] + ho[2][ch] + 32) >> 6;
        out[2][ch] = (4 * (he[1][ch] + he[2][ch]) + 32) >> 6;
        out[3][ch] = (4 * (ho[1][ch] + ho[2][ch]) + 32) >> 6;
    }
}

// (GPix: a pixel of the collapsed level-1 canvas in global memory, see the packed-pixel helpers)
typedef const GPix MST_GLOBAL* C1Ptr;
template <typename T> __device__ __forceinline__ uint2 as_u2(const T& t) { return make_uint2(t.x, t.y); }

template <typename Ptr>      // LDS (collapse inside a tile) or global memory (the collapsed level-1 canvas)
__device__ __forceinline__ void pyrup_quad_lds(Ptr s, int rx0, int ry0, int pitch, int sw, int sh,
                                               int j, int i, int out[4][3])
{
    int jm, jp, im, ip;
    up_idx(j, sw, jm, jp);
    up_idx(i, sh, im, ip);
    const int rows[3] = {im - ry0, i - ry0, ip - ry0};
    const int c0 = jm - rx0, c1 = j - rx0, c2 = jp - rx0;
    uint2 v[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const Ptr p = s + __mul24(rows[r], pitch);
        v[r][0] = as_u2(p[c0]); v[r][1] = as_u2(p[c1]); v[r][2] = as_u2(p[c2]);
    }
    pyrup_quad_s16(v, out);
}

// The normalised band of one 2x2 quad of canvas level `l` at even coordinates (x, y): the weighted sum over
// cameras of their Laplacians divided by the re-summed weights.  It does not depend on any other canvas
// level, so all levels of a tile are computed without a barrier in between; only the cheap pyrUp + add
// collapse that follows is sequential.  TOP: coarsest level (the Laplacian is the Gaussian itself and
// camera windows may start on odd coordinates); FINAL: level 0 (weights come from the masks).
//
// PURE: the tile lies strictly inside one camera's seam mask, far enough from every seam that this camera's
// weight is exactly 1.0f over the tile's region at every level and no other camera has a non-zero weight
// there (checked once per calibration, mb_plan.cpp).  Then  short(lap * 1.0f) == lap,  the weight sum is
// 1.0f, and  short(lap / (1.0f + 1e-5f)) == lap - sign(lap)  for every |lap| <= 255 (lap/1.00001f lies
// strictly between |lap|-1 and |lap|): the band needs no weight loads, no float conversion and no divide.
// W16: MultiBandBlender weight_type CV_16S (blenders.cpp:289-292,331-355,406-423): weights are the integers
// mask + (mask != 0) and their integer pyrDown (carried exactly in the float arrays), a camera adds
// short((lap * w) >> 8), and a band is normalised as short((v << 8) / (w + 1)) with C integer division.  The pure
// path is the same for both types: w == 256 gives (lap * 256) >> 8 == lap and (lap << 8) / 257 == lap - sign(lap).
template <bool TOP, bool FINAL, bool PURE, bool W16 = false>
__device__ __forceinline__ void band_quad(const CamDesc* __restrict__ cams, const BlendArgs& a, int l, int x, int y,
                                          const bool valid[4], uint32_t cam_bits, int out[4][3], float ws[4])
{
    if (PURE) {
        const CamDesc& c = cams[__builtin_ctz(cam_bits)];
        const int lw = c.lw[l];
        const int lx = x - c.ox[l], ly = y - c.oy[l];
        uint32_t g[4];
        uint32_t ubr[4] = {0, 0, 0, 0}, ug[4] = {0, 0, 0, 0};
        if (TOP) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                g[q] = !valid[q] ? 0u : FINAL ? l0_px(c, lx + (q & 1), ly + (q >> 1)) : gp(c.G[l])[__umul24(ly + (q >> 1), lw) + lx + (q & 1)];
        } else {
            const uint32_t off = __umul24(ly, lw) + lx;
            const uint2 g0 = FINAL ? l0_pair(c, lx, ly) : load_global(reinterpret_cast<const uint2*>(c.G[l] + off));
            const uint2 g1 = FINAL ? l0_pair(c, lx, ly + 1) : load_global(reinterpret_cast<const uint2*>(c.G[l] + off + lw));
            uint32_t nb[3][3];
            const int sw = c.lw[l + 1], sh = c.lh[l + 1];
            int jm, jp, im, ip;
            const int j = lx >> 1, i = ly >> 1;
            up_idx(j, sw, jm, jp);
            up_idx(i, sh, im, ip);
            const uint32_t MST_GLOBAL* __restrict__ gu = gp(c.G[l + 1]);
            const uint32_t r0 = __umul24(im, sw), r1 = __umul24(i, sw),
                           r2 = __umul24(ip, sw);
            nb[0][0] = gu[r0 + jm]; nb[0][1] = gu[r0 + j]; nb[0][2] = gu[r0 + jp];
            nb[1][0] = gu[r1 + jm]; nb[1][1] = gu[r1 + j]; nb[1][2] = gu[r1 + jp];
            nb[2][0] = gu[r2 + jm]; nb[2][1] = gu[r2 + j]; nb[2][2] = gu[r2 + jp];
            g[0] = g0.x; g[1] = g0.y; g[2] = g1.x; g[3] = g1.y;
            pyrup_quad_packed(nb, ubr, ug);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int lb = static_cast<int>(g[q] & 0xFF) - static_cast<int>(ubr[q] & 0xFFFF);
            const int lg = static_cast<int>((g[q] >> 8) & 0xFF) - static_cast<int>(ug[q]);
            const int lr = static_cast<int>((g[q] >> 16) & 0xFF) - static_cast<int>(ubr[q] >> 16);
            out[q][0] = lb - (lb > 0) + (lb < 0);
            out[q][1] = lg - (lg > 0) + (lg < 0);
            out[q][2] = lr - (lr > 0) + (lr < 0);
            ws[q] = 1.f;
        }
        return;
    }
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
        uint32_t g[4];
        uint32_t nb[3][3];
        if (TOP) {
            // windows may start on odd coordinates at the coarsest level: test pixel by pixel
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int px = lx + (q & 1), py = ly + (q >> 1);
                const bool in = valid[q] && static_cast<unsigned>(px) < static_cast<unsigned>(lw) &&
                                static_cast<unsigned>(py) < static_cast<unsigned>(lh);
                const uint32_t off = __umul24(py, lw) + px;
                w[q] = 0.f;
                g[q] = 0u;
                if (in) {
                    if (FINAL) {
                        const int m = gp(c.mask_b)[off];
                        w[q] = W16 ? static_cast<float>(m + (m != 0)) : static_cast<float>(m) * a.alpha + 0.f;
                    } else {
                        w[q] = gp(c.W[l])[off];
                    }
                    g[q] = FINAL ? l0_px(c, px, py) : gp(c.G[l])[off];
                    any = true;
                }
            }
            if (!any) continue;
        } else {
            // below the top level every window origin and size is even: the quad is inside or outside as a whole
            const bool inside = static_cast<unsigned>(lx) < static_cast<unsigned>(lw) &&
                                static_cast<unsigned>(ly) < static_cast<unsigned>(lh);
            if (!inside) continue;
            const uint32_t off = __umul24(ly, lw) + lx;
            // weights first: a camera whose four weights are zero here contributes short(v * 0.f) == 0 and its
            // Gaussian quad and pyrUp sources need not be fetched at all (in a seam tile that is about half of
            // the camera/quad pairs); the remaining loads are all issued before the first use
            if (FINAL) {
                // level-0 weight = mask.convertTo(32F, 1/255) with BORDER_CONSTANT 0 around the roi (mask_b)
                const uint32_t m01 = static_cast<uint32_t>(*(const uint16_t MST_GLOBAL*)(gp(c.mask_b) + off)) |
                                     static_cast<uint32_t>(*(const uint16_t MST_GLOBAL*)(gp(c.mask_b) + off + lw)) << 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int m = static_cast<int>((m01 >> (8 * q)) & 0xFFu);
                    w[q] = W16 ? static_cast<float>(m + (m != 0)) : static_cast<float>(m) * a.alpha + 0.f;
                }
            } else {
                const float2 w0 = load_global(reinterpret_cast<const float2*>(c.W[l] + off));
                const float2 w1 = load_global(reinterpret_cast<const float2*>(c.W[l] + off + lw));
                w[0] = w0.x; w[1] = w0.y; w[2] = w1.x; w[3] = w1.y;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (!valid[q]) w[q] = 0.f;
            if (w[0] == 0.f && w[1] == 0.f && w[2] == 0.f && w[3] == 0.f) continue;   // also adds nothing to ws
            const uint2 g0 = FINAL ? l0_pair(c, lx, ly) : load_global(reinterpret_cast<const uint2*>(c.G[l] + off));
            const uint2 g1 = FINAL ? l0_pair(c, lx, ly + 1) : load_global(reinterpret_cast<const uint2*>(c.G[l] + off + lw));
            {
                const int sw = c.lw[l + 1], sh = c.lh[l + 1];
                int jm, jp, im, ip;
                const int j = lx >> 1, i = ly >> 1;
                up_idx(j, sw, jm, jp);
                up_idx(i, sh, im, ip);
                const uint32_t MST_GLOBAL* __restrict__ gu = gp(c.G[l + 1]);
                const uint32_t r0 = __umul24(im, sw), r1 = __umul24(i, sw),
                               r2 = __umul24(ip, sw);
                nb[0][0] = gu[r0 + jm]; nb[0][1] = gu[r0 + j]; nb[0][2] = gu[r0 + jp];
                nb[1][0] = gu[r1 + jm]; nb[1][1] = gu[r1 + j]; nb[1][2] = gu[r1 + jp];
                nb[2][0] = gu[r2 + jm]; nb[2][1] = gu[r2 + j]; nb[2][2] = gu[r2 + jp];
            }
            g[0] = g0.x; g[1] = g0.y; g[2] = g1.x; g[3] = g1.y;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) ws[q] += w[q];              // dst_weight_row[x] += weight_row[x_]
        if (w[0] == 0.f && w[1] == 0.f && w[2] == 0.f && w[3] == 0.f) continue;   // short(v * 0.f) == 0

        uint32_t ubr[4] = {0, 0, 0, 0}, ug[4] = {0, 0, 0, 0};
        if (!TOP) pyrup_quad_packed(nb, ubr, ug);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // laplacian = cv::subtract(G_l, pyrUp(G_l+1)) (|.| <= 255, the saturation cannot trigger);
            // dst += static_cast<short>(src * w): |src * w| < 256, so plain truncation is the x64 conversion
            const int lb = static_cast<int>(g[q] & 0xFF) - static_cast<int>(ubr[q] & 0xFFFF);
            const int lg = static_cast<int>((g[q] >> 8) & 0xFF) - static_cast<int>(ug[q]);
            const int lr = static_cast<int>((g[q] >> 16) & 0xFF) - static_cast<int>(ubr[q] >> 16);
            if (W16) {       // dst += short((src * w) >> 8): int product, arithmetic shift, wrap to short, wrapping add
                const int wi = static_cast<int>(w[q]);
                acc[q][0] = wrap_s16(acc[q][0] + wrap_s16((lb * wi) >> 8));
                acc[q][1] = wrap_s16(acc[q][1] + wrap_s16((lg * wi) >> 8));
                acc[q][2] = wrap_s16(acc[q][2] + wrap_s16((lr * wi) >> 8));
            } else {
                acc[q][0] = wrap_s16(acc[q][0] + static_cast<int>(static_cast<float>(lb) * w[q]));
                acc[q][1] = wrap_s16(acc[q][1] + static_cast<int>(static_cast<float>(lg) * w[q]));
                acc[q][2] = wrap_s16(acc[q][2] + static_cast<int>(static_cast<float>(lr) * w[q]));
            }
        }
    }

    // normalizeUsingWeightMap: static_cast<short>(v / (w + eps)); |v / d| < 2^31 always (d >= 1e-5).
    // Where the weights sum to exactly 1.0f (most pixels away from a seam, and every covered pixel at level 0
    // with 0/255 masks) the accumulator is an integer with |v| <= 255 (non-negative weights summing to 1 times
    // Laplacians within +-255) and short(v / 1.00001f) == v - sign(v), the identity of the single-camera path;
    // where they sum to 0 the accumulator is 0 and so is the quotient.  Only the remaining pixels divide.
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        // (CV_16S weights: the floor of (lap * w) >> 8 can push a sum of several cameras past the +-257 where
        // (v << 8) / 257 == v - sign(v) stops holding, so only the all-zero case is short-cut there)
        const bool unit = (!W16 && ws[q] == 1.0f) || ws[q] == 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const int v = acc[q][ch];
            out[q][ch] = v - (v > 0) + (v < 0);
        }
        if (!unit) {
            if (W16) {       // (row[x] << 8) / (weight + 1): ints, division truncates toward zero; the sum of weights is a short
                const int d = wrap_s16(static_cast<int>(ws[q])) + 1;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) out[q][ch] = d != 0 ? wrap_s16((acc[q][ch] * 256) / d) : 0;
            } else {
                const float d = ws[q] + kWeightEps;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    out[q][ch] = wrap_s16(static_cast<int>(static_cast<float>(acc[q][ch]) / d));
            }
        }
    }
}

#ifndef MST_ABLATE_C
#define MST_ABLATE_C 0     // timing experiments only (results are wrong): 1 no coarse bands, 2 no collapse, 4 no final pyrUp,
#endif                     // 8 no final band, 16 no panorama stores (generic path), 64 no strips, 256 no blocked quad order,
                           // 512 no ownership, 1024 no wide stores; strips: 2048 no level-0 loads, 4096 no panorama
                           // stores, 8192 no collapsed-level loads; 16384 level 1 of every coarse tile as if owned
// LDS row pitch (in pixels) of a level region: its width rounded up to even, so that a quad row is one aligned
// 16-byte access even where the top level's width is odd (the pad column takes the pair's second pixel there)
__host__ __device__ inline int region_pitch(const Region& r) { return (r.x1 - r.x0 + 1) & ~1; }

// The regions of a canvas tile at every level, worked out once per block (lane l of the first wave derives level l)
// and kept in LDS: the phases below used to re-derive them -- a scalar loop over the levels -- for every 64-quad
// chunk and every collapse step, which made 29 % of this kernel's instructions scalar bookkeeping.  Row l:
// x0, y0, x1, y1 of the region, and ceil(2^20 / quads per row) for the quad index -> (row, column) split.
constexpr int kRegionRow = 8;
__device__ __forceinline__ void fill_region_table(const BlendArgs& a, int base, int tx, int ty, int* __restrict__ s_reg)
{
    const int l = static_cast<int>(threadIdx.x);
    if (l >= base && l < a.nlev) {
        const Region r = blend_region(a, base, tx, ty, l);
        const int qw = max(1, (r.x1 - r.x0 + 1) >> 1);
        s_reg[l * kRegionRow + 0] = r.x0; s_reg[l * kRegionRow + 1] = r.y0;
        s_reg[l * kRegionRow + 2] = r.x1; s_reg[l * kRegionRow + 3] = r.y1;
        s_reg[l * kRegionRow + 4] = ((1 << 20) + qw - 1) / qw;
    }
}
__device__ __forceinline__ Region table_region(const int* __restrict__ s_reg, int l)
{
    const int4 v = *reinterpret_cast<const int4*>(s_reg + l * kRegionRow);
    Region r;
    r.x0 = __builtin_amdgcn_readfirstlane(v.x); r.y0 = __builtin_amdgcn_readfirstlane(v.y);
    r.x1 = __builtin_amdgcn_readfirstlane(v.z); r.y1 = __builtin_amdgcn_readfirstlane(v.w);
    return r;
}

struct QuadIter {
    Region r;
    int rw, qw, nq, pitch;
    uint32_t inv = 0;            // ceil(2^20 / qw) when known: q / qw == (q * inv) >> 20 for q < 4096, qw <= 64
    __device__ __forceinline__ QuadIter(const int* __restrict__ s_reg, int l) : QuadIter(table_region(s_reg, l))
    {
        inv = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(s_reg[l * kRegionRow + 4]));
    }
    __device__ __forceinline__ QuadIter(const Region& reg) : r(reg)
    {
        rw = r.x1 - r.x0;
        pitch = region_pitch(r);
        qw = (rw + 1) >> 1;
        nq = qw * ((r.y1 - r.y0 + 1) >> 1);
    }
    __device__ __forceinline__ void at(int q, int& x, int& y, bool valid[4]) const
    {
        // (the multiply form needs q < 4096 and qw <= 64: a 128 x 64 tile has 2048 quads in rows of 64 at most)
        const int qy = (inv != 0 && C_TW <= 128 && C_TW * C_TH <= 4 * 4096 && CC_TW <= 128 && CC_TW * CC_TH_MAX <= 4 * 4096) ? static_cast<int>((static_cast<uint32_t>(q) * inv) >> 20) : q / qw;
        const int qx = q - __mul24(qy, qw);
        x = r.x0 + 2 * qx;
        y = r.y0 + 2 * qy;
#pragma unroll
        for (int k = 0; k < 4; ++k) valid[k] = (x + (k & 1)) < r.x1 && (y + (k >> 1)) < r.y1;
    }
    // The same quads in blocks of 8 x 8 (one block = 16 x 16 pixels = one wave): a seam that crosses the tile then
    // runs through few waves, and a wave whose pixels all belong to one camera skips the other cameras' work as a
    // whole (the per-camera `continue`s of band_quad are taken by every lane).  Needs qw and the quad rows to be
    // multiples of 8; row-major otherwise.
    __device__ __forceinline__ bool blockable() const { return (qw & 7) == 0 && ((nq / qw) & 7) == 0; }
    __device__ __forceinline__ void at_blocked(int q, int& x, int& y, bool valid[4]) const
    {
        const int blk = q >> 6, in = q & 63;
        const int bpr = qw >> 3;                          // blocks per block row
        const int by = blk / bpr, bx = blk - __mul24(by, bpr);
        x = r.x0 + 2 * (8 * bx + (in & 7));
        y = r.y0 + 2 * (8 * by + (in >> 3));
#pragma unroll
        for (int k = 0; k < 4; ++k) valid[k] = (x + (k & 1)) < r.x1 && (y + (k >> 1)) < r.y1;
    }
};

// phase 1 for all coarse levels at once.  The levels above 1 hold only a few dozen quads per tile, so looping
// over them one after the other leaves most of the block idle while each level's load -> band chain runs its
// course, five times in a row.  Here the quads of every level form one index space cut into 64-quad chunks
// (coarsest level first); the waves of the block take chunks round-robin, so the levels are in flight together.
// The level of a chunk is wave-uniform, which keeps every descriptor access scalar.
// Tiles that are not pure still own most of their levels outright (a tile a few hundred pixels from a seam mixes
// cameras on its coarsest levels only): `owned_levels` (bit l) / `owner_bits` send those levels down the integer path.
template <int NT, bool PURE, bool W16 = false>
__device__ __forceinline__ void band_levels(const CamDesc* __restrict__ cams, const BlendArgs& a, int top, int tx, int ty,
                                            uint32_t cam_bits, uint2* __restrict__ lds, const int* __restrict__ s_reg,
                                            int first, uint32_t owned_levels = 0, uint32_t owner_bits = 0)
{
    constexpr int kWaves = NT / 64;
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    int nch[kMaxLevels];          // chunks per level; only ever indexed by unrolled constants
    int total = 0;
#pragma unroll
    for (int k = 1; k < kMaxLevels; ++k) {
        nch[k] = (k >= first && k <= top) ? (QuadIter(s_reg, k).nq + 63) >> 6 : 0;
        total += nch[k];
    }
    for (int chunk = wave; chunk < total; chunk += kWaves) {
        int l = 0, first = 0, acc = 0;
#pragma unroll
        for (int k = kMaxLevels - 1; k >= 1; --k) {
            if (l == 0 && chunk < acc + nch[k]) { l = k; first = acc; }
            acc += nch[k];
        }
        const QuadIter it(s_reg, l);
        uint2* s_out = lds + a.lds_off[l];
        const int q = ((chunk - first) << 6) + lane;
        if (q >= it.nq) continue;
        int x, y;
        bool valid[4];
        it.at(q, x, y, valid);
        int n[4][3];
        float ws[4];
        if (MST_ABLATE_C & 1) {
            for (int k = 0; k < 4; ++k) n[k][0] = n[k][1] = n[k][2] = x + y + k;
        } else if (!PURE && ((owned_levels >> l) & 1u)) {        // (wave-uniform) this level of the tile is owned outright
            if (l == top) band_quad<true, false, true>(cams, a, l, x, y, valid, owner_bits, n, ws);
            else band_quad<false, false, true>(cams, a, l, x, y, valid, owner_bits, n, ws);
        } else if (l == top) {
            band_quad<true, false, PURE, W16>(cams, a, l, x, y, valid, cam_bits, n, ws);
        } else {
            band_quad<false, false, PURE, W16>(cams, a, l, x, y, valid, cam_bits, n, ws);
        }
        // (x, region origin and region width are even: both pixels of a quad row are valid or invalid together)
#pragma unroll
        for (int ry = 0; ry < 2; ++ry)
            if (valid[2 * ry])
                lds_store_pair(s_out, __mul24(y + ry - it.r.y0, it.pitch) + (x - it.r.x0), n[2 * ry], n[2 * ry + 1]);
    }
}

__device__ __forceinline__ void dump_level(const BlendArgs& a, int l, int px, int py, const int v[3])
{
    const uint32_t plane = static_cast<uint32_t>(a.cw[l]) * a.ch[l];
    int16_t* dd = a.dbg_C[l] + static_cast<uint32_t>(py) * a.cw[l] + px;
    dd[0] = static_cast<int16_t>(v[0]);
    dd[plane] = static_cast<int16_t>(v[1]);
    dd[2 * plane] = static_cast<int16_t>(v[2]);
}

// phase 2: restoreImageFromLaplacePyr inside LDS: level l <- saturate(pyrUp(level l+1) + level l)
template <int NT>
__device__ __forceinline__ void collapse_level(const BlendArgs& a, int l, int tx, int ty, uint2* __restrict__ lds,
                                               const int* __restrict__ s_reg)
{
    const QuadIter it(s_reg, l);
    const Region ur = table_region(s_reg, l + 1);
    const uint2* s_up = lds + a.lds_off[l + 1];
    uint2* s_io = lds + a.lds_off[l];
    for (int q = threadIdx.x; q < it.nq; q += NT) {
        int x, y;
        bool valid[4];
        it.at(q, x, y, valid);
        int up[4][3];
        pyrup_quad_lds(s_up, ur.x0, ur.y0, region_pitch(ur), a.cw[l + 1], a.ch[l + 1], x >> 1, y >> 1, up);
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
            if (!valid[2 * ry]) continue;
            const int py = y + ry;
            uint2* p = s_io + __mul24(py - it.r.y0, it.pitch) + (x - it.r.x0);
            const uint4 n = *reinterpret_cast<const uint4*>(p);          // the quad row: two pixels, 16 bytes, aligned
            const int v0[3] = {sat_s16(up[2 * ry][0] + s16_lo(n.x)), sat_s16(up[2 * ry][1] + s16_hi(n.x)),
                               sat_s16(up[2 * ry][2] + s16_lo(n.y))};     // cv::add saturates
            const int v1[3] = {sat_s16(up[2 * ry + 1][0] + s16_lo(n.z)), sat_s16(up[2 * ry + 1][1] + s16_hi(n.z)),
                               sat_s16(up[2 * ry + 1][2] + s16_lo(n.w))};
            lds_store_pair(s_io, __mul24(py - it.r.y0, it.pitch) + (x - it.r.x0), v0, v1);
            if (a.dbg_C[l]) { dump_level(a, l, x, py, v0); dump_level(a, l, x + 1, py, v1); }
        }
    }
}

// crop to dst_roi_final_, mask, convertTo(8U) results of one quad -> panorama (+ optional mask plane)
//
// store_mode 2 (wide): the panorama's base and stride are multiples of four and lane pairs (2i, 2i+1) hold
// horizontally adjacent quads starting at a multiple of four pixels.  The two lanes then swap half of their quads
// over DPP -- the even lane ends up with the upper row of both quads (4 pixels = 12 bytes), the odd lane with the
// lower row -- and each issues ONE aligned 12-byte store; a wave writes two runs of 384 contiguous bytes per
// instruction.  The narrow form it replaces (three 2-byte stores per quad row at a 6-byte lane stride) kept the
// write path busy for about half of this kernel's run time: every instruction touched every cache line of the run
// with a sparse byte mask.  store_mode 1: 2-byte stores (even base and stride); 0: byte stores.
__device__ __forceinline__ uint32_t lane_swap1(uint32_t v)       // value of lane ^ 1 (quad_perm [1,0,3,2])
{
    return static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(v), 0xB1, 0xF, 0xF, false));
}

__device__ __forceinline__ void store_quad(const BlendArgs& a, uint8_t* __restrict__ pano, uint8_t* __restrict__ pano_mask,
                                           int x, int y, const bool valid[4], const uint32_t o8[4][3], const bool m[4],
                                           int store_mode)
{
    bool narrow = true;
    if (store_mode == 2) {
        // every lane of an active pair takes part in the swap; the pair falls back to the narrow stores when its four
        // pixels are not all inside the crop and the region (right edge of the panorama)
        const bool odd = (threadIdx.x & 1u) != 0;
        const int xe = odd ? x - 2 : x;                                       // first pixel of the pair's run of four
        const bool mine = valid[0] && valid[1] && valid[2] && valid[3];
        const bool pair_ok = mine && lane_swap1(mine ? 1u : 0u) != 0u && xe + 3 < a.final_w && (xe & 3) == 0;
        // 6 bytes per quad row: B0 G0 R0 B1 | G1 R1
        const uint32_t lo0 = o8[0][0] | o8[0][1] << 8 | o8[0][2] << 16 | o8[1][0] << 24, hi0 = o8[1][1] | o8[1][2] << 8;
        const uint32_t lo1 = o8[2][0] | o8[2][1] << 8 | o8[2][2] << 16 | o8[3][0] << 24, hi1 = o8[3][1] | o8[3][2] << 8;
        const uint32_t rlo = lane_swap1(odd ? lo0 : lo1), rhi = lane_swap1(odd ? hi0 : hi1);   // what the partner needs
        const uint32_t flo = odd ? rlo : lo0, fhi = odd ? rhi : hi0;          // first six bytes of my row of four pixels
        const uint32_t slo = odd ? lo1 : rlo, shi = odd ? hi1 : rhi;          // last six
        const int py = odd ? y + 1 : y;
        if (pair_ok) {
            narrow = false;
            if (py < a.final_h && !((MST_ABLATE_C & 16) && flo != 77777u)) {
                uint32_t* o = reinterpret_cast<uint32_t*>(pano + __umul24(py, a.pano_stride) + xe * 3);
                uint3 v;
                v.x = flo;
                v.y = (fhi & 0xFFFFu) | slo << 16;
                v.z = slo >> 16 | shi << 16;
                *reinterpret_cast<uint3*>(o) = v;
            }
        }
    }
#pragma unroll
    for (int ry = 0; ry < 2; ++ry) {
        const int py = y + ry;
        if (py >= a.final_h || !valid[2 * ry]) continue;                     // crop to dst_roi_final_
        if ((MST_ABLATE_C & 16) && o8[2 * ry][0] != 77777u) continue;
        uint8_t* o = pano + __umul24(py, a.pano_stride) + x * 3;
        const uint32_t* p0 = o8[2 * ry];
        const uint32_t* p1 = o8[2 * ry + 1];
        const bool both = valid[2 * ry + 1] && x + 1 < a.final_w;
        if (x < a.final_w) {
            if (narrow) {
                if (both && store_mode >= 1) {
                    uint16_t* o16 = reinterpret_cast<uint16_t*>(o);
                    o16[0] = static_cast<uint16_t>(p0[0] | p0[1] << 8);
                    o16[1] = static_cast<uint16_t>(p0[2] | p1[0] << 8);
                    o16[2] = static_cast<uint16_t>(p1[1] | p1[2] << 8);
                } else {
                    o[0] = static_cast<uint8_t>(p0[0]); o[1] = static_cast<uint8_t>(p0[1]); o[2] = static_cast<uint8_t>(p0[2]);
                    if (both) { o[3] = static_cast<uint8_t>(p1[0]); o[4] = static_cast<uint8_t>(p1[1]); o[5] = static_cast<uint8_t>(p1[2]); }
                }
            }
            if (pano_mask) {
                uint8_t* om = pano_mask + __umul24(py, a.mask_stride) + x;
                om[0] = m[2 * ry] ? 255 : 0;
                if (both) om[1] = m[2 * ry + 1] ? 255 : 0;
            }
        }
    }
}

// how a tile's level-0 quads may be stored (see store_quad): `pairs` = consecutive lanes hold horizontally adjacent
// quads and even lanes start on a multiple of four pixels
__device__ __forceinline__ int pano_store_mode(const BlendArgs& a, const uint8_t* pano, bool pairs)
{
    const uint32_t al = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(pano)) | a.pano_stride;
    if (pairs && (al & 3u) == 0 && !(MST_ABLATE_C & 1024)) return 2;
    return (al & 1u) == 0 ? 1 : 0;
}

// v - sign(v)
__device__ __forceinline__ int toward_zero(int v)
{
    int sgn;
    asm("v_med3_i32 %0, %1, -1, 1" : "=v"(sgn) : "v"(v));
    return v - sgn;
}

// Level L (0 in k_blend_fine, 1 in k_blend_coarse) of a tile that is owned by one camera (see band_quad), is not the
// top level and covers a whole 128 x 64 region (for level 0: inside the final crop).  Every quad is valid; a wave
// takes a strip of the tile one quad row wide (64 quads: lane == quad column) and walks down it.  cv::pyrUp is
// separable, and both of its sources here -- the camera's Gaussian level L + 1 and the collapsed canvas level
// L + 1 -- advance by exactly one source row per quad row.  So a lane
//   * loads ONE source pixel of each per quad row (its own column; the two columns beside it come from the lanes
//     beside it over DPP, and only the first and last lane of each 16-lane DPP row fetch an outer column -- one
//     masked load instruction), not a 3 x 3 block,
//   * filters that row horizontally once and keeps the last three filtered rows in registers,
// and the vertical taps then only combine registers.  Per quad: 6 loads instead of 20, a third of the horizontal
// filter work; the loads of the next row are issued before the current quad row is evaluated.
// `up` is the collapsed level L + 1: the global level-1 canvas (L == 0; pitch cw[1], origin 0) or the tile's LDS
// region of level 2 (L == 1).  Output: the panorama (L == 0; 12-byte stores, see store_quad) or the collapsed
// level-1 canvas (L == 1).
//
// NC / MASKED (level 0 only): the tile lies under NC cameras (1 or 2) whose level-0 weights are all exactly 0 or 1
// and never both non-zero on a pixel (seam masks; image borders), every window covering the whole tile (checked
// per calibration, mb_plan.cpp).  A pixel's band is then its one covering camera's lap - sign(lap), or 0 under no
// camera (where dst_mask_ is 0 and the panorama pixel with it): still integers only, the masks pick per pixel.
template <int L, int NC, bool MASKED, int NT, int TH, typename UpPtr>
__device__ __forceinline__ void owned_strip(const CamDesc* __restrict__ cams, const BlendArgs& a, const Region& r,
                                            uint32_t cam_bits, UpPtr up, int up_x0, int up_y0, int up_pitch,
                                            uint8_t* __restrict__ pano, uint8_t* __restrict__ pano_mask, uint64_t* __restrict__ c1_out)
{
    constexpr int kTileH = TH;
    static_assert(C_TW == 128 && CC_TW == 128 && (kTileH / 2) % (NT / 64) == 0, "one wave per quad row of the tile");
    static_assert(L == 0 || (NC == 1 && !MASKED), "masks are level-0 weights");
    constexpr int kRows = (kTileH / 2) / (NT / 64);              // quad rows per wave
    constexpr bool kPair = L == 0 && NC == 1 && MST_L0_PACKED;   // packed level 0, one camera: lane pairs share aligned 12-byte loads
    const int lane = static_cast<int>(threadIdx.x & 63u);
    const int wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
    const int uw = a.cw[L + 1], uh = a.ch[L + 1];
    const int x = r.x0 + 2 * lane, y0 = r.y0 + 2 * kRows * wave;
    const int cj = x >> 1;
    int cjm, cjp;
    up_idx(cj, uw, cjm, cjp);
    const bool first = (lane & 15) == 0, last = (lane & 15) == 15, edge = first || last;
    const int cje = (first ? cjm : cjp) - up_x0;                 // the outer column an edge lane fetches itself
    const int cjc = cj - up_x0;
    const bool odd = (lane & 1) != 0;

    struct Cam {
        const uint32_t MST_GLOBAL* gup;      // Gaussian level L + 1
        const GPix MST_GLOBAL* gq;           // this lane's quad of level L, upper row (advanced as the strip is walked); level 0: the
                                             // quad's six packed bytes (2-byte aligned), rows gpitch bytes apart
        uint32_t gpitch;
        const uint8_t MST_GLOBAL* mq;        // MASKED: this lane's quad of the bordered level-0 mask
        int lw, sw, sh, i0, j, je;
    } cam[NC];
    {
        uint32_t bits = cam_bits;
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const CamDesc& c = cams[__builtin_ctz(bits)];
            bits &= bits - 1;
            Cam& k = cam[n];
            k.lw = c.lw[L]; k.sw = c.lw[L + 1]; k.sh = c.lh[L + 1];
            const int lx = x - c.ox[L], ly0 = y0 - c.oy[L];
            k.i0 = ly0 >> 1;
            k.j = lx >> 1;
            int jm, jp;
            up_idx(k.j, k.sw, jm, jp);
            k.je = first ? jm : jp;
            k.gup = gp(c.G[L + 1]);
            const uint32_t off = __umul24(ly0, k.lw) + lx;
            k.gpitch = L == 0 ? static_cast<uint32_t>(c.l0_pitch) : static_cast<uint32_t>(k.lw) * 4u;
            // Level 0 (packed BGR8): a lane pair (2 i, 2 i + 1) covers four pixels = twelve bytes per row.  One camera (kPair;
            // the caller guarantees that a pair's pixels start on a 12-byte group of the camera's rows: window origins are
            // multiples of 2^bands, i.e. always from two bands on): the even lane loads the upper row's twelve bytes, the odd lane
            // the lower row's -- ONE aligned load per lane and quad row instead of two -- and the halves are swapped over DPP.
            // Two cameras: every lane loads its own six bytes of both rows (2-byte aligned 8-byte loads; that path sets the
            // kernel's register allocation -- 128 VGPRs, four waves per SIMD -- and the pairing's temporaries would push it to 132
            // and three).  Compile-time, not a flag: a branch around the loads of the unrolled walk costs the overlap of one
            // quad row's loads with the row before (k_blend_fine 16.5 -> 20.3 us per frame set, measured).
            if (L == 0) k.gq = (const GPix MST_GLOBAL*)(kPair ? l0_at(c, lx - (odd ? 2 : 0), ly0 + (odd ? 1 : 0)) : l0_at(c, lx, ly0));
            else k.gq = (const GPix MST_GLOBAL*)(c.G[L] + off);
            k.mq = MASKED ? gp(c.mask_b) + off : nullptr;
        }
    }

    struct RawG { uint32_t g, ge; };                             // one source row as loaded: own column, outer column
    struct RawC { uint2 u, ue; };
    struct RowG { uint32_t hebr, hobr, heg, hog; };              // horizontally filtered: even / odd output column
    struct RowC { int he[3], ho[3]; };
    auto load_g = [&](RawG& w, const Cam& k, int gi) {           // source row gi of the camera's level L + 1
        const int go = __mul24(gi, k.sw);
        w.g = k.gup[go + k.j];
        w.ge = 0u;
        if (edge) w.ge = k.gup[go + k.je];
    };
    auto load_c = [&](RawC& w, int ci) {                         // source row ci of the collapsed canvas level L + 1
        const int co = __mul24(ci - up_y0, up_pitch);
        if (MST_ABLATE_C & 8192) { w.u = make_uint2(co + cjc, co * 3 + cjc); w.ue = w.u; return; }   // timing experiments
        w.u = as_u2(up[co + cjc]);
        w.ue = make_uint2(0u, 0u);
        if (edge) w.ue = as_u2(up[co + cje]);
    };
    // (the DPP shifts first, on every lane: a lane switched off by a branch would not be readable by its neighbour)
    auto filter_g = [&](const RawG& w, RowG& g) {
        const uint32_t gb = from_lane_below(w.g), ga = from_lane_above(w.g);
        const uint32_t gl = first ? w.ge : gb, gr = last ? w.ge : ga;
        g.hebr = lanes_br(gl) + 6u * lanes_br(w.g) + lanes_br(gr);
        g.hobr = 4u * (lanes_br(w.g) + lanes_br(gr));
        g.heg = lane_g(gl) + 6u * lane_g(w.g) + lane_g(gr);
        g.hog = 4u * (lane_g(w.g) + lane_g(gr));
    };
    auto filter_c = [&](const RawC& w, RowC& u) {
        const uint32_t uxb = from_lane_below(w.u.x), uyb = from_lane_below(w.u.y);
        const uint32_t uxa = from_lane_above(w.u.x), uya = from_lane_above(w.u.y);
        const uint32_t ulx = first ? w.ue.x : uxb, uly = first ? w.ue.y : uyb;
        const uint32_t urx = last ? w.ue.x : uxa, ury = last ? w.ue.y : uya;
        const int av[3] = {s16_lo(ulx), s16_hi(ulx), s16_lo(uly)};
        const int bv[3] = {s16_lo(w.u.x), s16_hi(w.u.x), s16_lo(w.u.y)};
        const int cv[3] = {s16_lo(urx), s16_hi(urx), s16_lo(ury)};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            u.he[ch] = av[ch] + 6 * bv[ch] + cv[ch];
            u.ho[ch] = 4 * (bv[ch] + cv[ch]);
        }
    };
    struct Quad { GPix q0, q1; uint32_t m; };                    // Gaussian quad (two rows) and its four mask bytes
    auto load_quad = [&](Quad& q, const Cam& k) {
        if (MST_ABLATE_C & 2048) {         // timing experiments: no Gaussian quad loads
            q.q0 = GPix{static_cast<uint32_t>(x * 3 + k.j), static_cast<uint32_t>(x * 5 + k.j)};
            q.q1 = GPix{static_cast<uint32_t>(x * 7 + k.j), static_cast<uint32_t>(x * 11 + k.j)};
        } else {
            // (level 0 of a camera is read exactly once per frame set: streamed past the caches -- non-temporal -- which leaves
            // them to the level-1 rows the strips above and below re-read: k_blend_fine 169.6 -> 160.4 us per 8 frame sets)
            if (kPair) {
                typedef u32x3_t __attribute__((aligned(4))) U32x3A;
#if !defined(MST_NO_NT_LOAD)
                const u32x3_t d = __builtin_nontemporal_load((const U32x3A MST_GLOBAL*)k.gq);
#else
                const u32x3_t d = *(const U32x3A MST_GLOBAL*)k.gq;
#endif
                // my row's twelve bytes (even lane: upper row, odd lane: lower row) and the partner's
                const uint32_t p0 = lane_swap1(d.x), p1 = lane_swap1(d.y), p2 = lane_swap1(d.z);
                // even lane: pixels 0, 1 = bytes 0 .. 5 of both rows; odd lane: pixels 2, 3 = bytes 6 .. 11
                q.q0.x = odd ? __builtin_amdgcn_alignbyte(p2, p1, 2) : d.x;
                q.q0.y = odd ? p2 >> 16 : d.y;
                q.q1.x = odd ? __builtin_amdgcn_alignbyte(d.z, d.y, 2) : p0;
                q.q1.y = odd ? d.z >> 16 : p1;
            } else if (L == 0) {
                // (packed: 2-byte aligned pairs; dword level 0: aligned 8-byte pairs)
                typedef std::conditional_t<MST_L0_PACKED != 0, GPixU, GPix> L0Pair;
#if !defined(MST_NO_NT_LOAD)
                q.q0 = __builtin_nontemporal_load((const L0Pair MST_GLOBAL*)k.gq);
                q.q1 = __builtin_nontemporal_load((const L0Pair MST_GLOBAL*)((const uint8_t MST_GLOBAL*)k.gq + k.gpitch));
#else
                q.q0 = *(const L0Pair MST_GLOBAL*)k.gq;
                q.q1 = *(const L0Pair MST_GLOBAL*)((const uint8_t MST_GLOBAL*)k.gq + k.gpitch);
#endif
            } else {
                q.q0 = k.gq[0];
                q.q1 = *(const GPix MST_GLOBAL*)((const uint32_t MST_GLOBAL*)k.gq + k.lw);
            }
            if (L == 0 && MST_L0_PACKED) {      // B0 G0 R0 B1 | G1 R1 . . -> the second pixel as a dword of its own (byte 3 of either is never read)
                q.q0.y = __builtin_amdgcn_alignbyte(q.q0.y, q.q0.x, 3);
                q.q1.y = __builtin_amdgcn_alignbyte(q.q1.y, q.q1.x, 3);
            }
        }
        q.m = 0xFFFFFFFFu;
        if (MASKED)
            q.m = static_cast<uint32_t>(*(const uint16_t MST_GLOBAL*)k.mq) |
                  static_cast<uint32_t>(*(const uint16_t MST_GLOBAL*)(k.mq + k.lw)) << 16;
    };

    // the three source rows of the strip's first quad row (clamped like cv::pyrUp's border); afterwards one new row each
    const int ci0 = y0 >> 1;
    RowG rg[NC][3];
    RowC rc[3];
    Quad qd[NC];
    {
        int cim, cip;
        up_idx(ci0, uh, cim, cip);
        RawC w0, w1, w2;
        load_c(w0, cim); load_c(w1, ci0); load_c(w2, cip);
        filter_c(w0, rc[0]); filter_c(w1, rc[1]); filter_c(w2, rc[2]);
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            int im, ip;
            up_idx(cam[n].i0, cam[n].sh, im, ip);
            RawG g0, g1, g2;
            load_g(g0, cam[n], im); load_g(g1, cam[n], cam[n].i0); load_g(g2, cam[n], ip);
            filter_g(g0, rg[n][0]); filter_g(g1, rg[n][1]); filter_g(g2, rg[n][2]);
            load_quad(qd[n], cam[n]);
        }
    }
    // output pointers of this lane, advanced by two rows per quad row
    bool aligned_rows = true;
    uint8_t* po = nullptr;           // L == 0: even lanes write the upper row of a lane pair's two quads, odd lanes the lower
    uint8_t* pm = nullptr;
    GPix MST_GLOBAL* co = nullptr;   // L == 1
    if (L == 0) {
        po = pano + __umul24(y0 + (odd ? 1 : 0), a.pano_stride) + (odd ? x - 2 : x) * 3;
        aligned_rows = pano_store_mode(a, pano, true) == 2;
        if (pano_mask) pm = pano_mask + __umul24(y0, a.mask_stride) + x;
    } else {
        co = (GPix MST_GLOBAL*)c1_out + __mul24(y0, a.cw[L]) + x;
    }
#pragma unroll
    for (int t = 0; t < kRows; ++t) {
        RawC wc;
        RawG wg[NC];
        Quad qn[NC];
        if (t + 1 < kRows) {          // next quad row: its new source rows (i + 2, clamped) and its Gaussian quads
            load_c(wc, min(ci0 + t + 2, uh - 1));
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                load_g(wg[n], cam[n], min(cam[n].i0 + t + 2, cam[n].sh - 1));
                cam[n].gq = (const GPix MST_GLOBAL*)((const uint8_t MST_GLOBAL*)cam[n].gq + 2u * cam[n].gpitch);
                if (MASKED) cam[n].mq += 2 * cam[n].lw;
                load_quad(qn[n], cam[n]);
            }
        }
        int upv[4][3];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            upv[0][ch] = (rc[0].he[ch] + 6 * rc[1].he[ch] + rc[2].he[ch] + 32) >> 6;      // (no saturation needed: see pyrup_quad_s16)
            upv[1][ch] = (rc[0].ho[ch] + 6 * rc[1].ho[ch] + rc[2].ho[ch] + 32) >> 6;
            upv[2][ch] = (4 * (rc[1].he[ch] + rc[2].he[ch]) + 32) >> 6;
            upv[3][ch] = (4 * (rc[1].ho[ch] + rc[2].ho[ch]) + 32) >> 6;
        }
        int band[4][3];              // normalised band of level L
        uint32_t many = 0;           // MASKED: a byte per quad pixel, non-zero under a camera
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const RowG* g3 = rg[n];
            uint32_t ubr[4], ug[4];
            ubr[0] = g3[0].hebr + 6u * g3[1].hebr + g3[2].hebr;  ug[0] = g3[0].heg + 6u * g3[1].heg + g3[2].heg;
            ubr[1] = g3[0].hobr + 6u * g3[1].hobr + g3[2].hobr;  ug[1] = g3[0].hog + 6u * g3[1].hog + g3[2].hog;
            ubr[2] = 4u * (g3[1].hebr + g3[2].hebr);             ug[2] = 4u * (g3[1].heg + g3[2].heg);
            ubr[3] = 4u * (g3[1].hobr + g3[2].hobr);             ug[3] = 4u * (g3[1].hog + g3[2].hog);
            const uint32_t g[4] = {qd[n].q0.x, qd[n].q0.y, qd[n].q1.x, qd[n].q1.y};
            many |= qd[n].m;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t pbr = ((ubr[q] + 0x00200020u) >> 6) & 0x03FF03FFu, pg = (ug[q] + 32u) >> 6;
                const int lb = static_cast<int>(g[q] & 0xFF) - static_cast<int>(pbr & 0xFFFF);
                const int lg = static_cast<int>((g[q] >> 8) & 0xFF) - static_cast<int>(pg);
                const int lr = static_cast<int>((g[q] >> 16) & 0xFF) - static_cast<int>(pbr >> 16);
                // normalised band of an owned pixel: lap - sign(lap) (see band_quad)
                const int nb[3] = {toward_zero(lb), toward_zero(lg), toward_zero(lr)};
                const bool mine = !MASKED || ((qd[n].m >> (8 * q)) & 0xFFu) != 0;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    if (n == 0) band[q][ch] = mine ? nb[ch] : 0;
                    else if (mine) band[q][ch] = nb[ch];
                }
            }
        }
        int v[4][3];                 // collapsed level L; cv::add saturates (level 0: convertTo(8U)'s clamp to [0, 255] absorbs it)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) v[q][ch] = L == 0 ? upv[q][ch] + band[q][ch] : sat_s16(upv[q][ch] + band[q][ch]);
        if (L == 0) {
            uint32_t o8[4][3];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool m = !MASKED || ((many >> (8 * q)) & 0xFFu) != 0;                // dst_mask_
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) o8[q][ch] = m ? static_cast<uint32_t>(sat_u8(v[q][ch])) : 0u;
            }
            // 6 bytes per quad row: B0 G0 R0 B1 | G1 R1; lane pairs swap so that each lane holds one row of four pixels
            const uint32_t lo0 = o8[0][0] | o8[0][1] << 8 | o8[0][2] << 16 | o8[1][0] << 24, hi0 = o8[1][1] | o8[1][2] << 8;
            const uint32_t lo1 = o8[2][0] | o8[2][1] << 8 | o8[2][2] << 16 | o8[3][0] << 24, hi1 = o8[3][1] | o8[3][2] << 8;
            const uint32_t rlo = lane_swap1(odd ? lo0 : lo1), rhi = lane_swap1(odd ? hi0 : hi1);
            const uint32_t flo = odd ? rlo : lo0, fhi = odd ? rhi : hi0;
            const uint32_t slo = odd ? lo1 : rlo, shi = odd ? hi1 : rhi;
            uint3 w;
            w.x = flo;
            w.y = (fhi & 0xFFFFu) | slo << 16;
            w.z = slo >> 16 | shi << 16;
            if (!(MST_ABLATE_C & 4096) || w.x == 0x12345678u) {
                // (a plain store: marked non-temporal it costs 5 % of this kernel, measured).  A panorama whose base or row
                // stride is not a multiple of four -- a tight cv::Mat of a width that is not -- takes the same 12 bytes as
                // one unaligned access: gfx950 under HSA serves unaligned global accesses.
                struct __attribute__((packed, aligned(1))) Bytes12 { uint32_t x, y, z; };
                if (aligned_rows) *reinterpret_cast<uint3*>(po) = w;
                else *reinterpret_cast<Bytes12*>(po) = Bytes12{w.x, w.y, w.z};
            }
            po += 2 * a.pano_stride;
            if (pm) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    pm[(q >> 1) * a.mask_stride + (q & 1)] = (!MASKED || ((many >> (8 * q)) & 0xFFu) != 0) ? 255 : 0;
                pm += 2 * a.mask_stride;
            }
        } else {
            typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
#pragma unroll
            for (int ry = 0; ry < 2; ++ry) {
                const uint2 p0 = pack3(v[2 * ry][0], v[2 * ry][1], v[2 * ry][2]);
                const uint2 p1 = pack3(v[2 * ry + 1][0], v[2 * ry + 1][1], v[2 * ry + 1][2]);
#if !defined(MST_NO_NT_STORE)      // (read again a kernel later, eight frame sets away: k_blend_coarse 86.5 -> 83.9 us)
                __builtin_nontemporal_store(u32x4{p0.x, p0.y, p1.x, p1.y}, reinterpret_cast<u32x4 MST_GLOBAL*>(co + __mul24(ry, a.cw[L])));
#else
                *reinterpret_cast<u32x4 MST_GLOBAL*>(co + __mul24(ry, a.cw[L])) = u32x4{p0.x, p0.y, p1.x, p1.y};
#endif
                if (a.dbg_C[L]) { dump_level(a, L, x, y0 + 2 * t + ry, v[2 * ry]); dump_level(a, L, x + 1, y0 + 2 * t + ry, v[2 * ry + 1]); }
            }
            co += 2 * a.cw[L];
        }
        if (t + 1 < kRows) {
            rc[0] = rc[1]; rc[1] = rc[2];
            filter_c(wc, rc[2]);
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                rg[n][0] = rg[n][1]; rg[n][1] = rg[n][2];
                filter_g(wg[n], rg[n][2]);
                qd[n] = qn[n];
            }
        }
    }
}

// level 0 band + pyrUp of the collapsed level-1 canvas -> crop, mask, convertTo(8U), panorama
template <int NT, bool TOP, bool PURE, bool W16 = false>
__device__ __forceinline__ void final_level(const CamDesc* __restrict__ cams, const BlendArgs& a, const Region& r,
                                            uint32_t cam_bits, C1Ptr __restrict__ s_up,
                                            uint8_t* __restrict__ pano, uint8_t* __restrict__ pano_mask)
{
    const QuadIter it(r);
    const bool blocked = !PURE && !(MST_ABLATE_C & 256) && it.blockable();
    // lane pairs hold horizontally adjacent quads when a quad row has an even number of quads (row-major order) or in
    // the 8 x 8 blocked order; the region starts on a multiple of C_TW
    const int store_mode = pano_store_mode(a, pano, blocked || (it.qw & 1) == 0);
    for (int q = threadIdx.x; q < it.nq; q += NT) {
        int x, y;
        bool valid[4];
        if (blocked) it.at_blocked(q, x, y, valid);
        else it.at(q, x, y, valid);
        int n[4][3];
        float ws[4];
        if (MST_ABLATE_C & 8) {
            for (int k = 0; k < 4; ++k) { n[k][0] = n[k][1] = n[k][2] = x + y + k; ws[k] = 1.f; }
        } else {
            band_quad<TOP, true, PURE, W16>(cams, a, 0, x, y, valid, cam_bits, n, ws);
        }
        int up[4][3];
        if (MST_ABLATE_C & 4) {
            for (int k = 0; k < 4; ++k) up[k][0] = up[k][1] = up[k][2] = x - y + k;
        } else if (!TOP) pyrup_quad_lds(s_up, 0, 0, a.cw[1], a.cw[1], a.ch[1], x >> 1, y >> 1, up);
        uint32_t o8[4][3];
        bool m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = ws[k] > kWeightEps;                                           // dst_mask_ = weights > eps
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int v = TOP ? n[k][ch] : up[k][ch] + n[k][ch];       // (cv::add's saturation is absorbed by the clamp to [0, 255])
                o8[k][ch] = m[k] ? static_cast<uint32_t>(sat_u8(v)) : 0u;
            }
        }
        store_quad(a, pano, pano_mask, x, y, valid, o8, m, store_mode);
    }
}

// The base level of the coarse canvas stage (level 1): its band is formed like any other, the collapsed level above comes
// out of LDS (k_blend_coarse, two-level pyramids) or out of the collapsed level-2 canvas in global memory (k_blend_mid), and
// the sum -- the collapsed level-1 canvas -- goes to global memory for k_blend_fine (8 bytes a pixel).
template <int NT, bool TOP, bool PURE, bool W16, typename Ptr>
__device__ __forceinline__ void base_level_from(const CamDesc* __restrict__ cams, const BlendArgs& a, uint32_t cam_bits,
                                                const QuadIter& it, Ptr s_up, int ux0, int uy0, int upitch,
                                                uint2* __restrict__ c1)
{
    constexpr int l = 1;
    const int cw = a.cw[l];
    const bool wide = (cw & 1) == 0;       // rows start 16-byte aligned (c1 is), x is even
    for (int q = threadIdx.x; q < it.nq; q += NT) {
        int x, y;
        bool valid[4];
        it.at(q, x, y, valid);
        int n[4][3];
        float ws[4];
        band_quad<TOP, false, PURE, W16>(cams, a, l, x, y, valid, cam_bits, n, ws);
        int up[4][3];
        if (!TOP) pyrup_quad_lds(s_up, ux0, uy0, upitch, a.cw[l + 1], a.ch[l + 1], x >> 1, y >> 1, up);
#pragma unroll
        for (int ry = 0; ry < 2; ++ry) {
            if (!valid[2 * ry]) continue;
            int v[2][3];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch)
                    v[k][ch] = TOP ? n[2 * ry + k][ch] : sat_s16(up[2 * ry + k][ch] + n[2 * ry + k][ch]);   // cv::add saturates
            const int py = y + ry;
            GPix MST_GLOBAL* o = (GPix MST_GLOBAL*)c1 + __mul24(py, cw) + x;
            const uint2 p0 = pack3(v[0][0], v[0][1], v[0][2]), p1 = pack3(v[1][0], v[1][1], v[1][2]);
            if (wide && valid[2 * ry + 1]) {
                typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
                const u32x4 w = {p0.x, p0.y, p1.x, p1.y};
#if !defined(MST_NO_NT_STORE)
                __builtin_nontemporal_store(w, reinterpret_cast<u32x4 MST_GLOBAL*>(o));
#else
                *reinterpret_cast<u32x4 MST_GLOBAL*>(o) = w;
#endif
            } else {
                o[0] = GPix{p0.x, p0.y};
                if (valid[2 * ry + 1]) o[1] = GPix{p1.x, p1.y};
            }
            if (a.dbg_C[l]) {
                dump_level(a, l, x, py, v[0]);
                if (valid[2 * ry + 1]) dump_level(a, l, x + 1, py, v[1]);
            }
        }
    }
}
template <int NT, bool TOP, bool PURE, bool W16 = false>
__device__ __forceinline__ void coarse_base_level(const CamDesc* __restrict__ cams, const BlendArgs& a, uint32_t cam_bits,
                                                  const uint2* __restrict__ lds, const int* __restrict__ s_reg,
                                                  uint2* __restrict__ c1)
{
    const QuadIter it(s_reg, 1);
    Region ur = it.r;
    const uint2* s_up = nullptr;
    if (!TOP) {
        ur = table_region(s_reg, 2);
        s_up = lds + a.lds_off[2];
    }
    base_level_from<NT, TOP, PURE, W16>(cams, a, cam_bits, it, s_up, ur.x0, ur.y0, region_pitch(ur), c1);
}

// W16 (weight_type CV_16S) is a separate instantiation: its integer divides would otherwise set the register
// allocation (and spill) for the default kernel as well.
#ifndef MST_C_WAVES
#define MST_C_WAVES 1      // minimum waves per SIMD the register allocation must allow (occupancy experiments)
#endif

// 1-D grids over (tile, frame set).  Blocks are dealt to the XCDs round-robin, so XCD k = blockIdx.x % 8 sees the
// blocks j = blockIdx.x / 8 in order: consecutive j are the frame sets of one tile (the weight pyramids and masks a
// seam tile reads are frame-invariant: fetched into that XCD's L2 once, shared by the other sets).  Tiles go to the
// XCDs in groups of C_XCD_GROUP consecutive entries of the launch order: entries of one cost class are neighbours
// along a tile row, and a tile reads one column of its left and right neighbours' level-1 rows (pyrUp halo) -- a
// whole memory sector for 8 bytes -- which an XCD that also works on the neighbour already has in its L2.  "Most
// expensive first" still holds on every XCD up to the group size.
#ifndef MST_C_XCD_GROUP
#define MST_C_XCD_GROUP 8          // k_blend_fine: 139.7 / 137.9 / 134.0 / 130.7 us per 8 frame sets with groups of 1 / 2 / 4 / 8
#endif
#ifndef MST_CC_XCD_GROUP
#define MST_CC_XCD_GROUP 1         // k_blend_coarse (few, long blocks: balance matters more): 85.1 / 85.0 / 86.5 / 88.6 us
#endif
constexpr int C_XCD_GROUP = MST_C_XCD_GROUP, CC_XCD_GROUP = MST_CC_XCD_GROUP;
template <int G> __host__ __device__ inline int canvas_grid(int n_tiles, int n_sets)
{
    return ((n_tiles + 8 * G - 1) / (8 * G)) * 8 * G * n_sets;
}
template <int G> __device__ __forceinline__ bool tile_of_block(int n_tiles, int n_sets, int& q, int& set)
{
    const int xj = blockIdx.x >> 3, xk = blockIdx.x & 7;
    const int m = xj / n_sets;                        // this XCD's m-th tile
    set = xj - m * n_sets;
    q = (m / G) * (8 * G) + xk * G + m % G;
    return q < n_tiles;
}

// Canvas kernel 1 of 2: levels 1 .. top of the canvas pyramid, one block per CC_TW x TH tile of LEVEL 1.
//   phase 1  bands of the levels 2 .. top over the tile's regions (tile + pyrUp halo), normalised, into LDS
//   phase 2  restoreImageFromLaplacePyr inside LDS down to level 2
//   phase 3  level-1 band + pyrUp(level 2) -> the collapsed level-1 canvas in global memory
// Level 1 and above hold a quarter of the pixels but -- with the halo every level adds around a tile -- took three
// quarters of the time of the single canvas kernel this replaces: tiles four times as large in panorama terms
// halve the halo at every level, and level 0 runs as a streaming kernel of its own with no LDS and no barrier.
template <bool W16, int TH, int NT = CC_THREADS>
__global__ __launch_bounds__(NT, MST_C_WAVES) void k_blend_coarse(const CamDesc* __restrict__ cams_all, BlendArgs a)
{
    int q, set;
    if (!tile_of_block<CC_XCD_GROUP>(a.n_coarse_tiles, a.n_sets, q, set)) return;
    const CamDesc* __restrict__ cams = cams_all + set * a.n_cams;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint2* lds = reinterpret_cast<uint2*>(smem);
    const uint32_t tile = a.coarse_order[q];
    const int tx = tile % a.coarse_tiles_x, ty = tile / a.coarse_tiles_x;
    const uint32_t entry = a.coarse_cams[tile];
    const bool pure = (entry >> 31) != 0;            // see band_quad: one camera, weight exactly 1 on every level
    const uint32_t cam_bits = entry & 0x7FFFFFFFu;
    // levels of this tile owned by one camera with weight exactly 1 (bit l), and that camera
    const uint32_t own = (MST_ABLATE_C & 512) ? 0u : a.coarse_pure[tile];
    const uint32_t owned_levels = own & 0xFFFFu, owner_bits = 1u << (own >> 16);
    const int top = a.nlev - 1;
    MST_STAMP(1, 0);
    MST_STAMP_VAL(1, 8, static_cast<unsigned long long>(entry));
    __shared__ __attribute__((aligned(16))) int s_reg[kMaxLevels * kRegionRow];     // 352 bytes: keeps the dynamic base 16-aligned
    fill_region_table(a, 1, tx, ty, s_reg);
    __syncthreads();
    if (top >= 2) {
        // phase 1: every band above level 1, independent of each other -> no barrier until all are in LDS
        if (pure) band_levels<NT, true>(cams, a, top, tx, ty, cam_bits, lds, s_reg, 2);
        else band_levels<NT, false, W16>(cams, a, top, tx, ty, cam_bits, lds, s_reg, 2, owned_levels, owner_bits);
        MST_STAMP(1, 1);
        __syncthreads();
        MST_STAMP(1, 2);
        // phase 2: collapse coarse to fine (LDS only)
        if (a.dbg_C[top]) {
            const QuadIter it(s_reg, top);
            const uint2* s = lds + a.lds_off[top];
            for (int i = threadIdx.x; i < it.rw * (it.r.y1 - it.r.y0); i += NT) {
                const int py = it.r.y0 + i / it.rw, px = it.r.x0 + i % it.rw;
                const uint2 n = s[(py - it.r.y0) * it.pitch + (px - it.r.x0)];
                const int v[3] = {s16_lo(n.x), s16_hi(n.x), s16_lo(n.y)};
                dump_level(a, top, px, py, v);
            }
        }
        for (int l = top - 1; l >= 2; --l) {
            if (!(MST_ABLATE_C & 2)) collapse_level<NT>(a, l, tx, ty, lds, s_reg);       // (timing experiments)
            __syncthreads();
        }
    }
    MST_STAMP(1, 3);
    if (a.split) {
        // the tile's interior of the collapsed level 2 -> global memory; k_blend_mid forms level 1 from it
        const Region r2 = table_region(s_reg, 2);
        const int pitch2 = region_pitch(r2), cw2 = a.cw[2];
        const int ix0 = tx * (CC_TW / 2), iy0 = ty * (TH / 2);
        const int iw = min(CC_TW / 2, cw2 - ix0), ih = min(TH / 2, a.ch[2] - iy0);      // (iw is even: canvas levels below the top are)
        const uint2* s2 = lds + a.lds_off[2];
        GPix MST_GLOBAL* c2 = (GPix MST_GLOBAL*)a.c2[set];
        const int pairs = iw >> 1;
        for (int i = threadIdx.x; i < pairs * ih; i += NT) {
            const int y = i / pairs, xp = i - y * pairs;
            const int gx = ix0 + 2 * xp, gy = iy0 + y;
            const uint4 v = *reinterpret_cast<const uint4*>(s2 + (gy - r2.y0) * pitch2 + (gx - r2.x0));
            GPix MST_GLOBAL* o = c2 + __mul24(gy, cw2) + gx;
            if ((cw2 & 1) == 0) {
                *reinterpret_cast<u32x4_t MST_GLOBAL*>(o) = u32x4_t{v.x, v.y, v.z, v.w};
            } else {
                o[0] = GPix{v.x, v.y}; o[1] = GPix{v.z, v.w};
            }
        }
        if (iw & 1)
            for (int y = threadIdx.x; y < ih; y += NT) {
                const uint2 v = s2[(iy0 + y - r2.y0) * pitch2 + (ix0 + iw - 1 - r2.x0)];
                c2[__mul24(iy0 + y, cw2) + ix0 + iw - 1] = GPix{v.x, v.y};
            }
        MST_STAMP(1, 4);
        return;
    }
    // phase 3: level 1, owned outright on every pure tile and on most tiles near -- but not on -- a seam
    const bool own1 = pure || (owned_levels & 2u) || (MST_ABLATE_C & 16384);          // (16384: every tile as if owned at level 1 -- timing only)
    const uint32_t bits1 = pure ? cam_bits : (MST_ABLATE_C & 16384) && !(owned_levels & 2u) ? (cam_bits & (0u - cam_bits)) : owner_bits;
    uint2* c1 = reinterpret_cast<uint2*>(a.c1[set]);
    if (top == 1) {
        if (own1) coarse_base_level<NT, true, true>(cams, a, bits1, lds, s_reg, c1);
        else coarse_base_level<NT, true, false, W16>(cams, a, cam_bits, lds, s_reg, c1);
    } else {
        const Region r1 = table_region(s_reg, 1), ur = table_region(s_reg, 2);
        const bool full = r1.x1 - r1.x0 == CC_TW && r1.y1 - r1.y0 == TH && (a.cw[1] & 1) == 0;
        if (own1 && full && !(MST_ABLATE_C & 64))
            owned_strip<1, 1, false, NT, TH>(cams, a, r1, bits1, static_cast<const uint2*>(lds + a.lds_off[2]), ur.x0, ur.y0, region_pitch(ur),
                           nullptr, nullptr, a.c1[set]);
        else if (own1) coarse_base_level<NT, false, true>(cams, a, bits1, lds, s_reg, c1);
        else coarse_base_level<NT, false, false, W16>(cams, a, cam_bits, lds, s_reg, c1);
    }
    MST_STAMP(1, 4);
}

// Canvas kernel between the two (round 4, pyramids of three and more levels): level 1, one block per CC_TW x TH tile of
// level 1 -- the same tiles, flags and launch order as k_blend_coarse, which now stops at level 2.  Level-1 band +
// pyrUp of the collapsed level-2 canvas (global memory) -> the collapsed level-1 canvas.  No LDS, no barrier: the
// level-1 pass was 58 of k_blend_coarse's 85 us per 8 frame sets, at 3.6 ps per pixel behind two barrier-separated LDS
// phases in 256-thread blocks; in k_blend_fine's shape it runs at that kernel's rate.
#ifndef MST_CM_XCD_GROUP
#define MST_CM_XCD_GROUP 8
#endif
constexpr int CM_XCD_GROUP = MST_CM_XCD_GROUP;
template <bool W16, int NT, int TH>
__global__ __launch_bounds__(NT, MST_C_WAVES) void k_blend_mid(const CamDesc* __restrict__ cams_all, BlendArgs a)
{
    int q, set;
    if (!tile_of_block<CM_XCD_GROUP>(a.n_coarse_tiles, a.n_sets, q, set)) return;
    const CamDesc* __restrict__ cams = cams_all + set * a.n_cams;
    const uint32_t tile = a.coarse_order[q];
    const int tx = tile % a.coarse_tiles_x, ty = tile / a.coarse_tiles_x;
    const uint32_t entry = a.coarse_cams[tile];
    const bool pure = (entry >> 31) != 0;
    const uint32_t cam_bits = entry & 0x7FFFFFFFu;
    const uint32_t own = (MST_ABLATE_C & 512) ? 0u : a.coarse_pure[tile];
    const uint32_t owned_levels = own & 0xFFFFu, owner_bits = 1u << (own >> 16);
    const bool own1 = pure || (owned_levels & 2u);
    const uint32_t bits1 = pure ? cam_bits : owner_bits;
    const Region r1 = blend_region(a, 1, tx, ty, 1);
    const C1Ptr c2 = (C1Ptr)a.c2[set];
    const bool full = r1.x1 - r1.x0 == CC_TW && r1.y1 - r1.y0 == TH && (a.cw[1] & 1) == 0;
    if (own1 && full && !(MST_ABLATE_C & 64)) {
        owned_strip<1, 1, false, NT, TH>(cams, a, r1, bits1, c2, 0, 0, a.cw[2], nullptr, nullptr, a.c1[set]);
        return;
    }
    QuadIter it(r1);
    it.inv = static_cast<uint32_t>(((1 << 20) + it.qw - 1) / max(it.qw, 1));
    uint2* c1 = reinterpret_cast<uint2*>(a.c1[set]);
    if (own1) base_level_from<NT, false, true, false>(cams, a, bits1, it, c2, 0, 0, a.cw[2], c1);
    else base_level_from<NT, false, false, W16>(cams, a, cam_bits, it, c2, 0, 0, a.cw[2], c1);
}

// Canvas kernel 2 of 2: level 0, one block per C_TW x C_TH tile of the panorama: level-0 band + pyrUp of the
// collapsed level-1 canvas (k_blend_coarse) -> crop, mask, convertTo(8U) -> panorama.  No LDS, no barrier.
// (One kernel for every tile class on purpose.  Round 4 tried one launch per class -- the owned tiles' streaming path then
// compiles to 84 VGPRs instead of the 128 the two-camera path dictates -- and lost: 25.8 us per frame set against 16.5,
// the short lists run one after the other with their tails exposed, and the owned tiles alone took as long as everything
// together did; profiles/r04_experiments.md.)
template <bool W16, int NT>
__global__ __launch_bounds__(NT, MST_C_WAVES) void k_blend_fine(const CamDesc* __restrict__ cams_all, BlendArgs a)
{
    int q, set;
    if (!tile_of_block<C_XCD_GROUP>(a.n_tiles, a.n_sets, q, set)) return;
    const CamDesc* __restrict__ cams = cams_all + set * a.n_cams;
    uint8_t* __restrict__ pano = a.pano[set];
    uint8_t* __restrict__ pano_mask = a.pano_mask[set];
    const uint32_t tile = a.block_order[q];
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    const uint32_t entry = a.block_cams[tile];       // bit 31: level 0 owned by ONE camera with weight exactly 1
    const bool own0 = (MST_ABLATE_C & 512) ? false : (entry >> 31) != 0;
    const uint32_t cam_bits = entry & 0x3FFFFFFFu;   // (that camera alone when owned)
    const Region r0 = blend_region(a, 0, tx, ty, 0);
    const C1Ptr c1 = (C1Ptr)a.c1[set];
    if (cam_bits == 0) {            // under no camera at all: dst_mask_ is 0 and the panorama with it
        const int w = min(r0.x1, a.final_w) - r0.x0, h = min(r0.y1, a.final_h) - r0.y0;
        if (pano_store_mode(a, pano, true) == 2 && (w & 3) == 0) {          // dword stores
            const int wd = w * 3 / 4;
            for (int i = threadIdx.x; i < wd * max(h, 0); i += NT) {
                const int y = i / wd, b = i - y * wd;
                reinterpret_cast<uint32_t*>(pano + __umul24(r0.y0 + y, a.pano_stride) + r0.x0 * 3)[b] = 0u;
            }
        } else {
            for (int i = threadIdx.x; i < max(w, 0) * 3 * max(h, 0); i += NT) {
                const int y = i / (w * 3), b = i - y * (w * 3);
                pano[__umul24(r0.y0 + y, a.pano_stride) + r0.x0 * 3 + b] = 0;
            }
        }
        if (pano_mask)
            for (int i = threadIdx.x; i < max(w, 0) * max(h, 0); i += NT) {
                const int y = i / w, b = i - y * w;
                pano_mask[__umul24(r0.y0 + y, a.mask_stride) + r0.x0 + b] = 0;
            }
        return;
    }
    MST_STAMP(2, 0);
    MST_STAMP_VAL(2, 8, static_cast<unsigned long long>(entry));
    if (a.nlev == 1) {
        if (own0) final_level<NT, true, true>(cams, a, r0, cam_bits, c1, pano, pano_mask);
        else final_level<NT, true, false, W16>(cams, a, r0, cam_bits, c1, pano, pano_mask);
        return;
    }
    const bool full = r0.x1 - r0.x0 == C_TW && r0.y1 - r0.y0 == C_TH && r0.x1 <= a.final_w && r0.y1 <= a.final_h;
    const bool strip = full && !(MST_ABLATE_C & 64);
    const bool binary = !(MST_ABLATE_C & 512) && (entry >> 30) & 1u;      // 0/1 weights, disjoint, windows cover the tile
    // (the one-camera strips load level 0 by lane pairs: four pixels from a 12-byte group of the camera's rows)
    const bool one = (cam_bits & (cam_bits - 1)) == 0;
    const bool pair_ok = one && (!MST_L0_PACKED || ((r0.x0 - cams[__builtin_ctz(cam_bits)].ox[0]) & 3) == 0);
    if (own0 && strip && pair_ok) owned_strip<0, 1, false, NT, C_TH>(cams, a, r0, cam_bits, c1, 0, 0, a.cw[1], pano, pano_mask, nullptr);
    else if (binary && strip && pair_ok) owned_strip<0, 1, true, NT, C_TH>(cams, a, r0, cam_bits, c1, 0, 0, a.cw[1], pano, pano_mask, nullptr);
    else if (binary && strip && one) final_level<NT, false, false, W16>(cams, a, r0, cam_bits, c1, pano, pano_mask);
    else if (binary && strip) owned_strip<0, 2, true, NT, C_TH>(cams, a, r0, cam_bits, c1, 0, 0, a.cw[1], pano, pano_mask, nullptr);
    else if (own0) final_level<NT, false, true>(cams, a, r0, cam_bits, c1, pano, pano_mask);
    else final_level<NT, false, false, W16>(cams, a, r0, cam_bits, c1, pano, pano_mask);
    MST_STAMP_VAL(2, 9, static_cast<unsigned long long>(full));
    MST_STAMP(2, 1);
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// launchers + host-side geometry helpers
// ------------------------------------------------------------------------------------------------------

// occupancy experiments: extra dynamic LDS per block (bytes) from the environment, 0 when unset
static size_t tuning_env(const char* name)
{
    const char* v = std::getenv(name);
    return v ? static_cast<size_t>(std::strtoul(v, nullptr, 10)) : 0;
}

void mb_tile_dims(int* a_tw0, int* a_th0, int* b_t, int* c_tw, int* c_th)
{
    *a_tw0 = A_TW0; *a_th0 = A_TH0; *b_t = B_T; *c_tw = C_TW; *c_th = C_TH;
}

int mb_chain_depth() { return B_KMAX; }

int mb_read_stamps(int which, unsigned long long* out, size_t n)
{
#if defined(MST_STAMPS)
    if (which < 0 || which > 2 || n > (size_t{1} << 17)) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long),
                               static_cast<size_t>(which) * sizeof(unsigned long long) * (size_t{1} << 17),
                               hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
#else
    (void)which; (void)out; (void)n;
    return -3;       // not a diagnostic build
#endif
}

// What this translation unit was built with (mst_build_info): a non-zero word means a TIMING build whose pixels are wrong
// (-DMST_ABLATE_*, tools/variants.sh) or a diagnostic one (-DMST_STAMPS) -- never to be shipped or timed for the record.
uint32_t mb_build_ablations()
{
    uint32_t bits = 0;
    if (MST_ABLATE_A != 0) bits |= 1u;
    if (MST_ABLATE_A2 != 0) bits |= 2u;
    if (MST_ABLATE_C != 0) bits |= 4u;
#if defined(MST_ABLATE_STAGE) || defined(MST_ABLATE_NV12STAGE)
    bits |= 8u;
#endif
#if defined(MST_STAMPS)
    bits |= 16u;
#endif
    return bits;
}

int mb_coarse_tile_width() { return CC_TW; }

int mb_coarse_tile_height(int variant) { return variant == 0 ? 16 : 32; }

int mb_launch_variant(int frame_sets)
{
    static const int big_min = [] { const char* v = std::getenv("MST_VARIANT_MIN"); return v ? std::atoi(v) : 3; }();      // (A/B runs)
    return frame_sets < big_min ? 0 : 1;
}

int mb_fine_threads(int variant) { return variant == 0 ? 256 : 128; }

void mb_blend_region(const BlendArgs& a, int base, int tx, int ty, int level, int out[4])
{
    const Region r = blend_region(a, base, tx, ty, level);
    out[0] = r.x0; out[1] = r.y0; out[2] = r.x1; out[3] = r.y1;
}

int mb_blend_region_lds_pixels(const BlendArgs& a, int base, int tx, int ty, int level)
{
    const Region r = blend_region(a, base, tx, ty, level);
    return region_pitch(r) * (r.y1 - r.y0);
}

size_t mb_warp_table_entries(int num_tiles) { return static_cast<size_t>(num_tiles) * A_LH * A_LW; }

size_t mb_warp_blob_entries(int num_tiles) { return static_cast<size_t>(num_tiles) * A_BLOB; }

void launch_warp_table(const CamDesc* d_cams, int kind, const TileList& list, uint32_t* d_table, uint32_t* d_blobs,
                       hipStream_t s)
{
    if (list.count <= 0) return;
#define MST_LAUNCH_T(K) hipLaunchKernelGGL(k_warp_table<K>, dim3(list.count), dim3(256), 0, s, d_cams, list, d_table, d_blobs)
    if (kind == 0) MST_LAUNCH_T(0); else if (kind == 1) MST_LAUNCH_T(1); else if (kind == 2) MST_LAUNCH_T(2); else MST_LAUNCH_T(3);
#undef MST_LAUNCH_T
}

const char* launch_warp_pyr(const CamDesc* d_cams, int n_cams, int kind, const TileList& list, const FrameBatch& frames,
                            int batch, const uint32_t* d_table, const uint32_t* d_blobs, hipStream_t s)
{
    if (list.count <= 0 || batch <= 0) return nullptr;
    static const size_t pad = tuning_env("MST_A_PAD_LDS");
#define MST_LAUNCH_A(K) \
    [&] { hipLaunchKernelGGL((k_warp_pyr<K>), dim3(list.count * batch), dim3(256), pad, s, d_cams, list, frames, d_table, d_blobs, n_cams, batch); return "k_warp_pyr<" #K ">"; }()
    if (kind == 0) return MST_LAUNCH_A(0);
    if (kind == 1) return MST_LAUNCH_A(1);
    if (kind == 2) return MST_LAUNCH_A(2);
    return MST_LAUNCH_A(3);
#undef MST_LAUNCH_A
}

size_t mb_warp_tile_px() { return static_cast<size_t>(A_LH) * A_LW; }

void launch_warp_table_lds(const CamDesc* d_cams, int kind, const TileList& list, uint32_t* d_table, float* d_gain, WarpWin* d_wins,
                           int lds_budget_px, hipStream_t s)
{
    if (list.count <= 0) return;
#define MST_LAUNCH_T2(K) hipLaunchKernelGGL(k_warp_table_lds<K>, dim3(list.count), dim3(256), 0, s, d_cams, list, d_table, d_gain, d_wins, lds_budget_px)
    if (kind == 0) MST_LAUNCH_T2(0); else if (kind == 1) MST_LAUNCH_T2(1); else if (kind == 2) MST_LAUNCH_T2(2); else MST_LAUNCH_T2(3);
#undef MST_LAUNCH_T2
}

const char* launch_warp_pyr_lds(const CamDesc* d_cams, int n_cams, int n_tiles, const FrameBatch& frames, int batch,
                                const uint32_t* d_table, const float* d_gain, const WarpWin* d_wins, size_t lds_bytes, hipStream_t s)
{
    if (n_tiles <= 0 || batch <= 0) return nullptr;
    static const size_t pad = tuning_env("MST_A2_PAD_LDS");          // occupancy experiments: extra LDS per block
    const size_t lds = std::max(lds_bytes, A2_TILE_LDS) + pad;
    hipLaunchKernelGGL(k_warp_pyr_lds, dim3(n_tiles * batch), dim3(A2_NT), lds, s, d_cams, n_tiles, frames, d_table, d_gain, d_wins, n_cams, batch);
    return "k_warp_pyr_lds";
}

const char* launch_pyr_chain(const CamDesc* d_cams, int n_cams, const TileList& list, int base, int k, int batch, hipStream_t s)
{
    if (list.count <= 0 || k <= 0 || batch <= 0) return nullptr;
    if (k == 1) { hipLaunchKernelGGL(k_pyr_chain<1>, dim3(list.count, batch), dim3(B_THREADS), 0, s, d_cams, list, base, n_cams); return "k_pyr_chain<1>"; }
    hipLaunchKernelGGL(k_pyr_chain<2>, dim3(list.count, batch), dim3(B_THREADS), 0, s, d_cams, list, base, n_cams);
    return "k_pyr_chain<2>";
}

int mb_stream_unit_rows() { return PS_R; }
int mb_stream_unit_cols() { return PS_W; }

const char* launch_pyrdown_stream(const CamDesc* d_cams, int n_cams, const uint32_t* d_units, int n_units, int level, int batch, hipStream_t s)
{
    if (n_units <= 0 || batch <= 0) return nullptr;
    hipLaunchKernelGGL(k_pyrdown_stream<PS_R>, dim3((n_units + 3) / 4, batch), dim3(256), 0, s, d_cams, d_units, n_units, level, n_cams);
    static_assert(PS_R == 8, "kernel name below");
    return "k_pyrdown_stream<8>";
}

const char* launch_blend(const CamDesc* d_cams, const BlendArgs& a_in, int num_blocks, int num_coarse, int batch, size_t lds_bytes,
                         hipStream_t s, int which)
{
    static const size_t pad = tuning_env("MST_C_PAD_LDS");
    if (num_blocks <= 0 || batch <= 0) return nullptr;
    const char* name = nullptr;
    BlendArgs a = a_in;
    a.n_tiles = num_blocks;
    a.n_coarse_tiles = num_coarse;
    a.n_sets = batch;
    a.split = (a.nlev >= 3 && a.c2[0] != nullptr) ? 1 : 0;
    if (a.nlev > 1 && (which & 1)) {
        BlendArgs ac = a;
        if (a.split && a.coarse_order2) { ac.coarse_order = a.coarse_order2; ac.n_coarse_tiles = a.n_coarse2; }
        const int grid = canvas_grid<CC_XCD_GROUP>(ac.n_coarse_tiles, batch);
#define MST_LAUNCH_COARSE(W, TH, NT) [&] { hipLaunchKernelGGL((k_blend_coarse<W, TH, NT>), dim3(grid), dim3(NT), lds_bytes + pad, s, d_cams, ac); return "k_blend_coarse<" #W ", " #TH ", " #NT ">"; }()
        // (round 5 experiment, MST_CC16_THREADS=512: the 16-row tiles of launches of one or two frame sets with 512 threads -- half the
        // rounds of the band and level-1 phases per block.  Measured SLOWER on a lone call: 26.7 against 23.4 us, profiles/r05_experiments.md)
        static const int cc16_threads = [] { const char* v = std::getenv("MST_CC16_THREADS"); return v ? std::atoi(v) : 256; }();
        if (a.cc_th == 16 && cc16_threads == 512) name = a.weights_16s ? MST_LAUNCH_COARSE(true, 16, 512) : MST_LAUNCH_COARSE(false, 16, 512);
        else if (a.cc_th == 16) name = a.weights_16s ? MST_LAUNCH_COARSE(true, 16, 256) : MST_LAUNCH_COARSE(false, 16, 256);
        else name = a.weights_16s ? MST_LAUNCH_COARSE(true, 32, 256) : MST_LAUNCH_COARSE(false, 32, 256);
#undef MST_LAUNCH_COARSE
    }
    if (a.nlev > 1 && a.split && (which & 4)) {
        const int grid = canvas_grid<CM_XCD_GROUP>(num_coarse, batch);
#define MST_LAUNCH_MID(W, NT, TH) [&] { hipLaunchKernelGGL((k_blend_mid<W, NT, TH>), dim3(grid), dim3(NT), 0, s, d_cams, a); return "k_blend_mid<" #W ", " #NT ", " #TH ">"; }()
        if (a.cc_th == 16) name = a.weights_16s ? MST_LAUNCH_MID(true, 256, 16) : MST_LAUNCH_MID(false, 256, 16);
        else name = a.weights_16s ? MST_LAUNCH_MID(true, 128, 32) : MST_LAUNCH_MID(false, 128, 32);
#undef MST_LAUNCH_MID
    }
    if (!(which & 2)) return name;
    const int grid = canvas_grid<C_XCD_GROUP>(num_blocks, batch);
#define MST_LAUNCH_FINE(W, NT) [&] { hipLaunchKernelGGL((k_blend_fine<W, NT>), dim3(grid), dim3(NT), 0, s, d_cams, a); return "k_blend_fine<" #W ", " #NT ">"; }()
    if (a.fine_threads == 128) name = a.weights_16s ? MST_LAUNCH_FINE(true, 128) : MST_LAUNCH_FINE(false, 128);
    else name = a.weights_16s ? MST_LAUNCH_FINE(true, 256) : MST_LAUNCH_FINE(false, 256);
#undef MST_LAUNCH_FINE
    return name;
}

}  // namespace mst
