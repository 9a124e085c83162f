// Micro-benchmark: forms of the backward-map bilinear sampler (cv::remap INTER_LINEAR on BGR8) on gfx950.
// One synthetic camera (1920x1080 source, 1664x1024 destination, mild rotation + stretch), NSETS frame sets per
// launch sharing one per-calibration table.  Every variant must produce the same bytes (checked on the host).
//
//   col_cur    thread = column x 16 rows (kernel A's shape), 4-byte (dx,dy,fx,fy) table, mul24 addressing,
//              two aligned 12-byte loads + funnel shifts, output one dword per pixel
//   col_rel    same shape, 4-byte (fx | fy<<6 | rel<<12) table: finished byte offset relative to the tile base
//   col_relu   col_rel with two UNALIGNED 8-byte loads (no funnel shift)
//   quad_rel   thread = 4 adjacent pixels x R rows, aligned loads, 12-byte BGR store
//   quad_relu  the same with unaligned 8-byte loads
// Build: hipcc -O3 --offload-arch=gfx950 sampler.hip -o sampler ; run: ./sampler
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define GLOBAL __attribute__((address_space(1)))

constexpr int SW = 1920, SH = 1080, SSTRIDE = SW * 3;
constexpr int DW = 1664, DH = 1024;
constexpr int NSETS = 8;
constexpr int NCAM = 6;                          // cameras per set (same table, own source / destination)
constexpr int TW = 128, TH = 32;                 // destination tile
constexpr int TILES_X = DW / TW, TILES_Y = DH / TH, NTILES = TILES_X * TILES_Y;

struct Args {
    const uint8_t* src[NSETS];
    uint32_t* out[NSETS];        // dword per pixel (col_*) or BGR bytes (quad_*)
    const uint32_t* table;       // [tile][row][col]
    const float* gain;           // [tile][row][col]
    const int2* tile_base;       // per tile: smallest window coordinate (bx, by)
    int n_sets;
    size_t src_cam_bytes, out_cam_px;   // per-camera pitch inside src[] / out[]
};

__device__ __forceinline__ int xcd_run_index(int b, int n)
{
    const int q = n >> 3, r = n & 7, k = b & 7, j = b >> 3;
    return k * q + min(k, r) + j;
}

// (sum_y wy * (sum_x wx * v) + 512) >> 10 per channel, then * gain, packed
template <bool BLEND_PLAIN = false>
__device__ __forceinline__ uint32_t blend_lohi(uint32_t lo0, uint32_t hi0, uint32_t lo1, uint32_t hi1, uint32_t wx_in,
                                               uint32_t wy0s_in, uint32_t wy1s_in, float gain)
{
    // wy0s / wy1s are the vertical weights * 64: the rounded sum lands in bits 16..23 (v_cvt_f32_ubyte2)
    const uint32_t wx = wx_in, wy0s = wy0s_in, wy1s = wy1s_in;
    const uint32_t g0 = __builtin_amdgcn_alignbyte(hi0, lo0, 1), r0 = __builtin_amdgcn_alignbyte(hi0, lo0, 2);
    const uint32_t g1 = __builtin_amdgcn_alignbyte(hi1, lo1, 1), r1 = __builtin_amdgcn_alignbyte(hi1, lo1, 2);
    const uint32_t hb0 = __builtin_amdgcn_udot4(lo0, wx, 0u, false), hb1 = __builtin_amdgcn_udot4(lo1, wx, 0u, false);
    const uint32_t hg0 = __builtin_amdgcn_udot4(g0, wx, 0u, false), hg1 = __builtin_amdgcn_udot4(g1, wx, 0u, false);
    const uint32_t hr0 = __builtin_amdgcn_udot4(r0, wx, 0u, false), hr1 = __builtin_amdgcn_udot4(r1, wx, 0u, false);
    uint32_t px;
    if (BLEND_PLAIN) {
        const uint32_t b = (hb0 * (wy0s >> 6) + hb1 * (wy1s >> 6) + 512u) >> 10;
        const uint32_t g = (hg0 * (wy0s >> 6) + hg1 * (wy1s >> 6) + 512u) >> 10;
        const uint32_t r = (hr0 * (wy0s >> 6) + hr1 * (wy1s >> 6) + 512u) >> 10;
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>(b) * gain, 0u, 0u);
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>(g) * gain, 1u, px);
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>(r) * gain, 2u, px);
    } else {
        uint32_t b = __umul24(hb1, wy1s) + (__umul24(hb0, wy0s) + 32768u);
        uint32_t g = __umul24(hg1, wy1s) + (__umul24(hg0, wy0s) + 32768u);
        uint32_t r = __umul24(hr1, wy1s) + (__umul24(hr0, wy0s) + 32768u);
        // (all 32 bits "used": when only bits 16..23 are demanded the 24-bit multiplies are rewritten as plain 32-bit ones
        // and selected as the quarter-rate v_mul_lo_u32)
        asm("" : "+v"(b)); asm("" : "+v"(g)); asm("" : "+v"(r));
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((b >> 16) & 0xFFu) * gain, 0u, 0u);
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((g >> 16) & 0xFFu) * gain, 1u, px);
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((r >> 16) & 0xFFu) * gain, 2u, px);
    }
    return px;
}

typedef uint32_t __attribute__((ext_vector_type(2))) u32x2;
typedef uint32_t __attribute__((ext_vector_type(3))) u32x3;
typedef u32x2 __attribute__((aligned(1))) U8x8;     // an 8-byte load from any byte address

// ---- col_cur: kernel A's present decode -------------------------------------------------------------------------
template <bool PLAIN>
__global__ __launch_bounds__(256) void k_col_cur(Args a)
{
    const int pair = xcd_run_index(blockIdx.x, NCAM * NTILES * a.n_sets);
    const int set = pair % a.n_sets, ctile = pair / a.n_sets, cam = ctile / NTILES, tile = ctile % NTILES;
    const int tx = tile % TILES_X, ty = tile / TILES_X;
    const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
    const uint32_t GLOBAL* tab = (const uint32_t GLOBAL*)a.table + (size_t)tile * TW * TH + col;
    const float GLOBAL* gt = (const float GLOBAL*)a.gain + (size_t)tile * TW * TH + col;
    const int2 tb = a.tile_base[tile];
    const uint8_t* src = a.src[set] + cam * a.src_cam_bytes;
    uint32_t GLOBAL* out = (uint32_t GLOBAL*)a.out[set] + cam * a.out_cam_px + (size_t)(ty * TH) * DW + tx * TW + col;
#pragma unroll
    for (int it = 0; it < TH / 2; ++it) {
        const int row = half * (TH / 2) + it;
        const uint32_t e = tab[row * TW];
        const float gain = gt[row * TW];
        const int sx = tb.x + (int)(e & 1023u), sy = tb.y + (int)((e >> 10) & 511u);
        const int fx = (e >> 19) & 63u, fy = (e >> 25) & 63u;
        const uint32_t wx = (uint32_t)(32 - fx) | (uint32_t)fx << 24;
        const uint32_t o0 = __umul24(sy, SSTRIDE) + sx * 3u, o1 = o0 + SSTRIDE;
        const uint32_t* q0 = (const uint32_t*)(src + (o0 & ~3u));
        const uint32_t* q1 = (const uint32_t*)(src + (o1 & ~3u));
        const uint32_t a0 = q0[0], a1 = q0[1], a2 = q0[2], c0 = q1[0], c1 = q1[1], c2 = q1[2];
        const uint32_t lo0 = __builtin_amdgcn_alignbyte(a1, a0, o0 & 3u), hi0 = __builtin_amdgcn_alignbyte(a2, a1, o0 & 3u);
        const uint32_t lo1 = __builtin_amdgcn_alignbyte(c1, c0, o1 & 3u), hi1 = __builtin_amdgcn_alignbyte(c2, c1, o1 & 3u);
        out[row * DW] = blend_lohi<PLAIN>(lo0, hi0, lo1, hi1, wx, (32 - fy) * 64, fy * 64, gain);
    }
}

// ---- rel decode ---------------------------------------------------------------------------------------------------
// entry = fx | fy << 6 | rel << 12, rel = byte offset of the window's first pixel from the (4-byte aligned) tile base
struct Dec { uint32_t wx, wy0s, wy1s, rel; };
__device__ __forceinline__ Dec decode_rel(uint32_t e)
{
    Dec d;
    const uint32_t fx = e & 63u;
    d.wx = __umul24(fx, 0xFFFFFFu) + 32u;            // (32 - fx) | fx << 24
    d.wy1s = e & 0xFC0u;                             // fy * 64
    d.wy0s = 2048u - d.wy1s;
    d.rel = e >> 12;
    return d;
}

template <bool UNALIGNED, bool MASK_SHIFT = true>
__device__ __forceinline__ void fetch(const uint8_t GLOBAL* base0, const uint8_t GLOBAL* base1, uint32_t rel, uint32_t& lo0,
                                      uint32_t& hi0, uint32_t& lo1, uint32_t& hi1)
{
    if (UNALIGNED) {
        const U8x8 p0 = *(const U8x8 GLOBAL*)(base0 + rel);
        const U8x8 p1 = *(const U8x8 GLOBAL*)(base1 + rel);
        lo0 = p0.x; hi0 = p0.y; lo1 = p1.x; hi1 = p1.y;
    } else {
        const uint32_t off = rel & ~3u;
        const u32x3 p0 = *(const u32x3 GLOBAL*)(base0 + off);
        const u32x3 p1 = *(const u32x3 GLOBAL*)(base1 + off);
        const uint32_t sh = MASK_SHIFT ? (rel & 3u) : rel;
        lo0 = __builtin_amdgcn_alignbyte(p0.y, p0.x, sh); hi0 = __builtin_amdgcn_alignbyte(p0.z, p0.y, sh);
        lo1 = __builtin_amdgcn_alignbyte(p1.y, p1.x, sh); hi1 = __builtin_amdgcn_alignbyte(p1.z, p1.y, sh);
    }
}

template <bool UNALIGNED, bool MASK_SHIFT = true, int ABL = 0>
__global__ __launch_bounds__(256) void k_col_rel(Args a)
{
    const int pair = xcd_run_index(blockIdx.x, NCAM * NTILES * a.n_sets);
    const int set = pair % a.n_sets, ctile = pair / a.n_sets, cam = ctile / NTILES, tile = ctile % NTILES;
    const int tx = tile % TILES_X, ty = tile / TILES_X;
    const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
    const uint32_t GLOBAL* tab = (const uint32_t GLOBAL*)a.table + (size_t)tile * TW * TH + col;
    const float GLOBAL* gt = (const float GLOBAL*)a.gain + (size_t)tile * TW * TH + col;
    const int2 tb = a.tile_base[tile];
    const uint32_t tbase = (uint32_t)tb.y * SSTRIDE + (uint32_t)tb.x * 3u;
    const uint8_t GLOBAL* base0 = (const uint8_t GLOBAL*)a.src[set] + cam * a.src_cam_bytes + (tbase & ~3u);
    const uint8_t GLOBAL* base1 = base0 + SSTRIDE;
    uint32_t GLOBAL* out = (uint32_t GLOBAL*)a.out[set] + cam * a.out_cam_px + (size_t)(ty * TH) * DW + tx * TW + col;
    uint32_t late[TH / 2];
#pragma unroll
    for (int it = 0; it < TH / 2; ++it) {
        const int row = half * (TH / 2) + it;
        const Dec d = decode_rel((ABL & 2) ? (uint32_t)((col * 3 + row * SSTRIDE) << 12 | (col & 31) | (row & 31) << 6) : tab[row * TW]);
        const float gain = (ABL & 4) ? 1.0f + col * 0.001f : gt[row * TW];
        uint32_t lo0, hi0, lo1, hi1;
        if (ABL & 8) { lo0 = d.rel * 7u; hi0 = d.rel ^ 0x1234567u; lo1 = d.rel * 13u; hi1 = d.rel + col; }
        else fetch<UNALIGNED, MASK_SHIFT>(base0, base1, d.rel, lo0, hi0, lo1, hi1);
        const uint32_t px = blend_lohi(lo0, hi0, lo1, hi1, d.wx, d.wy0s, d.wy1s, gain);
        if (ABL & 16) late[it] = px;
        else if (!(ABL & 1) || px == 0x12345678u) out[row * DW] = px;
    }
    if (ABL & 16) {
#pragma unroll
        for (int it = 0; it < TH / 2; ++it) out[(half * (TH / 2) + it) * DW] = late[it];
    }
}

// ---- quad: 4 adjacent pixels per lane, 12-byte BGR store -------------------------------------------------------
// table layout for this shape: [tile][row][k = 0..3][lane 0..31]: lane-contiguous dwords per pixel slot
template <bool UNALIGNED, int ROWS, bool LATE = false>
__global__ __launch_bounds__(256) void k_quad_rel(Args a)
{
    const int pair = xcd_run_index(blockIdx.x, NCAM * NTILES * a.n_sets);
    const int set = pair % a.n_sets, ctile = pair / a.n_sets, cam = ctile / NTILES, tile = ctile % NTILES;
    const int tx = tile % TILES_X, ty = tile / TILES_X;
    const int lane = threadIdx.x & 31, rgrp = threadIdx.x >> 5;       // 32 lanes x 4 px = 128 columns; 8 row groups
    const int2 tb = a.tile_base[tile];
    const uint32_t tbase = (uint32_t)tb.y * SSTRIDE + (uint32_t)tb.x * 3u;
    const uint8_t GLOBAL* base0 = (const uint8_t GLOBAL*)a.src[set] + cam * a.src_cam_bytes + (tbase & ~3u);
    const uint8_t GLOBAL* base1 = base0 + SSTRIDE;
    typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
    typedef float __attribute__((ext_vector_type(4))) f32x4;
    const u32x4 GLOBAL* tab = (const u32x4 GLOBAL*)a.table + (size_t)tile * (TW / 4) * TH + lane;
    const f32x4 GLOBAL* gt = (const f32x4 GLOBAL*)a.gain + (size_t)tile * (TW / 4) * TH + lane;
    uint8_t GLOBAL* out = (uint8_t GLOBAL*)a.out[set] + cam * a.out_cam_px * 4 + ((size_t)(ty * TH) * DW + tx * TW + lane * 4) * 3;
    u32x3 lw[ROWS];
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        const int row = rgrp * ROWS + it;
        const u32x4 e = tab[row * (TW / 4)];
        const f32x4 g = gt[row * (TW / 4)];
        uint32_t px[4];
        uint32_t lo0[4], hi0[4], lo1[4], hi1[4];
        Dec d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            d[k] = decode_rel(e[k]);
            fetch<UNALIGNED>(base0, base1, d[k].rel, lo0[k], hi0[k], lo1[k], hi1[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) px[k] = blend_lohi(lo0[k], hi0[k], lo1[k], hi1[k], d[k].wx, d[k].wy0s, d[k].wy1s, g[k]);
        u32x3 w;
        w.x = px[0] | px[1] << 24;
        w.y = px[1] >> 8 | px[2] << 16;
        w.z = px[2] >> 16 | px[3] << 8;
        if (LATE) lw[it] = w;
        else *(u32x3 GLOBAL*)(out + (size_t)row * DW * 3) = w;
    }
    if (LATE) {
#pragma unroll
        for (int it = 0; it < ROWS; ++it) *(u32x3 GLOBAL*)(out + (size_t)(rgrp * ROWS + it) * DW * 3) = lw[it];
    }
}


// ---- col_lds: the tile's source window staged in LDS as 4-byte pixels, sampled from there -------------------------
// Per tile (host / calibration): window origin (wx0 a multiple of 4 pixels, so that its rows start on a dword of an aligned
// frame), size in groups of 4 pixels and rows, LDS pitch in pixels.  Table entry: byte address of the window's top-left
// pixel in LDS | fx << 16 | fy << 24.
struct TileWin { int x0, y0, groups, rows, pitch, pad0, pad1, pad2; };
constexpr int LDS_PX = 6144;                     // 24 KB of staged pixels per block
typedef unsigned short __attribute__((ext_vector_type(2))) us2;
__device__ __forceinline__ uint32_t dot2(uint32_t a, uint32_t b, uint32_t c)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b), c, false);
}
template <int ABL = 0>
__global__ __launch_bounds__(256) void k_col_lds(Args a, const TileWin* __restrict__ wins)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_src[LDS_PX];
    const int pair = xcd_run_index(blockIdx.x, NCAM * NTILES * a.n_sets);
    const int set = pair % a.n_sets, ctile = pair / a.n_sets, cam = ctile / NTILES, tile = ctile % NTILES;
    const int tx = tile % TILES_X, ty = tile / TILES_X;
    const int col = threadIdx.x & 127, half = threadIdx.x >> 7;
    const uint32_t GLOBAL* tab = (const uint32_t GLOBAL*)a.table + (size_t)tile * TW * TH + col;
    const float GLOBAL* gt = (const float GLOBAL*)a.gain + (size_t)tile * TW * TH + col;
    const TileWin w = wins[tile];
    // stage: groups of 4 pixels = 12 source bytes (dword aligned) -> 16 LDS bytes
    {
        const uint8_t GLOBAL* base = (const uint8_t GLOBAL*)a.src[set] + cam * a.src_cam_bytes + (uint32_t)w.y0 * SSTRIDE + (uint32_t)w.x0 * 3u;
        const int n = w.groups * w.rows;
        typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
        for (int i = threadIdx.x; i < ((ABL & 8) ? 0 : n); i += 256) {
            const int r = i / w.groups, g = i - r * w.groups;
            const u32x3 d = *(const u32x3 GLOBAL*)(base + (uint32_t)r * SSTRIDE + (uint32_t)g * 12u);
            u32x4 p;
            p.x = d.x;                                            // B0 G0 R0 (B1)
            p.y = __builtin_amdgcn_alignbyte(d.y, d.x, 3);        // B1 G1 R1 (B2)
            p.z = __builtin_amdgcn_alignbyte(d.z, d.y, 2);        // B2 G2 R2 (B3)
            p.w = d.z >> 8;                                       // B3 G3 R3 0
            *(u32x4*)&s_src[r * w.pitch + 4 * g] = p;
        }
    }
    __syncthreads();
    uint32_t GLOBAL* out = (uint32_t GLOBAL*)a.out[set] + cam * a.out_cam_px + (size_t)(ty * TH) * DW + tx * TW + col;
    const uint32_t pitch_b = (uint32_t)w.pitch * 4u;
#pragma unroll
    for (int it = 0; it < TH / 2; ++it) {
        const int row = half * (TH / 2) + it;
        const uint32_t e = (ABL & 2) ? (uint32_t)((col + row * w.pitch / 2) * 4 | (col & 31) << 16 | (row & 31) << 24) : tab[row * TW];
        const float gain = (ABL & 4) ? 1.0f + col * 0.001f : gt[row * TW];
        const uint32_t addr = e & 0xFFFFu, fx = (e >> 16) & 63u, fy = e >> 24;
        const uint32_t wxp = __umul24(fx, 0x3FFFC0u) + 2048u;     // (32 - fx) * 64 | fx * 64 << 16
        const uint32_t wy1 = fy, wy0 = 32u - fy;
        const uint32_t* p0 = (const uint32_t*)((const char*)s_src + addr);
        const uint32_t* p1 = (const uint32_t*)((const char*)s_src + addr + pitch_b);
        const uint32_t lo0 = p0[0], hi0 = p0[1], lo1 = p1[0], hi1 = p1[1];
        // vertical pass in 16-bit lanes (8 bits x 6 bits), two channels or two pixels per register
        const uint32_t br0 = __umul24(lo0 & 0xFF00FFu, wy0) + __umul24(lo1 & 0xFF00FFu, wy1);      // VB0 | VR0 << 16
        const uint32_t br1 = __umul24(hi0 & 0xFF00FFu, wy0) + __umul24(hi1 & 0xFF00FFu, wy1);      // VB1 | VR1 << 16
        const uint32_t gt0 = __builtin_amdgcn_perm(hi0, lo0, 0x0C050C01u), gb0 = __builtin_amdgcn_perm(hi1, lo1, 0x0C050C01u);
        const uint32_t gg = __umul24(gt0, wy0) + __umul24(gb0, wy1);                                // VG0 | VG1 << 16
        // horizontal pass: u16 pairs x (weights * 64) + 32768 -> the byte sits in bits 16..23
        uint32_t b = dot2(__builtin_amdgcn_perm(br1, br0, 0x05040100u), wxp, 32768u);
        uint32_t r = dot2(__builtin_amdgcn_perm(br1, br0, 0x07060302u), wxp, 32768u);
        uint32_t g = dot2(gg, wxp, 32768u);
        asm("" : "+v"(b)); asm("" : "+v"(g)); asm("" : "+v"(r));
        uint32_t px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((b >> 16) & 0xFFu) * gain, 0u, 0u);
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((g >> 16) & 0xFFu) * gain, 1u, px);
        px = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((r >> 16) & 0xFFu) * gain, 2u, px);
        if (!(ABL & 1) || px == 0x12345678u) out[row * DW] = px;
    }
}


// ---- quad_lds: the LDS-staged window sampled four pixels per lane, 12-byte BGR store ---------------------------------
template <int ROWS>
__global__ __launch_bounds__(256) void k_quad_lds(Args a, const TileWin* __restrict__ wins)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_src[LDS_PX];
    const int pair = xcd_run_index(blockIdx.x, NCAM * NTILES * a.n_sets);
    const int set = pair % a.n_sets, ctile = pair / a.n_sets, cam = ctile / NTILES, tile = ctile % NTILES;
    const int tx = tile % TILES_X, ty = tile / TILES_X;
    const int lane = threadIdx.x & 31, rgrp = threadIdx.x >> 5;
    typedef uint32_t __attribute__((ext_vector_type(4))) u32x4;
    typedef float __attribute__((ext_vector_type(4))) f32x4;
    const u32x4 GLOBAL* tab = (const u32x4 GLOBAL*)a.table + (size_t)tile * (TW / 4) * TH + lane;
    const f32x4 GLOBAL* gt = (const f32x4 GLOBAL*)a.gain + (size_t)tile * (TW / 4) * TH + lane;
    const TileWin w = wins[tile];
    {
        const uint8_t GLOBAL* base = (const uint8_t GLOBAL*)a.src[set] + cam * a.src_cam_bytes + (uint32_t)w.y0 * SSTRIDE + (uint32_t)w.x0 * 3u;
        const int n = w.groups * w.rows;
        for (int i = threadIdx.x; i < n; i += 256) {
            const int r = i / w.groups, g = i - r * w.groups;
            const u32x3 d = *(const u32x3 GLOBAL*)(base + (uint32_t)r * SSTRIDE + (uint32_t)g * 12u);
            u32x4 p;
            p.x = d.x;
            p.y = __builtin_amdgcn_alignbyte(d.y, d.x, 3);
            p.z = __builtin_amdgcn_alignbyte(d.z, d.y, 2);
            p.w = d.z >> 8;
            *(u32x4*)&s_src[r * w.pitch + 4 * g] = p;
        }
    }
    // table and gain rows of this lane do not depend on the staging: issue them before the barrier
    u32x4 e[ROWS]; f32x4 gn[ROWS];
#pragma unroll
    for (int it = 0; it < ROWS; ++it) { e[it] = tab[(rgrp * ROWS + it) * (TW / 4)]; gn[it] = gt[(rgrp * ROWS + it) * (TW / 4)]; }
    __syncthreads();
    uint8_t GLOBAL* out = (uint8_t GLOBAL*)a.out[set] + cam * a.out_cam_px * 4 + ((size_t)(ty * TH) * DW + tx * TW + lane * 4) * 3;
    const uint32_t pitch_b = (uint32_t)w.pitch * 4u;
#pragma unroll
    for (int it = 0; it < ROWS; ++it) {
        const int row = rgrp * ROWS + it;
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t ee = e[it][k];
            const uint32_t addr = ee & 0xFFFFu, fx = (ee >> 16) & 63u, fy = ee >> 24;
            const uint32_t wxp = __umul24(fx, 0x3FFFC0u) + 2048u;
            const uint32_t wy1 = fy, wy0 = 32u - fy;
            const uint32_t* p0 = (const uint32_t*)((const char*)s_src + addr);
            const uint32_t* p1 = (const uint32_t*)((const char*)s_src + addr + pitch_b);
            const uint32_t lo0 = p0[0], hi0 = p0[1], lo1 = p1[0], hi1 = p1[1];
            const uint32_t br0 = __umul24(lo0 & 0xFF00FFu, wy0) + __umul24(lo1 & 0xFF00FFu, wy1);
            const uint32_t br1 = __umul24(hi0 & 0xFF00FFu, wy0) + __umul24(hi1 & 0xFF00FFu, wy1);
            const uint32_t gt0 = __builtin_amdgcn_perm(hi0, lo0, 0x0C050C01u), gb0 = __builtin_amdgcn_perm(hi1, lo1, 0x0C050C01u);
            const uint32_t gg = __umul24(gt0, wy0) + __umul24(gb0, wy1);
            uint32_t b = dot2(__builtin_amdgcn_perm(br1, br0, 0x05040100u), wxp, 32768u);
            uint32_t r = dot2(__builtin_amdgcn_perm(br1, br0, 0x07060302u), wxp, 32768u);
            uint32_t g = dot2(gg, wxp, 32768u);
            asm("" : "+v"(b)); asm("" : "+v"(g)); asm("" : "+v"(r));
            uint32_t q = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((b >> 16) & 0xFFu) * gn[it][k], 0u, 0u);
            q = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((g >> 16) & 0xFFu) * gn[it][k], 1u, q);
            q = __builtin_amdgcn_cvt_pk_u8_f32(static_cast<float>((r >> 16) & 0xFFu) * gn[it][k], 2u, q);
            px[k] = q;
        }
        u32x3 wv;
        wv.x = px[0] | px[1] << 24;
        wv.y = px[1] >> 8 | px[2] << 16;
        wv.z = px[2] >> 16 | px[3] << 8;
        *(u32x3 GLOBAL*)(out + (size_t)row * DW * 3) = wv;
    }
}

// -------------------------------------------------------------------------------------------------------------------
static int cv_round(float v) { return (int)lrintf(v); }

int main()
{
    // synthetic map: rotation by 2 degrees about the centre, stretch 1.08 x 1.02, offset
    std::vector<float> mx((size_t)DW * DH), my((size_t)DW * DH);
    const float ang = 2.0f * 3.14159265f / 180.f, ca = cosf(ang), sa = sinf(ang);
    for (int y = 0; y < DH; ++y)
        for (int x = 0; x < DW; ++x) {
            const float u = (x - DW / 2) * 1.08f, v = (y - DH / 2) * 1.02f;
            mx[(size_t)y * DW + x] = SW / 2 + ca * u - sa * v + 3.3f;
            my[(size_t)y * DW + x] = SH / 2 + sa * u + ca * v - 1.7f;
        }
    // tables
    std::vector<uint32_t> t_cur((size_t)DW * DH), t_rel((size_t)DW * DH), t_quad((size_t)DW * DH);
    std::vector<float> g_col((size_t)DW * DH), g_quad((size_t)DW * DH);
    std::vector<int2> tbase(NTILES);
    std::vector<int> SX((size_t)DW * DH), SY((size_t)DW * DH), FX((size_t)DW * DH), FY((size_t)DW * DH);
    for (int t = 0; t < NTILES; ++t) {
        const int tx = t % TILES_X, ty = t / TILES_X;
        int bx = 1 << 30, by = 1 << 30;
        for (int r = 0; r < TH; ++r)
            for (int c = 0; c < TW; ++c) {
                const size_t i = (size_t)(ty * TH + r) * DW + tx * TW + c;
                const int fsx = cv_round(mx[i] * 32.f), fsy = cv_round(my[i] * 32.f);
                int sx = fsx >> 5, sy = fsy >> 5;
                sx = std::min(std::max(sx, 0), SW - 4); sy = std::min(std::max(sy, 0), SH - 3);   // keep every window on the fast path
                SX[i] = sx; SY[i] = sy; FX[i] = fsx & 31; FY[i] = fsy & 31;
                bx = std::min(bx, sx); by = std::min(by, sy);
            }
        tbase[t] = make_int2(bx, by);
        const uint32_t tb = (uint32_t)by * SSTRIDE + (uint32_t)bx * 3u;
        for (int r = 0; r < TH; ++r)
            for (int c = 0; c < TW; ++c) {
                const size_t i = (size_t)(ty * TH + r) * DW + tx * TW + c;
                const size_t ti = (size_t)t * TW * TH + (size_t)r * TW + c;
                const uint32_t dx = SX[i] - bx, dy = SY[i] - by;
                if (dx >= 1024 || dy >= 512) { printf("window too far\n"); return 1; }
                t_cur[ti] = dx | dy << 10 | (uint32_t)FX[i] << 19 | (uint32_t)FY[i] << 25;
                const uint32_t rel = dy * SSTRIDE + dx * 3 + (tb & 3u);
                if (rel >= (1u << 20)) { printf("rel too large\n"); return 1; }
                t_rel[ti] = (uint32_t)FX[i] | (uint32_t)FY[i] << 6 | rel << 12;
                const float gain = 1.0f + 0.1f * sinf(0.01f * (tx * TW + c)) * cosf(0.013f * (ty * TH + r));
                g_col[ti] = gain;
                const size_t qi = (size_t)t * TW * TH + (size_t)r * TW + (size_t)(c / 4) * 4 + (c & 3);   // lane-major: [row][lane][k]
                t_quad[qi] = t_rel[ti];
                g_quad[qi] = gain;
            }
    }
    std::vector<uint8_t> src((size_t)SSTRIDE * SH + 64);
    uint32_t s = 12345;
    for (auto& b : src) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }

    Args a{};
    uint8_t* d_src[NSETS]; uint32_t* d_out[NSETS];
    for (int k = 0; k < NSETS; ++k) {
        CHECK(hipMalloc(&d_src[k], src.size() * NCAM));
        for (int c = 0; c < NCAM; ++c) CHECK(hipMemcpy(d_src[k] + c * src.size(), src.data(), src.size(), hipMemcpyHostToDevice));
        CHECK(hipMalloc(&d_out[k], (size_t)DW * DH * 4 * NCAM));
        a.src[k] = d_src[k]; a.out[k] = d_out[k];
    }
    uint32_t *d_cur, *d_rel, *d_quad; float *d_gcol, *d_gquad; int2* d_tb;
    CHECK(hipMalloc(&d_cur, t_cur.size() * 4)); CHECK(hipMalloc(&d_rel, t_rel.size() * 4)); CHECK(hipMalloc(&d_quad, t_quad.size() * 4));
    CHECK(hipMalloc(&d_gcol, g_col.size() * 4)); CHECK(hipMalloc(&d_gquad, g_quad.size() * 4)); CHECK(hipMalloc(&d_tb, NTILES * sizeof(int2)));
    CHECK(hipMemcpy(d_cur, t_cur.data(), t_cur.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_rel, t_rel.data(), t_rel.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_quad, t_quad.data(), t_quad.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_gcol, g_col.data(), g_col.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_gquad, g_quad.data(), g_quad.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tb, tbase.data(), NTILES * sizeof(int2), hipMemcpyHostToDevice));
    a.tile_base = d_tb; a.n_sets = NSETS; a.src_cam_bytes = src.size(); a.out_cam_px = (size_t)DW * DH;

    // CPU reference of one pixel (cv::remap's fixed-point bilinear + gain)
    auto cpu_px = [&](size_t i) -> uint32_t {
        const int x = i % DW, y = i / DW, tx = x / TW, ty = y / TH, t = ty * TILES_X + tx;
        const float gain = g_col[(size_t)t * TW * TH + (size_t)(y % TH) * TW + (x % TW)];
        uint32_t px = 0;
        for (int ch = 0; ch < 3; ++ch) {
            const uint8_t* p = src.data() + (size_t)SY[i] * SSTRIDE + SX[i] * 3 + ch;
            const int fx = FX[i], fy = FY[i];
            const int v = (p[0] * (32 - fx) * (32 - fy) * 32 + p[3] * fx * (32 - fy) * 32 + p[SSTRIDE] * (32 - fx) * fy * 32 + p[SSTRIDE + 3] * fx * fy * 32 + (1 << 14)) >> 15;
            float f = nearbyintf((float)v * gain);
            f = f < 0 ? 0 : (f > 255 ? 255 : f);
            px |= (uint32_t)f << (8 * ch);
        }
        return px;
    };

    // LDS-staged variant: per tile window (x0 rounded down to a multiple of 4 pixels) and the table of LDS addresses
    std::vector<TileWin> wins(NTILES);
    std::vector<uint32_t> t_lds((size_t)DW * DH);
    for (int t = 0; t < NTILES; ++t) {
        const int tx = t % TILES_X, ty = t / TILES_X;
        int x0 = 1 << 30, y0 = 1 << 30, x1 = 0, y1 = 0;
        for (int r = 0; r < TH; ++r)
            for (int c = 0; c < TW; ++c) {
                const size_t i = (size_t)(ty * TH + r) * DW + tx * TW + c;
                x0 = std::min(x0, SX[i]); y0 = std::min(y0, SY[i]); x1 = std::max(x1, SX[i] + 2); y1 = std::max(y1, SY[i] + 2);
            }
        x0 &= ~3;
        TileWin w{};
        w.x0 = x0; w.y0 = y0; w.groups = (x1 - x0 + 3) / 4; w.rows = y1 - y0; w.pitch = w.groups * 4;
        if (w.pitch * w.rows > LDS_PX) { printf("tile %d: window %d x %d does not fit LDS\n", t, w.pitch, w.rows); return 1; }
        wins[t] = w;
        for (int r = 0; r < TH; ++r)
            for (int c = 0; c < TW; ++c) {
                const size_t i = (size_t)(ty * TH + r) * DW + tx * TW + c;
                const size_t ti = (size_t)t * TW * TH + (size_t)r * TW + c;
                const uint32_t addr = ((SY[i] - y0) * w.pitch + (SX[i] - x0)) * 4;
                t_lds[ti] = addr | (uint32_t)FX[i] << 16 | (uint32_t)FY[i] << 24;
            }
    }
    printf("window of tile 0: %d groups x %d rows (pitch %d px)\n", wins[0].groups, wins[0].rows, wins[0].pitch);
    uint32_t* d_lds; TileWin* d_wins;
    CHECK(hipMalloc(&d_lds, t_lds.size() * 4)); CHECK(hipMalloc(&d_wins, NTILES * sizeof(TileWin)));
    CHECK(hipMemcpy(d_lds, t_lds.data(), t_lds.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_wins, wins.data(), NTILES * sizeof(TileWin), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<uint32_t> ref((size_t)DW * DH), got((size_t)DW * DH);
    std::vector<uint8_t> got8((size_t)DW * DH * 3);
    bool have_ref = false;
    auto time_it = [&](const char* name, auto launch, bool bgr) -> int {
        for (int k = 0; k < NSETS; ++k) hipMemset(d_out[k], 0, (size_t)DW * DH * 4 * NCAM);
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipGetLastError());
        const int reps = getenv("SAMPLER_REPS") ? atoi(getenv("SAMPLER_REPS")) : 40;
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / reps, px = (double)DW * DH * a.n_sets * NCAM;
        const int cs = std::min(3, a.n_sets - 1);
        // check set 3 against the first variant
        size_t bad = 0;
        if (!bgr) {
            CHECK(hipMemcpy(got.data(), d_out[cs] + (size_t)(NCAM - 1) * DW * DH, got.size() * 4, hipMemcpyDeviceToHost));
            if (!have_ref) { for (size_t i = 0; i < ref.size(); ++i) ref[i] = cpu_px(i); have_ref = true; }
            for (size_t i = 0; i < got.size(); ++i) {
                if (got[i] != ref[i] && bad == 0) printf("   first mismatch at %zu (x %zu y %zu): got %08x ref %08x cpu %08x\n", i, i % DW, i / DW, got[i], ref[i], cpu_px(i));
                bad += got[i] != ref[i];
            }
        } else {
            CHECK(hipMemcpy(got8.data(), (uint8_t*)d_out[cs] + (size_t)(NCAM - 1) * DW * DH * 4, got8.size(), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ref.size(); ++i)
                bad += got8[3 * i] != (ref[i] & 0xFF) || got8[3 * i + 1] != ((ref[i] >> 8) & 0xFF) || got8[3 * i + 2] != ((ref[i] >> 16) & 0xFF);
        }
        printf("%-14s %8.2f us per launch of %d sets  %7.1f Gpx/s  %6.2f us per 10.2 Mpx set   mismatches %zu\n", name, us, a.n_sets,
               px / us * 1e-3, us / a.n_sets, bad);
        return 0;
    };
    const dim3 grid(NCAM * NTILES * NSETS), block(256);
    a.table = d_cur; a.gain = d_gcol;
    time_it("col_cur plain", [&] { hipLaunchKernelGGL(k_col_cur<true>, grid, block, 0, 0, a); }, false);
    time_it("col_cur", [&] { hipLaunchKernelGGL(k_col_cur<false>, grid, block, 0, 0, a); }, false);
    a.table = d_rel;
    time_it("col_rel", [&] { hipLaunchKernelGGL(k_col_rel<false>, grid, block, 0, 0, a); }, false);
    time_it("col_rel nomask", [&] { hipLaunchKernelGGL((k_col_rel<false, false>), grid, block, 0, 0, a); }, false);
    time_it("col_rel late", [&] { hipLaunchKernelGGL((k_col_rel<false, true, 16>), grid, block, 0, 0, a); }, false);
    time_it("col_relu", [&] { hipLaunchKernelGGL(k_col_rel<true>, grid, block, 0, 0, a); }, false);
    a.table = d_lds; a.gain = d_gcol;
    time_it("col_lds", [&] { hipLaunchKernelGGL(k_col_lds<0>, grid, block, 0, 0, a, d_wins); }, false);
    a.table = d_lds; a.gain = d_gcol;
    time_it("quad_lds/4", [&] { hipLaunchKernelGGL(k_quad_lds<4>, grid, block, 0, 0, a, d_wins); }, true);
    a.table = d_quad; a.gain = d_gquad;
    time_it("quad_rel/4", [&] { hipLaunchKernelGGL((k_quad_rel<false, 4>), grid, block, 0, 0, a); }, true);
    time_it("quad_rel/4 late", [&] { hipLaunchKernelGGL((k_quad_rel<false, 4, true>), grid, block, 0, 0, a); }, true);
    time_it("quad_relu/4", [&] { hipLaunchKernelGGL((k_quad_rel<true, 4>), grid, block, 0, 0, a); }, true);
    // one frame set per launch (latency mode: the table comes from HBM every time)
    a.n_sets = 1;
    const dim3 grid1(NCAM * NTILES);
    a.table = d_rel; a.gain = d_gcol;
    time_it("col_relu x1", [&] { hipLaunchKernelGGL(k_col_rel<true>, grid1, block, 0, 0, a); }, false);
    a.table = d_lds; a.gain = d_gcol;
    time_it("col_lds x1", [&] { hipLaunchKernelGGL(k_col_lds<0>, grid1, block, 0, 0, a, d_wins); }, false);
    a.table = d_lds; a.gain = d_gcol;
    time_it("quad_lds x1", [&] { hipLaunchKernelGGL(k_quad_lds<4>, grid1, block, 0, 0, a, d_wins); }, true);
    a.table = d_quad; a.gain = d_gquad;
    time_it("quad_relu x1", [&] { hipLaunchKernelGGL((k_quad_rel<true, 4>), grid1, block, 0, 0, a); }, true);
    if (getenv("SAMPLER_ABLATE")) {
        printf("ablations (results are wrong on purpose): 1 no store, 2 no table load, 4 no gain load, 8 no gather / staging\n");
        a.n_sets = NSETS; a.table = d_rel; a.gain = d_gcol;
#define ABL_REL(N) time_it("col_rel abl " #N, [&] { hipLaunchKernelGGL((k_col_rel<false, true, N>), grid, block, 0, 0, a); }, false)
        ABL_REL(1); ABL_REL(2); ABL_REL(4); ABL_REL(6); ABL_REL(8); ABL_REL(7); ABL_REL(14); ABL_REL(15);
        a.table = d_lds;
#define ABL_LDS(N) time_it("col_lds abl " #N, [&] { hipLaunchKernelGGL(k_col_lds<N>, grid, block, 0, 0, a, d_wins); }, false)
        ABL_LDS(1); ABL_LDS(2); ABL_LDS(4); ABL_LDS(6); ABL_LDS(8); ABL_LDS(7); ABL_LDS(14); ABL_LDS(15);
    }
    return 0;
}
