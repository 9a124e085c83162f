// dp_seam.hip -- the dynamic-programming seam finder (detail::DpSeamFinder,
// R64/src/seam_finders.cpp:163-1045; --seam dp_color / dp_colorgrad, CLI:627-630), per calibration.
//
// The class is a sequential piece of geometry bookkeeping (connected components in discovery order, a std::set of
// component edges walked in order, a conflict loop that rewrites labels) around four data-parallel passes.  The passes
// run on the device, the bookkeeping on the host -- its order IS the result, a parallel formulation would have to
// reproduce the discovery order of every flood fill:
//
//   k_dp_union        union-rectangle masks, their contour masks (:199-232), the pixel class for findComponents (:261-274)
//   k_dp_gray/sobel   computeGradients (:519-543): cvtColor(BGR2GRAY) + Sobel 3x3 in the x64 build's float order
//   k_dp_costs        computeCosts (:703-778): costV / costH of one intersection component
//   k_dp_sweep        estimateSeam's recurrence (:822-869): one step per row (column), every cell of the step in parallel
//   k_dp_apply        the final mask update (:489-515)
//
// Masks stay on the device across pairs (a pair sees what the pairs before it carved, as in the reference); labels
// live on the host and travel per pair (classes down, labels of a component's box up for the costs, all of them up
// once more for the mask update).
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <utility>
#include <vector>

#include "engine.hpp"
#include "kernels_common.hpp"

#pragma clang fp contract(off)

namespace mst {

using namespace dev;

namespace {

struct DpGeom {
    int utlx, utly, uw, uh;          // union rectangle
    int w1, h1, dx1, dy1;            // image 1 size, unionTl - tl1
    int w2, h2, dx2, dy2;
};

// ---- kernels ------------------------------------------------------------------------------------------

// union masks, contour masks and the class of every union pixel (3: both masks, 1: first, 2: second, 0: none)
__global__ __launch_bounds__(256) void k_dp_union(DpGeom g, const uint8_t* __restrict__ m1, const uint8_t* __restrict__ m2,
                                                  uint8_t* __restrict__ c1, uint8_t* __restrict__ c2, uint8_t* __restrict__ cls)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= g.uw || y >= g.uh) return;
    auto at1 = [&](int xx, int yy) -> int {
        const int ix = xx + g.dx1, iy = yy + g.dy1;
        return (static_cast<unsigned>(ix) < static_cast<unsigned>(g.w1) && static_cast<unsigned>(iy) < static_cast<unsigned>(g.h1)) ? m1[static_cast<size_t>(iy) * g.w1 + ix] : 0;
    };
    auto at2 = [&](int xx, int yy) -> int {
        const int ix = xx + g.dx2, iy = yy + g.dy2;
        return (static_cast<unsigned>(ix) < static_cast<unsigned>(g.w2) && static_cast<unsigned>(iy) < static_cast<unsigned>(g.h2)) ? m2[static_cast<size_t>(iy) * g.w2 + ix] : 0;
    };
    const int a = at1(x, y), b = at2(x, y);
    const size_t i = static_cast<size_t>(y) * g.uw + x;
    c1[i] = (a && ((x == 0 || !at1(x - 1, y)) || (x == g.uw - 1 || !at1(x + 1, y)) || (y == 0 || !at1(x, y - 1)) || (y == g.uh - 1 || !at1(x, y + 1)))) ? 255 : 0;
    c2[i] = (b && ((x == 0 || !at2(x - 1, y)) || (x == g.uw - 1 || !at2(x + 1, y)) || (y == 0 || !at2(x, y - 1)) || (y == g.uh - 1 || !at2(x, y + 1)))) ? 255 : 0;
    cls[i] = static_cast<uint8_t>((a ? 1 : 0) | (b ? 2 : 0));
}

template <bool F32>
__global__ __launch_bounds__(256) void k_dp_gray(const void* __restrict__ img, size_t stride, int w, int h, float* __restrict__ gray)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    if (F32) {
        const float* p = reinterpret_cast<const float*>(static_cast<const char*>(img) + static_cast<size_t>(y) * stride) + 3 * x;
        gray[static_cast<size_t>(y) * w + x] = p[0] * 0.114f + p[1] * 0.587f + p[2] * 0.299f;        // RGB2Gray<float>, BGR order
    } else {
        // RGB2Gray<uchar>: 14-bit fixed point (B2Y 1868, G2Y 9617, R2Y 4899), the gray image is CV_8U; Sobel then runs on integers
        const uint8_t* p = static_cast<const uint8_t*>(img) + static_cast<size_t>(y) * stride + 3 * x;
        gray[static_cast<size_t>(y) * w + x] = static_cast<float>((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
    }
}

// Sobel(CV_32F, ksize 3, BORDER_REFLECT_101) = sepFilter2D, row pass then column pass.  The [1 2 1] sums are associated as the
// loop of OpenCV 2.4.11's SymmRowSmallFilter / SymmColumnSmallFilter<float> (x64, SSE2) that produces the element does -- SSE
// body for the first w & ~7 columns, then the scalar pairs and the generic tail (rows) / the scalar form (columns); on non-integer
// float images the forms differ by one ulp.  Read off the reference's shipped x64 DLL (oracle/REFERENCE_BINARY_AUDIT.md).
__global__ __launch_bounds__(256) void k_dp_sobel(const float* __restrict__ gray, int w, int h, float* __restrict__ gx, float* __restrict__ gy)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int xm = reflect<1>(x - 1, w), xp = reflect<1>(x + 1, w), ym = reflect<1>(y - 1, h), yp = reflect<1>(y + 1, h);
    const int w8 = w & ~7, rem = w - w8;
    const int row_form = x < w8 ? 0 : x < w8 + (rem & ~1) ? 1 : 2;
    const int col_form = x < w8 ? 0 : 1;       // (the shipped build has two column forms: SSE body, then (t + 2 m) + b to the end)
    auto row = [&](int yy, float& d, float& s) {
        const float a = gray[static_cast<size_t>(yy) * w + xm], c = gray[static_cast<size_t>(yy) * w + xp], m = gray[static_cast<size_t>(yy) * w + x];
        d = c - a;
        s = row_form == 0 ? a + ((m + m) + c) : row_form == 1 ? (a + m * 2.f) + c : m * 2.f + (c + a);
    };
    float d0, s0, d1, s1, d2, s2;
    row(ym, d0, s0); row(y, d1, s1); row(yp, d2, s2);
    gx[static_cast<size_t>(y) * w + x] = col_form == 0 ? d0 + (d2 + (d1 + d1)) : (d0 + d1 * 2.f) + d2;
    gy[static_cast<size_t>(y) * w + x] = s2 - s0;
}

struct DpImages {
    const void* img1; const void* img2; size_t stride1, stride2;
    const float* gx1; const float* gy1; const float* gx2; const float* gy2;
};

template <bool F32>
__device__ __forceinline__ float dp_diff(const void* a, size_t sa, int y1, int x1, const void* b, size_t sb, int y2, int x2)
{
    if (F32) {
        const float* r1 = reinterpret_cast<const float*>(static_cast<const char*>(a) + static_cast<size_t>(y1) * sa) + 3 * x1;
        const float* r2 = reinterpret_cast<const float*>(static_cast<const char*>(b) + static_cast<size_t>(y2) * sb) + 3 * x2;
        const float d0 = r1[0] - r2[0], d1 = r1[1] - r2[1], d2 = r1[2] - r2[2];
        return (d0 * d0 + d1 * d1) + d2 * d2;
    }
    const uint8_t* r1 = static_cast<const uint8_t*>(a) + static_cast<size_t>(y1) * sa + 3 * x1;
    const uint8_t* r2 = static_cast<const uint8_t*>(b) + static_cast<size_t>(y2) * sb + 3 * x2;
    const int d0 = r1[0] - r2[0], d1 = r1[1] - r2[1], d2 = r1[2] - r2[2];
    return static_cast<float>(d0 * d0 + d1 * d1 + d2 * d2);
}

// costV (rh x (rw + 1)) and costH ((rh + 1) x rw) of the component with label l whose box is (rx, ry, rw, rh); `lab` is the
// box's labels with one extra column and row ((rw + 1) x (rh + 1), what the reference reads past the box)
template <bool F32>
__global__ __launch_bounds__(256) void k_dp_costs(DpGeom g, DpImages im, const int* __restrict__ lab, int l, int rx, int ry, int rw, int rh,
                                                  int cost_func, float* __restrict__ cost_v, float* __restrict__ cost_h)
{
    const int ix = blockIdx.x * 64 + (threadIdx.x & 63), iy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (ix > rw || iy > rh) return;
    const float bad = (255.f * 255.f + 255.f * 255.f) + 255.f * 255.f;
    const int x = rx + ix, y = ry + iy, lp = rw + 1;
    auto L = [&](int yy, int xx) { return lab[(yy - ry) * lp + (xx - rx)]; };      // box coordinates incl. the extra row / column
    if (iy < rh) {            // costV(iy, ix), ix in [0, rw]
        float c = bad;
        if (L(y, x) == l && ix > 0 && L(y, x - 1) == l) {      // (x > 0 in union coordinates: ix > 0 implies it; ix == 0 reads outside the box)
            const float cc = (dp_diff<F32>(im.img1, im.stride1, y + g.dy1, x + g.dx1 - 1, im.img2, im.stride2, y + g.dy2, x + g.dx2) +
                              dp_diff<F32>(im.img1, im.stride1, y + g.dy1, x + g.dx1, im.img2, im.stride2, y + g.dy2, x + g.dx2 - 1)) / 2;
            if (cost_func == 0) c = cc;
            else {
                const float gr = fabsf(im.gx1[static_cast<size_t>(y + g.dy1) * g.w1 + x + g.dx1]) + fabsf(im.gx1[static_cast<size_t>(y + g.dy1) * g.w1 + x + g.dx1 - 1]) +
                                 fabsf(im.gx2[static_cast<size_t>(y + g.dy2) * g.w2 + x + g.dx2]) + fabsf(im.gx2[static_cast<size_t>(y + g.dy2) * g.w2 + x + g.dx2 - 1]) + 1.f;
                c = cc / gr;
            }
        }
        cost_v[static_cast<size_t>(iy) * (rw + 1) + ix] = c;
    }
    if (ix < rw) {            // costH(iy, ix), iy in [0, rh]
        float c = bad;
        if (L(y, x) == l && iy > 0 && L(y - 1, x) == l) {
            const float cc = (dp_diff<F32>(im.img1, im.stride1, y + g.dy1 - 1, x + g.dx1, im.img2, im.stride2, y + g.dy2, x + g.dx2) +
                              dp_diff<F32>(im.img1, im.stride1, y + g.dy1, x + g.dx1, im.img2, im.stride2, y + g.dy2 - 1, x + g.dx2)) / 2;
            if (cost_func == 0) c = cc;
            else {
                const float gr = fabsf(im.gy1[static_cast<size_t>(y + g.dy1) * g.w1 + x + g.dx1]) + fabsf(im.gy1[static_cast<size_t>(y + g.dy1 - 1) * g.w1 + x + g.dx1]) +
                                 fabsf(im.gy2[static_cast<size_t>(y + g.dy2) * g.w2 + x + g.dx2]) + fabsf(im.gy2[static_cast<size_t>(y + g.dy2 - 1) * g.w2 + x + g.dx2]) + 1.f;
                c = cc / gr;
            }
        }
        cost_h[static_cast<size_t>(iy) * rw + ix] = c;
    }
}

// estimateSeam's recurrence: step s of [s0 + 1, s1] along the seam direction, all cells across it in parallel (one
// block: a cell reads the previous step's three neighbours only).  control: 1 straight, 2 / 3 diagonal; 0 = unreachable.
__global__ __launch_bounds__(1024) void k_dp_sweep(const int* __restrict__ lab, int l, int rw, int rh, int horizontal, int s0, int s1, int src_cross,
                                                   const float* __restrict__ cost_v, const float* __restrict__ cost_h,
                                                   float* __restrict__ cost, uint8_t* __restrict__ control, uint8_t* __restrict__ reach)
{
    const int lp = rw + 1, across = horizontal ? rh : rw;
    // reachable(src) = 1, cost(src) = 0 (the buffers come zeroed)
    if (threadIdx.x == 0) reach[horizontal ? static_cast<size_t>(src_cross) * rw + s0 : static_cast<size_t>(s0) * rw + src_cross] = 1;
    __syncthreads();
    for (int s = s0 + 1; s <= s1; ++s) {
        for (int c = threadIdx.x; c < across; c += 1024) {
            const int x = horizontal ? s : c, y = horizontal ? c : s;
            if (lab[y * lp + x] != l) continue;
            float best = 0.f;
            int step = 0;
            // std::min_element over (cost, step) pairs in this order: the first minimum
            if (horizontal) {
                if (reach[static_cast<size_t>(y) * rw + x - 1]) { best = cost[static_cast<size_t>(y) * rw + x - 1] + cost_h[static_cast<size_t>(y) * rw + x - 1]; step = 1; }
                if (y > 0 && reach[static_cast<size_t>(y - 1) * rw + x - 1]) {
                    const float v = cost[static_cast<size_t>(y - 1) * rw + x - 1] + cost_h[static_cast<size_t>(y - 1) * rw + x - 1] + cost_v[static_cast<size_t>(y - 1) * lp + x];
                    if (!step || v < best) { best = v; step = 2; }
                }
                if (y < rh - 1 && reach[static_cast<size_t>(y + 1) * rw + x - 1]) {
                    const float v = cost[static_cast<size_t>(y + 1) * rw + x - 1] + cost_h[static_cast<size_t>(y + 1) * rw + x - 1] + cost_v[static_cast<size_t>(y) * lp + x];
                    if (!step || v < best) { best = v; step = 3; }
                }
            } else {
                if (reach[static_cast<size_t>(y - 1) * rw + x]) { best = cost[static_cast<size_t>(y - 1) * rw + x] + cost_v[static_cast<size_t>(y - 1) * lp + x]; step = 1; }
                if (x > 0 && reach[static_cast<size_t>(y - 1) * rw + x - 1]) {
                    const float v = cost[static_cast<size_t>(y - 1) * rw + x - 1] + cost_v[static_cast<size_t>(y - 1) * lp + x - 1] + cost_h[static_cast<size_t>(y) * rw + x - 1];
                    if (!step || v < best) { best = v; step = 2; }
                }
                if (x < rw - 1 && reach[static_cast<size_t>(y - 1) * rw + x + 1]) {
                    const float v = cost[static_cast<size_t>(y - 1) * rw + x + 1] + cost_v[static_cast<size_t>(y - 1) * lp + x + 1] + cost_h[static_cast<size_t>(y) * rw + x];
                    if (!step || v < best) { best = v; step = 3; }
                }
            }
            if (step) {
                cost[static_cast<size_t>(y) * rw + x] = best;
                control[static_cast<size_t>(y) * rw + x] = static_cast<uint8_t>(step);
                reach[static_cast<size_t>(y) * rw + x] = 255;
            }
        }
        __syncthreads();
    }
}

// the final mask update (:489-515): state bits per label (index l - 1) in `states`; which: 2 = update mask2, 1 = mask1
__global__ __launch_bounds__(256) void k_dp_apply(DpGeom g, const int* __restrict__ labels, const int* __restrict__ states, int which,
                                                  uint8_t* __restrict__ m1, uint8_t* __restrict__ m2)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int w = which == 2 ? g.w2 : g.w1, h = which == 2 ? g.h2 : g.h1;
    if (x >= w || y >= h) return;
    if (which == 2) {
        const int l = labels[static_cast<size_t>(y - g.dy2) * g.uw + (x - g.dx2)];
        const int yy = y - g.dy2 + g.dy1, xx = x - g.dx2 + g.dx1;
        if (l > 0 && (states[l - 1] & 1) && static_cast<unsigned>(xx) < static_cast<unsigned>(g.w1) && static_cast<unsigned>(yy) < static_cast<unsigned>(g.h1) &&
            m1[static_cast<size_t>(yy) * g.w1 + xx])
            m2[static_cast<size_t>(y) * g.w2 + x] = 0;
    } else {
        const int l = labels[static_cast<size_t>(y - g.dy1) * g.uw + (x - g.dx1)];
        const int yy = y - g.dy1 + g.dy2, xx = x - g.dx1 + g.dx2;
        if (l > 0 && (states[l - 1] & 2) && static_cast<unsigned>(xx) < static_cast<unsigned>(g.w2) && static_cast<unsigned>(yy) < static_cast<unsigned>(g.h2) &&
            m2[static_cast<size_t>(yy) * g.w2 + xx])
            m1[static_cast<size_t>(y) * g.w1 + x] = 0;
    }
}

inline dim3 grid64x4(int w, int h) { return dim3((w + 63) / 64, (h + 3) / 4); }

// ---- host bookkeeping ---------------------------------------------------------------------------------

struct P2 { int x, y; };
enum { kFirst = 1, kSecond = 2, kInters = 4 };

// cv::floodFill on CV_32S, 4-connectivity, zero tolerance
void flood_fill(std::vector<int>& img, int w, int h, int sx, int sy, int value)
{
    const int old = img[static_cast<size_t>(sy) * w + sx];
    if (old == value) return;
    std::vector<P2> todo{{sx, sy}};
    img[static_cast<size_t>(sy) * w + sx] = value;
    while (!todo.empty()) {
        const P2 p = todo.back();
        todo.pop_back();
        const P2 nb[4] = {{p.x - 1, p.y}, {p.x + 1, p.y}, {p.x, p.y - 1}, {p.x, p.y + 1}};
        for (const P2& q : nb) {
            if (q.x < 0 || q.y < 0 || q.x >= w || q.y >= h) continue;
            int& v = img[static_cast<size_t>(q.y) * w + q.x];
            if (v != old) continue;
            v = value;
            todo.push_back(q);
        }
    }
}

class PairSolver {
public:
    PairSolver(DevicePool& pool, hipStream_t s, int cost_func, bool f32) : pool_(pool), s_(s), cost_func_(cost_func), f32_(f32) {}

    void run(const DpGeom& g, const DpImages& im, uint8_t* d_m1, uint8_t* d_m2)
    {
        g_ = g; im_ = im;
        const size_t px = static_cast<size_t>(g.uw) * g.uh;
        uint8_t* d_c1 = pool_.alloc<uint8_t>(px);
        uint8_t* d_c2 = pool_.alloc<uint8_t>(px);
        uint8_t* d_cls = pool_.alloc<uint8_t>(px);
        hipLaunchKernelGGL(k_dp_union, grid64x4(g.uw, g.uh), dim3(256), 0, s_, g, d_m1, d_m2, d_c1, d_c2, d_cls);
        check_hip(hipGetLastError(), "k_dp_union");
        contour1_.resize(px); contour2_.resize(px);
        std::vector<uint8_t> cls(px);
        check_hip(hipMemcpyAsync(contour1_.data(), d_c1, px, hipMemcpyDeviceToHost, s_), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(contour2_.data(), d_c2, px, hipMemcpyDeviceToHost, s_), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(cls.data(), d_cls, px, hipMemcpyDeviceToHost, s_), "hipMemcpyAsync");
        check_hip(hipStreamSynchronize(s_), "hipStreamSynchronize");
        find_components(cls);
        find_edges();
        resolve_conflicts();
        // final mask update on the device
        int* d_labels = pool_.upload(labels_.data(), px, s_);
        std::vector<int32_t> st(states_.begin(), states_.end());
        if (st.empty()) st.push_back(0);
        int* d_states = pool_.upload(st.data(), st.size(), s_);
        hipLaunchKernelGGL(k_dp_apply, grid64x4(g.w2, g.h2), dim3(256), 0, s_, g, d_labels, d_states, 2, d_m1, d_m2);
        hipLaunchKernelGGL(k_dp_apply, grid64x4(g.w1, g.h1), dim3(256), 0, s_, g, d_labels, d_states, 1, d_m1, d_m2);
        check_hip(hipGetLastError(), "k_dp_apply");
        check_hip(hipStreamSynchronize(s_), "hipStreamSynchronize");       // `st` and labels_ are host-owned
    }

private:
    int& lab(int y, int x) { return labels_[static_cast<size_t>(y) * g_.uw + x]; }
    bool boundary(int y, int x, int l)
    {
        return (x == 0 || lab(y, x - 1) != l) || (x == g_.uw - 1 || lab(y, x + 1) != l) || (y == 0 || lab(y - 1, x) != l) ||
               (y == g_.uh - 1 || lab(y + 1, x) != l);
    }

    void find_components(const std::vector<uint8_t>& cls)              // :251-308
    {
        const int w = g_.uw, h = g_.uh;
        labels_.assign(static_cast<size_t>(w) * h, 0);
        for (size_t i = 0; i < labels_.size(); ++i)
            labels_[i] = cls[i] == 3 ? INT_MAX : (cls[i] == 1 ? INT_MAX - 1 : (cls[i] == 2 ? INT_MAX - 2 : 0));
        states_.clear(); tls_.clear(); brs_.clear(); contours_.clear();
        int ncomps = 0;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                const int v = lab(y, x);
                if (v >= INT_MAX - 2) {
                    states_.push_back(v == INT_MAX ? kInters : (v == INT_MAX - 1 ? kFirst : kSecond));
                    flood_fill(labels_, w, h, x, y, ++ncomps);
                    tls_.push_back(P2{x, y});
                    brs_.push_back(P2{x + 1, y + 1});
                    contours_.emplace_back();
                }
                if (const int l = lab(y, x)) {
                    const int ci = l - 1;
                    tls_[ci].x = std::min(tls_[ci].x, x); tls_[ci].y = std::min(tls_[ci].y, y);
                    brs_[ci].x = std::max(brs_[ci].x, x + 1); brs_[ci].y = std::max(brs_[ci].y, y + 1);
                    if (boundary(y, x, l)) contours_[ci].push_back(P2{x, y});
                }
            }
    }

    void find_edges()                                                   // :311-375
    {
        edges_.clear();
        const int w = g_.uw, h = g_.uh, n = static_cast<int>(states_.size());
        for (int ci = 0; ci < n; ++ci)
            for (const P2& p : contours_[ci]) {
                const int l = ci + 1;
                auto touch = [&](int yy, int xx) {
                    const int o = lab(yy, xx);
                    if (o && o != l) { edges_.insert({ci, o - 1}); edges_.insert({o - 1, ci}); }
                };
                if (p.x > 0) touch(p.y, p.x - 1);
                if (p.y > 0) touch(p.y - 1, p.x);
                if (p.x < w - 1) touch(p.y, p.x + 1);
                if (p.y < h - 1) touch(p.y + 1, p.x);
            }
    }

    bool only_one_neighbor(int c) const                                 // :546-552
    {
        auto b = edges_.lower_bound({c, INT_MIN});
        auto e = edges_.upper_bound({c, INT_MAX});
        return b != e && ++b == e;
    }

    bool close_to(const std::vector<uint8_t>& cm, int y, int x) const   // :555-575
    {
        for (int dy = -2; dy <= 2; ++dy) {
            if (y + dy < 0 || y + dy >= g_.uh) continue;
            for (int dx = -2; dx <= 2; ++dx)
                if (x + dx >= 0 && x + dx < g_.uw && cm[static_cast<size_t>(y + dy) * g_.uw + x + dx]) return true;
        }
        return false;
    }

    bool seam_tips(int c1, int c2, P2& p1, P2& p2)                      // :578-676
    {
        const int w = g_.uw, h = g_.uh, l2 = c2 + 1;
        std::vector<P2> sp;
        for (const P2& p : contours_[c1])
            if (close_to(contour1_, p.y, p.x) && close_to(contour2_, p.y, p.x) &&
                ((p.x > 0 && lab(p.y, p.x - 1) == l2) || (p.y > 0 && lab(p.y - 1, p.x) == l2) ||
                 (p.x < w - 1 && lab(p.y, p.x + 1) == l2) || (p.y < h - 1 && lab(p.y + 1, p.x) == l2)))
                sp.push_back(p);
        if (sp.size() < 2) return false;
        // cv::partition(ClosePoints(10)): classes of the "closer than 10" graph, numbered by their first member
        const int n = static_cast<int>(sp.size());
        std::vector<int> parent(n);
        for (int i = 0; i < n; ++i) parent[i] = i;
        auto root = [&](int i) { while (parent[i] != i) i = parent[i] = parent[parent[i]]; return i; };
        for (int i = 0; i < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                const int dx = sp[i].x - sp[j].x, dy = sp[i].y - sp[j].y;
                if (dx * dx + dy * dy < 100) parent[root(j)] = root(i);
            }
        std::map<int, int> cls_of_root;
        std::vector<int> label(n);
        for (int i = 0; i < n; ++i) {
            const int r = root(i);
            auto it = cls_of_root.find(r);
            if (it == cls_of_root.end()) it = cls_of_root.emplace(r, static_cast<int>(cls_of_root.size())).first;
            label[i] = it->second;
        }
        const int nl = static_cast<int>(cls_of_root.size());
        if (nl < 2) return false;
        std::vector<long> sx(nl, 0), sy(nl, 0);
        std::vector<int> cnt(nl, 0);
        for (int i = 0; i < n; ++i) { sx[label[i]] += sp[i].x; sy[label[i]] += sp[i].y; ++cnt[label[i]]; }
        auto rnd = [](double v) { return static_cast<double>(std::lrint(v)); };     // cvRound: half to even
        int idx[2] = {-1, -1};
        double max_dist = -std::numeric_limits<double>::max();
        for (int i = 0; i < nl - 1; ++i)
            for (int j = i + 1; j < nl; ++j) {
                const double n1 = cnt[i], n2 = cnt[j];
                const double cx1 = rnd(static_cast<int>(sx[i]) / n1), cy1 = rnd(static_cast<int>(sy[i]) / n1);
                const double cx2 = rnd(static_cast<int>(sx[j]) / n2), cy2 = rnd(static_cast<int>(sy[j]) / n1);     // (size1: as written, :634)
                const double dist = (cx1 - cx2) * (cx1 - cx2) + (cy1 - cy2) * (cy1 - cy2);
                if (dist > max_dist) { max_dist = dist; idx[0] = i; idx[1] = j; }
            }
        P2 out[2] = {{0, 0}, {0, 0}};
        for (int k = 0; k < 2; ++k) {
            const double nn = cnt[idx[k]];
            const double cx = rnd(static_cast<int>(sx[idx[k]]) / nn), cy = rnd(static_cast<int>(sy[idx[k]]) / nn);
            double min_dist = std::numeric_limits<double>::max();
            for (int i = 0; i < n; ++i) {
                if (label[i] != idx[k]) continue;
                const double dist = (sp[i].x - cx) * (sp[i].x - cx) + (sp[i].y - cy) * (sp[i].y - cy);
                if (dist < min_dist) { min_dist = dist; out[k] = sp[i]; }
            }
        }
        p1 = out[0]; p2 = out[1];
        return true;
    }

    bool estimate_seam(int comp, P2 p1, P2 p2, std::vector<P2>& seam, bool& horizontal)      // :781-909
    {
        const int rx = tls_[comp].x, ry = tls_[comp].y, rw = brs_[comp].x - rx, rh = brs_[comp].y - ry, l = comp + 1;
        // the box's labels with the extra column / row the cost pass reads (next row's first pixel past the right edge
        // of the union image, 0 past its bottom)
        const int lp = rw + 1;
        std::vector<int> box(static_cast<size_t>(lp) * (rh + 1));
        for (int y = 0; y <= rh; ++y)
            for (int x = 0; x <= rw; ++x) {
                const int ux = rx + x, uy = ry + y;
                int v = 0;
                if (uy < g_.uh) v = ux < g_.uw ? lab(uy, ux) : (uy + 1 < g_.uh ? lab(uy + 1, 0) : 0);
                box[static_cast<size_t>(y) * lp + x] = v;
            }
        int* d_box = pool_.upload(box.data(), box.size(), s_);
        float* d_cv = pool_.alloc<float>(static_cast<size_t>(rh) * lp);
        float* d_ch = pool_.alloc<float>(static_cast<size_t>(rh + 1) * rw);
        if (f32_) hipLaunchKernelGGL(k_dp_costs<true>, grid64x4(rw + 1, rh + 1), dim3(256), 0, s_, g_, im_, d_box, l, rx, ry, rw, rh, cost_func_, d_cv, d_ch);
        else hipLaunchKernelGGL(k_dp_costs<false>, grid64x4(rw + 1, rh + 1), dim3(256), 0, s_, g_, im_, d_box, l, rx, ry, rw, rh, cost_func_, d_cv, d_ch);
        P2 src{p1.x - rx, p1.y - ry}, dst{p2.x - rx, p2.y - ry};
        bool swapped = false;
        horizontal = std::abs(dst.x - src.x) > std::abs(dst.y - src.y);
        if (horizontal) { if (src.x > dst.x) { std::swap(src, dst); swapped = true; } }
        else if (src.y > dst.y) { std::swap(src, dst); swapped = true; }
        const size_t cells = static_cast<size_t>(rw) * rh;
        float* d_cost = pool_.alloc<float>(cells);
        uint8_t* d_ctl = pool_.alloc<uint8_t>(cells);
        uint8_t* d_reach = pool_.alloc<uint8_t>(cells);
        check_hip(hipMemsetAsync(d_cost, 0, cells * sizeof(float), s_), "hipMemsetAsync");
        check_hip(hipMemsetAsync(d_ctl, 0, cells, s_), "hipMemsetAsync");
        check_hip(hipMemsetAsync(d_reach, 0, cells, s_), "hipMemsetAsync");
        hipLaunchKernelGGL(k_dp_sweep, dim3(1), dim3(1024), 0, s_, d_box, l, rw, rh, horizontal ? 1 : 0, horizontal ? src.x : src.y,
                           horizontal ? dst.x : dst.y, horizontal ? src.y : src.x, d_cv, d_ch, d_cost, d_ctl, d_reach);
        check_hip(hipGetLastError(), "dp seam kernels");
        std::vector<uint8_t> ctl(cells), reach(cells);
        check_hip(hipMemcpyAsync(ctl.data(), d_ctl, cells, hipMemcpyDeviceToHost, s_), "hipMemcpyAsync");
        check_hip(hipMemcpyAsync(reach.data(), d_reach, cells, hipMemcpyDeviceToHost, s_), "hipMemcpyAsync");
        check_hip(hipStreamSynchronize(s_), "hipStreamSynchronize");
        if (!reach[static_cast<size_t>(dst.y) * rw + dst.x]) return false;
        P2 p = dst;
        seam.clear();
        seam.push_back(P2{p.x + rx, p.y + ry});
        if (horizontal) {
            while (p.x != src.x) {
                const int c = ctl[static_cast<size_t>(p.y) * rw + p.x];
                if (c == 2) --p.y; else if (c == 3) ++p.y;
                --p.x;
                seam.push_back(P2{p.x + rx, p.y + ry});
            }
        } else {
            while (p.y != src.y) {
                const int c = ctl[static_cast<size_t>(p.y) * rw + p.x];
                if (c == 2) --p.x; else if (c == 3) ++p.x;
                --p.y;
                seam.push_back(P2{p.x + rx, p.y + ry});
            }
        }
        if (!swapped) std::reverse(seam.begin(), seam.end());
        return true;
    }

    void update_labels(int c1, int c2, const std::vector<P2>& seam, bool horizontal)          // :912-1045
    {
        const int tx = tls_[c1].x, ty = tls_[c1].y, mw = brs_[c1].x - tx, mh = brs_[c1].y - ty;
        const int w = g_.uw, h = g_.uh, l1 = c1 + 1, l2 = c2 + 1;
        std::vector<int> mask(static_cast<size_t>(mw) * mh, 0);
        auto M = [&](int y, int x) -> int& { return mask[static_cast<size_t>(y) * mw + x]; };
        for (const P2& p : contours_[c1]) M(p.y - ty, p.x - tx) = 255;
        for (const P2& p : seam) M(p.y - ty, p.x - tx) = 255;
        int ncomps = 0;
        for (int y = 0; y < mh; ++y)
            for (int x = 0; x < mw; ++x)
                if (!M(y, x) && lab(y + ty, x + tx) == l1) flood_fill(mask, mw, mh, x, y, ++ncomps);
        static const int dx8[8] = {-1, +1, 0, 0, -1, +1, -1, +1}, dy8[8] = {0, 0, -1, +1, -1, -1, +1, +1};
        for (const P2& p : contours_[c1]) {
            const int x = p.x - tx, y = p.y - ty;
            bool ok = false;
            for (int j = 0; j < 8; ++j) {
                const int c = x + dx8[j], r = y + dy8[j];
                if (c >= 0 && c < mw && r >= 0 && r < mh && M(r, c) && M(r, c) != 255) { ok = true; M(y, x) = M(r, c); }
            }
            if (!ok) M(y, x) = 0;
        }
        for (const P2& p : seam) {
            const int x = p.x - tx, y = p.y - ty;
            if (horizontal) M(y, x) = (y < mh - 1 && M(y + 1, x) && M(y + 1, x) != 255) ? M(y + 1, x) : 0;
            else M(y, x) = (x < mw - 1 && M(y, x + 1) && M(y, x + 1) != 255) ? M(y, x + 1) : 0;
        }
        std::map<int, int> connect2, connect_other;
        for (int i = 1; i <= ncomps; ++i) { connect2[i] = 0; connect_other[i] = 0; }
        for (const P2& p : contours_[c1]) {
            const int x = p.x, y = p.y, key = M(y - ty, x - tx);
            if ((x > 0 && lab(y, x - 1) == l2) || (y > 0 && lab(y - 1, x) == l2) || (x < w - 1 && lab(y, x + 1) == l2) || (y < h - 1 && lab(y + 1, x) == l2))
                connect2[key]++;
            auto other = [&](int yy, int xx) { const int o = lab(yy, xx); return o != l1 && o != l2; };
            if ((x > 0 && other(y, x - 1)) || (y > 0 && other(y - 1, x)) || (x < w - 1 && other(y, x + 1)) || (y < h - 1 && other(y + 1, x)))
                connect_other[key]++;
        }
        std::vector<char> adj(static_cast<size_t>(ncomps) + 1, 0);
        const double len = static_cast<double>(contours_[c1].size());
        for (const auto& kv : connect2) {
            if (kv.first < 0 || kv.first > ncomps) continue;          // (keys past ncomps are never read back)
            auto it = connect_other.find(kv.first);
            const int oth = it == connect_other.end() ? 0 : it->second;
            adj[kv.first] = kv.second / len > 0.05 && oth / len < 0.1;
        }
        for (int y = 0; y < mh; ++y)
            for (int x = 0; x < mw; ++x) {
                const int v = M(y, x);
                if (v && v <= ncomps && adj[v]) lab(y + ty, x + tx) = l2;
            }
    }

    void resolve_conflicts()                                            // :378-486
    {
        const int w = g_.uw, h = g_.uh;
        for (;;) {
            int c1 = 0, c2 = 0;
            bool conflict = false;
            for (const auto& e : edges_) {
                c1 = e.first; c2 = e.second;
                if ((states_[c1] & kInters) && (states_[c1] & ~kInters) != states_[c2]) { conflict = true; break; }
            }
            if (!conflict) break;
            const int l1 = c1 + 1, l2 = c2 + 1;
            if (only_one_neighbor(c1)) {
                for (int y = tls_[c1].y; y < brs_[c1].y; ++y)
                    for (int x = tls_[c1].x; x < brs_[c1].x; ++x)
                        if (lab(y, x) == l1) lab(y, x) = l2;
                states_[c1] = states_[c2] == kFirst ? kSecond : kFirst;
            } else {
                P2 p1, p2;
                if (seam_tips(c1, c2, p1, p2)) {
                    std::vector<P2> seam;
                    bool horizontal;
                    if (estimate_seam(c1, p1, p2, seam, horizontal)) update_labels(c1, c2, seam, horizontal);
                }
                states_[c1] = states_[c2] == kFirst ? (kInters | kSecond) : (kInters | kFirst);
            }
            const int cs[2] = {c1, c2}, ls[2] = {l1, l2};
            for (int i = 0; i < 2; ++i) {
                const int c = cs[i], l = ls[i];
                const int x0 = tls_[c].x, x1 = brs_[c].x, y0 = tls_[c].y, y1 = brs_[c].y;
                tls_[c] = P2{INT_MAX, INT_MAX};
                brs_[c] = P2{INT_MIN, INT_MIN};
                contours_[c].clear();
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) {
                        if (lab(y, x) != l) continue;
                        tls_[c].x = std::min(tls_[c].x, x); tls_[c].y = std::min(tls_[c].y, y);
                        brs_[c].x = std::max(brs_[c].x, x + 1); brs_[c].y = std::max(brs_[c].y, y + 1);
                        if ((x == 0 || lab(y, x - 1) != l) || (x == w - 1 || lab(y, x + 1) != l) || (y == 0 || lab(y - 1, x) != l) ||
                            (y == h - 1 || lab(y + 1, x) != l))
                            contours_[c].push_back(P2{x, y});
                    }
            }
            edges_.erase({c1, c2});
            edges_.erase({c2, c1});
        }
    }

    DevicePool& pool_;
    hipStream_t s_;
    int cost_func_;
    bool f32_;
    DpGeom g_{};
    DpImages im_{};
    std::vector<int> labels_;
    std::vector<uint8_t> contour1_, contour2_;
    std::vector<int> states_;
    std::vector<P2> tls_, brs_;
    std::vector<std::vector<P2>> contours_;
    std::set<std::pair<int, int>> edges_;
};

}  // namespace

// DpSeamFinder::find (:163-189): every pair (i < j), ordered by the distance of the image centres, farthest first;
// std::sort below 32 elements is MSVC's insertion sort (stable), the driver has at most 28 pairs.
void dp_seams(int device, int n, const int32_t* corners_xy, const int32_t* sizes_wh, const void* const* images, const size_t* image_strides,
              int image_is_f32, uint8_t* const* masks, const size_t* mask_strides, int cost_func)
{
    select_device(device);
    hipStream_t s = nullptr;
    check_hip(hipStreamCreate(&s), "hipStreamCreate");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } guard{s};
    DevicePool pool;
    const bool f32 = image_is_f32 != 0;
    const size_t px_bytes = f32 ? 12 : 3;
    std::vector<uint8_t*> d_mask(n);
    std::vector<void*> d_img(n);
    std::vector<float*> d_gx(n, nullptr), d_gy(n, nullptr);
    for (int i = 0; i < n; ++i) {
        const int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        if (w <= 0 || h <= 0 || !masks[i] || !images[i] || mask_strides[i] < static_cast<size_t>(w) || image_strides[i] < w * px_bytes)
            throw Error{MST_ERR_INVALID_ARG, "bad image or mask"};
        d_mask[i] = pool.alloc<uint8_t>(static_cast<size_t>(w) * h);
        check_hip(hipMemcpy2DAsync(d_mask[i], w, masks[i], mask_strides[i], w, h, hipMemcpyHostToDevice, s), "hipMemcpy2DAsync(mask)");
        d_img[i] = pool.alloc_bytes(static_cast<size_t>(w) * h * px_bytes);
        check_hip(hipMemcpy2DAsync(d_img[i], w * px_bytes, images[i], image_strides[i], w * px_bytes, h, hipMemcpyHostToDevice, s), "hipMemcpy2DAsync(image)");
        if (cost_func == 1) {
            float* gray = pool.alloc<float>(static_cast<size_t>(w) * h);
            d_gx[i] = pool.alloc<float>(static_cast<size_t>(w) * h);
            d_gy[i] = pool.alloc<float>(static_cast<size_t>(w) * h);
            if (f32) hipLaunchKernelGGL(k_dp_gray<true>, grid64x4(w, h), dim3(256), 0, s, d_img[i], w * px_bytes, w, h, gray);
            else hipLaunchKernelGGL(k_dp_gray<false>, grid64x4(w, h), dim3(256), 0, s, d_img[i], w * px_bytes, w, h, gray);
            hipLaunchKernelGGL(k_dp_sobel, grid64x4(w, h), dim3(256), 0, s, gray, w, h, d_gx[i], d_gy[i]);
        }
    }
    check_hip(hipGetLastError(), "dp seam set-up");
    struct Pair { int i, j; int d; };
    std::vector<Pair> pairs;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const int c1x = corners_xy[2 * i] + sizes_wh[2 * i] / 2, c1y = corners_xy[2 * i + 1] + sizes_wh[2 * i + 1] / 2;
            const int c2x = corners_xy[2 * j] + sizes_wh[2 * j] / 2, c2y = corners_xy[2 * j + 1] + sizes_wh[2 * j + 1] / 2;
            pairs.push_back(Pair{i, j, (c1x - c2x) * (c1x - c2x) + (c1y - c2y) * (c1y - c2y)});
        }
    std::stable_sort(pairs.begin(), pairs.end(), [](const Pair& a, const Pair& b) { return a.d < b.d; });
    std::reverse(pairs.begin(), pairs.end());
    for (const Pair& p : pairs) {
        const int tl1x = corners_xy[2 * p.i], tl1y = corners_xy[2 * p.i + 1], tl2x = corners_xy[2 * p.j], tl2y = corners_xy[2 * p.j + 1];
        const int w1 = sizes_wh[2 * p.i], h1 = sizes_wh[2 * p.i + 1], w2 = sizes_wh[2 * p.j], h2 = sizes_wh[2 * p.j + 1];
        if (std::max(tl1x, tl2x) >= std::min(tl1x + w1, tl2x + w2) || std::max(tl1y, tl2y) >= std::min(tl1y + h1, tl2y + h2)) continue;
        DpGeom g;
        g.utlx = std::min(tl1x, tl2x); g.utly = std::min(tl1y, tl2y);
        g.uw = std::max(tl1x + w1, tl2x + w2) - g.utlx; g.uh = std::max(tl1y + h1, tl2y + h2) - g.utly;
        g.w1 = w1; g.h1 = h1; g.dx1 = g.utlx - tl1x; g.dy1 = g.utly - tl1y;
        g.w2 = w2; g.h2 = h2; g.dx2 = g.utlx - tl2x; g.dy2 = g.utly - tl2y;
        DpImages im{d_img[p.i], d_img[p.j], w1 * px_bytes, w2 * px_bytes, d_gx[p.i], d_gy[p.i], d_gx[p.j], d_gy[p.j]};
        PairSolver solver(pool, s, cost_func, f32);
        solver.run(g, im, d_mask[p.i], d_mask[p.j]);
    }
    for (int i = 0; i < n; ++i) {
        const int w = sizes_wh[2 * i], h = sizes_wh[2 * i + 1];
        check_hip(hipMemcpy2DAsync(masks[i], mask_strides[i], d_mask[i], w, w, h, hipMemcpyDeviceToHost, s), "hipMemcpy2DAsync(mask)");
    }
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(seams)");
}

}  // namespace mst
