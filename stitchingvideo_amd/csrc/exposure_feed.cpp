// exposure_feed.cpp -- exposure-gain ESTIMATION, the calibration-side half of the exposure compensators
// (SURVEY.md 8f row 2): GainCompensator::feed / BlocksGainCompensator::feed, R64/src/exposure_compensate.cpp:73-147,
// 165-222 (= the driver's BlockFeed, DRV:251-306).
//
// The part with any bandwidth in it -- for every pair of overlapping (block) images, the count of doubly covered
// pixels and the two sums of sqrt(B^2 + G^2 + R^2) over them -- runs on the device: one thread per pair, walking the
// pair's overlap in the reference's row-major order, so that every double sum is formed in exactly the reference's
// order (a parallel reduction would not be).  The normal equations, cv::solve and the two [1/4 1/2 1/4] smoothing
// passes of the block gain maps are a few thousand operations and stay on the host.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "engine.hpp"
#include "kernels.hpp"

namespace mst {

namespace {

struct SubImage {                 // a block (or a whole image) as the GainCompensator sees it
    int image;                    // parent image
    int ox, oy;                   // origin inside the parent
    int x, y, w, h;               // corner (canvas coordinates) and size
};

bool overlap(const SubImage& a, const SubImage& b, int& x, int& y, int& w, int& h)     // util.cpp:103-115
{
    const int x_tl = std::max(a.x, b.x), y_tl = std::max(a.y, b.y);
    const int x_br = std::min(a.x + a.w, b.x + b.w), y_br = std::min(a.y + a.h, b.y + b.h);
    if (x_tl < x_br && y_tl < y_br) { x = x_tl; y = y_tl; w = x_br - x_tl; h = y_br - y_tl; return true; }
    return false;
}

// cv::solve(A, b, x) as GainCompensator::feed calls it: DECOMP_LU on CV_64F, one right-hand side.  OpenCV 2.4.11
// takes closed forms for n <= 3 and LUImpl beyond (partial pivoting, strict > on magnitudes, rows scaled by
// -1 / pivot, reciprocals kept on the diagonal for the back substitution).
void solve_lu(std::vector<double> A, std::vector<double> b, int n, double* x)
{
    auto zero = [&] { for (int i = 0; i < n; ++i) x[i] = 0.; };
    if (n == 1) {
        if (A[0] == 0.) { zero(); return; }
        x[0] = b[0] * (1. / A[0]);
        return;
    }
    if (n == 2) {
        double d = A[0] * A[3] - A[1] * A[2];
        if (d == 0.) { zero(); return; }
        d = 1. / d;
        const double t = (b[0] * A[3] - b[1] * A[1]) * d;
        x[1] = (b[1] * A[0] - b[0] * A[2]) * d;
        x[0] = t;
        return;
    }
    if (n == 3) {
        auto m = [&](int r, int c) { return A[r * 3 + c]; };
        double d = m(0, 0) * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
                   m(0, 2) * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0));
        if (d == 0.) { zero(); return; }
        d = 1. / d;
        const double t0 = d * (b[0] * (m(1, 1) * m(2, 2) - m(1, 2) * m(2, 1)) - m(0, 1) * (b[1] * m(2, 2) - m(1, 2) * b[2]) +
                               m(0, 2) * (b[1] * m(2, 1) - m(1, 1) * b[2]));
        const double t1 = d * (m(0, 0) * (b[1] * m(2, 2) - m(1, 2) * b[2]) - b[0] * (m(1, 0) * m(2, 2) - m(1, 2) * m(2, 0)) +
                               m(0, 2) * (m(1, 0) * b[2] - b[1] * m(2, 0)));
        const double t2 = d * (m(0, 0) * (m(1, 1) * b[2] - b[1] * m(2, 1)) - m(0, 1) * (m(1, 0) * b[2] - b[1] * m(2, 0)) +
                               b[0] * (m(1, 0) * m(2, 1) - m(1, 1) * m(2, 0)));
        x[0] = t0; x[1] = t1; x[2] = t2;
        return;
    }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r)
            if (std::fabs(A[r * n + col]) > std::fabs(A[piv * n + col])) piv = r;
        if (std::fabs(A[piv * n + col]) < DBL_EPSILON) { zero(); return; }
        if (piv != col) {
            for (int c = col; c < n; ++c) std::swap(A[col * n + c], A[piv * n + c]);
            std::swap(b[col], b[piv]);
        }
        const double d = -1 / A[col * n + col];
        for (int r = col + 1; r < n; ++r) {
            const double f = A[r * n + col] * d;
            for (int c = col + 1; c < n; ++c) A[r * n + c] += f * A[col * n + c];
            b[r] += f * b[col];
        }
        A[col * n + col] = -d;
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int c = r + 1; c < n; ++c) s -= A[r * n + c] * b[c];
        b[r] = s * A[r * n + r];
    }
    for (int i = 0; i < n; ++i) x[i] = b[i];
}

int mirror101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// sepFilter2D(map, map, CV_32F, ker, ker) with ker = [0.25 0.5 0.25]: a symmetric 3-tap float kernel runs as
// centre * k0 + (left + right) * k1 along rows, then along columns, BORDER_REFLECT_101, float intermediate
void smooth3(std::vector<float>& m, int w, int h)
{
    std::vector<float> t(m.size());
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            const float* r = &m[static_cast<size_t>(y) * w];
            t[static_cast<size_t>(y) * w + x] = r[x] * 0.5f + (r[mirror101(x - 1, w)] + r[mirror101(x + 1, w)]) * 0.25f;
        }
    for (int y = 0; y < h; ++y) {
        const float* up = &t[static_cast<size_t>(mirror101(y - 1, h)) * w];
        const float* mid = &t[static_cast<size_t>(y) * w];
        const float* dn = &t[static_cast<size_t>(mirror101(y + 1, h)) * w];
        for (int x = 0; x < w; ++x) m[static_cast<size_t>(y) * w + x] = (up[x] + dn[x]) * 0.25f + mid[x] * 0.5f;
    }
}

}  // namespace

void gain_map_size(int w, int h, int bl_w, int bl_h, int* gw, int* gh)
{
    *gw = (w + bl_w - 1) / bl_w;
    *gh = (h + bl_h - 1) / bl_h;
}

void exposure_feed(int device, int expos_kind, int n, const int32_t* corners_xy, const uint8_t* const* images,
                   const size_t* image_strides, const int32_t* sizes_wh, const uint8_t* const* masks,
                   const size_t* mask_strides, int bl_w, int bl_h, double* gains, float* const* gain_maps)
{
    select_device(device);
    const bool blocks = expos_kind == MST_EXPOS_GAIN_BLOCKS;
    // ---- what the GainCompensator is fed: whole images, or (BlocksGainCompensator) every block as an image
    std::vector<SubImage> subs;
    std::vector<int> gw(n, 1), gh(n, 1);
    for (int i = 0; i < n; ++i) {
        const int cols = sizes_wh[2 * i], rows = sizes_wh[2 * i + 1];
        if (cols <= 0 || rows <= 0 || !images[i] || !masks[i] || image_strides[i] < static_cast<size_t>(cols) * 3 ||
            mask_strides[i] < static_cast<size_t>(cols))
            throw Error{MST_ERR_INVALID_ARG, "bad image or mask for exposure feed"};
        if (!blocks) {
            subs.push_back(SubImage{i, 0, 0, corners_xy[2 * i], corners_xy[2 * i + 1], cols, rows});
            continue;
        }
        gain_map_size(cols, rows, bl_w, bl_h, &gw[i], &gh[i]);
        const int bw = (cols + gw[i] - 1) / gw[i], bh = (rows + gh[i] - 1) / gh[i];
        for (int by = 0; by < gh[i]; ++by)
            for (int bx = 0; bx < gw[i]; ++bx) {
                const int x0 = bx * bw, y0 = by * bh;
                subs.push_back(SubImage{i, x0, y0, corners_xy[2 * i] + x0, corners_xy[2 * i + 1] + y0,
                                        std::min(x0 + bw, cols) - x0, std::min(y0 + bh, rows) - y0});
            }
    }
    const int m = static_cast<int>(subs.size());
    if (m > 8192) throw Error{MST_ERR_INVALID_ARG, "too many blocks for the gain solve (feed seam-scale images)"};
    // ---- overlapping pairs i <= j, with the overlap's origin inside each parent image
    std::vector<FeedPair> pairs;
    for (int i = 0; i < m; ++i)
        for (int j = i; j < m; ++j) {
            int x, y, w, h;
            if (!overlap(subs[i], subs[j], x, y, w, h)) continue;
            FeedPair p;
            p.a = subs[i].image; p.b = subs[j].image;
            p.ax = x - subs[i].x + subs[i].ox; p.ay = y - subs[i].y + subs[i].oy;
            p.bx = x - subs[j].x + subs[j].ox; p.by = y - subs[j].y + subs[j].oy;
            p.w = w; p.h = h;
            p.i = i; p.j = j;
            pairs.push_back(p);
        }
    // ---- device: upload, one thread per pair
    DevicePool pool;
    hipStream_t s = nullptr;
    check_hip(hipStreamCreate(&s), "hipStreamCreate");
    struct StreamGuard { hipStream_t s; ~StreamGuard() { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } } guard{s};
    std::vector<FeedImage> dimg(n);
    for (int i = 0; i < n; ++i) {
        const int cols = sizes_wh[2 * i], rows = sizes_wh[2 * i + 1];
        uint8_t* di = pool.alloc<uint8_t>(static_cast<size_t>(cols) * 3 * rows);
        uint8_t* dm = pool.alloc<uint8_t>(static_cast<size_t>(cols) * rows);
        check_hip(hipMemcpy2DAsync(di, static_cast<size_t>(cols) * 3, images[i], image_strides[i], static_cast<size_t>(cols) * 3, rows,
                                   hipMemcpyHostToDevice, s), "hipMemcpy2DAsync(feed image)");
        check_hip(hipMemcpy2DAsync(dm, cols, masks[i], mask_strides[i], cols, rows, hipMemcpyHostToDevice, s),
                  "hipMemcpy2DAsync(feed mask)");
        dimg[i] = FeedImage{di, dm, static_cast<uint32_t>(cols) * 3u, static_cast<uint32_t>(cols)};
    }
    const int np = static_cast<int>(pairs.size());
    const FeedImage* d_img = pool.upload(dimg, s);
    const FeedPair* d_pairs = pool.upload(pairs, s);
    FeedStat* d_stats = pool.alloc<FeedStat>(static_cast<size_t>(std::max(np, 1)));
    launch_overlap_stats(d_img, d_pairs, np, d_stats, s);
    check_hip(hipGetLastError(), "k_overlap_stats");
    std::vector<FeedStat> stats(np);
    if (np) check_hip(hipMemcpyAsync(stats.data(), d_stats, sizeof(FeedStat) * np, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(stats)");
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(feed)");
    // ---- N, I, the normal equations (exposure_compensate.cpp:123-141), cv::solve
    std::vector<int> N(static_cast<size_t>(m) * m, 0);
    std::vector<double> I(static_cast<size_t>(m) * m, 0.);
    for (int k = 0; k < np; ++k) {
        const int i = pairs[k].i, j = pairs[k].j;
        const int cnt = std::max(1, stats[k].count);
        N[static_cast<size_t>(i) * m + j] = N[static_cast<size_t>(j) * m + i] = cnt;
        I[static_cast<size_t>(i) * m + j] = stats[k].sum_a / cnt;
        I[static_cast<size_t>(j) * m + i] = stats[k].sum_b / cnt;
    }
    const double alpha = 0.01, beta = 100;
    std::vector<double> A(static_cast<size_t>(m) * m, 0.), b(m, 0.);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            const size_t ij = static_cast<size_t>(i) * m + j, ji = static_cast<size_t>(j) * m + i, ii = static_cast<size_t>(i) * m + i;
            b[i] += beta * N[ij];
            A[ii] += beta * N[ij];
            if (j == i) continue;
            A[ii] += 2 * alpha * I[ij] * I[ij] * N[ij];
            A[ij] -= 2 * alpha * I[ij] * I[ji] * N[ij];
        }
    std::vector<double> g(m, 0.);
    solve_lu(std::move(A), std::move(b), m, g.data());
    if (!blocks) {
        for (int i = 0; i < n; ++i) gains[i] = g[i];
        return;
    }
    // ---- gain maps: one value per block, smoothed twice (exposure_compensate.cpp:205-221)
    int k = 0;
    for (int i = 0; i < n; ++i) {
        std::vector<float> map(static_cast<size_t>(gw[i]) * gh[i]);
        for (float& v : map) v = static_cast<float>(g[k++]);
        smooth3(map, gw[i], gh[i]);
        smooth3(map, gw[i], gh[i]);
        std::memcpy(gain_maps[i], map.data(), map.size() * sizeof(float));
    }
}

}  // namespace mst
