// mb_plan.cpp -- see mb_plan.hpp.
#include "mb_plan.hpp"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "kernels.hpp"

namespace mst {

namespace {

int reflect101(int p, int len)
{
    if (static_cast<unsigned>(p) < static_cast<unsigned>(len)) return p;
    if (len == 1) return 0;
    do {
        if (p < 0) p = -p;
        else p = 2 * len - 2 - p;
    } while (static_cast<unsigned>(p) >= static_cast<unsigned>(len));
    return p;
}

// A small persistent pool for the loops below (the plan is a few dozen independent passes over byte maps -- per
// camera, per stripe of rows -- and used to create and join a set of std::threads for every one of them, a third of
// its run time).  A loop is published as a job with an atomic index; the caller works on its own job like any
// worker, so nested loops (cameras outside, stripes inside) always make progress, and waits only for the items
// other threads have already taken.
class LoopPool {
public:
    struct Job {
        std::function<void(int)> body;
        int count = 0;
        std::atomic<int> next{0}, done{0};
    };
    static LoopPool& instance() { static LoopPool p; return p; }
    void run(int count, int max_threads, std::function<void(int)> body)
    {
        if (count <= 0) return;
        if (count == 1 || max_threads <= 1 || workers_.empty()) { for (int i = 0; i < count; ++i) body(i); return; }
        auto job = std::make_shared<Job>();
        job->body = std::move(body);
        job->count = count;
        {
            std::lock_guard<std::mutex> lk(mu_);
            jobs_.push_back(job);
        }
        cv_.notify_all();
        work_on(*job);
        while (job->done.load(std::memory_order_acquire) < count) std::this_thread::yield();
        std::lock_guard<std::mutex> lk(mu_);
        jobs_.erase(std::find(jobs_.begin(), jobs_.end(), job));
    }
private:
    LoopPool()
    {
        const int hw = static_cast<int>(std::max(1u, std::thread::hardware_concurrency()));
        const int n = std::min(hw, 32) - 1;                 // the caller is a worker too
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { worker(); });
    }
    ~LoopPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    static void work_on(Job& j)
    {
        for (;;) {
            const int i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.count) return;
            j.body(i);
            j.done.fetch_add(1, std::memory_order_release);
        }
    }
    void worker()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            std::shared_ptr<Job> pick;
            for (const std::shared_ptr<Job>& j : jobs_)
                if (j->next.load(std::memory_order_relaxed) < j->count) { pick = j; break; }
            if (pick) {
                lk.unlock();
                work_on(*pick);
                lk.lock();
                continue;
            }
            if (stop_) return;
            cv_.wait(lk);
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::shared_ptr<Job>> jobs_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};

// run f(i) for i in [0, count) on the pool (the caller included)
template <typename F> void parallel_for(int count, int max_threads, F&& f)
{
    LoopPool::instance().run(count, max_threads, std::function<void(int)>(std::forward<F>(f)));
}

// index of the first non-zero byte of row[x .. n), or n: whole zero words are skipped eight bytes at a time (the byte
// maps of a wrap-around camera are mostly zero)
inline int next_set(const uint8_t* row, int x, int n)
{
    while (x < n && (reinterpret_cast<uintptr_t>(row + x) & 7u)) {
        if (row[x]) return x;
        ++x;
    }
    while (x + 8 <= n) {
        uint64_t w;
        std::memcpy(&w, row + x, 8);
        if (w) return x + (__builtin_ctzll(w) >> 3);
        x += 8;
    }
    while (x < n) {
        if (row[x]) return x;
        ++x;
    }
    return n;
}

// rows of a level cut into stripes for the passes below: each worker owns the OUTPUT rows of its stripe, reads the
// input rows that can reach them, and writes nowhere else
constexpr int kStripe = 64;

// source indices cv::pyrUp reads around coarse index j (same clamping as the kernels' up_idx)
void up_idx(int j, int n, int& jm, int& jp)
{
    jm = j == 0 ? std::min(1, n - 1) : j - 1;
    jp = std::min(j + 1, n - 1);
}

// mark, at the coarser level, the 3x3 pyrUp sources of every set pixel of the finer level
void mark_parents(const std::vector<uint8_t>& fine, int fw, int fh, std::vector<uint8_t>& coarse, int cw, int ch)
{
    const int stripes = (ch + kStripe - 1) / kStripe;
    parallel_for(stripes, 32, [&](int sidx) {
        const int c0 = sidx * kStripe, c1 = std::min(ch, c0 + kStripe);          // coarse rows this worker may write
        // fine rows whose sources {i-1, i, i+1} (clamped) can fall into [c0, c1)
        const int y0 = std::max(0, 2 * (c0 - 1)), y1 = std::min(fh, 2 * (c1 + 1));
        for (int y = y0; y < y1; ++y) {
            const uint8_t* row = fine.data() + static_cast<size_t>(y) * fw;
            int im, ip;
            const int i = y >> 1;
            up_idx(i, ch, im, ip);
            const int ys[3] = {im, i, ip};
            for (int x = next_set(row, 0, fw); x < fw; x = next_set(row, x + 1, fw)) {
                int jm, jp;
                const int j = x >> 1;
                up_idx(j, cw, jm, jp);
                for (int yy : ys) {
                    if (yy < c0 || yy >= c1) continue;
                    uint8_t* d = coarse.data() + static_cast<size_t>(yy) * cw;
                    d[jm] = 1; d[j] = 1; d[jp] = 1;
                }
            }
        }
    });
}

// mark, at the finer level, the 5x5 pyrDown footprint (REFLECT_101) of every set pixel of the coarser level
void mark_footprint(const std::vector<uint8_t>& coarse, int cw, int ch, std::vector<uint8_t>& fine, int fw, int fh)
{
    const int stripes = (fh + kStripe - 1) / kStripe;
    parallel_for(stripes, 32, [&](int sidx) {
        const int f0 = sidx * kStripe, f1 = std::min(fh, f0 + kStripe);          // fine rows this worker may write
        // coarse rows whose taps 2y-2 .. 2y+2 (reflected at the image border, so the border stripes look at a few
        // more rows) can land in [f0, f1)
        int y0 = std::max(0, (f0 - 2) / 2 - 1), y1 = std::min(ch, (f1 + 2) / 2 + 2);
        if (f0 < 4) y0 = 0;
        if (f1 > fh - 4) y1 = ch;
        for (int y = y0; y < y1; ++y) {
            const uint8_t* row = coarse.data() + static_cast<size_t>(y) * cw;
            int ys[5];
            bool mine = false;
            for (int t = 0; t < 5; ++t) { ys[t] = reflect101(2 * y + t - 2, fh); mine |= ys[t] >= f0 && ys[t] < f1; }
            if (!mine) continue;
            for (int x = next_set(row, 0, cw); x < cw; x = next_set(row, x + 1, cw)) {
                int xs[5];
                for (int t = 0; t < 5; ++t) xs[t] = reflect101(2 * x + t - 2, fw);
                for (int yy : ys) {
                    if (yy < f0 || yy >= f1) continue;
                    uint8_t* d = fine.data() + static_cast<size_t>(yy) * fw;
                    for (int xx : xs) d[xx] = 1;
                }
            }
        }
    });
}

}  // namespace


void build_mb_plan(const std::vector<CamDesc>& desc, const std::vector<std::vector<std::vector<uint8_t>>>& weight_class,
                   const BlendArgs& blend, bool keep_level0, MbPlanHost& plan, int strip_x0, int strip_x1)
{
    const int n = static_cast<int>(desc.size());
    const int nlev = blend.nlev, nb = nlev - 1;
    int a_tw0, a_th0, b_t, c_tw, c_th;
    mb_tile_dims(&a_tw0, &a_th0, &b_t, &c_tw, &c_th);

    const bool timing = std::getenv("MST_UPDATE_TIMING") != nullptr;
    auto t0 = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
        if (!timing) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mst_update]   plan: %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    plan = MbPlanHost{};
    plan.cams.resize(n);
    // Canvas split over GPUs: the canvas columns of every level that the strip's panorama columns can depend on --
    // level l + 1 from level l by the pyrUp reach the canvas kernels use for their tiles (blend_region).  Weights
    // outside count as zero below; nothing else changes: the lists then hold exactly the work the strip needs, and
    // inside these ranges every value is the one the whole-canvas run computes.
    strip_x0 = std::max(0, strip_x0);
    strip_x1 = std::min(blend.cw[0], strip_x1);
    const bool striped = strip_x0 > 0 || strip_x1 < blend.cw[0];
    // the level-l columns a strip [x0, x1) of the panorama can depend on
    auto strip_levels = [&](int x0, int x1, std::vector<int>& a0, std::vector<int>& a1) {
        a0.assign(nlev, 0); a1.assign(nlev, 0);
        a0[0] = x0; a1[0] = x1;
        for (int l = 1; l < nlev; ++l) {
            a0[l] = std::max(0, ((a0[l - 1] >> 1) - 1) & ~1);
            a1[l] = std::min(blend.cw[l], (((a1[l - 1] - 1) >> 1) + 3) & ~1);
        }
    };
    std::vector<int> sx0, sx1;
    strip_levels(strip_x0, strip_x1, sx0, sx1);
    // (split canvas stage: k_blend_coarse stores whole tile interiors of level 2 -- no clip there)
    if (striped && nlev >= 2 && !blend.split) { plan.clip_x0 = sx0[1]; plan.clip_x1 = sx1[1]; }
    // nz[l] of camera c for the levels lmin .. top, cut to the strip's columns a0[l] .. a1[l]
    auto fill_nz = [&](int c, MbCameraNeed& cn, int lmin, bool cut, const std::vector<int>& a0, const std::vector<int>& a1) {
        const CamDesc& d = desc[c];
        cn.nz.resize(nlev);
        cn.need.resize(nlev);
        for (int l = lmin; l < nlev; ++l) {
            const size_t px = static_cast<size_t>(d.lw[l]) * d.lh[l];
            cn.nz[l].resize(px);
            const uint8_t* w = weight_class[c][l].data();
            uint8_t* o = cn.nz[l].data();
            const int chunks = static_cast<int>((px + (size_t{1} << 18) - 1) >> 18);
            parallel_for(chunks, 32, [&](int k) {
                const size_t i0 = static_cast<size_t>(k) << 18, i1 = std::min(px, i0 + (size_t{1} << 18));
                for (size_t i = i0; i < i1; ++i) o[i] = w[i] & 1u;
            });
            if (cut) {
                const int lo = std::max(0, std::min(d.lw[l], a0[l] - d.ox[l])), hi = std::max(lo, std::min(d.lw[l], a1[l] - d.ox[l]));
                for (int y = 0; y < d.lh[l]; ++y) {
                    uint8_t* row = o + static_cast<size_t>(y) * d.lw[l];
                    std::memset(row, 0, lo);
                    std::memset(row + hi, 0, d.lw[l] - hi);
                }
            }
        }
    };
    // need[l] for l = from .. to (downwards); need[from + 1] must be there unless from is the top level
    auto fill_need = [&](int c, MbCameraNeed& cn, int from, int to) {
        const CamDesc& d = desc[c];
        for (int l = from; l >= to; --l) {
            cn.need[l] = cn.nz[l];
            if (l >= 1) mark_parents(cn.nz[l - 1], d.lw[l - 1], d.lh[l - 1], cn.need[l], d.lw[l], d.lh[l]);
            if (l < nb) mark_footprint(cn.need[l + 1], d.lw[l + 1], d.lh[l + 1], cn.need[l], d.lw[l], d.lh[l]);
        }
    };
    parallel_for(n, 16, [&](int c) {
        fill_nz(c, plan.cams[c], 0, striped, sx0, sx1);
        fill_need(c, plan.cams[c], nb, 0);
    });

    mark("nz / need per camera");
    // ---- kernel A: level-0 tiles of a_tw0 x a_th0 pixels
    {
        std::vector<std::vector<uint32_t>> per_cam(n);
        parallel_for(n, 16, [&](int c) {
            const CamDesc& d = desc[c];
            const int tiles_x = (d.lw[0] + a_tw0 - 1) / a_tw0, tiles_y = (d.lh[0] + a_th0 - 1) / a_th0;
            // used: some pixel of the tile is needed; weighted: some pixel has a level-0 weight.  A tile that is needed only
            // as pyrDown input (bit 31 of its list entry) keeps its level 0 on chip: nothing ever reads it from memory.
            std::vector<uint8_t> used(static_cast<size_t>(tiles_x) * tiles_y, 0), weighted(used.size(), 0);
            const std::vector<uint8_t>& need0 = plan.cams[c].need[0];
            const std::vector<uint8_t>& nz0 = plan.cams[c].nz[0];
            parallel_for(tiles_y, 32, [&](int ty) {                       // a tile row's flags belong to one worker
                for (int pass = 0; pass < 2; ++pass) {
                    uint8_t* urow = (pass ? weighted : used).data() + static_cast<size_t>(ty) * tiles_x;
                    const std::vector<uint8_t>& map = pass ? nz0 : need0;
                    for (int y = ty * a_th0; y < std::min(d.lh[0], (ty + 1) * a_th0); ++y) {
                        const uint8_t* row = map.data() + static_cast<size_t>(y) * d.lw[0];
                        for (int x = next_set(row, 0, d.lw[0]); x < d.lw[0];) {
                            const int t = x / a_tw0;
                            urow[t] = 1;
                            x = next_set(row, (t + 1) * a_tw0, d.lw[0]);   // the rest of this tile's columns cannot add anything
                        }
                    }
                }
            });
            for (int ty = 0; ty < tiles_y; ++ty)
                for (int tx = 0; tx < tiles_x; ++tx) {
                    const size_t ti = static_cast<size_t>(ty) * tiles_x + tx;
                    if (used[ti])
                        per_cam[c].push_back(static_cast<uint32_t>(c) << 24 | static_cast<uint32_t>(ty) << 12 | tx |
                                             ((weighted[ti] || keep_level0) ? 0u : 0x80000000u));
                }
        });
        for (int c = 0; c < n; ++c) {
            plan.warp_cam_start.push_back(static_cast<int>(plan.warp_tiles.size()));
            plan.warp_tiles.insert(plan.warp_tiles.end(), per_cam[c].begin(), per_cam[c].end());
        }
    }

    plan.warp_cam_start.push_back(static_cast<int>(plan.warp_tiles.size()));

    // ---- kernel B: chains of pyrDown levels, tiles of b_t x b_t pixels at the base level
    for (int base = 1; base < nb;) {
        const int k = std::min(mb_chain_depth(), nb - base);
        MbPlanHost::ChainStep step;
        step.base = base;
        step.k = k;
        for (int c = 0; c < n; ++c) {
            step.cam_start.push_back(static_cast<int>(step.tiles.size()));
            const CamDesc& d = desc[c];
            const int tiles_x = (d.lw[base] + b_t - 1) / b_t, tiles_y = (d.lh[base] + b_t - 1) / b_t;
            std::vector<uint8_t> used(static_cast<size_t>(tiles_x) * tiles_y, 0);
            for (int j = 1; j <= k; ++j) {
                int log2_bt = 0;
                while ((1 << log2_bt) < b_t) ++log2_bt;
                const int l = base + j, shift = log2_bt - j;
                const std::vector<uint8_t>& need = plan.cams[c].need[l];
                for (int y = 0; y < d.lh[l]; ++y) {
                    const uint8_t* row = need.data() + static_cast<size_t>(y) * d.lw[l];
                    uint8_t* urow = used.data() + static_cast<size_t>(y >> shift) * tiles_x;
                    for (int x = next_set(row, 0, d.lw[l]); x < d.lw[l];) {
                        const int t = x >> shift;
                        urow[t] = 1;
                        x = next_set(row, std::min(d.lw[l], (t + 1) << shift), d.lw[l]);
                    }
                }
            }
            for (int ty = 0; ty < tiles_y; ++ty)
                for (int tx = 0; tx < tiles_x; ++tx)
                    if (used[static_cast<size_t>(ty) * tiles_x + tx])
                        step.tiles.push_back(static_cast<uint32_t>(c) << 24 | static_cast<uint32_t>(ty) << 12 | tx);
        }
        step.cam_start.push_back(static_cast<int>(step.tiles.size()));
        // the streaming form for the large levels (every source level of the step must have even sizes)
        bool stream = base <= 2;
        for (int c = 0; c < n && stream; ++c)
            for (int j = 0; j < k; ++j) stream = stream && (desc[c].lw[base + j] % 2 == 0) && (desc[c].lh[base + j] % 2 == 0);
        if (stream) {
            const int ur = mb_stream_unit_rows(), uc = mb_stream_unit_cols();
            for (int j = 0; j < k; ++j) {
                const int l = base + 1 + j;
                for (int c = 0; c < n; ++c) {
                    step.unit_cam_start[j].push_back(static_cast<int>(step.units[j].size()));
                    const CamDesc& d = desc[c];
                    const int ux_n = (d.lw[l] + uc - 1) / uc, uy_n = (d.lh[l] + ur - 1) / ur;
                    std::vector<uint8_t> used(static_cast<size_t>(ux_n) * uy_n, 0);
                    const std::vector<uint8_t>& need = plan.cams[c].need[l];
                    for (int y = 0; y < d.lh[l]; ++y) {
                        const uint8_t* row = need.data() + static_cast<size_t>(y) * d.lw[l];
                        uint8_t* urow = used.data() + static_cast<size_t>(y / ur) * ux_n;
                        for (int x = next_set(row, 0, d.lw[l]); x < d.lw[l];) {
                            const int t = x / uc;
                            urow[t] = 1;
                            x = next_set(row, std::min(d.lw[l], (t + 1) * uc), d.lw[l]);
                        }
                    }
                    for (int uy = 0; uy < uy_n; ++uy)
                        for (int ux = 0; ux < ux_n; ++ux)
                            if (used[static_cast<size_t>(uy) * ux_n + ux])
                                step.units[j].push_back(static_cast<uint32_t>(c) << 24 | static_cast<uint32_t>(uy) << 12 | ux);
                }
                step.unit_cam_start[j].push_back(static_cast<int>(step.units[j].size()));
            }
        }
        plan.chain.push_back(std::move(step));
        base += k;
    }

    // ---- hand-off payload of the camera-sharded mode: tile interiors, level by level
    {
        const int a_tw1 = a_tw0 / 2, a_th1 = a_th0 / 2;
        for (int c = 0; c < n; ++c) {
            plan.payload_cam_start.push_back(static_cast<int>(plan.payload_rects.size()));
            const CamDesc& d = desc[c];
            uint32_t off = 0;
            auto add = [&](int level, int x0, int y0, int w, int h) {
                w = std::min(w, d.lw[level] - x0);
                h = std::min(h, d.lh[level] - y0);
                if (w <= 0 || h <= 0) return;
                plan.payload_rects.push_back(PayloadRect{level, x0, y0, w, h, off});
                // (`off` counts dwords: a pixel of the levels above 0, three quarters of one -- packed BGR8, rounded up per
                // rectangle -- on level 0)
                const uint32_t px = static_cast<uint32_t>(w) * static_cast<uint32_t>(h);
                off += level == 0 ? (3u * px + 3u) / 4u : px;
            };
            for (int i = plan.warp_cam_start[c]; i < plan.warp_cam_start[c + 1]; ++i) {
                const uint32_t t = plan.warp_tiles[i];
                const int ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
                if (!(t >> 31)) add(0, tx * a_tw0, ty * a_th0, a_tw0, a_th0);      // (level 0 the canvas never reads is not shipped either)
                if (nlev >= 2) add(1, tx * a_tw1, ty * a_th1, a_tw1, a_th1);
            }
            for (const MbPlanHost::ChainStep& step : plan.chain)
                for (int i = step.cam_start[c]; i < step.cam_start[c + 1]; ++i) {
                    const uint32_t t = step.tiles[i];
                    const int ty = (t >> 12) & 0xFFF, tx = t & 0xFFF;
                    for (int j = 1; j <= step.k; ++j)
                        add(step.base + j, (tx * b_t) >> j, (ty * b_t) >> j, b_t >> j, b_t >> j);
                }
            plan.payload_px.push_back(off);
        }
        plan.payload_cam_start.push_back(static_cast<int>(plan.payload_rects.size()));
    }
    mark("warp + chain tile lists");
    // ---- canvas kernels: panorama tiles (k_blend_fine, level 0) and level-1 tiles (k_blend_coarse, levels 1 .. top)
    const int cc_tw = mb_coarse_tile_width();
    plan.blend_tiles_x = (blend.cw[0] + c_tw - 1) / c_tw;
    plan.blend_tiles_y = (blend.ch[0] + c_th - 1) / c_th;
    plan.block_cams.assign(static_cast<size_t>(plan.blend_tiles_x) * plan.blend_tiles_y, 0u);
    // any non-zero weight of camera c at level l inside [x0,x1) x [y0,y1) of the canvas level?  all exactly `unit`
    // (and the camera's window covering the rectangle)?
    auto any_nonzero = [&](int c, int l, int rx0, int ry0, int rx1, int ry1) {
        const CamDesc& d = desc[c];
        const int x0 = std::max(rx0 - d.ox[l], 0), y0 = std::max(ry0 - d.oy[l], 0);
        const int x1 = std::min(rx1 - d.ox[l], d.lw[l]), y1 = std::min(ry1 - d.oy[l], d.lh[l]);
        if (x0 >= x1 || y0 >= y1) return false;
        const std::vector<uint8_t>& nz = plan.cams[c].nz[l];
        for (int y = y0; y < y1; ++y) {
            const uint8_t* row = nz.data() + static_cast<size_t>(y) * d.lw[l];
            if (next_set(row, x0, x1) < x1) return true;
        }
        return false;
    };
    auto all_unit = [&](int c, int l, int rx0, int ry0, int rx1, int ry1) {
        const CamDesc& d = desc[c];
        const int x0 = rx0 - d.ox[l], y0 = ry0 - d.oy[l], x1 = rx1 - d.ox[l], y1 = ry1 - d.oy[l];
        if (x0 < 0 || y0 < 0 || x1 > d.lw[l] || y1 > d.lh[l]) return false;
        const std::vector<uint8_t>& w = weight_class[c][l];
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x)
                if (!(w[static_cast<size_t>(y) * d.lw[l] + x] & 2u)) return false;
        return true;
    };
    auto covers = [&](int c, int l, int rx0, int ry0, int rx1, int ry1) {       // camera window over the whole rectangle?
        const CamDesc& d = desc[c];
        return rx0 - d.ox[l] >= 0 && ry0 - d.oy[l] >= 0 && rx1 - d.ox[l] <= d.lw[l] && ry1 - d.oy[l] <= d.lh[l];
    };
    // Panorama tiles.  A tile whose level 0 is owned by ONE camera with a weight of exactly 1.0f everywhere inside the
    // final crop (pixels outside are never stored), and no other camera with a non-zero weight there, takes the
    // integer-only band path (bit 31 + that camera's bit alone).
    parallel_for(plan.blend_tiles_y, 64, [&](int ty) {
        for (int tx = 0; tx < plan.blend_tiles_x; ++tx) {
            int r[4];
            mb_blend_region(blend, 0, tx, ty, 0, r);
            uint32_t bits = 0;
            for (int c = 0; c < n; ++c)
                if (any_nonzero(c, 0, r[0], r[1], r[2], r[3])) bits |= 1u << c;
            const int cx1 = std::min(r[2], blend.final_w), cy1 = std::min(r[3], blend.final_h);
            if (r[0] < cx1 && r[1] < cy1) {
                uint32_t lbits = 0;
                for (int c = 0; c < n; ++c)
                    if (any_nonzero(c, 0, r[0], r[1], cx1, cy1)) lbits |= 1u << c;
                if (lbits != 0 && (lbits & (lbits - 1)) == 0) {
                    const int c = __builtin_ctz(lbits);
                    if (all_unit(c, 0, r[0], r[1], cx1, cy1) && covers(c, 0, r[0], r[1], r[2], r[3])) bits = lbits | 0x80000000u;
                }
            }
            // bit 30: one or two cameras, every weight exactly 0 or 1 and never two non-zero on a pixel, both windows
            // over the whole tile -> the integer strip path with per-pixel masks (a whole tile inside the crop only)
            const int nc = __builtin_popcount(bits & 0x3FFFFFFFu);
            if (!(bits >> 31) && nc >= 1 && nc <= 2 && r[2] - r[0] == c_tw && r[3] - r[1] == c_th && r[2] <= blend.final_w &&
                r[3] <= blend.final_h) {
                int cs[2] = {__builtin_ctz(bits), nc == 2 ? __builtin_ctz(bits & (bits - 1)) : -1};
                bool ok = true;
                for (int k = 0; k < nc; ++k) ok = ok && covers(cs[k], 0, r[0], r[1], r[2], r[3]);
                for (int y = r[1]; ok && y < r[3]; ++y) {
                    const uint8_t* w0 = weight_class[cs[0]][0].data() + static_cast<size_t>(y - desc[cs[0]].oy[0]) * desc[cs[0]].lw[0] - desc[cs[0]].ox[0];
                    const uint8_t* w1 = nc == 2 ? weight_class[cs[1]][0].data() + static_cast<size_t>(y - desc[cs[1]].oy[0]) * desc[cs[1]].lw[0] - desc[cs[1]].ox[0] : nullptr;
                    for (int x = r[0]; x < r[2]; ++x) {
                        const uint8_t a0 = w0[x], a1 = w1 ? w1[x] : 0;
                        if ((a0 == 1) | (a1 == 1) | ((a0 & a1 & 1) != 0)) { ok = false; break; }   // fractional, or both non-zero
                    }
                }
                if (ok) bits |= 0x40000000u;
            }
            plan.block_cams[static_cast<size_t>(ty) * plan.blend_tiles_x + tx] = bits;
        }
    });
    // launch order: every tile is resident at once on a 256-CU part, so the dispatcher's round-robin placement is
    // the only load balancing there is -- hand out the expensive tiles (several cameras under the tile, float
    // path) first so that they land on different CUs, then the single-camera integer-path tiles
    auto ordered = [](const std::vector<uint32_t>& entries, auto cost) {
        std::vector<uint32_t> order(entries.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = static_cast<uint32_t>(i);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost(a) > cost(b); });
        return order;
    };
    plan.block_order = ordered(plan.block_cams, [&](uint32_t tile) {
        const uint32_t e = plan.block_cams[tile];
        if (e >> 31) return 4;
        const int nc = __builtin_popcount(e & 0x3FFFFFFFu);
        return (e >> 30) ? 4 + 2 * nc : 10 + 2 * nc;
    });
    if (striped) {            // only the strip's panorama tiles are written
        std::vector<uint32_t> kept;
        for (uint32_t tile : plan.block_order) {
            const int x = static_cast<int>(tile % plan.blend_tiles_x) * c_tw;
            if (x < strip_x1 && x + c_tw > strip_x0) kept.push_back(tile);
        }
        plan.block_order.swap(kept);
    }
    // Level-1 tiles, once per tile height the launcher may pick (MbPlanHost::coarse).  Levels on which the tile's region
    // (tile + pyrUp halo) is owned by ONE camera with a weight of exactly 1.0f everywhere and no other camera has a
    // non-zero weight there go down the integer path as well; a tile whose levels are all owned by the same camera
    // is "pure" (bit 31 of cams): no float arithmetic.
    int pure_tiles = 0, own1_tiles = 0;
    for (int v = 0; v < 2; ++v) {
        MbPlanHost::CoarseTiles& ct = plan.coarse[v];
        ct = MbPlanHost::CoarseTiles{};
        ct.th = mb_coarse_tile_height(v);
        if (nlev <= 1) continue;
        BlendArgs geo = blend;
        geo.cc_th = ct.th;
        geo.clip_x0 = plan.clip_x0; geo.clip_x1 = plan.clip_x1;
        ct.tiles_x = (blend.cw[1] + cc_tw - 1) / cc_tw;
        ct.tiles_y = (blend.ch[1] + ct.th - 1) / ct.th;
        ct.cams.assign(static_cast<size_t>(ct.tiles_x) * ct.tiles_y, 0u);
        ct.pure.assign(ct.cams.size(), 0u);
        std::vector<std::vector<int>> row_area(ct.tiles_y, std::vector<int>(nlev, 0));
        parallel_for(ct.tiles_y, 64, [&](int ty) {
            std::vector<int>& max_area = row_area[ty];
            for (int tx = 0; tx < ct.tiles_x; ++tx) {
                uint32_t bits = 0, level_mask = 0;
                int owner = -1;
                for (int l = 1; l < nlev; ++l) {
                    int r[4];
                    mb_blend_region(geo, 1, tx, ty, l, r);
                    max_area[l] = std::max(max_area[l], mb_blend_region_lds_pixels(geo, 1, tx, ty, l));
                    uint32_t lbits = 0;
                    for (int c = 0; c < n; ++c)
                        if (any_nonzero(c, l, r[0], r[1], r[2], r[3])) lbits |= 1u << c;
                    bits |= lbits;
                    if (lbits != 0 && (lbits & (lbits - 1)) == 0) {
                        const int c = __builtin_ctz(lbits);
                        if ((owner < 0 || owner == c) && all_unit(c, l, r[0], r[1], r[2], r[3])) {
                            owner = c;
                            level_mask |= 1u << l;
                        }
                    }
                }
                const size_t ti = static_cast<size_t>(ty) * ct.tiles_x + tx;
                if (owner >= 0 && level_mask == ((1u << nlev) - 2u) && bits == (1u << owner)) bits |= 0x80000000u;
                ct.cams[ti] = bits;
                ct.pure[ti] = owner >= 0 ? (level_mask | static_cast<uint32_t>(owner) << 16) : 0u;
            }
        });
        std::vector<int> max_area(nlev, 0);
        for (const std::vector<int>& ra : row_area)
            for (int l = 0; l < nlev; ++l) max_area[l] = std::max(max_area[l], ra[l]);
        ct.order = ordered(ct.cams, [&](uint32_t tile) {
            const uint32_t e = ct.cams[tile];
            if (e >> 31) return 4;                                  // relative: integer path everywhere
            const bool l1_owned = ct.pure[tile] & 2u;               // float bands on some coarser levels only
            return (l1_owned ? 6 : 10) + 2 * __builtin_popcount(e);
        });
        if (striped) {        // only the level-1 tiles the strip's panorama tiles read
            std::vector<uint32_t> kept, kept2;
            for (uint32_t tile : ct.order) {
                const int x = static_cast<int>(tile % ct.tiles_x) * cc_tw;
                if (x < sx1[1] && x + cc_tw > sx0[1]) kept.push_back(tile);
                // (level-2 interior of the tile: columns [x / 2, (x + cc_tw) / 2); the strip's level-1 tiles read level 2 over sx0[2] .. sx1[2])
                if (nlev >= 3 && x / 2 < sx1[2] && (x + cc_tw) / 2 > sx0[2]) kept2.push_back(tile);
            }
            ct.order.swap(kept);
            ct.order2.swap(kept2);
        }
        int off = 0;
        for (int l = 2; l < nlev; ++l) {
            ct.lds_off[l] = off;
            off += (max_area[l] + 1) & ~1;
        }
        ct.lds_bytes = static_cast<size_t>(off) * 8;
        if (v == 1) {
            for (uint32_t e : ct.cams) pure_tiles += e >> 31;
            for (uint32_t e : ct.pure) own1_tiles += (e >> 1) & 1u;
        }
    }
    mark("canvas tiles");
    if (timing) {
        int owned = 0, binary = 0;
        for (uint32_t e : plan.block_cams) { owned += e >> 31; binary += (e >> 30) & 1u; }
        std::fprintf(stderr, "[mst_update]   plan: %zu panorama tiles (%d with level 0 owned, %d more with 0/1 masks), %zu level-1 tiles (%d pure, %d with level 1 owned)\n",
                     plan.block_cams.size(), owned, binary, plan.coarse[1].cams.size(), pure_tiles, own1_tiles);
    }
}

}  // namespace mst
