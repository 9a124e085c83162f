// engine.cpp -- see engine.hpp.
#include "engine.hpp"
#include "../../include/mi355_ingest.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kernels.hpp"

namespace mst {

void check_hip(hipError_t e, const char* what)
{
    if (e != hipSuccess)
        throw Error{e == hipErrorOutOfMemory ? MST_ERR_OUT_OF_MEMORY : MST_ERR_HIP,
                    std::string(what) + ": " + hipGetErrorString(e)};
}

void select_device(int device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        throw Error{MST_ERR_NO_DEVICE, "no HIP device is visible (this engine has no CPU fallback)"};
    if (device < 0 || device >= count)
        throw Error{MST_ERR_NO_DEVICE, "device ordinal out of range"};
    check_hip(hipSetDevice(device), "hipSetDevice");
}

// ---- DevicePool ------------------------------------------------------------------------------

namespace {
constexpr size_t kPoolAlign = 256, kPoolSlab = size_t{64} << 20;
}

SlabCache::~SlabCache()
{
    for (void* p : free_) (void)hipFree(p);
}

void* SlabCache::take()
{
    std::lock_guard<std::mutex> lk(mu_);
    if (free_.empty()) return nullptr;
    void* p = free_.back();
    free_.pop_back();
    return p;
}

void SlabCache::give(void* slab)
{
    std::lock_guard<std::mutex> lk(mu_);
    free_.push_back(slab);
}

DevicePool::~DevicePool()
{
    for (void* p : big_) (void)hipFree(p);
    for (void* p : ptrs_) {
        if (cache_) cache_->give(p);
        else (void)hipFree(p);
    }
}

void* DevicePool::alloc_bytes(size_t bytes)
{
    const size_t need = (std::max<size_t>(bytes, 16) + kPoolAlign - 1) & ~(kPoolAlign - 1);
    if (need >= kPoolSlab) {         // a request of slab size or more gets an allocation of its own
        void* p = nullptr;
        check_hip(hipMalloc(&p, need + kPoolAlign), "hipMalloc");     // + guard: kernels may read a few bytes past a buffer
        big_.push_back(p);
        total_ += bytes;
        return p;
    }
    if (!cache_) {                   // a stage-level call: a few small buffers, one allocation each (no 64 MiB slab)
        void* p = nullptr;
        check_hip(hipMalloc(&p, need + kPoolAlign), "hipMalloc");
        big_.push_back(p);
        total_ += bytes;
        return p;
    }
    if (need > left_) {
        void* p = cache_ ? cache_->take() : nullptr;
        if (!p) check_hip(hipMalloc(&p, kPoolSlab + kPoolAlign), "hipMalloc");
        ptrs_.push_back(p);
        cur_ = static_cast<char*>(p);
        left_ = kPoolSlab;
    }
    void* out = cur_;
    cur_ += need;
    left_ -= need;
    total_ += bytes;
    return out;
}

template <typename T> T* DevicePool::upload(const T* host, size_t n, hipStream_t s)
{
    T* d = alloc<T>(n);
    if (n) check_hip(hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, s), "hipMemcpyAsync(H2D)");
    return d;
}
template float* DevicePool::upload<float>(const float*, size_t, hipStream_t);
template int32_t* DevicePool::upload<int32_t>(const int32_t*, size_t, hipStream_t);
template uint8_t* DevicePool::upload<uint8_t>(const uint8_t*, size_t, hipStream_t);
template CamDesc* DevicePool::upload<CamDesc>(const CamDesc*, size_t, hipStream_t);
template ColEntry* DevicePool::upload<ColEntry>(const ColEntry*, size_t, hipStream_t);
template RowEntry* DevicePool::upload<RowEntry>(const RowEntry*, size_t, hipStream_t);
template uint32_t* DevicePool::upload<uint32_t>(const uint32_t*, size_t, hipStream_t);
template int16_t* DevicePool::upload<int16_t>(const int16_t*, size_t, hipStream_t);
template uint16_t* DevicePool::upload<uint16_t>(const uint16_t*, size_t, hipStream_t);
template UndistortCam* DevicePool::upload<UndistortCam>(const UndistortCam*, size_t, hipStream_t);
template PayloadRect* DevicePool::upload<PayloadRect>(const PayloadRect*, size_t, hipStream_t);
template ComposeTile* DevicePool::upload<ComposeTile>(const ComposeTile*, size_t, hipStream_t);
template ComposeSeg* DevicePool::upload<ComposeSeg>(const ComposeSeg*, size_t, hipStream_t);
template WarpWin* DevicePool::upload<WarpWin>(const WarpWin*, size_t, hipStream_t);
template int2* DevicePool::upload<int2>(const int2*, size_t, hipStream_t);
template FeedImage* DevicePool::upload<FeedImage>(const FeedImage*, size_t, hipStream_t);
template FeedPair* DevicePool::upload<FeedPair>(const FeedPair*, size_t, hipStream_t);

CalibState::~CalibState()
{
    // kernels of an earlier compose may still read this state
    (void)hipDeviceSynchronize();
}

// ---- camera descriptor -------------------------------------------------------------------------

// Level 0 of a camera's pyramid is packed BGR8: rows of whole four-pixel groups (12 bytes), so that every group the warp
// kernel stores and every pixel pair the canvas kernel loads lies inside its row; the levels above are a dword a pixel.
// A level's footprint in dwords, rounded up so that the next level starts 16-byte aligned and level 0 carries the slack
// its 8-byte pair loads may touch behind the last pixel.
int level0_pitch(int lw0) { return ((lw0 + 3) / 4) * 4 * kL0Bpp; }
size_t level_dwords(const CamDesc& d, int l)
{
    const size_t dw = l == 0 ? (static_cast<size_t>(d.l0_pitch) * d.lh[0] + 16 + 3) / 4 : static_cast<size_t>(d.lw[l]) * d.lh[l];
    return (dw + 3) & ~size_t{3};
}

void fill_camera_desc(DevicePool& pool, hipStream_t s, const Projector& proj, const Rect& roi, int src_w, int src_h,
                      int expos_kind, const float* gain_map, int gw, int gh, float gain, CamDesc& d)
{
    d.kind = proj.kind;
    d.roi_w = roi.width; d.roi_h = roi.height;
    d.src_w = src_w; d.src_h = src_h;
    std::memcpy(d.kr, proj.k_rinv, sizeof(d.kr));
    d.one_minus_t2 = 1 - proj.t[2];
    // host tables are pageable and freed on return: the async copies must have consumed them
    MapTables t = build_map_tables(proj, roi);
    d.col_a = pool.upload(t.col_a, s); d.col_b = pool.upload(t.col_b, s);
    d.row_a = pool.upload(t.row_a, s); d.row_b = pool.upload(t.row_b, s);
    d.xmap = d.ymap = nullptr;
    std::vector<float> xmap, ymap;
    if (!separable_map(proj.kind)) {
        const size_t px = static_cast<size_t>(roi.width) * roi.height;
        xmap.resize(px); ymap.resize(px);
        build_maps_host(proj, roi, xmap.data(), ymap.data());
        d.xmap = pool.upload(xmap, s);
        d.ymap = pool.upload(ymap, s);
    }
    d.expos = expos_kind;
    d.gain = gain;
    d.gain_map = nullptr; d.gw = d.gh = 0;
    ResizeAxis ax, ay;
    if (expos_kind == MST_EXPOS_GAIN_BLOCKS) {
        if (!gain_map || gw <= 0 || gh <= 0) throw Error{MST_ERR_INVALID_ARG, "GAIN_BLOCKS needs a gain map"};
        d.gain_map = pool.upload(gain_map, static_cast<size_t>(gw) * gh, s);
        d.gw = gw; d.gh = gh;
        ax = build_gain_axis_x(gw, roi.width);
        ay = build_gain_axis_y(gh, roi.height);
        d.gx0 = pool.upload(ax.ofs0, s); d.gx1 = pool.upload(ax.ofs1, s);
        d.gxw0 = pool.upload(ax.w0, s);  d.gxw1 = pool.upload(ax.w1, s);
        d.gy0 = pool.upload(ay.ofs0, s); d.gy1 = pool.upload(ay.ofs1, s);
        d.gyw0 = pool.upload(ay.w0, s);  d.gyw1 = pool.upload(ay.w1, s);
    }
    // the same constants packed per column / per row for the fused kernels (device_types.hpp)
    const float* kr = proj.k_rinv;
    std::vector<ColEntry> cols(roi.width);
    for (int i = 0; i < roi.width; ++i) {
        ColEntry& e = cols[i];
        e = ColEntry{};
        e.ca = t.col_a[i]; e.cb = t.col_b[i];
        for (int r = 0; r < 3; ++r) {
            e.a[r] = kr[3 * r] * e.ca;
            e.c[r] = proj.kind == kPlane ? kr[3 * r + 2] * d.one_minus_t2 : kr[3 * r + 2] * e.cb;
        }
        if (expos_kind == MST_EXPOS_GAIN_BLOCKS) {
            e.gx0 = ax.ofs0[i]; e.gx1 = ax.ofs1[i]; e.gxw0 = ax.w0[i]; e.gxw1 = ax.w1[i];
        }
    }
    std::vector<RowEntry> rows(roi.height);
    for (int j = 0; j < roi.height; ++j) {
        RowEntry& e = rows[j];
        e = RowEntry{};
        e.ra = t.row_a[j];
        const float y_ = proj.kind == kSpherical ? t.row_b[j] : t.row_a[j];
        for (int r = 0; r < 3; ++r) e.b[r] = kr[3 * r + 1] * y_;
        if (expos_kind == MST_EXPOS_GAIN_BLOCKS) {
            e.gy0 = ay.ofs0[j]; e.gy1 = ay.ofs1[j]; e.gyw0 = ay.w0[j]; e.gyw1 = ay.w1[j];
        }
    }
    d.cols = pool.upload(cols, s);
    d.rows = pool.upload(rows, s);
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
}

// busy: other launch sequences run beside this one (mst_config.overlap_depth >= 3) -- the canvas kernels then take the
// large-batch shape (fewer, longer blocks) at any batch size: what the small-batch shape buys a lone launch, filling the chip,
// the neighbours already provide (C3, one-frame-set calls, 4 in flight: 15.0 k -> 16.0 k panoramas/s)
BlendArgs CalibState::blend_for(int frame_sets, int* n_coarse, size_t* lds_bytes, bool busy) const
{
    const int v = busy ? 1 : mb_launch_variant(frame_sets);
    const MbPlanHost::CoarseTiles& ct = plan.coarse[v];
    BlendArgs ba = blend;
    ba.cc_th = ct.th;
    ba.fine_threads = mb_fine_threads(v);
    ba.coarse_cams = coarse_dev[v].cams;
    ba.coarse_pure = coarse_dev[v].pure;
    ba.coarse_order = coarse_dev[v].order;
    ba.coarse_order2 = coarse_dev[v].order2;
    ba.n_coarse2 = static_cast<int>(ct.order2.size());
    ba.coarse_tiles_x = ct.tiles_x;
    for (int l = 0; l < kMaxLevels; ++l) ba.lds_off[l] = ct.lds_off[l];
    *n_coarse = static_cast<int>(ct.order.size());
    *lds_bytes = ct.lds_bytes;
    return ba;
}

// ---- StageTimer --------------------------------------------------------------------------------

StageTimer::~StageTimer()
{
    for (hipEvent_t e : events_) (void)hipEventDestroy(e);
}

void StageTimer::begin(hipStream_t s)
{
    names_.clear();
    used_ = 0;
    mark(nullptr, s);
}

void StageTimer::mark(const char* name, hipStream_t s)
{
    if (used_ == events_.size()) {
        hipEvent_t e;
        check_hip(hipEventCreate(&e), "hipEventCreate");
        events_.push_back(e);
    }
    check_hip(hipEventRecord(events_[used_++], s), "hipEventRecord");
    if (name) names_.push_back(name);
}

void StageTimer::collect(std::vector<std::pair<std::string, float>>& out)
{
    out.clear();
    if (used_ < 2) return;
    check_hip(hipEventSynchronize(events_[used_ - 1]), "hipEventSynchronize");
    for (size_t i = 0; i + 1 < used_; ++i) {
        float ms = 0.f;
        check_hip(hipEventElapsedTime(&ms, events_[i], events_[i + 1]), "hipEventElapsedTime");
        out.emplace_back(names_[i], ms);
    }
}

// ---- Engine ------------------------------------------------------------------------------------

Engine::Engine(const mst_config& cfg) : cfg_(cfg)
{
    if (cfg_.num_bands <= 0) cfg_.num_bands = 5;
    if (cfg_.feather_sharpness <= 0.f) cfg_.feather_sharpness = 0.02f;
}

void Engine::init()
{
    select_device(cfg_.device);
    // LDS a block may ask for: the staged samplers size their windows against it (MI355X: 160 KB).  The composites' 40 KB
    // budget is a compile-time constant of their tables, so a part below it is refused here -- loudly -- instead of
    // failing at the first launch; the warp kernel's window classes are clipped to it per calibration.
    int lds = 0;
    check_hip(hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg_.device), "hipDeviceGetAttribute(LDS)");
    max_lds_bytes_ = static_cast<size_t>(std::max(lds, 0));
    if (max_lds_bytes_ < static_cast<size_t>(kCompLdsPx) * 4)
        throw Error{MST_ERR_NO_DEVICE, "this device offers less LDS per block than the staged kernels need (40 KB; written for gfx950)"};
    check_hip(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking), "hipStreamCreate");
    check_hip(hipStreamCreateWithFlags(&calib_stream_, hipStreamNonBlocking), "hipStreamCreate");
    check_hip(hipStreamCreateWithFlags(&front_stream_, hipStreamNonBlocking), "hipStreamCreate");
    overlap_depth_ = cfg_.overlap_depth >= 3 ? std::min<int>(kMaxOverlap, cfg_.overlap_depth) : 2;
    if (const char* v = std::getenv("MST_OVERLAP_DEPTH")) overlap_depth_ = std::max(2, std::min(kMaxOverlap, std::atoi(v)));      // (A/B runs)
    for (int i = 0; i < kMaxOverlap; ++i) {
        if (overlap_depth_ > 2) check_hip(hipStreamCreateWithFlags(&call_streams_[i], hipStreamNonBlocking), "hipStreamCreate");
        check_hip(hipEventCreateWithFlags(&ev_front_done_[i], hipEventDisableTiming), "hipEventCreate");
        check_hip(hipEventCreateWithFlags(&ev_canvas_done_[i], hipEventDisableTiming), "hipEventCreate");
    }
}

Engine::~Engine()
{
    (void)hipSetDevice(cfg_.device);
    (void)hipDeviceSynchronize();
    drop_graphs();
    current_.reset();
    latched_.reset();
    in_flight_.reset();
    retired_.clear();
    for (uint8_t* p : d_frames_) (void)hipFree(p);
    if (d_pano_) (void)hipFree(d_pano_);
    if (d_mask_) (void)hipFree(d_mask_);
    if (cls_host_) (void)hipHostFree(cls_host_);
    for (PipeSlot& ps : pipe_) {
        for (uint8_t* p : ps.d_frames) if (p) (void)hipFree(p);
        if (ps.d_pano) (void)hipFree(ps.d_pano);
        if (ps.h2d_done) (void)hipEventDestroy(ps.h2d_done);
        if (ps.compute_done) (void)hipEventDestroy(ps.compute_done);
        if (ps.d2h_done) (void)hipEventDestroy(ps.d2h_done);
    }
    if (pipe_in_) (void)hipStreamDestroy(pipe_in_);
    if (pipe_out_) (void)hipStreamDestroy(pipe_out_);
    if (stream_) (void)hipStreamDestroy(stream_);
    if (calib_stream_) (void)hipStreamDestroy(calib_stream_);
    if (front_stream_) (void)hipStreamDestroy(front_stream_);
    for (int i = 0; i < kMaxOverlap; ++i) {
        if (call_streams_[i]) (void)hipStreamDestroy(call_streams_[i]);
        if (ev_front_done_[i]) (void)hipEventDestroy(ev_front_done_[i]);
        if (ev_canvas_done_[i]) (void)hipEventDestroy(ev_canvas_done_[i]);
    }
}

std::shared_ptr<CalibState> Engine::state() const
{
    std::lock_guard<std::mutex> lk(mu_);
    return latching_ ? latched_ : current_;
}

std::shared_ptr<CalibState> Engine::acquire()
{
    std::lock_guard<std::mutex> lk(mu_);
    if (!current_) return nullptr;        // nothing published yet: the engine does not start latching on nothing
    latched_ = current_;
    latching_ = true;
    return latched_;
}

void Engine::release()
{
    std::lock_guard<std::mutex> lk(mu_);
    latching_ = false;
    latched_.reset();
}

void Engine::set_in_flight(std::shared_ptr<CalibState> st)
{
    std::lock_guard<std::mutex> lk(mu_);
    if (in_flight_ == st) return;
    if (in_flight_) retired_.push_back(std::move(in_flight_));     // pointer move only: no teardown on this thread
    in_flight_ = std::move(st);
}

void Engine::drop_retired()
{
    std::vector<std::shared_ptr<CalibState>> old;
    {
        std::lock_guard<std::mutex> lk(mu_);
        old.swap(retired_);
    }
    old.clear();          // ~CalibState (device sync + frees) outside the lock
}

void Engine::synchronize()
{
    select_device(cfg_.device);
    check_hip(hipStreamSynchronize(stream_), "hipStreamSynchronize");
}

namespace {
// MST_UPDATE_TIMING=1: wall-clock marks of the stages of mst_update on stderr (tools/time_update.py)
struct UpdateClock {
    bool on = std::getenv("MST_UPDATE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void mark(const char* what)
    {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[mst_update] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    }
};
}  // namespace

void Engine::update(const mst_calibration& calib)
{
    UpdateClock clk;
    select_device(cfg_.device);
    drop_retired();
    clk.mark("drop retired states");
    if (calib.num_cameras != cfg_.num_cameras || !calib.cameras)
        throw Error{MST_ERR_INVALID_ARG, "calibration camera count does not match the engine"};
    if (!(calib.warped_image_scale > 0.f)) throw Error{MST_ERR_INVALID_ARG, "warped_image_scale must be > 0"};
    hipStream_t s = calib_stream_;
    auto st = std::make_shared<CalibState>();
    st->pool.set_cache(slab_cache_);
    const int n = cfg_.num_cameras;
    st->n = n;
    st->blend_kind = cfg_.blend_kind;
    st->cams.resize(n);
    st->desc.assign(n, CamDesc{});
    std::vector<Rect> rois(n);

    for (int i = 0; i < n; ++i) {
        const mst_camera& c = calib.cameras[i];
        if (c.src_width <= 0 || c.src_height <= 0) throw Error{MST_ERR_INVALID_ARG, "bad source size"};
        CameraSetup& cs = st->cams[i];
        set_projector_kind(cs.proj, cfg_.warp_kind, cfg_.warp_a, cfg_.warp_b);
        cs.proj.scale = calib.warped_image_scale;
        cs.proj.set(c.K, c.R, c.T);
        cs.src_w = c.src_width; cs.src_h = c.src_height;
        cs.roi = warp_roi(cs.proj, c.src_width, c.src_height);
        if (cs.roi.width <= 0 || cs.roi.height <= 0 || cs.roi.width > 65536 || cs.roi.height > 65536)
            throw Error{MST_ERR_INVALID_ARG, "degenerate warp ROI (DRV:661-665)"};
        rois[i] = cs.roi;
        CamDesc& d = st->desc[i];
        fill_camera_desc(st->pool, s, cs.proj, cs.roi, c.src_width, c.src_height, cfg_.expos_kind, c.gain_map,
                         c.gain_w, c.gain_h, c.gain, d);
        if (cfg_.flags & MST_FLAG_LENS_COMPOSED) {
            // every camera reads its backward map from memory (one kernel instantiation per launch); cameras with lens maps
            // get them composed in: a single resampling of the raw frame, no pre-pass (mst_compose_lens_maps)
            if ((c.undistort_map1 == nullptr) != (c.undistort_map2 == nullptr)) throw Error{MST_ERR_INVALID_ARG, "undistort maps come in pairs"};
            const size_t px = static_cast<size_t>(cs.roi.width) * cs.roi.height;
            std::vector<float> xm(px), ym(px);
            backward_map_host(cs.proj, cs.roi, build_map_tables(cs.proj, cs.roi), xm.data(), ym.data());
            if (c.undistort_map1) compose_lens_maps(xm.data(), ym.data(), px, c.undistort_map1, c.undistort_map2, c.src_width, c.src_height);
            d.xmap = st->pool.upload(xm, s);
            d.ymap = st->pool.upload(ym, s);
            d.kind = std::max(d.kind, 3);          // (>= 3: the kernels take xmap / ymap, kernels_common.hpp backward_map)
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");      // xm / ym are local
        }
        const size_t mask_px = static_cast<size_t>(cs.roi.width) * cs.roi.height;
        uint8_t* d_mask = st->pool.alloc<uint8_t>(mask_px);
        d.mask = d_mask;
        if (c.seam_mask) {
            // the compose loop's mask fix-up on the device (stitcher.cpp:291-294): dilate -> resize -> & warped mask
            if (c.seam_mask_width <= 0 || c.seam_mask_height <= 0 || c.seam_mask_stride < static_cast<size_t>(c.seam_mask_width))
                throw Error{MST_ERR_INVALID_ARG, "bad seam mask"};
            const int sw = c.seam_mask_width, sh = c.seam_mask_height;
            std::vector<uint8_t> seam(static_cast<size_t>(sw) * sh);
            for (int y = 0; y < sh; ++y)
                std::memcpy(&seam[static_cast<size_t>(y) * sw], c.seam_mask + static_cast<size_t>(y) * c.seam_mask_stride, sw);
            const uint8_t* d_seam = st->pool.upload(seam, s);
            const Resize8uAxis ax = build_resize8u_axis(sw, cs.roi.width, true), ay = build_resize8u_axis(sh, cs.roi.height, false);
            const CamDesc* d_one = st->pool.upload(&d, 1, s);
            launch_mask_fixup(d_one, d_seam, sw, sh, st->pool.upload(ax.ofs0, s), st->pool.upload(ax.ofs1, s),
                              st->pool.upload(ax.c0, s), st->pool.upload(ax.c1, s), st->pool.upload(ay.ofs0, s),
                              st->pool.upload(ay.ofs1, s), st->pool.upload(ay.c0, s), st->pool.upload(ay.c1, s), d_mask,
                              cs.roi.width, cs.roi.height, s);
            check_hip(hipGetLastError(), "k_mask_fixup");
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");      // pageable host tables are local
        } else {
            if (!c.mask || c.mask_width != cs.roi.width || c.mask_height != cs.roi.height ||
                c.mask_stride < static_cast<size_t>(c.mask_width))
                throw Error{MST_ERR_INVALID_ARG, "mask must be CV_8U of exactly the camera's warpRoi size"};
            // tight copy of the mask
            if (c.mask_stride == static_cast<size_t>(cs.roi.width)) {
                check_hip(hipMemcpyAsync(d_mask, c.mask, mask_px, hipMemcpyHostToDevice, s), "hipMemcpyAsync(mask)");
            } else {
                check_hip(hipMemcpy2DAsync(d_mask, cs.roi.width, c.mask, c.mask_stride, cs.roi.width, cs.roi.height,
                                           hipMemcpyHostToDevice, s), "hipMemcpy2DAsync(mask)");
            }
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");      // the caller's mask is only borrowed
        }
        if (cfg_.blend_kind == MST_BLEND_FEATHER) {
            // createWeightMap on the device (frame-invariant: once per calibration)
            float* fw = st->pool.alloc<float>(mask_px);
            int32_t* tmp = st->pool.alloc<int32_t>(mask_px);
            launch_feather_weight_map(d_mask, cs.roi.width, cs.roi.height, cfg_.feather_sharpness, tmp, fw, s);
            check_hip(hipGetLastError(), "feather weight map kernels");
            d.feather_w = fw;
        }
    }

    clk.mark("camera tables + masks");
    // frame sets in flight: one per frame set of a batch, two for frame pipelining
    // (pipelining alternates between two halves of the sets, each as large as a batch)
    st->sets_per_slot = std::max(1, std::min<int>(cfg_.max_batch, kMaxBatch));
    st->num_sets = st->sets_per_slot;
    if ((cfg_.flags & MST_FLAG_OVERLAP_FRAMES) && (cfg_.blend_kind == MST_BLEND_MULTI_BAND || overlap_depth_ > 2))
        st->num_sets = overlap_depth_ * st->sets_per_slot;

    if (cfg_.flags & MST_FLAG_FRAMES_NV12) {        // the BGR frames the callers' NV12 frames are converted into
        std::vector<int2> sizes(n);
        st->nv12_frames.assign(static_cast<size_t>(st->num_sets) * n, nullptr);
        for (int i = 0; i < n; ++i) {
            const mst_camera& c = calib.cameras[i];
            if ((c.src_width | c.src_height) & 1) throw Error{MST_ERR_INVALID_ARG, "NV12 frames have even sizes"};
            sizes[i] = make_int2(c.src_width, c.src_height);
            st->frame_max_w = std::max(st->frame_max_w, c.src_width);
            st->frame_max_h = std::max(st->frame_max_h, c.src_height);
            for (int k = 0; k < st->num_sets; ++k)
                st->nv12_frames[static_cast<size_t>(k) * n + i] = st->pool.alloc<uint8_t>(static_cast<size_t>(c.src_width) * c.src_height * 3 + 16);
        }
        st->d_frame_sizes = st->pool.upload(sizes, s);
        check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");
    }
    // optional lens-undistort maps (DRV:232): uploaded once, applied to every frame before the warp
    st->ucams.assign(n, UndistortCam{nullptr, nullptr, 0, 0});
    st->undist_frames.assign(static_cast<size_t>(st->num_sets) * n, nullptr);
    for (int i = 0; i < n; ++i) {
        const mst_camera& c = calib.cameras[i];
        if ((!c.undistort_map1 && !c.undistort_map2) || (cfg_.flags & MST_FLAG_LENS_COMPOSED)) continue;
        if (!c.undistort_map1 || !c.undistort_map2) throw Error{MST_ERR_INVALID_ARG, "undistort maps come in pairs"};
        const size_t px = static_cast<size_t>(c.src_width) * c.src_height;
        UndistortCam& u = st->ucams[i];
        u.map1 = st->pool.upload(c.undistort_map1, px * 2, s);
        u.map2 = st->pool.upload(c.undistort_map2, px, s);
        u.w = c.src_width; u.h = c.src_height;
        st->any_undistort = true;
        st->undist_max_w = std::max(st->undist_max_w, u.w);
        st->undist_max_h = std::max(st->undist_max_h, u.h);
        for (int k = 0; k < st->num_sets; ++k)
            st->undist_frames[static_cast<size_t>(k) * n + i] = st->pool.alloc<uint8_t>(px * 3 + 16);
        check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");      // the caller's maps are only borrowed
    }
    if (st->any_undistort) st->d_ucams = st->pool.upload(st->ucams, s);
    // the pre-pass in the staged form (kernels_compose.hip): per camera one segment per 128 x 32 tile of the frame, its map
    // recast into the composites' table form; cameras whose frames are not a multiple of four pixels wide, or with a tile
    // whose source window does not fit, keep the gathering k_remap_fixed
    st->undist_plans.assign(n, UndistortPlan{});
    for (int i = 0; i < n; ++i) {
        const UndistortCam& u = st->ucams[i];
        if (!u.map1 || (u.w & 3) || std::getenv("MST_UNDISTORT_GATHER")) continue;
        UndistortPlan& up = st->undist_plans[i];
        const int tx = (u.w + kCompTW - 1) / kCompTW, ty = (u.h + kCompTH - 1) / kCompTH, nt = tx * ty;
        ComposeSeg* d_segs = st->pool.alloc<ComposeSeg>(nt);
        ComposeTile* d_tiles = st->pool.alloc<ComposeTile>(nt);
        uint32_t* d_table = st->pool.alloc<uint32_t>(static_cast<size_t>(nt) * kCompPx);
        launch_undistort_table(u, tx, ty, d_segs, d_tiles, d_table, kCompLdsPx, s);
        check_hip(hipGetLastError(), "k_undistort_table");
        std::vector<ComposeTile> tiles(nt);
        std::vector<ComposeSeg> segs(nt);
        check_hip(hipMemcpyAsync(tiles.data(), d_tiles, nt * sizeof(ComposeTile), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H)");
        check_hip(hipMemcpyAsync(segs.data(), d_segs, nt * sizeof(ComposeSeg), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H)");
        check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(undistort tables)");
        bool ok = true;
        size_t lds_px = 4;
        for (int k = 0; k < nt; ++k) {
            ok &= !(tiles[k].flags & kCompTileGeneral);
            lds_px = std::max(lds_px, static_cast<size_t>(segs[k].groups) * 4 * segs[k].rows);
        }
        if (!ok) continue;
        ComposeArgs& ca = up.args;
        ca = ComposeArgs{};
        ca.mode = kCompModePaste;
        ca.n_cams = 1;
        ca.tiles = d_tiles; ca.segs = d_segs; ca.table = d_table;
        ca.n_tiles = nt;
        ca.out_w = u.w; ca.out_h = u.h; ca.canvas_w = u.w; ca.canvas_h = u.h; ca.tiles_x = tx;
        ca.pad_w = u.w; ca.pad_h = u.h;
        ca.pano_stride = static_cast<uint32_t>(u.w) * 3u;
        up.lds_bytes = std::max<size_t>(lds_px * 4, static_cast<size_t>(kCompPx) * 4);      // (the window, then the tile on its way out)
        up.staged = true;
        {
            std::vector<uint32_t> ids(nt);
            for (int k = 0; k < nt; ++k) ids[k] = static_cast<uint32_t>(k);
            launch_compose_permute(st->pool.upload(ids, s), nt, d_table, nullptr, nullptr, s);      // the staged kernel's lane order
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(undistort permute)");
        }
    }

    st->canvas = result_roi(rois);
    for (int i = 0; i < n; ++i) {
        st->desc[i].corner_x = rois[i].x - st->canvas.x;
        st->desc[i].corner_y = rois[i].y - st->canvas.y;
    }
    st->crop_x = st->crop_y = 0;
    st->out_w = st->canvas.width; st->out_h = st->canvas.height;

    if (cfg_.blend_kind == MST_BLEND_MULTI_BAND) {
        const int nb = multiband_num_bands(cfg_.num_bands, st->canvas);
        if (nb + 1 > kMaxLevels) throw Error{MST_ERR_INVALID_ARG, "too many bands"};
        st->num_bands = nb;
        st->padded = multiband_padded_roi(st->canvas, nb);
        CanvasDesc& cv = st->cv;
        cv.n_cams = n; cv.nlev = nb + 1;
        cv.final_w = st->canvas.width; cv.final_h = st->canvas.height;
        cv.cw[0] = st->padded.width; cv.ch[0] = st->padded.height;
        for (int l = 1; l <= nb; ++l) { cv.cw[l] = (cv.cw[l - 1] + 1) / 2; cv.ch[l] = (cv.ch[l - 1] + 1) / 2; }
        const bool debug = (cfg_.flags & MST_FLAG_DEBUG) != 0;
        std::vector<float*> ws(nb + 1);
        for (int l = 0; l <= nb; ++l) {
            const size_t px = static_cast<size_t>(cv.cw[l]) * cv.ch[l];
            cv.C[l] = nullptr;
            if (debug && l >= 1) {
                cv.C[l] = st->pool.alloc<int16_t>(px * 3);
                check_hip(hipMemsetAsync(cv.C[l], 0, px * 3 * sizeof(int16_t), s), "hipMemsetAsync");
            }
            ws[l] = st->pool.alloc<float>(px);
            cv.WS[l] = ws[l];
        }
        std::vector<std::vector<float*>> wpyr(n, std::vector<float*>(nb + 1));
        std::vector<uint8_t*> mask_b(n, nullptr);
        for (int i = 0; i < n; ++i) {
            CamDesc& d = st->desc[i];
            BandGeometry g = multiband_feed_geometry(st->padded, nb, rois[i]);
            d.top = g.top; d.left = g.left; d.nlev = nb + 1;
            d.lw[0] = g.tl_br.width; d.lh[0] = g.tl_br.height;
            d.ox[0] = g.tl_br.x - st->padded.x; d.oy[0] = g.tl_br.y - st->padded.y;
            for (int l = 1; l <= nb; ++l) {
                d.lw[l] = (d.lw[l - 1] + 1) / 2; d.lh[l] = (d.lh[l - 1] + 1) / 2;
                d.ox[l] = d.ox[l - 1] / 2; d.oy[l] = d.oy[l - 1] / 2;     // blenders.cpp:327-328
            }
            // one contiguous block per camera for the whole Gaussian pyramid (levels back to back): it is the
            // unit that travels in the camera-sharded multi-GPU mode (mst_pyramid_buffer).  Level 0 is packed BGR8 with a
            // row pitch of whole 12-byte groups; the levels above are a dword a pixel (device_types.hpp)
            d.l0_pitch = level0_pitch(d.lw[0]);
            size_t total_dw = 0;
            for (int l = 0; l <= nb; ++l) total_dw += level_dwords(d, l);
            uint32_t* base = st->pool.alloc<uint32_t>(total_dw);
            // tiles the work lists skip are never written: keep them deterministic
            check_hip(hipMemsetAsync(base, 0, total_dw * sizeof(uint32_t), s), "hipMemsetAsync");
            st->pyr_base.push_back(base);
            st->pyr_bytes.push_back(total_dw * sizeof(uint32_t));
            for (int l = 0; l <= nb; ++l) {
                const size_t px = static_cast<size_t>(d.lw[l]) * d.lh[l];
                d.G[l] = base;
                base += level_dwords(d, l);
                wpyr[i][l] = st->pool.alloc<float>(px);
                d.W[l] = wpyr[i][l];
            }
            mask_b[i] = st->pool.alloc<uint8_t>(static_cast<size_t>(d.lw[0]) * d.lh[0] + 16);
            d.mask_b = mask_b[i];
        }
        // Descriptor sets: set 0 owns the pyramids above; frame pipelining (MST_FLAG_OVERLAP_FRAMES) needs a second
        // set of Gaussian pyramids (frame t+1's per-camera kernels write it while frame t's canvas kernel still
        // reads the first), batched launches (mst_config.max_batch) one per frame set of a batch.
        std::vector<CamDesc> sets = st->desc;
        for (int k = 1; k < st->num_sets; ++k)
            for (int i = 0; i < n; ++i) {
                CamDesc d = st->desc[i];
                for (int l = 0; l <= nb; ++l) {
                    const size_t dw = level_dwords(d, l);
                    d.G[l] = st->pool.alloc<uint32_t>(dw);
                    check_hip(hipMemsetAsync(d.G[l], 0, dw * sizeof(uint32_t), s), "hipMemsetAsync");
                }
                sets.push_back(d);
            }
        st->d_desc = st->pool.upload(sets, s);
        check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");     // `sets` is pageable and local
        for (int i = 0; i < n; ++i) {
            const CamDesc& d = st->desc[i];
            launch_border_mask(st->d_desc, i, mask_b[i], d.lw[0], d.lh[0], s);
            const bool w16 = cfg_.weight_type == MST_WEIGHT_16S;
            launch_weight_level0(st->d_desc, i, wpyr[i][0], d.lw[0], d.lh[0], w16, s);
            for (int l = 0; l < nb; ++l) launch_pyrdown_f32(wpyr[i][l], d.lw[l], d.lh[l], wpyr[i][l + 1], w16, s);
        }
        for (int l = 0; l <= nb; ++l) launch_band_weight_sum(st->d_desc, n, l, ws[l], cv.cw[l], cv.ch[l], s);
        check_hip(hipGetLastError(), "weight pyramid kernels");
        clk.mark("alloc + weight pyramid launches");

        // work lists of the fused kernels: classify the weight pyramids on the device (zero / exactly one / other),
        // bring one byte per pixel back through a page-locked buffer, and walk the dependencies on the host
        size_t cls_total = 0;
        for (int i = 0; i < n; ++i)
            for (int l = 0; l <= nb; ++l) cls_total += static_cast<size_t>(st->desc[i].lw[l]) * st->desc[i].lh[l];
        uint8_t* d_cls = st->pool.alloc<uint8_t>(cls_total);
        if (cls_host_bytes_ < cls_total) {
            if (cls_host_) (void)hipHostFree(cls_host_);
            cls_host_ = nullptr;
            check_hip(hipHostMalloc(reinterpret_cast<void**>(&cls_host_), cls_total, hipHostMallocDefault), "hipHostMalloc");
            cls_host_bytes_ = cls_total;
        }
        {
            const float unit = cfg_.weight_type == MST_WEIGHT_16S ? 256.f : 1.f;
            size_t off = 0;
            for (int i = 0; i < n; ++i)
                for (int l = 0; l <= nb; ++l) {
                    const size_t px = static_cast<size_t>(st->desc[i].lw[l]) * st->desc[i].lh[l];
                    launch_weight_class(wpyr[i][l], d_cls + off, px, unit, s);
                    off += px;
                }
        }
        check_hip(hipMemcpyAsync(cls_host_, d_cls, cls_total, hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H weight classes)");
        check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(weights)");
        clk.mark("weight classes D2H");
        std::vector<std::vector<std::vector<uint8_t>>> wh(n, std::vector<std::vector<uint8_t>>(nb + 1));
        {
            size_t off = 0;
            for (int i = 0; i < n; ++i)
                for (int l = 0; l <= nb; ++l) {
                    const size_t px = static_cast<size_t>(st->desc[i].lw[l]) * st->desc[i].lh[l];
                    wh[i][l].assign(cls_host_ + off, cls_host_ + off + px);
                    off += px;
                }
        }
        BlendArgs& ba = st->blend;
        ba = BlendArgs{};
        ba.nlev = nb + 1; ba.n_cams = n;
        for (int l = 0; l <= nb; ++l) { ba.cw[l] = cv.cw[l]; ba.ch[l] = cv.ch[l]; ba.dbg_C[l] = cv.C[l]; }
        ba.final_w = cv.final_w; ba.final_h = cv.final_h;
        ba.alpha = static_cast<float>(1. / 255.);
        ba.weights_16s = cfg_.weight_type == MST_WEIGHT_16S ? 1 : 0;
        strip_columns(cv.cw[0], &st->strip_x0, &st->strip_x1);
        const char* split_env = std::getenv("MST_SPLIT");       // (read per calibration: the parity test switches it)
        const bool split = nb >= 2 && split_env && split_env[0] == '1';
        ba.split = split ? 1 : 0;                                // (the plan clips a strip's level-1 regions only when not split)
        build_mb_plan(st->desc, wh, ba, (cfg_.flags & MST_FLAG_DEBUG) != 0, st->plan, st->strip_x0, st->strip_x1);
        ba.clip_x0 = st->plan.clip_x0; ba.clip_x1 = st->plan.clip_x1;
        clk.mark("work-list plan (host)");
        const MbPlanHost& plan = st->plan;
        ba.tiles_x = plan.blend_tiles_x;
        ba.block_cams = st->pool.upload(plan.block_cams, s);
        ba.block_order = st->pool.upload(plan.block_order, s);
        for (int v = 0; v < 2; ++v) {
            const MbPlanHost::CoarseTiles& ct = plan.coarse[v];
            if (ct.lds_bytes > 64 * 1024) throw Error{MST_ERR_INVALID_ARG, "canvas kernel LDS budget exceeded"};
            if (nb >= 1) {
                st->coarse_dev[v].cams = st->pool.upload(ct.cams, s);
                st->coarse_dev[v].pure = st->pool.upload(ct.pure, s);
                st->coarse_dev[v].order = st->pool.upload(ct.order, s);
                st->coarse_dev[v].order2 = ct.order2.empty() ? nullptr : st->pool.upload(ct.order2, s);
            }
        }
        if (nb >= 1) {
            // the collapsed level-1 canvas k_blend_coarse hands to k_blend_fine, one per buffer set
            st->c1.assign(st->num_sets, nullptr);
            for (int k = 0; k < st->num_sets; ++k)
                st->c1[k] = st->pool.alloc<uint64_t>(static_cast<size_t>(cv.cw[1]) * cv.ch[1]);
            // ... and, opt-in (MST_SPLIT=1), the collapsed level-2 canvas between k_blend_coarse and k_blend_mid.  Measured on
            // C3 (profiles/r04_experiments.md): k_blend_coarse alone 83 us per 8 frame sets; split 49 + 59: the level-1 pass is
            // no faster in k_blend_fine's shape (most of its tiles carry fractional weights and take the float path either
            // way), and the levels above it cost more than the ablation of round 3 suggested.  Off by default.
            st->c2.clear();
            if (split) {
                st->c2.assign(st->num_sets, nullptr);
                for (int k = 0; k < st->num_sets; ++k)
                    st->c2[k] = st->pool.alloc<uint64_t>(static_cast<size_t>(cv.cw[2]) * cv.ch[2]);
            }
        }
        st->warp_list.tiles = st->pool.upload(plan.warp_tiles, s);
        st->warp_list.count = static_cast<int32_t>(plan.warp_tiles.size());
        st->warp_list.first = 0;
        // frame-invariant backward map of every launched tile, in the order kernel A consumes it
        st->warp_table = st->pool.alloc<uint32_t>(mb_warp_table_entries(st->warp_list.count));
        st->warp_blobs = st->pool.alloc<uint32_t>(mb_warp_blob_entries(st->warp_list.count));
        launch_warp_table(st->d_desc, map_kind(), st->warp_list, st->warp_table, st->warp_blobs, s);
        check_hip(hipGetLastError(), "k_warp_table");
        // the staged form of the warp kernel: its own map + gain tables (lane order), the tiles' source windows, and the
        // split of every camera's tiles into those that fit it and those that stay with the gathering kernel
        {
            const int nt = st->warp_list.count;
            const size_t px = mb_warp_tile_px();
            st->warp_table_lds = st->pool.alloc<uint32_t>(std::max<size_t>(nt * px, 4));
            st->warp_gain_lds = st->pool.alloc<float>(std::max<size_t>(nt * px, 4));
            st->warp_wins = st->pool.alloc<WarpWin>(std::max(nt, 1));
            // window size classes of the staged launches, in KB of LDS (one launch per class: a launch's LDS size is uniform, and
            // the blocks per CU follow from it); tiles past the last class stay with the gathering kernel.  C3: 2085 tiles up
            // to 40 KB, 443 (bounding boxes of slanted tiles near the frame corners) up to 72 KB.  With one load per staging
            // trip the 72 KB class was no faster than gathering; with the loads in batches it is: 28.7-29.3 -> 27.5 us.
            static const std::vector<int> classes_all_kb = [] {
                std::vector<int> v;
                const char* e = std::getenv("MST_WARP_LDS_CLASSES");
                for (const char* p = e ? e : "40,72"; *p;) { v.push_back(std::atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
                return v;
            }();
            // (classes the device cannot grant are dropped: their tiles stay with the gathering kernel)
            std::vector<int> classes_kb;
            for (int kb : classes_all_kb)
                if (static_cast<size_t>(kb) * 1024 <= max_lds_bytes_) classes_kb.push_back(kb);
            if (classes_kb.empty()) classes_kb.push_back(static_cast<int>(max_lds_bytes_ / 1024));
            const int budget_px = classes_kb.back() * 256;
            launch_warp_table_lds(st->d_desc, map_kind(), st->warp_list, st->warp_table_lds, st->warp_gain_lds, st->warp_wins,
                                  budget_px, s);
            check_hip(hipGetLastError(), "k_warp_table_lds");
            std::vector<WarpWin> wins(std::max(nt, 1));
            if (nt) check_hip(hipMemcpyAsync(wins.data(), st->warp_wins, nt * sizeof(WarpWin), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H windows)");
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(windows)");
            const int ncls = static_cast<int>(classes_kb.size());
            std::vector<std::vector<WarpWin>> staged(ncls);      // per class, in launch order: window + tile word + index
            std::vector<uint32_t> gathered;
            st->warp_classes.assign(ncls, CalibState::WarpClass{});
            for (int c = 0; c < ncls; ++c) st->warp_classes[c].start.assign(n + 1, 0);
            st->warp_gathered_start.assign(n + 1, 0);
            for (int i = 0; i < n; ++i) {
                for (int c = 0; c < ncls; ++c) st->warp_classes[c].start[i] = static_cast<int>(staged[c].size());
                st->warp_gathered_start[i] = static_cast<int>(gathered.size());
                for (int k = plan.warp_cam_start[i]; k < plan.warp_cam_start[i + 1]; ++k) {
                    if (wins[k].flags & 1u) { gathered.push_back(static_cast<uint32_t>(k)); continue; }
                    const size_t bytes = static_cast<size_t>(wins[k].pitch) * 16 * wins[k].rows;      // (rows at the padded LDS pitch)
                    int c = 0;
                    while (c + 1 < ncls && bytes > static_cast<size_t>(classes_kb[c]) * 1024) ++c;
                    WarpWin ww = wins[k];
                    ww.tile = plan.warp_tiles[k];
                    ww.index = k;
                    staged[c].push_back(ww);
                    st->warp_classes[c].lds_bytes = std::max(st->warp_classes[c].lds_bytes, bytes);
                }
            }
            st->warp_gathered_start[n] = static_cast<int>(gathered.size());
            int n_staged = 0;
            for (int c = 0; c < ncls; ++c) {
                st->warp_classes[c].start[n] = static_cast<int>(staged[c].size());
                n_staged += static_cast<int>(staged[c].size());
                if (staged[c].empty()) staged[c].push_back(WarpWin{});
                st->warp_classes[c].d_wins = st->pool.upload(staged[c], s);
            }
            if (gathered.empty()) gathered.push_back(0u);
            st->d_warp_gathered = st->pool.upload(gathered, s);
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");      // the index lists are pageable and local
            if (std::getenv("MST_COMPOSE_STATS")) {
                std::fprintf(stderr, "[mst_update] warp tiles: %d, %d staged in LDS (%d size classes), %d gathered\n", nt, n_staged, ncls,
                             st->warp_gathered_start[n]);
                int hist[12] = {0};
                for (int k = 0; k < nt; ++k) hist[std::min<size_t>(11, static_cast<size_t>(wins[k].pitch) * 16 * wins[k].rows / 8192)]++;
                std::fprintf(stderr, "[mst_update] warp window bytes, 8 KB bins:");
                for (int b = 0; b < 12; ++b) std::fprintf(stderr, " %d", hist[b]);
                std::fprintf(stderr, "\n");
            }
        }
        st->d_payload_rects = st->pool.upload(plan.payload_rects, s);
        st->chain_lists.clear();
        st->chain_units.clear();
        static const bool chain_only = [] { const char* v = std::getenv("MST_PYR_CHAIN"); return v && v[0] == '1'; }();   // A/B timing
        for (const MbPlanHost::ChainStep& step : plan.chain) {
            TileList tl{};
            tl.tiles = st->pool.upload(step.tiles, s);
            tl.count = static_cast<int32_t>(step.tiles.size());
            tl.first = 0;
            st->chain_lists.push_back(tl);
            CalibState::StreamUnits su;
            if (!chain_only && !step.unit_cam_start[0].empty())
                for (int j = 0; j < step.k; ++j) {
                    std::vector<uint32_t> u = step.units[j];
                    if (u.empty()) u.push_back(0u);
                    su.d[j] = st->pool.upload(u, s);
                }
            st->chain_units.push_back(su);
        }
        check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");      // (the padded unit lists are local)
    } else {
        st->d_desc = st->pool.upload(st->desc, s);
        if (cfg_.blend_kind == MST_BLEND_FEATHER) {
            st->d_feather_ws = st->pool.alloc<float>(static_cast<size_t>(st->canvas.width) * st->canvas.height);
            launch_feather_weight_sum(st->d_desc, n, st->d_feather_ws, st->canvas.width, st->canvas.height, s);
        } else if (cfg_.flags & MST_FLAG_DRIVER_LUT) {
            // UpdateMat / feedSizeRemap crop arithmetic (DRV:153, 702): int op float -> float -> int
            const float up = cfg_.upblack, down = cfg_.downblack, left = cfg_.leftblack, right = cfg_.rightblack;
            st->out_w = static_cast<int>(static_cast<float>(st->canvas.width) - left - right);
            st->out_h = static_cast<int>(static_cast<float>(st->canvas.height) * (1 - up - down));
            st->crop_y = static_cast<int>(static_cast<float>(st->out_h) / (1 - up - down) * up);
            st->crop_x = static_cast<int>(left);
            if (st->out_w <= 0 || st->out_h <= 0 || st->crop_x < 0 || st->crop_y < 0 ||
                st->crop_x + st->out_w > st->canvas.width || st->crop_y + st->out_h > st->canvas.height)
                throw Error{MST_ERR_INVALID_ARG, "crop margins leave no output"};
        }
        build_compose_plan(*st, s);
        clk.mark("compose tables");
    }
    check_hip(hipGetLastError(), "calibration kernels");
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(calibration)");
    clk.mark("tables, lists, final sync");

    std::shared_ptr<CalibState> old;
    {
        std::lock_guard<std::mutex> lk(mu_);
        st->generation = ++next_generation_;
        old = std::move(current_);
        current_ = std::move(st);     // takes effect at the next compose / acquire (Upflag / UpdateMat)
    }
    old.reset();                      // if nobody else holds it: torn down here, on the calibration thread
}


// Column strip [x0, x1) of this engine inside a canvas `width` columns wide (mst_config.strip_index / strip_count):
// equal shares, boundaries on multiples of 256 columns (a level-1 canvas tile), the last strip takes the rest.
void Engine::strip_columns(int width, int* x0, int* x1, int index) const
{
    *x0 = 0; *x1 = width;
    const int g = cfg_.strip_count, k = index < 0 ? cfg_.strip_index : index;
    if (g <= 1) return;
    auto bound = [&](int i) {
        if (i >= g) return width;
        const long b = (static_cast<long>(i) * width / g + 255) / 256 * 256;
        return static_cast<int>(std::min<long>(b, width));
    };
    *x0 = bound(k);
    *x1 = std::max(*x0, bound(k + 1));
}

// The lens-undistort pre-pass of `count` frame sets: src -> dst for every camera of `camera_mask` that has maps.  Cameras
// with a staged plan and raw frames on the 4-byte grid take the staged kernel (one launch per camera over all sets), the
// others the gathering k_remap_fixed (one launch for all of them).
void Engine::launch_undistort(const CalibState& st, const FrameBatch& src, const FrameBatch& dst, int count, uint32_t camera_mask, hipStream_t s)
{
    FrameBatch rest = src;
    bool any_rest = false;
    for (int i = 0; i < st.n; ++i) {
        if (!st.ucams[i].map1 || !(camera_mask >> i & 1u)) { for (int f = 0; f < count; ++f) rest.f[f].ptr[i] = nullptr; continue; }
        const UndistortPlan& up = st.undist_plans[i];
        bool staged = up.staged;
        for (int f = 0; f < count && staged; ++f)
            staged &= ((reinterpret_cast<uintptr_t>(src.f[f].ptr[i]) | src.f[f].stride[i]) & 3u) == 0;
        if (!staged) { any_rest = true; continue; }
        ComposeArgs ca = up.args;
        ca.n_sets = count;
        FrameBatch one{};
        for (int f = 0; f < count; ++f) {
            ca.pano[f] = const_cast<uint8_t*>(dst.f[f].ptr[i]);
            one.f[f].ptr[0] = src.f[f].ptr[i];
            one.f[f].stride[0] = src.f[f].stride[i];
            rest.f[f].ptr[i] = nullptr;
        }
        launch_compose(ca, one, up.lds_bytes, false, s);
    }
    if (any_rest) launch_remap_fixed(st.d_ucams, st.n, st.undist_max_w, st.undist_max_h, rest, dst, count, s);
}

// The warp kernel (A) over the tiles of cameras [cam0, cam1): the staged form for the tiles that fit it, the gathering
// kernel for the rest -- and for everything when a frame of the call is off the 4-byte grid (the staged form reads whole
// dwords) or MST_WARP_GATHER is set (A/B timing).
void Engine::launch_warp(const CalibState& st, const CamDesc* desc, const FrameBatch& fb, int count, int cam0, int cam1, hipStream_t s,
                         StageTimer* timer)
{
    auto mark = [&](const char* name) { if (timer && name) timer->mark(name, s); };
    static const bool force_gather = [] { const char* v = std::getenv("MST_WARP_GATHER"); return v && v[0] == '1'; }();
    // (one or two frame sets per launch: the gathering kernel's smaller blocks fill the chip better -- 9.5 k against 8.8 k
    // panoramas/s one at a time on C3; from three sets on the staged form wins)
    static const int staged_min = [] { const char* v = std::getenv("MST_WARP_STAGED_MIN"); return v ? std::atoi(v) : 3; }();      // (A/B runs)
    bool aligned = !force_gather && count >= staged_min;
    for (int f = 0; f < count && aligned; ++f)
        for (int i = cam0; i < cam1; ++i)
            if (fb.f[f].ptr[i]) aligned &= ((reinterpret_cast<uintptr_t>(fb.f[f].ptr[i]) | fb.f[f].stride[i]) & 3u) == 0;
    const MbPlanHost& plan = st.plan;
    if (!aligned) {
        TileList tl{};
        tl.tiles = st.warp_list.tiles + plan.warp_cam_start[cam0];
        tl.count = plan.warp_cam_start[cam1] - plan.warp_cam_start[cam0];
        tl.first = plan.warp_cam_start[cam0];
        mark(launch_warp_pyr(desc, st.n, map_kind(), tl, fb, count, st.warp_table, st.warp_blobs, s));
        return;
    }
    TileList tg{};
    tg.tiles = st.warp_list.tiles;
    tg.first = 0;
    tg.index = st.d_warp_gathered + st.warp_gathered_start[cam0];
    tg.count = st.warp_gathered_start[cam1] - st.warp_gathered_start[cam0];
    mark(launch_warp_pyr(desc, st.n, map_kind(), tg, fb, count, st.warp_table, st.warp_blobs, s));     // (the short list first)
    for (int c = static_cast<int>(st.warp_classes.size()) - 1; c >= 0; --c) {       // (the large-window classes first: fewer, longer blocks)
        const CalibState::WarpClass& wc = st.warp_classes[c];
        mark(launch_warp_pyr_lds(desc, st.n, wc.start[cam1] - wc.start[cam0], fb, count, st.warp_table_lds, st.warp_gain_lds,
                                 wc.d_wins + wc.start[cam0], wc.lds_bytes, s));
    }
}

// Per-calibration tables of the single-pass composites (kernels_compose.hip): which cameras reach each 128 x 32 output
// tile (classified on the device, one dword per tile back to the host), one segment per (tile, camera), the
// segments' map / gain / weight tables, the static mask plane.  This is what the reference's driver builds as its
// per-pixel LUT in prepareSize / feedSize (DRV:94-148), extended to the feather blend.
void Engine::build_compose_plan(CalibState& st, hipStream_t s)
{
    ComposeArgs& ca = st.compose;
    ca = ComposeArgs{};
    ca.cams = st.d_desc;
    ca.n_cams = st.n;
    ca.mode = cfg_.blend_kind == MST_BLEND_FEATHER ? kCompModeFeather
              : ((cfg_.flags & MST_FLAG_DRIVER_LUT) ? kCompModeLut : kCompModePaste);
    ca.out_w = st.out_w; ca.out_h = st.out_h; ca.crop_x = st.crop_x; ca.crop_y = st.crop_y;
    ca.canvas_w = st.canvas.width; ca.canvas_h = st.canvas.height;
    ca.feather_ws = st.d_feather_ws;
    const int tiles_x = (st.out_w + kCompTW - 1) / kCompTW, tiles_y = (st.out_h + kCompTH - 1) / kCompTH;
    if (tiles_x > 0xFFFF || tiles_y > 0xFFFF) throw Error{MST_ERR_INVALID_ARG, "output too large for the tile tables"};
    ca.tiles_x = tiles_x;
    const int n_canvas_tiles = tiles_x * tiles_y;
    strip_columns(st.out_w, &st.strip_x0, &st.strip_x1);
    uint32_t* d_bits = st.pool.alloc<uint32_t>(n_canvas_tiles);
    launch_compose_classify(ca, n_canvas_tiles, d_bits, s);
    check_hip(hipGetLastError(), "k_compose_classify");
    std::vector<uint32_t> bits(n_canvas_tiles);
    check_hip(hipMemcpyAsync(bits.data(), d_bits, bits.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H tile cameras)");
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(tile cameras)");
    // launch order: tiles with the most segments first (the longest blocks start earliest)
    std::vector<int> order;
    for (int i = 0; i < n_canvas_tiles; ++i) {          // (canvas split over GPUs: the tiles of this engine's strip)
        const int x = (i % tiles_x) * kCompTW;
        if (x < st.strip_x1 && x + kCompTW > st.strip_x0) order.push_back(i);
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return __builtin_popcount(bits[a] & 0x1FFFFu) > __builtin_popcount(bits[b] & 0x1FFFFu); });
    const int n_tiles = static_cast<int>(order.size());
    std::vector<ComposeTile> tiles(n_tiles);
    std::vector<ComposeSeg> segs;
    std::vector<uint32_t> seg_tile;
    for (int k = 0; k < n_tiles; ++k) {
        const int ti = order[k];
        ComposeTile& t = tiles[k];
        t.seg_first = static_cast<uint32_t>(segs.size());
        t.seg_count = 0;
        t.flags = 0;
        t.tx_ty = static_cast<uint32_t>(ti % tiles_x) | static_cast<uint32_t>(ti / tiles_x) << 16;
        for (int c = 0; c <= 16; ++c)          // (bit 16: the LUT's fill pixels, a segment of camera 0 of their own)
            if ((c < st.n || c == 16) && (bits[ti] >> c & 1u)) {
                segs.push_back(ComposeSeg{c == 16 ? 0 : c, 0, 0, c == 16 ? kCompSegFill : 0u, 0, 0, 0, 0});
                seg_tile.push_back(static_cast<uint32_t>(k));
                ++t.seg_count;
            }
    }
    const int n_segs = static_cast<int>(segs.size());
    if (static_cast<size_t>(n_segs) * kCompPx > 0xFFFFFFFFull) throw Error{MST_ERR_INVALID_ARG, "too many tile segments"};
    ComposeTile* d_tiles = st.pool.upload(tiles.data(), tiles.size(), s);
    if (segs.empty()) { segs.push_back(ComposeSeg{}); seg_tile.push_back(0u); }     // (n_segs stays 0)
    ComposeSeg* d_segs = st.pool.upload(segs, s);
    uint32_t* d_seg_tile = st.pool.upload(seg_tile, s);
    const size_t entries = std::max<size_t>(static_cast<size_t>(n_segs) * kCompPx, 4);
    uint32_t* d_table = st.pool.alloc<uint32_t>(entries);
    float* d_gain = cfg_.expos_kind == MST_EXPOS_NO ? nullptr : st.pool.alloc<float>(entries);      // (no compensation: every gain is 1)
    float* d_weight = ca.mode == kCompModeFeather ? st.pool.alloc<float>(entries) : nullptr;
    ca.tiles = d_tiles; ca.segs = d_segs; ca.table = d_table; ca.gain = d_gain; ca.weight = d_weight;
    ca.n_tiles = n_tiles;
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");     // tiles / segs / seg_tile are pageable and local
    launch_compose_table(ca, map_kind(), n_segs, d_segs, d_tiles, d_seg_tile, d_table, d_gain, d_weight, s);
    check_hip(hipGetLastError(), "k_compose_table");
    uint8_t* d_plane = st.pool.alloc<uint8_t>(static_cast<size_t>(st.out_w) * st.out_h + 16);
    launch_compose_mask(ca, d_plane, s);
    ca.mask_plane = d_plane;
    check_hip(hipGetLastError(), "compose calibration kernels");
    // Two launch lists (each in the cost order above): tiles whose segments all fit the LDS-staged form, and the rest
    // (gather from memory; per-pixel general path where flagged).  The table kernel decided per segment.
    check_hip(hipMemcpyAsync(tiles.data(), d_tiles, tiles.size() * sizeof(ComposeTile), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H tiles)");
    if (n_segs) check_hip(hipMemcpyAsync(segs.data(), d_segs, segs.size() * sizeof(ComposeSeg), hipMemcpyDeviceToHost, s), "hipMemcpyAsync(D2H segments)");
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(compose tables)");
    std::vector<ComposeTile> staged, direct, general;
    size_t lds_px = 0;
    // launch order by cost: segments, those with real blend weights (feather bands: converts, products, a division per
    // channel) counted twice
    auto tile_cost = [&](const ComposeTile& t) {
        int c = 0;
        for (uint32_t k = 0; k < t.seg_count; ++k)
            c += (ca.mode == kCompModeFeather && !(segs[t.seg_first + k].flags & kCompSegUnit)) ? 2 : 1;
        return c;
    };
    std::stable_sort(tiles.begin(), tiles.end(), [&](const ComposeTile& x, const ComposeTile& y) { return tile_cost(x) > tile_cost(y); });
    // MST_COMPOSE_ORDER: 0 cost order as sorted above, 1 heavy tiles spread evenly over it, 2 raster order,
    // 3 raster order with the heavy tiles of its last eighth moved to the front (no long block in the launch's tail)
    // Measured (C2 / C3 feather, C3 / C5 LUT, frac): cost order 0.26 / 0.24 / 0.37 / 0.45, spread 0.31 / 0.28 / 0.45 / 0.50,
    // raster 0.34 / 0.31 / 0.44 / 0.50, raster + tail fix 0.35 / 0.31 / 0.43 / 0.49.  The long tiles (several segments,
    // real feather weights) are chains of memory round trips: a CU full of them idles, so they must not lead the list.
    static const int order_env = [] { const char* v = std::getenv("MST_COMPOSE_ORDER"); return v ? std::atoi(v) : -1; }();
    const int order_kind = order_env >= 0 ? order_env : (ca.mode == kCompModeFeather ? 3 : 1);
    auto raster_key = [](const ComposeTile& t) { return t.tx_ty; };           // ty << 16 | tx
    if (order_kind == 1) {
        size_t n_heavy = 0;
        while (n_heavy < tiles.size() && tile_cost(tiles[n_heavy]) > 1) ++n_heavy;
        const size_t n_light = tiles.size() - n_heavy;
        if (n_heavy && n_light) {
            std::vector<ComposeTile> mixed;
            mixed.reserve(tiles.size());
            size_t h = 0, l = 0;
            for (size_t pos = 0; pos < tiles.size(); ++pos) {
                // heavy tile k sits at position k * size / n_heavy
                const bool want_heavy = h < n_heavy && (pos >= h * tiles.size() / n_heavy || l >= n_light);
                mixed.push_back(want_heavy ? tiles[h++] : tiles[n_heavy + l++]);
            }
            tiles.swap(mixed);
        }
    } else if (order_kind >= 2) {
        std::stable_sort(tiles.begin(), tiles.end(), [&](const ComposeTile& x, const ComposeTile& y) { return raster_key(x) < raster_key(y); });
        if (order_kind == 3) {
            const size_t tail = tiles.size() - tiles.size() / 8;
            std::vector<ComposeTile> front, rest;
            for (size_t i = 0; i < tiles.size(); ++i) ((i >= tail && tile_cost(tiles[i]) > 1) ? front : rest).push_back(tiles[i]);
            front.insert(front.end(), rest.begin(), rest.end());
            tiles.swap(front);
        }
    }
    std::vector<ComposeTile> staged_lds_only;
    static const int only_env = [] { const char* v = std::getenv("MST_COMPOSE_ONLY"); return v ? std::atoi(v) : 0; }();   // diagnostic
    for (const ComposeTile& t : tiles) {
        if (only_env && t.seg_count) {
            const bool pure = t.seg_count == 1 && (segs[t.seg_first].flags & kCompSegUnit);
            if ((only_env == 1) != pure) continue;
        }
        if ((t.flags & kCompTileGeneral) || ((t.flags & kCompTileDirect) && (t.flags & kCompTileNoGather))) { general.push_back(t); continue; }
        if (t.flags & kCompTileDirect) { direct.push_back(t); continue; }
        ((t.flags & kCompTileNoGather) ? staged_lds_only : staged).push_back(t);
        for (uint32_t k = 0; k < t.seg_count; ++k) {
            const ComposeSeg& sg = segs[t.seg_first + k];
            lds_px = std::max(lds_px, static_cast<size_t>(sg.groups) * 4 * sg.rows);
        }
    }
    // (one list: the tiles that may also gather first, then those only the staged form can read -- with frames off the
    // 4-byte grid the first part gathers and the second takes the general path)
    st.compose_staged_gatherable = static_cast<int>(staged.size());
    staged.insert(staged.end(), staged_lds_only.begin(), staged_lds_only.end());
    st.compose_staged = static_cast<int>(staged.size());
    st.compose_direct = static_cast<int>(direct.size());
    st.compose_general = static_cast<int>(general.size());
    st.compose_lds_bytes = std::max<size_t>(lds_px * 4, static_cast<size_t>(kCompPx) * 4);      // (the window, then the tile on its way out)
    {
        // the staged tiles' table rows into the staged kernel's lane order (k_compose_permute); such tiles can no longer be
        // read by the gathering kernel: with frames off the 4-byte grid they take the general path
        std::vector<uint32_t> ids;
        for (const ComposeTile& t : staged)
            for (uint32_t k = 0; k < t.seg_count; ++k) ids.push_back(t.seg_first + k);
        if (!ids.empty()) {
            launch_compose_permute(st.pool.upload(ids, s), static_cast<int>(ids.size()), d_table, d_gain, d_weight, s);
            check_hip(hipGetLastError(), "k_compose_permute");
            check_hip(hipStreamSynchronize(s), "hipStreamSynchronize(compose permute)");      // `ids` is local
        }
    }
    if (staged.empty()) staged.push_back(ComposeTile{});          // (never launched: the counts above are 0)
    if (direct.empty()) direct.push_back(ComposeTile{});
    if (general.empty()) general.push_back(ComposeTile{});
    st.d_compose_staged = st.pool.upload(staged, s);
    st.d_compose_direct = st.pool.upload(direct, s);
    st.d_compose_general = st.pool.upload(general, s);
    if (std::getenv("MST_COMPOSE_STATS"))
        std::fprintf(stderr, "[mst_update] compose plan: %d tiles, %d segments, %d staged in LDS (%zu bytes), %d gathered, %d general\n",
                     n_canvas_tiles, n_segs, st.compose_staged, st.compose_lds_bytes, st.compose_direct, st.compose_general);
    if (std::getenv("MST_COMPOSE_STATS")) {
        int one_unit = 0, one_w = 0, multi = 0, multi_unit_segs = 0, multi_segs = 0;
        size_t win_px = 0;
        for (const ComposeTile& t : tiles) {
            if (!t.seg_count) continue;
            const bool unit0 = segs[t.seg_first].flags & kCompSegUnit;
            if (t.seg_count == 1) (unit0 ? one_unit : one_w)++;
            else { ++multi; for (uint32_t k = 0; k < t.seg_count; ++k) { ++multi_segs; multi_unit_segs += (segs[t.seg_first + k].flags & kCompSegUnit) ? 1 : 0; } }
            for (uint32_t k = 0; k < t.seg_count; ++k) win_px += static_cast<size_t>(segs[t.seg_first + k].groups) * 4 * segs[t.seg_first + k].rows;
        }
        std::fprintf(stderr, "[mst_update] compose tiles: %d one unit segment, %d one weighted segment, %d with several (%d segments, %d of them unit); "
                             "mean window %zu px\n", one_unit, one_w, multi, multi_segs, multi_unit_segs, n_segs ? win_px / n_segs : 0);
    }
    check_hip(hipStreamSynchronize(s), "hipStreamSynchronize");     // the lists are pageable and local
    st.compose_segments = n_segs;
}

void Engine::compose_device(const void* const* d_frames, const size_t* strides, void* d_pano, size_t pano_stride,
                            void* d_mask, size_t mask_stride, hipStream_t stream, StageTimer* timer)
{
    void* panos[1] = {d_pano};
    void* masks[1] = {d_mask};
    compose_batch(1, d_frames, strides, panos, pano_stride, d_mask ? masks : nullptr, mask_stride, stream, timer);
}

void Engine::drop_graphs()
{
    for (GraphEntry& g : graphs_)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    graphs_.clear();
    for (RetiredGraph& r : retired_graphs_) {
        if (r.done) { (void)hipEventSynchronize(r.done); (void)hipEventDestroy(r.done); }
        if (r.exec) (void)hipGraphExecDestroy(r.exec);
    }
    retired_graphs_.clear();
}

// destroy the parked execs whose last launch has completed
void Engine::reap_graphs()
{
    for (size_t i = 0; i < retired_graphs_.size();) {
        RetiredGraph& r = retired_graphs_[i];
        if (r.done && hipEventQuery(r.done) != hipSuccess) { ++i; continue; }
        if (r.done) (void)hipEventDestroy(r.done);
        else (void)hipDeviceSynchronize();           // (no event could be made: the slow, safe way)
        (void)hipGraphExecDestroy(r.exec);
        retired_graphs_.erase(retired_graphs_.begin() + i);
    }
}

void Engine::compose_batch(int count, const void* const* d_frames, const size_t* strides, void* const* d_panos,
                           size_t pano_stride, void* const* d_masks, size_t mask_stride, hipStream_t stream,
                           StageTimer* timer, hipEvent_t frames_ready, hipEvent_t outputs_free)
{
    std::shared_ptr<CalibState> st = state();
    const bool replay = (cfg_.flags & MST_FLAG_GRAPH_REPLAY) && !(cfg_.flags & MST_FLAG_OVERLAP_FRAMES) && !timer && st &&
                        d_frames && strides && d_panos && count >= 1 && count <= kMaxBatch;
    if (!replay) {
        compose_batch_on(std::move(st), count, d_frames, strides, d_panos, pano_stride, d_masks, mask_stride, stream, timer,
                         frames_ready, outputs_free);
        return;
    }
    // the launch sequence depends on the calibration and on every pointer / stride of the call: that is the key
    select_device(cfg_.device);
    hipStream_t s = stream ? stream : stream_;
    // (the caller's events stay outside the captured sequence: they differ from call to call)
    if (frames_ready) check_hip(hipStreamWaitEvent(s, frames_ready, 0), "hipStreamWaitEvent(frames_ready)");
    if (outputs_free) check_hip(hipStreamWaitEvent(s, outputs_free, 0), "hipStreamWaitEvent(outputs_free)");
    std::vector<uint64_t> key;
    key.reserve(6 + 3 * static_cast<size_t>(count) * (st->n + 1));
    key.push_back(st->generation); key.push_back(static_cast<uint64_t>(count));
    key.push_back(pano_stride); key.push_back(mask_stride); key.push_back(d_masks != nullptr);
    for (int i = 0; i < count * st->n; ++i) { key.push_back(reinterpret_cast<uint64_t>(d_frames[i])); key.push_back(strides[i]); }
    for (int f = 0; f < count; ++f) {
        key.push_back(reinterpret_cast<uint64_t>(d_panos[f]));
        key.push_back(d_masks ? reinterpret_cast<uint64_t>(d_masks[f]) : 0);
    }
    reap_graphs();
    for (size_t i = 0; i < graphs_.size(); ++i)
        if (graphs_[i].key == key) {
            check_hip(hipGraphLaunch(graphs_[i].exec, s), "hipGraphLaunch");
            graphs_[i].last_stream = s;
            if (i + 1 != graphs_.size()) std::rotate(graphs_.begin() + i, graphs_.begin() + i + 1, graphs_.end());   // LRU: a hit moves to the back
            set_in_flight(std::move(st));
            return;
        }
    ++graph_misses_;
    // An exec that leaves the cache may still be queued or running (its last launch was enqueued microseconds ago):
    // it is parked behind an event recorded on the stream of ITS last launch (which need not be this call's stream) and
    // destroyed once that event has completed.
    auto retire = [&](size_t i) {
        RetiredGraph r;
        r.exec = graphs_[i].exec;
        hipStream_t on = graphs_[i].last_stream ? graphs_[i].last_stream : s;
        if (hipEventCreateWithFlags(&r.done, hipEventDisableTiming) == hipSuccess && hipEventRecord(r.done, on) == hipSuccess) {
        } else {
            if (r.done) (void)hipEventDestroy(r.done);
            r.done = nullptr;            // (reap_graphs then synchronises the device before destroying the exec)
        }
        retired_graphs_.push_back(r);
        graphs_.erase(graphs_.begin() + i);
    };
    // sequences of older calibrations can never be asked for again
    for (size_t i = 0; i < graphs_.size();)
        if (graphs_[i].key[0] != st->generation) retire(i);
        else ++i;
    if (graphs_.size() >= kGraphCacheEntries) {
        // (a ring of more buffer sets than the cache holds re-captures on every call: say so once)
        if (!graph_thrash_warned_ && graph_misses_ > 4 * kGraphCacheEntries) {
            graph_thrash_warned_ = true;
            std::fprintf(stderr, "[mi355-stitch] MST_FLAG_GRAPH_REPLAY: %llu captures so far -- more than %zu distinct buffer sets per calibration defeat the graph cache\n",
                         static_cast<unsigned long long>(graph_misses_), kGraphCacheEntries);
        }
        retire(0);                   // least recently used
    }
    hipGraph_t graph = nullptr;
    check_hip(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
    try {
        compose_batch_on(st, count, d_frames, strides, d_panos, pano_stride, d_masks, mask_stride, s, nullptr);
    } catch (...) {
        (void)hipStreamEndCapture(s, &graph);
        if (graph) (void)hipGraphDestroy(graph);
        throw;
    }
    check_hip(hipStreamEndCapture(s, &graph), "hipStreamEndCapture");
    GraphEntry e;
    e.key = std::move(key);
    const hipError_t ie = hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    check_hip(ie, "hipGraphInstantiate");
    check_hip(hipGraphLaunch(e.exec, s), "hipGraphLaunch");
    e.last_stream = s;
    graphs_.push_back(std::move(e));
}

// One calibration state for the whole call: callers that size buffers first (compose_host) pass the snapshot they
// sized them from, so a concurrent mst_update cannot slip a different canvas in between.
// Deep form of MST_FLAG_OVERLAP_FRAMES: calls on different slots run on different streams.  Two of them that write overlapping
// bytes -- the same panorama or mask buffer, or sub-views of one allocation that share rows -- must not overlap in time: the
// later one waits for the earlier one's completion event, which restores the order of the calls for those bytes.  A buffer
// counts as the byte interval from its first row to the end of its last row (rows that interleave without sharing a byte are
// ordered too: conservative, never wrong).
void Engine::order_after_writers(int slot, hipStream_t s, const std::vector<ByteRange>& outs)
{
    for (int k = 0; k < kMaxOverlap; ++k) {
        if (k == slot || !canvas_pending_[k]) continue;
        bool shared = false;
        for (const ByteRange& p : slot_outputs_[k])
            for (const ByteRange& q : outs) shared = shared || (p.lo < q.hi && q.lo < p.hi);
        if (shared) check_hip(hipStreamWaitEvent(s, ev_canvas_done_[k], 0), "hipStreamWaitEvent");
    }
    slot_outputs_[slot] = outs;
}

void Engine::compose_batch_on(std::shared_ptr<CalibState> st, int count, const void* const* d_frames, const size_t* strides,
                              void* const* d_panos, size_t pano_stride, void* const* d_masks, size_t mask_stride,
                              hipStream_t stream, StageTimer* timer, hipEvent_t frames_ready, hipEvent_t outputs_free)
{
    select_device(cfg_.device);
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "mst_compose before a successful mst_update"};
    if (!d_frames || !strides || !d_panos) throw Error{MST_ERR_INVALID_ARG, "null frame or panorama table"};
    if (count < 1 || count > kMaxBatch) throw Error{MST_ERR_INVALID_ARG, "batch size out of range"};
    const bool nv12 = !st->nv12_frames.empty();
    if ((st->blend_kind == MST_BLEND_MULTI_BAND || st->any_undistort || nv12) && count > 1 && count > st->sets_per_slot)
        throw Error{MST_ERR_INVALID_ARG, "batch larger than mst_config.max_batch"};
    hipStream_t s = stream ? stream : stream_;
    FrameBatch fb{};
    for (int f = 0; f < count; ++f)
        for (int i = 0; i < st->n; ++i) {
            const void* p = d_frames[f * st->n + i];
            const size_t stride = strides[f * st->n + i];
            if (!p) throw Error{MST_ERR_INVALID_ARG, "null frame pointer"};
            if (stride < static_cast<size_t>(st->cams[i].src_w) * (nv12 ? 1 : 3)) throw Error{MST_ERR_INVALID_ARG, "frame stride too small"};
            if (stride >= (size_t{1} << 24)) throw Error{MST_ERR_INVALID_ARG, "frame stride must be below 16 MiB"};
            if (stride * static_cast<size_t>(st->cams[i].src_h) * (nv12 ? 3 : 2) / 2 > 0xFFFFFFFFull)
                throw Error{MST_ERR_INVALID_ARG, "frames of 4 GiB and more are not supported (32-bit byte offsets)"};
            fb.f[f].ptr[i] = static_cast<const uint8_t*>(p);
            fb.f[f].stride[i] = static_cast<uint32_t>(stride);
        }
    // (an engine that composes one column strip only needs rows that reach from the strip's first to its last column)
    const int strip_w = std::min(st->strip_x1, st->out_w) - std::min(st->strip_x0, st->out_w);
    if (pano_stride < static_cast<size_t>(cfg_.strip_count > 1 ? strip_w : st->out_w) * 3) throw Error{MST_ERR_INVALID_ARG, "panorama stride too small"};
    if (pano_stride >= (size_t{1} << 24) || mask_stride >= (size_t{1} << 24))
        throw Error{MST_ERR_INVALID_ARG, "panorama and mask strides must be below 16 MiB"};
    if (d_masks && mask_stride < static_cast<size_t>(st->out_w)) throw Error{MST_ERR_INVALID_ARG, "mask stride too small"};
    if (pano_stride * static_cast<size_t>(st->out_h) > 0xFFFFFFFFull)
        throw Error{MST_ERR_INVALID_ARG, "panoramas of 4 GiB and more are not supported (32-bit byte offsets)"};
    for (int f = 0; f < count; ++f)
        if (!d_panos[f]) throw Error{MST_ERR_INVALID_ARG, "null panorama pointer"};

    // the bytes this call writes (deep form: what order_after_writers compares between calls in flight)
    auto output_ranges = [&] {
        std::vector<ByteRange> outs;
        const size_t rows = static_cast<size_t>(std::max(st->out_h, 1));
        // (a strip engine's panorama pointer is canvas column 0; it writes its own columns only)
        const size_t x0 = cfg_.strip_count > 1 ? static_cast<size_t>(std::min(st->strip_x0, st->out_w)) : 0;
        const size_t x1 = cfg_.strip_count > 1 ? static_cast<size_t>(std::min(st->strip_x1, st->out_w)) : static_cast<size_t>(st->out_w);
        const size_t mask_row = static_cast<size_t>(st->out_w);
        for (int f = 0; f < count; ++f) {
            const uintptr_t p = reinterpret_cast<uintptr_t>(d_panos[f]);
            outs.push_back(ByteRange{p + x0 * 3, p + (rows - 1) * pano_stride + x1 * 3});
            if (d_masks && d_masks[f]) {
                const uintptr_t m = reinterpret_cast<uintptr_t>(d_masks[f]);
                outs.push_back(ByteRange{m, m + (rows - 1) * mask_stride + mask_row});
            }
        }
        return outs;
    };
    // the caller's events (mst_compose_device_batch_ex): `on_frames` = the stream of the call's first frame-reading kernel,
    // `on_outputs` = the stream its panorama-writing kernels run on (the same stream except in the two-stream form)
    auto wait_for_caller = [&](hipStream_t on_frames, hipStream_t on_outputs) {
        if (frames_ready) check_hip(hipStreamWaitEvent(on_frames, frames_ready, 0), "hipStreamWaitEvent(frames_ready)");
        if (outputs_free) check_hip(hipStreamWaitEvent(on_outputs, outputs_free, 0), "hipStreamWaitEvent(outputs_free)");
    };
    auto mark = [&](const char* name) { if (timer && name) timer->mark(name, s); };      // (nullptr: the launcher had nothing to launch)
    if (timer) timer->begin(s);
    // lens-undistort pre-pass: remap the caller's frames into the engine's buffers of frame sets first_set.. and
    // point the frame table at them (cameras without maps keep the caller's frames)
    auto undistort = [&](int first_set, hipStream_t on) {
        if (nv12) {              // NV12 ingest: the callers' frames into the engine's BGR frames of these sets, first of all
            FrameBatch cb{};
            for (int f = 0; f < count; ++f)
                for (int i = 0; i < st->n; ++i) {
                    cb.f[f].ptr[i] = st->nv12_frames[static_cast<size_t>(first_set + f) * st->n + i];
                    cb.f[f].stride[i] = static_cast<uint32_t>(st->cams[i].src_w) * 3u;
                }
            launch_nv12_to_bgr(fb, cb, st->d_frame_sizes, st->n, st->frame_max_w, st->frame_max_h, count, on);
            if (timer) timer->mark("k_nv12_to_bgr", on);
            fb = cb;
        }
        if (!st->any_undistort) return;
        FrameBatch ub = fb;
        for (int f = 0; f < count; ++f)
            for (int i = 0; i < st->n; ++i)
                if (uint8_t* buf = st->undist_frames[static_cast<size_t>(first_set + f) * st->n + i]) {
                    ub.f[f].ptr[i] = buf;
                    ub.f[f].stride[i] = static_cast<uint32_t>(st->cams[i].src_w) * 3u;
                }
        launch_undistort(*st, fb, ub, count, ~0u, on);
        if (timer) timer->mark("k_remap_fixed", on);
        fb = ub;
    };
    if (st->blend_kind == MST_BLEND_MULTI_BAND) {
        // Frame pipelining: the per-camera kernels (A, B) of this call (one frame set or a batch) go to front_stream_
        // and only wait for the canvas kernel that last read this pyramid set (two frames ago); the canvas kernel
        // (C) stays on the caller's stream behind them.  Frame t+1's A/B therefore overlap frame t's C.
        const bool overlap = (cfg_.flags & MST_FLAG_OVERLAP_FRAMES) && st->num_sets >= 2 * st->sets_per_slot &&
                             timer == nullptr && !in_host_compose_;
        // deep form (MST_OVERLAP_DEPTH > 2): the whole call on the slot's own stream; the caller's stream only waits for it
        const int slots = st->num_sets / st->sets_per_slot;
        const bool deep = overlap && overlap_depth_ > 2 && slots > 2;
        hipStream_t caller = s;
        const int slot = overlap ? static_cast<int>(frame_counter_++ % (deep ? slots : 2)) : 0;
        if (deep) {
            s = call_streams_[slot];
            order_after_writers(slot, s, output_ranges());
        }
        // (every exit of the deep form, the throwing ones included, leaves the caller's stream behind the slot's: whatever was
        // queued on the slot's stream still completes before anything the caller queues next)
        struct Join {
            Engine* e; bool on; int slot; hipStream_t from, to; bool done = false;
            void now() {
                if (!on || done) return;
                done = true;
                if (hipEventRecord(e->ev_canvas_done_[slot], from) == hipSuccess) {
                    e->canvas_pending_[slot] = true;
                    (void)hipStreamWaitEvent(to, e->ev_canvas_done_[slot], 0);
                }
            }
            ~Join() { now(); }
        } join{this, deep, slot, s, caller};
        const int first_set = slot * st->sets_per_slot;           // the slot's run of descriptor / buffer sets
        const CamDesc* desc = st->d_desc + static_cast<size_t>(first_set) * st->n;
        hipStream_t sf = deep ? s : overlap ? front_stream_ : s;
        if (overlap && !deep && canvas_pending_[slot])
            check_hip(hipStreamWaitEvent(sf, ev_canvas_done_[slot], 0), "hipStreamWaitEvent");
        if (!overlap)          // a batch (or a timed frame) re-uses the sets an earlier pipelined frame may still read
            for (int k = 0; k < kMaxOverlap; ++k)
                if (canvas_pending_[k]) {
                    check_hip(hipStreamWaitEvent(s, ev_canvas_done_[k], 0), "hipStreamWaitEvent");
                    canvas_pending_[k] = false;
                }
        wait_for_caller(sf, s);
        undistort(first_set, sf);
        launch_warp(*st, desc, fb, count, 0, st->n, sf, timer);
        // where the frame crosses from the front stream to the caller's stream: after the warp kernel (the small
        // pyramid-chain launches then fill in beside the next frame's warp) or after the chain
        static const bool split_after_warp = [] { const char* v = std::getenv("MST_OVERLAP_SPLIT"); return v && v[0] == '1'; }();
        auto cross = [&] {
            check_hip(hipEventRecord(ev_front_done_[slot], sf), "hipEventRecord");
            check_hip(hipStreamWaitEvent(s, ev_front_done_[slot], 0), "hipStreamWaitEvent");
        };
        if (overlap && !deep && split_after_warp) cross();
        hipStream_t sc = (overlap && !split_after_warp) ? sf : s;
        for (size_t i = 0; i < st->chain_lists.size(); ++i) {
            const MbPlanHost::ChainStep& step = st->plan.chain[i];
            if (st->chain_units[i].d[0]) {
                for (int j = 0; j < step.k; ++j)
                    mark(launch_pyrdown_stream(desc, st->n, st->chain_units[i].d[j], static_cast<int>(step.units[j].size()), step.base + j, count, sc));
            } else {
                mark(launch_pyr_chain(desc, st->n, st->chain_lists[i], step.base, step.k, count, sc));
            }
        }
        if (overlap && !deep && !split_after_warp) cross();
        int n_coarse = 0;
        size_t lds_bytes = 0;
        BlendArgs ba = st->blend_for(count, &n_coarse, &lds_bytes, deep);
        for (int f = 0; f < count; ++f) {
            ba.pano[f] = static_cast<uint8_t*>(d_panos[f]);
            ba.pano_mask[f] = d_masks ? static_cast<uint8_t*>(d_masks[f]) : nullptr;
        }
        ba.pano_stride = static_cast<uint32_t>(pano_stride);
        ba.mask_stride = static_cast<uint32_t>(mask_stride);
        if (!st->c1.empty())
            for (int f = 0; f < count; ++f) ba.c1[f] = st->c1[first_set + f];
        if (!st->c2.empty())
            for (int f = 0; f < count; ++f) ba.c2[f] = st->c2[first_set + f];
        const int n_fine = static_cast<int>(st->plan.block_order.size());
        if (ba.nlev > 1) {
            mark(launch_blend(desc, ba, n_fine, n_coarse, count, lds_bytes, s, 1));
            if (ba.c2[0]) mark(launch_blend(desc, ba, n_fine, n_coarse, count, lds_bytes, s, 4));
        }
        mark(launch_blend(desc, ba, n_fine, n_coarse, count, lds_bytes, s, 2));
        if (deep) {
            join.now();
        } else if (overlap) {
            check_hip(hipEventRecord(ev_canvas_done_[slot], s), "hipEventRecord");
            canvas_pending_[slot] = true;
        }
    } else {
        // the single-pass composites: the deep form of MST_FLAG_OVERLAP_FRAMES only (a call is one or two launches; the slot
        // owns the pre-pass buffers of its frame sets)
        const int slots = st->num_sets / st->sets_per_slot;
        const bool deep = (cfg_.flags & MST_FLAG_OVERLAP_FRAMES) && overlap_depth_ > 2 && slots > 2 && timer == nullptr && !in_host_compose_;
        hipStream_t caller = s;
        int slot = 0;
        if (deep) {
            slot = static_cast<int>(frame_counter_++ % slots);
            s = call_streams_[slot];
            order_after_writers(slot, s, output_ranges());
        } else {
            for (int k = 0; k < kMaxOverlap; ++k)
                if (canvas_pending_[k]) {
                    check_hip(hipStreamWaitEvent(s, ev_canvas_done_[k], 0), "hipStreamWaitEvent");
                    canvas_pending_[k] = false;
                }
        }
        struct Join {            // (every exit of this branch, the throwing ones included, leaves the caller's stream behind the slot's)
            Engine* e; bool on; int slot; hipStream_t from, to;
            ~Join() {
                if (!on) return;
                if (hipEventRecord(e->ev_canvas_done_[slot], from) == hipSuccess) {
                    e->canvas_pending_[slot] = true;
                    (void)hipStreamWaitEvent(to, e->ev_canvas_done_[slot], 0);
                }
            }
        } join{this, deep, slot, s, caller};
        wait_for_caller(s, s);
        undistort(slot * st->sets_per_slot, s);
        ComposeArgs ca = st->compose;
        ca.n_sets = count;
        bool aligned = true;       // the staged form reads whole dwords: frames on a 4-byte grid (any cv::Mat of 4n columns is)
        for (int f = 0; f < count; ++f) {
            ca.pano[f] = static_cast<uint8_t*>(d_panos[f]);
            ca.pano_mask[f] = d_masks ? static_cast<uint8_t*>(d_masks[f]) : nullptr;
            for (int i = 0; i < st->n; ++i)
                aligned &= ((reinterpret_cast<uintptr_t>(fb.f[f].ptr[i]) | fb.f[f].stride[i]) & 3u) == 0;
        }
        ca.pano_stride = static_cast<uint32_t>(pano_stride);
        ca.mask_stride = static_cast<uint32_t>(mask_stride);
        // (the short launches first: they run into the long one's ramp-up instead of trailing it)
        if (st->compose_general) {
            ca.tiles = st->d_compose_general;
            ca.n_tiles = st->compose_general;
            mark(launch_compose(ca, fb, 0, true, s));
        }
        if (st->compose_direct) {
            ca.tiles = st->d_compose_direct;
            ca.n_tiles = st->compose_direct;
            mark(launch_compose(ca, fb, 0, false, s));
        }
        if (st->compose_staged && aligned) {
            ca.tiles = st->d_compose_staged;
            ca.n_tiles = st->compose_staged;
            mark(launch_compose(ca, fb, st->compose_lds_bytes, false, s));
        } else if (st->compose_staged) {
            // (frames off the 4-byte grid: the staged tiles' tables are in the staged kernel's lane order, which only that
            // kernel reads -- they take the per-pixel general path)
            ca.tiles = st->d_compose_staged;
            ca.n_tiles = st->compose_staged;
            mark(launch_compose(ca, fb, 0, true, s));
        }
    }
    check_hip(hipGetLastError(), "compose kernels");
    set_in_flight(std::move(st));
}

namespace {
// cv::Mat rows <-> tight device rows.  Continuous matrices (step == width bytes) take the 1-D path: the
// 2-D copy of odd-width rows is an order of magnitude slower on this stack (measured 12 ms vs 1 ms).
void copy_rows(void* dst, size_t dpitch, const void* src, size_t spitch, size_t row_bytes, size_t rows,
               hipMemcpyKind kind, hipStream_t s)
{
    if (dpitch == row_bytes && spitch == row_bytes)
        check_hip(hipMemcpyAsync(dst, src, row_bytes * rows, kind, s), "hipMemcpyAsync(image)");
    else
        check_hip(hipMemcpy2DAsync(dst, dpitch, src, spitch, row_bytes, rows, kind, s), "hipMemcpy2DAsync(image)");
}
}  // namespace

// ---- camera-sharded multi-GPU mode: per-camera stage and canvas stage as separate calls ----

void Engine::stage_cameras(uint32_t camera_mask, const void* const* d_frames, const size_t* strides, hipStream_t stream)
{
    select_device(cfg_.device);
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "mst_stage_cameras before a successful mst_update"};
    if (st->blend_kind != MST_BLEND_MULTI_BAND) throw Error{MST_ERR_INVALID_ARG, "stage calls exist for multi-band only"};
    if (!st->nv12_frames.empty()) throw Error{MST_ERR_INVALID_ARG, "stage calls take BGR frames (MST_FLAG_FRAMES_NV12 is for the compose calls)"};
    if (!d_frames || !strides) throw Error{MST_ERR_INVALID_ARG, "null frame table"};
    hipStream_t s = stream ? stream : stream_;
    FrameBatch fb{};
    FrameArgs& fa = fb.f[0];
    for (int i = 0; i < st->n; ++i) {
        if (!(camera_mask >> i & 1u)) continue;
        if (!d_frames[i] || strides[i] < static_cast<size_t>(st->cams[i].src_w) * 3 || strides[i] >= (size_t{1} << 24) ||
            strides[i] * static_cast<size_t>(st->cams[i].src_h) > 0xFFFFFFFFull)
            throw Error{MST_ERR_INVALID_ARG, "bad frame for a camera in the mask (stride below 16 MiB, frame below 4 GiB)"};
        fa.ptr[i] = static_cast<const uint8_t*>(d_frames[i]);
        fa.stride[i] = static_cast<uint32_t>(strides[i]);
    }
    if (stage_set_ >= st->num_sets) throw Error{MST_ERR_INVALID_ARG, "stage set beyond mst_config.max_batch"};
    const CamDesc* set_desc = st->d_desc + static_cast<size_t>(stage_set_) * st->n;
    if (st->any_undistort) {            // lens-undistort pre-pass for the cameras of this call (the selected frame set)
        FrameBatch ub = fb;
        for (int i = 0; i < st->n; ++i)
            if (uint8_t* buf = (camera_mask >> i & 1u) ? st->undist_frames[static_cast<size_t>(stage_set_) * st->n + i] : nullptr) {
                ub.f[0].ptr[i] = buf;
                ub.f[0].stride[i] = static_cast<uint32_t>(st->cams[i].src_w) * 3u;
            }
        launch_undistort(*st, fb, ub, 1, camera_mask, s);
        fb = ub;
    }
    const MbPlanHost& plan = st->plan;
    // one launch per run of consecutive cameras in the mask (the tile lists are in camera order)
    std::vector<std::pair<int, int>> runs;
    for (int i = 0; i < st->n; ++i) {
        if (!(camera_mask >> i & 1u)) continue;
        if (!runs.empty() && runs.back().second == i) runs.back().second = i + 1;
        else runs.emplace_back(i, i + 1);
    }
    for (const auto& r : runs) launch_warp(*st, set_desc, fb, 1, r.first, r.second, s);
    for (size_t k = 0; k < st->chain_lists.size(); ++k)
        for (const auto& r : runs) {
            if (st->chain_units[k].d[0]) {
                const MbPlanHost::ChainStep& step = plan.chain[k];
                for (int j = 0; j < step.k; ++j)
                    launch_pyrdown_stream(set_desc, st->n, st->chain_units[k].d[j] + step.unit_cam_start[j][r.first],
                                          step.unit_cam_start[j][r.second] - step.unit_cam_start[j][r.first], step.base + j, 1, s);
                continue;
            }
            TileList tl{};
            tl.tiles = st->chain_lists[k].tiles + plan.chain[k].cam_start[r.first];
            tl.count = plan.chain[k].cam_start[r.second] - plan.chain[k].cam_start[r.first];
            tl.first = plan.chain[k].cam_start[r.first];
            launch_pyr_chain(set_desc, st->n, tl, plan.chain[k].base, plan.chain[k].k, 1, s);
        }
    check_hip(hipGetLastError(), "camera stage kernels");
    set_in_flight(std::move(st));
}

void Engine::stage_canvas(void* d_pano, size_t pano_stride, void* d_mask, size_t mask_stride, hipStream_t stream)
{
    select_device(cfg_.device);
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "mst_stage_canvas before a successful mst_update"};
    if (st->blend_kind != MST_BLEND_MULTI_BAND) throw Error{MST_ERR_INVALID_ARG, "stage calls exist for multi-band only"};
    if (!d_pano || pano_stride < static_cast<size_t>(st->out_w) * 3) throw Error{MST_ERR_INVALID_ARG, "bad panorama buffer"};
    if (pano_stride >= (size_t{1} << 24) || mask_stride >= (size_t{1} << 24))
        throw Error{MST_ERR_INVALID_ARG, "panorama and mask strides must be below 16 MiB"};
    if (d_mask && mask_stride < static_cast<size_t>(st->out_w)) throw Error{MST_ERR_INVALID_ARG, "mask stride too small"};
    if (pano_stride * static_cast<size_t>(st->out_h) > 0xFFFFFFFFull)
        throw Error{MST_ERR_INVALID_ARG, "panoramas of 4 GiB and more are not supported (32-bit byte offsets)"};
    hipStream_t s = stream ? stream : stream_;
    int n_coarse = 0;
    size_t lds_bytes = 0;
    BlendArgs ba = st->blend_for(1, &n_coarse, &lds_bytes);
    ba.pano[0] = static_cast<uint8_t*>(d_pano); ba.pano_stride = static_cast<uint32_t>(pano_stride);
    ba.pano_mask[0] = static_cast<uint8_t*>(d_mask); ba.mask_stride = static_cast<uint32_t>(mask_stride);
    if (stage_set_ >= st->num_sets) throw Error{MST_ERR_INVALID_ARG, "stage set beyond mst_config.max_batch"};
    if (!st->c1.empty()) ba.c1[0] = st->c1[stage_set_];
    if (!st->c2.empty()) ba.c2[0] = st->c2[stage_set_];
    launch_blend(st->d_desc + static_cast<size_t>(stage_set_) * st->n, ba, static_cast<int>(st->plan.block_order.size()),
                 n_coarse, 1, lds_bytes, s);
    check_hip(hipGetLastError(), "canvas stage kernel");
    set_in_flight(std::move(st));
}

void Engine::select_stage_set(int set)
{
    if (set < 0 || set >= kMaxBatch) throw Error{MST_ERR_INVALID_ARG, "bad stage set"};
    stage_set_ = set;
}

size_t Engine::payload_bytes(int camera)
{
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "no calibration"};
    if (st->blend_kind != MST_BLEND_MULTI_BAND || camera < 0 || camera >= st->n)
        throw Error{MST_ERR_INVALID_ARG, "no payload for that camera"};
    return static_cast<size_t>(st->plan.payload_px[camera]) * sizeof(uint32_t);
}

void Engine::payload_copy(int camera, void* d_buf, bool pack, hipStream_t stream)
{
    select_device(cfg_.device);
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "no calibration"};
    if (st->blend_kind != MST_BLEND_MULTI_BAND || camera < 0 || camera >= st->n || !d_buf)
        throw Error{MST_ERR_INVALID_ARG, "no payload for that camera, or null buffer"};
    if (stage_set_ >= st->num_sets) throw Error{MST_ERR_INVALID_ARG, "stage set beyond mst_config.max_batch"};
    const int r0 = st->plan.payload_cam_start[camera], r1 = st->plan.payload_cam_start[camera + 1];
    launch_payload_copy(st->d_desc + static_cast<size_t>(stage_set_) * st->n + camera, st->d_payload_rects + r0, r1 - r0,
                        static_cast<uint32_t*>(d_buf), pack, stream ? stream : stream_);
    check_hip(hipGetLastError(), "k_payload_copy");
    set_in_flight(std::move(st));
}

void Engine::camera_mask(int camera, uint8_t* mask, size_t stride)
{
    select_device(cfg_.device);
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "no calibration"};
    if (camera < 0 || camera >= st->n || !mask) throw Error{MST_ERR_INVALID_ARG, "bad camera or null buffer"};
    const Rect& r = st->cams[camera].roi;
    if (stride < static_cast<size_t>(r.width)) throw Error{MST_ERR_INVALID_ARG, "mask stride too small"};
    check_hip(hipMemcpy2D(mask, stride, st->desc[camera].mask, r.width, r.width, r.height, hipMemcpyDeviceToHost),
              "hipMemcpy2D(mask)");
}

void Engine::pyramid_buffer(int camera, void** d_ptr, size_t* bytes)
{
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "no calibration"};
    if (st->blend_kind != MST_BLEND_MULTI_BAND || camera < 0 || camera >= st->n)
        throw Error{MST_ERR_INVALID_ARG, "no pyramid for that camera"};
    if (d_ptr) *d_ptr = st->pyr_base[camera];
    if (bytes) *bytes = st->pyr_bytes[camera];
}

void Engine::compose_host(const uint8_t* const* frames, const size_t* strides, uint8_t* pano, size_t pano_stride,
                          uint8_t* pano_mask, size_t mask_stride)
{
    select_device(cfg_.device);
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "mst_compose before a successful mst_update"};
    if (!frames || !strides || !pano) throw Error{MST_ERR_INVALID_ARG, "null frame or panorama pointer"};
    const int n = st->n;
    if (d_frames_.size() != static_cast<size_t>(n)) { d_frames_.assign(n, nullptr); d_frame_bytes_.assign(n, 0); }
    std::vector<const void*> dptr(n);
    std::vector<size_t> dstride(n);
    for (int i = 0; i < n; ++i) {
        const int w = st->cams[i].src_w, h = st->cams[i].src_h;
        const bool nv12 = (cfg_.flags & MST_FLAG_FRAMES_NV12) != 0;      // (h * 3 / 2 rows of w bytes instead of h rows of 3 w)
        if (!frames[i] || strides[i] < static_cast<size_t>(w) * (nv12 ? 1 : 3)) throw Error{MST_ERR_INVALID_ARG, "bad frame"};
        const size_t row = static_cast<size_t>(w) * (nv12 ? 1 : 3), rows = nv12 ? static_cast<size_t>(h) * 3 / 2 : h, bytes = row * rows;
        if (d_frame_bytes_[i] < bytes) {
            if (d_frames_[i]) (void)hipFree(d_frames_[i]);
            d_frames_[i] = nullptr;
            check_hip(hipMalloc(reinterpret_cast<void**>(&d_frames_[i]), bytes), "hipMalloc(frame)");
            d_frame_bytes_[i] = bytes;
        }
        copy_rows(d_frames_[i], row, frames[i], strides[i], row, rows, hipMemcpyHostToDevice, stream_);
        dptr[i] = d_frames_[i];
        dstride[i] = row;
    }
    const size_t prow = static_cast<size_t>(st->out_w) * 3, pbytes = prow * st->out_h;
    const size_t mrow = st->out_w, mbytes = mrow * st->out_h;
    if (d_pano_bytes_ < pbytes) {
        if (d_pano_) (void)hipFree(d_pano_);
        d_pano_ = nullptr;
        check_hip(hipMalloc(reinterpret_cast<void**>(&d_pano_), pbytes), "hipMalloc(pano)");
        d_pano_bytes_ = pbytes;
    }
    if (pano_mask && d_mask_bytes_ < mbytes) {
        if (d_mask_) (void)hipFree(d_mask_);
        d_mask_ = nullptr;
        check_hip(hipMalloc(reinterpret_cast<void**>(&d_mask_), mbytes), "hipMalloc(mask)");
        d_mask_bytes_ = mbytes;
    }
    in_host_compose_ = true;     // the H2D copies above are queued on stream_: keep everything in stream order
    try {
        void* panos[1] = {d_pano_};
        void* masks[1] = {d_mask_};
        compose_batch_on(st, 1, dptr.data(), dstride.data(), panos, prow, pano_mask ? masks : nullptr, mrow, stream_, nullptr);
    } catch (...) {
        in_host_compose_ = false;
        throw;
    }
    in_host_compose_ = false;
    // (a strip engine writes its own columns only: those -- and nothing of the rest of the device buffer, which is
    // uninitialised -- go back to the caller's canvas-shaped buffer)
    const size_t x0 = cfg_.strip_count > 1 ? static_cast<size_t>(std::min(st->strip_x0, st->out_w)) : 0;
    const size_t x1 = cfg_.strip_count > 1 ? static_cast<size_t>(std::min(st->strip_x1, st->out_w)) : static_cast<size_t>(st->out_w);
    if (x1 > x0) {
        copy_rows(pano + x0 * 3, pano_stride, d_pano_ + x0 * 3, prow, (x1 - x0) * 3, st->out_h, hipMemcpyDeviceToHost, stream_);
        if (pano_mask) copy_rows(pano_mask + x0, mask_stride, d_mask_ + x0, mrow, x1 - x0, st->out_h, hipMemcpyDeviceToHost, stream_);
    }
    check_hip(hipStreamSynchronize(stream_), "hipStreamSynchronize(compose)");
}

void Engine::pipeline_submit(const uint8_t* const* frames, const size_t* strides, uint8_t* pano, size_t pano_stride,
                             ::mst_shm* const* clients, int timeout_ms)
{
    select_device(cfg_.device);
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "mst_pipeline_submit before a successful mst_update"};
    if ((!clients && (!frames || !strides)) || !pano) throw Error{MST_ERR_INVALID_ARG, "null frame or panorama pointer"};
    if (pipeline_pending() >= kPipeDepth) throw Error{MST_ERR_INVALID_ARG, "pipeline full: call mst_pipeline_wait first"};
    const size_t prow = static_cast<size_t>(st->out_w) * 3, pbytes = prow * st->out_h;
    if (pano_stride < prow) throw Error{MST_ERR_INVALID_ARG, "panorama stride too small"};
    if (!pipe_in_) {
        check_hip(hipStreamCreateWithFlags(&pipe_in_, hipStreamNonBlocking), "hipStreamCreate");
        check_hip(hipStreamCreateWithFlags(&pipe_out_, hipStreamNonBlocking), "hipStreamCreate");
        for (PipeSlot& ps : pipe_) {
            check_hip(hipEventCreateWithFlags(&ps.h2d_done, hipEventDisableTiming), "hipEventCreate");
            check_hip(hipEventCreateWithFlags(&ps.compute_done, hipEventDisableTiming), "hipEventCreate");
            check_hip(hipEventCreateWithFlags(&ps.d2h_done, hipEventDisableTiming), "hipEventCreate");
        }
    }
    PipeSlot& ps = pipe_[pipe_submitted_ % kPipeDepth];       // free: its previous frame has been waited for
    const int n = st->n;
    if (ps.d_frames.size() != static_cast<size_t>(n)) { ps.d_frames.assign(n, nullptr); ps.d_frame_bytes.assign(n, 0); }
    std::vector<const void*> dptr(n);
    std::vector<size_t> dstride(n);
    // stage 1 (copy-in stream): the slot's device frames were last read by kernels whose completion the matching
    // wait() has already observed (d2h_done follows compute_done), so the copies may start right away
    const bool nv12 = (cfg_.flags & MST_FLAG_FRAMES_NV12) != 0;
    for (int i = 0; i < n; ++i) {
        const int w = st->cams[i].src_w, h = st->cams[i].src_h;
        if (!clients && (!frames[i] || strides[i] < static_cast<size_t>(w) * (nv12 ? 1 : 3))) throw Error{MST_ERR_INVALID_ARG, "bad frame"};
        const size_t row = static_cast<size_t>(w) * (nv12 ? 1 : 3), rows = nv12 ? static_cast<size_t>(h) * 3 / 2 : h, bytes = row * rows;
        if (ps.d_frame_bytes[i] < bytes) {
            if (ps.d_frames[i]) (void)hipFree(ps.d_frames[i]);
            ps.d_frames[i] = nullptr;
            check_hip(hipMalloc(reinterpret_cast<void**>(&ps.d_frames[i]), bytes + 16), "hipMalloc(frame)");
            ps.d_frame_bytes[i] = bytes;
        }
        if (!clients) copy_rows(ps.d_frames[i], row, frames[i], strides[i], row, rows, hipMemcpyHostToDevice, pipe_in_);
        dptr[i] = ps.d_frames[i];
        dstride[i] = row;
    }
    if (clients) {
        // the frame set straight out of the capture processes' shared-memory segments (mi355_ingest.h): a camera's copy is queued on
        // the copy-in stream when its packet lands, the camera is asked for its next frame by a host function behind that copy
        // (MST_SHM_ASYNC) -- this call waits for packets, never for the bus
        std::vector<mst_frame_packet> hdr(n);
        std::vector<void*> dst(n);
        std::vector<size_t> cap(n);
        for (int i = 0; i < n; ++i) { dst[i] = ps.d_frames[i]; cap[i] = ps.d_frame_bytes[i]; }
        const int rc = mst_shm_get_device_multi_ex(clients, n, hdr.data(), dst.data(), dstride.data(), cap.data(), pipe_in_, timeout_ms, MST_SHM_ASYNC);
        if (rc != MST_OK) throw Error{rc, rc == MST_ERR_TIMEOUT ? "a capture process did not deliver its frame in time" : "bad frame packet"};
        for (int i = 0; i < n; ++i)
            if (hdr[i].width != st->cams[i].src_w || hdr[i].height != st->cams[i].src_h ||
                mst_shm_packet_format(&hdr[i]) != (nv12 ? MST_PIXEL_NV12 : MST_PIXEL_BGR))
                throw Error{MST_ERR_INVALID_ARG, "a frame packet's size or pixel format is not what the calibration expects"};
    }
    check_hip(hipEventRecord(ps.h2d_done, pipe_in_), "hipEventRecord");
    if (ps.d_pano_bytes < pbytes) {
        if (ps.d_pano) (void)hipFree(ps.d_pano);
        ps.d_pano = nullptr;
        check_hip(hipMalloc(reinterpret_cast<void**>(&ps.d_pano), pbytes), "hipMalloc(pano)");
        ps.d_pano_bytes = pbytes;
    }
    // stage 2 (compute stream): kernels of this frame behind its copies
    check_hip(hipStreamWaitEvent(stream_, ps.h2d_done, 0), "hipStreamWaitEvent");
    in_host_compose_ = true;
    try {
        void* panos[1] = {ps.d_pano};
        compose_batch_on(st, 1, dptr.data(), dstride.data(), panos, prow, nullptr, 0, stream_, nullptr);
    } catch (...) {
        in_host_compose_ = false;
        throw;
    }
    in_host_compose_ = false;
    check_hip(hipEventRecord(ps.compute_done, stream_), "hipEventRecord");
    // stage 3 (copy-out stream): the panorama back to the caller's buffer
    check_hip(hipStreamWaitEvent(pipe_out_, ps.compute_done, 0), "hipStreamWaitEvent");
    {
        const size_t x0 = cfg_.strip_count > 1 ? static_cast<size_t>(std::min(st->strip_x0, st->out_w)) : 0;
        const size_t x1 = cfg_.strip_count > 1 ? static_cast<size_t>(std::min(st->strip_x1, st->out_w)) : static_cast<size_t>(st->out_w);
        if (x1 > x0)     // (a strip engine: its own columns only, see compose_host)
            copy_rows(pano + x0 * 3, pano_stride, ps.d_pano + x0 * 3, prow, (x1 - x0) * 3, st->out_h, hipMemcpyDeviceToHost, pipe_out_);
    }
    check_hip(hipEventRecord(ps.d2h_done, pipe_out_), "hipEventRecord");
    ps.host_pano = pano;
    ps.canvas = mst_rect{st->canvas.x + st->crop_x, st->canvas.y + st->crop_y, st->out_w, st->out_h};
    ps.busy = true;
    ++pipe_submitted_;
}

uint8_t* Engine::pipeline_wait(mst_rect* canvas)
{
    select_device(cfg_.device);
    if (pipeline_pending() <= 0) throw Error{MST_ERR_INVALID_ARG, "mst_pipeline_wait without a submitted frame"};
    PipeSlot& ps = pipe_[pipe_completed_ % kPipeDepth];
    check_hip(hipEventSynchronize(ps.d2h_done), "hipEventSynchronize(pipeline)");
    ps.busy = false;
    ++pipe_completed_;
    if (canvas) *canvas = ps.canvas;
    return ps.host_pano;
}

void Engine::debug_read(int what, int camera, int level, void* buf, size_t buf_bytes, int* w, int* h)
{
    select_device(cfg_.device);
    std::shared_ptr<CalibState> st = state();
    if (!st) throw Error{MST_ERR_NOT_CALIBRATED, "no calibration"};
    if (st->blend_kind != MST_BLEND_MULTI_BAND) throw Error{MST_ERR_INVALID_ARG, "debug buffers exist for multi-band only"};
    if (level < 0 || level > st->num_bands) throw Error{MST_ERR_INVALID_ARG, "bad level"};
    const void* src = nullptr;
    size_t bytes = 0;
    int ww = 0, hh = 0;
    const bool per_cam = what == MST_DBG_CAM_WARPED_S16 || what == MST_DBG_CAM_GAUSS_S16 ||
                         what == MST_DBG_CAM_WEIGHT_F32 || what == MST_DBG_CAM_NEED_U8;
    if (per_cam) {
        if (camera < 0 || camera >= st->n) throw Error{MST_ERR_INVALID_ARG, "bad camera"};
        const CamDesc& d = st->desc[camera];
        if (what == MST_DBG_CAM_WARPED_S16) level = 0;
        ww = d.lw[level]; hh = d.lh[level];
        const size_t px = static_cast<size_t>(ww) * hh;
        if (what == MST_DBG_CAM_NEED_U8) {
            // host-side work-list mask: which Gaussian pixels of this level can reach the panorama
            if (w) *w = ww;
            if (h) *h = hh;
            if (!buf) return;
            if (buf_bytes < px) throw Error{MST_ERR_INVALID_ARG, "debug buffer too small"};
            std::memcpy(buf, st->plan.cams[camera].need[level].data(), px);
            return;
        }
        if (what == MST_DBG_CAM_WEIGHT_F32) { src = d.W[level]; bytes = sizeof(float) * px; }
        else {
            // packed B | G<<8 | R<<16 dwords on the device -> planar CV_16S for the caller
            if (w) *w = ww;
            if (h) *h = hh;
            if (!buf) return;
            if (buf_bytes < sizeof(int16_t) * 3 * px) throw Error{MST_ERR_INVALID_ARG, "debug buffer too small"};
            int16_t* out = static_cast<int16_t*>(buf);
            check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
            if (level == 0) {        // packed BGR8 (or B G R x dwords, MST_L0_PACKED=0), rows l0_pitch bytes apart
                std::vector<uint8_t> bytes8(static_cast<size_t>(d.l0_pitch) * hh);
                check_hip(hipMemcpy(bytes8.data(), d.G[0], bytes8.size(), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
                for (int y = 0; y < hh; ++y)
                    for (int x = 0; x < ww; ++x) {
                        const uint8_t* p8 = &bytes8[static_cast<size_t>(y) * d.l0_pitch + kL0Bpp * static_cast<size_t>(x)];
                        const size_t i = static_cast<size_t>(y) * ww + x;
                        out[i] = p8[0]; out[px + i] = p8[1]; out[2 * px + i] = p8[2];
                    }
                return;
            }
            std::vector<uint32_t> packed(px);
            check_hip(hipMemcpy(packed.data(), d.G[level], px * sizeof(uint32_t), hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
            for (size_t i = 0; i < px; ++i) {
                out[i] = static_cast<int16_t>(packed[i] & 0xFF);
                out[px + i] = static_cast<int16_t>((packed[i] >> 8) & 0xFF);
                out[2 * px + i] = static_cast<int16_t>((packed[i] >> 16) & 0xFF);
            }
            return;
        }
    } else if (what == MST_DBG_BAND_WEIGHT_F32) {
        ww = st->cv.cw[level]; hh = st->cv.ch[level];
        src = st->cv.WS[level]; bytes = sizeof(float) * static_cast<size_t>(ww) * hh;
    } else if (what == MST_DBG_CANVAS_S16) {
        if (level == 0) throw Error{MST_ERR_INVALID_ARG, "canvas level 0 is written straight to the panorama"};
        if (!st->cv.C[level]) throw Error{MST_ERR_INVALID_ARG, "create the engine with MST_FLAG_DEBUG to dump canvas levels"};
        ww = st->cv.cw[level]; hh = st->cv.ch[level];
        src = st->cv.C[level]; bytes = sizeof(int16_t) * 3 * static_cast<size_t>(ww) * hh;
    } else {
        throw Error{MST_ERR_INVALID_ARG, "unknown debug buffer"};
    }
    if (w) *w = ww;
    if (h) *h = hh;
    if (!buf) return;
    if (buf_bytes < bytes) throw Error{MST_ERR_INVALID_ARG, "debug buffer too small"};
    check_hip(hipDeviceSynchronize(), "hipDeviceSynchronize");
    check_hip(hipMemcpy(buf, src, bytes, hipMemcpyDeviceToHost), "hipMemcpy(D2H)");
}

}  // namespace mst
