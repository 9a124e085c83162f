// api.cpp -- the C ABI declared in include/mi355_stitch.h.  No exception crosses it.
#include <cmath>
#include <cstring>
#include <new>

#include <algorithm>

#include "engine.hpp"
#include "../../include/mi355_ingest.h"
#include "kernels.hpp"

struct mst_engine {
    mst::Engine impl;
    explicit mst_engine(const mst_config& c) : impl(c) {}
};

namespace {

thread_local std::string g_last_error;

int fail(mst_engine* e, const mst::Error& err)
{
    // per thread, not per engine: the calibration thread (mst_update) and the frame thread (mst_compose*) may fail
    // concurrently on one engine, and a shared string would race and dangle
    (void)e;
    g_last_error = err.msg;
    return err.code;
}

template <typename F> int guarded(mst_engine* e, F&& f)
{
    try {
        f();
        return MST_OK;
    } catch (const mst::Error& err) {
        return fail(e, err);
    } catch (const std::bad_alloc&) {
        return fail(e, mst::Error{MST_ERR_OUT_OF_MEMORY, "host allocation failed"});
    } catch (const std::exception& ex) {
        return fail(e, mst::Error{MST_ERR_INVALID_ARG, ex.what()});
    }
}

void require(bool cond, const char* msg)
{
    if (!cond) throw mst::Error{MST_ERR_INVALID_ARG, msg};
}

bool valid_warp(int k)
{
    const int base = k & 0xFF;
    return k >= 0 && (k >> 10) == 0 && base < mst::kWarpKinds;
}

mst::Projector make_projector(int kind, float scale, const float K[9], const float R[9], const float T[3])
{
    require(valid_warp(kind), "unknown warp kind");
    require((kind & MST_WARP_AB_CUSTOM) != MST_WARP_AB_CUSTOM, "MST_WARP_AB_CUSTOM needs an engine (mst_config.warp_a / warp_b)");
    require(K && R, "K and R must not be null");
    require(scale > 0.f, "scale must be > 0");
    mst::Projector p;
    mst::set_projector_kind(p, kind, 1.f, 1.f);
    p.scale = scale;
    p.set(K, R, T);
    return p;
}

// one-camera device context for the stage-level calls
struct StageCtx {
    mst::DevicePool pool;
    hipStream_t s = nullptr;
    mst::CamDesc desc{};
    mst::CamDesc* d_desc = nullptr;
    ~StageCtx() { if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); } }
};

}  // namespace

extern "C" {

int mst_warp_kind_from_name(const char* name)
{
    if (!name) return -1;
    static const struct { const char* name; int kind; } kNames[] = {      // CLI:556-570
        {"plane", MST_WARP_PLANE}, {"cylindrical", MST_WARP_CYLINDRICAL}, {"spherical", MST_WARP_SPHERICAL},
        {"fisheye", MST_WARP_FISHEYE}, {"stereographic", MST_WARP_STEREOGRAPHIC},
        {"compressedPlaneA2B1", MST_WARP_COMPRESSED_PLANE | MST_WARP_A2B1},
        {"compressedPlaneA1.5B1", MST_WARP_COMPRESSED_PLANE | MST_WARP_A1_5B1},
        {"compressedPlanePortraitA2B1", MST_WARP_COMPRESSED_PLANE_PORTRAIT | MST_WARP_A2B1},
        {"compressedPlanePortraitA1.5B1", MST_WARP_COMPRESSED_PLANE_PORTRAIT | MST_WARP_A1_5B1},
        {"paniniA2B1", MST_WARP_PANINI | MST_WARP_A2B1}, {"paniniA1.5B1", MST_WARP_PANINI | MST_WARP_A1_5B1},
        {"paniniPortraitA2B1", MST_WARP_PANINI_PORTRAIT | MST_WARP_A2B1},
        {"paniniPortraitA1.5B1", MST_WARP_PANINI_PORTRAIT | MST_WARP_A1_5B1},
        {"mercator", MST_WARP_MERCATOR}, {"transverseMercator", MST_WARP_TRANSVERSE_MERCATOR},
    };
    for (const auto& e : kNames)
        if (std::strcmp(name, e.name) == 0) return e.kind;
    return -1;
}

int mst_create(const mst_config* cfg, mst_engine** out)
{
    if (out) *out = nullptr;
    {
        // a timing build (-DMST_ABLATE_*: wrong pixels) or a stamped one must never pass for the product silently
        static const bool warned = [] {
            const uint32_t bits = mst::mb_build_ablations() | mst::compose_build_ablations();
            if (bits) std::fprintf(stderr, "[mi355-stitch] WARNING: this library is a TIMING / DIAGNOSTIC build (mst_build_info.timing_build = %u): "
                                           "its pixels may be wrong; not for production or recorded measurements\n", bits);
            return true;
        }();
        (void)warned;
    }
    mst_engine* created = nullptr;
    int rc = guarded(nullptr, [&] {
        require(cfg && out, "null config or output pointer");
        require(cfg->struct_size == static_cast<int32_t>(sizeof(mst_config)), "mst_config.struct_size mismatch");
        require(cfg->strip_count >= 0 && cfg->strip_count <= 64 && cfg->strip_index >= 0 &&
                (cfg->strip_count <= 1 ? cfg->strip_index == 0 : cfg->strip_index < cfg->strip_count), "bad strip_index / strip_count");
        require(cfg->overlap_depth >= 0 && cfg->overlap_depth <= 8, "overlap_depth out of range (0 .. 8)");
        require(cfg->num_cameras >= 1 && cfg->num_cameras <= MST_MAX_CAMERAS, "num_cameras out of range");
        require(valid_warp(cfg->warp_kind), "unknown warp kind");
        require((cfg->warp_kind & MST_WARP_AB_CUSTOM) != MST_WARP_AB_CUSTOM || (cfg->warp_a > 0.f && cfg->warp_b > 0.f),
                "MST_WARP_AB_CUSTOM needs positive mst_config.warp_a / warp_b");
        require(cfg->blend_kind >= MST_BLEND_NO && cfg->blend_kind <= MST_BLEND_MULTI_BAND, "unsupported blending method");
        require(cfg->expos_kind >= MST_EXPOS_NO && cfg->expos_kind <= MST_EXPOS_GAIN_BLOCKS,
                "unsupported exposure compensation method");
        require(cfg->num_bands <= MST_MAX_BANDS, "too many bands");
        require(!(cfg->flags & MST_FLAG_DRIVER_LUT) || cfg->blend_kind == MST_BLEND_NO,
                "MST_FLAG_DRIVER_LUT requires MST_BLEND_NO");
        require(!(cfg->flags & MST_FLAG_OVERLAP_FRAMES) || !(cfg->flags & MST_FLAG_DEBUG),
                "MST_FLAG_OVERLAP_FRAMES cannot be combined with MST_FLAG_DEBUG");
        require(!(cfg->flags & 64u), "flag 64 (the round-4 strip halo exchange) was removed: strips are self-sufficient");
        require(cfg->max_batch >= 0 && cfg->max_batch <= MST_MAX_BATCH, "max_batch out of range");
        require(cfg->weight_type == MST_WEIGHT_32F || cfg->weight_type == MST_WEIGHT_16S, "weight_type must be CV_32F or CV_16S");
        static_assert(MST_MAX_BATCH == mst::kMaxBatch, "public and device batch limits differ");
        created = new mst_engine(*cfg);
        created->impl.init();
    });
    if (rc != MST_OK) {
        delete created;
        return rc;
    }
    *out = created;
    return MST_OK;
}

void mst_destroy(mst_engine* e)
{
    delete e;
}

const char* mst_last_error(const mst_engine* e)
{
    (void)e;
    return g_last_error.c_str();
}

int mst_update(mst_engine* e, const mst_calibration* calib)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        require(calib != nullptr, "null calibration");
        require(calib->struct_size == static_cast<int32_t>(sizeof(mst_calibration)), "mst_calibration.struct_size mismatch");
        e->impl.update(*calib);
    });
}

int mst_release_calibration(mst_engine* e)
{
    if (!e) return MST_ERR_INVALID_ARG;
    e->impl.release();
    return MST_OK;
}

int mst_acquire_calibration(mst_engine* e, mst_rect* canvas, uint64_t* generation)
{
    if (!e) return MST_ERR_INVALID_ARG;
    auto st = e->impl.acquire();
    if (!st) return MST_ERR_NOT_CALIBRATED;
    if (canvas) {
        canvas->x = st->canvas.x + st->crop_x;
        canvas->y = st->canvas.y + st->crop_y;
        canvas->width = st->out_w;
        canvas->height = st->out_h;
    }
    if (generation) *generation = st->generation;
    return MST_OK;
}

int mst_canvas(const mst_engine* e, mst_rect* canvas)
{
    if (!e || !canvas) return MST_ERR_INVALID_ARG;
    auto st = e->impl.state();
    if (!st) return MST_ERR_NOT_CALIBRATED;
    canvas->x = st->canvas.x + st->crop_x;
    canvas->y = st->canvas.y + st->crop_y;
    canvas->width = st->out_w;
    canvas->height = st->out_h;
    return MST_OK;
}

int mst_strip(const mst_engine* e, mst_rect* strip)
{
    if (!e || !strip) return MST_ERR_INVALID_ARG;
    auto st = e->impl.state();
    if (!st) return MST_ERR_NOT_CALIBRATED;
    const int x0 = std::min(st->strip_x0, st->out_w), x1 = std::min(st->strip_x1, st->out_w);
    strip->x = st->canvas.x + st->crop_x + x0;
    strip->y = st->canvas.y + st->crop_y;
    strip->width = x1 - x0;
    strip->height = st->out_h;
    return MST_OK;
}

int mst_camera_roi(const mst_engine* e, int camera, mst_rect* roi)
{
    if (!e || !roi) return MST_ERR_INVALID_ARG;
    auto st = e->impl.state();
    if (!st) return MST_ERR_NOT_CALIBRATED;
    if (camera < 0 || camera >= st->n) return MST_ERR_INVALID_ARG;
    const mst::Rect& r = st->cams[camera].roi;
    roi->x = r.x; roi->y = r.y; roi->width = r.width; roi->height = r.height;
    return MST_OK;
}

int mst_camera_mask(mst_engine* e, int camera, uint8_t* mask, size_t stride)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.camera_mask(camera, mask, stride); });
}

int mst_num_bands(const mst_engine* e, int* num_bands)
{
    if (!e || !num_bands) return MST_ERR_INVALID_ARG;
    auto st = e->impl.state();
    if (!st) return MST_ERR_NOT_CALIBRATED;
    *num_bands = st->num_bands;
    return MST_OK;
}

int mst_compose(mst_engine* e, const uint8_t* const* frames, const size_t* frame_strides, uint8_t* pano,
                size_t pano_stride, uint8_t* pano_mask, size_t pano_mask_stride)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.compose_host(frames, frame_strides, pano, pano_stride, pano_mask, pano_mask_stride); });
}

int mst_compose_device(mst_engine* e, const void* const* d_frames, const size_t* frame_strides, void* d_pano,
                       size_t pano_stride, void* d_pano_mask, size_t pano_mask_stride, void* stream)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        e->impl.compose_device(d_frames, frame_strides, d_pano, pano_stride, d_pano_mask, pano_mask_stride,
                               static_cast<hipStream_t>(stream));
    });
}

int mst_compose_device_batch(mst_engine* e, int count, const void* const* d_frames, const size_t* frame_strides,
                             void* const* d_panos, size_t pano_stride, void* const* d_pano_masks,
                             size_t pano_mask_stride, void* stream)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        e->impl.compose_batch(count, d_frames, frame_strides, d_panos, pano_stride, d_pano_masks, pano_mask_stride,
                              static_cast<hipStream_t>(stream));
    });
}

int mst_compose_device_batch_ex(mst_engine* e, int count, const void* const* d_frames, const size_t* frame_strides,
                                void* const* d_panos, size_t pano_stride, void* const* d_pano_masks,
                                size_t pano_mask_stride, void* stream, void* frames_ready, void* outputs_free)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        e->impl.compose_batch(count, d_frames, frame_strides, d_panos, pano_stride, d_pano_masks, pano_mask_stride,
                              static_cast<hipStream_t>(stream), nullptr, static_cast<hipEvent_t>(frames_ready),
                              static_cast<hipEvent_t>(outputs_free));
    });
}

int mst_compose_profile(mst_engine* e, const void* const* d_frames, const size_t* frame_strides, void* d_pano,
                        size_t pano_stride, void* stream, int max_stages, char (*names)[32], float* stage_ms,
                        int* num_stages)
{
    void* panos[1] = {d_pano};
    return mst_compose_profile_batch(e, 1, d_frames, frame_strides, panos, pano_stride, stream, max_stages, names,
                                     stage_ms, num_stages);
}

int mst_compose_profile_batch(mst_engine* e, int count, const void* const* d_frames, const size_t* frame_strides,
                              void* const* d_panos, size_t pano_stride, void* stream, int max_stages,
                              char (*names)[32], float* stage_ms, int* num_stages)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        require(max_stages >= 0 && num_stages != nullptr, "bad stage buffer");
        mst::StageTimer timer;
        e->impl.compose_batch(count, d_frames, frame_strides, d_panos, pano_stride, nullptr, 0,
                              static_cast<hipStream_t>(stream), &timer);
        std::vector<std::pair<std::string, float>> stages;
        timer.collect(stages);
        *num_stages = static_cast<int>(stages.size());
        for (int i = 0; i < *num_stages && i < max_stages; ++i) {
            if (names) {
                std::strncpy(names[i], stages[i].first.c_str(), 31);
                names[i][31] = 0;
            }
            if (stage_ms) stage_ms[i] = stages[i].second;
        }
    });
}

int mst_stage_cameras(mst_engine* e, uint32_t camera_mask, const void* const* d_frames, const size_t* frame_strides,
                      void* stream)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.stage_cameras(camera_mask, d_frames, frame_strides, static_cast<hipStream_t>(stream)); });
}

int mst_stage_canvas(mst_engine* e, void* d_pano, size_t pano_stride, void* d_pano_mask, size_t pano_mask_stride,
                     void* stream)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        e->impl.stage_canvas(d_pano, pano_stride, d_pano_mask, pano_mask_stride, static_cast<hipStream_t>(stream));
    });
}

int mst_pyramid_buffer(mst_engine* e, int camera, void** d_ptr, size_t* bytes)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.pyramid_buffer(camera, d_ptr, bytes); });
}

int mst_payload_bytes(mst_engine* e, int camera, size_t* bytes)
{
    if (!e || !bytes) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { *bytes = e->impl.payload_bytes(camera); });
}

int mst_payload_pack(mst_engine* e, int camera, void* d_payload, void* stream)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.payload_copy(camera, d_payload, true, static_cast<hipStream_t>(stream)); });
}

int mst_payload_unpack(mst_engine* e, int camera, const void* d_payload, void* stream)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.payload_copy(camera, const_cast<void*>(d_payload), false, static_cast<hipStream_t>(stream)); });
}

int mst_stage_select_set(mst_engine* e, int set)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.select_stage_set(set); });
}

void* mst_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        g_last_error = "hipHostMalloc failed (no device, or out of page-locked memory)";
        return nullptr;
    }
    return p;
}

void mst_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

int mst_pipeline_submit(mst_engine* e, const uint8_t* const* frames, const size_t* frame_strides, uint8_t* pano,
                        size_t pano_stride)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.pipeline_submit(frames, frame_strides, pano, pano_stride); });
}

int mst_pipeline_submit_shm(mst_engine* e, mst_shm* const* clients, int num_clients, uint8_t* pano, size_t pano_stride, int timeout_ms)
{
    if (!e || !clients) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        if (num_clients != e->impl.config().num_cameras) throw mst::Error{MST_ERR_INVALID_ARG, "one shared-memory client per camera"};
        e->impl.pipeline_submit(nullptr, nullptr, pano, pano_stride, clients, timeout_ms);
    });
}

int mst_pipeline_wait(mst_engine* e, uint8_t** pano, mst_rect* canvas)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] {
        uint8_t* p = e->impl.pipeline_wait(canvas);
        if (pano) *pano = p;
    });
}

int mst_pipeline_pending(const mst_engine* e)
{
    return e ? e->impl.pipeline_pending() : MST_ERR_INVALID_ARG;
}

int mst_synchronize(mst_engine* e)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.synchronize(); });
}

int mst_debug_read(mst_engine* e, int what, int camera, int level, void* buf, size_t buf_bytes, int* width, int* height)
{
    if (!e) return MST_ERR_INVALID_ARG;
    return guarded(e, [&] { e->impl.debug_read(what, camera, level, buf, buf_bytes, width, height); });
}

// ---- stage-level ---------------------------------------------------------------------------------

int mst_warp_roi(int warp_kind, float scale, const float K[9], const float R[9], const float T[3], int src_width,
                 int src_height, mst_rect* roi)
{
    return guarded(nullptr, [&] {
        require(roi != nullptr, "null roi");
        require(src_width > 0 && src_height > 0, "bad source size");
        mst::Projector p = make_projector(warp_kind, scale, K, R, T);
        mst::Rect r = mst::warp_roi(p, src_width, src_height);
        roi->x = r.x; roi->y = r.y; roi->width = r.width; roi->height = r.height;
    });
}

int mst_build_maps_host(int warp_kind, float scale, const float K[9], const float R[9], const float T[3],
                        int src_width, int src_height, float* xmap, float* ymap)
{
    return guarded(nullptr, [&] {
        require(xmap && ymap, "null map pointer");
        require(src_width > 0 && src_height > 0, "bad source size");
        mst::Projector p = make_projector(warp_kind, scale, K, R, T);
        mst::Rect r = mst::warp_roi(p, src_width, src_height);
        mst::MapTables t = mst::build_map_tables(p, r);
        mst::backward_map_host(p, r, t, xmap, ymap);
    });
}

int mst_build_maps(int device, int warp_kind, float scale, const float K[9], const float R[9], const float T[3],
                   int src_width, int src_height, float* xmap, float* ymap)
{
    return guarded(nullptr, [&] {
        require(xmap && ymap, "null map pointer");
        require(src_width > 0 && src_height > 0, "bad source size");
        mst::Projector p = make_projector(warp_kind, scale, K, R, T);
        mst::Rect r = mst::warp_roi(p, src_width, src_height);
        mst::select_device(device);
        StageCtx ctx;
        mst::check_hip(hipStreamCreate(&ctx.s), "hipStreamCreate");
        mst::fill_camera_desc(ctx.pool, ctx.s, p, r, src_width, src_height, MST_EXPOS_NO, nullptr, 0, 0, 1.f, ctx.desc);
        ctx.d_desc = ctx.pool.upload(&ctx.desc, 1, ctx.s);
        const size_t n = static_cast<size_t>(r.width) * r.height;
        float* dx = ctx.pool.alloc<float>(n);
        float* dy = ctx.pool.alloc<float>(n);
        mst::launch_build_maps(ctx.d_desc, dx, dy, r.width, r.height, ctx.s);
        mst::check_hip(hipGetLastError(), "k_build_maps");
        mst::check_hip(hipMemcpyAsync(xmap, dx, n * sizeof(float), hipMemcpyDeviceToHost, ctx.s), "hipMemcpyAsync");
        mst::check_hip(hipMemcpyAsync(ymap, dy, n * sizeof(float), hipMemcpyDeviceToHost, ctx.s), "hipMemcpyAsync");
        mst::check_hip(hipStreamSynchronize(ctx.s), "hipStreamSynchronize");
    });
}

int mst_warp(int device, int warp_kind, float scale, const float K[9], const float R[9], const float T[3],
             const uint8_t* src, int src_width, int src_height, size_t src_stride, int channels, int interp,
             int border, uint8_t* dst, size_t dst_stride)
{
    return guarded(nullptr, [&] {
        require(src && dst, "null image pointer");
        require(src_width > 0 && src_height > 0, "bad source size");
        require(channels == 1 || channels == 3, "only CV_8UC1 / CV_8UC3");
        require(interp == MST_INTER_NEAREST || interp == MST_INTER_LINEAR, "unsupported interpolation");
        require(border == MST_BORDER_CONSTANT || border == MST_BORDER_REFLECT, "unsupported border mode");
        require(src_stride >= static_cast<size_t>(src_width) * channels, "source stride too small");
        mst::Projector p = make_projector(warp_kind, scale, K, R, T);
        mst::Rect r = mst::warp_roi(p, src_width, src_height);
        require(dst_stride >= static_cast<size_t>(r.width) * channels, "destination stride too small");
        mst::select_device(device);
        StageCtx ctx;
        mst::check_hip(hipStreamCreate(&ctx.s), "hipStreamCreate");
        mst::fill_camera_desc(ctx.pool, ctx.s, p, r, src_width, src_height, MST_EXPOS_NO, nullptr, 0, 0, 1.f, ctx.desc);
        ctx.d_desc = ctx.pool.upload(&ctx.desc, 1, ctx.s);
        const size_t srow = static_cast<size_t>(src_width) * channels, drow = static_cast<size_t>(r.width) * channels;
        uint8_t* d_src = ctx.pool.alloc<uint8_t>(srow * src_height);
        uint8_t* d_dst = ctx.pool.alloc<uint8_t>(drow * r.height);
        mst::check_hip(hipMemcpy2DAsync(d_src, srow, src, src_stride, srow, src_height, hipMemcpyHostToDevice, ctx.s),
                       "hipMemcpy2DAsync");
        mst::launch_warp_u8(ctx.d_desc, d_src, srow, channels, interp, border, d_dst, drow, r.width, r.height, ctx.s);
        mst::check_hip(hipGetLastError(), "k_warp_u8");
        mst::check_hip(hipMemcpy2DAsync(dst, dst_stride, d_dst, drow, drow, r.height, hipMemcpyDeviceToHost, ctx.s),
                       "hipMemcpy2DAsync");
        mst::check_hip(hipStreamSynchronize(ctx.s), "hipStreamSynchronize");
    });
}

int mst_cvt_nv12_bgr(int device, const uint8_t* yuv, int width, int height, size_t yuv_stride, uint8_t* bgr, size_t bgr_stride)
{
    return guarded(nullptr, [&] {
        require(yuv && bgr, "null image pointer");
        require(width > 0 && height > 0 && width % 2 == 0 && height % 2 == 0, "NV12 frames have even sizes");
        const size_t row = static_cast<size_t>(width) * 3, rows_in = static_cast<size_t>(height) * 3 / 2;
        require(yuv_stride >= static_cast<size_t>(width) && bgr_stride >= row && row < (size_t{1} << 24), "bad stride");
        mst::select_device(device);
        StageCtx ctx;
        mst::check_hip(hipStreamCreate(&ctx.s), "hipStreamCreate");
        uint8_t* d_src = ctx.pool.alloc<uint8_t>(static_cast<size_t>(width) * rows_in + 16);
        uint8_t* d_dst = ctx.pool.alloc<uint8_t>(row * height + 16);
        mst::check_hip(hipMemcpy2DAsync(d_src, width, yuv, yuv_stride, width, rows_in, hipMemcpyHostToDevice, ctx.s), "hipMemcpy2DAsync");
        const int2 size = make_int2(width, height);
        const int2* d_size = ctx.pool.upload(&size, 1, ctx.s);
        mst::FrameBatch fs{}, fd{};
        fs.f[0].ptr[0] = d_src; fs.f[0].stride[0] = static_cast<uint32_t>(width);
        fd.f[0].ptr[0] = d_dst; fd.f[0].stride[0] = static_cast<uint32_t>(row);
        mst::launch_nv12_to_bgr(fs, fd, d_size, 1, width, height, 1, ctx.s);
        mst::check_hip(hipGetLastError(), "k_nv12_to_bgr");
        mst::check_hip(hipMemcpy2DAsync(bgr, bgr_stride, d_dst, row, row, height, hipMemcpyDeviceToHost, ctx.s), "hipMemcpy2DAsync");
        mst::check_hip(hipStreamSynchronize(ctx.s), "hipStreamSynchronize");
    });
}

int mst_compose_lens_maps(float* xmap, float* ymap, size_t n, const int16_t* map1, const uint16_t* map2, int width, int height)
{
    return guarded(nullptr, [&] {
        require(xmap && ymap && map1 && map2, "null map pointer");
        require(width > 0 && height > 0, "bad lens map size");
        mst::compose_lens_maps(xmap, ymap, n, map1, map2, width, height);
    });
}

int mst_init_undistort_rectify_map(const double K[9], const double* dist, int num_dist, const double R[9],
                                   const double newK[9], int width, int height, int16_t* map1, uint16_t* map2)
{
    return guarded(nullptr, [&] {
        require(K && newK && map1 && map2, "null matrix or map pointer");
        require(width > 0 && height > 0, "bad map size");
        require(num_dist == 0 || dist != nullptr, "null distortion coefficients");
        require(mst::init_undistort_rectify_map(K, dist, num_dist, R, newK, width, height, map1, map2),
                "distortion coefficients must be 0, 4, 5 or 8 values");
    });
}

int mst_remap_fixed(int device, const uint8_t* src, int width, int height, size_t src_stride, const int16_t* map1,
                    const uint16_t* map2, uint8_t* dst, size_t dst_stride)
{
    return guarded(nullptr, [&] {
        require(src && dst && map1 && map2, "null image or map pointer");
        require(width > 0 && height > 0, "bad image size");
        const size_t row = static_cast<size_t>(width) * 3;
        require(src_stride >= row && dst_stride >= row && row < (size_t{1} << 24), "bad stride");
        mst::select_device(device);
        StageCtx ctx;
        mst::check_hip(hipStreamCreate(&ctx.s), "hipStreamCreate");
        const size_t px = static_cast<size_t>(width) * height;
        mst::UndistortCam u{ctx.pool.upload(map1, px * 2, ctx.s), ctx.pool.upload(map2, px, ctx.s), width, height};
        mst::UndistortCam* d_u = ctx.pool.upload(&u, 1, ctx.s);
        uint8_t* d_src = ctx.pool.alloc<uint8_t>(row * height);
        uint8_t* d_dst = ctx.pool.alloc<uint8_t>(row * height);
        mst::check_hip(hipMemcpy2DAsync(d_src, row, src, src_stride, row, height, hipMemcpyHostToDevice, ctx.s),
                       "hipMemcpy2DAsync");
        mst::FrameBatch fs{}, fd{};
        fs.f[0].ptr[0] = d_src; fs.f[0].stride[0] = static_cast<uint32_t>(row);
        fd.f[0].ptr[0] = d_dst; fd.f[0].stride[0] = static_cast<uint32_t>(row);
        mst::launch_remap_fixed(d_u, 1, width, height, fs, fd, 1, ctx.s);
        mst::check_hip(hipGetLastError(), "k_remap_fixed");
        mst::check_hip(hipMemcpy2DAsync(dst, dst_stride, d_dst, row, row, height, hipMemcpyDeviceToHost, ctx.s),
                       "hipMemcpy2DAsync");
        mst::check_hip(hipStreamSynchronize(ctx.s), "hipStreamSynchronize");
    });
}

int mst_exposure_apply(int device, int expos_kind, uint8_t* image, int width, int height, size_t stride,
                       const float* gain_map, int gain_w, int gain_h, float gain)
{
    return guarded(nullptr, [&] {
        require(image != nullptr, "null image");
        require(width > 0 && height > 0 && stride >= static_cast<size_t>(width) * 3, "bad CV_8UC3 image");
        require(expos_kind >= MST_EXPOS_NO && expos_kind <= MST_EXPOS_GAIN_BLOCKS, "unsupported exposure compensation method");
        if (expos_kind == MST_EXPOS_NO) return;      // NoExposureCompensator::apply
        mst::select_device(device);
        StageCtx ctx;
        mst::check_hip(hipStreamCreate(&ctx.s), "hipStreamCreate");
        // identity projector: only the gain fields of the descriptor are used by the kernel
        const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        mst::Projector p;
        p.kind = MST_WARP_PLANE; p.scale = 1.f;
        p.set(I, I, nullptr);
        mst::Rect r; r.x = 0; r.y = 0; r.width = width; r.height = height;
        mst::fill_camera_desc(ctx.pool, ctx.s, p, r, width, height, expos_kind, gain_map, gain_w, gain_h, gain, ctx.desc);
        ctx.d_desc = ctx.pool.upload(&ctx.desc, 1, ctx.s);
        const size_t row = static_cast<size_t>(width) * 3;
        uint8_t* d_img = ctx.pool.alloc<uint8_t>(row * height);
        mst::check_hip(hipMemcpy2DAsync(d_img, row, image, stride, row, height, hipMemcpyHostToDevice, ctx.s), "hipMemcpy2DAsync");
        mst::launch_exposure_apply(ctx.d_desc, d_img, row, width, height, ctx.s);
        mst::check_hip(hipGetLastError(), "k_exposure_apply");
        mst::check_hip(hipMemcpy2DAsync(image, stride, d_img, row, row, height, hipMemcpyDeviceToHost, ctx.s), "hipMemcpy2DAsync");
        mst::check_hip(hipStreamSynchronize(ctx.s), "hipStreamSynchronize");
    });
}

int mst_seam_voronoi(int device, int num_images, const int32_t* corners_xy, const int32_t* sizes_wh, uint8_t* const* masks,
                     const size_t* mask_strides)
{
    return guarded(nullptr, [&] {
        require(num_images >= 1 && num_images <= MST_MAX_CAMERAS, "num_images out of range");
        require(corners_xy && sizes_wh && masks && mask_strides, "null argument");
        mst::voronoi_seams(device, num_images, corners_xy, sizes_wh, masks, mask_strides);
    });
}

int mst_seam_dp(int device, int num_images, const int32_t* corners_xy, const int32_t* sizes_wh, const void* const* images,
                const size_t* image_strides, int image_type, uint8_t* const* masks, const size_t* mask_strides, int cost_function)
{
    return guarded(nullptr, [&] {
        require(num_images >= 1 && num_images <= MST_MAX_CAMERAS, "num_images out of range");
        require(corners_xy && sizes_wh && images && image_strides && masks && mask_strides, "null argument");
        require(image_type == MST_IMAGE_8UC3 || image_type == MST_IMAGE_32FC3, "images must be CV_8UC3 or CV_32FC3 (seam_finders.cpp:709-720)");
        require(cost_function == MST_SEAM_DP_COLOR || cost_function == MST_SEAM_DP_COLOR_GRAD, "unknown cost function");
        mst::dp_seams(device, num_images, corners_xy, sizes_wh, images, image_strides, image_type == MST_IMAGE_32FC3 ? 1 : 0, masks,
                      mask_strides, cost_function);
    });
}

int mst_gain_map_size(int image_width, int image_height, int block_width, int block_height, int* gain_w, int* gain_h)
{
    return guarded(nullptr, [&] {
        require(image_width > 0 && image_height > 0 && block_width > 0 && block_height > 0 && gain_w && gain_h, "bad sizes");
        mst::gain_map_size(image_width, image_height, block_width, block_height, gain_w, gain_h);
    });
}

int mst_exposure_feed(int device, int expos_kind, int num_images, const int32_t* corners_xy, const uint8_t* const* images,
                      const size_t* image_strides, const int32_t* sizes_wh, const uint8_t* const* masks,
                      const size_t* mask_strides, int block_width, int block_height, double* gains, float* const* gain_maps)
{
    return guarded(nullptr, [&] {
        require(expos_kind == MST_EXPOS_GAIN || expos_kind == MST_EXPOS_GAIN_BLOCKS, "feed exists for GAIN and GAIN_BLOCKS");
        require(num_images >= 1 && num_images <= MST_MAX_CAMERAS, "num_images out of range");
        require(corners_xy && images && image_strides && sizes_wh && masks && mask_strides, "null argument");
        require(expos_kind == MST_EXPOS_GAIN ? gains != nullptr : gain_maps != nullptr, "null output");
        require(expos_kind == MST_EXPOS_GAIN || (block_width > 0 && block_height > 0), "bad block size");
        mst::exposure_feed(device, expos_kind, num_images, corners_xy, images, image_strides, sizes_wh, masks, mask_strides,
                           block_width, block_height, gains, gain_maps);
    });
}

int mst_multiband_geometry(mst_rect canvas, int num_bands, mst_rect image, mst_rect* padded_canvas, int* used_bands,
                           int out[8])
{
    return guarded(nullptr, [&] {
        require(canvas.width > 0 && canvas.height > 0 && image.width > 0 && image.height > 0, "empty rectangle");
        require(num_bands >= 0 && num_bands <= MST_MAX_BANDS, "bad band count");
        mst::Rect c; c.x = canvas.x; c.y = canvas.y; c.width = canvas.width; c.height = canvas.height;
        mst::Rect im; im.x = image.x; im.y = image.y; im.width = image.width; im.height = image.height;
        const int nb = mst::multiband_num_bands(num_bands, c);
        mst::Rect pad = mst::multiband_padded_roi(c, nb);
        if (padded_canvas) { padded_canvas->x = pad.x; padded_canvas->y = pad.y; padded_canvas->width = pad.width; padded_canvas->height = pad.height; }
        if (used_bands) *used_bands = nb;
        if (out) {
            mst::BandGeometry g = mst::multiband_feed_geometry(pad, nb, im);
            out[0] = g.tl_br.x; out[1] = g.tl_br.y; out[2] = g.tl_br.x + g.tl_br.width; out[3] = g.tl_br.y + g.tl_br.height;
            out[4] = g.top; out[5] = g.left; out[6] = g.bottom; out[7] = g.right;
        }
    });
}

int mst_build_info(mst_build* out)
{
    if (!out || out->struct_size != static_cast<int32_t>(sizeof(mst_build))) return MST_ERR_INVALID_ARG;
    out->version = MST_VERSION;
    out->timing_build = mst::mb_build_ablations() | mst::compose_build_ablations();
    out->level0_packed = MST_L0_PACKED;
    std::snprintf(out->description, sizeof(out->description), "mi355-stitch %d gfx950%s%s%s", MST_VERSION,
                  MST_L0_PACKED ? " l0=bgr8" : " l0=dword",
                  (out->timing_build & 15u) ? " TIMING-BUILD(-DMST_ABLATE_*: wrong pixels)" : "",
                  (out->timing_build & 16u) ? " STAMPS" : "");
    return MST_OK;
}

// Diagnostic builds only (make CXXFLAGS+=-DMST_STAMPS): in-kernel phase stamps of the last launches of kernels A (0)
// and C (1), 16 slots per block.  Not declared in the public header; tools/stamps.py binds it by name.
int mst_debug_stamps(int which, unsigned long long* out, size_t n)
{
    return mst::mb_read_stamps(which, out, n);
}

// ---- synthetic frames ------------------------------------------------------------------------------

int mst_synth_frame(int camera, int frame_index, int width, int height, uint8_t* dst, size_t stride)
{
    return guarded(nullptr, [&] {
        require(dst && width > 0 && height > 0 && stride >= static_cast<size_t>(width) * 3, "bad frame buffer");
        // PCG32 (O'Neill, pcg32_srandom_r / pcg32_random_r): state seeded per camera, stream per frame
        uint64_t state = 0u, inc = (static_cast<uint64_t>(static_cast<uint32_t>(frame_index)) << 1u) | 1u;
        auto next = [&]() -> uint32_t {
            uint64_t old = state;
            state = old * 6364136223846793005ULL + inc;
            uint32_t xorshifted = static_cast<uint32_t>(((old >> 18u) ^ old) >> 27u);
            uint32_t rot = static_cast<uint32_t>(old >> 59u);
            return (xorshifted >> rot) | (xorshifted << ((32u - rot) & 31u));
        };
        next();
        state += 0x5717C4ULL + static_cast<uint64_t>(camera);
        next();
        const double two_pi = 6.283185307179586476925286766559;
        std::vector<double> sx(width), cy(height);
        for (int x = 0; x < width; ++x) sx[x] = std::sin(two_pi * (x / 97.0 + camera / 7.0));
        for (int y = 0; y < height; ++y) cy[y] = std::cos(two_pi * y / 61.0);
        for (int y = 0; y < height; ++y) {
            uint8_t* row = dst + static_cast<size_t>(y) * stride;
            for (int x = 0; x < width; ++x) {
                const double base = 128.0 + 80.0 * sx[x] * cy[y];
                for (int c = 0; c < 3; ++c) {
                    const int noise = static_cast<int>((static_cast<uint64_t>(next()) * 33u) >> 32) - 16;
                    double v = base + 20.0 * c + noise;
                    v = v < 0.0 ? 0.0 : (v > 255.0 ? 255.0 : v);
                    row[3 * x + c] = static_cast<uint8_t>(std::lrint(v));
                }
            }
        }
    });
}

}  // extern "C"
