// frame_loop_mi355 -- the driver's frame loop (R64/stitching_detailed.cpp:724-770: N frames in, one panorama out, again
// and again) as a C++ host on the C ABI, frames resident in HBM: what a call costs the HOST to enqueue, what a lone call
// takes, and what `depth` calls in flight deliver (mst_config.overlap_depth).  bench.py measures the same through
// Python / ctypes, where the enqueue cost (~ 23 us per call) caps the small rigs; this is the figure a C++ integrator gets.
//
//   frame_loop_mi355 <calibration file> <warp> <blend> <expos> [--depth 4] [--batch 1] [--calls 2000] [--lut]
//
// The calibration file is what mst_calibration_save writes (stitchingvideo_amd.engine.save_calibration, or
// stitching_detailed_mi355 --save_calibration); frame sizes come from it, frames are mst_synth_frame's.
// Build: make -C stitchingvideo_amd/csrc loop     (hipcc: the tool allocates device memory itself)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/mi355_stitch.h"

#define CHECK(call)                                                                                      \
    do {                                                                                                 \
        const int rc_ = (call);                                                                          \
        if (rc_ != MST_OK) { std::printf("%s failed: %d (%s)\n", #call, rc_, mst_last_error(engine)); return 1; } \
    } while (0)
#define HIP(call)                                                                                        \
    do {                                                                                                 \
        const hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) { std::printf("%s failed: %s\n", #call, hipGetErrorString(e_)); return 1; } \
    } while (0)

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv)
{
    mst_engine* engine = nullptr;
    if (argc < 5) {
        std::printf("usage: frame_loop_mi355 <calibration file> <warp> <blend no|feather|multiband> <expos no|gain|gain_blocks> "
                    "[--depth D] [--batch B] [--calls N] [--lut]\n");
        return 2;
    }
    int depth = 4, batch = 1, calls = 2000;
    bool lut = false;
    for (int i = 5; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--depth" && i + 1 < argc) depth = std::atoi(argv[++i]);
        else if (a == "--batch" && i + 1 < argc) batch = std::atoi(argv[++i]);
        else if (a == "--calls" && i + 1 < argc) calls = std::atoi(argv[++i]);
        else if (a == "--lut") lut = true;
        else { std::printf("bad flag %s\n", a.c_str()); return 2; }
    }
    // the calibration blob (8-byte aligned storage: mst_calibration_load points into it)
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) { std::printf("cannot open %s\n", argv[1]); return 1; }
    std::fseek(f, 0, SEEK_END);
    const long blob_bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> blob((blob_bytes + 7) / 8);
    if (std::fread(blob.data(), 1, blob_bytes, f) != static_cast<size_t>(blob_bytes)) { std::printf("short read\n"); return 1; }
    std::fclose(f);
    mst_calibration cal;
    mst_camera cams[MST_MAX_CAMERAS];
    std::memset(&cal, 0, sizeof cal);
    CHECK(mst_calibration_load(blob.data(), static_cast<size_t>(blob_bytes), &cal, cams, MST_MAX_CAMERAS));
    const int n = cal.num_cameras;

    const std::string blend = argv[3], expos = argv[4];
    auto make = [&](int in_flight, mst_engine** out) {
        mst_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.num_cameras = n;
        cfg.warp_kind = mst_warp_kind_from_name(argv[2]);
        cfg.blend_kind = blend == "multiband" ? MST_BLEND_MULTI_BAND : blend == "feather" ? MST_BLEND_FEATHER : MST_BLEND_NO;
        cfg.expos_kind = expos == "gain_blocks" ? MST_EXPOS_GAIN_BLOCKS : expos == "gain" ? MST_EXPOS_GAIN : MST_EXPOS_NO;
        cfg.num_bands = 5;
        cfg.max_batch = batch;
        cfg.flags = (lut ? MST_FLAG_DRIVER_LUT : 0u) | (in_flight > 1 ? MST_FLAG_OVERLAP_FRAMES : 0u);
        cfg.overlap_depth = in_flight > 1 ? in_flight : 0;
        cfg.upblack = cfg.downblack = 0.1f; cfg.leftblack = cfg.rightblack = 10.f;       // DRV:47
        return mst_create(&cfg, out);
    };
    mst_engine* lone = nullptr;
    mst_engine* deep = nullptr;
    if (make(1, &lone) != MST_OK || make(depth, &deep) != MST_OK) { std::printf("mst_create: %s\n", mst_last_error(nullptr)); return 1; }
    engine = lone;
    CHECK(mst_update(lone, &cal));
    engine = deep;
    CHECK(mst_update(deep, &cal));
    mst_rect canvas;
    uint64_t gen;
    CHECK(mst_acquire_calibration(deep, &canvas, &gen));
    engine = lone;
    CHECK(mst_acquire_calibration(lone, &canvas, &gen));

    // three distinct frame sets in HBM, a ring of panoramas (calls in flight write different buffers)
    const int kSets = 3;
    std::vector<void*> d_frames(static_cast<size_t>(kSets) * n);
    std::vector<size_t> strides(static_cast<size_t>(batch) * n);
    for (int s = 0; s < kSets; ++s)
        for (int i = 0; i < n; ++i) {
            const int w = cams[i].src_width, h = cams[i].src_height;
            std::vector<uint8_t> host(static_cast<size_t>(w) * h * 3);
            CHECK(mst_synth_frame(i, s, w, h, host.data(), static_cast<size_t>(w) * 3));
            HIP(hipMalloc(&d_frames[static_cast<size_t>(s) * n + i], host.size()));
            HIP(hipMemcpy(d_frames[static_cast<size_t>(s) * n + i], host.data(), host.size(), hipMemcpyHostToDevice));
        }
    for (int b = 0; b < batch; ++b)
        for (int i = 0; i < n; ++i) strides[static_cast<size_t>(b) * n + i] = static_cast<size_t>(cams[i].src_width) * 3;
    const size_t pano_stride = static_cast<size_t>(canvas.width) * 3, pano_bytes = pano_stride * canvas.height;
    const int ring = depth * batch;
    std::vector<void*> d_panos(ring);
    for (int k = 0; k < ring; ++k) HIP(hipMalloc(&d_panos[k], pano_bytes));
    hipStream_t stream;
    HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));

    std::vector<const void*> ftab(static_cast<size_t>(batch) * n);
    auto call = [&](mst_engine* e, int k) {
        for (int b = 0; b < batch; ++b)
            for (int i = 0; i < n; ++i) ftab[static_cast<size_t>(b) * n + i] = d_frames[static_cast<size_t>((k + b) % kSets) * n + i];
        void* const* panos = &d_panos[static_cast<size_t>(k % depth) * batch];
        return mst_compose_device_batch(e, batch, ftab.data(), strides.data(), panos, pano_stride, nullptr, 0, stream);
    };
    const double alg_bytes = [&] {
        double b = 3.0 * canvas.width * canvas.height;
        for (int i = 0; i < n; ++i) b += 3.0 * cams[i].src_width * cams[i].src_height;
        return b;
    }();

    // lone calls: a synchronize after each
    engine = lone;
    for (int k = 0; k < 20; ++k) CHECK(call(lone, k));
    HIP(hipStreamSynchronize(stream));
    double t0 = now_s();
    for (int k = 0; k < calls / 4; ++k) { CHECK(call(lone, k)); HIP(hipStreamSynchronize(stream)); }
    const double t_lone = (now_s() - t0) / (calls / 4);
    // the same engine, back to back
    t0 = now_s();
    for (int k = 0; k < calls; ++k) CHECK(call(lone, k));
    const double t_enq_lone = (now_s() - t0) / calls;
    HIP(hipStreamSynchronize(stream));
    const double t_serial = (now_s() - t0) / calls;
    // `depth` calls in flight
    engine = deep;
    for (int k = 0; k < 4 * depth; ++k) CHECK(call(deep, k));
    HIP(hipStreamSynchronize(stream));
    t0 = now_s();
    for (int k = 0; k < calls; ++k) CHECK(call(deep, k));
    const double t_enq_deep = (now_s() - t0) / calls;
    HIP(hipStreamSynchronize(stream));
    const double t_deep = (now_s() - t0) / calls;

    std::printf("{\"cameras\": %d, \"canvas\": [%d, %d], \"blend\": \"%s\", \"frame_sets_per_call\": %d, \"calls_in_flight\": %d, "
                "\"lone_call_us\": %.1f, \"back_to_back_us_per_call\": %.1f, \"in_flight_us_per_call\": %.1f, "
                "\"host_enqueue_us_per_call\": %.2f, \"host_enqueue_us_per_call_in_flight\": %.2f, "
                "\"panoramas_per_s_back_to_back\": %.0f, \"panoramas_per_s_in_flight\": %.0f, \"roofline_frac_in_flight\": %.4f}\n",
                n, canvas.width, canvas.height, blend.c_str(), batch, depth, t_lone * 1e6, t_serial * 1e6, t_deep * 1e6, t_enq_lone * 1e6,
                t_enq_deep * 1e6, batch / t_serial, batch / t_deep, alg_bytes * batch / t_deep / 8.0e12);
    mst_destroy(lone);
    mst_destroy(deep);
    return 0;
}
