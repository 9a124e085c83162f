// engine.hpp -- host side of the compositing engine: calibration state (double-buffered),
// device memory ownership, and the per-frame launch sequence.
//
// Mirrors the reference's cached-calibration design: GetPreStruct() produces a state,
// UpdateMat() swaps it in between frames (R64/stitching_detailed.cpp:696-722, 835-836), and
// StitchingAll() only runs per-frame work against it (R64/stitching_detailed.cpp:724-770).
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/mi355_stitch.h"

struct mst_shm;          // a shared-memory frame client (include/mi355_ingest.h)
#include "device_types.hpp"
#include "geometry.hpp"
#include "mb_plan.hpp"

namespace mst {

struct Error {
    int code;
    std::string msg;
};

// Owns every device allocation of one calibration state (or one stage-level call).
// Allocations are carved out of a few large slabs (bump pointer, 256-byte aligned): a calibration state holds
// several hundred buffers, and one hipMalloc / hipFree each is what used to dominate mst_update and the
// teardown of a retired state.
// Standard-size slabs of retired calibration states, kept for the next mst_update (a recalibration every 30 s,
// DRV:842-847, would otherwise hipMalloc and hipFree ~1.5 GB each time).  Shared by an engine and its states.
class SlabCache {
public:
    ~SlabCache();
    void* take();                 // nullptr when empty
    void give(void* slab);
private:
    std::mutex mu_;
    std::vector<void*> free_;
};

class DevicePool {
public:
    DevicePool() = default;
    DevicePool(const DevicePool&) = delete;
    DevicePool& operator=(const DevicePool&) = delete;
    ~DevicePool();
    void* alloc_bytes(size_t bytes);
    template <typename T> T* alloc(size_t n) { return static_cast<T*>(alloc_bytes(n * sizeof(T))); }
    template <typename T> T* upload(const T* host, size_t n, hipStream_t s);
    template <typename T> T* upload(const std::vector<T>& v, hipStream_t s) { return upload(v.data(), v.size(), s); }
    size_t bytes() const { return total_; }
    void set_cache(std::shared_ptr<SlabCache> cache) { cache_ = std::move(cache); }
private:
    std::shared_ptr<SlabCache> cache_;
    std::vector<void*> big_;        // allocations larger than a slab: always freed
    std::vector<void*> ptrs_;       // slabs
    char* cur_ = nullptr;           // bump pointer inside the newest slab
    size_t left_ = 0;
    size_t total_ = 0;
};

// Host-side description of one camera, enough to fill a CamDesc.
struct CameraSetup {
    Projector proj;
    Rect roi;
    int src_w = 0, src_h = 0;
};

// the lens-undistort pre-pass of one camera in the staged form (kernels_compose.hip); staged == false: k_remap_fixed
struct UndistortPlan {
    bool staged = false;
    ComposeArgs args{};
    size_t lds_bytes = 0;
};

struct CalibState {
    DevicePool pool;
    int n = 0;
    int blend_kind = 0;
    std::vector<CameraSetup> cams;
    std::vector<CamDesc> desc;      // host copy of the device array
    CamDesc* d_desc = nullptr;      // num_sets copies back to back: same cameras, own Gaussian pyramids per set
    int num_sets = 1;               // a batched launch uses set f for frame set f; frame pipelining alternates
    int sets_per_slot = 1;          // between two runs of sets_per_slot sets
    std::vector<uint32_t*> pyr_base;   // per camera: the contiguous Gaussian pyramid block (first set)
    std::vector<size_t> pyr_bytes;
    Rect canvas;                    // resultRoi(corners, sizes)
    // multi-band
    int num_bands = 0;
    Rect padded;
    CanvasDesc cv{};
    MbPlanHost plan;                    // work lists of the fused kernels (host copy, for introspection)
    TileList warp_list{};               // kernel A
    // lens-undistort pre-pass (optional per camera)
    std::vector<UndistortCam> ucams;    // host copy (device map pointers)
    UndistortCam* d_ucams = nullptr;
    std::vector<UndistortPlan> undist_plans;
    bool any_undistort = false;
    int undist_max_w = 0, undist_max_h = 0;
    struct CoarseDev { const uint32_t* cams = nullptr; const uint32_t* pure = nullptr; const uint32_t* order = nullptr; const uint32_t* order2 = nullptr; };
    CoarseDev coarse_dev[2];               // k_blend_coarse tile tables of plan.coarse[v] on the device
    // `blend` completed for a launch of `frame_sets` frame sets: the tile tables / LDS layout of its launch variant
    BlendArgs blend_for(int frame_sets, int* n_coarse, size_t* lds_bytes, bool busy = false) const;
    std::vector<uint64_t*> c1;               // per buffer set: collapsed level-1 canvas (k_blend_coarse / k_blend_mid -> k_blend_fine)
    std::vector<uint64_t*> c2;               // per buffer set: collapsed level-2 canvas (k_blend_coarse -> k_blend_mid), three and more levels
    // MST_FLAG_FRAMES_NV12: num_sets x n engine-owned BGR frames the callers' NV12 frames are converted into, (w, h) per camera
    std::vector<uint8_t*> nv12_frames;
    const int2* d_frame_sizes = nullptr;
    int frame_max_w = 0, frame_max_h = 0;
    std::vector<uint8_t*> undist_frames;   // num_sets x n engine-owned undistorted frames (nullptr: camera has no maps)
    uint32_t* warp_table = nullptr;     // per-tile backward-map table (frame-invariant), see kernels_mb.hip
    uint32_t* warp_blobs = nullptr;     // per tile: calibration constants of kernel A (gain rows, row weights, table origin)
    // the staged form of kernel A (k_warp_pyr_lds): map + gain tables in its lane order, the tiles' source windows, the
    // tile indices it takes and those left to the gathering kernel, both grouped by camera
    uint32_t* warp_table_lds = nullptr; float* warp_gain_lds = nullptr; WarpWin* warp_wins = nullptr;
    struct WarpClass { const WarpWin* d_wins = nullptr; std::vector<int> start; size_t lds_bytes = 0; };    // one staged launch
    std::vector<WarpClass> warp_classes;                        // by window size; start: n + 1 offsets into d_wins (launch order)
    const uint32_t* d_warp_gathered = nullptr;
    std::vector<int> warp_gathered_start;                       // n + 1 offsets
    std::vector<TileList> chain_lists;  // kernel B, one per plan.chain step
    struct StreamUnits { const uint32_t* d[2] = {nullptr, nullptr}; };
    std::vector<StreamUnits> chain_units;   // per plan.chain step: the streaming kernel's unit lists (null: k_pyr_chain runs the step)
    PayloadRect* d_payload_rects = nullptr;   // camera-sharded hand-off (plan.payload_*)
    BlendArgs blend{};                  // canvas kernels (pano pointers filled per frame)
    // feather
    float* d_feather_ws = nullptr;
    // single-pass composites (no blend / driver LUT / feather): tile and segment tables, see kernels_compose.hip
    ComposeArgs compose{};
    int compose_segments = 0;
    const ComposeTile* d_compose_staged = nullptr; int compose_staged = 0, compose_staged_gatherable = 0;     // tiles of the LDS-staged launch
    const ComposeTile* d_compose_direct = nullptr; int compose_direct = 0;     // tiles that gather from memory
    const ComposeTile* d_compose_general = nullptr; int compose_general = 0;   // tiles on the per-pixel path
    size_t compose_lds_bytes = 0;
    // output rectangle (== canvas unless the driver LUT crop is active)
    int crop_x = 0, crop_y = 0, out_w = 0, out_h = 0;
    int strip_x0 = 0, strip_x1 = 0;     // the output columns this engine writes (canvas split over GPUs), output coordinates
    uint64_t generation = 0;            // 1, 2, ... in mst_update order (mst_acquire_calibration)
    ~CalibState();
};

// Records a HIP event on the compose stream between consecutive kernel launches of one frame.
class StageTimer {
public:
    ~StageTimer();
    void begin(hipStream_t s);
    void mark(const char* name, hipStream_t s);
    void collect(std::vector<std::pair<std::string, float>>& out);   // waits for the last event
private:
    std::vector<hipEvent_t> events_;
    std::vector<const char*> names_;
    size_t used_ = 0;
};

class Engine {
public:
    explicit Engine(const mst_config& cfg);
    ~Engine();
    void init();                                       // throws Error
    void update(const mst_calibration& calib);         // throws Error
    void compose_device(const void* const* d_frames, const size_t* strides, void* d_pano, size_t pano_stride,
                        void* d_mask, size_t mask_stride, hipStream_t stream, StageTimer* timer = nullptr);
    // `count` frame sets (frame set f's camera i at index f * n + i) -> `count` panoramas, one launch sequence
    // frames_ready / outputs_free (mst_compose_device_batch_ex): events the call's first kernel waits for, on whichever stream
    // the engine runs it
    void compose_batch(int count, const void* const* d_frames, const size_t* strides, void* const* d_panos,
                       size_t pano_stride, void* const* d_masks, size_t mask_stride, hipStream_t stream,
                       StageTimer* timer = nullptr, hipEvent_t frames_ready = nullptr, hipEvent_t outputs_free = nullptr);
    void compose_host(const uint8_t* const* frames, const size_t* strides, uint8_t* pano, size_t pano_stride,
                      uint8_t* pano_mask, size_t mask_stride);
    void stage_cameras(uint32_t camera_mask, const void* const* d_frames, const size_t* strides, hipStream_t stream);
    void stage_canvas(void* d_pano, size_t pano_stride, void* d_mask, size_t mask_stride, hipStream_t stream);
    void pyramid_buffer(int camera, void** d_ptr, size_t* bytes);
    // camera-sharded hand-off: the packed tile interiors of one camera (see mb_plan.hpp)
    size_t payload_bytes(int camera);
    void payload_copy(int camera, void* d_buf, bool pack, hipStream_t stream);
    void select_stage_set(int set);                  // which pyramid set the mst_stage_* / mst_payload_* calls use
    void camera_mask(int camera, uint8_t* mask, size_t stride);
    // Host-buffer frame loop as a three-stage pipeline: H2D of frame t+1, kernels of frame t and D2H
    // of frame t-1 run concurrently on three streams over a ring of kPipeDepth device frame sets / panoramas.
    // submit() returns at once (it blocks only while the ring is full); wait() blocks until the oldest submitted
    // frame's panorama is complete in its host buffer and returns that buffer.  The host buffers should be
    // page-locked (host_alloc) -- pageable memory works but serialises the copies -- and must stay untouched between
    // submit and the matching wait.
    static constexpr int kPipeDepth = 3;
    void pipeline_submit(const uint8_t* const* frames, const size_t* strides, uint8_t* pano, size_t pano_stride,
                         ::mst_shm* const* clients = nullptr, int timeout_ms = -1);
    uint8_t* pipeline_wait(mst_rect* canvas);
    int pipeline_pending() const { return pipe_submitted_ - pipe_completed_; }
    void synchronize();
    // The calibration the next compose call will use: the newest published one, or -- once the frame thread has
    // called acquire() -- the one it latched there (the reference's `if (Upflag) UpdateMat()`, DRV:835-836).
    std::shared_ptr<CalibState> state() const;
    std::shared_ptr<CalibState> acquire();             // latch the newest calibration for the following composes
    void release();                                    // stop latching: every call picks up the newest calibration again
    void debug_read(int what, int camera, int level, void* buf, size_t buf_bytes, int* w, int* h);

    const mst_config& config() const { return cfg_; }

private:
    void compose_batch_on(std::shared_ptr<CalibState> st, int count, const void* const* d_frames, const size_t* strides,
                          void* const* d_panos, size_t pano_stride, void* const* d_masks, size_t mask_stride,
                          hipStream_t stream, StageTimer* timer, hipEvent_t frames_ready = nullptr, hipEvent_t outputs_free = nullptr);
    void strip_columns(int width, int* x0, int* x1, int index = -1) const;
    void launch_undistort(const CalibState& st, const FrameBatch& src, const FrameBatch& dst, int count, uint32_t camera_mask, hipStream_t s);
    // which map_px<KIND> the table kernels and the gathering warp kernel are launched with: the projection's own
    // per-axis form, or "read the map" (3) for the non-separable projectors and for MST_FLAG_LENS_COMPOSED
    int map_kind() const { return (cfg_.flags & MST_FLAG_LENS_COMPOSED) ? 3 : (cfg_.warp_kind & 0xFF); }
    void launch_warp(const CalibState& st, const CamDesc* desc, const FrameBatch& fb, int count, int cam0, int cam1, hipStream_t s,
                     StageTimer* timer = nullptr);
    void build_compose_plan(CalibState& st, hipStream_t s);     // mst_update, blend kinds NO and FEATHER
    void set_in_flight(std::shared_ptr<CalibState> st);   // keeps `st` alive; the previous one goes to retired_
    void drop_retired();                                   // calibration thread / destructor: tear old states down
    // MST_FLAG_GRAPH_REPLAY: launch sequences captured per (calibration generation, buffers) and replayed
    struct GraphEntry { std::vector<uint64_t> key; hipGraphExec_t exec = nullptr; hipStream_t last_stream = nullptr; };
    std::vector<GraphEntry> graphs_;                     // least recently used first
    struct RetiredGraph { hipGraphExec_t exec = nullptr; hipEvent_t done = nullptr; };
    std::vector<RetiredGraph> retired_graphs_;           // left the cache; destroyed once `done` (recorded on their launch stream) has completed
    static constexpr size_t kGraphCacheEntries = 32;
    uint64_t graph_misses_ = 0;
    bool graph_thrash_warned_ = false;
    void drop_graphs();
    void reap_graphs();
    mst_config cfg_;
    size_t max_lds_bytes_ = 0;            // hipDeviceAttributeMaxSharedMemoryPerBlock of cfg_.device
    hipStream_t stream_ = nullptr;        // compose stream
    hipStream_t calib_stream_ = nullptr;  // mst_update builds the next state here
    // Frame pipelining (MST_FLAG_OVERLAP_FRAMES): the per-camera kernels of frame t+1 run on front_stream_
    // while the canvas kernel of frame t runs on the caller's stream; two pyramid sets alternate.
    hipStream_t front_stream_ = nullptr;
    static constexpr int kMaxOverlap = 8;
    hipEvent_t ev_front_done_[kMaxOverlap] = {};     // per-camera work of the slot finished
    hipEvent_t ev_canvas_done_[kMaxOverlap] = {};    // canvas kernel that read the slot finished
    bool canvas_pending_[kMaxOverlap] = {};
    // MST_OVERLAP_DEPTH = 3 or 4 (experiment): whole calls on their own streams, round-robin, one pyramid slot each
    int overlap_depth_ = 2;
    hipStream_t call_streams_[kMaxOverlap] = {};
    struct ByteRange { uintptr_t lo, hi; };                // [lo, hi)
    std::vector<ByteRange> slot_outputs_[kMaxOverlap];     // panorama / mask bytes the slot's latest call writes
    // a deep call on `slot` about to write `outs`: wait for every other slot whose latest call writes bytes that overlap them
    void order_after_writers(int slot, hipStream_t s, const std::vector<ByteRange>& outs);
    bool in_host_compose_ = false;
    uint64_t frame_counter_ = 0;
    int stage_set_ = 0;
    mutable std::mutex mu_;
    std::shared_ptr<CalibState> current_;
    std::shared_ptr<CalibState> latched_;     // acquire(): what compose uses while latching_ (else current_)
    bool latching_ = false;
    uint64_t next_generation_ = 0;
    std::shared_ptr<CalibState> in_flight_;   // state used by the most recent compose
    // States no compose will start on any more.  Their teardown (device sync + frees) must not run on the frame
    // thread: they are parked here by pointer move and destroyed by the next mst_update or by the destructor.
    std::vector<std::shared_ptr<CalibState>> retired_;
    std::shared_ptr<SlabCache> slab_cache_ = std::make_shared<SlabCache>();
    // staging for compose_host
    std::vector<uint8_t*> d_frames_;
    std::vector<size_t> d_frame_bytes_;
    uint8_t* d_pano_ = nullptr; size_t d_pano_bytes_ = 0;
    uint8_t* d_mask_ = nullptr; size_t d_mask_bytes_ = 0;
    // pipeline_submit / pipeline_wait state (frame thread only)
    struct PipeSlot {
        std::vector<uint8_t*> d_frames;
        std::vector<size_t> d_frame_bytes;
        uint8_t* d_pano = nullptr; size_t d_pano_bytes = 0;
        hipEvent_t h2d_done = nullptr, compute_done = nullptr, d2h_done = nullptr;
        uint8_t* host_pano = nullptr;
        mst_rect canvas{};
        bool busy = false;
    };
    PipeSlot pipe_[kPipeDepth];
    hipStream_t pipe_in_ = nullptr, pipe_out_ = nullptr;
    long pipe_submitted_ = 0, pipe_completed_ = 0;
    // mst_update: page-locked landing zone of the weight-class bytes (calibration thread only)
    uint8_t* cls_host_ = nullptr; size_t cls_host_bytes_ = 0;
};

// Fill the warp + exposure part of a CamDesc (tables uploaded through `pool`).
void fill_camera_desc(DevicePool& pool, hipStream_t s, const Projector& proj, const Rect& roi, int src_w, int src_h,
                      int expos_kind, const float* gain_map, int gw, int gh, float gain, CamDesc& d);

// exposure-gain estimation (exposure_feed.cpp): GAIN -> gains[n]; GAIN_BLOCKS -> gain_maps[i] of gain_map_size floats
void gain_map_size(int w, int h, int bl_w, int bl_h, int* gw, int* gh);
void exposure_feed(int device, int expos_kind, int n, const int32_t* corners_xy, const uint8_t* const* images,
                   const size_t* image_strides, const int32_t* sizes_wh, const uint8_t* const* masks,
                   const size_t* mask_strides, int bl_w, int bl_h, double* gains, float* const* gain_maps);

// VoronoiSeamFinder::find on the device (seams.cpp): masks (host, CV_8U of sizes_wh) updated in place
void voronoi_seams(int device, int n, const int32_t* corners_xy, const int32_t* sizes_wh, uint8_t* const* masks,
                   const size_t* mask_strides);

// DpSeamFinder::find (dp_seam.hip): images CV_8UC3 or CV_32FC3 (host), masks (host, CV_8U) updated in place; cost_func 0 COLOR, 1 COLOR_GRAD
void dp_seams(int device, int n, const int32_t* corners_xy, const int32_t* sizes_wh, const void* const* images, const size_t* image_strides,
              int image_is_f32, uint8_t* const* masks, const size_t* mask_strides, int cost_func);

void check_hip(hipError_t e, const char* what);
void select_device(int device);

}  // namespace mst
