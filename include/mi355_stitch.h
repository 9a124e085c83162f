/*
 * mi355_stitch.h -- C ABI of the MI355X-native per-frame panorama compositing engine
 * (libmi355stitch.so).  Plain pointers and sizes only: no C++ types, no OpenCV types,
 * no torch types.  This is what the reference's VC++ host binds instead of its own
 * per-frame loop; INTEGRATION.md shows the reference-side shim.
 *
 * Reference interfaces replaced (R64 = /root/reference/stitching/
 * OpenCV2.4.11-Stitching-64bit/OpenCV2.4.11-Stitching):
 *
 *   mst_create / mst_update   <- Stitcher setters, createDefault R64/src/stitcher.cpp:49-88,
 *                                 the cached calibration globals + UpdateMat()
 *                                 R64/stitching_detailed.cpp:29-53, 696-722
 *   mst_compose               <- StitchingAll() frame loop       R64/stitching_detailed.cpp:724-770
 *                                 Stitcher::composePanorama      R64/src/stitcher.cpp:264-313
 *                                 (warp -> exposure apply -> convertTo(16S) -> Blender::feed ...
 *                                  Blender::blend -> convertTo(8U))
 *   mst_canvas                <- resultRoi()                     R64/src/util.cpp:127-140
 *   mst_camera_roi, mst_warp_roi <- RotationWarper::warpRoi      detail/warpers_inl.hpp:129-140
 *   mst_build_maps            <- RotationWarper::buildMaps       detail/warpers_inl.hpp:62-85
 *   mst_warp                  <- RotationWarper::warp            detail/warpers_inl.hpp:88-99
 *   mst_exposure_apply        <- ExposureCompensator::apply      R64/src/exposure_compensate.cpp:150-153,
 *                                                                 225-246 (== BlockApply, DRV:310-331)
 *   mst_multiband_geometry, mst_num_bands <- MultiBandBlender::prepare/feed ROI maths  R64/src/blenders.cpp:203-269
 *   mst_compose_device, _device_batch, _device_batch_ex <- the same frame loop with the frames already on the device: what
 *                                 StitchingAll() is once GetMemData's memcpy (R64/stitching_sharemem.cpp:96-123) lands in HBM
 *                                 (include/mi355_ingest.h); the batch and the event arguments have no reference counterpart
 *   mst_pipeline_submit / _wait / _pending, mst_host_alloc / _free <- the loop's three stages overlapped: GetMemData ->
 *                                 remap ... feedSizeRemap -> convertTo(8U) / imshow            DRV:736-764
 *   mst_acquire_calibration / mst_release_calibration <- the Upflag / UpdateMat() swap between the calibration thread and
 *                                 the frame loop                                                DRV:829-836, 696-722
 *   mst_camera_mask           <- the warped + fixed-up seam mask the loop blends with          R64/src/stitcher.cpp:278-294, DRV:676-678
 *   mst_build_maps_host       <- buildMaps of the projectors that call libm per pixel          detail/warpers_inl.hpp:302-759
 *   mst_warp_kind_from_name   <- the --warp switch                                             CLI:556-570
 *   mst_exposure_feed, mst_gain_map_size <- GainCompensator / BlocksGainCompensator::feed      R64/src/exposure_compensate.cpp:73-147, 165-222
 *   mst_seam_voronoi          <- VoronoiSeamFinder (PairwiseSeamFinder::run + findInPair)      R64/src/seam_finders.cpp:74-85, 108-160
 *   mst_seam_dp               <- DpSeamFinder::find                                            R64/src/seam_finders.cpp:163-1045
 *   mst_init_undistort_rectify_map, mst_remap_fixed <- RunTestEyeFishMap (initUndistortRectifyMap, CV_16SC2) and the per-frame
 *                                 remap through those maps                                      DRV:201-238, 741
 *   mst_compose_lens_maps     <- SURVEY.md 8 f1 as written (lens maps composed into the warp map); no reference call
 *   mst_cvt_nv12_bgr          <- SURVEY.md 8 f3 (cv::cvtColor(COLOR_YUV2BGR_NV12) of 2.4.11); the reference is BGR-only
 *                                 (R64/stitching_sharemem.cpp:112-115)
 *   mst_strip                 <- SURVEY.md 8e, second form (the canvas in column strips, one per GPU); no reference counterpart
 *   mst_stage_cameras / _canvas / _select_set, mst_payload_bytes / _pack / _unpack, mst_pyramid_buffer
 *                             <- SURVEY.md 8e, first form (cameras one per GPU, one gather of the per-camera pyramids before
 *                                 Blender::blend, R64/src/blenders.cpp:359-377); no reference counterpart
 *   mst_calibration_blob_size / _save / _load <- none (what GetPreStruct() produces, DRV:348-694, kept in a file)
 *   mst_compose_profile, _profile_batch, mst_debug_read, mst_synth_frame, mst_build_info, mst_synchronize, mst_last_error,
 *   mst_destroy               <- none: per-stage timing (the reference prints its own, 20160308.cpp:807-836), diagnostics, lifetime
 *
 * Conventions (mirroring cv::Mat): images are 8-bit, BGR interleaved for colour, row-major
 * with an explicit byte stride (cv::Mat::data / cv::Mat::step); the caller owns every frame
 * and panorama buffer; the engine owns device memory and the cached calibration.  Every
 * function returns MST_OK (0) or a negative error code; no exception crosses the ABI.  A bad
 * type/size is an error (where the reference would CV_Assert).  mst_compose* may be called from
 * one thread at a time per engine (the reference's frame loop is single-caller); mst_update may
 * be called concurrently from a calibration thread and takes effect at the next compose.
 *
 * There is NO CPU fallback: every entry point that computes pixels runs hand-written HIP
 * kernels on gfx950 and fails with MST_ERR_NO_DEVICE / MST_ERR_HIP when no GPU is usable.
 */
#ifndef MI355_STITCH_H
#define MI355_STITCH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101 (round 6): the round-4 strip-halo entries (mst_strip_halo_bytes, mst_strip_front_batch, mst_strip_back_batch, MST_FLAG_STRIP_HALO)
 * are gone from the ABI -- a host built against 100 that used them must not bind this library silently */
#define MST_VERSION 101

/* error codes */
#define MST_OK                 0
#define MST_ERR_INVALID_ARG   -1   /* CV_Assert-class failure: bad kind, size, null pointer */
#define MST_ERR_NOT_CALIBRATED -2  /* compose before the first successful mst_update */
#define MST_ERR_HIP           -3   /* a HIP runtime call or kernel launch failed */
#define MST_ERR_NO_DEVICE     -4   /* no usable gfx950 device */
#define MST_ERR_NEED_MORE_IMGS -5  /* Stitcher::ERR_NEED_MORE_IMGS (stitcher.hpp:62) */
#define MST_ERR_OUT_OF_MEMORY -6

/* --warp (CLI:102); projector kinds of detail/warpers.hpp:88-503.  Plane, spherical and cylindrical maps are
 * evaluated on the device from per-column / per-row tables; the others call libm per pixel in the reference
 * (detail/warpers_inl.hpp:302-759), so the engine evaluates their backward maps on the host once per calibration --
 * RotationWarperBase::buildMaps -- and the kernels read them.  The per-frame path is the same for all. */
#define MST_WARP_PLANE        0
#define MST_WARP_SPHERICAL    1
#define MST_WARP_CYLINDRICAL  2
#define MST_WARP_FISHEYE                   3
#define MST_WARP_STEREOGRAPHIC             4
#define MST_WARP_COMPRESSED_PLANE          5    /* CompressedRectilinearWarper(scale, A, B) */
#define MST_WARP_COMPRESSED_PLANE_PORTRAIT 6
#define MST_WARP_PANINI                    7    /* PaniniWarper(scale, A, B) */
#define MST_WARP_PANINI_PORTRAIT           8
#define MST_WARP_MERCATOR                  9
#define MST_WARP_TRANSVERSE_MERCATOR       10
#define MST_WARP_SPHERICAL_PORTRAIT        11   /* detail::SphericalPortraitWarper (no --warp name) */
#define MST_WARP_CYLINDRICAL_PORTRAIT      12
#define MST_WARP_PLANE_PORTRAIT            13
/* A, B of the compressed-rectilinear and Panini projectors ride in bits 8-9 of the kind: 0 = the constructors'
 * defaults (1, 1); MST_WARP_A2B1 / MST_WARP_A1_5B1 = the pairs behind the command line's "...A2B1" / "...A1.5B1"
 * names (CLI:561-568); MST_WARP_AB_CUSTOM = mst_config.warp_a / warp_b (engine only). */
#define MST_WARP_A2B1         (1 << 8)
#define MST_WARP_A1_5B1       (2 << 8)
#define MST_WARP_AB_CUSTOM    (3 << 8)
/* the reference's --warp names (CLI:102,556-570) -> kind; -1 for an unknown name */
int mst_warp_kind_from_name(const char *name);
/* --blend (CLI:112); Blender::NO/FEATHER/MULTI_BAND, detail/blenders.hpp:58 */
#define MST_BLEND_NO          0
#define MST_BLEND_FEATHER     1
#define MST_BLEND_MULTI_BAND  2
/* --expos_comp (CLI:110); ExposureCompensator::NO/GAIN/GAIN_BLOCKS */
#define MST_EXPOS_NO          0
#define MST_EXPOS_GAIN        1
#define MST_EXPOS_GAIN_BLOCKS 2
/* cv:: interpolation / border codes used by RotationWarper::warp */
#define MST_INTER_NEAREST     0
#define MST_INTER_LINEAR      1
#define MST_BORDER_CONSTANT   0
#define MST_BORDER_REFLECT    2

/* mst_config.flags */
#define MST_FLAG_DRIVER_LUT   1u   /* the driver's hard-seam LUT composite with crop margins
                                      (feedSizeRemap, DRV:150-177); requires MST_BLEND_NO */

#define MST_FLAG_DEBUG        2u   /* keep a copy of the collapsed canvas levels for mst_debug_read */
#define MST_FLAG_GRAPH_REPLAY 32u   /* mst_compose_device(_batch): the launch sequence of a call (5-6 kernels for the multi-band blend)
                                      is captured into a hipGraph the first time a given set of buffers (frames, panoramas, masks,
                                      strides, count) is seen under a calibration, and replayed with one hipGraphLaunch from then
                                      on -- a frame loop over a ring of buffers pays the kernel-by-kernel enqueue once per ring
                                      slot.  Same kernels, same results; not with MST_FLAG_OVERLAP_FRAMES (whose calls depend on
                                      each other).  Up to 32 sequences are kept per engine. */
#define MST_FLAG_FRAMES_NV12 16u   /* the frames of every compose call (host, device, batch, pipeline) are NV12 as decoders deliver
                                      them: one plane pointer per camera, src_height rows of Y then src_height / 2 rows of
                                      interleaved U V, all `stride` bytes apart (a CV_8UC1 Mat of height * 3 / 2 rows, what
                                      cvtColor(COLOR_YUV2BGR_NV12) takes); stride >= src_width; width and height even.  The engine
                                      converts on the device first (OpenCV 2.4.11's fixed-point BT.601, bit for bit) -- half the
                                      bytes over PCIe for host frames.  The reference itself is BGR-only
                                      (stitching_sharemem.cpp:112-115); SURVEY 8 f3.  Not for the mst_stage_* calls. */
#define MST_FLAG_LENS_COMPOSED 8u  /* SURVEY 8 f1 as written: cameras that carry undistort maps get them COMPOSED into the warp's
                                      backward map at mst_update (mst_compose_lens_maps) and every frame is resampled ONCE, from the
                                      raw frame -- no pre-pass, the plain rate.  Not the reference's pixels (it resamples twice,
                                      DRV:741 then DRV:752): one interpolation blur less; parity is against the oracle's
                                      statement of the same composition. */
#define MST_FLAG_OVERLAP_FRAMES 4u /* throughput mode for mst_compose_device(_batch): the per-camera kernels of call t+1
                                      run on an engine-owned stream while the canvas kernel of call t is still
                                      running on the caller's stream (two runs of pyramid sets alternate).  Results and
                                      their order on `stream` are unchanged.  Contract: the frames passed to a call
                                      must be complete in device memory when the call is made (they are read without
                                      waiting for work still queued on `stream`) -- or the call is made through
                                      mst_compose_device_batch_ex with a `frames_ready` event -- and must stay untouched until
                                      the panorama of that call is complete.  With overlap_depth >= 3 the panorama is WRITTEN
                                      on an engine-owned stream as well: a buffer that is passed again must have no pending
                                      reader on any stream (`outputs_free`, see mst_compose_device_batch_ex).
                                      Multi-band only; not with MST_FLAG_DEBUG. */

/* MultiBandBlender weight pyramid type (blenders.cpp:189-194): CV_32F (default) or CV_16S (weights 0, 2..256,
 * integer pyrDown, `dst += short((src * w) >> 8)`, normalise `(v << 8) / (w + 1)`; blenders.cpp:289-292,331-355,406-423) */
#define MST_WEIGHT_32F 0
#define MST_WEIGHT_16S 1

#define MST_MAX_CAMERAS 16
#define MST_MAX_BATCH 8
#define MST_MAX_BANDS   10

typedef struct mst_engine mst_engine;

typedef struct mst_config {
    int32_t  struct_size;        /* sizeof(mst_config) */
    int32_t  num_cameras;        /* 1..MST_MAX_CAMERAS (DRV:29 num_images) */
    int32_t  warp_kind;          /* MST_WARP_* */
    int32_t  blend_kind;         /* MST_BLEND_* */
    int32_t  num_bands;          /* MultiBandBlender::setNumBands; <= 0 -> 5 (blenders.hpp:101) */
    float    feather_sharpness;  /* FeatherBlender::setSharpness; <= 0 -> 0.02f (blenders.hpp:75) */
    int32_t  expos_kind;         /* MST_EXPOS_* */
    int32_t  device;             /* HIP device ordinal */
    uint32_t flags;              /* MST_FLAG_* */
    float    upblack, downblack, leftblack, rightblack;   /* DRV:47 crop margins (LUT path) */
    int32_t  max_batch;          /* frame sets per mst_compose_device_batch call, 0/1 = none, <= MST_MAX_BATCH;
                                  * multi-band keeps one set of Gaussian pyramids per frame set of a batch */
    int32_t  weight_type;        /* MST_WEIGHT_*: MultiBandBlender's weight_type (blenders.hpp:101), default CV_32F */
    float    warp_a, warp_b;     /* A, B of CompressedRectilinear* / Panini* warpers when warp_kind carries
                                  * MST_WARP_AB_CUSTOM (detail/warpers.hpp:232-305) */
    /* Canvas split over GPUs (second form): this engine composes column strip `strip_index` of
     * `strip_count` equal strips of the canvas (boundaries on multiples of 256 panorama columns; mst_strip tells which).
     * Every kernel of the path runs only on the tiles that strip can depend on -- for the multi-band blend the strip
     * widened by the pyrUp reach of every level -- so its pixels are exactly those of the whole panorama.  The panorama
     * buffer keeps the whole canvas's geometry (pointer = canvas column 0; only the strip's columns are written and the
     * row stride only has to cover the strip's last column).  strip_count 0 or 1: the whole canvas. */
    int32_t  strip_index, strip_count;
    /* MST_FLAG_OVERLAP_FRAMES: calls in flight.  0 .. 2: the two-stream form described at the flag.  3 .. 8: every call runs
     * WHOLE on one of that many engine-owned streams, round-robin, with its own run of pyramid / pre-pass buffer sets, and
     * the caller's stream waits for it -- independent launch sequences fill each other's launch boundaries and tails (C3,
     * 4 in flight: one-frame-set calls 9.5 k -> 14.6 k panoramas/s, batches of 8 15.9 k -> 18.5 k).  Same contract as the
     * flag; this form also serves the feather / paste / LUT composites.  Memory: overlap_depth x max_batch buffer sets. */
    int32_t  overlap_depth;
} mst_config;

typedef struct mst_camera {
    int32_t  src_width, src_height;       /* frame size (cv::Mat cols, rows) */
    float    K[9];                        /* CameraParams::K() as CV_32F, row-major (stitcher.cpp:271-272) */
    float    R[9];                        /* CameraParams::R as CV_32F, row-major */
    float    T[3];                        /* PlaneWarper translation; zeros otherwise */
    const float *gain_map;                /* GAIN_BLOCKS: gain_maps_[i], gain_h x gain_w floats, tight */
    int32_t  gain_w, gain_h;
    float    gain;                        /* GAIN: gains_(i) */
    const uint8_t *mask;                  /* final mask_warped (stitcher.cpp:294): CV_8U, must be
                                             exactly the warpRoi() size of this camera */
    int32_t  mask_width, mask_height;
    size_t   mask_stride;                 /* bytes */
    /* Optional lens-undistort pre-pass (the driver's mapEye1/mapEye2, DRV:232,741): fixed-point maps as
     * initUndistortRectifyMap(..., CV_16SC2, map1, map2) / mst_init_undistort_rectify_map produce them,
     * src_height x src_width, tightly packed.  Both NULL: the frames are used as they come.  When set,
     * every compose call first runs remap(frame, undistorted, map1, map2, INTER_LINEAR) for this camera into
     * an engine-owned buffer and warps that. */
    const int16_t  *undistort_map1;       /* CV_16SC2: (x, y) integer source coordinate per pixel */
    const uint16_t *undistort_map2;       /* CV_16UC1: (fy << 5) | fx */
    /* Optional: let the engine do the mask fix-up of the compose loop on the device (stitcher.cpp:291-294 =
     * DRV:676-678):  mask_warped = resize(dilate(seam_mask), warpRoi size) & warp(all-255 mask, NEAREST, CONSTANT).
     * seam_mask is the seam finder's output for this camera at seam scale (CV_8U, any size); when it is set,
     * `mask` may be NULL (it is ignored).  The final mask can be read back with mst_camera_mask. */
    const uint8_t *seam_mask;
    int32_t  seam_mask_width, seam_mask_height;
    size_t   seam_mask_stride;            /* bytes */
} mst_camera;

typedef struct mst_calibration {
    int32_t  struct_size;                 /* sizeof(mst_calibration) */
    int32_t  num_cameras;                 /* must equal the engine's */
    float    warped_image_scale;          /* warper scale (stitcher.cpp:239-240) */
    const mst_camera *cameras;
} mst_calibration;

typedef struct mst_rect { int32_t x, y, width, height; } mst_rect;

/* ---- a calibration as one binary (host only) --------------------------------------------------
 * Everything mst_update takes -- K, R, T, scale, gain maps, masks, lens maps, seam masks -- in a self-contained,
 * checksummed blob: a calibration that GetPreStruct() (DRV:348-694) produced once can be replayed (benchmarks and parity
 * runs on the same inputs; a restart without re-registration).  The reference has no such thing.
 * mst_calibration_blob_size: bytes needed, 0 for an invalid calibration.  mst_calibration_save: `written` = bytes used.
 * mst_calibration_load: fills `out` and `cameras` (room for max_cameras); their array pointers point INTO the blob, which
 * must stay alive and unchanged while the calibration is in use (mst_update copies what it needs) and must START ON AN
 * 8-BYTE BOUNDARY (malloc, mmap at a page offset and numpy buffers do; a slice at an odd offset is MST_ERR_INVALID_ARG).
 * Masks come back tightly packed.  A camera with exactly one of undistort_map1 / undistort_map2 cannot be saved (size 0).  A blob that is truncated, altered or of another version is MST_ERR_INVALID_ARG. */
size_t mst_calibration_blob_size(const mst_calibration *calib);
int    mst_calibration_save(const mst_calibration *calib, void *blob, size_t capacity, size_t *written);
int    mst_calibration_load(const void *blob, size_t bytes, mst_calibration *out, mst_camera *cameras, int max_cameras);

/* ---- what this library binary is ------------------------------------------------------------
 * The kernel sources carry compile-time switches for timing experiments (-DMST_ABLATE_*: a part of a kernel removed to see what it
 * costs -- the pixels are WRONG) and for in-kernel phase stamps (-DMST_STAMPS).  A library built with any of them says so here,
 * and mst_create prints a warning to stderr: such a binary is for tools/variants.sh, never for a product or a recorded number
 * (tests/test_capi_host.py asserts that the shipped library has none).  struct_size = sizeof(mst_build) on entry. */
typedef struct mst_build {
    int32_t  struct_size;
    int32_t  version;          /* MST_VERSION of the library */
    uint32_t timing_build;     /* 0 = product build; bits: 1 MST_ABLATE_A, 2 MST_ABLATE_A2, 4 MST_ABLATE_C, 8 MST_ABLATE_STAGE, 16 MST_STAMPS */
    int32_t  level0_packed;    /* level 0 of the per-camera pyramids: 1 packed BGR8 (default), 0 a dword a pixel (-DMST_L0_PACKED=0) */
    char     description[160];
} mst_build;
int  mst_build_info(mst_build *out);

/* ---- engine lifetime ------------------------------------------------------------------- */
int  mst_create(const mst_config *cfg, mst_engine **out);
void mst_destroy(mst_engine *e);
/* message of the last failed call made by the CALLING THREAD (on any engine or without one; `e` is accepted for
 * symmetry and ignored); valid until that thread's next failing call.  Per thread because mst_update and
 * mst_compose* may run -- and fail -- concurrently on one engine. */
const char *mst_last_error(const mst_engine *e);

/* Publish a calibration: builds every frame-invariant product on the device (ROI geometry,
 * backward-map tables, gain tables, weight pyramids, weight sums, canvas mask) into a fresh
 * state and swaps it in atomically; the next mst_compose uses it (UpdateMat, DRV:696-722). */
int  mst_update(mst_engine *e, const mst_calibration *calib);

/* Latch the newest published calibration for the frame thread and return its canvas and generation (1, 2, ... in
 * mst_update order).  This is the reference's `if (Upflag) UpdateMat();` at the top of the frame loop
 * (DRV:835-836): the thread that composes decides when a new calibration takes effect, so the panorama size
 * cannot change between sizing the output buffer and composing into it.  From the first call on, every
 * mst_compose* / mst_stage* / mst_canvas of this engine uses the latched calibration until the next
 * mst_acquire_calibration; an engine that never calls it keeps the old behaviour (each call picks up the newest
 * calibration on entry -- then size panorama buffers for the largest canvas a recalibration can produce, or do
 * not recalibrate concurrently).  Call it from the compose thread only. */
int  mst_acquire_calibration(mst_engine *e, mst_rect *canvas, uint64_t *generation);
/* Leave the latching mode again: from here on every call picks up the newest published calibration on entry (the
 * behaviour of an engine that never called mst_acquire_calibration).  Before the first successful mst_update
 * mst_acquire_calibration returns MST_ERR_NOT_CALIBRATED and does not start latching.  Compose thread only. */
int  mst_release_calibration(mst_engine *e);
/* resultRoi of the calibration the next compose will use (see mst_acquire_calibration); with
 * MST_FLAG_DRIVER_LUT the cropped output rect */
int  mst_canvas(const mst_engine *e, mst_rect *canvas);
/* the columns this engine writes (mst_config.strip_index / strip_count), as a rectangle inside mst_canvas's */
int  mst_strip(const mst_engine *e, mst_rect *strip);
/* corners[i] / sizes[i] */
int  mst_camera_roi(const mst_engine *e, int camera, mst_rect *roi);
/* final mask_warped of a camera as the engine holds it (warpRoi-sized CV_8U, `stride` bytes per row): the caller's
 * mask, or the result of the device-side fix-up when mst_camera.seam_mask was given */
int  mst_camera_mask(mst_engine *e, int camera, uint8_t *mask, size_t stride);
/* number of bands actually used (MultiBandBlender::prepare crop, blenders.cpp:207-209) */
int  mst_num_bands(const mst_engine *e, int *num_bands);

/* ---- the per-frame hot path ------------------------------------------------------------ */
/* Host buffers (cv::Mat data/step): H2D copy of the N frames, compose, D2H of the panorama.
 * pano is canvas.width x canvas.height BGR; pano_mask (nullable) is the blender's dst_mask. */
int  mst_compose(mst_engine *e, const uint8_t *const *frames, const size_t *frame_strides,
                 uint8_t *pano, size_t pano_stride, uint8_t *pano_mask, size_t pano_mask_stride);
/* Device buffers already resident in HBM; enqueued on `stream` (a hipStream_t, NULL = the
 * engine's own stream) and asynchronous: call mst_synchronize or sync the stream yourself.
 * The engine's own stream is NON-BLOCKING (it does not wait for the legacy default stream): work that
 * produces the frames or still touches the panorama buffer on another stream -- a decoder's copy, a fill
 * of a freshly allocated buffer -- must have completed, or the caller passes the stream that work is on. */
int  mst_compose_device(mst_engine *e, const void *const *d_frames, const size_t *frame_strides,
                        void *d_pano, size_t pano_stride, void *d_pano_mask, size_t pano_mask_stride,
                        void *stream);
/* Throughput mode: `count` independent frame sets (1 <= count <= mst_config.max_batch) through one launch
 * sequence -- every kernel of the path runs once over all sets, which amortises the launch boundaries and fills
 * the chip during the small pyramid-chain launches.  The reference stitches one frame set per StitchingAll()
 * call (DRV:724-770); results are exactly those of `count` mst_compose_device calls.  d_frames / frame_strides
 * hold count * num_cameras entries (frame set f, camera i at f * num_cameras + i); d_panos (and d_pano_masks,
 * or NULL) hold one pointer per frame set, all with the same strides. */
int  mst_compose_device_batch(mst_engine *e, int count, const void *const *d_frames, const size_t *frame_strides,
                              void *const *d_panos, size_t pano_stride, void *const *d_pano_masks,
                              size_t pano_mask_stride, void *stream);
/* mst_compose_device_batch for a host whose frames ARRIVE on a stream (hipMemcpyAsync from a capture buffer, a decoder: the
 * reference's GetMemData -> compose shape, DRV:739-756 / stitching_sharemem.cpp:96-123) and whose panoramas are READ on one (an
 * asynchronous copy to the display / encoder), without a host synchronisation per call:
 *   frames_ready   a hipEvent_t recorded behind the last write of this call's frames (on the copy / decode stream), or NULL.
 *                  The call's first kernel waits for it -- on whichever stream the engine runs the call.
 *   outputs_free   a hipEvent_t recorded behind the last READ of the panorama / mask buffers passed to this call (the consumer's
 *                  copy of what an earlier call left in them), or NULL.  The call's kernels wait for it before they write.
 * With both NULL this is mst_compose_device_batch.  They matter with calls in flight (MST_FLAG_OVERLAP_FRAMES, overlap_depth >= 3):
 * such a call runs on an engine-owned stream that is NOT ordered behind work queued earlier on `stream`, so (a) frames that are
 * still being written by work queued on any stream need `frames_ready` (or must be complete when the call is made), and (b) an
 * output buffer must have no pending reader on any stream when it is passed again -- give `outputs_free`, or keep a ring of more
 * buffers than calls can be in flight and synchronise the consumer.  Writers are ordered by the engine itself: two calls in flight
 * whose output byte ranges overlap run in call order.  Results appear on `stream` in call order, as always.  The events are only
 * waited for; the engine neither records nor destroys them, and they may be re-recorded
 * once the call has returned (a wait captures the event's state at the time it is queued). */
int  mst_compose_device_batch_ex(mst_engine *e, int count, const void *const *d_frames, const size_t *frame_strides,
                                 void *const *d_panos, size_t pano_stride, void *const *d_pano_masks,
                                 size_t pano_mask_stride, void *stream, void *frames_ready, void *outputs_free);
/* The host-buffer frame loop as a three-stage pipeline: while the kernels of frame t run, frame t+1 is copied in and
 * the panorama of frame t-1 is copied out (three streams, a ring of MST_PIPELINE_DEPTH device frame sets).  The
 * reference's loop is strictly serial (GetMemData -> remap ... -> imshow per frame, DRV:736-763); with host
 * buffers mst_compose is bound by its two PCIe copies in series, the pipeline by the larger of them.
 *   mst_pipeline_submit  queue one frame set (same arguments as mst_compose, no mask plane); returns at once.
 *                        At most MST_PIPELINE_DEPTH frames may be in flight (MST_ERR_INVALID_ARG beyond that).
 *   mst_pipeline_wait    block until the OLDEST submitted frame's panorama is complete in the buffer it was
 *                        submitted with; returns that buffer and the canvas rectangle it was composed for.
 * The frames and the panorama buffer must stay untouched from submit to the matching wait, and should be
 * page-locked: allocate them with mst_host_alloc (or hipHostRegister your own, as mi355_ingest does for its shared
 * memory segment) -- with pageable memory the copies serialise and nothing overlaps.  Frame thread only. */
#define MST_PIPELINE_DEPTH 3
void *mst_host_alloc(size_t bytes);                 /* page-locked host memory; NULL on failure */
void  mst_host_free(void *p);
int  mst_pipeline_submit(mst_engine *e, const uint8_t *const *frames, const size_t *frame_strides,
                         uint8_t *pano, size_t pano_stride);
int  mst_pipeline_wait(mst_engine *e, uint8_t **pano, mst_rect *canvas);
int  mst_pipeline_pending(const mst_engine *e);     /* frames submitted and not yet waited for */
int  mst_synchronize(mst_engine *e);
/* One frame of mst_compose_device with a HIP event recorded on `stream` between consecutive
 * kernel launches: the counterpart of the reference's getTickCount() brackets
 * ("Stitching, time: ..." DRV:760).  names[i] / stage_ms[i] receive the kernel name and its
 * duration in milliseconds for up to max_stages launches; *num_stages = launches in the frame. */
int  mst_compose_profile(mst_engine *e, const void *const *d_frames, const size_t *frame_strides,
                         void *d_pano, size_t pano_stride, void *stream, int max_stages,
                         char (*names)[32], float *stage_ms, int *num_stages);
/* the same for one mst_compose_device_batch call */
int  mst_compose_profile_batch(mst_engine *e, int count, const void *const *d_frames, const size_t *frame_strides,
                               void *const *d_panos, size_t pano_stride, void *stream, int max_stages,
                               char (*names)[32], float *stage_ms, int *num_stages);

/* ---- camera streams sharded over GPUs (multi-band only) ---------------------------------------
 * The reference runs its per-camera work under `#pragma omp parallel for` over cameras (DRV:736,748) and
 * composites on one thread (DRV:759).  Across GPUs the same split is: every rank holds the full
 * calibration; rank r runs mst_stage_cameras for the cameras it owns, the per-camera Gaussian pyramid
 * blocks (mst_pyramid_buffer, one contiguous device buffer per camera, identical size on every rank)
 * are gathered once per frame into the blending rank's buffers (RCCL send/recv over xGMI), and that rank
 * runs mst_stage_canvas.  mst_compose_device == stage_cameras(all) + stage_canvas on one GPU. */
int  mst_stage_cameras(mst_engine *e, uint32_t camera_mask, const void *const *d_frames,
                       const size_t *frame_strides, void *stream);   /* entries outside the mask are ignored */
int  mst_stage_canvas(mst_engine *e, void *d_pano, size_t pano_stride, void *d_pano_mask,
                      size_t pano_mask_stride, void *stream);
/* The hand-off payload of one camera: the interiors of its launched tiles at every pyramid level, packed back to
 * back -- everything the canvas stage can read of that camera and nothing else (the wrap-around camera of a
 * 360-degree rig, whose pyramid block spans the whole canvas, ships about as much as any other camera).  Sizes
 * and layout are static per calibration and identical on every rank that published the same calibration.
 * mst_payload_pack gathers it from the engine's pyramids into `d_payload` (device memory, mst_payload_bytes
 * long) on the sending rank; mst_payload_unpack scatters a received one into the blending rank's pyramids.
 * mst_stage_select_set picks which of the engine's pyramid sets (0 .. mst_config.max_batch - 1) the mst_stage_* and
 * mst_payload_* calls work on: with two sets the exchange of frame t overlaps the per-camera stage of frame t+1. */
int  mst_payload_bytes(mst_engine *e, int camera, size_t *bytes);
int  mst_payload_pack(mst_engine *e, int camera, void *d_payload, void *stream);
int  mst_payload_unpack(mst_engine *e, int camera, const void *d_payload, void *stream);
int  mst_stage_select_set(mst_engine *e, int set);
/* The buffer belongs to the calibration the call resolves to (see mst_acquire_calibration) and is freed when that
 * calibration is retired: query it again after every mst_update / mst_acquire_calibration, never cache it across. */
int  mst_pyramid_buffer(mst_engine *e, int camera, void **d_ptr, size_t *bytes);

/* ---- stage-level entry points (the reference's plugin interfaces) ------------------------ */
/* RotationWarper::warpRoi -- host only, no device needed */
int  mst_warp_roi(int warp_kind, float scale, const float K[9], const float R[9], const float T[3],
                  int src_width, int src_height, mst_rect *roi);
/* RotationWarper::buildMaps: xmap/ymap are roi.height x roi.width floats (host, tight).
 * Runs the same in-register backward map the warp kernel uses and materialises it. */
int  mst_build_maps(int device, int warp_kind, float scale, const float K[9], const float R[9],
                    const float T[3], int src_width, int src_height, float *xmap, float *ymap);
/* Host-only variant of the above (same tables, same float operation order, no device): used to
 * derive calibration inputs such as all-255 warped masks where no GPU is present. */
int  mst_build_maps_host(int warp_kind, float scale, const float K[9], const float R[9],
                         const float T[3], int src_width, int src_height, float *xmap, float *ymap);
/* cv::initUndistortRectifyMap(K, dist, R, newK, Size(width, height), CV_16SC2, map1, map2) as RunTestEyeFishMap calls
 * it (DRV:201-238); host only.  K, newK: 3x3 row-major doubles; R: 3x3 or NULL (identity); dist: 0, 4, 5 or 8
 * coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6]]).  map1: height x width x 2 shorts, map2: height x width ushorts. */
/* cv::cvtColor(yuv, bgr, COLOR_YUV2BGR_NV12) on the device (the conversion behind MST_FLAG_FRAMES_NV12), host buffers:
 * yuv = (height * 3 / 2) rows of yuv_stride bytes, bgr = height rows of bgr_stride bytes; width and height even. */
int  mst_cvt_nv12_bgr(int device, const uint8_t *yuv, int width, int height, size_t yuv_stride, uint8_t *bgr, size_t bgr_stride);
/* The map composition of MST_FLAG_LENS_COMPOSED, host only: (xmap, ymap)[k] <- the lens maps (map1 + map2's fractions / 32)
 * interpolated bilinearly at (xmap, ymap)[k] clamped to the width x height frame, in float32:
 * x0 = min((int)u, width - 2), ax = u - x0; r0 = m00 * (1 - ax) + m10 * ax; r1 = m01 * (1 - ax) + m11 * ax;
 * L = r0 * (1 - ay) + r1 * ay.  n map entries, updated in place. */
int  mst_compose_lens_maps(float *xmap, float *ymap, size_t n, const int16_t *map1, const uint16_t *map2, int width, int height);
int  mst_init_undistort_rectify_map(const double K[9], const double *dist, int num_dist, const double R[9],
                                    const double newK[9], int width, int height, int16_t *map1, uint16_t *map2);
/* remap(src, dst, map1, map2, INTER_LINEAR) for a CV_8UC3 host image with such maps (default constant border,
 * DRV:741); dst has the size of src. */
int  mst_remap_fixed(int device, const uint8_t *src, int width, int height, size_t src_stride,
                     const int16_t *map1, const uint16_t *map2, uint8_t *dst, size_t dst_stride);
/* RotationWarper::warp for CV_8UC1 / CV_8UC3 host images; dst is the warpRoi() size.
 * interp/border: (LINEAR, REFLECT) for images, (NEAREST, CONSTANT) for masks. */
int  mst_warp(int device, int warp_kind, float scale, const float K[9], const float R[9],
              const float T[3], const uint8_t *src, int src_width, int src_height, size_t src_stride,
              int channels, int interp, int border, uint8_t *dst, size_t dst_stride);
/* ExposureCompensator::apply on a CV_8UC3 host image, in place.
 * GAIN_BLOCKS: gain_map gain_h x gain_w; GAIN: scalar `gain`. */
int  mst_exposure_apply(int device, int expos_kind, uint8_t *image, int width, int height,
                        size_t stride, const float *gain_map, int gain_w, int gain_h, float gain);
/* VoronoiSeamFinder::find (--seam voronoi, CLI:106; seam_finders.cpp:74-85,108-160) on the device: for every
 * overlapping pair i < j, in order, the overlap pixels go to the image whose un-contested region is nearer in the L1
 * metric.  masks[i]: CV_8U of sizes_wh[2i] x sizes_wh[2i+1] at corner corners_xy[2i..] (host memory), updated in
 * place.  (The graph-cut seam finder stays on the host: out of scope) */
int  mst_seam_voronoi(int device, int num_images, const int32_t *corners_xy, const int32_t *sizes_wh,
                      uint8_t *const *masks, const size_t *mask_strides);
/* DpSeamFinder::find (--seam dp_color / dp_colorgrad, CLI:627-630; seam_finders.cpp:163-1045): every pair of images,
 * farthest centres first; per pair the connected components of "first only / second only / both", the conflicts between
 * neighbouring components resolved by a minimum-cost seam (dynamic programming over the colour difference, optionally
 * divided by the gradient magnitude) between the two most distant contact points.  images[i]: CV_8UC3 or CV_32FC3 (the
 * sample feeds CV_32F, CLI:488-489) of sizes_wh[2i] x sizes_wh[2i+1] at corners_xy[2i..]; masks[i]: CV_8U, updated in
 * place.  The per-pixel passes (contour masks, gradients, cost maps, the DP recurrence, the mask update) run on the device;
 * the component bookkeeping, whose order is the result, on the host. */
#define MST_IMAGE_8UC3   0
#define MST_IMAGE_32FC3  1
#define MST_SEAM_DP_COLOR       0
#define MST_SEAM_DP_COLOR_GRAD  1
int  mst_seam_dp(int device, int num_images, const int32_t *corners_xy, const int32_t *sizes_wh, const void *const *images,
                 const size_t *image_strides, int image_type, uint8_t *const *masks, const size_t *mask_strides,
                 int cost_function);
/* ExposureCompensator::feed -- the calibration-side half (exposure_compensate.cpp:63-147, 165-222; the driver's
 * BlockFeed, DRV:251-306): gains from the overlaps of the seam-scale warped images.  images[i]: CV_8UC3 of
 * sizes_wh[2i] x sizes_wh[2i+1] at canvas corner corners_xy[2i..]; masks[i]: CV_8U, 255 = valid.
 * GAIN: gains[num_images] (doubles, GainCompensator::gains()).  GAIN_BLOCKS: gain_maps[i] receives the camera's
 * block gain map, mst_gain_map_size floats (block size 32 x 32 in the reference, exposure_compensate.hpp:96) -- exactly
 * what mst_camera.gain_map takes.  The overlap statistics run on the device (one thread per pair of overlapping
 * blocks, in the reference's summation order), the normal equations, cv::solve and the smoothing on the host. */
int  mst_gain_map_size(int image_width, int image_height, int block_width, int block_height, int *gain_w, int *gain_h);
int  mst_exposure_feed(int device, int expos_kind, int num_images, const int32_t *corners_xy,
                       const uint8_t *const *images, const size_t *image_strides, const int32_t *sizes_wh,
                       const uint8_t *const *masks, const size_t *mask_strides, int block_width, int block_height,
                       double *gains, float *const *gain_maps);
/* MultiBandBlender::prepare + feed integer geometry -- host only.
 * out[0..3] = tl_new.x, tl_new.y, br_new.x, br_new.y; out[4..7] = top, left, bottom, right */
int  mst_multiband_geometry(mst_rect canvas, int num_bands, mst_rect image, mst_rect *padded_canvas,
                            int *used_bands, int out[8]);

/* ---- introspection for parity tests ------------------------------------------------------ */
#define MST_DBG_CAM_WARPED_S16   1   /* bordered level-0 image of camera c, planar 3 x h x w int16 (pixels no panorama pixel
                                       can depend on read 0; without MST_FLAG_DEBUG so do tiles that only feed level 1) */
#define MST_DBG_CAM_GAUSS_S16    2   /* Gaussian level `level` of camera c, planar int16 */
#define MST_DBG_CAM_WEIGHT_F32   3   /* weight pyramid level of camera c, h x w float */
#define MST_DBG_BAND_WEIGHT_F32  4   /* dst_band_weights_[level], h x w float */
#define MST_DBG_CANVAS_S16       5   /* collapsed canvas level (>= 1), planar 3 x h x w int16; needs MST_FLAG_DEBUG */
#define MST_DBG_CAM_NEED_U8      6   /* h x w bytes: 1 where the Gaussian pixel of camera c / level can reach
                                        the panorama (the engine skips tiles without any such pixel; skipped
                                        pixels of the two buffers above read as 0) */
/* Query (buf == NULL) or copy out the buffer of the last compose / current calibration. */
int  mst_debug_read(mst_engine *e, int what, int camera, int level, void *buf, size_t buf_bytes,
                    int *width, int *height);

/* ---- synthetic inputs (SURVEY.md section 8d / BASELINE.md section 3) ---------------------- */
/* Frame `frame_index` of camera `camera`: smooth texture + PCG32 noise, BGR u8.  Host only. */
int  mst_synth_frame(int camera, int frame_index, int width, int height, uint8_t *dst, size_t stride);

#ifdef __cplusplus
}
#endif
#endif /* MI355_STITCH_H */
