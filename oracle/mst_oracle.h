/*
 * mst_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C restatement of the reference's per-frame compositing arithmetic
 * (wangzjpku/StitchingVideo = vendored OpenCV 2.4.11 `stitching` module + the
 * OpenCV 2.4.11 core/imgproc primitives it calls).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (libmi355stitch.so) never links or calls it.
 *
 * PARITY STATUS (round 5): PINNED for remap (8-bit bilinear, every border mode), pyrDown / pyrUp (16S), pyrDown (32F),
 * normalizeUsingWeightMap (both weight types), cvtScale (gain apply, mask -> weight), distanceTransform (L1, 3 x 3), the row loops of
 * resize(INTER_LINEAR) (32F, 8U), the float Sobel row / column filters, the gain-map smoothing, BGR2GRAY (8U, 32F), borderInterpolate
 * and the NV12 conversion against
 * OUTPUTS OF THE REFERENCE ITSELF: its shipped x64 DLLs are run in place (oracle/ref_run/pe_run.c, tests/ref_run.py) and the
 * results committed as tests/golden/ref_runs.npz; everything below that is not in that list stays as it was:
 * "parity unpinned" -- the reference ships no tests, no golden
 * vectors and no source for OpenCV 2.4.11 imgproc/core (Windows DLLs only), and
 * cannot be compiled or run in this image.  Reference-owned
 * loops (the .cpp files under R64/src) are followed line by line and cited below; the
 * OpenCV 2.4.11 primitive semantics (remap, pyrDown, pyrUp, resize,
 * copyMakeBorder, saturating add/subtract, convertTo) are restated from the
 * published 2.4.11 algorithm, each isolated in one function with a
 * known-answer test in tests/test_oracle_kat.py.
 * Round 5, a first pin: the reference ships the x64 DLLs it runs, and the order-sensitive
 * FLOAT primitives -- Sobel(CV_32F, ksize 3) and the float pyrDown of the blend weights,
 * by loop (SSE body / scalar tails) -- plus the fixed-point gray and NV12 constants were read
 * off opencv_imgproc2411.dll's machine code; the scalar float steps of the blenders (feed
 * product + truncation, normalisation by true division: opencv_stitching2411.dll) and of the gain
 * applies (float product, double rounding to nearest even: the exe's own
 * BlocksGainCompensator::apply, opencv_core2411.dll's cvtScale) were read the same way
 * (tools/ref_binary_audit.py, oracle/REFERENCE_BINARY_AUDIT.md,
 * tests/test_reference_binary_audit.py).  Everything else stays "parity unpinned".
 *
 * Path alias: R64 = /root/reference/stitching/OpenCV2.4.11-Stitching-64bit/
 *                   OpenCV2.4.11-Stitching
 */
#ifndef MST_ORACLE_H
#define MST_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- enums (values follow OpenCV 2.4.11 / the reference headers) ---------- */
enum { MO_WARP_PLANE = 0, MO_WARP_SPHERICAL = 1, MO_WARP_CYLINDRICAL = 2,
       /* the other projectors of detail/warpers.hpp:190-493 (warpers_inl.hpp:302-759) */
       MO_WARP_FISHEYE = 3, MO_WARP_STEREOGRAPHIC = 4, MO_WARP_COMPRESSED_PLANE = 5,
       MO_WARP_COMPRESSED_PLANE_PORTRAIT = 6, MO_WARP_PANINI = 7, MO_WARP_PANINI_PORTRAIT = 8,
       MO_WARP_MERCATOR = 9, MO_WARP_TRANSVERSE_MERCATOR = 10, MO_WARP_SPHERICAL_PORTRAIT = 11,
       MO_WARP_CYLINDRICAL_PORTRAIT = 12, MO_WARP_PLANE_PORTRAIT = 13, MO_WARP_KINDS = 14 };
/* The A, B parameters of the compressed-rectilinear and Panini projectors travel in bits 8-9 of the kind:
 * 0 = the constructors' defaults (1, 1); 1 = (2, 1) and 2 = (1.5, 1), the two the command line offers
 * (CLI:561-568); 3 = the pair last given to mo_set_custom_ab. */
#define MO_WARP_AB(kind, preset) ((kind) | ((preset) << 8))
void mo_set_custom_ab(float a, float b);
enum { MO_INTER_NEAREST = 0, MO_INTER_LINEAR = 1 };
enum { MO_BORDER_CONSTANT = 0, MO_BORDER_REPLICATE = 1, MO_BORDER_REFLECT = 2,
       MO_BORDER_WRAP = 3, MO_BORDER_REFLECT_101 = 4 };
/* R64/include/opencv2/stitching/detail/blenders.hpp:58 */
enum { MO_BLEND_NO = 0, MO_BLEND_FEATHER = 1, MO_BLEND_MULTI_BAND = 2 };
/* R64/include/opencv2/stitching/detail/exposure_compensate.hpp:56 */
enum { MO_EXPOS_NO = 0, MO_EXPOS_GAIN = 1, MO_EXPOS_GAIN_BLOCKS = 2 };
enum { MO_WEIGHT_32F = 0, MO_WEIGHT_16S = 1 };

typedef struct { int x, y, width, height; } mo_rect;

/* ProjectorBase state, R64/include/opencv2/stitching/detail/warpers.hpp:75-86 */
typedef struct {
    float scale;
    float k[9];
    float rinv[9];
    float r_kinv[9];
    float k_rinv[9];
    float t[3];
    float a, b;               /* CompressedRectilinear* / Panini* projectors */
} mo_projector;

/* worker threads for the row-parallel loops (OpenMP); results do not depend on it */
void mo_set_num_threads(int n);
int  mo_get_num_threads(void);

/* ---- scalar conversion helpers (x64/SSE2 semantics) ------------------------ */
int      mo_cv_round(double v);            /* cvRound: RNE, INT_MIN when unrepresentable */
int      mo_cv_floor(double v);
int16_t  mo_sat_s16(int v);
uint8_t  mo_sat_u8(int v);
int16_t  mo_trunc_s16(float v);            /* static_cast<short>(float) as MSVC x64 emits it */
int      mo_border_interpolate(int p, int len, int border_type);

/* ---- camera / projector (a1-a5) ------------------------------------------- */
void mo_camera_K(double focal, double aspect, double ppx, double ppy, float K[9]);
void mo_invert3x3_32f(const float m[9], float out[9], int *ok);
void mo_matmul3x3_32f(const float a[9], const float b[9], float out[9]);
void mo_set_camera_params(mo_projector *p, const float K[9], const float R[9], const float T[3]);
void mo_map_forward(const mo_projector *p, int kind, float x, float y, float *u, float *v);
void mo_map_backward(const mo_projector *p, int kind, float u, float v, float *x, float *y);
/* inclusive tl / br, exactly as RotationWarperBase::detectResultRoi returns them */
void mo_detect_result_roi(const mo_projector *p, int kind, int src_w, int src_h,
                          int *tl_x, int *tl_y, int *br_x, int *br_y);
/* RotationWarper::warpRoi: Rect(tl, br+1) */
mo_rect mo_warp_roi(int kind, float scale, const float K[9], const float R[9], const float T[3],
                    int src_w, int src_h);
/* RotationWarper::buildMaps: xmap/ymap must hold roi.width*roi.height floats of warpRoi() */
mo_rect mo_build_maps(int kind, float scale, const float K[9], const float R[9], const float T[3],
                      int src_w, int src_h, float *xmap, float *ymap);

/* ---- cv::remap (a6) --------------------------------------------------------- */
void mo_remap_8u(const uint8_t *src, int sw, int sh, size_t sstride, int cn,
                 uint8_t *dst, int dw, int dh, size_t dstride,
                 const float *xmap, const float *ymap, size_t map_stride_elems,
                 int interp, int border_type);
/* RotationWarper::warp = buildMaps + remap; dst is warpRoi() sized, tightly packed */
mo_rect mo_warp_8u(int kind, float scale, const float K[9], const float R[9], const float T[3],
                   const uint8_t *src, int sw, int sh, size_t sstride, int cn,
                   int interp, int border_type, uint8_t *dst);
const int16_t *mo_bilinear_tab_i(void);   /* [1024][4] fixed-point weights, for KATs */

/* ---- lens undistort (a22): initUndistortRectifyMap(CV_16SC2) + remap with fixed-point maps ------ */
/* K, newK: 3x3 row-major CV_64F; dist: 0, 4, 5 or 8 coefficients (k1 k2 p1 p2 [k3 [k4 k5 k6]]); R: 3x3 or NULL
 * (identity).  map1: height x width x 2 shorts, map2: height x width ushorts, tightly packed.  -1 on a bad shape. */
int mo_init_undistort_rectify_map(const double K[9], const double *dist, int ndist, const double *R,
                                  const double newK[9], int width, int height, int16_t *map1, uint16_t *map2);
/* remap(src, dst, map1, map2, INTER_LINEAR) with the default BORDER_CONSTANT / Scalar() border (DRV:741) */
void mo_remap_8u_fixed(const uint8_t *src, int sw, int sh, size_t sstride, int cn,
                       uint8_t *dst, int dw, int dh, size_t dstride,
                       const int16_t *map1, const uint16_t *map2);

/* SURVEY 8 f1 as written: "composing the distortion model into the warp's backward map" -- ONE resampling of the raw
 * frame.  This is NOT what the reference computes (it resamples twice, DRV:741 then DRV:752); it is this project's
 * definition of the fused variant and is its own parity target:
 *   (u, v)  = the warp's backward map at the ROI pixel (float, as buildMaps stores it), clamped to the frame;
 *   L(u, v) = the lens maps (map1 + fractions of map2 / 32, as floats) interpolated bilinearly at (u, v):
 *             x0 = min((int)u, w - 2), ax = u - x0;  r0 = m00 * (1 - ax) + m10 * ax;  r1 = m01 * (1 - ax) + m11 * ax;
 *             L = r0 * (1 - ay) + r1 * ay      (float32, no contraction);
 *   the composed maps (Lx, Ly) then go through the warp's own remap(INTER_LINEAR, BORDER_REFLECT) of the RAW frame.
 * xmap / ymap: n floats each, updated in place.  Frames of one pixel in a dimension use that row / column alone. */
void mo_compose_lens_maps(float *xmap, float *ymap, size_t n, const int16_t *map1, const uint16_t *map2, int w, int h);

/* ---- NV12 ingest (optional): cv::cvtColor(yuv, bgr, COLOR_YUV2BGR_NV12) of OpenCV 2.4.11
 * (imgproc/src/color.cpp, YUV420sp2RGB888Invoker<bIdx = 0, uIdx = 0>): BT.601 limited range in 20-bit fixed point.
 * The reference itself is BGR-only (stitching_sharemem.cpp:112-115); NV12 is what hardware decoders deliver.
 * yuv: (h * 3 / 2) rows of `stride` bytes -- h rows of Y, then h / 2 rows of interleaved U V; w and h even. */
void mo_cvt_nv12_bgr(const uint8_t *yuv, int w, int h, size_t stride, uint8_t *bgr, size_t bgr_stride);

/* ---- exposure (a7) ---------------------------------------------------------- */
void mo_resize_linear_32f(const float *src, int sw, int sh, float *dst, int dw, int dh);
void mo_blocks_gain_apply(uint8_t *img, int w, int h, size_t stride,
                          const float *gain_map, int gw, int gh);
void mo_gain_apply(uint8_t *img, int w, int h, size_t stride, int cn, double gain);

/* exposure ESTIMATION (calibration side): GainCompensator::feed / BlocksGainCompensator::feed
 * (exposure_compensate.cpp:73-147, 165-222).  Images are CV_8UC3, masks CV_8U (level 255), corners_xy / sizes_wh
 * hold n (x, y) / (w, h) pairs. */
int  mo_solve_lu_64f(const double *A, const double *b, int n, double *x);
void mo_smooth_gain_map(float *m, int w, int h);
void mo_gain_feed(int n, const int *corners_xy, const uint8_t *const *images, const size_t *image_strides,
                  const int *sizes_wh, const uint8_t *const *masks, const size_t *mask_strides, double *gains);
void mo_gain_map_size(int w, int h, int bl_width, int bl_height, int *gw, int *gh);
void mo_blocks_gain_feed(int n, const int *corners_xy, const uint8_t *const *images, const size_t *image_strides,
                         const int *sizes_wh, const uint8_t *const *masks, const size_t *mask_strides,
                         int bl_width, int bl_height, float *const *gain_maps);

/* ---- core/imgproc primitives used by the blenders --------------------------- */
void mo_convert_8u_16s(const uint8_t *src, size_t sstride, int16_t *dst, size_t dstride_elems,
                       int w_elems, int h);
void mo_convert_16s_8u(const int16_t *src, size_t sstride_elems, uint8_t *dst, size_t dstride,
                       int w_elems, int h);
void mo_copy_make_border_16s(const int16_t *src, int w, int h, int cn, int16_t *dst,
                             int top, int bottom, int left, int right, int border_type);
void mo_copy_make_border_32f(const float *src, int w, int h, float *dst,
                             int top, int bottom, int left, int right, int border_type);
void mo_pyr_down_16s(const int16_t *src, int w, int h, int cn, int16_t *dst);   /* dst ((w+1)/2,(h+1)/2) */
void mo_pyr_down_32f(const float *src, int w, int h, int cn, float *dst);
void mo_pyr_up_16s(const int16_t *src, int sw, int sh, int cn, int16_t *dst, int dw, int dh);
void mo_add_16s(const int16_t *a, const int16_t *b, int16_t *dst, size_t n);
void mo_subtract_16s(const int16_t *a, const int16_t *b, int16_t *dst, size_t n);
void mo_distance_transform_l1(const uint8_t *mask, int w, int h, float *dist);

/* ---- seam-mask fix-up (a9: stitcher.cpp:291-294) ------------------------------ */
void mo_dilate3x3_8u(const uint8_t *src, int w, int h, uint8_t *dst);
void mo_resize_linear_8u(const uint8_t *src, int sw, int sh, uint8_t *dst, int dw, int dh);
void mo_resize_linear_8u_cn(const uint8_t *src, int sw, int sh, int cn, uint8_t *dst, int dw, int dh, double scale_x, double scale_y);
void mo_mask_fixup(const uint8_t *seam_mask, int sw, int sh, const uint8_t *mask_warped, int w, int h, uint8_t *out);

/* ---- VoronoiSeamFinder (seam_finders.cpp:74-85, 108-160): masks tightly packed, updated in place ---- */
void mo_voronoi_seams(int n, const int *corners_xy, const int *sizes_wh, uint8_t *const *masks);
/* ---- DpSeamFinder (seam_finders.cpp:163-1045): images CV_32FC3 (is_float) or CV_8UC3, BGR, tight; cost_func 0 = COLOR,
 * 1 = COLOR_GRAD; masks tightly packed, updated in place ---- */
void mo_dp_seams(int n, const int *corners_xy, const int *sizes_wh, const void *const *images, int is_float, int cost_func,
                 uint8_t *const *masks);
/* DpSeamFinder::computeGradients (seam_finders.cpp:519-543) of one image: BGR2GRAY + Sobel(CV_32F, ksize 3) x 2 -> h x w floats */
void mo_dp_gradients(const void *image, int is_float, int w, int h, float *gx, float *gy);

/* ---- util (a10) -------------------------------------------------------------- */
mo_rect mo_result_roi(const int *corners_xy, const int *sizes_wh, int n);
int mo_overlap_roi(int tl1x, int tl1y, int tl2x, int tl2y, int w1, int h1, int w2, int h2, mo_rect *roi);

/* ---- blenders (a11, a13-a21) -------------------------------------------------- */
void mo_normalize_using_weight_map_32f(const float *weight, int16_t *src, int w, int h);
void mo_normalize_using_weight_map_16s(const int16_t *weight, int16_t *src, int w, int h);
void mo_create_weight_map(const uint8_t *mask, int w, int h, float sharpness, float *weight);
/* pyr[i] must hold level sizes ((w_i+1)/2 chain) * cn shorts; pyr[0] is overwritten in place */
void mo_create_laplace_pyr_16s(int16_t **pyr, int w, int h, int cn, int num_levels);
void mo_restore_image_from_laplace_pyr(int16_t **pyr, int w, int h, int cn, int num_levels);

typedef struct mo_blender mo_blender;
mo_blender *mo_blender_create(int kind, int num_bands, float sharpness, int weight_type);
void mo_blender_destroy(mo_blender *b);
void mo_blender_prepare(mo_blender *b, const int *corners_xy, const int *sizes_wh, int n);
void mo_blender_prepare_rect(mo_blender *b, mo_rect dst_roi);
/* img: CV_16SC3 tightly packed w*h*3, mask: CV_8U tightly packed w*h */
void mo_blender_feed(mo_blender *b, const int16_t *img, const uint8_t *mask, int w, int h,
                     int tl_x, int tl_y);
/* returns result size; *dst (16SC3) and *dst_mask (8U) are malloc'd, caller frees with mo_free */
mo_rect mo_blender_blend(mo_blender *b, int16_t **dst, uint8_t **dst_mask);
void mo_free(void *p);
/* introspection for stage-level parity tests (valid between feed and blend) */
int  mo_blender_num_bands(const mo_blender *b);
mo_rect mo_blender_dst_roi(const mo_blender *b);
const int16_t *mo_blender_band(const mo_blender *b, int level, int *w, int *h);
const float   *mo_blender_band_weight(const mo_blender *b, int level, int *w, int *h);
/* MultiBandBlender::feed ROI alignment only (a15): out[8] = tl_new.x, tl_new.y, br_new.x,
 * br_new.y, top, left, bottom, right */
void mo_multiband_feed_geometry(mo_rect dst_roi, int num_bands, int tl_x, int tl_y, int w, int h,
                                int out[8]);

/* ---- whole compose loop (stitcher.cpp:264-313 / CLI:705-760) ------------------- */
typedef struct {
    int   warp_kind;
    float scale;                 /* warped_image_scale */
    int   blend_kind;
    int   num_bands;             /* MultiBandBlender::setNumBands */
    float sharpness;             /* FeatherBlender::setSharpness */
    int   weight_type;
    int   expos_kind;
    int   num_threads;           /* OpenMP threads over cameras for warp/gain (DRV:736,748); 1 = serial */
} mo_compose_config;

typedef struct {
    float K[9], R[9], T[3];
    const uint8_t *frame;        /* CV_8UC3 BGR */
    int src_w, src_h; size_t src_stride;
    const float *gain_map; int gain_w, gain_h;   /* GAIN_BLOCKS */
    double gain;                                 /* GAIN */
    const uint8_t *mask;         /* final mask_warped (stitcher.cpp:294), warpRoi sized, tight */
    /* optional: the camera's lens maps (src_w x src_h), composed into the warp's map (mo_compose_lens_maps); the frame is RAW */
    const int16_t *lens_map1; const uint16_t *lens_map2;
} mo_compose_camera;

/* pano (CV_8UC3) and pano_mask (CV_8U) malloc'd; free with mo_free.  Returns canvas rect. */
mo_rect mo_compose_panorama(const mo_compose_config *cfg, const mo_compose_camera *cams, int n,
                            uint8_t **pano, uint8_t **pano_mask);

/* the reference driver's own fast path (DRV:94-177): hard-seam LUT composite */
typedef struct mo_lut mo_lut;
mo_lut *mo_lut_build(const int *corners_xy, const int *sizes_wh, const uint8_t *const *masks, int n);
mo_rect mo_lut_canvas(const mo_lut *l);
/* feedSizeRemap + convertTo(8U); imgs are per-camera CV_16SC3 warped images; crop margins as DRV:47 */
mo_rect mo_lut_compose(const mo_lut *l, const int16_t *const *imgs, float upblack, float downblack,
                       float leftblack, float rightblack, uint8_t **pano);
void mo_lut_destroy(mo_lut *l);

#ifdef __cplusplus
}
#endif
#endif /* MST_ORACLE_H */
