// geometry.hpp -- host-side, per-calibration geometry of the compositing engine.
//
// Everything here runs once per mst_update on the host CPU (the north star keeps
// registration-time work on the host); nothing here touches pixels.  It produces the small
// tables the HIP kernels consume so that the backward map and the gain field are evaluated
// on the device, bit-exactly, from per-column / per-row constants (the multi-band kernel additionally
// caches the resulting fixed-point coordinates per calibration, see kernels_mb.hip).
//
// Reference behaviour followed (R64 = /root/reference/stitching/OpenCV2.4.11-Stitching-64bit/
// OpenCV2.4.11-Stitching):
//   Projector::set             ProjectorBase::setCameraParams      R64/src/warpers.cpp:50-78
//   Projector::forward         {Plane,Spherical,Cylindrical}Projector::mapForward
//                                                                  detail/warpers_inl.hpp:206-300
//   detect_result_roi          detectResultRoi / ...ByBorder       detail/warpers_inl.hpp:143-203,
//                              PlaneWarper / SphericalWarper        R64/src/warpers.cpp:139-212
//   result_roi                 resultRoi                            R64/src/util.cpp:127-140
//   multiband_prepare/feed     MultiBandBlender::prepare / feed     R64/src/blenders.cpp:203-269
//   build_gain_axis            cv::resize(INTER_LINEAR) index/weight tables as used by
//                              BlocksGainCompensator::apply         R64/src/exposure_compensate.cpp:233
#pragma once

#include <cstdint>
#include <vector>

namespace mst {

struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    int br_x() const { return x + width; }
    int br_y() const { return y + height; }
};

// MST_WARP_* of the public header.  Plane, spherical and cylindrical maps separate into per-column and per-row
// terms (MapTables below) and are evaluated on the device; the other eleven projectors of
// detail/warpers_inl.hpp:302-759 need libm per pixel, so their backward maps are evaluated on the host once per
// calibration -- RotationWarperBase::buildMaps, detail/warpers_inl.hpp:62-85 -- and the kernels read them.
enum WarpKind {
    kPlane = 0, kSpherical = 1, kCylindrical = 2,
    kFisheye = 3, kStereographic = 4, kCompressedPlane = 5, kCompressedPlanePortrait = 6, kPanini = 7,
    kPaniniPortrait = 8, kMercator = 9, kTransverseMercator = 10, kSphericalPortrait = 11,
    kCylindricalPortrait = 12, kPlanePortrait = 13, kWarpKinds = 14
};
inline bool separable_map(int kind) { return kind <= kCylindrical; }

struct Projector {
    int kind = kSpherical;
    float scale = 1.f;
    float a = 1.f, b = 1.f;           // CompressedRectilinear* / Panini* (detail/warpers.hpp:226-305)
    float k[9], rinv[9], r_kinv[9], k_rinv[9], t[3];

    void set(const float K[9], const float R[9], const float T[3]);
    void forward(float x, float y, float& u, float& v) const;
    void backward(float u, float v, float& x, float& y) const;     // kinds >= kFisheye only
};

// kind = MST_WARP_* with the A/B preset in bits 8-9 (include/mi355_stitch.h); custom_a/b serve preset 3
void set_projector_kind(Projector& p, int kind, float custom_a, float custom_b);

// RotationWarperBase::buildMaps for the non-separable projectors: roi.width x roi.height floats each, rows in
// parallel on the host's cores.
void build_maps_host(const Projector& p, const Rect& roi, float* xmap, float* ymap);

// inclusive top-left / bottom-right, as the reference's detectResultRoi returns them
void detect_result_roi(const Projector& p, int src_w, int src_h, int& tl_x, int& tl_y, int& br_x, int& br_y);
Rect warp_roi(const Projector& p, int src_w, int src_h);   // Rect(tl, br + 1)
Rect result_roi(const std::vector<Rect>& rois);

// Per-axis tables of the backward map.  For destination column u (= roi.x + i) and row v
// (= roi.y + j) the kernels form the unit ray / plane point from
//   plane:        col_a[i] = u/scale - t0                      row_a[j] = v/scale - t1
//   spherical:    col_a[i] = sinf(u/scale), col_b[i] = cosf(u/scale)
//                 row_a[j] = sinf(pi - v/scale), row_b[j] = cosf(pi - v/scale)
//   cylindrical:  col_a[i] = sinf(u/scale), col_b[i] = cosf(u/scale)   row_a[j] = v/scale
// and finish with k_rinv multiply-adds and the perspective divide in IEEE float, in the
// operation order of mapBackward (detail/warpers_inl.hpp:221-300).
struct MapTables {
    std::vector<float> col_a, col_b, row_a, row_b;
};
MapTables build_map_tables(const Projector& p, const Rect& roi);

// Host evaluation of the same table-based backward map (mst_build_maps_host).
void backward_map_host(const Projector& p, const Rect& roi, const MapTables& t, float* xmap, float* ymap);

// cv::resize(..., INTER_LINEAR) tables for one axis, float source (resizeGeneric_ set-up code).
// ofs0/ofs1 are the two source indices (already clamped), w0/w1 their weights.
struct ResizeAxis {
    std::vector<int32_t> ofs0, ofs1;
    std::vector<float> w0, w1;
};
ResizeAxis build_gain_axis_x(int src_len, int dst_len);
ResizeAxis build_gain_axis_y(int src_len, int dst_len);

// cv::resize(..., INTER_LINEAR) tables for one axis of a CV_8U image (HResizeLinear<uchar,int,short,2048>):
// the two source indices and their 11-bit fixed-point weights, saturate_cast<short>(w * 2048).
struct Resize8uAxis {
    std::vector<int32_t> ofs0, ofs1;
    std::vector<int16_t> c0, c1;
};
Resize8uAxis build_resize8u_axis(int src_len, int dst_len, bool is_x);

// MultiBandBlender integer geometry.
struct BandGeometry {
    Rect tl_br;                 // x,y = tl_new ; width,height = br_new - tl_new (bordered size)
    int top = 0, left = 0, bottom = 0, right = 0;
};
// Lens undistortion: cv::initUndistortRectifyMap(K, dist, R, newK, size, CV_16SC2, map1, map2)
// as the reference's RunTestEyeFishMap calls it (DRV:201-238).  K, newK, R (or nullptr = identity): 3x3 row-major
// doubles; dist: 0, 4, 5 or 8 coefficients k1 k2 p1 p2 [k3 [k4 k5 k6]].  map1: h x w x 2 shorts (integer source
// coordinate), map2: h x w ushorts ((fy << 5) | fx, the row of cv::remap's bilinear weight table).  False if
// the coefficient count is not one OpenCV accepts.
// MST_FLAG_LENS_COMPOSED: the lens maps interpolated at the warp's backward map (mi355_stitch.h: mst_compose_lens_maps)
void compose_lens_maps(float* xmap, float* ymap, size_t n, const int16_t* map1, const uint16_t* map2, int w, int h);
bool init_undistort_rectify_map(const double K[9], const double* dist, int ndist, const double* R, const double newK[9],
                                int width, int height, int16_t* map1, uint16_t* map2);

int  multiband_num_bands(int actual_num_bands, const Rect& dst_roi);     // blenders.cpp:207-209
Rect multiband_padded_roi(const Rect& dst_roi, int num_bands);           // blenders.cpp:211-213
BandGeometry multiband_feed_geometry(const Rect& padded_roi, int num_bands, const Rect& image);

}  // namespace mst
