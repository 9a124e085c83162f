// geometry.cpp -- see geometry.hpp for the reference lines each function follows.
#include "geometry.hpp"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstring>
#include <thread>

namespace mst {

namespace {

constexpr double kPi = 3.1415926535897932384626433832795;   // CV_PI

// cv::Mat::inv() for a 3x3 CV_32F matrix (closed form of cv::invert): determinant and adjugate terms in double (every operand
// widened first -- read off the reference's shipped opencv_core2411.dll in round 5, oracle/REFERENCE_BINARY_AUDIT.md; rounds 1-4 had
// the determinant in float), each term scaled by 1/det and rounded to float.
void invert3x3(const float m[9], float out[9])
{
    auto S = [&](int r, int c) { return m[r * 3 + c]; };
    auto D = [&](int r, int c) { return static_cast<double>(m[r * 3 + c]); };
    double det = D(0, 0) * (D(1, 1) * D(2, 2) - D(1, 2) * D(2, 1)) -
                 D(0, 1) * (D(1, 0) * D(2, 2) - D(1, 2) * D(2, 0)) +
                 D(0, 2) * (D(1, 0) * D(2, 1) - D(1, 1) * D(2, 0));
    if (det == 0.) {
        std::memset(out, 0, 9 * sizeof(float));
        return;
    }
    const double d = 1. / det;
    float t[9];
    t[0] = static_cast<float>((D(1, 1) * S(2, 2) - D(1, 2) * S(2, 1)) * d);
    t[1] = static_cast<float>((D(0, 2) * S(2, 1) - D(0, 1) * S(2, 2)) * d);
    t[2] = static_cast<float>((D(0, 1) * S(1, 2) - D(0, 2) * S(1, 1)) * d);
    t[3] = static_cast<float>((D(1, 2) * S(2, 0) - D(1, 0) * S(2, 2)) * d);
    t[4] = static_cast<float>((D(0, 0) * S(2, 2) - D(0, 2) * S(2, 0)) * d);
    t[5] = static_cast<float>((D(0, 2) * S(1, 0) - D(0, 0) * S(1, 2)) * d);
    t[6] = static_cast<float>((D(1, 0) * S(2, 1) - D(1, 1) * S(2, 0)) * d);
    t[7] = static_cast<float>((D(0, 1) * S(2, 0) - D(0, 0) * S(2, 1)) * d);
    t[8] = static_cast<float>((D(0, 0) * S(1, 1) - D(0, 1) * S(1, 0)) * d);
    std::memcpy(out, t, sizeof(t));
}

// cv::Mat operator* on 3x3 CV_32F (gemm small-matrix path): float products, summed in order.
void mul3x3(const float a[9], const float b[9], float out[9])
{
    float t[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            t[r * 3 + c] = a[r * 3] * b[c] + a[r * 3 + 1] * b[3 + c] + a[r * 3 + 2] * b[6 + c];
    std::memcpy(out, t, sizeof(t));
}

struct Bounds {
    float tl_u = FLT_MAX, tl_v = FLT_MAX, br_u = -FLT_MAX, br_v = -FLT_MAX;
    void add(float u, float v)
    {
        tl_u = std::min(tl_u, u); tl_v = std::min(tl_v, v);
        br_u = std::max(br_u, u); br_v = std::max(br_v, v);
    }
};

Bounds border_walk(const Projector& p, int w, int h)
{
    Bounds b;
    float u, v;
    for (float x = 0; x < w; ++x) {
        p.forward(x, 0, u, v);                          b.add(u, v);
        p.forward(x, static_cast<float>(h - 1), u, v);  b.add(u, v);
    }
    for (int y = 0; y < h; ++y) {
        p.forward(0, static_cast<float>(y), u, v);                         b.add(u, v);
        p.forward(static_cast<float>(w - 1), static_cast<float>(y), u, v); b.add(u, v);
    }
    return b;
}

int floor_to_int(double v)
{
    double r = std::floor(v);
    if (!(r >= -2147483648.0 && r <= 2147483647.0)) return INT_MIN;
    return static_cast<int>(r);
}

}  // namespace

void Projector::set(const float K[9], const float R[9], const float T[3])
{
    std::memcpy(k, K, sizeof(k));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            rinv[r * 3 + c] = R[c * 3 + r];
    float kinv[9];
    invert3x3(K, kinv);
    mul3x3(R, kinv, r_kinv);
    mul3x3(K, rinv, k_rinv);
    for (int i = 0; i < 3; ++i) t[i] = T ? T[i] : 0.f;
}

namespace {

// The ray through source pixel (x, y), R * K^-1 * (x, y, 1); the portrait projectors read it with the first two
// components exchanged.
struct Ray { float x, y, z; };
inline Ray ray_through(const float m[9], float x, float y)
{
    Ray r;
    r.x = m[0] * x + m[1] * y + m[2];
    r.y = m[3] * x + m[4] * y + m[5];
    r.z = m[6] * x + m[7] * y + m[8];
    return r;
}
inline float norm3(const Ray& r) { return sqrtf(r.x * r.x + r.y * r.y + r.z * r.z); }

}  // namespace

// detail/warpers_inl.hpp:302-752 (forward maps of the eleven non-separable projectors); float libm calls exactly
// where the reference has them, the double ones (cos, sin, fabs without suffix) likewise
static void forward_other(const Projector& p, float x, float y, float& u, float& v)
{
    const float scale = p.scale, a = p.a, b = p.b;
    Ray r = ray_through(p.r_kinv, x, y);
    const bool swapped_in = p.kind == kCompressedPlanePortrait || p.kind == kPaniniPortrait;
    if (swapped_in) std::swap(r.x, r.y);
    switch (p.kind) {
    case kFisheye: {
        const float lon = atan2f(r.x, r.z);
        const float lat = static_cast<float>(kPi) - acosf(r.y / norm3(r));
        u = scale * lat * cosf(lon);
        v = scale * lat * sinf(lon);
        return;
    }
    case kStereographic: {
        const float lon = atan2f(r.x, r.z);
        const float lat = static_cast<float>(kPi) - acosf(r.y / norm3(r));
        const float rad = sinf(lat) / (1 - cosf(lat));
        u = static_cast<float>(scale * rad * std::cos(static_cast<double>(lon)));
        v = static_cast<float>(scale * rad * std::sin(static_cast<double>(lon)));
        return;
    }
    case kCompressedPlane:
    case kCompressedPlanePortrait: {
        const float lon = atan2f(r.x, r.z);
        const float lat = asinf(r.y / norm3(r));
        u = p.kind == kCompressedPlane ? scale * a * tanf(lon / a) : -scale * a * tanf(lon / a);
        v = scale * b * tanf(lat) / cosf(lon);
        return;
    }
    case kPanini:
    case kPaniniPortrait: {
        const float lon = atan2f(r.x, r.z);
        const float lat = asinf(r.y / norm3(r));
        const float tg = a * tanf(lon / a);
        u = p.kind == kPanini ? scale * tg : -scale * tg;
        const float sinu = sinf(lon);
        if (std::fabs(static_cast<double>(sinu)) < 1E-7) v = scale * b * tanf(lat);
        else v = scale * b * tg * tanf(lat) / sinu;
        return;
    }
    case kMercator: {
        const float lon = atan2f(r.x, r.z);
        const float lat = asinf(r.y / norm3(r));
        u = scale * lon;
        v = scale * logf(tanf(static_cast<float>(kPi / 4) + lat / 2));
        return;
    }
    case kTransverseMercator: {
        const float lon = atan2f(r.x, r.z);
        const float lat = asinf(r.y / norm3(r));
        const float B = cosf(lat) * sinf(lon);
        u = scale / 2 * logf((1 + B) / (1 - B));
        v = scale * atan2f(tanf(lat), cosf(lon));
        return;
    }
    default:
        break;
    }
    // the three portrait variants of the separable projectors: components exchanged, u negated
    std::swap(r.x, r.y);
    float pu, pv;
    if (p.kind == kSphericalPortrait) {
        pu = scale * atan2f(r.x, r.z);
        pv = scale * (static_cast<float>(kPi) - acosf(r.y / norm3(r)));
    } else if (p.kind == kCylindricalPortrait) {
        pu = scale * atan2f(r.x, r.z);
        pv = scale * r.y / sqrtf(r.x * r.x + r.z * r.z);
    } else {
        const float px = p.t[0] + r.x / r.z * (1 - p.t[2]);
        const float py = p.t[1] + r.y / r.z * (1 - p.t[2]);
        pu = scale * px;
        pv = scale * py;
    }
    u = -pu;
    v = pv;
}

// detail/warpers_inl.hpp:318-773 (backward maps)
void Projector::backward(float u, float v, float& x, float& y) const
{
    const float* kr = k_rinv;
    if (kind == kPlanePortrait) {
        const float pu = -u / scale - t[0];
        const float pv = v / scale - t[1];
        x = kr[0] * pv + kr[1] * pu + kr[2] * (1 - t[2]);
        y = kr[3] * pv + kr[4] * pu + kr[5] * (1 - t[2]);
        const float z = kr[6] * pv + kr[7] * pu + kr[8] * (1 - t[2]);
        x /= z;
        y /= z;
        return;
    }
    Ray r{0.f, 0.f, 0.f};
    auto from_lon_lat = [&](float lon, float lat, bool portrait) {      // unit ray of (longitude, latitude)
        const float cl = cosf(lat);
        const float along = cl * sinf(lon), up = sinf(lat);
        r.x = portrait ? up : along;
        r.y = portrait ? along : up;
        r.z = cl * cosf(lon);
    };
    auto from_polar = [&](float lon, float colat) {                      // fisheye family: angle from the -y pole
        const float s = sinf(static_cast<float>(kPi) - colat);
        r.x = s * sinf(lon);
        r.y = cosf(static_cast<float>(kPi) - colat);
        r.z = s * cosf(lon);
    };
    switch (kind) {
    case kFisheye: {
        u /= scale; v /= scale;
        const float lon = atan2f(v, u);
        from_polar(lon, sqrtf(u * u + v * v));
        break;
    }
    case kStereographic: {
        u /= scale; v /= scale;
        const float lon = atan2f(v, u);
        const float rad = sqrtf(u * u + v * v);
        from_polar(lon, 2 * atanf(1.f / rad));
        break;
    }
    case kCompressedPlane:
    case kCompressedPlanePortrait: {
        const bool portrait = kind == kCompressedPlanePortrait;
        u /= portrait ? -scale : scale;
        v /= scale;
        const float lon = a * atanf(u / a);
        from_lon_lat(lon, atanf(v * cosf(lon) / b), portrait);
        break;
    }
    case kPanini:
    case kPaniniPortrait: {
        const bool portrait = kind == kPaniniPortrait;
        u /= portrait ? -scale : scale;
        v /= scale;
        const float lon = a * atanf(u / a);
        float lat;
        if (std::fabs(static_cast<double>(lon)) > 1E-7) lat = atanf(v * sinf(lon) / (b * a * tanf(lon / a)));
        else lat = atanf(v / b);
        from_lon_lat(lon, lat, portrait);
        break;
    }
    case kMercator: {
        u /= scale; v /= scale;
        from_lon_lat(u, atanf(sinhf(v)), false);
        break;
    }
    case kTransverseMercator: {
        u /= scale; v /= scale;
        const float lat = asinf(sinf(v) / coshf(u));
        const float lon = atan2f(sinhf(u), static_cast<float>(std::cos(static_cast<double>(v))));
        from_lon_lat(lon, lat, false);
        break;
    }
    case kSphericalPortrait: {
        const float pu = -u / scale, pv = v / scale;
        const float s = sinf(static_cast<float>(kPi) - pv);
        r.y = s * sinf(pu);                                   // x0_ -> y_
        r.x = cosf(static_cast<float>(kPi) - pv);             // y0_ -> x_
        r.z = s * cosf(pu);
        break;
    }
    default: {  // kCylindricalPortrait
        const float pu = -u / scale, pv = v / scale;
        r.y = sinf(pu);
        r.x = pv;
        r.z = cosf(pu);
        break;
    }
    }
    x = kr[0] * r.x + kr[1] * r.y + kr[2] * r.z;
    y = kr[3] * r.x + kr[4] * r.y + kr[5] * r.z;
    const float z = kr[6] * r.x + kr[7] * r.y + kr[8] * r.z;
    if (z > 0) { x /= z; y /= z; }
    else x = y = -1;
}

void set_projector_kind(Projector& p, int kind, float custom_a, float custom_b)
{
    p.kind = kind & 0xFF;
    switch ((kind >> 8) & 3) {
    case 1:  p.a = 2.f;  p.b = 1.f; break;          // "...A2B1"   (CLI:561,563,565,567)
    case 2:  p.a = 1.5f; p.b = 1.f; break;          // "...A1.5B1" (CLI:562,564,566,568)
    case 3:  p.a = custom_a; p.b = custom_b; break;
    default: p.a = 1.f;  p.b = 1.f; break;          // float A = 1, float B = 1
    }
}

void build_maps_host(const Projector& p, const Rect& roi, float* xmap, float* ymap)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const int workers = static_cast<int>(std::min<unsigned>(hw, 64u));
    auto rows = [&](int j0, int j1) {
        for (int j = j0; j < j1; ++j)
            for (int i = 0; i < roi.width; ++i) {
                float x, y;
                p.backward(static_cast<float>(roi.x + i), static_cast<float>(roi.y + j), x, y);
                xmap[static_cast<size_t>(j) * roi.width + i] = x;
                ymap[static_cast<size_t>(j) * roi.width + i] = y;
            }
    };
    if (workers == 1 || roi.height < 64) { rows(0, roi.height); return; }
    std::vector<std::thread> pool;
    const int per = (roi.height + workers - 1) / workers;
    for (int w = 0; w < workers; ++w) {
        const int j0 = w * per, j1 = std::min(roi.height, j0 + per);
        if (j0 < j1) pool.emplace_back(rows, j0, j1);
    }
    for (std::thread& t : pool) t.join();
}

void Projector::forward(float x, float y, float& u, float& v) const
{
    if (kind >= kFisheye) { forward_other(*this, x, y, u, v); return; }
    float x_ = r_kinv[0] * x + r_kinv[1] * y + r_kinv[2];
    float y_ = r_kinv[3] * x + r_kinv[4] * y + r_kinv[5];
    float z_ = r_kinv[6] * x + r_kinv[7] * y + r_kinv[8];
    switch (kind) {
    case kPlane:
        x_ = t[0] + x_ / z_ * (1 - t[2]);
        y_ = t[1] + y_ / z_ * (1 - t[2]);
        u = scale * x_;
        v = scale * y_;
        break;
    case kSpherical: {
        u = scale * atan2f(x_, z_);
        float w = y_ / sqrtf(x_ * x_ + y_ * y_ + z_ * z_);
        v = scale * (static_cast<float>(kPi) - acosf(w == w ? w : 0));
        break;
    }
    default:
        u = scale * atan2f(x_, z_);
        v = scale * y_ / sqrtf(x_ * x_ + z_ * z_);
        break;
    }
}

void detect_result_roi(const Projector& p, int src_w, int src_h, int& tl_x, int& tl_y, int& br_x, int& br_y)
{
    Bounds b;
    if (p.kind == kSphericalPortrait || p.kind == kCylindricalPortrait || p.kind == kPlanePortrait) {
        // detectResultRoiByBorder (detail/warpers.hpp:479-502); the spherical one then adds the poles it contains,
        // which lie on the x axis for this projector (warpers.cpp:389-430)
        b = border_walk(p, src_w, src_h);
        if (p.kind == kSphericalPortrait) {
            float tl_uf = static_cast<float>(static_cast<int>(b.tl_u)), tl_vf = static_cast<float>(static_cast<int>(b.tl_v));
            float br_uf = static_cast<float>(static_cast<int>(b.br_u)), br_vf = static_cast<float>(static_cast<int>(b.br_v));
            for (int pole = 0; pole < 2; ++pole) {
                const float x = p.rinv[0], y = pole == 0 ? p.rinv[3] : -p.rinv[3], z = p.rinv[6];
                if (!(y > 0.f)) continue;
                const float x_ = (p.k[0] * x + p.k[1] * y) / z + p.k[2];
                const float y_ = p.k[4] * y / z + p.k[5];
                if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                    const float pv = pole == 0 ? static_cast<float>(kPi * p.scale) : 0.f;
                    tl_uf = std::min(tl_uf, 0.f); tl_vf = std::min(tl_vf, pv);
                    br_uf = std::max(br_uf, 0.f); br_vf = std::max(br_vf, pv);
                }
            }
            b.tl_u = tl_uf; b.tl_v = tl_vf; b.br_u = br_uf; b.br_v = br_vf;
        }
    } else if (p.kind >= kFisheye) {
        // RotationWarperBase::detectResultRoi (detail/warpers_inl.hpp:143-166): every source pixel is projected.  The bounds
        // are minima / maxima -- the same whatever the order -- so the rows are spread over the host's cores (round 4: this
        // loop was 70 of the 73 ms per camera that mst_update took for these projectors)
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const int workers = src_h < 64 ? 1 : static_cast<int>(std::min<unsigned>(hw, 64u));
        std::vector<Bounds> part(workers);
        auto rows = [&](int k) {
            const int per = (src_h + workers - 1) / workers, y0 = k * per, y1 = std::min(src_h, y0 + per);
            Bounds lb;
            float u, v;
            for (int y = y0; y < y1; ++y)
                for (int x = 0; x < src_w; ++x) {
                    p.forward(static_cast<float>(x), static_cast<float>(y), u, v);
                    lb.add(u, v);
                }
            part[k] = lb;
        };
        if (workers == 1) rows(0);
        else {
            std::vector<std::thread> pool;
            for (int k = 0; k < workers; ++k) pool.emplace_back(rows, k);
            for (std::thread& t : pool) t.join();
        }
        for (const Bounds& lb : part) {
            b.tl_u = std::min(b.tl_u, lb.tl_u); b.tl_v = std::min(b.tl_v, lb.tl_v);
            b.br_u = std::max(b.br_u, lb.br_u); b.br_v = std::max(b.br_v, lb.br_v);
        }
    } else if (p.kind == kPlane) {
        float u, v;
        p.forward(0, 0, u, v);                                                        b.add(u, v);
        p.forward(0, static_cast<float>(src_h - 1), u, v);                            b.add(u, v);
        p.forward(static_cast<float>(src_w - 1), 0, u, v);                            b.add(u, v);
        p.forward(static_cast<float>(src_w - 1), static_cast<float>(src_h - 1), u, v); b.add(u, v);
    } else {
        b = border_walk(p, src_w, src_h);
        if (p.kind == kSpherical) {
            // the integer bounds go back to float before the pole tests (warpers.cpp:175-178)
            float tl_uf = static_cast<float>(static_cast<int>(b.tl_u));
            float tl_vf = static_cast<float>(static_cast<int>(b.tl_v));
            float br_uf = static_cast<float>(static_cast<int>(b.br_u));
            float br_vf = static_cast<float>(static_cast<int>(b.br_v));
            for (int pole = 0; pole < 2; ++pole) {
                float x = p.rinv[1];
                float y = pole == 0 ? p.rinv[4] : -p.rinv[4];
                float z = p.rinv[7];
                if (y > 0.f) {
                    float x_ = (p.k[0] * x + p.k[1] * y) / z + p.k[2];
                    float y_ = p.k[4] * y / z + p.k[5];
                    if (x_ > 0.f && x_ < src_w && y_ > 0.f && y_ < src_h) {
                        float pv = pole == 0 ? static_cast<float>(kPi * p.scale) : 0.f;
                        tl_uf = std::min(tl_uf, 0.f); tl_vf = std::min(tl_vf, pv);
                        br_uf = std::max(br_uf, 0.f); br_vf = std::max(br_vf, pv);
                    }
                }
            }
            b.tl_u = tl_uf; b.tl_v = tl_vf; b.br_u = br_uf; b.br_v = br_vf;
        }
    }
    tl_x = static_cast<int>(b.tl_u); tl_y = static_cast<int>(b.tl_v);
    br_x = static_cast<int>(b.br_u); br_y = static_cast<int>(b.br_v);
}

Rect warp_roi(const Projector& p, int src_w, int src_h)
{
    int tlx, tly, brx, bry;
    detect_result_roi(p, src_w, src_h, tlx, tly, brx, bry);
    Rect r;
    r.x = tlx; r.y = tly; r.width = brx + 1 - tlx; r.height = bry + 1 - tly;
    return r;
}

Rect result_roi(const std::vector<Rect>& rois)
{
    int tlx = INT_MAX, tly = INT_MAX, brx = INT_MIN, bry = INT_MIN;
    for (const Rect& r : rois) {
        tlx = std::min(tlx, r.x); tly = std::min(tly, r.y);
        brx = std::max(brx, r.br_x()); bry = std::max(bry, r.br_y());
    }
    Rect out;
    out.x = tlx; out.y = tly; out.width = brx - tlx; out.height = bry - tly;
    return out;
}

MapTables build_map_tables(const Projector& p, const Rect& roi)
{
    MapTables t;
    t.col_a.assign(roi.width, 0.f); t.col_b.assign(roi.width, 0.f);
    t.row_a.assign(roi.height, 0.f); t.row_b.assign(roi.height, 0.f);
    if (!separable_map(p.kind)) return t;          // those kernels read the full map instead (build_maps_host)
    for (int i = 0; i < roi.width; ++i) {
        float u = static_cast<float>(roi.x + i);
        if (p.kind == kPlane) {
            t.col_a[i] = u / p.scale - p.t[0];
            t.col_b[i] = 0.f;
        } else {
            u /= p.scale;
            t.col_a[i] = sinf(u);
            t.col_b[i] = cosf(u);
        }
    }
    for (int j = 0; j < roi.height; ++j) {
        float v = static_cast<float>(roi.y + j);
        if (p.kind == kPlane) {
            t.row_a[j] = v / p.scale - p.t[1];
            t.row_b[j] = 0.f;
        } else if (p.kind == kSpherical) {
            v /= p.scale;
            t.row_a[j] = sinf(static_cast<float>(kPi) - v);
            t.row_b[j] = cosf(static_cast<float>(kPi) - v);
        } else {
            t.row_a[j] = v / p.scale;
            t.row_b[j] = 0.f;
        }
    }
    return t;
}

void backward_map_host(const Projector& p, const Rect& roi, const MapTables& t, float* xmap, float* ymap)
{
    if (!separable_map(p.kind)) { build_maps_host(p, roi, xmap, ymap); return; }
    const float* kr = p.k_rinv;
    for (int j = 0; j < roi.height; ++j)
        for (int i = 0; i < roi.width; ++i) {
            float x, y, z;
            if (p.kind == kPlane) {
                float u = t.col_a[i], v = t.row_a[j];
                x = kr[0] * u + kr[1] * v + kr[2] * (1 - p.t[2]);
                y = kr[3] * u + kr[4] * v + kr[5] * (1 - p.t[2]);
                z = kr[6] * u + kr[7] * v + kr[8] * (1 - p.t[2]);
                x /= z;
                y /= z;
            } else {
                float x_, y_, z_;
                if (p.kind == kSpherical) {
                    x_ = t.row_a[j] * t.col_a[i];
                    y_ = t.row_b[j];
                    z_ = t.row_a[j] * t.col_b[i];
                } else {
                    x_ = t.col_a[i];
                    y_ = t.row_a[j];
                    z_ = t.col_b[i];
                }
                x = kr[0] * x_ + kr[1] * y_ + kr[2] * z_;
                y = kr[3] * x_ + kr[4] * y_ + kr[5] * z_;
                z = kr[6] * x_ + kr[7] * y_ + kr[8] * z_;
                if (z > 0) { x /= z; y /= z; }
                else x = y = -1;
            }
            xmap[static_cast<size_t>(j) * roi.width + i] = x;
            ymap[static_cast<size_t>(j) * roi.width + i] = y;
        }
}

Resize8uAxis build_resize8u_axis(int src_len, int dst_len, bool is_x)
{
    Resize8uAxis a;
    a.ofs0.resize(dst_len); a.ofs1.resize(dst_len); a.c0.resize(dst_len); a.c1.resize(dst_len);
    const double inv_scale = static_cast<double>(dst_len) / src_len;
    const double scale = 1. / inv_scale;
    auto coef = [](float w) {      // saturate_cast<short>(w * INTER_RESIZE_COEF_SCALE)
        const double r = std::nearbyint(static_cast<double>(w * 2048.f));
        return static_cast<int16_t>(std::min(32767., std::max(-32768., r)));
    };
    for (int d = 0; d < dst_len; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = floor_to_int(f);
        f -= s;
        if (is_x) {                    // the x axis zeroes the fraction where it clamps; the y axis only clamps rows
            if (s < 0) { f = 0; s = 0; }
            if (s >= src_len - 1) { f = 0; s = src_len - 1; }
        }
        a.ofs0[d] = std::min(std::max(s, 0), src_len - 1);
        a.ofs1[d] = std::min(std::max(s + 1, 0), src_len - 1);
        a.c0[d] = coef(1.f - f);
        a.c1[d] = coef(f);
    }
    return a;
}

ResizeAxis build_gain_axis_x(int src_len, int dst_len)
{
    ResizeAxis a;
    a.ofs0.resize(dst_len); a.ofs1.resize(dst_len); a.w0.resize(dst_len); a.w1.resize(dst_len);
    const double inv_scale = static_cast<double>(dst_len) / src_len;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dst_len; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = floor_to_int(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= src_len - 1) { f = 0; s = src_len - 1; }
        a.ofs0[d] = s;
        a.ofs1[d] = std::min(s + 1, src_len - 1);
        a.w0[d] = 1.f - f;
        a.w1[d] = f;
    }
    return a;
}

ResizeAxis build_gain_axis_y(int src_len, int dst_len)
{
    ResizeAxis a;
    a.ofs0.resize(dst_len); a.ofs1.resize(dst_len); a.w0.resize(dst_len); a.w1.resize(dst_len);
    const double inv_scale = static_cast<double>(dst_len) / src_len;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dst_len; ++d) {
        float f = static_cast<float>((d + 0.5) * scale - 0.5);
        int s = floor_to_int(f);
        f -= s;
        // rows are clipped into the image, the weights are kept
        a.ofs0[d] = std::clamp(s, 0, src_len - 1);
        a.ofs1[d] = std::clamp(s + 1, 0, src_len - 1);
        a.w0[d] = 1.f - f;
        a.w1[d] = f;
    }
    return a;
}

int multiband_num_bands(int actual_num_bands, const Rect& dst_roi)
{
    double max_len = static_cast<double>(std::max(dst_roi.width, dst_roi.height));
    return std::min(actual_num_bands, static_cast<int>(std::ceil(std::log(max_len) / std::log(2.0))));
}

Rect multiband_padded_roi(const Rect& dst_roi, int num_bands)
{
    Rect r = dst_roi;
    const int m = 1 << num_bands;
    r.width += (m - r.width % m) % m;
    r.height += (m - r.height % m) % m;
    return r;
}

BandGeometry multiband_feed_geometry(const Rect& roi, int num_bands, const Rect& img)
{
    const int gap = 3 * (1 << num_bands);
    const int m = 1 << num_bands;
    int tlx = std::max(roi.x, img.x - gap), tly = std::max(roi.y, img.y - gap);
    int brx = std::min(roi.br_x(), img.x + img.width + gap);
    int bry = std::min(roi.br_y(), img.y + img.height + gap);
    tlx = roi.x + (((tlx - roi.x) >> num_bands) << num_bands);
    tly = roi.y + (((tly - roi.y) >> num_bands) << num_bands);
    int width = brx - tlx, height = bry - tly;
    width += (m - width % m) % m;
    height += (m - height % m) % m;
    brx = tlx + width;
    bry = tly + height;
    const int dy = std::max(bry - roi.br_y(), 0);
    const int dx = std::max(brx - roi.br_x(), 0);
    tlx -= dx; brx -= dx;
    tly -= dy; bry -= dy;
    BandGeometry g;
    g.tl_br.x = tlx; g.tl_br.y = tly; g.tl_br.width = brx - tlx; g.tl_br.height = bry - tly;
    g.top = img.y - tly;
    g.left = img.x - tlx;
    g.bottom = bry - img.y - img.height;
    g.right = brx - img.x - img.width;
    return g;
}

// ---- lens undistortion maps -------------------------------------------------------------------------
// OpenCV 2.4.11 imgproc/undistort.cpp (not vendored by the reference; restated from the published algorithm):
// the inverse of newK * R in closed form (cv::invert on a 3x3 CV_64F: reciprocal determinant times adjugate),
// the normalised ray advanced incrementally along each row, the radial/tangential model in double, and the
// CV_16SC2 fixed-point split cvRound(u * 32) -> (>> 5, & 31).
namespace {
double round_half_even_to_int(double v, bool& ok)
{
    const double r = std::nearbyint(v);             // default rounding mode: to nearest, ties to even (cvtsd2si)
    ok = r >= -2147483648.0 && r <= 2147483647.0;
    return r;
}
int cv_round64(double v)
{
    bool ok;
    const double r = round_half_even_to_int(v, ok);
    return ok ? static_cast<int>(r) : INT_MIN;      // the x64 conversion's "integer indefinite"
}
}  // namespace

void compose_lens_maps(float* xmap, float* ymap, size_t n, const int16_t* map1, const uint16_t* map2, int w, int h)
{
    auto lens = [&](int x, int y, float& mx, float& my) {
        const size_t i = static_cast<size_t>(y) * w + x;
        mx = static_cast<float>(map1[2 * i]) + static_cast<float>(map2[i] & 31) * (1.0f / 32.0f);
        my = static_cast<float>(map1[2 * i + 1]) + static_cast<float>((map2[i] >> 5) & 31) * (1.0f / 32.0f);
    };
    const float xmax = static_cast<float>(w - 1), ymax = static_cast<float>(h - 1);
    for (size_t k = 0; k < n; ++k) {
        float u = xmap[k], v = ymap[k];
        if (!(u > 0.0f)) u = 0.0f;             // (NaN lands on 0 too)
        if (!(v > 0.0f)) v = 0.0f;
        if (u > xmax) u = xmax;
        if (v > ymax) v = ymax;
        const int x0 = std::max(0, std::min(static_cast<int>(u), w - 2)), y0 = std::max(0, std::min(static_cast<int>(v), h - 2));
        const int x1 = std::min(x0 + 1, w - 1), y1 = std::min(y0 + 1, h - 1);
        const float ax = u - static_cast<float>(x0), ay = v - static_cast<float>(y0);
        const float ax0 = 1.0f - ax, ay0 = 1.0f - ay;
        float ax_[4], ay_[4];
        lens(x0, y0, ax_[0], ay_[0]); lens(x1, y0, ax_[1], ay_[1]); lens(x0, y1, ax_[2], ay_[2]); lens(x1, y1, ax_[3], ay_[3]);
        const float r0x = ax_[0] * ax0 + ax_[1] * ax, r1x = ax_[2] * ax0 + ax_[3] * ax;
        const float r0y = ay_[0] * ax0 + ay_[1] * ax, r1y = ay_[2] * ax0 + ay_[3] * ax;
        xmap[k] = r0x * ay0 + r1x * ay;
        ymap[k] = r0y * ay0 + r1y * ay;
    }
}

bool init_undistort_rectify_map(const double K[9], const double* dist, int ndist, const double* R, const double newK[9],
                                int width, int height, int16_t* map1, uint16_t* map2)
{
    if (!(ndist == 0 || ndist == 4 || ndist == 5 || ndist == 8)) return false;
    double m[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            if (!R) { m[3 * i + j] = newK[3 * i + j]; continue; }
            double acc = 0;
            for (int k = 0; k < 3; ++k) acc += newK[3 * i + k] * R[3 * k + j];
            m[3 * i + j] = acc;
        }
    double ir[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double det = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
                 m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (det != 0.) {
        det = 1. / det;
        ir[0] = (m[4] * m[8] - m[5] * m[7]) * det; ir[1] = (m[2] * m[7] - m[1] * m[8]) * det;
        ir[2] = (m[1] * m[5] - m[2] * m[4]) * det; ir[3] = (m[5] * m[6] - m[3] * m[8]) * det;
        ir[4] = (m[0] * m[8] - m[2] * m[6]) * det; ir[5] = (m[2] * m[3] - m[0] * m[5]) * det;
        ir[6] = (m[3] * m[7] - m[4] * m[6]) * det; ir[7] = (m[1] * m[6] - m[0] * m[7]) * det;
        ir[8] = (m[0] * m[4] - m[1] * m[3]) * det;
    }
    const double u0 = K[2], v0 = K[5], fx = K[0], fy = K[4];
    const double k1 = ndist > 0 ? dist[0] : 0., k2 = ndist > 1 ? dist[1] : 0.;
    const double p1 = ndist > 2 ? dist[2] : 0., p2 = ndist > 3 ? dist[3] : 0.;
    const double k3 = ndist >= 5 ? dist[4] : 0.;
    const double k4 = ndist >= 8 ? dist[5] : 0., k5 = ndist >= 8 ? dist[6] : 0., k6 = ndist >= 8 ? dist[7] : 0.;
    for (int i = 0; i < height; ++i) {
        int16_t* m1 = map1 + static_cast<size_t>(i) * width * 2;
        uint16_t* m2 = map2 + static_cast<size_t>(i) * width;
        double xr = i * ir[1] + ir[2], yr = i * ir[4] + ir[5], wr = i * ir[7] + ir[8];
        for (int j = 0; j < width; ++j, xr += ir[0], yr += ir[3], wr += ir[6]) {
            const double w = 1. / wr, x = xr * w, y = yr * w;
            const double x2 = x * x, y2 = y * y;
            const double r2 = x2 + y2, xy2 = 2 * x * y;
            const double kr = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2);
            const double u = fx * (x * kr + p1 * xy2 + p2 * (r2 + 2 * x2)) + u0;
            const double v = fy * (y * kr + p1 * (r2 + 2 * y2) + p2 * xy2) + v0;
            const int iu = cv_round64(u * 32), iv = cv_round64(v * 32);
            m1[2 * j] = static_cast<int16_t>(iu >> 5);
            m1[2 * j + 1] = static_cast<int16_t>(iv >> 5);
            m2[j] = static_cast<uint16_t>((iv & 31) * 32 + (iu & 31));
        }
    }
    return true;
}

}  // namespace mst
