// calib_blob.cpp -- a calibration (mst_calibration: K, R, T, scale, gain maps, masks, lens maps, seam masks) as one
// self-contained binary, so that a calibration GetPreStruct() (DRV:348-694) once produced can be replayed: benchmarks and
// parity runs on the same inputs, a stitcher that restarts without re-registering.  The reference has nothing of the kind
// ("checkpoint / resume: none"; REL64/camera-mp4.yml is read by no code).  Host only.
//
// Layout (little endian, every array 8-byte aligned):
//   BlobHeader | CameraRecord[num_cameras] | arrays ...      with array offsets relative to the start of the blob.
// mst_calibration_load points the cameras' array pointers INTO the blob: it must outlive the calibration's use.
#include <cstdint>
#include <cstring>

#include "../../include/mi355_stitch.h"

namespace {

constexpr uint32_t kMagic = 0x4354534Du;          // "MSTC"
constexpr uint32_t kVersion = 1;

struct BlobHeader {
    uint32_t magic, version;
    uint32_t num_cameras;
    float warped_image_scale;
    uint64_t total_bytes;
    uint64_t checksum;                             // FNV-1a 64 of everything after the header
};

struct CameraRecord {
    int32_t src_width, src_height;
    float K[9], R[9], T[3];
    float gain;
    int32_t gain_w, gain_h;
    int32_t mask_width, mask_height;
    int32_t seam_mask_width, seam_mask_height;
    int32_t has_lens_maps, pad;
    uint64_t off_gain_map, off_mask, off_map1, off_map2, off_seam_mask;      // 0: absent
};

size_t align8(size_t v) { return (v + 7) & ~size_t{7}; }

uint64_t fnv1a(const uint8_t* p, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= p[i]; h *= 0x100000001b3ull; }
    return h;
}

struct Sizes { size_t gain, mask, map1, map2, seam; };
Sizes sizes_of(const mst_camera& c)
{
    Sizes s{};
    if (c.gain_map && c.gain_w > 0 && c.gain_h > 0) s.gain = sizeof(float) * static_cast<size_t>(c.gain_w) * c.gain_h;
    if (c.mask && c.mask_width > 0 && c.mask_height > 0) s.mask = static_cast<size_t>(c.mask_width) * c.mask_height;
    if (c.undistort_map1 && c.undistort_map2) {
        s.map1 = sizeof(int16_t) * 2 * static_cast<size_t>(c.src_width) * c.src_height;
        s.map2 = sizeof(uint16_t) * static_cast<size_t>(c.src_width) * c.src_height;
    }
    if (c.seam_mask && c.seam_mask_width > 0 && c.seam_mask_height > 0) s.seam = static_cast<size_t>(c.seam_mask_width) * c.seam_mask_height;
    return s;
}

bool valid(const mst_calibration* c)
{
    return c && c->struct_size == static_cast<int32_t>(sizeof(mst_calibration)) && c->num_cameras >= 1 &&
           c->num_cameras <= MST_MAX_CAMERAS && c->cameras;
}

}  // namespace

extern "C" {

size_t mst_calibration_blob_size(const mst_calibration* calib)
{
    if (!valid(calib)) return 0;
    size_t total = align8(sizeof(BlobHeader)) + align8(sizeof(CameraRecord) * static_cast<size_t>(calib->num_cameras));
    for (int i = 0; i < calib->num_cameras; ++i) {
        const mst_camera& c = calib->cameras[i];
        if (c.src_width <= 0 || c.src_height <= 0) return 0;
        if (c.mask && c.mask_stride < static_cast<size_t>(c.mask_width)) return 0;
        if (c.seam_mask && c.seam_mask_stride < static_cast<size_t>(c.seam_mask_width)) return 0;
        if ((c.undistort_map1 == nullptr) != (c.undistort_map2 == nullptr)) return 0;      // a lone lens map: mst_update rejects it too
        const Sizes s = sizes_of(c);
        total += align8(s.gain) + align8(s.mask) + align8(s.map1) + align8(s.map2) + align8(s.seam);
    }
    return total;
}

int mst_calibration_save(const mst_calibration* calib, void* blob, size_t capacity, size_t* written)
{
    const size_t total = mst_calibration_blob_size(calib);
    if (!total || !blob) return MST_ERR_INVALID_ARG;
    if (capacity < total) return MST_ERR_INVALID_ARG;
    uint8_t* base = static_cast<uint8_t*>(blob);
    std::memset(base, 0, total);
    const int n = calib->num_cameras;
    BlobHeader h{kMagic, kVersion, static_cast<uint32_t>(n), calib->warped_image_scale, total, 0};
    size_t off = align8(sizeof(BlobHeader)) + align8(sizeof(CameraRecord) * static_cast<size_t>(n));
    CameraRecord* recs = reinterpret_cast<CameraRecord*>(base + align8(sizeof(BlobHeader)));
    auto put_rows = [&](const uint8_t* src, size_t stride, size_t row, size_t rows) -> uint64_t {
        const uint64_t at = off;
        for (size_t y = 0; y < rows; ++y) std::memcpy(base + off + y * row, src + y * stride, row);      // tight in the blob
        off += align8(row * rows);
        return at;
    };
    for (int i = 0; i < n; ++i) {
        const mst_camera& c = calib->cameras[i];
        CameraRecord& r = recs[i];
        r.src_width = c.src_width; r.src_height = c.src_height;
        std::memcpy(r.K, c.K, sizeof(r.K)); std::memcpy(r.R, c.R, sizeof(r.R)); std::memcpy(r.T, c.T, sizeof(r.T));
        r.gain = c.gain;
        const Sizes s = sizes_of(c);
        if (s.gain) { r.gain_w = c.gain_w; r.gain_h = c.gain_h; r.off_gain_map = put_rows(reinterpret_cast<const uint8_t*>(c.gain_map), s.gain, s.gain, 1); }
        if (s.mask) { r.mask_width = c.mask_width; r.mask_height = c.mask_height; r.off_mask = put_rows(c.mask, c.mask_stride, c.mask_width, c.mask_height); }
        if (s.map1) {
            r.has_lens_maps = 1;
            r.off_map1 = put_rows(reinterpret_cast<const uint8_t*>(c.undistort_map1), s.map1, s.map1, 1);
            r.off_map2 = put_rows(reinterpret_cast<const uint8_t*>(c.undistort_map2), s.map2, s.map2, 1);
        }
        if (s.seam) {
            r.seam_mask_width = c.seam_mask_width; r.seam_mask_height = c.seam_mask_height;
            r.off_seam_mask = put_rows(c.seam_mask, c.seam_mask_stride, c.seam_mask_width, c.seam_mask_height);
        }
    }
    h.checksum = fnv1a(base + sizeof(BlobHeader), total - sizeof(BlobHeader));
    std::memcpy(base, &h, sizeof(h));
    if (written) *written = total;
    return MST_OK;
}

int mst_calibration_load(const void* blob, size_t bytes, mst_calibration* out, mst_camera* cameras, int max_cameras)
{
    if (!blob || !out || !cameras || bytes < sizeof(BlobHeader)) return MST_ERR_INVALID_ARG;
    // the records and arrays are handed out as typed pointers INTO the blob (offsets are multiples of 8): the blob itself
    // must start on an 8-byte boundary (a slice of a larger buffer at an odd offset is refused, not cast)
    if (reinterpret_cast<uintptr_t>(blob) % 8 != 0) return MST_ERR_INVALID_ARG;
    const uint8_t* base = static_cast<const uint8_t*>(blob);
    BlobHeader h;
    std::memcpy(&h, base, sizeof(h));
    if (h.magic != kMagic || h.version != kVersion || h.total_bytes > bytes || h.num_cameras < 1 ||
        h.num_cameras > static_cast<uint32_t>(MST_MAX_CAMERAS) || static_cast<int>(h.num_cameras) > max_cameras)
        return MST_ERR_INVALID_ARG;
    const size_t recs_at = align8(sizeof(BlobHeader)), arrays_at = recs_at + align8(sizeof(CameraRecord) * static_cast<size_t>(h.num_cameras));
    if (arrays_at > h.total_bytes) return MST_ERR_INVALID_ARG;
    if (fnv1a(base + sizeof(BlobHeader), h.total_bytes - sizeof(BlobHeader)) != h.checksum) return MST_ERR_INVALID_ARG;
    const CameraRecord* recs = reinterpret_cast<const CameraRecord*>(base + recs_at);
    // (no sum of file-supplied values: a crafted offset must not wrap)
    auto inside = [&](uint64_t off, size_t len) { return off >= arrays_at && off % 8 == 0 && off <= h.total_bytes && len <= h.total_bytes - off; };
    for (uint32_t i = 0; i < h.num_cameras; ++i) {
        const CameraRecord& r = recs[i];
        mst_camera& c = cameras[i];
        std::memset(&c, 0, sizeof(c));
        if (r.src_width <= 0 || r.src_height <= 0 || r.src_width > 65536 || r.src_height > 65536) return MST_ERR_INVALID_ARG;
        c.src_width = r.src_width; c.src_height = r.src_height;
        std::memcpy(c.K, r.K, sizeof(c.K)); std::memcpy(c.R, r.R, sizeof(c.R)); std::memcpy(c.T, r.T, sizeof(c.T));
        c.gain = r.gain;
        const size_t px = static_cast<size_t>(r.src_width) * r.src_height;
        if (r.off_gain_map) {
            if (r.gain_w <= 0 || r.gain_h <= 0 || !inside(r.off_gain_map, sizeof(float) * static_cast<size_t>(r.gain_w) * r.gain_h)) return MST_ERR_INVALID_ARG;
            c.gain_map = reinterpret_cast<const float*>(base + r.off_gain_map); c.gain_w = r.gain_w; c.gain_h = r.gain_h;
        }
        if (r.off_mask) {
            if (r.mask_width <= 0 || r.mask_height <= 0 || !inside(r.off_mask, static_cast<size_t>(r.mask_width) * r.mask_height)) return MST_ERR_INVALID_ARG;
            c.mask = base + r.off_mask; c.mask_width = r.mask_width; c.mask_height = r.mask_height; c.mask_stride = static_cast<size_t>(r.mask_width);
        }
        if (r.has_lens_maps) {
            if (!inside(r.off_map1, 4 * px) || !inside(r.off_map2, 2 * px)) return MST_ERR_INVALID_ARG;
            c.undistort_map1 = reinterpret_cast<const int16_t*>(base + r.off_map1);
            c.undistort_map2 = reinterpret_cast<const uint16_t*>(base + r.off_map2);
        }
        if (r.off_seam_mask) {
            if (r.seam_mask_width <= 0 || r.seam_mask_height <= 0 || !inside(r.off_seam_mask, static_cast<size_t>(r.seam_mask_width) * r.seam_mask_height)) return MST_ERR_INVALID_ARG;
            c.seam_mask = base + r.off_seam_mask; c.seam_mask_width = r.seam_mask_width; c.seam_mask_height = r.seam_mask_height;
            c.seam_mask_stride = static_cast<size_t>(r.seam_mask_width);
        }
    }
    out->struct_size = static_cast<int32_t>(sizeof(mst_calibration));
    out->num_cameras = static_cast<int32_t>(h.num_cameras);
    out->warped_image_scale = h.warped_image_scale;
    out->cameras = cameras;
    return MST_OK;
}

}  // extern "C"
