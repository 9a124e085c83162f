// stitching_detailed_mi355 -- the reference's command-line / frame-loop surface on top of the C ABI.
//
// Mirrors the compositing half of the OpenCV `stitching_detailed` sample the reference ships
// (/root/reference/stitching/OpenCV2.4.11-Stitching/stitching_detailed.cpp: flags :64-119, defaults :122-143,
// blender set-up rule :729-749, compositing loop :656-756) and the driver's frame loop
// (R64/stitching_detailed.cpp:724-770: per frame N Mats in, one Mat out, "Stitching, time: ... sec").
// Registration (features / matching / bundle adjustment / seams) is out of scope: cameras come from
// --yaw / --focal (a ring rig, R = R_y(yaw)); seams: `--seam no` (all-255 warped masks), voronoi or the two DP finders.
//
//   stitching_detailed_mi355 --synthetic 6 1920x1080 --focal 1250 --warp spherical --blend multiband
//                            --expos_comp gain_blocks --frames 100 --output result.ppm
//   stitching_detailed_mi355 img0.ppm img1.ppm --yaw -15,15 --focal 1250 --warp plane --blend feather
//
// Build: make -C stitchingvideo_amd/csrc cli     (g++; links libmi355stitch.so only)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/mi355_stitch.h"

namespace {

struct Image { int w = 0, h = 0; std::vector<uint8_t> bgr; };

bool read_ppm(const std::string& path, Image& im)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[3] = {0};
    int maxv = 0;
    if (std::fscanf(f, "%2s %d %d %d", magic, &im.w, &im.h, &maxv) != 4 || std::strcmp(magic, "P6") || maxv != 255) {
        std::fclose(f);
        return false;
    }
    std::fgetc(f);
    std::vector<uint8_t> rgb(static_cast<size_t>(im.w) * im.h * 3);
    const bool ok = std::fread(rgb.data(), 1, rgb.size(), f) == rgb.size();
    std::fclose(f);
    im.bgr.resize(rgb.size());
    for (size_t i = 0; i + 2 < rgb.size(); i += 3) { im.bgr[i] = rgb[i + 2]; im.bgr[i + 1] = rgb[i + 1]; im.bgr[i + 2] = rgb[i]; }
    return ok;
}

bool write_ppm(const std::string& path, const uint8_t* bgr, int w, int h)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", w, h);
    std::vector<uint8_t> row(static_cast<size_t>(w) * 3);
    for (int y = 0; y < h; ++y) {
        const uint8_t* s = bgr + static_cast<size_t>(y) * w * 3;
        for (int x = 0; x < w; ++x) { row[3 * x] = s[3 * x + 2]; row[3 * x + 1] = s[3 * x + 1]; row[3 * x + 2] = s[3 * x]; }
        std::fwrite(row.data(), 1, row.size(), f);
    }
    std::fclose(f);
    return true;
}

void usage()
{
    std::printf(
        "Rotation model images stitcher (compositing on MI355X).\n\n"
        "stitching_detailed_mi355 img1.ppm img2.ppm [...imgN.ppm] [flags]\n"
        "stitching_detailed_mi355 --synthetic <N> <WxH> [flags]\n\n"
        "Flags:\n"
        "  --yaw a0,a1,...          camera yaw angles in degrees (default: evenly spaced ring)\n"
        "  --focal <float>          focal length in pixels = warp scale (default 1250)\n"
        "  --warp (plane|cylindrical|spherical|fisheye|stereographic|compressedPlaneA2B1|compressedPlaneA1.5B1|\n"
        "          compressedPlanePortraitA2B1|compressedPlanePortraitA1.5B1|paniniA2B1|paniniA1.5B1|paniniPortraitA2B1|\n"
        "          paniniPortraitA1.5B1|mercator|transverseMercator)                 default spherical (CLI:102)\n"
        "  --expos_comp (no|gain|gain_blocks)                default gain_blocks (CLI:111; the gains are synthetic: registration is\n"
        "                             out of scope -- `gain`: one factor per camera, `gain_blocks`: 32 x 32 block maps)\n"
        "  --compose_megapix <float>  resolution of the compositing step, -1 = original (default, CLI:108-110): the inputs are resized\n"
        "                             by min(1, sqrt(Mpx * 1e6 / area)) and focal / ppx / ppy / the warp scale follow (CLI:662-700)\n"
        "  --seam_megapix <float>     accepted for compatibility (CLI:104): this tool finds seams at compose resolution\n"
        "  --parse_only               print the parsed configuration and exit (no GPU touched)\n"
        "  --blend (no|feather|multiband)                    default multiband\n"
        "  --blend_strength <float>   [0,100], default 5 (same rule as the sample: width = sqrt(area)*s/100)\n"
        "  --seam (no|voronoi|dp_color|dp_colorgrad)          seam finder on the full-resolution warped images (CLI:106 has\n"
        "                             the graph-cut finders too: those stay on the host side of the real application); default no\n"
        "  --save_calibration <file>  write the calibration (K, R, scale, gains, final masks) as one binary (mst_calibration_save)\n"
        "  --load_calibration <file>  compose from such a file instead of --yaw / --focal / --seam\n"
        "  --lut                      the driver's hard-seam LUT composite with its crop margins\n"
        "  --frames <K>               frame-loop iterations (default 1)\n"
        "  --output <result.ppm>      default result.ppm\n");
}

// cv::resize(src, dst, Size(), f, f, INTER_LINEAR) on CV_8UC3 as the sample applies it to every input when --compose_megapix shrinks
// them (CLI:699-700; imgwarp.cpp of 2.4.11: dsize = cvRound(size * f), scale = 1 / f, 11-bit horizontal coefficients, the vertical pass
// `(((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`).  Host code like the sample's own imread + resize: it runs once per
// input image, before the frame loop.
int cv_round(double v) { return static_cast<int>(std::nearbyint(v)); }

short resize_coef(float v) { return static_cast<short>(cv_round(static_cast<double>(v * 2048.f))); }

Image resize_linear_bgr(const Image& src, double f)
{
    Image dst;
    dst.w = cv_round(src.w * f); dst.h = cv_round(src.h * f);
    dst.bgr.resize(static_cast<size_t>(dst.w) * dst.h * 3);
    const double scale = 1. / f;
    std::vector<int> xofs(dst.w);
    std::vector<short> alpha(2 * static_cast<size_t>(dst.w));
    int xmax = dst.w;
    for (int dx = 0; dx < dst.w; ++dx) {
        float fx = static_cast<float>((dx + 0.5) * scale - 0.5);
        int sx = static_cast<int>(std::floor(fx));
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx + 1 >= src.w) xmax = std::min(xmax, dx);
        if (sx >= src.w - 1) { fx = 0; sx = src.w - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = resize_coef(1.f - fx); alpha[2 * dx + 1] = resize_coef(fx);
    }
    std::vector<int> rows[2] = {std::vector<int>(static_cast<size_t>(dst.w) * 3), std::vector<int>(static_cast<size_t>(dst.w) * 3)};
    for (int dy = 0; dy < dst.h; ++dy) {
        float fy = static_cast<float>((dy + 0.5) * scale - 0.5);
        const int sy = static_cast<int>(std::floor(fy));
        fy -= sy;
        const short b0 = resize_coef(1.f - fy), b1 = resize_coef(fy);
        for (int k = 0; k < 2; ++k) {
            const int yy = std::min(std::max(sy + k, 0), src.h - 1);                  // rows are clipped, the weights stay
            const uint8_t* S = src.bgr.data() + static_cast<size_t>(yy) * src.w * 3;
            int* D = rows[k].data();
            for (int dx = 0; dx < dst.w; ++dx)
                for (int c = 0; c < 3; ++c)
                    D[3 * dx + c] = dx < xmax ? S[3 * xofs[dx] + c] * alpha[2 * dx] + S[3 * (xofs[dx] + 1) + c] * alpha[2 * dx + 1]
                                              : S[3 * xofs[dx] + c] * 2048;
        }
        uint8_t* out = dst.bgr.data() + static_cast<size_t>(dy) * dst.w * 3;
        for (int i = 0; i < dst.w * 3; ++i) {
            const int v = (((b0 * (rows[0][i] >> 4)) >> 16) + ((b1 * (rows[1][i] >> 4)) >> 16) + 2) >> 2;
            out[i] = static_cast<uint8_t>(std::min(std::max(v, 0), 255));
        }
    }
    return dst;
}

void rot_y(double deg, float R[9])
{
    const double a = deg * 3.14159265358979323846 / 180.0, c = std::cos(a), s = std::sin(a);
    const double m[9] = {c, 0, s, 0, 1, 0, -s, 0, c};
    for (int i = 0; i < 9; ++i) R[i] = static_cast<float>(m[i]);
}

#define CHECK(call)                                                                                  \
    do {                                                                                             \
        int rc_ = (call);                                                                            \
        if (rc_ != MST_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mst_last_error(engine)); return -1; } \
    } while (0)

}  // namespace

int main(int argc, char** argv)
{
    std::vector<std::string> img_names;
    std::vector<double> yaws;
    int synthetic_n = 0, W = 0, H = 0, frames = 1;
    double focal = 1250.0, blend_strength = 5.0, compose_megapix = -1, seam_megapix = 0.1;
    bool parse_only = false;
    std::string warp_type = "spherical", expos = "gain_blocks", blend = "multiband", result_name = "result.ppm";
    std::string seam = "no", save_calib, load_calib;
    bool lut = false;
    mst_engine* engine = nullptr;

    if (argc == 1) { usage(); return -1; }
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); std::exit(-1); } return argv[++i]; };
        if (a == "--help" || a == "/?") { usage(); return -1; }
        else if (a == "--synthetic") { synthetic_n = std::atoi(next()); if (std::sscanf(next(), "%dx%d", &W, &H) != 2) { usage(); return -1; } }
        else if (a == "--yaw") { const char* p = next(); while (*p) { yaws.push_back(std::strtod(p, const_cast<char**>(&p))); if (*p == ',') ++p; } }
        else if (a == "--focal") focal = std::atof(next());
        else if (a == "--warp") warp_type = next();
        else if (a == "--expos_comp") expos = next();
        else if (a == "--blend") blend = next();
        else if (a == "--blend_strength") blend_strength = std::atof(next());
        else if (a == "--compose_megapix") compose_megapix = std::atof(next());
        else if (a == "--seam_megapix") seam_megapix = std::atof(next());
        else if (a == "--parse_only") parse_only = true;
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--output" || a == "--result") result_name = next();
        else if (a == "--lut") lut = true;
        else if (a == "--seam") seam = next();
        else if (a == "--save_calibration") save_calib = next();
        else if (a == "--load_calibration") load_calib = next();
        else if (a.rfind("--", 0) == 0) { std::printf("Bad flag: %s\n", a.c_str()); return -1; }
        else img_names.push_back(a);
    }

    std::vector<Image> stills;
    int num_images = synthetic_n;
    if (!synthetic_n) {
        num_images = static_cast<int>(img_names.size());
        stills.resize(num_images);
        for (int i = 0; i < num_images; ++i)
            if (!read_ppm(img_names[i], stills[i])) { std::printf("Can't open image %s\n", img_names[i].c_str()); return -1; }
        if (num_images) { W = stills[0].w; H = stills[0].h; }
    }
    if (num_images < 2) { std::printf("Need more images\n"); return -1; }           // CLI:343-347
    if (yaws.empty()) for (int i = 0; i < num_images; ++i) yaws.push_back(360.0 * i / num_images);
    if (static_cast<int>(yaws.size()) != num_images) { std::printf("--yaw needs one angle per image\n"); return -1; }

    const int warp_kind = mst_warp_kind_from_name(warp_type.c_str());
    if (warp_kind < 0) { std::fprintf(stderr, "Bad warping method\n"); return 1; }       // CLI:573-577
    int blend_kind = blend == "no" ? MST_BLEND_NO : blend == "feather" ? MST_BLEND_FEATHER : MST_BLEND_MULTI_BAND;
    if (expos != "no" && expos != "gain" && expos != "gain_blocks") { std::printf("Bad exposure compensation method\n"); return -1; }   // CLI:286-290
    if (blend != "no" && blend != "feather" && blend != "multiband") { std::printf("Bad blending method\n"); return -1; }             // CLI:312-316
    const int expos_kind = expos == "no" ? MST_EXPOS_NO : expos == "gain" ? MST_EXPOS_GAIN : MST_EXPOS_GAIN_BLOCKS;
    // --compose_megapix (CLI:662-700): one scale for the inputs, the intrinsics and the warp scale; the inputs are only resized when the
    // scale is more than 10 % away from 1 (CLI:686, 699)
    double compose_scale = 1.0;
    if (compose_megapix > 0) compose_scale = std::min(1.0, std::sqrt(compose_megapix * 1e6 / (static_cast<double>(W) * H)));
    const bool resize_inputs = std::abs(compose_scale - 1) > 1e-1;
    const int full_w = W, full_h = H;
    const double ppx = full_w / 2.0 * compose_scale, ppy = full_h / 2.0 * compose_scale, focal_c = focal * compose_scale;
    const float scale = static_cast<float>(focal) * static_cast<float>(compose_scale);      // warped_image_scale *= (float)compose_work_aspect
    if (resize_inputs) {
        W = cv_round(full_w * compose_scale); H = cv_round(full_h * compose_scale);
        for (auto& im : stills) im = resize_linear_bgr(im, compose_scale);
    }
    if (parse_only) {
        std::printf("images %d size %dx%d -> %dx%d compose_scale %.9g focal %.9g ppx %.9g ppy %.9g warp %s scale %.9g seam %s seam_megapix %g "
                    "expos_comp %s blend %s blend_strength %g output %s\n", num_images, full_w, full_h, W, H, compose_scale, focal_c, ppx, ppy,
                    warp_type.c_str(), static_cast<double>(scale), seam.c_str(), seam_megapix, expos.c_str(), blend.c_str(), blend_strength,
                    result_name.c_str());
        return 0;
    }

    // cameras: K from focal / principal point (CameraParams::K), R = R_y(yaw); corners and sizes from warpRoi
    std::vector<mst_camera> cams(num_images);
    std::vector<std::vector<float>> gain_maps(num_images);
    std::vector<std::vector<uint8_t>> masks(num_images);
    int tlx = 1 << 30, tly = 1 << 30, brx = -(1 << 30), bry = -(1 << 30);
    for (int i = 0; i < num_images; ++i) {
        mst_camera& c = cams[i];
        std::memset(&c, 0, sizeof(c));
        c.src_width = W; c.src_height = H;
        const float K[9] = {static_cast<float>(focal_c), 0, static_cast<float>(ppx), 0, static_cast<float>(focal_c),
                            static_cast<float>(ppy), 0, 0, 1};
        std::memcpy(c.K, K, sizeof(K));
        rot_y(yaws[i], c.R);
        mst_rect roi;
        CHECK(mst_warp_roi(warp_kind, scale, c.K, c.R, c.T, W, H, &roi));
        // mask = warp(all-255, INTER_NEAREST, BORDER_CONSTANT)  (CLI:714-717), no seam finder
        std::vector<uint8_t> ones(static_cast<size_t>(W) * H, 255);
        masks[i].resize(static_cast<size_t>(roi.width) * roi.height);
        CHECK(mst_warp(0, warp_kind, scale, c.K, c.R, c.T, ones.data(), W, H, W, 1, MST_INTER_NEAREST, MST_BORDER_CONSTANT,
                       masks[i].data(), roi.width));
        c.mask = masks[i].data(); c.mask_width = roi.width; c.mask_height = roi.height; c.mask_stride = roi.width;
        if (expos_kind == MST_EXPOS_GAIN_BLOCKS) {
            const int gw = (roi.width + 31) / 32, gh = (roi.height + 31) / 32;        // bl_width_ = bl_height_ = 32
            gain_maps[i].resize(static_cast<size_t>(gw) * gh);
            for (int by = 0; by < gh; ++by)
                for (int bx = 0; bx < gw; ++bx)
                    gain_maps[i][static_cast<size_t>(by) * gw + bx] =
                        static_cast<float>(1.0 + 0.1 * std::sin(i + bx / 5.0) * std::cos(by / 3.0));
            c.gain_map = gain_maps[i].data(); c.gain_w = gw; c.gain_h = gh;
        }
        c.gain = expos_kind == MST_EXPOS_GAIN ? static_cast<float>(1.0 + 0.05 * std::sin(1.0 + i)) : 1.f;     // (synthetic, as stitchingvideo_amd/rig.py)
        tlx = std::min(tlx, roi.x); tly = std::min(tly, roi.y);
        brx = std::max(brx, roi.x + roi.width); bry = std::max(bry, roi.y + roi.height);
    }

    if (seam != "no" && seam != "voronoi" && seam != "dp_color" && seam != "dp_colorgrad") { std::printf("Bad seam finding method\n"); return 1; }   // CLI:631-635
    if (seam != "no" && load_calib.empty()) {
        // the sample finds seams on warped images at seam scale and resizes the masks (CLI:477-640, 714-722); this tool works at
        // full resolution throughout: warp every first frame, hand images + masks to the finder, use its masks as the final ones
        std::vector<int32_t> xy(2 * num_images), wh(2 * num_images);
        std::vector<std::vector<uint8_t>> warped(num_images);
        std::vector<const void*> iptr(num_images);
        std::vector<size_t> istr(num_images), mstr(num_images);
        std::vector<uint8_t*> mptr(num_images);
        std::vector<uint8_t> first(static_cast<size_t>(W) * H * 3);
        for (int i = 0; i < num_images; ++i) {
            mst_rect roi;
            CHECK(mst_warp_roi(warp_kind, scale, cams[i].K, cams[i].R, cams[i].T, W, H, &roi));
            xy[2 * i] = roi.x; xy[2 * i + 1] = roi.y; wh[2 * i] = roi.width; wh[2 * i + 1] = roi.height;
            if (synthetic_n) {
                Image full; full.w = full_w; full.h = full_h; full.bgr.resize(static_cast<size_t>(full_w) * full_h * 3);
                CHECK(mst_synth_frame(i, 0, full_w, full_h, full.bgr.data(), static_cast<size_t>(full_w) * 3));
                first = resize_inputs ? resize_linear_bgr(full, compose_scale).bgr : full.bgr;
            }
            const uint8_t* src = synthetic_n ? first.data() : stills[i].bgr.data();
            warped[i].resize(static_cast<size_t>(roi.width) * roi.height * 3);
            CHECK(mst_warp(0, warp_kind, scale, cams[i].K, cams[i].R, cams[i].T, src, W, H, static_cast<size_t>(W) * 3, 3, MST_INTER_LINEAR,
                           MST_BORDER_REFLECT, warped[i].data(), static_cast<size_t>(roi.width) * 3));
            iptr[i] = warped[i].data(); istr[i] = static_cast<size_t>(roi.width) * 3;
            mptr[i] = masks[i].data(); mstr[i] = roi.width;
        }
        const auto t0 = std::chrono::steady_clock::now();
        if (seam == "voronoi") CHECK(mst_seam_voronoi(0, num_images, xy.data(), wh.data(), mptr.data(), mstr.data()));
        else CHECK(mst_seam_dp(0, num_images, xy.data(), wh.data(), iptr.data(), istr.data(), MST_IMAGE_8UC3, mptr.data(), mstr.data(),
                               seam == "dp_color" ? MST_SEAM_DP_COLOR : MST_SEAM_DP_COLOR_GRAD));
        std::printf("Finding seams, time: %g sec\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());   // CLI:639
    }
    mst_calibration cal = {static_cast<int32_t>(sizeof(mst_calibration)), num_images, scale, cams.data()};
    std::vector<uint8_t> blob;
    std::vector<mst_camera> loaded(MST_MAX_CAMERAS);
    if (!load_calib.empty()) {
        FILE* f = std::fopen(load_calib.c_str(), "rb");
        if (!f) { std::printf("Can't open %s\n", load_calib.c_str()); return -1; }
        std::fseek(f, 0, SEEK_END); blob.resize(static_cast<size_t>(std::ftell(f))); std::fseek(f, 0, SEEK_SET);
        const bool ok = std::fread(blob.data(), 1, blob.size(), f) == blob.size();
        std::fclose(f);
        if (!ok || mst_calibration_load(blob.data(), blob.size(), &cal, loaded.data(), MST_MAX_CAMERAS) != MST_OK || cal.num_cameras != num_images) {
            std::printf("%s is not a calibration of %d cameras\n", load_calib.c_str(), num_images); return -1;
        }
        std::printf("Calibration of %d cameras from %s\n", cal.num_cameras, load_calib.c_str());
    }
    if (!save_calib.empty()) {
        std::vector<uint8_t> out(mst_calibration_blob_size(&cal));
        size_t n = 0;
        if (out.empty() || mst_calibration_save(&cal, out.data(), out.size(), &n) != MST_OK) { std::printf("Can't serialise the calibration\n"); return -1; }
        FILE* f = std::fopen(save_calib.c_str(), "wb");
        if (!f || std::fwrite(out.data(), 1, n, f) != n) { std::printf("Can't write %s\n", save_calib.c_str()); return -1; }
        std::fclose(f);
        std::printf("Calibration written to %s (%zu bytes)\n", save_calib.c_str(), n);
    }

    // blender set-up rule of the sample (CLI:729-749)
    mst_config cfg;
    std::memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = sizeof(cfg);
    cfg.num_cameras = num_images;
    cfg.warp_kind = warp_kind;
    cfg.expos_kind = expos_kind;
    const float blend_width = std::sqrt(static_cast<float>((brx - tlx)) * static_cast<float>(bry - tly)) *
                              static_cast<float>(blend_strength) / 100.f;
    if (blend_width < 1.f) blend_kind = MST_BLEND_NO;
    else if (blend_kind == MST_BLEND_MULTI_BAND) {
        cfg.num_bands = static_cast<int>(std::ceil(std::log(blend_width) / std::log(2.)) - 1.);
        std::printf("Multi-band blender, number of bands: %d\n", cfg.num_bands);
    } else if (blend_kind == MST_BLEND_FEATHER) {
        cfg.feather_sharpness = 1.f / blend_width;
        std::printf("Feather blender, sharpness: %g\n", cfg.feather_sharpness);
    }
    cfg.blend_kind = blend_kind;
    if (lut) {
        cfg.blend_kind = MST_BLEND_NO; cfg.flags |= MST_FLAG_DRIVER_LUT;
        cfg.upblack = cfg.downblack = 0.1f; cfg.leftblack = cfg.rightblack = 10.f;   // DRV:47
    }
    CHECK(mst_create(&cfg, &engine));
    CHECK(mst_update(engine, &cal));
    mst_rect canvas;
    CHECK(mst_canvas(engine, &canvas));
    std::printf("Warped image scale: %g, canvas [%d x %d] at (%d, %d)\n", scale, canvas.width, canvas.height, canvas.x, canvas.y);

    // the frame loop (StitchingAll): N frames in, one panorama out
    std::vector<std::vector<uint8_t>> frame(num_images, std::vector<uint8_t>(static_cast<size_t>(W) * H * 3));
    std::vector<const uint8_t*> ptrs(num_images);
    std::vector<size_t> strides(num_images, static_cast<size_t>(W) * 3);
    std::vector<uint8_t> pano(static_cast<size_t>(canvas.width) * canvas.height * 3);
    double total = 0.0;
    for (int k = 0; k < frames; ++k) {
        for (int i = 0; i < num_images; ++i) {
            if (synthetic_n) {
                if (resize_inputs) {
                    Image full; full.w = full_w; full.h = full_h; full.bgr.resize(static_cast<size_t>(full_w) * full_h * 3);
                    CHECK(mst_synth_frame(i, k, full_w, full_h, full.bgr.data(), static_cast<size_t>(full_w) * 3));
                    frame[i] = resize_linear_bgr(full, compose_scale).bgr;
                } else CHECK(mst_synth_frame(i, k, W, H, frame[i].data(), strides[i]));
            }
            ptrs[i] = synthetic_n ? frame[i].data() : stills[i].bgr.data();
        }
        const auto t0 = std::chrono::steady_clock::now();
        CHECK(mst_compose(engine, ptrs.data(), strides.data(), pano.data(), static_cast<size_t>(canvas.width) * 3, nullptr, 0));
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        total += dt;
        std::printf("Stitching, time: %g sec\n", dt);                                  // DRV:760
    }
    std::printf("Finished, %d frame(s), mean %g sec (host buffers in, panorama out)\n", frames, total / frames);
    if (!write_ppm(result_name, pano.data(), canvas.width, canvas.height)) { std::printf("Can't write %s\n", result_name.c_str()); return -1; }
    mst_destroy(engine);
    return 0;
}
