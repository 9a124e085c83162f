// multi_gpu_loop_mi355 -- the frame loop over several GPUs as a C++ host: one process per GPU, the engine through its C ABI
// (include/mi355_stitch.h), the exchange through RCCL (ncclSend / ncclRecv, /opt/rocm/include/rccl/rccl.h:700,722) -- no Python,
// no torch.  The reference's loop is C++ with OpenMP over the cameras (R64/stitching_detailed.cpp:724-770, `#pragma omp parallel
// for` at :736 and :748); across GPUs the same split is one of
//
//   --mode cameras   BASELINE C4 as the north star words it: camera i lives on rank i % N.  Per frame every rank runs the
//                    per-camera kernels of its cameras (mst_stage_cameras), packs their hand-off payload (mst_payload_pack), ONE
//                    grouped RCCL step carries the payloads to rank 0 (ncclGroupStart / ncclSend ... / ncclGroupEnd; the root
//                    posts the matching ncclRecv's), and rank 0 unpacks (mst_payload_unpack) and blends (mst_stage_canvas).
//                    Two pyramid / payload sets: the exchange of frame t is in flight while the per-camera stage of t + 1 runs.
//   --mode strips    the scaling mode: rank r composes column strip r of every panorama
//                    (mst_config.strip_index / strip_count, mst_compose_device_batch with calls in flight); no data-path
//                    collective.  --gather adds the one step a single consumer needs: every strip to rank 0 (grouped
//                    ncclSend / ncclRecv) and into the panorama's columns.
//
// Ranks: RANK / WORLD_SIZE / LOCAL_RANK from the environment (torchrun, mpirun -x), or `--ranks N` -- the program then forks its
// own N - 1 peers before it touches the GPU.  The ncclUniqueId travels through a file (--id-file, default /tmp/mst_rccl_id_<port>).
// Timing: a barrier (all-reduce), K calls, stream synchronisation, the MAX over ranks (all-reduce); rank 0 prints one JSON line.
// --check: rank 0 also composes the last frame set(s) with a plain whole-canvas engine and compares byte for byte.
//
//   multi_gpu_loop_mi355 <calibration file> <warp> <expos no|gain|gain_blocks> --mode cameras|strips [--ranks N] [--calls K]
//                        [--batch B] [--depth D] [--gather] [--check] [--id-file F]
//
// Build: make -C stitchingvideo_amd/csrc multi     (hipcc, links libmi355stitch.so and librccl.so)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/mi355_stitch.h"

static int g_rank = 0;
#define FAIL(...) do { std::fprintf(stderr, "[rank %d] ", g_rank); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); std::exit(1); } while (0)
#define MST(call) do { const int rc_ = (call); if (rc_ != MST_OK) FAIL("%s failed: %d (%s)", #call, rc_, mst_last_error(nullptr)); } while (0)
#define HIP(call) do { const hipError_t e_ = (call); if (e_ != hipSuccess) FAIL("%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define NCCL(call) do { const ncclResult_t r_ = (call); if (r_ != ncclSuccess) FAIL("%s failed: %s", #call, ncclGetErrorString(r_)); } while (0)

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static int env_int(const char* name, int fallback)
{
    const char* v = std::getenv(name);
    return v && *v ? std::atoi(v) : fallback;
}

struct Job {
    int rank = 0, world = 1, local = 0;
    ncclComm_t comm = nullptr;
    hipStream_t sc = nullptr, sx = nullptr;          // compute / exchange
    double* d_scalar = nullptr;
    // all-reduce of one double on the exchange stream (barrier, MAX of the elapsed times)
    double all_reduce(double v, ncclRedOp_t op)
    {
        HIP(hipMemcpyAsync(d_scalar, &v, sizeof v, hipMemcpyHostToDevice, sx));
        if (world > 1) NCCL(ncclAllReduce(d_scalar, d_scalar, 1, ncclDouble, op, comm, sx));
        HIP(hipMemcpyAsync(&v, d_scalar, sizeof v, hipMemcpyDeviceToHost, sx));
        HIP(hipStreamSynchronize(sx));
        return v;
    }
    void barrier() { HIP(hipStreamSynchronize(sc)); (void)all_reduce(1.0, ncclSum); }
};

static void exchange_unique_id(const std::string& path, int rank, ncclUniqueId* id)
{
    if (rank == 0) {
        NCCL(ncclGetUniqueId(id));
        const std::string tmp = path + ".tmp";
        FILE* f = std::fopen(tmp.c_str(), "wb");
        if (!f || std::fwrite(id, sizeof *id, 1, f) != 1) FAIL("cannot write %s", tmp.c_str());
        std::fclose(f);
        if (std::rename(tmp.c_str(), path.c_str()) != 0) FAIL("cannot publish %s", path.c_str());
        return;
    }
    for (int tries = 0; tries < 1200; ++tries) {            // (a minute)
        struct stat st;
        if (::stat(path.c_str(), &st) == 0 && st.st_size == static_cast<off_t>(sizeof *id)) {
            FILE* f = std::fopen(path.c_str(), "rb");
            if (f && std::fread(id, sizeof *id, 1, f) == 1) { std::fclose(f); return; }
            if (f) std::fclose(f);
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    FAIL("no ncclUniqueId in %s after 60 s (is rank 0 running?)", path.c_str());
}

int main(int argc, char** argv)
{
    if (argc < 4) {
        std::printf("usage: multi_gpu_loop_mi355 <calibration file> <warp> <expos no|gain|gain_blocks> --mode cameras|strips [--ranks N] "
                    "[--calls K] [--batch B] [--depth D] [--gather] [--check] [--id-file F]\n");
        return 2;
    }
    std::string mode = "strips", id_file;
    int spawn = 0, calls = 400, batch = 1, depth = 4;
    bool gather = false, check = false;
    for (int i = 4; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&] { if (i + 1 >= argc) { std::printf("%s needs a value\n", a.c_str()); std::exit(2); } return std::string(argv[++i]); };
        if (a == "--mode") mode = next();
        else if (a == "--ranks") spawn = std::atoi(next().c_str());
        else if (a == "--calls") calls = std::atoi(next().c_str());
        else if (a == "--batch") batch = std::atoi(next().c_str());
        else if (a == "--depth") depth = std::atoi(next().c_str());
        else if (a == "--id-file") id_file = next();
        else if (a == "--gather") gather = true;
        else if (a == "--check") check = true;
        else { std::printf("bad flag %s\n", a.c_str()); return 2; }
    }
    if (mode != "cameras" && mode != "strips") { std::printf("--mode cameras|strips\n"); return 2; }
    Job job;
    job.rank = env_int("RANK", 0);
    job.world = env_int("WORLD_SIZE", 1);
    job.local = env_int("LOCAL_RANK", job.rank);
    if (id_file.empty()) id_file = "/tmp/mst_rccl_id_" + std::to_string(env_int("MASTER_PORT", static_cast<int>(::getppid())));
    std::vector<pid_t> children;
    if (spawn > 1 && !std::getenv("WORLD_SIZE")) {
        // our own launcher: fork the peers BEFORE anything initialises the GPU (a process that has must not fork workers)
        id_file += "_" + std::to_string(static_cast<int>(::getpid()));
        job.world = spawn;
        for (int r = 1; r < spawn; ++r) {
            const pid_t pid = ::fork();
            if (pid < 0) FAIL("fork failed");
            if (pid == 0) { job.rank = job.local = r; children.clear(); break; }
            children.push_back(pid);
        }
    }
    g_rank = job.rank;
    if (job.rank == 0) (void)::unlink(id_file.c_str());

    int ndev = 0;
    HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) FAIL("no GPU");
    const int device = job.local % ndev;
    HIP(hipSetDevice(device));
    if (job.world > ndev && job.rank == 0)
        std::fprintf(stderr, "[rank 0] %d ranks on %d visible GPU(s): RCCL refuses two ranks of one communicator on one device\n", job.world, ndev);
    ncclUniqueId id;
    exchange_unique_id(id_file, job.rank, &id);
    NCCL(ncclCommInitRank(&job.comm, job.world, id, job.rank));
    HIP(hipStreamCreateWithFlags(&job.sc, hipStreamNonBlocking));
    HIP(hipStreamCreateWithFlags(&job.sx, hipStreamNonBlocking));
    HIP(hipMalloc(&job.d_scalar, sizeof(double)));

    // ---- the calibration (every rank publishes the same one)
    FILE* f = std::fopen(argv[1], "rb");
    if (!f) FAIL("cannot open %s", argv[1]);
    std::fseek(f, 0, SEEK_END);
    const long blob_bytes = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> blob((blob_bytes + 7) / 8);
    if (std::fread(blob.data(), 1, blob_bytes, f) != static_cast<size_t>(blob_bytes)) FAIL("short read");
    std::fclose(f);
    mst_calibration cal;
    mst_camera cams[MST_MAX_CAMERAS];
    std::memset(&cal, 0, sizeof cal);
    MST(mst_calibration_load(blob.data(), static_cast<size_t>(blob_bytes), &cal, cams, MST_MAX_CAMERAS));
    const int n = cal.num_cameras;
    const std::string expos = argv[3];
    auto make_engine = [&](int strip_index, int strip_count, int max_batch, int in_flight) {
        mst_config cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.struct_size = sizeof cfg;
        cfg.num_cameras = n;
        cfg.warp_kind = mst_warp_kind_from_name(argv[2]);
        cfg.blend_kind = MST_BLEND_MULTI_BAND;
        cfg.expos_kind = expos == "gain_blocks" ? MST_EXPOS_GAIN_BLOCKS : expos == "gain" ? MST_EXPOS_GAIN : MST_EXPOS_NO;
        cfg.num_bands = 5;
        cfg.device = device;
        cfg.max_batch = max_batch;
        cfg.strip_index = strip_index;
        cfg.strip_count = strip_count;
        cfg.flags = in_flight > 2 ? MST_FLAG_OVERLAP_FRAMES : 0u;
        cfg.overlap_depth = in_flight > 2 ? in_flight : 0;
        mst_engine* e = nullptr;
        MST(mst_create(&cfg, &e));
        MST(mst_update(e, &cal));
        return e;
    };
    // synthetic frames in HBM (kSets distinct sets): all cameras, or only those in `mask`
    const int kSets = 3;
    std::vector<void*> d_frames(static_cast<size_t>(kSets) * n, nullptr);
    std::vector<size_t> strides(static_cast<size_t>(MST_MAX_BATCH) * n);
    auto upload_frames = [&](uint32_t mask) {
        for (int s = 0; s < kSets; ++s)
            for (int i = 0; i < n; ++i) {
                if (!(mask >> i & 1u) || d_frames[static_cast<size_t>(s) * n + i]) continue;
                const int w = cams[i].src_width, h = cams[i].src_height;
                std::vector<uint8_t> host(static_cast<size_t>(w) * h * 3);
                MST(mst_synth_frame(i, s, w, h, host.data(), static_cast<size_t>(w) * 3));
                HIP(hipMalloc(&d_frames[static_cast<size_t>(s) * n + i], host.size()));
                HIP(hipMemcpy(d_frames[static_cast<size_t>(s) * n + i], host.data(), host.size(), hipMemcpyHostToDevice));
            }
    };
    for (int b = 0; b < MST_MAX_BATCH; ++b)
        for (int i = 0; i < n; ++i) strides[static_cast<size_t>(b) * n + i] = static_cast<size_t>(cams[i].src_width) * 3;
    // the whole-canvas reference of --check (rank 0): frame set s -> host bytes
    mst_rect canvas{};
    auto reference = [&](int s, std::vector<uint8_t>& out) {
        upload_frames(~0u);
        mst_engine* ref = make_engine(0, 0, 1, 0);
        MST(mst_canvas(ref, &canvas));
        const size_t bytes = static_cast<size_t>(canvas.width) * 3 * canvas.height;
        void* d = nullptr;
        HIP(hipMalloc(&d, bytes));
        std::vector<const void*> ft(n);
        for (int i = 0; i < n; ++i) ft[i] = d_frames[static_cast<size_t>(s) * n + i];
        MST(mst_compose_device(ref, ft.data(), strides.data(), d, static_cast<size_t>(canvas.width) * 3, nullptr, 0, job.sc));
        HIP(hipStreamSynchronize(job.sc));
        out.resize(bytes);
        HIP(hipMemcpy(out.data(), d, bytes, hipMemcpyDeviceToHost));
        HIP(hipFree(d));
        mst_destroy(ref);
    };
    double alg_bytes = 0;
    int check_ok = -1;
    double elapsed = 0;
    long panoramas = 0;
    size_t exchanged_bytes_per_panorama = 0;

    if (mode == "cameras") {
        // ---------------------------------------------------------------- camera i on rank i % world, root 0 blends
        const int root = 0;
        uint32_t mine = 0;
        for (int c = 0; c < n; ++c) if (c % job.world == job.rank) mine |= 1u << c;
        upload_frames(mine);
        mst_engine* e = make_engine(0, 0, 2, 0);         // two pyramid sets
        uint64_t gen;
        MST(mst_acquire_calibration(e, &canvas, &gen));
        std::vector<size_t> pay(n);
        for (int c = 0; c < n; ++c) MST(mst_payload_bytes(e, c, &pay[c]));
        // per set: send buffers of my cameras (ranks != root), landing buffers of the others' cameras (root)
        std::vector<void*> buf(static_cast<size_t>(2) * n, nullptr);
        for (int k = 0; k < 2; ++k)
            for (int c = 0; c < n; ++c) {
                const bool remote = c % job.world != root;
                if (remote && (job.rank == root || (mine >> c & 1u))) HIP(hipMalloc(&buf[static_cast<size_t>(k) * n + c], pay[c]));
                if (remote && k == 0) exchanged_bytes_per_panorama += pay[c];
            }
        void* d_pano = nullptr;
        const size_t pano_stride = static_cast<size_t>(canvas.width) * 3;
        if (job.rank == root) HIP(hipMalloc(&d_pano, pano_stride * canvas.height));
        hipEvent_t packed[2], moved[2], freed[2];
        for (int k = 0; k < 2; ++k) {
            HIP(hipEventCreateWithFlags(&packed[k], hipEventDisableTiming));
            HIP(hipEventCreateWithFlags(&moved[k], hipEventDisableTiming));
            HIP(hipEventCreateWithFlags(&freed[k], hipEventDisableTiming));
        }
        bool moved_valid[2] = {false, false}, freed_valid[2] = {false, false};
        std::vector<const void*> ft(n, nullptr);
        auto submit = [&](int t) {          // per-camera stage of frame t + its hand-off towards the root
            const int k = t & 1;
            for (int c = 0; c < n; ++c) ft[c] = d_frames[static_cast<size_t>(t % kSets) * n + c];
            MST(mst_stage_select_set(e, k));
            MST(mst_stage_cameras(e, mine, ft.data(), strides.data(), job.sc));
            if (job.world == 1) return;
            if (job.rank != root) {
                if (moved_valid[k]) HIP(hipStreamWaitEvent(job.sc, moved[k], 0));       // the send of frame t - 2 has left these buffers
                for (int c = 0; c < n; ++c)
                    if (mine >> c & 1u) MST(mst_payload_pack(e, c, buf[static_cast<size_t>(k) * n + c], job.sc));
                HIP(hipEventRecord(packed[k], job.sc));
                HIP(hipStreamWaitEvent(job.sx, packed[k], 0));
                NCCL(ncclGroupStart());
                for (int c = 0; c < n; ++c)
                    if (mine >> c & 1u) NCCL(ncclSend(buf[static_cast<size_t>(k) * n + c], pay[c], ncclUint8, root, job.comm, job.sx));
                NCCL(ncclGroupEnd());
            } else {
                if (freed_valid[k]) HIP(hipStreamWaitEvent(job.sx, freed[k], 0));       // frame t - 2 has been unpacked out of them
                NCCL(ncclGroupStart());
                for (int c = 0; c < n; ++c)
                    if (c % job.world != root) NCCL(ncclRecv(buf[static_cast<size_t>(k) * n + c], pay[c], ncclUint8, c % job.world, job.comm, job.sx));
                NCCL(ncclGroupEnd());
            }
            HIP(hipEventRecord(moved[k], job.sx));
            moved_valid[k] = true;
        };
        auto finish = [&](int t) {          // root: unpack frame t's payloads, blend
            if (job.rank != root) return;
            const int k = t & 1;
            MST(mst_stage_select_set(e, k));
            if (job.world > 1) {
                HIP(hipStreamWaitEvent(job.sc, moved[k], 0));
                for (int c = 0; c < n; ++c)
                    if (c % job.world != root) MST(mst_payload_unpack(e, c, buf[static_cast<size_t>(k) * n + c], job.sc));
                HIP(hipEventRecord(freed[k], job.sc));
                freed_valid[k] = true;
            }
            MST(mst_stage_canvas(e, d_pano, pano_stride, nullptr, 0, job.sc));
        };
        auto loop = [&](int count) {
            for (int t = 0; t <= count; ++t) {
                if (t < count) submit(t);
                if (t > 0) finish(t - 1);
            }
        };
        loop(8);
        job.barrier();
        const double t0 = now_s();
        loop(calls);
        HIP(hipStreamSynchronize(job.sc));
        HIP(hipStreamSynchronize(job.sx));
        elapsed = job.all_reduce(now_s() - t0, ncclMax);
        panoramas = calls;
        if (check && job.rank == root) {
            std::vector<uint8_t> want, got(pano_stride * canvas.height);
            HIP(hipMemcpy(got.data(), d_pano, got.size(), hipMemcpyDeviceToHost));
            reference((calls - 1) % kSets, want);
            check_ok = want == got ? 1 : 0;
        }
        mst_destroy(e);
    } else {
        // ---------------------------------------------------------------- rank r composes column strip r of every panorama
        upload_frames(~0u);
        if (batch < 1 || batch > MST_MAX_BATCH) FAIL("--batch 1..%d", MST_MAX_BATCH);
        mst_engine* e = make_engine(job.rank, job.world, batch, depth);
        uint64_t gen;
        MST(mst_acquire_calibration(e, &canvas, &gen));
        mst_rect strip;
        MST(mst_strip(e, &strip));
        const int x0 = strip.x - canvas.x, w = strip.width, H = canvas.height;
        const size_t strip_stride = static_cast<size_t>(std::max(w, 1)) * 3, strip_bytes = strip_stride * H;
        const int ring = std::max(depth, 1);
        std::vector<void*> d_strips(static_cast<size_t>(ring) * batch);
        for (void*& p : d_strips) HIP(hipMalloc(&p, strip_bytes));
        // every rank's (x, w): one all-gather per calibration
        std::vector<int> rects(static_cast<size_t>(2) * job.world, 0);
        {
            int* d_r = nullptr;
            HIP(hipMalloc(&d_r, sizeof(int) * 2 * job.world));
            const int my[2] = {x0, w};
            HIP(hipMemcpy(d_r + 2 * job.rank, my, sizeof my, hipMemcpyHostToDevice));
            if (job.world > 1) NCCL(ncclAllGather(d_r + 2 * job.rank, d_r, 2, ncclInt32, job.comm, job.sx));
            HIP(hipStreamSynchronize(job.sx));
            HIP(hipMemcpy(rects.data(), d_r, sizeof(int) * 2 * job.world, hipMemcpyDeviceToHost));
            HIP(hipFree(d_r));
        }
        // root: landing buffers for the other ranks' strips (per frame set of a call) and the assembled panoramas
        const int root = 0;
        const size_t pano_stride = static_cast<size_t>(canvas.width) * 3;
        std::vector<void*> d_panos(batch, nullptr), land(static_cast<size_t>(job.world) * batch, nullptr);
        if (gather && job.rank == root) {
            for (int b = 0; b < batch; ++b) HIP(hipMalloc(&d_panos[b], pano_stride * H));
            for (int r = 0; r < job.world; ++r)
                for (int b = 0; b < batch && r != root; ++b)
                    if (rects[2 * r + 1] > 0) HIP(hipMalloc(&land[static_cast<size_t>(r) * batch + b], static_cast<size_t>(rects[2 * r + 1]) * 3 * H));
        }
        for (int r = 0; r < job.world; ++r)
            if (r != root) exchanged_bytes_per_panorama += static_cast<size_t>(rects[2 * r + 1]) * 3 * H;
        std::vector<hipEvent_t> done(ring), sent(ring);
        std::vector<bool> sent_valid(ring, false);
        for (int k = 0; k < ring; ++k) {
            HIP(hipEventCreateWithFlags(&done[k], hipEventDisableTiming));
            HIP(hipEventCreateWithFlags(&sent[k], hipEventDisableTiming));
        }
        std::vector<const void*> ft(static_cast<size_t>(batch) * n);
        std::vector<void*> panos(batch);
        auto call = [&](int t) {
            const int k = t % ring;
            for (int b = 0; b < batch; ++b) {
                for (int i = 0; i < n; ++i) ft[static_cast<size_t>(b) * n + i] = d_frames[static_cast<size_t>((t + b) % kSets) * n + i];
                // (the engine addresses a panorama from canvas column 0: the strip buffer shifted left by the strip's origin)
                panos[b] = static_cast<uint8_t*>(d_strips[static_cast<size_t>(k) * batch + b]) - static_cast<size_t>(x0) * 3;
            }
            if (w > 0) {
                // the strip buffers of this ring slot are free once their last send (or copy into the panorama) has completed
                hipEvent_t out_free = gather && sent_valid[k] ? sent[k] : nullptr;
                MST(mst_compose_device_batch_ex(e, batch, ft.data(), strides.data(), panos.data(), strip_stride, nullptr, 0, job.sc, nullptr, out_free));
            }
            if (!gather) return;
            HIP(hipEventRecord(done[k], job.sc));
            HIP(hipStreamWaitEvent(job.sx, done[k], 0));
            if (job.world > 1) {
                NCCL(ncclGroupStart());
                if (job.rank != root) {
                    for (int b = 0; b < batch && w > 0; ++b)
                        NCCL(ncclSend(d_strips[static_cast<size_t>(k) * batch + b], strip_bytes, ncclUint8, root, job.comm, job.sx));
                } else {
                    for (int r = 0; r < job.world; ++r)
                        for (int b = 0; b < batch && r != root; ++b)
                            if (rects[2 * r + 1] > 0)
                                NCCL(ncclRecv(land[static_cast<size_t>(r) * batch + b], static_cast<size_t>(rects[2 * r + 1]) * 3 * H, ncclUint8, r, job.comm, job.sx));
                }
                NCCL(ncclGroupEnd());
            }
            if (job.rank == root)
                for (int b = 0; b < batch; ++b)
                    for (int r = 0; r < job.world; ++r) {
                        const int rx = rects[2 * r], rw = rects[2 * r + 1];
                        if (rw <= 0) continue;
                        const void* src = r == root ? d_strips[static_cast<size_t>(k) * batch + b] : land[static_cast<size_t>(r) * batch + b];
                        HIP(hipMemcpy2DAsync(static_cast<uint8_t*>(d_panos[b]) + static_cast<size_t>(rx) * 3, pano_stride, src, static_cast<size_t>(rw) * 3,
                                             static_cast<size_t>(rw) * 3, H, hipMemcpyDeviceToDevice, job.sx));
                    }
            HIP(hipEventRecord(sent[k], job.sx));
            sent_valid[k] = true;
        };
        for (int t = 0; t < 3 * ring; ++t) call(t);
        job.barrier();
        const double t0 = now_s();
        for (int t = 0; t < calls; ++t) call(t);
        HIP(hipStreamSynchronize(job.sc));
        HIP(hipStreamSynchronize(job.sx));
        elapsed = job.all_reduce(now_s() - t0, ncclMax);
        panoramas = static_cast<long>(calls) * batch;
        if (check && job.rank == root) {
            std::vector<uint8_t> want;
            check_ok = 1;
            for (int b = 0; b < batch; ++b) {
                reference((calls - 1 + b) % kSets, want);
                if (gather) {
                    std::vector<uint8_t> got(pano_stride * H);
                    HIP(hipMemcpy(got.data(), d_panos[b], got.size(), hipMemcpyDeviceToHost));
                    if (got != want) check_ok = 0;
                } else if (w > 0) {          // strips left in place: rank 0's own columns
                    std::vector<uint8_t> got(strip_bytes);
                    HIP(hipMemcpy(got.data(), d_strips[static_cast<size_t>((calls - 1) % ring) * batch + b], got.size(), hipMemcpyDeviceToHost));
                    for (int y = 0; y < H && check_ok; ++y)
                        if (std::memcmp(&got[static_cast<size_t>(y) * strip_stride], &want[static_cast<size_t>(y) * pano_stride + static_cast<size_t>(x0) * 3], static_cast<size_t>(w) * 3) != 0)
                            check_ok = 0;
                }
            }
        }
        mst_destroy(e);
    }
    alg_bytes = 3.0 * canvas.width * canvas.height;
    for (int i = 0; i < n; ++i) alg_bytes += 3.0 * cams[i].src_width * cams[i].src_height;
    job.barrier();
    if (job.rank == 0) {
        std::printf("{\"tool\": \"multi_gpu_loop_mi355\", \"mode\": \"%s\", \"ranks\": %d, \"cameras\": %d, \"canvas\": [%d, %d], \"calls\": %d, "
                    "\"frame_sets_per_call\": %d, \"calls_in_flight\": %d, \"gather_to_rank0\": %s, \"panoramas_per_s\": %.1f, \"us_per_panorama\": %.2f, "
                    "\"exchanged_bytes_per_panorama\": %zu, \"roofline_frac_of_one_gpu\": %.4f, \"check\": %s}\n",
                    mode.c_str(), job.world, n, canvas.width, canvas.height, calls, mode == "cameras" ? 1 : batch, mode == "cameras" ? 1 : depth,
                    (mode == "cameras" || gather) ? "true" : "false", panoramas / elapsed, elapsed * 1e6 / panoramas, exchanged_bytes_per_panorama,
                    alg_bytes * panoramas / elapsed / 8.0e12, check_ok < 0 ? "null" : check_ok ? "\"bit-exact\"" : "\"MISMATCH\"");
        std::fflush(stdout);
        (void)::unlink(id_file.c_str());
    }
    NCCL(ncclCommDestroy(job.comm));
    int rc = check_ok == 0 ? 1 : 0;
    for (pid_t pid : children) {
        int status = 0;
        if (::waitpid(pid, &status, 0) < 0 || !WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = 1;
    }
    return rc;
}
