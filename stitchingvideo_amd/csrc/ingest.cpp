// ingest.cpp -- include/mi355_ingest.h: the reference's shared-memory frame hand-off (Stitching_sharemem,
// R64/stitching_sharemem.{h,cpp}) over POSIX shared memory and process-shared semaphores.
#include "../../include/mi355_ingest.h"

#include <fcntl.h>
#include <sched.h>
#include <semaphore.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <emmintrin.h>

#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include <hip/hip_runtime.h>

#include "../../include/mi355_stitch.h"

namespace {

constexpr uint32_t kMagic = 0x4D535446u;        // "MSTF"

// Control block at the start of the segment; the reference's layout (packet, then pixels at buferPointer + 1)
// starts at data_offset.
struct Control {
    uint32_t magic;
    uint32_t data_offset;
    uint64_t segment_bytes;
    uint64_t max_frame_bytes;
    sem_t client_ready;           // hClientReadOver: "the client wants a frame"
    sem_t server_wrote;           // hServerWriteOver: "header and pixels are in place"
};

constexpr size_t kDataOffset = (sizeof(Control) + 63) & ~size_t{63};

int wait_sem(sem_t* s, int timeout_ms)
{
    if (timeout_ms < 0) {
        while (sem_wait(s) != 0)
            if (errno != EINTR) return MST_ERR_INVALID_ARG;
        return MST_OK;
    }
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += timeout_ms / 1000;
    ts.tv_nsec += static_cast<long>(timeout_ms % 1000) * 1000000L;
    if (ts.tv_nsec >= 1000000000L) { ts.tv_sec += 1; ts.tv_nsec -= 1000000000L; }
    while (sem_timedwait(s, &ts) != 0) {
        if (errno == EINTR) continue;
        return errno == ETIMEDOUT ? MST_ERR_TIMEOUT : MST_ERR_INVALID_ARG;
    }
    return MST_OK;
}

}  // namespace

struct mst_shm {
    std::string name;
    bool server = false;
    int fd = -1;
    void* base = nullptr;
    size_t bytes = 0;
    bool registered = false;      // page-locked for the device copies
    bool requested = false;       // a request is outstanding (mst_shm_get_device_multi_ex with MST_SHM_PREFETCH left one behind)
    hipEvent_t copied = nullptr;  // "this camera's pixels have left the segment" (created on first use)
    Control* ctl() const { return static_cast<Control*>(base); }
    uint8_t* data() const { return static_cast<uint8_t*>(base) + ctl()->data_offset; }
};

extern "C" {

int mst_shm_serve(const char* name, size_t max_frame_bytes, mst_shm** out)
{
    if (out) *out = nullptr;
    if (!name || !out || name[0] != '/' || max_frame_bytes == 0) return MST_ERR_INVALID_ARG;
    mst_shm* s = new (std::nothrow) mst_shm;
    if (!s) return MST_ERR_OUT_OF_MEMORY;
    s->name = name;
    s->server = true;
    s->bytes = kDataOffset + sizeof(mst_frame_packet) + 1 + max_frame_bytes;
    shm_unlink(name);
    s->fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (s->fd < 0 || ftruncate(s->fd, static_cast<off_t>(s->bytes)) != 0) { mst_shm_close(s); return MST_ERR_INVALID_ARG; }
    s->base = mmap(nullptr, s->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, s->fd, 0);
    if (s->base == MAP_FAILED) { s->base = nullptr; mst_shm_close(s); return MST_ERR_OUT_OF_MEMORY; }
    Control* c = s->ctl();
    std::memset(c, 0, kDataOffset);
    c->data_offset = static_cast<uint32_t>(kDataOffset);
    c->segment_bytes = s->bytes;
    c->max_frame_bytes = max_frame_bytes;
    if (sem_init(&c->client_ready, 1, 0) != 0 || sem_init(&c->server_wrote, 1, 0) != 0) { mst_shm_close(s); return MST_ERR_INVALID_ARG; }
    __atomic_store_n(&c->magic, kMagic, __ATOMIC_RELEASE);          // attachers look for this last
    *out = s;
    return MST_OK;
}

namespace {
// The frame's bytes into the segment with non-temporal stores: the capture process never reads them again (the consumer's DMA engine
// does), so an ordinary copy would first READ every destination line into this core's caches (read-for-ownership) and evict what the
// capture process works on -- a third of the copy's memory traffic, and with six capture processes beside two DMA streams it is the
// host's memory system that bounds the frame loop.  OPT-IN (MST_SHM_STREAM_COPY=1): a consumer that copies
// with the CPU (mst_shm_get) then misses on every line (measured here: 1.26 against 1.07 ms per 1080p frame), and with the device
// consumer the gain measured on the GPU box (+ 7 % in both A/B pairs) is inside that loop's run-to-run scatter.
void copy_into_segment(uint8_t* dst, const uint8_t* src, size_t n)
{
    static const bool stream = [] { const char* v = std::getenv("MST_SHM_STREAM_COPY"); return v && v[0] == '1'; }();
    if (!stream || n < 4096) { std::memcpy(dst, src, n); return; }
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(dst) & 15)) & 15;          // the pixels start at packet + buferPointer + 1: odd
    std::memcpy(dst, src, head);
    dst += head; src += head; n -= head;
    size_t i = 0;
    for (; i + 64 <= n; i += 64) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 16));
        const __m128i c = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 32));
        const __m128i d = _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + i + 48));
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i), a);
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 32), c);
        _mm_stream_si128(reinterpret_cast<__m128i*>(dst + i + 48), d);
    }
    std::memcpy(dst + i, src + i, n - i);
    _mm_sfence();                                             // the stores are globally visible before "server wrote" is posted
}

// header + `rows` rows of `row` bytes each, packed tightly behind the packet
int publish_rows(mst_shm* s, const char* file_path, int width, int height, const uint8_t* src, size_t stride, size_t row, size_t rows,
                 int timeout_ms)
{
    const size_t nbytes = row * rows;
    if (nbytes > s->ctl()->max_frame_bytes || nbytes > 0x7FFFFFFFu) return MST_ERR_INVALID_ARG;
    const int rc = wait_sem(&s->ctl()->client_ready, timeout_ms);
    if (rc != MST_OK) return rc;
    mst_frame_packet p;
    std::memset(&p, 0, sizeof(p));
    if (file_path) std::strncpy(p.filePath, file_path, sizeof(p.filePath) - 1);
    p.width = width;
    p.height = height;
    p.flag = static_cast<int32_t>(nbytes);
    p.buferPointer = static_cast<int32_t>(sizeof(mst_frame_packet));     // pixels at packet + buferPointer + 1
    uint8_t* q = s->data();
    std::memcpy(q, &p, sizeof(p));
    uint8_t* px = q + p.buferPointer + 1;
    if (stride == row) copy_into_segment(px, src, nbytes);
    else for (size_t y = 0; y < rows; ++y) copy_into_segment(px + y * row, src + y * stride, row);
    sem_post(&s->ctl()->server_wrote);
    return MST_OK;
}
}  // namespace

int mst_shm_publish(mst_shm* s, const char* file_path, int width, int height, const uint8_t* bgr, size_t stride,
                    int timeout_ms)
{
    if (!s || !s->server || !bgr || width <= 0 || height <= 0 || stride < static_cast<size_t>(width) * 3)
        return MST_ERR_INVALID_ARG;
    return publish_rows(s, file_path, width, height, bgr, stride, static_cast<size_t>(width) * 3, static_cast<size_t>(height), timeout_ms);
}

int mst_shm_publish_nv12(mst_shm* s, const char* file_path, int width, int height, const uint8_t* yuv, size_t stride,
                         int timeout_ms)
{
    if (!s || !s->server || !yuv || width <= 0 || height <= 0 || (width & 1) || (height & 1) || stride < static_cast<size_t>(width))
        return MST_ERR_INVALID_ARG;
    return publish_rows(s, file_path, width, height, yuv, stride, static_cast<size_t>(width), static_cast<size_t>(height) * 3 / 2, timeout_ms);
}

int mst_shm_packet_format(const mst_frame_packet* hdr)
{
    if (!hdr || hdr->width <= 0 || hdr->height <= 0 || hdr->flag < 0) return MST_ERR_INVALID_ARG;
    const size_t px = static_cast<size_t>(hdr->width) * static_cast<size_t>(hdr->height), flag = static_cast<size_t>(hdr->flag);
    if (flag == px * 3) return MST_PIXEL_BGR;
    if (!(hdr->width & 1) && !(hdr->height & 1) && flag == px * 3 / 2) return MST_PIXEL_NV12;
    return MST_ERR_INVALID_ARG;
}

int mst_shm_attach(const char* name, mst_shm** out)
{
    if (out) *out = nullptr;
    if (!name || !out || name[0] != '/') return MST_ERR_INVALID_ARG;
    mst_shm* s = new (std::nothrow) mst_shm;
    if (!s) return MST_ERR_OUT_OF_MEMORY;
    s->name = name;
    s->fd = shm_open(name, O_RDWR, 0600);
    struct stat st;
    if (s->fd < 0 || fstat(s->fd, &st) != 0 || static_cast<size_t>(st.st_size) < kDataOffset + sizeof(mst_frame_packet) + 2) {
        mst_shm_close(s);
        return MST_ERR_INVALID_ARG;          // "OpenFileMapping fail" (cpp:50-58)
    }
    s->bytes = static_cast<size_t>(st.st_size);
    s->base = mmap(nullptr, s->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, s->fd, 0);
    if (s->base == MAP_FAILED) { s->base = nullptr; mst_shm_close(s); return MST_ERR_OUT_OF_MEMORY; }
    const Control* c = s->ctl();
    if (__atomic_load_n(&c->magic, __ATOMIC_ACQUIRE) != kMagic || c->segment_bytes != s->bytes ||
        c->data_offset != kDataOffset) {
        mst_shm_close(s);
        return MST_ERR_INVALID_ARG;
    }
    *out = s;
    return MST_OK;
}

namespace {
// GetMemData in two halves: raise "client ready" ...
int post_request(mst_shm* s)
{
    if (!s || s->server) return MST_ERR_INVALID_ARG;
    sem_post(&s->ctl()->client_ready);                                   // SetEvent(hClientReadOver)
    return MST_OK;
}
// ... and wait for "server wrote", then validate the header the server wrote
int take_frame(mst_shm* s, mst_frame_packet* hdr, const uint8_t** pixels, int timeout_ms)
{
    const int rc = wait_sem(&s->ctl()->server_wrote, timeout_ms);        // WaitForSingleObject(hServerWriteOver)
    if (rc != MST_OK) {
        // withdraw the request if the server has not taken it; if it has, the frame is on its way and has to be
        // consumed (a stale "server wrote" would let the next request overlap with the server's next write)
        if (sem_trywait(&s->ctl()->client_ready) == 0) return rc;
        const int rc2 = wait_sem(&s->ctl()->server_wrote, -1);
        if (rc2 != MST_OK) return rc2;
    }
    const uint8_t* q = s->data();
    std::memcpy(hdr, q, sizeof(*hdr));
    hdr->filePath[sizeof(hdr->filePath) - 1] = 0;
    const size_t avail = s->bytes - s->ctl()->data_offset;
    if (hdr->width <= 0 || hdr->height <= 0 || hdr->flag < 0 || hdr->buferPointer < 0 ||
        static_cast<size_t>(hdr->buferPointer) + 1 + static_cast<size_t>(hdr->flag) > avail || mst_shm_packet_format(hdr) < 0)
        return MST_ERR_INVALID_ARG;
    *pixels = q + hdr->buferPointer + 1;
    return MST_OK;
}
int request_frame(mst_shm* s, mst_frame_packet* hdr, const uint8_t** pixels, int timeout_ms)
{
    if (!hdr) return MST_ERR_INVALID_ARG;
    // (a request a prefetching multi-get left outstanding is the one this call consumes)
    const int rc = s && s->requested ? MST_OK : post_request(s);
    if (rc != MST_OK) return rc;
    s->requested = false;
    return take_frame(s, hdr, pixels, timeout_ms);
}
// rows of a packet's pixel block: bytes per row, number of rows (BGR: h rows of 3 w; NV12: 3 h / 2 rows of w)
void packet_rows(const mst_frame_packet& h, size_t* row, size_t* rows)
{
    const bool nv12 = mst_shm_packet_format(&h) == MST_PIXEL_NV12;
    *row = static_cast<size_t>(h.width) * (nv12 ? 1 : 3);
    *rows = nv12 ? static_cast<size_t>(h.height) * 3 / 2 : static_cast<size_t>(h.height);
}
// page-lock the segment once: device copies then run as DMA straight out of the shared mapping
void register_segment(mst_shm* s)
{
    if (s->registered) return;
    s->registered = hipHostRegister(s->base, s->bytes, hipHostRegisterDefault) == hipSuccess;
    (void)hipGetLastError();
}
// (stream callback of MST_SHM_ASYNC: "this camera's pixels have left the segment" -> SetEvent(hClientReadOver) for its next frame)
void release_segment(void* p) { sem_post(&static_cast<mst_shm*>(p)->ctl()->client_ready); }
hipError_t queue_copy(void* d_dst, size_t dst_stride, const uint8_t* px, size_t row, size_t rows, hipStream_t st)
{
    return dst_stride == row ? hipMemcpyAsync(d_dst, px, row * rows, hipMemcpyHostToDevice, st)
                             : hipMemcpy2DAsync(d_dst, dst_stride, px, row, row, rows, hipMemcpyHostToDevice, st);
}
}  // namespace

int mst_shm_get(mst_shm* s, mst_frame_packet* hdr, uint8_t* dst, size_t dst_stride, size_t dst_capacity, int timeout_ms)
{
    const uint8_t* px = nullptr;
    const int rc = request_frame(s, hdr, &px, timeout_ms);
    if (rc != MST_OK || !dst) return rc;
    size_t row, rows;
    packet_rows(*hdr, &row, &rows);
    if (dst_stride < row || dst_capacity < dst_stride * (rows - 1) + row) return MST_ERR_INVALID_ARG;
    for (size_t y = 0; y < rows; ++y) std::memcpy(dst + y * dst_stride, px + y * row, row);
    return MST_OK;
}

int mst_shm_get_device(mst_shm* s, mst_frame_packet* hdr, void* d_dst, size_t dst_stride, size_t dst_capacity,
                       void* stream, int timeout_ms)
{
    if (!d_dst) return MST_ERR_INVALID_ARG;
    const uint8_t* px = nullptr;
    const int rc = request_frame(s, hdr, &px, timeout_ms);
    if (rc != MST_OK) return rc;
    size_t row, rows;
    packet_rows(*hdr, &row, &rows);
    if (dst_stride < row || dst_capacity < dst_stride * (rows - 1) + row) return MST_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MST_ERR_NO_DEVICE;
    register_segment(s);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = queue_copy(d_dst, dst_stride, px, row, rows, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);       // the server may overwrite the segment after the next request
    return e == hipSuccess ? MST_OK : MST_ERR_HIP;
}

int mst_shm_get_device_multi_ex(mst_shm* const* s, int n, mst_frame_packet* hdr, void* const* d_dst, const size_t* dst_stride,
                                const size_t* dst_capacity, void* stream, int timeout_ms, int flags)
{
    if (!s || !hdr || !d_dst || !dst_stride || !dst_capacity || n <= 0 || n > MST_MAX_CAMERAS) return MST_ERR_INVALID_ARG;
    for (int i = 0; i < n; ++i)
        if (!s[i] || s[i]->server || !d_dst[i]) return MST_ERR_INVALID_ARG;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return MST_ERR_NO_DEVICE;
    const bool async_release = (flags & MST_SHM_ASYNC) != 0;
    const bool prefetch = (flags & MST_SHM_PREFETCH) != 0 || async_release;
    // every capture process fills its segment at the same time (a camera whose next frame was asked for by the previous call's
    // prefetch has been filling it since that camera's copy left the segment)
    for (int i = 0; i < n; ++i) {
        if (!s[i]->requested) { (void)post_request(s[i]); s[i]->requested = true; }
        if (!s[i]->copied && hipEventCreateWithFlags(&s[i]->copied, hipEventDisableTiming) != hipSuccess) return MST_ERR_HIP;
        register_segment(s[i]);
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    auto elapsed_ms = [&] {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        return static_cast<long>(t.tv_sec - t0.tv_sec) * 1000L + (t.tv_nsec - t0.tv_nsec) / 1000000L;
    };
    enum { kWaiting, kCopying, kDone };
    int state[MST_MAX_CAMERAS];
    for (int i = 0; i < n; ++i) state[i] = kWaiting;
    int first = MST_OK, open_cams = n;
    auto fail = [&](int i, int rc) { if (first == MST_OK) first = rc; state[i] = kDone; --open_cams; };
    // Frames are copied in the order they ARRIVE (one hipMemcpyAsync per camera as its packet lands), and a camera's segment is handed
    // back to its capture process -- with MST_SHM_PREFETCH: by asking for its next frame -- the moment ITS copy has left it, not when the
    // whole set has: the capture processes refill while the other cameras' copies are still on the bus.
    for (unsigned spins = 0; open_cams > 0; ++spins) {
        bool progress = false;
        for (int i = 0; i < n; ++i) {
            if (state[i] == kWaiting && sem_trywait(&s[i]->ctl()->server_wrote) == 0) {
                progress = true;
                s[i]->requested = false;
                const uint8_t* q = s[i]->data();
                std::memcpy(&hdr[i], q, sizeof(hdr[i]));
                hdr[i].filePath[sizeof(hdr[i].filePath) - 1] = 0;
                const size_t avail = s[i]->bytes - s[i]->ctl()->data_offset;
                size_t row = 0, rows = 0;
                if (hdr[i].width <= 0 || hdr[i].height <= 0 || hdr[i].flag < 0 || hdr[i].buferPointer < 0 ||
                    static_cast<size_t>(hdr[i].buferPointer) + 1 + static_cast<size_t>(hdr[i].flag) > avail || mst_shm_packet_format(&hdr[i]) < 0) {
                    fail(i, MST_ERR_INVALID_ARG);
                    continue;
                }
                packet_rows(hdr[i], &row, &rows);
                if (dst_stride[i] < row || dst_capacity[i] < dst_stride[i] * (rows - 1) + row) { fail(i, MST_ERR_INVALID_ARG); continue; }
                if (queue_copy(d_dst[i], dst_stride[i], q + hdr[i].buferPointer + 1, row, rows, st) != hipSuccess) { fail(i, MST_ERR_HIP); continue; }
                if (async_release) {
                    // the camera is asked for its next frame by a host function that runs in stream order behind the copy: this
                    // call does not wait for the bus (the request WILL be posted: the next call finds it outstanding)
                    if (hipLaunchHostFunc(st, release_segment, s[i]) != hipSuccess) { fail(i, MST_ERR_HIP); continue; }
                    s[i]->requested = true;
                    state[i] = kDone;
                    --open_cams;
                    continue;
                }
                if (hipEventRecord(s[i]->copied, st) != hipSuccess) { fail(i, MST_ERR_HIP); continue; }
                state[i] = kCopying;
            } else if (state[i] == kCopying) {
                const hipError_t e = hipEventQuery(s[i]->copied);
                if (e == hipErrorNotReady) continue;
                progress = true;
                if (e != hipSuccess) { (void)hipGetLastError(); fail(i, MST_ERR_HIP); continue; }
                state[i] = kDone;
                --open_cams;
                if (prefetch) { (void)post_request(s[i]); s[i]->requested = true; }
            }
        }
        if (progress) { spins = 0; continue; }
        if (timeout_ms >= 0 && (spins & 63u) == 63u && elapsed_ms() > timeout_ms) {
            // withdraw the requests the servers have not taken; one that was taken is on its way and has to be consumed (a stale
            // "server wrote" would let the next request overlap with the server's next write)
            for (int i = 0; i < n; ++i) {
                if (state[i] != kWaiting) continue;
                if (sem_trywait(&s[i]->ctl()->client_ready) != 0) (void)wait_sem(&s[i]->ctl()->server_wrote, -1);
                s[i]->requested = false;
                fail(i, MST_ERR_TIMEOUT);
            }
        }
        if (spins > 2000u) sched_yield();
    }
    return first;
}

int mst_shm_get_device_multi(mst_shm* const* s, int n, mst_frame_packet* hdr, void* const* d_dst, const size_t* dst_stride,
                             const size_t* dst_capacity, void* stream, int timeout_ms)
{
    return mst_shm_get_device_multi_ex(s, n, hdr, d_dst, dst_stride, dst_capacity, stream, timeout_ms, 0);
}

void mst_shm_close(mst_shm* s)
{
    if (!s) return;
    if (s->copied) (void)hipEventDestroy(s->copied);
    if (s->base) {
        // (a prefetch request the server has not taken yet is withdrawn; one it has taken leaves a frame nobody reads: harmless)
        // (one it has taken is consumed here, so that the next client to attach does not find a stale "server wrote")
        if (!s->server && s->requested && sem_trywait(&s->ctl()->client_ready) != 0) (void)wait_sem(&s->ctl()->server_wrote, 1000);
        if (s->registered) (void)hipHostUnregister(s->base);
        if (s->server) {
            sem_destroy(&s->ctl()->client_ready);
            sem_destroy(&s->ctl()->server_wrote);
        }
        munmap(s->base, s->bytes);
    }
    if (s->fd >= 0) close(s->fd);
    if (s->server) shm_unlink(s->name.c_str());
    delete s;
}

}  // extern "C"
