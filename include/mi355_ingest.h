/*
 * mi355_ingest.h -- frame ingest over shared memory, the step in front of the compositing path.
 *
 * The reference's frame loop does not read cameras itself: a capture process ("server", GetFromRtsp.exe) publishes
 * one decoded BGR frame at a time in a named file mapping, and the stitcher ("client") pulls it with
 * Stitching_sharemem::GetMemData (R64/stitching_sharemem.h:9-48, R64/stitching_sharemem.cpp:22-123; one instance per
 * camera, DRV:739-740).  The wire format is the reference's:
 *
 *     struct DatePacket { char filePath[120]; int width; int height; int flag; int buferPointer; };
 *     pixels (CV_8UC3, tightly packed, `flag` bytes) start at  base + buferPointer + 1      (cpp:111-115)
 *
 * and the handshake is the reference's two events: the client raises "client ready" and waits for "server wrote";
 * the server waits for "client ready", writes header + pixels, raises "server wrote" (cpp:101-118).
 *
 * Windows named objects become POSIX ones: shm_open(name) for the file mapping, two process-shared semaphores inside
 * the segment for the events (the named mutex only guards attachment in the reference and has no counterpart here).
 * mst_shm_get_device copies straight from the (page-locked) segment into device memory, so a frame crosses the host
 * once: capture process -> shared memory -> HBM.
 *
 * Every function returns MST_OK (0) or a negative MST_ERR_* code of mi355_stitch.h; MST_ERR_TIMEOUT is specific to
 * this header.  No function throws.
 */
#ifndef MI355_INGEST_H
#define MI355_INGEST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MST_ERR_TIMEOUT (-7)

/* R64/stitching_sharemem.h:40-46 */
typedef struct mst_frame_packet {
    char    filePath[120];
    int32_t width;
    int32_t height;
    int32_t flag;            /* number of pixel bytes that follow (BGR: width * height * 3; NV12: width * height * 3 / 2) */
    int32_t buferPointer;    /* pixels start at (start of this packet) + buferPointer + 1 */
} mst_frame_packet;

typedef struct mst_shm mst_shm;

/* server side (capture process, tests): create the segment `name` ("/cam0", ...) with room for frames of up to
 * max_frame_bytes pixel bytes; an existing segment of that name is replaced */
int  mst_shm_serve(const char *name, size_t max_frame_bytes, mst_shm **out);
/* server side: wait until the client asked for a frame (timeout_ms < 0: forever), write header + pixels, signal.
 * bgr rows are `stride` bytes apart and are packed tightly into the segment. */
/* (Environment, capture side: MST_SHM_STREAM_COPY=1 makes mst_shm_publish* copy into the segment with non-temporal stores -- right when the
 * consumer is the GPU's DMA engine (mst_shm_get_device*, mst_pipeline_submit_shm: no read-for-ownership of 6 MB per frame, a third of the
 * copy's memory traffic), wrong when it copies with the CPU (mst_shm_get then misses on every line).  Default: plain memcpy.) */
int  mst_shm_publish(mst_shm *s, const char *file_path, int width, int height, const uint8_t *bgr, size_t stride,
                     int timeout_ms);

/* Pixel formats of a packet (round 4).  The reference's header has no format field and its server only writes BGR
 * (stitching_sharemem.cpp:112-115: `flag` = width * height * 3 pixel bytes).  The layout stays the reference's; the
 * format is what `flag` already says: width * height * 3 bytes -> BGR, width * height * 3 / 2 bytes -> NV12 (height rows
 * of Y, height / 2 rows of interleaved U V, as decoders deliver it: half the bytes across the host, converted on the
 * device -- MST_FLAG_FRAMES_NV12 of mi355_stitch.h).  A reference client handed an NV12 packet sees a `flag` that does
 * not match its BGR Mat and a server that only speaks BGR keeps working unchanged. */
#define MST_PIXEL_BGR  0
#define MST_PIXEL_NV12 1
/* MST_PIXEL_* of a received header, or a negative MST_ERR_* code when `flag` fits neither format */
int  mst_shm_packet_format(const mst_frame_packet *hdr);
/* server side: publish an NV12 frame (width and height even; yuv = height * 3 / 2 rows, `stride` bytes apart) */
int  mst_shm_publish_nv12(mst_shm *s, const char *file_path, int width, int height, const uint8_t *yuv, size_t stride,
                          int timeout_ms);

/* client side: ShareMemPre() -- attach to the segment the server created */
int  mst_shm_attach(const char *name, mst_shm **out);
/* client side: GetMemData() -- ask for a frame, wait for it, copy it out.  dst rows are dst_stride bytes apart and
 * dst must hold height rows (dst_capacity bytes available); *hdr receives the packet header.  dst may be NULL to
 * fetch the header only (the frame is then consumed without being copied). */
int  mst_shm_get(mst_shm *s, mst_frame_packet *hdr, uint8_t *dst, size_t dst_stride, size_t dst_capacity,
                 int timeout_ms);
/* the same, straight into device memory (d_dst: device pointer; the copy is ordered on `stream` and complete on
 * return, because the server may overwrite the segment as soon as the next frame is requested) */
int  mst_shm_get_device(mst_shm *s, mst_frame_packet *hdr, void *d_dst, size_t dst_stride, size_t dst_capacity,
                        void *stream, int timeout_ms);
/* One frame from each of n cameras straight into device memory (DRV:736-741 pulls the cameras one after the other; here
 * every server is asked first and writes its segment while the others do, then the copies are queued on `stream` as
 * the frames arrive and awaited once).  d_dst[i]: device buffer of camera i with rows dst_stride[i] bytes apart (BGR:
 * width * 3 bytes per row; NV12: width bytes per row, height * 3 / 2 rows) and dst_capacity[i] bytes; hdr[i] receives
 * the header (mst_shm_packet_format tells the format).  On a timeout or a bad packet the remaining cameras' frames
 * are still consumed, and the first error is returned. */
int  mst_shm_get_device_multi(mst_shm *const *s, int n, mst_frame_packet *hdr, void *const *d_dst, const size_t *dst_stride,
                              const size_t *dst_capacity, void *stream, int timeout_ms);
/* The same with flags.  MST_SHM_PREFETCH: the moment a camera's pixels have left its segment (per camera, not per set) the camera is
 * asked for its NEXT frame, so the capture processes refill their segments while the other cameras' copies are still on the bus and
 * the next call finds the packets waiting -- the frame a call delivers was then requested when the previous call consumed that
 * camera (at most one frame period earlier than Stitching_sharemem::GetMemData would have asked, R64/stitching_sharemem.cpp:96-123).
 * Either way a camera's copy is queued when ITS packet lands (arrival order) and every copy is complete on return. */
#define MST_SHM_PREFETCH 1
/* MST_SHM_ASYNC (implies the prefetch): the call returns as soon as every camera's copy is QUEUED on `stream` (it still waits for the
 * packets to land, not for the bus); the copies complete in stream order, and each camera is asked for its next frame by a host function
 * queued behind its copy (hipLaunchHostFunc).  The caller orders its kernels behind `stream` (an event) and must keep the device buffers
 * alive and unread-from-elsewhere until then -- the frame loop then never waits for a transfer, exactly as mst_pipeline_submit does not.
 * Close a client only after `stream` has drained. */
#define MST_SHM_ASYNC    2
int  mst_shm_get_device_multi_ex(mst_shm *const *s, int n, mst_frame_packet *hdr, void *const *d_dst, const size_t *dst_stride,
                                 const size_t *dst_capacity, void *stream, int timeout_ms, int flags);
/* The reference's frame loop in ONE call per frame set (DRV:736-763: GetMemData x N -> StitchingAll -> imshow), as a stage of the
 * engine's host pipeline (mst_pipeline_wait / mst_pipeline_pending of mi355_stitch.h): the frame set is pulled from the N capture
 * processes' segments into the pipeline's device frames (MST_SHM_ASYNC on the pipeline's copy-in stream), composed behind the copies and
 * copied into `pano` (page-locked host memory: mst_host_alloc) -- the call waits for the packets to land, never for a transfer; up to
 * MST_PIPELINE_DEPTH frame sets are in flight.  clients[i] serves camera i; its packets must have the calibration's frame size and the
 * engine's pixel format (MST_FLAG_FRAMES_NV12 or BGR). */
struct mst_engine;
int  mst_pipeline_submit_shm(struct mst_engine *e, mst_shm *const *clients, int num_clients, uint8_t *pano, size_t pano_stride,
                             int timeout_ms);
/* SharememClose(); the server side also unlinks the segment */
void mst_shm_close(mst_shm *s);

#ifdef __cplusplus
}
#endif
#endif /* MI355_INGEST_H */
