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
    int32_t flag;            /* number of pixel bytes that follow (width * height * 3) */
    int32_t buferPointer;    /* pixels start at (start of this packet) + buferPointer + 1 */
} mst_frame_packet;

typedef struct mst_shm mst_shm;

/* server side (capture process, tests): create the segment `name` ("/cam0", ...) with room for frames of up to
 * max_frame_bytes pixel bytes; an existing segment of that name is replaced */
int  mst_shm_serve(const char *name, size_t max_frame_bytes, mst_shm **out);
/* server side: wait until the client asked for a frame (timeout_ms < 0: forever), write header + pixels, signal.
 * bgr rows are `stride` bytes apart and are packed tightly into the segment. */
int  mst_shm_publish(mst_shm *s, const char *file_path, int width, int height, const uint8_t *bgr, size_t stride,
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
/* SharememClose(); the server side also unlinks the segment */
void mst_shm_close(mst_shm *s);

#ifdef __cplusplus
}
#endif
#endif /* MI355_INGEST_H */
