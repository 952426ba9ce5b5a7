/* include/ac3mi_stream.h — byte-stream layer of libac3mi.so: the buffering state machine of the
 * reference's ACM driver (ACMDM_STREAM_OPEN / _SIZE / _CONVERT / _CLOSE), restated as plain C over
 * the batched engine.
 *
 * Reference: src/AC3ACM.cpp
 *   stream_open         :1862-2097     stream_close        :2100-2131
 *   stream_size         :2139-2363     frame-size table    :128-149, ac3_framesize :432-488
 *   stream_convert_ac3  :1430-1628     AC-3 bytes -> s16 PCM: resync by one byte, partial frames,
 *                                      blocks left over when the destination is short
 *   stream_convert_pcm  :1665-1798     s16 PCM -> AC-3 bytes: 1536-sample accumulation, encoded
 *                                      bytes left over when the destination is short
 * Names, argument meaning and result codes follow the ACM structures (WAVEFORMATEX,
 * ACMDRVSTREAMHEADER, ACMDRVSTREAMSIZE) without the Win32 types.
 *
 * What is new: ac3mi_stream_convert_many() advances many streams in lock step and decodes /
 * encodes the frames that become ready in ONE batched engine call per round (the reference
 * converts one stream per call on the caller's thread).  Per-stream carry-over state lives in
 * the slots of an ac3mi_pool (device memory; ac3mi_set_state_slots in ac3mi.h).
 *
 * Documented differences from the reference:
 *   - a frame is decoded whole when it is complete; blocks that do not fit the destination are
 *     kept as PCM and handed out by the next call (the reference keeps the frame bytes and runs
 *     a52_block later: same PCM, but it reads the frame from a buffer it may already have
 *     refilled, src/AC3ACM.cpp:1574-1611)
 *   - if liba52 grants fewer channels than the destination format has (mono stream opened as
 *     stereo, or a stream whose acmod disagrees with its WAVEFORMATEX) the reference converts
 *     planes liba52 never wrote; here the missing frames are silence.  1 -> 2 channel opens are
 *     refused (MMSYSERR_NOTSUPPORTED) for that reason
 *   - WAVE_FORMAT_EXTENSIBLE is accepted for PCM with the default channel masks
 *     (src/AC3ACM.cpp:207-239); the reference's AC-3 EXTENSIBLE test can never pass (:303-304)
 */
#ifndef AC3MI_STREAM_H
#define AC3MI_STREAM_H

#include "ac3mi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* WAVEFORMATEX.wFormatTag values the driver knows (src/AC3ACM.cpp:105, mmreg.h) */
#define AC3MI_WAVE_FORMAT_PCM 0x0001
#define AC3MI_WAVE_FORMAT_AC3 0x2000
#define AC3MI_WAVE_FORMAT_EXTENSIBLE 0xFFFE

/* MyData.dwFlags, src/AC3ACM.cpp:75-81 */
#define AC3MI_ACM_MULTICHANNEL 1    /* allow > 2 channel WAVE_FORMAT_PCM output */
#define AC3MI_ACM_DYNAMICRANGE 2    /* apply the stream's dynamic range words */
#define AC3MI_ACM_DOLBYSURROUND 4   /* 2-channel downmix is Dolby Surround compatible */
#define AC3MI_ACM_NOEXTENSIBLE 32   /* refuse WAVE_FORMAT_EXTENSIBLE */

/* result codes of the ACM messages (mmsystem.h / msacm.h values) */
#define AC3MI_MMSYSERR_NOERROR 0
#define AC3MI_MMSYSERR_NOMEM 7
#define AC3MI_MMSYSERR_NOTSUPPORTED 8
#define AC3MI_MMSYSERR_INVALPARAM 11
#define AC3MI_ACMERR_NOTPOSSIBLE 512

#define AC3MI_STREAMCONVERTF_START 0x00000010   /* ACM_STREAMCONVERTF_START */
#define AC3MI_STREAMSIZEF_SOURCE 0              /* ACM_STREAMSIZEF_SOURCE: given src bytes, dst bytes needed */
#define AC3MI_STREAMSIZEF_DESTINATION 1         /* ACM_STREAMSIZEF_DESTINATION: given dst bytes, src bytes accepted */

/* the WAVEFORMATEX fields the driver reads (+ dwChannelMask for EXTENSIBLE) */
typedef struct {
    uint16_t format_tag;
    uint16_t channels;
    uint32_t samples_per_sec;
    uint32_t avg_bytes_per_sec;
    uint16_t block_align;
    uint16_t bits_per_sample;
    uint32_t channel_mask;
} ac3mi_wavefmt;

/* ACMDRVSTREAMHEADER: pbSrc/cbSrcLength/cbSrcLengthUsed, pbDst/cbDstLength/cbDstLengthUsed, fdwConvert */
typedef struct {
    const uint8_t *src;
    uint32_t src_len;
    uint32_t src_used;
    uint8_t *dst;
    uint32_t dst_len;
    uint32_t dst_used;
    uint32_t flags;
} ac3mi_stream_header;

typedef struct ac3mi_pool ac3mi_pool;
typedef struct ac3mi_stream ac3mi_stream;

/* Device state slots and staging for up to `capacity` concurrently open streams. */
ac3mi_pool *ac3mi_pool_create(ac3mi_ctx *ctx, int capacity);
void ac3mi_pool_destroy(ac3mi_pool *pool);

/* ACMDM_STREAM_OPEN (src/AC3ACM.cpp:1862-2097): same acceptance rules and result codes.
 * query != 0 = ACM_STREAMOPENF_QUERY (nothing is allocated, *out untouched).
 * PCM->PCM and AC-3->AC-3 opens succeed with *out = NULL (the driver copies those, :1800-1826). */
int ac3mi_stream_open(ac3mi_pool *pool, const ac3mi_wavefmt *src, const ac3mi_wavefmt *dst,
                      uint32_t driver_flags, int query, ac3mi_stream **out);
int ac3mi_stream_close(ac3mi_stream *stream);

/* ACMDM_STREAM_SIZE (src/AC3ACM.cpp:2139-2363). */
int ac3mi_stream_size(const ac3mi_stream *stream, int query, uint32_t in_bytes, uint32_t *out_bytes);

/* ACMDM_STREAM_CONVERT for one stream (src/AC3ACM.cpp:1430-1628 / 1665-1798). */
int ac3mi_stream_convert(ac3mi_stream *stream, ac3mi_stream_header *hdr);

/* The same for n streams of one pool; streams[i] pairs with hdrs[i].  Every header ends up exactly
 * as n separate ac3mi_stream_convert calls would leave it. */
int ac3mi_stream_convert_many(ac3mi_stream *const *streams, ac3mi_stream_header *const *hdrs, int n);

/* AC-3 frame size in bytes the driver assumes for a format (ac3_framesize, src/AC3ACM.cpp:432-488) */
int ac3mi_stream_framesize(const ac3mi_wavefmt *fmt);

#ifdef __cplusplus
}
#endif
#endif
