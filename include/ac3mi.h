/* include/ac3mi.h — C-ABI of libac3mi.so, the MI355X-native AC-3 block-transform engine.
 *
 * Plain C: pointers and sizes only, no HIP or torch types.  "d_" parameters are
 * device (HBM) addresses on the context's GPU, obtained from ac3mi_dev_alloc()
 * or from any other allocator on that device (e.g. a torch tensor's data_ptr).
 *
 * Two groups of entry points:
 *
 *  (1) the batched engine (ac3mi_*) — new; this is what feeds the GPU.  One call
 *      processes many independent streams; per-stream carry-over state (overlap
 *      tails, dither LFSR, encoder history) is explicit device memory.
 *  (2) the per-stream drop-in surface of the reference (a52_* / AC3_encode_*),
 *      declared in ac3mi_dropin.h and implemented on top of (1).
 *
 * Every function cites the reference interface it stands in for (paths relative
 * to the reference tree; L52 = a52dec-0.7.5-cvs/liba52, ENC = src/ac3enc).
 */
#ifndef AC3MI_H
#define AC3MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* liba52 channel-configuration flags, a52dec-0.7.5-cvs/include/a52.h:40-54 */
#define AC3MI_CHANNEL 0
#define AC3MI_MONO 1
#define AC3MI_STEREO 2
#define AC3MI_3F 3
#define AC3MI_2F1R 4
#define AC3MI_3F1R 5
#define AC3MI_2F2R 6
#define AC3MI_3F2R 7
#define AC3MI_CHANNEL1 8
#define AC3MI_CHANNEL2 9
#define AC3MI_DOLBY 10
#define AC3MI_CHANNEL_MASK 15
#define AC3MI_LFE 16
#define AC3MI_ADJUST_LEVEL 32

#define AC3MI_OK 0
#define AC3MI_ERR_ARG (-1)          /* bad argument (shape, flags, NULL) */
#define AC3MI_ERR_HIP (-2)          /* a HIP call failed; see ac3mi_last_error() */
#define AC3MI_ERR_UNSUPPORTED (-3)

typedef struct ac3mi_ctx ac3mi_ctx;

/* ---- context / device memory ------------------------------------------------ */

/* Binds to HIP device `device`, creates the engine's stream and uploads the
 * transform tables (the job of a52_imdct_init, L52/imdct.c:358-429, and of the
 * table half of AC3_encode_init, ENC/ac3enc.cpp:1094-1104).
 * Returns NULL when no usable GPU is present (message: ac3mi_last_error(NULL));
 * there is no CPU fallback. */
ac3mi_ctx *ac3mi_create(int device);
void ac3mi_destroy(ac3mi_ctx *ctx);
const char *ac3mi_last_error(const ac3mi_ctx *ctx);
int ac3mi_device_count(void);

void *ac3mi_dev_alloc(ac3mi_ctx *ctx, size_t bytes);
void ac3mi_dev_free(ac3mi_ctx *ctx, void *d_ptr);
int ac3mi_memcpy_h2d(ac3mi_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int ac3mi_memcpy_d2h(ac3mi_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
int ac3mi_memset(ac3mi_ctx *ctx, void *d_dst, int byte, size_t bytes);
/* All engine calls are asynchronous on the context's own HIP stream. */
int ac3mi_sync(ac3mi_ctx *ctx);

/* HIP-event stopwatch on the engine's stream (bench.py times kernels with it:
 * an event on any other stream would not see these launches). */
int ac3mi_timer_start(ac3mi_ctx *ctx);
int ac3mi_timer_stop(ac3mi_ctx *ctx, float *elapsed_ms);   /* synchronises */

/* ---- block transform: IMDCT-512/256 + KBD window + overlap-add + downmix ---- */

/* Replaces the synthesis stage of a52_block (L52/parse.c:881-937): a52_imdct_512
 * (L52/imdct.c:258-293), a52_imdct_256 (:295-345), a52_downmix (L52/downmix.c:
 * 480-619) with its frequency-/time-domain paths, the LFE transform
 * (parse.c:867-873) and the `downmixed` overlap bookkeeping (:888-891,923-927).
 *
 *  acmod/lfeon  coded configuration of the input planes
 *  output       liba52 output flags (channel config | AC3MI_LFE), as returned by
 *               a52_frame(); must be a configuration a52_downmix_init() can
 *               grant for this acmod (L52/downmix.c:34-160)
 *  bias         added once to every output sample (L52/imdct.c:121-124)
 */
typedef struct {
    int acmod;
    int lfeon;
    int output;
    float bias;
} ac3mi_xform_desc;

/* Number of input planes (lfeon + fbw channels of acmod) and of output planes
 * for a descriptor; negative on an invalid combination. */
int ac3mi_xform_planes(const ac3mi_xform_desc *desc, int *n_in, int *n_out);

/* d_coeffs  [n_streams][frames_per_stream][6][n_in][256] float — dequantised,
 *           gain-scaled coefficients exactly as a52_block holds them before the
 *           transform (plane order: LFE first when lfeon, then coded channels)
 * d_blksw   NULL (all long blocks) or [n_streams][frames_per_stream][6][nfchans] u8
 * d_delay   [n_streams][n_out][128] float, read and rewritten: the live half of
 *           liba52's per-channel delay plane, kept per OUTPUT
 *           channel (the already-mixed tail; mathematically what planes 6-11 of
 *           a52_state_s hold in `downmixed` state)
 * d_pcm     [n_streams][frames_per_stream][6][n_out][256] float — what
 *           a52_samples() exposes after each a52_block
 */
int ac3mi_imdct_batch(ac3mi_ctx *ctx, const ac3mi_xform_desc *desc,
                      const float *d_coeffs, const uint8_t *d_blksw,
                      float *d_delay, float *d_pcm,
                      int n_streams, int frames_per_stream);

#ifdef __cplusplus
}
#endif
#endif
