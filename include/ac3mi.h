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
/* device-to-device copy on the context's stream (asynchronous like the batch calls): e.g. stream state back to its
 * initial values between benchmark passes */
int ac3mi_memcpy_d2d(ac3mi_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
/* All engine calls are asynchronous on the context's own HIP stream. */
int ac3mi_sync(ac3mi_ctx *ctx);

/* HIP-event stopwatch on the engine's stream (bench.py times kernels with it:
 * an event on any other stream would not see these launches). */
int ac3mi_timer_start(ac3mi_ctx *ctx);
int ac3mi_timer_stop(ac3mi_ctx *ctx, float *elapsed_ms);   /* synchronises */

/* Measurement aid: 10^9 plain 32-bit VALU instructions per second one SIMD sustains with every SIMD of the device busy
 * doing the same (a gfx950 SIMD issues a wave64 VALU instruction in 2 cycles).  DESIGN.md prices the decode / encode
 * kernels against it. */
int ac3mi_probe_valu_rate(ac3mi_ctx *ctx, double *ginst_per_s_per_simd);
/* the same for scalar (SALU) instructions: the scalar unit issues about half as fast as a SIMD's vector pipe, so for the
 * front ends and the packer - a third or more of whose instructions are scalar - it is the tighter of the two ceilings */
int ac3mi_probe_salu_rate(ac3mi_ctx *ctx, double *ginst_per_s_per_simd);
/* both at once: every wavefront issues three vector instructions per scalar one (the mix of the encoder and the decode front
 * end), `waves_per_simd` (1..8) wavefronts per SIMD: the two rates it sustains together.  If they added up to the rates above the
 * two pipes would overlap perfectly; DESIGN.md 4.3 prices the integer kernels with what this measures. */
int ac3mi_probe_mixed_rate(ac3mi_ctx *ctx, int waves_per_simd, double *valu_ginst_per_s_per_simd, double *salu_ginst_per_s_per_simd);
/* Measurement aid: the rate (read + written GB/s) of a bare float4 copy of `bytes` bytes, one element per lane - the copy
 * MI355X_MICROARCH.md quotes for this part; bench.py reports the transform's rate next to it.  Allocates 2 x bytes. */
int ac3mi_probe_copy_rate(ac3mi_ctx *ctx, size_t bytes, double *gbytes_per_s);

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

/* Optional state-slot indirection (new, for hosts that multiplex many live streams over a pool of state
 * slots: include/ac3mi_stream.h).  While d_slots is non-NULL, stream s of every following ac3mi_imdct_batch /
 * ac3mi_decode_batch / ac3mi_encode_batch call keeps its carry-over state in slot d_slots[s] (device array of
 * n_streams int32) of the state arrays passed to that call, with fixed slot strides: d_delay 6*128 floats,
 * d_lfsr 1, d_last 6*256 samples, d_csnroffst 1.  Pass NULL to return to "stream s uses entry s". */
int ac3mi_set_state_slots(ac3mi_ctx *ctx, const int32_t *d_slots);

/* Optional second overlap state for liba52's exact behaviour around frames whose surround mix level is 0 (new).
 * liba52 keeps one overlap plane per CODED channel and, when a frame's surmixlev is "no surround", leaves the surround
 * channels out of transform and mix altogether (a52dec-0.7.5-cvs/liba52/parse.c:900-913, downmix.c:494-583: the slev == 0
 * cases of MONO, STEREO and 3F outputs): their overlap tails are then dropped, or wait in their planes until the level
 * comes back, depending on which of a52_block's two synthesis paths runs (parse.c:884-937).  The engine keeps one overlap
 * tail per OUTPUT channel (d_delay), mixed; with this state it holds the surround channels' share apart where liba52's
 * bookkeeping can make a difference and reproduces it.  d_pending: laid out and indexed exactly like d_delay
 * ([n_streams][n_out][128] floats, or slot strides of 6*128 under ac3mi_set_state_slots); d_flags: 6 int32 per stream
 * (or slot).  Both zero-initialised for a new stream (as after a52_init).  Used by the following ac3mi_decode_batch /
 * ac3mi_decode_s16_batch / ac3mi_transcode_batch calls whose request mixes surround channels into MONO, STEREO or 3F;
 * ignored otherwise.  Pass NULL, NULL (the default) for a plain linear mix: identical output unless a stream CHANGES its
 * surround mix level to or from "no surround" between two frames, and then different only in the 256 samples per channel
 * that follow the change - and, at a non-zero bias, in the blocks of such a frame where liba52 forgets to add the bias to the
 * left and right outputs (2/1, 2/2 to STEREO and 3/1, 3/2 to 3F with different block sizes in one block: downmix.c:530-583).
 * The byte-stream layer and the a52_* drop-in always use it. */
int ac3mi_set_mix_state(ac3mi_ctx *ctx, float *d_pending, int32_t *d_flags);

/* How ac3mi_decode_batch / ac3mi_decode_s16_batch / ac3mi_transcode_batch spread the work over the GPU (new):
 *   1  one wavefront per stream walks its frames in order (the dither generator's state carries from frame to frame) and
 *      does everything up to the coefficient planes, which go through HBM to the transform kernel: the front end of
 *      rounds 1-2, no longer chosen by 0 - kept as the reference the other variants are compared with bit for bit, and
 *      for A/B runs;
 *   3  one 512-thread workgroup per stream: a wavefront per channel beside a parser and a transformer wavefront, the
 *      coefficient planes stay in LDS and the transform is fused in (a third of the latency of variant 1 per frame,
 *      no plane traffic in HBM; ahead for batches of up to 512 streams);
 *   4  the split front end, parse kernel per stream: one wavefront per stream parses side information, decodes
 *      exponents and allocates bits, frames in order, and only COUNTS each block's mantissas (from per-row totals); it
 *      leaves a descriptor per audio block plus the exponent / allocation rows that changed, and a second kernel unpacks
 *      and dequantises every block with a wavefront of its own (six per frame), before the transform kernel;
 *   5  the same with the parse kernel per FRAME and a prefix pass over the frames' dither draws (few, long streams);
 *   6  as 4, and for one-frame streams whose coded planes are the output planes (no downmix) the second kernel also
 *      transforms: the six wavefronts of a frame hand each other their overlap tails through LDS, the coefficient planes
 *      never reach HBM and the transform kernel is not launched (other calls: as 4);
 *   0  (default) choose by batch shape: 3 for up to 512 streams of at most four frames, else 4 - with 6's fused kernel where
 *      the output is s16 (ac3mi_decode_s16_batch, ac3mi_transcode_batch; to float the two kernels are faster) - or 5 for
 *      fewer than 5 120 streams of more than one frame.
 * (2, a one-kernel front end per frame, was retired in round 4: AC3MI_ERR_ARG.)
 * Conforming streams decode to the same bits in every variant: block 0 of a frame re-sends exponents, coupling and
 * bit-allocation parameters, so only the dither generator's state and the overlap tails carry from frame to frame, and
 * the fused and the separate transform execute the same arithmetic.  A frame whose block 0 reuses state it did not send
 * (damaged or non-conforming) gets status bit AC3MI_STATUS_REUSE0 (0x200): variants 1, 3 and 4 then continue from what the
 * previous frame of the call left behind (as liba52 does), variant 5 from zeros - the result depends on the batch shape. */
#define AC3MI_STATUS_REUSE0 0x200u
int ac3mi_set_decode_mode(ac3mi_ctx *ctx, int mode);

/* How ac3mi_encode_batch / ac3mi_transcode_batch pack a frame once its SNR offsets are found (new; same bytes either way -
 * the searches always run first, one wavefront per stream, frames in order):
 *   1  one wavefront per frame packs it;
 *   2  a workgroup of six wavefronts per frame, one per audio block: every block's first bit follows from bit counts, so
 *      the six pack at once into the frame they share in LDS (a third less latency per frame; ahead for batches of up to
 *      about 1 000 frames);
 *   0  (default) 2 for up to 1 024 frames per call, else 1. */
int ac3mi_set_encode_mode(ac3mi_ctx *ctx, int mode);

/* Workspace bound (new; results do not depend on it, except for frames flagged AC3MI_STATUS_REUSE0 at a tile boundary).  ac3mi_decode_batch, ac3mi_encode_batch and ac3mi_transcode_batch keep
 * their intermediates (coefficient planes, MDCT coefficients, exponents, PCM between decoder and encoder: 37 / 60 /
 * 102 - 139 KB per 5.1 frame) in workspaces owned by the context.  A batch of more than `frames` frames goes through in tiles
 * of whole streams of at most that many frames each (at least one stream), one after the other on the context's
 * streams, so the workspaces stop growing with the batch: a million-stream call needs its own input, output and state
 * arrays plus a fixed 13 - 18 GB.  Default 131072; 0 = never tile.  Calls that ask for stage taps are not tiled. */
int ac3mi_set_tile_frames(ac3mi_ctx *ctx, long long frames);

/* Workspace accounting (new).  ac3mi_workspace_bytes: device bytes the context's workspaces hold right now (they only grow,
 * up to the tile bound).  ac3mi_transcode_workspace_plan: what ac3mi_transcode_batch holds for a call - or, above the tile
 * bound, a tile - of `frames` frames in streams of frames_per_stream with n_in coded planes (lfe included), nfchans
 * full-bandwidth channels and n_out output / encoder channels; pure arithmetic on the allocation's own expressions, callable without a context or a GPU (the
 * multi-GPU planner, ac-3-acm-codec_amd/sharding.py, sizes a rank's shard with it). */
size_t ac3mi_workspace_bytes(const ac3mi_ctx *ctx);
size_t ac3mi_transcode_workspace_plan(size_t frames, int frames_per_stream, int n_in, int nfchans, int n_out);

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

/* ---- frame decode: bitstream -> PCM ------------------------------------------- */

/* Replaces, for a batch of independent streams, the reference's decode inner loop
 *   a52_syncinfo -> a52_frame -> [a52_dynrng] -> 6 x (a52_block -> a52_samples)
 * (src/AC3ACM.cpp:1498,1555-1581; a52dec-0.7.5-cvs/src/a52dec.c:270-305), i.e.
 * L52/parse.c:86-940, L52/bit_allocate.c, L52/bitstream.c/.h, L52/downmix.c, L52/imdct.c.
 *
 *  flags   requested output, exactly a52_frame()'s *flags argument
 *          (channel configuration | AC3MI_LFE | AC3MI_ADJUST_LEVEL)
 *  level   a52_frame()'s *level argument;  bias: its bias argument
 *  dynrng  1 = apply the stream's dynamic-range words (liba52's default),
 *          0 = a52_dynrng(state, NULL, NULL).  A dynrng callback cannot run on the GPU.
 *  acmod, lfeon   coded configuration shared by every frame of the batch (what a52_syncinfo()
 *          reports for any one of them); frames that disagree are reported in d_status and
 *          produce silence
 *  frame_bytes    size of the largest frame of the batch (44.1 kHz streams alternate between two
 *          sizes); every frame starts on its frame_stride slot and carries its own size
 */
typedef struct {
    int flags;
    float level;
    float bias;
    int dynrng;
    int acmod;
    int lfeon;
    int frame_bytes;
} ac3mi_decode_desc;

/* optional per-stage outputs for parity tests; any pointer may be NULL */
typedef struct {
    float *d_coef;      /* [S][F][6][n_in][256] dequantised planes handed to the transform */
    uint8_t *d_blksw;   /* [S][F][6][nfchans] */
    uint8_t *d_exp;     /* [S][F][6][7][256] exponents after each block: 0-4 fbw, 5 lfe, 6 coupling */
    int8_t *d_bap;      /* [S][F][6][7][256] bits per mantissa (liba52 convention, L52/bit_allocate.c:49-72) */
    /* dynamic-range words, for hosts that registered an a52_dynrng() callback (L52/parse.c:207-216, 580-597): [S][F][6][2]
     * floats, word 1 only in dual-mono streams.  d_dynrng_out receives the range factor of every word the stream carries
     * (NaN where there is none) as liba52 would hand it to the callback; d_dynrng_in replaces them (NaN = keep). */
    float *d_dynrng_out;
    const float *d_dynrng_in;
} ac3mi_decode_taps;

/* a52_syncinfo (L52/parse.c:86-129), host side: frame size in bytes, 0 if not a frame */
int ac3mi_syncinfo(const uint8_t *buf, int *flags, int *sample_rate, int *bit_rate);

/* output planes and granted output flags for a descriptor (a52_downmix_init's table,
 * L52/downmix.c:37-67); negative when liba52 would refuse the request */
int ac3mi_decode_planes(const ac3mi_decode_desc *desc, int *n_out, int *out_flags);

/* d_frames  [n_streams][frames_per_stream] frames, frame_stride bytes apart (multiple of 4,
 *           >= frame_bytes rounded up to 4); base 4-byte aligned
 * d_delay   [n_streams][n_out][128] overlap tails, read and rewritten (see ac3mi_imdct_batch)
 * d_lfsr    [n_streams] dither generator state (a52_init sets 1, L52/parse.c:75), read and rewritten
 * d_pcm     [n_streams][frames_per_stream][6][n_out][256] float, a52_samples() plane order
 * d_status  [n_streams][frames_per_stream]: bit b (0..5) = a52_block b returned 1 (that block and
 *           the rest of the frame are silence), bit 8 = a52_syncinfo/a52_frame refused the frame,
 *           bit 9 (AC3MI_STATUS_REUSE0) = block 0 reused state the frame did not send, bits 16..23 = output
 *           flags a52_frame granted
 * Across frames only d_delay and d_lfsr matter for a valid stream (block 0 of every AC-3 frame re-sends
 * exponents, coupling and bit-allocation parameters), and only they persist across calls.  Inside one call
 * the serial variants (ac3mi_set_decode_mode 1 and 3) additionally carry exponent / bit-allocation / coupling
 * state from frame to frame as a52_state_s does, which only shows on frames flagged AC3MI_STATUS_REUSE0;
 * the frame-parallel front end (mode 2, chosen automatically for few long streams) starts every frame clean.
 */
int ac3mi_decode_batch(ac3mi_ctx *ctx, const ac3mi_decode_desc *desc, const uint8_t *d_frames,
                       int frame_stride, int n_streams, int frames_per_stream, float *d_delay,
                       uint16_t *d_lfsr, float *d_pcm, uint32_t *d_status,
                       const ac3mi_decode_taps *taps);

/* The same with the reference's PCM converter folded in (new): what the ACM driver's decode loop hands to the client,
 *   a52_frame(level 1, bias 384) -> 6 x (a52_block -> MapTab[..][..][..](a52_samples(), dst, flags))
 * (src/AC3ACM.cpp:1553-1581; converters src/AC3ASM.asm:174-318, saturating as the x64 build's).  desc->level and
 * desc->bias are ignored (1 and 384 are what the converters presuppose).
 * d_pcm16  [n_streams][frames_per_stream][6][256][n_out] s16 (16-byte aligned), channels interleaved in WAVE order
 *          (ac3mi_convert_s16_batch's layout; bit-identical to ac3mi_decode_batch + ac3mi_convert_s16_batch, without
 *          the float planes going through HBM) */
int ac3mi_decode_s16_batch(ac3mi_ctx *ctx, const ac3mi_decode_desc *desc, const uint8_t *d_frames,
                           int frame_stride, int n_streams, int frames_per_stream, float *d_delay,
                           uint16_t *d_lfsr, int16_t *d_pcm16, uint32_t *d_status);

/* ---- float -> s16 conversion --------------------------------------------------- */

/* Replaces the MMX converters of src/AC3ASM.asm (mmx_convert_N_to_N: psubd 0x43C00000 + packssdw,
 * :303-318; C twin a52dec-0.7.5-cvs/libao/convert2s16.c:33-41) for whole batches: samples decoded
 * with bias 384 and level 1 carry their 16-bit value in the low mantissa bits; subtract the constant,
 * saturate, interleave in WAVE channel order (FL FR FC LFE BL BR; per-flags maps AC3ASM.asm:347-350,
 * 501-505, 679-684, 854-858, 1083-1094).
 *   d_planes [n_blocks][n_out][256] float (a52_samples() layout), d_out [n_blocks][256][n_out] s16 */
int ac3mi_convert_s16_batch(ac3mi_ctx *ctx, const float *d_planes, int16_t *d_out, int flags,
                            size_t n_blocks);

/* ---- frame encode: PCM -> bitstream ------------------------------------------- */

/* Replaces, for a batch of independent streams, AC3_encode_init + AC3_encode_frame
 * (ENC/ac3enc.h:6-7; ENC/ac3enc.cpp:1019-1110, 1640-1763) as driven by stream_convert_pcm
 * (src/AC3ACM.cpp:1762).  The descriptor holds AC3_encode_init's three arguments. */
typedef struct {
    int sample_rate;    /* 48000/44100/32000 and their halves and quarters */
    int bit_rate;       /* bits per second, one of the 19 AC-3 rates (shifted for half rates) */
    int channels;       /* 1..6; 6 = 3/2 + LFE */
} ac3mi_encode_desc;

/* optional per-stage outputs for parity tests; any pointer may be NULL
 * (d_bap and d_encoded_exp only together) */
typedef struct {
    int32_t *d_mdct;            /* [S][F][6][nch][256]  mdct_coef   (ENC/ac3enc.cpp:82)  */
    uint8_t *d_exponent;        /* [S][F][6][nch][256]  exponent    (:83), before min-merge */
    int8_t *d_exp_samples;      /* [S][F][6][nch]       exp_samples (:87) */
    uint8_t *d_encoded_exp;     /* [S][F][6][nch][256]  encoded_exp (:85) */
    uint8_t *d_bap;             /* [S][F][6][nch][256]  bap         (:86) */
    uint8_t *d_exp_strategy;    /* [S][F][6][nch]       exp_strategy (:84) */
    int32_t *d_snroffst;        /* [S][F][2]            csnroffst, fsnroffst chosen for the frame */
} ac3mi_encode_taps;

/* AC3_encode_init's return value: frame size in bytes, 0 for an unsupported combination */
int ac3mi_encode_frame_bytes(const ac3mi_encode_desc *desc);

/* the Q15 tables AC3_encode_init builds on the host (fft_init, xcos1/xsin1) and the window */
int ac3mi_encode_tables(int16_t *costab64, int16_t *sintab64, int16_t *xcos128, int16_t *xsin128, int16_t *window256);

/* the encoder's spec tables as the kernels use them, in the reference's own form (src/ac3enc/ac3tab.h:3-171:
 * ac3_window, latab (first 256 entries), hth[50][3], baptab, bndsz, sdecaytab, fdecaytab, sgaintab, dbkneetab, floortab,
 * fgaintab, ac3_freqs, ac3_bitratetab); any pointer may be NULL.  tests/test_oracle_golden.py checks them against
 * tests/golden/ac3tab.npz, which is frozen from the reference's header. */
int ac3mi_encode_spec_tables(int16_t *window256, uint8_t *latab256, uint16_t *hth50x3, uint8_t *baptab64, uint8_t *bndsz50,
                             uint16_t *sdecay4, uint16_t *fdecay4, uint16_t *sgain4, uint16_t *dbknee4, uint16_t *floor8,
                             uint16_t *fgain8, uint16_t *freqs3, uint16_t *bitrate19);

/* d_pcm        [n_streams][frames_per_stream][1536][channels] s16 interleaved (AC3_encode_frame's `samples`)
 * chmap        HOST array, `channels` entries: input slot of coded channel ch (AC3_encode_frame's `chmap`;
 *              the driver passes {0,2,1,4,5,3} for 6-channel WAVE order, src/AC3ACM.cpp:1631-1662)
 * d_last       [n_streams][channels][256] s16: last_samples (ENC/ac3enc.cpp:55), read and rewritten
 * d_csnroffst  [n_streams] int32: the stream's search state, read and rewritten: bits 0-7 the coarse SNR offset the search
 *              starts from (AC3_encode_init sets 40, :1092; each frame stores its result, :969), bits 8-11 the fine SNR
 *              offset of the last frame whose search succeeded (s->fsnroffst[], :970-972; 0 for a new stream, so
 *              initialise the word to 40) - a frame whose search fails repeats both in its header (:930-933, :1752)
 * d_frames     [n_streams][frames_per_stream] frames, frame_stride bytes apart (multiple of 4)
 */
int ac3mi_encode_batch(ac3mi_ctx *ctx, const ac3mi_encode_desc *desc, const int16_t *d_pcm,
                       const uint8_t *chmap, int16_t *d_last, int32_t *d_csnroffst, uint8_t *d_frames,
                       int frame_stride, int n_streams, int frames_per_stream,
                       const ac3mi_encode_taps *taps);

/* ---- transcode: bitstream -> bitstream ----------------------------------------- */

/* BASELINE configs[4]: decode -> s16 -> re-encode for a batch of independent streams in one call (what a host does
 * with stream_convert_ac3 followed by stream_convert_pcm, src/AC3ACM.cpp:1498-1581, 1762).  Equivalent to
 * ac3mi_decode_batch (level 1, bias 384; dec->level / dec->bias are ignored) + ac3mi_convert_s16_batch +
 * ac3mi_encode_batch with the same state arrays, bit for bit; the transform writes the s16 PCM itself, into the
 * engine's workspace (no float PCM in between), and for large batches the HBM-bound transform of one chunk of
 * streams runs on a second stream under the instruction-bound front end and encoder of its neighbours.
 * The decoder's output channel count (ac3mi_decode_planes) must equal enc->channels; chmap as in
 * ac3mi_encode_batch, applied to the WAVE-order s16 frames. */
int ac3mi_transcode_batch(ac3mi_ctx *ctx, const ac3mi_decode_desc *dec, const ac3mi_encode_desc *enc,
                          const uint8_t *d_frames_in, int in_stride, int n_streams, int frames_per_stream,
                          float *d_delay, uint16_t *d_lfsr, const uint8_t *chmap, int16_t *d_last,
                          int32_t *d_csnroffst, uint8_t *d_frames_out, int out_stride, uint32_t *d_status);

#ifdef __cplusplus
}
#endif
#endif
