/* include/ac3mi_dropin.h — the per-stream call surface of the reference, exported by libac3mi.so
 * with identical names, argument meaning and return codes, implemented on the GPU engine
 * (one-frame launches).  A program written against liba52 / ac3enc / AC3ASM links against
 * libac3mi.so instead and keeps its source unchanged.
 *
 *   decoder   a52dec-0.7.5-cvs/include/a52.h:56-65 (sample_t = level_t = float: vc++/config.h:72-79
 *             leaves LIBA52_FIXED / LIBA52_DOUBLE undefined)
 *   encoder   src/ac3enc/ac3enc.h:6-7 (C++ linkage in the reference: both the Itanium-mangled
 *             names _Z15AC3_encode_initiii / _Z16AC3_encode_framePhPsS_ and extern "C" names exist)
 *   converter src/AC3ACM.cpp:87-90 (ConvertProc, MapTab[2][6][6], IsMMX), src/AC3ASM.asm
 *
 * Differences a caller can observe (also listed in INTEGRATION.md):
 *   - a52_init() returns NULL when no GPU is visible (there is no CPU fallback)
 *   - the work of a52_frame() is deferred to the first a52_block(): all six blocks are decoded by one
 *     launch, a52_block() then hands them out; return codes are the same
 *   - a52_dynrng() with a callback: the callback is host code, so the frame is decoded twice - a look-ahead pass
 *     reports the range factor of every dynamic-range word, the callback maps them in stream order, the second
 *     pass uses the mapped values (same samples as liba52).  All calls of a frame happen at its first a52_block()
 *     rather than one inside each a52_block().  a52_dynrng(state, NULL, NULL) works as in liba52
 *   - after a52_block() has returned 1 for a block, the remaining blocks of that frame also return 1
 *     (liba52 would continue parsing from a corrupted position)
 *   - samples: float PCM within 1e-6 RMS of liba52's, not bit-identical; at bias 384 / level 1 (what the MapTab
 *     converters expect, src/AC3ACM.cpp:1555-1561) one float ulp is one 16-bit step, so converted s16 samples are within
 *     ONE step of liba52's, never promised identical.  Exponents, bit allocation and dequantised coefficients are exact,
 *     and AC3_encode_frame's bytes are exact for given samples
 */
#ifndef AC3MI_DROPIN_H
#define AC3MI_DROPIN_H

#include <stdbool.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float sample_t;
typedef float level_t;
typedef struct a52_state_s a52_state_t;

/* the flags word of a52_syncinfo / a52_frame: a52dec-0.7.5-cvs/include/a52.h:40-54 */
#define A52_CHANNEL 0
#define A52_MONO 1
#define A52_STEREO 2
#define A52_3F 3
#define A52_2F1R 4
#define A52_3F1R 5
#define A52_2F2R 6
#define A52_3F2R 7
#define A52_CHANNEL1 8
#define A52_CHANNEL2 9
#define A52_DOLBY 10
#define A52_CHANNEL_MASK 15

#define A52_LFE 16
#define A52_ADJUST_LEVEL 32

a52_state_t *a52_init(uint32_t mm_accel);
sample_t *a52_samples(a52_state_t *state);
int a52_syncinfo(uint8_t *buf, int *flags, int *sample_rate, int *bit_rate);
int a52_frame(a52_state_t *state, uint8_t *buf, int *flags, level_t *level, sample_t bias);
void a52_dynrng(a52_state_t *state, level_t (*call)(level_t, void *), void *data);
int a52_block(a52_state_t *state);
void a52_free(a52_state_t *state);

/* extern "C" aliases of the encoder entry points.  The library also exports them under the C++ names the reference's
 * own translation units link against (src/ac3enc/ac3enc.h:6-7 is included without extern "C" by src/AC3ACM.cpp:60):
 * a C++ host declares `int AC3_encode_init(int freq, int bitrate, int channels); int AC3_encode_frame(unsigned char *dst,
 * short *samples, unsigned char *chmap);` exactly as that header does and links libac3mi.so (tests/dropin_c/enc_host.cpp). */
int ac3mi_AC3_encode_init(int freq, int bitrate, int channels);
int ac3mi_AC3_encode_frame(unsigned char *dst, short *samples, unsigned char *chmap);

/* float (bias 384) -> s16 interleaved WAVE order, 256 samples; src = a52_samples() */
typedef void (*ConvertProc)(const void *src, void *dst, int flags);
extern ConvertProc MapTab[2][6][6];
bool IsMMX(void);                      /* extern "C" bool IsMMX(), src/AC3ACM.cpp:90 */

/* Secondary liba52 entry points (a52dec-0.7.5-cvs/liba52/a52_internal.h:106-120), same argument meaning.
 * a52_imdct_512 / a52_imdct_256: data[256] coefficients in, PCM out in place; delay[256] overlap plane of which liba52
 * reads and writes [0,128) only (imdct.c:276-292), and so does this library.  One call = one single-plane launch of the
 * batched transform kernel: a per-transform hook, not a fast path.  a52_downmix_init / a52_downmix_coeff are the host
 * arithmetic of liba52/downmix.c:34-330, bit-identical floats.  Not provided: a52_bit_allocate (needs liba52's private
 * state struct), a52_downmix / a52_upmix (the engine mixes in the frequency domain inside the transform kernel),
 * a52_bitstream_* (the bit reader lives in LDS). */
void a52_imdct_init(uint32_t mm_accel);
void a52_imdct_512(sample_t *data, sample_t *delay, sample_t bias);
void a52_imdct_256(sample_t *data, sample_t *delay, sample_t bias);
int a52_downmix_init(int input, int flags, level_t *level, level_t clev, level_t slev);
int a52_downmix_coeff(level_t *coeff, int acmod, int output, level_t level, level_t clev, level_t slev);

/* Threading: every call of this header that reaches the GPU is serialised on one process-wide lock (all states share one
 * engine context).  Different a52_state_t may be used from different threads, as with liba52; the encoder is one stream
 * per process, as in the reference (src/ac3enc/ac3enc.cpp:78-87). */

#ifdef __cplusplus
}
#endif
#endif
