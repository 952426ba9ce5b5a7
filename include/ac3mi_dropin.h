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
 *   - a52_dynrng() with a callback: the callback cannot run on the GPU; it is ignored and the stream's
 *     own dynamic-range words apply (as if no callback were given).  a52_dynrng(state, NULL, NULL)
 *     works as in liba52
 *   - after a52_block() has returned 1 for a block, the remaining blocks of that frame also return 1
 *     (liba52 would continue parsing from a corrupted position)
 */
#ifndef AC3MI_DROPIN_H
#define AC3MI_DROPIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef float sample_t;
typedef float level_t;
typedef struct a52_state_s a52_state_t;

a52_state_t *a52_init(uint32_t mm_accel);
sample_t *a52_samples(a52_state_t *state);
int a52_syncinfo(uint8_t *buf, int *flags, int *sample_rate, int *bit_rate);
int a52_frame(a52_state_t *state, uint8_t *buf, int *flags, level_t *level, sample_t bias);
void a52_dynrng(a52_state_t *state, level_t (*call)(level_t, void *), void *data);
int a52_block(a52_state_t *state);
void a52_free(a52_state_t *state);

/* extern "C" aliases of the encoder entry points (the C++-mangled ones are exported too) */
int ac3mi_AC3_encode_init(int freq, int bitrate, int channels);
int ac3mi_AC3_encode_frame(unsigned char *dst, short *samples, unsigned char *chmap);

/* float (bias 384) -> s16 interleaved WAVE order, 256 samples; src = a52_samples() */
typedef void (*ConvertProc)(const void *src, void *dst, int flags);
extern ConvertProc MapTab[2][6][6];
int IsMMX(void);

#ifdef __cplusplus
}
#endif
#endif
