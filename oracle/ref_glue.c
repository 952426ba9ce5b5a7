/* oracle/ref_glue.c — TEST INFRASTRUCTURE ONLY.
 *
 * Compiled INTO oracle/_ref/liba52_ref.so next to the reference's own liba52
 * sources (never into the product).  liba52 keeps its per-stream fields in an
 * opaque a52_state_t (a52dec-0.7.5-cvs/liba52/a52_internal.h:35-88); the golden
 * generator wants the per-stage values (exponents, bap, output mode), so these
 * accessors read them out through the reference's own struct definition.
 */
#include <string.h>
#include "a52.h"
#include "a52_internal.h"

/* which: 0..4 = fbw channel, 5 = lfe, 6 = coupling channel */
static expbap_t * pick (a52_state_t * st, int which)
{
    if (which == 5) return &st->lfe_expbap;
    if (which == 6) return &st->cpl_expbap;
    return &st->fbw_expbap[which];
}

void refglue_get_exp (a52_state_t * st, int which, uint8_t * dst)
{
    memcpy (dst, pick (st, which)->exp, 256);
}

void refglue_get_bap (a52_state_t * st, int which, int8_t * dst)
{
    memcpy (dst, pick (st, which)->bap, 256);
}

int refglue_get_endmant (a52_state_t * st, int ch) { return st->endmant[ch]; }
int refglue_get_output (a52_state_t * st) { return st->output; }
int refglue_get_lfsr (a52_state_t * st) { return st->lfsr_state; }
void refglue_set_lfsr (a52_state_t * st, int v) { st->lfsr_state = (uint16_t) v; }
float refglue_get_level (a52_state_t * st) { return st->level; }
float refglue_get_clev (a52_state_t * st) { return st->clev; }
float refglue_get_slev (a52_state_t * st) { return st->slev; }
int refglue_get_csnroffst (a52_state_t * st) { return st->csnroffst; }
int refglue_get_downmixed (a52_state_t * st) { return st->downmixed; }

/* bits consumed so far, measured from 'base' (the pointer given to a52_frame) */
long refglue_bitpos (a52_state_t * st, uint8_t * base)
{
    return ((uint8_t *) st->buffer_start - base) * 8L - (long) st->bits_left;
}

/* C-side loop over the reference's own a52_imdct_512 so that bench.py's cpu_baseline
 * ("kind": "reference") times liba52 itself and not ctypes call overhead.
 * data: n planes of 256 floats (transformed in place), delay: n_chains planes of 256
 * floats; plane k uses delay plane k % n_chains (a chain = one channel of one stream,
 * walked block after block like a52_block does). */
void refglue_imdct512_batch (sample_t * data, sample_t * delay, long n, long n_chains, sample_t bias)
{
    long k;
    for (k = 0; k < n; k++)
	a52_imdct_512 (data + 256 * k, delay + 256 * (k % n_chains), bias);
}

/* Whole-frame decode loop in C (bench.py's CPU rate beside the decode leg): n frames of frame_bytes each, played as one
 * stream; returns the number of a52_frame / a52_block failures.  *sink receives a checksum so the work cannot be
 * optimised away. */
int refglue_decode_frames (uint8_t * buf, int n, int frame_bytes, int flags, float level, float bias, float * sink)
{
    a52_state_t * st = a52_init (0);
    int f, b, i, errs = 0;
    float acc = 0;
    if (!st) return -1;
    for (f = 0; f < n; f++) {
	int fl = flags;
	level_t lv = level;
	if (a52_frame (st, buf + (size_t) f * frame_bytes, &fl, &lv, bias)) { errs++; continue; }
	for (b = 0; b < 6; b++) {
	    if (a52_block (st)) { errs++; break; }
	    for (i = 0; i < 256 * 6; i += 97) acc += a52_samples (st)[i];
	}
    }
    a52_free (st);
    if (sink) *sink = acc;
    return errs;
}
