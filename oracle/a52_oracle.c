/* oracle/a52_oracle.c — TEST INFRASTRUCTURE ONLY (see orc.h).
 *
 * CPU restatement, written from scratch, of the reference's AC-3 decode path.
 * Every routine names the reference lines whose behaviour it reproduces
 * (paths relative to /root/reference/a52dec-0.7.5-cvs/).  Float arithmetic keeps
 * the reference's operand types and evaluation order so that, compiled with
 * -ffp-contract=off, the output is bit-identical to oracle/_ref/liba52_ref.so
 * (checked in tests/test_oracle_vs_ref.py and pinned by tests/golden/).
 *
 * Structure is our own: a bit cursor over a private padded copy of the frame,
 * table generation instead of literal LUTs, one generic split-radix recursion
 * instead of hand-unrolled ifft8/16/32/64/128, and a block routine split into
 * side-info / exponents / allocation / mantissas / synthesis stages.
 */
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc.h"
#include "orc_spec_tables.h"

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* flag values: include/a52.h:40-54 */
enum { CH_DUAL = 0, CH_MONO = 1, CH_STEREO = 2, CH_3F = 3, CH_2F1R = 4, CH_3F1R = 5,
       CH_2F2R = 6, CH_3F2R = 7, CH_DUAL1 = 8, CH_DUAL2 = 9, CH_DOLBY = 10,
       CH_MASK = 15, F_LFE = 16, F_ADJUST = 32 };

/* liba52/a52_internal.h:90-94 */
#define G_PLUS6DB 2.0
#define G_PLUS3DB 1.4142135623730951
#define G_3DB 0.7071067811865476
#define G_45DB 0.5946035575013605
#define G_6DB 0.5

#define MAX_FRAME 3840

typedef struct {
    uint8_t bai;            /* fsnroffst<<3 | fgaincod */
    uint8_t deltbae;
    int8_t deltba[50];
} chan_ba;

typedef struct {
    uint8_t exp[256];
    int8_t bap[256];
} chan_eb;

struct orc_a52 {
    uint8_t frame[MAX_FRAME + 16];
    uint32_t bitpos;

    int fscod, halfrate, acmod, lfeon;
    float clev, slev;
    int output;
    float level, bias;
    int dynrnge;
    float dynrng;
    float (*dyncall)(float, void *);
    void *dyndata;

    int chincpl, phsflginu, cplstrtmant, cplendmant;
    uint32_t cplbndstrc;
    float cplco[5][18];
    int cplstrtbnd, ncplbnd;
    int rematflg;
    int endmant[5];
    int bai;                /* 11-bit sdcycod..floorcod */
    uint16_t lfsr;
    int csnroffst;
    chan_ba cplba, ba[5], lfeba;
    int cplfleak, cplsleak;
    chan_eb cpl, fbw[5], lfe;

    float *samples;         /* 12 planes of 256: 0-5 output, 6-11 overlap */
    float coef_tap[6][256]; /* test tap: dequantised planes (LFE first if output) before synthesis */
    uint8_t blksw_tap[5];
    float lfe_tap[256];
    int downmixed;
};

static const uint8_t nfchans_of[11] = { 2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2 };

/* ------------------------------------------------------------------ */
/* generated tables                                                    */

static int tables_ready;
static float q3lvl[3], q5lvl[5], q7lvl[8], q11lvl[11], q15lvl[16];
static float sf[25];
static uint16_t lfsr_step8[256];
/* log-addition table, negated (bit_allocate.c:78-101), built from orc_spec_tables.h */
static int8_t la_neg[256];

static float win[256];
static float rt16[3], rt32[7], rt64[15], rt128[31];
static float pre1[128][2], post1[64][2], pre2[64][2], post2[32][2];
static uint8_t order128[128];

/* liba52/tables.h:49 — Q(x) = ROUND(32768.0 * num / den) */
static float qlevel(int num, int den)
{
    double x = 32768.0 * num / den;
    return (float)(int)(x + (x > 0 ? 0.5 : -0.5));
}

/* The split-radix input order (liba52/imdct.c:49-58) is the recursion's own
 * leaf order: a block of n takes the even-stride half first, then the two
 * odd quarters (offsets +s and -s).  Values are doubled, as the reference
 * stores the coefficient index 2m rather than m. */
static void gen_order(uint8_t *dst, int n, int base, int stride)
{
    if (n == 1) { dst[0] = (uint8_t)(2 * (base & 127)); return; }
    if (n == 2) { gen_order(dst, 1, base, 0); gen_order(dst + 1, 1, base + stride, 0); return; }
    gen_order(dst, n / 2, base, stride * 2);
    gen_order(dst + n / 2, n / 4, base + stride, stride * 4);
    gen_order(dst + 3 * n / 4, n / 4, base - stride, stride * 4);
}

static double bessel_i0_series(double x)
{
    /* liba52/imdct.c:347-356: 100-term Horner form of sum (x/ (i^2))^k */
    double b = 1;
    for (int i = 100; i > 0; i--)
        b = b * x / (i * i) + 1;
    return b;
}

static void build_tables(void)
{
    int i, k;
    if (tables_ready) return;

    for (i = 0; i < 3; i++) q3lvl[i] = qlevel(2 * (i - 1), 3);
    for (i = 0; i < 5; i++) q5lvl[i] = qlevel(2 * (i - 2), 5);
    for (i = 0; i < 7; i++) q7lvl[i] = qlevel(2 * (i - 3), 7);
    q7lvl[7] = 0;
    for (i = 0; i < 11; i++) q11lvl[i] = qlevel(2 * (i - 5), 11);
    for (i = 0; i < 15; i++) q15lvl[i] = qlevel(2 * (i - 7), 15);
    q15lvl[15] = 0;

    /* liba52/tables.h:184-210 — 2^-(15+e) */
    for (i = 0; i < 25; i++) sf[i] = ldexpf(1.0f, -(15 + i));

    /* liba52/tables.h:213-246 is the 8-step advance of a 16-bit Galois LFSR
     * with feedback 0xa011, indexed by the byte that is shifted out. */
    lfsr_step8[0] = 0;
    for (i = 1; i < 256; i <<= 1) {
        uint16_t v;
        if (i == 1) v = 0xa011;
        else {
            uint16_t h = lfsr_step8[i >> 1];
            v = (uint16_t)((h << 1) ^ ((h & 0x8000) ? 0xa011 : 0));
        }
        lfsr_step8[i] = v;
        for (k = 1; k < i; k++) lfsr_step8[i + k] = (uint16_t)(v ^ lfsr_step8[k]);
    }

    /* liba52/imdct.c:358-413 */
    {
        double acc = 0, cum[256];
        for (i = 0; i < 256; i++) {
            acc += bessel_i0_series(i * (256 - i) * (5 * M_PI / 256) * (5 * M_PI / 256));
            cum[i] = acc;
        }
        acc++;
        for (i = 0; i < 256; i++) win[i] = (float)sqrt(cum[i] / acc);
    }
    for (i = 0; i < 3; i++) rt16[i] = (float)cos((M_PI / 8) * (i + 1));
    for (i = 0; i < 7; i++) rt32[i] = (float)cos((M_PI / 16) * (i + 1));
    for (i = 0; i < 15; i++) rt64[i] = (float)cos((M_PI / 32) * (i + 1));
    for (i = 0; i < 31; i++) rt128[i] = (float)cos((M_PI / 64) * (i + 1));

    gen_order(order128, 128, 0, 1);
    {
        uint8_t la[256];
        orc_build_logadd(la);
        for (i = 0; i < 256; i++) la_neg[i] = (int8_t)-la[i];
    }

    for (i = 0; i < 128; i++) {
        k = order128[i] / 2 + 64;
        double c = cos((M_PI / 256) * (k - 0.25)), s = sin((M_PI / 256) * (k - 0.25));
        if (i >= 64) { c = -c; s = -s; }
        pre1[i][0] = (float)c;
        pre1[i][1] = (float)s;
    }
    for (i = 0; i < 64; i++) {
        post1[i][0] = (float)cos((M_PI / 256) * (i + 0.5));
        post1[i][1] = (float)sin((M_PI / 256) * (i + 0.5));
        k = order128[i] / 4;
        pre2[i][0] = (float)cos((M_PI / 128) * (k - 0.25));
        pre2[i][1] = (float)sin((M_PI / 128) * (k - 0.25));
    }
    for (i = 0; i < 32; i++) {
        post2[i][0] = (float)cos((M_PI / 128) * (i + 0.5));
        post2[i][1] = (float)sin((M_PI / 128) * (i + 0.5));
    }
    tables_ready = 1;
}

void orc_imdct_tables(float *window256, float *pre1_256, float *post1_128, float *pre2_128, float *post2_64)
{
    build_tables();
    memcpy(window256, win, sizeof win);
    memcpy(pre1_256, pre1, sizeof pre1);
    memcpy(post1_128, post1, sizeof post1);
    memcpy(pre2_128, pre2, sizeof pre2);
    memcpy(post2_64, post2, sizeof post2);
}

/* ------------------------------------------------------------------ */
/* bit cursor — replaces liba52/bitstream.c/.h (MSB-first, same values) */

static inline uint32_t peek32(const orc_a52_t *st)
{
    const uint8_t *p = st->frame + (st->bitpos >> 3);
    uint32_t w = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
    uint32_t sh = st->bitpos & 7;
    return sh ? ((w << sh) | (p[4] >> (8 - sh))) : w;
}

static inline uint32_t ubits(orc_a52_t *st, int n)
{
    uint32_t v;
    if (n == 0) return 0;
    if (st->bitpos + (uint32_t)n > (MAX_FRAME + 8) * 8u) { st->bitpos = (MAX_FRAME + 8) * 8u; return 0; }
    v = peek32(st) >> (32 - n);
    st->bitpos += n;
    return v;
}

static inline int32_t sbits(orc_a52_t *st, int n)
{
    int32_t v;
    if (n == 0) return 0;
    if (st->bitpos + (uint32_t)n > (MAX_FRAME + 8) * 8u) { st->bitpos = (MAX_FRAME + 8) * 8u; return 0; }
    v = ((int32_t)peek32(st)) >> (32 - n);
    st->bitpos += n;
    return v;
}

/* ------------------------------------------------------------------ */
/* lifecycle: liba52/parse.c:54-84, 207-216, 942-946                   */

orc_a52_t *orc_a52_init(void)
{
    orc_a52_t *st = (orc_a52_t *)calloc(1, sizeof *st);
    if (!st) return NULL;
    st->samples = (float *)calloc(256 * 12, sizeof(float));
    if (!st->samples) { free(st); return NULL; }
    st->downmixed = 1;
    st->lfsr = 1;
    build_tables();
    return st;
}

float *orc_a52_samples(orc_a52_t *st) { return st->samples; }

void orc_a52_free(orc_a52_t *st)
{
    if (!st) return;
    free(st->samples);
    free(st);
}

void orc_a52_dynrng(orc_a52_t *st, float (*call)(float, void *), void *data)
{
    st->dynrnge = 0;
    if (call) {
        st->dynrnge = 1;
        st->dyncall = call;
        st->dyndata = data;
    }
}

static chan_eb *tap(orc_a52_t *st, int which)
{
    return which == 5 ? &st->lfe : which == 6 ? &st->cpl : &st->fbw[which];
}
void orc_a52_get_exp(orc_a52_t *st, int which, uint8_t *dst) { memcpy(dst, tap(st, which)->exp, 256); }
void orc_a52_get_bap(orc_a52_t *st, int which, int8_t *dst) { memcpy(dst, tap(st, which)->bap, 256); }
void orc_a52_get_coefs(orc_a52_t *st, float *dst6x256, uint8_t *blksw5)
{
    memcpy(dst6x256, st->coef_tap, sizeof st->coef_tap);
    memcpy(blksw5, st->blksw_tap, 5);
}
int orc_a52_get_lfsr(orc_a52_t *st) { return st->lfsr; }
void orc_a52_set_lfsr(orc_a52_t *st, int v) { st->lfsr = (uint16_t)v; }
int orc_a52_get_output(orc_a52_t *st) { return st->output; }
/* endmant[0..4], cplstrtmant, cplendmant, chincpl of the block just parsed: lets a test find the bins liba52 never
 * writes when a damaged frame moves the coupling region away from a channel that reuses its exponents
 * (parse.c:813-835 fills [0, endmant), the coupling range and [cplendmant, 256): a gap between endmant and
 * cplstrtmant keeps the previous block's PCM, the buffer being transformed in place). */
/* deltbae of the five fbw channels and of the coupling channel after the block just parsed.  liba52's a52_init does not
 * clear its state (malloc, parse.c:59): a damaged frame that says "reuse" (0) or the reserved 3 before any "new" (1)
 * makes it allocate bits from uninitialised deltba[] arrays (bit_allocate.c:  deltba = deltbae == NONE ? NULL : ...);
 * this restatement starts from zeros.  Tests use the getter to leave such frames out. */
void orc_a52_get_deltbae(orc_a52_t *st, int *out6)
{
    int i;
    for (i = 0; i < 5; i++) out6[i] = st->ba[i].deltbae;
    out6[5] = st->cplba.deltbae;
}
void orc_a52_get_layout(orc_a52_t *st, int *out8)
{
    int i;
    for (i = 0; i < 5; i++) out8[i] = st->endmant[i];
    out8[5] = st->cplstrtmant;
    out8[6] = st->cplendmant;
    out8[7] = st->chincpl;
}

/* ------------------------------------------------------------------ */
/* sync + BSI: liba52/parse.c:86-205                                   */

static const uint8_t halfrate_of_bsid[12] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 2, 3 };

int orc_a52_syncinfo(const uint8_t *buf, int *flags, int *sample_rate, int *bit_rate)
{
    static const int kbps[19] = { 32, 40, 48, 56, 64, 80, 96, 112, 128, 160,
                                  192, 224, 256, 320, 384, 448, 512, 576, 640 };
    /* position of the lfeon bit inside byte 6 depends on how many 2-bit mix
     * fields precede it (parse.c:93) */
    static const uint8_t lfebit[8] = { 0x10, 0x10, 0x04, 0x04, 0x04, 0x01, 0x04, 0x01 };
    int half, acmod, code, rate;

    if (buf[0] != 0x0b || buf[1] != 0x77) return 0;
    if (buf[5] >= 0x60) return 0;               /* bsid >= 12 */
    half = halfrate_of_bsid[buf[5] >> 3];
    acmod = buf[6] >> 5;
    *flags = (((buf[6] & 0xf8) == 0x50) ? CH_DOLBY : acmod) | ((buf[6] & lfebit[acmod]) ? F_LFE : 0);
    code = buf[4] & 63;
    if (code >= 38) return 0;
    rate = kbps[code >> 1];
    *bit_rate = (rate * 1000) >> half;
    switch (buf[4] & 0xc0) {
    case 0x00: *sample_rate = 48000 >> half; return 4 * rate;
    case 0x40: *sample_rate = 44100 >> half; return 2 * (320 * rate / 147 + (code & 1));
    case 0x80: *sample_rate = 32000 >> half; return 6 * rate;
    }
    return 0;
}

int orc_downmix_init(int input, int flags, float *level, float clev, float slev);

int orc_a52_frame(orc_a52_t *st, const uint8_t *buf, int *flags, float *level, float bias)
{
    static const float clev_tab[4] = { (float)G_3DB, (float)G_45DB, (float)G_6DB, (float)G_45DB };
    static const float slev_tab[4] = { (float)G_3DB, (float)G_6DB, 0, (float)G_6DB };
    int acmod, twice, n, f, sr, br, i;

    /* private padded copy of the frame (the reference keeps the caller's pointer) */
    n = orc_a52_syncinfo(buf, &f, &sr, &br);
    if (n <= 0 || n > MAX_FRAME) n = MAX_FRAME;
    memcpy(st->frame, buf, n);
    memset(st->frame + n, 0, MAX_FRAME + 16 - n);

    st->fscod = buf[4] >> 6;
    st->halfrate = halfrate_of_bsid[(buf[5] >> 3) < 12 ? (buf[5] >> 3) : 0];
    st->acmod = acmod = buf[6] >> 5;
    st->bitpos = 6 * 8 + 3;

    if (acmod == 2 && ubits(st, 2) == 2) acmod = CH_DOLBY;       /* dsurmod */
    st->clev = st->slev = 0;
    if ((acmod & 1) && acmod != 1) st->clev = clev_tab[ubits(st, 2)];
    if (acmod & 4) st->slev = slev_tab[ubits(st, 2)];
    st->lfeon = ubits(st, 1);

    st->output = orc_downmix_init(acmod, *flags, level, st->clev, st->slev);
    if (st->output < 0) return 1;
    if (st->lfeon && (*flags & F_LFE)) st->output |= F_LFE;
    *flags = st->output;
    st->dynrng = st->level = *level * 2;        /* parse.c:168-169 */
    st->bias = bias;
    st->dynrnge = 1;
    st->dyncall = NULL;
    st->cplba.deltbae = 2;
    for (i = 0; i < 5; i++) st->ba[i].deltbae = 2;

    twice = !acmod;
    do {
        ubits(st, 5);                           /* dialnorm */
        if (ubits(st, 1)) ubits(st, 8);         /* compr */
        if (ubits(st, 1)) ubits(st, 8);         /* langcod */
        if (ubits(st, 1)) ubits(st, 7);         /* mixlevel, roomtyp */
    } while (twice--);
    ubits(st, 2);                               /* copyrightb, origbs */
    if (ubits(st, 1)) ubits(st, 14);            /* timecod1 */
    if (ubits(st, 1)) ubits(st, 14);            /* timecod2 */
    if (ubits(st, 1)) {                         /* addbsi */
        int len = ubits(st, 6);
        do ubits(st, 8); while (len--);
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* downmix: liba52/downmix.c                                           */

#define PAIR(acmod, out) (((out) << 3) + (acmod))

int orc_downmix_init(int input, int flags, float *level, float clev, float slev)
{
    /* row = requested output, column = coded acmod (downmix.c:37-60) */
    static const uint8_t grant[11][8] = {
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_STEREO, CH_STEREO, CH_STEREO, CH_STEREO, CH_STEREO },
        { CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO },
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_STEREO, CH_STEREO, CH_STEREO, CH_STEREO, CH_STEREO },
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_3F, CH_STEREO, CH_3F, CH_STEREO, CH_3F },
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_STEREO, CH_2F1R, CH_2F1R, CH_2F1R, CH_2F1R },
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_STEREO, CH_2F1R, CH_3F1R, CH_2F1R, CH_3F1R },
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_3F, CH_2F2R, CH_2F2R, CH_2F2R, CH_2F2R },
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_3F, CH_2F2R, CH_3F2R, CH_2F2R, CH_3F2R },
        { CH_DUAL1, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO },
        { CH_DUAL2, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO, CH_MONO },
        { CH_DUAL, CH_DOLBY, CH_STEREO, CH_DOLBY, CH_DOLBY, CH_DOLBY, CH_DOLBY, CH_DOLBY }
    };
    int out = flags & CH_MASK;
    float adj;

    if (out > CH_DOLBY) return -1;
    out = grant[out][input & 7];
    /* float-vs-double comparison kept as in downmix.c:68-70 */
    if (out == CH_STEREO && (input == CH_DOLBY || (input == CH_3F && clev == G_3DB)))
        out = CH_DOLBY;

    if (!(flags & F_ADJUST)) return out;

    /* downmix.c:72-157; operand types (int / float / double) as in the reference */
    switch (PAIR(input & 7, out)) {
    case PAIR(CH_3F, CH_MONO):
        adj = G_3DB / (1 + clev);
        break;
    case PAIR(CH_STEREO, CH_MONO):
    case PAIR(CH_2F2R, CH_2F1R):
    case PAIR(CH_3F2R, CH_3F1R):
        adj = G_3DB;
        break;
    case PAIR(CH_3F2R, CH_2F1R):
        if (clev < (G_PLUS3DB - 1)) { adj = G_3DB; break; }
        /* fall through */
    case PAIR(CH_3F, CH_STEREO):
    case PAIR(CH_3F1R, CH_2F1R):
    case PAIR(CH_3F1R, CH_2F2R):
    case PAIR(CH_3F2R, CH_2F2R):
        adj = 1 / (1 + clev);
        break;
    case PAIR(CH_2F1R, CH_MONO):
        adj = G_PLUS3DB / (2 + slev);
        break;
    case PAIR(CH_2F1R, CH_STEREO):
    case PAIR(CH_3F1R, CH_3F):
        adj = 1 / (1 + slev * G_3DB);
        break;
    case PAIR(CH_3F1R, CH_MONO):
        adj = G_3DB / (1 + clev + slev * 0.5);
        break;
    case PAIR(CH_3F1R, CH_STEREO):
        adj = 1 / (1 + clev + slev * G_3DB);
        break;
    case PAIR(CH_2F2R, CH_MONO):
        adj = G_3DB / (1 + slev);
        break;
    case PAIR(CH_2F2R, CH_STEREO):
    case PAIR(CH_3F2R, CH_3F):
        adj = 1 / (1 + slev);
        break;
    case PAIR(CH_3F2R, CH_MONO):
        adj = G_3DB / (1 + clev + slev);
        break;
    case PAIR(CH_3F2R, CH_STEREO):
        adj = 1 / (1 + clev + slev);
        break;
    case PAIR(CH_MONO, CH_DOLBY):
        adj = G_PLUS3DB;
        break;
    case PAIR(CH_3F, CH_DOLBY):
    case PAIR(CH_2F1R, CH_DOLBY):
        adj = 1 / (1 + G_3DB);
        break;
    case PAIR(CH_3F1R, CH_DOLBY):
    case PAIR(CH_2F2R, CH_DOLBY):
        adj = 1 / (1 + 2 * G_3DB);
        break;
    case PAIR(CH_3F2R, CH_DOLBY):
        adj = 1 / (1 + 3 * G_3DB);
        break;
    default:
        return out;
    }
    *level = *level * adj;
    return out;
}

/* per-input-channel gain vector + mask of channels that are summed into
 * another one (downmix.c:162-330) */
int orc_downmix_coeff(float *g, int acmod, int output, float level, float clev, float slev)
{
    float l3 = level * G_3DB;
    int i;

    switch (PAIR(acmod, output & CH_MASK)) {
    case PAIR(CH_DUAL, CH_DUAL):
    case PAIR(CH_MONO, CH_MONO):
    case PAIR(CH_STEREO, CH_STEREO):
    case PAIR(CH_3F, CH_3F):
    case PAIR(CH_2F1R, CH_2F1R):
    case PAIR(CH_3F1R, CH_3F1R):
    case PAIR(CH_2F2R, CH_2F2R):
    case PAIR(CH_3F2R, CH_3F2R):
    case PAIR(CH_STEREO, CH_DOLBY):
        for (i = 0; i < 5; i++) g[i] = level;
        return 0;
    case PAIR(CH_DUAL, CH_MONO):
        g[0] = g[1] = level * G_6DB;
        return 3;
    case PAIR(CH_STEREO, CH_MONO):
        g[0] = g[1] = l3;
        return 3;
    case PAIR(CH_3F, CH_MONO):
        g[0] = g[2] = l3;
        g[1] = (l3 * clev) * G_PLUS6DB;
        return 7;
    case PAIR(CH_2F1R, CH_MONO):
        g[0] = g[1] = l3;
        g[2] = l3 * slev;
        return 7;
    case PAIR(CH_2F2R, CH_MONO):
        g[0] = g[1] = l3;
        g[2] = g[3] = l3 * slev;
        return 15;
    case PAIR(CH_3F1R, CH_MONO):
        g[0] = g[2] = l3;
        g[1] = (l3 * clev) * G_PLUS6DB;
        g[3] = l3 * slev;
        return 15;
    case PAIR(CH_3F2R, CH_MONO):
        g[0] = g[2] = l3;
        g[1] = (l3 * clev) * G_PLUS6DB;
        g[3] = g[4] = l3 * slev;
        return 31;
    case PAIR(CH_MONO, CH_DOLBY):
        g[0] = l3;
        return 0;
    case PAIR(CH_3F, CH_DOLBY):
        g[0] = g[2] = g[3] = g[4] = level;
        g[1] = l3;
        return 7;
    case PAIR(CH_3F, CH_STEREO):
    case PAIR(CH_3F1R, CH_2F1R):
    case PAIR(CH_3F2R, CH_2F2R):
        g[0] = g[2] = g[3] = g[4] = level;
        g[1] = level * clev;
        return 7;
    case PAIR(CH_2F1R, CH_DOLBY):
        g[0] = g[1] = level;
        g[2] = l3;
        return 7;
    case PAIR(CH_2F1R, CH_STEREO):
        g[0] = g[1] = level;
        g[2] = l3 * slev;
        return 7;
    case PAIR(CH_3F1R, CH_DOLBY):
        g[0] = g[2] = level;
        g[1] = g[3] = l3;
        return 15;
    case PAIR(CH_3F1R, CH_STEREO):
        g[0] = g[2] = level;
        g[1] = level * clev;
        g[3] = l3 * slev;
        return 15;
    case PAIR(CH_2F2R, CH_DOLBY):
        g[0] = g[1] = level;
        g[2] = g[3] = l3;
        return 15;
    case PAIR(CH_2F2R, CH_STEREO):
        g[0] = g[1] = level;
        g[2] = g[3] = level * slev;
        return 15;
    case PAIR(CH_3F2R, CH_DOLBY):
        g[0] = g[2] = level;
        g[1] = g[3] = g[4] = l3;
        return 31;
    case PAIR(CH_3F2R, CH_2F1R):
        g[0] = g[2] = level;
        g[1] = level * clev;
        g[3] = g[4] = l3;
        return 31;
    case PAIR(CH_3F2R, CH_STEREO):
        g[0] = g[2] = level;
        g[1] = level * clev;
        g[3] = g[4] = level * slev;
        return 31;
    case PAIR(CH_3F1R, CH_3F):
        g[0] = g[1] = g[2] = level;
        g[3] = l3 * slev;
        return 13;
    case PAIR(CH_3F2R, CH_3F):
        g[0] = g[1] = g[2] = level;
        g[3] = g[4] = level * slev;
        return 29;
    case PAIR(CH_2F2R, CH_2F1R):
        g[0] = g[1] = level;
        g[2] = g[3] = l3;
        return 12;
    case PAIR(CH_3F2R, CH_3F1R):
        g[0] = g[1] = g[2] = level;
        g[3] = g[4] = l3;
        return 24;
    case PAIR(CH_2F1R, CH_2F2R):
        g[0] = g[1] = level;
        g[2] = l3;
        return 0;
    case PAIR(CH_3F1R, CH_2F2R):
        g[0] = g[2] = level;
        g[1] = level * clev;
        g[3] = l3;
        return 7;
    case PAIR(CH_3F1R, CH_3F2R):
        g[0] = g[1] = g[2] = level;
        g[3] = l3;
        return 0;
    case PAIR(CH_DUAL, CH_DUAL1):
        g[0] = level;
        g[1] = 0;
        return 0;
    case PAIR(CH_DUAL, CH_DUAL2):
        g[0] = 0;
        g[1] = level;
        return 0;
    }
    return -1;
}

/* plane helpers; p(k) = plane k of the 256-sample planes starting at s */
#define P(k) (s + 256 * (k))

static void add_into(float *dst, const float *src, float bias)        /* downmix.c:332-338 */
{
    for (int i = 0; i < 256; i++) dst[i] += src[i] + bias;
}

static void fold_centre(float *s, float bias)                          /* downmix.c:366-376 */
{
    for (int i = 0; i < 256; i++) {
        float c = P(1)[i] + bias;
        P(0)[i] += c;
        P(1)[i] = P(2)[i] + c;
    }
}

static void copy_plane(float *dst, const float *src) { memcpy(dst, src, 256 * sizeof(float)); }
static void clear_plane(float *dst) { memset(dst, 0, 256 * sizeof(float)); }

void orc_downmix(float *s, int acmod, int output, float bias, float clev, float slev)
{
    int i;
    (void)clev;
    switch (PAIR(acmod, output & CH_MASK)) {
    case PAIR(CH_DUAL, CH_DUAL2):
        copy_plane(P(0), P(1));
        break;

    case PAIR(CH_2F1R, CH_MONO):
        if (slev != 0) goto three_to_one;
        /* fall through */
    case PAIR(CH_DUAL, CH_MONO):
    case PAIR(CH_STEREO, CH_MONO):
    two_to_one:
        add_into(P(0), P(1), bias);
        break;

    case PAIR(CH_3F1R, CH_MONO):
        if (slev != 0) goto four_to_one;
        /* fall through */
    case PAIR(CH_3F, CH_MONO):
    three_to_one:
        for (i = 0; i < 256; i++) P(0)[i] += (P(1)[i] + P(2)[i]) + bias;
        break;

    case PAIR(CH_2F2R, CH_MONO):
        if (slev == 0) goto two_to_one;
    four_to_one:
        for (i = 0; i < 256; i++) P(0)[i] += (P(1)[i] + P(2)[i] + P(3)[i]) + bias;
        break;

    case PAIR(CH_3F2R, CH_MONO):
        if (slev == 0) goto three_to_one;
        for (i = 0; i < 256; i++) P(0)[i] += (P(1)[i] + P(2)[i] + P(3)[i] + P(4)[i]) + bias;
        break;

    case PAIR(CH_MONO, CH_DOLBY):
        copy_plane(P(1), P(0));
        break;

    case PAIR(CH_3F1R, CH_STEREO):
        if (slev != 0) {
            for (i = 0; i < 256; i++) {                /* downmix.c:403-413 */
                float c = (P(1)[i] + P(3)[i]) + bias;
                P(0)[i] += c;
                P(1)[i] = P(2)[i] + c;
            }
            break;
        }
        /* fall through */
    case PAIR(CH_3F, CH_STEREO):
    case PAIR(CH_3F, CH_DOLBY):
    centre_only:
        fold_centre(s, bias);
        break;

    case PAIR(CH_2F1R, CH_STEREO):
        if (slev == 0) break;
        for (i = 0; i < 256; i++) {                    /* downmix.c:378-388 */
            float c = P(2)[i] + bias;
            P(0)[i] += c;
            P(1)[i] += c;
        }
        break;

    case PAIR(CH_2F1R, CH_DOLBY):
        for (i = 0; i < 256; i++) {                    /* downmix.c:390-401 */
            float sur = P(2)[i];
            P(0)[i] += -sur + bias;
            P(1)[i] += sur + bias;
        }
        break;

    case PAIR(CH_3F1R, CH_DOLBY):
        for (i = 0; i < 256; i++) {                    /* downmix.c:415-427 */
            float c = P(1)[i] + bias, sur = P(3)[i];
            P(0)[i] += c - sur;
            P(1)[i] = P(2)[i] + c + sur;
        }
        break;

    case PAIR(CH_2F2R, CH_STEREO):
        if (slev == 0) break;
        add_into(P(0), P(2), bias);
        add_into(P(1), P(3), bias);
        break;

    case PAIR(CH_2F2R, CH_DOLBY):
        for (i = 0; i < 256; i++) {                    /* downmix.c:429-439 */
            float sur = P(2)[i] + P(3)[i];
            P(0)[i] += -sur + bias;
            P(1)[i] += sur + bias;
        }
        break;

    case PAIR(CH_3F2R, CH_STEREO):
        if (slev == 0) goto centre_only;
        for (i = 0; i < 256; i++) {                    /* downmix.c:441-451 */
            float c = P(1)[i] + bias;
            P(0)[i] += c + P(3)[i];
            P(1)[i] = c + P(2)[i] + P(4)[i];
        }
        break;

    case PAIR(CH_3F2R, CH_DOLBY):
        for (i = 0; i < 256; i++) {                    /* downmix.c:453-465 */
            float c = P(1)[i] + bias, sur = P(3)[i] + P(4)[i];
            P(0)[i] += c - sur;
            P(1)[i] = P(2)[i] + c + sur;
        }
        break;

    case PAIR(CH_3F1R, CH_3F):
        if (slev == 0) break;
        for (i = 0; i < 256; i++) {
            float c = P(3)[i] + bias;
            P(0)[i] += c;
            P(2)[i] += c;
        }
        break;

    case PAIR(CH_3F2R, CH_3F):
        if (slev == 0) break;
        add_into(P(0), P(3), bias);
        add_into(P(2), P(4), bias);
        break;

    case PAIR(CH_3F1R, CH_2F1R):
        fold_centre(s, bias);
        copy_plane(P(2), P(3));
        break;

    case PAIR(CH_2F2R, CH_2F1R):
        add_into(P(2), P(3), bias);
        break;

    case PAIR(CH_3F2R, CH_2F1R):
        fold_centre(s, bias);
        for (i = 0; i < 256; i++) P(2)[i] = (P(3)[i] + P(4)[i]) + bias;   /* downmix.c:467-473 */
        break;

    case PAIR(CH_3F2R, CH_3F1R):
        add_into(P(3), P(4), bias);
        break;

    case PAIR(CH_2F1R, CH_2F2R):
        copy_plane(P(3), P(2));
        break;

    case PAIR(CH_3F1R, CH_2F2R):
        fold_centre(s, bias);
        copy_plane(P(2), P(3));
        break;

    case PAIR(CH_3F2R, CH_2F2R):
        fold_centre(s, bias);
        copy_plane(P(2), P(3));
        copy_plane(P(3), P(4));
        break;

    case PAIR(CH_3F1R, CH_3F2R):
        copy_plane(P(4), P(3));
        break;
    }
}

/* re-expand downmixed overlap planes to coded-channel slots (downmix.c:621-685) */
void orc_upmix(float *s, int acmod, int output)
{
    int clr_from = -1;      /* clear planes [clr_from, 5) first */
    int kind = 0;           /* 1: move R to slot 2 + clear slot 1 (3-front inputs)
                               2: additionally re-open the rear slot */
    switch (PAIR(acmod, output & CH_MASK)) {
    case PAIR(CH_DUAL, CH_DUAL2):
        copy_plane(P(1), P(0));
        return;

    case PAIR(CH_3F2R, CH_MONO): clr_from = 1; break;
    case PAIR(CH_3F1R, CH_MONO):
    case PAIR(CH_2F2R, CH_MONO):
        clear_plane(P(3)); clear_plane(P(2)); clear_plane(P(1));
        return;
    case PAIR(CH_3F, CH_MONO):
    case PAIR(CH_2F1R, CH_MONO):
        clear_plane(P(2)); clear_plane(P(1));
        return;
    case PAIR(CH_DUAL, CH_MONO):
    case PAIR(CH_STEREO, CH_MONO):
        clear_plane(P(1));
        return;

    case PAIR(CH_3F2R, CH_STEREO):
    case PAIR(CH_3F2R, CH_DOLBY):
        clear_plane(P(4));
        /* fall through */
    case PAIR(CH_3F1R, CH_STEREO):
    case PAIR(CH_3F1R, CH_DOLBY):
        clear_plane(P(3));
        /* fall through */
    case PAIR(CH_3F, CH_STEREO):
    case PAIR(CH_3F, CH_DOLBY):
        kind = 1;
        break;

    case PAIR(CH_2F2R, CH_STEREO):
    case PAIR(CH_2F2R, CH_DOLBY):
        clear_plane(P(3));
        /* fall through */
    case PAIR(CH_2F1R, CH_STEREO):
    case PAIR(CH_2F1R, CH_DOLBY):
        clear_plane(P(2));
        return;

    case PAIR(CH_3F2R, CH_3F):
        clear_plane(P(4));
        /* fall through */
    case PAIR(CH_3F1R, CH_3F):
    case PAIR(CH_2F2R, CH_2F1R):
        clear_plane(P(3));
        return;

    case PAIR(CH_3F2R, CH_3F1R):
        clear_plane(P(4));
        return;

    case PAIR(CH_3F2R, CH_2F1R):
        clear_plane(P(4));
        /* fall through */
    case PAIR(CH_3F1R, CH_2F1R):
        kind = 2;
        break;

    case PAIR(CH_3F2R, CH_2F2R):
        copy_plane(P(4), P(3));
        kind = 2;
        break;

    default:
        return;
    }
    if (clr_from == 1) {
        clear_plane(P(4)); clear_plane(P(3)); clear_plane(P(2)); clear_plane(P(1));
        return;
    }
    if (kind == 2) copy_plane(P(3), P(2));
    copy_plane(P(2), P(1));
    clear_plane(P(1));
}

#undef P

/* ------------------------------------------------------------------ */
/* IMDCT: liba52/imdct.c:80-345                                        */

typedef struct { float re, im; } cpx;

/* one split-radix combine step: z[0..q) z[q..2q) z[2q..3q) z[3q..4q) */
static void sr_combine(cpx *z, const float *roots, int q)
{
    for (int t = 0; t < q; t++) {
        cpx *a0 = z + t, *a1 = z + q + t, *a2 = z + 2 * q + t, *a3 = z + 3 * q + t;
        float u_re, u_im, v_re, v_im;     /* rotated a2, a3 */
        if (t == 0) {                      /* imdct.c:142-155: w = 1 */
            u_re = a2->re; u_im = a2->im;
            v_re = a3->re; v_im = a3->im;
        } else if (q == 2) {               /* imdct.c:159-176: wr == wi, factored form */
            float w = rt16[1];
            u_re = (a2->re + a2->im) * w;
            u_im = (a2->im - a2->re) * w;
            v_re = (a3->re - a3->im) * w;
            v_im = (a3->im + a3->re) * w;
        } else {                           /* imdct.c:128-138 with weights :209-210 */
            float wr = roots[t - 1], wi = roots[q - t - 1];
            u_re = wi * a2->im + wr * a2->re;
            u_im = wr * a2->im - wi * a2->re;
            v_im = wi * a3->re + wr * a3->im;
            v_re = wr * a3->re - wi * a3->im;
        }
        {
            float s1 = u_re + v_re, s2 = u_im + v_im, s3 = u_im - v_im, s4 = v_re - u_re;
            a2->re = a0->re - s1; a2->im = a0->im - s2;
            a3->re = a1->re - s3; a3->im = a1->im - s4;
            a0->re += s1; a0->im += s2;
            a1->re += s3; a1->im += s4;
        }
    }
}

static void sr_ifft(cpx *z, int n)
{
    if (n == 2) {                          /* imdct.c:80-90 */
        float r = z[0].re, i = z[0].im;
        z[0].re += z[1].re; z[0].im += z[1].im;
        z[1].re = r - z[1].re; z[1].im = i - z[1].im;
        return;
    }
    if (n == 4) {                          /* imdct.c:92-113 */
        float t1 = z[0].re + z[1].re, t2 = z[3].re + z[2].re;
        float t3 = z[0].im + z[1].im, t4 = z[2].im + z[3].im;
        float t5 = z[0].re - z[1].re, t6 = z[0].im - z[1].im;
        float t7 = z[2].im - z[3].im, t8 = z[3].re - z[2].re;
        z[0].re = t1 + t2; z[0].im = t3 + t4;
        z[2].re = t1 - t2; z[2].im = t3 - t4;
        z[1].re = t5 + t7; z[1].im = t6 + t8;
        z[3].re = t5 - t7; z[3].im = t6 - t8;
        return;
    }
    sr_ifft(z, n / 2);
    sr_ifft(z + n / 2, n / 4);
    sr_ifft(z + 3 * n / 4, n / 4);
    sr_combine(z, n == 128 ? rt128 : n == 64 ? rt64 : n == 32 ? rt32 : rt16, n / 4);
}

void orc_imdct_512(float *x, float *delay, float bias)
{
    cpx z[128];
    int i;
    build_tables();
    for (i = 0; i < 128; i++) {            /* imdct.c:265-270 */
        int k = order128[i];
        float c = pre1[i][0], s = pre1[i][1];
        z[i].re = s * x[255 - k] + c * x[k];
        z[i].im = c * x[255 - k] - s * x[k];
    }
    sr_ifft(z, 128);
    for (i = 0; i < 64; i++) {             /* imdct.c:276-292 */
        float c = post1[i][0], s = post1[i][1];
        float a_re = c * z[i].re + s * z[i].im;
        float a_im = s * z[i].re - c * z[i].im;
        float b_re = s * z[127 - i].re + c * z[127 - i].im;
        float b_im = c * z[127 - i].re - s * z[127 - i].im;
        float wl = win[2 * i], wh = win[255 - 2 * i], d;

        d = delay[2 * i];
        x[255 - 2 * i] = (d * wl + a_re * wh) + bias;
        x[2 * i] = (d * wh - a_re * wl) + bias;
        delay[2 * i] = a_im;

        wl = win[2 * i + 1]; wh = win[254 - 2 * i];
        d = delay[2 * i + 1];
        x[2 * i + 1] = (d * wh + b_re * wl) + bias;
        x[254 - 2 * i] = (d * wl - b_re * wh) + bias;
        delay[2 * i + 1] = b_im;
    }
}

void orc_imdct_256(float *x, float *delay, float bias)
{
    cpx z1[64], z2[64];
    int i;
    build_tables();
    for (i = 0; i < 64; i++) {             /* imdct.c:303-309 */
        int k = order128[i];
        float c = pre2[i][0], s = pre2[i][1];
        z1[i].re = s * x[254 - k] + c * x[k];
        z1[i].im = c * x[254 - k] - s * x[k];
        z2[i].re = s * x[255 - k] + c * x[k + 1];
        z2[i].im = c * x[255 - k] - s * x[k + 1];
    }
    sr_ifft(z1, 64);
    sr_ifft(z2, 64);
    for (i = 0; i < 32; i++) {             /* imdct.c:316-344 */
        float c = post2[i][0], s = post2[i][1];
        float a_re = c * z1[i].re + s * z1[i].im,       a_im = s * z1[i].re - c * z1[i].im;
        float b_re = s * z1[63 - i].re + c * z1[63 - i].im, b_im = c * z1[63 - i].re - s * z1[63 - i].im;
        float c_re = c * z2[i].re + s * z2[i].im,       c_im = s * z2[i].re - c * z2[i].im;
        float d_re = s * z2[63 - i].re + c * z2[63 - i].im, d_im = c * z2[63 - i].re - s * z2[63 - i].im;
        float w1, w2, d;

        w1 = win[2 * i]; w2 = win[255 - 2 * i]; d = delay[2 * i];
        x[255 - 2 * i] = (d * w1 + a_re * w2) + bias;
        x[2 * i] = (d * w2 - a_re * w1) + bias;
        delay[2 * i] = c_im;

        w1 = win[128 + 2 * i]; w2 = win[127 - 2 * i]; d = delay[127 - 2 * i];
        x[128 + 2 * i] = (d * w2 + a_im * w1) + bias;
        x[127 - 2 * i] = (d * w1 - a_im * w2) + bias;
        delay[127 - 2 * i] = c_re;

        w1 = win[2 * i + 1]; w2 = win[254 - 2 * i]; d = delay[2 * i + 1];
        x[254 - 2 * i] = (d * w1 + b_im * w2) + bias;
        x[2 * i + 1] = (d * w2 - b_im * w1) + bias;
        delay[2 * i + 1] = d_re;

        w1 = win[129 + 2 * i]; w2 = win[126 - 2 * i]; d = delay[126 - 2 * i];
        x[129 + 2 * i] = (d * w2 + b_re * w1) + bias;
        x[126 - 2 * i] = (d * w1 - b_re * w2) + bias;
        delay[126 - 2 * i] = d_im;
    }
}

/* ------------------------------------------------------------------ */
/* parametric bit allocation (decoder formulation): bit_allocate.c     */

static const uint16_t hth_tab[3][50] = {    /* hearing threshold, bit_allocate.c:31-47 */
    { 0x730, 0x730, 0x7c0, 0x800, 0x820, 0x840, 0x850, 0x850, 0x860, 0x860,
      0x860, 0x860, 0x860, 0x870, 0x870, 0x870, 0x880, 0x880, 0x890, 0x890,
      0x8a0, 0x8a0, 0x8b0, 0x8b0, 0x8c0, 0x8c0, 0x8d0, 0x8e0, 0x8f0, 0x900,
      0x910, 0x910, 0x910, 0x910, 0x900, 0x8f0, 0x8c0, 0x870, 0x820, 0x7e0,
      0x7a0, 0x770, 0x760, 0x7a0, 0x7c0, 0x7c0, 0x6e0, 0x400, 0x3c0, 0x3c0 },
    { 0x710, 0x710, 0x7a0, 0x7f0, 0x820, 0x830, 0x840, 0x850, 0x850, 0x860,
      0x860, 0x860, 0x860, 0x860, 0x870, 0x870, 0x870, 0x880, 0x880, 0x880,
      0x890, 0x890, 0x8a0, 0x8a0, 0x8b0, 0x8b0, 0x8c0, 0x8c0, 0x8e0, 0x8f0,
      0x900, 0x910, 0x910, 0x910, 0x910, 0x900, 0x8e0, 0x8b0, 0x870, 0x820,
      0x7e0, 0x7b0, 0x760, 0x770, 0x7a0, 0x7c0, 0x780, 0x5d0, 0x3c0, 0x3c0 },
    { 0x680, 0x680, 0x750, 0x7b0, 0x7e0, 0x810, 0x820, 0x830, 0x840, 0x850,
      0x850, 0x850, 0x860, 0x860, 0x860, 0x860, 0x860, 0x860, 0x860, 0x860,
      0x870, 0x870, 0x870, 0x870, 0x880, 0x880, 0x880, 0x890, 0x8a0, 0x8b0,
      0x8c0, 0x8d0, 0x8e0, 0x8f0, 0x900, 0x910, 0x910, 0x910, 0x900, 0x8f0,
      0x8d0, 0x8b0, 0x840, 0x7f0, 0x790, 0x760, 0x7a0, 0x7c0, 0x7b0, 0x720 }
};

/* number of bits (negative = grouped code id) for "address" a = mask + 4*exp,
 * bit_allocate.c:49-72 without its padding (a = -63..0): a <= -64 -> 16, a > 0 -> 0 */
static const int8_t width_tab[64] = {
    16, 16, 16, 16, 16, 16, 16, 16, 16, 14, 14, 14, 14, 14, 14, 14,
    14, 12, 12, 12, 12, 11, 11, 11, 11, 10, 10, 10, 10, 9, 9, 9,
    9, 8, 8, 8, 8, 7, 7, 7, 7, 6, 6, 6, 6, 5, 5, 5,
    5, 4, 4, -3, -3, 3, 3, 3, -2, -2, -1, -1, -1, -1, -1, 0
};
static inline int8_t width_of(int a) { return a <= -64 ? 16 : a >= 0 ? 0 : width_tab[a + 63]; }

static const uint8_t band_end[30] = {       /* first bin after band 20+i, bit_allocate.c:74-76 */
    21, 22, 23, 24, 25, 26, 27, 28, 31, 34, 37, 40, 43, 46, 49, 55, 61, 67, 73, 79,
    85, 97, 109, 121, 133, 157, 181, 205, 229, 253
};


typedef struct {
    int fdecay, fgain, sdecay, sgain, dbknee, floor, snroffset, halfrate;
    const uint16_t *hth;
    const int8_t *deltba;   /* NULL = none */
    int fast, slow;
} ba_ctx;

static inline void leak(ba_ctx *c, int psd)                  /* bit_allocate.c:103-111 */
{
    c->fast += c->fdecay;
    if (c->fast > psd + c->fgain) c->fast = psd + c->fgain;
    c->slow += c->sdecay;
    if (c->slow > psd + c->sgain) c->slow = psd + c->sgain;
}

static inline int finish_mask(const ba_ctx *c, int mask, int psd, int band)   /* :113-122 */
{
    int h = c->hth[band >> c->halfrate];
    if (psd > c->dbknee) mask -= (psd - c->dbknee) >> 2;
    if (mask > h) mask = h;
    mask -= c->snroffset + 128 * (c->deltba ? c->deltba[band] : 0);
    mask = (mask > 0) ? 0 : ((-mask) >> 5);
    return mask - c->floor;
}

static void bit_allocate(orc_a52_t *st, chan_ba *ba, int bndstart, int start, int end,
                         int fastleak, int slowleak, chan_eb *eb)
{
    static const int slowgain[4] = { 0x540, 0x4d8, 0x478, 0x410 };
    static const int dbpb[4] = { 0xc00, 0x500, 0x300, 0x100 };
    static const int floors[8] = { 0x910, 0x950, 0x990, 0x9d0, 0xa10, 0xa90, 0xb10, 0x1400 };
    ba_ctx c;
    const uint8_t *e = eb->exp;
    int8_t *bap = eb->bap;
    int band, bin, psd = 0, mask, fl;

    /* bit_allocate.c:140-158 */
    c.halfrate = st->halfrate;
    c.fdecay = (63 + 20 * ((st->bai >> 7) & 3)) >> c.halfrate;
    c.fgain = 128 + 128 * (ba->bai & 7);
    c.sdecay = (15 + 2 * (st->bai >> 9)) >> c.halfrate;
    c.sgain = slowgain[(st->bai >> 5) & 3];
    c.dbknee = dbpb[(st->bai >> 3) & 3];
    c.hth = hth_tab[st->fscod];
    c.deltba = (ba->deltbae == 2) ? NULL : ba->deltba;
    fl = floors[st->bai & 7];
    c.snroffset = 960 - 64 * st->csnroffst - 4 * (ba->bai >> 3) + fl;
    c.floor = fl >> 5;
    c.fast = fastleak;
    c.slow = slowleak;

    band = bndstart;
    bin = start;
    if (start == 0) {
        /* bands 0..19(+2) are one bin wide: band == bin.  bit_allocate.c:165-228 */
        int lowcomp = 0, last = end - 1;

        do {                                /* :171-184 — no leak yet */
            if (band < last) {
                if (e[band + 1] == e[band] - 2) lowcomp = 384;
                else if (lowcomp && e[band + 1] > e[band]) lowcomp -= 64;
            }
            psd = 128 * e[band];
            mask = finish_mask(&c, psd + c.fgain + lowcomp, psd, band);
            bap[band] = width_of(mask + 4 * e[band]);
            band++;
        } while (band < 3 || (band < 7 && e[band] > e[band - 1]));
        c.fast = psd + c.fgain;
        c.slow = psd + c.sgain;

        while (band < 7) {                  /* :186-199 */
            if (band < last) {
                if (e[band + 1] == e[band] - 2) lowcomp = 384;
                else if (lowcomp && e[band + 1] > e[band]) lowcomp -= 64;
            }
            psd = 128 * e[band];
            leak(&c, psd);
            mask = (c.fast + lowcomp < c.slow) ? c.fast + lowcomp : c.slow;
            mask = finish_mask(&c, mask, psd, band);
            bap[band] = width_of(mask + 4 * e[band]);
            band++;
        }
        if (end == 7) return;               /* lfe: :201-202 */

        do {                                /* :204-216 */
            if (e[band + 1] == e[band] - 2) lowcomp = 320;
            else if (lowcomp && e[band + 1] > e[band]) lowcomp -= 64;
            psd = 128 * e[band];
            leak(&c, psd);
            mask = (c.fast + lowcomp < c.slow) ? c.fast + lowcomp : c.slow;
            mask = finish_mask(&c, mask, psd, band);
            bap[band] = width_of(mask + 4 * e[band]);
            band++;
        } while (band < 20);

        while (lowcomp > 128) {             /* :218-227 */
            lowcomp -= 128;
            psd = 128 * e[band];
            leak(&c, psd);
            mask = (c.fast + lowcomp < c.slow) ? c.fast + lowcomp : c.slow;
            mask = finish_mask(&c, mask, psd, band);
            bap[band] = width_of(mask + 4 * e[band]);
            band++;
        }
        bin = band;
    }

    do {                                    /* banded part: :231-264 */
        int first = bin, stop = band_end[band - 20] < end ? band_end[band - 20] : end;
        psd = 128 * e[bin++];
        while (bin < stop) {
            int next = 128 * e[bin++], d = next - psd;
            switch (d >> 9) {
            case -6: case -5: case -4: case -3: case -2: psd = next; break;
            case -1:
                /* d == -512 would index latab[256], one past the reference's table
                 * (bit_allocate.c:246).  Unreachable for decodable streams: adjacent
                 * exponents differ by at most 2 (|d| <= 256).  Clamp like A/52. */
                psd = next + la_neg[((-d) >> 1) > 255 ? 255 : ((-d) >> 1)];
                break;
            case 0: psd += la_neg[d >> 1]; break;
            }
        }
        leak(&c, psd);
        mask = finish_mask(&c, c.fast < c.slow ? c.fast : c.slow, psd, band);
        band++;
        for (bin = first; bin < stop; bin++) bap[bin] = width_of(mask + 4 * e[bin]);
    } while (bin < end);
}

/* ------------------------------------------------------------------ */
/* exponents, delta bit allocation: parse.c:218-294                    */

static int read_exponents(orc_a52_t *st, int strategy, int ngrps, int absexp, uint8_t *dst)
{
    int rep = 1 << (strategy - 1), e = absexp;
    while (ngrps--) {
        int code = ubits(st, 7), d[3], j, r;
        if (code >= 125) return 1;          /* exp_1[125..127] = 25 forces the >24 error */
        d[0] = code / 25 - 2; d[1] = (code / 5) % 5 - 2; d[2] = code % 5 - 2;
        for (j = 0; j < 3; j++) {
            e += d[j];
            if (e < 0 || e > 24) return 1;  /* uint8 wrap or >24 (parse.c:227-228) */
            for (r = 0; r < rep; r++) *dst++ = (uint8_t)e;
        }
    }
    return 0;
}

static int read_deltba(orc_a52_t *st, int8_t *deltba)
{
    int nseg, band = 0;
    memset(deltba, 0, 50);
    nseg = ubits(st, 3);
    do {
        int len, d;
        band += ubits(st, 5);
        len = ubits(st, 4);
        d = ubits(st, 3);
        d -= (d >= 4) ? 3 : 4;
        if (!len) continue;
        if (band + len >= 50) return 1;
        while (len--) deltba[band++] = (int8_t)d;
    } while (nseg--);
    return 0;
}

/* ------------------------------------------------------------------ */
/* mantissas: parse.c:310-556                                          */

typedef struct {
    float pend3[2]; int n3;     /* 3-level group: values still to hand out */
    float pend5[2]; int n5;
    float pend11;   int n11;
} grp_state;

static inline int16_t dither_draw(orc_a52_t *st)                       /* parse.c:310-319 */
{
    int16_t ns = (int16_t)(lfsr_step8[st->lfsr >> 8] ^ (st->lfsr << 8));
    st->lfsr = (uint16_t)ns;
    return (int16_t)((3 * ns) >> 2);
}

/* one dequantised mantissa (before exponent/gain scaling) for a non-zero bap */
static inline float mantissa(orc_a52_t *st, grp_state *g, int width)
{
    int code;
    switch (width) {
    case -1:
        if (g->n3) return g->pend3[--g->n3];
        code = ubits(st, 5);
        if (code >= 27) { g->pend3[0] = g->pend3[1] = 0; g->n3 = 2; return 0; }
        g->pend3[0] = q3lvl[code % 3]; g->pend3[1] = q3lvl[(code / 3) % 3]; g->n3 = 2;
        return q3lvl[code / 9];
    case -2:
        if (g->n5) return g->pend5[--g->n5];
        code = ubits(st, 7);
        if (code >= 125) { g->pend5[0] = g->pend5[1] = 0; g->n5 = 2; return 0; }
        g->pend5[0] = q5lvl[code % 5]; g->pend5[1] = q5lvl[(code / 5) % 5]; g->n5 = 2;
        return q5lvl[code / 25];
    case 3:
        return q7lvl[ubits(st, 3)];
    case -3:
        if (g->n11) { g->n11 = 0; return g->pend11; }
        code = ubits(st, 7);
        if (code >= 121) { g->pend11 = 0; g->n11 = 1; return 0; }
        g->pend11 = q11lvl[code % 11]; g->n11 = 1;
        return q11lvl[code / 11];
    case 4:
        return q15lvl[ubits(st, 4)];
    default:
        return (float)(sbits(st, width) * (1 << (16 - width)));
    }
}

static void unpack_channel(orc_a52_t *st, float *coef, chan_eb *eb, grp_state *g,
                           float gain, int dither, int end)
{
    float fac[25];
    int i;
    for (i = 0; i <= 24; i++) fac[i] = sf[i] * gain;                  /* parse.c:345-348 */
    for (i = 0; i < end; i++) {
        int w = eb->bap[i];
        if (w == 0) coef[i] = dither ? dither_draw(st) * fac[eb->exp[i]] : 0;
        else coef[i] = mantissa(st, g, w) * fac[eb->exp[i]];
    }
}

static void unpack_coupling(orc_a52_t *st, int nfchans, const float *gain, float *planes,
                            grp_state *g, const uint8_t *dithflag)
{
    uint32_t strc = st->cplbndstrc;
    int bnd = 0, i = st->cplstrtmant, ch;
    float co[5];

    while (i < st->cplendmant) {
        int stop = i + 12;
        while (strc & 1) { strc >>= 1; stop += 12; }
        strc >>= 1;
        for (ch = 0; ch < nfchans; ch++) co[ch] = st->cplco[ch][bnd] * gain[ch];
        bnd++;
        for (; i < stop; i++) {
            int w = st->cpl.bap[i];
            if (w == 0) {                                             /* parse.c:466-481 */
                for (ch = 0; ch < nfchans; ch++)
                    if ((st->chincpl >> ch) & 1)
                        planes[256 * ch + i] = dithflag[ch]
                            ? (sf[st->cpl.exp[i]] * co[ch]) * dither_draw(st) : 0;
            } else {
                float m = mantissa(st, g, w);
                m *= sf[st->cpl.exp[i]];
                for (ch = 0; ch < nfchans; ch++)
                    if ((st->chincpl >> ch) & 1) planes[256 * ch + i] = m * co[ch];
            }
        }
    }
}


/* ------------------------------------------------------------------ */
/* synthesis stage of a block: parse.c:881-937.  `s` points at the first fbw
 * plane (LFE, when output, sits at s-256 and has been transformed already);
 * overlap planes live 1536 floats further on. */
static void synth_stage(float *s, int *downmixed, int acmod, int output, float bias, float clev, float slev,
                        const uint8_t *blksw, const float *gain, int biasmask)
{
    int nf = nfchans_of[acmod], i, j;

    i = 0;
    if (nfchans_of[output & CH_MASK] < nf)
        for (i = 1; i < nf; i++)
            if (blksw[i] != blksw[0]) break;

    if (i < nf) {
        /* path A: block sizes differ -> transform every coded channel, mix in time domain */
        if (*downmixed) {
            *downmixed = 0;
            orc_upmix(s + 1536, acmod, output);
        }
        for (i = 0; i < nf; i++) {
            float b = (biasmask & (1 << i)) ? 0 : bias;
            if (gain[i]) {
                if (blksw[i]) orc_imdct_256(s + 256 * i, s + 1536 + 256 * i, b);
                else orc_imdct_512(s + 256 * i, s + 1536 + 256 * i, b);
            } else
                for (j = 0; j < 256; j++) s[256 * i + j] = b;
        }
        orc_downmix(s, acmod, output, bias, clev, slev);
    } else {
        /* path B: mix coefficients first, transform only the output channels */
        int nout = nfchans_of[output & CH_MASK];
        orc_downmix(s, acmod, output, 0, clev, slev);
        if (!*downmixed) {
            *downmixed = 1;
            orc_downmix(s + 1536, acmod, output, 0, clev, slev);
        }
        for (i = 0; i < nout; i++) {
            if (blksw[0]) orc_imdct_256(s + 256 * i, s + 1536 + 256 * i, bias);
            else orc_imdct_512(s + 256 * i, s + 1536 + 256 * i, bias);
        }
    }
}

/* Transform-only oracle for BASELINE configs 2 and 4: runs the synthesis stage of
 * a52_block (LFE transform parse.c:867-873 + synth_stage) over coefficient planes
 * laid out like the product's ac3mi_imdct_batch: coef [S][F][6][n_in][256] with the
 * LFE plane first when lfeon, blksw NULL or [S][F][6][nfchans], pcm
 * [S][F][6][n_out][256].  state: per stream 12*256 floats (liba52's sample
 * buffer, planes 6-11 = overlap) + downmixed flag, both carried across calls. */
int orc_xform_batch(const float *coef, const uint8_t *blksw, float *state_planes, int *state_downmixed,
                    float *pcm, int n_streams, int frames, int acmod, int lfeon, int output, float bias,
                    float clev, float slev)
{
    static const uint8_t zero_sw[5] = { 0, 0, 0, 0, 0 };
    int nf, nout_f, n_in, n_out, out_lfe, biasmask;
    float gain[5];
    if (acmod < 0 || acmod > 7 || (output & CH_MASK) > CH_DOLBY) return -1;
    build_tables();
    nf = nfchans_of[acmod];
    nout_f = nfchans_of[output & CH_MASK];
    out_lfe = (output & F_LFE) ? 1 : 0;
    n_in = nf + (lfeon ? 1 : 0);
    n_out = nout_f + out_lfe;
    biasmask = orc_downmix_coeff(gain, acmod, output, 1.0f, clev, slev);
    for (int i = 0; i < 5; i++) gain[i] = 1.0f;     /* transform-only: every plane is live */
    for (int st = 0; st < n_streams; st++) {
        float *planes = state_planes + (size_t)st * 3072;
        float *s = planes + (out_lfe ? 256 : 0);
        for (int fb = 0; fb < frames * 6; fb++) {
            const float *c = coef + ((size_t)st * frames * 6 + fb) * n_in * 256;
            const uint8_t *sw = blksw ? blksw + ((size_t)st * frames * 6 + fb) * nf : zero_sw;
            if (lfeon && out_lfe) {
                memcpy(s - 256, c, 256 * sizeof(float));
                orc_imdct_512(s - 256, s + 1536 - 256, bias);
            }
            memcpy(s, c + (lfeon ? 256 : 0), (size_t)nf * 256 * sizeof(float));
            synth_stage(s, &state_downmixed[st], acmod, output, bias, clev, slev, sw, gain, biasmask);
            memcpy(pcm + ((size_t)st * frames * 6 + fb) * n_out * 256, planes, (size_t)n_out * 256 * sizeof(float));
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ */
/* one audio block: parse.c:558-940                                    */

int orc_a52_block(orc_a52_t *st)
{
    static const int remat_edge[4] = { 25, 37, 61, 253 };
    static const uint8_t cpl_bnd0[16] = { 31, 35, 37, 39, 41, 42, 43, 44, 45, 45, 46, 46, 47, 47, 48, 48 };
    int nf = nfchans_of[st->acmod], i, j, twice, realloc = 0;
    uint8_t blksw[5], dithflag[5], chexpstr[5];
    int cplexpstr = 0, lfeexpstr = 0;
    float gain[5], *s;
    int biasmask;
    grp_state grp;

    /* ---- side information (parse.c:572-701) ---- */
    for (i = 0; i < nf; i++) blksw[i] = ubits(st, 1);
    for (i = 0; i < nf; i++) dithflag[i] = ubits(st, 1);

    twice = !st->acmod;
    do {
        if (ubits(st, 1)) {                                           /* dynrnge */
            int code = sbits(st, 8);
            if (st->dynrnge) {
                float range = (((code & 0x1f) | 0x20) << 13) * sf[3 - (code >> 5)];
                if (st->dyncall) range = st->dyncall(range, st->dyndata);
                st->dynrng = st->level * range;
            }
        }
    } while (twice--);

    if (ubits(st, 1)) {                                               /* cplstre */
        st->chincpl = 0;
        if (ubits(st, 1)) {                                           /* cplinu */
            int begf, endf, nsub;
            for (i = 0; i < nf; i++) st->chincpl |= ubits(st, 1) << i;
            if (st->acmod < 2) return 1;
            if (st->acmod == 2) st->phsflginu = ubits(st, 1);
            begf = ubits(st, 4);
            endf = ubits(st, 4);
            if (endf + 3 - begf < 0) return 1;
            st->ncplbnd = nsub = endf + 3 - begf;
            st->cplstrtbnd = cpl_bnd0[begf];
            st->cplstrtmant = begf * 12 + 37;
            st->cplendmant = endf * 12 + 73;
            st->cplbndstrc = 0;
            for (i = 0; i < nsub - 1; i++)
                if (ubits(st, 1)) { st->cplbndstrc |= 1u << i; st->ncplbnd--; }
        }
    }

    if (st->chincpl) {                                                /* coupling coordinates */
        int any = 0;
        for (i = 0; i < nf; i++)
            if ((st->chincpl >> i) & 1)
                if (ubits(st, 1)) {
                    int master = 3 * ubits(st, 2);
                    any = 1;
                    for (j = 0; j < st->ncplbnd; j++) {
                        int ex = ubits(st, 4), ma = ubits(st, 4);
                        ma = (ex == 15) ? (ma << 14) : ((ma | 0x10) << 13);
                        st->cplco[i][j] = ma * sf[ex + master];
                    }
                }
        if (st->acmod == 2 && st->phsflginu && any)
            for (j = 0; j < st->ncplbnd; j++)
                if (ubits(st, 1)) st->cplco[1][j] = -st->cplco[1][j];
    }

    if (st->acmod == 2 && ubits(st, 1)) {                             /* rematstr */
        int end = st->chincpl ? st->cplstrtmant : 253;
        st->rematflg = 0;
        i = 0;
        do st->rematflg |= ubits(st, 1) << i; while (remat_edge[i++] < end);
    }

    if (st->chincpl) cplexpstr = ubits(st, 2);
    for (i = 0; i < nf; i++) chexpstr[i] = ubits(st, 2);
    if (st->lfeon) lfeexpstr = ubits(st, 1);

    for (i = 0; i < nf; i++)
        if (chexpstr[i]) {
            if ((st->chincpl >> i) & 1) st->endmant[i] = st->cplstrtmant;
            else {
                int bw = ubits(st, 6);
                if (bw > 60) return 1;
                st->endmant[i] = bw * 3 + 73;
            }
        }

    /* ---- exponents (parse.c:703-736) ---- */
    if (cplexpstr) {
        int ngrp = (st->cplendmant - st->cplstrtmant) / (3 << (cplexpstr - 1));
        int e0 = ubits(st, 4) << 1;
        realloc = 64;
        if (read_exponents(st, cplexpstr, ngrp, e0, st->cpl.exp + st->cplstrtmant)) return 1;
    }
    for (i = 0; i < nf; i++)
        if (chexpstr[i]) {
            int gs = 3 << (chexpstr[i] - 1), ngrp = (st->endmant[i] + gs - 4) / gs;
            realloc |= 1 << i;
            st->fbw[i].exp[0] = ubits(st, 4);
            if (read_exponents(st, chexpstr[i], ngrp, st->fbw[i].exp[0], st->fbw[i].exp + 1)) return 1;
            ubits(st, 2);                                             /* gainrng */
        }
    if (lfeexpstr) {
        realloc |= 32;
        st->lfe.exp[0] = ubits(st, 4);
        if (read_exponents(st, lfeexpstr, 2, st->lfe.exp[0], st->lfe.exp + 1)) return 1;
    }

    /* ---- bit-allocation parameters (parse.c:738-772) ---- */
    if (ubits(st, 1)) { realloc = 127; st->bai = ubits(st, 11); }
    if (ubits(st, 1)) {
        realloc = 127;
        st->csnroffst = ubits(st, 6);
        if (st->chincpl) st->cplba.bai = ubits(st, 7);
        for (i = 0; i < nf; i++) st->ba[i].bai = ubits(st, 7);
        if (st->lfeon) st->lfeba.bai = ubits(st, 7);
    }
    if (st->chincpl && ubits(st, 1)) {
        realloc |= 64;
        st->cplfleak = 9 - ubits(st, 3);
        st->cplsleak = 9 - ubits(st, 3);
    }
    if (ubits(st, 1)) {                                               /* deltbaie */
        realloc = 127;
        if (st->chincpl) st->cplba.deltbae = ubits(st, 2);
        for (i = 0; i < nf; i++) st->ba[i].deltbae = ubits(st, 2);
        if (st->chincpl && st->cplba.deltbae == 1 && read_deltba(st, st->cplba.deltba)) return 1;
        for (i = 0; i < nf; i++)
            if (st->ba[i].deltbae == 1 && read_deltba(st, st->ba[i].deltba)) return 1;
    }

    /* ---- allocation (parse.c:774-798) ---- */
    if (realloc) {
        int allzero = !st->csnroffst && !(st->chincpl && (st->cplba.bai >> 3)) &&
                      !(st->lfeon && (st->lfeba.bai >> 3));
        for (i = 0; allzero && i < nf; i++)
            if (st->ba[i].bai >> 3) allzero = 0;
        if (allzero) {
            memset(st->cpl.bap, 0, 256);
            for (i = 0; i < nf; i++) memset(st->fbw[i].bap, 0, 256);
            memset(st->lfe.bap, 0, 256);
        } else {
            if (st->chincpl && (realloc & 64))
                bit_allocate(st, &st->cplba, st->cplstrtbnd, st->cplstrtmant, st->cplendmant,
                             st->cplfleak << 8, st->cplsleak << 8, &st->cpl);
            for (i = 0; i < nf; i++)
                if (realloc & (1 << i))
                    bit_allocate(st, &st->ba[i], 0, 0, st->endmant[i], 0, 0, &st->fbw[i]);
            if (st->lfeon && (realloc & 32)) {
                st->lfeba.deltbae = 2;
                bit_allocate(st, &st->lfeba, 0, 0, 7, 0, 0, &st->lfe);
            }
        }
    }

    if (ubits(st, 1)) {                                               /* skip field */
        int n = ubits(st, 9);
        while (n--) ubits(st, 8);
    }

    /* ---- mantissas (parse.c:806-879) ---- */
    s = st->samples;
    if (st->output & F_LFE) s += 256;
    biasmask = orc_downmix_coeff(gain, st->acmod, st->output, st->dynrng, st->clev, st->slev);

    memset(&grp, 0, sizeof grp);
    {
        int cpl_done = 0;
        for (i = 0; i < nf; i++) {
            unpack_channel(st, s + 256 * i, &st->fbw[i], &grp, gain[i], dithflag[i], st->endmant[i]);
            if ((st->chincpl >> i) & 1) {
                if (!cpl_done) {
                    cpl_done = 1;
                    unpack_coupling(st, nf, gain, s, &grp, dithflag);
                }
                j = st->cplendmant;
            } else
                j = st->endmant[i];
            do s[256 * i + j] = 0; while (++j < 256);
        }
    }

    if (st->acmod == 2) {                                             /* rematrix: parse.c:837-865 */
        int end = st->endmant[0] < st->endmant[1] ? st->endmant[0] : st->endmant[1];
        int flg = st->rematflg, band;
        i = 0;
        j = 13;
        do {
            if (!(flg & 1)) { flg >>= 1; j = remat_edge[i++]; continue; }
            flg >>= 1;
            band = remat_edge[i++];
            if (band > end) band = end;
            do {
                float a = s[j], b = s[256 + j];
                s[j] = a + b;
                s[256 + j] = a - b;
            } while (++j < band);
        } while (j < end);
    }

    if (st->lfeon) {
        if (st->output & F_LFE) {
            unpack_channel(st, s - 256, &st->lfe, &grp, st->dynrng, 0, 7);
            for (i = 7; i < 256; i++) (s - 256)[i] = 0;
            memcpy(st->lfe_tap, s - 256, sizeof st->lfe_tap);
            orc_imdct_512(s - 256, s + 1536 - 256, st->bias);
        } else
            unpack_channel(st, s + 1280, &st->lfe, &grp, 0, 0, 7);
    }

    /* test taps (not in the reference): coefficient planes as the transform stage sees them */
    memcpy(st->coef_tap, st->samples, sizeof st->coef_tap);
    memcpy(st->blksw_tap, blksw, 5);
    if (st->lfeon && (st->output & F_LFE)) memcpy(st->coef_tap[0], st->lfe_tap, sizeof st->lfe_tap);

    /* ---- synthesis (parse.c:881-937) ---- */
    synth_stage(s, &st->downmixed, st->acmod, st->output, st->bias, st->clev, st->slev, blksw, gain, biasmask);
    return 0;
}

/* ------------------------------------------------------------------ */
/* float (bias 384) -> s16, interleaved in WAVE order.
 * Behaviour of src/AC3ASM.asm mmx_convert_N_to_N (psubd 0x43C00000 +
 * packssdw: signed saturation, :303-318) = libao/convert2s16.c:33-41;
 * channel order per flags as in AC3ASM.asm:347-350,501-505,679-684,854-858,
 * 1083-1094 (liba52 plane order LFE,L,C,R,SL,SR -> FL,FR,FC,LFE,BL,BR). */

static inline int16_t to_s16(float f)
{
    uint32_t u;
    int32_t i;
    memcpy(&u, &f, 4);
    i = (int32_t)(u - 0x43c00000u);         /* psubd wraps modulo 2^32 */
    return (int16_t)(i > 32767 ? 32767 : i < -32768 ? -32768 : i);
}

void orc_convert_s16(const float *planes, int16_t *dst, int flags)
{
    int map[6], n = 0, lfe = (flags & F_LFE) ? 1 : 0, o = lfe, i, c;
    switch (flags & CH_MASK) {
    case CH_MONO: case CH_DUAL1: case CH_DUAL2:
        map[n++] = o; break;
    case CH_DUAL: case CH_STEREO: case CH_DOLBY:
        map[n++] = o; map[n++] = o + 1; break;
    case CH_3F:                                   /* L C R -> FL FR FC */
        map[n++] = o; map[n++] = o + 2; map[n++] = o + 1; break;
    case CH_2F1R:                                 /* L R S -> FL FR BC */
        map[n++] = o; map[n++] = o + 1; if (lfe) map[n++] = 0; map[n++] = o + 2; lfe = 0; break;
    case CH_3F1R:                                 /* L C R S -> FL FR FC (LFE) BC */
        map[n++] = o; map[n++] = o + 2; map[n++] = o + 1; if (lfe) map[n++] = 0; map[n++] = o + 3; lfe = 0; break;
    case CH_2F2R:                                 /* L R SL SR -> FL FR (LFE) BL BR */
        map[n++] = o; map[n++] = o + 1; if (lfe) map[n++] = 0; map[n++] = o + 2; map[n++] = o + 3; lfe = 0; break;
    case CH_3F2R:
        map[n++] = o; map[n++] = o + 2; map[n++] = o + 1; if (lfe) map[n++] = 0;
        map[n++] = o + 3; map[n++] = o + 4; lfe = 0; break;
    }
    if (lfe) map[n++] = 0;
    for (i = 0; i < 256; i++)
        for (c = 0; c < n; c++) dst[i * n + c] = to_s16(planes[256 * map[c] + i]);
}

/* ------------------------------------------------------------------ */

int orc_a52_decode_frames(const uint8_t *frames, int n, int frame_bytes, int flags,
                          float level, float bias, float *pcm)
{
    orc_a52_t *st = orc_a52_init();
    int f, b, errors = 0;
    if (!st) return -1;
    for (f = 0; f < n; f++) {
        int fl = flags, nout;
        float lv = level;
        const uint8_t *buf = frames + (size_t)f * frame_bytes;
        if (orc_a52_frame(st, buf, &fl, &lv, bias)) { errors++; continue; }
        nout = nfchans_of[fl & CH_MASK] + ((fl & F_LFE) ? 1 : 0);
        for (b = 0; b < 6; b++) {
            if (orc_a52_block(st)) errors++;
            if (pcm) memcpy(pcm + ((size_t)f * 6 + b) * nout * 256, st->samples, nout * 256 * sizeof(float));
        }
    }
    orc_a52_free(st);
    return errors;
}

long orc_a52_bitpos(orc_a52_t *st) { return (long)st->bitpos; }

