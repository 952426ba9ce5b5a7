// capi.hip — the C-ABI of libac3mi.so (include/ac3mi.h): context, device memory,
// table construction and the batched entry points.
#include "ac3mi_internal.h"
#include "a52_levels.h"
#include "spec_tables.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <vector>

namespace ac3mi {

static std::string g_err;   // errors raised without a context

static const uint8_t kNfchans[11] = {2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2};

// ---------------------------------------------------------------------------
// tables (double precision on the host, rounded once to float)

static double bessel_i0(double x)
{
    // L52/imdct.c:347-356: 100-term series in Horner form
    double b = 1;
    for (int i = 100; i > 0; i--) b = b * x / (i * i) + 1;
    return b;
}

void build_host_tables(float *window, float2 *tw_long, float2 *tw_short)
{
    const double pi = 3.14159265358979323846;
    // KBD window, alpha = 5 (L52/imdct.c:364-372)
    double acc = 0, cum[256];
    for (int i = 0; i < 256; i++) {
        acc += bessel_i0(i * (256 - i) * (5 * pi / 256) * (5 * pi / 256));
        cum[i] = acc;
    }
    acc += 1;
    for (int i = 0; i < 256; i++) window[i] = (float)sqrt(cum[i] / acc);

    // merged lane twiddles (xform_core.h): lane n2, register k1
    //   long : (-1)^n2 e^{-j pi (n2+63.75)/256} . e^{-j 2 pi n2 k1/128} . e^{-j pi (k1+.5)/256}
    //   short: e^{-j pi (n-.25)/128} . e^{-j 2 pi n k1/64} . e^{-j pi (k1+.5)/128},  n = lane & 3
    for (int l = 0; l < 8; l++)
        for (int k = 0; k < 16; k++) {
            double a = -pi * (l + 63.75) / 256 - 2 * pi * l * k / 128 - pi * (k + 0.5) / 256 + (l & 1 ? pi : 0);
            tw_long[l * 16 + k] = make_float2((float)cos(a), (float)sin(a));
            int n = l & 3;
            double b = -pi * (n - 0.25) / 128 - 2 * pi * n * k / 64 - pi * (k + 0.5) / 128;
            tw_short[l * 16 + k] = make_float2((float)cos(b), (float)sin(b));
        }
}

// decoder tables: L52/bit_allocate.c:31-101, L52/tables.h:49-246
// descriptor of a row byte (same function as decode_common.h's mant_desc, host copy)
static uint32_t mant_desc_host(uint32_t b)
{
    const uint32_t k1 = b >> 5, nbp = b & 31u;
    const uint32_t qbase = k1 == 1 ? 0u : k1 == 2 ? 96u : k1 == 3 ? 480u : nbp == 3 ? 736u : nbp == 4 ? 744u : 0u;
    const uint32_t per = k1 == 3 ? 2u : k1 ? 3u : 0u, obits = k1 == 1 ? 5u : k1 ? 7u : 0u;
    const uint32_t coded = (k1 || nbp == 3 || nbp == 4) ? 1u : 0u;
    return qbase | (per << 10) | (obits << 12) | (coded << 15);
}

void build_dec_tables(DecTables *t, uint16_t *lfsr_seq, uint16_t *lfsr_idx)
{
    uint8_t la[256];
    build_logadd(la);
    for (int i = 0; i < 256; i++) t->la_neg[i] = (int8_t)-la[i];
    memcpy(t->hth, kHth, sizeof t->hth);
    memcpy(t->width, kWidth, sizeof t->width);
    memcpy(t->band_end, kBandEnd, sizeof t->band_end);
    for (int i = 0; i < 256; i++) {
        int b = i < 20 ? i : 20;
        while (i >= 20 && b < 49 && i >= kBandEnd[b - 20]) b++;
        t->band_of_bin[i] = (uint8_t)b;
    }
    // Q(x) = ROUND(32768 x) for the symmetric quantiser levels (L52/tables.h:49)
    auto q = [](int num, int den) {
        double x = 32768.0 * num / den;
        return (float)(int)(x + (x > 0 ? 0.5 : -0.5));
    };
    memset(t->qlev, 0, sizeof t->qlev);
    for (int i = 0; i < 3; i++) t->qlev[i] = q(2 * (i - 1), 3);
    for (int i = 0; i < 5; i++) t->qlev[3 + i] = q(2 * (i - 2), 5);
    for (int i = 0; i < 7; i++) t->qlev[8 + i] = q(2 * (i - 3), 7);
    for (int i = 0; i < 11; i++) t->qlev[16 + i] = q(2 * (i - 5), 11);
    for (int i = 0; i < 15; i++) t->qlev[27 + i] = q(2 * (i - 7), 15);
    memset(t->qtab, 0, sizeof t->qtab);
    for (int code = 0; code < 27; code++) {
        t->qtab[code * 3 + 0] = t->qlev[code / 9];
        t->qtab[code * 3 + 1] = t->qlev[(code / 3) % 3];
        t->qtab[code * 3 + 2] = t->qlev[code % 3];
    }
    for (int code = 0; code < 125; code++) {
        t->qtab[96 + code * 3 + 0] = t->qlev[3 + code / 25];
        t->qtab[96 + code * 3 + 1] = t->qlev[3 + (code / 5) % 5];
        t->qtab[96 + code * 3 + 2] = t->qlev[3 + code % 5];
    }
    for (int code = 0; code < 121; code++) {
        t->qtab[480 + code * 2 + 0] = t->qlev[16 + code / 11];
        t->qtab[480 + code * 2 + 1] = t->qlev[16 + code % 11];
    }
    for (int code = 0; code < 7; code++) t->qtab[736 + code] = t->qlev[8 + code];
    for (int code = 0; code < 15; code++) t->qtab[744 + code] = t->qlev[27 + code];
    for (int b = 0; b < 128; b++) t->desc[b] = mant_desc_host((uint32_t)b);
    // dither generator (L52/parse.c:310-319, table L52/tables.h:213-246): one call advances a
    // 16-bit Galois LFSR (feedback 0xa011) by 8 steps.  It is GF(2)-linear with period 65535,
    // so the sequence from state 1 plus its inverse index give O(1) access to any later draw.
    uint16_t step8[256];
    step8[0] = 0;
    for (int i = 1; i < 256; i <<= 1) {
        uint16_t v = (i == 1) ? 0xa011 : (uint16_t)((step8[i >> 1] << 1) ^ ((step8[i >> 1] & 0x8000) ? 0xa011 : 0));
        step8[i] = v;
        for (int k = 1; k < i; k++) step8[i + k] = (uint16_t)(v ^ step8[k]);
    }
    memset(lfsr_idx, 0, 65536 * sizeof(uint16_t));
    uint16_t s = 1;
    for (int i = 0; i < 65535; i++) {
        lfsr_seq[i] = s;
        lfsr_idx[s] = (uint16_t)i;
        s = (uint16_t)(step8[s >> 8] ^ (uint16_t)(s << 8));
    }
}

// encoder tables: ENC/ac3tab.h and AC3_encode_init (ENC/ac3enc.cpp:977-1016, 1094-1104)
static int16_t fix15(float a)                        // ENC/ac3enc.cpp:428-439
{
    int v = (int)(a * (float)(1 << 15));
    if (v < -32767) v = -32767; else if (v > 32767) v = 32767;
    return (int16_t)v;
}

void build_enc_tables(EncTables *t)
{
    const double pi = 3.14159265358979323846;
    // ac3_window (ENC/ac3tab.h:15-48) = floor(32768 * KBD alpha-5 window)
    double acc = 0, cum[256];
    for (int i = 0; i < 256; i++) {
        acc += bessel_i0(i * (256 - i) * (5 * pi / 256) * (5 * pi / 256));
        cum[i] = acc;
    }
    acc += 1;
    for (int i = 0; i < 256; i++) t->win[i] = (int16_t)floor(32768.0 * sqrt(cum[i] / acc));
    // fft_init(7): cos/sin of a float argument -> float overloads in the reference's C++ unit (:441-459)
    for (int i = 0; i < 64; i++) {
        float alpha = (float)(2 * pi * (float)i / (float)128);
        t->cos[i] = fix15(cosf(alpha));
        t->sin[i] = fix15(sinf(alpha));
    }
    for (int i = 0; i < 128; i++) {
        int m = 0;
        for (int j = 0; j < 7; j++) m |= ((i >> j) & 1) << (6 - j);
        t->bitrev[i] = (uint8_t)m;
        float alpha = (float)(2 * pi * (i + 1.0 / 8.0) / (float)512);      // :1098-1102
        t->xcos[i] = fix15(-cosf(alpha));
        t->xsin[i] = fix15(-sinf(alpha));
    }
    build_logadd(t->latab);
    for (int b = 0; b < 50; b++)
        for (int f = 0; f < 3; f++) t->hth[b][f] = (uint16_t)(0xc00 - kHth[f][b]);
    // baptab: address (psd - mask) >> 5 -> bap code; same table as the decoder's widths, as codes
    for (int a = 0; a < 64; a++) {
        const int w = kWidth[63 - a];
        int code;
        switch (w) {
        case 0: code = 0; break;
        case -1: code = 1; break;
        case -2: code = 2; break;
        case 3: code = 3; break;
        case -3: code = 4; break;
        case 4: code = 5; break;
        case 14: code = 14; break;
        case 16: code = 15; break;
        default: code = w + 1; break;       // widths 5..12 -> codes 6..13
        }
        t->baptab[a] = (uint8_t)code;
    }
    // band structure: bins 0..27 are 1-wide, then kBandEnd
    int start = 0, k = 0;
    for (int b = 0; b < 50; b++) {
        const int end = b < 20 ? b + 1 : kBandEnd[b - 20];
        t->band_start[b] = (uint8_t)start;
        t->band_size[b] = (uint8_t)(end - start);
        for (; k < end; k++) t->band_of_bin[k] = (uint8_t)b;
        start = end;
    }
    for (; k < 256; k++) t->band_of_bin[k] = 49;
    t->band_start[50] = 0;
    for (int n = 0; n < 256; n++) {                                          // ac3_crc_init :998-1016
        unsigned c = (unsigned)n << 8;
        for (int j = 0; j < 8; j++) c = (c & 0x8000) ? (((c << 1) & 0xffff) ^ 0x8005) : (c << 1);
        t->crc_tab[n] = (uint16_t)c;
    }
}

int enc_config(int freq, int bitrate, int channels, EncConfig *c)
{
    static const uint8_t acmod_of[6] = {1, 2, 3, 6, 7, 7};
    const int *rates = kSampleRates, *kbps = kKbps;
    if (channels < 1 || channels > 6) return 0;
    c->nch = channels;
    c->acmod = acmod_of[channels - 1];
    c->lfe = channels == 6;
    c->nfbw = channels > 5 ? 5 : channels;
    bool found = false;
    for (int i = 0; i < 3 && !found; i++)
        for (int j = 0; j < 3; j++)
            if ((rates[j] >> i) == freq) { c->halfrate = i; c->fscod = j; found = true; break; }
    if (!found) return 0;
    c->bsid = 8 + c->halfrate;
    bitrate /= 1000;
    int i;
    for (i = 0; i < 19; i++) if ((kbps[i] >> c->halfrate) == bitrate) break;
    if (i == 19) return 0;
    c->frmsizecod = i << 1;
    c->frame_words = (bitrate * 1000 * 1536) / (freq * 16);
    return c->frame_words * 2;
}

// ---------------------------------------------------------------------------
// a52_downmix() as a plane-mixing matrix (L52/downmix.c:480-619) and the set of
// outputs a52_downmix_init() can grant (L52/downmix.c:37-67)

int build_mix_plan(int acmod, int lfeon, int output, MixPlan *plan)
{
    // row = requested output, column = coded acmod
    static const uint8_t grant[11][8] = {
        {0, 10, 2, 2, 2, 2, 2, 2},  {1, 1, 1, 1, 1, 1, 1, 1},  {0, 10, 2, 2, 2, 2, 2, 2}, {0, 10, 2, 3, 2, 3, 2, 3},
        {0, 10, 2, 2, 4, 4, 4, 4},  {0, 10, 2, 2, 4, 5, 4, 5}, {0, 10, 2, 3, 6, 6, 6, 6}, {0, 10, 2, 3, 6, 7, 6, 7},
        {8, 1, 1, 1, 1, 1, 1, 1},   {9, 1, 1, 1, 1, 1, 1, 1},  {0, 10, 2, 10, 10, 10, 10, 10}};
    if (acmod < 0 || acmod > 7) return AC3MI_ERR_ARG;
    const int out = output & AC3MI_CHANNEL_MASK;
    if (out > AC3MI_DOLBY) return AC3MI_ERR_ARG;
    if (grant[out][acmod] != out) return AC3MI_ERR_ARG;
    if ((output & AC3MI_LFE) && !lfeon) return AC3MI_ERR_ARG;

    memset(plan, 0, sizeof *plan);
    plan->nfchans = kNfchans[acmod];
    plan->in_lfe = lfeon ? 1 : 0;
    plan->n_in = plan->nfchans + plan->in_lfe;
    const int out_lfe = (output & AC3MI_LFE) ? 1 : 0;
    const int nfo = kNfchans[out];
    plan->n_out = nfo + out_lfe;

    int8_t m[5][5];
    memset(m, 0, sizeof m);
    auto id = [&](int n) { for (int i = 0; i < n; i++) m[i][i] = 1; };
    auto fold_centre = [&]() { m[0][0] = 1; m[0][1] = 1; m[1][2] = 1; m[1][1] = 1; };   // mix3to2
    const int A = acmod;

    switch (out) {
    case AC3MI_CHANNEL:  case AC3MI_CHANNEL1:
        id(nfo);                                                    // (0,0) identity; CHANNEL1 keeps plane 0
        break;
    case AC3MI_CHANNEL2:
        m[0][1] = 1;                                                // downmix.c:485-487
        break;
    case AC3MI_MONO:
        for (int c = 0; c < plan->nfchans; c++) m[0][c] = 1;        // mix2to1..mix5to1
        break;
    case AC3MI_STEREO:
        switch (A) {
        case 2: id(2); break;
        case 3: fold_centre(); break;
        case 4: m[0][0] = 1; m[1][1] = 1; m[0][2] = 1; m[1][2] = 1; break;                     // mix21to2
        case 5: fold_centre(); m[0][3] = 1; m[1][3] = 1; break;                                // mix31to2
        case 6: m[0][0] = 1; m[0][2] = 1; m[1][1] = 1; m[1][3] = 1; break;                     // 2x mix2to1
        case 7: fold_centre(); m[0][3] = 1; m[1][4] = 1; break;                                // mix32to2
        default: return AC3MI_ERR_ARG;
        }
        break;
    case AC3MI_DOLBY:
        switch (A) {
        case 1: m[0][0] = 1; m[1][0] = 1; break;                                               // memcpy
        case 2: id(2); break;
        case 3: fold_centre(); break;
        case 4: m[0][0] = 1; m[1][1] = 1; m[0][2] = -1; m[1][2] = 1; break;                    // mix21toS
        case 5: fold_centre(); m[0][3] = -1; m[1][3] = 1; break;                               // mix31toS
        case 6: m[0][0] = 1; m[1][1] = 1; m[0][2] = m[0][3] = -1; m[1][2] = m[1][3] = 1; break; // mix22toS
        case 7: fold_centre(); m[0][3] = m[0][4] = -1; m[1][3] = m[1][4] = 1; break;           // mix32toS
        default: return AC3MI_ERR_ARG;
        }
        break;
    case AC3MI_3F:
        id(3);
        if (A == 5) { m[0][3] = 1; m[2][3] = 1; }                                              // mix21to2
        if (A == 7) { m[0][3] = 1; m[2][4] = 1; }
        break;
    case AC3MI_2F1R:
        if (A == 4) id(3);
        else if (A == 5) { fold_centre(); m[2][3] = 1; }
        else if (A == 6) { id(2); m[2][2] = 1; m[2][3] = 1; }
        else if (A == 7) { fold_centre(); m[2][3] = 1; m[2][4] = 1; }                          // move2to1
        else return AC3MI_ERR_ARG;
        break;
    case AC3MI_3F1R:
        id(4);
        if (A == 7) m[3][4] = 1;
        break;
    case AC3MI_2F2R:
        if (A == 6) id(4);
        else if (A == 4) { id(3); m[3][2] = 1; }
        else if (A == 5) { fold_centre(); m[2][3] = 1; m[3][3] = 1; }
        else if (A == 7) { fold_centre(); m[2][3] = 1; m[3][4] = 1; }
        else return AC3MI_ERR_ARG;
        break;
    case AC3MI_3F2R:
        if (A == 7) id(5);
        else if (A == 5) { id(4); m[4][3] = 1; }
        else return AC3MI_ERR_ARG;
        break;
    }
    for (int o = 0; o < nfo; o++)
        for (int c = 0; c < plan->nfchans; c++) plan->mix[o + out_lfe][c + plan->in_lfe] = m[o][c];
    if (out_lfe) plan->mix[0][0] = 1;
    // the outputs whose time-domain mixers test slev == 0 (downmix.c:494-583), from the acmods with surround channels
    if ((A & 4) && (out == AC3MI_MONO || out == AC3MI_STEREO || out == AC3MI_3F)) {
        const int nsurr = A >= 6 ? 2 : 1;
        for (int c = plan->nfchans - nsurr; c < plan->nfchans; c++) plan->surr_mask |= (uint8_t)(1u << (c + plan->in_lfe));
        if (out == AC3MI_STEREO && !(A & 1)) plan->nobias_mask = (uint8_t)(3u << out_lfe);             // 2/1, 2/2: L and R
        if (out == AC3MI_3F) plan->nobias_mask = (uint8_t)(5u << out_lfe);                             // 3/1, 3/2: L and R, not C
    }
    return AC3MI_OK;
}

}  // namespace ac3mi

using namespace ac3mi;

#define HIPCHK(ctx, call)                                                              \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            std::string msg_ = std::string(#call) + ": " + hipGetErrorString(e_);      \
            if (ctx) (ctx)->err = msg_; else g_err = msg_;                             \
            return AC3MI_ERR_HIP;                                                      \
        }                                                                              \
    } while (0)

// Measurement aid: what one SIMD sustains in plain 32-bit VALU instructions while the whole chip is busy with them
// (the issue ceiling the instruction-bound kernels are priced against; the clock under such a load is not the
// data-sheet peak).  Every wavefront executes iters x 32 v_add_u32 / v_xor_b32 on eight independent registers.
namespace ac3mi {
__global__ __launch_bounds__(256) void valu_probe_kernel(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t k = blockIdx.x | 1u;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
            "v_add_u32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
            "v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
            "v_add_u32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
            "v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
            "v_add_u32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
            "v_add_u32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
            "v_add_u32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
            : "v"(k));
    }
    const uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (r == 0x12345678u) out[0] = r;               // keeps the chain alive; practically never taken
}
// the same for the scalar unit: iters x 32 s_add_u32 / s_xor_b32 on four independent registers per wavefront
__global__ __launch_bounds__(256) void salu_probe_kernel(uint32_t *out, int iters)
{
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    for (int i = 0; i < iters; i++) {
        asm volatile(
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            "s_add_u32 %0, %0, 1\n s_xor_b32 %1, %1, 3\n s_add_u32 %2, %2, 5\n s_xor_b32 %3, %3, 7\n"
            : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    }
    if ((s0 ^ s1 ^ s2 ^ s3) == 0x12345678u) out[0] = s0;
}
// both at once, three vector instructions to one scalar one (the encoder's and the front end's mix): iters x (24 + 8) per wavefront
__global__ __launch_bounds__(256) void mixed_probe_kernel(uint32_t *out, int iters)
{
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5;
    uint32_t s0 = blockIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    const uint32_t k = blockIdx.x | 1u;
    for (int i = 0; i < iters; i++) {
#define AC3MI_MIX4 "v_add_u32 %0, %0, %10\n v_xor_b32 %1, %1, %10\n v_add_u32 %2, %2, %10\n s_add_u32 %6, %6, 1\n" \
                   "v_xor_b32 %3, %3, %10\n v_add_u32 %4, %4, %10\n v_xor_b32 %5, %5, %10\n s_xor_b32 %7, %7, 3\n" \
                   "v_add_u32 %0, %0, %10\n v_xor_b32 %1, %1, %10\n v_add_u32 %2, %2, %10\n s_add_u32 %8, %8, 5\n" \
                   "v_xor_b32 %3, %3, %10\n v_add_u32 %4, %4, %10\n v_xor_b32 %5, %5, %10\n s_xor_b32 %9, %9, 7\n"
        asm volatile(AC3MI_MIX4 AC3MI_MIX4
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3)
                     : "v"(k) : "scc");
#undef AC3MI_MIX4
    }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ s0 ^ s1 ^ s2 ^ s3) == 0x12345678u) out[0] = a0;
}
}  // namespace ac3mi

extern "C" {

int ac3mi_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *ac3mi_last_error(const ac3mi_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

static int ctx_init(ac3mi_ctx *ctx)
{
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIPCHK(ctx, hipEventCreate(&ctx->ev0));
    HIPCHK(ctx, hipEventCreate(&ctx->ev1));
    HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
    std::vector<float> win(256);
    std::vector<float2> twl(128), tws(128);
    build_host_tables(win.data(), twl.data(), tws.data());
    HIPCHK(ctx, hipMalloc(&ctx->tab.window, 256 * sizeof(float)));
    HIPCHK(ctx, hipMalloc(&ctx->tab.tw_long, 128 * sizeof(float2)));
    HIPCHK(ctx, hipMalloc(&ctx->tab.tw_short, 128 * sizeof(float2)));
    HIPCHK(ctx, hipMemcpy(ctx->tab.window, win.data(), 256 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->tab.tw_long, twl.data(), 128 * sizeof(float2), hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->tab.tw_short, tws.data(), 128 * sizeof(float2), hipMemcpyHostToDevice));
    {
        DecTables dt;
        // (the sequence runs on for LFSR_EXT entries past its period: mant_kernel indexes a block's draws without a modulo)
        constexpr int LFSR_EXT = 16384;
        std::vector<uint16_t> seq(65535 + LFSR_EXT), idx(65536);
        build_dec_tables(&dt, seq.data(), idx.data());
        for (int i = 0; i < LFSR_EXT; i++) seq[65535 + i] = seq[i];
        HIPCHK(ctx, hipMalloc(&ctx->tab.dec, sizeof dt));
        HIPCHK(ctx, hipMalloc(&ctx->tab.lfsr_seq, seq.size() * sizeof(uint16_t)));
        HIPCHK(ctx, hipMalloc(&ctx->tab.lfsr_idx, 65536 * sizeof(uint16_t)));
        HIPCHK(ctx, hipMemcpy(ctx->tab.dec, &dt, sizeof dt, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->tab.lfsr_seq, seq.data(), seq.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(ctx->tab.lfsr_idx, idx.data(), 65536 * sizeof(uint16_t), hipMemcpyHostToDevice));
    }
    {
        EncTables et;
        build_enc_tables(&et);
        HIPCHK(ctx, hipMalloc(&ctx->tab.enc, sizeof et));
        HIPCHK(ctx, hipMemcpy(ctx->tab.enc, &et, sizeof et, hipMemcpyHostToDevice));
    }
    return AC3MI_OK;
}

ac3mi_ctx *ac3mi_create(int device)
{
    int n = ac3mi_device_count();
    if (n <= 0) {
        g_err = "ac3mi_create: no HIP device visible (libac3mi has no CPU fallback)";
        return nullptr;
    }
    if (device < 0 || device >= n) {
        g_err = "ac3mi_create: device index out of range";
        return nullptr;
    }
    ac3mi_ctx *ctx = new ac3mi_ctx();
    ctx->device = device;
    ctx->stream = nullptr;
    ctx->tab = DeviceTables{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    ctx->ws_enc = nullptr;
    ctx->ws_enc_bytes = 0;
    ctx->ws_tc = nullptr;
    ctx->ws_tc_bytes = 0;
    ctx->slots = nullptr;
    ctx->decode_mode = 0;
    if (const char *e = getenv("AC3MI_DECODE_MODE")) {          // test aid: default front-end variant (ac3mi_set_decode_mode)
        const int m = atoi(e);
        if (m >= 0 && m <= 6 && m != 2) ctx->decode_mode = m;
    }
    ctx->tile_frames = 131072;
    ctx->encode_mode = 0;
    if (const char *e = getenv("AC3MI_ENCODE_MODE")) {          // test aid: default packer variant (ac3mi_set_encode_mode)
        const int m = atoi(e);
        if (m >= 0 && m <= 2) ctx->encode_mode = m;
    }
    ctx->ws_draws = nullptr;
    ctx->ws_draws_bytes = 0;
    ctx->ws_split = nullptr;
    ctx->ws_split_bytes = 0;
    ctx->ws_coef = nullptr;
    ctx->ws_blksw = nullptr;
    ctx->mix_pending = nullptr;
    ctx->mix_flags = nullptr;
    ctx->ws_coef_bytes = ctx->ws_blksw_bytes = 0;
    if (ctx_init(ctx) != AC3MI_OK) {
        g_err = ctx->err;
        delete ctx;
        return nullptr;
    }
    return ctx;
}

void ac3mi_destroy(ac3mi_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(ctx->tab.window);
    (void)hipFree(ctx->tab.tw_long);
    (void)hipFree(ctx->tab.tw_short);
    (void)hipFree(ctx->tab.dec);
    (void)hipFree(ctx->tab.lfsr_seq);
    (void)hipFree(ctx->tab.lfsr_idx);
    (void)hipFree(ctx->ws_coef);
    (void)hipFree(ctx->ws_blksw);
    (void)hipFree(ctx->ws_enc);
    (void)hipFree(ctx->ws_tc);
    (void)hipFree(ctx->ws_draws);
    (void)hipFree(ctx->ws_split);
    (void)hipFree(ctx->tab.enc);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

void *ac3mi_dev_alloc(ac3mi_ctx *ctx, size_t bytes)
{
    void *p = nullptr;
    if (!ctx) return nullptr;
    if (hipSetDevice(ctx->device) != hipSuccess) return nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) {
        ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return nullptr;
    }
    return p;
}

void ac3mi_dev_free(ac3mi_ctx *ctx, void *d_ptr)
{
    if (!ctx || !d_ptr) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_ptr);
}

int ac3mi_memcpy_h2d(ac3mi_ctx *ctx, void *d_dst, const void *h_src, size_t bytes)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AC3MI_OK;
}

int ac3mi_memcpy_d2h(ac3mi_ctx *ctx, void *h_dst, const void *d_src, size_t bytes)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AC3MI_OK;
}

int ac3mi_memset(ac3mi_ctx *ctx, void *d_dst, int byte, size_t bytes)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipMemsetAsync(d_dst, byte, bytes, ctx->stream));
    return AC3MI_OK;
}

int ac3mi_memcpy_d2d(ac3mi_ctx *ctx, void *d_dst, const void *d_src, size_t bytes)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return AC3MI_OK;
}

int ac3mi_sync(ac3mi_ctx *ctx)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return AC3MI_OK;
}

int ac3mi_timer_start(ac3mi_ctx *ctx)
{
    if (!ctx) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return AC3MI_OK;
}

int ac3mi_timer_stop(ac3mi_ctx *ctx, float *elapsed_ms)
{
    if (!ctx || !elapsed_ms) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    HIPCHK(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return AC3MI_OK;
}

int ac3mi_probe_valu_rate(ac3mi_ctx *ctx, double *ginst_per_s_per_simd)
{
    if (!ctx || !ginst_per_s_per_simd) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount, wg_per_cu = 8, iters = 8192;        // 8 wavefronts per SIMD
    uint32_t *d = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d, 64));
    hipLaunchKernelGGL(valu_probe_kernel, dim3(cus * wg_per_cu), dim3(256), 0, ctx->stream, d, 64);       // warm-up
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(valu_probe_kernel, dim3(cus * wg_per_cu), dim3(256), 0, ctx->stream, d, iters);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    (void)hipFree(d);
    // per SIMD: wg_per_cu workgroups x 4 wavefronts / 4 SIMDs = wg_per_cu wavefronts, 32 instructions per iteration each
    *ginst_per_s_per_simd = (double)wg_per_cu * iters * 32.0 / (ms * 1e-3) / 1e9;
    return AC3MI_OK;
}

int ac3mi_probe_salu_rate(ac3mi_ctx *ctx, double *ginst_per_s_per_simd)
{
    if (!ctx || !ginst_per_s_per_simd) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount, wg_per_cu = 8, iters = 4096;        // 8 wavefronts per SIMD
    uint32_t *d = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d, 64));
    hipLaunchKernelGGL(salu_probe_kernel, dim3(cus * wg_per_cu), dim3(256), 0, ctx->stream, d, 64);       // warm-up
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(salu_probe_kernel, dim3(cus * wg_per_cu), dim3(256), 0, ctx->stream, d, iters);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    (void)hipFree(d);
    *ginst_per_s_per_simd = (double)wg_per_cu * iters * 32.0 / (ms * 1e-3) / 1e9;
    return AC3MI_OK;
}

int ac3mi_probe_mixed_rate(ac3mi_ctx *ctx, int waves_per_simd, double *valu_ginst_per_s_per_simd, double *salu_ginst_per_s_per_simd)
{
    if (!ctx || !valu_ginst_per_s_per_simd || !salu_ginst_per_s_per_simd || waves_per_simd < 1 || waves_per_simd > 8) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipDeviceProp_t prop;
    HIPCHK(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount, wg_per_cu = waves_per_simd, iters = 8192;      // a 256-thread workgroup = one wavefront per SIMD
    uint32_t *d = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&d, 64));
    hipLaunchKernelGGL(mixed_probe_kernel, dim3(cus * wg_per_cu), dim3(256), 0, ctx->stream, d, 64);      // warm-up
    HIPCHK(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    hipLaunchKernelGGL(mixed_probe_kernel, dim3(cus * wg_per_cu), dim3(256), 0, ctx->stream, d, iters);
    HIPCHK(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    (void)hipFree(d);
    *valu_ginst_per_s_per_simd = (double)wg_per_cu * iters * 24.0 / (ms * 1e-3) / 1e9;
    *salu_ginst_per_s_per_simd = (double)wg_per_cu * iters * 8.0 / (ms * 1e-3) / 1e9;
    return AC3MI_OK;
}

// one float4 per lane and the workgroup ends: the copy the guide quotes (MI355X_MICROARCH.md, HBM) and the fastest one
// measured on this part (profiles/hbm_calibrate)
__global__ __launch_bounds__(256) void copy_probe_kernel(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}

int ac3mi_probe_copy_rate(ac3mi_ctx *ctx, size_t bytes, double *gbytes_per_s)
{
    if (!ctx || !gbytes_per_s || bytes < (1u << 20)) return AC3MI_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const size_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    HIPCHK(ctx, hipMalloc((void **)&a, n * 16));
    if (hipMalloc((void **)&b, n * 16) != hipSuccess) { (void)hipFree(a); ctx->err = "ac3mi_probe_copy_rate: out of memory"; return AC3MI_ERR_HIP; }
    (void)hipMemsetAsync(a, 0, n * 16, ctx->stream);
    const unsigned grid = (unsigned)((n + 255) / 256);
    float best = 1e30f;
    for (int it = 0; it < 6; it++) {
        (void)hipEventRecord(ctx->ev0, ctx->stream);
        hipLaunchKernelGGL(copy_probe_kernel, dim3(grid), dim3(256), 0, ctx->stream, a, b, n);
        (void)hipEventRecord(ctx->ev1, ctx->stream);
        if (hipEventSynchronize(ctx->ev1) != hipSuccess) break;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess && it > 0 && ms < best) best = ms;
    }
    (void)hipFree(a);
    (void)hipFree(b);
    if (best > 1e29f) { ctx->err = "ac3mi_probe_copy_rate: timing failed"; return AC3MI_ERR_HIP; }
    *gbytes_per_s = 2.0 * (double)(n * 16) / (best * 1e-3) / 1e9;
    return AC3MI_OK;
}

int ac3mi_set_state_slots(ac3mi_ctx *ctx, const int32_t *d_slots)
{
    if (!ctx) return AC3MI_ERR_ARG;
    ctx->slots = d_slots;
    return AC3MI_OK;
}

int ac3mi_set_mix_state(ac3mi_ctx *ctx, float *d_pending, int32_t *d_flags)
{
    if (!ctx || ((d_pending == nullptr) != (d_flags == nullptr))) return AC3MI_ERR_ARG;
    ctx->mix_pending = d_pending;
    ctx->mix_flags = d_flags;
    return AC3MI_OK;
}

int ac3mi_set_decode_mode(ac3mi_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 6 || mode == 2) return AC3MI_ERR_ARG;       // (2: the one-kernel front end per frame, retired in round 4)
    ctx->decode_mode = mode;
    return AC3MI_OK;
}

int ac3mi_set_encode_mode(ac3mi_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return AC3MI_ERR_ARG;
    ctx->encode_mode = mode;
    return AC3MI_OK;
}

int ac3mi_set_tile_frames(ac3mi_ctx *ctx, long long frames)
{
    if (!ctx || frames < 0) return AC3MI_ERR_ARG;
    ctx->tile_frames = frames;
    return AC3MI_OK;
}

// Streams per tile when a batch of n_streams x frames_per_stream is above the workspace bound, else 0 (no tiling).
static int tile_streams(const ac3mi_ctx *ctx, int n_streams, int frames_per_stream)
{
    if (!ctx->tile_frames || n_streams < 2 || frames_per_stream < 1) return 0;
    if ((long long)n_streams * frames_per_stream <= ctx->tile_frames) return 0;
    const long long g = ctx->tile_frames / frames_per_stream;
    return (int)(g < 1 ? 1 : g);
}

// frame-parallel front end for few long streams?  (auto: more than one frame per stream and too few streams to fill
// the chip with one wavefront each: 256 CUs x 20 wavefronts)
static bool use_frame_parallel(const ac3mi_ctx *ctx, int n_streams, int frames_per_stream)
{
    if (ctx->decode_mode) return ctx->decode_mode == 5;            // (forced: also for one-frame streams, A/B runs)
    if (frames_per_stream < 2) return false;
    return n_streams < 5120;
}

// split front end (parse kernel, then one wavefront per audio block: decode.hip MODE 4 / 5 + mant_kernel)?  auto: yes;
// mode 1 forces the one-kernel front end (one wavefront per stream), 4 / 5 the split one per stream / per frame.
static bool use_split(const ac3mi_ctx *ctx)
{
    return ctx->decode_mode == 0 || ctx->decode_mode >= 4;
}

// ... and its mantissa kernel with the transform fused in (decode_mx.hip: no coefficient planes in HBM)?  Needs one-frame
// streams (a block's overlap tail goes to the next block inside the frame's workgroup) and every coded plane an output
// plane.  Mode 6: wherever that holds; auto: for s16 output only - measured per 65 536 5.1 frames (profiles/r04_kernel_stats.csv,
// profiles/EXPERIMENTS.md): to s16 2.12 - 2.17 ms fused against 1.31 + 0.77 = 2.08 - 2.15 ms in two kernels (a wash in time,
// 36.9 KB per frame less HBM traffic and workspace), to float 2.45 against 1.31 + 0.97 = 2.28 ms (the two kernels stay);
// modes 4 / 5 always keep the two kernels (the bit-identity reference, A/B runs).
static bool use_mantx(const ac3mi_ctx *ctx, int frames_per_stream, bool identity, bool s16)
{
    return ((ctx->decode_mode == 0 && s16) || ctx->decode_mode == 6) && frames_per_stream == 1 && identity;
}

// the encoder's workspace for `rows` channel-blocks (6 x channels per frame):
// mdct | raw exponents | encoded exponents | masking curves | exp_samples | strategies | exponent bits | search results | verdict tables
struct EncWs { size_t off_eexp, off_emask, off_shift, off_strat, off_ebits, off_snr, off_memo, need; };
static EncWs enc_ws_layout(size_t rows)
{
    EncWs w;
    const size_t rows_pad = (rows + 255) & ~(size_t)255;
    w.off_eexp = rows * 256 * 4 + rows * 256;
    w.off_emask = w.off_eexp + rows * 256;
    w.off_shift = w.off_emask + ((rows * 100 + 255) & ~(size_t)255);
    w.off_strat = w.off_shift + rows_pad;
    w.off_ebits = w.off_strat + rows_pad;
    w.off_snr = w.off_ebits + rows_pad * 4;
    w.off_memo = w.off_snr + rows_pad * 2;              // [S][F][2] int32 <= rows / 3 entries
    w.need = w.off_memo + rows_pad * 6 + 1024;          // [S][F][8] uint32
    return w;
}

// workspace of the split front end for nfr frames: descriptors, generator positions, coupling coordinates, row sets
struct SplitWs { void *desc; uint32_t *fpos; float *cplco; uint8_t *rows; };
static size_t split_bytes(size_t nfr) { return nfr * (6 * 80 + 16 + 6 * 90 * 4 + 6 * 7 * 512) + 256; }
static SplitWs split_ws(const ac3mi_ctx *ctx, size_t nfr, size_t f0)
{
    SplitWs w;
    uint8_t *p = (uint8_t *)ctx->ws_split;
    w.desc = p + f0 * 6 * 80;
    p += nfr * 6 * 80;
    w.fpos = (uint32_t *)p + f0;
    p += nfr * 16;
    w.cplco = (float *)p + f0 * 6 * 90;
    p += nfr * 6 * 90 * 4;
    w.rows = p + f0 * 6 * 7 * 512;
    return w;
}
static int ensure_split(ac3mi_ctx *ctx, size_t nfr)
{
    const size_t need = split_bytes(nfr);
    if (need <= ctx->ws_split_bytes) return AC3MI_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->ws_split);
    ctx->ws_split = nullptr;
    ctx->ws_split_bytes = 0;
    HIPCHK(ctx, hipMalloc((void **)&ctx->ws_split, need));
    ctx->ws_split_bytes = need;
    return AC3MI_OK;
}

// one workgroup per stream (decode_wg.hip)?  Its eight wavefronts cut the latency of a frame to a third and nothing but the
// frame and the PCM touches HBM, but a workgroup's wavefronts wait for each other at the block's barriers, so the chip holds
// fewer busy wavefronts than the other front ends.  Measured on one-frame streams to s16 (round 3, profiles/decode_ab.py;
// fused / split front end + transform / one-kernel front end + transform): 64 streams 0.084 / 0.118 / 0.167 ms, 256: 0.086 /
// 0.121 / 0.165, 1 024: 0.188 / 0.150 / 0.184, 2 048: 0.362 / 0.186 / 0.206, 4 096: 0.70 / 0.28 / 0.28, 65 536: 10.7 / 3.5 / 4.2.
// auto: batches of up to 512 streams of at most four frames (round 2: 1 024, against the one-kernel front end); mode 3 forces it.
static bool use_wg_kernel(const ac3mi_ctx *ctx, int n_streams, int frames_per_stream)
{
    if (ctx->decode_mode) return ctx->decode_mode == 3;
    return n_streams <= 512 && frames_per_stream <= 4;
}

static int ensure_draws(ac3mi_ctx *ctx, size_t nfr)
{
    const size_t need = nfr * 4 + nfr * 2 + 256;
    if (need <= ctx->ws_draws_bytes) return AC3MI_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->ws_draws);
    ctx->ws_draws = nullptr;
    ctx->ws_draws_bytes = 0;
    HIPCHK(ctx, hipMalloc((void **)&ctx->ws_draws, need));
    ctx->ws_draws_bytes = need;
    return AC3MI_OK;
}

size_t ac3mi_workspace_bytes(const ac3mi_ctx *ctx)
{
    if (!ctx) return 0;
    return ctx->ws_coef_bytes + ctx->ws_blksw_bytes + ctx->ws_enc_bytes + ctx->ws_tc_bytes + ctx->ws_draws_bytes + ctx->ws_split_bytes;
}

size_t ac3mi_transcode_workspace_plan(size_t frames, int frames_per_stream, int n_in, int nfchans, int n_out)
{
    // what ac3mi_transcode_batch holds for a call (or tile) of `frames` frames: coefficient planes + block-switch flags and
    // per-frame level flags (ensure_ws), the split front end's arrays (split_bytes), the s16 PCM between transform and
    // encoder, the encoder's arrays (enc_ws_layout) - the same expressions the call allocates with
    // (one-frame streams without a downmix: no coefficient planes, use_mantx)
    const size_t planes = (frames_per_stream == 1 && n_in == n_out) ? 0 : frames * 6 * (size_t)n_in * 256 * sizeof(float);
    return planes + (frames * 6 * (size_t)nfchans + 4 + frames) + split_bytes(frames) +
           (frames * 1536 * (size_t)n_out * 2 + 512) + enc_ws_layout(frames * 6 * (size_t)n_out).need;
}

int ac3mi_xform_planes(const ac3mi_xform_desc *desc, int *n_in, int *n_out)
{
    MixPlan plan;
    if (!desc) return AC3MI_ERR_ARG;
    int r = build_mix_plan(desc->acmod, desc->lfeon, desc->output, &plan);
    if (r != AC3MI_OK) return r;
    if (n_in) *n_in = plan.n_in;
    if (n_out) *n_out = plan.n_out;
    return AC3MI_OK;
}

int ac3mi_imdct_batch(ac3mi_ctx *ctx, const ac3mi_xform_desc *desc, const float *d_coeffs,
                      const uint8_t *d_blksw, float *d_delay, float *d_pcm, int n_streams,
                      int frames_per_stream)
{
    if (!ctx) return AC3MI_ERR_ARG;
    if (!desc || !d_coeffs || !d_delay || !d_pcm || n_streams < 0 || frames_per_stream < 0) {
        ctx->err = "ac3mi_imdct_batch: bad argument";
        return AC3MI_ERR_ARG;
    }
    XformLaunch L;
    int r = build_mix_plan(desc->acmod, desc->lfeon, desc->output, &L.plan);
    if (r != AC3MI_OK) {
        ctx->err = "ac3mi_imdct_batch: output configuration not reachable from acmod (a52_downmix_init)";
        return r;
    }
    if ((long long)n_streams * L.plan.n_out > 0x7fffffffLL) {
        ctx->err = "ac3mi_imdct_batch: too many chains for one launch";
        return AC3MI_ERR_ARG;
    }
    L.coef = d_coeffs;
    L.blksw = d_blksw;
    L.delay = d_delay;
    L.slot = ctx->slots;
    L.delay_stride = 6 * 128;
    L.pcm = d_pcm;
    L.n_streams = n_streams;
    L.frames = frames_per_stream;
    L.bias = desc->bias;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, launch_xform(ctx->tab, L, ctx->stream));
    return AC3MI_OK;
}

int ac3mi_syncinfo(const uint8_t *buf, int *flags, int *sample_rate, int *bit_rate)
{
    static const int kbps[19] = {32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640};
    static const uint8_t lfebit[8] = {0x10, 0x10, 0x04, 0x04, 0x04, 0x01, 0x04, 0x01};
    if (!buf || buf[0] != 0x0b || buf[1] != 0x77) return 0;
    if (buf[5] >= 0x60) return 0;
    const int bsid = buf[5] >> 3, half = bsid < 9 ? 0 : bsid - 8;
    const int acmod = buf[6] >> 5;
    if (flags) *flags = (((buf[6] & 0xf8) == 0x50) ? AC3MI_DOLBY : acmod) | ((buf[6] & lfebit[acmod]) ? AC3MI_LFE : 0);
    const int code = buf[4] & 63;
    if (code >= 38) return 0;
    const int rate = kbps[code >> 1];
    if (bit_rate) *bit_rate = (rate * 1000) >> half;
    switch (buf[4] & 0xc0) {
    case 0x00: if (sample_rate) *sample_rate = 48000 >> half; return 4 * rate;
    case 0x40: if (sample_rate) *sample_rate = 44100 >> half; return 2 * (320 * rate / 147 + (code & 1));
    case 0x80: if (sample_rate) *sample_rate = 32000 >> half; return 6 * rate;
    }
    return 0;
}

int ac3mi_decode_planes(const ac3mi_decode_desc *desc, int *n_out, int *out_flags)
{
    if (!desc || desc->acmod < 0 || desc->acmod > 7) return AC3MI_ERR_ARG;
    const int out = a52_granted_output(desc->flags & AC3MI_CHANNEL_MASK, desc->acmod);
    if (out < 0) return AC3MI_ERR_ARG;
    const int lfe = (desc->lfeon && (desc->flags & AC3MI_LFE)) ? AC3MI_LFE : 0;
    if (n_out) *n_out = kNfchans[out] + (lfe ? 1 : 0);
    if (out_flags) *out_flags = out | lfe;
    return AC3MI_OK;
}

static int ensure_ws(ac3mi_ctx *ctx, size_t coef_bytes, size_t blksw_bytes)
{
    if (coef_bytes > ctx->ws_coef_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->ws_coef);
        ctx->ws_coef = nullptr;
        ctx->ws_coef_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->ws_coef, coef_bytes));
        ctx->ws_coef_bytes = coef_bytes;
    }
    if (blksw_bytes > ctx->ws_blksw_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->ws_blksw);
        ctx->ws_blksw = nullptr;
        ctx->ws_blksw_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->ws_blksw, blksw_bytes));
        ctx->ws_blksw_bytes = blksw_bytes;
    }
    return AC3MI_OK;
}

// d_pcm (float planes) or d_pcm16 (interleaved s16, written by the transform itself)
static int decode_impl(ac3mi_ctx *ctx, const ac3mi_decode_desc *desc, const uint8_t *d_frames,
                       int frame_stride, int n_streams, int frames_per_stream, float *d_delay,
                       uint16_t *d_lfsr, float *d_pcm, int16_t *d_pcm16, uint32_t *d_status, const ac3mi_decode_taps *taps)
{
    if (!ctx) return AC3MI_ERR_ARG;
    if (!desc || !d_frames || !d_delay || !d_lfsr || (!d_pcm && !d_pcm16) || !d_status || n_streams < 0 ||
        frames_per_stream < 0 || desc->frame_bytes < 8 || desc->frame_bytes > 3840 ||
        frame_stride < ((desc->frame_bytes + 3) & ~3) || (frame_stride & 3) || ((uintptr_t)d_frames & 3)) {
        ctx->err = "ac3mi_decode_batch: bad argument";
        return AC3MI_ERR_ARG;
    }
    int n_out = 0, out_flags = 0;
    if (ac3mi_decode_planes(desc, &n_out, &out_flags) != AC3MI_OK) {
        ctx->err = "ac3mi_decode_batch: requested output not supported (a52_frame would return 1)";
        return AC3MI_ERR_ARG;
    }
    XformLaunch X;
    if (build_mix_plan(desc->acmod, desc->lfeon, out_flags, &X.plan) != AC3MI_OK) {
        ctx->err = "ac3mi_decode_batch: no mix plan for this acmod/output";
        return AC3MI_ERR_ARG;
    }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    bool identity = X.plan.n_in == X.plan.n_out;
    for (int o = 0; o < X.plan.n_out && identity; o++)
        for (int c = 0; c < X.plan.n_in; c++)
            if (X.plan.mix[o][c] != (o == c ? 1 : 0)) identity = false;
    const bool wgk = use_wg_kernel(ctx, n_streams, frames_per_stream);
    const bool fused = wgk && identity && !taps;                 // no workspace at all: nothing to tile
    if (const int g = (taps || fused) ? 0 : tile_streams(ctx, n_streams, frames_per_stream)) {
        // bounded workspace: whole streams at a time (streams are independent; state arrays move with them)
        const int32_t *slots0 = ctx->slots;
        float *const mp0 = ctx->mix_pending;
        int32_t *const mf0 = ctx->mix_flags;
        int rc = AC3MI_OK;
        for (int s0 = 0; s0 < n_streams && rc == AC3MI_OK; s0 += g) {
            const int ns = n_streams - s0 < g ? n_streams - s0 : g;
            const size_t f0 = (size_t)s0 * frames_per_stream;
            if (slots0) ctx->slots = slots0 + s0;
            else if (mp0) { ctx->mix_pending = mp0 + (size_t)s0 * n_out * 128; ctx->mix_flags = mf0 + (size_t)s0 * 6; }
            rc = decode_impl(ctx, desc, d_frames + f0 * frame_stride, frame_stride, ns, frames_per_stream,
                             slots0 ? d_delay : d_delay + (size_t)s0 * n_out * 128, slots0 ? d_lfsr : d_lfsr + s0,
                             d_pcm ? d_pcm + f0 * 6 * n_out * 256 : nullptr, d_pcm16 ? d_pcm16 + f0 * 6 * n_out * 256 : nullptr,
                             d_status + f0, nullptr);
        }
        ctx->slots = slots0;
        ctx->mix_pending = mp0;
        ctx->mix_flags = mf0;
        return rc;
    }
    const size_t nfr = (size_t)n_streams * frames_per_stream;
    float *coef = taps && taps->d_coef ? taps->d_coef : nullptr;
    uint8_t *blksw = taps && taps->d_blksw ? taps->d_blksw : nullptr;
    // liba52's overlap bookkeeping around frames with surround level 0: the front end tells the transform which they are
    const bool mixstate = ctx->mix_pending && X.plan.surr_mask && !identity;
    uint8_t *zs = nullptr;
    const bool mantx = !wgk && !taps && use_split(ctx) && use_mantx(ctx, frames_per_stream, identity, d_pcm16 != nullptr);
    if (!fused) {
        const size_t zs_off = blksw ? 0 : nfr * 6 * X.plan.nfchans + 4;
        int r = ensure_ws(ctx, (coef || mantx) ? 0 : nfr * 6 * X.plan.n_in * 256 * sizeof(float), zs_off + (mixstate ? nfr : 0));
        if (r != AC3MI_OK) return r;
        if (mixstate) zs = ctx->ws_blksw + zs_off;
        if (!coef) coef = ctx->ws_coef;
        if (!blksw) blksw = ctx->ws_blksw;
    }
    X.mix_pending = mixstate ? ctx->mix_pending : nullptr;
    X.mix_flags = mixstate ? ctx->mix_flags : nullptr;

    const bool fp = !wgk && use_frame_parallel(ctx, n_streams, frames_per_stream);
    if (fp) { const int r = ensure_draws(ctx, nfr); if (r != AC3MI_OK) return r; }
    const bool split = !wgk && use_split(ctx);
    if (split) { const int r = ensure_split(ctx, nfr); if (r != AC3MI_OK) return r; }
    // One workgroup per stream, one wavefront per channel, the transform fused in when every
    // coded plane is an output plane (decode_wg.hip) - no coefficient planes in HBM.  Stage taps and mixing outputs
    // take the same front end with the planes written out, then the transform kernel.
    if (wgk) {
        DecodeLaunch D;
        D.frames = d_frames;
        D.frame_bytes = desc->frame_bytes;
        D.frame_stride = frame_stride;
        D.n_streams = n_streams;
        D.frames_per_stream = frames_per_stream;
        D.req_flags = desc->flags;
        D.acmod = desc->acmod;
        D.lfeon = desc->lfeon ? 1 : 0;
        D.dynrng_on = desc->dynrng ? 1 : 0;
        D.level = desc->level;
        D.coef = coef;
        D.blksw = blksw;
        D.zs = zs;
        D.status = d_status;
        D.lfsr = d_lfsr;
        D.slot = ctx->slots;
        D.tap_exp = taps && taps->d_exp ? taps->d_exp : nullptr;
        D.tap_bap = taps && taps->d_bap ? taps->d_bap : nullptr;
        D.dyn_out = taps ? taps->d_dynrng_out : nullptr;
        D.dyn_in = taps ? taps->d_dynrng_in : nullptr;
        D.frame_parallel = 0;
        D.frame_draws = nullptr;
        D.frame_lfsr = nullptr;
        X.coef = coef;
        X.blksw = blksw;
        X.zs = zs;
        X.delay = d_delay;
        X.slot = ctx->slots;
        X.delay_stride = 6 * 128;
        X.pcm = d_pcm;
        X.pcm16 = d_pcm16;
        X.s16_flags = out_flags;
        X.n_streams = n_streams;
        X.frames = frames_per_stream;
        X.bias = desc->bias;
        if (fused) {
            HIPCHK(ctx, launch_decode_wg(ctx->tab, D, &X, 0, ctx->stream));
            return AC3MI_OK;
        }
        HIPCHK(ctx, launch_decode_wg(ctx->tab, D, nullptr, 0, ctx->stream));
        HIPCHK(ctx, launch_xform(ctx->tab, X, ctx->stream));
        return AC3MI_OK;
    }
    // One pass over the batch, kernels back to back on the context's stream.  Rounds 1-2 sent large batches through in two
    // chunks with the transform of chunk k on a second stream beside the front end of chunk k+1 (and round 3 tried the
    // mantissa kernel on a third): measured again with the split front end, 1 / 2 / 3 / 4 chunks = 3.545 / 3.549 / 3.572 /
    // 3.616 ms per 65 536 frames - a kernel of this size fills the chip, only the tails overlap - so the pipeline is gone.
    // (Also tried in round 3: the batch in 2 / 4 / 8 / 16 tiles, front end + transform per tile, so that a tile's coefficient
    // planes would still sit in the 256 MiB Infinity Cache when the transform reads them: 3.49 / 3.63 / 3.89 / 4.47 ms against
    // 3.48 ms in one piece - each kernel's tail costs more than the cache gives.)
    const int n_chunks = 1;
    const size_t F = (size_t)frames_per_stream;
    for (int k = 0; k < n_chunks; k++) {
        const int s0 = (int)((long long)n_streams * k / n_chunks), s1 = (int)((long long)n_streams * (k + 1) / n_chunks);
        const int ns = s1 - s0;
        if (ns <= 0) continue;
        const size_t f0 = (size_t)s0 * F;
        DecodeLaunch D;
        D.frames = d_frames + f0 * frame_stride;
        D.frame_bytes = desc->frame_bytes;
        D.frame_stride = frame_stride;
        D.n_streams = ns;
        D.frames_per_stream = frames_per_stream;
        D.req_flags = desc->flags;
        D.acmod = desc->acmod;
        D.lfeon = desc->lfeon ? 1 : 0;
        D.dynrng_on = desc->dynrng ? 1 : 0;
        D.level = desc->level;
        D.coef = mantx ? nullptr : coef + f0 * 6 * X.plan.n_in * 256;
        D.blksw = blksw + f0 * 6 * X.plan.nfchans;
        D.zs = zs ? zs + f0 : nullptr;
        D.status = d_status + f0;
        D.lfsr = ctx->slots ? d_lfsr : d_lfsr + s0;
        D.slot = ctx->slots ? ctx->slots + s0 : nullptr;
        D.tap_exp = taps && taps->d_exp ? taps->d_exp + f0 * 6 * 7 * 256 : nullptr;
        D.tap_bap = taps && taps->d_bap ? taps->d_bap + f0 * 6 * 7 * 256 : nullptr;
        D.dyn_out = taps && taps->d_dynrng_out ? taps->d_dynrng_out + f0 * 12 : nullptr;
        D.dyn_in = taps && taps->d_dynrng_in ? taps->d_dynrng_in + f0 * 12 : nullptr;
        D.frame_parallel = fp ? 1 : 0;
        D.frame_draws = fp ? ctx->ws_draws + f0 : nullptr;
        D.frame_lfsr = fp ? (uint16_t *)(ctx->ws_draws + nfr) + f0 : nullptr;
        if (split) {
            const SplitWs w = split_ws(ctx, nfr, f0);
            D.split = 1;
            D.ws_desc = w.desc; D.ws_fpos = w.fpos; D.ws_cplco = w.cplco; D.ws_rows = w.rows;
        }
        hipStream_t xs = ctx->stream;
        X.coef = D.coef;
        X.blksw = D.blksw;
        X.zs = D.zs;
        if (mixstate && !ctx->slots) {
            X.mix_pending = ctx->mix_pending + (size_t)s0 * X.plan.n_out * 128;
            X.mix_flags = ctx->mix_flags + (size_t)s0 * 6;
        }
        X.delay = ctx->slots ? d_delay : d_delay + (size_t)s0 * X.plan.n_out * 128;
        X.slot = D.slot;
        X.delay_stride = 6 * 128;
        X.pcm = d_pcm ? d_pcm + f0 * 6 * X.plan.n_out * 256 : nullptr;
        X.pcm16 = d_pcm16 ? d_pcm16 + f0 * 6 * X.plan.n_out * 256 : nullptr;
        X.s16_flags = out_flags;
        X.n_streams = ns;
        X.frames = frames_per_stream;
        X.bias = desc->bias;
        if (mantx) D.fuse = &X;                             // the mantissa kernel transforms too (decode_mx.hip)
        HIPCHK(ctx, launch_decode(ctx->tab, D, ctx->stream));
        if (!mantx) HIPCHK(ctx, launch_xform(ctx->tab, X, xs));
    }
    return AC3MI_OK;
}

int ac3mi_decode_batch(ac3mi_ctx *ctx, const ac3mi_decode_desc *desc, const uint8_t *d_frames,
                       int frame_stride, int n_streams, int frames_per_stream, float *d_delay,
                       uint16_t *d_lfsr, float *d_pcm, uint32_t *d_status, const ac3mi_decode_taps *taps)
{
    if (ctx && !d_pcm) {
        ctx->err = "ac3mi_decode_batch: bad argument";
        return AC3MI_ERR_ARG;
    }
    return decode_impl(ctx, desc, d_frames, frame_stride, n_streams, frames_per_stream, d_delay, d_lfsr, d_pcm, nullptr, d_status, taps);
}

int ac3mi_decode_s16_batch(ac3mi_ctx *ctx, const ac3mi_decode_desc *desc, const uint8_t *d_frames,
                           int frame_stride, int n_streams, int frames_per_stream, float *d_delay,
                           uint16_t *d_lfsr, int16_t *d_pcm16, uint32_t *d_status)
{
    if (!ctx) return AC3MI_ERR_ARG;
    if (!desc || !d_pcm16 || ((uintptr_t)d_pcm16 & 15)) {
        ctx->err = "ac3mi_decode_s16_batch: bad argument (d_pcm16 must be 16-byte aligned)";
        return AC3MI_ERR_ARG;
    }
    ac3mi_decode_desc dd = *desc;
    dd.level = 1.0f;                                    // what the reference's converters presuppose (src/AC3ACM.cpp:1553)
    dd.bias = 384.0f;
    return decode_impl(ctx, &dd, d_frames, frame_stride, n_streams, frames_per_stream, d_delay, d_lfsr, nullptr, d_pcm16, d_status, nullptr);
}

int ac3mi_encode_frame_bytes(const ac3mi_encode_desc *desc)
{
    EncConfig c;
    if (!desc) return 0;
    return enc_config(desc->sample_rate, desc->bit_rate, desc->channels, &c);
}

int ac3mi_encode_tables(int16_t *costab64, int16_t *sintab64, int16_t *xcos128, int16_t *xsin128, int16_t *window256)
{
    EncTables et;
    build_enc_tables(&et);
    if (costab64) memcpy(costab64, et.cos, sizeof et.cos);
    if (sintab64) memcpy(sintab64, et.sin, sizeof et.sin);
    if (xcos128) memcpy(xcos128, et.xcos, sizeof et.xcos);
    if (xsin128) memcpy(xsin128, et.xsin, sizeof et.xsin);
    if (window256) memcpy(window256, et.win, sizeof et.win);
    return AC3MI_OK;
}

int ac3mi_encode_spec_tables(int16_t *window256, uint8_t *latab256, uint16_t *hth50x3, uint8_t *baptab64, uint8_t *bndsz50,
                             uint16_t *sdecay4, uint16_t *fdecay4, uint16_t *sgain4, uint16_t *dbknee4, uint16_t *floor8,
                             uint16_t *fgain8, uint16_t *freqs3, uint16_t *bitrate19)
{
    EncTables et;
    build_enc_tables(&et);
    if (window256) memcpy(window256, et.win, sizeof et.win);
    if (latab256) memcpy(latab256, et.latab, sizeof et.latab);
    if (hth50x3) memcpy(hth50x3, et.hth, sizeof et.hth);
    if (baptab64) memcpy(baptab64, et.baptab, sizeof et.baptab);
    if (bndsz50) memcpy(bndsz50, et.band_size, sizeof et.band_size);
    for (int i = 0; i < 4; i++) {
        if (sdecay4) sdecay4[i] = (uint16_t)enc_sdecay(i);
        if (fdecay4) fdecay4[i] = (uint16_t)enc_fdecay(i);
        if (sgain4) sgain4[i] = (uint16_t)enc_sgain(i);
        if (dbknee4) dbknee4[i] = (uint16_t)enc_dbknee(i);
    }
    for (int i = 0; i < 8; i++) {
        if (floor8) floor8[i] = (uint16_t)enc_floor(i);
        if (fgain8) fgain8[i] = (uint16_t)enc_fgain(i);
    }
    for (int i = 0; i < 3; i++) if (freqs3) freqs3[i] = (uint16_t)kSampleRates[i];
    for (int i = 0; i < 19; i++) if (bitrate19) bitrate19[i] = (uint16_t)kKbps[i];
    return AC3MI_OK;
}

int ac3mi_encode_batch(ac3mi_ctx *ctx, const ac3mi_encode_desc *desc, const int16_t *d_pcm, const uint8_t *chmap,
                       int16_t *d_last, int32_t *d_csnroffst, uint8_t *d_frames, int frame_stride, int n_streams,
                       int frames_per_stream, const ac3mi_encode_taps *taps)
{
    if (!ctx) return AC3MI_ERR_ARG;
    EncodeLaunch E;
    if (!desc || !d_pcm || !chmap || !d_last || !d_csnroffst || !d_frames || n_streams < 0 || frames_per_stream < 0) {
        ctx->err = "ac3mi_encode_batch: bad argument";
        return AC3MI_ERR_ARG;
    }
    const int fb = enc_config(desc->sample_rate, desc->bit_rate, desc->channels, &E.cfg);
    if (fb <= 0) {
        ctx->err = "ac3mi_encode_batch: AC3_encode_init would return 0 for this rate/bitrate/channels";
        return AC3MI_ERR_ARG;
    }
    if (frame_stride < ((fb + 3) & ~3) || (frame_stride & 3) || ((uintptr_t)d_frames & 3) || ((uintptr_t)d_pcm & 1)) {
        ctx->err = "ac3mi_encode_batch: frame_stride must be a multiple of 4 and >= the frame size";
        return AC3MI_ERR_ARG;
    }
    for (int i = 0; i < 8; i++) E.chmap[i] = i < desc->channels ? chmap[i] : 0;
    for (int i = 0; i < desc->channels; i++)
        if (E.chmap[i] >= desc->channels) {
            ctx->err = "ac3mi_encode_batch: chmap entry out of range";
            return AC3MI_ERR_ARG;
        }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int g = taps ? 0 : tile_streams(ctx, n_streams, frames_per_stream)) {
        const int32_t *slots0 = ctx->slots;
        int rc = AC3MI_OK;
        for (int s0 = 0; s0 < n_streams && rc == AC3MI_OK; s0 += g) {
            const int ns = n_streams - s0 < g ? n_streams - s0 : g;
            const size_t f0 = (size_t)s0 * frames_per_stream;
            if (slots0) ctx->slots = slots0 + s0;
            rc = ac3mi_encode_batch(ctx, desc, d_pcm + f0 * 1536 * E.cfg.nch, chmap,
                                    slots0 ? d_last : d_last + (size_t)s0 * E.cfg.nch * 256, slots0 ? d_csnroffst : d_csnroffst + s0,
                                    d_frames + f0 * frame_stride, frame_stride, ns, frames_per_stream, nullptr);
        }
        ctx->slots = slots0;
        return rc;
    }
    const size_t rows = (size_t)n_streams * frames_per_stream * 6 * E.cfg.nch;
    const EncWs W = enc_ws_layout(rows);
    const size_t off_eexp = W.off_eexp, off_emask = W.off_emask, off_shift = W.off_shift, off_strat = W.off_strat;
    const size_t off_ebits = W.off_ebits, off_snr = W.off_snr, off_memo = W.off_memo, need = W.need;
    if (need > ctx->ws_enc_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->ws_enc);
        ctx->ws_enc = nullptr;
        ctx->ws_enc_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->ws_enc, need));
        ctx->ws_enc_bytes = need;
    }
    E.ws_mdct = (int32_t *)ctx->ws_enc;
    E.ws_expo = nullptr;                                                   // raw exponents leave the MDCT kernel only as a tap
    E.ws_eexp = (uint8_t *)ctx->ws_enc + off_eexp;
    E.ws_emask = (int16_t *)((uint8_t *)ctx->ws_enc + off_emask);
    E.ws_shift = (int8_t *)ctx->ws_enc + off_shift;
    E.ws_strat = (uint8_t *)ctx->ws_enc + off_strat;
    E.ws_ebits = (int32_t *)((uint8_t *)ctx->ws_enc + off_ebits);
    E.ws_snr = (int32_t *)((uint8_t *)ctx->ws_enc + off_snr);
    E.ws_memo = (uint32_t *)((uint8_t *)ctx->ws_enc + off_memo);
    if (taps && taps->d_mdct) { E.ws_mdct = taps->d_mdct; E.mdct_full_rows = true; }
    if (taps && taps->d_exponent) E.ws_expo = taps->d_exponent;
    if (taps && taps->d_exp_samples) E.ws_shift = taps->d_exp_samples;
    if (taps && taps->d_encoded_exp) E.ws_eexp = taps->d_encoded_exp;      // the exponent stage writes the tap directly
    E.pcm = d_pcm;
    E.last = d_last;
    E.csnr = d_csnroffst;
    E.slot = ctx->slots;
    E.pack_mode = ctx->encode_mode;
    E.frames = d_frames;
    E.frame_stride = frame_stride;
    E.n_streams = n_streams;
    E.frames_per_stream = frames_per_stream;
    E.tap_eexp = taps ? taps->d_encoded_exp : nullptr;
    E.tap_bap = taps ? taps->d_bap : nullptr;
    E.tap_strat = taps ? taps->d_exp_strategy : nullptr;
    E.tap_snr = taps ? taps->d_snroffst : nullptr;
    if ((E.tap_bap == nullptr) != (E.tap_eexp == nullptr)) {
        ctx->err = "ac3mi_encode_batch: d_bap and d_encoded_exp taps must be given together";
        return AC3MI_ERR_ARG;
    }
    HIPCHK(ctx, launch_encode(ctx->tab, E, ctx->stream));
    return AC3MI_OK;
}

int ac3mi_transcode_batch(ac3mi_ctx *ctx, const ac3mi_decode_desc *dec, const ac3mi_encode_desc *enc,
                          const uint8_t *d_frames_in, int in_stride, int n_streams, int frames_per_stream,
                          float *d_delay, uint16_t *d_lfsr, const uint8_t *chmap, int16_t *d_last,
                          int32_t *d_csnroffst, uint8_t *d_frames_out, int out_stride, uint32_t *d_status)
{
    if (!ctx) return AC3MI_ERR_ARG;
    if (!dec || !enc || !d_frames_in || !d_delay || !d_lfsr || !chmap || !d_last || !d_csnroffst || !d_frames_out || !d_status ||
        n_streams < 0 || frames_per_stream < 0 || dec->frame_bytes < 8 || dec->frame_bytes > 3840 ||
        in_stride < ((dec->frame_bytes + 3) & ~3) || (in_stride & 3) || ((uintptr_t)d_frames_in & 3)) {
        ctx->err = "ac3mi_transcode_batch: bad argument";
        return AC3MI_ERR_ARG;
    }
    ac3mi_decode_desc dd = *dec;
    dd.level = 1.0f;                                    // the s16 converter reads the 16-bit value out of the float (bias 384)
    dd.bias = 384.0f;
    int n_out = 0, out_flags = 0;
    XformLaunch X;
    if (ac3mi_decode_planes(&dd, &n_out, &out_flags) != AC3MI_OK || build_mix_plan(dd.acmod, dd.lfeon, out_flags, &X.plan) != AC3MI_OK) {
        ctx->err = "ac3mi_transcode_batch: requested output not supported (a52_frame would return 1)";
        return AC3MI_ERR_ARG;
    }
    EncodeLaunch E;
    E.pack_mode = ctx->encode_mode;
    const int fb = enc_config(enc->sample_rate, enc->bit_rate, enc->channels, &E.cfg);
    if (fb <= 0 || enc->channels != n_out) {
        ctx->err = "ac3mi_transcode_batch: encoder configuration rejected, or its channel count differs from the decoder's output";
        return AC3MI_ERR_ARG;
    }
    if (out_stride < ((fb + 3) & ~3) || (out_stride & 3) || ((uintptr_t)d_frames_out & 3)) {
        ctx->err = "ac3mi_transcode_batch: out_stride must be a multiple of 4 and >= the frame size";
        return AC3MI_ERR_ARG;
    }
    for (int i = 0; i < 8; i++) E.chmap[i] = i < enc->channels ? chmap[i] : 0;
    for (int i = 0; i < enc->channels; i++)
        if (E.chmap[i] >= enc->channels) { ctx->err = "ac3mi_transcode_batch: chmap entry out of range"; return AC3MI_ERR_ARG; }
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (const int g = tile_streams(ctx, n_streams, frames_per_stream)) {
        const int32_t *slots0 = ctx->slots;
        float *const mp0 = ctx->mix_pending;
        int32_t *const mf0 = ctx->mix_flags;
        int rc = AC3MI_OK;
        for (int s0 = 0; s0 < n_streams && rc == AC3MI_OK; s0 += g) {
            const int ns = n_streams - s0 < g ? n_streams - s0 : g;
            const size_t f0 = (size_t)s0 * frames_per_stream;
            if (slots0) ctx->slots = slots0 + s0;
            else if (mp0) { ctx->mix_pending = mp0 + (size_t)s0 * n_out * 128; ctx->mix_flags = mf0 + (size_t)s0 * 6; }
            rc = ac3mi_transcode_batch(ctx, dec, enc, d_frames_in + f0 * in_stride, in_stride, ns, frames_per_stream,
                                       slots0 ? d_delay : d_delay + (size_t)s0 * n_out * 128, slots0 ? d_lfsr : d_lfsr + s0, chmap,
                                       slots0 ? d_last : d_last + (size_t)s0 * E.cfg.nch * 256, slots0 ? d_csnroffst : d_csnroffst + s0,
                                       d_frames_out + f0 * out_stride, out_stride, d_status + f0);
        }
        ctx->slots = slots0;
        ctx->mix_pending = mp0;
        ctx->mix_flags = mf0;
        return rc;
    }
    const size_t F = (size_t)frames_per_stream, nfr = (size_t)n_streams * F;
    // workspaces: decoder planes, float PCM + s16 PCM, encoder arrays
    const bool mixstate = ctx->mix_pending && X.plan.surr_mask;       // as in decode_impl
    const size_t zs_off = nfr * 6 * X.plan.nfchans + 4;
    bool identity = X.plan.n_in == X.plan.n_out;
    for (int o = 0; o < X.plan.n_out && identity; o++)
        for (int c = 0; c < X.plan.n_in; c++)
            if (X.plan.mix[o][c] != (o == c ? 1 : 0)) identity = false;
    const bool fused = identity && use_wg_kernel(ctx, n_streams, frames_per_stream);     // decode_wg.hip writes the s16 PCM itself
    const bool split = !fused && use_split(ctx);
    const bool mantx = split && use_mantx(ctx, frames_per_stream, identity, true);             // decode_mx.hip: mantissas + transform
    { const int r = ensure_ws(ctx, mantx ? 0 : nfr * 6 * X.plan.n_in * 256 * sizeof(float), zs_off + (mixstate ? nfr : 0)); if (r != AC3MI_OK) return r; }
    uint8_t *const zs = mixstate ? ctx->ws_blksw + zs_off : nullptr;
    const size_t s16_bytes = nfr * 1536 * n_out * 2;       // the transform writes s16 itself: no float PCM in between
    if (s16_bytes + 512 > ctx->ws_tc_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->ws_tc);
        ctx->ws_tc = nullptr;
        ctx->ws_tc_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->ws_tc, s16_bytes + 512));
        ctx->ws_tc_bytes = s16_bytes + 512;
    }
    int16_t *ws_s16 = (int16_t *)ctx->ws_tc;
    const size_t rows = nfr * 6 * E.cfg.nch;
    const EncWs W = enc_ws_layout(rows);
    const size_t off_eexp = W.off_eexp, off_emask = W.off_emask, off_shift = W.off_shift, off_strat = W.off_strat;
    const size_t off_ebits = W.off_ebits, off_snr = W.off_snr, off_memo = W.off_memo, need = W.need;
    if (need > ctx->ws_enc_bytes) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->ws_enc);
        ctx->ws_enc = nullptr;
        ctx->ws_enc_bytes = 0;
        HIPCHK(ctx, hipMalloc(&ctx->ws_enc, need));
        ctx->ws_enc_bytes = need;
    }
    const bool fp = !fused && use_frame_parallel(ctx, n_streams, frames_per_stream);
    if (fp) { const int r = ensure_draws(ctx, nfr); if (r != AC3MI_OK) return r; }
    if (split) { const int r = ensure_split(ctx, nfr); if (r != AC3MI_OK) return r; }
    // Decoder front end, transform to s16, encoder: back to back on the context's stream.  (Until round 2 two chunks were
    // pipelined over two streams; profiles/transcode_overlap.py measured 12.20 ms with and 12.23 - 12.27 ms without it per
    // 65 536 cold frames, and the sum of the separate decode-to-s16 and encode calls at 12.1 - 12.2 ms: no overlap to keep.
    // Round 4 tried it for MID-SIZE calls, where the one-wavefront-per-stream kernels take their single-wavefront latency of
    // 90 - 140 us whatever the batch: chunks of 2 048 streams on two HIP streams, one chunk's encoder beside the next one's
    // decoder - 3 072 / 4 096 / 8 192 / 16 384 streams 0.561 / 0.644 / 1.265 / 2.175 ms against 0.537 / 0.639 / 1.142 / 2.110
    // in one piece: a kernel of 2 048 frames already occupies every CU (the six-wavefront-per-frame kernels) or is dispatched
    // whole before the other stream's (the others), so the chunks' latency floors add up instead of overlapping.)
    const int n_chunks = 1;
    auto chunk_lo = [&](int k) { return (int)((long long)n_streams * k / n_chunks); };
    auto front = [&](int k) -> hipError_t {
        const int s0 = chunk_lo(k), ns = chunk_lo(k + 1) - s0;
        const size_t f0 = (size_t)s0 * F;
        DecodeLaunch D;
        D.frames = d_frames_in + f0 * in_stride;
        D.frame_bytes = dd.frame_bytes;
        D.frame_stride = in_stride;
        D.n_streams = ns;
        D.frames_per_stream = frames_per_stream;
        D.req_flags = dd.flags;
        D.acmod = dd.acmod;
        D.lfeon = dd.lfeon ? 1 : 0;
        D.dynrng_on = dd.dynrng ? 1 : 0;
        D.level = dd.level;
        D.coef = mantx ? nullptr : ctx->ws_coef + f0 * 6 * X.plan.n_in * 256;
        D.blksw = ctx->ws_blksw + f0 * 6 * X.plan.nfchans;
        D.zs = zs ? zs + f0 : nullptr;
        D.status = d_status + f0;
        D.lfsr = ctx->slots ? d_lfsr : d_lfsr + s0;
        D.slot = ctx->slots ? ctx->slots + s0 : nullptr;
        D.tap_exp = nullptr;
        D.tap_bap = nullptr;
        D.frame_parallel = fp ? 1 : 0;
        D.frame_draws = fp ? ctx->ws_draws + f0 : nullptr;
        D.frame_lfsr = fp ? (uint16_t *)(ctx->ws_draws + nfr) + f0 : nullptr;
        if (split) {
            const SplitWs w = split_ws(ctx, nfr, f0);
            D.split = 1;
            D.ws_desc = w.desc; D.ws_fpos = w.fpos; D.ws_cplco = w.cplco; D.ws_rows = w.rows;
        }
        if (fused || mantx) {
            XformLaunch Y = X;
            Y.coef = nullptr;
            Y.blksw = nullptr;
            Y.delay = ctx->slots ? d_delay : d_delay + (size_t)s0 * n_out * 128;
            Y.slot = D.slot;
            Y.delay_stride = 6 * 128;
            Y.pcm = nullptr;
            Y.pcm16 = ws_s16 + f0 * 1536 * n_out;
            Y.s16_flags = out_flags;
            Y.n_streams = ns;
            Y.frames = frames_per_stream;
            Y.bias = 384.0f;
            if (fused) return launch_decode_wg(ctx->tab, D, &Y, 0, ctx->stream);
            D.fuse = &Y;
            return launch_decode(ctx->tab, D, ctx->stream);
        }
        return launch_decode(ctx->tab, D, ctx->stream);
    };
    auto middle = [&](int k, hipStream_t st) -> hipError_t {
        const int s0 = chunk_lo(k), ns = chunk_lo(k + 1) - s0;
        const size_t f0 = (size_t)s0 * F;
        XformLaunch Y = X;
        Y.coef = ctx->ws_coef + f0 * 6 * X.plan.n_in * 256;
        Y.blksw = ctx->ws_blksw + f0 * 6 * X.plan.nfchans;
        Y.zs = zs ? zs + f0 : nullptr;
        if (mixstate) {
            Y.mix_pending = ctx->slots ? ctx->mix_pending : ctx->mix_pending + (size_t)s0 * n_out * 128;
            Y.mix_flags = ctx->slots ? ctx->mix_flags : ctx->mix_flags + (size_t)s0 * 6;
        }
        Y.delay = ctx->slots ? d_delay : d_delay + (size_t)s0 * n_out * 128;
        Y.slot = ctx->slots ? ctx->slots + s0 : nullptr;
        Y.delay_stride = 6 * 128;
        Y.pcm = nullptr;
        Y.pcm16 = ws_s16 + f0 * 1536 * n_out;
        Y.s16_flags = out_flags;
        Y.n_streams = ns;
        Y.frames = frames_per_stream;
        Y.bias = 384.0f;
        return launch_xform(ctx->tab, Y, st);
    };
    auto back = [&](int k) -> hipError_t {
        const int s0 = chunk_lo(k), ns = chunk_lo(k + 1) - s0;
        const size_t f0 = (size_t)s0 * F, r0 = f0 * 6 * E.cfg.nch;
        EncodeLaunch G = E;
        G.ws_mdct = (int32_t *)ctx->ws_enc + r0 * 256;
        G.ws_expo = nullptr;
        G.ws_eexp = (uint8_t *)ctx->ws_enc + off_eexp + r0 * 256;
        G.ws_emask = (int16_t *)((uint8_t *)ctx->ws_enc + off_emask) + r0 * 50;
        G.ws_shift = (int8_t *)ctx->ws_enc + off_shift + r0;
        G.ws_strat = (uint8_t *)ctx->ws_enc + off_strat + r0;
        G.ws_ebits = (int32_t *)((uint8_t *)ctx->ws_enc + off_ebits) + f0 * E.cfg.nch;
        G.ws_snr = (int32_t *)((uint8_t *)ctx->ws_enc + off_snr) + f0 * 2;
        G.ws_memo = (uint32_t *)((uint8_t *)ctx->ws_enc + off_memo) + f0 * 8;
        G.pcm = ws_s16 + f0 * 1536 * n_out;
        G.last = ctx->slots ? d_last : d_last + (size_t)s0 * E.cfg.nch * 256;
        G.csnr = ctx->slots ? d_csnroffst : d_csnroffst + s0;
        G.slot = ctx->slots ? ctx->slots + s0 : nullptr;
        G.frames = d_frames_out + f0 * out_stride;
        G.frame_stride = out_stride;
        G.n_streams = ns;
        G.frames_per_stream = frames_per_stream;
        G.tap_eexp = G.tap_bap = G.tap_strat = nullptr;
        G.tap_snr = nullptr;
        if (split) {                                        // block 0's descriptor of every frame carries the source's SNR offsets
            const SplitWs w = split_ws(ctx, nfr, f0);
            G.search_hint = reinterpret_cast<const uint32_t *>(w.desc) + 72 / 4;         // BlkDesc::src_snr (decode_common.h asserts the offset)
            G.search_hint_stride = 6 * 80 / 4;                                           // six 80-byte descriptors per frame
        }
        return launch_encode(ctx->tab, G, ctx->stream);
    };
    HIPCHK(ctx, front(0));
    if (!fused && !mantx) HIPCHK(ctx, middle(0, ctx->stream));
    HIPCHK(ctx, back(0));
    return AC3MI_OK;
}

}  // extern "C"
