// spec_tables.h — ATSC A/52 parametric bit-allocation constants (host side; uploaded to the
// device at context creation).  Same numbers as the reference's L52/bit_allocate.c:31-101
// (decoder form) and ENC/ac3tab.h:51-171 (encoder form); the encoder form is derived:
//   enc_hth[band][fs] = 0xc00 - hth[fs][band],   enc_latab[i] = -la_neg[i]
#pragma once
#include <stdint.h>

namespace ac3mi {

// hearing threshold, decoder sign convention, [fscod][band]
static const uint16_t kHth[3][50] = {    /* hearing threshold, bit_allocate.c:31-47 */
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

// bits per mantissa for address a = mask + 4*exp in -63..0 (index a+63); negative = grouped
// code (-1: 3-level, -2: 5-level, -3: 11-level); a <= -64 -> 16 bits, a > 0 -> 0 bits
static const int8_t kWidth[64] = {
    16, 16, 16, 16, 16, 16, 16, 16, 16, 14, 14, 14, 14, 14, 14, 14,
    14, 12, 12, 12, 12, 11, 11, 11, 11, 10, 10, 10, 10, 9, 9, 9,
    9, 8, 8, 8, 8, 7, 7, 7, 7, 6, 6, 6, 6, 5, 5, 5,
    5, 4, 4, -3, -3, 3, 3, 3, -2, -2, -1, -1, -1, -1, -1, 0
};

// first bin after band 20+i
static const uint8_t kBandEnd[30] = {       /* first bin after band 20+i, bit_allocate.c:74-76 */
    21, 22, 23, 24, 25, 26, 27, 28, 31, 34, 37, 40, 43, 46, 49, 55, 61, 67, 73, 79,
    85, 97, 109, 121, 133, 157, 181, 205, 229, 253
};

// log-addition table as run lengths: kLaRun[k] consecutive entries hold the value 64-k
static const uint8_t kLaRun[65] = {
    1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1,
    1, 2, 1, 1, 2, 1, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2,
    2, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 3, 2, 3, 3, 3,
    3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 8, 8, 11, 14, 19, 32,
    46
};

// encoder form of the bit-allocation parameter codes (ENC/ac3tab.h:143-165) and the rate tables
// (ENC/ac3tab.h:3-12); constexpr so that the kernels fold them and ac3mi_encode_spec_tables() hands
// out the very values the kernels use
constexpr int enc_sdecay(int cod) { return 15 + 2 * cod; }
constexpr int enc_fdecay(int cod) { return 63 + 20 * cod; }
constexpr int enc_fgain(int cod) { return 128 * (cod + 1); }
constexpr int enc_sgain(int cod) { return cod == 0 ? 0x540 : cod == 1 ? 0x4d8 : cod == 2 ? 0x478 : 0x410; }
constexpr int enc_dbknee(int cod) { return cod == 0 ? 0 : 0x500 + 0x200 * cod; }
constexpr int enc_floor(int cod) { return cod < 5 ? 0x2f0 - 0x40 * cod : cod == 5 ? 0x170 : cod == 6 ? 0x0f0 : 0xf800; }
static const int kSampleRates[3] = {48000, 44100, 32000};
static const int kKbps[19] = {32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640};

static inline void build_logadd(uint8_t *tab256)
{
    int n = 0;
    for (int k = 0; k <= 64; k++)
        for (int r = 0; r < kLaRun[k]; r++) tab256[n++] = (uint8_t)(64 - k);
}

}  // namespace ac3mi
