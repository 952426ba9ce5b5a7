// a52_levels.h — output-mode negotiation and per-channel gains, host + device.
//
// Replaces a52_downmix_init (L52/downmix.c:34-160) and a52_downmix_coeff
// (L52/downmix.c:162-330).  Operand types (int / float / double) and evaluation
// order follow the reference so that the resulting floats are bit-identical; the
// translation units that include this header are built with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace ac3mi {

#define AC3MI_G_PLUS6DB 2.0
#define AC3MI_G_PLUS3DB 1.4142135623730951
#define AC3MI_G_3DB 0.7071067811865476
#define AC3MI_G_45DB 0.5946035575013605
#define AC3MI_G_6DB 0.5

#define AC3MI_PAIR(acmod, out) (((out) << 3) + (acmod))

// requested flags x coded mode -> granted output (L52/downmix.c:37-60)
__host__ __device__ inline int a52_granted_output(int req, int input)
{
    // row = requested output, column = input & 7
    const unsigned char grant[11][8] = {
        {0, 10, 2, 2, 2, 2, 2, 2},  {1, 1, 1, 1, 1, 1, 1, 1},  {0, 10, 2, 2, 2, 2, 2, 2}, {0, 10, 2, 3, 2, 3, 2, 3},
        {0, 10, 2, 2, 4, 4, 4, 4},  {0, 10, 2, 2, 4, 5, 4, 5}, {0, 10, 2, 3, 6, 6, 6, 6}, {0, 10, 2, 3, 6, 7, 6, 7},
        {8, 1, 1, 1, 1, 1, 1, 1},   {9, 1, 1, 1, 1, 1, 1, 1},  {0, 10, 2, 10, 10, 10, 10, 10}};
    if (req > 10) return -1;
    return grant[req][input & 7];
}

// input: coded acmod, or 10 (DOLBY) for acmod 2 with dsurmod 2.  Returns the output
// channel configuration (<0: unsupported request) and scales *level when
// ADJUST_LEVEL (32) is set in flags.
__host__ __device__ inline int a52_downmix_init_hd(int input, int flags, float *level, float clev, float slev)
{
    int out = a52_granted_output(flags & 15, input);
    float adj;
    if (out < 0) return -1;
    // float-vs-double comparison kept as in L52/downmix.c:68-70 (never true for a float clev)
    if (out == 2 && (input == 10 || (input == 3 && (double)clev == AC3MI_G_3DB))) out = 10;
    if (!(flags & 32)) return out;

    switch (AC3MI_PAIR(input & 7, out)) {
    case AC3MI_PAIR(3, 1): adj = AC3MI_G_3DB / (1 + clev); break;
    case AC3MI_PAIR(2, 1):
    case AC3MI_PAIR(6, 4):
    case AC3MI_PAIR(7, 5): adj = AC3MI_G_3DB; break;
    case AC3MI_PAIR(7, 4):
        if (clev < (AC3MI_G_PLUS3DB - 1)) { adj = AC3MI_G_3DB; break; }
        adj = 1 / (1 + clev);
        break;
    case AC3MI_PAIR(3, 2):
    case AC3MI_PAIR(5, 4):
    case AC3MI_PAIR(5, 6):
    case AC3MI_PAIR(7, 6): adj = 1 / (1 + clev); break;
    case AC3MI_PAIR(4, 1): adj = AC3MI_G_PLUS3DB / (2 + slev); break;
    case AC3MI_PAIR(4, 2):
    case AC3MI_PAIR(5, 3): adj = 1 / (1 + slev * AC3MI_G_3DB); break;
    case AC3MI_PAIR(5, 1): adj = AC3MI_G_3DB / (1 + clev + slev * 0.5); break;
    case AC3MI_PAIR(5, 2): adj = 1 / (1 + clev + slev * AC3MI_G_3DB); break;
    case AC3MI_PAIR(6, 1): adj = AC3MI_G_3DB / (1 + slev); break;
    case AC3MI_PAIR(6, 2):
    case AC3MI_PAIR(7, 3): adj = 1 / (1 + slev); break;
    case AC3MI_PAIR(7, 1): adj = AC3MI_G_3DB / (1 + clev + slev); break;
    case AC3MI_PAIR(7, 2): adj = 1 / (1 + clev + slev); break;
    case AC3MI_PAIR(1, 10): adj = AC3MI_G_PLUS3DB; break;
    case AC3MI_PAIR(3, 10):
    case AC3MI_PAIR(4, 10): adj = 1 / (1 + AC3MI_G_3DB); break;
    case AC3MI_PAIR(5, 10):
    case AC3MI_PAIR(6, 10): adj = 1 / (1 + 2 * AC3MI_G_3DB); break;
    case AC3MI_PAIR(7, 10): adj = 1 / (1 + 3 * AC3MI_G_3DB); break;
    default: return out;
    }
    *level = *level * adj;
    return out;
}

// gains of the coded channels for (acmod -> output); returns liba52's chanbias mask
__host__ __device__ inline int a52_downmix_coeff_hd(float *g, int acmod, int output, float level, float clev, float slev)
{
    float l3 = level * AC3MI_G_3DB;
    switch (AC3MI_PAIR(acmod, output & 15)) {
    case AC3MI_PAIR(0, 0): case AC3MI_PAIR(1, 1): case AC3MI_PAIR(2, 2): case AC3MI_PAIR(3, 3):
    case AC3MI_PAIR(4, 4): case AC3MI_PAIR(5, 5): case AC3MI_PAIR(6, 6): case AC3MI_PAIR(7, 7):
    case AC3MI_PAIR(2, 10):
        g[0] = g[1] = g[2] = g[3] = g[4] = level; return 0;
    case AC3MI_PAIR(0, 1): g[0] = g[1] = level * AC3MI_G_6DB; return 3;
    case AC3MI_PAIR(2, 1): g[0] = g[1] = l3; return 3;
    case AC3MI_PAIR(3, 1): g[0] = g[2] = l3; g[1] = (l3 * clev) * AC3MI_G_PLUS6DB; return 7;
    case AC3MI_PAIR(4, 1): g[0] = g[1] = l3; g[2] = l3 * slev; return 7;
    case AC3MI_PAIR(6, 1): g[0] = g[1] = l3; g[2] = g[3] = l3 * slev; return 15;
    case AC3MI_PAIR(5, 1): g[0] = g[2] = l3; g[1] = (l3 * clev) * AC3MI_G_PLUS6DB; g[3] = l3 * slev; return 15;
    case AC3MI_PAIR(7, 1): g[0] = g[2] = l3; g[1] = (l3 * clev) * AC3MI_G_PLUS6DB; g[3] = g[4] = l3 * slev; return 31;
    case AC3MI_PAIR(1, 10): g[0] = l3; return 0;
    case AC3MI_PAIR(3, 10): g[0] = g[2] = g[3] = g[4] = level; g[1] = l3; return 7;
    case AC3MI_PAIR(3, 2): case AC3MI_PAIR(5, 4): case AC3MI_PAIR(7, 6):
        g[0] = g[2] = g[3] = g[4] = level; g[1] = level * clev; return 7;
    case AC3MI_PAIR(4, 10): g[0] = g[1] = level; g[2] = l3; return 7;
    case AC3MI_PAIR(4, 2): g[0] = g[1] = level; g[2] = l3 * slev; return 7;
    case AC3MI_PAIR(5, 10): g[0] = g[2] = level; g[1] = g[3] = l3; return 15;
    case AC3MI_PAIR(5, 2): g[0] = g[2] = level; g[1] = level * clev; g[3] = l3 * slev; return 15;
    case AC3MI_PAIR(6, 10): g[0] = g[1] = level; g[2] = g[3] = l3; return 15;
    case AC3MI_PAIR(6, 2): g[0] = g[1] = level; g[2] = g[3] = level * slev; return 15;
    case AC3MI_PAIR(7, 10): g[0] = g[2] = level; g[1] = g[3] = g[4] = l3; return 31;
    case AC3MI_PAIR(7, 4): g[0] = g[2] = level; g[1] = level * clev; g[3] = g[4] = l3; return 31;
    case AC3MI_PAIR(7, 2): g[0] = g[2] = level; g[1] = level * clev; g[3] = g[4] = level * slev; return 31;
    case AC3MI_PAIR(5, 3): g[0] = g[1] = g[2] = level; g[3] = l3 * slev; return 13;
    case AC3MI_PAIR(7, 3): g[0] = g[1] = g[2] = level; g[3] = g[4] = level * slev; return 29;
    case AC3MI_PAIR(6, 4): g[0] = g[1] = level; g[2] = g[3] = l3; return 12;
    case AC3MI_PAIR(7, 5): g[0] = g[1] = g[2] = level; g[3] = g[4] = l3; return 24;
    case AC3MI_PAIR(4, 6): g[0] = g[1] = level; g[2] = l3; return 0;
    case AC3MI_PAIR(5, 6): g[0] = g[2] = level; g[1] = level * clev; g[3] = l3; return 7;
    case AC3MI_PAIR(5, 7): g[0] = g[1] = g[2] = level; g[3] = l3; return 0;
    case AC3MI_PAIR(0, 8): g[0] = level; g[1] = 0; return 0;
    case AC3MI_PAIR(0, 9): g[0] = 0; g[1] = level; return 0;
    }
    return -1;
}

}  // namespace ac3mi
