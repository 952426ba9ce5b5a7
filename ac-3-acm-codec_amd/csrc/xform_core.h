// xform_core.h — device-side core of the AC-3 synthesis transform for gfx950.
//
// One IMDCT (512- or 256-point window) is computed by an 8-lane group; a 64-wide
// wavefront therefore runs 8 independent transforms side by side.  Each lane keeps
// 16 complex points in registers:
//
//   long block  : 128-point complex DFT = 16 (in registers) x 8 (across the 8 lanes)
//   short block : 2 x 64-point complex DFT = 16 (in registers) x 4, two of them
//
// The only cross-lane traffic is one 8x16 transpose through LDS between the two
// stages plus one DPP half-mirror per input coefficient.  All twiddles that depend
// on a register index are compile-time constants; the lane-dependent ones are
// merged into a single 16-entry table per lane (tw_long / tw_short).
//
// Math (checked against liba52's a52_imdct_512/256, L52/imdct.c:258-345; see
// DESIGN.md §4): with X[256] the coefficient plane,
//   long : z[m]  = (-1)^m (X[2m] + j X[255-2m]) e^{-j pi (m+63.75)/256},  m < 128
//          Y[q]  = e^{-j pi (q+.5)/256} * sum_m z[m] e^{-j 2 pi m q/128}
//          first[2i] = -Re Y[i]  first[2i+1] =  Re Y[127-i]
//          tail [2i] = -Im Y[i]  tail [2i+1] = -Im Y[127-i]                i < 64
//   short: z1[m] = (X[4m]   + j X[254-4m]) e^{-j pi (m-.25)/128},          m < 64
//          z2[m] = (X[4m+1] + j X[255-4m]) e^{-j pi (m-.25)/128}
//          Y1,Y2 = e^{-j pi (q+.5)/128} * DFT64(z1,z2)
//          first[2i] = -Re Y1[i] first[2i+1] = Im Y1[63-i]
//          tail [2i] = -Im Y2[i] tail [2i+1] = Re Y2[63-i]                 i < 64
//   out[p]     =  first[p] w[p]     + prev_tail[p] w[255-p] + bias
//   out[255-p] = -first[p] w[255-p] + prev_tail[p] w[p]     + bias         p < 128
// (first is antisymmetric and tail symmetric about 127.5, which is why liba52 only
// keeps 128 live floats of its 256-float delay plane.)
//
// Lane L of a group ends up owning i in { L+16k, 15-L+16k : k<4 } and hence the
// output pairs (2i,2i+1) and (254-2i,255-2i): 8-byte stores, 64 contiguous bytes
// per group per instruction.
#pragma once
#include <hip/hip_runtime.h>

namespace ac3mi {

struct cf { float re, im; };

__device__ __forceinline__ cf operator+(cf a, cf b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cf operator-(cf a, cf b) { return {a.re - b.re, a.im - b.im}; }
// Fused multiply-adds are written out (and the translation units that include this header are built with
// -ffp-contract=off), so that every kernel that inlines this arithmetic - the transform kernel of xform.hip and the fused
// decoder of decode_wg.hip, in all their instantiations - produces the same bits from the same coefficients.
__device__ __forceinline__ cf cmul(cf a, float cr, float ci)
{
    return {__builtin_fmaf(a.re, cr, -(a.im * ci)), __builtin_fmaf(a.re, ci, a.im * cr)};
}
__device__ __forceinline__ cf cmul(cf a, cf c) { return cmul(a, c.re, c.im); }

// e^{-j pi n/32}, n = 0..15
__device__ constexpr float C32_RE[16] = {
    1.000000000e+00f, 9.951847267e-01f, 9.807852804e-01f, 9.569403357e-01f, 9.238795325e-01f,
    8.819212643e-01f, 8.314696123e-01f, 7.730104534e-01f, 7.071067812e-01f, 6.343932842e-01f,
    5.555702330e-01f, 4.713967368e-01f, 3.826834324e-01f, 2.902846773e-01f, 1.950903220e-01f,
    9.801714033e-02f};
__device__ constexpr float C32_IM[16] = {
    -0.000000000e+00f, -9.801714033e-02f, -1.950903220e-01f, -2.902846773e-01f, -3.826834324e-01f,
    -4.713967368e-01f, -5.555702330e-01f, -6.343932842e-01f, -7.071067812e-01f, -7.730104534e-01f,
    -8.314696123e-01f, -8.819212643e-01f, -9.238795325e-01f, -9.569403357e-01f, -9.807852804e-01f,
    -9.951847267e-01f};

constexpr float K_C1 = 9.238795325e-01f;   // cos(pi/8)
constexpr float K_S1 = 3.826834324e-01f;   // sin(pi/8)
constexpr float K_R2 = 7.071067812e-01f;   // sqrt(1/2)

// forward 4-point DFT, in place, natural order
__device__ __forceinline__ void dft4(cf &a, cf &b, cf &c, cf &d)
{
    cf s0 = a + c, s1 = a - c, s2 = b + d, s3 = b - d;
    a = s0 + s2;
    c = s0 - s2;
    b = {s1.re + s3.im, s1.im - s3.re};     // s1 - j s3
    d = {s1.re - s3.im, s1.im + s3.re};     // s1 + j s3
}

// forward 16-point DFT of v[0..15] (e^{-j 2 pi nk/16}), result in natural order
__device__ __forceinline__ void dft16(cf (&v)[16])
{
#pragma unroll
    for (int n2 = 0; n2 < 4; n2++) dft4(v[n2], v[4 + n2], v[8 + n2], v[12 + n2]);
    // v[4*k1+n2] *= W16^(n2*k1)
    v[5] = cmul(v[5], K_C1, -K_S1);                                  // W^1
    v[6] = {(v[6].re + v[6].im) * K_R2, (v[6].im - v[6].re) * K_R2}; // W^2
    v[7] = cmul(v[7], K_S1, -K_C1);                                  // W^3
    v[9] = {(v[9].re + v[9].im) * K_R2, (v[9].im - v[9].re) * K_R2}; // W^2
    v[10] = {v[10].im, -v[10].re};                                   // W^4 = -j
    v[11] = {(v[11].im - v[11].re) * K_R2, -(v[11].re + v[11].im) * K_R2}; // W^6
    v[13] = cmul(v[13], K_S1, -K_C1);                                // W^3
    v[14] = {(v[14].im - v[14].re) * K_R2, -(v[14].re + v[14].im) * K_R2}; // W^6
    v[15] = cmul(v[15], -K_C1, K_S1);                                // W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; k1++) dft4(v[4 * k1], v[4 * k1 + 1], v[4 * k1 + 2], v[4 * k1 + 3]);
    // now v[4*k1+k2] = X[k1+4*k2]: transpose the 4x4 index
    cf t;
    t = v[1]; v[1] = v[4]; v[4] = t;
    t = v[2]; v[2] = v[8]; v[8] = t;
    t = v[3]; v[3] = v[12]; v[12] = t;
    t = v[6]; v[6] = v[9]; v[9] = t;
    t = v[7]; v[7] = v[13]; v[13] = t;
    t = v[11]; v[11] = v[14]; v[14] = t;
}

// forward 8-point DFT of a[0..7], result in natural order
__device__ __forceinline__ void dft8(cf &a0, cf &a1, cf &a2, cf &a3, cf &a4, cf &a5, cf &a6, cf &a7)
{
    // n = 2*n1+n2: 4-point DFTs over n1 for n2 = 0 (even) and n2 = 1 (odd)
    dft4(a0, a2, a4, a6);       // a0,a2,a4,a6 = E[0..3]
    dft4(a1, a3, a5, a7);       // a1,a3,a5,a7 = O[0..3]
    // O[k] *= W8^k
    a3 = {(a3.re + a3.im) * K_R2, (a3.im - a3.re) * K_R2};
    a5 = {a5.im, -a5.re};
    a7 = {(a7.im - a7.re) * K_R2, -(a7.re + a7.im) * K_R2};
    cf x0 = a0 + a1, x4 = a0 - a1, x1 = a2 + a3, x5 = a2 - a3;
    cf x2 = a4 + a5, x6 = a4 - a5, x3 = a6 + a7, x7 = a6 - a7;
    a0 = x0; a1 = x1; a2 = x2; a3 = x3; a4 = x4; a5 = x5; a6 = x6; a7 = x7;
}

// lane <-> 7-lane inside every group of 8 lanes (DPP row_half_mirror)
__device__ __forceinline__ float mirror8(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
}

// LDS staging of the 8x16 transpose: row = producing lane, 16 complex + 2 pad
constexpr int EX_ROW = 18;                       // float2 per row (144 B)
constexpr int EX_GROUP = 8 * EX_ROW + 8;         // float2 per 8-lane group (1216 B)
constexpr int EX_WAVE = 8 * EX_GROUP;            // float2 per wavefront

// v[k1] (k1 = 0..15) held by lane `l8` of the group  ->  r[0..7] = column l8,
// r[8..15] = column 15-l8 of the 8x16 matrix [lane][k1].  `ex` = this group's region.
__device__ __forceinline__ void transpose_8x16_put(float2 *ex, int l8, const cf (&v)[16])
{
    float4 *row = reinterpret_cast<float4 *>(ex + l8 * EX_ROW);
#pragma unroll
    for (int j = 0; j < 8; j++) row[j] = make_float4(v[2 * j].re, v[2 * j].im, v[2 * j + 1].re, v[2 * j + 1].im);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void transpose_8x16_get(const float2 *ex, int l8, cf (&r)[16])
{
#pragma unroll
    for (int l = 0; l < 8; l++) {
        float2 a = ex[l * EX_ROW + l8];
        float2 b = ex[l * EX_ROW + 15 - l8];
        r[l] = {a.x, a.y};
        r[8 + l] = {b.x, b.y};
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void transpose_8x16(float2 *ex, int l8, const cf (&v)[16], cf (&r)[16])
{
    transpose_8x16_put(ex, l8, v);
    transpose_8x16_get(ex, l8, r);
}

// e^{-j pi k/16}, k = 0..7 and e^{-j pi k/8}, k = 0..3
__device__ constexpr float P16_RE[8] = {1.000000000e+00f, 9.807852804e-01f, 9.238795325e-01f, 8.314696123e-01f,
                                        7.071067812e-01f, 5.555702330e-01f, 3.826834324e-01f, 1.950903220e-01f};
__device__ constexpr float P16_IM[8] = {-0.000000000e+00f, -1.950903220e-01f, -3.826834324e-01f, -5.555702330e-01f,
                                        -7.071067812e-01f, -8.314696123e-01f, -9.238795325e-01f, -9.807852804e-01f};
__device__ constexpr float P8_RE[4] = {1.000000000e+00f, 9.238795325e-01f, 7.071067812e-01f, 3.826834324e-01f};
__device__ constexpr float P8_IM[4] = {-0.000000000e+00f, -3.826834324e-01f, -7.071067812e-01f, -9.238795325e-01f};

// Per-lane result of one block: slot j = 2k   -> i = l8 + 16k
//                               slot j = 2k+1 -> i = 15 - l8 + 16k
// f0/t0 belong to p = 2i, f1/t1 to p = 2i+1.
struct FirstTail {
    float f0[8], f1[8], t0[8], t1[8];
};

__device__ __forceinline__ cf tw_at(const cf (&tw)[16], int k) { return tw[k]; }
__device__ __forceinline__ cf tw_at(const float2 *tw, int k) { float2 t = tw[k]; return {t.x, t.y}; }

// Long block.  xa[n1] = X[2m], xb[n1] = X[255-2m], m = 8*n1 + l8.  tw = this lane's
// 16 merged twiddles.  Accumulates into ft.
// The two halves of a transform (the fused decoder, decode_wg.hip, puts a workgroup barrier between them):
// first half = pre-twiddle, DFT-16, lane twiddles, 8x16 transpose; common to the long and the short block.
template <typename TW>
__device__ __forceinline__ void imdct_first_half_put(const float (&xa)[16], const float (&xb)[16], const TW &tw, float2 *ex, int l8)
{
    cf v[16];
#pragma unroll
    for (int n = 0; n < 16; n++) v[n] = cmul(cf{xa[n], xb[n]}, C32_RE[n], C32_IM[n]);
    dft16(v);
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = cmul(v[k], tw_at(tw, k));
    transpose_8x16_put(ex, l8, v);
}
template <typename TW>
__device__ __forceinline__ void imdct_first_half(const float (&xa)[16], const float (&xb)[16], const TW &tw,
                                                 float2 *ex, int l8, cf (&r)[16])
{
    imdct_first_half_put(xa, xb, tw, ex, l8);
    transpose_8x16_get(ex, l8, r);
}

// window + overlap-add of one output pair position (both halves of the block) - the same expression in every kernel:
//   out[p] = first[p] w[p] + prev_tail[p] w[255-p] + bias,  out[255-p] = -first[p] w[255-p] + prev_tail[p] w[p] + bias
__device__ __forceinline__ void window_pair(float f0, float f1, float2 d, float2 wlo, float2 whi, float bias, float2 &lo, float2 &hi)
{
    lo.x = __builtin_fmaf(f0, wlo.x, __builtin_fmaf(d.x, whi.y, bias));        // out[2i]
    lo.y = __builtin_fmaf(f1, wlo.y, __builtin_fmaf(d.y, whi.x, bias));        // out[2i+1]
    hi.x = __builtin_fmaf(-f1, whi.x, __builtin_fmaf(d.y, wlo.y, bias));       // out[254-2i]
    hi.y = __builtin_fmaf(-f0, whi.y, __builtin_fmaf(d.x, wlo.x, bias));       // out[255-2i]
}

// what the reference's converters make of a float sample at bias 384 (src/AC3ASM.asm:303-318: psubd, packssdw)
__device__ __forceinline__ int16_t to_s16(float v)
{
    int i = (int)(__float_as_uint(v) - 0x43c00000u);
    i = i > 32767 ? 32767 : i < -32768 ? -32768 : i;
    return (int16_t)i;
}

__device__ __forceinline__ void imdct_long_second_half(cf (&r)[16], FirstTail &ft);
__device__ __forceinline__ void imdct_short_second_half(cf (&r)[16], FirstTail &ft);

template <typename TW>
__device__ __forceinline__ void imdct_long(const float (&xa)[16], const float (&xb)[16], const TW &tw,
                                           float2 *ex, int l8, FirstTail &ft)
{
    cf r[16];
    imdct_first_half(xa, xb, tw, ex, l8, r);
    imdct_long_second_half(r, ft);
}

__device__ __forceinline__ void imdct_long_second_half(cf (&r)[16], FirstTail &ft)
{
    dft8(r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
    dft8(r[8], r[9], r[10], r[11], r[12], r[13], r[14], r[15]);
#pragma unroll
    for (int k = 1; k < 8; k++) {
        r[k] = cmul(r[k], P16_RE[k], P16_IM[k]);
        r[8 + k] = cmul(r[8 + k], P16_RE[k], P16_IM[k]);
    }
    // r[k2] = Y[l8+16 k2], r[8+k2] = Y[15-l8+16 k2]
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ft.f0[2 * k] += -r[k].re;
        ft.t0[2 * k] += -r[k].im;
        ft.f1[2 * k] += r[15 - k].re;
        ft.t1[2 * k] += -r[15 - k].im;
        ft.f0[2 * k + 1] += -r[8 + k].re;
        ft.t0[2 * k + 1] += -r[8 + k].im;
        ft.f1[2 * k + 1] += r[7 - k].re;
        ft.t1[2 * k + 1] += -r[7 - k].im;
    }
}

// Short block.  Lane l8 = 4*f + n2 feeds DFT f (0: even, 1: odd coefficients):
// xa[n1] = X[16 n1 + 4 n2 + f], xb[n1] = X[254 + f - 16 n1 - 4 n2].
template <typename TW>
__device__ __forceinline__ void imdct_short(const float (&xa)[16], const float (&xb)[16], const TW &tw,
                                            float2 *ex, int l8, FirstTail &ft)
{
    cf r[16];
    imdct_first_half(xa, xb, tw, ex, l8, r);
    imdct_short_second_half(r, ft);
}

__device__ __forceinline__ void imdct_short_second_half(cf (&r)[16], FirstTail &ft)
{
    dft4(r[0], r[1], r[2], r[3]);       // Y1[l8+16 k2]
    dft4(r[4], r[5], r[6], r[7]);       // Y2[l8+16 k2]
    dft4(r[8], r[9], r[10], r[11]);     // Y1[15-l8+16 k2]
    dft4(r[12], r[13], r[14], r[15]);   // Y2[15-l8+16 k2]
#pragma unroll
    for (int k = 1; k < 4; k++) {
        r[k] = cmul(r[k], P8_RE[k], P8_IM[k]);
        r[4 + k] = cmul(r[4 + k], P8_RE[k], P8_IM[k]);
        r[8 + k] = cmul(r[8 + k], P8_RE[k], P8_IM[k]);
        r[12 + k] = cmul(r[12 + k], P8_RE[k], P8_IM[k]);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ft.f0[2 * k] += -r[k].re;
        ft.t0[2 * k] += -r[4 + k].im;
        ft.f1[2 * k] += r[11 - k].im;
        ft.t1[2 * k] += r[15 - k].re;
        ft.f0[2 * k + 1] += -r[8 + k].re;
        ft.t0[2 * k + 1] += -r[12 + k].im;
        ft.f1[2 * k + 1] += r[3 - k].im;
        ft.t1[2 * k + 1] += r[7 - k].re;
    }
}

}  // namespace ac3mi
