// wave_ops.h — wavefront (64-lane) prefix sums and reductions on the gfx950 DPP network.
// No LDS traffic: four row_shr steps inside each row of 16 lanes, then row_bcast:15 carries the
// totals of rows 0 and 2 into rows 1 and 3, then row_bcast:31 carries the lower half into the upper.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ac3mi {

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);    // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// lane 63's value as a wave-uniform scalar
__device__ __forceinline__ uint32_t wave_last(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return wave_last(wave_incl_scan_u32(v)); }

// bitwise OR of all lanes, wave-uniform
__device__ __forceinline__ uint32_t wave_or_u32(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return wave_last(v);
}

// a * b on the low 24 bits of both, as v_mul_u32_u24 and nothing else: where only the low bits of `__umul24(x & 0xffff, w)`
// are used, hipcc drops the mask (the low bits of a product need only the low bits of the factors), then no longer knows the
// factor is short and emits v_mul_lo_u32 - a quarter of the rate.
__device__ __forceinline__ uint32_t mul24_asm(uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Prefix minimum / maximum.  One `v_max_i32_dpp v, v, v` per step: a lane whose source lies outside its row (or whose row is
// masked out) is left unwritten, which IS the identity - no identity register, no separate move.  hipcc does not form this
// from __builtin_amdgcn_update_dpp + max (it emits v_mov_b32 identity, s_nop, v_mov_b32_dpp, v_max: four instructions per
// step), so the steps are spelled out; `s_nop 1` = the two wait states a DPP read needs behind the VALU write of its source
// (gfx9 data hazard), which the compiler cannot place inside an asm block - nor behind it, hence the trailing one.
#define AC3MI_DPP6(op) \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" \
    "s_nop 1"
// two independent scans interleaved: each fills one of the other's wait states
#define AC3MI_DPP6x2(op) \
    "s_nop 1\n\t" \
    op " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t" op " %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t" \
    op " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t" op " %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t" \
    op " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t" op " %1, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t" \
    op " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t" op " %1, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 0\n\t" \
    op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t" op " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0\n\t" \
    op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t" op " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"

__device__ __forceinline__ int wave_incl_scan_min(int v)
{
    asm volatile(AC3MI_DPP6("v_min_i32_dpp") : "+v"(v));
    return v;
}
__device__ __forceinline__ int wave_incl_scan_max(int v)
{
    asm volatile(AC3MI_DPP6("v_max_i32_dpp") : "+v"(v));
    return v;
}
__device__ __forceinline__ void wave_incl_scan_min2(int &a, int &b)
{
    asm volatile(AC3MI_DPP6x2("v_min_i32_dpp") : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void wave_incl_scan_max2(int &a, int &b)
{
    asm volatile(AC3MI_DPP6x2("v_max_i32_dpp") : "+v"(a), "+v"(b));
}

// maximum of all lanes' non-negative values, wave-uniform
__device__ __forceinline__ int wave_max_nonneg(int v) { return __builtin_amdgcn_readlane(wave_incl_scan_max(v), 63); }

// suffix minimum: lane i gets min over lanes i..63 (row_shl inside the rows, the row totals through scalars)
__device__ __forceinline__ int wave_suffix_scan_min(int v, int lane)
{
    constexpr int ID = 0x3fffffff;
    asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shl:1 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shl:2 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_min_i32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    const int t1 = __builtin_amdgcn_readlane(v, 16), t2 = __builtin_amdgcn_readlane(v, 32), t3 = __builtin_amdgcn_readlane(v, 48);
    const int a2 = t2 < t3 ? t2 : t3, a1 = t1 < a2 ? t1 : a2;
    const int r = lane >> 4;
    const int extra = r == 0 ? a1 : r == 1 ? a2 : r == 2 ? t3 : ID;
    return extra < v ? extra : v;
}

}  // namespace ac3mi
