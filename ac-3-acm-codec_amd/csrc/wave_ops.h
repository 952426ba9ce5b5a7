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

// maximum of all lanes' non-negative values, wave-uniform
__device__ __forceinline__ int wave_max_nonneg(int v)
{
#define AC3MI_STEP(ctrl, rows, bc) { const int t = __builtin_amdgcn_update_dpp(0, v, ctrl, rows, 0xf, bc); v = t > v ? t : v; }
    AC3MI_STEP(0x111, 0xf, true) AC3MI_STEP(0x112, 0xf, true) AC3MI_STEP(0x114, 0xf, true) AC3MI_STEP(0x118, 0xf, true)
    AC3MI_STEP(0x142, 0xa, false) AC3MI_STEP(0x143, 0xc, false)
#undef AC3MI_STEP
    return __builtin_amdgcn_readlane(v, 63);
}

// prefix minimum / maximum (lanes shifted in from outside a row keep the identity)
__device__ __forceinline__ int wave_incl_scan_min(int v)
{
    constexpr int ID = 0x3fffffff;
#define AC3MI_STEP(ctrl, rows) { const int t = __builtin_amdgcn_update_dpp(ID, v, ctrl, rows, 0xf, false); v = t < v ? t : v; }
    AC3MI_STEP(0x111, 0xf) AC3MI_STEP(0x112, 0xf) AC3MI_STEP(0x114, 0xf) AC3MI_STEP(0x118, 0xf)
    AC3MI_STEP(0x142, 0xa) AC3MI_STEP(0x143, 0xc)
#undef AC3MI_STEP
    return v;
}
__device__ __forceinline__ int wave_incl_scan_max(int v)
{
    constexpr int ID = -0x3fffffff;
#define AC3MI_STEP(ctrl, rows) { const int t = __builtin_amdgcn_update_dpp(ID, v, ctrl, rows, 0xf, false); v = t > v ? t : v; }
    AC3MI_STEP(0x111, 0xf) AC3MI_STEP(0x112, 0xf) AC3MI_STEP(0x114, 0xf) AC3MI_STEP(0x118, 0xf)
    AC3MI_STEP(0x142, 0xa) AC3MI_STEP(0x143, 0xc)
#undef AC3MI_STEP
    return v;
}
// suffix minimum: lane i gets min over lanes i..63 (row_shl inside the rows, the row totals through scalars)
__device__ __forceinline__ int wave_suffix_scan_min(int v, int lane)
{
    constexpr int ID = 0x3fffffff;
#define AC3MI_STEP(ctrl) { const int t = __builtin_amdgcn_update_dpp(ID, v, ctrl, 0xf, 0xf, false); v = t < v ? t : v; }
    AC3MI_STEP(0x101) AC3MI_STEP(0x102) AC3MI_STEP(0x104) AC3MI_STEP(0x108)
#undef AC3MI_STEP
    const int t1 = __builtin_amdgcn_readlane(v, 16), t2 = __builtin_amdgcn_readlane(v, 32), t3 = __builtin_amdgcn_readlane(v, 48);
    const int a2 = t2 < t3 ? t2 : t3, a1 = t1 < a2 ? t1 : a2;
    const int r = lane >> 4;
    const int extra = r == 0 ? a1 : r == 1 ? a2 : r == 2 ? t3 : ID;
    return extra < v ? extra : v;
}

}  // namespace ac3mi
