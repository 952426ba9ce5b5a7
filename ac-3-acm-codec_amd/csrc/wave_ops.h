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

}  // namespace ac3mi
