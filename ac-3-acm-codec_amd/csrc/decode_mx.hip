// decode_mx.hip — mantx_kernel: the mantissa half of the split front end with the synthesis transform fused in, for
// batches of one-frame streams whose coded planes are the output planes (no downmix).  Replaces mant_kernel + xform_kernel
// for those calls: the dequantised coefficient planes (L52/parse.c:813-879) never touch HBM - 36.9 KB written and read again
// per 5.1 frame otherwise - they go from the unpacking wavefront's LDS straight into a52_imdct_512 / a52_imdct_256
// (L52/imdct.c:258-345), window, overlap-add and, for the s16 variant, the reference's float -> s16 converter
// (src/AC3ASM.asm:174-318).
//
// A workgroup of six wavefronts takes one frame (as mant_kernel): wavefront b unpacks audio block b into six padded planes
// in its own LDS region (mant_block2, mant2.h - the same code, the planes' address is the only difference), then runs the
// transform of xform_core.h on them, one plane per 8-lane group - exactly the arithmetic of xform_kernel, so PCM and overlap
// state are bit-identical to the two-kernel path (tests/test_decode_gpu.py).  Block b's overlap tail goes to block b + 1
// through LDS, a flag per wavefront instead of a workgroup barrier; block 0 takes the stream's state from HBM, block 5 leaves the new state.
//
// LDS per wavefront (MX_WAVE = 7296 bytes, used in turn): planes 6 x 272 floats + the code ring of the mantissa stage |
// the six 8x16 transposes | tails (3 KB, read by the next wavefront) + the s16 tile (3 KB).
#include <stdlib.h>
#include "mant2.h"
#include "xform_core.h"
#include "ac3mi_internal.h"

namespace ac3mi {

constexpr int MX_PLANE = 256 + 16;              // floats from plane to plane: the six groups' loads fall on different banks
constexpr int MX_WAVE = 6 * EX_GROUP * 8;       // bytes per wavefront region
constexpr int MX_RING = 6 * MX_PLANE * 4;       // byte offset of the mantissa stage's code ring inside the region
constexpr int MX_TILE = 3072;                   // byte offset of the s16 tile (the tails come first)
static_assert(MX_RING + M2_LDS_WAVE <= MX_WAVE && MX_TILE + 256 * 6 * 2 <= MX_WAVE && 48 * 8 * 8 <= MX_TILE, "region layout");

struct MantxParams {
    MantParams m;
    const uint8_t *blksw;       // [frames][6][nfchans] from the parse kernel
    float *delay;               // overlap state, [streams or slots][delay_stride]
    const int32_t *slot;
    int delay_stride;
    float *pcm;                 // [frames][6][n_in][256], or
    int16_t *pcm16;             // [frames][6][256][n_in] interleaved, plane o in WAVE slot wslot[o]
    int8_t wslot[6];
    const float2 *tw_long, *tw_short;
    const float *window;
    float bias;
};

struct MantxLDS {
    uint4 dsc[M2_NDESC];
    float qtab[760];
    float2 tw[2][128];          // merged lane twiddles [long, short][lane][k]
    float win[256];
    uint8_t cplbnd[6][20];
    uint32_t ready[6];          // wavefront b's tails are in its region
    alignas(16) uint8_t wave[6][MX_WAVE];
};

#ifndef MANTX_LB
#define MANTX_LB 3
#endif
// Measurement aid (make EXTRA=-DMX_STAMPS, a separate library): lane 0 of every wavefront adds the s_memtime ticks of its
// sections to g_mx_cycles[block][section]: 0 staging up to the barrier, 1 mantissas, 2 transform, 3 waiting for the block
// before, 4 window + output.  ac3mi_debug_mx_cycles reads them (profiles/mx_stamps.py).
#ifdef MX_STAMPS
__device__ unsigned long long g_mx_cycles[6][8];
#define MX_T0() unsigned long long mx_t = __builtin_readcyclecounter()
#define MX_LAP(id) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_mx_cycles[blk][id], t_ - mx_t); mx_t = t_; } while (0)
#else
#define MX_T0() do { } while (0)
#define MX_LAP(id) do { } while (0)
#endif
template <bool S16>
__global__ __launch_bounds__(384, MANTX_LB) void mantx_kernel(const MantxParams Q)
{
    const MantParams &P = Q.m;
    __shared__ MantxLDS L;
    extern __shared__ uint32_t frw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int blk = __builtin_amdgcn_readfirstlane(tid >> 6);
    MX_T0();
    unsigned fidx;                                              // consecutive frames on one XCD, as mant_kernel
    {
        const unsigned n = gridDim.x, q = n >> 3, r = n & 7u, x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        fidx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const size_t unit = (size_t)fidx * 6 + blk;
    const uint4 *dq = reinterpret_cast<const uint4 *>(P.desc + unit);
    const uint4 w0v = dq[0], w1v = dq[1], w2v = dq[2], w3v = dq[3], w4v = dq[4];
    const uint32_t fposv = P.frame_pos[fidx];
    const int nw = (P.frame_bytes + 3) >> 2;
    {
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(P.frames + (size_t)fidx * P.frame_stride);
        for (int i = tid; i < nw + 6; i += 384) {
            uint32_t v = 0;
            if (i < nw) {
                v = s32[i];
                const int rem = P.frame_bytes - 4 * i;
                if (rem < 4) v &= (1u << (8 * rem)) - 1u;
                v = __builtin_bswap32(v);
            }
            frw[i] = v;
        }
        if (tid < M2_NDESC) L.dsc[tid] = mant_desc2((uint32_t)tid);
        for (int i = tid; i < 760; i += 384) L.qtab[i] = P.tab->qtab[i];
        if (tid < 128) L.tw[0][tid] = Q.tw_long[tid];
        else if (tid < 256) L.tw[1][tid - 128] = Q.tw_short[tid - 128];
        else L.win[tid - 256] = Q.window[tid - 256];
        if (tid < 128) L.win[128 + tid] = Q.window[128 + tid];
        if (tid < 6) L.ready[tid] = 0u;
    }
    uint8_t *const region = L.wave[blk];
    float *const planes = reinterpret_cast<float *>(region);
    const uint32_t flags = rfl(w0v.z);
    const bool failed = (flags & 1u) != 0u;
    MantBlk B;
    B.nf = P.nfchans; B.lfeon = P.lfeon; B.acmod = P.acmod; B.in_lfe = P.lfeon ? 1 : 0;
    B.chincpl = (int)((flags >> 8) & 31u); B.dithmask = (int)((flags >> 16) & 31u); B.rematflg = (int)((flags >> 24) & 15u);
    const uint64_t rve = (uint64_t)rfl(w2v.x) | ((uint64_t)rfl(w2v.y) << 32), rvb = (uint64_t)rfl(w2v.z) | ((uint64_t)rfl(w2v.w) << 32);
    const uint8_t *rowbase = P.rows + (size_t)fidx * 6 * ROWSET;
    auto fetch = [&](int slot) -> uint2 {
        const uint8_t *er = rowbase + (size_t)((rve >> (8 * slot)) & 7u) * ROWSET + slot * 512;
        const uint8_t *br = rowbase + (size_t)((rvb >> (8 * slot)) & 7u) * ROWSET + slot * 512 + 256;
        if (slot == 5) return lane < 7 ? make_uint2(br[lane], er[lane]) : make_uint2(1u, 0u);
        return make_uint2(reinterpret_cast<const uint32_t *>(br)[lane], reinterpret_cast<const uint32_t *>(er)[lane]);
    };
    const int slot0 = seg_slot(0, B.nf, B.chincpl, B.chincpl ? __builtin_ctz(B.chincpl) : 99);
    uint2 first = make_uint2(0u, 0u);
    if (!failed) first = fetch(slot0);
    __syncthreads();
    MX_LAP(0);

    // ---- mantissas -> planes in LDS (a failed block leaves zero planes) ----
    if (failed) {
        for (int c = 0; c < P.n_in; c++)
            *reinterpret_cast<float4 *>(planes + c * MX_PLANE + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        {
            const uint32_t a = rfl(w1v.x), b = rfl(w1v.y), c = rfl(w1v.z), d = rfl(w1v.w);
            B.cplstrtmant = (int)(c >> 16); B.cplendmant = (int)(d & 0xffffu);
            B.ends = (uint64_t)((a & 0xffu) | ((a >> 8) & 0xff00u) | ((b & 0xffu) << 16) | ((b << 8) & 0xff000000u)) | ((uint64_t)(c & 0xffu) << 32);
        }
        // the gains, lane k = slot k's (lane 5: the LFE's): one v_readlane per segment in mant_block2
        B.gainv = lane == 0 ? __uint_as_float(w3v.x) : lane == 1 ? __uint_as_float(w3v.y) : lane == 2 ? __uint_as_float(w3v.z)
                : lane == 3 ? __uint_as_float(w3v.w) : lane == 4 ? __uint_as_float(w4v.x) : __uint_as_float(w4v.y);
        if (B.chincpl && lane < 18) {                               // sub-band -> band (parse.c:448-456)
            const uint32_t below = rfl(w0v.w) & ((1u << lane) - 1u);
            L.cplbnd[blk][lane] = (uint8_t)(lane - __popc(below));
        }
        const uint32_t fpos = rfl(fposv);
        const bool lfsr_live = fpos != 0xffffffffu;
        const uint32_t i0 = lfsr_live ? (fpos + rfl(w0v.y)) % 65535u : 0u;
        const float *cc = P.cplco + unit * 90;
        mant_block2<MX_PLANE, true>(B, fetch, first, [&](int c, int bnd) { return cc[c * 18 + bnd]; }, L.cplbnd[blk], L.dsc, region + MX_RING, frw,
                              (uint32_t)nw + 2u, L.qtab, reinterpret_cast<const int16_t *>(P.lfsr_seq) + 1 + i0, lfsr_live, planes, rfl(w0v.x), lane);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    MX_LAP(1);
    // ---- transform: 8-lane group g takes plane g (left-over groups shadow plane 0 and store nothing) ----
    const int l8 = lane & 7, grp = lane >> 3;
    const bool own = grp < P.n_in;
    const int o = own ? grp : 0;
    const int fb = o - B.in_lfe;
    const bool sw = fb >= 0 && Q.blksw[unit * P.nfchans + fb] != 0;
    const size_t sidx = Q.slot ? (size_t)Q.slot[fidx] : (size_t)fidx;      // one frame per stream: stream = frame
    float *const dptr = Q.delay + sidx * Q.delay_stride + (size_t)o * 128;
    float2 dl[8];
    if (blk == 0) {                                                 // the stream's overlap state, in flight during the transform
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
            dl[j] = *reinterpret_cast<const float2 *>(dptr + 2 * i);
        }
    }
    FirstTail ft;
#pragma unroll
    for (int j = 0; j < 8; j++) ft.f0[j] = ft.f1[j] = ft.t0[j] = ft.t1[j] = 0.f;
    {
        // the loads of xform.hip's load_long / load_short on the LDS plane; every group's loads come before any group's
        // transpose (the transposes overwrite the planes)
        const float *plane = planes + o * MX_PLANE;
        float xa[16], xb[16];
        if (!sw) {
            const float2 *p = reinterpret_cast<const float2 *>(plane) + l8;
            float2 v[16];
#pragma unroll
            for (int n = 0; n < 16; n++) v[n] = p[8 * n];
#pragma unroll
            for (int n = 0; n < 16; n++) {
                xa[n] = __builtin_fmaf(1.f, v[n].x, 0.f);
                xb[n] = __builtin_fmaf(1.f, mirror8(v[15 - n].y), 0.f);
            }
        } else {
            const int f = l8 >> 2, n2 = l8 & 3;
            const float *pa = plane + 4 * n2 + f;
            const float *pb = plane + 254 + f - 4 * n2;
#pragma unroll
            for (int n = 0; n < 16; n++) {
                xa[n] = __builtin_fmaf(1.f, pa[16 * n], 0.f);
                xb[n] = __builtin_fmaf(1.f, pb[-16 * n], 0.f);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float2 *ex = reinterpret_cast<float2 *>(region) + o * EX_GROUP;
        const float2 *tw = &L.tw[sw ? 1 : 0][l8 * 16];
        cf r[16];
        imdct_first_half(xa, xb, tw, ex, l8, r);
        if (!sw) imdct_long_second_half(r, ft);
        else imdct_short_second_half(r, ft);
    }
    // ---- tails to the next block's wavefront ----
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (blk < 5 && lane < 48) {
        float2 *tl = reinterpret_cast<float2 *>(region);
#pragma unroll
        for (int j = 0; j < 8; j++) tl[j * 48 + lane] = make_float2(ft.t0[j], ft.t1[j]);
    }
    // Wavefront b waits for wavefront b - 1 only (a workgroup barrier here would hold all six until the slowest block is
    // unpacked and transformed).  The six are resident together and none leaves before it has published, so the wait ends;
    // the bound on the spin is a backstop, not an exit anyone takes.  Wavefront 0 publishes once the stream's old state has
    // ARRIVED in its registers; wavefront 5 stores the new one only after it has seen that flag as well.
    MX_LAP(2);
    if (blk == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the old state has arrived, not merely been requested)
    if (blk < 5) __hip_atomic_store(&L.ready[blk], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (blk > 0) {
        for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(&L.ready[blk - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u; spin++)
            __builtin_amdgcn_s_sleep(1);
        const float2 *tl = reinterpret_cast<const float2 *>(L.wave[blk - 1]);
        const int ls = lane < 48 ? lane : l8;                       // (shadow groups read group 0's)
#pragma unroll
        for (int j = 0; j < 8; j++) dl[j] = tl[j * 48 + ls];
    }
    if (blk == 5) {
        for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(&L.ready[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0u; spin++)
            __builtin_amdgcn_s_sleep(1);
    }
    if (blk == 5 && own) {                                          // the last block's tails are the stream's new state
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
            *reinterpret_cast<float2 *>(dptr + 2 * i) = make_float2(ft.t0[j], ft.t1[j]);
        }
    }

    MX_LAP(3);
    // ---- window + overlap-add + bias (xform.hip's output stage) ----
    int16_t *tile = reinterpret_cast<int16_t *>(region + MX_TILE);
    int wsl = 0;
#pragma unroll
    for (int oo = 0; oo < 6; oo++) wsl = oo == o ? Q.wslot[oo] : wsl;
    float *oblk = S16 ? nullptr : Q.pcm + (unit * P.n_in + o) * 256;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
        const float2 wlo = *reinterpret_cast<const float2 *>(&L.win[2 * i]);
        const float2 whi = *reinterpret_cast<const float2 *>(&L.win[254 - 2 * i]);
        float2 lo, hi;
        window_pair(ft.f0[j], ft.f1[j], dl[j], wlo, whi, Q.bias, lo, hi);
        if (!S16 && own) {
            *reinterpret_cast<float2 *>(oblk + 2 * i) = lo;
            *reinterpret_cast<float2 *>(oblk + 254 - 2 * i) = hi;
        }
        if (S16 && own) {                                           // (24-bit multiplies: v_mul_lo_u32 runs at a quarter of the rate)
            const int at = wsl + (int)__umul24((uint32_t)(2 * i), (uint32_t)P.n_in), top = wsl + (int)__umul24(254u, (uint32_t)P.n_in) - (at - wsl);
            tile[at] = to_s16(lo.x);
            tile[at + P.n_in] = to_s16(lo.y);
            tile[top] = to_s16(hi.x);
            tile[top + P.n_in] = to_s16(hi.y);
        }
    }
    if (S16) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int upb = 32 * P.n_in;                                // 16-byte units of the block
        int16_t *dst = Q.pcm16 + unit * 256 * P.n_in;
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int u = it * 64 + lane;
            if (u < upb) *reinterpret_cast<uint4 *>(dst + u * 8) = *reinterpret_cast<const uint4 *>(tile + u * 8);
        }
    }
    MX_LAP(4);
}

// mant_kernel + xform_kernel in one launch (launch_decode calls this when DecodeLaunch::fuse is set)
hipError_t launch_mantx(const DeviceTables &tab, const DecodeLaunch &L, const MantParams &M, hipStream_t stream)
{
    const XformLaunch &X = *L.fuse;
    if (L.frames_per_stream != 1 || X.plan.n_in != X.plan.n_out || M.n_in > 6) return hipErrorInvalidValue;
    MantxParams Q;
    Q.m = M;
    Q.m.coef = nullptr;
    Q.blksw = L.blksw;
    Q.delay = X.delay;
    Q.slot = X.slot;
    Q.delay_stride = X.slot ? X.delay_stride : X.plan.n_out * 128;
    Q.pcm = X.pcm;
    Q.pcm16 = X.pcm16;
    Q.tw_long = tab.tw_long;
    Q.tw_short = tab.tw_short;
    Q.window = tab.window;
    Q.bias = X.bias;
    for (int o = 0; o < 6; o++) Q.wslot[o] = 0;
    static const int lds_pad = getenv("AC3MI_MANTX_LDS_PAD") ? atoi(getenv("AC3MI_MANTX_LDS_PAD")) : 0;      // profiling aid: occupancy sweeps
    const size_t dyn = (size_t)(((L.frame_bytes + 3) >> 2) + 6) * 4 + lds_pad;
    if (X.pcm16) {
        int map[6];
        if (s16_channel_map(X.s16_flags, map) != X.plan.n_out || ((uintptr_t)X.pcm16 & 15)) return hipErrorInvalidValue;
        for (int w = 0; w < X.plan.n_out; w++) Q.wslot[map[w]] = (int8_t)w;
        hipLaunchKernelGGL(mantx_kernel<true>, dim3(M.n_frames), dim3(384), dyn, stream, Q);
    } else {
        hipLaunchKernelGGL(mantx_kernel<false>, dim3(M.n_frames), dim3(384), dyn, stream, Q);
    }
    return hipGetLastError();
}

}  // namespace ac3mi

#ifdef MX_STAMPS
extern "C" __attribute__((visibility("default"))) int ac3mi_debug_mx_cycles(unsigned long long *out48, int reset)
{
    if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(ac3mi::g_mx_cycles), sizeof(unsigned long long) * 48) != hipSuccess) return -1;
    if (reset) { unsigned long long z[48] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ac3mi::g_mx_cycles), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
