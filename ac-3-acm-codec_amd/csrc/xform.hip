// xform.hip — batched AC-3 synthesis transform for gfx950 (config 2 and config 4 of
// BASELINE.json): IMDCT-512 / IMDCT-256 + KBD window + overlap-add, with the
// liba52 downmix folded in.  Replaces L52/imdct.c:258-345 + L52/downmix.c:480-619 +
// the dispatch at L52/parse.c:881-937.
//
// Work decomposition: one "chain" = one OUTPUT channel of one stream, walked
// sequentially through its frames x 6 blocks because each block's tail is the next
// block's overlap.  A chain is executed by an 8-lane group (xform_core.h); a
// wavefront carries 8 chains, a 256-thread workgroup 32.  The 128-float overlap
// tail lives in registers for the whole walk and touches HBM once per call.
//
// HBM traffic per channel-block: 1 KiB coefficients in (x number of mixed input
// planes), 1 KiB PCM out; all accesses are 8 B per lane, 64 B contiguous per group.
// LDS: one 8x16 complex transpose per transform (1216 B per group, padded rows).
#include "ac3mi_internal.h"
#include "xform_core.h"

namespace ac3mi {

struct XformParams {
    const float *coef;
    const uint8_t *blksw;
    float *delay;
    float *pcm;
    const float2 *tw_long;
    const float2 *tw_short;
    const float *window;
    int n_chains;
    int frames;
    int n_in, n_out, nfchans, in_lfe;
    float bias;
    int8_t mix[6][6];
    const int32_t *slot;        // optional: stream s keeps its overlap state in slot[s]
    int delay_stride;           // floats per state slot (n_out * 128 without slots)
    // few long streams: a chain is cut into n_seg segments of seg_blocks blocks, one 8-lane group each.  A block's
    // overlap tail depends on the block before it only, so a segment first re-derives the tail of the block ahead of it.
    int n_seg, seg_blocks;
};

// long-block input pattern: lane l8 owns m = 8*n1 + l8
__device__ __forceinline__ void load_long(const float *plane, int l8, float sign, float (&xa)[16], float (&xb)[16])
{
    const float2 *p = reinterpret_cast<const float2 *>(plane) + l8;
    float2 v[16];
#pragma unroll
    for (int n = 0; n < 16; n++) v[n] = p[8 * n];
#pragma unroll
    for (int n = 0; n < 16; n++) {
        xa[n] += sign * v[n].x;                     // X[2m]
        xb[n] += sign * mirror8(v[15 - n].y);       // X[255-2m] = X[2m'+1], m' = 127-m
    }
}

// short-block input pattern: lane l8 = 4f + n2
__device__ __forceinline__ void load_short(const float *plane, int l8, float sign, float (&xa)[16], float (&xb)[16])
{
    const int f = l8 >> 2, n2 = l8 & 3;
    const float *pa = plane + 4 * n2 + f;
    const float *pb = plane + 254 + f - 4 * n2;
#pragma unroll
    for (int n = 0; n < 16; n++) {
        xa[n] += sign * pa[16 * n];
        xb[n] += sign * pb[-16 * n];
    }
}

template <bool MIX, int WPS>
__global__ __launch_bounds__(256, WPS) void xform_kernel(const XformParams P)
{
    __shared__ float2 lds_ex[4 * EX_WAVE];
    __shared__ float2 lds_twl[128];                 // merged long-block twiddles [lane][k]
    __shared__ float lds_win[256];                  // KBD window

    const int tid = threadIdx.x;
    const int l8 = tid & 7;
    const int group = tid >> 3;                     // 0..31 inside the workgroup
    const int vchain = blockIdx.x * 32 + group;     // (chain, segment)
    const int chain = vchain / P.n_seg, seg = vchain - chain * P.n_seg;
    // constants shared by the whole workgroup live in LDS (registers are the scarce resource here)
    if (tid < 128) lds_twl[tid] = P.tw_long[tid];
    lds_win[tid] = P.window[tid];
    __syncthreads();
    if (chain >= P.n_chains) return;                // whole 8-lane groups leave together (chain = vchain / n_seg)
    float2 *ex = lds_ex + group * EX_GROUP;
    const float2 *twl = lds_twl + l8 * 16;

    const int s = chain / P.n_out;
    const int o = chain - s * P.n_out;
    // overlap tail of this chain
    float2 dl[8];
    float *dptr = P.delay + (size_t)(P.slot ? P.slot[s] : s) * P.delay_stride + (size_t)o * 128;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
        dl[j] = seg == 0 ? *reinterpret_cast<const float2 *>(dptr + 2 * i) : make_float2(0.f, 0.f);
    }

    const size_t in_stride_blk = (size_t)P.n_in * 256;
    const size_t out_stride_blk = (size_t)P.n_out * 256;
    const float *cbase = P.coef + (size_t)s * P.frames * 6 * in_stride_blk;
    float *obase = P.pcm + (size_t)s * P.frames * 6 * out_stride_blk + (size_t)o * 256;
    const uint8_t *swbase = P.blksw ? P.blksw + (size_t)s * P.frames * 6 * P.nfchans : nullptr;

    const int nblk = P.frames * 6;
    const int b_lo = seg * P.seg_blocks, b_hi = b_lo + P.seg_blocks < nblk ? b_lo + P.seg_blocks : nblk;
    for (int b = seg ? b_lo - 1 : b_lo; b < b_hi; b++) {
        const bool emit = b >= b_lo;                // the block ahead of a segment only supplies its tail
        const float *cblk = cbase + (size_t)b * in_stride_blk;
        FirstTail ft;
#pragma unroll
        for (int j = 0; j < 8; j++) ft.f0[j] = ft.f1[j] = ft.t0[j] = ft.t1[j] = 0.f;

        if (!MIX) {
            // identity routing: input plane index == output plane index
            const int fb = o - P.in_lfe;
            const int sw = (swbase && fb >= 0) ? swbase[(size_t)b * P.nfchans + fb] : 0;
            float xa[16], xb[16];
#pragma unroll
            for (int n = 0; n < 16; n++) xa[n] = xb[n] = 0.f;
            if (!sw) {
                load_long(cblk + (size_t)o * 256, l8, 1.f, xa, xb);
                imdct_long(xa, xb, twl, ex, l8, ft);
            } else {
                const float2 *tws = P.tw_short + l8 * 16;      // rare path: straight from L1/L2
                load_short(cblk + (size_t)o * 256, l8, 1.f, xa, xb);
                imdct_short(xa, xb, tws, ex, l8, ft);
            }
        } else {
            // sum the long-block inputs and the short-block inputs of this output
            // separately (the transforms are linear), at most one transform of each kind
            float xa[16], xb[16];
            bool any;
#pragma unroll
            for (int n = 0; n < 16; n++) xa[n] = xb[n] = 0.f;
            any = false;
            for (int c = 0; c < P.n_in; c++) {
                const int m = P.mix[o][c];
                if (!m) continue;
                const int fb = c - P.in_lfe;
                const int sw = (swbase && fb >= 0) ? swbase[(size_t)b * P.nfchans + fb] : 0;
                if (sw) continue;
                load_long(cblk + (size_t)c * 256, l8, (float)m, xa, xb);
                any = true;
            }
            if (any) imdct_long(xa, xb, twl, ex, l8, ft);
            if (swbase) {
#pragma unroll
                for (int n = 0; n < 16; n++) xa[n] = xb[n] = 0.f;
                any = false;
                for (int c = 0; c < P.n_in; c++) {
                    const int m = P.mix[o][c];
                    if (!m) continue;
                    const int fb = c - P.in_lfe;
                    const int sw = (fb >= 0) ? swbase[(size_t)b * P.nfchans + fb] : 0;
                    if (!sw) continue;
                    load_short(cblk + (size_t)c * 256, l8, (float)m, xa, xb);
                    any = true;
                }
                if (any) {
                    const float2 *tws = P.tw_short + l8 * 16;
                    imdct_short(xa, xb, tws, ex, l8, ft);
                }
            }
        }

        // window + overlap-add + bias, then the new tail
        float *oblk = obase + (size_t)b * out_stride_blk;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
            const float2 wlo = *reinterpret_cast<const float2 *>(&lds_win[2 * i]);         // w[2i], w[2i+1]
            const float2 whi = *reinterpret_cast<const float2 *>(&lds_win[254 - 2 * i]);   // w[254-2i], w[255-2i]
            float2 lo, hi;
            lo.x = ft.f0[j] * wlo.x + (dl[j].x * whi.y + P.bias);          // out[2i]
            lo.y = ft.f1[j] * wlo.y + (dl[j].y * whi.x + P.bias);          // out[2i+1]
            hi.x = dl[j].y * wlo.y + P.bias - ft.f1[j] * whi.x;            // out[254-2i]
            hi.y = dl[j].x * wlo.x + P.bias - ft.f0[j] * whi.y;            // out[255-2i]
            if (emit) {
                *reinterpret_cast<float2 *>(oblk + 2 * i) = lo;
                *reinterpret_cast<float2 *>(oblk + 254 - 2 * i) = hi;
            }
            dl[j].x = ft.t0[j];
            dl[j].y = ft.t1[j];
        }
    }

    if (b_hi == nblk) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
            *reinterpret_cast<float2 *>(dptr + 2 * i) = dl[j];
        }
    }
}

hipError_t launch_xform(const DeviceTables &tab, const XformLaunch &L, hipStream_t stream)
{
    XformParams P;
    P.coef = L.coef;
    P.blksw = L.blksw;
    P.delay = L.delay;
    P.slot = L.slot;
    P.delay_stride = L.slot ? L.delay_stride : L.plan.n_out * 128;
    P.pcm = L.pcm;
    P.tw_long = tab.tw_long;
    P.tw_short = tab.tw_short;
    P.window = tab.window;
    P.n_chains = L.n_streams * L.plan.n_out;
    P.frames = L.frames;
    P.n_in = L.plan.n_in;
    P.n_out = L.plan.n_out;
    P.nfchans = L.plan.nfchans;
    P.in_lfe = L.plan.in_lfe;
    P.bias = L.bias;
    bool identity = (L.plan.n_in == L.plan.n_out);
    for (int o = 0; o < 6; o++)
        for (int c = 0; c < 6; c++) {
            P.mix[o][c] = L.plan.mix[o][c];
            if (o < L.plan.n_out && c < L.plan.n_in && L.plan.mix[o][c] != (o == c ? 1 : 0)) identity = false;
        }
    if (P.n_chains <= 0 || P.frames <= 0) return hipSuccess;
    // enough 8-lane groups to fill the chip (256 CUs x 4 workgroups x 32 groups): cut long chains into whole-frame segments
    P.n_seg = 1;
    if (P.frames > 1 && P.n_chains < 32768) {
        int want = (32768 + P.n_chains - 1) / P.n_chains;
        if (want > P.frames) want = P.frames;
        const int frames_per_seg = (P.frames + want - 1) / want;
        P.n_seg = (P.frames + frames_per_seg - 1) / frames_per_seg;
        P.seg_blocks = frames_per_seg * 6;
    }
    if (P.n_seg == 1) P.seg_blocks = P.frames * 6;
    const int grid = (int)(((long long)P.n_chains * P.n_seg + 31) / 32);
    // 4 workgroups per CU (LDS: 40 KB each); measured on MI355X: 3 vs 4 waves/SIMD, packed vs scalar f32 and
    // 8- vs 16-byte accesses all land within 1 % - the kernel runs at the rate of a plain copy with the same
    // addressing (profiles/r01_xform_probes.md)
    if (identity)
        hipLaunchKernelGGL((xform_kernel<false, 4>), dim3(grid), dim3(256), 0, stream, P);
    else
        hipLaunchKernelGGL((xform_kernel<true, 2>), dim3(grid), dim3(256), 0, stream, P);
    return hipGetLastError();
}

}  // namespace ac3mi
