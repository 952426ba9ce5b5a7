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
//
// S16 variants: the reference's float -> s16 converter (src/AC3ASM.asm:174-318) folded into the
// output stage; a wavefront carries whole streams and writes interleaved 16-bit samples, see
// XformParams::pcm16.
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
    int frames;                 // stride of a stream in frames (coefficients, PCM, block-switch flags)
    int nblk;                   // blocks per chain (frames * 6 but for the single-transform hooks)
    int n_in, n_out, nfchans, in_lfe;
    float bias;
    int8_t mix[6][6];
    const int32_t *slot;        // optional: stream s keeps its overlap state in slot[s]
    int delay_stride;           // floats per state slot (n_out * 128 without slots)
    // few long streams: a chain is cut into n_seg segments of seg_blocks blocks, one 8-lane group each.  A block's
    // overlap tail depends on the block before it only, so a segment first re-derives the tail of the block ahead of it
    // (and segment 0 the tail of the chain's last block, which it writes back as the new state).
    int n_seg, seg_blocks;
    // S16 kernels: interleaved 16-bit output, plane o goes to WAVE slot wslot[o].  A wavefront then carries whole
    // streams (gps of them, n_out groups each; left-over groups shadow group 0 without storing anything), gathers a
    // block's samples in LDS and writes them out 16 bytes per lane.
    int16_t *pcm16;
    int8_t wslot[6];
    int n_streams, gps;
    // liba52's overlap bookkeeping around frames whose surround level is 0 (XformLaunch::mix_pending): null = linear mix
    const uint8_t *zs;          // [S][frames] or null
    float *mix_pending;         // per chain 128 floats, indexed like `delay`
    int32_t *mix_flags;         // per stream slot 6 words (one per output chain): bit 0 `downmixed`, bit 1 share pending
    uint32_t surr_mask;         // input planes that are surround channels mixed at slev
    uint32_t nobias_mask;       // output planes liba52 leaves without the bias in a per-channel-path block of a frame with slev == 0
    int downmixing;             // fewer full-bandwidth outputs than coded channels (parse.c:881-883)
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
        xa[n] = __builtin_fmaf(sign, v[n].x, xa[n]);                     // X[2m]
        xb[n] = __builtin_fmaf(sign, mirror8(v[15 - n].y), xb[n]);       // X[255-2m] = X[2m'+1], m' = 127-m
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
        xa[n] = __builtin_fmaf(sign, pa[16 * n], xa[n]);
        xb[n] = __builtin_fmaf(sign, pb[-16 * n], xb[n]);
    }
}

// MS: with liba52's surround-tail bookkeeping (XformParams::mix_pending), a variant of its own so that the plain mixing
// kernel keeps its registers and its speed
template <bool MIX, int WPS, bool S16 = false, bool MS = false>
__global__ __launch_bounds__(256, WPS) void xform_kernel(const XformParams P)
{
    __shared__ float2 lds_ex[4 * EX_WAVE];
    __shared__ float2 lds_twl[128];                 // merged long-block twiddles [lane][k]
    __shared__ float lds_win[256];                  // KBD window

    const int tid = threadIdx.x;
    const int l8 = tid & 7;
    const int group = tid >> 3;                     // 0..31 inside the workgroup
    int chain, seg, pack = 0, sl = 0;
    bool active = true;
    if (!S16) {
        const int vchain = blockIdx.x * 32 + group; // (chain, segment)
        chain = vchain / P.n_seg;
        seg = vchain - chain * P.n_seg;
    } else {
        const int unit = blockIdx.x * 4 + (tid >> 6);       // (pack of gps streams, segment) per wavefront
        pack = unit / P.n_seg;
        seg = unit - pack * P.n_seg;
        sl = (group & 7) / P.n_out;
        active = sl < P.gps && pack * P.gps + sl < P.n_streams;
        chain = active ? (pack * P.gps + sl) * P.n_out + ((group & 7) - sl * P.n_out) : pack * P.gps * P.n_out;
    }
    // constants shared by the whole workgroup live in LDS (registers are the scarce resource here)
    if (tid < 128) lds_twl[tid] = P.tw_long[tid];
    lds_win[tid] = P.window[tid];
    __syncthreads();
    if (chain >= P.n_chains) return;                // whole 8-lane groups leave together (S16: whole wavefronts)
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
    float *obase = S16 ? nullptr : P.pcm + (size_t)s * P.frames * 6 * out_stride_blk + (size_t)o * 256;
    const uint8_t *swbase = P.blksw ? P.blksw + (size_t)s * P.frames * 6 * P.nfchans : nullptr;
    int16_t *tile = reinterpret_cast<int16_t *>(lds_ex + (tid >> 6) * EX_WAVE);       // free between two transforms
    int wsl = 0;
#pragma unroll
    for (int oo = 0; oo < 6; oo++) wsl = oo == o ? P.wslot[oo] : wsl;      // (no per-lane indexing of kernel arguments)
    const int tbase = sl * 256 * P.n_out + wsl;
    // MIX: the input planes of this output as a packed list (3 bits each) with their signs
    uint32_t plist = 0, psign = 0, psurr = 0;     // psurr: entry k is a surround plane (mix-state bookkeeping)
    int pcnt = 0;
    if (MIX) {
#pragma unroll
        for (int oo = 0; oo < 6; oo++)
#pragma unroll
            for (int c = 0; c < 6; c++) {
                const int m = P.mix[oo][c];
                if (oo == o && c < P.n_in && m) {
                    plist |= (uint32_t)c << (3 * pcnt);
                    psign |= (m < 0 ? 1u : 0u) << pcnt;
                    psurr |= ((P.surr_mask >> c) & 1u) << pcnt;
                    pcnt++;
                }
            }
    }

    // ---- liba52's treatment of the surround channels' overlap tails around frames with slev == 0 (decode paths only) ----
    // liba52 keeps an overlap plane per coded channel and mixes in the time domain when the channels of a block differ in
    // block size ("path A", parse.c:884-916) or mixes coefficients and overlap planes first ("path B", :917-937, flag
    // `downmixed`).  With slev == 0 it leaves the surround channels out of transform and mix altogether: in path A their
    // overlap planes stay as they are (and come back when the level does), in path B with the planes not yet mixed they
    // are dropped.  The engine mixes linearly and keeps one tail per OUTPUT, so it holds the surround planes' share of the
    // tail apart whenever the next frame could be such a frame - P.mix_pending, written at a frame's last block - and then
    // lets it join, wait or vanish as liba52's planes would.
    constexpr bool mixstate = MIX && MS;
    float *pptr = nullptr;
    int32_t *fptr = nullptr;
    bool dm_flag = true, pend = false;
    if (mixstate) {
        const size_t sidx = (size_t)(P.slot ? P.slot[s] : s);
        pptr = P.mix_pending + sidx * P.delay_stride + (size_t)o * 128;
        fptr = P.mix_flags + sidx * 6 + o;
        const int fl = *fptr;
        dm_flag = fl & 1;
        pend = (fl >> 1) & 1;
    }
    const uint32_t fbw_planes = ((1u << P.nfchans) - 1u) << P.in_lfe;

    const int nblk = P.nblk;
    const int b_lo = seg * P.seg_blocks, b_hi = b_lo + P.seg_blocks < nblk ? b_lo + P.seg_blocks : nblk;
    // The overlap state is read and rewritten in place, and the segments of a chain run in groups (possibly workgroups)
    // of their own with nothing ordering them: so the group that READS the chain's state (segment 0) is also the one
    // that writes it back - it transforms the chain's last block once more for its tail (which depends on that block's
    // coefficients only).  The last segment never touches the state.
    const bool tail_pass = P.n_seg > 1 && seg == 0;
    const int b_first = seg ? b_lo - 1 : b_lo, n_iter = b_hi - b_first + (tail_pass ? 1 : 0);
    for (int it = 0; it < n_iter; it++) {
        const bool extra = tail_pass && it == n_iter - 1;
        const int b = extra ? nblk - 1 : b_first + it;
        const bool emit = !extra && b >= b_lo;      // the block ahead of a segment only supplies its tail
        const float *cblk = cbase + (size_t)b * in_stride_blk;
        FirstTail ft;
        float bias_blk = P.bias;
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
            uint32_t swm = 0;                       // bit c: input plane c is a short block here
            if (swbase) {
                const uint8_t *q = swbase + (size_t)b * P.nfchans;
#pragma unroll
                for (int fb = 0; fb < 5; fb++)
                    if (fb < P.nfchans) swm |= (q[fb] ? 1u : 0u) << (fb + P.in_lfe);
            }
            bool split = false;
            bias_blk = P.bias;
            if (mixstate) {
                const int fcur = b / 6;
                const uint8_t *zf = P.zs ? P.zs + (size_t)s * P.frames + fcur : nullptr;
                const int zs_now = zf ? zf[0] : 0;
                const uint32_t sf = swm & fbw_planes;
                const bool path_a = !P.downmixing || (sf != 0 && sf != fbw_planes);
                if (pend) {
                    if (!zs_now || dm_flag) {           // the surround planes are transformed again, or were mixed in already
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
                            const volatile float *pv = pptr + 2 * i;      // (this wavefront may have written it a block ago)
                            dl[j].x += pv[0];
                            dl[j].y += pv[1];
                        }
                        pend = false;
                    } else if (!path_a) {
                        pend = false;                   // path B mixes the overlap planes without the surrounds: gone
                    }                                   // path A: the planes wait
                }
                dm_flag = !path_a;
                if (zs_now && path_a && ((P.nobias_mask >> o) & 1u)) bias_blk = 0.f;       // liba52 forgets the bias here (MixPlan::nobias_mask)
                // a frame's last block: the surround share of the new tail goes aside unless the next frame is known to mix it
                split = !zs_now && psurr != 0 && b - 6 * fcur == 5 && (fcur + 1 >= P.frames || zf[1] != 0 || !P.zs);
            }
            float xa[16], xb[16];
            // entries of the plane list to take: all; or, for a split block, first the surround planes, then the others
#pragma unroll 1
            for (int pass = (mixstate && split) ? 0 : 1; pass < 2; pass++) {
                const uint32_t take = (mixstate && split) ? (pass == 0 ? psurr : ~psurr) : ~0u;
                bool any;
#pragma unroll
                for (int n = 0; n < 16; n++) xa[n] = xb[n] = 0.f;
                any = false;
#pragma unroll 1
                for (int k = 0; k < 6; k++) {
                    if (k >= pcnt || !((take >> k) & 1)) continue;
                    const int c = (plist >> (3 * k)) & 7;
                    if ((swm >> c) & 1) continue;
                    load_long(cblk + (size_t)c * 256, l8, ((psign >> k) & 1) ? -1.f : 1.f, xa, xb);
                    any = true;
                }
                if (any) imdct_long(xa, xb, twl, ex, l8, ft);
                if (swm) {
#pragma unroll
                    for (int n = 0; n < 16; n++) xa[n] = xb[n] = 0.f;
                    any = false;
#pragma unroll 1
                    for (int k = 0; k < 6; k++) {
                        if (k >= pcnt || !((take >> k) & 1)) continue;
                        const int c = (plist >> (3 * k)) & 7;
                        if (!((swm >> c) & 1)) continue;
                        load_short(cblk + (size_t)c * 256, l8, ((psign >> k) & 1) ? -1.f : 1.f, xa, xb);
                        any = true;
                    }
                    if (any) {
                        const float2 *tws = P.tw_short + l8 * 16;
                        imdct_short(xa, xb, tws, ex, l8, ft);
                    }
                }
                if (mixstate && split && pass == 0) {   // the surround planes' tail: aside, and out of the chain's own tail
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
                        volatile float *pv = pptr + 2 * i;
                        if (active) { pv[0] = ft.t0[j]; pv[1] = ft.t1[j]; }
                        ft.t0[j] = ft.t1[j] = 0.f;
                    }
                    pend = true;
                }
            }
        }

        // window + overlap-add + bias, then the new tail
        float *oblk = S16 ? nullptr : obase + (size_t)b * out_stride_blk;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
            const float2 wlo = *reinterpret_cast<const float2 *>(&lds_win[2 * i]);         // w[2i], w[2i+1]
            const float2 whi = *reinterpret_cast<const float2 *>(&lds_win[254 - 2 * i]);   // w[254-2i], w[255-2i]
            float2 lo, hi;
            window_pair(ft.f0[j], ft.f1[j], dl[j], wlo, whi, MIX && MS ? bias_blk : P.bias, lo, hi);
            if (emit && !S16) {
                *reinterpret_cast<float2 *>(oblk + 2 * i) = lo;
                *reinterpret_cast<float2 *>(oblk + 254 - 2 * i) = hi;
            }
            if (S16 && emit && active) {
                tile[tbase + (2 * i) * P.n_out] = to_s16(lo.x);
                tile[tbase + (2 * i + 1) * P.n_out] = to_s16(lo.y);
                tile[tbase + (254 - 2 * i) * P.n_out] = to_s16(hi.x);
                tile[tbase + (255 - 2 * i) * P.n_out] = to_s16(hi.y);
            }
            dl[j].x = ft.t0[j];
            dl[j].y = ft.t1[j];
        }
        if (S16 && emit) {                          // wave-uniform: a wavefront's groups share the segment
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int upb = 32 * P.n_out;           // 16-byte units per stream-block
#pragma unroll
            for (int it = 0; it < 4; it++) {
                const int u = it * 64 + (tid & 63);
                const int su = u / upb, stream = pack * P.gps + su;
                if (su < P.gps && stream < P.n_streams) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(tile + u * 8);
                    *reinterpret_cast<uint4 *>(P.pcm16 + ((size_t)stream * P.frames * 6 + b) * out_stride_blk + (size_t)(u - su * upb) * 8) = v;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }

    if (seg == 0 && active) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
            *reinterpret_cast<float2 *>(dptr + 2 * i) = dl[j];
        }
        if (mixstate && l8 == 0) *fptr = (dm_flag ? 1 : 0) | (pend ? 2 : 0);
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
    P.nblk = L.blocks > 0 ? L.blocks : L.frames * 6;
    P.n_in = L.plan.n_in;
    P.n_out = L.plan.n_out;
    P.nfchans = L.plan.nfchans;
    P.in_lfe = L.plan.in_lfe;
    P.zs = L.zs;
    P.mix_pending = L.mix_pending;
    P.mix_flags = L.mix_flags;
    P.surr_mask = L.plan.surr_mask;
    P.nobias_mask = L.plan.nobias_mask;
    {
        int out_fbw = L.plan.n_out;
        if (L.plan.n_out > 0 && L.plan.in_lfe && L.plan.mix[0][0] == 1) {        // output plane 0 is the LFE plane
            bool only_lfe = true;
            for (int c = 1; c < L.plan.n_in; c++) only_lfe = only_lfe && L.plan.mix[0][c] == 0;
            if (only_lfe) out_fbw--;
        }
        P.downmixing = out_fbw < L.plan.nfchans ? 1 : 0;
    }
    P.bias = L.bias;
    bool identity = (L.plan.n_in == L.plan.n_out);
    for (int o = 0; o < 6; o++)
        for (int c = 0; c < 6; c++) {
            P.mix[o][c] = L.plan.mix[o][c];
            if (o < L.plan.n_out && c < L.plan.n_in && L.plan.mix[o][c] != (o == c ? 1 : 0)) identity = false;
        }
    if (P.n_chains <= 0 || P.frames <= 0) return hipSuccess;
    // enough 8-lane groups to fill the chip (256 CUs x 4 workgroups x 32 groups): cut long chains into whole-frame segments
    const bool mixstate = !identity && P.mix_pending && P.mix_flags && P.surr_mask;
    if (!mixstate) P.mix_pending = nullptr;
    P.n_seg = 1;
    if (P.frames > 1 && P.n_chains < 32768 && !mixstate) {       // (the mix-state bookkeeping walks a stream's blocks in order)
        int want = (32768 + P.n_chains - 1) / P.n_chains;
        if (want > P.frames) want = P.frames;
        const int frames_per_seg = (P.frames + want - 1) / want;
        P.n_seg = (P.frames + frames_per_seg - 1) / frames_per_seg;
        P.seg_blocks = frames_per_seg * 6;
    }
    if (P.n_seg == 1) P.seg_blocks = P.frames * 6;
    int grid = (int)(((long long)P.n_chains * P.n_seg + 31) / 32);
    P.n_streams = L.n_streams;
    P.gps = 8 / P.n_out;
    if (L.pcm16) grid = (int)((((long long)(L.n_streams + P.gps - 1) / P.gps) * P.n_seg + 3) / 4);
    // 4 workgroups per CU (LDS: 40 KB each); measured on MI355X: 3 vs 4 waves/SIMD, packed vs scalar f32 and
    // 8- vs 16-byte accesses all land within 1 % - the kernel runs at the rate of a plain copy with the same
    // addressing (profiles/r01_xform_probes.md)
    // Workgroups per CU are capped through the LDS request: the float identity kernel fits four per CU (40 KB, 116 VGPRs) but
    // runs 3.4 % faster with three - fewer chains in flight keep the DRAM pages of the coefficient and PCM streams open longer
    // (measured on MI355X, 65 536 frames: 4 per CU 0.979 / 0.950 ms, 3 per CU 0.947 / 0.916 ms, 2 per CU 1.082 ms; the bare copy
    // with this addressing shows the same trend, profiles/hbm_calibrate).  AC3MI_XFORM_LDS_PAD overrides (occupancy sweeps).
    static const int lds_pad = getenv("AC3MI_XFORM_LDS_PAD") ? atoi(getenv("AC3MI_XFORM_LDS_PAD")) : 14 * 1024;
    static const int lds_pad_s16 = getenv("AC3MI_XFORM_LDS_PAD_S16") ? atoi(getenv("AC3MI_XFORM_LDS_PAD_S16")) : 0;
    static const int lds_pad_mix = getenv("AC3MI_XFORM_LDS_PAD_MIX") ? atoi(getenv("AC3MI_XFORM_LDS_PAD_MIX")) : 0;
    P.pcm16 = L.pcm16;
    if (L.pcm16) {
        int map[6];
        if (s16_channel_map(L.s16_flags, map) != P.n_out || ((uintptr_t)L.pcm16 & 15)) return hipErrorInvalidValue;
        for (int w = 0; w < P.n_out; w++) P.wslot[map[w]] = (int8_t)w;
        if (identity)
            hipLaunchKernelGGL((xform_kernel<false, 3, true>), dim3(grid), dim3(256), lds_pad_s16, stream, P);
        else if (mixstate)
            hipLaunchKernelGGL((xform_kernel<true, 2, true, true>), dim3(grid), dim3(256), 0, stream, P);
        else
            hipLaunchKernelGGL((xform_kernel<true, 2, true>), dim3(grid), dim3(256), 0, stream, P);
        return hipGetLastError();
    }
    if (identity)
        hipLaunchKernelGGL((xform_kernel<false, 4>), dim3(grid), dim3(256), lds_pad, stream, P);
    else if (mixstate)
        hipLaunchKernelGGL((xform_kernel<true, 2, false, true>), dim3(grid), dim3(256), 0, stream, P);
    else        // 167 VGPRs with the planes of an output accumulated one after the other (187 with their loads unrolled): 3 workgroups per CU
        hipLaunchKernelGGL((xform_kernel<true, 3>), dim3(grid), dim3(256), lds_pad_mix, stream, P);
    return hipGetLastError();
}

}  // namespace ac3mi
