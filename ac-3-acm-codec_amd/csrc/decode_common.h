// decode_common.h — device-side pieces shared by the two AC-3 front ends (decode.hip: one wavefront per stream or
// per frame; decode_wg.hip: one workgroup per stream, one wavefront per channel, transform fused in): bit reader,
// exponent decode (L52/parse.c:218-270), parametric bit allocation (L52/bit_allocate.c:124-265), dither access.
#pragma once
#include "ac3mi_internal.h"
#include "a52_levels.h"
#include "wave_ops.h"

namespace ac3mi {

constexpr int ROW = 260;                  // exp/bap row pitch (bytes): 65 dwords, conflict-free across rows
// The LFE row holds 7 exponents (bit_allocate_wave's lowcomp stage looks at bins up to 64 of any row): kept short and
// last; with the dequantiser table read through L1 the wavefront's LDS is 6.5 KB = 24 wavefronts per CU, what the
// 80 VGPRs of __launch_bounds__(64, 6) allow (the kernel is latency-bound, occupancy pays).
constexpr int LFE_ROW = 68;
constexpr int GRING = 128;                // see DecLDS::gcode
constexpr int ROWS = 6 * ROW + LFE_ROW;
__device__ __forceinline__ int row_off(int slot) { return slot < 5 ? slot * ROW : slot == 6 ? 5 * ROW : 6 * ROW; }

__device__ const uint8_t k_nfchans[11] = {2, 1, 2, 3, 3, 4, 4, 5, 1, 1, 2};
__device__ const float k_clev[4] = {(float)AC3MI_G_3DB, (float)AC3MI_G_45DB, (float)AC3MI_G_6DB, (float)AC3MI_G_45DB};
__device__ const float k_slev[4] = {(float)AC3MI_G_3DB, (float)AC3MI_G_6DB, 0.f, (float)AC3MI_G_6DB};
__device__ const uint8_t k_cpl_bnd0[16] = {31, 35, 37, 39, 41, 42, 43, 44, 45, 45, 46, 46, 47, 47, 48, 48};
__device__ const int k_remat_edge[5] = {13, 25, 37, 61, 253};
__device__ const int k_slowgain[4] = {0x540, 0x4d8, 0x478, 0x410};
__device__ const int k_dbpb[4] = {0xc00, 0x500, 0x300, 0x100};
__device__ const int k_floors[8] = {0x910, 0x950, 0x990, 0x9d0, 0xa10, 0xa90, 0xb10, 0x1400};
__device__ const uint16_t k_kbps[19] = {32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640};

// channel slots: 0..4 = fbw, 5 = lfe, 6 = coupling channel
struct DecLDS {
    uint8_t exp[ROWS];                    // rows at row_off(slot)
    int8_t bap[ROWS];
    int8_t deltba[6][52];                 // 0..4 fbw, 5 = cpl
    float cplco[5][18];
    uint8_t gcode[3 * 128 + 4];           // open 3/5/11-level codes: a ring of GRING groups per kind (a segment opens < 128), then the sink
    uint8_t cplbnd[20];                   // coupling sub-band -> band
    int8_t la_neg[256];
    uint16_t hth[50];
    int8_t width[64];
    uint8_t band_end[30];
    uint8_t band_of_bin[256];
    int16_t bmask[52];                    // per-band mask of the channel being allocated
    uint32_t desc[100];                   // mant_desc of the row bytes 0..96
    uint32_t tot[7][4];                   // parse modes: row_totals of the slot's row (a, b) under key start | end << 10 | valid
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// the staged frame: byte-swapped dwords in dynamic LDS, frame_bytes/4 (rounded up) + 6 zero words
struct FrameBits {
    const uint32_t *w;
    uint32_t last;          // highest index a 2-dword read may start at
};

// n in 1..32: the n bits starting at bit `pos` (reads past the padding return the padding)
__device__ __forceinline__ uint32_t peek(const FrameBits fr, uint32_t pos, int n)
{
    uint32_t w = pos >> 5;
    w = w < fr.last ? w : fr.last;
    const uint64_t v = ((uint64_t)fr.w[w] << 32) | fr.w[w + 1];
    return (uint32_t)((v << (pos & 31)) >> (64 - n));
}
__device__ __forceinline__ int32_t speek(const FrameBits fr, uint32_t pos, int n)
{
    return ((int32_t)(peek(fr, pos, n) << (32 - n))) >> (32 - n);
}

struct Rd {                                   // wave-uniform serial reader
    FrameBits fr;
    // 64-bit window (scalar registers): one LDS read per ~32 bits instead of one per field.  The next bit to read is the
    // window's top bit, at position end - avail; a field costs two shifts and a subtraction.
    uint32_t end;
    uint64_t win;
    int avail;
    __device__ __forceinline__ uint32_t pos() const { return end - (uint32_t)avail; }
    __device__ __forceinline__ void seek(uint32_t p) { end = p; avail = 0; }
    __device__ __forceinline__ void refill()
    {
        const uint32_t p = pos();
        uint32_t w = p >> 5;
        w = w < fr.last ? w : fr.last;
        const uint32_t hi = rfl(fr.w[w]), lo = rfl(fr.w[w + 1]);
        win = (((uint64_t)hi << 32) | lo) << (p & 31);
        avail = 64 - (int)(p & 31);
        end = p + (uint32_t)avail;
    }
    __device__ __forceinline__ uint32_t get(int n)
    {
        if (n == 0) return 0;
        if (avail < n) refill();
        const uint32_t v = (uint32_t)(win >> (64 - n));
        win <<= n;
        avail -= n;
        return v;
    }
    __device__ __forceinline__ int32_t sget(int n)
    {
        const uint32_t v = get(n);
        return ((int32_t)(v << (32 - n))) >> (32 - n);
    }
    __device__ __forceinline__ void skip(uint32_t n) { seek(pos() + n); }
};

__device__ __forceinline__ float sf_of(int e) { return __int_as_float((127 - 15 - e) << 23); }   // 2^-(15+e)

__device__ __forceinline__ int wave_incl_scan(int v, int) { return (int)wave_incl_scan_u32((uint32_t)v); }

// liba52 drops the surround channels of such a frame from transform and mix instead of mixing them at level 0
// (L52/downmix.c:494-583: the slev == 0 cases of MONO, STEREO and 3F outputs)
__device__ __forceinline__ int surround_level_is_zero(int acmod, int output, float slev)
{
    const int out = output & 15;
    return ((acmod & 4) && slev == 0.f && (out == 1 || out == 2 || out == 3)) ? 1 : 0;
}

// stream-persistent decoder fields (wave-uniform)
struct St {
    int acmod, lfeon, nf;
    float clev, slev, level, bias_unused, dynrng;
    int output;
    int chincpl, cplstrtmant;
    uint32_t cplbndstrc;
    // Everything else packed (round 4: the parse kernel spills scalar registers by the hundred; these fields were 38 of them,
    // and a slot picked at run time is now a shift instead of a chain of selects)
    uint64_t ends;          // byte k: end of slot k's mantissas - 0..4 the full-bandwidth channels' endmant, 5: 7 (LFE), 6: cplendmant
    uint64_t cbai8;         // byte k: slot k's fsnroffst << 3 | fgaincod (0..4 fbw, 5 lfe, 6 cpl)
    uint32_t deltbae2;      // 2 bits per slot: 0..4 fbw, 5 = cpl
    uint32_t cplw;          // phsflginu 0 | ncplbnd 1-5 | cplstrtbnd 6-11 | cplfleak 12-15 | cplsleak 16-19 | rematflg 20-23
    uint32_t cfg;           // fscod 0-1 | halfrate 2-3 | dynrnge 4 | bai 5-15 | csnroffst 16-21
    __device__ __forceinline__ int endm(int k) const { return (int)((uint32_t)(ends >> (8 * k)) & 0xffu); }
    __device__ __forceinline__ void set_endm(int k, int v) { ends = (ends & ~(0xffull << (8 * k))) | ((uint64_t)(uint32_t)v << (8 * k)); }
    __device__ __forceinline__ int cbai(int k) const { return (int)((uint32_t)(cbai8 >> (8 * k)) & 0xffu); }
    __device__ __forceinline__ void set_cbai(int k, int v) { cbai8 = (cbai8 & ~(0xffull << (8 * k))) | ((uint64_t)(uint32_t)v << (8 * k)); }
    __device__ __forceinline__ int deltbae(int k) const { return (int)((deltbae2 >> (2 * k)) & 3u); }
    __device__ __forceinline__ void set_deltbae(int k, int v) { deltbae2 = (deltbae2 & ~(3u << (2 * k))) | ((uint32_t)v << (2 * k)); }
    __device__ __forceinline__ int cplendmant() const { return endm(6); }
#define AC3MI_ST_FIELD(name, word, pos, bits) \
    __device__ __forceinline__ int name() const { return (int)((word >> (pos)) & ((1u << (bits)) - 1u)); } \
    __device__ __forceinline__ void set_##name(int v) { word = (word & ~(((1u << (bits)) - 1u) << (pos))) | ((uint32_t)v << (pos)); }
    AC3MI_ST_FIELD(phsflginu, cplw, 0, 1)
    AC3MI_ST_FIELD(ncplbnd, cplw, 1, 5)
    AC3MI_ST_FIELD(cplstrtbnd, cplw, 6, 6)
    AC3MI_ST_FIELD(cplfleak, cplw, 12, 4)
    AC3MI_ST_FIELD(cplsleak, cplw, 16, 4)
    AC3MI_ST_FIELD(rematflg, cplw, 20, 4)
    AC3MI_ST_FIELD(fscod, cfg, 0, 2)
    AC3MI_ST_FIELD(halfrate, cfg, 2, 2)
    AC3MI_ST_FIELD(dynrnge, cfg, 4, 1)
    AC3MI_ST_FIELD(bai, cfg, 5, 11)
    AC3MI_ST_FIELD(csnroffst, cfg, 16, 6)
#undef AC3MI_ST_FIELD
    uint32_t lfsr;
};

struct BlkDesc;
struct DecodeParams {
    // split front end (decode_kernel<4 / 5> -> mant_kernel): block descriptors, row sets, coupling coordinates, and the
    // dither generator's position at the start of every frame
    BlkDesc *desc;
    uint8_t *rows;
    float *cplco;
    uint32_t *frame_pos;
    const uint8_t *frames;
    float *coef;
    uint8_t *blksw;
    uint8_t *zs;            // optional [S][F]: the frame mixes its surround channels at level 0 (liba52: coeff[i] == 0)
    uint32_t *status;
    uint16_t *lfsr_state;
    uint8_t *tap_exp;       // optional [S][F][6][7][256]
    int8_t *tap_bap;        // optional, same shape
    const uint16_t *lfsr_seq;   // [65535] LFSR states in cycle order starting at 1
    const uint16_t *lfsr_idx;   // [65536] position of a state in that cycle
    const DecTables *tab;
    int n_streams, frames_per_stream, frame_stride, frame_bytes;
    int req_flags;
    float level;
    int dynrng_on;
    int acmod, lfeon;       // expected coded configuration (frame 0 of the batch)
    int n_in, nfchans;
    const int32_t *slot;    // optional: stream s keeps its LFSR state in lfsr_state[slot[s]]
    uint32_t *frame_draws;  // [S][F] dither draws of each frame (written by the counting pass)
    const uint16_t *frame_lfsr;   // [S][F] LFSR state at the start of each frame (frame-parallel pass)
    float *dyn_out;               // optional [S][F][6][2], see ac3mi_decode_taps
    const float *dyn_in;
};

// the range factor of a dynamic-range word (parse.c:587-595), through the optional per-word taps
__device__ __forceinline__ float dynrng_range(const DecodeParams &P, int code, size_t word_index, int lane)
{
    float range = (float)(((code & 0x1f) | 0x20) << 13) * __int_as_float((127 - 15 - (3 - (code >> 5))) << 23);
    if (P.dyn_out && lane == 0) P.dyn_out[word_index] = range;
    if (P.dyn_in) {
        const float r = P.dyn_in[word_index];
        if (r == r) range = r;                           // NaN = keep the stream's own
    }
    return range;
}

// ---------------------------------------------------------------------------
// exponents: L52/parse.c:218-270.  ngrps groups of 7 bits at `pos`; returns 1 on a
// reserved code or an exponent outside 0..24.
__device__ int read_exponents(const FrameBits fr, uint32_t pos, int strategy, int ngrps, int absexp,
                              uint8_t *dst, int lane)
{
    const int rep = 1 << (strategy - 1);
    int carry = absexp, bad = 0;
    for (int g0 = 0; g0 < ngrps; g0 += 64) {
        const int g = g0 + lane;
        int d0 = 0, d1 = 0, d2 = 0, ok = 1;
        if (g < ngrps) {
            const int code = (int)peek(fr, pos + 7 * g, 7);
            ok = code < 125;
            d0 = code / 25 - 2;
            d1 = (code / 5) % 5 - 2;
            d2 = code % 5 - 2;
            if (!ok) d0 = d1 = d2 = 0;
        }
        const int incl = wave_incl_scan(d0 + d1 + d2, lane);
        const int e0 = carry + incl - (d1 + d2), e1 = e0 + d1, e2 = e1 + d2;
        if (g < ngrps) {
            if (!ok || e0 < 0 || e0 > 24 || e1 < 0 || e1 > 24 || e2 < 0 || e2 > 24) bad = 1;
            uint8_t *p = dst + 3 * g * rep;
            for (int r = 0; r < rep; r++) {
                p[r] = (uint8_t)e0;
                p[rep + r] = (uint8_t)e1;
                p[2 * rep + r] = (uint8_t)e2;
            }
        }
        carry += __shfl(incl, 63, 64);
    }
    return __any(bad) ? 1 : 0;
}

// ---------------------------------------------------------------------------
// bit allocation, one lane = one channel: L52/bit_allocate.c:124-265

struct BaCtx {
    int fdecay, fgain, sdecay, sgain, dbknee, floor, snroffset, halfrate, fast, slow;
    const uint16_t *hth;
    const int8_t *deltba;       // may be null
};

__device__ __forceinline__ void ba_leak(BaCtx &c, int psd)
{
    c.fast += c.fdecay;
    if (c.fast > psd + c.fgain) c.fast = psd + c.fgain;
    c.slow += c.sdecay;
    if (c.slow > psd + c.sgain) c.slow = psd + c.sgain;
}

__device__ __forceinline__ int ba_mask(const BaCtx &c, int mask, int psd, int band)
{
    const int h = c.hth[band >> c.halfrate];
    if (psd > c.dbknee) mask -= (psd - c.dbknee) >> 2;
    if (mask > h) mask = h;
    mask -= c.snroffset + 128 * (c.deltba ? c.deltba[band] : 0);
    mask = (mask > 0) ? 0 : ((-mask) >> 5);
    return mask - c.floor;
}

__device__ __forceinline__ int8_t ba_width(const int8_t *width, int a)
{
    return a <= -64 ? 16 : a >= 0 ? 0 : width[a + 63];
}

__device__ void bit_allocate_lane(const DecLDS &L, BaCtx c, int bndstart, int start, int end, const uint8_t *e,
                                  int8_t *bap)
{
    int band = bndstart, bin = start, psd = 0, mask;
    if (start == 0) {
        int lowcomp = 0;
        const int last = end - 1;
        do {
            if (band < last) {
                if (e[band + 1] == e[band] - 2) lowcomp = 384;
                else if (lowcomp && e[band + 1] > e[band]) lowcomp -= 64;
            }
            psd = 128 * e[band];
            mask = ba_mask(c, psd + c.fgain + lowcomp, psd, band);
            bap[band] = ba_width(L.width, mask + 4 * e[band]);
            band++;
        } while (band < 3 || (band < 7 && e[band] > e[band - 1]));
        c.fast = psd + c.fgain;
        c.slow = psd + c.sgain;
        while (band < 7) {
            if (band < last) {
                if (e[band + 1] == e[band] - 2) lowcomp = 384;
                else if (lowcomp && e[band + 1] > e[band]) lowcomp -= 64;
            }
            psd = 128 * e[band];
            ba_leak(c, psd);
            mask = (c.fast + lowcomp < c.slow) ? c.fast + lowcomp : c.slow;
            mask = ba_mask(c, mask, psd, band);
            bap[band] = ba_width(L.width, mask + 4 * e[band]);
            band++;
        }
        if (end == 7) return;
        do {
            if (e[band + 1] == e[band] - 2) lowcomp = 320;
            else if (lowcomp && e[band + 1] > e[band]) lowcomp -= 64;
            psd = 128 * e[band];
            ba_leak(c, psd);
            mask = (c.fast + lowcomp < c.slow) ? c.fast + lowcomp : c.slow;
            mask = ba_mask(c, mask, psd, band);
            bap[band] = ba_width(L.width, mask + 4 * e[band]);
            band++;
        } while (band < 20);
        while (lowcomp > 128) {
            lowcomp -= 128;
            psd = 128 * e[band];
            ba_leak(c, psd);
            mask = (c.fast + lowcomp < c.slow) ? c.fast + lowcomp : c.slow;
            mask = ba_mask(c, mask, psd, band);
            bap[band] = ba_width(L.width, mask + 4 * e[band]);
            band++;
        }
        bin = band;
    }
    do {
        const int first = bin;
        const int stop = L.band_end[band - 20] < end ? L.band_end[band - 20] : end;
        psd = 128 * e[bin++];
        while (bin < stop) {
            const int next = 128 * e[bin++], d = next - psd;
            const int q = d >> 9;
            if (q <= -2) psd = next;
            else if (q == -1) { int a = (-d) >> 1; psd = next + L.la_neg[a > 255 ? 255 : a]; }
            else if (q == 0) psd += L.la_neg[d >> 1];
        }
        ba_leak(c, psd);
        mask = ba_mask(c, c.fast < c.slow ? c.fast : c.slow, psd, band);
        band++;
        for (bin = first; bin < stop; bin++) bap[bin] = ba_width(L.width, mask + 4 * e[bin]);
    } while (bin < end);
}

// The same allocation for one channel with the whole wavefront (one lane per band, then per bin):
//  * band PSDs: each of the <= 50 bands integrated by its own lane (<= 24 dependent log-adds)
//  * lowcomp (bands 0..21 of a channel that starts at bin 0) is a reset-or-decrement automaton:
//    value = max(0, R(last reset) - 64 * decrements since); bands 20/21 take 128 off while > 128
//  * the fast / slow leaks are prefix minima of psd + gain - band * decay seeded at the last band
//    of the first loop (bit_allocate.c:150-166), or at the coupling leak values
template <class LDS>
__device__ void bit_allocate_wave(const LDS &L, int16_t *bmask, const BaCtx &c, int bndstart, int start, int end, const uint8_t *e,
                                  int8_t *bap, int lane)
{
    constexpr int INF = 0x3fffffff, NEG = -0x3fffffff;
    const int b = lane;
    // S1: band PSD
    int lo = b < 21 ? b : (int)L.band_end[b < 50 ? b - 21 : 0];
    int hi = b < 20 ? b + 1 : (int)L.band_end[b < 50 ? b - 20 : 0];
    lo = lo > start ? lo : start;
    hi = hi < end ? hi : end;
    const int w = b < 50 ? hi - lo : 0;
    const bool live = w > 0;
    lo = live ? lo : start;
    int psd = 128 * e[lo];
    const int wmax = (int)wave_last((uint32_t)wave_incl_scan_max(w));
    for (int j = 1; j < wmax; j++) {
        const int next = 128 * e[j < w ? lo + j : lo];
        const int d = next - psd, q = d >> 9;
        int idx = q == -1 ? (-d) >> 1 : d >> 1;
        idx = idx < 0 ? 0 : idx > 255 ? 255 : idx;
        const int la = L.la_neg[idx];
        const int nv = q <= -2 ? next : q == -1 ? next + la : q == 0 ? psd + la : psd;
        psd = j < w ? nv : psd;
    }
    // S2: lowcomp and the extent of the first loop
    int lc = 0, bA = 0;
    if (start == 0) {
        const int eb = e[b < 253 ? b : 0], eb1 = e[b < 252 ? b + 1 : 0], ebm = e[b > 0 && b < 254 ? b - 1 : 0];
        const bool upd = b < 20 && (b >= 7 || b < end - 1);
        const bool reset = upd && eb1 == eb - 2;
        const int dec = upd && !reset && eb1 > eb ? 1 : 0;
        const int D = (int)wave_incl_scan_u32((uint32_t)dec);
        const int rix = wave_incl_scan_max(reset ? b : NEG);
        const int Dr = __shfl(D, rix < 0 ? 0 : rix, 64);
        if (rix >= 0) { lc = (rix < 7 ? 384 : 320) - 64 * (D - Dr); lc = lc < 0 ? 0 : lc; }
        const int l19 = __builtin_amdgcn_readlane(lc, 19);
        lc = b == 20 ? (l19 > 128 ? l19 - 128 : 0) : b == 21 ? (l19 > 256 ? l19 - 256 : 0) : b >= 22 ? 0 : lc;
        const unsigned long long stopm = __ballot(b >= 3 && b < 7 && !(eb > ebm));
        bA = stopm ? __builtin_ctzll(stopm) : 7;
    }
    // S3: leaks
    const int seed = start == 0 ? bA - 1 : bndstart;
    const bool in = live && b >= seed;
    int fast = in ? psd + c.fgain - b * c.fdecay : INF, slow = in ? psd + c.sgain - b * c.sdecay : INF;
    wave_incl_scan_min2(fast, slow);                    // (two scans interleaved, wave_ops.h)
    fast += b * c.fdecay;
    slow += b * c.sdecay;
    if (start != 0) {
        const int ff = c.fast + (b - bndstart + 1) * c.fdecay, sl = c.slow + (b - bndstart + 1) * c.sdecay;
        fast = ff < fast ? ff : fast;
        slow = sl < slow ? sl : slow;
    }
    // S4: mask
    int mask = (start == 0 && b < bA) ? psd + c.fgain + lc : (fast + lc < slow ? fast + lc : slow);
    if (live) bmask[b] = (int16_t)ba_mask(c, mask, psd, b);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // S5: bins
    const int shift = bndstart - (int)L.band_of_bin[start];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int bin = 64 * k + lane;
        if (bin >= start && bin < end) bap[bin] = ba_width(L.width, (int)bmask[L.band_of_bin[bin] + shift] + 4 * e[bin]);
    }
}

// ---------------------------------------------------------------------------
// The same allocation with the band PSDs of TWO rows integrated in one sweep (decode.hip): only the 22 bands above bin 27
// are wider than a bin, so lanes 0-31 take the wide bands of one row and lanes 32-63 those of another.  A band's exponents
// are fetched once (seven dwords, shifted so that byte 0 is its first bin): the sweep's only dependent LDS access per step
// is the log-add table.  psd' = min(psd, next) + la_neg[min(|next - psd| >> 1, 255)] is bit_allocate.c:236-251's four
// cases in one expression (la_neg[255] = 0).

// bins of the wide band 28 + (lane & 31) (0, 0 for lanes without one): constant per lane, computed once per kernel
template <class LDS>
__device__ __forceinline__ void ba_lane_band(const LDS &L, int lane, int &lo0, int &hi0)
{
    const int t = lane & 31;
    lo0 = t < 22 ? (int)L.band_end[t + 7] : 0;
    hi0 = t < 22 ? (int)L.band_end[t + 8] : 0;
}

template <class LDS>
__device__ __forceinline__ int ba_wide_psd(const LDS &L, int lo0, int hi0, const uint8_t *eA, int startA, int endA,
                                           const uint8_t *eB, int startB, int endB, bool two, int lane)
{
    const bool upper = lane >= 32;
    const uint8_t *e = upper ? eB : eA;
    const int start = upper ? startB : startA, end = upper ? endB : endA;
    const bool on = (lane & 31) < 22 && (!upper || two);
    int lo = lo0 > start ? lo0 : start;
    const int hi = hi0 < end ? hi0 : end;
    int w = on ? hi - lo : 0;
    w = w > 0 ? w : 0;
    lo = w > 0 ? lo : start;
    const uint32_t *q = reinterpret_cast<const uint32_t *>(e + (lo & ~3));
    uint32_t d[7], a[6];
#pragma unroll
    for (int i = 0; i < 7; i++) d[i] = q[i];
    const uint32_t sh = (uint32_t)lo & 3u;
#pragma unroll
    for (int i = 0; i < 6; i++) a[i] = __builtin_amdgcn_alignbyte(d[i + 1], d[i], sh);
    // Bins past the band's width read as exponent 0xef: a neighbour that far down leaves the sum as it is (the difference
    // clamps to la_neg[255] = 0 and min() keeps psd), so no step needs a test of its own - hipcc had turned `j < w ? nv : psd`
    // into a compare, two exec-mask instructions and a branch per step.  The sweep runs on psd + 2048 (every exponent byte
    // + 16: a sum of 24 bins can fall to -23 x 64 below its smallest term), so that |next - psd| is ONE v_sad_u16.
    int psd = (int)((a[0] & 0xffu) + 16u) << 7;
#pragma unroll
    for (int i = 0; i < 6; i++) {
        int n = w - 4 * i;
        n = n < 0 ? 0 : n > 4 ? 4 : n;
        const uint32_t past = n >= 4 ? 0u : 0xffffffffu << (8 * n);
        a[i] = ((a[i] & ~past) | (0xefefefefu & past)) + 0x10101010u;          // (exponents are 0 .. 24: no carry between bytes)
    }
    auto step = [&](int j) __attribute__((always_inline)) {
        const int next = (int)((a[j >> 2] >> (8 * (j & 3))) & 0xffu) << 7;
        int idx = (int)(__builtin_amdgcn_sad_u16((uint32_t)next, (uint32_t)psd, 0u) >> 1);      // both in 0 .. 2^15
        idx = idx > 255 ? 255 : idx;
        psd = (next < psd ? next : psd) + L.la_neg[idx];
    };
    // bands are 3, 6, 12 or 24 bins wide (fewer at a channel's edges)
    if (__any(w > 1)) { step(1); step(2); }
    if (__any(w > 3)) { step(3); step(4); step(5); }
    if (__any(w > 6)) {
#pragma unroll
        for (int j = 6; j < 12; j++) step(j);
    }
    if (__any(w > 12)) {
#pragma unroll
        for (int j = 12; j < 24; j++) step(j);
    }
    return psd - 2048;
}

// everything after the band PSDs for one row; `wide` = ba_wide_psd's result, this row's in the lanes of half `half`
template <class LDS>
__device__ void bit_allocate_finish(const LDS &L, int16_t *bmask, const BaCtx &c, int bndstart, int start, int end, const uint8_t *e,
                                    int8_t *bap, int wide, int half, int lane)
{
    constexpr int INF = 0x3fffffff, NEG = -0x3fffffff;
    const int b = lane;
    int lo = b < 21 ? b : (int)L.band_end[b < 50 ? b - 21 : 0];
    int hi = b < 20 ? b + 1 : (int)L.band_end[b < 50 ? b - 20 : 0];
    lo = lo > start ? lo : start;
    hi = hi < end ? hi : end;
    const int w = b < 50 ? hi - lo : 0;
    const bool live = w > 0;
    lo = live ? lo : start;
    const int pw = __shfl(wide, (b >= 28 ? b - 28 : 0) + 32 * half, 64);
    const int psd = b >= 28 ? pw : 128 * e[lo];
    // S2: lowcomp and the extent of the first loop
    int lc = 0, bA = 0;
    if (start == 0) {
        const int eb = e[b < 253 ? b : 0], eb1 = e[b < 252 ? b + 1 : 0], ebm = e[b > 0 && b < 254 ? b - 1 : 0];
        const bool upd = b < 20 && (b >= 7 || b < end - 1);
        const bool reset = upd && eb1 == eb - 2;
        const int dec = upd && !reset && eb1 > eb ? 1 : 0;
        const int D = (int)wave_incl_scan_u32((uint32_t)dec);
        const int rix = wave_incl_scan_max(reset ? b : NEG);
        const int Dr = __shfl(D, rix < 0 ? 0 : rix, 64);
        if (rix >= 0) { lc = (rix < 7 ? 384 : 320) - 64 * (D - Dr); lc = lc < 0 ? 0 : lc; }
        const int l19 = __builtin_amdgcn_readlane(lc, 19);
        lc = b == 20 ? (l19 > 128 ? l19 - 128 : 0) : b == 21 ? (l19 > 256 ? l19 - 256 : 0) : b >= 22 ? 0 : lc;
        const unsigned long long stopm = __ballot(b >= 3 && b < 7 && !(eb > ebm));
        bA = stopm ? __builtin_ctzll(stopm) : 7;
    }
    // S3: leaks
    const int seed = start == 0 ? bA - 1 : bndstart;
    const bool in = live && b >= seed;
    int fast = in ? psd + c.fgain - b * c.fdecay : INF, slow = in ? psd + c.sgain - b * c.sdecay : INF;
    wave_incl_scan_min2(fast, slow);                    // (two scans interleaved, wave_ops.h)
    fast += b * c.fdecay;
    slow += b * c.sdecay;
    if (start != 0) {
        const int ff = c.fast + (b - bndstart + 1) * c.fdecay, sl = c.slow + (b - bndstart + 1) * c.sdecay;
        fast = ff < fast ? ff : fast;
        slow = sl < slow ? sl : slow;
    }
    // S4: mask
    int mask = (start == 0 && b < bA) ? psd + c.fgain + lc : (fast + lc < slow ? fast + lc : slow);
    if (live) bmask[b] = (int16_t)ba_mask(c, mask, psd, b);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // S5: bins
    const int shift = bndstart - (int)L.band_of_bin[start];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int bin = 64 * k + lane;
        if (bin >= start && bin < end) bap[bin] = ba_width(L.width, (int)bmask[L.band_of_bin[bin] + shift] + 4 * e[bin]);
    }
}

// ---------------------------------------------------------------------------
// mantissa helpers

__device__ __forceinline__ int16_t dither_at(const uint16_t *lfsr_seq, uint32_t idx0, int k)
{
    uint32_t i = idx0 + (uint32_t)k + 1;
    i %= 65535u;
    const int16_t ns = (int16_t)lfsr_seq[i];
    return (int16_t)((3 * ns) >> 2);
}
__device__ __forceinline__ int16_t dither_value(const DecodeParams &P, uint32_t idx0, int k)
{
    // k-th draw (k = 0 first) = state after k+1 steps
    uint32_t i = idx0 + (uint32_t)k + 1;
    i %= 65535u;
    const int16_t ns = (int16_t)P.lfsr_seq[i];
    return (int16_t)((3 * ns) >> 2);
}


// ---------------------------------------------------------------------------
// The mantissa stage shared by both front ends, four bins per lane, branch-free (L52/parse.c:336-433 coeff_get).
//
// Row bytes (the bap rows in LDS): 0 = no bits, 3..16 = that many plain bits, 32 / 64 / 96 = member of a 3- / 5- /
// 11-level code (ba_width's -1 / -2 / -3, remapped when the width table is staged: remap_width), so that
// plain bits = b & 31 and kind = b >> 5.  The per-code constants come from a 128-entry descriptor table (mant_desc).

__device__ __forceinline__ int8_t remap_width(int w) { return (int8_t)(w >= 0 ? w : -32 * w); }
__device__ __forceinline__ int8_t unmap_width(int b) { return (int8_t)(b >= 32 ? -(b >> 5) : b); }      // liba52's form, for the taps

// descriptor of a row byte: dequantiser table base | members per code << 10 | opener bits << 12 | coded << 15
__host__ __device__ inline uint32_t mant_desc(uint32_t b)
{
    const uint32_t k1 = b >> 5, nbp = b & 31u;
    const uint32_t qbase = k1 == 1 ? 0u : k1 == 2 ? 96u : k1 == 3 ? 480u : nbp == 3 ? 736u : nbp == 4 ? 744u : 0u;
    const uint32_t per = k1 == 3 ? 2u : k1 ? 3u : 0u, obits = k1 == 1 ? 5u : k1 ? 7u : 0u;
    const uint32_t coded = (k1 || nbp == 3 || nbp == 4) ? 1u : 0u;
    return qbase | (per << 10) | (obits << 12) | (coded << 15);
}

// where a segment's mantissas start
struct SegBase {
    uint32_t bit;           // first bit of the segment
    int r3, r5, r11;        // 3/5/11-level mantissas of the block before the segment (global ranks)
    int draw;               // dither draws of the block before the segment
    int mult;               // draws per zero-bit bin of the segment
    uint32_t total_bits;    // of the block (decode_wg.hip's prefix only)
    int total_draws;
};
struct SegTotals {
    uint32_t bits;
    int n3, n5, n11, draws;
};

// registers kept from the first to the second half of the stage for the lane's 4 bins
struct BinRegs {
    uint32_t raw[4];
    uint32_t bap4, exp4;    // row bytes (bins outside the segment zeroed) and exponents
    uint32_t gm[4];         // group | member << 12
    int cd;                 // draw index of the lane's first zero-bit bin
};

// First half: ranks, bit offsets, field extraction; openers publish their codes in gcode[(kind-1)*GCN + (group & GMASK)]
// (gcode[3*GCN] is a sink for lanes that open nothing).  COUNT: offsets and totals only.  `lanes4` = lanes that may read
// their row dword (the short LFE row).
template <int GCN, int GMASK, bool COUNT>
__device__ __forceinline__ SegTotals mant_first_half(const uint8_t *erow, const int8_t *brow, const uint32_t *desc, uint8_t *gcode,
                                                     const uint32_t *frw, uint32_t frw_last, int start, int end, int lanes4,
                                                     const SegBase &sb, BinRegs &R, int lane)
{
    const bool have = lane < lanes4;
    const uint32_t bap4 = have ? *reinterpret_cast<const uint32_t *>(brow + 4 * lane) : 0u;
    R.exp4 = have ? *reinterpret_cast<const uint32_t *>(erow + 4 * lane) : 0u;
    uint32_t b[4], d[4], inc[4], zero[4];
    uint32_t gl = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int bin = 4 * lane + j;
        const uint32_t act = (uint32_t)(bin >= start) & (uint32_t)(bin < end);
        b[j] = ((bap4 >> (8 * j)) & 0xffu) * act;
        zero[j] = act & (uint32_t)(b[j] == 0u);
        d[j] = desc[b[j]];
        inc[j] = (1u << (30u - 10u * (b[j] >> 5))) & 0x3fffffffu;      // 1 in the 10-bit field of the bin's kind (3-level: bits 20-29, 5-level: 10-19, 11-level: 0-9), 0 for plain bins
        gl += inc[j];
    }
    R.bap4 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    const uint32_t gin = wave_incl_scan_u32(gl), gex = gin - gl;
    // phase of each kind at the start of the segment and the group its first member belongs to
    const int q3 = (int)(((uint32_t)sb.r3 * 0xaaabu) >> 17), q5 = (int)(((uint32_t)sb.r5 * 0xaaabu) >> 17), q11 = sb.r11 >> 1;
    uint32_t run = gex + ((uint32_t)(sb.r3 - 3 * q3) << 20) + ((uint32_t)(sb.r5 - 3 * q5) << 10) + (uint32_t)(sb.r11 & 1);
    uint32_t nb[4];
    uint32_t nbsum = 0, ndsum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t k1 = b[j] >> 5;
        const uint32_t x = (run >> (30u - 10u * k1)) & 0x3ffu;        // phase + rank inside the segment (0 for plain bins: bits 30, 31)
        const uint32_t two = (uint32_t)(k1 == 3u);
        const uint32_t q = two ? x >> 1 : (x * 171u) >> 9;            // x / members per code, x < 256 + 3
        const uint32_t mem = x - q * (3u - two);
        const uint32_t opens = (uint32_t)(k1 != 0u) & (uint32_t)(mem == 0u);
        nb[j] = (b[j] & 31u) + opens * ((d[j] >> 12) & 7u);
        const uint32_t gbase = k1 == 1u ? (uint32_t)q3 : k1 == 2u ? (uint32_t)q5 : (uint32_t)q11;
        R.gm[j] = (gbase + q) | (mem << 12);
        run += inc[j];
        nbsum += nb[j];
        ndsum += zero[j] * (uint32_t)sb.mult;
        // the opener publishes its code where the other members will look for it; everybody else writes to the sink
        zero[j] = opens ? (k1 - 1u) * GCN + ((gbase + q) & (uint32_t)GMASK) : 3u * GCN;      // (zero[] reused: gcode slot)
    }
    const uint32_t bl = nbsum | (ndsum << 16);
    const uint32_t bin_ = wave_incl_scan_u32(bl);
    const uint32_t gtot = wave_last(gin), btot = wave_last(bin_);
    SegTotals T;
    T.bits = btot & 0xffffu;
    T.draws = (int)(btot >> 16);
    T.n3 = (int)((gtot >> 20) & 0x3ffu);
    T.n5 = (int)((gtot >> 10) & 0x3ffu);
    T.n11 = (int)(gtot & 0x3ffu);
    R.cd = sb.draw + (int)(bin_ >> 16) - (int)ndsum;
    if (COUNT) return T;
    const uint32_t off = sb.bit + (bin_ & 0xffffu) - nbsum;
    // the lane's fields are at most 64 consecutive bits starting at `off`: a 64-bit window out of three dwords
    uint32_t wi = off >> 5;
    wi = wi < frw_last ? wi : frw_last;
    const uint32_t d0 = frw[wi], d1 = frw[wi + 1], d2 = frw[wi + 2];
    const uint32_t k = off & 31u;
    uint64_t win = ((((uint64_t)d0 << 32) | d1) << k) | (uint64_t)((d2 >> 1) >> (31u - k));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        R.raw[j] = ((uint32_t)(win >> 32) >> 1) >> (31u - nb[j]);       // top nb bits (0 for nb = 0)
        win <<= nb[j];
        gcode[zero[j]] = (uint8_t)R.raw[j];
    }
    return T;
}

// dequantised value of bin j before the exponent / gain scale (0 for a zero-bit bin)
template <int GCN, int GMASK>
__device__ __forceinline__ float mant_value(const BinRegs &R, int j, const uint32_t *desc, const uint8_t *gcode, const float *qtab)
{
    const uint32_t b = (R.bap4 >> (8 * j)) & 0xffu, k1 = b >> 5, nbp = b & 31u;
    const uint32_t d = desc[b];
    const uint32_t grp = R.gm[j] & 0xfffu, mem = R.gm[j] >> 12;
    const uint32_t code = gcode[k1 ? (k1 - 1u) * GCN + (grp & (uint32_t)GMASK) : 3u * GCN];
    const uint32_t coded = (d >> 15) & 1u;
    const uint32_t qi = (d & 0x3ffu) + (k1 ? code * ((d >> 10) & 3u) + mem : R.raw[j]);
    const float tv = qtab[coded ? qi : 0u];
    const float pv = (float)(((int32_t)(R.raw[j] << ((32u - nbp) & 31u))) >> 16);      // two's complement fraction, scaled by 2^15
    return coded ? tv : pv;
}


// ---------------------------------------------------------------------------
// One audio block's mantissas: parse.c:813-879.  Shared by decode.hip's one-kernel front ends (rows and coupling
// coordinates in LDS) and by mant_kernel (rows and coordinates from the parse kernel's workspace in HBM).

// wave-uniform description of the block
struct MantBlk {
    int nf, lfeon, acmod, in_lfe;
    int chincpl, dithmask, rematflg, cplstrtmant, cplendmant;
    int endmant[5];
    float gain[5], lfe_gain;
    // mant_block2 (mant2.h) takes a slot's values from these instead - a slot picked at run time is then one shift / one
    // v_readlane where hipcc built a tree of compares and branches per segment out of `slot == 0 ? gain[0] : ...`:
    uint64_t ends;          // byte k: endmant[k]
    float gainv;            // PER LANE: lane k = gain[k] (k < 5), lane 5 = lfe_gain
};

// segment k of the block in bitstream order -> slot (0..4 fbw, 5 lfe, 6 coupling channel): channel 0, the coupling channel
// right after the first coupled channel, ..., LFE last
__device__ __forceinline__ int seg_slot(int k, int nf, int chincpl, int cplfirst)
{
    if (chincpl) return k <= cplfirst ? k : k == cplfirst + 1 ? 6 : k - 1 < nf ? k - 1 : 5;
    return k < nf ? k : 5;
}

// The segments of the block, one step of four bins per lane each (mant_first_half / mant_value).  Ranks of the grouped
// codes, bit offsets and dither draw indices run on from segment to segment (sb); planes go straight to HBM, 16 bytes per
// lane.  COUNT: ranks, bit offsets and draw counts only.  erow_of(slot) / brow_of(slot): the slot's exponent / bap row;
// cplco_of(c, bnd): coupling coordinate of channel c (gain not applied); cplbnd[18]: sub-band -> band.
template <bool COUNT, class ERow, class BRow, class Cplco>
__device__ __forceinline__ void mant_block(const MantBlk &B, ERow erow_of, BRow brow_of, Cplco cplco_of, const uint8_t *cplbnd,
                                           const uint32_t *desc, uint8_t *gcode, const uint32_t *frw, uint32_t frw_last,
                                           const float *qtab, const uint16_t *lfsr_seq, uint32_t lfsr_i0, bool lfsr_live,
                                           float *cblk, SegBase &sb, int lane)
{
    const int nf = B.nf;
    const int ncpl_dith = __popc(B.chincpl & B.dithmask);
    const int remat_end = B.endmant[0] < B.endmant[1] ? B.endmant[0] : B.endmant[1];
    // Only a damaged frame can put rematrixed bins inside the coupling range (a coupled channel that reuses its
    // exponents keeps the previous block's end): liba52 rematrixes the planes as they stand after coupling and
    // zeroing (parse.c:837-865), so that case runs as a pass of its own after the segments.
    const bool remat_late = B.acmod == 2 && B.rematflg != 0 && B.chincpl != 0 && remat_end > B.cplstrtmant;
    const int cplfirst = B.chincpl ? __builtin_ctz(B.chincpl) : 99;
    const int nseg = nf + (B.chincpl ? 1 : 0) + (B.lfeon ? 1 : 0);
    for (int k = 0; k < nseg; k++) {
        const int slot = seg_slot(k, nf, B.chincpl, cplfirst);
        int start = 0, end, draws = 0;
        float g = 0.f;
        if (slot < 5) {
            end = slot == 0 ? B.endmant[0] : slot == 1 ? B.endmant[1] : slot == 2 ? B.endmant[2] : slot == 3 ? B.endmant[3] : B.endmant[4];
            g = slot == 0 ? B.gain[0] : slot == 1 ? B.gain[1] : slot == 2 ? B.gain[2] : slot == 3 ? B.gain[3] : B.gain[4];
            draws = (B.dithmask >> slot) & 1;
        } else if (slot == 5) {
            end = 7;
            g = B.lfe_gain;
        } else {
            start = B.cplstrtmant;
            end = B.cplendmant;
            draws = ncpl_dith;
        }
        sb.mult = draws;
        BinRegs R;
        const SegTotals T = mant_first_half<GRING, GRING - 1, COUNT>(erow_of(slot), brow_of(slot), desc, gcode, frw,
                                                                    frw_last, start, end, slot == 5 ? LFE_ROW / 4 : 64, sb, R, lane);
        sb.bit += T.bits;
        sb.r3 += T.n3;
        sb.r5 += T.n5;
        sb.r11 += T.n11;
        sb.draw += T.draws;
        if (COUNT) continue;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (slot < 6) {
            float out[4];
            int cd = R.cd;
            const bool dith = draws != 0 && lfsr_live;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int bin = 4 * lane + j;
                const bool zero = bin < end && ((R.bap4 >> (8 * j)) & 0xffu) == 0u;
                const int e = (int)((R.exp4 >> (8 * j)) & 0xffu);
                float q = mant_value<GRING, GRING - 1>(R, j, desc, gcode, qtab);
                if (dith && __any(zero)) {                // wave-uniform: no table access when no lane draws
                    const float dv = (float)dither_at(lfsr_seq, lfsr_i0, cd);
                    q = zero ? dv : q;
                }
                cd += (zero && draws) ? 1 : 0;
                out[j] = q * (sf_of(e) * g);              // (bins past the channel's end have no bits: 0)
            }
            float *plane = cblk + (slot == 5 ? 0 : slot + B.in_lfe) * 256;
            if (slot == 1 && B.acmod == 2 && B.rematflg != 0 && !remat_late) {
                // rematrix: parse.c:837-865.  Channel 0's bins were stored by this same lane.
                float4 a4 = *reinterpret_cast<const float4 *>(plane - 256 + 4 * lane);
                float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int bin = 4 * lane + j;
                    const int band = bin < 25 ? 0 : bin < 37 ? 1 : bin < 61 ? 2 : 3;
                    // (liba52's loop is a do-while, parse.c:846-862: with the first band's flag set it rematrixes bin 13
                    // even when the channels end at or below it - a damaged frame whose block 0 reuses exponents)
                    if (bin >= 13 && (bin < remat_end || (bin == 13 && remat_end <= 13)) && ((B.rematflg >> band) & 1)) {
                        const float x = a[j], v = out[j];
                        a[j] = x + v;
                        out[j] = x - v;
                    }
                }
                *reinterpret_cast<float4 *>(plane - 256 + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
            }
            if (slot < 5 && ((B.chincpl >> slot) & 1)) {
                // a coupled channel: its own bins, zeros up to the coupling range (a damaged frame can leave a gap
                // there: liba52 then keeps the previous block's PCM, its buffer being transformed in place; here
                // zeros) and from the END OF THE COUPLING RANGE on - also where a damaged frame left the channel's own
                // end (it keeps the previous block's when the exponents are reused) beyond it: liba52 zeroes from
                // cplendmant (parse.c:826-834).  The coupling channel's share in between is written by that segment,
                // which comes after the first coupled channel and before the others: as in liba52, a later coupled
                // channel's own bins inside the range win over the coupling channel's, the first one's lose.
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int bin = 4 * lane + j;
                    if (bin < end || bin < B.cplstrtmant || bin >= B.cplendmant) plane[bin] = bin >= B.cplendmant ? 0.f : out[j];
                }
            } else {
                *reinterpret_cast<float4 *>(plane + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
            }
        } else {
            // coupling channel: parse.c:435-556
            int cd = R.cd;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int bin = 4 * lane + j;
                const bool in = bin >= start && bin < end;
                const bool zero = in && ((R.bap4 >> (8 * j)) & 0xffu) == 0u;
                const int e = (int)((R.exp4 >> (8 * j)) & 0xffu);
                const float m = mant_value<GRING, GRING - 1>(R, j, desc, gcode, qtab) * sf_of(e);
                const int bnd = cplbnd[in ? (bin - start) / 12 : 0];
                int cdc = cd;
                for (int c = 0; c < nf; c++) {
                    if (!((B.chincpl >> c) & 1)) continue;
                    const float gc = c == 0 ? B.gain[0] : c == 1 ? B.gain[1] : c == 2 ? B.gain[2] : c == 3 ? B.gain[3] : B.gain[4];
                    const float co = cplco_of(c, bnd) * gc;
                    float v = m * co;
                    if (zero) {
                        v = 0.f;
                        if ((B.dithmask >> c) & 1) { v = (sf_of(e) * co) * (float)(lfsr_live ? dither_at(lfsr_seq, lfsr_i0, cdc) : 0); cdc++; }
                    }
                    if (in) cblk[(c + B.in_lfe) * 256 + bin] = v;
                }
                cd += zero ? draws : 0;
            }
        }
    }
    if (!COUNT && remat_late) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        volatile float *p0 = cblk + (size_t)B.in_lfe * 256, *p1 = p0 + 256;
        for (int bin = 13 + lane; bin < remat_end; bin += 64) {
            const int band = bin < 25 ? 0 : bin < 37 ? 1 : bin < 61 ? 2 : 3;
            if ((B.rematflg >> band) & 1) {
                const float a = p0[bin], v = p1[bin];
                p0[bin] = a + v;
                p1[bin] = a - v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// The split front end: decode_kernel<4 / 5> parses (side information, exponents, bit allocation) and leaves, per audio
// block, this descriptor plus the exponent / bap rows that changed; mant_kernel (one wavefront per block) unpacks and
// dequantises from them.  What a block needs of the blocks before it is all here: where its mantissas start, how many
// dither draws came before, which rows are current.

struct alignas(16) BlkDesc {
    uint32_t bitpos;            // first mantissa bit of the block
    uint32_t draw_off;          // dither draws of the frame before this block
    uint32_t flags;             // bit 0: the block failed (zero planes); 8-12 chincpl; 16-20 dither flags; 24-27 rematflg
    uint32_t cplbndstrc;
    uint16_t endmant[5], cplstrt, cplend, pad0;
    uint8_t rv_exp[8], rv_bap[8];   // block of THIS frame whose row holds the slot's current exponents / bap
    float gain[5], lfe_gain;        // parse.c:810-811 (dynamic range folded in); LFE: 0 when it is not an output
    uint32_t src_snr;           // 16 csnroffst + fsnroffst (channel 0) the frame was coded with, as of this block
    uint32_t pad1;
};
static_assert(sizeof(BlkDesc) == 80 && offsetof(BlkDesc, src_snr) == 72, "BlkDesc layout");

constexpr int ROWSET = 7 * 512;         // bytes of one (frame, block) row set: 7 slots x (256 exponents + 256 row bytes)

// totals of one slot's row over [start, end): a = plain bits | 3-level members << 13 | 5-level members << 22,
// b = 11-level members | zero-bit bins << 9.  Wave-uniform.
struct RowTotals { uint32_t a, b; };
__device__ __forceinline__ RowTotals row_totals(const int8_t *brow, int start, int end, int lanes4, int lane)
{
    const uint32_t bap4 = lane < lanes4 ? *reinterpret_cast<const uint32_t *>(brow + 4 * lane) : 0u;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int bin = 4 * lane + j;
        const uint32_t act = (uint32_t)(bin >= start) & (uint32_t)(bin < end);
        const uint32_t x = (bap4 >> (8 * j)) & 0xffu, k1 = x >> 5;
        a += act * ((x & 31u) + ((uint32_t)(k1 == 1u) << 13) + ((uint32_t)(k1 == 2u) << 22));
        b += act * ((uint32_t)(k1 == 3u) + ((uint32_t)(x == 0u) << 9));
    }
    RowTotals t;
    t.a = wave_sum_u32(a);
    t.b = wave_sum_u32(b);
    return t;
}

}  // namespace ac3mi
