// encode.hip — AC-3 encoder on gfx950, bit-exact restatement of ENC/ac3enc.cpp (AC3_encode_frame, :1640-1763).
//
//  enc_mdct_kernel     one wavefront per (stream, frame, channel): gather/deinterleave via chmap, Q15 window,
//                      block-floating-point normalisation, the reference's 16-bit radix-2 DIT FFT in registers (one
//                      butterfly per lane and pass with exactly the reference's operands, shifts and int16
//                      truncations; lanes trade one point per pass), post-rotation, exponent extraction
//                      [ac3enc.cpp:1665-1722, 462-603]; then, still per channel (exp_stage): exponent strategy,
//                      min-merge over reuse runs, the +-2 constraint in closed form, band PSDs and masking curves of
//                      the blocks that send exponents                                   [ac3enc.cpp:606-761, 220-367]
//  enc_search_kernel   <1>: one wavefront per stream, frames in order (the SNR-offset search starts from the previous
//                      frame's csnroffst): the reference's exact search sequence replayed from exact verdicts, three
//                      offsets costed per sweep over the frame's run-start rows; <3>: one wavefront per frame
//                      tabulates verdicts for few long streams                          [ac3enc.cpp:764-975]
//  enc_packf_kernel    one wavefront per frame packs it: header, side information, exponent groups, the mantissas
//                      (enc_mant.h), both CRCs by per-lane chunk CRC + GF(2) combine    [ac3enc.cpp:1113-1638]
//  enc_packb_kernel    the same with a workgroup of six wavefronts per frame, one per audio block (small batches)
//
// Integer arithmetic only (no -ffast-math dependence); the Q15 tables come from the host (capi.hip)
// with the reference's expressions.
#include "ac3mi_internal.h"
#include "wave_ops.h"
#include "spec_tables.h"
#include "enc_mant.h"

namespace ac3mi {

#define WAVE_SYNC()                                          \
    do {                                                     \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                     \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
    } while (0)

__device__ __forceinline__ int ilog2u(unsigned v) { return v ? 31 - __builtin_clz(v) : 0; }   // av_log2, :1539-1567

__device__ __forceinline__ int wave_sum(int v) { return (int)wave_sum_u32((uint32_t)v); }

// ---------------------------------------------------------------------------------------------
// kernel 2: exponent coding, bit allocation, quantisation, packing

struct PackParams {
    const int32_t *mdct;
    const uint8_t *eexp;        // [S][F][6][nch][256] encoded exponents   (exp_stage)
    const int16_t *emask;       // [S][F][6][nch][50]  masking curve minus the floor
    const uint8_t *strat;       // [S][F][6][nch]
    const int32_t *ebits;       // [S][F][nch] bits of the coded exponents
    const int8_t *shift;
    int32_t *csnr_state;        // [S] in/out
    const int32_t *slot;        // optional: stream s uses csnr_state[slot[s]]
    int32_t *snr;               // [S][F][2] csnroffst, fsnroffst between the parts of the split kernel
    uint32_t *memo;             // [S][F][8] tabulated fit verdicts (PART 3 -> PART 1): known_c, fits_c (64 bit each), known_f, fits_f, f_cc
    uint8_t *frames;            // [S][F][stride]
    const EncTables *tab;
    // taps (optional)
    uint8_t *tap_eexp;          // [S][F][6][nch][256]
    uint8_t *tap_bap;           // [S][F][6][nch][256]
    uint8_t *tap_strat;         // [S][F][6][nch]
    int32_t *tap_snr;           // [S][F][2]
    int n_streams, frames_per_stream, frame_stride;
    int nch, nfbw, lfe, acmod, fscod, halfrate, bsid, frmsizecod, frame_words;
    int nbc;                    // coefficients per full-bandwidth channel (223)
    int chbwcod;
    // CRC constants as multiplication tables: entry i = constant * x^i mod poly, so that a product with a per-lane value is 16
    // select-and-xor steps with no scalar chain (gf_mul_tab)
    uint16_t crc_inv_t[16];     // x^-(16*fs58-16) mod poly (:1627)
    uint16_t pw1_t[6][16], pw2_t[6][16];    // x^(8*C*2^k) mod poly for the two CRC regions
    int c1, c2;                 // CRC chunk bytes per lane
    int frw;                    // dwords of the frame buffer in (dynamic) LDS: the frame + 256 bytes of headroom, multiple of 4
    int marker;                 // the reference's "member already merged" value, 128 (:1375-1413); AC3MI_ENC_MARKER: test aid
    const uint32_t *hint;       // optional, [frame * hint_stride]: 16 csnroffst + fsnroffst the frame's SOURCE was coded with (transcode)
    int hint_stride;
};



// tables of the PSD / masking-curve computation
struct MaskTabs {
    uint8_t latab[256];
    uint16_t hth[50];
    uint8_t band_of_bin[256];
    uint8_t band_start[52];
};

// The search kernel's LDS: the frame's 36 masking curves, the cost table and the list of run-start rows.
struct alignas(16) SearchLDS {
    int16_t mask[36][50];       // masking curve minus the floor, row blk * nch + ch as exp_stage leaves them (a straight copy)
    uint32_t bitlut[64];        // see "bap of one coefficient" below
    uint8_t strat[6][6];
    uint8_t band_of_bin[256];
    // the frame's run-start rows in (block, channel) order, for the search's sweeps: byte offset of the row's encoded
    // exponents (bits 0-13) | LFE row (7 coefficients, bit 14) | blocks the run covers (bits 16-21) | mask row (24-29)
    uint32_t rowdesc[36];
    // per band of the row being costed (two buffers: the next row's are worked out a row ahead), for each candidate offset:
    // 320 - max(0, ((mask - snroffset) >> 3) & ~3) as int16, so that a coefficient's table address x 4 is
    // clamp(term - 16 exponent, 0, 252)
    uint2 terms[2][50];
};

// put_bits (:148-176).  `v` may be wider than n bits (the release build does not mask it): the excess is OR-ed onto
// the bits before the field as far as the 32-bit word it starts in, which is what the 64-bit shift below does.
__device__ __forceinline__ void put_bits(uint32_t *fr, int frw, uint32_t pos, int n, uint32_t v)
{
    if (n <= 0) return;
    const uint32_t w = pos >> 5;
    if (w + 1 >= (uint32_t)frw) return;
    const uint64_t x = (uint64_t)v << (64 - n - (pos & 31));
    const uint32_t hi = (uint32_t)(x >> 32), lo = (uint32_t)x;
    if (hi) atomicOr(&fr[w], hi);
    if (lo) atomicOr(&fr[w + 1], lo);
}

// ---------------------------------------------------------------------------------------------
// exp_stage's routines, wavefront-wide.

// per byte (values < 128): b's byte where it is smaller and `where` selects the byte, else a's
__device__ __forceinline__ uint32_t bytes_min_where(uint32_t a, uint32_t b, uint32_t where)
{
    const uint32_t ge = (((a | 0x80808080u) - b) >> 7) & 0x01010101u;       // 1: a >= b (no borrow crosses a byte)
    const uint32_t m = ((ge << 8) - ge) & where;            // ge * 0xff without the quarter-rate v_mul_lo_u32
    return (b & m) | (a & ~m);
}

// encode_exp (:684-761) on one 256-byte row in LDS, a dword per lane.  Lane l takes bins 4l+1 .. 4l+4: four entries for D15,
// two for D25, one for D45 (entry i covers bins 1 + (i-1) gs .. + gs-1, so groups never straddle lanes).  The +-2 delta
// constraint, min over j of g[j] + 2|i-j|, is a prefix minimum of g[j] - 2j followed by a suffix minimum of g[j] + 2j: inside
// the lane first, then ONE scan over the lane totals per direction.
// PER = entries per lane, a compile-time constant per strategy (D15 4, D25 2, D45 1): a reuse-poor frame (decoded audio
// re-encoded sends 24 sets, most of them D45 / D25) spends its exponent stage here, and the coarse strategies need a quarter /
// half of the per-entry work.
template <int PER>
__device__ __forceinline__ int encode_exp_wave_t(uint8_t *row, int n, int lane)
{
    constexpr int INF = 0x3fffffff;
    constexpr int gs = PER == 4 ? 1 : PER == 2 ? 2 : 4;
    const int ng = ((n + gs * 3 - 4) / (3 * gs)) * 3;
    uint32_t *q = reinterpret_cast<uint32_t *>(row);
    const uint32_t own = q[lane], nxt = q[lane + 1];        // lane 63 reads the dword behind the row: entries beyond ng, never used
    const uint32_t S = __builtin_amdgcn_alignbyte(nxt, own, 1u);
    const int b0 = (int)(S & 0xffu), b1 = (int)((S >> 8) & 0xffu), b2 = (int)((S >> 16) & 0xffu), b3 = (int)(S >> 24);
    int row0 = (int)(__builtin_amdgcn_readfirstlane((int)own) & 0xff);
    row0 = row0 > 15 ? 15 : row0;
    int g[PER], ix[PER];
    bool valid[PER];
    if (PER == 4) { g[0] = b0; g[PER > 1 ? 1 : 0] = b1; g[PER > 2 ? 2 : 0] = b2; g[PER > 3 ? 3 : 0] = b3; }
    else if (PER == 2) { g[0] = b1 < b0 ? b1 : b0; g[PER > 1 ? 1 : 0] = b3 < b2 ? b3 : b2; }
    else { const int m01 = b1 < b0 ? b1 : b0, m23 = b3 < b2 ? b3 : b2; g[0] = m23 < m01 ? m23 : m01; }
#pragma unroll
    for (int c = 0; c < PER; c++) {
        ix[c] = 2 * (PER * lane + 1 + c);
        valid[c] = PER * lane + 1 + c <= ng;
    }
    int t[PER];
#pragma unroll
    for (int c = 0; c < PER; c++) {
        const int a = valid[c] ? g[c] - ix[c] : INF;
        t[c] = c == 0 ? a : (a < t[c > 0 ? c - 1 : 0] ? a : t[c > 0 ? c - 1 : 0]);
    }
    {
        int ex = __builtin_amdgcn_update_dpp(INF, wave_incl_scan_min(t[PER - 1]), 0x138, 0xf, 0xf, false);       // wave_shr:1
        ex = row0 < ex ? row0 : ex;
#pragma unroll
        for (int c = 0; c < PER; c++) g[c] = (t[c] < ex ? t[c] : ex) + ix[c];
    }
#pragma unroll
    for (int c = PER - 1; c >= 0; c--) {
        const int a = valid[c] ? g[c] + ix[c] : INF;
        t[c] = c == PER - 1 ? a : (a < t[c < PER - 1 ? c + 1 : c] ? a : t[c < PER - 1 ? c + 1 : c]);
    }
    const int suf = wave_suffix_scan_min(t[0], lane);
    {
        const int ex = __builtin_amdgcn_update_dpp(INF, suf, 0x130, 0xf, 0xf, false);                       // wave_shl:1
#pragma unroll
        for (int c = 0; c < PER; c++) g[c] = (t[c] < ex ? t[c] : ex) - ix[c];
    }
    const int head = __builtin_amdgcn_readfirstlane(suf);
    const int row0new = head < row0 ? head : row0;
    // back to bins (entries beyond ng leave their bins alone)
    int o0, o1, o2, o3;
    if (PER == 4) {
        o0 = valid[0] ? g[0] : b0; o1 = valid[PER > 1 ? 1 : 0] ? g[PER > 1 ? 1 : 0] : b1;
        o2 = valid[PER > 2 ? 2 : 0] ? g[PER > 2 ? 2 : 0] : b2; o3 = valid[PER > 3 ? 3 : 0] ? g[PER > 3 ? 3 : 0] : b3;
    } else if (PER == 2) {
        o0 = valid[0] ? g[0] : b0; o1 = valid[0] ? g[0] : b1;
        o2 = valid[PER > 1 ? 1 : 0] ? g[PER > 1 ? 1 : 0] : b2; o3 = valid[PER > 1 ? 1 : 0] ? g[PER > 1 ? 1 : 0] : b3;
    } else {
        o0 = valid[0] ? g[0] : b0; o1 = valid[0] ? g[0] : b1; o2 = valid[0] ? g[0] : b2; o3 = valid[0] ? g[0] : b3;
    }
    const uint32_t o = (uint32_t)o0 | (uint32_t)o1 << 8 | (uint32_t)o2 << 16 | (uint32_t)o3 << 24;
    const uint32_t prev = (uint32_t)__builtin_amdgcn_update_dpp(row0new << 24, (int)o, 0x138, 0xf, 0xf, false);
    q[lane] = __builtin_amdgcn_alignbyte(o, prev, 3u);
    return 4 + (ng / 3) * 7;
}

__device__ int encode_exp_wave(uint8_t *row, int n, int strategy, int lane)
{
    if (strategy == 1) return encode_exp_wave_t<4>(row, n, lane);
    if (strategy == 2) return encode_exp_wave_t<2>(row, n, lane);
    return encode_exp_wave_t<1>(row, n, lane);
}

// Masking curve of one row, one lane per band (:220-367).  bndpsd[] must hold the band PSDs.
//  * lowcomp is a reset-or-decrement automaton: value = max(0, R(last reset) - 64 * decrements since)
//  * the fast / slow leaks are prefix maxima of psd - gain + band * decay, seeded at band begin-1
__device__ void mask_row_wave(const MaskTabs &T, int16_t *bndpsd, int bndend, bool is_lfe, int sdecay, int fdecay, int sgain,
                              int dbknee, int fgain, int halfrate, int lane)
{
    constexpr int NEG = -0x3fffffff;
    const int b = lane;
    const int cur = b < bndend ? bndpsd[b] : 0;
    const int nxt = (b + 1 < bndend && !(is_lfe && b == 6)) ? bndpsd[b + 1] : 0;
    const bool skip = (is_lfe && b == 6) || b >= 22 || b >= bndend;
    const bool reset = !skip && b < 20 && cur + 256 == nxt;
    const int dec = skip || reset ? 0 : b >= 20 ? 2 : (cur > nxt ? 1 : 0);
    const int D = (int)wave_incl_scan_u32((uint32_t)dec);
    const int rix = wave_incl_scan_max(reset ? b : NEG);
    const int Dr = __shfl(D, rix < 0 ? 0 : rix, 64);
    int lowcomp = 0;
    if (rix >= 0) { lowcomp = (rix < 7 ? 384 : 320) - 64 * (D - Dr); lowcomp = lowcomp < 0 ? 0 : lowcomp; }

    const bool stop = b >= 2 && b < 7 && b < bndend && !(is_lfe && b == 6) && cur <= nxt;
    const unsigned long long sm = __ballot(stop);
    const int begin = sm ? __builtin_ctzll(sm) + 1 : 7;
    const bool live = b >= begin - 1 && b < bndend;
    int fast = live ? cur - fgain + b * fdecay : NEG, slow = live ? cur - sgain + b * sdecay : NEG;
    wave_incl_scan_max2(fast, slow);                    // (two scans interleaved, wave_ops.h)
    fast -= b * fdecay;
    slow -= b * sdecay;
    int excite;
    if (b < begin) excite = (int16_t)(cur - fgain - lowcomp);
    else if (b < 22) { const int v = fast - lowcomp; excite = (int16_t)(slow > v ? slow : v); }
    else excite = (int16_t)(fast > slow ? fast : slow);
    const int t = dbknee - cur;
    if (t > 0) excite += t >> 2;
    const int h = T.hth[(b < 50 ? b : 49) >> halfrate];
    if (b < 50) bndpsd[b] = b < bndend ? (int16_t)(excite > h ? excite : h) : (int16_t)0;
}

// ---------------------------------------------------------------------------------------------
// second half of kernel 1: exponent strategy, min-merge, constraint and masking curves.  Everything here is
// independent per channel, like the MDCT: the wavefront that transformed the six blocks of one
// (stream, frame, channel) goes straight on with their exponents.

struct ExpParams {
    uint8_t *eexp;              // [S][F][6][nch][256] encoded exponents
    int16_t *emask;             // [S][F][6][nch][50]
    uint8_t *strat;             // [S][F][6][nch]
    int32_t *ebits;             // [S][F][nch]
    const EncTables *tab;
    int nch, lfe, fscod, halfrate, nbc;
};

struct ExpLDS {
    uint8_t E[6][256];
    int16_t mask[6][50];
    MaskTabs t;
    uint8_t strat[8];
};

// Runs at the end of enc_mdct_kernel: L.t holds the tables, L.E the six blocks' raw exponents of this channel.
__device__ void exp_stage(const ExpParams &P, ExpLDS &L, size_t fidx, int ch, int lane)
{
    const int nch = P.nch;
    const bool is_lfe = P.lfe && ch == nch - 1;
    const int n = is_lfe ? 7 : P.nbc;
    uint32_t raw[6];
#pragma unroll
    for (int b = 0; b < 6; b++) raw[b] = *reinterpret_cast<const uint32_t *>(&L.E[b][4 * lane]);

    // ---- exponent strategy (:617-669): sum of |differences| over all 256 bins against the block before (two sums per
    //      reduction: each stays below 2^16) ----
    int st[6];
    st[0] = 1;
    {
        const uint32_t d1 = __builtin_amdgcn_sad_u8(raw[1], raw[0], 0u), d2 = __builtin_amdgcn_sad_u8(raw[2], raw[1], 0u);
        const uint32_t d3 = __builtin_amdgcn_sad_u8(raw[3], raw[2], 0u), d4 = __builtin_amdgcn_sad_u8(raw[4], raw[3], 0u);
        const uint32_t d5 = __builtin_amdgcn_sad_u8(raw[5], raw[4], 0u);
        const uint32_t t12 = wave_sum_u32(d1 | d2 << 16), t34 = wave_sum_u32(d3 | d4 << 16), t5 = wave_sum_u32(d5);
        st[1] = (t12 & 0xffffu) > 1000u; st[2] = (t12 >> 16) > 1000u;
        st[3] = (t34 & 0xffffu) > 1000u; st[4] = (t34 >> 16) > 1000u;
        st[5] = t5 > 1000u;
    }
    uint32_t starts = 0;                                        // bit b: block b sends exponents (wave-uniform)
#pragma unroll
    for (int b = 0; b < 6; b++) starts |= (st[b] != 0 ? 1u : 0u) << b;
    if (!is_lfe) {
#pragma unroll
        for (int b = 0; b < 6; b++) {
            if (st[b] == 0) continue;
            int run = 1;                                                // blocks until the next new set
#pragma unroll
            for (int e = 1; e < 6; e++) if (b + e < 6 && run == e && st[b + e] == 0) run = e + 1;
            st[b] = run == 1 ? 3 : run <= 3 ? 2 : 1;
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int b = 0; b < 6; b++) L.strat[b] = (uint8_t)st[b];
    }

    // ---- min-merge over reuse runs (:1731-1737), bins below n only: in registers, four bins per lane ----
    {
        uint32_t below = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) below |= (4 * lane + c < n ? 0xffu : 0u) << (8 * c);
#pragma unroll
        for (int b = 0; b < 5; b++) {
            if (!((starts >> b) & 1u)) continue;
            bool open = true;
            bool touched = false;
#pragma unroll
            for (int e = b + 1; e < 6; e++) {
                open = open && !((starts >> e) & 1u);
                if (open) { raw[b] = bytes_min_where(raw[b], raw[e], below); touched = true; }
            }
            if (touched) *reinterpret_cast<uint32_t *>(&L.E[b][4 * lane]) = raw[b];
        }
    }
    WAVE_SYNC();

    // ---- encode_exp on run starts (:1739-1746); the blocks of a run are sent the start's exponents ----
    int exp_bits = 0;
    for (uint32_t m = starts; m; m &= m - 1) {
        const int b = __builtin_ctz(m);
        exp_bits += encode_exp_wave(L.E[b], n, L.strat[b], lane);
    }
    WAVE_SYNC();
    {
        uint32_t cur = 0;
        for (int b = 0; b < 6; b++) {
            if ((starts >> b) & 1u) cur = *reinterpret_cast<const uint32_t *>(&L.E[b][4 * lane]);
            *reinterpret_cast<uint32_t *>(P.eexp + ((fidx * 6 + b) * nch + ch) * 256 + 4 * lane) = cur;
        }
    }

    // ---- band PSDs (:220-258) and masking curves (:259-367) of the blocks that send exponents; a block that
    //      reuses them has the same curve.  Bands 0..27 are single bins; the 22 wider ones are integrated by
    //      one lane each, two rows per sweep ----
    {
        const int nsingle = n < 28 ? n : 28;
        for (uint32_t m = starts; m; m &= m - 1) {
            const int r = __builtin_ctz(m);
            if (lane < nsingle) L.mask[r][lane] = (int16_t)(3072 - ((int)(int8_t)L.E[r][lane] << 7));
        }
        if (n > 28) {
            // Ten lanes per row, six rows in ONE sweep of 23 steps: a lane integrates a stretch of equally wide bands one after
            // the other (bands 28..49 start at 28 31 .. 46 | 49 55 .. 79 | 85 97 109 121 | 133 157 181 205 229): slots 0-4 a
            // 24-bin band each, 5-6 two 12-bin bands, 7 four and 8 two 6-bin bands, 9 the seven 3-bin bands - at most 24 bins.
            const int r = (lane * 205) >> 11, slot = lane - 10 * r;                 // lane / 10, lane % 10
            const int fb = slot < 5 ? 45 + slot : slot == 5 ? 41 : slot == 6 ? 43 : slot == 7 ? 35 : slot == 8 ? 39 : 28;
            const int nb = slot < 5 ? 1 : slot < 7 ? 2 : slot == 7 ? 4 : slot == 8 ? 2 : 7;
            const int lw = slot < 5 ? 3 : slot < 7 ? 2 : slot < 9 ? 1 : 0;          // bands of 3 << lw bins
            const bool rowon = lane < 60 && ((starts >> r) & 1u);
            const int rr = rowon ? r : 0;
            const int start = L.t.band_start[fb];
            int end1 = L.t.band_start[fb + nb];
            end1 = end1 < n ? end1 : n;
            const int len = rowon ? end1 - start : 0;                               // bins of the stretch inside the channel
            // the stretch's exponents: seven dwords of the row, shifted so that byte 0 is its first bin; the sweep's only
            // dependent LDS access per step is then the log-add table (as in the decoder's bit allocation, decode_common.h)
            const uint32_t *q = reinterpret_cast<const uint32_t *>(&L.E[rr][start & ~3]);
            uint32_t dw[7], ab[6];
#pragma unroll
            for (int i = 0; i < 7; i++) dw[i] = q[i];
#pragma unroll
            for (int i = 0; i < 6; i++) ab[i] = __builtin_amdgcn_alignbyte(dw[i + 1], dw[i], (uint32_t)start & 3u);
            int16_t *mrow = &L.mask[rr][fb];
            const bool r3 = lw == 0, r6 = lw <= 1, r12 = lw <= 2;
            int v = 3072 - ((int)(ab[0] & 0xffu) << 7);
#pragma unroll
            for (int j = 1; j < 24; j++) {
                const int pj = 3072 - ((int)((ab[j >> 2] >> (8 * (j & 3))) & 0xffu) << 7);
                // |v - pj| in one v_sad_u16 and the larger of the two in one v_max (both are 0 .. 2^15 here: exponents are
                // 0 .. 24 and a band's sum stays below 3072 + 23 x 64; hipcc spent seven instructions on the two)
                int t = (int)(__builtin_amdgcn_sad_u16((uint32_t)v, (uint32_t)pj, 0u) >> 1);
                t = t > 255 ? 255 : t;
                int nv = (v > pj ? v : pj) + (int)L.t.latab[t];
                if (j % 3 == 0) {                                                   // a band may end here
                    const bool ends = j % 12 == 0 ? r12 : j % 6 == 0 ? r6 : r3;
                    if (ends && j < len) mrow[((j / 3) >> lw) - 1] = (int16_t)v;
                    nv = ends ? pj : nv;
                }
                v = j < len ? nv : v;
            }
            if (len > 0) mrow[((((uint32_t)(len - 1)) * 0xaaabu) >> 17) >> lw] = (int16_t)v;
        }
        WAVE_SYNC();
        // fixed allocation codes (:861-879)
        const int sdecaycod = 2, fdecaycod = 1, fgaincod = 4;
        constexpr int sgaincod = 1, dbkneecod = 2;
        const int sdecay = enc_sdecay(sdecaycod) >> P.halfrate, fdecay = enc_fdecay(fdecaycod) >> P.halfrate;
        const int sgain = enc_sgain(sgaincod), dbknee = enc_dbknee(dbkneecod), fgain = enc_fgain(fgaincod);
        const int bndend = L.t.band_of_bin[n - 1] + 1;
        for (uint32_t m = starts; m; m &= m - 1)
            mask_row_wave(L.t, L.mask[__builtin_ctz(m)], bndend, is_lfe, sdecay, fdecay, sgain, dbknee, fgain, P.halfrate, lane);
    }
    WAVE_SYNC();

    // ---- results: the pack kernel wants the masks minus the floor (floorcod 4: 0x1f0); a block inside a run has its start's
    //      curve.  Two rows of 25 dwords per step ----
    static_assert(enc_floor(4) == 0x1f0 && enc_sgain(1) == 0x4d8 && enc_dbknee(2) == 0x900, "ENC/ac3tab.h:151-161");
    {
        typedef short short2v __attribute__((ext_vector_type(2)));
        const int half = lane >= 25 ? 1 : 0, k2 = lane - 25 * half;
        int src[6];
        src[0] = 0;
#pragma unroll
        for (int b = 1; b < 6; b++) src[b] = ((starts >> b) & 1u) ? b : src[b - 1];
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int b = 2 * it + half, sr = half ? src[2 * it + 1] : src[2 * it];
            if (lane < 50) {
                short2v v = *reinterpret_cast<const short2v *>(&L.mask[sr][2 * k2]);
                v -= (short2v){(short)enc_floor(4), (short)enc_floor(4)};
                *reinterpret_cast<short2v *>(P.emask + ((fidx * 6 + b) * nch + ch) * 50 + 2 * k2) = v;
            }
        }
    }
    if (lane < 6) P.strat[(fidx * 6 + lane) * nch + ch] = L.strat[lane];
    if (lane == 0) P.ebits[fidx * nch + ch] = exp_bits;
}

// ---------------------------------------------------------------------------------------------
// kernel 1: window + normalise + MDCT + exponents

struct MdctParams {
    const int16_t *pcm;         // [S][F][1536][nch] interleaved
    int16_t *last;              // [S][nch][256] history before frame 0 (rewritten here when store_history)
    int32_t *mdct;              // [S][F][6][nch][256]
    uint8_t *expo;              // [S][F][6][nch][256]
    int8_t *shift;              // [S][F][6][nch]
    const EncTables *tab;
    int n_streams, frames, nch;
    uint8_t chmap[8];
    const int32_t *slot;        // optional: stream s keeps its history in slot[s] (stride 6*256 samples)
    int store_history;          // one frame per stream: this kernel also leaves the new history (else enc_history_kernel)
    int full_rows;              // store all 256 coefficients of a row (stage tap); else only the bins the packer codes
    ExpParams x;                // the exponent stage that follows the transform
};

struct c16 { int16_t re, im; };

__device__ __forceinline__ void bfly(c16 &p, c16 &q, int bx, int by, int ax, int ay)
{
    p.re = (int16_t)((bx + ax) >> 1);
    p.im = (int16_t)((by + ay) >> 1);
    q.re = (int16_t)((bx - ax) >> 1);
    q.im = (int16_t)((by - ay) >> 1);
}

#ifndef ENC_MDCT_LB
#define ENC_MDCT_LB 7        // 71 VGPRs, no scratch: 1.97 ms against 2.00 at 5 or 6 (75 VGPRs) and 2.07 at 8 (64 VGPRs, 24 bytes of scratch)
#endif
__global__ __launch_bounds__(64, ENC_MDCT_LB) void enc_mdct_kernel(const MdctParams P)
{
    __shared__ int32_t out[256];
    __shared__ ExpLDS XL;

    const int lane = threadIdx.x;
    // The nch wavefronts of a frame de-interleave the same PCM lines, one channel each.  Workgroups are dealt round-robin over the
    // 8 XCDs (private L2s), so with unit = blockIdx the channels of a frame sat on different XCDs and each fetched the frame's
    // samples for itself (PMC: 114 KB fetched per frame against 18 KB of PCM).  The bijective remap of cdna_hip_programming.md
    // (T1) makes consecutive units share an XCD: blocks with equal blockIdx % 8 take one contiguous range of units.
    int unit;                                       // (s*F + f)*nch + ch
    {
        const unsigned n = gridDim.x, q = n >> 3, r = n & 7u, x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        unit = (int)((x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i);
        unit = __builtin_amdgcn_readfirstlane(unit);        // (wave-uniform by construction: row and state addresses on the scalar unit)
    }
    const int ch = unit % P.nch;
    const int sf = unit / P.nch;
    const int f = sf % P.frames;
    const int s = sf / P.frames;

    // Which samples a lane owns.  Point i of the pre-rotation (:578-591) takes four samples of the 512 windowed ones:
    //   i < 64 :  re <- -in[384 + 2i], in[383 - 2i]     im <- in[128 + 2i], in[127 - 2i]
    //   i >= 64:  re <-  in[2i - 128], in[383 - 2i]     im <- in[128 + 2i], -in[639 - 2i]
    // so a lane that holds positions {2L, 127 - 2L, 128 + 2L, 255 - 2L} of the old AND of the new half has every operand of
    // the points L and L + 64 in its own registers; and with L = the lane number's six bits reversed those two points are
    // the ones the bit-reversed order (:496-504) puts at 2 lane and 2 lane + 1: its first butterfly.  Windowing, block floating
    // point, rotation and the reordering then need no LDS at all (same operands, same 16-bit truncations).
    const int L = (int)(__builtin_bitreverse32((unsigned)lane) >> 26);
    const int jpos[4] = {2 * L, 127 - 2 * L, 128 + 2 * L, 255 - 2 * L};
    int wa[4], wb[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { wa[k] = P.tab->win[jpos[k]]; wb[k] = P.tab->win[255 - jpos[k]]; }
    static_assert(offsetof(EncTables, latab) % 4 == 0 && offsetof(EncTables, band_of_bin) % 4 == 0 && offsetof(MaskTabs, band_of_bin) % 4 == 0, "dword copies");
    reinterpret_cast<uint32_t *>(XL.t.latab)[lane] = reinterpret_cast<const uint32_t *>(P.tab->latab)[lane];
    reinterpret_cast<uint32_t *>(XL.t.band_of_bin)[lane] = reinterpret_cast<const uint32_t *>(P.tab->band_of_bin)[lane];
    if (lane < 50) XL.t.hth[lane] = P.tab->hth[lane][P.x.fscod];
    if (lane < 52) XL.t.band_start[lane] = lane < 51 ? P.tab->band_start[lane] : 0;
    // rotation factors of the points the lane rotates BEFORE the passes (L, L + 64) and AFTER them (lane, lane + 64; :596-602),
    // and the twiddles of its butterfly in passes 2..6 (:533-567: twiddle index (lane mod nloops) * nblocks): constant per lane
    int xcv[2], xsv[2], pcv[2], psv[2], twc[5], tws[5];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        xcv[k] = P.tab->xcos[lane + 64 * k]; xsv[k] = P.tab->xsin[lane + 64 * k];
        pcv[k] = -P.tab->xcos[L + 64 * k]; psv[k] = P.tab->xsin[L + 64 * k];
    }
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int nloops = 4 << k, nblocks = 16 >> k, m = lane & (nloops - 1);
        twc[k] = P.tab->cos[m * nblocks];
        tws[k] = -P.tab->sin[m * nblocks];
    }

    const int16_t *frame_pcm = P.pcm + ((size_t)s * P.frames + f) * 1536 * P.nch + P.chmap[ch];
    // the lane's four samples of: the block before (history), this block, and - in flight while this block is
    // transformed - the next one.  Every sample is read from HBM once.
    int16_t oldv[4], newv[4], nxtv[4];
    int joff[4];                                    // the lane's positions as sample offsets (32-bit index arithmetic)
#pragma unroll
    for (int k = 0; k < 4; k++) joff[k] = __mul24(jpos[k], P.nch);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = jpos[k];
        if (f > 0) oldv[k] = frame_pcm[joff[k] - 256 * P.nch];                   // block 5 of the previous frame
        else oldv[k] = P.slot ? P.last[((size_t)P.slot[s] * 6 + ch) * 256 + j] : P.last[((size_t)s * P.nch + ch) * 256 + j];
        newv[k] = frame_pcm[joff[k]];
    }
    for (int blk = 0; blk < 6; blk++) {
        // ---- 512 input samples: 256 old + 256 new (:1673-1683), windowed (:1686-1693) - in registers ----
        int win_[8];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int j = jpos[k];
            nxtv[k] = blk < 5 ? frame_pcm[(blk + 1) * 256 * P.nch + joff[k]] : (int16_t)0;
            if (P.store_history && blk == 5) {
                if (P.slot) P.last[((size_t)P.slot[s] * 6 + ch) * 256 + j] = newv[k];
                else P.last[((size_t)s * P.nch + ch) * 256 + j] = newv[k];
            }
            // (a 16-bit sample times a window value in 0 .. 32767, >> 15, is a 16-bit value again: no truncation to restate)
            win_[k] = __mul24(oldv[k], wa[k]) >> 15;
            win_[4 + k] = __mul24(newv[k], wb[k]) >> 15;
            oldv[k] = newv[k];
        }
        // ---- block floating point (:1697-1700): v from the position of the largest magnitude's top bit (the reference ORs the
        //      magnitudes, :1570-1596; the maximum has the same top bit) ----
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) acc = max(acc, max(win_[k], -win_[k]));
        acc = wave_max_nonneg(acc);
        int v = 14 - ilog2u((unsigned)acc);
        if (v < 0) v = 0;
        const int shift = v - 9;
        int O[4], N[4];                                                     // in[jpos[k]], in[256 + jpos[k]] as 16-bit values
        // (|x| << v < 2^15 by the choice of v, or v = 0: the shifted values are 16-bit values already)
#pragma unroll
        for (int k = 0; k < 4; k++) { O[k] = win_[k] << v; N[k] = win_[4 + k] << v; }
        // ---- rotation + pre-rotation (:578-591) of the points L (-> p) and L + 64 (-> q) ----
        int pr, pi, qr, qi;
        {
            const int re0 = ((int)(int16_t)(-N[2]) - N[1]) >> 1, im0 = (-(O[2] - O[1])) >> 1;
            const int re1 = (O[0] - O[3]) >> 1, im1 = (-(N[0] - (int)(int16_t)(-N[3]))) >> 1;
            // (every product of this kernel has operands of 17 bits at most: the 24-bit multiplier, not the quarter-rate 32-bit one)
            pr = (int16_t)((__mul24(re0, pcv[0]) - __mul24(im0, psv[0])) >> 15);
            pi = (int16_t)((__mul24(re0, psv[0]) + __mul24(pcv[0], im0)) >> 15);
            qr = (int16_t)((__mul24(re1, pcv[1]) - __mul24(im1, psv[1])) >> 15);
            qi = (int16_t)((__mul24(re1, psv[1]) + __mul24(pcv[1], im1)) >> 15);
        }
        // ---- the seven passes (:508-567) in registers: the lane's butterfly of pass k takes the points ip = (lane / d) 2d +
        //      lane mod d and ip + d (d = 2^k); between two passes every lane trades ONE point with lane ^ d (the upper half of
        //      a 2d-lane group gives its first point and keeps the second, the lower half the other way round).  Same operands,
        //      shifts and 16-bit truncations as the reference's in-place loops, no LDS round trip per pass. ----
        auto bfly_r = [&](int ax, int ay) {
            const int bx = pr, by = pi;
            pr = (int16_t)((bx + ax) >> 1);
            pi = (int16_t)((by + ay) >> 1);
            qr = (int16_t)((bx - ax) >> 1);
            qi = (int16_t)((by - ay) >> 1);
        };
        // lane ^ d without LDS: quad permutes (d = 1, 2), row shifts by bank (4), a row rotation (8); for d = 16 and 32 the trade
        // IS gfx950's v_permlane16_swap / v_permlane32_swap of (p, q): odd rows / the upper half of p against even rows / the
        // lower half of q
        auto trade = [&](auto dc) {
            constexpr int d = decltype(dc)::value;
            if constexpr (d == 32) {
                const auto r = __builtin_amdgcn_permlane32_swap((unsigned)pr, (unsigned)qr, false, false);
                const auto i = __builtin_amdgcn_permlane32_swap((unsigned)pi, (unsigned)qi, false, false);
                pr = (int)r[0]; qr = (int)r[1]; pi = (int)i[0]; qi = (int)i[1];
            } else if constexpr (d == 16) {
                const auto r = __builtin_amdgcn_permlane16_swap((unsigned)pr, (unsigned)qr, false, false);
                const auto i = __builtin_amdgcn_permlane16_swap((unsigned)pi, (unsigned)qi, false, false);
                pr = (int)r[0]; qr = (int)r[1]; pi = (int)i[0]; qi = (int)i[1];
            } else if constexpr (d == 4 || d == 8) {
                // the DPP bank mask does the selecting: the lower lanes' q takes the partner's p, the upper lanes' p the
                // partner's (old) q - four moves and two copies where selects around one move per component took eight to ten
                constexpr int lo_banks = d == 4 ? 0x5 : 0x3, up_banks = d == 4 ? 0xa : 0xc;
                constexpr int from_above = d == 4 ? 0x104 : 0x128, from_below = d == 4 ? 0x114 : 0x128;     // row_shl:4 / row_shr:4, row_ror:8
                const int qr0 = qr, qi0 = qi;
                qr = __builtin_amdgcn_update_dpp(qr, pr, from_above, 0xf, lo_banks, false);
                qi = __builtin_amdgcn_update_dpp(qi, pi, from_above, 0xf, lo_banks, false);
                pr = __builtin_amdgcn_update_dpp(pr, qr0, from_below, 0xf, up_banks, false);
                pi = __builtin_amdgcn_update_dpp(pi, qi0, from_below, 0xf, up_banks, false);
            } else {
                const bool up = (lane & d) != 0;
                const int sr = up ? pr : qr, si = up ? pi : qi;
                int rr, ri;
                if constexpr (d == 1) {
                    rr = __builtin_amdgcn_update_dpp(0, sr, 0xb1, 0xf, 0xf, false);         // quad_perm [1,0,3,2]
                    ri = __builtin_amdgcn_update_dpp(0, si, 0xb1, 0xf, 0xf, false);
                } else {
                    static_assert(d == 2, "");
                    rr = __builtin_amdgcn_update_dpp(0, sr, 0x4e, 0xf, 0xf, false);         // quad_perm [2,3,0,1]
                    ri = __builtin_amdgcn_update_dpp(0, si, 0x4e, 0xf, 0xf, false);
                }
                pr = up ? rr : pr; pi = up ? ri : pi;
                qr = up ? qr : rr; qi = up ? qi : ri;
            }
        };
        bfly_r(qr, qi);                                                     // pass 0
        trade(std::integral_constant<int, 1>{});
        if (lane & 1) bfly_r(qi, -qr); else bfly_r(qr, qi);                 // pass 1: twiddles 1 and -j
        trade(std::integral_constant<int, 2>{});
        auto pass = [&](int k) {                                            // passes 2..6
            const int nloops = 4 << k;
            const int c = twc[k], sx = tws[k];
            const int tr = (__mul24(c, qr) - __mul24(sx, qi)) >> 15;
            const int ti = (__mul24(c, qi) + __mul24(qr, sx)) >> 15;
            const bool plain = (lane & (nloops - 1)) == 0;
            bfly_r(plain ? qr : tr, plain ? qi : ti);
        };
        pass(0); trade(std::integral_constant<int, 4>{});
        pass(1); trade(std::integral_constant<int, 8>{});
        pass(2); trade(std::integral_constant<int, 16>{});
        pass(3); trade(std::integral_constant<int, 32>{});
        pass(4);
        // ---- post-rotation (:596-602): the lane now holds points lane and lane + 64 ----
        {
            const int sx0 = xsv[0], c0 = xcv[0], sx1 = xsv[1], c1 = xcv[1];
            out[2 * lane] = (__mul24(pr, c0) + __mul24(sx0, pi)) >> 15;
            out[255 - 2 * lane] = (__mul24(pr, sx0) - __mul24(pi, c0)) >> 15;
            out[2 * (lane + 64)] = (__mul24(qr, c1) + __mul24(sx1, qi)) >> 15;
            out[255 - 2 * (lane + 64)] = (__mul24(qr, sx1) - __mul24(qi, c1)) >> 15;
        }
        WAVE_SYNC();
        // ---- exponents (:1707-1722) ----
        const size_t row = (((size_t)s * P.frames + f) * 6 + blk) * P.nch + ch;
        int4 cv = *reinterpret_cast<const int4 *>(&out[4 * lane]);
        int cc[4] = {cv.x, cv.y, cv.z, cv.w};
        uint32_t epack = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int a = cc[k] < 0 ? -cc[k] : cc[k];
            int e;
            if (a == 0) e = 24;
            else {
                e = 23 - ilog2u((unsigned)a) + shift;
                if (e >= 24) { e = 24; cc[k] = 0; }
            }
            epack |= (uint32_t)(e & 0xff) << (8 * k);
        }
        // (the packers read a row's coded bins only: 223 of a full-bandwidth channel, 7 of the LFE - a quarter of the store traffic)
        if (P.full_rows || 4 * lane < ((P.x.lfe && ch == P.nch - 1) ? 7 : P.x.nbc))
            *reinterpret_cast<int4 *>(P.mdct + row * 256 + 4 * lane) = make_int4(cc[0], cc[1], cc[2], cc[3]);
        if (P.expo) *reinterpret_cast<uint32_t *>(P.expo + row * 256 + 4 * lane) = epack;       // tap only
        *reinterpret_cast<uint32_t *>(&XL.E[blk][4 * lane]) = epack;
        if (lane == 0) P.shift[row] = (int8_t)shift;
#pragma unroll
        for (int k = 0; k < 4; k++) newv[k] = nxtv[k];
        WAVE_SYNC();
    }
    exp_stage(P.x, XL, (size_t)sf, ch, lane);
}


// bap of one coefficient for SNR offset `snroffset` (:393-420):
//   v = ((max(mask - snroffset - floor, 0)) & 0x1fe0) + floor,  address = (psd - v) >> 5,  psd = 3072 - 128 exp
//   =>  address = clamp(80 - 4 exp - max(0, (mask - floor - snroffset) >> 5), 0, 63)       (floor = 0x1f0)
// The search kernel works on 4 x address (a byte offset into bitlut): clamp(term - 16 exp, 0, 252) with the band's
// term = 320 - max(0, ((mask - floor - snroffset) >> 3) & ~3) (SearchLDS::terms); the packers on the address itself (enc_mant.h).
// L.bitlut[address] = plain mantissa width | (bap==1) << 9 | (bap==2) << 14 | (bap==4) << 19: 24 bits, so that a lane's sums
// over a frame's rows (width <= 6 x 4 x 16 = 384, counts <= 24) stay clear of each other and a sum can be added to a block's
// account with one v_mad_u32_u24.
typedef short pk2 __attribute__((ext_vector_type(2)));

// plain (ungrouped) mantissa width of a bap code; 0 for the grouped codes 1, 2, 4 and for 0
__device__ __forceinline__ int plain_bits(int bp)
{
    return bp == 3 ? 3 : bp == 5 ? 4 : bp == 14 ? 14 : bp == 15 ? 16 : bp >= 6 ? bp - 1 : 0;
}

// The reference's SNR-offset search (:921-967) as a resumable state machine: next() skips the steps that
// need no evaluation and names the next (csnroffst, fsnroffst) to try, consume() takes the verdict.
struct SnrSearch {
    int csnr, fsnr, phase;
    bool failed;
    __device__ __forceinline__ bool next(int &cc, int &ff)
    {
        for (;;) {
            cc = csnr; ff = fsnr;
            if (phase == 0) { if (csnr < 0) { failed = true; phase = 5; return false; } return true; }
            if (phase == 1) { if (csnr + 4 > 63) { phase = 2; continue; } cc = csnr + 4; return true; }
            if (phase == 2) { if (csnr + 1 > 63) { phase = 3; continue; } cc = csnr + 1; return true; }
            if (phase == 3) { if (fsnr + 4 > 15) { phase = 4; continue; } ff = fsnr + 4; return true; }
            if (phase == 4) { if (fsnr + 1 > 15) { phase = 5; return false; } ff = fsnr + 1; return true; }
            return false;
        }
    }
    __device__ __forceinline__ void consume(bool ok)
    {
        if (phase == 0) { if (ok) phase = 1; else csnr -= 4; }
        else if (phase == 1) { if (ok) csnr += 4; else phase = 2; }
        else if (phase == 2) { if (ok) csnr += 1; else phase = 3; }
        else if (phase == 3) { if (ok) fsnr += 4; else phase = 4; }
        else if (phase == 4) { if (ok) fsnr += 1; else phase = 5; }
    }
};

// CRC-16 of `len` bytes ending at byte `end` (exclusive) of the MSB-first frame, per-lane chunks of C
// bytes aligned to the end of the region, combined in GF(2)[x]/poly.  Bytes < zero_below count as 0.
// a * t[0] in GF(2)[x]/poly for a table t[i] = t[0] * x^i mod poly (wave-uniform, from the kernel arguments)
__device__ __forceinline__ uint32_t gf_mul_tab(uint32_t a, const uint16_t *t)
{
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) c ^= (uint32_t)__builtin_amdgcn_sbfe((int)a, i, 1) & (uint32_t)t[i];
    return c;
}

template <class LDS>
__device__ uint32_t region_crc(const LDS &L, const uint32_t *fr, int end, int len, int C, const uint16_t (*pw)[16], int zero_below, int lane)
{
    const int start = end - 64 * C;                 // may be negative: leading zero padding
    int p = start + lane * C;
    uint32_t crc = 0;
    for (int i = 0; i < C; i++, p++) {
        uint32_t byte = 0;
        if (p >= end - len && p >= zero_below) byte = (fr[p >> 2] >> (24 - 8 * (p & 3))) & 0xff;
        crc = (L.crc_tab[byte ^ (crc >> 8)] ^ (crc << 8)) & 0xffff;
    }
#pragma unroll
    for (int k = 0; k < 6; k++) {
        const int d = 1 << k;
        const uint32_t left = __shfl_up(crc, d, 64);
        if (((lane + 1) & (2 * d - 1)) == 0) crc = gf_mul_tab(left, pw[k]) ^ crc;
    }
    return __shfl(crc, 63, 64);
}

// PART 0: one wavefront per stream does everything, frames in order.
// PART 3 + PART 1 + PART 2: for few, long streams.  Only the search carries something from frame to frame (it starts
// from the previous frame's csnroffst).  PART 3 (one wavefront per frame) tabulates the fit verdicts around every
// frame's own optimum; PART 1 (one wavefront per stream) replays the reference's search sequence frame after frame
// from those tables - costing an offset itself only when the table has no answer - and leaves csnroffst / fsnroffst
// of every frame in P.snr; PART 2 (one wavefront per frame) packs all frames at once.
// Measurement aid (make EXTRA=-DPACK_STAMPS, a separate library): lane 0 of every wavefront adds the s_memtime cycles it
// spent in each section of a frame to g_pack_cycles: search kernel: 0 frame set-up + SNR-offset search, 4 = calls of
// cost_and_record, 5 = frames; enc_packf_kernel: 6 set-up, 1 header / side information / exponent groups, 2 mantissas, 3 CRCs +
// store, 7 = frames.  ac3mi_debug_pack_cycles reads them.
#ifdef PACK_STAMPS
__device__ unsigned long long g_pack_cycles[16];
#define PK_DECL() unsigned long long pk_t = 0, pk_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}
#define PK_T0() pk_t = __builtin_readcyclecounter()
#define PK_LAP(id) do { const unsigned long long t_ = __builtin_readcyclecounter(); pk_acc[id] += t_ - pk_t; pk_t = t_; } while (0)
#define PK_COUNT(id) pk_acc[id] += 1
#define PK_END() do { if (lane == 0) for (int i_ = 0; i_ < 9; i_++) atomicAdd(&g_pack_cycles[i_], pk_acc[i_]); } while (0)
#else
#define PK_DECL() do { } while (0)
#define PK_T0() do { } while (0)
#define PK_LAP(id) do { } while (0)
#define PK_COUNT(id) do { } while (0)
#define PK_END() do { } while (0)
#endif

// offsets costed per sweep of the SNR-offset search (two per packed 16-bit pipeline).  4 saves a sweep now and then but costs
// registers and reductions: 6.01 - 6.04 against 5.92 ms per 65 536 one-frame streams, 1 % ahead on long streams; 3 stays
#ifndef ENC_NC
#define ENC_NC 3
#endif

#ifndef ENC_SEARCH_LB
#define ENC_SEARCH_LB 4          // 128 VGPRs; a fifth wavefront per SIMD (96 VGPRs, 28 bytes of scratch) measured the same 1.02 ms
#endif
// enc_search_kernel<1>: one wavefront per stream, frames in order; <3>: one wavefront per frame tabulates (see above).
template <int PART>
__global__ __launch_bounds__(64, ENC_SEARCH_LB) void enc_search_kernel(const PackParams P)
{
    static_assert(PART == 1 || PART == 3, "the packers are enc_packf_kernel / enc_packb_kernel");
    __shared__ SearchLDS L;
    const int lane = threadIdx.x;
    constexpr bool PER_FRAME = PART == 3;
    const int s = PER_FRAME ? (int)(blockIdx.x / (unsigned)P.frames_per_stream) : (int)blockIdx.x;
    const int f_first = PER_FRAME ? (int)(blockIdx.x - (unsigned)s * (unsigned)P.frames_per_stream) : 0;
    const int f_end = PER_FRAME ? f_first + 1 : P.frames_per_stream;
    if (s >= P.n_streams) return;

    for (int i = lane; i < 256; i += 64) L.band_of_bin[i] = P.tab->band_of_bin[i];
    {
        const int bp = P.tab->baptab[lane];
        L.bitlut[lane] = (uint32_t)plain_bits(bp) | ((bp == 1) << 9) | ((bp == 2) << 14) | ((bp == 4) << 19);
    }

    const int nch = P.nch, nfbw = P.nfbw, nbc = P.nbc;
    const int fs = P.frame_words;

    const int sslot = P.slot ? P.slot[s] : s;
    // the stream's search state: csnroffst in bits 0-7, the fsnroffst of its last coded frame in bits 8-11 (what the
    // reference's s->csnroffst / s->fsnroffst[] hold between frames, ENC/ac3enc.cpp:969-972)
    const int state_in = P.csnr_state[sslot];
    int csnr_prev = state_in & 0xff, fsnr_prev = (state_in >> 8) & 15;
    PK_DECL();

    for (int f = f_first; f < f_end; f++) {
        const size_t fidx = (size_t)s * P.frames_per_stream + f;
        PK_T0();
        PK_COUNT(5);
        const uint8_t *ex = P.eexp + fidx * 6 * nch * 256;
        int frame_bits = 0;
        uint64_t run_starts = 0, row_set = 0;
        bool loaded = false;
        auto load_frame = [&]() {
            loaded = true;
        // ---- masking curves, strategies and exponent bit counts from exp_stage (the encoded exponents
            //      stay in HBM/L2: [blk][ch][256] bytes at `ex`) ----
            {
                const uint32_t *gm = reinterpret_cast<const uint32_t *>(P.emask + fidx * 6 * nch * 50);
                uint32_t mv[15];
#pragma unroll
                for (int j = 0; j < 15; j++) {
                    const int i = lane + 64 * j;
                    mv[j] = i < 6 * nch * 25 ? gm[i] : 0;
                }
#pragma unroll
                for (int j = 0; j < 15; j++) {
                    const int i = lane + 64 * j;
                    if (i < 6 * nch * 25) reinterpret_cast<uint32_t *>(&L.mask[0][0])[i] = mv[j];
                }
            }
            if (lane < 6 * nch) { const int b = lane / nch, ch = lane - b * nch; L.strat[b][ch] = P.strat[fidx * 6 * nch + lane]; }
            frame_bits = wave_sum(lane < nch ? P.ebits[fidx * nch + lane] : 0);
            WAVE_SYNC();
            // rows (blk * 6 + ch) that send new exponents, i.e. start a run, as a bit set; and the same as bit 8*ch + b
            {
                const int b6 = lane / 6, c6 = lane - 6 * b6;
                row_set = __ballot(lane < 36 && c6 < nch && L.strat[lane < 36 ? b6 : 0][c6] != 0);
                const int c8 = lane >> 3, b8 = lane & 7;
                run_starts = __ballot(b8 < 6 && c8 < nch && L.strat[b8 < 6 ? b8 : 0][c8 < 6 ? c8 : 0] != 0);
                // one descriptor per run-start row, computed by the row's lane, compacted in row order: the sweeps then
                // walk a list instead of deriving block range, channel and addresses from bit sets (scalar work per row
                // and sweep: the scalar unit is what bounds the search)
                if (lane < 36 && ((row_set >> lane) & 1)) {
                    const uint32_t later = (uint32_t)((run_starts >> (8 * c6)) & 0x3f) | 0x40u;       // bit b: block b of this channel sends exponents
                    const int b1 = __builtin_ctz(later >> (b6 + 1)) + b6 + 1;
                    const uint32_t cover = ((1u << b1) - 1u) & ~((1u << b6) - 1u);
                    const uint32_t d = (uint32_t)((b6 * nch + c6) * 256) | ((P.lfe && c6 == nch - 1) ? 1u << 14 : 0u) | (cover << 16) | ((uint32_t)(b6 * nch + c6) << 24);
                    L.rowdesc[__builtin_popcountll(row_set & ((1ull << lane) - 1ull))] = d;
                }
            }
            // ---- fixed side information (:880-916) ----
            {
                const int extra[8] = {0, 0, 2, 2, 2, 4, 2, 4};
                frame_bits += 65 + extra[P.acmod & 7];
                frame_bits += 6 * (nfbw * 2 + 2 + (P.acmod == 2 ? 1 : 0) + 2 * nfbw + (P.lfe ? 1 : 0) + 1 + 1 + 2);
                // chbwcod (6 bits) and gainrng (2 bits) of every full-bandwidth channel-block that sends exponents
                uint64_t fbw_rows = 0;
                for (int b = 0; b < 6; b++) fbw_rows |= ((1ull << nfbw) - 1) << (6 * b);
                frame_bits += 8 * __builtin_popcountll(row_set & fbw_rows);
                frame_bits++;
                frame_bits += 2 * 4 + 3 + 6 + nch * (4 + 3);
                frame_bits += 2;
                frame_bits += 16;
            }
        };
        // PART 1 replays the search from tabulated verdicts and needs the frame's data only for a verdict that is missing
        if (PART != 1 || P.tap_strat) load_frame();

        // ---- SNR offset search, exactly the reference's sequence (:921-967).  Up to three candidates are
        //      evaluated per sweep over the coefficients, chosen by running the reference's loop ahead on
        //      the assumption that each one fits; the verdicts are then consumed in the reference's order
        //      and everything after the first surprise is discarded. ----
        const uint32_t bandoff = *reinterpret_cast<const uint32_t *>(&L.band_of_bin[4 * lane]);    // bands of bins 4*lane..+3
        SnrSearch ss{csnr_prev, 0, 0, false};
        // Verdicts already known for this frame: bit cc of known_c / fits_c for (cc, fsnroffst 0), bit ff of
        // known_f / fits_f for (csnroffst f_cc, ff > 0).  The reference asks for some offsets twice.
        uint64_t known_c = 0, fits_c = 0;
        uint32_t known_f = 0, fits_f = 0;
        int f_cc = -1;
        if (PART == 1 && P.memo) {                                  // tabulated by PART 3
            const uint32_t *m = P.memo + fidx * 8;
            auto word = [&](int i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)m[i]); };    // (the builtin returns int)
            known_c = (uint64_t)word(0) | ((uint64_t)word(1) << 32);
            fits_c = (uint64_t)word(2) | ((uint64_t)word(3) << 32);
            known_f = word(4);
            fits_f = word(5);
            f_cc = (int)word(6);
        }
        bool went_down = false, went_up = false;
        // Verdicts that follow from a costed offset WITHOUT costing.  A frame's mantissa bits are N + E: N = the sum over the
        // coefficients of a nominal width (0, 5/3, 7/3, 3, 7/2, 4, 5 ... 16 by bap), E = what the grouped codes' ceilings add - per
        // block 10/3 or 5/3 bits for one or two 3-level mantissas left over, 14/3 or 7/3 for 5-level ones, 7/2 for an odd 11-level
        // one: 0 <= E <= 69 bits a frame.  N cannot fall when the offset rises (a coefficient's bap table address, :393-420, never
        // falls with snroffset, and the widths rise with the address).  So with the E of a COSTED offset g1 in hand (it follows from
        // the counts the sweep reduces anyway): every g > g1 spends at least N(g1) = bits(g1) - E(g1): all fail if spare(g1) <
        // -E(g1); every g < g1 spends at most N(g1) + 69: all fit if spare(g1) >= 69 - E(g1) - exactly, whatever their own ceilings
        // do.  (Rounds 2-3 used +-72 for both; the band of undecided offsets is half as wide now: profiles/search_sim.py, 3.39 ->
        // 3.22 sweeps per fresh frame, 2.57 -> 2.33 with a transcode's hint.)  Sixths of a bit; offsets as g = 16 csnroffst + fsnroffst.
        int fit_hi = -1, fail_lo = 1 << 20;
        // Every offset costed for this frame, whether the reference asks for it or not: 1024-bit maps "costed" / "fits", word w in
        // lane w of one register each; and the two costed offsets that enclose the boundary most closely, with their spare bits.
        uint32_t costed_map = 0, fits_map = 0;
        int gl = -1, spare_l = 0, gh = -1, spare_h = 0;
        int probe_sweeps = 0;
        auto lookup = [&](int cc, int ff, bool &fits) {
            const int g = 16 * cc + ff;
            if (g <= fit_hi) { fits = true; return true; }
            if (g >= fail_lo) { fits = false; return true; }
            if (((uint32_t)__builtin_amdgcn_readlane((int)costed_map, g >> 5) >> (g & 31)) & 1u) {
                fits = ((uint32_t)__builtin_amdgcn_readlane((int)fits_map, g >> 5) >> (g & 31)) & 1u;
                return true;
            }
            if (ff == 0) { fits = (fits_c >> cc) & 1; return (bool)((known_c >> cc) & 1); }
            fits = (fits_f >> ff) & 1;
            return cc == f_cc && ((known_f >> ff) & 1);
        };
        // Mantissa bits of the whole frame at up to three offsets, and the verdicts into the memo.  Bit allocation
        // is a function of the (encoded) exponents alone, so a block that reuses a channel's exponents has that
        // channel's counts of the block that sent them: every run of blocks is counted once, 64 bins per step, and
        // added to the per-lane accumulators of all blocks of the run.
        auto cost_and_record = [&](const int (&cg)[ENC_NC], int n_cand, bool probing) {
            if (!loaded) load_frame();
            int so[ENC_NC];                 // snroffset (:393-420) of the candidates; unused slots repeat the first
#pragma unroll
            for (int k = 0; k < ENC_NC; k++) so[k] = ((k < n_cand ? cg[k] : cg[0]) - 240) << 2;
            PK_COUNT(4);
            const int budget = 16 * fs - frame_bits;
            uint32_t acc[6][ENC_NC];
#pragma unroll
            for (int B = 0; B < 6; B++)
#pragma unroll
                for (int c = 0; c < ENC_NC; c++) acc[B][c] = 0;
            {
                // one run-start row per step: four bins per lane from one dword (HBM/L2); the exponent dwords of the next
                // three rows are in flight while one is costed
                const int nrows = __builtin_popcountll(row_set);
                auto desc_of = [&](int i) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)L.rowdesc[i < nrows ? i : nrows - 1]); };
                auto fetch = [&](uint32_t d) { return *reinterpret_cast<const uint32_t *>(ex + (d & 0x3fffu) + 4 * lane); };
                uint32_t d0 = desc_of(0), d1 = desc_of(1), d2 = desc_of(2);
                uint32_t ev = fetch(d0), ev1 = fetch(d1), ev2 = fetch(d2);
                static_assert(ENC_NC == 3, "two packed candidates + one");
                // a row's band terms (lane = band), a row ahead of its coefficients
                auto terms = [&](uint32_t d, int buf) {
                    const int m = L.mask[d >> 24][lane < 50 ? lane : 49];
                    int D[3];
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        int q = ((m - so[c]) >> 3) & ~3;
                        q = q < 0 ? 0 : q;
                        D[c] = 320 - q;
                    }
                    if (lane < 50) L.terms[buf][lane] = make_uint2((uint32_t)(D[0] & 0xffff) | ((uint32_t)D[1] << 16), (uint32_t)(D[2] & 0xffff));
                };
                terms(d0, 0);
                // bins the row does not code: exponent 255 - term - 16 x 255 < 0, table address 0, bap 0, no bits
                auto beyond = [&](int n) { const int k = n - 4 * lane; return k >= 4 ? 0u : k <= 0 ? 0xffffffffu : 0xffffffffu << (8 * k); };
                const uint32_t um_fbw = beyond(nbc), um_lfe = beyond(7);
                uint32_t boff[4];                                           // byte offsets of the lane's four bands in a terms buffer
#pragma unroll
                for (int j = 0; j < 4; j++) boff[j] = ((bandoff >> (8 * j)) & 0xffu) * 8u;
#pragma unroll 1
                for (int i = 0; i < nrows; i++) {
                    const uint32_t d3 = desc_of(i + 3);
                    const uint32_t ev3 = fetch(d3);
                    if (i + 1 < nrows) terms(d1, (i + 1) & 1);
                    const char *Tr = reinterpret_cast<const char *>(&L.terms[i & 1][0]);
                    const uint32_t em = ev | ((d0 & (1u << 14)) ? um_lfe : um_fbw);
                    uint32_t sum[ENC_NC];
#pragma unroll
                    for (int c = 0; c < ENC_NC; c++) sum[c] = 0;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const uint32_t e16 = __umul24((em >> (8 * j)) & 0xffu, 0x00100010u);           // 16 x exponent, twice
                        const uint2 t = *reinterpret_cast<const uint2 *>(Tr + boff[j]);
                        pk2 a01 = __builtin_bit_cast(pk2, t.x) - __builtin_bit_cast(pk2, e16);
                        pk2 a2 = __builtin_bit_cast(pk2, t.y) - __builtin_bit_cast(pk2, e16);
                        a01 = __builtin_elementwise_min(__builtin_elementwise_max(a01, (pk2){0, 0}), (pk2){252, 252});
                        a2 = __builtin_elementwise_min(__builtin_elementwise_max(a2, (pk2){0, 0}), (pk2){252, 252});
                        const char *lut = reinterpret_cast<const char *>(&L.bitlut[0]);
                        sum[0] += *reinterpret_cast<const uint32_t *>(lut + (uint16_t)a01.x);
                        sum[1] += *reinterpret_cast<const uint32_t *>(lut + (uint16_t)a01.y);
                        sum[2] += *reinterpret_cast<const uint32_t *>(lut + (uint16_t)a2.x);
                    }
#pragma unroll
                    for (int B = 0; B < 6; B++) {
                        const uint32_t in_run = (d0 >> (16 + B)) & 1u;               // wave-uniform
#pragma unroll
                        for (int k = 0; k < ENC_NC; k++)    // (written out: the compiler turns a multiply by 0/1 back into select + add)
                            asm("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[B][k]) : "v"(sum[k]), "s"(in_run));
                    }
                    d0 = d1; d1 = d2; d2 = d3;
                    ev = ev1; ev1 = ev2; ev2 = ev3;
                }
            }
            // Frame totals.  Plain bits: summed over the blocks in the lane, one reduction per candidate.  The 3/5/11-level counts
            // of a (block, candidate) pair: three 10-bit fields reduced over either half of the wavefront (32 x 24 < 1024) and
            // dropped into lane 8 c + B of two collecting registers, so that the ceilings (:1194-1238: a code per 3, 3 or 2
            // mantissas of a block) are worked out once, across lanes, instead of pair by pair on the scalar unit.
            int total[ENC_NC], extra6[ENC_NC];
            {
                uint32_t col_lo = 0, col_hi = 0, bits_l[ENC_NC];
#pragma unroll
                for (int c = 0; c < ENC_NC; c++) bits_l[c] = 0;
#pragma unroll
                for (int B = 0; B < 6; B++) {
#pragma unroll
                    for (int c = 0; c < ENC_NC; c++) {
                        const uint32_t a = acc[B][c];
                        bits_l[c] += a & 0x1ffu;
                        uint32_t v = ((a >> 9) & 31u) | (((a >> 14) & 31u) << 10) | (((a >> 19) & 31u) << 20);
                        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);
                        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);
                        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);
                        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);
                        v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
                        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)v, 31), hi = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
                        const bool mine = lane == 8 * c + B;
                        col_lo = mine ? lo : col_lo;
                        col_hi = mine ? hi : col_hi;
                    }
                }
                const uint32_t n1 = (col_lo & 1023u) + (col_hi & 1023u), n2 = ((col_lo >> 10) & 1023u) + ((col_hi >> 10) & 1023u);
                const uint32_t n4 = (col_lo >> 20) + (col_hi >> 20);
                uint32_t t = 5u * (((n1 + 2u) * 0xaaabu) >> 17) + 7u * (((n2 + 2u) * 0xaaabu) >> 17) + 7u * ((n4 + 1u) >> 1);
                // ... and the ceilings' share of them in sixths of a bit (<= 69 per block): bits << 10 | sixths ride one reduction
                t = (t << 10) | (6u * t - (10u * n1 + 14u * n2 + 21u * n4));
                t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x111, 0xf, 0xf, true);      // lanes 8 c .. 8 c + 7 -> lane 8 c + 7
                t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x112, 0xf, 0xf, true);
                t += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)t, 0x114, 0xf, 0xf, true);
#pragma unroll
                for (int c = 0; c < ENC_NC; c++) {
                    const uint32_t tc = (uint32_t)__builtin_amdgcn_readlane((int)t, 8 * c + 7);
                    total[c] = (int)wave_sum_u32(bits_l[c]) + (int)(tc >> 10);
                    extra6[c] = (int)(tc & 1023u);
                }
            }
#pragma unroll
            for (int i = 0; i < ENC_NC; i++) {
                if (i >= n_cand) break;
                const int spare = budget - total[i];
                const bool ok = spare >= 0;
                const int g = cg[i], cand_ci = g >> 4, cand_fi = g & 15;
                if (6 * spare >= 414 - extra6[i] && g > fit_hi) fit_hi = g;
                if (6 * spare < -extra6[i] && g < fail_lo) fail_lo = g;
                {
                    const uint32_t bit = lane == (g >> 5) ? 1u << (g & 31) : 0u;
                    costed_map |= bit;
                    fits_map |= ok ? bit : 0u;
                }
                {                                   // (selects: an if / else here makes hipcc keep the four in scratch)
                    const bool up = ok && g > gl, dn = !ok && (gh < 0 || g < gh);
                    gl = up ? g : gl; spare_l = up ? spare : spare_l;
                    gh = dn ? g : gh; spare_h = dn ? spare : spare_h;
                }
                if (cand_fi == 0) { known_c |= 1ull << cand_ci; fits_c |= (uint64_t)ok << cand_ci; }
                else if (!probing || cand_ci == f_cc) {
                    if (cand_ci != f_cc) { f_cc = cand_ci; known_f = fits_f = 0; }     // costed ahead for another csnroffst
                    known_f |= 1u << cand_fi;
                    fits_f |= (uint32_t)ok << cand_fi;
                }
            }
        };
        bool first_sweep = true;
        const bool cold_hint = csnr_prev == 40 && fsnr_prev == 0;          // the state AC3_encode_init leaves (a guess that only steers which offsets are costed first)
        for (;;) {
            // advance the reference's loop as far as the known verdicts reach
            int cc, ff;
            bool more;
            while ((more = ss.next(cc, ff))) {
                bool fits;
                if (!lookup(cc, ff, fits)) break;
                if (ss.phase == 0 && !fits) went_down = true;
                if (ss.phase == 1 && fits) went_up = true;
                ss.consume(fits);
            }
            if (!more) break;
            // up to three offsets not costed yet, along the likeliest continuation: the start value fits unless
            // an earlier one did not, +4 steps fail unless one has fitted, the finer steps fit
            // (candidates as g = 16 csnroffst + fsnroffst in a list written through static indices only: a dynamic index sends
            // such arrays to scratch)
            int cg[ENC_NC], n_cand = 0;
#pragma unroll
            for (int k = 0; k < ENC_NC; k++) cg[k] = -1;
            auto add = [&](int g) {                 // false: already in the list
                bool dup = false;
#pragma unroll
                for (int k = 0; k < ENC_NC; k++) dup = dup || cg[k] == g;
                if (dup) return false;
#pragma unroll
                for (int k = 0; k < ENC_NC; k++) cg[k] = k == n_cand ? g : cg[k];
                n_cand++;
                return true;
            };
            // While the boundary is only known to lie between two costed offsets more than three steps apart, the next offsets
            // to cost are chosen where the line through their spare bits crosses zero (and one step - a twelfth of a wide
            // bracket - either side): the monotone bounds then answer every rung the reference asks about outside the new
            // bracket.  profiles/search_sim.py replays the policies on the oracle's spare-bit curves: 5.0 -> 3.7 sweeps per cold
            // frame, 3.0 -> 2.4 per warm one.
            bool probing = false;
            if (probe_sweeps < 3 && gl >= 0 && gh > gl + 3) {
                const int w = gh - gl;
                const float est = (float)w * ((float)spare_l / (float)(spare_l - spare_h));       // relative to gl
                int d = w > 48 ? (w * 85) >> 10 : 1;
                d = d < 2 && w > 48 ? 2 : d;
                const int ge = gl + (int)(est + 0.5f);
                auto clampg = [&](int g) { return g < gl + 1 ? gl + 1 : g > gh - 1 ? gh - 1 : g; };
                add(clampg(ge));
                add(clampg(ge + d));
                add(clampg(ge - d));
                // candidates that clamped onto each other (an estimate at the bracket's edge): the nearest uncosted neighbours
                // take their places - a sweep costs the same with one offset or three (2.35 -> 2.29 sweeps per transcoded frame,
                // 3.23 -> 3.16 per fresh one, and the tails shorter: profiles/search_sim.py)
                for (int k = 1; n_cand < ENC_NC && k < w; k++) {
                    if (ge - k > gl && ge - k < gh) add(ge - k);
                    if (n_cand < ENC_NC && ge + k > gl && ge + k < gh) add(ge + k);
                }
                probe_sweeps++;
                probing = true;
            }
            // Phase 0 of a search that has to come down a long way (a fresh stream starts at 40): the ladder csnroffst,
            // csnroffst - 4, ... is probed at three points that cut its unknown stretch into quarters instead of walked three
            // rungs per sweep - the bounds above turn a rung that fails or fits by a margin into the verdict of every rung
            // beyond it.  (Which offsets are COSTED never changes a result: the reference's sequence is replayed from exact
            // verdicts only.)
            const int hint0 = (first_sweep && cold_hint && P.hint) ? (int)__builtin_amdgcn_readfirstlane((int)P.hint[fidx * (size_t)P.hint_stride]) : 0;
            if (n_cand == 0 && hint0 > 0) {                      // (0: the source frame's first block failed to parse - no hint)
                // a transcode: decoded audio re-encoded at its source's rate lands within -5 .. +11 steps of the offsets the source
                // frame carried: the first sweep costs around them - 3.4 -> 2.45 sweeps per frame (profiles/search_sim.py on
                // second-generation content).  A hint far off (another target rate) costs a sweep; results never depend on it.
                const int g0 = hint0 < 8 ? 8 : hint0 > 1000 ? 1000 : hint0;
                add(g0 - 8); add(g0 + 2); add(g0 + 12);
            }
            if (n_cand == 0 && first_sweep && cold_hint) {
                // a fresh stream's first sweep: csnroffst 8, 13 and 20, around where material lands at the usual rates (64 kbps
                // mono to 640 kbps 5.1: boundary at 8 .. 16 1/2 in profiles/search_sim.py) - 3.8 -> 3.5 sweeps per cold frame
                // against quartering 0 .. 40, no worse at the extremes
                add(16 * 8); add(16 * 13); add(16 * 20);
            }
            if (n_cand == 0 && ss.phase == 0 && (went_down || (first_sweep && cold_hint))) {
                int n = 0;
                for (int c = ss.csnr; c >= 0 && n < 16; c -= 4, n++) { bool f; if (lookup(c, 0, f)) break; }
                if (n > 3) {
                    add(16 * (ss.csnr - 4 * ((n - 1) / 4)));
                    add(16 * (ss.csnr - 4 * ((n - 1) / 2)));
                    add(16 * (ss.csnr - 4 * ((3 * (n - 1) + 2) / 4)));
                }
            }
            first_sweep = false;
            if (n_cand == 0) {
                SnrSearch ahead = ss;
                while (n_cand < ENC_NC && ahead.next(cc, ff)) {
                    bool fits;
                    if (!lookup(cc, ff, fits)) {
                        if (!add(16 * cc + ff)) break;
                        fits = ahead.phase == 0 ? !went_down : ahead.phase == 1 ? went_up : true;
                    }
                    ahead.consume(fits);
                }
            }
            cost_and_record(cg, n_cand, probing);
        }
        if (PART == 3) {
            // Tabulation for the replay (PART 1), whose start value is the previous frame's result, not this pass's:
            // every csnroffst within reach of this frame's own optimum, so that the replay finds its questions
            // answered unless the fit is not monotone around here or the level jumps between frames.
            const int C = ss.failed ? 0 : ss.csnr;
            for (;;) {
                int cg[ENC_NC], n_cand = 0;
#pragma unroll
                for (int k = 0; k < ENC_NC; k++) cg[k] = -1;
                for (int c = C - 7 < 0 ? 0 : C - 7; c <= (C + 8 > 63 ? 63 : C + 8) && n_cand < ENC_NC; c++) {
                    if ((known_c >> c) & 1) continue;
#pragma unroll
                    for (int k = 0; k < ENC_NC; k++) cg[k] = k == n_cand ? 16 * c : cg[k];
                    n_cand++;
                }
                if (n_cand == 0) break;
                cost_and_record(cg, n_cand, false);
            }
            if (lane == 0) {
                uint32_t *m = P.memo + fidx * 8;
                m[0] = (uint32_t)known_c; m[1] = (uint32_t)(known_c >> 32);
                m[2] = (uint32_t)fits_c; m[3] = (uint32_t)(fits_c >> 32);
                m[4] = known_f; m[5] = fits_f; m[6] = (uint32_t)f_cc; m[7] = 0;
            }
            continue;
        }
        int csnr = ss.csnr, fsnr = ss.fsnr;
        int failed_alloc = -1;
        {
            if (!ss.failed) { csnr_prev = csnr; fsnr_prev = fsnr; }
            else {
                // The reference's error path ("Yack, Error !!!", :930-933), reached when the start value and every start value
                // minus a multiple of 4 down to 0..3 fail - even if an offset in between would fit: compute_bit_allocation
                // returns without touching s->csnroffst / s->fsnroffst, its caller ignores the result (:1752), the header
                // carries the previous frame's offsets and the mantissas follow the allocation of the LAST attempt,
                // (start & 3, 0).  Such a frame overflows; the reference writes on, the engine drops what is beyond its buffer.
                failed_alloc = csnr_prev & 3;
                csnr = csnr_prev;
                fsnr = fsnr_prev;
            }
        }
        if (P.tap_snr && lane == 0) { P.tap_snr[fidx * 2] = csnr; P.tap_snr[fidx * 2 + 1] = fsnr; }
        if (P.tap_strat && lane < 36) {
            const int b = lane / 6, ch = lane - 6 * b;
            if (ch < nch) P.tap_strat[(fidx * 6 + b) * nch + ch] = L.strat[b][ch];
        }
        // the packer of this frame is another wavefront (enc_packf_kernel) or workgroup (enc_packb_kernel)
        if (lane == 0) { P.snr[fidx * 2] = csnr; P.snr[fidx * 2 + 1] = fsnr | (failed_alloc >= 0 ? 0x100 | (failed_alloc << 9) : 0); }
        PK_LAP(0);
    }
    if (PART == 1 && lane == 0) P.csnr_state[sslot] = csnr_prev | (fsnr_prev << 8);
    PK_END();
}


// ---------------------------------------------------------------------------------------------
// enc_packf_kernel: a frame whose SNR offsets are known (P.snr, from enc_search_kernel<1>) is packed by ONE wavefront -
// the packer of large batches.  Header, side information and exponent groups as wave-uniform fields through a 64-bit
// scalar accumulator, the mantissas block by block on enc_mant.h, both CRCs, the frame out.
struct alignas(16) PackfLDS {
    int16_t mask[6][50];                // this block's masking curves as band terms (mant_band_terms), one row per channel
    uint32_t packlut[64];               // mant_pack_word per bap table address
    alignas(16) uint16_t glist[GL_ENTRIES];
    alignas(4) uint8_t erow[256];       // encoded exponents of the channel whose exponent groups are being packed
    alignas(4) uint16_t crc_tab[256];
    // (the frame itself, MSB-first dwords + 256 bytes of headroom for the overshoot quirk, is dynamic LDS: PackParams::frw)
};

#ifndef ENC_PACK2_LB
#define ENC_PACK2_LB 4           // 115 VGPRs, no scratch: 1.78 ms per 65 536 frames against 1.84 at 5 per SIMD (96 VGPRs, 72 bytes of scratch)
#endif
// FIXED51: the 5.1 configuration (five full-bandwidth channels + LFE, acmod 7) as compile-time constants - the shape large batches
// have; its five mantissa passes, the merged LFE lanes and the side information's field list then need no tests
template <bool FIXED51>
__global__ __launch_bounds__(64, ENC_PACK2_LB) void enc_packf_kernel(const PackParams P)
{
    __shared__ PackfLDS L;
    extern __shared__ uint4 pk_dyn[];
    uint32_t *fr = reinterpret_cast<uint32_t *>(pk_dyn);
    const int lane = threadIdx.x;
    PK_DECL();

    // (byte loop kept: copying the table as dwords measured 4 % SLOWER in round 3 - code placement, not work)
    for (int i = lane; i < 256; i += 64) L.crc_tab[i] = P.tab->crc_tab[i];
    {
        const int bp = P.tab->baptab[lane];
        L.packlut[lane] = mant_pack_word(bp, plain_bits(bp));
    }
    mant_lists_init(L.glist, lane);
    const uint32_t bandoff = *reinterpret_cast<const uint32_t *>(&P.tab->band_of_bin[4 * lane]);     // bands of bins 4*lane..+3
    // fixed allocation codes (:861-879)
    constexpr int sdecaycod = 2, fdecaycod = 1, sgaincod = 1, dbkneecod = 2, floorcod = 4, fgaincod = 4;
    const int nch = FIXED51 ? 6 : P.nch, nfbw = FIXED51 ? 5 : P.nfbw, nbc = FIXED51 ? 223 : P.nbc;
    const int acmod = FIXED51 ? 7 : P.acmod;
    const bool lfe = FIXED51 ? true : P.lfe != 0;
    const int fs = P.frame_words;
    // (One wavefront per frame, on purpose.  A persistent grid - 16 wavefronts per CU walking the frames, tables set up once per
    // wavefront - was measured in round 4: 2.06 ms per 65 536 frames against 1.76; with the loop but one frame per wavefront
    // 1.84.  Wavefronts that start together stay in step - all in their scalar side information, then all in their mantissa
    // passes - and the units they share are used in turns instead of side by side; the dispatcher's staggered starts mix the phases.)
    const size_t fidx = blockIdx.x;
    PK_T0();
    PK_COUNT(7);
    for (int i = lane; i < P.frw / 4; i += 64) reinterpret_cast<uint4 *>(fr)[i] = make_uint4(0, 0, 0, 0);
    const int32_t *md = P.mdct + fidx * 6 * nch * 256;
    const uint8_t *ex = P.eexp + fidx * 6 * nch * 256;
    // lane 6 b + ch: exponent strategy and exp_samples of channel ch in block b
    int strat_l = 0, shift_l = 0;
    if (lane < 36) {
        const int b = lane / 6, ch = lane - 6 * b;
        if (ch < nch) { strat_l = (int)P.strat[(fidx * 6 + b) * nch + ch]; shift_l = (int)P.shift[(fidx * 6 + b) * nch + ch]; }
    }
    int csnr, fsnr, snroffset;
    {
        const int w1 = __builtin_amdgcn_readfirstlane(P.snr[fidx * 2 + 1]);
        csnr = __builtin_amdgcn_readfirstlane(P.snr[fidx * 2]);
        fsnr = w1 & 15;
        snroffset = (((csnr - 15) << 4) + fsnr) << 2;
        if (w1 & 0x100) snroffset = (((w1 >> 9) - 15) << 4) << 2;       // a frame whose search failed: the last attempt's allocation under the stale header offsets
    }
    WAVE_SYNC();
    PK_LAP(6);

    // ---- header (:1113-1147) ----
    // Side information is wave-uniform: its fields collect in a 64-bit accumulator that stays in scalar registers and
    // reach the LDS frame up to 64 bits at a time (one lane, two put_bits): `flush` empties it before the lanes write at
    // `pos` themselves (exponent groups, mantissas) and wherever the next stretch of fields could overflow it - the block's
    // first stretch is 54 bits at most, the one after the exponents 63 (block 0 of six channels).
    uint32_t pos = 0;                   // first bit not yet in the frame
    uint64_t acc = 0;
    int nacc = 0;                       // pending bits, the low `nacc` of acc
    auto put = [&](int n, uint32_t v) { acc = (acc << n) | v; nacc += n; };       // (no test per field: at most 64 bits between two flushes)
    auto flush = [&]() {
        if (nacc > 32) {
            if (lane == 0) put_bits(fr, P.frw, pos, nacc - 32, (uint32_t)(acc >> 32) & (0xffffffffu >> (64 - nacc)));
            pos += nacc - 32;
            nacc = 32;
        }
        if (nacc > 0) {
            if (lane == 0) put_bits(fr, P.frw, pos, nacc, (uint32_t)acc & (0xffffffffu >> (32 - nacc)));
            pos += nacc;
            nacc = 0;
        }
    };
    put(16, 0x0b77); put(16, 0); put(2, P.fscod); put(6, P.frmsizecod); put(5, P.bsid); put(3, 0); put(3, acmod);
    flush();
    if ((acmod & 1) && acmod != 1) put(2, 1);
    if (acmod & 4) put(2, 1);
    if (acmod == 2) put(2, 0);
    put(1, lfe); put(5, 31); put(3, 0); put(1, 0); put(1, 1); put(3, 0);

    // ---- audio blocks (:1194-1502) ----
#pragma unroll 1
    for (int b = 0; b < 6; b++) {
        // this block's encoded exponents, one dword (four bins) per lane and channel, all rows in flight together: the
        // mantissa passes use them from registers, the exponent groups of a channel through L.erow
        uint32_t ew[6];
#pragma unroll
        for (int ch = 0; ch < 6; ch++) ew[ch] = ch < nch ? *reinterpret_cast<const uint32_t *>(ex + ((size_t)b * nch + ch) * 256 + 4 * lane) : 0u;
        {                                           // this block's mask rows (25 dwords per channel), in flight during the side information
            const uint32_t *gm = reinterpret_cast<const uint32_t *>(P.emask + (fidx * 6 + b) * nch * 50);
            uint32_t mv[3];
#pragma unroll
            for (int j = 0; j < 3; j++) mv[j] = lane + 64 * j < nch * 25 ? gm[lane + 64 * j] : 0u;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (lane + 64 * j < nch * 25) reinterpret_cast<uint32_t *>(&L.mask[0][0])[lane + 64 * j] = mant_band_terms(mv[j], snroffset);
        }
        auto strat_of = [&](int ch) { return (uint32_t)__builtin_amdgcn_readlane(strat_l, 6 * b + ch); };
        flush();
        for (int ch = 0; ch < nfbw; ch++) put(1, 0);
        for (int ch = 0; ch < nfbw; ch++) put(1, 1);
        put(1, 0);
        if (b == 0) { put(1, 1); put(1, 0); } else put(1, 0);
        if (acmod == 2) { if (b == 0) { put(1, 1); put(4, 0); } else put(1, 0); }
        for (int ch = 0; ch < nfbw; ch++) put(2, strat_of(ch));
        if (lfe) put(1, strat_of(nch - 1));
        for (int ch = 0; ch < nfbw; ch++) if (strat_of(ch) != 0) put(6, P.chbwcod);
        // exponents: lanes over groups
        for (int ch = 0; ch < nch; ch++) {
            const int stg = (int)strat_of(ch);
            if (stg == 0) continue;
            const bool is_lfe = lfe && ch == nch - 1;
            const int gs = stg == 1 ? 1 : stg == 2 ? 2 : 4;
            const int ng = ((is_lfe ? 7 : nbc) + gs * 3 - 4) / (3 * gs);
            const uint8_t *e = &L.erow[0];
            {
                uint32_t row = 0;
#pragma unroll
                for (int c2 = 0; c2 < 6; c2++) row = c2 == ch ? ew[c2] : row;      // (no indexing of the register array)
                WAVE_SYNC();                                            // the previous channel's groups have read the row
                *reinterpret_cast<uint32_t *>(&L.erow[4 * lane]) = row;
                WAVE_SYNC();
            }
            put(4, (uint32_t)__builtin_amdgcn_readfirstlane((int)e[0]));
            flush();
            for (int g = lane; g < ng; g += 64) {
                const int k0 = 1 + 3 * g * gs;
                const int prev = g ? e[k0 - gs] : e[0];
                const int d0 = e[k0] - prev + 2, d1 = e[k0 + gs] - e[k0] + 2, d2 = e[k0 + 2 * gs] - e[k0 + gs] + 2;
                put_bits(fr, P.frw, pos + 7 * g, 7, (uint32_t)((d0 * 5 + d1) * 5 + d2));
            }
            pos += 7 * ng;
            if (!is_lfe) put(2, 0);
        }
        if (b == 0) flush();
        put(1, b == 0);
        if (b == 0) { put(2, sdecaycod); put(2, fdecaycod); put(2, sgaincod); put(2, dbkneecod); put(3, floorcod); }
        put(1, b == 0);
        if (b == 0) {
            put(6, csnr);
            for (int ch = 0; ch < nch; ch++) { put(4, fsnr); put(3, fgaincod); }
        }
        put(1, 0);
        put(1, 0);
        flush();
        PK_LAP(1);

        // ---- mantissas (:1341-1501): enc_mant.h ----
        {
            int shv[6];
#pragma unroll
            for (int ch = 0; ch < 6; ch++) shv[ch] = __builtin_amdgcn_readlane(shift_l, 6 * b + ch);
            uint32_t ad[6];
            int neg;
            uint32_t em[6];
            mant_block_addresses(ad, em, neg, ew, shv, L.mask, bandoff, nch, nbc, lfe, lane);
            PK_LAP(8);
            MantBlock B;
            B.fr = fr; B.frw = P.frw; B.glist = L.glist; B.packlut = L.packlut;
            B.mdb = md + (size_t)b * nch * 256;
            B.tap_bap = P.tap_bap ? P.tap_bap + (fidx * 6 + b) * nch * 256 : nullptr;
            B.nch = nch; B.nbc = nbc; B.lfe = lfe; B.marker = (uint32_t)P.marker;
            pos = mant_pack_block(B, em, ad, shv, __ballot(neg < 0) != 0, pos, lane);
        }
        PK_LAP(2);
    }

    // ---- frame end (:1599-1638): bytes past 2*fs are dropped, CRCs stored over whatever is there ----
    const int fs58 = (fs >> 1) + (fs >> 3);
    // (the reference only zero-pads when the data is short; data bits beyond byte 2*fs-2 stay and are then overwritten by
    // crc2 - its own overshoot quirk)
    uint32_t crc1 = region_crc(L, fr, 2 * fs58, 2 * fs58, P.c1, P.pw1_t, 4, lane);
    crc1 = gf_mul_tab(crc1, P.crc_inv_t);
    const uint32_t crc2 = region_crc(L, fr, 2 * fs - 2, (fs - fs58) * 2 - 2, P.c2, P.pw2_t, 0, lane);
    WAVE_SYNC();
    if (lane == 0) {
        fr[0] = (fr[0] & 0xffff0000u) | (crc1 & 0xffff);                      // bytes 2,3
        const int p = 2 * fs - 2;                                                 // even -> inside one dword
        const int shft = 16 - 8 * (p & 3);
        fr[p >> 2] = (fr[p >> 2] & ~(0xffffu << shft)) | ((crc2 & 0xffff) << shft);
    }
    WAVE_SYNC();
    uint8_t *dst = P.frames + fidx * P.frame_stride;
    for (int i = lane; i < (2 * fs + 3) / 4; i += 64) {
        uint32_t v = __builtin_bswap32(fr[i]);
        const int rem = 2 * fs - 4 * i;
        if (rem >= 4) *reinterpret_cast<uint32_t *>(dst + 4 * i) = v;
        else for (int k = 0; k < rem; k++) dst[4 * i + k] = (uint8_t)(v >> (8 * k));
    }
    PK_LAP(3);
    PK_END();
}

// ---------------------------------------------------------------------------------------------
// enc_packb_kernel: a frame whose SNR offsets are known (P.snr, from enc_search_kernel<1>) is packed by a workgroup of six
// wavefronts, one per AUDIO BLOCK.  Nothing but its first bit ties a block to the blocks before it, and that follows from
// bit COUNTS: every wavefront first counts its block (side information from the strategies, mantissas from one table look-up
// per coefficient), the counts meet in LDS, then all six pack at once into the frame they share (fields reach it by LDS
// atomics, so neighbours may write into one dword); both CRCs run side by side on two wavefronts; 384 lanes store the frame.
// The reference's merged-code marker quirk (a grouped code whose value equals 128 is not written, :1466-1480) shortens a
// block, i.e. moves every later block: any wavefront that meets one reports it, and the workgroup packs the frame once more
// with those fields at width 0 - as enc_pack_kernel does per block.

struct PackbWave {                  // per wavefront = audio block
    int16_t mask[6][50];            // masking curves of the block's channels as band terms (mant_band_terms)
    alignas(16) uint16_t glist[GL_ENTRIES];     // members of the block's grouped codes (enc_mant.h)
    alignas(4) uint8_t erow[256];
};
struct alignas(16) PackbLDS {
    PackbWave w[6];
    uint32_t packlut[64];
    uint16_t crc_tab[256];
    uint8_t band_of_bin[256];
    uint32_t blk_bits[6];           // bits of each block: nominal (from the bap codes) ...
    uint32_t blk_bits2[6];          // ... and as packed (a grouped code equal to the merged-marker takes none)
    uint32_t crc[2];
    int any_coll;
};

#ifndef ENC_PACKB_LB
#define ENC_PACKB_LB 4        // 128 VGPRs, 12 bytes of scratch (6: 80 VGPRs, 204 bytes).  Its batches fit the chip several times over, so a frame's latency
                              // counts, not occupancy: cold encode 0.107 / 0.113 / 0.189 / 0.286 ms per 64 / 256 / 1 024 / 2 048 frames against 0.110 /
                              // 0.121 / 0.186 / 0.291 at 6
#endif
__global__ __launch_bounds__(384, ENC_PACKB_LB) void enc_packb_kernel(const PackParams P)
{
    __shared__ PackbLDS L;
    extern __shared__ uint4 pk_dyn[];
    uint32_t *fr = reinterpret_cast<uint32_t *>(pk_dyn);
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = __builtin_amdgcn_readfirstlane(tid >> 6);
    PackbWave &W = L.w[b];
    // consecutive frames on one XCD (cdna_hip_programming.md T1): the six wavefronts' rows sit side by side in memory
    size_t fidx;
    {
        const unsigned n = gridDim.x, q = n >> 3, r = n & 7u, x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        fidx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const int nch = P.nch, nfbw = P.nfbw, nbc = P.nbc, fs = P.frame_words;
    constexpr int sdecaycod = 2, fdecaycod = 1, sgaincod = 1, dbkneecod = 2, floorcod = 4, fgaincod = 4;

    for (int i = tid; i < 256; i += 384) {
        L.band_of_bin[i] = P.tab->band_of_bin[i];
        L.crc_tab[i] = P.tab->crc_tab[i];
    }
    if (tid < 64) { const int bp = P.tab->baptab[tid]; L.packlut[tid] = mant_pack_word(bp, plain_bits(bp)); }
    for (int i = tid; i < P.frw / 4; i += 384) reinterpret_cast<uint4 *>(fr)[i] = make_uint4(0, 0, 0, 0);
    const size_t rowb = (fidx * 6 + b) * nch;
    uint32_t mrows[3] = {0u, 0u, 0u};                   // the block's masking curves, two bands per dword (to LDS as band terms once the offsets are known)
    {
        const uint32_t *gm = reinterpret_cast<const uint32_t *>(P.emask + rowb * 50);
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (lane + 64 * j < nch * 25) mrows[j] = gm[lane + 64 * j];
    }
    // lane = channel: exponent strategy and exp_samples of the block's channels
    const int strat_l = lane < nch ? (int)P.strat[rowb + lane] : 0;
    const int shift_l = lane < nch ? (int)P.shift[rowb + lane] : 0;
    uint32_t ew[6];                                     // the block's encoded exponents, four bins per lane and channel
#pragma unroll
    for (int ch = 0; ch < 6; ch++) ew[ch] = ch < nch ? *reinterpret_cast<const uint32_t *>(P.eexp + (rowb + ch) * 256 + 4 * lane) : 0u;
    int csnr, fsnr, snroffset;
    {
        const int w1 = P.snr[fidx * 2 + 1];
        csnr = P.snr[fidx * 2];
        fsnr = w1 & 15;
        snroffset = (((csnr - 15) << 4) + fsnr) << 2;
        if (w1 & 0x100) snroffset = (((w1 >> 9) - 15) << 4) << 2;      // a frame whose search failed: see enc_pack_kernel
    }
    if (tid == 0) L.any_coll = 0;
    mant_lists_init(W.glist, lane);
#pragma unroll
    for (int j = 0; j < 3; j++)
        if (lane + 64 * j < nch * 25) reinterpret_cast<uint32_t *>(&W.mask[0][0])[lane + 64 * j] = mant_band_terms(mrows[j], snroffset);
    const uint32_t bandoff = *reinterpret_cast<const uint32_t *>(&P.tab->band_of_bin[4 * lane]);
    __syncthreads();

    const uint32_t strat_set = (uint32_t)__ballot(strat_l != 0);        // bit ch: the channel sends exponents in this block
    auto strat_of = [&](int ch) { return __builtin_amdgcn_readlane(strat_l, ch); };
    // bits of the frame header (:1113-1147) and of this block's side information with its exponents (:1194-1332)
    const int hdr_bits = 16 + 16 + 2 + 6 + 5 + 3 + 3 + (((P.acmod & 1) && P.acmod != 1) ? 2 : 0) + ((P.acmod & 4) ? 2 : 0) +
                         (P.acmod == 2 ? 2 : 0) + 1 + 5 + 3 + 1 + 1 + 3;
    int side_bits = 2 * nfbw + 1 + (b == 0 ? 2 : 1) + (P.acmod == 2 ? (b == 0 ? 5 : 1) : 0) + 2 * nfbw + (P.lfe ? 1 : 0) +
                    1 + (b == 0 ? 11 : 0) + 1 + (b == 0 ? 6 + 7 * nch : 0) + 2;
    for (int ch = 0; ch < nch; ch++) {
        const int stg = strat_of(ch);
        if (stg == 0) continue;
        const bool is_lfe = P.lfe && ch == nch - 1;
        const int gs = stg == 1 ? 1 : stg == 2 ? 2 : 4;
        side_bits += 4 + 7 * (((is_lfe ? 7 : nbc) + gs * 3 - 4) / (3 * gs)) + (is_lfe ? 0 : 8);      // (fbw: chbwcod 6 + gainrng 2)
    }

    // the addresses of the block's coefficients into the bap table (:393-420), four per lane and channel
    uint32_t ad[6], em[6];
    int shv[6], neg;
#pragma unroll
    for (int ch = 0; ch < 6; ch++) shv[ch] = __builtin_amdgcn_readlane(shift_l, ch);
    mant_block_addresses(ad, em, neg, ew, shv, W.mask, bandoff, nch, nbc, P.lfe != 0, lane);
    const bool garbage = __ballot(neg < 0) != 0;

#pragma unroll 1
    for (int attempt = 0; attempt < 2; attempt++) {
        // ---- this block's bit count -> where it starts (second attempt: the counts as packed) ----
        int mant_bits = 0;
        if (attempt == 0) {
            uint32_t cnt = 0, plain = 0;            // kinds as 10-bit fields (<= 24 per lane), plain bits
#pragma unroll
            for (int ch = 0; ch < 6; ch++) {
                if (ch >= nch) continue;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t w = L.packlut[(ad[ch] >> (8 * j)) & 63u];
                    cnt += 1u << (w >> 24);
                    plain += w & 31u;
                }
            }
            const uint32_t s1 = wave_sum_u32((cnt & 1023u) | (((cnt >> 10) & 1023u) << 16));
            const uint32_t s2 = wave_sum_u32(((cnt >> 20) & 1023u) | (plain << 16));
            const int n1 = s1 & 0xffff, n2 = s1 >> 16, n4 = s2 & 0xffff;
            mant_bits = (int)(s2 >> 16) + 5 * ((n1 + 2) / 3) + 7 * ((n2 + 2) / 3) + 7 * ((n4 + 1) / 2);
            if (lane == 0) L.blk_bits[b] = (uint32_t)(side_bits + mant_bits);
        }
        __syncthreads();
        uint32_t pos = (uint32_t)hdr_bits;
        for (int q = 0; q < b; q++) pos += attempt == 0 ? L.blk_bits[q] : L.blk_bits2[q];

        // ---- side information (wave-uniform fields through a 64-bit accumulator, see enc_pack_kernel) ----
        uint64_t acc = 0;
        int nacc = 0;
        auto put = [&](int n, uint32_t v) { acc = (acc << n) | v; nacc += n; };       // (no test per field: at most 64 bits between two flushes)
        auto flush = [&]() {
            if (nacc > 32) {
                if (lane == 0) put_bits(fr, P.frw, pos, nacc - 32, (uint32_t)(acc >> 32) & (0xffffffffu >> (64 - nacc)));
                pos += nacc - 32;
                nacc = 32;
            }
            if (nacc > 0) {
                if (lane == 0) put_bits(fr, P.frw, pos, nacc, (uint32_t)acc & (0xffffffffu >> (32 - nacc)));
                pos += nacc;
                nacc = 0;
            }
        };
        if (b == 0) {                               // the frame header is block 0's to write
            const uint32_t mine = pos;
            pos = 0;
            put(16, 0x0b77); put(16, 0); put(2, P.fscod); put(6, P.frmsizecod); put(5, P.bsid); put(3, 0); put(3, P.acmod);
            flush();
            if ((P.acmod & 1) && P.acmod != 1) put(2, 1);
            if (P.acmod & 4) put(2, 1);
            if (P.acmod == 2) put(2, 0);
            put(1, P.lfe); put(5, 31); put(3, 0); put(1, 0); put(1, 1); put(3, 0);
            flush();
            pos = mine;
        }
        for (int ch = 0; ch < nfbw; ch++) put(1, 0);
        for (int ch = 0; ch < nfbw; ch++) put(1, 1);
        put(1, 0);
        if (b == 0) { put(1, 1); put(1, 0); } else put(1, 0);
        if (P.acmod == 2) { if (b == 0) { put(1, 1); put(4, 0); } else put(1, 0); }
        for (int ch = 0; ch < nfbw; ch++) put(2, (uint32_t)strat_of(ch));
        if (P.lfe) put(1, (uint32_t)strat_of(nch - 1));
        for (int ch = 0; ch < nfbw; ch++) if ((strat_set >> ch) & 1u) put(6, P.chbwcod);
        // exponents: lanes over groups
        for (int ch = 0; ch < nch; ch++) {
            const int stg = strat_of(ch);
            if (stg == 0) continue;
            const bool is_lfe = P.lfe && ch == nch - 1;
            const int gs = stg == 1 ? 1 : stg == 2 ? 2 : 4;
            const int ng = ((is_lfe ? 7 : nbc) + gs * 3 - 4) / (3 * gs);
            const uint8_t *e = &W.erow[0];
            {
                uint32_t row = 0;
#pragma unroll
                for (int c2 = 0; c2 < 6; c2++) row = c2 == ch ? ew[c2] : row;
                WAVE_SYNC();                                            // the previous channel's groups have read the row
                *reinterpret_cast<uint32_t *>(&W.erow[4 * lane]) = row;
                WAVE_SYNC();
            }
            put(4, (uint32_t)__builtin_amdgcn_readfirstlane((int)e[0]));
            flush();
            for (int g = lane; g < ng; g += 64) {
                const int k0 = 1 + 3 * g * gs;
                const int prev = g ? e[k0 - gs] : e[0];
                const int d0 = e[k0] - prev + 2, d1 = e[k0 + gs] - e[k0] + 2, d2 = e[k0 + 2 * gs] - e[k0 + gs] + 2;
                put_bits(fr, P.frw, pos + 7 * g, 7, (uint32_t)((d0 * 5 + d1) * 5 + d2));
            }
            pos += 7 * ng;
            if (!is_lfe) put(2, 0);
        }
        if (b == 0) flush();
        put(1, b == 0);
        if (b == 0) { put(2, sdecaycod); put(2, fdecaycod); put(2, sgaincod); put(2, dbkneecod); put(3, floorcod); }
        put(1, b == 0);
        if (b == 0) {
            put(6, (uint32_t)csnr);
            for (int ch = 0; ch < nch; ch++) { put(4, (uint32_t)fsnr); put(3, fgaincod); }
        }
        put(1, 0);
        put(1, 0);
        flush();

        // ---- mantissas (:1341-1501): enc_mant.h ----
        {
            MantBlock B;
            B.fr = fr; B.frw = P.frw; B.glist = W.glist; B.packlut = L.packlut;
            B.mdb = P.mdct + rowb * 256;
            B.tap_bap = P.tap_bap ? P.tap_bap + rowb * 256 : nullptr;
            B.nch = nch; B.nbc = nbc; B.lfe = P.lfe != 0; B.marker = (uint32_t)P.marker;
            const uint32_t pos_m = pos;
            pos = mant_pack_block(B, em, ad, shv, garbage, pos, lane);
            // a grouped code that equals the merged-marker (:1466-1480) takes no bits: the block is shorter than its bap codes say,
            // i.e. every later block moves - the workgroup then packs the frame once more from the counts as packed
            if (attempt == 0 && lane == 0) {
                L.blk_bits2[b] = (uint32_t)side_bits + (pos - pos_m);
                if ((int)(pos - pos_m) != mant_bits) L.any_coll = 1;
            }
        }
        __syncthreads();
        if (attempt == 1 || !L.any_coll) break;
        __syncthreads();
        for (int i = tid; i < P.frw / 4; i += 384) reinterpret_cast<uint4 *>(fr)[i] = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }

    // ---- frame end (:1599-1638): the two CRC regions on two wavefronts, then the frame goes out ----
    const int fs58 = (fs >> 1) + (fs >> 3);
    if (b == 0) {
        uint32_t crc1 = region_crc(L, fr, 2 * fs58, 2 * fs58, P.c1, P.pw1_t, 4, lane);
        crc1 = gf_mul_tab(crc1, P.crc_inv_t);
        if (lane == 0) L.crc[0] = crc1;
    } else if (b == 1) {
        const uint32_t crc2 = region_crc(L, fr, 2 * fs - 2, (fs - fs58) * 2 - 2, P.c2, P.pw2_t, 0, lane);
        if (lane == 0) L.crc[1] = crc2;
    }
    __syncthreads();
    if (tid == 0) {
        fr[0] = (fr[0] & 0xffff0000u) | (L.crc[0] & 0xffff);                      // bytes 2,3
        const int p = 2 * fs - 2;                                                     // even -> inside one dword
        const int shft = 16 - 8 * (p & 3);
        fr[p >> 2] = (fr[p >> 2] & ~(0xffffu << shft)) | ((L.crc[1] & 0xffff) << shft);
    }
    __syncthreads();
    uint8_t *dst = P.frames + fidx * P.frame_stride;
    for (int i = tid; i < (2 * fs + 3) / 4; i += 384) {
        const uint32_t v = __builtin_bswap32(fr[i]);
        const int rem = 2 * fs - 4 * i;
        if (rem >= 4) *reinterpret_cast<uint32_t *>(dst + 4 * i) = v;
        else for (int k = 0; k < rem; k++) dst[4 * i + k] = (uint8_t)(v >> (8 * k));
    }
}

#ifdef PACK_STAMPS
extern "C" __attribute__((visibility("default"))) int ac3mi_debug_pack_cycles(unsigned long long *out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pack_cycles), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
    if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_pack_cycles), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------

static uint32_t h_gf_mul(uint32_t a, uint32_t b)
{
    uint32_t c = 0;
    while (a) {
        if (a & 1) c ^= b;
        a >>= 1;
        b <<= 1;
        if (b & 0x10000u) b ^= 0x18005u;
    }
    return c;
}
static uint32_t h_gf_pow(uint32_t a, uint32_t n)
{
    uint32_t r = 1;
    while (n) {
        if (n & 1) r = h_gf_mul(r, a);
        a = h_gf_mul(a, a);
        n >>= 1;
    }
    return r;
}

hipError_t launch_encode(const DeviceTables &tab, const EncodeLaunch &E, hipStream_t stream)
{
    if (E.n_streams <= 0 || E.frames_per_stream <= 0) return hipSuccess;
    const EncConfig &c = E.cfg;
    MdctParams M;
    M.pcm = E.pcm;
    M.last = E.last;
    M.mdct = E.ws_mdct;
    M.full_rows = E.mdct_full_rows ? 1 : 0;
    M.expo = E.ws_expo;
    M.shift = E.ws_shift;
    M.tab = tab.enc;
    M.n_streams = E.n_streams;
    M.frames = E.frames_per_stream;
    M.nch = c.nch;
    for (int i = 0; i < 8; i++) M.chmap[i] = E.chmap[i];
    M.slot = E.slot;
    M.store_history = E.frames_per_stream == 1;
    M.x.eexp = E.ws_eexp;
    M.x.emask = E.ws_emask;
    M.x.strat = E.ws_strat;
    M.x.ebits = E.ws_ebits;
    M.x.tab = tab.enc;
    M.x.nch = c.nch;
    M.x.lfe = c.lfe;
    M.x.fscod = c.fscod;
    M.x.halfrate = c.halfrate;
    M.x.nbc = 223;      // the only reader of last[] is this same wavefront's block 0
    hipLaunchKernelGGL(enc_mdct_kernel, dim3(E.n_streams * E.frames_per_stream * c.nch), dim3(64), 0, stream, M);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;

    PackParams P;
    P.mdct = E.ws_mdct;
    P.eexp = E.ws_eexp;
    P.emask = E.ws_emask;
    P.strat = E.ws_strat;
    P.ebits = E.ws_ebits;
    P.shift = E.ws_shift;
    P.csnr_state = E.csnr;
    P.slot = E.slot;
    P.frames = E.frames;
    P.tab = tab.enc;
    P.tap_eexp = E.tap_eexp;
    P.tap_bap = E.tap_bap;
    P.tap_strat = E.tap_strat;
    P.tap_snr = E.tap_snr;
    P.n_streams = E.n_streams;
    P.frames_per_stream = E.frames_per_stream;
    P.frame_stride = E.frame_stride;
    P.nch = c.nch;
    P.nfbw = c.nfbw;
    P.lfe = c.lfe;
    P.acmod = c.acmod;
    P.fscod = c.fscod;
    P.halfrate = c.halfrate;
    P.bsid = c.bsid;
    P.frmsizecod = c.frmsizecod;
    P.frame_words = c.frame_words;
    P.nbc = 223;
    P.chbwcod = 50;
    const int fs = c.frame_words, fs58 = (fs >> 1) + (fs >> 3);
    auto times_x = [](uint16_t *t, uint32_t v) {                    // t[i] = v * x^i mod poly
        for (int i = 0; i < 16; i++) { t[i] = (uint16_t)v; v = h_gf_mul(v, 2); }
    };
    times_x(P.crc_inv_t, h_gf_pow(0x18005 >> 1, 16 * fs58 - 16));
    const int len1 = 2 * fs58, len2 = (fs - fs58) * 2 - 2;
    P.c1 = (len1 + 63) / 64;
    P.c2 = (len2 + 63) / 64;
    for (int k = 0; k < 6; k++) {
        times_x(P.pw1_t[k], h_gf_pow(2, 8u * P.c1 * (1u << k)));
        times_x(P.pw2_t[k], h_gf_pow(2, 8u * P.c2 * (1u << k)));
    }
    P.snr = E.ws_snr;
    P.memo = nullptr;
    P.hint = E.search_hint;
    P.hint_stride = E.search_hint_stride;
#ifndef ENC_FR_HEADROOM
#define ENC_FR_HEADROOM 256
#endif
    P.frw = ((2 * fs + ENC_FR_HEADROOM + 15) / 16) * 4;
    P.marker = getenv("AC3MI_ENC_MARKER") ? atoi(getenv("AC3MI_ENC_MARKER")) : 128;      // test aid (read per launch), see PackParams::marker
    static const int lds_pad = getenv("AC3MI_ENC_LDS_PAD") ? atoi(getenv("AC3MI_ENC_LDS_PAD")) : 0;      // profiling aid: occupancy sweeps
    const size_t fr_lds = (size_t)P.frw * 4 + lds_pad;
    // The SNR-offset searches run one wavefront per stream (PART 1; for few long streams behind PART 3's frame-parallel
    // tabulation), then every frame is packed by a workgroup of six wavefronts, one per audio block (enc_packb_kernel).
    // Measured per call on one-frame streams, cold (profiles/encode_cold.py, end of round 3): 64 frames 0.107 against 0.162 ms
    // with one wavefront per frame, 256: 0.111 / 0.168, 512: 0.134 / 0.179, 1 024: 0.191 / 0.192, 2 048: 0.291 / 0.262,
    // 4 096: 0.497 / 0.394, 8 192: 0.917 / 0.700 - a frame's chain is a sixth as long, but a workgroup's wavefronts wait for each
    // other five times per frame, so the chip holds fewer busy wavefronts: the block packer takes batches of up to 1 024
    // frames (the byte-stream layer's chunks), one wavefront per frame the rest.
    // E.pack_mode (ac3mi_set_encode_mode): 0 = that rule, 1 = never, 2 = always the block packer.
    const unsigned nfr = (unsigned)E.n_streams * (unsigned)E.frames_per_stream;
    const bool long_streams = E.frames_per_stream > 1 && E.n_streams < 5120;
    const bool packb = E.pack_mode == 2 || (E.pack_mode == 0 && nfr <= 1024);
    // Always two steps: the searches (enc_search_kernel), then the packers - one wavefront per frame (enc_packf_kernel) beyond
    // 1 024 frames.  (Rounds 1-3 also had a one-kernel packer per stream; it held neither half's registers comfortably.)
    P.memo = long_streams && E.n_streams < 2048 ? E.ws_memo : nullptr;      // worth its cost only when the per-stream replay is the long pole
    if (P.memo) hipLaunchKernelGGL(enc_search_kernel<3>, dim3(nfr), dim3(64), 0, stream, P);
    hipLaunchKernelGGL(enc_search_kernel<1>, dim3(E.n_streams), dim3(64), 0, stream, P);
    if (packb) hipLaunchKernelGGL(enc_packb_kernel, dim3(nfr), dim3(384), fr_lds, stream, P);
    else if (c.nch == 6 && c.nfbw == 5 && c.lfe && c.acmod == 7 && P.nbc == 223) hipLaunchKernelGGL(enc_packf_kernel<true>, dim3(nfr), dim3(64), fr_lds, stream, P);
    else hipLaunchKernelGGL(enc_packf_kernel<false>, dim3(nfr), dim3(64), fr_lds, stream, P);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    // the new history: last 256 samples per channel of each stream's final frame
    if (M.store_history) return hipSuccess;
    return launch_enc_history(E, stream);
}

__global__ void enc_history_kernel(const int16_t *pcm, int16_t *last, int n_streams, int frames, int nch, const uint8_t c0,
                                   const uint8_t c1, const uint8_t c2, const uint8_t c3, const uint8_t c4, const uint8_t c5,
                                   const int32_t *slot)
{
    const uint8_t chmap[6] = {c0, c1, c2, c3, c4, c5};
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // (s*nch + ch)*256 + j
    if (idx >= n_streams * nch * 256) return;
    const int j = idx & 255, ch = (idx >> 8) % nch, s = (idx >> 8) / nch;
    const int16_t *fp = pcm + ((size_t)s * frames + (frames - 1)) * 1536 * nch;
    const int16_t v = fp[(size_t)(5 * 256 + j) * nch + chmap[ch]];
    if (slot) last[((size_t)slot[s] * 6 + ch) * 256 + j] = v;
    else last[idx] = v;
}

hipError_t launch_enc_history(const EncodeLaunch &E, hipStream_t stream)
{
    const int n = E.n_streams * E.cfg.nch * 256;
    hipLaunchKernelGGL(enc_history_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, E.pcm, E.last, E.n_streams,
                       E.frames_per_stream, E.cfg.nch, E.chmap[0], E.chmap[1], E.chmap[2], E.chmap[3], E.chmap[4], E.chmap[5], E.slot);
    return hipGetLastError();
}

}  // namespace ac3mi
