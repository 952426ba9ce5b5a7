// decode.hip — AC-3 frame front end on gfx950: BSI + audio-block side information,
// exponent decode, parametric bit allocation, mantissa unpack + dequantisation, dither,
// coupling, rematrixing.  Output: the dequantised, gain-scaled coefficient planes that
// xform.hip turns into PCM.  Replaces a52_frame (L52/parse.c:131-205) and everything in
// a52_block up to the transform stage (L52/parse.c:558-879), a52_bit_allocate
// (L52/bit_allocate.c:124-265) and the bit reader (L52/bitstream.c/.h).
//
// One wavefront (= one 64-thread workgroup) owns one stream and walks its frames in
// order, because the dither LFSR and the exponent / bit-allocation state carry over.
//  * the frame is staged once into (dynamic) LDS as byte-swapped dwords
//  * side information is serial: parsed by all lanes on wave-uniform values through a 64-bit
//    scalar bit window (struct Rd)
//  * exponents: one lane per 7-bit group, DPP prefix sum of the deltas
//  * bit allocation: the whole wavefront per channel (bit_allocate_wave): one lane per band for
//    the PSD integration, lowcomp as an automaton evaluated by scans, leaks as prefix minima
//  * mantissas: one sweep, 64 consecutive coefficients of a channel segment per step; two packed
//    DPP scans give grouped-code ranks, bit offsets and dither draw indices (the LFSR is
//    GF(2)-linear, so the k-th draw is a table lookup: lfsr_seq[(lfsr_idx[state] + k) mod 65535]);
//    dequantisation through one table (L1-resident); planes go straight to HBM
//
// Built with -ffp-contract=off: coefficient values are bit-identical to liba52's.
#include "decode_common.h"
#include "mant2.h"

namespace ac3mi {

// ---------------------------------------------------------------------------

// MODE 0: one wavefront per stream, frames in order (the dither LFSR carries from frame to frame), everything in one
// kernel: the front end of rounds 1-2, kept as the bit-identity reference of the tests and for A/B runs (ac3mi_set_decode_mode 1).
// Nothing else carries across the frames of a valid stream (block 0 re-sends exponents, coupling and bit-allocation
// parameters): the frame-parallel variant below (MODE 5) rests on that.
// Wavefronts per SIMD the register budget is set for.  MODE 0 runs as fast with 4 (128 VGPRs, 52 B of scratch) as with 5
// (96 VGPRs, 160 B of scratch) on one-frame streams and 9 % faster on 8-frame streams, and leaves a third of the spill traffic.
#ifndef DEC_LB0
#define DEC_LB0 4
#endif
// MODE 4 (one wavefront per stream) / MODE 5 + lfsr_prefix_kernel (one per frame): the parse half of the split front
// end - side information, exponents, bit allocation; the mantissas of a block are only COUNTED, from per-row totals, and
// mant_kernel (one wavefront per audio block, a workgroup per frame) unpacks them from the block descriptors and rows
// this kernel leaves in the workspace (BlkDesc, decode_common.h).
#ifndef DEC_LBP
#define DEC_LBP 5
#endif
// Measurement aid (make EXTRA=-DDEC_STAMPS, a separate library): lane 0 of every wavefront adds the s_memtime cycles of
// a frame's sections to g_dec_cycles: 0 staging + header, 1 side information, 2 exponents, 3 bit-allocation parameters +
// bit allocation, 4 mantissas (+ coupling, rematrix, stores), 5 the rest.  ac3mi_debug_dec_cycles reads them.
#ifdef DEC_STAMPS
__device__ unsigned long long g_dec_cycles[8];
#define DK_DECL() unsigned long long dk_t = 0, dk_acc[6] = {0, 0, 0, 0, 0, 0}
#define DK_T0() dk_t = __builtin_readcyclecounter()
#define DK_LAP(id) do { const unsigned long long t_ = __builtin_readcyclecounter(); dk_acc[id] += t_ - dk_t; dk_t = t_; } while (0)
#define DK_END() do { if (lane == 0) for (int i_ = 0; i_ < 6; i_++) atomicAdd(&g_dec_cycles[i_], dk_acc[i_]); } while (0)
#else
#define DK_DECL() do { } while (0)
#define DK_T0() do { } while (0)
#define DK_LAP(id) do { } while (0)
#define DK_END() do { } while (0)
#endif

template <int MODE>
__global__ __launch_bounds__(64, MODE >= 4 ? DEC_LBP : DEC_LB0) void decode_kernel(const DecodeParams P)
{
    static_assert(MODE == 0 || MODE == 4 || MODE == 5, "the one-kernel front ends per frame (1, 2) were retired in round 4");
    constexpr bool SERIAL = MODE == 0 || MODE == 4;         // one wavefront per stream, frames in order
    constexpr bool PARSE = MODE >= 4;                       // no mantissa values: block descriptors + rows for mant_kernel
    __shared__ DecLDS L;
    extern __shared__ uint32_t frw[];
    const FrameBits FB{frw, (uint32_t)((P.frame_bytes + 3) >> 2) + 2u};
    const int lane = threadIdx.x;
    const int s = SERIAL ? (int)blockIdx.x : (int)(blockIdx.x / (unsigned)P.frames_per_stream);
    const int f_first = SERIAL ? 0 : (int)(blockIdx.x - (unsigned)s * (unsigned)P.frames_per_stream);
    const int f_end = SERIAL ? P.frames_per_stream : f_first + 1;
    if (s >= P.n_streams) return;

    // ---- constant tables into LDS ----
    for (int i = lane; i < 256; i += 64) L.la_neg[i] = P.tab->la_neg[i];
    if (lane < 50) L.hth[lane] = 0;
    L.width[lane] = remap_width(P.tab->width[lane]);        // row bytes: see decode_common.h, mantissa stage
    for (int i = lane; i < 100; i += 64) L.desc[i] = mant_desc((uint32_t)i);
    if (lane < 30) L.band_end[lane] = P.tab->band_end[lane];
    for (int i = lane; i < 256; i += 64) L.band_of_bin[i] = P.tab->band_of_bin[i];
    {
        // exp, bap, deltba and cplco are adjacent and dword-sized together: zeroed as dwords (a byte loop took 26 rounds)
        static_assert(offsetof(DecLDS, exp) == 0 && offsetof(DecLDS, bap) == ROWS && offsetof(DecLDS, deltba) == 2 * ROWS &&
                      offsetof(DecLDS, cplco) == 2 * ROWS + 6 * 52 && offsetof(DecLDS, gcode) == 2 * ROWS + 6 * 52 + 90 * 4 && ROWS % 4 == 0, "layout");
        uint32_t *z = reinterpret_cast<uint32_t *>(&L);
        for (int i = lane; i < (2 * ROWS + 6 * 52 + 90 * 4) / 4; i += 64) z[i] = 0u;
    }

    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int ba_lo0, ba_hi0;                 // the lane's wide band (ba_wide_psd)
    ba_lane_band(L, lane, ba_lo0, ba_hi0);

    St st;
    st.acmod = st.lfeon = 0;
    st.nf = 0;
    st.cfg = 1u << 4;                   // dynrnge 1, the rest 0
    st.cplw = 0;
    st.clev = st.slev = st.level = st.dynrng = 0.f;
    st.output = 0;
    st.chincpl = st.cplstrtmant = 0;
    st.cplbndstrc = 0;
    st.ends = 7ull << 40;
    st.cbai8 = 0;
    st.deltbae2 = 0xaaau;
    const int sslot = P.slot ? P.slot[s] : s;
    DK_DECL();
    st.lfsr = (MODE == 0 || MODE == 4) ? (uint32_t)P.lfsr_state[sslot] : 1u;
    int hth_fscod = -1;
    uint32_t frame_draws = 0;
    // MODE 4: the generator's position along its cycle instead of its state (k draws = k positions)
    const bool pos_live = st.lfsr != 0;
    uint32_t lfsr_pos = MODE == 4 ? (uint32_t)P.lfsr_idx[st.lfsr] : 0u;
    if (PARSE && lane < 7) L.tot[lane][2] = 0;

    for (int f = f_first; f < f_end; f++) {
        const size_t fidx = (size_t)s * P.frames_per_stream + f;
        const uint8_t *src = P.frames + fidx * P.frame_stride;
        float *cout = P.coef + fidx * 6 * P.n_in * 256;
        uint32_t status = 0;
        // block 0 takes exponents, coupling or bit-allocation parameters the frame did not send (not a conforming frame):
        // what it reuses is whatever the variant at hand has carried so far, so results may depend on the batch shape
        bool reuse0 = false;
        // parse modes: rows of this frame's row sets not written yet (bit = slot), and the block that holds each slot's
        // current row (4 bits per slot)
        int dirty_exp = 0x7f, dirty_bap = 0x7f;
        uint32_t rv_exp = 0, rv_bap = 0;            // lane = slot
        if (MODE == 4) frame_draws = 0;

        DK_T0();
        // ---- stage the frame: byte-swapped dwords, zero padded ----
        {
            const int nw = (P.frame_bytes + 3) >> 2;
            const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
            // (the mantissa stage reads three dwords from index <= nw + 2.)  Eight loads in flight per lane before the first
            // is used: a 1536-byte frame is one round instead of seven dependent ones
            for (int base = 0; base < nw + 6; base += 512) {
                uint32_t v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = base + lane + 64 * k;
                    v[k] = i < nw ? s32[i] : 0u;
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int i = base + lane + 64 * k;
                    const int rem = P.frame_bytes - 4 * i;
                    uint32_t x = v[k];
                    if (rem < 4 && rem > 0) x &= (1u << (8 * rem)) - 1u;
                    if (i < nw + 6) frw[i] = __builtin_bswap32(x);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        Rd rd{FB, 0, 0, 0};
        // ---- a52_syncinfo (parse.c:86-129) + a52_frame (parse.c:131-205) ----
        bool hdr_ok = true;
        {
            const uint32_t w0 = rfl(frw[0]), w1 = rfl(frw[1]);
            const int b4 = (w1 >> 24) & 0xff, b5 = (w1 >> 16) & 0xff, b6 = (w1 >> 8) & 0xff;
            if ((w0 >> 16) != 0x0b77) hdr_ok = false;
            if (b5 >= 0x60) hdr_ok = false;
            if ((b4 & 63) >= 38 || (b4 & 0xc0) == 0xc0) hdr_ok = false;
            if (hdr_ok) {
                st.set_fscod(b4 >> 6);
                const int bsid = b5 >> 3;
                st.set_halfrate(bsid < 9 ? 0 : bsid - 8);
                st.acmod = b6 >> 5;
                if (st.acmod != P.acmod) hdr_ok = false;
                const int code = b4 & 63, rate = k_kbps[code >> 1];
                const int fbytes = st.fscod() == 0 ? 4 * rate : st.fscod() == 1 ? 2 * (320 * rate / 147 + (code & 1)) : 6 * rate;
                if (fbytes > P.frame_bytes) hdr_ok = false;        // frame_bytes = the largest frame of the batch (44.1 kHz alternates)
            }
        }
        if (hdr_ok) {
            int acmod = st.acmod;
            rd.seek(6 * 8 + 3);
            if (acmod == 2 && rd.get(2) == 2) acmod = 10;                 // dsurmod -> DOLBY
            st.clev = st.slev = 0.f;
            if ((acmod & 1) && acmod != 1) st.clev = k_clev[rd.get(2)];
            if (acmod & 4) st.slev = k_slev[rd.get(2)];
            st.lfeon = rd.get(1);
            if (st.lfeon != P.lfeon) hdr_ok = false;
            float level = P.level;
            st.output = a52_downmix_init_hd(acmod, P.req_flags, &level, st.clev, st.slev);
            if (st.output < 0) hdr_ok = false;
            if (hdr_ok) {
                if (st.lfeon && (P.req_flags & AC3MI_LFE)) st.output |= AC3MI_LFE;
                st.dynrng = st.level = level * 2;
                st.set_dynrnge(P.dynrng_on ? 1 : 0);
                st.deltbae2 = 0xaaau;
                int twice = !acmod;
                do {
                    rd.get(5);
                    if (rd.get(1)) rd.get(8);
                    if (rd.get(1)) rd.get(8);
                    if (rd.get(1)) rd.get(7);
                } while (twice--);
                rd.get(2);
                if (rd.get(1)) rd.get(14);
                if (rd.get(1)) rd.get(14);
                if (rd.get(1)) {
                    int len = rd.get(6);
                    do rd.get(8); while (len--);
                }
                st.nf = k_nfchans[st.acmod];
                if (hth_fscod != st.fscod()) {
                    if (lane < 50) L.hth[lane] = P.tab->hth[st.fscod()][lane];
                    hth_fscod = st.fscod();
                }
                status |= (uint32_t)st.output << 16;
            }
        }
        if (!hdr_ok) status |= 0x100u | 0x3fu;

        bool frame_dead = !hdr_ok;
        DK_LAP(0);
        for (int blk = 0; blk < 6; blk++) {
            float *cblk = cout + (size_t)blk * P.n_in * 256;
            const int in_lfe = P.lfeon ? 1 : 0;
            int err = frame_dead ? 1 : 0;
            int blkswm = 0, dithmask = 0;
            bool bd_ok = false;
            float gain[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            const int nf = st.nf;

            if (!err) do {
                // ---- side information: parse.c:572-701 ----
                // blksw[ch], dithflag[ch]: nf flags each, channel 0 first = in the field's top bit
                blkswm = (int)(__builtin_bitreverse32(rd.get(nf)) >> (32 - nf));
                dithmask = (int)(__builtin_bitreverse32(rd.get(nf)) >> (32 - nf));
                int twice = !st.acmod, word = 0;
                do {
                    if (rd.get(1)) {
                        const int code = rd.sget(8);
                        if (st.dynrnge()) st.dynrng = st.level * dynrng_range(P, code, (fidx * 6 + blk) * 2 + word, lane);
                    }
                    word++;
                } while (twice--);

                if (rd.get(1)) {                                            // cplstre
                    st.chincpl = 0;
                    if (rd.get(1)) {                                        // cplinu
                        st.chincpl = (int)(__builtin_bitreverse32(rd.get(nf)) >> (32 - nf));
                        if (st.acmod < 2) { err = 1; break; }
                        if (st.acmod == 2) st.set_phsflginu(rd.get(1));
                        const int begf = rd.get(4), endf = rd.get(4);
                        if (endf + 3 - begf < 0) { err = 1; break; }
                        const int nsub = endf + 3 - begf;
                        int ncplbnd = nsub;
                        st.set_cplstrtbnd(k_cpl_bnd0[begf]);
                        st.cplstrtmant = begf * 12 + 37;
                        st.set_endm(6, endf * 12 + 73);
                        st.cplbndstrc = 0;
                        for (int i = 0; i < nsub - 1; i++)
                            if (rd.get(1)) { st.cplbndstrc |= 1u << i; ncplbnd--; }
                        st.set_ncplbnd(ncplbnd);
                    }
                } else if (blk == 0) reuse0 = true;
                if (st.chincpl) {                                           // coupling coordinates
                    int any = 0;
                    for (int i = 0; i < nf; i++)
                        if ((st.chincpl >> i) & 1) {
                            if (rd.get(1)) {
                                const int master = 3 * rd.get(2);
                                any = 1;
                                for (int j = 0, nb = st.ncplbnd(); j < nb; j++) {
                                    const int ex = rd.get(4);
                                    int ma = rd.get(4);
                                    ma = (ex == 15) ? (ma << 14) : ((ma | 0x10) << 13);
                                    const float co = (float)ma * sf_of(ex + master);
                                    if (lane == 0) L.cplco[i][j] = co;
                                }
                            } else if (blk == 0) reuse0 = true;
                        }
                    if (st.acmod == 2 && st.phsflginu() && any)
                        for (int j = 0, nb = st.ncplbnd(); j < nb; j++)
                            if (rd.get(1) && lane == 0) L.cplco[1][j] = -L.cplco[1][j];
                }
                if (st.acmod == 2) {
                    if (rd.get(1)) {                                        // rematstr
                        const int end = st.chincpl ? st.cplstrtmant : 253;
                        int i = 0;
                        int rematflg = 0;
                        do rematflg |= rd.get(1) << i; while (k_remat_edge[1 + i++] < end);
                        st.set_rematflg(rematflg);
                    } else if (blk == 0) reuse0 = true;
                }
                int cplexpstr = 0, lfeexpstr = 0, chexp = 0;   // chexp: 2 bits per channel
                if (st.chincpl) cplexpstr = rd.get(2);
                {                                               // chexpstr[ch]: nf two-bit codes, channel 0 first
                    const uint32_t r = __builtin_bitreverse32(rd.get(2 * nf)) >> (32 - 2 * nf);       // channel order right, each code's bits swapped
                    chexp = (int)(((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u));
                }
                if (st.lfeon) lfeexpstr = rd.get(1);
                if (blk == 0) {
                    if (st.chincpl && !cplexpstr) reuse0 = true;
                    if (st.lfeon && !lfeexpstr) reuse0 = true;
                    for (int i = 0; i < nf; i++) if (!((chexp >> (2 * i)) & 3)) reuse0 = true;
                }
#pragma unroll
                for (int i = 0; i < 5; i++)
                    if (i < nf && !err && ((chexp >> (2 * i)) & 3)) {
                        if ((st.chincpl >> i) & 1) st.set_endm(i, st.cplstrtmant);
                        else {
                            const int bw = rd.get(6);
                            if (bw > 60) err = 1;
                            else st.set_endm(i, bw * 3 + 73);
                        }
                    }
                if (err) break;

                DK_LAP(1);
                // ---- exponents: parse.c:703-736 ----
                int redo = 0;
                if (cplexpstr) {
                    const int ngrp = (st.cplendmant() - st.cplstrtmant) / (3 << (cplexpstr - 1));
                    const int e0 = rd.get(4) << 1;
                    redo = 64;
                    if (read_exponents(FB, rd.pos(), cplexpstr, ngrp, e0, L.exp + row_off(6) + st.cplstrtmant, lane)) { err = 1; break; }
                    rd.skip(7 * ngrp);
                }
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const int es = (chexp >> (2 * i)) & 3;
                    if (i < nf && !err && es) {
                        const int gs = 3 << (es - 1), ngrp = (st.endm(i) + gs - 4) / gs;
                        redo |= 1 << i;
                        const int e0 = rd.get(4);
                        if (lane == 0) L.exp[row_off(i)] = (uint8_t)e0;
                        if (read_exponents(FB, rd.pos(), es, ngrp, e0, L.exp + row_off(i) + 1, lane)) err = 1;
                        rd.skip(7 * ngrp);
                        rd.get(2);                                          // gainrng
                    }
                }
                if (err) break;
                if (lfeexpstr) {
                    redo |= 32;
                    const int e0 = rd.get(4);
                    if (lane == 0) L.exp[row_off(5)] = (uint8_t)e0;
                    if (read_exponents(FB, rd.pos(), lfeexpstr, 2, e0, L.exp + row_off(5) + 1, lane)) { err = 1; break; }
                    rd.skip(14);
                }

                DK_LAP(2);
                dirty_exp |= redo;                                          // (bits: 0..4 fbw, 5 lfe, 6 coupling channel)
                // ---- bit-allocation parameters: parse.c:738-772 ----
                if (rd.get(1)) { redo = 127; st.set_bai(rd.get(11)); }
                else if (blk == 0) reuse0 = true;
                if (rd.get(1)) {
                    redo = 127;
                    st.set_csnroffst(rd.get(6));
                    if (st.chincpl) st.set_cbai(6, rd.get(7));
#pragma unroll
                    for (int i = 0; i < 5; i++) if (i < nf) st.set_cbai(i, rd.get(7));
                    if (st.lfeon) st.set_cbai(5, rd.get(7));
                } else if (blk == 0) reuse0 = true;
                if (st.chincpl) {
                    if (rd.get(1)) {
                        redo |= 64;
                        st.set_cplfleak(9 - rd.get(3));
                        st.set_cplsleak(9 - rd.get(3));
                    } else if (blk == 0) reuse0 = true;
                }
                if (rd.get(1)) {                                            // deltbaie
                    redo = 127;
                    if (st.chincpl) st.set_deltbae(5, rd.get(2));
#pragma unroll
                    for (int i = 0; i < 5; i++) if (i < nf) st.set_deltbae(i, rd.get(2));
#pragma unroll
                    for (int pass = 0; pass < 6; pass++) {
                        const int slot = pass == 0 ? 5 : pass - 1;          // cpl first, then fbw (parse.c:763-771)
                        if (err || pass > nf) continue;
                        if (slot == 5 && !st.chincpl) continue;
                        if (st.deltbae(slot) != 1) continue;
                        // parse_deltba: parse.c:272-294
                        if (lane < 50) L.deltba[slot][lane] = 0;
                        int nseg = rd.get(3), band = 0;
                        do {
                            band += rd.get(5);
                            int len = rd.get(4), d = rd.get(3);
                            d -= (d >= 4) ? 3 : 4;
                            if (!len) continue;
                            if (band + len >= 50) { err = 1; break; }
                            if (lane < len) L.deltba[slot][band + lane] = (int8_t)d;
                            band += len;
                        } while (nseg--);
                    }
                    if (err) break;
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

                // ---- bit allocation: parse.c:774-798 ----
                if (redo) {
                    bool allzero = !st.csnroffst() && !(st.chincpl && (st.cbai(6) >> 3)) && !(st.lfeon && (st.cbai(5) >> 3));
#pragma unroll
                    for (int i = 0; i < 5; i++)
                        if (i < nf && (st.cbai(i) >> 3)) allzero = false;
                    dirty_bap |= allzero ? 0x7f : redo;
                    if (PARSE && lane < 7 && (allzero || ((redo >> lane) & 1))) L.tot[lane][2] = 0;
                    if (allzero) {
                        for (int i = lane; i < ROWS; i += 64) L.bap[i] = 0;
                    } else {
                        // the channel slots that need a new allocation, two per sweep of the band PSDs (wave-uniform; written
                        // out rather than as lambdas over `st`: a closure that holds its address keeps the whole state in scratch)
#define AC3MI_SLOT_END(slot) st.endm(slot)
                        int todo = redo & (((1 << nf) - 1) | (st.lfeon ? 32 : 0) | (st.chincpl ? 64 : 0));
#pragma unroll
                        for (int i = 0; i < 5; i++) if (st.endm(i) <= 0) todo &= ~(1 << i);
                        if (st.cplendmant() <= st.cplstrtmant) todo &= ~64;
                        while (todo) {
                            const int sA = __builtin_ctz(todo);
                            todo &= todo - 1;
                            const bool two = todo != 0;
                            const int sB = two ? __builtin_ctz(todo) : sA;
                            if (two) todo &= todo - 1;
                            const int stA = sA == 6 ? st.cplstrtmant : 0, stB = sB == 6 ? st.cplstrtmant : 0;
                            const int enA = AC3MI_SLOT_END(sA), enB = AC3MI_SLOT_END(sB);
                            const int wide = ba_wide_psd(L, ba_lo0, ba_hi0, L.exp + row_off(sA), stA, enA, L.exp + row_off(sB), stB, enB, two, lane);
                            for (int h = 0; h < (two ? 2 : 1); h++) {          // (one call site: the routine is inlined once)
                                const int slot = h ? sB : sA, start = h ? stB : stA, end = h ? enB : enA;
                                const int mybai = st.cbai(slot);
                                const int mydeltbae = slot == 5 ? 2 : st.deltbae(slot == 6 ? 5 : slot);
                                BaCtx c;
                                const int bai = st.bai();
                                c.halfrate = st.halfrate();
                                c.fdecay = (63 + 20 * ((bai >> 7) & 3)) >> c.halfrate;
                                c.fgain = 128 + 128 * (mybai & 7);
                                c.sdecay = (15 + 2 * (bai >> 9)) >> c.halfrate;
                                c.sgain = k_slowgain[(bai >> 5) & 3];
                                c.dbknee = k_dbpb[(bai >> 3) & 3];
                                c.hth = L.hth;
                                c.deltba = (mydeltbae == 2) ? nullptr : L.deltba[slot == 6 ? 5 : slot];
                                const int fl = k_floors[bai & 7];
                                c.snroffset = 960 - 64 * st.csnroffst() - 4 * (mybai >> 3) + fl;
                                c.floor = fl >> 5;
                                c.fast = slot == 6 ? st.cplfleak() << 8 : 0;
                                c.slow = slot == 6 ? st.cplsleak() << 8 : 0;
                                bit_allocate_finish(L, L.bmask, c, slot == 6 ? st.cplstrtbnd() : 0, start, end, L.exp + row_off(slot), L.bap + row_off(slot), wide, h, lane);
                            }
                        }
#undef AC3MI_SLOT_END
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                if (rd.get(1)) {                                            // skip field
                    const int n = rd.get(9);
                    rd.skip(8 * n);
                }
            } while (0);

            // optional stage taps
            if (P.tap_exp) {
                uint8_t *te = P.tap_exp + (fidx * 6 + blk) * 7 * 256;
                int8_t *tb = P.tap_bap + (fidx * 6 + blk) * 7 * 256;
                for (int c = 0; c < 7; c++)
                    for (int i = lane; i < 256; i += 64) {
                        const bool in = c != 5 || i < LFE_ROW;
                        te[c * 256 + i] = in ? L.exp[row_off(c) + i] : 0;
                        tb[c * 256 + i] = in ? unmap_width(L.bap[row_off(c) + i]) : 0;
                    }
            }

            if (!err) {
                DK_LAP(3);
                // ---- gains: parse.c:810-811 ----
                a52_downmix_coeff_hd(gain, st.acmod, st.output, st.dynrng, st.clev, st.slev);

                // ---- mantissas: the segments of the block in bitstream order (mant_block, decode_common.h) ----
                if (!PARSE && st.chincpl && lane < 18) {                    // sub-band -> band (parse.c:448-456)
                    const uint32_t below = st.cplbndstrc & ((1u << lane) - 1u);
                    L.cplbnd[lane] = (uint8_t)(lane - __popc(below));
                }
                SegBase sb;
                sb.bit = rd.pos();
                sb.r3 = sb.r5 = sb.r11 = sb.draw = 0;
                if constexpr (!PARSE) {
                    MantBlk B;
                    B.nf = nf; B.lfeon = st.lfeon; B.acmod = st.acmod; B.in_lfe = in_lfe;
                    B.chincpl = st.chincpl; B.dithmask = dithmask; B.rematflg = st.rematflg();
                    B.cplstrtmant = st.cplstrtmant; B.cplendmant = st.cplendmant();
#pragma unroll
                    for (int i = 0; i < 5; i++) { B.endmant[i] = st.endm(i); B.gain[i] = gain[i]; }
                    B.lfe_gain = (st.output & AC3MI_LFE) ? st.dynrng : 0.f;
                    const uint32_t lfsr_i0 = P.lfsr_idx[st.lfsr];
                    const bool lfsr_live = st.lfsr != 0;
                    mant_block<false>(B, [&](int slot) { return (const uint8_t *)L.exp + row_off(slot); },
                                          [&](int slot) { return (const int8_t *)L.bap + row_off(slot); },
                                          [&](int c, int bnd) { return L.cplco[c][bnd]; }, L.cplbnd, L.desc, L.gcode, frw, FB.last,
                                          P.tab->qtab, P.lfsr_seq, lfsr_i0, lfsr_live, cblk, sb, lane);
                    // advance the dither generator past this block's draws
                    if (lfsr_live && sb.draw) st.lfsr = P.lfsr_seq[(lfsr_i0 + (uint32_t)sb.draw) % 65535u];
                } else {
                    // parse only: the totals of every segment's row (cached per slot while row and range stay) give the bits and
                    // the dither draws of the block; rows that changed go to the workspace; the descriptor names them.
                    // Lane k takes segment k (the scalar unit is this kernel's bottleneck: a loop over the segments on
                    // wave-uniform values costs 600 scalar instructions per block, this form about 80 vector ones).
                    const int cplfirst = st.chincpl ? __builtin_ctz(st.chincpl) : 99;
                    const int nseg = nf + (st.chincpl ? 1 : 0) + (st.lfeon ? 1 : 0);
                    const int ncpl_dith = __popc(st.chincpl & dithmask);
                    const int k = lane;
                    const int slot_l = st.chincpl ? (k <= cplfirst ? k : k == cplfirst + 1 ? 6 : k - 1 < nf ? k - 1 : 5) : (k < nf ? k : 5);
                    const bool seg_l = k < nseg;
                    const int end_l = (int)((uint32_t)(st.ends >> (8 * slot_l)) & 0xffu);
                    const int start_l = slot_l == 6 ? st.cplstrtmant : 0;
                    const int mult_l = slot_l < 5 ? (dithmask >> slot_l) & 1 : slot_l == 6 ? ncpl_dith : 0;
                    const uint32_t key_l = 0x80000000u | (uint32_t)start_l | ((uint32_t)end_l << 10);
                    uint32_t ca = L.tot[slot_l][0], cb = L.tot[slot_l][1];
                    for (unsigned long long miss = __ballot(seg_l && L.tot[slot_l][2] != key_l); miss; miss &= miss - 1) {
                        const int k0 = __builtin_ctzll(miss);
                        const int s0 = __builtin_amdgcn_readlane(slot_l, k0);
                        const RowTotals T = row_totals(L.bap + row_off(s0), __builtin_amdgcn_readlane(start_l, k0), __builtin_amdgcn_readlane(end_l, k0),
                                                       s0 == 5 ? LFE_ROW / 4 : 64, lane);
                        if (lane == k0) { ca = T.a; cb = T.b; L.tot[s0][0] = T.a; L.tot[s0][1] = T.b; L.tot[s0][2] = key_l; }
                    }
                    {
                        const uint32_t n3 = seg_l ? (ca >> 13) & 511u : 0u, n5 = seg_l ? ca >> 22 : 0u, n11 = seg_l ? cb & 511u : 0u;
                        const uint32_t nz = seg_l ? cb >> 9 : 0u, plain = seg_l ? ca & 0x1fffu : 0u;
                        // 3/5/11-level members of the block before the segment
                        const uint32_t own35 = n3 | (n5 << 16), p35 = wave_incl_scan_u32(own35) - own35, r11 = wave_incl_scan_u32(n11) - n11;
                        const uint32_t r3 = p35 & 0xffffu, r5 = p35 >> 16;
                        // a grouped code takes its bits where the member of rank 0 mod 3 (mod 2) stands
                        auto div3 = [](uint32_t x) { return (x * 0xaaabu) >> 17; };
                        const uint32_t o3 = div3(r3 + n3 + 2u) - div3(r3 + 2u), o5 = div3(r5 + n5 + 2u) - div3(r5 + 2u);
                        const uint32_t o11 = ((r11 + n11 + 1u) >> 1) - ((r11 + 1u) >> 1);
                        const uint32_t bits_l = plain + 5u * o3 + 7u * (o5 + o11);
                        const uint32_t tot = wave_sum_u32(bits_l | ((nz * (uint32_t)mult_l) << 16));
                        sb.bit += tot & 0xffffu;
                        sb.draw = (int)(tot >> 16);
                    }
                    {
                        uint8_t *rowset = P.rows + (fidx * 6 + blk) * (size_t)ROWSET;
                        const int used = ((1 << nf) - 1) | (st.lfeon ? 32 : 0) | (st.chincpl ? 64 : 0);
                        const int we = dirty_exp & used, wb = dirty_bap & used;
                        for (int m = we; m; m &= m - 1) {
                            const int s0 = __builtin_ctz(m);
                            if (lane < (s0 == 5 ? LFE_ROW / 4 : 64))
                                *reinterpret_cast<uint32_t *>(rowset + s0 * 512 + 4 * lane) = *reinterpret_cast<const uint32_t *>(L.exp + row_off(s0) + 4 * lane);
                        }
                        for (int m = wb; m; m &= m - 1) {
                            const int s0 = __builtin_ctz(m);
                            if (lane < (s0 == 5 ? LFE_ROW / 4 : 64))
                                *reinterpret_cast<uint32_t *>(rowset + s0 * 512 + 256 + 4 * lane) = *reinterpret_cast<const uint32_t *>(L.bap + row_off(s0) + 4 * lane);
                        }
                        dirty_exp &= ~we;
                        dirty_bap &= ~wb;
                        // lane = slot: the block of this frame that holds the slot's current rows
                        rv_exp = ((we >> lane) & 1) ? (uint32_t)blk : rv_exp;
                        rv_bap = ((wb >> lane) & 1) ? (uint32_t)blk : rv_bap;
                    }
                    if (st.chincpl) {
                        float *cc = P.cplco + (fidx * 6 + blk) * 90;
                        cc[lane] = (&L.cplco[0][0])[lane];
                        if (lane < 26) cc[64 + lane] = (&L.cplco[0][0])[64 + lane];
                    }
                    // the descriptor, straight from the lanes (BlkDesc's layout)
                    {
                        uint8_t *dp = reinterpret_cast<uint8_t *>(P.desc + (fidx * 6 + blk));
                        const uint32_t fl = ((uint32_t)st.chincpl << 8) | ((uint32_t)dithmask << 16) | ((uint32_t)st.rematflg() << 24);
                        const uint32_t w0 = lane == 0 ? rd.pos() : lane == 1 ? frame_draws : lane == 2 ? fl : st.cplbndstrc;
                        if (lane < 4) reinterpret_cast<uint32_t *>(dp)[lane] = w0;
                        const int em = lane == 5 ? st.cplstrtmant : (int)((uint32_t)(st.ends >> (8 * (lane & 7))) & 0xffu);
                        if (lane < 7) reinterpret_cast<uint16_t *>(dp + 16)[lane] = (uint16_t)em;
                        if (lane < 8) { dp[32 + lane] = (uint8_t)rv_exp; dp[40 + lane] = (uint8_t)rv_bap; }
                        const float gl = lane == 0 ? gain[0] : lane == 1 ? gain[1] : lane == 2 ? gain[2] : lane == 3 ? gain[3] : lane == 4 ? gain[4]
                                       : (st.output & AC3MI_LFE) ? st.dynrng : 0.f;
                        // spare word: the frame's own SNR offsets, 16 csnroffst + fsnroffst of channel 0 - a transcode's encoder starts
                        // costing its search there (a hint: which offsets are costed never changes a result, encode.hip)
                        const uint32_t w12 = lane < 6 ? __float_as_uint(gl) : (uint32_t)(16 * st.csnroffst() + (st.cbai(0) >> 3));
                        if (lane < 7) reinterpret_cast<uint32_t *>(dp + 48)[lane] = w12;
                    }
                    bd_ok = true;
                }
                rd.seek(sb.bit);
                frame_draws += (uint32_t)sb.draw;
            }

            DK_LAP(4);
            // ---- a failed block leaves zero planes ----
            if (err) { status |= 1u << blk; frame_dead = true; }
            if constexpr (PARSE) {
                if ((err || !bd_ok) && lane == 0) {
                    reinterpret_cast<uint32_t *>(P.desc + (fidx * 6 + blk))[2] = 1u;      // flags: the block failed
                    // (no descriptor was stored: the word a transcode's encoder reads as its search hint must not be a leftover
                    // of an earlier call - 0 = no hint, see enc_search_kernel)
                    P.desc[fidx * 6 + blk].src_snr = 0u;
                }
            }
            {
                if (err && !PARSE)
                    for (int c = 0; c < P.n_in; c++)
                        *reinterpret_cast<float4 *>(cblk + (size_t)c * 256 + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.blksw && lane < P.nfchans)
                    P.blksw[(fidx * 6 + blk) * P.nfchans + lane] = (uint8_t)(err ? 0 : ((blkswm >> lane) & 1));
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        DK_LAP(5);
        if (lane == 0) P.status[fidx] = status | (reuse0 ? 0x200u : 0u);
        if (lane == 0 && P.zs) P.zs[fidx] = (uint8_t)((status & 0x100u) ? 0 : surround_level_is_zero(st.acmod, st.output, st.slev));
        if (MODE == 5 && lane == 0) P.frame_draws[fidx] = frame_draws;
        if (MODE == 4) {
            if (lane == 0) P.frame_pos[fidx] = pos_live ? lfsr_pos : 0xffffffffu;
            lfsr_pos = (lfsr_pos + frame_draws) % 65535u;
        }
    }
    if (MODE == 0 && lane == 0) P.lfsr_state[sslot] = (uint16_t)st.lfsr;
    if (MODE == 4 && lane == 0 && pos_live) P.lfsr_state[sslot] = P.lfsr_seq[lfsr_pos];
    DK_END();
}

// LFSR state at the start of every frame: one thread per stream walks its frames' draw counts (the generator is
// GF(2)-linear with period 65535: k draws = k positions along the cycle; state 0 is a fixed point).
// frame_pos (split front end): the position along the cycle instead of the state (0xffffffff: state 0), and the stream's
// final state written back here.
__global__ void lfsr_prefix_kernel(const uint32_t *draws, uint16_t *frame_lfsr, uint32_t *frame_pos, uint16_t *lfsr_state, const int32_t *slot,
                                   const uint16_t *seq, const uint16_t *idx, int n_streams, int frames)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    uint32_t state = lfsr_state[slot ? slot[s] : s];
    uint32_t pos = idx[state];
    for (int f = 0; f < frames; f++) {
        if (frame_pos) frame_pos[(size_t)s * frames + f] = state ? pos : 0xffffffffu;
        else frame_lfsr[(size_t)s * frames + f] = (uint16_t)state;
        const uint32_t k = draws[(size_t)s * frames + f];
        if (state != 0 && k) { pos = (pos + k) % 65535u; state = seq[pos]; }
    }
    if (frame_pos) lfsr_state[slot ? slot[s] : s] = (uint16_t)state;
}

// ---------------------------------------------------------------------------
// mant_kernel: the mantissa half of the split front end.  A workgroup of six wavefronts takes one frame, staged once
// into LDS; wavefront b unpacks, dequantises and stores audio block b from its BlkDesc (mant_block, decode_common.h:
// the code of the one-kernel front ends, rows and coupling coordinates from the workspace).  Nothing carries from one
// block to the next, so the six run concurrently and a wavefront's dependent chain is a sixth of a frame.
struct MantLDS {
    uint4 dsc[M2_NDESC];
    float qtab[760];
    uint8_t ring[6][M2_LDS_WAVE];
    uint8_t cplbnd[6][20];
};

#ifndef MANT_LB
#define MANT_LB 7       // (wavefronts per SIMD the register budget is set for: 8 -> 64 VGPRs but 78 scalar registers with 69 spilled,
#endif                  //  7 -> 71 / 94 / 39, 6 -> 75 / 106 / 28; decode to s16 through this kernel 3.24 / 3.18 / 3.27 ms, round 4)
__global__ __launch_bounds__(384, MANT_LB) void mant_kernel(const MantParams P)
{
    __shared__ MantLDS L;
    extern __shared__ uint32_t frw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int blk = __builtin_amdgcn_readfirstlane(tid >> 6);
    // consecutive frames on one XCD (its L2 then serves each frame's bytes and rows to the six wavefronts once):
    // the bijective remap of cdna_hip_programming.md T1
    unsigned fidx;
    {
        const unsigned n = gridDim.x, q = n >> 3, r = n & 7u, x = blockIdx.x & 7u, i = blockIdx.x >> 3;
        fidx = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
    }
    const size_t unit = (size_t)fidx * 6 + blk;
    // the block's descriptor: five 16-byte loads of one address, in flight while the frame is staged
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
    }
    float *cblk = P.coef + unit * P.n_in * 256;
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
    // the first segment's rows are requested before the workgroup meets at the barrier
    const int slot0 = seg_slot(0, B.nf, B.chincpl, B.chincpl ? __builtin_ctz(B.chincpl) : 99);
    uint2 first = make_uint2(0u, 0u);
    if (!failed) first = fetch(slot0);
    __syncthreads();
    if (failed) {                                               // a failed block leaves zero planes
        for (int c = 0; c < P.n_in; c++)
            *reinterpret_cast<float4 *>(cblk + (size_t)c * 256 + 4 * lane) = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
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
    const uint32_t i0 = lfsr_live ? (fpos + rfl(w0v.y)) % 65535u : 0u;            // the generator's position before the block's first draw
    const float *cc = P.cplco + unit * 90;
    mant_block2<256, true>(B, fetch, first, [&](int c, int bnd) { return cc[c * 18 + bnd]; }, L.cplbnd[blk], L.dsc, L.ring[blk], frw, (uint32_t)nw + 2u,
                L.qtab, reinterpret_cast<const int16_t *>(P.lfsr_seq) + 1 + i0, lfsr_live, cblk, rfl(w0v.x), lane);
}

}  // namespace ac3mi

namespace ac3mi {

#ifdef DEC_STAMPS
}  // namespace ac3mi
extern "C" __attribute__((visibility("default"))) int ac3mi_debug_dec_cycles(unsigned long long *out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(ac3mi::g_dec_cycles), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(ac3mi::g_dec_cycles), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
namespace ac3mi {
#endif

hipError_t launch_mantx(const DeviceTables &tab, const DecodeLaunch &L, const MantParams &M, hipStream_t stream);       // decode_mx.hip

hipError_t launch_decode(const DeviceTables &tab, const DecodeLaunch &L, hipStream_t stream)
{
    DecodeParams P;
    P.frames = L.frames;
    P.coef = L.coef;
    P.blksw = L.blksw;
    P.zs = L.zs;
    P.status = L.status;
    P.lfsr_state = L.lfsr;
    P.slot = L.slot;
    P.tap_exp = L.tap_exp;
    P.tap_bap = L.tap_bap;
    P.lfsr_seq = tab.lfsr_seq;
    P.lfsr_idx = tab.lfsr_idx;
    P.tab = tab.dec;
    P.n_streams = L.n_streams;
    P.frames_per_stream = L.frames_per_stream;
    P.frame_stride = L.frame_stride;
    P.frame_bytes = L.frame_bytes;
    P.req_flags = L.req_flags;
    P.level = L.level;
    P.dynrng_on = L.dynrng_on;
    P.acmod = L.acmod;
    P.lfeon = L.lfeon;
    static const int nfch[8] = {2, 1, 2, 3, 3, 4, 4, 5};
    P.nfchans = nfch[L.acmod & 7];
    P.n_in = P.nfchans + (L.lfeon ? 1 : 0);
    if (L.n_streams <= 0 || L.frames_per_stream <= 0) return hipSuccess;
    P.frame_draws = L.frame_draws;
    P.frame_lfsr = L.frame_lfsr;
    P.desc = nullptr;
    P.rows = nullptr;
    P.cplco = nullptr;
    P.frame_pos = nullptr;
    P.dyn_out = L.dyn_out;
    P.dyn_in = L.dyn_in;
    static const int lds_pad = getenv("AC3MI_DEC_LDS_PAD") ? atoi(getenv("AC3MI_DEC_LDS_PAD")) : 0;      // profiling aid: occupancy sweeps
    const size_t fr_bytes = (size_t)(((L.frame_bytes + 3) >> 2) + 6) * 4 + lds_pad;
    const unsigned units = (unsigned)L.n_streams * (unsigned)L.frames_per_stream;
    if (L.split) {
        // parse (per stream, or per frame + the generator's prefix), then one wavefront per audio block
        P.desc = (BlkDesc *)L.ws_desc;
        P.rows = L.ws_rows;
        P.cplco = L.ws_cplco;
        P.frame_pos = L.ws_fpos;
        if (!L.frame_parallel) hipLaunchKernelGGL(decode_kernel<4>, dim3(L.n_streams), dim3(64), fr_bytes, stream, P);
        else {
            hipLaunchKernelGGL(decode_kernel<5>, dim3(units), dim3(64), fr_bytes, stream, P);
            hipLaunchKernelGGL(lfsr_prefix_kernel, dim3((L.n_streams + 63) / 64), dim3(64), 0, stream, (const uint32_t *)L.frame_draws,
                               (uint16_t *)nullptr, L.ws_fpos, L.lfsr, L.slot, tab.lfsr_seq, tab.lfsr_idx, L.n_streams, L.frames_per_stream);
        }
        MantParams M;
        M.frames = L.frames;
        M.desc = (const BlkDesc *)L.ws_desc;
        M.rows = L.ws_rows;
        M.cplco = L.ws_cplco;
        M.frame_pos = L.ws_fpos;
        M.coef = L.coef;
        M.lfsr_seq = tab.lfsr_seq;
        M.tab = tab.dec;
        M.n_frames = units;
        M.frame_stride = L.frame_stride;
        M.frame_bytes = L.frame_bytes;
        M.acmod = L.acmod;
        M.lfeon = L.lfeon;
        M.n_in = P.n_in;
        M.nfchans = P.nfchans;
        static const int mant_pad = getenv("AC3MI_MANT_LDS_PAD") ? atoi(getenv("AC3MI_MANT_LDS_PAD")) : 0;      // profiling aid: occupancy proxy of a fused mantissa + transform workgroup
        if (L.fuse) return launch_mantx(tab, L, M, stream);         // one-frame streams, no downmix: the transform in the same kernel
        hipLaunchKernelGGL(mant_kernel, dim3(units), dim3(384), (size_t)(((L.frame_bytes + 3) >> 2) + 6) * 4 + mant_pad, stream, M);
        return hipGetLastError();
    }
    if (!L.frame_parallel) {
        hipLaunchKernelGGL(decode_kernel<0>, dim3(L.n_streams), dim3(64), fr_bytes, stream, P);
        return hipGetLastError();
    }
    // (rounds 1-2 also had a one-kernel front end per FRAME for few long streams - counting pass, generator prefix, full pass;
    // the split front end's per-frame parse kernel replaced it in round 3 and it was retired in round 4)
    return hipErrorInvalidValue;
}

}  // namespace ac3mi
