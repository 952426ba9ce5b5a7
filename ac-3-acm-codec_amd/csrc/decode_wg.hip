// decode_wg.hip — the AC-3 decoder as ONE kernel on gfx950: bitstream in, PCM out, no coefficient planes in HBM.
// Replaces a52_frame + a52_block (L52/parse.c:131-205, 558-940), a52_bit_allocate (L52/bit_allocate.c:124-265),
// the bit reader (L52/bitstream.c/.h) and the transform stage (L52/imdct.c:258-345, dispatch L52/parse.c:881-937)
// for batches of many independent streams.  (Few long streams keep the frame-parallel path of decode.hip + xform.hip.)
//
// One 512-thread workgroup owns one stream and walks its frames in order (dither LFSR, exponent / bit-allocation
// state and the overlap tails carry over).  Its eight wavefronts have fixed roles:
//   0..4  one full-bandwidth channel each      5  LFE and the coupling channel      6  parser      7  transformer
// and meet at three workgroup barriers per audio block (a fourth when coupling or rematrixing is in use):
//
//   B1 ----------------------------------------------------------------------------------------------------------
//      channel waves: exponents (one lane per 7-bit group), bit allocation (bit_allocate_wave), then the channel's
//                     mantissa census: plain bits, members of 3/5/11-level codes, zero-bit bins        [T1]
//      transformer:   block b-1: planes (LDS) -> registers, pre-twiddle, DFT-16, transpose             [part A]
//   B2 ----------------------------------------------------------------------------------------------------------
//      every wave:    prefix over the channel censuses in bitstream order (one lane per channel, DPP scans) ->
//                     each channel's first bit, code phases and dither draw index; the block's length
//      channel waves: mantissa fields out of the LDS frame, 4 bins per lane; openers publish their codes [T2a]
//      parser:        side information of block b+1, first half (it starts where block b ends)
//      transformer:   block b-1: DFT-8 and post-twiddle                                                [part B]
//   B3 ----------------------------------------------------------------------------------------------------------
//      channel waves: dequantise, dither, scale -> coefficient plane in LDS, zero tail included        [T2b]
//      parser:        block b+1, second half (bit-allocation parameters, skip field), gains
//      transformer:   block b-1: window, overlap-add, PCM out (float planes or interleaved s16)        [part C]
//   (B4: coupled channels take their share of the coupling channel; rematrixing                        [T2c])
//
// So the serial part of the format (side information; a block starts where the previous one's mantissas end) runs
// beside the data-parallel part, the channels of a block run side by side, and the transform of the previous block
// runs beside both.  Coefficients live in LDS for one block period; HBM sees the frame once and the PCM once.
//
// Built with -ffp-contract=off: dequantised coefficients are bit-identical to liba52's; the transform arithmetic is
// xform_core.h's (same operations as xform.hip).
#include "decode_common.h"
// the transform's arithmetic may contract to FMA like xform.hip (1e-6 RMS tolerance); everything else in this file is
// built with -ffp-contract=off (bit-identical coefficients)
#pragma clang fp contract(fast)
#include "xform_core.h"
#pragma clang fp contract(off)

namespace ac3mi {

namespace wg {

constexpr int N_WAVES = 8;
#ifndef WG_LB
#define WG_LB 4                             // waves per SIMD the register budget is set for (512-thread blocks: 2 per block)
#endif
constexpr int W_LFE = 5, W_PARSE = 6, W_XFORM = 7;
constexpr int PLANE = 272;                  // plane pitch in floats: 8-lane groups of the transformer hit different banks
constexpr int GC = 640;                     // open codes of a block per kind (<= 1272 mantissas / 2)

// what the parser publishes for one block
struct BlkInfo {
    int err;
    int halfrate;
    int mant_pos;                           // bit position of the first mantissa
    int blkswm, dithmask;
    int chincpl, cplstrtmant, cplendmant, cplstrtbnd, rematflg, remat_end, need_fix;
    int expstr[7], exp_pos[7], ngrp[7], absexp[7];      // slots 0..4 fbw, 5 lfe, 6 coupling channel
    int endmant[5];
    int redo, allzero;
    int bai, csnroffst, cbai[7], deltbae[6], cplfleak, cplsleak;
    uint32_t lfsr_i0;
    int lfsr_live;
    float gain[5], lfe_gain;
    float cplco[5][18];
    uint8_t cplbnd[20];
};

struct WgLDS {
    uint8_t exp[ROWS];
    int8_t bap[ROWS];
    int8_t deltba[6][52];
    int8_t la_neg[256];
    uint16_t hth[50];
    int8_t width[64];
    uint8_t band_end[30];
    uint8_t band_of_bin[256];
    int16_t bmask[6][52];                   // bit_allocate_wave scratch, one per channel wave
    float qtab[760];
    uint8_t gcode[3][GC];
    uint32_t cnt[7][2];                     // census per slot: [0] plain bits | n3 << 16 | n5 << 24, [1] n11 | zeros << 8
    int exp_err;                            // a channel wave found a reserved exponent code / an exponent outside 0..24
    int blkswm_hist[6];
    BlkInfo bi[2];
    float planes[6 * PLANE];
    float cplq[256];                        // coupling channel: mantissa * 2^-exp
    int16_t cplcd[256];                     //                   dither draw index of a zero-bit bin
    float2 ex[6 * EX_GROUP];                // transformer: 8x16 transposes; the s16 tile of a block afterwards
    float2 twl[128], tws[128];
    float win[256];
    float dly[6][128];                      // overlap tails of the stream
};

// what the reference's converters make of a float sample at bias 384 (src/AC3ASM.asm:303-318: psubd, packssdw)
__device__ __forceinline__ int16_t to_s16(float v)
{
    int i = (int)(__float_as_uint(v) - 0x43c00000u);
    i = i > 32767 ? 32767 : i < -32768 ? -32768 : i;
    return (int16_t)i;
}

__device__ __forceinline__ void wg_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int ldsu(const int &v) { return (int)rfl((uint32_t)v); }           // wave-uniform LDS word
__device__ __forceinline__ float ldsf(const float &v) { return __uint_as_float(rfl(__float_as_uint(v))); }

struct WgParams {
    DecodeParams d;             // frames, status, lfsr_state, slot, taps, tables, shape, request (coef/blksw: OUT 0 and taps)
    float *delay;               // [S or slots][delay_stride]
    int delay_stride;
    float *pcm;                 // OUT 1: [S][F][6][n_out][256]
    int16_t *pcm16;             // OUT 2: [S][F][6][256][n_out] interleaved WAVE order
    const float2 *tw_long, *tw_short;
    const float *window;
    float bias;
    int n_out;
    int8_t wslot[6];            // OUT 2: plane o -> WAVE slot
};

// parser-private values that travel from the first to the second half of a block's side information
struct ParseCarry {
    int cplexpstr, lfeexpstr, chexp, redo;
};

// ---- side information, first half: L52/parse.c:572-736 without the exponent payloads (their positions are recorded) ----
__device__ void parse_block_a(Rd &rd, St &st, BlkInfo &B, const BlkInfo &prev, ParseCarry &pc, int blk, uint32_t &reuse0, int lane)
{
    const int nf = st.nf;
    int err = 0, blkswm = 0, dithmask = 0;
    pc.cplexpstr = pc.lfeexpstr = pc.chexp = pc.redo = 0;
    // coupling coordinates and band structure persist from block to block (and frame to frame)
    for (int i = lane; i < 90; i += 64) (&B.cplco[0][0])[i] = (&prev.cplco[0][0])[i];
    do {
        for (int i = 0; i < nf; i++) blkswm |= rd.get(1) << i;
        for (int i = 0; i < nf; i++) dithmask |= rd.get(1) << i;
        int twice = !st.acmod;
        do {
            if (rd.get(1)) {
                const int code = rd.sget(8);
                if (st.dynrnge) {
                    const float range = (float)(((code & 0x1f) | 0x20) << 13) * sf_of(3 - (code >> 5));
                    st.dynrng = st.level * range;
                }
            }
        } while (twice--);

        if (rd.get(1)) {                                            // cplstre
            st.chincpl = 0;
            if (rd.get(1)) {                                        // cplinu
                for (int i = 0; i < nf; i++) st.chincpl |= rd.get(1) << i;
                if (st.acmod < 2) { err = 1; break; }
                if (st.acmod == 2) st.phsflginu = rd.get(1);
                const int begf = rd.get(4), endf = rd.get(4);
                if (endf + 3 - begf < 0) { err = 1; break; }
                const int nsub = endf + 3 - begf;
                st.ncplbnd = nsub;
                st.cplstrtbnd = k_cpl_bnd0[begf];
                st.cplstrtmant = begf * 12 + 37;
                st.cplendmant = endf * 12 + 73;
                st.cplbndstrc = 0;
                for (int i = 0; i < nsub - 1; i++)
                    if (rd.get(1)) { st.cplbndstrc |= 1u << i; st.ncplbnd--; }
            }
        } else if (blk == 0) reuse0 = 1;
        if (st.chincpl) {                                           // coupling coordinates
            int any = 0;
            for (int i = 0; i < nf; i++)
                if ((st.chincpl >> i) & 1) {
                    if (rd.get(1)) {
                        const int master = 3 * rd.get(2);
                        any = 1;
                        for (int j = 0; j < st.ncplbnd; j++) {
                            const int ex = rd.get(4);
                            int ma = rd.get(4);
                            ma = (ex == 15) ? (ma << 14) : ((ma | 0x10) << 13);
                            const float co = (float)ma * sf_of(ex + master);
                            if (lane == 0) B.cplco[i][j] = co;
                        }
                    } else if (blk == 0) reuse0 = 1;
                }
            wave_sync();
            if (st.acmod == 2 && st.phsflginu && any)
                for (int j = 0; j < st.ncplbnd; j++)
                    if (rd.get(1) && lane == 0) B.cplco[1][j] = -B.cplco[1][j];
        }
        if (st.acmod == 2) {
            if (rd.get(1)) {                                        // rematstr
                const int end = st.chincpl ? st.cplstrtmant : 253;
                int i = 0;
                st.rematflg = 0;
                do st.rematflg |= rd.get(1) << i; while (k_remat_edge[1 + i++] < end);
            } else if (blk == 0) reuse0 = 1;
        }
        int cplexpstr = 0, lfeexpstr = 0, chexp = 0;
        if (st.chincpl) cplexpstr = rd.get(2);
        for (int i = 0; i < nf; i++) chexp |= rd.get(2) << (2 * i);
        if (st.lfeon) lfeexpstr = rd.get(1);
        if (blk == 0) {
            if (st.chincpl && !cplexpstr) reuse0 = 1;
            if (st.lfeon && !lfeexpstr) reuse0 = 1;
            for (int i = 0; i < nf; i++) if (!((chexp >> (2 * i)) & 3)) reuse0 = 1;
        }
#pragma unroll
        for (int i = 0; i < 5; i++)
            if (i < nf && !err && ((chexp >> (2 * i)) & 3)) {
                if ((st.chincpl >> i) & 1) st.endmant[i] = st.cplstrtmant;
                else {
                    const int bw = rd.get(6);
                    if (bw > 60) err = 1;
                    else st.endmant[i] = bw * 3 + 73;
                }
            }
        if (err) break;
        // exponent fields: recorded, not read (the channel waves decode them side by side)
        int redo = 0;
        if (cplexpstr) {
            const int ngrp = (st.cplendmant - st.cplstrtmant) / (3 << (cplexpstr - 1));
            const int e0 = rd.get(4) << 1;
            redo = 64;
            if (lane == 0) { B.absexp[6] = e0; B.exp_pos[6] = (int)rd.pos; B.ngrp[6] = ngrp; }
            rd.pos += 7 * ngrp;
        }
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int es = (chexp >> (2 * i)) & 3;
            if (i < nf && es) {
                const int gs = 3 << (es - 1), ngrp = (st.endmant[i] + gs - 4) / gs;
                redo |= 1 << i;
                const int e0 = rd.get(4);
                if (lane == 0) { B.absexp[i] = e0; B.exp_pos[i] = (int)rd.pos; B.ngrp[i] = ngrp; }
                rd.pos += 7 * ngrp;
                rd.get(2);                                          // gainrng
            }
        }
        if (lfeexpstr) {
            redo |= 32;
            const int e0 = rd.get(4);
            if (lane == 0) { B.absexp[5] = e0; B.exp_pos[5] = (int)rd.pos; B.ngrp[5] = 2; }
            rd.pos += 14;
        }
        pc.cplexpstr = cplexpstr;
        pc.lfeexpstr = lfeexpstr;
        pc.chexp = chexp;
        pc.redo = redo;
    } while (0);
    if (lane == 0) {
        B.err = err;
        B.blkswm = blkswm;
        B.dithmask = dithmask;
#pragma unroll
        for (int i = 0; i < 5; i++) B.expstr[i] = (pc.chexp >> (2 * i)) & 3;
        B.expstr[5] = pc.lfeexpstr;
        B.expstr[6] = pc.cplexpstr;
    }
    // sub-band -> band of the coupling channel (parse.c:448-456): one lane per sub-band
    if (lane < 18) {
        const uint32_t below = st.cplbndstrc & ((1u << lane) - 1u);
        B.cplbnd[lane] = (uint8_t)(lane - __popc(below));
    }
}

// ---- second half: bit-allocation parameters, delta bit allocation, skip field (parse.c:738-772, 800-804) ----
__device__ void parse_block_b(Rd &rd, St &st, BlkInfo &B, WgLDS &L, ParseCarry &pc, int blk, uint32_t &reuse0, int lane)
{
    const int nf = st.nf;
    int err = ldsu(B.err), redo = pc.redo;
    if (!err) do {
        if (rd.get(1)) { redo = 127; st.bai = rd.get(11); }
        else if (blk == 0) reuse0 = 1;
        if (rd.get(1)) {
            redo = 127;
            st.csnroffst = rd.get(6);
            if (st.chincpl) st.cbai[6] = rd.get(7);
#pragma unroll
            for (int i = 0; i < 5; i++) if (i < nf) st.cbai[i] = rd.get(7);
            if (st.lfeon) st.cbai[5] = rd.get(7);
        } else if (blk == 0) reuse0 = 1;
        if (st.chincpl) {
            if (rd.get(1)) {
                redo |= 64;
                st.cplfleak = 9 - rd.get(3);
                st.cplsleak = 9 - rd.get(3);
            } else if (blk == 0) reuse0 = 1;
        }
        if (rd.get(1)) {                                            // deltbaie
            redo = 127;
            if (st.chincpl) st.deltbae[5] = rd.get(2);
#pragma unroll
            for (int i = 0; i < 5; i++) if (i < nf) st.deltbae[i] = rd.get(2);
#pragma unroll
            for (int pass = 0; pass < 6; pass++) {
                const int slot = pass == 0 ? 5 : pass - 1;          // cpl first, then fbw (parse.c:763-771)
                if (err || pass > nf) continue;
                if (slot == 5 && !st.chincpl) continue;
                if (st.deltbae[slot] != 1) continue;
                if (lane < 50) L.deltba[slot][lane] = 0;            // parse_deltba: parse.c:272-294
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
        if (rd.get(1)) {                                            // skip field
            const int n = rd.get(9);
            rd.pos += 8 * n;
        }
    } while (0);
    bool allzero = !st.csnroffst && !(st.chincpl && (st.cbai[6] >> 3)) && !(st.lfeon && (st.cbai[5] >> 3));
#pragma unroll
    for (int i = 0; i < 5; i++)
        if (i < nf && (st.cbai[i] >> 3)) allzero = false;
    float gain[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (!err) a52_downmix_coeff_hd(gain, st.acmod, st.output, st.dynrng, st.clev, st.slev);     // parse.c:810-811
    if (lane == 0) {
        B.err = err;
        B.halfrate = st.halfrate;
        B.mant_pos = (int)rd.pos;
        B.chincpl = st.chincpl;
        B.cplstrtmant = st.cplstrtmant;
        B.cplendmant = st.cplendmant;
        B.cplstrtbnd = st.cplstrtbnd;
        B.rematflg = st.acmod == 2 ? st.rematflg : 0;
        B.remat_end = st.endmant[0] < st.endmant[1] ? st.endmant[0] : st.endmant[1];
        B.need_fix = (st.chincpl || (st.acmod == 2 && st.rematflg)) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 5; i++) { B.endmant[i] = st.endmant[i]; B.gain[i] = gain[i]; }
        B.redo = redo;
        B.allzero = allzero ? 1 : 0;
        B.bai = st.bai;
        B.csnroffst = st.csnroffst;
#pragma unroll
        for (int i = 0; i < 7; i++) B.cbai[i] = st.cbai[i];
#pragma unroll
        for (int i = 0; i < 6; i++) B.deltbae[i] = st.deltbae[i];
        B.cplfleak = st.cplfleak;
        B.cplsleak = st.cplsleak;
        B.lfe_gain = (st.output & AC3MI_LFE) ? st.dynrng : 0.f;
    }
}

// ---- T1 of one slot: exponents, bit allocation, census --------------------------------------------------------------
__device__ void slot_t1(WgLDS &L, const BlkInfo &B, const FrameBits FB, int slot, int wave, int lane)
{
    const int halfrate = ldsu(B.halfrate);
    const int es = ldsu(B.expstr[slot]);
    uint8_t *erow = L.exp + row_off(slot);
    int8_t *brow = L.bap + row_off(slot);
    int start = 0, end;
    if (slot < 5) end = ldsu(B.endmant[slot]);
    else if (slot == 5) end = 7;
    else { start = ldsu(B.cplstrtmant); end = ldsu(B.cplendmant); }
    if (es) {
        const int e0 = ldsu(B.absexp[slot]), ngrp = ldsu(B.ngrp[slot]);
        const uint32_t pos = (uint32_t)ldsu(B.exp_pos[slot]);
        int bad;
        if (slot == 6) bad = read_exponents(FB, pos, es, ngrp, e0, erow + start, lane);
        else {
            if (lane == 0) erow[0] = (uint8_t)e0;
            bad = read_exponents(FB, pos, es, ngrp, e0, erow + 1, lane);
        }
        if (bad && lane == 0) atomicOr(&L.exp_err, 1);
        wave_sync();
    }
    const int redo = (ldsu(B.redo) >> slot) & 1;
    if (!redo) return;
    if (ldsu(B.allzero)) {
        for (int i = lane; i < (slot == 5 ? LFE_ROW : ROW); i += 64) brow[i] = 0;
    } else if (end > start) {
        const int bai = ldsu(B.bai), mybai = ldsu(B.cbai[slot]);
        const int mydeltbae = slot == 5 ? 2 : ldsu(B.deltbae[slot == 6 ? 5 : slot]);
        BaCtx c;
        c.halfrate = halfrate;
        c.fdecay = (63 + 20 * ((bai >> 7) & 3)) >> halfrate;
        c.fgain = 128 + 128 * (mybai & 7);
        c.sdecay = (15 + 2 * (bai >> 9)) >> halfrate;
        c.sgain = k_slowgain[(bai >> 5) & 3];
        c.dbknee = k_dbpb[(bai >> 3) & 3];
        c.hth = L.hth;
        c.deltba = (mydeltbae == 2) ? nullptr : L.deltba[slot == 6 ? 5 : slot];
        const int fl = k_floors[bai & 7];
        c.snroffset = 960 - 64 * ldsu(B.csnroffst) - 4 * (mybai >> 3) + fl;
        c.floor = fl >> 5;
        c.fast = slot == 6 ? ldsu(B.cplfleak) << 8 : 0;
        c.slow = slot == 6 ? ldsu(B.cplsleak) << 8 : 0;
        bit_allocate_wave(L, L.bmask[wave], c, slot == 6 ? ldsu(B.cplstrtbnd) : 0, start, end, erow, brow, lane);
    }
    wave_sync();
}

// census of a slot's mantissas (also run when nothing was re-allocated: the coupling range may have moved)
__device__ void slot_census(WgLDS &L, int slot, int start, int end, int lane)
{
    const int8_t *brow = L.bap + row_off(slot);
    uint32_t a = 0, b = 0;
    if (slot != 5 || lane < LFE_ROW / 4) {
        const uint32_t bap4 = *reinterpret_cast<const uint32_t *>(brow + 4 * lane);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int bin = 4 * lane + j;
            const int w = (int)(int8_t)(bap4 >> (8 * j));
            if (bin >= start && bin < end) {
                a += w > 0 ? (uint32_t)w : w == -1 ? 1u << 16 : w == -2 ? 1u << 24 : 0u;
                b += w == -3 ? 1u : w == 0 ? 1u << 8 : 0u;
            }
        }
    }
    a = wave_sum_u32(a);
    b = wave_sum_u32(b);
    if (lane == 0) { L.cnt[slot][0] = a; L.cnt[slot][1] = b; }
}

// where a slot's mantissas start: prefix over the segments of the block in bitstream order (parse.c:813-879: channel 0,
// the coupling channel right after the first coupled channel, ..., LFE last).  One lane per segment.
struct SegBase {
    uint32_t bit;           // first bit of the segment
    int r3, r5, r11;        // 3/5/11-level mantissas of the block before the segment (global ranks)
    int draw;               // dither draws of the block before the segment
    int mult;               // draws per zero-bit bin of the segment
    uint32_t total_bits;    // of the block
    int total_draws;
};

__device__ __forceinline__ int openers(int phase, int n, int per)     // members r in [phase, phase + n) with r % per == 0
{
    return (phase + n + per - 1) / per - (phase + per - 1) / per;
}

__device__ SegBase segment_prefix(const WgLDS &L, const BlkInfo &B, int slot, int nf, bool lfeon, int lane)
{
    const int chincpl = ldsu(B.chincpl), dithmask = ldsu(B.dithmask);
    const int cplfirst = chincpl ? __builtin_ctz(chincpl) : 99;
    // lane i = i-th segment in bitstream order
    int sl = -1;
    if (chincpl) {
        if (lane <= cplfirst) sl = lane < nf ? lane : -1;
        else if (lane == cplfirst + 1) sl = 6;
        else sl = lane - 1 < nf ? lane - 1 : (lfeon && lane - 1 == nf) ? 5 : -1;
    } else sl = lane < nf ? lane : (lfeon && lane == nf) ? 5 : -1;
    uint32_t a = 0, b = 0;
    int mult = 0;
    if (sl >= 0) {
        a = L.cnt[sl][0];
        b = L.cnt[sl][1];
        mult = sl < 5 ? (dithmask >> sl) & 1 : sl == 6 ? __popc(chincpl & dithmask) : 0;
    }
    // (a block holds up to 1272 mantissas of one kind: 16-bit fields)
    const int n3 = (int)((a >> 16) & 0xffu), n5 = (int)((a >> 24) & 0xffu), n11 = (int)(b & 0xffu);
    const uint32_t g = (uint32_t)n3 | ((uint32_t)n5 << 16);
    const uint32_t gex = wave_incl_scan_u32(g) - g;
    const int e3 = (int)(gex & 0xffffu), e5 = (int)(gex >> 16);
    const int e11 = (int)wave_incl_scan_u32((uint32_t)n11) - n11;
    const int bits = (int)(a & 0xffffu) + 5 * openers(e3 % 3, n3, 3) + 7 * openers(e5 % 3, n5, 3) + 7 * openers(e11 & 1, n11, 2);
    const int draws = (int)((b >> 8) & 0xffu) * mult;
    const uint32_t t = (uint32_t)bits | ((uint32_t)draws << 16);
    const uint32_t tin = wave_incl_scan_u32(t), tex = tin - t;
    // my segment's position in the order
    int pos = slot;
    if (chincpl) pos = slot == 6 ? cplfirst + 1 : slot == 5 ? nf + 1 : slot > cplfirst ? slot + 1 : slot;
    else if (slot == 5) pos = nf;
    SegBase r;
    r.bit = (uint32_t)ldsu(B.mant_pos) + ((uint32_t)__builtin_amdgcn_readlane((int)tex, pos) & 0xffffu);
    r.draw = (int)((uint32_t)__builtin_amdgcn_readlane((int)tex, pos) >> 16);
    r.r3 = __builtin_amdgcn_readlane(e3, pos);
    r.r5 = __builtin_amdgcn_readlane(e5, pos);
    r.r11 = __builtin_amdgcn_readlane(e11, pos);
    r.mult = __builtin_amdgcn_readlane(mult, pos);
    const uint32_t tot = wave_last(tin);
    r.total_bits = tot & 0xffffu;
    r.total_draws = (int)(tot >> 16);
    return r;
}

// registers a channel wave keeps from T2a to T2b for its 4 bins
struct BinRegs {
    uint32_t raw[4];
    uint32_t bap4, exp4;
    int gm[4];              // group | member << 16 | kind << 20 (kind 3 = not a grouped code)
    int cd;                 // draw index of the lane's first zero-bit bin
};

__device__ __forceinline__ void slot_t2a(WgLDS &L, const FrameBits FB, const uint32_t *frw, int slot, int start, int end, const SegBase &sb,
                                         BinRegs &R, int lane)
{
    const uint8_t *erow = L.exp + row_off(slot);
    const int8_t *brow = L.bap + row_off(slot);
    const bool have = slot != 5 || lane < LFE_ROW / 4;
    R.bap4 = have ? *reinterpret_cast<const uint32_t *>(brow + 4 * lane) : 0u;
    R.exp4 = have ? *reinterpret_cast<const uint32_t *>(erow + 4 * lane) : 0u;
    int w[4], kind[4];
    uint32_t gl = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int bin = 4 * lane + j;
        const bool act = bin >= start && bin < end;
        w[j] = act ? (int)(int8_t)(R.bap4 >> (8 * j)) : -9;
        kind[j] = w[j] == -1 ? 0 : w[j] == -2 ? 1 : w[j] == -3 ? 2 : 3;
        gl += kind[j] < 3 ? 1u << (10 * kind[j]) : 0u;
    }
    const uint32_t gin = wave_incl_scan_u32(gl), gex = gin - gl;
    int c3 = sb.r3 + (int)(gex & 0x3ffu), c5 = sb.r5 + (int)((gex >> 10) & 0x3ffu), c11 = sb.r11 + (int)(gex >> 20);
    int nb[4];
    uint32_t nbsum = 0, ndsum = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int rank = 0;
        if (kind[j] == 0) rank = c3++;
        else if (kind[j] == 1) rank = c5++;
        else if (kind[j] == 2) rank = c11++;
        const int per = kind[j] == 2 ? 2 : 3;
        const int grp = kind[j] == 2 ? rank >> 1 : (int)(((uint32_t)rank * 0xaaabu) >> 17);       // rank / 3, rank < 2^15
        const int mem = rank - grp * per;
        const bool opens = kind[j] < 3 && mem == 0;
        nb[j] = w[j] > 0 ? w[j] : opens ? (kind[j] == 0 ? 5 : 7) : 0;
        R.gm[j] = grp | (mem << 16) | (kind[j] << 20);
        nbsum += (uint32_t)nb[j];
        ndsum += w[j] == 0 ? (uint32_t)sb.mult : 0u;
    }
    const uint32_t bl = nbsum | (ndsum << 16);
    const uint32_t bin_ = wave_incl_scan_u32(bl);
    const uint32_t off = sb.bit + (bin_ & 0xffffu) - nbsum;
    R.cd = sb.draw + (int)(bin_ >> 16) - (int)ndsum;
    // the lane's fields sit in at most 79 consecutive bits: four dwords of the frame
    uint32_t wi = off >> 5;
    wi = wi < FB.last ? wi : FB.last;
    const uint32_t d0 = frw[wi], d1 = frw[wi + 1], d2 = frw[wi + 2], d3 = frw[wi + 3];
    uint32_t r = off & 31u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t sel = r >> 5;
        const uint32_t hi = sel == 0 ? d0 : sel == 1 ? d1 : d2, lo = sel == 0 ? d1 : sel == 1 ? d2 : d3;
        const uint64_t v = (((uint64_t)hi << 32) | lo) << (r & 31u);
        R.raw[j] = nb[j] ? (uint32_t)(v >> 32) >> (32 - nb[j]) : 0u;
        const int k = (R.gm[j] >> 20) & 3;
        if (k < 3 && ((R.gm[j] >> 16) & 15) == 0) L.gcode[k][R.gm[j] & 0xffff] = (uint8_t)R.raw[j];
        r += (uint32_t)nb[j];
    }
}

// dequantised value of bin j (before the exponent / gain scale); coupling-channel and zero-bit handling is the caller's
__device__ __forceinline__ float bin_q(const WgLDS &L, const BinRegs &R, int j, int w)
{
    const int k = (R.gm[j] >> 20) & 3, mem = (R.gm[j] >> 16) & 15, grp = R.gm[j] & 0xffff;
    const bool coded = k < 3 || w == 3 || w == 4;
    const int code = k < 3 ? (int)L.gcode[k][grp] : (int)R.raw[j];
    const int per = k == 2 ? 2 : 3;
    const int base = k == 0 ? 0 : k == 1 ? 96 : k == 2 ? 480 : w == 3 ? 736 : 744;
    const int ti = base + code * (k < 3 ? per : 1) + (k < 3 ? mem : 0);
    const float tv = L.qtab[coded ? ti : 0];
    const float pv = (float)((((int32_t)(R.raw[j] << ((32 - w) & 31))) >> ((32 - w) & 31)) * (1 << ((16 - w) & 31)));
    return coded ? tv : w > 0 ? pv : 0.f;
}

}  // namespace wg

using namespace wg;

namespace wg {

#pragma clang fp contract(fast)
// ---- the transformer's three parts for one block (xform_core.h arithmetic, as xform.hip) ----------------------------

// part A: coefficient plane (LDS) -> registers, first half of the transform.  Identity routing: output o = input plane o.
__device__ __forceinline__ void xform_part_a(WgLDS &L, int o, int sw, int l8, float2 *ex, cf (&r)[16])
{
    const float *plane = L.planes + o * PLANE;
    float xa[16], xb[16];
    if (!sw) {
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int m = 8 * n + l8;
            xa[n] = plane[2 * m];
            xb[n] = plane[255 - 2 * m];
        }
        imdct_first_half(xa, xb, L.twl + l8 * 16, ex, l8, r);
    } else {
        const int f = l8 >> 2, n2 = l8 & 3;
#pragma unroll
        for (int n = 0; n < 16; n++) {
            xa[n] = plane[16 * n + 4 * n2 + f];
            xb[n] = plane[254 + f - 16 * n - 4 * n2];
        }
        imdct_first_half(xa, xb, L.tws + l8 * 16, ex, l8, r);
    }
}

template <int OUT>
__device__ __forceinline__ void xform_part_c(WgLDS &L, const WgParams &W, const FirstTail &ft, int o, int l8, int lane, bool store,
                                             size_t blk_index /* fidx * 6 + blk */)
{
    const int n_out = W.n_out;
    float *dl = L.dly[o];
    int16_t *tile = reinterpret_cast<int16_t *>(L.ex);
    int wsl = 0;
#pragma unroll
    for (int oo = 0; oo < 6; oo++) wsl = oo == o ? W.wslot[oo] : wsl;
    float *oblk = OUT == 1 ? W.pcm + (blk_index * n_out + o) * 256 : nullptr;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int i = ((j & 1) ? 15 - l8 : l8) + 16 * (j >> 1);
        const float2 wlo = *reinterpret_cast<const float2 *>(&L.win[2 * i]);
        const float2 whi = *reinterpret_cast<const float2 *>(&L.win[254 - 2 * i]);
        const float2 d = *reinterpret_cast<const float2 *>(&dl[2 * i]);
        float2 lo, hi;
        lo.x = ft.f0[j] * wlo.x + (d.x * whi.y + W.bias);
        lo.y = ft.f1[j] * wlo.y + (d.y * whi.x + W.bias);
        hi.x = d.y * wlo.y + W.bias - ft.f1[j] * whi.x;
        hi.y = d.x * wlo.x + W.bias - ft.f0[j] * whi.y;
        if (store) {
            if (OUT == 1) {
                *reinterpret_cast<float2 *>(oblk + 2 * i) = lo;
                *reinterpret_cast<float2 *>(oblk + 254 - 2 * i) = hi;
            } else {
                tile[wsl + (2 * i) * n_out] = to_s16(lo.x);
                tile[wsl + (2 * i + 1) * n_out] = to_s16(lo.y);
                tile[wsl + (254 - 2 * i) * n_out] = to_s16(hi.x);
                tile[wsl + (255 - 2 * i) * n_out] = to_s16(hi.y);
            }
            *reinterpret_cast<float2 *>(&dl[2 * i]) = make_float2(ft.t0[j], ft.t1[j]);
        }
    }
    if (OUT == 2) {
        wave_sync();
        const int upb = 32 * n_out;                         // 16-byte units of one block of the stream
        int16_t *dst = W.pcm16 + blk_index * (size_t)n_out * 256;
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int u = it * 64 + lane;
            if (u < upb) *reinterpret_cast<uint4 *>(dst + (size_t)u * 8) = *reinterpret_cast<const uint4 *>(tile + u * 8);
        }
        wave_sync();
    }
}

#pragma clang fp contract(off)

}  // namespace wg

// OUT 0: coefficient planes (and block-switch flags) to HBM, for stage taps and the unfused paths
// OUT 1: float PCM planes          OUT 2: interleaved s16 PCM
template <int OUT>
__global__ __launch_bounds__(512, WG_LB) void decode_wg_kernel(const WgParams W)
{
    __shared__ WgLDS L;
    extern __shared__ uint32_t frw[];
    const DecodeParams &P = W.d;
    const int tid = threadIdx.x, lane = tid & 63, l8 = tid & 7;
    const int wave = (int)rfl((uint32_t)(tid >> 6));
    const FrameBits FB{frw, (uint32_t)((P.frame_bytes + 3) >> 2) + 2u};
    const int in_lfe = P.lfeon ? 1 : 0;
    const int nf = P.nfchans;
    const int n_out = W.n_out;

    // ---- constant tables ----
    for (int i = tid; i < 256; i += 512) { L.la_neg[i] = P.tab->la_neg[i]; L.band_of_bin[i] = P.tab->band_of_bin[i]; L.win[i] = W.window[i]; }
    if (tid < 64) L.width[tid] = P.tab->width[tid];
    if (tid < 30) L.band_end[tid] = P.tab->band_end[tid];
    if (tid < 50) L.hth[tid] = 0;
    for (int i = tid; i < 760; i += 512) L.qtab[i] = P.tab->qtab[i];
    if (tid < 128) { L.twl[tid] = W.tw_long[tid]; L.tws[tid] = W.tw_short[tid]; }
    int hth_fscod = -1;

    // transformer: 8-lane group g = output plane g (groups past n_out shadow plane 0 and store nothing)
    const int xgroup = lane >> 3;
    const bool xstore = xgroup < n_out;
    const int xo = xstore ? xgroup : 0;
    float2 *xex = L.ex + xo * EX_GROUP;

    for (int s = blockIdx.x; s < P.n_streams; s += gridDim.x) {
        // ---- per-stream state ----
        wg_barrier();                                                // the previous stream's tails have been written back
        for (int i = tid; i < ROWS; i += 512) { L.exp[i] = 0; L.bap[i] = 0; }
        for (int i = tid; i < 6 * 52; i += 512) (&L.deltba[0][0])[i] = 0;
        for (int i = tid; i < 90; i += 512) { (&L.bi[0].cplco[0][0])[i] = 0.f; (&L.bi[1].cplco[0][0])[i] = 0.f; }
        const int sslot = P.slot ? P.slot[s] : s;
        if (OUT != 0) {
            const float *dsrc = W.delay + (size_t)sslot * W.delay_stride;
            for (int i = tid; i < n_out * 128; i += 512) (&L.dly[0][0])[i] = dsrc[i];
        }
        St st;
        st.fscod = st.halfrate = st.acmod = st.lfeon = 0;
        st.nf = 0;
        st.clev = st.slev = st.level = st.dynrng = 0.f;
        st.output = 0;
        st.dynrnge = 1;
        st.chincpl = st.phsflginu = st.cplstrtmant = st.cplendmant = st.ncplbnd = st.cplstrtbnd = 0;
        st.cplbndstrc = 0;
        st.rematflg = 0;
        for (int i = 0; i < 5; i++) st.endmant[i] = 0;
        st.bai = st.csnroffst = 0;
        for (int i = 0; i < 7; i++) st.cbai[i] = 0;
        for (int i = 0; i < 6; i++) st.deltbae[i] = 2;
        st.cplfleak = st.cplsleak = 0;
        st.lfsr = 0;
        uint32_t lfsr_idx = 0;                                       // position in the dither generator's cycle
        bool lfsr_live = false;
        if (wave == W_PARSE) {
            st.lfsr = (uint32_t)P.lfsr_state[sslot];
            lfsr_live = st.lfsr != 0;
            lfsr_idx = (uint32_t)P.lfsr_idx[st.lfsr];
        }

        for (int f = 0; f < P.frames_per_stream; f++) {
            const size_t fidx = (size_t)s * P.frames_per_stream + f;
            // ---- stage the frame: byte-swapped dwords, zero padded ----
            {
                const uint32_t *s32 = reinterpret_cast<const uint32_t *>(P.frames + fidx * P.frame_stride);
                const int nw = (P.frame_bytes + 3) >> 2;
                for (int i = tid; i < nw + 6; i += 512) {
                    uint32_t v = 0;
                    if (i < nw) {
                        v = s32[i];
                        const int rem = P.frame_bytes - 4 * i;
                        if (rem < 4) v &= (1u << (8 * rem)) - 1u;
                        v = __builtin_bswap32(v);
                    }
                    frw[i] = v;
                }
                if (tid == 0) L.exp_err = 0;
            }
            wg_barrier();

            // ---- frame header + block 0 (parser alone) ----
            uint32_t status = 0, reuse0 = 0;
            bool frame_dead = false;
            Rd rd{FB, 0, ~0u, 0, 0};
            ParseCarry pc;
            pc.cplexpstr = pc.lfeexpstr = pc.chexp = pc.redo = 0;
            if (wave == W_PARSE) {
                bool hdr_ok = true;                                  // a52_syncinfo (parse.c:86-129) + a52_frame (parse.c:131-205)
                {
                    const uint32_t w0 = rfl(frw[0]), w1 = rfl(frw[1]);
                    const int b4 = (w1 >> 24) & 0xff, b5 = (w1 >> 16) & 0xff, b6 = (w1 >> 8) & 0xff;
                    if ((w0 >> 16) != 0x0b77) hdr_ok = false;
                    if (b5 >= 0x60) hdr_ok = false;
                    if ((b4 & 63) >= 38 || (b4 & 0xc0) == 0xc0) hdr_ok = false;
                    if (hdr_ok) {
                        st.fscod = b4 >> 6;
                        const int bsid = b5 >> 3;
                        st.halfrate = bsid < 9 ? 0 : bsid - 8;
                        st.acmod = b6 >> 5;
                        if (st.acmod != P.acmod) hdr_ok = false;
                        const int code = b4 & 63, rate = k_kbps[code >> 1];
                        const int fbytes = st.fscod == 0 ? 4 * rate : st.fscod == 1 ? 2 * (320 * rate / 147 + (code & 1)) : 6 * rate;
                        if (fbytes > P.frame_bytes) hdr_ok = false;
                    }
                }
                if (hdr_ok) {
                    int acmod = st.acmod;
                    rd.pos = 6 * 8 + 3;
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
                        st.dynrnge = P.dynrng_on;
                        for (int i = 0; i < 6; i++) st.deltbae[i] = 2;
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
                        if (hth_fscod != st.fscod) {
                            if (lane < 50) L.hth[lane] = P.tab->hth[st.fscod][lane];
                            hth_fscod = st.fscod;
                        }
                        status |= (uint32_t)st.output << 16;
                    }
                }
                if (!hdr_ok) { status |= 0x100u; frame_dead = true; }
                BlkInfo &B0 = L.bi[0];
                if (!frame_dead) {
                    parse_block_a(rd, st, B0, L.bi[1], pc, 0, reuse0, lane);
                    wave_sync();
                    parse_block_b(rd, st, B0, L, pc, 0, reuse0, lane);
                } else if (lane == 0) B0.err = 1;
                if (lane == 0) {
                    B0.lfsr_i0 = lfsr_idx;
                    B0.lfsr_live = lfsr_live ? 1 : 0;
                }
                wave_sync();
                if (lane == 0) L.blkswm_hist[0] = frame_dead ? 0 : B0.blkswm;
            }

            // ---- six blocks ----
            cf xr[16];                                               // transformer: block blk-1 between its parts
            FirstTail xft;
            int xsw = 0;
            for (int blk = 0; blk < 6; blk++) {
                const BlkInfo &B = L.bi[blk & 1];
                BlkInfo &Bn = L.bi[(blk + 1) & 1];
                wg_barrier();                                        // ---- B1: B is valid, the planes hold block blk-1 ----
                const int err_side = ldsu(B.err);
                const int chincpl = err_side ? 0 : ldsu(B.chincpl);
                const int cplstrt = ldsu(B.cplstrtmant), cplend = ldsu(B.cplendmant);
                const int my_end = wave < 5 ? ldsu(B.endmant[wave < 5 ? wave : 0]) : 7;
                if (wave <= W_LFE) {
                    if (!err_side) {
                        if (wave < nf) {
                            slot_t1(L, B, FB, wave, wave, lane);
                            slot_census(L, wave, 0, my_end, lane);
                        } else if (wave == W_LFE) {
                            if (chincpl) {
                                slot_t1(L, B, FB, 6, wave, lane);
                                slot_census(L, 6, cplstrt, cplend, lane);
                            }
                            if (P.lfeon) {
                                slot_t1(L, B, FB, 5, wave, lane);
                                slot_census(L, 5, 0, 7, lane);
                            }
                        }
                    }
                    if (OUT == 0 && P.tap_exp) {                     // optional stage taps: this wave's rows
                        uint8_t *te = P.tap_exp + (fidx * 6 + blk) * 7 * 256;
                        int8_t *tb = P.tap_bap + (fidx * 6 + blk) * 7 * 256;
                        for (int i = lane; i < 256; i += 64) {       // (the LFE row is short)
                            const bool in = wave != 5 || i < LFE_ROW;
                            te[wave * 256 + i] = in ? L.exp[row_off(wave) + i] : 0;
                            tb[wave * 256 + i] = in ? L.bap[row_off(wave) + i] : 0;
                        }
                        if (wave == W_LFE)
                            for (int i = lane; i < 256; i += 64) {
                                te[6 * 256 + i] = L.exp[row_off(6) + i];
                                tb[6 * 256 + i] = L.bap[row_off(6) + i];
                            }
                    }
                } else if (wave == W_XFORM && blk > 0) {
                    if (OUT == 0) {
                        // planes of block blk-1 and its block-switch flags to HBM
                        float *cblk = P.coef + (fidx * 6 + (blk - 1)) * (size_t)P.n_in * 256;
                        for (int c = 0; c < P.n_in; c++)
                            *reinterpret_cast<float4 *>(cblk + c * 256 + 4 * lane) = *reinterpret_cast<const float4 *>(L.planes + c * PLANE + 4 * lane);
                        if (P.blksw && lane < nf)
                            P.blksw[(fidx * 6 + (blk - 1)) * nf + lane] = (uint8_t)((L.blkswm_hist[blk - 1] >> lane) & 1);
                    } else {
                        const int fb = xo - in_lfe;
                        xsw = fb >= 0 ? (L.blkswm_hist[blk - 1] >> fb) & 1 : 0;
                        xform_part_a(L, xo, xsw, l8, xex, xr);
                    }
                }
                wg_barrier();                                        // ---- B2: censuses and exponent verdicts are in; planes are free ----
                const int err = err_side | ldsu(L.exp_err);
                BinRegs R, R2;
                SegBase sb, sb2;
                if (wave <= W_LFE) {
                    if (!err) {
                        if (wave < nf) {
                            sb = segment_prefix(L, B, wave, nf, P.lfeon != 0, lane);
                            slot_t2a(L, FB, frw, wave, 0, my_end, sb, R, lane);
                        } else if (wave == W_LFE) {
                            if (chincpl) {
                                sb = segment_prefix(L, B, 6, nf, P.lfeon != 0, lane);
                                slot_t2a(L, FB, frw, 6, cplstrt, cplend, sb, R, lane);
                            }
                            if (P.lfeon) {
                                sb2 = segment_prefix(L, B, 5, nf, P.lfeon != 0, lane);
                                slot_t2a(L, FB, frw, 5, 0, 7, sb2, R2, lane);
                            }
                        }
                    }
                } else if (wave == W_PARSE) {
                    if (err) {
                        status |= 1u << blk;
                        frame_dead = true;
                        if (lane == 0) L.blkswm_hist[blk] = 0;
                    } else {
                        sb = segment_prefix(L, B, 0, nf, P.lfeon != 0, lane);
                        if (lfsr_live && sb.total_draws) lfsr_idx = (lfsr_idx + (uint32_t)sb.total_draws) % 65535u;
                        rd.pos = (uint32_t)ldsu(B.mant_pos) + sb.total_bits;
                    }
                    if (blk < 5) {
                        if (!frame_dead) parse_block_a(rd, st, Bn, B, pc, blk + 1, reuse0, lane);
                        else if (lane == 0) Bn.err = 1;
                    }
                } else if (OUT != 0 && blk > 0) {
                    xft = FirstTail{};
                    if (!xsw) imdct_long_second_half(xr, xft);
                    else imdct_short_second_half(xr, xft);
                }
                wg_barrier();                                        // ---- B3: the open codes are published ----
                const int need_fix = err ? 0 : ldsu(B.need_fix);
                if (wave <= W_LFE) {
                    const uint32_t i0 = (uint32_t)ldsu((const int &)B.lfsr_i0);
                    const bool live = ldsu(B.lfsr_live) != 0;
                    const int dithmask = ldsu(B.dithmask);
                    if (wave < nf) {
                        float out[4] = {0.f, 0.f, 0.f, 0.f};
                        if (!err) {
                            const float g = ldsf(B.gain[wave < 5 ? wave : 0]);
                            const int dith = (dithmask >> wave) & 1;
                            int cd = R.cd;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int bin = 4 * lane + j;
                                const bool act = bin < my_end;
                                const int w = act ? (int)(int8_t)(R.bap4 >> (8 * j)) : -9;
                                const int e = (int)((R.exp4 >> (8 * j)) & 0xffu);
                                float q = bin_q(L, R, j, w);
                                if (w == 0 && dith) { q = (float)(live ? dither_at(P.lfsr_seq, i0, cd) : 0); cd++; }
                                const float v = q * (sf_of(e) * g);
                                out[j] = act ? v : 0.f;
                            }
                        }
                        *reinterpret_cast<float4 *>(L.planes + (wave + in_lfe) * PLANE + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
                    } else if (wave == W_LFE) {
                        if (!err && chincpl) {                       // coupling channel: mantissa * 2^-exp and the draw index of zero-bit bins
                            int cd = R.cd;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int bin = 4 * lane + j;
                                const bool act = bin >= cplstrt && bin < cplend;
                                const int w = act ? (int)(int8_t)(R.bap4 >> (8 * j)) : -9;
                                const int e = (int)((R.exp4 >> (8 * j)) & 0xffu);
                                const float q = bin_q(L, R, j, w);
                                L.cplq[bin] = q * sf_of(e);
                                L.cplcd[bin] = (int16_t)cd;
                                if (w == 0) cd += sb.mult;
                            }
                        }
                        if (P.lfeon) {
                            float out[4] = {0.f, 0.f, 0.f, 0.f};
                            if (!err) {
                                const float g = ldsf(B.lfe_gain);
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const int bin = 4 * lane + j;
                                    const bool act = bin < 7;
                                    const int w = act ? (int)(int8_t)(R2.bap4 >> (8 * j)) : -9;
                                    const int e = (int)((R2.exp4 >> (8 * j)) & 0xffu);
                                    const float v = bin_q(L, R2, j, w) * (sf_of(e) * g);
                                    out[j] = act ? v : 0.f;
                                }
                            }
                            *reinterpret_cast<float4 *>(L.planes + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
                        }
                    }
                } else if (wave == W_PARSE) {
                    if (blk < 5) {
                        if (!frame_dead) parse_block_b(rd, st, Bn, L, pc, blk + 1, reuse0, lane);
                        if (lane == 0) {
                            Bn.lfsr_i0 = lfsr_idx;
                            Bn.lfsr_live = lfsr_live ? 1 : 0;
                        }
                        wave_sync();
                        if (lane == 0) L.blkswm_hist[blk + 1] = frame_dead ? 0 : Bn.blkswm;
                    }
                } else if (OUT != 0 && blk > 0) {
                    xform_part_c<OUT>(L, W, xft, xo, l8, lane, xstore, fidx * 6 + (blk - 1));
                }
                if (need_fix) {
                    wg_barrier();                                    // ---- B4: every plane and the coupling channel are in LDS ----
                    if (wave < nf && ((chincpl >> wave) & 1)) {
                        const uint32_t i0 = (uint32_t)ldsu((const int &)B.lfsr_i0);
                        const bool live = ldsu(B.lfsr_live) != 0;
                        const int dithmask = ldsu(B.dithmask);
                        const float gc = ldsf(B.gain[wave < 5 ? wave : 0]);
                        const int dith = (dithmask >> wave) & 1;
                        const int kc = __popc(chincpl & dithmask & ((1 << wave) - 1));
                        const int8_t *bapc = L.bap + row_off(6);
                        const uint8_t *expc = L.exp + row_off(6);
                        float *plane = L.planes + (wave + in_lfe) * PLANE;
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int bin = 4 * lane + j;
                            if (bin >= cplstrt && bin < cplend) {
                                const int bnd = B.cplbnd[(bin - cplstrt) / 12];
                                const float co = B.cplco[wave < 5 ? wave : 0][bnd] * gc;
                                const int w = bapc[bin], e = expc[bin];
                                float v = L.cplq[bin] * co;
                                if (w == 0) {
                                    v = 0.f;
                                    if (dith) v = (sf_of(e) * co) * (float)(live ? dither_at(P.lfsr_seq, i0, (int)L.cplcd[bin] + kc) : 0);
                                }
                                plane[bin] = v;
                            }
                        }
                    }
                    const int rematflg = ldsu(B.rematflg);
                    if (rematflg) {                                  // (a coupled channel's share lies above the rematrixed bins)
                        if (wave == 0) {                             // rematrix: parse.c:837-865
                            const int rend = ldsu(B.remat_end);
                            float *p0 = L.planes + in_lfe * PLANE, *p1 = p0 + PLANE;
                            for (int bin = 13 + lane; bin < rend; bin += 64) {
                                const int band = bin < 25 ? 0 : bin < 37 ? 1 : bin < 61 ? 2 : 3;
                                if ((rematflg >> band) & 1) {
                                    const float a = p0[bin], v = p1[bin];
                                    p0[bin] = a + v;
                                    p1[bin] = a - v;
                                }
                            }
                        }
                    }
                }
            }
            // ---- the transformer owes block 5 ----
            wg_barrier();
            if (wave == W_XFORM) {
                if (OUT == 0) {
                    float *cblk = P.coef + (fidx * 6 + 5) * (size_t)P.n_in * 256;
                    for (int c = 0; c < P.n_in; c++)
                        *reinterpret_cast<float4 *>(cblk + c * 256 + 4 * lane) = *reinterpret_cast<const float4 *>(L.planes + c * PLANE + 4 * lane);
                    if (P.blksw && lane < nf) P.blksw[(fidx * 6 + 5) * nf + lane] = (uint8_t)((L.blkswm_hist[5] >> lane) & 1);
                } else {
                    const int fb = xo - in_lfe;
                    xsw = fb >= 0 ? (L.blkswm_hist[5] >> fb) & 1 : 0;
                    xform_part_a(L, xo, xsw, l8, xex, xr);
                    xft = FirstTail{};
                    if (!xsw) imdct_long_second_half(xr, xft);
                    else imdct_short_second_half(xr, xft);
                    xform_part_c<OUT>(L, W, xft, xo, l8, lane, xstore, fidx * 6 + 5);
                }
            }
            if (wave == W_PARSE && lane == 0) P.status[fidx] = status | ((status & 0x100u) ? 0x3fu : 0u) | (reuse0 ? 0x200u : 0u);
            wg_barrier();                                            // the frame buffer and the planes are free
        }
        // ---- carry-over state of the stream ----
        if (wave == W_PARSE && lane == 0) P.lfsr_state[sslot] = lfsr_live ? P.lfsr_seq[lfsr_idx] : (uint16_t)0;
        if (OUT != 0) {
            float *ddst = W.delay + (size_t)sslot * W.delay_stride;
            for (int i = tid; i < n_out * 128; i += 512) ddst[i] = (&L.dly[0][0])[i];
        }
    }
}

hipError_t launch_decode_wg(const DeviceTables &tab, const DecodeLaunch &D, const XformLaunch *X, int grid_cap, hipStream_t stream)
{
    WgParams W;
    DecodeParams &P = W.d;
    P.frames = D.frames;
    P.coef = D.coef;
    P.blksw = D.blksw;
    P.status = D.status;
    P.lfsr_state = D.lfsr;
    P.slot = D.slot;
    P.tap_exp = D.tap_exp;
    P.tap_bap = D.tap_bap;
    P.lfsr_seq = tab.lfsr_seq;
    P.lfsr_idx = tab.lfsr_idx;
    P.tab = tab.dec;
    P.n_streams = D.n_streams;
    P.frames_per_stream = D.frames_per_stream;
    P.frame_stride = D.frame_stride;
    P.frame_bytes = D.frame_bytes;
    P.req_flags = D.req_flags;
    P.level = D.level;
    P.dynrng_on = D.dynrng_on;
    P.acmod = D.acmod;
    P.lfeon = D.lfeon;
    static const int nfch[8] = {2, 1, 2, 3, 3, 4, 4, 5};
    P.nfchans = nfch[D.acmod & 7];
    P.n_in = P.nfchans + (D.lfeon ? 1 : 0);
    P.frame_draws = nullptr;
    P.frame_lfsr = nullptr;
    W.tw_long = tab.tw_long;
    W.tw_short = tab.tw_short;
    W.window = tab.window;
    W.delay = nullptr;
    W.delay_stride = 0;
    W.pcm = nullptr;
    W.pcm16 = nullptr;
    W.bias = 0.f;
    W.n_out = P.n_in;
    for (int i = 0; i < 6; i++) W.wslot[i] = (int8_t)i;
    if (D.n_streams <= 0 || D.frames_per_stream <= 0) return hipSuccess;
    const size_t fr_bytes = (size_t)(((D.frame_bytes + 3) >> 2) + 6) * 4;
    // persistent grid: as many workgroups as the chip holds at once (each walks streams blockIdx.x, + gridDim.x, ...)
    if (grid_cap <= 0) {
        int dev = 0, cus = 256, occ = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const void *fn = !X ? (const void *)decode_wg_kernel<0> : X->pcm16 ? (const void *)decode_wg_kernel<2> : (const void *)decode_wg_kernel<1>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 512, fr_bytes) != hipSuccess || occ < 1) occ = 1;
        grid_cap = occ * cus;
    }
    int grid = D.n_streams < grid_cap ? D.n_streams : grid_cap;
    if (!X) {
        hipLaunchKernelGGL(decode_wg_kernel<0>, dim3(grid), dim3(512), fr_bytes, stream, W);
        return hipGetLastError();
    }
    // fused: identity routing only (every coded plane is an output plane)
    if (X->plan.n_in != X->plan.n_out || X->plan.n_in != P.n_in) return hipErrorInvalidValue;
    for (int o = 0; o < X->plan.n_out; o++)
        for (int c = 0; c < X->plan.n_in; c++)
            if (X->plan.mix[o][c] != (o == c ? 1 : 0)) return hipErrorInvalidValue;
    W.delay = X->delay;
    W.delay_stride = X->slot ? X->delay_stride : X->plan.n_out * 128;
    W.bias = X->bias;
    W.n_out = X->plan.n_out;
    if (X->pcm16) {
        int map[6];
        if (s16_channel_map(X->s16_flags, map) != W.n_out || ((uintptr_t)X->pcm16 & 15)) return hipErrorInvalidValue;
        for (int w = 0; w < W.n_out; w++) W.wslot[map[w]] = (int8_t)w;
        W.pcm16 = X->pcm16;
        hipLaunchKernelGGL(decode_wg_kernel<2>, dim3(grid), dim3(512), fr_bytes, stream, W);
    } else {
        W.pcm = X->pcm;
        hipLaunchKernelGGL(decode_wg_kernel<1>, dim3(grid), dim3(512), fr_bytes, stream, W);
    }
    return hipGetLastError();
}

}  // namespace ac3mi
