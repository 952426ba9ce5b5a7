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
// xform_core.h's with its fused multiply-adds written out: the same bits as xform.hip produces from the same planes.
#include "decode_common.h"
#include <stdio.h>
#include <stdlib.h>
#include "xform_core.h"

namespace ac3mi {

namespace wg {

#ifndef WG_LB
#define WG_LB 4                             // waves per SIMD the register budget is set for (512-thread blocks: 2 per block):
                                            // 128 VGPRs, two workgroups per CU, practically no spills.  6 (80 VGPRs, three per CU)
                                            // spills: 0.199 / 0.326 / 8.9 ms per 1 024 / 2 048 / 65 536 one-frame streams against
                                            // 0.186 / 0.358 / 10.7 ms, and its scratch traffic is 4 x the frame's own bytes
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
    uint8_t gcode[3 * GC + 4];              // open codes by kind and group; the last bytes are a sink for lanes that open nothing
    uint32_t desc[128];                     // per row byte: dequantiser base | members per code << 10 | opener bits << 12 | coded << 15
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

// Measurement aid (make EXTRA=-DWG_STAMPS, a separate library): wave w of workgroup 0 records s_memtime at fixed points of
// the first frame of its SECOND stream (warm caches) into a buffer nothing else reads; dumped by launch_decode_wg.
#ifdef WG_STAMPS
#define STAMP(id) do { if (W.stamps && blockIdx.x == 0 && s == (int)gridDim.x && f == 0 && lane == 0) W.stamps[wave * 64 + (id)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(id) do { } while (0)
#endif

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
// An opaque copy of a lane-dependent value: address arithmetic derived from it stays where it is used instead of being
// hoisted out of the (long) block loop and spilled.
__device__ __forceinline__ int opaque(int v) { asm volatile("" : "+v"(v)); return v; }
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
    unsigned long long *stamps; // WG_STAMPS builds only: [8 waves][64] s_memtime values of one workgroup's second stream
};

// The parser's bit reader: the 64 frame words from the current position on sit one per lane in a VGPR; a field is two
// v_readlane (scalar index) and a funnel shift - no LDS round trip per field.  The window is reloaded when the position
// leaves it (after the exponent payloads are skipped).  Wave-uniform like Rd.
struct VRd {
    const uint32_t *w;
    uint32_t last;          // as FrameBits::last (words up to last + 1 exist)
    uint32_t pos, base;
    uint32_t vwin;
    int lane;
    __device__ __forceinline__ void load()
    {
        base = pos >> 5;
        uint32_t i = base + (uint32_t)lane;
        i = i < last + 1u ? i : last + 1u;
        vwin = w[i];
    }
    __device__ __forceinline__ uint32_t get(int n)
    {
        if (n == 0) return 0;
        uint32_t wi = (pos >> 5) - base;
        if (wi >= 63u) { load(); wi = 0; }
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)vwin, (int)wi);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)vwin, (int)wi + 1);
        const uint32_t v = (uint32_t)(((((uint64_t)hi << 32) | lo) << (pos & 31u)) >> (64 - n));
        pos += (uint32_t)n;
        return v;
    }
    __device__ __forceinline__ int32_t sget(int n)
    {
        const uint32_t v = get(n);
        return ((int32_t)(v << (32 - n))) >> (32 - n);
    }
};
__device__ __forceinline__ VRd make_reader(const FrameBits FB, uint32_t pos, int lane)
{
    VRd rd{FB.w, FB.last, pos, 0, 0, lane};
    rd.load();
    return rd;
}

// The parser's view of the stream (a52_state_t's side-information fields, L52/a52_internal.h:35-88) lives in LDS, one copy
// per workgroup, touched by the parser wavefront only: every lane reads the same word (ldsu), lane 0 writes.  Nothing of
// it occupies registers across the workgroup barriers.
struct ParserState {
    int fscod, halfrate, acmod, lfeon, nf;
    float clev, slev, level, dynrng;
    int output, dynrnge;
    int chincpl, phsflginu, cplstrtmant, cplendmant, ncplbnd, cplstrtbnd;
    uint32_t cplbndstrc;
    int rematflg;
    int endmant[5];
    int bai, csnroffst, cbai[7], deltbae[6], cplfleak, cplsleak;
    uint32_t pos;               // the bit reader's position
    int cplexpstr, lfeexpstr, chexp, redo;      // from the first to the second half of a block's side information
    uint32_t status, reuse0;
    int frame_dead;
    uint32_t lfsr_idx;          // position in the dither generator's cycle
    int lfsr_live;
    int hth_fscod;
};

#define PSET(field, value) do { const auto pset_v_ = (value); if (lane == 0) (field) = pset_v_; } while (0)

// ---- side information, first half: L52/parse.c:572-736 without the exponent payloads (their positions are recorded) ----
#ifdef WG_STAMPS
#define PSTAMP(i) do { if (dbg && lane == 0) dbg[i] = __builtin_readcyclecounter(); } while (0)
#else
#define PSTAMP(i) do { } while (0)
#endif
__device__ __forceinline__ void parse_block_a(const FrameBits FB, ParserState &S, BlkInfo &B, const BlkInfo &prev, int blk, int lane, const DecodeParams &P, size_t fidx,
                              unsigned long long *dbg = nullptr)
{
    PSTAMP(0);
    VRd rd = make_reader(FB, (uint32_t)ldsu((const int &)S.pos), lane);
    const int nf = ldsu(S.nf), acmod = ldsu(S.acmod), lfeon = ldsu(S.lfeon);
    int err = 0, blkswm = 0, dithmask = 0, reuse0 = 0;
    int cplexpstr = 0, lfeexpstr = 0, chexp = 0, redo = 0;
    // coupling coordinates persist from block to block (and frame to frame)
    for (int i = lane; i < 90; i += 64) (&B.cplco[0][0])[i] = (&prev.cplco[0][0])[i];
    PSTAMP(1);
    do {
        for (int i = 0; i < nf; i++) blkswm |= rd.get(1) << i;
        for (int i = 0; i < nf; i++) dithmask |= rd.get(1) << i;
        int twice = !acmod, word = 0;
        do {
            if (rd.get(1)) {
                const int code = rd.sget(8);
                if (ldsu(S.dynrnge)) PSET(S.dynrng, ldsf(S.level) * dynrng_range(P, code, (fidx * 6 + blk) * 2 + word, lane));
            }
            word++;
        } while (twice--);

        PSTAMP(2);
        int chincpl = ldsu(S.chincpl);
        if (rd.get(1)) {                                            // cplstre
            chincpl = 0;
            if (rd.get(1)) {                                        // cplinu
                for (int i = 0; i < nf; i++) chincpl |= rd.get(1) << i;
                PSET(S.chincpl, chincpl);
                if (acmod < 2) { err = 1; break; }
                if (acmod == 2) PSET(S.phsflginu, (int)rd.get(1));
                const int begf = rd.get(4), endf = rd.get(4);
                if (endf + 3 - begf < 0) { err = 1; break; }
                const int nsub = endf + 3 - begf;
                int ncplbnd = nsub;
                uint32_t strc = 0;
                for (int i = 0; i < nsub - 1; i++)
                    if (rd.get(1)) { strc |= 1u << i; ncplbnd--; }
                PSET(S.ncplbnd, ncplbnd);
                PSET(S.cplstrtbnd, (int)k_cpl_bnd0[begf]);
                PSET(S.cplstrtmant, begf * 12 + 37);
                PSET(S.cplendmant, endf * 12 + 73);
                PSET(S.cplbndstrc, strc);
            } else PSET(S.chincpl, 0);
        } else if (blk == 0) reuse0 = 1;
        if (chincpl) {                                              // coupling coordinates
            const int ncplbnd = ldsu(S.ncplbnd);
            int any = 0;
            for (int i = 0; i < nf; i++)
                if ((chincpl >> i) & 1) {
                    if (rd.get(1)) {
                        const int master = 3 * rd.get(2);
                        any = 1;
                        for (int j = 0; j < ncplbnd; j++) {
                            const int ex = rd.get(4);
                            int ma = rd.get(4);
                            ma = (ex == 15) ? (ma << 14) : ((ma | 0x10) << 13);
                            const float co = (float)ma * sf_of(ex + master);
                            if (lane == 0) B.cplco[i][j] = co;
                        }
                    } else if (blk == 0) reuse0 = 1;
                }
            if (acmod == 2 && ldsu(S.phsflginu) && any)
                for (int j = 0; j < ncplbnd; j++)
                    if (rd.get(1) && lane == 0) B.cplco[1][j] = -B.cplco[1][j];
        }
        if (acmod == 2) {
            if (rd.get(1)) {                                        // rematstr
                const int end = chincpl ? ldsu(S.cplstrtmant) : 253;
                int i = 0, flg = 0;
                do flg |= rd.get(1) << i; while (k_remat_edge[1 + i++] < end);
                PSET(S.rematflg, flg);
            } else if (blk == 0) reuse0 = 1;
        }
        PSTAMP(3);
        if (chincpl) cplexpstr = rd.get(2);
        for (int i = 0; i < nf; i++) chexp |= rd.get(2) << (2 * i);
        if (lfeon) lfeexpstr = rd.get(1);
        if (blk == 0) {
            if (chincpl && !cplexpstr) reuse0 = 1;
            if (lfeon && !lfeexpstr) reuse0 = 1;
            for (int i = 0; i < nf; i++) if (!((chexp >> (2 * i)) & 3)) reuse0 = 1;
        }
        const int cplstrtmant = ldsu(S.cplstrtmant), cplendmant = ldsu(S.cplendmant);
        for (int i = 0; i < nf; i++)
            if (!err && ((chexp >> (2 * i)) & 3)) {
                if ((chincpl >> i) & 1) PSET(S.endmant[i], cplstrtmant);
                else {
                    const int bw = rd.get(6);
                    if (bw > 60) err = 1;
                    else PSET(S.endmant[i], bw * 3 + 73);
                }
            }
        PSTAMP(4);
        if (err) break;
        // exponent fields: recorded, not read (the channel waves decode them side by side)
        if (cplexpstr) {
            const int ngrp = (cplendmant - cplstrtmant) / (3 << (cplexpstr - 1));
            const int e0 = rd.get(4) << 1;
            redo = 64;
            if (lane == 0) { B.absexp[6] = e0; B.exp_pos[6] = (int)rd.pos; B.ngrp[6] = ngrp; }
            rd.pos += 7 * ngrp;
        }
        for (int i = 0; i < nf; i++) {
            const int es = (chexp >> (2 * i)) & 3;
            if (es) {
                const int gs = 3 << (es - 1), ngrp = (ldsu(S.endmant[i]) + gs - 4) / gs;
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
    } while (0);
    PSTAMP(5);
    if (lane == 0) {
        B.err = err;
        B.blkswm = blkswm;
        B.dithmask = dithmask;
        for (int i = 0; i < 5; i++) B.expstr[i] = (chexp >> (2 * i)) & 3;
        B.expstr[5] = lfeexpstr;
        B.expstr[6] = cplexpstr;
        S.pos = rd.pos;
        S.cplexpstr = cplexpstr;
        S.lfeexpstr = lfeexpstr;
        S.chexp = chexp;
        S.redo = redo;
        if (reuse0) S.reuse0 = 1;
    }
    // sub-band -> band of the coupling channel (parse.c:448-456): one lane per sub-band
    if (lane < 18) {
        const uint32_t below = (uint32_t)ldsu((const int &)S.cplbndstrc) & ((1u << lane) - 1u);
        B.cplbnd[lane] = (uint8_t)(lane - __popc(below));
    }
    PSTAMP(6);
}

// ---- second half: bit-allocation parameters, delta bit allocation, skip field (parse.c:738-772, 800-804) ----
__device__ __forceinline__ void parse_block_b(const FrameBits FB, ParserState &S, BlkInfo &B, int8_t (*deltba)[52], int blk, int lane)
{
    VRd rd = make_reader(FB, (uint32_t)ldsu((const int &)S.pos), lane);
    const int nf = ldsu(S.nf), lfeon = ldsu(S.lfeon), chincpl = ldsu(S.chincpl), acmod = ldsu(S.acmod);
    int err = ldsu(B.err), redo = ldsu(S.redo), reuse0 = 0;
    if (!err) do {
        if (rd.get(1)) { redo = 127; PSET(S.bai, (int)rd.get(11)); }
        else if (blk == 0) reuse0 = 1;
        if (rd.get(1)) {
            redo = 127;
            PSET(S.csnroffst, (int)rd.get(6));
            if (chincpl) PSET(S.cbai[6], (int)rd.get(7));
            for (int i = 0; i < nf; i++) PSET(S.cbai[i], (int)rd.get(7));
            if (lfeon) PSET(S.cbai[5], (int)rd.get(7));
        } else if (blk == 0) reuse0 = 1;
        if (chincpl) {
            if (rd.get(1)) {
                redo |= 64;
                PSET(S.cplfleak, 9 - (int)rd.get(3));
                PSET(S.cplsleak, 9 - (int)rd.get(3));
            } else if (blk == 0) reuse0 = 1;
        }
        if (rd.get(1)) {                                            // deltbaie
            redo = 127;
            if (chincpl) PSET(S.deltbae[5], (int)rd.get(2));
            for (int i = 0; i < nf; i++) PSET(S.deltbae[i], (int)rd.get(2));
            for (int pass = 0; pass <= nf && !err; pass++) {
                const int slot = pass == 0 ? 5 : pass - 1;          // cpl first, then fbw (parse.c:763-771)
                if (slot == 5 && !chincpl) continue;
                if (ldsu(S.deltbae[slot]) != 1) continue;
                if (lane < 50) deltba[slot][lane] = 0;              // parse_deltba: parse.c:272-294
                int nseg = rd.get(3), band = 0;
                do {
                    band += rd.get(5);
                    int len = rd.get(4), d = rd.get(3);
                    d -= (d >= 4) ? 3 : 4;
                    if (!len) continue;
                    if (band + len >= 50) { err = 1; break; }
                    if (lane < len) deltba[slot][band + lane] = (int8_t)d;
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
    wave_sync();
    // publish what the channel waves and the transformer need (one field per lane where it is a plain copy)
    const int csnroffst = ldsu(S.csnroffst);
    bool allzero = !csnroffst && !(chincpl && (ldsu(S.cbai[6]) >> 3)) && !(lfeon && (ldsu(S.cbai[5]) >> 3));
    for (int i = 0; i < nf; i++)
        if (ldsu(S.cbai[i]) >> 3) allzero = false;
    float gain[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const int output = ldsu(S.output);
    const float dynrng = ldsf(S.dynrng);
    if (!err) a52_downmix_coeff_hd(gain, acmod, output, dynrng, ldsf(S.clev), ldsf(S.slev));      // parse.c:810-811
    if (lane < 5) { B.endmant[lane] = S.endmant[lane]; }
    if (lane < 7) B.cbai[lane] = S.cbai[lane];
    if (lane < 6) B.deltbae[lane] = S.deltbae[lane];
    if (lane == 0) {
        const int e0 = S.endmant[0], e1 = S.endmant[1], rematflg = acmod == 2 ? S.rematflg : 0;
        B.err = err;
        B.halfrate = S.halfrate;
        B.mant_pos = (int)rd.pos;
        B.chincpl = chincpl;
        B.cplstrtmant = S.cplstrtmant;
        B.cplendmant = S.cplendmant;
        B.cplstrtbnd = S.cplstrtbnd;
        B.rematflg = rematflg;
        B.remat_end = e0 < e1 ? e0 : e1;
        B.need_fix = (chincpl || rematflg) ? 1 : 0;
        B.gain[0] = gain[0]; B.gain[1] = gain[1]; B.gain[2] = gain[2]; B.gain[3] = gain[3]; B.gain[4] = gain[4];
        B.redo = redo;
        B.allzero = allzero ? 1 : 0;
        B.bai = S.bai;
        B.csnroffst = csnroffst;
        B.cplfleak = S.cplfleak;
        B.cplsleak = S.cplsleak;
        B.lfe_gain = (output & AC3MI_LFE) ? dynrng : 0.f;
        S.pos = rd.pos;
        if (reuse0) S.reuse0 = 1;
    }
}

// ---- a52_syncinfo (parse.c:86-129) + a52_frame (parse.c:131-205): frame header and BSI; returns false for a frame the
// batch cannot hold (no sync word, other channel configuration, too long, an output liba52 would refuse) ----
__device__ __forceinline__ bool parse_frame_header(const FrameBits FB, const uint32_t *frw, ParserState &S, uint16_t *hth, const DecodeParams &P, int lane)
{
    const uint32_t w0 = rfl(frw[0]), w1 = rfl(frw[1]);
    const int b4 = (w1 >> 24) & 0xff, b5 = (w1 >> 16) & 0xff, b6 = (w1 >> 8) & 0xff;
    if ((w0 >> 16) != 0x0b77) return false;
    if (b5 >= 0x60) return false;
    if ((b4 & 63) >= 38 || (b4 & 0xc0) == 0xc0) return false;
    const int fscod = b4 >> 6, bsid = b5 >> 3, acmod0 = b6 >> 5;
    PSET(S.fscod, fscod);
    PSET(S.halfrate, bsid < 9 ? 0 : bsid - 8);
    PSET(S.acmod, acmod0);
    if (acmod0 != P.acmod) return false;
    {
        const int code = b4 & 63, rate = k_kbps[code >> 1];
        const int fbytes = fscod == 0 ? 4 * rate : fscod == 1 ? 2 * (320 * rate / 147 + (code & 1)) : 6 * rate;
        if (fbytes > P.frame_bytes) return false;                // frame_bytes = the largest frame of the batch (44.1 kHz alternates)
    }
    VRd rd = make_reader(FB, 6 * 8 + 3, lane);
    int acmod = acmod0;
    if (acmod == 2 && rd.get(2) == 2) acmod = 10;                 // dsurmod -> DOLBY
    float clev = 0.f, slev = 0.f;
    if ((acmod & 1) && acmod != 1) clev = k_clev[rd.get(2)];
    if (acmod & 4) slev = k_slev[rd.get(2)];
    PSET(S.clev, clev);
    PSET(S.slev, slev);
    const int lfeon = rd.get(1);
    PSET(S.lfeon, lfeon);
    if (lfeon != P.lfeon) return false;
    float level = P.level;
    int output = a52_downmix_init_hd(acmod, P.req_flags, &level, clev, slev);
    if (output < 0) return false;
    if (lfeon && (P.req_flags & AC3MI_LFE)) output |= AC3MI_LFE;
    PSET(S.output, output);
    PSET(S.level, level * 2);
    PSET(S.dynrng, level * 2);
    PSET(S.dynrnge, P.dynrng_on);
    if (lane < 6) S.deltbae[lane] = 2;
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
    PSET(S.nf, (int)k_nfchans[acmod0]);
    if (ldsu(S.hth_fscod) != fscod) {
        if (lane < 50) hth[lane] = P.tab->hth[fscod][lane];
        PSET(S.hth_fscod, fscod);
    }
    PSET(S.pos, rd.pos);
    PSET(S.status, (uint32_t)output << 16);
    return true;
}



// ---- T1 of one slot: exponents, bit allocation, census --------------------------------------------------------------
__device__ __forceinline__ void slot_t1(WgLDS &L, const BlkInfo &B, const FrameBits FB, int slot, int wave, int lane)
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

// Row bytes (L.bap): 0 = no bits, 3..16 = that many plain bits, 32 / 64 / 96 = member of a 3- / 5- / 11-level code
// (ba_width's -1 / -2 / -3, remapped when the width table is staged), so that  plain bits = b & 31,  kind = b >> 5.
// Everything below is branch-free: four bins per lane, the per-code constants come from L.desc.

// census of a slot's mantissas (also run when nothing was re-allocated: the coupling range may have moved)
__device__ __forceinline__ void slot_census(WgLDS &L, int slot, int start, int end, int lane)
{
    const int8_t *brow = L.bap + row_off(slot);
    const bool have = slot != 5 || lane < LFE_ROW / 4;
    const uint32_t bap4 = have ? *reinterpret_cast<const uint32_t *>(brow + 4 * lane) : 0u;
    uint32_t a = 0, z = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int bin = 4 * lane + j;
        const uint32_t act = (uint32_t)(bin >= start) & (uint32_t)(bin < end);
        const uint32_t b = ((bap4 >> (8 * j)) & 0xffu) * act;
        const uint32_t k1 = b >> 5;
        a += (b & 31u) | (((1u << (8 * k1)) >> 8) << 16);         // plain bits | n3 << 16 | n5 << 24 | (n11: carried in z)
        z += (k1 == 3u ? 1u : 0u) | ((act & (uint32_t)(b == 0u)) << 8);
    }
    // (1 << 8 k1) >> 8 puts an 11-level member at bit 32: dropped above, counted in z instead
    a = wave_sum_u32(a);
    z = wave_sum_u32(z);
    if (lane == 0) { L.cnt[slot][0] = a; L.cnt[slot][1] = z; }
}

// where a slot's mantissas start: prefix over the segments of the block in bitstream order (parse.c:813-879: channel 0,
// the coupling channel right after the first coupled channel, ..., LFE last).  One lane per segment.
__device__ __forceinline__ int openers(int phase, int n, int per)     // members r in [phase, phase + n) with r % per == 0
{
    return (phase + n + per - 1) / per - (phase + per - 1) / per;
}

__device__ __forceinline__ SegBase segment_prefix(const WgLDS &L, const BlkInfo &B, int slot, int nf, bool lfeon, int lane)
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

__device__ __forceinline__ void slot_t2a(WgLDS &L, const FrameBits FB, const uint32_t *frw, int slot, int start, int end, const SegBase &sb,
                                         BinRegs &R, int lane)
{
    (void)mant_first_half<GC, 0xfff, false>(L.exp + row_off(slot), L.bap + row_off(slot), L.desc, L.gcode, frw, FB.last, start, end,
                                            slot == 5 ? LFE_ROW / 4 : 64, sb, R, lane);
}
__device__ __forceinline__ float bin_q(const WgLDS &L, const BinRegs &R, int j) { return mant_value<GC, 0xfff>(R, j, L.desc, L.gcode, L.qtab); }

// ---- the transformer: one block of all output planes per pass, 8 lanes per plane (xform_core.h arithmetic, as xform.hip).
// Nothing stays in registers between its three parts: the transposed points and the first/tail values wait in the
// transformer's own LDS scratch (L.ex), so the parts can sit in different phases of the workgroup's schedule.

// part A: coefficient plane (LDS) -> registers, pre-twiddle, DFT-16, lane twiddles, rows of the 8x16 transpose
__device__ __forceinline__ void xform_part_a(const WgLDS &L, int o, int sw, int l8, float2 *ex)
{
    const float *plane = L.planes + o * PLANE;
    float xa[16], xb[16];
    if (!sw) {
#pragma unroll
        for (int n = 0; n < 16; n++) {
            const int m = 8 * n + l8;
            xa[n] = __builtin_fmaf(1.f, plane[2 * m], 0.f);         // as xform.hip's load_long with one plane of weight +1
            xb[n] = __builtin_fmaf(1.f, plane[255 - 2 * m], 0.f);
        }
        imdct_first_half_put(xa, xb, L.twl + l8 * 16, ex, l8);
    } else {
        const int f = l8 >> 2, n2 = l8 & 3;
#pragma unroll
        for (int n = 0; n < 16; n++) {
            xa[n] = __builtin_fmaf(1.f, plane[16 * n + 4 * n2 + f], 0.f);
            xb[n] = __builtin_fmaf(1.f, plane[254 + f - 16 * n - 4 * n2], 0.f);
        }
        imdct_first_half_put(xa, xb, L.tws + l8 * 16, ex, l8);
    }
}

constexpr int FT_PITCH = 36;                // floats per lane of the parked first/tail values (32 + pad)

// part B: columns of the transpose, second half of the transform, first/tail values parked in the scratch
__device__ __forceinline__ void xform_part_b(int sw, int l8, float2 *ex)
{
    cf r[16];
    transpose_8x16_get(ex, l8, r);
    FirstTail ft = FirstTail{};
    if (!sw) imdct_long_second_half(r, ft);
    else imdct_short_second_half(r, ft);
    wave_sync();                                                // every lane of the group has read its columns
    float4 *park = reinterpret_cast<float4 *>(reinterpret_cast<float *>(ex) + l8 * FT_PITCH);
#pragma unroll
    for (int k = 0; k < 2; k++) {
        park[k] = make_float4(ft.f0[4 * k], ft.f0[4 * k + 1], ft.f0[4 * k + 2], ft.f0[4 * k + 3]);
        park[2 + k] = make_float4(ft.f1[4 * k], ft.f1[4 * k + 1], ft.f1[4 * k + 2], ft.f1[4 * k + 3]);
        park[4 + k] = make_float4(ft.t0[4 * k], ft.t0[4 * k + 1], ft.t0[4 * k + 2], ft.t0[4 * k + 3]);
        park[6 + k] = make_float4(ft.t1[4 * k], ft.t1[4 * k + 1], ft.t1[4 * k + 2], ft.t1[4 * k + 3]);
    }
}

// part C: window + overlap-add + bias, new tails, PCM out
template <int OUT>
__device__ __forceinline__ void xform_part_c(WgLDS &L, const WgParams &W, int o, int l8, int lane, bool store, float2 *ex,
                                             size_t blk_index /* fidx * 6 + blk */)
{
    const int n_out = W.n_out;
    l8 = opaque(l8);
    o = opaque(o);
    float v[32];
    {
        const float4 *park = reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(ex) + l8 * FT_PITCH);
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float4 q = park[k];
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    }
    wave_sync();                                                // the scratch becomes the s16 tile of this block
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
        window_pair(v[j], v[8 + j], d, wlo, whi, W.bias, lo, hi);
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
            *reinterpret_cast<float2 *>(&dl[2 * i]) = make_float2(v[16 + j], v[24 + j]);
        }
    }
    if (OUT == 2) {
        wave_sync();
        const int upb = 32 * n_out;                             // 16-byte units of one block of the stream
        int16_t *dst = W.pcm16 + blk_index * (size_t)n_out * 256;
#pragma unroll
        for (int it = 0; it < 3; it++) {
            const int u = it * 64 + lane;
            if (u < upb) *reinterpret_cast<uint4 *>(dst + (size_t)u * 8) = *reinterpret_cast<const uint4 *>(tile + u * 8);
        }
        wave_sync();
    }
}

}  // namespace wg

using namespace wg;

// OUT 0: coefficient planes (and block-switch flags) to HBM, for stage taps and the unfused paths
// OUT 1: float PCM planes          OUT 2: interleaved s16 PCM
template <int OUT>
__global__ __launch_bounds__(512, WG_LB) void decode_wg_kernel(const WgParams W)
{
    __shared__ WgLDS L;
    __shared__ ParserState S;
    extern __shared__ uint32_t frw[];
    const DecodeParams &P = W.d;
    const int tid = threadIdx.x, lane0 = tid & 63, lane = lane0;
    const int wave = (int)rfl((uint32_t)(tid >> 6));
    const FrameBits FB{frw, (uint32_t)((P.frame_bytes + 3) >> 2) + 2u};
    const int in_lfe = P.lfeon ? 1 : 0;
    const int nf = P.nfchans;
    const int n_out = W.n_out;

    // ---- constant tables ----
    for (int i = tid; i < 256; i += 512) { L.la_neg[i] = P.tab->la_neg[i]; L.band_of_bin[i] = P.tab->band_of_bin[i]; L.win[i] = W.window[i]; }
    if (tid < 64) {                                                  // row-byte form of the width table: see slot_census
        L.width[tid] = remap_width(P.tab->width[tid]);
    }
    if (tid < 128) L.desc[tid] = mant_desc((uint32_t)tid);
    if (tid < 30) L.band_end[tid] = P.tab->band_end[tid];
    if (tid < 50) L.hth[tid] = 0;
    for (int i = tid; i < 760; i += 512) L.qtab[i] = P.tab->qtab[i];
    if (tid < 128) { L.twl[tid] = W.tw_long[tid]; L.tws[tid] = W.tw_short[tid]; }
    if (tid == 0) S.hth_fscod = -1;

    // transformer: 8-lane group g = output plane g (groups past n_out shadow plane 0 and store nothing)

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
        if (wave == W_PARSE) {
            const int hf = ldsu(S.hth_fscod);
            for (int i = lane; i < (int)(sizeof(ParserState) / 4); i += 64) reinterpret_cast<uint32_t *>(&S)[i] = 0u;
            wave_sync();
            if (lane < 6) S.deltbae[lane] = 2;
            const uint32_t state = (uint32_t)P.lfsr_state[sslot];
            if (lane == 0) {
                S.dynrnge = 1;
                S.hth_fscod = hf;
                S.lfsr_live = state != 0;
                S.lfsr_idx = (uint32_t)P.lfsr_idx[state];
            }
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
            STAMP(0);
            wg_barrier();
            STAMP(1);

            // ---- frame header + side information of block 0 (parser alone) ----
            if (wave == W_PARSE) {
                PSET(S.status, 0u);
                PSET(S.reuse0, 0u);
                const bool ok = parse_frame_header(FB, frw, S, L.hth, P, lane);
                PSET(S.frame_dead, ok ? 0 : 1);
                if (!ok) PSET(S.status, 0x100u);
                BlkInfo &B0 = L.bi[0];
                if (ok) {
                    parse_block_a(FB, S, B0, L.bi[1], 0, lane, P, fidx);
                    wave_sync();
                    parse_block_b(FB, S, B0, L.deltba, 0, lane);
                } else PSET(B0.err, 1);
                wave_sync();
                if (lane == 0) {
                    B0.lfsr_i0 = S.lfsr_idx;
                    B0.lfsr_live = S.lfsr_live;
                    L.blkswm_hist[0] = ok ? B0.blkswm : 0;
                }
            }

            STAMP(2);
            // ---- six blocks ----
            for (int blk = 0; blk < 6; blk++) {
                const BlkInfo &B = L.bi[blk & 1];
                BlkInfo &Bn = L.bi[(blk + 1) & 1];
                wg_barrier();                                        // ---- B1: B is valid, the planes hold block blk-1 ----
                STAMP(4 + 8 * blk);
                const int lane = opaque(lane0), l8 = lane & 7;
                const int xgroup = lane >> 3;
                const bool xstore = xgroup < n_out;
                const int xo = xstore ? xgroup : 0;
                float2 *xex = L.ex + xo * EX_GROUP;
                const int err_side = ldsu(B.err);
                const int chincpl = err_side ? 0 : ldsu(B.chincpl);
                const int cplstrt = ldsu(B.cplstrtmant), cplend = ldsu(B.cplendmant);
                const int my_end = wave < 5 ? ldsu(B.endmant[wave < 5 ? wave : 0]) : 7;
                int xsw = 0;
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
                            const int bb = in ? L.bap[row_off(wave) + i] : 0;
                            te[wave * 256 + i] = in ? L.exp[row_off(wave) + i] : 0;
                            tb[wave * 256 + i] = unmap_width(bb);          // liba52's form: -1 / -2 / -3 for grouped codes
                        }
                        if (wave == W_LFE)
                            for (int i = lane; i < 256; i += 64) {
                                const int bb = L.bap[row_off(6) + i];
                                te[6 * 256 + i] = L.exp[row_off(6) + i];
                                tb[6 * 256 + i] = unmap_width(bb);
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
                        xform_part_a(L, xo, xsw, l8, xex);
                    }
                }
                STAMP(5 + 8 * blk);
                wg_barrier();                                        // ---- B2: censuses and exponent verdicts are in; planes are free ----
                STAMP(6 + 8 * blk);
                const int err = err_side | ldsu(L.exp_err);
                BinRegs R, R2;
                int cpl_mult = 0;
                if (wave <= W_LFE) {
                    if (!err) {
                        if (wave < nf) {
                            const SegBase sb = segment_prefix(L, B, wave, nf, P.lfeon != 0, lane);
                            slot_t2a(L, FB, frw, wave, 0, my_end, sb, R, lane);
                        } else if (wave == W_LFE) {
                            if (chincpl) {
                                const SegBase sb = segment_prefix(L, B, 6, nf, P.lfeon != 0, lane);
                                cpl_mult = sb.mult;
                                slot_t2a(L, FB, frw, 6, cplstrt, cplend, sb, R, lane);
                            }
                            if (P.lfeon) {
                                const SegBase sb2 = segment_prefix(L, B, 5, nf, P.lfeon != 0, lane);
                                slot_t2a(L, FB, frw, 5, 0, 7, sb2, R2, lane);
                            }
                        }
                    }
                } else if (wave == W_PARSE) {
                    int dead = ldsu(S.frame_dead);
                    if (err) {
                        PSET(S.status, (uint32_t)ldsu((const int &)S.status) | (1u << blk));
                        PSET(S.frame_dead, 1);
                        dead = 1;
                        if (lane == 0) L.blkswm_hist[blk] = 0;
                    } else {
                        const SegBase sb = segment_prefix(L, B, 0, nf, P.lfeon != 0, lane);
                        if (ldsu(S.lfsr_live) && sb.total_draws)
                            PSET(S.lfsr_idx, ((uint32_t)ldsu((const int &)S.lfsr_idx) + (uint32_t)sb.total_draws) % 65535u);
                        PSET(S.pos, (uint32_t)ldsu(B.mant_pos) + sb.total_bits);
                    }
                    if (blk < 5) {
#ifdef WG_STAMPS
                        if (!dead) parse_block_a(FB, S, Bn, B, blk + 1, lane, P, fidx, (W.stamps && blockIdx.x == 0 && s == (int)gridDim.x && f == 0 && blk == 2) ? W.stamps + wave * 64 + 56 : nullptr);
#else
                        if (!dead) parse_block_a(FB, S, Bn, B, blk + 1, lane, P, fidx);
#endif
                        else PSET(Bn.err, 1);
                    }
                } else if (OUT != 0 && blk > 0) {
                    const int fb = xo - in_lfe;
                    xsw = fb >= 0 ? (L.blkswm_hist[blk - 1] >> fb) & 1 : 0;
                    xform_part_b(xsw, l8, xex);
                }
                STAMP(7 + 8 * blk);
                wg_barrier();                                        // ---- B3: the open codes are published ----
                STAMP(8 + 8 * blk);
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
                            const bool any_dither = dith && live;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int bin = 4 * lane + j;
                                const bool zero = bin < my_end && ((R.bap4 >> (8 * j)) & 0xffu) == 0u;
                                const int e = (int)((R.exp4 >> (8 * j)) & 0xffu);
                                float q = bin_q(L, R, j);
                                if (any_dither && __any(zero)) {          // wave-uniform: skip the table access when no lane draws
                                    const float dv = (float)dither_at(P.lfsr_seq, i0, cd);
                                    q = zero ? dv : q;
                                }
                                cd += (zero && dith) ? 1 : 0;
                                out[j] = q * (sf_of(e) * g);          // (bins outside the channel have no bits: q = 0)
                            }
                        }
                        *reinterpret_cast<float4 *>(L.planes + (wave + in_lfe) * PLANE + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
                    } else if (wave == W_LFE) {
                        if (!err && chincpl) {                       // coupling channel: mantissa * 2^-exp and the draw index of zero-bit bins
                            int cd = R.cd;
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const int bin = 4 * lane + j;
                                const bool zero = bin >= cplstrt && bin < cplend && ((R.bap4 >> (8 * j)) & 0xffu) == 0u;
                                const int e = (int)((R.exp4 >> (8 * j)) & 0xffu);
                                L.cplq[bin] = bin_q(L, R, j) * sf_of(e);
                                L.cplcd[bin] = (int16_t)cd;
                                cd += zero ? cpl_mult : 0;
                            }
                        }
                        if (P.lfeon) {
                            float out[4] = {0.f, 0.f, 0.f, 0.f};
                            if (!err) {
                                const float g = ldsf(B.lfe_gain);
#pragma unroll
                                for (int j = 0; j < 4; j++) {
                                    const int e = (int)((R2.exp4 >> (8 * j)) & 0xffu);
                                    out[j] = bin_q(L, R2, j) * (sf_of(e) * g);
                                }
                            }
                            *reinterpret_cast<float4 *>(L.planes + 4 * lane) = make_float4(out[0], out[1], out[2], out[3]);
                        }
                    }
                } else if (wave == W_PARSE) {
                    if (blk < 5) {
                        const int dead = ldsu(S.frame_dead);
                        if (!dead) parse_block_b(FB, S, Bn, L.deltba, blk + 1, lane);
                        wave_sync();
                        if (lane == 0) {
                            Bn.lfsr_i0 = S.lfsr_idx;
                            Bn.lfsr_live = S.lfsr_live;
                            L.blkswm_hist[blk + 1] = dead ? 0 : Bn.blkswm;
                        }
                    }
                } else if (OUT != 0 && blk > 0) {
                    xform_part_c<OUT>(L, W, xo, l8, lane, xstore, xex, fidx * 6 + (blk - 1));
                }
                STAMP(9 + 8 * blk);
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
                        // (a damaged frame can leave a coupled channel's own end - the previous block's, when it reuses its
                        // exponents - inside or beyond the coupling range: liba52 decodes the channel's own bins, writes the
                        // coupling channel's share when it meets the FIRST coupled channel, zeroes every coupled channel from
                        // cplendmant on (parse.c:813-834); so a later coupled channel's own bins inside the range stay)
                        const int cplfirst = __builtin_ctz(chincpl), own_end = ldsu(B.endmant[wave < 5 ? wave : 0]);
#pragma unroll
                        for (int j = 0; j < 4; j++) {
                            const int bin = 4 * lane + j;
                            if (bin >= cplend) plane[bin] = 0.f;
                            if (bin >= cplstrt && bin < cplend && (wave == cplfirst || bin >= own_end)) {
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
                    // (only a damaged frame can put rematrixed bins inside the coupling range: then the coupled channels'
                    // waves must have finished above before wave 0 reads their planes)
                    if (rematflg && chincpl && ldsu(B.remat_end) > cplstrt) wg_barrier();
                    if (rematflg && wave == 0) {                     // rematrix: parse.c:837-865 (a coupled channel's share lies above these bins)
                        // (liba52's loop is a do-while, parse.c:846-862: with the first band's flag set it rematrixes bin 13 even
                        // when the channels end at or below it - a damaged frame whose block 0 reuses exponents)
                        const int rend0 = ldsu(B.remat_end), rend = rend0 <= 13 && (rematflg & 1) ? 14 : rend0;
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
            // ---- the transformer owes block 5 ----
            wg_barrier();
            STAMP(52);
            if (wave == W_XFORM) {
                const int lane = opaque(lane0), l8 = lane & 7;
                const int xgroup = lane >> 3;
                const bool xstore = xgroup < n_out;
                const int xo = xstore ? xgroup : 0;
                float2 *xex = L.ex + xo * EX_GROUP;
                if (OUT == 0) {
                    float *cblk = P.coef + (fidx * 6 + 5) * (size_t)P.n_in * 256;
                    for (int c = 0; c < P.n_in; c++)
                        *reinterpret_cast<float4 *>(cblk + c * 256 + 4 * lane) = *reinterpret_cast<const float4 *>(L.planes + c * PLANE + 4 * lane);
                    if (P.blksw && lane < nf) P.blksw[(fidx * 6 + 5) * nf + lane] = (uint8_t)((L.blkswm_hist[5] >> lane) & 1);
                } else {
                    const int fb = xo - in_lfe;
                    const int xsw = fb >= 0 ? (L.blkswm_hist[5] >> fb) & 1 : 0;
                    xform_part_a(L, xo, xsw, l8, xex);
                    xform_part_b(xsw, l8, xex);
                    xform_part_c<OUT>(L, W, xo, l8, lane, xstore, xex, fidx * 6 + 5);
                }
            }
            if (wave == W_PARSE && lane == 0) {
                const uint32_t st = S.status;
                P.status[fidx] = st | ((st & 0x100u) ? 0x3fu : 0u) | (S.reuse0 ? 0x200u : 0u);
                if (P.zs) P.zs[fidx] = (uint8_t)((st & 0x100u) ? 0 : surround_level_is_zero(S.acmod, S.output, S.slev));
            }
            STAMP(53);
            wg_barrier();                                            // the frame buffer and the planes are free
            STAMP(54);
        }
        // ---- carry-over state of the stream ----
        if (wave == W_PARSE && lane == 0) P.lfsr_state[sslot] = S.lfsr_live ? P.lfsr_seq[S.lfsr_idx] : (uint16_t)0;
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
    P.zs = D.zs;
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
    P.dyn_out = D.dyn_out;
    P.dyn_in = D.dyn_in;
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
    W.stamps = nullptr;
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
#ifdef WG_STAMPS
    static unsigned long long *d_stamps = nullptr;
    if (!d_stamps) (void)hipMalloc((void **)&d_stamps, 8 * 64 * sizeof(unsigned long long));
    (void)hipMemsetAsync(d_stamps, 0, 8 * 64 * sizeof(unsigned long long), stream);
    W.stamps = d_stamps;
    struct Dump {
        unsigned long long *d; hipStream_t st;
        ~Dump() {
            unsigned long long h[8 * 64];
            (void)hipStreamSynchronize(st);
            (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
            if (const char *path = getenv("AC3MI_WG_STAMPS")) {
                FILE *fp = fopen(path, "a");
                if (fp) {
                    unsigned long long t0 = ~0ull;
                    for (int i = 0; i < 8 * 64; i++) if (h[i] && h[i] < t0) t0 = h[i];
                    for (int w = 0; w < 8; w++) {
                        fprintf(fp, "wave %d:", w);
                        for (int i = 0; i < 64; i++) fprintf(fp, " %lld", h[w * 64 + i] ? (long long)(h[w * 64 + i] - t0) : -1ll);
                        fprintf(fp, "\n");
                    }
                    fprintf(fp, "\n");
                    fclose(fp);
                }
            }
        }
    } dump{d_stamps, stream};
#endif
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
