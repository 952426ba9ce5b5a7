// mant2.h — the mantissa stage of mant_kernel (decode.hip): one audio block per wavefront from the parse kernel's
// descriptors (L52/parse.c:336-433 coeff_get, :435-556 coupling, :813-879).  The same results, bit for bit, as the stage
// of decode_common.h (mant_first_half / mant_value) that the one-kernel front ends and decode_wg.hip use; written for
// fewer vector instructions per bin:
//  * everything a row byte implies comes from ONE 16-byte LDS read (mant_desc2): rank increment, widths, the divider of
//    the rank -> (code, member) split, table bases, flags - byte- and word-aligned fields the compiler reaches by SDWA
//  * bins outside [start, end) are remapped to a row byte of their own (1: no bits, no draw) once per lane
//  * the rank fields of a segment start at (codes opened so far mod 128) * members + phase, so a member's rank / members
//    IS its slot in the ring where the opener publishes the code - no per-kind base to select
//  * dither draws index a generator table that runs on past its period: no modulo per draw
//  * the LFE's seven bins take a pass of one bin per lane
#pragma once
#include "decode_common.h"

namespace ac3mi {

constexpr int M2_RING = 3 * 128;                // bytes: open 3- / 5- / 11-level codes, 128 slots per kind
constexpr int M2_LDS_WAVE = M2_RING + 256;      // + one sink dword per lane for bins that publish nothing
constexpr int M2_NDESC = 97;
constexpr int LFSR_EXT = 16384;                 // entries of lfsr_seq past the period (a block draws < 5 * 1280 values)

//  x: rank increment: 1 << 20 / 10 / 0 for a member of a 3- / 5- / 11-level code, 0 else
//  y: plain bits | opener bits << 8 | members per code << 16 | rank field shift << 24
//  z: 2048 / members rounded up (683, 1024; 0: ungrouped) | dequantiser table base << 16
//  w: ring base (128 (kind - 1)) | zero-bit bin << 16 | value from the table << 24
__host__ __device__ inline uint4 mant_desc2(uint32_t b)
{
    const uint32_t k1 = b >> 5, nbp = b >= 32 ? 0u : b == 1 ? 0u : b;
    const uint32_t inc = k1 == 1 ? 1u << 20 : k1 == 2 ? 1u << 10 : k1 == 3 ? 1u : 0u;
    const uint32_t obits = k1 == 1 ? 5u : k1 ? 7u : 0u, per = k1 == 3 ? 2u : k1 ? 3u : 0u, sh = 30u - 10u * k1;
    const uint32_t recip = k1 == 3 ? 1024u : k1 ? 683u : 0u;
    const uint32_t qbase = k1 == 1 ? 0u : k1 == 2 ? 96u : k1 == 3 ? 480u : nbp == 3 ? 736u : nbp == 4 ? 744u : 0u;
    const uint32_t coded = (k1 || nbp == 3 || nbp == 4) ? 1u : 0u, zero = b == 0 ? 1u : 0u;
    const uint32_t ring = k1 ? 128u * (k1 - 1u) : 0u;
    return make_uint4(inc, nbp | (obits << 8) | (per << 16) | (sh << 24), recip | (qbase << 16), ring | (zero << 16) | (coded << 24));
}

// arguments of the mantissa kernels (mant_kernel in decode.hip, mantx_kernel in decode_mx.hip)
struct MantParams {
    const uint8_t *frames;
    const BlkDesc *desc;        // [S*F][6]
    const uint8_t *rows;        // [S*F][6] row sets
    const float *cplco;         // [S*F][6][5][18]
    const uint32_t *frame_pos;  // [S*F]
    float *coef;
    const uint16_t *lfsr_seq;
    const DecTables *tab;
    unsigned n_frames;
    int frame_stride, frame_bytes;
    int acmod, lfeon, n_in, nfchans;
};

// wave-uniform state that runs on from segment to segment of a block
struct Seg2 {
    uint32_t bit;           // first bit of the next segment
    uint32_t phase;         // per kind, 10-bit fields at 20 / 10 / 0: (codes opened so far mod 128) * members + members of the open code
    uint32_t draw;          // dither draws so far
};

// what a lane keeps of its bins between the halves of a segment
template <int NB>
struct Bins2 {
    uint32_t raw[NB];       // the bin's field
    uint32_t rs[NB];        // ring slot of the bin's code | member << 16
    uint32_t dy[NB], dz[NB], dw[NB];
    uint32_t cd;            // index of the lane's first draw
};

// First half: ranks, bit offsets, field extraction; openers publish their codes in ring[] (per-wavefront LDS, M2_LDS_WAVE
// bytes).  bytes4: the lane's NB row bytes, bins outside the segment already remapped to 1.  mult: draws per zero-bit bin.
template <int NB>
__device__ __forceinline__ void seg2_first(uint32_t bytes4, const uint4 *dsc, uint8_t *ring, const uint32_t *frw, uint32_t frw_last,
                                           int mult, Seg2 &S, Bins2<NB> &R, int lane)
{
    uint32_t inc[NB], gl = 0;
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const uint32_t b = (bytes4 >> (8 * j)) & 0xffu;
        const uint4 d = dsc[b];
        inc[j] = d.x; R.dy[j] = d.y; R.dz[j] = d.z; R.dw[j] = d.w;
        gl += d.x;
    }
    const uint32_t gin = wave_incl_scan_u32(gl);
    uint32_t run = gin - gl + S.phase;
    uint32_t nb[NB], nbsum = 0, ndsum = 0, slot[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const uint32_t x = __builtin_amdgcn_ubfe(run, R.dy[j] >> 24, 10);     // (ungrouped: bits 30-31 of run, never incremented: 0)
        run += inc[j];
        const uint32_t q = (x * (R.dz[j] & 0xffffu)) >> 11;
        const uint32_t mem = x - q * ((R.dy[j] >> 16) & 0xffu);
        const uint32_t ob = mem == 0u ? (R.dy[j] >> 8) & 0xffu : 0u;          // an opener's code bits (0 for an ungrouped bin)
        nb[j] = (R.dy[j] & 0xffu) + ob;
        nbsum += nb[j];
        ndsum += ((R.dw[j] >> 16) & 0xffu) * (uint32_t)mult;
        const uint32_t rslot = (R.dw[j] & 0xffffu) + (q & 127u);
        R.rs[j] = rslot | (mem << 16);
        slot[j] = ob != 0u ? rslot : (uint32_t)(M2_RING + 4 * lane);
    }
    const uint32_t bl = nbsum | (ndsum << 16);
    const uint32_t bin_ = wave_incl_scan_u32(bl);
    const uint32_t gtot = wave_last(gin), btot = wave_last(bin_);
    R.cd = S.draw + (bin_ >> 16) - ndsum;
    const uint32_t off = S.bit + (bin_ & 0xffffu) - nbsum;
    // the lane's fields are at most 64 consecutive bits starting at `off`: a 64-bit window out of three dwords
    uint32_t wi = off >> 5;
    wi = wi < frw_last ? wi : frw_last;
    const uint32_t d0 = frw[wi], d1 = frw[wi + 1], d2 = frw[wi + 2];
    const uint32_t k = off & 31u;
    uint64_t win = ((((uint64_t)d0 << 32) | d1) << k) | (uint64_t)((d2 >> 1) >> (31u - k));
#pragma unroll
    for (int j = 0; j < NB; j++) {
        R.raw[j] = ((uint32_t)(win >> 32) >> 1) >> (31u - nb[j]);       // top nb bits (0 for nb = 0)
        win <<= nb[j];
        ring[slot[j]] = (uint8_t)R.raw[j];
    }
    // where the next segment starts
    S.bit += btot & 0xffffu;
    S.draw += btot >> 16;
    {
        const uint32_t x3 = ((S.phase >> 20) & 1023u) + ((gtot >> 20) & 1023u), x5 = ((S.phase >> 10) & 1023u) + ((gtot >> 10) & 1023u),
                       x11 = (S.phase & 1023u) + (gtot & 1023u);
        const uint32_t q3 = (x3 * 683u) >> 11, q5 = (x5 * 683u) >> 11, q11 = x11 >> 1;
        S.phase = (((q3 & 127u) * 3u + (x3 - 3u * q3)) << 20) | (((q5 & 127u) * 3u + (x5 - 3u * q5)) << 10) | ((q11 & 127u) * 2u + (x11 & 1u));
    }
}

// dequantised values of the lane's bins before the exponent / gain scale (0 for a zero-bit bin).  The ring and table reads
// of all bins are issued together (the empty asm pins each code where it is read: the compiler otherwise sinks every
// ring read into a branch of its own, one LDS round trip per bin).
template <int NB>
__device__ __forceinline__ void seg2_values(const Bins2<NB> &R, const uint8_t *ring, const float *qtab, float *q)
{
    uint32_t code[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) code[j] = ring[R.rs[j] & 0xffffu];
#pragma unroll
    for (int j = 0; j < NB; j++) { asm volatile("" : "+v"(code[j])); code[j] &= 0xffu; }
    float tv[NB];
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const uint32_t per = (R.dy[j] >> 16) & 0xffu;
        const uint32_t sel = per ? code[j] * per + (R.rs[j] >> 16) : R.raw[j];
        tv[j] = qtab[(R.dw[j] >> 24) ? (R.dz[j] >> 16) + sel : 0u];
    }
#pragma unroll
    for (int j = 0; j < NB; j++) {
        const uint32_t nbp = R.dy[j] & 0xffu;
        const float pv = (float)(((int32_t)(R.raw[j] << ((32u - nbp) & 31u))) >> 16);      // two's complement fraction, scaled by 2^15
        q[j] = (R.dw[j] >> 24) ? tv[j] : pv;
    }
}

// dither value of draw index i (position along the generator's cycle, table extended past the period): parse.c:310-319
__device__ __forceinline__ float dither2(const int16_t *seq, uint32_t i) { return (float)((3 * (int)seq[i]) >> 2); }

// One audio block.  fetch(slot) -> the lane's row bytes and exponents of that slot's segment (uint2: row bytes, exponents):
// one dword each (bins 4 lane .. 4 lane + 3), for the LFE (slot 5) one byte each (bin = lane, lanes 0..6).
// seq1 = lfsr_seq + 1 + the generator's position before the block's first draw (draw k is seq1[k]).
// PS: floats from one plane of cblk to the next (256 in HBM; mantx_kernel's planes in LDS are padded).
// PK: a slot's end and gain come from MantBlk's packed fields (ends, gainv: one shift / one v_readlane) instead of the arrays -
// hipcc builds a tree of compares and branches per segment out of `slot == 0 ? gain[0] : ...` (240 scalar instructions per
// frame).  Both kernels use it since mant_kernel is compiled for 7 wavefronts per SIMD; at 8 its 78-register scalar budget made
// the two extra 64-bit values cost more in spills than the trees (10.8 k -> 14.2 k vector instructions per frame).
template <int PS = 256, bool PK = false, class Fetch, class Cplco>
__device__ __forceinline__ void mant_block2(const MantBlk &B, Fetch fetch, uint2 first, Cplco cplco_of, const uint8_t *cplbnd, const uint4 *dsc, uint8_t *ring,
                                            const uint32_t *frw, uint32_t frw_last, const float *qtab, const int16_t *seq1, bool lfsr_live,
                                            float *cblk, uint32_t bitpos, int lane)
{
    const int nf = B.nf;
    const int ncpl_dith = __popc(B.chincpl & B.dithmask);
    const int em0 = PK ? (int)((uint32_t)B.ends & 0xffu) : B.endmant[0], em1 = PK ? (int)(((uint32_t)B.ends >> 8) & 0xffu) : B.endmant[1];
    const int remat_end = em0 < em1 ? em0 : em1;
    const bool remat_late = B.acmod == 2 && B.rematflg != 0 && B.chincpl != 0 && remat_end > B.cplstrtmant;      // see mant_block
    const int cplfirst = B.chincpl ? __builtin_ctz(B.chincpl) : 99;
    const int nseg = nf + (B.chincpl ? 1 : 0) + (B.lfeon ? 1 : 0);
    Seg2 S;
    S.bit = bitpos;
    S.phase = 0;
    S.draw = 0;
    uint2 nxt = first;                  // = fetch(slot of segment 0), requested by the caller
    for (int k = 0; k < nseg; k++) {
        const int slot = seg_slot(k, nf, B.chincpl, cplfirst);
        const uint2 cur = nxt;
        if (k + 1 < nseg) nxt = fetch(seg_slot(k + 1, nf, B.chincpl, cplfirst));      // in flight while this segment is unpacked
        if (slot == 5) {
            // LFE: bins 0..6, one per lane; never coupled, never dithered, not rematrixed
            Bins2<1> R;
            seg2_first<1>(lane < 7 ? cur.x & 0xffu : 1u, dsc, ring, frw, frw_last, 0, S, R, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float q;
            seg2_values<1>(R, ring, qtab, &q);
            const float lfe_gain = PK ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, B.gainv), 5)) : B.lfe_gain;
            const float v = q * (sf_of((int)(cur.y & 0xffu)) * lfe_gain);
            const float z = 0.f * (sf_of(0) * lfe_gain);            // what the four-bin pass gives the bins past the seventh
            if (lane < 8) cblk[lane] = lane < 7 ? v : z;
            if (lane >= 2) *reinterpret_cast<float4 *>(cblk + 4 * lane) = make_float4(z, z, z, z);
            continue;
        }
        int start = 0, end, draws;
        float g = 0.f;
        if (slot < 5) {
            if (PK) {
                end = (int)((uint32_t)(B.ends >> (8 * slot)) & 0xffu);
                g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, B.gainv), slot));
            } else {
                end = slot == 0 ? B.endmant[0] : slot == 1 ? B.endmant[1] : slot == 2 ? B.endmant[2] : slot == 3 ? B.endmant[3] : B.endmant[4];
                g = slot == 0 ? B.gain[0] : slot == 1 ? B.gain[1] : slot == 2 ? B.gain[2] : slot == 3 ? B.gain[3] : B.gain[4];
            }
            draws = (B.dithmask >> slot) & 1;
        } else {
            start = B.cplstrtmant;
            end = B.cplendmant;
            draws = ncpl_dith;
        }
        // bins outside [start, end) -> row byte 1
        uint32_t bytes4;
        {
            int lo = start - 4 * lane, hi = end - 4 * lane;
            lo = lo < 0 ? 0 : lo > 4 ? 4 : lo;
            hi = hi < 0 ? 0 : hi > 4 ? 4 : hi;
            const uint32_t below_hi = (uint32_t)((1ull << (8 * hi)) - 1ull), below_lo = (uint32_t)((1ull << (8 * lo)) - 1ull);
            const uint32_t m = below_hi & ~below_lo;
            bytes4 = (cur.x & m) | (0x01010101u & ~m);
        }
        Bins2<4> R;
        seg2_first<4>(bytes4, dsc, ring, frw, frw_last, draws, S, R, lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (slot < 5) {
            float out[4], qv[4];
            seg2_values<4>(R, ring, qtab, qv);
            const bool dith = draws != 0 && lfsr_live;
            const uint32_t z0 = (R.dw[0] >> 16) & 0xffu, z1 = (R.dw[1] >> 16) & 0xffu, z2 = (R.dw[2] >> 16) & 0xffu, z3 = (R.dw[3] >> 16) & 0xffu;
            if (dith && __any((z0 | z1 | z2 | z3) != 0u)) {       // wave-uniform: no table access when no lane draws
                // the lane's draws are consecutive: bin j takes draw cd + (zero-bit bins before it in the lane)
                const int16_t *sp = seq1 + R.cd;
                const float d0 = dither2(sp, 0), d1 = dither2(sp, z0), d2 = dither2(sp, z0 + z1), d3 = dither2(sp, z0 + z1 + z2);
                qv[0] = z0 ? d0 : qv[0]; qv[1] = z1 ? d1 : qv[1]; qv[2] = z2 ? d2 : qv[2]; qv[3] = z3 ? d3 : qv[3];
            }
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int e = (int)((cur.y >> (8 * j)) & 0xffu);
                out[j] = qv[j] * (sf_of(e) * g);              // (bins past the channel's end have no bits: 0)
            }
            float *plane = cblk + (slot + B.in_lfe) * PS;
            if (slot == 1 && B.acmod == 2 && B.rematflg != 0 && !remat_late) {
                // rematrix: parse.c:837-865.  Channel 0's bins were stored by this same lane.
                float4 a4 = *reinterpret_cast<const float4 *>(plane - PS + 4 * lane);
                float a[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int bin = 4 * lane + j;
                    const int band = bin < 25 ? 0 : bin < 37 ? 1 : bin < 61 ? 2 : 3;
                    if (bin >= 13 && (bin < remat_end || (bin == 13 && remat_end <= 13)) && ((B.rematflg >> band) & 1)) {      // (see mant_block)
                        const float x = a[j], v = out[j];
                        a[j] = x + v;
                        out[j] = x - v;
                    }
                }
                *reinterpret_cast<float4 *>(plane - PS + 4 * lane) = make_float4(a[0], a[1], a[2], a[3]);
            }
            if ((B.chincpl >> slot) & 1) {
                // a coupled channel: its own bins, zeros up to the coupling range and from its end on (see mant_block)
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
            uint32_t cd = R.cd;
            float qv[4];
            seg2_values<4>(R, ring, qtab, qv);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int bin = 4 * lane + j;
                const bool in = bin >= start && bin < end;
                const bool zero = ((R.dw[j] >> 16) & 0xffu) != 0u;          // (only bins of the segment carry the flag)
                const int e = (int)((cur.y >> (8 * j)) & 0xffu);
                const float m = qv[j] * sf_of(e);
                const int bnd = cplbnd[in ? (bin - start) / 12 : 0];
                uint32_t cdc = cd;
                for (int c = 0; c < nf; c++) {
                    if (!((B.chincpl >> c) & 1)) continue;
                    const float gc = PK ? __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, B.gainv), c))
                                        : c == 0 ? B.gain[0] : c == 1 ? B.gain[1] : c == 2 ? B.gain[2] : c == 3 ? B.gain[3] : B.gain[4];
                    const float co = cplco_of(c, bnd) * gc;
                    float v = m * co;
                    if (zero) {
                        v = 0.f;
                        if ((B.dithmask >> c) & 1) { v = (sf_of(e) * co) * (lfsr_live ? dither2(seq1, cdc) : 0.f); cdc++; }
                    }
                    if (in) cblk[(c + B.in_lfe) * PS + bin] = v;
                }
                cd += zero ? (uint32_t)draws : 0u;
            }
        }
    }
    if (remat_late) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        volatile float *p0 = cblk + (size_t)B.in_lfe * PS, *p1 = p0 + PS;
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

}  // namespace ac3mi
