// enc_mant.h — the mantissas of one audio block: quantise, group, pack (ENC/ac3enc.cpp:1341-1501), shared by the packers
// of encode.hip (one wavefront per frame: enc_pack_kernel<2>; one wavefront per audio block: enc_packb_kernel).
//
// The reference quantises a whole block into qmant[] - the first member of a 3- / 5- / 11-level group collects the
// code, the others are marked "merged" - and then writes the block out.  Here (round 4; rounds 1-3 assembled the codes
// per coefficient in rings of LDS atomics, two bit-field writes per coefficient):
//   stage 1, one pass per channel, four consecutive coefficients per lane: bap table address -> what the code implies
//            (packlut), quantise, the coefficient's rank among its kind in the block (one wavefront scan of three packed
//            counters per pass); the 2nd / 3rd member of a group drops its 16-bit value into the kind's member list at
//            entry 2 g + m - 1 - dense, so a block's lists are written before they are read and never cleared;
//            everything stage 2 needs is ONE word per coefficient kept in registers;
//   stage 2, same passes: a group's opener reads its members (one aligned 32-bit LDS read), forms the code - so a code equal
//            to the reference's "merged" marker (:1466-1480, never written) is simply a field of no bits, decided before
//            the offsets are summed: no second attempt - one scan over the lanes' bits gives the offsets, the lane's
//            four fields are concatenated in a 64-bit register and reach the frame with three LDS ORs.
// Out-of-contract values (sym_quant with a negative shift, :1150-1166) are wider than their fields and
// the release build's put_bits does not mask them: the excess bits are OR-ed in a second sweep that only blocks with a
// negative shift run (wave-uniform flag), exactly where put_bits would have put them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "wave_ops.h"

namespace ac3mi {

// section timers of a -DPACK_STAMPS build (profiles/pack_stamps.py): 9 stage 1, 10 stage 2 of mant_pack_block
#ifdef PACK_STAMPS
extern __device__ unsigned long long g_pack_cycles[16];
#define MANT_T0() unsigned long long mt_ = __builtin_readcyclecounter()
#define MANT_LAP(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_pack_cycles[k], t_ - mt_); mt_ = __builtin_readcyclecounter(); } while (0)
#else
#define MANT_T0() do { } while (0)
#define MANT_LAP(k) do { } while (0)
#endif

// member lists (uint16 entries): the 2nd / 3rd member of group g of a kind sits at entries 2 g / 2 g + 1 of the kind's list, so that
// the group's opener fetches both with ONE aligned 32-bit read.  Kinds 0 / 1 (3- / 5-level codes, three members) at 0 / 752:
// at most 748 members + the two entries cleared behind the last one; kind 2 (11-level, two members: entry 2 g + 1 is never
// written and stays zero from mant_lists_init) at 1504: entries up to 2 x 561 + 1; then one sink entry per lane.
constexpr int GL_STRIDE = 752;
constexpr int GL_KIND2 = 1128;
constexpr int GL_SINK = 2 * GL_STRIDE + GL_KIND2;
constexpr int GL_ENTRIES = GL_SINK + 64;

// once per wavefront and kernel: the two-member kind's odd entries must read as zero
__device__ __forceinline__ void mant_lists_init(uint16_t *glist, int lane)
{
    static_assert((2 * GL_STRIDE) % 8 == 0 && GL_KIND2 % 8 == 0, "16-byte stores");
    uint4 *q = reinterpret_cast<uint4 *>(glist + 2 * GL_STRIDE);
    for (int i = lane; i < GL_KIND2 / 8; i += 64) q[i] = make_uint4(0, 0, 0, 0);
}

// What the packing passes need to know about a bap code, as fields of one word (packlut[address]):
//   0-4 plain bits (0 for the grouped codes and bap 0)   5-6 kind (0/1/2 = member of a 3-/5-/11-level code, 3 = not grouped)
//   7-10 bap   11-14 levels of the symmetric quantiser   15 symmetric
//   16-23 0x80 | bits of the grouped code (0 when not grouped)   24-28 10 * kind (position of the kind's counter in the packed rank words)
__device__ __forceinline__ uint32_t mant_pack_word(int bp, int plain_bits)
{
    const uint32_t kind = bp == 1 ? 0u : bp == 2 ? 1u : bp == 4 ? 2u : 3u;
    const uint32_t gbits = kind == 0 ? 5u : kind < 3 ? 7u : 0u;
    const uint32_t levels = bp == 1 ? 3u : bp == 2 ? 5u : bp == 4 ? 11u : bp == 3 ? 7u : 15u;
    const uint32_t sym = (kind < 3 || bp == 3 || bp == 5) ? 1u : 0u;
    return (uint32_t)plain_bits | (kind << 5) | ((uint32_t)bp << 7) | (levels << 11) | (sym << 15) | ((kind < 3 ? 0x80u | gbits : 0u) << 16) |
           ((10u * kind) << 24);
}

__device__ __forceinline__ int mant_quant_sym(int c, int e, int levels)       // :1150-1166
{
    // out of contract when e < 0 (a reuse run pulled the exponent below the block's shift): as the x86 build runs it -
    // shift count masked to 5 bits, 32-bit wrap-around multiply, arithmetic right shift
    const uint32_t a = (uint32_t)(c >= 0 ? c : -c) << (e & 31);
    int v = (int32_t)((uint32_t)levels * a) >> 24;
    v = (v + 1) >> 1;
    return c >= 0 ? (levels >> 1) + v : (levels >> 1) - v;
}
__device__ __forceinline__ int mant_quant_asym(int c, int e, int qbits)       // :1169-1190
{
    const int lshift = e + qbits - 24;
    const int up = (int)((unsigned)c << (lshift & 31)), down = c >> ((-lshift) & 31);      // both, then a select: no branch
    int v = lshift >= 0 ? up : down;
    v = (v + 1) >> 1;
    const int m = 1 << (qbits - 1);
    if (v >= m) v = m - 1;
    return v & ((1 << qbits) - 1);
}

// Both quantisers for a coefficient in contract - shift e >= 0 and |c| << e < 2^24, which the exponent the encoder sends
// guarantees unless a reuse run pulled it below the block's exp_samples - with fewer instructions and no 32-bit multiply:
//   Y = c << e;   symmetric: ((levels |Y|) >> 24 + 1) >> 1 = (levels |Y| + 2^24) >> 25, sign of c;
//   asymmetric, w bits: c << (e + w - 24) or c >> (24 - w - e) is Y >> (24 - w) either way; ((Y >> s) + 1) >> 1 = (Y + 2^s) >> (s + 1).
// Identical results there (tests/test_encode_gpu.py compares every frame with the oracle); blocks with a negative shift
// anywhere take mant_quant_sym / mant_quant_asym, which restate what the x86 build does out of contract.
__device__ __forceinline__ uint32_t mant_quant_fast(int c, int e, uint32_t pw)
{
    const int w = (int)(pw & 31u), levels = (int)((pw >> 11) & 15u);
    const int Y = (int)((uint32_t)c << (e & 31));
    const int aY = Y < 0 ? -Y : Y;
    const int vs0 = (int)((__umul24((uint32_t)levels, (uint32_t)aY) + (1u << 24)) >> 25);
    const int vs = (levels >> 1) + (Y < 0 ? -vs0 : vs0);
    const int s = 24 - w;
    int va = (Y + (1 << (s & 31))) >> ((s + 1) & 31);
    const int cm = (int)((1u << ((w - 1) & 31)) - 1u);
    va = va < cm ? va : cm;
    va = (int)__builtin_amdgcn_ubfe((uint32_t)va, 0u, (uint32_t)w);
    return (uint32_t)(((pw >> 15) & 1u) ? vs : va);
}

// put_bits (:148-176) for every lane at once, no branch: a field of no bits (or outside the frame) goes to the lane's own
// sink words.  `v` may be wider than n bits (the release build does not mask it): the excess is OR-ed onto the bits before
// the field as far as the 32-bit word it starts in, which is what the 64-bit shift does.
__device__ __forceinline__ void mant_put_bits_always(uint32_t *fr, int frw, uint32_t *sink, uint32_t pos, int n, uint32_t v)
{
    const int n_in = (pos >> 5) + 1 < (uint32_t)frw ? n : 0;
    uint32_t *dst = n_in > 0 ? fr + (pos >> 5) : sink;
    const uint64_t x = (uint64_t)v << ((64 - n - (int)(pos & 31)) & 63);
    atomicOr(dst, (uint32_t)(x >> 32));
    atomicOr(dst + 1, (uint32_t)x);
}

// bap of one coefficient for SNR offset `snroffset` (:393-420):
//   v = ((max(mask - snroffset - floor, 0)) & 0x1fe0) + floor,  address = (psd - v) >> 5,  psd = 3072 - 128 exp
//   =>  address = clamp(80 - 4 exp - max(0, (mask - floor - snroffset) >> 5), 0, 63)       (floor = 0x1f0)
// mant_band_term turns a band's (mask - floor) into T = 80 - max(0, (mask - floor - snroffset) >> 5), once per band and
// block, so that a coefficient's address is clamp(T[band] - 4 exp, 0, 63): one multiply-add and one median.  Two bands per dword.
__device__ __forceinline__ uint32_t mant_band_terms(uint32_t two_masks, int snroffset)
{
    const int m0 = (int)(int16_t)(two_masks & 0xffffu), m1 = (int)two_masks >> 16;
    int q0 = (m0 - snroffset) >> 5, q1 = (m1 - snroffset) >> 5;
    q0 = q0 < 0 ? 0 : q0;
    q1 = q1 < 0 ? 0 : q1;
    return (uint32_t)((80 - q0) & 0xffff) | ((uint32_t)(80 - q1) << 16);
}

// The bap table addresses of a block's coefficients, four per lane and channel, one byte each.  T: the block's band terms
// [nch][50] (mant_band_terms) in LDS; bandoff: the bands of the lane's four bins, one byte each.  The exponent bytes of bins the
// channel does not code (beyond nbc, beyond the LFE's 7) are set to 255 in `em`, the copy the mantissa passes work on:
// address 0 = bap 0, and a shift that is never negative.  `neg` collects the sign of every coefficient's shift e = exponent - exp_samples: negative anywhere in the
// block = out-of-contract values possible (see the header comment).
__device__ __forceinline__ void mant_block_addresses(uint32_t (&ad)[6], uint32_t (&em)[6], int &neg, const uint32_t (&ew)[6], const int (&shv)[6],
                                                     const int16_t (*T)[50], uint32_t bandoff, int nch, int nbc, bool lfe, int lane)
{
    neg = 0;
    auto beyond = [&](int n) {                  // 0xff in the bytes of bins >= n
        const int k = n - 4 * lane;             // coded bins of this lane: k <= 0 none, k >= 4 all
        return k >= 4 ? 0u : k <= 0 ? 0xffffffffu : 0xffffffffu << (8 * k);
    };
    const uint32_t um_fbw = beyond(nbc), um_lfe = beyond(7);
#pragma unroll
    for (int ch = 0; ch < 6; ch++) {
        ad[ch] = 0;
        em[ch] = 0;
        if (ch < nch) {
            em[ch] = ew[ch] | ((lfe && ch == nch - 1) ? um_lfe : um_fbw);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int xe = (int)((em[ch] >> (8 * j)) & 0xff);
                int a = (int)T[ch][(bandoff >> (8 * j)) & 0xff] - 4 * xe;
                a = a < 0 ? 0 : a > 63 ? 63 : a;
                ad[ch] |= (uint32_t)a << (8 * j);
                neg |= xe - shv[ch];
            }
        }
    }
}

struct MantBlock {
    uint32_t *fr;               // the frame, MSB-first dwords (LDS)
    int frw;                    // its dwords
    uint16_t *glist;            // GL_ENTRIES member-list entries of this wavefront (LDS)
    const uint32_t *packlut;    // mant_pack_word per bap table address (LDS)
    const int32_t *mdb;         // the block's coefficient rows [nch][256]
    uint8_t *tap_bap;           // optional: the block's bap rows [nch][256]
    int nch, nbc;
    bool lfe;
    uint32_t marker;            // the reference's "member already merged" value, 128 (:1375-1413)
};

// Packs the block's mantissas from bit `pos` on; returns the first bit after them.  ew: the encoded exponents (`em`), ad: the
// bap table addresses (mant_block_addresses), shv: exp_samples per channel (wave-uniform), garbage: a coded coefficient
// of the block has a negative shift (wave-uniform).
__device__ __forceinline__ uint32_t mant_pack_block(const MantBlock &B, const uint32_t (&ew)[6], const uint32_t (&ad)[6], const int (&shv)[6],
                                                    bool garbage, uint32_t pos, int lane)
{
    const int nch = B.nch, nbc = B.nbc;
    // The LFE's seven coefficients ride in the LAST full-bandwidth channel's pass: that channel's 223 bins fill lanes 0..55,
    // lane 56 + k takes LFE bin k in its first slot - lane order is bitstream order (the LFE follows the last channel,
    // :1341-1501), so ranks, offsets and grouped codes come out as from a pass of its own, which would cost as much as a
    // full channel's.
    const bool lfe_rides = B.lfe && nch >= 2 && nbc <= 224;
    const int npass = lfe_rides ? nch - 1 : nch;
    const int lk = lane - 56;                                   // the LFE bin of this lane in the merged pass
    int lfe_c = 0;
    uint32_t lfe_e = 0, lfe_a = 0;
    int lfe_sh = 0;
    if (lfe_rides) {
        if (lk >= 0 && lk < 7) lfe_c = B.mdb[(nch - 1) * 256 + lk];
        uint32_t le = 0, la = 0;
#pragma unroll
        for (int c2 = 0; c2 < 6; c2++) { le = c2 == nch - 1 ? ew[c2] : le; la = c2 == nch - 1 ? ad[c2] : la; lfe_sh = c2 == nch - 1 ? shv[c2] : lfe_sh; }
        const int srcl = lk >= 0 ? lk >> 2 : 0;
        // (slots 1..3 of an LFE lane: exponent 24, address 0 = bap 0, no bits)
        lfe_e = (((uint32_t)__shfl((int)le, srcl, 64) >> (8 * (lk & 3))) & 0xffu) | 0x18181800u;
        lfe_a = lk < 7 ? ((uint32_t)__shfl((int)la, srcl, 64) >> (8 * (lk & 3))) & 63u : 0u;
    }

    uint32_t st[6][4];          // per coefficient: value (0-15) | list position relative to the pass (16-23) | bits (24-28) | kind (29-30) | opens a code (31)
    uint32_t bw[6];             // per pass: groups of each kind complete before it (10-bit fields, wave-uniform)
    int P0 = 0, P1 = 0, P2 = 0; // 3- / 5- / 11-level mantissas of the block so far

    int4 nx_c = *reinterpret_cast<const int4 *>(B.mdb + 4 * lane);

    // ---- stage 1 ----
    MANT_T0();
#pragma unroll
    for (int p = 0; p < 6; p++) {
        if (p >= npass) continue;
        int4 c4 = nx_c;
        {
            // (the next channel's coefficients are in flight meanwhile.  Requesting all of a block's rows before its side
            // information was measured too: 123 VGPRs and 1.88 ms per 65 536 frames against 1.78 - the rows wait in registers the
            // passes would rather use, and the loads were never what the wavefronts waited for: a build without them runs 1.74)
            const int nc = p + 1 < npass ? p + 1 : p;
            nx_c = *reinterpret_cast<const int4 *>(B.mdb + nc * 256 + 4 * lane);
        }
        const bool merged = lfe_rides && p == npass - 1;        // wave-uniform
        const bool lfe_lane = merged && lk >= 0;
        uint32_t e4 = ew[p], a4 = ad[p];
        int sv = shv[p];
        if (merged) {
            e4 = lfe_lane ? lfe_e : e4;
            a4 = lfe_lane ? lfe_a : a4;
            c4 = lfe_lane ? make_int4(lfe_c, 0, 0, 0) : c4;
            sv = lfe_lane ? lfe_sh : sv;
        }
        const int cj[4] = {c4.x, c4.y, c4.z, c4.w};
        uint32_t pw[4], cnt_lane = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            pw[j] = B.packlut[(a4 >> (8 * j)) & 63u];
            cnt_lane += 1u << (pw[j] >> 24);                    // (a bin that is not grouped counts in bits 30-31: ignored)
        }
        // ranks: the kind's count in the block modulo its group size (0..2 / 0..1) + the lanes before + the lane's own bins
        const int G0 = (int)(((uint32_t)P0 * 0xaaabu) >> 17), G1 = (int)(((uint32_t)P1 * 0xaaabu) >> 17), G2 = P2 >> 1;
        const uint32_t phase = (uint32_t)(P0 - 3 * G0) | ((uint32_t)(P1 - 3 * G1) << 10) | ((uint32_t)(P2 & 1) << 20);
        bw[p] = (uint32_t)G0 | ((uint32_t)G1 << 10) | ((uint32_t)G2 << 20);
        const uint32_t gin = wave_incl_scan_u32(cnt_lane);
        uint32_t run = gin - cnt_lane + phase;
        // quantise (:1150-1190): 16-bit values (qmant[] is unsigned short, :1347)
        uint32_t qv[4];
        if (garbage) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int levels = (int)((pw[j] >> 11) & 15u), w = (int)(pw[j] & 31u);
                const int e = (int)((e4 >> (8 * j)) & 0xff) - sv;
                const int vs = mant_quant_sym(cj[j], e, levels), va = mant_quant_asym(cj[j], e, w ? w : 1);
                qv[j] = (uint32_t)(((pw[j] >> 15) & 1u) ? vs : va) & 0xffffu;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) qv[j] = mant_quant_fast(cj[j], (int)((e4 >> (8 * j)) & 0xff) - sv, pw[j]) & 0xffffu;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t sh = pw[j] >> 24, kind = (pw[j] >> 5) & 3u;
            const uint32_t r = __builtin_amdgcn_ubfe(run, sh, 10u);         // rank counted from the opener of the group open at the pass's start
            run += 1u << sh;
            const uint32_t k1 = kind >> 1;                                   // members per group: 3 - k1
            const uint32_t g = __umul24(r, 0xaaabu + k1 * 0x5555u) >> 17;   // r / 3 or r / 2 (r <= 258)
            const bool first = r == __umul24(g, 3u - k1);                    // the group's first member
            const uint32_t arel = __umul24(k1, g) + (r - g);                 // 2 g + m (m = 0, 1, 2: which member): a later member's list entry + 1, the opener's pair
            const uint32_t gfield = (pw[j] >> 16) & 0xffu;                  // 0x80 | code bits of a grouped bap, else 0
            const uint32_t t = first ? gfield : 0u;                         // the opener's
            const uint32_t u = first ? 0u : gfield;                         // a later member's
            const uint32_t q = qv[j];
            const uint32_t lpos = kind * (uint32_t)GL_STRIDE + 2u * __builtin_amdgcn_ubfe(bw[p], sh, 10u) + arel - 1u;
            B.glist[u ? lpos : (uint32_t)(GL_SINK + lane)] = (uint16_t)q;
            st[p][j] = q | (arel << 16) | (((pw[j] & 0x7fu) | t) << 24);    // (0x80 of t: bit 31; its code bits replace the - zero - plain bits)
        }
        if (B.tap_bap) {
            uint8_t *tb = B.tap_bap + p * 256;
            const uint32_t four = ((pw[0] >> 7) & 15u) | (((pw[1] >> 7) & 15u) << 8) | (((pw[2] >> 7) & 15u) << 16) | (((pw[3] >> 7) & 15u) << 24);
            *reinterpret_cast<uint32_t *>(tb + 4 * lane) = lfe_lane ? 0u : four;
            if (merged) {                                       // the LFE's row: bins 0..6 from lanes 56..62, zeros beyond
                uint8_t *tl = B.tap_bap + (nch - 1) * 256;
                if (lane >= 2) *reinterpret_cast<uint32_t *>(tl + 4 * lane) = 0u;
                if (lfe_lane) tl[lk] = (uint8_t)(lk < 7 ? (pw[0] >> 7) & 15u : 0u);
            }
        }
        {
            const uint32_t gtot = wave_last(gin);
            P0 += (int)(gtot & 1023u); P1 += (int)((gtot >> 10) & 1023u); P2 += (int)((gtot >> 20) & 1023u);
        }
    }
    // the two entries behind each kind's last member: what the opener of an unfinished group reads for the members that never
    // came ("a trailing group is written as it stands"); a block's lists are otherwise written before they are read
    {
        const int E0 = P0 - (int)(((uint32_t)(P0 + 2) * 0xaaabu) >> 17), E1 = P1 - (int)(((uint32_t)(P1 + 2) * 0xaaabu) >> 17), E2 = P2 & ~1;
        const int k = lane >> 1;
        const int E = k == 0 ? E0 : k == 1 ? E1 : E2;
        if (lane < 6) B.glist[k * GL_STRIDE + E + (lane & 1)] = 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    MANT_LAP(9);
    // ---- stage 2 ----
    // (fields beyond the buffer - a failed search's overflow - land in its last three dwords: the headroom behind the frame's bytes)
    uint32_t *const sink = B.fr + B.frw - 3;
#pragma unroll
    for (int p = 0; p < 6; p++) {
        if (p >= npass) continue;
        uint32_t mm[4];                                         // the openers' 2nd | 3rd member << 16
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t S = st[p][j];
            const uint32_t kind = (S >> 29) & 3u;
            const uint32_t lpos = kind * (uint32_t)GL_STRIDE + 2u * __builtin_amdgcn_ubfe(bw[p], kind * 10u, 10u) + ((S >> 16) & 0xffu);
            mm[j] = *reinterpret_cast<const uint32_t *>(B.glist + ((int)S < 0 ? lpos : 0u));
        }
        uint32_t nb[4], val[4], bits_lane = 0;
        uint64_t cat = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t S = st[p][j];
            const bool opens = (int)S < 0;
            const uint32_t k8 = ((S >> 29) & 3u) * 8u;
            // code modulo 2^16 (:1365-1431): 9 q + 3 m1 + m2, 25 q + 5 m1 + m2, 11 q + m1 (+ the entry that is always 0)
            const uint32_t W0 = __builtin_amdgcn_ubfe(0x0b1909u, k8, 8u), W1 = __builtin_amdgcn_ubfe(0x010503u, k8, 8u);
            const uint32_t code = (mul24_asm(S & 0xffffu, W0) + mul24_asm(mm[j] & 0xffffu, W1) + (mm[j] >> 16)) & 0xffffu;
            const uint32_t n0 = (S >> 24) & 31u;
            nb[j] = opens && code == B.marker ? 0u : n0;
            val[j] = opens ? code : S & 0xffffu;
            bits_lane += nb[j];
            cat = (cat << nb[j]) | __builtin_amdgcn_ubfe(val[j], 0u, nb[j]);
        }
        const uint32_t bin_ = wave_incl_scan_u32(bits_lane);
        const uint32_t off = pos + bin_ - bits_lane;
        {
            // the lane's bits, left-aligned in 64, shifted to the bit they start at inside their first dword: three dwords
            const uint64_t T = cat << ((64u - bits_lane) & 63u);
            const uint32_t hi = (uint32_t)(T >> 32), lo = (uint32_t)T, p0 = off & 31u;
            const uint32_t d0 = hi >> p0, d1 = __builtin_amdgcn_alignbit(hi, lo, p0), d2 = __builtin_amdgcn_alignbit(lo, 0u, p0);
            uint32_t *dst = (off >> 5) + 3u <= (uint32_t)B.frw ? B.fr + (off >> 5) : sink;
            atomicOr(dst, d0);
            atomicOr(dst + 1, d1);
            if (__ballot(d2 != 0u)) atomicOr(dst + 2, d2);     // (only a lane with more than 32 bits reaches a third dword)
        }
        if (garbage) {
            uint32_t o = off;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t excess = val[j] & ~((1u << nb[j]) - 1u);
                mant_put_bits_always(B.fr, B.frw, sink, o, excess ? (int)nb[j] : 0, excess);
                o += nb[j];
            }
        }
        pos += wave_last(bin_);
    }
    MANT_LAP(10);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return pos;
}

}  // namespace ac3mi
