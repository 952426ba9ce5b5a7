"""Test-bitstream packer: random but syntactically valid AC-3 frames that use the
features the reference's own encoder never emits - coupling (with band structure, coordinates, phase flags),
rematrixing, block switching, delta bit allocation, dynamic-range words, skip fields, every acmod, LFE
on/off, all three sample rates, half-rate bsid 9/10, optional BSI fields.

Side information and exponents are written by this module following the A/52 syntax exactly as
liba52 consumes it (a52dec-0.7.5-cvs/liba52/parse.c:131-205, 558-804).  Mantissas are random bits: the
decoder under test decides how many it reads.  To place the next block's side information the packer asks
the decode ORACLE where the previous block ended (orc_a52_bitpos) - the oracle is pinned bit-for-bit to the
real liba52 on these very streams (tests/test_packer_streams.py), so the GPU decoder can then be compared
with the oracle on the GPU box, where /root/reference does not exist.
"""
import ctypes

import numpy as np

from tests import _harness as H

NFCHANS = (2, 1, 2, 3, 3, 4, 4, 5)
KBPS = (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640)


class Overflow(Exception):
    pass


class Bits:
    def __init__(self, nbytes):
        self.b = np.zeros(nbytes * 8, np.uint8)

    def put(self, pos, n, v):
        if pos + n > self.b.size:
            raise Overflow()
        for i in range(n):
            self.b[pos + i] = (v >> (n - 1 - i)) & 1
        return pos + n

    def bytes(self):
        return np.packbits(self.b)


def frame_bytes(fscod, frmsizecod):
    rate = KBPS[frmsizecod >> 1]
    if fscod == 0:
        return 4 * rate
    if fscod == 1:
        return 2 * (320 * rate // 147 + (frmsizecod & 1))
    return 6 * rate


def _end_of_block(L, buf, blk, acmod):
    """Bit position at which the decode oracle finishes block `blk` of the frame in buf (-1: it refuses)."""
    st = L.orc_a52_init()
    fl, lv = H.ci(acmod | 16), H.cf(1.0)
    pos = -1
    if L.orc_a52_frame(st, H.P(buf, H.u8p), ctypes.byref(fl), ctypes.byref(lv), 0.0) == 0:
        for b in range(blk + 1):
            if L.orc_a52_block(st) != 0:
                break
        else:
            pos = L.orc_a52_bitpos(st)
    L.orc_a52_free(st)
    return pos


def _put_exp_groups(bits, pos, prev, vals):
    """vals: 3*ngrps exponent values following `prev`; returns new pos"""
    for g in range(len(vals) // 3):
        d = []
        for v in vals[3 * g:3 * g + 3]:
            d.append(v - prev + 2)
            prev = v
        pos = bits.put(pos, 7, 25 * d[0] + 5 * d[1] + d[2])
    return pos


def make_frame(rng, acmod, lfeon, fscod=0, frmsizecod=36, bsid=8, features=None, max_tries=40):
    """One frame.  features: dict of probabilities / switches (see defaults)."""
    f = dict(cpl=0.7, remat=0.7, blksw=0.3, dynrng=0.4, deltba=0.4, skip=0.3, reuse=0.5, bsi_opts=0.5, dsur=0.3,
             cmixlev=int(rng.integers(0, 4)), surmixlev=int(rng.integers(0, 4)))
    if features:
        f.update(features)
    L = H.orc()
    nbytes = frame_bytes(fscod, frmsizecod)
    nf = NFCHANS[acmod]
    for attempt in range(max_tries):
        try:
            frame = _try_frame(rng, L, f, acmod, lfeon, fscod, frmsizecod, bsid, nbytes, nf, attempt)
        except Overflow:
            frame = None
        if frame is not None:
            return frame
    raise RuntimeError("packer: could not fit a frame (acmod %d)" % acmod)


def _try_frame(rng, L, f, acmod, lfeon, fscod, frmsizecod, bsid, nbytes, nf, attempt):
    if True:
        bits = Bits(nbytes + 64)
        # random payload everywhere first: whatever the side info does not overwrite is "mantissas"
        bits.b[:] = rng.integers(0, 2, bits.b.size, dtype=np.uint8)
        pos = bits.put(0, 16, 0x0b77)
        pos = bits.put(pos, 16, int(rng.integers(0, 65536)))            # crc1: never checked by liba52
        pos = bits.put(pos, 2, fscod)
        pos = bits.put(pos, 6, frmsizecod)
        pos = bits.put(pos, 5, bsid)
        pos = bits.put(pos, 3, int(rng.integers(0, 8)))                  # bsmod
        pos = bits.put(pos, 3, acmod)
        if (acmod & 1) and acmod != 1:
            pos = bits.put(pos, 2, f["cmixlev"])
        if acmod & 4:
            pos = bits.put(pos, 2, f["surmixlev"])
        if acmod == 2:
            pos = bits.put(pos, 2, 2 if rng.random() < f["dsur"] else int(rng.integers(0, 2)))   # dsurmod
        pos = bits.put(pos, 1, lfeon)
        for _ in range(2 if acmod == 0 else 1):
            pos = bits.put(pos, 5, int(rng.integers(0, 32)))             # dialnorm
            for width in (8, 8, 7):                                       # compr, langcod, audprodi
                if rng.random() < f["bsi_opts"]:
                    pos = bits.put(pos, 1, 1)
                    pos = bits.put(pos, width, int(rng.integers(0, 1 << width)))
                else:
                    pos = bits.put(pos, 1, 0)
        pos = bits.put(pos, 2, int(rng.integers(0, 4)))                  # copyrightb, origbs
        for _ in range(2):                                                # timecod1/2
            if rng.random() < f["bsi_opts"]:
                pos = bits.put(pos, 1, 1)
                pos = bits.put(pos, 14, int(rng.integers(0, 1 << 14)))
            else:
                pos = bits.put(pos, 1, 0)
        if rng.random() < f["bsi_opts"] * 0.5:                            # addbsi
            n = int(rng.integers(0, 4))
            pos = bits.put(pos, 1, 1)
            pos = bits.put(pos, 6, n)
            for _ in range(n + 1):
                pos = bits.put(pos, 8, int(rng.integers(0, 256)))
        else:
            pos = bits.put(pos, 1, 0)

        # persistent encoder-side view of the decoder state
        cplinu, chincpl, cplbegf, cplendf, ncplbnd, phsflginu = 0, 0, 0, 0, 0, 0
        endmant = [0] * 5
        have_exp = [False] * 5
        have_cpl_exp = have_lfe_exp = False
        have_dba = [False] * 6
        ok = True
        for blk in range(6):
            for _ in range(nf):
                pos = bits.put(pos, 1, int(rng.random() < f["blksw"]))
            for _ in range(nf):
                pos = bits.put(pos, 1, int(rng.random() < 0.7))          # dithflag
            for _ in range(2 if acmod == 0 else 1):
                if rng.random() < f["dynrng"]:
                    pos = bits.put(pos, 1, 1)
                    pos = bits.put(pos, 8, int(rng.integers(0, 256)))
                else:
                    pos = bits.put(pos, 1, 0)
            # ---- coupling strategy ----
            new_cpl = blk == 0 or rng.random() < 0.25
            cpl_changed = False
            if new_cpl:
                pos = bits.put(pos, 1, 1)
                use = acmod >= 2 and rng.random() < f["cpl"]
                pos = bits.put(pos, 1, int(use))
                cplinu = int(use)
                chincpl = 0
                cpl_changed = True
                if use:
                    while chincpl == 0:
                        chincpl = int(rng.integers(0, 1 << nf))
                    for i in range(nf):
                        pos = bits.put(pos, 1, (chincpl >> i) & 1)
                    if acmod == 2:
                        phsflginu = int(rng.integers(0, 2))
                        pos = bits.put(pos, 1, phsflginu)
                    cplbegf = int(rng.integers(0, 16))
                    cplendf = int(rng.integers(max(0, cplbegf - 2), 16))
                    pos = bits.put(pos, 4, cplbegf)
                    pos = bits.put(pos, 4, cplendf)
                    nsub = 3 + cplendf - cplbegf
                    ncplbnd = nsub
                    for _ in range(nsub - 1):
                        b = int(rng.integers(0, 2))
                        pos = bits.put(pos, 1, b)
                        ncplbnd -= b
            else:
                pos = bits.put(pos, 1, 0)
            # ---- coupling coordinates ----
            if cplinu:
                anyco = 0
                for i in range(nf):
                    if (chincpl >> i) & 1:
                        co = cpl_changed or rng.random() < 0.5
                        pos = bits.put(pos, 1, int(co))
                        if co:
                            anyco = 1
                            pos = bits.put(pos, 2, int(rng.integers(0, 4)))
                            for _ in range(ncplbnd):
                                pos = bits.put(pos, 4, int(rng.integers(0, 16)))
                                pos = bits.put(pos, 4, int(rng.integers(0, 16)))
                if acmod == 2 and phsflginu and anyco:
                    for _ in range(ncplbnd):
                        pos = bits.put(pos, 1, int(rng.integers(0, 2)))
            # ---- rematrixing ----
            if acmod == 2:
                if blk == 0 or rng.random() < f["remat"] * 0.5:
                    pos = bits.put(pos, 1, 1)
                    end = (cplbegf * 12 + 37) if cplinu else 253
                    edges = (25, 37, 61, 253)
                    i = 0
                    while True:
                        pos = bits.put(pos, 1, int(rng.random() < f["remat"]))
                        i += 1
                        if not edges[i - 1] < end:
                            break
                else:
                    pos = bits.put(pos, 1, 0)
            # ---- exponent strategies ----
            cplexpstr = 0
            if cplinu:
                cplexpstr = int(rng.integers(1, 4)) if (cpl_changed or not have_cpl_exp or rng.random() > f["reuse"]) else 0
                pos = bits.put(pos, 2, cplexpstr)
            chexpstr = []
            for i in range(nf):
                coupled = cplinu and (chincpl >> i) & 1
                need = blk == 0 or not have_exp[i] or (cpl_changed and coupled) or (cpl_changed and not cplinu and endmant[i] <= 0)
                # a channel whose coupling membership changed needs a new endmant -> new exponents
                if cpl_changed:
                    need = True
                sgy = int(rng.integers(1, 4)) if (need or rng.random() > f["reuse"]) else 0
                chexpstr.append(sgy)
                pos = bits.put(pos, 2, sgy)
            lfeexpstr = 0
            if lfeon:
                lfeexpstr = 1 if (blk == 0 or not have_lfe_exp or rng.random() > f["reuse"]) else 0
                pos = bits.put(pos, 1, lfeexpstr)
            for i in range(nf):
                if chexpstr[i]:
                    if cplinu and (chincpl >> i) & 1:
                        endmant[i] = cplbegf * 12 + 37
                    else:
                        bw = int(rng.integers(0, 61))
                        pos = bits.put(pos, 6, bw)
                        endmant[i] = bw * 3 + 73
            # ---- exponents ----
            if cplexpstr:
                gs = 3 << (cplexpstr - 1)
                ngrp = ((cplendf * 12 + 73) - (cplbegf * 12 + 37)) // gs
                absexp = int(rng.integers(0, 13))                          # cplabsexp (<<1 in the decoder)
                pos = bits.put(pos, 4, absexp)
                pos = _put_exp_groups(bits, pos, 2 * absexp, _chain(rng, 2 * absexp, 3 * ngrp))
                have_cpl_exp = True
            for i in range(nf):
                if chexpstr[i]:
                    gs = 3 << (chexpstr[i] - 1)
                    ngrp = (endmant[i] + gs - 4) // gs
                    e0 = int(rng.integers(0, 16))
                    pos = bits.put(pos, 4, e0)
                    pos = _put_exp_groups(bits, pos, e0, _chain(rng, e0, 3 * ngrp))
                    pos = bits.put(pos, 2, int(rng.integers(0, 4)))        # gainrng
                    have_exp[i] = True
            if lfeexpstr:
                e0 = int(rng.integers(0, 16))
                pos = bits.put(pos, 4, e0)
                pos = _put_exp_groups(bits, pos, e0, _chain(rng, e0, 6))
                have_lfe_exp = True
            # ---- bit allocation parameters ----
            if blk == 0 or rng.random() < 0.3:
                pos = bits.put(pos, 1, 1)
                pos = bits.put(pos, 11, int(rng.integers(0, 1 << 11)))    # sdcycod fdcycod sgaincod dbpbcod floorcod
            else:
                pos = bits.put(pos, 1, 0)
            if blk == 0 or (cpl_changed and cplinu) or rng.random() < 0.3:   # cpl snr offset travels only here
                pos = bits.put(pos, 1, 1)
                # modest SNR offsets so that the six blocks fit the frame
                pos = bits.put(pos, 6, int(rng.integers(0, max(4, 28 - 3 * attempt))))
                if cplinu:
                    pos = bits.put(pos, 7, int(rng.integers(0, 128)))
                for _ in range(nf):
                    pos = bits.put(pos, 7, int(rng.integers(0, 128)))
                if lfeon:
                    pos = bits.put(pos, 7, int(rng.integers(0, 128)))
            else:
                pos = bits.put(pos, 1, 0)
            if cplinu:
                if cpl_changed or rng.random() < 0.3:
                    pos = bits.put(pos, 1, 1)
                    pos = bits.put(pos, 3, int(rng.integers(0, 8)))
                    pos = bits.put(pos, 3, int(rng.integers(0, 8)))
                else:
                    pos = bits.put(pos, 1, 0)
            if rng.random() < f["deltba"]:
                pos = bits.put(pos, 1, 1)
                codes = []
                # "reuse" (0) is only meaningful after a "new" (1) earlier in the same frame: a52_frame
                # resets deltbae but not the deltba arrays (parse.c:174-176)
                if cplinu:
                    codes.append(int(rng.choice([0, 1, 2] if have_dba[5] else [1, 2])))
                    have_dba[5] = have_dba[5] or codes[-1] == 1
                    pos = bits.put(pos, 2, codes[-1])
                for i in range(nf):
                    codes.append(int(rng.choice([0, 1, 2] if have_dba[i] else [1, 2])))
                    have_dba[i] = have_dba[i] or codes[-1] == 1
                    pos = bits.put(pos, 2, codes[-1])
                for c in codes:
                    if c == 1:
                        nseg = int(rng.integers(0, 4))
                        pos = bits.put(pos, 3, nseg)
                        band = 0
                        for _ in range(nseg + 1):
                            room = 49 - band
                            off = int(rng.integers(0, max(1, min(6, room - 1))))
                            ln = int(rng.integers(0, max(1, min(6, room - off - 1))))
                            pos = bits.put(pos, 5, off)
                            pos = bits.put(pos, 4, ln)
                            pos = bits.put(pos, 3, int(rng.integers(0, 8)))
                            band += off + ln
            else:
                pos = bits.put(pos, 1, 0)
            if rng.random() < f["skip"]:
                n = int(rng.integers(0, 6))
                pos = bits.put(pos, 1, 1)
                pos = bits.put(pos, 9, n)
                pos += 8 * n                                               # skipped bytes stay random
            else:
                pos = bits.put(pos, 1, 0)
            # ---- where do this block's mantissas end?  ask the oracle ----
            pos = _end_of_block(L, bits.bytes(), blk, acmod)
            if pos < 0 or pos > nbytes * 8 - 18:
                ok = False
                break
        if ok:
            return bits.bytes()[:nbytes].copy()
        return None


def _chain(rng, prev, n):
    """n exponent values, each within +-2 of its predecessor (starting after `prev`), in 0..24"""
    out = []
    for _ in range(n):
        lo, hi = max(0, prev - 2), min(24, prev + 2)
        prev = int(rng.integers(lo, hi + 1))
        out.append(prev)
    return out


def make_stream(seed, nframes, acmod, lfeon, **kw):
    """nframes frames of one stream.  The mix levels are a property of the programme, so they are drawn once
    per stream (liba52 itself behaves oddly when surmixlev flips to "0" between frames while block switching
    differs across channels: the time-domain mixer then drops the surround overlap tails, downmix.c:560-565)."""
    rng = np.random.default_rng(seed)
    feats = dict(kw.pop("features", None) or {})
    feats.setdefault("cmixlev", int(rng.integers(0, 4)))
    feats.setdefault("surmixlev", int(rng.integers(0, 4)))
    return np.stack([make_frame(rng, acmod, lfeon, features=feats, **kw) for _ in range(nframes)])


def make_flip_stream(seed, surmixlevs, acmod=7, lfeon=0, **kw):
    """A stream whose surround mix level changes from frame to frame (surmixlevs: one 2-bit code per frame; code 2 = "no
    surround", level 0).  Valid AC-3, never seen in practice - and the one case where liba52's per-channel overlap
    planes and time-domain mixer (parse.c:884-937, downmix.c:559-562) do not behave like a linear mix."""
    rng = np.random.default_rng(seed)
    cmix = int(rng.integers(0, 4))
    return np.stack([make_frame(rng, acmod, lfeon, features={"cmixlev": cmix, "surmixlev": int(lv)}, **kw) for lv in surmixlevs])
