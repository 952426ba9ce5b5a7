#!/usr/bin/env python3
"""How many sweeps (up to NC offsets costed at once) does a candidate policy need to replay the reference's SNR-offset search
(ENC/ac3enc.cpp:921-967) exactly?  CPU only: the oracle tabulates a frame's spare bits at all 1024 offsets
(orc_ac3enc_set_spare_curve), the policies below are then run on that curve.  `python profiles/search_sim.py [frames] [start]`.
Policies mirror enc_pack_kernel's host-free scalar logic (encode.hip): `ladder` = round 3's (quartered ladder + look-ahead),
`probe` = interpolation probes while the window between the monotone bounds is wide."""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import _harness as H

NC = 3
MARGIN = 72         # rounds 2-3: a fit / miss by this many bits decides every lower / higher offset


class Curve(np.ndarray):
    """a frame's spare bits at all 1024 offsets; .extra: the group ceilings in each count, in sixths of a bit (0 .. 414) -
    what the round-4 kernel turns into tighter bounds: spare >= 69 - extra/6 decides every lower offset, spare < -extra/6
    every higher one (encode.hip, enc_search_kernel)"""
    extra = None


def curves(n, seed=99, second_gen=False, with_source=False):
    """spare-bit curves of n frames; second_gen: of decoded audio fed back (the transcode's encoder half), with_source: as
    (curve, 16 csnroffst + fsnroffst of the frame the audio was decoded from) pairs"""
    import torch
    g = torch.Generator().manual_seed(seed)
    S = n
    t = torch.arange(1536, dtype=torch.float32)
    ph = torch.rand((S, 1, 6), generator=g) * 6.28
    fr = 0.01 * torch.arange(1, 7, dtype=torch.float32)
    pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), generator=g) - 0.5) * 4096
    env = torch.where(torch.rand((S, 3, 1, 6), generator=g) < 0.5, 1.0, 1.0 / 32)
    pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6).round().clamp(-32768, 32767).to(torch.int16).numpy()
    O = H.orc()
    O.orc_ac3enc_encode_frames.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, H.i16p, ctypes.c_int, H.u8p, H.u8p]
    O.orc_ac3enc_set_spare_curve.argtypes = [ctypes.c_void_p]
    cm = (ctypes.c_uint8 * 8)(*H.CHMAP6)
    out = []
    buf = np.zeros(1024, np.int32)
    xbuf = np.zeros(1024, np.int32)
    O.orc_ac3enc_set_extra_curve.argtypes = [ctypes.c_void_p]
    O.orc_ac3enc_set_spare_curve(buf.ctypes.data)
    O.orc_ac3enc_set_extra_curve(xbuf.ctypes.data)
    frame = np.zeros(1536, np.uint8)
    for i in range(n):
        src = np.ascontiguousarray(pcm[i].reshape(-1))
        g_src = None
        if second_gen:          # decoded audio fed back: encode, decode to s16, encode that
            assert O.orc_ac3enc_encode_frames(48000, 384000, 6, H.P(src, H.i16p), 1, cm, H.P(frame, H.u8p)) == 0
            c_src, f_src = reference(buf, 40)
            g_src = 16 * c_src + f_src
            pcmf, errs, oflags = H.orc_decode(frame[None, :], 7 | 16 | 32, 1.0, 384.0)
            s16 = np.zeros((6, 256, 6), np.int16)
            ref16 = np.zeros((256, 6), np.int16)
            for b in range(6):
                O.orc_convert_s16(H.P(np.ascontiguousarray(pcmf[0, b]), H.fp), H.P(ref16, H.i16p), oflags)
                s16[b] = ref16
            src = np.ascontiguousarray(s16.reshape(-1))
        assert O.orc_ac3enc_encode_frames(48000, 384000, 6, H.P(src, H.i16p), 1, cm, H.P(frame, H.u8p)) == 0
        cv = buf.copy().view(Curve)
        cv.extra = xbuf.copy()
        out.append((cv, g_src) if with_source else cv)
    O.orc_ac3enc_set_spare_curve(None)
    O.orc_ac3enc_set_extra_curve(None)
    return out


class Search:       # SnrSearch of encode.hip
    def __init__(self, c):
        self.c, self.f, self.phase, self.failed = c, 0, 0, False

    def next(self):
        while True:
            cc, ff = self.c, self.f
            if self.phase == 0:
                if self.c < 0:
                    self.failed, self.phase = True, 5
                    return None
                return cc, ff
            if self.phase == 1:
                if self.c + 4 > 63: self.phase = 2; continue
                return self.c + 4, ff
            if self.phase == 2:
                if self.c + 1 > 63: self.phase = 3; continue
                return self.c + 1, ff
            if self.phase == 3:
                if self.f + 4 > 15: self.phase = 4; continue
                return cc, self.f + 4
            if self.phase == 4:
                if self.f + 1 > 15: self.phase = 5; return None
                return cc, self.f + 1
            return None

    def consume(self, ok):
        if self.phase == 0:
            if ok: self.phase = 1
            else: self.c -= 4
        elif self.phase == 1:
            if ok: self.c += 4
            else: self.phase = 2
        elif self.phase == 2:
            if ok: self.c += 1
            else: self.phase = 3
        elif self.phase == 3:
            if ok: self.f += 4
            else: self.phase = 4
        elif self.phase == 4:
            if ok: self.f += 1
            else: self.phase = 5

    def copy(self):
        s = Search(self.c); s.f, s.phase, s.failed = self.f, self.phase, self.failed
        return s


def quad_root(pts, gl, gh, ge):
    """zero of the parabola through the last sweep's three costed points, inside (gl, gh); the line's estimate otherwise"""
    (x0, y0), (x1, y1), (x2, y2) = sorted(pts)
    if x0 == x1 or x1 == x2: return ge
    d1 = (y1 - y0) / (x1 - x0); d2 = (y2 - y1) / (x2 - x1)
    a = (d2 - d1) / (x2 - x0)
    b = d1 - a * (x0 + x1)
    c = y0 - x0 * (a * x0 + b)
    if abs(a) < 1e-9: return ge
    disc = b * b - 4 * a * c
    if disc < 0: return ge
    r = disc ** 0.5
    best = None
    for z in ((-b - r) / (2 * a), (-b + r) / (2 * a)):
        if gl <= z <= gh and (best is None or abs(z - ge) < abs(best - ge)): best = z
    return ge if best is None else best


def run(curve, start, policy, hint=None, tight=True, quad=False, qd=(2, 1), rnd=0.5, fill=True):
    """returns (sweeps, (csnr, fsnr)); hint: the source frame's offsets (a transcode's first sweep costs there); tight: the
    round-4 bounds from each costed offset's own ceilings (needs curve.extra), else the fixed +-72 bits of rounds 2-3; quad: the
    parabola probe (measured, not kept: longer tail); fill: probe candidates that clamp onto each other are replaced by the nearest
    uncosted neighbours (round 4, kept)"""
    extra = getattr(curve, "extra", None) if tight else None
    known = {}
    st = {"fit_hi": -1, "fail_lo": 1 << 20, "gl": None, "gh": None}
    ss = Search(start)
    sweeps = 0
    went_down = went_up = False
    first = True
    cold = start == 40
    probes_done = 0
    last = []

    def lookup(g):
        if g <= st["fit_hi"]: return True
        if g >= st["fail_lo"]: return False
        return known.get(g)

    def cost(gs):
        nonlocal sweeps
        sweeps += 1
        last.clear()
        for g in gs:
            sp = int(curve[g])
            known[g] = sp >= 0
            last.append((g, sp))
            if extra is None:
                if sp >= MARGIN and g > st["fit_hi"]: st["fit_hi"] = g
                if sp <= -MARGIN and g < st["fail_lo"]: st["fail_lo"] = g
            else:
                e6 = int(extra[g])
                if 6 * sp >= 414 - e6 and g > st["fit_hi"]: st["fit_hi"] = g
                if 6 * sp < -e6 and g < st["fail_lo"]: st["fail_lo"] = g
            if sp >= 0 and (st["gl"] is None or g > st["gl"][0]): st["gl"] = (g, sp)
            if sp < 0 and (st["gh"] is None or g < st["gh"][0]): st["gh"] = (g, sp)

    while True:
        q = None
        while True:
            q = ss.next()
            if q is None: break
            v = lookup(16 * q[0] + q[1])
            if v is None: break
            if ss.phase == 0 and not v: went_down = True
            if ss.phase == 1 and v: went_up = True
            ss.consume(v)
        if q is None: break
        cand = []
        if policy == "probe" and probes_done < 3 and st["gl"] is not None and st["gh"] is not None:
            gl, sl = st["gl"]; gh, sh = st["gh"]
            if gh > gl + 3:                     # as encode.hip: the line through the two nearest costed points, +- a step
                w = gh - gl
                ge = gl + int(w * (sl / (sl - sh)) + 0.5)
                d = max(2, (w * 85) >> 10) if w > 48 else 1
                if quad and len(last) == 3:
                    z = quad_root(last, gl, gh, gl + w * (sl / (sl - sh)))
                    ge = int(z + rnd)
                    d = qd[0] if w > 48 else qd[1]
                for g in (ge, ge + d, ge - d):
                    g = min(max(g, gl + 1), gh - 1)
                    if g not in cand: cand.append(g)
                k = 1
                while fill and len(cand) < NC and k < w:      # candidates that clamped onto each other: the nearest uncosted neighbours instead
                    for g in (ge - k, ge + k):
                        if gl < g < gh and g not in cand and len(cand) < NC: cand.append(g)
                    k += 1
                probes_done += 1
        if policy == "probe" and not cand and first and cold and hint is not None:
            g0 = min(max(hint, 8), 1000)
            cand = [g0 - 8, g0 + 2, g0 + 12]
        if policy == "probe" and not cand and first and cold:
            cand = [16 * 8, 16 * 13, 16 * 20]
        if not cand and ss.phase == 0 and (went_down or (first and cold)):
            n = 0
            c = ss.c
            while c >= 0 and n < 16:
                if lookup(16 * c) is not None: break
                c -= 4; n += 1
            if n > 3:
                for idx in ((n - 1) // 4, (n - 1) // 2, (3 * (n - 1) + 2) // 4):
                    g = 16 * (ss.c - 4 * idx)
                    if g not in cand: cand.append(g)
        first = False
        if not cand:
            ahead = ss.copy()
            while len(cand) < NC:
                q2 = ahead.next()
                if q2 is None: break
                g = 16 * q2[0] + q2[1]
                v = lookup(g)
                if v is None:
                    if g in cand: break
                    cand.append(g)
                    v = (not went_down) if ahead.phase == 0 else went_up if ahead.phase == 1 else True
                ahead.consume(v)
        cost(cand[:NC])
    return sweeps, (ss.c, ss.f)


def reference(curve, start):
    ss = Search(start)
    while True:
        q = ss.next()
        if q is None: break
        ss.consume(curve[16 * q[0] + q[1]] >= 0)
    return ss.c, ss.f


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    for name, sg in (("bench PCM", False), ("decoded audio fed back", True)):
        cs = curves(n, second_gen=sg)
        for policy in ("ladder", "probe"):
            for start in ("cold", "warm"):
                tot = 0
                for c in cs:
                    want = reference(c, 40)
                    s0 = 40 if start == "cold" else want[0]
                    sw, got = run(c, s0, policy)
                    assert got == reference(c, s0), (got, reference(c, s0))
                    tot += sw
                old = sum(run(c, 40 if start == "cold" else reference(c, 40)[0], policy, tight=False)[0] for c in cs)
                print("%-24s %-7s %-5s sweeps per frame %.2f (with the fixed +-72-bit bounds: %.2f)" % (name, policy, start, tot / len(cs), old / len(cs)))
    pairs = curves(n, seed=7, second_gen=True, with_source=True)
    tot = 0
    for c, g_src in pairs:
        sw, got = run(c, 40, "probe", hint=g_src)
        assert got == reference(c, 40)
        tot += sw
    d = [16 * reference(c, 40)[0] + reference(c, 40)[1] - g for c, g in pairs]
    print("decoded audio fed back, first sweep around the source frame's offsets (-8, +2, +12): %.2f sweeps per frame; "
          "the re-encode lands %d .. %d steps above its source" % (tot / len(pairs), min(d), max(d)))
