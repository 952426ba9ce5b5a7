#!/usr/bin/env python3
"""Generates tests/golden/*.npz.  Runs ONLY in the build container, where oracle/_ref/liba52_ref.so
(the real liba52, compiled from /root/reference by oracle/Makefile) exists.

What is pinned and by what:
  decode_*.npz   bitstreams (made by OUR encoder oracle - ac3enc itself cannot be built here) and what
                 the REAL liba52 decodes from them: float PCM for several output modes, exponents, bap
  imdct.npz      random coefficient planes + delay -> REAL a52_imdct_512 / a52_imdct_256 output
  downmix.npz    REAL a52_downmix_init / a52_downmix_coeff / a52_downmix / a52_upmix results
  encoder.npz    our encoder oracle's own output and stage dumps (regression pin only: PARITY UNPINNED
                 against ac3enc, see oracle/ac3enc_oracle.c)
  ac3tab.npz     the REAL encoder's constant tables (src/ac3enc/ac3tab.h:3-171 compiled unmodified behind
                 oracle/ref_ac3tab_glue.cpp): ac3_window, latab, hth, baptab, sdecaytab ... fgaintab, bndsz,
                 ac3_freqs, ac3_bitratetab   (`make_golden.py --only ac3tab` regenerates just this file)
Fixtures are data (inputs + expected outputs); no reference source text is stored.
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import _harness as H  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def ref_decode_with_taps(frames, flags, level, bias):
    L = H.ref()
    F, fb = frames.shape
    st = L.a52_init(0)
    buf = np.zeros(F * fb + 64, np.uint8)
    buf[:F * fb] = frames.reshape(-1)
    exps = np.zeros((F, 6, 6, 256), np.uint8)
    baps = np.zeros((F, 6, 6, 256), np.int8)
    pcm = None
    for f in range(F):
        fl, lv = H.ci(flags), H.cf(level)
        p = ctypes.cast(buf.ctypes.data + f * fb, H.u8p)
        assert L.a52_frame(st, p, ctypes.byref(fl), ctypes.byref(lv), bias) == 0
        nout = H.NFCHANS[fl.value & 15] + (1 if fl.value & 16 else 0)
        if pcm is None:
            pcm = np.zeros((F, 6, nout, 256), np.float32)
        for b in range(6):
            assert L.a52_block(st) == 0
            pcm[f, b] = np.ctypeslib.as_array(L.a52_samples(st), (1536,))[:nout * 256].reshape(nout, 256)
            for w in range(6):
                L.refglue_get_exp(st, w, H.P(exps[f, b, w], H.u8p))
                L.refglue_get_bap(st, w, H.P(baps[f, b, w], H.i8p))
    lfsr = L.refglue_get_lfsr(st)
    L.a52_free(st)
    return pcm, exps, baps, lfsr, fl.value


AC3TAB_NAMES = ("ac3_freqs", "ac3_bitratetab", "ac3_window", "latab", "hth", "baptab", "sdecaytab", "fdecaytab",
                "sgaintab", "dbkneetab", "floortab", "fgaintab", "bndsz")


def make_ac3tab():
    path = os.path.join(ROOT, "oracle", "_ref", "ac3tab_ref.so")
    assert os.path.exists(path), "oracle/_ref/ac3tab_ref.so missing: run `make -C oracle` in the build container"
    T = ctypes.CDLL(path)
    T.refglue_ac3tab.restype = ctypes.c_void_p
    T.refglue_ac3tab.argtypes = [ctypes.c_char_p, H.ip, H.ip]
    d = {}
    for name in AC3TAB_NAMES:
        n, eb = H.ci(), H.ci()
        ptr = T.refglue_ac3tab(name.encode(), ctypes.byref(n), ctypes.byref(eb))
        assert ptr and n.value > 0, name
        signed = name == "ac3_window"                       # the only signed table (short); the rest are unsigned
        dt = {1: np.uint8, 2: np.int16 if signed else np.uint16}[eb.value]
        d[name] = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(np.ctypeslib.as_ctypes_type(dt))), (n.value,)).copy()
    d["hth"] = d["hth"].reshape(50, 3)
    np.savez_compressed(os.path.join(OUT, "ac3tab.npz"), **d)
    print("ac3tab.npz", os.path.getsize(os.path.join(OUT, "ac3tab.npz")), {k: v.shape for k, v in d.items()})


MIXFLIP = (("a7_st", 7, 2, (1, 1, 2, 2, 0, 0, 2, 1), 0.0), ("a7_mono", 7, 1, (0, 2, 2, 1, 2, 0), 0.0), ("a7_dolby", 7, 10, (3, 2, 0, 2, 2, 1), 0.0),
           ("a6_st", 6, 2, (1, 2, 2, 0, 2, 1), 0.0), ("a5_st", 5, 2, (0, 2, 1, 2, 2, 3), 0.0), ("a4_mono", 4, 1, (1, 2, 0, 2, 2, 1), 0.0),
           # at bias 384 (what the ACM driver decodes at): 2/x -> stereo and 3/x -> 3F lose the bias in blocks of mixed block sizes
           ("a6_st_b384", 6, 2, (1, 2, 2, 0, 2, 1), 384.0), ("a7_3f_b384", 7, 3, (2, 2, 1, 2, 2, 0), 384.0), ("a4_st_b384", 4, 2, (2, 2, 2, 3, 2, 2), 384.0))


def make_mixflip():
    """Streams whose surmixlev changes between frames (tests/packer.make_flip_stream), decoded by the real liba52."""
    from tests import packer
    assert H.have_ref()
    d = {}
    for tag, acmod, flags, levels, bias in MIXFLIP:
        fr = packer.make_flip_stream(2600 + acmod + flags, levels, acmod=acmod)
        pcm, errs, oflags = H.ref_decode(fr, flags, 1.0, bias)
        assert errs == 0
        d["frames_" + tag] = fr
        d["pcm_" + tag] = pcm
        d["args_" + tag] = np.array([flags, oflags], np.int32)
        d["levels_" + tag] = np.array(levels, np.int32)
        d["bias_" + tag] = np.array([bias], np.float32)
    np.savez_compressed(os.path.join(OUT, "mixflip.npz"), **d)
    print("mixflip.npz", os.path.getsize(os.path.join(OUT, "mixflip.npz")))


def main():
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "mixflip":
        make_mixflip()
        return
    if "--only" in sys.argv and sys.argv[sys.argv.index("--only") + 1] == "ac3tab":
        make_ac3tab()
        return
    make_ac3tab()
    assert H.have_ref(), "oracle/_ref/liba52_ref.so missing: run `make -C oracle` in the build container"
    R = H.ref()
    # ---- decode fixtures -------------------------------------------------
    for kind, nfr in (("tones", 4), ("noise", 2), ("quiet", 2)):
        pcm_in = H.gen_pcm(nfr, 6, seed=2024, kind=kind)
        frames = H.orc_encode(pcm_in)
        d = {"pcm_in": pcm_in, "frames": frames}
        for tag, flags, level, bias in (("51", 7 | 16, 1.0, 0.0), ("stereo", 2, 1.0, 0.0),
                                         ("dolby_adj", 10 | 32, 1.0, 0.0), ("51_bias384", 7 | 16 | 32, 1.0, 384.0)):
            pcm, exps, baps, lfsr, oflags = ref_decode_with_taps(frames, flags, level, bias)
            d["pcm_" + tag] = pcm
            d["args_" + tag] = np.array([flags, level, bias, oflags, lfsr], np.float64)
            if tag == "51":
                d["exp"], d["bap"] = exps, baps
            if tag == "51_bias384":
                s16 = np.zeros((nfr, 6, 256, 6), np.int16)
                for f in range(nfr):
                    for b in range(6):
                        R.convert2s16_multi(H.P(np.ascontiguousarray(pcm[f, b]), H.fp), H.P(s16[f, b], H.i16p), oflags)
                d["s16_multi_" + tag] = s16      # libao channel order (convert2s16.c:113-181), in-range values
        np.savez_compressed(os.path.join(OUT, "decode_%s.npz" % kind), **d)

    # ---- transform-only fixtures ----------------------------------------
    rng = np.random.default_rng(99)
    x = (rng.standard_normal((8, 256)) * 0.1).astype(np.float32)
    dl = (rng.standard_normal((8, 256)) * 0.1).astype(np.float32)
    kinds = np.array([0, 0, 1, 1, 0, 1, 0, 1], np.uint8)        # 0: imdct_512, 1: imdct_256
    biases = np.array([0, 384, 0, 384, 0.5, 0, -1, 0], np.float32)
    y, dn = x.copy(), dl.copy()
    for i in range(8):
        (R.a52_imdct_256 if kinds[i] else R.a52_imdct_512)(H.P(y[i], H.fp), H.P(dn[i], H.fp), float(biases[i]))
    # a chained sequence long/short/long through one delay plane
    seq_x = (rng.standard_normal((6, 256)) * 0.1).astype(np.float32)
    seq_k = np.array([0, 1, 1, 0, 1, 0], np.uint8)
    seq_d = np.zeros(256, np.float32)
    seq_y = seq_x.copy()
    for i in range(6):
        (R.a52_imdct_256 if seq_k[i] else R.a52_imdct_512)(H.P(seq_y[i], H.fp), H.P(seq_d, H.fp), 0.0)
    np.savez_compressed(os.path.join(OUT, "imdct.npz"), x=x, delay_in=dl, kind=kinds, bias=biases, y=y, delay_out=dn,
                        seq_x=seq_x, seq_kind=seq_k, seq_y=seq_y, seq_delay=seq_d)

    # ---- downmix fixtures ------------------------------------------------
    cases, res_init, res_coeff, res_mix, res_up = [], [], [], [], []
    planes0 = (rng.standard_normal((6, 256)) * 0.1).astype(np.float32)
    for acmod in range(8):
        for req in range(11):
            for adj in (0, 32):
                for clev, slev in ((0.7071067811865476, 0.7071067811865476), (0.5946035575013605, 0.5), (0.5, 0.0)):
                    lv = H.cf(1.0)
                    out = R.a52_downmix_init(acmod, req | adj, ctypes.byref(lv), np.float32(clev), np.float32(slev))
                    if out < 0:
                        continue
                    g = np.zeros(5, np.float32)
                    mask = R.a52_downmix_coeff(H.P(g, H.fp), acmod, out, lv.value, np.float32(clev), np.float32(slev))
                    p = planes0.copy()
                    R.a52_downmix(H.P(p, H.fp), acmod, out, 0.25, np.float32(clev), np.float32(slev))
                    u = planes0.copy()
                    R.a52_upmix(H.P(u, H.fp), acmod, out)
                    cases.append([acmod, req | adj, clev, slev])
                    res_init.append([out, lv.value])
                    n = H.NFCHANS[acmod]
                    gg = np.zeros(5, np.float32)
                    gg[:n] = g[:n]
                    if (acmod, out) == (1, 10):
                        gg[1:] = 0                      # only coeff[0] is defined for mono -> dolby
                    res_coeff.append(np.concatenate([gg, [mask]]))
                    res_mix.append(p)
                    res_up.append(u)
    np.savez_compressed(os.path.join(OUT, "downmix.npz"), planes=planes0, cases=np.array(cases, np.float64),
                        init=np.array(res_init, np.float64), coeff=np.array(res_coeff, np.float32),
                        mixed=np.array(res_mix, np.float32), upmixed=np.array(res_up, np.float32))

    # ---- encoder oracle regression pin (PARITY UNPINNED vs ac3enc) --------
    L = H.orc()
    pcm_in = H.gen_pcm(3, 6, seed=7, kind="tones")
    fb = H.ci()
    h = L.orc_ac3enc_init(48000, 384000, 6, ctypes.byref(fb))
    cm = (ctypes.c_uint8 * 8)(*H.CHMAP6)
    frames = np.zeros((3, fb.value), np.uint8)
    mdct = np.zeros((3, 6, 6, 256), np.int32)
    expo = np.zeros((3, 6, 6, 256), np.uint8)
    eexp = np.zeros((3, 6, 6, 256), np.uint8)
    bap = np.zeros((3, 6, 6, 256), np.uint8)
    strat = np.zeros((3, 6, 6), np.uint8)
    shift = np.zeros((3, 6, 6), np.int8)
    snr = np.zeros((3, 2), np.int32)
    for f in range(3):
        assert L.orc_ac3enc_frame(h, H.P(frames[f], H.u8p), ctypes.cast(pcm_in.ctypes.data + f * 1536 * 12, H.i16p), cm) == fb.value
        L.orc_ac3enc_get_mdct(h, H.P(mdct[f], H.i32p))
        L.orc_ac3enc_get_exp(h, H.P(expo[f], H.u8p), H.P(eexp[f], H.u8p))
        L.orc_ac3enc_get_bap(h, H.P(bap[f], H.u8p))
        c, fs = H.ci(), H.ci()
        L.orc_ac3enc_get_misc(h, H.P(strat[f], H.u8p), H.P(shift[f], H.i8p), ctypes.byref(c), ctypes.byref(fs))
        snr[f] = (c.value, fs.value)
    L.orc_ac3enc_free(h)
    cos, sin, xc, xs = (np.zeros(n, np.int16) for n in (64, 64, 128, 128))
    crc = np.zeros(256, np.uint16)
    L.orc_ac3enc_tables(H.P(cos, H.i16p), H.P(sin, H.i16p), H.P(xc, H.i16p), H.P(xs, H.i16p), H.P(crc, H.u16p))
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), pcm_in=pcm_in, frames=frames, mdct=mdct, exponent=expo,
                        encoded_exp=eexp, bap=bap, exp_strategy=strat, exp_samples=shift, snroffst=snr,
                        costab=cos, sintab=sin, xcos1=xc, xsin1=xs, crc_table=crc)
    # ---- packer streams: coupling, rematrix, delta bit allocation, dynrng, block switching ... --------
    from tests import packer
    d = {}
    for tag, acmod, lfe, fscod, bsid, fsz, flags in (("a7", 7, 1, 0, 8, 36, 7 | 16), ("a7_st", 7, 1, 0, 8, 36, 2 | 32),
                                                    ("a2", 2, 0, 0, 8, 30, 2), ("a2_mono", 2, 0, 1, 8, 31, 1),
                                                    ("a0", 0, 0, 2, 8, 28, 0), ("a5_half", 5, 1, 0, 9, 36, 5 | 16),
                                                    ("a3_dolby", 3, 0, 0, 10, 34, 10)):
        fr = packer.make_stream(4242 + acmod, 3, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz)
        pcm, errs, oflags = H.ref_decode(fr, flags, 1.0, 0.0)
        assert errs == 0
        d["frames_" + tag] = fr
        d["pcm_" + tag] = pcm
        d["args_" + tag] = np.array([flags, oflags], np.int32)
    np.savez_compressed(os.path.join(OUT, "packer.npz"), **d)
    make_mixflip()

    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".npz"):
            print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
