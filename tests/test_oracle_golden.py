"""CPU: pin the oracle (our liba52 / ac3enc restatement) to the committed golden vectors.

tests/golden/decode_*.npz, imdct.npz and downmix.npz hold outputs of the REAL liba52 (built from
/root/reference by oracle/Makefile in the build container; generator: tests/golden/make_golden.py).
The decode oracle must reproduce them BIT FOR BIT.  encoder.npz is a regression pin of our own
encoder oracle only (parity with ac3enc is unpinned: it cannot be compiled in this image).
"""
import ctypes
import os

import numpy as np
import pytest

from tests import _harness as H

G = H.GOLDEN


def _load(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("kind", ["tones", "noise", "quiet"])
@pytest.mark.parametrize("tag", ["51", "stereo", "dolby_adj", "51_bias384"])
def test_decode_bit_exact(kind, tag):
    d = _load("decode_%s.npz" % kind)
    flags, level, bias, oflags, lfsr = d["args_" + tag]
    pcm, errs, outflags = H.orc_decode(d["frames"], int(flags), float(level), float(bias))
    assert errs == 0 and outflags == int(oflags)
    assert np.array_equal(_bits(pcm), _bits(d["pcm_" + tag]))


@pytest.mark.parametrize("kind", ["tones", "noise", "quiet"])
def test_decode_stage_taps(kind):
    """exponents (D6), bit allocation (D8) and the dither generator state (D9) after every block."""
    d = _load("decode_%s.npz" % kind)
    L = H.orc()
    frames = d["frames"]
    st = L.orc_a52_init()
    buf = np.zeros(frames.size + 64, np.uint8)
    buf[:frames.size] = frames.reshape(-1)
    e, b = np.zeros(256, np.uint8), np.zeros(256, np.int8)
    for f in range(frames.shape[0]):
        fl, lv = H.ci(23), H.cf(1.0)
        p = ctypes.cast(buf.ctypes.data + f * frames.shape[1], H.u8p)
        assert L.orc_a52_frame(st, p, ctypes.byref(fl), ctypes.byref(lv), 0.0) == 0
        for blk in range(6):
            assert L.orc_a52_block(st) == 0
            for w in range(6):
                n = 223 if w < 5 else 7
                L.orc_a52_get_exp(st, w, H.P(e, H.u8p))
                L.orc_a52_get_bap(st, w, H.P(b, H.i8p))
                assert np.array_equal(e[:n], d["exp"][f, blk, w, :n])
                assert np.array_equal(b[:n], d["bap"][f, blk, w, :n])
    assert L.orc_a52_get_lfsr(st) == int(d["args_51"][4])
    L.orc_a52_free(st)


def test_quiet_stream_is_all_dither():
    """digital silence -> every coefficient has bap 0 and comes from the dither LFSR (parse.c:310-319)"""
    d = _load("decode_quiet.npz")
    assert (d["bap"][:, :, :5, :223] == 0).mean() > 0.95
    assert np.abs(d["pcm_51"]).max() > 0          # dither noise, not zeros


def test_s16_conversion_matches_libao():
    """D18 on in-range values: our converter (AC3ASM.asm semantics, WAVE order) vs libao's convert2s16_multi
    (convert2s16.c:113-181), which orders 5.1 as L,R,C(?),...: compare as per-channel sets via the plane map."""
    d = _load("decode_tones.npz")
    L = H.orc()
    pcm = d["pcm_51_bias384"]
    out = np.zeros((256, 6), np.int16)
    for f in range(pcm.shape[0]):
        for b in range(6):
            L.orc_convert_s16(H.P(np.ascontiguousarray(pcm[f, b]), H.fp), H.P(out, H.i16p), 7 | 16)
            # plane p of liba52 (LFE,L,C,R,SL,SR) lands in WAVE slot (FL,FR,FC,LFE,BL,BR)
            for plane, slot in ((0, 3), (1, 0), (2, 2), (3, 1), (4, 4), (5, 5)):
                want = (pcm[f, b, plane].view(np.int32) - 0x43c00000).clip(-32768, 32767).astype(np.int16)
                assert np.array_equal(out[:, slot], want)
            # libao writes the same sample values (its own channel order: convert2s16.c:157-166)
            assert sorted(out.reshape(-1).tolist()) == sorted(d["s16_multi_51_bias384"][f, b].reshape(-1).tolist())


def test_imdct_vectors_bit_exact():
    d = _load("imdct.npz")
    L = H.orc()
    for i in range(d["x"].shape[0]):
        x, dl = d["x"][i].copy(), d["delay_in"][i].copy()
        (L.orc_imdct_256 if d["kind"][i] else L.orc_imdct_512)(H.P(x, H.fp), H.P(dl, H.fp), float(d["bias"][i]))
        assert np.array_equal(_bits(x), _bits(d["y"][i])) and np.array_equal(_bits(dl), _bits(d["delay_out"][i]))
    dl = np.zeros(256, np.float32)
    for i in range(d["seq_x"].shape[0]):
        x = d["seq_x"][i].copy()
        (L.orc_imdct_256 if d["seq_kind"][i] else L.orc_imdct_512)(H.P(x, H.fp), H.P(dl, H.fp), 0.0)
        assert np.array_equal(_bits(x), _bits(d["seq_y"][i]))
    assert np.array_equal(_bits(dl), _bits(d["seq_delay"]))
    # only the first half of the delay plane is live
    assert np.array_equal(d["delay_out"][:, 128:], d["delay_in"][:, 128:])


def test_downmix_vectors_bit_exact():
    d = _load("downmix.npz")
    L = H.orc()
    for k, (acmod, flags, clev, slev) in enumerate(d["cases"]):
        acmod, flags = int(acmod), int(flags)
        lv = H.cf(1.0)
        out = L.orc_downmix_init(acmod, flags, ctypes.byref(lv), np.float32(clev), np.float32(slev))
        assert out == int(d["init"][k, 0])
        assert np.float32(lv.value) == np.float32(d["init"][k, 1])
        g = np.zeros(5, np.float32)
        mask = L.orc_downmix_coeff(H.P(g, H.fp), acmod, out, lv.value, np.float32(clev), np.float32(slev))
        n = 1 if (acmod, out) == (1, 10) else H.NFCHANS[acmod]
        assert mask == int(d["coeff"][k, 5])
        assert np.array_equal(_bits(g[:n]), _bits(d["coeff"][k, :n]))
        p = d["planes"].copy()
        L.orc_downmix(H.P(p, H.fp), acmod, out, 0.25, np.float32(clev), np.float32(slev))
        assert np.array_equal(_bits(p), _bits(d["mixed"][k])), (acmod, out)
        u = d["planes"].copy()
        L.orc_upmix(H.P(u, H.fp), acmod, out)
        assert np.array_equal(_bits(u), _bits(d["upmixed"][k])), (acmod, out)


def test_encoder_oracle_regression():
    """Regression pin of our encoder oracle (bitstream + every stage).  NOT a parity claim against
    ac3enc: that is unpinned."""
    d = _load("encoder.npz")
    L = H.orc()
    fb = H.ci()
    h = L.orc_ac3enc_init(48000, 384000, 6, ctypes.byref(fb))
    assert fb.value == 1536
    cm = (ctypes.c_uint8 * 8)(*H.CHMAP6)
    out = np.zeros(1536, np.uint8)
    pcm = np.ascontiguousarray(d["pcm_in"])
    for f in range(3):
        assert L.orc_ac3enc_frame(h, H.P(out, H.u8p), ctypes.cast(pcm.ctypes.data + f * 1536 * 12, H.i16p), cm) == 1536
        assert np.array_equal(out, d["frames"][f])
        m = np.zeros((6, 6, 256), np.int32)
        L.orc_ac3enc_get_mdct(h, H.P(m, H.i32p))
        assert np.array_equal(m, d["mdct"][f])
        ba = np.zeros((6, 6, 256), np.uint8)
        L.orc_ac3enc_get_bap(h, H.P(ba, H.u8p))
        assert np.array_equal(ba[:, :5, :223], d["bap"][f][:, :5, :223])
    L.orc_ac3enc_free(h)
    cos, sin, xc, xs = (np.zeros(n, np.int16) for n in (64, 64, 128, 128))
    crc = np.zeros(256, np.uint16)
    L.orc_ac3enc_tables(H.P(cos, H.i16p), H.P(sin, H.i16p), H.P(xc, H.i16p), H.P(xs, H.i16p), H.P(crc, H.u16p))
    for got, key in ((cos, "costab"), (sin, "sintab"), (xc, "xcos1"), (xs, "xsin1"), (crc, "crc_table")):
        assert np.array_equal(got, d[key]), key


def test_encoder_rejects_bad_parameters():
    """AC3_encode_init returns 0 for unsupported rate / bitrate / channel count (ac3enc.cpp:1039,1060,1070)."""
    L = H.orc()
    fb = H.ci()
    for args in ((48000, 384000, 0), (48000, 384000, 7), (47999, 384000, 6), (48000, 383000, 6)):
        assert not L.orc_ac3enc_init(*args, ctypes.byref(fb)) and fb.value == 0
    for freq, br, ch, want in ((48000, 384000, 6, 1536), (48000, 192000, 2, 768), (32000, 128000, 2, 768),
                               (24000, 64000, 1, 512)):
        h = L.orc_ac3enc_init(freq, br, ch, ctypes.byref(fb))
        assert h and fb.value == want, (freq, br, ch, fb.value)
        L.orc_ac3enc_free(h)


# ---- encoder spec tables: pinned to the reference's own header ----------------------------------------------------
# tests/golden/ac3tab.npz is frozen from src/ac3enc/ac3tab.h:3-171, compiled unmodified behind
# oracle/ref_ac3tab_glue.cpp (oracle/Makefile, `make_golden.py --only ac3tab`).  Both the encoder oracle and the
# engine must use exactly these numbers.  What stays PARITY UNPINNED is the code of ac3enc.cpp itself.

_SPEC_ORDER = ("ac3_window", "latab", "hth", "baptab", "bndsz", "sdecaytab", "fdecaytab", "sgaintab", "dbkneetab",
               "floortab", "fgaintab", "ac3_freqs", "ac3_bitratetab")


def _spec_buffers():
    return {"ac3_window": np.zeros(256, np.int16), "latab": np.zeros(256, np.uint8), "hth": np.zeros((50, 3), np.uint16),
            "baptab": np.zeros(64, np.uint8), "bndsz": np.zeros(50, np.uint8), "sdecaytab": np.zeros(4, np.uint16),
            "fdecaytab": np.zeros(4, np.uint16), "sgaintab": np.zeros(4, np.uint16), "dbkneetab": np.zeros(4, np.uint16),
            "floortab": np.zeros(8, np.uint16), "fgaintab": np.zeros(8, np.uint16), "ac3_freqs": np.zeros(3, np.uint16),
            "ac3_bitratetab": np.zeros(19, np.uint16)}


def _check_spec_tables(got, who):
    ref = _load("ac3tab.npz")
    assert set(ref.files) == set(_SPEC_ORDER)
    for name in _SPEC_ORDER:
        want = ref[name].astype(np.int64)
        if name == "latab":
            # the reference pads latab to 260 entries with zeros (ac3tab.h:51-78); addresses are clamped to 255
            assert want.shape == (260,) and not want[256:].any()
            want = want[:256]
        assert np.array_equal(got[name].astype(np.int64), want), "%s: %s differs from src/ac3enc/ac3tab.h" % (who, name)


def test_encoder_oracle_spec_tables_equal_the_reference_header():
    L = H.orc()
    b = _spec_buffers()
    L.orc_ac3enc_spec_tables.restype = None
    L.orc_ac3enc_spec_tables.argtypes = [ctypes.c_void_p] * 13
    L.orc_ac3enc_spec_tables(*(b[n].ctypes.data for n in _SPEC_ORDER))
    _check_spec_tables(b, "oracle/ac3enc_oracle.c")


def test_engine_spec_tables_equal_the_reference_header():
    """ac3mi_encode_spec_tables returns the values the encoder kernels are built with (host call, no GPU needed)."""
    lib = H.pkg().load_library()
    b = _spec_buffers()
    lib.ac3mi_encode_spec_tables.argtypes = [ctypes.c_void_p] * 13
    assert lib.ac3mi_encode_spec_tables(*(b[n].ctypes.data for n in _SPEC_ORDER)) == 0
    _check_spec_tables(b, "libac3mi.so")


def test_engine_window_equals_oracle_window_and_decoder_tables_agree():
    """The decoder's bit-allocation tables are the encoder's in liba52's sign convention (spec_tables.h): cross-check
    the two forms against the same fixture (hth: 0xc00 - value; latab: negated) through the oracle's decoder, which is
    itself pinned to the real liba52."""
    ref = _load("ac3tab.npz")
    lib = H.pkg().load_library()
    win = np.zeros(256, np.int16)
    lib.ac3mi_encode_tables.argtypes = [ctypes.c_void_p] * 5
    assert lib.ac3mi_encode_tables(None, None, None, None, win.ctypes.data) == 0
    assert np.array_equal(win, ref["ac3_window"])
