"""CPU, build container only: the oracle against the REAL liba52 (oracle/_ref/liba52_ref.so) on
fresh seeded inputs - broader than the committed fixtures.  Skipped where /root/reference was not
available to build _ref (the prebuilt binary does travel to the GPU box)."""
import ctypes

import numpy as np
import pytest

from tests import _harness as H

pytestmark = pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref/liba52_ref.so not built")


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("kind,nch,bitrate", [("tones", 6, 384000), ("noise", 6, 448000), ("music", 6, 384000),
                                              ("tones", 2, 192000), ("noise", 1, 96000), ("music", 5, 320000),
                                              ("tones", 4, 256000), ("music", 3, 192000)])
def test_decode_matches_liba52_bit_for_bit(kind, nch, bitrate):
    pcm = H.gen_pcm(10, nch, seed=nch * 7 + 1, kind=kind)
    chmap = H.CHMAP6 if nch == 6 else tuple(range(8))
    frames = H.orc_encode(pcm, nch=nch, bitrate=bitrate, chmap=chmap)
    for flags, level, bias in ((7 | 16, 1.0, 0.0), (2, 1.0, 0.0), (2 | 32, 0.7, 384.0), (10 | 32, 1.0, 0.0),
                               (1, 1.0, 0.0), (6 | 16 | 32, 1.0, 0.0), (3 | 32, 1.0, 0.0)):
        a, ea, fa = H.ref_decode(frames, flags, level, bias)
        b, eb, fb = H.orc_decode(frames, flags, level, bias)
        assert ea == 0 and eb == 0 and fa == fb
        assert np.array_equal(_bits(a), _bits(b)), (kind, nch, flags)


def test_dynrng_off_matches():
    frames = H.orc_encode(H.gen_pcm(4, 6, seed=3, kind="tones"))
    a, _, _ = H.ref_decode(frames, 2, 1.0, 0.0, dynrng_off=True)
    b, _, _ = H.orc_decode(frames, 2, 1.0, 0.0, dynrng_off=True)
    assert np.array_equal(_bits(a), _bits(b))


def test_corrupt_frames_same_return_codes():
    """Flip bits: a52_syncinfo / a52_frame / a52_block must agree on every return code and, where both
    succeed, on the output."""
    frames = H.orc_encode(H.gen_pcm(6, 6, seed=5, kind="tones"))
    rng = np.random.default_rng(0)
    R, L = H.ref(), H.orc()
    for trial in range(60):
        bad = frames.copy()
        f = trial % 6
        for _ in range(3):
            pos = rng.integers(5 * 8, 400 if trial % 2 else 1536 * 8)
            bad[f, pos >> 3] ^= 0x80 >> (pos & 7)
        # every frame in its own zero-padded buffer: a parse that runs off the end of a corrupt frame
        # then reads zeros in both decoders (liba52 reads whatever follows in memory, the oracle keeps a
        # padded private copy of the frame)
        buf = np.zeros((6, 4096), np.uint8)
        buf[:, :1536] = bad
        sr, so = R.a52_init(0), L.orc_a52_init()
        for k in range(6):
            p = ctypes.cast(buf.ctypes.data + k * 4096, H.u8p)
            f1, f2, x, y, z = H.ci(), H.ci(), H.ci(), H.ci(), H.ci()
            n1 = R.a52_syncinfo(p, ctypes.byref(f1), ctypes.byref(x), ctypes.byref(y))
            n2 = L.orc_a52_syncinfo(p, ctypes.byref(f2), ctypes.byref(z), ctypes.byref(y))
            assert n1 == n2
            if n1 != 1536 or (f1.value & 7) != 7:
                continue                                  # sizes/modes this harness does not follow
            fl1, fl2, lv1, lv2 = H.ci(23), H.ci(23), H.cf(1.0), H.cf(1.0)
            assert R.a52_frame(sr, p, ctypes.byref(fl1), ctypes.byref(lv1), 0.0) == \
                L.orc_a52_frame(so, p, ctypes.byref(fl2), ctypes.byref(lv2), 0.0)
            for b in range(6):
                r1, r2 = R.a52_block(sr), L.orc_a52_block(so)
                if R.refglue_bitpos(sr, p) > 1536 * 8:
                    break           # the parse ran off the end of the frame: liba52 now reads foreign memory
                assert r1 == r2, (trial, k, b)
                assert R.refglue_bitpos(sr, p) == L.orc_a52_bitpos(so)
                if r1:
                    break
        R.a52_free(sr)
        L.orc_a52_free(so)


def test_imdct_random_bit_exact():
    rng = np.random.default_rng(11)
    R, L = H.ref(), H.orc()
    d1 = (rng.standard_normal(256) * 0.2).astype(np.float32)
    d2 = d1.copy()
    for i in range(200):
        x = (rng.standard_normal(256) * 0.2).astype(np.float32)
        a, b = x.copy(), x.copy()
        short = rng.random() < 0.4
        bias = float(rng.choice([0.0, 384.0, 0.5]))
        (R.a52_imdct_256 if short else R.a52_imdct_512)(H.P(a, H.fp), H.P(d1, H.fp), bias)
        (L.orc_imdct_256 if short else L.orc_imdct_512)(H.P(b, H.fp), H.P(d2, H.fp), bias)
        assert np.array_equal(_bits(a), _bits(b)) and np.array_equal(_bits(d1), _bits(d2))


@pytest.mark.parametrize("acmod,lfe,seed", [(7, 1, 31), (2, 0, 39597), (3, 1, 33), (0, 0, 34)])
def test_damaged_packer_frames_same_samples(acmod, lfe, seed):
    """The damaged frames of tests/fuzz_corrupt.py (packer frames with coupling, rematrixing, block switching, delta bit
    allocation; random bit flips and bursts) through the real liba52 and the oracle: same return codes and, after every
    block both accept, bit-identical sample planes.  This pins the oracle on frames that decode garbage side information
    One thing is left out: when damage moves the coupling region above the endmant of a channel that reuses its
    exponents, liba52 writes neither coefficients nor zeros to the bins in between (L52/parse.c:813-835) and transforms
    what its in-place buffer holds from the block before; neither the oracle nor the engine keep liba52's buffer history,
    so samples are not compared from such a block on (seed 39597 holds such a frame; return codes still are).  Nor
    from a block on that tells a channel to reuse (or gives the reserved code for) delta bit allocation values that no
    block has sent: a52_init does not clear liba52's state (malloc, L52/parse.c:59), so its bit allocation then runs on
    uninitialised deltba[] arrays, and even the number of mantissa bits consumed is arbitrary."""
    from tests import fuzz_corrupt
    kw = dict(fscod=1, bsid=8, frmsizecod=25) if seed == 39597 else {}
    frames, _, want_fail, want_foreign, _ = fuzz_corrupt.make_damaged(seed, acmod, lfe, **kw)
    R, L = H.ref(), H.orc()
    S, fb = frames.shape
    flags = acmod | (16 if lfe else 0)
    nplanes = H.NFCHANS[acmod] + lfe
    compared = 0
    lay = (ctypes.c_int * 8)()
    L.orc_a52_get_layout.argtypes = [H.vp, ctypes.POINTER(ctypes.c_int)]
    L.orc_a52_get_layout.restype = None
    R.refglue_bitpos.restype = ctypes.c_long
    dba = (ctypes.c_int * 6)()
    L.orc_a52_get_deltbae.argtypes = [H.vp, ctypes.POINTER(ctypes.c_int)]
    L.orc_a52_get_deltbae.restype = None
    for s in range(S):
        buf = np.zeros(fb + 4096, np.uint8)              # zero padding: a parse that runs off the end reads zeros in both
        buf[:fb] = frames[s]
        p = H.P(buf, H.u8p)
        sr, so = R.a52_init(0), L.orc_a52_init()
        f1, f2, lv1, lv2 = H.ci(flags), H.ci(flags), H.cf(1.0), H.cf(1.0)
        r1, r2 = R.a52_frame(sr, p, ctypes.byref(f1), ctypes.byref(lv1), 0.0), L.orc_a52_frame(so, p, ctypes.byref(f2), ctypes.byref(lv2), 0.0)
        assert r1 == r2 and f1.value == f2.value
        if r1 == 0:
            stale = False
            sent = [False] * 6                           # deltba values sent so far (5 fbw channels, coupling channel)
            for b in range(6):
                r1, r2 = R.a52_block(sr), L.orc_a52_block(so)
                L.orc_a52_get_deltbae(so, dba)
                L.orc_a52_get_layout(so, lay)
                used = [c < H.NFCHANS[acmod] for c in range(5)] + [lay[7] != 0]
                if any(used[c] and dba[c] in (0, 3) and not sent[c] for c in range(6)):
                    break                                # liba52 allocates from uninitialised memory from here on
                sent = [sent[c] or dba[c] == 1 for c in range(6)]
                if R.refglue_bitpos(sr, p) > fb * 8:
                    break                                # liba52 is reading the padding now
                assert r1 == r2, (s, b)
                if r1:
                    assert b == want_fail[s]
                    break
                L.orc_a52_get_layout(so, lay)
                stale = stale or any((lay[7] >> c) & 1 and lay[c] < lay[5] for c in range(H.NFCHANS[acmod]))
                if stale:
                    continue
                a = np.ctypeslib.as_array(R.a52_samples(sr), (1536,))[:nplanes * 256]
                o = np.ctypeslib.as_array(L.orc_a52_samples(so), (1536,))[:nplanes * 256]
                assert np.array_equal(a.view(np.uint32), o.view(np.uint32)), (s, b)
                compared += 1
        R.a52_free(sr)
        L.orc_a52_free(so)
    assert compared > 200
