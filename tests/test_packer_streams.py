"""Streams from tests/packer.py exercise what the reference's own encoder never emits: coupling
(parse.c:600-667, coeff_get_coupling :435-556), rematrixing (:669-678, 837-865), block switching, delta bit
allocation (:272-294, 757-772), dynamic-range words (:578-598), skip fields, every acmod, three sample
rates and the half-rate bsids.

CPU part: the decode oracle equals the REAL liba52 on them bit for bit - against the committed fixture
tests/golden/packer.npz everywhere, and on a fresh sweep where oracle/_ref exists.
GPU part: the HIP decoder equals the oracle (integer stages and coefficient planes bit-exact, PCM <= 1e-6 RMS).
"""
import os

import numpy as np
import pytest

from tests import _harness as H
from tests import packer

TAGS = ["a7", "a7_st", "a2", "a2_mono", "a0", "a5_half", "a3_dolby"]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_reproduces_liba52_on_packer_fixture(tag):
    d = np.load(os.path.join(H.GOLDEN, "packer.npz"), allow_pickle=False)
    flags, oflags = (int(x) for x in d["args_" + tag])
    pcm, errs, out = H.orc_decode(d["frames_" + tag], flags, 1.0, 0.0)
    assert errs == 0 and out == oflags
    assert np.array_equal(_bits(pcm), _bits(d["pcm_" + tag]))


def test_packer_is_deterministic_and_uses_the_features():
    """Same seed -> same stream as the fixture; and the streams really contain coupling etc."""
    d = np.load(os.path.join(H.GOLDEN, "packer.npz"), allow_pickle=False)
    fr = packer.make_stream(4242 + 7, 3, 7, 1, fscod=0, bsid=8, frmsizecod=36)
    assert np.array_equal(fr, d["frames_a7"])
    import ctypes
    L = H.orc()
    seen_cpl = seen_short = 0
    for acmod in (2, 5, 7):
        fr = packer.make_stream(99 + acmod, 4, acmod, 1)
        st = L.orc_a52_init()
        L.orc_a52_get_coefs.argtypes = [H.vp, H.fp, H.u8p]
        buf = np.zeros(fr.size + 64, np.uint8)
        buf[:fr.size] = fr.reshape(-1)
        for f in range(4):
            fl, lv = H.ci(acmod | 16), H.cf(1.0)
            assert L.orc_a52_frame(st, ctypes.cast(buf.ctypes.data + f * fr.shape[1], H.u8p), ctypes.byref(fl),
                                   ctypes.byref(lv), 0.0) == 0
            for b in range(6):
                assert L.orc_a52_block(st) == 0
                bap = np.zeros(256, np.int8)
                L.orc_a52_get_bap(st, 6, H.P(bap, H.i8p))
                seen_cpl += int(np.any(bap != 0))
                c, sw = np.zeros((6, 256), np.float32), np.zeros(5, np.uint8)
                L.orc_a52_get_coefs(st, H.P(c, H.fp), H.P(sw, H.u8p))
                seen_short += int(sw.any())
        L.orc_a52_free(st)
    assert seen_cpl > 10 and seen_short > 10


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref/liba52_ref.so not built")
def test_oracle_matches_liba52_on_fresh_packer_streams():
    n = 0
    for acmod in range(8):
        for lfe in (0, 1):
            for fscod, bsid, fsz in ((0, 8, 36), (1, 8, 37), (2, 10, 30), (0, 9, 36)):
                fr = packer.make_stream(31337 + acmod * 2 + lfe + fscod * 100 + bsid, 3, acmod, lfe, fscod=fscod,
                                        bsid=bsid, frmsizecod=fsz)
                for flags in (acmod | 16, 2 | 32, 1, 10):
                    a, ea, fa = H.ref_decode(fr, flags, 1.0, 0.0)
                    b, eb, fb = H.orc_decode(fr, flags, 1.0, 0.0)
                    assert ea == 0 and eb == 0 and fa == fb
                    assert np.array_equal(_bits(a), _bits(b)), (acmod, lfe, fscod, bsid, flags)
                    n += 1
    assert n == 256


# ---------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("acmod,lfe,fscod,bsid,fsz", [(7, 1, 0, 8, 36), (7, 0, 1, 8, 37), (2, 0, 0, 8, 30), (2, 1, 2, 10, 30),
                                                     (0, 0, 0, 8, 28), (1, 1, 0, 8, 24), (3, 0, 0, 9, 36), (4, 1, 0, 8, 32),
                                                     (5, 1, 1, 8, 35), (6, 0, 0, 8, 34)])
def test_gpu_decoder_on_packer_streams(engine, acmod, lfe, fscod, bsid, fsz):
    import ctypes
    import torch
    pkg = H.pkg()
    S, F = 4, 3
    frames = np.stack([packer.make_stream(777 + 13 * s + acmod, F, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz)
                       for s in range(S)])
    fb = frames.shape[2]
    stride = (fb + 3) & ~3
    padded = np.zeros((S, F, stride), np.uint8)
    padded[:, :, :fb] = frames
    nf = H.NFCHANS[acmod]
    out_req = {7: [7 | 16, 2 | 32, 10, 1], 2: [2, 1 | 32], 0: [0, 1, 8, 9]}.get(acmod, [acmod | 16, 2, 1])
    L = H.orc()
    L.orc_a52_get_coefs.argtypes = [H.vp, H.fp, H.u8p]
    for flags in out_req:
        # ---- oracle, with taps ----
        want_pcm = None
        want_coef = np.zeros((S, F, 6, 6, 256), np.float32)
        want_sw = np.zeros((S, F, 6, 5), np.uint8)
        want_exp = np.zeros((S, F, 6, 7, 256), np.uint8)
        want_bap = np.zeros((S, F, 6, 7, 256), np.int8)
        want_lfsr = np.zeros(S, np.int64)
        want_flags = np.zeros((S, F), np.int64)      # per frame: dsurmod can turn STEREO into DOLBY
        for s in range(S):
            st = L.orc_a52_init()
            buf = np.zeros(F * fb + 64, np.uint8)
            buf[:F * fb] = frames[s].reshape(-1)
            for f in range(F):
                fl, lv = H.ci(flags), H.cf(1.0)
                assert L.orc_a52_frame(st, ctypes.cast(buf.ctypes.data + f * fb, H.u8p), ctypes.byref(fl), ctypes.byref(lv), 0.0) == 0
                nout = H.NFCHANS[fl.value & 15] + (1 if fl.value & 16 else 0)
                want_flags[s, f] = fl.value
                if want_pcm is None:
                    want_pcm = np.zeros((S, F, 6, nout, 256), np.float32)
                for b in range(6):
                    assert L.orc_a52_block(st) == 0
                    want_pcm[s, f, b] = np.ctypeslib.as_array(L.orc_a52_samples(st), (1536,))[:nout * 256].reshape(nout, 256)
                    L.orc_a52_get_coefs(st, H.P(want_coef[s, f, b], H.fp), H.P(want_sw[s, f, b], H.u8p))
                    for w in range(7):
                        L.orc_a52_get_exp(st, w, H.P(want_exp[s, f, b, w], H.u8p))
                        L.orc_a52_get_bap(st, w, H.P(want_bap[s, f, b, w], H.i8p))
            want_lfsr[s] = L.orc_a52_get_lfsr(st)
            oflags = fl.value
            L.orc_a52_free(st)
        # ---- GPU ----
        desc = pkg.DecodeDesc(flags=flags, level=1.0, bias=0.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
        n_out, _ = engine.decode_planes(desc)
        delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
        lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
        pcm, status, taps = engine.decode_batch(desc, torch.from_numpy(padded).cuda(), delay, lfsr, taps=True)
        engine.sync()
        status = status.cpu().numpy()
        assert (status & 0x1ff).max() == 0, status
        assert np.array_equal((status >> 16) & 0xff, want_flags), (status >> 16, want_flags)
        assert np.array_equal(lfsr.cpu().numpy().astype(np.int64) & 0xffff, want_lfsr)
        assert np.array_equal(taps["blksw"].cpu().numpy()[..., :nf], want_sw[..., :nf])
        # exponents / bap only where the stream defines them: compare through the coefficient planes, which
        # depend on every exponent, bap, mantissa, dither draw, coupling coordinate and rematrix flag
        got = taps["coef"].cpu().numpy()
        lfe_out = 1 if oflags & 16 else 0
        assert np.array_equal(_bits(got[:, :, :, lfe:lfe + nf]), _bits(want_coef[:, :, :, lfe_out:lfe_out + nf])), \
            "fbw coefficient planes differ (acmod %d flags %d)" % (acmod, flags)
        if lfe_out:
            assert np.array_equal(_bits(got[:, :, :, 0]), _bits(want_coef[:, :, :, 0]))
        # random mantissas at random exponents/dynrng gains give |PCM| far above +-1.0 full scale: the 1e-6
        # bar is relative to full scale, so scale it by the signal level where that exceeds 1
        err = pcm.cpu().numpy().astype(np.float64) - want_pcm
        scale_rms, scale_max = max(1.0, H.rms(want_pcm)), max(1.0, float(np.abs(want_pcm).max()))
        assert H.rms(err) <= 1e-6 * scale_rms and np.abs(err).max() <= 1e-5 * scale_max, \
            (acmod, flags, H.rms(err), np.abs(err).max(), scale_rms, scale_max)


@pytest.mark.gpu
def test_gpu_decoder_frames_of_two_sizes_in_one_stream(engine):
    """44.1 kHz streams alternate between two frame sizes (frmsizecod 2k / 2k+1): every frame sits on its stride
    slot and carries its own size; the descriptor names the larger one."""
    import ctypes
    import torch
    pkg = H.pkg()
    S, F = 3, 4
    a = [packer.make_stream(50 + s, F, 2, 0, fscod=1, bsid=8, frmsizecod=28) for s in range(S)]
    b = [packer.make_stream(90 + s, F, 2, 0, fscod=1, bsid=8, frmsizecod=29) for s in range(S)]
    fa, fb = a[0].shape[1], b[0].shape[1]
    assert fb == fa + 2
    stride = (fb + 3) & ~3
    padded = np.zeros((S, F, stride), np.uint8)
    L = H.orc()
    want = np.zeros((S, F, 6, 2, 256), np.float32)
    for s in range(S):
        st = L.orc_a52_init()
        for f in range(F):
            fr = (a if f % 2 == 0 else b)[s][f]
            padded[s, f, :fr.size] = fr
            buf = np.zeros(fr.size + 64, np.uint8)
            buf[:fr.size] = fr
            fl, lv = H.ci(2), H.cf(1.0)
            assert L.orc_a52_frame(st, H.P(buf, H.u8p), ctypes.byref(fl), ctypes.byref(lv), 0.0) == 0
            for blk in range(6):
                assert L.orc_a52_block(st) == 0
                want[s, f, blk] = np.ctypeslib.as_array(L.orc_a52_samples(st), (1536,))[:512].reshape(2, 256)
        L.orc_a52_free(st)
    desc = pkg.DecodeDesc(flags=2, level=1.0, bias=0.0, dynrng=1, acmod=2, lfeon=0, frame_bytes=fb)
    delay = torch.zeros((S, 2, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    pcm, status, _ = engine.decode_batch(desc, torch.from_numpy(padded).cuda(), delay, lfsr, taps=True)
    engine.sync()
    assert (status.cpu().numpy() & 0x1ff).max() == 0
    err = pcm.cpu().numpy().astype(np.float64) - want
    assert H.rms(err) <= 1e-6 * max(1.0, H.rms(want))
