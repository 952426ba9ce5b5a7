"""GPU parity: ac3mi_decode_batch (HIP front end + transform) vs the liba52 restatement.

Reference behaviour under test: a52_frame + 6 x a52_block (liba52/parse.c:131-940,
bit_allocate.c, bitstream.c/.h) on frames produced by the encoder oracle.
 * exponents, bit allocation (bap), block-switch flags: bit-exact (integer stages)
 * dequantised coefficient planes: bit-exact (built with -ffp-contract=off)
 * PCM: <= 1e-6 RMS of +-1.0 full scale (north_star tolerance), float32 transform
 * dither generator state after the call: exact
"""
import ctypes

import numpy as np
import pytest

from tests import _harness as H

pytestmark = pytest.mark.gpu


def _streams(kind, S, F, seed0=0):
    """S independent streams of F frames each -> ([S][F][1536] u8 frames, [S] pcm arrays)."""
    frames = []
    for s in range(S):
        pcm = H.gen_pcm(F, 6, seed=seed0 + 1000 * s + 17, kind=kind)
        frames.append(H.orc_encode(pcm))
    return np.stack(frames)


def _oracle_decode_with_taps(frames, flags, level, bias, dynrng_off=False):
    """Per stream: oracle decode collecting PCM and the per-block stage taps."""
    L = H.orc()
    L.orc_a52_get_coefs.argtypes = [H.vp, H.fp, H.u8p]
    S, F, fb = frames.shape
    out = None
    taps = {"exp": np.zeros((S, F, 6, 7, 256), np.uint8), "bap": np.zeros((S, F, 6, 7, 256), np.int8),
            "coef": np.zeros((S, F, 6, 6, 256), np.float32), "blksw": np.zeros((S, F, 6, 5), np.uint8)}
    lfsr = np.zeros(S, np.int64)
    for s in range(S):
        st = L.orc_a52_init()
        buf = np.zeros(F * fb + 64, np.uint8)
        buf[:F * fb] = frames[s].reshape(-1)
        for f in range(F):
            fl, lv = H.ci(flags), H.cf(level)
            p = ctypes.cast(buf.ctypes.data + f * fb, H.u8p)
            assert L.orc_a52_frame(st, p, ctypes.byref(fl), ctypes.byref(lv), bias) == 0
            if dynrng_off:
                L.orc_a52_dynrng(st, None, None)
            nout = H.NFCHANS[fl.value & 15] + (1 if fl.value & 16 else 0)
            if out is None:
                out = np.zeros((S, F, 6, nout, 256), np.float32)
            for b in range(6):
                assert L.orc_a52_block(st) == 0
                out[s, f, b] = np.ctypeslib.as_array(L.orc_a52_samples(st), (1536,))[:nout * 256].reshape(nout, 256)
                for w in range(7):
                    L.orc_a52_get_exp(st, w, H.P(taps["exp"][s, f, b, w], H.u8p))
                    L.orc_a52_get_bap(st, w, H.P(taps["bap"][s, f, b, w], H.i8p))
                L.orc_a52_get_coefs(st, H.P(taps["coef"][s, f, b], H.fp), H.P(taps["blksw"][s, f, b], H.u8p))
        lfsr[s] = L.orc_a52_get_lfsr(st)
        L.orc_a52_free(st)
    return out, taps, lfsr, fl.value


def _gpu_decode(engine, frames, flags, level, bias, dynrng=1, taps=True):
    import torch
    pkg = H.pkg()
    S, F, fb = frames.shape
    n, sflags, _, _ = pkg.syncinfo(frames[0, 0])
    assert n == fb
    desc = pkg.DecodeDesc(flags=flags, level=level, bias=bias, dynrng=dynrng, acmod=sflags & 7 if (sflags & 15) != 10 else 2,
                          lfeon=1 if sflags & 16 else 0, frame_bytes=fb)
    n_out, _ = engine.decode_planes(desc)
    d_frames = torch.from_numpy(frames).cuda()
    delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    res = engine.decode_batch(desc, d_frames, delay, lfsr, taps=taps)
    engine.sync()
    pcm, status = res[0].cpu().numpy(), res[1].cpu().numpy()
    t = {k: v.cpu().numpy() for k, v in res[2].items()} if taps else None
    return pcm, status, t, lfsr.cpu().numpy().astype(np.int64) & 0xffff


@pytest.mark.parametrize("kind", ["tones", "noise", "quiet", "music", "bursts"])
def test_decode_5_1_all_stages(engine, kind):
    S, F = 6, 3
    frames = _streams(kind, S, F)
    ref_pcm, ref_taps, ref_lfsr, ref_flags = _oracle_decode_with_taps(frames, 7 | 16, 1.0, 0.0)
    pcm, status, taps, lfsr = _gpu_decode(engine, frames, 7 | 16, 1.0, 0.0)
    assert (status & 0x1ff).max() == 0, status
    assert ((status >> 16) & 0xff == ref_flags).all()
    # integer stages: bit-exact.  slots 0-4 fbw (223 bins), 5 lfe (7 bins); no coupling in these streams
    for w, n in [(0, 223), (1, 223), (2, 223), (3, 223), (4, 223), (5, 7)]:
        assert np.array_equal(taps["exp"][:, :, :, w, :n], ref_taps["exp"][:, :, :, w, :n]), "exp ch %d" % w
        assert np.array_equal(taps["bap"][:, :, :, w, :n], ref_taps["bap"][:, :, :, w, :n]), "bap ch %d" % w
    assert np.array_equal(taps["blksw"], ref_taps["blksw"])
    # dequantised planes: same bits (LFE first in both layouts for 5.1 output)
    assert np.array_equal(taps["coef"].view(np.uint32), ref_taps["coef"].view(np.uint32)), "coefficients differ"
    # dither generator advanced identically
    assert np.array_equal(lfsr, ref_lfsr)
    err = pcm.astype(np.float64) - ref_pcm
    assert H.rms(err) <= 1e-6 and np.abs(err).max() <= 4e-6, (H.rms(err), np.abs(err).max())


@pytest.mark.parametrize("flags,level,bias", [(2, 1.0, 0.0), (2 | 32, 1.0, 0.0), (10, 1.0, 0.0), (1 | 32, 0.5, 0.0),
                                               (3, 1.0, 0.0), (4 | 16, 1.0, 0.0), (5 | 32, 1.0, 0.0),
                                               (6 | 16 | 32, 2.0, 0.0), (7, 1.0, 0.0), (7 | 16 | 32, 1.0, 384.0)])
def test_decode_output_modes(engine, flags, level, bias):
    """Every output request the ACM driver or a52dec can make for a 5.1 stream
    (src/AC3ACM.cpp:1520-1553), incl. A52_ADJUST_LEVEL and the driver's bias 384."""
    S, F = 4, 2
    frames = _streams("tones", S, F, seed0=flags)
    ref_pcm, ref_taps, _, ref_flags = _oracle_decode_with_taps(frames, flags, level, bias)
    pcm, status, taps, _ = _gpu_decode(engine, frames, flags, level, bias)
    assert (status & 0x1ff).max() == 0
    assert ((status >> 16) & 0xff == ref_flags).all(), (status >> 16, ref_flags)
    # coefficient planes carry the per-channel downmix gains: compare bit for bit
    lfe_out = 1 if ref_flags & 16 else 0
    got = taps["coef"]                                   # [.., 6 planes: LFE, L, C, R, SL, SR]
    want = ref_taps["coef"]                              # liba52 layout: LFE first only when output
    assert np.array_equal(got[:, :, :, 1:6].view(np.uint32), want[:, :, :, lfe_out:lfe_out + 5].view(np.uint32))
    if lfe_out:
        assert np.array_equal(got[:, :, :, 0].view(np.uint32), want[:, :, :, 0].view(np.uint32))
    err = pcm.astype(np.float64) - ref_pcm
    tol = 1e-6 if bias == 0 else 4e-5                   # at bias 384 one float32 ulp is 3.05e-5
    assert H.rms(err) <= tol and np.abs(err).max() <= 4 * tol, (H.rms(err), np.abs(err).max())


def test_dynrng_off_and_long_stream(engine):
    """a52_dynrng(state, NULL, NULL) (src/AC3ACM.cpp:2047-2050) and a longer stream: the dither
    LFSR, csnroffst carry-over in the encoder and the overlap tails all thread through 12 frames."""
    frames = _streams("tones", 2, 12, seed0=5)
    ref_pcm, _, ref_lfsr, _ = _oracle_decode_with_taps(frames, 7 | 16, 1.0, 0.0, dynrng_off=True)
    pcm, status, _, lfsr = _gpu_decode(engine, frames, 7 | 16, 1.0, 0.0, dynrng=0, taps=False)
    assert (status & 0x1ff).max() == 0
    assert np.array_equal(lfsr, ref_lfsr)
    err = pcm.astype(np.float64) - ref_pcm
    assert H.rms(err) <= 1e-6, H.rms(err)


def test_corrupt_and_foreign_frames_are_flagged(engine):
    """Frames liba52 refuses: bad sync word (a52_syncinfo returns 0, parse.c:100-101) and a frame of
    another configuration inside the batch.  They must be reported, decode to silence, and must not
    disturb their neighbours."""
    frames = _streams("tones", 3, 2, seed0=9)
    bad = frames.copy()
    bad[1, 0, 0] = 0x0c                                  # stream 1 frame 0: sync word broken
    bad[2, 1, 6] = (bad[2, 1, 6] & 0x1f) | (2 << 5)      # stream 2 frame 1: acmod field says 2.0
    pcm, status, _, _ = _gpu_decode(engine, bad, 7 | 16, 1.0, 0.0, taps=False)
    ref_pcm, _, _, _ = _oracle_decode_with_taps(frames[:1], 7 | 16, 1.0, 0.0)
    assert status[0].tolist() == [23 << 16, 23 << 16]
    assert status[1, 0] & 0x100 and status[2, 1] & 0x100
    assert status[1, 1] & 0x1ff == 0 and status[2, 0] & 0x1ff == 0
    assert H.rms(pcm[0].astype(np.float64) - ref_pcm[0]) <= 1e-6
    assert np.abs(pcm[1, 0]).max() == 0.0                # refused frame + zero history -> silence


def test_exponent_error_is_reported(engine):
    """A reserved exponent-group code (>= 125) makes a52_block return 1 (parse.c:227-228 via
    tables.h exp_1[125..127] = 25).  Block 0 of a 5.1 frame from this encoder starts its first
    exponent group at a fixed position, so overwrite it with 127."""
    frames = _streams("tones", 1, 1, seed0=11)
    L = H.orc()
    st = L.orc_a52_init()
    buf = np.zeros(1600, np.uint8)
    buf[:1536] = frames[0, 0]
    fl, lv = H.ci(23), H.cf(1.0)
    L.orc_a52_frame(st, H.P(buf, H.u8p), ctypes.byref(fl), ctypes.byref(lv), 0.0)
    hdr_bits = L.orc_a52_bitpos(st)
    L.orc_a52_free(st)
    # block 0 side info of this encoder: 5 blksw, 5 dith, dynrnge, cplstre, cplinu, 5x2 chexpstr,
    # lfeexpstr, 5 x chbwcod(6) -> then channel 0: 4-bit absolute exponent, then the first 7-bit group
    pos = hdr_bits + 5 + 5 + 1 + 2 + 10 + 1 + 30 + 4
    bits = np.unpackbits(frames[0, 0])
    bits[pos:pos + 7] = 1
    bad = np.packbits(bits).reshape(1, 1, 1536)
    pcm, status, _, _ = _gpu_decode(engine, bad, 7 | 16, 1.0, 0.0, taps=False)
    assert status[0, 0] & 0x3f == 0x3f and not (status[0, 0] & 0x100)
    # the oracle agrees that block 0 fails
    st = L.orc_a52_init()
    buf[:1536] = bad[0, 0]
    fl, lv = H.ci(23), H.cf(1.0)
    assert L.orc_a52_frame(st, H.P(buf, H.u8p), ctypes.byref(fl), ctypes.byref(lv), 0.0) == 0
    assert L.orc_a52_block(st) == 1
    L.orc_a52_free(st)


@pytest.mark.parametrize("acmod,lfe,seed", [(7, 1, 11), (2, 0, 12), (0, 1, 13)])
def test_damaged_frames_match_liba52_block_by_block(engine, acmod, lfe, seed):
    """Random bit flips and bursts after the acmod field (tests/fuzz_corrupt.py): same first failing block as the
    oracle's a52_block, bit-identical coefficient planes before it, zero planes from it on, same dither state."""
    from tests import fuzz_corrupt
    bad, failed, _ = fuzz_corrupt.damaged_round(engine, seed, acmod, lfe)
    assert failed > 5          # the round exercises the error paths at all
    assert bad == 0


def test_replicas_in_a_large_batch_decode_alike(engine):
    """16 800 frames in one call (5 600 streams x 3 frames: above the frame-parallel front end's stream bound, so the parse
    kernel walks each stream's frames in order).  Replicated streams must decode to identical PCM, taps and state wherever
    they sit in the batch.  (Rounds 1-2 sent such batches through a two-chunk pipeline over two HIP streams, which this
    test was written for; the pipelines went in round 3.)"""
    import torch
    pkg = H.pkg()
    F, S = 3, 5600
    base = [H.orc_encode(H.gen_pcm(F, 6, seed=60 + s, kind=("tones", "bursts", "music", "quiet", "noise", "strobe", "tones")[s])) for s in range(7)]
    frames = torch.from_numpy(np.stack([base[s % 7] for s in range(S)])).cuda()
    desc = pkg.DecodeDesc(flags=7 | 16, level=1.0, bias=0.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=base[0].shape[1])
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    pcm, status, taps = engine.decode_batch(desc, frames, delay, lfsr, taps=True)
    engine.sync()
    assert int((status.cpu() & 0x1ff).max()) == 0
    p, d, l = pcm.cpu().numpy(), delay.cpu().numpy(), lfsr.cpu().numpy()
    c, b = taps["coef"].cpu().numpy(), taps["bap"].cpu().numpy()
    for s in range(7, S, 13):
        assert np.array_equal(p[s].view(np.uint32), p[s % 7].view(np.uint32)), s
        assert np.array_equal(d[s].view(np.uint32), d[s % 7].view(np.uint32)) and l[s] == l[s % 7], s
        assert np.array_equal(c[s].view(np.uint32), c[s % 7].view(np.uint32)) and np.array_equal(b[s], b[s % 7]), s
    # and the first replicas agree with the oracle
    want, errs, _ = H.orc_decode(base[1], 7 | 16, 1.0, 0.0)
    assert errs == 0 and H.rms(p[1].astype(np.float64) - want) <= 1e-6


@pytest.mark.parametrize("source", ["encoder", "packer"])
def test_frame_parallel_front_end_equals_the_serial_one(engine, source):
    """ac3mi_set_decode_mode: 1 = one wavefront per stream (frames in order, the one-kernel reference), 3 = one workgroup
    per stream, 4 / 5 = the split front end (parse kernel per stream / per frame + generator prefix, then one wavefront per
    audio block); 2 was retired in round 4.  Same PCM, status, taps and final LFSR state, bit for bit - on encoder output (dither-heavy
    "quiet" streams included) and on packer streams with coupling, rematrixing, delta bit allocation."""
    import torch
    from tests import packer
    pkg = H.pkg()
    if source == "encoder":
        F, acmod, lfe = 6, 7, 1
        streams = [H.orc_encode(H.gen_pcm(F, 6, seed=900 + s, kind=("quiet", "tones", "bursts", "music", "noise")[s % 5])) for s in range(10)]
    else:
        F, acmod, lfe = 5, 2, 0
        streams = [packer.make_stream(4000 + s, F, acmod, lfe, fscod=0, bsid=8, frmsizecod=30) for s in range(10)]
    frames = np.stack(streams)
    fb = frames.shape[2]
    stride = (fb + 3) & ~3
    padded = np.zeros((frames.shape[0], F, stride), np.uint8)
    padded[:, :, :fb] = frames
    S = padded.shape[0]
    flags = acmod | (16 if lfe else 0)
    desc = pkg.DecodeDesc(flags=flags, level=1.0, bias=0.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
    n_out, _ = engine.decode_planes(desc)
    res = {}
    try:
        for mode in (1, 3, 4, 5):
            engine.set_decode_mode(mode)
            delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
            lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
            lfsr[3] = 0x1234                         # a stream in mid-sequence
            lfsr[4] = 0                              # the generator's fixed point: no dither at all
            pcm, status, taps = engine.decode_batch(desc, torch.from_numpy(padded).cuda(), delay, lfsr, taps=True)
            engine.sync()
            # (the exponent tap also dumps rows and bins the frame does not define: leftovers of earlier frames in one mode,
            # zeros in the other; every defined exponent shapes the coefficient planes compared here)
            res[mode] = [x.cpu().numpy() for x in (pcm, status, delay, lfsr, taps["coef"], taps["blksw"])]
    finally:
        import os
        engine.set_decode_mode(int(os.environ.get("AC3MI_DECODE_MODE", "0")))
    assert (res[1][1] & 0x1ff).max() == 0
    for other in (3, 4, 5):                           # 3 = one workgroup per stream (decode_wg.hip), planes written out for the taps
        for a, b in zip(res[1], res[other]):
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), other


@pytest.mark.parametrize("acmod,lfe,source", [(7, 1, "encoder"), (7, 1, "packer"), (2, 0, "packer"), (3, 1, "packer"), (1, 0, "packer"), (6, 0, "packer")])
def test_fused_mantissa_transform_kernel_equals_the_two_kernels(engine, acmod, lfe, source):
    """ac3mi_set_decode_mode 6 (what auto picks for one-frame streams without a downmix): mantx_kernel (decode_mx.hip) unpacks
    a block's mantissas into LDS planes and transforms them in the same wavefront, overlap tails from block to block through
    LDS - against mode 4 (mant_kernel, planes through HBM, xform_kernel).  Same PCM (float and s16), status, overlap state and
    dither state, bit for bit: encoder output (dither-heavy quiet frames), packer streams with coupling, rematrixing, block
    switching and delta bit allocation, non-zero overlap state coming in, damaged frames (failed blocks leave zero planes),
    and state slots."""
    import torch
    from tests import packer
    pkg = H.pkg()
    if source == "encoder":
        F = 4
        streams = [H.orc_encode(H.gen_pcm(F, 6, seed=700 + s, kind=("quiet", "tones", "bursts", "music", "noise")[s % 5])) for s in range(10)]
    else:
        F = 5
        streams = [packer.make_stream(5100 + 7 * s + acmod, F, acmod, lfe) for s in range(10)]
    frames = np.stack(streams)
    fb = frames.shape[2]
    stride = (fb + 3) & ~3
    S = frames.shape[0] * F                                 # every frame a one-frame stream of its own
    padded = np.zeros((S, 1, stride), np.uint8)
    padded[:, 0, :fb] = frames.reshape(S, fb)
    rng = np.random.default_rng(acmod * 10 + lfe)
    for s in (3, 11):                                       # damage: a header that fails, and bits flipped mid-frame
        padded[s, 0, 0] ^= 0xff
    padded[7, 0, fb // 2:fb // 2 + 8] ^= rng.integers(1, 255, 8).astype(np.uint8)
    d_frames = torch.from_numpy(padded).cuda()
    flags = acmod | (16 if lfe else 0)
    res = {}
    try:
        for mode in (4, 6):
            engine.set_decode_mode(mode)
            out = []
            for bias, s16, slots in ((0.0, False, False), (384.0, True, False), (384.0, True, True)):
                desc = pkg.DecodeDesc(flags=flags | 32, level=1.0, bias=bias, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
                n_out, _ = engine.decode_planes(desc)
                if slots and n_out != 6:                    # (slot strides are those of six planes)
                    continue
                g = torch.Generator().manual_seed(5)
                delay = ((torch.rand((S, n_out, 128), generator=g) - 0.5) * 0.25).cuda()
                lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
                lfsr[5] = 0x1234
                lfsr[6] = 0
                if slots:                                   # stream s keeps its state in slot perm[s]
                    perm = torch.randperm(S, generator=g).to(torch.int32).cuda()
                    engine._check(engine.lib.ac3mi_set_state_slots(ctypes.c_void_p(engine.ctx), ctypes.c_void_p(perm.data_ptr())))
                try:
                    if s16:
                        pcm, status = engine.decode_s16_batch(desc, d_frames, delay, lfsr)
                    else:
                        pcm, status = engine.decode_batch(desc, d_frames, delay, lfsr)
                    engine.sync()
                finally:
                    if slots:
                        engine._check(engine.lib.ac3mi_set_state_slots(ctypes.c_void_p(engine.ctx), None))
                out += [x.cpu().numpy() for x in (pcm, status, delay, lfsr)]
            res[mode] = out
    finally:
        import os
        engine.set_decode_mode(int(os.environ.get("AC3MI_DECODE_MODE", "0")))
    st = res[4][1].reshape(-1)
    assert (st[[3, 11]] & 0x100).all() and (np.delete(st, [3, 7, 11]) & 0x1ff).max() == 0
    assert np.abs(res[4][0]).max() > 1e-3
    for k, (a, b) in enumerate(zip(res[4], res[6])):
        diff = np.argwhere(a.view(np.uint8).reshape(a.shape[0], -1) != b.view(np.uint8).reshape(b.shape[0], -1))
        assert diff.size == 0, (k, a.shape, sorted(set(diff[:, 0].tolist())), diff[:8].tolist())


@pytest.mark.parametrize("acmod,lfe,req", [(7, 1, 7 | 16), (7, 1, 2), (2, 0, 2), (7, 1, 10), (3, 1, 3 | 16), (1, 0, 1), (6, 0, 6), (7, 0, 4)])
def test_decode_s16_equals_decode_plus_converter(engine, acmod, lfe, req):
    """ac3mi_decode_s16_batch (the transform writes interleaved s16 itself) against ac3mi_decode_batch at level 1 / bias
    384 followed by ac3mi_convert_s16_batch, on packer streams with block switching and downmixes: same samples, same
    carry-over state.  Includes a hot stream whose samples saturate (packssdw)."""
    import torch
    from tests import packer
    pkg = H.pkg()
    S, F = 7, 3
    frames = np.stack([packer.make_stream(4000 + 13 * s + acmod, F, acmod, lfe) for s in range(S)])
    fb = frames.shape[2]
    stride = (fb + 3) & ~3
    padded = np.zeros((S, F, stride), np.uint8)
    padded[:, :, :fb] = frames
    d_frames = torch.from_numpy(padded).cuda()
    desc = pkg.DecodeDesc(flags=req | 32, level=1.0, bias=384.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
    n_out, oflags = engine.decode_planes(desc)
    delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    pcm, status = engine.decode_batch(desc, d_frames, delay, lfsr)
    engine.sync()
    want = torch.empty((S, F, 6, 256, n_out), dtype=torch.int16, device="cuda")
    engine._check(engine.lib.ac3mi_convert_s16_batch(ctypes.c_void_p(engine.ctx), ctypes.c_void_p(pcm.data_ptr()),
                                                     ctypes.c_void_p(want.data_ptr()), oflags, ctypes.c_size_t(S * F * 6)))
    delay2 = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
    lfsr2 = torch.ones((S,), dtype=torch.int16, device="cuda")
    got, status2 = engine.decode_s16_batch(desc, d_frames, delay2, lfsr2)
    engine.sync()
    assert int((status.cpu() & 0x1ff).max()) == 0
    assert torch.equal(got.cpu(), want.cpu())
    assert torch.equal(status.cpu(), status2.cpu()) and torch.equal(delay.cpu(), delay2.cpu()) and torch.equal(lfsr.cpu(), lfsr2.cpu())
    assert int(want.cpu().abs().max()) > 100
