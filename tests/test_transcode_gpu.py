"""GPU: ac3mi_transcode_batch == ac3mi_decode_batch (level 1, bias 384) + ac3mi_convert_s16_batch + ac3mi_encode_batch,
byte for byte and state for state, for a batch small enough to run unchunked and one large enough to go through the
two-stream chunk pipeline; and against the oracle chain (oracle decode -> s16 -> oracle encode) on a few streams."""
import ctypes

import numpy as np
import pytest

from tests import _harness as H

pytestmark = pytest.mark.gpu


def _three_calls(engine, pkg, frames_t, dec, enc, chmap, S, F, n_out):
    import torch
    dev = frames_t.device
    delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device=dev)
    lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
    pcm, status = engine.decode_batch(dec, frames_t, delay, lfsr)
    s16 = torch.empty((S * F * 6, 256, n_out), dtype=torch.int16, device=dev)
    engine.sync()
    _, oflags = engine.decode_planes(dec)
    engine._check(engine.lib.ac3mi_convert_s16_batch(ctypes.c_void_p(engine.ctx), ctypes.c_void_p(pcm.data_ptr()),
                                                     ctypes.c_void_p(s16.data_ptr()), oflags, ctypes.c_size_t(S * F * 6)))
    last = torch.zeros((S, n_out, 256), dtype=torch.int16, device=dev)
    csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
    out = engine.encode_batch(enc, s16.view(S, F, 1536, n_out), chmap, last, csnr)
    engine.sync()
    return out, status, delay, lfsr, last, csnr


@pytest.mark.parametrize("S,F,bitrate", [(5, 3, 448000), (6000, 3, 448000), (3000, 2, 224000), (3000, 2, 640000)])
def test_transcode_equals_the_three_calls(engine, S, F, bitrate):
    """(The large batches go through the split front end, whose descriptors carry the source frames' SNR offsets to the
    encoder's search as a starting point; the target rates far from the source's 384 kbps make that hint a bad one:
    it must not change a byte.)"""
    import torch
    pkg = H.pkg()
    base = [H.orc_encode(H.gen_pcm(F, 6, seed=400 + s, kind=("tones", "music", "bursts", "noise")[s % 4])) for s in range(8)]
    frames = np.stack([base[s % 8] for s in range(S)])
    fb = frames.shape[2]
    frames_t = torch.from_numpy(frames).cuda()
    dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
    enc = pkg.EncodeDesc(48000, bitrate, 6)
    want = _three_calls(engine, pkg, frames_t, dec, enc, H.CHMAP6, S, F, 6)
    dev = frames_t.device
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
    lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
    last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
    csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
    out, status = engine.transcode_batch(dec, enc, frames_t, delay, lfsr, H.CHMAP6, last, csnr)
    engine.sync()
    for got, exp in zip((out, status, delay, lfsr, last, csnr), want):
        assert torch.equal(got.cpu(), exp.cpu())
    assert int((status.cpu() & 0x1ff).max()) == 0
    # the streams repeat with period 8: whatever chunk of the pipeline a stream went through, it must come out like its twin
    o = out.cpu().numpy()
    for s in range(8, S):
        assert np.array_equal(o[s], o[s % 8]), s


def test_transcode_against_the_oracle_chain(engine):
    import torch
    pkg = H.pkg()
    S, F = 3, 3
    streams = [H.orc_encode(H.gen_pcm(F, 6, seed=500 + s, kind=("tones", "music", "bursts")[s])) for s in range(S)]
    frames_t = torch.from_numpy(np.stack(streams)).cuda()
    dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=streams[0].shape[1])
    enc = pkg.EncodeDesc(48000, 384000, 6)
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    last = torch.zeros((S, 6, 256), dtype=torch.int16, device="cuda")
    csnr = torch.full((S,), 40, dtype=torch.int32, device="cuda")
    out, status = engine.transcode_batch(dec, enc, frames_t, delay, lfsr, H.CHMAP6, last, csnr)
    engine.sync()
    L = H.orc()
    for s in range(S):
        want, errs, oflags = H.orc_decode(streams[s], 7 | 16 | 32, 1.0, 384.0)
        s16 = np.zeros((F * 6, 256, 6), np.int16)
        for f in range(F):
            for b in range(6):
                L.orc_convert_s16(H.P(np.ascontiguousarray(want[f, b]), H.fp), H.P(s16[f * 6 + b], H.i16p), oflags)
        ref = H.orc_encode(s16.reshape(F * 1536, 6))
        # The decoded float PCM may differ from the oracle's by one float32 ulp at bias 384 = one s16 step, and the
        # encoder amplifies that: perturbing 2 % of the oracle chain's own s16 input by one step moves its output by
        # up to 15 % RMS on noisy content.  So the two second-generation signals are compared through their
        # distance to the first generation: the same coding noise, within a fraction of a dB.
        got_pcm, errs_g, _ = H.orc_decode(out[s].cpu().numpy()[:, :ref.shape[1]].copy(), 7 | 16, 1.0, 0.0)
        ref_pcm, errs_r, _ = H.orc_decode(ref, 7 | 16, 1.0, 0.0)
        first, _, _ = H.orc_decode(streams[s], 7 | 16, 1.0, 0.0)
        assert errs_g == 0 and errs_r == 0
        # the second generation lags the first by 256 samples (one more analysis/synthesis overlap)
        a = np.moveaxis(first, 2, 0).reshape(6, -1)[:, :-256].astype(np.float64)
        noise_g = H.rms(np.moveaxis(got_pcm, 2, 0).reshape(6, -1)[:, 256:] - a)
        noise_r = H.rms(np.moveaxis(ref_pcm, 2, 0).reshape(6, -1)[:, 256:] - a)
        assert noise_g <= 1.1 * noise_r + 1e-6, (noise_g, noise_r)


@pytest.mark.parametrize("tile", [4, 7, 30])
def test_workspace_tiles_do_not_change_anything(engine, tile):
    """ac3mi_set_tile_frames: a batch above the bound goes through in tiles of whole streams (tile 4 with F = 3: one
    stream at a time; 7: two streams; 30: ten).  Decode, encode and transcode must give the bytes, PCM and carry-over
    state of the untiled call."""
    import torch
    pkg = H.pkg()
    S, F = 23, 3
    kinds = ("tones", "music", "bursts", "noise", "quiet", "strobe")
    streams = np.stack([H.orc_encode(H.gen_pcm(F, 6, seed=900 + s, kind=kinds[s % 6])) for s in range(S)])
    frames_t = torch.from_numpy(streams).cuda()
    pcm_in = torch.from_numpy(np.stack([H.gen_pcm(F, 6, seed=950 + s, kind=kinds[(s + 2) % 6]) for s in range(S)]).reshape(S, F, 1536, 6)).cuda()
    dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=streams.shape[2])
    enc = pkg.EncodeDesc(48000, 384000, 6)

    def run():
        dev = frames_t.device
        delay = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
        lfsr = torch.ones((S,), dtype=torch.int16, device=dev)
        pcm, st = engine.decode_batch(dec, frames_t, delay, lfsr)
        last = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
        csnr = torch.full((S,), 40, dtype=torch.int32, device=dev)
        fr = engine.encode_batch(enc, pcm_in, H.CHMAP6, last, csnr)
        delay2 = torch.zeros((S, 6, 128), dtype=torch.float32, device=dev)
        lfsr2 = torch.ones((S,), dtype=torch.int16, device=dev)
        last2 = torch.zeros((S, 6, 256), dtype=torch.int16, device=dev)
        csnr2 = torch.full((S,), 40, dtype=torch.int32, device=dev)
        out, st2 = engine.transcode_batch(dec, enc, frames_t, delay2, lfsr2, H.CHMAP6, last2, csnr2)
        engine.sync()
        return [t.cpu() for t in (pcm, st, delay, lfsr, fr, last, csnr, out, st2, delay2, lfsr2, last2, csnr2)]

    engine.set_tile_frames(0)
    want = run()
    try:
        engine.set_tile_frames(tile)
        got = run()
    finally:
        engine.set_tile_frames(131072)
    names = "pcm status delay lfsr frames last csnr tc_frames tc_status tc_delay tc_lfsr tc_last tc_csnr".split()
    for n, g, w in zip(names, got, want):
        assert torch.equal(g, w), n
