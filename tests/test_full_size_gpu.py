"""BASELINE configs[2], [3] and [4] at the size bench.py times them (65 536 independent one-frame streams), through the
C-ABI.  The oracle cannot do 65 536 frames in test time, so every frame is checked through size-independent properties
(both CRCs of every produced frame on the host, ENC/ac3enc.cpp:1599-1638; clean status words; linearity of the mixing
transform, L52/parse.c:881-937) and a slice of the batch is compared with the oracle exactly as the small tests do:
byte for byte (encoder), within one s16 step (decoder), 1e-6 RMS (transform)."""
import ctypes

import numpy as np
import pytest

from tests import _harness as H

pytestmark = pytest.mark.gpu
S = 65536
CHMAP = (0, 2, 1, 4, 5, 3)


@pytest.fixture(scope="module")
def engine():
    pkg = H.pkg()
    eng = pkg.Engine(0)
    yield eng
    eng.close()


def _bench_pcm(seed=99):
    """bench.py's encoder input: tones + noise with level steps per 512-sample segment and channel (frames then carry a
    realistic mix of new and reused exponent sets)."""
    import torch
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(seed)
    t = torch.arange(1536, device=dev, dtype=torch.float32)
    ph = torch.rand((S, 1, 6), device=dev, generator=g) * 6.28
    fr = 0.01 * torch.arange(1, 7, device=dev, dtype=torch.float32)
    pcm = 8000.0 * torch.sin(ph + fr * t[None, :, None]) + (torch.rand((S, 1536, 6), device=dev, generator=g) - 0.5) * 4096
    env = torch.where(torch.rand((S, 3, 1, 6), device=dev, generator=g) < 0.5, 1.0, 1.0 / 32)
    pcm = (pcm.reshape(S, 3, 512, 6) * env).reshape(S, 1536, 6)
    return pcm.round().clamp(-32768, 32767).to(torch.int16).reshape(S, 1, 1536, 6).contiguous()


def _oracle_frames(pcm_host, n):
    """the oracle's encoding of the first n one-frame streams (fresh state each: history 0, csnroffst 40)"""
    O = H.orc()
    O.orc_ac3enc_encode_frames.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, H.i16p, ctypes.c_int, H.u8p, H.u8p]
    cm = (ctypes.c_uint8 * 8)(*H.CHMAP6)
    want = np.zeros((n, 1536), np.uint8)
    for i in range(n):
        src = np.ascontiguousarray(pcm_host[i].reshape(1536 * 6))
        assert O.orc_ac3enc_encode_frames(48000, 384000, 6, H.P(src, H.i16p), 1, cm, H.P(want[i], H.u8p)) == 0
    return want


def test_encode_full_size(engine):
    """configs[2]: 65 536 x (6 ch x 1536 samples) through ac3mi_encode_batch from fresh stream state.  Every frame: sync
    word, both CRCs; the first 256 streams byte for byte against the oracle."""
    import torch
    import bench
    pkg = H.pkg()
    enc = pkg.EncodeDesc(48000, 384000, 6)
    fb = enc.frame_bytes()
    pcm = _bench_pcm()
    last = torch.zeros((S, 6, 256), dtype=torch.int16, device="cuda")
    csnr = torch.full((S,), 40, dtype=torch.int32, device="cuda")
    frames = engine.encode_batch(enc, pcm, CHMAP, last, csnr)
    engine.sync()
    host = frames.cpu().numpy().reshape(S, -1)[:, :fb]
    assert (host[:, 0] == 0x0b).all() and (host[:, 1] == 0x77).all()
    assert bench.ac3_crc_ok(host) == 0
    c = csnr.cpu().numpy()
    assert ((c & 0xff) <= 63).all() and ((c & 0xff) > 0).any()
    n = 256
    want = _oracle_frames(pcm[:n].cpu().numpy(), n)
    assert np.array_equal(host[:n], want), np.nonzero((host[:n] != want).any(axis=1))[0][:8]
    # the stream history the call leaves: the last 256 samples per channel, in coded channel order
    assert torch.equal(last[:, 0], pcm[:, 0, 1280:, CHMAP[0]])


def test_transcode_full_size(engine):
    """configs[4]'s per-GPU step at bench size: 65 536 frames decoded to s16 and re-encoded in one ac3mi_transcode_batch
    call from fresh stream state.  Every frame: decoder status clean, both CRCs of the new frame; the first 256 streams
    against the oracle's decode (<= 1 s16 step, the float PCM may differ by an ulp at bias 384) and, from the engine's
    own samples, against the oracle's encoding byte for byte."""
    import torch
    import bench
    pkg = H.pkg()
    enc = pkg.EncodeDesc(48000, 384000, 6)
    fb = enc.frame_bytes()
    pcm = _bench_pcm(seed=1234)
    last = torch.zeros((S, 6, 256), dtype=torch.int16, device="cuda")
    csnr = torch.full((S,), 40, dtype=torch.int32, device="cuda")
    frames = engine.encode_batch(enc, pcm, CHMAP, last, csnr)
    engine.sync()
    del pcm
    dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    last2 = torch.zeros((S, 6, 256), dtype=torch.int16, device="cuda")
    csnr2 = torch.full((S,), 40, dtype=torch.int32, device="cuda")
    status = torch.zeros((S, 1), dtype=torch.int32, device="cuda")
    out, _ = engine.transcode_batch(dec, enc, frames, delay, lfsr, CHMAP, last2, csnr2, status=status)
    engine.sync()
    assert int((status & 0x3ff).max().item()) == 0
    host = out.cpu().numpy().reshape(S, -1)[:, :fb]
    assert (host[:, 0] == 0x0b).all() and (host[:, 1] == 0x77).all()
    assert bench.ac3_crc_ok(host) == 0
    # the same two steps through the separate calls: identical frames and state (the one-call path is a pipeline of them)
    delay_b = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    lfsr_b = torch.ones((S,), dtype=torch.int16, device="cuda")
    s16, st_b = engine.decode_s16_batch(dec, frames, delay_b, lfsr_b)
    last_b = torch.zeros((S, 6, 256), dtype=torch.int16, device="cuda")
    csnr_b = torch.full((S,), 40, dtype=torch.int32, device="cuda")
    again = engine.encode_batch(enc, s16.reshape(S, 1, 1536, 6), CHMAP, last_b, csnr_b)
    engine.sync()
    assert torch.equal(out, again) and torch.equal(lfsr, lfsr_b) and torch.equal(delay, delay_b) and torch.equal(csnr2, csnr_b)
    # a slice against the oracle
    O = H.orc()
    n = 256
    src = frames[:n].cpu().numpy().reshape(n, -1)[:, :fb]
    got16 = s16[:n].cpu().numpy().reshape(n, 6, 256, 6)
    ref16 = np.zeros((256, 6), np.int16)
    for i in range(n):
        pcmf, errs, oflags = H.orc_decode(src[i:i + 1], 7 | 16 | 32, 1.0, 384.0)
        assert errs == 0
        frame16 = np.zeros((6, 256, 6), np.int16)
        for b in range(6):
            O.orc_convert_s16(H.P(np.ascontiguousarray(pcmf[0, b]), H.fp), H.P(ref16, H.i16p), oflags)
            frame16[b] = ref16
        assert np.abs(got16[i].astype(np.int32) - frame16.astype(np.int32)).max() <= 1, i
    # ... and the encoder half on exactly the samples the decoder half produced
    want = _oracle_frames(got16.reshape(n, 1536, 6), n)
    assert np.array_equal(host[:n], want), np.nonzero((host[:n] != want).any(axis=1))[0][:8]


def test_mixed_blocks_downmix_full_size(engine):
    """configs[3] at bench size: 65 536 frames of 5.1 coefficients, a quarter of the channel-blocks switched to IMDCT-256,
    mixed down to 2.0 inside the transform.  Linearity over the whole batch (the mix and both transforms are linear in
    the coefficients for fixed block-switch flags: T(a x + b y) = a T(x) + b T(y) from zero state), and a 16-stream slice
    against the oracle (<= 1e-6 RMS)."""
    import torch
    pkg = H.pkg()
    desc = pkg.XformDesc(7, 1, 2, 0.0)
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn((S, 1, 6, 6, 256), device="cuda", generator=g) * 0.05
    y = torch.randn((S, 1, 6, 6, 256), device="cuda", generator=g) * 0.05
    blksw = (torch.rand((S, 1, 6, 5), device="cuda", generator=g) < 0.25).to(torch.uint8)
    z = lambda: torch.zeros((S, 2, 128), device="cuda")
    tx = engine.imdct_batch(desc, x, z(), blksw=blksw)
    ty = engine.imdct_batch(desc, y, z(), blksw=blksw)
    txy = engine.imdct_batch(desc, (0.5 * x - 2.0 * y).contiguous(), z(), blksw=blksw)
    engine.sync()
    assert tuple(tx.shape) == (S, 1, 6, 2, 256)
    err = (txy - (0.5 * tx - 2.0 * ty)).abs().max().item()
    assert err < 2e-5, err
    sl = slice(40000, 40016)
    ref, _ = H.orc_xform(x[sl].cpu().numpy(), blksw[sl].cpu().numpy(), 7, 1, 2, bias=0.0, clev=0.7071, slev=0.7071)
    got = tx[sl].cpu().numpy()
    assert H.rms(got.astype(np.float64) - ref) <= 1e-6
    assert np.abs(got - ref).max() <= 1e-5


def test_million_stream_transcode_and_the_workspace_plan():
    """BASELINE configs[4], one GPU's share: 2^20 independent one-frame streams decoded and re-encoded in ONE
    ac3mi_transcode_batch call (eight tiles of 131 072 frames under the default workspace bound).  The batch is sixteen
    replicas of 65 536 streams: every replica must come out byte-identical to the first whatever tile it fell in, every
    status word clean, and the first replica's frames carry valid CRCs (hence all do).  On the way, the multi-GPU planner's
    workspace figure (sharding.plan_transcode_bytes, which asks the library) against what a fresh engine really holds."""
    import torch
    import bench
    pkg = H.pkg()
    eng = pkg.Engine(0)                                   # a fresh context: its workspaces start empty
    try:
        enc = pkg.EncodeDesc(48000, 384000, 6)
        fb = enc.frame_bytes()
        pcm = _bench_pcm(seed=7)
        last = torch.zeros((S, 6, 256), dtype=torch.int16, device="cuda")
        csnr = torch.full((S,), 40, dtype=torch.int32, device="cuda")
        base = eng.encode_batch(enc, pcm, CHMAP, last, csnr)
        eng.sync()
        del pcm, last, csnr
        N = 1 << 20
        big = base.repeat(N // S, 1, 1).contiguous()
        dec = pkg.DecodeDesc(flags=7 | 16 | 32, level=1.0, bias=384.0, dynrng=1, acmod=7, lfeon=1, frame_bytes=fb)
        delay = torch.zeros((N, 6, 128), dtype=torch.float32, device="cuda")
        lfsr = torch.ones((N,), dtype=torch.int16, device="cuda")
        last = torch.zeros((N, 6, 256), dtype=torch.int16, device="cuda")
        csnr = torch.full((N,), 40, dtype=torch.int32, device="cuda")
        status = torch.zeros((N, 1), dtype=torch.int32, device="cuda")
        ws0 = eng.workspace_bytes()
        out, _ = eng.transcode_batch(dec, enc, big, delay, lfsr, CHMAP, last, csnr, status=status)
        eng.sync()
        assert int((status & 0x3ff).max().item()) == 0
        first = out[:S]
        for r in range(1, N // S):
            assert torch.equal(out[r * S:(r + 1) * S], first), "replica %d differs from replica 0" % r
        assert torch.equal(csnr[S:2 * S], csnr[:S]) and torch.equal(lfsr[-S:], lfsr[:S])
        host = first.cpu().numpy().reshape(S, -1)[:, :fb]
        assert (host[:, 0] == 0x0b).all() and (host[:, 1] == 0x77).all()
        assert bench.ac3_crc_ok(host) == 0
        # the plan: the call went through in tiles of 131 072 frames, so the engine holds exactly one tile's workspaces
        held = eng.workspace_bytes()
        plan = pkg.sharding.plan_transcode_bytes(N)
        assert plan["tile_frames"] == 131072 and ws0 < held <= plan["workspace"], (ws0, held, plan)
        assert plan["workspace"] - held < 64 << 20            # (the encode call above left its own, smaller arrays)
        free, total = torch.cuda.mem_get_info()
        assert plan["total"] < total and plan["fits"]
    finally:
        eng.close()
