"""GPU: the byte-stream layer (include/ac3mi_stream.h), i.e. the ACM driver's stream_open / stream_size /
stream_convert messages (src/AC3ACM.cpp:1430-1628, 1665-1798, 1862-2131, 2139-2363) over the batched engine.

PARITY UNPINNED for the buffering state machine itself: src/AC3ACM.cpp cannot be built here (Win32 SDK) and
the reference holds no fixtures for it.  What is pinned:
  * the PCM / AC-3 bytes that come out are the oracle decoder's / encoder's for the same frames, whatever
    the chunking (s16 within one step: the float PCM may differ by one float32 ulp at bias 384),
  * byte accounting follows tests/stream_model.py, an independent Python restatement of the reference's
    two convert functions (call-by-call src_used / dst_used),
  * n streams advanced together (one batched launch per round) end exactly as n separate streams."""
import ctypes
import os

import numpy as np
import pytest

from tests import _harness as H
from tests import stream_model as M

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def S():
    import importlib
    return importlib.import_module("ac-3-acm-codec_amd.stream")


@pytest.fixture()
def pool(engine, S):
    p = S.Pool(engine, 16)
    yield p
    p.close()


def _oracle_s16(frames, flags):
    """oracle decode at level 1 / bias 384 + the s16 converter -> [nframes*6][256][nout] s16"""
    want, errs, oflags = H.orc_decode(frames, flags | 32, 1.0, 384.0)
    assert errs == 0
    nout = want.shape[2]
    out = np.zeros((want.shape[0] * 6, 256, nout), np.int16)
    L = H.orc()
    for f in range(want.shape[0]):
        for b in range(6):
            L.orc_convert_s16(H.P(np.ascontiguousarray(want[f, b]), H.fp), H.P(out[f * 6 + b], H.i16p), oflags)
    return out


def _run_calls(S, stream, data, rng, max_chunk, dst_choices, model, flush=4):
    """offer `data` in random chunks (what a call does not use is offered again) and random destination sizes;
    after every call compare the header with the model; a few final calls with no input collect leftovers"""
    out = bytearray()
    pos = 0
    first = True
    flushes = 0
    while flushes < flush:
        n = int(min(len(data) - pos, rng.integers(1, max_chunk)))
        dcap = int(rng.choice(dst_choices))
        if n == 0:
            flushes += 1
            dcap = max(dst_choices)
        src = np.frombuffer(data[pos:pos + n], np.uint8).copy() if n else np.zeros(1, np.uint8)
        dst = np.zeros(max(dcap, 1), np.uint8)
        h = S.StreamHeader(src.ctypes.data, n, 0, dst.ctypes.data, dcap, 0, S.STREAMCONVERTF_START if first else 0)
        assert stream.convert(h) == 0
        su, du = model.convert(bytes(data[pos:pos + n]), dcap, first)
        assert (h.src_used, h.dst_used) == (su, du), (pos, n, dcap, h.src_used, h.dst_used, su, du)
        out += dst[:h.dst_used].tobytes()
        pos += h.src_used
        first = False
    return bytes(out), pos


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_decode_any_chunking(S, pool, seed):
    rng = np.random.default_rng(seed)
    frames = H.orc_encode(H.gen_pcm(5, 6, seed=40 + seed, kind=("tones", "music", "bursts")[seed % 3]))
    fb = frames.shape[1]
    junk = rng.integers(0, 256, 37, dtype=np.uint8).tobytes()       # resync: garbage before the first frame
    junk = junk.replace(b"\x0b\x77", b"\x0b\x76")
    data = junk + frames.tobytes()
    rc, st = pool.open(S.ac3_format(6, 48000, 384, block_align=fb), S.pcm_format(6, 48000))
    assert rc == 0 and st is not None
    model = M.DecodeModel(src_channels=6, dst_channels=6)
    got, used = _run_calls(S, st, data, rng, 2500, [1536 * 12, 512 * 6 * 2, 4000, 40000], model)
    st.close()
    assert used == len(data)
    want = _oracle_s16(frames, 7 | 16).reshape(-1)
    g = np.frombuffer(got, np.int16)
    assert g.size == want.size
    assert int(np.abs(g.astype(np.int32) - want).max()) <= 1


def test_decode_downmix_and_leftover_blocks(S, pool):
    frames = H.orc_encode(H.gen_pcm(4, 6, seed=77, kind="tones"))
    fb = frames.shape[1]
    rc, st = pool.open(S.ac3_format(6, 48000, 384, block_align=fb), S.pcm_format(2, 48000), S.ACM_DYNAMICRANGE)
    assert rc == 0
    data = frames.tobytes()
    # whole stream in one call, destination holds only 4 blocks at a time: blocks carry over between calls
    chunks = [len(data)] + [0] * 12
    dsts = [4 * 512 * 2] * len(chunks)
    model = M.DecodeModel(src_channels=6, dst_channels=2)
    out = bytearray()
    pos = 0
    for i, (n, dcap) in enumerate(zip(chunks, dsts)):
        src = np.frombuffer(data[pos:pos + max(n, 0)] if n else b"\0", np.uint8).copy()
        left = len(data) - pos
        n = left if i else n
        src = np.frombuffer(data[pos:pos + n] if n else b"\0", np.uint8).copy()
        dst = np.zeros(dcap, np.uint8)
        h = S.StreamHeader(src.ctypes.data, n, 0, dst.ctypes.data, dcap, 0, S.STREAMCONVERTF_START if i == 0 else 0)
        assert st.convert(h) == 0
        assert (h.src_used, h.dst_used) == model.convert(data[pos:pos + n], dcap, i == 0)
        out += dst[:h.dst_used].tobytes()
        pos += h.src_used
    st.close()
    want = _oracle_s16(frames, 2).reshape(-1)
    g = np.frombuffer(bytes(out), np.int16)
    assert g.size == want.size
    assert int(np.abs(g.astype(np.int32) - want).max()) <= 1


def test_encode_any_chunking(S, pool):
    rng = np.random.default_rng(9)
    pcm = H.gen_pcm(4, 6, seed=5, kind="music")
    rc, st = pool.open(S.pcm_format(6, 48000), S.ac3_format(6, 48000, 384))
    assert rc == 0
    data = pcm.tobytes()
    chunks, left = [], len(data)
    while left:
        n = int(min(left, rng.integers(1, 30000)))
        chunks.append(n)
        left -= n
    chunks += [0, 0, 0]
    dsts = [int(rng.choice([700, 1536, 4096])) for _ in chunks]
    model = M.EncodeModel(channels=6, frame_bytes=1536)
    # a short destination makes the reference stop consuming: re-offer what was not used
    out = bytearray()
    pos = 0
    first = True
    for dcap in dsts + [4096] * 40:
        n = int(min(len(data) - pos, rng.integers(0, 30000)))
        src = np.frombuffer(data[pos:pos + n] if n else b"\0", np.uint8).copy()
        dst = np.zeros(dcap, np.uint8)
        h = S.StreamHeader(src.ctypes.data, n, 0, dst.ctypes.data, dcap, 0, S.STREAMCONVERTF_START if first else 0)
        assert st.convert(h) == 0
        assert (h.src_used, h.dst_used) == model.convert(n, dcap, first)
        out += dst[:h.dst_used].tobytes()
        pos += h.src_used
        first = False
    st.close()
    assert pos == len(data)
    want = H.orc_encode(pcm).tobytes()
    assert bytes(out) == want


@pytest.mark.parametrize("tile", [131072, 2])
def test_many_streams_one_launch_per_round(S, pool, engine, tile):
    """tile = 2: the engine's workspace bound (ac3mi_set_tile_frames) forces every batched call of the round into tiles
    of whole streams, with the state-slot table moving along."""
    engine.set_tile_frames(tile)
    try:
        _many_streams(S, pool)
    finally:
        engine.set_tile_frames(131072)


def _many_streams(S, pool):
    rng = np.random.default_rng(4)
    kinds = ("tones", "music", "noise", "bursts", "quiet")
    frames = [H.orc_encode(H.gen_pcm(3, 6, seed=100 + i, kind=kinds[i % 5])) for i in range(5)]
    pcms = [H.gen_pcm(3, 2, seed=200 + i, kind=kinds[i % 5]) for i in range(3)]
    streams, datas, wants = [], [], []
    for fr in frames:
        rc, st = pool.open(S.ac3_format(6, 48000, 384, block_align=1536), S.pcm_format(6, 48000))
        assert rc == 0
        streams.append(st); datas.append(fr.tobytes()); wants.append(_oracle_s16(fr, 7 | 16).tobytes())
    for p in pcms:
        rc, st = pool.open(S.pcm_format(2, 48000), S.ac3_format(2, 48000, 192))
        assert rc == 0
        streams.append(st); datas.append(p.tobytes()); wants.append(H.orc_encode(p, nch=2, bitrate=192000, chmap=(0, 1)).tobytes())
    n = len(streams)
    pos = [0] * n
    outs = [bytearray() for _ in range(n)]
    first = True
    for _ in range(40):
        hs, keep = [], []
        for i in range(n):
            k = int(min(len(datas[i]) - pos[i], rng.integers(0, 5000)))
            src = np.frombuffer(datas[i][pos[i]:pos[i] + k] if k else b"\0", np.uint8).copy()
            dst = np.zeros(40000, np.uint8)
            keep.append((src, dst))
            hs.append(S.StreamHeader(src.ctypes.data, k, 0, dst.ctypes.data, dst.size, 0, S.STREAMCONVERTF_START if first else 0))
        assert pool.convert_many(streams, hs) == 0
        for i in range(n):
            assert hs[i].src_used == hs[i].src_len          # a large destination always drains the source
            pos[i] += hs[i].src_used
            outs[i] += keep[i][1][:hs[i].dst_used].tobytes()
        first = False
    for i in range(n):
        assert pos[i] == len(datas[i])
        if i < len(frames):
            g = np.frombuffer(bytes(outs[i]), np.int16).astype(np.int32)
            w = np.frombuffer(wants[i], np.int16).astype(np.int32)
            assert g.size == w.size and int(np.abs(g - w).max()) <= 1
        else:
            assert bytes(outs[i]) == wants[i]
    for st in streams:
        st.close()


def test_open_rules_and_sizes(S, pool):
    ac3 = S.ac3_format(6, 48000, 384, block_align=1536)
    # result codes of stream_open (src/AC3ACM.cpp:1862-2097)
    assert pool.open(ac3, S.pcm_format(6, 48000), 0, query=True)[0] == S.MMSYSERR_NOTSUPPORTED     # > 2 ch needs MULTICHANNEL
    assert pool.open(ac3, S.pcm_format(6, 48000, extensible=True), 0, query=True)[0] == 0
    assert pool.open(ac3, S.pcm_format(6, 44100), query=True)[0] == S.MMSYSERR_NOTSUPPORTED        # no resampling
    assert pool.open(ac3, S.pcm_format(4, 48000), query=True)[0] == S.MMSYSERR_NOTSUPPORTED        # MapTab hole
    assert pool.open(ac3, S.pcm_format(2, 48000), query=True)[0] == 0
    assert pool.open(ac3, S.pcm_format(1, 48000), query=True)[0] == 0
    assert pool.open(S.pcm_format(6, 48000), S.ac3_format(2, 48000, 384), query=True)[0] == S.ACMERR_NOTPOSSIBLE
    assert pool.open(S.pcm_format(2, 24000), S.ac3_format(2, 24000, 96), query=True)[0] == S.ACMERR_NOTPOSSIBLE
    assert pool.open(S.pcm_format(2, 48000), S.ac3_format(2, 48000, 200), query=True)[0] == S.MMSYSERR_NOTSUPPORTED
    assert pool.open(S.pcm_format(2, 48000), S.pcm_format(2, 48000), query=True)[0] == 0
    bad = S.pcm_format(2, 48000); bad.bits_per_sample = 8
    assert pool.open(bad, S.ac3_format(2, 48000, 192), query=True)[0] == S.ACMERR_NOTPOSSIBLE
    # frame size guessing (ac3_framesize, :432-488)
    assert S.framesize(S.ac3_format(6, 48000, 384, block_align=1536)) == 1536
    assert S.framesize(S.ac3_format(2, 44100, 192, block_align=1)) == 2 * 417
    assert S.framesize(S.ac3_format(2, 32000, 190, block_align=1)) == 2 * 576
    # stream_size (:2139-2363)
    rc, st = pool.open(ac3, S.pcm_format(6, 48000))
    assert rc == 0
    assert st.size(S.STREAMSIZEF_SOURCE, 1536 * 3 + 1) == (0, 4 * 1536 * 12)
    assert st.size(S.STREAMSIZEF_DESTINATION, 1536 * 12 * 2 + 5) == (0, 2 * 1538)
    assert st.size(S.STREAMSIZEF_DESTINATION, 256 * 12) == (0, 1538)
    assert st.size(S.STREAMSIZEF_DESTINATION, 256 * 12 - 1)[0] == S.ACMERR_NOTPOSSIBLE
    st.close()
    rc, st = pool.open(S.pcm_format(6, 48000), S.ac3_format(6, 48000, 384))
    assert rc == 0
    assert st.size(S.STREAMSIZEF_SOURCE, 1536 * 12 + 1) == (0, 2 * 1536)
    assert st.size(S.STREAMSIZEF_DESTINATION, 5000) == (0, 3 * 1536 * 12)
    st.close()
    # the pool hands its slots back
    opened = []
    for _ in range(16):
        rc, s_ = pool.open(ac3, S.pcm_format(2, 48000))
        assert rc == 0
        opened.append(s_)
    assert pool.open(ac3, S.pcm_format(2, 48000))[0] == S.MMSYSERR_NOMEM
    for s_ in opened:
        s_.close()
    rc, s_ = pool.open(ac3, S.pcm_format(2, 48000))
    assert rc == 0
    s_.close()


def test_slot_reuse_starts_clean(S, pool):
    """a closed stream's state slot goes to the next open, which must find a52_init's state in it"""
    fmt = (S.ac3_format(6, 48000, 384, block_align=1536), S.pcm_format(6, 48000))
    for seed in (301, 302):
        frames = H.orc_encode(H.gen_pcm(3, 6, seed=seed, kind="quiet" if seed == 302 else "tones"))   # quiet: dither state shows
        rc, st = pool.open(*fmt)
        assert rc == 0
        data = frames.tobytes()
        src = np.frombuffer(data, np.uint8).copy()
        dst = np.zeros(3 * 6 * 256 * 12, np.uint8)
        h = S.StreamHeader(src.ctypes.data, src.size, 0, dst.ctypes.data, dst.size, 0, S.STREAMCONVERTF_START)
        assert st.convert(h) == 0 and h.src_used == src.size and h.dst_used == dst.size
        want = _oracle_s16(frames, 7 | 16).reshape(-1).astype(np.int32)
        got = np.frombuffer(dst.tobytes(), np.int16).astype(np.int32)
        assert int(np.abs(got - want).max()) <= 1
        st.close()


def test_decode_downmix_follows_liba52_through_a_surround_level_change(S, pool):
    """The pool carries ac3mi_set_mix_state for every slot: a 3/2 stream whose surmixlev goes to "no surround" and back,
    decoded to stereo through the stream layer, gives liba52's samples (tests/test_mixlevel_switch.py has the details)."""
    d = np.load(os.path.join(H.GOLDEN, "mixflip.npz"), allow_pickle=False)
    frames = d["frames_a7_st"]
    fb = frames.shape[1]
    rc, st = pool.open(S.ac3_format(5, 48000, fb // 4, block_align=fb), S.pcm_format(2, 48000), S.ACM_DYNAMICRANGE)
    assert rc == 0
    out = bytearray()
    for f in range(len(frames)):                  # a frame per call, as a live stream arrives
        src = frames[f].copy()
        dst = np.zeros(6 * 256 * 2 * 2, np.uint8)
        h = S.StreamHeader(src.ctypes.data, fb, 0, dst.ctypes.data, dst.size, 0, S.STREAMCONVERTF_START if f == 0 else 0)
        assert st.convert(h) == 0 and h.src_used == fb
        out += dst[:h.dst_used].tobytes()
    st.close()
    want = _oracle_s16(frames, 2).reshape(-1)
    g = np.frombuffer(bytes(out), np.int16)
    assert g.size == want.size
    # random mantissas: loud, many saturated samples, and the sum of five planes is rounded in another order than liba52's
    # time-domain mix - two s16 steps (two float32 ulps at bias 384) at a handful of samples; a missed quirk is hundreds
    diff = np.abs(g.astype(np.int32) - want)
    assert int(diff.max()) <= 2 and int((diff > 1).sum()) <= diff.size // 200


def test_rounds_of_thousands_of_streams_cross_pcie_in_chunks(S, engine):
    """From 2 048 streams of one configuration on, a round's batch is cut into chunks whose copies and kernels overlap
    (stream.hip, decode_group / encode_group): every stream still gets exactly its own samples / frame, for two rounds in a
    row (carry-over state through the slot table), decode and encode."""
    n = 2300
    pool = S.Pool(engine, 2 * n)
    try:
        kinds = ("tones", "music", "noise", "bursts", "quiet", "strobe", "tones")
        frames = [H.orc_encode(H.gen_pcm(2, 6, seed=300 + i, kind=kinds[i])) for i in range(7)]
        want_dec = [_oracle_s16(fr, 7 | 16).reshape(2, -1) for fr in frames]                  # per frame
        pcms = [H.gen_pcm(2, 2, seed=400 + i, kind=kinds[i]) for i in range(7)]
        want_enc = [H.orc_encode(p, nch=2, bitrate=192000, chmap=(0, 1)) for p in pcms]
        dec = [pool.open(S.ac3_format(6, 48000, 384, block_align=1536), S.pcm_format(6, 48000))[1] for _ in range(n)]
        enc = [pool.open(S.pcm_format(2, 48000), S.ac3_format(2, 48000, 192))[1] for _ in range(n)]
        assert all(s is not None for s in dec + enc)
        for f in range(2):
            keep, hs = [], []
            for i in range(n):
                src = frames[i % 7][f].copy()
                dst = np.zeros(6 * 256 * 6 * 2, np.uint8)
                keep.append((src, dst))
                hs.append(S.StreamHeader(src.ctypes.data, src.size, 0, dst.ctypes.data, dst.size, 0, S.STREAMCONVERTF_START if f == 0 else 0))
            for i in range(n):
                src = np.frombuffer(pcms[i % 7][f * 1536:(f + 1) * 1536].tobytes(), np.uint8).copy()
                dst = np.zeros(4096, np.uint8)
                keep.append((src, dst))
                hs.append(S.StreamHeader(src.ctypes.data, src.size, 0, dst.ctypes.data, dst.size, 0, S.STREAMCONVERTF_START if f == 0 else 0))
            assert pool.convert_many(dec + enc, hs) == 0
            for i in range(n):
                assert hs[i].src_used == hs[i].src_len and hs[i].dst_used == keep[i][1].size
                g = np.frombuffer(keep[i][1].tobytes(), np.int16).astype(np.int32)
                assert int(np.abs(g - want_dec[i % 7][f].astype(np.int32)).max()) <= 1, (f, i)
            for i in range(n):
                h, (src, dst) = hs[n + i], keep[n + i]
                w = want_enc[i % 7][f]
                assert h.src_used == h.src_len and h.dst_used == w.size, (f, i, h.dst_used)
                assert np.array_equal(dst[:w.size], w), (f, i)
        for st in dec + enc:
            st.close()
    finally:
        pool.close()
