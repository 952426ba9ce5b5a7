"""CPU: host-only parts of the byte-stream layer (include/ac3mi_stream.h) and the Python model the GPU tests
check it against (tests/stream_model.py).  No compute calls."""
import ctypes
import importlib

import numpy as np

from tests import _harness as H
from tests import stream_model as M


def _S():
    return importlib.import_module("ac-3-acm-codec_amd.stream")


def test_framesize_guessing_needs_no_gpu():
    S = _S()
    # ac3_framesize (src/AC3ACM.cpp:432-488): nBlockAlign if it is a frame size, else the nearest bit rate
    assert S.framesize(S.ac3_format(6, 48000, 384, block_align=1536)) == 1536
    assert S.framesize(S.ac3_format(6, 48000, 448, block_align=1)) == 1792
    assert S.framesize(S.ac3_format(2, 44100, 128, block_align=1)) == 2 * 278
    assert S.framesize(S.ac3_format(2, 32000, 640, block_align=4)) == 2 * 1920
    assert S.framesize(S.ac3_format(2, 48000, 100, block_align=1)) == 2 * 192          # 12500 B/s: nearest is 96 kbps
    # 24 kHz and 12 kHz: (rate >> 6) & 3 == 3 selects the kbps column of the reference's table (AC3ACM.cpp:128-149,
    # 445, 455, 472): by hand, 192 kbps (24000 B/s) -> 2 * 192, block_align 640 = 2 * 320 is accepted as is,
    # 9000 B/s is nearest to 72 -> 64 or 80 kbps: |9000 - 8000| = |9000 - 10000|, the first (64) wins -> 128;
    # above 81000 B/s the worst case 2 * 640
    assert S.framesize(S.ac3_format(2, 24000, 192, block_align=1)) == 2 * 192
    assert S.framesize(S.ac3_format(2, 12000, 96, block_align=640)) == 640
    assert S.framesize(S.ac3_format(2, 24000, 72, block_align=1)) == 2 * 64
    assert S.framesize(S.ac3_format(6, 12000, 700, block_align=0)) == 2 * 640
    # the other half / quarter rates hash to columns 0..2 (22050, 11025 -> 0; 16000 -> 2; 8000 -> 1)
    assert S.framesize(S.ac3_format(2, 22050, 128, block_align=1)) == 2 * 384
    assert S.framesize(S.ac3_format(2, 16000, 128, block_align=1)) == 2 * 256
    assert S.framesize(S.ac3_format(2, 8000, 128, block_align=1)) == 2 * 278
    assert S._lib().ac3mi_pool_create(None, 4) is None


def test_model_syncinfo_matches_engine_and_oracle():
    pkg = H.pkg()
    L = H.orc()
    rng = np.random.default_rng(11)
    hdr0 = H.orc_encode(H.gen_pcm(1, 2, seed=2, kind="tones"), nch=2, bitrate=192000, chmap=(0, 1))[0, :8].copy()
    for _ in range(2000):
        hdr = hdr0.copy()
        k = rng.integers(0, 3)
        if k == 0:
            hdr[4] = rng.integers(0, 256)
        elif k == 1:
            hdr[5] = rng.integers(0, 256)
        else:
            hdr[rng.integers(0, 8)] = rng.integers(0, 256)
        fl, sr, br = H.ci(), H.ci(), H.ci()
        want = L.orc_a52_syncinfo(H.P(hdr, H.u8p), ctypes.byref(fl), ctypes.byref(sr), ctypes.byref(br))
        assert M.syncinfo_size(bytes(hdr)) == want
        assert pkg.syncinfo(hdr)[0] == want


def test_model_consumes_everything_and_counts_blocks():
    rng = np.random.default_rng(5)
    frames = H.orc_encode(H.gen_pcm(6, 6, seed=8, kind="tones"))
    data = bytes(29) + frames.tobytes()
    for trial in range(20):
        m = M.DecodeModel(6, 6)
        pos, out, first = 0, 0, True
        for _ in range(400):
            n = int(min(len(data) - pos, rng.integers(0, 3000)))
            su, du = m.convert(data[pos:pos + n], int(rng.choice([3072, 6144, 20000])), first)
            assert su <= n and du % 3072 == 0
            pos += su
            out += du
            first = False
        assert pos == len(data) and out == 6 * 6 * 3072
        e = M.EncodeModel(6, 1536)
        pos, out, first = 0, 0, True
        total = 5 * 18432 + 777
        for _ in range(400):
            n = int(min(total - pos, rng.integers(0, 40000)))
            su, du = e.convert(n, int(rng.choice([100, 1536, 5000])), first)
            pos += su
            out += du
            first = False
        assert pos == total and out == 5 * 1536
