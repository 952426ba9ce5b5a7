"""Extended randomised encoder parity run (not collected by pytest): random channel counts, sample rates, bit rates
and signal kinds, multi-frame streams; GPU bitstreams byte-exact against the encoder oracle.
    python tests/fuzz_encode.py [n_rounds] [seed0]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import _harness as H                      # noqa: E402
from tests.test_encode_gpu import _gpu, _oracle      # noqa: E402

KBPS = (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640)
KINDS = ("tones", "noise", "quiet", "music", "bursts", "strobe")


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = H.pkg()
    eng = pkg.Engine(0)
    rng = np.random.default_rng(seed0)
    bad = 0
    done = 0
    while done < rounds:
        nch = int(rng.integers(1, 7))
        freq = int(rng.choice([48000, 44100, 32000, 24000, 22050, 16000]))
        kb = int(rng.choice(KBPS))
        shift = 0 if freq >= 32000 else 1
        bitrate = (kb * 1000) >> shift
        if pkg.EncodeDesc(freq, bitrate, nch).frame_bytes() == 0 or kb < 32 * nch:
            continue
        chmap = H.CHMAP6 if nch == 6 else tuple(range(8))
        pcm = [H.gen_pcm(4, nch, seed=seed0 * 1000 + done * 7 + s, kind=KINDS[int(rng.integers(0, len(KINDS)))]) for s in range(4)]
        try:
            want, _ = _oracle(pcm, nch, bitrate, freq, chmap)
        except AssertionError as e:
            print("config %d ch %d Hz %d bps rejected by the oracle: %s" % (nch, freq, bitrate, e))
            continue
        got, _ = _gpu(eng, pcm, nch, bitrate, freq, chmap, taps=False)
        ok = np.array_equal(got, want)
        print("round %3d: %d ch %5d Hz %6d bps: %s" % (done, nch, freq, bitrate, "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
        done += 1
    print("mismatching rounds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
