"""Extended randomised run for the fused mantissa + transform kernel (decode_mx.hip; not collected by pytest): random packer
streams over all channel modes, sample rates, bsid 8..10, frame sizes and feature mixes, every frame a one-frame stream with
random overlap and dither state coming in, some frames damaged.  ac3mi_set_decode_mode 6 (fused) against 4 (mantissa kernel,
planes through HBM, transform kernel): float PCM, s16 PCM, status, overlap state, dither state - bit for bit.
    python tests/fuzz_mantx.py [n_rounds] [seed0]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import _harness as H          # noqa: E402
from tests import packer                 # noqa: E402


def main():
    import torch
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = H.pkg()
    eng = pkg.Engine(0)
    rng = np.random.default_rng(seed0)
    bad = 0
    for r in range(rounds):
        acmod = int(rng.integers(0, 8))
        lfe = int(rng.integers(0, 2))
        fscod = int(rng.integers(0, 3))
        bsid = int(rng.choice([8, 8, 9, 10]))
        fsz = int(rng.integers(20, 38))
        S0, F = 8, 3
        try:
            frames = np.stack([packer.make_stream(seed0 * 100000 + r * 89 + s, F, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz)
                               for s in range(S0)])
        except Exception as e:      # the packer could not fit a frame at this size
            print("round %d skipped: %s" % (r, e))
            continue
        fb = frames.shape[2]
        stride = (fb + 3) & ~3
        S = S0 * F
        padded = np.zeros((S, 1, stride), np.uint8)
        padded[:, 0, :fb] = frames.reshape(S, fb)
        for _ in range(int(rng.integers(0, 3))):            # damage
            s = int(rng.integers(0, S))
            at = int(rng.integers(0, fb - 4))
            padded[s, 0, at:at + 4] ^= rng.integers(1, 255, 4).astype(np.uint8)
        d_frames = torch.from_numpy(padded).cuda()
        flags = acmod | (16 if lfe else 0)
        dyn = int(rng.integers(0, 2))
        res = {}
        for mode in (4, 6):
            eng.set_decode_mode(mode)
            out = []
            for bias, s16 in ((0.0, False), (384.0, True)):
                desc = pkg.DecodeDesc(flags=flags | 32, level=1.0, bias=bias, dynrng=dyn, acmod=acmod, lfeon=lfe, frame_bytes=fb)
                n_out, _ = eng.decode_planes(desc)
                g = torch.Generator().manual_seed(r)
                delay = ((torch.rand((S, n_out, 128), generator=g) - 0.5) * 0.25).cuda()
                lfsr = torch.randint(0, 65536, (S,), generator=g).to(torch.int16).cuda()
                if s16:
                    pcm, status = eng.decode_s16_batch(desc, d_frames, delay, lfsr)
                else:
                    pcm, status = eng.decode_batch(desc, d_frames, delay, lfsr)
                eng.sync()
                out += [x.cpu().numpy() for x in (pcm, status, delay, lfsr)]
            res[mode] = out
        same = all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(res[4], res[6]))
        failed = int(np.count_nonzero(res[4][1] & 0x1ff))
        if not same:
            bad += 1
            print("round %d MISMATCH acmod %d lfe %d fscod %d bsid %d fsz %d" % (r, acmod, lfe, fscod, bsid, fsz))
        elif r % 20 == 0:
            print("round %d ok (acmod %d lfe %d fscod %d bsid %d frame %d bytes, %d frames failed)" % (r, acmod, lfe, fscod, bsid, fb, failed), flush=True)
    eng.set_decode_mode(0)
    print("mismatching rounds: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
