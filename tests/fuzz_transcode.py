"""Randomised run of ac3mi_transcode_batch (not collected by pytest): random packer streams (every second round with
surround levels that change), a random granted output request - downmixes included - re-encoded with as many channels at a
random bit rate, the mix state set as the stream layer does; the one-call transcoder against decode-to-s16 followed by
encode (two calls, same carry-over state): frames, status words and all state arrays bit for bit.
    python tests/fuzz_transcode.py [n_rounds] [seed0]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import _harness as H          # noqa: E402
from tests import packer                 # noqa: E402

KBPS = (32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384, 448, 512, 576, 640)
RATES = (48000, 44100, 32000)


def main():
    import torch
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = H.pkg()
    eng = pkg.Engine(0)
    rng = np.random.default_rng(seed0)
    bad = done = 0
    r = -1
    while done < rounds:
        r += 1
        acmod, lfe = int(rng.integers(1, 8)), int(rng.integers(0, 2))
        fscod, fsz = int(rng.integers(0, 3)), int(rng.integers(24, 38))
        S, F = int(rng.integers(1, 9)), int(rng.integers(1, 5))
        req = int(rng.choice([acmod, 2, 1, 10, 3, 6, 7])) | (16 if lfe and rng.integers(0, 2) else 0) | 32
        try:
            if r % 2 and (acmod & 4):
                streams = [packer.make_flip_stream(seed0 * 100000 + r * 50 + s, [int(x) for x in rng.choice([0, 1, 2, 2, 3], F)], acmod=acmod,
                                                   lfeon=lfe, fscod=fscod, frmsizecod=fsz) for s in range(S)]
            else:
                streams = [packer.make_stream(seed0 * 100000 + r * 50 + s, F, acmod, lfe, fscod=fscod, frmsizecod=fsz) for s in range(S)]
        except Exception:
            continue
        fb = streams[0].shape[1]
        dec = pkg.DecodeDesc(flags=req, level=1.0, bias=384.0, dynrng=int(rng.integers(0, 2)), acmod=acmod, lfeon=lfe, frame_bytes=fb)
        try:
            n_out, oflags = eng.decode_planes(dec)
        except Exception:
            continue
        kb = int(rng.choice(KBPS))
        enc = pkg.EncodeDesc(RATES[fscod], kb * 1000, n_out)
        if kb < 32 * n_out or enc.frame_bytes() == 0:
            continue
        chmap = [0, 1, 2, 3, 4, 5, 6, 7]
        if n_out in (3, 5):
            chmap[1], chmap[2] = 2, 1
        if n_out == 6:
            chmap[:6] = [0, 2, 1, 4, 5, 3]
        chmap = tuple(chmap[:n_out])
        stride = (fb + 3) & ~3
        padded = np.zeros((S, F, stride), np.uint8)
        for s in range(S):
            padded[s, :, :fb] = streams[s]
        dev = torch.device("cuda:0")
        frames_t = torch.from_numpy(padded).to(dev)

        def fresh():
            return (torch.zeros((S, n_out, 128), dtype=torch.float32, device=dev), torch.ones((S,), dtype=torch.int16, device=dev),
                    torch.zeros((S, n_out, 256), dtype=torch.int16, device=dev), torch.full((S,), 40, dtype=torch.int32, device=dev),
                    torch.zeros((S, n_out, 128), dtype=torch.float32, device=dev), torch.zeros((S, 6), dtype=torch.int32, device=dev))
        try:
            d1, l1, la1, c1, p1, f1 = fresh()
            eng.set_mix_state(p1, f1)
            s16, st1 = eng.decode_s16_batch(dec, frames_t, d1, l1)
            eng.sync()
            out1 = eng.encode_batch(enc, s16.view(S, F, 1536, n_out), chmap, la1, c1)
            eng.sync()
            d2, l2, la2, c2, p2, f2 = fresh()
            eng.set_mix_state(p2, f2)
            out2, st2 = eng.transcode_batch(dec, enc, frames_t, d2, l2, chmap, la2, c2)
            eng.sync()
        finally:
            eng.set_mix_state(None, None)
        pairs = (("frames", out1, out2), ("status", st1, st2), ("delay", d1, d2), ("lfsr", l1, l2), ("last", la1, la2), ("csnr", c1, c2),
                 ("pending", p1, p2), ("mixflags", f1, f2))
        diff = [n for n, a, b in pairs if not torch.equal(a.cpu(), b.cpu())]
        ok = not diff and int((st1.cpu() & 0x1ff).max()) == 0
        print("round %3d acmod %d lfe %d %5d Hz size %2d request %2d -> %d ch, encode %3d kbps, %d x %d%s: %s%s"
              % (done, acmod, lfe, RATES[fscod], fsz, req, n_out, kb, S, F, " flip" if (r % 2 and acmod & 4) else "", "ok" if ok else "MISMATCH ", ",".join(diff)), flush=True)
        bad += not ok
        done += 1
    print("mismatching rounds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
