"""Randomised PCM-level decoder run (not collected by pytest): random packer streams (all channel modes, sample rates,
bsid 8..10, feature mixes; every third round with surround levels that change between frames), a random output request
(any channel layout, with / without LFE, with / without level adjustment), bias 0 or 384, float planes or the s16 flavour,
dynamic range on or off, the frames cut into calls at random, the mix state of ac3mi_set_mix_state set as the drop-in does
- GPU PCM against the oracle (pinned to liba52).  Requests liba52 would not grant are skipped.
    python tests/fuzz_pcm.py [n_rounds] [seed0]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import _harness as H          # noqa: E402
from tests import packer                 # noqa: E402


def main():
    import torch
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = H.pkg()
    eng = pkg.Engine(0)
    L = H.orc()
    rng = np.random.default_rng(seed0)
    bad = done = 0
    r = -1
    while done < rounds:
        r += 1
        acmod, lfe = int(rng.integers(0, 8)), int(rng.integers(0, 2))
        fscod, bsid, fsz = int(rng.integers(0, 3)), int(rng.choice([8, 8, 9, 10])), int(rng.integers(22, 38))
        S, F = int(rng.integers(1, 6)), int(rng.integers(1, 7))
        req = int(rng.integers(0, 11)) | (16 if rng.integers(0, 2) else 0) | (32 if rng.integers(0, 2) else 0)
        s16 = bool(rng.integers(0, 2))
        bias = 384.0 if (s16 or rng.integers(0, 2)) else 0.0
        dynoff = bool(rng.integers(0, 4) == 0)
        try:
            if r % 3 == 0 and (acmod & 4):
                streams = [packer.make_flip_stream(seed0 * 100000 + r * 50 + s, [int(x) for x in rng.choice([0, 1, 2, 2, 3], F)], acmod=acmod,
                                                   lfeon=lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz) for s in range(S)]
            else:
                streams = [packer.make_stream(seed0 * 100000 + r * 50 + s, F, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz) for s in range(S)]
        except Exception as e:      # the packer could not fit a frame at this size
            continue
        fb = streams[0].shape[1]
        desc = pkg.DecodeDesc(flags=req, level=1.0, bias=bias, dynrng=0 if dynoff else 1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
        try:
            n_out, _ = eng.decode_planes(desc)
        except Exception:
            continue                # a52_frame would refuse the request
        want0, errs, oflags = H.orc_decode(streams[0], req, 1.0, bias, dynrng_off=dynoff)
        if errs:
            continue
        want = np.stack([want0] + [H.orc_decode(streams[s], req, 1.0, bias, dynrng_off=dynoff)[0] for s in range(1, S)])
        if want.shape[3] != n_out:
            print("round %d: plane count %d vs %d" % (r, want.shape[3], n_out)); bad += 1; done += 1
            continue
        stride = (fb + 3) & ~3
        padded = np.zeros((S, F, stride), np.uint8)
        for s in range(S):
            padded[s, :, :fb] = streams[s]
        cuts = sorted(set([0, F] + [int(x) for x in rng.integers(1, max(F, 2), size=int(rng.integers(0, 3))) if x < F]))
        delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
        lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
        pend = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
        mflags = torch.zeros((S, 6), dtype=torch.int32, device="cuda")
        dev = torch.from_numpy(padded).cuda()
        got = []
        try:
            eng.set_mix_state(pend, mflags)
            for a, b in zip(cuts[:-1], cuts[1:]):
                fn = eng.decode_s16_batch if s16 else eng.decode_batch
                pcm, status = fn(desc, dev[:, a:b].contiguous(), delay, lfsr)
                eng.sync()
                assert (status.cpu().numpy() & 0x1ff).max() == 0
                got.append(pcm.cpu().numpy())
        finally:
            eng.set_mix_state(None, None)
        got = np.concatenate(got, axis=1)
        if s16:
            w = np.zeros((S, F, 6, 256, n_out), np.int16)
            for s_ in range(S):
                for f_ in range(F):
                    for b_ in range(6):
                        L.orc_convert_s16(H.P(np.ascontiguousarray(want[s_, f_, b_]), H.fp), H.P(w[s_, f_, b_], H.i16p), oflags)
            d = np.abs(got.astype(np.int32) - w.astype(np.int32))
            ok = int(d.max()) <= 2 and int((d > 1).sum()) <= max(d.size // 200, 4)
            note = "max step %d, %d of %d over 1" % (int(d.max()), int((d > 1).sum()), d.size)
        else:
            w = want.reshape(got.shape)
            err = got.astype(np.float64) - w
            tol = 40.0 if bias else 1.0             # float32 resolution at 384
            scale_rms, scale_max = max(1.0, H.rms(w - bias)), max(1.0, float(np.abs(w - bias).max()))
            ok = H.rms(err) <= tol * 1e-6 * scale_rms and np.abs(err).max() <= tol * 1e-5 * scale_max
            note = "rms %.2e max %.2e (scale %.1f / %.1f)" % (H.rms(err), float(np.abs(err).max()), scale_rms, scale_max)
        print("round %3d acmod %d lfe %d fscod %d bsid %2d size %2d request %2d -> %2d bias %3d %s dynrng %d, %d x %d, calls %s: %s%s"
              % (done, acmod, lfe, fscod, bsid, fsz, req, oflags, int(bias), "s16" if s16 else "f32", 0 if dynoff else 1, S, F, cuts,
                 "ok" if ok else "MISMATCH ", "" if ok else note), flush=True)
        bad += not ok
        done += 1
    print("mismatching rounds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
