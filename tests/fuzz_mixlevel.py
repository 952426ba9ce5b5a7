"""Randomised run of the surround-level bookkeeping (not collected by pytest; tests/test_mixlevel_switch.py has the fixed
cases): streams whose surmixlev changes at random, random acmod (with surround channels) / LFE / MONO-STEREO-3F request,
random partition of the frames into calls, several streams per call - GPU with ac3mi_set_mix_state against the oracle,
which tests/test_mixlevel_switch.py pins to the real liba52.
    python tests/fuzz_mixlevel.py [n_rounds] [seed0]"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import _harness as H          # noqa: E402
from tests import packer                 # noqa: E402


def main():
    import torch
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    pkg = H.pkg()
    eng = pkg.Engine(0)
    rng = np.random.default_rng(seed0)
    bad = 0
    for r in range(rounds):
        acmod = int(rng.choice([4, 5, 6, 7]))
        lfe = int(rng.integers(0, 2))
        out = int(rng.choice([1, 2, 3] if acmod & 1 else [1, 2]))
        S, F = int(rng.integers(1, 7)), int(rng.integers(2, 9))
        seqs = [[int(x) for x in rng.choice([0, 1, 2, 2, 2, 3], F)] for _ in range(S)]
        try:
            streams = [packer.make_flip_stream(seed0 * 100000 + r * 50 + s, seqs[s], acmod=acmod, lfeon=lfe) for s in range(S)]
        except RuntimeError as e:
            print("round %d skipped: %s" % (r, e))
            continue
        fb = streams[0].shape[1]
        stride = (fb + 3) & ~3
        padded = np.zeros((S, F, stride), np.uint8)
        for s in range(S):
            padded[s, :, :fb] = streams[s]
        req = out | (16 if lfe and rng.integers(0, 2) else 0)
        s16 = bool(r & 1)                                 # every other round: the s16 flavour of the transform (bias 384)
        bias = 384.0 if s16 else 0.0
        want = np.stack([H.orc_decode(streams[s], req, 1.0, bias)[0] for s in range(S)])
        oflags = H.orc_decode(streams[0][:1], req, 1.0, bias)[2]
        desc = pkg.DecodeDesc(flags=req, level=1.0, bias=bias, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
        n_out, _ = eng.decode_planes(desc)
        cuts = sorted(set([0, F] + [int(x) for x in rng.integers(1, F, size=int(rng.integers(0, 4)))]))
        delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
        lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
        pend = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
        mflags = torch.zeros((S, 6), dtype=torch.int32, device="cuda")
        dev = torch.from_numpy(padded).cuda()
        got = []
        try:
            if not __import__('os').environ.get('FUZZ_NO_MIXSTATE'):
                eng.set_mix_state(pend, mflags)
            for a, b in zip(cuts[:-1], cuts[1:]):
                if s16:
                    pcm, status = eng.decode_s16_batch(desc, dev[:, a:b].contiguous(), delay, lfsr)
                else:
                    pcm, status = eng.decode_batch(desc, dev[:, a:b].contiguous(), delay, lfsr)
                eng.sync()
                assert (status.cpu().numpy() & 0x1ff).max() == 0
                got.append(pcm.cpu().numpy())
        finally:
            eng.set_mix_state(None, None)
        got = np.concatenate(got, axis=1)
        if s16:
            # the oracle's planes through the reference's converter; loud random content: two s16 steps at a few samples
            L = H.orc()
            nout = want.shape[3]
            w = np.zeros((S, F, 6, 256, nout), np.int16)
            for s_ in range(S):
                for f_ in range(F):
                    for b_ in range(6):
                        L.orc_convert_s16(H.P(np.ascontiguousarray(want[s_, f_, b_]), H.fp), H.P(w[s_, f_, b_], H.i16p), oflags)
            d = np.abs(got.astype(np.int32) - w.astype(np.int32))
            ok = int(d.max()) <= 2 and int((d > 1).sum()) <= d.size // 200
            if not ok:
                print("   s16: max step %d, %d of %d samples off by more than 1, more than 2: %d" % (int(d.max()), int((d > 1).sum()), d.size, int((d > 2).sum())))
            err = d.transpose(0, 1, 2, 4, 3).astype(np.float64)
            scale_max = 1e5 * 3.0
        else:
            w = want.reshape(got.shape)
            err = got.astype(np.float64) - w
            scale_rms, scale_max = max(1.0, H.rms(w)), max(1.0, float(np.abs(w).max()))
            ok = H.rms(err) <= 1e-6 * scale_rms and np.abs(err).max() <= 1e-5 * scale_max
        print("round %3d acmod %d lfe %d request %2d, %d streams x %d frames, calls at %s: %s" % (r, acmod, lfe, req, S, F, cuts, "ok" if ok else "MISMATCH"), flush=True)
        if not ok:
            pb = np.abs(err).max(axis=(3, 4))
            print("   first blocks off:", np.argwhere(pb > 1e-5 * scale_max)[:6].tolist(), "levels", seqs)
        bad += not ok
    print("mismatching rounds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
