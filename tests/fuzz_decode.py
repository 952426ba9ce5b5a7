"""Extended randomised decoder parity run (not collected by pytest): random packer streams over all channel modes,
sample rates, bsid 8..10 and feature mixes, GPU coefficient planes bit-exact against the oracle.
    python tests/fuzz_decode.py [n_rounds] [seed0]"""
import ctypes
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
    L = H.orc()
    L.orc_a52_get_coefs.argtypes = [H.vp, H.fp, H.u8p]
    rng = np.random.default_rng(seed0)
    bad = 0
    for r in range(rounds):
        acmod = int(rng.integers(0, 8))
        lfe = int(rng.integers(0, 2))
        fscod = int(rng.integers(0, 3))
        bsid = int(rng.choice([8, 8, 9, 10]))
        fsz = int(rng.integers(20, 38))
        S, F = 6, 3
        try:
            frames = np.stack([packer.make_stream(seed0 * 100000 + r * 97 + s, F, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz)
                               for s in range(S)])
        except Exception as e:      # the packer could not fit a frame at this size
            print("round %d skipped: %s" % (r, e))
            continue
        fb = frames.shape[2]
        stride = (fb + 3) & ~3
        padded = np.zeros((S, F, stride), np.uint8)
        padded[:, :, :fb] = frames
        nf = H.NFCHANS[acmod]
        flags = acmod | (16 if lfe else 0)
        want_coef = np.zeros((S, F, 6, 6, 256), np.float32)
        want_sw = np.zeros((S, F, 6, 5), np.uint8)
        want_lfsr = np.zeros(S, np.int64)
        for s in range(S):
            st = L.orc_a52_init()
            buf = np.zeros(F * fb + 64, np.uint8)
            buf[:F * fb] = frames[s].reshape(-1)
            for f in range(F):
                fl, lv = H.ci(flags), H.cf(1.0)
                assert L.orc_a52_frame(st, ctypes.cast(buf.ctypes.data + f * fb, H.u8p), ctypes.byref(fl), ctypes.byref(lv), 0.0) == 0
                for b in range(6):
                    assert L.orc_a52_block(st) == 0
                    L.orc_a52_get_coefs(st, H.P(want_coef[s, f, b], H.fp), H.P(want_sw[s, f, b], H.u8p))
            want_lfsr[s] = L.orc_a52_get_lfsr(st)
            L.orc_a52_free(st)
        desc = pkg.DecodeDesc(flags=flags, level=1.0, bias=0.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
        n_out, _ = eng.decode_planes(desc)
        delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
        lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
        pcm, status, taps = eng.decode_batch(desc, torch.from_numpy(padded).cuda(), delay, lfsr, taps=True)
        eng.sync()
        got = taps["coef"].cpu().numpy()
        ok = (status.cpu().numpy() & 0x1ff).max() == 0
        ok = ok and np.array_equal(lfsr.cpu().numpy().astype(np.int64) & 0xffff, want_lfsr)
        ok = ok and np.array_equal(got[:, :, :, lfe:lfe + nf].view(np.uint32), want_coef[:, :, :, lfe:lfe + nf].view(np.uint32))
        if lfe:
            ok = ok and np.array_equal(got[:, :, :, 0].view(np.uint32), want_coef[:, :, :, 0].view(np.uint32))
        print("round %3d acmod %d lfe %d fscod %d bsid %2d frmsizecod %2d: %s" % (r, acmod, lfe, fscod, bsid, fsz, "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("mismatching rounds:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
