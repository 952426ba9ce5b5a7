"""Damaged-frame parity run (not collected by pytest; tests/test_decode_gpu.py::test_damaged_frames_match_liba52_block_by_block
runs one round of it): packer frames with random bit flips after the acmod field.  For every frame the GPU must report
the same first failing block as the oracle (a52_block returning 1, L52/parse.c:228-771), produce bit-identical
coefficient planes for the blocks before it and leave the dither generator where the oracle leaves it.
    python tests/fuzz_corrupt.py [n_rounds] [seed0] [acmod]      (acmod given: every round uses it, e.g. 2 for rematrixing)"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tests import _harness as H          # noqa: E402
from tests import packer                 # noqa: E402


def make_damaged(seed, acmod, lfe, S=96, fscod=0, bsid=8, frmsizecod=30, base_streams=6):
    """S one-frame streams (the first `base_streams` undamaged) and what the oracle makes of them."""
    L = H.orc()
    L.orc_a52_get_coefs.argtypes = [H.vp, H.fp, H.u8p]
    rng = np.random.default_rng(seed)
    base = [packer.make_stream(seed * 1000 + s, 1, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=frmsizecod)[0] for s in range(base_streams)]
    fb = base[0].shape[0]
    frames = np.stack([base[s % base_streams] for s in range(S)]).copy()
    for s in range(S):
        if s < base_streams:
            continue                                    # undamaged controls
        bits = np.unpackbits(frames[s])
        n = int(rng.integers(1, 9))
        lo = 51 if rng.integers(0, 4) else 200          # mostly anywhere after acmod; sometimes audio blocks only
        pos = rng.integers(lo, fb * 8 - 16, size=n)
        if rng.integers(0, 3) == 0:                      # a burst instead of isolated flips
            p0 = int(rng.integers(lo, fb * 8 - 80))
            pos = np.arange(p0, p0 + int(rng.integers(8, 64)))
            bits[pos] = rng.integers(0, 2, size=pos.size)
        else:
            bits[pos] ^= 1
        frames[s] = np.packbits(bits)
    flags = acmod | (16 if lfe else 0)
    want_coef = np.zeros((S, 6, 6, 256), np.float32)
    want_fail = np.full(S, 6)                            # first failing block (6 = none)
    want_foreign = np.zeros(S, bool)
    want_lfsr = np.zeros(S, np.int64)
    gap = np.zeros((S, 6, 6, 256), bool)                 # bins liba52 leaves untouched (see orc_a52_get_layout): engine = 0
    lay = (ctypes.c_int * 8)()
    L.orc_a52_get_layout.argtypes = [H.vp, ctypes.POINTER(ctypes.c_int)]
    L.orc_a52_get_layout.restype = None
    sw = np.zeros(5, np.uint8)
    for s in range(S):
        st = L.orc_a52_init()
        buf = np.zeros(fb + 64, np.uint8)
        buf[:fb] = frames[s]
        fl, lv = H.ci(flags), H.cf(1.0)
        rc = L.orc_a52_frame(st, H.P(buf, H.u8p), ctypes.byref(fl), ctypes.byref(lv), 0.0)
        # lfeon as the frame carries it (a52_syncinfo reads the bit; a52_frame only reports it when LFE output was requested)
        sflags, srate, brate = H.ci(), H.ci(), H.ci()
        L.orc_a52_syncinfo(H.P(buf, H.u8p), ctypes.byref(sflags), ctypes.byref(srate), ctypes.byref(brate))
        if rc != 0 or (1 if sflags.value & 16 else 0) != lfe:
            want_foreign[s] = True                       # lfeon flipped: not the batch's configuration
        else:
            for b in range(6):
                if L.orc_a52_block(st):
                    want_fail[s] = b
                    break
                L.orc_a52_get_coefs(st, H.P(want_coef[s, b], H.fp), H.P(sw, H.u8p))
                L.orc_a52_get_layout(st, lay)
                for c in range(H.NFCHANS[acmod]):
                    if (lay[7] >> c) & 1 and lay[c] < lay[5]:
                        gap[s, b, c + lfe, lay[c]:lay[5]] = True
        want_lfsr[s] = L.orc_a52_get_lfsr(st)
        L.orc_a52_free(st)
    want_coef[gap] = 0.0
    # stereo: rematrixing (parse.c:837-865) mixes a gap bin of one channel - whatever liba52's buffer held - into BOTH
    # channels: neither is defined there.  `ignore` marks those bins; the caller blanks them on both sides.
    ignore = np.zeros_like(gap)
    if acmod == 2:
        both = gap[:, :, lfe, 13:] | gap[:, :, lfe + 1, 13:]
        ignore[:, :, lfe, 13:] = both
        ignore[:, :, lfe + 1, 13:] = both
    want_coef[ignore] = 0.0
    make_damaged.ignore = ignore
    return frames, want_coef, want_fail, want_foreign, want_lfsr


def damaged_round(eng, seed, acmod, lfe, **kw):
    """One batch on the GPU.  Returns (n_mismatching_frames, n_failed_frames, n_foreign_frames)."""
    import torch
    pkg = H.pkg()
    frames, want_coef, want_fail, want_foreign, want_lfsr = make_damaged(seed, acmod, lfe, **kw)
    S, fb = frames.shape
    nf = H.NFCHANS[acmod]
    flags = acmod | (16 if lfe else 0)
    stride = (fb + 3) & ~3
    padded = np.zeros((S, 1, stride), np.uint8)
    padded[:, 0, :fb] = frames
    desc = pkg.DecodeDesc(flags=flags, level=1.0, bias=0.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
    n_out, _ = eng.decode_planes(desc)
    delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
    pcm, status, taps = eng.decode_batch(desc, torch.from_numpy(padded).cuda(), delay, lfsr, taps=True)
    eng.sync()
    status = status.cpu().numpy()[:, 0]
    got = taps["coef"].cpu().numpy()[:, 0]
    got[make_damaged.ignore[:, :, :got.shape[2]]] = 0.0
    got_lfsr = lfsr.cpu().numpy().astype(np.int64) & 0xffff
    pcm = pcm.cpu().numpy()[:, 0]
    bad = 0
    for s in range(S):
        ok = True
        if want_foreign[s]:
            ok = bool(status[s] & 0x100) and np.abs(pcm[s]).max() == 0.0
        else:
            f = int(want_fail[s])
            ok = (status[s] & 0x1ff) == ((0x3f << f) & 0x3f)
            ok = ok and got_lfsr[s] == want_lfsr[s]
            lo = 1 if lfe else 0
            ok = ok and np.array_equal(got[s, :f, lo:lo + nf].view(np.uint32), want_coef[s, :f, lo:lo + nf].view(np.uint32))
            if lfe:
                ok = ok and np.array_equal(got[s, :f, 0].view(np.uint32), want_coef[s, :f, 0].view(np.uint32))
            ok = ok and np.abs(got[s, f:]).max(initial=0.0) == 0.0 and np.isfinite(pcm[s]).all()
        if not ok:
            print("  stream %d: status %#x, oracle first failing block %d foreign %d" % (s, status[s], want_fail[s], want_foreign[s]), flush=True)
        bad += not ok
    return bad, int((want_fail < 6).sum()), int(want_foreign.sum())


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only_acmod = int(sys.argv[3]) if len(sys.argv) > 3 else None
    eng = H.pkg().Engine(0)
    rng = np.random.default_rng(seed0)
    total = 0
    for r in range(rounds):
        acmod, lfe = int(rng.integers(0, 8)), int(rng.integers(0, 2))
        if only_acmod is not None:
            acmod = only_acmod
        fscod, bsid = int(rng.integers(0, 3)), int(rng.choice([8, 8, 9, 10]))
        fsz = int(rng.integers(24, 38))
        try:
            bad, failed, foreign = damaged_round(eng, seed0 * 7919 + r, acmod, lfe, fscod=fscod, bsid=bsid, frmsizecod=fsz)
        except RuntimeError as e:                        # the packer could not fit a frame at this size
            print("round %d skipped: %s" % (r, e))
            continue
        print("round %3d acmod %d lfe %d fscod %d bsid %2d frmsizecod %2d: %d frames failed a block, %d refused, %d mismatches"
              % (r, acmod, lfe, fscod, bsid, fsz, failed, foreign, bad), flush=True)
        total += bad
    print("mismatching frames:", total)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
