"""Streams whose surround mix level changes between frames (tests/packer.make_flip_stream).

Valid AC-3 that no encoder emits - and the one place where liba52 is not a linear mixer: with surmixlev "no surround"
(slev 0) and a MONO / STEREO / 3F request it leaves the surround channels out of transform and mix
(a52dec-0.7.5-cvs/liba52/parse.c:900-913, downmix.c:494-583); their overlap tails are then dropped or wait in their planes
until the level comes back, depending on a52_block's synthesis path (parse.c:884-937) and its `downmixed` flag.

CPU: the oracle equals the REAL liba52 bit for bit on such streams (fixture tests/golden/mixflip.npz everywhere, a
fresh sweep where oracle/_ref exists).  GPU: with ac3mi_set_mix_state the engine equals liba52 too; without it (plain
linear mix) it differs exactly in the first block after a change of the level to or from zero, nowhere else.
"""
import os

import numpy as np
import pytest

from tests import _harness as H
from tests import packer

TAGS = ["a7_st", "a7_mono", "a7_dolby", "a6_st", "a5_st", "a4_mono", "a6_st_b384", "a7_3f_b384", "a4_st_b384"]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@pytest.mark.parametrize("tag", TAGS)
def test_oracle_reproduces_liba52_when_the_surround_level_changes(tag):
    d = np.load(os.path.join(H.GOLDEN, "mixflip.npz"), allow_pickle=False)
    flags, oflags = (int(x) for x in d["args_" + tag])
    pcm, errs, out = H.orc_decode(d["frames_" + tag], flags, 1.0, float(d["bias_" + tag][0]))
    assert errs == 0 and out == oflags
    assert np.array_equal(_bits(pcm), _bits(d["pcm_" + tag]))


@pytest.mark.skipif(not H.have_ref(), reason="oracle/_ref/liba52_ref.so not built")
def test_oracle_matches_liba52_on_fresh_level_changes():
    n = 0
    for seed in range(6):
        for acmod in (4, 5, 6, 7):
            levels = [1, 1, 2, 2, 0, 0, 2, 1] if seed % 2 else [0, 2, 2, 1, 2, 0, 0, 2]
            fr = packer.make_flip_stream(1000 + seed, levels, acmod=acmod)
            for flags in (2, 1, 10, 3):
                a, ea, fa = H.ref_decode(fr, flags, 1.0, 0.0)
                b, eb, fb = H.orc_decode(fr, flags, 1.0, 0.0)
                assert ea == 0 and eb == 0 and fa == fb
                assert np.array_equal(_bits(a), _bits(b)), (seed, acmod, flags)
                n += 1
    assert n == 96


# ---------------------------------------------------------------------------------------------------

def _gpu_decode(engine, frames, acmod, flags, frames_per_call, mix_state, bias=0.0):
    """frames [F][bytes] of one stream -> PCM [F][6][n_out][256], decoded in calls of `frames_per_call` frames"""
    import torch
    pkg = H.pkg()
    F, fb = frames.shape
    stride = (fb + 3) & ~3
    padded = np.zeros((1, F, stride), np.uint8)
    padded[0, :, :fb] = frames
    desc = pkg.DecodeDesc(flags=flags, level=1.0, bias=bias, dynrng=1, acmod=acmod, lfeon=0, frame_bytes=fb)
    n_out, _ = engine.decode_planes(desc)
    delay = torch.zeros((1, n_out, 128), dtype=torch.float32, device="cuda")
    lfsr = torch.ones((1,), dtype=torch.int16, device="cuda")
    pend = torch.zeros((1, n_out, 128), dtype=torch.float32, device="cuda")
    mflags = torch.zeros((1, 6), dtype=torch.int32, device="cuda")
    dev = torch.from_numpy(padded).cuda()
    out = []
    try:
        if mix_state:
            engine.set_mix_state(pend, mflags)
        for f0 in range(0, F, frames_per_call):
            pcm, status = engine.decode_batch(desc, dev[:, f0:f0 + frames_per_call].contiguous(), delay, lfsr)
            engine.sync()
            assert (status.cpu().numpy() & 0x1ff).max() == 0
            out.append(pcm.cpu().numpy()[0])
    finally:
        engine.set_mix_state(None, None)
    return np.concatenate(out, axis=0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", TAGS)
@pytest.mark.parametrize("frames_per_call", [1, 3, 8])
def test_gpu_equals_liba52_when_the_surround_level_changes(engine, tag, frames_per_call):
    d = np.load(os.path.join(H.GOLDEN, "mixflip.npz"), allow_pickle=False)
    flags = int(d["args_" + tag][0])
    acmod = int(tag[1])
    frames, want = d["frames_" + tag], d["pcm_" + tag]
    bias = float(d["bias_" + tag][0])
    got = _gpu_decode(engine, frames, acmod, flags, frames_per_call, mix_state=True, bias=bias)
    want = want.reshape(got.shape)
    if bias:
        # liba52 forgets the bias in some blocks of these streams (MixPlan::nobias_mask): make sure the fixture shows it
        means = want.mean(axis=3)
        assert (means < 200).any() and (means > 300).any()
    err = got.astype(np.float64) - want
    scale_rms, scale_max = max(1.0, H.rms(want - bias)), max(1.0, float(np.abs(want - bias).max()))
    if bias:
        scale_rms, scale_max = 40.0 * scale_rms, 40.0 * scale_max       # float32 resolution at 384 (as tests/test_dropin_gpu.py)
    per_block = np.abs(err).max(axis=(2, 3))
    assert H.rms(err) <= 1e-6 * scale_rms and np.abs(err).max() <= 1e-5 * scale_max, (tag, np.argwhere(per_block > 1e-5 * scale_max)[:8].tolist())


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["a7_st", "a7_mono", "a6_st"])
def test_plain_linear_mix_differs_only_right_after_a_level_change(engine, tag):
    """Without ac3mi_set_mix_state (the default of the batched API): same samples as liba52 except in the first block after
    a change of the surround level to or from zero - and there it does differ, so the test above checks something."""
    d = np.load(os.path.join(H.GOLDEN, "mixflip.npz"), allow_pickle=False)
    flags = int(d["args_" + tag][0])
    frames, want, levels = d["frames_" + tag], d["pcm_" + tag], d["levels_" + tag]
    got = _gpu_decode(engine, frames, int(tag[1]), flags, len(frames), mix_state=False)
    want = want.reshape(got.shape)
    scale_max = max(1.0, float(np.abs(want).max()))
    per_block = np.abs(got.astype(np.float64) - want).max(axis=(2, 3))         # [F][6]
    zero = levels == 2
    changed = np.zeros(len(levels), bool)
    changed[1:] = zero[1:] != zero[:-1]
    allowed = np.zeros_like(per_block, bool)
    allowed[changed, 0] = True
    assert per_block[~allowed].max() <= 1e-5 * scale_max
    assert per_block[allowed].max() > 1e-3 * scale_max


@pytest.mark.gpu
def test_mix_state_follows_the_streams_through_workspace_tiles(engine):
    """ac3mi_set_tile_frames small enough that a call works through its streams in tiles of whole streams: the mix state
    arrays move along like the overlap tails."""
    engine.set_tile_frames(4)
    try:
        test_gpu_equals_oracle_on_fresh_level_changes_many_streams(engine, acmods=(7,), per_calls=(2, 6))
    finally:
        engine.set_tile_frames(131072)


@pytest.mark.gpu
def test_gpu_equals_oracle_on_fresh_level_changes_many_streams(engine, acmods=(4, 5, 6, 7), per_calls=(1, 2, 6)):
    """Several streams per call with different level sequences, with and without an LFE channel, all three outputs the
    quirk applies to, one / two / all frames per call - against the oracle, which the tests above pin to liba52."""
    import torch
    pkg = H.pkg()
    rng = np.random.default_rng(5)
    for acmod in acmods:
        for lfe in (0, 1):
            S, F = 5, 6
            seqs = [[int(x) for x in rng.choice([0, 1, 2, 2, 3], F)] for _ in range(S)]
            streams = [packer.make_flip_stream(4000 + 10 * acmod + s, seqs[s], acmod=acmod, lfeon=lfe) for s in range(S)]
            fb = streams[0].shape[1]
            stride = (fb + 3) & ~3
            padded = np.zeros((S, F, stride), np.uint8)
            for s in range(S):
                padded[s, :, :fb] = streams[s]
            for flags in ((1, 2, 3) if acmod & 1 else (1, 2)):
                req = flags | (16 if lfe else 0)
                want = np.stack([H.orc_decode(streams[s], req, 1.0, 0.0)[0] for s in range(S)])
                desc = pkg.DecodeDesc(flags=req, level=1.0, bias=0.0, dynrng=1, acmod=acmod, lfeon=lfe, frame_bytes=fb)
                n_out, _ = engine.decode_planes(desc)
                for per_call in per_calls:
                    delay = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
                    lfsr = torch.ones((S,), dtype=torch.int16, device="cuda")
                    pend = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda")
                    mflags = torch.zeros((S, 6), dtype=torch.int32, device="cuda")
                    dev = torch.from_numpy(padded).cuda()
                    got = []
                    try:
                        engine.set_mix_state(pend, mflags)
                        for f0 in range(0, F, per_call):
                            pcm, status = engine.decode_batch(desc, dev[:, f0:f0 + per_call].contiguous(), delay, lfsr)
                            engine.sync()
                            assert (status.cpu().numpy() & 0x1ff).max() == 0
                            got.append(pcm.cpu().numpy())
                    finally:
                        engine.set_mix_state(None, None)
                    got = np.concatenate(got, axis=1)
                    w = want.reshape(got.shape)
                    err = got.astype(np.float64) - w
                    scale_rms, scale_max = max(1.0, H.rms(w)), max(1.0, float(np.abs(w).max()))
                    assert H.rms(err) <= 1e-6 * scale_rms and np.abs(err).max() <= 1e-5 * scale_max, (acmod, lfe, flags, per_call)


@pytest.mark.gpu
def test_drop_in_a52_follows_liba52_through_a_level_change(tmp_path):
    """The a52_* drop-in always carries the mix state: a plain-C host (tests/dropin_c/dropin_host.c) decoding a stream whose
    surround level goes to zero and back gets liba52's samples."""
    import subprocess
    libdir = os.path.join(H.ROOT, "ac-3-acm-codec_amd")
    exe = str(tmp_path / "dropin_host")
    subprocess.check_call(["gcc", "-O1", "-Wall", "-I", os.path.join(H.ROOT, "include"),
                           os.path.join(H.ROOT, "tests", "dropin_c", "dropin_host.c"), "-o", exe,
                           "-L", libdir, "-l:libac3mi.so", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    d = np.load(os.path.join(H.GOLDEN, "mixflip.npz"), allow_pickle=False)
    for tag in ("a7_st", "a6_st"):
        frames, want = d["frames_" + tag], d["pcm_" + tag]
        flags = int(d["args_" + tag][0])
        (tmp_path / "in.ac3").write_bytes(frames.tobytes())
        r = subprocess.run([exe, "dec", str(tmp_path / "in.ac3"), str(tmp_path / "o.f32"), str(tmp_path / "o.s16"),
                            str(flags), "1.0", "0.0", "0"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "frames %d errors 0" % len(frames) in r.stdout
        got = np.fromfile(tmp_path / "o.f32", np.float32).reshape(want.shape)
        err = got.astype(np.float64) - want
        assert H.rms(err) <= 1e-6 * max(1.0, H.rms(want)) and np.abs(err).max() <= 1e-5 * max(1.0, float(np.abs(want).max()))
