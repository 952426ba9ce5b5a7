"""GPU parity: ac3mi_imdct_batch (HIP) vs the liba52 restatement in oracle/.

Reference behaviour under test: the synthesis stage of a52_block
(liba52/parse.c:867-937 -> a52_imdct_512/256, a52_downmix, a52_upmix).
Tolerance: float32 transform; north_star bar = 1e-6 RMS relative to +-1.0 full
scale.  The oracle itself is bit-identical to the real liba52 (test_oracle_vs_ref).
"""
import numpy as np
import pytest

from tests import _harness as H

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-6          # north_star: "PCM within 1e-6 RMS of liba52"
MAX_TOL = 4e-6          # and no single sample further off than a few float32 ulps of full scale


def _coefs(rng, S, F, n_in, scale=0.05):
    # dequantised AC-3 coefficients are |x| < 1 and decay with frequency
    env = np.exp(-np.arange(256) / 90.0).astype(np.float32)
    c = rng.standard_normal((S, F, 6, n_in, 256)).astype(np.float32) * scale * env
    c[..., 253:] = 0          # bins >= 253 are never coded (chbwcod <= 60)
    return c


def _run_gpu(engine, desc_args, coef, blksw=None, delay=None):
    import torch
    pkg = H.pkg()
    desc = pkg.XformDesc(*desc_args)
    n_in, n_out = engine.planes(desc)
    S = coef.shape[0]
    d = torch.zeros((S, n_out, 128), dtype=torch.float32, device="cuda") if delay is None else delay
    sw = None if blksw is None else torch.from_numpy(blksw).cuda()
    torch.cuda.synchronize()
    out = engine.imdct_batch(desc, torch.from_numpy(coef).cuda(), d, sw)
    engine.sync()
    return out.cpu().numpy(), d


def _check(gpu, ref, what):
    err = gpu.astype(np.float64) - ref.astype(np.float64)
    r, m = H.rms(err), float(np.abs(err).max())
    assert r <= RMS_TOL and m <= MAX_TOL, "%s: rms %.3e max %.3e" % (what, r, m)


def test_config2_long_blocks_5_1(engine):
    """BASELINE config 2 shape (5.1 -> 5.1, all long blocks), several frames per stream so
    the register-resident overlap tail is exercised across blocks and frames."""
    rng = np.random.default_rng(2)
    S, F = 37, 3                                   # 37*6 chains: not a multiple of 32 (ragged last workgroup)
    coef = _coefs(rng, S, F, 6)
    ref, (planes, _) = H.orc_xform(coef, None, 7, 1, 7 | 16, bias=0.0)
    gpu, delay = _run_gpu(engine, (7, 1, 7 | 16, 0.0), coef)
    _check(gpu, ref, "pcm")
    # final overlap state: liba52 delay plane o, live half
    d = delay.cpu().numpy()
    want = planes.reshape(S, 12, 256)[:, 6:12, :128]
    _check(d, want, "delay")


def test_bias_and_streaming_state(engine):
    """bias 384 (the ACM driver's setting, src/AC3ACM.cpp:1555-1561) and a second call that
    continues from the first call's tails."""
    import torch
    rng = np.random.default_rng(3)
    S = 8
    c1, c2 = _coefs(rng, S, 1, 6), _coefs(rng, S, 2, 6)
    r1, st = H.orc_xform(c1, None, 7, 1, 7 | 16, bias=384.0)
    r2, st = H.orc_xform(c2, None, 7, 1, 7 | 16, bias=384.0, state=st)
    delay = torch.zeros((S, 6, 128), dtype=torch.float32, device="cuda")
    # a third call: the second one ran as whole-frame segments (few chains, two frames), and the tails it left behind
    # were written by segment 0's extra pass over the chain's last block
    c3 = _coefs(rng, S, 3, 6)
    r3, st = H.orc_xform(c3, None, 7, 1, 7 | 16, bias=384.0, state=st)
    g1, _ = _run_gpu(engine, (7, 1, 7 | 16, 384.0), c1, delay=delay)
    g2, _ = _run_gpu(engine, (7, 1, 7 | 16, 384.0), c2, delay=delay)
    g3, _ = _run_gpu(engine, (7, 1, 7 | 16, 384.0), c3, delay=delay)
    # with bias 384 one float32 ulp is 3e-5: compare after removing the bias exactly as the
    # s16 converter does (integer part), i.e. in units of 1/32768 full scale
    for g, r in ((g1, r1), (g2, r2), (g3, r3)):
        err = (g.astype(np.float64) - r.astype(np.float64))
        assert np.abs(err).max() <= 2 * 3.0518e-5, np.abs(err).max()


def test_short_blocks_identity(engine):
    """IMDCT-256 (blksw = 1) and arbitrary long/short sequences per channel, no downmix."""
    rng = np.random.default_rng(4)
    S, F = 16, 2
    coef = _coefs(rng, S, F, 5)
    blksw = (rng.random((S, F, 6, 5)) < 0.5).astype(np.uint8)
    blksw[0] = 1
    blksw[1] = 0
    ref, _ = H.orc_xform(coef, blksw, 7, 0, 7, bias=0.0)
    gpu, _ = _run_gpu(engine, (7, 0, 7, 0.0), coef, blksw)
    _check(gpu, ref, "mixed long/short")


@pytest.mark.parametrize("output", [2, 10, 1, 3, 4, 5, 6])
def test_config4_downmix_mixed_blocks(engine, output):
    """BASELINE config 4: 5.1 coded, fewer output channels, per-channel block switching.
    Exercises liba52's path A (time-domain mix, parse.c:887-916), path B (frequency-domain
    mix, :917-937) and the `downmixed` overlap re-mix when the path changes between blocks."""
    rng = np.random.default_rng(40 + output)
    S, F = 12, 3
    coef = _coefs(rng, S, F, 6)
    blksw = (rng.random((S, F, 6, 5)) < 0.3).astype(np.uint8)
    blksw[0] = 0                                    # stream 0: path B only
    blksw[1] = 1                                    # stream 1: all short (still path B)
    blksw[2, :, ::2] = 0                            # stream 2: alternate uniform / mixed blocks
    coef2 = _coefs(rng, S, 2, 6)
    blksw2 = (rng.random((S, 2, 6, 5)) < 0.3).astype(np.uint8)
    for lfe_out in (0, 16):
        ref, st = H.orc_xform(coef, blksw, 7, 1, output | lfe_out, bias=0.0, clev=0.5946, slev=0.5)
        gpu, delay = _run_gpu(engine, (7, 1, output | lfe_out, 0.0), coef, blksw)
        _check(gpu, ref, "5.1 -> %d" % (output | lfe_out))
        # a second call continues from the tails the first one (run as whole-frame segments) left behind
        ref2, _ = H.orc_xform(coef2, blksw2, 7, 1, output | lfe_out, bias=0.0, clev=0.5946, slev=0.5, state=st)
        gpu2, _ = _run_gpu(engine, (7, 1, output | lfe_out, 0.0), coef2, blksw2, delay=delay)
        _check(gpu2, ref2, "5.1 -> %d, second call" % (output | lfe_out))


@pytest.mark.parametrize("acmod,output", [(0, 0), (0, 1), (0, 8), (0, 9), (1, 1), (1, 10), (2, 1), (2, 2), (3, 2),
                                          (3, 10), (4, 2), (4, 10), (4, 6), (5, 2), (5, 10), (5, 3), (5, 4),
                                          (5, 6), (5, 7), (6, 2), (6, 10), (6, 4), (6, 1)])
def test_other_channel_modes(engine, acmod, output):
    """Every (acmod, output) pair a52_downmix_init can grant (liba52/downmix.c:37-60), long blocks
    plus a few short ones."""
    rng = np.random.default_rng(100 + acmod * 16 + output)
    nf = H.NFCHANS[acmod]
    S, F = 5, 2
    coef = _coefs(rng, S, F, nf)
    blksw = (rng.random((S, F, 6, nf)) < 0.25).astype(np.uint8)
    ref, _ = H.orc_xform(coef, blksw, acmod, 0, output, bias=0.0, clev=0.7071, slev=0.7071)
    gpu, _ = _run_gpu(engine, (acmod, 0, output, 0.0), coef, blksw)
    _check(gpu, ref, "acmod %d -> %d" % (acmod, output))


def test_rejects_unreachable_output(engine):
    pkg = H.pkg()
    with pytest.raises(pkg.AC3MIError):
        engine.planes(pkg.XformDesc(2, 0, 7, 0.0))       # stereo can never become 3F2R
    with pytest.raises(pkg.AC3MIError):
        engine.planes(pkg.XformDesc(7, 0, 7 | 16, 0.0))  # LFE output without a coded LFE


def test_full_size_properties(engine):
    """BASELINE config 2 at full size (65536 frames): size-independent properties.
    (a) linearity: T(a x + b y) = a T(x) + b T(y) with zero state;
    (b) TDAC: coefficients of a forward MDCT of a smooth signal reconstruct it."""
    import torch
    pkg = H.pkg()
    S = 65536
    desc = pkg.XformDesc(7, 1, 7 | 16, 0.0)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn((S, 1, 6, 6, 256), device="cuda", generator=g) * 0.05
    y = torch.randn((S, 1, 6, 6, 256), device="cuda", generator=g) * 0.05
    z = lambda: torch.zeros((S, 6, 128), device="cuda")
    tx = engine.imdct_batch(desc, x, z())
    ty = engine.imdct_batch(desc, y, z())
    txy = engine.imdct_batch(desc, (0.5 * x - 2.0 * y).contiguous(), z())
    engine.sync()
    err = (txy - (0.5 * tx - 2.0 * ty)).abs().max().item()
    assert err < 2e-5, err
    # spot-check a slice of the big batch against the oracle
    sl = slice(31000, 31016)
    ref, _ = H.orc_xform(x[sl].cpu().numpy(), None, 7, 1, 7 | 16)
    _check(tx[sl].cpu().numpy(), ref, "slice of full batch")
