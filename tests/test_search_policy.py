"""The SNR-offset search is replayed from exact verdicts, so the ORDER in which offsets are costed may be anything
(DESIGN §4.3 items 2b and 9).  profiles/search_sim.py restates the engine's scalar policy logic on the CPU; here both
policies are run on the oracle's spare-bit curves of a few frames and must end where the reference's own loop
(ENC/ac3enc.cpp:921-967, restated in oracle/ac3enc_oracle.c search_allocation) ends, from any start value."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))


@pytest.fixture(scope="module")
def curves():
    import search_sim
    return search_sim, search_sim.curves(6, seed=5) + search_sim.curves(3, seed=6, second_gen=True)


def test_curve_is_what_the_oracle_searches_on(curves):
    sim, cs = curves
    for c in cs:
        assert c.shape == (1024,)
        # more bits are spent as the offset rises: spare bits fall (up to the grouped codes' ceilings, < 69 bits a frame)
        assert (np.diff(c.astype(np.int64)) <= 69).all()
        csnr, fsnr = sim.reference(c, 40)
        assert 0 <= csnr <= 63 and 0 <= fsnr <= 15
        assert c[16 * csnr + fsnr] >= 0


@pytest.mark.parametrize("policy", ["ladder", "probe"])
def test_any_costing_order_replays_the_reference(curves, policy):
    sim, cs = curves
    for c in cs:
        for start in (40, 0, 63, 7, 11, 12, 25):
            sweeps, got = sim.run(c, start, policy)
            assert got == sim.reference(c, start), (policy, start)
            assert 1 <= sweeps <= 12


def test_probing_needs_fewer_sweeps(curves):
    sim, cs = curves
    ladder = sum(sim.run(c, 40, "ladder")[0] for c in cs)
    probe = sum(sim.run(c, 40, "probe")[0] for c in cs)
    assert probe < ladder


def test_bounds_from_a_costed_offsets_own_ceilings_are_exact(curves):
    """enc_search_kernel (round 4): with E = the group ceilings inside a costed offset's count (sixths of a bit, from the
    oracle: orc_ac3enc_set_extra_curve), spare >= 69 - E/6 proves every lower offset fits and spare < -E/6 proves every
    higher one fails.  Checked against the whole curve of every frame, for every offset that could be costed."""
    sim, cs = curves
    for c in cs:
        fits = np.asarray(c) >= 0
        x = c.extra
        assert x is not None and 0 <= x.min() and x.max() <= 414
        for g in range(1024):
            sp, e6 = int(c[g]), int(x[g])
            if 6 * sp >= 414 - e6:
                assert fits[:g + 1].all(), g
            if 6 * sp < -e6:
                assert not fits[g:].any(), g


def test_tight_bounds_never_need_more_sweeps(curves):
    sim, cs = curves
    for c in cs:
        for start in (40, 12):
            a, ga = sim.run(c, start, "probe", tight=False)
            b, gb = sim.run(c, start, "probe", tight=True)
            assert ga == gb == sim.reference(c, start)
            assert b <= a
