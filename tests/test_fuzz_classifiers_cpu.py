"""The differential tests (tests/test_fuzz_gpu.py) accept three kinds of difference between the library and the reference behind the
tolerance stages, each on evidence computed from the REFERENCE's own data.  These tests pin the classifiers themselves: what they accept is
accepted for the stated reason, and look-alikes without the evidence are refused."""
import numpy as np

import test_fuzz_gpu as T


def _planes(h, n_ant):
    """complex estimate [n_ant, 14, n_sc] -> the (2 + 2 n_ant) x 16 x 1200 float planes the tests hold"""
    p = np.zeros((2 + 2 * n_ant, 16, 1200), np.float32)
    n_sc = h.shape[2]
    p[2:2 + n_ant, :14, :n_sc] = h.real
    p[2 + n_ant:2 + 2 * n_ant, :14, :n_sc] = h.imag
    return p


def _estimate(rng, n_ant, n_sc):
    return (rng.normal(size=(n_ant, 14, n_sc)) + 1j * rng.normal(size=(n_ant, 14, n_sc))) * 0.7 + 1.0


def test_phase_tie_is_accepted_only_next_to_a_step_of_pi():
    rng = np.random.default_rng(1)
    n_ant, n_sc, k = 2, 300, 149
    h = _estimate(rng, n_ant, n_sc)
    # the reference's estimate turns by pi - 5e-6 between the CRS symbols 4 and 7 of port 0 on sub-carrier k ...
    h[0, 7, k] = h[0, 4, k] * np.exp(1j * (np.pi - 5e-6))
    want = _planes(h, n_ant)
    other = h.copy()
    other[0, 5, k] = np.conj(h[0, 5, k])  # ... and the other front end interpolated symbols 5 and 6 the other way round
    other[0, 6, k] = np.conj(h[0, 6, k])
    tie = T.ce_phase_tie(_planes(other, n_ant), want, n_ant, n_sc)
    assert tie["accepted"] and tie["port"] == [0] and tie["subcarriers"] == [k] and tie["symbols"] == [5, 6] and tie["gap"] < 1e-5
    # the same difference WITHOUT such a step in the reference's estimate is a failure
    h2 = _estimate(rng, n_ant, n_sc)
    h2[0, 7, k] = h2[0, 4, k] * np.exp(1j * 1.0)
    h2[0, 7, k - 6:k + 7] = h2[0, 4, k - 6:k + 7] * np.exp(1j * 1.0)
    for s0, s1 in ((0, 4), (7, 11)):
        h2[0, s1, k - 12:k + 13] = h2[0, s0, k - 12:k + 13] * np.exp(1j * 0.5)
    for s in (0, 4, 7, 11):
        h2[0, s, k - 12:k + 13] = np.abs(h2[0, s, k - 12:k + 13]) * np.exp(1j * (0.1 * np.arange(25) + {0: 0.0, 4: 0.5, 7: 1.5, 11: 2.0}[s]))
    o2 = h2.copy()
    o2[0, 5, k] = np.conj(h2[0, 5, k]) * 1.3
    tie2 = T.ce_phase_tie(_planes(o2, n_ant), _planes(h2, n_ant), n_ant, n_sc)
    assert tie2 is not None and not tie2["accepted"] and tie2["gap"] > 1e-3
    # a tie's evidence does not cover differences on two ports, or spread over more than six sub-carriers
    o3 = other.copy()
    o3[1, 5, k] *= 1.5
    assert not T.ce_phase_tie(_planes(o3, n_ant), want, n_ant, n_sc)["accepted"]
    o4 = other.copy()
    o4[0, 5, k + 20] *= 1.5
    assert not T.ce_phase_tie(_planes(o4, n_ant), want, n_ant, n_sc)["accepted"]
    # and no difference is no finding
    assert T.ce_phase_tie(want, want, n_ant, n_sc) is None


def test_soft_bit_classifiers():
    # [allocation, position, library, reference, library's other bit of the symbol, reference's, steps the symbol's weakest estimate allows]
    flip = ["a", 10, 5, -5, 9, 9, 0.0]
    step = ["a", 10, 21, 22, -21, -22, 0.0]
    far = ["a", 10, 21, 24, -21, -24, 0.4]
    far_ok = ["a", 10, 21, 24, -21, -24, 13.4]
    too_far = ["a", 10, 21, 60, -21, -60, 13.4]
    assert T.sign_flip(flip) and not T.magnitude_step(flip) and not T.outlier(flip)
    assert T.magnitude_step(step) and not T.sign_flip(step) and not T.outlier(step)
    assert not (T.sign_flip(far) or T.magnitude_step(far) or T.outlier(far))          # three steps with nothing to account for them: a failure
    assert T.outlier(far_ok) and not T.sign_flip(far_ok) and not T.magnitude_step(far_ok)
    assert not T.outlier(too_far)                                                     # more than the estimate accounts for
    assert isinstance(T.outlier(far_ok), bool)
    # an estimate that passes through zero makes the formula's room unbounded: no such evidence excuses more than OUTLIER_ROOM_CAP steps
    assert T.outlier(["a", 10, 21, 50, -21, -50, 1e9]) and not T.outlier(["a", 10, 21, 60, -21, -60, 1e9]) and T.OUTLIER_ROOM_CAP == 32.0
    # the accepted kinds are a closed set (a new one needs a test here that fails without it, and an entry in DESIGN.md section 4)
    assert T.ACCEPTED_KINDS == ("sign_flip", "magnitude_step", "outlier", "ce_phase_tie", "pss_near_tie")


def test_steps_behind_an_extrapolated_estimate_grows_as_the_estimate_vanishes():
    """127 sqrt(M) 2e-7 / w^2 from the reference's received symbols: an allocation whose DMRS magnitudes extrapolate to (almost) zero on one
    element of symbol s accounts for many steps there and for none in the symbols whose estimate stays healthy"""
    import openlte_amd as m
    n_prb, cell, sf, ulc = 6, 17, 4, (3, 0, 0, 2, 5)
    prbs = list(range(10, 16))
    d = m.ul_dmrs_pusch(m.UlCfg(*ulc), cell, sf, n_prb).astype(np.float64)
    M = 12 * n_prb
    sc = 12 * prbs[0] + np.arange(M)
    z = np.zeros((1, 2, 14, 1200), np.float32)
    # received DMRS symbols = dmrs * gain: gain 1 in slot 0; 1 in slot 1 as well, except sub-carrier 7 where it is 10/3 + 1e-3:
    # mag_0 - 3 (mag_1 - mag_0) / 7 = 1 - 3 (7/3 + 1e-3) / 7 = -4.3e-4 of ~1 at symbol 0 (n = -3)
    g1 = np.ones(M)
    g1[7] = 10.0 / 3.0 + 1e-3
    for b, L, g in ((0, 3, np.ones(M)), (1, 10, g1)):
        t = (d[2 * b] + 1j * d[2 * b + 1]) * g
        z[0, 0, L, sc], z[0, 1, L, sc] = t.real, t.imag
    grp = dict(ref_symb=z, ulcfg=m.UlCfg(*ulc), cell=cell, sfs=[sf])
    rooms = [T.steps_behind_an_extrapolated_estimate(grp, 0, prbs, s) for s in range(12)]
    assert rooms[0] > 100 and rooms[0] > 20 * rooms[1] and max(rooms[3:]) < 2 and all(isinstance(r, float) for r in rooms)
