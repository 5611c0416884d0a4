"""GPU parity: PRACH detection (mi_lte_prach_*) vs liblte_phy_detect_prach.  The correlation powers are float
(tolerance-level: FFTW's operation order is unspecified), the three outputs per occasion -- detected or not, preamble
index, timing advance -- are integers and must be identical, the reference's quirks included (a zero-delay preamble is
reported as the next index with timing advance (uint32)-1)."""
import os

import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu


def gpu_detect(ctx, case, roots_fft=None):
    plan = ctx.prach_plan(case["cfg"], case["pc"], roots_fft)
    try:
        n_occ, ln = case["iq"].shape[0], case["iq"].shape[1]
        assert plan.occasion_samples <= ln
        n, p, ta = plan.detect(case["iq"].reshape(-1, 2), np.arange(n_occ) * ln)
        return np.stack([n, p, ta], axis=1), plan.n_roots
    finally:
        plan.close()


@pytest.mark.parametrize("name", list(td.PRACH_CASES))
def test_prach_vs_golden_from_reference(ctx, name):
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "prach_ref.npz"))
    case = td.prach_case(name)
    assert (case["iq"] == z[name + "_iq"]).all(), "the host transmitter no longer reproduces the fixture's capture"
    got, _ = gpu_detect(ctx, case)
    assert (got == z[name + "_det"]).all(), (got.tolist(), z[name + "_det"].tolist())


@pytest.mark.parametrize("name", ["1p4MHz_8roots", "3MHz_restricted", "1p4MHz_format1", "5MHz_format2", "3MHz_format3", "5MHz_format4", "1p4MHz_format4"])
def test_prach_vs_reference_live(ctx, ref, name):
    case = td.prach_case(name, seed=17)
    want, roots = td.ref_prach_detect(ref, case)
    got, n_roots = gpu_detect(ctx, case)
    assert n_roots == roots[0].shape[0]
    assert (got == want).all(), (got.tolist(), want.tolist())
    got2, _ = gpu_detect(ctx, case, roots_fft=roots)  # the shim's form: root spectra taken from the caller's LIBLTE_PHY_STRUCT
    assert (got2 == want).all()


def test_prach_20mhz_batch(ctx):
    """BASELINE config 5 bandwidth: 24 576-sample occasions, a batch of 32, every preamble found where it was put."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg, pc = m.DlCfg(2048, 100, 1, 0), m.PrachCfg(0, 0, 12, 0, 4)
    pre = [(7 * k) % 64 for k in range(4)]
    iq = synth.prach_occasions(cfg, pc, pre, [64 * (k + 1) for k in range(4)], snr_db=0.0, seed=5)
    plan = ctx.prach_plan(cfg, pc)
    idx = np.arange(32) % 4
    n, p, ta = plan.detect(iq[idx].reshape(-1, 2), np.arange(32) * iq.shape[1])
    plan.close()
    assert (n == 1).all() and (p == np.array(pre)[idx]).all()
    assert (ta == ta[:4][idx]).all() and (np.diff(ta[:4].astype(np.int64)) > 0).all()  # longer delays -> larger timing advances
