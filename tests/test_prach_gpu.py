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


def test_prach_root_set_that_wraps_past_the_table_is_self_consistent(ctx):
    """A cell whose 64 preambles need more roots than are left in the logical root table (format 0: root_seq_idx 836 of 838, seven cyclic
    shifts per root -> ten roots: 836, 837, then 0 .. 7 as 36.211 5.7.2 orders them cyclically).  The reference indexes past its table
    there (liblte_phy.cc:7168-7171), so there is nothing of its to compare with (tests/test_fuzz_gpu.py counts and skips such draws);
    the library's transmitter and detector must agree with each other and with the un-wrapped cell that owns the same physical roots:
    preamble 14 + k of the wrapping cell IS preamble k of the cell with root_seq_idx 0, sample for sample, so the detector must report the
    same timing advance for both, and every preamble must be found where it was put."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(512, 25, 1, 0)
    wrap, first = m.PrachCfg(836, 0, 12, 0, 2), m.PrachCfg(0, 0, 12, 0, 2)
    L = ctx.L
    import ctypes as C
    u_w, u_f, n_w, n_f = (C.c_uint32 * 64)(), (C.c_uint32 * 64)(), C.c_uint32(), C.c_uint32()
    assert L.mi_lte_prach_root_set(C.byref(wrap), u_w, C.byref(n_w)) == 0 and L.mi_lte_prach_root_set(C.byref(first), u_f, C.byref(n_f)) == 0
    assert n_w.value == 10 and list(u_w[2:10]) == list(u_f[0:8]), "the wrapped part of the set is the head of the logical order"
    pre_w, dly = [0, 13, 14, 27, 35, 63], [40, 80, 120, 160, 200, 240]
    iq_w = synth.prach_occasions(cfg, wrap, pre_w, dly, snr_db=3.0, seed=11)
    plan = ctx.prach_plan(cfg, wrap)
    assert plan.n_roots == 10
    n, p, ta = plan.detect(iq_w.reshape(-1, 2), np.arange(len(pre_w)) * iq_w.shape[1])
    plan.close()
    assert (n == 1).all() and (p == np.array(pre_w)).all(), (n.tolist(), p.tolist())
    # the same physical sequences sent by the cell that owns them without wrapping: identical samples, identical timing advances
    pre_f = [q - 14 for q in pre_w[2:]]
    iq_f = synth.prach_occasions(cfg, first, pre_f, dly[2:], snr_db=3.0, seed=11)
    plan = ctx.prach_plan(cfg, first)
    n2, p2, ta2 = plan.detect(iq_f.reshape(-1, 2), np.arange(len(pre_f)) * iq_f.shape[1])
    plan.close()
    assert (n2 == 1).all() and (p2 == np.array(pre_f)).all() and (ta2 == ta[2:]).all(), (ta.tolist(), ta2.tolist())


def test_prach_launch_and_fetch_equal_the_one_call_form(ctx):
    """mi_lte_prach_detect_launch / _fetch (the detection in two halves, for a caller that keeps the stream busy): the verdicts of a batch of
    2 000 occasions (past the few-occasions route that writes results straight into pinned memory) and of 3 occasions equal
    mi_lte_prach_detect_run's, with other work queued on the stream between the launch and the fetch; one launch in flight per plan."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg, pc = m.DlCfg(2048, 100, 1, 0), m.PrachCfg(22, 0, 11, 0, 4)
    pre = [(7 * k + 3) % 64 for k in range(8)]
    iq = synth.prach_occasions(cfg, pc, pre, [16 * (k + 1) for k in range(8)], snr_db=5.0, seed=9)
    plan = ctx.prach_plan(cfg, pc)
    try:
        for n_occ in (2000, 3):
            idx = np.arange(n_occ) % 8
            d_a = ctx.to_device(iq[idx].reshape(-1, 2))
            d_s = ctx.to_device((np.arange(n_occ) * iq.shape[1]).astype(np.uint64))
            want = plan.detect_dev(d_a, None, d_s, n_occ)
            plan.launch_dev(d_a, None, d_s, n_occ)
            with pytest.raises(m.MiLteError):
                plan.launch_dev(d_a, None, d_s, n_occ)  # one in flight
            junk = ctx.to_device(np.zeros(1 << 20, np.uint8))  # something else on the stream meanwhile
            got = plan.fetch()
            junk.free()
            for w, g in zip(want, got):
                assert (w == g).all()
            assert (got[0] == 1).all() and (got[1] == np.array(pre)[idx]).all()
            with pytest.raises(m.MiLteError):
                ctx._check(ctx.L.mi_lte_prach_detect_fetch(ctx.h, plan.h, got[0], got[1], got[2], n_occ))  # nothing in flight
            d_a.free()
            d_s.free()
    finally:
        plan.close()
