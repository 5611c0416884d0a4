"""GPU parity: OFDM demod + CRS channel estimate (HIP) vs the oracle.  Floating point: the
tolerances are SURVEY 8d's -- rel-L2 <= 1e-5 on rx_symb (FFT), <= 1e-4 on rx_ce (libm-dependent)."""
import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu

TOL_SYMB = 1e-5
TOL_CE = 1e-4


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


@pytest.mark.parametrize("fft,n_rb", [(2048, 100), (1024, 50), (512, 25), (256, 15), (128, 6)])
def test_frontend_vs_oracle(ctx, port, fft, n_rb):
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(fft, n_rb, 1, 0)
    sfs, cells = [1, 7, 0, 5], [17, 301, 503, 0]
    tbs = {100: 2024, 50: 1384, 25: 680, 15: 680, 6: 256}[n_rb]
    nprb = {100: 8, 50: 6, 25: 4, 15: 4, 6: 3}[n_rb]
    allocs = []
    for u in range(4):
        allocs += td.small_allocs(u, n_rb, 3, tbs, nprb)
    iq, _ = synth.dl_units(cfg, sfs, cells, allocs, 1, snr_db=30, max_delay=5, seed=fft)
    ul = iq.shape[1]
    got = ctx.dl_frontend(cfg, iq.reshape(-1, 2), np.arange(4) * ul, sfs, cells)
    n_sc = 12 * n_rb
    for u in range(4):
        _, s = td.oracle_frontend(port, fft, n_rb, 1, iq[u], sfs[u], cells[u])
        # rows 0-14: with one or two ports the second look-ahead symbol (row 15, read by ports 2 and 3 only) is not produced
        for name, plane in (("rx_symb_re", 0), ("rx_symb_im", 1)):
            assert rel_l2(got[u, plane, :15, :n_sc], s.arr(name)[:15, :n_sc]) < TOL_SYMB, (u, name)
        assert rel_l2(got[u, 2, :14, :n_sc], s.arr("rx_ce_re")[0, :14, :n_sc]) < TOL_CE
        assert rel_l2(got[u, 3, :14, :n_sc], s.arr("rx_ce_im")[0, :14, :n_sc]) < TOL_CE
        # element-wise on the estimates too (relative to the estimate's own magnitude)
        h = np.hypot(s.arr("rx_ce_re")[0, :14, :n_sc], s.arr("rx_ce_im")[0, :14, :n_sc])
        err = np.hypot(got[u, 2, :14, :n_sc] - s.arr("rx_ce_re")[0, :14, :n_sc], got[u, 3, :14, :n_sc] - s.arr("rx_ce_im")[0, :14, :n_sc])
        assert (err / h).max() < 10 * TOL_CE


@pytest.mark.parametrize("n_ant", [2, 4])
def test_frontend_multi_port_estimates(ctx, port, n_ant):
    """Ports 1-3 reuse the same kernels with other pilot positions; the single-port capture simply
    yields noise-like estimates there, which must still match the oracle's arithmetic."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg1 = m.DlCfg(2048, 100, 1, 0)
    iq, _ = synth.dl_units(cfg1, [3], [42], td.small_allocs(0, 100, 3, 2024, 8), 1, snr_db=25, seed=5)
    cfg = m.DlCfg(2048, 100, n_ant, 0)
    got = ctx.dl_frontend(cfg, iq.reshape(-1, 2), [0], [3], [42])
    _, s = td.oracle_frontend(port, 2048, 100, n_ant, iq[0], 3, 42)
    for p in range(n_ant):
        a_re, a_im = got[0, 2 + p, :14], got[0, 2 + n_ant + p, :14]
        b_re, b_im = s.arr("rx_ce_re")[p, :14], s.arr("rx_ce_im")[p, :14]
        # estimates on pilot-free ports are dominated by data REs -> phase unwrap decisions near +-pi can
        # differ by libm ULPs; compare on the bulk
        err = np.hypot(a_re - b_re, a_im - b_im) / np.maximum(np.hypot(b_re, b_im), 1e-6)
        assert np.quantile(err, 0.99) < 1e-3, p
    assert rel_l2(got[0, 2, :14], s.arr("rx_ce_re")[0, :14]) < TOL_CE


def test_frontend_float_planar_input(ctx, port):
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, 0)
    iq, _ = synth.dl_units(cfg, [2], [9], td.small_allocs(0, 100, 1, 680, 8), 1, seed=3)
    a = ctx.dl_frontend(cfg, iq.reshape(-1, 2), [0], [2], [9])
    cfg2 = m.DlCfg(2048, 100, 1, 1)
    b = ctx.dl_frontend(cfg2, (iq[0, :, 0].astype(np.float32), iq[0, :, 1].astype(np.float32)), [0], [2], [9])
    assert (a == b).all()


def test_frontend_many_phase_wraps(ctx, port):
    """Large timing offsets -> steep phase ramps over frequency (dozens of 2*pi wraps across the band): exercises
    the wave-parallel unwrap of k_dl_ce (wrap counts from a prefix sum, verified link by link against the
    reference's own wrap_phase rule) far from the near-flat channels of the other cases."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, 0)
    sfs, cells = [2, 4, 8, 9, 1, 3], [5, 100, 250, 333, 444, 501]
    allocs = []
    for u in range(6):
        allocs += td.small_allocs(u, 100, 3, 2024, 8)
    iq, _ = synth.dl_units(cfg, sfs, cells, allocs, 1, snr_db=30, max_delay=90, seed=99)
    ul = iq.shape[1]
    got = ctx.dl_frontend(cfg, iq.reshape(-1, 2), np.arange(6) * ul, sfs, cells)
    for u in range(6):
        _, s = td.oracle_frontend(port, 2048, 100, 1, iq[u], sfs[u], cells[u])
        assert rel_l2(got[u, 2, :14, :1200], s.arr("rx_ce_re")[0, :14, :1200]) < TOL_CE, u
        assert rel_l2(got[u, 3, :14, :1200], s.arr("rx_ce_im")[0, :14, :1200]) < TOL_CE, u


@pytest.mark.parametrize("n_ant", [2, 4])
def test_frontend_real_multi_port_cell(ctx, ref, n_ant):
    """Channel estimates of every port of a real 2- / 4-port cell (CRS on all ports, per-antenna gains; the reference's transmitter)
    against the compiled reference's estimator: ports 2, 3 have two CRS symbols per slot instead of four and their own interpolation."""
    import ctypes as C
    import openlte_amd as m
    from oracle import pyoracle as po
    cap = td.multi_port_capture(ref, n_ant)
    n_samp, sf, cell = 30720, cap["sf"], cap["cell"]
    iq = cap["iq"]
    i_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * n_samp, np.float32), iq[:, 0].astype(np.float32)]))
    q_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * n_samp, np.float32), iq[:, 1].astype(np.float32)]))
    rx = ref.ref_subframe_new()
    assert ref.ref_get_dl_subframe_and_ce(cap["phy"], i_f, q_f, 0, sf, cell, n_ant, rx) == 0
    got = ctx.dl_frontend(m.DlCfg(2048, 100, n_ant, 0), iq, [0], [sf], [cell])[0]
    assert rel_l2(got[0, :14], po.ref_subframe_view(ref, rx, 0)[:14]) < TOL_SYMB
    assert rel_l2(got[1, :14], po.ref_subframe_view(ref, rx, 1)[:14]) < TOL_SYMB
    for p in range(n_ant):
        w_re, w_im = po.ref_subframe_view(ref, rx, 2, True)[p, :14], po.ref_subframe_view(ref, rx, 3, True)[p, :14]
        assert rel_l2(got[2 + p, :14], w_re) < TOL_CE and rel_l2(got[2 + n_ant + p, :14], w_im) < TOL_CE, p
        h = np.hypot(w_re, w_im)
        assert (np.hypot(got[2 + p, :14] - w_re, got[2 + n_ant + p, :14] - w_im) / h).max() < 10 * TOL_CE, p
    ref.ref_subframe_free(rx)
    ref.ref_phy_free(cap["phy"])


def test_two_port_cell_from_the_reference_transmitter_vs_golden(ctx):
    """A real two-port cell (tests/golden/dl_two_port_units.npz: the reference's transmitter, CRS on both ports, per-antenna gains): symbol
    rows and both ports' estimate rows against what the compiled reference made of unit 0 when the fixture was generated."""
    import os
    import openlte_amd as m
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dl_two_port_units.npz"))
    cfg = m.DlCfg(2048, 100, 2, 0)
    n = len(z["sfs"])
    ul = z["iq"].shape[1]
    got = ctx.dl_frontend(cfg, z["iq"].reshape(-1, 2), np.arange(n) * ul, z["sfs"], z["cells"])
    assert rel_l2(got[0, 0, :14], z["symb_re"]) < TOL_SYMB and rel_l2(got[0, 1, :14], z["symb_im"]) < TOL_SYMB
    for p in range(2):
        assert rel_l2(got[0, 2 + p, :14], z["ce_re"][p]) < TOL_CE and rel_l2(got[0, 4 + p, :14], z["ce_im"][p]) < TOL_CE
