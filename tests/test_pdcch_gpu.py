"""GPU parity: PCFICH + PDCCH common-search-space decoding (HIP) vs the compiled reference's
liblte_phy_pdcch_channel_decode.

Stage parity is exact: both sides read the same received grids and channel estimates, the de-mapper's soft values are
integers, and everything after it (rate un-matching, the K = 7 Viterbi decoder, CRC16, RNTI test, DCI unpacking) is
integer work -- so the CFI, the number of PDCCH symbols, the allocation count and every allocation field must be equal,
including at low SNR where what is decoded is partly noise."""
import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu


def run_case(ctx, case):
    import openlte_amd as m
    cfg = m.DlCfg(case["fft"], case["nrb"], case["n_ant"], 0)
    plan = ctx.pdcch_plan(cfg, [case["cell"]], case["phich_res"])
    n = len(case["sfs"])
    d_g = ctx.to_device(np.ascontiguousarray(case["grids"], np.float32))
    d_sf, d_cell = ctx.to_device(np.asarray(case["sfs"], np.uint32)), ctx.to_device(np.full(n, case["cell"], np.uint32))
    try:
        return plan.decode_dev(d_g, d_sf, d_cell, n)
    finally:
        for d in (d_g, d_sf, d_cell):
            d.free()
        plan.close()


@pytest.mark.parametrize("name", list(td.PDCCH_CASES))
def test_pdcch_matches_reference(ctx, ref, name):
    case = td.pdcch_case(ref, name)
    want = td.ref_pdcch_decode(ref, case)
    rc, cfi, nsym, dcis = run_case(ctx, case)
    n_found = 0
    for u, (w_rc, w_cfi, w_nsym, recs) in enumerate(want):
        assert int(rc[u]) == w_rc, (name, u)
        assert int(cfi[u]) == w_cfi, (name, u)
        assert int(nsym[u]) == w_nsym, (name, u)
        assert td.dci_records(dcis[u]) == recs, (name, u)
        n_found += len(recs)
    if name in ("20MHz_1ant", "5MHz_2ant", "3MHz_sixth"):  # at a workable SNR everything the transmitter sent is found
        for u, (sf, tx_cfi, tx) in enumerate(case["units"]):
            assert int(cfi[u]) == tx_cfi
            if tx_cfi + (1 if case["nrb"] <= 10 else 0) < 2:
                continue  # too few CCEs for a whole aggregation-4 candidate
            found = {(d.alloc.rnti, d.mcs, d.alloc.N_prb, d.alloc.prb[0][0], d.alloc.rv_idx) for d in dcis[u]}
            assert found == {tuple(t) for t in tx}, (name, u)
    if name == "15MHz_noisy":
        assert 0 < n_found


def test_pdcch_golden_fixture(ctx):
    """The same comparison against outputs of the reference recorded by tools/gen_golden.py (no oracle at run time)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "pdcch_ref.npz"), allow_pickle=False)
    for name in [str(x) for x in g["names"]]:
        fft, nrb, n_ant, cell = [int(x) for x in g[name + "_cfg"]]
        case = dict(fft=fft, nrb=nrb, n_ant=n_ant, cell=cell, phich_res=float(g[name + "_phich_res"]), sfs=g[name + "_sfs"].tolist(),
                    grids=np.zeros((len(g[name + "_sfs"]), 2 + 2 * n_ant, 16, 1200), np.float32))
        case["grids"][:, :, :4, :] = g[name + "_grids"]
        rc, cfi, nsym, dcis = run_case(ctx, case)
        assert rc.tolist() == g[name + "_rc"].tolist()
        assert cfi.tolist() == g[name + "_cfi"].tolist()
        assert nsym.tolist() == g[name + "_nsym"].tolist()
        got = [[u] + list(r[:7]) + [r[7][0] if r[7] else 0, r[8][-1] if r[8] else 0] for u in range(len(dcis)) for r in td.dci_records(dcis[u])]
        assert got == g[name + "_dci"].tolist(), name


def test_pdcch_unknown_cell_and_two_cells(ctx, ref):
    """One plan serves several cells; a cell it was not built for reports cfi = 0 instead of decoding with wrong tables."""
    import openlte_amd as m
    a, b = td.pdcch_case(ref, "5MHz_2ant"), dict(td.pdcch_case(ref, "5MHz_2ant", seed=12))
    cfg = m.DlCfg(a["fft"], a["nrb"], a["n_ant"], 0)
    plan = ctx.pdcch_plan(cfg, [3, a["cell"]], a["phich_res"])
    n = len(a["sfs"])
    d_g, d_sf = ctx.to_device(np.ascontiguousarray(a["grids"], np.float32)), ctx.to_device(np.asarray(a["sfs"], np.uint32))
    d_c1, d_c2 = ctx.to_device(np.full(n, a["cell"], np.uint32)), ctx.to_device(np.full(n, 5, np.uint32))
    _, cfi, _, dcis = plan.decode_dev(d_g, d_sf, d_c1, n)
    want = td.ref_pdcch_decode(ref, a)
    assert [td.dci_records(x) for x in dcis] == [w[3] for w in want]
    rc2, cfi2, nsym2, dcis2 = plan.decode_dev(d_g, d_sf, d_c2, n)
    assert rc2.tolist() == [3] * n and cfi2.tolist() == [0] * n and all(len(x) == 0 for x in dcis2)
    for d in (d_g, d_sf, d_c1, d_c2):
        d.free()
    plan.close()


@pytest.mark.parametrize("n_ant,fft,nrb,cell", [(2, 512, 25, 44), (4, 1024, 50, 100), (2, 2048, 100, 17), (4, 2048, 100, 301)])
def test_pdcch_per_port_estimates_decode_standard_transmit_diversity(ctx, ref, n_ant, fft, nrb, cell):
    """MI_LTE_PDCCH_PER_PORT_ESTIMATES: every port's own channel estimate goes into the transmit-diversity combiner (the
    reference mis-strides that array on both its transmit and its receive side, include/mi_lte.h).  Checked end to end on
    control regions transmitted on all ports as 36.211 6.3.4.3 says: everything sent is found, 4 ports included -- and the
    reference-parity mode, fed the same grids, does worse."""
    import openlte_amd as m
    units = [(3, 2, [(0xFFFF, 6, 4, 3, 0), (0x0001, 2, 2, 20, 3)]), (8, 3, [(0xFFFF, 11, 6, 0, 1), (0xFFFE, 4, 3, 9, 0), (0x0011, 1, 2, 2, 0)])]
    case = td.pdcch_per_port_case(ref, fft, nrb, n_ant, cell, 1.0, units, snr_db=8.0)
    cfg = m.DlCfg(fft, nrb, n_ant, 0)
    n = len(units)
    d_g = ctx.to_device(np.ascontiguousarray(case["grids"], np.float32))
    d_sf, d_cell = ctx.to_device(np.asarray(case["sfs"], np.uint32)), ctx.to_device(np.full(n, cell, np.uint32))
    res = {}
    for per_port in (True, False):
        plan = ctx.pdcch_plan(cfg, [cell], 1.0, per_port_estimates=per_port)
        res[per_port] = plan.decode_dev(d_g, d_sf, d_cell, n)
        plan.close()
    for d in (d_g, d_sf, d_cell):
        d.free()
    rc, cfi, nsym, dcis = res[True]
    n_ref_mode = 0
    for u, (sf, tx_cfi, tx) in enumerate(units):
        assert int(cfi[u]) == tx_cfi and int(rc[u]) == 0
        found = {(d.alloc.rnti, d.mcs, d.alloc.N_prb, d.alloc.prb[0][0], d.alloc.rv_idx) for d in dcis[u]}
        assert found == {tuple(t) for t in tx}, u
        n_ref_mode += len({(d.alloc.rnti, d.mcs) for d in res[False][3][u]} & {(t[0], t[1]) for t in tx})
    if n_ant == 4:
        assert n_ref_mode == 0


@pytest.mark.parametrize("n_ant,fft,nrb", [(1, 2048, 100), (2, 2048, 100), (4, 2048, 100), (1, 128, 6), (2, 512, 25), (1, 1024, 50), (2, 2048, 75), (1, 256, 15)])
def test_pdcch_batch_finds_every_dci_sent(ctx, ref, n_ant, fft, nrb):
    """Size-independent property on a batch of synthetic control regions (the library's own standard-conformant transmitter,
    random cells / subframes / CFIs / DCIs): every DCI sent is found with the fields it was sent with, nothing else is, and --
    with one port, where the reference's arithmetic is the standard's -- the compiled reference agrees on a sample."""
    import openlte_amd as m
    from openlte_amd import synth
    rng = np.random.default_rng(fft + n_ant)
    n, cells = 96, [int(c) for c in rng.choice(504, 5, replace=False)]
    cfg = m.DlCfg(fft, nrb, n_ant, 0)
    sfs, cell = rng.integers(0, 10, n), rng.choice(cells, n)
    cfis = rng.integers(2 if nrb > 10 else 1, 4, n)
    rntis = [0xFFFF, 0xFFFE] + list(range(1, 0x3D))
    dcis = []
    for u in range(n):
        k = int(rng.integers(0, 4 if nrb > 6 else 2))
        lst = []
        for r in rng.choice(rntis, k, replace=False):
            npb = int(rng.integers(1, min(nrb // 2, 8) + 1))
            lst.append((int(r), int(rng.integers(0, 27)), npb, int(rng.integers(0, nrb - npb + 1)), int(rng.integers(0, 4))))
        dcis.append(lst)
    g = synth.ctrl_grids(cfg, sfs, cell, cfis, dcis, snr_db=12.0, seed=nrb + n_ant)
    plan = ctx.pdcch_plan(cfg, cells, 1.0, per_port_estimates=True)
    d_g, d_sf, d_cell = ctx.to_device(g), ctx.to_device(sfs.astype(np.uint32)), ctx.to_device(cell.astype(np.uint32))
    rc, cfi, nsym, got = plan.decode_dev(d_g, d_sf, d_cell, n)
    for d in (d_g, d_sf, d_cell):
        d.free()
    plan.close()
    assert cfi.tolist() == cfis.tolist()
    n_extra = 0
    for u in range(n):
        _, cand = m.pdcch_re_tables(nrb, n_ant, int(cell[u]), 1.0, int(nsym[u]))
        sent = {t for a, t in enumerate(dcis[u]) if cand[a, 143] != 0xFFFFFFFF}  # a candidate without all its CCEs is not transmitted
        found = {(d.alloc.rnti, d.mcs, d.alloc.N_prb, d.alloc.prb[0][0], d.alloc.rv_idx) for d in got[u] if d.format == 0}
        assert sent <= found, (u, dcis[u])
        # the reference's Viterbi decoder is not tail-biting aware, so the last bits it puts out -- the low bits of the RNTI-masked
        # CRC -- are its least reliable: a random-access DCI can come back a second time (aggregation level 8) under a neighbouring
        # RA-RNTI; and with 60 RA-RNTIs accepted, noise passes the 16-bit CRC once in about a thousand decodes.  Anything found
        # beyond what was sent must be an RA-RNTI, and there must be few of them.
        for f in found - sent:
            assert f[0] <= 0x3C, (u, f, dcis[u])
        n_extra += len(found - sent)
    assert n_extra <= n // 8
    if n_ant == 1:
        case = dict(fft=fft, nrb=nrb, n_ant=1, cell=None, phich_res=1.0)
        for u in range(0, n, 8):
            c1 = dict(case, cell=int(cell[u]), sfs=[int(sfs[u])], grids=g[u:u + 1])
            w_rc, w_cfi, w_nsym, recs = td.ref_pdcch_decode(ref, c1)[0]
            assert (w_rc, w_cfi, w_nsym) == (int(rc[u]), int(cfi[u]), int(nsym[u]))
            assert recs == td.dci_records(got[u]), u
