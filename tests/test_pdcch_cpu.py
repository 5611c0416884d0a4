"""CPU checks of the PDCCH host arithmetic against the compiled reference (no GPU): the DCI unpackers on random
payloads, and the PCFICH / candidate resource-element tables against where the reference's transmitter really puts the
symbols, for 1, 2 and 4 antenna ports."""
import ctypes as C

import numpy as np
import pytest

import lte_testdata as td

DCI_SIZES = {6: (21, 9), 15: (22, 11), 25: (25, 13), 50: (27, 13), 75: (27, 14), 100: (28, 15)}


def test_dci_unpackers_match_reference(ref, capfd):
    import openlte_amd as m
    from oracle import pyoracle as po
    rng = np.random.default_rng(1)
    for nrb, sizes in DCI_SIZES.items():
        for fmt, nb in enumerate(sizes):
            for t in range(300):
                payload, rnti, n_ant = int(rng.integers(0, 1 << nb)), [0xFFFF, 0xFFFE, 1, 0x3C][t % 4], [1, 2, 4][t % 3]
                bits = np.array([(payload >> (nb - 1 - i)) & 1 for i in range(nb)] + [0] * 8, np.uint8)
                ra, mcs, prb1 = po.LoAlloc(), C.c_uint32(), np.zeros(110, np.uint32)
                rc_r = ref.ref_dci_unpack(fmt, bits, nb, rnti, nrb, n_ant, C.byref(ra), C.byref(mcs), prb1)
                rc, d = m.dci_unpack(fmt, payload, nb, rnti, nrb, n_ant)
                assert rc == rc_r, (nrb, fmt, hex(payload))
                if rc == 0:
                    n = min(ra.N_prb, 110)
                    assert (ra.N_prb, ra.tbs, mcs.value, ra.rv_idx, ra.tx_mode, ra.mod_type, ra.rnti) == \
                           (d.alloc.N_prb, d.alloc.tbs, d.mcs, d.alloc.rv_idx, d.alloc.tx_mode, d.alloc.mod_type, d.alloc.rnti)
                    assert [x & 255 for x in ra.prb[:n]] == list(d.alloc.prb[0][:n])
                    assert [int(x) & 255 for x in prb1[:n]] == list(d.alloc.prb[1][:n])
    capfd.readouterr()  # the reference prints on every format-0 flag


@pytest.mark.parametrize("fft,nrb,cell,phich_res,cfi", [(2048, 100, 17, 1.0, 2), (1024, 50, 100, 2.0, 3), (512, 25, 44, 0.5, 1), (256, 15, 503, 1.0 / 6, 3),
                                                     (2048, 75, 7, 1.0, 2), (128, 6, 301, 1.0, 2), (2048, 100, 150, 0.5, 1)])
def test_re_tables_match_reference_transmitter(ref, fft, nrb, cell, phich_res, cfi):
    """Four DCIs fill candidates 0..3.  The symbols the 1-port transmitter puts at the table's positions are the QPSK
    sequence d; with 2 ports port 0 carries d/sqrt(2) at the 2-port table's positions (36.211 6.3.4.3); with 4 ports the
    positions are compared."""
    import openlte_amd as m
    dcis = [(0xFFFF, 5, 3, 1, 0), (0xFFFE, 3, 2, 2, 0), (0x0002, 1, 2, 0, 1), (0x0030, 9, 3, 2, 0)]
    n_symbs = cfi + (1 if nrb <= 10 else 0)
    d_pc = d = None
    for n_ant in (1, 2, 4):
        pc, cand = m.pdcch_re_tables(nrb, n_ant, cell, phich_res, n_symbs)
        g_all = td.pdcch_tx_grid(ref, fft, nrb, n_ant, cell, phich_res, 3, cfi, dcis)
        g = g_all[0].reshape(-1)  # port 0
        if n_ant == 4:
            # the transmitter's pre-coder output rows overlap for 4 ports (the same 576-vs-288 stride slip as in the receiver): the
            # values are not 36.211's and stale rows spill over the whole control region, so only containment can be checked
            used = set(np.flatnonzero(np.abs(g_all).sum(axis=0).reshape(-1) > 0).tolist())
            whole4 = [c for c in range(4) if (cand[c, :144] != 0xFFFFFFFF).all()]
            assert set(pc.tolist()) <= used and all({int(x) for x in cand[c, :144]} <= used for c in whole4)
            continue
        present = cand[:4, :144] != 0xFFFFFFFF
        got_pc, got = g[pc], np.where(present, g[np.where(present, cand[:4, :144], 0)], 0)
        if n_ant == 1:
            d_pc, d = got_pc, got
            assert np.allclose(np.abs(d_pc), 1.0, atol=1e-5)
            # candidates whose CCEs all exist carry a DCI each (the transmitter only uses complete ones)
            whole = present.all(axis=1)
            assert np.allclose(np.abs(d[whole]), 1.0, atol=1e-5)
            assert whole.any() or nrb == 6
        else:
            assert np.allclose(got_pc, d_pc / np.sqrt(2), atol=1e-5)
            assert np.allclose(got, d / np.sqrt(2), atol=1e-5)
