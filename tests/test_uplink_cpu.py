"""CPU checks of the uplink host code (no GPU): the library's PUSCH DMRS generator against the reference's
generate_dmrs_pusch bit for bit, and its uplink transmitter against the reference's receiver."""
import numpy as np
import pytest

import lte_testdata as td


@pytest.mark.parametrize("cell,ulc", [(17, (3, 0, 0, 2, 5)), (301, (0, 1, 0, 0, 0)), (44, (7, 0, 1, 3, 1)), (503, (29, 1, 1, 7, 7))])
def test_dmrs_bit_exact_vs_reference(ref, cell, ulc):
    import openlte_amd as m
    phy = ref.ref_phy_new(4, cell, 1, 100)
    assert ref.ref_ul_init(phy, cell, *ulc) == 0
    ul = m.UlCfg(*ulc)
    for sf in range(10):
        for n_prb in (1, 2, 3, 4, 5, 6, 8, 10, 14, 22, 25, 50, 96):
            want = np.zeros(4 * 12 * n_prb, np.float32)
            ref.ref_get_pusch_dmrs(phy, sf, n_prb, want)
            got = m.ul_dmrs_pusch(ul, cell, sf, n_prb).reshape(-1)
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (sf, n_prb)
    ref.ref_phy_free(phy)


@pytest.mark.parametrize("name", ["1p4MHz_hop", "5MHz_seqhop_16qam", "20MHz_16ue"])
def test_reference_receiver_decodes_the_host_transmitter(ref, name):
    case = td.ul_case(name)
    _, res = td.ref_ul_decode(ref, case)
    for i, (rc, bits, g) in enumerate(res):
        al = case["allocs"][i]
        if al.mod_type == 1:  # QPSK decodes; every soft bit saturates at +-1 (the reference's pre-decoder scaling)
            assert rc == 0 and (bits == case["tx"][i // case["n_alloc"], i % case["n_alloc"], :al.tbs]).all()
            assert np.abs(g).max() == 1
        else:                 # 16QAM: inner bits are lost to that scaling -> CRC fails, reported as INVALID_INPUTS (1)
            assert rc == 1


def test_prach_transmitter_refuses_what_the_reference_cannot_process():
    """zeroCorrelationZoneConfig 15 with the restricted set is past the 15-entry N_cs table (the reference indexes past it and divides
    by zero, liblte_phy.cc:295, :7190, :7258): the library's PRACH transmitter returns an error instead (it crashed once)."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(128, 6, 1, 0)
    with pytest.raises(m.MiLteError):
        synth.prach_occasions(cfg, m.PrachCfg(175, 0, 15, 1, 0), [3], [0])
    with pytest.raises(m.MiLteError):
        synth.prach_occasions(cfg, m.PrachCfg(175, 0, 16, 0, 0), [3], [0])
    assert synth.prach_occasions(cfg, m.PrachCfg(300, 0, 6, 1, 0), [3], [0]).shape[0] == 1  # a restricted-set configuration it can process


@pytest.mark.parametrize("cell,delta_ss,hop,n_cs_an,shift", [(17, 3, 0, 0, 1), (301, 0, 1, 0, 2), (44, 7, 0, 2, 3), (503, 29, 1, 6, 1), (100, 11, 0, 4, 2)])
def test_pucch_tables_equal_the_references(ref, cell, delta_ss, hop, n_cs_an, shift):
    """mi_lte_ul_pucch_tables (ul_rs.cc) restates what liblte_phy_ul_init computes for PUCCH formats 1 / 1a / 1b through generate_dmrs_pucch
    (liblte_phy.cc:2401-2421, :6986-7129) and what the decoder derives from it (:3058-3083): for every subframe and the resources the
    reference's own tests use, all 352 floats per (subframe, resource) must equal the compiled reference's bit for bit (same host libm).
    shift is the value liblte_phy_ul_init is given (deltaPUCCH-Shift - 1)."""
    import ctypes as C
    import openlte_amd as m
    from oracle import pyoracle as po
    L = m.load_library()
    L.mi_lte_ul_pucch_tables.restype = C.c_int
    phy = ref.ref_phy_new(po.FS_ENUM[2048], cell, 1, 100)
    assert ref.ref_ul_init_pucch(phy, cell, delta_ss, hop, n_cs_an, shift) == 0
    ul = m.UlCfg(delta_ss, hop, 0, 0, 0)
    worst = 0
    for sf in range(10):
        for n1 in (0, 1, 2, 5, 11, 17, 35, 49):
            want, got = np.zeros(352, np.float32), np.zeros(352, np.float32)
            ref.ref_get_pucch_tables(phy, sf, n1, want)
            rc = L.mi_lte_ul_pucch_tables(C.byref(ul), C.c_uint32(cell), C.c_uint32(sf), C.c_uint32(n1), C.c_uint32(n_cs_an), C.c_uint32(shift + 1), C.c_uint32(1),
                                          got.ctypes.data_as(C.c_void_p))
            assert rc == 0
            worst = max(worst, int((got.view(np.uint32) != want.view(np.uint32)).sum()))
            assert (got.view(np.uint32) == want.view(np.uint32)).all(), (sf, n1, np.flatnonzero(got != want)[:8], got[got != want][:4], want[got != want][:4])
    ref.ref_phy_free(phy)
