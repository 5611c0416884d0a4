"""GPU: the product-level batch forms around the kernels -- dynamic PDSCH plans, the multi-device host pipeline,
per-unit allocation lists, contiguous captures split with the look-ahead halo, contexts driven from concurrent host threads, and the
scanner's device-side frequency correction.  None of these has a counterpart in the reference (its API is one subframe per call): the
checks are self-consistency ones -- every form must give, bit for bit, what the device-resident batch entry points give, which the
parity tests (test_chain_gpu.py, test_fuzz_gpu.py) tie to the reference."""
import ctypes as C
import threading

import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu


def device_resident(ctx, cfg, iq, sfs, cells, allocs, cfi):
    """Front end + one static plan over units held in HBM -> (status, [bits])."""
    n = len(sfs)
    d_iq = ctx.to_device(iq.reshape(-1, 2))
    d_start = ctx.to_device((np.arange(n) * iq.shape[1]).astype(np.uint64))
    d_sf, d_cell = ctx.to_device(np.asarray(sfs, np.uint32)), ctx.to_device(np.asarray(cells, np.uint32))
    d_sub = ctx.alloc(n * ctx.subframe_floats(cfg.N_ant) * 4)
    ctx.dl_frontend_dev(cfg, d_iq, None, d_start, d_sf, d_cell, n, d_sub)
    plan = ctx.pdsch_plan(cfg, cfi, allocs)
    st, bits = plan.run(d_sub, sfs, cells)
    plan.close()
    for b in (d_iq, d_start, d_sf, d_cell, d_sub):
        b.free()
    return st, bits


def random_unit_lists(rng, n_units, cfis):
    """0-3 allocations per unit with their own size, modulation and place; the unit's control-region size in every allocation."""
    import openlte_amd as m
    sizes = [k - 24 for k in td.ALL_K if k - 24 >= 16]
    allocs, first = [], [0]
    for u in range(n_units):
        pos = int(rng.integers(0, 20))
        for a in range(int(rng.integers(0, 4))):
            mod, n_prb = int(rng.integers(1, 4)), int(rng.integers(1, 13))
            if pos + n_prb > 100:
                break
            e = n_prb * 12 * (14 - cfis[u]) * (2, 4, 6)[mod - 1]
            fit = [t for t in sizes if 3 * (t + 28) <= 0.85 * e]
            if not fit:
                continue
            allocs.append(m.make_alloc(u, mod, int(fit[-1 - int(rng.integers(0, min(5, len(fit))))]), list(range(pos, pos + n_prb)), 0x200 + 8 * u + a,
                                       n_pdcch_symbs=cfis[u]))
            pos += n_prb + int(rng.integers(0, 9))
        first.append(len(allocs))
    return allocs, np.array(first, np.uint32)


def test_dynamic_plan_mixed_control_region_sizes(ctx):
    """One dynamic plan over units with control regions of 1, 2 and 3 symbols (the allocations carry them, as mi_lte_pdcch_decode_run
    returns them) == one static plan per control-region size; assigned twice: the second assignment reuses the device arrays."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
    rng = np.random.default_rng(5)
    n = 9
    plan = ctx.pdsch_plan_dynamic(cfg, 64, 64 * 20000)
    for rnd in range(2):
        cfis = [1 + (u + rnd) % 3 for u in range(n)]
        sfs, cells = [int(x) for x in rng.integers(0, 10, n)], [int(x) for x in rng.integers(0, 504, n)]
        allocs, first = random_unit_lists(rng, n, cfis)
        # synthesise unit by unit (the transmitter takes one control-region size per call)
        iq = np.zeros((n, synth.unit_len(2048), 2), np.int8)
        tx = {}
        for u in range(n):
            mine = [a for a in allocs if a.unit == u]
            if not mine:
                iq[u], _ = (x[0] for x in synth.dl_units(cfg, [sfs[u]], [cells[u]], [], 0, n_pdcch_symbs=cfis[u], seed=100 * rnd + u))
                continue
            loc = [m.PdschAlloc.from_buffer_copy(a) for a in mine]
            for a in loc:
                a.unit = 0
            q, t = synth.dl_units(cfg, [sfs[u]], [cells[u]], loc, len(loc), n_pdcch_symbs=cfis[u], snr_db=28, seed=100 * rnd + u)
            iq[u] = q[0]
            for k, a in enumerate(mine):
                tx[(u, a.rnti)] = t[0, k, :a.tbs]
        d_iq = ctx.to_device(iq.reshape(-1, 2))
        d_start = ctx.to_device((np.arange(n) * iq.shape[1]).astype(np.uint64))
        d_sf, d_cell = ctx.to_device(np.asarray(sfs, np.uint32)), ctx.to_device(np.asarray(cells, np.uint32))
        d_sub = ctx.alloc(n * ctx.subframe_floats(1) * 4)
        ctx.dl_frontend_dev(cfg, d_iq, None, d_start, d_sf, d_cell, n, d_sub)
        plan.assign(2, allocs)
        st, bits = plan.run(d_sub, sfs, cells)
        assert len(st) == len(allocs) and (st == 0).sum() >= 0.8 * len(allocs)
        for k, a in enumerate(allocs):
            if st[k] == 0:
                assert (bits[k] == tx[(a.unit, a.rnti)]).all()
        for c in (1, 2, 3):  # the static way: one plan per control-region size, n_pdcch_symbs left to the plan
            idx = [k for k, a in enumerate(allocs) if a.n_pdcch_symbs == c]
            if not idx:
                continue
            sub = [m.PdschAlloc.from_buffer_copy(allocs[k]) for k in idx]
            for a in sub:
                a.n_pdcch_symbs = 0
            sp = ctx.pdsch_plan(cfg, c, sub)
            st2, bits2 = sp.run(d_sub, sfs, cells)
            for j, k in enumerate(idx):
                assert st2[j] == st[k] and (bits2[j] == bits[k]).all()
                assert (sp.soft_bits(j) == plan.soft_bits(k)).all()
            sp.close()
        for b in (d_iq, d_start, d_sf, d_cell, d_sub):
            b.free()
    plan.close()


@pytest.mark.parametrize("devices,n_units,chunk", [([0], 10, 4), ([0, 0, 0], 22, 4), ([0, 0], 7, 16)])
def test_multi_device_pipeline_template_equals_device_resident(ctx, devices, n_units, chunk):
    """G "devices" (all ordinal 0 here: one host thread, lanes and contexts per entry) == the single pipeline == the device-resident
    path, ragged last chunk included (n_units is no multiple of the chunk)."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8 | m.CE_COMPACT)
    rng = np.random.default_rng(n_units)
    sfs = [int(x) for x in rng.choice([1, 2, 3, 4, 6, 7, 8, 9], n_units)]
    cells = [int(x) for x in rng.integers(0, 504, n_units)]
    allocs = []
    for u in range(n_units):
        allocs += td.w4_allocs(u)
    iq, tx = synth.dl_units(cfg, sfs, cells, allocs, 9, snr_db=30, max_delay=4, seed=3)
    want_st, want_bits = device_resident(ctx, cfg, iq, sfs, cells, allocs, 2)
    pipe = m.DlPipeline(devices, cfg, 2, td.w4_allocs(0), chunk, n_lanes=2)
    assert pipe.n_devices == len(devices)
    # host placement: the samples next to device 0 (mi_lte_host_alloc_on), the results interleaved -- both must behave like plain pinned
    # memory whatever the box allows (node bound, interleaved, or the runtime's own placement where binding is refused)
    h_iq, h_sf, h_cell = m.HostBuffer(iq.shape, np.int8, device=0), m.HostBuffer((n_units,), np.uint32), m.HostBuffer((n_units,), np.uint32)
    h_out, h_st = m.HostBuffer((n_units * 9, pipe.out_stride), np.uint8, device=-1), m.HostBuffer((n_units * 9,), np.int32)
    dev_node = m.load_library().mi_lte_device_numa_node(0)
    assert h_iq.node in (-1, dev_node) and h_out.node in (-1, -2) and h_sf.node == -1
    h_iq.arr[:], h_sf.arr[:], h_cell.arr[:] = iq, sfs, cells
    for rep in range(2):
        h_out.arr[:], h_st.arr[:] = 0xEE, -7
        pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n_units, h_out.arr, h_st.arr)
        assert (h_st.arr == want_st).all() and (want_st == 0).all()
        for k in range(n_units * 9):
            t = allocs[k].tbs
            assert (np.unpackbits(h_out.arr[k, :t // 8]) == want_bits[k]).all(), k
            assert (want_bits[k] == tx[k // 9, k % 9, :t]).all()
        # the run's own account of itself: every chunk and unit counted once, on the device slot the block-cyclic rule gives it, its
        # bytes over the link, its stream phases timed; host threads of a multi-device pipeline pinned to their device's node where
        # the system names one (the single-device pipeline runs on the caller's thread and leaves it alone)
        st = pipe.device_stats()
        n_chunks = (n_units + chunk - 1) // chunk
        assert [d["chunks"] for d in st] == [len(range(i, n_chunks, len(devices))) for i in range(len(devices))]
        assert sum(d["units"] for d in st) == n_units and all(d["device"] == 0 for d in st)
        assert sum(d["h2d_bytes"] for d in st) == n_units * (pipe.unit_samples * 2 + 8)
        assert sum(d["d2h_bytes"] for d in st) == n_units * 9 * (pipe.out_stride + 4)
        for d in st:
            if d["chunks"]:
                assert d["wall_s"] > 0 and d["kernel_s"] > 0 and d["h2d_s"] > 0 and d["d2h_s"] > 0
            if len(devices) > 1 and dev_node >= 0:
                assert d["numa_node"] == dev_node and d["n_cpus"] > 0
            if len(devices) == 1:
                assert d["numa_node"] == -1 and d["n_cpus"] == 0
    pipe.close()
    for b in (h_iq, h_sf, h_cell, h_out, h_st):
        b.free()


def test_pipeline_per_unit_allocation_lists(ctx):
    """Every unit its own list (0-3 allocations, own control-region size): two "devices", small chunks == one static plan per
    control-region size over the device-resident units."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
    rng = np.random.default_rng(77)
    n = 13
    cfis = [int(x) for x in rng.integers(1, 4, n)]
    sfs, cells = [int(x) for x in rng.integers(0, 10, n)], [int(x) for x in rng.integers(0, 504, n)]
    allocs, first = random_unit_lists(rng, n, cfis)
    iq = np.zeros((n, synth.unit_len(2048), 2), np.int8)
    for u in range(n):
        loc = [m.PdschAlloc.from_buffer_copy(a) for a in allocs if a.unit == u]
        for a in loc:
            a.unit = 0
        iq[u] = synth.dl_units(cfg, [sfs[u]], [cells[u]], loc, len(loc), n_pdcch_symbs=cfis[u], snr_db=26, seed=u)[0][0]
    want_st, want_bits = device_resident(ctx, cfg, iq, sfs, cells, allocs, 2)  # (the allocations carry their own control-region size)
    assert (want_st == 0).sum() >= 0.8 * len(allocs)
    pipe = m.DlPipeline([0, 0], cfg, 2, None, 3, n_lanes=2, max_alloc_per_unit=3)
    arr = (m.PdschAlloc * len(allocs))(*allocs)
    h_iq, h_sf, h_cell = m.HostBuffer(iq.shape, np.int8), m.HostBuffer((n,), np.uint32), m.HostBuffer((n,), np.uint32)
    h_out, h_st = m.HostBuffer((len(allocs), pipe.out_stride), np.uint8), m.HostBuffer((len(allocs),), np.int32)
    h_iq.arr[:], h_sf.arr[:], h_cell.arr[:] = iq, sfs, cells
    h_st.arr[:] = -7
    pipe.run_units(h_iq.arr, h_sf.arr, h_cell.arr, n, arr, first, 2, h_out.arr, h_st.arr)
    assert (h_st.arr == want_st).all()
    for k, a in enumerate(allocs):
        if want_st[k] == 0:
            assert (np.unpackbits(h_out.arr[k, :a.tbs // 8]) == want_bits[k]).all(), k
    # An allocation the plans refuse -- a DCI that passed its CRC on noise: resource blocks past the carrier, more than one code block, a
    # modulation that does not exist -- costs its own slot (status 2, as liblte_phy_pdsch_channel_decode would report for it, and as
    # the per-call forms do), not the run
    L = m.load_library()
    for k, spoil in enumerate((lambda a: a.prb[1].__setitem__(0, 117), lambda a: setattr(a, "tbs", 6200), lambda a: setattr(a, "mod_type", 5),
                               lambda a: setattr(a, "N_prb", 0))):
        junk = (m.PdschAlloc * len(allocs))(*allocs)
        victim = [3, len(allocs) - 1, 0, 7][k]
        spoil(junk[victim])
        assert L.mi_lte_pdsch_alloc_decodable(C.byref(cfg), C.byref(junk[victim]), 2) == 0
        assert L.mi_lte_pdsch_alloc_decodable(C.byref(cfg), C.byref(junk[(victim + 1) % len(allocs)]), 2) == 1
        h_st.arr[:] = -7
        pipe.run_units(h_iq.arr, h_sf.arr, h_cell.arr, n, junk, first, 2, h_out.arr, h_st.arr)
        expect = want_st.copy()
        expect[victim] = 2
        assert (h_st.arr == expect).all(), (k, h_st.arr, expect)
        for j, a in enumerate(allocs):
            if j != victim and want_st[j] == 0:
                assert (np.unpackbits(h_out.arr[j, :a.tbs // 8]) == want_bits[j]).all(), (k, j)
    # a list that is not sorted by unit is refused, not mis-decoded
    bad = (m.PdschAlloc * len(allocs))(*allocs)
    bad[0].unit = 5
    with pytest.raises(m.MiLteError):
        pipe.run_units(h_iq.arr, h_sf.arr, h_cell.arr, n, bad, first, 2, h_out.arr, h_st.arr)
    pipe.close()
    for b in (h_iq, h_sf, h_cell, h_out, h_st):
        b.free()


def test_contiguous_capture_split_with_the_look_ahead_halo(ctx):
    """a contiguous capture split on subframe boundaries, every chunk copied with the 4 400 samples behind its last subframe:
    chunks of 3 on two "devices" == one chunk holding the whole capture == the front end pointed into the capture in HBM."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
    n, cell, sf0 = 11, 321, 7
    sfs = [(sf0 + u) % 10 for u in range(n)]
    allocs = []
    for u in range(n):
        allocs += td.w4_allocs(u)[:4]
    units, tx = synth.dl_units(cfg, sfs, [cell] * n, allocs, 4, snr_db=30, max_delay=-1, gain=(1.0, 1.0), seed=9)  # static channel: the units join up
    lead = 777
    cap = np.zeros((lead + n * 30720 + 4400, 2), np.int8)
    for u in range(n):  # subframe u, then (overwritten by the next subframe) its own look-ahead symbols
        cap[lead + u * 30720: lead + u * 30720 + 30720 + 4400] = units[u, :30720 + 4400]
    first = np.arange(0, 4 * n + 1, 4).astype(np.uint32)
    arr = (m.PdschAlloc * len(allocs))(*allocs)
    # reference point: the capture in HBM, unit starts pointing into it
    d_iq = ctx.to_device(cap)
    d_start = ctx.to_device((lead + 30720 * np.arange(n)).astype(np.uint64))
    d_sf, d_cell = ctx.to_device(np.asarray(sfs, np.uint32)), ctx.to_device(np.full(n, cell, np.uint32))
    d_sub = ctx.alloc(n * ctx.subframe_floats(1) * 4)
    ctx.dl_frontend_dev(cfg, d_iq, None, d_start, d_sf, d_cell, n, d_sub)
    plan = ctx.pdsch_plan(cfg, 2, allocs)
    want_st, want_bits = plan.run(d_sub, sfs, [cell] * n)
    plan.close()
    for b in (d_iq, d_start, d_sf, d_cell, d_sub):
        b.free()
    assert (want_st == 0).all()
    h_cap = m.HostBuffer(cap.shape, np.int8)
    h_cap.arr[:] = cap
    for devices, chunk in (([0, 0], 3), ([0], 16)):
        pipe = m.DlPipeline(devices, cfg, 2, None, chunk, n_lanes=2, max_alloc_per_unit=4)
        h_out, h_st = m.HostBuffer((len(allocs), pipe.out_stride), np.uint8), m.HostBuffer((len(allocs),), np.int32)
        h_st.arr[:] = -7
        pipe.run_capture(h_cap.arr, lead, n, sf0, cell, arr, first, 2, h_out.arr, h_st.arr)
        assert (h_st.arr == want_st).all()
        for k, a in enumerate(allocs):
            assert (np.unpackbits(h_out.arr[k, :a.tbs // 8]) == want_bits[k]).all(), (devices, chunk, k)
        # too short a capture for the last subframe's look-ahead is refused
        with pytest.raises(m.MiLteError):
            pipe.run_capture(h_cap.arr[:-10], lead, n, sf0, cell, arr, first, 2, h_out.arr, h_st.arr)
        pipe.close()
        h_out.free()
        h_st.free()
    h_cap.free()


def test_contexts_driven_from_concurrent_host_threads(port):
    """"callable from N host threads", one context each -- four threads run front end + PDSCH chain + a stand-alone turbo
    decode on their own contexts at the same time, three rounds; every thread must get what a single thread gets."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
    n_thr = 4
    jobs = []
    for t in range(n_thr):
        sfs, cells = [1 + t, 6 + t % 3], [11 * t + 3, 400 - 7 * t]
        allocs = td.w4_allocs(0) + td.w4_allocs(1)
        iq, tx = synth.dl_units(cfg, sfs, cells, allocs, 9, snr_db=30, seed=40 + t)
        K = (1088, 3264, 6144, 512)[t]
        txb, soft = td.turbo_blocks(port, K, 70, "awgn0.5", seed=t)
        jobs.append((sfs, cells, allocs, iq, tx, K, soft))
    serial = m.Context(0)
    want = [(device_resident(serial, cfg, j[3], j[0], j[1], j[2], 2), serial.turbo_decode(j[6], j[5])) for j in jobs]
    serial.close()
    out, errs = [None] * n_thr, []

    def work(t):
        try:
            c = m.Context(0)
            res = None
            for _ in range(3):
                sfs, cells, allocs, iq, tx, K, soft = jobs[t]
                res = (device_resident(c, cfg, iq, sfs, cells, allocs, 2), c.turbo_decode(soft, K))
            out[t] = res
            c.close()
        except Exception as e:  # noqa: BLE001 -- reported by the assertion below
            errs.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(n_thr)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    for t in range(n_thr):
        (st, bits), dec = out[t]
        (wst, wbits), wdec = want[t]
        assert (st == wst).all() and (st == 0).all() and (dec == wdec).all()
        for a, b in zip(bits, wbits):
            assert (a == b).all()


@pytest.mark.parametrize("fs,f_off,n,first", [(30720000, 2350.5, 400000, 0), (1920000, -731.25, 100000, 12345), (7680000, 97.0, 200000, 0)])
def test_device_frequency_shift_is_the_reference_expression(ctx, fs, f_off, n, first):
    """mi_lte_freq_shift_run against LTE_fdd_dl_fs_samp_buf::freq_shift (LTE_fdd_dl_fs_samp_buf.cc:696-713) restated in numpy with the C++
    expression's types: (i+1) -> float, float product with freq_offset, x 2 in float, x M_PI / fs in double, the argument rounded to float
    for cosf / sinf.  Tolerance: the device library's cosf / sinf against the host's (both within a couple of ulp)."""
    rng = np.random.default_rng(3)
    iq = rng.integers(-100, 101, (n, 2)).astype(np.int8)
    d_iq, d_i, d_q = ctx.to_device(iq), ctx.alloc(4 * n), ctx.alloc(4 * n)
    ctx.iq_to_planar(d_iq, n, d_i, d_q)
    assert (d_i.download(np.float32) == iq[:, 0]).all() and (d_q.download(np.float32) == iq[:, 1]).all()
    ctx.freq_shift(d_i, d_q, first, n, f_off, fs)
    gi, gq = d_i.download(np.float32), d_q.download(np.float32)
    i1 = (np.arange(n, dtype=np.uint32) + np.uint32(first) + np.uint32(1)).astype(np.float32)
    t = (i1 * np.float32(f_off)) * np.float32(2)
    arg = (t.astype(np.float64) * np.pi / float(fs)).astype(np.float32)
    cr, ci = np.cos(arg.astype(np.float64)).astype(np.float32), np.sin(arg.astype(np.float64)).astype(np.float32)  # correctly rounded cosf / sinf
    a, b = iq[:, 0].astype(np.float32), iq[:, 1].astype(np.float32)
    wi, wq = a * cr + b * ci, b * cr - a * ci
    err = np.sqrt(((gi - wi) ** 2 + (gq - wq) ** 2).sum() / (wi ** 2 + wq ** 2).sum())
    assert err < 5e-7, err
    assert np.abs(gi - wi).max() <= 3e-5 * 142 and np.abs(gq - wq).max() <= 3e-5 * 142
    for b_ in (d_iq, d_i, d_q):
        b_.free()
