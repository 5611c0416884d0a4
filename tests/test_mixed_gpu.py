"""Mixed traffic: PDSCH batches whose allocations have dozens of code-block sizes -- the merged decode (turbo.hip: KSeg,
mi_turbo_ref_multi: every kernel launched once over all sizes) against the compiled reference and against the per-size launches.

The allocation lists are bench.py's `chain-mixed` workload (ChainMixedWorkload.draw_lists: fully loaded 20 MHz subframes, 1-25 PRB
allocations inside the reference's 10 000-soft-bit scratch, QPSK / 16QAM / 64QAM, rv 0-3, CFI 1-3, transport block sizes of 36.213 table
7.1.7.2.1-1).  Reference: liblte_phy_pdsch_channel_decode (liblte_phy.cc:3690-3853) -> dlsch_channel_decode (:12762-12872) -> turbo_decode
(:10620-10845), compiled in place (oracle/_ref)."""
import ctypes as C

import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu


def _lists(n_unique, seed):
    import bench
    return bench.ChainMixedWorkload.draw_lists(n_unique, 0, seed=seed)


def _allocs(m, u, lst, cfi):
    return [m.make_alloc(u, mod, tbs, list(range(p0, p0 + n_prb)), rnti, rv, 1, None, cfi) for (mod, tbs, p0, n_prb, rnti, rv) in lst]


def _ref_decode_all(ref, po, lists, iq):
    """Per unique subframe: the reference's grid (symbols + estimates as device-subframe floats) and [(rc, bits)] per allocation."""
    grids, res = [], []
    for u, (sf, cell, cfi, lst) in enumerate(lists):
        q = iq[u]
        i_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), q[:, 0].astype(np.float32)]))
        q_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), q[:, 1].astype(np.float32)]))
        phy, rx = ref.ref_phy_new(4, cell, 1, 100), ref.ref_subframe_new()
        assert ref.ref_get_dl_subframe_and_ce(phy, i_f, q_f, 0, sf, cell, 1, rx) == 0
        grids.append(np.concatenate([po.ref_subframe_view(ref, rx, 0).ravel(), po.ref_subframe_view(ref, rx, 1).ravel(),
                                     po.ref_subframe_view(ref, rx, 2, True)[:1].ravel(), po.ref_subframe_view(ref, rx, 3, True)[:1].ravel()]).astype(np.float32))
        for (mod, tbs, p0, n_prb, rnti, rv) in lst:
            out, n = np.zeros(6200, np.uint8), C.c_uint32()
            la = po.make_alloc(mod, tbs, list(range(p0, p0 + n_prb)), rnti, rv, 1)
            rc = ref.ref_pdsch_channel_decode(phy, rx, C.byref(la), cfi, cell, 1, out, C.byref(n))
            # the decoder's hard decisions whatever the CRC says (the reference copies them out only when it passes, :12860-12869); F = 0 here
            c = np.ctypeslib.as_array(ref.ref_dlsch_c_bits_ptr(phy), shape=(tbs + 24,)).copy()
            assert rc != 0 or (n.value == tbs and (out[:tbs] == c[:tbs]).all())
            res.append((rc, c[:tbs]))
        ref.ref_subframe_free(rx)
        ref.ref_phy_free(phy)
    return grids, res


def _synth(m, cfg, lists, seed):
    from openlte_amd import synth
    iq, tx, allocs = [], [], []
    for u, (sf, cell, cfi, lst) in enumerate(lists):
        al = _allocs(m, u, lst, cfi)
        q, t = synth.dl_units(cfg, [sf], [cell], al, len(al), n_pdcch_symbs=cfi, snr_db=(30.0, 27.0, 24.0)[u % 3], max_delay=8, seed=seed + u)
        iq.append(q[0])
        tx += [t[0, a, :al[a].tbs].copy() for a in range(len(al))]
        allocs += al
    return np.array(iq), tx, allocs


def test_merged_decode_equals_the_compiled_reference(ctx, ref):
    """24 mixed subframes (~220 allocations, ~60 block sizes; every size's last tile partly filled) decoded in ONE launch set over all
    sizes from the REFERENCE's received grid: verdict, bit count and bits of every allocation are the reference's -- including the blocks
    whose CRC fails (their bits are what the reference's decoder leaves)."""
    import openlte_amd as m
    from oracle import pyoracle as po
    lists = _lists(24, 77)
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
    iq, tx, allocs = _synth(m, cfg, lists, 9100)
    grids, want = _ref_decode_all(ref, po, lists, iq)
    assert len({a.tbs for a in allocs}) >= 40, "the case is meant to hold dozens of block sizes"
    d_sub = ctx.to_device(np.concatenate(grids))
    ctx.set_turbo_small_batch(0)  # no "handful of blocks" path: several sizes -> the merged launches
    try:
        plan = ctx.pdsch_plan(cfg, 2, allocs)
        st, bits = plan.run(d_sub, [l[0] for l in lists], [l[1] for l in lists])
        assert "over all block sizes" in ctx.last_kernels(), ctx.last_kernels()
    finally:
        ctx.set_turbo_small_batch(4096)
    n_ok = 0
    for a, (rc, b) in enumerate(want):
        assert st[a] == rc, (a, st[a], rc)
        assert (bits[a] == b).all(), (a, rc)
        n_ok += rc == 0 and bool((b == tx[a]).all())
    assert n_ok >= len(want) // 2, "most blocks should decode to the transmitted bits (%d of %d did)" % (n_ok, len(want))
    plan.close()
    d_sub.free()


def test_merged_decode_equals_the_per_size_launches(ctx):
    """The same batch (96 mixed subframes from the library's own front end, compact estimates, ~900 allocations) through the merged
    launches and through the per-size launches: identical verdicts and bits.  One allocation repeats its block over more than 258 laps of
    the circular buffer (K = 40 behind 13 PRB of 64QAM): the merged kernels' int16 sums do not hold that, it takes the per-size path
    next to the merged ones.  The merged launches are run with either shape of the trellis kernel."""
    import openlte_amd as m
    lists = _lists(96, 78)
    sf, cell, cfi, lst = lists[5]
    mod, tbs, p0, n_prb, rnti, rv = lst[0]
    lists[5] = (sf, cell, 2, [(3, 16, 0, 13, rnti, 0)] + [(mo, t, p, n, r, v) for (mo, t, p, n, r, v) in lst if p >= 13])
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8 | m.CE_COMPACT)
    iq, tx, allocs = _synth(m, cfg, lists, 9200)
    n = len(lists)
    d_iq, d_start = ctx.to_device(iq.reshape(-1, 2)), ctx.to_device((np.arange(n) * iq.shape[1]).astype(np.uint64))
    sfs, cells = np.array([l[0] for l in lists], np.uint32), np.array([l[1] for l in lists], np.uint32)
    d_sf, d_cell = ctx.to_device(sfs), ctx.to_device(cells)
    d_sub = ctx.alloc(n * ctx.subframe_floats(1) * 4)
    ctx.dl_frontend_dev(cfg, d_iq, None, d_start, d_sf, d_cell, n, d_sub)
    plan = ctx.pdsch_plan(cfg, 2, allocs)
    got = {}
    try:
        # merged launches with the lock-step trellis kernel (code blocks on the lanes), merged with the state-parallel one (what a decode of
        # this size takes by itself), and size by size
        for name, small, merged in (("merged", 0, True), ("merged_small", 4096, True), ("per_size", 4096, False)):
            ctx.set_turbo_small_batch(small)
            ctx.set_turbo_merged(merged)
            got[name] = plan.run(d_sub, sfs, cells)
            assert ("over all block sizes" in ctx.last_kernels()) == merged, (name, ctx.last_kernels())
    finally:
        ctx.set_turbo_small_batch(4096)
        ctx.set_turbo_merged(True)
    (st_a, bits_a), (st_b, bits_b) = got["merged"], got["per_size"]
    for other in ("merged", "merged_small"):
        assert (got[other][0] == st_b).all(), other
        assert all((x == y).all() for x, y in zip(got[other][1], bits_b)), other
    ok = sum(int(st_a[a] == 0 and (bits_a[a] == tx[a]).all()) for a in range(len(allocs)))
    assert ok == int((st_a == 0).sum()) and ok >= len(allocs) * 0.7, (ok, len(allocs))
    plan.close()
    for b in (d_iq, d_start, d_sf, d_cell, d_sub):
        b.free()


def test_merged_decode_after_reassignment_of_a_dynamic_plan(ctx):
    """A dynamic plan re-assigned between runs: the merged decode's tables follow the plan's groups (rebuilt when they change, kept when
    they do not)."""
    import openlte_amd as m
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
    ctx.set_turbo_small_batch(0)
    try:
        plan, res = None, []
        for rnd, seed in enumerate((81, 82, 81)):
            lists = _lists(12, seed)
            iq, tx, allocs = _synth(m, cfg, lists, 9300 + seed)
            n = len(lists)
            sfs, cells = np.array([l[0] for l in lists], np.uint32), np.array([l[1] for l in lists], np.uint32)
            got = ctx.dl_frontend(cfg, iq.reshape(-1, 2), (np.arange(n) * iq.shape[1]).astype(np.uint64), sfs, cells)
            d_sub = ctx.to_device(np.ascontiguousarray(got, np.float32))
            if plan is None:
                plan = ctx.pdsch_plan_dynamic(cfg, 256, 256 * 12 * 13 * 25 * 6)
            plan.assign(2, allocs)
            st, bits = plan.run(d_sub, sfs, cells)
            ok = sum(int(st[a] == 0 and (bits[a] == tx[a]).all()) for a in range(len(allocs)))
            assert ok == int((st == 0).sum()) and ok >= len(allocs) * 0.7, (rnd, ok, len(allocs))
            res.append((st.copy(), [b.copy() for b in bits]))
            d_sub.free()
        assert (res[0][0] == res[2][0]).all() and all((x == y).all() for x, y in zip(res[0][1], res[2][1]))
        plan.close()
    finally:
        ctx.set_turbo_small_batch(4096)
