"""GPU parity at scale: thousands of seeded random cases against the COMPILED REFERENCE (not the restatement).

Downlink (~20 000 cases, tests/fuzz_cases.py): the reference's own transmitter makes the capture (real 1 / 2 / 4-port transmit
diversity), the reference's receiver -- liblte_phy_get_dl_subframe_and_ce + liblte_phy_pdsch_channel_decode, run in
oracle/ref/ref_fuzz.cc on every host core, from the `oracle_big` build so that full-band allocations are inside its scratch --
gives the expected grid, soft bits, verdict and transport block.  The GPU side batches every case of a (bandwidth, ports,
control-region size) group into ONE front-end launch and ONE PDSCH plan run:

* exact stage: the reference's receive grid is uploaded, so RE extraction, pre-decoding, layer de-mapping, de-mapping,
  descrambling, rate un-matching, the REF turbo decoder and the CRC see the reference's inputs: soft bits, verdict and bits must be
  IDENTICAL, every case;
* tolerance stage: the library's own front end on the same capture: rx_symb / rx_ce within the FFT tolerance, and the
  chain on its own grid must reach the reference's verdict (a soft bit may sit within float rounding of a decision boundary, so this
  is counted, not demanded bit for bit -- the count is asserted and reported).

Uplink (>= 3000 allocations): random cells / hopping modes / cyclic shifts / widths / positions through the library's uplink
transmitter, liblte_phy_get_ul_subframe + liblte_phy_pusch_channel_decode as the checker.  The uplink has no exact stage: the
SC-FDMA demodulator AND the transform pre-decoding DFT in front of the de-mapper are floating point in an order FFTW does not
specify, so a soft bit whose equalised value lies within float rounding of zero may come out with the other sign, and a QPSK symbol whose
127 * sd lies within that rounding of an integer with a magnitude one unit off (both of its bits).  Measured: 2 of 3.9 M soft bits in the
first 1278 allocations; round 4's soaks: 14 of 158 M, ten sign flips and two magnitude steps (four bits).  The test therefore COUNTS differing soft bits (<= 1e-5 of all, reported), and
demands identical verdicts and transport blocks for every allocation whose soft bits are identical.

A report of what was run (counts per dimension, verdict mix, worst tolerances) goes to gpurun_out/fuzz_report.json."""
import json
import os
import time

import numpy as np
import pytest

import fuzz_cases as fz
import lte_testdata as td

pytestmark = pytest.mark.gpu

N_DL_CHUNKS, DL_CHUNK = 20, 1000
N_UL_GROUPS = 250
N_SYNC = 40
# every draw below is offset by this: the suite runs with 0; tools/fuzz_soak.sh runs the same tests over further seeds
SEED = int(os.environ.get("MI_LTE_FUZZ_SEED", "0"))
TOL_SYMB, TOL_CE = 1e-5, 1e-4
REPORT = {}


@pytest.fixture(scope="module")
def ref_big():
    from oracle import pyoracle
    L = pyoracle.ref_big()
    if L is None:
        pytest.skip("oracle/_ref/libref_oracle_big.so not built (needs /root/reference)")
    return L


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


def write_report():
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "fuzz_report.json"), "w") as f:
            json.dump(REPORT, f, indent=1, sort_keys=True)
    except OSError:
        pass


def run_group(ctx, cases, idx, r, stats, bad):
    """One (fft, N_rb, N_ant, N_pdcch_symbs) group of a chunk on the GPU."""
    import openlte_amd as m
    c0 = cases[idx[0]]
    n_ant, n = c0["n_ant"], len(idx)
    cfg = m.DlCfg(c0["fft"], c0["n_rb"], n_ant, m.IQ_I8)
    sfs = np.array([cases[i]["sf"] for i in idx], np.uint32)
    cells = np.array([cases[i]["cell"] for i in idx], np.uint32)
    allocs = [m.make_alloc(j, cases[i]["mod"], cases[i]["tbs"], cases[i]["prb0"], cases[i]["rnti"], cases[i]["rv"], cases[i]["tx_mode"], cases[i]["prb1"])
              for j, i in enumerate(idx)]
    nf = ctx.subframe_floats(n_ant)
    assert nf == (2 + 2 * n_ant) * 16 * 1200
    # ---- exact stage on the reference's grid
    grid = np.ascontiguousarray(r["planes"][idx, :2 + 2 * n_ant]).reshape(-1)
    d_sub = ctx.to_device(grid)
    plan = ctx.pdsch_plan(cfg, c0["n_sym"], allocs)
    st, bits = plan.run(d_sub, sfs, cells)
    for j, i in enumerate(idx):
        c = cases[i]
        key = (i, c["n_rb"], n_ant, c["cell"], c["sf"], c["n_sym"], c["mod"], c["tbs"], c["rv"], c["tx_mode"], len(c["prb0"]), c["prb0"] != c["prb1"], c["snr"])
        e = plan.soft_bits(j)
        ne = int(r["n_soft"][i])
        if len(e) != ne or not (e == r["soft"][i, :ne]).all():
            bad.append(("soft", key, int(len(e)), ne, int((e[:min(len(e), ne)] != r["soft"][i, :min(len(e), ne)]).sum())))
            continue
        if (st[j] == 0) != (r["rc"][i] == 0):
            bad.append(("verdict", key, int(st[j]), int(r["rc"][i])))
            continue
        if st[j] == 0 and not (bits[j] == r["bits"][i, :c["tbs"]]).all():
            bad.append(("bits", key))
            continue
        stats["exact_ok"] += 1
        stats["decoded"] += int(st[j] == 0)
    plan.close()
    d_sub.free()
    # ---- tolerance stage: own front end on the capture, then the chain on its own grid
    d_iq = ctx.to_device(np.ascontiguousarray(r["iq"][idx]).reshape(-1, 2))
    d_start = ctx.to_device((np.arange(n) * fz.UNIT_CAP).astype(np.uint64))
    d_sf, d_cell = ctx.to_device(sfs), ctx.to_device(cells)
    d_own = ctx.alloc(n * nf * 4)
    d_own.zero()
    ctx.dl_frontend_dev(cfg, d_iq, None, d_start, d_sf, d_cell, n, d_own)
    own = d_own.download(np.float32).reshape(n, 2 + 2 * n_ant, 16, 1200)
    n_sc = 12 * c0["n_rb"]
    for j, i in enumerate(idx):
        want = r["planes"][i]
        es = max(rel_l2(own[j, 0, :14, :n_sc], want[0, :14, :n_sc]), rel_l2(own[j, 1, :14, :n_sc], want[1, :14, :n_sc]))
        ec = max(rel_l2(own[j, 2:2 + n_ant, :14, :n_sc], want[2:2 + n_ant, :14, :n_sc]), rel_l2(own[j, 2 + n_ant:, :14, :n_sc], want[2 + n_ant:2 + 2 * n_ant, :14, :n_sc]))
        stats["worst_symb"], stats["worst_ce"] = max(stats["worst_symb"], es), max(stats["worst_ce"], ec)
        if es >= TOL_SYMB or ec >= TOL_CE:
            tie = ce_phase_tie(own[j], want, n_ant, n_sc) if es < TOL_SYMB else None
            key = (i, cases[i]["n_rb"], n_ant, cases[i]["cell"], cases[i]["sf"], cases[i]["snr"], es, ec)
            if tie is not None and tie["accepted"]:
                stats.setdefault("ce_ties", []).append([str(key), tie])
            else:
                stats["tol_fail"].append(key + ((str(tie),) if tie is not None else ()))
    plan = ctx.pdsch_plan(cfg, c0["n_sym"], allocs)
    st2, bits2 = plan.run(d_own, sfs, cells)
    for j, i in enumerate(idx):
        same = (st2[j] == 0) == (r["rc"][i] == 0) and (st2[j] != 0 or (bits2[j] == r["bits"][i, :cases[i]["tbs"]]).all())
        stats["own_same"] += int(same)
        if not same:
            stats["own_diff"].append((i, cases[i]["n_rb"], n_ant, cases[i]["mod"], cases[i]["tbs"], cases[i]["snr"], int(st2[j]), int(r["rc"][i])))
    plan.close()
    for b in (d_iq, d_start, d_sf, d_cell, d_own):
        b.free()


def test_downlink_fuzz_against_the_compiled_reference(ctx, ref_big):
    stats = dict(exact_ok=0, decoded=0, own_same=0, worst_symb=0.0, worst_ce=0.0, tol_fail=[], own_diff=[])
    bad, total = [], 0
    dims = dict(n_rb={}, n_ant={}, mod={}, sf={}, n_sym={}, rv={}, tx_mode={}, kind={})
    t_ref = t_gpu = 0.0
    for chunk in range(N_DL_CHUNKS):
        cases = fz.draw_dl_cases(DL_CHUNK, 1000 + chunk + 100000 * SEED)
        t0 = time.time()
        r = fz.run_ref_dl(ref_big, cases)
        t_ref += time.time() - t0
        assert (r["rc_tx"] == 0).all() and (r["rc_fe"] == 0).all()
        groups = {}
        for i, c in enumerate(cases):
            groups.setdefault((c["fft"], c["n_rb"], c["n_ant"], c["n_sym"]), []).append(i)
            for k in ("n_rb", "n_ant", "mod", "sf", "n_sym", "rv", "tx_mode"):
                dims[k][str(c[k])] = dims[k].get(str(c[k]), 0) + 1
            kind = ("filler" if (c["tbs"] + 24) not in td.ALL_K else "punctured" if 3 * (c["tbs"] + 28) > c["e"] else "repeated" if c["e"] >= 6 * (c["tbs"] + 28) else "rate_third")
            dims["kind"][kind] = dims["kind"].get(kind, 0) + 1
            dims["kind"]["one_prb"] = dims["kind"].get("one_prb", 0) + (len(c["prb0"]) == 1)
            dims["kind"]["full_band"] = dims["kind"].get("full_band", 0) + (len(c["prb0"]) == c["n_rb"])
            dims["kind"]["slot_hopping"] = dims["kind"].get("slot_hopping", 0) + (c["prb0"] != c["prb1"])
            dims["kind"]["beyond_10000_soft_bits"] = dims["kind"].get("beyond_10000_soft_bits", 0) + (c["e"] > 10000)
            dims["kind"]["four_ports_subframe_0_or_5"] = dims["kind"].get("four_ports_subframe_0_or_5", 0) + (c["n_ant"] == 4 and c["sf"] in (0, 5))
        t0 = time.time()
        for key, idx in groups.items():
            run_group(ctx, cases, idx, r, stats, bad)
        t_gpu += time.time() - t0
        total += len(cases)
    REPORT["downlink"] = dict(cases=total, identical_soft_bits_verdict_and_bits=stats["exact_ok"], decoded_by_both=stats["decoded"], mismatches=[list(map(str, b)) for b in bad[:50]],
                              own_front_end_same_verdict_and_bits=stats["own_same"], own_front_end_differences=stats["own_diff"][:50],
                              worst_rel_l2_rx_symb=stats["worst_symb"], worst_rel_l2_rx_ce=stats["worst_ce"], tolerance_failures=[list(map(str, x)) for x in stats["tol_fail"][:50]],
                              channel_estimate_phase_ties=stats.get("ce_ties", []),
                              dimensions=dims, seconds_reference=round(t_ref, 1), seconds_gpu_side=round(t_gpu, 1))
    write_report()
    assert total >= 19000
    assert not bad, bad[:10]
    assert stats["exact_ok"] == total
    assert stats["decoded"] >= 0.3 * total
    assert not stats["tol_fail"], stats["tol_fail"][:10]
    # own grid: a verdict may flip where a soft bit sits within float rounding of a decision boundary and the block is marginal
    # (observed: 20 000 of 20 000 per run, 840 000 of 840 000 over round 5's soak; the gate was 0.5 % until round 6)
    assert stats["own_same"] >= total - 2, (stats["own_same"], total, stats["own_diff"][:10])


def test_uplink_fuzz_against_the_compiled_reference(ctx, ref):
    groups = fz.draw_ul_groups(N_UL_GROUPS, 2026 + 100000 * SEED)
    fz.synth_ul_groups(groups)
    t0 = time.time()
    fz.run_ref_ul(ref, groups)
    t_ref = time.time() - t0
    n_alloc = n_ok = n_soft = n_soft_diff = verdict_diff_after_soft_diff = 0
    bad, soft_diff, worst, soft_values = [], [], 0.0, []
    for gi, g in enumerate(groups):
        n = len(g["sfs"])
        ul_len = g["iq"].shape[1]
        symb, d_sub = ctx.ul_frontend(g["cfg"], g["iq"].reshape(-1, 2), np.arange(n) * ul_len, keep=True)
        plan = ctx.pusch_plan(g["cfg"], g["ulcfg"], g["sfs"], [g["cell"]] * n, g["mi_allocs"])
        try:
            st, bits = plan.run(d_sub)
            soft = [plan.soft_bits(a) for a in range(len(g["mi_allocs"]))]
        finally:
            plan.close()
            d_sub.free()
        n_sc = 12 * g["n_rb"]
        err = rel_l2(symb[:, :, :14, :n_sc], g["ref_symb"][:, :, :, :n_sc])
        worst = max(worst, err)
        if err >= TOL_SYMB:
            bad.append(("symb", gi, g["n_rb"], g["cell"], err))
        for k, ((u, mod, tbs, prbs, rnti), (rc, wbits, wsoft)) in enumerate(zip(g["allocs"], g["ref"])):
            key = (gi, g["n_rb"], g["cell"], g["ulc"], g["sfs"][u], len(prbs), prbs[0], tbs, g["snr"])
            n_alloc += 1
            n_soft += len(wsoft)
            if soft[k].shape != wsoft.shape:
                bad.append(("soft count", key, len(soft[k]), len(wsoft)))
            elif not (soft[k] == wsoft).all():
                nd = int((soft[k] != wsoft).sum())
                n_soft_diff += nd
                soft_diff.append((key, nd, int(st[k]), rc))
                # what the differing values are: (position, library, reference).  A QPSK symbol's two soft bits carry ONE magnitude,
                # (int8)(127 * sd), and the quadrant's signs (liblte_phy.cc:9543-9570), so a lone differing bit with the same magnitude on both
                # sides is a quadrant decision on a component that the two SC-FDMA transforms' rounding put on opposite sides of zero; a
                # magnitude step -- 127 * sd within the transforms' rounding of an integer -- shows in both bits of the symbol, one unit, signs kept
                # (two symbols in 63 M soft bits over eight seeds of round 4's soak, profiles/r04_fuzz_soak/)
                q_m = {0: 1, 1: 2, 2: 4, 3: 6}[mod]
                for i in np.nonzero(soft[k] != wsoft)[0][:8]:
                    room = steps_behind_an_extrapolated_estimate(g, u, prbs, (int(i) // q_m) % 12)  # (soft bit (k 12 + s) Q_m + q belongs to symbol s)
                    soft_values.append([str(key), int(i), int(soft[k][i]), int(wsoft[i]), int(soft[k][i ^ 1]), int(wsoft[i ^ 1]), room])
                if (st[k] == 0) != (rc == 0):
                    verdict_diff_after_soft_diff += 1
            elif (st[k] == 0) != (rc == 0):
                bad.append(("verdict", key, int(st[k]), rc))
            elif rc == 0 and not (bits[k] == wbits).all():
                bad.append(("bits", key))
            else:
                n_ok += int(rc == 0)
    REPORT["uplink"] = dict(groups=len(groups), units=sum(len(g["sfs"]) for g in groups), allocations=n_alloc, decoded_by_both=n_ok, mismatches=[list(map(str, b)) for b in bad[:50]],
                            worst_rel_l2_rx_symb=worst, seconds_reference=round(t_ref, 1), soft_bits=n_soft, soft_bits_differing=n_soft_diff,
                            allocations_with_differing_soft_bits=[list(map(str, x)) for x in soft_diff[:50]],
                            verdicts_differing_in_those=verdict_diff_after_soft_diff,
                            differing_soft_bits_position_library_reference_and_the_symbols_other_bit=soft_values[:50],
                            differing_soft_bits_that_are_sign_flips_of_equal_magnitude=sum(sign_flip(v) for v in soft_values),
                            differing_soft_bits_that_are_one_step_of_the_symbols_magnitude=sum(magnitude_step(v) for v in soft_values),
                            differing_soft_bits_behind_an_equalised_outlier=int(sum(bool(outlier(v) and not sign_flip(v) and not magnitude_step(v)) for v in soft_values)))
    write_report()
    # every differing soft bit is either the sign of a component at zero (same magnitude, opposite sign, the symbol's other bit untouched),
    # or one quantisation step of its symbol's magnitude (both bits of the symbol move by one unit, signs kept), or lies in an allocation
    # whose extrapolated channel magnitude passes next to zero on some resource element of the bit's symbol (steps_behind_an_extrapolated_estimate)
    assert all(sign_flip(v) or magnitude_step(v) or outlier(v) for v in soft_values), soft_values[:10]
    assert n_alloc >= 2500
    assert not bad, bad[:10]
    # Observed per run of ~8 million soft bits / ~3 200 allocations: 0 to 6 differing bits in 0 to 4 allocations (54 seeds: profiles/r05_fuzz_soak,
    # profiles/r06_fuzz_soak -- a magnitude step moves both bits of its symbol, so the counts come in pairs), every one of an accepted kind and none
    # with a verdict behind it.  The gates are twice the worst run seen; until round 6 they were 1e-5 of the bits (80) and 0.5 % of the allocations (15).
    # (Round 6 first set them at 5e-7 / 3 allocations from the one run the review quoted: seeds 149, 153 and 155 of the next soak had 5, 5 and 6 bits.)
    assert n_soft_diff <= 12 and len(soft_diff) <= 8, (n_soft_diff, n_soft, soft_diff[:10])
    assert verdict_diff_after_soft_diff == 0
    assert sum(bool(outlier(v) and not sign_flip(v) and not magnitude_step(v)) for v in soft_values) <= 2
    assert n_ok >= 0.4 * n_alloc


def ce_phase_tie(own, want, n_ant, n_sc):
    """Where the library's channel estimate of a subframe leaves the tolerance, and whether a phase tie explains it.  The reference's
    estimate is polar: the pilots' phases are unwrapped along frequency (u_j = wrap_phase(r_j, u_j-1), liblte_phy.cc:6033-6035), interpolated
    linearly between pilots, and between the CRS symbols along time after another wrap_phase (:6066-6193).  wrap_phase decides by comparing
    a phase step with +-pi; a step that IS pi to within the 1e-7 the two front ends' received symbols differ by goes one way in one of them
    and the other way in the other, and everything interpolated across that step turns the other way round: a handful of neighbouring
    sub-carriers of ONE port, in the symbols between two CRS symbols, are simply different (both estimates are equally valid continuations
    of a channel whose phase jumped by half a turn between two noisy pilots).
    Returns dict(port, subcarriers, symbols, elements, gap = how far the nearest such step in the REFERENCE's own estimate is from pi,
    accepted = the differing elements lie on <= 6 adjacent sub-carriers of one port and gap < 1e-3 rad)."""
    h_o = own[2:2 + n_ant, :14, :n_sc].astype(np.float64) + 1j * own[2 + n_ant:2 + 2 * n_ant, :14, :n_sc]
    h_w = want[2:2 + n_ant, :14, :n_sc].astype(np.float64) + 1j * want[2 + n_ant:2 + 2 * n_ant, :14, :n_sc]
    scale = np.sqrt(np.mean(np.abs(h_w) ** 2))
    bad = np.argwhere(np.abs(h_o - h_w) > 1e-3 * scale)
    if len(bad) == 0:
        return None
    ports, syms, scs = sorted(set(bad[:, 0].tolist())), sorted(set(bad[:, 1].tolist())), sorted(set(bad[:, 2].tolist()))
    out = dict(port=ports, symbols=syms, subcarriers=scs[:12], elements=int(len(bad)), gap=None, accepted=False)
    if len(ports) != 1 or scs[-1] - scs[0] > 6:
        return out
    p = ports[0]
    crs = [0, 4, 7, 11] if p < 2 else [1, 8]
    gaps = []
    lo, hi = max(0, scs[0] - 6), min(n_sc - 1, scs[-1] + 6)
    for k in range(lo, hi + 1):  # along time, between neighbouring CRS symbols (the next subframe's first one is not in the grid)
        for s0, s1 in zip(crs[:-1], crs[1:]):
            gaps.append(np.pi - abs(np.angle(h_w[p, s1, k] * np.conj(h_w[p, s0, k]))))
    for s in crs:                # along frequency, between neighbouring sub-carriers six apart (whatever the port's pilot offset is)
        for k in range(lo, hi - 5):
            gaps.append(np.pi - abs(np.angle(h_w[p, s, k + 6] * np.conj(h_w[p, s, k]))))
    out["gap"] = float(min(gaps))
    out["accepted"] = bool(out["gap"] < 1e-3)
    return out


def sign_flip(v):  # v = [allocation, position, library, reference, library's other bit of the symbol, reference's, the allocation's weakest estimate]
    return v[2] == -v[3] and v[4] == v[5]


def steps_behind_an_extrapolated_estimate(g, u, prbs, s):
    """The reference's uplink channel estimate is polar and LINEAR in time: magnitude mag_b + n f_mag, n = -3 .. 3 steps from the slot's DMRS
    symbol, f_mag = (mag_1 - mag_0) / 7 (get_ulsch_ce, liblte_phy.cc:13745-13780).  Over the outer symbols that is an extrapolation, and where
    one DMRS magnitude is more than 10/3 of the other it passes through zero.  On a resource element next to the crossing the estimate is
    what two nearly equal numbers leave, w = |mag_b + n f_mag| / mean(mag) << 1: the 1e-7 by which the two front ends' received symbols (and
    so the magnitudes) differ becomes 1e-7 / w of the estimate, the element's equalised value (:6708-6736) is 1 / w of a normal one, and the
    transform pre-decoding hands sqrt(M) times its error to EVERY output of that SC-FDMA symbol (:6627-6660; the reference's scaling leaves
    the outputs M times the constellation, so all but the few that land next to a constellation point sit at the de-mapper's cap and show
    nothing).  Returns how many quantisation steps 1 / 127 of the de-mapper that is for data symbol s of the allocation,
    127 sqrt(M) 2e-7 / w^2, from the reference's own received symbols in float64 (contiguous allocations, the same resource blocks in both
    slots: what draw_ul_groups makes).  Seed 123 of round 5's soak: w = 0.0055 on one element, one unsaturated output in that symbol, three
    steps between the library and the reference (13 allowed) -- profiles/r05_fuzz_soak/."""
    import openlte_amd as m
    M = 12 * len(prbs)
    sc = 12 * prbs[0] + np.arange(M)
    z = g["ref_symb"][u].astype(np.float64)
    d = m.ul_dmrs_pusch(g["ulcfg"], g["cell"], g["sfs"][u], len(prbs)).astype(np.float64)
    mags = []
    for b, L in ((0, 3), (1, 10)):
        t = (z[0, L, sc] + 1j * z[1, L, sc]) * np.conj(d[2 * b] + 1j * d[2 * b + 1])
        mags.append(np.abs(t))
    b, sp = s // 6, s % 6
    n = sp - 3 if sp < 3 else sp - 2
    w = float(np.abs(mags[b] + n * (mags[1] - mags[0]) / 7).min()) / float(np.mean(mags))
    return float(127 * np.sqrt(M) * 2e-7 / max(w * w, 1e-30))


# (where the extrapolated magnitude passes through zero inside a symbol the formula's room is unbounded; no evidence of that kind excuses more than
# a quarter of the de-mapper's range -- a de-mapper or transform regression confined to such allocations must still fail)
OUTLIER_ROOM_CAP = 32.0
# The set of accepted kinds of difference is FROZEN (sign_flip, magnitude_step, outlier, ce_phase_tie here; the PSS near tie in test_sync_gpu.py):
# a new kind needs a CPU test in test_fuzz_classifiers_cpu.py that fails without it and an entry in DESIGN.md section 4 before it may be added.
ACCEPTED_KINDS = ("sign_flip", "magnitude_step", "outlier", "ce_phase_tie", "pss_near_tie")


def outlier(v):
    """a differing soft bit within the steps its symbol's weakest extrapolated estimate accounts for (at least two: below that it is one of
    the two ordinary kinds or nothing; at most OUTLIER_ROOM_CAP)"""
    room = min(v[6], OUTLIER_ROOM_CAP)
    return bool(v[6] >= 2 and abs(v[2] - v[3]) <= 1 + room and abs(v[4] - v[5]) <= 1 + room)


def magnitude_step(v):
    return abs(v[2] - v[3]) == 1 and abs(v[4] - v[5]) == 1 and abs(v[2]) == abs(v[4]) and abs(v[3]) == abs(v[5]) and v[2] * v[3] > 0 and v[4] * v[5] > 0


def test_control_region_fuzz_against_the_compiled_reference(ctx, ref):
    """PCFICH + PDCCH common search space on ~3 800 random control regions -- six bandwidths x 1 / 2 / 4 ports, random cells,
    subframes, CFIs, 0-3 DCIs per subframe with random RNTI / MCS / allocation, two noise levels (at the lower one part of what is decoded is
    noise) -- in the reference-parity mode (the reference's own arithmetic, port-stride slip included, include/mi_lte.h): return code, CFI,
    N_symbs, and every field of every allocation must equal liblte_phy_pdcch_channel_decode's on a fresh LIBLTE_PHY_STRUCT, unit by unit.
    One kernel launch per configuration; the reference runs on the host cores."""
    import openlte_amd as m
    from openlte_amd import synth
    configs = [(128, 6, 1), (128, 6, 2), (256, 15, 1), (256, 15, 4), (512, 25, 2), (512, 25, 1), (1024, 50, 1), (1024, 50, 2), (2048, 75, 4), (2048, 75, 1),
               (2048, 100, 1), (2048, 100, 2)]
    n, total, n_dci, bad = 320, 0, 0, []
    rntis = [0xFFFF, 0xFFFE] + list(range(1, 0x3D))
    for ci, (fft, nrb, n_ant) in enumerate(configs):
        rng = np.random.default_rng(900 + ci + 100000 * SEED)
        cells = [int(c) for c in rng.choice(504, 6, replace=False)]
        cfg = m.DlCfg(fft, nrb, n_ant, 0)
        sfs, cell = rng.integers(0, 10, n), rng.choice(cells, n)
        cfis = rng.integers(1, 4, n)
        dcis = []
        for u in range(n):
            lst = []
            for r in rng.choice(rntis, int(rng.integers(0, 4 if nrb > 6 else 2)), replace=False):
                npb = int(rng.integers(1, min(nrb // 2, 8) + 1))
                lst.append((int(r), int(rng.integers(0, 27)), npb, int(rng.integers(0, nrb - npb + 1)), int(rng.integers(0, 4))))
            dcis.append(lst)
        half = n // 2
        g = np.concatenate([synth.ctrl_grids(cfg, sfs[:half], cell[:half], cfis[:half], dcis[:half], snr_db=14.0, seed=ci),
                            synth.ctrl_grids(cfg, sfs[half:], cell[half:], cfis[half:], dcis[half:], snr_db=1.0, seed=100 + ci)])
        plan = ctx.pdcch_plan(cfg, cells, 1.0)  # reference-parity mode
        d_g, d_sf, d_cell = ctx.to_device(g), ctx.to_device(sfs.astype(np.uint32)), ctx.to_device(cell.astype(np.uint32))
        rc, cfi, nsym, got = plan.decode_dev(d_g, d_sf, d_cell, n)
        for d in (d_g, d_sf, d_cell):
            d.free()
        plan.close()

        def ref_unit(u):
            return td.ref_pdcch_decode(ref, dict(fft=fft, nrb=nrb, n_ant=n_ant, cell=int(cell[u]), phich_res=1.0, sfs=[int(sfs[u])], grids=g[u:u + 1]))[0]
        want = td.parallel_map(ref_unit, range(n), threads=min(32, fz.n_threads()))
        for u, (w_rc, w_cfi, w_nsym, recs) in enumerate(want):
            mine = (int(rc[u]), int(cfi[u]), int(nsym[u]), td.dci_records(got[u]))
            if mine != (w_rc, w_cfi, w_nsym, recs):
                bad.append((fft, nrb, n_ant, u, int(cell[u]), int(sfs[u]), int(cfis[u]), mine[:3], (w_rc, w_cfi, w_nsym)))
            n_dci += len(recs)
        total += n
    REPORT["control_region"] = dict(subframes=total, configurations=len(configs), dcis_found_by_both=n_dci, mismatches=[list(map(str, b)) for b in bad[:50]])
    write_report()
    assert total >= 3800 and not bad, bad[:10]
    assert n_dci >= total // 6


def test_pbch_fuzz_against_the_compiled_reference(ctx, ref):
    """PBCH: 100 random configurations x 16 units -- bandwidth, transmitted ports 1 / 2 / 4, cell, frame number, SNR from
    hopeless to clean -- through liblte_phy_bch_channel_encode, a random per-port channel and noise (tests/lte_testdata.pbch_case), against
    liblte_phy_bch_channel_decode fed the same grids: return code, port count, position in the 40 ms period and the 24 MIB bits
    identical for every unit, decodes on a wrong hypothesis included."""
    import test_pbch_gpu as tp
    rng = np.random.default_rng(2031 + 100000 * SEED)
    n_units = n_ok = 0
    ports = {1: 0, 2: 0, 4: 0}
    for c in range(100):
        _, fft, nrb = fz.BANDWIDTHS[int(rng.integers(len(fz.BANDWIDTHS)))]
        n_ant = int(rng.choice([1, 2, 4]))
        units = [(int(rng.integers(504)), int(rng.integers(1024))) for _ in range(16)]
        case = td.pbch_case(ref, (fft, nrb, n_ant, units, float(rng.uniform(-7.0, 12.0))), seed=100 + c)
        want = td.ref_pbch_decode(ref, case)
        got = tp.run_case(ctx, case)
        assert got.tolist() == want.tolist(), (c, fft, nrb, n_ant)
        n_units += len(units)
        n_ok += int((want[:, 0] == 0).sum())
        ports[n_ant] += len(units)
    assert 0.25 * n_units < n_ok < n_units  # the draw covers both outcomes
    REPORT["pbch"] = {"units": n_units, "decoded": n_ok, "units_by_transmitted_ports": ports}
    write_report()


def test_prach_fuzz_against_the_compiled_reference(ctx, ref):
    """PRACH: random bandwidths, preamble formats 0-4, root sequences, zero-correlation-zone configurations, unrestricted
    and restricted sets, frequency offsets, preambles, delays and SNRs through the library's transmitter, liblte_phy_detect_prach as the
    checker: detected or not, preamble index and timing advance identical for every occasion."""
    import test_prach_gpu as tpr
    import openlte_amd as m
    rng = np.random.default_rng(839 + 100000 * SEED)
    n_occ = n_det = n_cfg = skipped = past_table = 0
    fmts = {0: 0, 1: 0, 2: 0, 3: 0, 4: 0}
    bws = [(128, 6), (256, 15), (512, 25), (1024, 50), (2048, 100)]
    while n_cfg < 96:
        fft, nrb = bws[int(rng.integers(len(bws) if n_cfg % 7 == 0 else 3))]  # (the two wide ones cost the reference a second per occasion: a few)
        fmt, root, zczc = int(rng.integers(4)), int(rng.integers(838)), int(rng.integers(16))
        hs = int(rng.random() < 0.2)
        if n_cfg >= 80:  # the last sixteen: format 4 (139-point sequences, 138 roots, its own N_cs table; the flag still switches the set arithmetic)
            fmt, root, zczc = 4, int(rng.integers(138)), int(rng.integers(7))
        fo = int(rng.integers(0, nrb - 6 + 1))
        pre = [int(x) for x in rng.integers(0, 64, 4)]
        dly = [int(x) for x in rng.integers(0, 24 * fft // 128, 4)]
        spec = (fft, nrb, root, fmt, zczc, hs, fo, pre, dly, float(rng.uniform(-24.0, 10.0)))
        try:
            case = td.prach_case(spec, seed=200 + n_cfg + skipped)
        except m.MiLteError:  # a restricted-set configuration without 64 preambles: the transmitter refuses it
            skipped += 1
            continue
        try:
            want, _ = td.ref_prach_detect(ref, case)
        except AssertionError:  # ... or the reference's init does
            skipped += 1
            continue
        got, n_roots = tpr.gpu_detect(ctx, case)
        if root + n_roots > (138 if fmt == 4 else 838):
            # the 64-preamble set runs past the end of the root table: the library wraps to index 0 (36.211 5.7.2), the reference reads whatever
            # follows its table (liblte_phy.cc:7168-7171) and, through the recursively averaged threshold of liblte_phy_detect_prach, lets that
            # decide -- found by seed 91 of the soak (format 4, root 134, six roots).  Nothing to compare (the shim, which hands the reference's own
            # root spectra to the detector, stays identical there: tests/test_prach_gpu.py, the caller's-roots form)
            past_table += 1
            continue
        assert (got == want).all(), (spec, got.tolist(), want.tolist())
        n_cfg += 1
        n_occ += len(pre)
        n_det += int((want[:, 0] > 0).sum())
        fmts[fmt] += 1
    assert 0 < n_det < n_occ and skipped < 80 and fmts[4] == 16
    REPORT["prach"] = {"configurations": n_cfg, "occasions": n_occ, "detected": n_det, "refused_configurations": skipped, "configurations_by_format": fmts,
                       "configurations_whose_root_set_wraps_past_the_table_not_compared": past_table}
    write_report()


def test_sync_fuzz_against_the_compiled_reference(ctx, ref, tmp_path):
    """Initial synchronisation: forty random captures from the reference's transmitter (shim/_build/capture_gen) --
    bandwidth, cell, delay anywhere in a frame, carrier offset up to +-3 kHz, 0 to 20 dB -- through the three searches.  Coarse timing
    bit for bit (peak count, symbol starts, the float frequency offsets), PSS / SSS decisions identical, the PSS threshold within 1e-4
    (tests/test_sync_gpu.compare), for every coarse peak of every capture."""
    import test_sync_gpu as ts
    if td.capture_gen_path() is None:
        pytest.skip("shim/_build/capture_gen not built (needs the reference tree at build time)")
    rng = np.random.default_rng(62 + 100000 * SEED)
    bws = [(128, 6, 18), (256, 15, 14), (512, 25, 12), (2048, 100, 10)]
    n_peaks = n_cells = near_ties = 0
    for c in range(N_SYNC):
        fft, nrb, frames = bws[int(rng.integers(3)) if c % 7 else 3]
        n_frame = 307200 * fft // 2048
        spec = (fft, nrb, int(rng.integers(504)), frames, int(rng.integers(n_frame)), float(rng.uniform(-3000.0, 3000.0)), float(rng.uniform(0.0, 20.0)))
        case = td.sync_case(spec, tmp_path, seed=300 + c)
        want = td.ref_sync(ref, case)
        got = ts.gpu_sync(ctx, case)
        near_ties += ts.compare(got, want)
        n_peaks += want["coarse"][0]
        n_cells += sum(1 for p, s in want["per_peak"] if s is not None and 3 * s[0] + p[1] == case["cell"])
    assert n_cells >= N_SYNC // 3  # (a carrier offset beyond ~1 kHz hides the cell from the uncorrected searches: the scanner's loop corrects it first)
    assert near_ties <= 1 + n_peaks // 100  # (see tests/test_sync_gpu.compare: seed 107 has one)
    REPORT["sync"] = {"captures": N_SYNC, "coarse_peaks": n_peaks, "captures_whose_cell_was_found": n_cells, "pss_fine_timing_near_ties": near_ties}
    write_report()
