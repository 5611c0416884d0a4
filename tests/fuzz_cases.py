"""Seeded random cases for the differential tests against the compiled reference (test side only).

Downlink: one PDSCH allocation per case, every dimension the receive chain branches on drawn at random -- bandwidth (all six, odd
N_rb included), antenna ports 1 / 2 / 4, cell, subframe 0-9 (0 and 5: PBCH / PSS / SSS windows), control-region size 1-3 (+1 at
N_rb <= 10), modulation, PRB set (contiguous of any width incl. one PRB and the full band, scattered, different per slot), transport
block size (rate <= 1/3, punctured, heavy repetition, filler bits F > 0), redundancy version, transmission mode, SNR, channel gains
and delay.  The capture of a case is made by the REFERENCE's transmitter inside oracle/ref/ref_fuzz.cc, so 2- and 4-port cases are
real transmit-diversity signals.

Uplink: units (cell, hopping mode, cyclic shifts, subframe) with up to four QPSK PUSCH allocations of every width the reference has
a DFT plan for.
"""
import ctypes as C
import os

import numpy as np

import lte_testdata as td

BANDWIDTHS = [(0, 128, 6), (1, 256, 15), (2, 512, 25), (3, 1024, 50), (4, 2048, 75), (4, 2048, 100)]  # (fs_enum, fft, N_rb)
SIZES = [k - 24 for k in td.ALL_K if k - 24 >= 16]


def n_threads():
    return max(1, min(64, len(os.sched_getaffinity(0))))


def re_count(n_rb, n_ant, cell, sf, n_sym, prb0, prb1):
    """PDSCH resource elements of an allocation (liblte_phy.cc:3744-3802)."""
    win = {6: (0, 71), 15: (54, 125), 25: (114, 185), 50: (264, 335), 75: (414, 485), 100: (564, 635)}[n_rb]
    n = 0
    for L in range(n_sym, 14):
        l7 = L % 7
        crs = (2 if l7 in (0, 4) else 0) if n_ant == 1 else (4 if (l7 in (0, 4) or (n_ant == 4 and l7 == 1)) else 0)
        inwin = (sf == 0 and 7 <= L <= 10) or (sf in (0, 5) and L in (5, 6))
        for p in (prb0 if L < 7 else prb1):
            if not inwin or p * 12 + 11 < win[0] or p * 12 > win[1]:
                n += 12 - crs
                continue
            for j in range(12):
                sc = p * 12 + j
                if n_ant == 1:
                    is_crs = (l7 == 0 and cell % 6 == j % 6) or (l7 == 4 and (cell + 3) % 6 == j % 6)
                else:
                    is_crs = crs and cell % 3 == j % 3
                if not is_crs and not (win[0] <= sc <= win[1]):
                    n += 1
    return n


def draw_dl_cases(n, seed, bw_weights=(3, 3, 3, 2, 2, 3), big_share=0.06):
    """List of dicts, sorted by (bandwidth, N_ant) so that the reference's workers re-initialise rarely."""
    rng = np.random.default_rng(seed)
    out = []
    w = np.array(bw_weights, float) / sum(bw_weights)
    for k in range(n):
        fs, fft, n_rb = BANDWIDTHS[int(rng.choice(6, p=w))]
        n_ant = int(rng.choice([1, 2, 4], p=[0.5, 0.25, 0.25]))
        cell, sf = int(rng.integers(0, 504)), int(rng.integers(0, 10))
        cfi = int(rng.integers(1, 4))
        n_sym = cfi + (1 if n_rb <= 10 else 0)
        mod = int(rng.choice([0, 1, 2, 3], p=[0.04, 0.32, 0.32, 0.32]))
        qm = (1, 2, 4, 6)[mod]
        # PRB set
        shape = rng.random()
        if shape < 0.08:
            n_prb = 1
        elif shape < 0.08 + big_share:
            n_prb = n_rb if rng.random() < 0.5 else int(rng.integers(max(1, n_rb // 2), n_rb + 1))
        else:
            n_prb = int(rng.integers(1, min(n_rb, 20) + 1))
        kind = rng.random()
        if kind < 0.7:
            first = int(rng.integers(0, n_rb - n_prb + 1))
            if rng.random() < 0.3 and n_rb > 6:  # across the PBCH / PSS / SSS window
                lo = max(0, n_rb // 2 - 3 - n_prb + 1)
                first = int(rng.integers(lo, min(n_rb - n_prb, n_rb // 2 + 3) + 1))
            prb0 = list(range(first, first + n_prb))
            prb1 = prb0
        elif kind < 0.85:
            prb0 = sorted(int(x) for x in rng.choice(n_rb, n_prb, replace=False))
            prb1 = prb0
        else:  # distributed: another set in the second slot
            prb0 = sorted(int(x) for x in rng.choice(n_rb, n_prb, replace=False))
            prb1 = sorted(int(x) for x in rng.choice(n_rb, n_prb, replace=False))
        m_re = re_count(n_rb, n_ant, cell, sf, n_sym, prb0, prb1)
        e = m_re * qm
        # transport block
        t = rng.random()
        if t < 0.62:      # E >= 3(K+4): what the reference's decoder is built for
            fit = [s for s in SIZES if 3 * (s + 28) <= e]
            tbs = fit[int(rng.integers(max(0, len(fit) - 8), len(fit)))] if fit else SIZES[0]
        elif t < 0.74:    # punctured
            fit = [s for s in SIZES if e < 3 * (s + 28) <= 2.2 * e]
            tbs = fit[int(rng.integers(0, len(fit)))] if fit else SIZES[-1]
        elif t < 0.90:    # repetition
            fit = [s for s in SIZES if 3 * (s + 28) <= e]
            tbs = fit[int(rng.integers(0, max(1, len(fit) // 3)))] if fit else SIZES[0]
        else:             # filler bits: tbs + 24 between two interleaver sizes
            fit = [s for s in SIZES if 3 * (s + 28) <= e and s > 200]
            base = fit[int(rng.integers(0, len(fit)))] if fit else 504
            tbs = base - 8 * int(rng.integers(1, 4))
            if tbs + 24 in td.ALL_K or tbs < 16:
                tbs = base
        gains = rng.uniform(0.5, 1.5, 4) * np.exp(2j * np.pi * rng.random(4))
        out.append(dict(fs=fs, fft=fft, n_rb=n_rb, n_ant=n_ant, cell=cell, sf=sf, n_sym=n_sym, mod=mod, tbs=int(tbs),
                        rv=int(rng.choice([0, 0, 1, 2, 3])), tx_mode=int(rng.choice([1, 2, 3, 4, 8]) if n_ant > 1 else rng.choice([1, 1, 1, 3, 4, 8])),
                        rnti=int(rng.integers(1, 0xFFF0)), prb0=prb0, prb1=prb1, snr=float(rng.choice([30.0, 20.0, 12.0, 6.0, 1.0])),
                        gains=gains, delay=int(rng.integers(0, max(1, fft * 9 // 256))), seed=int(seed * 100003 + k), n_re=m_re, e=e))
    out.sort(key=lambda c: (c["fs"], c["n_rb"], c["n_ant"]))
    return out


def make_case(fft, n_rb, n_ant, cell, sf, n_sym, mod, tbs, prbs, rnti, rv=0, tx_mode=1, prbs_slot1=None, snr=30.0, seed=1):
    """A hand-written case in the generator's format (for tests that bring their own capture or want one specific geometry)."""
    fs = {128: 0, 256: 1, 512: 2, 1024: 3, 2048: 4}[fft]
    prb1 = list(prbs_slot1 or prbs)
    n_re = re_count(n_rb, n_ant, cell, sf, n_sym, list(prbs), prb1)
    return dict(fs=fs, fft=fft, n_rb=n_rb, n_ant=n_ant, cell=cell, sf=sf, n_sym=n_sym, mod=mod, tbs=tbs, rv=rv, tx_mode=tx_mode, rnti=rnti,
                prb0=list(prbs), prb1=prb1, snr=snr, gains=np.ones(4, complex), delay=0, seed=seed, n_re=n_re, e=n_re * (1, 2, 4, 6)[mod])


def pad_units(iq):
    """int8 [n, unit_len, 2] of any bandwidth -> [n, UNIT_CAP, 2] (the layout run_ref_dl takes captures in)."""
    out = np.zeros((iq.shape[0], UNIT_CAP, 2), np.int8)
    out[:, :iq.shape[1]] = iq
    return out


def to_ref_cases(cases):
    from oracle import pyoracle as po
    arr = (po.RefDlCase * len(cases))()
    for r, c in zip(arr, cases):
        r.fs_enum, r.N_rb_dl, r.N_ant, r.N_id_cell, r.subfr_num, r.N_pdcch_symbs = c["fs"], c["n_rb"], c["n_ant"], c["cell"], c["sf"], c["n_sym"]
        r.mod_type, r.tbs, r.rv_idx, r.tx_mode, r.rnti, r.N_prb = c["mod"], c["tbs"], c["rv"], c["tx_mode"], c["rnti"], len(c["prb0"])
        for i, (a, b) in enumerate(zip(c["prb0"], c["prb1"])):
            r.prb[0][i], r.prb[1][i] = a, b
        r.snr_db, r.peak, r.delay, r.seed = c["snr"], 100.0, c["delay"], c["seed"] & 0xFFFFFFFF
        for p in range(4):
            r.gain_re[p], r.gain_im[p] = float(c["gains"][p].real), float(c["gains"][p].imag)
    return arr


UNIT_CAP = 30720 + 4400  # samples of the largest unit


def run_ref_dl(R, cases, want_planes=True, iq=None):
    """Run the reference (transmitter unless iq is given, then receiver) over the cases.  Returns dict of arrays:
    iq int8 [n, UNIT_CAP, 2], planes float32 [n, 10, 16, 1200] (or None), soft int8 [n, soft_cap], bits uint8 [n, 6144], tx uint8 [n, 6144],
    rc / rc_fe / rc_tx int32 [n], n_soft / n_out uint32 [n]."""
    n = len(cases)
    arr = to_ref_cases(cases)
    gen = iq is None
    if gen:
        iq = np.zeros((n, UNIT_CAP, 2), np.int8)
    soft_cap = max(c["e"] for c in cases) + 64
    planes = np.zeros((n, 10, 16, 1200), np.float32) if want_planes else None
    soft = np.zeros((n, soft_cap), np.int8)
    bits, tx = np.zeros((n, 6144), np.uint8), np.zeros((n, 6144), np.uint8)
    rc = R.ref_dl_cases_run(C.cast(arr, C.c_void_p), n, 1 if gen else 0, iq.ctypes.data, UNIT_CAP * 2, planes.ctypes.data if want_planes else None,
                            10 * 16 * 1200, soft.ctypes.data, soft_cap, bits.ctypes.data, 6144, tx.ctypes.data, n_threads())
    assert rc == 0
    f = lambda name, dt: np.array([getattr(a, name) for a in arr], dt)
    return dict(iq=iq, planes=planes, soft=soft, bits=bits, tx=tx, rc=f("rc", np.int32), rc_fe=f("rc_fe", np.int32), rc_tx=f("rc_tx", np.int32),
                n_soft=f("N_soft", np.uint32), n_out=f("N_out", np.uint32))


# ---------------------------------------------------------------------------------------------------
# uplink

def draw_ul_groups(n_groups, seed, units_per_group=6):
    """Groups of uplink subframe units sharing (bandwidth, cell, ul config) -- liblte_phy_ul_init dominates the reference's time, and a
    PUSCH plan of the library is per configuration too.  Each unit carries n_ue QPSK allocations (the modulation the reference's
    uplink receiver decodes) of random width (every width the reference plans a DFT for, liblte_phy.cc:2360-2377) and position, with a
    transport block sized to the allocation.  Returns a list of dicts(fs, fft, n_rb, cell, ulc, sfs, snr, n_ue, allocs=[(unit, mod, tbs, prbs, rnti)])."""
    rng = np.random.default_rng(seed)
    groups = []
    for g in range(n_groups):
        fs, fft, n_rb = BANDWIDTHS[int(rng.choice(6, p=[0.25, 0.2, 0.2, 0.15, 0.08, 0.12]))]
        cell = int(rng.integers(0, 504))
        hop = int(rng.integers(0, 3))
        ulc = (int(rng.integers(0, 30)), 1 if hop == 1 else 0, 1 if hop == 2 else 0, int(rng.integers(0, 8)), int(rng.integers(0, 8)))
        widths = [w for w in range(2, min(n_rb, 31)) if w % 2 == 0 or w % 3 == 0 or w % 5 == 0]
        n_ue = int(rng.integers(1, 5)) if n_rb >= 15 else 1
        sfs = [int(x) for x in rng.integers(0, 10, units_per_group)]
        allocs = []
        for u in range(units_per_group):
            pos = 0
            for a in range(n_ue):
                room = (n_rb - pos) // (n_ue - a)
                ok = [x for x in widths if x <= room]
                w = int(rng.choice(ok))
                start = pos + int(rng.integers(0, room - w + 1))
                e = 12 * 12 * w * 2
                fit = [t for t in SIZES if 3 * (t + 28) <= e]
                tbs = int(fit[-1 - int(rng.integers(0, min(6, len(fit))))]) if rng.random() < 0.8 else int(fit[int(rng.integers(0, len(fit)))])
                # QPSK is what the reference's uplink receiver decodes (its transform pre-decoding scales by 12 N_prb: 16QAM / 64QAM lose their
                # inner bits and fail the CRC) -- one allocation in eight is 16QAM or 64QAM all the same: both sides must fail alike
                mod = int(rng.choice([1, 1, 1, 1, 1, 1, 1, 2, 3][a % 2:])) if rng.random() < 0.25 else 1
                allocs.append((u, mod, tbs, list(range(start, start + w)), 0x100 + 16 * u + a))
                pos = start + w
        groups.append(dict(fs=fs, fft=fft, n_rb=n_rb, cell=cell, ulc=ulc, sfs=sfs, snr=float(rng.choice([25.0, 12.0, 4.0])), n_ue=n_ue, allocs=allocs,
                           seed=int(seed * 7919 + g)))
    return groups


def synth_ul_groups(groups):
    """The library's host-side uplink transmitter (the reference's own is broken) -> g["iq"] int8 [units, ul_unit_len, 2], g["tx"]."""
    import openlte_amd as m
    from openlte_amd import synth
    for g in groups:
        cfg, ul = m.DlCfg(g["fft"], g["n_rb"], 1, 0), m.UlCfg(*g["ulc"])
        al = [m.make_alloc(u, mod, tbs, prbs, rnti) for (u, mod, tbs, prbs, rnti) in g["allocs"]]
        g["cfg"], g["ulcfg"], g["mi_allocs"] = cfg, ul, al
        g["iq"], g["tx"] = synth.ul_units(cfg, ul, g["sfs"], [g["cell"]] * len(g["sfs"]), al, g["n_ue"], snr_db=g["snr"], max_delay=3, seed=g["seed"] & 0x7FFFFFFF)


def run_ref_ul(R, groups):
    """The reference's uplink receiver over every unit and allocation of the groups (after synth_ul_groups).  Per group: g["ref_symb"]
    float32 [units, 2, 14, 1200], g["ref"] = [(rc, bits or None, soft int8)] per allocation."""
    from oracle import pyoracle as po
    n_units = sum(len(g["sfs"]) for g in groups)
    n_allocs = sum(len(g["allocs"]) for g in groups)
    units, allocs = (po.RefUlUnitCase * n_units)(), (po.RefUlAllocCase * n_allocs)()
    stride = max(g["iq"].shape[1] for g in groups) * 2
    iq = np.zeros((n_units, stride), np.int8)
    u0 = a0 = 0
    for g in groups:
        for k, sf in enumerate(g["sfs"]):
            r = units[u0 + k]
            r.fs_enum, r.N_rb_ul, r.N_id_cell, r.subfr_num = g["fs"], g["n_rb"], g["cell"], sf
            (r.group_assignment_pusch, r.group_hopping_enabled, r.sequence_hopping_enabled, r.cyclic_shift, r.cyclic_shift_dci) = g["ulc"]
            flat = g["iq"][k].reshape(-1)
            iq[u0 + k, :len(flat)] = flat
        for k, (u, mod, tbs, prbs, rnti) in enumerate(g["allocs"]):
            r = allocs[a0 + k]
            r.unit, r.mod_type, r.tbs, r.rnti, r.N_prb = u0 + u, mod, tbs, rnti, len(prbs)
            for i, p in enumerate(prbs):
                r.prb[i] = p
        g["_u0"], g["_a0"] = u0, a0
        u0 += len(g["sfs"])
        a0 += len(g["allocs"])
    soft_cap = max(12 * 12 * len(a[3]) * (1, 2, 4, 6)[a[1]] for g in groups for a in g["allocs"])
    symb = np.zeros((n_units, 2, 14, 1200), np.float32)
    soft, bits = np.zeros((n_allocs, soft_cap), np.int8), np.zeros((n_allocs, 6200), np.uint8)
    rc = R.ref_ul_cases_run(C.cast(units, C.c_void_p), n_units, C.cast(allocs, C.c_void_p), n_allocs, iq.ctypes.data, stride, symb.ctypes.data,
                            soft.ctypes.data, soft_cap, bits.ctypes.data, 6200, n_threads())
    assert rc == 0
    for g in groups:
        nu, na = len(g["sfs"]), len(g["allocs"])
        assert all(units[g["_u0"] + k].rc_fe == 0 for k in range(nu))
        g["ref_symb"] = symb[g["_u0"]:g["_u0"] + nu]
        g["ref"] = []
        for k in range(na):
            r = allocs[g["_a0"] + k]
            g["ref"].append((int(r.rc), bits[g["_a0"] + k, :r.tbs].copy() if r.rc == 0 else None, soft[g["_a0"] + k, :r.N_soft].copy()))
