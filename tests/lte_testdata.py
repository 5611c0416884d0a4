"""Seeded synthetic inputs shared by the tests (uses the oracle's encoder -- test side only)."""
import numpy as np
import pytest

# Collect a CPU-only checker test twice: unmarked (the `-m "not gpu"` suite) and marked `gpu` (the driver's `-m gpu` run on the GPU
# box), so that the tests which tie the CPU checkers to the compiled reference are proven where the kernels they underwrite run.
on_both_boxes = pytest.mark.parametrize("box", ["cpu", pytest.param("gpu_box", marks=pytest.mark.gpu)])


def turbo_blocks(port, K, n, kind, seed):
    """Return (tx_bits [n,K] uint8, soft [n,3(K+4)] in the reference's interleaved layout).

    kind: 'clean' +-1 floats, 'awgn0.5'/'awgn0.8' BPSK+noise floats, 'hard127' int8 +-127 with 2 % flips,
          'int' integer-valued floats with exact zeros, 'i16' repetition-combined int16 values,
          'rand127' +-127 with random signs and 'noise' uniform int8 -- no code word underneath: every step moves the path
          metrics by the most it can, in directions the trellis does not agree with (the stress case for the kernel's
          modulo-2^16 path metrics).
    """
    rng = np.random.default_rng(seed)
    D = K + 4
    tx = rng.integers(0, 2, (n, K)).astype(np.uint8)
    soft = []
    for b in range(n):
        d = np.zeros(3 * D, np.uint8)
        port.lo_turbo_encode(np.ascontiguousarray(tx[b]), K, d)
        x = 1.0 - 2.0 * d.reshape(3, D).astype(np.float32)  # bit 0 -> +1
        if kind == "clean":
            y = x
        elif kind.startswith("awgn"):
            y = x + float(kind[4:]) * rng.standard_normal((3, D)).astype(np.float32)
        elif kind == "hard127":
            flip = rng.random((3, D)) < 0.02
            y = (127 * x * np.where(flip, -1, 1)).astype(np.int8)
        elif kind == "int":
            y = np.round((x + 0.7 * rng.standard_normal((3, D))) * 9).astype(np.float32)
        elif kind == "rand127":
            y = (127 * (1 - 2 * rng.integers(0, 2, (3, D)))).astype(np.int8)
        elif kind == "noise":
            y = rng.integers(-127, 128, (3, D)).astype(np.int8)
        elif kind == "i16":
            reps = rng.integers(1, 5, (3, D))
            y = (127 * x * reps).astype(np.int16)
            y[rng.random((3, D)) < 0.03] *= -1
        else:
            raise ValueError(kind)
        soft.append(np.ascontiguousarray(y.T).reshape(-1))
    return tx, np.stack(soft)


def oracle_turbo_ref(port, soft, K):
    out = np.zeros((soft.shape[0], K), np.uint8)
    for b in range(soft.shape[0]):
        port.lo_turbo_decode_ref(np.ascontiguousarray(soft[b], dtype=np.float32), K, out[b])
    return out


ALL_K = list(range(40, 513, 8)) + list(range(528, 1025, 16)) + list(range(1056, 2049, 32)) + list(range(2112, 6145, 64))  # the 188 LTE turbo block sizes
OVERFLOW_K = [3584, 4224, 4352, 4480, 4608, 4736, 4928, 4992, 5248, 5312, 5376, 5440, 5568, 5632, 5696, 5824, 5888, 5952, 6080, 6144]


def parallel_map(fn, items, threads=None):
    """The oracle's C functions release the interpreter lock (ctypes) and hold no static state: run them on every host core."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or max(1, min(32, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(fn, items))


# ---------------------------------------------------------------------------------------------------
# subframe units (uses the library's host-side transmitter, openlte_amd.synth)

def w4_allocs(unit, n_prb_total=100, mod=3):
    """8 x 12 PRB (TBS 3240) + 1 x 4 PRB (TBS 1064), 64QAM, rnti 0x100+a."""
    import openlte_amd as m
    out = []
    for a in range(9):
        prbs = list(range(a * 12, a * 12 + 12)) if a < 8 else list(range(96, 100))
        out.append(m.make_alloc(unit, mod, 3240 if a < 8 else 1064, prbs, 0x100 + a))
    return out


def small_allocs(unit, N_rb_dl, mod, tbs, n_prb, rnti=0x1234, first=0):
    import openlte_amd as m
    return [m.make_alloc(unit, mod, tbs, list(range(first, first + n_prb)), rnti)]


def to_lo_alloc(a):
    from oracle import pyoracle as po
    return po.make_alloc(a.mod_type, a.tbs, [a.prb[0][i] for i in range(a.N_prb)], a.rnti, a.rv_idx, a.tx_mode)


def oracle_frontend(port, fft, n_rb, n_ant, iq_unit, sf, cell):
    """Run the port oracle's get_dl_subframe_and_ce on one int8 unit; returns LoSubframe."""
    import ctypes as C
    from oracle import pyoracle as po
    lc = po.LoCfg()
    port.lo_cfg_init(C.byref(lc), fft, n_rb)
    per_sf = 30720 * fft // 2048
    i = np.concatenate([np.zeros(sf * per_sf, np.float32), iq_unit[:, 0].astype(np.float32)])
    q = np.concatenate([np.zeros(sf * per_sf, np.float32), iq_unit[:, 1].astype(np.float32)])
    s = po.LoSubframe()
    rc = port.lo_get_dl_subframe_and_ce(C.byref(lc), np.ascontiguousarray(i), np.ascontiguousarray(q), 0, sf, cell, n_ant,
                                        C.byref(s))
    assert rc == 0
    return lc, s


# ---------------------------------------------------------------------------------------------------
# uplink units: PUSCH allocations the reference's receiver decodes -- QPSK, E >= 3(K+4)
# (its REF decoder treats punctured parity as hard zeros), tbs + 24 a QPP size

UL_CASES = {
    # name: (fft, N_rb_ul, cell, (delta_ss, group_hop, seq_hop, cyclic_shift, cyclic_shift_dci), subframes, [(mod, tbs, prbs, rnti)...])
    "20MHz_3ue": (2048, 100, 17, (3, 0, 0, 2, 5), [1, 4], [(1, 504, list(range(0, 6)), 0x40), (1, 504, list(range(6, 12)), 0x41),
                                                             (1, 904, list(range(20, 30)), 0x42)]),
    "1p4MHz_hop": (128, 6, 301, (0, 1, 0, 0, 0), [0, 9], [(1, 120, [1, 2], 0x55), (1, 256, [3, 4, 5], 0x56)]),
    "5MHz_seqhop_16qam": (512, 25, 44, (7, 0, 1, 3, 1), [3], [(1, 904, list(range(2, 12)), 0x77), (2, 1000, list(range(12, 18)), 0x78)]),
    # the PUSCH widths up to 10 PRB the reference has DFT plans for (N_prb divisible by 2, 3 or 5, liblte_phy.cc:2360-2377):
    # transform sizes 24 .. 120, i.e. every radix of the pre-decoding DFT (9, 3, 5, 8, 4, 2) in first and later passes
    "20MHz_radices": (2048, 100, 23, (5, 0, 0, 1, 2), [2], [(1, 144, [1, 2], 0x62), (1, 256, [3, 4, 5], 0x63),
                                                            (1, 328, list(range(6, 10)), 0x64), (1, 408, list(range(10, 15)), 0x65),
                                                            (1, 504, list(range(15, 21)), 0x66), (1, 712, list(range(21, 29)), 0x67),
                                                            (1, 776, list(range(29, 38)), 0x68), (1, 904, list(range(38, 48)), 0x69)]),
    # BASELINE config 5's shape: 16 UEs in one 20 MHz subframe, 6 PRB QPSK each (what bench.py --workload uplink batches)
    "20MHz_16ue": (2048, 100, 17, (3, 0, 0, 2, 5), [2, 7], [(1, 504, list(range(6 * a, 6 * a + 6)), 0x100 + a) for a in range(16)]),
    "10MHz_prime": (1024, 50, 100, (11, 0, 0, 7, 3), [7, 8], [(1, 1256, list(range(4, 18)), 0x99), (1, 2024, list(range(20, 42)), 0x9A)]),
}


def ul_case(name, snr_db=30.0, seed=5):
    """Build one uplink case: returns dict(cfg, ulcfg, cell, sfs, allocs (unit-major), n_alloc, iq, tx)."""
    import openlte_amd as m
    from openlte_amd import synth
    fft, nrb, cell, ulc, sfs, al = UL_CASES[name]
    cfg, ul = m.DlCfg(fft, nrb, 1, 0), m.UlCfg(*ulc)
    allocs = [m.make_alloc(u, mod, tbs, prbs, rnti) for u in range(len(sfs)) for (mod, tbs, prbs, rnti) in al]
    iq, tx = synth.ul_units(cfg, ul, sfs, [cell] * len(sfs), allocs, len(al), snr_db=snr_db, max_delay=3, seed=seed)
    return dict(cfg=cfg, ulcfg=ul, ulc=ulc, cell=cell, sfs=sfs, allocs=allocs, n_alloc=len(al), iq=iq, tx=tx, fft=fft, nrb=nrb)


def ref_ul_decode(R, case):
    """Run the compiled reference's uplink receiver over a case.  Returns (rx_symb [n_units, 2, 14, 1200],
    [(rc, bits, g_soft int8) per allocation])."""
    import ctypes as C
    from oracle import pyoracle as po
    phy = R.ref_phy_new(po.FS_ENUM[case["fft"]], case["cell"], 1, case["nrb"])
    assert R.ref_ul_init(phy, case["cell"], *case["ulc"]) == 0
    sfp = R.ref_subframe_new()
    symb, res = [], []
    for u, sf in enumerate(case["sfs"]):
        re = np.ascontiguousarray(case["iq"][u, :, 0].astype(np.float32))
        im = np.ascontiguousarray(case["iq"][u, :, 1].astype(np.float32))
        R.ref_subframe_set_num(sfp, sf)
        assert R.ref_get_ul_subframe(phy, re, im, sfp) == 0
        symb.append(np.stack([po.ref_subframe_view(R, sfp, 0)[:14].copy(), po.ref_subframe_view(R, sfp, 1)[:14].copy()]))
        for a in range(case["n_alloc"]):
            al = case["allocs"][u * case["n_alloc"] + a]
            la = po.make_alloc(al.mod_type, al.tbs, [al.prb[0][i] for i in range(al.N_prb)], al.rnti, al.rv_idx, al.tx_mode)
            out, n = np.zeros(6200, np.uint8), C.c_uint32()
            slot1 = [al.prb[1][i] for i in range(al.N_prb)]
            if slot1 != [al.prb[0][i] for i in range(al.N_prb)]:  # a second-slot list of its own (PUSCH hopping)
                rc = R.ref_pusch_channel_decode_slots(phy, sfp, C.byref(la), (C.c_uint32 * 110)(*slot1), case["cell"], 1, out, C.byref(n))
            else:
                rc = R.ref_pusch_channel_decode(phy, sfp, C.byref(la), case["cell"], 1, out, C.byref(n))
            qm = {0: 1, 1: 2, 2: 4, 3: 6}[al.mod_type]
            ng = 12 * 12 * al.N_prb * qm
            g = np.ctypeslib.as_array(R.ref_ulsch_rx_g_bits_ptr(phy), shape=(ng,)).astype(np.int8).copy()
            res.append((rc, out[:al.tbs].copy() if rc == 0 else None, g))
    R.ref_subframe_free(sfp)
    R.ref_phy_free(phy)
    return np.stack(symb), res


# ---------------------------------------------------------------------------------------------------
# PRACH occasions (rest of SURVEY 8f N1)

PRACH_CASES = {
    # name: (fft, N_rb_ul, root_seq_idx, preamble_format, zczc, hs_flag, freq_offset, preamble indices, delays, snr_db)
    "1p4MHz_8roots": (128, 6, 22, 0, 11, 0, 0, [0, 5, 63, 17, 33], [0, 3, 10, 0, 7], 10.0),
    "5MHz_1root": (512, 25, 100, 0, 1, 0, 2, [0, 40, 63], [0, 20, 5], 5.0),
    "3MHz_restricted": (256, 15, 300, 0, 6, 1, 1, [2, 9, 30], [1, 4, 0], 10.0),
    "1p4MHz_noise_only": (128, 6, 22, 0, 11, 0, 0, [0], [0], -30.0),
    # the longer preamble formats: 1 (long prefix), 2 and 3 (the sequence sent twice)
    "1p4MHz_format1": (128, 6, 50, 1, 11, 0, 0, [3, 44], [0, 9], 10.0),
    "5MHz_format2": (512, 25, 700, 2, 4, 0, 3, [0, 21, 63], [0, 30, 11], 5.0),
    "3MHz_format3": (256, 15, 123, 3, 8, 0, 1, [7, 50], [2, 15], 8.0),
    # format 4, the short TDD preamble: N_zc = 139 on 7.5 kHz sub-carriers, its own root and N_cs tables (liblte_phy.cc:2462-2468, :3339-3350)
    "5MHz_format4": (512, 25, 20, 4, 3, 0, 2, [0, 21, 63, 40], [0, 10, 3, 7], 8.0),
    "1p4MHz_format4": (128, 6, 101, 4, 6, 0, 0, [5, 62], [1, 0], 10.0),
}


def prach_case(name, seed=3):
    import openlte_amd as m
    from openlte_amd import synth
    fft, nrb, root, fmt, zczc, hs, fo, pre, dly, snr = PRACH_CASES[name] if isinstance(name, str) else name
    cfg, pc = m.DlCfg(fft, nrb, 1, 0), m.PrachCfg(root, fmt, zczc, hs, fo)
    iq = synth.prach_occasions(cfg, pc, pre, dly, snr_db=snr, seed=seed)
    return dict(cfg=cfg, pc=pc, iq=iq, fft=fft, nrb=nrb, args=(root, fmt, zczc, hs), fo=fo)


def ref_prach_detect(R, case):
    """The compiled reference's liblte_phy_detect_prach over a case -> uint32 [n_occ, 3] (N_det_pre, det_pre, det_ta); the last two
    are reported as 0 when nothing was detected (the reference leaves them untouched)."""
    import ctypes as C
    from oracle import pyoracle as po
    phy = R.ref_phy_new(po.FS_ENUM[case["fft"]], 1, 1, case["nrb"])
    assert R.ref_ul_init_prach(phy, 1, *case["args"]) == 0
    out = []
    for o in range(case["iq"].shape[0]):
        re = np.ascontiguousarray(case["iq"][o, :, 0].astype(np.float32))
        im = np.ascontiguousarray(case["iq"][o, :, 1].astype(np.float32))
        n, p, ta = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        assert R.ref_detect_prach(phy, re, im, case["fo"], C.byref(n), C.byref(p), C.byref(ta)) == 0
        out.append((n.value, p.value if n.value else 0, ta.value if n.value else 0))
    n_roots = R.ref_prach_n_roots(phy)
    re, im = np.zeros((n_roots, 839), np.float32), np.zeros((n_roots, 839), np.float32)
    for r in range(n_roots):
        R.ref_get_prach_root_fft(phy, r, re[r], im[r])
    R.ref_phy_free(phy)
    return np.array(out, np.uint32), (re, im)


# ---------------------------------------------------------------------------------------------------
# control region: PCFICH + PDCCH subframes built with the reference's own transmitter
# (liblte_phy_pdcch_channel_encode puts every DCI, format 1A, at aggregation level 4 in the common search space)

PDCCH_CASES = {
    # name: (fft, N_rb_dl, N_ant, cell, phich_res, [(subframe, cfi, [(rnti, mcs, N_prb, rb_start, rv)...]) per unit], snr_db)
    "20MHz_1ant": (2048, 100, 1, 17, 1.0, [(0, 2, [(0xFFFF, 5, 4, 10, 0)]), (5, 3, [(0xFFFF, 9, 8, 0, 2), (0xFFFE, 3, 2, 40, 0), (0x0002, 1, 3, 20, 0)]),
                                         (7, 1, [(0x0010, 2, 2, 2, 0)]), (9, 2, [])], 12.0),
    # 1.4 MHz: N_symbs = cfi + 1; with cfi = 1 only two CCEs exist and the reference still decodes the half-present candidate
    "1p4MHz_1ant": (128, 6, 1, 301, 1.0, [(1, 2, [(0xFFFF, 4, 2, 1, 1)]), (6, 1, [(0xFFFE, 0, 3, 0, 0)]), (4, 3, [(0x003C, 7, 3, 3, 0)])], 15.0),
    "5MHz_2ant": (512, 25, 2, 44, 0.5, [(3, 2, [(0xFFFF, 6, 4, 3, 0), (0x0001, 2, 2, 20, 3)]), (8, 3, [(0xFFFF, 11, 6, 0, 1)])], 12.0),
    "10MHz_4ant": (1024, 50, 4, 100, 2.0, [(4, 3, [(0xFFFF, 8, 5, 7, 0), (0xFFFE, 1, 2, 30, 0)]), (2, 2, [(0xFFFF, 3, 3, 3, 2)])], 14.0),
    "3MHz_sixth": (256, 15, 1, 503, 1.0 / 6, [(0, 3, [(0xFFFF, 2, 3, 1, 0), (0x0005, 5, 2, 9, 0), (0xFFFE, 7, 2, 6, 0)]), (5, 1, [(0xFFFF, 2, 3, 1, 0)])], 10.0),
    "15MHz_noisy": (2048, 75, 2, 7, 1.0, [(s, 2, [(0xFFFF, 5, 4, 10, 0), (0x0020, 5, 4, 30, 0)]) for s in range(10)], 2.5),
}


def pdcch_case(R, name, seed=11):
    """Units for one case.  Returns dict with the per-unit received grids in the device-subframe layout
    (float32 [n_units, 2 + 2*N_ant, 16, 1200]: rx_symb re, im, rx_ce re[N_ant], im[N_ant]), built in the frequency domain:
    rx = sum_p h_p * tx_p + noise with a smooth random channel per port and a noisy copy of it as the estimate."""
    import ctypes as C
    from oracle import pyoracle as po
    fft, nrb, n_ant, cell, phich_res, units, snr_db = PDCCH_CASES[name]
    rng = np.random.default_rng(seed)
    phy = R.ref_phy_new_phich(po.FS_ENUM[fft], cell, n_ant, nrb, phich_res)
    sfp = R.ref_subframe_new()
    grids = np.zeros((len(units), 2 + 2 * n_ant, 16, 1200), np.float32)
    k = np.arange(1200)
    for u, (sf, cfi, dcis) in enumerate(units):
        R.ref_subframe_clear_tx(sfp, sf)
        assert R.ref_map_crs(phy, sfp, cell, n_ant) == 0
        al = (po.LoAlloc * 6)()
        mcs = np.zeros(6, np.uint32)
        for i, (rnti, m_, nprb, rb0, rv) in enumerate(dcis):
            al[i] = po.make_alloc(1, 0, list(range(rb0, rb0 + nprb)), rnti, rv, 1 if n_ant == 1 else 2, 0)
            mcs[i] = m_
        assert R.ref_pdcch_channel_encode(phy, sfp, cfi, al, mcs, len(dcis), cell, n_ant, phich_res) == 0
        tx_re = np.ctypeslib.as_array(R.ref_subframe_ptr(sfp, 4), shape=(4, 16, 1200))
        tx_im = np.ctypeslib.as_array(R.ref_subframe_ptr(sfp, 5), shape=(4, 16, 1200))
        rx = np.zeros((16, 1200), np.complex64)
        for p in range(n_ant):
            a, tau, ph = rng.uniform(0.6, 1.4), rng.uniform(-2e-3, 2e-3), rng.uniform(-np.pi, np.pi)
            h = (a * np.exp(1j * (ph + 2 * np.pi * tau * k)))[None, :] * np.ones((16, 1))
            rx += (h * (tx_re[p] + 1j * tx_im[p])).astype(np.complex64)
            sig = 10 ** (-snr_db / 20) / np.sqrt(2)
            est = h + 0.2 * sig * (rng.standard_normal(h.shape) + 1j * rng.standard_normal(h.shape))
            grids[u, 2 + p], grids[u, 2 + n_ant + p] = est.real, est.imag
        sig = 10 ** (-snr_db / 20) / np.sqrt(2)
        rx += (sig * (rng.standard_normal(rx.shape) + 1j * rng.standard_normal(rx.shape))).astype(np.complex64)
        grids[u, 0], grids[u, 1] = rx.real, rx.imag
    R.ref_subframe_free(sfp)
    R.ref_phy_free(phy)
    return dict(fft=fft, nrb=nrb, n_ant=n_ant, cell=cell, phich_res=phich_res, sfs=[x[0] for x in units], units=units, grids=grids)


def ref_pdcch_decode(R, case):
    """liblte_phy_pdcch_channel_decode of the compiled reference over a case's grids -> per unit
    (rc, cfi, N_symbs, [(rnti, mcs, tbs, rv_idx, N_prb, tx_mode, mod_type, prb slot 0, prb slot 1) ...])."""
    import ctypes as C
    from oracle import pyoracle as po
    n_ant = case["n_ant"]
    sfp = R.ref_subframe_new()
    out = []
    for u, sf in enumerate(case["sfs"]):
        # a fresh LIBLTE_PHY_STRUCT per subframe: the reference reads CCE scratch left by earlier calls for candidates
        # reaching past the last CCE (include/mi_lte.h), so its result is only a function of the subframe on a fresh struct
        phy = R.ref_phy_new(po.FS_ENUM[case["fft"]], case["cell"], n_ant, case["nrb"])
        R.ref_subframe_set_num(sfp, sf)
        g = case["grids"][u]
        po.ref_subframe_view(R, sfp, 0)[:] = g[0]
        po.ref_subframe_view(R, sfp, 1)[:] = g[1]
        po.ref_subframe_view(R, sfp, 2, True)[:n_ant] = g[2:2 + n_ant]
        po.ref_subframe_view(R, sfp, 3, True)[:n_ant] = g[2 + n_ant:]
        cfi, nsym, nal = C.c_uint32(), C.c_uint32(), C.c_uint32()
        al, mcs, prb1 = (po.LoAlloc * 6)(), np.zeros(6, np.uint32), np.zeros(6 * 110, np.uint32)
        rc = R.ref_pdcch_channel_decode(phy, sfp, case["cell"], n_ant, case["phich_res"], C.byref(cfi), C.byref(nsym), C.byref(nal), al, mcs, prb1)
        recs = []
        if rc != 3:  # 3 = LIBLTE_ERROR_INVALID_CRC: the PCFICH did not decode and nothing else was written
            for a in range(nal.value):
                x, n = al[a], min(al[a].N_prb, 110)
                recs.append((x.rnti, int(mcs[a]), x.tbs, x.rv_idx, x.N_prb, x.tx_mode, x.mod_type, [int(v) & 255 for v in x.prb[:n]],
                             [int(v) & 255 for v in prb1[110 * a:110 * a + n]]))
        out.append((rc, cfi.value if rc != 3 else 0, nsym.value if rc != 3 else 0, recs))
        R.ref_phy_free(phy)
    R.ref_subframe_free(sfp)
    return out


def dci_records(dcis):
    """The same tuple form for a list of openlte_amd.PdcchDci."""
    out = []
    for d in dcis:
        a, n = d.alloc, min(d.alloc.N_prb, 110)
        out.append((a.rnti, d.mcs, a.tbs, a.rv_idx, a.N_prb, a.tx_mode, a.mod_type, list(a.prb[0][:n]), list(a.prb[1][:n])))
    return out


def pdcch_tx_grid(ref, fft, nrb, n_ant, cell, phich_res, sf, cfi, dcis):
    """tx_symb of the reference's transmitter (complex [4, 16, 1200]) for one control region: PCFICH + the DCIs, no CRS."""
    from oracle import pyoracle as po
    phy = ref.ref_phy_new_phich(po.FS_ENUM[fft], cell, n_ant, nrb, phich_res)
    sfp = ref.ref_subframe_new()
    ref.ref_subframe_clear_tx(sfp, sf)
    al, mcs = (po.LoAlloc * 6)(), np.zeros(6, np.uint32)
    for i, (rnti, m_, nprb, rb0, rv) in enumerate(dcis):
        al[i] = po.make_alloc(1, 0, list(range(rb0, rb0 + nprb)), rnti, rv, 1 if n_ant == 1 else 2, 0)
        mcs[i] = m_
    assert ref.ref_pdcch_channel_encode(phy, sfp, cfi, al, mcs, len(dcis), cell, n_ant, phich_res) == 0
    g = np.ctypeslib.as_array(ref.ref_subframe_ptr(sfp, 4), shape=(4, 16, 1200)) + 1j * np.ctypeslib.as_array(ref.ref_subframe_ptr(sfp, 5), shape=(4, 16, 1200))
    g = g.copy()
    ref.ref_subframe_free(sfp)
    ref.ref_phy_free(phy)
    return g


def pdcch_per_port_case(ref, fft, nrb, n_ant, cell, phich_res, units, snr_db, seed=21):
    """Control regions as 36.211 6.3.4.3 transmits them on 2 or 4 ports, which the reference's transmitter does not (its
    pre-coder output rows overlap): the QPSK symbols d come from the reference's 1-port transmitter (read off the grid
    through the library's 1-port tables), and are placed, Alamouti-coded, on the N-port tables' positions."""
    import openlte_amd as m
    rng = np.random.default_rng(seed)
    grids = np.zeros((len(units), 2 + 2 * n_ant, 16, 1200), np.float32)
    k = np.arange(1200)
    r2 = 1 / np.sqrt(2)
    for u, (sf, cfi, dcis) in enumerate(units):
        n_symbs = cfi + (1 if nrb <= 10 else 0)
        g1 = pdcch_tx_grid(ref, fft, nrb, 1, cell, phich_res, sf, cfi, dcis)[0].reshape(-1)
        pc1, cand1 = m.pdcch_re_tables(nrb, 1, cell, phich_res, n_symbs)
        pcn, candn = m.pdcch_re_tables(nrb, n_ant, cell, phich_res, n_symbs)
        tx = np.zeros((n_ant, 16 * 1200), np.complex64)

        def place(pos, d):
            if n_ant == 2:
                x0, x1 = d[0::2], d[1::2]
                tx[0, pos[0::2]], tx[0, pos[1::2]] = r2 * x0, r2 * x1
                tx[1, pos[0::2]], tx[1, pos[1::2]] = -r2 * np.conj(x1), r2 * np.conj(x0)
            else:
                x0, x1, x2, x3 = d[0::4], d[1::4], d[2::4], d[3::4]
                tx[0, pos[0::4]], tx[0, pos[1::4]] = r2 * x0, r2 * x1
                tx[2, pos[0::4]], tx[2, pos[1::4]] = -r2 * np.conj(x1), r2 * np.conj(x0)
                tx[1, pos[2::4]], tx[1, pos[3::4]] = r2 * x2, r2 * x3
                tx[3, pos[2::4]], tx[3, pos[3::4]] = -r2 * np.conj(x3), r2 * np.conj(x2)

        place(pcn, g1[pc1])
        for c in range(len(dcis)):
            assert (cand1[c, :144] != 0xFFFFFFFF).all() and (candn[c, :144] != 0xFFFFFFFF).all()
            place(candn[c, :144], g1[cand1[c, :144]])
        rx = np.zeros(16 * 1200, np.complex64)
        sig = 10 ** (-snr_db / 20) / np.sqrt(2)
        for p in range(n_ant):
            a, tau, ph = rng.uniform(0.6, 1.4), rng.uniform(-2e-3, 2e-3), rng.uniform(-np.pi, np.pi)
            h = np.tile(a * np.exp(1j * (ph + 2 * np.pi * tau * k)), 16)
            rx += (h * tx[p]).astype(np.complex64)
            est = h + 0.2 * sig * (rng.standard_normal(h.shape) + 1j * rng.standard_normal(h.shape))
            grids[u, 2 + p], grids[u, 2 + n_ant + p] = est.real.reshape(16, 1200), est.imag.reshape(16, 1200)
        rx += (sig * (rng.standard_normal(rx.shape) + 1j * rng.standard_normal(rx.shape))).astype(np.complex64)
        grids[u, 0], grids[u, 1] = rx.real.reshape(16, 1200), rx.imag.reshape(16, 1200)
    return dict(fft=fft, nrb=nrb, n_ant=n_ant, cell=cell, phich_res=phich_res, sfs=[x[0] for x in units], units=units, grids=grids)


# ---------------------------------------------------------------------------------------------------
# PBCH (rest of SURVEY 8f N3): subframe 0 with the MIB of frame sfn, built with the reference's transmitter

PBCH_CASES = {
    # name: (fft, N_rb_dl, N_ant (transmitted), [(cell, sfn) per unit], snr_db)
    "20MHz_1ant": (2048, 100, 1, [(17, 0), (17, 1), (17, 2), (17, 3), (301, 6), (0, 1023)], 6.0),
    "1p4MHz_1ant": (128, 6, 1, [(301, 5), (150, 2), (503, 3)], 8.0),
    "5MHz_2ant": (512, 25, 2, [(44, 0), (45, 1), (46, 7)], 8.0),
    "10MHz_4ant": (1024, 50, 4, [(100, 0), (101, 2)], 10.0),
    "15MHz_noisy": (2048, 75, 1, [(7, s) for s in range(8)], -2.0),
}


def pbch_case(R, name, seed=31):
    """Units for one case: grids in the device-subframe layout with FOUR estimate planes (float32 [n, 10, 16, 1200]), rx = sum over the
    transmitted ports of h_p * tx_p + noise; ports that do not transmit get a noise-only estimate (what a CRS estimator sees).
    name: a key of PBCH_CASES, or such a tuple itself (the fuzz test draws them)."""
    from oracle import pyoracle as po
    fft, nrb, n_ant, units, snr_db = PBCH_CASES[name] if isinstance(name, str) else name
    rng = np.random.default_rng(seed)
    grids = np.zeros((len(units), 10, 16, 1200), np.float32)
    k = np.arange(1200)
    mibs = []
    sig = 10 ** (-snr_db / 20) / np.sqrt(2)
    for u, (cell, sfn) in enumerate(units):
        phy = R.ref_phy_new(po.FS_ENUM[fft], cell, n_ant, nrb)
        sfp = R.ref_subframe_new()
        R.ref_subframe_clear_tx(sfp, 0)
        mib = rng.integers(0, 2, 24).astype(np.uint8)
        mibs.append(mib)
        assert R.ref_bch_channel_encode(phy, sfp, mib, cell, n_ant, sfn) == 0
        tx_re = np.ctypeslib.as_array(R.ref_subframe_ptr(sfp, 4), shape=(4, 16, 1200))
        tx_im = np.ctypeslib.as_array(R.ref_subframe_ptr(sfp, 5), shape=(4, 16, 1200))
        rx = np.zeros((16, 1200), np.complex64)
        for p in range(4):
            if p < n_ant:
                a, tau, ph = rng.uniform(0.6, 1.4), rng.uniform(-2e-3, 2e-3), rng.uniform(-np.pi, np.pi)
                h = (a * np.exp(1j * (ph + 2 * np.pi * tau * k)))[None, :] * np.ones((16, 1))
                rx += (h * (tx_re[p] + 1j * tx_im[p])).astype(np.complex64)
            else:
                h = np.zeros((16, 1200), np.complex64)
            est = h + 0.2 * sig * (rng.standard_normal(h.shape) + 1j * rng.standard_normal(h.shape))
            grids[u, 2 + p], grids[u, 6 + p] = est.real, est.imag
        rx += (sig * (rng.standard_normal(rx.shape) + 1j * rng.standard_normal(rx.shape))).astype(np.complex64)
        grids[u, 0], grids[u, 1] = rx.real, rx.imag
        R.ref_subframe_free(sfp)
        R.ref_phy_free(phy)
    return dict(fft=fft, nrb=nrb, n_ant=n_ant, units=units, grids=grids, mibs=mibs)


def ref_pbch_decode(R, case):
    """liblte_phy_bch_channel_decode of the compiled reference over a case's grids -> uint32 [n, 4]: rc, N_ant, offset, mib (24 bits, MSB first;
    N_ant, offset, mib = 0 when rc != 0)."""
    import ctypes as C
    from oracle import pyoracle as po
    out = []
    sfp = R.ref_subframe_new()
    for u, (cell, sfn) in enumerate(case["units"]):
        phy = R.ref_phy_new(po.FS_ENUM[case["fft"]], cell, 4, case["nrb"])
        R.ref_subframe_set_num(sfp, 0)
        g = case["grids"][u]
        po.ref_subframe_view(R, sfp, 0)[:] = g[0]
        po.ref_subframe_view(R, sfp, 1)[:] = g[1]
        po.ref_subframe_view(R, sfp, 2, True)[:] = g[2:6]
        po.ref_subframe_view(R, sfp, 3, True)[:] = g[6:10]
        na, off, bits = C.c_uint32(), C.c_uint32(), np.zeros(32, np.uint8)
        rc = R.ref_bch_channel_decode(phy, sfp, cell, C.byref(na), bits, C.byref(off))
        mib = int("".join(str(int(b)) for b in bits[:24]), 2) if rc == 0 else 0
        out.append((rc, na.value if rc == 0 else 0, off.value if rc == 0 else 0, mib))
        R.ref_phy_free(phy)
    R.ref_subframe_free(sfp)
    return np.array(out, np.uint32)


# ---------------------------------------------------------------------------------------------------
# initial synchronisation: int8 captures written by shim/_build/capture_gen (the reference's TX API), impaired in numpy

SYNC_CASES = {
    # name: (fft, N_rb_dl, cell, frames, delay samples prepended, frequency offset Hz, snr_db)
    "1p4MHz_clean": (128, 6, 17, 18, 0, 0.0, 300.0),
    "1p4MHz_offset": (128, 6, 301, 18, 1234, 700.0, 12.0),
    "5MHz_noisy": (512, 25, 150, 18, 40000, -300.0, 6.0),
    "20MHz": (2048, 100, 77, 10, 100001, 150.0, 15.0),
}


def capture_gen_path():
    import os
    p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shim", "_build", "capture_gen")
    return p if os.path.exists(p) else None


def sync_case(name, tmp_dir, seed=41):
    """int8 interleaved I,Q capture for one case (numpy int8 [n, 2])."""
    import os
    import subprocess
    fft, nrb, cell, frames, delay, f_off, snr_db = SYNC_CASES[name] if isinstance(name, str) else name
    path = os.path.join(str(tmp_dir), "cap_%s.bin" % (name if isinstance(name, str) else "drawn"))
    subprocess.run([capture_gen_path(), path, str(nrb), str(cell), str(frames)], check=True, timeout=600, stdout=subprocess.DEVNULL)
    x = np.fromfile(path, np.int8).reshape(-1, 2).astype(np.float32)
    os.remove(path)
    rng = np.random.default_rng(seed)
    z = x[:, 0] + 1j * x[:, 1]
    rms = float(np.sqrt(np.mean(np.abs(z) ** 2)))
    z = np.concatenate([np.zeros(delay, np.complex64), z])
    fs = 30.72e6 * fft / 2048
    z = z * np.exp(2j * np.pi * f_off * np.arange(len(z)) / fs)
    if snr_db < 200:
        sig = rms * 10 ** (-snr_db / 20) / np.sqrt(2)
        z = z + sig * (rng.standard_normal(len(z)) + 1j * rng.standard_normal(len(z)))
    iq = np.stack([np.clip(np.round(z.real), -127, 127), np.clip(np.round(z.imag), -127, 127)], axis=1).astype(np.int8)
    return dict(fft=fft, nrb=nrb, cell=cell, iq=iq, delay=delay, f_off=f_off)


def ref_sync(R, case, n_slots=160):
    """The compiled reference's three searches over a capture (the scanner's sequence, first coarse peak that yields a cell):
    dict(coarse=(n_peaks, freq_offset[5], symb_starts[5,7]), per_peak=[(pss tuple, sss tuple or None) ...])."""
    import ctypes as C
    from oracle import pyoracle as po
    phy = R.ref_phy_new(po.FS_ENUM[case["fft"]], 0, 1, case["nrb"])
    i = np.ascontiguousarray(case["iq"][:, 0].astype(np.float32))
    q = np.ascontiguousarray(case["iq"][:, 1].astype(np.float32))
    n, fo, ss = C.c_uint32(), np.zeros(5, np.float32), np.zeros(35, np.uint32)
    assert R.ref_find_coarse_timing(phy, i, q, n_slots, C.byref(n), fo, ss) == 0
    ss = ss.reshape(5, 7)
    peaks = []
    for p in range(n.value):
        s = ss[p].copy()
        n2, ps, th, f = C.c_uint32(), C.c_uint32(), C.c_float(), C.c_float()
        assert R.ref_find_pss(phy, i, q, s, C.byref(n2), C.byref(ps), C.byref(th), C.byref(f)) == 0
        pss = (s.copy(), n2.value, ps.value, th.value, f.value)
        n1, fs = C.c_uint32(), C.c_uint32()
        rc = R.ref_find_sss(phy, i, q, n2.value, s, th.value, C.byref(n1), C.byref(fs))
        peaks.append((pss, (n1.value, fs.value, s.copy()) if rc == 0 else None))
    R.ref_phy_free(phy)
    return dict(coarse=(n.value, fo, ss), per_peak=peaks)


def multi_port_capture(R, n_ant, seed=70, fft=2048, nrb=100, cell=101, sf=4, cfi=2, mod=2, tbs=2024, prbs=None, noise=0.5, rv=0):
    """One subframe (+ the next one's CRS, which the estimator's interpolation reads) of a real n_ant-port cell from the reference's
    own transmitter: CRS on every port, one transmit-diversity PDSCH allocation, each antenna through its own complex gain, summed,
    scaled to int8.  Returns dict with iq int8 [2*30720*fft/2048, 2], the message bits, the reference allocation and its phy."""
    from oracle import pyoracle as po
    rng = np.random.default_rng(seed + n_ant)
    prbs = prbs or list(range(10, 22))
    phy = R.ref_phy_new(po.FS_ENUM[fft], cell, n_ant, nrb)
    sfp = R.ref_subframe_new()
    la = po.make_alloc(mod, tbs, prbs, 0x2345, rv, 1 if n_ant == 1 else 2, 0)  # LIBLTE_PHY_PRE_CODER_TYPE_TX_DIVERSITY
    msg = rng.integers(0, 2, tbs).astype(np.uint8)
    n_samp = 30720 * fft // 2048
    z = np.zeros(2 * n_samp, np.complex64)
    gains = [rng.uniform(0.6, 1.2) * np.exp(1j * rng.uniform(-np.pi, np.pi)) for _ in range(n_ant)]
    for k in range(2):
        R.ref_subframe_clear_tx(sfp, sf + k)
        assert R.ref_map_crs(phy, sfp, cell, n_ant) == 0
        if k == 0:
            arr = (po.LoAlloc * 1)(la)
            assert R.ref_pdsch_channel_encode(phy, sfp, arr, 1, msg, tbs, cfi, cell, n_ant) == 0
        for p in range(n_ant):
            i_s, q_s = np.zeros(n_samp, np.float32), np.zeros(n_samp, np.float32)
            assert R.ref_create_dl_subframe(phy, sfp, p, i_s, q_s) == 0
            z[k * n_samp:(k + 1) * n_samp] += gains[p] * (i_s + 1j * q_s)
    z *= 90.0 / np.abs(np.concatenate([z.real, z.imag])).max()
    z += noise * (rng.standard_normal(len(z)) + 1j * rng.standard_normal(len(z)))
    iq = np.stack([np.round(z.real), np.round(z.imag)], axis=1).astype(np.int8)
    R.ref_subframe_free(sfp)
    return dict(fft=fft, nrb=nrb, cell=cell, sf=sf, cfi=cfi, tbs=tbs, prbs=prbs, msg=msg, iq=iq, la=la, phy=phy)
