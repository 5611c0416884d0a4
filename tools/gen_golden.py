#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref, built in place from
/root/reference by oracle/ref/Makefile).  Run in the build container; the small .npz fixtures are
committed so that the GPU box (which has no /root/reference) can check against them.

    python tools/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import pyoracle as po  # noqa: E402
import lte_testdata as td  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def pack(bits):
    return np.packbits(bits.astype(np.uint8), axis=-1)


def turbo_ref_vectors(R, P, phy):
    """REF turbo decode: seeded inputs + the reference's hard bits (packed)."""
    rec = {}
    cases = [(40, "awgn0.8", 8), (104, "int", 8), (1088, "awgn0.5", 6), (3264, "hard127", 4), (3584, "awgn0.5", 4),
             (6016, "awgn0.8", 3), (6144, "awgn0.5", 3), (6144, "hard127", 3), (2048, "i16", 4)]
    for K, kind, n in cases:
        seed = 9000 + K + len(kind)
        tx, soft = td.turbo_blocks(P, K, n, kind, seed)
        out = np.zeros((n, K), np.uint8)
        for b in range(n):
            R.ref_turbo_decode(phy, np.ascontiguousarray(soft[b], dtype=np.float32), 3 * (K + 4), out[b])
        key = "K%d_%s" % (K, kind)
        rec[key + "_seed"] = np.array([seed, n])
        rec[key + "_soft"] = soft  # inputs are stored too, so the fixture does not depend on numpy's RNG stream
        rec[key + "_bits"] = pack(out)
    np.savez_compressed(os.path.join(OUT, "turbo_ref.npz"), **rec)
    print("turbo_ref.npz:", len(cases), "cases")


def uplink_vectors(R):
    """Uplink receive chain: seeded int8 captures from the library's host transmitter + what the
    reference's liblte_phy_get_ul_subframe / liblte_phy_pusch_channel_decode make of them."""
    rec = {}
    for name in ("1p4MHz_hop", "5MHz_seqhop_16qam", "10MHz_prime"):
        case = td.ul_case(name)
        symb, res = td.ref_ul_decode(R, case)
        rec[name + "_iq"] = case["iq"]
        n_sc = 12 * case["nrb"]
        rec[name + "_symb"] = symb[:, :, :, :n_sc].astype(np.float32)
        for i, (rc, bits, g) in enumerate(res):
            rec["%s_a%d_rc" % (name, i)] = np.array([rc])
            rec["%s_a%d_g" % (name, i)] = g
            if bits is not None:
                rec["%s_a%d_bits" % (name, i)] = pack(bits)
    np.savez_compressed(os.path.join(OUT, "uplink_ref.npz"), **rec)
    print("uplink_ref.npz:", sorted({k.split("_a")[0].rsplit("_", 1)[0] for k in rec}))


def prach_vectors(R):
    """PRACH detection: seeded occasions from the library's host transmitter + the reference's three outputs per occasion."""
    rec = {}
    for name in td.PRACH_CASES:
        case = td.prach_case(name)
        want, _ = td.ref_prach_detect(R, case)
        rec[name + "_iq"] = case["iq"]
        rec[name + "_det"] = want
    np.savez_compressed(os.path.join(OUT, "prach_ref.npz"), **rec)
    print("prach_ref.npz:", {k: rec[k + "_det"].tolist() for k in td.PRACH_CASES})


def pdcch_vectors(R):
    """PCFICH/PDCCH: received control-region grids (symbols 0-3 only) built by tests/lte_testdata.pdcch_case with the reference's
    transmitter, and what liblte_phy_pdcch_channel_decode returned for them."""
    rec, names = {}, []
    for name in td.PDCCH_CASES:
        case = td.pdcch_case(R, name)
        keep = min(len(case["sfs"]), 3)
        case["sfs"], case["grids"] = case["sfs"][:keep], case["grids"][:keep]
        want = td.ref_pdcch_decode(R, case)
        names.append(name)
        rec[name + "_cfg"] = np.array([case["fft"], case["nrb"], case["n_ant"], case["cell"]], np.uint32)
        rec[name + "_phich_res"] = np.float32(case["phich_res"])
        rec[name + "_sfs"] = np.array(case["sfs"], np.uint32)
        rec[name + "_grids"] = case["grids"][:, :, :4, :].astype(np.float32)
        rec[name + "_rc"] = np.array([w[0] for w in want], np.uint32)
        rec[name + "_cfi"] = np.array([w[1] for w in want], np.uint32)
        rec[name + "_nsym"] = np.array([w[2] for w in want], np.uint32)
        # per DCI: unit, rnti, mcs, tbs, rv, N_prb, tx_mode, mod_type, first PRB of slot 0, last PRB of slot 1
        rows = [[u] + list(r[:7]) + [r[7][0] if r[7] else 0, r[8][-1] if r[8] else 0] for u, w in enumerate(want) for r in w[3]]
        rec[name + "_dci"] = np.array(rows, np.int64).reshape(len(rows), 10)
    rec["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "pdcch_ref.npz"), **rec)
    print("pdcch_ref.npz:", {n: rec[n + "_dci"].shape[0] for n in names})


def pbch_vectors(R):
    """PBCH: received grids (symbols 7-10 only, four estimate planes) built by tests/lte_testdata.pbch_case with the reference's
    transmitter, and what liblte_phy_bch_channel_decode returned for them (rc, N_ant, offset, MIB)."""
    rec, names = {}, []
    for name in td.PBCH_CASES:
        case = td.pbch_case(R, name)
        keep = min(len(case["units"]), 4)
        case["units"], case["grids"] = case["units"][:keep], case["grids"][:keep]
        names.append(name)
        rec[name + "_cfg"] = np.array([case["fft"], case["nrb"]], np.uint32)
        rec[name + "_cells"] = np.array([c for c, _ in case["units"]], np.uint32)
        rec[name + "_grids"] = case["grids"][:, :, 7:11, :].astype(np.float32)
        rec[name + "_want"] = td.ref_pbch_decode(R, case)
    rec["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "pbch_ref.npz"), **rec)
    print("pbch_ref.npz:", {n: rec[n + "_want"].tolist() for n in names})


def sync_vectors(R):
    """Initial synchronisation: one impaired 1.4 MHz capture (int8) and the outputs of the reference's coarse-timing, PSS and SSS
    searches over it, per coarse peak."""
    import tempfile
    if td.capture_gen_path() is None:
        print("sync_ref.npz: skipped (shim/_build/capture_gen not built)")
        return
    with tempfile.TemporaryDirectory() as tmp:
        case = td.sync_case("1p4MHz_offset", tmp)
    want = td.ref_sync(R, case)
    n, fo, ss = want["coarse"]
    pss = np.array([[p[1], p[2]] + p[0].tolist() for p, _ in want["per_peak"]], np.uint32)
    np.savez_compressed(os.path.join(OUT, "sync_ref.npz"), cfg=np.array([case["fft"], case["nrb"]], np.uint32), iq=case["iq"], n_peaks=np.uint32(n),
                        freq_offset=fo, symb_starts=ss, pss=pss, pss_thresh=np.array([p[3] for p, _ in want["per_peak"]], np.float32),
                        sss=np.array([[1, s[0], s[1]] if s is not None else [0, 0, 0] for _, s in want["per_peak"]], np.uint32))
    print("sync_ref.npz:", n, "peaks;", [(3 * s[0] + p[1], s[1]) for p, s in want["per_peak"] if s is not None])


def two_port_units(R):
    """Four 20 MHz subframe units of a TWO-port cell from the reference's own transmitter (CRS on both ports, one transmit-diversity
    PDSCH allocation, each antenna through its own complex gain, noise, int8): the input of bench.py's two-port front-end leg (BASELINE
    config 2), plus, for unit 0, what liblte_phy_get_dl_subframe_and_ce makes of it (symbol rows and both ports' estimate rows)."""
    import ctypes as C
    from openlte_amd import synth
    ul = synth.unit_len(2048)
    units, sfs, cells, want = [], [], [], None
    for k, (cell, sf) in enumerate(((101, 4), (7, 1), (333, 8), (450, 6))):
        cap = td.multi_port_capture(R, 2, seed=200 + k, cell=cell, sf=sf, noise=0.7)
        iq = cap["iq"][:ul]
        units.append(iq); sfs.append(sf); cells.append(cell)
        if k == 0:
            i_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[:, 0].astype(np.float32)]))
            q_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[:, 1].astype(np.float32)]))
            rx = R.ref_subframe_new()
            assert R.ref_get_dl_subframe_and_ce(cap["phy"], i_f, q_f, 0, sf, cell, 2, rx) == 0
            want = [po.ref_subframe_view(R, rx, 0)[:14].copy(), po.ref_subframe_view(R, rx, 1)[:14].copy(),
                    po.ref_subframe_view(R, rx, 2, True)[:2, :14].copy(), po.ref_subframe_view(R, rx, 3, True)[:2, :14].copy()]
            R.ref_subframe_free(rx)
        R.ref_phy_free(cap["phy"])
    np.savez_compressed(os.path.join(OUT, "dl_two_port_units.npz"), iq=np.stack(units), sfs=np.array(sfs, np.uint32), cells=np.array(cells, np.uint32),
                        symb_re=want[0], symb_im=want[1], ce_re=want[2], ce_im=want[3])
    print("dl_two_port_units.npz:", len(units), "units of", ul, "samples")


def main():
    os.makedirs(OUT, exist_ok=True)
    R, P = po.ref(), po.port()
    assert R is not None, "oracle/_ref is not built (needs /root/reference)"
    if len(sys.argv) > 1 and sys.argv[1] == "two_port":  # (the other fixtures are left as they are)
        return two_port_units(R)
    if len(sys.argv) > 1 and sys.argv[1] == "prach":
        return prach_vectors(R)
    phy = R.ref_phy_new(4, 17, 1, 100)
    turbo_ref_vectors(R, P, phy)
    R.ref_phy_free(phy)
    uplink_vectors(R)
    prach_vectors(R)
    pdcch_vectors(R)
    pbch_vectors(R)
    sync_vectors(R)
    two_port_units(R)


if __name__ == "__main__":
    main()
