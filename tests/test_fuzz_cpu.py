"""CPU side of the randomized differential tests (tests/fuzz_cases.py): the three CPU checkers against each other.

* oracle/_ref/libref_oracle_big.so (the reference with only its PDSCH scratch literals enlarged, SURVEY 7.1 `oracle_big`) must
  equal the unmodified build wherever the unmodified build can run a case at all;
* the plain-C restatement (oracle/lte_oracle.c), which most GPU stage tests use as their checker, must equal the reference on
  hundreds of random cases -- large allocations included, which its round-2 pin never reached;
* the 4-port pre-decoder tail (liblte_phy.cc:7766-7795) is shown unreachable from liblte_phy_pdsch_channel_decode by enumeration.

Every test here is collected twice (lte_testdata.on_both_boxes): once unmarked for the CPU suite and once marked `gpu`, so that the
driver's `-m gpu` run proves the checkers on the box where the kernels they underwrite run."""
import ctypes as C

import numpy as np
import pytest

import fuzz_cases as fz
import lte_testdata as td


def rel_l2(a, b):
    return float(np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-30))


@pytest.fixture(scope="module")
def ref_big():
    from oracle import pyoracle
    L = pyoracle.ref_big()
    if L is None:
        pytest.skip("oracle/_ref/libref_oracle_big.so not built (needs /root/reference)")
    return L


@td.on_both_boxes
def test_enlarged_reference_equals_the_unmodified_one_inside_its_envelope(ref, ref_big, box):
    cases = [c for c in fz.draw_dl_cases(700, 31, big_share=0.0) if c["e"] <= 10000 and c["n_re"] <= 5000]
    assert len(cases) >= 400
    a = fz.run_ref_dl(ref_big, cases)
    b = fz.run_ref_dl(ref, cases, iq=a["iq"].copy())
    assert (a["rc"] == b["rc"]).all() and (a["n_soft"] == b["n_soft"]).all() and (a["n_out"] == b["n_out"]).all()
    assert (a["soft"] == b["soft"]).all() and (a["planes"] == b["planes"]).all()
    for i, c in enumerate(cases):
        if a["rc"][i] == 0:
            assert (a["bits"][i, :c["tbs"]] == b["bits"][i, :c["tbs"]]).all()
    assert (a["rc"] == 0).sum() >= 150


def port_case(port, c, planes, iq):
    """The restatement on one case: front end from the capture (tolerance stage) and the PDSCH decode on the REFERENCE's grid (exact)."""
    from oracle import pyoracle as po
    lc, s_own = td.oracle_frontend(port, c["fft"], c["n_rb"], c["n_ant"], iq[:, :], c["sf"], c["cell"])
    s = po.LoSubframe()
    s.num = c["sf"]
    n = c["n_ant"]
    s.arr("rx_symb_re")[:] = planes[0]
    s.arr("rx_symb_im")[:] = planes[1]
    s.arr("rx_ce_re")[:n] = planes[2:2 + n]
    s.arr("rx_ce_im")[:n] = planes[2 + n:2 + 2 * n]
    la = po.make_alloc(c["mod"], c["tbs"], c["prb0"], c["rnti"], c["rv"], c["tx_mode"])
    out, nb = np.zeros(6200, np.uint8), C.c_uint32()
    soft, ns = np.zeros(c["e"] + 64, np.int8), C.c_uint32()
    err = port.lo_pdsch_channel_decode(C.byref(lc), C.byref(s), C.byref(la), c["n_sym"], c["cell"], n, out, C.byref(nb),
                                       soft.ctypes.data_as(C.c_void_p), C.byref(ns))
    cw = np.zeros(ns.value, np.uint8)
    port.lo_prs_c((c["rnti"] << 14) | (c["sf"] << 9) | c["cell"], ns.value, cw)
    desc = (soft[:ns.value].astype(np.int16) * (1 - 2 * cw.astype(np.int16))).astype(np.int8)
    n_sc = 12 * c["n_rb"]
    tol = (rel_l2(s_own.arr("rx_symb_re")[:14, :n_sc], planes[0][:14, :n_sc]), rel_l2(s_own.arr("rx_ce_re")[:n, :14, :n_sc], planes[2:2 + n, :14, :n_sc]))
    return err, out[:nb.value].copy(), desc, tol


@td.on_both_boxes
def test_restatement_equals_the_reference_on_random_cases(port, ref_big, box):
    """~2 000 random downlink cases (one PRB list for both slots: the restatement's interface), every bandwidth, 1 / 2 / 4 ports,
    allocations up to the full band: soft bits, verdict and transport block identical; its own front end within the FFT tolerance."""
    cases = [c for c in fz.draw_dl_cases(2400, 77, big_share=0.08) if c["prb0"] == c["prb1"]]
    assert len(cases) >= 1800
    r = fz.run_ref_dl(ref_big, cases)
    res = td.parallel_map(lambda i: port_case(port, cases[i], r["planes"][i], r["iq"][i]), range(len(cases)))
    n_ok = n_big = 0
    for i, (c, (err, out, desc, tol)) in enumerate(zip(cases, res)):
        key = (i, c["n_rb"], c["n_ant"], c["cell"], c["sf"], c["n_sym"], c["mod"], c["tbs"], c["rv"], c["tx_mode"], len(c["prb0"]))
        assert len(desc) == r["n_soft"][i] and (desc == r["soft"][i, :len(desc)]).all(), key
        assert (err == 0) == (r["rc"][i] == 0), key + (err, int(r["rc"][i]))
        if err == 0:
            assert (out == r["bits"][i, :c["tbs"]]).all(), key
            n_ok += 1
        assert tol[0] < 1e-5 and tol[1] < 1e-4, key + tol
        n_big += c["e"] > 10000
    assert n_ok >= 700 and n_big >= 90


@td.on_both_boxes
def test_four_port_pre_decoder_tail_is_unreachable_from_the_pdsch_decode(box):
    """liblte_phy.cc:7766-7795 handles M_ap % 4 != 0 (and mis-strides the layer de-mapper when it does, :7473-7514).  The PDSCH RE
    extraction (:3744-3802) never produces such a count: every (slot, PRB) contributes a multiple of N_ant resource elements for
    every bandwidth, cell, subframe and control-region size -- enumerated here -- so the GPU kernel's `M_ap / N_ant` is exact."""
    for n_ant in (2, 4):
        for fs, fft, n_rb in fz.BANDWIDTHS:
            for sf in range(10):
                for cell in range(6):  # the masks depend on cell % 6 only
                    for n_sym in range(1, 5):
                        for prb in range(n_rb):
                            n0 = fz.re_count(n_rb, n_ant, cell, sf, n_sym, [prb], [])
                            n1 = fz.re_count(n_rb, n_ant, cell, sf, n_sym, [], [prb])
                            assert n0 % n_ant == 0 and n1 % n_ant == 0, (n_ant, n_rb, sf, cell, n_sym, prb, n0, n1)


@td.on_both_boxes
def test_re_count_closed_form_equals_the_reference_loop(ref_big, box):
    """fuzz_cases.re_count (used by the enumeration above and by the case generator) against the reference's own count of soft bits
    (dlsch_N_e_bits[0] after liblte_phy_pdsch_channel_decode), distributed allocations included."""
    cases = fz.draw_dl_cases(600, 5)
    r = fz.run_ref_dl(ref_big, cases, want_planes=False)
    for i, c in enumerate(cases):
        assert r["n_soft"][i] == c["e"], (i, c["n_rb"], c["n_ant"], c["cell"], c["sf"], c["n_sym"], c["mod"], c["prb0"], c["prb1"])


@td.on_both_boxes
def test_reference_on_a_float32_fft_against_itself_on_a_float64_one(box):
    """The second opinion on the FFT boundary (the reference links FFTW3f, which no box here has): the SAME captures through the
    reference built on the float64 DFT stand-in (the parity build every test compares with) and through the reference built on the
    single-precision Stockham stand-in (oracle/ref/fftw_shim_f32.c: the operation order FFTW3f would plausibly pick).  What moves between
    the two is the noise floor any float32 FFT -- FFTW3f's included -- sits in: soft bits that differ, verdicts that flip.  The library's
    own front end against the float64 build moves LESS than that (test_fuzz_gpu.py: 0 of 20 000 verdicts per run), which is what 'within the
    FFT's own rounding' means here.  The numbers go to gpurun_out/fft_noise_floor.json; the asserts only bound the floor itself."""
    import json
    import os
    import fuzz_cases as fz
    from oracle import pyoracle as po
    r64, r32 = po.ref(), po.ref_f32fft()
    if r64 is None or r32 is None:
        pytest.skip("oracle/_ref not built (needs /root/reference)")
    n = 4000 if fz.n_threads() >= 32 else 600
    cases = [c for c in fz.draw_dl_cases(2 * n, 424242, big_share=0.0) if c["e"] <= 10000][:n]  # (both builds have the unmodified 10 000-soft-bit scratch)
    a = fz.run_ref_dl(r64, cases, want_planes=True)
    b = fz.run_ref_dl(r32, cases, want_planes=True, iq=a["iq"])
    assert (a["rc_tx"] == 0).all() and (a["rc_fe"] == 0).all() and (b["rc_fe"] == 0).all()
    n_soft = int(a["n_soft"].sum())
    soft_diff = sum(int((a["soft"][i, :a["n_soft"][i]] != b["soft"][i, :a["n_soft"][i]]).sum()) for i in range(len(cases)))
    cases_soft = sum(bool((a["soft"][i, :a["n_soft"][i]] != b["soft"][i, :a["n_soft"][i]]).any()) for i in range(len(cases)))
    verdicts = int((a["rc"] != b["rc"]).sum())
    bits = sum(bool(a["rc"][i] == 0 and b["rc"][i] == 0 and (a["bits"][i, :a["n_out"][i]] != b["bits"][i, :a["n_out"][i]]).any()) for i in range(len(cases)))
    rel = [float(np.linalg.norm((a["planes"][i, :2] - b["planes"][i, :2]).ravel()) / max(np.linalg.norm(a["planes"][i, :2].ravel()), 1e-30)) for i in range(len(cases))]
    out = dict(cases=len(cases), soft_bits=n_soft, soft_bits_differing=soft_diff, cases_with_differing_soft_bits=int(cases_soft), verdicts_differing=verdicts,
               decoded_by_both_with_different_bits=int(bits), worst_rel_l2_rx_symb_f32_vs_f64=max(rel),
               note="reference on fftw_shim_f32.c vs reference on fftw_shim.c (float64), same int8 captures; QPSK soft values and 16/64QAM decisions next to a threshold are what moves")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(root, "gpurun_out", "fft_noise_floor.json"), "w"), indent=1)
    assert max(rel) < 1e-5                       # the two transforms agree as transforms
    assert verdicts <= max(2, len(cases) // 500)  # and almost always on the verdict
    assert bits == 0                              # a block both decode is the same block
