"""GPU: the drop-in claim, end to end.

shim/_build/dropin_gpu is a caller written purely against the reference's liblte_phy API, linked
against the reference's own objects EXCEPT the three hot-path entry points, which come from
shim/liblte_phy_shim.cc -> libmi_lte.so (built by shim/Makefile where /root/reference exists; the
binary travels with the repo snapshot).  Its output must equal what the same caller prints when
linked against the unmodified reference (tests/golden/dropin_demo_reference_cpu.txt)."""
import os
import re
import subprocess

import numpy as np
import pytest

import lte_testdata as td

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PURE_SYMBOLS = 35  # liblte_phy_* functions the build without any reference PHY object defines: all 34 of the reference's + the one-call uplink entry


def test_dropin_demo_matches_reference_output():
    exe = os.path.join(ROOT, "shim", "_build", "dropin_gpu")
    if not os.path.exists(exe):
        pytest.skip("shim/_build/dropin_gpu not built (needs the reference tree at build time)")
    got = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    want = open(os.path.join(ROOT, "tests", "golden", "dropin_demo_reference_cpu.txt")).read().strip().splitlines()
    lines = got.stdout.strip().splitlines()
    assert got.returncode == 0, got.stdout + got.stderr
    assert lines[1:] == want[1:], (lines, want)  # per-allocation verdicts: identical text
    h_got, h_want = float(re.findall(r"[-0-9.]+$", lines[0])[0]), float(re.findall(r"[-0-9.]+$", want[0])[0])
    assert abs(h_got - h_want) / h_want < 1e-4  # channel-estimate power: float tolerance


def test_host_rate_unmatch_bit_exact(ctx, port):
    """mi_lte_rate_unmatch_turbo_host == liblte_phy_rate_unmatch_turbo (oracle), incl. the 10000.0f sentinels."""
    import ctypes as C
    L = ctx.L
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    L.mi_lte_rate_unmatch_turbo_host.argtypes = [C.c_void_p, f32p] + [C.c_uint32] * 8 + [f32p, C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(8)
    for K in (40, 104, 1088, 3264, 6144):
        D = K + 4
        for (rv, C_, M, txm, chan, ratio) in ((0, 1, 4, 2, 0, 3.0), (1, 1, 8, 1, 0, 2.1), (2, 2, 8, 3, 0, 4.7), (3, 1, 8, 1, 0, 1.6),
                                             (0, 1, 1, 1, 2, 3.0)):
            E = int(ratio * D) // 2 * 2
            e = rng.integers(-127, 128, E).astype(np.float32)
            n_soft = 250368 if chan == 0 else 1
            want, got = np.zeros(3 * D, np.float32), np.zeros(3 * D, np.float32)
            port.lo_rate_unmatch_turbo(e.copy(), E, D, C_, txm, n_soft, M, chan, rv, want)
            n = C.c_uint32()
            rc = L.mi_lte_rate_unmatch_turbo_host(ctx.h, e, E, D, C_, txm, n_soft, M, chan, rv, got, C.byref(n))
            assert rc == 0 and n.value == 3 * D
            assert (got == want).all(), (K, rv, C_, M, txm, chan, ratio)


def test_host_front_end_and_pdsch(ctx, port):
    """The two per-call host-pointer entry points against the oracle on one W4 subframe."""
    import ctypes as C
    import openlte_amd as m
    from openlte_amd import synth
    L = ctx.L
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L.mi_lte_get_dl_subframe_and_ce_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, f32p, f32p] + [C.c_uint32] * 4 + [f32p] * 4
    L.mi_lte_pdsch_channel_decode_host.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, f32p, f32p, C.c_uint32, C.c_void_p,
                                                   C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.POINTER(C.c_uint32)]
    cfg = m.DlCfg(2048, 100, 1, 0)
    sf, cell = 4, 211
    allocs = td.w4_allocs(0)
    iq, tx = synth.dl_units(cfg, [sf], [cell], allocs, 9, snr_db=28, seed=12)
    i_s = np.concatenate([np.zeros(sf * 30720, np.float32), iq[0, :, 0].astype(np.float32)])
    q_s = np.concatenate([np.zeros(sf * 30720, np.float32), iq[0, :, 1].astype(np.float32)])
    sr, si = np.zeros((16, 1200), np.float32), np.zeros((16, 1200), np.float32)
    cr, ci = np.zeros((4, 16, 1200), np.float32), np.zeros((4, 16, 1200), np.float32)
    assert 0 == L.mi_lte_get_dl_subframe_and_ce_host(ctx.h, 2048, 100, i_s, q_s, 0, sf, cell, 1, sr, si, cr, ci)
    _, s = td.oracle_frontend(port, 2048, 100, 1, iq[0], sf, cell)
    assert np.linalg.norm(sr - s.arr("rx_symb_re")) / np.linalg.norm(s.arr("rx_symb_re")) < 1e-5
    assert np.linalg.norm(cr[0, :14] - s.arr("rx_ce_re")[0, :14]) / np.linalg.norm(s.arr("rx_ce_re")[0, :14]) < 1e-4
    for a in range(9):
        out, n = np.zeros(6200, np.uint8), C.c_uint32()
        rc = L.mi_lte_pdsch_channel_decode_host(ctx.h, 100, sr, si, cr, ci, sf, C.addressof(allocs[a]), 2, cell, 1, out, C.byref(n))
        assert rc == 0 and n.value == allocs[a].tbs
        assert (out[:n.value] == tx[0, a, :n.value]).all()


def test_host_forms_keep_the_subframe_on_the_device(port):
    """The per-call forms allocate nothing per call and reuse the device copy of a subframe this context produced: after
    get_dl_subframe_and_ce the PDCCH-less decode calls of the same host arrays upload nothing; a subframe the context has not seen (or
    one whose arrays changed) is uploaded once; plans are cached by allocation.  Results stay the oracle's throughout."""
    import ctypes as C
    import openlte_amd as m
    from openlte_amd import synth
    ctx = m.Context(0)  # a context of its own: the counters start at zero
    L = ctx.L
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L.mi_lte_get_dl_subframe_and_ce_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, f32p, f32p] + [C.c_uint32] * 4 + [f32p] * 4
    L.mi_lte_pdsch_channel_decode_host.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, f32p, f32p, C.c_uint32, C.c_void_p,
                                                   C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.POINTER(C.c_uint32)]
    L.mi_lte_host_cache_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]

    def stats():
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        assert L.mi_lte_host_cache_stats(ctx.h, C.byref(a), C.byref(b), C.byref(c)) == 0
        return a.value, b.value, c.value

    cfg = m.DlCfg(2048, 100, 1, 0)
    sfs, cells = [4, 7], [211, 38]
    allocs = td.w4_allocs(0) + td.w4_allocs(1)
    iq, tx = synth.dl_units(cfg, sfs, cells, allocs, 9, snr_db=28, seed=12)
    subs = []
    for u in range(2):
        i_s = np.concatenate([np.zeros(sfs[u] * 30720, np.float32), iq[u, :, 0].astype(np.float32)])
        q_s = np.concatenate([np.zeros(sfs[u] * 30720, np.float32), iq[u, :, 1].astype(np.float32)])
        sr, si = np.zeros((16, 1200), np.float32), np.zeros((16, 1200), np.float32)
        cr, ci = np.full((4, 16, 1200), 7.0, np.float32), np.full((4, 16, 1200), 7.0, np.float32)
        assert 0 == L.mi_lte_get_dl_subframe_and_ce_host(ctx.h, 2048, 100, i_s, q_s, 0, sfs[u], cells[u], 1, sr, si, cr, ci)
        assert (cr[0, 14:] == 7.0).all() and (cr[1:] == 7.0).all()  # estimate rows 14, 15 and the other ports: never written, as in the reference
        subs.append((sr, si, cr, ci))

    def decode(u, a):
        out, n = np.zeros(6200, np.uint8), C.c_uint32()
        sr, si, cr, ci = subs[u]
        rc = L.mi_lte_pdsch_channel_decode_host(ctx.h, 100, sr, si, cr, ci, sfs[u], C.addressof(allocs[9 * u + a]), 2, cells[u], 1, out, C.byref(n))
        assert rc == 0 and n.value == allocs[9 * u + a].tbs and (out[:n.value] == tx[u, a, :n.value]).all(), (u, a, rc)

    assert stats()[:2] == (0, 0)
    for a in range(9):  # the subframe produced last is still on the device
        decode(1, a)
    assert stats() == (9, 0, 9)  # nine reuses, no upload, nine plans
    decode(0, 0)  # the other subframe: one upload ...
    for a in range(1, 9):
        decode(0, a)  # ... then reuse; the plans are the cached ones (same allocations)
    assert stats() == (17, 1, 9)
    subs[0][0][3, 100:110] += 1.0  # the caller edits a row the fingerprint may or may not sample: it says so
    assert L.mi_lte_host_cache_invalidate(ctx.h) == 0
    out, n = np.zeros(6200, np.uint8), C.c_uint32()
    L.mi_lte_pdsch_channel_decode_host(ctx.h, 100, *subs[0], sfs[0], C.addressof(allocs[0]), 2, cells[0], 1, out, C.byref(n))
    assert stats()[1] == 2
    ctx.close()


def test_host_forms_on_a_narrow_carrier_touch_only_its_columns(port):
    """5 MHz (25 RB, 300 sub-carriers of the struct's 1200-wide rows): get_dl_subframe_and_ce writes columns 0..299 of each row and
    nothing else, as the reference does (samples_to_symbols_dl, liblte_phy.cc:8628-8632) -- the subframe comes back as packed columns
    written by the device into pinned host memory -- and the fingerprint that decides whether a decode call must upload the struct covers
    the same columns: an edit outside them is not a change, an edit inside them is found without being announced."""
    import ctypes as C
    import openlte_amd as m
    from openlte_amd import synth
    ctx = m.Context(0)
    L = ctx.L
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L.mi_lte_get_dl_subframe_and_ce_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, f32p, f32p] + [C.c_uint32] * 4 + [f32p] * 4
    L.mi_lte_pdsch_channel_decode_host.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, f32p, f32p, C.c_uint32, C.c_void_p,
                                                   C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.POINTER(C.c_uint32)]
    L.mi_lte_host_cache_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]

    def uploads():
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint32()
        assert L.mi_lte_host_cache_stats(ctx.h, C.byref(a), C.byref(b), C.byref(c)) == 0
        return b.value

    fft, nrb, sf, cell = 512, 25, 3, 77
    cfg = m.DlCfg(fft, nrb, 1, 0)
    allocs = [m.make_alloc(0, 1, 224, list(range(0, 6)), 0x500), m.make_alloc(0, 3, 776, list(range(19, 25)), 0x501)]
    iq, tx = synth.dl_units(cfg, [sf], [cell], allocs, 2, snr_db=25.0, seed=5)
    per_sf = 30720 // (2048 // fft)
    i_s = np.concatenate([np.zeros(sf * per_sf, np.float32), iq[0, :, 0].astype(np.float32)])
    q_s = np.concatenate([np.zeros(sf * per_sf, np.float32), iq[0, :, 1].astype(np.float32)])
    sr, si = np.full((16, 1200), 7.0, np.float32), np.full((16, 1200), 7.0, np.float32)
    cr, ci = np.full((4, 16, 1200), 7.0, np.float32), np.full((4, 16, 1200), 7.0, np.float32)
    assert 0 == L.mi_lte_get_dl_subframe_and_ce_host(ctx.h, fft, nrb, i_s, q_s, 0, sf, cell, 1, sr, si, cr, ci)
    n_sc = 12 * nrb
    for arr in (sr, si):
        assert (arr[:, n_sc:] == 7.0).all() and (arr[:, :n_sc] != 7.0).all()
    assert (cr[0, :14, n_sc:] == 7.0).all() and (cr[0, 14:] == 7.0).all() and (cr[1:] == 7.0).all() and (ci[0, :14, n_sc:] == 7.0).all()
    _, s = td.oracle_frontend(port, fft, nrb, 1, iq[0], sf, cell)
    ref_re, ref_ce = s.arr("rx_symb_re")[:, :n_sc], s.arr("rx_ce_re")[0, :14, :n_sc]
    assert np.linalg.norm(sr[:, :n_sc] - ref_re) / np.linalg.norm(ref_re) < 1e-5
    assert np.linalg.norm(cr[0, :14, :n_sc] - ref_ce) / np.linalg.norm(ref_ce) < 1e-4

    def decode(a):
        out, n = np.zeros(6200, np.uint8), C.c_uint32()
        rc = L.mi_lte_pdsch_channel_decode_host(ctx.h, nrb, sr, si, cr, ci, sf, C.addressof(allocs[a]), 2, cell, 1, out, C.byref(n))
        return rc, out[:n.value].copy()

    for a in range(2):
        rc, out = decode(a)
        assert rc == 0 and (out == tx[0, a, :allocs[a].tbs]).all(), a
    assert uploads() == 0
    sr[5, n_sc + 10] = -3.0  # outside the carrier: nobody reads it, so it is not a change
    assert decode(0)[0] == 0 and uploads() == 0
    keep = sr[5, 20:40].copy()
    sr[5, 20:40] = 0.0  # inside: the next call sees it (one upload) and decodes what the arrays now hold
    rc, out = decode(0)
    assert uploads() == 1
    sr[5, 20:40] = keep
    rc, out = decode(0)
    assert uploads() == 2 and rc == 0 and (out == tx[0, 0, :allocs[0].tbs]).all()
    ctx.close()


def _ul_demo_args(tmp_path):
    """One 20 MHz uplink subframe with three UEs, written as an int8 capture + the demo's command line."""
    case = td.ul_case("20MHz_3ue")
    path = os.path.join(str(tmp_path), "ul_capture.bin")
    case["iq"][0].tofile(path)
    args = [path, "100", str(case["cell"]), str(case["sfs"][0])] + [str(v) for v in case["ulc"]]
    for a in range(case["n_alloc"]):
        al = case["allocs"][a]
        args += [str(al.mod_type), str(al.tbs), str(al.rnti), str(al.prb[0][0]), str(al.N_prb)]
    return args


def _ul_demo_env(tmp_path):
    """One 20 MHz PRACH occasion (preamble 37 of root index 5, 80 samples late) for the demo's liblte_phy_detect_prach call."""
    import openlte_amd as m
    from openlte_amd import synth
    cfg, pc = m.DlCfg(2048, 100, 1, 0), m.PrachCfg(5, 0, 12, 0, 3)
    iq = synth.prach_occasions(cfg, pc, [37], [80], snr_db=5.0, seed=8)
    path = os.path.join(str(tmp_path), "prach.bin")
    iq[0].tofile(path)
    return dict(os.environ, PRACH_CAPTURE=path, PRACH_CFG="5,0,12,0,3", PUCCH_DEMO="1")


def test_uplink_dropin_demo_matches_reference_output(tmp_path):
    """liblte_phy_get_ul_subframe + liblte_phy_pusch_channel_decode + liblte_phy_detect_prach + PUCCH 1/1a/1b through the shim == the
    unmodified reference."""
    exe = os.path.join(ROOT, "shim", "_build", "dropin_ul_gpu")
    if not os.path.exists(exe):
        pytest.skip("shim/_build/dropin_ul_gpu not built (needs the reference tree at build time)")
    args, env = _ul_demo_args(tmp_path), _ul_demo_env(tmp_path)
    got = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
    assert got.returncode == 0, got.stdout + got.stderr
    lines = got.stdout.strip().splitlines()
    want = open(os.path.join(ROOT, "tests", "golden", "dropin_ul_demo_reference_cpu.txt")).read().strip().splitlines()
    cpu = os.path.join(ROOT, "shim", "_build", "dropin_ul_cpu")
    if os.path.exists(cpu):  # the reference-only build travelled too: it must still print the committed text
        ref_now = subprocess.run([cpu] + args, capture_output=True, text=True, timeout=600, env=env).stdout.strip().splitlines()
        assert ref_now == want
    assert lines[1:] == want[1:], (lines, want)  # per-UE verdict, bit count and hash of the decoded bits: identical text
    e_got, e_want = float(lines[0].split("=")[1]), float(want[0].split("=")[1])
    assert abs(e_got - e_want) / e_want < 1e-4


def test_uplink_dropin_demo_with_no_reference_phy_object_in_the_link(tmp_path):
    """dropin_ul_gpu_pure = the same uplink caller linked with NO object of the reference's PHY: liblte_phy_init / _ul_init / _cleanup are the
    shim's own (-DMI_LTE_SHIM_OWN_LIFECYCLE), the PUSCH reference signals and the PUCCH sequence tables come from the library's generators
    (mi_lte_ul_dmrs_pusch, mi_lte_ul_pucch_tables, asked with what liblte_phy_ul_init was given) and the PRACH detector builds the cell's
    root set itself.  Per-UE verdicts, bit counts and hashes of the decoded bits and the PRACH detection must be the all-reference build's.
    (The demo's own PUCCH section BUILDS its resources from the reference's private tables in the struct, which this build leaves empty:
    the PUCCH decoder of this build is checked by `lifecycle_check pucch` below.)"""
    exe = os.path.join(ROOT, "shim", "_build", "dropin_ul_gpu_pure")
    if not os.path.exists(exe):
        pytest.skip("shim/_build/dropin_ul_gpu_pure not built (needs the reference tree at build time)")
    args, env = _ul_demo_args(tmp_path), _ul_demo_env(tmp_path)
    env.pop("PUCCH_DEMO")
    got = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
    assert got.returncode == 0, got.stdout + got.stderr
    lines = got.stdout.strip().splitlines()
    want = open(os.path.join(ROOT, "tests", "golden", "dropin_ul_demo_reference_cpu.txt")).read().strip().splitlines()[:5]
    assert lines[1:5] == want[1:], (lines, want)
    assert abs(float(lines[0].split("=")[1]) - float(want[0].split("=")[1])) / float(want[0].split("=")[1]) < 1e-4
    syms = subprocess.run(["nm", "-C", "--defined-only", exe], capture_output=True, text=True).stdout
    assert " T liblte_phy_ul_init" in syms and "generate_dmrs_pusch" not in syms and "prach_preamble_seq_gen" not in syms and "fftwf_" not in syms


@pytest.mark.parametrize("exe_name", ["dropin_ul_gpu", "dropin_ul_gpu_pure"])
def test_uplink_one_call_per_subframe_prints_the_per_call_lines(tmp_path, exe_name):
    """The shim's liblte_phy_ul_subframe_decode (shim/liblte_phy_ext.h: get_ul_subframe + every UE's PUSCH decode as one launch chain with one
    wait, mi_lte_ul_subframe_decode_host) inside the same uplink caller: its per-UE lines -- verdict, bit count, hash of the decoded bits --
    are the per-call sequence's, which are the all-reference build's (the tests above)."""
    exe = os.path.join(ROOT, "shim", "_build", exe_name)
    if not os.path.exists(exe):
        pytest.skip("shim/_build/%s not built (needs the reference tree at build time)" % exe_name)
    args, env = _ul_demo_args(tmp_path), _ul_demo_env(tmp_path)
    env.pop("PUCCH_DEMO", None)
    env["UL_DEMO_ONE_CALL"] = "1"
    got = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
    assert got.returncode == 0, got.stdout + got.stderr
    lines = got.stdout.strip().splitlines()
    per_call = [l for l in lines if l.startswith("rnti ")]
    one_call = [l[len("one call: "):] for l in lines if l.startswith("one call: ")]
    assert len(per_call) == 3 and one_call == per_call, lines
    assert all("err=0" in l for l in per_call)


@pytest.mark.parametrize("exe_name", ["dropin_ul_gpu", "dropin_ul_gpu_pure"])
def test_enodeb_tti_receive_and_transmit_prints_the_references_lines(tmp_path, exe_name):
    """Both halves of an eNodeB's TTI in one caller (LTE_fdd_enb_phy.cc:557-770 and :832-917): the uplink subframe received (get_ul_subframe, a PUSCH
    decode per UE, the PRACH occasion) and the downlink subframe of the same TTI built and modulated (map_pss / _sss / _crs, the MIB, PCFICH + PHICH +
    a downlink assignment + an uplink grant sized by the TBS searches, get_n_cce, the SR configuration, PDSCH, create_dl_subframe).  That is every
    liblte_phy function LTE_fdd_enodeb calls.  The build with the receive chains on the GPU, and the build with NO object of the reference's PHY in
    its link (transmit side and helpers from libmi_lte.so's host code), print what the all-reference build prints -- incl. the hash of the samples."""
    build = os.path.join(ROOT, "shim", "_build")
    exe, cpu = os.path.join(build, exe_name), os.path.join(build, "dropin_ul_cpu")
    if not (os.path.exists(exe) and os.path.exists(cpu)):
        pytest.skip("shim/_build/%s / dropin_ul_cpu not built (need the reference tree at build time)" % exe_name)
    for sf in (0, 4, 5):
        args, env = _ul_demo_args(tmp_path), _ul_demo_env(tmp_path)
        args[3] = str(sf)  # (the capture was made for another subframe number: the PUSCH verdicts are whatever the reference's are, and the same)
        env.pop("PUCCH_DEMO", None)
        env["ENB_DL_TX"] = "1"
        want = subprocess.run([cpu] + args, capture_output=True, text=True, timeout=600, env=env)
        got = subprocess.run([exe] + args, capture_output=True, text=True, timeout=300, env=env)
        assert got.returncode == 0 and want.returncode == 0, got.stdout + got.stderr
        w, g = want.stdout.strip().splitlines(), got.stdout.strip().splitlines()
        assert len([l for l in w if l.startswith("dl tx frame")]) == 2 and "encode 0 0 0" in w[-1]
        assert g[1:] == w[1:], (g, w)  # (line 0 is the grid energy: a float sum, compared to rounding)
        assert abs(float(g[0].split("=")[1]) - float(w[0].split("=")[1])) <= 1e-4 * abs(float(w[0].split("=")[1]))


def test_pucch_decoder_of_the_build_without_the_reference_phy():
    """`shim/_build/lifecycle_check pucch` (a TEST binary that links the reference's PHY under other names): 360 PUCCH format 1 / 1a / 1b
    resources built from the tables the REFERENCE's liblte_phy_ul_init computed, decoded by the reference on its struct and by the
    own-lifecycle shim on a struct of its own, whose decoder asks mi_lte_ul_pucch_tables instead of reading tables out of the struct.
    Error code, bit count and bits equal for every resource (three cell configurations, mixed and unmixed resource blocks)."""
    exe = os.path.join(ROOT, "shim", "_build", "lifecycle_check")
    if not os.path.exists(exe):
        pytest.skip("shim/_build/lifecycle_check not built (needs the reference tree at build time)")
    r = subprocess.run([exe, "pucch"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "decodes equal" in r.stdout, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("n_rb,cell,frames,fs", [(6, 17, 30, "1.92"), (25, 301, 24, "7.68"), (100, 77, 12, "30.72")])
def test_cell_scan_matches_reference_output(tmp_path, n_rb, cell, frames, fs):
    """BASELINE config 1 / SURVEY 8d W1 (plumbing): a GNU-Radio-free cell scan in LTE_fdd_dl_file_scan's call order --
    coarse timing, PSS, SSS, PBCH, PDCCH, PDSCH -- over an int8 capture written by capture_gen (the reference's TX API).
    scan_gpu = the scanner linked against the reference's objects with the hot-path entry points replaced by the shim;
    its report (cell id, MIB, SIB1, SIB2 fields, transport-block counts) must equal the all-reference build's text."""
    build = os.path.join(ROOT, "shim", "_build")
    gen, scan_gpu, scan_cpu = (os.path.join(build, n) for n in ("capture_gen", "scan_gpu", "scan_cpu"))
    if not (os.path.exists(gen) and os.path.exists(scan_gpu)):
        pytest.skip("shim/_build/capture_gen / scan_gpu not built (need the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, str(n_rb), str(cell), str(frames)], check=True, timeout=600)
    want = open(os.path.join(ROOT, "tests", "golden", "scan_%drb_reference_cpu.txt" % n_rb)).read()
    if os.path.exists(scan_cpu):
        assert subprocess.run([scan_cpu, cap, fs], capture_output=True, text=True, timeout=900).stdout == want
    got = subprocess.run([scan_gpu, cap, fs], capture_output=True, text=True, timeout=900)
    assert got.returncode == 0, got.stdout + got.stderr
    assert got.stdout == want
    # wall-clock of the two builds' phases (stderr of the scanner), kept next to the test run for INTEGRATION.md
    if os.path.exists(scan_cpu):
        cpu = subprocess.run([scan_cpu, cap, fs], capture_output=True, text=True, timeout=900)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "scan_timing_%drb.txt" % n_rb), "w") as f:
            f.write("scan_gpu %s" % got.stderr)
            f.write("scan_cpu %s" % cpu.stderr)


@pytest.mark.parametrize("n_rb,cell,frames,fs", [(6, 17, 30, "1.92"), (25, 301, 24, "7.68"), (100, 77, 12, "30.72")])
def test_cell_scan_with_no_reference_phy_object_in_the_link(tmp_path, n_rb, cell, frames, fs):
    """scan_gpu_pure = the same scanner source (scan_demo.cc) linked with NO object of the reference's PHY: liblte_phy_init,
    liblte_phy_cleanup and liblte_phy_update_n_rb_dl are the shim's own (liblte_phy_shim.cc, -DMI_LTE_SHIM_OWN_LIFECYCLE), the other
    seven calls of LTE_fdd_dl_file_scan are the replaced entry points.  Its report must equal the all-reference build's text, and the
    executable must not contain a single function of the reference's PHY besides the ones the shim defines."""
    build = os.path.join(ROOT, "shim", "_build")
    gen, pure = os.path.join(build, "capture_gen"), os.path.join(build, "scan_gpu_pure")
    if not (os.path.exists(gen) and os.path.exists(pure)):
        pytest.skip("shim/_build/capture_gen / scan_gpu_pure not built (need the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, str(n_rb), str(cell), str(frames)], check=True, timeout=600)
    want = open(os.path.join(ROOT, "tests", "golden", "scan_%drb_reference_cpu.txt" % n_rb)).read()
    got = subprocess.run([pure, cap, fs], capture_output=True, text=True, timeout=900)
    assert got.returncode == 0, got.stdout + got.stderr
    assert got.stdout == want
    syms = subprocess.run(["nm", "-C", "--defined-only", pure], capture_output=True, text=True).stdout
    phy = sorted({l.split(" T ")[1].split("(")[0] for l in syms.splitlines() if " T liblte_phy_" in l})
    # the reference's 34 symbols (the receive side, the lifecycle, the seven scheduler-side helpers, the ten transmit functions) + the shim's own
    # one-call uplink entry
    assert len(phy) == N_PURE_SYMBOLS and all(f in phy for f in ("liblte_phy_init", "liblte_phy_update_n_rb_dl", "liblte_phy_ul_init", "liblte_phy_get_tbs_mcs_and_n_prb_for_dl",
                                                     "liblte_phy_get_n_cce", "liblte_phy_code_block_segmentation", "liblte_phy_ul_subframe_decode",
                                                                "liblte_phy_pdsch_channel_encode", "liblte_phy_create_dl_subframe")), phy
    assert "pdcch_permute_pre_calc" not in syms and "turbo_decode" not in syms and "fftwf_" not in syms


@pytest.mark.parametrize("n_rb,cell,frames,fs", [(6, 17, 30, "1.92"), (100, 77, 12, "30.72")])
def test_transmit_and_scan_with_no_reference_phy_object_on_either_side(tmp_path, n_rb, cell, frames, fs):
    """The whole loop of BASELINE config 1 without liblte_phy.cc: capture_gen_pure (the generator on the library's host-side transmit functions)
    writes the capture, scan_gpu_pure (the scanner on the GPU receive chains) reads it -- and prints what the all-reference scanner printed for
    the all-reference generator's file (tests/golden/scan_*_reference_cpu.txt)."""
    build = os.path.join(ROOT, "shim", "_build")
    gen, pure = os.path.join(build, "capture_gen_pure"), os.path.join(build, "scan_gpu_pure")
    if not (os.path.exists(gen) and os.path.exists(pure)):
        pytest.skip("shim/_build/capture_gen_pure / scan_gpu_pure not built (need the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, str(n_rb), str(cell), str(frames)], check=True, timeout=600, capture_output=True)
    want = open(os.path.join(ROOT, "tests", "golden", "scan_%drb_reference_cpu.txt" % n_rb)).read()
    got = subprocess.run([pure, cap, fs], capture_output=True, text=True, timeout=900)
    assert got.returncode == 0, got.stdout + got.stderr
    assert got.stdout == want


@pytest.mark.parametrize("n_rb,cell,frames,fs,cfo,lead", [(6, 17, 30, "1.92", 731, 311), (25, 301, 24, "7.68", -1180, 1000), (100, 77, 12, "30.72", 2350, 4321)])
def test_batch_scanner_with_carrier_offset_matches_reference_output(tmp_path, n_rb, cell, frames, fs, cfo, lead):
    """The same scan on the library's BATCH entry points (shim/scan_batch.cc: no liblte_phy object linked, one launch per stage for all
    subframes of a phase, the capture resident in HBM) over captures with a carrier offset and a frame that does not start at sample 0:
    the frequency correction between the synchronisation stages (LTE_fdd_dl_fs_samp_buf.cc:696-713) runs on the device
    (mi_lte_freq_shift_run).  Its report must equal what the all-reference per-call scanner prints for the same file."""
    build = os.path.join(ROOT, "shim", "_build")
    gen, scan_batch, scan_cpu = (os.path.join(build, n) for n in ("capture_gen", "scan_batch", "scan_cpu"))
    if not (os.path.exists(gen) and os.path.exists(scan_batch) and os.path.exists(scan_cpu)):
        pytest.skip("shim/_build/capture_gen / scan_batch / scan_cpu not built (need the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, str(n_rb), str(cell), str(frames), str(cfo), str(lead)], check=True, timeout=600)
    want = subprocess.run([scan_cpu, cap, fs], capture_output=True, text=True, timeout=900)
    assert want.returncode == 0 and "SIB1:" in want.stdout, want.stdout  # the reference finds the cell through the offset
    got = subprocess.run([scan_batch, cap, fs], capture_output=True, text=True, timeout=900)
    assert got.returncode == 0, got.stdout + got.stderr
    assert got.stdout == want.stdout
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "scan_batch_timing_%drb.txt" % n_rb), "w") as f:
        f.write("scan_batch %s" % got.stderr)
        f.write("scan_cpu   %s" % want.stderr)


@pytest.mark.parametrize("n_rb,fft,cell,frames,mode", [(25, 512, 301, 4, 0), (100, 2048, 77, 4, 1), (6, 128, 17, 8, 1)])
def test_one_call_per_subframe_equals_the_per_call_sequence(tmp_path, n_rb, fft, cell, frames, mode):
    """mi_lte_dl_subframe_decode_host == mi_lte_get_dl_subframe_and_ce_host + mi_lte_pdcch_channel_decode_host +
    mi_lte_pdsch_channel_decode_host per DCI (the loop of LTE_fdd_dl_fs_samp_buf.cc:445-515; those three are pinned to the reference's
    output by the scan tests above) on every subframe of a capture written by the reference's transmitter: same return value, control
    format, DCIs, verdicts and transport blocks.  mode 1 runs the per-call sequence under the explicit cache contract
    (mi_lte_host_cache_set_mode): nothing is hashed, every decode still reads the copy the front end left in HBM."""
    import ctypes as C
    import openlte_amd as m
    from openlte_amd.lib import PdcchDci
    gen = os.path.join(ROOT, "shim", "_build", "capture_gen")
    if not os.path.exists(gen):
        pytest.skip("shim/_build/capture_gen not built (needs the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, str(n_rb), str(cell), str(frames)], check=True, timeout=600)
    raw = np.fromfile(cap, np.int8)
    per_sf = 15 * fft
    i_s = np.zeros(len(raw) // 2 + 4 * fft, np.float32)
    q_s = np.zeros_like(i_s)
    i_s[:len(raw) // 2], q_s[:len(raw) // 2] = raw[0::2], raw[1::2]
    ctx = m.Context(0)
    L = ctx.L
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u32 = C.c_uint32
    L.mi_lte_get_dl_subframe_and_ce_host.argtypes = [C.c_void_p, u32, u32, f32p, f32p] + [u32] * 4 + [f32p] * 4
    L.mi_lte_pdcch_channel_decode_host.argtypes = [C.c_void_p, u32, f32p, f32p, f32p, f32p, u32, u32, u32, C.c_float, u32, u32] + [C.POINTER(u32)] * 3 + [C.POINTER(PdcchDci)]
    L.mi_lte_pdsch_channel_decode_host.argtypes = [C.c_void_p, u32, f32p, f32p, f32p, f32p, u32, C.c_void_p, u32, u32, u32,
                                                   np.ctypeslib.ndpointer(np.uint8), C.POINTER(u32)]
    L.mi_lte_dl_subframe_decode_host.argtypes = [C.c_void_p, u32, u32, f32p, f32p] + [u32] * 4 + [C.c_float, u32, u32] + [C.POINTER(u32)] * 3 + [
        C.POINTER(PdcchDci), np.ctypeslib.ndpointer(np.uint8), u32, C.POINTER(u32), C.POINTER(C.c_int32)]
    L.mi_lte_host_cache_set_mode.argtypes = [C.c_void_p, u32]
    L.mi_lte_host_cache_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(u32)]
    assert L.mi_lte_host_cache_set_mode(ctx.h, 2) != 0
    assert L.mi_lte_host_cache_set_mode(ctx.h, mode) == 0
    sr, si = np.zeros((16, 1200), np.float32), np.zeros((16, 1200), np.float32)
    cr, ci = np.zeros((4, 16, 1200), np.float32), np.zeros((4, 16, 1200), np.float32)
    n_blocks = n_found = 0
    for f in range(frames - 1):
        for sf in range(10):
            start = f * 10 * per_sf
            # the per-call sequence
            assert 0 == L.mi_lte_get_dl_subframe_and_ce_host(ctx.h, fft, n_rb, i_s, q_s, start, sf, cell, 1, sr, si, cr, ci)
            cfi, nsym, ndci = u32(), u32(), u32()
            dci = (PdcchDci * 6)()
            rc_a = L.mi_lte_pdcch_channel_decode_host(ctx.h, n_rb, sr, si, cr, ci, sf, cell, 1, 1.0, 0, 0, C.byref(cfi), C.byref(nsym), C.byref(ndci), dci)
            blocks = []
            if rc_a == 0:
                for k in range(ndci.value):
                    out, n = np.zeros(6144, np.uint8), u32()
                    rc = L.mi_lte_pdsch_channel_decode_host(ctx.h, n_rb, sr, si, cr, ci, sf, C.addressof(dci[k].alloc), nsym.value, cell, 1, out, C.byref(n))
                    blocks.append((rc, out[:n.value].copy() if rc == 0 else None))
            # one call
            cfi2, nsym2, ndci2 = u32(), u32(), u32()
            dci2 = (PdcchDci * 6)()
            out2, n2, st2 = np.zeros((6, 6144), np.uint8), (u32 * 6)(), (C.c_int32 * 6)()
            rc_b = L.mi_lte_dl_subframe_decode_host(ctx.h, fft, n_rb, i_s, q_s, start, sf, cell, 1, 1.0, 0, 0, C.byref(cfi2), C.byref(nsym2), C.byref(ndci2), dci2,
                                                    out2.reshape(-1), 6144, n2, st2)
            assert rc_b == rc_a, (f, sf, rc_a, rc_b, L.mi_lte_last_error(ctx.h))
            if rc_a != 0:
                continue
            assert (cfi2.value, nsym2.value, ndci2.value) == (cfi.value, nsym.value, ndci.value)
            n_found += ndci.value
            for k in range(ndci.value):
                assert bytes(dci2[k]) == bytes(dci[k]), (f, sf, k)
                rc, bits = blocks[k]
                assert st2[k] == rc, (f, sf, k, rc, st2[k])
                if rc == 0:
                    assert n2[k] == len(bits) and (out2[k, :n2[k]] == bits).all(), (f, sf, k)
                    n_blocks += 1
    # the capture carries SIB1 in subframe 5 of even frames and further system information behind it: something was decoded
    assert n_found >= (frames - 1) // 2 and n_blocks >= (frames - 1) // 2, (n_found, n_blocks)
    a, b, c = C.c_uint64(), C.c_uint64(), u32()
    assert L.mi_lte_host_cache_stats(ctx.h, C.byref(a), C.byref(b), C.byref(c)) == 0
    assert b.value == 0, "a decode of the per-call sequence uploaded the subframe its front end had just produced"


def test_cell_scan_under_the_explicit_cache_contract(tmp_path):
    """The 20 MHz scan through the shim with MI_LTE_SHIM_EXPLICIT_CACHE set (no content hash per decode call): same report."""
    build = os.path.join(ROOT, "shim", "_build")
    gen, scan_gpu = os.path.join(build, "capture_gen"), os.path.join(build, "scan_gpu")
    if not (os.path.exists(gen) and os.path.exists(scan_gpu)):
        pytest.skip("shim/_build/capture_gen / scan_gpu not built (need the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, "100", "77", "12"], check=True, timeout=600)
    want = open(os.path.join(ROOT, "tests", "golden", "scan_100rb_reference_cpu.txt")).read()
    runs = {}
    for name, extra in (("fingerprint", {}), ("explicit", {"MI_LTE_SHIM_EXPLICIT_CACHE": "1"})):
        got = subprocess.run([scan_gpu, cap, "30.72"], capture_output=True, text=True, timeout=900, env=dict(os.environ, MI_LTE_SHIM_STATS="1", **extra))
        assert got.returncode == 0 and got.stdout == want, (name, got.stdout, got.stderr)
        runs[name] = got.stderr
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "scan_timing_cache_modes.txt"), "w") as f:
        for name, err in runs.items():
            f.write("%s: %s" % (name, err))


def test_per_call_forms_fail_an_off_carrier_allocation_like_the_reference(ctx, ref):
    """A DCI whose CRC matched by chance can name resource blocks past the carrier.  The reference demodulates whatever lies behind the
    grid row and fails the transport block; the per-call downlink form must hand back the same return code (LIBLTE_ERROR_DECODE_FAIL, 2)
    instead of the plans' argument error, and the good allocation of the same subframe must still decode afterwards.  The uplink form
    reports LIBLTE_ERROR_INVALID_INPUTS (1), where the reference's NaN accident reports a decoded all-zero block (see below)."""
    import ctypes as C
    import openlte_amd as m
    from oracle import pyoracle as po
    L = ctx.L
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    L.mi_lte_get_dl_subframe_and_ce_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, f32p, f32p] + [C.c_uint32] * 4 + [f32p] * 4
    L.mi_lte_pdsch_channel_decode_host.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, f32p, f32p, C.c_uint32, C.c_void_p,
                                                   C.c_uint32, C.c_uint32, C.c_uint32, u8p, C.POINTER(C.c_uint32)]
    # ---- downlink, 5 MHz: resource blocks 22..26 of a 25-block carrier
    cap = td.multi_port_capture(ref, 1, seed=5, fft=512, nrb=25, cell=77, sf=3, cfi=2, mod=2, tbs=1064, prbs=list(range(4, 12)))
    n_samp, sf, cell, iq, la, phy = 30720 // 4, cap["sf"], cap["cell"], cap["iq"], cap["la"], cap["phy"]
    i_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * n_samp, np.float32), iq[:, 0].astype(np.float32)]))
    q_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * n_samp, np.float32), iq[:, 1].astype(np.float32)]))
    rx = ref.ref_subframe_new()
    assert ref.ref_get_dl_subframe_and_ce(phy, i_f, q_f, 0, sf, cell, 1, rx) == 0
    off_prbs = list(range(22, 27))
    la_off = po.make_alloc(2, 1064, off_prbs, 0x2345, 0, 1, 0)
    out, n = np.zeros(6200, np.uint8), C.c_uint32()
    rc_ref_off = ref.ref_pdsch_channel_decode(phy, rx, C.byref(la_off), 2, cell, 1, out, C.byref(n))
    rc_ref_ok = ref.ref_pdsch_channel_decode(phy, rx, C.byref(la), 2, cell, 1, out, C.byref(n))
    assert rc_ref_off == 2 and rc_ref_ok == 0
    sr, si = np.zeros((16, 1200), np.float32), np.zeros((16, 1200), np.float32)
    cr, ci = np.zeros((4, 16, 1200), np.float32), np.zeros((4, 16, 1200), np.float32)
    assert 0 == L.mi_lte_get_dl_subframe_and_ce_host(ctx.h, 512, 25, i_f, q_f, 0, sf, cell, 1, sr, si, cr, ci)
    a_off, a_ok = m.make_alloc(0, 2, 1064, off_prbs, 0x2345), m.make_alloc(0, 2, 1064, cap["prbs"], 0x2345)
    got, gn = np.zeros(6200, np.uint8), C.c_uint32()
    assert L.mi_lte_pdsch_channel_decode_host(ctx.h, 25, sr, si, cr, ci, sf, C.addressof(a_off), 2, cell, 1, got, C.byref(gn)) == rc_ref_off
    assert L.mi_lte_pdsch_channel_decode_host(ctx.h, 25, sr, si, cr, ci, sf, C.addressof(a_ok), 2, cell, 1, got, C.byref(gn)) == rc_ref_ok
    assert gn.value == 1064 and (got[:1064] == cap["msg"]).all()
    ref.ref_subframe_free(rx)
    ref.ref_phy_free(phy)
    # ---- uplink, 5 MHz: a four-block PUSCH allocation that ends one block past the carrier
    L.mi_lte_pusch_channel_decode_host.argtypes = [C.c_void_p, C.c_uint32, f32p, f32p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32] + [f32p] * 4 + [
        u8p, C.POINTER(C.c_uint32)]
    ulc = (3, 0, 0, 2, 5)
    phy = ref.ref_phy_new(po.FS_ENUM[512], 17, 1, 25)
    assert ref.ref_ul_init(phy, 17, *ulc) == 0
    rng = np.random.default_rng(9)
    sfp = ref.ref_subframe_new()
    ref.ref_subframe_set_num(sfp, 2)
    re, im = (rng.standard_normal(7680).astype(np.float32) * 20 for _ in range(2))
    assert ref.ref_get_ul_subframe(phy, re, im, sfp) == 0
    la_ul = po.make_alloc(1, 504, [22, 23, 24, 25], 0x100, 0, 1)
    rc_ref_ul = ref.ref_pusch_channel_decode(phy, sfp, C.byref(la_ul), 17, 1, out, C.byref(n))
    # What the reference does here is an accident, pinned as such: the columns past the carrier were never written, its channel estimate
    # over them divides by zero, the NaNs swallow the whole allocation, the decoder turns NaN soft bits into an all-zero block and the CRC
    # of an all-zero block is zero -- "decoded", rc 0, 504 zero bits, whatever was received on the other three blocks.  The library
    # reports the allocation as undecodable (1 = LIBLTE_ERROR_INVALID_INPUTS, the reference's failure value on this path): INTEGRATION.md,
    # table of differences.
    assert rc_ref_ul == 0 and n.value == 504 and not out[:504].any()
    usr, usi = np.zeros((16, 1200), np.float32), np.zeros((16, 1200), np.float32)
    usr[:14], usi[:14] = po.ref_subframe_view(ref, sfp, 0)[:14], po.ref_subframe_view(ref, sfp, 1)[:14]
    dm = np.zeros(4 * 48, np.float32)
    ref.ref_get_pusch_dmrs(phy, 2, 4, dm)
    d = [np.ascontiguousarray(dm[k * 48:(k + 1) * 48]) for k in range(4)]
    a_ul = m.make_alloc(0, 1, 504, [22, 23, 24, 25], 0x100)
    assert L.mi_lte_pusch_channel_decode_host(ctx.h, 25, usr, usi, 2, C.addressof(a_ul), 17, 1, d[0], d[1], d[2], d[3], got, C.byref(gn)) == 1
    la_in = po.make_alloc(1, 504, [21, 22, 23, 24], 0x100, 0, 1)  # the same noise inside the carrier: both fail, the same way
    a_in = m.make_alloc(0, 1, 504, [21, 22, 23, 24], 0x100)
    assert ref.ref_pusch_channel_decode(phy, sfp, C.byref(la_in), 17, 1, out, C.byref(n)) == 1
    assert L.mi_lte_pusch_channel_decode_host(ctx.h, 25, usr, usi, 2, C.addressof(a_in), 17, 1, d[0], d[1], d[2], d[3], got, C.byref(gn)) == 1
    ref.ref_subframe_free(sfp)
    ref.ref_phy_free(phy)


def test_scanners_read_gr_complex_captures(tmp_path):
    """LTE_fdd_dl_file_scan reads int8 pairs or gr_complex (LTE_fdd_dl_fs_samp_buf.cc:657-694); so do the three scanners.  The same capture
    as float32 pairs must give the report of the int8 file -- per-call scanner on the shim, batch scanner on the C-ABI (device-side
    de-interleave, mi_lte_iq_f32_pairs_to_planar), and the all-reference build."""
    build = os.path.join(ROOT, "shim", "_build")
    gen, scan_gpu, scan_cpu, scan_batch = (os.path.join(build, n) for n in ("capture_gen", "scan_gpu", "scan_cpu", "scan_batch"))
    if not all(os.path.exists(x) for x in (gen, scan_gpu, scan_batch)):
        pytest.skip("shim/_build not built (needs the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, "25", "301", "24"], check=True, timeout=600)
    capf = os.path.join(str(tmp_path), "capture.fc32")
    np.fromfile(cap, np.int8).astype(np.float32).tofile(capf)
    want = open(os.path.join(ROOT, "tests", "golden", "scan_25rb_reference_cpu.txt")).read()
    for exe in (scan_gpu, scan_batch) + ((scan_cpu,) if os.path.exists(scan_cpu) else ()):
        got = subprocess.run([exe, capf, "7.68", "9", "gr_complex"], capture_output=True, text=True, timeout=900)
        assert got.returncode == 0, (exe, got.stdout + got.stderr)
        assert got.stdout == want, exe
