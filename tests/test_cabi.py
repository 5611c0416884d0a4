"""CPU tests: the C-ABI library loads and exports every symbol include/*.h declares (no compute
calls here -- this container has no GPU), and refuses to create a context without a GPU."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "*.h")):
        txt = re.sub(r"/\*.*?\*/", "", open(h).read(), flags=re.S)
        names |= set(re.findall(r"\b(mi_lte_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    import openlte_amd
    L = openlte_amd.load_library()
    syms = declared_symbols()
    assert len(syms) >= 20
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, "declared in include/*.h but not exported: %s" % missing


def test_header_cites_reference_interfaces():
    txt = open(os.path.join(ROOT, "include", "mi_lte.h")).read()
    assert "liblte_phy.cc:" in txt and "liblte_phy.h" in txt


def test_no_cpu_fallback_without_gpu():
    import openlte_amd
    L = openlte_amd.load_library()
    if L.mi_lte_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(openlte_amd.MiLteError):
        openlte_amd.Context(0)


def test_product_never_touches_the_oracle():
    """The product package and the library sources must not reference oracle/."""
    bad = []
    for path in glob.glob(os.path.join(ROOT, "openlte_amd", "**", "*"), recursive=True):
        if os.path.isfile(path) and path.endswith((".py", ".cc", ".hip", ".hpp", ".h")):
            if re.search(r"\boracle\b", open(path, errors="ignore").read()):
                bad.append(os.path.relpath(path, ROOT))
    assert not bad, bad


def test_own_lifecycle_equals_the_references():
    """shim/_build/lifecycle_check: the shim's own liblte_phy_init / liblte_phy_update_n_rb_dl / liblte_phy_cleanup against the
    reference's (linked into the TEST binary under other names): return codes and every field a caller or a replaced entry point reads,
    for every sampling rate x bandwidth x PHICH resource x prefix (liblte_phy.cc:2210-2335, :2592-2647).  Needs no GPU: without one the
    struct simply has no context."""
    import subprocess
    exe = os.path.join(ROOT, "shim", "_build", "lifecycle_check")
    if not os.path.exists(exe):
        pytest.skip("shim/_build/lifecycle_check not built (needs the reference tree at build time)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "combinations equal" in r.stdout, r.stdout[-2000:]


def test_scheduler_helpers_equal_the_references():
    """`lifecycle_check helpers`: the seven scheduler-side functions the own-lifecycle shim defines (sched.cc: the TBS / MCS / PRB searches of
    liblte_phy.cc:6251-6475, liblte_phy_get_n_cce :6477-6505, liblte_phy_pucch_map_sr_config_idx :3183-3217, code block segmentation and
    desegmentation :9753-9987) against the reference's own, over their whole argument ranges (4.5 million calls): return values and every
    output, including what a failed search leaves untouched and the multi-code-block path's CRC over the caller's buffer.  CPU only."""
    import subprocess
    exe = os.path.join(ROOT, "shim", "_build", "lifecycle_check")
    if not os.path.exists(exe):
        pytest.skip("shim/_build/lifecycle_check not built (needs the reference tree at build time)")
    r = subprocess.run([exe, "helpers"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "comparisons equal" in r.stdout, r.stdout[-2000:]


def test_transmit_functions_equal_the_references():
    """`lifecycle_check tx`: the transmit functions the own-lifecycle shim defines (tx.cc ...: liblte_phy_rate_match_turbo :11081-11237,
    _pdsch_channel_encode :3489-3688, _bch_channel_encode :3863-3966, _pdcch_channel_encode :4113-4517, _map_crs / _pss / _sss :5144 / :5265 / :5520,
    _create_dl_subframe :5862-5903, _pusch_channel_encode :2664-2799, _generate_prach :3219-3297) against the reference's own on the same sequences of calls: grids bit for bit (every cell, subframe, port count and bandwidth for the
    signals; random allocations incl. filler bits, several code blocks, two codewords, four ports, BPSK for the PDSCH; 40 ms periods entered in the
    middle for the PBCH; PCFICH + PHICH groups + up to six DCIs per control region for every PHICH resource and port count, with what the call
    writes back into the caller's structs), return codes, and the three transforms (OFDM, SC-FDMA, PRACH) to float rounding.  CPU only."""
    import subprocess
    exe = os.path.join(ROOT, "shim", "_build", "lifecycle_check")
    if not os.path.exists(exe):
        pytest.skip("shim/_build/lifecycle_check not built (needs the reference tree at build time)")
    r = subprocess.run([exe, "tx"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "comparisons equal" in r.stdout, r.stdout[-3000:]


@pytest.mark.parametrize("n_rb,cell,frames", [(6, 17, 4), (25, 301, 3), (100, 77, 2)])
def test_capture_generator_without_the_references_phy_writes_the_same_file(tmp_path, n_rb, cell, frames):
    """shim/_build/capture_gen_pure: the capture generator (LTE_fdd_dl_file_gen's sequence of transmit calls: map_pss / _sss / _crs, bch / pdcch /
    pdsch channel encode with SIB1, create_dl_subframe, the TBS search) linked with NO object of the reference's PHY -- the shim's own lifecycle
    and the library's host-side transmit functions -- must write the file the all-reference build writes, byte for byte.  CPU only."""
    import subprocess
    build = os.path.join(ROOT, "shim", "_build")
    ref, own = os.path.join(build, "capture_gen"), os.path.join(build, "capture_gen_pure")
    if not (os.path.exists(ref) and os.path.exists(own)):
        pytest.skip("shim/_build/capture_gen / capture_gen_pure not built (need the reference tree at build time)")
    a, b = os.path.join(str(tmp_path), "ref.bin"), os.path.join(str(tmp_path), "own.bin")
    subprocess.run([ref, a, str(n_rb), str(cell), str(frames)], check=True, timeout=600, capture_output=True)
    subprocess.run([own, b, str(n_rb), str(cell), str(frames)], check=True, timeout=600, capture_output=True)
    assert os.path.getsize(a) > 100000 and open(a, "rb").read() == open(b, "rb").read()
    syms = subprocess.run(["nm", "-C", own], capture_output=True, text=True).stdout
    assert "fftwf_" not in syms and "turbo_constituent_encoder" not in syms and "dci_1a_pack" not in syms  # nothing of liblte_phy.cc or FFTW in the link


def test_transmit_functions_refuse_what_the_reference_has_no_defined_behaviour_for():
    """The host-side transmit entry points return 1 (the reference's LIBLTE_ERROR_INVALID_INPUTS) for NULL grids, a cell identity past 503, three
    antenna ports, a pre-coder type the reference leaves without output, an allocation beyond its 10 000-bit arrays and a PUSCH size it has no
    transform plan for -- and leave the grid alone.  Through the Python mirror (openlte_amd.Transmitter); CPU only."""
    import ctypes as C
    import numpy as np
    import openlte_amd as m
    L = m.load_library()
    t = m.Transmitter(2048, 100, 17)
    re, im = t.re, t.im
    assert L.mi_lte_map_crs(100, 12, 0, 504, 1, re, im) == 1 and L.mi_lte_map_crs(100, 12, 0, 17, 5, re, im) == 1
    assert L.mi_lte_map_crs(100, 12, 0, 17, 1, re, im) == 0 and np.count_nonzero(re[0, 0]) == 200
    a = m.make_alloc(0, 3, 3240, list(range(12)), 0x100)
    bits = np.zeros(3240, np.uint8)
    t.pdsch(1, 2, [(a, bits)])  # W4's allocation: fine
    before = re.copy()
    big = m.make_alloc(0, 3, 3240, list(range(13)), 0x100)  # 13 PRB x 138 elements x 6 bits = 10 764 > 10 000
    with pytest.raises(m.MiLteError):
        t.pdsch(1, 2, [(big, bits)])
    assert np.array_equal(before, re)
    t.n_ant = 3
    with pytest.raises(m.MiLteError):
        t.pdsch(1, 2, [(a, bits)])
    t.n_ant = 1
    arr = (m.TxAlloc * 1)()
    arr[0].N_prb, arr[0].mod_type, arr[0].tbs, arr[0].N_codewords, arr[0].pre_coder_type = 2, 1, 120, 1, 1  # spatial multiplexing on two ports
    assert L.mi_lte_pdsch_channel_encode(t.h, 100, 12, arr, 1, 2, 17, 2, 1, re, im) == 1
    assert L.mi_lte_pdsch_channel_encode(None, 100, 12, arr, 1, 2, 17, 1, 1, re, im) == 1
    t.close()


def test_transmit_functions_under_address_and_undefined_behaviour_sanitizers():
    """tools/asan/run.sh: the host-side transmit sources (tx.cc, tx_ctrl.cc, tx_ul.cc, sched.cc, synth.cc, ul_rs.cc) compiled with g++
    -fsanitize=address,undefined and driven through every transmit entry point with random inputs, out-of-range ones included (PRB numbers past
    the grid, transport blocks past five code blocks, MCS past the table, control regions without a REG): no report.  CPU build only."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan", "run.sh"), "400"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "asan driver:" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, (r.stdout + r.stderr)[-3000:]


def test_tbs_table_lookup():
    """mi_lte_tbs: corner entries of 36.213 table 7.1.7.2.1-1 (values every LTE reference agrees on) and the out-of-range answer."""
    import openlte_amd
    L = openlte_amd.load_library()
    assert [L.mi_lte_tbs(0, 1), L.mi_lte_tbs(26, 1), L.mi_lte_tbs(0, 110), L.mi_lte_tbs(26, 110), L.mi_lte_tbs(9, 25)] == [16, 712, 3112, 75376, 4008]
    assert L.mi_lte_tbs(27, 1) == 0 and L.mi_lte_tbs(0, 0) == 0 and L.mi_lte_tbs(0, 111) == 0
