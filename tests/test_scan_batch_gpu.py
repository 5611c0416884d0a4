"""GPU: the whole receiver on the batch entry points (tools/scan_batch.py) against the all-reference scanner's report.

capture_gen (the reference's transmit API) writes a capture with PSS/SSS/CRS/PBCH, SIB1 in subframe 5 of even frames and SIB2 in
subframe 3 of every 8th; scan_cpu (the reference's receive API, one call at a time) printed tests/golden/scan_*rb_reference_cpu.txt.
The batch scan must find the same cell at the same frame start with the same MIB, and decode -- with a passing CRC -- every
transport block the reference's scanner reports, with the same CFI / size / PRB count / redundancy version."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.parametrize("n_rb,cell,frames,fft", [(6, 17, 30, 128), (25, 301, 24, 512), (100, 77, 12, 2048)])
def test_batch_scan_agrees_with_the_reference_scanner(ctx, tmp_path, n_rb, cell, frames, fft):
    import scan_batch
    gen = os.path.join(ROOT, "shim", "_build", "capture_gen")
    if not os.path.exists(gen):
        pytest.skip("shim/_build/capture_gen not built (needs the reference tree at build time)")
    cap = os.path.join(str(tmp_path), "capture.bin")
    subprocess.run([gen, cap, str(n_rb), str(cell), str(frames)], check=True, timeout=600, stdout=subprocess.DEVNULL)
    iq = np.fromfile(cap, np.int8).reshape(-1, 2)
    rep = scan_batch.scan(ctx, iq, fft)
    want = open(os.path.join(ROOT, "tests", "golden", "scan_%drb_reference_cpu.txt" % n_rb)).read()
    m = re.search(r"cell (\d+): frame start (\d+), (\d+) antenna port\(s\), MIB: N_rb_dl=(\d+) phich_dur=(\d+) phich_res=(\d+) sfn=(\d+)", want)
    w_cell, w_start, w_ant, w_rb, w_dur, w_res, w_sfn = map(int, m.groups())
    assert len(rep["cells"]) == 1
    c = rep["cells"][0]
    assert (c["cell"], c["frame_start"], c["N_ant"], c["N_rb_dl"], c["phich_dur"], c["phich_res"]) == (w_cell, w_start, w_ant, w_rb, w_dur, w_res)
    assert c["sfn_first_frame"] == w_sfn
    assert c["pbch_frames_decoded"] == c["frames"] and c["cfi_decoded"] >= len(c["transport_blocks"]) > 0  # (only SI subframes carry a PCFICH here)
    ok = {(b["sfn"], b["subframe"]): b for b in c["transport_blocks"] if b["crc_ok"]}
    assert all(b["crc_ok"] for b in c["transport_blocks"]), "an announced transport block failed its CRC on a noise-free capture"
    lines = re.findall(r"sfn (\d+) subframe (\d+): CFI=(\d+) tbs=(\d+) N_prb=(\d+)(?: rv=(\d+))?", want)
    assert len(lines) >= 2  # SIB1 and SIB2
    for sfn, sf, ns, tbs, nprb, rv in lines:
        b = ok[(int(sfn), int(sf))]
        assert (b["n_symbs"], b["tbs"], b["N_prb"]) == (int(ns), int(tbs), int(nprb)) and b["rnti"] == 0xFFFF
        if rv:
            assert b["rv"] == int(rv)
    n_after = int(re.search(r"(\d+) PDSCH transport blocks decoded after SIB1", want).group(1))
    assert len(ok) >= n_after + 1
    # SIB1 repeats with the same content every 20 ms (other redundancy versions): all copies decode to the same bits
    sib1 = [b["bits"][:b["tbs"]] for k, b in sorted(ok.items()) if k[1] == 5]
    assert len(sib1) >= 2 and all((x == sib1[0]).all() for x in sib1)
