#!/usr/bin/env python3
"""GPU diagnostic: how many QPSK soft bits differ from the oracle's, and by how much (stage parity: the oracle's received grid is
uploaded, so the equaliser sees identical inputs).  Prints one line per case."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lte_testdata as td  # noqa: E402
import openlte_amd as m  # noqa: E402
from openlte_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from test_chain_gpu import oracle_pdsch, upload_oracle_subframe  # noqa: E402

port = po.port()
ctx = m.Context(0)
cfg = m.DlCfg(2048, 100, 1, 0)
tot = bad = 0
for snr in (300, 30, 12, 6, 0):
    for nprb, tbs in ((8, 680), (50, 1000), (25, 256)):
        sfs, cells = [1, 5, 0, 9], [17, 301, 503, 44]
        allocs = []
        for u in range(4):
            allocs += td.small_allocs(u, 100, 1, tbs, nprb, rnti=0x100 + u, first=45 if u in (1, 2) else 3 * u)
        iq, tx = synth.dl_units(cfg, sfs, cells, allocs, 1, snr_db=snr, max_delay=3, seed=tbs + snr)
        subs, want = [], []
        for u in range(4):
            lc, s = td.oracle_frontend(port, 2048, 100, 1, iq[u], sfs[u], cells[u])
            subs.append(upload_oracle_subframe(ctx, s, 1))
            want.append(oracle_pdsch(port, lc, s, allocs[u], 2, cells[u], 1))
        d_sub = ctx.to_device(np.concatenate(subs))
        plan = ctx.pdsch_plan(cfg, 2, allocs)
        st, bits = plan.run(d_sub, sfs, cells)
        for u in range(4):
            e, desc = plan.soft_bits(u), want[u][2]
            d = np.nonzero(e[:len(desc)] != desc)[0]
            tot += len(desc)
            bad += len(d)
            if len(d):
                print("snr %d nprb %d unit %d: %d of %d soft bits differ; max |delta| %d; first at %s: gpu %s oracle %s" %
                      (snr, nprb, u, len(d), len(desc), int(np.abs(e[d].astype(int) - desc[d]).max()), d[:4], e[d[:4]], desc[d[:4]]))
        plan.close()
        d_sub.free()
print("QPSK downlink stage parity: %d of %d soft bits differ" % (bad, tot))

# uplink: own front end (float tolerance upstream), reference live
R = po.ref()
if R is not None:
    from test_uplink_gpu import gpu_ul_decode
    for name, snr in (("20MHz_3ue", 30.0), ("20MHz_3ue", 3.0), ("1p4MHz_hop", 8.0), ("10MHz_prime", 20.0), ("20MHz_radices", 25.0), ("20MHz_16ue", 20.0)):
        case = td.ul_case(name, snr_db=snr, seed=int(snr) + 11)
        ws, want = td.ref_ul_decode(R, case)
        symb, st, bits, soft = gpu_ul_decode(ctx, case)
        n = sum(len(g) for _, _, g in want)
        nb = sum(int((soft[i] != g).sum()) for i, (_, _, g) in enumerate(want))
        mx = max(int(np.abs(soft[i].astype(int) - g).max()) for i, (_, _, g) in enumerate(want))
        print("uplink %s @%g dB: %d of %d soft bits differ, max |delta| %d" % (name, snr, nb, n, mx))
ctx.close()
