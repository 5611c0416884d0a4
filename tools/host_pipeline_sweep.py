#!/usr/bin/env python3
"""GPU: mi_lte_dl_pipeline throughput for a few (chunk, lanes) choices, next to the bare pinned H2D copy rate of the same bytes."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import lte_testdata as td  # noqa: E402
import openlte_amd as m  # noqa: E402
from openlte_amd import synth  # noqa: E402

cfg = m.DlCfg(2048, 100, 1, m.IQ_I8 | m.CE_COMPACT)
U, n = 32, 32768
sfs = np.array([[1, 2, 3, 4, 6, 7, 8, 9][i % 8] for i in range(U)], np.uint32)
cells = ((np.arange(U) * 37) % 504).astype(np.uint32)
allocs = []
for u in range(U):
    allocs += td.w4_allocs(u)
iq, tx = synth.dl_units(cfg, sfs, cells, allocs, 9, snr_db=30.0, max_delay=8, seed=1)
ul = iq.shape[1]
h_iq, h_sf, h_cell = m.HostBuffer((n, ul, 2), np.int8), m.HostBuffer((n,), np.uint32), m.HostBuffer((n,), np.uint32)
for c0 in range(0, n, U):
    h_iq.arr[c0:c0 + U] = iq
h_sf.arr[:], h_cell.arr[:] = sfs[np.arange(n) % U], cells[np.arange(n) % U]
ctx = m.Context(0)
d = ctx.alloc(h_iq.arr.nbytes)
L = ctx.L
L.mi_lte_memcpy_h2d(ctx.h, d.ptr, h_iq.arr.ctypes.data, h_iq.arr.nbytes)
t0 = time.perf_counter()
for _ in range(3):
    L.mi_lte_memcpy_h2d(ctx.h, d.ptr, h_iq.arr.ctypes.data, h_iq.arr.nbytes)
dt = (time.perf_counter() - t0) / 3
print("bare pinned H2D of the batch (%.2f GB): %.1f GB/s = %.0f k subframes/s if nothing else happened" % (h_iq.arr.nbytes / 1e9, h_iq.arr.nbytes / dt / 1e9, n / dt / 1e3))
d.free()
for chunk, lanes in ((8192, 2), (4096, 3), (4096, 4), (2048, 4), (2048, 6), (1024, 8)):
    pipe = m.DlPipeline(0, cfg, 2, td.w4_allocs(0), chunk, lanes)
    h_out, h_st = m.HostBuffer((n * 9, pipe.out_stride), np.uint8), m.HostBuffer((n * 9,), np.int32)
    pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n, h_out.arr, h_st.arr)
    t0 = time.perf_counter()
    for _ in range(3):
        pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n, h_out.arr, h_st.arr)
    dt = (time.perf_counter() - t0) / 3
    print("chunk %5d x %d lanes: %.0f k subframes/s, %.1f GB/s H2D, crc ok %d/%d" % (chunk, lanes, n / dt / 1e3, h_iq.arr.nbytes / dt / 1e9, int((h_st.arr == 0).sum()), n * 9))
    pipe.close()
    h_out.free(); h_st.free()
