#!/usr/bin/env python3
"""GPU: where a run of mi_lte_dl_pipeline spends its time.  For a few (chunk, lanes) choices, in both copy modes
(copies on streams of their own -- the default since round 5 -- and on the lanes' streams, MI_LTE_PIPELINE_LANE_COPIES=1, run as a
second process by the caller), template mode and capture mode: wall time, the sums of the per-chunk H2D / kernel / D2H intervals
(HIP events, mi_lte_dl_pipeline_device_stats) and the rates they imply, next to the bare pinned copy of the same bytes."""
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

mode = "lane streams (MI_LTE_PIPELINE_LANE_COPIES=1)" if os.environ.get("MI_LTE_PIPELINE_LANE_COPIES") else "copy streams (default)"
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
L = ctx.L
import ctypes as C
stream = C.c_void_p(L.mi_lte_stream(ctx.h))
hip = C.CDLL("libamdhip64.so")
d = ctx.alloc(h_iq.arr.nbytes)
def bare(nbytes, reps=3):
    hip.hipMemcpyAsync(C.c_void_p(d.ptr), C.c_void_p(h_iq.arr.ctypes.data), C.c_size_t(nbytes), 1, stream); hip.hipStreamSynchronize(stream)
    t0 = time.perf_counter()
    for _ in range(reps):
        hip.hipMemcpyAsync(C.c_void_p(d.ptr), C.c_void_p(h_iq.arr.ctypes.data), C.c_size_t(nbytes), 1, stream)
    hip.hipStreamSynchronize(stream)
    return nbytes * reps / (time.perf_counter() - t0) / 1e9
print("== %s" % mode)
print("bare pinned H2D (hipMemcpyAsync on one stream): whole batch %.2f GB at %.1f GB/s; in pieces of 2048 / 4096 subframes back to back: %.1f / %.1f GB/s"
      % (h_iq.arr.nbytes / 1e9, bare(h_iq.arr.nbytes), bare(2048 * ul * 2, 16), bare(4096 * ul * 2, 8)))
d.free()
for chunk, lanes in ((2048, 2), (2048, 4), (2048, 6), (4096, 3), (4096, 4), (1024, 6)):
    pipe = m.DlPipeline(0, cfg, 2, td.w4_allocs(0), chunk, lanes)
    h_out, h_st = m.HostBuffer((n * 9, pipe.out_stride), np.uint8), m.HostBuffer((n * 9,), np.int32)
    pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n, h_out.arr, h_st.arr)
    best = None
    for _ in range(4):
        t0 = time.perf_counter()
        pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n, h_out.arr, h_st.arr)
        dt = time.perf_counter() - t0
        st = type('S', (), pipe.device_stats()[0])
        if best is None or dt < best[0]:
            best = (dt, st)
    dt, st = best
    print("template, chunk %5d x %d lanes: %6.0f k subframes/s = %5.1f GB/s H2D | wall %.1f ms; %d chunks: sum H2D %.1f ms (%.1f GB/s while copying), sum kernels %.1f ms, sum D2H %.1f ms (%.1f GB/s) | crc ok %d/%d"
          % (chunk, lanes, n / dt / 1e3, st.h2d_bytes / dt / 1e9, 1e3 * dt, st.chunks, 1e3 * st.h2d_s, st.h2d_bytes / st.h2d_s / 1e9, 1e3 * st.kernel_s, 1e3 * st.d2h_s,
             st.d2h_bytes / max(st.d2h_s, 1e-9) / 1e9, int((h_st.arr == 0).sum()), n * 9))
    pipe.close()
    h_out.free(); h_st.free()
