import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lte_testdata as td, openlte_amd as m
from openlte_amd import synth
cfg = m.DlCfg(2048, 100, 1, m.IQ_I8 | m.CE_COMPACT)
U, n = 32, 32768
sfs = np.array([[1, 2, 3, 4, 6, 7, 8, 9][i % 8] for i in range(U)], np.uint32)
cells = ((np.arange(U) * 37) % 504).astype(np.uint32)
allocs = []
for u in range(U): allocs += td.w4_allocs(u)
iq, tx = synth.dl_units(cfg, sfs, cells, allocs, 9, snr_db=30.0, max_delay=8, seed=1)
ul = iq.shape[1]
h_iq, h_sf, h_cell = m.HostBuffer((n, ul, 2), np.int8), m.HostBuffer((n,), np.uint32), m.HostBuffer((n,), np.uint32)
for c0 in range(0, n, U): h_iq.arr[c0:c0 + U] = iq
h_sf.arr[:], h_cell.arr[:] = sfs[np.arange(n) % U], cells[np.arange(n) % U]
chunk, lanes = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3:  # another pipeline first, in the same process
    p0 = m.DlPipeline(0, cfg, 2, td.w4_allocs(0), chunk, int(sys.argv[3]))
    o0, s0 = m.HostBuffer((n * 9, p0.out_stride), np.uint8), m.HostBuffer((n * 9,), np.int32)
    p0.run(h_iq.arr, h_sf.arr, h_cell.arr, n, o0.arr, s0.arr)
    p0.close(); o0.free(); s0.free()
pipe = m.DlPipeline(0, cfg, 2, td.w4_allocs(0), chunk, lanes)
h_out, h_st = m.HostBuffer((n * 9, pipe.out_stride), np.uint8), m.HostBuffer((n * 9,), np.int32)
os.environ.pop("MI_LTE_PIPELINE_TRACE", None)
for _ in range(2): pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n, h_out.arr, h_st.arr)
os.environ["MI_LTE_PIPELINE_TRACE"] = "1"
t0 = time.perf_counter(); pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n, h_out.arr, h_st.arr); print("wall %.1f ms" % (1e3 * (time.perf_counter() - t0)))
