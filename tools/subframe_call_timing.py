"""Per-subframe host time of the three ways to decode a 20 MHz subframe from host sample buffers: the per-call sequence (as the shim
forwards it) under the fingerprint rule, the same under the explicit cache contract, and mi_lte_dl_subframe_decode_host."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import openlte_amd as m  # noqa: E402
from openlte_amd.lib import PdcchDci  # noqa: E402

gen = os.path.join(ROOT, "shim", "_build", "capture_gen")
for n_rb, fft, cell in ((100, 2048, 77), (25, 512, 301), (6, 128, 17)):
    frames = 10
    cap = os.path.join(tempfile.mkdtemp(), "c.bin")
    subprocess.run([gen, cap, str(n_rb), str(cell), str(frames)], check=True)
    raw = np.fromfile(cap, np.int8)
    per_sf = 15 * fft
    i_s = np.zeros(len(raw) // 2 + 4 * fft, np.float32); q_s = np.zeros_like(i_s)
    i_s[:len(raw) // 2], q_s[:len(raw) // 2] = raw[0::2], raw[1::2]
    ctx = m.Context(0)
    L = ctx.L
    f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
    u32 = C.c_uint32
    L.mi_lte_get_dl_subframe_and_ce_host.argtypes = [C.c_void_p, u32, u32, f32p, f32p] + [u32] * 4 + [f32p] * 4
    L.mi_lte_pdcch_channel_decode_host.argtypes = [C.c_void_p, u32, f32p, f32p, f32p, f32p, u32, u32, u32, C.c_float, u32, u32] + [C.POINTER(u32)] * 3 + [C.POINTER(PdcchDci)]
    L.mi_lte_pdsch_channel_decode_host.argtypes = [C.c_void_p, u32, f32p, f32p, f32p, f32p, u32, C.c_void_p, u32, u32, u32, np.ctypeslib.ndpointer(np.uint8), C.POINTER(u32)]
    L.mi_lte_dl_subframe_decode_host.argtypes = [C.c_void_p, u32, u32, f32p, f32p] + [u32] * 4 + [C.c_float, u32, u32] + [C.POINTER(u32)] * 3 + [
        C.POINTER(PdcchDci), np.ctypeslib.ndpointer(np.uint8), u32, C.POINTER(u32), C.POINTER(C.c_int32)]
    L.mi_lte_host_cache_set_mode.argtypes = [C.c_void_p, u32]
    sr, si = np.zeros((16, 1200), np.float32), np.zeros((16, 1200), np.float32)
    cr, ci = np.zeros((4, 16, 1200), np.float32), np.zeros((4, 16, 1200), np.float32)
    cfi, nsym, ndci = u32(), u32(), u32()
    dci = (PdcchDci * 6)()
    out, n = np.zeros(6 * 6144, np.uint8), (u32 * 6)()
    st = (C.c_int32 * 6)()
    subframes = [(f, sf) for f in range(frames - 1) for sf in range(10)]

    def per_call():
        blocks = 0
        for f, sf in subframes:
            L.mi_lte_get_dl_subframe_and_ce_host(ctx.h, fft, n_rb, i_s, q_s, f * 10 * per_sf, sf, cell, 1, sr, si, cr, ci)
            if 0 == L.mi_lte_pdcch_channel_decode_host(ctx.h, n_rb, sr, si, cr, ci, sf, cell, 1, 1.0, 0, 0, C.byref(cfi), C.byref(nsym), C.byref(ndci), dci):
                for k in range(ndci.value):
                    blocks += 0 == L.mi_lte_pdsch_channel_decode_host(ctx.h, n_rb, sr, si, cr, ci, sf, C.addressof(dci[k].alloc), nsym.value, cell, 1, out, n)
        return blocks

    def one_call():
        blocks = 0
        for f, sf in subframes:
            if 0 == L.mi_lte_dl_subframe_decode_host(ctx.h, fft, n_rb, i_s, q_s, f * 10 * per_sf, sf, cell, 1, 1.0, 0, 0, C.byref(cfi), C.byref(nsym), C.byref(ndci), dci,
                                                     out, 6144, n, st):
                blocks += sum(1 for k in range(ndci.value) if st[k] == 0)
        return blocks

    res = {}
    for name, mode, fn in (("per-call, fingerprint", 0, per_call), ("per-call, explicit contract", 1, per_call), ("one call per subframe", 0, one_call)):
        L.mi_lte_host_cache_set_mode(ctx.h, mode)
        fn()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter(); b = fn(); best = min(best, time.perf_counter() - t0)
        res[name] = (best / len(subframes) * 1e6, b)
    print("N_rb_dl %3d: " % n_rb + "; ".join("%s %.1f us/subframe (%d blocks)" % (k, v[0], v[1]) for k, v in res.items()))
