"""Junk-argument runs of the library's host-only entry points (no GPU): the DCI unpackers, the control-region tables, the DMRS generator
and the five input generators of the synth API with bandwidths, transform lengths, port counts, allocations, PRACH configurations and
PHICH resources far outside LTE.  Nothing is checked but survival -- tests/test_host_robustness_cpu.py runs each family in a process of its
own and requires a clean exit (a zeroCorrelationZoneConfig past the N_cs table once divided by zero in the PRACH generator; a negative
PHICH resource sized a table by a wrapped count; fft_size = 0 divided by zero in three generators).

    python tests/host_fuzz.py <seed> <dci|tables|dmrs|dl|ul|prach|ctrl|turbo> [scale]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import openlte_amd as m  # noqa: E402
from openlte_amd import lib as ml, synth  # noqa: E402

L = m.load_library()
rng = np.random.default_rng(int(sys.argv[1]))
which = sys.argv[2]
scale = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
n = ok = 0


def reps(k):
    return max(1, int(k * scale))

def junk_alloc(valid_fft_nrb):
    a = m.PdschAlloc()
    a.unit = 0
    a.mod_type = int(rng.choice([0, 1, 2, 3, 4, 255]))
    a.tbs = int(rng.choice([0, 8, 16, 40, 1064, 3240, 6120, 6121, 6144, 20000, 2**31]))
    a.rv_idx = int(rng.integers(0, 6)); a.tx_mode = int(rng.integers(0, 5)); a.rnti = int(rng.integers(0, 70000))
    a.N_prb = int(rng.choice([0, 1, 2, 6, 12, 100, 110, 111, 112, 113, 200, 2**31]))
    for s in range(2):
        for i in range(112):
            a.prb[s][i] = int(rng.integers(0, 256))
    return a

if which == "dci":
    for _ in range(reps(200000)):
        fmt = int(rng.integers(2)); nrb = int(rng.choice([0, 1, 5, 6, 7, 15, 25, 50, 75, 100, 101, 110, 200, 65535, 2**31]))
        nb = int(rng.choice([0, 1, 8, 12, 13, 15, 21, 22, 25, 27, 28, 31, 32, 33, 64]))
        rc, d = ml.dci_unpack(fmt, int(rng.integers(2**32)), nb, int(rng.integers(65536)), nrb, int(rng.choice([0,1,2,3,4,5])))
        n += 1
elif which == "tables":
    for _ in range(reps(20000)):
        nrb = int(rng.choice([0, 1, 5, 6, 7, 15, 25, 50, 75, 100, 101, 110, 200, 65535]))
        pc, cand = np.zeros(16, np.uint32), np.zeros((6, 288), np.uint32)
        rc = L.mi_lte_pdcch_re_tables(nrb, int(rng.choice([0,1,2,3,4,8])), int(rng.integers(0, 600)), float(rng.choice([0.0, 1/6, 0.5, 1.0, 2.0, 3.0, -1.0])), int(rng.integers(0, 6)), pc, cand)
        n += 1
elif which == "dmrs":
    for _ in range(reps(20000)):
        ul = m.UlCfg(int(rng.integers(0, 40)), int(rng.integers(0,3)), int(rng.integers(0,3)), int(rng.integers(0, 10)), int(rng.integers(0, 10)))
        nprb = int(rng.choice([0, 1, 2, 3, 5, 7, 11, 13, 100, 110, 111, 200]))
        out = np.zeros((4, 12 * max(nprb,1)), np.float32)
        rc = L.mi_lte_ul_dmrs_pusch(C.byref(ul), int(rng.integers(0, 600)), int(rng.integers(0, 12)), nprb, out[0], out[1], out[2], out[3])
        n += 1
elif which == "dl":
    for _ in range(reps(3000)):
        cfg = m.DlCfg(int(rng.choice([0, 64, 128, 256, 512, 1000, 1024, 2048, 4096])), int(rng.choice([0, 5, 6, 15, 25, 50, 75, 100, 101, 110, 200])), int(rng.choice([0, 1, 2, 3, 4])), int(rng.choice([0, 1, 0x100, 0x200, 7])))
        if cfg.fft_size not in (128, 256, 512, 1024, 2048):
            # unit_len of a bad fft is the caller's problem: size the buffer for the largest
            pass
        al = [junk_alloc(0) for _ in range(2)]
        try:
            ul = synth.unit_len(cfg.fft_size) if cfg.fft_size in (128, 256, 512, 1024, 2048) else 35200
            iq = np.zeros((1, max(ul, 35200), 2), np.int8)
            tx = np.zeros((1, 2, 6144), np.uint8)
            arr = (m.PdschAlloc * 2)(*al)
            ch = synth.SynthChannel(0.5, 1.5, float(rng.choice([-1, 0, 8, 1e9])), float(rng.choice([-100, 0, 30, 400])), 100.0, 1)
            rc = synth._lib().mi_lte_synth_dl_units_i8(C.byref(cfg), 1, np.array([int(rng.integers(0, 12))], np.uint32), np.array([int(rng.integers(0, 600))], np.uint32), int(rng.integers(0, 6)), C.cast(arr, C.c_void_p), 2, C.byref(ch), iq, tx, 6144)
            ok += rc == 0
        except m.MiLteError:
            pass
        n += 1
elif which == "ul":
    for _ in range(reps(3000)):
        cfg = m.DlCfg(int(rng.choice([0, 128, 256, 512, 1000, 1024, 2048, 4096])), int(rng.choice([0, 5, 6, 15, 25, 50, 75, 100, 101, 110, 200])), 1, 0)
        ulc = m.UlCfg(int(rng.integers(0, 40)), int(rng.integers(0,3)), int(rng.integers(0,3)), int(rng.integers(0, 10)), int(rng.integers(0, 10)))
        al = [junk_alloc(0) for _ in range(2)]
        iq = np.zeros((1, 40000, 2), np.int8); tx = np.zeros((1, 2, 6144), np.uint8)
        arr = (m.PdschAlloc * 2)(*al)
        ch = synth.SynthChannel(0.5, 1.5, 4.0, 30.0, 100.0, 1)
        rc = synth._lib().mi_lte_synth_ul_units_i8(C.byref(cfg), C.byref(ulc), 1, np.array([int(rng.integers(0, 12))], np.uint32), np.array([int(rng.integers(0, 600))], np.uint32), C.cast(arr, C.c_void_p), 2, C.byref(ch), iq, tx, 6144)
        ok += rc == 0; n += 1
elif which == "prach":
    for _ in range(reps(3000)):
        cfg = m.DlCfg(int(rng.choice([0, 128, 256, 512, 1000, 1024, 2048, 4096])), int(rng.choice([0, 5, 6, 15, 25, 50, 75, 100, 101, 200])), 1, 0)
        pc = m.PrachCfg(int(rng.choice([0, 1, 400, 837, 838, 2**31])), int(rng.integers(0, 6)), int(rng.integers(0, 18)), int(rng.integers(0, 3)), int(rng.choice([0, 1, 5, 94, 95, 200, 2**31])))
        iq = np.zeros((1, 80000, 2), np.int8)
        ch = synth.SynthChannel(0.5, 1.5, 0.0, 10.0, 100.0, 1)
        rc = synth._lib().mi_lte_synth_prach_i8(C.byref(cfg), C.byref(pc), 1, np.array([int(rng.integers(0, 100))], np.uint32), np.array([int(rng.choice([0, 5, 100000, 2**31]))], np.uint32), C.byref(ch), iq)
        ok += rc == 0; n += 1
elif which == "ctrl":
    for _ in range(reps(3000)):
        cfg = m.DlCfg(int(rng.choice([0, 128, 256, 512, 1024, 2048])), int(rng.choice([0, 6, 7, 15, 25, 50, 75, 100, 101])), int(rng.choice([0, 1, 2, 3, 4])), 0)
        g = np.zeros((1, 10, 16, 1200), np.float32)
        tab = rng.integers(0, 2**32, (1, 4, 5), dtype=np.uint64).astype(np.uint32)
        ch = synth.SynthChannel(0.6, 1.4, 0.0, 10.0, 0.0, 1)
        rc = synth._lib().mi_lte_synth_ctrl_grids(C.byref(cfg), float(rng.choice([-1, 0, 1/6, 0.5, 1, 2, 9])), 1, np.array([int(rng.integers(0, 12))], np.uint32), np.array([int(rng.integers(0, 600))], np.uint32), np.array([int(rng.integers(0, 6))], np.uint32), tab.reshape(-1), 4, C.byref(ch), g.reshape(-1))
        ok += rc == 0; n += 1
elif which == "turbo":
    for _ in range(reps(300)):
        K = int(rng.choice([0, 8, 39, 40, 41, 6144, 6145, 6200, 100000]))
        tx = np.zeros((2, 7000), np.uint8); soft = np.zeros((2, 3 * 7004), np.int8)
        rc = synth._lib().mi_lte_synth_turbo_soft_i8(K, 2 if K <= 6144 else 0, 0.02, 127, 1, 1, soft.reshape(-1), tx.reshape(-1))
        ok += rc == 0; n += 1
print(which, n, ok, "survived")
