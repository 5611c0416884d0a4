"""Synthetic inputs for benchmarks and tests, produced by the library's host-side transmitter
(csrc/synth.cc); independent of the test-side CPU checkers."""
import ctypes as C

import numpy as np

from .lib import load_library, MiLteError, DlCfg, UlCfg, PrachCfg, PdschAlloc

_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


class SynthChannel(C.Structure):
    """mi_lte_synth_channel"""
    _fields_ = [("gain_min", C.c_double), ("gain_max", C.c_double), ("max_delay", C.c_double), ("snr_db", C.c_double),
                ("peak", C.c_double), ("seed", C.c_uint64)]


def _lib():
    L = load_library()
    if not getattr(L, "_synth_bound", False):
        L.mi_lte_synth_turbo_soft_i8.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_uint64, C.c_int, _i8p, _u8p]
        L.mi_lte_synth_turbo_soft_f32.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_uint64, C.c_int, _f32p, _u8p]
        L.mi_lte_synth_unit_len.argtypes = [C.c_uint32]
        L.mi_lte_synth_unit_len.restype = C.c_size_t
        L.mi_lte_synth_dl_units_i8.argtypes = [C.POINTER(DlCfg), C.c_uint32, _u32p, _u32p, C.c_uint32, C.c_void_p, C.c_uint32,
                                               C.POINTER(SynthChannel), _i8p, _u8p, C.c_uint32]
        L.mi_lte_synth_ul_unit_len.argtypes = [C.c_uint32]
        L.mi_lte_synth_ul_unit_len.restype = C.c_size_t
        L.mi_lte_synth_ul_units_i8.argtypes = [C.POINTER(DlCfg), C.POINTER(UlCfg), C.c_uint32, _u32p, _u32p, C.c_void_p, C.c_uint32,
                                               C.POINTER(SynthChannel), _i8p, _u8p, C.c_uint32]
        L.mi_lte_synth_prach_len.argtypes = [C.c_uint32, C.c_uint32]
        L.mi_lte_synth_prach_len.restype = C.c_size_t
        L.mi_lte_synth_prach_i8.argtypes = [C.POINTER(DlCfg), C.POINTER(PrachCfg), C.c_uint32, _u32p, _u32p, C.POINTER(SynthChannel), _i8p]
        L.mi_lte_synth_ctrl_grids.argtypes = [C.POINTER(DlCfg), C.c_float, C.c_uint32, _u32p, _u32p, _u32p, _u32p, C.c_uint32,
                                              C.POINTER(SynthChannel), _f32p]
        L._synth_bound = True
    return L


def turbo_soft_blocks(K, n, flip=0.02, amp=127, seed=1, ref_wrap=True):
    """(tx_bits [n,K] uint8, soft [n,3(K+4)] int8): turbo-encoded random blocks as hard +-amp soft values."""
    soft = np.zeros((n, 3 * (K + 4)), np.int8)
    tx = np.zeros((n, K), np.uint8)
    rc = _lib().mi_lte_synth_turbo_soft_i8(K, n, float(flip), int(amp), int(seed), 1 if ref_wrap else 0, soft, tx)
    if rc != 0:
        raise MiLteError("mi_lte_synth_turbo_soft_i8 failed: %d" % rc)
    return tx, soft


def turbo_soft_blocks_awgn(K, n, sigma=0.5, seed=1, ref_wrap=True):
    soft = np.zeros((n, 3 * (K + 4)), np.float32)
    tx = np.zeros((n, K), np.uint8)
    rc = _lib().mi_lte_synth_turbo_soft_f32(K, n, float(sigma), int(seed), 1 if ref_wrap else 0, soft, tx)
    if rc != 0:
        raise MiLteError("mi_lte_synth_turbo_soft_f32 failed: %d" % rc)
    return tx, soft


def unit_len(fft_size=2048):
    return int(_lib().mi_lte_synth_unit_len(fft_size))


def dl_units(cfg, subfr_num, n_id_cell, allocs, n_alloc, n_pdcch_symbs=2, gain=(0.5, 1.5), max_delay=8, snr_db=30.0,
             peak=100.0, seed=1):
    """Synthesise len(subfr_num) single-port subframe units.

    allocs: list of PdschAlloc, n_alloc per unit (unit-major).  Returns (iq int8 [n, unit_len, 2],
    tx_bits uint8 [n, n_alloc, max_tbs])."""
    n = len(subfr_num)
    ul = unit_len(cfg.fft_size)
    iq = np.zeros((n, ul, 2), np.int8)
    max_tbs = max([a.tbs for a in allocs], default=8)
    tx = np.zeros((n, max(n_alloc, 1), max_tbs), np.uint8)
    arr = (PdschAlloc * max(len(allocs), 1))(*allocs)
    ch = SynthChannel(gain[0], gain[1], float(max_delay), float(snr_db), float(peak), int(seed))
    rc = _lib().mi_lte_synth_dl_units_i8(C.byref(cfg), n, np.ascontiguousarray(subfr_num, np.uint32),
                                         np.ascontiguousarray(n_id_cell, np.uint32), n_pdcch_symbs,
                                         C.cast(arr, C.c_void_p), n_alloc, C.byref(ch), iq, tx, max_tbs)
    if rc != 0:
        raise MiLteError("mi_lte_synth_dl_units_i8 failed: %d" % rc)
    return iq, tx


def ul_unit_len(fft_size=2048):
    return int(_lib().mi_lte_synth_ul_unit_len(fft_size))


def ul_units(cfg, ulcfg, subfr_num, n_id_cell, allocs, n_alloc, gain=(0.5, 1.5), max_delay=4, snr_db=30.0, peak=100.0, seed=1):
    """Synthesise len(subfr_num) uplink subframe units, n_alloc PUSCH transmissions each (allocs unit-major).
    Returns (iq int8 [n, ul_unit_len, 2], tx_bits uint8 [n, n_alloc, max_tbs])."""
    n = len(subfr_num)
    ul = ul_unit_len(cfg.fft_size)
    iq = np.zeros((n, ul, 2), np.int8)
    max_tbs = max([a.tbs for a in allocs], default=8)
    tx = np.zeros((n, max(n_alloc, 1), max_tbs), np.uint8)
    arr = (PdschAlloc * max(len(allocs), 1))(*allocs)
    ch = SynthChannel(gain[0], gain[1], float(max_delay), float(snr_db), float(peak), int(seed))
    rc = _lib().mi_lte_synth_ul_units_i8(C.byref(cfg), C.byref(ulcfg), n, np.ascontiguousarray(subfr_num, np.uint32),
                                         np.ascontiguousarray(n_id_cell, np.uint32), C.cast(arr, C.c_void_p), n_alloc,
                                         C.byref(ch), iq, tx, max_tbs)
    if rc != 0:
        raise MiLteError("mi_lte_synth_ul_units_i8 failed: %d" % rc)
    return iq, tx


def prach_occasions(cfg, prach_cfg, preamble_idx, delay, gain=(0.5, 1.5), snr_db=20.0, peak=100.0, seed=1):
    """int8 [n_occ, prach_len, 2]: one PRACH preamble per occasion (index preamble_idx[o] of the cell's 64, delayed by delay[o] samples)."""
    n = len(preamble_idx)
    ln = int(_lib().mi_lte_synth_prach_len(cfg.fft_size, prach_cfg.preamble_format))
    iq = np.zeros((n, ln, 2), np.int8)
    ch = SynthChannel(gain[0], gain[1], 0.0, float(snr_db), float(peak), int(seed))
    rc = _lib().mi_lte_synth_prach_i8(C.byref(cfg), C.byref(prach_cfg), n, np.ascontiguousarray(preamble_idx, np.uint32),
                                      np.ascontiguousarray(delay, np.uint32), C.byref(ch), iq)
    if rc != 0:
        raise MiLteError("mi_lte_synth_prach_i8 failed: %d" % rc)
    return iq


def ctrl_grids(cfg, subfr_num, n_id_cell, cfi, dcis, phich_res=1.0, gain=(0.6, 1.4), snr_db=10.0, seed=1):
    """Control regions as device-subframe grids, float32 [n, 2 + 2*N_ant, 16, 1200] (symbols 0-3 filled): PCFICH + the format-1A
    DCIs dcis[u] = [(rnti, mcs, N_prb, rb_start, rv_idx), ...] (at most 4, candidate = list position) with standard transmit
    diversity on cfg.N_ant ports."""
    n = len(subfr_num)
    n_dci = max([len(d) for d in dcis] + [1])
    tab = np.zeros((n, n_dci, 5), np.uint32)
    for u, lst in enumerate(dcis):
        for a, t in enumerate(lst):
            tab[u, a] = t
    g = np.zeros((n, 2 + 2 * cfg.N_ant, 16, 1200), np.float32)
    ch = SynthChannel(gain[0], gain[1], 0.0, float(snr_db), 0.0, int(seed))
    rc = _lib().mi_lte_synth_ctrl_grids(C.byref(cfg), float(phich_res), n, np.ascontiguousarray(subfr_num, np.uint32),
                                        np.ascontiguousarray(n_id_cell, np.uint32), np.ascontiguousarray(cfi, np.uint32), tab.reshape(-1), n_dci,
                                        C.byref(ch), g.reshape(-1))
    if rc != 0:
        raise MiLteError("mi_lte_synth_ctrl_grids failed: %d" % rc)
    return g
