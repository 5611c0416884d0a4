"""Synthetic inputs for benchmarks and tests, produced by the library's host-side transmitter
(csrc/synth.cc); independent of the test-side CPU checkers."""
import ctypes as C

import numpy as np

from .lib import load_library, MiLteError

_i8p = np.ctypeslib.ndpointer(np.int8, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def _lib():
    L = load_library()
    if not getattr(L, "_synth_bound", False):
        L.mi_lte_synth_turbo_soft_i8.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_int, C.c_uint64, C.c_int, _i8p, _u8p]
        L.mi_lte_synth_turbo_soft_f32.argtypes = [C.c_uint32, C.c_uint32, C.c_double, C.c_uint64, C.c_int, _f32p, _u8p]
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
