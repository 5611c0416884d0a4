"""Seeded synthetic inputs shared by the tests (uses the oracle's encoder -- test side only)."""
import numpy as np


def turbo_blocks(port, K, n, kind, seed):
    """Return (tx_bits [n,K] uint8, soft [n,3(K+4)] in the reference's interleaved layout).

    kind: 'clean' +-1 floats, 'awgn0.5'/'awgn0.8' BPSK+noise floats, 'hard127' int8 +-127 with 2 % flips,
          'int' integer-valued floats with exact zeros, 'i16' repetition-combined int16 values.
    """
    rng = np.random.default_rng(seed)
    D = K + 4
    tx = rng.integers(0, 2, (n, K)).astype(np.uint8)
    soft = []
    for b in range(n):
        d = np.zeros(3 * D, np.uint8)
        port.lo_turbo_encode(np.ascontiguousarray(tx[b]), K, d)
        x = 1.0 - 2.0 * d.reshape(3, D).astype(np.float32)  # bit 0 -> +1
        if kind == "clean":
            y = x
        elif kind.startswith("awgn"):
            y = x + float(kind[4:]) * rng.standard_normal((3, D)).astype(np.float32)
        elif kind == "hard127":
            flip = rng.random((3, D)) < 0.02
            y = (127 * x * np.where(flip, -1, 1)).astype(np.int8)
        elif kind == "int":
            y = np.round((x + 0.7 * rng.standard_normal((3, D))) * 9).astype(np.float32)
        elif kind == "i16":
            reps = rng.integers(1, 5, (3, D))
            y = (127 * x * reps).astype(np.int16)
            y[rng.random((3, D)) < 0.03] *= -1
        else:
            raise ValueError(kind)
        soft.append(np.ascontiguousarray(y.T).reshape(-1))
    return tx, np.stack(soft)


def oracle_turbo_ref(port, soft, K):
    out = np.zeros((soft.shape[0], K), np.uint8)
    for b in range(soft.shape[0]):
        port.lo_turbo_decode_ref(np.ascontiguousarray(soft[b], dtype=np.float32), K, out[b])
    return out


# ---------------------------------------------------------------------------------------------------
# subframe units (uses the library's host-side transmitter, openlte_amd.synth)

def w4_allocs(unit, n_prb_total=100, mod=3):
    """8 x 12 PRB (TBS 3240) + 1 x 4 PRB (TBS 1064), 64QAM, rnti 0x100+a."""
    import openlte_amd as m
    out = []
    for a in range(9):
        prbs = list(range(a * 12, a * 12 + 12)) if a < 8 else list(range(96, 100))
        out.append(m.make_alloc(unit, mod, 3240 if a < 8 else 1064, prbs, 0x100 + a))
    return out


def small_allocs(unit, N_rb_dl, mod, tbs, n_prb, rnti=0x1234, first=0):
    import openlte_amd as m
    return [m.make_alloc(unit, mod, tbs, list(range(first, first + n_prb)), rnti)]


def to_lo_alloc(a):
    from oracle import pyoracle as po
    return po.make_alloc(a.mod_type, a.tbs, [a.prb[0][i] for i in range(a.N_prb)], a.rnti, a.rv_idx, a.tx_mode)


def oracle_frontend(port, fft, n_rb, n_ant, iq_unit, sf, cell):
    """Run the port oracle's get_dl_subframe_and_ce on one int8 unit; returns LoSubframe."""
    import ctypes as C
    from oracle import pyoracle as po
    lc = po.LoCfg()
    port.lo_cfg_init(C.byref(lc), fft, n_rb)
    per_sf = 30720 * fft // 2048
    i = np.concatenate([np.zeros(sf * per_sf, np.float32), iq_unit[:, 0].astype(np.float32)])
    q = np.concatenate([np.zeros(sf * per_sf, np.float32), iq_unit[:, 1].astype(np.float32)])
    s = po.LoSubframe()
    rc = port.lo_get_dl_subframe_and_ce(C.byref(lc), np.ascontiguousarray(i), np.ascontiguousarray(q), 0, sf, cell, n_ant,
                                        C.byref(s))
    assert rc == 0
    return lc, s
