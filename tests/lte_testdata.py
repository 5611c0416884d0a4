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
