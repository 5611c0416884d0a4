#!/usr/bin/env python3
"""Soak: the reference-faithful decoder's two trellis kernels against each other on random block sizes, batch sizes and inputs (no oracle:
device against device; the oracle comparisons are tests/test_turbo_gpu.py).  Prints the first disagreement, or the count of decodes compared."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import openlte_amd as m
import lte_testdata as td

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
rng = np.random.default_rng(2026)
ctx = m.Context(0)
t0, n_dec, n_blocks = time.time(), 0, 0
while time.time() - t0 < budget:
    K = int(rng.choice(td.ALL_K))
    n = int(rng.choice([1, 2, 3, 5, 8, 9, 17, 64, 65, 130, 300, 2100, 4500]))
    kind = str(rng.choice(["noise", "rand127", "small", "sparse"]))
    D = K + 4
    if kind == "noise":
        soft = rng.integers(-127, 128, (n, 3 * D)).astype(np.int8)
    elif kind == "rand127":
        soft = (127 * (1 - 2 * rng.integers(0, 2, (n, 3 * D)))).astype(np.int8)
    elif kind == "small":
        soft = rng.integers(-3, 4, (n, 3 * D)).astype(np.int8)          # many ties and zeros
    else:
        soft = (rng.integers(-127, 128, (n, 3 * D)) * (rng.random((n, 3 * D)) < 0.3)).astype(np.int8)
    ctx.set_turbo_small_batch(0)
    a = ctx.turbo_decode(soft, K)
    ctx.set_turbo_small_batch(1 << 30)
    b = ctx.turbo_decode(soft, K)
    n_dec += 1; n_blocks += n
    if not (a == b).all():
        bad = np.nonzero((a != b).any(axis=1))[0]
        print("DISAGREE: K", K, "n", n, kind, "blocks", bad[:8]); sys.exit(1)
print("%d decodes (%d code blocks, %d block sizes drawn from all 188) agree bit for bit between k_turbo_siso and k_turbo_siso_small" % (n_dec, n_blocks, len(td.ALL_K)))
