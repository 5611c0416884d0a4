#!/usr/bin/env python3
"""Max-log-MAP, 8 iterations, K = 6144: wall time per decode call (host round trip included) of the batch kernels (MI_LTE_TURBO_BCJR) and of the
one-block-per-wavefront kernel (MI_LTE_TURBO_BCJR_BLOCK) over the number of code blocks in the call -- where the second stops paying."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openlte_amd as m
import openlte_amd.synth as synth

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6144
ctx = m.Context(0)
for n in (1, 9, 64, 256, 512, 1024, 2048, 4096, 8192):
    _, soft = synth.turbo_soft_blocks(K, min(n, 64), flip=0.02, amp=24, seed=n)
    soft = np.ascontiguousarray(np.tile(soft, ((n + soft.shape[0] - 1) // soft.shape[0], 1))[:n])
    d_soft, d_out = ctx.to_device(soft), ctx.alloc(n * K)
    row = []
    for mode in (m.TURBO_BCJR, m.TURBO_BCJR_BLOCK):
        ctx.turbo_decode_dev(d_soft, m.SOFT_I8, K, n, d_out, mode=mode, n_iter=8); ctx.sync()
        reps = 10 if n <= 1024 else 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.turbo_decode_dev(d_soft, m.SOFT_I8, K, n, d_out, mode=mode, n_iter=8)
        ctx.sync()
        row.append((time.perf_counter() - t0) / reps * 1e3)
    print("K %d  %5d blocks: batch kernels %8.3f ms   one block per wavefront %8.3f ms   (%.1f / %.1f Mbit/s)" % (K, n, row[0], row[1], n * K / row[0] / 1e3, n * K / row[1] / 1e3), flush=True)
    d_soft.free(); d_out.free()
