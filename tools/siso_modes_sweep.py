#!/usr/bin/env python3
"""Reference-faithful decoder: wall time per decode call over the number of code blocks in the call, with the trellis kernel that puts code
blocks on the lanes (k_turbo_siso) and with the one that puts the states there (k_turbo_siso_small) -- where the second stops paying
(mi_lte_set_turbo_small_batch's default)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openlte_amd as m
import openlte_amd.synth as synth

ctx = m.Context(0)
for K in (528, 6144):
    for n in (1, 9, 64, 512, 2048, 4096, 8192, 16384, 32768):
        _, soft = synth.turbo_soft_blocks(K, min(n, 64), flip=0.02, amp=24, seed=n)
        soft = np.ascontiguousarray(np.tile(soft, ((n + soft.shape[0] - 1) // soft.shape[0], 1))[:n])
        d_soft, d_out = ctx.to_device(soft), ctx.alloc(n * K)
        row, outs = [], []
        for small_max in (0, 1 << 30):
            ctx.set_turbo_small_batch(small_max)
            ctx.turbo_decode_dev(d_soft, m.SOFT_I8, K, n, d_out); ctx.sync()
            outs.append(d_out.download(np.uint8).copy())
            reps = 20 if n <= 2048 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.turbo_decode_dev(d_soft, m.SOFT_I8, K, n, d_out)
            ctx.sync()
            row.append((time.perf_counter() - t0) / reps * 1e3)
        same = bool((outs[0] == outs[1]).all())
        print("K %4d  %5d blocks: blocks on lanes %8.3f ms   states on lanes %8.3f ms   same bits: %s" % (K, n, row[0], row[1], same), flush=True)
        d_soft.free(); d_out.free()
