#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

    python bench.py --gpus N --steps K --warmup W [--workload chain|frontend|turbo|uplink|control|sync]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM.  Work is sharded by unit (subframes / code blocks) over ranks with no data-path collective
(weak scaling: the per-GPU batch is fixed); torch.distributed is used only for the barrier and the
max-over-ranks of the elapsed time.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling


def dist_setup(n_gpus):
    """Returns (rank, world, barrier, max_reduce).  torch is imported only for multi-rank runs and
    BEFORE libmi_lte.so so that both share one HIP runtime."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1, (lambda: None), (lambda x: x)
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)  # control plane only

    def barrier():
        dist.barrier()

    def max_reduce(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    return rank, world, barrier, max_reduce


# ------------------------------------------------------------------------------------------------
DECODER = "ref"  # --decoder: "ref" = the reference-faithful decoder (parity mode), "bcjr" = max-log-MAP, 8 iterations


class TurboWorkload:
    """BASELINE config 3: PDSCH turbo decode, K=6144, 64QAM-like int8 soft values, 64k code blocks
    per GPU.  REF mode (bit-exact with the reference decoder) by default; --decoder bcjr runs the
    fixed-point max-log-MAP mode with 8 iterations on the same blocks."""
    name = "turbo"
    K = 6144
    unit = "Mbit/s"
    dtype = "i8 soft values, i16 path metrics (differences exact modulo 2^16)"
    alg_bytes_per_unit = 19216  # 3(K+4) int8 in + K/8 packed out + 4 B status, K=6144

    @property
    def metric(self):
        if DECODER == "bcjr":
            return "turbo-decode Mbit/s (K=6144 code blocks, 64QAM hard +-127 soft values, max-log-MAP BCJR, 8 iterations, per SURVEY 8d W3)"
        return "turbo-decode Mbit/s (K=6144 code blocks, 64QAM hard +-127 soft values, REF decoder, per SURVEY 8d W3)"

    @property
    def dominant(self):
        return "k_bcjr_bwd" if DECODER == "bcjr" else "k_turbo_siso"

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        self.ctx, self.m, self.np = ctx, m, np
        self.n_cb = n_units or 65536
        uniq = 64
        self.tx, soft = synth.turbo_soft_blocks(self.K, uniq, flip=0.02, seed=1234 + rank, ref_wrap=(DECODER != "bcjr"))
        self.uniq_soft = soft
        idx = (np.arange(self.n_cb) * 7 + np.arange(self.n_cb) // 64) % uniq
        self.d_in = ctx.to_device(soft[idx])
        self.d_out = ctx.alloc(self.n_cb * self.K)
        self.idx = idx

    def step(self):
        if DECODER == "bcjr":
            self.ctx.turbo_decode_dev(self.d_in, self.m.SOFT_I8, self.K, self.n_cb, self.d_out, mode=self.m.TURBO_BCJR, n_iter=8, qpp_spec=True)
        else:
            self.ctx.turbo_decode_dev(self.d_in, self.m.SOFT_I8, self.K, self.n_cb, self.d_out)

    def units_per_step(self):
        return self.n_cb

    def roofline_bytes(self, kernel, n_launch_per_step):
        return self.alg_bytes_per_unit * self.n_cb * n_launch_per_step

    def extra(self, value):
        if DECODER != "bcjr":
            return {}
        got = self.d_out.download(self.np.uint8, count=64 * self.K).reshape(64, self.K)
        return {"decoder": "max-log-MAP, 8 iterations, fixed point (specified by oracle/lte_oracle.c lo_turbo_decode_bcjr)",
                "sampled_blocks_equal_tx_bits": bool((got == self.tx[self.idx[:64]]).all())}

    def value_per_unit(self):
        return self.K / 1e6  # information Mbit per code block

    def config(self, world):
        return {"workload": "W3 turbo decode: K=6144 x %d code blocks per GPU, int8 soft in HBM, %s mode" % (self.n_cb, DECODER.upper()),
                "K": self.K, "blocks_per_gpu": self.n_cb,
                "decoder": "BCJR max-log-MAP x8" if DECODER == "bcjr" else "REF (reference-faithful, bit-exact)",
                "sharding": "code blocks block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=12.0):
        """Time the CPU reference on a bounded sample of the same blocks (rank 0, N=1 only)."""
        np = self.np
        from oracle import pyoracle
        K, D = self.K, self.K + 4
        if DECODER == "bcjr":  # the reference has no such decoder: the CPU leg is the plain-C specification of the mode
            P, s16 = pyoracle.port(), np.ascontiguousarray(self.uniq_soft.astype(np.int16))
            out, n, t0 = np.zeros(K, np.uint8), 0, time.perf_counter()
            while time.perf_counter() - t0 < budget_s:
                P.lo_turbo_decode_bcjr(s16[n % len(s16)], K, 8, 1, out)
                n += 1
            t = time.perf_counter() - t0
            return {"value": round(n * K / t / 1e6, 4), "unit": self.unit, "cores": 1, "kind": "port",
                    "sample": "%d of the benchmark's K=%d code blocks through oracle/lte_oracle.c lo_turbo_decode_bcjr (8 iterations), "
                              "1 thread, %.1f s" % (n, K, t)}
        soft_f = np.ascontiguousarray(self.uniq_soft.astype(np.float32))
        R = pyoracle.ref()
        if R is not None:
            kind, phy = "reference", R.ref_phy_new(4, 17, 1, 100)
            run = lambda n: R.ref_turbo_decode_batch(phy, np.ascontiguousarray(np.tile(soft_f, ((n + 63) // 64, 1))[:n]),
                                                     3 * D, n, np.zeros(n * K, np.uint8), K)
        else:
            kind, P = "port", pyoracle.port()
            run = lambda n: P.lo_time_turbo_decode_ref(np.ascontiguousarray(np.tile(soft_f, ((n + 63) // 64, 1))[:n]),
                                                       K, n, np.zeros(n * K, np.uint8))
        t = run(32)
        n = int(max(64, min(20000, budget_s / (t / 32))))
        t = run(n)
        return {"value": round(n * K / t / 1e6, 4), "unit": self.unit, "cores": 1, "kind": kind,
                "sample": "%d of the benchmark's K=%d code blocks, float soft values, 1 thread, %.1f s" % (n, K, t)}


def _turbo_alg_bytes(K):
    return 3 * (K + 4) + K // 8 + 4  # int8 soft in, packed bits + status out


class ChainWorkload:
    """BASELINE config 4 / SURVEY 8d W4 on one GPU per rank: 20 MHz, 100 RB, N_ant = 1, CFI = 2, every
    subframe fully loaded with 9 x 64QAM allocations (8 x 12 PRB TBS 3240 + 1 x 4 PRB TBS 1064, one
    code block each).  One step = FFT -> CE -> demap -> rate-unmatch -> turbo (REF) -> CRC over the
    whole batch of subframes, int8 IQ resident in HBM, decoded bits + status left in HBM."""
    name = "chain"
    metric = "DL subframes/sec @20MHz 100RB 64QAM, full chain FFT->CE->demap->rate-unmatch->turbo(REF)->CRC (SURVEY 8d W4)"
    unit = "subframes/s"
    dtype = "i8 IQ in, f32 FFT/CE/equaliser, i8 soft bits, i16 path metrics (differences exact modulo 2^16)"
    alg_bytes_per_unit = 70240 + 26984 // 8  # fused accounting, SURVEY 8d: int8 IQ in + packed info bits out
    dominant = "k_turbo_siso"
    info_bits = 8 * 3240 + 1064

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        import lte_testdata as td
        self.ctx, self.m, self.np = ctx, m, np
        self.n = n_units or 65536
        self.cfg = m.DlCfg(2048, 100, 1, 0)
        U = min(96, self.n)
        sfs = np.array([[1, 2, 3, 4, 6, 7, 8, 9][i % 8] for i in range(U)], np.uint32)  # subframes 0/5 carry sync signals
        cells = ((np.arange(U) * 37 + 11 * rank) % 504).astype(np.uint32)
        allocs = []
        for u in range(U):
            allocs += td.w4_allocs(u)
        iq, tx = synth.dl_units(self.cfg, sfs, cells, allocs, 9, snr_db=30.0, max_delay=8, seed=4242 + rank)
        self.uniq = (iq, tx, sfs, cells, allocs)
        idx = np.arange(self.n) % U
        self.idx = idx
        ul = iq.shape[1]
        self.d_iq = ctx.to_device(iq[idx].reshape(-1, 2))
        self.d_start = ctx.to_device((np.arange(self.n) * ul).astype(np.uint64))
        self.d_sf = ctx.to_device(sfs[idx])
        self.d_cell = ctx.to_device(cells[idx])
        self.d_sub = ctx.alloc(self.n * ctx.subframe_floats(1) * 4)
        all_allocs = []
        for i in range(self.n):
            all_allocs += td.w4_allocs(i)
        self.plan = ctx.pdsch_plan(self.cfg, 2, all_allocs)
        if DECODER == "bcjr":  # the max-log-MAP decoder instead of the reference's (8 iterations; the reference transmitter's interleaver)
            self.plan.set_decoder(m.TURBO_BCJR, 8, 0)
        self.d_out = ctx.alloc(self.n * 9 * self.plan.out_stride)
        self.d_status = ctx.alloc(self.n * 9 * 4)

    def step(self):
        self.ctx.dl_frontend_dev(self.cfg, self.d_iq, None, self.d_start, self.d_sf, self.d_cell, self.n, self.d_sub)
        self.plan.run_dev(self.d_sub, self.d_sf, self.d_cell, self.d_out, self.d_status)

    def units_per_step(self):
        return self.n

    def value_per_unit(self):
        return 1.0

    def extra(self, value):
        np = self.np
        st = self.d_status.download(np.int32)
        bits = self.d_out.download(np.uint8).reshape(self.n * 9, self.plan.out_stride)
        tx = self.uniq[1]
        ok = int((st == 0).sum())
        exact = all((bits[i * 9 + a, :(3240 if a < 8 else 1064)] == tx[self.idx[i], a, :(3240 if a < 8 else 1064)]).all()
                    for i in range(0, self.n, max(1, self.n // 64)) for a in range(9))
        # the same step with the samples coming from host memory and every verdict and bit going back (pageable buffers, one
        # stream, no overlap): what a caller holding host buffers sees -- reported next to the device-resident rate, never as `value`
        import time
        m = min(self.n, 4096)
        h_iq = np.ascontiguousarray(self.uniq[0][self.idx[:m]].reshape(-1, 2))  # the batch repeats every 96 subframes: one chunk serves
        t0 = time.perf_counter()
        for c in range(self.n // m):
            self.d_iq.upload(h_iq, offset=c * h_iq.nbytes)
        self.step()
        st2 = self.d_status.download(np.int32)
        bits2 = self.d_out.download(np.uint8)
        dt = time.perf_counter() - t0
        h2d, d2h = (self.n // m) * h_iq.nbytes, st2.nbytes + bits2.nbytes
        return {"turbo_info_mbit_per_s": round(value * self.info_bits / 1e6, 2),
                "crc_pass": "%d/%d allocations" % (ok, st.size), "sampled_blocks_equal_tx_bits": bool(exact),
                "from_host_buffers": {"subframes_per_s": round((self.n // m) * m / dt, 1),
                                      "note": "int8 IQ uploaded from pageable host memory (%.1f GB), one step, all verdicts and one-byte-per-bit "
                                              "outputs downloaded (%.1f GB), serial on one stream; PCIe-inclusive, not the headline value"
                                              % (h2d / 1e9, d2h / 1e9)}}

    def roofline_bytes(self, kernel, n_launch_per_step):
        """Algorithmic bytes the launches of `kernel` in ONE step account for (roofline table)."""
        n = self.n
        turbo = 8 * n * _turbo_alg_bytes(3264) + n * _turbo_alg_bytes(1088)
        return {"k_dl_fft": n * (70240 + 16 * 1200 * 8), "k_dl_ce": n * (5 * 1200 * 8 + 14 * 1200 * 8),
                "k_pdsch_demod": n * (8 * 1656 + 552) * (16 + 6),
                "k_turbo_siso": 2 * turbo, "k_turbo_prep": turbo, "k_turbo_perm": turbo, "k_turbo_vote": turbo,
                "k_rm_to_i8": turbo, "k_crc_finish": turbo, "k_bcjr_prep": turbo, "k_bcjr_fwd": 16 * turbo, "k_bcjr_bwd": 16 * turbo,
                "k_bcjr_perm": 16 * turbo}.get(kernel)

    def config(self, world):
        return {"workload": "W4 full DL chain: 20 MHz/100 RB/64QAM, 9 allocations per subframe (8x12 PRB TBS 3240 + 1x4 PRB "
                            "TBS 1064), %d subframes per GPU, int8 IQ in HBM" % self.n,
                "subframes_per_gpu": self.n, "N_ant": 1, "CFI": 2,
                "decoder": "BCJR (max-log-MAP, 8 iterations; specified by the plain-C model, not by the reference)" if DECODER == "bcjr" else "REF (reference-faithful, bit-exact)",
                "unique_subframes": len(self.uniq[2]),
                "sharding": "subframes block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=14.0):
        """Reference CPU path on a bounded sample of the same subframes (1 thread)."""
        import ctypes as C
        np = self.np
        from oracle import pyoracle as po
        import lte_testdata as td
        iq, tx, sfs, cells, allocs = self.uniq
        R, P = po.ref(), po.port()
        t_total, done, i = 0.0, 0, 0
        phy = R.ref_phy_new(4, 0, 1, 100) if R is not None else None
        sf_struct = R.ref_subframe_new() if R is not None else None
        while t_total < budget_s and done < 4000:
            u = i % len(sfs)
            i += 1
            sf, cell = int(sfs[u]), int(cells[u])
            re = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[u, :, 0].astype(np.float32)]))
            im = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[u, :, 1].astype(np.float32)]))
            out, nb = np.zeros(6200, np.uint8), C.c_uint32()
            t0 = time.perf_counter()
            if R is not None:
                R.ref_get_dl_subframe_and_ce(phy, re, im, 0, sf, cell, 1, sf_struct)
                for a in range(9):
                    la = td.to_lo_alloc(allocs[u * 9 + a])
                    rc = R.ref_pdsch_channel_decode(phy, sf_struct, C.byref(la), 2, cell, 1, out, C.byref(nb))
            else:
                lc = po.LoCfg()
                P.lo_cfg_init(C.byref(lc), 2048, 100)
                s = po.LoSubframe()
                P.lo_get_dl_subframe_and_ce(C.byref(lc), re, im, 0, sf, cell, 1, C.byref(s))
                for a in range(9):
                    la = td.to_lo_alloc(allocs[u * 9 + a])
                    rc = P.lo_pdsch_channel_decode(C.byref(lc), C.byref(s), C.byref(la), 2, cell, 1, out, C.byref(nb), None, None)
            t_total += time.perf_counter() - t0
            done += 1
        return {"value": round(done / t_total, 3), "unit": self.unit, "cores": 1,
                "kind": "reference" if R is not None else "port",
                "sample": "%d of the benchmark's subframes (get_dl_subframe_and_ce + 9 x pdsch_channel_decode each), 1 thread, "
                          "%.1f s; FFT = float64 radix-2 stand-in for FFTW3f" % (done, t_total)}


class FrontendWorkload:
    """BASELINE config 2 / SURVEY 8d W2: 20 MHz OFDM demod + CRS channel estimate, 10k subframe units."""
    name = "frontend"
    metric = "DL subframes/sec @20MHz 100RB, OFDM demod + CRS channel estimate only (SURVEY 8d W2)"
    unit = "subframes/s"
    dtype = "i8 IQ in, f32 FFT/CE"
    alg_bytes_per_unit = 339040
    dominant = "k_dl_ce"

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        import lte_testdata as td
        self.ctx, self.np = ctx, np
        self.n = n_units or 10000
        self.cfg = m.DlCfg(2048, 100, 1, 0)
        U = 64
        sfs = (np.arange(U) % 10).astype(np.uint32)
        cells = ((np.arange(U) * 37 + rank) % 504).astype(np.uint32)
        allocs = []
        for u in range(U):
            allocs += td.w4_allocs(u)
        iq, _ = synth.dl_units(self.cfg, sfs, cells, allocs, 9, snr_db=30.0, max_delay=8, seed=77 + rank)
        self.uniq = (iq, sfs, cells)
        idx = np.arange(self.n) % U
        ul = iq.shape[1]
        self.d_iq = ctx.to_device(iq[idx].reshape(-1, 2))
        self.d_start = ctx.to_device((np.arange(self.n) * ul).astype(np.uint64))
        self.d_sf, self.d_cell = ctx.to_device(sfs[idx]), ctx.to_device(cells[idx])
        self.d_sub = ctx.alloc(self.n * ctx.subframe_floats(1) * 4)

    def step(self):
        self.ctx.dl_frontend_dev(self.cfg, self.d_iq, None, self.d_start, self.d_sf, self.d_cell, self.n, self.d_sub)

    def units_per_step(self):
        return self.n

    def value_per_unit(self):
        return 1.0

    def roofline_bytes(self, kernel, n_launch_per_step):
        return {"k_dl_fft": self.n * (70240 + 16 * 1200 * 8), "k_dl_ce": self.n * (5 * 1200 * 8 + 14 * 1200 * 8)}.get(kernel)

    def config(self, world):
        return {"workload": "W2 front end: 20 MHz/100 RB, %d subframe units per GPU, int8 IQ in HBM" % self.n,
                "subframes_per_gpu": self.n, "N_ant": 1, "sharding": "subframes over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=10.0):
        np = self.np
        from oracle import pyoracle as po
        R = po.ref()
        iq, sfs, cells = self.uniq
        if R is None:
            return None
        phy, sfp = R.ref_phy_new(4, 0, 1, 100), R.ref_subframe_new()
        re = np.ascontiguousarray(iq[0, :, 0].astype(np.float32))
        im = np.ascontiguousarray(iq[0, :, 1].astype(np.float32))
        t = R.ref_time_get_dl_subframe_and_ce(phy, re, im, 0, 0, int(cells[0]), 1, sfp, 50)
        reps = int(max(100, budget_s / (t / 50)))
        t = R.ref_time_get_dl_subframe_and_ce(phy, re, im, 0, 0, int(cells[0]), 1, sfp, reps)
        return {"value": round(reps / t, 2), "unit": self.unit, "cores": 1, "kind": "reference",
                "sample": "%d calls of liblte_phy_get_dl_subframe_and_ce, 1 thread, %.1f s; FFT = float64 radix-2 stand-in" % (reps, t)}


class UplinkWorkload:
    """BASELINE config 5 / SURVEY 8d W5 (PUSCH part): eNodeB uplink, 20 MHz, 16 UEs per subframe, 6 PRB QPSK each
    (TBS 504, inside the reference's own envelope: E >= 3(K+4), N_prb <= 10 as its scheduler caps it,
    liblte_phy.cc:6426-6428).  One step = SC-FDMA demodulation -> per-UE DMRS estimate / equaliser / transform
    pre-decoding / de-map / descramble / de-interleave -> rate un-match -> REF turbo -> CRC."""
    name = "uplink"
    metric = "UL subframes/sec @20MHz, 16 UEs x 6 PRB QPSK PUSCH + 1 PRACH occasion per 10 subframes: SC-FDMA demod + PUSCH demod + UL-SCH turbo decode + PRACH correlation (SURVEY 8d W5)"
    unit = "subframes/s"
    dtype = "i8 IQ in, f32 FFT/CE/equaliser/DFT, i8 soft bits, i16 path metrics (differences exact modulo 2^16)"
    N_UE, N_PRB, TBS = 16, 6, 504
    alg_bytes_per_unit = 61440 + 16 * 504 // 8
    dominant = "k_pusch_demod"

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        self.ctx, self.m, self.np = ctx, m, np
        self.n = n_units or 16384
        self.cfg, self.ul = m.DlCfg(2048, 100, 1, 0), m.UlCfg(3, 0, 0, 2, 5)
        U = min(20, self.n)
        self.cell = 17 + rank
        sfs = (np.arange(U) % 10).astype(np.uint32)
        mk = lambda u: [m.make_alloc(u, 1, self.TBS, list(range(6 * a, 6 * a + 6)), 0x100 + a) for a in range(self.N_UE)]
        allocs = []
        for u in range(U):
            allocs += mk(u)
        iq, tx = synth.ul_units(self.cfg, self.ul, sfs, [self.cell] * U, allocs, self.N_UE, snr_db=20.0, max_delay=3, seed=777 + rank)
        self.uniq = (iq, tx, sfs, allocs)
        idx = np.arange(self.n) % U
        self.idx = idx
        ul_len = iq.shape[1]
        self.d_iq = ctx.to_device(iq[idx].reshape(-1, 2))
        self.d_start = ctx.to_device((np.arange(self.n) * ul_len).astype(np.uint64))
        self.d_sub = ctx.alloc(self.n * ctx.ul_subframe_floats() * 4)
        all_allocs = []
        for i in range(self.n):
            all_allocs += mk(i)
        self.plan = ctx.pusch_plan(self.cfg, self.ul, sfs[idx], [self.cell] * self.n, all_allocs)
        self.d_out = ctx.alloc(self.n * self.N_UE * self.plan.out_stride)
        self.d_status = ctx.alloc(self.n * self.N_UE * 4)
        # one PRACH occasion (format 0) per 10 subframes, a preamble in each (BASELINE config 5: "PRACH correlate + ...")
        self.pc = m.PrachCfg(22, 0, 11, 0, 4)
        self.n_occ = max(1, self.n // 10)
        PU = min(8, self.n_occ)
        self.pre = [(7 * k + 3) % 64 for k in range(PU)]
        piq = synth.prach_occasions(self.cfg, self.pc, self.pre, [16 * (k + 1) for k in range(PU)], snr_db=5.0, seed=31 + rank)
        self.pidx = np.arange(self.n_occ) % PU
        self.d_prach = ctx.to_device(piq[self.pidx].reshape(-1, 2))
        self.d_pstart = ctx.to_device((np.arange(self.n_occ) * piq.shape[1]).astype(np.uint64))
        self.pplan = ctx.prach_plan(self.cfg, self.pc)
        self.prach_last = None

    def step(self):
        self.ctx.ul_frontend_dev(self.cfg, self.d_iq, None, self.d_start, self.n, self.d_sub)
        self.plan.run_dev(self.d_sub, self.d_out, self.d_status)
        self.prach_last = self.pplan.detect_dev(self.d_prach, None, self.d_pstart, self.n_occ)

    def units_per_step(self):
        return self.n

    def value_per_unit(self):
        return 1.0

    def extra(self, value):
        np = self.np
        st = self.d_status.download(np.int32)
        bits = self.d_out.download(np.uint8).reshape(self.n * self.N_UE, self.plan.out_stride)
        tx = self.uniq[1]
        exact = all((bits[i * self.N_UE + a, :self.TBS] == tx[self.idx[i], a, :self.TBS]).all()
                    for i in range(0, self.n, max(1, self.n // 64)) for a in range(self.N_UE))
        nd, pr, _ = self.prach_last
        return {"turbo_info_mbit_per_s": round(value * self.N_UE * self.TBS / 1e6, 2),
                "crc_pass": "%d/%d allocations" % (int((st == 0).sum()), st.size), "sampled_blocks_equal_tx_bits": bool(exact),
                "prach": "%d/%d occasions: the transmitted preamble detected" % (int(((nd == 1) & (pr == np.array(self.pre)[self.pidx])).sum()), self.n_occ)}

    def roofline_bytes(self, kernel, n_launch_per_step):
        n, M = self.n, 12 * self.N_PRB
        turbo = n * self.N_UE * _turbo_alg_bytes(self.TBS + 24)
        return {"k_prach_bins": self.n_occ * (24576 * 2 + 839 * 8), "k_prach_corr": self.n_occ * (839 * 8 + 64 * 12),
                "k_ul_fft": n * (61440 + 14 * 1200 * 8), "k_pusch_demod": n * self.N_UE * (14 * M * 8 + 12 * M * 2),
                "k_turbo_siso": 2 * turbo, "k_turbo_prep": turbo, "k_turbo_perm": turbo, "k_turbo_vote": turbo}.get(kernel)

    def config(self, world):
        return {"workload": "W5 uplink: 20 MHz, %d UEs x %d PRB QPSK PUSCH (TBS %d) per subframe, %d subframes per GPU + %d PRACH occasions "
                            "(format 0, one per 10 subframes), int8 IQ in HBM" % (self.N_UE, self.N_PRB, self.TBS, self.n, self.n_occ),
                "subframes_per_gpu": self.n, "decoder": "REF (reference-faithful, bit-exact)", "unique_subframes": len(self.uniq[2]),
                "sharding": "subframes block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=12.0):
        """The reference's liblte_phy_get_ul_subframe + 16 x liblte_phy_pusch_channel_decode on one core."""
        import ctypes as C
        np = self.np
        from oracle import pyoracle as po
        R = po.ref()
        if R is None:
            return None
        iq, tx, sfs, allocs = self.uniq
        phy = R.ref_phy_new(4, self.cell, 1, 100)
        if R.ref_ul_init(phy, self.cell, 3, 0, 0, 2, 5) != 0:
            return None
        sfp = R.ref_subframe_new()
        R.ref_subframe_set_num(sfp, int(sfs[0]))
        la = (po.LoAlloc * self.N_UE)(*[po.make_alloc(1, self.TBS, list(range(6 * a, 6 * a + 6)), 0x100 + a) for a in range(self.N_UE)])
        re, im = np.ascontiguousarray(iq[0, :, 0].astype(np.float32)), np.ascontiguousarray(iq[0, :, 1].astype(np.float32))
        t = R.ref_time_pusch(phy, re, im, sfp, la, self.N_UE, self.cell, 3)
        reps = int(max(5, min(4000, budget_s / (t / 3))))
        t = R.ref_time_pusch(phy, re, im, sfp, la, self.N_UE, self.cell, reps)
        return {"value": round(reps / t, 3), "unit": self.unit, "cores": 1, "kind": "reference",
                "sample": "%d repetitions of one of the benchmark's subframes (get_ul_subframe + 16 x pusch_channel_decode), 1 thread, "
                          "%.1f s; FFT/DFT = float64 stand-in for FFTW3f (O(n^2) for the 72-point DFTs)" % (reps, t)}


class ControlWorkload:
    """the control region of every downlink subframe -- PCFICH + the PDCCH common search space (six candidates x
    DCI formats 1A / 1C: rate un-matching, K = 7 Viterbi, CRC16, RNTI test, DCI unpacking) at 20 MHz, 1 port, CFI 2, with an
    SI-RNTI and an RA-RNTI format-1A DCI per subframe.  One step = mi_lte_pdcch_decode_run over the batch of device
    subframes (what the front end leaves in HBM), results on the host."""
    name = "control"
    metric = "DL subframes/sec @20MHz, PCFICH + PDCCH common-search-space decode (6 candidates x DCI 1A/1C), liblte_phy_pdcch_channel_decode (SURVEY 8f N3)"
    unit = "subframes/s"
    dtype = "f32 combiner / de-mapper, i32 soft bits and path metrics"
    N_RE = 16 + 4 * 144  # distinct resource elements one subframe's decode reads (the two level-8 candidates re-read level-4 ones)
    alg_bytes_per_unit = N_RE * 16 + 112
    dominant = "k_pdcch_decode"

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        self.ctx, self.m, self.np = ctx, m, np
        self.n = n_units or 8192
        self.cfg = m.DlCfg(2048, 100, 1, 0)
        U = min(128, self.n)
        self.cell = 17 + rank
        rng = np.random.default_rng(99 + rank)
        sfs = (np.arange(U) % 10).astype(np.uint32)
        self.dcis = [[(0xFFFF, int(rng.integers(0, 27)), int(rng.integers(1, 9)), int(rng.integers(0, 90)), int(rng.integers(0, 4))),
                      (int(rng.integers(1, 0x3D)), int(rng.integers(0, 27)), int(rng.integers(1, 9)), int(rng.integers(0, 90)), 0)] for _ in range(U)]
        g = synth.ctrl_grids(self.cfg, sfs, [self.cell] * U, [2] * U, self.dcis, snr_db=15.0, seed=4242 + rank)
        self.uniq = (g, sfs)
        self.idx = np.arange(self.n) % U
        self.d_sub = ctx.to_device(g[self.idx])
        self.d_sf, self.d_cell = ctx.to_device(sfs[self.idx]), ctx.to_device(np.full(self.n, self.cell, np.uint32))
        self.plan = ctx.pdcch_plan(self.cfg, [self.cell], 1.0)
        self.last = None

    def step(self):
        self.last = self.plan.decode_raw(self.d_sub, self.d_sf, self.d_cell, self.n)

    def units_per_step(self):
        return self.n

    def value_per_unit(self):
        return 1.0

    def extra(self, value):
        rc, cfi, nsym, ndci, dci = self.last
        ok = sum(1 for i in range(self.n)
                 if {t[:4] for t in self.dcis[self.idx[i]]} <= {(dci[6 * i + k].alloc.rnti, dci[6 * i + k].mcs, dci[6 * i + k].alloc.N_prb,
                                                                 dci[6 * i + k].alloc.prb[0][0]) for k in range(int(ndci[i]))})
        return {"cfi_ok": "%d/%d" % (int((cfi == 2).sum()), self.n), "subframes_with_every_sent_dci_found": "%d/%d" % (ok, self.n),
                "dci_per_s": round(value * 2, 1)}

    def roofline_bytes(self, kernel, n_launch_per_step):
        return {"k_pdcch_decode": self.n * self.alg_bytes_per_unit}.get(kernel)

    def config(self, world):
        return {"workload": "N3 control region: 20 MHz, 1 port, CFI 2, SI-RNTI + RA-RNTI DCI 1A per subframe, %d device subframes per GPU "
                            "resident in HBM, results (CFI, DCIs, allocations) to the host" % self.n,
                "subframes_per_gpu": self.n, "unique_subframes": len(self.uniq[1]), "mode": "reference arithmetic (parity mode)",
                "sharding": "subframes block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=10.0):
        """The reference's liblte_phy_pdcch_channel_decode on one core over one of the benchmark's subframes."""
        np = self.np
        from oracle import pyoracle as po
        R = po.ref()
        if R is None:
            return None
        g, sfs = self.uniq
        phy = R.ref_phy_new(4, self.cell, 1, 100)
        sfp = R.ref_subframe_new()
        R.ref_subframe_set_num(sfp, int(sfs[0]))
        po.ref_subframe_view(R, sfp, 0)[:] = g[0, 0]
        po.ref_subframe_view(R, sfp, 1)[:] = g[0, 1]
        po.ref_subframe_view(R, sfp, 2, True)[0] = g[0, 2]
        po.ref_subframe_view(R, sfp, 3, True)[0] = g[0, 3]
        t = R.ref_time_pdcch(phy, sfp, self.cell, 1, 1.0, 20)
        reps = int(max(20, min(200000, budget_s / (t / 20))))
        t = R.ref_time_pdcch(phy, sfp, self.cell, 1, 1.0, reps)
        R.ref_subframe_free(sfp)
        R.ref_phy_free(phy)
        return {"value": round(reps / t, 3), "unit": self.unit, "cores": 1, "kind": "reference",
                "sample": "%d repetitions of liblte_phy_pdcch_channel_decode on one of the benchmark's subframes, 1 thread, %.1f s" % (reps, t)}


class SyncWorkload:
    """initial synchronisation of a 20 MHz capture -- the CP-autocorrelation coarse timing search over 160 slots
    (the dominant cost of scanning, 354 M complex MACs), the PSS search with fine timing (164 FFTs) and the SSS search.  One unit =
    one capture position searched (the three calls the scanner makes before it can read a subframe); captures are int8 in HBM."""
    name = "sync"
    metric = "cell searches/sec @20MHz: coarse timing over 160 slots + PSS/fine timing + SSS (liblte_phy_dl_find_coarse_timing_and_freq_offset, _find_pss_and_fine_timing, _find_sss; SURVEY 8f N4)"
    unit = "searches/s"
    dtype = "i8 IQ in, f32 correlations (reference summation order), f32 FFT"
    N_SLOTS = 160
    alg_bytes_per_unit = 161 * 15360 * 2 + 2208 + 15360 * 4
    dominant = "k_cp_corr"

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        self.ctx, self.m, self.np = ctx, m, np
        self.n = n_units or 32
        self.cfg = m.DlCfg(2048, 100, 1, m.IQ_I8)
        # a synthetic 20 MHz capture: the library's downlink subframes back to back (CRS + PDSCH give the cyclic-prefix structure
        # the coarse search looks for; PSS/SSS are not needed to time the searches)
        n_sf = 100
        cfg1 = m.DlCfg(2048, 100, 1, 0)
        allocs = []
        for u in range(n_sf):
            allocs += [m.make_alloc(u, 2, 1384, list(range(8 * a, 8 * a + 8)), 0x100 + a) for a in range(4)]
        iq, _ = synth.dl_units(cfg1, np.arange(n_sf) % 10, [17 + rank] * n_sf, allocs, 4, snr_db=20.0, max_delay=0, seed=5 + rank)
        cap = iq[:, :30720, :].reshape(-1, 2)
        self.cap = np.ascontiguousarray(cap)
        self.d_cap = ctx.to_device(self.cap)
        need = int(ctx.L.mi_lte_coarse_timing_samples(2048, self.N_SLOTS))
        self.starts = [int(x) for x in (np.arange(self.n) * 3001) % (len(cap) - need - 13 * 15360)]
        self.last = None

    def step(self):
        for st in self.starts:
            t = self.ctx.coarse_timing_dev(self.cfg, self.d_cap, None, self.N_SLOTS, start=st)
            ss, n2, ps, th, fo = self.ctx.find_pss_dev(self.cfg, self.d_cap, None, list(t.symb_starts[0]), start=st)
            self.last = (t.n_corr_peaks, n2, self.ctx.find_sss_dev(self.cfg, self.d_cap, None, n2, ss, th, start=st))

    def units_per_step(self):
        return self.n

    def value_per_unit(self):
        return 1.0

    def extra(self, value):
        return {"slots_searched_per_s": round(value * self.N_SLOTS, 1), "capture_seconds_per_second": round(value * self.N_SLOTS * 0.0005, 2),
                "last_search": "peaks=%d N_id_2=%d sss_found=%s" % (self.last[0], self.last[1], self.last[2][0])}

    def roofline_bytes(self, kernel, n_launch_per_step):
        n = self.n
        return {"k_cp_corr": n * (161 * 15360 * 2 + 160 * 15360 * 8), "k_cp_accum": n * (160 * 15360 * 8 + 15360 * 4), "k_cp_gather": n * 160 * 5 * 16,
                "k_sync_fft": n * 165 * (2048 * 2 + 1200 * 8), "k_seq_corr": n * (164 * 1200 * 8 + 1200 * 8)}.get(kernel)

    def config(self, world):
        return {"workload": "N4 initial sync: 20 MHz int8 capture resident in HBM, %d search positions per GPU per step, each = coarse timing over "
                            "160 slots + PSS search (84 + 80 FFTs) + SSS search" % self.n,
                "searches_per_gpu": self.n, "sharding": "search positions / captures over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=12.0):
        """The reference's liblte_phy_dl_find_coarse_timing_and_freq_offset (the dominant call) on one core."""
        np = self.np
        from oracle import pyoracle as po
        R = po.ref()
        if R is None:
            return None
        phy = R.ref_phy_new(4, 17, 1, 100)
        st = self.starts[0]
        i = np.ascontiguousarray(self.cap[st:, 0].astype(np.float32))
        q = np.ascontiguousarray(self.cap[st:, 1].astype(np.float32))
        t1 = R.ref_time_coarse_timing(phy, i, q, self.N_SLOTS, 1)
        reps = int(max(2, min(200, budget_s / t1)))
        t = R.ref_time_coarse_timing(phy, i, q, self.N_SLOTS, reps)
        R.ref_phy_free(phy)
        return {"value": round(reps / t, 3), "unit": self.unit, "cores": 1, "kind": "reference",
                "sample": "%d repetitions of the coarse-timing call alone (the PSS/SSS calls add ~10%% on the CPU), 1 thread, %.1f s" % (reps, t)}


class MultiStream:
    """Run S independent shards of a workload on S contexts (= S HIP streams) of the same GPU, launched
    back to back and synchronised together.  Units are independent, so this is the same "shard by
    unit, no exchange" split that is used across GPUs.  Default S = 1: with a batch large enough to give
    the lock-step trellis kernel 4+ waves per SIMD (32k subframes), one stream is the fastest."""

    def __init__(self, cls, ctxs, n_units, rank):
        n_units = n_units or {"chain": 65536, "frontend": 10000, "turbo": 65536, "uplink": 16384, "control": 8192, "sync": 32}[cls.name]
        per = max(64, (n_units // len(ctxs) + 63) // 64 * 64)
        self.parts = [cls(c, per, rank * 16 + k) for k, c in enumerate(ctxs)]
        self.ctxs = ctxs
        p0 = self.parts[0]
        self.name, self.metric, self.unit, self.dtype = p0.name, p0.metric, p0.unit, p0.dtype
        self.alg_bytes_per_unit, self.dominant = p0.alg_bytes_per_unit, p0.dominant

    def step(self):
        for p in self.parts:
            p.step()

    def sync(self):
        for c in self.ctxs:
            c.sync()

    def profile(self, on):
        for c in self.ctxs:
            c.profile(on)

    def profile_report(self):
        out = {}
        for c in self.ctxs:
            for k, (n, ms) in c.profile_report().items():
                a = out.get(k, (0, 0.0))
                out[k] = (a[0] + n, a[1] + ms)
        return out

    def units_per_step(self):
        return sum(p.units_per_step() for p in self.parts)

    def value_per_unit(self):
        return self.parts[0].value_per_unit()

    def roofline_bytes(self, kernel, n_launch_per_step):
        vals = [p.roofline_bytes(kernel, max(1, n_launch_per_step // len(self.parts))) for p in self.parts]
        return None if any(v is None for v in vals) else sum(vals)

    def config(self, world):
        c = self.parts[0].config(world)
        c["streams_per_gpu"] = len(self.parts)
        c["units_per_gpu_per_step"] = self.units_per_step()
        return c

    def extra(self, value):
        if not hasattr(self.parts[0], "extra"):
            return {}
        ex = [p.extra(value) for p in self.parts]
        out = dict(ex[0])
        if "crc_pass" in out:
            ok = sum(int(e["crc_pass"].split("/")[0]) for e in ex)
            tot = sum(int(e["crc_pass"].split("/")[1].split()[0]) for e in ex)
            out["crc_pass"] = "%d/%d allocations" % (ok, tot)
            out["sampled_blocks_equal_tx_bits"] = all(e["sampled_blocks_equal_tx_bits"] for e in ex)
        return out

    def cpu_baseline(self):
        return self.parts[0].cpu_baseline()


WORKLOADS = {"turbo": TurboWorkload, "frontend": FrontendWorkload, "chain": ChainWorkload, "uplink": UplinkWorkload, "control": ControlWorkload, "sync": SyncWorkload}


def pick_workload(name):
    if name != "auto":
        return WORKLOADS[name]
    return WORKLOADS["chain"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--units", type=int, default=0, help="units (subframes / code blocks) per GPU per step")
    ap.add_argument("--streams", type=int, default=1, help="independent shards (contexts/streams) per GPU")
    ap.add_argument("--decoder", default="ref", choices=["ref", "bcjr"], help="turbo workload only: decoder mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    global DECODER
    DECODER = args.decoder
    rank, world, barrier, max_reduce = dist_setup(args.gpus)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    import openlte_amd as m
    n_dev = max(1, m.load_library().mi_lte_device_count())
    # one GPU per rank; the modulo only matters when ranks outnumber GPUs (testing)
    ctxs = [m.Context(local_rank % n_dev) for _ in range(max(1, args.streams))]
    ctx = ctxs[0]
    wl = MultiStream(pick_workload(args.workload), ctxs, args.units, rank)

    for _ in range(args.warmup):
        wl.step()
    wl.sync()
    wl.profile(True)  # HIP events around every kernel launch, on the launch stream
    barrier()
    wl.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    wl.sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = max_reduce(t1 - t0)
    prof = wl.profile_report()
    wl.profile(False)

    if rank == 0:
        units = wl.units_per_step() * world * args.steps
        value = units * wl.value_per_unit() / elapsed
        dom = max(prof, key=lambda k: prof[k][1])  # the kernel with the largest share of the timed region
        n_launch, tot_ms = prof[dom]
        avg_ms = tot_ms / n_launch
        per_kernel = {}
        for k, (nl, ms) in prof.items():
            b = wl.roofline_bytes(k, nl // args.steps) if hasattr(wl, "roofline_bytes") else None
            if b is None:
                b = wl.alg_bytes_per_unit * wl.units_per_step() * (nl // args.steps)
            gbs = b * args.steps / (ms * 1e-3) / 1e9
            per_kernel[k] = {"ms_per_step": round(ms / args.steps, 4), "launches_per_step": nl // args.steps,
                             "alg_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / HBM_PEAK_GBS, 4)}
        alg_bytes = (wl.roofline_bytes(dom, n_launch // args.steps) if hasattr(wl, "roofline_bytes") else None) or \
            wl.alg_bytes_per_unit * wl.units_per_step() * (n_launch // args.steps)
        alg_bytes = alg_bytes / (n_launch // args.steps)  # per launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        try:  # HBM bytes per launch of the dominant kernel, from the committed rocprofv3 --pmc passes of this command
            tj = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic_%s.json" % wl.name)))
            traffic = tj["bytes_per_launch"].get({"k_ul_fft": "k_dl_fft"}.get(dom, dom))  # rocprof sees the kernel's own name
        except Exception:
            pass
        out = {
            "metric": wl.metric, "value": round(value, 3), "unit": wl.unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
            "config": wl.config(world),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": "profiles/pmc_traffic_%s.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)" % wl.name if traffic else None,
                         "avg_launch_ms": round(avg_ms, 4), "launches": n_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "algorithmic bytes / measured launch time; trellis kernels are issue-bound, see DESIGN.md"},
            "kernels": per_kernel,
            "whole_chain_alg_GBps": round(wl.alg_bytes_per_unit * units / elapsed / 1e9, 2),
            "device": ctx.device_name,
        }
        if hasattr(wl, "extra"):
            out.update(wl.extra(value))
        if world == 1 and not args.no_cpu_baseline:
            cb = wl.cpu_baseline()
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out))
    barrier()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
