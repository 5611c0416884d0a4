#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

    python bench.py --gpus N --steps K --warmup W [--workload chain|frontend|turbo|uplink|control|sync]

A "step" is one pass of the hot path over one batch of synthetic input that is already resident in
HBM.  Work is sharded by unit (subframes / code blocks) over ranks with no data-path collective
(weak scaling: the per-GPU batch is fixed); torch.distributed (gloo, CPU tensors) is used only for the
barrier and the max-over-ranks of the elapsed time.

`--gpus N` with N > 1 and no WORLD_SIZE in the environment re-executes this script under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU; under an
external launcher (the driver's torchrun line) WORLD_SIZE must equal N.  Every rank must own a distinct gfx950 device.

Roofline accounting, stated once:
  * a STAGE's algorithmic bytes are SURVEY 8d's figure for it, charged ONCE per step against the summed time of the stage's kernels;
  * a KERNEL's "own_io" is the minimal bytes that kernel has to read and write at its own interface as the data is laid out;
  * the headline `roofline` is the kernel with the largest share of the timed region, priced at its STAGE's bytes (charged once)
    over that kernel's time per step -- never a multiple of the stage bytes.
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


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def maybe_relaunch(n_gpus, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: become N ranks.  Returns only in the single-process case or inside a rank."""
    if n_gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def dist_setup(n_gpus):
    """Returns (rank, world, barrier, max_reduce).  torch is imported only for multi-rank runs and
    BEFORE libmi_lte.so so that both share one HIP runtime.  Fails loudly when the launcher's world size is not --gpus."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != max(1, n_gpus):
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; refusing to report a wrong n_gpus" % (n_gpus, world))
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


def timing_ref(po):
    """The compiled reference a cpu_baseline leg TIMES: the build against the single-precision FFT stand-in (oracle/ref/fftw_shim_f32.c:
    Stockham radix 4 / mixed radix, what FFTW3f would pick, scalar) where it exists, else the parity build (float64 radix-2, slower than
    the FFTW3f the reference really links).  Returns (library or None, the note for the sample string)."""
    R = po.ref_f32fft()
    if R is not None:
        return R, "FFT = single-precision Stockham radix-4 / mixed-radix stand-in for FFTW3f (scalar C, gcc -O3)"
    return po.ref(), "FFT = float64 radix-2 stand-in for FFTW3f (pessimistic for the transform's share)"


def cpu_info():
    """(model string, logical CPUs this process may run on)."""
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return model, n


def run_threads(workers):
    """workers: callables that do their (untimed) set-up -- allocations are first touched by the thread that uses them -- and return
    the callable to time.  All timed callables are released together; each spends its time inside a C loop (the interpreter lock is
    released).  Returns (results, wall seconds from the common start to the last finisher)."""
    import threading
    n = len(workers)
    ready, gate, res, t_end = threading.Barrier(n + 1), threading.Barrier(n + 1), [None] * n, [0.0] * n

    def body(k):
        run = workers[k]()
        ready.wait()
        gate.wait()
        res[k] = run()
        t_end[k] = time.perf_counter()

    th = [threading.Thread(target=body, args=(k,)) for k in range(n)]
    for t in th:
        t.start()
    ready.wait()
    gate.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    return res, max(t_end) - t0


def all_cores_rate(make_worker, n_cpu, budget_s):
    """Units per second with one worker per CPU: a short calibration round sizes the timed round to about budget_s (per-thread speed
    under full load is far below the single-thread speed: the reference's 46 MB scratch struct per thread is memory-bound)."""
    res, wall = run_threads([make_worker(k, 4) for k in range(n_cpu)])
    per_rep = wall / 4
    reps = int(max(2, min(100000, budget_s / per_rep)))
    res, wall = run_threads([make_worker(k, reps) for k in range(n_cpu)])
    return sum(r[0] for r in res), wall, reps


# ------------------------------------------------------------------------------------------------
NO_HOST_LEG = False  # --no-host-leg: skip the host-buffer pipeline measurement after the timed region (profiling runs)
CE_MODE = "compact"  # --ce: form of the channel estimate between the front end and the PDSCH demodulator (chain workload)
DECODER = "ref"  # --decoder: "ref" = the reference-faithful decoder (parity mode), "bcjr" = max-log-MAP, 8 iterations


def _turbo_alg_bytes(K):
    return 3 * (K + 4) + K // 8 + 4  # int8 soft in, packed bits + status out


def _kp(K):
    return (K + 63) // 64 * 64


def turbo_own_io(K, n_cb, in_bytes, out_bytes, bcjr_iters=0):
    """Minimal bytes each turbo kernel reads + writes per step for n_cb code blocks of size K, as the tile arrays are laid out
    (DESIGN.md 3 / 6.0).  in_bytes: what the first kernel reads per block (3(K+4) int8 soft values, or E soft bits);
    out_bytes: what the last kernel writes per block."""
    k = _kp(K)
    if bcjr_iters:
        h = 2 * bcjr_iters  # half-iterations: systematic, parity, a-priori in, extrinsic out, one byte of boundary state per step
        return {"k_bcjr_prep": n_cb * (in_bytes + 4 * k), "k_bcjr_half": n_cb * h * 5 * K, "k_bcjr_final": n_cb * 2 * K,
                "k_rm_bcjr_prep": n_cb * (in_bytes + 4 * k),  # the chain: soft bits in, the four granule arrays out (k_rm_to_i8 + k_bcjr_prep in one)
                "k_crc_finish": n_cb * (K + out_bytes)}
    return {"k_turbo_prep": n_cb * (in_bytes + 6 * k),          # X0 X1 X2 I0 M1 M2
            "k_turbo_siso": n_cb * 3 * 5 * k,                   # per pass: two inputs, magnitudes, output, traceback bits out and back in
            "k_turbo_perm": n_cb * 4 * k,                       # A1 X2 in, I1 M3 out
            "k_turbo_vote": n_cb * (4 * k + out_bytes)}         # X0 A1 B1 B2 in, bits out


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
        if DECODER.startswith("bcjr"):
            return "turbo-decode Mbit/s (K=6144 code blocks, 64QAM hard +-127 soft values, max-log-MAP BCJR, 8 iterations, per SURVEY 8d W3)"
        return "turbo-decode Mbit/s (K=6144 code blocks, 64QAM hard +-127 soft values, REF decoder, per SURVEY 8d W3)"

    @property
    def dominant(self):
        return "k_bcjr_half" if DECODER.startswith("bcjr") else "k_turbo_siso"

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        self.ctx, self.m, self.np = ctx, m, np
        self.n_cb = n_units or 65536
        uniq = 64
        self.tx, soft = synth.turbo_soft_blocks(self.K, uniq, flip=0.02, seed=1234 + rank, ref_wrap=(not DECODER.startswith("bcjr")))
        self.uniq_soft = soft
        idx = (np.arange(self.n_cb) * 7 + np.arange(self.n_cb) // 64) % uniq
        self.d_in = ctx.to_device(soft[idx])
        self.d_out = ctx.alloc(self.n_cb * self.K)
        self.idx = idx

    def step(self):
        if DECODER.startswith("bcjr"):
            self.ctx.turbo_decode_dev(self.d_in, self.m.SOFT_I8, self.K, self.n_cb, self.d_out,
                                      mode=self.m.TURBO_BCJR_EARLY if DECODER == "bcjr_early" else self.m.TURBO_BCJR, n_iter=8, qpp_spec=True)
        else:
            self.ctx.turbo_decode_dev(self.d_in, self.m.SOFT_I8, self.K, self.n_cb, self.d_out)

    def units_per_step(self):
        return self.n_cb

    def accounting(self):
        ks = ["k_bcjr_prep", "k_bcjr_half", "k_bcjr_final"] if DECODER.startswith("bcjr") else ["k_turbo_prep", "k_turbo_siso", "k_turbo_perm", "k_turbo_vote"]
        return {"stages": {"turbo": (self.alg_bytes_per_unit * self.n_cb, ks)},
                "own_io": turbo_own_io(self.K, self.n_cb, 3 * (self.K + 4), self.K, 8 if DECODER.startswith("bcjr") else 0)}

    def extra(self, value):
        if not DECODER.startswith("bcjr"):
            return {}
        got = self.d_out.download(self.np.uint8, count=64 * self.K).reshape(64, self.K)
        return {"decoder": "max-log-MAP, 8 iterations, fixed point (specified by oracle/lte_oracle.c lo_turbo_decode_bcjr)",
                "sampled_blocks_equal_tx_bits": bool((got == self.tx[self.idx[:64]]).all())}

    def value_per_unit(self):
        return self.K / 1e6  # information Mbit per code block

    def config(self, world):
        return {"workload": "W3 turbo decode: K=6144 x %d code blocks per GPU, int8 soft in HBM, %s mode" % (self.n_cb, DECODER.upper()),
                "K": self.K, "blocks_per_gpu": self.n_cb,
                "decoder": "BCJR max-log-MAP x8" if DECODER.startswith("bcjr") else "REF (reference-faithful, bit-exact)",
                "sharding": "code blocks block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=12.0):
        """Time the CPU reference on a bounded sample of the same blocks (rank 0, N=1 only)."""
        np = self.np
        from oracle import pyoracle
        K, D = self.K, self.K + 4
        if DECODER.startswith("bcjr"):  # the reference has no such decoder: the CPU leg is the plain-C specification of the mode
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
        model, n_cpu = cpu_info()
        out = {"value": round(n * K / t / 1e6, 4), "unit": self.unit, "cores": 1, "kind": kind, "cpu": model,
               "sample": "%d of the benchmark's K=%d code blocks, float soft values, 1 thread, %.1f s" % (n, K, t)}
        if R is not None and n_cpu > 1:
            def worker(k, reps):
                def setup():
                    phy_k = R.ref_phy_new(4, 17, 1, 100)
                    x, o = np.ascontiguousarray(np.tile(soft_f, ((reps + 63) // 64, 1))[:reps]), np.zeros(reps * K, np.uint8)

                    def run():
                        t_ = R.ref_turbo_decode_batch(phy_k, x, 3 * D, reps, o, K)
                        R.ref_phy_free(phy_k)
                        return reps, t_
                    return run
                return setup
            tot, wall, reps = all_cores_rate(worker, n_cpu, budget_s)
            out["all_cores"] = {"value": round(tot * K / wall / 1e6, 3), "unit": self.unit, "cores": n_cpu, "kind": kind, "cpu": model,
                                "sample": "%d threads, one private LIBLTE_PHY_STRUCT each, %d code blocks per thread, wall %.1f s" % (n_cpu, reps, wall)}
        return out


class ChainWorkload:
    """BASELINE config 4 / SURVEY 8d W4 on one GPU per rank: 20 MHz, 100 RB, N_ant = 1, CFI = 2, every
    subframe fully loaded with 9 x 64QAM allocations (8 x 12 PRB TBS 3240 + 1 x 4 PRB TBS 1064, one
    code block each).  One step = FFT -> CE -> demap -> rate-unmatch -> turbo (REF) -> CRC over the
    whole batch of subframes, int8 IQ resident in HBM, decoded bits + status left in HBM."""
    name = "chain"
    unit = "subframes/s"

    @property
    def metric(self):
        return "DL subframes/sec @20MHz 100RB 64QAM, full chain FFT->CE->demap->rate-unmatch->turbo(%s)->CRC (SURVEY 8d W4)" % \
               ("max-log-MAP BCJR, early termination, <= 8 iterations" if DECODER == "bcjr_early" else "max-log-MAP BCJR x8" if DECODER.startswith("bcjr") else "REF")

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
        # MI_LTE_CE_COMPACT: the estimator hands the demodulator magnitude / phase rows at the CRS symbols instead of 14 estimate rows
        # (identical results, 172 KB less HBM traffic per subframe); --ce full runs the reference's stage boundary as it is
        self.cfg = m.DlCfg(2048, 100, 1, m.IQ_I8 | (m.CE_COMPACT if CE_MODE == "compact" else 0))
        host = self.host_setup(self.n, rank, self.cfg)
        self.uniq, self.idx = host["uniq"], host["idx"]
        iq, tx, sfs, cells, allocs = self.uniq
        idx, ul = self.idx, iq.shape[1]
        if ctx is None:  # --dry-setup: the host side of the set-up only (what an 8-rank launch does eight times over before the first kernel)
            self.host = host
            return
        # the batch in HBM: the unique subframes uploaded over and over until n distinct copies lie there (unit i = unique subframe i mod U);
        # materialising the 4.6 GB on the host first cost every rank of a multi-GPU launch half a minute and 5 GB before its first kernel
        self.d_iq = ctx.alloc(self.n * ul * 2)
        U = iq.shape[0]
        for c0 in range(0, self.n, U):
            self.d_iq.upload(iq[:min(U, self.n - c0)], c0 * ul * 2)
        self.d_start = ctx.to_device((np.arange(self.n) * ul).astype(np.uint64))
        self.d_sf = ctx.to_device(sfs[idx])
        self.d_cell = ctx.to_device(cells[idx])
        self.d_sub = ctx.alloc(self.n * ctx.subframe_floats(1) * 4)
        all_allocs = host["all_allocs"]
        self.plan = ctx.pdsch_plan(self.cfg, 2, all_allocs)
        if DECODER.startswith("bcjr"):  # the max-log-MAP decoder instead of the reference's (8 iterations; the reference transmitter's interleaver)
            self.plan.set_decoder(m.TURBO_BCJR_EARLY if DECODER == "bcjr_early" else m.TURBO_BCJR, 8, 0)
        self.d_out = ctx.alloc(self.n * 9 * self.plan.out_stride)
        self.d_status = ctx.alloc(self.n * 9 * 4)

    @staticmethod
    def host_setup(n, rank, cfg):
        """Everything of the set-up that runs on the host: the 96 unique synthetic subframes (the C transmitter), the batch of n subframes
        index of the batch (unit i = unique subframe i mod 96; the batch itself -- n x 70 240 bytes, 4.6 GB at the default size, distinct bytes in
        HBM -- is laid down by repeated uploads of the unique block, not built here) and the n x 9 allocation descriptors."""
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        import lte_testdata as td
        U = min(96, n)
        sfs = np.array([[1, 2, 3, 4, 6, 7, 8, 9][i % 8] for i in range(U)], np.uint32)  # subframes 0/5 carry sync signals
        cells = ((np.arange(U) * 37 + 11 * rank) % 504).astype(np.uint32)
        allocs = []
        for u in range(U):
            allocs += td.w4_allocs(u)
        # three thirds of the unique subframes at 30 / 27 / 25 dB: the demapper's hard 64QAM decisions carry a growing share of wrong
        # +-127 soft bits into the decoder (the kernels are branch-free, so this changes the data, not the timing)
        parts = [synth.dl_units(cfg, sfs[k::3], cells[k::3], [a for u in range(k, U, 3) for a in td.w4_allocs(u // 3)], 9,
                                snr_db=snr, max_delay=8, seed=4242 + rank + k) for k, snr in enumerate((30.0, 27.0, 25.0)) if len(sfs[k::3])]
        iq = np.zeros((U,) + parts[0][0].shape[1:], parts[0][0].dtype)
        tx = np.zeros((U,) + parts[0][1].shape[1:], parts[0][1].dtype)
        for k, (q, t) in enumerate(parts):
            iq[k::3], tx[k::3] = q, t
        idx = np.arange(n) % U
        return {"uniq": (iq, tx, sfs, cells, allocs), "idx": idx, "all_allocs": m.tile_allocs(td.w4_allocs(0), n)}

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
        # where the CRC passed the bits must be the transmitted ones (all allocations of 64 subframes spread over the batch) ...
        exact = all(st[i * 9 + a] != 0 or (bits[i * 9 + a, :(3240 if a < 8 else 1064)] == tx[self.idx[i], a, :(3240 if a < 8 else 1064)]).all()
                    for i in range(0, self.n, max(1, self.n // 64)) for a in range(9))
        # ... and verdict + bits must be the CPU restatement's on a sample that covers every SNR class (the checker, not the product)
        import ctypes as C
        import lte_testdata as td
        from oracle import pyoracle as po
        P, iq_u, sfs_u, cells_u, allocs_u = po.port(), self.uniq[0], self.uniq[2], self.uniq[3], self.uniq[4]
        same = True
        for u in range(0, len(sfs_u), max(1, len(sfs_u) // 6)):
            lc, sfr = td.oracle_frontend(P, 2048, 100, 1, iq_u[u], int(sfs_u[u]), int(cells_u[u]))
            for a in range(9):
                o, nb = np.zeros(6200, np.uint8), C.c_uint32()
                la = td.to_lo_alloc(allocs_u[u * 9 + a])
                rc = P.lo_pdsch_channel_decode(C.byref(lc), C.byref(sfr), C.byref(la), 2, int(cells_u[u]), 1, o, C.byref(nb), None, None)
                same &= int(st[u * 9 + a]) == rc and (rc != 0 or bool((bits[u * 9 + a, :nb.value] == o[:nb.value]).all()))
        # the same work for a caller that holds HOST buffers: mi_lte_dl_pipeline -- pinned int8 units in, chunks of 2048 subframes
        # on two lanes with the copies on streams of their own (H2D / kernels / D2H overlap), packed transport blocks + verdicts out.  PCIe-inclusive: reported next to the
        # device-resident rate, never as `value`
        res = {"turbo_info_mbit_per_s": round(value * self.info_bits / 1e6, 2),
               "crc_pass": "%d/%d allocations" % (ok, st.size), "sampled_blocks_equal_tx_bits": bool(exact),
               "sampled_subframes_equal_cpu_restatement": bool(same)}
        if DECODER != "ref" or NO_HOST_LEG:
            return res
        import time
        m_ = self.m
        n_h = min(self.n, 32768)
        pipe = m_.DlPipeline(self.ctx.device, self.cfg, 2, td.w4_allocs(0), 2048, 2)
        ul = self.uniq[0].shape[1]
        h_iq, h_sf, h_cell = m_.HostBuffer((n_h, ul, 2), np.int8), m_.HostBuffer((n_h,), np.uint32), m_.HostBuffer((n_h,), np.uint32)
        h_out, h_st = m_.HostBuffer((n_h * 9, pipe.out_stride), np.uint8), m_.HostBuffer((n_h * 9,), np.int32)
        U = len(self.uniq[2])
        for c0 in range(0, n_h, U):
            k = min(U, n_h - c0)
            h_iq.arr[c0:c0 + k] = self.uniq[0][:k]
        h_sf.arr[:], h_cell.arr[:] = self.uniq[2][np.arange(n_h) % U], self.uniq[3][np.arange(n_h) % U]
        for _ in range(2):  # warm-up: tables, scratch, and the copy engines' queues (the runtime makes one per engine at its first use, 8 ms each)
            pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n_h, h_out.arr, h_st.arr)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            pipe.run(h_iq.arr, h_sf.arr, h_cell.arr, n_h, h_out.arr, h_st.arr)
        dt = (time.perf_counter() - t0) / reps
        hst = pipe.device_stats()[0]
        host_ok = bool((h_st.arr == st[:n_h * 9]).all()) and all(
            (np.unpackbits(h_out.arr[i * 9 + a, :(3240 if a < 8 else 1064) // 8]) == bits[i * 9 + a, :(3240 if a < 8 else 1064)]).all()
            for i in range(0, n_h, max(1, n_h // 32)) for a in range(9))
        h2d, d2h = n_h * ul * 2, h_out.arr.nbytes + h_st.arr.nbytes
        pipe.close()
        for b in (h_iq, h_sf, h_cell, h_out, h_st):
            b.free()
        res.update({
                "from_host_buffers": {"subframes_per_s": round(n_h / dt, 1), "equal_to_device_resident_results": host_ok,
                                      "h2d_GBps": round(h2d / dt / 1e9, 1), "d2h_GBps": round(d2h / dt / 1e9, 2),
                                      "last_run_stream_seconds": {"wall": round(hst["wall_s"], 4), "h2d": round(hst["h2d_s"], 4), "kernels": round(hst["kernel_s"], 4), "d2h": round(hst["d2h_s"], 4)},
                                      "note": "mi_lte_dl_pipeline: %d subframes of int8 IQ from pinned host memory (%.1f GB) in chunks of 2048 on 2 lanes, input and result copies on streams of their own (the bare pinned copy of the same bytes runs at 57.6 GB/s), "
                                              "copies overlapped with the kernels, packed transport blocks + verdicts back (%.2f GB); PCIe-inclusive, "
                                              "not the headline value" % (n_h, h2d / 1e9, d2h / 1e9)}})
        return res

    def accounting(self):
        """SURVEY 8d per-stage bytes (charged once per step) and each kernel's own minimal I/O."""
        n = self.n
        res = 8 * 1656 + 552  # PDSCH resource elements per subframe
        bc = 8 if DECODER.startswith("bcjr") else 0
        own = {"k_dl_fft": n * (15 * 2048 * 2 + 15 * 1200 * 8),       # the 15 symbol windows in, 15 rows of symbols out
               "k_dl_ce": n * (5 * 200 * 8 + 14 * 1200 * 8),           # pilots in, 14 estimate rows out
               "k_pdsch_demod": n * res * (16 + 6)}                    # y and h per element in, six soft bits out
        if CE_MODE == "compact":  # five magnitude + five phase rows out; y per element + ten values per (slot, sub-carrier) in
            own["k_dl_ce"] = n * (5 * 200 * 8 + 10 * 1200 * 4)
            own["k_pdsch_demod"] = n * (res * (8 + 6) + 2 * 1200 * 40)
        for K, cnt, E, tbs in ((3264, 8, 9936, 3240), (1088, 1, 3312, 1064)):
            for k, v in turbo_own_io(K, n * cnt, E, tbs, bc).items():
                own[k] = own.get(k, 0) + v
        tk = ["k_cb_desc", "k_rm_bcjr_prep", "k_bcjr_half", "k_bcjr_final", "k_crc_finish"] if bc else \
             ["k_cb_desc", "k_turbo_prep", "k_turbo_siso", "k_turbo_perm", "k_turbo_vote"]
        own["k_cb_desc"] = n * 9 * 80  # per code block: its allocation's fields in, a 32-byte descriptor out
        if bc:
            own.pop("k_bcjr_prep", None)  # (the stand-alone decoder's first kernel; the chain's is k_rm_bcjr_prep)
        # front end: int8 IQ in, 14 symbol rows out, and the estimate in the form this run hands to the demodulator: 14 rows of re / im
        # (339 040 B per subframe) or, compact, five magnitude + five phase rows (252 640 B)
        fe = 70240 + 14 * 1200 * 8 + (10 * 1200 * 4 if CE_MODE == "compact" else 14 * 1200 * 8)
        return {"stages": {"frontend": (n * fe, ["k_dl_fft", "k_dl_ce"]),
                           "demod": (n * res * (16 + 6), ["k_pdsch_demod"]),
                           "turbo": (n * (8 * _turbo_alg_bytes(3264) + _turbo_alg_bytes(1088)), tk)},
                "own_io": own}

    def config(self, world):
        return {"workload": "W4 full DL chain: 20 MHz/100 RB/64QAM, 9 allocations per subframe (8x12 PRB TBS 3240 + 1x4 PRB "
                            "TBS 1064), %d subframes per GPU, int8 IQ in HBM" % self.n,
                "subframes_per_gpu": self.n, "N_ant": 1, "CFI": 2, "channel_estimate_form": CE_MODE,
                "decoder": "BCJR (max-log-MAP, 8 iterations; specified by the plain-C model, not by the reference)" if DECODER.startswith("bcjr") else "REF (reference-faithful, bit-exact)",
                "unique_subframes": len(self.uniq[2]),
                "batch_note": "the %d subframes of a step are %d unique synthetic subframes (30 / 27 / 25 dB) repeated; every kernel on the path is "
                              "branch-free, so the repetition does not shorten the timed work" % (self.n, len(self.uniq[2])),
                "sharding": "subframes block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=10.0):
        """The reference's own CPU path on a bounded sample of the benchmark's subframes: (a) one thread, as the reference ships
        (it is single-threaded), (b) one private LIBLTE_PHY_STRUCT per thread on every CPU this process may use."""
        import ctypes as C
        np = self.np
        from oracle import pyoracle as po
        import lte_testdata as td
        iq, tx, sfs, cells, allocs = self.uniq
        R, fft_note = timing_ref(po)
        model, n_cpu = cpu_info()
        if R is None:  # no compiled reference on this machine: the plain-C restatement, one thread
            P = po.port()
            t_total, done = 0.0, 0
            while t_total < budget_s and done < 4000:
                u = done % len(sfs)
                sf, cell = int(sfs[u]), int(cells[u])
                re = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[u, :, 0].astype(np.float32)]))
                im = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[u, :, 1].astype(np.float32)]))
                out, nb = np.zeros(6200, np.uint8), C.c_uint32()
                t0 = time.perf_counter()
                lc, s = po.LoCfg(), po.LoSubframe()
                P.lo_cfg_init(C.byref(lc), 2048, 100)
                P.lo_get_dl_subframe_and_ce(C.byref(lc), re, im, 0, sf, cell, 1, C.byref(s))
                for a in range(9):
                    la = td.to_lo_alloc(allocs[u * 9 + a])
                    P.lo_pdsch_channel_decode(C.byref(lc), C.byref(s), C.byref(la), 2, cell, 1, out, C.byref(nb), None, None)
                t_total += time.perf_counter() - t0
                done += 1
            return {"value": round(done / t_total, 3), "unit": self.unit, "cores": 1, "kind": "port", "cpu": model,
                    "sample": "%d of the benchmark's subframes through oracle/lte_oracle.c, 1 thread, %.1f s" % (done, t_total)}

        def unit_inputs(u):
            sf, cell = int(sfs[u]), int(cells[u])
            re = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[u, :, 0].astype(np.float32)]))
            im = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), iq[u, :, 1].astype(np.float32)]))
            la = (po.LoAlloc * 9)(*[td.to_lo_alloc(allocs[u * 9 + a]) for a in range(9)])
            return sf, cell, re, im, la

        def worker(u, reps):
            def setup():
                sf, cell, re, im, la = unit_inputs(u % len(sfs))
                phy, sfp = R.ref_phy_new(4, 0, 1, 100), R.ref_subframe_new()

                def run():
                    ok = C.c_uint32()
                    t = R.ref_time_dl_chain(phy, re, im, sf, cell, sfp, la, 9, 2, reps, C.byref(ok))
                    R.ref_subframe_free(sfp)
                    R.ref_phy_free(phy)
                    return reps, t, ok.value
                return run
            return setup

        _, t3, _ = worker(0, 3)()()
        reps1 = int(max(8, min(4000, budget_s / (t3 / 3))))
        n1, t1, ok1 = worker(1, reps1)()()
        out = {"value": round(n1 / t1, 3), "unit": self.unit, "cores": 1, "kind": "reference", "cpu": model,
               "sample": "%d repetitions of one of the benchmark's subframes (liblte_phy_get_dl_subframe_and_ce + 9 x liblte_phy_pdsch_channel_decode, "
                         "%d/%d CRC pass), 1 thread, %.1f s; %s"
                         % (n1, ok1, 9 * n1, t1, fft_note)}
        if n_cpu > 1:
            tot, wall, reps = all_cores_rate(worker, n_cpu, budget_s)
            out["all_cores"] = {"value": round(tot / wall, 2), "unit": self.unit, "cores": n_cpu, "kind": "reference", "cpu": model,
                                "sample": "%d threads (every CPU this process may run on), one private LIBLTE_PHY_STRUCT each, %d repetitions per "
                                          "thread of a benchmark subframe, wall %.1f s" % (n_cpu, reps, wall)}
        return out


class ChainMixedWorkload(ChainWorkload):
    """The W4 chain on MIXED traffic: 65 536 fully loaded 20 MHz subframes whose allocation lists differ from subframe to subframe --
    every subframe its own control-region size (1-3 symbols) and 4-25 allocations of 1-25 PRB each, QPSK / 16QAM / 64QAM, redundancy
    versions 0-3, transport block sizes from 36.213 table 7.1.7.2.1-1 (F = 0, one code block: the reference's envelope) at code rates
    around 1/3 with some repetition and some puncturing: ~150 distinct code-block sizes in one batch.  Same kernels, same device-resident
    accounting as W4; the allocation list lives in a DYNAMIC plan (mi_lte_pdsch_plan_create_dynamic / _assign), as a caller that decodes
    DCIs per subframe has it (LTE_fdd_enb_phy.cc:557-770, LTE_fdd_dl_fs_samp_buf.cc:445-515)."""
    name = "chain_mixed"
    dominant = "k_turbo_siso"
    N_UNIQUE = 192

    @property
    def metric(self):
        return "DL subframes/sec @20MHz 100RB, MIXED traffic (QPSK/16QAM/64QAM, 1-25 PRB allocations, per-subframe lists), full chain FFT->CE->demap->rate-unmatch->turbo(REF)->CRC"

    @staticmethod
    def draw_lists(n_unique, rank, seed=20260601):
        """Per unique subframe: (subframe number, cell, CFI, [allocation tuples (mod, tbs, first PRB, N_prb, rnti, rv)])."""
        import numpy as np
        import openlte_amd as m
        import lte_testdata as td
        L = m.load_library()
        sizes = set(td.ALL_K)
        rng = np.random.default_rng(seed + rank)
        out = []
        for u in range(n_unique):
            sf = [1, 2, 3, 4, 6, 7, 8, 9][u % 8]  # subframes 0 / 5 carry the synchronisation signals (W4's choice too)
            cell, cfi = int((u * 37 + 11 * rank) % 504), int(rng.integers(1, 4))
            first, lst = 0, []
            while first < 100:
                n_prb = min(int(rng.integers(1, 26)), 100 - first)
                mod = int(rng.integers(1, 4))
                per_prb = ((14 - cfi) * 12 - 6) * (2, 4, 6)[mod - 1]    # single port: two CRS elements per PRB in symbols 4, 7, 11
                n_prb = min(n_prb, 10000 // per_prb)                    # the unmodified reference holds 10 000 soft bits per allocation (liblte_phy.h:355-363): 64QAM 11-13 PRB, 16QAM 16-19
                e = n_prb * per_prb
                cand = sorted({int(L.mi_lte_tbs(i, n_prb)) for i in range(27)})
                cand = [t for t in cand if t + 24 in sizes]             # one code block, no filler bits
                fit = [t for t in cand if 3 * (t + 28) <= e]
                r = rng.random()
                if not fit:
                    tbs = cand[0]
                elif r < 0.75:    # the reference's decoder is built for rate <= 1/3: the largest sizes that fit
                    tbs = fit[max(0, len(fit) - 1 - int(rng.integers(0, 3)))]
                elif r < 0.90:    # repetition (soft combining over several laps of the circular buffer)
                    tbs = fit[int(rng.integers(0, len(fit)))]
                else:             # punctured: the next sizes up
                    above = [t for t in cand if t > fit[-1]][:3]
                    tbs = above[int(rng.integers(0, len(above)))] if above else fit[-1]
                lst.append((mod, tbs, first, n_prb, 0x100 + len(lst), int(rng.choice([0, 0, 0, 0, 1, 2, 3]))))
                first += n_prb
            out.append((sf, cell, cfi, lst))
        return out

    @staticmethod
    def host_setup(n, rank, cfg):
        import ctypes as C
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        U = min(ChainMixedWorkload.N_UNIQUE, n)
        lists = ChainMixedWorkload.draw_lists(U, rank)
        sfs = np.array([l[0] for l in lists], np.uint32)
        cells = np.array([l[1] for l in lists], np.uint32)
        max_tbs = max(t[1] for l in lists for t in l[3])
        iq, tx, allocs, first = None, [], [], [0]
        for u, (sf, cell, cfi, lst) in enumerate(lists):
            al = [m.make_alloc(u, mod, tbs, list(range(p0, p0 + n_prb)), rnti, rv, 1, None, cfi) for (mod, tbs, p0, n_prb, rnti, rv) in lst]
            q, t = synth.dl_units(cfg, [sf], [cell], al, len(al), n_pdcch_symbs=cfi, snr_db=(30.0, 27.0, 25.0)[u % 3], max_delay=8, seed=777 + 13 * rank + u)
            if iq is None:
                iq = np.zeros((U,) + q.shape[1:], q.dtype)
            iq[u] = q[0]
            tx += [np.pad(t[0, a], (0, max_tbs - t.shape[2])) for a in range(len(al))]
            allocs += al
            first.append(len(allocs))
        first = np.array(first, np.int64)
        # the batch's list: unit i carries unique subframe (i mod U)'s allocations, `unit` = i -- built on the struct bytes by numpy
        sz, off = C.sizeof(m.PdschAlloc), m.PdschAlloc.unit.offset
        base = np.frombuffer((m.PdschAlloc * len(allocs))(*allocs), np.uint8).reshape(len(allocs), sz)
        reps, rem = divmod(n, U)
        full = np.concatenate([np.tile(base, (reps, 1)), base[:first[rem]]]) if rem else np.tile(base, (reps, 1))
        uidx = np.repeat(np.arange(U), np.diff(first))
        unit = np.concatenate([uidx + r * U for r in range(reps)] + ([uidx[:first[rem]] + reps * U] if rem else [])).astype(np.uint32)
        full = np.ascontiguousarray(full)
        full[:, off:off + 4] = unit.view(np.uint8).reshape(-1, 4)
        return {"uniq": (iq, np.array(tx, np.uint8), sfs, cells, allocs), "idx": np.arange(n) % U, "lists": lists, "first": first,
                "all_allocs": (m.PdschAlloc * len(full)).from_buffer(full), "_keep": full}

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        self.ctx, self.m, self.np = ctx, m, np
        self.n = n_units or 65536
        self.cfg = m.DlCfg(2048, 100, 1, m.IQ_I8 | (m.CE_COMPACT if CE_MODE == "compact" else 0))
        host = self.host_setup(self.n, rank, self.cfg)
        self.host, self.uniq, self.idx, self.lists, self.first = host, host["uniq"], host["idx"], host["lists"], host["first"]
        if ctx is None:
            return
        iq, ul, U = self.uniq[0], self.uniq[0].shape[1], self.uniq[0].shape[0]
        self.d_iq = ctx.alloc(self.n * ul * 2)
        for c0 in range(0, self.n, U):
            self.d_iq.upload(iq[:min(U, self.n - c0)], c0 * ul * 2)
        self.d_start = ctx.to_device((np.arange(self.n) * ul).astype(np.uint64))
        self.d_sf = ctx.to_device(self.uniq[2][self.idx])
        self.d_cell = ctx.to_device(self.uniq[3][self.idx])
        self.d_sub = ctx.alloc(self.n * ctx.subframe_floats(1) * 4)
        arr = host["all_allocs"]
        self.n_alloc = len(arr)
        # every allocation of a unique subframe with its E (soft bits), K and information bits, for the accounting
        rows = []
        for (sf, cell, cfi, lst) in self.lists:
            rows += [(n_prb * ((14 - cfi) * 12 - 6), (2, 4, 6)[mod - 1], tbs + 24, tbs) for (mod, tbs, p0, n_prb, rnti, rv) in lst]
        self.rows = np.array(rows, np.int64)
        reps, rem = divmod(self.n, U)
        self.mult = np.full(len(rows), reps, np.int64)
        self.mult[:self.first[rem]] += 1  # how often each unique allocation occurs in the batch
        # the plan reserves an allocation's soft bits by (14 - CFI) x 12 x N_prb x Q_m, CRS positions included, in 64-byte lines
        cap = np.array([(14 - cfi) * 12 * n_prb * (2, 4, 6)[mod - 1] for (sf, cell, cfi, lst) in self.lists for (mod, tbs, p0, n_prb, rnti, rv) in lst], np.int64)
        soft_bytes = int((((cap + 63) // 64 * 64) * self.mult).sum())
        self.plan = ctx.pdsch_plan_dynamic(self.cfg, self.n_alloc, soft_bytes)
        self.assign = lambda: ctx._check(ctx.L.mi_lte_pdsch_plan_assign(ctx.h, self.plan.h, 2, C_cast_void(arr), self.n_alloc))
        self.assign()
        self.plan.out_stride = ctx.L.mi_lte_pdsch_plan_out_stride(self.plan.h)
        self.d_out = ctx.alloc(self.n_alloc * self.plan.out_stride)
        self.d_status = ctx.alloc(self.n_alloc * 4)
        self.info_bits_per_step = int((self.rows[:, 3] * self.mult).sum())
        self.info_bits = self.info_bits_per_step / self.n  # per subframe (mean)

    def extra(self, value):
        import ctypes as C
        np = self.np
        st = self.d_status.download(np.int32)
        U, first = len(self.lists), self.first
        n_u = int(first[-1])
        tx = self.uniq[1]
        # where the CRC passed the bits must be the transmitted ones: every allocation of the batch's first pass over the unique subframes ...
        bits = self.d_out.download(np.uint8, count=n_u * self.plan.out_stride).reshape(n_u, self.plan.out_stride)
        tbs = self.rows[:, 3]
        exact = all(st[a] != 0 or bool((bits[a, :tbs[a]] == tx[a, :tbs[a]]).all()) for a in range(n_u))
        # ... the verdicts repeat with the subframes (unit i = unique subframe i mod U) ...
        reps = self.n // U
        periodic = bool((st[:reps * n_u].reshape(reps, n_u) == st[:n_u]).all())
        # ... and verdict + bits must be the compiled reference's on a sample of the unique subframes (the checker, not the product)
        from oracle import pyoracle as po
        import lte_testdata as td
        R = po.ref()
        same, n_cmp, kind, differ = True, 0, "compiled reference (oracle/_ref)", []
        if R is None:
            kind = "plain-C restatement (oracle/lte_oracle.c)"
        P = po.port()
        for u in range(0, U, max(1, U // 8)):
            sf, cell, cfi, lst = self.lists[u]
            q = self.uniq[0][u]
            if R is not None:
                i_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), q[:, 0].astype(np.float32)]))
                q_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), q[:, 1].astype(np.float32)]))
                phy, rx = R.ref_phy_new(4, cell, 1, 100), R.ref_subframe_new()
                R.ref_get_dl_subframe_and_ce(phy, i_f, q_f, 0, sf, cell, 1, rx)
            else:
                lc, sfr = td.oracle_frontend(P, 2048, 100, 1, q, sf, cell)
            for k, (mod, t, p0, n_prb, rnti, rv) in enumerate(lst):
                a = int(first[u]) + k
                o, nb = np.zeros(6200, np.uint8), C.c_uint32()
                la = po.make_alloc(mod, t, list(range(p0, p0 + n_prb)), rnti, rv, 1)
                if R is not None:
                    rc = R.ref_pdsch_channel_decode(phy, rx, C.byref(la), cfi, cell, 1, o, C.byref(nb))
                else:
                    rc = P.lo_pdsch_channel_decode(C.byref(lc), C.byref(sfr), C.byref(la), cfi, cell, 1, o, C.byref(nb), None, None)
                eq = int(st[a]) == rc and (rc != 0 or bool((bits[a, :nb.value] == o[:nb.value]).all()))
                if not eq:
                    differ.append({"unique_subframe": u, "allocation": k, "mod": mod, "tbs": t, "N_prb": n_prb, "rv": rv, "cfi": cfi, "library": int(st[a]), "checker": int(rc)})
                same &= eq
                n_cmp += 1
            if R is not None:
                R.ref_subframe_free(rx)
                R.ref_phy_free(phy)
        ok_bits = int((self.rows[:, 3] * self.mult * (st[:n_u] == 0)).sum()) if periodic else None
        ks = np.unique(self.rows[:, 2])
        res = {"turbo_info_mbit_per_s": round(value * self.info_bits / 1e6, 2),
               "turbo_info_mbit_per_s_crc_passed": round(value / self.n * ok_bits / 1e6, 2) if ok_bits is not None else None,
               "info_bits_per_subframe_mean": round(self.info_bits, 1),
               "allocations_per_step": self.n_alloc, "distinct_code_block_sizes": int(len(ks)),
               "crc_pass": "%d/%d allocations" % (int((st == 0).sum()), st.size), "verdicts_repeat_with_the_unique_subframes": periodic,
               "sampled_blocks_equal_tx_bits": bool(exact),
               "sampled_allocations_equal_checker": {"equal": bool(same), "allocations": n_cmp, "checker": kind, "differing": differ[:8]}}
        # the same step with the list handed over anew every step (a caller whose grants change per batch): mi_lte_pdsch_plan_assign + run
        ctx = self.ctx
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            self.assign()
            self.step()
        ctx.sync()
        res["ms_per_step_with_plan_assign"] = round((time.perf_counter() - t0) / 3 * 1e3, 3)
        return res

    def accounting(self):
        n, np = self.n, self.np
        re_, qm, K, tbs = (self.rows[:, k] for k in range(4))
        mult = self.mult
        own = {"k_dl_fft": n * (15 * 2048 * 2 + 15 * 1200 * 8), "k_dl_ce": n * (5 * 200 * 8 + 14 * 1200 * 8),
               "k_pdsch_demod": int((re_ * (16 + qm) * mult).sum())}
        if CE_MODE == "compact":
            own["k_dl_ce"] = n * (5 * 200 * 8 + 10 * 1200 * 4)
            own["k_pdsch_demod"] = int((re_ * (8 + qm) * mult).sum()) + n * 2 * 1200 * 40
        for k_ in np.unique(K):
            sel = K == k_
            for kn, v in turbo_own_io(int(k_), 1, 0, 0).items():
                own[kn] = own.get(kn, 0) + v * int(mult[sel].sum())
            own["k_turbo_prep"] += int((re_[sel] * qm[sel] * mult[sel]).sum())
            own["k_turbo_vote"] += int((tbs[sel] * mult[sel]).sum())
        own["k_cb_desc"] = self.n_alloc * 80
        fe = 70240 + 14 * 1200 * 8 + (10 * 1200 * 4 if CE_MODE == "compact" else 14 * 1200 * 8)
        tk = ["k_cb_desc", "k_turbo_prep", "k_turbo_siso", "k_turbo_perm", "k_turbo_vote", "k_turbo_tail"]
        return {"stages": {"frontend": (n * fe, ["k_dl_fft", "k_dl_ce"]),
                           "demod": (int((re_ * (16 + qm) * mult).sum()), ["k_pdsch_demod"]),
                           "turbo": (int(((3 * (K + 4) + K // 8 + 4) * mult).sum()), tk)},
                "own_io": own}

    @property
    def alg_bytes_per_unit(self):  # fused accounting: int8 IQ in + packed information bits out
        return 70240 + self.info_bits / 8

    def config(self, world):
        ks = self.np.unique(self.rows[:, 2])
        return {"workload": "W4-mixed: full DL chain, 20 MHz/100 RB, every subframe fully loaded with its own list of 1-25 PRB allocations "
                            "(QPSK/16QAM/64QAM, at most 10 000 soft bits each = the reference's PDSCH scratch, rv 0-3, CFI 1-3, TBS from 36.213 table 7.1.7.2.1-1 with F = 0), %d subframes per GPU, int8 IQ in HBM" % self.n,
                "subframes_per_gpu": self.n, "N_ant": 1, "channel_estimate_form": CE_MODE, "decoder": "REF (reference-faithful, bit-exact)",
                "unique_subframes": len(self.lists), "allocations_per_subframe_mean": round(self.n_alloc / self.n, 2),
                "distinct_code_block_sizes": int(len(ks)), "K_min": int(ks.min()), "K_max": int(ks.max()),
                "plan": "dynamic (mi_lte_pdsch_plan_create_dynamic + _assign), list resident on the device during the timed steps",
                "sharding": "subframes block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=10.0):
        """The compiled reference, one thread, on the unique subframes of this workload in turn (front end + every allocation)."""
        import ctypes as C
        np = self.np
        from oracle import pyoracle as po
        R, fft_note = timing_ref(po)
        if R is None:
            return None
        model, n_cpu = cpu_info()
        t_total, done, ok, n_al = 0.0, 0, 0, 0
        out, nb = np.zeros(6200, np.uint8), C.c_uint32()
        while t_total < budget_s and done < 2000:
            u = done % len(self.lists)
            sf, cell, cfi, lst = self.lists[u]
            q = self.uniq[0][u]
            i_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), q[:, 0].astype(np.float32)]))
            q_f = np.ascontiguousarray(np.concatenate([np.zeros(sf * 30720, np.float32), q[:, 1].astype(np.float32)]))
            las = [po.make_alloc(mod, t, list(range(p0, p0 + n_prb)), rnti, rv, 1) for (mod, t, p0, n_prb, rnti, rv) in lst]
            phy, rx = R.ref_phy_new(4, cell, 1, 100), R.ref_subframe_new()
            t0 = time.perf_counter()
            R.ref_get_dl_subframe_and_ce(phy, i_f, q_f, 0, sf, cell, 1, rx)
            for la in las:
                ok += R.ref_pdsch_channel_decode(phy, rx, C.byref(la), cfi, cell, 1, out, C.byref(nb)) == 0
            t_total += time.perf_counter() - t0
            R.ref_subframe_free(rx)
            R.ref_phy_free(phy)
            done += 1
            n_al += len(las)
        return {"value": round(done / t_total, 3), "unit": self.unit, "cores": 1, "kind": "reference", "cpu": model,
                "sample": "%d of this workload's unique subframes (liblte_phy_get_dl_subframe_and_ce + every allocation's liblte_phy_pdsch_channel_decode, "
                          "%d/%d CRC pass), 1 thread, %.1f s; %s" % (done, ok, n_al, t_total, fft_note)}


def C_cast_void(arr):
    import ctypes as C
    return C.cast(arr, C.c_void_p)


class FrontendWorkload:
    """BASELINE config 2 / SURVEY 8d W2: 20 MHz OFDM demod + CRS channel estimate, 10k subframe units."""
    name = "frontend"
    unit = "subframes/s"
    dtype = "i8 IQ in, f32 FFT/CE"
    N_ANT = 1
    dominant = "k_dl_ce"

    @property
    def metric(self):
        return "DL subframes/sec @20MHz 100RB, OFDM demod + CRS channel estimate only, %d antenna port(s) (SURVEY 8d W2)" % self.N_ANT

    @property
    def alg_bytes_per_unit(self):  # int8 IQ in, 14 symbol rows + 14 estimate rows per port out
        return 70240 + 14 * 1200 * 8 * (1 + self.N_ANT)

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        import lte_testdata as td
        self.ctx, self.np = ctx, np
        self.n = n_units or 10000
        self.cfg = m.DlCfg(2048, 100, self.N_ANT, 0)
        if self.N_ANT == 1:
            U = 64
            sfs = (np.arange(U) % 10).astype(np.uint32)
            cells = ((np.arange(U) * 37 + rank) % 504).astype(np.uint32)
            allocs = []
            for u in range(U):
                allocs += td.w4_allocs(u)
            iq, _ = synth.dl_units(m.DlCfg(2048, 100, 1, 0), sfs, cells, allocs, 9, snr_db=30.0, max_delay=8, seed=77 + rank)
        else:
            # a REAL two-port cell: four subframe units from the reference's own transmitter (CRS on both ports, a transmit-diversity
            # allocation, each antenna through its own gain), generated once by tools/gen_golden.py and committed -- nothing of oracle/ or
            # /root/reference is touched here
            z = np.load(os.path.join(ROOT, "tests", "golden", "dl_two_port_units.npz"))
            iq, sfs, cells = z["iq"], z["sfs"], z["cells"]
            U = len(sfs)
        self.uniq = (iq, sfs, cells)
        idx = np.arange(self.n) % U
        ul = iq.shape[1]
        self.d_iq = ctx.to_device(iq[idx].reshape(-1, 2))
        self.d_start = ctx.to_device((np.arange(self.n) * ul).astype(np.uint64))
        self.d_sf, self.d_cell = ctx.to_device(sfs[idx]), ctx.to_device(cells[idx])
        self.d_sub = ctx.alloc(self.n * ctx.subframe_floats(self.N_ANT) * 4)

    def step(self):
        self.ctx.dl_frontend_dev(self.cfg, self.d_iq, None, self.d_start, self.d_sf, self.d_cell, self.n, self.d_sub)

    def units_per_step(self):
        return self.n

    def value_per_unit(self):
        return 1.0

    def extra(self, value):
        """After the clock: a sample of the step's device subframes against the CPU restatement of liblte_phy_get_dl_subframe_and_ce
        (the checker, not the product) -- symbol rows within 1e-5, every port's estimate rows within 1e-4, relative L2."""
        np = self.np
        import lte_testdata as td
        from oracle import pyoracle as po
        P = po.port()
        iq, sfs, cells = self.uniq
        nf, p, worst = self.ctx.subframe_floats(self.N_ANT), self.N_ANT, [0.0, 0.0]
        rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))
        picks = sorted({int(x) for x in np.linspace(0, self.n - 1, 6)})
        for i in picks:
            got = self.d_sub.download(np.float32, count=nf, offset=i * nf * 4).reshape(2 + 2 * p, 16, 1200)
            u = i % len(sfs)
            _, s = td.oracle_frontend(P, 2048, 100, p, iq[u], int(sfs[u]), int(cells[u]))
            worst[0] = max(worst[0], rel(got[0, :14], s.arr("rx_symb_re")[:14]), rel(got[1, :14], s.arr("rx_symb_im")[:14]))
            for q in range(p):
                worst[1] = max(worst[1], rel(got[2 + q, :14], s.arr("rx_ce_re")[q, :14]), rel(got[2 + p + q, :14], s.arr("rx_ce_im")[q, :14]))
        return {"sampled_subframes_vs_cpu_restatement": {"units": len(picks), "worst_rel_l2_symbols": float("%.3g" % worst[0]), "tolerance_symbols": 1e-5,
                                                         "worst_rel_l2_estimates": float("%.3g" % worst[1]), "tolerance_estimates": 1e-4,
                                                         "within_tolerance": bool(worst[0] < 1e-5 and worst[1] < 1e-4)}}

    def accounting(self):
        n, p = self.n, self.N_ANT
        return {"stages": {"frontend": (n * self.alg_bytes_per_unit, ["k_dl_fft", "k_dl_ce"])},
                "own_io": {"k_dl_fft": n * (15 * 2048 * 2 + 15 * 1200 * 8), "k_dl_ce": n * p * (5 * 200 * 8 + 14 * 1200 * 8)}}

    def config(self, world):
        return {"workload": "W2 front end: 20 MHz/100 RB, %d subframe units per GPU, %d antenna port(s), int8 IQ in HBM" % (self.n, self.N_ANT),
                "subframes_per_gpu": self.n, "N_ant": self.N_ANT, "unique_subframes": len(self.uniq[1]),
                "capture": "own transmitter (openlte_amd/synth), single-port cell" if self.N_ANT == 1 else
                           "two-port cell from the reference's transmitter (tests/golden/dl_two_port_units.npz, tools/gen_golden.py)",
                "sharding": "subframes over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=10.0):
        np = self.np
        from oracle import pyoracle as po
        R, fft_note = timing_ref(po)
        iq, sfs, cells = self.uniq
        if R is None:
            return None
        phy, sfp = R.ref_phy_new(4, 0, 1, 100), R.ref_subframe_new()
        re = np.ascontiguousarray(iq[0, :, 0].astype(np.float32))
        im = np.ascontiguousarray(iq[0, :, 1].astype(np.float32))
        t = R.ref_time_get_dl_subframe_and_ce(phy, re, im, 0, 0, int(cells[0]), self.N_ANT, sfp, 50)
        reps = int(max(100, budget_s / (t / 50)))
        t = R.ref_time_get_dl_subframe_and_ce(phy, re, im, 0, 0, int(cells[0]), self.N_ANT, sfp, reps)
        return {"value": round(reps / t, 2), "unit": self.unit, "cores": 1, "kind": "reference",
                "sample": "%d calls of liblte_phy_get_dl_subframe_and_ce (N_ant = %d), 1 thread, %.1f s; %s" % (reps, self.N_ANT, t, fft_note)}


class Frontend2Workload(FrontendWorkload):
    """W2 with two antenna ports: the estimator runs once per port (liblte_phy.cc:5959-6194 loops over p)."""
    name = "frontend2"
    N_ANT = 2


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
        self.prach_last, self.prach_in_flight = None, False

    def step(self):
        """One batch of subframes and its PRACH occasions.  The random-access verdicts come back through mi_lte_prach_detect_launch / _fetch:
        this step's are fetched (a wait for THAT launch, then a few microseconds of host arithmetic) after the next step's subframes have been
        handed to the device, as a receiver that keeps its stream busy would; finish() fetches the last ones inside the timed region."""
        self.ctx.ul_frontend_dev(self.cfg, self.d_iq, None, self.d_start, self.n, self.d_sub)
        self.plan.run_dev(self.d_sub, self.d_out, self.d_status)
        self.finish()
        self.pplan.launch_dev(self.d_prach, None, self.d_pstart, self.n_occ)
        self.prach_in_flight = True

    def finish(self):
        if self.prach_in_flight:
            self.prach_last = self.pplan.fetch()
            self.prach_in_flight = False

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

    def accounting(self):
        n, M, K = self.n, 12 * self.N_PRB, self.TBS + 24
        E = 12 * M * 2  # QPSK soft bits per allocation
        own = {"k_prach_fft": self.n_occ * (24576 * 2 + 24576 * 8), "k_prach_bins": self.n_occ * (24576 * 8 + 839 * 8),
               "k_prach_corr": self.n_occ * (839 * 8 + 64 * 12),
               "k_ul_fft": n * (14 * 2048 * 2 + 14 * 1200 * 8), "k_pusch_demod": n * self.N_UE * (14 * M * 8 + E)}
        own.update(turbo_own_io(K, n * self.N_UE, E, self.TBS))
        own["k_cb_desc"] = n * self.N_UE * 80
        return {"stages": {"frontend": (n * (61440 + 14 * 1200 * 8), ["k_ul_fft"]),
                           "demod": (n * self.N_UE * (14 * M * 8 + E), ["k_pusch_demod"]),
                           "turbo": (n * self.N_UE * _turbo_alg_bytes(K), ["k_cb_desc", "k_turbo_prep", "k_turbo_siso", "k_turbo_perm", "k_turbo_vote"]),
                           "prach": (self.n_occ * (24576 * 2 + 64 * 12), ["k_prach_fft", "k_prach_bins", "k_prach_corr"])},
                "own_io": own}

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
        R, fft_note = timing_ref(po)
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
                          "%.1f s; %s" % (reps, t, fft_note)}


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

    def accounting(self):
        b = self.n * self.alg_bytes_per_unit
        return {"stages": {"control": (b, ["k_pdcch_decode"])}, "own_io": {"k_pdcch_decode": b}}

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

    def accounting(self):
        n = self.n
        own = {"k_cp_corr": n * (161 * 15360 * 2 + 160 * 15360 * 8), "k_cp_accum": n * (160 * 15360 * 8 + 15360 * 4), "k_cp_gather": n * 160 * 5 * 16,
               "k_sync_fft": n * 165 * (2048 * 2 + 1200 * 8), "k_seq_corr": n * (164 * 1200 * 8 + 1200 * 8)}
        return {"stages": {"sync": (n * self.alg_bytes_per_unit, list(own))}, "own_io": own}

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
        n_units = n_units or {"chain": 65536, "chain_mixed": 65536, "frontend": 10000, "frontend2": 10000, "turbo": 65536, "uplink": 16384, "control": 8192, "sync": 32}[cls.name]
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
        for p in self.parts:  # (results a workload fetches behind its last step: part of the step, inside the timed region)
            if hasattr(p, "finish"):
                p.finish()
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

    def accounting(self):
        accs = [p.accounting() for p in self.parts]
        st = {k: (sum(a["stages"][k][0] for a in accs), v[1]) for k, v in accs[0]["stages"].items()}
        own = {k: sum(a["own_io"].get(k, 0) for a in accs) for k in accs[0]["own_io"]}
        return {"stages": st, "own_io": own}

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


WORKLOADS = {"turbo": TurboWorkload, "frontend": FrontendWorkload, "frontend2": Frontend2Workload, "chain": ChainWorkload, "chain-mixed": ChainMixedWorkload, "uplink": UplinkWorkload, "control": ControlWorkload, "sync": SyncWorkload}


def pick_workload(name):
    if name != "auto":
        return WORKLOADS[name]
    return WORKLOADS["chain"]


def turbo_leg(ctx, decoder, n_cb, steps):
    """The second half of BASELINE.json's metric inside the default line: W3 (K = 6144 code blocks, int8 soft values resident in HBM)
    decoded `steps` times, timed with HIP events on the launch stream.  Returns information Mbit/s."""
    import numpy as np
    import openlte_amd as m
    from openlte_amd import synth
    K = 6144
    tx, soft = synth.turbo_soft_blocks(K, 64, flip=0.02, seed=4321, ref_wrap=(decoder not in ("bcjr", "bcjr_early")))
    idx = (np.arange(n_cb) * 7 + np.arange(n_cb) // 64) % 64
    d_in, d_out = ctx.to_device(soft[idx]), ctx.alloc(n_cb * K)

    def step():
        if decoder == "bcjr":
            ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, n_cb, d_out, mode=m.TURBO_BCJR, n_iter=8, qpp_spec=True)
        elif decoder == "bcjr_early":
            ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, n_cb, d_out, mode=m.TURBO_BCJR_EARLY, n_iter=8, qpp_spec=True)
        else:
            ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, n_cb, d_out)
    step()
    ctx.sync()
    ctx.timer_start()
    for _ in range(steps):
        step()
    ms = ctx.timer_stop()
    got = d_out.download(np.uint8, count=64 * K).reshape(64, K)
    out = {"mbit_per_s": round(n_cb * K * steps / (ms * 1e-3) / 1e6, 1), "ms_per_decode": round(ms / steps, 3), "code_blocks": n_cb, "K": K, "steps": steps}
    if decoder == "bcjr_early":
        it = ctx.turbo_early_exit_iterations()
        out["sampled_blocks_equal_tx_bits"] = bool((got == tx[idx[:64]]).all())
        out["iterations_per_tile_pair"] = {str(k): int((it == k).sum()) for k in sorted(set(int(x) for x in it))}
        out["note"] = ("hard-decision-aided early termination per tile pair (128 code blocks), at most 8 iterations: a throughput mode of its own, "
                       "NOT the 8-iteration figure; the soft values here are hard +-127 with 2 % flips, which two iterations settle")
    if decoder == "bcjr":
        out["sampled_blocks_equal_tx_bits"] = bool((got == tx[idx[:64]]).all())
        # the same decoder for a per-call caller's handful of blocks: MI_LTE_TURBO_BCJR_BLOCK, one code block per wavefront, one launch
        lat = {}
        for nb in (1, 9):
            for name, mode in (("batch_kernels", m.TURBO_BCJR), ("one_block_per_wavefront", m.TURBO_BCJR_BLOCK)):
                ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, nb, d_out, mode=mode, n_iter=8, qpp_spec=True)
                ctx.sync()
                ctx.timer_start()
                for _ in range(5):
                    ctx.turbo_decode_dev(d_in, m.SOFT_I8, K, nb, d_out, mode=mode, n_iter=8, qpp_spec=True)
                lat["%s_%d_block%s_ms" % (name, nb, "" if nb == 1 else "s")] = round(ctx.timer_stop() / 5, 3)
        out["latency_8_iterations"] = lat
    d_in.free()
    d_out.free()
    if decoder not in ("bcjr", "bcjr_early"):  # K = 6144 is one of the sizes whose interleaver the reference computes with uint32 overflow: it never decodes to the transmitted
        # bits there; the check is bit-equality with the CPU restatement of the reference's decoder
        from oracle import pyoracle
        P, want = pyoracle.port(), np.zeros(K, np.uint8)
        ok = True
        for b in range(4):
            P.lo_turbo_decode_ref(np.ascontiguousarray(soft[idx[b]], dtype=np.float32), K, want)
            ok &= bool((got[b] == want).all())
        out["sampled_blocks_equal_cpu_restatement"] = ok
    return out


_COPY_RATE = {}


def traffic_of(tab, label):
    """PMC bytes per launch of the kernel behind a launch label.  rocprof sees the kernel's own name: the 2048-point transform of all
    three front ends (labels k_dl_fft / k_ul_fft / k_sync_fft) is k_dl_fft2k since round 4, k_dl_fft in older tables."""
    if label in ("k_dl_fft", "k_ul_fft", "k_sync_fft"):
        return tab.get("k_dl_fft2k", tab.get("k_dl_fft"))
    return tab.get(label, tab.get(label + "_multi"))  # (k_cb_desc: the merged decode's kernel is k_cb_desc_multi)


KERNEL_FILE = (("k_turbo", "turbo.hip"), ("k_cb_desc", "turbo.hip"), ("k_rm_", "turbo.hip"), ("k_crc_finish", "turbo.hip"), ("k_rate_unmatch", "turbo.hip"),
               ("k_bcjr", "bcjr.hip"), ("k_pdsch", "chain.hip"), ("k_dl_", "frontend.hip"), ("k_ul_fft", "frontend.hip"), ("k_sync_fft", "frontend.hip"),
               ("k_pusch", "uplink.hip"), ("k_pucch", "uplink.hip"), ("k_prach", "prach.hip"), ("k_pdcch", "pdcch.hip"), ("k_pbch", "pdcch.hip"),
               ("k_cp_corr", "sync.hip"), ("k_seq_corr", "sync.hip"), ("k_freq_shift", "sync.hip"), ("k_pss", "sync.hip"), ("k_sss", "sync.hip"))


def build_id():
    import openlte_amd as m
    return m.load_library().mi_lte_build_id().decode()


def table_is_current(kernel, table_id):
    """Was the committed profiler table (profiles/pmc_traffic_*.json, sq_counters_*.json: "build_id") measured on THIS build of the kernel's
    source file?  The id is "file.hip:<sha1 of the file><sha1 of the shared headers>;" per file (mi_lte_build_id); a table without an id, or a kernel
    whose file has changed since, is stale: its number is not reported."""
    if not table_id:
        return False
    f = next((fn for pre, fn in KERNEL_FILE if kernel.startswith(pre)), None)
    then = dict(x.split(":") for x in table_id.split(";") if x)
    now = dict(x.split(":") for x in build_id().split(";") if x)
    return f is not None and f in then and then[f] == now.get(f)


def load_table(name):
    """(table dict or None, its build id)"""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", name)))
        return tj, tj.get("build_id")
    except Exception:
        return None, None


# cycles one SIMD spends per wave64 vector instruction in a stream that mixes the opcode classes -- what every kernel on these paths is
# (profiles/r04_ubench_issue_mix.txt: alternating fast and slow opcodes run both at 4.1-4.5; profiles/r06_isa_classes.txt: 1-7 % of the
# kernels' vector instructions sit in runs of fast opcodes long enough to issue at their own 2.2-2.5)
VALU_CYCLES_PER_INST = 4.3
N_SIMD = 1024


def valu_of(wl_name, ms_per_kernel, step_ms, units_per_step):
    """The `roofline.valu` object: the vector ALUs' ISSUE time per step from the committed SQ counters of this workload (profiles/
    sq_counters_<workload>.json: wave64 vector instructions issued per step and kernel, SQ_INSTS_VALU) at the measured issue cost and
    the clock the same counters give (busy cycles over this run's kernel times).  None when the table is missing or stale for a timed kernel."""
    tab, tid = load_table("sq_counters_%s.json" % wl_name)
    if not tab:
        return None
    per = tab["per_step"]
    timed = [k for k in ms_per_kernel if ms_per_kernel[k] > 0]
    def prof_name(k):  # rocprof sees the kernel's own name: the 2048-point transform behind three launch labels, the merged decode's descriptor kernel
        for cand in (("k_dl_fft2k",) if k in ("k_dl_fft", "k_ul_fft", "k_sync_fft") else ()) + (k, k + "_multi"):
            if cand in per:
                return cand
        return k
    names = {k: prof_name(k) for k in timed}
    if any(names[k] not in per or not table_is_current(k, tid) for k in timed):
        return {"note": "profiles/sq_counters_%s.json was measured on another build of a timed kernel (or lacks one): no VALU figure" % wl_name}
    scale = units_per_step / tab["units_per_step"] if tab.get("units_per_step") else 1.0
    insts = sum(per[names[k]].get("SQ_INSTS_VALU", 0) for k in timed) * scale
    busy = sum(per[names[k]].get("SQ_BUSY_CYCLES", 0) for k in timed) * scale / 32.0  # (one count per shader engine)
    kern_ms = sum(ms_per_kernel[k] for k in timed)
    clock_ghz = busy / (kern_ms * 1e-3) / 1e9 if busy else 2.3
    issue_ms = insts * VALU_CYCLES_PER_INST / (N_SIMD * clock_ghz * 1e9) * 1e3
    return {"wave_instructions_per_step": int(insts), "issue_cycles_per_instruction": VALU_CYCLES_PER_INST, "simds": N_SIMD, "clock_GHz": round(clock_ghz, 3),
            "valu_issue_ms_per_step": round(issue_ms, 3), "valu_issue_frac": round(issue_ms / step_ms, 4),
            "per_kernel_issue_ms": {k: round(per[names[k]].get("SQ_INSTS_VALU", 0) * scale * VALU_CYCLES_PER_INST / (N_SIMD * clock_ghz * 1e9) * 1e3, 3) for k in timed},
            "source": "profiles/sq_counters_%s.json (rocprofv3 --pmc SQ_INSTS_VALU, SQ_BUSY_CYCLES; build %s) x %.1f cycles per wave64 instruction "
                      "(profiles/r04_ubench_issue_rate_run2_wall.txt, r04_ubench_issue_mix.txt, r06_isa_classes.txt)" % (wl_name, "current", VALU_CYCLES_PER_INST),
            "reading": "share of the step during which every one of the 1024 SIMDs would be issuing vector instructions if the instruction stream were spread evenly: "
                       "the bound this path runs against (no dense contraction, nothing for the matrix cores); the HBM fractions beside it say how little of the "
                       "memory system the same step needs"}


def measured_copy_rate(ctx):
    """GB/s of the library's copy kernel on ctx's device (1 GiB, 10 launches; once per process and device)."""
    key = id(ctx)
    if key not in _COPY_RATE:
        try:
            _COPY_RATE[key] = round(ctx.device_copy_rate(1 << 30, 10), 1)
            _COPY_RATE["shapes"] = ctx.device_copy_rates()
        except Exception:
            _COPY_RATE[key] = None
    return _COPY_RATE[key]


def roofline_of(wl, prof, steps):
    """The `roofline` object of a run: the kernel with the largest share of the timed region, its STAGE's algorithmic bytes charged once
    per step and spread over its launches, over its average launch time (HIP events).  Also returns {kernel: ms per step}."""
    acc = wl.accounting()
    stage_of = {k: st for st, (_, ks) in acc["stages"].items() for k in ks}
    dom = max(prof, key=lambda k: prof[k][1])
    n_launch, tot_ms = prof[dom]
    lps = max(1, n_launch // steps)
    st_bytes = acc["stages"][stage_of[dom]][0] if dom in stage_of else wl.alg_bytes_per_unit * wl.units_per_step()
    avg_ms = tot_ms / n_launch
    achieved = st_bytes / lps / (avg_ms * 1e-3) / 1e9
    traffic = None
    tj, tid = load_table("pmc_traffic_%s.json" % wl.name)
    if tj and table_is_current(dom, tid):
        traffic = traffic_of(tj["bytes_per_launch"], dom)
    st_ms = sum(prof[k][1] for k in acc["stages"][stage_of[dom]][1] if k in prof) / steps if dom in stage_of else tot_ms / steps
    copy = measured_copy_rate(wl.ctx) if hasattr(wl, "ctx") else None
    alone = None
    if wl.name.startswith("frontend"):
        # W2 IS its stage: both kernels run once per step and the stage's bytes are the sum of what each must move, so the honest price
        # is the stage's bytes over the stage's time -- charging them to the dominant kernel's time alone credits it with the other
        # kernel's output (round-3 review, item 6).  The dominant kernel's own figure stays beside it.
        alone = {"kernel": dom, "stage_bytes_over_this_kernels_time_GBps": round(achieved, 2), "avg_launch_ms": round(avg_ms, 4)}
        achieved = st_bytes / (st_ms * 1e-3) / 1e9
    step_ms = sum(ms for (nl, ms) in prof.values()) / steps
    valu = valu_of(wl.name, {k: ms / steps for k, (nl, ms) in prof.items()}, step_ms, wl.units_per_step())
    whole_frac = wl.alg_bytes_per_unit * wl.units_per_step() / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {"bound": "valu" if valu and valu.get("valu_issue_frac", 0) > max(whole_frac, achieved / HBM_PEAK_GBS) else "hbm", "valu": valu,
            "kernel": dom if alone is None else "+".join(k for k in acc["stages"][stage_of[dom]][1] if k in prof),
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "dominant_kernel_alone": alone,
            # SURVEY 8d's second denominator: what a 16-bytes-per-lane copy kernel reaches on THIS device, measured in this run
            "measured_copy_GBps": copy, "frac_of_measured_copy": round(achieved / copy, 5) if copy else None,
            "traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": lps, "algorithmic_bytes_per_launch": st_bytes / lps,
            "stage": stage_of.get(dom, "whole path"),
            # the same bytes over ALL of the stage's kernels (the stricter reading when the dominant kernel is only part of its stage)
            "stage_ms_per_step": round(st_ms, 4), "stage_frac": round(st_bytes / (st_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}, \
           {k: round(ms / steps, 4) for k, (nl, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])}


def config_leg(cls, ctx, steps, cpu_budget_s):
    """Another BASELINE config inside the default line (after the timed region, like the turbo legs): the workload at its own default
    size on the same context, `steps` steps between two synchronisations with the per-launch HIP events on, its own `roofline` and
    `cpu_baseline` (bounded)."""
    wl = cls(ctx, 0, 0)
    finish = getattr(wl, "finish", lambda: None)  # (results a workload fetches behind its last step: inside the timed interval)
    wl.step()
    finish()
    ctx.sync()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    finish()
    ctx.sync()
    dt = time.perf_counter() - t0
    prof = ctx.profile_report()
    ctx.profile(False)
    roof, kern = roofline_of(wl, prof, steps)
    out = {"metric": wl.metric, "value": round(wl.units_per_step() * wl.value_per_unit() * steps / dt, 1), "unit": wl.unit, "steps": steps,
           "ms_per_step": round(dt / steps * 1e3, 4), "config": wl.config(1), "dtype": wl.dtype, "roofline": roof, "kernels_ms_per_step": kern}
    if hasattr(wl, "extra"):
        out.update(wl.extra(out["value"]))
    if cpu_budget_s > 0:
        cb = wl.cpu_baseline(cpu_budget_s)
        if cb:
            out["cpu_baseline"] = cb
    for name, v in list(vars(wl).items()):  # release the leg's device buffers before the next one
        if hasattr(v, "free") and name.startswith("d_"):
            v.free()
    for name in ("plan", "pplan"):
        if hasattr(wl, name):
            getattr(wl, name).close()
    return out


def strong_run(args):
    """--strong: ONE contiguous host-resident capture through the product's multi-device entry point (mi_lte_dl_pipeline_create_multi +
    mi_lte_dl_pipeline_run_capture): one process, one host thread per GPU, chunks block-cyclic over the devices, the capture split on
    subframe boundaries with the look-ahead halo, per-subframe allocation lists, no collective.  Total work is fixed as N grows
    (`scaling: strong`); PCIe-inclusive by construction (the capture starts in pinned host memory), so this is NOT the headline metric."""
    import numpy as np
    import openlte_amd as m
    from openlte_amd import synth
    import lte_testdata as td
    n_dev = m.load_library().mi_lte_device_count()
    devices = [d % max(1, n_dev) for d in range(args.gpus)] if args.oversubscribe else list(range(args.gpus))
    if max(devices) >= n_dev:
        raise SystemExit("bench.py --strong: --gpus %d but only %d HIP device(s) are visible (--oversubscribe maps them modulo, for testing)" % (args.gpus, n_dev))
    cfg = m.DlCfg(2048, 100, 1, m.IQ_I8 | m.CE_COMPACT)
    n, U, cell = args.units or 32768, 40, 77
    sfs = [(3 + u) % 10 for u in range(U)]
    per_u = [9 if sfs[u] not in (0, 5) else 3 for u in range(U)]  # (subframes 0 / 5: the first PRBs only, clear of the sync signals)
    # the transmitter takes a fixed number of allocations per unit: synthesise the two kinds of subframes separately, static channel, and lay them end to end
    full = [u for u in range(U) if per_u[u] == 9]
    part = [u for u in range(U) if per_u[u] == 3]
    iq_u = np.zeros((U, synth.unit_len(2048), 2), np.int8)
    if full:
        q, _ = synth.dl_units(cfg, [sfs[u] for u in full], [cell] * len(full), [a for k, u in enumerate(full) for a in td.w4_allocs(k)], 9, snr_db=30.0, max_delay=-1, gain=(1.0, 1.0), seed=11)
        iq_u[full] = q
    if part:
        q, _ = synth.dl_units(cfg, [sfs[u] for u in part], [cell] * len(part), [a for k, u in enumerate(part) for a in td.w4_allocs(k)[:3]], 3, snr_db=30.0, max_delay=-1, gain=(1.0, 1.0), seed=12)
        iq_u[part] = q
    n = n // U * U
    multi = len(set(devices)) > 1
    h_cap = m.HostBuffer((n * 30720 + 4400, 2), np.int8, device=-1 if multi else devices[0])  # several devices: pages interleaved over the memory nodes
    for u in range(U):
        h_cap.arr[u * 30720:(u + 1) * 30720] = iq_u[u, :30720]
    for r in range(1, n // U):
        h_cap.arr[r * U * 30720:(r + 1) * U * 30720] = h_cap.arr[:U * 30720]
    h_cap.arr[n * 30720:] = iq_u[0, :4400]  # the next subframe's first symbols (subframe sfs[0] again)
    first, allocs = [0], []
    for i in range(n):
        u = i % U
        for a in (td.w4_allocs(i) if per_u[u] == 9 else td.w4_allocs(i)[:3]):
            allocs.append(a)
        first.append(len(allocs))
    arr = (m.PdschAlloc * len(allocs))(*allocs)
    first = np.array(first, np.uint32)
    pipe = m.DlPipeline(devices, cfg, 2, None, args.chunk, n_lanes=args.lanes, max_alloc_per_unit=9, max_soft_bytes_per_unit=8 * 9984 + 3328)
    h_out = m.HostBuffer((len(allocs), pipe.out_stride), np.uint8, device=-1 if multi else devices[0])
    h_st = m.HostBuffer((len(allocs),), np.int32, device=-1 if multi else devices[0])
    for _ in range(max(1, args.warmup)):
        pipe.run_capture(h_cap.arr, 0, n, sfs[0], cell, arr, first, 2, h_out.arr, h_st.arr)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pipe.run_capture(h_cap.arr, 0, n, sfs[0], cell, arr, first, 2, h_out.arr, h_st.arr)
    dt = time.perf_counter() - t0
    ok = int((h_st.arr == 0).sum())
    # what every device did in the LAST run, so that a scaling run explains its own bottleneck: its share, its link, its kernels, its host thread
    per_dev = []
    for d in pipe.device_stats():
        per_dev.append({"device": d["device"], "chunks": d["chunks"], "subframes": d["units"],
                        "h2d_GBps_while_copying": round(d["h2d_bytes"] / d["h2d_s"] / 1e9, 2) if d["h2d_s"] > 0 else None,
                        "h2d_GBps_over_the_run": round(d["h2d_bytes"] / d["wall_s"] / 1e9, 2) if d["wall_s"] > 0 else None,
                        "stream_seconds": {"h2d": round(d["h2d_s"], 4), "kernels": round(d["kernel_s"], 4), "d2h": round(d["d2h_s"], 4)},
                        "host_thread_wall_s": round(d["wall_s"], 4), "host_thread_numa_node": d["numa_node"], "host_thread_cpus": d["n_cpus"],
                        "device_numa_node": m.load_library().mi_lte_device_numa_node(d["device"])})
    print(json.dumps({"metric": "DL subframes/sec @20MHz 100RB 64QAM, full chain from ONE host-resident capture (PCIe-inclusive), strong scaling over the product's "
                                "multi-device pipeline", "value": round(n * args.steps / dt, 1), "unit": "subframes/s", "n_gpus": len(devices), "steps": args.steps,
                      "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                      "dtype": ChainWorkload.dtype, "data": "synthetic",
                      "config": {"workload": "one contiguous int8 capture of %d subframes (%.2f GB, pinned host memory), 9 allocations per subframe (3 in subframes 0 / 5), "
                                             "mi_lte_dl_pipeline_run_capture: chunks of %d subframes block-cyclic over %d device(s), %d lanes each, look-ahead halo per chunk"
                                             % (n, h_cap.arr.nbytes / 1e9, args.chunk, len(devices), args.lanes), "devices": devices},
                      "crc_pass": "%d/%d allocations" % (ok, len(allocs)), "h2d_GBps": round(h_cap.arr.nbytes * args.steps / dt / 1e9, 2),
                      "per_device_last_run": per_dev,
                      "host_memory": {"capture_numa": {-2: "interleaved over all nodes", -1: "runtime's default placement"}.get(h_cap.node, "node %d" % h_cap.node),
                                      "results_numa": {-2: "interleaved over all nodes", -1: "runtime's default placement"}.get(h_out.node, "node %d" % h_out.node)},
                      "note": "PCIe-inclusive and host-driven: not comparable with the device-resident headline value"}))
    pipe.close()
    for b in (h_cap, h_out, h_st):
        b.free()


def selftest(args, rank, world, barrier, max_reduce):
    """--workload selftest: the launch / sharding / timing control plane with no GPU work (what tests/test_dist_cpu.py runs through
    `--gpus 2` on CPU).  A step is a host-side walk over this rank's shard of a 1000-unit batch."""
    from openlte_amd.sharding import shard_units, shard_counts
    mine = list(shard_units(1000, rank, world))
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        acc = sum(mine)
    barrier()
    elapsed = max_reduce(time.perf_counter() - t0 + 1e-6)
    import torch.distributed as dist
    ranks = [None] * world
    if world > 1:
        dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", rank)), len(mine), mine[:3]))
    else:
        ranks = [(0, 0, len(mine), mine[:3])]
    if rank == 0:
        print(json.dumps({"metric": "selftest (control plane only, no GPU work)", "value": round(1000 * args.steps / elapsed, 1), "unit": "units/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none", "data": "synthetic",
                          "config": {"workload": "selftest"}, "ranks": ranks, "shard_counts": shard_counts(1000, world), "checksum": acc}))
    barrier()


def dry_setup(args, rank, world, barrier, max_reduce):
    """--workload chain-setup: the HOST side of the chain workload's set-up on every rank, no GPU (needs none): what `--gpus 8` does eight
    times over on one node before its first kernel -- the C transmitter's 96 unique subframes and the allocation descriptors (the 4.6 GB batch is laid down
    in HBM by repeated uploads of the unique block; until round 5 it was built on the host first: 29 s and 5 GB per rank).  Prints seconds and resident memory per rank, so that the 8-GPU launch is
    known to fit and how long its silence lasts."""
    import resource
    import openlte_amd as m
    barrier()
    t0 = time.perf_counter()
    n = args.units or 65536
    host = ChainWorkload.host_setup(n, rank, m.DlCfg(2048, 100, 1, m.IQ_I8 | m.CE_COMPACT))
    dt = time.perf_counter() - t0
    rss = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6  # GB (ru_maxrss is in KB)
    worst, peak = max_reduce(dt), max_reduce(rss)
    barrier()
    if rank == 0:
        print(json.dumps({"metric": "chain workload, host side of the set-up only (no GPU work)", "value": round(worst, 2), "unit": "s", "n_gpus": world,
                          "steps": 0, "warmup": 0, "ms_per_step": None, "higher_is_better": False, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "synthetic", "config": {"workload": "chain-setup", "subframes_per_rank": n},
                          "slowest_rank_s": round(worst, 2), "peak_resident_GB_per_rank": round(peak, 2), "ranks": world,
                          "unique_block_bytes_per_rank": int(host["uniq"][0].nbytes), "batch_bytes_in_hbm_per_rank": int(n * host["uniq"][0][0].nbytes),
                          "descriptor_bytes_per_rank": n * 9 * 260}))
    barrier()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--units", type=int, default=0, help="units (subframes / code blocks) per GPU per step")
    ap.add_argument("--streams", type=int, default=1, help="independent shards (contexts/streams) per GPU")
    ap.add_argument("--decoder", default="ref", choices=["ref", "bcjr", "bcjr_early"], help="turbo workload only: decoder mode")
    ap.add_argument("--ce", default="compact", choices=["compact", "full"], help="chain workload: channel-estimate form handed to the demodulator")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="time the steps without the per-launch HIP events (no roofline in the line): measures what the events cost")
    ap.add_argument("--no-host-leg", action="store_true", help="chain workload: skip the host-buffer pipeline measurement after the timed region")
    ap.add_argument("--no-turbo-leg", action="store_true", help="chain workload: skip the short W3 turbo-decode legs after the timed region")
    ap.add_argument("--strong", action="store_true", help="one host-resident capture through the multi-device pipeline (one process, a host thread per GPU); strong scaling")
    ap.add_argument("--oversubscribe", action="store_true", help="--strong: map --gpus N onto the visible devices modulo (testing the multi-device path on one GPU)")
    ap.add_argument("--chunk", type=int, default=2048, help="--strong: subframes per chunk")
    ap.add_argument("--lanes", type=int, default=2, help="--strong: lanes per device (the copies have streams of their own: two lanes double-buffer the kernels; more streams than hardware queues serialise)")
    args = ap.parse_args()

    global DECODER, CE_MODE, NO_HOST_LEG
    DECODER, CE_MODE, NO_HOST_LEG = args.decoder, args.ce, args.no_host_leg
    if args.strong:
        return strong_run(args)
    maybe_relaunch(args.gpus, sys.argv[1:])
    rank, world, barrier, max_reduce = dist_setup(args.gpus)
    if world > 1:
        NO_HOST_LEG = True  # rank 0's host-pipeline leg would keep the other ranks at the closing barrier: the N-rank wall time is the timed region's
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if args.workload == "selftest":
        return selftest(args, rank, world, barrier, max_reduce)
    if args.workload == "chain-setup":
        return dry_setup(args, rank, world, barrier, max_reduce)
    import openlte_amd as m
    n_dev = m.load_library().mi_lte_device_count()
    if local_rank >= n_dev:  # one GPU per rank, no sharing: n_gpus in the line means N contexts on N distinct devices
        raise SystemExit("bench.py: rank %d needs GPU %d but only %d HIP device(s) are visible; --gpus %d cannot be honoured here"
                         % (rank, local_rank, n_dev, args.gpus))
    ctxs = [m.Context(local_rank) for _ in range(max(1, args.streams))]
    ctx = ctxs[0]
    wl = MultiStream(pick_workload(args.workload), ctxs, args.units, rank)

    for _ in range(args.warmup):
        wl.step()
    wl.sync()
    wl.profile(not args.no_kernel_events)  # HIP events around every kernel launch, on the launch stream (two event records per launch)
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
    devices = [local_rank]
    if world > 1:
        import torch.distributed as dist
        got = [None] * world
        dist.all_gather_object(got, (int(os.environ.get("GROUP_RANK", "0")), local_rank))
        if len(set(got)) != world:
            raise SystemExit("bench.py: ranks share a device: %s" % (got,))
        devices = [g[1] for g in got]

    if rank == 0 and args.no_kernel_events:
        units = wl.units_per_step() * world * args.steps
        print(json.dumps({"metric": wl.metric, "value": round(units * wl.value_per_unit() / elapsed, 3), "unit": wl.unit, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic", "config": wl.config(world), "kernel_events": False, "build_id": build_id()}))
    elif rank == 0:
        steps = args.steps
        units = wl.units_per_step() * world * steps
        value = units * wl.value_per_unit() / elapsed
        acc = wl.accounting()
        ms_step = {k: ms / steps for k, (nl, ms) in prof.items()}
        stage_of = {k: st for st, (_, ks) in acc["stages"].items() for k in ks}
        # HBM bytes per launch from the committed rocprofv3 --pmc passes of this command (tools/profile_bench.sh) -- only for kernels whose
        # source file is the one the table was measured on (mi_lte_build_id): an edited kernel without a re-profile reports no traffic
        tj, tid = load_table("pmc_traffic_%s%s.json" % (wl.name, "_bcjr" if DECODER.startswith("bcjr") else ""))
        traffic_tab = {k: v for k, v in (tj["bytes_per_launch"] if tj else {}).items() if table_is_current(k, tid)}
        stale_tables = sorted(k for k in (tj["bytes_per_launch"] if tj else {}) if any(k.startswith(pre) for pre, _ in KERNEL_FILE) and not table_is_current(k, tid))
        copy_rate = measured_copy_rate(ctx)  # (the timed region is over)
        per_kernel = {}
        for k, (nl, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1]):
            own = acc["own_io"].get(k)
            ent = {"ms_per_step": round(ms / steps, 4), "launches_per_step": nl // steps, "stage": stage_of.get(k)}
            if own:
                ent["own_io_bytes_per_step"] = own
                ent["own_io_GBps"] = round(own / (ms / steps * 1e-3) / 1e9, 1)
                ent["own_io_frac_of_peak"] = round(own / (ms / steps * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            tr = traffic_of(traffic_tab, k)
            if tr:
                ent["hbm_traffic_bytes_per_step"] = int(tr * (nl // steps))
                ent["hbm_traffic_GBps"] = round(tr * (nl // steps) / (ms / steps * 1e-3) / 1e9, 1)
                if copy_rate:  # how close the kernel's real traffic runs to what a copy kernel streams on this device
                    ent["hbm_traffic_frac_of_measured_copy"] = round(ent["hbm_traffic_GBps"] / copy_rate, 3)
                if own:
                    ent["traffic_over_own_io"] = round(tr * (nl // steps) / own, 2)
            per_kernel[k] = ent
        stages = {}
        for st, (b, ks) in acc["stages"].items():
            ms = sum(ms_step.get(k, 0.0) for k in ks)
            if ms > 0:
                tr = sum(per_kernel[k].get("hbm_traffic_bytes_per_step", 0) for k in ks if k in per_kernel)
                stages[st] = {"algorithmic_bytes_per_step": b, "ms_per_step": round(ms, 4), "alg_GBps": round(b / (ms * 1e-3) / 1e9, 1),
                              "frac_of_peak": round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "kernels": [k for k in ks if k in ms_step]}
                if tr:
                    stages[st]["hbm_traffic_over_algorithmic"] = round(tr / b, 2)
        dom = max(prof, key=lambda k: prof[k][1])  # the kernel with the largest share of the timed region
        n_launch, tot_ms = prof[dom]
        lps = max(1, n_launch // steps)
        st_bytes = acc["stages"][stage_of[dom]][0] if dom in stage_of else wl.alg_bytes_per_unit * wl.units_per_step()
        alg_per_launch = st_bytes / lps                       # the stage's bytes, charged once per step, spread over this kernel's launches
        avg_ms = tot_ms / n_launch
        achieved = alg_per_launch / (avg_ms * 1e-3) / 1e9
        dom_alone = None
        if wl.name.startswith("frontend") and dom in stage_of:  # W2 is its stage: the stage's bytes over the stage's time (see roofline_of)
            st_ms = sum(prof[k][1] for k in acc["stages"][stage_of[dom]][1] if k in prof) / steps
            dom_alone = {"kernel": dom, "stage_bytes_over_this_kernels_time_GBps": round(achieved, 2), "avg_launch_ms": round(avg_ms, 4)}
            achieved = st_bytes / (st_ms * 1e-3) / 1e9
        tr = traffic_of(traffic_tab, dom)
        whole = wl.alg_bytes_per_unit * units / elapsed / 1e9
        valu = valu_of(wl.name + ("_bcjr" if DECODER.startswith("bcjr") else ""), ms_step, elapsed / steps * 1e3, wl.units_per_step())
        out = {
            "metric": wl.metric, "value": round(value, 3), "unit": wl.unit, "n_gpus": world, "steps": steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
            "config": wl.config(world),
            "roofline": {"bound": "valu" if valu and valu.get("valu_issue_frac", 0) > max(whole / HBM_PEAK_GBS, achieved / HBM_PEAK_GBS) else "hbm",
                         "bound_note": "the larger of the two fractions: `frac` / `stage_frac` / `chain_frac` price SURVEY 8d's algorithmic bytes against the 8 TB/s of HBM3E, "
                                       "`valu.valu_issue_frac` prices the vector instructions the step issues against the SIMDs' measured issue rate",
                         "valu": valu, "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": tr, "traffic_tables_stale_for": stale_tables or None, "dominant_kernel_alone": dom_alone,
                         # SURVEY 8d's second denominator: a 16-bytes-per-lane copy kernel on THIS device, measured after the timed region
                         "measured_copy_GBps": copy_rate, "frac_of_measured_copy": round(achieved / copy_rate, 5) if copy_rate else None,
                         "measured_copy_shapes_GBps": dict(zip(("one_access_per_thread", "one_access_per_thread_non_temporal", "grid_stride_loop"), _COPY_RATE.get("shapes", ()))),
                         "traffic_source": ("profiles/pmc_traffic_%s.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)" % wl.name) if tr else None,
                         "avg_launch_ms": round(avg_ms, 4), "launches": n_launch, "launches_per_step": lps,
                         "algorithmic_bytes_per_launch": alg_per_launch,
                         # the stricter readings, inside this object so that they travel with it: the same stage bytes over ALL of the stage's
                         # kernels, and the whole path's algorithmic bytes over the whole step
                         "stage": stage_of.get(dom, "whole path"),
                         "stage_ms_per_step": stages.get(stage_of.get(dom), {}).get("ms_per_step"),
                         "stage_frac": stages.get(stage_of.get(dom), {}).get("frac_of_peak"),
                         "chain_ms_per_step": round(elapsed / steps * 1e3, 4), "chain_frac": round(whole / HBM_PEAK_GBS, 5),
                         "accounting": "stage '%s' algorithmic bytes (SURVEY 8d, %d B per step) charged ONCE per step and spread over this kernel's %d "
                                       "launches, divided by its measured average launch time (HIP events on the launch stream)"
                                       % (stage_of.get(dom, "whole path"), st_bytes, lps)},
            "stages": stages,
            "kernels": per_kernel,
            "whole_chain_alg_GBps": round(whole, 2), "whole_chain_frac_of_peak": round(whole / HBM_PEAK_GBS, 5),
            "device": ctx.device_name, "devices": devices, "build_id": build_id(),
        }
        if hasattr(wl, "extra"):
            out.update(wl.extra(value))
        if wl.name == "chain" and DECODER == "ref" and not args.no_turbo_leg and world == 1:
            out["turbo_decode"] = {"note": "BASELINE.json's second metric, measured in this run after the timed region: W3, K = 6144 x 65536 code blocks, "
                                           "int8 soft values resident in HBM, information Mbit/s",
                                   "bcjr_max_log_map_8_iterations": turbo_leg(ctx, "bcjr", 65536, 3),
                                   "bcjr_early_termination_up_to_8_iterations": turbo_leg(ctx, "bcjr_early", 65536, 3),
                                   "ref_decoder": turbo_leg(ctx, "ref", 65536, 3)}
        if wl.name == "chain" and DECODER == "ref" and not args.no_turbo_leg and world == 1:
            # BASELINE.json's other single-GPU configurations, measured in this run after the timed region (each with its own roofline and
            # a bounded CPU baseline): config 4 again on MIXED traffic (per-subframe allocation lists, ~100 code-block sizes: what the callers
            # of the reference see; the headline's W4 is its best-case shape), config 2 (W2 front end, one and two antenna ports), config 5 (W5 uplink)
            cpu_s = 0 if args.no_cpu_baseline else 4.0
            out["other_configs"] = {"W4_mixed_traffic": config_leg(ChainMixedWorkload, ctx, 5, cpu_s),
                                    "W2_frontend_1_port": config_leg(FrontendWorkload, ctx, 10, cpu_s),
                                    "W2_frontend_2_ports": config_leg(Frontend2Workload, ctx, 10, cpu_s),
                                    "W5_uplink": config_leg(UplinkWorkload, ctx, 5, cpu_s)}
        if world == 1 and not args.no_cpu_baseline:
            cb = wl.cpu_baseline()
            if cb:
                allc = cb.pop("all_cores", None)
                out["cpu_baseline"] = cb
                if allc:
                    out["cpu_baseline_all_cores"] = allc
        print(json.dumps(out))
    barrier()
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
