#!/usr/bin/env python3
"""bench.py -- one JSON line per run (driver contract).

    python bench.py --gpus N --steps K --warmup W [--workload chain|frontend|turbo]

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
class TurboWorkload:
    """BASELINE config 3: PDSCH turbo decode, K=6144, 64QAM-like int8 soft values, 64k code blocks
    per GPU.  REF mode (bit-exact with the reference decoder)."""
    name = "turbo"
    K = 6144
    metric = "turbo-decode Mbit/s (K=6144 code blocks, 64QAM hard +-127 soft values, REF decoder, per SURVEY 8d W3)"
    unit = "Mbit/s"
    dtype = "i8 soft values, i32 path metrics"
    alg_bytes_per_unit = 19216  # 3(K+4) int8 in + K/8 packed out + 4 B status, K=6144
    dominant = "k_turbo_siso"

    def __init__(self, ctx, n_units, rank):
        import numpy as np
        import openlte_amd as m
        from openlte_amd import synth
        self.ctx, self.m, self.np = ctx, m, np
        self.n_cb = n_units or 65536
        uniq = 64
        _, soft = synth.turbo_soft_blocks(self.K, uniq, flip=0.02, seed=1234 + rank)
        self.uniq_soft = soft
        idx = (np.arange(self.n_cb) * 7 + np.arange(self.n_cb) // 64) % uniq
        self.d_in = ctx.to_device(soft[idx])
        self.d_out = ctx.alloc(self.n_cb * self.K)
        self.idx = idx

    def step(self):
        self.ctx.turbo_decode_dev(self.d_in, self.m.SOFT_I8, self.K, self.n_cb, self.d_out)

    def units_per_step(self):
        return self.n_cb

    def value_per_unit(self):
        return self.K / 1e6  # information Mbit per code block

    def config(self, world):
        return {"workload": "W3 turbo decode: K=6144 x %d code blocks per GPU, int8 soft in HBM, REF mode" % self.n_cb,
                "K": self.K, "blocks_per_gpu": self.n_cb, "decoder": "REF (reference-faithful, bit-exact)",
                "sharding": "code blocks block-cyclic over %d GPU(s), no collective" % world}

    def cpu_baseline(self, budget_s=12.0):
        """Time the CPU reference on a bounded sample of the same blocks (rank 0, N=1 only)."""
        np = self.np
        from oracle import pyoracle
        K, D = self.K, self.K + 4
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


WORKLOADS = {"turbo": TurboWorkload}


def pick_workload(name):
    if name != "auto":
        return WORKLOADS[name]
    for k in ("chain", "frontend", "turbo"):
        if k in WORKLOADS:
            return WORKLOADS[k]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="auto")
    ap.add_argument("--units", type=int, default=0, help="units (subframes / code blocks) per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank, world, barrier, max_reduce = dist_setup(args.gpus)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    import openlte_amd as m
    ctx = m.Context(local_rank)
    wl = pick_workload(args.workload)(ctx, args.units, rank)

    for _ in range(args.warmup):
        wl.step()
    ctx.sync()
    ctx.profile(True)  # HIP events around every kernel launch, on the launch stream
    barrier()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    ctx.sync()
    barrier()
    t1 = time.perf_counter()
    elapsed = max_reduce(t1 - t0)
    prof = ctx.profile_report()
    ctx.profile(False)

    if rank == 0:
        units = wl.units_per_step() * world * args.steps
        value = units * wl.value_per_unit() / elapsed
        dom = wl.dominant if wl.dominant in prof else max(prof, key=lambda k: prof[k][1])
        n_launch, tot_ms = prof[dom]
        avg_ms = tot_ms / n_launch
        alg_bytes = wl.alg_bytes_per_unit * wl.units_per_step()
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        out = {
            "metric": wl.metric, "value": round(value, 3), "unit": wl.unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": wl.dtype, "data": "synthetic",
            "config": wl.config(world),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "avg_launch_ms": round(avg_ms, 4), "launches": n_launch,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "serial-trellis kernels are ALU/latency bound; see DESIGN.md"},
            "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items())},
            "device": ctx.device_name,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = wl.cpu_baseline()
        print(json.dumps(out))
    barrier()
    ctx.close()


if __name__ == "__main__":
    main()
