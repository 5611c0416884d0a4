#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory into profiles/<tag>_*.{csv,md,json}.

FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3).  Per MI355X_MICROARCH.md (HBM section) this
rocprofv3 reports exactly half the bytes of a wide coalesced streaming read on gfx950, so the
read side is doubled ("fetch x2"); WRITE_SIZE is uncalibrated and reported as is.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys


def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def main(src, tag, workload="chain"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dst = os.path.join(root, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "k_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    stats = collections.OrderedDict()
    with open(os.path.join(src, "trace", "k_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            s = stats.setdefault(short(r["Name"]), {"calls": 0, "total_ms": 0.0, "pct": 0.0})
            s["calls"] += int(r["Calls"])
            s["total_ms"] += float(r["TotalDurationNs"]) / 1e6
            s["pct"] += float(r["Percentage"])
    for s in stats.values():
        s["avg_us"] = s["total_ms"] * 1e3 / s["calls"]
    # per-dispatch grouping by (kernel, grid) so that differently sized launches are not averaged together
    disp = collections.defaultdict(lambda: {"n": 0, "ns": 0.0})
    with open(os.path.join(src, "trace", "k_kernel_trace.csv")) as f:
        for r in csv.DictReader(f):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
            disp[key]["n"] += 1
            disp[key]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    pmc = collections.defaultdict(dict)
    for cname, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        path = os.path.join(src, sub, "k_counter_collection.csv")
        if not os.path.exists(path):
            continue
        acc = collections.defaultdict(lambda: [0, 0.0])
        with open(path) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != cname:
                    continue
                key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
                acc[key][0] += 1
                acc[key][1] += float(r["Counter_Value"])
        for key, (n, v) in acc.items():
            pmc[key][cname] = v / n * 1024.0  # bytes per launch
    rows = []
    for key in sorted(disp, key=lambda k: -disp[k]["ns"]):
        if not key[0].startswith("k_"):
            continue
        d = disp[key]
        fetch, write = pmc[key].get("FETCH_SIZE"), pmc[key].get("WRITE_SIZE")
        rows.append({"kernel": key[0], "grid_threads": key[1], "launches": d["n"], "avg_us": round(d["ns"] / d["n"] / 1e3, 2),
                     "fetch_bytes_raw": None if fetch is None else int(fetch),
                     "fetch_bytes_x2": None if fetch is None else int(2 * fetch),
                     "write_bytes": None if write is None else int(write)})
    # average HBM bytes per launch per kernel (what bench.py reports as roofline.traffic)
    traffic = {}
    for r in rows:
        if r["fetch_bytes_x2"] is None or r["write_bytes"] is None:
            continue
        t = traffic.setdefault(r["kernel"], [0, 0.0])
        t[0] += r["launches"]
        t[1] += r["launches"] * (r["fetch_bytes_x2"] + r["write_bytes"])
    with open(os.path.join(dst, "pmc_traffic_%s.json" % workload), "w") as f:
        bid = None
        try:  # the bench line of the profiled run names the build the numbers belong to (mi_lte_build_id)
            for l in open(os.path.join(src, "bench_trace.json")):
                if l.startswith("{"):
                    bid = json.loads(l).get("build_id")
        except OSError:
            pass
        json.dump({"source": tag, "workload": workload, "build_id": bid, "note": "avg (2*FETCH_SIZE + WRITE_SIZE) bytes per launch, rocprofv3 --pmc passes of bench.py",
                   "bytes_per_launch": {k: int(v[1] / v[0]) for k, v in traffic.items()}}, f, indent=1)
    with open(os.path.join(dst, tag + "_summary.json"), "w") as f:
        json.dump({"kernel_stats": stats, "per_launch": rows}, f, indent=1)
    with open(os.path.join(dst, tag + "_summary.md"), "w") as f:
        f.write("# %s -- rocprofv3 summary (MI355X)\n\n" % tag)
        f.write("Source: `tools/profile_bench.sh` (kernel-trace `--stats` pass + separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` "
                "passes of the same `bench.py` command; raw stats in `%s_kernel_stats.csv`).\n\n" % tag)
        f.write("| kernel | calls | avg us | total ms | % |\n|---|---|---|---|---|\n")
        for k, v in stats.items():
            if k.startswith("k_"):
                f.write("| %s | %d | %.1f | %.2f | %.1f |\n" % (k, v["calls"], v["avg_us"], v["total_ms"], v["pct"]))
        f.write("\n(`k_copy_words`, `k_copy16`, `k_copy16_loop`, `k_done_flag`: the workload's set-up -- the batch is laid down in HBM by repeated uploads "
                "through the 4 MiB bounce block -- and `bench.py`'s copy-rate measurement; outside the timed region.  The percentages are of the whole process.)\n")
        f.write("\nPer launch shape (grid in threads), HBM bytes per launch from the PMC passes "
                "(FETCH_SIZE x2 per the gfx950 correction in MI355X_MICROARCH.md; WRITE_SIZE uncorrected):\n\n")
        f.write("| kernel | grid | launches | avg us | fetch B (x2) | write B | (fetch+write)/time GB/s |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            tot = (r["fetch_bytes_x2"] or 0) + (r["write_bytes"] or 0)
            f.write("| %s | %d | %d | %.1f | %s | %s | %.0f |\n" % (r["kernel"], r["grid_threads"], r["launches"], r["avg_us"],
                                                              r["fetch_bytes_x2"], r["write_bytes"], tot / (r["avg_us"] * 1e-6) / 1e9))
    print(open(os.path.join(dst, tag + "_summary.md")).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], *(sys.argv[3:4]))
