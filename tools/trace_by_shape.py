#!/usr/bin/env python3
"""tools/trace_by_shape.py <rocprofv3 output dir>: average duration per (kernel, workgroup size, grid) from a --kernel-trace csv --
the per-width launches of a merged decode carry one launch label (k_turbo_prep ...) in bench.py's own events."""
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"])
    if m:
        acc[(m.group(1), int(r["Workgroup_Size_X"]), int(r["Grid_Size_X"]))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sorted(acc.items()):
    if not k[0].startswith("k_copy"):
        print("%-22s wg %4d grid %11d launches %4d avg ms %.4f" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))
