#!/bin/bash
# tools/ab/siso_launch_times.sh <workload> <pad1> <pad23> [order]: median duration of the trellis kernel's two launches (rocprofv3 kernel trace, 8 steps)
cd "$(dirname "$0")/../.."
W=$1; export MI_LTE_SISO1_LDS=$2 MI_LTE_SISO23_LDS=$3; [ -n "$4" ] && export MI_LTE_SISO_ORDER=$4
D=$(mktemp -d)
( cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --output-format csv -d $D -o k -- python $OLDPWD/bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
python - "$D" "$W $2 $3 $4" <<'PY'
import csv, glob, sys, statistics, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "k_turbo_siso" in r["Kernel_Name"] and "small" not in r["Kernel_Name"]:
        acc[int(r["Grid_Size_X"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
print(sys.argv[2], " ".join("grid %d: median %.3f min %.3f (n %d)" % (g, statistics.median(v), min(v), len(v)) for g, v in sorted(acc.items())))
PY
rm -rf $D
