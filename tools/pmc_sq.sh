#!/bin/bash
# SQ counters for one bench workload (own run: PMC + kernel-trace only)
TAG=${1:-sq}; shift || true
OUT=gpurun_out/pmc_${TAG}; mkdir -p $OUT; export TMPDIR=/tmp
CTRS=${SQ_COUNTERS:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY}
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o k -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-turbo-leg --no-host-leg "$@" > $OUT/bench.json 2> $OUT/log.txt
python - <<PY
import csv, collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
import re
for r in csv.DictReader(open("$OUT/k_counter_collection.csv")):
    m=re.search(r"(k_[a-z0-9_]+)", r["Kernel_Name"]); k=(m.group(1) if m else r["Kernel_Name"][:20], r["Grid_Size"])
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
    if r["Counter_Name"]=="$(echo $CTRS | cut -d" " -f1)": cnt[k]+=1
for k,v in acc.items():
    n=cnt[k] or 1
    print(k, "launches",n, {c: round(x/n) for c,x in v.items()})
PY
