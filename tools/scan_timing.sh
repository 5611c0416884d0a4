#!/bin/bash
# the three scanners' phase timings on each bandwidth, five runs each: scan_cpu (all reference), scan_gpu (per-call shim), scan_batch (batch entry points;
# MI_LTE_SCAN_TRACE=1 splits its PDSCH stage: the first batch of a process is the cold one -- table builds, first launches -- the second is warm)
cd "$(dirname "$0")/../shim/_build"
for cfg in "6 17 30 1.92" "25 301 24 7.68" "100 77 12 30.72"; do
  set -- $cfg
  ./capture_gen /tmp/cap_$1.bin $1 $2 $3 > /dev/null 2>&1
  for i in 1 2 3 4 5; do ./scan_gpu /tmp/cap_$1.bin $4 2>&1 >/dev/null | grep timing | sed "s/^/$1 RB scan_gpu  : /"; done
  for i in 1 2 3 4 5; do MI_LTE_SCAN_TRACE=1 ./scan_batch /tmp/cap_$1.bin $4 2>&1 >/dev/null | grep "timing\|pdsch stage" | sed "s/^/$1 RB scan_batch: /"; done
  ./scan_cpu /tmp/cap_$1.bin $4 2>&1 >/dev/null | grep timing | sed "s/^/$1 RB scan_cpu  : /"
done
