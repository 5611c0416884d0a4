#!/bin/bash
cd "$(dirname "$0")/../shim/_build"
./capture_gen /tmp/cap_25.bin 25 301 24 > /dev/null 2>&1
./capture_gen /tmp/cap_100.bin 100 77 12 > /dev/null 2>&1
for e in "" "HSA_ENABLE_INTERRUPT=0" "GPU_MAX_HW_QUEUES=1"; do
 for c in "25 7.68" "100 30.72"; do set -- $c
  for i in 1 2 3; do env $e ./scan_gpu /tmp/cap_$1.bin $2 2>&1 >/dev/null | grep timing | sed "s/.*per-subframe loop[^)]*)//" | sed "s/^/[$e] $1 RB:/"; done
 done
done
