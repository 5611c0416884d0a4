#!/bin/bash
# per-call (shim) path: the scanner's phase timings on each bandwidth, three runs each (tests/test_dropin_gpu.py keeps one run's)
cd /root/repo/shim/_build
for cfg in "6 17 30 1.92" "25 301 24 7.68" "100 77 12 30.72"; do
  set -- $cfg
  ./capture_gen /tmp/cap_$1.bin $1 $2 $3 > /dev/null 2>&1
  for i in 1 2 3; do ./scan_gpu /tmp/cap_$1.bin $4 2>&1 >/dev/null | grep timing | sed "s/^/$1 RB gpu: /"; done
  ./scan_cpu /tmp/cap_$1.bin $4 2>&1 >/dev/null | grep timing | sed "s/^/$1 RB cpu: /"
done
