#!/bin/bash
# which HIP calls the per-call scan spends its time in (1.4 MHz and 20 MHz captures)
cd /root/repo/shim/_build; export TMPDIR=/tmp
for cfg in "6 17 30 1.92" "100 77 12 30.72"; do
  set -- $cfg
  ./capture_gen /tmp/cap_$1.bin $1 $2 $3 > /dev/null 2>&1
  rm -rf /tmp/tr; rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d /tmp/tr -o t -- ./scan_gpu /tmp/cap_$1.bin $4 > /dev/null 2>/tmp/tr.err
  echo "== $1 RB: HIP API"; f=$(find /tmp/tr -name "*hip_api_stats.csv" | head -1); head -12 $f | cut -d, -f1-6
  echo "== $1 RB: kernels"; f=$(find /tmp/tr -name "*kernel_stats.csv" | head -1); head -16 $f | cut -d, -f1-6
done
