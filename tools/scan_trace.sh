#!/bin/bash
# which HIP calls the per-call scan spends its time in (1.4 MHz capture)
cd /root/repo/shim/_build; export TMPDIR=/tmp
./capture_gen /tmp/cap_6.bin 6 17 30 > /dev/null 2>&1
rm -rf /tmp/tr; rocprofv3 --hip-trace --stats --output-format csv -d /tmp/tr -o t -- ./scan_gpu /tmp/cap_6.bin 1.92 > /dev/null 2>/tmp/tr.err
f=$(find /tmp/tr -name "*hip_api_stats.csv" | head -1); head -25 $f | cut -d, -f1-6
