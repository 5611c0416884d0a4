#!/bin/bash
# round 3, call 16: the one-call-per-subframe form with the plan's descriptors in mapped host memory
cd /root/repo
mkdir -p gpurun_out/r03r
timeout 600 python -m pytest tests/test_dropin_gpu.py -m gpu -q -x -k "one_call" -p no:cacheprovider 2>&1 | tail -3
timeout 600 python tools/r3/subframe_call_timing.py 2>&1 | tail -6 | tee gpurun_out/r03r/subframe_call_timing.txt
