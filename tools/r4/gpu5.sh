#!/bin/bash
# round 4, call 5: the transform with hand-selected packed products and scalar-base addressing
cd /root/repo
o=gpurun_out/r04e; mkdir -p $o
timeout 600 python -m pytest tests/test_frontend_gpu.py tests/test_uplink_gpu.py tests/test_sync_gpu.py tests/test_prach_gpu.py tests/test_dropin_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -5 | tee $o/pytest_subset.txt
for w in frontend chain uplink; do timeout 300 python tools/ab/bench_kernels.py $w --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
