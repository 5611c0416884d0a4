#!/bin/bash
cd /root/repo
o=gpurun_out/r04i; mkdir -p $o
timeout 900 python -m pytest tests/test_turbo_gpu.py tests/test_chain_gpu.py tests/test_uplink_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_subset.txt
for i in 1 2; do timeout 300 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
