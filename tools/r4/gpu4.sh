#!/bin/bash
# round 4, call 4: SISO forward pass with 67 instead of 70 instructions per step; the new refusal tests
cd /root/repo
o=gpurun_out/r04d; mkdir -p $o
timeout 600 python -m pytest tests/test_turbo_gpu.py tests/test_pipeline_gpu.py tests/test_dropin_gpu.py tests/test_args_gpu.py tests/test_bcjr_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -15 | tee $o/pytest_subset.txt
for i in 1 2; do timeout 300 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
