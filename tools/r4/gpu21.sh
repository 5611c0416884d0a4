#!/bin/bash
cd /root/repo
o=gpurun_out/r04q; mkdir -p $o
timeout 900 python -m pytest tests/test_prach_gpu.py tests/test_fuzz_gpu.py tests/test_uplink_gpu.py tests/test_dropin_gpu.py tests/test_cabi.py -m gpu -q -x -p no:cacheprovider -k "prach or uplink or cabi or dropin" 2>&1 | grep -v "^ERROR: DCI" | tail -8 | tee $o/pytest_prach.txt
timeout 200 python tools/ab/bench_kernels.py uplink --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee $o/bench_uplink.txt
