#!/bin/bash
cd /root/repo
o=gpurun_out/r04k; mkdir -p $o
for t in 64 128 192 256; do echo "== MI_LTE_PUSCH_THREADS=$t (one binary, __launch_bounds__(256))"; MI_LTE_PUSCH_THREADS=$t timeout 120 python tools/ab/bench_kernels.py uplink --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1; done | tee $o/pusch_threads_runtime.txt
timeout 600 python -m pytest tests/test_uplink_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x -p no:cacheprovider -k "uplink or ul or pusch" 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee $o/pytest_uplink.txt
