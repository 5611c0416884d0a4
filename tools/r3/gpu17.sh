#!/bin/bash
# round 3, call 17: the de-mappers on the host libm's atan2f -- the soak's one differing case, seed 17 again, the whole suite
cd /root/repo
mkdir -p gpurun_out/r03t
timeout 400 python tools/r3/repro_seed17.py 2>&1 | grep -v "^ERROR: DCI" | tail -3 | tee gpurun_out/r03t/repro_seed17.txt
MI_LTE_FUZZ_SEED=17 timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -2 | tee gpurun_out/r03t/seed17.txt
SECONDS=0
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee gpurun_out/r03t/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a gpurun_out/r03t/pytest_gpu.txt
timeout 600 python tools/ab/bench_kernels.py chain --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/r03t/bench_kernels.txt
