#!/bin/bash
# round 3, call 3: the new product-level forms (dynamic plans, multi-device pipeline, capture split, threads, freq_shift, C++ batch scanner), then the whole suite
cd /root/repo
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_pipeline_gpu.py tests/test_dropin_gpu.py -m gpu -q -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -40 | tee gpurun_out/r03c/pytest_new.txt
SECONDS=0
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -8 | tee gpurun_out/r03c/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a gpurun_out/r03c/pytest_gpu.txt
cp gpurun_out/scan_batch_timing_*.txt gpurun_out/r03c/ 2>/dev/null
cat gpurun_out/scan_batch_timing_*.txt 2>/dev/null
