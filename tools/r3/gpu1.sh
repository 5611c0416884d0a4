#!/bin/bash
# round 3, call 1: the whole GPU suite with the new differential fuzz tests, per-test durations
cd /root/repo
mkdir -p gpurun_out/r03a
SECONDS=0
timeout 1500 python -m pytest tests -m gpu -q --durations=15 -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -60 > gpurun_out/r03a/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" >> gpurun_out/r03a/pytest_gpu.txt
tail -45 gpurun_out/r03a/pytest_gpu.txt
cp gpurun_out/fuzz_report.json gpurun_out/r03a/ 2>/dev/null
