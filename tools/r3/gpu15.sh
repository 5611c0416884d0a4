#!/bin/bash
# round 3, call 15: the whole suite after the argument-validation changes (chain / uplink / pdcch / synth), with the scaled-up PBCH / PRACH fuzz
cd /root/repo
mkdir -p gpurun_out/r03q
SECONDS=0
timeout 1200 python -m pytest tests -m gpu -q -x --durations=6 -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -14 | tee gpurun_out/r03q/pytest_gpu.txt
echo "pytest -m gpu: $SECONDS s" | tee -a gpurun_out/r03q/pytest_gpu.txt
cp gpurun_out/fuzz_report.json gpurun_out/r03q/
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
