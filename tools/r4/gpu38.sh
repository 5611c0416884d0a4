#!/bin/bash
cd /root/repo
o=gpurun_out/r04x; mkdir -p $o
for sd in 91 0; do MI_LTE_FUZZ_SEED=$sd timeout 600 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider -k prach 2>&1 | grep -v "^ERROR: DCI" | tail -3; python -c "
import json; print(json.load(open('gpurun_out/fuzz_report.json'))['prach'])"; done | tee $o/prach_seed91_fixed.txt
timeout 600 python -m pytest tests/test_prach_gpu.py tests/test_dropin_gpu.py tests/test_uplink_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -3
