#!/bin/bash
cd /root/repo
o=gpurun_out/r04h; mkdir -p $o
timeout 600 python -m pytest tests/test_bcjr_gpu.py tests/test_chain_gpu.py -m gpu -q -x -p no:cacheprovider -k "bcjr or BCJR" 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_bcjr.txt
for d in bcjr bcjr_early; do timeout 300 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
timeout 300 python tools/ab/bench_kernels.py chain --decoder bcjr --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt
