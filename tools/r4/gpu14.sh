#!/bin/bash
cd /root/repo
o=gpurun_out/r04m; mkdir -p $o
timeout 600 python -m pytest tests/test_bcjr_gpu.py tests/test_chain_gpu.py -m gpu -q -x -p no:cacheprovider -k "bcjr or BCJR" 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_bcjr.txt
for d in bcjr bcjr_early; do timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
for d in bcjr bcjr_early; do MI_LTE_BCJR_LAUNCH_PER_HALF=1 timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | sed 's/^/launch-per-half: /' | tee -a $o/bench_kernels.txt; done
for d in bcjr bcjr_early; do timeout 300 python tools/ab/bench_kernels.py chain --decoder $d --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_kernels.txt; done
