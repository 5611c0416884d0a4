#!/bin/bash
# k_bcjr_final rewritten (rows fetched together), k_bcjr_prep8 (eight blocks per workgroup: whole lines): parity, then W3 in both BCJR modes
cd /root/repo
o=gpurun_out/r04y; mkdir -p $o; rm -f $o/bench_turbo_bcjr.txt
timeout 900 python -m pytest tests/test_bcjr_gpu.py tests/test_turbo_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_bcjr.txt
for p1 in 1 0 1 0; do for d in bcjr_early; do echo "MI_LTE_BCJR_PREP1=$p1" | tee -a $o/bench_turbo_bcjr.txt; MI_LTE_BCJR_PREP1=$p1 timeout 200 python tools/ab/bench_kernels.py turbo --decoder $d --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_turbo_bcjr.txt; done; done
timeout 200 python tools/ab/bench_kernels.py turbo --decoder bcjr --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $o/bench_turbo_bcjr.txt
