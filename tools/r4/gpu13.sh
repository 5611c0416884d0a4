#!/bin/bash
cd /root/repo
o=gpurun_out/r04l; mkdir -p $o
for t in 256 128 192 256; do echo "== MI_LTE_PDSCH_THREADS=$t"; MI_LTE_PDSCH_THREADS=$t timeout 120 python tools/ab/bench_kernels.py chain --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1; done | tee $o/pdsch_threads_runtime.txt
