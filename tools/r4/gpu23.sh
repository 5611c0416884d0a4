#!/bin/bash
# k_dl_fft2k: symbols per workgroup (MI_LTE_FFT_SPW), same library, same box
cd /root/repo
o=gpurun_out/r04t; mkdir -p $o
for spw in 1 3 5 15 1; do
  echo "== spw $spw"
  MI_LTE_FFT_SPW=$spw timeout 150 python tools/ab/bench_kernels.py frontend --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1
done | tee $o/fft_spw.txt
MI_LTE_FFT_SPW=5 timeout 600 python -m pytest tests/test_frontend_gpu.py tests/test_sync_gpu.py tests/test_uplink_gpu.py -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v "^ERROR: DCI" | tail -4 | tee $o/pytest_spw5.txt
