#!/bin/bash
# k_dl_fft2k knock-outs: what the 8 x 16 x 16 transform waits for
cd /root/repo
o=gpurun_out/r04u; mkdir -p $o
AB_TIMEOUT=150 tools/ab/run_variants.sh frontend --steps 20 --warmup 5 2>&1 | tee $o/variants_fft2k_knockouts.txt
