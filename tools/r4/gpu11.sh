#!/bin/bash
cd /root/repo
o=gpurun_out/r04j; mkdir -p $o
AB_TIMEOUT=120 bash tools/ab/run_variants.sh uplink --steps 10 --warmup 3 2>&1 | tee $o/variants_pusch_threads.txt
